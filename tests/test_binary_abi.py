"""The binary number path of the C ABI (sdpb_hip_*_mpf): numbers cross as fixed-width records in
GMP's mpf_t layout.  The records used here are copied field by field from real mpf_t values held
by the oracle (oracle/sdpb_oracle.cpp: orc_get_records), i.e. what a C++ caller holding
El::BigFloat would memcpy.  CPU: emulation build; the same test runs on the gfx950 library."""
import mpmath
import numpy as np
import pytest

from oracle.oracle import Oracle
from sdpb_amd.solver import SDPSolver
from tests import libs, parity


def _value(rec):
    """mpf record -> mpmath value."""
    size = int(np.int64(rec[0]))
    exp = int(np.int64(rec[1]))
    n = abs(size)
    d = sum(int(rec[2 + i]) << (64 * i) for i in range(n))
    v = mpmath.mpf(d) * mpmath.mpf(2) ** (64 * (exp - n))
    return -v if size < 0 else v


def _check(lib):
    name = "1d-constraints"
    sdp, meta, iters, _ = parity.load_case(name)
    p = meta["precision"]
    o = Oracle(sdp, p, meta["params"], param_prec=64)
    params = parity.reference_params(meta["params"], o)
    text = SDPSolver(sdp, p, params, lib_path=lib)
    binary = SDPSolver(sdp, p, params, lib_path=lib)
    L = binary.limbs // 2 + 1                      # GMP's _mp_prec + 1 at this precision
    for j in range(sdp.J):
        binary.set_block_mpf(j, o.records("bases_even", j, limbs64=L), o.records("bases_odd", j, limbs64=L),
                             o.records("B", j, limbs64=L), o.records("c", j, limbs64=L))
    binary.set_objective_mpf(o.records("b", limbs64=L), o.records("constant", limbs64=L))
    binary.reset()
    # the uploaded constants agree with the text path to the last bits of the device mantissa
    bits = 32 * binary.limbs
    for j in range(sdp.J):
        for which in ("c", "BT"):
            a, b = text.array(which, j), binary.array(which, j)
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert parity.log2_rel(x, y) <= -(bits - 3), (which, j)
    # and the iteration fed through the binary path reproduces the reference trace
    for rec in iters[:5]:
        assert not binary.iterate()
        bad, _ = parity.compare_iteration(binary.scalars(), rec)
        assert not bad, (rec["iteration"], bad)
    # reading back: records decode to exactly the numbers the text getter prints
    ytxt = binary.array("y")
    yrec = binary.array_mpf("y")
    assert yrec.shape == (len(ytxt), 2 + L)
    for t, r in zip(ytxt, yrec):
        assert parity.log2_rel(t, _value(r)) <= -(bits - 2)   # to_decimal prints ~all digits
    # bit-exact round trip of the state through records
    for which, j, par in (("x", 0, 0), ("X", 1, 1), ("Y", 0, 0), ("y", 0, 0)):
        before = binary.array(which, j, par)
        recs = binary.array_mpf(which, j, par)
        binary.set_array_mpf(which, recs, j, par)
        assert binary.array(which, j, par) == before, which
    # short records truncate, never corrupt
    short = binary.array_mpf("y", limbs64=3)
    for t, r in zip(ytxt, short):
        assert parity.log2_rel(t, _value(r)) <= -120
    text.close()
    binary.close()
    o.close()


def test_binary_number_path_on_the_emulation_build():
    _check(libs.emu_lib())


@pytest.mark.gpu
def test_binary_number_path_on_the_device():
    _check(libs.product_lib())
