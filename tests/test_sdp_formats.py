"""The SDP input flavours `pmp2sdp` can produce (docs/SDPB_input_format.md): plain directory or
archive (zip — the reference's own test/data/sdp.zip —, tar.*), block data as JSON or as the
Boost-binary `.bin` that pmp2sdp writes by default.  Both flavours must load to IDENTICAL device
state (checked on the emulation build here, on the gfx950 library under -m gpu)."""
import os
import shutil
import tarfile
import zipfile

import pytest

from sdpb_amd import sdp_bin
from sdpb_amd.sdp_io import read_sdp, write_sdp
from sdpb_amd.solver import SDPSolver
from tests import libs, parity


def _same(a, b):
    assert a.J == b.J and a.b == b.b and a.constant == b.constant and a.normalization == b.normalization
    for x, y in zip(a.blocks, b.blocks):
        assert (x.dim, x.num_points, x.bases_even, x.bases_odd, x.B, x.c) == \
               (y.dim, y.num_points, y.bases_even, y.bases_odd, y.B, y.c)


def test_reference_zip_archive_reads_like_its_directory(tmp_path):
    z = os.path.join(parity.GOLDEN, "sdp.zip")
    from_zip = read_sdp(z)
    zipfile.ZipFile(z).extractall(tmp_path / "sdp")
    _same(from_zip, read_sdp(str(tmp_path / "sdp")))
    assert from_zip.J == 1 and from_zip.N == 1 and from_zip.blocks[0].num_points == 5


@pytest.mark.parametrize("mode", ["w", "w:gz", "w:xz"])
def test_tar_archives(tmp_path, mode):
    src = os.path.join(parity.GOLDEN, "1d-constraints", "sdp")
    t = tmp_path / "sdp.tar"
    with tarfile.open(t, mode) as tf:
        tf.add(src, arcname="sdp")
    _same(read_sdp(str(t)), read_sdp(src))


def test_number_records_are_exact_and_truncate_like_mpf():
    import mpmath
    for p in (128, 512, 768):
        nl = sdp_bin.num_limbs(p)
        for text in ("0", "1", "-1", "0.1", "-123456789.987654321e-40", "3e50", "1.5", "0.999999999999999999999"):
            size, exp, limbs = sdp_bin._decimal_to_record(text, p)
            back = sdp_bin._record_to_decimal(size, exp, limbs)
            old = mpmath.mp.prec
            mpmath.mp.prec = 64 * nl + 200
            try:
                a, b = mpmath.mpf(text), mpmath.mpf(back)
                assert abs(b) <= abs(a)                                   # toward zero
                assert a == b or abs(a - b) < abs(a) * mpmath.mpf(2) ** -(64 * (nl - 1))
            finally:
                mpmath.mp.prec = old
            assert sdp_bin._decimal_to_record(back, p) == (size, exp, limbs)  # idempotent


def _bin_copy(tmp_path, name):
    sdp, meta, iters, _ = parity.load_case(name)
    d = tmp_path / "sdp_bin"
    write_sdp(sdp, str(d), fmt="bin", precision=meta["precision"])
    assert all(f.endswith((".bin", ".json")) for f in os.listdir(d)) and os.path.exists(d / "block_data_0.bin")
    assert not os.path.exists(d / "block_data_0.json")
    return sdp, meta, iters, str(d)


def _device_state_identical(lib, tmp_path):
    name = "1d-constraints"
    sdp, meta, iters, d = _bin_copy(tmp_path, name)
    p = meta["precision"]
    binary = read_sdp(d, p)
    a = SDPSolver(sdp, p, meta["params"], lib_path=lib)
    b = SDPSolver(binary, p, meta["params"], lib_path=lib)
    for j in range(sdp.J):
        for which in ("c", "BT"):
            assert a.array(which, j) == b.array(which, j), (which, j)
    assert a.array("b") == b.array("b")
    for rec in iters[:4]:
        assert not a.iterate() and not b.iterate()
        assert a.scalars() == b.scalars()
        bad, _ = parity.compare_iteration(b.scalars(), rec)
        assert not bad
    a.close()
    b.close()


def test_bin_and_json_flavours_load_to_identical_device_state(tmp_path):
    _device_state_identical(libs.emu_lib(), tmp_path)


@pytest.mark.gpu
def test_bin_and_json_flavours_load_to_identical_device_state_on_the_device(tmp_path):
    _device_state_identical(libs.product_lib(), tmp_path)


def test_bin_reader_rejects_what_it_does_not_understand(tmp_path):
    _, meta, _, d = _bin_copy(tmp_path, "1d")
    p = meta["precision"]
    with open(os.path.join(d, "block_data_0.bin"), "rb") as f:
        good = f.read()
    sdp_bin.read_block_data_bin(good, p)
    with pytest.raises(sdp_bin.BinFormatError, match="Read GMP precision"):
        sdp_bin.read_block_data_bin(good, p + 128)          # SDP_Block_Data.cxx:41-43
    with pytest.raises((sdp_bin.BinFormatError, ValueError, Exception)):
        sdp_bin.read_block_data_bin(good[:-7], p)
    with pytest.raises(sdp_bin.BinFormatError):
        sdp_bin.read_block_data_bin(good + b"\0", p)
    with pytest.raises(sdp_bin.BinFormatError):
        sdp_bin.read_block_data_bin(b"\x16" + good[1:8] + b"serialization::archivX" + good[30:], p)
    with pytest.raises(ValueError, match="needs the --precision"):
        read_sdp(d)


def test_driver_solves_a_zipped_binary_sdp_like_the_json_directory(tmp_path):
    """python -m sdpb_amd.run -s sdp.zip with .bin blocks writes the same out.txt objectives."""
    from sdpb_amd import run
    name = "1d"
    sdp, meta, iters, d = _bin_copy(tmp_path, name)
    z = shutil.make_archive(str(tmp_path / "sdp"), "zip", d)
    outs = []
    for src in (os.path.join(parity.GOLDEN, name, "sdp"), z):
        out = tmp_path / ("out_" + str(len(outs)))
        argv = ["-s", src, "-o", str(out), "-c", str(out) + ".ck", "--precision", str(meta["precision"]), "--lib", libs.emu_lib(),
                "--maxIterations", "6", "--verbosity", "0"]
        for k, v in meta["params"].items():
            if k != "maxIterations":
                argv += [f"--{k}"] if v is True else [f"--{k}", str(v)]
        run.solve(argv)
        with open(out / "out.txt") as f:
            outs.append([ln for ln in f.read().splitlines() if not ln.startswith("Solver runtime")])
    assert outs[0] == outs[1]
