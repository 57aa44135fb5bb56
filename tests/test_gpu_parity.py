"""Parity of the gfx950 library (through the C ABI) with the oracle and the reference's
golden traces.  Run on the GPU box: python -m pytest tests -m gpu."""
import random

import mpmath
import pytest

from sdpb_amd.solver import SDPSolver
from tests import libs, parity

pytestmark = pytest.mark.gpu


def _solver(sdp, precision, params=None):
    return SDPSolver(sdp, precision, params or {}, lib_path=libs.product_lib())


# ---- arithmetic: device ops vs GMP mpf (oracle), tolerance 2 ulp of the device mantissa
def _check_arithmetic(s, o, dense, count, seed):
    """add, sub, mul, div, sqrt of the device against GMP at four times the precision.  dense: every limb of both operands is
    random (round 6: the 40-digit operands of the earlier rounds have 133 significant bits -- the low limbs of both are zero,
    and tests.parity compares at 1400 bits, so above 43 limbs the old test proved 1400 bits, not 2 ulp).  Returns the worst
    log2 relative error per operation, compared at four times the device mantissa."""
    bits = 32 * s.limbs
    digits = int(bits * 0.30103) + 12
    rng = random.Random(seed)
    worst = {}

    def number():
        if not dense:
            return mpmath.nstr(mpmath.mpf(rng.uniform(-1, 1)) * mpmath.mpf(10) ** rng.randint(-40, 40), 40)
        body = rng.choice("123456789") + "".join(rng.choice("0123456789") for _ in range(digits - 1))
        return ("-" if rng.random() < 0.5 else "") + "0." + body + "e" + str(rng.randint(-40, 40))

    with mpmath.workprec(4 * bits):
        for _ in range(count):
            sa, sb = number(), number()
            for op in ("add", "sub", "mul", "div", "sqrt"):
                xa = sa.lstrip("-") if op == "sqrt" else sa
                got, want = s.op_scalar(op, xa, sb), o.scalar_op(op, xa, sb)
                err = parity.log2_rel(got, want)
                assert err <= -(bits - 2), (op, err, sa[:30], sb[:30])
                worst[op] = max(worst.get(op, float("-inf")), err)
    return worst


@pytest.mark.parametrize("precision", [128, 256, 512, 768, 1024, 1280, 1536, 2048])
def test_device_arithmetic_matches_mpf(precision):
    from oracle.oracle import Oracle
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    # oracle computes at a much higher precision: exact reference values
    o = Oracle(sdp, 4 * 32 * s.limbs + 256)
    _check_arithmetic(s, o, dense=False, count=40, seed=precision)
    worst = _check_arithmetic(s, o, dense=True, count=16, seed=precision + 1)
    # the comparison really sees the last limb: a dense result cannot agree with the exact value beyond its own mantissa
    assert all(-(32 * s.limbs + 40) < w for w in worst.values()), worst
    s.close()
    o.close()


# ---- the syrk_Q stage as an operator: calculate_matrix_square.test.cxx recipe + saturated columns
@pytest.mark.parametrize("precision", [128, 256, 400, 512, 664, 768, 1024, 1280, 1536, 2048])
def test_syrk_Q_stage_and_saturated_columns(precision):
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    worst = parity.check_syrk_Q(s, precision, rows=70, cols=37)
    assert worst <= -(precision - 40), worst
    s.close()


@pytest.mark.parametrize("precision,rows", [(512, 2560), (400, 1000), (768, 600), (1024, 600), (256, 600), (2048, 300)])
def test_int_syrk_with_every_entry_at_the_largest_magnitude_over_one_sweep(precision, rows, monkeypatch):
    """parity.check_int_syrk_extremes: one row split (at 512 bits the longest sweep the host ever launches, 2560 rows), every
    limb of every piece at its maximum: the inputs that decide whether the carry schedule of the column sums holds."""
    monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", "1")
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    parity.check_int_syrk_extremes(s, rows, 37)
    s.close()


# ---- BASELINE.json config C1 at its stated --precision 128 (whole iterations at 6 limbs)
def test_config_C1_at_its_stated_precision_128():
    assert parity.check_c1_at_precision_128(libs.product_lib()) <= -64


# ---- above 1024 bits (the reference accepts any --precision, Solver_Parameters.cxx:20-26)
@pytest.mark.parametrize("precision,limbs", [(1100, 42), (1280, 42), (1536, 50), (1700, 66), (2048, 66)])
def test_iterations_above_1024_bits(precision, limbs):
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d-constraints")
    o = Oracle(sdp, precision, meta["params"], param_prec=64)
    s = _solver(sdp, precision, parity.reference_params(meta["params"], o))
    assert s.limbs == limbs
    for it in range(5):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=precision // 2)
        assert not bad, (it + 1, bad)
    s.close()
    o.close()


# ---- sdpb's DEFAULT --precision 400 (16 limbs; the fixed-point image padded from 14 to 16 limbs takes the Toom-4 kernel)
def test_iterations_at_the_default_precision_400():
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("singlet_cT")
    o = Oracle(sdp, 400, meta["params"], param_prec=64)
    s = _solver(sdp, 400, parity.reference_params(meta["params"], o))
    assert s.limbs == 16 and s.fx_frac_bits == 509   # (Toom-5 x Karatsuba image: round 6; 487 with -DSDPB_SYRK_NO_TOOM5K)
    for it in range(12):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=200)
        assert not bad, (it + 1, bad)
    s.close()
    o.close()


# ---- the smallest eigenvalue behind the step lengths, on spectra clustered like those of a run that converges
@pytest.mark.parametrize("precision", [128, 256, 400, 512, 768, 1024, 1536, 2048])
def test_min_eigenvalue_of_clustered_spectra(precision):
    from oracle.oracle import Oracle
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    o = Oracle(sdp, 2 * precision + 256)
    worst = parity.check_min_eigenvalue(s, o, s.limbs, n=40)
    print(f"lambda_min at {precision} bits ({s.limbs} limbs): worst 2^{worst:.1f} of the largest entry")
    s.close()
    o.close()


def test_precision_beyond_the_compiled_widths_is_a_clear_error():
    from sdpb_amd.solver import SDPBError
    sdp, _, _, _ = parity.load_case("1d")
    with pytest.raises(SDPBError, match="built for 128 ... 2048 bits"):
        _solver(sdp, 2100)


# ---- the dominant kernel: fixed-point syrk is bit exact (integers)
@pytest.mark.parametrize("precision,rows,cols,splits", [(128, 37, 21, None), (512, 300, 50, None), (512, 9, 1, None),
                                                        (1024, 64, 33, None), (512, 300, 50, "4"), (256, 100, 40, "16"),
                                                        (768, 130, 40, "3"), (1024, 200, 33, "2"), (1280, 64, 33, None),
                                                        (1536, 90, 20, "2"), (400, 50, 20, None), (2048, 90, 20, "2"), (2048, 64, 33, None),
                                                        (664, 64, 33, "2"),
                                                        # k_syrk_fx3 (32 x 32 tiles, 2 x 2 outputs per lane): odd widths (pair loads at
                                                        # 8-byte alignment, second column of the last pair past N), every quadrant mask
                                                        (512, 200, 81, None), (512, 333, 113, "3"), (400, 70, 47, None), (512, 90, 96, "2"),
                                                        (1024, 100, 81, None), (1024, 150, 47, "3"), (768, 120, 81, None), (664, 90, 49, "2")])
def test_int_syrk_bit_exact(precision, rows, cols, splits, monkeypatch):
    from oracle.oracle import Oracle
    if splits:
        monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", splits)  # row-split partial sums + k_syrk_reduce
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    o = Oracle(sdp, precision)
    fxbits = s.fx_frac_bits   # kernels.hpp: fx_frac_bits (bias + carry-free Karatsuba sums)
    rng = random.Random(rows * cols)
    vals = [rng.randrange(-(2 ** fxbits) + 1, 2 ** fxbits) for _ in range(rows * cols)]
    vals[0] = 0
    vals[-1] = 2 ** fxbits - 1
    vals[1] = -(2 ** fxbits) + 1
    vals[2] = 2 ** (fxbits // 2)
    vals[3] = -1
    got = s.op_int_syrk(rows, cols, vals)
    want = o.int_syrk(rows, cols, vals)
    for j in range(cols):
        for i in range(j, cols):
            assert got[i + j * cols] == want[j + i * cols], (i, j)
    s.close()


@pytest.mark.parametrize("precision,rows,cols,splits,budget", [(512, 700, 200, "4", 4.0e6), (512, 333, 113, None, 1.5e6), (256, 300, 90, "4", 2.0e5),
                                                               (1024, 150, 81, "3", 3.0e6), (1280, 200, 50, "2", 1.0e6), (768, 120, 81, "2", 2.0e6)])
def test_int_syrk_in_chunks_under_a_memory_budget_is_bit_exact(precision, rows, cols, splits, budget, monkeypatch):
    """The analogue of the reference's output windows (bigint_syrk_blas.cxx:200-220, BigInt_Shared_Memory_Syrk_Context.cxx:149-215,
    --maxSharedMemory): a budget far below the partial planes of the whole Q' (row splits x limb planes x tile-packed lower
    triangle) makes Solver::syrk_G walk the output tiles in chunks through ONE bounded buffer; every entry stays bit-exact
    against GMP."""
    from oracle.oracle import Oracle
    if splits:
        monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", splits)
    monkeypatch.setenv("SDPB_HIP_SYRK_PART_BYTES", str(int(budget)))
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    o = Oracle(sdp, precision)
    fxbits = s.fx_frac_bits
    rng = random.Random(rows + cols)
    vals = [rng.randrange(-(2 ** fxbits) + 1, 2 ** fxbits) for _ in range(rows * cols)]
    vals[0], vals[1], vals[-1] = 0, -(2 ** fxbits) + 1, 2 ** fxbits - 1
    got = s.op_int_syrk(rows, cols, vals)
    want = o.int_syrk(rows, cols, vals)
    call = s.memory_plan()["last_syrk_call"]
    assert call["chunks"] >= 3 and call["partial_bytes"] <= budget, call
    for j in range(cols):
        for i in range(j, cols):
            assert got[i + j * cols] == want[j + i * cols], (i, j)
    s.close()


# (precision, rows, cols, forced row splits, image budget, partial-plane budget or None)
@pytest.mark.parametrize("precision,rows,cols,splits,image_budget,part_budget",
                         [(512, 700, 200, None, 9.0e6, None), (512, 333, 113, "3", 2.0e6, 1.5e6), (256, 300, 90, "4", 4.5e5, None),
                          (1024, 150, 81, None, 1.8e6, 3.0e6), (1280, 200, 50, "2", 6.0e5, None), (768, 250, 81, None, 1.4e6, None),
                          (2048, 90, 20, None, 1.5e5, None), (400, 130, 47, None, 3.0e5, None)])
def test_int_syrk_with_the_image_in_row_windows_is_bit_exact(precision, rows, cols, splits, image_budget, part_budget, monkeypatch):
    """The analogue of the reference's INPUT windows (BigInt_Shared_Memory_Syrk_Context.cxx:70-110,172-186:
    input_window_split_factor; bigint_syrk_blas.cxx:239-285 loops over them): an image budget far below the fixed-point image of
    all rows makes Solver::syrk_G_windows build the image for one row window at a time in ONE bounded buffer and add each
    window's product into Q' (k_acc_add_tri) -- every entry stays bit-exact against GMP, alone and together with chunked
    output windows; so does the whole syrk_Q stage (norms, normalise-and-shift per window, diagonal check, restore)."""
    from oracle.oracle import Oracle
    if splits:
        monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", splits)
    monkeypatch.setenv("SDPB_HIP_SYRK_IMAGE_BYTES", str(int(image_budget)))
    if part_budget:
        monkeypatch.setenv("SDPB_HIP_SYRK_PART_BYTES", str(int(part_budget)))
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    o = Oracle(sdp, precision)
    fxbits = s.fx_frac_bits
    rng = random.Random(rows + 7 * cols)
    vals = [rng.randrange(-(2 ** fxbits) + 1, 2 ** fxbits) for _ in range(rows * cols)]
    vals[0], vals[1], vals[-1], vals[-2] = 0, -(2 ** fxbits) + 1, 2 ** fxbits - 1, -(2 ** fxbits) + 1
    got = s.op_int_syrk(rows, cols, vals)
    want = o.int_syrk(rows, cols, vals)
    plan = s.memory_plan()
    assert plan["image"]["last_call_windows"] >= 3, plan["image"]
    if part_budget:
        assert plan["last_syrk_call"]["chunks"] >= 2 and plan["last_syrk_call"]["partial_bytes"] <= part_budget, plan["last_syrk_call"]
    for j in range(cols):
        for i in range(j, cols):
            assert got[i + j * cols] == want[j + i * cols], (i, j)
    assert parity.check_syrk_Q(s, precision, rows=min(rows, 120), cols=min(cols, 40)) <= -(precision - 40)
    s.close()
    o.close()


@pytest.mark.parametrize("limbs,precision", [(6, 128), (10, 256), (16, 448), (18, 512), (24, 704), (26, 768), (34, 1024), (42, 1280), (50, 1536), (66, 2048)])
def test_Q_image_keeps_at_least_precision_minus_32_bits(limbs, precision):
    """The floor under the fixed-point image of P' the exact integer Q' = P'^T P' is formed from, at the widest --precision
    every compiled limb count serves (the reference truncates P' at 2^precision: Matrix_Normalizer.cxx:174-192,
    compute_Q.cxx:107).  Round-4 review: the image went 505 -> 495 -> 487 bits at --precision 512 as multiplication levels
    were added; one more guard bit below precision - 32 fails here.  A user who needs >= p bits in Q' passes --precision p + 64
    (INTEGRATION.md section 3)."""
    sdp, _, _, _ = parity.load_case("1d")
    s = _solver(sdp, precision)
    assert s.limbs == limbs
    assert s.fx_frac_bits >= precision - 32, (precision, s.fx_frac_bits)
    wider = _solver(sdp, precision + 64) if precision + 64 <= 2048 else None
    if wider is not None:
        assert wider.fx_frac_bits >= precision, (precision + 64, wider.fx_frac_bits)   # the documented way to a full-width image
        wider.close()
    s.close()


# ---- whole iterations vs the reference's golden traces (reference tolerance 2^-99)
GOLDEN = [("1d", None), ("1d-old-sampling", 40), ("1d-duplicate-poles", 40), ("1d-constraints", None),
          ("dfibo", None), ("singlet_cT", None), ("singlet_allowed_primal_jump", None),
          ("singlet_allowed_dual_jump", None)]


@pytest.mark.parametrize("name,limit", GOLDEN)
def test_gpu_matches_reference_golden(name, limit):
    from oracle.oracle import Oracle
    sdp, meta, iters, out = parity.load_case(name)
    o = Oracle(sdp, meta["precision"], meta["params"], param_prec=64)
    s = _solver(sdp, meta["precision"], parity.reference_params(meta["params"], o))
    n = len(iters) if limit is None else min(limit, len(iters))
    worst = float("-inf")
    for rec in iters[:n]:
        assert not s.iterate(), (name, rec["iteration"], s.terminate_reason)
        bad, w = parity.compare_iteration(s.scalars(), rec)
        worst = max(worst, w)
        assert not bad, f"{name} iteration {rec['iteration']}: {bad}"
    print(f"{name}: {n} iterations, worst log2 relative difference to the reference trace {worst:.1f}")
    if limit is None:
        assert s.iterate(), f"{name}: did not terminate after {n} iterations"
        assert s.terminate_reason == out["terminateReason"]
        for key in ("primalObjective", "dualObjective"):
            assert parity.log2_rel(s.scalar(key), out[key]) <= -99, key
    s.close()


# ---- vs the oracle at the headline precision on a synthetic SDP (SURVEY.md §8d gate:
# first 10 iterations within 2^-(p/2))
@pytest.mark.parametrize("cfg,scale,iters", [("C2", 1.0, 10), ("C3", 0.02, 10), ("C4", 0.01, 6), ("C5", 0.002, 5)])
def test_gpu_matches_oracle_on_synthetic(cfg, scale, iters):
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import config, make_sdp
    c = config(cfg, scale)
    sdp = make_sdp(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"])
    p = c["precision"]
    s = _solver(sdp, p, parity.DEFAULT_PARAMS)
    o = Oracle(sdp, p, parity.DEFAULT_PARAMS, param_prec=0)
    for it in range(iters):
        ts, to = s.iterate(), o.iterate()
        assert ts == to
        if ts:
            break
        got, want = s.scalars(), o.scalars()
        bad, _ = parity.compare_iteration(got, want, tol_bits=p // 2)
        assert not bad, f"{cfg} iteration {it + 1}: {bad}"
    s.close()


def test_errors_name_the_block_like_the_reference():
    from sdpb_amd.solver import SDPBError
    sdp, meta, _, _ = parity.load_case("1d")
    s = _solver(sdp, meta["precision"])
    n = len(s.array("X", 0, 0))
    s.set_array("X", ["-1"] + ["0"] * (n - 1), 0, 0)  # not positive definite
    with pytest.raises(SDPBError) as e:
        s.iterate()
    assert e.value.code == 1
    assert "Block_Diagonal_Matrix X, block index = 0, parity = 0" in str(e.value)
    s.close()


def test_rccl_callbacks_alias_device_memory_zero_copy():
    """The multi-GPU exchange path hands raw device pointers to torch.distributed (backend
    nccl = RCCL).  With a 1-rank group a SUM all-reduce and an all-gather must leave the
    bytes intact — and the aliasing tensor must really be the library's memory."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from sdpb_amd.distributed import make_collectives, tensor_from_pointer

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        allreduce, allgather = make_collectives(dev)
        a = torch.arange(4096, dtype=torch.int64, device=dev) * 1234567
        view = tensor_from_pointer(a.data_ptr(), a.numel() * 8, dev)
        view[0] = 77  # writes through to `a`: zero copy
        torch.cuda.synchronize()
        assert int(a[0].item() & 0xFF) == 77
        ref = a.clone()
        assert allreduce(a.data_ptr(), a.numel()) == 0
        assert torch.equal(a, ref)
        send = torch.randint(0, 255, (1000,), dtype=torch.uint8, device=dev)
        recv = torch.zeros(1000, dtype=torch.uint8, device=dev)
        assert allgather(send.data_ptr(), recv.data_ptr(), 1000) == 0
        assert torch.equal(send, recv)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_measured_copy_ceiling_is_sane():
    """sdpb_hip_copy_bandwidth (the measured HBM ceiling bench.py reports next to the 8 TB/s peak)."""
    from sdpb_amd.solver import copy_bandwidth_gbs
    gbs = copy_bandwidth_gbs(1 << 28, 3)
    assert 500.0 < gbs < 8000.0, gbs


@pytest.mark.gpu
def test_c_minus_By_matches_the_oracle_arithmetic():
    """c - B y of the current iterate (save_c_minus_By.hxx:18-47) against a direct evaluation."""
    sdp, meta, iters, out = parity.load_case("1d-constraints")
    s = _solver(sdp, meta["precision"], meta["params"])
    for _ in range(3):
        assert not s.iterate()
    y = [parity.mpmath.mpf(v) for v in s.array("y")]
    for j, blk in enumerate(sdp.blocks):
        got = s.array("c_minus_By", j)
        for p_, (crow, brow) in enumerate(zip(blk.c, blk.B)):
            want = parity.mpmath.mpf(crow) - sum(parity.mpmath.mpf(b) * yy for b, yy in zip(brow, y))
            assert abs(parity.mpmath.mpf(got[p_]) - want) <= parity.mpmath.mpf(2) ** -(meta["precision"] - 40) * (1 + abs(want))
    s.close()
