"""Host logic + kernel index arithmetic of the product sources, exercised on the CPU
emulation build (tests/emu) against the reference's golden traces.  Tolerance: the
reference's own 2^-99 (tests/parity.py).  The real gfx950 library is checked by
tests/test_gpu_parity.py."""
import pytest

from sdpb_amd.solver import SDPSolver
from tests import libs, parity

CASES = [("1d", 12), ("1d-constraints", 6), ("dfibo", None)]


@pytest.mark.parametrize("name,limit", CASES)
def test_emulated_library_matches_reference_golden(name, limit):
    sdp, meta, iters, out = parity.load_case(name)
    s = SDPSolver(sdp, meta["precision"], meta["params"], lib_path=libs.emu_lib())
    n = len(iters) if limit is None else limit
    for rec in iters[:n]:
        assert not s.iterate(), (name, rec["iteration"], s.terminate_reason)
        bad, _ = parity.compare_iteration(s.scalars(), rec)
        assert not bad, f"{name} iteration {rec['iteration']}: {bad}"
    if limit is None:
        assert s.iterate()
        assert s.terminate_reason == out["terminateReason"]
        assert parity.log2_rel(s.scalar("primalObjective"), out["primalObjective"]) <= -99
    s.close()


_INT_SYRK = [(128, 37, 21, None), (128, 70, 21, "3"), (512, 45, 18, None), (512, 100, 18, "3"), (512, 40, 47, None), (512, 70, 81, "2"),
             (512, 33, 34, None), (768, 33, 17, None), (768, 37, 45, "2"), (664, 20, 17, None), (1024, 40, 18, None), (1024, 40, 18, "2"),
             (1024, 35, 49, None), (1280, 36, 17, None), (1280, 70, 17, "2")]
# a memory budget (bytes) below the partial planes of the whole output: Q' in chunks of output tiles (Solver::syrk_plan; 32 x 32
# tiles at 512 ... 1024 bits, 16 x 16 else) -- the analogue of the reference's output windows (bigint_syrk_blas.cxx:200-220)
_INT_SYRK_CHUNKED = [(512, 70, 81, "2", 1.0e6), (512, 100, 97, None, 6.0e5), (128, 70, 41, "3", 2.5e4), (1024, 40, 49, "2", 1.6e6),
                     (1280, 70, 33, "2", 7.0e5), (768, 37, 45, "2", 1.3e6)]


@pytest.mark.parametrize("precision,rows,cols,splits,budget", [c + (None,) for c in _INT_SYRK] + _INT_SYRK_CHUNKED)
def test_emulated_int_syrk_is_exact(precision, rows, cols, splits, budget, monkeypatch):
    import random
    if splits:
        monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", splits)  # row-split partial sums + k_syrk_reduce
    if budget:
        monkeypatch.setenv("SDPB_HIP_SYRK_PART_BYTES", str(int(budget)))
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, precision, lib_path=libs.emu_lib())
    o = Oracle(sdp, precision)
    fb = s.fx_frac_bits           # 509 (Toom-5 x Karatsuba on 28-bit limbs: FX = 16), 32 FX - 25 (Toom-4 x Karatsuba: FX = 24, 32), - 17 (Toom-4: FX = 40, 48), - 7 (two Karatsuba levels: other FX % 4 == 0), else - 3
    fx = s.limbs - 2
    if fx >= 14 and fx % 4:
        fx += 4 - fx % 4      # kernels.hpp: fx_limbs — from 400 bits up the image is padded to a multiple of four limbs
    assert fb == (509 if fx == 16 else 32 * fx - (25 if fx in (24, 32) else 17 if fx in (40, 48) else 7 if fx % 4 == 0 else 3))
    rng = random.Random(7)
    vals = [rng.randrange(-(2 ** fb) + 1, 2 ** fb) for _ in range(rows * cols)]
    vals[5] = 0
    vals[11] = 2 ** fb - 1
    vals[12] = -(2 ** fb) + 1
    vals[13] = 1
    vals[14] = -1
    # exactly at / next to the split points of the image a' = v + 2^fb (Karatsuba: first and second level;
    # Toom-4: the piece boundaries at multiples of 8 fx - 4 bits)
    # (Toom-4 x Karatsuba: pieces of 8 fx - 6 bits, halves of 4 fx - 1 bits)
    for k, bit in enumerate((102, 204, 306, 408, 55, 110, 28, 83) if fx == 16 else      # Toom-5 pieces of 102 bits, halves of 55 bits, 28-bit limbs
                            (8 * fx - 6, 16 * fx - 12, 24 * fx - 18, 4 * fx - 1, 12 * fx - 7) if fx in (24, 32) else
                            (8 * fx - 4, 16 * fx - 8, 24 * fx - 12, 8 * fx - 5) if fx in (40, 48) else
                            (16 * fx - 1, 16 * fx - 3, 8 * fx - 1, 24 * fx - 4)):
        vals[15 + 2 * k] = 2 ** bit - 2 ** fb if bit < fb else 2 ** (bit - 1)
        vals[16 + 2 * k] = 2 ** bit - 2 ** fb - 1 if bit < fb else -(2 ** (bit - 1))
    got = s.op_int_syrk(rows, cols, vals)
    want = o.int_syrk(rows, cols, vals)  # upper triangle, column-major
    if budget:
        call = s.memory_plan()["last_syrk_call"]
        assert call["chunks"] > 1 and call["partial_bytes"] <= budget, call
    for j in range(cols):
        for i in range(cols):
            if i >= j:
                assert got[i + j * cols] == want[j + i * cols], (i, j)
            else:
                assert got[i + j * cols] == 0


# an image budget (bytes) below the fixed-point image of all rows: P' in row chunks through ONE bounded image buffer, each
# window's product accumulated into Q' (Solver::q_window / syrk_G_windows) -- the analogue of the reference's input windows
# (BigInt_Shared_Memory_Syrk_Context.cxx:70-110,172-186: input_window_split_factor; bigint_syrk_blas.cxx:239-285);
# (precision, rows, cols, forced row splits, image budget, partial-plane budget or None)
_INT_SYRK_WINDOWS = [(512, 100, 18, None, 1.1e5, None), (512, 140, 47, "2", 6.0e5, 1.0e6), (128, 70, 21, "3", 2.5e4, None),
                     (1024, 60, 18, None, 1.3e5, None), (1280, 70, 17, "2", 8.0e4, 3.5e5), (768, 100, 45, None, 4.0e5, None),
                     (664, 70, 17, None, 1.6e5, None)]


@pytest.mark.parametrize("precision,rows,cols,splits,image_budget,part_budget", _INT_SYRK_WINDOWS)
def test_emulated_int_syrk_with_the_image_in_row_windows_is_exact(precision, rows, cols, splits, image_budget, part_budget, monkeypatch):
    import random
    from oracle.oracle import Oracle
    if splits:
        monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", splits)
    monkeypatch.setenv("SDPB_HIP_SYRK_IMAGE_BYTES", str(int(image_budget)))
    if part_budget:
        monkeypatch.setenv("SDPB_HIP_SYRK_PART_BYTES", str(int(part_budget)))
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, precision, lib_path=libs.emu_lib())
    o = Oracle(sdp, precision)
    fb = s.fx_frac_bits
    rng = random.Random(rows * 1000 + cols)
    vals = [rng.randrange(-(2 ** fb) + 1, 2 ** fb) for _ in range(rows * cols)]
    vals[0], vals[1], vals[2], vals[-1], vals[-2] = 0, 2 ** fb - 1, -(2 ** fb) + 1, 2 ** fb - 1, -(2 ** fb) + 1
    got = s.op_int_syrk(rows, cols, vals)
    want = o.int_syrk(rows, cols, vals)  # upper triangle, column-major
    plan = s.memory_plan()
    assert plan["image"]["last_call_windows"] >= 3, plan["image"]
    if part_budget:
        assert plan["last_syrk_call"]["chunks"] > 1 and plan["last_syrk_call"]["partial_bytes"] <= part_budget, plan["last_syrk_call"]
    for j in range(cols):
        for i in range(cols):
            assert got[i + j * cols] == (want[j + i * cols] if i >= j else 0), (i, j)
    # the whole stage (norms, normalise-and-shift per window, product, diagonal check, restore) through the same windows
    assert parity.check_syrk_Q(s, precision, rows=rows, cols=cols) <= -(precision - 40)
    s.close()
    o.close()


def test_emulated_max_shared_memory_splits_both_windows_and_keeps_every_bit():
    """sdpb_hip_set_max_shared_memory bounds the image of P' (input window) and the partial planes (output window) TOGETHER,
    like --maxSharedMemory bounds the reference's two residue windows (BigInt_Shared_Memory_Syrk_Context.cxx:149-215): whole
    iterations of singlet_cT (N = 20, 322 rows) with both windows split agree with the default plan to the last bit; a
    bound below the smallest window is reported, and a non-zero bound never means 'unbounded' (round-5 advisor)."""
    sdp, meta, iters, _ = parity.load_case("singlet_cT")
    traces = []
    for bound in (0, 1_400_000, 3):
        s = SDPSolver(sdp, meta["precision"], meta["params"], lib_path=libs.emu_lib())
        if bound:
            s.set_max_shared_memory(bound)
        plan = s.memory_plan()
        if bound == 1_400_000:
            assert plan["image"]["image_chunks"] >= 3 and not plan["image"]["bound_exceeded_min_chunk"], plan["image"]
            assert plan["image"]["image_bytes"] + plan["syrk"]["partial_bytes"] <= bound, plan
            assert plan["bytes"]["P_fixed_point_image"] == plan["image"]["image_bytes"]
        elif bound == 3:
            assert plan["image"]["bound_exceeded_min_chunk"] and plan["syrk"]["bound_exceeded_min_chunk"], plan
            assert plan["syrk"]["budget_bytes"] > 0 and plan["syrk"]["chunks"] == plan["syrk"]["tiles"], plan["syrk"]
        else:
            assert plan["image"]["image_chunks"] == 1 and plan["syrk"]["chunks"] == 1
        t = []
        for _ in range(3):
            assert not s.iterate()
            t.append(s.scalars())
        if bound:
            assert s.memory_plan()["image"]["last_call_windows"] == plan["image"]["image_chunks"] >= 3
        traces.append(t)
        s.close()
    assert traces[0] == traces[1] == traces[2]
    bad, _ = parity.compare_iteration(traces[0][2], iters[2])
    assert not bad, bad


@pytest.mark.parametrize("precision", [128, 512, 664, 768, 1024, 1280])
def test_emulated_syrk_Q_stage_and_saturated_columns(precision):
    """compute_Q.cxx:94-132 as an operator (calculate_matrix_square.test.cxx recipe) incl. columns with
    a single negative entry: the normalised value saturates the fixed-point image, whose clamp must
    be 2^FB - 1 for the one-level (664 bits) AND the two-level image (the others)."""
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, precision, lib_path=libs.emu_lib())
    worst = parity.check_syrk_Q(s, precision)
    assert worst <= -(precision - 40), worst   # observed: within a few dozen bits of the full mantissa
    s.close()


def test_documented_build_without_toom4_keeps_working():
    """-DSDPB_SYRK_NO_TOOM4 (the two-level Karatsuba image at 512 bits, 505 instead of 495
    fraction bits at 3/4 of the speed) is an option a maintainer is told about: the exact product and the Q stage on
    that build (round-3 advisor: keep the fallback covered)."""
    import random
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, 512, lib_path=libs.emu_lib(variant="notoom4"))
    assert s.limbs == 18 and s.fx_frac_bits == 32 * 16 - 7
    o = Oracle(sdp, 512)
    rng = random.Random(11)
    rows, cols, fb = 37, 18, s.fx_frac_bits
    vals = [rng.randrange(-(2 ** fb) + 1, 2 ** fb) for _ in range(rows * cols)]
    vals[3], vals[4], vals[5] = 2 ** fb - 1, -(2 ** fb) + 1, 0
    got, want = s.op_int_syrk(rows, cols, vals), o.int_syrk(rows, cols, vals)
    assert all(got[i + j * cols] == want[j + i * cols] for j in range(cols) for i in range(j, cols))
    assert parity.check_syrk_Q(s, 512) <= -(512 - 40)
    s.close()
    o.close()


def test_emulated_config_C1_at_its_stated_precision_128():
    """Whole iterations at 6 limbs (the narrowest compiled width) on the shipped SDP of BASELINE.json's
    first config; the GPU twin is tests/test_gpu_parity.py::test_config_C1_at_its_stated_precision_128."""
    assert parity.check_c1_at_precision_128(libs.emu_lib()) <= -64


@pytest.mark.parametrize("name,limit", [("1d-constraints", 5), ("singlet_cT", 3), ("dfibo", None)])
def test_multi_panel_paths_with_4_column_panels(name, limit):
    """Same sources built with PB = 4: every Cholesky, triangular solve and Q solve of these small
    SDPs runs through several (ragged) panels — diagonal-block factor+inverse in LDS, panel solve,
    trailing update, blocked substitutions."""
    sdp, meta, iters, out = parity.load_case(name)
    s = SDPSolver(sdp, meta["precision"], meta["params"], lib_path=libs.emu_lib(panel=4))
    n = len(iters) if limit is None else limit
    for rec in iters[:n]:
        assert not s.iterate(), (name, rec["iteration"], s.terminate_reason)
        bad, _ = parity.compare_iteration(s.scalars(), rec)
        assert not bad, (rec["iteration"], bad)
    s.close()


@pytest.mark.parametrize("name,limit", [("singlet_cT", 3), ("dfibo", 3)])
def test_chased_cholesky_Q_with_4_column_panels(name, limit, monkeypatch):
    """Q' in two column chunks and Cholesky(Q) chasing it (SDPB_HIP_Q_CHASE=1) on one rank of the PB = 4 build: N = 20 /
    19 = five panels, the first four (one 16-column tile column) factored before the fifth is restored, then applied
    to it.  Same trace as the reference and the same bits as the one-piece schedule."""
    sdp, meta, iters, out = parity.load_case(name)
    traces = []
    for chase in ("1", "0"):
        monkeypatch.setenv("SDPB_HIP_Q_CHASE", chase)
        s = SDPSolver(sdp, meta["precision"], meta["params"], lib_path=libs.emu_lib(panel=4))
        t = []
        for rec in iters[:limit]:
            assert not s.iterate(), (name, rec["iteration"], s.terminate_reason)
            bad, _ = parity.compare_iteration(s.scalars(), rec)
            assert not bad, (rec["iteration"], bad)
            t.append(s.scalars())
        assert s.timers()["comm.q_chase"] == int(chase)
        traces.append(t)
        s.close()
    assert traces[0] == traces[1]


def test_documented_build_without_the_karatsuba_level_keeps_working():
    """-DSDPB_SYRK_NO_TOOM4K (Toom-4 alone, the round-3 kernel k_syrk_fx2<16,32,toom4> and its
    495-bit image at 512 bits) stays a working option: exact product incl. a width beyond one 16-column tile, and the Q stage."""
    import random
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, 512, lib_path=libs.emu_lib(variant="notoom4k"))
    assert s.limbs == 18 and s.fx_frac_bits == 32 * 16 - 17
    o = Oracle(sdp, 512)
    rng = random.Random(13)
    rows, cols, fb = 41, 35, s.fx_frac_bits
    vals = [rng.randrange(-(2 ** fb) + 1, 2 ** fb) for _ in range(rows * cols)]
    vals[3], vals[4], vals[5] = 2 ** fb - 1, -(2 ** fb) + 1, 0
    got, want = s.op_int_syrk(rows, cols, vals), o.int_syrk(rows, cols, vals)
    assert all(got[i + j * cols] == want[j + i * cols] for j in range(cols) for i in range(j, cols))
    assert parity.check_syrk_Q(s, 512) <= -(512 - 40)
    s.close()
    o.close()


@pytest.mark.parametrize("precision,rows", [(512, 2560), (768, 300), (1024, 300)])
def test_emulated_int_syrk_with_every_entry_at_the_largest_magnitude_over_one_sweep(precision, rows, monkeypatch):
    """CPU twin of the GPU test: parity.check_int_syrk_extremes on the emulation build (the lazy-carry mode adds into plain
    64-bit sums here as on the device, so an overflow of the carry schedule would wrap the same way)."""
    monkeypatch.setenv("SDPB_HIP_SYRK_SPLITS", "1")
    sdp, _, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, precision, lib_path=libs.emu_lib())
    parity.check_int_syrk_extremes(s, rows, 37)
    s.close()


def test_documented_build_without_toom5_keeps_working():
    """-DSDPB_SYRK_NO_TOOM5K (Toom-4 x Karatsuba with carried 96-bit column sums, the kernel of
    rounds 4-5 and its 487-bit image at 512 bits) stays a working option: exact product over several 32-column tiles and every
    quadrant mask, forced row splits, and the Q stage."""
    import random
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, 512, lib_path=libs.emu_lib(variant="notoom5k"))
    assert s.limbs == 18 and s.fx_frac_bits == 32 * 16 - 25
    o = Oracle(sdp, 512)
    rng = random.Random(17)
    rows, cols, fb = 70, 81, s.fx_frac_bits
    vals = [rng.randrange(-(2 ** fb) + 1, 2 ** fb) for _ in range(rows * cols)]
    vals[3], vals[4], vals[5] = 2 ** fb - 1, -(2 ** fb) + 1, 0
    got, want = s.op_int_syrk(rows, cols, vals), o.int_syrk(rows, cols, vals)
    assert all(got[i + j * cols] == want[j + i * cols] for j in range(cols) for i in range(j, cols))
    assert parity.check_syrk_Q(s, 512) <= -(512 - 40)
    s.close()
    o.close()


def test_emulated_library_matches_oracle_on_dim6_blocks():
    """BASELINE.json config 5 shape (m_j = 6, K_j = 2: 21 (r,s) pairs per block) at reduced size
    and precision 512: exercises the (r,s) tile decoding of pairings, Schur assembly, constraint
    sums and the Schur right-hand side against the oracle (tolerance 2^-(p/2))."""
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_sdp
    sdp = make_sdp([6] * 3, [2] * 3, 5, 512, seed=5)
    s = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib())
    o = Oracle(sdp, 512, parity.DEFAULT_PARAMS, param_prec=0)
    for it in range(4):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=256)
        assert not bad, (it + 1, bad)
    s.close()


def test_emulated_library_beyond_one_panel_of_Q():
    """N = 40 > PB = 32: two panels of Cholesky(Q) with the look-ahead schedule and its strip kernels
    (eight lanes per output), two panels of the Q solves — index arithmetic against the oracle."""
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_sdp
    sdp = make_sdp([1] * 5, [18] * 5, 40, 512, seed=11)
    s = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib())
    o = Oracle(sdp, 512, parity.DEFAULT_PARAMS, param_prec=0)
    for it in range(3):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=256)
        assert not bad, (it + 1, bad)
    s.close()
    o.close()


def test_emulated_iterations_above_1024_bits():
    """--precision 1280 (42 limbs, 16-column panels, Toom-4 syrk in two sweeps): the reference accepts any
    precision (Solver_Parameters.cxx:20-26); whole iterations of the m = 2 golden SDP against the oracle at
    the same precision, tolerance 2^-(p/2)."""
    from oracle.oracle import Oracle
    sdp, meta, _, _ = parity.load_case("1d-constraints")
    o = Oracle(sdp, 1280, meta["params"], param_prec=64)
    s = SDPSolver(sdp, 1280, parity.reference_params(meta["params"], o), lib_path=libs.emu_lib())
    assert s.limbs == 42
    for it in range(4):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=640)
        assert not bad, (it + 1, bad)
    s.close()
    o.close()


def test_emulated_five_panels_of_Q():
    """N = 150: five 32-column panels of Cholesky(Q) (the last one ragged, 22 columns) and of both sweeps of
    the Q substitution, against the oracle."""
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_sdp
    sdp = make_sdp([1] * 6, [30] * 6, 150, 512, seed=13)
    s = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib())
    o = Oracle(sdp, 512, parity.DEFAULT_PARAMS, param_prec=0)
    for it in range(2):
        assert not s.iterate() and not o.iterate()
        bad, _ = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=256)
        assert not bad, (it + 1, bad)
        assert parity.maxrel(s.array("dy"), o.array("dy")) <= -256
    s.close()
    o.close()


def test_emulated_memory_plan_and_max_shared_memory():
    """sdpb_hip_memory_plan / sdpb_hip_set_max_shared_memory (--maxSharedMemory; run.cxx:79-181,
    BigInt_Shared_Memory_Syrk_Context.cxx:149-215): with a bound below the partial planes of the whole Q' the iteration computes
    Q' in chunks of output tiles and every field of every iteration keeps its bits."""
    from sdpb_amd.synthetic import make_sdp
    sdp = make_sdp([1] * 6, [30] * 6, 150, 512, seed=13)   # N = 150: 5 x 6 / 2 = 15 tiles of 32 x 32
    a = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib())
    plan = a.memory_plan()
    assert plan["syrk"]["chunks"] == 1 and plan["syrk"]["tiles"] == 15 and plan["syrk"]["budget_source"] == "device"
    assert plan["bytes"]["syrk_partial_planes"] >= plan["syrk"]["partial_bytes"] > 0
    # tile-packed planes: the lower triangle only, about half of the N x N planes of rounds 1-4
    assert plan["syrk"]["partial_bytes_unbounded"] < 0.75 * plan["syrk"]["partial_bytes_full_square_layout"]
    assert sum(plan["bytes"].values()) > 0 and plan["bytes"]["B"] == plan["bytes"]["P"] > 0
    b = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib())
    bound = plan["syrk"]["partial_bytes"] // 4
    b.set_max_shared_memory(bound)
    pb = b.memory_plan()
    assert pb["syrk"]["budget_source"] == "maxSharedMemory" and pb["syrk"]["chunks"] >= 4, pb["syrk"]
    assert pb["syrk"]["partial_bytes"] <= bound and pb["bytes"]["syrk_partial_planes"] <= bound
    for it in range(2):
        assert not a.iterate() and not b.iterate()
        assert a.scalars() == b.scalars(), it + 1
    assert b.memory_plan()["last_syrk_call"]["chunks"] == pb["syrk"]["chunks"]
    b.set_max_shared_memory(0)   # back to the default plan
    assert b.memory_plan()["syrk"]["chunks"] == 1
    assert not a.iterate() and not b.iterate()
    assert a.scalars() == b.scalars()
    a.close()
    b.close()


# widest --precision each compiled limb count serves (limbs = 2 floor((p + 127) / 64), GMP's allocation)
TOP_PRECISION = {6: 128, 10: 256, 16: 448, 18: 512, 24: 704, 26: 768, 34: 1024, 42: 1280, 50: 1536, 66: 2048}


@pytest.mark.parametrize("limbs", [6, 18, 24, 26, 34, 42])
def test_emulated_Q_image_keeps_at_least_precision_minus_32_bits(limbs):
    """The floor under the fixed-point image of P' (round-4 review: 505 -> 495 -> 487 fraction bits at --precision 512, each
    multiplication level paid partly in guard bits; the reference truncates at 2^precision, Matrix_Normalizer.cxx:174-192,
    compute_Q.cxx:107).  A further level that takes the image below precision - 32 bits must fail here, not pass silently
    behind thresholds calibrated on the device's own error."""
    p = TOP_PRECISION[limbs]
    sdp, _, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, p, lib_path=libs.emu_lib())
    assert s.limbs == limbs
    assert s.fx_frac_bits >= p - 32, (p, s.fx_frac_bits)
    s.close()


def test_emulated_big_and_ragged_blocks_match_the_oracle():
    """CPU twin of tests/test_gpu_parity_at_size.py::test_big_and_ragged_blocks_match_the_live_oracle: m in {1, 3, 4, 6}, seven
    distinct K in one SDP, Schur blocks up to P_j = 714 (23 panels), PSD blocks up to n = 102, N = 60 -- host logic, descriptor
    tables and kernel index arithmetic at block shapes of a real mixed-correlator SDP (Block_Info.hxx:54-119)."""
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_lazy
    from tests.test_gpu_parity_at_size import RAGGED
    sdp, src = make_lazy(RAGGED["dims"], RAGGED["num_points"], RAGGED["N"], 512, RAGGED["seed"])
    s = SDPSolver(sdp, 512, parity.DEFAULT_PARAMS, lib_path=libs.emu_lib(), block_source=src)
    o = Oracle(sdp, 512, parity.DEFAULT_PARAMS, param_prec=0, block_source=src)
    for it in range(2):
        assert not s.iterate() and not o.iterate()
        bad, w = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=256)
        assert not bad and w <= -280, (it + 1, w, bad)
    s.close()
    o.close()


@pytest.mark.parametrize("precision", [128, 512, 768])
def test_emulated_min_eigenvalue_of_clustered_spectra(precision):
    from oracle.oracle import Oracle
    sdp, _, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, precision, lib_path=libs.emu_lib())
    o = Oracle(sdp, 2 * precision + 256)     # reference values at more than twice the width
    worst = parity.check_min_eigenvalue(s, o, s.limbs, n=24)
    print(f"lambda_min at {precision} bits: worst 2^{worst:.1f} of the largest entry")
    s.close()
    o.close()
