"""Parity of the gfx950 library at the sizes bench.py runs (through the C ABI).

The golden traces of the reference all have N <= 20 < PB = 32, so they never leave panel 0
of the look-ahead Cholesky(Q), run one panel of the Q solves and one 16x16 tile row of the
fixed-point syrk.  The cases here have N > 32, hundreds of blocks, row-split syrk launches
and all three streams busy — compared with the live oracle (all host cores), with the
committed full-size oracle fixtures (tests/golden/synthetic/*.json, the exact workload of
bench.py), array by array on one mixed m=1/m=2 case, and with themselves across repeated
runs (bit-identical: a stream race would show up as a run-to-run difference).
Run on the GPU box: python -m pytest tests -m gpu."""
import json
import os

import mpmath
import pytest

from sdpb_amd.solver import SDPSolver
from tests import libs, parity

pytestmark = pytest.mark.gpu
SYN = os.path.join(parity.GOLDEN, "synthetic")


def _shape(cfg, scale=1.0, **over):
    from sdpb_amd.synthetic import config
    c = config(cfg, scale)
    c.update(over)
    return c


def _pair(c, oracle=True):
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_lazy
    sdp, src = make_lazy(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"], feasible=c.get("feasible", False))
    s = SDPSolver(sdp, c["precision"], parity.DEFAULT_PARAMS, lib_path=libs.product_lib(), block_source=src)
    o = Oracle(sdp, c["precision"], parity.DEFAULT_PARAMS, param_prec=0, block_source=src) if oracle else None
    return sdp, s, o


# C5-shaped (m=6, K=2, --precision 1024) with N > PB: 1024-bit kernels incl. the FX=32 syrk variant
C5_SLICE = dict(dims=[6] * 24, num_points=[2] * 24, N=72, precision=1024, seed=5)

LIVE = [("C3 full size", _shape("C3"), 3),                      # J=600, N=100: 4 Q panels, 28 syrk tiles x 16 splits
        ("C4 x0.25", _shape("C4", 0.25), 3),                    # J=150, N=250: 8 Q panels, m=2 blocks with 4 panels
        ("C5 slice", C5_SLICE, 3),
        # the widest compiled mantissa (66 limbs, 16-column panels, one-lane roots): N = 40 is three panels of Cholesky(Q)
        ("2048-bit slice", dict(dims=[6] * 12, num_points=[2] * 12, N=40, precision=2048, seed=6), 2)]


@pytest.mark.parametrize("name,c,iters", LIVE, ids=[x[0] for x in LIVE])
def test_gpu_matches_live_oracle_beyond_one_panel(name, c, iters):
    assert c["N"] > 32
    sdp, s, o = _pair(c)
    p = c["precision"]
    worst = float("-inf")
    for it in range(iters):
        ts, to = s.iterate(), o.iterate()
        assert ts == to, (s.terminate_reason, o.terminate_reason)
        bad, w = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=p // 2)
        worst = max(worst, w)
        assert not bad, f"{name} iteration {it + 1}: {bad}"
    print(f"{name}: J={sdp.J} N={sdp.N} P_tot={sdp.P_total} p={p}: {iters} iterations, worst log2 rel diff {worst:.1f} "
          f"(oracle on {o.threads} threads)")
    s.close()
    o.close()


@pytest.mark.parametrize("precision", [400, 768, 1024])
def test_small_feasible_run_to_optimality_matches_the_live_oracle(precision):
    """The strictly feasible family (sdpb_amd/synthetic.py) at the other compiled widths, followed against the LIVE oracle until
    both stop with 'found primal-dual optimal solution' in the same iteration: the end game of a convergent run -- step
    lengths gamma / (1 - beta) from a spectrum that collapses onto one point (k_tridiag_min's shifted, multiplicity-aware
    Newton on the 16-, 26- and 34-limb ladders), block condition numbers past 2^(p/2) -- on the Toom-4 x Karatsuba syrk and
    the float triangular solves that 512 bits no longer use.  N = 40: two panels of Cholesky(Q).  Bars: 2^-(p/2), relaxed to
    cond 2^-(p-16) only where the iteration's own condition numbers pass 2^(p/2-16) (parity.conditioned_tol_bits)."""
    from oracle.oracle import Oracle
    from sdpb_amd.synthetic import make_lazy
    sdp, src = make_lazy([2] * 4 + [1] * 8, [12] * 12, 40, precision, 9, feasible=True)
    s = SDPSolver(sdp, precision, parity.DEFAULT_PARAMS, lib_path=libs.product_lib(), block_source=src)
    o = Oracle(sdp, precision, parity.DEFAULT_PARAMS, param_prec=0, block_source=src)
    worst, it, relaxed = float("-inf"), 0, 0
    while True:
        ts, to = s.iterate(), o.iterate()
        assert ts == to, (it + 1, s.terminate_reason, o.terminate_reason)
        if ts:
            break
        it += 1
        rec = o.scalars()
        tol = parity.conditioned_tol_bits(rec, precision, precision // 2)
        relaxed += tol < precision // 2
        bad, w = parity.compare_iteration(s.scalars(), rec, tol_bits=tol)
        worst = max(worst, w)
        assert not bad, (it, bad, tol)
        assert it < 400
    assert s.terminate_reason == o.terminate_reason == "found primal-dual optimal solution"
    print(f"feasible J={sdp.J} N={sdp.N} p={precision}: {it} iterations to optimality, worst log2 rel diff {worst:.1f}, "
          f"{relaxed} iterations on the conditioned bar")
    s.close()
    o.close()


def test_full_size_feasible_run_reaches_optimality_on_the_device():
    """C4's full shape (J = 600, N = 1000, P_tot = 40 000, 512 bits) from the strictly feasible family, run on the device alone
    until SDP_Solver::run stops: no oracle record exists at this size (an oracle iteration takes minutes), so the checks are
    the size-independent ones the problem offers -- the run ends with 'found primal-dual optimal solution'
    (compute_feasible_and_termination.cxx:16-64) in about as many iterations as the quarter-size fixture the oracle followed
    (160), primal and dual objective agree to the duality-gap threshold, both errors are below theirs, mu fell by more than
    60 orders of magnitude, every step length stayed in (0, 1], and a fresh solver repeats the first dozen iterations bit for
    bit."""
    c = _shape("C4f")
    assert c.get("feasible") and c["N"] == 1000 and len(c["dims"]) == 600
    keys = ("mu", "P-obj", "D-obj", "P-step", "D-step", "Q_cond_number")
    sdp, s, _ = _pair(c, oracle=False)
    it, mu0, twelfth = 0, None, None
    while not s.iterate():
        it += 1
        sc = s.scalars()
        mu0 = mpmath.mpf(sc["mu"]) if mu0 is None else mu0
        assert 0 < mpmath.mpf(sc["P-step"]) <= 1 and 0 < mpmath.mpf(sc["D-step"]) <= 1, (it, sc["P-step"], sc["D-step"])
        if it == 12:
            twelfth = [sc[k] for k in keys]
        assert it < 400
    assert s.terminate_reason == "found primal-dual optimal solution", s.terminate_reason
    assert 120 <= it <= 220, it
    po, do = mpmath.mpf(s.scalar("primalObjective")), mpmath.mpf(s.scalar("dualObjective"))
    gap = abs(po - do) / max(abs(po) + abs(do), 1)
    sc = s.scalars()
    assert gap < mpmath.mpf("1e-30") and abs(mpmath.mpf(sc["P-err"])) < mpmath.mpf("1e-30") and abs(mpmath.mpf(sc["D-err"])) < mpmath.mpf("1e-30")
    assert mpmath.mpf(sc["mu"]) < mu0 * mpmath.mpf("1e-60")
    print(f"C4f full size: {it} iterations to optimality, primal objective {mpmath.nstr(po, 20)}, relative gap {mpmath.nstr(gap, 3)}, "
          f"mu {mpmath.nstr(mu0, 3)} -> {mpmath.nstr(mpmath.mpf(sc['mu']), 3)}, "
          f"block condition 2^{float(mpmath.log(mpmath.mpf(sc['max_block_cond_number']), 2)):.0f}")
    s.close()
    sdp, s, _ = _pair(c, oracle=False)
    for _ in range(12):
        assert not s.iterate()
    assert [s.scalars()[k] for k in keys] == twelfth
    s.close()


FIXTURES = sorted(f for f in (os.listdir(SYN) if os.path.isdir(SYN) else []) if f.endswith(".json") and f != "gate_thresholds.json")


@pytest.mark.parametrize("fixture", FIXTURES)
def test_gpu_matches_oracle_fixture_at_full_size(fixture):
    """The exact workload bench.py times (full C4) against scalars the oracle produced in the build
    container (tests/golden/synthetic/make_synthetic_golden.py): the parity gate of the metric."""
    with open(os.path.join(SYN, fixture)) as f:
        fx = json.load(f)
    c = _shape(fx["config"], fx["scale"])
    assert c["seed"] == fx["seed"] and c["N"] == fx["N"] and len(c["dims"]) == fx["J"]
    sdp, s, _ = _pair(c, oracle=False)
    p = c["precision"]
    worst = float("-inf")
    for rec in fx["iterations"]:
        assert not s.iterate(), s.terminate_reason
        # 2^-(p/2), relaxed to cond 2^-(p-16) only where the iteration's own condition numbers pass 2^(p/2-16)
        bad, w = parity.compare_iteration(s.scalars(), rec, tol_bits=parity.conditioned_tol_bits(rec, p, p // 2))
        worst = max(worst, w)
        assert not bad, f"{fixture} iteration {rec['iteration']}: {bad}"
    if "terminate_reason" in fx:
        # fixtures made with `term`: the oracle's loop ended in the NEXT iteration (run.cxx:380-467; that
        # iteration has no record, as in iterations.json) — same iteration, same reason, same final objectives here
        assert len(fx["iterations"]) + 1 == fx["terminated_in_iteration"]
        assert s.iterate(), f"{fixture}: no termination in iteration {fx['terminated_in_iteration']}"
        assert s.terminate_reason == fx["terminate_reason"]
        for key in ("primalObjective", "dualObjective"):
            assert parity.log2_rel(s.scalar(key), fx[key]) <= -parity.conditioned_tol_bits(fx["iterations"][-1], p, p // 2), key
    print(f"{fixture}: J={sdp.J} N={sdp.N} P_tot={sdp.P_total}: {len(fx['iterations'])} iterations, "
          f"worst log2 rel diff {worst:.1f}" + (f", then '{fx['terminate_reason']}'" if "terminate_reason" in fx else ""))
    s.close()


def test_repeated_runs_are_bit_identical():
    """Five runs of the same 3 iterations in one process (C4 x0.25: look-ahead Cholesky(Q) on two
    streams, X/Y chains and step lengths on concurrent streams, row-split syrk) must agree to
    the last bit: every reduction has a fixed order, so any difference is an ordering bug."""
    c = _shape("C4", 0.25)
    sdp, s, _ = _pair(c, oracle=False)
    runs = []
    for rep in range(5):
        s.reset()
        trace = []
        for _ in range(3):
            assert not s.iterate()
            trace.append(s.scalars())
        trace.append(s.array("dy")[:50])
        runs.append(trace)
    for rep in range(1, 5):
        assert runs[rep] == runs[0], f"run {rep} differs from run 0"
    s.close()


def _maxrel(got, want):
    g = [mpmath.mpf(v) for v in got]
    w = [mpmath.mpf(v) for v in want]
    assert len(g) == len(w)
    scale = max(abs(v) for v in w)
    if scale == 0:
        return float("-inf") if all(v == 0 for v in g) else 0.0
    d = max(abs(a - b) for a, b in zip(g, w)) / scale
    return float(mpmath.log(d, 2)) if d > 0 else float("-inf")


def _compare_arrays(sdp, s, o, p, blocks):
    """The arrays each stage of the iteration leaves behind, device against oracle, relative to the largest entry of the
    array: localises a failure to a SURVEY §8a row.  Returns {(array, block, parity): log2 of the largest relative difference}."""
    N = sdp.N
    report = {}
    steps = (mpmath.mpf(s.scalar("P-step")), mpmath.mpf(s.scalar("D-step")))

    def lower(v, n):  # column-major n x n -> lower triangle entries
        return [v[i + j * n] for j in range(n) for i in range(j, n)]

    def upper_as_lower(v, n):  # U (column-major) read as L = U^T
        return [v[j + i * n] for j in range(n) for i in range(j, n)]

    for j in blocks:
        P = sdp.num_points[j] * sdp.dims[j] * (sdp.dims[j] + 1) // 2
        for b in (0, 1):
            for w in ("X", "Y", "AXinv", "AY", "primal_residues"):                   # a2, a3, a13, a14
                report[(w, j, b)] = _maxrel(s.array(w, j, b), o.array(w, j, b))
            # the reference scales dX, dY by the step lengths in place (step.cxx:214-216) and so does
            # the oracle; the device keeps the search direction itself                   # a10, a12
            for w, step in (("dX", steps[0]), ("dY", steps[1])):
                report[(w, j, b)] = _maxrel([mpmath.mpf(v) * step for v in s.array(w, j, b)], o.array(w, j, b))
            n = len(s.array("Xc", j, b))
            n = int(round(n ** 0.5))
            report[("Xc", j, b)] = _maxrel(lower(s.array("Xc", j, b), n), lower(o.array("Xc", j, b), n))  # a1
            report[("Yc", j, b)] = _maxrel(lower(s.array("Yc", j, b), n), lower(o.array("Yc", j, b), n))
        report[("L_S", j)] = _maxrel(lower(s.array("L", j), P), lower(o.array("L", j), P))                # a4 + a5
        pt = s.array("PT", j)                                      # N x P column-major
        po = o.array("P", j)                                       # P x N column-major
        report[("P", j)] = _maxrel([pt[n + q * N] for n in range(N) for q in range(P)], po)               # a5
        for w in ("x", "dx", "dual_residues"):                                                           # a10, a14
            report[(w, j)] = _maxrel(s.array(w, j), o.array(w, j))
    report[("chol(Q)",)] = _maxrel(lower(s.array("Q"), N), upper_as_lower(o.array("Q"), N))               # a6-a8
    for w in ("y", "dy", "primal_residue_p"):
        report[(w,)] = _maxrel(s.array(w), o.array(w))
    return report


def test_intermediate_arrays_match_the_oracle():
    """Localises a failure to a SURVEY §8a row: the arrays each stage leaves behind after two
    iterations of C4 x0.05 (m=1 and m=2 blocks, N=50 > PB, Schur blocks of 4 panels) against the
    oracle's, relative to the largest entry of the array, tolerance 2^-(p/2)."""
    c = _shape("C4", 0.05)
    sdp, s, o = _pair(c)
    p = c["precision"]
    for _ in range(2):
        assert not s.iterate() and not o.iterate()
    blocks = [0, sdp.J // 2, sdp.J - 1]      # an m=2 block, the middle, an m=1 block
    assert {sdp.dims[j] for j in blocks} == {1, 2}
    report = _compare_arrays(sdp, s, o, p, blocks)
    bad = {k: v for k, v in report.items() if v > -(p // 2)}
    print("worst arrays:", sorted(report.items(), key=lambda kv: -kv[1])[:5])
    assert not bad, bad
    s.close()
    o.close()


# Block shapes of a real mixed-correlator SDP (Block_Info.hxx:54-119: dim m = 3-6, K = 20-70 sample points) in ONE SDP: Schur
# blocks of P_j = 714 (23 panels of 32 columns), 440, 300, 240, 144, 28, 20 rows, PSD blocks of n = 102, 88, 60, 60, 36,
# 14, 10; seven distinct K.  (The synthetic sample points put the block condition number at 2^228 for K = 44, 2^260 for K = 50 and 2^366 for K = 70 -- the degree is
# kept where half of the mantissa survives, which is what the bar 2^-(p/2) asks.)  Until round 6 the device had never run a
# block above P_j = 120 / n = 40: the LDS images, the size-sorted k_tridiag list (n = 10 ... 102 in one launch), the look-ahead
# schedules and the occupancy of the batched panel kernels are exercised here at sizes the benchmark SDPs do not have.
RAGGED = dict(dims=[6, 4, 3, 1, 1, 3, 4], num_points=[34, 44, 40, 20, 28, 24, 30], N=60, seed=11)


@pytest.mark.parametrize("precision", [512, 768])
def test_big_and_ragged_blocks_match_the_live_oracle(precision):
    c = dict(RAGGED, precision=precision)
    sdp, s, o = _pair(c)
    assert max(sdp.num_points[j] * sdp.dims[j] * (sdp.dims[j] + 1) // 2 for j in range(sdp.J)) >= 420
    assert max(sdp.dims[j] * ((sdp.num_points[j] + 1) // 2) for j in range(sdp.J)) >= 100
    worst = float("-inf")
    for it in range(3):
        assert not s.iterate() and not o.iterate()
        bad, w = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=precision // 2)
        worst = max(worst, w)
        assert not bad, (it + 1, bad)
    report = _compare_arrays(sdp, s, o, precision, list(range(sdp.J)))
    bad = {k: v for k, v in report.items() if v > -(precision // 2)}
    print(f"ragged blocks at {precision} bits: scalars worst 2^{worst:.1f}; worst arrays:", sorted(report.items(), key=lambda kv: -kv[1])[:5])
    assert not bad, bad
    s.close()
    o.close()


def test_cholesky_Q_failure_is_reported_like_the_reference():
    """A free variable that appears in no constraint gives Q a zero row: both the oracle and the
    device path must stop with the reference's message (initialize_schur_complement_solver.cxx:95-103)."""
    from oracle.oracle import Oracle, OracleError
    from sdpb_amd.solver import SDPBError
    from sdpb_amd.synthetic import config, make_sdp
    c = config("C2")
    sdp = make_sdp(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"])
    for blk in sdp.blocks:
        for row in blk.B:
            row[3] = "0"
    o = Oracle(sdp, c["precision"], parity.DEFAULT_PARAMS, param_prec=0)
    with pytest.raises(OracleError) as eo:
        o.iterate()
    s = SDPSolver(sdp, c["precision"], parity.DEFAULT_PARAMS, lib_path=libs.product_lib())
    with pytest.raises(SDPBError) as es:
        s.iterate()
    assert es.value.code == 1
    assert "Error when computing Cholesky(Q)" in str(eo.value)
    assert "Error when computing Cholesky(Q)" in str(es.value)
    s.close()
    o.close()


def test_Y_cholesky_failure_names_block_and_parity():
    from sdpb_amd.solver import SDPBError
    sdp, meta, _, _ = parity.load_case("1d-constraints")
    s = SDPSolver(sdp, meta["precision"], lib_path=libs.product_lib())
    n = len(s.array("Y", 1, 1))
    s.set_array("Y", ["-1"] + ["0"] * (n - 1), 1, 1)
    with pytest.raises(SDPBError) as e:
        s.iterate()
    assert e.value.code == 1
    assert "Block_Diagonal_Matrix Y, block index = 1, parity = 1" in str(e.value)
    s.close()


def test_rccl_binding_inside_the_library():
    """The in-library exchange (csrc/rccl_comm.hpp) links and drives RCCL on the library's stream:
    one-rank communicator, all-gather + 64-bit SUM all-reduce through the solver's Comm class."""
    import ctypes
    from sdpb_amd.solver import load_library
    L = load_library(libs.product_lib())
    L.sdpb_hip_rccl_selftest.argtypes = [ctypes.c_size_t]
    rc = L.sdpb_hip_rccl_selftest(1 << 20)
    assert rc == 0, L.sdpb_hip_last_error(None).decode()


def test_three_host_synchronisation_points_per_iteration_on_the_device():
    c = _shape("C3", 0.1)
    sdp, s, _ = _pair(c, oracle=False)
    for _ in range(3):
        assert not s.iterate()
    assert s.host_syncs == 9
    s.close()


def test_concurrent_streams_give_the_same_bits_as_one_stream(monkeypatch):
    """Three streams (main; Y chain / Cholesky(Q) / dual step length; look-ahead bulk) versus everything
    on one stream (SDPB_HIP_SINGLE_STREAM=1): identical iterations to the last bit, so no result
    depends on an ordering the events do not enforce."""
    c = _shape("C4", 0.25)
    traces = []
    for single in ("0", "1"):
        monkeypatch.setenv("SDPB_HIP_SINGLE_STREAM", single)
        sdp, s, _ = _pair(c, oracle=False)
        t = []
        for _ in range(3):
            assert not s.iterate()
            t.append(s.scalars())
        t.append(s.array("dy")[:64])
        traces.append(t)
        s.close()
    assert traces[0] == traces[1]


def test_chased_cholesky_Q_gives_the_same_bits(monkeypatch):
    """Q' in two column chunks with Cholesky(Q) chasing it (SDPB_HIP_Q_CHASE=1, opt-in): the left
    panels are factored while the right chunk is multiplied, then applied to it.  Every entry of Q sees the same
    operations in the same order as in the one-piece schedule, so whole iterations agree to the last bit (C4 x0.25:
    N = 250, eight panels, split after the fourth; the single-stream run covers the ordering the events enforce)."""
    c = _shape("C4", 0.25)
    traces = []
    for chase, single in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("SDPB_HIP_Q_CHASE", chase)
        monkeypatch.setenv("SDPB_HIP_SINGLE_STREAM", single)
        sdp, s, _ = _pair(c, oracle=False)
        t = []
        for _ in range(3):
            assert not s.iterate()
            t.append(s.scalars())
        t.append(s.array("dy")[:64])
        traces.append(t)
        assert s.timers()["comm.q_chase"] == int(chase)
        s.close()
    assert traces[0] == traces[1] == traces[2]



def test_max_shared_memory_bounds_the_syrk_and_keeps_every_bit(monkeypatch):
    """sdpb_hip_set_max_shared_memory (--maxSharedMemory: run.cxx:79-181, BigInt_Shared_Memory_Syrk_Context.cxx:70-215) and
    sdpb_hip_memory_plan on C4 x0.25 (N = 250: 36 output tiles of 32 x 32, 10 000 rows in row splits).  The bound covers
    BOTH windows of the Q stage like the reference's: with a fifth of (image + partial planes) the fixed-point image of P' is
    built in >= 5 row windows through one bounded buffer (input_window_split_factor), the output tiles are walked in chunks
    (output windows), image + planes stay inside the bound, and whole iterations agree with the unbounded schedule to the
    last bit -- also with the chased Cholesky(Q) (whose image cannot be split: reported) under a bound on the planes alone."""
    c = _shape("C4", 0.25)
    traces, plans = [], []
    for bound_frac, chase in ((0, "0"), (5, "0"), (5, "1")):
        monkeypatch.setenv("SDPB_HIP_Q_CHASE", chase)
        if bound_frac and chase == "1":
            monkeypatch.setenv("SDPB_HIP_SYRK_PART_BYTES", str(plans[0]["syrk"]["partial_bytes"] // bound_frac))
        sdp, s, _ = _pair(c, oracle=False)
        plan = s.memory_plan()
        if bound_frac and chase == "0":
            bound = (plans[0]["syrk"]["partial_bytes"] + plans[0]["image"]["image_bytes"]) // bound_frac
            s.set_max_shared_memory(bound)
            plan = s.memory_plan()
            assert plan["syrk"]["budget_source"] == "maxSharedMemory" and plan["image"]["budget_source"] == "maxSharedMemory/2"
            assert plan["image"]["image_chunks"] >= 5 and plan["image"]["rows_per_chunk"] % 32 == 0, plan["image"]
            assert plan["image"]["image_bytes"] + plan["syrk"]["partial_bytes"] <= bound, (plan["image"], plan["syrk"])
            assert plan["bytes"]["syrk_partial_planes"] + plan["bytes"]["P_fixed_point_image"] <= bound, plan["bytes"]
            assert not plan["image"]["bound_exceeded_min_chunk"] and not plan["syrk"]["bound_exceeded_min_chunk"], plan
        elif bound_frac:
            bound = plans[0]["syrk"]["partial_bytes"] // bound_frac
            assert plan["syrk"]["budget_source"] == "SDPB_HIP_SYRK_PART_BYTES" and plan["image"]["image_chunks"] == 1
            assert plan["syrk"]["partial_bytes"] <= bound and plan["bytes"]["syrk_partial_planes"] <= bound, plan["syrk"]
            assert plan["syrk"]["chunks"] >= 3, plan["syrk"]
        else:
            assert plan["syrk"]["chunks"] == 1 and plan["syrk"]["tiles"] == 36 and plan["image"]["image_chunks"] == 1, plan
            assert plan["syrk"]["partial_bytes_unbounded"] < 0.6 * plan["syrk"]["partial_bytes_full_square_layout"]
            total = sum(plan["bytes"].values())
            assert 0 < total < plan["device"]["total_bytes"] and plan["bytes"]["B"] == plan["bytes"]["P"] > 0
        plans.append(plan)
        t = []
        for _ in range(3):
            assert not s.iterate()
            t.append(s.scalars())
        t.append(s.array("dy")[:64])
        traces.append(t)
        if bound_frac:
            last = s.memory_plan()
            assert last["last_syrk_call"]["chunks"] >= (2 if chase == "1" else 1)
            assert last["image"]["last_call_windows"] == plan["image"]["image_chunks"]
        s.close()
    assert traces[0] == traces[1] == traces[2]
