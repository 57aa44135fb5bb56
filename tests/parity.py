"""Shared comparison helpers for the parity tests.

Tolerances are the reference's own:
  * relative 2^-99 on every numeric field of every iterations.json record and on
    out.txt's primalObjective/dualObjective (end-to-end.test.cxx:27, diff.hxx:50-76:
    |a-b| < 2^-99 (|a|+|b|));
  * error-like fields are skipped when below 2^-49 (diff_sdpb_out.cxx:270-281).
"""
import json
import os
import re

import mpmath

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NUMERIC_KEYS = ["mu", "P-obj", "D-obj", "gap", "P-err", "p-err", "D-err", "R-err",
                "P-step", "D-step", "beta", "Q_cond_number", "max_block_cond_number"]
ERROR_KEYS = {"P-err", "p-err", "D-err", "R-err"}
mpmath.mp.prec = 1400


def cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


def load_case(name):
    from sdpb_amd.sdp_io import read_sdp
    meta = cases()[name]
    d = os.path.join(GOLDEN, name)
    sdp = read_sdp(os.path.join(d, "sdp"))
    with open(os.path.join(d, "iterations.json")) as f:
        iters = json.load(f)
    with open(os.path.join(d, "out.txt")) as f:
        out_txt = f.read()
    out = {}
    for m in re.finditer(r"(\w[\w ]*?)\s*=\s*([^;]+);", out_txt):
        out[m.group(1).strip()] = m.group(2).strip().strip('"')
    return sdp, meta, iters, out


def rel_diff(a, b):
    a = mpmath.mpf(a)
    b = mpmath.mpf(b)
    if a == b:
        return mpmath.mpf(0)
    return abs(a - b) / (abs(a) + abs(b))


def log2_rel(a, b):
    r = rel_diff(a, b)
    return float(mpmath.log(r, 2)) if r > 0 else float("-inf")


def compare_iteration(got: dict, want: dict, tol_bits: int = 99, skip_err_below_bits: int = 49):
    """Return list of (key, log2 relative diff) that violate 2^-tol_bits."""
    bad = []
    worst = float("-inf")
    for k in NUMERIC_KEYS:
        if k not in want:
            continue
        w = mpmath.mpf(want[k])
        g = mpmath.mpf(got[k])
        if k in ERROR_KEYS and abs(w) < mpmath.mpf(2) ** -skip_err_below_bits \
                and abs(g) < mpmath.mpf(2) ** -skip_err_below_bits:
            continue
        l2 = log2_rel(g, w)
        worst = max(worst, l2)
        if l2 > -tol_bits:
            bad.append((k, l2))
    return bad, worst


def check_int_syrk_extremes(s, rows, cols):
    """The exact integer syrk on the inputs that load its carry-free column sums the most: EVERY entry at the largest
    magnitude the image holds (all limbs of all pieces at their maximum), one full sweep of `rows` rows (the caller forces
    one row split), columns with constant sign +, -, and rows of alternating sign.  Random inputs sit at a quarter of the bound
    the kernel's carry schedule is derived from (k_syrk_fx3, LAZY: a 64-bit column absorbs 64 rows of products < 2^56), so
    only these prove the schedule.  Expected values are plain Python integers."""
    fb = s.fx_frac_bits
    big = 2 ** fb - 1
    patterns = [lambda r, c: big,
                lambda r, c: big if c % 2 == 0 else -big,
                lambda r, c: big if (r + c) % 2 == 0 else -big,
                lambda r, c: -big if c % 3 == 0 else big - (c + 1) * (2 ** (fb // 2))]
    for k, f in enumerate(patterns):
        cols_v = [[f(r, c) for r in range(rows)] for c in range(cols)]
        vals = [cols_v[c][r] for c in range(cols) for r in range(rows)]   # column-major rows x cols
        got = s.op_int_syrk(rows, cols, vals)
        for j in range(cols):
            for i in range(j, cols):
                want = sum(a * b for a, b in zip(cols_v[i], cols_v[j]))
                assert got[i + j * cols] == want, (k, i, j)


def conditioned_tol_bits(rec: dict, precision: int, tol_bits: int, guard_bits: int = 16) -> int:
    """The bar an iteration can be held to when its linear algebra is ill-conditioned: two p-bit computations of the same
    iteration that round in different orders (GMP's mpf products truncate, the device rounds to nearest; sums are taken
    in different orders) agree to about cond 2^-p, where cond is the larger of the two condition numbers the iteration
    itself reports (max_block_cond_number: the Cholesky factors of the PSD blocks, Q_cond_number: the Schur complement).
    The SURVEY 8d bar 2^-(p/2) stays wherever cond < 2^(p/2 - guard); past that (the last iterations of a run that
    converges to 'found primal-dual optimal solution': block condition 2^296 at 512 bits in the strictly feasible
    fixture C4f, measured difference cond 2^-(p+38)) the bar is 2^-(p - log2 cond - guard)."""
    cond = max(mpmath.mpf(rec.get("max_block_cond_number", 1)), mpmath.mpf(rec.get("Q_cond_number", 1)), mpmath.mpf(1))
    return min(tol_bits, precision - int(mpmath.ceil(mpmath.log(cond, 2))) - guard_bits)


DEFAULT_PARAMS = {  # Solver_Parameters.cxx:10-157
    "dualityGapThreshold": "1e-30", "primalErrorThreshold": "1e-30", "dualErrorThreshold": "1e-30",
    "initialMatrixScalePrimal": "1e20", "initialMatrixScaleDual": "1e20",
    "feasibleCenteringParameter": "0.1", "infeasibleCenteringParameter": "0.3",
    "stepLengthReduction": "0.7", "maxComplementarity": "1e100", "minPrimalStep": "0", "minDualStep": "0"}
FLAG_KEYS = {"maxIterations", "findPrimalFeasible", "findDualFeasible", "detectPrimalFeasibleJump",
             "detectDualFeasibleJump"}


def reference_params(params: dict, oracle) -> dict:
    """The parameter VALUES the reference actually ran with: sdpb parses its options before
    --precision is applied, i.e. at GMP's initial 64-bit precision (visible in the golden
    traces: beta = 0.2999...98725e-58 in 1d/iterations.json).  Returns exact decimals."""
    full = dict(DEFAULT_PARAMS)
    full.update({k: v for k, v in params.items() if k not in FLAG_KEYS})
    out = {k: oracle.parse_exact(v, 64) for k, v in full.items()}
    out.update({k: v for k, v in params.items() if k in FLAG_KEYS})
    return out


def check_syrk_Q(solver, precision, rows=23, cols=9, seed=3):
    """The reference's own unit test of syrk_Q (calculate_matrix_square.test.cxx: random P in
    (-1, 1), Q = P^T P against the plain product, tolerance p/2 bits `:213`), plus the columns that
    test never produces: a single non-zero NEGATIVE entry, whose normalised value is -1 (or rounds
    to it) and saturates the fixed-point image."""
    import random
    from fractions import Fraction
    rng = random.Random(seed)
    P = [[Fraction(rng.randrange(-2 ** 52, 2 ** 52), 2 ** 52) for _ in range(cols)] for _ in range(rows)]
    # columns 0..3: one non-zero entry each; 0 and 1 share a row (their Q entry is a product of two
    # saturated values), 3 is positive, the scales differ wildly
    singles = {0: (4, Fraction(-3)), 1: (4, Fraction(-7, 1024)), 2: (9, Fraction(-1)), 3: (11, Fraction(5, 2 ** 40))}
    for c, (r, v) in singles.items():
        for rr in range(rows):
            P[rr][c] = Fraction(0)
        P[r][c] = v

    def dec(fr):   # every entry is a dyadic rational: its decimal expansion is finite and exact
        import decimal
        with decimal.localcontext() as ctx:
            ctx.prec = 400
            return format(decimal.Decimal(fr.numerator) / decimal.Decimal(fr.denominator), "f")
    colmajor = [dec(P[r][c]) for c in range(cols) for r in range(rows)]
    got = solver.op_syrk_Q(rows, cols, colmajor)
    assert len(got) == cols * cols
    worst = float("-inf")
    for j in range(cols):
        for i in range(cols):
            g = mpmath.mpf(got[i + j * cols])
            if i < j:
                assert g == 0
                continue
            want = sum(P[r][i] * P[r][j] for r in range(rows))
            w = mpmath.mpf(want.numerator) / want.denominator
            if w == 0:
                # orthogonal columns: exact zero up to the truncation of the image (2^-FB per term)
                ni = mpmath.sqrt(sum(float(P[r][i]) ** 2 for r in range(rows)))
                nj = mpmath.sqrt(sum(float(P[r][j]) ** 2 for r in range(rows)))
                assert abs(g) <= ni * nj * mpmath.mpf(2) ** (-precision + 16), (i, j, got[i + j * cols])
                continue
            l2 = log2_rel(g, w)
            # a random inner product of 23 terms may cancel a few bits; p/2 is the reference's bar
            assert l2 <= -(precision // 2), (i, j, l2)
            worst = max(worst, l2)
    return worst


def check_c1_at_precision_128(lib_path, n_iter=10):
    """BASELINE.json config C1 at its STATED precision: the shipped single-correlator SDP
    (SingletScalar_cT_test_nmax6, J=11, N=20) at --precision 128, first n_iter iterations against
    the mpf oracle at p=128 (at 128 bits this SDP is under-resolved, so the check is
    "the first k iterations equal the oracle", not convergence).  Tolerance 2^-(p/2) = 2^-64 on every
    field but p-err: p = b - B^T x cancels ~85 bits at this precision (terms ~1e26, result ~10), so
    BOTH implementations carry p to ~2^-15 only; there the bar is the same 2^-64 applied to the
    cancelling terms (forward error <= conditioning x backward error): |dp| <= 2^-64 max_n sum_p |B_pn x_p|."""
    from oracle.oracle import Oracle
    from sdpb_amd.solver import SDPSolver
    sdp, meta, _, _ = load_case("singlet_cT")
    o = Oracle(sdp, 128, meta["params"], param_prec=64)
    s = SDPSolver(sdp, 128, reference_params(meta["params"], o), lib_path=lib_path)
    assert s.limbs == 6
    worst = float("-inf")
    for it in range(n_iter):
        assert not s.iterate() and not o.iterate(), (it + 1, s.terminate_reason)
        got, want = s.scalars(), o.scalars()
        amp = mpmath.mpf(0)
        xs = [[mpmath.mpf(v) for v in s.array("x", j)] for j in range(sdp.J)]
        for n in range(sdp.N):
            t = mpmath.mpf(0)
            for j, blk in enumerate(sdp.blocks):
                for pp, row in enumerate(blk.B):
                    t += abs(mpmath.mpf(row[n]) * xs[j][pp])
            amp = max(amp, t)
        dp = abs(mpmath.mpf(got["p-err"]) - mpmath.mpf(want["p-err"]))
        assert dp <= amp * mpmath.mpf(2) ** -64, (it + 1, float(mpmath.log(dp / amp, 2)))
        got = dict(got)
        got["p-err"] = want["p-err"]
        bad, w = compare_iteration(got, want, tol_bits=64)
        assert not bad, (it + 1, bad)
        worst = max(worst, w)
    s.close()
    o.close()
    return worst


def maxrel(got, want):
    """log2 of max |got - want| / max |want| over two lists of decimals"""
    g = [mpmath.mpf(v) for v in got]
    w = [mpmath.mpf(v) for v in want]
    assert len(g) == len(w)
    scale = max(abs(v) for v in w)
    if scale == 0:
        return float("-inf") if all(v == 0 for v in g) else 0.0
    d = max(abs(a - b) for a, b in zip(g, w)) / scale
    return float(mpmath.log(d, 2)) if d > 0 else float("-inf")


def check_schur_hook_against_oracle(lib_path, name="singlet_cT", warm_iterations=3):
    """SURVEY.md §8f row 4 against the ORACLE (not against the iteration's own solver): from a loaded
    solution (x, X, y, Y) — what approx_objective/setup_solver.cxx:204-220 and outer_limits/
    compute_optimal.cxx:188-215 start from — sdpb_hip_schur_solver_init must leave the same
    schur_complement_cholesky L_j, schur_off_diagonal P_j and Cholesky(Q) as the reference sequence
    restated in oracle/sdpb_oracle.cpp (orc_schur_solver_init), and sdpb_hip_schur_solve the same
    solution of solve_schur_complement_equation.cxx:16-79 for arbitrary right-hand sides.
    Tolerance 2^-(p/2) relative to the largest entry of each array."""
    import random
    from oracle.oracle import Oracle
    from sdpb_amd.solver import SDPSolver
    sdp, meta, _, _ = load_case(name)
    p = meta["precision"]
    o = Oracle(sdp, p, meta["params"], param_prec=64)
    for _ in range(warm_iterations):
        assert not o.iterate()
    s = SDPSolver(sdp, p, reference_params(meta["params"], o), lib_path=lib_path)
    limbs64 = s.limbs // 2 + 1
    for j in range(sdp.J):                      # the loaded solution: bit-exact records of the oracle's state
        s.set_array_mpf("x", o.records("x", j, 0, limbs64), j)
        for b in (0, 1):
            s.set_array_mpf("X", o.records("X", j, b, limbs64), j, b)
            s.set_array_mpf("Y", o.records("Y", j, b, limbs64), j, b)
    s.set_array_mpf("y", o.records("y", 0, 0, limbs64))
    s.schur_solver_init()
    o.schur_solver_init()
    N, tol, report = sdp.N, -(p // 2), {}

    def lower(v, n):
        return [v[i + j * n] for j in range(n) for i in range(j, n)]
    for j in range(sdp.J):
        P = sdp.num_points[j] * sdp.dims[j] * (sdp.dims[j] + 1) // 2
        report[("L", j)] = maxrel(lower(s.array("L", j), P), lower(o.array("L", j), P))
        pt, po = s.array("PT", j), o.array("P", j)   # N x P resp. P x N, column-major
        report[("P", j)] = maxrel([pt[n + q * N] for n in range(N) for q in range(P)], po)
    q_dev, q_orc = s.array("Q"), o.array("Q")        # lower factor resp. El::Cholesky(UPPER)
    report[("chol(Q)",)] = maxrel(lower(q_dev, N), [q_orc[j + i * N] for j in range(N) for i in range(j, N)])
    rng = random.Random(11)
    for trial in range(2):
        for j in range(sdp.J):
            P = sdp.num_points[j] * sdp.dims[j] * (sdp.dims[j] + 1) // 2
            rhs = [repr(rng.uniform(-1, 1) * 10.0 ** rng.randint(-3, 3)) for _ in range(P)]
            s.set_array("dx", rhs, j)
            o.set_array("dx", rhs, j)
        rhs = [repr(rng.uniform(-1, 1)) for _ in range(N)]
        s.set_array("dy", rhs)
        o.set_array("dy", rhs)
        s.schur_solve()
        o.schur_solve()
        report[("dy", trial)] = maxrel(s.array("dy"), o.array("dy"))
        for j in range(sdp.J):
            report[("dx", trial, j)] = maxrel(s.array("dx", j), o.array("dx", j))
    bad = {k: v for k, v in report.items() if v > tol}
    assert not bad, bad
    s.close()
    o.close()
    return max(report.values())


def check_min_eigenvalue(solver, oracle, limbs, n=40, seed=5):
    """min_eigenvalue.cxx:8-33 as an operator on spectra that whole iterations reach only late in a convergent run: every
    eigenvalue of L^-1 dX L^-T within 2^-k of -(1 - beta) (then both step lengths are gamma / (1 - beta), step.cxx:202-206).
    A = c I + 2^-k S with S random symmetric: Newton on det(T - x) sees one root of multiplicity n unless the matrix is
    shifted first (kernels.hpp: k_tridiag_min) -- the unshifted iteration kept the 2^-48 of its fp64 start (found on the
    strictly feasible C4f x0.25 fixture, iteration 81).  Also: a generic matrix, c I itself, a half-clustered spectrum, n = 1, 2.
    Bar: 2^-(32 limbs - 64) of the largest entry of A."""
    import random
    from fractions import Fraction
    rng = random.Random(seed)

    def dec(fr):
        import decimal
        with decimal.localcontext() as ctx:
            ctx.prec = 900
            return format(decimal.Decimal(fr.numerator) / decimal.Decimal(fr.denominator), "f")

    def sym(m, scale):
        S = [[Fraction(0)] * m for _ in range(m)]
        for i in range(m):
            for j in range(i + 1):
                S[i][j] = S[j][i] = Fraction(rng.randrange(-2 ** 40, 2 ** 40), 2 ** 40) * scale
        return S
    c = Fraction(-9, 10)
    cases = []
    for k in (0, 12, 40, 90, 200, 400):
        S = sym(n, Fraction(1, 2 ** k))
        cases.append((f"c I + 2^-{k} S", n, [[S[i][j] + (c if i == j else 0) for j in range(n)] for i in range(n)]))
    cases.append(("c I", n, [[c if i == j else Fraction(0) for j in range(n)] for i in range(n)]))
    S, R = sym(n // 2, Fraction(1, 2 ** 70)), sym(n, Fraction(1, 2 ** 70))
    half = [[R[i][j] for j in range(n)] for i in range(n)]
    for i in range(n):
        half[i][i] += c if i < n // 2 else Fraction(rng.randrange(1, 2 ** 20), 2 ** 18)
    cases.append(("half of the spectrum clustered", n, half))
    cases.append(("n = 1", 1, [[Fraction(-7, 3)]]))
    cases.append(("n = 2", 2, [[Fraction(1, 3), Fraction(5, 7)], [Fraction(5, 7), Fraction(-2, 9)]]))
    worst = float("-inf")
    for name, m, A in cases:
        col = [dec(A[i][j]) for j in range(m) for i in range(m)]
        got, want = mpmath.mpf(solver.op_min_eigenvalue(m, col)), mpmath.mpf(oracle.min_eigenvalue(m, col))
        scale = max(abs(mpmath.mpf(A[i][j].numerator) / A[i][j].denominator) for i in range(m) for j in range(m))
        d = abs(got - want) / scale
        l2 = float(mpmath.log(d, 2)) if d > 0 else float("-inf")
        assert l2 <= -(32 * limbs - 64), (name, l2)
        worst = max(worst, l2)
    return worst
