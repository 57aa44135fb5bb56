"""Deterministic synthetic SDPs in the SDP directory format (SURVEY.md §8d generator
contract): structurally faithful stand-ins for bootstrap SDPs that are not in the repo.

  u(seed, i)  = ((splitmix64(seed + i) >> 11) * 2^-52) - 1      in [-1, 1)
  x_k         = (k + 1/2)^2 pi^2 / (4 K)        increasing positive sample points
  s_k         = exp(-x_k)                       scalings
  bases_even[i][k] = sqrt(s_k) x_k^i / i!,  bases_odd[i][k] = sqrt(x_k s_k) x_k^i / i!
  B_j[p][n] = u(seed_B + j, p N + n),  c_j[p] = u(seed_c + j, p),  b[n] = u(seed_b, n)

Numbers are emitted as decimal strings, so every consumer (this library, the oracle,
the real sdpb) parses identical inputs.
"""
from __future__ import annotations

from typing import List, Sequence

from decimal import Decimal

import mpmath
import numpy as np

from .sdp_io import SDP, SDPBlock

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u(seed: int, idx: np.ndarray) -> np.ndarray:
    """Uniform dyadic rationals in [-1,1) as float64 (exactly representable)."""
    with np.errstate(over="ignore"):
        k = splitmix64(np.uint64(seed) + idx.astype(np.uint64)) >> np.uint64(11)
    return k.astype(np.float64) * 2.0 ** -52 - 1.0


def _fmt(a: np.ndarray) -> List[str]:
    # exact decimal expansion of the dyadic rational (<= 52 fractional digits): every
    # consumer parses exactly the same number
    return [format(Decimal(float(v)), "f") for v in a]


def bases(K: int, parity: int, digits: int) -> List[List[str]]:
    d = K - 1
    rows = (d + parity) // 2 + 1 - parity  # Block_Info.hxx:110-114
    old = mpmath.mp.dps
    mpmath.mp.dps = digits + 15
    try:
        out = []
        xs = [(mpmath.mpf(k) + 0.5) ** 2 * mpmath.pi ** 2 / (4 * K) for k in range(K)]
        pref = [mpmath.sqrt(mpmath.exp(-x) * (x if parity else 1)) for x in xs]
        for i in range(rows):
            out.append([mpmath.nstr(pref[k] * xs[k] ** i / mpmath.factorial(i), digits, strip_zeros=False,
                                    min_fixed=-10 ** 9, max_fixed=-10 ** 9 + 1) for k in range(K)])
        return out
    finally:
        mpmath.mp.dps = old


# ---- the strictly feasible family (feasible=True) ------------------------------------------------------
# The plain family above takes b and c at random, so its SDPs are infeasible and SDP_Solver::run ends them with
# "maxComplementarity exceeded".  The feasible family keeps bases and B and CONSTRUCTS b and c from an interior
# point of both problems (run.cxx:380-467 then ends with "found primal-dual optimal solution"):
#   primal  x0:  X(x0) = sum_p A_p x0_p = sum_k M_k (x) q_k q_k^T  with M_k the m x m matrix of the x0 entries of
#                sample point k (diagonal in [3/4, 5/4), off-diagonal |.| <= 1/64: diagonally dominant, and the
#                bases have full row rank), so X(x0) > 0;    b := B^T x0   (exact: dyadic rationals)
#   dual  (y0, Y0 = W (x) I > 0, W like M_k):   c_p := Tr(A_p Y0) + (B y0)_p
#                = W_cr (|q_even(x_k)|^2 + |q_odd(x_k)|^2) + (B y0)_p   for p = (c, r, k)
#                (the pairing term at the bases' precision, rounded to `digits` digits -- a 10^-digits relative
#                perturbation of an interior point -- the B y0 term exact).
_XS = 4000037


def _h(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) + idx.astype(np.uint64))


def _pair_index(m: int):
    """(c, r) of the pair index q = c (c + 1) / 2 + r, r <= c (compute_schur_complement.cxx:44-46)."""
    return [(c, r) for c in range(m) for r in range(c + 1)]


def _interior_entries(seed: int, m: int, K: int) -> np.ndarray:
    """Entries (pair q, point k) of an m x m family of diagonally dominant matrices, as integers / 2048."""
    P = K * m * (m + 1) // 2
    h = _h(seed, np.arange(P)).reshape(m * (m + 1) // 2, K)
    out = np.empty_like(h, dtype=np.int64)
    for q, (c, r) in enumerate(_pair_index(m)):
        if c == r:
            out[q] = 1536 + (h[q] & np.uint64(1023)).astype(np.int64)          # [3/4, 5/4)
        else:
            out[q] = (h[q] & np.uint64(63)).astype(np.int64) - 32               # |.| <= 1/64
    return out


def _exact_B_times(Bv: np.ndarray, w: np.ndarray, axis: int) -> list:
    """sum over `axis` of B * w with B = k / 2^52 (float64, exact) and integer weights |w| < 2^12: Python integers,
    scaled by 2^52.  The 53-bit numerators are split into 26-bit halves so that every int64 sum is exact."""
    k = np.round(Bv * 2.0 ** 52).astype(np.int64)
    lo = k & np.int64((1 << 26) - 1)
    hi = k >> np.int64(26)
    ww = w.astype(np.int64)
    if axis == 0:
        slo, shi = ww @ lo, ww @ hi                    # (N,)
    else:
        slo, shi = lo @ ww, hi @ ww                    # (P,)
    return [int(a) + (int(b) << 26) for a, b in zip(slo.tolist(), shi.tolist())]


def _dyadic(num: int, log2den: int) -> str:
    """Exact decimal expansion of num / 2^log2den."""
    import decimal
    with decimal.localcontext() as ctx:
        ctx.prec = len(str(abs(num))) + log2den + 5
        return format(decimal.Decimal(num) / (decimal.Decimal(2) ** log2den), "f")


def _pairing_norms(be, bo, K: int, digits: int):
    """|q_even(x_k)|^2 + |q_odd(x_k)|^2 per sample point from the bases AS EMITTED (the decimal strings)."""
    old = mpmath.mp.dps
    mpmath.mp.dps = digits + 15
    try:
        return [sum(mpmath.mpf(row[k]) ** 2 for row in be) + sum(mpmath.mpf(row[k]) ** 2 for row in bo) for k in range(K)]
    finally:
        mpmath.mp.dps = old


def _feasible_c(seed: int, j: int, m: int, K: int, Bv: np.ndarray, y0: np.ndarray, norms, digits: int) -> List[str]:
    W = _interior_entries(_XS * seed + 15485863 * j + 5, m, 1)[:, 0]             # one matrix per block, / 2048
    By = _exact_B_times(Bv, y0, axis=1)                                           # scaled 2^52 * 2^8
    old = mpmath.mp.dps
    mpmath.mp.dps = digits + 15
    try:
        out = []
        for q in range(m * (m + 1) // 2):
            for k in range(K):
                v = mpmath.mpf(int(W[q])) / 2048 * norms[k] + mpmath.mpf(By[q * K + k]) / mpmath.mpf(2) ** 60
                out.append(mpmath.nstr(v, digits, strip_zeros=False, min_fixed=-10 ** 9, max_fixed=-10 ** 9 + 1))
        return out
    finally:
        mpmath.mp.dps = old


def _y0(seed: int, N: int) -> np.ndarray:
    return (_h(_XS * seed + 7, np.arange(N)) & np.uint64(255)).astype(np.int64) - 128    # / 256


def make_sdp(dims: Sequence[int], num_points: Sequence[int], N: int, precision: int, seed: int = 1, feasible: bool = False) -> SDP:
    digits = int(precision * 0.30103) + 10
    blocks = []
    cache = {}
    b_acc = [0] * N
    y0 = _y0(seed, N)
    for j, (m, K) in enumerate(zip(dims, num_points)):
        P = K * m * (m + 1) // 2
        if K not in cache:
            be, bo = bases(K, 0, digits), bases(K, 1, digits)
            cache[K] = (be, bo, _pairing_norms(be, bo, K, digits) if feasible else None)
        be, bo, norms = cache[K]
        Bv = u(1000003 * seed + 7919 * j + 1, np.arange(P * N))
        Bs = _fmt(Bv)
        if feasible:
            Bm = Bv.reshape(P, N)
            x0 = _interior_entries(_XS * seed + 32452843 * j + 3, m, K).reshape(P)   # / 2048
            for n, v in enumerate(_exact_B_times(Bm, x0, axis=0)):
                b_acc[n] += v
            c = _feasible_c(seed, j, m, K, Bm, y0, norms, digits)
        else:
            c = _fmt(u(2000003 * seed + 104729 * j + 2, np.arange(P)))
        blocks.append(SDPBlock(dim=m, num_points=K, bases_even=be, bases_odd=bo,
                               B=[Bs[p * N:(p + 1) * N] for p in range(P)], c=c))
    b = [_dyadic(v, 63) for v in b_acc] if feasible else _fmt(u(3000017 * seed + 3, np.arange(N)))
    sdp = SDP(blocks=blocks, b=b, constant="0")
    assert sdp.P_total >= N, "need P_total >= N so that Q is positive definite"
    return sdp


def make_lazy(dims: Sequence[int], num_points: Sequence[int], N: int, precision: int, seed: int = 1, feasible: bool = False):
    """Same SDP as make_sdp, but blocks are produced on demand as float64 arrays (the B entries and, in the plain
    family, the c entries are dyadic rationals, so float64 holds them exactly; in the feasible family c comes as the
    same decimal strings make_sdp emits).  Returns (sdp, block_source) for SDPSolver(..., block_source=...)."""
    digits = int(precision * 0.30103) + 10
    cache = {}
    y0 = _y0(seed, N)

    def _bases(K):
        if K not in cache:
            be, bo = bases(K, 0, digits), bases(K, 1, digits)
            cache[K] = (be, bo, _pairing_norms(be, bo, K, digits) if feasible else None)
        return cache[K]

    def _B(j, P):
        return u(1000003 * seed + 7919 * j + 1, np.arange(P * N)).reshape(P, N)

    def source(j):
        m, K = dims[j], num_points[j]
        P = K * m * (m + 1) // 2
        be, bo, norms = _bases(K)
        Bv = _B(j, P)
        if feasible:
            return be, bo, Bv, _feasible_c(seed, j, m, K, Bv, y0, norms, digits)
        cv = u(2000003 * seed + 104729 * j + 2, np.arange(P))
        return be, bo, Bv, cv

    if feasible:
        b_acc = [0] * N
        for j, (m, K) in enumerate(zip(dims, num_points)):
            P = K * m * (m + 1) // 2
            x0 = _interior_entries(_XS * seed + 32452843 * j + 3, m, K).reshape(P)
            for n, v in enumerate(_exact_B_times(_B(j, P), x0, axis=0)):
                b_acc[n] += v
        b = [_dyadic(v, 63) for v in b_acc]
    else:
        b = _fmt(u(3000017 * seed + 3, np.arange(N)))
    sdp = SDP(blocks=[], b=b, constant="0", shape=(list(dims), list(num_points)))
    assert sdp.P_total >= N
    return sdp, source


# BASELINE.json configs; C1 is the shipped fixture, not synthetic.
CONFIGS = {
    "C2": dict(dims=[1] * 8, num_points=[20] * 8, N=9, precision=256, seed=2),
    "C3": dict(dims=[1] * 600, num_points=[30] * 600, N=100, precision=512, seed=3),
    "C4": dict(dims=[2] * 200 + [1] * 400, num_points=[40] * 600, N=1000, precision=512, seed=4),
    "C5": dict(dims=[6] * 8192, num_points=[2] * 8192, N=2048, precision=1024, seed=5),
    # one GPU's share of C5 under 8-way block sharding, with the full N: what a rank of the 8-GPU job holds
    "C5slice": dict(dims=[6] * 1024, num_points=[2] * 1024, N=2048, precision=1024, seed=5),
    # C4's shape with b and c constructed from an interior point of both problems (feasible family above): the regime a
    # user runs -- mu -> 0, condition numbers climbing -- ending in "found primal-dual optimal solution"
    "C4f": dict(dims=[2] * 200 + [1] * 400, num_points=[40] * 600, N=1000, precision=512, seed=4, feasible=True),
}


def config(name: str, scale: float = 1.0) -> dict:
    """Shape of a BASELINE.json config; scale < 1 shrinks J and N proportionally (tests)."""
    c = dict(CONFIGS[name])
    if scale != 1.0:
        J = max(2, int(len(c["dims"]) * scale))
        # keep the mix of block kinds
        idx = np.linspace(0, len(c["dims"]) - 1, J).astype(int)
        c["dims"] = [c["dims"][i] for i in idx]
        c["num_points"] = [c["num_points"][i] for i in idx]
        c["N"] = max(2, int(c["N"] * scale))
    return c


def lazy(c: dict):
    """make_lazy for a config() dictionary."""
    return make_lazy(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"], feasible=c.get("feasible", False))
