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


def make_sdp(dims: Sequence[int], num_points: Sequence[int], N: int, precision: int, seed: int = 1) -> SDP:
    digits = int(precision * 0.30103) + 10
    blocks = []
    cache = {}
    for j, (m, K) in enumerate(zip(dims, num_points)):
        P = K * m * (m + 1) // 2
        if K not in cache:
            cache[K] = (bases(K, 0, digits), bases(K, 1, digits))
        be, bo = cache[K]
        Bv = u(1000003 * seed + 7919 * j + 1, np.arange(P * N))
        cv = u(2000003 * seed + 104729 * j + 2, np.arange(P))
        Bs = _fmt(Bv)
        blocks.append(SDPBlock(dim=m, num_points=K, bases_even=be, bases_odd=bo,
                               B=[Bs[p * N:(p + 1) * N] for p in range(P)], c=_fmt(cv)))
    b = _fmt(u(3000017 * seed + 3, np.arange(N)))
    sdp = SDP(blocks=blocks, b=b, constant="0")
    assert sdp.P_total >= N, "need P_total >= N so that Q is positive definite"
    return sdp


def make_lazy(dims: Sequence[int], num_points: Sequence[int], N: int, precision: int, seed: int = 1):
    """Same SDP as make_sdp, but blocks are produced on demand as float64 arrays (the B and c
    entries are dyadic rationals, so float64 holds them exactly).  Returns (sdp, block_source)
    for SDPSolver(..., block_source=...)."""
    digits = int(precision * 0.30103) + 10
    cache = {}

    def source(j):
        m, K = dims[j], num_points[j]
        P = K * m * (m + 1) // 2
        if K not in cache:
            cache[K] = (bases(K, 0, digits), bases(K, 1, digits))
        be, bo = cache[K]
        Bv = u(1000003 * seed + 7919 * j + 1, np.arange(P * N)).reshape(P, N)
        cv = u(2000003 * seed + 104729 * j + 2, np.arange(P))
        return be, bo, Bv, cv

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
