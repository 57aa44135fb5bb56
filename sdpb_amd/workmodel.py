"""Analytic work / traffic model of one interior-point iteration.

MACs = multi-word multiply-adds; bytes = compulsory HBM traffic when every stage reads its
inputs once and writes its outputs once.  Used by bench.py for the roofline figures and to
scale the bounded CPU-baseline sample to the metric's unit.
"""
from __future__ import annotations

from typing import Dict, Sequence


def shapes(dims: Sequence[int], num_points: Sequence[int]):
    for m, K in zip(dims, num_points):
        P = K * m * (m + 1) // 2
        n0 = m * ((K + 1) // 2)
        n1 = m * K - n0
        yield m, K, P, (n0, n1), m * K


def macs_per_iteration(dims, num_points, N) -> Dict[str, float]:
    w = dict(cholesky_XY=0.0, pairings=0.0, schur=0.0, schur_cholesky=0.0, schur_trsm=0.0, syrk_Q=0.0,
             cholesky_Q=0.0, XY=0.0, search_direction=0.0, step_length=0.0, residues=0.0)
    for m, K, P, ns, q in shapes(dims, num_points):
        for n in ns:
            w["cholesky_XY"] += 2 * n ** 3 / 3.0            # a1
            w["pairings"] += 2.5 * n * n * q + n * q * q     # a2+a3 (dense E as in the reference)
            w["XY"] += n ** 3                                # a9
            w["search_direction"] += 2 * 7 * n ** 3          # a10, two calls
            w["step_length"] += 2 * (2 * n ** 3 + 6 * n ** 3)  # a12: congruence + eigenvalues
            w["residues"] += n * n * K
        w["schur"] += 8.0 * P * P                            # a4
        w["schur_cholesky"] += P ** 3 / 3.0                  # a5
        w["schur_trsm"] += P * P * N / 2.0                   # a5
        w["syrk_Q"] += P * N * (N + 1) / 2.0                 # a7
        w["search_direction"] += 2 * (2 * P * P + 2 * P * N)
        w["residues"] += 2 * P * N
    w["cholesky_Q"] = N ** 3 / 3.0                           # a8
    w["search_direction"] += 2 * 2 * N * N
    w["total"] = sum(w.values())
    return w


def algorithmic_bytes_per_iteration(dims, num_points, N, bytes_per_number: int) -> float:
    """SURVEY.md §8d formula with W = bytes_per_number."""
    t = 0.0
    for m, K, P, ns, q in shapes(dims, num_points):
        for n in ns:
            t += 2 * n * n + 2 * q * q
        t += 4 * q * q + P * P
        t += P * P + 2 * P * N
        t += P * N
    t += N * (N + 1) / 2 + N * N
    return t * bytes_per_number
