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


# ---- planned HBM footprint of a rank (round 5) ------------------------------------------------------------------
# The arrays Solver::build_layout (csrc/solver.hpp) allocates, from the shapes alone, so that a launcher can refuse a
# plan that cannot fit BEFORE anything is uploaded (the reference estimates its memory per node the same way before it
# allocates: run.cxx:79-181).  tests/test_gpu_parity_at_size.py compares it with sdpb_hip_memory_plan on the device.
COMPILED_LIMBS = (6, 10, 16, 18, 24, 26, 34, 42, 50, 66)
HBM_BYTES = 288 * 10 ** 9


def limbs_for(precision: int) -> int:
    want = 2 * ((precision + 127) // 64)
    for nl in COMPILED_LIMBS:
        if nl >= want:
            return nl
    raise ValueError(f"--precision {precision} is beyond the compiled widths")


def _fx(nl: int):
    """(FX, image pieces in words per element, tile edge, planes one row split writes, staged rows, waves per SIMD)"""
    fx = nl - 2
    if fx >= 14 and fx % 4:
        fx += 4 - fx % 4
    toom4 = fx >= 16 and fx % 4 == 0
    toom4k = fx in (16, 24, 32)
    toom5k = fx == 16          # kernels.hpp: fx_toom5k (Toom-5 x Karatsuba on 28-bit limbs: 27 pieces of two words, 27 x 5 limbs per split)
    two = fx % 4 == 0
    words = 54 if toom5k else 21 * (fx // 8) if toom4k else 7 * (fx // 4) if toom4 else 9 * (fx // 4) if two else 3 * (fx // 2)
    planes = 135 if toom5k else 21 * (2 * (fx // 8) + 1) if toom4k else 7 * (2 * (fx // 4) + 1) if toom4 else 2 * fx + 2
    rb = (16 if fx >= 32 else 32) if two else (16 if fx <= 24 else 8)
    waves = 3 if toom4k else (3 if fx <= 16 else 2)
    return fx, words, 32 if toom4k else 16, planes, rb, waves, toom4, toom4k


def syrk_row_splits(ntile: int, nrows: int, slots: int, rb: int, max_rows: int) -> int:
    """kernels.hpp: syrk_row_splits"""
    import math
    smin = 1
    while max_rows and smin < 32 and nrows // smin > max_rows and nrows // (smin + 1) >= 64 * rb:
        smin += 1
    best, best_eff = smin, 0.0
    for s in range(smin, 33):
        if s > smin and nrows // s < 64 * rb:
            break
        items = ntile * s
        eff = items / (math.ceil(items / slots) * slots)
        if eff > best_eff + 0.02:
            best, best_eff = s, eff
    return best


def q_window(rows: int, N: int, nl: int, image_budget_bytes: int):
    """solver.hpp: Solver::q_window -- (rows per input window, windows, bytes of the window's image)."""
    fx, words, edge, planes, rb, waves, toom4, toom4k = _fx(nl)

    def image_bytes(r):
        stride = (-(-r // rb) * rb * N + 64) if toom4k else r * N
        return (max(stride, 1) * words + 4) * 4
    r = max(rows, 1)
    budget_words = max(1, image_budget_bytes // 4)
    if image_bytes(r) // 4 > budget_words:
        per_row = N * words
        fixed = 64 * words + 4
        fit = ((budget_words - fixed) // per_row if budget_words > fixed else 0) // rb * rb
        fit = max(fit, rb)
        f = -(-rows // fit)
        cr = -(-(-(-rows // f)) // rb) * rb
        if toom4k and cr > 2560:
            up = -(-cr // 2560) * 2560
            if up <= fit:
                cr = up
        r = cr
    return r, (-(-rows // r) if rows else 1), image_bytes(r)


def planned_footprint(dims, num_points, N, precision, owners=None, rank=0, world=1, num_cus=256, hbm_bytes=HBM_BYTES,
                      max_shared_bytes=0, dist_cholq=False, free_bytes=None) -> Dict[str, float]:
    """Bytes per array class this rank will allocate (same classes as sdpb_hip_memory_plan) and their sum.
    hbm_bytes / free_bytes / num_cus: pass the device's own figures (torch.cuda.mem_get_info, the CU count) so that the
    plan follows the library's rule for the two windows of the Q stage -- min(free - reserve, total / 8) with reserve =
    total / 16 + 1 GiB, taken after everything else is allocated (solver.hpp: build_layout); the data-sheet values are
    only the default."""
    nl = limbs_for(precision)
    W = 4 * (nl + 1)
    fx, words, edge, planes, rb, waves, toom4, toom4k = _fx(nl)
    psd = bases = scaled = E = pair = schur = bt = vecn = rows = jl = 0
    for j, (m, K, P, ns, q) in enumerate(shapes(dims, num_points)):
        if owners is not None and owners[j] != rank:
            continue
        jl += 1
        d = K - 1
        pr = m * (m + 1) // 2
        for n, r in zip(ns, (d // 2 + 1, (d + 1) // 2)):
            psd += n * n
            vecn += n
            bases += r * K
            scaled += K * pr * r
            E += n * q
            pair += q * q
        schur += P * P
        bt += N * P
        rows += P
    out = {"psd_state_and_scratch": 13 * max(psd, 1) * W, "bases_and_pairings": (2 * max(bases, 1) + max(scaled, 1) + 4 * max(E, 1) + 2 * max(pair, 1)) * W,
           "schur_blocks": 2 * max(schur, 1) * W, "B": max(bt, 1) * W, "P": max(bt, 1) * W}
    accw = 2 * fx + 2
    q_bytes = 2 * N * N * W + (N * N + N) * accw * 4
    if world > 1:
        q_bytes += (N * (N + 1) // 2 + N) * accw * 8
        if dist_cholq:
            pb = 16 if nl > 34 else 32
            q_bytes += ((N * pb + pb * pb + pb) * (nl + 1) + 2) * 4
    out["Q"] = q_bytes
    out["vectors_and_small"] = (5 * max(rows, 1) + 6 * max(vecn, 1) + 8 * N + 2 * max(jl, 1) * N) * W
    # the two windows of the Q stage, planned together against what the rest leaves (solver.hpp: build_layout, q_window, syrk_plan)
    rest = sum(out.values())
    free = (hbm_bytes if free_bytes is None else free_bytes) - rest
    reserve = hbm_bytes // 16 + (1 << 30)
    window = max_shared_bytes or max(min(max(free - reserve, 0), hbm_bytes // 8), 64 << 20)
    chunk_rows, windows, image = q_window(rows, N, nl, max(window // 2, 4))
    out["P_fixed_point_image"] = image
    if windows > 1:
        out["Q"] += (N * N + N) * accw * 4   # the accumulator of the windows after the first
    tiles = -(-N // edge)
    ntile = tiles * (tiles + 1) // 2

    def planes_unbounded(r):
        ns = syrk_row_splits(ntile * (21 if toom4k else 1), r, num_cus * waves, rb, 2560 if toom4k else 0) if r else 1
        return ns * planes * ntile * edge * edge * 4 if (ns > 1 or toom4) else 0
    unbounded = planes_unbounded(chunk_rows)
    budget = max(window - min(image, window // 2), 4)
    out["syrk_partial_planes"] = min(unbounded, max(budget, planes * edge * edge * 4))
    out["total"] = float(sum(out.values()))
    out["rows"] = rows
    out["owned_blocks"] = jl
    out["syrk_partial_planes_unbounded"] = planes_unbounded(rows)
    out["image_chunks"] = windows
    out["image_rows_per_chunk"] = chunk_rows
    out["window_budget_bytes"] = window
    return out
