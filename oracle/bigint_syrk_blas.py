"""The reference's OWN algorithm for the dominant stage, Q' = P'^T P' over the integers — TEST
INFRASTRUCTURE / CPU-baseline leg only (bench.py's cpu_baseline, tests/); nothing under sdpb_amd/ imports it.

The reference does not multiply big integers directly: BigInt_Shared_Memory_Syrk_Context::bigint_syrk_blas
(src/sdp_solve/SDP_Solver/run/bigint_syrk/BigInt_Shared_Memory_Syrk_Context/bigint_syrk_blas.cxx:183-302)
  1. picks word-size primes p_1 > p_2 > ... whose product exceeds the largest possible entry of Q'
     (fmpz/Fmpz_Comb.cxx:14-73: start below 2 + 2 floor(sqrt((2^53 - 1) / k)), k = rows of P', capped at 1664544
     when more than 200 bits are needed; primes are taken downwards until prod > 2^bits,
     bits = Abits + Bbits + bit_count(k) + sign),
  2. reduces every entry of P' modulo every prime into fp64 matrices (compute_block_residues.cxx:223-330,
     _fmpz_multi_mod_precomp / fmpz_get_nmod, fmpz_mul_blas_util.hxx:71,84; residues are centred, so a column of
     k products stays below 2^53),
  3. runs ONE cblas_dsyrk per prime on the k x N residue matrix (bigint_syrk_blas.cxx:62; off-diagonal blocks of a
     split Q use cblas_dgemm :36),
  4. recombines the N(N+1)/2 outputs by the Chinese remainder theorem (restore_bigint_from_residues.hxx:28,
     fmpz_multi_CRT_ui) and reduces over nodes (restore_and_reduce.cxx:34-219).
FLINT is a third-party dependency that is absent from /root/reference (vasdommes/flint:main, unpinned,
Dockerfile:27-28); steps 1-4 restate its published fmpz_mat_mul_blas recipe, which Fmpz_Comb.cxx says it adapts.

Restated here on numpy/OpenBLAS so that bench.py can time the reference's algorithm for this stage on the GPU box's
host cores beside the GMP port (oracle/sdpb_oracle.cpp: syrk_Q, one mpz_addmul per term), and checked against exact
Python integers in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
import time
from typing import List, Sequence

import numpy as np

DIGIT_BITS = 24          # P' is handed over as base-2^24 digits: digit x (2^(24 i) mod p) sums stay exact in fp64
MAX_BLAS_DP_INT = 1 << 53


def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    if n % 2 == 0:
        return n == 2
    r = int(math.isqrt(n))
    f = 3
    while f <= r:
        if n % f == 0:
            return False
        f += 2
    return True


def calculate_primes(bits: int, k: int) -> List[int]:
    """Fmpz_Comb.cxx:23-73 (_calculate_primes)."""
    p = 2 + 2 * math.isqrt((MAX_BLAS_DP_INT - 1) // k)
    if bits > 200:
        p = min(p, 1664544)
    primes, prod = [], 1
    while True:
        while True:
            if p < 1000:
                raise ValueError(f"Failed to calculate primes for bits={bits}, k={k}")
            p -= 1
            if _is_prime(p):
                break
        primes.append(p)
        prod *= p
        if prod.bit_length() > bits:
            return primes


def output_bits(abits: int, bbits: int, k: int, sign: int = 1) -> int:
    """Fmpz_Comb.cxx:14-18 (calculate_output_bits)."""
    return abits + bbits + k.bit_length() + sign


def to_digits(values: Sequence[int], ndigits: int) -> np.ndarray:
    """Signed integers -> (len, ndigits) float64 base-2^24 digits of |v| and a sign vector (test helper)."""
    out = np.zeros((len(values), ndigits), dtype=np.float64)
    sign = np.ones(len(values), dtype=np.float64)
    mask = (1 << DIGIT_BITS) - 1
    for i, v in enumerate(values):
        if v < 0:
            sign[i] = -1.0
            v = -v
        for d in range(ndigits):
            out[i, d] = v & mask
            v >>= DIGIT_BITS
        assert v == 0, "value needs more digits"
    return out, sign


def residues(digits: np.ndarray, sign: np.ndarray, primes: Sequence[int]) -> np.ndarray:
    """Step 2: (count, ndigits) digits -> (len(primes), count) centred residues in fp64.
    digit < 2^24, weight < p/2 < 2^20, at most 44 digits: every partial sum is an integer below 2^50 (exact)."""
    nd = digits.shape[1]
    assert nd * (1 << DIGIT_BITS) * max(primes) < MAX_BLAS_DP_INT * 2
    W = np.empty((nd, len(primes)), dtype=np.float64)
    for q, p in enumerate(primes):
        for d in range(nd):
            w = pow(2, DIGIT_BITS * d, p)
            W[d, q] = w - p if w > p // 2 else w
    r = digits @ W                              # one GEMM: (count, nd) x (nd, primes)
    r *= sign[:, None]
    pr = np.asarray(primes, dtype=np.float64)[None, :]
    r -= np.rint(r / pr) * pr                   # centred remainder; |r| <= p/2 (r/pr is far from a rounding tie or the result still is a valid residue)
    return np.ascontiguousarray(r.T)


def syrk_residues(res: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """Step 3: one dsyrk per prime.  res[q] is the rows x cols residue matrix, row-major.  Returns (primes, cols, cols)
    with the LOWER triangle filled (exact integers in fp64: |sum| <= rows p^2 / 4 < 2^53)."""
    from scipy.linalg.blas import dsyrk
    out = np.empty((res.shape[0], cols, cols), dtype=np.float64)
    for q in range(res.shape[0]):
        a = res[q].reshape(rows, cols)          # C-contiguous rows x cols == Fortran cols x rows: A^T
        out[q] = dsyrk(1.0, a.T, trans=0, lower=1)   # (A^T)(A^T)^T = A^T A, cols x cols
    return out


def crt(residue_values: Sequence[int], primes: Sequence[int]) -> int:
    """Step 4 for one entry: the signed integer of smallest magnitude with the given residues
    (restore_bigint_from_residues.hxx:28 passes sign = 1 to fmpz_multi_CRT_ui)."""
    M = 1
    for p in primes:
        M *= p
    x = 0
    for r, p in zip(residue_values, primes):
        Mp = M // p
        x += (int(r) % p) * Mp * pow(Mp, -1, p)
    x %= M
    return x - M if x > M // 2 else x


def int_syrk(values_colmajor: Sequence[int], rows: int, cols: int, value_bits: int) -> List[List[int]]:
    """The whole stage on exact integers (small sizes: tests).  values: P' column-major, |v| < 2^value_bits.
    Returns the lower triangle of P'^T P' as a dense list of lists (upper part 0)."""
    bits = output_bits(value_bits, value_bits, rows)
    primes = calculate_primes(bits, rows)
    nd = -(-value_bits // DIGIT_BITS)
    rowmajor = [values_colmajor[r + c * rows] for r in range(rows) for c in range(cols)]
    digits, sign = to_digits(rowmajor, nd)
    res = residues(digits, sign, primes)
    prod = syrk_residues(res, rows, cols)
    out = [[0] * cols for _ in range(cols)]
    for i in range(cols):
        for j in range(i + 1):
            out[i][j] = crt([int(prod[q, i, j]) for q in range(len(primes))], primes)
    return out


# ---- steps 2 and 4 compiled, on GMP (oracle/sdpb_oracle.cpp: orc_refq_residues, orc_refq_crt) -------------------------
def _lib():
    import ctypes
    from oracle.oracle import lib
    L = lib()
    if not getattr(L, "_refq_ready", False):
        ulp = ctypes.POINTER(ctypes.c_ulong)
        L.orc_refq_residues.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_ulong, ctypes.c_ulong, ulp, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.orc_refq_crt.argtypes = [ctypes.c_void_p, ctypes.c_long, ulp, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t), ulp, ctypes.POINTER(ctypes.c_double)]
        L._refq_ready = True
    return L


def residues_gmp(values: Sequence[int] | None, count: int, bits: int, primes: Sequence[int], seed: int = 7, first: int = 0):
    """Step 2 with mpz_fdiv_ui per (entry, prime), OpenMP over entries (orc_set_threads / $ORACLE_THREADS).
    values None: `count` pseudo-random entries |v| < 2^bits generated inside (not timed).  -> ((primes, count) float64, seconds)."""
    import ctypes
    L = _lib()
    out = np.empty((len(primes), count), dtype=np.float64)
    pr = (ctypes.c_ulong * len(primes))(*primes)
    sec = ctypes.c_double(0.0)
    txt = None if values is None else " ".join(str(int(v)) for v in values).encode()
    rc = L.orc_refq_residues(txt, count, bits, seed, first, pr, len(primes), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(sec))
    if rc:
        raise RuntimeError(f"orc_refq_residues failed ({rc})")
    return out, sec.value


def crt_gmp(prod: np.ndarray, primes: Sequence[int], want_values: bool = False):
    """Step 4: prod (primes, n) float64 exact integers -> (list of n Python ints or None, seconds, checksum);
    mpz_addmul_ui over the primes + one mpz_fdiv_r per output, OpenMP over outputs."""
    import ctypes
    L = _lib()
    prod = np.ascontiguousarray(prod, dtype=np.float64)
    n = prod.shape[1]
    pr = (ctypes.c_ulong * len(primes))(*primes)
    sec, chk, need = ctypes.c_double(0.0), ctypes.c_ulong(0), ctypes.c_size_t(0)
    buf = ctypes.create_string_buffer(n * (int(sum(math.log10(p) for p in primes)) + 4) + 16) if want_values else None
    rc = L.orc_refq_crt(prod.ctypes.data_as(ctypes.c_void_p), n, pr, len(primes), buf, len(buf) if buf else 0,
                        ctypes.byref(need) if want_values else None, ctypes.byref(chk), ctypes.byref(sec))
    if rc:
        raise RuntimeError(f"orc_refq_crt failed ({rc})")
    vals = [int(t) for t in buf.value.decode().split()] if want_values else None
    return vals, sec.value, chk.value


def int_syrk_gmp(values_colmajor: Sequence[int], rows: int, cols: int, value_bits: int) -> List[List[int]]:
    """int_syrk with the compiled steps 2 and 4 (small sizes: tests)."""
    primes = calculate_primes(output_bits(value_bits, value_bits, rows), rows)
    rowmajor = [values_colmajor[r + c * rows] for r in range(rows) for c in range(cols)]
    res, _ = residues_gmp(rowmajor, rows * cols, value_bits, primes)
    prod = syrk_residues(res, rows, cols)
    tri = [(i, j) for i in range(cols) for j in range(i + 1)]
    flat = np.stack([np.array([prod[q, i, j] for i, j in tri]) for q in range(len(primes))])
    vals, _, _ = crt_gmp(flat, primes, want_values=True)
    out = [[0] * cols for _ in range(cols)]
    for (i, j), v in zip(tri, vals):
        out[i][j] = v
    return out


def time_q_stage_gmp(rows: int, cols: int, precision: int, total_rows: int, seed: int = 7, threads: int = 0,
                     chunk_rows: int = 4000) -> dict:
    """Wall time of the WHOLE stage, steps 2-4, the reference's way: residues on GMP (compiled, all threads), one dsyrk per
    prime on a multi-threaded OpenBLAS (accumulated over row chunks, beta = 1: the sums of `total_rows` centred residue
    products stay below 2^53 by the choice of the primes), CRT on GMP for all cols (cols + 1) / 2 outputs.  Steps 2 and 3
    are exactly linear in the rows (`rows` of `total_rows` are run, the caller scales); step 4 does not depend on them."""
    import os
    from scipy.linalg.blas import dsyrk
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                      # pragma: no cover
        from contextlib import contextmanager

        @contextmanager
        def threadpool_limits(limits=None):
            yield
    from oracle.oracle import lib
    threads = lib().orc_set_threads(int(threads or len(os.sched_getaffinity(0))))
    primes = calculate_primes(output_bits(precision, precision, total_rows), total_rows)
    acc = [np.zeros((cols, cols), dtype=np.float64, order="F") for _ in primes]
    t_res = t_syrk = 0.0
    for r0 in range(0, rows, chunk_rows):
        nr = min(chunk_rows, rows - r0)
        res, sec = residues_gmp(None, nr * cols, precision, primes, seed, first=r0 * cols)
        t_res += sec
        t1 = time.perf_counter()
        with threadpool_limits(limits=threads):
            for q in range(len(primes)):
                a = res[q].reshape(nr, cols)        # C-contiguous rows x cols == Fortran cols x rows: A^T
                acc[q] = dsyrk(1.0, a.T, beta=1.0, c=acc[q], trans=0, lower=1, overwrite_c=1)
        t_syrk += time.perf_counter() - t1
    iu = np.tril_indices(cols)
    flat = np.stack([a[iu] for a in acc])
    _, t_crt, chk = crt_gmp(flat, primes)
    return {"primes": len(primes), "prime_bits": round(math.log2(primes[0]), 2), "rows": rows, "cols": cols, "threads": threads,
            "residues_s": t_res, "dsyrk_s": t_syrk, "crt_s": t_crt, "crt_outputs": int(flat.shape[1]), "crt_included": True,
            "seconds_linear_in_rows": t_res + t_syrk, "checksum": chk}


def time_q_stage(rows: int, cols: int, precision: int, total_rows: int, seed: int = 7, crt_sample: int = 2000,
                 threads: int = 0) -> dict:
    """Wall time of steps 2-4 for a rows x cols slice of a P' with `total_rows` rows in all (the primes depend on
    the TOTAL height, as in the reference) at --precision `precision`, on `threads` host threads (0: all usable).
    Step 2 runs row chunks on a thread pool with single-threaded BLAS inside (numpy releases the GIL; the reference
    converts its blocks on all ranks at once), step 3 hands each dsyrk to a multi-threaded OpenBLAS (the reference
    schedules its BLAS jobs over the ranks of a node, create_blas_job_schedule.cxx:149-238).  Steps 2 and 3 are
    exactly linear in the number of rows; step 4 does not depend on it and is timed on `crt_sample` entries
    (pure-Python big integers, one thread — FLINT's is compiled and runs on every rank, so the figure is reported
    separately and NOT included in `seconds`)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                      # pragma: no cover
        from contextlib import contextmanager

        @contextmanager
        def threadpool_limits(limits=None):
            yield
    threads = threads or len(os.sched_getaffinity(0))
    bits = output_bits(precision, precision, total_rows)
    primes = calculate_primes(bits, total_rows)
    nd = -(-precision // DIGIT_BITS)
    chunk = max(1, (1 << 20) // cols)           # ~1M entries per conversion GEMM
    spans = [(r0, min(rows, r0 + chunk)) for r0 in range(0, rows, chunk)]
    res = np.empty((len(primes), rows * cols), dtype=np.float64)

    def make(span):                              # synthetic P' digits (not timed: the reference is handed its P')
        rng = np.random.default_rng(seed + span[0])
        n = (span[1] - span[0]) * cols
        return (rng.integers(0, 1 << DIGIT_BITS, size=(n, nd)).astype(np.float64),
                np.where(rng.integers(0, 2, size=n) == 0, 1.0, -1.0))

    def convert(args):
        span, (digits, sign) = args
        res[:, span[0] * cols:span[1] * cols] = residues(digits, sign, primes)

    t_res = 0.0
    with threadpool_limits(limits=1), ThreadPoolExecutor(threads) as pool:
        for g0 in range(0, len(spans), 2 * threads):   # bounded memory: 2 x threads chunks of digits at a time
            group = spans[g0:g0 + 2 * threads]
            inputs = list(pool.map(make, group))
            t0 = time.perf_counter()
            list(pool.map(convert, zip(group, inputs)))
            t_res += time.perf_counter() - t0
    t1 = time.perf_counter()
    with threadpool_limits(limits=threads):
        prod = syrk_residues(res, rows, cols)
    t_syrk = time.perf_counter() - t1
    t2 = time.perf_counter()
    rng = np.random.default_rng(seed)
    n_crt = min(crt_sample, cols * (cols + 1) // 2)
    idx = rng.integers(0, cols, size=(n_crt, 2))
    for i, j in idx:
        i, j = max(i, j), min(i, j)
        crt([int(prod[q, i, j]) for q in range(len(primes))], primes)
    t_crt = (time.perf_counter() - t2) * (cols * (cols + 1) // 2) / max(n_crt, 1)
    return {"primes": len(primes), "prime_bits": round(math.log2(primes[0]), 2), "rows": rows, "cols": cols, "threads": threads,
            "residues_s": t_res, "dsyrk_s": t_syrk, "seconds": t_res + t_syrk,
            "crt_python_one_thread_s_not_included": t_crt}
