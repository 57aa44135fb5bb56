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
