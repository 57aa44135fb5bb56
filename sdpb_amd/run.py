"""`python -m sdpb_amd.run -s <sdpDir> --precision <p> -o <outDir> [sdpb solver options]`

The driver that sits where `sdpb`'s `solve()` sits (src/sdpb/solve.cxx:23-106): reads the unchanged
SDP directory format, runs the device-resident interior-point loop through the C ABI and writes
sdpb's own result files:

  out/out.txt            terminateReason, primalObjective, ... (src/sdpb/save_solution.cxx:29-39)
  out/iterations.json    one record per iteration (print_iteration.cxx:77-108)
  out/y.txt, out/z.txt   (save_solution.cxx:41-107; z needs normalization.json)
  out/x_<j>.txt, X_matrix_<q>.txt, Y_matrix_<q>.txt  (save_solution.cxx:123-149, write_distmatrix.hxx)

Option names and defaults are sdpb's (Solver_Parameters.cxx:10-157, SDPB_Parameters.cxx:21-82).
Only I/O and the loop live here; every number is computed on the GPU.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from typing import List, Optional

from .sdp_io import read_sdp
from .solver import ITERATION_KEYS, PARAM_NAMES, SDPSolver

HEADER = ("\n          time    mu     P-obj       D-obj      gap         P-err       p-err       D-err"
          "      P-step   D-step   beta\n" + "-" * 117)


def _options(argv: Optional[List[str]] = None):
    ap = argparse.ArgumentParser(prog="sdpb_amd.run", description=__doc__.split("\n\n")[0])
    ap.add_argument("-s", "--sdpDir", required=True)
    ap.add_argument("-o", "--outDir", default=None, help="default: <sdpDir>_out")
    ap.add_argument("--precision", type=int, default=400,
                    help="bits of the mantissa, rounded up to the next compiled width (128, 256, 448, 512, 704, 768, 1024, 1280, 1536, 2048: "
                         "GMP rounds up to whole limbs too, Solver_Parameters.cxx:26); above 2048 the library refuses with a message "
                         "naming this range (the reference accepts any precision).  The exact integer Q' is formed from a fixed-point "
                         "image of at least precision - 32 bits (sdpb_hip_fx_frac_bits); pass precision + 64 for at least `precision`")
    ap.add_argument("--maxSharedMemory", default="0",
                    help="bound on the scratch of the Q stage like sdpb's option (bytes, or with a suffix K/M/G as sdpb accepts, e.g. "
                         "100.1K): here the partial planes of the exact integer syrk; Q' is then computed in chunks of output tiles "
                         "that fit, with identical results (default 0: what is free on the device, at most 1/8 of it)")
    ap.add_argument("--maxIterations", type=int, default=500)
    ap.add_argument("--maxRuntime", type=float, default=float("inf"))
    for name, default in (("dualityGapThreshold", "1e-30"), ("primalErrorThreshold", "1e-30"),
                          ("dualErrorThreshold", "1e-30"), ("initialMatrixScalePrimal", "1e20"),
                          ("initialMatrixScaleDual", "1e20"), ("feasibleCenteringParameter", "0.1"),
                          ("infeasibleCenteringParameter", "0.3"), ("stepLengthReduction", "0.7"),
                          ("maxComplementarity", "1e100"), ("minPrimalStep", "0"), ("minDualStep", "0")):
        ap.add_argument("--" + name, default=default)
    for flag in ("findPrimalFeasible", "findDualFeasible", "detectPrimalFeasibleJump", "detectDualFeasibleJump"):
        ap.add_argument("--" + flag, action="store_true")
    ap.add_argument("--writeSolution", default="x,y", help="comma separated subset of x,y,X,Y,z")
    ap.add_argument("-i", "--initialCheckpointDir", default=None,
                    help="text checkpoint: a directory written with --writeSolution=x,y,X,Y (load_text_checkpoint.cxx:6-44)")
    ap.add_argument("-c", "--checkpointDir", default=None,
                    help="where block_timings is written and looked for (default: <sdpDir>.ck, as sdpb)")
    ap.add_argument("--verbosity", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--noFinalCheckpoint", action="store_true", help="accepted for compatibility")
    ap.add_argument("--lib", default=None, help="path of the C-ABI library (tests: the CPU emulation build)")
    return ap.parse_args(argv)


def parse_memory_size(text: str) -> int:
    """--maxSharedMemory as sdpb writes it: a number of bytes with an optional K / M / G suffix or B, e.g. 100.1K
    (Solver_Parameters.cxx:61-72: optional suffixes B, K or KB, M or MB, G or GB; test/src/end-to-end.test.cxx:348-357)."""
    t = str(text).strip().upper().rstrip("B")
    mult = 1
    if t and t[-1] in "KMGT":
        mult = 1024 ** ("KMGT".index(t[-1]) + 1)
        t = t[:-1]
    return int(float(t or 0) * mult)


def parse_option_like_sdpb(text: str) -> str:
    """sdpb parses its numeric options before --precision is applied, i.e. with GMP's start-up
    64-bit mpf precision (visible in the golden traces: beta = 0.2999...98725e-58 for
    --infeasibleCenteringParameter 0.3).  mpf_set_str keeps prec+1 = 3 limbs of 64 bits,
    limb-aligned and truncated toward zero; this returns the exact decimal expansion of that
    number so the device solver starts from the same parameter VALUES as the reference.
    (For |decimal exponent| > 57 GMP also truncates the power of ten it multiplies by; the
    result then differs from this one by < 2^-128 relative — thresholds only.)"""
    from fractions import Fraction
    t = text.strip().lower()
    mant, _, e = t.partition("e")
    e10 = int(e) if e else 0
    neg = mant.startswith("-")
    mant = mant.lstrip("+-")
    ip, _, fp = mant.partition(".")
    v = Fraction(int((ip + fp) or "0")) * Fraction(10) ** (e10 - len(fp))
    if v == 0:
        return "0"
    num, den = v.numerator, v.denominator
    bl = num.bit_length() - den.bit_length()
    if bl >= 0:
        fl = bl if num >= (den << bl) else bl - 1
    else:
        fl = bl if (num << -bl) >= den else bl - 1
    shift = 64 * (3 - (fl // 64 + 1))          # value * 2^shift fills exactly three limbs
    q = (num << shift) // den if shift >= 0 else (num // (den << -shift)) << -shift
    shift = max(shift, 0)
    digits = str(q * 5 ** shift).rjust(shift + 1, "0")   # q / 2^shift = q 5^shift / 10^shift
    out = digits[:-shift] + "." + digits[-shift:] if shift else digits
    return ("-" if neg else "") + out


def _digits(precision: int) -> int:
    # set_stream_precision: ceil(precision * log10(2)) + 1 significant digits
    return int(math.ceil(precision * math.log10(2.0))) + 1


def _write_vector(path: str, values: List[str]):
    with open(path, "w") as f:
        f.write(f"{len(values)} 1\n" + "\n".join(values) + "\n\n")


def _write_matrix(path: str, colmajor: List[str], n: int):
    with open(path, "w") as f:
        f.write(f"{n} {n}\n")
        for i in range(n):
            f.write(" ".join(colmajor[i + j * n] for j in range(n)) + "\n")
        f.write("\n")


def _z_from_y(y: List[str], normalization: List[str], precision: int) -> List[str]:
    """save_solution.cxx:67-107: insert the component fixed by n.z = 1 at max |n_i|."""
    import mpmath
    with mpmath.workprec(precision + 64):
        n = [mpmath.mpf(v) for v in normalization]
        yv = [mpmath.mpf(v) for v in y]
        k = max(range(len(n)), key=lambda i: abs(n[i]))
        z = yv[:k] + [mpmath.mpf(0)] + yv[k:]
        nz = mpmath.fsum(a * b for a, b in zip(n, z))
        z[k] = (1 - nz) / n[k]
        return [mpmath.nstr(v, _digits(precision), strip_zeros=False) for v in z]


def _read_numbers(path: str, rows: int, cols: int) -> List[str]:
    with open(path) as f:
        tok = f.read().split()
    if int(tok[0]) != rows or int(tok[1]) != cols or len(tok) != 2 + rows * cols:
        raise ValueError(f"{path}: expected a {rows} x {cols} block")
    return tok[2:]


def load_text_checkpoint(solver: SDPSolver, sdp, directory: str):
    """x_<j>.txt, y.txt, X_matrix_<q>.txt, Y_matrix_<q>.txt -> solver state (load_text_checkpoint.cxx:6-44)."""
    solver.set_array("y", _read_numbers(os.path.join(directory, "y.txt"), sdp.N, 1))
    for j, blk in enumerate(sdp.blocks):
        solver.set_array("x", _read_numbers(os.path.join(directory, f"x_{j}.txt"), blk.schur_size, 1), j)
        for parity, n in enumerate(blk.psd_sizes):
            if n == 0:
                continue
            for name in ("X", "Y"):
                rowmajor = _read_numbers(os.path.join(directory, f"{name}_matrix_{2 * j + parity}.txt"), n, n)
                solver.set_array(name, [rowmajor[i * n + jj] for jj in range(n) for i in range(n)], j, parity)


def solve(argv: Optional[List[str]] = None) -> str:
    o = _options(argv)
    out_dir = o.outDir or (o.sdpDir.rstrip("/") + "_out")
    os.makedirs(out_dir, exist_ok=True)
    sdp = read_sdp(o.sdpDir, o.precision)
    params = {k: parse_option_like_sdpb(getattr(o, k)) for k in PARAM_NAMES}
    params.update(maxIterations=o.maxIterations, findPrimalFeasible=int(o.findPrimalFeasible),
                  findDualFeasible=int(o.findDualFeasible),
                  detectPrimalFeasibleJump=int(o.detectPrimalFeasibleJump),
                  detectDualFeasibleJump=int(o.detectDualFeasibleJump))
    start = time.time()
    # measured block costs of an earlier run balance the blocks over the GPUs
    # (Block_Info/read_block_costs.cxx:14-59: the checkpoint's file wins over the SDP's)
    ck_dir = o.checkpointDir or (o.sdpDir.rstrip("/") + ".ck")
    costs = None
    for cand in (os.path.join(ck_dir, "block_timings"), os.path.join(o.sdpDir, "block_timings")):
        if os.path.isfile(cand):
            with open(cand) as f:
                costs = [int(t) for t in f.read().split()]
            break
    solver = SDPSolver(sdp, o.precision, params, device=o.device, lib_path=o.lib, block_costs=costs)
    timing_run = costs is None   # run.cxx:442-453: iteration 1 is unrepresentative, iteration 2 is timed
    if o.initialCheckpointDir:
        load_text_checkpoint(solver, sdp, o.initialCheckpointDir)
    # --maxRuntime is tested inside the iteration, where the reference tests it
    # (compute_feasible_and_termination.cxx:51-56), against the time left of the budget
    if o.maxRuntime < float("inf"):
        solver.set_max_runtime(max(0.0, o.maxRuntime - (time.time() - start)))
    # SIGTERM: finish gracefully at the next iteration boundary (run.cxx:332-355, Environment.cxx:58-59)
    import signal
    previous_handler = None
    try:
        previous_handler = signal.signal(signal.SIGTERM, lambda *_: solver.request_stop())
    except ValueError:  # not the main thread (tests): the caller may still use solver.request_stop()
        pass
    shared = parse_memory_size(o.maxSharedMemory)
    if shared:
        solver.set_max_shared_memory(shared)
    if o.verbosity >= 2:
        # the reference prints its memory estimates at this verbosity (run.cxx:79-181)
        plan = solver.memory_plan()
        print("Memory plan of this rank (bytes): " + ", ".join(f"{k} {v}" for k, v in plan["bytes"].items())
              + f"; total {sum(plan['bytes'].values())} of {plan['device']['total_bytes']} on the device")
        # the two windows of the Q stage that --maxSharedMemory bounds together (BigInt_Shared_Memory_Syrk_Context.cxx:70-215)
        print(f"Q' = P'^T P' windows: window_budget_bytes {plan.get('window_budget_bytes')}; input (image of P'): "
              + ", ".join(f"{k} {v}" for k, v in plan.get("image", {}).items()))
        print("Q' = P'^T P' output (partial planes): " + ", ".join(f"{k} {v}" for k, v in plan["syrk"].items()))
    if o.verbosity >= 1:
        print(f"Initialize SDP solver\n\tprimal dimension: {sdp.P_total}\n\tdual dimension: {sdp.N}"
              f"\n\tSDP blocks: {sdp.J}")
        print(HEADER)
    it_path = os.path.join(out_dir, "iterations.json")
    reason = None
    with open(it_path, "w") as itf:
        itf.write("[")
        while reason is None:
            t_it = time.time()
            if timing_run and solver.iteration == 1:
                solver.set_profiling(True)
            if solver.iterate():
                reason = solver.terminate_reason
                break
            if timing_run and solver.iteration == 2:
                # write_timing.cxx:34-68: one integer per block and line, in <checkpointDir>/block_timings
                solver.set_profiling(False)
                timing_run = False
                os.makedirs(ck_dir, exist_ok=True)
                with open(os.path.join(ck_dir, "block_timings"), "w") as f:
                    f.write("".join(f"{t}\n" for t in solver.block_timings()))
            rec = solver.scalars()
            now = time.time()
            itf.write(("," if solver.iteration > 1 else "") + "\n{ \"iteration\":%d, \"total_time\": %.3f, "
                      "\"iter_time\": %.3f" % (solver.iteration, now - start, now - t_it)
                      + "".join(f", \"{k}\": \"{rec[k]}\"" for k in ITERATION_KEYS)
                      + f", \"block_name\": \"{rec['block_name']}\" }}")
            if o.verbosity >= 1:
                f = lambda k: float(rec[k])  # noqa: E731
                print(f"{solver.iteration:<4d}  {int(now - start):>8d} {f('mu'):<8.2g} {f('P-obj'):<+11.3g} "
                      f"{f('D-obj'):<+11.3g} {f('gap'):<10.3g} {f('P-err'):<+11.3g} {f('p-err'):<+11.3g} "
                      f"{f('D-err'):<+11.3g} {f('P-step'):<8.3g} {f('D-step'):<8.3g} {f('beta'):<4.3g}", flush=True)
        itf.write("\n]")
    if previous_handler is not None:
        signal.signal(signal.SIGTERM, previous_handler)
    runtime = int(time.time() - start)
    out = solver.out_txt()
    out["terminateReason"] = reason
    with open(os.path.join(out_dir, "out.txt"), "w") as f:
        f.write(f"terminateReason = \"{reason}\";\n"
                f"primalObjective = {out['primalObjective']};\ndualObjective   = {out['dualObjective']};\n"
                f"dualityGap      = {out['dualityGap']};\nprimalError     = {out['primalError']};\n"
                f"dualError       = {out['dualError']};\nSolver runtime  = {runtime};\n")
    # out/c_minus_By/c_minus_By.json (save_c_minus_By.hxx; read by `spectrum`)
    os.makedirs(os.path.join(out_dir, "c_minus_By"), exist_ok=True)
    with open(os.path.join(out_dir, "c_minus_By", "c_minus_By.json"), "w") as f:
        f.write("{\"c_minus_By\":[" + ",".join(
            "[" + ",".join(f"\"{v}\"" for v in solver.array("c_minus_By", j)) + "]" for j in range(sdp.J)) + "]}")
    what = set(w for w in o.writeSolution.split(",") if w)
    y = solver.array("y")
    if "y" in what:
        _write_vector(os.path.join(out_dir, "y.txt"), y)
    if "z" in what and sdp.normalization:
        _write_vector(os.path.join(out_dir, "z.txt"), _z_from_y(y, sdp.normalization, o.precision))
    for j, blk in enumerate(sdp.blocks):
        if "x" in what:
            _write_vector(os.path.join(out_dir, f"x_{j}.txt"), solver.array("x", j))
        for parity, n in enumerate(blk.psd_sizes):
            if n == 0:
                continue
            for name in ("X", "Y"):
                if name in what:
                    _write_matrix(os.path.join(out_dir, f"{name}_matrix_{2 * j + parity}.txt"),
                                  solver.array(name, j, parity), n)
    if o.verbosity >= 1:
        print(f"-----{reason}-----\n")
        for k in ("primalObjective", "dualObjective", "dualityGap", "primalError", "dualError"):
            print(f"{k:16s}= {out[k]}")
        print(f"Saving solution to      : {out_dir}")
    solver.close()
    return reason


if __name__ == "__main__":
    solve()
    sys.exit(0)
