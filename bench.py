#!/usr/bin/env python
"""bench.py — interior-point iterations/sec at --precision 512 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path = one interior-point iteration (run.cxx:380-467 incl.
SDP_Solver::step) over the synthetic "3d Ising mixed-correlator" SDP C4 of SURVEY.md §8d
(J=600 blocks, N=1000, P_tot=40000, --precision 512), the configuration BASELINE.json's
metric is quoted on; it fits one GPU, so it is the workload at every N (strong scaling:
blocks shard across ranks, Q' is summed with one integer all-reduce).  Inputs are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LIMB_MAC_PEAK = 19.3e12        # measured on MI355X: v_mad_u64_u32 + v_addc_co_u32 pairs/s at >= 2 wavefronts per SIMD (profiles/r04_ubench_macpairs.txt;
                               # the 17.0e12 of profiles/r01_ubench.txt, used until the last build of round 4, came from a loop of 8 pairs per 19 instructions)
LIMB_MAD_PEAK = 37.6e12        # measured on MI355X: v_mad_u64_u32 ALONE (the lazy-carry kernels: 28-/27-bit limbs, no v_addc) at 4 wavefronts per SIMD,
                               # 34.7e12 at 2 (profiles/r06d_ubench_mad_only.txt): 4.2 cycles per instruction and SIMD against 8.2 for the pair


def _probe_reference_binary():
    """a real `sdpb` ($SDPB_BIN or PATH) + mpirun would be the literal
    Elemental+MPI baseline.  Returns (sdpb, mpirun) or None."""
    import shutil
    sdpb = os.environ.get("SDPB_BIN") or shutil.which("sdpb")
    mpirun = shutil.which("mpirun") or shutil.which("mpiexec")
    if sdpb and os.path.exists(sdpb) and mpirun:
        return sdpb, mpirun
    return None


def _reference_binary_baseline(cfg_name, precision, found, seconds_budget):
    """Time the real sdpb on the same SDP written in its own directory format."""
    import subprocess
    import tempfile
    from sdpb_amd import synthetic
    from sdpb_amd.sdp_io import write_sdp
    sdpb, mpirun = found
    scale = float(os.environ.get("SDPB_BENCH_CPU_SCALE", "0.25"))
    c = synthetic.config(cfg_name, scale)
    sdp = synthetic.make_sdp(c["dims"], c["num_points"], c["N"], precision, c["seed"])
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as tmp:
        write_sdp(sdp, os.path.join(tmp, "sdp"))
        cmd = [mpirun, "-n", str(min(cores, sdp.J)), sdpb, "-s", os.path.join(tmp, "sdp"), "--precision", str(precision),
               "--maxIterations", "6", "--noFinalCheckpoint", "--verbosity", "1", "-o", os.path.join(tmp, "out")]
        subprocess.run(cmd, check=True, timeout=max(seconds_budget * 4, 120), capture_output=True)
        with open(os.path.join(tmp, "out", "iterations.json")) as f:
            its = json.load(f)
    times = sorted(float(r["iter_time"]) for r in its[1:])
    dt = times[len(times) // 2]
    return dt, sdp, min(cores, sdp.J), scale, c, f"{sdpb} under {mpirun}"


def cpu_baseline(cfg_name: str, precision: int, seconds_budget: float = 45.0, full: bool = False):
    """The CPU path timed beside the GPU one, on this node's host cores (count stated).

    First choice: a real `sdpb` binary ("kind": "reference").  It cannot be built
    in this image, so normally the parity oracle is timed instead ("kind": "port"): oracle/
    sdpb_oracle.cpp, the GMP-mpf restatement of the reference iteration, OpenMP over SDP blocks /
    matrix columns / Q entries on ALL host cores (bit-identical to its 1-thread run).  The sample is a
    structurally identical, proportionally smaller SDP (same block mix, J and N scaled by 0.5: 1/7.6 of the
    full iteration by the multi-word MAC model of sdpb_amd/workmodel.py; `--cpu-full` runs the full size
    instead, minutes per iteration); `value` scales it to the full workload by that model and the factor is
    reported (`extrapolation_factor`; 1.0 with --cpu-full).

    The port's dominant stage is NOT the reference's algorithm: it forms Q' = P'^T P' with one mpz_addmul per
    term, the reference with CRT residues + one fp64 dsyrk per prime (bigint_syrk_blas.cxx:183-302,
    Changelog.md:71).  Both are timed: `q_stage_port_s` (the mpz stage inside the sample iterations, scaled to the
    full rows/columns by its own MAC count) and `q_stage_reference_algorithm_s` (oracle/bigint_syrk_blas.py: the
    reference's recipe on numpy/OpenBLAS — primes as Fmpz_Comb.cxx, residue conversion, 53 dsyrk calls at FULL N on a
    row slice, exactly linear in the rows).  `value` is the port as measured; `value_with_reference_q_stage`
    replaces the port's Q stage by the reference's algorithm — the fairer estimate of what the reference itself
    would reach on these cores, and the number to hold the GPU figure against."""
    from sdpb_amd import synthetic, workmodel
    fullc = synthetic.config(cfg_name)
    wm_full = workmodel.macs_per_iteration(fullc["dims"], fullc["num_points"], fullc["N"])
    w_full = wm_full["total"]
    found = _probe_reference_binary()
    if found:
        try:
            dt, sdp, cores, scale, c, what = _reference_binary_baseline(cfg_name, precision, found, seconds_budget)
            w_s = workmodel.macs_per_iteration(c["dims"], c["num_points"], c["N"])["total"]
            return {"value": 1.0 / (dt * w_full / w_s), "unit": "iterations/s", "cores": cores, "kind": "reference",
                    "extrapolation_factor": w_full / w_s, "sample_iterations_per_s": 1.0 / dt,
                    "sample": f"{what} on {cfg_name} x{scale}: J={sdp.J}, N={sdp.N}; median iter_time {dt:.3f} s"}
        except Exception as e:  # fall through to the port, say why
            probe_note = f"sdpb binary found but unusable ({type(e).__name__}: {e}); "
    else:
        probe_note = "no sdpb binary on $SDPB_BIN/PATH; "
    from oracle import bigint_syrk_blas as ref_q
    from oracle.oracle import Oracle, usable_cpus
    cores = usable_cpus()   # cgroup quota and affinity, not just the logical CPU count
    scale = 1.0 if full else float(os.environ.get("SDPB_BENCH_CPU_SCALE", "0.5"))
    c = synthetic.config(cfg_name, scale)
    sdp, src = synthetic.make_lazy(c["dims"], c["num_points"], c["N"], precision, c["seed"], feasible=c.get("feasible", False))
    t_setup = time.time()
    o = Oracle(sdp, precision, param_prec=0, threads=cores, block_source=src)
    t_setup = time.time() - t_setup
    o.iterate()  # iteration 1 is unrepresentative (X, Y diagonal): run.cxx:442-453
    q0 = float(o.scalar("seconds.syrk_Q"))
    times = []
    t0 = time.time()
    want = 1 if full else 3
    while len(times) < want or (not full and len(times) < 10 and time.time() - t0 < seconds_budget):
        t = time.time()
        assert not o.iterate()
        times.append(time.time() - t)
    q_port_sample = (float(o.scalar("seconds.syrk_Q")) - q0) / len(times)
    threads = o.threads
    o.close()
    times.sort()
    dt = times[len(times) // 2]
    wm_s = workmodel.macs_per_iteration(c["dims"], c["num_points"], c["N"])
    factor = w_full / wm_s["total"]
    t_full = dt * factor
    q_port_full = q_port_sample * wm_full["syrk_Q"] / wm_s["syrk_Q"]
    # the reference's algorithm for the same stage: full N, a row slice (linear in rows), primes for the FULL height
    P_full, N_full = sum(K * m * (m + 1) // 2 for m, K in zip(fullc["dims"], fullc["num_points"])), fullc["N"]
    rows = P_full if full else max(1, min(P_full, int(float(os.environ.get("SDPB_BENCH_REFQ_ROWS", "6000")))))
    # steps 2-4 all timed (round 5): residues and CRT compiled on GMP (oracle/sdpb_oracle.cpp), dsyrk through OpenBLAS; the
    # residues and the dsyrk calls are linear in the rows and scaled, the CRT of the N (N + 1) / 2 outputs is run in full
    rq = ref_q.time_q_stage_gmp(rows, N_full, precision, P_full, threads=cores)
    q_ref_full = rq["seconds_linear_in_rows"] * P_full / rows + rq["crt_s"]
    t_ref = max(t_full - q_port_full, 0.0) + q_ref_full
    return {"value": 1.0 / t_full, "unit": "iterations/s", "cores": threads, "kind": "port",
            "extrapolation_factor": factor, "sample_iterations_per_s": 1.0 / dt,
            "seconds_per_iteration_full_size": t_full,
            "q_stage_port_s": q_port_full, "q_stage_reference_algorithm_s": q_ref_full,
            "q_stage_reference_algorithm": {**{k: (round(v, 3) if isinstance(v, float) else v) for k, v in rq.items()},
                                            "scaled_by_rows": P_full / rows,
                                            "what": "oracle/bigint_syrk_blas.py + oracle/sdpb_oracle.cpp: Fmpz_Comb primes, centred "
                                                    "residues by mpz_fdiv_ui per prime (OpenMP), one scipy.linalg.blas.dsyrk per prime at "
                                                    "full N accumulated over row chunks, CRT of all N(N+1)/2 outputs by mpz_addmul_ui over "
                                                    "the primes (OpenMP): the whole stage, crt_included",
                                            # round-5 advisor: the residues and the CRT are the plain per-prime GMP forms; the reference
                                            # uses FLINT's precomputed remainder / product trees (fmpz_multi_mod, fmpz_multi_CRT_ui),
                                            # which are cheaper -- FLINT is not in this image, so this leg cannot be calibrated here
                                            "bound": "UPPER bound on the reference's time for residues and CRT (no FLINT trees); dsyrk is like for like"},
            "value_with_reference_q_stage": 1.0 / t_ref,
            "sample": probe_note + f"`value` = the PORT as measured: oracle (GMP mpf restatement, OpenMP over blocks/columns, "
                      f"{threads} threads on {cores} host cores) on {cfg_name} x{scale}: J={sdp.J}, N={sdp.N}, P_tot={sdp.P_total}; "
                      f"median of {len(times)} steady-state iteration(s) = {dt:.3f} s (setup {t_setup:.1f} s), of which "
                      f"{q_port_sample:.3f} s in the mpz Q stage; value = sample rate / {factor:.1f} (MAC model "
                      f"{w_full:.3g}/{wm_s['total']:.3g}); value_with_reference_q_stage swaps the port's Q stage "
                      f"({q_port_full:.1f} s at full size) for the reference's CRT+dsyrk algorithm measured here on {rows} of "
                      f"{P_full} rows x N={N_full} ({q_ref_full:.1f} s at full size)"}


def fixture_path(workload: str, scale: float):
    """The deepest committed oracle fixture of the workload: <name>_to_termination.json (every iteration
    until SDP_Solver::run stops, with the terminate reason) if it exists, else <name>.json."""
    name = workload if scale == 1.0 else f"{workload}_x{scale}"
    d = os.path.join(ROOT, "tests", "golden", "synthetic")
    deep = os.path.join(d, f"{name}_to_termination.json")
    return deep if os.path.exists(deep) else os.path.join(d, f"{name}.json")


def parity_gate(solver, workload: str, scale: float, precision: int):
    """"parity gate before any timing counts".  Replays the committed oracle fixture of
    the selected workload (tests/golden/synthetic/<workload>.json: scalars the GMP oracle produced on
    the SAME synthetic SDP, tests/golden/synthetic/make_synthetic_golden.py) from the initial point
    and compares every numeric field of every fixture iteration at relative 2^-(p/2)
    (calculate_matrix_square.test.cxx:213's convention; the fixture is data, the oracle itself is not
    called here).  Every rank iterates (the step is collective); every rank compares the same bits.
    Returns the gate record; the caller refuses to print a value if it failed."""
    path = fixture_path(workload, scale)
    if not os.path.exists(path):
        return {"fixture": None, "passed": None,
                "note": "no committed oracle fixture for this workload/scale: the value below is NOT parity-gated"}
    from tests import parity  # comparison helpers only (mpmath); no oracle code
    with open(path) as f:
        fx = json.load(f)
    assert fx["precision"] == precision and fx["N"] == solver.sdp.N and fx["J"] == solver.sdp.J
    # Gate tolerance: the SURVEY §8d bar 2^-(p/2), tightened per fixture to the worst difference this path has
    # been MEASURED at on the device plus a 32-bit margin (tests/golden/synthetic/gate_thresholds.json), so that a
    # regression that loses hundreds of bits but stays inside 2^-(p/2) is still refused a value.
    tol_bits = precision // 2
    thr_path = os.path.join(os.path.dirname(path), "gate_thresholds.json")
    early = None          # (n, bits): a tighter bar for the first n iterations (the ones the clock sees), "<fixture>#first<n>"
    pinned = None         # (measured worst per iteration, margin): "<fixture>#measured_by_iteration" -- every iteration is held to
                          # what this path was measured at on the device plus a few bits, so that a loss of bits in the LATE
                          # iterations (where the one floor 2^-(p/2) leaves tens of bits of slack) is noticed too (round-5 advisor)
    if os.path.exists(thr_path):
        with open(thr_path) as f:
            thr = json.load(f)
        tol_bits = max(tol_bits, int(thr.get(os.path.basename(path), tol_bits)))
        for key, bits in thr.items():
            if key.startswith(os.path.basename(path) + "#first"):
                early = (int(key.split("#first")[1]), int(bits))
        if os.path.basename(path) + "#measured_by_iteration" in thr:
            pinned = (thr[os.path.basename(path) + "#measured_by_iteration"], float(thr.get("#drift_margin_bits", 12)))
    worst, bad_all, by_iteration, relaxed = float("-inf"), [], [], 0
    for rec in fx["iterations"]:
        if solver.iterate():
            bad_all.append((rec["iteration"], "terminated: " + solver.terminate_reason))
            break
        # the bar of this iteration: tol_bits, relaxed to cond 2^-(p-16) only where the iteration's own condition numbers
        # pass 2^(p/2-16) (parity.conditioned_tol_bits: the last iterations of a run that converges to optimality)
        tol_it = parity.conditioned_tol_bits(rec, precision, tol_bits)
        relaxed += tol_it < tol_bits
        bad, w = parity.compare_iteration(solver.scalars(), rec, tol_bits=tol_it)
        worst = max(worst, w)
        by_iteration.append(round(w, 1))
        if early and rec["iteration"] <= early[0] and w > -early[1] and not bad:
            bad_all.append((rec["iteration"], [("worst field of an early iteration", w, -early[1])]))
        if pinned and rec["iteration"] <= len(pinned[0]) and w > pinned[0][rec["iteration"] - 1] + pinned[1] and not bad:
            bad_all.append((rec["iteration"], [("drift against the measured trajectory", w, pinned[0][rec["iteration"] - 1] + pinned[1])]))
        if bad:
            bad_all.append((rec["iteration"], bad))
    terminated = None
    if "terminate_reason" in fx and not bad_all:
        # the fixture follows the oracle until SDP_Solver::run stops: same iteration, same reason, same objectives
        if not solver.iterate():
            bad_all.append((fx["terminated_in_iteration"], "did not terminate; the oracle stopped with: " + fx["terminate_reason"]))
        elif solver.terminate_reason != fx["terminate_reason"]:
            bad_all.append((fx["terminated_in_iteration"], f"terminate reason {solver.terminate_reason!r} != {fx['terminate_reason']!r}"))
        else:
            terminated = {"iteration": fx["terminated_in_iteration"], "reason": solver.terminate_reason}
            for key in ("primalObjective", "dualObjective"):
                w = parity.log2_rel(solver.scalar(key), fx[key])
                worst = max(worst, w)
                if w > -parity.conditioned_tol_bits(fx["iterations"][-1], precision, tol_bits):
                    bad_all.append((fx["terminated_in_iteration"], [(key, w)]))
    return {"fixture": os.path.relpath(path, ROOT), "iterations": len(fx["iterations"]), "tolerance_log2_rel": -tol_bits,
            "worst_log2_rel": worst, "passed": not bad_all, "violations": bad_all[:4], "followed_to_termination": terminated,
            "worst_log2_rel_by_iteration": by_iteration,
            "iterations_on_the_conditioned_bar": relaxed,  # cond 2^-(p-16) where cond > 2^(p/2-16); 0: 2^-tol everywhere
            "pinned_to_measured_trajectory": ({"margin_bits": pinned[1], "iterations": len(pinned[0])} if pinned else None),
            "early_iterations_bar": ({"first": early[0], "tolerance_log2_rel": -early[1],
                                      "worst_log2_rel": max(by_iteration[:early[0]]) if by_iteration else None} if early else None),
            "fields": "mu P-obj D-obj gap P-err p-err D-err R-err P-step D-step beta Q_cond_number max_block_cond_number"}


def syrk_source_digest() -> str:
    """Digest of the source region of the dominant kernel (kernels.hpp from the fixed-point image
    to k_restore_Q): a committed PMC traffic figure is only reported for the build it measured."""
    import hashlib
    with open(os.path.join(ROOT, "sdpb_amd", "csrc", "kernels.hpp")) as f:
        txt = f.read()
    a, b = txt.find("// Q = P^T P in fixed point"), txt.find("k_restore_Q(")
    return hashlib.sha1(txt[a:b].encode()).hexdigest()[:16]


def _self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one
    rank per GPU (the driver may also launch the ranks itself; then WORLD_SIZE is set and this
    is skipped)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def dry_run(args, lib, rank, world):
    import torch
    import torch.distributed as dist
    from sdpb_amd import synthetic
    from sdpb_amd.solver import SDPSolver
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synthetic.config(args.workload, args.scale)
    sdp, source = synthetic.lazy(cfg)
    solver = SDPSolver(sdp, cfg["precision"], rank=rank, world_size=world, upload_all_blocks=False, block_source=source,
                       lib_path=lib)
    if world > 1:
        from sdpb_amd.distributed import make_collectives
        solver.set_collectives(*make_collectives(torch.device("cpu")))
    for _ in range(args.warmup + args.steps):
        assert not solver.iterate(), solver.terminate_reason
    obj = solver.scalar("P-obj")
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, obj)
        assert len(set(objs)) == 1, "ranks diverged"
        dist.barrier()
    if rank == 0:
        emit({"metric": "interior-point iterations/sec at --precision 512", "value": None, "unit": "iterations/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "DRY_RUN_NOT_A_MEASUREMENT": True,
              "config": {"workload": f"{args.workload} x{args.scale}: J={sdp.J}, N={sdp.N}", "exchange": solver.comm_name},
              "P-obj": obj})
    solver.close()
    if world > 1:
        dist.destroy_process_group()


class Watchdog:
    """Turns a hung collective into a diagnostic line instead of a silent timeout (multi-rank runs only).
    A daemon thread polls the solver's lock-free progress record (sdpb_hip_progress: iteration, synchronisation
    points, collectives enqueued, their sequence hash, the last collective, the transport's asynchronous error);
    if nothing moves for `limit` seconds it prints ONE JSON line tagged WATCHDOG with this rank's record and phase
    and ends the process with exit code 124.  Every rank has its own, so the lines together show who waits where."""

    def __init__(self, rank, world, limit):
        import threading
        self.rank, self.world, self.limit = rank, world, limit
        self.phase, self.solver, self.last, self.t_last, self.armed = "start", None, None, time.time(), limit > 0
        self._t = threading.Thread(target=self._run, daemon=True)
        if self.armed:
            self._t.start()

    def enter(self, phase, solver=None):
        self.phase, self.t_last = phase, time.time()
        if solver is not None:
            self.solver = solver

    def disarm(self):
        self.armed = False

    def _run(self):
        while self.armed:
            time.sleep(2.0)
            rec = None
            try:
                rec = self.solver.progress() if self.solver is not None else None
            except Exception:
                pass
            key = (self.phase, json.dumps(rec, sort_keys=True))
            if key != self.last:
                self.last, self.t_last = key, time.time()
            elif self.armed and time.time() - self.t_last > self.limit:
                emit({"metric": "interior-point iterations/sec", "value": None, "WATCHDOG": {
                    "rank": self.rank, "world": self.world, "phase": self.phase, "no_progress_for_s": round(time.time() - self.t_last, 1),
                    "progress": rec, "note": "no iteration, synchronisation point or collective completed on this rank within the "
                                             "limit: a collective is waiting for a rank that never enqueued its part (compare "
                                             "`collectives` and `sequence_hash` across the ranks' lines), or the transport hangs"}})
                os._exit(124)


_JSON_FD = None


def claim_stdout():
    """Keep stdout for the ONE JSON line: everything else that writes to fd 1 in this process and its children (RCCL's
    version banner, gloo's connection chatter, rocm-smi warnings) is sent to stderr from here on."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(record):
    line = (json.dumps(record) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def load_workload(args):
    """-> (label, sdp, block_source or None, precision, params, gate_records or None, gate_fixture)
    `C3`, `C4`, `C5slice`, `C5` (full: one 288-GB GPU holds it since round 6), `C4f` (C4's shape, strictly feasible: ends with
    "found primal-dual optimal solution") (+ --scale): the synthetic SDPs of SURVEY.md §8d with their committed oracle fixtures;
    `golden:<case>`: one of the reference's own end-to-end SDPs (tests/golden/<case>/sdp, e.g. golden:singlet_cT =
    test/data/end-to-end_tests/SingletScalar_cT_test_nmax6/primal_dual_optimal — the only SDP the reference publishes
    a speed for: BASELINE.md §1), run with the parameter values of the reference run (tests/golden/exact_params.json)
    and gated on the reference's golden trace at its own tolerance 2^-99."""
    from sdpb_amd import synthetic
    if args.workload.startswith("golden:"):
        from tests import parity   # data loaders + comparison helpers only
        name = args.workload.split(":", 1)[1]
        sdp, meta, iters, out = parity.load_case(name)
        with open(os.path.join(parity.GOLDEN, "exact_params.json")) as f:
            params = json.load(f)[name]
        return (f"golden:{name}: the reference's own end-to-end SDP ({meta['reference_dir']}), J={sdp.J}, N={sdp.N}, "
                f"P_tot={sdp.P_total}, --precision {meta['precision']}", sdp, None, meta["precision"], params,
                {"iterations": iters, "out": out, "tol_bits": 99}, f"tests/golden/{name}/iterations.json")
    cfg = synthetic.config(args.workload, args.scale)
    sdp, source = synthetic.lazy(cfg)
    kind = "3d-Ising mixed-correlator" if args.workload == "C4" else "stress" if args.workload.startswith("C5") else "bootstrap-shaped"
    label = (f"{args.workload}: synthetic {kind} SDP (SURVEY.md §8d), J={sdp.J}, N={sdp.N}, P_tot={sdp.P_total}, "
             f"--precision {cfg['precision']}" + ("" if args.scale == 1.0 else f" [scaled x{args.scale}]"))
    return label, sdp, source, cfg["precision"], None, None, None


def golden_gate(solver, recs, fixture):
    """The reference's golden trace as the gate: every numeric field of every iteration at the reference's own
    2^-99 (end-to-end.test.cxx:27), then the terminate reason and the final objectives of out.txt."""
    from tests import parity
    worst, bad_all = float("-inf"), []
    for rec in recs["iterations"]:
        if solver.iterate():
            bad_all.append((rec["iteration"], "terminated: " + solver.terminate_reason))
            break
        bad, w = parity.compare_iteration(solver.scalars(), rec, tol_bits=recs["tol_bits"])
        worst = max(worst, w)
        if bad:
            bad_all.append((rec["iteration"], bad))
    terminated = None
    if not bad_all:
        if not solver.iterate() or solver.terminate_reason != recs["out"]["terminateReason"]:
            bad_all.append((len(recs["iterations"]) + 1, "terminate reason: " + solver.terminate_reason))
        else:
            terminated = {"iteration": len(recs["iterations"]) + 1, "reason": solver.terminate_reason}
            for key in ("primalObjective", "dualObjective"):
                w = parity.log2_rel(solver.scalar(key), recs["out"][key])
                worst = max(worst, w)
                if w > -recs["tol_bits"]:
                    bad_all.append((key, w))
    return {"fixture": fixture, "iterations": len(recs["iterations"]), "tolerance_log2_rel": -recs["tol_bits"],
            "worst_log2_rel": worst, "passed": not bad_all, "violations": bad_all[:4], "followed_to_termination": terminated,
            "fields": "mu P-obj D-obj gap P-err p-err D-err R-err P-step D-step beta Q_cond_number max_block_cond_number"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SDPB_BENCH_WORKLOAD", "C4"),
                    help="C3 | C4 (default: the config the metric is quoted on) | C5slice | golden:<case> (e.g. golden:singlet_cT)")
    ap.add_argument("--scale", type=float, default=float(os.environ.get("SDPB_BENCH_SCALE", "1.0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true",
                    help="CPU baseline at FULL size (one steady-state iteration of the port, minutes) instead of the bounded sample")
    ap.add_argument("--exchange", default=os.environ.get("SDPB_BENCH_EXCHANGE", "auto"), choices=["auto", "rccl", "callbacks"],
                    help="world > 1: auto = the in-library RCCL communicator if its pre-flight passes, else torch.distributed "
                         "(backend nccl = RCCL) behind the C-ABI callbacks; rccl / callbacks force one")
    ap.add_argument("--share-one-gpu", action="store_true", default=os.environ.get("SDPB_BENCH_SHARE_ONE_GPU") == "1",
                    help="all ranks of --gpus N use device 0 and RCCL connects them over its socket transport "
                         "(sdpb_amd/rccl_preflight.one_gpu_env): executes the multi-rank exchange on a 1-GPU box; the "
                         "line is flagged NOT_A_SCALING_MEASUREMENT")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("SDPB_BENCH_WATCHDOG_S", "300")),
                    help="world > 1: seconds without progress after which a rank prints a WATCHDOG line and exits 124 (0 = off)")
    ap.add_argument("--lib", default=None, help="developer aid: another gfx950 build of libsdpb_hip.so (A/B of kernel variants)")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="developer aid, NOT a measurement: run rank 0 of an N-rank job on one GPU with the "
                         "other ranks' contributions faked as copies of its own (per-rank timing without xGMI)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args))

    import torch
    import torch.distributed as dist
    from sdpb_amd import workmodel
    from sdpb_amd.solver import SDPSolver

    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = bool(args.share_one_gpu and world > 1)
    if share:
        from sdpb_amd import rccl_preflight
        os.environ.update(rccl_preflight.one_gpu_env(rank))
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    sim = args.simulate_world if world == 1 else 0
    # Launch-plumbing dry run for the CPU test suite (tests/test_bench_launch.py): the ranks, the
    # rendezvous, the exchange callbacks and the one JSON line, on the CPU EMULATION build over gloo.
    # Its output is flagged DRY_RUN_NOT_A_MEASUREMENT and carries no timing claim.
    dry = os.environ.get("SDPB_BENCH_DRYRUN_LIB")
    if dry:
        return dry_run(args, dry, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dog = Watchdog(rank, world, args.watchdog if world > 1 else 0)
    if world > 1:
        # Control plane on gloo (host memory: the 128-byte id, the timing barrier, the max over ranks, the per-rank
        # records) so that the ONLY RCCL communicator on a device is the one that carries the exchange — the
        # library's own, or torch's when the run falls back to the callbacks.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dog.enter("rendezvous (gloo)")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    label, sdp, source, precision, params, golden, golden_fixture = load_workload(args)
    t_setup = time.time()
    if sim:   # the faked exchange cannot supply the panels other ranks would factor
        os.environ["SDPB_HIP_DIST_CHOLQ"] = "0"
    # Planned HBM footprint of this rank, BEFORE anything is allocated or uploaded (sdpb_amd/workmodel.py: the arrays of
    # Solver::build_layout from the shapes alone): a plan that cannot fit fails here, on every rank, with the table --
    # not minutes later inside an allocation (full C5 on fewer than four GPUs, say).
    from sdpb_amd.solver import plan_blocks
    owners = plan_blocks(sdp.dims, sdp.num_points, sdp.N, sim or world, lib_path=args.lib)
    # the device's own figures, so that the plan follows the library's rule for the windows of the Q stage (round-5 advisor)
    free_b, total_b = torch.cuda.mem_get_info(device)
    planned = workmodel.planned_footprint(sdp.dims, sdp.num_points, sdp.N, precision, owners, rank,
                                          sim or world, dist_cholq=os.environ.get("SDPB_HIP_DIST_CHOLQ") == "1",
                                          num_cus=torch.cuda.get_device_properties(device).multi_processor_count,
                                          hbm_bytes=total_b, free_bytes=free_b)
    _not_bytes = ("total", "rows", "owned_blocks", "image_chunks", "image_rows_per_chunk")
    print(f"[bench rank {rank}/{world}] planned HBM footprint {planned['total'] / 1e9:.2f} GB of {total_b / 1e9:.0f} GB "
          f"({free_b / 1e9:.0f} GB free): " + ", ".join(f"{k} {v / 1e9:.2f}" for k, v in planned.items() if k not in _not_bytes) +
          f"; {planned['owned_blocks']} blocks, {planned['rows']} rows, the image of P' in {planned['image_chunks']} row window(s) of "
          f"{planned['image_rows_per_chunk']} rows", file=sys.stderr, flush=True)
    if planned["total"] > free_b and not share:
        raise SystemExit(f"bench.py rank {rank}: the planned footprint {planned['total'] / 1e9:.1f} GB exceeds the {free_b / 1e9:.1f} GB free on "
                         f"device {local_rank}: use more GPUs (--gpus) or a smaller workload")
    dog.enter("upload")
    solver = SDPSolver(sdp, precision, params, device=local_rank, rank=rank, world_size=sim or world, upload_all_blocks=False,
                       block_source=source, lib_path=args.lib)
    exchange_record = None
    if world > 1:
        def all_ok(flag):
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def exchange_id(hexid):
            box = [hexid]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        use_rccl = args.exchange != "callbacks"
        pre = None
        if args.exchange == "auto":
            # the in-library communicator has to prove itself with THIS many ranks in a child process first
            from sdpb_amd import rccl_preflight
            dog.enter("rccl pre-flight (child processes)")
            pre = rccl_preflight.run(rank, world, local_rank, exchange_id, timeout=min(120.0, max(30.0, args.watchdog / 2 or 120.0)),
                                     lib_path=args.lib)   # the child inherits os.environ (one_gpu_env when the ranks share a GPU)
            use_rccl = all_ok(pre["ok"])
        if use_rccl:
            dog.enter("sdpb_hip_rccl_init")
            ok, err = True, None
            try:
                uid = exchange_id(solver.rccl_unique_id().hex() if rank == 0 else None)
                solver.rccl_init(bytes.fromhex(uid))
            except Exception as e:
                ok, err = False, f"{type(e).__name__}: {e}"
                print(f"[bench rank {rank}] in-library RCCL init failed ({err})", flush=True)
            use_rccl = all_ok(ok and solver.comm_name == "rccl")
            if not use_rccl and args.exchange == "rccl":
                raise SystemExit("bench.py --exchange rccl: sdpb_hip_rccl_init failed on some rank")
        if not use_rccl:
            # torch.distributed's own RCCL communicator behind the C-ABI callbacks (zero-copy on device pointers)
            from sdpb_amd.distributed import make_collectives
            dog.enter("callbacks: torch.distributed nccl group")
            group = dist.new_group(backend="nccl", device_id=device)
            solver.set_collectives(*make_collectives(device, group))
        exchange_record = {"requested": args.exchange, "preflight": pre, "transport": solver.comm_name}
    elif sim:
        from sdpb_amd.distributed import tensor_from_pointer

        def fake_allreduce(ptr, count):
            tensor_from_pointer(ptr, count * 8, device).view(torch.int64).mul_(sim)
            torch.cuda.current_stream(device).synchronize()
            return 0

        def fake_allgather(send, recv, nbytes):
            r = tensor_from_pointer(recv, nbytes * sim, device).view(sim, nbytes)
            r.copy_(tensor_from_pointer(send, nbytes, device).unsqueeze(0).expand(sim, nbytes))
            torch.cuda.current_stream(device).synchronize()
            return 0

        solver.set_collectives(fake_allreduce, fake_allgather)
    t_setup = time.time() - t_setup

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # parity gate BEFORE anything is timed; then back to the initial point, so the timed
    # iterations are W+1 .. W+K of the run, as in every earlier round
    dog.enter("parity gate", solver)
    if sim:
        gate = None
    elif golden:
        gate = golden_gate(solver, golden, golden_fixture)
    else:
        gate = parity_gate(solver, args.workload, args.scale, precision)
    if gate and gate["passed"] is False:
        if rank == 0:
            emit({"metric": f"interior-point iterations/sec at --precision {precision}", "value": None,
                  "PARITY_GATE_FAILED": True, "parity_gate": gate})
        raise SystemExit("bench.py: the device iteration does not reproduce the oracle fixture; no value reported")
    solver.reset()

    # The synthetic SDP is not feasible: SDP_Solver::run ends it with "maxComplementarity exceeded" after 49
    # iterations (profiles/c4_iterations_to_termination.py).  A run longer than that starts over from the initial
    # point (a memset); the iterate() call that only detects the termination does no step and is not counted.
    restarts = [0]

    def steps(k):
        done = 0
        while done < k:
            if solver.iterate():
                solver.reset()
                restarts[0] += 1
                assert restarts[0] < 1000, solver.terminate_reason
                continue
            done += 1

    dog.enter("warm-up")
    steps(args.warmup)
    timers0 = solver.timers()
    syncs0 = solver.host_syncs
    dog.enter("timed region")
    barrier()
    restarts[0] = 0
    t0 = time.perf_counter()
    steps(args.steps)
    torch.cuda.synchronize(device)
    dt_local = time.perf_counter() - t0
    barrier()
    dt = dt_local
    if world > 1:
        t = torch.tensor([dt_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dog.enter("records")
    timers1 = solver.timers()
    syncs1 = solver.host_syncs
    # proof that N ranks exchanged: what every rank owns and what it handed to the transport
    mine = {k[5:]: (timers1[k] - timers0.get(k, 0)) if k.endswith(("_calls", "_bytes")) or k == "comm.collectives" else timers1[k]
            for k in timers1 if k.startswith("comm.")}
    mine["transport"] = solver.comm_name
    mine["launches"] = timers1.get("launches", 0) - timers0.get("launches", 0)
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # per-stage breakdown: one extra, UNTIMED iteration with the stage timers on (they synchronise
    # the stream at every stage boundary, so they are off inside the timed region)
    dog.enter("profiled iteration")
    solver.set_profiling(True)
    if solver.iteration >= 40 or (golden and solver.iteration >= len(golden["iterations"]) - 2):   # keep clear of the termination
        solver.reset()
        solver.iterate()
    tp0 = solver.timers()
    assert not solver.iterate(), solver.terminate_reason
    tp1 = solver.timers()
    solver.set_profiling(False)
    dog.disarm()

    try:
        mem_plan = solver.memory_plan()
    except AttributeError:   # --lib with a build older than round 5 (A/B timing only)
        mem_plan = {"syrk": None, "bytes": {}, "device": None}
    if rank == 0:
        ms_per_step = 1000.0 * dt / args.steps
        value = args.steps / dt
        # dominant kernel: the fixed-point syrk Q' = P'^T P' (HIP events on the launch stream)
        k_ms = timers1["kernel.k_syrk_fx.ms"] - timers0["kernel.k_syrk_fx.ms"]
        k_n = timers1["kernel.k_syrk_fx.launches"] - timers0["kernel.k_syrk_fx.launches"]
        k_avg_s = (k_ms / max(k_n, 1)) / 1000.0
        k_bytes = timers1["kernel.k_syrk_fx.algorithmic_bytes"]
        k_macs = timers1["kernel.k_syrk_fx.limb_macs"]
        achieved = k_bytes / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        # which instantiation that is: two Karatsuba levels keep 7 spare bits in the image, one level 3
        fb = solver.fx_frac_bits
        lazy = bool(timers1.get("kernel.k_syrk_fx.toom5k_lazy_carries", 0))   # Toom-5 x Karatsuba on 28-bit limbs: multiply-adds without carry instructions
        mac_peak = LIMB_MAD_PEAK if lazy else LIMB_MAC_PEAK
        fx = 16 if lazy else next(f for f in range(solver.limbs - 2, solver.limbs + 1) if 32 * f - fb in (25, 17, 7, 3))   # limbs of the image (kernels.hpp: fx_limbs)
        rbg = 16 if fx >= 32 else 32   # rows per pass (solver.hpp: SDPB_SYRK2_RBG)
        k_name = (f"k_syrk_fx3<{fx},{rbg}> in lazy-carry mode (Toom-5 x Karatsuba on 28-bit limbs: 27 products of 2 x 2 limbs, one v_mad_u64_u32 per "
                  "limb pair; +k_syrk3_sum_splits +k_syrk5_finish)" if lazy else
                  f"k_syrk_fx3<{fx},{rbg}> (Toom-4 x Karatsuba, +k_syrk4_finish)" if 32 * fx - 25 == fb else
                  f"k_syrk_fx2<{fx},{rbg},toom4> (+k_syrk4_finish)" if 32 * fx - 17 == fb else
                  f"k_syrk_fx2<{fx},{rbg}> (+k_syrk_reduce)" if 32 * fx - 7 == fb else f"k_syrk_fx<{fx}> (+k_syrk_reduce)")
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_k_syrk_fx.json")
        if os.path.exists(pmc) and args.workload == "C4" and args.scale == 1.0 and world == 1:
            with open(pmc) as f:   # rocprofv3 --pmc cannot run inside this process: committed measurement of this command
                rec = json.load(f)
            if rec.get("kernel_source_digest") == syrk_source_digest():
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/pmc_k_syrk_fx.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the "
                                  f"build with kernel source digest {rec['kernel_source_digest']}, FETCH_SIZE x2 per MI355X_MICROARCH.md)")
            else:
                traffic_source = ("dropped: profiles/pmc_k_syrk_fx.json was measured on kernel source digest "
                                  f"{rec.get('kernel_source_digest')}, this build is {syrk_source_digest()}")
        skip = ("kernel.", "host_syncs", "iterations", "comm.", "launches")
        stages = {k: round(tp1[k] - tp0.get(k, 0.0), 3) for k in tp1 if not k.startswith(skip)}
        nl = solver.limbs
        from sdpb_amd.solver import copy_bandwidth_gbs
        copy_gbs = copy_bandwidth_gbs(1 << 30, 5, lib_path=args.lib)
        hashes = [r.get("sequence_hash") for r in per_rank]
        out = {
            "metric": f"interior-point iterations/sec at --precision {precision}",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": f"mw{32 * nl} (fixed-width multi-word float, {nl}x32-bit limbs; Q syrk: exact integer product of a fixed-point image with {solver.fx_frac_bits} fraction bits)",
            "data": "reference end-to-end test SDP" if golden else "synthetic",
            "config": {"workload": label,
                       "parallelism": f"blocks sharded over {world} GPU(s); Q' summed by integer all-reduce",
                       "exchange": solver.comm_name},
            "host_syncs_per_step": (syncs1 - syncs0) / args.steps,
            "kernel_launches_per_step": mine["launches"] / args.steps,
            "restarts_in_timed_region": restarts[0],   # > 0 only when warmup + steps exceeds the iterations the SDP runs
            "parity_gate": gate,
            # rccl_ranks: size of the RCCL communicator as ncclCommCount reports it on every rank (min over ranks;
            # 1 for a single GPU, 0 if the exchange ran on callbacks); owned_blocks/rows: the block shard of each rank
            "rccl_ranks": (min(int(r["ranks"]) for r in per_rank) if all(r["transport"] == "rccl" for r in per_rank)
                           else (1 if world == 1 else 0)),
            "exchange_setup": exchange_record,
            **({"ranks_share_one_gpu": "NOT_A_SCALING_MEASUREMENT: all ranks ran on device 0 and RCCL used its socket "
                                       "transport over loopback; this line shows the multi-rank exchange executing, "
                                       "nothing about xGMI or speed-up"} if share else {}),
            # every rank hashes the (kind, bytes, root) of each collective it enqueues; the hashes are compared on the
            # device at every synchronisation point (a mismatch ends all ranks with the same error) and reported here
            "collective_sequence": {"hash": hashes[0], "identical_on_all_ranks": len(set(hashes)) == 1,
                                    "collectives_per_step": [r.get("collectives", 0) / args.steps for r in per_rank]},
            "exchange_per_rank": {"transport": [r["transport"] for r in per_rank],
                                  "owned_blocks": [int(r["owned_blocks"]) for r in per_rank],
                                  "owned_rows": [int(r["owned_rows"]) for r in per_rank],
                                  "allreduce_calls_per_step": [r["allreduce_calls"] / args.steps for r in per_rank],
                                  "allreduce_MB_per_step": [round(r["allreduce_bytes"] / args.steps / 1e6, 3) for r in per_rank],
                                  "allgather_calls_per_step": [r["allgather_calls"] / args.steps for r in per_rank],
                                  "allgather_MB_per_step": [round(r["allgather_bytes"] / args.steps / 1e6, 3) for r in per_rank],
                                  "cholesky_Q": per_rank[0]["cholesky_Q"],
                                  "broadcast_calls_per_step": [r["broadcast_calls"] / args.steps for r in per_rank],
                                  "broadcast_MB_per_step": [round(r["broadcast_bytes"] / args.steps / 1e6, 3) for r in per_rank]},
            "roofline": {"bound": "hbm", "kernel": k_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "measured_copy_peak": copy_gbs,
                         "launch_ms": 1000.0 * k_avg_s, "algorithmic_bytes_per_launch": k_bytes,
                         "limb_mac_per_s": k_macs / k_avg_s if k_avg_s > 0 else 0.0,
                         "limb_mac_frac_of_measured_valu_peak": (k_macs / k_avg_s / mac_peak) if k_avg_s > 0 else 0.0,
                         "measured_valu_peak_limb_mac_per_s": mac_peak,
                         "valu_peak_is": ("v_mad_u64_u32 alone (lazy carries)" if lazy else "v_mad_u64_u32 + v_addc_co_u32 pairs"),
                         # how the launch was cut (solver.hpp: syrk_plan): output tiles per chunk under the memory bound, row
                         # splits per tile and their length -- above P_tot = 81 920 rows the "<= 2560 rows per split" rule that
                         # keeps the operand panels in L2 stops holding (32 splits at most) and this shows it
                         "syrk_plan": mem_plan["syrk"],
                         # the input window: rows of P' whose fixed-point image is resident at a time (solver.hpp: q_window)
                         "image_plan": mem_plan.get("image"),
                         # fraction bits of the fixed-point image of P' the exact integer product is formed from, beside the
                         # reference's (Matrix_Normalizer.cxx:174-192 truncates at 2^precision)
                         "q_image_bits": solver.fx_frac_bits, "reference_bits": precision},
            "memory_plan": {"bytes": mem_plan["bytes"], "total_GB": round(sum(mem_plan["bytes"].values()) / 1e9, 2),
                            "planned_before_upload_GB": round(planned["total"] / 1e9, 2), "device": mem_plan["device"]},
            "algorithmic_bytes_per_iteration": workmodel.algorithmic_bytes_per_iteration(
                sdp.dims, sdp.num_points, sdp.N, 4 * (nl + 1)),
            "stage_ms_profiled_iteration": stages,
            "setup_s": t_setup,
        }
        if golden and args.workload == "golden:singlet_cT":
            # the only speed the reference publishes, embedded in its own fixture
            out["reference_published"] = {"iterations_per_s": 3.8, "where": "BASELINE.md §1: test/data/end-to-end_tests/"
                                          "SingletScalar_cT_test_nmax6/primal_dual_optimal/output/out/iterations.1.json (iter_time), "
                                          "6 MPI ranks on the reference authors' CPU", "same_input": True}
        if sim:
            out["SIMULATED_WORLD_NOT_A_MEASUREMENT"] = sim
        if world == 1 and not args.no_cpu_baseline and not golden:
            out["cpu_baseline"] = cpu_baseline(args.workload, precision, full=args.cpu_full)
        emit(out)
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
