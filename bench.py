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
LIMB_MAC_PEAK = 17.0e12        # measured on MI355X: v_mad_u64_u32 + v_addc_co_u32 pairs/s (profiles/r01_ubench.txt)


def cpu_baseline(cfg_name: str, precision: int, seconds_budget: float = 25.0):
    """Time the oracle (oracle/sdpb_oracle.cpp, GMP mpf restatement of the reference iteration,
    1 thread) on a bounded, structurally identical sample of the workload and scale it to the
    metric's unit with the analytic MAC model (sdpb_amd/workmodel.py)."""
    from oracle.oracle import Oracle
    from sdpb_amd import synthetic, workmodel
    full = synthetic.config(cfg_name)
    scale = float(os.environ.get("SDPB_BENCH_CPU_SCALE", "0.04"))
    c = synthetic.config(cfg_name, scale)
    sdp = synthetic.make_sdp(c["dims"], c["num_points"], c["N"], precision, c["seed"])
    o = Oracle(sdp, precision, param_prec=0)
    o.iterate()  # iteration 1 is unrepresentative (X, Y diagonal): run.cxx:442-453
    t0 = time.time()
    n = 0
    while n < 3 and time.time() - t0 < seconds_budget:
        assert not o.iterate()
        n += 1
    dt = (time.time() - t0) / max(n, 1)
    o.close()
    w_full = workmodel.macs_per_iteration(full["dims"], full["num_points"], full["N"])["total"]
    w_s = workmodel.macs_per_iteration(c["dims"], c["num_points"], c["N"])["total"]
    its = 1.0 / (dt * w_full / w_s)
    return {"value": its, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"oracle (GMP mpf, 1 thread) on {cfg_name} scaled x{scale}: J={sdp.J}, N={sdp.N}, "
                      f"P_tot={sdp.P_total}; {n} steady-state iterations at {dt:.2f} s each, scaled by the "
                      f"multi-word MAC model ({w_full:.3g}/{w_s:.3g})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SDPB_BENCH_WORKLOAD", "C4"))
    ap.add_argument("--scale", type=float, default=float(os.environ.get("SDPB_BENCH_SCALE", "1.0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="developer aid, NOT a measurement: run rank 0 of an N-rank job on one GPU with the "
                         "other ranks' contributions faked as copies of its own (per-rank timing without xGMI)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sdpb_amd import synthetic, workmodel
    from sdpb_amd.solver import SDPSolver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    sim = args.simulate_world if world == 1 else 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = synthetic.config(args.workload, args.scale)
    precision = cfg["precision"]
    sdp, source = synthetic.make_lazy(cfg["dims"], cfg["num_points"], cfg["N"], precision, cfg["seed"])
    t_setup = time.time()
    solver = SDPSolver(sdp, precision, device=local_rank, rank=rank, world_size=sim or world, upload_all_blocks=False,
                       block_source=source)
    if world > 1:
        from sdpb_amd.distributed import make_collectives
        solver.set_collectives(*make_collectives(device))
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
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        assert not solver.iterate(), solver.terminate_reason
    timers0 = solver.timers()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assert not solver.iterate(), solver.terminate_reason
    torch.cuda.synchronize(device)
    dt_local = time.perf_counter() - t0
    barrier()
    dt = dt_local
    if world > 1:
        t = torch.tensor([dt_local], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timers1 = solver.timers()

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
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_k_syrk_fx.json")
        if os.path.exists(pmc) and args.workload == "C4" and args.scale == 1.0 and world == 1:
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        stages = {k: round((timers1[k] - timers0.get(k, 0.0)) / args.steps, 3) for k in timers1
                  if not k.startswith("kernel.")}
        nl = solver.limbs
        from sdpb_amd.solver import copy_bandwidth_gbs
        copy_gbs = copy_bandwidth_gbs(1 << 30, 5)
        out = {
            "metric": "interior-point iterations/sec at --precision 512",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": f"mw{32 * nl} (fixed-width multi-word float, {nl}x32-bit limbs; Q syrk in {32 * (nl - 2)}-bit fixed point)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic 3d-Ising mixed-correlator SDP (SURVEY.md §8d), "
                                   f"J={sdp.J}, N={sdp.N}, P_tot={sdp.P_total}, --precision {precision}"
                                   + ("" if args.scale == 1.0 else f" [scaled x{args.scale}]"),
                       "parallelism": f"blocks sharded over {world} GPU(s); Q' summed by integer all-reduce"},
            "roofline": {"bound": "hbm", "kernel": "k_syrk_fx", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_peak": copy_gbs,
                         "launch_ms": 1000.0 * k_avg_s, "algorithmic_bytes_per_launch": k_bytes,
                         "limb_mac_per_s": k_macs / k_avg_s if k_avg_s > 0 else 0.0,
                         "limb_mac_frac_of_measured_valu_peak": (k_macs / k_avg_s / LIMB_MAC_PEAK) if k_avg_s > 0 else 0.0},
            "algorithmic_bytes_per_iteration": workmodel.algorithmic_bytes_per_iteration(
                cfg["dims"], cfg["num_points"], cfg["N"], 4 * (nl + 1)),
            "stage_ms_per_step": stages,
            "setup_s": t_setup,
        }
        if sim:
            out["SIMULATED_WORLD_NOT_A_MEASUREMENT"] = sim
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, precision)
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
