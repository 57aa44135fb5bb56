"""Half of C5 (J = 4096 blocks of m = 6, N = 2048, --precision 1024: P_tot = 172 032, 232 GB of device arrays + the bounded
partial planes of the syrk: SDPB_HIP_SYRK_PART_BYTES = 8 GiB per rank with four ranks, 16 GiB with two) through the
multi-rank device path with the in-library RCCL communicator: 4 ranks (each holds what a rank of the 8-GPU job holds: a
C5slice) and 2 ranks (two slices each) sharing the one GPU of the box (NCCL_HOSTID per rank, socket transport).  No oracle
exists at this size (the full-size C5slice fixture cost 40 min per iteration on 8 cores); what is checked: the ranks of
a run are bit-identical, the two decompositions agree to 2^-900 in every field of every iteration (they differ in the
rank-order sums only), owners are a partition, collective sequences match.  Full C5 (J = 8192, 400 GB) does not fit
one GPU.     python profiles/tools/c5_half_multirank.py [n_iter] [dist_cholq 0/1]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import parity                                    # noqa: E402
from tests.test_multirank_gpu import _load, check_ranks, run_ranks   # noqa: E402


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dist = len(sys.argv) > 2 and sys.argv[2] == "1"
    case = "C5J4096"
    sdp = _load(case)[0]
    out = {"case": case, "J": sdp.J, "N": sdp.N, "P_tot": sdp.P_total, "precision": 1024, "iterations": n_iter,
           "cholesky_Q": "distributed" if dist else "replicated", "transport": "in-library RCCL, ranks share one GPU (socket transport)"}
    traces = {}
    for world in (4, 2):
        t0 = time.time()
        env = {"SDPB_HIP_SYRK_PART_BYTES": str((32 << 30) // world)}   # the ranks share one GPU: bound what each takes for the syrk
        if dist:
            env["SDPB_HIP_DIST_CHOLQ"] = "1"
        res = run_ranks(world, case, n_iter, timeout=3000, env=env, transport="rccl-one-gpu")
        check_ranks(res, world, sdp.J, "distributed" if dist else "replicated", -(-sdp.N // 32), transport="rccl-one-gpu")
        traces[world] = res[0][2]
        out[f"world{world}"] = {"seconds_incl_setup": round(time.time() - t0, 1), "owned_blocks": [r[3]["comm.owned_blocks"] for r in res],
                                "allreduce_MB_per_iteration": round(res[0][3]["comm.allreduce_bytes"] / n_iter / 1e6, 1),
                                "collectives": res[0][3]["comm.collectives"], "sequence_hash": res[0][3]["comm.sequence_hash"],
                                "seconds_per_iteration_rank0": res[0][3]["seconds_per_iteration"],
                                "memory_plan_bytes_per_rank": [sum(r[3]["memory_plan"]["bytes"].values()) for r in res],
                                "syrk_plan_rank0": res[0][3]["memory_plan"]["syrk"],
                                "device_free_bytes_seen_by_rank0": res[0][3]["memory_plan"]["device"],
                                "P-obj": [r["P-obj"][:48] for r in res[0][2]]}
    worst = float("-inf")
    for it, (a, b) in enumerate(zip(traces[4], traces[2])):
        bad, w = parity.compare_iteration(a, b, tol_bits=900)
        worst = max(worst, w)
        assert not bad, (it + 1, bad)
    out["world4_vs_world2_worst_log2_rel"] = worst
    print(json.dumps(out, indent=1))


if __name__ == "__main__":   # the ranks are spawned: they import this module again
    main()
