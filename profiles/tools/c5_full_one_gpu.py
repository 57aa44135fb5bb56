"""FULL C5 (BASELINE.json config 5: J = 8192 blocks of m = 6, K = 2, N = 2048, --precision 1024: P_tot = 344 064 rows) on the ONE
288-GB GPU of the box.  What made it fit (round 6): the fixed-point image of P' -- 237 GB for all rows at once -- is built one
INPUT WINDOW of rows at a time in a bounded buffer and every window's exact product is added into Q' (Solver::q_window,
syrk_G_windows; the reference's input_window_split_factor, BigInt_Shared_Memory_Syrk_Context.cxx:70-110,172-186,
bigint_syrk_blas.cxx:239-285).  B and P (2 x 98.7 GB) stay resident.

No oracle can hold this SDP (the J = 1024 slice costs it 40 min per iteration and 45 GB).  What is checked instead:
 (a) block-local arrays of iteration 1 -- L_j = chol(S_j) and P_j = L_j^-1 B_j at the FULL N -- of >= 8 blocks spread over the
     SDP against the LIVE oracle run on a sub-SDP that holds exactly those blocks (plus the first 49, so that its Q is
     positive definite; at the initial point X = Omega_p I, Y = Omega_d I these arrays depend on the block's own data only);
 (b) the same SDP on 2 in-library RCCL ranks sharing the GPU (J = 4096 blocks each): ranks bit-identical, every field of
     every iteration within 2^-900 of the one-rank run (the decompositions differ in the order of the cross-rank sums only);
 (c) the memory plan: image windows, partial planes, everything inside the device.

    python profiles/tools/c5_full_one_gpu.py [n_iter] [J] [skip: a,b]      -> JSON on stdout
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import libs, parity                                         # noqa: E402
from tests.test_multirank_gpu import _load, check_ranks, run_ranks    # noqa: E402


def one_rank(case, n_iter, probe_blocks):
    from sdpb_amd.solver import SDPSolver
    sdp, precision, params, src, _ = _load(case)
    t0 = time.time()
    s = SDPSolver(sdp, precision, params, lib_path=libs.product_lib(), block_source=src)
    setup = time.time() - t0
    plan = s.memory_plan()
    print(f"[c5] {case}: J={sdp.J} N={sdp.N} P_tot={sdp.P_total} uploaded in {setup:.0f} s; "
          f"device arrays {sum(plan['bytes'].values()) / 1e9:.1f} GB, image windows {plan['image']['image_chunks']} x "
          f"{plan['image']['rows_per_chunk']} rows, free {plan['device']['free_bytes'] / 1e9:.1f} GB", file=sys.stderr, flush=True)
    recs, secs, arrays = [], [], {}
    for it in range(n_iter):
        if it == 1:
            s.set_profiling(True)      # stage timers of the second iteration (they synchronise: its time is not quoted)
        t = time.time()
        assert not s.iterate(), s.terminate_reason
        secs.append(round(time.time() - t, 3))
        recs.append(s.scalars())
        print(f"[c5] iteration {it + 1}: {secs[-1]} s  P-obj {recs[-1]['P-obj'][:24]}", file=sys.stderr, flush=True)
        if it == 0:
            for j in probe_blocks:
                arrays[j] = (s.array("L", j), s.array("PT", j))
    timers = s.timers()
    out = {"setup_seconds": round(setup, 1), "seconds_per_iteration": secs, "memory_plan": s.memory_plan(),
           "stage_ms_profiled_iteration": {k: round(v, 1) for k, v in timers.items() if k[0].isalpha() and "." in k and not k.startswith(("comm.", "kernel."))},
           "syrk_events_ms_per_launch": round(timers["kernel.k_syrk_fx.ms"] / max(1, timers["kernel.k_syrk_fx.launches"]), 1),
           "host_syncs": s.host_syncs, "limbs": s.limbs, "fx_frac_bits": s.fx_frac_bits}
    s.close()
    return sdp, precision, params, src, recs, arrays, out


def oracle_blocks(sdp, precision, params, src, probe_blocks):
    """L_j and P_j of the probe blocks from the oracle on a sub-SDP of just those blocks (+ the first 49: P_tot >= N)."""
    from oracle.oracle import Oracle
    from sdpb_amd.sdp_io import SDP
    need = -(-sdp.N // 42) + 1
    sel = list(range(need)) + [j for j in probe_blocks if j >= need]
    sub = SDP(blocks=[], b=sdp.b, constant="0", shape=([sdp.dims[j] for j in sel], [sdp.num_points[j] for j in sel]))
    t0 = time.time()
    o = Oracle(sub, precision, params, param_prec=0, block_source=lambda i: src(sel[i]))
    o.schur_solver_init()
    arrays = {j: (o.array("L", sel.index(j)), o.array("P", sel.index(j))) for j in probe_blocks}
    secs = time.time() - t0
    threads = o.threads
    o.close()
    return arrays, secs, threads, len(sel)


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    J = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    skip = sys.argv[3].split(",") if len(sys.argv) > 3 else []
    case = f"C5J{J}"
    probe = sorted({0, 7, J // 8 + 3, J // 4 + 1, J // 2, (J // 2 + 1001) % J, (3 * J) // 4 + 5, J - 2, J - 1})
    sdp, precision, params, src, recs, dev_arrays, out = one_rank(case, n_iter, probe)
    out = {"case": case, "J": sdp.J, "N": sdp.N, "P_tot": sdp.P_total, "precision": precision, "iterations": n_iter, "one_rank": out,
           "P-obj": [r["P-obj"][:60] for r in recs], "D-obj": [r["D-obj"][:60] for r in recs], "mu": [r["mu"][:40] for r in recs]}
    if "a" not in skip:
        want, secs, threads, nsub = oracle_blocks(sdp, precision, params, src, probe)
        N, rep = sdp.N, {}
        for j in probe:
            P = sdp.num_points[j] * sdp.dims[j] * (sdp.dims[j] + 1) // 2
            low = lambda v: [v[i + c * P] for c in range(P) for i in range(c, P)]
            pt = dev_arrays[j][1]                                                   # N x P column-major on the device
            rep[j] = {"L": round(parity.maxrel(low(dev_arrays[j][0]), low(want[j][0])), 1),
                      "P": round(parity.maxrel([pt[n + q * N] for n in range(N) for q in range(P)], want[j][1]), 1)}
        worst = max(max(v.values()) for v in rep.values())
        out["block_local_arrays_vs_live_oracle"] = {"blocks": probe, "log2_maxrel": rep, "worst": worst, "bar": -(precision - 64),
                                                    "oracle_sub_sdp_blocks": nsub, "oracle_seconds": round(secs, 1), "oracle_threads": threads}
        assert worst <= -(precision - 64), rep
        print(f"[c5] (a) {len(probe)} blocks: L_j, P_j vs the live oracle, worst 2^{worst}", file=sys.stderr, flush=True)
    if "b" not in skip:
        t0 = time.time()
        env = {"SDPB_HIP_SYRK_PART_BYTES": str(10 << 30), "SDPB_HIP_SYRK_IMAGE_BYTES": str(10 << 30)}   # two ranks share the GPU: bound both windows of each
        res = run_ranks(2, case, n_iter, timeout=5000, env=env, transport="rccl-one-gpu")
        check_ranks(res, 2, sdp.J, "replicated", -(-sdp.N // 32), transport="rccl-one-gpu")
        worst = float("-inf")
        for it, (a, b) in enumerate(zip(res[0][2], recs)):
            bad, w = parity.compare_iteration(a, b, tol_bits=900)
            worst = max(worst, w)
            assert not bad, (it + 1, bad)
        out["two_ranks_sharing_the_gpu"] = {
            "seconds_incl_setup": round(time.time() - t0, 1), "owned_blocks": [r[3]["comm.owned_blocks"] for r in res],
            "ranks_bit_identical": True, "vs_one_rank_worst_log2_rel": round(worst, 1), "bar": -900,
            "allreduce_MB_per_iteration": round(res[0][3]["comm.allreduce_bytes"] / n_iter / 1e6, 1),
            "sequence_hash": res[0][3]["comm.sequence_hash"], "seconds_per_iteration_rank0": res[0][3]["seconds_per_iteration"],
            "memory_plan_bytes_per_rank": [sum(r[3]["memory_plan"]["bytes"].values()) for r in res],
            "image_plan_rank0": res[0][3]["memory_plan"]["image"], "syrk_plan_rank0": res[0][3]["memory_plan"]["syrk"],
            "transport": "in-library RCCL, ranks share one GPU (socket transport): NOT an xGMI run"}
        print(f"[c5] (b) two ranks vs one: worst 2^{worst:.1f}", file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":   # the ranks are spawned: they import this module again
    main()
