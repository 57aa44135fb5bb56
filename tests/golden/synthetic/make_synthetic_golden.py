#!/usr/bin/env python
"""Generate tests/golden/synthetic/<cfg>.json: per-iteration scalars of the parity oracle
(oracle/sdpb_oracle.cpp, GMP mpf, pinned against the reference's golden traces) on the
FULL-SIZE synthetic configurations of BASELINE.json (SURVEY.md §8d generator contract).

The full C4 (J=600, N=1000, P_tot=40000, --precision 512) is the exact workload bench.py
times; its oracle iteration costs minutes of CPU on all cores, too slow for the GPU box's
test run, so the expected scalars are produced once here (build container) and committed as
data.  tests/test_gpu_parity.py::test_gpu_matches_oracle_fixture_at_full_size replays them.

    python tests/golden/synthetic/make_synthetic_golden.py C4 2
    python tests/golden/synthetic/make_synthetic_golden.py C3 4
    python tests/golden/synthetic/make_synthetic_golden.py C3 term            # until SDP_Solver::run stops
    python tests/golden/synthetic/make_synthetic_golden.py C4 term 0.25 C4_x0.25_to_termination
    python tests/golden/synthetic/make_synthetic_golden.py C4f term 0.25 C4f_x0.25_to_termination   # the strictly feasible
                                  # family (sdpb_amd/synthetic.py): 159 iterations to 'found primal-dual optimal solution'

With `term` the oracle runs until its loop ends (run.cxx:380-467: a terminate reason from
compute_feasible_and_termination.cxx or step.cxx:145-153) and the fixture also records
`terminate_reason`, `terminated_in_iteration` (the iteration whose loop body stopped: it has no
record, exactly like iterations.json) and the final `primalObjective`/`dualObjective`.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1]
    to_term = sys.argv[2] == "term"
    iters = 10 ** 6 if to_term else int(sys.argv[2])
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    from oracle.oracle import Oracle
    from sdpb_amd import synthetic
    from tests import parity
    c = synthetic.config(cfg, scale)
    sdp, src = synthetic.lazy(c)
    t0 = time.time()
    o = Oracle(sdp, c["precision"], parity.DEFAULT_PARAMS, param_prec=0, block_source=src)
    print(f"{cfg}: J={sdp.J} N={sdp.N} P_tot={sdp.P_total} p={c['precision']} threads={o.threads} "
          f"setup {time.time() - t0:.0f}s", flush=True)
    recs, secs = [], []
    name = sys.argv[4] if len(sys.argv) > 4 else (cfg if scale == 1.0 else f"{cfg}_x{scale}")
    final = {}

    def write():
        # rewritten after every iteration: a long run that is cut short keeps what it has
        out = {"config": cfg, "scale": scale, "J": sdp.J, "N": sdp.N, "P_tot": sdp.P_total,
               "precision": c["precision"], "seed": c["seed"], "params": parity.DEFAULT_PARAMS,
               "generator": "tests/golden/synthetic/make_synthetic_golden.py (oracle/sdpb_oracle.cpp)",
               "oracle_threads": o.threads, "oracle_seconds_per_iteration": [round(s, 1) for s in secs],
               "iterations": recs}
        out.update(final)
        dest = os.environ.get("SDPB_FIXTURE_DIR", HERE)   # a long run may write elsewhere and be moved in when complete
        tmp = os.path.join(dest, f".{name}.json.tmp")
        with open(tmp, "w") as f:
            json.dump(out, f, indent=1)
        os.replace(tmp, os.path.join(dest, f"{name}.json"))

    # A run of hours survives being stopped: after every iteration the state (x, X, y, Y as the oracle's own mpf records,
    # orc_get_records / orc_set_records: every bit) goes to <dest>/.<name>.state.pkl, and a later call with the same
    # arguments continues from it -- the iterate is a function of that state alone, so the continued run is the run that
    # was stopped (checked on C3 x0.05: identical fixtures).  `resumed_after` in the fixture says where it was picked up.
    import pickle
    dest0 = os.environ.get("SDPB_FIXTURE_DIR", HERE)
    state_path = os.path.join(dest0, f".{name}.state.pkl")
    start = 0
    if os.path.exists(state_path) and os.path.exists(os.path.join(dest0, f"{name}.json")) and os.environ.get("SDPB_FIXTURE_RESUME", "1") == "1":
        with open(os.path.join(dest0, f"{name}.json")) as f:
            prev = json.load(f)
        with open(state_path, "rb") as f:
            st = pickle.load(f)
        if st["iteration"] == len(prev["iterations"]) and prev["config"] == cfg and prev["scale"] == scale:
            recs, secs, start = prev["iterations"], prev["oracle_seconds_per_iteration"], st["iteration"]
            final["resumed_after"] = prev.get("resumed_after", []) + [start]
            o.set_records("y", st["y"])
            for j in range(sdp.J):
                o.set_records("x", st["x"][j], j)
                for b in (0, 1):
                    o.set_records("X", st["X"][j][b], j, b)
                    o.set_records("Y", st["Y"][j][b], j, b)
            print(f"resumed after iteration {start}", flush=True)

    def save_state(done):
        L64 = c["precision"] // 64 + 4
        st = {"iteration": done, "y": o.records("y", limbs64=L64), "x": [o.records("x", j, limbs64=L64) for j in range(sdp.J)],
              "X": [[o.records("X", j, b, limbs64=L64) for b in (0, 1)] for j in range(sdp.J)],
              "Y": [[o.records("Y", j, b, limbs64=L64) for b in (0, 1)] for j in range(sdp.J)]}
        with open(state_path + ".tmp", "wb") as f:
            pickle.dump(st, f, protocol=4)
        os.replace(state_path + ".tmp", state_path)

    for it in range(start, iters):
        t = time.time()
        if o.iterate():
            assert to_term, o.terminate_reason
            final.update({"terminate_reason": o.terminate_reason, "terminated_in_iteration": it + 1,
                          "primalObjective": o.scalar("primalObjective"), "dualObjective": o.scalar("dualObjective")})
            print(f"terminated in iteration {it + 1}: {o.terminate_reason}", flush=True)
            write()
            break
        secs.append(time.time() - t)
        rec = o.scalars()
        rec["iteration"] = it + 1
        recs.append(rec)
        print(f"iteration {it + 1}: {secs[-1]:.0f}s  P-obj={rec['P-obj'][:30]}", flush=True)
        write()
        if to_term:
            save_state(it + 1)
    print("wrote", name)


if __name__ == "__main__":
    main()
