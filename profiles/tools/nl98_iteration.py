"""Developer probe (round 6): one iteration of a library whose 66-limb slot holds a 98-limb solver
(-DSDPB_NL=98 -Dmake_solver_98=make_solver_66, requested as --precision 2048) against the oracle at 2048 bits, scalars and
every intermediate array: where does the 98-limb width lose its bits?   usage: python profiles/tools/nl98_iteration.py <lib.so>"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle.oracle import Oracle          # noqa: E402
from sdpb_amd.solver import SDPSolver     # noqa: E402
from tests import parity                  # noqa: E402
from tests import test_gpu_parity_at_size as at_size  # noqa: E402

lib = sys.argv[1]
for case in ("1d",):
    sdp, meta, _, _ = parity.load_case(case)
    p = 2048
    o = Oracle(sdp, p, meta["params"], param_prec=64)
    s = SDPSolver(sdp, p, parity.reference_params(meta["params"], o), lib_path=lib)
    print(case, "limbs reported by the library:", s.limbs, flush=True)
    for it in range(1):
        assert not s.iterate() and not o.iterate()
        bad, w = parity.compare_iteration(s.scalars(), o.scalars(), tol_bits=p // 2)
        print(f"  iteration {it + 1}: worst scalar 2^{w:.1f}; below 2^-{p // 2}: {bad}", flush=True)
        rep = at_size._compare_arrays(sdp, s, o, p, range(sdp.J))
        worst = sorted(rep.items(), key=lambda kv: -kv[1])
        print("  arrays, worst first:", ", ".join(f"{k}: {v:.0f}" for k, v in worst), flush=True)
    s.close()
    o.close()
