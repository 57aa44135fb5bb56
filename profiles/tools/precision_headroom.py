"""Worst log2 relative difference per field against the live oracle over 5 iterations of 1d-constraints, per compiled width:
how many bits the device path really carries at each width (a parity test passes at precision/2).   GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
from sdpb_amd.solver import SDPSolver
from tests import libs, parity
for precision in (512, 768, 1024, 1280, 1536, 1700, 2048):
    sdp, meta, _, _ = parity.load_case("1d-constraints")
    o = Oracle(sdp, precision, meta["params"], param_prec=64)
    s = SDPSolver(sdp, precision, parity.reference_params(meta["params"], o), lib_path=libs.product_lib())
    worst = {}
    for it in range(5):
        assert not s.iterate() and not o.iterate()
        a, b = s.scalars(), o.scalars()
        for k in a:
            if k == "block_name":
                continue
            try:
                w = parity.log2_rel(a[k], b[k])
            except Exception:
                continue
            worst[k] = max(worst.get(k, float("-inf")), w)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"--precision {precision}: {s.limbs} limbs = {32 * s.limbs} bits; worst fields: " + ", ".join(f"{k} {v:.1f}" for k, v in top), flush=True)
    s.close(); o.close()
