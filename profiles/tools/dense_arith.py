"""Developer probe (round 6): device arithmetic on DENSE mantissas (every limb random) against GMP at 4x the precision --
tests/test_gpu_parity.py::test_device_arithmetic_matches_mpf used 40-digit operands (133 significant bits: the low limbs
of both operands are zero) until round 6.   usage: python profiles/tools/dense_arith.py <lib.so> <precision> [...]"""
import os
import random
import sys

import mpmath

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle.oracle import Oracle          # noqa: E402
from sdpb_amd.solver import SDPSolver     # noqa: E402
from tests import parity                  # noqa: E402

lib = sys.argv[1]
sdp, _, _, _ = parity.load_case("1d")
for p in [int(x) for x in sys.argv[2:]]:
    s = SDPSolver(sdp, p, {}, lib_path=lib)
    o = Oracle(sdp, 4 * p + 256)
    bits = 32 * s.limbs
    digits = int(bits * 0.30103) + 12
    rng = random.Random(p)
    worst = {}
    mpmath.mp.prec = 4 * bits   # tests.parity compares at 1400 bits by default
    for _ in range(12):
        def num():
            return ("-" if rng.random() < 0.5 else "") + "0." + "".join(rng.choice("0123456789") for _ in range(digits)).lstrip("0")[:digits] + \
                "e" + str(rng.randint(-40, 40))
        sa, sb = num(), num()
        for op in ("add", "sub", "mul", "div", "sqrt"):
            xa = sa.lstrip("-") if op == "sqrt" else sa
            got, want = s.op_scalar(op, xa, sb), o.scalar_op(op, xa, sb)
            worst[op] = max(worst.get(op, -1e9), parity.log2_rel(got, want))
    print(f"{os.path.basename(lib)} --precision {p}: {s.limbs} limbs ({bits} bits), dense operands, worst log2 rel error:",
          " ".join(f"{k} {v:.1f}" for k, v in worst.items()), flush=True)
    s.close()
    o.close()
