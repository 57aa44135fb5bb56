"""sdpb_hip_bench_op("syrk") on a 40 000 x 1000 image at several --precision values (measurement aid)."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdpb_amd.solver import SDPSolver
from tests import parity
sdp, _, _, _ = parity.load_case("1d")
for p in (256, 400, 512, 664, 768, 1024):
    s = SDPSolver(sdp, p)
    ms = s.bench_op("syrk", 40000, 1000, 3)
    print(f"--precision {p:5d}: {s.limbs} limbs, image {s.fx_frac_bits} fraction bits, syrk 40000 x 1000: {ms:8.2f} ms", flush=True)
    s.close()
