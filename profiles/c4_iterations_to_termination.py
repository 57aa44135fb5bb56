"""How many iterations the bench workload runs before SDP_Solver::run would stop (bench.py must not run past it)."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdpb_amd import synthetic
from sdpb_amd.solver import SDPSolver
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = synthetic.config(name, 1.0)
sdp, source = synthetic.make_lazy(cfg["dims"], cfg["num_points"], cfg["N"], cfg["precision"], cfg["seed"])
s = SDPSolver(sdp, cfg["precision"], {"maxIterations": 1000}, upload_all_blocks=False, block_source=source)
t = time.time()
while not s.iterate():
    if s.iteration % 20 == 0:
        print(s.iteration, "gap", s.scalar("gap")[:12], "P-err", s.scalar("P-err")[:12], "D-err", s.scalar("D-err")[:12], flush=True)
print(name, "terminated after", s.iteration, "iterations:", s.terminate_reason, f"({time.time() - t:.0f} s)")
