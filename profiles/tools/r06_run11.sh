#!/bin/bash
# row splits of the lazy syrk forced at run time (SDPB_HIP_SYRK_SPLITS): time of syrk_G and of the C4 iteration
set +e
O=gpurun_out/${1:-r06u}; mkdir -p $O
export TMPDIR=/tmp
for S in 16 19 24 32; do
SDPB_HIP_SYRK_SPLITS=$S python /dev/stdin <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/syrk_splits.txt
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdpb_amd.solver import SDPSolver
from tests import parity
sdp, _, _, _ = parity.load_case("1d")
s = SDPSolver(sdp, 512)
ms = [s.bench_op("syrk", 40000, 1000, 5) for _ in range(3)]
print("splits", os.environ["SDPB_HIP_SYRK_SPLITS"], "rows per split", -(-40000 // int(os.environ["SDPB_HIP_SYRK_SPLITS"])), " ".join(f"{m:.2f}" for m in ms), "ms (syrk_G: kernel + split sums + finish)", flush=True)
s.close()
PY
done
for S in 16 24 32; do
SDPB_HIP_SYRK_SPLITS=$S timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_splits$S.json 2>> $O/err.log
python - "$O/bench_C4_splits$S.json" <<'PY' | tee -a $O/syrk_splits.txt
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "syrk", d["roofline"].get("launch_ms"), "stage", d["stage_ms_profiled_iteration"].get("initializeSchurComplementSolver.Q.syrk"), (d.get("parity_gate") or {}).get("passed"))
PY
done
