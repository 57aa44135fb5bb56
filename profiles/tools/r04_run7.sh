#!/bin/bash
# chase tuning: row splits of the two chunk launches (simulated 8 ranks, one rank)
set +e
O=gpurun_out/r04h; mkdir -p $O
for sp in 2 4 8; do
  for ch in 0 1; do
    SDPB_HIP_SYRK_SPLITS=$sp SDPB_HIP_Q_CHASE=$ch timeout 600 python bench.py --simulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world8_chase${ch}_splits$sp.json 2>> $O/err.log
  done
done
for sp in 8 16; do
  SDPB_HIP_SYRK_SPLITS=$sp SDPB_HIP_Q_CHASE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_chase1_splits$sp.json 2>> $O/err.log
done
SDPB_HIP_SYRK_SPLITS=8 SDPB_HIP_Q_CHASE=1 SDPB_HIP_BESIDE_CUS=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_chase1_splits8_nobeside.json 2>> $O/err.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "join", st.get("initializeSchurComplementSolver.Cholesky_Q(join)"), "syrk", st.get("initializeSchurComplementSolver.Q.syrk"), d["roofline"].get("launch_ms"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
