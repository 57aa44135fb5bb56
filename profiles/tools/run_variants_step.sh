#!/bin/bash
# developer tool: C4 bench per variant library, printing the step-length stage
set +e
O=gpurun_out/${1:-variants}; mkdir -p $O
export TMPDIR=/tmp
cp sdpb_amd/libsdpb_hip.so /tmp/libsdpb_hip.orig.so
for v in sdpb_amd/_variants/*.so; do
  n=$(basename $v .so)
  cp $v sdpb_amd/libsdpb_hip.so
  for rep in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_$n.json 2>> $O/err.log
  python - "$O/bench_C4_$n.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "stepLength", st.get("stepLength"), "step", st.get("step"), (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
  done
done
cp /tmp/libsdpb_hip.orig.so sdpb_amd/libsdpb_hip.so
