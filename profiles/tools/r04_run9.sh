#!/bin/bash
# row splits of the syrk launch: more, shorter workgroups against the tail of the last round
set +e
O=gpurun_out/r04k; mkdir -p $O
for sp in 6 8 10 12 14 16; do
  SDPB_HIP_SYRK_SPLITS=$sp timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_splits$sp.json 2>> $O/err.log
done
for sp in 1 2 4 8 16; do
  SDPB_HIP_SYRK_SPLITS=$sp timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5_splits$sp.json 2>> $O/err.log
done
timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5_default.json 2>> $O/err.log
for sp in 1 2 4 8; do
  SDPB_HIP_SYRK_SPLITS=$sp timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3_splits$sp.json 2>> $O/err.log
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "syrk", d["roofline"].get("launch_ms"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
