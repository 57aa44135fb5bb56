#!/bin/bash
# round 5, first GPU call: the whole GPU suite with the tile-packed / chunked syrk, the default bench line, and the
# C5 slice with the default memory bound (chunks) against an unbounded buffer
set +e
O=gpurun_out/${1:-r05a}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_C4_default.json 2> $O/bench_C4_default.err; tail -c 600 $O/bench_C4_default.json
timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_default_bound.json 2> $O/bench_C5slice.err
SDPB_HIP_SYRK_PART_BYTES=120000000000 timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_unbounded.json 2>> $O/bench_C5slice.err
python - <<PY
import json
for f in ("bench_C5slice_default_bound","bench_C5slice_unbounded"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel_ms"))
    except Exception as e: print(f, "ERR", e)
PY
