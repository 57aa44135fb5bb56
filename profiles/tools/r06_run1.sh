#!/bin/bash
# Round 6, first GPU call: the input-window split of the Q stage on the device (new tests), the default bench (C4 must not be
# worse), the C5 slice with its image in two windows, and FULL C5 (J = 8192) on the one GPU: profiles/tools/c5_full_one_gpu.py.
set +e
TAG=${1:-r06a}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_at_size.py -m gpu -x -q \
  -k "row_windows or max_shared_memory or in_chunks or ragged or intermediate_arrays or smoke or beyond_one_panel" > $O/gpu_tests_new.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests_new.log; tail -5 $O/gpu_tests_new.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2> $O/bench_C4_20.err; echo "bench C4 rc=$?"
timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice.json 2> $O/bench_C5slice.err; echo "bench C5slice rc=$?"
SDPB_HIP_SYRK_IMAGE_BYTES=64000000000 timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_one_window.json 2>> $O/bench_C5slice.err
# a small C5-shaped run of the script first (J = 512: seconds), then the real thing
timeout 900 python profiles/tools/c5_full_one_gpu.py 2 512 > $O/c5_J512.json 2> $O/c5_J512.err; echo "c5 J512 rc=$?"; tail -4 $O/c5_J512.err
timeout 3300 python profiles/tools/c5_full_one_gpu.py 2 8192 > $O/c5_full.json 2> $O/c5_full.err; echo "c5 full rc=$?"; tail -12 $O/c5_full.err
python - <<PY
import json
for f in ("bench_C4_20","bench_C5slice","bench_C5slice_one_window"):
    try:
        d=[json.loads(l) for l in open("$O/"+f+".json") if l.startswith('{"metric"')][-1]
        print(f, d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], (d["roofline"].get("image_plan") or {}).get("image_chunks"), d["parity_gate"].get("worst_log2_rel"), d["stage_ms_profiled_iteration"].get("stepLength"))
    except Exception as e: print(f, "unreadable", e)
PY
