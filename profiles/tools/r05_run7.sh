#!/bin/bash
# round 5: the other workloads on the build that ships (r05z): C3, the reference's singlet_cT case, C5 slice x0.5, two and
# four in-library RCCL ranks sharing the box's GPU on full C4 (an execution of the exchange, NOT a scaling measurement), rank 0
# of 2 / 4 / 8 simulated ranks (planning aid, NOT a measurement)
set +e
O=gpurun_out/${1:-r05y}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3.json 2>> $O/err.log
timeout 600 python bench.py --workload golden:singlet_cT --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_singlet_cT.json 2>> $O/err.log
timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5.json 2>> $O/err.log
for w in 2 4; do
  timeout 900 python bench.py --gpus $w --share-one-gpu --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C4_share_one_gpu_w$w.json 2>> $O/err.log
done
for w in 2 4 8; do
  timeout 600 python bench.py --simulate-world $w --steps 10 --warmup 3 --no-cpu-baseline > $O/simulated_world${w}_rank0_NOT_A_MEASUREMENT.json 2>> $O/err.log
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    g=d.get("parity_gate") or {}
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "syrk", d["roofline"].get("launch_ms"), "steplen", st.get("stepLength"), "gate", g.get("iterations"), g.get("worst_log2_rel"), g.get("tolerance_log2_rel"), g.get("passed"), "ranks", d.get("rccl_ranks"), d.get("collective_sequence",{}).get("identical_on_all_ranks"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -5 $O/err.log
