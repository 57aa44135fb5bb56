#!/bin/bash
# final build of round 4: simulated ranks (planning aid), the other workloads, shared-GPU RCCL bench lines
set +e
O=gpurun_out/${1:-r04n}; mkdir -p $O
export TMPDIR=/tmp
for w in 2 4 8; do
  timeout 600 python bench.py --simulate-world $w --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world${w}.json 2>> $O/err.log
done
timeout 600 python bench.py --workload golden:singlet_cT --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_singlet_cT.json 2>> $O/err.log
timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3.json 2>> $O/err.log
timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5.json 2>> $O/err.log
timeout 1200 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice.json 2>> $O/err.log
for n in 2 4; do
  timeout 900 python bench.py --gpus $n --share-one-gpu --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_share_one_gpu_w$n.json 2>> $O/err.log
done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "join", st.get("initializeSchurComplementSolver.Cholesky_Q(join)"), "steplen", st.get("stepLength"), "syrk", d["roofline"].get("launch_ms"), (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("tolerance_log2_rel"), d.get("rccl_ranks"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -5 $O/err.log
