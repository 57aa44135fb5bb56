#!/bin/bash
# Round 4: Q' in two column chunks with Cholesky(Q) chasing it (SDPB_HIP_Q_CHASE): parity, A/B at one rank, simulated ranks
set +e
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity_at_size.py -m gpu -x -q -k "chased or concurrent" > $O/chase_tests.log 2>&1; echo "pytest rc=$?" >> $O/chase_tests.log; tail -5 $O/chase_tests.log
for ch in 0 1; do
  SDPB_HIP_Q_CHASE=$ch timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_chase$ch.json 2>> $O/err.log
  SDPB_HIP_Q_CHASE=$ch timeout 600 python bench.py --simulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world8_chase$ch.json 2>> $O/err.log
  SDPB_HIP_Q_CHASE=$ch timeout 600 python bench.py --simulate-world 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world4_chase$ch.json 2>> $O/err.log
  SDPB_HIP_Q_CHASE=$ch timeout 600 python bench.py --simulate-world 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world2_chase$ch.json 2>> $O/err.log
  SDPB_HIP_Q_CHASE=$ch timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5_chase$ch.json 2>> $O/err.log
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "join", st.get("initializeSchurComplementSolver.Cholesky_Q(join)"), "syrk", st.get("initializeSchurComplementSolver.Q.syrk"), d["roofline"].get("launch_ms"), (d.get("parity_gate") or {}).get("worst_log2_rel"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -5 $O/err.log
