#!/bin/bash
# Round 4, planning data: rank 0 of a W-rank job on one GPU (--simulate-world, NOT a measurement), the distributed
# Cholesky(Q) with look-ahead against the replicated one at C5slice x0.5 (one rank, and two ranks over RCCL sharing the GPU,
# kernel trace for the overlap).
set +e
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
for w in 2 4 8; do
  timeout 600 python bench.py --simulate-world $w --steps 10 --warmup 3 --no-cpu-baseline > $O/sim_world${w}_rank0_NOT_A_MEASUREMENT.json 2> $O/sim_world$w.err; echo "sim w$w rc=$?"
done
for d in 0 1; do
  SDPB_HIP_DIST_CHOLQ=$d timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5_1rank_distcholq$d.json 2> $O/c5_$d.err; echo "c5 1 rank dist=$d rc=$?"
  SDPB_HIP_DIST_CHOLQ=$d timeout 900 python bench.py --workload C5slice --scale 0.5 --gpus 2 --share-one-gpu --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5_2ranks_one_gpu_distcholq$d.json 2> $O/c5_2r_$d.err; echo "c5 2 ranks dist=$d rc=$?"
done
R=$GRAFT_REPO_ROOT
cd /tmp && PYTHONPATH=$R SDPB_HIP_DIST_CHOLQ=1 timeout 900 rocprofv3 --kernel-trace -d $R/$O/prof_c5_2ranks -- python $R/bench.py --workload C5slice --scale 0.5 --gpus 2 --share-one-gpu --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_c5_2ranks.json 2> $R/$O/prof_c5_2ranks.err
cd $R
find $O/prof_c5_2ranks -name "*.db" | head
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), (d.get("parity_gate") or {}).get("worst_log2_rel"), d["exchange_per_rank"]["cholesky_Q"])
    print("   ", {k:v for k,v in st.items() if v>1.0})
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -n 3 $O/*.err | tail -40
