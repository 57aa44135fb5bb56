#!/bin/bash
# Round 4, RCCL with more than one rank on the 1-GPU box (NCCL_HOSTID per rank, socket transport): the multi-rank tests,
# bench.py --gpus 2/4 --share-one-gpu, kernel trace of two ranks (RCCL device kernels + distributed Cholesky(Q) look-ahead).
set +e
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -rs > $O/multirank_tests.log 2>&1; echo "pytest rc=$?" >> $O/multirank_tests.log
tail -5 $O/multirank_tests.log
for n in 2 4; do
  timeout 900 python bench.py --gpus $n --share-one-gpu --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_share_one_gpu_w$n.json 2> $O/bench_C4_share_one_gpu_w$n.err; echo "bench w$n rc=$?"
done
SDPB_HIP_DIST_CHOLQ=1 timeout 900 python bench.py --gpus 2 --share-one-gpu --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_share_one_gpu_w2_distcholq.json 2> $O/bench_C4_share_one_gpu_w2_distcholq.err; echo "bench w2 dist rc=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && PYTHONPATH=$R SDPB_HIP_DIST_CHOLQ=1 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_2ranks_rccl -- python -m tests.test_multirank_gpu 2 C4x0.25 3 rccl-one-gpu > $R/$O/two_ranks_rccl_dist_cholq.log 2>&1
cd $R
find $O/prof_2ranks_rccl -name "*kernel_stats.csv" | head
for f in $(find $O/prof_2ranks_rccl -name "*kernel_stats.csv"); do head -12 $f; done
tail -4 $O/two_ranks_rccl_dist_cholq.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1], d.get("value"), d.get("ms_per_step"), d.get("rccl_ranks"), d.get("collective_sequence"), d.get("exchange_setup"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -3 $O/*.err
