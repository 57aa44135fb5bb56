#!/bin/bash
# round 5: where k_tridiag_min spends its time -- diagnostic variant libraries that stop after the fp64 bisection (d1), after
# the 6-limb rung (d2), after the 10-limb rung (d3); the results are wrong on purpose, only the kernel trace is read
set +e
O=gpurun_out/${1:-r05g}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in $R/sdpb_amd/_variants/d*.so; do
  n=$(basename $v .so)
  timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace_$n -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lib $v > $R/$O/bench_trace_$n.json 2>> $R/$O/err.log
  for f in $(find $R/$O/trace_$n -name "*_results.db"); do python $R/profiles/tools/rocpd_stats.py $f --top 70 > $R/$O/kernel_stats_$n.txt; done
  rm -rf $R/$O/trace_$n
  echo $n; grep "k_tridiag" $R/$O/kernel_stats_$n.txt
done
