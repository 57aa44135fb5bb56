#!/bin/bash
# The records of a build that ships: GPU tests, the default bench invocation, kernel trace of the bench, PMC passes
# (FETCH_SIZE / WRITE_SIZE, one counter per run, no trace domains beside them).   usage: bash profiles/tools/r04_final_records.sh <tag>
set +e
TAG=${1:-r04z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_C4_default_invocation.json 2> $O/bench_default.err; echo "default bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2>> $O/err.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/trace -- python $R/bench.py --steps 15 --warmup 2 --no-cpu-baseline > $R/$O/bench_C4_under_rocprof.json 2>> $R/$O/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_$C.json 2>> $R/$O/err.log
done
cd $R
for f in $(find $O/trace -name "*_results.db"); do python profiles/tools/rocpd_stats.py $f --top 70 > $O/kernel_stats_C4.txt; done
for C in FETCH_SIZE WRITE_SIZE; do
  for f in $(find $O/pmc_$C -name "*_results.db"); do python profiles/summarize_pmc_db.py $f $C > $O/pmc_$C.txt; done
done
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -12 $O/kernel_stats_C4.txt; head -6 $O/pmc_FETCH_SIZE.txt; grep -h "k_schur_complement\|k_copy16" $O/pmc_*.txt
python - <<PY
import json
for f in ("bench_C4_default_invocation","bench_C4_20","bench_C4_under_rocprof"):
    try:
        d=[json.loads(l) for l in open("$O/"+f+".json") if l.startswith('{"metric"')][-1]
        print(f, d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "unreadable", e)
PY
