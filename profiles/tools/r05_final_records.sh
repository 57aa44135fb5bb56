#!/bin/bash
# The records of the build that ships at the end of round 5: GPU tests, the default bench invocation, kernel trace of the
# bench, PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per run, no trace domains beside them), the chip-wide product-major
# order of the syrk (time + FETCH_SIZE), an SQ instruction-count pass.   usage: bash profiles/tools/r05_final_records.sh <tag>
set +e
TAG=${1:-r05z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_C4_default_invocation.json 2> $O/bench_default.err; echo "default bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2>> $O/err.log
SDPB_HIP_SYRK_ORDER=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_order1.json 2>> $O/err.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_order0.json 2>> $O/err.log
# cost of chunking: C4 with a 1-GB bound (4 chunks), the C5 slice with 8 GiB (4 chunks) against the default plan (1 chunk)
SDPB_HIP_SYRK_PART_BYTES=1000000000 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_4chunks.json 2>> $O/err.log
timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_1chunk.json 2>> $O/err.log
SDPB_HIP_SYRK_PART_BYTES=8589934592 timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_4chunks.json 2>> $O/err.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/trace -- python $R/bench.py --steps 15 --warmup 2 --no-cpu-baseline > $R/$O/bench_C4_under_rocprof.json 2>> $R/$O/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_$C.json 2>> $R/$O/err.log
done
SDPB_HIP_SYRK_ORDER=1 timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_FETCH_order1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_FETCH_order1.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d $R/$O/pmc_sqinst -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_sqinst.json 2>> $R/$O/err.log
cd $R
for f in $(find $O/trace -name "*_results.db"); do python profiles/tools/rocpd_stats.py $f --top 70 > $O/kernel_stats_C4.txt; done
for C in FETCH_SIZE WRITE_SIZE; do
  for f in $(find $O/pmc_$C -name "*_results.db"); do python profiles/summarize_pmc_db.py $f $C > $O/pmc_$C.txt; done
done
for f in $(find $O/pmc_FETCH_order1 -name "*_results.db"); do python profiles/summarize_pmc_db.py $f FETCH_SIZE > $O/pmc_FETCH_SIZE_order1.txt; done
for f in $(find $O/pmc_sqinst -name "*_results.db"); do for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|k_syrk_fx3\|k_trsm_rlt_panel\|k_tridiag<\|k_chol_inv_lds"; done > $O/pmc_SQ_INSTS_summary.txt; done
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_FETCH_order1 $O/pmc_sqinst
head -14 $O/kernel_stats_C4.txt; grep -h "k_syrk\|k_copy16" $O/pmc_FETCH_SIZE.txt $O/pmc_FETCH_SIZE_order1.txt $O/pmc_WRITE_SIZE.txt; cat $O/pmc_SQ_INSTS_summary.txt
python - <<PY
import json
for f in ("bench_C4_default_invocation","bench_C4_20","bench_C4_order0","bench_C4_order1","bench_C4_4chunks","bench_C5slice_1chunk","bench_C5slice_4chunks","bench_C4_under_rocprof"):
    try:
        d=[json.loads(l) for l in open("$O/"+f+".json") if l.startswith('{"metric"')][-1]
        print(f, d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"].get("syrk_plan") or {}).get("chunks"), d.get("cpu_baseline",{}).get("value"), d["stage_ms_profiled_iteration"].get("stepLength"), d["stage_ms_profiled_iteration"].get("initializeSchurComplementSolver.Q.solve"))
    except Exception as e: print(f, "unreadable", e)
PY
tail -3 $O/err.log
