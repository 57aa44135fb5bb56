#!/bin/bash
# The records of the build that ships at the end of round 6: GPU tests, the default bench invocation, the other workloads,
# kernel trace of the bench, PMC passes (FETCH_SIZE / WRITE_SIZE one counter per run; two SQ passes; no trace domains
# beside counters).   usage: bash profiles/tools/r06_final_records.sh <tag>
set +e
TAG=${1:-r06z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(time timeout 3000 python -m pytest tests -m gpu -q --durations=12) > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_C4_default_invocation.json 2> $O/bench_default.err; echo "default bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2>> $O/err.log
timeout 900 python bench.py --workload C4f --scale 0.25 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4f_x0.25.json 2>> $O/err.log
timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3.json 2>> $O/err.log
timeout 600 python bench.py --workload golden:singlet_cT --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_singlet_cT.json 2>> $O/err.log
timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice.json 2>> $O/err.log
SDPB_HIP_SYRK_IMAGE_BYTES=8000000000 timeout 900 python bench.py --workload C5slice --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5slice_image_windows.json 2>> $O/err.log
SDPB_HIP_TILEDOT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_float_trsm.json 2>> $O/err.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/trace -- python $R/bench.py --steps 15 --warmup 2 --no-cpu-baseline > $R/$O/bench_C4_under_rocprof.json 2>> $R/$O/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_$C.json 2>> $R/$O/err.log
done
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_sq.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/pmc_inst -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_inst.json 2>> $R/$O/err.log
cd $R
for f in $(find $O/trace -name "*_results.db"); do python profiles/tools/rocpd_stats.py $f --top 80 > $O/kernel_stats_C4.txt; done
for C in FETCH_SIZE WRITE_SIZE; do
  for f in $(find $O/pmc_$C -name "*_results.db"); do python profiles/summarize_pmc_db.py $f $C > $O/pmc_$C.txt; done
done
for f in $(find $O/pmc_sq -name "*_results.db"); do for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python profiles/summarize_pmc_db.py $f $c; done > $O/pmc_SQ_raw.txt; done
for f in $(find $O/pmc_inst -name "*_results.db"); do for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python profiles/summarize_pmc_db.py $f $c; done > $O/pmc_SQ_INSTS_raw.txt; done
python profiles/tools/sq_table.py $O/pmc_SQ_raw.txt $O/pmc_SQ_INSTS_raw.txt $O/kernel_stats_C4.txt 17 > $O/pmc_SQ_second_tier.txt
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_inst
head -24 $O/kernel_stats_C4.txt; grep -h "k_syrk\|k_copy16\|k_fx_colsum" $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt; head -30 $O/pmc_SQ_second_tier.txt
python - <<PY
import json
for f in ("bench_C4_default_invocation","bench_C4_20","bench_C4f_x0.25","bench_C3","bench_singlet_cT","bench_C5slice","bench_C5slice_image_windows","bench_C4_float_trsm","bench_C4_under_rocprof"):
    try:
        d=[json.loads(l) for l in open("$O/"+f+".json") if l.startswith('{"metric"')][-1]
        g=d.get("parity_gate") or {}
        print(f, d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"].get("image_plan") or {}).get("image_chunks"), d.get("cpu_baseline",{}).get("value"), d["stage_ms_profiled_iteration"].get("stepLength"), d["stage_ms_profiled_iteration"].get("initializeSchurComplementSolver.Q.solve"), g.get("worst_log2_rel"), g.get("passed"), (g.get("followed_to_termination") or {}).get("reason"))
    except Exception as e: print(f, "unreadable", e)
PY
tail -3 $O/err.log
