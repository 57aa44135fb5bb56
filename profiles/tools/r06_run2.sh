#!/bin/bash
# Round 6, second GPU call: (1) A/B of the tile-dot P = L^-1 B (sdpb_amd/_variants/*.so, NL = 18 only): C4 bench line with its
# 48-iteration parity gate per variant; (2) SQ counter pass over the second-tier kernels on the float build.
set +e
TAG=${1:-r06b}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash profiles/tools/run_variants.sh $TAG 2>&1 | tee $O/variants.txt
cp sdpb_amd/libsdpb_hip.so /tmp/libsdpb_hip.orig.so
cp sdpb_amd/_variants/notd.so sdpb_amd/libsdpb_hip.so
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$O/bench_C4_under_rocprof.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_sq.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/pmc_inst -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_inst.json 2>> $R/$O/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_$C.json 2>> $R/$O/err.log
done
cd $R
cp /tmp/libsdpb_hip.orig.so sdpb_amd/libsdpb_hip.so
KERN="k_chol_inv_lds\|k_chol_syrk_down\|k_gemm\|k_chol_panel_solve\|k_gemv_n\|k_gemv_t_partial\|k_constraint_weighted_sum\|k_trsm_rlt_panel\|k_trsm_rln_panel\|k_tridiag\|k_schur_rhs2\|k_normalize_fx\|k_syrk_fx3\|k_qsolve_panel3\|k_vec_trsm"
for f in $(find $O/trace -name "*_results.db"); do python profiles/tools/rocpd_stats.py $f --top 70 > $O/kernel_stats_C4.txt; done
for f in $(find $O/pmc_sq -name "*_results.db"); do for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|$KERN"; done > $O/pmc_SQ_second_tier.txt; done
for f in $(find $O/pmc_inst -name "*_results.db"); do for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|$KERN"; done > $O/pmc_SQ_INSTS_second_tier.txt; done
for C in FETCH_SIZE WRITE_SIZE; do
  for f in $(find $O/pmc_$C -name "*_results.db"); do python profiles/summarize_pmc_db.py $f $C > $O/pmc_$C.txt; done
done
rm -rf $O/trace $O/pmc_sq $O/pmc_inst $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -30 $O/kernel_stats_C4.txt; tail -3 $O/err.log
