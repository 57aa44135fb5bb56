#!/bin/bash
# one SQ counter pass over the bench command: where the wave cycles of k_syrk_fx3 and k_trsm_rlt_panel go
set +e
O=gpurun_out/${1:-r04sq}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cd /tmp
timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_sq.json 2>> $R/$O/err.log
cd $R
for f in $(find $O/pmc_sq -name "*_results.db"); do
  for c in $C; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|k_syrk_fx3\|k_trsm_rlt_panel\|k_chol_inv_lds\|k_tridiag<" ; done > $O/pmc_SQ_summary.txt
done
rm -rf $O/pmc_sq
cat $O/pmc_SQ_summary.txt; tail -3 $O/err.log
