#!/bin/bash
# kernel trace + SQ counters of one variant library (sdpb_amd/_variants/<name>.so) on the C4 bench
set +e
V=${1:-td4}; TAG=${2:-r06c}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp sdpb_amd/libsdpb_hip.so /tmp/libsdpb_hip.orig.so
cp sdpb_amd/_variants/$V.so sdpb_amd/libsdpb_hip.so
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/trace -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/bench_$V.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_sq.json 2>> $R/$O/err.log
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/pmc_inst -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_inst.json 2>> $R/$O/err.log
cd $R
cp /tmp/libsdpb_hip.orig.so sdpb_amd/libsdpb_hip.so
for f in $(find $O/trace -name "*_results.db"); do python profiles/tools/rocpd_stats.py $f --top 30 > $O/kernel_stats_$V.txt
python - $f <<'PY' > $O/trsm_launches_$V.txt
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
sel = [(n, e - s) for n, s, e in rows if "k_trsm_rlt_panel" in n or "k_td_image" in n]
for n, dt in sel[-40:]:
    print(f"{dt / 1e3:10.1f} us  {n[:60]}")
PY
done
for f in $(find $O/pmc_sq -name "*_results.db"); do for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|k_trsm_rlt\|k_td_image\|k_syrk\|k_normalize\|k_fx_colsum"; done > $O/pmc_SQ_$V.txt; done
for f in $(find $O/pmc_inst -name "*_results.db"); do for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES; do python profiles/summarize_pmc_db.py $f $c | grep "^#\|k_trsm_rlt\|k_td_image\|k_syrk\|k_normalize\|k_fx_colsum"; done > $O/pmc_SQ_INSTS_$V.txt; done
rm -rf $O/trace $O/pmc_sq $O/pmc_inst
cat $O/trsm_launches_$V.txt | tail -24; cat $O/pmc_SQ_$V.txt $O/pmc_SQ_INSTS_$V.txt; tail -3 $O/err.log
