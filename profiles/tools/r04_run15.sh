#!/bin/bash
# super-block edge of the tile order of k_syrk_fx3 (SDPB_HIP_SYRK_SB, in 32-column tiles): kernel time, then FETCH_SIZE of chosen ones
set +e
O=gpurun_out/${1:-r04w}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sb in 4 6 8 11 16 32; do
SDPB_HIP_SYRK_SB=$sb timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>> $O/err.log | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric\"')][-1]
print('SB=$sb', d['value'], d['ms_per_step'], 'syrk', d['roofline']['launch_ms'])"
done
cd /tmp
for sb in ${2:-8 16}; do
  SDPB_HIP_SYRK_SB=$sb timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_sb$sb -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $R/$O/err.log
  for f in $(find $R/$O/pmc_sb$sb -name "*_results.db"); do python $R/profiles/summarize_pmc_db.py $f FETCH_SIZE > $R/$O/pmc_FETCH_SIZE_sb$sb.txt; done
  rm -rf $R/$O/pmc_sb$sb
  echo "SB=$sb"; head -3 $R/$O/pmc_FETCH_SIZE_sb$sb.txt
done
