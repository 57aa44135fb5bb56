#!/bin/bash
# FETCH_SIZE of k_syrk_fx3 against the shape of its work items (products per workgroup, row splits)
set +e
O=gpurun_out/${1:-r04x}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for cfg in "SDPB_HIP_SYRK_GSPLIT=1" "SDPB_HIP_SYRK_GSPLIT=7" "SDPB_HIP_SYRK_SPLITS=8" "SDPB_HIP_SYRK_SPLITS=16" "SDPB_HIP_SYRK_SPLITS=32" "SDPB_HIP_SYRK_GSPLIT=7 SDPB_HIP_SYRK_SPLITS=8"; do
  i=$((i+1))
  env $cfg timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_$i.json 2>> $R/$O/err.log
  for f in $(find $R/$O/pmc_$i -name "*_results.db"); do python $R/profiles/summarize_pmc_db.py $f FETCH_SIZE > $R/$O/pmc_FETCH_SIZE_$i.txt; done
  rm -rf $R/$O/pmc_$i
  echo "$cfg: $(grep k_syrk_fx3 $R/$O/pmc_FETCH_SIZE_$i.txt)  $(python -c "
import json
d=[json.loads(l) for l in open('$R/$O/bench_$i.json') if l.startswith('{\"metric\"')][-1]
print('syrk ms', d['roofline']['launch_ms'])")"
done
