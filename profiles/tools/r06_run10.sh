#!/bin/bash
# Round 6, the binary that ships (rebuilt after the last source change): probes, the whole GPU suite, the PMC traffic passes the
# bench line's roofline.traffic comes from (the digest of the kernel's source region changed), smoke, the default bench
set +e
TAG=${1:-r06y}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in profiles/tools/nl_probe_*.bin; do echo "== $b"; timeout 600 $b; done 2>&1 | tee $O/nl_probe.txt
(time timeout 3000 python -m pytest tests -m gpu -q --durations=8) > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $R/$O/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_pmc_$C.json 2>> $R/$O/err.log
done
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  for f in $(find $O/pmc_$C -name "*_results.db"); do python profiles/summarize_pmc_db.py $f $C > $O/pmc_$C.txt; done
done
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python profiles/make_pmc_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $TAG > /dev/null; cp profiles/pmc_k_syrk_fx.json $O/pmc_k_syrk_fx.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_C4_default_invocation.json 2> $O/bench_default.err; echo "default bench rc=$?"
python - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_C4_default_invocation.json") if l.startswith('{"metric"')][-1]
print(d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["parity_gate"]["worst_log2_rel"], d["parity_gate"]["passed"])
PY
