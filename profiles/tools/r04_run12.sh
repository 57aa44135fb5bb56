#!/bin/bash
# Toom-4 x Karatsuba syrk (k_syrk_fx3): exactness on the device, the C4 bench line, C3
set +e
O=gpurun_out/${1:-r04p}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "int_syrk or syrk_Q or precision_400 or golden" > $O/gpu_tests_syrk.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_syrk.log; tail -5 $O/gpu_tests_syrk.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2>> $O/err.log
timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3.json 2>> $O/err.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "syrk", d["roofline"].get("launch_ms"), "macfrac", d["roofline"].get("limb_mac_frac_of_measured_valu_peak"), (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("tolerance_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
tail -5 $O/err.log
