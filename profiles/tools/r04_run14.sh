#!/bin/bash
# full GPU suite + C4 / C5slice x0.5 / C3 / singlet lines of the build with k_syrk_fx3 at 512 and 1024 bits
set +e
O=gpurun_out/${1:-r04t}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C4_20.json 2>> $O/err.log
timeout 900 python bench.py --workload C5slice --scale 0.5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_C5slice_x0.5.json 2>> $O/err.log
timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3.json 2>> $O/err.log
timeout 600 python bench.py --workload golden:singlet_cT --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_singlet_cT.json 2>> $O/err.log
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
# one product per workgroup where the output has few tiles?
SDPB_HIP_SYRK_GSPLIT=21 timeout 600 python bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/err.log | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric\"')][-1]
print('C3 gsplit=21', d['value'], d['ms_per_step'], 'syrk', d['roofline']['launch_ms'])"
SDPB_HIP_SYRK_GSPLIT=21 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/err.log | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{\"metric\"')][-1]
print('C4 gsplit=21', d['value'], d['ms_per_step'], 'syrk', d['roofline']['launch_ms'])"
