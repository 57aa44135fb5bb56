#!/bin/bash
# developer tool: time the C4 bench line with every library under sdpb_amd/_variants (on the GPU box's scratch copy of the repo)
set +e
O=gpurun_out/${1:-variants}; mkdir -p $O
export TMPDIR=/tmp
cp sdpb_amd/libsdpb_hip.so /tmp/libsdpb_hip.orig.so
for v in sdpb_amd/_variants/*.so; do
  n=$(basename $v .so)
  cp $v sdpb_amd/libsdpb_hip.so
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "int_syrk and 512" 2>&1 | tail -1
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C4_$n.json 2>> $O/err.log
  python - "$O/bench_C4_$n.json" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("ms_per_step"), "trsm", st.get("initializeSchurComplementSolver.Q.solve"), "chol", st.get("choleskyDecomposition"), "pred", st.get("computeSearchDirection(betaPredictor)"), "syrk", d["roofline"].get("launch_ms"), "macfrac", d["roofline"].get("limb_mac_frac_of_measured_valu_peak"), (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
cp /tmp/libsdpb_hip.orig.so sdpb_amd/libsdpb_hip.so
tail -3 $O/err.log
