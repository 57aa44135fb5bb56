#!/bin/bash
# round 5, fourth GPU call: A/B of 512-bit variant libraries on one box (sdpb_amd/_variants/*.so, built by hand from a scratch
# copy of csrc): base, limb-shifted LDS operand reads in k_trsm_rlt_panel, odd-stride LDS image in k_tridiag; then the
# 512-bit parity tests on the shifted-read variant (copied over the product library on the box's scratch copy only)
set +e
O=gpurun_out/${1:-r05d}; mkdir -p $O
line() { python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], "it/s %.3f ms %.2f" % (d.get("value"), d.get("ms_per_step")), "syrk", round(d["roofline"].get("launch_ms"),2), "trsm", st.get("initializeSchurComplementSolver.Q.solve"), "steplen", st.get("stepLength"), "chol", st.get("choleskyDecomposition"), "Qchol", st.get("initializeSchurComplementSolver.Q.cholesky"), "pred", st.get("computeSearchDirection(betaPredictor)"), "gate", (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2; do for v in sdpb_amd/_variants/[b-z]*.so; do
  n=$(basename $v .so)
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lib $v > $O/var_${n}_$rep.json 2>> $O/err.log; line $O/var_${n}_$rep.json
done; done
if [ -f sdpb_amd/_variants/c_shift3.so ]; then
  cp sdpb_amd/libsdpb_hip.so /tmp/orig.so; cp sdpb_amd/_variants/c_shift3.so sdpb_amd/libsdpb_hip.so
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_at_size.py -m gpu -q -k "512 or golden or fixture or intermediate or live or C1 or synthetic" > $O/gpu_tests_shift3.log 2>&1; tail -6 $O/gpu_tests_shift3.log
  cp /tmp/orig.so sdpb_amd/libsdpb_hip.so
fi
tail -3 $O/err.log
