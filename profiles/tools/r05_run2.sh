#!/bin/bash
# round 5, second GPU call: (1) A/B on ONE box: the round-4 final build against the current one (tile-packed partial
# planes, B -> P without the copy, the reworked k_tridiag); (2) lanes per matrix of k_tridiag per size bucket;
# (3) the parity tests the tridiagonalisation feeds
set +e
O=gpurun_out/${1:-r05b}; mkdir -p $O
line() { python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], "it/s %.3f ms %.2f" % (d.get("value"), d.get("ms_per_step")), "syrk", round(d["roofline"].get("launch_ms"),2), "trsm", st.get("initializeSchurComplementSolver.Q.solve"), "steplen", st.get("stepLength"), "chol", st.get("choleskyDecomposition"), "gate", (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lib sdpb_amd/_variants/a_r04_final.so > $O/ab_r04_$rep.json 2>> $O/err.log; line $O/ab_r04_$rep.json
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/ab_r05_$rep.json 2>> $O/err.log; line $O/ab_r05_$rep.json
done
for ts in 64 128 256; do for tl in 128 256 512; do
  SDPB_HIP_TRI_T_SMALL=$ts SDPB_HIP_TRI_T_LARGE=$tl timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/tri_s${ts}_l${tl}.json 2>> $O/err.log; line $O/tri_s${ts}_l${tl}.json
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_at_size.py -m gpu -x -q > $O/gpu_tests_subset.log 2>&1; tail -4 $O/gpu_tests_subset.log
tail -5 $O/err.log
