#!/bin/bash
# round 5, GPU call: (1) the tests of the new 66-limb width (--precision 2048) on the product library; (2) A/B of 512-bit
# variant libraries on one box (sdpb_amd/_variants/*.so): base against the division-free Newton iteration of k_tridiag_min
# (-DSDPB_TRIMIN_POLY=1), bench lines + a kernel trace of each; (3) the step-length parity tests on the variant
set +e
O=gpurun_out/${1:-r05f}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "2048 or 1700 or beyond or 66" > $O/gpu_tests_2048.log 2>&1; tail -4 $O/gpu_tests_2048.log
line() { python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    st=d.get("stage_ms_profiled_iteration",{})
    print(sys.argv[1].split('/')[-1], "it/s %.3f ms %.2f" % (d.get("value"), d.get("ms_per_step")), "syrk", round(d["roofline"].get("launch_ms"),2), "trsm", st.get("initializeSchurComplementSolver.Q.solve"), "steplen", st.get("stepLength"), "gate", (d.get("parity_gate") or {}).get("worst_log2_rel"), (d.get("parity_gate") or {}).get("passed"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
for rep in 1 2; do for v in sdpb_amd/_variants/[b-z]*.so; do
  n=$(basename $v .so)
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lib $v > $O/var_${n}_$rep.json 2>> $O/err.log; line $O/var_${n}_$rep.json
done; done
cd /tmp
for v in $R/sdpb_amd/_variants/[b-z]*.so; do
  n=$(basename $v .so)
  timeout 600 rocprofv3 --kernel-trace -d $R/$O/trace_$n -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --lib $v > $R/$O/bench_trace_$n.json 2>> $R/$O/err.log
  for f in $(find $R/$O/trace_$n -name "*_results.db"); do python $R/profiles/tools/rocpd_stats.py $f --top 70 > $R/$O/kernel_stats_$n.txt; done
  rm -rf $R/$O/trace_$n
  grep "k_tridiag" $R/$O/kernel_stats_$n.txt
done
cd $R
for v in sdpb_amd/_variants/[c-z]*.so; do
  n=$(basename $v .so)
  cp sdpb_amd/libsdpb_hip.so /tmp/orig.so; cp $v sdpb_amd/libsdpb_hip.so
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_at_size.py -m gpu -q -k "512 or golden or fixture or intermediate or live or C1 or synthetic" > $O/gpu_tests_$n.log 2>&1; tail -6 $O/gpu_tests_$n.log
  cp /tmp/orig.so sdpb_amd/libsdpb_hip.so
done
tail -3 $O/err.log
