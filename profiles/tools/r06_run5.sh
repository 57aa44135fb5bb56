#!/bin/bash
# Round 6: (1) cross-width probe of the step-length / condition-number kernels; (2) timing experiment: the syrk kernel with
# half of its LDS reads (wrong results, timing only) against the shipped one
set +e
TAG=${1:-r06l}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for b in profiles/tools/nl_probe_*.bin; do echo "== $b"; timeout 600 $b; done 2>&1 | tee $O/nl_probe.txt
cat > /tmp/syrk_time.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdpb_amd.solver import SDPSolver
from tests import parity
sdp, _, _, _ = parity.load_case("1d")
for lib in sys.argv[1:]:
    s = SDPSolver(sdp, 512, lib_path=lib)
    for rows, cols in ((40000, 1000), (10000, 250)):
        ms = [s.bench_op("syrk", rows, cols, 5) for _ in range(3)]
        print(os.path.basename(lib), rows, cols, " ".join(f"{m:.2f}" for m in ms), "ms (syrk_G: kernel + split sums + finish)", flush=True)
    s.close()
PY
python /tmp/syrk_time.py sdpb_amd/libsdpb_hip.so sdpb_amd/_variants/*.so 2>&1 | tee $O/syrk_lds_experiment.txt
