#!/bin/bash
set +e
TAG=${1:-r06p}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
python /dev/stdin sdpb_amd/_variants/*.so <<'PY' 2>&1 | tee $O/syrk_variants.txt
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdpb_amd.solver import SDPSolver
from tests import parity
sdp, _, _, _ = parity.load_case("1d")
for lib in sys.argv[1:]:
    s = SDPSolver(sdp, 512, lib_path=lib)
    for rows, cols in ((40000, 1000),):
        ms = [s.bench_op("syrk", rows, cols, 5) for _ in range(3)]
        print(os.path.basename(lib), rows, cols, " ".join(f"{m:.2f}" for m in ms), "ms (syrk_G: kernel + split sums + finish)", flush=True)
    s.close()
PY
for b in profiles/tools/nl_probe_66_98.bin; do [ -f $b ] && timeout 900 $b; done 2>&1 | tee $O/nl_probe_66_98.txt
