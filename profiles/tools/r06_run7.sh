#!/bin/bash
# Round 6: (1) one iteration of the 98-limb solver (in the 66-limb slot of a variant library) against the oracle, scalars and
# every intermediate array; (2) the new small feasible runs to optimality; (3) smoke()
set +e
TAG=${1:-r06n}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python profiles/tools/nl98_iteration.py sdpb_amd/_variants/nl98.so 2>&1 | tee $O/nl98_iteration.txt
timeout 1500 python -m pytest tests/test_gpu_parity_at_size.py -m gpu -q -s -k "small_feasible" 2>&1 | tee $O/small_feasible.txt | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
