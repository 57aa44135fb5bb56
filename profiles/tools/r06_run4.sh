#!/bin/bash
# Round 6: the whole GPU suite and the bench lines on the build that ships.
set +e
TAG=${1:-r06k}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
(time timeout 3000 python -m pytest tests -m gpu -q -x --durations=15) > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_C4_default.json 2> $O/bench_C4_default.err
tail -c 3000 $O/bench_C4_default.json
timeout 1200 python bench.py --workload C4f --scale 0.25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_C4f.json 2> $O/bench_C4f.err
tail -c 1500 $O/bench_C4f.json; tail -3 $O/bench_C4f.err
