#!/bin/bash
# round 5, third GPU call: half of C5 (J = 4096, N = 2048, 1024 bits) with 4 and 2 in-library RCCL ranks sharing the GPU,
# replicated and distributed Cholesky(Q)
set +e
O=gpurun_out/${1:-r05c}; mkdir -p $O
rocm-smi --showmeminfo vram > $O/vram_before.txt 2>&1
timeout 3300 python profiles/tools/c5_half_multirank.py 2 0 > $O/c5_half_replicated.json 2> $O/c5_half_replicated.err; echo "replicated rc $?"
tail -c 1500 $O/c5_half_replicated.json; tail -5 $O/c5_half_replicated.err
timeout 3300 python profiles/tools/c5_half_multirank.py 2 1 > $O/c5_half_distributed.json 2> $O/c5_half_distributed.err; echo "distributed rc $?"
tail -c 1500 $O/c5_half_distributed.json; tail -5 $O/c5_half_distributed.err
