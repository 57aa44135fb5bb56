#!/bin/bash
set +e
TAG=${1:-r06o}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python profiles/tools/dense_arith.py sdpb_amd/_variants/nl98.so 2048 2>&1 | tee $O/dense_arith_nl98.txt
timeout 900 python profiles/tools/dense_arith.py sdpb_amd/libsdpb_hip.so 128 512 768 1024 1536 2048 2>&1 | tee $O/dense_arith_product.txt
