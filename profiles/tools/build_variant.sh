#!/bin/bash
# developer tool: a 512-bit-only (NL = 18) library built with extra -D flags, for A/B timing of kernel variants
#   usage: bash profiles/tools/build_variant.sh <name> [-DFLAG ...]   ->  sdpb_amd/_variants/<name>.so
set -e
NAME=$1; shift
OUT=sdpb_amd/_variants; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-pass-failed"
/opt/rocm/bin/hipcc $F "$@" -DSDPB_NL=18 -c sdpb_amd/csrc/solver_nl.hip -o $OUT/$NAME.solver_18.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name: _ZN4sdpb${KFILT:-10k_syrk_fx3}" | grep "VGPRs:\|Scratch\|Occupancy" || true
[ -f $OUT/capi.o ] || /opt/rocm/bin/hipcc $F -c sdpb_amd/csrc/capi.hip -o $OUT/capi.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/$NAME.solver_18.o $OUT/capi.o -L/opt/rocm/lib -lrccl
ls -la $OUT/$NAME.so
