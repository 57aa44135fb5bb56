#!/bin/bash
# Read-only probe of the GPU box: how many HIP devices, which compute/memory partition mode, and whether the
# partition controls are reachable from inside the container.  Changes nothing.
set +e
echo "== id / env"; id; env | grep -i -E "HIP_|ROCR_|CUDA_VISIBLE|GPU_DEVICE|HSA_" 
echo "== /dev/dri /dev/kfd"; ls -la /dev/dri /dev/kfd 2>&1
echo "== rocm-smi"; timeout 30 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -30
echo "== amd-smi version/list"; timeout 30 amd-smi version 2>&1 | head -5; timeout 30 amd-smi list 2>&1 | head -40
echo "== amd-smi partition"; timeout 30 amd-smi partition --current 2>&1 | head -40
timeout 30 amd-smi partition --accelerator 2>&1 | head -60
echo "== sysfs partition files"
for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition /sys/class/kfd/kfd/topology/nodes/*/gpu_id; do
  [ -e "$f" ] && { printf "%s : " "$f"; cat "$f" 2>&1; ls -la "$f"; }
done
echo "== mounts of sysfs"; grep -E " /sys( |/)" /proc/mounts | head
echo "== rocminfo agents"; timeout 60 rocminfo 2>&1 | grep -E "Marketing Name|Compute Unit|Uuid|Node:|Name: +gfx" | head -40
echo "== torch devices"; timeout 180 python -c "import torch;print(torch.cuda.device_count());[print(i,torch.cuda.get_device_properties(i)) for i in range(torch.cuda.device_count())]" 2>&1 | tail -5
echo "== capabilities"; grep -i cap /proc/self/status
