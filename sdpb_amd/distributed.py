"""Cross-GPU exchange for world_size > 1: torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) behind the C-ABI collective callbacks.

The library hands raw device pointers it owns; they are wrapped zero-copy as torch
tensors.  PyTorch is plumbing here (process group + collectives), not compute.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.distributed as dist


class _CudaView:
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}


def tensor_from_pointer(ptr: int, nbytes: int, device: torch.device) -> torch.Tensor:
    """uint8 tensor aliasing `nbytes` at `ptr` (device memory for cuda, host for cpu)."""
    if device.type == "cuda":
        return torch.as_tensor(_CudaView(ptr, nbytes), device=device)
    buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


def make_collectives(device: torch.device, group=None):
    """Returns (allreduce_sum_u64, allgather_bytes) for SDPSolver.set_collectives.

    With device memory and a host-only backend (`gloo`: several ranks sharing ONE GPU, which
    RCCL refuses — the multi-rank device tests on a 1-GPU box) the buffers are staged through
    host memory around each collective."""
    world = dist.get_world_size(group)
    staged = device.type == "cuda" and dist.get_backend(group) == "gloo"

    def allreduce_sum_u64(ptr, count):
        try:
            t = tensor_from_pointer(ptr, count * 8, device).view(torch.int64)
            if staged:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
                torch.cuda.current_stream(device).synchronize()
                return 0
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # wrapping int64 sum == uint64 sum
            if device.type == "cuda":
                # only the stream the collective was ordered on: the library keeps an independent
                # stream busy (Cholesky(Q)) that a device-wide synchronize would wait for
                torch.cuda.current_stream(device).synchronize()
            return 0
        except Exception as e:  # pragma: no cover - surfaced through the C ABI as code 3
            print("allreduce callback failed:", e, flush=True)
            return 1

    def allgather_bytes(send, recv, nbytes):
        try:
            s = tensor_from_pointer(send, nbytes, device)
            r = tensor_from_pointer(recv, nbytes * world, device)
            if staged:
                hs = s.cpu()
                hr = torch.empty(world, nbytes, dtype=torch.uint8)
                dist.all_gather(list(hr.unbind(0)), hs, group=group)
                r.copy_(hr.view(-1))
                torch.cuda.current_stream(device).synchronize()
                return 0
            dist.all_gather_into_tensor(r, s, group=group) if device.type == "cuda" else \
                dist.all_gather(list(r.view(world, nbytes).unbind(0)), s, group=group)
            if device.type == "cuda":
                torch.cuda.current_stream(device).synchronize()
            return 0
        except Exception as e:  # pragma: no cover
            print("allgather callback failed:", e, flush=True)
            return 1

    return allreduce_sum_u64, allgather_bytes
