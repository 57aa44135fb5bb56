"""Build libsdpb_hip.so (gfx950) in-tree with hipcc: one object per compiled limb count.

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libsdpb_hip.so")
ALL_LIMBS = (6, 10, 16, 18, 24, 26, 34, 42, 50, 66)  # 128, 256, 400/448, 512, 640-704, 768, 1024, 1280, 1536, 2048 bits
# SDPB_LIMBS=18 builds a subset (developer iterations); the default builds every width.
LIMBS = tuple(int(x) for x in os.environ["SDPB_LIMBS"].split(",")) if os.environ.get("SDPB_LIMBS") else ALL_LIMBS
# Above 1024 bits the 32-column panel images of the chain kernels (k_chol_inv_lds: factor + inverse of a
# diagonal block in LDS, 182 KB at 42 limbs; k_qsolve_panel2: 191 KB) exceed the CU's 160 KB of LDS, so those
# widths are compiled with 16-column panels (every kernel and the host driver take the panel width from
# SDPB_PB; each limb count is its own set of template instantiations).
EXTRA_FLAGS = {42: ["-DSDPB_PB=16"], 50: ["-DSDPB_PB=16"], 66: ["-DSDPB_PB=16"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wno-unused-result", "-Wno-pass-failed"]


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]


def _digest(deps):
    import hashlib
    h = hashlib.sha1()
    for d in sorted(deps):
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, digest):
    """An object is fresh only if it was compiled from exactly the current sources (a content
    hash taken when its build started), so edits made while a compile is in flight are caught."""
    stamp = target + ".stamp"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def _mark(target, digest):
    with open(target + ".stamp", "w") as f:
        f.write(digest)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    digest = _digest(_sources() + [os.path.join(HERE, "..", "include", "sdpb_hip.h")])
    # Compile from a private snapshot of the sources: hipcc preprocesses a file twice (device
    # pass, then host pass), so an edit made while a build is running would otherwise give a
    # library whose host stubs and device code objects disagree.
    import shutil
    snap = os.path.join(OUT, "src_snapshot")
    shutil.rmtree(snap, ignore_errors=True)
    os.makedirs(os.path.join(snap, "sdpb_amd", "csrc"))
    os.makedirs(os.path.join(snap, "include"))
    for f in _sources():
        shutil.copy(f, os.path.join(snap, "sdpb_amd", "csrc"))
    shutil.copy(os.path.join(HERE, "..", "include", "sdpb_hip.h"), os.path.join(snap, "include"))
    if _digest([os.path.join(snap, "sdpb_amd", "csrc", f) for f in os.listdir(os.path.join(snap, "sdpb_amd", "csrc"))]
               + [os.path.join(snap, "include", "sdpb_hip.h")]) != digest:
        raise RuntimeError("sources changed while they were being snapshotted; run the build again")
    csrc = os.path.join(snap, "sdpb_amd", "csrc")
    import hashlib
    digest = hashlib.sha1((digest + repr((FLAGS, sorted(EXTRA_FLAGS.items())))).encode()).hexdigest()  # objects also depend on their flags
    jobs = []
    objs = []
    todo = []
    for nl in sorted(LIMBS, reverse=True):   # the widest mantissas compile longest: start them first
        obj = os.path.join(OUT, f"solver_{nl}.o")
        objs.append(obj)
        if force or _stale(obj, digest):
            todo.append(obj)
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(nl, []), f"-DSDPB_NL={nl}", "-c", os.path.join(csrc, "solver_nl.hip"), "-o", obj])
    obj = os.path.join(OUT, "capi.o")
    objs.append(obj)
    if force or _stale(obj, digest):
        todo.append(obj)
        jobs.append([HIPCC, *FLAGS, "-c", os.path.join(csrc, "capi.hip"), "-o", obj])
    if jobs:
        if verbose:
            print(f"[sdpb_amd.build] compiling {len(jobs)} objects with hipcc", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
        for o in todo:
            _mark(o, digest)
    if jobs or not os.path.exists(LIB):
        # RCCL (the in-library cross-GPU exchange, csrc/rccl_comm.hpp); in a process that has already
        # loaded PyTorch the dynamic linker binds librccl.so.1 / libamdhip64 to torch's copies
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lrccl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
