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
LIMBS = (6, 10, 16, 18, 24, 26, 34)  # 128, 256, 400/448, 512, 640-704, 768, 1024 bits
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wno-unused-result", "-Wno-pass-failed"]


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    deps = _sources() + [os.path.join(HERE, "..", "include", "sdpb_hip.h")]
    jobs = []
    objs = []
    for nl in LIMBS:
        obj = os.path.join(OUT, f"solver_{nl}.o")
        objs.append(obj)
        if force or _stale(obj, deps):
            jobs.append([HIPCC, *FLAGS, f"-DSDPB_NL={nl}", "-c", os.path.join(CSRC, "solver_nl.hip"), "-o", obj])
    obj = os.path.join(OUT, "capi.o")
    objs.append(obj)
    if force or _stale(obj, deps):
        jobs.append([HIPCC, *FLAGS, "-c", os.path.join(CSRC, "capi.hip"), "-o", obj])
    if jobs:
        if verbose:
            print(f"[sdpb_amd.build] compiling {len(jobs)} objects with hipcc", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not os.path.exists(LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
