"""TEST INFRASTRUCTURE ONLY: build the CPU-emulated twin of libsdpb_hip.so.

Compiles the *unmodified* sdpb_amd/csrc sources with g++ against tests/emu/hip_emu.hpp
(a stand-in for <hip/hip_runtime.h>) so the host logic and the kernels' index arithmetic
can be exercised without a GPU by `pytest -m "not gpu"`.  Never used by the product,
bench.py or smoke(); see hip_emu.hpp.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sdpb_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libsdpb_hip_emu.so")
# only the mantissa widths the CPU tests use (the factories are weak symbols): 128, 512, 664, 768, 1024, 1280 bits
LIMBS = (6, 18, 24, 26, 34, 42)
CXX = os.environ.get("CXX", "g++")
FLAGS = ["-O1", "-std=c++17", "-fPIC", "-fopenmp", "-x", "c++", "-I" + os.path.join(HERE, "include"),
         "-Wno-unknown-pragmas", "-Wno-attributes", "-DSDPB_NO_RCCL"]


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
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-4000:] + r.stderr[-8000:])
        raise RuntimeError("emu build failed")


VARIANTS = {"notoom4": (["-DSDPB_SYRK_NO_TOOM4"], (18,)),    # name -> (extra flags, limb counts): documented build options kept alive
            "notoom4k": (["-DSDPB_SYRK_NO_TOOM4K"], (18,)),
            "notoom5k": (["-DSDPB_SYRK_NO_TOOM5K"], (18,)),   # Toom-4 x Karatsuba (k_syrk_fx3 with carries) at 512 bits: the 487-bit image of rounds 4-5
            "dev18": ([], (18,)),
            "trace18": (["-DSDPB_TRACE_TRIMIN"], (18,))}                             # developer iterations: the 512-bit width alone


def build(force=False, panel=None, variant=None):
    """panel=None: the product configuration; panel=4: same sources with 4-column panels so that
    even the small golden SDPs run through the multi-panel Cholesky / triangular-solve paths;
    variant="notoom4": -DSDPB_SYRK_NO_TOOM4 (the two-level Karatsuba image at every precision);
    variant="notoom4k": -DSDPB_SYRK_NO_TOOM4K (Toom-4 alone, k_syrk_fx2<.., true>, the 495-bit image of round 3)."""
    global OUT, LIB
    base_out = os.path.join(HERE, "_build")
    OUT = base_out if panel is None else os.path.join(base_out, f"pb{panel}")
    extra = [] if panel is None else [f"-DSDPB_PB={panel}"]
    limbs = LIMBS if panel is None else (26,)
    if variant is not None:
        OUT = os.path.join(base_out, variant)
        extra, limbs = list(VARIANTS[variant][0]), VARIANTS[variant][1]
    LIB = os.path.join(OUT, "libsdpb_hip_emu.so")
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip_emu.hpp"),
                                                                 os.path.join(HERE, "hip_emu.cpp"),
                                                                 os.path.join(ROOT, "include", "sdpb_hip.h")]
    digest = _digest(deps)
    sys.path.insert(0, ROOT)
    from sdpb_amd.build import EXTRA_FLAGS as product_flags
    jobs, objs, todo = [], [], []
    for nl in limbs:
        obj = os.path.join(OUT, f"solver_{nl}.o")
        objs.append(obj)
        if force or _stale(obj, digest):
            todo.append(obj)
            # same per-width flags as the product build (16-column panels above 1024 bits) unless a panel width is forced
            per_nl = [] if (panel is not None or variant is not None) else product_flags.get(nl, [])
            jobs.append([CXX, *FLAGS, *extra, *per_nl, f"-DSDPB_NL={nl}", "-c", os.path.join(CSRC, "solver_nl.hip"), "-o", obj])
    for src, name in ((os.path.join(CSRC, "capi.hip"), "capi.o"), (os.path.join(HERE, "hip_emu.cpp"), "hip_emu.o")):
        obj = os.path.join(OUT, name)
        objs.append(obj)
        if force or _stale(obj, digest):
            todo.append(obj)
            jobs.append([CXX, *FLAGS, *extra, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(_run, jobs))
        for o in todo:
            _mark(o, digest)
    if jobs or not os.path.exists(LIB):
        _run([CXX, "-shared", "-fPIC", "-fopenmp", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
