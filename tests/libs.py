"""Which shared library a test drives.

  * product_lib(): sdpb_amd/libsdpb_hip.so built by hipcc for gfx950 — the thing that is
    shipped and measured; used by every `-m gpu` test.
  * emu_lib(): tests/emu/_build/libsdpb_hip_emu.so — the same sources compiled for the CPU
    against tests/emu/hip_emu.hpp; used by `-m "not gpu"` tests to cover host logic
    (launch sequences, descriptor tables, termination logic) without a GPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def product_lib() -> str:
    from sdpb_amd import build
    return build.LIB if os.path.exists(build.LIB) else build.build()


def emu_lib(panel=None, variant=None) -> str:
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return build_emu.build(panel=panel, variant=variant)
