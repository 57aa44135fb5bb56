"""The C-ABI library loads and exports every symbol include/sdpb_hip.h declares; pure
host entry points work without a GPU; the product fails loudly when no GPU is present."""
import ctypes
import os
import re

import pytest

from tests import libs

ROOT = libs.ROOT


def declared_symbols():
    with open(os.path.join(ROOT, "include", "sdpb_hip.h")) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(sdpb_hip_\w+)\s*\(", txt)) - {"sdpb_hip_ctx"})


def test_product_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(libs.product_lib())
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdpb_hip.h but not exported"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sdpb_amd.solver import SDPSolver, SDPBError
    from tests import parity
    sdp, meta, _, _ = parity.load_case("1d")
    with pytest.raises(SDPBError) as e:
        SDPSolver(sdp, meta["precision"])
    assert e.value.code == 3 and "no CPU path" in str(e.value)


def test_plan_blocks_is_a_partition_and_balanced():
    from sdpb_amd.solver import plan_blocks
    dims = [2] * 200 + [1] * 400
    K = [40] * 600
    for world in (1, 2, 4, 8):
        owners = plan_blocks(dims, K, 1000, world, lib_path=libs.product_lib())
        assert len(owners) == 600 and set(owners) == set(range(world))
        heavy = [sum(1 for j in range(200) if owners[j] == r) for r in range(world)]
        assert max(heavy) - min(heavy) <= 1  # the 200 expensive blocks are spread evenly


def test_host_u64_lane_image_roundtrip():
    lib = ctypes.CDLL(libs.product_lib())
    lib.sdpb_hip_host_encode_u64.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    lib.sdpb_hip_host_decode_u64.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_char_p,
                                             ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    planes = 10
    vals = [0, 1, -1, 2 ** 200 + 12345, -(2 ** 250) + 7, 2 ** 287 - 1]
    total = [0] * planes
    for v in vals:
        lanes = (ctypes.c_ulonglong * planes)()
        assert lib.sdpb_hip_host_encode_u64(str(v).encode(), planes, lanes) == 0
        total = [a + b for a, b in zip(total, lanes)]  # what an integer SUM all-reduce does
    lanes = (ctypes.c_ulonglong * planes)(*total)
    buf = ctypes.create_string_buffer(400)
    need = ctypes.c_size_t()
    assert lib.sdpb_hip_host_decode_u64(lanes, planes, buf, len(buf), ctypes.byref(need)) == 0
    assert int(buf.value) == sum(vals)


def test_header_is_usable_from_c_and_cxx(tmp_path):
    """include/sdpb_hip.h is the boundary: a C translation unit and a C++ one that take the address
    of every declared entry point must compile (gcc -std=c99 / g++ -std=c++11) and link against the
    product library."""
    import subprocess
    syms = declared_symbols()
    body = "#include \"sdpb_hip.h\"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n  fn p[] = {\n" + \
        ",\n".join(f"    (fn){s}" for s in syms) + "\n  };\n" \
        "  printf(\"%d entry points\\n\", (int)(sizeof p / sizeof p[0]));\n  return p[0] == 0;\n}\n"
    lib = libs.product_lib()
    libdir, libname = os.path.dirname(lib), os.path.basename(lib)
    for cc, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        src = tmp_path / f"use_header.{ext}"
        src.write_text(body)
        exe = tmp_path / f"use_header_{ext}"
        r = subprocess.run([cc, std, "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src),
                            "-o", str(exe), "-L" + libdir, "-l:" + libname, "-Wl,-rpath," + libdir,
                            "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_terminate_reason_codes_follow_the_reference_enum():
    """sdpb_hip_terminate_reason returns SDP_Solver_Terminate_Reason's enumerators in declaration
    order (SDP_Solver_Terminate_Reason.hxx:9-21), so the shim can static_cast."""
    from sdpb_amd.solver import SDPSolver
    from tests import parity
    sdp, meta, _, _ = parity.load_case("1d")
    s = SDPSolver(sdp, meta["precision"], dict(meta["params"], maxIterations=1), lib_path=libs.emu_lib())
    assert s.L.sdpb_hip_terminate_reason(s.h) == -1
    assert not s.iterate() and s.iterate()
    assert (s.L.sdpb_hip_terminate_reason(s.h), s.terminate_reason) == (6, "maxIterations exceeded")
    s.reset()
    s.set_max_runtime(0.0)
    s.set_params(dict(maxIterations=100))
    assert s.iterate()
    assert (s.L.sdpb_hip_terminate_reason(s.h), s.terminate_reason) == (7, "maxRuntime exceeded")
    s.set_max_runtime(1e9)
    s.reset()
    s.request_stop()
    assert s.iterate()
    assert (s.L.sdpb_hip_terminate_reason(s.h), s.terminate_reason) == (10, "SIGTERM signal received")
    s.close()


def test_integration_shim_helpers_compile_and_round_trip_with_real_gmp(tmp_path):
    """INTEGRATION.md's put/get on a real mpf_t (tests/shim/mpf_shim_check.cpp) against the library's
    binary ABI: g++ + libgmp + the emulation build (the gfx950 library exports the same entry points)."""
    import subprocess
    lib = libs.emu_lib()
    exe = tmp_path / "mpf_shim_check"
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I/opt/conda/include",
                        os.path.join(ROOT, "tests", "shim", "mpf_shim_check.cpp"), "-o", str(exe), "-L" + os.path.dirname(lib),
                        "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib), "-l:libgmp.so.10"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "round-trip exactly" in r.stdout, r.stdout + r.stderr
