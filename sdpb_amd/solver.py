"""Host-side mirror of the reference's SDP_Solver interface over the C ABI.

`SDPSolver` plays the role of `SDP_Solver` (src/sdp_solve/SDP_Solver.hxx:28-112):
construct it from an SDP + Solver_Parameters, call `run()` (the loop of
SDP_Solver::run, run/run.cxx:322-467) or `iterate()` for a single pass of the loop
body; per-iteration records carry the keys of out/iterations.json
(print_iteration.cxx:91-104).  All arithmetic happens in HIP kernels behind
include/sdpb_hip.h; this module only moves strings and never computes.
"""
from __future__ import annotations

import ctypes
import json
import os
import time
from typing import Callable, Dict, List, Optional

from .sdp_io import SDP, block_text

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libsdpb_hip.so")

ITERATION_KEYS = ["mu", "P-obj", "D-obj", "gap", "P-err", "p-err", "D-err", "R-err",
                  "P-step", "D-step", "beta", "Q_cond_number", "max_block_cond_number"]
PARAM_NAMES = ["dualityGapThreshold", "primalErrorThreshold", "dualErrorThreshold",
               "initialMatrixScalePrimal", "initialMatrixScaleDual", "feasibleCenteringParameter",
               "infeasibleCenteringParameter", "stepLengthReduction", "maxComplementarity",
               "minPrimalStep", "minDualStep"]
FLAG_NAMES = ["maxIterations", "findPrimalFeasible", "findDualFeasible",
              "detectPrimalFeasibleJump", "detectDualFeasibleJump"]

ALLREDUCE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
ALLGATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class _Collectives(ctypes.Structure):
    _fields_ = [("allreduce_sum_u64", ALLREDUCE_CB), ("allgather_bytes", ALLGATHER_CB),
                ("user", ctypes.c_void_p)]


class SDPBError(RuntimeError):
    """Mirrors the reference's RUNTIME_ERROR; .code is the C-ABI return code."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


RCCL_ID_BYTES = 128  # SDPB_HIP_RCCL_ID_BYTES
_libs: Dict[str, ctypes.CDLL] = {}


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    path = path or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    # One HIP runtime per process: PyTorch's ROCm wheel bundles its own libamdhip64 (same
    # soname).  If torch is loaded AFTER this library the process ends up with two runtimes and
    # torch sees "no GPUs"; loaded first, the dynamic linker binds this library to torch's copy.
    # Processes that also use torch.distributed (bench.py, sdpb_amd/distributed.py) therefore
    # import torch first; do it here so the order never depends on the caller.
    if os.environ.get("SDPB_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(path):
        raise SDPBError(3, f"{path} not found: build the HIP extension first "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = ctypes.CDLL(path)
    c_int_p = ctypes.POINTER(ctypes.c_int)
    size_p = ctypes.POINTER(ctypes.c_size_t)
    L.sdpb_hip_create.argtypes = [ctypes.c_int, ctypes.c_int, c_int_p, c_int_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    ll_p = ctypes.POINTER(ctypes.c_longlong)
    L.sdpb_hip_create_with_costs.argtypes = [ctypes.c_int, ctypes.c_int, c_int_p, c_int_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ll_p, ctypes.POINTER(ctypes.c_void_p)]
    L.sdpb_hip_block_timings.argtypes = [ctypes.c_void_p, ll_p]
    L.sdpb_hip_block_clock_ticks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    L.sdpb_hip_plan_blocks_with_costs.argtypes = [ctypes.c_int, ll_p, ctypes.c_int, c_int_p]
    L.sdpb_hip_destroy.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_destroy.restype = None
    L.sdpb_hip_last_error.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_last_error.restype = ctypes.c_char_p
    L.sdpb_hip_set_param.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    L.sdpb_hip_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 4
    L.sdpb_hip_set_block.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_char_p] * 4
    L.sdpb_hip_set_block_f64.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p,
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.sdpb_hip_set_objective.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    L.sdpb_hip_init_state.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_iterate.argtypes = [ctypes.c_void_p, c_int_p]
    L.sdpb_hip_schur_solver_init.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_schur_solve.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_terminate_reason.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_terminate_string.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_terminate_string.restype = ctypes.c_char_p
    L.sdpb_hip_get_scalar.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, size_p]
    L.sdpb_hip_get_array.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_char_p, ctypes.c_size_t, size_p]
    L.sdpb_hip_set_array.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    ull_p_ = ctypes.POINTER(ctypes.c_ulonglong)
    L.sdpb_hip_set_block_mpf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ull_p_, ull_p_, ull_p_, ull_p_]
    L.sdpb_hip_set_objective_mpf.argtypes = [ctypes.c_void_p, ctypes.c_int, ull_p_, ull_p_]
    L.sdpb_hip_get_array_mpf.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ull_p_,
                                         ctypes.c_size_t, size_p]
    L.sdpb_hip_set_array_mpf.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ull_p_,
                                         ctypes.c_size_t]
    L.sdpb_hip_block_owner.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sdpb_hip_limbs.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_fx_frac_bits.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_bench_op.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_double)]
    L.sdpb_hip_set_collectives.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Collectives)]
    L.sdpb_hip_timers.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, size_p]
    L.sdpb_hip_rccl_unique_id.argtypes = [ctypes.c_char_p]
    L.sdpb_hip_rccl_init.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.sdpb_hip_comm_name.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_comm_name.restype = ctypes.c_char_p
    L.sdpb_hip_set_max_runtime.argtypes = [ctypes.c_void_p, ctypes.c_double]
    if hasattr(L, "sdpb_hip_memory_plan"):   # (absent only from an older build passed as lib_path for A/B timing)
        L.sdpb_hip_set_max_shared_memory.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
        L.sdpb_hip_memory_plan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, size_p]
    L.sdpb_hip_request_stop.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_request_stop.restype = None
    L.sdpb_hip_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sdpb_hip_host_syncs.argtypes = [ctypes.c_void_p]
    L.sdpb_hip_host_syncs.restype = ctypes.c_long
    L.sdpb_hip_progress.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    L.sdpb_hip_plan_blocks.argtypes = [ctypes.c_int, c_int_p, c_int_p, ctypes.c_int, ctypes.c_int, c_int_p]
    L.sdpb_hip_op_scalar.argtypes = [ctypes.c_void_p] + [ctypes.c_char_p] * 4 + [ctypes.c_size_t, size_p]
    L.sdpb_hip_op_int_syrk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                       ctypes.c_char_p, ctypes.c_size_t, size_p]
    L.sdpb_hip_op_syrk_Q.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                     ctypes.c_char_p, ctypes.c_size_t, size_p]
    if hasattr(L, "sdpb_hip_op_min_eigenvalue"):   # (absent only from an older build passed as lib_path for A/B timing)
        L.sdpb_hip_op_min_eigenvalue.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, size_p]
    ull_p = ctypes.POINTER(ctypes.c_ulonglong)
    L.sdpb_hip_host_encode_u64.argtypes = [ctypes.c_char_p, ctypes.c_int, ull_p]
    L.sdpb_hip_host_decode_u64.argtypes = [ull_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, size_p]
    _libs[path] = L
    return L


def copy_bandwidth_gbs(nbytes: int = 1 << 30, reps: int = 5, lib_path: Optional[str] = None) -> float:
    """Measured streaming-copy rate (read + write) of the current device in GB/s (sdpb_hip_copy_bandwidth)."""
    L = load_library(lib_path)
    out = ctypes.c_double(0.0)
    L.sdpb_hip_copy_bandwidth.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    rc = L.sdpb_hip_copy_bandwidth(nbytes, reps, ctypes.byref(out))
    if rc:
        raise SDPBError(rc, L.sdpb_hip_last_error(None).decode())
    return out.value


def plan_blocks(dims: List[int], num_points: List[int], N: int, world_size: int,
                lib_path: Optional[str] = None) -> List[int]:
    """Block -> rank assignment (pure host logic, no GPU needed)."""
    L = load_library(lib_path)
    J = len(dims)
    out = (ctypes.c_int * J)()
    rc = L.sdpb_hip_plan_blocks(J, (ctypes.c_int * J)(*dims), (ctypes.c_int * J)(*num_points), N, world_size, out)
    if rc:
        raise SDPBError(rc, "sdpb_hip_plan_blocks failed")
    return list(out)


class SDPSolver:
    def __init__(self, sdp: SDP, precision: int, params: Optional[dict] = None, device: int = -1,
                 rank: int = 0, world_size: int = 1, lib_path: Optional[str] = None,
                 upload_all_blocks: bool = True, block_source: Optional[Callable] = None,
                 block_costs: Optional[List[int]] = None):
        """block_source(j) -> (bases_even_rows, bases_odd_rows, B float64 [P,N], c float64 [P]) supplies
        blocks lazily as arrays (bulk synthetic inputs); otherwise sdp.blocks[j] holds decimal strings."""
        self.L = load_library(lib_path)
        self.sdp = sdp
        self.precision = precision
        self.rank, self.world_size = rank, world_size
        J = sdp.J
        h = ctypes.c_void_p()
        if block_costs is not None and len(block_costs) != J:   # read_block_costs.cxx:50-55
            raise SDPBError(4, f"Incompatible number of entries in block_timings: expected {J} but found {len(block_costs)}")
        costs = (ctypes.c_longlong * J)(*[int(c) for c in block_costs]) if block_costs is not None else None
        rc = self.L.sdpb_hip_create_with_costs(precision, J, (ctypes.c_int * J)(*sdp.dims),
                                               (ctypes.c_int * J)(*sdp.num_points), sdp.N, device, rank, world_size,
                                               costs, ctypes.byref(h))
        if rc:
            raise SDPBError(rc, self.L.sdpb_hip_last_error(None).decode())
        self.h = h
        self._cb_keepalive = None
        self.set_params(params or {})
        for j in range(J):
            if not (upload_all_blocks or self.block_owner(j) == rank):
                continue
            if block_source is not None:
                import numpy as np
                be, bo, B, c = block_source(j)
                B = np.ascontiguousarray(B, dtype=np.float64)
                c_text = list(c) if len(c) and isinstance(c[0], str) else None   # feasible synthetic family: c is no double
                c = np.zeros(len(c)) if c_text else np.ascontiguousarray(c, dtype=np.float64)
                dp = ctypes.POINTER(ctypes.c_double)
                self._chk(self.L.sdpb_hip_set_block_f64(
                    self.h, j, "\n".join(" ".join(r) for r in be).encode(),
                    "\n".join(" ".join(r) for r in bo).encode(), B.ctypes.data_as(dp), c.ctypes.data_as(dp)))
                if c_text:
                    self._chk(self.L.sdpb_hip_set_array(self.h, b"c", j, 0, " ".join(c_text).encode()))
            else:
                self._chk(self.L.sdpb_hip_set_block(self.h, j, *block_text(sdp.blocks[j])))
        self._chk(self.L.sdpb_hip_set_objective(self.h, " ".join(sdp.b).encode(), sdp.constant.encode()))
        self._chk(self.L.sdpb_hip_init_state(self.h))
        self.iteration = 0
        self.terminated = False

    # -- plumbing ---------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise SDPBError(rc, self.L.sdpb_hip_last_error(self.h).decode())

    def _string(self, fn, *args) -> str:
        need = ctypes.c_size_t(0)
        buf = ctypes.create_string_buffer(4096)
        rc = fn(self.h, *args, buf, len(buf), ctypes.byref(need))
        if rc == 4 and need.value > len(buf):
            buf = ctypes.create_string_buffer(need.value)
            rc = fn(self.h, *args, buf, len(buf), ctypes.byref(need))
        self._chk(rc)
        return buf.value.decode()

    def set_params(self, params: dict):
        flags = dict(maxIterations=500, findPrimalFeasible=0, findDualFeasible=0,
                     detectPrimalFeasibleJump=0, detectDualFeasibleJump=0)
        for k, v in params.items():
            if k in flags:
                flags[k] = int(v)
            elif k in PARAM_NAMES:
                self._chk(self.L.sdpb_hip_set_param(self.h, k.encode(), str(v).encode()))
            else:
                raise SDPBError(4, f"unknown solver parameter {k}")
        self._chk(self.L.sdpb_hip_set_flags(self.h, flags["maxIterations"], flags["findPrimalFeasible"],
                                            flags["findDualFeasible"], flags["detectPrimalFeasibleJump"],
                                            flags["detectDualFeasibleJump"]))

    def set_collectives(self, allreduce_sum_u64: Callable, allgather_bytes: Callable):
        """Register the cross-GPU exchange callbacks (device pointers in, 0 = ok out)."""
        cb = _Collectives(ALLREDUCE_CB(lambda user, p, n: int(allreduce_sum_u64(p, n))),
                          ALLGATHER_CB(lambda user, s, r, n: int(allgather_bytes(s, r, n))), None)
        self._cb_keepalive = cb
        self._chk(self.L.sdpb_hip_set_collectives(self.h, ctypes.byref(cb)))

    def rccl_unique_id(self) -> bytes:
        """Rank 0: the id every rank passes to rccl_init (distribute it by any means)."""
        buf = ctypes.create_string_buffer(RCCL_ID_BYTES)
        rc = self.L.sdpb_hip_rccl_unique_id(buf)
        if rc:
            raise SDPBError(rc, self.L.sdpb_hip_last_error(None).decode())
        return buf.raw

    def rccl_init(self, unique_id: bytes):
        """Collective over all ranks: the exchange runs on RCCL inside the library from now on."""
        assert len(unique_id) == RCCL_ID_BYTES
        self._chk(self.L.sdpb_hip_rccl_init(self.h, ctypes.create_string_buffer(unique_id, RCCL_ID_BYTES)))

    @property
    def comm_name(self) -> str:
        return self.L.sdpb_hip_comm_name(self.h).decode()

    def set_max_runtime(self, seconds: float):
        self._chk(self.L.sdpb_hip_set_max_runtime(self.h, float(seconds)))

    def request_stop(self):
        """Graceful stop at the next iteration boundary (SIGTERM semantics of run.cxx:332-355)."""
        self.L.sdpb_hip_request_stop(self.h)

    def set_profiling(self, on: bool):
        self._chk(self.L.sdpb_hip_set_profiling(self.h, int(bool(on))))

    @property
    def host_syncs(self) -> int:
        return int(self.L.sdpb_hip_host_syncs(self.h))

    def progress(self) -> dict:
        """Lock-free progress record (sdpb_hip_progress): safe to call from a watchdog thread while another
        thread is inside iterate() (ctypes releases the GIL during the call)."""
        out = (ctypes.c_ulonglong * 8)()
        self.L.sdpb_hip_progress(self.h, out)
        kinds = {0: None, 1: "all-gather", 2: "all-reduce", 3: "broadcast"}
        return {"iteration": int(out[0]), "host_syncs": int(out[1]), "collectives": int(out[2]),
                "sequence_hash": f"{int(out[3]):016x}", "last_collective": kinds.get(int(out[4]), int(out[4])),
                "last_bytes": int(out[5]), "last_root": int(out[6]) - 1 if out[6] else None,
                "transport_async_error": int(out[7])}

    # -- SDP_Solver surface -------------------------------------------------------
    def block_owner(self, j: int) -> int:
        return self.L.sdpb_hip_block_owner(self.h, j)

    @property
    def limbs(self) -> int:
        return self.L.sdpb_hip_limbs(self.h)

    @property
    def fx_frac_bits(self) -> int:
        """Fraction bits of the fixed-point image the exact integer Q syrk works on."""
        return self.L.sdpb_hip_fx_frac_bits(self.h)

    def iterate(self) -> bool:
        """One pass of the loop body of SDP_Solver::run; True when the loop ends."""
        t = ctypes.c_int(0)
        self._chk(self.L.sdpb_hip_iterate(self.h, ctypes.byref(t)))
        self.iteration += 1
        self.terminated = bool(t.value)
        return self.terminated

    def reset(self):
        """Back to the initial point X = Omega_p I, Y = Omega_d I, x = y = 0 (SDP_Solver.cxx:23-38)."""
        self._chk(self.L.sdpb_hip_init_state(self.h))
        self.iteration = 0
        self.terminated = False

    def schur_solver_init(self):
        """schur_complement_cholesky ("L"), schur_off_diagonal ("PT") and Cholesky(Q) ("Q") from the
        current X, Y — what approx_objective/outer_limits reuse (setup_solver.cxx:204-220)."""
        self._chk(self.L.sdpb_hip_schur_solver_init(self.h))

    def schur_solve(self):
        """Solve the Schur complement equation for the right-hand sides in "dx"/"dy" (in place)."""
        self._chk(self.L.sdpb_hip_schur_solve(self.h))

    def scalar(self, name: str) -> str:
        return self._string(self.L.sdpb_hip_get_scalar, name.encode())

    def scalars(self) -> dict:
        d = {k: self.scalar(k) for k in ITERATION_KEYS}
        d["block_name"] = self.scalar("block_name")
        return d

    def array(self, which: str, j: int = 0, parity: int = 0) -> List[str]:
        return self._string(self.L.sdpb_hip_get_array, which.encode(), j, parity).split()

    # -- binary number path: numpy uint64 arrays of mpf_t-layout records, shape (count, 2 + limbs64) --
    @staticmethod
    def _rec(a):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), a.shape[-1] - 2

    def set_block_mpf(self, j: int, bases_even, bases_odd, B, c):
        (e, pe, l1), (o, po, l2), (b, pb, l3), (cc, pc, l4) = map(self._rec, (bases_even, bases_odd, B, c))
        assert l1 == l2 == l3 == l4
        self._chk(self.L.sdpb_hip_set_block_mpf(self.h, j, l1, pe, po, pb, pc))

    def set_objective_mpf(self, b, constant):
        (bb, pb, l1), (cc, pc, l2) = self._rec(b), self._rec(constant)
        assert l1 == l2
        self._chk(self.L.sdpb_hip_set_objective_mpf(self.h, l1, pb, pc))

    def array_mpf(self, which: str, j: int = 0, parity: int = 0, limbs64: Optional[int] = None):
        import numpy as np
        limbs64 = limbs64 or self.limbs // 2 + 1
        n = ctypes.c_size_t(0)
        self._chk(self.L.sdpb_hip_get_array_mpf(self.h, which.encode(), j, parity, limbs64, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 2 + limbs64), dtype=np.uint64)
        self._chk(self.L.sdpb_hip_get_array_mpf(self.h, which.encode(), j, parity, limbs64,
                                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), n.value, ctypes.byref(n)))
        return out

    def set_array_mpf(self, which: str, values, j: int = 0, parity: int = 0):
        v, pv, l = self._rec(values)
        self._chk(self.L.sdpb_hip_set_array_mpf(self.h, which.encode(), j, parity, l, pv, v.shape[0]))

    def set_array(self, which: str, values: List[str], j: int = 0, parity: int = 0):
        self._chk(self.L.sdpb_hip_set_array(self.h, which.encode(), j, parity, " ".join(values).encode()))

    @property
    def terminate_reason(self) -> str:
        return self.L.sdpb_hip_terminate_string(self.h).decode()

    def timers(self) -> dict:
        return json.loads(self._string(self.L.sdpb_hip_timers))

    def set_max_shared_memory(self, nbytes: int):
        """--maxSharedMemory: bound on the scratch of the Q stage (here the syrk's partial planes); 0 = default plan."""
        self._chk(self.L.sdpb_hip_set_max_shared_memory(self.h, int(nbytes)))

    def memory_plan(self) -> dict:
        """Bytes per array class on this rank, the syrk's chunk plan, free/total device memory (run.cxx:79-181)."""
        return json.loads(self._string(self.L.sdpb_hip_memory_plan))

    def run(self, max_runtime: float = float("inf"), on_iteration: Optional[Callable] = None) -> str:
        """SDP_Solver::run: iterate until a terminate reason is set; returns it.  max_runtime is
        tested inside the iteration (compute_feasible_and_termination.cxx:51-56)."""
        if max_runtime < float("inf"):
            self.set_max_runtime(max_runtime)
        while True:
            if self.iterate():
                return self.terminate_reason
            rec = {"iteration": self.iteration, **self.scalars()}
            if on_iteration:
                on_iteration(rec)

    def out_txt(self) -> dict:
        """The keys of out.txt (src/sdpb/save_solution.cxx:32-37)."""
        return {"terminateReason": self.terminate_reason,
                **{k: self.scalar(k) for k in ("primalObjective", "dualObjective", "dualityGap",
                                               "primalError", "dualError")}}

    # -- operator-level entry points -----------------------------------------------
    def op_scalar(self, op: str, a, b="0") -> str:
        return self._string(self.L.sdpb_hip_op_scalar, op.encode(), str(a).encode(), str(b).encode())

    def op_int_syrk(self, rows: int, cols: int, ints_colmajor) -> List[int]:
        txt = " ".join(str(v) for v in ints_colmajor).encode()
        return [int(s) for s in self._string(self.L.sdpb_hip_op_int_syrk, rows, cols, txt).split()]

    def op_syrk_Q(self, rows: int, cols: int, P_colmajor) -> List[str]:
        """Q = P^T P through the whole syrk_Q stage (norms, fixed-point image, exact integer syrk,
        diagonal check, restore): lower triangle, column-major decimals."""
        txt = " ".join(str(v) for v in P_colmajor).encode()
        return self._string(self.L.sdpb_hip_op_syrk_Q, rows, cols, txt).split()

    def op_min_eigenvalue(self, n: int, A_colmajor) -> str:
        """Smallest eigenvalue of a symmetric n x n matrix through the kernels of the step length (min_eigenvalue.cxx:8-33)."""
        return self._string(self.L.sdpb_hip_op_min_eigenvalue, n, " ".join(str(v) for v in A_colmajor).encode())

    def block_timings(self) -> List[int]:
        """Microseconds per iteration for the blocks this rank owns (0 elsewhere); needs profiled iterations."""
        out = (ctypes.c_longlong * self.sdp.J)()
        self._chk(self.L.sdpb_hip_block_timings(self.h, out))
        return list(out)

    def block_clock_ticks(self):
        """Raw per-block counters behind block_timings: (cholesky, solve) ticks of the 100 MHz device clock summed
        over the workgroups that worked on each block during the profiled iterations (0 for other ranks' blocks)."""
        J = self.sdp.J
        a, b = (ctypes.c_ulonglong * J)(), (ctypes.c_ulonglong * J)()
        self._chk(self.L.sdpb_hip_block_clock_ticks(self.h, a, b))
        return list(a), list(b)

    def bench_op(self, op: str, a: int, b: int, reps: int = 3) -> float:
        """Average HIP-event time (ms) of one kernel of the iteration on synthetic operands."""
        ms = ctypes.c_double(0.0)
        self._chk(self.L.sdpb_hip_bench_op(self.h, op.encode(), a, b, reps, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            self.L.sdpb_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
