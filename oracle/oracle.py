"""ctypes wrapper around oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; nothing under sdpb_amd/ does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

SCALARS = ["mu", "P-obj", "D-obj", "gap", "P-err", "p-err", "D-err", "R-err",
           "P-step", "D-step", "beta", "Q_cond_number", "max_block_cond_number"]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "sdpb_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_last_error.argtypes = [ctypes.c_void_p]
        L.orc_set_param.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 4
        L.orc_set_block.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_char_p] * 4
        L.orc_set_block_f64.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p,
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.orc_set_block_c.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p]
        L.orc_set_objective.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        L.orc_init_state.argtypes = [ctypes.c_void_p]
        L.orc_iterate.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.orc_terminate_reason.argtypes = [ctypes.c_void_p]
        L.orc_terminate_string.restype = ctypes.c_char_p
        L.orc_terminate_string.argtypes = [ctypes.c_void_p]
        L.orc_get_scalar.restype = ctypes.c_char_p
        L.orc_get_scalar.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_get_array.restype = ctypes.c_char_p
        L.orc_get_array.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.orc_int_syrk.restype = ctypes.c_char_p
        L.orc_int_syrk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        L.orc_parse_exact.restype = ctypes.c_char_p
        L.orc_parse_exact.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_scalar_op.restype = ctypes.c_char_p
        L.orc_scalar_op.argtypes = [ctypes.c_void_p] + [ctypes.c_char_p] * 3
        L.orc_min_eigenvalue.restype = ctypes.c_char_p
        L.orc_min_eigenvalue.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p]
        L.orc_get_records.restype = ctypes.c_long
        L.orc_get_records.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_long]
        L.orc_schur_solver_init.argtypes = [ctypes.c_void_p]
        L.orc_schur_solve.argtypes = [ctypes.c_void_p]
        L.orc_set_array.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        L.orc_set_threads.restype = ctypes.c_int
        L.orc_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def usable_cpus() -> int:
    """Host cores this process may really use: min(cpu_count, affinity mask, cgroup CPU quota).
    (A GPU box can show 256 logical CPUs while its container is limited to 16: 256 OpenMP threads
    then run 16x slower than 16.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = float(g.read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


class Oracle:
    """GMP-mpf restatement of the reference iteration, driven like the product solver."""

    def __init__(self, sdp, precision: int, params: dict | None = None, param_prec: int = 64,
                 threads: int = 0, block_source=None):
        """threads: host threads for the block/column loops (0 = $ORACLE_THREADS or all usable cores;
        results are bit-identical for any count).  block_source(j) -> (bases_even, bases_odd,
        B float64 [P,N], c float64 [P]) feeds blocks lazily (sdpb_amd.synthetic.make_lazy)."""
        from sdpb_amd.sdp_io import block_text  # pure-python I/O helper, no compute
        self.L = lib()
        if not threads and not os.environ.get("ORACLE_THREADS"):
            threads = usable_cpus()
        self.threads = self.L.orc_set_threads(int(threads))
        J = sdp.J
        dims = (ctypes.c_int * J)(*sdp.dims)
        npts = (ctypes.c_int * J)(*sdp.num_points)
        self.h = ctypes.c_void_p(self.L.orc_create(precision, J, dims, npts, sdp.N))
        self.sdp = sdp
        flags = dict(maxIterations=500, findPrimalFeasible=0, findDualFeasible=0,
                     detectPrimalFeasibleJump=0, detectDualFeasibleJump=0)
        for k, v in (params or {}).items():
            if k in flags:
                flags[k] = int(v)
            else:
                self._chk(self.L.orc_set_param(self.h, k.encode(), str(v).encode(), param_prec))
        self.L.orc_set_flags(self.h, flags["maxIterations"], flags["findPrimalFeasible"],
                             flags["findDualFeasible"], flags["detectPrimalFeasibleJump"],
                             flags["detectDualFeasibleJump"])
        if block_source is not None:
            import numpy as np
            dp = ctypes.POINTER(ctypes.c_double)
            for j in range(J):
                be, bo, Bv, cv = block_source(j)
                Bv = np.ascontiguousarray(Bv, dtype=np.float64)
                c_text = list(cv) if len(cv) and isinstance(cv[0], str) else None   # feasible family: c is no double
                cv = np.zeros(len(cv)) if c_text else np.ascontiguousarray(cv, dtype=np.float64)
                flat = lambda rows: " ".join(" ".join(r) for r in rows).encode()
                self._chk(self.L.orc_set_block_f64(self.h, j, flat(be), flat(bo), Bv.ctypes.data_as(dp),
                                                   cv.ctypes.data_as(dp)))
                if c_text:
                    self._chk(self.L.orc_set_block_c(self.h, j, " ".join(c_text).encode()))
        else:
            for j, blk in enumerate(sdp.blocks):
                self._chk(self.L.orc_set_block(self.h, j, *block_text(blk)))
        self._chk(self.L.orc_set_objective(self.h, " ".join(sdp.b).encode(),
                                           sdp.constant.encode()))
        self._chk(self.L.orc_init_state(self.h))

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self.L.orc_last_error(self.h).decode())

    def iterate(self) -> bool:
        """One loop body of SDP_Solver::run; returns True when the loop terminates."""
        t = ctypes.c_int(0)
        self._chk(self.L.orc_iterate(self.h, ctypes.byref(t)))
        return bool(t.value)

    def scalar(self, name: str) -> str:
        return self.L.orc_get_scalar(self.h, name.encode()).decode()

    def scalars(self) -> dict:
        d = {k: self.scalar(k) for k in SCALARS}
        d["block_name"] = self.scalar("block_name")
        return d

    def array(self, which: str, j: int = 0, parity: int = 0):
        return self.L.orc_get_array(self.h, which.encode(), j, parity).decode().split()

    def records(self, which: str, j: int = 0, parity: int = 0, limbs64: int = 10):
        """The oracle's own mpf_t values as raw records (_mp_size, _mp_exp, _mp_d[0..limbs64)):
        numpy uint64 array of shape (count, 2 + limbs64)."""
        import numpy as np
        n = self.L.orc_get_records(self.h, which.encode(), j, parity, limbs64, None, 0)
        if n < 0:
            raise OracleError(f"unknown array {which}")
        out = np.zeros((n, 2 + limbs64), dtype=np.uint64)
        self.L.orc_get_records(self.h, which.encode(), j, parity, limbs64, out.ctypes.data_as(ctypes.c_void_p), n)
        return out

    def set_records(self, which: str, recs, j: int = 0, parity: int = 0):
        """Bit-exact restore of x, X, y or Y from what records() returned (same precision, limbs64 wide enough)."""
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint64)
        self.L.orc_set_records.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        self._chk(self.L.orc_set_records(self.h, which.encode(), j, parity, recs.shape[1] - 2, recs.ctypes.data_as(ctypes.c_void_p), recs.shape[0]))

    def schur_solver_init(self):
        """L_j, P_j, Cholesky(Q) from the current X, Y (approx_objective/setup_solver.cxx:204-220)."""
        self._chk(self.L.orc_schur_solver_init(self.h))

    def schur_solve(self):
        """solve_schur_complement_equation.cxx:16-79 on the right-hand sides in dx / dy (in place)."""
        self._chk(self.L.orc_schur_solve(self.h))

    def set_array(self, which: str, values, j: int = 0, parity: int = 0):
        self._chk(self.L.orc_set_array(self.h, which.encode(), j, parity, " ".join(values).encode()))

    @property
    def terminate_reason(self) -> str:
        return self.L.orc_terminate_string(self.h).decode()

    def int_syrk(self, rows, cols, ints_colmajor):
        txt = " ".join(str(v) for v in ints_colmajor).encode()
        return [int(s) for s in self.L.orc_int_syrk(self.h, rows, cols, txt).decode().split()]

    def parse_exact(self, value, prec_bits=64) -> str:
        """Exact decimal of the number GMP obtains when parsing `value` at prec_bits."""
        n, k = self.L.orc_parse_exact(self.h, str(value).encode(), prec_bits).decode().split()
        n, k = int(n), int(k)
        sign = "-" if n < 0 else ""
        digits = str(abs(n) * 5 ** k)  # n / 2^k = n 5^k / 10^k
        if k == 0:
            return sign + digits
        digits = digits.rjust(k + 1, "0")
        return sign + digits[:-k] + "." + digits[-k:]

    def scalar_op(self, op, a, b="0"):
        return self.L.orc_scalar_op(self.h, op.encode(), str(a).encode(), str(b).encode()).decode()

    def min_eigenvalue(self, n, A_colmajor) -> str:
        """min_eigenvalue.cxx:8-33 on one symmetric matrix (column-major decimals)."""
        return self.L.orc_min_eigenvalue(self.h, n, " ".join(str(v) for v in A_colmajor).encode()).decode()

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
