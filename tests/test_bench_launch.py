"""`python bench.py --gpus N` must work unaided (the driver may also launch the ranks itself):
with WORLD_SIZE unset it re-executes itself under torch.distributed.run, one rank per GPU, and rank
0 prints ONE JSON line with n_gpus = N.  Checked here as a dry run of the launch plumbing on the
CPU emulation build over gloo (SDPB_BENCH_DRYRUN_LIB) — flagged as not a measurement."""
import json
import os
import subprocess
import sys

from tests import libs


def _run(cmd, env_extra):
    env = dict(os.environ, SDPB_BENCH_DRYRUN_LIB=libs.emu_lib(), OMP_NUM_THREADS="2", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, cwd=libs.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "C2"], {})
    assert out["n_gpus"] == 2 and out["DRY_RUN_NOT_A_MEASUREMENT"] and out["config"]["exchange"] == "callbacks"


def test_bench_under_the_drivers_launcher_and_single_rank_agree():
    """The driver's own command line (python -m torch.distributed.run ... bench.py --gpus 2) and the
    1-rank run compute the same iteration (P-obj at the second iteration, bit for bit)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "0",
                "--workload", "C2"], {})
    one = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "0", "--workload", "C2"], {})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    import mpmath
    with mpmath.workprec(400):   # local: other tests rely on the precision tests/parity.py sets
        a, b = mpmath.mpf(two["P-obj"]), mpmath.mpf(one["P-obj"])
        assert b != 0 and abs(a - b) <= mpmath.mpf(2) ** -200 * abs(b)   # rank-order sums differ in the last bits only
