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


class _Replay:
    """A stand-in for SDPSolver that replays a fixture (optionally with one field of one iteration perturbed): the gate's
    own logic is what is under test here, no device, no library."""

    def __init__(self, fx, perturb=None, wrong_reason=False):
        self.fx, self.k, self.perturb, self.wrong_reason = fx, 0, perturb, wrong_reason
        self.sdp = type("S", (), {"N": fx["N"], "J": fx["J"]})()
        self.terminate_reason = ""

    def iterate(self):
        if self.k >= len(self.fx["iterations"]):
            self.terminate_reason = "maxIterations exceeded" if self.wrong_reason else self.fx["terminate_reason"]
            return True
        self.k += 1
        return False

    def scalars(self):
        import mpmath
        rec = dict(self.fx["iterations"][self.k - 1])
        if self.perturb and self.perturb[0] == self.k:
            mpmath.mp.prec = 1200
            rec["mu"] = mpmath.nstr(mpmath.mpf(rec["mu"]) * (1 + mpmath.mpf(2) ** -self.perturb[1]), 330)
        return rec

    def scalar(self, key):
        return self.fx[key]


def test_parity_gate_of_the_bench_covers_every_timed_iteration_and_the_termination():
    """bench.py's gate on the full-size C4 fixture (round 5: 48 oracle iterations + 'maxComplementarity exceeded' in iteration
    49): a faithful replay passes and is followed to termination; a field that is 2^-260 off in an EARLY iteration fails the
    tighter bar of the iterations the clock sees although it is inside the whole-run floor 2^-256; the same error late fails the
    per-iteration pin to the measured trajectory;
    2^-250 anywhere fails; a different terminate reason fails."""
    sys.path.insert(0, libs.ROOT)
    import bench
    with open(bench.fixture_path("C4", 1.0)) as f:
        fx = json.load(f)
    assert len(fx["iterations"]) == 48 and fx["terminated_in_iteration"] == 49 and fx["terminate_reason"] == "maxComplementarity exceeded"
    assert len(fx["iterations"]) >= 25          # the driver's K + W (20 + 5) and the default invocation (4 + 1) are inside it
    g = bench.parity_gate(_Replay(fx), "C4", 1.0, 512)
    assert g["passed"] and g["iterations"] == 48 and g["followed_to_termination"] == {"iteration": 49, "reason": "maxComplementarity exceeded"}
    assert len(g["worst_log2_rel_by_iteration"]) == 48 and g["early_iterations_bar"]["first"] == 25
    assert not bench.parity_gate(_Replay(fx, perturb=(7, 260)), "C4", 1.0, 512)["passed"]
    # round 6: every iteration is pinned to what the device was measured at (gate_thresholds.json "#measured_by_iteration",
    # iteration 40: 2^-330.3) plus 12 bits, so an error in a LATE iteration that the whole-run floor would let through is refused
    late = bench.parity_gate(_Replay(fx, perturb=(40, 260)), "C4", 1.0, 512)
    assert not late["passed"] and late["pinned_to_measured_trajectory"] == {"margin_bits": 12.0, "iterations": 48}
    assert "drift against the measured trajectory" in str(late["violations"])
    assert bench.parity_gate(_Replay(fx, perturb=(40, 325)), "C4", 1.0, 512)["passed"]
    assert not bench.parity_gate(_Replay(fx, perturb=(40, 250)), "C4", 1.0, 512)["passed"]
    assert not bench.parity_gate(_Replay(fx, wrong_reason=True), "C4", 1.0, 512)["passed"]


def test_parity_gate_on_the_run_to_optimality_uses_the_conditioned_bar_only_where_the_iteration_is_ill_conditioned():
    """The strictly feasible fixture (C4 x0.25 shape, 159 iterations, then 'found primal-dual optimal solution'): the floor
    2^-(p/2) holds through iteration 142; the last 17 iterations, whose block condition number passes 2^(p/2-16), are held to
    cond 2^-(p-16) (parity.conditioned_tol_bits) -- and every iteration to the measured trajectory + 12 bits."""
    sys.path.insert(0, libs.ROOT)
    import bench
    from tests import parity
    with open(bench.fixture_path("C4f", 0.25)) as f:
        fx = json.load(f)
    assert fx["terminate_reason"] == "found primal-dual optimal solution" and fx["terminated_in_iteration"] == 160
    bars = [parity.conditioned_tol_bits(r, 512, 256) for r in fx["iterations"]]
    assert bars[:142] == [256] * 142 and all(196 <= b < 256 for b in bars[142:]) and bars == sorted(bars, reverse=True)
    g = bench.parity_gate(_Replay(fx), "C4f", 0.25, 512)
    assert g["passed"] and g["iterations_on_the_conditioned_bar"] == sum(b < 256 for b in bars)
    assert g["followed_to_termination"] == {"iteration": 160, "reason": "found primal-dual optimal solution"}
    assert not bench.parity_gate(_Replay(fx, perturb=(159, 240)), "C4f", 0.25, 512)["passed"]   # inside the conditioned bar, outside the pin
    assert not bench.parity_gate(_Replay(fx, perturb=(100, 300)), "C4f", 0.25, 512)["passed"]   # inside 2^-256, outside the pin
    with open(bench.fixture_path("C4", 1.0)) as f:
        assert all(parity.conditioned_tol_bits(r, 512, 256) == 256 for r in json.load(f)["iterations"])


def test_preflight_reads_the_child_through_one_reader(monkeypatch):
    """sdpb_amd/rccl_preflight.run with stand-in children (no GPU): (1) a child that writes its id and its verdict in ONE burst --
    the lines buffered behind the id must not be lost (round-5 advisor: the first line went through the TextIOWrapper, the rest
    through communicate() on the raw descriptor); (2) a child that never produces an id ends the wait at the deadline and is
    killed; (3) a rank other than 0 receives the id and reports its own child's verdict; (4) a non-zero exit is not ok."""
    import subprocess
    import sys
    import time
    from sdpb_amd import rccl_preflight as pf
    real_popen = subprocess.Popen
    script = {}

    def fake_popen(argv, **kw):
        return real_popen([sys.executable, "-c", script["src"]] + [a for a in argv if a.startswith("--id")] + argv[-1:], **kw)

    monkeypatch.setattr(pf.subprocess, "Popen", fake_popen)
    script["src"] = "import sys; sys.stdout.write('warming up\\nID abcd\\nsome chatter\\nPREFLIGHT OK {}\\n'); sys.stdout.flush()"
    seen = {}
    r = pf.run(0, 2, 0, lambda h: seen.setdefault("id", h), timeout=20.0)
    assert r["ok"] and seen["id"] == "abcd" and r["detail"].startswith("PREFLIGHT OK"), r
    script["src"] = "import time; print('starting', flush=True); time.sleep(60)"
    t0 = time.time()
    r = pf.run(0, 2, 0, lambda h: h, timeout=1.5)
    assert not r["ok"] and "no id" in r["detail"] and time.time() - t0 < 10, r
    script["src"] = "import sys; print('PREFLIGHT OK {}' if '--id' in sys.argv else 'no id given', flush=True)"
    r = pf.run(1, 2, 0, lambda h: "ef01", timeout=20.0)
    assert r["ok"], r
    script["src"] = "import sys; print('ID 12', flush=True); print('PREFLIGHT OK {}', flush=True); sys.exit(3)"
    r = pf.run(0, 2, 0, lambda h: h, timeout=20.0)
    assert not r["ok"], r
