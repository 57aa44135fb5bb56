"""the sdpb-compatible driver (sdpb_amd/run.py) reads the unchanged SDP
directory and writes sdpb's own result files.  The files it writes are parsed back and
diffed against the reference's golden out/ directories with the reference's tolerance
(end-to-end.test.cxx:27: 2^-99 relative; diff_sdpb_out.cxx compares out.txt, y.txt, x_<j>.txt
and iterations.json).  CPU: the emulation build; GPU: the real library."""
import json
import os

import pytest

from sdpb_amd import run
from tests import libs, parity


def _vector(path):
    with open(path) as f:
        tok = f.read().split()
    rows, cols = int(tok[0]), int(tok[1])
    assert cols == 1 and len(tok) == 2 + rows, path
    return tok[2:]


def _check_outputs(name, out_dir, tol_bits=99):
    sdp, meta, iters, out = parity.load_case(name)
    got = {}
    import re
    with open(os.path.join(out_dir, "out.txt")) as f:
        for m in re.finditer(r"(\w[\w ]*?)\s*=\s*([^;]+);", f.read()):
            got[m.group(1).strip()] = m.group(2).strip().strip('"')
    assert list(got) == ["terminateReason", "primalObjective", "dualObjective", "dualityGap", "primalError",
                         "dualError", "Solver runtime"]
    assert got["terminateReason"] == out["terminateReason"]
    for k in ("primalObjective", "dualObjective"):
        assert parity.log2_rel(got[k], out[k]) <= -tol_bits, k
    with open(os.path.join(out_dir, "iterations.json")) as f:
        mine = json.load(f)
    assert len(mine) == len(iters)
    for g, w in zip(mine, iters):
        assert g["iteration"] == w["iteration"] and set(w) <= set(g)
        bad, _ = parity.compare_iteration(g, w, tol_bits)
        assert not bad, (name, w["iteration"], bad)
    golden = os.path.join(parity.GOLDEN, name)
    if os.path.exists(os.path.join(golden, "c_minus_By.json")):   # save_c_minus_By.hxx
        with open(os.path.join(out_dir, "c_minus_By", "c_minus_By.json")) as f:
            mine_c = json.load(f)["c_minus_By"]
        with open(os.path.join(golden, "c_minus_By.json")) as f:
            want_c = json.load(f)["c_minus_By"]
        assert [len(b) for b in mine_c] == [len(b) for b in want_c]
        scale = max(abs(parity.mpmath.mpf(v)) for b in want_c for v in b)
        for bm, bw in zip(mine_c, want_c):
            for u, v in zip(bm, bw):
                assert abs(parity.mpmath.mpf(u) - parity.mpmath.mpf(v)) <= parity.mpmath.mpf(2) ** -tol_bits * scale
    files = ["y.txt"] + [f"x_{j}.txt" for j in range(sdp.J)]
    for fn in files:
        a, b = _vector(os.path.join(out_dir, fn)), _vector(os.path.join(golden, fn))
        assert len(a) == len(b), fn
        scale = max(abs(parity.mpmath.mpf(v)) for v in b)
        for u, v in zip(a, b):
            # diff.hxx:50-76 on each element; elements far below the vector's scale carry no
            # information at 2^-99 relative, compare those against the scale instead
            d = abs(parity.mpmath.mpf(u) - parity.mpmath.mpf(v))
            assert d <= parity.mpmath.mpf(2) ** -tol_bits * (abs(parity.mpmath.mpf(u)) + abs(parity.mpmath.mpf(v)) + scale), fn


def _argv(name, out_dir, lib=None):
    _, meta, _, _ = parity.load_case(name)
    # -c: the checkpoint directory (where the timing run writes block_timings) next to the OUTPUT, never next to the
    # fixture: sdpb's default <sdpDir>.ck would write into tests/golden/ and later runs would plan on a stale file
    argv = ["-s", os.path.join(parity.GOLDEN, name, "sdp"), "-o", out_dir, "-c", out_dir.rstrip("/") + ".ck",
            "--precision", str(meta["precision"]), "--verbosity", "0", "--writeSolution", "x,y,z,X,Y"]
    for k, v in meta["params"].items():   # the flags end-to-end.test.cxx passes, verbatim
        if k in parity.FLAG_KEYS and k != "maxIterations":
            if int(v):
                argv.append("--" + k)
        else:
            argv += ["--" + k, str(v)]
    if lib:
        argv += ["--lib", lib]
    return argv


def test_driver_writes_sdpb_result_files_emulated(tmp_path):
    out_dir = str(tmp_path / "out")
    reason = run.solve(_argv("dfibo", out_dir, libs.emu_lib()))
    assert reason == "found primal-dual optimal solution" or reason
    _check_outputs("dfibo", out_dir)
    # X/Y files: "n n" header then n rows of n numbers (write_distmatrix.hxx)
    with open(os.path.join(out_dir, "X_matrix_0.txt")) as f:
        n, m = map(int, f.readline().split())
        rows = [ln.split() for ln in f if ln.strip()]
    assert n == m == len(rows) and all(len(r) == n for r in rows)


def test_max_shared_memory_option_and_memory_plan_of_the_driver(tmp_path, capsys):
    """--maxSharedMemory as the reference's end-to-end tests pass it (100.1K: end-to-end.test.cxx:348-357; suffixes of
    Solver_Parameters.cxx:61-72) reaches sdpb_hip_set_max_shared_memory, --verbosity 2 prints the rank's memory plan
    (the reference prints its estimates there, run.cxx:79-181), and the result files are the reference's."""
    assert run.parse_memory_size("100.1K") == int(100.1 * 1024) and run.parse_memory_size("2GB") == 2 << 30
    assert run.parse_memory_size("0") == 0 and run.parse_memory_size("3M") == 3 << 20 and run.parse_memory_size("17B") == 17
    out_dir = str(tmp_path / "out")
    argv = _argv("dfibo", out_dir, libs.emu_lib()) + ["--maxSharedMemory", "100.1K"]
    argv[argv.index("--verbosity") + 1] = "2"
    assert run.solve(argv)
    text = capsys.readouterr().out
    assert "Memory plan of this rank (bytes): psd_state_and_scratch" in text and "budget_source maxSharedMemory" in text
    assert f"window_budget_bytes {int(100.1 * 1024) // 4 * 4}" in text and "image_chunks" in text   # both windows under the one bound
    _check_outputs("dfibo", out_dir)


def test_text_checkpoint_restart_continues_the_same_trajectory(tmp_path):
    _checkpoint_restart(tmp_path, libs.emu_lib())


@pytest.mark.gpu
def test_text_checkpoint_restart_continues_the_same_trajectory_gpu(tmp_path):
    _checkpoint_restart(tmp_path, None)


@pytest.mark.gpu
def test_sigterm_stops_the_device_run_gracefully_with_a_usable_checkpoint(tmp_path):
    """run.cxx:332-355 + solve.cxx:99-104: SIGTERM -> finish the iteration in flight, write the text
    checkpoint, terminate reason "SIGTERM signal received"; restarting from it continues the trajectory."""
    import signal
    import subprocess
    import sys
    import time
    out = tmp_path / "out"
    argv = [a for a in _argv("singlet_cT", str(out)) if a]
    k = argv.index("--maxIterations") if "--maxIterations" in argv else None
    if k is not None:
        del argv[k:k + 2]
    p = subprocess.Popen([sys.executable, "-m", "sdpb_amd.run"] + argv + ["--maxIterations", "100000", "--verbosity", "1"],
                         cwd=libs.ROOT, stdout=subprocess.PIPE, text=True)
    seen = 0
    for line in p.stdout:                      # wait until a few iterations have been printed
        if line[:1].isdigit():
            seen += 1
            if seen == 5:
                p.send_signal(signal.SIGTERM)
                break
    rest = p.stdout.read()
    assert p.wait(timeout=300) == 0
    assert "SIGTERM signal received" in rest
    with open(out / "out.txt") as f:
        assert 'terminateReason = "SIGTERM signal received"' in f.read()
    with open(out / "iterations.json") as f:
        done = json.load(f)
    assert 5 <= len(done) < 170                 # stopped early, file is well-formed JSON
    assert os.path.exists(out / "X_matrix_0.txt") and os.path.exists(out / "y.txt")
    # restart from the checkpoint: the next iterations follow the reference trace
    _, meta, iters, _ = parity.load_case("singlet_cT")
    cont = [a for a in _argv("singlet_cT", str(tmp_path / "cont")) if a]
    if "--maxIterations" in cont:
        k = cont.index("--maxIterations")
        del cont[k:k + 2]
    run.solve(cont + ["--maxIterations", "3", "-i", str(out)])
    with open(tmp_path / "cont" / "iterations.json") as f:
        got = json.load(f)
    for g, w in zip(got, iters[len(done):len(done) + 3]):
        bad, _ = parity.compare_iteration(g, w, tol_bits=99)
        assert not bad, bad


def _checkpoint_restart(tmp_path, lib):
    """--writeSolution=x,y,X,Y then -i <dir>: 3 + 3 iterations equal 6 iterations
    up to the decimal round trip of the checkpoint files."""
    base = _argv("1d-constraints", str(tmp_path / "full"), lib)
    it = base.index("--maxIterations") if "--maxIterations" in base else None
    def with_max(argv, n):
        argv = list(argv)
        if "--maxIterations" in argv:
            k = argv.index("--maxIterations")
            del argv[k:k + 2]
        return argv + ["--maxIterations", str(n)]
    run.solve(with_max(base, 6))
    first = with_max(_argv("1d-constraints", str(tmp_path / "a"), lib), 3)
    run.solve(first)
    second = with_max(_argv("1d-constraints", str(tmp_path / "b"), lib), 3) + ["-i", str(tmp_path / "a")]
    run.solve(second)
    with open(tmp_path / "full" / "iterations.json") as f:
        full = json.load(f)
    with open(tmp_path / "b" / "iterations.json") as f:
        cont = json.load(f)
    for g, w in zip(cont, full[3:6]):
        bad, _ = parity.compare_iteration(g, w, tol_bits=600)
        assert not bad, bad


def test_options_are_parsed_like_sdpb_parses_them():
    """64-bit GMP parse (SDPB_Parameters.cxx runs before El::gmp::SetPrecision): compare with real GMP."""
    from fractions import Fraction
    from oracle.oracle import Oracle
    sdp, _, _, _ = parity.load_case("1d")
    o = Oracle(sdp, 128)
    for text in list(parity.DEFAULT_PARAMS.values()) + ["1e-10", "1e10", "1e30", "1e-153", "1.0e-30", "1.0e20",
                                                        "12345.678e-3", "-0.7", "1e-57", "1e57"]:
        assert Fraction(run.parse_option_like_sdpb(text)) == Fraction(o.parse_exact(text, 64)), text
    # beyond 10^57 GMP truncates the power of ten as well: agreement to 2^-128 only
    for text in ["1.0e-200", "1e300"]:
        a, b = Fraction(run.parse_option_like_sdpb(text)), Fraction(o.parse_exact(text, 64))
        assert abs(a - b) < b / 2 ** 128, text
    o.close()


def test_z_from_y_inserts_normalized_component():
    y = ["0.5", "-2"]
    nrm = ["1", "4", "2"]
    z = run._z_from_y(y, nrm, 128)
    assert len(z) == 3
    total = sum(parity.mpmath.mpf(a) * parity.mpmath.mpf(b) for a, b in zip(nrm, z))
    assert abs(total - 1) < parity.mpmath.mpf(2) ** -120
    assert parity.mpmath.mpf(z[0]) == parity.mpmath.mpf("0.5") and parity.mpmath.mpf(z[2]) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["1d", "1d-constraints", "dfibo"])
def test_driver_writes_sdpb_result_files_gpu(name, tmp_path):
    out_dir = str(tmp_path / "out")
    run.solve(_argv(name, out_dir))
    _check_outputs(name, out_dir)
