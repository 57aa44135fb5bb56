"""The N>1 path on CPU: two (and three) processes, torch.distributed `gloo`, the emulation build
of the library (its "device" pointers are host pointers).  Blocks are sharded across the ranks; the
fixed-point Q' image is summed with an integer all-reduce and the small vectors are all-gathered,
exactly as on RCCL.  Results must still match the reference trace.  Same worker and the same checks
as the device test (tests/test_multirank_gpu.py), which runs the ranks on one MI355X."""
import pytest

from tests import parity
from tests.test_multirank_gpu import _load, check_ranks, run_ranks


@pytest.mark.parametrize("name,n_iter,world", [("1d-constraints", 4, 2), ("dfibo", 3, 2), ("dfibo", 2, 3)])
def test_sharded_iteration_matches_reference(name, n_iter, world):
    sdp, _, _, _, iters = _load(name)
    results = run_ranks(world, name, n_iter, timeout=600, gpu=False)
    check_ranks(results, world, sdp.J)
    for got, want in zip(results[0][2], iters):
        bad, _ = parity.compare_iteration(got, want)
        assert not bad, (want["iteration"], bad)


@pytest.mark.parametrize("name,n_iter,world", [("singlet_cT", 3, 2), ("dfibo", 3, 3)])
def test_cholesky_Q_distributed_over_the_ranks(name, n_iter, world):
    """The 1-D block-cyclic Cholesky(Q) (one broadcast per column panel, every rank updates the panels it
    owns) on the emulation build with 4-column panels: N = 20 / 19 gives five panels dealt to 2 / 3 ranks.
    Same trace as the reference, ranks bit-identical."""
    sdp, _, _, _, iters = _load(name)
    results = run_ranks(world, name, n_iter, timeout=900, gpu=False, env={"SDPB_HIP_DIST_CHOLQ": "1"}, emu_panel=4)
    check_ranks(results, world, sdp.J, "distributed", -(-sdp.N // 4))
    for got, want in zip(results[0][2], iters):
        bad, _ = parity.compare_iteration(got, want)
        assert not bad, (want["iteration"], bad)


@pytest.mark.parametrize("name,n_iter,world", [("singlet_cT", 3, 2), ("dfibo", 3, 3)])
def test_chased_cholesky_Q_over_the_ranks(name, n_iter, world):
    """Q' in two column chunks (two all-reduces) with the replicated Cholesky(Q) chasing it (SDPB_HIP_Q_CHASE=1; opt-in, a
    measured loss on the hardware: DESIGN section 9) on the 4-column-panel build: N = 20 / 19 = five panels, the left four
    (16 columns = one tile column) are factored before the right one is reduced.  Same trace as the reference, ranks
    bit-identical, and the same bits as the one-piece schedule."""
    sdp, _, _, _, iters = _load(name)
    results = run_ranks(world, name, n_iter, timeout=900, gpu=False, env={"SDPB_HIP_Q_CHASE": "1"}, emu_panel=4)
    check_ranks(results, world, sdp.J)
    assert all(r[3]["comm.q_chase"] == 1 and r[3]["comm.allreduce_calls"] == 2 * n_iter for r in results)
    for got, want in zip(results[0][2], iters):
        bad, _ = parity.compare_iteration(got, want)
        assert not bad, (want["iteration"], bad)
    plain = run_ranks(world, name, n_iter, timeout=900, gpu=False, env={"SDPB_HIP_Q_CHASE": "0"}, emu_panel=4)
    assert plain[0][3]["comm.q_chase"] == 0 and plain[0][3]["comm.allreduce_calls"] == n_iter
    assert plain[0][2] == results[0][2]


def test_ranks_that_fall_out_of_step_fail_alike():
    """The collective-sequence self-check (kernels.hpp: XW_SEQ_LO; include/sdpb_hip.h: sdpb_hip_progress): rank 1's
    hash of the (kind, bytes, root) sequence is perturbed (SDPB_HIP_TEST_SEQ_FAULT=1 — the collectives themselves stay
    matched, so nothing hangs); at the first synchronisation point BOTH ranks must raise the same code-3 error instead
    of iterating on."""
    results = run_ranks(2, "1d-constraints", 2, timeout=600, gpu=False, env={"SDPB_HIP_TEST_SEQ_FAULT": "1"})
    errs = [r[2][-1].get("error") for r in results]
    assert errs[0] is not None and errs[0] == errs[1], errs
    assert errs[0][0] == 3 and "collective sequence mismatch: rank 1 and rank 0" in errs[0][1], errs[0]
    assert all(len(r[2]) == 1 for r in results)          # raised at synchronisation point 1 of iteration 1
