"""The world_size > 1 DEVICE path on the hardware a 1-GPU box has: 2 and 4 processes share
device 0, each a rank of the job (its own HIP context, its own shard of the SDP blocks).  RCCL
refuses two ranks on one GPU, so the exchange goes through the C ABI's collective callbacks
(sdpb_hip_set_collectives) over torch.distributed `gloo`, staged through host memory
(sdpb_amd/distributed.py).  Everything else is the product path: the block plan, Jl < J on every
rank, the u64 widening / carry kernels around the Q' all-reduce (k_widen_tri_u64,
k_narrow_tri_carry; reference: restore_and_reduce.cxx:137-212), the rank-order combines of result
blocks and N-vectors (k_combine_slots, k_combine_vec; reference: the El::mpi::AllReduce sites and
solve_schur_complement_equation.cxx:57-66).

Checked: every rank against the reference's golden trace (2^-99) or the live oracle (2^-(p/2));
ranks bit-identical with each other; owners a partition of the blocks with every rank non-empty.

    python -m tests.test_multirank_gpu 2 C4x0.25 3     # the same run outside pytest (rocprofv3 wraps this)
"""
import os
import socket
import sys

import pytest

from tests import libs, parity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load(case):
    """-> (sdp, precision, params, block_source or None, golden iterations or None)"""
    if case.startswith("C"):
        from sdpb_amd import synthetic
        name, _, scale = case.partition("x")
        c = synthetic.config(name, float(scale) if scale else 1.0)
        sdp, src = synthetic.make_lazy(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"])
        return sdp, c["precision"], dict(parity.DEFAULT_PARAMS), src, None
    sdp, meta, iters, _ = parity.load_case(case)
    return sdp, meta["precision"], meta["params"], None, iters


def _worker(rank, world, port, case, n_iter, q, gpu=True, env=None, emu_panel=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.update(env or {})
    sys.path.insert(0, libs.ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sdpb_amd.distributed import make_collectives
        from sdpb_amd.solver import SDPSolver
        if gpu:
            torch.cuda.set_device(0)
        dev = torch.device("cuda", 0) if gpu else torch.device("cpu")
        sdp, precision, params, src, _ = _load(case)
        # gpu=False: the CPU twin of this test (tests/test_multirank.py) on the emulation build
        s = SDPSolver(sdp, precision, params, device=0, rank=rank, world_size=world,
                      lib_path=libs.product_lib() if gpu else libs.emu_lib(panel=emu_panel), upload_all_blocks=False,
                      block_source=src)
        s.set_collectives(*make_collectives(dev))
        owners = [s.block_owner(j) for j in range(sdp.J)]
        recs = []
        for _ in range(n_iter):
            if s.iterate():
                recs.append({"terminated": s.terminate_reason})
                break
            recs.append(s.scalars())
        t = s.timers()
        q.put((rank, owners, recs, {k: v for k, v in t.items() if k.startswith("comm.")}, s.comm_name))
        s.close()
    finally:
        dist.destroy_process_group()


def run_ranks(world, case, n_iter, timeout=900, gpu=True, env=None, emu_panel=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, n_iter, q, gpu, env, emu_panel)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def check_ranks(results, world, J, cholesky_Q="replicated", panels=None):
    owners0 = results[0][1]
    for rank, owners, recs, comm, name in results:
        assert owners == owners0                       # the same plan on every rank
        assert name == "callbacks"
        assert comm["comm.world"] == world
        assert comm["comm.owned_blocks"] == owners0.count(rank) and 0 < comm["comm.owned_blocks"] < J
        # per iteration: one Q' all-reduce, three result-block all-gathers + three N-vector all-gathers
        assert comm["comm.allreduce_calls"] >= len(recs) and comm["comm.allgather_calls"] >= 6 * len(recs)
        assert recs == results[0][2], f"rank {rank} diverged from rank 0"   # lock-step, bit for bit
        assert comm["comm.cholesky_Q"] == cholesky_Q
        if cholesky_Q == "distributed":   # one broadcast per column panel of Q and iteration
            assert comm["comm.broadcast_calls"] == panels * len(recs), (comm["comm.broadcast_calls"], panels, len(recs))
    assert sorted(set(owners0)) == list(range(world)) and len(owners0) == J  # a partition, nobody idle


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_one_gpu_match_the_reference_trace(world):
    """dfibo: J = 98 blocks (K in {1, 4}, empty odd parities), N = 19, the reference's golden trace."""
    sdp, _, _, _, iters = _load("dfibo")
    results = run_ranks(world, "dfibo", len(iters))
    check_ranks(results, world, sdp.J)
    for got, want in zip(results[0][2], iters):
        bad, _ = parity.compare_iteration(got, want)
        assert not bad, (want["iteration"], bad)


@pytest.mark.gpu
@pytest.mark.parametrize("world,case,n_iter", [(2, "C4x0.25", 3), (3, "dfibo", 3)])
def test_cholesky_Q_distributed_over_the_ranks_on_the_device(world, case, n_iter):
    """Cholesky(Q) factored over the ranks (1-D block-cyclic column panels, one broadcast per panel;
    reference: initialize_schur_complement_solver.cxx:95-103 factors Q over COMM_WORLD) instead of
    replicated: C4 x0.25 (N = 250, eight panels) against the live oracle, dfibo (one panel) against
    the golden trace; ranks bit-identical."""
    sdp, precision, params, src, iters = _load(case)
    results = run_ranks(world, case, n_iter, env={"SDPB_HIP_DIST_CHOLQ": "1"})
    check_ranks(results, world, sdp.J, "distributed", -(-sdp.N // 32))
    if iters is not None:
        for got, want in zip(results[0][2], iters):
            bad, _ = parity.compare_iteration(got, want)
            assert not bad, (want["iteration"], bad)
        return
    from oracle.oracle import Oracle
    o = Oracle(sdp, precision, params, param_prec=0, block_source=src)
    for it in range(n_iter):
        assert not o.iterate()
        bad, _ = parity.compare_iteration(results[0][2][it], o.scalars(), tol_bits=precision // 2)
        assert not bad, (it + 1, bad)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_one_gpu_match_the_oracle_at_bench_shape(world):
    """C4 x0.25 (J = 150 mixed m = 1 / m = 2 blocks, N = 250: eight panels of Cholesky(Q), a 63 000-entry
    lower triangle of Q' through the u64 all-reduce) against the live oracle at 2^-(p/2)."""
    from oracle.oracle import Oracle
    n_iter = 3
    sdp, precision, params, src, _ = _load("C4x0.25")
    results = run_ranks(world, "C4x0.25", n_iter)
    check_ranks(results, world, sdp.J)
    o = Oracle(sdp, precision, params, param_prec=0, block_source=src)
    for it in range(n_iter):
        assert not o.iterate()
        bad, worst = parity.compare_iteration(results[0][2][it], o.scalars(), tol_bits=precision // 2)
        assert not bad, (it + 1, bad)
    o.close()


if __name__ == "__main__":
    world, case, n_iter = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    res = run_ranks(world, case, n_iter)
    sdp = _load(case)[0]
    check_ranks(res, world, sdp.J)
    for rank, owners, recs, comm, name in res:
        print(f"rank {rank}: {comm} P-obj={recs[-1].get('P-obj', '')[:40]}")
