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
    if case.startswith("C5J"):
        # C5 (m = 6, two sample points, N = 2048, 1024 bits) with J blocks and the FULL N: "C5J4096" is half of C5, what four
        # ranks of an 8-rank job hold together (profiles/tools/c5_half_multirank.py)
        from sdpb_amd import synthetic
        J = int(case[3:])
        c = synthetic.config("C5")
        sdp, src = synthetic.make_lazy([6] * J, [2] * J, c["N"], c["precision"], c["seed"])
        return sdp, c["precision"], dict(parity.DEFAULT_PARAMS), src, None
    if case.startswith("C"):
        from sdpb_amd import synthetic
        name, _, scale = case.partition("x")
        c = synthetic.config(name, float(scale) if scale else 1.0)
        sdp, src = synthetic.make_lazy(c["dims"], c["num_points"], c["N"], c["precision"], c["seed"], feasible=c.get("feasible", False))
        return sdp, c["precision"], dict(parity.DEFAULT_PARAMS), src, None
    sdp, meta, iters, _ = parity.load_case(case)
    return sdp, meta["precision"], meta["params"], None, iters


def rccl_one_gpu_env(rank):
    from sdpb_amd.rccl_preflight import one_gpu_env
    return one_gpu_env(rank)


def _worker(rank, world, port, case, n_iter, q, gpu=True, env=None, emu_panel=None, transport="callbacks"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.update(env or {})
    if transport == "rccl-one-gpu":
        os.environ.update(rccl_one_gpu_env(rank))
    sys.path.insert(0, libs.ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sdpb_amd.distributed import make_collectives
        from sdpb_amd.solver import SDPSolver
        # transport "rccl": one rank per GPU and the library's own communicator (needs `world` devices);
        # "callbacks": the ranks share device 0 and exchange through host-staged gloo
        rccl, transport = transport.startswith("rccl"), "rccl" if transport.startswith("rccl") else transport
        devno = rank if rccl and os.environ.get("NCCL_HOSTID") is None else 0
        if gpu:
            torch.cuda.set_device(devno)
        dev = torch.device("cuda", devno) if gpu else torch.device("cpu")
        sdp, precision, params, src, _ = _load(case)
        # gpu=False: the CPU twin of this test (tests/test_multirank.py) on the emulation build
        s = SDPSolver(sdp, precision, params, device=devno, rank=rank, world_size=world,
                      lib_path=libs.product_lib() if gpu else libs.emu_lib(panel=emu_panel), upload_all_blocks=False,
                      block_source=src)
        if transport == "rccl":
            box = [s.rccl_unique_id().hex() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            s.rccl_init(bytes.fromhex(box[0]))
        else:
            s.set_collectives(*make_collectives(dev))
        owners = [s.block_owner(j) for j in range(sdp.J)]
        from sdpb_amd.solver import SDPBError
        recs, secs = [], []
        import time as _t
        for _ in range(n_iter):
            t_it = _t.time()
            try:
                if s.iterate():
                    recs.append({"terminated": s.terminate_reason})
                    break
            except SDPBError as e:   # the ranks must fail alike: the error is part of the record that is compared
                recs.append({"error": (e.code, str(e))})
                break
            recs.append(s.scalars())
            secs.append(round(_t.time() - t_it, 3))
        t = s.timers()
        comm = {k: v for k, v in t.items() if k.startswith("comm.")}
        comm["progress"] = s.progress()
        comm["memory_plan"] = s.memory_plan()
        comm["seconds_per_iteration"] = secs
        q.put((rank, owners, recs, comm, s.comm_name))
        s.close()
    finally:
        dist.destroy_process_group()


def run_ranks(world, case, n_iter, timeout=900, gpu=True, env=None, emu_panel=None, transport="callbacks"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, n_iter, q, gpu, env, emu_panel, transport)) for r in range(world)]
    for p in procs:
        p.start()
    # collect the records; a rank that dies (an exception in the worker, a device out of memory) ends the wait at once
    # instead of leaving the others in a collective until the timeout
    import queue as _queue
    import time as _time
    results, deadline = [], _time.time() + timeout
    while len(results) < world:
        try:
            results.append(q.get(timeout=2.0))
        except _queue.Empty:
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead or _time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"ranks died (rank, exit code): {dead}" if dead else f"no result within {timeout} s")
    results.sort()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def check_ranks(results, world, J, cholesky_Q="replicated", panels=None, transport="callbacks"):
    owners0 = results[0][1]
    for rank, owners, recs, comm, name in results:
        assert owners == owners0                       # the same plan on every rank
        transport = "rccl" if transport.startswith("rccl") else transport
        assert name == transport
        if transport == "rccl":
            assert comm["comm.ranks"] == world         # ncclCommCount
        assert comm["comm.world"] == world
        assert comm["comm.owned_blocks"] == owners0.count(rank) and 0 < comm["comm.owned_blocks"] < J
        # per iteration: one Q' all-reduce, three result-block all-gathers + three N-vector all-gathers
        assert comm["comm.allreduce_calls"] >= len(recs) and comm["comm.allgather_calls"] >= 6 * len(recs)
        assert recs == results[0][2], f"rank {rank} diverged from rank 0"   # lock-step, bit for bit
        assert comm["comm.cholesky_Q"] == cholesky_Q
        # the collective-sequence self-check: same number of collectives, same (kind, bytes, root) hash everywhere
        pr, pr0 = comm["progress"], results[0][3]["progress"]
        assert pr["collectives"] == pr0["collectives"] == comm["comm.collectives"] > 0
        assert pr["sequence_hash"] == pr0["sequence_hash"] == comm["comm.sequence_hash"].rjust(16, "0")
        assert pr["iteration"] == len(recs) and pr["host_syncs"] >= 3 * (len(recs) - 1) + 1 and pr["transport_async_error"] == 0
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
@pytest.mark.parametrize("transport", ["callbacks", "rccl-one-gpu"])
def test_two_ranks_follow_the_run_to_optimality(transport):
    """The strictly feasible fixture (C4f x0.25: J = 150, N = 250, 159 oracle iterations before 'found primal-dual optimal
    solution') on two ranks sharing the GPU: the end game of a convergent run -- step lengths from spectra that collapse onto
    one point, reduced over the ranks; block condition numbers past 2^(p/2) -- with the cross-rank sums in the loop.  Every
    iteration against the committed oracle record at the bar of the one-rank fixture test (2^-(p/2), conditioned past
    cond = 2^(p/2-16)); ranks bit-identical."""
    import json
    with open(os.path.join(parity.GOLDEN, "synthetic", "C4f_x0.25_to_termination.json")) as f:
        fx = json.load(f)
    sdp, precision, params, src, _ = _load("C4fx0.25")
    assert sdp.J == fx["J"] and sdp.N == fx["N"] and precision == fx["precision"]
    results = run_ranks(2, "C4fx0.25", len(fx["iterations"]), timeout=1500, transport=transport)   # host callbacks / in-library RCCL
    check_ranks(results, 2, sdp.J, transport=transport)
    worst = float("-inf")
    for got, rec in zip(results[0][2], fx["iterations"]):
        bad, w = parity.compare_iteration(got, rec, tol_bits=parity.conditioned_tol_bits(rec, precision, precision // 2))
        worst = max(worst, w)
        assert not bad, (rec["iteration"], bad)
    print(f"two ranks, {len(fx['iterations'])} iterations of the run to optimality: worst log2 rel diff {worst:.1f}")


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


@pytest.mark.gpu
def test_two_C5_slices_over_the_in_library_rccl_exchange():
    """The largest piece of BASELINE.json's config 5 a test run can afford: C5J2048 = a QUARTER of C5 (J = 2048 blocks of
    m = 6, the full N = 2048, --precision 1024, P_tot = 86 016), two ranks that each hold exactly what a rank of the 8-GPU
    job holds (a C5 slice: 58 GB of device arrays) and exchange through the library's own RCCL communicator -- the
    1.1-GB u64 all-reduce of Q' (restore_and_reduce.cxx:137-212), the result blocks and the N-vectors.  The ranks share
    the box's one GPU (NCCL_HOSTID, socket transport), so this says nothing about xGMI.  Checked: ranks bit-identical,
    owners a partition, collective sequences equal, every rank's syrk partial planes inside its memory bound
    (sdpb_hip_memory_plan), and every field of every iteration within 2^-900 of the SAME SDP solved by one rank (the two
    decompositions differ in the order of the cross-rank sums only).  profiles/tools/c5_half_multirank.py is the same
    at twice the size with 4 and 2 ranks."""
    from sdpb_amd.solver import SDPSolver
    case, n_iter, bound = "C5J2048", 2, 12 << 30
    sdp, precision, params, src, _ = _load(case)
    res = run_ranks(2, case, n_iter, timeout=2400, env={"SDPB_HIP_SYRK_PART_BYTES": str(bound)}, transport="rccl-one-gpu")
    check_ranks(res, 2, sdp.J, transport="rccl-one-gpu")
    for rank, owners, recs, comm, name in res:
        plan = comm["memory_plan"]
        assert plan["owned_blocks"] == sdp.J // 2 and plan["rows"] == sdp.P_total // 2
        assert plan["syrk"]["partial_bytes"] <= bound < plan["syrk"]["partial_bytes_unbounded"] and plan["syrk"]["chunks"] >= 2, plan["syrk"]
        assert plan["bytes"]["syrk_partial_planes"] <= bound
        assert comm["comm.allreduce_bytes"] / n_iter > 1.0e9     # the lower triangle of Q' + column sums, 66 u64 planes
    one = SDPSolver(sdp, precision, params, lib_path=libs.product_lib(), block_source=src)
    for it in range(n_iter):
        assert not one.iterate()
        bad, worst = parity.compare_iteration(res[0][2][it], one.scalars(), tol_bits=900)
        assert not bad, (it + 1, bad)
    one.close()


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


# ---- the in-library RCCL communicator: one rank PER GPU wherever the box has the devices; on the 1-GPU test box the
# ---- ranks share device 0 (compute partitions cannot be changed from inside the container,
# ---- profiles/r04_partition_probe.txt) and RCCL is made to accept them through NCCL_HOSTID (rccl_preflight.one_gpu_env)
@pytest.mark.gpu
@pytest.mark.parametrize("world,case,n_iter,dist_cholq", [(2, "dfibo", 3, False), (2, "C4x0.25", 3, False), (2, "C4x0.25", 3, True),
                                                          (4, "C4x0.25", 3, False), (4, "C4x0.25", 3, True), (8, "C4x0.25", 2, True)])
def test_in_library_rccl_with_more_than_one_rank(world, case, n_iter, dist_cholq):
    """restore_and_reduce.cxx:137-212 and the distributed El::Cholesky of initialize_schur_complement_solver.cxx:95-103
    on the production transport: ncclAllReduce / ncclAllGather / ncclBroadcast inside the library, ranks bit-identical,
    golden trace (dfibo) or live oracle (C4 x0.25)."""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    # one rank per GPU where the box has them; otherwise the ranks share device 0 and RCCL connects them over its
    # socket transport (rccl_one_gpu_env)
    transport = "rccl" if _gpus() >= world else "rccl-one-gpu"
    sdp, precision, params, src, iters = _load(case)
    env = {"SDPB_HIP_DIST_CHOLQ": "1"} if dist_cholq else None
    results = run_ranks(world, case, n_iter, env=env, transport=transport)
    check_ranks(results, world, sdp.J, "distributed" if dist_cholq else "replicated", -(-sdp.N // 32), transport="rccl")
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
def test_rccl_preflight_child_process():
    """sdpb_amd/rccl_preflight.py (what bench.py runs before it trusts the in-library exchange): with one rank on the
    one GPU the test box has, and with one rank per GPU wherever there are more."""
    from sdpb_amd import rccl_preflight
    rec = rccl_preflight.run(0, 1, 0, lambda h: h, timeout=300, nbytes=1 << 20)
    assert rec["ok"], rec
    n = _gpus()
    share = n < 2          # the ranks share device 0 (rccl_preflight.one_gpu_env)
    if True:
        import threading
        world = 2 if share else min(n, 8)
        box, recs = {}, [None] * world
        have = threading.Event()

        def xid(r):
            def f(h):
                if r == 0:
                    box["id"] = h
                    have.set()
                have.wait(300)
                return box.get("id")
            return f

        def one(r):
            recs[r] = rccl_preflight.run(r, world, 0 if share else r, xid(r), timeout=300, nbytes=16 << 20,
                                         extra_env=rccl_preflight.one_gpu_env(r) if share else None)
        ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert all(r and r["ok"] for r in recs), recs


if __name__ == "__main__":
    world, case, n_iter = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    transport = sys.argv[4] if len(sys.argv) > 4 else "callbacks"     # callbacks | rccl | rccl-one-gpu
    dist_q = os.environ.get("SDPB_HIP_DIST_CHOLQ") == "1"
    res = run_ranks(world, case, n_iter, transport=transport)
    sdp = _load(case)[0]
    check_ranks(res, world, sdp.J, "distributed" if dist_q else "replicated", -(-sdp.N // 32), transport=transport)
    for rank, owners, recs, comm, name in res:
        print(f"rank {rank}: {comm} P-obj={recs[-1].get('P-obj', '')[:40]}")
