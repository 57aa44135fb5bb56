"""The N>1 path on CPU: two processes, torch.distributed `gloo`, the emulation build of
the library (its "device" pointers are host pointers).  Blocks are sharded across the two
ranks; the fixed-point Q' image is summed with an integer all-reduce and the small vectors
are all-gathered, exactly as on RCCL.  Results must still match the reference trace."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import libs, parity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, n_iter, lib, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, libs.ROOT)
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sdpb_amd.distributed import make_collectives
        from sdpb_amd.solver import SDPSolver
        sdp, meta, iters, out = parity.load_case(name)
        s = SDPSolver(sdp, meta["precision"], meta["params"], rank=rank, world_size=world, lib_path=lib)
        s.set_collectives(*make_collectives(torch.device("cpu")))
        owners = [s.block_owner(j) for j in range(sdp.J)]
        problems = []
        for rec in iters[:n_iter]:
            if s.iterate():
                problems.append(("terminated early", rec["iteration"], s.terminate_reason))
                break
            bad, _ = parity.compare_iteration(s.scalars(), rec)
            if bad:
                problems.append((rec["iteration"], bad))
        q.put((rank, owners, problems, s.scalars()))
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,n_iter", [("1d-constraints", 4), ("dfibo", 3)])
def test_two_rank_sharded_iteration_matches_reference(name, n_iter):
    lib = libs.emu_lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, n_iter, lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, owners0, prob0, sc0), (r1, owners1, prob1, sc1) = results
    assert owners0 == owners1 and set(owners0) == {0, 1}   # same plan everywhere, both ranks own blocks
    assert not prob0 and not prob1, (prob0, prob1)
    assert sc0 == sc1                                      # ranks stay in lock-step bit for bit
