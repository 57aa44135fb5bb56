"""Pre-flight of the in-library RCCL exchange with more than one rank, in short-lived child processes.

The library's own communicator (csrc/rccl_comm.hpp: ncclAllGather / ncclAllReduce / ncclBroadcast on the
library's streams) replaces the El::mpi collectives of the reference's step (restore_and_reduce.cxx:137-212,
initialize_schur_complement_solver.cxx:95-103).  A launcher that is about to hand the iteration to it first lets
every rank run `sdpb_hip_rccl_preflight` (include/sdpb_hip.h) in a CHILD process under a timeout: a bootstrap that
cannot connect, a collective that never completes or wrong bytes then cost a few seconds and a clear record
instead of a hung job, and the launcher can fall back to its own collectives (sdpb_hip_set_collectives).

Child:   python -m sdpb_amd.rccl_preflight --rank R --world W --device D [--id HEX] [--bytes B]
         rank 0 without --id creates the id and prints "ID <hex>" first; every child ends with "PREFLIGHT OK {...}".
Parent:  run(rank, world, device, exchange_id, timeout) -> record; exchange_id(hex or None) -> hex is the launcher's
         way to move 128 bytes from rank 0 to everybody (torch.distributed, MPI_Bcast, a file).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_gpu_env(rank: int) -> dict:
    """Environment that lets several RCCL ranks share ONE GPU (test boxes with a single device).
    RCCL refuses two ranks whose (host hash, PCI bus id) coincide ("Duplicate GPU detected").  NCCL_HOSTID replaces
    the host hash: with one id per rank the ranks look like one-GPU hosts, the check passes and RCCL connects them with
    its net/Socket transport over loopback (profiles/r04_rccl_two_ranks_one_gpu.log).  Not xGMI -- no bandwidth claim
    follows from such a run -- but every ncclAllReduce / ncclAllGather / ncclBroadcast of csrc/rccl_comm.hpp executes
    with world > 1 on the device.  Must be in os.environ before the communicator is created."""
    return {"NCCL_HOSTID": f"sdpb-rank-{rank}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_NET": "Socket", "NCCL_IB_DISABLE": "1",
            "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1"}


def _child(args) -> int:
    from sdpb_amd.solver import load_library
    import torch  # the library binds to torch's HIP runtime / RCCL (solver.load_library imports it first as well)
    torch.cuda.set_device(args.device)
    torch.cuda.synchronize()
    L = load_library(args.lib)
    L.sdpb_hip_rccl_preflight.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
    L.sdpb_hip_last_error.restype = ctypes.c_char_p
    if args.id:
        uid = bytes.fromhex(args.id)
    else:
        buf = ctypes.create_string_buffer(128)
        if L.sdpb_hip_rccl_unique_id(buf) != 0:
            print("PREFLIGHT FAILED unique id: " + L.sdpb_hip_last_error(None).decode(), flush=True)
            return 3
        uid = buf.raw
        print("ID " + uid.hex(), flush=True)
    t0 = time.time()
    rc = L.sdpb_hip_rccl_preflight(uid, args.rank, args.world, args.bytes)
    if rc != 0:
        print("PREFLIGHT FAILED " + L.sdpb_hip_last_error(None).decode(), flush=True)
        return 3
    print("PREFLIGHT OK " + json.dumps({"rank": args.rank, "world": args.world, "bytes": args.bytes,
                                        "seconds": round(time.time() - t0, 3)}), flush=True)
    return 0


def run(rank: int, world: int, device: int, exchange_id, timeout: float = 120.0, nbytes: int = 64 << 20,
        lib_path: str | None = None, extra_env: dict | None = None) -> dict:
    """One rank's part of the pre-flight: start the child, move rank 0's id through `exchange_id`, wait.
    Returns {"ok": bool, "seconds": s, "detail": str}.  Never raises for a failing or hanging child."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):   # the child is not a torchrun worker
        env.pop(k, None)
    cmd = [sys.executable, "-m", "sdpb_amd.rccl_preflight", "--rank", str(rank), "--world", str(world),
           "--device", str(device), "--bytes", str(nbytes)]
    if lib_path:
        cmd += ["--lib", lib_path]
    t0 = time.time()
    deadline = t0 + timeout
    child, hexid, lines = None, None, []
    import queue
    import threading
    q: "queue.Queue[str | None]" = queue.Queue()

    def _start(argv):
        """The child with ONE reader of its pipe: a thread that drains stdout to EOF into the queue (round-5 advisor: the
        earlier code read the first line through the TextIOWrapper and the rest with communicate(), which reads the raw
        descriptor -- lines already buffered behind the id, a 'PREFLIGHT OK' among them, were lost, and a reader still
        blocked in readline() shared the pipe with communicate())."""
        c = subprocess.Popen(argv, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

        def _reader(pipe=c.stdout):
            for ln in iter(pipe.readline, ""):
                q.put(ln)
            q.put(None)                 # EOF

        threading.Thread(target=_reader, daemon=True).start()
        return c

    def _next_line():
        """The next line of the child, None at EOF, "" when the deadline passes first."""
        try:
            return q.get(timeout=max(0.05, deadline - time.time()))
        except queue.Empty:
            return ""

    try:
        eof = False
        if rank == 0:
            child = _start(cmd)
            # The first lines lead to the id (or a failure).  readline() has no timeout, so THIS thread waits on the queue
            # with one (round-4 advisor: a child stalled before "ID ..." -- torch import, HIP initialisation,
            # ncclGetUniqueId -- would otherwise block rank 0 here and every other rank in exchange_id).
            while True:
                line = _next_line()
                if line == "":
                    lines.append(f"no id within {timeout:.0f} s")
                    child.kill()
                    break
                if line is None:
                    eof = True
                    break
                lines.append(line.strip())
                if line.startswith("ID "):
                    hexid = line.split()[1]
                    break
        hexid = exchange_id(hexid)              # collective over the launcher's own transport; None = rank 0 has none
        if hexid is None:
            if child:
                child.kill()
            return {"ok": False, "seconds": time.time() - t0, "detail": "rank 0 produced no id: " + " | ".join(lines[-3:])}
        if rank != 0:
            child = _start(cmd + ["--id", hexid])
        while not eof:                           # the rest of the output, to EOF or to the deadline
            line = _next_line()
            if line == "":
                child.kill()
                return {"ok": False, "seconds": time.time() - t0, "detail": f"timed out after {timeout:.0f} s: " + " | ".join(lines[-3:])}
            if line is None:
                break
            lines.append(line.strip())
        try:
            child.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            child.kill()
            return {"ok": False, "seconds": time.time() - t0, "detail": f"no exit within {timeout:.0f} s: " + " | ".join(lines[-3:])}
        lines = [l for l in lines if l]
        ok = child.returncode == 0 and any(l.startswith("PREFLIGHT OK") for l in lines)
        return {"ok": ok, "seconds": time.time() - t0, "detail": (lines[-1] if lines else f"exit code {child.returncode}")[:400]}
    except Exception as e:                      # pragma: no cover - the pre-flight must never take the job down
        if child and child.poll() is None:
            child.kill()
        return {"ok": False, "seconds": time.time() - t0, "detail": f"{type(e).__name__}: {e}"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--id", default=None)
    ap.add_argument("--bytes", type=int, default=64 << 20)
    ap.add_argument("--lib", default=None)
    return _child(ap.parse_args())


if __name__ == "__main__":
    sys.exit(main())
