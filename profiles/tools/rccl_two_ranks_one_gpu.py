"""Probe: can the in-library RCCL communicator run with TWO ranks on the ONE GPU of the test box?
RCCL refuses two ranks whose (hostHash, busId) coincide ("Duplicate GPU detected").  NCCL_HOSTID overrides the host
hash: giving every rank its own makes the ranks look like different hosts, so the duplicate check passes and the
transport becomes net/Socket over loopback -- not xGMI, but every ncclAllReduce / ncclAllGather / ncclBroadcast of
csrc/rccl_comm.hpp then executes with world > 1 (device kernels, stream order, the unique-id hand-off).
    python profiles/tools/rccl_two_ranks_one_gpu.py [world]
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2

def env(r, debug):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_HOSTID=f"sdpb-rank-{r}", NCCL_SOCKET_IFNAME="lo",
             NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
    if debug:
        e["NCCL_DEBUG"] = "INFO"
    return e

cmd = lambda r: [sys.executable, "-m", "sdpb_amd.rccl_preflight", "--rank", str(r), "--world", str(world), "--device", "0", "--bytes", str(16 << 20)]
t0 = time.time()
p0 = subprocess.Popen(cmd(0), cwd=ROOT, env=env(0, True), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
hexid = None
for line in p0.stdout:
    print("[0]", line.rstrip())
    if line.startswith("ID "):
        hexid = line.split()[1]; break
if hexid is None:
    sys.exit("rank 0 gave no id")
others = [subprocess.Popen(cmd(r) + ["--id", hexid], cwd=ROOT, env=env(r, False), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(1, world)]
ok = True
try:
    out, _ = p0.communicate(timeout=240)
    print("\n".join("[0] " + l for l in out.splitlines()[-60:]))
    ok &= "PREFLIGHT OK" in out
    for r, p in enumerate(others, 1):
        o, _ = p.communicate(timeout=60)
        print("\n".join(f"[{r}] " + l for l in o.splitlines()[-8:]))
        ok &= "PREFLIGHT OK" in o
except subprocess.TimeoutExpired:
    ok = False
    print("TIMEOUT")
    for p in [p0] + others:
        p.kill()
print("RESULT", "OK" if ok else "FAILED", f"{time.time()-t0:.1f}s")
sys.exit(0 if ok else 1)
