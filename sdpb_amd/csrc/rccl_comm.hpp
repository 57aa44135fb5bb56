// rccl_comm.hpp — the cross-GPU exchange on RCCL over xGMI, inside the library.
//
// Replaces the El::mpi collectives of the reference's step:
//   * the sum of the fixed-point Q' images: one ncclAllReduce(ncclSum) on 64-bit unsigned
//     lanes (restore_and_reduce.cxx:137-212 reduces residues over MPI);
//   * the all-gathers of result blocks and N-vectors that every rank then combines in
//     rank order (El::mpi::AllReduce of scalars / dy, compute_search_direction.cxx:74).
// Everything is enqueued on the library's own stream: no host synchronisation.
// One process per GPU; rank 0 creates the id (sdpb_hip_rccl_unique_id), the host
// program distributes those bytes by whatever it already has (MPI_Bcast, a file,
// torch.distributed) and every rank calls sdpb_hip_rccl_init.
#pragma once
#ifndef SDPB_NO_RCCL
#include <rccl/rccl.h>
#endif

namespace sdpb
{
#ifndef SDPB_NO_RCCL
struct RcclComm : Comm
{
  ncclComm_t comm = nullptr;
  static void check(ncclResult_t r, const char *what)
  {
    if(r != ncclSuccess)
      throw HipError(3, std::string(what) + ": " + ncclGetErrorString(r));
  }
  RcclComm(const void *id, size_t id_bytes, int rank, int world)
  {
    if(id_bytes != sizeof(ncclUniqueId))
      throw SolverError(4, "sdpb_hip_rccl_init: the unique id must be " + std::to_string(sizeof(ncclUniqueId)) + " bytes");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    check(ncclCommInitRank(&comm, world, uid, rank), "ncclCommInitRank");
  }
  ~RcclComm() override
  {
    if(comm)
      (void)ncclCommDestroy(comm);
  }
  void allgather(const void *send, void *recv, size_t bytes, hipStream_t st) override
  {
    check(ncclAllGather(send, recv, bytes, ncclUint8, comm, st), "ncclAllGather");
  }
  void allreduce_sum_u64(void *buf, size_t count, hipStream_t st) override
  {
    check(ncclAllReduce(buf, buf, count, ncclUint64, ncclSum, comm, st), "ncclAllReduce");
  }
  bool broadcast(void *buf, size_t bytes, int root, hipStream_t st) override
  {
    check(ncclBroadcast(buf, buf, bytes, ncclUint8, root, comm, st), "ncclBroadcast");
    return true;
  }
  const char *name() const override { return "rccl"; }
  int async_error() const override
  {
    ncclResult_t r = ncclSuccess;
    if(!comm || ncclCommGetAsyncError(comm, &r) != ncclSuccess)
      return -1;
    return (int)r;
  }
  int ranks() const override
  {
    int n = 0;
    return ncclCommCount(comm, &n) == ncclSuccess ? n : 0;
  }
};
inline Comm *make_rccl_comm(const void *id, size_t id_bytes, int rank, int world) { return new RcclComm(id, id_bytes, rank, world); }
inline size_t rccl_unique_id(void *out, size_t capacity)
{
  if(capacity < sizeof(ncclUniqueId))
    throw SolverError(4, "sdpb_hip_rccl_unique_id: buffer too small");
  ncclUniqueId uid;
  RcclComm::check(ncclGetUniqueId(&uid), "ncclGetUniqueId");
  std::memcpy(out, &uid, sizeof uid);
  return sizeof uid;
}
#else
inline Comm *make_rccl_comm(const void *, size_t, int, int) { throw SolverError(4, "this build of the library has no RCCL (CPU emulation)"); }
inline size_t rccl_unique_id(void *, size_t) { throw SolverError(4, "this build of the library has no RCCL (CPU emulation)"); }
#endif
} // namespace sdpb
