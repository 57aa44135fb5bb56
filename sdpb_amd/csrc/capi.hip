// capi.hip — the C ABI declared in include/sdpb_hip.h.
#include "../../include/sdpb_hip.h"

#include "solver.hpp"

#include <cstring>
#include <memory>

struct sdpb_hip_ctx
{
  std::unique_ptr<sdpb::SolverBase> solver;
  std::string error, strbuf;
};

namespace
{
thread_local std::string g_create_error;

int fail(sdpb_hip_ctx *ctx, int code, const std::string &msg)
{
  if(ctx)
    ctx->error = msg;
  else
    g_create_error = msg;
  return code;
}
template <class F> int guarded(sdpb_hip_ctx *ctx, F f)
{
  if(!ctx)
    return fail(nullptr, 4, "null context");
  try
    {
      f();
      return 0;
    }
  catch(sdpb::SolverError &e)
    {
      return fail(ctx, e.code, e.what());
    }
  catch(sdpb::HipError &e)
    {
      return fail(ctx, e.code, e.what());
    }
  catch(std::bad_alloc &)
    {
      return fail(ctx, 2, "out of host memory");
    }
  catch(std::exception &e)
    {
      return fail(ctx, 4, e.what());
    }
}
int copy_out(sdpb_hip_ctx *ctx, const std::string &s, char *buf, size_t buflen, size_t *needed)
{
  if(needed)
    *needed = s.size() + 1;
  if(!buf || buflen < s.size() + 1)
    return fail(ctx, 4, "output buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}
} // namespace

// streaming copy, 16 bytes per lane (sdpb_hip_copy_bandwidth: the measured HBM ceiling)
struct alignas(16) Quad
{
  uint32_t x, y, z, w;
};
// four 16-byte nontemporal loads in flight per lane, one workgroup per 16 KiB (no grid-stride loop): 6.2 TB/s on
// MI355X, against 4.7 TB/s for 8192 workgroups striding over the buffer (scratch measurement, round 2)
__global__ void __launch_bounds__(256) k_copy16(const Quad *src, Quad *dst, size_t n)
{
#if defined(__HIPCC__)
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const v4 *s = reinterpret_cast<const v4 *>(src);
  v4 *d = reinterpret_cast<v4 *>(dst);
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  v4 v[4];
#pragma unroll
  for(int u = 0; u < 4; ++u)
    if(base + (size_t)u * 256 < n)
      v[u] = __builtin_nontemporal_load(s + base + (size_t)u * 256);
#pragma unroll
  for(int u = 0; u < 4; ++u)
    if(base + (size_t)u * 256 < n)
      __builtin_nontemporal_store(v[u], d + base + (size_t)u * 256);
#else // CPU emulation build (tests): same index map, plain copies
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  for(int u = 0; u < 4; ++u)
    if(base + (size_t)u * 256 < n)
      dst[base + (size_t)u * 256] = src[base + (size_t)u * 256];
#endif
}

extern "C" {

int sdpb_hip_create_with_costs(int precision_bits, int num_blocks, const int *dims, const int *num_points, int N, int device_id,
                               int rank, int world_size, const long long *block_costs, sdpb_hip_ctx **out);
int sdpb_hip_create(int precision_bits, int num_blocks, const int *dims, const int *num_points, int N, int device_id,
                    int rank, int world_size, sdpb_hip_ctx **out)
{
  return sdpb_hip_create_with_costs(precision_bits, num_blocks, dims, num_points, N, device_id, rank, world_size, nullptr, out);
}
int sdpb_hip_create_with_costs(int precision_bits, int num_blocks, const int *dims, const int *num_points, int N, int device_id,
                               int rank, int world_size, const long long *block_costs, sdpb_hip_ctx **out)
{
  if(!out || !dims || !num_points || num_blocks <= 0 || world_size < 1 || rank < 0 || rank >= world_size)
    return fail(nullptr, 4, "sdpb_hip_create: bad argument");
  *out = nullptr;
  try
    {
      int ndev = 0;
      if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, 3, "sdpb_hip_create: no HIP device found — this library has no CPU path");
      if(device_id >= 0)
        HIP_CHECK(hipSetDevice(device_id));
      std::vector<int> d(dims, dims + num_blocks), k(num_points, num_points + num_blocks);
      for(int j = 0; j < num_blocks; ++j)
        if(d[j] <= 0 || k[j] <= 0)
          return fail(nullptr, 4, "sdpb_hip_create: dim and num_points must be positive");
      std::vector<long long> costs;
      if(block_costs)
        {
          costs.assign(block_costs, block_costs + num_blocks);
          for(long long c : costs)
            if(c < 0)
              return fail(nullptr, 4, "sdpb_hip_create_with_costs: negative block cost");
        }
      const int want = mw::limbs_for_precision(precision_bits);
      std::unique_ptr<sdpb_hip_ctx> ctx(new sdpb_hip_ctx);
      sdpb::SolverBase *s = nullptr;
      // round the request up to the next compiled mantissa width ("GMP will round this
      // up", Solver_Parameters.cxx:26-28)
      struct Entry
      {
        int limbs;
        sdpb::SolverBase *(*make)(int, const std::vector<int> &, const std::vector<int> &, int, int, int, const std::vector<long long> &);
      };
      const Entry table[] = {{6, sdpb::make_solver_6},   {10, sdpb::make_solver_10}, {16, sdpb::make_solver_16},
                             {18, sdpb::make_solver_18}, {24, sdpb::make_solver_24}, {26, sdpb::make_solver_26},
                             {34, sdpb::make_solver_34}, {42, sdpb::make_solver_42}, {50, sdpb::make_solver_50},
                             {66, sdpb::make_solver_66}};
      for(const Entry &e : table)
        if(want <= e.limbs && e.make)
          {
            s = e.make(precision_bits, d, k, N, rank, world_size, costs);
            break;
          }
      if(!s)
        return fail(nullptr, 4, "sdpb_hip_create: no compiled mantissa width covers --precision " + std::to_string(precision_bits)
                                   + ": this library is built for 128 ... 2048 bits (limb counts 6, 10, 16, 18, 24, 26, 34, 42, 50, 66; "
                                     "sdpb_amd/build.py, ALL_LIMBS)");
      ctx->solver.reset(s);
      *out = ctx.release();
      return 0;
    }
  catch(sdpb::SolverError &e)
    {
      return fail(nullptr, e.code, e.what());
    }
  catch(sdpb::HipError &e)
    {
      return fail(nullptr, e.code, e.what());
    }
  catch(std::exception &e)
    {
      return fail(nullptr, 4, e.what());
    }
}

void sdpb_hip_destroy(sdpb_hip_ctx *ctx) { delete ctx; }

const char *sdpb_hip_last_error(sdpb_hip_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int sdpb_hip_set_param(sdpb_hip_ctx *ctx, const char *name, const char *value)
{
  return guarded(ctx, [&] { ctx->solver->set_param(name, value); });
}
int sdpb_hip_set_flags(sdpb_hip_ctx *ctx, long max_iterations, int fpf, int fdf, int dpfj, int ddfj)
{
  return guarded(ctx, [&] { ctx->solver->set_flags(max_iterations, fpf, fdf, dpfj, ddfj); });
}
int sdpb_hip_set_block(sdpb_hip_ctx *ctx, int j, const char *be, const char *bo, const char *B, const char *c)
{
  return guarded(ctx, [&] { ctx->solver->set_block(j, be, bo, B, c); });
}
int sdpb_hip_set_block_f64(sdpb_hip_ctx *ctx, int j, const char *be, const char *bo, const double *B, const double *c)
{
  return guarded(ctx, [&] { ctx->solver->set_block_f64(j, be, bo, B, c); });
}
int sdpb_hip_set_objective(sdpb_hip_ctx *ctx, const char *b, const char *constant)
{
  return guarded(ctx, [&] { ctx->solver->set_objective(b, constant); });
}
int sdpb_hip_init_state(sdpb_hip_ctx *ctx)
{
  return guarded(ctx, [&] { ctx->solver->init_state(); });
}
int sdpb_hip_iterate(sdpb_hip_ctx *ctx, int *terminated)
{
  return guarded(ctx, [&] {
    const bool t = ctx->solver->iterate();
    if(terminated)
      *terminated = t ? 1 : 0;
  });
}
int sdpb_hip_terminate_reason(sdpb_hip_ctx *ctx) { return ctx ? ctx->solver->terminate_reason() : -1; }
const char *sdpb_hip_terminate_string(sdpb_hip_ctx *ctx)
{
  return ctx ? sdpb::terminate_string(ctx->solver->terminate_reason()) : "";
}
int sdpb_hip_get_scalar(sdpb_hip_ctx *ctx, const char *name, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->get_scalar(name); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}
int sdpb_hip_get_array(sdpb_hip_ctx *ctx, const char *which, int j, int parity, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->get_array(which, j, parity); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}
int sdpb_hip_set_array(sdpb_hip_ctx *ctx, const char *which, int j, int parity, const char *values)
{
  return guarded(ctx, [&] { ctx->solver->set_array(which, j, parity, values); });
}
int sdpb_hip_schur_solver_init(sdpb_hip_ctx *ctx)
{
  return guarded(ctx, [&] { ctx->solver->schur_solver_init(); });
}
int sdpb_hip_schur_solve(sdpb_hip_ctx *ctx)
{
  return guarded(ctx, [&] { ctx->solver->schur_solve(); });
}
int sdpb_hip_set_block_mpf(sdpb_hip_ctx *ctx, int j, int limbs64, const unsigned long long *be, const unsigned long long *bo,
                           const unsigned long long *B, const unsigned long long *c)
{
  return guarded(ctx, [&] {
    ctx->solver->set_block_mpf(j, limbs64, (const uint64_t *)be, (const uint64_t *)bo, (const uint64_t *)B, (const uint64_t *)c);
  });
}
int sdpb_hip_set_objective_mpf(sdpb_hip_ctx *ctx, int limbs64, const unsigned long long *b, const unsigned long long *constant)
{
  return guarded(ctx, [&] { ctx->solver->set_objective_mpf(limbs64, (const uint64_t *)b, (const uint64_t *)constant); });
}
int sdpb_hip_get_array_mpf(sdpb_hip_ctx *ctx, const char *which, int j, int parity, int limbs64, unsigned long long *out,
                           size_t capacity, size_t *count)
{
  return guarded(ctx, [&] {
    if(!which || !count)
      throw sdpb::SolverError(4, "sdpb_hip_get_array_mpf: null argument");
    *count = ctx->solver->get_array_mpf(which, j, parity, limbs64, (uint64_t *)out, capacity);
  });
}
int sdpb_hip_set_array_mpf(sdpb_hip_ctx *ctx, const char *which, int j, int parity, int limbs64,
                           const unsigned long long *values, size_t count)
{
  return guarded(ctx, [&] {
    if(!which)
      throw sdpb::SolverError(4, "sdpb_hip_set_array_mpf: null argument");
    ctx->solver->set_array_mpf(which, j, parity, limbs64, (const uint64_t *)values, count);
  });
}

int sdpb_hip_block_owner(sdpb_hip_ctx *ctx, int j)
{
  int r = -1;
  guarded(ctx, [&] { r = ctx->solver->block_owner(j); });
  return r;
}
int sdpb_hip_limbs(sdpb_hip_ctx *ctx) { return ctx ? ctx->solver->limbs() : 0; }
int sdpb_hip_fx_frac_bits(sdpb_hip_ctx *ctx) { return ctx ? ctx->solver->fx_frac_bits() : 0; }
int sdpb_hip_bench_op(sdpb_hip_ctx *ctx, const char *op, int a, int b, int reps, double *ms)
{
  return guarded(ctx, [&] {
    if(!op || !ms)
      throw sdpb::SolverError(4, "sdpb_hip_bench_op: null argument");
    *ms = ctx->solver->bench_op(op, a, b, reps);
  });
}

int sdpb_hip_set_collectives(sdpb_hip_ctx *ctx, const sdpb_hip_collectives *c)
{
  return guarded(ctx, [&] {
    sdpb::Collectives cc;
    if(c)
      {
        cc.allreduce_sum_u64 = c->allreduce_sum_u64;
        cc.allgather_bytes = c->allgather_bytes;
        cc.user = c->user;
      }
    ctx->solver->set_collectives(cc);
  });
}
int sdpb_hip_rccl_unique_id(char id[SDPB_HIP_RCCL_ID_BYTES])
{
  try
    {
      if(!id)
        return fail(nullptr, 4, "sdpb_hip_rccl_unique_id: null buffer");
      std::memset(id, 0, SDPB_HIP_RCCL_ID_BYTES);
      sdpb::rccl_unique_id(id, SDPB_HIP_RCCL_ID_BYTES);
      return 0;
    }
  catch(sdpb::SolverError &e)
    {
      return fail(nullptr, e.code, e.what());
    }
  catch(std::exception &e)
    {
      return fail(nullptr, 3, e.what());
    }
}
int sdpb_hip_rccl_init(sdpb_hip_ctx *ctx, const char id[SDPB_HIP_RCCL_ID_BYTES])
{
  return guarded(ctx, [&] {
    if(!id)
      throw sdpb::SolverError(4, "sdpb_hip_rccl_init: null id");
    ctx->solver->init_rccl(id, SDPB_HIP_RCCL_ID_BYTES);
  });
}
const char *sdpb_hip_comm_name(sdpb_hip_ctx *ctx) { return ctx ? ctx->solver->comm_name() : ""; }
int sdpb_hip_rccl_selftest(size_t bytes)
{
  try
    {
      const size_t n = std::max<size_t>(bytes / 8, 1);
      char id[SDPB_HIP_RCCL_ID_BYTES];
      sdpb::rccl_unique_id(id, sizeof id);
      std::unique_ptr<sdpb::Comm> comm(sdpb::make_rccl_comm(id, sizeof id, 0, 1));
      std::vector<unsigned long long> h(n), back(n);
      for(size_t i = 0; i < n; ++i)
        h[i] = 0x9E3779B97F4A7C15ull * (i + 1);
      sdpb::DevBuf<unsigned long long> a, b;
      a.upload(h);
      b.alloc(n);
      hipStream_t st;
      HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      comm->allgather(a.p, b.p, n * 8, st);
      comm->allreduce_sum_u64(b.p, n, st);
      HIP_CHECK(hipMemcpyAsync(back.data(), b.p, n * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      (void)hipStreamDestroy(st);
      if(back != h)
        return fail(nullptr, 3, "sdpb_hip_rccl_selftest: data changed in a one-rank all-gather + all-reduce");
      return 0;
    }
  catch(sdpb::SolverError &e)
    {
      return fail(nullptr, e.code, e.what());
    }
  catch(std::exception &e)
    {
      return fail(nullptr, 3, e.what());
    }
}
// The same transport class with MORE than one rank, outside any solver: what a launcher runs once (in a
// short-lived process, under a timeout) before it trusts the in-library exchange with the iteration.
int sdpb_hip_rccl_preflight(const char id[SDPB_HIP_RCCL_ID_BYTES], int rank, int world, size_t bytes)
{
  try
    {
      if(world < 1 || rank < 0 || rank >= world)
        return fail(nullptr, 4, "sdpb_hip_rccl_preflight: need 0 <= rank < world");
      const size_t n = std::max<size_t>(bytes / 8, 1);
      std::unique_ptr<sdpb::Comm> comm(sdpb::make_rccl_comm(id, SDPB_HIP_RCCL_ID_BYTES, rank, world));
      if(comm->ranks() != world)
        return fail(nullptr, 3, "sdpb_hip_rccl_preflight: ncclCommCount disagrees with the world size");
      auto pattern = [](int r, size_t i) { return 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1) + 0xD1B54A32D192ED03ull * (unsigned long long)(r + 1); };
      std::vector<unsigned long long> mine(n), back(n), all((size_t)world * n);
      for(size_t i = 0; i < n; ++i)
        mine[i] = pattern(rank, i);
      sdpb::DevBuf<unsigned long long> a, g, m;
      a.upload(mine);
      g.alloc((size_t)world * n);
      m.alloc(n);
      // released on every way out of this function (round-4 advisor: the early "return fail" paths leaked them)
      struct Handles
      {
        hipStream_t s1 = nullptr, s2 = nullptr;
        hipEvent_t ev = nullptr;
        ~Handles()
        {
          if(ev)
            (void)hipEventDestroy(ev);
          if(s1)
            (void)hipStreamDestroy(s1);
          if(s2)
            (void)hipStreamDestroy(s2);
        }
      } h;
      HIP_CHECK(hipStreamCreateWithFlags(&h.s1, hipStreamNonBlocking));
      HIP_CHECK(hipStreamCreateWithFlags(&h.s2, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&h.ev, hipEventDisableTiming));
      const hipStream_t s1 = h.s1, s2 = h.s2;
      const hipEvent_t ev = h.ev;
      // 1. all-gather in rank order (result blocks, N-vectors)
      comm->allgather(a.p, g.p, n * 8, s1);
      HIP_CHECK(hipMemcpyAsync(all.data(), g.p, (size_t)world * n * 8, hipMemcpyDeviceToHost, s1));
      HIP_CHECK(hipStreamSynchronize(s1));
      for(int r = 0; r < world; ++r)
        for(size_t i = 0; i < n; i += std::max<size_t>(n / 257, 1))
          if(all[(size_t)r * n + i] != pattern(r, i))
            return fail(nullptr, 3, "sdpb_hip_rccl_preflight: all-gather returned wrong data");
      // 2. in-place 64-bit SUM all-reduce (the Q' image); wrapping sums are exact
      comm->allreduce_sum_u64(a.p, n, s1);
      HIP_CHECK(hipMemcpyAsync(back.data(), a.p, n * 8, hipMemcpyDeviceToHost, s1));
      HIP_CHECK(hipStreamSynchronize(s1));
      for(size_t i = 0; i < n; i += std::max<size_t>(n / 257, 1))
        {
          unsigned long long want = 0;
          for(int r = 0; r < world; ++r)
            want += pattern(r, i);
          if(back[i] != want)
            return fail(nullptr, 3, "sdpb_hip_rccl_preflight: 64-bit all-reduce returned a wrong sum");
        }
      // 3. in-place broadcasts from every root, issued alternately from two streams with a dependent kernel-free
      //    hand-over in between (the panel messages of the distributed Cholesky(Q) come from the chain stream
      //    while other collectives of the iteration are issued from the main stream)
      for(int root = 0; root < world; ++root)
        {
          hipStream_t st = (root & 1) ? s2 : s1, other = (root & 1) ? s1 : s2;
          for(size_t i = 0; i < n; ++i)
            mine[i] = pattern(rank + 7 * (root + 1), i);
          HIP_CHECK(hipMemcpyAsync(m.p, mine.data(), n * 8, hipMemcpyHostToDevice, st));
          if(!comm->broadcast(m.p, n * 8, root, st))
            return fail(nullptr, 3, "sdpb_hip_rccl_preflight: the transport offers no broadcast");
          HIP_CHECK(hipEventRecord(ev, st));
          HIP_CHECK(hipStreamWaitEvent(other, ev, 0));
          HIP_CHECK(hipMemcpyAsync(back.data(), m.p, n * 8, hipMemcpyDeviceToHost, other));
          HIP_CHECK(hipStreamSynchronize(other));
          for(size_t i = 0; i < n; i += std::max<size_t>(n / 257, 1))
            if(back[i] != pattern(root + 7 * (root + 1), i))
              return fail(nullptr, 3, "sdpb_hip_rccl_preflight: in-place broadcast returned wrong data");
        }
      if(comm->async_error() != 0)
        return fail(nullptr, 3, "sdpb_hip_rccl_preflight: the communicator reports an asynchronous error");
      return 0;
    }
  catch(sdpb::SolverError &e)
    {
      return fail(nullptr, e.code, e.what());
    }
  catch(std::exception &e)
    {
      return fail(nullptr, 3, e.what());
    }
}
int sdpb_hip_set_max_runtime(sdpb_hip_ctx *ctx, double seconds)
{
  return guarded(ctx, [&] { ctx->solver->set_max_runtime(seconds); });
}
int sdpb_hip_set_max_shared_memory(sdpb_hip_ctx *ctx, unsigned long long bytes)
{
  return guarded(ctx, [&] { ctx->solver->set_max_shared_memory(bytes); });
}
int sdpb_hip_memory_plan(sdpb_hip_ctx *ctx, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->memory_plan_json(); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}
void sdpb_hip_request_stop(sdpb_hip_ctx *ctx)
{
  if(ctx)
    ctx->solver->request_stop();
}
int sdpb_hip_set_profiling(sdpb_hip_ctx *ctx, int on)
{
  return guarded(ctx, [&] { ctx->solver->set_profiling(on != 0); });
}
long sdpb_hip_host_syncs(sdpb_hip_ctx *ctx) { return ctx ? ctx->solver->host_syncs() : 0; }
// lock-free by construction (atomics only): no guarded(), no last_error, callable while another thread is inside
// sdpb_hip_iterate
int sdpb_hip_progress(sdpb_hip_ctx *ctx, unsigned long long out[8])
{
  if(!ctx || !out)
    return 4;
  ctx->solver->progress(out);
  return 0;
}

int sdpb_hip_timers(sdpb_hip_ctx *ctx, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->timers_json(); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}

int sdpb_hip_block_timings(sdpb_hip_ctx *ctx, long long *microseconds)
{
  return guarded(ctx, [&] {
    if(!microseconds)
      throw sdpb::SolverError(4, "sdpb_hip_block_timings: null output");
    ctx->solver->block_timings(microseconds);
  });
}
int sdpb_hip_block_clock_ticks(sdpb_hip_ctx *ctx, unsigned long long *cholesky, unsigned long long *solve)
{
  return guarded(ctx, [&] {
    if(!cholesky || !solve)
      throw sdpb::SolverError(4, "sdpb_hip_block_clock_ticks: null output");
    ctx->solver->block_clock_ticks(cholesky, solve);
  });
}
int sdpb_hip_plan_blocks_with_costs(int num_blocks, const long long *block_costs, int world_size, int *owners)
{
  if(num_blocks <= 0 || !block_costs || !owners || world_size < 1)
    return 4;
  const std::vector<int> ones(num_blocks, 1);
  const std::vector<long long> costs(block_costs, block_costs + num_blocks);
  const std::vector<int> o = sdpb::plan_block_owners(ones, ones, 1, world_size, costs);
  std::copy(o.begin(), o.end(), owners);
  return 0;
}
int sdpb_hip_plan_blocks(int num_blocks, const int *dims, const int *num_points, int N, int world_size, int *owners)
{
  if(num_blocks <= 0 || !dims || !num_points || !owners || world_size < 1)
    return 4;
  std::vector<int> d(dims, dims + num_blocks), k(num_points, num_points + num_blocks);
  const std::vector<int> o = sdpb::plan_block_owners(d, k, N, world_size);
  std::copy(o.begin(), o.end(), owners);
  return 0;
}

int sdpb_hip_op_scalar(sdpb_hip_ctx *ctx, const char *op, const char *a, const char *b, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->op_scalar(op, a, b); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}
int sdpb_hip_op_int_syrk(sdpb_hip_ctx *ctx, int rows, int cols, const char *P, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] { ctx->strbuf = ctx->solver->op_int_syrk(rows, cols, P); });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}

int sdpb_hip_op_syrk_Q(sdpb_hip_ctx *ctx, int rows, int cols, const char *P, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] {
    if(!P)
      throw sdpb::SolverError(4, "sdpb_hip_op_syrk_Q: null argument");
    ctx->strbuf = ctx->solver->op_syrk_Q(rows, cols, P);
  });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}

int sdpb_hip_op_min_eigenvalue(sdpb_hip_ctx *ctx, int n, const char *A, char *buf, size_t buflen, size_t *needed)
{
  int rc = guarded(ctx, [&] {
    if(!A)
      throw sdpb::SolverError(4, "sdpb_hip_op_min_eigenvalue: null argument");
    ctx->strbuf = ctx->solver->op_min_eigenvalue(n, A);
  });
  return rc ? rc : copy_out(ctx, ctx->strbuf, buf, buflen, needed);
}

int sdpb_hip_host_encode_u64(const char *s, int planes, unsigned long long *lanes)
{
  if(!s || !lanes || planes <= 0)
    return 4;
  bool negative = false;
  if(*s == '-')
    {
      negative = true;
      ++s;
    }
  mw::BigNat n;
  for(; *s; ++s)
    {
      if(*s < '0' || *s > '9')
        return 4;
      if(n.w.empty())
        {
          if(*s != '0')
            n.w.push_back((uint32_t)(*s - '0'));
        }
      else
        n.mul_small(10, (uint32_t)(*s - '0'));
    }
  if((int)n.w.size() >= planes)
    return 4;
  std::vector<uint32_t> w(planes, 0);
  std::copy(n.w.begin(), n.w.end(), w.begin());
  if(negative && !n.w.empty())
    {
      uint64_t carry = 1;
      for(auto &x : w)
        {
          const uint64_t t = (uint64_t)(~x) + carry;
          x = (uint32_t)t;
          carry = t >> 32;
        }
    }
  for(int k = 0; k < planes; ++k)
    lanes[k] = w[k];
  return 0;
}
int sdpb_hip_host_decode_u64(const unsigned long long *lanes, int planes, char *buf, size_t buflen, size_t *needed)
{
  if(!lanes || planes <= 0)
    return 4;
  mw::BigNat n;
  n.w.resize(planes);
  unsigned long long carry = 0;
  for(int k = 0; k < planes; ++k)
    {
      const unsigned long long t = lanes[k] + carry;
      n.w[k] = (uint32_t)t;
      carry = t >> 32;
    }
  const bool negative = n.w[planes - 1] >> 31;
  if(negative)
    {
      uint64_t c = 1;
      for(auto &x : n.w)
        {
          const uint64_t t = (uint64_t)(~x) + c;
          x = (uint32_t)t;
          c = t >> 32;
        }
    }
  n.trim();
  std::string dig;
  while(!n.is_zero())
    {
      uint32_t rem = n.div_small(1000000000u);
      for(int k = 0; k < 9; ++k)
        {
          dig.push_back((char)('0' + rem % 10));
          rem /= 10;
        }
    }
  while(!dig.empty() && dig.back() == '0')
    dig.pop_back();
  std::string out;
  if(dig.empty())
    out = "0";
  else
    {
      if(negative)
        dig.push_back('-');
      out.assign(dig.rbegin(), dig.rend());
    }
  if(needed)
    *needed = out.size() + 1;
  if(!buf || buflen < out.size() + 1)
    return 4;
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}
int sdpb_hip_copy_bandwidth(size_t bytes, int reps, double *gb_per_s)
{
  try
    {
      const size_t n = std::max<size_t>(bytes / 16, 1);
      sdpb::DevBuf<Quad> src, dst;
      src.alloc(n);
      dst.alloc(n);
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      double best = 0;
      for(int r = 0; r < std::max(reps, 1) + 1; ++r)
        {
          HIP_CHECK(hipEventRecord(e0, nullptr));
          hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, nullptr, (const Quad *)src.p, dst.p, n);
          HIP_CHECK(hipEventRecord(e1, nullptr));
          HIP_CHECK(hipEventSynchronize(e1));
          float ms = 0;
          HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
          if(r > 0 && ms > 0) // repetition 0 warms up
            best = std::max(best, 2.0 * 16.0 * (double)n / (ms * 1e-3) / 1e9);
        }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      *gb_per_s = best;
      return 0;
    }
  catch(const std::exception &e)
    {
      g_create_error = e.what();
      return 3;
    }
}
} // extern "C"
