// dev.hpp — device memory containers and launch plumbing for the sdp_solve hot path.
//
// Layout in HBM: every array of multi-word numbers is stored
// limb-major (structure of arrays): plane 0 holds the sign/exponent words, plane
// 1+i holds limb i of every element; `stride` elements per plane.  Matrices are
// column-major inside an array (as El::Matrix is), a batch of per-block matrices
// shares one array and is addressed through a small descriptor table.
#pragma once
#include <hip/hip_runtime.h>

#include "mw.hpp"

#include <stdexcept>
#include <string>
#include <vector>

#ifndef HIP_KERNEL_NAME
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#endif

namespace sdpb
{
struct HipError : std::runtime_error
{
  int code; // 2 = out of memory, 3 = HIP/RCCL (include/sdpb_hip.h)
  HipError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void hip_check(hipError_t e, const char *what, const char *file, int line)
{
  if(e != hipSuccess)
    throw HipError(e == hipErrorOutOfMemory ? 2 : 3, std::string(what) + ": " + hipGetErrorString(e) + " (" + file + ":"
                                                       + std::to_string(line) + ")");
}
#define HIP_CHECK(x) ::sdpb::hip_check((x), #x, __FILE__, __LINE__)

// One matrix of a batch: element (i,j) lives at off + i + j*ld of the owning array.
struct MatDesc
{
  unsigned long long off;
  int rows, cols, ld;
  int aux; // kernel-specific (e.g. block's num_points)
};

// A batch of matrices inside one limb-major array.
struct Batch
{
  mw::Ptr p;
  const MatDesc *d; // device pointer, `count` entries
  int count;
};

// Owning device array of `n` multi-word numbers with NL limbs each.
class DevArray
{
public:
  uint32_t *base = nullptr;
  size_t n = 0;
  int planes = 0;
  DevArray() = default;
  DevArray(const DevArray &) = delete;
  DevArray &operator=(const DevArray &) = delete;
  ~DevArray() { release(); }
  void alloc(size_t count, int NL)
  {
    release();
    n = count ? count : 1;
    planes = NL + 1;
    HIP_CHECK(hipMalloc(&base, bytes()));
    HIP_CHECK(hipMemset(base, 0, bytes()));
  }
  void release()
  {
    if(base)
      (void)hipFree(base);
    base = nullptr;
  }
  size_t bytes() const { return n * (size_t)planes * sizeof(uint32_t); }
  mw::Ptr ptr() const { return mw::Ptr{base, n}; }
  mw::CPtr cptr() const { return mw::CPtr(base, n); }
};

template <class T> class DevBuf
{
public:
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf()
  {
    if(p)
      (void)hipFree(p);
  }
  void alloc(size_t count)
  {
    if(p)
      (void)hipFree(p);
    n = count ? count : 1;
    HIP_CHECK(hipMalloc(&p, n * sizeof(T)));
    HIP_CHECK(hipMemset(p, 0, n * sizeof(T)));
  }
  void upload(const std::vector<T> &h)
  {
    alloc(h.size());
    if(!h.empty())
      HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  std::vector<T> download() const
  {
    std::vector<T> h(n);
    HIP_CHECK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

// Host-side staging of multi-word numbers <-> device arrays.
template <int NL> void upload(const DevArray &a, size_t first, const std::vector<mw::Mw<NL>> &h)
{
  if(h.empty())
    return;
  const size_t cnt = h.size();
  std::vector<uint32_t> planes(cnt * (NL + 1));
  for(size_t i = 0; i < cnt; ++i)
    {
      const mw::Mw<NL> &v = h[i];
      const bool z = v.e == mw::EZERO;
      planes[i] = z ? 0u : ((v.neg << 31) | (uint32_t)(v.e + (int32_t)mw::EBIAS));
      for(int k = 0; k < NL; ++k)
        planes[(size_t)(k + 1) * cnt + i] = z ? 0u : v.m[k];
    }
  // one strided copy for all NL+1 planes
  HIP_CHECK(hipMemcpy2D(a.base + first, a.n * sizeof(uint32_t), planes.data(), cnt * sizeof(uint32_t), cnt * sizeof(uint32_t),
                        NL + 1, hipMemcpyHostToDevice));
}
template <int NL> std::vector<mw::Mw<NL>> download(const DevArray &a, size_t first, size_t count)
{
  std::vector<mw::Mw<NL>> h(count);
  if(!count)
    return h;
  std::vector<uint32_t> planes(count * (NL + 1));
  HIP_CHECK(hipMemcpy2D(planes.data(), count * sizeof(uint32_t), a.base + first, a.n * sizeof(uint32_t), count * sizeof(uint32_t),
                        NL + 1, hipMemcpyDeviceToHost));
  for(size_t i = 0; i < count; ++i)
    {
      const uint32_t hd = planes[i];
      h[i].neg = hd >> 31;
      h[i].e = hd ? (int32_t)((hd & 0x7fffffffu) - mw::EBIAS) : mw::EZERO;
      for(int k = 0; k < NL; ++k)
        h[i].m[k] = planes[(size_t)(k + 1) * count + i];
    }
  return h;
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
} // namespace sdpb
