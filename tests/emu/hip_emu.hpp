// hip_emu.hpp — TEST INFRASTRUCTURE ONLY.
//
// A minimal CPU stand-in for the part of the HIP runtime that sdpb_amd/csrc uses, so
// that the *host logic* of the library (block bookkeeping, launch sequences, index
// arithmetic inside kernels) can be exercised by `pytest -m "not gpu"` in a container
// with no GPU.  The product library (sdpb_amd/csrc -> libsdpb_hip.so, built by hipcc
// for gfx950) never includes this file and has no CPU path: it fails loudly when no
// GPU is present.  The emulated library is built from the same unmodified sources into
// tests/emu/_build/libsdpb_hip_emu.so and is loaded only by tests.
//
// Model: blocks run one after another (OpenMP across blocks); the threads of a block
// are cooperative fibers (ucontext) so __syncthreads() works; `__shared__` becomes a
// per-OS-thread static.
#pragma once
#include <ucontext.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <vector>

#define SDPB_HIP_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

struct uint3
{
  unsigned x, y, z;
};
struct dim3
{
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef int hipError_t;
enum
{
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNoDevice = 100
};
typedef void *hipStream_t;
struct hipEmuEvent
{
  std::chrono::steady_clock::time_point t;
};
typedef hipEmuEvent *hipEvent_t;
enum hipMemcpyKind
{
  hipMemcpyHostToHost,
  hipMemcpyHostToDevice,
  hipMemcpyDeviceToHost,
  hipMemcpyDeviceToDevice,
  hipMemcpyDefault
};
struct hipDeviceProp_t
{
  char name[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
};

inline const char *hipGetErrorString(hipError_t e)
{
  return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "out of memory (emu)" : "hip error (emu)");
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n)
{
  *n = 1;
  return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d)
{
  *d = 0;
  return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
  std::snprintf(p->name, sizeof p->name, "cpu-emulation");
  p->multiProcessorCount = 8;
  p->totalGlobalMem = (size_t)16 << 30;
  return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes)
{
  *free_bytes = (size_t)8 << 30;
  *total_bytes = (size_t)16 << 30;
  return hipSuccess;
}
template <class T> inline hipError_t hipMalloc(T **p, size_t n)
{
  *p = (T *)std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void *p)
{
  std::free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind)
{
  for(size_t r = 0; r < height; ++r)
    std::memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
  return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n)
{
  std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr)
{
  std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s)
{
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum
{
  hipStreamDefault = 0,
  hipStreamNonBlocking = 1
};
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *)
{
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int)
{
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest)
{
  *least = 0;
  *greatest = -1;
  return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e)
{
  *e = new hipEmuEvent;
  return hipSuccess;
}
enum
{
  hipEventDisableTiming = 2
};
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e)
{
  delete e;
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr)
{
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

// ---- device-side intrinsics used by the kernels ---------------------------
void __syncthreads();
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
// the device's constant-rate wall clock (100 MHz on the GPU; here: 10 ns ticks of the host's steady clock)
inline unsigned long long wall_clock64()
{
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10ull;
}
inline int atomicOr(int *p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMax(int *p, int v)
{
  int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while(old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
    {}
  return old;
}
inline unsigned atomicMin(unsigned *p, unsigned v)
{
  unsigned old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while(old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
    ;
  return old;
}
inline int atomicMin(int *p, int v)
{
  int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while(old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
    {}
  return old;
}
inline int atomicCAS(int *p, int cmp, int v)
{
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

// ---- launch ---------------------------------------------------------------
void hip_emu_launch(dim3 grid, dim3 block, const std::function<void()> &body);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)            \
  hip_emu_launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
