// Ceiling of v_mad_u64_u32 + v_addc_co_u32 pairs on gfx950, measured the way the syrk kernels issue them: asm statements
// of 8 pairs on 12 column accumulators (the 2 x 2-limb products of k_syrk_fx3's row), PAIRS_PER_TRIP pairs per loop trip,
// occupancy 1 ... 4 wavefronts per SIMD.  (profiles/r01_ubench.txt measured 8 pairs per 19-instruction trip: 17.0e12/s.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/tools/ubench_macpairs.hip -o /tmp/ubench_macpairs && /tmp/ubench_macpairs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while(0)
#define P(C, H, X, Y) "v_mad_u64_u32 %" #C ", vcc, %" #X ", %" #Y ", %" #C "\n\tv_addc_co_u32 %" #H ", vcc, 0, %" #H ", vcc\n\t"
#define BOTH(c, h, o0, o1, a0x, a0y, a1x, a1y, bx, by)                                                                                   \
  asm volatile(P(0, 6, 12, 16) P(3, 9, 14, 16) P(1, 7, 12, 17) P(4, 10, 14, 17) P(1, 7, 13, 16) P(4, 10, 15, 16) P(2, 8, 13, 17)        \
                 P(5, 11, 15, 17)                                                                                                        \
               : "+v"(c[o0][0]), "+v"(c[o0][1]), "+v"(c[o0][2]), "+v"(c[o1][0]), "+v"(c[o1][1]), "+v"(c[o1][2]), "+v"(h[o0][0]),         \
                 "+v"(h[o0][1]), "+v"(h[o0][2]), "+v"(h[o1][0]), "+v"(h[o1][1]), "+v"(h[o1][2])                                          \
               : "v"(a0x), "v"(a0y), "v"(a1x), "v"(a1y), "v"(bx), "v"(by)                                                                \
               : "vcc")
template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k(uint32_t *out, uint32_t seed, int trips)
{
  extern __shared__ uint32_t dyn[];
  if(seed == 0xffffffffu)
    dyn[threadIdx.x] = seed; // keeps the allocation
  uint64_t c[4][3];
  uint32_t h[4][3];
  for(int o = 0; o < 4; ++o)
    for(int q = 0; q < 3; ++q)
      c[o][q] = seed + o + q + threadIdx.x, h[o][q] = 0;
  uint32_t a0 = seed * 3 + threadIdx.x, a1 = a0 * 5 + 1, a2 = a0 * 7 + 3, a3 = a0 * 11 + 5, b0 = a0 ^ 0x55aa, b1 = a1 ^ 0x1234, b2 = a2 + 77, b3 = a3 + 99;
#pragma unroll 1
  for(int t = 0; t < trips; ++t)
    {
#pragma unroll
      for(int r = 0; r < 4; ++r) // 4 rows of 16 pairs per trip
        {
          BOTH(c, h, 0, 1, a0, a1, a2, a3, b0, b1);
          BOTH(c, h, 2, 3, a0, a1, a2, a3, b2, b3);
        }
    }
  uint32_t x = 0;
  for(int o = 0; o < 4; ++o)
    for(int q = 0; q < 3; ++q)
      x ^= (uint32_t)c[o][q] ^ (uint32_t)(c[o][q] >> 32) ^ h[o][q];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int WAVES> int run()
{
  uint32_t *d;
  const int blocks = 256 * WAVES * 8, trips = 20000;
  CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  // residency: one workgroup (4 wavefronts = 1 per SIMD) per (160 KB / WAVES) of LDS
  const size_t lds = (size_t)160 * 1024 / WAVES - 2048;
  CHK(hipFuncSetAttribute((const void *)k<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k<WAVES>, dim3(blocks), dim3(256), lds, 0, d, 123u, 100);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<WAVES>, dim3(blocks), dim3(256), lds, 0, d, 123u, trips);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double pairs = (double)blocks * 256 * trips * 64;
  printf("%d wavefront(s)/SIMD resident: %8.3f ms  %6.2f e12 MAC pairs/s\n", WAVES, ms, pairs / ms / 1e9);
  CHK(hipFree(d));
  return 0;
}
int main()
{
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  printf("dev %s CUs %d clock %d kHz; 64 pairs per loop trip in asm statements of 8\n", p.name, p.multiProcessorCount, p.clockRate);
  return run<1>() || run<2>() || run<3>() || run<4>();
}
