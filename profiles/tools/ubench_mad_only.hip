// Is v_mad_u64_u32 a full-rate instruction on gfx950?  The tile dot products of csrc/tiledot.hpp issue it WITHOUT the
// v_addc_co_u32 that follows it everywhere else (27-bit limbs: a column sum cannot overflow 64 bits), 298 per term at 18
// limbs.  Three loops, 64 multiply-adds per trip on 24 independent 64-bit accumulators (the tile dot's column sums), occupancy
// 1 ... 4 wavefronts per SIMD:   (a) v_mad_u64_u32 alone (SGPR carry-out, as the compiler emits it);   (b) the pair
// v_mad_u64_u32 + v_addc_co_u32 (profiles/tools/ubench_macpairs.hip's loop);   (c) v_mad_u64_u32 with a full-rate
// v_xor_b32 after each one (does a cheap instruction hide behind the multiplier?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/tools/ubench_mad_only.hip -o /tmp/ubench_mad_only && /tmp/ubench_mad_only
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while(0)
template <int MODE, int WAVES> __global__ void __launch_bounds__(256, WAVES) k(uint32_t *out, uint32_t seed, int trips)
{
  extern __shared__ uint32_t dyn[];
  if(seed == 0xffffffffu)
    dyn[threadIdx.x] = seed;
  uint64_t c[24];
  uint32_t h[24];
  for(int o = 0; o < 24; ++o)
    c[o] = seed + o + threadIdx.x, h[o] = o;
  uint32_t a[8], b[8];
  for(int i = 0; i < 8; ++i)
    a[i] = (seed * (2 * i + 3) + threadIdx.x) & 0x7ffffffu, b[i] = (a[i] ^ (0x55aa * (i + 1))) & 0x7ffffffu;
#pragma unroll 1
  for(int t = 0; t < trips; ++t)
    {
#pragma unroll
      for(int r = 0; r < 64; ++r)
        {
          const int o = (r * 7) % 24, i = r % 8, j = (r / 8) % 8;
          if(MODE == 0)
            asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(c[o]) : "v"(a[i]), "v"(b[j]) : "s20", "s21");
          else if(MODE == 1)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(c[o]), "+v"(h[o]) : "v"(a[i]), "v"(b[j]) : "vcc");
          else
            asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_xor_b32 %1, %2, %1" : "+v"(c[o]), "+v"(h[o]) : "v"(a[i]), "v"(b[j]) : "s20", "s21");
        }
    }
  uint32_t x = 0;
  for(int o = 0; o < 24; ++o)
    x ^= (uint32_t)c[o] ^ (uint32_t)(c[o] >> 32) ^ h[o];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int MODE, int WAVES> int run()
{
  uint32_t *d;
  const int blocks = 256 * WAVES * 8, trips = 10000;
  CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const size_t lds = (size_t)160 * 1024 / WAVES - 2048;
  CHK(hipFuncSetAttribute((const void *)k<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), lds, 0, d, 123u, 100);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), lds, 0, d, 123u, trips);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double macs = (double)blocks * 256 * trips * 64;
  const char *names[3] = {"v_mad_u64_u32 alone      ", "v_mad_u64_u32 + v_addc   ", "v_mad_u64_u32 + v_xor_b32"};
  printf("%s  %d wavefront(s)/SIMD: %8.3f ms  %6.2f e12 multiply-adds/s = %5.2f cycles per multiply-add and SIMD\n", names[MODE], WAVES, ms, macs / ms / 1e9,
         1024.0 * 64 * 2.4e9 / (macs / ms * 1e3));
  CHK(hipFree(d));
  return 0;
}
int main()
{
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  printf("dev %s CUs %d clock %d kHz; 64 multiply-adds per loop trip on 24 accumulators\n", p.name, p.multiProcessorCount, p.clockRate);
  return run<0, 1>() || run<0, 2>() || run<0, 4>() || run<1, 1>() || run<1, 2>() || run<1, 4>() || run<2, 1>() || run<2, 2>() || run<2, 4>();
}
