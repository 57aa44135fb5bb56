// Latency of the dependent multi-word operations a Cholesky pivot chain is made of, one lane of one wavefront
// (what k_chol_inv_lds' diagonal owner executes per column), measured with the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdpb_amd/csrc profiles/tools/ubench_chain.hip -o /tmp/ubench_chain && /tmp/ubench_chain
#include <hip/hip_runtime.h>
#include "mw.hpp"
#include <cstdio>
using namespace mw;
template <int NL> __device__ Mw<NL> seed(int i)
{
  Mw<NL> v = from_u32<NL>(3u + (uint32_t)i);
  for(int l = 0; l < NL - 1; ++l)
    v.m[l] = 0x9e3779b9u * (uint32_t)(l + 7 + i);
  return v;
}
template <int NL, int OP> __global__ void k(int reps, unsigned long long *out, uint32_t *sink, int active)
{
  Mw<NL> x = seed<NL>(threadIdx.x), y = seed<NL>(threadIdx.x + 5);
  Acc<NL> acc = acc_zero<NL>();
  acc_add(acc, x);
  const unsigned long long t0 = wall_clock64();
  if((int)threadIdx.x < active)
    for(int r = 0; r < reps; ++r)
      {
        if constexpr(OP == 0)
          x = rsqrt(x), x.e = 0; // x in [0.5, 2) again
        else if constexpr(OP == 1)
          x = mul(x, y), x.e = 0;
        else if constexpr(OP == 2)
          acc_fma(acc, x, y, r & 1), x.m[0] ^= acc.w[1];
        else if constexpr(OP == 3)
          x = acc_result(acc), acc.w[2] ^= x.m[1], x.e = 0;
        else if constexpr(OP == 4)
          x = rcp(x), x.e = 0;
        else if constexpr(OP == 5)
          x = add(x, y), x.e = 0;
        else if constexpr(OP == 6)
          x = sqrt(x), x.e = 0;
      }
  const unsigned long long t1 = wall_clock64();
  if(threadIdx.x == 0)
    out[0] = t1 - t0;
  sink[threadIdx.x] = x.m[0] ^ acc.w[0];
}
template <int NL, int OP> double run(const char *name, int active)
{
  unsigned long long *d;
  uint32_t *s;
  hipMalloc(&d, 8);
  hipMalloc(&s, 4 * 64);
  const int reps = 200;
  k<NL, OP><<<1, 64>>>(reps, d, s, active);
  k<NL, OP><<<1, 64>>>(reps, d, s, active);
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double us = (double)h / 100.0 / reps; // 100 MHz
  printf("NL=%2d  %-28s lanes active %2d : %7.3f us per dependent operation\n", NL, name, active, us);
  hipFree(d);
  hipFree(s);
  return us;
}
template <int NL> void all()
{
  for(int active : {1, 64})
    {
      run<NL, 0>("rsqrt", active);
      run<NL, 6>("sqrt", active);
      run<NL, 4>("rcp", active);
      run<NL, 1>("mul", active);
      run<NL, 2>("acc_fma (product + aligned add)", active);
      run<NL, 3>("acc_result (normalise)", active);
      run<NL, 5>("add", active);
    }
}
int main()
{
  all<18>();
  all<34>();
  return 0;
}
