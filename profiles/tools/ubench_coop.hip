// mw::wv::rsqrt (one wavefront, one number) against mw::rsqrt (one lane): bit-for-bit equality on random operands and the
// latency of a dependent chain of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdpb_amd/csrc profiles/tools/ubench_coop.hip -o profiles/tools/ubench_coop.bin
#include <hip/hip_runtime.h>
#include "mw_wave.hpp"
#include <cstdio>
using namespace mw;
__device__ uint32_t rnd(uint32_t &s)
{
  s ^= s << 13;
  s ^= s >> 17;
  s ^= s << 5;
  return s;
}
template <int NL> __device__ Mw<NL> random_positive(uint32_t &s, int mode)
{
  Mw<NL> v;
  for(int l = 0; l < NL; ++l)
    v.m[l] = mode == 1 ? 0u : (mode == 2 ? 0xffffffffu : rnd(s));
  v.m[NL - 1] |= 0x80000000u;
  v.e = (int32_t)(rnd(s) % 2001u) - 1000;
  v.neg = 0;
  return v;
}
// every lane draws the same operand stream (same seed); the owner rotates
template <int NL> __global__ void check(int cases, unsigned *bad, uint32_t *first)
{
  uint32_t s = 0x1234567u + blockIdx.x * 7919u;
  for(int c = 0; c < cases; ++c)
    {
      const Mw<NL> a = random_positive<NL>(s, c % 17 == 3 ? 1 : (c % 17 == 5 ? 2 : 0));
      const int owner = c % 64;
      Mw<NL> mine = a;
      if((int)threadIdx.x != owner) // only the owner holds the operand
        for(int l = 0; l < NL; ++l)
          mine.m[l] = 0xdeadbeefu;
      if((int)threadIdx.x != owner)
        mine.e = 77;
      Mw<NL> w = wv::rsqrt<NL>(mine, owner);
      Mw<NL> r = rsqrt<NL>(a);
      bool same = w.e == r.e && w.neg == r.neg;
      for(int l = 0; l < NL; ++l)
        same = same && w.m[l] == r.m[l];
      {
        Mw<NL> an = a, mn = mine; // the reciprocal also of negative numbers and of exact powers of two
        an.neg = mn.neg = (uint32_t)(c & 1);
        const Mw<NL> w2 = wv::rcp<NL>(mn, owner), r2 = rcp<NL>(an);
        bool same2 = w2.e == r2.e && w2.neg == r2.neg;
        for(int l = 0; l < NL; ++l)
          same2 = same2 && w2.m[l] == r2.m[l];
        const Mw<NL> w3 = wv::sqrt<NL>(mine, owner), r3 = sqrt<NL>(a);
        bool same3 = w3.e == r3.e && w3.neg == r3.neg;
        for(int l = 0; l < NL; ++l)
          same3 = same3 && w3.m[l] == r3.m[l];
        if(!same2 || !same3)
          {
            same = false;
            if(!same2)
              w = w2, r = r2;
            else
              w = w3, r = r3;
          }
      }
      if(!same && atomicAdd(bad, 1u) == 0 && threadIdx.x == 0)
        {
          first[0] = (uint32_t)c;
          first[1] = (uint32_t)w.e;
          first[2] = (uint32_t)r.e;
          for(int l = 0; l < 4; ++l)
            first[3 + l] = w.m[NL - 1 - l], first[7 + l] = r.m[NL - 1 - l];
          for(int l = 0; l < 4; ++l)
            first[11 + l] = w.m[l], first[15 + l] = r.m[l];
        }
    }
}
template <int NL, int WAVE> __global__ void chain(int reps, unsigned long long *out, uint32_t *sink)
{
  uint32_t s = 99;
  Mw<NL> x = random_positive<NL>(s, 0);
  x.e = 0;
  const unsigned long long t0 = wall_clock64();
  for(int r = 0; r < reps; ++r)
    {
      if constexpr(WAVE == 1)
        x = wv::rsqrt<NL>(x, r & 63);
      else if constexpr(WAVE == 3)
        x = wv::rcp<NL>(x, r & 63);
      else if constexpr(WAVE == 2)
        {
          if(threadIdx.x == 0)
            x = rcp<NL>(x);
        }
      else if(threadIdx.x == 0)
        x = rsqrt<NL>(x);
      x.e = 0;
    }
  const unsigned long long t1 = wall_clock64();
  if(threadIdx.x == 0)
    out[0] = t1 - t0;
  sink[threadIdx.x] = x.m[0];
}
template <int NL> int one()
{
  unsigned *bad;
  uint32_t *first, *sink;
  unsigned long long *d;
  hipMalloc(&bad, 4);
  hipMalloc(&first, 4 * 32);
  hipMalloc(&sink, 4 * 64);
  hipMalloc(&d, 8);
  hipMemset(bad, 0, 4);
  const int cases = 2000, blocks = 64;
  check<NL><<<blocks, 64>>>(cases, bad, first);
  unsigned hb = 0;
  uint32_t hf[32];
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  hipMemcpy(hf, first, 4 * 32, hipMemcpyDeviceToHost);
  printf("NL=%2d  %d x %d operands: %u lane-results of rsqrt / rcp / sqrt differ from the one-lane functions%s\n", NL, blocks, cases, hb, hb ? "  <-- MISMATCH" : " (bit-identical)");
  if(hb)
    {
      printf("   first: case %u  e %d vs %d  top %08x %08x %08x %08x vs %08x %08x %08x %08x  low %08x %08x %08x %08x vs %08x %08x %08x %08x\n", hf[0], (int)hf[1],
             (int)hf[2], hf[3], hf[4], hf[5], hf[6], hf[7], hf[8], hf[9], hf[10], hf[11], hf[12], hf[13], hf[14], hf[15], hf[16], hf[17], hf[18]);
    }
  double us[4];
  for(int w = 0; w < 4; ++w)
    {
      const int reps = 200;
      unsigned long long h = 0;
      for(int rep = 0; rep < 2; ++rep)
        {
          if(w == 1)
            chain<NL, 1><<<1, 64>>>(reps, d, sink);
          else if(w == 2)
            chain<NL, 2><<<1, 64>>>(reps, d, sink);
          else if(w == 3)
            chain<NL, 3><<<1, 64>>>(reps, d, sink);
          else
            chain<NL, 0><<<1, 64>>>(reps, d, sink);
        }
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      us[w] = (double)h / 100.0 / reps;
    }
  printf("NL=%2d  dependent rsqrt: one lane %7.3f us, one wavefront %7.3f us (x%.2f)\n", NL, us[0], us[1], us[0] / us[1]);
  printf("NL=%2d  dependent rcp:   one lane %7.3f us, one wavefront %7.3f us (x%.2f)\n", NL, us[2], us[3], us[2] / us[3]);
  return hb != 0;
}
int main()
{
  int rc = 0;
  rc |= one<18>();
  rc |= one<26>();
  rc |= one<34>();
  rc |= one<50>();
  rc |= one<6>();
  rc |= one<10>();
  return rc;
}
