// The record helpers of INTEGRATION.md §1 (put/get on an mpf_t, as a shim inside sdpb would
// write them for El::BigFloat::gmp_float), compiled against include/sdpb_hip.h with real GMP
// and run against the library: values set through sdpb_hip_set_array_mpf come back bit for bit
// through sdpb_hip_get_array_mpf.  Built and run by tests/test_abi.py (emulation build on CPU).
#include "sdpb_hip.h"

#include <gmp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int L64; // (max(p,53)+127)/64 + 1, El::gmp::num_limbs

static void put(const mpf_t x, unsigned long long *rec)
{
  const long n = std::labs(x->_mp_size);
  rec[0] = (unsigned long long)(long long)x->_mp_size;
  rec[1] = (unsigned long long)(long long)x->_mp_exp;
  for(long i = 0; i < L64; ++i)
    rec[2 + i] = i < n ? x->_mp_d[i] : 0;
}
static void get(const unsigned long long *rec, mpf_t x) // x initialised at the run's precision: _mp_prec + 1 = L64 limbs
{
  x->_mp_size = (int)(long long)rec[0];
  x->_mp_exp = (long)(long long)rec[1];
  std::copy(rec + 2, rec + 2 + std::labs(x->_mp_size), x->_mp_d);
}
#define CHECK(call)                                                                      \
  do                                                                                     \
    {                                                                                    \
      if((call) != 0)                                                                    \
        {                                                                                \
          std::fprintf(stderr, "%s failed: %s\n", #call, sdpb_hip_last_error(ctx));      \
          return 1;                                                                      \
        }                                                                                \
    }                                                                                    \
  while(0)

int main()
{
  const int precision = 512, N = 7;
  mpf_set_default_prec(precision);
  L64 = (std::max(precision, 53) + 127) / 64 + 1;
  const int dims[1] = {1}, num_points[1] = {3};
  sdpb_hip_ctx *ctx = nullptr;
  if(sdpb_hip_create(precision, 1, dims, num_points, N, -1, 0, 1, &ctx) != 0)
    {
      std::fprintf(stderr, "create failed: %s\n", sdpb_hip_last_error(nullptr));
      return 1;
    }
  if(sdpb_hip_limbs(ctx) / 2 + 1 != L64)
    return 2;
  gmp_randstate_t st;
  gmp_randinit_default(st);
  std::vector<unsigned long long> rec((size_t)N * (L64 + 2)), back(rec.size());
  mpf_t y[7], z;
  mpf_init(z);
  for(int i = 0; i < N; ++i)
    {
      mpf_init(y[i]);
      mpf_urandomb(y[i], st, precision);
      if(i % 2)
        mpf_neg(y[i], y[i]);
      if(i == 3)
        mpf_set_ui(y[i], 0);
      if(i == 4)
        mpf_mul_2exp(y[i], y[i], 1234);
      if(i == 5)
        mpf_div_2exp(y[i], y[i], 4321);
      put(y[i], &rec[(size_t)i * (L64 + 2)]);
    }
  CHECK(sdpb_hip_set_array_mpf(ctx, "y", 0, 0, L64, rec.data(), N));
  size_t count = 0;
  CHECK(sdpb_hip_get_array_mpf(ctx, "y", 0, 0, L64, back.data(), N, &count));
  if(count != (size_t)N)
    return 3;
  for(int i = 0; i < N; ++i)
    {
      get(&back[(size_t)i * (L64 + 2)], z);
      if(mpf_cmp(z, y[i]) != 0)
        {
          gmp_fprintf(stderr, "entry %d: %.40Fe != %.40Fe\n", i, z, y[i]);
          return 4;
        }
    }
  sdpb_hip_destroy(ctx);
  std::puts("mpf records round-trip exactly");
  return 0;
}
