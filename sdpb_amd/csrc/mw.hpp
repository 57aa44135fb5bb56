// mw.hpp — fixed-width multi-word floating point for gfx950 (and the host side of
// the same library).  This is the number format that replaces El::BigFloat
// (GMP mpf; reference: src/sdpb_util/Environment.cxx:29-36 sets its precision) on
// the device.
//
//   value = (-1)^neg * (M / 2^(32*NL)) * 2^e ,   M = sum m[i] 2^(32 i)
//
// M is bit-normalised (top bit of m[NL-1] set) unless the value is zero, so the
// precision is a constant 32*NL bits (GMP's mpf is limb-normalised and fluctuates
// between 64*(l-1)+1 and 64*(l+1) bits).  All operations truncate toward zero in
// magnitude like mpf.  Limbs are 32-bit because the CDNA4 integer multiplier is
// v_mad_u64_u32 (32x32+64 -> 64, quarter rate); every limb loop is fully unrolled
// with compile-time indices so mantissas live in VGPRs, never in scratch.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MW_HD __host__ __device__ __forceinline__
#else
#define MW_HD inline __attribute__((always_inline))
#endif

namespace mw
{
constexpr int32_t EZERO = -(1 << 29); // exponent tag of the value zero
constexpr uint32_t EBIAS = 1u << 30;  // header = sign<<31 | (e + EBIAS)

template <int NL> struct Mw
{
  uint32_t m[NL];
  int32_t e;
  uint32_t neg;
};

// ---- 96-bit column accumulator for product scanning ------------------------
// acc(lo:64, hi:32) += a*b
#if defined(__HIP_DEVICE_COMPILE__)
#define MW_MAC(lo, hi, a, b)                                                   \
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, " \
               "0, %1, vcc"                                                    \
               : "+v"(lo), "+v"(hi)                                            \
               : "v"(a), "v"(b)                                                \
               : "vcc")
#else
#define MW_MAC(lo, hi, a, b)                                                   \
  do                                                                           \
    {                                                                          \
      const uint64_t p__ = (uint64_t)(a) * (uint64_t)(b);                      \
      lo += p__;                                                               \
      hi += (lo < p__) ? 1u : 0u;                                              \
    }                                                                          \
  while(0)
#endif

// four MACs in one asm statement (the compiler pads every asm statement with an s_nop;
// grouping keeps the integer pipe busy back to back)
#if defined(__HIP_DEVICE_COMPILE__)
#define MW_MAC4(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3)                        \
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, " \
               "0, %1, vcc\n\t"                                              \
               "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, " \
               "0, %1, vcc\n\t"                                              \
               "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, " \
               "0, %1, vcc\n\t"                                              \
               "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, " \
               "0, %1, vcc"                                                    \
               : "+v"(lo), "+v"(hi)                                            \
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2),         \
                 "v"(a3), "v"(b3)                                              \
               : "vcc")
#else
#define MW_MAC4(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3)                        \
  do                                                                           \
    {                                                                          \
      MW_MAC(lo, hi, a0, b0);                                                  \
      MW_MAC(lo, hi, a1, b1);                                                  \
      MW_MAC(lo, hi, a2, b2);                                                  \
      MW_MAC(lo, hi, a3, b3);                                                  \
    }                                                                          \
  while(0)
#endif

// two .. eight MACs in one statement (every asm statement costs a wait state: one statement per column)
#if defined(__HIP_DEVICE_COMPILE__)
#define MW_MAC_ASM1(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MW_MAC2(lo, hi, a0, b0, a1, b1) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc")
#define MW_MAC3(lo, hi, a0, b0, a1, b1, a2, b2) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) MW_MAC_ASM1(6, 7) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2) : "vcc")
#define MW_MAC5(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) MW_MAC_ASM1(6, 7) MW_MAC_ASM1(8, 9) MW_MAC_ASM1(10, 11) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4) : "vcc")
#define MW_MAC6(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) MW_MAC_ASM1(6, 7) MW_MAC_ASM1(8, 9) MW_MAC_ASM1(10, 11) MW_MAC_ASM1(12, 13) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5) : "vcc")
#define MW_MAC7(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) MW_MAC_ASM1(6, 7) MW_MAC_ASM1(8, 9) MW_MAC_ASM1(10, 11) MW_MAC_ASM1(12, 13) MW_MAC_ASM1(14, 15) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6) : "vcc")
#define MW_MAC8(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7) \
  asm volatile(MW_MAC_ASM1(2, 3) MW_MAC_ASM1(4, 5) MW_MAC_ASM1(6, 7) MW_MAC_ASM1(8, 9) MW_MAC_ASM1(10, 11) MW_MAC_ASM1(12, 13) MW_MAC_ASM1(14, 15) MW_MAC_ASM1(16, 17) : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7) : "vcc")
#else
#define MW_MAC2(lo, hi, a0, b0, a1, b1) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); } while(0)
#define MW_MAC3(lo, hi, a0, b0, a1, b1, a2, b2) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); MW_MAC(lo, hi, a2, b2); } while(0)
#define MW_MAC5(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); MW_MAC(lo, hi, a2, b2); MW_MAC(lo, hi, a3, b3); MW_MAC(lo, hi, a4, b4); } while(0)
#define MW_MAC6(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); MW_MAC(lo, hi, a2, b2); MW_MAC(lo, hi, a3, b3); MW_MAC(lo, hi, a4, b4); MW_MAC(lo, hi, a5, b5); } while(0)
#define MW_MAC7(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); MW_MAC(lo, hi, a2, b2); MW_MAC(lo, hi, a3, b3); MW_MAC(lo, hi, a4, b4); MW_MAC(lo, hi, a5, b5); MW_MAC(lo, hi, a6, b6); } while(0)
#define MW_MAC8(lo, hi, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7) \
  do { MW_MAC(lo, hi, a0, b0); MW_MAC(lo, hi, a1, b1); MW_MAC(lo, hi, a2, b2); MW_MAC(lo, hi, a3, b3); MW_MAC(lo, hi, a4, b4); MW_MAC(lo, hi, a5, b5); MW_MAC(lo, hi, a6, b6); MW_MAC(lo, hi, a7, b7); } while(0)
#endif

// column k of a product: acc += sum_{i=i0..i1} a[i]*b[k-i], compile-time bounds
template <int I0, int I1, int K, class A, class B> MW_HD void mac_column(uint64_t &lo, uint32_t &hi, const A &a, const B &b)
{
  constexpr int CNT = I1 - I0 + 1;
  if constexpr(CNT >= 8)
    {
      MW_MAC8(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2], a[I0 + 3], b[K - I0 - 3], a[I0 + 4], b[K - I0 - 4], a[I0 + 5], b[K - I0 - 5], a[I0 + 6], b[K - I0 - 6], a[I0 + 7], b[K - I0 - 7]);
      mac_column<I0 + 8, I1, K>(lo, hi, a, b);
    }
  else if constexpr(CNT == 7)
    MW_MAC7(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2], a[I0 + 3], b[K - I0 - 3], a[I0 + 4], b[K - I0 - 4], a[I0 + 5], b[K - I0 - 5], a[I0 + 6], b[K - I0 - 6]);
  else if constexpr(CNT == 6)
    MW_MAC6(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2], a[I0 + 3], b[K - I0 - 3], a[I0 + 4], b[K - I0 - 4], a[I0 + 5], b[K - I0 - 5]);
  else if constexpr(CNT == 5)
    MW_MAC5(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2], a[I0 + 3], b[K - I0 - 3], a[I0 + 4], b[K - I0 - 4]);
  else if constexpr(CNT == 4)
    MW_MAC4(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2], a[I0 + 3], b[K - I0 - 3]);
  else if constexpr(CNT == 3)
    MW_MAC3(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1], a[I0 + 2], b[K - I0 - 2]);
  else if constexpr(CNT == 2)
    MW_MAC2(lo, hi, a[I0 + 0], b[K - I0 - 0], a[I0 + 1], b[K - I0 - 1]);
  else if constexpr(CNT == 1)
    MW_MAC(lo, hi, a[I0], b[K - I0]);
}

// ---- carry chains that span statements -----------------------------------------
// Carry: a carry flag that lives across statements (on the device a lane mask in an SGPR
// pair, so that chains can be interleaved with the VCC-based MAC statements above).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long Carry;
// cy = (bit != 0)
#define MW_CY_SET(cy, bit) asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(cy) : "v"(bit))
// d += x, cy = carry out
#define MW_ADD_CO(d, x, cy) asm("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(d), "=s"(cy) : "v"(x))
// d += x + cy, cy = carry out
#define MW_ADDC(d, x, cy) asm("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(d), "+s"(cy) : "v"(x))
#else
typedef uint32_t Carry;
#define MW_CY_SET(cy, bit) cy = ((bit) != 0) ? 1u : 0u
#define MW_ADD_CO(d, x, cy)                                                    \
  do                                                                           \
    {                                                                          \
      const uint64_t s__ = (uint64_t)(d) + (uint64_t)(x);                      \
      d = (uint32_t)s__;                                                       \
      cy = (uint32_t)(s__ >> 32);                                              \
    }                                                                          \
  while(0)
#define MW_ADDC(d, x, cy)                                                      \
  do                                                                           \
    {                                                                          \
      const uint64_t s__ = (uint64_t)(d) + (uint64_t)(x) + (uint64_t)(cy);     \
      d = (uint32_t)s__;                                                       \
      cy = (uint32_t)(s__ >> 32);                                              \
    }                                                                          \
  while(0)
#endif

MW_HD uint32_t clz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__clz((int)x);
#else
  return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}
// (hi:lo) << c, upper word; 0 <= c <= 31
MW_HD uint32_t funnel_l(uint32_t hi, uint32_t lo, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return c ? __builtin_amdgcn_alignbit(hi, lo, 32u - c) : hi; // v_alignbit_b32
#else
  return c ? ((hi << c) | (lo >> (32 - c))) : hi;
#endif
}
// (hi:lo) >> c, lower word; 0 <= c <= 31
MW_HD uint32_t funnel_r(uint32_t hi, uint32_t lo, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, c); // one v_alignbit_b32; a shift of 0 returns lo
#else
  return c ? ((lo >> c) | (hi << (32 - c))) : lo;
#endif
}
// true if the predicate holds in any active lane of the wavefront: lets the log-step
// shift networks skip the stages that no lane needs (the host build evaluates its own lane)
MW_HD bool any_lane(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(p) != 0;
#else
  return p;
#endif
}

template <int NL> MW_HD Mw<NL> zero()
{
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = 0;
  r.e = EZERO;
  r.neg = 0;
  return r;
}
template <int NL> MW_HD bool is_zero(const Mw<NL> &a) { return a.e == EZERO; }

template <int NL> MW_HD Mw<NL> neg(Mw<NL> a)
{
  if(a.e != EZERO)
    a.neg ^= 1u;
  return a;
}
template <int NL> MW_HD Mw<NL> abs(Mw<NL> a)
{
  a.neg = 0;
  return a;
}
// a * 2^k
template <int NL> MW_HD Mw<NL> mul_2exp(Mw<NL> a, int k)
{
  if(a.e != EZERO)
    a.e += k;
  return a;
}

// small non-negative integer
template <int NL> MW_HD Mw<NL> from_u32(uint32_t v)
{
  Mw<NL> r = zero<NL>();
  if(v == 0)
    return r;
  const uint32_t c = clz32(v);
  r.m[NL - 1] = v << c;
  r.e = 32 - (int)c;
  return r;
}
template <int NL> MW_HD Mw<NL> from_i32(int32_t v)
{
  Mw<NL> r = from_u32<NL>(v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v);
  r.neg = v < 0;
  return r;
}

// from a finite double (exact)
template <int NL> MW_HD Mw<NL> from_double(double d)
{
  Mw<NL> r = zero<NL>();
  if(d == 0.0)
    return r;
  union
  {
    double d;
    uint64_t u;
  } cv;
  cv.d = d;
  const uint64_t bits = cv.u;
  r.neg = (uint32_t)(bits >> 63);
  int ex = (int)((bits >> 52) & 0x7ff);
  uint64_t man = bits & 0xfffffffffffffull;
  if(ex == 0)
    ex = 1; // subnormal
  else
    man |= 1ull << 52;
  // value = man * 2^(ex-1075); normalise man to bit 63
  int lz = 0;
  while(!(man >> 63))
    {
      man <<= 1;
      ++lz;
    }
  r.m[NL - 1] = (uint32_t)(man >> 32);
  if(NL > 1)
    r.m[NL > 1 ? NL - 2 : 0] = (uint32_t)man;
  r.e = ex - 1075 + 64 - lz;
  return r;
}
// nearest-ish double (truncated); no overflow handling beyond inf/0
template <int NL> MW_HD double to_double(const Mw<NL> &a)
{
  if(a.e == EZERO)
    return 0.0;
  uint64_t top = ((uint64_t)a.m[NL - 1] << 32) | (NL > 1 ? a.m[NL > 1 ? NL - 2 : 0] : 0u);
  // top has bit 63 set; value = top/2^64 * 2^e
  int e = a.e;
  if(e > 1023)
    return a.neg ? -1e308 * 10 : 1e308 * 10;
  if(e < -1020)
    return 0.0;
  union
  {
    double d;
    uint64_t u;
  } cv;
  cv.u = ((uint64_t)(e - 1 + 1023) << 52) | ((top >> 11) & 0xfffffffffffffull);
  return a.neg ? -cv.d : cv.d;
}

// compare magnitudes: -1,0,1
template <int NL> MW_HD int cmp_abs(const Mw<NL> &a, const Mw<NL> &b)
{
  if(a.e != b.e)
    return a.e < b.e ? -1 : 1;
  int r = 0;
#pragma unroll
  for(int i = NL - 1; i >= 0; --i)
    if(r == 0 && a.m[i] != b.m[i])
      r = a.m[i] < b.m[i] ? -1 : 1;
  return r;
}
template <int NL> MW_HD int cmp(const Mw<NL> &a, const Mw<NL> &b)
{
  const bool az = a.e == EZERO, bz = b.e == EZERO;
  if(az && bz)
    return 0;
  if(az)
    return b.neg ? 1 : -1;
  if(bz)
    return a.neg ? -1 : 1;
  if(a.neg != b.neg)
    return a.neg ? -1 : 1;
  const int c = cmp_abs(a, b);
  return a.neg ? -c : c;
}
template <int NL> MW_HD bool lt(const Mw<NL> &a, const Mw<NL> &b) { return cmp(a, b) < 0; }
template <int NL> MW_HD bool gt(const Mw<NL> &a, const Mw<NL> &b) { return cmp(a, b) > 0; }
template <int NL> MW_HD Mw<NL> max(const Mw<NL> &a, const Mw<NL> &b) { return cmp(a, b) < 0 ? b : a; }
template <int NL> MW_HD Mw<NL> min(const Mw<NL> &a, const Mw<NL> &b) { return cmp(a, b) < 0 ? a : b; }

// ---- multiplication: short product (top NL limbs + one guard column) -------
// The dropped low columns contribute < NL * 2^-32 ulp, far inside mpf's own
// 1-ulp truncation.
template <int NL, int K> struct MulColumns
{
  // columns K .. 2NL-2 of a*b; r[k-(NL-1)] receives limb k
  static MW_HD void run(const uint32_t (&a)[NL], const uint32_t (&b)[NL], uint64_t &lo, uint32_t &hi, uint32_t (&r)[NL + 1])
  {
    constexpr int I0 = K - (NL - 1) > 0 ? K - (NL - 1) : 0, I1 = K < NL - 1 ? K : NL - 1;
    mac_column<I0, I1, K>(lo, hi, a, b);
    r[K - (NL - 1)] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
    if constexpr(K < 2 * NL - 2)
      MulColumns<NL, K + 1>::run(a, b, lo, hi, r);
  }
};
template <int NL> MW_HD Mw<NL> mul(const Mw<NL> &a, const Mw<NL> &b)
{
  if(a.e == EZERO || b.e == EZERO)
    return zero<NL>();
  uint32_t r[NL + 1]; // r[0] = product limb NL-1 (guard for the 1-bit shift), r[k] = limb NL-1+k
  uint64_t lo = 0;
  uint32_t hi = 0;
  // guard column k = NL-2: only its carry is kept
  if constexpr(NL >= 2)
    {
      mac_column<0, NL - 2, NL - 2>(lo, hi, a.m, b.m);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
  MulColumns<NL, NL - 1>::run(a.m, b.m, lo, hi, r);
  // top limb (product limb 2NL-1) = remaining lo
  Mw<NL> out;
  const uint32_t top = (uint32_t)lo;
  out.neg = a.neg ^ b.neg;
  if(top >> 31)
    {
#pragma unroll
      for(int i = 0; i < NL - 1; ++i)
        out.m[i] = r[i + 1];
      out.m[NL - 1] = top;
      out.e = a.e + b.e;
    }
  else
    {
#pragma unroll
      for(int i = 0; i < NL - 1; ++i)
        out.m[i] = (r[i + 1] << 1) | (r[i] >> 31);
      out.m[NL - 1] = (top << 1) | (r[NL - 1] >> 31);
      out.e = a.e + b.e - 1;
    }
  return out;
}
template <int NL> MW_HD Mw<NL> sqr(const Mw<NL> &a) { return mul(a, a); }

// ---- addition / subtraction ------------------------------------------------
// helpers on NL+1-limb working arrays (index NL = most significant)
template <int W, bool SKIP = false> MW_HD void shr_limbs(uint32_t (&x)[W], uint32_t q)
{
#pragma unroll
  for(int s = 1; s < W; s <<= 1)
    {
      const bool on = (q & (uint32_t)s) != 0;
      if constexpr(SKIP) // most alignments are short: skip the stages no lane of the wavefront needs
        if(!any_lane(on))
          continue;
#pragma unroll
      for(int i = 0; i < W; ++i)
        {
          const uint32_t from = (i + s < W) ? x[i + s < W ? i + s : 0] : 0u;
          x[i] = on ? from : x[i];
        }
    }
}
template <int W> MW_HD void shl_limbs(uint32_t (&x)[W], uint32_t q)
{
#pragma unroll
  for(int s = 1; s < W; s <<= 1)
    {
      const bool on = (q & (uint32_t)s) != 0;
#pragma unroll
      for(int i = W - 1; i >= 0; --i)
        {
          const uint32_t from = (i - s >= 0) ? x[i - s >= 0 ? i - s : 0] : 0u;
          x[i] = on ? from : x[i];
        }
    }
}
template <int W> MW_HD void shr_bits(uint32_t (&x)[W], uint32_t c)
{
#pragma unroll
  for(int i = 0; i < W - 1; ++i)
    x[i] = funnel_r(x[i + 1], x[i], c);
  x[W - 1] = x[W - 1] >> c; // c in 0..31
}
template <int W> MW_HD void shl_bits(uint32_t (&x)[W], uint32_t c)
{
#pragma unroll
  for(int i = W - 1; i >= 1; --i)
    x[i] = funnel_l(x[i], x[i - 1], c);
  x[0] = x[0] << c;
}

// bit (i - BASE) of the result = (w[i] != 0) for i in [I0, I1], as a balanced OR tree
template <int I0, int I1, int BASE, int W> MW_HD uint32_t nonzero_mask(const uint32_t (&w)[W])
{
  if constexpr(I0 == I1)
    return (w[I0] != 0 ? 1u : 0u) << (I0 - BASE);
  else
    {
      constexpr int MID = (I0 + I1) / 2;
      return nonzero_mask<I0, MID, BASE, W>(w) | nonzero_mask<MID + 1, I1, BASE, W>(w);
    }
}
// index of the most significant non-zero limb, -1 if all are zero (log depth: this sits
// on the dependent path of every subtraction)
template <int W> MW_HD int top_nonzero(const uint32_t (&w)[W])
{
  static_assert(W <= 128, "four 32-bit masks");
  if constexpr(W <= 32)
    {
      const uint32_t m = nonzero_mask<0, W - 1, 0, W>(w);
      return m ? 31 - (int)clz32(m) : -1;
    }
  else if constexpr(W <= 64)
    {
      const uint32_t mh = nonzero_mask<32, W - 1, 32, W>(w), ml = nonzero_mask<0, 31, 0, W>(w);
      return mh ? 63 - (int)clz32(mh) : (ml ? 31 - (int)clz32(ml) : -1);
    }
  else
    {
      // the widths above 1536 bits (--precision 2048: 66 limbs and their guard limbs)
      uint32_t m3 = 0;
      if constexpr(W > 96)
        m3 = nonzero_mask<96, W - 1, 96, W>(w);
      const uint32_t m2 = nonzero_mask<64, (W > 96 ? 95 : W - 1), 64, W>(w);
      const uint32_t m1 = nonzero_mask<32, 63, 32, W>(w), m0 = nonzero_mask<0, 31, 0, W>(w);
      return m3 ? 127 - (int)clz32(m3) : (m2 ? 95 - (int)clz32(m2) : (m1 ? 63 - (int)clz32(m1) : (m0 ? 31 - (int)clz32(m0) : -1)));
    }
}

// Normalise a (NL+1)-limb magnitude w (value w/2^(32(NL+1)) * 2^e) into r.
template <int NL> MW_HD Mw<NL> normalize_w(uint32_t (&w)[NL + 1], int32_t e, uint32_t neg)
{
  const int top = top_nonzero<NL + 1>(w);
  if(top < 0)
    return zero<NL>();
  const uint32_t zl = (uint32_t)(NL - top);
  shl_limbs<NL + 1>(w, zl);
  const uint32_t c = clz32(w[NL]);
  shl_bits<NL + 1>(w, c);
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = w[i + 1];
  r.e = e - (int32_t)(32u * zl + c);
  r.neg = neg;
  return r;
}

template <int NL> MW_HD Mw<NL> add(const Mw<NL> &a_in, const Mw<NL> &b_in)
{
  if(a_in.e == EZERO)
    return b_in;
  if(b_in.e == EZERO)
    return a_in;
  // order by exponent only (no limb-by-limb magnitude compare on the dependent path);
  // with equal exponents a subtraction may come out negative and is then negated
  const bool swap = a_in.e < b_in.e;
  const Mw<NL> &a = swap ? b_in : a_in;
  const Mw<NL> &b = swap ? a_in : b_in;
  const uint32_t d = (uint32_t)(a.e - b.e);
  if(d >= 32u * (NL + 1))
    return a;
  uint32_t x[NL + 1], y[NL + 1];
  x[0] = 0;
  y[0] = 0;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    {
      x[i + 1] = a.m[i];
      y[i + 1] = b.m[i];
    }
  shr_limbs<NL + 1>(y, d >> 5);
  shr_bits<NL + 1>(y, d & 31u);
  if(a.neg == b.neg)
    {
      uint32_t carry = 0;
#pragma unroll
      for(int i = 0; i <= NL; ++i)
        {
          const uint64_t s = (uint64_t)x[i] + y[i] + carry;
          x[i] = (uint32_t)s;
          carry = (uint32_t)(s >> 32);
        }
      Mw<NL> r;
      r.neg = a.neg;
      if(carry)
        {
#pragma unroll
          for(int i = 0; i < NL - 1; ++i)
            r.m[i] = (x[i + 1] >> 1) | (x[i + 2] << 31);
          r.m[NL - 1] = (x[NL] >> 1) | 0x80000000u;
          r.e = a.e + 1;
        }
      else
        {
#pragma unroll
          for(int i = 0; i < NL; ++i)
            r.m[i] = x[i + 1];
          r.e = a.e;
        }
      return r;
    }
  uint32_t borrow = 0;
#pragma unroll
  for(int i = 0; i <= NL; ++i)
    {
      const uint64_t s = (uint64_t)x[i] - y[i] - borrow;
      x[i] = (uint32_t)s;
      borrow = (uint32_t)(s >> 63);
    }
  uint32_t sign = a.neg;
  if(borrow)
    {
      // only possible when d == 0 and |b| > |a|: result = -(two's complement)
      uint32_t carry = 1;
#pragma unroll
      for(int i = 0; i <= NL; ++i)
        {
          const uint64_t s = (uint64_t)(~x[i]) + carry;
          x[i] = (uint32_t)s;
          carry = (uint32_t)(s >> 32);
        }
      sign ^= 1u;
    }
  return normalize_w<NL>(x, a.e, sign);
}
template <int NL> MW_HD Mw<NL> sub(const Mw<NL> &a, const Mw<NL> &b) { return add(a, neg(b)); }

// acc + a*b, acc - a*b (two roundings, like mpf's `acc += a*b`)
template <int NL> MW_HD Mw<NL> fma(const Mw<NL> &a, const Mw<NL> &b, const Mw<NL> &acc)
{
  return add(acc, mul(a, b));
}
template <int NL> MW_HD Mw<NL> fms(const Mw<NL> &a, const Mw<NL> &b, const Mw<NL> &acc)
{
  return add(acc, neg(mul(a, b)));
}

// ---- dot-product accumulator ------------------------------------------------
// Sum of products without per-term normalisation: the running sum is a signed
// (NL+2)-limb two's-complement integer W with a window exponent etop,
//     value = W / 2^(32 (NL+1)) * 2^etop ,
// i.e. NL+1 limbs of fraction below the largest term seen so far and one limb of
// headroom (2^31 terms).  A term costs its product columns plus ONE aligned add
// (shift network + carry chain); the expensive normalisation (leading-zero search,
// left shift, magnitude ordering) happens once, in acc_result.  Terms are truncated
// at 2^-(32(NL+1)) relative to the largest term, the rounding regime of a sequential
// `acc += a*b` in mpf.
template <int NL> struct Acc
{
  uint32_t w[NL + 2];
  int32_t etop; // EZERO while empty
};
template <int NL> MW_HD Acc<NL> acc_zero()
{
  Acc<NL> a;
#pragma unroll
  for(int i = 0; i < NL + 2; ++i)
    a.w[i] = 0;
  a.etop = EZERO;
  return a;
}
// arithmetic right shift of a two's-complement array by q limbs / c bits
template <int W> MW_HD void sar_limbs(uint32_t (&x)[W], uint32_t q)
{
  const uint32_t fill = 0u - (x[W - 1] >> 31);
#pragma unroll
  for(int s = 1; s < W; s <<= 1)
    {
      const bool on = (q & (uint32_t)s) != 0;
#pragma unroll
      for(int i = 0; i < W; ++i)
        {
          const uint32_t from = (i + s < W) ? x[i + s < W ? i + s : 0] : fill;
          x[i] = on ? from : x[i];
        }
    }
}
template <int W> MW_HD void sar_bits(uint32_t (&x)[W], uint32_t c)
{
  const uint32_t fill = 0u - (x[W - 1] >> 31);
#pragma unroll
  for(int i = 0; i < W - 1; ++i)
    x[i] = funnel_r(x[i + 1], x[i], c);
  x[W - 1] = funnel_r(fill, x[W - 1], c);
}
// acc += (-1)^negate * P / 2^(32(NL+1)) * 2^e with P an (NL+1)-limb magnitude
template <int NL> MW_HD void acc_add_raw(Acc<NL> &acc, const uint32_t (&P)[NL + 1], int32_t e, uint32_t negative)
{
  if(e > acc.etop)
    {
      if(acc.etop != EZERO)
        {
          const uint32_t up = (uint32_t)(e - acc.etop);
          if(up >= 32u * (NL + 2))
            {
              const uint32_t fill = 0u - (acc.w[NL + 1] >> 31);
#pragma unroll
              for(int i = 0; i < NL + 2; ++i)
                acc.w[i] = fill; // -1 ulp or 0: what an arithmetic shift leaves
            }
          else
            {
              sar_limbs<NL + 2>(acc.w, up >> 5);
              sar_bits<NL + 2>(acc.w, up & 31u);
            }
        }
      acc.etop = e;
    }
  const uint32_t d = (uint32_t)(acc.etop - e);
  if(d >= 32u * (NL + 1))
    return;
  uint32_t x[NL + 2];
#pragma unroll
  for(int i = 0; i <= NL; ++i)
    x[i] = P[i];
  x[NL + 1] = 0;
  shr_limbs<NL + 2, true>(x, d >> 5);
  shr_bits<NL + 2>(x, d & 31u);
  // acc.w += negative ? -x : x  (two's complement: (x ^ mask) with the +1 as first carry)
  const uint32_t mask = 0u - negative;
  Carry cy;
  MW_CY_SET(cy, negative);
#pragma unroll
  for(int i = 0; i < NL + 2; ++i)
    {
      const uint32_t t = x[i] ^ mask;
      MW_ADDC(acc.w[i], t, cy);
    }
}
// acc += a*b (negate: acc -= a*b)
template <int NL> MW_HD void acc_fma(Acc<NL> &acc, const Mw<NL> &a, const Mw<NL> &b, uint32_t negate = 0)
{
  if(a.e == EZERO || b.e == EZERO)
    return;
  uint32_t r[NL + 1];
  uint64_t lo = 0;
  uint32_t hi = 0;
  if constexpr(NL >= 2)
    {
      mac_column<0, NL - 2, NL - 2>(lo, hi, a.m, b.m);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
  MulColumns<NL, NL - 1>::run(a.m, b.m, lo, hi, r);
  uint32_t P[NL + 1];
#pragma unroll
  for(int i = 0; i < NL; ++i)
    P[i] = r[i];
  P[NL] = (uint32_t)lo; // product limbs NL-1 .. 2NL-1: fraction in [1/4, 1)
  acc_add_raw<NL>(acc, P, a.e + b.e, a.neg ^ b.neg ^ (negate & 1u));
}
template <int NL> MW_HD void acc_fms(Acc<NL> &acc, const Mw<NL> &a, const Mw<NL> &b) { acc_fma(acc, a, b, 1u); }
// ---- raw terms: the pieces of acc_fma for sums that are formed ACROSS lanes ------------------------------
// A term is an (NL+1)-limb magnitude P, an exponent e and a sign: value = (-1)^neg P / 2^(32(NL+1)) 2^e (what
// acc_add_raw takes).  A workgroup that sums many terms per output lets every lane form ONE term, agrees on the
// largest exponent E of the output's terms, aligns every term to the window below E (term_align: exactly the shift
// acc_add_raw applies, so a term is truncated at 2^-(32(NL+1)) relative to the largest one) and adds the windows
// limb by limb as integers — associative, hence independent of the order and of the lane layout; carries are
// propagated once per output (kernels.hpp: the Q substitution).
template <int NL> MW_HD void term_mul(const Mw<NL> &a, const Mw<NL> &b, uint32_t (&P)[NL + 1], int32_t &e, uint32_t &neg)
{
  if(a.e == EZERO || b.e == EZERO)
    {
#pragma unroll
      for(int i = 0; i <= NL; ++i)
        P[i] = 0;
      e = EZERO;
      neg = 0;
      return;
    }
  uint32_t r[NL + 1];
  uint64_t lo = 0;
  uint32_t hi = 0;
  if constexpr(NL >= 2)
    {
      mac_column<0, NL - 2, NL - 2>(lo, hi, a.m, b.m);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
  MulColumns<NL, NL - 1>::run(a.m, b.m, lo, hi, r);
#pragma unroll
  for(int i = 0; i < NL; ++i)
    P[i] = r[i];
  P[NL] = (uint32_t)lo;
  e = a.e + b.e;
  neg = a.neg ^ b.neg;
}
// the (NL+2)-limb window of a term below the top exponent E >= e (magnitude; the sign is kept by the caller)
template <int NL> MW_HD void term_align(const uint32_t (&P)[NL + 1], int32_t e, int32_t E, uint32_t (&x)[NL + 2])
{
  const uint32_t d = (uint32_t)(E - e);
  const bool gone = e == EZERO || d >= 32u * (NL + 1);
#pragma unroll
  for(int i = 0; i <= NL; ++i)
    x[i] = gone ? 0u : P[i];
  x[NL + 1] = 0;
  const uint32_t dd = gone ? 0u : d;
  shr_limbs<NL + 2, true>(x, dd >> 5);
  shr_bits<NL + 2>(x, dd & 31u);
}
// acc += x (negate: acc -= x)
template <int NL> MW_HD void acc_add(Acc<NL> &acc, const Mw<NL> &x, uint32_t negate = 0)
{
  if(x.e == EZERO)
    return;
  uint32_t P[NL + 1];
  P[0] = 0;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    P[i + 1] = x.m[i];
  acc_add_raw<NL>(acc, P, x.e, x.neg ^ (negate & 1u));
}
// the accumulated sum as a normalised number (one truncation)
template <int NL> MW_HD Mw<NL> acc_result(const Acc<NL> &acc)
{
  if(acc.etop == EZERO)
    return zero<NL>();
  uint32_t w[NL + 2];
  const uint32_t negative = acc.w[NL + 1] >> 31;
  const uint32_t mask = 0u - negative;
  uint32_t carry = negative;
#pragma unroll
  for(int i = 0; i < NL + 2; ++i)
    {
      const uint64_t s = (uint64_t)(acc.w[i] ^ mask) + carry;
      w[i] = (uint32_t)s;
      carry = (uint32_t)(s >> 32);
    }
  // magnitude w / 2^(32(NL+2)) * 2^(etop+32): locate the top limb, shift, keep NL limbs
  const int top = top_nonzero<NL + 2>(w);
  if(top < 0)
    return zero<NL>();
  const uint32_t zl = (uint32_t)(NL + 1 - top);
  shl_limbs<NL + 2>(w, zl);
  const uint32_t c = clz32(w[NL + 1]);
  shl_bits<NL + 2>(w, c);
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = w[i + 2];
  r.e = acc.etop + 32 - (int32_t)(32u * zl + c);
  r.neg = negative;
  return r;
}

// ---- reciprocal, division, square root ---------------------------------------
// Newton iterations with precision doubling: the seed comes from fp64, every
// further step runs at (roughly) twice the limb count of the previous one, so a
// reciprocal costs ~3 full-width multiplications and an inverse square root ~4.5
// instead of 2 resp. 3 per iteration at full width.
template <int NH, int NL> MW_HD Mw<NH> narrow(const Mw<NL> &a)
{
  Mw<NH> r;
#pragma unroll
  for(int i = 0; i < NH; ++i)
    r.m[i] = a.m[NL - NH + i];
  r.e = a.e;
  r.neg = a.neg;
  return r;
}
template <int NL, int NH> MW_HD Mw<NL> widen(const Mw<NH> &a)
{
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL - NH; ++i)
    r.m[i] = 0;
#pragma unroll
  for(int i = 0; i < NH; ++i)
    r.m[NL - NH + i] = a.m[i];
  r.e = a.e;
  r.neg = a.neg;
  return r;
}

// 1/man for man in [0.5,1) (e = 0, positive), accurate to ~32*NL-3 bits
template <int NL> struct RcpMant
{
  static MW_HD Mw<NL> run(const Mw<NL> &man)
  {
    constexpr int NH = NL / 2 + 1;
    const Mw<NL> r0 = widen<NL, NH>(RcpMant<NH>::run(narrow<NH, NL>(man)));
    // r0 (2 - man r0): three dependent operations per level
    return mul(r0, sub(from_u32<NL>(2), mul(man, r0)));
  }
};
template <> struct RcpMant<3>
{
  static MW_HD Mw<3> run(const Mw<3> &man)
  {
    Mw<3> r = from_double<3>(1.0 / to_double(man));
    const Mw<3> one = from_u32<3>(1);
#pragma unroll
    for(int it = 0; it < 2; ++it)
      r = add(r, mul(r, sub(one, mul(man, r))));
    return r;
  }
};
template <> struct RcpMant<2>
{
  static MW_HD Mw<2> run(const Mw<2> &man)
  {
    Mw<2> r = from_double<2>(1.0 / to_double(man));
    return add(r, mul(r, sub(from_u32<2>(1), mul(man, r))));
  }
};
template <> struct RcpMant<1>
{
  static MW_HD Mw<1> run(const Mw<1> &man) { return from_double<1>(1.0 / to_double(man)); }
};

template <int NL> MW_HD Mw<NL> rcp_mant_fx(const Mw<NL> &man); // fixed-point Newton, below
template <int NL> MW_HD Mw<NL> rcp(const Mw<NL> &a)
{
  // caller guarantees a != 0
  Mw<NL> man = a;
  man.e = 0;
  man.neg = 0;
  Mw<NL> r;
  if constexpr(NL >= 4)
    r = rcp_mant_fx<NL>(man);
  else
    r = RcpMant<NL>::run(man);
  r.e -= a.e;
  r.neg = a.neg;
  return r;
}
template <int NL> MW_HD Mw<NL> div(const Mw<NL> &a, const Mw<NL> &b)
{
  if(a.e == EZERO)
    return a;
  // q = a*r, one correction step: q += r*(a - b*q)
  const Mw<NL> r = rcp(b);
  Mw<NL> q = mul(a, r);
  const Mw<NL> rem = sub(a, mul(b, q));
  return add(q, mul(r, rem));
}

MW_HD double host_device_sqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return ::sqrt(x);
#else
  return __builtin_sqrt(x);
#endif
}
// 1/sqrt(man) for man in [0.5,2) (e in {0,1}, positive)
template <int NL> struct RsqrtMant
{
  static MW_HD Mw<NL> run(const Mw<NL> &man)
  {
    constexpr int NH = NL / 2 + 1;
    const Mw<NL> r0 = widen<NL, NH>(RsqrtMant<NH>::run(narrow<NH, NL>(man)));
    // r0 (3 - man r0^2) / 2: four dependent operations per level
    const Mw<NL> t = sub(from_u32<NL>(3), mul(man, mul(r0, r0)));
    return mul_2exp(mul(r0, t), -1);
  }
};
template <> struct RsqrtMant<3>
{
  static MW_HD Mw<3> run(const Mw<3> &man)
  {
    Mw<3> r = from_double<3>(1.0 / host_device_sqrt(to_double(man)));
    const Mw<3> one = from_u32<3>(1);
#pragma unroll
    for(int it = 0; it < 2; ++it)
      r = add(r, mul_2exp(mul(r, sub(one, mul(man, mul(r, r)))), -1));
    return r;
  }
};
template <> struct RsqrtMant<2>
{
  static MW_HD Mw<2> run(const Mw<2> &man)
  {
    Mw<2> r = from_double<2>(1.0 / host_device_sqrt(to_double(man)));
    return add(r, mul_2exp(mul(r, sub(from_u32<2>(1), mul(man, mul(r, r)))), -1));
  }
};
template <> struct RsqrtMant<1>
{
  static MW_HD Mw<1> run(const Mw<1> &man) { return from_double<1>(1.0 / host_device_sqrt(to_double(man))); }
};

// ---- fixed-point Newton for the inverse square root -----------------------------
// The Mw iteration above spends most of its time aligning and normalising (an `add` costs
// as much as a `mul`).  Here every quantity has a compile-time binary point, so a level is
// two integer products + one more at half width and two carry chains, and the half-width
// iterate is never padded to full width:
//   y' = y + y (1 - x y^2)/2,   X, Y in "Q2" fixed point: value = limbs / 2^(32 L - 2).
// limbs [F, LA+LB) of a*b by product scanning; columns below F-2 are dropped (< 2^-26 of limb F)
template <int LA, int LB, int F, int K> struct FxColumns
{
  static MW_HD void run(const uint32_t (&a)[LA], const uint32_t (&b)[LB], uint64_t &lo, uint32_t &hi, uint32_t (&out)[LA + LB - F])
  {
    constexpr int I0 = K - (LB - 1) > 0 ? K - (LB - 1) : 0, I1 = K < LA - 1 ? K : LA - 1;
    mac_column<I0, I1, K>(lo, hi, a, b);
    if constexpr(K >= F)
      out[K - F] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
    if constexpr(K < LA + LB - 2)
      FxColumns<LA, LB, F, K + 1>::run(a, b, lo, hi, out);
    else
      out[LA + LB - 1 - F] = (uint32_t)lo;
  }
};
template <int LA, int LB, int F> MW_HD void fx_mul_from(const uint32_t (&a)[LA], const uint32_t (&b)[LB], uint32_t (&out)[LA + LB - F])
{
  constexpr int K0 = F - 2 > 0 ? F - 2 : 0;
  uint64_t lo = 0;
  uint32_t hi = 0;
  FxColumns<LA, LB, F, K0>::run(a, b, lo, hi, out);
}
// Y (L limbs, Q2) ~ 1/sqrt(x), x in [0.5, 2) given as X (NX limbs, Q2); error < 2^-(32L-8)
template <int NX, int L> struct RsqrtFx
{
  static MW_HD void run(const uint32_t (&X)[NX], uint32_t (&Y)[L])
  {
    if constexpr(L <= 2)
      {
        static_assert(L == 2, "seed width");
        const uint64_t xt = ((uint64_t)X[NX - 1] << 32) | X[NX - 2];
        const double xd = (double)xt * (1.0 / 4611686018427387904.0); // 2^-62
        const uint64_t y = (uint64_t)((1.0 / host_device_sqrt(xd)) * 4611686018427387904.0);
        Y[0] = (uint32_t)y;
        Y[1] = (uint32_t)(y >> 32);
      }
    else
      {
        constexpr int H = L / 2 + 1, T = L + 1 + 2 * H, LO = L + 3, LD = L - H + 4;
        static_assert(NX >= L + 1, "x needs one guard limb");
        uint32_t Yh[H];
        RsqrtFx<NX, H>::run(X, Yh);
        uint32_t S[2 * H]; // y^2 exactly, value = S / 2^(64H - 4)
        fx_mul_from<H, H, 0>(Yh, Yh, S);
        uint32_t Xs[L + 1];
#pragma unroll
        for(int i = 0; i <= L; ++i)
          Xs[i] = X[NX - (L + 1) + i];
        uint32_t D[LO]; // x y^2 with 1.0 = 2^(32 LO - 6)
        fx_mul_from<L + 1, 2 * H, T - LO>(Xs, S, D);
        // D <- 1 - x y^2 (two's complement), then sign + magnitude
        uint64_t bw = 0;
#pragma unroll
        for(int i = 0; i < LO; ++i)
          {
            const uint64_t one = (i == LO - 1) ? 0x04000000u : 0u;
            const uint64_t d = one - (uint64_t)D[i] - bw;
            D[i] = (uint32_t)d;
            bw = (d >> 63) & 1u;
          }
        const uint32_t negative = D[LO - 1] >> 31, mask = 0u - negative;
        uint64_t cy = negative;
        uint32_t Dm[LD]; // |1 - x y^2| < 2^-(32H-8): its limbs above LD are zero
#pragma unroll
        for(int i = 0; i < LD; ++i)
          {
            const uint64_t t = (uint64_t)(D[i] ^ mask) + cy;
            Dm[i] = (uint32_t)t;
            cy = t >> 32;
          }
        // y |D| / 2 at Y's scale: product limbs [H+2, H+LD) shifted left by 5 bits
        uint32_t Pm[LD - 2];
        fx_mul_from<H, LD, H + 2>(Yh, Dm, Pm);
        // Y = Yh 2^(32(L-H)) +/- C
        uint64_t c = 0;
#pragma unroll
        for(int i = 0; i < L; ++i)
          {
            const uint32_t base = i >= L - H ? Yh[i >= L - H ? i - (L - H) : 0] : 0u;
            uint32_t corr = 0;
            if(i <= L - H)
              corr = (Pm[i + 1 < LD - 2 ? i + 1 : LD - 3] << 5) | (Pm[i] >> 27);
            if(negative)
              {
                const uint64_t t = (uint64_t)base - corr - c;
                Y[i] = (uint32_t)t;
                c = (t >> 63) & 1u;
              }
            else
              {
                const uint64_t t = (uint64_t)base + corr + c;
                Y[i] = (uint32_t)t;
                c = t >> 32;
              }
          }
      }
  }
};
// 1/sqrt(man) for man in [0.5,2) (e in {0,1}, positive)
template <int NL> MW_HD Mw<NL> rsqrt_mant_fx(const Mw<NL> &man)
{
  constexpr int NX = NL + 2, L = NL + 1;
  uint32_t X[NX]; // x 2^(32 NX - 2) = M 2^(62 + e)
  const uint32_t sh = 2u - (uint32_t)man.e;
#pragma unroll
  for(int i = 0; i < NX; ++i)
    {
      const uint32_t lo = (i >= 2) ? man.m[i >= 2 ? i - 2 : 0] : 0u;
      const uint32_t up = (i >= 1 && i - 1 < NL) ? man.m[(i >= 1 && i - 1 < NL) ? i - 1 : 0] : 0u;
      X[i] = (lo >> sh) | (up << (32u - sh));
    }
  uint32_t Y[L];
  RsqrtFx<NX, L>::run(X, Y);
  const uint32_t ge1 = (Y[L - 1] >> 30) & 1u, shy = 30u + ge1;
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = (Y[i] >> shy) | (Y[i + 1] << (32u - shy));
  r.e = (int32_t)ge1;
  r.neg = 0;
  return r;
}

// The reciprocal the same way: y' = y + y (1 - x y), x in [0.5, 1), y in (1, 2], both Q2.
template <int NX, int L> struct RcpFx
{
  static MW_HD void run(const uint32_t (&X)[NX], uint32_t (&Y)[L])
  {
    if constexpr(L <= 2)
      {
        static_assert(L == 2, "seed width");
        const uint64_t xt = ((uint64_t)X[NX - 1] << 32) | X[NX - 2];
        const double xd = (double)xt * (1.0 / 4611686018427387904.0); // 2^-62
        double yd = (1.0 / xd) * 4611686018427387904.0;
        if(yd > 9223372036854775808.0)
          yd = 9223372036854775808.0; // y <= 2
        const uint64_t y = yd >= 9223372036854775808.0 ? 0x8000000000000000ull : (uint64_t)yd;
        Y[0] = (uint32_t)y;
        Y[1] = (uint32_t)(y >> 32);
      }
    else
      {
        constexpr int H = L / 2 + 1, T = L + 1 + H, LO = L + 3, LD = L - H + 4;
        static_assert(NX >= L + 1 && T >= LO, "x needs one guard limb");
        uint32_t Yh[H];
        RcpFx<NX, H>::run(X, Yh);
        uint32_t Xs[L + 1];
#pragma unroll
        for(int i = 0; i <= L; ++i)
          Xs[i] = X[NX - (L + 1) + i];
        uint32_t D[LO]; // x y with 1.0 = 2^(32 LO - 4)
        fx_mul_from<L + 1, H, T - LO>(Xs, Yh, D);
        uint64_t bw = 0;
#pragma unroll
        for(int i = 0; i < LO; ++i)
          {
            const uint64_t one = (i == LO - 1) ? 0x10000000u : 0u;
            const uint64_t d = one - (uint64_t)D[i] - bw;
            D[i] = (uint32_t)d;
            bw = (d >> 63) & 1u;
          }
        const uint32_t negative = D[LO - 1] >> 31, mask = 0u - negative;
        uint64_t cy = negative;
        uint32_t Dm[LD];
#pragma unroll
        for(int i = 0; i < LD; ++i)
          {
            const uint64_t t = (uint64_t)(D[i] ^ mask) + cy;
            Dm[i] = (uint32_t)t;
            cy = t >> 32;
          }
        uint32_t Pm[LD - 2]; // y |D| at Y's scale: product limbs [H+2, H+LD) shifted left by 4 bits
        fx_mul_from<H, LD, H + 2>(Yh, Dm, Pm);
        uint64_t c = 0;
#pragma unroll
        for(int i = 0; i < L; ++i)
          {
            const uint32_t base = i >= L - H ? Yh[i >= L - H ? i - (L - H) : 0] : 0u;
            uint32_t corr = 0;
            if(i <= L - H)
              corr = (Pm[i + 1 < LD - 2 ? i + 1 : LD - 3] << 4) | (Pm[i] >> 28);
            if(negative)
              {
                const uint64_t t = (uint64_t)base - corr - c;
                Y[i] = (uint32_t)t;
                c = (t >> 63) & 1u;
              }
            else
              {
                const uint64_t t = (uint64_t)base + corr + c;
                Y[i] = (uint32_t)t;
                c = t >> 32;
              }
          }
      }
  }
};
// 1/man for man in [0.5,1) (e = 0, positive)
template <int NL> MW_HD Mw<NL> rcp_mant_fx(const Mw<NL> &man)
{
  constexpr int NX = NL + 2, L = NL + 1;
  uint32_t X[NX]; // x 2^(32 NX - 2) = M 2^62
#pragma unroll
  for(int i = 0; i < NX; ++i)
    {
      const uint32_t lo = (i >= 2) ? man.m[i >= 2 ? i - 2 : 0] : 0u;
      const uint32_t up = (i >= 1 && i - 1 < NL) ? man.m[(i >= 1 && i - 1 < NL) ? i - 1 : 0] : 0u;
      X[i] = (lo >> 2) | (up << 30);
    }
  uint32_t Y[L];
  RcpFx<NX, L>::run(X, Y);
  const uint32_t two = Y[L - 1] >> 31; // y == 2 (x == 1/2)
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = two ? Y[i + 1] : ((Y[i] >> 31) | (Y[i + 1] << 1));
  r.e = 1 + (int32_t)two;
  r.neg = 0;
  return r;
}

// 1/sqrt(a), a > 0
template <int NL> MW_HD Mw<NL> rsqrt(const Mw<NL> &a)
{
  Mw<NL> man = a;
  int32_t e = a.e;
  const int odd = e & 1; // make the exponent even: man in [0.5,2)
  man.e = odd;
  man.neg = 0;
  e -= odd;
  Mw<NL> r;
  if constexpr(NL >= 4)
    r = rsqrt_mant_fx<NL>(man);
  else
    r = RsqrtMant<NL>::run(man);
  r.e -= e / 2;
  return r;
}
// sqrt(a), a >= 0 : s = a*r, one correction s += r*(a - s^2)/2
template <int NL> MW_HD Mw<NL> sqrt(const Mw<NL> &a)
{
  if(a.e == EZERO)
    return a;
  const Mw<NL> r = rsqrt(a);
  Mw<NL> s = mul(a, r);
  const Mw<NL> rem = sub(a, mul(s, s));
  return add(s, mul_2exp(mul(r, rem), -1));
}

// ---- global-memory layout: limb-major structure of arrays -------------------
// plane 0 = header (sign<<31 | e+EBIAS, 0 for zero); plane 1+i = limb i.
// A wavefront touching 64 consecutive elements issues one coalesced 256-B
// transaction per plane.
struct Ptr
{
  uint32_t *base;
  size_t stride; // elements per plane
};
struct CPtr
{
  const uint32_t *base;
  size_t stride;
  CPtr() = default;
  MW_HD CPtr(const uint32_t *b, size_t s) : base(b), stride(s) {}
  MW_HD CPtr(const Ptr &p) : base(p.base), stride(p.stride) {}
};
MW_HD Ptr offset(Ptr p, size_t off) { return Ptr{p.base + off, p.stride}; }
MW_HD CPtr offset(CPtr p, size_t off) { return CPtr(p.base + off, p.stride); }

template <int NL> MW_HD Mw<NL> load(CPtr p, size_t i)
{
  Mw<NL> r;
  const uint32_t h = p.base[i];
#pragma unroll
  for(int k = 0; k < NL; ++k)
    r.m[k] = p.base[(size_t)(k + 1) * p.stride + i];
  r.neg = h >> 31;
  r.e = h ? (int32_t)((h & 0x7fffffffu) - EBIAS) : EZERO;
  return r;
}
template <int NL> MW_HD Mw<NL> load(Ptr p, size_t i) { return load<NL>(CPtr(p), i); }
template <int NL> MW_HD void store(Ptr p, size_t i, const Mw<NL> &a)
{
  const bool z = a.e == EZERO;
  p.base[i] = z ? 0u : ((a.neg << 31) | (uint32_t)(a.e + (int32_t)EBIAS));
#pragma unroll
  for(int k = 0; k < NL; ++k)
    p.base[(size_t)(k + 1) * p.stride + i] = z ? 0u : a.m[k];
}
MW_HD bool is_zero_at(CPtr p, size_t i) { return p.base[i] == 0; }

// limbs for a requested --precision: 2*floor((p+127)/64), i.e. the same 64-bit
// limb count l(p) GMP allocates for mpf, expressed in 32-bit limbs.
inline int limbs_for_precision(int precision_bits)
{
  const int p = precision_bits < 53 ? 53 : precision_bits;
  return 2 * ((p + 127) / 64);
}
} // namespace mw
