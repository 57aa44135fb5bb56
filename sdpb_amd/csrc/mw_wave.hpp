// mw_wave.hpp — one wavefront computes ONE reciprocal square root (reciprocal, square root).
//
// The pivot of a Cholesky column (k_chol_inv_lds) is 1/sqrt(d_k) of a single number: a dependent chain that one lane
// walks in 4.8 us at 576 bits (14.6 us at 1088; profiles/r04i_ubench_chain_latency.txt) while the other 63 lanes of its
// wavefront idle — and the chain of N such pivots is the part of Cholesky(Q) that no number of GPUs shortens.  Here the
// 64 lanes share the work: a fixed-point number lives one limb per lane, a product is formed column by column (lane j
// sums column j: v_readlane broadcasts a limb of one operand, ds_bpermute fetches the matching limb of the other), and
// carries are resolved for all lanes at once from two wavefront ballots (generate / propagate masks added as 64-bit
// integers).  The Newton ladder is the one of mw::RsqrtFx, level for level and bit for bit — same truncated products,
// same two's-complement steps — so the result is IDENTICAL to mw::rsqrt (the CPU emulation build and the oracle
// comparison see no difference); only the top levels, where the products are long, run cooperatively, the short ones
// are computed redundantly by every lane.  Replaces nothing in the reference: El::Cholesky's pivots are mpf_sqrt calls.
#pragma once
#include "mw.hpp"

namespace mw
{
namespace wv
{
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
// limb of lane `src` (uniform) for everybody
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
// limb of lane `src` (per lane); out of range -> 0
__device__ __forceinline__ uint32_t gather(uint32_t v, int src)
{
  const uint32_t r = (uint32_t)__shfl((int)v, src & 63);
  return (src >= 0 && src < 64) ? r : 0u;
}
// the number shifted by `d` limbs towards the top (lane j <- lane j - d) / the bottom (lane j <- lane j + d), zero fill
__device__ __forceinline__ uint32_t limbs_up(uint32_t v, int d) { return gather(v, lane_id() - d); }
__device__ __forceinline__ uint32_t limbs_down(uint32_t v, int d) { return gather(v, lane_id() + d); }

// x + y + cin (cin = 0 / 1, uniform), limb j in lane j, every carry resolved: lane j generates (x_j + y_j >= 2^32) or
// propagates (x_j + y_j == 2^32 - 1); with A = G | P and B = G the carries INTO the lanes are the carries of the 64-bit
// addition A + B + cin, (A + B + cin) ^ A ^ B.
__device__ __forceinline__ uint32_t add(uint32_t x, uint32_t y, uint32_t cin)
{
  const uint64_t s = (uint64_t)x + y;
  const uint32_t v = (uint32_t)s;
  const uint64_t G = __ballot((uint32_t)(s >> 32) != 0u), P = __ballot(v == 0xffffffffu);
  const uint64_t A = G | P, S = A + G + cin, C = S ^ A ^ G;
  return v + (uint32_t)((C >> lane_id()) & 1u);
}
// x - y - bin
__device__ __forceinline__ uint32_t sub(uint32_t x, uint32_t y, uint32_t bin) { return add(x, ~y, 1u - bin); }

// every lane takes the limb of its lower neighbour, lane 0 that of lane 63 (DPP wave_ror:1: one VALU move, where
// ds_bpermute is a round trip through the LDS crossbar)
__device__ __forceinline__ uint32_t rotate_up1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xf, 0xf, false); }
// the number one limb up / down, zero fill (DPP wave_shr:1 / wave_shl:1)
__device__ __forceinline__ uint32_t up1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t down1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
// acc(lo:64, hi:32) += a * b with a in a scalar register (the broadcast limb)
#define MW_MAC_S(lo, hi, a, b) \
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "s"(a), "v"(b) : "vcc")

// limbs [F, LA + LB) of a * b with the columns below F - 2 dropped — mw::fx_mul_from<LA, LB, F> — limb F + j in lane j.
// a: LA limbs, b: LB limbs, one per lane from lane 0; the lanes above a number hold zeros (every producer here
// guarantees it).  Lane j sums column K0 + j: it needs b[K0 + j - i] beside the broadcast a[i], i = 0 .. LA - 1 — the
// limbs of b rotate past the lanes, one lane per step; what rotates in from below b[0] is zero because the ring is long
// enough (checked below), what lies above ring position 63 is masked where the columns reach that far.
template <int LA, int LB, int F> __device__ __forceinline__ uint32_t mul_from(uint32_t a, uint32_t b)
{
  constexpr int K0 = F - 2 > 0 ? F - 2 : 0, NC = LA + LB - 1 - K0; // columns K0 .. LA + LB - 2, then the top limb
  static_assert(NC + 1 <= 64 && LA <= 64 && LB <= 64, "one position per lane");
  // indices K0 + j - i of the lanes j < NC run from K0 - (LA - 1) to K0 + NC - 1 = LA + LB - 2: the negative ones
  // wrap to ring positions >= 64 - (LA - 1 - K0), which must lie above b
  static_assert(K0 >= LA - 1 || 64 - (LA - 1 - K0) >= LB, "the ring wraps into the operand");
  const int j = lane_id();
  uint32_t bw = (uint32_t)__shfl((int)b, (K0 + j) & 63); // ring position K0 + j
  uint64_t lo = 0;
  uint32_t hi = 0;
#pragma unroll
  for(int i = 0; i < LA; ++i)
    {
      const uint32_t ai = bcast(a, i);
      if constexpr(LA + LB - 2 >= 64)
        {
          // the top columns reach indices past the ring (they lie beyond b: zero), where it has wrapped to b[0 ..]
          const uint32_t bv = K0 + j - i < 64 ? bw : 0u;
          MW_MAC_S(lo, hi, ai, bv);
        }
      else
        MW_MAC_S(lo, hi, ai, bw);
      if(i + 1 < LA)
        bw = rotate_up1(bw); // lane j now holds ring position K0 + j - (i + 1)
    }
  if(j >= NC) // lanes above the last column (the top limb starts from the carries alone)
    lo = 0, hi = 0;
  // sum_j column_j 2^(32 j) = W0 + W1 2^32 + W2 2^64 with the three words of every column: first the lane-wise sum of
  // the three contributions (< 2^34), then ONE carry resolution
  const uint32_t w1 = up1((uint32_t)(lo >> 32)), w2 = up1(up1(hi));
  const uint64_t t = (uint64_t)(uint32_t)lo + w1 + w2;
  uint32_t r = add((uint32_t)t, up1((uint32_t)(t >> 32)), 0u);
  if constexpr(F - K0 >= 1)
    r = down1(r);
  if constexpr(F - K0 >= 2)
    r = down1(r);
  return r;
}

// mw::RsqrtFx<NX, L> with the number one limb per lane: x (NX limbs, Q2) -> y ~ 1/sqrt(x) (L limbs, Q2).
// Levels of at most LC limbs: every lane runs mw::RsqrtFx itself on the top limbs of x (short products: cheaper
// than talking), then keeps its own limb.
constexpr int LC = 8;
template <int NX, int L> struct RsqrtFxW
{
  static __device__ __forceinline__ uint32_t run(uint32_t x)
  {
    if constexpr(L <= LC)
      {
        // RsqrtFx<.., L> reads the top L + 1 limbs of x only (and passes x down): a top-aligned copy gives the same bits
        constexpr int NT = L + 1;
        uint32_t X[NT], Y[L];
#pragma unroll
        for(int i = 0; i < NT; ++i)
          X[i] = bcast(x, NX - NT + i);
        RsqrtFx<NT, L>::run(X, Y);
        uint32_t y = 0;
#pragma unroll
        for(int i = 0; i < L; ++i)
          y = lane_id() == i ? Y[i] : y;
        return y;
      }
    else
      {
        constexpr int H = L / 2 + 1, T = L + 1 + 2 * H, LO = L + 3, LD = L - H + 4;
        static_assert(NX >= L + 1, "x needs one guard limb");
        const int j = lane_id();
        const uint32_t yh = RsqrtFxW<NX, H>::run(x);                 // H limbs
        const uint32_t S = mul_from<H, H, 0>(yh, yh);                  // y^2 exactly, 2H limbs
        const uint32_t xs = limbs_down(x, NX - (L + 1));               // top L + 1 limbs of x
        uint32_t D = mul_from<L + 1, 2 * H, T - LO>(xs, S);            // x y^2, LO limbs, 1.0 = 2^(32 LO - 6)
        D = j < LO ? D : 0u;
        D = sub(j == LO - 1 ? 0x04000000u : 0u, D, 0u);               // 1 - x y^2 (two's complement)
        D = j < LO ? D : 0u;
        const uint32_t negative = bcast(D, LO - 1) >> 31, mask = 0u - negative;
        uint32_t Dm = add(j < LD ? D ^ mask : 0u, 0u, negative);       // |1 - x y^2|, LD limbs
        Dm = j < LD ? Dm : 0u;
        const uint32_t Pm = mul_from<H, LD, H + 2>(yh, Dm);            // y |D| / 2 at Y's scale, LD - 2 limbs
        // (every shuffle is executed by all 64 lanes and selected afterwards: ds_bpermute reads 0 from a lane that sits
        // out a divergent branch)
        const uint32_t base = limbs_up(yh, L - H), pm1 = down1(Pm);
        const uint32_t corr = j <= L - H ? ((pm1 << 5) | (Pm >> 27)) : 0u;
        const uint32_t y = add(base, corr ^ mask, negative); // base - corr or base + corr
        return j < L ? y : 0u;
      }
  }
};

// mw::rsqrt(a) of the value lane `owner` (uniform) holds, computed by the whole wavefront and returned to every lane.
// Bit-identical to mw::rsqrt<NL>.  a > 0.
template <int NL> __device__ __forceinline__ Mw<NL> rsqrt_lanes(const Mw<NL> &a, int owner)
{
  static_assert(NL >= 4 && NL + 6 <= 64, "fixed-point path of mw::rsqrt; one limb per lane");
  constexpr int NX = NL + 2, L = NL + 1;
  const int j = lane_id();
  int32_t e = (int32_t)bcast((uint32_t)a.e, owner);
  const int odd = e & 1; // mantissa in [0.5, 2): as mw::rsqrt
  e -= odd;
  // the mantissa one limb per lane, then x 2^(32 NX - 2) = M 2^(62 + odd): as mw::rsqrt_mant_fx
  uint32_t m = 0;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    {
      const uint32_t li = bcast(a.m[i], owner);
      m = j == i ? li : m;
    }
  const uint32_t sh = 2u - (uint32_t)odd;
  const uint32_t up = up1(m), lo = up1(up); // zero below lane 1 / 2 and above the mantissa
  const uint32_t x = j < NX ? ((lo >> sh) | (up << (32u - sh))) : 0u;
  const uint32_t y = RsqrtFxW<NX, L>::run(x);
  const uint32_t ge1 = (bcast(y, L - 1) >> 30) & 1u, shy = 30u + ge1;
  const uint32_t y1 = down1(y);
  const uint32_t rl = (y >> shy) | (y1 << (32u - shy)); // limb j of the result, j < NL
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = bcast(rl, i);
  r.e = (int32_t)ge1 - e / 2;
  r.neg = 0;
  return r;
}

// mw::RcpFx<NX, L> the same way: y' = y + y (1 - x y), x in [0.5, 1), y in (1, 2], both Q2.
template <int NX, int L> struct RcpFxW
{
  static __device__ __forceinline__ uint32_t run(uint32_t x)
  {
    if constexpr(L <= LC)
      {
        constexpr int NT = L + 1;
        uint32_t X[NT], Y[L];
#pragma unroll
        for(int i = 0; i < NT; ++i)
          X[i] = bcast(x, NX - NT + i);
        RcpFx<NT, L>::run(X, Y);
        uint32_t y = 0;
#pragma unroll
        for(int i = 0; i < L; ++i)
          y = lane_id() == i ? Y[i] : y;
        return y;
      }
    else
      {
        constexpr int H = L / 2 + 1, T = L + 1 + H, LO = L + 3, LD = L - H + 4;
        static_assert(NX >= L + 1 && T >= LO, "x needs one guard limb");
        const int j = lane_id();
        const uint32_t yh = RcpFxW<NX, H>::run(x);
        const uint32_t xs = limbs_down(x, NX - (L + 1));
        uint32_t D = mul_from<L + 1, H, T - LO>(xs, yh); // x y, LO limbs, 1.0 = 2^(32 LO - 4)
        D = j < LO ? D : 0u;
        D = sub(j == LO - 1 ? 0x10000000u : 0u, D, 0u);
        D = j < LO ? D : 0u;
        const uint32_t negative = bcast(D, LO - 1) >> 31, mask = 0u - negative;
        uint32_t Dm = add(j < LD ? D ^ mask : 0u, 0u, negative);
        Dm = j < LD ? Dm : 0u;
        const uint32_t Pm = mul_from<H, LD, H + 2>(yh, Dm); // y |D| at Y's scale, LD - 2 limbs
        const uint32_t base = limbs_up(yh, L - H), pm1 = down1(Pm);
        const uint32_t corr = j <= L - H ? ((pm1 << 4) | (Pm >> 28)) : 0u;
        const uint32_t y = add(base, corr ^ mask, negative);
        return j < L ? y : 0u;
      }
  }
};
// the value of lane `owner` for everybody
template <int NL> __device__ __forceinline__ Mw<NL> bcast(const Mw<NL> &a, int owner)
{
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = bcast(a.m[i], owner);
  r.e = (int32_t)bcast((uint32_t)a.e, owner);
  r.neg = bcast(a.neg, owner);
  return r;
}
// mw::rcp(a) of the value lane `owner` holds (a != 0), by the whole wavefront, returned to every lane; bit-identical
template <int NL> __device__ __forceinline__ Mw<NL> rcp_lanes(const Mw<NL> &a, int owner)
{
  static_assert(NL >= 4 && NL + 6 <= 64, "fixed-point path of mw::rcp; one limb per lane");
  constexpr int NX = NL + 2, L = NL + 1;
  const int j = lane_id();
  uint32_t m = 0;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    {
      const uint32_t li = bcast(a.m[i], owner);
      m = j == i ? li : m;
    }
  const uint32_t up = up1(m), lo = up1(up);
  const uint32_t x = j < NX ? ((lo >> 2) | (up << 30)) : 0u; // x 2^(32 NX - 2) = M 2^62: as mw::rcp_mant_fx
  const uint32_t y = RcpFxW<NX, L>::run(x);
  const uint32_t two = bcast(y, L - 1) >> 31; // y == 2 (x == 1/2)
  const uint32_t y1 = down1(y);
  const uint32_t rl = two ? y1 : ((y >> 31) | (y1 << 1));
  Mw<NL> r;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    r.m[i] = bcast(rl, i);
  r.e = 1 + (int32_t)two - (int32_t)bcast((uint32_t)a.e, owner);
  r.neg = bcast(a.neg, owner);
  return r;
}
// Above 58 limbs (--precision 2048: 66) a number no longer fits the wavefront one limb per lane: every lane runs the
// one-lane ladder on its own copy of the owner's value instead (same bits, the latency of round 3's chains).
template <int NL> __device__ __forceinline__ Mw<NL> rsqrt(const Mw<NL> &a, int owner)
{
  if constexpr(NL + 6 <= 64)
    return rsqrt_lanes<NL>(a, owner);
  else
    return mw::rsqrt<NL>(bcast(a, owner));
}
template <int NL> __device__ __forceinline__ Mw<NL> rcp(const Mw<NL> &a, int owner)
{
  if constexpr(NL + 6 <= 64)
    return rcp_lanes<NL>(a, owner);
  else
    return mw::rcp<NL>(bcast(a, owner));
}
// mw::sqrt(a) of the value lane `owner` holds (a >= 0): the reciprocal square root together, the correction step
// s = a r, s += r (a - s^2) / 2 by every lane on its own copy (four short dependent operations); bit-identical
template <int NL> __device__ __forceinline__ Mw<NL> sqrt(const Mw<NL> &a, int owner)
{
  const Mw<NL> av = bcast(a, owner);
  if(av.e == EZERO)
    return av;
  const Mw<NL> r = rsqrt<NL>(av, owner);
  const Mw<NL> s = mul(av, r);
  const Mw<NL> rem = mw::sub(av, mul(s, s));
  return mw::add(s, mul_2exp(mul(r, rem), -1));
}
#else
// host pass of hipcc / CPU emulation build: kernels keep to the one-lane path (same bits); this only has to parse
template <int NL> MW_HD Mw<NL> rsqrt(const Mw<NL> &a, int) { return mw::rsqrt<NL>(a); }
template <int NL> MW_HD Mw<NL> rcp(const Mw<NL> &a, int) { return mw::rcp<NL>(a); }
template <int NL> MW_HD Mw<NL> sqrt(const Mw<NL> &a, int) { return mw::sqrt<NL>(a); }
#endif
} // namespace wv
} // namespace mw
