// tiledot.hpp — exact fixed-point dot products over one tile of terms, radix 2^27, carry-free column sums.
//
// The triangular solves and trailing updates of the iteration are dot products of multi-word FLOATS:
// every term pays its product (188 v_mad_u64_u32 + v_addc_co_u32 pairs at 18 limbs) and then an aligned
// add into the mw::Acc window (a barrel shift of the product, a two's-complement add: ~250 more
// instructions, profiles/r04_trsm_isa_analysis.txt).  A tile of KT consecutive terms can be summed without
// either cost (round 6):
//   * both operands of the tile are rewritten once as FIXED-POINT numbers relative to one exponent per
//     (row, tile): x_k = x'_k 2^(F-(TB-1)), l_k = l'_k 2^(E-(TB-1)) with integers |x'|, |l'| < C = 2^(TB-1)
//     (TB = 27 W bits, W limbs of 27 bits: the NL-limb mantissa, the bias bit and >= 32 spare bits for the
//     spread of the magnitudes inside the tile: profiles/r05_trsm_tile_exponents.txt measured <= 21 bits of
//     spread that matters on production-like blocks);
//   * the images are BIASED, b = C + x' in [0, 2^TB), so every limb product is non-negative and
//       sum_k x'_k l'_k = G - C (Sx + Sl) + KT C^2,   G = sum_k bx_k bl_k,  Sx = sum_k bx_k,  Sl = sum_k bl_k
//     (the trick of the Q syrk's image, kernels.hpp);
//   * limbs have 27 bits: a limb product is < 2^54, so a COLUMN of the product, summed over the column's
//     <= W limb pairs and over all KT terms of the tile, stays below 2^64 (W KT (2^27-1)^2 < 2^64): one
//     v_mad_u64_u32 per limb pair and NOTHING else per term -- no carry instruction, no alignment, no
//     normalisation.  Carries are resolved once per tile, when the 2W-1-cut column sums are folded.
// The tile's sum re-enters the float world as ONE term of the surrounding mw::Acc (acc_add_raw), so a
// dot product of n terms pays n/KT aligned adds instead of n.
// Accuracy: an image keeps a number to 2^-(TB-1) of the LARGEST entry of its tile row; the float product
// keeps it to 2^-(32 NL) of itself.  With s spare bits (TB - 1 - 32 NL >= 32) a tile sum is at least as
// accurate as the float sum whenever max_k|x_k| max_k|l_k| <= 2^s max_k|x_k l_k|; beyond that it degrades
// by the excess, bit for bit (never wrong, only shorter).  Columns below `cut` are not formed: what they
// would add is < 2^-(27 W + 15) of C^2, below the last bit a (NL+1)-limb term keeps.
//
// Everything here is plain integer C++ (the CPU emulation build of the tests runs the same code).
#pragma once
#include "mw.hpp"

namespace sdpb
{
namespace td
{
using mw::Mw;
constexpr int B = 27; // bits per limb of a tile image
constexpr uint32_t MASK = (1u << B) - 1u;
struct alignas(16) Quad // 16-byte moves of image words
{
  uint32_t w[4];
};
// limbs of an image entry for an NL-limb mantissa (bias bit + >= 32 spare bits)
template <int NL> constexpr int limbs() { return (32 * NL + 1 + 32 + B - 1) / B; }
// a column sum of KT terms fits 64 bits
template <int W, int KT> constexpr bool fits() { return (double)W * KT * 18014398241046529.0 < 18446744073709551616.0; }
template <int W> constexpr int cut() { return W - 2; }                  // lowest product column that is formed
template <int W> constexpr int ncol() { return 2 * W - 1 - cut<W>(); }  // columns cut .. 2W-2
template <int W> constexpr int nres() { return W + 4; }                 // 27-bit limbs of a tile's signed sum (columns cut .. 2W+1)
template <int W> constexpr int nres32() { return (B * nres<W>() + 31) / 32; }
template <int W> constexpr int padded() { return (W + 3) / 4 * 4; }      // words per image entry in memory (16-byte accesses)

// image of x relative to the tile exponent F >= x.e:  b = C + sign(x) floor(|x| 2^(TB-1-F)),  C = 2^(TB-1)
template <int NL, int W> MW_HD void to_image(const Mw<NL> &x, int32_t F, uint32_t (&b)[W])
{
  constexpr int TB = B * W, LA = (TB + 31) / 32 + 1, UP = TB - 1 - 32 * NL, UQ = UP / 32, UR = UP % 32;
  static_assert(UP >= 0 && UQ + NL + 1 <= LA, "the mantissa fits below the bias bit");
  uint32_t a[LA];
#pragma unroll
  for(int i = 0; i < LA; ++i)
    a[i] = 0;
#pragma unroll
  for(int i = 0; i < NL; ++i)
    {
      a[i + UQ] |= UR ? (x.m[i] << UR) : x.m[i];
      if(UR)
        a[i + UQ + 1] |= x.m[i] >> (32 - UR);
    }
  const bool z = x.e == mw::EZERO;
  const uint32_t d = z ? 0u : (uint32_t)(F - x.e);
  const bool gone = z || d >= (uint32_t)(TB - 1);
  const uint32_t dd = gone ? 0u : d;
  mw::shr_limbs<LA>(a, dd >> 5);
  mw::shr_bits<LA>(a, dd & 31u);
  // two's complement of a negative entry (the low TB bits are the digits of 2^TB - v)
  const uint32_t mask = (x.neg && !gone) ? 0xffffffffu : 0u;
  uint32_t carry = mask & 1u;
#pragma unroll
  for(int i = 0; i < LA; ++i)
    {
      const uint64_t s = (uint64_t)((gone ? 0u : a[i]) ^ mask) + carry;
      a[i] = (uint32_t)s;
      carry = (uint32_t)(s >> 32);
    }
#pragma unroll
  for(int i = 0; i < W; ++i)
    {
      const int bit0 = B * i, q = bit0 / 32, r = bit0 % 32;
      b[i] = mw::funnel_r(a[q + 1], a[q], (uint32_t)r) & MASK;
    }
  b[W - 1] ^= 1u << (B - 1); // + C for x >= 0, 2^TB - v - C for x < 0
}

// the column sums of a tile: c[j] = sum over the terms and over i + k = cut + j of bx_i bl_k  (< 2^64)
template <int W> struct Cols
{
  uint64_t c[ncol<W>()];
};
template <int W> MW_HD void cols_zero(Cols<W> &g)
{
#pragma unroll
  for(int j = 0; j < ncol<W>(); ++j)
    g.c[j] = 0;
}
// one term: g += bx * bl (columns cut .. 2W-2); one 32 x 32 + 64 multiply-add per limb pair, nothing else
template <int W, int C = cut<W>()> MW_HD void mac(Cols<W> &g, const uint32_t (&x)[W], const uint32_t (&l)[W])
{
  constexpr int I0 = C - (W - 1) > 0 ? C - (W - 1) : 0, I1 = C < W - 1 ? C : W - 1;
  uint64_t s = g.c[C - cut<W>()];
#pragma unroll
  for(int i = I0; i <= I1; ++i)
    s += (uint64_t)x[i] * (uint64_t)l[C - i];
  g.c[C - cut<W>()] = s;
  if constexpr(C < 2 * W - 2)
    mac<W, C + 1>(g, x, l);
}

// The signed sum of a tile, sum_k x'_k l'_k = G - C (Sx + Sl) + terms C^2, from the column sums and the
// limb-wise sums sx, sl of the two biased images (sx[i] = sum_k bx_k[i] < 2^32: not carried): magnitude as
// nres32 32-bit limbs (value = mag 2^(27 cut)), sign in `negative`.
template <int W> MW_HD void tile_sum(const Cols<W> &g, const uint32_t (&sx)[W], const uint32_t (&sl)[W], uint32_t terms,
                                     uint32_t (&mag)[nres32<W>()], uint32_t &negative)
{
  constexpr int CUT = cut<W>(), NC = ncol<W>(), NR = nres<W>(), LM = nres32<W>();
  // T = Sx + Sl in 27-bit limbs
  uint32_t t[W + 1];
  {
    uint64_t cy = 0;
#pragma unroll
    for(int i = 0; i < W; ++i)
      {
        const uint64_t v = (uint64_t)sx[i] + sl[i] + cy;
        t[i] = (uint32_t)v & MASK;
        cy = v >> B;
      }
    t[W] = (uint32_t)cy;
  }
  // signed digits D_c, c = cut .. 2W+1, carried on the fly:  G limb - 2^26 T_(c-W+1) (+ terms 2^25 at c = 2W-1)
  uint32_t r[NR];
  uint64_t gc = 0; // carry of the column sums
  int64_t sc = 0;  // signed carry of the result
#pragma unroll
  for(int j = 0; j < NR; ++j)
    {
      const int c = CUT + j;
      if(j < NC)
        gc += g.c[j < NC ? j : 0]; // < 2^64: the previous carry is < 2^37
      int64_t v = (int64_t)(gc & MASK) + sc;
      gc >>= B;
      const int ti = c - (W - 1);
      if(ti >= 0 && ti <= W)
        v -= (int64_t)((uint64_t)t[ti >= 0 && ti <= W ? ti : 0] << (B - 1));
      if(c == 2 * W - 1)
        v += (int64_t)((uint64_t)terms << (B - 2));
      r[j] = (uint32_t)((uint64_t)v & MASK);
      sc = v >> B; // arithmetic
    }
  negative = sc < 0 ? 1u : 0u;
  // magnitude: two's complement over the NR digits
  {
    const uint32_t m = negative ? MASK : 0u;
    uint32_t cy = negative;
#pragma unroll
    for(int j = 0; j < NR; ++j)
      {
        const uint32_t v = (r[j] ^ m) + cy;
        r[j] = v & MASK;
        cy = v >> B;
      }
  }
  // 27-bit limbs -> 32-bit limbs
#pragma unroll
  for(int k = 0; k < LM; ++k)
    {
      const int bit0 = 32 * k, i0 = bit0 / B, s0 = bit0 - B * i0; // compile-time after unrolling
      uint64_t v = (uint64_t)r[i0 < NR ? i0 : 0] >> s0;
      if(i0 + 1 < NR)
        v |= (uint64_t)r[i0 + 1 < NR ? i0 + 1 : 0] << (B - s0);
      if(i0 + 2 < NR && 2 * B - s0 < 32)
        v |= (uint64_t)r[i0 + 2 < NR ? i0 + 2 : 0] << (2 * B - s0);
      mag[k] = i0 < NR ? (uint32_t)v : 0u;
    }
}

// acc += (-1)^negate (tile sum) 2^(F + E - 2 TB + 2): the tile's sum as ONE term of a float accumulator
template <int NL, int W>
MW_HD void acc_add_tile(mw::Acc<NL> &acc, const Cols<W> &g, const uint32_t (&sx)[W], const uint32_t (&sl)[W], uint32_t terms, int32_t F, int32_t E,
                        uint32_t negate)
{
  constexpr int LM = nres32<W>(), TB = B * W;
  static_assert(LM >= NL + 1, "a term of the accumulator is cut from the tile sum");
  if(F == mw::EZERO || E == mw::EZERO)
    return; // a row of zeros
  uint32_t mag[LM], negative;
  tile_sum<W>(g, sx, sl, terms, mag, negative);
  const int top = mw::top_nonzero<LM>(mag);
  if(top < 0)
    return;
  mw::shl_limbs<LM>(mag, (uint32_t)(LM - 1 - top));
  uint32_t P[NL + 1];
#pragma unroll
  for(int i = 0; i <= NL; ++i)
    P[i] = mag[LM - 1 - NL + i];
  mw::acc_add_raw<NL>(acc, P, 32 * (top + 1) + B * cut<W>() + F + E - 2 * TB + 2, negative ^ (negate & 1u));
}
} // namespace td
} // namespace sdpb
