// TEST INFRASTRUCTURE: sdpb_amd/csrc/tiledot.hpp against exact GMP arithmetic on the host (tests/test_host_logic.py compiles
// and runs it).  For random tiles of multi-word floats with a controlled spread of magnitudes it checks that
//   (1) to_image is the exact floor of |x| 2^(TB-1-F) with the right sign, biased by C;
//   (2) the column sums + tile_sum reproduce sum_k x'_k l'_k up to the columns that are not formed;
//   (3) acc_add_tile leaves the float accumulator within 2^-(32 NL - 2) of the exact sum of the tile's products, relative
//       to the largest term, whenever the spread stays inside the spare bits.
#include <gmp.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "tiledot.hpp"

using namespace sdpb;
template <int NL> static void to_mpz(mpz_t z, const mw::Mw<NL> &x) // mantissa as an integer
{
  mpz_import(z, NL, -1, 4, 0, 0, x.m);
}
template <int NL> static void value(mpf_t f, const mw::Mw<NL> &x)
{
  if(x.e == mw::EZERO)
    {
      mpf_set_ui(f, 0);
      return;
    }
  mpz_t z;
  mpz_init(z);
  to_mpz<NL>(z, x);
  mpf_set_z(f, z);
  const long sh = (long)x.e - 32L * NL;
  if(sh >= 0)
    mpf_mul_2exp(f, f, (unsigned long)sh);
  else
    mpf_div_2exp(f, f, (unsigned long)(-sh));
  if(x.neg)
    mpf_neg(f, f);
  mpz_clear(z);
}

template <int NL, int KT> static int run(unsigned seed, int spread_bits, int trials)
{
  constexpr int W = td::limbs<NL>(), TB = td::B * W;
  static_assert(td::fits<W, KT>(), "column sums fit 64 bits");
  std::mt19937_64 rng(seed);
  mpf_set_default_prec(64 * NL + 512);
  int bad = 0;
  double worst = -1e9;
  for(int tr = 0; tr < trials; ++tr)
    {
      std::vector<mw::Mw<NL>> x(KT), l(KT);
      int32_t F = mw::EZERO, E = mw::EZERO;
      const int terms = tr % 7 == 3 ? 1 + (int)(rng() % KT) : KT;
      for(int k = 0; k < KT; ++k)
        for(int which = 0; which < 2; ++which)
          {
            mw::Mw<NL> v;
            for(int i = 0; i < NL; ++i)
              v.m[i] = (uint32_t)rng();
            if(tr % 5 == 1)
              for(int i = 0; i < NL; ++i)
                v.m[i] = 0xffffffffu; // all ones: the largest mantissa
            v.m[NL - 1] |= 0x80000000u;
            v.e = (int32_t)(rng() % 2000) - 1000 - (int32_t)(rng() % (spread_bits + 1)) + (which ? 17 * k % 11 : -3 * k % 13);
            v.neg = (uint32_t)(rng() & 1);
            if(rng() % 19 == 0 || k >= terms)
              v = mw::zero<NL>();
            (which ? l : x)[k] = v;
          }
      for(int k = 0; k < KT; ++k)
        {
          F = x[k].e > F ? x[k].e : F;
          E = l[k].e > E ? l[k].e : E;
        }
      td::Cols<W> g;
      td::cols_zero<W>(g);
      uint32_t sx[W] = {0}, sl[W] = {0};
      mpz_t exact, bxz, blz, t, Cz;
      mpz_inits(exact, bxz, blz, t, Cz, NULL);
      mpz_ui_pow_ui(Cz, 2, TB - 1);
      for(int k = 0; k < KT; ++k)
        {
          uint32_t bx[W], bl[W];
          td::to_image<NL, W>(x[k], F == mw::EZERO ? 0 : F, bx);
          td::to_image<NL, W>(l[k], E == mw::EZERO ? 0 : E, bl);
          for(int i = 0; i < W; ++i)
            {
              if(bx[i] > td::MASK || bl[i] > td::MASK)
                ++bad;
              sx[i] += bx[i];
              sl[i] += bl[i];
            }
          td::mac<W>(g, bx, bl);
          // (1) the image: b - C == sign floor(|x| 2^(TB-1-F))
          for(int which = 0; which < 2; ++which)
            {
              const mw::Mw<NL> &v = which ? l[k] : x[k];
              const uint32_t *b = which ? bl : bx;
              mpz_set_ui(t, 0);
              for(int i = W - 1; i >= 0; --i)
                {
                  mpz_mul_2exp(t, t, td::B);
                  mpz_add_ui(t, t, b[i]);
                }
              mpz_sub(t, t, Cz);
              mpz_t want;
              mpz_init(want);
              if(v.e != mw::EZERO)
                {
                  to_mpz<NL>(want, v);
                  const long sh = (long)(TB - 1) - 32L * NL - ((which ? E : F) - v.e);
                  if(sh >= 0)
                    mpz_mul_2exp(want, want, (unsigned long)sh);
                  else
                    mpz_tdiv_q_2exp(want, want, (unsigned long)(-sh));
                  if(v.neg)
                    mpz_neg(want, want);
                }
              if(mpz_cmp(t, want) != 0)
                {
                  if(bad < 5)
                    gmp_printf("image mismatch NL=%d trial %d k %d: got %Zd want %Zd\n", NL, tr, k, t, want);
                  ++bad;
                }
              if(which)
                mpz_set(blz, t);
              else
                mpz_set(bxz, t);
              mpz_clear(want);
            }
          mpz_addmul(exact, bxz, blz); // sum_k x'_k l'_k
        }
      // (2) the tile sum against the exact integer, in units of 2^(27 cut)
      uint32_t mag[td::nres32<W>()], negative;
      td::tile_sum<W>(g, sx, sl, (uint32_t)KT, mag, negative);
      mpz_import(t, td::nres32<W>(), -1, 4, 0, 0, mag);
      if(negative)
        mpz_neg(t, t);
      mpz_mul_2exp(t, t, td::B * td::cut<W>());
      mpz_sub(t, exact, t); // what the unformed columns would have added: 0 <= t < 2^(27 cut + 37)
      if(mpz_sgn(t) < 0 || mpz_sizeinbase(t, 2) > (size_t)(td::B * td::cut<W>() + 37))
        {
          if(bad < 5)
            gmp_printf("tile sum off NL=%d trial %d: diff %Zd (bits %zu)\n", NL, tr, t, mpz_sizeinbase(t, 2));
          ++bad;
        }
      // (3) through the float accumulator
      mw::Acc<NL> acc = mw::acc_zero<NL>();
      mw::Mw<NL> seed_term = mw::from_u32<NL>(3);
      seed_term.e += (F == mw::EZERO || E == mw::EZERO) ? 0 : F + E - 40;
      mw::acc_add(acc, seed_term);
      td::acc_add_tile<NL, W>(acc, g, sx, sl, (uint32_t)KT, F, E, 1u);
      const mw::Mw<NL> got = mw::acc_result(acc);
      mpf_t want, prod, a, b, gotf, big, err;
      mpf_inits(want, prod, a, b, gotf, big, err, NULL);
      value<NL>(want, seed_term);
      mpf_abs(big, want);
      int32_t spread_seen = 0;
      int32_t emax = mw::EZERO;
      for(int k = 0; k < KT; ++k)
        {
          value<NL>(a, x[k]);
          value<NL>(b, l[k]);
          mpf_mul(prod, a, b);
          mpf_sub(want, want, prod);
          mpf_abs(prod, prod);
          if(mpf_cmp(prod, big) > 0)
            mpf_set(big, prod);
          if(x[k].e != mw::EZERO && l[k].e != mw::EZERO && x[k].e + l[k].e > emax)
            emax = x[k].e + l[k].e;
        }
      if(emax != mw::EZERO)
        spread_seen = F + E - emax;
      value<NL>(gotf, got);
      mpf_sub(err, gotf, want);
      mpf_abs(err, err);
      if(mpf_sgn(err) != 0)
        {
          mpf_div(err, err, big);
          long ex;
          const double d = mpf_get_d_2exp(&ex, err);
          (void)d;
          const double lg = (double)ex;
          const int spare = TB - 1 - 32 * NL;
          const double bar = -(32.0 * NL - 4) + (spread_seen > spare ? spread_seen - spare : 0);
          if(lg > worst)
            worst = lg;
          if(lg > bar)
            {
              if(bad < 5)
                printf("accumulator off NL=%d trial %d: 2^%.0f relative to the largest term (bar 2^%.0f, spread %d)\n", NL, tr, lg, bar, spread_seen);
              ++bad;
            }
        }
      mpf_clears(want, prod, a, b, gotf, big, err, NULL);
      mpz_clears(exact, bxz, blz, t, Cz, NULL);
    }
  printf("NL=%d W=%d KT=%d spread<=%d: %d trials, worst error 2^%.0f of the largest term, %d failures\n", NL, W, KT, spread_bits, trials, worst, bad);
  return bad;
}

int main()
{
  int bad = 0;
  bad += run<18, 32>(1, 0, 300);
  bad += run<18, 32>(2, 20, 300);
  bad += run<18, 32>(3, 80, 300);   // beyond the spare bits: degrades by the excess only
  bad += run<16, 32>(4, 30, 200);
  bad += run<6, 32>(5, 10, 200);
  bad += run<10, 32>(6, 25, 200);
  bad += run<24, 32>(7, 25, 100);
  bad += run<26, 16>(8, 25, 100);   // 16-term tiles where 32 terms would overflow a column
  return bad ? 1 : 0;
}
