// nl_probe.hip -- cross-width consistency of the kernels behind the step lengths and the block condition numbers.
// Round 5 built a 98-limb width (3072 bits) once: arithmetic, exact syrk and image floor passed on the device, whole
// iterations gave P-step, D-step and max_block_cond_number to 2^-64 of the oracle only.  This probe runs the kernels those
// three scalars pass through -- k_tridiag, k_tridiag_min, k_reduce<RED_MIN/RED_MAX>, k_diag_ratio, the LDS tree of
// k_cond_best -- at two widths on the SAME matrices (entries with 64 significant bits: exact at every width) and prints
// how far the wide result is from the narrow one: a healthy pair agrees to the narrow width's last bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPROBE_A=66 -DPROBE_B=98 -DSDPB_PB=16 profiles/tools/nl_probe.hip -o nl_probe
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../sdpb_amd/csrc/kernels.hpp"
using namespace sdpb;
#ifndef PROBE_A
#define PROBE_A 18
#endif
#ifndef PROBE_B
#define PROBE_B 34
#endif
static uint64_t lcg_state = 88172645463325252ull;
static uint64_t lcg()
{
  lcg_state ^= lcg_state << 13;
  lcg_state ^= lcg_state >> 7;
  lcg_state ^= lcg_state << 17;
  return lcg_state;
}
template <int NL> Mw<NL> from_bits(uint64_t mant, int e, bool neg)
{
  Mw<NL> v = mw::zero<NL>();
  mant |= 1ull << 63;
  v.m[NL - 1] = (uint32_t)(mant >> 32);
  v.m[NL - 2] = (uint32_t)mant;
  v.e = e;
  v.neg = neg ? 1u : 0u;
  return v;
}
struct Entry
{
  uint64_t mant;
  int e;
  bool neg;
};
template <int NL> struct Host
{
  std::vector<uint32_t> w;
  size_t n;
  explicit Host(size_t count) : w((size_t)(NL + 1) * count, 0u), n(count) {}
  mw::Ptr ptr() { return mw::Ptr{w.data(), n}; }
};
template <int NL> struct Dev
{
  uint32_t *p = nullptr;
  size_t n;
  explicit Dev(size_t count) : n(count)
  {
    HIP_CHECK(hipMalloc(&p, (size_t)(NL + 1) * n * 4));
    HIP_CHECK(hipMemset(p, 0, (size_t)(NL + 1) * n * 4));
  }
  ~Dev() { (void)hipFree(p); }
  mw::Ptr ptr() { return mw::Ptr{p, n}; }
  void up(Host<NL> &h) { HIP_CHECK(hipMemcpy(p, h.w.data(), h.w.size() * 4, hipMemcpyHostToDevice)); }
  void down(Host<NL> &h) { HIP_CHECK(hipMemcpy(h.w.data(), p, h.w.size() * 4, hipMemcpyDeviceToHost)); }
};
struct Results
{
  std::vector<std::vector<uint32_t>> m; // top limbs first
  std::vector<int> e, neg;
};
template <int NL> void push(Results &r, const Mw<NL> &v)
{
  std::vector<uint32_t> m(NL);
  for(int i = 0; i < NL; ++i)
    m[i] = v.m[NL - 1 - i];
  r.m.push_back(m);
  r.e.push_back(v.e);
  r.neg.push_back((int)v.neg);
}
// log2 |a - b| / |a| from the leading limbs the two have in common
static double log2_rel(const Results &a, const Results &b, size_t i)
{
  if(a.e[i] != b.e[i] || a.neg[i] != b.neg[i])
    return 0.0;
  const size_t n = std::min(a.m[i].size(), b.m[i].size());
  for(size_t k = 0; k < n; ++k)
    if(a.m[i][k] != b.m[i][k])
      {
        const double d = std::fabs((double)a.m[i][k] - (double)b.m[i][k]);
        return -32.0 * (double)k - 32.0 + std::log2(d) + 1.0; // leading limb is normalised: value in [1/2, 1)
      }
  return -32.0 * (double)n;
}

template <int NL> Results run(const std::vector<Entry> &ent, int M, int n)
{
  Results out;
  const size_t per = (size_t)n * n;
  Host<NL> hA(per * M);
  for(int q = 0; q < M; ++q)
    for(int i = 0; i < n; ++i)
      for(int j = 0; j <= i; ++j)
        {
          const Entry &x = ent[(size_t)q * per + (size_t)i * n + j];
          Mw<NL> v = from_bits<NL>(x.mant, i == j ? 3 : x.e, i == j ? false : x.neg);
          mw::store<NL>(hA.ptr(), (size_t)q * per + i + (size_t)j * n, v);
          mw::store<NL>(hA.ptr(), (size_t)q * per + j + (size_t)i * n, v);
        }
  Dev<NL> dA(per * M), dD((size_t)n * M + 1), dE((size_t)n * M + 1), dlam(M), dred(512), dres(8), dratio(M);
  dA.up(hA);
  std::vector<MatDesc> ha(M), hd(M);
  std::vector<int> ids(M);
  for(int q = 0; q < M; ++q)
    {
      ha[q] = MatDesc{(unsigned long long)q * per, n, n, n, 0};
      hd[q] = MatDesc{(unsigned long long)q * n, n, 1, n, 0};
      ids[q] = q;
    }
  MatDesc *da, *dd;
  int *dids;
  double *dF;
  HIP_CHECK(hipMalloc(&da, M * sizeof(MatDesc)));
  HIP_CHECK(hipMalloc(&dd, M * sizeof(MatDesc)));
  HIP_CHECK(hipMalloc(&dids, M * sizeof(int)));
  HIP_CHECK(hipMalloc(&dF, 2 * ((size_t)n * M + 2) * sizeof(double)));
  HIP_CHECK(hipMemcpy(da, ha.data(), M * sizeof(MatDesc), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dd, hd.data(), M * sizeof(MatDesc), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dids, ids.data(), M * sizeof(int), hipMemcpyHostToDevice));
  const Batch A{dA.ptr(), da, M}, D{dD.ptr(), dd, M}, Eb{dE.ptr(), dd, M};
  // (1) ratio of the diagonal (k_diag_ratio) BEFORE the matrices are overwritten by the tridiagonalisation
  hipLaunchKernelGGL((k_diag_ratio<NL>), dim3(M), dim3(DR_T), 0, 0, A, dratio.ptr(), (size_t)0);
  // (2) Householder tridiagonalisation and lambda_min
  hipLaunchKernelGGL((k_tridiag<NL, 256>), dim3(M), dim3(256), 0, 0, A, D, Eb, (const int *)dids);
  HIP_CHECK(hipDeviceSynchronize());
  Host<NL> hD((size_t)n * M + 1), hE((size_t)n * M + 1);
  dD.down(hD);
  dE.down(hE);
  hipLaunchKernelGGL((k_tridiag_min<NL>), dim3(cdiv(M, EIG_T)), dim3(EIG_T), 0, 0, D, Eb, dF, dF + (size_t)n * M + 1, dlam.ptr());
  // (3) the two-level reductions of Solver::reduce_to
  mw::CPtr lp = dlam.ptr(), rp = dratio.ptr();
  auto ldl = [=] __device__(size_t i) { return mw::load<NL>(lp, i); };
  auto ldr = [=] __device__(size_t i) { return mw::load<NL>(rp, i); };
  const unsigned g = std::min<unsigned>(cdiv((size_t)M, WG), 512);
  mw::CPtr redp = dred.ptr();
  auto ld2 = [=] __device__(size_t i) { return mw::load<NL>(redp, i); };
  hipLaunchKernelGGL((k_reduce<NL, RED_MIN, decltype(ldl)>), dim3(g), dim3(WG), 0, 0, (size_t)M, ldl, dred.ptr());
  hipLaunchKernelGGL((k_reduce<NL, RED_MIN, decltype(ld2)>), dim3(1), dim3(WG), 0, 0, (size_t)g, ld2, mw::Ptr{dres.p + 0, dres.n});
  hipLaunchKernelGGL((k_reduce<NL, RED_MAX, decltype(ldr)>), dim3(g), dim3(WG), 0, 0, (size_t)M, ldr, dred.ptr());
  hipLaunchKernelGGL((k_reduce<NL, RED_MAX, decltype(ld2)>), dim3(1), dim3(WG), 0, 0, (size_t)g, ld2, mw::Ptr{dres.p + 1, dres.n});
  HIP_CHECK(hipDeviceSynchronize());
  Host<NL> hlam(M), hres(8), hratio(M);
  dlam.down(hlam);
  dres.down(hres);
  dratio.down(hratio);
  // records: D and E of matrix 0, every lambda_min, every ratio, the two reductions
  for(int i = 0; i < n; ++i)
    push<NL>(out, mw::load<NL>(hD.ptr(), i));
  for(int i = 1; i < n; ++i)
    push<NL>(out, mw::load<NL>(hE.ptr(), i));
  for(int q = 0; q < M; ++q)
    push<NL>(out, mw::load<NL>(hlam.ptr(), q));
  for(int q = 0; q < M; ++q)
    push<NL>(out, mw::load<NL>(hratio.ptr(), q));
  push<NL>(out, mw::load<NL>(hres.ptr(), 0));
  push<NL>(out, mw::load<NL>(hres.ptr(), 1));
  // the reductions against the host's own minimum / maximum of the downloaded arrays: bit for bit
  Mw<NL> mn = mw::load<NL>(hlam.ptr(), 0), mx = mw::load<NL>(hratio.ptr(), 0);
  for(int q = 1; q < M; ++q)
    {
      mn = mw::min(mn, mw::load<NL>(hlam.ptr(), q));
      mx = mw::max(mx, mw::load<NL>(hratio.ptr(), q));
    }
  const bool okmin = mw::cmp(mn, mw::load<NL>(hres.ptr(), 0)) == 0, okmax = mw::cmp(mx, mw::load<NL>(hres.ptr(), 1)) == 0;
  std::printf("NL=%d: k_reduce<RED_MIN> %s the host minimum of the %d lambda_min; k_reduce<RED_MAX> %s the host maximum of the ratios\n", NL,
              okmin ? "==" : "!=", M, okmax ? "==" : "!=");
  (void)hipFree(da);
  (void)hipFree(dd);
  (void)hipFree(dids);
  (void)hipFree(dF);
  return out;
}

// Ingredients of k_chol_inv_lds one at a time, device against the HOST's evaluation of the same mw:: functions (the host
// arithmetic is exact against GMP at every width: tests/shim, profiles/r06_*): (1) the wavefront's reciprocal square root
// wv::rsqrt (above 58 limbs: the one-lane ladder on a broadcast copy), (2) an mw::Acc sum of products with mixed signs and
// exponents, (3) the limb-major LDS image ci_st -> ci_ld.
template <int NL> __global__ void __launch_bounds__(64) k_ingredients(mw::CPtr in, int K, mw::Ptr out)
{
  const int t = threadIdx.x;
  __shared__ uint32_t img[(NL + CI_PLANES_EXTRA) * CI_NPK];
  // (1) lane 5 owns the argument
  Mw<NL> a = mw::zero<NL>();
  if(t == 5)
    a = mw::load<NL>(in, 0);
  const Mw<NL> r = mw::wv::rsqrt<NL>(a, 5);
  if(t == 9)
    mw::store<NL>(out, 0, r);
  const Mw<NL> sq = mw::wv::sqrt<NL>(a, 5);
  if(t == 11)
    mw::store<NL>(out, 1, sq);
  const Mw<NL> rc = mw::wv::rcp<NL>(a, 5);
  if(t == 13)
    mw::store<NL>(out, 2, rc);
  // (2) every lane the same sum (lane 17 stores)
  mw::Acc<NL> acc = mw::acc_zero<NL>();
  for(int k = 0; k < K; ++k)
    {
      const Mw<NL> x = mw::load<NL>(in, 1 + 2 * k), y = mw::load<NL>(in, 2 + 2 * k);
      if(k % 3 == 2)
        mw::acc_fms(acc, x, y);
      else
        mw::acc_fma(acc, x, y);
    }
  const Mw<NL> sum = mw::acc_result(acc);
  if(t == 17)
    mw::store<NL>(out, 3, sum);
  // (3) LDS image round trip of the first CI_NPK inputs
  for(int i = t; i < CI_NPK && i < 2 * K; i += 64)
    ci_st<NL>(img, i, mw::load<NL>(in, 1 + i));
  __syncthreads();
  for(int i = t; i < CI_NPK && i < 2 * K; i += 64)
    mw::store<NL>(out, 4 + i, ci_ld<NL>(img, i));
}
template <int NL> void run_ingredients()
{
  const int K = 40;
  Host<NL> hin(1 + 2 * K), hout(4 + 2 * K);
  auto dense = [&](int e, bool neg) {
    Mw<NL> v;
    for(int i = 0; i < NL; ++i)
      v.m[i] = (uint32_t)lcg();
    v.m[NL - 1] |= 0x80000000u;
    v.e = e;
    v.neg = neg ? 1u : 0u;
    return v;
  };
  mw::store<NL>(hin.ptr(), 0, dense(7, false));
  for(int k = 0; k < 2 * K; ++k)
    mw::store<NL>(hin.ptr(), 1 + k, dense((int)(lcg() % 200) - 100, lcg() & 1));
  Dev<NL> din(1 + 2 * K), dout(4 + 2 * K);
  din.up(hin);
  hipLaunchKernelGGL((k_ingredients<NL>), dim3(1), dim3(64), 0, 0, mw::CPtr(din.p, din.n), K, dout.ptr());
  HIP_CHECK(hipDeviceSynchronize());
  dout.down(hout);
  const Mw<NL> a = mw::load<NL>(hin.ptr(), 0);
  mw::Acc<NL> acc = mw::acc_zero<NL>();
  for(int k = 0; k < K; ++k)
    {
      const Mw<NL> x = mw::load<NL>(hin.ptr(), 1 + 2 * k), y = mw::load<NL>(hin.ptr(), 2 + 2 * k);
      if(k % 3 == 2)
        mw::acc_fms(acc, x, y);
      else
        mw::acc_fma(acc, x, y);
    }
  Results dev, host;
  push<NL>(dev, mw::load<NL>(hout.ptr(), 0));
  push<NL>(host, mw::rsqrt<NL>(a));
  push<NL>(dev, mw::load<NL>(hout.ptr(), 1));
  push<NL>(host, mw::sqrt<NL>(a));
  push<NL>(dev, mw::load<NL>(hout.ptr(), 2));
  push<NL>(host, mw::rcp<NL>(a));
  push<NL>(dev, mw::load<NL>(hout.ptr(), 3));
  push<NL>(host, mw::acc_result(acc));
  double img = -1e9;
  for(int i = 0; i < CI_NPK && i < 2 * K; ++i)
    {
      Results x, y;
      push<NL>(x, mw::load<NL>(hout.ptr(), 4 + i));
      push<NL>(y, mw::load<NL>(hin.ptr(), 1 + i));
      img = std::max(img, log2_rel(x, y, 0));
    }
  std::printf("NL=%d ingredients, device against host (log2 relative difference; %d = identical): wv::rsqrt %.1f  wv::sqrt %.1f  wv::rcp %.1f  Acc sum of %d products %.1f  LDS image round trip %.1f\n",
              NL, -32 * NL, log2_rel(dev, host, 0), log2_rel(dev, host, 1), log2_rel(dev, host, 2), K, log2_rel(dev, host, 3), img);
}

// The batched Cholesky chain (Solver::blocked_cholesky: k_chol_inv_lds, k_chol_panel_solve, k_chol_syrk_down per panel) of
// diagonally dominant SPD matrices, and P = L^-1 B by k_trsm_rlt_panel on a right-hand side of all ones.
template <int NL> Results run_chol(const std::vector<Entry> &ent, int M, int n)
{
  Results out;
  const size_t per = (size_t)n * n;
  Host<NL> hA(per * M), hX(per * M);
  for(int q = 0; q < M; ++q)
    for(int i = 0; i < n; ++i)
      for(int j = 0; j <= i; ++j)
        {
          const Entry &x = ent[(size_t)q * per + (size_t)i * n + j];
          Mw<NL> v = from_bits<NL>(x.mant, i == j ? 3 : -3 - (x.e & 1), i == j ? false : x.neg);
          mw::store<NL>(hA.ptr(), (size_t)q * per + i + (size_t)j * n, v);
          mw::store<NL>(hA.ptr(), (size_t)q * per + j + (size_t)i * n, v);
        }
  for(size_t k = 0; k < per * M; ++k)
    mw::store<NL>(hX.ptr(), k, from_bits<NL>(ent[k].mant, 1, ent[k].neg));
  Dev<NL> dA(per * M), dLi(per * M), dinv((size_t)n * M + 1), dX(per * M);
  dA.up(hA);
  dX.up(hX);
  std::vector<MatDesc> ha(M), hd(M);
  for(int q = 0; q < M; ++q)
    {
      ha[q] = MatDesc{(unsigned long long)q * per, n, n, n, 0};
      hd[q] = MatDesc{(unsigned long long)q * n, n, 1, n, 0};
    }
  MatDesc *da, *dd;
  int *fail;
  HIP_CHECK(hipMalloc(&da, M * sizeof(MatDesc)));
  HIP_CHECK(hipMalloc(&dd, M * sizeof(MatDesc)));
  HIP_CHECK(hipMalloc(&fail, 64 * sizeof(int)));
  HIP_CHECK(hipMemset(fail, 0, 64 * sizeof(int)));
  HIP_CHECK(hipMemcpy(da, ha.data(), M * sizeof(MatDesc), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dd, hd.data(), M * sizeof(MatDesc), hipMemcpyHostToDevice));
  const Batch A{dA.ptr(), da, M}, Li{dLi.ptr(), da, M}, invd{dinv.ptr(), dd, M}, X{dX.ptr(), da, M};
  const int panels = (n + PB - 1) / PB;
  for(int p = 0; p < panels; ++p)
    {
      hipLaunchKernelGGL((k_chol_inv_lds<NL>), dim3(M), dim3(CI_T), 0, 0, A, invd, Li, p, fail, (unsigned long long *)nullptr);
      const int below = n - PB * (p + 1), above = PB * p, rows = below > above ? below : above;
      if(rows > 0)
        hipLaunchKernelGGL((k_chol_panel_solve<NL>), dim3(cdiv(rows, TR), M), dim3(WG), 0, 0, A, Li, p, 0, (unsigned long long *)nullptr);
      if(below > 0)
        {
          const unsigned tiles = cdiv(below, 16);
          hipLaunchKernelGGL((k_chol_syrk_down<NL>), dim3(tiles * (tiles + 1) / 2, M), dim3(WG), 0, 0, A, p, 0, (unsigned long long *)nullptr, 0, 0x7fffffff);
        }
      if(p == 0)
        {
          HIP_CHECK(hipDeviceSynchronize());
          Host<NL> h0(per * M);
          dA.down(h0);
          for(int q = 0; q < 4; ++q) // the first diagonal block after k_chol_inv_lds alone, and what the panel solve left below it
            for(int j = 0; j < (n < PB ? n : PB); ++j)
              for(int i = j; i < n; ++i)
                push<NL>(out, mw::load<NL>(h0.ptr(), (size_t)q * per + i + (size_t)j * n));
        }
    }
  const mw::CPtr src(dX.p, dX.n);
  for(int p = 0; p < panels; ++p)
    hipLaunchKernelGGL((k_trsm_rlt_panel<NL>), dim3(cdiv(n, TR), M), dim3(WG), 0, 0, A, Li, X, src, p, (unsigned long long *)nullptr);
  HIP_CHECK(hipDeviceSynchronize());
  Host<NL> hL(per * M), hP(per * M);
  dA.down(hL);
  dX.down(hP);
  for(int q = 0; q < M; ++q)
    for(int j = 0; j < n; ++j)
      for(int i = j; i < n; ++i)
        push<NL>(out, mw::load<NL>(hL.ptr(), (size_t)q * per + i + (size_t)j * n));
  for(size_t k = 0; k < per * M; ++k)
    push<NL>(out, mw::load<NL>(hP.ptr(), k));
  int hf[4];
  HIP_CHECK(hipMemcpy(hf, fail, sizeof hf, hipMemcpyDeviceToHost));
  std::printf("NL=%d: Cholesky failure flags %d %d\n", NL, hf[0], hf[1]);
  (void)hipFree(da);
  (void)hipFree(dd);
  (void)hipFree(fail);
  return out;
}

int main()
{
#ifdef PROBE_INGREDIENTS
  run_ingredients<PROBE_A>();
  run_ingredients<PROBE_B>();
#endif
  {
    const int M = 64, n = 40;
    std::vector<Entry> ent((size_t)M * n * n);
    for(auto &x : ent)
      {
        x.mant = lcg();
        x.e = (int)(lcg() % 5) - 2;
        x.neg = lcg() & 1;
      }
    const Results a = run_chol<PROBE_A>(ent, M, n), b = run_chol<PROBE_B>(ent, M, n);
    auto worst = [&](size_t lo, size_t hi) {
      double w = -1e9;
      for(size_t i = lo; i < hi; ++i)
        w = std::max(w, log2_rel(a, b, i));
      return w;
    };
    const int nb = n < PB ? n : PB;
    size_t first = 0;
    for(int j = 0; j < nb; ++j)
      first += (size_t)(n - j);
    const size_t tri = (size_t)n * (n + 1) / 2;
    std::printf("Cholesky chain, %d matrices of order %d, panels of %d: worst log2 relative difference between %d and %d limbs\n", M, n, PB, PROBE_A, PROBE_B);
    std::printf("  after panel 0 (k_chol_inv_lds + k_chol_panel_solve), 4 matrices   %8.1f\n", worst(0, 4 * first));
    {
      // the same entries split: the diagonal block (k_chol_inv_lds alone), column by column, and the rows below it (k_chol_panel_solve)
      double wd = -1e9, wb = -1e9;
      size_t o = 0;
      std::printf("  diagonal block of matrix 0, worst per column:");
      for(int j = 0; j < nb; ++j)
        {
          double wc = -1e9;
          for(int i = j; i < n; ++i, ++o)
            {
              const double l = log2_rel(a, b, o);
              if(i < nb)
                {
                  wd = std::max(wd, l);
                  wc = std::max(wc, l);
                }
              else
                wb = std::max(wb, l);
            }
          std::printf(" %.0f", wc);
        }
      std::printf("\n  matrix 0: diagonal block (k_chol_inv_lds) %8.1f   rows below it (k_chol_panel_solve) %8.1f\n", wd, wb);
    }
    std::printf("  the finished factors L                                           %8.1f\n", worst(4 * first, 4 * first + M * tri));
    std::printf("  X L^-T by k_trsm_rlt_panel                                       %8.1f\n", worst(4 * first + M * tri, a.e.size()));
  }
  const int M = 600, n = 20;
  std::vector<Entry> ent((size_t)M * n * n);
  for(auto &x : ent)
    {
      x.mant = lcg();
      x.e = (int)(lcg() % 5) - 2;
      x.neg = lcg() & 1;
    }
  const Results a = run<PROBE_A>(ent, M, n), b = run<PROBE_B>(ent, M, n);
  auto worst = [&](size_t lo, size_t hi) {
    double w = -1e9;
    for(size_t i = lo; i < hi; ++i)
      w = std::max(w, log2_rel(a, b, i));
    return w;
  };
  size_t o = 0;
  std::printf("widths %d and %d limbs on the same %d matrices of order %d: worst log2 relative difference\n", PROBE_A, PROBE_B, M, n);
  std::printf("  k_tridiag D of matrix 0      %8.1f\n", worst(o, o + n));
  o += n;
  std::printf("  k_tridiag E of matrix 0      %8.1f\n", worst(o, o + n - 1));
  o += n - 1;
  std::printf("  k_tridiag_min, all matrices  %8.1f\n", worst(o, o + M));
  o += M;
  std::printf("  k_diag_ratio, all matrices   %8.1f\n", worst(o, o + M));
  o += M;
  std::printf("  min of lambda_min (k_reduce) %8.1f\n", worst(o, o + 1));
  std::printf("  max of ratios (k_reduce)     %8.1f\n", worst(o + 1, o + 2));
  return 0;
}
