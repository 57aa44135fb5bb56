// kernels.hpp — HIP kernels (gfx950) for one interior-point iteration of sdp_solve.
//
// Work decomposition: SDP blocks are independent, so every per-block
// operation is a batched launch — blockIdx.y (or blockIdx.x for workgroup-per-matrix
// kernels) selects the matrix through a descriptor table, threads of a 256-lane
// workgroup (4 wavefronts of 64) own output elements.  Each lane carries whole
// multi-word numbers in VGPRs (mw.hpp); with limb-major storage a wavefront's loads
// of "limb l of 64 consecutive elements" are single coalesced 256-B transactions.
// The arithmetic intensity is ~NL^2/2 integer MACs per 4(NL+1) bytes, so these
// kernels are bound by the v_mad_u64_u32 pipe, not by HBM.
//
// Reference functions each kernel replaces are cited at the kernel.
#pragma once
#include "dev.hpp"
#include "mw_wave.hpp"
#include "tiledot.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace sdpb
{
using mw::Mw;
using mw::Acc;
constexpr int WG = 256; // workgroup size used by every kernel

template <int NL> MW_HD Mw<NL> mat_ld(const Batch &b, const MatDesc &d, int i, int j)
{
  return mw::load<NL>(b.p, (size_t)d.off + (size_t)i + (size_t)j * (size_t)d.ld);
}
template <int NL> MW_HD void mat_st(const Batch &b, const MatDesc &d, int i, int j, const Mw<NL> &v)
{
  mw::store<NL>(b.p, (size_t)d.off + (size_t)i + (size_t)j * (size_t)d.ld, v);
}

// limb-major LDS image of STRIDE numbers (element idx of limb l at l*STRIDE + idx):
// consecutive lanes reading consecutive elements hit distinct banks
template <int NL, int STRIDE> MW_HD Mw<NL> smem_ld(const uint32_t *s, int idx)
{
  Mw<NL> v;
#pragma unroll
  for(int l = 0; l < NL; ++l)
    v.m[l] = s[l * STRIDE + idx];
  v.e = (int32_t)s[NL * STRIDE + idx];
  v.neg = s[(NL + 1) * STRIDE + idx];
  return v;
}
template <int NL, int STRIDE> MW_HD void smem_st(uint32_t *s, int idx, const Mw<NL> &v)
{
#pragma unroll
  for(int l = 0; l < NL; ++l)
    s[l * STRIDE + idx] = v.m[l];
  s[NL * STRIDE + idx] = (uint32_t)v.e;
  s[(NL + 1) * STRIDE + idx] = v.neg;
}

// ---------------------------------------------------------------------------
// Generic element-wise and reduction drivers
// ---------------------------------------------------------------------------
template <class F> __global__ void __launch_bounds__(WG) k_foreach(size_t count, F f)
{
  for(size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < count; i += (size_t)gridDim.x * WG)
    f(i);
}

enum ReduceOp
{
  RED_SUM = 0,
  RED_MAX = 1,
  RED_MIN = 2
};
template <int NL, int OP> MW_HD Mw<NL> red_combine(const Mw<NL> &a, const Mw<NL> &b)
{
  if(OP == RED_SUM)
    return mw::add(a, b);
  if(OP == RED_MAX)
    return mw::max(a, b);
  return mw::min(a, b);
}
// the multi-word number lane `src` of this wavefront holds (every lane of the wavefront must call it)
template <int NL> __device__ Mw<NL> wave_get(const Mw<NL> &v, int src)
{
  Mw<NL> o;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for(int l = 0; l < NL; ++l)
    o.m[l] = (uint32_t)__shfl((int)v.m[l], src, 64);
  o.e = __shfl(v.e, src, 64);
  o.neg = (uint32_t)__shfl((int)v.neg, src, 64);
#else
  o = v; // the emulation build never calls it (its lanes are fibres, not a wavefront)
  (void)src;
#endif
  return o;
}
// Workgroup reduction; result valid in thread 0.  Inside a wavefront the 64 partial results meet by
// wavefront shuffles (__shfl_down: lane t takes lane t + s, s = 32 ... 1, one word of the multi-word
// number at a time — no LDS traffic, no barrier); the four wavefront results then meet through LDS in a
// fixed order.  The CPU emulation build walks the same tree through memory, so both give the same bits.
template <int NL, int OP> __device__ Mw<NL> wg_reduce(Mw<NL> v, bool has)
{
  constexpr int WAVES = WG / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int h = has ? 1 : 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for(int s = 32; s > 0; s >>= 1)
    {
      Mw<NL> o;
#pragma unroll
      for(int l = 0; l < NL; ++l)
        o.m[l] = (uint32_t)__shfl_down((int)v.m[l], s, 64);
      o.e = __shfl_down(v.e, s, 64);
      o.neg = (uint32_t)__shfl_down((int)v.neg, s, 64);
      const int oh = __shfl_down(h, s, 64);
      if(lane + s < 64 && oh)
        {
          v = h ? red_combine<NL, OP>(v, o) : o;
          h = 1;
        }
    }
#else
  {
    __shared__ Mw<NL> sm[WG];
    __shared__ int sh[WG];
    sm[t] = v;
    sh[t] = h;
    __syncthreads();
    for(int s = 32; s > 0; s >>= 1)
      {
        // every lane reads its partner before any lane of the step writes (the shuffle's semantics)
        const bool take = lane + s < 64 && sh[lane + s < 64 ? t + s : t];
        const Mw<NL> o = sm[lane + s < 64 ? t + s : t];
        __syncthreads();
        if(take)
          {
            sm[t] = sh[t] ? red_combine<NL, OP>(sm[t], o) : o;
            sh[t] = 1;
          }
        __syncthreads();
      }
    v = sm[t];
    h = sh[t];
    __syncthreads();
  }
#endif
  __shared__ Mw<NL> sw[WAVES];
  __shared__ int shw[WAVES];
  if(lane == 0)
    {
      sw[wave] = v;
      shw[wave] = h;
    }
  __syncthreads();
  Mw<NL> r = mw::zero<NL>();
  if(t == 0)
    {
      // pairs of wavefronts, then the pairs: ((w0 . w1) . (w2 . w3))
      for(int step = 1; step < WAVES; step <<= 1)
        for(int w = 0; w + step < WAVES; w += 2 * step)
          if(shw[w + step])
            {
              sw[w] = shw[w] ? red_combine<NL, OP>(sw[w], sw[w + step]) : sw[w + step];
              shw[w] = 1;
            }
      r = sw[0];
    }
  __syncthreads();
  return r;
}
// out[blockIdx.x] = OP over f(i), i in this block's grid-stride range; empty -> 0
template <int NL, int OP, class F> __global__ void __launch_bounds__(WG) k_reduce(size_t count, F f, mw::Ptr out)
{
  Mw<NL> acc = mw::zero<NL>();
  bool has = false;
  if(OP == RED_SUM)
    {
      Acc<NL> sum = mw::acc_zero<NL>();
      for(size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < count; i += (size_t)gridDim.x * WG)
        {
          mw::acc_add(sum, f(i));
          has = true;
        }
      acc = mw::acc_result(sum);
    }
  else
    for(size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < count; i += (size_t)gridDim.x * WG)
      {
        const Mw<NL> v = f(i);
        acc = has ? red_combine<NL, OP>(acc, v) : v;
        has = true;
      }
  const Mw<NL> r = wg_reduce<NL, OP>(acc, has);
  if(threadIdx.x == 0)
    mw::store<NL>(out, blockIdx.x, r);
}

// ---------------------------------------------------------------------------
// Blocked triangular machinery (batched).  Every dependent multi-word multiply-add
// costs ~1.5 us of latency, so anything triangular is processed in panels of
// PB = 32 columns: the PB x PB diagonal factor is factored AND inverted inside LDS
// (k_chol_inv_lds), after which panel solves, trailing updates and all later
// triangular solves are short dot products spread over many lanes.  `Li` arrays have
// the shape of the factor L and hold, in each diagonal PB x PB block, the inverse of
// that block of L.  Panel p covers rows/columns [PB*p, min(PB*(p+1), n)).
// Replaces El::Cholesky(LOWER/UPPER) (cholesky_decomposition.cxx:17, compute_Q.cxx:31,
// initialize_schur_complement_solver.cxx:98) and El::Trsm (compute_A_X_inv.cxx:21,
// cholesky_solve.cxx:9, compute_Q.cxx:48, lower_triangular_inverse_congruence.cxx:8-13,
// lower_triangular_solve.hxx:10, lower_triangular_transpose_solve.cxx:6).
// ---------------------------------------------------------------------------
#ifndef SDPB_PB
#define SDPB_PB 32
#endif
constexpr int PB = SDPB_PB; // panel width (tests also build a PB = 4 variant to exercise ragged multi-panel paths)
constexpr int CI_T = 256;      // lanes of the diagonal-block kernel: one wavefront per SIMD of a CU
constexpr int CI_D0 = CI_T - 64; // first lane of the wavefront that owns the diagonal
static_assert(CI_D0 % 64 == 0 && PB <= 64, "lane k of the diagonal's wavefront owns (k, k)");
constexpr int CI_NPK = PB * (PB + 1) / 2;                             // packed lower triangle (LDS images)
// Off-diagonal elements are dealt to lanes as a round-robin tournament: round m (PB-1 of
// them) pairs up all PB indices, and a lane takes up to three pairs of ONE round, so the
// (up to three) elements of a lane never share a row or column index — at every pivot k
// a lane finishes at most one element of column k / row k.
constexpr int CI_E = 3;                                 // off-diagonal elements per lane
constexpr int CI_GPR = (PB / 2 + CI_E - 1) / CI_E;      // lanes per round
static_assert(PB % 2 == 0 && PB <= 64 && (PB - 1) * CI_GPR <= CI_D0, "k_chol_inv_lds: element-to-lane mapping");

// limb-major LDS image of a packed triangle (conflict-free for consecutive elements); sign and
// exponent share one word (NL + 1 planes), which brings the two triangles of a 32 x 32 block at
// 576 bits under half of the CU's LDS: two workgroups per CU in the batched launches
constexpr int CI_PLANES_EXTRA = 1;
template <int NL> MW_HD Mw<NL> ci_ld(const uint32_t *s, int idx)
{
  Mw<NL> v;
#pragma unroll
  for(int l = 0; l < NL; ++l)
    v.m[l] = s[l * CI_NPK + idx];
  const uint32_t hd = s[NL * CI_NPK + idx];
  v.neg = hd >> 31;
  v.e = hd ? (int32_t)((hd & 0x7fffffffu) - mw::EBIAS) : mw::EZERO;
  return v;
}
template <int NL> MW_HD void ci_st(uint32_t *s, int idx, const Mw<NL> &v)
{
  const bool z = v.e == mw::EZERO;
#pragma unroll
  for(int l = 0; l < NL; ++l)
    s[l * CI_NPK + idx] = z ? 0u : v.m[l];
  s[NL * CI_NPK + idx] = z ? 0u : ((v.neg << 31) | (uint32_t)(v.e + (int32_t)mw::EBIAS));
}

// One element (r,c), r > c, of a diagonal block, or a diagonal element (r == c).
template <int NL> struct CiElem
{
  int r, c; // r < 0: no element
  Acc<NL> acc;
};
#define SDPB_PK(r, c) ((c) * n - (c) * ((c)-1) / 2 + ((r) - (c)))
// pair number i of round m of the tournament on PB players (circle method); the element is
// dropped when one of its indices lies outside the n x n block
template <int NL> MW_HD void ci_init_offdiag(CiElem<NL> &el, int m, int i, int n, const Batch &A, const MatDesc &d, int k0)
{
  int a, b;
  if(i == 0)
    {
      a = PB - 1;
      b = m;
    }
  else
    {
      a = (m + i) % (PB - 1);
      b = (m - i + (PB - 1)) % (PB - 1);
    }
  const bool valid = m < PB - 1 && i < PB / 2 && a < n && b < n;
  el.r = valid ? (a > b ? a : b) : -1;
  el.c = valid ? (a > b ? b : a) : 0;
  el.acc = mw::acc_zero<NL>();
  if(valid)
    mw::acc_add(el.acc, mat_ld<NL>(A, d, k0 + el.r, k0 + el.c));
}
// (2) finish column k of L and row k of X:  L(r,k) = acc / L_kk,  X(k,c) = -acc / L_kk
// `sum` = acc_result(el.acc), taken by the caller BEFORE the barrier that publishes the pivot: the normalisation does
// not depend on the pivot, so it runs while the diagonal's wavefront is busy with the reciprocal square root
template <int NL> MW_HD void ci_finish(CiElem<NL> &el, const Mw<NL> &sum, int k, int n, uint32_t *sL, uint32_t *sX, const uint32_t *s_inv)
{
  const bool colk = el.c == k && el.r > k, rowk = el.r == k && el.c < k;
  if(!(colk || rowk))
    return;
  Mw<NL> inv;
#pragma unroll
  for(int l = 0; l < NL; ++l)
    inv.m[l] = s_inv[l];
  inv.e = (int32_t)s_inv[NL];
  inv.neg = s_inv[NL + 1];
  Mw<NL> v = mw::mul(sum, inv);
  v.neg ^= (rowk && !mw::is_zero(v)) ? 1u : 0u;
  ci_st<NL>(rowk ? sX : sL, SDPB_PK(el.r, el.c), v); // one code path for both kinds (lanes of a wavefront hold both)
  el.acc = mw::acc_zero<NL>();
}
// (3) one product for an element below row k: trailing update (c > k) acc -= L(r,k) L(c,k),
// inverse (c <= k) acc += L(r,k) X(k,c) — one code path
template <int NL> MW_HD void ci_update(CiElem<NL> &el, int k, int n, const uint32_t *sL, const uint32_t *sX)
{
  if(el.r <= k)
    return;
  const bool trail = el.c > k;
  const Mw<NL> lrk = ci_ld<NL>(sL, SDPB_PK(el.r, k));
  const Mw<NL> other = ci_ld<NL>(trail ? sL : sX, trail ? SDPB_PK(el.c, k) : SDPB_PK(k, el.c));
  mw::acc_fma(el.acc, lrk, other, trail ? 1u : 0u);
}

// Diagonal block p of every matrix: A_pp = L L^T in place (upper part zeroed),
// Li_pp = X = L^{-1}, invd = 1/L_ii.  fail[q] = 1 + index of a non-positive pivot.
// Every element (r,c) of the lower triangle is owned by one lane and lives in a
// register accumulator (mw::Acc: one aligned add per term, no normalisation) that is
// used twice: until pivot c it collects A(r,c) - sum_{k<c} L(r,k) L(c,k); after it
// has been finished into L(r,c) it collects sum_{c<=k<r} L(r,k) X(k,c), finished at
// pivot r into X(r,c) = -acc / L_rr.  Per pivot k: (1) the owner of (k,k) takes the
// reciprocal square root — the only long dependent chain, ~9 us at 576 bits — (2) column
// k of L and row k of X are finished and published in LDS, (3) every element below row
// k absorbs exactly one product.  The diagonal has a wavefront of its own (lane
// CI_D0 + k owns (k,k)) whose step (3) is a single product, and there is no barrier
// between (3) and the next (1): the next rsqrt starts while the other wavefronts are
// still busy with their three products per lane.
// Wavefronts of the dependent chains (the pivot chain of Cholesky(Q), its strip kernels, the Q
// substitutions) share their SIMDs with whatever the other streams run beside them; s_setprio lets the
// arbiter issue the chain's instructions first (a lone latency-bound wavefront otherwise waits its turn
// behind four throughput-bound ones).
#ifndef SDPB_CHAIN_PRIO
#define SDPB_CHAIN_PRIO 3
#endif
// Per-block cost measurement (the reference times Cholesky and Trsm of every
// block: compute_Q.cxx:40-53): while an iteration is profiled, every workgroup of the Schur-stage
// kernels adds its residence time (constant 100 MHz wall clock) to the counter of the block it
// works on.  cyc == nullptr (every unprofiled iteration): no clock read, no atomic.
struct WgClock
{
  unsigned long long *slot;
  unsigned long long t0;
  __device__ WgClock(unsigned long long *cyc, int idx) : slot(cyc ? cyc + idx : nullptr), t0(cyc ? wall_clock64() : 0ull) {}
  __device__ ~WgClock()
  {
    if(slot && threadIdx.x == 0)
      atomicAdd(slot, wall_clock64() - t0);
  }
};
MW_HD void raise_chain_priority()
{
#if defined(__HIP_DEVICE_COMPILE__)
  if(SDPB_CHAIN_PRIO > 0)
    __builtin_amdgcn_s_setprio(SDPB_CHAIN_PRIO);
#endif
}
template <int NL> __global__ void __launch_bounds__(CI_T) k_chol_inv_lds(Batch A, Batch invd, Batch Li, int p, int *fail, unsigned long long *cyc)
{
  raise_chain_priority();
  const int q = blockIdx.x;
  WgClock clk(cyc, q);
  const MatDesc d = A.d[q], dv = invd.d[q], di = Li.d[q];
  const int k0 = PB * p, t = threadIdx.x;
  if(k0 >= d.rows)
    return;
  const int n = d.rows - k0 < PB ? d.rows - k0 : PB;
  __shared__ uint32_t sL[(NL + CI_PLANES_EXTRA) * CI_NPK], sX[(NL + CI_PLANES_EXTRA) * CI_NPK];
  __shared__ uint32_t s_inv[NL + 2];
  __shared__ int s_fail;
  // the lane's elements, as separate objects so that the accumulators stay in registers
  CiElem<NL> e0, e1, e2;
  e0.r = e1.r = e2.r = -1;
  e0.c = e1.c = e2.c = 0;
  const bool diag_lane = t >= CI_D0;
  if(diag_lane)
    {
      // e0 is the diagonal element (t - CI_D0); the other two stay empty
      const int k = t - CI_D0;
      e0.acc = mw::acc_zero<NL>();
      if(k < n)
        {
          e0.r = e0.c = k;
          mw::acc_add(e0.acc, mat_ld<NL>(A, d, k0 + k, k0 + k));
        }
    }
  else
    {
      const int m = t / CI_GPR, g = t % CI_GPR;
      ci_init_offdiag<NL>(e0, m, CI_E * g, n, A, d, k0);
      ci_init_offdiag<NL>(e1, m, CI_E * g + 1, n, A, d, k0);
      ci_init_offdiag<NL>(e2, m, CI_E * g + 2, n, A, d, k0);
    }
  if(t == 0)
    s_fail = 0;
  __syncthreads();
  for(int k = 0; k < n; ++k)
    {
      // (0) the off-diagonal lanes pick the element they will finish at this pivot (at most one per lane lies in column
      //     k or row k) and normalise its sum now, beside the pivot's reciprocal square root
      bool m0 = false, m1 = false, m2 = false;
      Mw<NL> fsum = mw::zero<NL>();
      if(!diag_lane)
        {
          m0 = e0.r >= 0 && (e0.r == k || e0.c == k) && e0.r >= k;
          m1 = e1.r >= 0 && (e1.r == k || e1.c == k) && e1.r >= k;
          m2 = e2.r >= 0 && (e2.r == k || e2.c == k) && e2.r >= k;
          if(m0 || m1 || m2)
            fsum = mw::acc_result(m0 ? e0.acc : (m1 ? e1.acc : e2.acc));
        }
      // (1) pivot: X_kk = 1/L_kk = rsqrt(d_k)
      Mw<NL> dk = mw::zero<NL>();
#if defined(__HIP_DEVICE_COMPILE__)
      // The diagonal's wavefront computes the ONE reciprocal square root together (mw_wave.hpp: the same Newton ladder
      // with a limb per lane, bit-identical to mw::rsqrt; 3.0 instead of 4.8-6 us at 576 bits, 4.5 instead of 11-15 us
      // at 1088: profiles/r04i_ubench_wave_rsqrt.txt) — it is the longest link of the pivot chain.  Lane k of that
      // wavefront owns (k, k); the branch is uniform over the wavefront.
      if(diag_lane)
        {
          if(e0.r == k)
            dk = mw::acc_result(e0.acc);
          const int owner = k; // CI_D0 is a multiple of the wavefront size
          const uint32_t bad = mw::wv::bcast((mw::is_zero(dk) || dk.neg) ? 1u : 0u, owner);
          if(bad)
            {
              if(e0.r == k)
                s_fail = k0 + k + 1;
            }
          else
            {
              const Mw<NL> r = mw::wv::rsqrt<NL>(dk, owner);
              if(e0.r == k)
                {
#pragma unroll
                  for(int l = 0; l < NL; ++l)
                    s_inv[l] = r.m[l];
                  s_inv[NL] = (uint32_t)r.e;
                  s_inv[NL + 1] = r.neg;
                }
            }
        }
#else
      if(diag_lane && e0.r == k)
        {
          dk = mw::acc_result(e0.acc);
          if(mw::is_zero(dk) || dk.neg)
            s_fail = k0 + k + 1;
          else
            {
              const Mw<NL> r = mw::rsqrt(dk);
#pragma unroll
              for(int l = 0; l < NL; ++l)
                s_inv[l] = r.m[l];
              s_inv[NL] = (uint32_t)r.e;
              s_inv[NL + 1] = r.neg;
            }
        }
#endif
      __syncthreads();
      if(s_fail)
        {
          if(t == 0)
            fail[q] = s_fail;
          return;
        }
      // (2) column k of L, row k of X; the diagonal owner (off the critical path now) L_kk = d_k / L_kk
      if(diag_lane)
        {
          if(e0.r == k)
            {
              Mw<NL> inv;
#pragma unroll
              for(int l = 0; l < NL; ++l)
                inv.m[l] = s_inv[l];
              inv.e = (int32_t)s_inv[NL];
              inv.neg = s_inv[NL + 1];
              ci_st<NL>(sL, SDPB_PK(k, k), mw::mul(dk, inv));
              ci_st<NL>(sX, SDPB_PK(k, k), inv);
              mw::store<NL>(invd.p, (size_t)dv.off + k0 + k, inv);
            }
        }
      else
        {
          // at most one of the lane's elements lies in column k or row k: finish it once
          if(m0 || m1 || m2)
            {
              CiElem<NL> sel = m0 ? e0 : (m1 ? e1 : e2);
              ci_finish<NL>(sel, fsum, k, n, sL, sX, s_inv);
              if(m0)
                e0.acc = sel.acc;
              else if(m1)
                e1.acc = sel.acc;
              else
                e2.acc = sel.acc;
            }
        }
      __syncthreads();
      // (3) one product per element below row k
      if(diag_lane)
        {
          if(e0.r > k)
            {
              const Mw<NL> l = ci_ld<NL>(sL, SDPB_PK(e0.r, k));
              mw::acc_fma(e0.acc, l, l, 1u);
            }
        }
      else
        {
          ci_update<NL>(e0, k, n, sL, sX);
          ci_update<NL>(e1, k, n, sL, sX);
          ci_update<NL>(e2, k, n, sL, sX);
        }
    }
  __syncthreads();
  for(int idx = t; idx < n * n; idx += CI_T)
    {
      const int cc = idx / n, r = idx % n;
      mat_st<NL>(A, d, k0 + r, k0 + cc, r >= cc ? ci_ld<NL>(sL, SDPB_PK(r, cc)) : mw::zero<NL>());
      mat_st<NL>(Li, di, k0 + r, k0 + cc, r >= cc ? ci_ld<NL>(sX, SDPB_PK(r, cc)) : mw::zero<NL>());
    }
#undef SDPB_PK
}

// Tile helper shared by the panel kernels: a workgroup owns TR = 8 rows x PB columns;
// lane t -> (row t % 8, column t / 8), so consecutive lanes read consecutive rows.
constexpr int TR = WG / PB;
// 4 columns per chunk: 31 KB of LDS per workgroup = 5 workgroups per CU; measured on C4 (P = L^{-1}B stage):
// KC 16: 41.5 ms, 8: 37.5 ms, 4: 35.6 ms, 2: 35.9 ms
#ifndef SDPB_TRSM_KC
#define SDPB_TRSM_KC 4
#endif
constexpr int TRSM_KC = PB < SDPB_TRSM_KC ? PB : SDPB_TRSM_KC; // columns per LDS-staged operand chunk of the panel kernels
#ifndef SDPB_TRSM_ALIAS
#define SDPB_TRSM_ALIAS 1 // measured on C4 (profiles/r04t_trsm_variants.txt): P = L^-1 B 35.1 -> 34.1 ms, 34.8 -> 34.4 / 34.1 ms on another box
#endif
// ... of the two Cholesky panel kernels (k_chol_panel_solve, chol_syrk_tile): half the barriers per term
#ifndef SDPB_CHOL_KC
#define SDPB_CHOL_KC SDPB_TRSM_KC
#endif
constexpr int CHOL_KC = PB < SDPB_CHOL_KC ? PB : SDPB_CHOL_KC;

// Cholesky panel solve: A(r, panel p) := A(r, panel p) * Li_pp^T for the rows below the
// diagonal block, and zero the part of the panel above it.   grid = (row tiles, batch)
// row tiles [tile0, tile0 + gridDim.x) (the look-ahead schedule of Cholesky(Q) splits them).
// The rows of the tile and chunks of KC columns of Li are staged in limb-major LDS.
template <int NL> __global__ void __launch_bounds__(WG) k_chol_panel_solve(Batch A, Batch Li, int p, int tile0, unsigned long long *cyc)
{
  constexpr int KC = CHOL_KC, SLN = PB * KC, STN = TR * PB;
  const int q = blockIdx.y;
  WgClock clk(cyc, q);
  const MatDesc d = A.d[q], di = Li.d[q];
  const int k0 = PB * p;
  if(k0 >= d.rows)
    return;
  const int nb = d.rows - k0 < PB ? d.rows - k0 : PB;
  const int rl = threadIdx.x % TR, j = threadIdx.x / TR;
  const int bx = (int)blockIdx.x + tile0;
  const int r = k0 + nb + bx * TR + rl;
  __shared__ uint32_t sl[(NL + 2) * SLN], st[(NL + 2) * STN];
  const bool ok = r < d.rows && j < nb;
  if(bx * TR >= d.rows - k0 - nb && bx * TR >= k0)
    return;
  smem_st<NL, STN>(st, j * TR + rl, ok ? mat_ld<NL>(A, d, r, k0 + j) : mw::zero<NL>());
  // rows above the diagonal block: upper triangle is zero in a lower factor
  const int ru = bx * TR + rl;
  if(ru < k0 && j < nb)
    mat_st<NL>(A, d, ru, k0 + j, mw::zero<NL>());
  if(bx * TR >= d.rows - k0 - nb) // nothing below the diagonal block in this tile (uniform)
    return;
  Acc<NL> acc = mw::acc_zero<NL>();
  for(int c = 0; c < nb; c += KC)
    {
      for(int f = threadIdx.x; f < SLN; f += WG)
        {
          const int kk = f / PB, jj = f % PB;
          smem_st<NL, SLN>(sl, f, (jj < nb && c + kk <= jj) ? mat_ld<NL>(Li, di, k0 + jj, k0 + c + kk) : mw::zero<NL>());
        }
      __syncthreads();
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            if(c + kk <= j)
              mw::acc_fma(acc, smem_ld<NL, STN>(st, (c + kk) * TR + rl), smem_ld<NL, SLN>(sl, kk * PB + j));
        }
      __syncthreads();
    }
  if(ok)
    mat_st<NL>(A, d, r, k0 + j, mw::acc_result(acc));
}

// Trailing update of the blocked Cholesky: A22(i,j) -= sum_k A21(i,k) A21(j,k), i >= j,
// with A21 = rows below panel p, columns of panel p.   grid = (lower tiles, batch)
// lower tiles [tile0, tile0 + gridDim.x) of the trailing matrix; the two 16 x KC operand
// chunks of a tile are staged in limb-major LDS
template <int NL>
__device__ void chol_syrk_tile(const Batch &A, const MatDesc &d, int k0, int nb, int b0, int M, int ti, int tj, int cmin = 0, int cmax = 1 << 30);
// [col_lo, col_hi): only the columns of the matrix inside this window are updated (the chased Cholesky(Q) applies a
// panel to the columns that exist so far and to the others when they arrive; 0, INT_MAX = all)
template <int NL> __global__ void __launch_bounds__(WG) k_chol_syrk_down(Batch A, int p, int tile0, unsigned long long *cyc, int col_lo, int col_hi)
{
  const int q = blockIdx.y;
  WgClock clk(cyc, q);
  const MatDesc d = A.d[q];
  const int k0 = PB * p;
  if(k0 >= d.rows)
    return;
  const int nb = d.rows - k0 < PB ? d.rows - k0 : PB;
  const int b0 = k0 + nb, M = d.rows - b0;
  const int tiles = (M + 15) / 16;
  int tile = (int)blockIdx.x + tile0;
  if(M <= 0 || tile >= tiles * (tiles + 1) / 2)
    return;
  int ti = 0;
  while((ti + 1) * (ti + 2) / 2 <= tile)
    ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  // the window in the trailing matrix's own column numbers; a tile that straddles an end is masked per element
  const int c_lo = col_lo > b0 ? col_lo - b0 : 0, c_hi = col_hi - b0 < M ? col_hi - b0 : M;
  if(16 * tj + 16 <= c_lo || 16 * tj >= c_hi)
    return;
  chol_syrk_tile<NL>(A, d, k0, nb, b0, M, ti, tj, c_lo, c_hi);
}
// The same update restricted to the column panels q = q0, q0 + qstride, ... (count of them) of
// the single matrix A.d[0]: the 1-D block-cyclic Cholesky(Q) over ranks updates only the panels a
// rank owns.  grid = (row tiles of the trailing matrix, 2 tile columns per owned panel).
template <int NL> __global__ void __launch_bounds__(WG) k_chol_syrk_cols(Batch A, int p, int q0, int qstride, int count)
{
  static_assert(PB % 16 == 0 || PB < 16, "a panel is a whole number of 16-column tiles (or a single one)");
  constexpr int TPP = PB >= 16 ? PB / 16 : 1; // tile columns per panel
  const MatDesc d = A.d[0];
  const int k0 = PB * p;
  const int nb = d.rows - k0 < PB ? d.rows - k0 : PB;
  const int b0 = k0 + nb, M = d.rows - b0;
  const int which = (int)blockIdx.y / TPP, sub = (int)blockIdx.y % TPP;
  if(M <= 0 || which >= count)
    return;
  const int q = q0 + which * qstride;            // owned panel, q > p
  const int col0 = PB * q - b0;                   // first column of the panel inside the trailing matrix
  if(col0 < 0 || col0 >= M)
    return;
  if constexpr(PB >= 16)
    {
      const int tj = col0 / 16 + sub, ti = (int)blockIdx.x;
      if(tj * 16 >= M || ti < tj || ti * 16 >= M)
        return;
      chol_syrk_tile<NL>(A, d, k0, nb, b0, M, ti, tj);
    }
  else
    {
      // panels narrower than a tile (test builds with SDPB_PB = 4): the tile column that holds the panel;
      // entries of neighbouring panels inside the tile are skipped by the column mask below
      const int tj = col0 / 16, ti = (int)blockIdx.x;
      if(ti < tj || ti * 16 >= M)
        return;
      chol_syrk_tile<NL>(A, d, k0, nb, b0, M, ti, tj, col0, col0 + PB);
    }
}
template <int NL> __device__ void chol_syrk_tile(const Batch &A, const MatDesc &d, int k0, int nb, int b0, int M, int ti, int tj, int cmin, int cmax)
{
  constexpr int KC = CHOL_KC, SN = 16 * KC;
  const int li = threadIdx.x & 15, lj = threadIdx.x >> 4;
  const int i = ti * 16 + li, j = tj * 16 + lj;
  const bool ok = i < M && j <= i && j >= cmin && j < cmax;
  __shared__ uint32_t sa[(NL + 2) * SN], sb[(NL + 2) * SN];
  Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    mw::acc_add(acc, mat_ld<NL>(A, d, b0 + i, b0 + j));
  for(int c = 0; c < nb; c += KC)
    {
      for(int e = threadIdx.x; e < 2 * SN; e += WG)
        {
          const int f = e % SN, kk = f / 16, rr = f % 16;
          const int row = (e < SN ? ti : tj) * 16 + rr;
          smem_st<NL, SN>(e < SN ? sa : sb, f, (row < M && c + kk < nb) ? mat_ld<NL>(A, d, b0 + row, k0 + c + kk) : mw::zero<NL>());
        }
      __syncthreads();
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            mw::acc_fms(acc, smem_ld<NL, SN>(sa, kk * 16 + li), smem_ld<NL, SN>(sb, kk * 16 + lj));
        }
      __syncthreads();
    }
  if(ok)
    mat_st<NL>(A, d, b0 + i, b0 + j, mw::acc_result(acc));
}

// Panel message of the distributed Cholesky(Q) (1-D block-cyclic over ranks, one broadcast per
// panel): the finished column panel p of the factor (rows k0 .. n-1, nb columns), the inverted diagonal
// block Li_pp (nb x nb) and 1/L_ii (nb), limb-major with stride cnt = (n - k0) nb + nb nb + nb, then one
// word: the owner's failure flag.  Unpacking also zeroes the panel's rows above the diagonal block
// (the factor is lower triangular) and raises the local flag.
template <int NL>
__global__ void __launch_bounds__(WG) k_qpanel_pack(Batch A, Batch Li, Batch invd, int p, const int *fail, uint32_t *msg)
{
  const MatDesc d = A.d[0], di = Li.d[0], dv = invd.d[0];
  const int k0 = PB * p, n = d.rows, nb = n - k0 < PB ? n - k0 : PB, rows = n - k0;
  const size_t cnt = (size_t)rows * nb + (size_t)nb * nb + nb;
  const size_t e = (size_t)blockIdx.x * WG + threadIdx.x;
  if(e == 0)
    msg[cnt * (NL + 1)] = (uint32_t)fail[0];
  if(e >= cnt)
    return;
  Mw<NL> v;
  if(e < (size_t)rows * nb)
    v = mat_ld<NL>(A, d, k0 + (int)(e % rows), k0 + (int)(e / rows));
  else if(e < (size_t)rows * nb + (size_t)nb * nb)
    {
      const size_t f = e - (size_t)rows * nb;
      v = mat_ld<NL>(Li, di, k0 + (int)(f % nb), k0 + (int)(f / nb));
    }
  else
    v = mat_ld<NL>(invd, dv, k0 + (int)(e - (size_t)rows * nb - (size_t)nb * nb), 0);
  mw::store<NL>(mw::Ptr{msg, cnt}, e, v);
}
template <int NL>
__global__ void __launch_bounds__(WG) k_qpanel_unpack(Batch A, Batch Li, Batch invd, int p, int *fail, const uint32_t *msg)
{
  const MatDesc d = A.d[0], di = Li.d[0], dv = invd.d[0];
  const int k0 = PB * p, n = d.rows, nb = n - k0 < PB ? n - k0 : PB, rows = n - k0;
  const size_t cnt = (size_t)rows * nb + (size_t)nb * nb + nb;
  const size_t e = (size_t)blockIdx.x * WG + threadIdx.x;
  if(e == 0 && msg[cnt * (NL + 1)])
    atomicMax(fail, (int)msg[cnt * (NL + 1)]);
  if(e < (size_t)k0 * nb) // rows above the diagonal block
    mat_st<NL>(A, d, (int)(e % k0), k0 + (int)(e / k0), mw::zero<NL>());
  if(e >= cnt)
    return;
  const Mw<NL> v = mw::load<NL>(mw::CPtr{msg, cnt}, e);
  if(e < (size_t)rows * nb)
    mat_st<NL>(A, d, k0 + (int)(e % rows), k0 + (int)(e / rows), v);
  else if(e < (size_t)rows * nb + (size_t)nb * nb)
    {
      const size_t f = e - (size_t)rows * nb;
      mat_st<NL>(Li, di, k0 + (int)(f % nb), k0 + (int)(f / nb), v);
    }
  else
    mat_st<NL>(invd, dv, k0 + (int)(e - (size_t)rows * nb - (size_t)nb * nb), 0, v);
}

// The two strip steps of the look-ahead Cholesky(Q) sit between consecutive diagonal blocks on the
// critical path of the iteration, and each of their outputs is a PB-term dot product: eight lanes
// per output (four terms each, partial sums through LDS) instead of one lane running PB dependent
// products — 62 + 47 us -> ~2 x 20 us per panel.  Single matrix (A.d[0]).
// (a) the PB rows below diagonal block p:  A(r, panel p) := A(r, panel p) Li_pp^T; one workgroup per row
//     (it also zeroes row blockIdx.x of the panel above the diagonal block, as k_chol_panel_solve does)
template <int NL> __global__ void __launch_bounds__(WG) k_chol_strip_solve(Batch A, Batch Li, int p)
{
  raise_chain_priority();
  static_assert(WG % PB == 0 || PB > WG, "lanes per output");
  constexpr int SEG = PB <= WG ? WG / PB : 1;
  const MatDesc d = A.d[0], di = Li.d[0];
  const int k0 = PB * p, t = threadIdx.x;
  if(k0 >= d.rows)
    return;
  const int nb = d.rows - k0 < PB ? d.rows - k0 : PB;
  const int j = t / SEG, s = t % SEG;
  const int r = k0 + nb + (int)blockIdx.x;
  const bool ok = r < d.rows && j < nb;
  __shared__ Mw<NL> part[WG];
  Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    for(int k = s; k <= j; k += SEG)
      mw::acc_fma(acc, mat_ld<NL>(A, d, r, k0 + k), mat_ld<NL>(Li, di, k0 + j, k0 + k));
  part[t] = mw::acc_result(acc);
  __syncthreads(); // every read of row r precedes the writes below
  if(ok && s == 0)
    {
      Acc<NL> sum = mw::acc_zero<NL>();
      for(int g = 0; g < SEG; ++g)
        mw::acc_add(sum, part[t + g]);
      mat_st<NL>(A, d, r, k0 + j, mw::acc_result(sum));
    }
  const int ru = (int)blockIdx.x;
  if(ru < k0 && t < nb)
    mat_st<NL>(A, d, ru, k0 + t, mw::zero<NL>());
}
// (b) the leading PB x PB block of the trailing update: A(i,j) -= sum_k A(i, k0+k) A(j, k0+k), i >= j;
//     WG / SEG outputs per workgroup in packed lower-triangle order
template <int NL> __global__ void __launch_bounds__(WG) k_chol_strip_update(Batch A, int p)
{
  raise_chain_priority();
  constexpr int SEG = PB <= WG ? WG / PB : 1, OUT = WG / SEG;
  const MatDesc d = A.d[0];
  const int k0 = PB * p, t = threadIdx.x;
  if(k0 >= d.rows)
    return;
  const int nb = d.rows - k0 < PB ? d.rows - k0 : PB;
  const int b0 = k0 + nb, M = d.rows - b0 < PB ? d.rows - b0 : PB;
  const int o = (int)blockIdx.x * OUT + t / SEG, s = t % SEG;
  int i = 0;
  while((i + 1) * (i + 2) / 2 <= o)
    ++i;
  const int j = o - i * (i + 1) / 2;
  const bool ok = i < M;
  __shared__ Mw<NL> part[WG];
  Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    for(int k = s; k < nb; k += SEG)
      mw::acc_fms(acc, mat_ld<NL>(A, d, b0 + i, k0 + k), mat_ld<NL>(A, d, b0 + j, k0 + k));
  part[t] = mw::acc_result(acc);
  __syncthreads();
  if(ok && s == 0)
    {
      Acc<NL> sum = mw::acc_zero<NL>();
      mw::acc_add(sum, mat_ld<NL>(A, d, b0 + i, b0 + j));
      for(int g = 0; g < SEG; ++g)
        mw::acc_add(sum, part[t + g]);
      mat_st<NL>(A, d, b0 + i, b0 + j, mw::acc_result(sum));
    }
}

// X := X L^{-T}, panel p (forward over panels):
//   T = X(:,panel p) - X(:,cols < k0) L(panel p, cols < k0)^T ;  X(:,panel p) = T Li_pp^T
// One lane per (row, column of the panel); rows are the independent right-hand sides
// (with B stored transposed this is schur_off_diagonal = L^{-1} B, compute_Q.cxx:48).
// Both dot products run over chunks of KC columns whose operands — TR x KC numbers of X
// (or of T) and PB x KC numbers of L (or of Li) — are fetched once per workgroup into
// limb-major LDS tiles, so the inner loop is LDS reads + one Acc product per term.
// One workgroup tile of COLS panel columns x WG/COLS rows (COLS = the power of two that
// covers the panel's width: a ragged last panel of 8 columns is worked on by 32 rows x 8
// columns instead of 8 rows x 32 columns with three quarters of the lanes idle).
template <int NL, int COLS>
MW_HD void trsm_rlt_tile(const Batch &L, const Batch &Li, const Batch &X, const mw::CPtr &src, const MatDesc &dl, const MatDesc &di,
                         const MatDesc &dx, int k0, int nb, uint32_t *smem)
{
  // src: where the right-hand sides of THIS panel's columns are read (the solved columns k < k0 always come from X): X
  // itself for the in-place solve, or another array with X's descriptors -- P = L^{-1} B reads B and writes P, which
  // saves the 3-GB device-to-device copy per iteration the in-place form needed in front of it (round 5)
  constexpr int KC = TRSM_KC, ROWS = WG / COLS, SXN = ROWS * KC, SLN = COLS * KC, STN = ROWS * COLS;
#if SDPB_TRSM_ALIAS
  // the T tile of the second phase lies over the X chunk of the first (whose last pass ends with a barrier): 384
  // instead of 416 numbers of LDS per workgroup = five workgroups per CU instead of four
  static_assert(SXN <= STN, "the X chunk fits under the T tile");
  uint32_t *st = smem, *sx = smem, *sl = smem + (NL + 2) * STN;
#else
  uint32_t *sx = smem, *sl = sx + (NL + 2) * SXN, *st = sl + (NL + 2) * SLN;
#endif
  if((int)(blockIdx.x * ROWS) >= dx.rows)
    return;
  const int rl = threadIdx.x % ROWS, j = threadIdx.x / ROWS;
  const int r0 = blockIdx.x * ROWS, r = r0 + rl;
  const bool ok = r < dx.rows && j < nb;
  Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    mw::acc_add(acc, mw::load<NL>(src, (size_t)dx.off + (size_t)r + (size_t)(k0 + j) * dx.ld));
  for(int k = 0; k < k0; k += KC) // k0 is a multiple of PB, PB of KC
    {
      for(int e = threadIdx.x; e < SXN + SLN; e += WG)
        {
          if(e < SXN)
            {
              const int kk = e / ROWS, rr = e % ROWS;
              smem_st<NL, SXN>(sx, e, r0 + rr < dx.rows ? mat_ld<NL>(X, dx, r0 + rr, k + kk) : mw::zero<NL>());
            }
          else
            {
              const int f = e - SXN, kk = f / COLS, jj = f % COLS;
              smem_st<NL, SLN>(sl, f, jj < nb ? mat_ld<NL>(L, dl, k0 + jj, k + kk) : mw::zero<NL>());
            }
        }
      __syncthreads();
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            mw::acc_fms(acc, smem_ld<NL, SXN>(sx, kk * ROWS + rl), smem_ld<NL, SLN>(sl, kk * COLS + j));
        }
      __syncthreads();
    }
  smem_st<NL, STN>(st, j * ROWS + rl, ok ? mw::acc_result(acc) : mw::zero<NL>());
  acc = mw::acc_zero<NL>();
  // X(r, k0+j) = sum_{j2 <= j} T(r, j2) Li(k0+j, k0+j2)
  for(int c = 0; c < nb; c += KC)
    {
      for(int f = threadIdx.x; f < SLN; f += WG)
        {
          const int kk = f / COLS, jj = f % COLS;
          smem_st<NL, SLN>(sl, f, (jj < nb && c + kk <= jj) ? mat_ld<NL>(Li, di, k0 + jj, k0 + c + kk) : mw::zero<NL>());
        }
      __syncthreads(); // also orders the writes of st before the first read
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            if(c + kk <= j)
              mw::acc_fma(acc, smem_ld<NL, STN>(st, (c + kk) * ROWS + rl), smem_ld<NL, SLN>(sl, kk * COLS + j));
        }
      __syncthreads();
    }
  if(ok)
    mat_st<NL>(X, dx, r, k0 + j, mw::acc_result(acc));
}
template <int NL> __global__ void __launch_bounds__(WG) k_trsm_rlt_panel(Batch L, Batch Li, Batch X, mw::CPtr src, int p, unsigned long long *cyc)
{
  constexpr int KC = TRSM_KC;
  const int q = blockIdx.y;
  WgClock clk(cyc, q);
  const MatDesc dl = L.d[q], di = Li.d[q], dx = X.d[q];
  const int k0 = PB * p;
  if(k0 >= dl.rows)
    return;
  const int nb = dl.rows - k0 < PB ? dl.rows - k0 : PB;
  // X chunk + L chunk + T tile: (WG/COLS + COLS) KC + WG numbers, largest at COLS = PB and COLS = 8
  constexpr int MINC = PB < 8 ? PB : 8, NMAX = (WG / MINC + MINC) * KC > (WG / PB + PB) * KC ? (WG / MINC + MINC) * KC : (WG / PB + PB) * KC;
#if SDPB_TRSM_ALIAS
  __shared__ uint32_t smem[(NL + 2) * (WG + PB * KC)];
#else
  __shared__ uint32_t smem[(NL + 2) * (NMAX + WG)];
#endif
  if constexpr(PB >= 32)
    {
      if(nb <= 8)
        return trsm_rlt_tile<NL, 8>(L, Li, X, src, dl, di, dx, k0, nb, smem);
      if(nb <= 16)
        return trsm_rlt_tile<NL, 16>(L, Li, X, src, dl, di, dx, k0, nb, smem);
    }
  trsm_rlt_tile<NL, PB>(L, Li, X, src, dl, di, dx, k0, nb, smem);
}
// ---- P = L^{-1} B as fixed-point tile dot products (tiledot.hpp; round 6) ---------------------------------------------
// Panel p of the solve is  T = X_p - sum_{kt < p} X_kt L(p, kt)^T  followed by  X_p = T Li_pp^T: p + 1 dot products over
// tiles of PB terms per output.  k_td_image_L rewrites every tile of every block -- the tiles L(panel p, tile kt) below the
// diagonal and the inverted diagonal blocks Li_pp -- ONCE per factorisation as biased fixed-point images (+ row exponents
// and limb sums); k_trsm_rlt_panel_td rewrites its own 8 rows of the X tile (and of T) on the fly, one entry per lane, and a
// term is then W (W+1)/2 + W - 1 multiply-adds (298 at 18 limbs, one v_mad_u64_u32 each at 4.2 cycles:
// profiles/r06d_ubench_mad_only.txt) against 188 pairs + ~250 instructions for the float term.  Each tile's sum enters the
// float accumulator as one term.  Panels of <= 16 columns keep the float path (trsm_rlt_tile).
template <int NL> constexpr bool td_trsm_enabled()
{
#if defined(SDPB_NO_TILEDOT)
  return false;
#else
  return PB == 32 && NL <= 18 && td::fits<td::limbs<NL>(), PB>(); // (24 limbs fit the columns but not 128 VGPRs: not tuned)
#endif
}
// tiles of a block with `rows` rows: (p, kt), 0 <= kt < p < panels, numbered p (p - 1) / 2 + kt, then the diagonal tiles
MW_HD int td_offdiag_tiles(int rows) { const int np = (rows + PB - 1) / PB; return np * (np - 1) / 2; }
MW_HD int td_tiles_of(int rows) { const int np = (rows + PB - 1) / PB; return np * (np + 1) / 2; }
constexpr int TD_JW = 64 / (WG / PB); // panel columns per wavefront of k_trsm_rlt_panel_td: a wavefront's lanes share the trip count of the triangular phase
// image layout: tile-major, then [k][column of the panel][W words]; exponents [tile][column]; limb sums [tile][column][W].
// Diagonal tiles hold Li(c, k) for k <= c and (biased) zeros above; their limb sums run over k < TD_JW (c / TD_JW + 1), the
// terms the wavefront of column c executes.
template <int NL>
__global__ void __launch_bounds__(WG) k_td_image_L(Batch L, Batch Li, const int *tile_off, uint32_t *img, int32_t *exps, uint32_t *sums)
{
  constexpr int W = td::limbs<NL>(), KQ = WG / PB < PB ? WG / PB : PB, KPL = PB / KQ; // KQ lanes share a column, KPL entries each
  static_assert(PB % KQ == 0 || !td_trsm_enabled<NL>(), "the k range of a tile is dealt to the lanes of a column");
  const int q = blockIdx.y, t = blockIdx.x;
  const MatDesc d = L.d[q], di = Li.d[q];
  if(t >= td_tiles_of(d.rows))
    return;
  const int noff = td_offdiag_tiles(d.rows);
  const bool diag = t >= noff;
  int p = 1, kt = 0;
  if(diag)
    p = kt = t - noff;
  else
    {
      while((p + 1) * p / 2 <= t)
        ++p;
      kt = t - p * (p - 1) / 2;
    }
  const int row0 = PB * p, k0 = PB * kt;
  const int c = threadIdx.x % PB, kq = threadIdx.x / PB;
  const bool ok = row0 + c < d.rows;
  const int limit = diag ? TD_JW * (c / TD_JW + 1) : PB; // terms whose images enter the limb sum of column c
  const size_t tile = (size_t)tile_off[q] + t;
  __shared__ int32_t sE[KQ][PB];
  __shared__ uint32_t sS[KQ][PB][W];
  Mw<NL> v[KPL];
  int32_t e = mw::EZERO;
#pragma unroll
  for(int u = 0; u < KPL; ++u)
    {
      const int k = kq + KQ * u;
      v[u] = mw::zero<NL>();
      if(ok && (!diag || (k <= c && k0 + k < d.rows)))
        v[u] = diag ? mat_ld<NL>(Li, di, row0 + c, k0 + k) : mat_ld<NL>(L, d, row0 + c, k0 + k);
      e = v[u].e > e ? v[u].e : e;
    }
  sE[kq][c] = e;
  __syncthreads();
#pragma unroll
  for(int u = 0; u < KQ; ++u)
    e = sE[u][c] > e ? sE[u][c] : e;
  uint32_t s[W];
#pragma unroll
  for(int i = 0; i < W; ++i)
    s[i] = 0;
#pragma unroll
  for(int u = 0; u < KPL; ++u)
    {
      const int k = kq + KQ * u;
      uint32_t b[W];
      td::to_image<NL, W>(v[u], e == mw::EZERO ? 0 : e, b);
      uint32_t *o = img + ((tile * PB + (size_t)k) * PB + c) * W;
#pragma unroll
      for(int i = 0; i < W; ++i)
        o[i] = b[i];
      if(k < limit)
        {
#pragma unroll
          for(int i = 0; i < W; ++i)
            s[i] += b[i];
        }
    }
#pragma unroll
  for(int i = 0; i < W; ++i)
    sS[kq][c][i] = s[i];
  __syncthreads();
  if(kq == 0)
    {
#pragma unroll 1
      for(int u = 1; u < KQ; ++u)
#pragma unroll
        for(int i = 0; i < W; ++i)
          s[i] += sS[u][c][i];
      uint32_t *o = sums + (tile * PB + c) * W;
#pragma unroll
      for(int i = 0; i < W; ++i)
        o[i] = s[i];
      exps[tile * PB + c] = e;
    }
}

#ifndef SDPB_TD_KC
#define SDPB_TD_KC 4 // k-slices of the right operand's image per LDS pass of the tile dot products
#endif
#ifndef SDPB_TD_PREFETCH
#define SDPB_TD_PREFETCH 0 // 1: the next slice of the right operand requested before the products of this one.  Measured and
                           // left off: Q.solve 29.6 against 27.5 ms on C4 (three more 16-byte registers per lane spill at
                           // the 128 VGPRs that four workgroups per CU allow; profiles/r06r_variants.txt)
#endif
#ifndef SDPB_TD_TRSM_WG_PER_CU
#define SDPB_TD_TRSM_WG_PER_CU 4 // measured on C4 (profiles/r06b_variants.txt): 2 / 3 / 4 workgroups per CU (208 / 168 / 128 VGPRs) 36.9 / 32.6 / 31.2 ms
#endif
// One tile of one output per lane: the ROWS x PB image of the left operand is in sX ([k][row][W]), the KC-slices of the
// right operand's image are streamed from `tile_img` through sL; `terms` (wavefront-uniform) of them are multiplied.
template <int NL, int W, int ROWS, int KC>
MW_HD void td_tile_product(td::Cols<W> &g, const uint32_t *sX, uint32_t *sL, const uint32_t *tile_img, int rl, int j, bool ok, int terms)
{
  constexpr int SL_WORDS = KC * PB * W;
  static_assert(SL_WORDS % 4 == 0, "16-byte moves of a slice");
  td::cols_zero<W>(g);
#if SDPB_TD_PREFETCH
  // the slice of the NEXT pass is requested before the products of this one and written to LDS when they are done: the
  // round trip to L2 (every workgroup of a block streams the same tile image) is no longer exposed at every barrier
  constexpr int NQ = (SL_WORDS / 4 + WG - 1) / WG;
  td::Quad pre[NQ];
  auto request = [&](int c) {
    const td::Quad *gsrc = reinterpret_cast<const td::Quad *>(tile_img + (size_t)c * PB * W);
#pragma unroll
    for(int u = 0; u < NQ; ++u)
      {
        const int f = threadIdx.x + u * WG;
        if(f < SL_WORDS / 4)
          pre[u] = gsrc[f];
      }
  };
  request(0);
#endif
  for(int c = 0; c < PB; c += KC)
    {
#if SDPB_TD_PREFETCH
      {
        td::Quad *dst = reinterpret_cast<td::Quad *>(sL);
#pragma unroll
        for(int u = 0; u < NQ; ++u)
          {
            const int f = threadIdx.x + u * WG;
            if(f < SL_WORDS / 4)
              dst[f] = pre[u];
          }
      }
      __syncthreads();
      if(c + KC < PB) // (workgroup-uniform: `terms` differs between the wavefronts of a diagonal tile, the slice is everybody's)
        request(c + KC);
#else
      {
        const td::Quad *gsrc = reinterpret_cast<const td::Quad *>(tile_img + (size_t)c * PB * W);
        td::Quad *dst = reinterpret_cast<td::Quad *>(sL);
        for(int f = threadIdx.x; f < SL_WORDS / 4; f += WG)
          dst[f] = gsrc[f];
      }
      __syncthreads();
#endif
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            if(c + kk < terms)
              {
                uint32_t xb[W], lb[W];
                const uint32_t *px = sX + ((c + kk) * ROWS + rl) * W, *pl = sL + (kk * PB + j) * W;
#pragma unroll
                for(int i = 0; i < W; ++i)
                  {
                    xb[i] = px[i];
                    lb[i] = pl[i];
                  }
                td::mac<W>(g, xb, lb);
              }
        }
      __syncthreads();
    }
}
// The workgroup's ROWS x PB tile of numbers -- lane (rl, j) holds entry (rl, j) -- as a fixed-point image in sX, the row
// exponent in F, the limb sums of the first TD_JW, 2 TD_JW, ... entries of every row in sS ([prefix][row][W]).
template <int NL, int W, int ROWS>
MW_HD void td_stage_rows(const Mw<NL> &xv, int rl, int j, uint32_t *sX, int32_t *sExp, uint32_t *sS, int32_t &F)
{
  sExp[j * ROWS + rl] = xv.e;
  __syncthreads(); // (also: the previous tile's reads of sX and sS are over)
  F = mw::EZERO;
#pragma unroll 8
  for(int u = 0; u < PB; ++u)
    {
      const int32_t e = sExp[u * ROWS + rl];
      F = e > F ? e : F;
    }
  {
    uint32_t b[W];
    td::to_image<NL, W>(xv, F == mw::EZERO ? 0 : F, b);
    uint32_t *o = sX + (j * ROWS + rl) * W;
#pragma unroll
    for(int i = 0; i < W; ++i)
      o[i] = b[i];
  }
  __syncthreads();
  if(threadIdx.x < ROWS * W) // limb sums of the image rows (not carried: 32 x 2^27 < 2^32)
    {
      const int rr = threadIdx.x % ROWS, i = threadIdx.x / ROWS;
      uint32_t s = 0;
#pragma unroll 8
      for(int u = 0; u < PB; ++u)
        {
          s += sX[(u * ROWS + rr) * W + i];
          if((u + 1) % TD_JW == 0)
            sS[((u / TD_JW) * ROWS + rr) * W + i] = s;
        }
    }
}
template <int NL>
__global__ void __launch_bounds__(WG, SDPB_TD_TRSM_WG_PER_CU)
  k_trsm_rlt_panel_td(Batch L, Batch Li, Batch X, mw::CPtr src, int p, unsigned long long *cyc, const int *tile_off, const uint32_t *img,
                      const int32_t *exps, const uint32_t *sums)
{
  constexpr int KC = SDPB_TD_KC, W = td::limbs<NL>(), ROWS = WG / PB, NPRE = PB / TD_JW;
  const int q = blockIdx.y;
  WgClock clk(cyc, q);
  const MatDesc dl = L.d[q], di = Li.d[q], dx = X.d[q];
  const int k0 = PB * p;
  if(k0 >= dl.rows)
    return;
  const int nb = dl.rows - k0 < PB ? dl.rows - k0 : PB;
  // LDS: the float path's tiles (narrow panels) and, over the same bytes, the fixed-point tiles
  constexpr int FLOAT_WORDS = (NL + 2) * (WG + PB * TRSM_KC);
  constexpr int SX_WORDS = PB * ROWS * W, SL_WORDS = KC * PB * W;
  constexpr int TD_WORDS = SX_WORDS + SL_WORDS + PB * ROWS + NPRE * ROWS * W;
  __shared__ __attribute__((aligned(16))) uint32_t smem[FLOAT_WORDS > TD_WORDS ? FLOAT_WORDS : TD_WORDS];
  if constexpr(PB >= 32)
    {
      if(nb <= 8)
        return trsm_rlt_tile<NL, 8>(L, Li, X, src, dl, di, dx, k0, nb, smem);
      if(nb <= 16)
        return trsm_rlt_tile<NL, 16>(L, Li, X, src, dl, di, dx, k0, nb, smem);
    }
  if((int)(blockIdx.x * ROWS) >= dx.rows)
    return;
  uint32_t *sL = smem, *sX = smem + SL_WORDS;          // (sL first: its 16-byte moves want the aligned base)
  int32_t *sExp = (int32_t *)(sX + SX_WORDS);           // [k][row]
  uint32_t *sS = (uint32_t *)(sExp + PB * ROWS);        // [prefix][row][W]
  const int rl = threadIdx.x % ROWS, j = threadIdx.x / ROWS;
  const int r0 = blockIdx.x * ROWS, r = r0 + rl;
  const bool ok = r < dx.rows && j < nb;
  mw::Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    mw::acc_add(acc, mw::load<NL>(src, (size_t)dx.off + (size_t)r + (size_t)(k0 + j) * dx.ld));
  const size_t tbase = (size_t)tile_off[q], tile0 = tbase + (size_t)p * (p - 1) / 2;
  for(int kt = 0; kt < p; ++kt)
    {
      // T -= X(rows, tile kt) L(panel p, tile kt)^T
      const Mw<NL> xv = r < dx.rows ? mat_ld<NL>(X, dx, r, PB * kt + j) : mw::zero<NL>();
      int32_t F;
      td_stage_rows<NL, W, ROWS>(xv, rl, j, sX, sExp, sS, F);
      td::Cols<W> g;
      td_tile_product<NL, W, ROWS, KC>(g, sX, sL, img + (tile0 + kt) * (size_t)PB * PB * W, rl, j, ok, PB);
      if(ok)
        {
          uint32_t sx[W], sl[W];
          const uint32_t *ps = sums + ((tile0 + kt) * PB + j) * W;
#pragma unroll
          for(int i = 0; i < W; ++i)
            {
              sx[i] = sS[((NPRE - 1) * ROWS + rl) * W + i];
              sl[i] = ps[i];
            }
          td::acc_add_tile<NL, W>(acc, g, sx, sl, (uint32_t)PB, F, exps[(tile0 + kt) * PB + j], 1u);
        }
    }
  // X(r, k0 + j) = sum_{j2 <= j} T(r, j2) Li(k0 + j, k0 + j2): the diagonal tile; the wavefront of columns [TD_JW w, TD_JW (w+1))
  // multiplies the first TD_JW (w + 1) terms (the entries above the diagonal are zeros of the image: they cancel exactly)
  {
    const Mw<NL> tv = ok ? mw::acc_result(acc) : mw::zero<NL>();
    const int pre = j / TD_JW, terms = TD_JW * (pre + 1);
    const size_t dtile = tbase + td_offdiag_tiles(dl.rows) + p;
    int32_t F;
    td_stage_rows<NL, W, ROWS>(tv, rl, j, sX, sExp, sS, F);
    td::Cols<W> g;
    td_tile_product<NL, W, ROWS, KC>(g, sX, sL, img + dtile * (size_t)PB * PB * W, rl, j, ok, terms);
    if(ok)
      {
        uint32_t sx[W], sl[W];
        const uint32_t *ps = sums + (dtile * PB + j) * W;
#pragma unroll
        for(int i = 0; i < W; ++i)
          {
            sx[i] = sS[(pre * ROWS + rl) * W + i];
            sl[i] = ps[i];
          }
        acc = mw::acc_zero<NL>();
        td::acc_add_tile<NL, W>(acc, g, sx, sl, (uint32_t)terms, F, exps[dtile * PB + j], 0u);
        mat_st<NL>(X, dx, r, k0 + j, mw::acc_result(acc));
      }
  }
}

// X := X L^{-1}, panel p (backward over panels):
//   T = X(:,panel p) - X(:,cols >= k0+nb) L(rows >= k0+nb, panel p) ;  X(:,panel p) = T Li_pp
// The mirror image of trsm_rlt_tile: the same workgroup tile, the same limb-major LDS staging of
// KC-column chunks (an array-of-numbers LDS tile and per-lane loads of Li down a column cost this
// kernel 1.0 ms per launch on the PSD blocks of C4 against 0.5 ms for its twin), and the same order
// of the terms as before (k ascending, then j2 ascending), so the results keep their bits.
template <int NL, int COLS>
MW_HD void trsm_rln_tile(const Batch &L, const Batch &Li, const Batch &X, const MatDesc &dl, const MatDesc &di, const MatDesc &dx, int k0, int nb,
                         uint32_t *smem)
{
  constexpr int KC = TRSM_KC, ROWS = WG / COLS, SXN = ROWS * KC, SLN = COLS * KC, STN = ROWS * COLS;
#if SDPB_TRSM_ALIAS
  // the T tile of the second phase lies over the X chunk of the first (whose last pass ends with a barrier): 384
  // instead of 416 numbers of LDS per workgroup = five workgroups per CU instead of four
  static_assert(SXN <= STN, "the X chunk fits under the T tile");
  uint32_t *st = smem, *sx = smem, *sl = smem + (NL + 2) * STN;
#else
  uint32_t *sx = smem, *sl = sx + (NL + 2) * SXN, *st = sl + (NL + 2) * SLN;
#endif
  if((int)(blockIdx.x * ROWS) >= dx.rows)
    return;
  const int n = dl.rows;
  const int rl = threadIdx.x % ROWS, j = threadIdx.x / ROWS;
  const int r0 = blockIdx.x * ROWS, r = r0 + rl;
  const bool ok = r < dx.rows && j < nb;
  Acc<NL> acc = mw::acc_zero<NL>();
  if(ok)
    mw::acc_add(acc, mat_ld<NL>(X, dx, r, k0 + j));
  for(int k = k0 + nb; k < n; k += KC)
    {
      for(int e = threadIdx.x; e < SXN + SLN; e += WG)
        {
          if(e < SXN)
            {
              const int kk = e / ROWS, rr = e % ROWS;
              smem_st<NL, SXN>(sx, e, (r0 + rr < dx.rows && k + kk < n) ? mat_ld<NL>(X, dx, r0 + rr, k + kk) : mw::zero<NL>());
            }
          else
            {
              const int f = e - SXN, kk = f / COLS, jj = f % COLS;
              smem_st<NL, SLN>(sl, f, (jj < nb && k + kk < n) ? mat_ld<NL>(L, dl, k + kk, k0 + jj) : mw::zero<NL>());
            }
        }
      __syncthreads();
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            mw::acc_fms(acc, smem_ld<NL, SXN>(sx, kk * ROWS + rl), smem_ld<NL, SLN>(sl, kk * COLS + j));
        }
      __syncthreads();
    }
  smem_st<NL, STN>(st, j * ROWS + rl, ok ? mw::acc_result(acc) : mw::zero<NL>());
  acc = mw::acc_zero<NL>();
  // X(r, k0+j) = sum_{j2 >= j} T(r, j2) Li(k0+j2, k0+j)
  for(int c = 0; c < nb; c += KC)
    {
      for(int f = threadIdx.x; f < SLN; f += WG)
        {
          const int kk = f / COLS, jj = f % COLS;
          smem_st<NL, SLN>(sl, f, (jj < nb && c + kk < nb && c + kk >= jj) ? mat_ld<NL>(Li, di, k0 + c + kk, k0 + jj) : mw::zero<NL>());
        }
      __syncthreads(); // also orders the writes of st before the first read
      if(ok)
        {
#pragma unroll 1
          for(int kk = 0; kk < KC; ++kk)
            if(c + kk >= j && c + kk < nb)
              mw::acc_fma(acc, smem_ld<NL, STN>(st, (c + kk) * ROWS + rl), smem_ld<NL, SLN>(sl, kk * COLS + j));
        }
      __syncthreads();
    }
  if(ok)
    mat_st<NL>(X, dx, r, k0 + j, mw::acc_result(acc));
}
template <int NL> __global__ void __launch_bounds__(WG) k_trsm_rln_panel(Batch L, Batch Li, Batch X, int p)
{
  constexpr int KC = TRSM_KC;
  const int q = blockIdx.y;
  const MatDesc dl = L.d[q], di = Li.d[q], dx = X.d[q];
  const int k0 = PB * p;
  if(k0 >= dl.rows)
    return;
  const int nb = dl.rows - k0 < PB ? dl.rows - k0 : PB;
  constexpr int MINC = PB < 8 ? PB : 8, NMAX = (WG / MINC + MINC) * KC > (WG / PB + PB) * KC ? (WG / MINC + MINC) * KC : (WG / PB + PB) * KC;
#if SDPB_TRSM_ALIAS
  __shared__ uint32_t smem[(NL + 2) * (WG + PB * KC)];
#else
  __shared__ uint32_t smem[(NL + 2) * (NMAX + WG)];
#endif
  if constexpr(PB >= 32)
    {
      if(nb <= 8)
        return trsm_rln_tile<NL, 8>(L, Li, X, dl, di, dx, k0, nb, smem);
      if(nb <= 16)
        return trsm_rln_tile<NL, 16>(L, Li, X, dl, di, dx, k0, nb, smem);
    }
  trsm_rln_tile<NL, PB>(L, Li, X, dl, di, dx, k0, nb, smem);
}

// ---------------------------------------------------------------------------
// Batched GEMM, one lane per output element, 16x16 output tile per workgroup.
//   C = (+/-) op(A) op(B) (+ C),   op = transpose if TA / TB
// sym: only tiles on/below the diagonal are computed and mirrored (El::Syrk LOWER +
// MakeSymmetric, compute_A_X_inv.cxx:28-29; compute_A_Y.cxx:35,45).
// Replaces El::Gemm at compute_A_Y.cxx:32,35, scale_multiply_add.cxx:10.
// ---------------------------------------------------------------------------
// Extras: TB multiplies by B^T; trans_out stores C^T (so that the triangular solves
// that follow run on rows with coalesced loads); has_sub fuses "- Sub(i,j)".
template <int NL, bool TA, bool TB>
__global__ void __launch_bounds__(WG)
  k_gemm(Batch A, Batch B, Batch C, int alpha_neg, int beta_one, int sym, Batch Sub, int has_sub, int trans_out)
{
  const int q = blockIdx.y;
  const MatDesc da = A.d[q], db = B.d[q], dc = C.d[q];
  const int M = trans_out ? dc.cols : dc.rows, Nn = trans_out ? dc.rows : dc.cols, K = TA ? da.rows : da.cols;
  const int tiles_i = (M + 15) / 16, tiles_j = (Nn + 15) / 16;
  const int tile = blockIdx.x;
  if(tile >= tiles_i * tiles_j)
    return;
  int i, j;
  if(sym)
    {
      const int ti = tile % tiles_i, tj = tile / tiles_i;
      if(tj > ti)
        return;
      i = ti * 16 + (threadIdx.x & 15);
      j = tj * 16 + (threadIdx.x >> 4);
    }
  else
    {
      // full products: outputs in column-major order, WG consecutive ones per workgroup (a 40 x 40 block
      // fills 6.25 workgroups instead of nine 16 x 16 tiles of which five are ragged: 89 % instead of
      // 69 % of the launched lanes hold an output); the grid is still sized in tiles, the surplus exits
      const int idx = tile * WG + threadIdx.x;
      i = idx % M;
      j = idx / M;
    }
  if(i >= M || j >= Nn || (sym && j > i))
    return;
  Acc<NL> sum = mw::acc_zero<NL>();
  for(int k = 0; k < K; ++k)
    {
      const Mw<NL> a = TA ? mat_ld<NL>(A, da, k, i) : mat_ld<NL>(A, da, i, k);
      const Mw<NL> b = TB ? mat_ld<NL>(B, db, j, k) : mat_ld<NL>(B, db, k, j);
      mw::acc_fma(sum, a, b, (uint32_t)alpha_neg);
    }
  if(has_sub)
    mw::acc_add(sum, mat_ld<NL>(Sub, Sub.d[q], i, j), 1u);
  const int oi = trans_out ? j : i, oj = trans_out ? i : j;
  if(beta_one)
    mw::acc_add(sum, mat_ld<NL>(C, dc, oi, oj));
  const Mw<NL> acc = mw::acc_result(sum);
  mat_st<NL>(C, dc, oi, oj, acc);
  if(sym && i != j)
    mat_st<NL>(C, dc, oj, oi, acc);
}

// in-place transpose of square matrices
template <int NL> __global__ void __launch_bounds__(WG) k_transpose(Batch A)
{
  const int q = blockIdx.y;
  const MatDesc d = A.d[q];
  const int idx = blockIdx.x * WG + threadIdx.x;
  if(idx >= d.rows * d.rows)
    return;
  const int i = idx % d.rows, j = idx / d.rows;
  if(i <= j)
    return;
  const Mw<NL> x = mat_ld<NL>(A, d, i, j), y = mat_ld<NL>(A, d, j, i);
  mat_st<NL>(A, d, i, j, y);
  mat_st<NL>(A, d, j, i, x);
}

// A = (A + A^T)/2, optionally negated (Block_Diagonal_Matrix::symmetrize, :95-109)
template <int NL> __global__ void __launch_bounds__(WG) k_symmetrize(Batch A, int negate)
{
  const int q = blockIdx.y;
  const MatDesc d = A.d[q];
  const int idx = blockIdx.x * WG + threadIdx.x;
  if(idx >= d.rows * d.rows)
    return;
  const int i = idx % d.rows, j = idx / d.rows;
  if(i < j)
    return;
  Mw<NL> s = mw::mul_2exp(mw::add(mat_ld<NL>(A, d, i, j), mat_ld<NL>(A, d, j, i)), -1);
  if(negate)
    s = mw::neg(s);
  mat_st<NL>(A, d, i, j, s);
  if(i != j)
    mat_st<NL>(A, d, j, i, s);
}

// One panel step of the blocked substitution with the Cholesky factor of Q
// (El::cholesky::SolveAfter, solve_schur_complement_equation.cxx:64):
//   xp = Linv_pp * rhs[k0..k0+nb)            (TRANS: Linv_pp^T)
//   out[k0..k0+nb) = xp                      (written by workgroup 0)
//   forward : rhs[r] -= sum_k L(r, k0+k) xp[k]   for r >= k0+nb
//   backward: rhs[r] -= sum_k L(k0+k, r) xp[k]   for r <  k0        (TRANS)
// Every workgroup recomputes the small diagonal product (so one launch per panel
// suffices) and owns QS_ROWS rows of the update; each dot product is split over
// QS_SEG lanes and reduced through LDS, so the dependent chain per launch is
// 2*(nb/QS_SEG + QS_SEG) multi-word operations instead of 2*nb.
constexpr int QS_ROWS = PB, QS_SEG = WG / PB; // QS_ROWS * QS_SEG == WG; nb <= QS_ROWS
template <int NL, bool TRANS>
__global__ void __launch_bounds__(WG) k_qsolve_panel(Batch Q, Batch Linv, mw::Ptr rhs, mw::Ptr out, int k0)
{
  const MatDesc dq = Q.d[0], di = Linv.d[0];
  const int N = dq.rows, nb = di.rows, t = threadIdx.x;
  const int i = t % QS_ROWS, seg = t / QS_ROWS;
  const int kper = (nb + QS_SEG - 1) / QS_SEG;
  __shared__ Mw<NL> sx[QS_ROWS], sxp[QS_ROWS], part[WG];
  if(t < nb)
    sx[t] = mw::load<NL>(rhs, (size_t)k0 + t);
  __syncthreads();
  Acc<NL> acc = mw::acc_zero<NL>();
  if(i < nb)
    for(int k = seg * kper; k < (seg + 1) * kper && k < nb; ++k)
      {
        if(!TRANS && k <= i)
          mw::acc_fma(acc, mat_ld<NL>(Linv, di, i, k), sx[k]);
        if(TRANS && k >= i)
          mw::acc_fma(acc, mat_ld<NL>(Linv, di, k, i), sx[k]);
      }
  part[t] = mw::acc_result(acc);
  __syncthreads();
  if(t < nb)
    {
      Acc<NL> ss = mw::acc_zero<NL>();
      for(int g = 0; g < QS_SEG; ++g)
        mw::acc_add(ss, part[g * QS_ROWS + t]);
      const Mw<NL> s = mw::acc_result(ss);
      sxp[t] = s;
      if(blockIdx.x == 0)
        mw::store<NL>(out, (size_t)k0 + t, s);
    }
  __syncthreads();
  const int first = TRANS ? 0 : k0 + nb, count = TRANS ? k0 : N - k0 - nb;
  const int ri = blockIdx.x * QS_ROWS + i;
  acc = mw::acc_zero<NL>();
  if(ri < count)
    {
      const int r = first + ri;
      for(int k = seg * kper; k < (seg + 1) * kper && k < nb; ++k)
        {
          const Mw<NL> l = TRANS ? mat_ld<NL>(Q, dq, k0 + k, r) : mat_ld<NL>(Q, dq, r, k0 + k);
          mw::acc_fma(acc, l, sxp[k]);
        }
    }
  __syncthreads();
  part[t] = mw::acc_result(acc);
  __syncthreads();
  if(seg == 0 && ri < count)
    {
      const size_t r = (size_t)first + ri;
      Acc<NL> ss = mw::acc_zero<NL>();
      mw::acc_add(ss, mw::load<NL>(rhs, r));
      for(int g = 0; g < QS_SEG; ++g)
        mw::acc_add(ss, part[g * QS_ROWS + t], 1u);
      mw::store<NL>(rhs, r, mw::acc_result(ss));
    }
}

// The same panel step with the dependent chain cut to 2 x (one product + a two-level sum):
// 1024-lane workgroups, lane (i, k) forms ONE product of row i's dot product; the PB products of a
// row meet in a limb-major (conflict-free) LDS image and are summed exactly, eight at a time by four
// lanes and then by one (mw::Acc: an aligned add per term, one rounding per level), instead of a
// five-level tree of rounded adds over an array of 80-byte structures.  Every operand that does not
// depend on the first sum is loaded before it (the rows of L outside the panel, the old right-hand
// side).  In the transposed (backward) sweep the column index is the fast lane index (L(k0+k, r) is
// contiguous in k): with the row index fast every lane read a cache line of its own, 14 instead of 3 us
// before the first product.  Measured per launch on C4 with the in-kernel clock (forward / backward):
//   five-level tree, row index fast in both sweeps                43 / 47 us
//   this kernel                                                   38 / 35 us
//   256 lanes x 4 products in one accumulator, sum of 8           39 / 39 us
//   512 lanes x 2 products, sums 4 + 4                            38 / 40 us
// (a lone wavefront issues a multiply-add pair every 18 cycles, half of what its SIMD can take, and an
// aligned add costs it ~1 us: fewer, fatter lanes do not pay).  The substitution is a chain of 2 N/PB
// such launches per right-hand side, replicated on every GPU, so its length is what matters.
// Round 3, measured and rejected (profiles/r03f_qsolve_big_steps.txt): the same substitution in steps of
// 128 columns with explicitly inverted 128 x 128 diagonal blocks (assembled from the 32 x 32 inverses by block
// substitution).  16 instead of 64 launches per right-hand side, but a launch then carries 12 products per
// lane of a 1024-lane workgroup, and with four wavefronts per SIMD a dependent multi-word multiply-add takes
// 3.8 us (the SIMD's 2280 cycles per wavefront-term are a throughput figure): 123 / 91 us per launch
// forward / backward, 3.4 instead of 4.8 ms per iteration for the substitutions, plus 1.2 ms for the block
// inverses on the Cholesky(Q) chain — no net gain at one rank or at eight.
constexpr int QS2_T = PB * PB <= 1024 ? PB * PB : 1024;
constexpr int QS2_G = PB < 4 ? PB : 4; // lanes per row in the first level of the sum
#ifdef SDPB_QS_TRACE
__device__ long long qs_trace[2][64][8];
#define QS_MARK(n)                                                                                                                         \
  if(blockIdx.x == 0 && threadIdx.x == 0)                                                                                                    \
  qs_trace[TRANS ? 1 : 0][(k0 / PB) & 63][n] = wall_clock64()
#else
#define QS_MARK(n)
#endif
template <int NL, bool TRANS>
__global__ void __launch_bounds__(QS2_T) k_qsolve_panel2(Batch Q, Batch Linv, mw::Ptr rhs, mw::Ptr out, int k0)
{
  raise_chain_priority();
  QS_MARK(0);
  static_assert(PB * PB <= 1024 && PB % QS2_G == 0, "one lane per (row, column) of a panel");
  const MatDesc dq = Q.d[0], di = Linv.d[0];
  const int N = dq.rows, nb = di.rows, t = threadIdx.x;
  // product lanes (ip, kp): the fast lane index runs along memory.  Row i's products sit at slots
  // i (PB+1) + k in the transposed sweep (odd stride: the writes along k and the reads along i are
  // both conflict-free), at k PB + i in the forward sweep.
  const int ip = TRANS ? t / PB : t % PB, kp = TRANS ? t % PB : t / PB;
  // summing lanes: row i1, group g1
  const int i1 = t % PB, g1 = t / PB;
  constexpr int STR = PB * (PB + 1);
  auto slot = [](int i, int k) { return TRANS ? i * (PB + 1) + k : k * PB + i; };
  __shared__ uint32_t part[(NL + 2) * STR], sx[(NL + 2) * PB];
  const int first = TRANS ? 0 : k0 + nb, count = TRANS ? k0 : N - k0 - nb;
  const int rp = blockIdx.x * PB + ip, r1 = blockIdx.x * PB + i1;
  const bool tri = ip < nb && kp < nb && (TRANS ? kp >= ip : kp <= ip), upd = rp < count && kp < nb;
  const Mw<NL> li = tri ? (TRANS ? mat_ld<NL>(Linv, di, kp, ip) : mat_ld<NL>(Linv, di, ip, kp)) : mw::zero<NL>();
  const Mw<NL> lq = upd ? (TRANS ? mat_ld<NL>(Q, dq, k0 + kp, first + rp) : mat_ld<NL>(Q, dq, first + rp, k0 + kp)) : mw::zero<NL>();
  const bool last = g1 == 0 && r1 < count;
  const Mw<NL> old = last ? mw::load<NL>(rhs, (size_t)first + r1) : mw::zero<NL>(); // not touched by this launch before its own store
  if(t < PB)
    smem_st<NL, PB>(sx, t, t < nb ? mw::load<NL>(rhs, (size_t)k0 + t) : mw::zero<NL>());
  __syncthreads();
  QS_MARK(1);
  // level 1 of the sum over k: lanes g1 < QS2_G take every QS2_G-th product of their row
  auto level1 = [&](const Mw<NL> &p) __attribute__((always_inline)) {
    smem_st<NL, STR>(part, slot(ip, kp), p);
    __syncthreads();
    if(g1 < QS2_G)
      {
        Acc<NL> a = mw::acc_zero<NL>();
        for(int kk = g1; kk < nb; kk += QS2_G)
          mw::acc_add(a, smem_ld<NL, STR>(part, slot(i1, kk)));
        smem_st<NL, STR>(part, slot(i1, g1), mw::acc_result(a)); // in place: this slot is read by this lane only
      }
    __syncthreads();
  };
  // xp = Linv_pp rhs_p (lower triangular; TRANS: its transpose)
  level1(mw::mul(li, smem_ld<NL, PB>(sx, kp)));
  QS_MARK(2);
  if(g1 == 0)
    {
      Acc<NL> a = mw::acc_zero<NL>();
      for(int g = 0; g < QS2_G; ++g)
        mw::acc_add(a, smem_ld<NL, STR>(part, slot(i1, g)));
      const Mw<NL> xp = mw::acc_result(a);
      smem_st<NL, PB>(sx, i1, xp); // every lane read its sx[kp] before the barriers of level1
      if(blockIdx.x == 0 && i1 < nb)
        mw::store<NL>(out, (size_t)k0 + i1, xp);
    }
  __syncthreads();
  QS_MARK(3);
  // rows outside the panel: rhs[r] -= sum_k L(r, k0 + k) xp[k]
  level1(mw::mul(lq, smem_ld<NL, PB>(sx, kp)));
  QS_MARK(4);
  if(last)
    {
      Acc<NL> a = mw::acc_zero<NL>();
      mw::acc_add(a, old);
      for(int g = 0; g < QS2_G; ++g)
        mw::acc_add(a, smem_ld<NL, STR>(part, slot(i1, g)), 1u);
      mw::store<NL>(rhs, (size_t)first + r1, mw::acc_result(a));
    }
  QS_MARK(5);
}

// Round 4: the same panel step with both sums formed ACROSS the lanes as exact integer sums (mw.hpp: raw terms).
// In k_qsolve_panel2 a sum of 32 products is 8 + 4 dependent aligned adds (an aligned add costs a lone
// wavefront ~0.8 us: ~10 us per sum, two sums per launch, 2 N/PB launches per right-hand side, replicated on every
// GPU).  Here lane (i, k) forms the raw product, the lanes of a row agree on the largest exponent (LDS atomicMax) and
// on the signs (LDS atomicOr), every lane aligns its product to the window below that exponent and stores the
// window limb-major; lane (i, l) then adds limb l of the 32 windows of row i as a signed 64-bit integer — 32
// independent LDS reads instead of a chain of adds — and one lane per row propagates the carries and normalises
// once.  Integer addition is associative: the result does not depend on lane order or layout (the emulation build
// gives the same bits by construction).  Slots: forward k PB + i (the row index is the fast lane index, as the
// loads of Q want), transposed i PB + (k xor i) (writes along k and reads along i both hit 32 distinct banks).
template <int NL, bool TRANS>
__global__ void __launch_bounds__(QS2_T) k_qsolve_panel3(Batch Q, Batch Linv, mw::Ptr rhs, mw::Ptr out, int k0)
{
  raise_chain_priority();
  static_assert(PB * PB <= 1024 && (PB & (PB - 1)) == 0, "one lane per (row, column) of a panel; xor slots");
  constexpr int W = NL + 2, STR = PB * PB;
  const MatDesc dq = Q.d[0], di = Linv.d[0];
  const int N = dq.rows, nb = di.rows, t = threadIdx.x;
  const int ip = TRANS ? t / PB : t % PB, kp = TRANS ? t % PB : t / PB;
  auto slot = [](int i, int k) { return TRANS ? i * PB + (k ^ i) : k * PB + i; };
  __shared__ uint32_t win[W * STR], sx[(NL + 2) * PB], smask[PB];
  __shared__ long long csum[W * PB];
  __shared__ int semax[PB];
  const int first = TRANS ? 0 : k0 + nb, count = TRANS ? k0 : N - k0 - nb;
  const int rp = blockIdx.x * PB + ip;
  const bool tri = ip < nb && kp < nb && (TRANS ? kp >= ip : kp <= ip), upd = rp < count && kp < nb;
  const Mw<NL> li = tri ? (TRANS ? mat_ld<NL>(Linv, di, kp, ip) : mat_ld<NL>(Linv, di, ip, kp)) : mw::zero<NL>();
  const Mw<NL> lq = upd ? (TRANS ? mat_ld<NL>(Q, dq, k0 + kp, first + rp) : mat_ld<NL>(Q, dq, first + rp, k0 + kp)) : mw::zero<NL>();
  const bool last = t < PB && (int)blockIdx.x * PB + t < count;
  const Mw<NL> old = last ? mw::load<NL>(rhs, (size_t)first + blockIdx.x * PB + t) : mw::zero<NL>(); // not touched by this launch before its own store
  if(t < PB)
    {
      smem_st<NL, PB>(sx, t, t < nb ? mw::load<NL>(rhs, (size_t)k0 + t) : mw::zero<NL>());
      semax[t] = mw::EZERO;
      smask[t] = 0u;
    }
  __syncthreads();
  // sum over k of the terms (P, e, neg) of row ip, for every row: the window of row i ends up in csum[. PB + i]
  auto rowsum = [&](const uint32_t(&P)[NL + 1], int32_t e, uint32_t neg) __attribute__((always_inline)) {
    if(e != mw::EZERO)
      {
        atomicMax(&semax[ip], e);
        if(neg)
          atomicOr(&smask[ip], 1u << kp);
      }
    __syncthreads();
    uint32_t x[W];
    mw::term_align<NL>(P, e, semax[ip], x);
#pragma unroll
    for(int l = 0; l < W; ++l)
      win[l * STR + slot(ip, kp)] = x[l];
    __syncthreads();
    for(int idx = t; idx < W * PB; idx += QS2_T)
      {
        const int i = idx % PB, l = idx / PB;
        const uint32_t m = smask[i];
        long long c = 0;
#pragma unroll 8
        for(int k = 0; k < PB; ++k)
          {
            const long long w = (long long)win[l * STR + slot(i, k)];
            c += ((m >> k) & 1u) ? -w : w;
          }
        csum[l * PB + i] = c;
      }
    __syncthreads();
  };
  // carries of row i -> an accumulator window (two's complement, top limb = headroom)
  auto window = [&](int i) __attribute__((always_inline)) {
    Acc<NL> a;
    long long carry = 0;
#pragma unroll
    for(int l = 0; l < W; ++l)
      {
        const long long v = csum[l * PB + i] + carry;
        a.w[l] = (uint32_t)v;
        carry = v >> 32;
      }
    a.etop = semax[i];
    return a;
  };
  uint32_t P[NL + 1], neg;
  int32_t e;
  // xp = Linv_pp rhs_p (lower triangular; TRANS: its transpose)
  mw::term_mul<NL>(li, smem_ld<NL, PB>(sx, kp), P, e, neg);
  rowsum(P, e, neg);
  if(t < PB)
    {
      const Mw<NL> xp = mw::acc_result(window(t));
      smem_st<NL, PB>(sx, t, xp); // every lane read its sx[kp] before the barriers of rowsum
      if(blockIdx.x == 0 && t < nb)
        mw::store<NL>(out, (size_t)k0 + t, xp);
      semax[t] = mw::EZERO;
      smask[t] = 0u;
    }
  __syncthreads();
  // rows outside the panel: rhs[r] -= sum_k L(r, k0 + k) xp[k]
  mw::term_mul<NL>(lq, smem_ld<NL, PB>(sx, kp), P, e, neg);
  rowsum(P, e, neg ^ 1u);
  if(last)
    {
      Acc<NL> a = window(t);
      mw::acc_add(a, old);
      mw::store<NL>(rhs, (size_t)first + blockIdx.x * PB + t, mw::acc_result(a));
    }
}

// ---------------------------------------------------------------------------
// Block-structure helpers shared by the SDP-specific kernels
// ---------------------------------------------------------------------------
struct BlockDesc // one SDP block j (local index)
{
  int m, K, P;            // dim, num_points, schur size (Block_Info.hxx:54-58)
  int rows[2], n[2];      // bilinear basis heights, psd sizes (Block_Info.hxx:86-114)
  unsigned long long voff; // offset of this block's length-P vectors (x, c, dx, d)
  int global_index;
};
// p in [0,P) -> (column_block, row_block, k) with p = ((cb(cb+1))/2 + rb) K + k
MW_HD void decode_p(int p, int K, int &cb, int &rb, int &k)
{
  const int t = p / K;
  k = p - t * K;
  cb = 0;
  while((cb + 1) * (cb + 2) / 2 <= t)
    ++cb;
  rb = t - cb * (cb + 1) / 2;
}

// bases_blocks[2j+b] = I_m (x) bilinear_bases[2j+b]   (set_bases_blocks.cxx:3-22); Et
// receives the transpose (rows of E^T are the right-hand sides of the pairing solve).
template <int NL>
__global__ void __launch_bounds__(WG) k_build_bases_block(Batch bases, Batch E, Batch Et, const BlockDesc *blk)
{
  const int q = blockIdx.y;
  const MatDesc de = E.d[q], dbs = bases.d[q], det = Et.d[q];
  const int rs = blk[q >> 1].rows[q & 1], K = blk[q >> 1].K;
  const int idx = blockIdx.x * WG + threadIdx.x;
  if(idx >= de.rows * de.cols)
    return;
  const int row = idx % de.rows, col = idx / de.rows;
  Mw<NL> v = mw::zero<NL>();
  if(row / rs == col / K)
    v = mat_ld<NL>(bases, dbs, row % rs, col % K);
  mat_st<NL>(E, de, row, col, v);
  mat_st<NL>(Et, det, col, row, v);
}

// Schur complement assembly (compute_schur_complement.cxx:15-125): element-wise in
// the pairing tiles, lower triangle computed and mirrored (MakeSymmetric LOWER :121).
// AX, AY batches are indexed 2j+parity; S by j.
template <int NL>
__global__ void __launch_bounds__(WG) k_schur_complement(Batch AX, Batch AY, Batch S, const BlockDesc *blk, int strips)
{
  // XCD-aware order (1-D grid of 8 * ceil(count / 8) * strips workgroups): workgroup b runs on XCD b % 8,
  // and all `strips` workgroups of a block go to the SAME XCD (blocks j = xcd, xcd + 8, ...), so the
  // block's pairing tiles — read 16 times over by the (r,s)-pair sub-blocks of S_j — are fetched into
  // one L2 once instead of into all eight (the kernel is bound by that traffic, not by its products).
  const int xcd = (int)(blockIdx.x % 8), slot = (int)(blockIdx.x / 8);
  const int j = (slot / strips) * 8 + xcd, strip = slot % strips;
  if(j >= S.count)
    return;
  const MatDesc ds = S.d[j];
  const int P = ds.rows, K = blk[j].K;
  // one lane per entry of the LOWER triangle, packed column by column (rows fastest: a wavefront walks down a
  // column, so `row` below stays the fast lane index).  Round 4: the P x P grid left every lane above the diagonal
  // idle — the kernel is bound by its 8 multi-word products per entry (1.4 * 10^7 on C4 = 0.21 ms at the chip's
  // product rate, twice its 0.11 ms of HBM time), not by HBM, and spent them at half occupancy.
  const size_t idx = (size_t)strip * WG + threadIdx.x;
  if(idx >= (size_t)P * (P + 1) / 2)
    return;
  // column C: the largest C with C P - C (C - 1) / 2 <= idx
  const double tp1 = 2.0 * P + 1.0;
  int C = (int)((tp1 - mw::host_device_sqrt(tp1 * tp1 - 8.0 * (double)idx)) * 0.5);
  C = C < 0 ? 0 : (C > P - 1 ? P - 1 : C);
  while(C > 0 && (size_t)C * P - (size_t)C * (C - 1) / 2 > idx)
    --C;
  while(C + 1 < P && (size_t)(C + 1) * P - (size_t)(C + 1) * C / 2 <= idx)
    ++C;
  const int R = C + (int)(idx - ((size_t)C * P - (size_t)C * (C - 1) / 2));
  int c0, r0, row, c1, r1, col;
  decode_p(R, K, c0, r0, row);
  decode_p(C, K, c1, r1, col);
  Acc<NL> es = mw::acc_zero<NL>();
  for(int b = 0; b < 2; ++b)
    {
      const MatDesc dx = AX.d[2 * j + b], dy = AY.d[2 * j + b];
      // A_X_inv tile [cb][rb](r,c) = AX(cb K + r, rb K + c)  (compute_A_X_inv.cxx:39-56)
      // A_Y tile     [cb][rb](r,c) = AY(cb K + c, rb K + r)  (compute_A_Y.cxx:47-64)
#define AXT(cb, rb) mat_ld<NL>(AX, dx, (cb)*K + row, (rb)*K + col)
      // A_Y is formed as a symmetric product (k_gemm sym = 1: lower tiles computed and mirrored, as El::Syrk +
      // MakeSymmetric in compute_A_Y.cxx:35,45), so AY(cb K + c, rb K + r) and AY(rb K + r, cb K + c) are the same
      // bits; the second form runs along `row`, the fast lane index: one cache line per word plane instead of one
      // per lane (round 4)
#define AYT(cb, rb) mat_ld<NL>(AY, dy, (rb)*K + row, (cb)*K + col)
      mw::acc_fma(es, AXT(c0, r1), AYT(c1, r0));
      mw::acc_fma(es, AXT(r0, r1), AYT(c1, c0));
      mw::acc_fma(es, AXT(c0, c1), AYT(r1, r0));
      mw::acc_fma(es, AXT(r0, c1), AYT(r1, c0));
#undef AXT
#undef AYT
    }
  // The lower triangle only.  The reference mirrors it (MakeSymmetric LOWER, :121) for El::Cholesky; here S_j goes
  // straight into blocked_cholesky, which reads the lower triangle and leaves zeros above the diagonal — the mirror
  // store was dead work, and a costly one: lanes of a wavefront run down a column, so the mirrored entries lie a whole
  // row apart, 64 cache lines per store instruction and word plane (WRITE_SIZE 0.77 GB per launch for 0.28 GB of S,
  // profiles/r04l_pmc_WRITE_SIZE.txt).
  const Mw<NL> e = mw::mul_2exp(mw::acc_result(es), -2);
  mat_st<NL>(S, ds, R, C, e);
}

// dual_residues[p] = c[p] - sum_b diag(A_Y tile)[k]   (the -(B y)[p] term is added by
// k_gemv_n; compute_dual_residues_and_error.cxx:7-66).  One lane per p.
template <int NL>
__global__ void __launch_bounds__(WG) k_dual_residues(Batch AY, mw::CPtr c, mw::Ptr d, const BlockDesc *blk)
{
  const int j = blockIdx.y;
  const BlockDesc bl = blk[j];
  const int p = blockIdx.x * WG + threadIdx.x;
  if(p >= bl.P)
    return;
  int cb, rb, k;
  decode_p(p, bl.K, cb, rb, k);
  Mw<NL> acc = mw::load<NL>(c, (size_t)bl.voff + p);
  for(int b = 0; b < 2; ++b)
    {
      const MatDesc dy = AY.d[2 * j + b];
      acc = mw::sub(acc, mat_ld<NL>(AY, dy, cb * bl.K + k, rb * bl.K + k));
    }
  mw::store<NL>(d, (size_t)bl.voff + p, acc);
}

// result = sum_p a[p] A_p (+/- addend)   (constraint_matrix_weighted_sum.cxx:14-66)
// out batch is indexed 2j+parity.  addend_sign: 0 none, +1 add, -1 subtract.
// The scaled samples t(pair, j, k) = q_j(x_k) a[pair, k] are rounded products that every row i of the
// output shares, so they are formed once by k_scale_bases (into `scaled`, K x (pairs rs) per parity,
// k fastest) instead of rs times over inside the sum: same values, same order, half the products.
template <int NL> __global__ void __launch_bounds__(WG) k_scale_bases(Batch bases, mw::CPtr a, Batch scaled, const BlockDesc *blk)
{
  const int q = blockIdx.y;
  const BlockDesc bl = blk[q >> 1];
  const MatDesc dbs = bases.d[q], dsc = scaled.d[q];
  const int rs = bl.rows[q & 1], K = bl.K;
  const int idx = blockIdx.x * WG + threadIdx.x;
  if(idx >= dsc.rows * dsc.cols)
    return;
  const int k = idx % K, col = idx / K, pair = col / rs, jj = col % rs;
  mat_st<NL>(scaled, dsc, k, col, mw::mul(mat_ld<NL>(bases, dbs, jj, k), mw::load<NL>(a, (size_t)bl.voff + (size_t)pair * K + k)));
}
template <int NL>
__global__ void __launch_bounds__(WG) k_constraint_weighted_sum(Batch bases, Batch scaled, Batch out, Batch addend,
                                                                int addend_sign, const BlockDesc *blk)
{
  const int q = blockIdx.y;
  const BlockDesc bl = blk[q >> 1];
  const MatDesc dout = out.d[q], dbs = bases.d[q], dsc = scaled.d[q];
  const int n = dout.rows, rs = bl.rows[q & 1];
  const int idx = blockIdx.x * WG + threadIdx.x;
  if(idx >= n * n)
    return;
  const int I = idx % n, Jc = idx / n;
  const int bi = I / rs, i = I % rs, bj = Jc / rs, jj = Jc % rs;
  const int hi = bi > bj ? bi : bj, lo = bi > bj ? bj : bi;
  const int col = (hi * (hi + 1) / 2 + lo) * rs + jj;
  Acc<NL> sum = mw::acc_zero<NL>();
  for(int k = 0; k < bl.K; ++k)
    mw::acc_fma(sum, mat_ld<NL>(bases, dbs, i, k), mat_ld<NL>(scaled, dsc, k, col));
  Mw<NL> acc = mw::acc_result(sum);
  if(hi != lo)
    acc = mw::mul_2exp(acc, -1);
  if(addend_sign)
    {
      const Mw<NL> ad = mat_ld<NL>(addend, addend.d[q], I, Jc);
      acc = addend_sign > 0 ? mw::add(acc, ad) : mw::sub(acc, ad);
    }
  mat_st<NL>(out, dout, I, Jc, acc);
}

// dx[p] = -dual_residues[p] - Tr(A_p Z)   (compute_schur_RHS.cxx:21-86)
template <int NL>
__global__ void __launch_bounds__(WG)
  k_schur_rhs(Batch bases, Batch Z, mw::CPtr dres, mw::Ptr dx, const BlockDesc *blk)
{
  const int j = blockIdx.y;
  const BlockDesc bl = blk[j];
  const int p = blockIdx.x * WG + threadIdx.x;
  if(p >= bl.P)
    return;
  int cb, rb, k;
  decode_p(p, bl.K, cb, rb, k);
  Mw<NL> acc = mw::neg(mw::load<NL>(dres, (size_t)bl.voff + p));
  for(int b = 0; b < 2; ++b)
    {
      const int rs = bl.rows[b];
      const MatDesc dz = Z.d[2 * j + b], dbs = bases.d[2 * j + b];
      Acc<NL> colsum = mw::acc_zero<NL>();
      for(int i = 0; i < rs; ++i)
        {
          Acc<NL> zq = mw::acc_zero<NL>();
          for(int l = 0; l < rs; ++l)
            mw::acc_fma(zq, mat_ld<NL>(Z, dz, rb * rs + i, cb * rs + l), mat_ld<NL>(bases, dbs, l, k));
          mw::acc_fma(colsum, mw::acc_result(zq), mat_ld<NL>(bases, dbs, i, k));
        }
      acc = mw::sub(acc, mw::acc_result(colsum));
    }
  mw::store<NL>(dx, (size_t)bl.voff + p, acc);
}

// The same right-hand side with the work of one (block, (r,s) pair) spread over a workgroup: lane
// (k, g) owns sample point k and every G-th basis row i; per row it forms
// W(i,k) = sum_l Z(rb rs + i, cb rs + l) q(l,k) and adds W(i,k) q(i,k) to its partial column sum
// (the Z operand is a broadcast across the lanes of a group, q is read along k).  The dependent
// chain per lane is 2 (rs/G)(rs + 1) products instead of 2 rs (rs + 1) (C4: 168 instead of 840), which
// is what this latency-bound kernel costs: 3.0 -> 0.4 ms per launch.  grid (max pairs, blocks).
template <int NL>
__global__ void __launch_bounds__(WG)
  k_schur_rhs2(Batch basesT, Batch Z, mw::CPtr dres, mw::Ptr dx, const BlockDesc *blk)
{
  // basesT is the K x rs transpose (sample index fastest): the lanes of a group, which differ in k,
  // read consecutive words.  Z is symmetric to the bit (symmetrize() ran just before), so the
  // broadcast operand is read as Z(cb rs + l, rb rs + i): consecutive l are consecutive words.
  // With the rs x K layout and Z along a row both loads touched one cache line per word and the
  // kernel was bound by 4.4 GB of L2 misses for 0.15 GB of data.
  const int j = blockIdx.y, t = threadIdx.x;
  const BlockDesc bl = blk[j];
  const int pair = blockIdx.x;
  if(pair >= bl.m * (bl.m + 1) / 2)
    return;
  int cb = 0;
  while((cb + 1) * (cb + 2) / 2 <= pair)
    ++cb;
  const int rb = pair - cb * (cb + 1) / 2;
  const int K = bl.K, KL = K < WG ? K : WG, G = WG / KL;
  const int kk = t % KL, g = t / KL;
  __shared__ Mw<NL> part[WG];
  for(int k0 = 0; k0 < K; k0 += KL)
    {
      const int k = k0 + kk;
      Acc<NL> colsum = mw::acc_zero<NL>();
      if(g < G && k < K)
        for(int b = 0; b < 2; ++b)
          {
            const int rs = bl.rows[b];
            const MatDesc dz = Z.d[2 * j + b], dbt = basesT.d[2 * j + b];
            for(int i = g; i < rs; i += G)
              {
                Acc<NL> zq = mw::acc_zero<NL>();
                for(int l = 0; l < rs; ++l)
                  mw::acc_fma(zq, mat_ld<NL>(Z, dz, cb * rs + l, rb * rs + i), mat_ld<NL>(basesT, dbt, k, l));
                mw::acc_fma(colsum, mw::acc_result(zq), mat_ld<NL>(basesT, dbt, k, i));
              }
          }
      part[t] = mw::acc_result(colsum);
      __syncthreads();
      if(g == 0 && k < K)
        {
          const size_t p = (size_t)bl.voff + (size_t)pair * K + k;
          Acc<NL> sum = mw::acc_zero<NL>();
          mw::acc_add(sum, mw::load<NL>(dres, p), 1u);
          for(int gg = 0; gg < G; ++gg)
            mw::acc_add(sum, part[gg * KL + kk], 1u);
          mw::store<NL>(dx, p, mw::acc_result(sum));
        }
      __syncthreads();
    }
}

// x := L^{-1} x (TRANS: L^{-T} x) for one vector per block, all panels inside ONE workgroup
// (the row-vector solves of solve_schur_complement_equation.cxx:24-31,76-78 were four launches of the
// matrix kernel with a single row each: a chain of P_j dependent products in one lane).  Per panel
//   t = x_p - L(p, other) x(other)   32 outputs, each dot product split over 8 lanes + LDS tree
//   x_p = Linv_pp t                  (TRANS: Linv_pp^T), same split
// Forward runs the panels upwards and "other" = the panels already solved (columns < k0); backward
// runs them downwards with rows > k0 + nb of L^T.  X.d[q] is a 1 x P row descriptor.
template <int NL, bool TRANS>
__global__ void __launch_bounds__(WG) k_vec_trsm(Batch L, Batch Li, Batch X)
{
  const int q = blockIdx.x, t = threadIdx.x;
  const MatDesc dl = L.d[q], di = Li.d[q], dx = X.d[q];
  const int n = dl.rows;
  if(n == 0)
    return;
  constexpr int SEG = WG / PB; // lanes per output
  const int i = t % PB, seg = t / PB;
  const int panels = (n + PB - 1) / PB;
  __shared__ Mw<NL> sx[PB], part[WG];
  auto vec = [&](int r) { return (size_t)dx.off + (size_t)r * dx.ld; }; // element r of the 1 x P row
  for(int pp = 0; pp < panels; ++pp)
    {
      const int p = TRANS ? panels - 1 - pp : pp;
      const int k0 = p * PB, nb = n - k0 < PB ? n - k0 : PB;
      // (1) t_i = x(k0+i) - sum over the solved part
      Acc<NL> acc = mw::acc_zero<NL>();
      if(i < nb)
        {
          const int lo = TRANS ? k0 + nb : 0, hi = TRANS ? n : k0;
          for(int k = lo + seg; k < hi; k += SEG)
            {
              const Mw<NL> l = TRANS ? mat_ld<NL>(L, dl, k, k0 + i) : mat_ld<NL>(L, dl, k0 + i, k);
              mw::acc_fms(acc, l, mw::load<NL>(X.p, vec(k)));
            }
          if(seg == 0)
            mw::acc_add(acc, mw::load<NL>(X.p, vec(k0 + i)));
        }
      part[t] = mw::acc_result(acc);
      __syncthreads();
      if(t < nb)
        {
          Acc<NL> s = mw::acc_zero<NL>();
          for(int gsg = 0; gsg < SEG; ++gsg)
            mw::acc_add(s, part[gsg * PB + t]);
          sx[t] = mw::acc_result(s);
        }
      __syncthreads();
      // (2) x_p = Linv_pp t (lower triangular; TRANS: its transpose)
      acc = mw::acc_zero<NL>();
      if(i < nb)
        for(int k = seg; k < nb; k += SEG)
          if(TRANS ? k >= i : k <= i)
            mw::acc_fma(acc, TRANS ? mat_ld<NL>(Li, di, k0 + k, k0 + i) : mat_ld<NL>(Li, di, k0 + i, k0 + k), sx[k]);
      part[t] = mw::acc_result(acc);
      __syncthreads();
      if(t < nb)
        {
          Acc<NL> s = mw::acc_zero<NL>();
          for(int gsg = 0; gsg < SEG; ++gsg)
            mw::acc_add(s, part[gsg * PB + t]);
          mw::store<NL>(X.p, vec(k0 + t), mw::acc_result(s));
        }
      __syncthreads(); // the solved entries are read from global memory by the next panel
    }
}

// part[j*N + n] = sum_p MT_j(n,p) * v_j[p]  (square: MT^2) — per-block partials of
// B^T x (compute_primal_residues_and_error_p_b_Bx.cxx:26-28), P^T dx
// (solve_schur_complement_equation.cxx:33-35) and the column norms^2 of P
// (Matrix_Normalizer.cxx:75-91).  One lane per (block, n); loads are coalesced in n.
template <int NL, bool SQUARE>
__global__ void __launch_bounds__(WG) k_gemv_t_partial(Batch MT, mw::CPtr v, mw::Ptr part, const BlockDesc *blk, int N)
{
  const int j = blockIdx.y;
  const BlockDesc bl = blk[j];
  const int n = blockIdx.x * WG + threadIdx.x;
  if(n >= N)
    return;
  const MatDesc dm = MT.d[j];
  Acc<NL> acc = mw::acc_zero<NL>();
  for(int p = 0; p < bl.P; ++p)
    {
      const Mw<NL> a = mat_ld<NL>(MT, dm, n, p);
      if(SQUARE)
        mw::acc_fma(acc, a, a);
      else
        mw::acc_fma(acc, a, mw::load<NL>(v, (size_t)bl.voff + p));
    }
  mw::store<NL>(part, (size_t)j * N + n, mw::acc_result(acc));
}
// out[n] = (base ? base[n] : 0) + sign * sum_j part[j*N+n]
template <int NL>
__global__ void __launch_bounds__(WG) k_sum_partials(mw::CPtr part, int J, int N, mw::CPtr base, int has_base, int sign, mw::Ptr out)
{
  // SP_G lanes share one n (J sequential adds in one lane were a J-long dependent chain on N lanes
  // only): lane (i, g) adds blocks g, g + SP_G, ... and the SP_G partial sums meet in LDS.
  // workgroup = SP_N consecutive n (coalesced loads along n) x SP_G groups
  constexpr int SP_G = 8, SP_N = WG / SP_G;
  const int i = threadIdx.x % SP_N, g = threadIdx.x / SP_N;
  const int n = blockIdx.x * SP_N + i;
  __shared__ Mw<NL> sm[WG];
  Acc<NL> acc = mw::acc_zero<NL>();
  if(n < N)
    for(int j = g; j < J; j += SP_G)
      mw::acc_add(acc, mw::load<NL>(part, (size_t)j * N + n), sign < 0 ? 1u : 0u);
  sm[threadIdx.x] = mw::acc_result(acc);
  __syncthreads();
  if(g != 0 || n >= N)
    return;
  Acc<NL> tot = mw::acc_zero<NL>();
  for(int gg = 0; gg < SP_G; ++gg)
    mw::acc_add(tot, sm[gg * SP_N + i]);
  if(has_base)
    mw::acc_add(tot, mw::load<NL>(base, n));
  mw::store<NL>(out, n, mw::acc_result(tot));
}
constexpr int SP_ROWS = WG / 8; // values of n per workgroup of k_sum_partials
// out_j[p] += sign * sum_n MT_j(n,p) v[n]: dx += P dy (solve_schur_complement_equation.
// cxx:69-74) and d -= B y (compute_dual_residues_and_error.cxx:52-54).  A workgroup
// owns 4 values of p; 64 lanes stride over n (coalesced column reads) and meet in an
// LDS tree, so the dependent chain is N/64 + 6 instead of N.
template <int NL>
__global__ void __launch_bounds__(WG) k_gemv_n(Batch MT, mw::CPtr v, mw::Ptr out, const BlockDesc *blk, int N, int sign)
{
  const int j = blockIdx.y;
  const BlockDesc bl = blk[j];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + sub;
  // the 64 partial sums of a row meet in a limb-major LDS image and are added exactly, eight at a time
  // by eight lanes and then by one (two barriers instead of a six-level tree of rounded adds)
  Acc<NL> sum = mw::acc_zero<NL>();
  if(p < bl.P)
    {
      const MatDesc dm = MT.d[j];
      for(int n = lane; n < N; n += 64)
        mw::acc_fma(sum, mat_ld<NL>(MT, dm, n, p), mw::load<NL>(v, n));
    }
#if defined(__HIP_DEVICE_COMPILE__)
  // A row is one wavefront: its 64 partial sums meet by wavefront shuffles (no LDS image, no barrier), in
  // the same two exact levels and the same order as the LDS path below (which the emulation build runs):
  // lane l adds the partials of lanes l % 8, l % 8 + 8, ..., then the eight level-1 sums are added to out.
  {
    const Mw<NL> mine = mw::acc_result(sum);
    Acc<NL> a = mw::acc_zero<NL>();
    for(int g = 0; g < 8; ++g)
      mw::acc_add(a, wave_get<NL>(mine, g * 8 + (lane & 7)));
    const Mw<NL> s8 = mw::acc_result(a);
    Acc<NL> t = mw::acc_zero<NL>();
    if(p < bl.P) // uniform in the wavefront
      mw::acc_add(t, mw::load<NL>(out, (size_t)bl.voff + p));
    for(int g = 0; g < 8; ++g)
      mw::acc_add(t, wave_get<NL>(s8, g), sign < 0 ? 1u : 0u);
    if(lane == 0 && p < bl.P)
      mw::store<NL>(out, (size_t)bl.voff + p, mw::acc_result(t));
  }
#else
  __shared__ uint32_t sm[(NL + 2) * WG];
  smem_st<NL, WG>(sm, threadIdx.x, mw::acc_result(sum));
  __syncthreads();
  if(lane < 8)
    {
      Acc<NL> a = mw::acc_zero<NL>();
      for(int g = 0; g < 8; ++g)
        mw::acc_add(a, smem_ld<NL, WG>(sm, sub * 64 + g * 8 + lane));
      smem_st<NL, WG>(sm, threadIdx.x, mw::acc_result(a)); // slot sub*64 + lane: read by this lane only (g = 0)
    }
  __syncthreads();
  if(lane == 0 && p < bl.P)
    {
      Acc<NL> a = mw::acc_zero<NL>();
      mw::acc_add(a, mw::load<NL>(out, (size_t)bl.voff + p));
      for(int g = 0; g < 8; ++g)
        mw::acc_add(a, smem_ld<NL, WG>(sm, sub * 64 + g), sign < 0 ? 1u : 0u);
      mw::store<NL>(out, (size_t)bl.voff + p, mw::acc_result(a));
    }
#endif
}

// ---------------------------------------------------------------------------
// Q = P^T P in fixed point  (the reference's bigint_syrk: Matrix_Normalizer.cxx:
// 174-192 normalise-and-shift, bigint_syrk_blas.cxx:183-302 exact integer syrk,
// Matrix_Normalizer.cxx:245-264 restore).  After column normalisation |P'| <= 1, so
// P' * 2^FXB is an FX-limb integer plus sign; the product of two of them is
// accumulated exactly in a (2FX+2)-limb two's-complement integer — no alignment,
// no normalisation inside the hot loop, just v_mad_u64_u32 + carry.
// ---------------------------------------------------------------------------
// The fixed-point image is BIASED so that every operand of the hot loop is a
// non-negative integer and whole columns of products can be accumulated across rows
// without a sign decision per product:
//   v  = trunc(P' 2^FB) (signed),  FB = 32 FX - 3 fractional bits,
//   a' = v + C,  C = 2^FB,  0 < a' < 2^(32FX-2),
//   a' = hi B + lo,  B = 2^(32M-1),  M = FX/2,  lo, hi < B,  s = lo + hi < 2^(32M).
// Planes of fx (limb-major, 3M planes): lo[0..M), hi[0..M), s[0..M).
//   sum_r a'_ri a'_rj = HH B^2 + (SS - LL - HH) B + LL        (one Karatsuba level over
//   LL = sum lo lo,  HH = sum hi hi,  SS = sum s s             the SUMS: 3/4 of the MACs)
//   sum_r v_ri v_rj   = G(i,j) - C (S_i + S_j) + n C^2,  S_i = sum_r a'_ri  (k_fx_colsum)
// Everything is integer and exact; three bits of the 32 FX-bit image pay for the bias
// and for the carry-free s = lo + hi.
//
// TWO Karatsuba levels (fx_two_level<FX>(): FX divisible by 4): each of lo, hi, s is split
// once more, x = x1 B2 + x0, B2 = 2^(32 M2 - 1), M2 = FX/4, t = x0 + x1 < 2^(32 M2), so one row
// pair costs 9 products of M2 x M2 limbs (9/16 of the plain FX x FX product instead of 3/4).
// The carry-free sums of the second level cost four more bits: FB = 32 FX - 7,
// lo, hi < B = 2^(32M-3), s < 2^(32M-2), pieces < 2^(32 M2 - 1).  The image then holds nine
// M2-limb pieces per element, piece-major: words [(g * stride + idx) * M2, +M2), g = 3 k + u
// with k in (lo, hi, s) and u in (x0, x1, t), so a piece is one 16-byte load when M2 = 4.
template <int FX> constexpr bool fx_two_level()
{
#ifdef SDPB_SYRK_ONE_LEVEL
  return false;
#else
  return FX % 4 == 0 && FX <= 32;
#endif
}
// TOOM-4 (fx_toom4<FX>(): FX = 16, 24, 32, 40, 48, i.e. --precision 512, 768, 1024, 1280, 1536):
// a' = a0 + a1 b + a2 b^2 + a3 b^3 with pieces of w = 32 M2 - 4 bits (b = 2^w, M2 = FX/4) is
// evaluated at the seven points 0, 1, -1, 2, -2, 1/2, inf; the product polynomial of a row pair has
// seven coefficients, so SEVEN products of M2 x M2 limbs per row pair (7/16 of the plain product)
// replace the nine of the two Karatsuba levels.  The evaluations are linear, hence the sums over the
// rows of the seven products ARE the evaluations of sum_r a_r(x) b_r(x), and the interpolation (exact
// integer divisions by 2, 3, 4, 9, 15: GMP's toom_interpolate_7pts sequence) runs once per output
// element after the row loop (k_syrk4_finish).  Every stored piece is a non-negative integer below
// 2^(32 M2): the evaluations at -1 and -2 are stored with a bias (K1 = 2 b, K2 = 10 b) that the
// finish kernel removes exactly with the column sums.  The four spare bits per piece are the
// headroom of p(2) < 15 b, which costs 17 bits of the image: FB = 4 w - 1 = 32 FX - 17 (495
// fraction bits at --precision 512, where two Karatsuba levels keep 505 and the reference 512).
// Image: seven M2-limb pieces per element, piece-major like the two-level image, group g =
//   0: a0   1: p(1)   2: p(-1) + K1   3: p(2)   4: p(-2) + K2   5: 8 p(1/2)   6: a3.
// Limbs of the fixed-point image for an NL-limb mantissa: GMP's rounded precision 64 (l - 1) = 32 (NL - 2) bits
// (compute_Q.cxx:107), rounded UP to a multiple of four limbs from 14 limbs on, so that every precision from
// sdpb's default 400 bits upwards takes the Toom-4 kernel (400/448 bits: 16 instead of 14 limbs, 640-704 bits:
// 24 instead of 22).  The image then holds more fraction bits than the reference's (495 against 448, 751 against
// 704) at 7/16 of 16^2 = 112 instead of 3/4 of 14^2 = 147 limb products per row pair.
template <int NL> constexpr int fx_limbs() { return (NL - 2 >= 14 && (NL - 2) % 4 != 0) ? ((NL - 2 + 3) / 4) * 4 : NL - 2; }
#ifndef SDPB_TOOM4_MAX_FX
#define SDPB_TOOM4_MAX_FX 48
#endif
template <int FX> constexpr bool fx_toom4()
{
#if defined(SDPB_SYRK_ONE_LEVEL) || defined(SDPB_SYRK_NO_TOOM4)
  return false;
#else
  return FX % 4 == 0 && FX >= 16 && FX <= SDPB_TOOM4_MAX_FX; // 512 bits and up: below, 17 bits are too large a share of the image
#endif
}
// TOOM-4 x KARATSUBA (fx_toom4k<FX>(): FX = 16, 24, 32, i.e. --precision 400 ... 1024): each of the seven evaluated pieces
// e < 2^(32 M2 - 2) is split once more, e = e_lo + e_hi 2^H with halves of H = 16 M2 - 1 bits and e_mid = e_lo + e_hi
// < 2^(16 M2), so a row pair costs 21 products of M3 x M3 limbs (M3 = FX/8): 84 limb products at FX = 16 instead of the
// 112 of Toom-4 alone.  The two spare bits per evaluated piece come out of the piece width: w = 32 M2 - 6, hence
// FB = 4 w - 1 = 32 FX - 25 (487 fraction bits at --precision 512, 743 at 768, 999 at 1024).  Image: 21 M3-limb pieces per element, piece-major,
// group 3 g + u with g the Toom-4 group above and u in (lo, hi, mid); k_syrk_fx3 sums the 21 products over the rows and
// k_syrk4_finish recombines the three of a Toom-4 group (e e' = lo lo' + (mid mid' - lo lo' - hi hi') 2^H + hi hi' 2^(2H))
// before it interpolates, so everything after the product sums is shared with k_syrk_fx2<.., true>.
template <int FX> constexpr bool fx_toom4k()
{
#if defined(SDPB_SYRK_NO_TOOM4K)
  return false;
#else
  return fx_toom4<FX>() && (FX == 16 || FX == 24 || FX == 32); // M3 = 2, 3, 4 limbs per piece (above, the 4 x (2 M3 - 1) x 3 accumulator registers of a lane no longer fit)
#endif
}
// TOOM-5 x KARATSUBA WITH LAZY CARRIES (fx_toom5k<FX>(): FX = 16, i.e. --precision 400 ... 512; round 6).  v_mad_u64_u32 alone
// issues every 4.2 cycles, the pair with v_addc_co_u32 that a 96-bit column accumulator needs every 8.2
// (profiles/r06d_ubench_mad_only.txt) -- and the carry instruction is only there because a product of two 32-bit limbs fills
// 64 bits.  With 28-bit limbs a limb product is < 2^56 and a column of a 2 x 2-limb product can absorb 64 rows before it
// is carried: the row loop is multiply-adds and nothing else.  To keep the image wide with narrower pieces the row
// polynomial gets FIVE pieces instead of four: a' = a0 + a1 b + ... + a4 b^4, pieces of w = 102 bits, evaluated at the nine
// points 0, 1, -1, 2, -2, 1/2, -1/2, 3, inf (every value < 121 b < 2^109; the three signed points stored with a bias
// K = 2 b, 10 b, 10 b that the column sums remove exactly, as in Toom-4), each evaluated value split once more
// (Karatsuba: halves of H = 55 bits and their sum, < 2^56 = two 28-bit limbs): 27 products of 2 x 2 limbs per row pair =
// 108 multiply-adds of ONE instruction each, against the 84 pairs = 168 instructions of Toom-4 x Karatsuba -- and
// FB = 5 w - 1 = 509 fraction bits, where Toom-4 x Karatsuba keeps 487 and the reference 512.  The interpolation (a 9 x 9
// integer matrix with exact divisions, derived in profiles/tools/toom5_matrix.py) runs once per output element in
// k_syrk5_finish.  Image: 27 pieces of two words (limbs < 2^28), group 3 g + u, g the point, u in (lo, hi, mid).
template <int FX> constexpr bool fx_toom5k()
{
#if defined(SDPB_SYRK_NO_TOOM5K)
  return false;
#else
  return fx_toom4k<FX>() && FX == 16;
#endif
}
constexpr int T5_LB = 28;                         // bits per limb of a Toom-5 half-piece
constexpr uint32_t T5_MASK = (1u << T5_LB) - 1u;
constexpr int T5_H = 55;                          // bits per half of an evaluated piece
// products per row pair, limbs of one product's sum over a row split
template <int FX> constexpr int fx_nprod() { return fx_toom5k<FX>() ? 27 : 21; }
template <int FX> constexpr int fx_part_limbs() { return fx_toom5k<FX>() ? 5 : 2 * (FX / 8) + 1; } // (2^112 per row, < 2^32 rows)
// bits per Toom piece
template <int FX> constexpr int toom_wb() { return fx_toom5k<FX>() ? 102 : 32 * (FX / 4) - (fx_toom4k<FX>() ? 6 : 4); }
template <int FX> constexpr int fx_planes()
{
  return fx_toom5k<FX>() ? 27 * 2 : fx_toom4k<FX>() ? 21 * (FX / 8) : fx_toom4<FX>() ? 7 * (FX / 4) : fx_two_level<FX>() ? 9 * (FX / 4) : 3 * (FX / 2);
}
template <int FX> constexpr int fx_frac_bits()
{
  return fx_toom5k<FX>() ? 5 * toom_wb<FX>() - 1 : fx_toom4<FX>() ? 4 * toom_wb<FX>() - 1 : fx_two_level<FX>() ? 32 * FX - 7 : 32 * FX - 3;
}
// elements per group plane of the image of a rows x cols operand: k_syrk_fx3 stages whole blocks of `rb` rows and
// whole 32-column tiles without bounds checks, so its image is padded (the pad is zeroed once, when the image is allocated)
template <int FX> constexpr size_t fx_image_stride(size_t rows, size_t cols, int rb)
{
  return fx_toom4k<FX>() ? (rows + rb - 1) / rb * rb * cols + 64 : rows * cols;
}
// edge of the output tiles of the syrk kernel in use: k_syrk_fx3 gives a lane 2 x 2 outputs
template <int FX> constexpr int syrk_tile_edge() { return fx_toom4k<FX>() ? 32 : 16; }

// out = (x >> BIT0) mod 2^NB as OUT limbs (compile-time positions)
template <int BIT0, int NB, int W, int OUT> MW_HD void bits_slice(const uint32_t (&x)[W], uint32_t (&out)[OUT])
{
  constexpr int q = BIT0 / 32, r = BIT0 % 32;
#pragma unroll
  for(int i = 0; i < OUT; ++i)
    {
      const uint32_t lo = (q + i < W) ? x[q + i < W ? q + i : 0] : 0u;
      const uint32_t hi = (q + i + 1 < W) ? x[q + i + 1 < W ? q + i + 1 : 0] : 0u;
      uint32_t v = r ? ((lo >> r) | (hi << ((32 - r) & 31))) : lo;
      const int left = NB - 32 * i;
      if(left <= 0)
        v = 0;
      else if(left < 32)
        v &= (1u << (left & 31)) - 1u;
      out[i] = v;
    }
}

// add x << SH (x an A-limb unsigned integer) into the W-limb integer w; negate: subtract
template <int W, int A> MW_HD void add_shifted(uint32_t (&w)[W], const uint32_t (&x)[A], int sh, bool negate)
{
  const int q = sh >> 5, r = sh & 31;
  uint64_t cy = negate ? 1u : 0u;
  const uint32_t mask = negate ? 0xffffffffu : 0u;
#pragma unroll
  for(int k = 0; k < W; ++k)
    {
      const int i0 = k - q;
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for(int t = 0; t < A; ++t)
        {
          lo = (t == i0 - 1) ? x[t] : lo;
          hi = (t == i0) ? x[t] : hi;
        }
      const uint32_t limb = r ? ((hi << r) | (lo >> (32 - r))) : hi;
      const uint64_t s = (uint64_t)w[k] + (uint64_t)(limb ^ mask) + cy;
      w[k] = (uint32_t)s;
      cy = s >> 32;
    }
}

// write the 3M planes of one element from sign + FX-limb magnitude (|v| < 2^FB)
template <int FX> MW_HD void fx_store2(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx);
template <int FX> MW_HD void fx_store4(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx);
template <int FX> MW_HD void fx_store5(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx);
template <int FX> MW_HD void fx_store(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx)
{
  constexpr int M = FX / 2;
  static_assert(FX % 2 == 0 && FX >= 4, "FX = NL - 2 is even");
  if constexpr(fx_toom5k<FX>())
    {
      fx_store5<FX>(mag, negative, fx, fx_stride, idx);
      return;
    }
  else if constexpr(fx_toom4<FX>())
    {
      fx_store4<FX>(mag, negative, fx, fx_stride, idx);
      return;
    }
  else if constexpr(fx_two_level<FX>())
    {
      fx_store2<FX>(mag, negative, fx, fx_stride, idx);
      return;
    }
  // a' = C +/- |v|, C = bit 29 of the top limb
  uint32_t a[FX];
  uint64_t borrow = 0;
#pragma unroll
  for(int i = 0; i < FX; ++i)
    {
      const uint32_t c = (i == FX - 1) ? (1u << 29) : 0u;
      if(negative)
        {
          const uint64_t d = (uint64_t)c - (uint64_t)mag[i] - borrow;
          a[i] = (uint32_t)d;
          borrow = (d >> 63) & 1u;
        }
      else
        a[i] = mag[i] | c;
    }
  uint32_t lo[M], hi[M];
#pragma unroll
  for(int i = 0; i < M; ++i)
    {
      lo[i] = (i == M - 1) ? (a[i] & 0x7fffffffu) : a[i];
      const uint32_t up = (M + i < FX) ? a[M + i < FX ? M + i : 0] : 0u;
      hi[i] = (a[M - 1 + i] >> 31) | (up << 1);
    }
  uint64_t cy = 0;
#pragma unroll
  for(int i = 0; i < M; ++i)
    {
      const uint64_t s = (uint64_t)lo[i] + hi[i] + cy;
      cy = s >> 32;
      fx[(size_t)i * fx_stride + idx] = lo[i];
      fx[(size_t)(M + i) * fx_stride + idx] = hi[i];
      fx[(size_t)(2 * M + i) * fx_stride + idx] = (uint32_t)s;
    }
}

// two-level image of one element (|v| < 2^FB, FB = 32 FX - 7)
template <int FX> MW_HD void fx_store2(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx)
{
  constexpr int M = FX / 2, M2 = FX / 4, FB = fx_frac_bits<FX>();
  static_assert(FB / 32 == FX - 1, "the bias bit sits in the top limb");
  uint32_t a[FX];
  uint64_t borrow = 0;
#pragma unroll
  for(int i = 0; i < FX; ++i)
    {
      const uint32_t c = (i == FX - 1) ? (1u << (FB % 32)) : 0u;
      if(negative)
        {
          const uint64_t d = (uint64_t)c - (uint64_t)mag[i] - borrow;
          a[i] = (uint32_t)d;
          borrow = (d >> 63) & 1u;
        }
      else
        a[i] = mag[i] | c;
    }
  uint32_t x[3][M];
  bits_slice<0, 32 * M - 3>(a, x[0]);
  bits_slice<32 * M - 3, 32 * M - 3>(a, x[1]);
  {
    uint64_t cy = 0;
#pragma unroll
    for(int i = 0; i < M; ++i)
      {
        const uint64_t t = (uint64_t)x[0][i] + x[1][i] + cy;
        x[2][i] = (uint32_t)t;
        cy = t >> 32;
      }
  }
#pragma unroll
  for(int k = 0; k < 3; ++k)
    {
      uint32_t p0[M2], p1[M2];
      bits_slice<0, 32 * M2 - 1>(x[k], p0);
      bits_slice<32 * M2 - 1, 32 * M2 - 1>(x[k], p1);
      uint32_t *o0 = fx + ((size_t)(3 * k) * fx_stride + idx) * M2, *o1 = fx + ((size_t)(3 * k + 1) * fx_stride + idx) * M2,
               *ot = fx + ((size_t)(3 * k + 2) * fx_stride + idx) * M2;
      uint64_t cy = 0;
#pragma unroll
      for(int i = 0; i < M2; ++i)
        {
          const uint64_t t = (uint64_t)p0[i] + p1[i] + cy;
          cy = t >> 32;
          o0[i] = p0[i];
          o1[i] = p1[i];
          ot[i] = (uint32_t)t;
        }
    }
}

// one M2-limb piece with the widest loads its size allows (pieces are M2*4-byte aligned)
struct alignas(16) PieceQuad
{
  uint32_t w[4];
};
struct alignas(8) PiecePair
{
  uint32_t w[2];
};
template <int M2> MW_HD void piece_load(const uint32_t *p, uint32_t (&x)[M2])
{
  if constexpr(M2 % 4 == 0)
    {
#pragma unroll
      for(int q = 0; q < M2 / 4; ++q)
        {
          const PieceQuad v = *reinterpret_cast<const PieceQuad *>(p + 4 * q);
#pragma unroll
          for(int l = 0; l < 4; ++l)
            x[4 * q + l] = v.w[l];
        }
    }
  else if constexpr(M2 % 2 == 0)
    {
#pragma unroll
      for(int q = 0; q < M2 / 2; ++q)
        {
          const PiecePair v = *reinterpret_cast<const PiecePair *>(p + 2 * q);
          x[2 * q] = v.w[0];
          x[2 * q + 1] = v.w[1];
        }
    }
  else
    {
#pragma unroll
      for(int l = 0; l < M2; ++l)
        x[l] = p[l];
    }
}
template <int M2> MW_HD void piece_store(uint32_t *p, const uint32_t (&x)[M2])
{
  if constexpr(M2 % 4 == 0)
    {
#pragma unroll
      for(int q = 0; q < M2 / 4; ++q)
        {
          PieceQuad v;
#pragma unroll
          for(int l = 0; l < 4; ++l)
            v.w[l] = x[4 * q + l];
          *reinterpret_cast<PieceQuad *>(p + 4 * q) = v;
        }
    }
  else if constexpr(M2 % 2 == 0)
    {
#pragma unroll
      for(int q = 0; q < M2 / 2; ++q)
        {
          PiecePair v;
          v.w[0] = x[2 * q];
          v.w[1] = x[2 * q + 1];
          *reinterpret_cast<PiecePair *>(p + 2 * q) = v;
        }
    }
  else
    {
#pragma unroll
      for(int l = 0; l < M2; ++l)
        p[l] = x[l];
    }
}

// d -= x (both A limbs, d >= x)
template <int A> MW_HD void sub_limbs(uint32_t (&d)[A], const uint32_t (&x)[A])
{
  uint64_t bw = 0;
#pragma unroll
  for(int k = 0; k < A; ++k)
    {
      const uint64_t t = (uint64_t)d[k] - (uint64_t)x[k] - bw;
      d[k] = (uint32_t)t;
      bw = (t >> 63) & 1u;
    }
}

// ---- Toom-5 x Karatsuba image (fx_toom5k) ---------------------------------------------------------------------------
// out = sum_k c_k p_k for small non-negative c_k (< 2^7: no overflow of the ML-limb result by construction)
template <int ML> MW_HD void toom5_lin(uint32_t (&out)[ML], const uint32_t (&p)[5][ML], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4)
{
  uint64_t cy = 0;
#pragma unroll
  for(int i = 0; i < ML; ++i)
    {
      const uint64_t t = (uint64_t)p[0][i] * c0 + (uint64_t)p[1][i] * c1 + (uint64_t)p[2][i] * c2 + (uint64_t)p[3][i] * c3 + (uint64_t)p[4][i] * c4 + cy;
      out[i] = (uint32_t)t;
      cy = t >> 32;
    }
}
// bits [POS, POS + NB) of x (NB <= 32, compile-time position)
template <int POS, int NB, int ML> MW_HD uint32_t toom5_bits(const uint32_t (&x)[ML])
{
  constexpr int q = POS / 32, r = POS % 32;
  const uint32_t lo = q < ML ? x[q < ML ? q : 0] : 0u, hi = q + 1 < ML ? x[q + 1 < ML ? q + 1 : 0] : 0u;
  const uint32_t v = r ? ((lo >> r) | (hi << (32 - r))) : lo;
  return NB < 32 ? (v & ((1u << (NB < 32 ? NB : 0)) - 1u)) : v;
}
// The points' order in the image: 0: a0, 1: p(1), 2: p(-1) + 2 b, 3: p(2), 4: p(-2) + 10 b, 5: 16 p(1/2), 6: 16 p(-1/2) + 10 b, 7: p(3), 8: a4
template <int FX> MW_HD void fx_store5(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx)
{
  constexpr int WB = toom_wb<FX>(), FB = fx_frac_bits<FX>(), ML = (WB + 7 + 31) / 32, H = T5_H, LB = T5_LB;
  constexpr int TOPBIT = WB - 32 * (ML - 1); // bit of b = 2^WB inside the top limb of an evaluated piece
  static_assert(FB == 5 * WB - 1 && FB / 32 == FX - 1, "a' = v + 2^FB fills five pieces of WB bits");
  static_assert(TOPBIT >= 0 && TOPBIT + 7 <= 32 && WB + 7 <= 2 * H && H + 1 <= 2 * LB, "121 b fits, the halves fit two limbs");
  uint32_t a[FX];
  uint64_t borrow = 0;
#pragma unroll
  for(int i = 0; i < FX; ++i)
    {
      const uint32_t c = (i == FX - 1) ? (1u << (FB % 32)) : 0u;
      if(negative)
        {
          const uint64_t d = (uint64_t)c - (uint64_t)mag[i] - borrow;
          a[i] = (uint32_t)d;
          borrow = (d >> 63) & 1u;
        }
      else
        a[i] = mag[i] | c;
    }
  uint32_t p[5][ML];
  bits_slice<0, WB>(a, p[0]);
  bits_slice<WB, WB>(a, p[1]);
  bits_slice<2 * WB, WB>(a, p[2]);
  bits_slice<3 * WB, WB>(a, p[3]);
  bits_slice<4 * WB, WB>(a, p[4]);
  uint32_t e[9][ML];
#pragma unroll
  for(int i = 0; i < ML; ++i)
    {
      e[0][i] = p[0][i];
      e[8][i] = p[4][i];
    }
  toom5_lin<ML>(e[1], p, 1, 1, 1, 1, 1);
  toom5_lin<ML>(e[3], p, 1, 2, 4, 8, 16);
  toom5_lin<ML>(e[5], p, 16, 8, 4, 2, 1);
  toom5_lin<ML>(e[7], p, 1, 3, 9, 27, 81);
  {
    // the signed points: positive part (+ bias) minus negative part, never below zero
    uint32_t pos[ML], neg[ML];
    auto diff = [&](uint32_t (&out)[ML], uint32_t bias) {
      pos[ML - 1] += bias << TOPBIT;
      uint64_t bw = 0;
#pragma unroll
      for(int i = 0; i < ML; ++i)
        {
          const uint64_t t = (uint64_t)pos[i] - (uint64_t)neg[i] - bw;
          out[i] = (uint32_t)t;
          bw = (t >> 63) & 1u;
        }
    };
    toom5_lin<ML>(pos, p, 1, 0, 1, 0, 1);
    toom5_lin<ML>(neg, p, 0, 1, 0, 1, 0);
    diff(e[2], 2u); // p(-1) + 2 b
    toom5_lin<ML>(pos, p, 1, 0, 4, 0, 16);
    toom5_lin<ML>(neg, p, 0, 2, 0, 8, 0);
    diff(e[4], 10u); // p(-2) + 10 b
    toom5_lin<ML>(pos, p, 16, 0, 4, 0, 1);
    toom5_lin<ML>(neg, p, 0, 8, 0, 2, 0);
    diff(e[6], 10u); // 16 p(-1/2) + 10 b
  }
#pragma unroll
  for(int g = 0; g < 9; ++g)
    {
      // halves of H bits as two limbs of LB bits each, and their sum
      uint32_t lo[2], hi[2], mid[2];
      lo[0] = toom5_bits<0, LB>(e[g]);
      lo[1] = toom5_bits<LB, H - LB>(e[g]);
      hi[0] = toom5_bits<H, LB>(e[g]);
      hi[1] = toom5_bits<H + LB, H - LB>(e[g]);
      const uint32_t s0 = lo[0] + hi[0];
      mid[0] = s0 & T5_MASK;
      mid[1] = lo[1] + hi[1] + (s0 >> LB);
      piece_store<2>(fx + ((size_t)(3 * g + 0) * fx_stride + idx) * 2, lo);
      piece_store<2>(fx + ((size_t)(3 * g + 1) * fx_stride + idx) * 2, hi);
      piece_store<2>(fx + ((size_t)(3 * g + 2) * fx_stride + idx) * 2, mid);
    }
}

// ---- Toom-4 image and the signed multi-limb helpers of its interpolation ------------------
// out = sum_k c_k p_k for small non-negative c_k (no overflow by construction: < 2^(32 M2))
template <int M2> MW_HD void toom_lin(uint32_t (&out)[M2], const uint32_t (&p)[4][M2], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
  uint64_t cy = 0;
#pragma unroll
  for(int i = 0; i < M2; ++i)
    {
      // four 32 x 4-bit products and a carry: below 2^38
      const uint64_t t = (uint64_t)p[0][i] * c0 + (uint64_t)p[1][i] * c1 + (uint64_t)p[2][i] * c2 + (uint64_t)p[3][i] * c3 + cy;
      out[i] = (uint32_t)t;
      cy = t >> 32;
    }
}
// seven-piece image of one element (|v| < 2^FB, FB = 32 FX - 17)
template <int FX> MW_HD void fx_store4(const uint32_t (&mag)[FX], bool negative, uint32_t *fx, size_t fx_stride, size_t idx)
{
  constexpr int M2 = FX / 4, WB = toom_wb<FX>(), FB = fx_frac_bits<FX>();
  constexpr int TOPBIT = WB - 32 * (M2 - 1); // bit of b = 2^WB inside the top limb of a piece
  static_assert(FB == 4 * WB - 1 && FB / 32 == FX - 1, "a' = v + 2^FB fills four pieces of WB bits");
  static_assert(TOPBIT >= 0 && TOPBIT + 4 <= 32, "15 b fits the top limb");
  uint32_t a[FX];
  uint64_t borrow = 0;
#pragma unroll
  for(int i = 0; i < FX; ++i)
    {
      const uint32_t c = (i == FX - 1) ? (1u << (FB % 32)) : 0u;
      if(negative)
        {
          const uint64_t d = (uint64_t)c - (uint64_t)mag[i] - borrow;
          a[i] = (uint32_t)d;
          borrow = (d >> 63) & 1u;
        }
      else
        a[i] = mag[i] | c;
    }
  uint32_t p[4][M2];
  bits_slice<0, WB>(a, p[0]);
  bits_slice<WB, WB>(a, p[1]);
  bits_slice<2 * WB, WB>(a, p[2]);
  bits_slice<3 * WB, WB>(a, p[3]);
  uint32_t e[7][M2];
#pragma unroll
  for(int i = 0; i < M2; ++i)
    {
      e[0][i] = p[0][i];
      e[6][i] = p[3][i];
    }
  toom_lin<M2>(e[1], p, 1, 1, 1, 1);
  toom_lin<M2>(e[3], p, 1, 2, 4, 8);
  toom_lin<M2>(e[5], p, 8, 4, 2, 1);
  {
    // p(-1) + 2 b and p(-2) + 10 b: positive part (+ bias) minus negative part, never below zero
    uint32_t pos[M2], neg[M2];
    toom_lin<M2>(pos, p, 1, 0, 1, 0);
    toom_lin<M2>(neg, p, 0, 1, 0, 1);
    pos[M2 - 1] += 2u << TOPBIT; // 2 b = 2^(WB+1)
    uint64_t bw = 0;
#pragma unroll
    for(int i = 0; i < M2; ++i)
      {
        const uint64_t t = (uint64_t)pos[i] - (uint64_t)neg[i] - bw;
        e[2][i] = (uint32_t)t;
        bw = (t >> 63) & 1u;
      }
    toom_lin<M2>(pos, p, 1, 0, 4, 0);
    toom_lin<M2>(neg, p, 0, 2, 0, 8);
    pos[M2 - 1] += 10u << TOPBIT; // 10 b
    bw = 0;
#pragma unroll
    for(int i = 0; i < M2; ++i)
      {
        const uint64_t t = (uint64_t)pos[i] - (uint64_t)neg[i] - bw;
        e[4][i] = (uint32_t)t;
        bw = (t >> 63) & 1u;
      }
  }
  if constexpr(fx_toom4k<FX>())
    {
      // every e[g] < 15 b < 2^(32 M2 - 2): halves of H = 16 M2 - 1 bits and their sum, M3 limbs each
      constexpr int M3 = FX / 8, H = 16 * M2 - 1;
#pragma unroll
      for(int g = 0; g < 7; ++g)
        {
          uint32_t lo[M3], hi[M3], mid[M3];
          bits_slice<0, H>(e[g], lo);
          bits_slice<H, H>(e[g], hi);
          uint64_t cy = 0;
#pragma unroll
          for(int i = 0; i < M3; ++i)
            {
              const uint64_t t = (uint64_t)lo[i] + hi[i] + cy;
              mid[i] = (uint32_t)t;
              cy = t >> 32;
            }
          piece_store<M3>(fx + ((size_t)(3 * g + 0) * fx_stride + idx) * M3, lo);
          piece_store<M3>(fx + ((size_t)(3 * g + 1) * fx_stride + idx) * M3, hi);
          piece_store<M3>(fx + ((size_t)(3 * g + 2) * fx_stride + idx) * M3, mid);
        }
    }
  else
    {
#pragma unroll
      for(int g = 0; g < 7; ++g)
        {
          uint32_t *o = fx + ((size_t)g * fx_stride + idx) * M2;
#pragma unroll
          for(int i = 0; i < M2; ++i)
            o[i] = e[g][i];
        }
    }
}
// the M2-limb evaluated piece of Toom-4 group g of element e (either image)
template <int FX> MW_HD void toom_piece_load(const uint32_t *fx, size_t fx_stride, int g, size_t e, uint32_t (&x)[FX / 4])
{
  constexpr int M2 = FX / 4;
  if constexpr(fx_toom4k<FX>())
    {
      constexpr int M3 = FX / 8, H = 16 * M2 - 1;
      uint32_t lo[M2], lo3[M3], hi[M3];
      piece_load<M3>(fx + ((size_t)(3 * g + 0) * fx_stride + e) * M3, lo3);
      piece_load<M3>(fx + ((size_t)(3 * g + 1) * fx_stride + e) * M3, hi);
#pragma unroll
      for(int i = 0; i < M2; ++i)
        lo[i] = i < M3 ? lo3[i < M3 ? i : 0] : 0u;
      add_shifted<M2, M3>(lo, hi, H, false);
#pragma unroll
      for(int i = 0; i < M2; ++i)
        x[i] = lo[i];
    }
  else
    {
      const uint32_t *src = fx + ((size_t)g * fx_stride + e) * M2;
#pragma unroll
      for(int i = 0; i < M2; ++i)
        x[i] = src[i];
    }
}
// Z-limb two's-complement integers (Z = 2 M2 + 2: a sum over < 2^32 rows of products of two
// M2-limb pieces, times the small factors of the interpolation, with a sign)
template <int Z> MW_HD void z_add(uint32_t (&d)[Z], const uint32_t (&x)[Z])
{
  uint64_t cy = 0;
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      const uint64_t t = (uint64_t)d[k] + x[k] + cy;
      d[k] = (uint32_t)t;
      cy = t >> 32;
    }
}
template <int Z> MW_HD void z_sub(uint32_t (&d)[Z], const uint32_t (&x)[Z])
{
  uint64_t bw = 0;
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      const uint64_t t = (uint64_t)d[k] - (uint64_t)x[k] - bw;
      d[k] = (uint32_t)t;
      bw = (t >> 63) & 1u;
    }
}
// d += c x  /  d -= c x  for a small constant c (two's complement in, two's complement out)
template <int Z> MW_HD void z_addmul(uint32_t (&d)[Z], const uint32_t (&x)[Z], uint32_t c, bool subtract)
{
  uint32_t t[Z];
  uint64_t cy = 0;
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      const uint64_t u = (uint64_t)x[k] * c + cy;
      t[k] = (uint32_t)u;
      cy = u >> 32;
    }
  if(subtract)
    z_sub<Z>(d, t);
  else
    z_add<Z>(d, t);
}
template <int Z> MW_HD void z_sar(uint32_t (&d)[Z], int bits) // exact halvings: arithmetic shift right by 1 or 2
{
  const uint32_t fill = 0u - (d[Z - 1] >> 31);
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      const uint32_t up = k + 1 < Z ? d[k + 1 < Z ? k + 1 : 0] : fill;
      d[k] = (d[k] >> bits) | (up << (32 - bits));
    }
}
template <int Z> MW_HD void z_shl(uint32_t (&d)[Z], int sh) // d <<= sh (0 <= sh < 32 Z)
{
  const int q = sh >> 5, r = sh & 31;
  uint32_t t[Z];
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for(int u = 0; u < Z; ++u)
        {
          hi = (u == k - q) ? d[u] : hi;
          lo = (u == k - q - 1) ? d[u] : lo;
        }
      t[k] = r ? ((hi << r) | (lo >> (32 - r))) : hi;
    }
#pragma unroll
  for(int k = 0; k < Z; ++k)
    d[k] = t[k];
}
// d /= c for an odd constant c that divides d exactly (c^-1 mod 2^32 given): the quotient of a
// two's-complement value comes out in two's complement (arithmetic modulo 2^(32 Z))
template <int Z> MW_HD void z_divexact(uint32_t (&d)[Z], uint32_t c, uint32_t cinv)
{
  uint32_t carry = 0;
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      const uint32_t b = d[k] < carry ? 1u : 0u;
      const uint32_t q = (d[k] - carry) * cinv;
      d[k] = q;
      carry = (uint32_t)(((uint64_t)q * c) >> 32) + b;
    }
}
// The interpolation of GMP's mpn_toom_interpolate_7pts (points 0, -2, 1, -1, 2, 1/2 (x 64), inf in
// w[0..6]); on return w[k] is coefficient k of the product polynomial.  Checked against plain
// integer polynomial products in tests (bit-exact syrk vs GMP mpz).
template <int Z> MW_HD void toom4_interpolate(uint32_t (&w)[7][Z])
{
  z_add<Z>(w[5], w[4]);                 // W5 += W4
  {                                     // W1 = (W4 - W1) / 2
    uint32_t t[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      t[k] = w[4][k];
    z_sub<Z>(t, w[1]);
    z_sar<Z>(t, 1);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      w[1][k] = t[k];
  }
  z_sub<Z>(w[4], w[0]);                 // W4 -= W0
  z_sub<Z>(w[4], w[1]);                 // W4 = (W4 - W1) / 4 - 16 W6
  z_sar<Z>(w[4], 2);
  z_addmul<Z>(w[4], w[6], 16, true);
  {                                     // W3 = (W2 - W3) / 2
    uint32_t t[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      t[k] = w[2][k];
    z_sub<Z>(t, w[3]);
    z_sar<Z>(t, 1);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      w[3][k] = t[k];
  }
  z_sub<Z>(w[2], w[3]);                 // W2 -= W3
  z_addmul<Z>(w[5], w[2], 65, true);    // W5 -= 65 W2
  z_sub<Z>(w[2], w[6]);                 // W2 -= W6 + W0
  z_sub<Z>(w[2], w[0]);
  z_addmul<Z>(w[5], w[2], 45, false);   // W5 = (W5 + 45 W2) / 2
  z_sar<Z>(w[5], 1);
  z_sub<Z>(w[4], w[2]);                 // W4 = (W4 - W2) / 3
  z_divexact<Z>(w[4], 3, 0xAAAAAAABu);
  z_sub<Z>(w[2], w[4]);                 // W2 -= W4
  {                                     // W1 = W5 - W1
    uint32_t t[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      t[k] = w[5][k];
    z_sub<Z>(t, w[1]);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      w[1][k] = t[k];
  }
  z_addmul<Z>(w[5], w[3], 8, true);     // W5 = (W5 - 8 W3) / 9
  z_divexact<Z>(w[5], 9, 0x38E38E39u);
  z_sub<Z>(w[3], w[5]);                 // W3 -= W5
  z_divexact<Z>(w[1], 15, 0xEEEEEEEFu); // W1 = (W1 / 15 + W5) / 2
  z_add<Z>(w[1], w[5]);
  z_sar<Z>(w[1], 1);
  z_sub<Z>(w[5], w[1]);                 // W5 -= W1
}

// fx[idx] = image of trunc(PT[idx] * inv_norm[idx % N] * 2^FB)
template <int NL, int FX>
__global__ void __launch_bounds__(WG) k_normalize_fx(mw::CPtr PT, size_t count, int N, mw::CPtr inv_norm, uint32_t *fx, size_t fx_stride)
{
  constexpr int FB = fx_frac_bits<FX>();
  for(size_t idx = (size_t)blockIdx.x * WG + threadIdx.x; idx < count; idx += (size_t)gridDim.x * WG)
    {
      const Mw<NL> t = mw::mul(mw::load<NL>(PT, idx), mw::load<NL>(inv_norm, idx % N));
      uint32_t w[NL];
      bool sat = false;
      if(mw::is_zero(t))
        {
#pragma unroll
          for(int i = 0; i < NL; ++i)
            w[i] = 0;
        }
      else
        {
#pragma unroll
          for(int i = 0; i < NL; ++i)
            w[i] = t.m[i];
          // integer = M * 2^(e + FB - 32NL): shift right by s
          const int s = 32 * NL - FB - t.e;
          if(t.e >= 1)
            sat = true; // |t| >= 1 (only t == 1 up to rounding): clamp to 2^FB - 1
          else if(s >= 32 * NL)
            {
#pragma unroll
              for(int i = 0; i < NL; ++i)
                w[i] = 0;
            }
          else
            {
              mw::shr_limbs<NL>(w, (uint32_t)s >> 5);
              mw::shr_bits<NL>(w, (uint32_t)s & 31u);
            }
        }
      uint32_t v[FX];
#pragma unroll
      for(int i = 0; i < FX; ++i)
        v[i] = sat ? ((i == FX - 1) ? ((1u << (FB % 32)) - 1u) : 0xffffffffu) : w[i]; // 2^FB - 1 for either image (FB = 32FX-3 or 32FX-7)
      fx_store<FX>(v, t.neg != 0, fx, fx_stride, idx);
    }
}

// the same image from staged planes: plane 0 = sign, planes 1..FX = |v| (sdpb_hip_op_int_syrk);
// in and out are distinct buffers
// (in_stride = elements per staged plane; `in` may point at a row window of the staged matrix)
template <int FX> __global__ void __launch_bounds__(WG) k_fx_from_int(const uint32_t *in, size_t in_stride, size_t count, uint32_t *fx, size_t fx_stride)
{
  const size_t idx = (size_t)blockIdx.x * WG + threadIdx.x;
  if(idx >= count)
    return;
  uint32_t v[FX];
#pragma unroll
  for(int i = 0; i < FX; ++i)
    v[i] = in[(size_t)(i + 1) * in_stride + idx];
  fx_store<FX>(v, in[idx] != 0, fx, fx_stride, idx);
}

// Column sums S_n = sum_r a'_rn, as (2FX+2)-limb integers stored behind the N x N
// outputs of the accumulator array (element N*N + n of every plane), so that they ride
// through the same cross-GPU reduction as G.  Stage 1: workgroup (x, y) sums row slice y
// of 64 columns (4 row phases per column, meeting in LDS) into partial[y]; stage 2 adds
// the slices.  partial element (slice, k, n) at (slice * 2(M+2) + k) * N + n; k < M+2:
// sum of lo, else sum of hi.
template <int FX>
__global__ void __launch_bounds__(WG) k_fx_colsum(const uint32_t *fx, size_t fx_stride, unsigned nrows, int N, unsigned rows_per_slice, uint32_t *partial)
{
  constexpr int M = FX / 2, A = M + 2;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), phase = threadIdx.x >> 6;
  const unsigned r_begin = blockIdx.y * rows_per_slice, r_end = (r_begin + rows_per_slice < nrows) ? r_begin + rows_per_slice : nrows;
  uint32_t sl[A], sh[A];
#pragma unroll
  for(int k = 0; k < A; ++k)
    sl[k] = sh[k] = 0;
  if(col < N)
    for(unsigned r = r_begin + phase; r < r_end; r += 4)
      {
        const size_t e = (size_t)r * N + col;
        uint64_t cl = 0, ch = 0;
#pragma unroll
        for(int k = 0; k < A; ++k)
          {
            const uint64_t l = (uint64_t)sl[k] + (k < M ? fx[(size_t)(k < M ? k : 0) * fx_stride + e] : 0u) + cl;
            const uint64_t h = (uint64_t)sh[k] + (k < M ? fx[(size_t)(M + (k < M ? k : 0)) * fx_stride + e] : 0u) + ch;
            sl[k] = (uint32_t)l;
            cl = l >> 32;
            sh[k] = (uint32_t)h;
            ch = h >> 32;
          }
      }
  __shared__ uint32_t sm[3 * 2 * A * 64];
  if(phase > 0)
    {
#pragma unroll
      for(int k = 0; k < A; ++k)
        {
          sm[((phase - 1) * 2 * A + k) * 64 + (threadIdx.x & 63)] = sl[k];
          sm[((phase - 1) * 2 * A + A + k) * 64 + (threadIdx.x & 63)] = sh[k];
        }
    }
  __syncthreads();
  if(phase == 0 && col < N)
    {
      for(int p = 0; p < 3; ++p)
        {
          uint64_t cl = 0, ch = 0;
#pragma unroll
          for(int k = 0; k < A; ++k)
            {
              const uint64_t l = (uint64_t)sl[k] + sm[(p * 2 * A + k) * 64 + threadIdx.x] + cl;
              const uint64_t h = (uint64_t)sh[k] + sm[(p * 2 * A + A + k) * 64 + threadIdx.x] + ch;
              sl[k] = (uint32_t)l;
              cl = l >> 32;
              sh[k] = (uint32_t)h;
              ch = h >> 32;
            }
        }
#pragma unroll
      for(int k = 0; k < A; ++k)
        {
          partial[((size_t)blockIdx.y * 2 * A + k) * N + col] = sl[k];
          partial[((size_t)blockIdx.y * 2 * A + A + k) * N + col] = sh[k];
        }
    }
}
template <int FX>
__global__ void __launch_bounds__(WG) k_fx_colsum_final(const uint32_t *partial, int nslices, int N, uint32_t *acc, size_t acc_stride)
{
  constexpr int M = FX / 2, A = M + 2, W = 2 * FX + 2;
  const int col = blockIdx.x * WG + threadIdx.x;
  if(col >= N)
    return;
  uint32_t sl[A], sh[A];
#pragma unroll
  for(int k = 0; k < A; ++k)
    sl[k] = sh[k] = 0;
  for(int s = 0; s < nslices; ++s)
    {
      uint64_t cl = 0, ch = 0;
#pragma unroll
      for(int k = 0; k < A; ++k)
        {
          const uint64_t l = (uint64_t)sl[k] + partial[((size_t)s * 2 * A + k) * N + col] + cl;
          const uint64_t h = (uint64_t)sh[k] + partial[((size_t)s * 2 * A + A + k) * N + col] + ch;
          sl[k] = (uint32_t)l;
          cl = l >> 32;
          sh[k] = (uint32_t)h;
          ch = h >> 32;
        }
    }
  uint32_t w[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    w[k] = k < A ? sl[k < A ? k : 0] : 0u;
  add_shifted<W, A>(w, sh, 32 * M - 1, false);
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + (size_t)N * N + col] = w[k];
}

// Lower-triangle tiles of an N x N output, enumerated super-block by super-block
// (8x8 tiles) so that consecutive entries share operand panels.
// split_col > 0: the tiles that hold outputs of the columns [0, split_col) come first (super-block order inside each
// group), *count_left = how many they are; the tiles that hold outputs of the columns [split_col, N) follow -- the two
// launches of the chunked Q' take the two parts of the list.  Where split_col is not a multiple of the tile edge, the
// tiles of the straddling tile column are in BOTH parts (their products are computed twice; each part's finishing
// kernel reads only its own columns).
// edge: 16, or 32 for k_syrk_fx3 (super-blocks of 4x4 tiles then cover the same 128 columns)
inline std::vector<uint32_t> syrk_tile_order(int N, int split_col = 0, int *count_left = nullptr, int edge = 16)
{
  int SB = edge == 16 ? 8 : 4;
  if(const char *env = std::getenv("SDPB_HIP_SYRK_SB")) // tuning knob: super-block edge in tiles
    SB = std::max(1, std::atoi(env));
  const int tiles = (N + edge - 1) / edge, nsb = (tiles + SB - 1) / SB;
  std::vector<uint32_t> all;
  for(int bi = 0; bi < nsb; ++bi)
    for(int bj = 0; bj <= bi; ++bj)
      for(int ti = bi * SB; ti < std::min(tiles, (bi + 1) * SB); ++ti)
        for(int tj = bj * SB; tj < std::min(tiles, (bj + 1) * SB); ++tj)
          if(tj <= ti)
            all.push_back((uint32_t)ti << 16 | (uint32_t)tj);
  if(split_col <= 0)
    {
      if(count_left)
        *count_left = (int)all.size();
      return all;
    }
  std::vector<uint32_t> out;
  for(uint32_t t : all)
    if((int)(t & 0xffffu) * edge < split_col)
      out.push_back(t);
  if(count_left)
    *count_left = (int)out.size();
  for(uint32_t t : all)
    if(((int)(t & 0xffffu) + 1) * edge > split_col)
      out.push_back(t);
  return out;
}

// all 2M-1 columns of one row's M x M limb product into column accumulators that
// persist across rows: c[K] (64 bits) + h[K] 2^64 collects column K
template <int M, int K> struct SyrkColumns
{
  static MW_HD void run(const uint32_t (&a)[M], const uint32_t (&b)[M], uint64_t (&c)[2 * M - 1], uint32_t (&h)[2 * M - 1])
  {
    constexpr int I0 = K - (M - 1) > 0 ? K - (M - 1) : 0, I1 = K < M - 1 ? K : M - 1;
    mw::mac_column<I0, I1, K>(c[K], h[K], a, b);
    if constexpr(K < 2 * M - 2)
      SyrkColumns<M, K + 1>::run(a, b, c, h);
  }
};
// acc += sum_K (c[K] + h[K] 2^64) 2^(32K): three carry chains (low words, high words, overflows)
template <int M, int A> MW_HD void syrk_fold(uint32_t (&acc)[A], const uint64_t (&c)[2 * M - 1], const uint32_t (&h)[2 * M - 1])
{
  static_assert(A >= 2 * M + 1, "column 2M-2 reaches limb 2M");
  constexpr int NC = 2 * M - 1;
  const uint32_t zero = 0;
  mw::Carry cy;
  {
    const uint32_t x = (uint32_t)c[0];
    MW_ADD_CO(acc[0], x, cy);
  }
#pragma unroll
  for(int k = 1; k < A; ++k)
    {
      const uint32_t x = k < NC ? (uint32_t)c[k < NC ? k : 0] : zero;
      MW_ADDC(acc[k], x, cy);
    }
  {
    const uint32_t x = (uint32_t)(c[0] >> 32);
    MW_ADD_CO(acc[1], x, cy);
  }
#pragma unroll
  for(int k = 2; k < A; ++k)
    {
      const uint32_t x = k - 1 < NC ? (uint32_t)(c[k - 1 < NC ? k - 1 : 0] >> 32) : zero;
      MW_ADDC(acc[k], x, cy);
    }
  MW_ADD_CO(acc[2], h[0], cy);
#pragma unroll
  for(int k = 3; k < A; ++k)
    {
      const uint32_t x = k - 2 < NC ? h[k - 2 < NC ? k - 2 : 0] : zero;
      MW_ADDC(acc[k], x, cy);
    }
}
// one of the three products over the RB staged rows: planes [P0, P0+M) of both operands
template <int M, int RB, int PL>
MW_HD void syrk_rows(const uint32_t (&sa)[PL * RB * 16], const uint32_t (&sb)[PL * RB * 16], int p0, int li, int lj, uint32_t (&acc)[2 * M + 2])
{
  uint64_t c[2 * M - 1];
  uint32_t h[2 * M - 1];
#pragma unroll
  for(int k = 0; k < 2 * M - 1; ++k)
    {
      c[k] = 0;
      h[k] = 0;
    }
  auto row = [&](int rr) __attribute__((always_inline)) {
    uint32_t a[M], b[M];
#pragma unroll
    for(int l = 0; l < M; ++l)
      {
        a[l] = sa[((p0 + l) * RB + rr) * 16 + li];
        b[l] = sb[((p0 + l) * RB + rr) * 16 + lj];
      }
    SyrkColumns<M, 0>::run(a, b, c, h);
  };
  if constexpr(M <= 12)
    {
#pragma unroll 4
      for(int rr = 0; rr < RB; ++rr)
        row(rr);
    }
  else
    {
#pragma unroll 1
      for(int rr = 0; rr < RB; ++rr)
        row(rr);
    }
  syrk_fold<M, 2 * M + 2>(acc, c, h);
}

// TILE-PACKED partial outputs (round 5).  Whenever the product kernels do not write G itself (row splits, the Toom
// images) their output planes hold only the tiles of the launch: element (i, j) of tile t of the launch's tile list at
//   t E E + (i - E ti) + (j - E tj) E        (E = syrk_tile_edge<FX>(), plane stride = ntile E E words)
// -- half the words of the N x N planes of rounds 1-4 (only the lower triangle has tiles), and a launch over a SUBSET of
// the tiles needs planes for that subset only: Solver::syrk_G walks the tile list in chunks whose planes fit a memory
// budget (the reference bounds the same stage by looping over output windows: bigint_syrk_blas.cxx:200-220,
// BigInt_Shared_Memory_Syrk_Context.cxx:149-215, --maxSharedMemory).  The finishing kernels run one lane per packed
// word, find (i, j) through the tile list and drop what lies outside the lower triangle, N, or the columns [col0, col1).
template <int E> MW_HD bool syrk_packed_decode(size_t pidx, const uint32_t *tile_list, int N, int col0, int col1, int &i, int &j)
{
  const uint32_t tt = tile_list[pidx / (size_t)(E * E)];
  const int r = (int)(pidx % (size_t)(E * E));
  i = (int)(tt >> 16) * E + r % E;
  j = (int)(tt & 0xffffu) * E + r / E;
  return i < N && j <= i && j >= col0 && j < col1;
}
// acc(i,j) (i >= j, tiles of 16x16) = G(i,j) = sum_r a'(r,i) a'(r,j), rows r in
// [0, nrows).  fx element (r,n) at r*N + n.  acc element (i,j) at i + j*N in
// a (2FX+2)-plane limb-major array of non-negative integers.  Row chunks of RB rows are
// staged through LDS (limb-major, so lanes of a wavefront hit distinct banks for
// the i operand and broadcast the j operand).  The hot loop is nothing but
// v_mad_u64_u32 + v_addc_co_u32 pairs: per chunk and product, 2M-1 column accumulators
// run over the RB rows and are folded into that product's (2M+2)-limb sum once.
// tile_list[t] = ti << 16 | tj (tj <= ti), built by syrk_tile_order().
// Row splits: workgroup (tile, split) covers rows [split * rows_per_split, ...) and writes
// its partial G to part + split * W * acc_stride (k_syrk_reduce adds the splits); with
// one split the output goes straight to acc.  The host picks nsplit so that
// tiles * nsplit fills a whole number of rounds of resident workgroups (syrk_row_splits).
#ifndef SDPB_SYRK_WAVES
#define SDPB_SYRK_WAVES (FX <= 16 ? 3 : 2) // 3 waves x 168 VGPRs hold the staging registers of the pipeline without spills
#endif
template <int FX> constexpr int syrk_waves_per_simd();
// nsplit in [1, 32]: fewest splits within 2% of the best occupancy of the last round.  (Up to 16 until round 4: with
// N = 100 the output has 28 tiles, and 16 splits filled 448 of the chip's 768 workgroup slots — C3's product took
// 2.08 ms at 8 splits, 3.8 at 4, 15 at 1: profiles/r04k_syrk_row_splits.txt; the row floor of 64 passes per split
// still applies.)
// max_rows > 0 (k_syrk_fx3): at least so many splits that one has no more rows than that.  The workgroups of an XCD that
// stream the same operand panels drift apart by no more than a split's rows, so short splits are what lets them meet in
// that XCD's L2: on C4 (profiles/r04x_syrk3_fetch_vs_splits.txt) FETCH_SIZE per launch 99.7 M KB with 2 splits of 20 000
// rows, 78 with 8, 49 with 16 (2500 rows: same kernel time), 22 with 32 (+ 3.5 % time: the finishing kernel adds 32 x 105 planes).
constexpr int SYRK_MAX_SPLITS = 32;
inline int syrk_row_splits(int ntile, unsigned nrows, int slots, int rb, unsigned max_rows = 0)
{
  if(const char *env = std::getenv("SDPB_HIP_SYRK_SPLITS")) // tests force the split path on small inputs
    return std::max(1, std::min(SYRK_MAX_SPLITS, std::atoi(env)));
  int smin = 1;
  while(max_rows && smin < SYRK_MAX_SPLITS && nrows / (unsigned)smin > max_rows && nrows / (unsigned)(smin + 1) >= 64u * (unsigned)rb)
    ++smin;
  int best = smin;
  double best_eff = 0;
  for(int s = smin; s <= SYRK_MAX_SPLITS; ++s)
    {
      if(s > smin && nrows / (unsigned)s < 64u * (unsigned)rb)
        break;
      const double items = (double)ntile * s, rounds = std::ceil(items / slots), eff = items / (rounds * slots);
      if(eff > best_eff + 0.02)
        {
          best = s;
          best_eff = eff;
        }
    }
  return best;
}
template <int FX, int RB>
__global__ void __launch_bounds__(WG, SDPB_SYRK_WAVES)
  k_syrk_fx(const uint32_t *fx, size_t fx_stride, unsigned nrows, int N, uint32_t *acc, size_t acc_stride, const uint32_t *tile_list,
            int ntile, int nsplit, unsigned rows_per_split, int packed)
{
  // packed: acc is the tile-packed partial array (plane stride acc_stride = ntile 256), else G itself (i + j N)
  constexpr int M = FX / 2, A = 2 * M + 2, W = 2 * FX + 2, PL = fx_planes<FX>();
  // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch), so XCD x is given
  // the contiguous range [x*per, (x+1)*per) of the items (split, tile), tiles enumerated
  // by `tile_list` in 8x8 super-blocks: the workgroups resident on one XCD share a few row
  // and column panels of P' in that XCD's 4-MiB L2 (placement affects speed only, never
  // correctness).
  const int nitem = ntile * nsplit, per = (nitem + 7) / 8;
  const int item = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
  if((int)(blockIdx.x / 8) >= per || item >= nitem)
    return;
  const int split = item / ntile, tile = item % ntile;
  const unsigned row_begin = (unsigned)split * rows_per_split;
  const unsigned row_end = (row_begin + rows_per_split < nrows && split + 1 < nsplit) ? row_begin + rows_per_split : nrows;
  acc += (size_t)split * W * acc_stride;
  const uint32_t tt = tile_list[tile];
  const int ti = (int)(tt >> 16), tj = (int)(tt & 0xffffu);
  const int li = threadIdx.x & 15, lj = threadIdx.x >> 4;
  const int i = ti * 16 + li, j = tj * 16 + lj;
  __shared__ uint32_t sa[PL * RB * 16];
  __shared__ uint32_t sb[PL * RB * 16];
  uint32_t ll[A], hh[A], ss[A];
#pragma unroll
  for(int k = 0; k < A; ++k)
    ll[k] = hh[k] = ss[k] = 0;
  // The three products read disjoint plane groups (lo, hi, s), so the staging of the next
  // chunk is software-pipelined through the passes with a single LDS image: while one group
  // is being multiplied, the group that became free one pass ago is fetched into GL
  // registers per lane, and it is written to LDS when the pass ends:
  //   LL(c) | store s(c)    | barrier | fetch lo(c+1)
  //   HH(c) | store lo(c+1) | barrier | fetch hi(c+1)
  //   SS(c) | store hi(c+1) | barrier | fetch s(c+1)
  // (a barrier both publishes the stores and certifies that every wavefront has left the
  // pass whose group is fetched next).  Without this the global-load latency of a chunk
  // was exposed: 13 % of the kernel.
  constexpr int GE = M * RB * 16, GL = (GE + WG - 1) / WG; // elements of one plane group, per lane
  uint32_t va[GL], vb[GL];
  auto fetch = [&](int g, unsigned r0) __attribute__((always_inline)) {
#pragma unroll
    for(int t = 0; t < GL; ++t)
      {
        const int e = threadIdx.x + t * WG;
        const int col = e & 15, rr = (e >> 4) % RB, pl = g * M + (e >> 4) / RB;
        const unsigned r = r0 + rr;
        const int ca = ti * 16 + col, cb = tj * 16 + col;
        const bool ok = e < GE && r < row_end;
        // lanes outside the image read a valid element and drop it: no divergent branches
        const uint32_t *row = fx + (size_t)(e < GE ? pl : 0) * fx_stride + (size_t)(ok ? r : row_begin) * (size_t)N;
        const uint32_t xa = row[ca < N ? ca : 0], xb = row[cb < N ? cb : 0];
        va[t] = (ok && ca < N) ? xa : 0u; // rows past the end and columns past N: zero limbs,
        vb[t] = (ok && cb < N) ? xb : 0u; // they add nothing to any product
      }
  };
  auto store = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for(int t = 0; t < GL; ++t)
      {
        const int e = threadIdx.x + t * WG;
        if(e < GE)
          {
            sa[g * GE + e] = va[t];
            sb[g * GE + e] = vb[t];
          }
      }
  };
  if constexpr(FX <= 24)
    {
      fetch(0, row_begin);
      store(0);
      fetch(1, row_begin);
      store(1);
      __syncthreads();
      fetch(2, row_begin);
      for(unsigned r0 = row_begin; r0 < row_end; r0 += RB)
        {
          syrk_rows<M, RB, PL>(sa, sb, 0, li, lj, ll);
          store(2);
          __syncthreads();
          fetch(0, r0 + RB);
          syrk_rows<M, RB, PL>(sa, sb, M, li, lj, hh);
          store(0);
          __syncthreads();
          fetch(1, r0 + RB);
          syrk_rows<M, RB, PL>(sa, sb, 2 * M, li, lj, ss);
          store(1);
          __syncthreads();
          fetch(2, r0 + RB);
        }
    }
  else
    {
      // 1024-bit operands: accumulators and column sums leave no registers for the pipeline
      for(unsigned r0 = row_begin; r0 < row_end; r0 += RB)
        {
          for(int e = threadIdx.x; e < PL * RB * 16; e += WG)
            {
              const int col = e & 15, rr = (e >> 4) % RB, pl = (e >> 4) / RB;
              const unsigned r = r0 + rr;
              const int ca = ti * 16 + col, cb = tj * 16 + col;
              const bool okr = r < row_end;
              const uint32_t *row = fx + (size_t)pl * fx_stride + (size_t)r * (size_t)N;
              sa[e] = (okr && ca < N) ? row[ca] : 0u;
              sb[e] = (okr && cb < N) ? row[cb] : 0u;
            }
          __syncthreads();
          syrk_rows<M, RB, PL>(sa, sb, 0, li, lj, ll);
          syrk_rows<M, RB, PL>(sa, sb, M, li, lj, hh);
          syrk_rows<M, RB, PL>(sa, sb, 2 * M, li, lj, ss);
          __syncthreads();
        }
    }
  if(i < N && j <= i)
    {
      // ss <- SS - LL - HH (>= 0), then G = LL + ss B + HH B^2, B = 2^(32M-1)
      uint64_t bw = 0;
#pragma unroll
      for(int k = 0; k < A; ++k)
        {
          const uint64_t d = (uint64_t)ss[k] - (uint64_t)ll[k] - bw;
          ss[k] = (uint32_t)d;
          bw = (d >> 63) & 1u;
        }
      bw = 0;
#pragma unroll
      for(int k = 0; k < A; ++k)
        {
          const uint64_t d = (uint64_t)ss[k] - (uint64_t)hh[k] - bw;
          ss[k] = (uint32_t)d;
          bw = (d >> 63) & 1u;
        }
      uint32_t w[W];
#pragma unroll
      for(int k = 0; k < W; ++k)
        w[k] = k < A ? ll[k < A ? k : 0] : 0u;
      add_shifted<W, A>(w, ss, 32 * M - 1, false);
      add_shifted<W, A>(w, hh, 64 * M - 2, false);
      const size_t o = packed ? (size_t)tile * 256 + threadIdx.x : (size_t)i + (size_t)j * N;
#pragma unroll
      for(int k = 0; k < W; ++k)
        acc[(size_t)k * acc_stride + o] = w[k];
    }
}

// acc(i,j) = sum over the row splits of part[split](i,j)  (i >= j); part tile-packed (syrk_packed_decode), one lane per
// packed word of the launch's `total` = ntile E E.  [col0, col1): the columns of the output the launch covers -- the chunked
// Q' (Solver::q_chase_) finishes, reduces and restores the left columns while the right ones are still being multiplied
template <int FX>
__global__ void __launch_bounds__(WG) k_syrk_reduce(const uint32_t *part, int nsplit, size_t part_stride, const uint32_t *tile_list, size_t total,
                                                    uint32_t *acc, size_t acc_stride, int N, int col0, int col1)
{
  constexpr int W = 2 * FX + 2;
  const size_t pidx = (size_t)blockIdx.x * WG + threadIdx.x;
  int i, j;
  if(pidx >= total || !syrk_packed_decode<syrk_tile_edge<FX>()>(pidx, tile_list, N, col0, col1, i, j))
    return;
  const size_t idx = (size_t)i + (size_t)j * N;
  uint64_t cy = 0;
#pragma unroll
  for(int k = 0; k < W; ++k)
    {
      for(int s = 0; s < nsplit; ++s)
        cy += part[((size_t)s * W + k) * part_stride + pidx];
      acc[(size_t)k * acc_stride + idx] = (uint32_t)cy;
      cy >>= 32;
    }
}

// ---------------------------------------------------------------------------
// The same G with TWO Karatsuba levels (piece-major image, fx_two_level<FX>()).
// ---------------------------------------------------------------------------
// Column sums S_n = sum_r a'_rn from the pieces lo0, lo1, hi0, hi1 (groups 0, 1, 3, 4):
//   S = sum lo0 + sum lo1 B2 + (sum hi0 + sum hi1 B2) B.
// Stage 1: workgroup (x, y) sums row slice y of 64 columns (4 row phases per column, meeting in
// LDS) into partial[y]; element (slice, u, k, n), u < 4, k < M2 + 2, at
// ((slice * 4 + u) * (M2 + 2) + k) * N + n.  Stage 2 adds the slices and recombines.
// (TOOM: the same sums over the groups a0, p(1), p(2), a3 of the Toom-4 image, k_fx_colsum4_final)
template <int FX, bool TOOM = false>
__global__ void __launch_bounds__(WG) k_fx_colsum2(const uint32_t *fx, size_t fx_stride, unsigned nrows, int N, unsigned rows_per_slice, uint32_t *partial)
{
  constexpr int M2 = FX / 4, A = M2 + 2;
  const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane, phase = threadIdx.x >> 6;
  const unsigned r_begin = blockIdx.y * rows_per_slice, r_end = (r_begin + rows_per_slice < nrows) ? r_begin + rows_per_slice : nrows;
  uint32_t sum[4][A];
#pragma unroll
  for(int u = 0; u < 4; ++u)
#pragma unroll
    for(int k = 0; k < A; ++k)
      sum[u][k] = 0;
  if(col < N)
    for(unsigned r = r_begin + phase; r < r_end; r += 4)
      {
        const size_t e = (size_t)r * N + col;
#pragma unroll
        for(int u = 0; u < 4; ++u)
          {
            const int grp = TOOM ? (u == 0 ? 0 : u == 1 ? 1 : u == 2 ? 3 : 6) : (u < 2 ? u : u + 1);
            uint32_t src[M2];
            if constexpr(TOOM)
              toom_piece_load<FX>(fx, fx_stride, grp, e, src);
            else
              piece_load<M2>(fx + ((size_t)grp * fx_stride + e) * M2, src);
            uint64_t cy = 0;
#pragma unroll
            for(int k = 0; k < A; ++k)
              {
                const uint64_t t = (uint64_t)sum[u][k] + (k < M2 ? src[k < M2 ? k : 0] : 0u) + cy;
                sum[u][k] = (uint32_t)t;
                cy = t >> 32;
              }
          }
      }
  __shared__ uint32_t sm[3 * 4 * A * 64];
  if(phase > 0)
    {
#pragma unroll
      for(int u = 0; u < 4; ++u)
#pragma unroll
        for(int k = 0; k < A; ++k)
          sm[(((phase - 1) * 4 + u) * A + k) * 64 + lane] = sum[u][k];
    }
  __syncthreads();
  if(phase == 0 && col < N)
    {
      for(int p = 0; p < 3; ++p)
#pragma unroll
        for(int u = 0; u < 4; ++u)
          {
            uint64_t cy = 0;
#pragma unroll
            for(int k = 0; k < A; ++k)
              {
                const uint64_t t = (uint64_t)sum[u][k] + sm[((p * 4 + u) * A + k) * 64 + lane] + cy;
                sum[u][k] = (uint32_t)t;
                cy = t >> 32;
              }
          }
#pragma unroll
      for(int u = 0; u < 4; ++u)
#pragma unroll
        for(int k = 0; k < A; ++k)
          partial[(((size_t)blockIdx.y * 4 + u) * A + k) * N + col] = sum[u][k];
    }
}
template <int FX>
__global__ void __launch_bounds__(WG) k_fx_colsum2_final(const uint32_t *partial, int nslices, int N, uint32_t *acc, size_t acc_stride)
{
  constexpr int M2 = FX / 4, M = FX / 2, A = M2 + 2, W = 2 * FX + 2;
  const int col = blockIdx.x * WG + threadIdx.x;
  if(col >= N)
    return;
  uint32_t sum[4][A];
#pragma unroll
  for(int u = 0; u < 4; ++u)
#pragma unroll
    for(int k = 0; k < A; ++k)
      sum[u][k] = 0;
  for(int s = 0; s < nslices; ++s)
#pragma unroll
    for(int u = 0; u < 4; ++u)
      {
        uint64_t cy = 0;
#pragma unroll
        for(int k = 0; k < A; ++k)
          {
            const uint64_t t = (uint64_t)sum[u][k] + partial[(((size_t)s * 4 + u) * A + k) * N + col] + cy;
            sum[u][k] = (uint32_t)t;
            cy = t >> 32;
          }
      }
  uint32_t w[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    w[k] = k < A ? sum[0][k < A ? k : 0] : 0u;
  add_shifted<W, A>(w, sum[1], 32 * M2 - 1, false);
  add_shifted<W, A>(w, sum[2], 32 * M - 3, false);
  add_shifted<W, A>(w, sum[3], 32 * M - 3 + 32 * M2 - 1, false);
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + (size_t)N * N + col] = w[k];
}

// Toom-4 image: from the slice sums of the groups a0, p(1), p(2), a3 (k_fx_colsum2<FX, true>)
//   s0 = sum a0, s3 = sum a3, u = sum p(1) - s0 - s3 = s1 + s2, v = (sum p(2) - s0 - 8 s3)/2 = s1 + 2 s2
//   S   = s0 + s1 b + s2 b^2 + s3 b^3                      -> behind the N x N block of acc (k_syrk_unbias)
//   E1  = s0 - s1 + s2 - s3 = sum_r p_r(-1),  E2 = s0 - 2 s1 + 4 s2 - 8 s3 = sum_r p_r(-2)
//   U1  = K1 E1 + n K1^2 / 2,  U2 = K2 E2 + n K2^2 / 2       -> toomU (Z limbs each; k_syrk4_finish)
// so that sum_r p_ri(-1) p_rj(-1) = sum_r (p_ri(-1) + K1)(p_rj(-1) + K1) - (U1_i + U1_j): with T = E + n K
// the stored (biased) pieces sum to T, and K (T_i + T_j) - n K^2 = K (E_i + E_j) + n K^2 = U_i + U_j.
// n = the rows of THIS rank (the bias of the stored pieces is removed before any cross-GPU sum).
template <int FX>
__global__ void __launch_bounds__(WG)
  k_fx_colsum4_final(const uint32_t *partial, int nslices, int N, uint32_t *acc, size_t acc_stride, uint32_t *toomU, unsigned long long nrows_local)
{
  constexpr int M2 = FX / 4, A = M2 + 2, W = 2 * FX + 2, WB = toom_wb<FX>(), Z = 2 * M2 + 2;
  const int col = blockIdx.x * WG + threadIdx.x;
  if(col >= N)
    return;
  uint32_t sum[4][A];
#pragma unroll
  for(int u = 0; u < 4; ++u)
#pragma unroll
    for(int k = 0; k < A; ++k)
      sum[u][k] = 0;
  for(int s = 0; s < nslices; ++s)
#pragma unroll
    for(int u = 0; u < 4; ++u)
      {
        uint64_t cy = 0;
#pragma unroll
        for(int k = 0; k < A; ++k)
          {
            const uint64_t t = (uint64_t)sum[u][k] + partial[(((size_t)s * 4 + u) * A + k) * N + col] + cy;
            sum[u][k] = (uint32_t)t;
            cy = t >> 32;
          }
      }
  uint32_t s0[Z], s1[Z], s2[Z], s3[Z], t[Z];
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      s0[k] = k < A ? sum[0][k < A ? k : 0] : 0u;
      s1[k] = k < A ? sum[1][k < A ? k : 0] : 0u; // sum p(1)
      s2[k] = k < A ? sum[2][k < A ? k : 0] : 0u; // sum p(2)
      s3[k] = k < A ? sum[3][k < A ? k : 0] : 0u;
    }
  z_sub<Z>(s1, s0); // u = s1 + s2
  z_sub<Z>(s1, s3);
  z_sub<Z>(s2, s0); // v = s1 + 2 s2
  z_addmul<Z>(s2, s3, 8, true);
  z_sar<Z>(s2, 1);
  z_sub<Z>(s2, s1); // s2 = v - u
  z_sub<Z>(s1, s2); // s1 = u - s2
  uint32_t w[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    w[k] = k < Z ? s0[k < Z ? k : 0] : 0u;
  add_shifted<W, Z>(w, s1, WB, false);
  add_shifted<W, Z>(w, s2, 2 * WB, false);
  add_shifted<W, Z>(w, s3, 3 * WB, false);
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + (size_t)N * N + col] = w[k];
  uint32_t n[Z];
#pragma unroll
  for(int k = 0; k < Z; ++k)
    n[k] = k == 0 ? (uint32_t)nrows_local : k == 1 ? (uint32_t)(nrows_local >> 32) : 0u;
  // U1 = (E1 << (WB+1)) + (n << (2 WB + 1)),  K1 = 2^(WB+1)
#pragma unroll
  for(int k = 0; k < Z; ++k)
    t[k] = s0[k];
  z_sub<Z>(t, s1);
  z_add<Z>(t, s2);
  z_sub<Z>(t, s3);
  z_shl<Z>(t, WB + 1);
  {
    uint32_t m[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      m[k] = n[k];
    z_shl<Z>(m, 2 * WB + 1);
    z_add<Z>(t, m);
  }
#pragma unroll
  for(int k = 0; k < Z; ++k)
    toomU[(size_t)k * N + col] = t[k];
  // U2 = (10 E2 << WB) + (50 n << 2 WB),  K2 = 10 2^WB
#pragma unroll
  for(int k = 0; k < Z; ++k)
    t[k] = s0[k];
  z_addmul<Z>(t, s1, 2, true);
  z_addmul<Z>(t, s2, 4, false);
  z_addmul<Z>(t, s3, 8, true);
  {
    uint32_t m[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      m[k] = 0;
    z_addmul<Z>(m, t, 10, false);
    z_shl<Z>(m, WB);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      t[k] = 0;
    z_addmul<Z>(t, n, 50, false);
    z_shl<Z>(t, 2 * WB);
    z_add<Z>(t, m);
  }
#pragma unroll
  for(int k = 0; k < Z; ++k)
    toomU[(size_t)(Z + k) * N + col] = t[k];
}
// acc(i,j) (i >= j) = G(i,j) = sum_r a'_ri a'_rj from the seven product sums the row splits of
// k_syrk_fx2<FX, RBG, true> left in part (split s, group g, limb k at ((s 7 + g) A2 + k) part_stride + packed index):
// add the splits, remove the bias of the two signed evaluation points, interpolate, recombine.
template <int FX>
__global__ void __launch_bounds__(WG)
  k_syrk4_finish(const uint32_t *part, int nsplit, size_t part_stride, const uint32_t *tile_list, size_t total, const uint32_t *toomU,
                 uint32_t *acc, size_t acc_stride, int N, int col0, int col1)
{
  constexpr int M2 = FX / 4, A2 = 2 * M2 + 1, Z = 2 * M2 + 2, W = 2 * FX + 2, WB = toom_wb<FX>();
  const size_t pidx = (size_t)blockIdx.x * WG + threadIdx.x;
  int i, j;
  if(pidx >= total || !syrk_packed_decode<syrk_tile_edge<FX>()>(pidx, tile_list, N, col0, col1, i, j))
    return;
  const size_t idx = (size_t)i + (size_t)j * N;
  uint32_t w[7][Z];
  // GMP's order: w0 = f(0), w1 = f(-2), w2 = f(1), w3 = f(-1), w4 = f(2), w5 = 64 f(1/2), w6 = f(inf)
  constexpr int GRP[7] = {0, 4, 1, 2, 3, 5, 6};
  if constexpr(fx_toom4k<FX>())
    {
      // k_syrk_fx3 left the sums of the 21 products, A3 limbs each: e e' = lo lo' + (mid mid' - lo lo' - hi hi') 2^H + hi hi' 2^(2H)
      constexpr int M3 = FX / 8, A3 = 2 * M3 + 1, H = 16 * M2 - 1;
#pragma unroll
      for(int q = 0; q < 7; ++q)
        {
          uint32_t g3[3][A3];
#pragma unroll
          for(int u = 0; u < 3; ++u)
            {
              uint64_t cy = 0;
#pragma unroll
              for(int k = 0; k < A3; ++k)
                {
                  for(int s = 0; s < nsplit; ++s)
                    cy += part[(((size_t)s * 21 + 3 * GRP[q] + u) * A3 + k) * part_stride + pidx];
                  g3[u][k] = (uint32_t)cy;
                  cy >>= 32;
                }
            }
          sub_limbs<A3>(g3[2], g3[0]);
          sub_limbs<A3>(g3[2], g3[1]);
#pragma unroll
          for(int k = 0; k < Z; ++k)
            w[q][k] = k < A3 ? g3[0][k < A3 ? k : 0] : 0u;
          add_shifted<Z, A3>(w[q], g3[2], H, false);
          add_shifted<Z, A3>(w[q], g3[1], 2 * H, false);
        }
    }
  else
    {
#pragma unroll
  for(int q = 0; q < 7; ++q)
    {
      uint64_t cy = 0;
#pragma unroll
      for(int k = 0; k < Z; ++k)
        {
          if(k < A2)
            for(int s = 0; s < nsplit; ++s)
              cy += part[(((size_t)s * 7 + GRP[q]) * A2 + k) * part_stride + pidx];
          w[q][k] = (uint32_t)cy;
          cy >>= 32;
        }
    }
    }
  {
    uint32_t u[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      u[k] = toomU[(size_t)k * N + i];
    z_sub<Z>(w[3], u);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      u[k] = toomU[(size_t)k * N + j];
    z_sub<Z>(w[3], u);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      u[k] = toomU[(size_t)(Z + k) * N + i];
    z_sub<Z>(w[1], u);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      u[k] = toomU[(size_t)(Z + k) * N + j];
    z_sub<Z>(w[1], u);
  }
  toom4_interpolate<Z>(w);
  uint32_t g[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    g[k] = k < Z ? w[0][k < Z ? k : 0] : 0u;
#pragma unroll
  for(int q = 1; q < 7; ++q)
    add_shifted<W, Z>(g, w[q], q * WB, false);
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + idx] = g[k];
}


// acc(i,j) (i >= j) = G(i,j) = sum_r a'(r,i) a'(r,j) exactly as k_syrk_fx (same tiles, same
// XCD-aware item order, same row splits), but nine M2 x M2 products per row pair.  One PASS =
// one product g over RBG rows of the tile's two operand panels: the 16 x RBG pieces of group g
// of each panel sit in LDS ([row][column][M2 limbs], a lane reads its piece with one
// ds_read_b128 when M2 = 4; the i operand is conflict-free, the j operand a broadcast), the
// 2 M2 - 1 column accumulators run over the RBG rows and are folded into that product's
// (2 M2 + 2)-limb sum once.  The pieces of the NEXT pass are fetched from HBM/L2 into registers
// while the MACs run and are written to the other LDS buffer when the pass ends (one barrier
// per pass).  The nine sums are recombined once per output element after the row loop.
// Tuning knobs of the row loop, measured on C4 (40 000 x 1000, 512 bits; profiles/r02e_syrk_staging_variants.txt):
//   direct-to-LDS staging, rows unrolled x4        185.1 ms, no spills   <- default
//   register staging,      rows unrolled x4        183.9 ms, 2 spilled VGPRs
//   register staging,      explicit LDS prefetch   186.6 ms, 11 spilled VGPRs = 8.9 GB of scratch writes per launch
//   direct-to-LDS staging, explicit LDS prefetch   192.6 ms, no spills
#ifndef SDPB_SYRK2_PREFETCH
#define SDPB_SYRK2_PREFETCH 0
#endif
#ifndef SDPB_SYRK2_UNROLL
#define SDPB_SYRK2_UNROLL 4 // row loop of the non-prefetching variant
#endif
// TOOM: the seven products of the Toom-4 image instead (one sweep); the seven row sums are written
// as they are, A2 limbs each, to split * 7 A2 planes of the output, and k_syrk4_finish interpolates.
template <int FX, int RBG, bool TOOM = false>
__global__ void __launch_bounds__(WG, SDPB_SYRK_WAVES)
  k_syrk_fx2(const uint32_t *__restrict__ fx_in, size_t fx_stride, unsigned nrows, int N, uint32_t *acc, size_t acc_stride,
             const uint32_t *tile_list, int ntile, int nsplit, unsigned rows_per_split, const uint32_t *zero_piece, int packed)
{
  // packed (always with TOOM): acc is the tile-packed partial array (plane stride acc_stride = ntile 256), else G itself
  // A2 limbs hold the sum of one second-level product over a split's rows: pieces < 2^(32 M2), fewer
  // than 2^32 rows, so the sum stays below 2^(64 M2 + 32)
  constexpr int M2 = FX / 4, M = FX / 2, A2 = 2 * M2 + 1, A = 2 * M + 2, W = 2 * FX + 2;
  constexpr int NP = RBG * 16, GL = NP / WG; // pieces per operand per pass, per lane
  static_assert(NP % WG == 0, "a pass stages a whole number of pieces per lane");
  const uint32_t *fx = (const uint32_t *)__builtin_assume_aligned(fx_in, 16);
  const int nitem = ntile * nsplit, per = (nitem + 7) / 8;
  const int item = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
  if((int)(blockIdx.x / 8) >= per || item >= nitem)
    return;
  const int split = item / ntile, tile = item % ntile;
  const unsigned row_begin = (unsigned)split * rows_per_split;
  const unsigned row_end = (row_begin + rows_per_split < nrows && split + 1 < nsplit) ? row_begin + rows_per_split : nrows;
  acc += (size_t)split * (TOOM ? 7 * A2 : W) * acc_stride;
  const uint32_t tt = tile_list[tile];
  const int ti = (int)(tt >> 16), tj = (int)(tt & 0xffffu);
  const int li = threadIdx.x & 15, lj = threadIdx.x >> 4;
  const int i = ti * 16 + li, j = tj * 16 + lj;
  __shared__ __attribute__((aligned(16))) uint32_t sa[2 * NP * M2];
  __shared__ __attribute__((aligned(16))) uint32_t sb[2 * NP * M2];
  // NG of the nine products are accumulated per sweep over the rows: all nine when their sums fit
  // the register file, else (1024-bit operands: 9 x 18 limbs) the three of one first-level operand
  // at a time — three sweeps, the same passes in another order, each closed by its second-level
  // recombination into x[sweep]
  // (Toom-4: all seven up to 1024 bits — 226 VGPRs at FX = 32 — and four + three in two sweeps above)
  constexpr int NGRP = TOOM ? 7 : 9;
  constexpr int NG = TOOM ? (FX > 32 ? 4 : 7) : FX >= 32 ? 3 : 9, SWEEPS = (NGRP + NG - 1) / NG;
  uint32_t g2[NG][A2];
  uint32_t x[TOOM ? 1 : 3][TOOM ? 1 : A];
  // Staging of the NEXT pass.  16-byte pieces (M2 = 4) go HBM/L2 -> LDS directly
  // (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16 B, exactly the [row][column] piece
  // order), with no staging registers and no ds_write pass; the barrier that ends the pass drains
  // them (vmcnt) before any wavefront reads the buffer.  Rows past the end and columns past N load
  // a zero piece: they add nothing to any product.  Other piece sizes stage through registers.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDPB_SYRK2_NO_GLDS)
  constexpr bool DIRECT = (M2 == 4);
#else
  constexpr bool DIRECT = false;
#endif
  uint32_t va[DIRECT ? 1 : GL][M2], vb[DIRECT ? 1 : GL][M2];
  auto fetch = [&](int g, unsigned r0, int into) __attribute__((always_inline)) {
#pragma unroll
    for(int t = 0; t < GL; ++t)
      {
        const int e = threadIdx.x + t * WG;
        const int col = e & 15, rr = e >> 4;
        const unsigned r = r0 + rr;
        const int ca = ti * 16 + col, cb = tj * 16 + col;
        const bool ok = r < row_end;
        const uint32_t *row = fx + ((size_t)g * fx_stride + (size_t)(ok ? r : row_begin) * (size_t)N) * M2;
        if constexpr(DIRECT)
          {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t *pa = (ok && ca < N) ? row + (size_t)ca * M2 : zero_piece;
            const uint32_t *pb = (ok && cb < N) ? row + (size_t)cb * M2 : zero_piece;
            const int wbase = (into * NP + t * WG + (int)(threadIdx.x & ~63u)) * M2; // this wavefront's 64 pieces
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pa,
                                             (__attribute__((address_space(3))) void *)(sa + wbase), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pb,
                                             (__attribute__((address_space(3))) void *)(sb + wbase), 16, 0, 0);
#endif
          }
        else
          {
            // lanes outside the image read a valid piece and drop it: no divergent branches
            piece_load<M2>(row + (size_t)(ca < N ? ca : 0) * M2, va[t]);
            piece_load<M2>(row + (size_t)(cb < N ? cb : 0) * M2, vb[t]);
            const uint32_t ma = (ok && ca < N) ? 0xffffffffu : 0u, mb = (ok && cb < N) ? 0xffffffffu : 0u;
#pragma unroll
            for(int l = 0; l < M2; ++l)
              {
                va[t][l] &= ma;
                vb[t][l] &= mb;
              }
          }
      }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
    if constexpr(!DIRECT)
      {
#pragma unroll
        for(int t = 0; t < GL; ++t)
          {
            const int e = threadIdx.x + t * WG;
            piece_store<M2>(sa + (buf * NP + e) * M2, va[t]);
            piece_store<M2>(sb + (buf * NP + e) * M2, vb[t]);
          }
      }
  };
  // second level: XX = X0X0 + (XtXt - X0X0 - X1X1) B2 + X1X1 B2^2 for X = first-level operand k
  auto second_level = [&](int k, int g0) __attribute__((always_inline)) {
    if constexpr(TOOM)
      return;
    else
      {
    sub_limbs<A2>(g2[g0 + 2], g2[g0]);
    sub_limbs<A2>(g2[g0 + 2], g2[g0 + 1]);
#pragma unroll
    for(int q = 0; q < A; ++q)
      x[k][q] = q < A2 ? g2[g0][q < A2 ? q : 0] : 0u;
    add_shifted<A, A2>(x[k], g2[g0 + 2], 32 * M2 - 1, false);
    add_shifted<A, A2>(x[k], g2[g0 + 1], 64 * M2 - 2, false);
      }
  };
#pragma unroll
  for(int sweep = 0; sweep < SWEEPS; ++sweep)
  {
#pragma unroll
  for(int g = 0; g < NG; ++g)
#pragma unroll
    for(int k = 0; k < A2; ++k)
      g2[g][k] = 0;
  if(sweep > 0)
    __syncthreads(); // every wavefront has left the last pass of the previous sweep
  const int ngs = NGRP - sweep * NG < NG ? NGRP - sweep * NG : NG; // products of this sweep (compile time once unrolled)
  fetch(sweep * NG, row_begin, 0);
  store(0);
  __syncthreads();
  int buf = 0;
  for(unsigned r0 = row_begin; r0 < row_end; r0 += RBG)
    {
#pragma unroll
      for(int g = 0; g < NG; ++g)
        {
          if(g >= ngs)
            continue;
          fetch(sweep * NG + (g < ngs - 1 ? g + 1 : 0), g < ngs - 1 ? r0 : r0 + RBG, buf ^ 1);
          uint64_t c[2 * M2 - 1];
          uint32_t h[2 * M2 - 1];
#pragma unroll
          for(int k = 0; k < 2 * M2 - 1; ++k)
            {
              c[k] = 0;
              h[k] = 0;
            }
          const uint32_t *pa = sa + (buf * NP + li) * M2, *pb = sb + (buf * NP + lj) * M2;
          if constexpr(SDPB_SYRK2_PREFETCH && M2 <= 6)
            {
              // the LDS reads of the next row are issued before the MACs of the current one
              uint32_t a0[M2], b0[M2], a1[M2], b1[M2];
              piece_load<M2>(pa, a0);
              piece_load<M2>(pb, b0);
#pragma unroll 1
              for(int rr = 0; rr < RBG; rr += 2)
                {
                  piece_load<M2>(pa + (rr + 1) * 16 * M2, a1);
                  piece_load<M2>(pb + (rr + 1) * 16 * M2, b1);
                  SyrkColumns<M2, 0>::run(a0, b0, c, h);
                  const int nx = rr + 2 < RBG ? rr + 2 : 0; // the read past the last row is dropped
                  piece_load<M2>(pa + nx * 16 * M2, a0);
                  piece_load<M2>(pb + nx * 16 * M2, b0);
                  SyrkColumns<M2, 0>::run(a1, b1, c, h);
                }
            }
          else
            {
#pragma unroll SDPB_SYRK2_UNROLL
              for(int rr = 0; rr < RBG; ++rr)
                {
                  uint32_t a[M2], b[M2];
                  piece_load<M2>(pa + rr * 16 * M2, a);
                  piece_load<M2>(pb + rr * 16 * M2, b);
                  SyrkColumns<M2, 0>::run(a, b, c, h);
                }
            }
          syrk_fold<M2, A2>(g2[g], c, h);
          store(buf ^ 1);
          __syncthreads();
          buf ^= 1;
        }
    }
  if constexpr(TOOM)
    {
      if(i < N && j <= i)
        {
          const size_t o = (size_t)tile * 256 + threadIdx.x;
#pragma unroll
          for(int g = 0; g < NG; ++g)
            if(g < ngs)
              {
#pragma unroll
                for(int k = 0; k < A2; ++k)
                  acc[(size_t)((sweep * NG + g) * A2 + k) * acc_stride + o] = g2[g][k];
              }
        }
    }
  else if constexpr(SWEEPS == 3)
    second_level(sweep, 0);
  }
  if constexpr(TOOM)
    return;
  else
  {
  if constexpr(SWEEPS == 1)
    {
#pragma unroll
      for(int k = 0; k < 3; ++k)
        second_level(k, 3 * k);
    }
  if(i < N && j <= i)
    {
      // first level: G = LL + (SS - LL - HH) B + HH B^2, B = 2^(32M-3)
      sub_limbs<A>(x[2], x[0]);
      sub_limbs<A>(x[2], x[1]);
      uint32_t w[W];
#pragma unroll
      for(int k = 0; k < W; ++k)
        w[k] = k < A ? x[0][k < A ? k : 0] : 0u;
      add_shifted<W, A>(w, x[2], 32 * M - 3, false);
      add_shifted<W, A>(w, x[1], 64 * M - 6, false);
      const size_t o = packed ? (size_t)tile * 256 + threadIdx.x : (size_t)i + (size_t)j * N;
#pragma unroll
      for(int k = 0; k < W; ++k)
        acc[(size_t)k * acc_stride + o] = w[k];
    }
  }
}

// The Toom-4 x Karatsuba product (fx_toom4k<FX>()): the sums over a row split of the 21 products x(r,i) x(r,j) of the
// M3-limb pieces of the 21-group image, A3 = 2 M3 + 1 limbs each, plane (split * 21 + product) * A3 + limb of the output;
// k_syrk4_finish adds the splits, recombines the three products of a Toom-4 group and interpolates.
// A workgroup owns a 32 x 32 tile, a lane 2 x 2 outputs: i in {i0, i0 + 16}, j in {j0, j0 + 16} with i0 = 32 ti + li,
// j0 = 32 tj + lj, so per staged row it reads two pieces of each operand (8 B each at M3 = 2: the i pieces
// conflict-free, the j pieces a broadcast) for four M3 x M3 products -- at M3 = 2 the same 16 MAC pairs per 32 B of LDS
// reads as the 4 x 4-limb product of k_syrk_fx2, and the same staging: 16-byte global_load_lds units of a row of
// adjacent columns' pieces, [row][column][M3 limbs] in LDS.  Quadrants of the tile that hold no output (above the
// diagonal of a diagonal tile, past column N) are skipped workgroup-wide, so the executed products are those of the
// 16 x 16 tiling.  One SWEEP over the split's rows = one of the 21 products: a pass per block of RBG rows, the column
// accumulators (96 bits each, 4 x (2 M3 - 1) per lane) live in registers for the whole sweep and are folded when it ends.
#ifndef SDPB_SYRK3_PREFETCH
#define SDPB_SYRK3_PREFETCH 1 // measured on C4 (profiles/r04s_syrk3_variants.txt): 101.9 ms against 103.8 without
#endif
#ifndef SDPB_SYRK3_WAVES
#define SDPB_SYRK3_WAVES 3
#endif
#ifndef SDPB_SYRK3_NBUF
#define SDPB_SYRK3_NBUF 2 // staging buffers of the lazy-carry mode; 3 (requests two passes ahead) was measured: no gain, see below
#endif
template <int FX, int RBG>
__global__ void __launch_bounds__(WG, SDPB_SYRK3_WAVES)
  k_syrk_fx3(const uint32_t *__restrict__ fx_in, size_t fx_stride, unsigned nrows, int N, uint32_t *acc, size_t acc_stride,
             const uint32_t *tile_list, int ntile, int nsplit, unsigned rows_per_split, int gsplit, int order)
{
  // gsplit = 7 (21): a workgroup takes the three products of ONE Toom-4 group (one product) of its (tile, row split)
  // instead of all 21 (the sweeps are independent): more workgroups where the output has few tiles, a shorter tail everywhere
  // LAZY (fx_toom5k: 27 products of pieces with 28-bit limbs): the column accumulators are plain 64-bit sums -- a multiply-add
  // per limb pair and no carry instruction; they are carried into each other every second pass (64 rows)
  constexpr bool LAZY = fx_toom5k<FX>();
  constexpr int M3 = FX / 8, A3 = fx_part_limbs<FX>(), NPROD = fx_nprod<FX>();
  static_assert(M3 >= 2 && M3 <= 4 && (!LAZY || (M3 == 2 && RBG <= 32)), "accumulators of 2 x 2 outputs in registers; 2 x 32 rows between carries");
  // a staged row = the M3-limb pieces of the tile's 32 columns = ROWW words = UPR 16-byte units (a unit is two pieces at
  // M3 = 2, one at 4, and straddles pieces at 3: the row is contiguous in the image and in LDS either way)
  constexpr int ROWW = 32 * M3, UPR = ROWW / 4, NPAIR = RBG * UPR, GL = NPAIR / WG; // units per operand per pass, per lane
  static_assert(NPAIR % WG == 0, "a pass stages a whole number of units per lane");
  const uint32_t *fx = (const uint32_t *)__builtin_assume_aligned(fx_in, 4);
  const int nitem = ntile * nsplit * gsplit, per = (nitem + 7) / 8;
  int item = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
  if(order == 1)
    {
      // (split, product)-major over the WHOLE chip (an experiment, SDPB_HIP_SYRK_ORDER=1; profiles/r05_syrk_order.txt): every
      // XCD takes a contiguous eighth of the tiles of the same (split, product), so that all resident workgroups stream
      // ONE panel set of rows_per_split x N pieces (20 MB on C4) at a time instead of eight of them
      const int tper = (ntile + 7) / 8, slot = (int)(blockIdx.x / 8), sgx = slot / tper, tl = (int)(blockIdx.x % 8) * tper + slot % tper;
      if(sgx >= nsplit * gsplit || tl >= ntile)
        return;
      item = sgx * ntile + tl;
    }
  else if((int)(blockIdx.x / 8) >= per || item >= nitem)
    return;
  // items that follow each other share the rows and the group, i.e. the operand panels of neighbouring tiles
  const int sg = item / ntile, tile = item % ntile, split = sg / gsplit;
  const int prod_begin = (sg % gsplit) * (NPROD / gsplit), prod_end = prod_begin + NPROD / gsplit;
  const unsigned row_begin = (unsigned)split * rows_per_split;
  const unsigned row_end = (row_begin + rows_per_split < nrows && split + 1 < nsplit) ? row_begin + rows_per_split : nrows;
  acc += (size_t)split * NPROD * A3 * acc_stride;
  const uint32_t tt = tile_list[tile];
  const int ti = (int)(tt >> 16), tj = (int)(tt & 0xffffu);
  const int li = threadIdx.x & 15, lj = threadIdx.x >> 4;
  const int i0 = ti * 32 + li, j0 = tj * 32 + lj;
  // quadrant (p, q): rows i0 + 16 p of the output against columns j0 + 16 q; workgroup-uniform
  const bool half_i = ti * 32 + 16 < N, half_j = tj * 32 + 16 < N;
  const int mask = 1 | (half_i ? 2 : 0) | ((half_j && tj < ti) ? 4 : 0) | ((half_i && half_j) ? 8 : 0); // bit p + 2 q
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDPB_SYRK2_NO_GLDS)
  // -DSDPB_SYRK3_NBUF=3 (LAZY): three staging buffers, the rows of pass i + 2 requested while pass i multiplies.  Measured and
  // NOT the default (profiles/r06_syrk_prefetch_depth.txt): the kernel without any fetch after the first pass of a sweep
  // (wrong results, timing only: -DSDPB_SYRK3_EXPERIMENT_NO_FETCH) takes 74.3 instead of 82.2 ms for the whole syrk_G, which
  // looked like passes waiting for rows that miss the XCD's L2 -- but requesting the rows a pass earlier changes nothing
  // (81.9 against 80.9 ms, bit-exact either way): the 8 ms are the cost of the requests themselves, not of waiting for them.
  constexpr int NBUF = LAZY ? SDPB_SYRK3_NBUF : 2;
#else
  constexpr int NBUF = 2;
#endif
  static_assert(NBUF == 2 || NBUF == 3, "double or triple buffering");
  // (+ 64 words: the prefetching row loop reads one row past the pass it is in)
  __shared__ __attribute__((aligned(16))) uint32_t sa[NBUF * NPAIR * 4 + 64];
  __shared__ __attribute__((aligned(16))) uint32_t sb[NBUF * NPAIR * 4 + 64];
  uint64_t cc[4][2 * M3 - 1];
  uint32_t hh[4][2 * M3 - 1];
  uint64_t dd[4][2 * M3 - 1]; // LAZY: the high words taken out of the columns (weight 2^32 of their column)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDPB_SYRK2_NO_GLDS)
  constexpr bool DIRECT = true;
#else
  constexpr bool DIRECT = false;
#endif
  // Staging of the NEXT pass: the image is padded with zero rows to a multiple of RBG and every row block a workgroup
  // touches lies inside it (fx3_image_stride), so a pass is the same per-lane offsets from a workgroup-uniform base:
  // no per-lane pointer selects.  Columns past N load whatever follows the row (the pad behind the last one): they
  // only reach outputs that are not stored.
  uint32_t va[DIRECT ? 1 : GL][4], vb[DIRECT ? 1 : GL][4];
  uint32_t offa[GL], offb[GL];
#pragma unroll
  for(int t = 0; t < GL; ++t)
    {
      const int e = threadIdx.x + t * WG;
      const int unit = e % UPR, rr = e / UPR;
      offa[t] = (uint32_t)(((size_t)rr * N + ti * 32) * M3 + 4 * unit);
      offb[t] = (uint32_t)(((size_t)rr * N + tj * 32) * M3 + 4 * unit);
    }
  auto fetch = [&](int g, unsigned r0, int into) __attribute__((always_inline)) {
    const uint32_t *base = fx + ((size_t)g * fx_stride + (size_t)r0 * (size_t)N) * M3;
#pragma unroll
    for(int t = 0; t < GL; ++t)
      {
        if constexpr(DIRECT)
          {
#if defined(__HIP_DEVICE_COMPILE__)
            const int wbase = (into * NPAIR + t * WG + (int)(threadIdx.x & ~63u)) * 4; // this wavefront's 64 pairs
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + offa[t]),
                                             (__attribute__((address_space(3))) void *)(sa + wbase), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + offb[t]),
                                             (__attribute__((address_space(3))) void *)(sb + wbase), 16, 0, 0);
#endif
          }
        else
          {
#pragma unroll
            for(int l = 0; l < 4; ++l)
              {
                va[t][l] = base[offa[t] + l];
                vb[t][l] = base[offb[t] + l];
              }
          }
      }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
    if constexpr(!DIRECT)
      {
#pragma unroll
        for(int t = 0; t < GL; ++t)
          {
            const int e = threadIdx.x + t * WG;
            piece_store<4>(sa + (buf * NPAIR + e) * 4, va[t]);
            piece_store<4>(sb + (buf * NPAIR + e) * 4, vb[t]);
          }
      }
  };
  // the RBG staged rows of one pass for the quadrants of MASK: the column accumulators c + h 2^64 (96 bits: a sweep has
  // fewer than 2^31 rows) persist over the whole sweep and are folded once, when it ends
  auto rows = [&](auto mask_c, int buf, uint64_t (&c)[4][2 * M3 - 1], uint32_t (&h)[4][2 * M3 - 1]) __attribute__((always_inline)) {
    constexpr int MASK = decltype(mask_c)::value;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDPB_SYRK3_NO_ASM)
    constexpr bool ASM_ROWS = (M3 == 2);
#else
    constexpr bool ASM_ROWS = false;
#endif
    if constexpr(ASM_ROWS)
      {
#if defined(__HIP_DEVICE_COMPILE__)
    // LDS reads and MACs as a few large asm statements per row (every asm statement costs a wait state), and the
    // reads invisible to the compiler's wait-count pass, which would otherwise drain the global_load_lds of the NEXT
    // pass (vmcnt(0)) before the first read of this one -- those land in the other buffer; the barrier that ends the
    // pass waits for them
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(sa + buf * NPAIR * 4 + li * M3);
    const uint32_t lb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(sb + buf * NPAIR * 4 + lj * M3);
#define FX3_P(C, H, X, Y) "v_mad_u64_u32 %" #C ", vcc, %" #X ", %" #Y ", %" #C "\n\tv_addc_co_u32 %" #H ", vcc, 0, %" #H ", vcc\n\t"
#define FX3_L(C, X, Y) "v_mad_u64_u32 %" #C ", vcc, %" #X ", %" #Y ", %" #C "\n\t"
#define FX3_MAC_BOTH(o0, o1, a0x, a0y, a1x, a1y, bx, by)                                                                                   \
  if constexpr(LAZY)                                                                                                                       \
    asm volatile(FX3_L(0, 6, 10) FX3_L(3, 8, 10) FX3_L(1, 6, 11) FX3_L(4, 8, 11) FX3_L(1, 7, 10) FX3_L(4, 9, 10) FX3_L(2, 7, 11)           \
                   FX3_L(5, 9, 11)                                                                                                         \
                 : "+v"(c[o0][0]), "+v"(c[o0][1]), "+v"(c[o0][2]), "+v"(c[o1][0]), "+v"(c[o1][1]), "+v"(c[o1][2])                          \
                 : "v"(a0x), "v"(a0y), "v"(a1x), "v"(a1y), "v"(bx), "v"(by)                                                                \
                 : "vcc");                                                                                                                 \
  else                                                                                                                                     \
  asm volatile(FX3_P(0, 6, 12, 16) FX3_P(3, 9, 14, 16) FX3_P(1, 7, 12, 17) FX3_P(4, 10, 14, 17) FX3_P(1, 7, 13, 16) FX3_P(4, 10, 15, 16)   \
                 FX3_P(2, 8, 13, 17) FX3_P(5, 11, 15, 17)                                                                                  \
               : "+v"(c[o0][0]), "+v"(c[o0][1]), "+v"(c[o0][2]), "+v"(c[o1][0]), "+v"(c[o1][1]), "+v"(c[o1][2]), "+v"(h[o0][0]),           \
                 "+v"(h[o0][1]), "+v"(h[o0][2]), "+v"(h[o1][0]), "+v"(h[o1][1]), "+v"(h[o1][2])                                            \
               : "v"(a0x), "v"(a0y), "v"(a1x), "v"(a1y), "v"(bx), "v"(by)                                                                  \
               : "vcc")
#define FX3_MAC_ONE(o0, ax, ay, bx, by)                                                                                                    \
  if constexpr(LAZY)                                                                                                                       \
    asm volatile(FX3_L(0, 3, 5) FX3_L(1, 3, 6) FX3_L(1, 4, 5) FX3_L(2, 4, 6)                                                               \
                 : "+v"(c[o0][0]), "+v"(c[o0][1]), "+v"(c[o0][2])                                                                          \
                 : "v"(ax), "v"(ay), "v"(bx), "v"(by)                                                                                      \
                 : "vcc");                                                                                                                 \
  else                                                                                                                                     \
  asm volatile(FX3_P(0, 3, 6, 8) FX3_P(1, 4, 6, 9) FX3_P(1, 4, 7, 8) FX3_P(2, 5, 7, 9)                                                     \
               : "+v"(c[o0][0]), "+v"(c[o0][1]), "+v"(c[o0][2]), "+v"(h[o0][0]), "+v"(h[o0][1]), "+v"(h[o0][2])                            \
               : "v"(ax), "v"(ay), "v"(bx), "v"(by)                                                                                        \
               : "vcc")
#define FX3_ROW(O0, O1)                                                                                                                    \
  {                                                                                                                                        \
    u32x4 va, vb; /* both pieces of either operand, also where MASK needs one (their columns are staged all the same) */                   \
    asm volatile("ds_read2_b64 %0, %2 offset0:" #O0 " offset1:" #O1 "\n\tds_read2_b64 %1, %3 offset0:" #O0 " offset1:" #O1                \
                 "\n\ts_waitcnt lgkmcnt(0)"                                                                                                \
                 : "=&v"(va), "=&v"(vb)                                                                                                    \
                 : "v"(xa), "v"(xb)                                                                                                        \
                 : "memory");                                                                                                              \
    if constexpr(MASK == 15)                                                                                                               \
      {                                                                                                                                    \
        FX3_MAC_BOTH(0, 1, va.x, va.y, va.z, va.w, vb.x, vb.y);                                                                            \
        FX3_MAC_BOTH(2, 3, va.x, va.y, va.z, va.w, vb.z, vb.w);                                                                            \
      }                                                                                                                                    \
    else if constexpr(MASK == 11)                                                                                                          \
      {                                                                                                                                    \
        FX3_MAC_BOTH(0, 1, va.x, va.y, va.z, va.w, vb.x, vb.y);                                                                            \
        FX3_MAC_ONE(3, va.z, va.w, vb.z, vb.w);                                                                                            \
      }                                                                                                                                    \
    else if constexpr(MASK == 5)                                                                                                           \
      {                                                                                                                                    \
        FX3_MAC_ONE(0, va.x, va.y, vb.x, vb.y);                                                                                            \
        FX3_MAC_ONE(2, va.x, va.y, vb.z, vb.w);                                                                                            \
      }                                                                                                                                    \
    else                                                                                                                                   \
      FX3_MAC_ONE(0, va.x, va.y, vb.x, vb.y);                                                                                              \
  }
    static_assert(RBG % 4 == 0 && M3 == 2, "four rows per trip, offsets in units of 8 bytes");
#if SDPB_SYRK3_PREFETCH
    // The reads of row r + 1 are issued before the products of row r: two register sets, each handed from the
    // statement that issues its reads to the statement that waits for them without the compiler touching it in between.
#define FX3_MACS(va, vb)                                                                                                                   \
  if constexpr(MASK == 15)                                                                                                                 \
    {                                                                                                                                      \
      FX3_MAC_BOTH(0, 1, va.x, va.y, va.z, va.w, vb.x, vb.y);                                                                              \
      FX3_MAC_BOTH(2, 3, va.x, va.y, va.z, va.w, vb.z, vb.w);                                                                              \
    }                                                                                                                                      \
  else if constexpr(MASK == 11)                                                                                                            \
    {                                                                                                                                      \
      FX3_MAC_BOTH(0, 1, va.x, va.y, va.z, va.w, vb.x, vb.y);                                                                              \
      FX3_MAC_ONE(3, va.z, va.w, vb.z, vb.w);                                                                                              \
    }                                                                                                                                      \
  else if constexpr(MASK == 5)                                                                                                             \
    {                                                                                                                                      \
      FX3_MAC_ONE(0, va.x, va.y, vb.x, vb.y);                                                                                              \
      FX3_MAC_ONE(2, va.x, va.y, vb.z, vb.w);                                                                                              \
    }                                                                                                                                      \
  else                                                                                                                                     \
    FX3_MAC_ONE(0, va.x, va.y, vb.x, vb.y);
    // wait for the set (wa, wb) that is in flight, then issue the reads of (ra, rb)
#ifdef SDPB_SYRK3_EXPERIMENT_HALF_LDS
    // TIMING EXPERIMENT ONLY (wrong results): the pieces of the second operand are not read -- what the kernel would cost
    // with half of its LDS reads (profiles/r06_syrk_lds_experiment.txt)
#define FX3_NEXT(wa, wb, ra, rb, O0, O1)                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read2_b64 %2, %4 offset0:" #O0 " offset1:" #O1                                                  \
               : "+v"(wa), "+v"(wb), "=&v"(ra), "+v"(rb)                                                                                   \
               : "v"(xa), "v"(xb)                                                                                                          \
               : "memory")
#else
#define FX3_NEXT(wa, wb, ra, rb, O0, O1)                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read2_b64 %2, %4 offset0:" #O0 " offset1:" #O1 "\n\tds_read2_b64 %3, %5 offset0:" #O0            \
               " offset1:" #O1                                                                                                             \
               : "+v"(wa), "+v"(wb), "=&v"(ra), "=&v"(rb)                                                                                  \
               : "v"(xa), "v"(xb)                                                                                                          \
               : "memory")
#endif
    u32x4 a0, b0, a1, b1;
    {
      const uint32_t xa = la, xb = lb;
      asm volatile("ds_read2_b64 %0, %2 offset0:0 offset1:16\n\tds_read2_b64 %1, %3 offset0:0 offset1:16" : "=&v"(a0), "=&v"(b0) : "v"(xa), "v"(xb) : "memory");
    }
#pragma unroll 1
    for(int rr = 0; rr < RBG; rr += 4)
      {
        const uint32_t xa = la + rr * 256, xb = lb + rr * 256; // a staged row is 64 words
        FX3_NEXT(a0, b0, a1, b1, 32, 48);
        FX3_MACS(a0, b0)
        FX3_NEXT(a1, b1, a0, b0, 64, 80);
        FX3_MACS(a1, b1)
        FX3_NEXT(a0, b0, a1, b1, 96, 112);
        FX3_MACS(a0, b0)
        // row rr + 4; past the last row of the pass this reads the first row of the other buffer (or, behind sb, LDS
        // that is not ours): the data is dropped
        FX3_NEXT(a1, b1, a0, b0, 128, 144);
        FX3_MACS(a1, b1)
      }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(b0)::"memory");
#undef FX3_NEXT
#undef FX3_MACS
#else
#pragma unroll 1
    for(int rr = 0; rr < RBG; rr += 4)
      {
        const uint32_t xa = la + rr * 256, xb = lb + rr * 256; // a staged row is 64 words
        FX3_ROW(0, 16)
        FX3_ROW(32, 48)
        FX3_ROW(64, 80)
        FX3_ROW(96, 112)
      }
#endif
#undef FX3_ROW
#undef FX3_MAC_ONE
#undef FX3_MAC_BOTH
#undef FX3_P
#undef FX3_L
#endif
      }
    else
      {
        // column x of the tile sits at word M3 x of a staged row: the lane's two pieces are 16 columns apart
        const uint32_t *pa = sa + buf * NPAIR * 4 + li * M3, *pb = sb + buf * NPAIR * 4 + lj * M3;
#pragma unroll SDPB_SYRK2_UNROLL
        for(int rr = 0; rr < RBG; ++rr)
          {
            uint32_t a0[M3], a1[M3], b0[M3], b1[M3];
            piece_load<M3>(pa + rr * ROWW, a0);
            piece_load<M3>(pb + rr * ROWW, b0);
            if constexpr((MASK & 10) != 0)
              piece_load<M3>(pa + rr * ROWW + 16 * M3, a1);
            if constexpr((MASK & 12) != 0)
              piece_load<M3>(pb + rr * ROWW + 16 * M3, b1);
            if constexpr(LAZY)
              {
                auto lazy = [&](const uint32_t (&x)[M3], const uint32_t (&y)[M3], uint64_t (&cs)[2 * M3 - 1]) {
                  cs[0] += (uint64_t)x[0] * y[0];
                  cs[1] += (uint64_t)x[0] * y[1] + (uint64_t)x[1] * y[0];
                  cs[2] += (uint64_t)x[1] * y[1];
                };
                lazy(a0, b0, c[0]);
                if constexpr((MASK & 2) != 0)
                  lazy(a1, b0, c[1]);
                if constexpr((MASK & 4) != 0)
                  lazy(a0, b1, c[2]);
                if constexpr((MASK & 8) != 0)
                  lazy(a1, b1, c[3]);
              }
            else
              {
            SyrkColumns<M3, 0>::run(a0, b0, c[0], h[0]);
            if constexpr((MASK & 2) != 0)
              SyrkColumns<M3, 0>::run(a1, b0, c[1], h[1]);
            if constexpr((MASK & 4) != 0)
              SyrkColumns<M3, 0>::run(a0, b1, c[2], h[2]);
            if constexpr((MASK & 8) != 0)
              SyrkColumns<M3, 0>::run(a1, b1, c[3], h[3]);
              }
          }
      }
  };
  // One sweep over the split's rows per product: its accumulators are the only sums a lane holds (36 registers at M3 = 2,
  // 84 at M3 = 4); the folded sum (A3 limbs per output) goes to plane (split, product) of the output, and
  // k_syrk4_finish recombines the three products of a Toom-4 group.
  for(int prod = prod_begin; prod < prod_end; ++prod)
    {
#pragma unroll
      for(int o = 0; o < 4; ++o)
#pragma unroll
        for(int k = 0; k < 2 * M3 - 1; ++k)
          {
            cc[o][k] = 0;
            hh[o][k] = 0;
          }
#pragma unroll
      for(int o = 0; o < 4; ++o)
#pragma unroll
        for(int k = 0; k < 2 * M3 - 1; ++k)
          dd[o][k] = 0;
      // LAZY: column k of an output absorbs products < 2^56, two of them per row in the middle column.  Nothing is carried
      // from column to column during the sweep: the HIGH WORD of a column is moved to a second sum of the same column (an
      // add, an add-with-carry and a move per column: no 64-bit shifts, no masks) -- the middle column every 64 rows (it is
      // < 2^32 + 64 * 2 * 2^56 < 2^64 by then), the outer ones every 128 rows.  (First version of the round: carries from
      // column to column every 64 rows, 15 instructions per output with three 64-bit shifts; profiles/r06_syrk_light_carry.txt)
      auto carry_columns = [&](bool all) __attribute__((always_inline)) {
        if constexpr(LAZY)
          {
#pragma unroll
            for(int o = 0; o < 4; ++o)
              {
                dd[o][1] += cc[o][1] >> 32;
                cc[o][1] &= 0xffffffffull;
              }
            if(all)
              {
#pragma unroll
                for(int o = 0; o < 4; ++o)
                  {
                    dd[o][0] += cc[o][0] >> 32;
                    cc[o][0] &= 0xffffffffull;
                    dd[o][2] += cc[o][2] >> 32;
                    cc[o][2] &= 0xffffffffull;
                  }
              }
          }
      };
      if(prod > prod_begin)
        __syncthreads(); // every wavefront has left the last pass of the previous sweep (and what it over-fetched has landed)
      // the barrier that ends a pass: the rows of the NEXT pass have landed in LDS; with three buffers the requests of
      // the pass after it (2 GL per lane, the youngest) may still be in flight
      auto pass_barrier = [&]() __attribute__((always_inline)) {
        if constexpr(NBUF == 3)
          {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * GL) : "memory");
#endif
          }
        else
          __syncthreads();
      };
      fetch(prod, row_begin, 0);
      if constexpr(NBUF == 3)
        fetch(prod, row_begin + RBG < row_end ? row_begin + RBG : row_begin, 1);
      store(0);
      pass_barrier();
      // the row blocks of the sweep; the quadrant mask is chosen outside the loop, so that the accumulators stay in
      // the same registers from block to block
      auto sweep = [&](auto mask_c) __attribute__((always_inline)) {
        int buf = 0;
        unsigned pass = 0;
        for(unsigned r0 = row_begin; r0 < row_end; r0 += RBG)
          {
            // (after the last block of the sweep: a block that exists, staged and never read)
#ifdef SDPB_SYRK3_EXPERIMENT_NO_FETCH
            if(r0 == row_begin) // TIMING EXPERIMENT ONLY (wrong results): what the kernel costs when no pass waits for its rows
#endif
            {
              if constexpr(NBUF == 3)
                fetch(prod, r0 + 2 * RBG < row_end ? r0 + 2 * RBG : row_begin, buf >= 1 ? buf - 1 : 2);
              else
                fetch(prod, r0 + RBG < row_end ? r0 + RBG : row_begin, buf ^ 1);
            }
            rows(mask_c, buf, cc, hh);
            ++pass;
            if(LAZY && (pass & 1u) == 0) // every second pass
              carry_columns((pass & 3u) == 0);
            if constexpr(NBUF == 3)
              {
                pass_barrier();
                buf = buf == 2 ? 0 : buf + 1;
              }
            else
              {
                store(buf ^ 1);
                __syncthreads();
                buf ^= 1;
              }
          }
      };
      switch(mask)
        {
        case 15: sweep(std::integral_constant<int, 15>()); break;
        case 11: sweep(std::integral_constant<int, 11>()); break; // diagonal tile: quadrant (0, 1) lies above the diagonal
        case 5: sweep(std::integral_constant<int, 5>()); break;   // last tile row, fewer than 17 of its 32 rows inside N
        default: sweep(std::integral_constant<int, 1>()); break;  // ... and its diagonal tile
        }
#pragma unroll
      for(int o = 0; o < 4; ++o)
        {
          const int i = i0 + 16 * (o & 1), j = j0 + 16 * (o >> 1);
          if(!((mask >> o) & 1) || i >= N || j > i)
            continue;
          uint32_t g[A3];
#pragma unroll
          for(int k = 0; k < A3; ++k)
            g[k] = 0;
          if constexpr(LAZY)
            {
              // sum_k (c_k + d_k 2^32) 2^(28 k): below 2^124 (fewer than 2^12 rows of products < 2^112): 4 words
              unsigned __int128 v = 0;
#pragma unroll
              for(int k = 0; k < 2 * M3 - 1; ++k)
                v += ((unsigned __int128)cc[o][k] + ((unsigned __int128)dd[o][k] << 32)) << (T5_LB * k);
              g[0] = (uint32_t)v;
              g[1] = (uint32_t)(v >> 32);
              g[2] = (uint32_t)(v >> 64);
              g[3] = (uint32_t)(v >> 96);
            }
          else
          syrk_fold<M3, A3>(g, cc[o], hh[o]);
          const size_t at = (size_t)tile * 1024 + (size_t)(li + 16 * (o & 1)) + (size_t)(lj + 16 * (o >> 1)) * 32; // tile-packed
#pragma unroll
          for(int k = 0; k < A3; ++k)
            acc[(size_t)(prod * A3 + k) * acc_stride + at] = g[k];
        }
    }
}

// The row splits of k_syrk_fx3 added in place (into the planes of split 0), one lane per (product, output element): the
// 16 x 105 planes of C4 read once by 21 x N^2 / 2 lanes at full occupancy instead of by the N^2 / 2 lanes of
// k_syrk4_finish with their 196 registers (that kernel alone took 3.98 ms per launch with 16 splits; product + sum +
// finish: 98.5 -> 95.6 ms); A3 limbs hold the sum over ALL rows (< 2^32 of them).
template <int FX>
__global__ void __launch_bounds__(WG) k_syrk3_sum_splits(uint32_t *part, int nsplit, size_t part_stride, const uint32_t *tile_list, size_t total, int N,
                                                         int col0, int col1)
{
  constexpr int A3 = fx_part_limbs<FX>(), NPROD = fx_nprod<FX>(); // (grid.y = NPROD)
  const size_t pidx = (size_t)blockIdx.x * WG + threadIdx.x;
  const int prod = blockIdx.y;
  int i, j;
  if(pidx >= total || !syrk_packed_decode<32>(pidx, tile_list, N, col0, col1, i, j))
    return;
  uint32_t out[A3];
  uint64_t cy = 0;
#pragma unroll
  for(int k = 0; k < A3; ++k)
    {
#pragma unroll 8
      for(int s = 0; s < nsplit; ++s)
        cy += part[(((size_t)s * NPROD + prod) * A3 + k) * part_stride + pidx];
      out[k] = (uint32_t)cy;
      cy >>= 32;
    }
#pragma unroll
  for(int k = 0; k < A3; ++k)
    part[((size_t)prod * A3 + k) * part_stride + pidx] = out[k];
}

// ---- Toom-5 x Karatsuba: column sums and the finishing kernel (fx_toom5k) --------------------------------------------
// inverse of an odd constant modulo 2^32 (Newton)
constexpr uint32_t inv_mod_2_32(uint32_t c)
{
  uint32_t x = c; // correct to 3 bits
  for(int i = 0; i < 5; ++i)
    x *= 2u - c * x;
  return x;
}
constexpr int T5_Z = 10; // limbs of the signed integers of the interpolation
// the evaluated value lo + hi 2^55 of a point from its two stored halves (limbs of 28 bits), as four 32-bit words
MW_HD void toom5_value(const uint32_t *fx, size_t fx_stride, int point, size_t e, uint32_t (&w)[4])
{
  const PiecePair lo = *reinterpret_cast<const PiecePair *>(fx + ((size_t)(3 * point + 0) * fx_stride + e) * 2);
  const PiecePair hi = *reinterpret_cast<const PiecePair *>(fx + ((size_t)(3 * point + 1) * fx_stride + e) * 2);
  w[0] = lo.w[0] | (lo.w[1] << 28);
  w[1] = (lo.w[1] >> 4) | (hi.w[0] << 23);
  w[2] = (hi.w[0] >> 9) | (hi.w[1] << 19);
  w[3] = hi.w[1] >> 13;
}
// Stage 1 (the shape of k_fx_colsum2): workgroup (x, y) sums row slice y of 64 columns of the five groups a0, p(1), p(2),
// 16 p(1/2), a4 -- enough to recover the column sums of all five pieces -- into partial[y]; element (slice, u, k, n), u < 5,
// k < 5 at ((slice * 5 + u) * 5 + k) * N + n.
template <int FX>
__global__ void __launch_bounds__(WG) k_fx_colsum5(const uint32_t *fx, size_t fx_stride, unsigned nrows, int N, unsigned rows_per_slice, uint32_t *partial)
{
  constexpr int A = 5, G = 5;
  const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane, phase = threadIdx.x >> 6;
  const unsigned r_begin = blockIdx.y * rows_per_slice, r_end = (r_begin + rows_per_slice < nrows) ? r_begin + rows_per_slice : nrows;
  uint32_t sum[G][A];
#pragma unroll
  for(int u = 0; u < G; ++u)
#pragma unroll
    for(int k = 0; k < A; ++k)
      sum[u][k] = 0;
  if(col < N)
    for(unsigned r = r_begin + phase; r < r_end; r += 4)
      {
        const size_t e = (size_t)r * N + col;
#pragma unroll
        for(int u = 0; u < G; ++u)
          {
            const int point = u == 0 ? 0 : u == 1 ? 1 : u == 2 ? 3 : u == 3 ? 5 : 8;
            uint32_t src[4];
            toom5_value(fx, fx_stride, point, e, src);
            uint64_t cy = 0;
#pragma unroll
            for(int k = 0; k < A; ++k)
              {
                const uint64_t t = (uint64_t)sum[u][k] + (k < 4 ? src[k < 4 ? k : 0] : 0u) + cy;
                sum[u][k] = (uint32_t)t;
                cy = t >> 32;
              }
          }
      }
  __shared__ uint32_t sm[3 * G * A * 64];
  if(phase > 0)
    {
#pragma unroll
      for(int u = 0; u < G; ++u)
#pragma unroll
        for(int k = 0; k < A; ++k)
          sm[(((phase - 1) * G + u) * A + k) * 64 + lane] = sum[u][k];
    }
  __syncthreads();
  if(phase == 0 && col < N)
    {
      for(int p = 0; p < 3; ++p)
#pragma unroll
        for(int u = 0; u < G; ++u)
          {
            uint64_t cy = 0;
#pragma unroll
            for(int k = 0; k < A; ++k)
              {
                const uint64_t t = (uint64_t)sum[u][k] + sm[((p * G + u) * A + k) * 64 + lane] + cy;
                sum[u][k] = (uint32_t)t;
                cy = t >> 32;
              }
          }
#pragma unroll
      for(int u = 0; u < G; ++u)
#pragma unroll
        for(int k = 0; k < A; ++k)
          partial[(((size_t)blockIdx.y * G + u) * A + k) * N + col] = sum[u][k];
    }
}
// Stage 2: the column sums s_0 .. s_4 of the five pieces from the five group sums; S_n = sum_k s_k b^k behind the N x N block
// of acc; the bias terms of the three signed points, U = K E + n K^2 / 2 with E the column sum of the point's UNBIASED values:
//   toomU[(t Z + k) N + n], t = 0: p(-1) (K = 2 b), 1: p(-2) (K = 10 b), 2: 16 p(-1/2) (K = 10 b).
template <int FX>
__global__ void __launch_bounds__(WG)
  k_fx_colsum5_final(const uint32_t *partial, int nslices, int N, uint32_t *acc, size_t acc_stride, uint32_t *toomU, unsigned long long nrows_local)
{
  constexpr int A = 5, G = 5, W = 2 * FX + 2, WB = toom_wb<FX>(), Z = T5_Z;
  const int col = blockIdx.x * WG + threadIdx.x;
  if(col >= N)
    return;
  uint32_t g[G][Z];
#pragma unroll
  for(int u = 0; u < G; ++u)
#pragma unroll
    for(int k = 0; k < Z; ++k)
      g[u][k] = 0;
  for(int sl = 0; sl < nslices; ++sl)
#pragma unroll
    for(int u = 0; u < G; ++u)
      {
        uint64_t cy = 0;
#pragma unroll
        for(int k = 0; k < A + 1; ++k)
          {
            const uint64_t t = (uint64_t)g[u][k] + (k < A ? partial[(((size_t)sl * G + u) * A + (k < A ? k : 0)) * N + col] : 0u) + cy;
            g[u][k] = (uint32_t)t;
            cy = t >> 32;
          }
      }
  // g: 0 = s0, 1 = p(1), 2 = p(2), 3 = 16 p(1/2), 4 = s4
  uint32_t s[5][Z], u[Z], v[Z], t[Z];
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      s[0][k] = g[0][k];
      s[4][k] = g[4][k];
      u[k] = g[1][k];
      v[k] = g[2][k];
      t[k] = g[3][k];
    }
  z_sub<Z>(u, s[0]);
  z_sub<Z>(u, s[4]);                  // u = s1 + s2 + s3
  z_sub<Z>(v, s[0]);
  z_addmul<Z>(v, s[4], 16, true);     // v = 2 s1 + 4 s2 + 8 s3
  z_addmul<Z>(t, s[0], 16, true);
  z_sub<Z>(t, s[4]);                  // t = 8 s1 + 4 s2 + 2 s3
#pragma unroll
  for(int k = 0; k < Z; ++k)
    s[2][k] = 0;
  z_addmul<Z>(s[2], u, 10, false);
  z_sub<Z>(s[2], v);
  z_sub<Z>(s[2], t);
  z_sar<Z>(s[2], 1);                  // s2 = (10 u - v - t) / 2
  z_sub<Z>(v, t);
  z_sar<Z>(v, 1);
  z_divexact<Z>(v, 3u, inv_mod_2_32(3u)); // d = (v - t) / 6 = s3 - s1
  z_sub<Z>(u, s[2]);                  // u = s1 + s3
#pragma unroll
  for(int k = 0; k < Z; ++k)
    {
      s[3][k] = u[k];
      s[1][k] = u[k];
    }
  z_add<Z>(s[3], v);
  z_sar<Z>(s[3], 1);
  z_sub<Z>(s[1], v);
  z_sar<Z>(s[1], 1);
  {
    uint32_t w[W];
#pragma unroll
    for(int k = 0; k < W; ++k)
      w[k] = k < Z ? s[0][k < Z ? k : 0] : 0u;
#pragma unroll
    for(int q = 1; q < 5; ++q)
      add_shifted<W, Z>(w, s[q], q * WB, false);
#pragma unroll
    for(int k = 0; k < W; ++k)
      acc[(size_t)k * acc_stride + (size_t)N * N + col] = w[k];
  }
  uint32_t n[Z];
#pragma unroll
  for(int k = 0; k < Z; ++k)
    n[k] = k == 0 ? (uint32_t)nrows_local : k == 1 ? (uint32_t)(nrows_local >> 32) : 0u;
  auto emit = [&](int which, uint32_t c0, bool n0, uint32_t c1, bool n1, uint32_t c2, bool n2, uint32_t c3, bool n3, uint32_t c4, bool n4, bool ten) {
    uint32_t e[Z], m[Z];
#pragma unroll
    for(int k = 0; k < Z; ++k)
      e[k] = m[k] = 0;
    z_addmul<Z>(e, s[0], c0, n0);
    z_addmul<Z>(e, s[1], c1, n1);
    z_addmul<Z>(e, s[2], c2, n2);
    z_addmul<Z>(e, s[3], c3, n3);
    z_addmul<Z>(e, s[4], c4, n4);
    if(ten)
      {
        // U = 10 b E + 50 n b^2
#pragma unroll
        for(int k = 0; k < Z; ++k)
          t[k] = 0;
        z_addmul<Z>(t, e, 10, false);
        z_shl<Z>(t, WB);
        z_addmul<Z>(m, n, 50, false);
        z_shl<Z>(m, 2 * WB);
      }
    else
      {
        // U = 2 b E + 2 n b^2
#pragma unroll
        for(int k = 0; k < Z; ++k)
          {
            t[k] = e[k];
            m[k] = n[k];
          }
        z_shl<Z>(t, WB + 1);
        z_shl<Z>(m, 2 * WB + 1);
      }
    z_add<Z>(t, m);
#pragma unroll
    for(int k = 0; k < Z; ++k)
      toomU[(size_t)(which * Z + k) * N + col] = t[k];
  };
  emit(0, 1, false, 1, true, 1, false, 1, true, 1, false, false);   // E = s0 - s1 + s2 - s3 + s4
  emit(1, 1, false, 2, true, 4, false, 8, true, 16, false, true);   // E = s0 - 2 s1 + 4 s2 - 8 s3 + 16 s4
  emit(2, 16, false, 8, true, 4, false, 2, true, 1, false, true);   // E = 16 s0 - 8 s1 + 4 s2 - 2 s3 + s4
}
// acc(i,j) (i >= j) = G(i,j) = sum_r a'_ri a'_rj from the 27 product sums k_syrk_fx3 (lazy mode) left in part: add the
// splits, recombine the three products of a point (e e' = lo lo' + (mid mid' - lo lo' - hi hi') 2^H + hi hi' 2^(2H)), remove
// the bias of the three signed points, interpolate (c = M V / D row by row, exact divisions), recombine G = sum c_k b^k.
template <int FX>
__global__ void __launch_bounds__(WG)
  k_syrk5_finish(const uint32_t *part, int nsplit, size_t part_stride, const uint32_t *tile_list, size_t total, const uint32_t *toomU,
                 uint32_t *acc, size_t acc_stride, int N, int col0, int col1)
{
  constexpr int A3 = fx_part_limbs<FX>(), Z = T5_Z, W = 2 * FX + 2, WB = toom_wb<FX>(), H = T5_H;
  const size_t pidx = (size_t)blockIdx.x * WG + threadIdx.x;
  int i, j;
  if(pidx >= total || !syrk_packed_decode<syrk_tile_edge<FX>()>(pidx, tile_list, N, col0, col1, i, j))
    return;
  const size_t idx = (size_t)i + (size_t)j * N;
  uint32_t V[9][Z];
#pragma unroll
  for(int q = 0; q < 9; ++q)
    {
      uint32_t g3[3][A3 + 1];
#pragma unroll
      for(int u = 0; u < 3; ++u)
        {
          uint64_t cy = 0;
#pragma unroll
          for(int k = 0; k < A3; ++k)
            {
              for(int sp = 0; sp < nsplit; ++sp)
                cy += part[(((size_t)sp * 27 + 3 * q + u) * A3 + k) * part_stride + pidx];
              g3[u][k] = (uint32_t)cy;
              cy >>= 32;
            }
          g3[u][A3] = (uint32_t)cy;
        }
      sub_limbs<A3 + 1>(g3[2], g3[0]);
      sub_limbs<A3 + 1>(g3[2], g3[1]);
#pragma unroll
      for(int k = 0; k < Z; ++k)
        V[q][k] = k < A3 + 1 ? g3[0][k < A3 + 1 ? k : 0] : 0u;
      add_shifted<Z, A3 + 1>(V[q], g3[2], H, false);
      add_shifted<Z, A3 + 1>(V[q], g3[1], 2 * H, false);
    }
  {
    uint32_t u[Z];
#pragma unroll
    for(int t = 0; t < 3; ++t)
      {
        const int q = 2 + 2 * t; // points 2, 4, 6
#pragma unroll
        for(int k = 0; k < Z; ++k)
          u[k] = toomU[(size_t)(t * Z + k) * N + i];
        z_sub<Z>(V[q], u);
#pragma unroll
        for(int k = 0; k < Z; ++k)
          u[k] = toomU[(size_t)(t * Z + k) * N + j];
        z_sub<Z>(V[q], u);
      }
  }
  // rows 1 .. 7 of the inverse of the evaluation matrix (points 0, 1, -1, 2, -2, 1/2, -1/2, 3, inf; the halves scaled by 2^8),
  // each over its common denominator D = 2^SH * ODD (profiles/tools/toom5_matrix.py)
  constexpr int MI[7][9] = {{-700, -700, 350, 35, -7, 14, -10, -2, 6300},      {-1890, -80, -80, 1, 1, 4, 4, 0, -360},
                            {3150, 2750, -1175, -155, 29, -23, 5, 9, -28350},  {378, 68, 68, -1, -1, -1, -1, 0, 378},
                            {-3150, -1450, -125, 145, -19, 13, 5, -9, 28350},  {-360, -80, -80, 4, 4, 1, 1, 0, -1890},
                            {2100, 700, 350, -70, -14, -7, -5, 6, -18900}};
  constexpr int SH[7] = {2, 3, 3, 3, 3, 3, 2};
  constexpr uint32_t ODD[7] = {525, 45, 225, 9, 225, 45, 1575};
  uint32_t gg[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    gg[k] = k < Z ? V[0][k < Z ? k : 0] : 0u;
#pragma unroll
  for(int r = 0; r < 7; ++r)
    {
      uint32_t c[Z];
#pragma unroll
      for(int k = 0; k < Z; ++k)
        c[k] = 0;
#pragma unroll
      for(int q = 0; q < 9; ++q)
        if(MI[r][q] != 0)
          z_addmul<Z>(c, V[q], (uint32_t)(MI[r][q] < 0 ? -MI[r][q] : MI[r][q]), MI[r][q] < 0);
      z_sar<Z>(c, SH[r]);
      z_divexact<Z>(c, ODD[r], inv_mod_2_32(ODD[r]));
      add_shifted<W, Z>(gg, c, (r + 1) * WB, false);
    }
  add_shifted<W, Z>(gg, V[8], 8 * WB, false);
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + idx] = gg[k];
}

template <int FX> constexpr int syrk_waves_per_simd() { return fx_toom4k<FX>() ? SDPB_SYRK3_WAVES : SDPB_SYRK_WAVES; }

// dst += src on the lower triangle (i >= j) and on the N column sums behind the N x N block: the partial G of two
// INPUT windows (row chunks of the fixed-point image; the reference loops over its input windows in
// bigint_syrk_blas.cxx:239-285 and lets BLAS accumulate into the output window) are plain non-negative integers
// of W planes, so they add exactly -- the same sum the cross-GPU all-reduce forms over the ranks' rows.
template <int W> __global__ void __launch_bounds__(WG) k_acc_add_tri(uint32_t *dst, const uint32_t *src, size_t stride, int N)
{
  const size_t idx = (size_t)blockIdx.x * WG + threadIdx.x;
  if(idx >= (size_t)N * N + N)
    return;
  if(idx < (size_t)N * N && idx % N < idx / N)
    return;
  uint64_t cy = 0;
#pragma unroll
  for(int k = 0; k < W; ++k)
    {
      const uint64_t t = (uint64_t)dst[(size_t)k * stride + idx] + src[(size_t)k * stride + idx] + cy;
      dst[(size_t)k * stride + idx] = (uint32_t)t;
      cy = t >> 32;
    }
}

// Remove the bias in place (after any cross-GPU sum): for i >= j
//   acc(i,j) <- G(i,j) - C (S_i + S_j) + n C^2 = sum_r v_ri v_rj   (two's complement),
// C = 2^FB, n = total number of rows, S behind the N x N block (k_fx_colsum).
template <int FX>
__global__ void __launch_bounds__(WG) k_syrk_unbias(uint32_t *acc, size_t acc_stride, int N, unsigned long long nrows_total, size_t idx0, size_t idx1)
{
  constexpr int W = 2 * FX + 2, FB = fx_frac_bits<FX>();
  const size_t idx = idx0 + (size_t)blockIdx.x * WG + threadIdx.x;
  if(idx >= idx1)
    return;
  const int i = (int)(idx % N), j = (int)(idx / N);
  if(i < j)
    return;
  uint32_t w[W], s[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    w[k] = acc[(size_t)k * acc_stride + idx];
  {
    uint64_t cy = 0;
#pragma unroll
    for(int k = 0; k < W; ++k)
      {
        const uint64_t t = (uint64_t)acc[(size_t)k * acc_stride + (size_t)N * N + i] + acc[(size_t)k * acc_stride + (size_t)N * N + j] + cy;
        s[k] = (uint32_t)t;
        cy = t >> 32;
      }
  }
  add_shifted<W, W>(w, s, FB, true);
  {
    const uint32_t nn[2] = {(uint32_t)nrows_total, (uint32_t)(nrows_total >> 32)};
    add_shifted<W, 2>(w, nn, 2 * FB, false);
  }
#pragma unroll
  for(int k = 0; k < W; ++k)
    acc[(size_t)k * acc_stride + idx] = w[k];
}

// Q(i,j) = (acc(i,j) / 2^(2FB)) * norm_i * norm_j for i >= j (acc unbiased, signed);
// checks the diagonal (check_normalized_Q_diagonal, compute_Q.cxx:65-91):
// |Q'_ii - 1| < 2^(-16 FX).
template <int NL, int FX>
__global__ void __launch_bounds__(WG)
  k_restore_Q(const uint32_t *acc, size_t acc_stride, int N, mw::CPtr norms, mw::Ptr Q, int *diag_fail, size_t idx0, size_t idx1)
{
  constexpr int W = 2 * FX + 2, FB = fx_frac_bits<FX>();
  const size_t idx = idx0 + (size_t)blockIdx.x * WG + threadIdx.x;
  if(idx >= idx1)
    return;
  const int i = (int)(idx % N), j = (int)(idx / N);
  if(i < j)
    {
      mw::store<NL>(Q, idx, mw::zero<NL>());
      return;
    }
  uint32_t w[W];
#pragma unroll
  for(int k = 0; k < W; ++k)
    w[k] = acc[(size_t)k * acc_stride + idx];
  const uint32_t negative = w[W - 1] >> 31;
  if(negative)
    {
      uint32_t carry = 1;
#pragma unroll
      for(int k = 0; k < W; ++k)
        {
          const uint64_t t = (uint64_t)(~w[k]) + carry;
          w[k] = (uint32_t)t;
          carry = (uint32_t)(t >> 32);
        }
    }
  // magnitude w as a W-limb integer; value = w / 2^(2FB) = (w / 2^(32W)) * 2^(32W - 2FB)
  uint32_t zl = 0, topw = 0;
  bool found = false;
#pragma unroll
  for(int k = W - 1; k >= 0; --k)
    {
      const bool nz = w[k] != 0;
      if(!found && nz)
        topw = w[k];
      if(!found && !nz)
        zl += 1;
      found = found || nz;
    }
  Mw<NL> v = mw::zero<NL>();
  if(found)
    {
      const uint32_t c = mw::clz32(topw);
      mw::shl_limbs<W>(w, zl);
      mw::shl_bits<W>(w, c);
#pragma unroll
      for(int k = 0; k < NL; ++k)
        v.m[k] = (W - NL + k >= 0) ? w[W - NL + k >= 0 ? W - NL + k : 0] : 0u;
      v.e = 32 * W - 2 * FB - (int32_t)(32u * zl + c);
      v.neg = negative;
    }
  const Mw<NL> ni = mw::load<NL>(norms, i);
  if(i == j && !mw::is_zero(ni))
    {
      const Mw<NL> diff = mw::abs(mw::sub(v, mw::from_u32<NL>(1)));
      if(!mw::is_zero(diff) && diff.e > -16 * (NL - 2)) // the reference's 2^-(precision/2), precision = 64 (l - 1) = 32 (NL - 2): not the padded image width FX
        atomicMax(diag_fail, i + 1);
    }
  v = mw::mul(mw::mul(v, ni), mw::load<NL>(norms, j));
  mw::store<NL>(Q, idx, v);
}

// ---------------------------------------------------------------------------
// Smallest eigenvalue of a batch of symmetric matrices (in place, lower triangle
// referenced).  Replaces El::HermitianEig + El::Min at min_eigenvalue.cxx:8-33.
// Like Elemental: Householder reduction to tridiagonal form (one wavefront per
// matrix, reductions through LDS; k_tridiag), then only the eigenvalue that is needed
// (k_tridiag_min, one lane per matrix):
// lambda_min of the tridiagonal matrix by an fp64 Sturm bisection (a safe starting
// point below the spectrum) refined with Newton's method on det(T - lambda) in full
// multi-word precision — monotone from below, quadratically convergent.
// lam[q] = lambda_min (zero-size matrices write +huge so they never win the MIN).
// ---------------------------------------------------------------------------
constexpr int EIG_T = 64; // lanes per workgroup of the one-lane-per-matrix stage

// number of eigenvalues of the tridiagonal (a, b2 = offdiag^2) below x, in fp64
__device__ inline int sturm_count_f64(const double *a, const double *b2, int n, double x)
{
  int cnt = 0;
  double q = a[0] - x;
  if(q < 0)
    ++cnt;
  for(int i = 1; i < n; ++i)
    {
      if(q == 0.0)
        q = 1e-300;
      q = a[i] - x - b2[i] / q;
      if(q < 0)
        ++cnt;
    }
  return cnt;
}

// Stage 1: Householder tridiagonalisation (EISPACK tred1 organisation), one workgroup
// of TRI_T lanes per matrix.  D = diagonal, E[1..n) = off-diagonal of the tridiagonal
// matrix.  Row dot products are split over teams of lanes and meet in LDS so that the
// dependent chain per Householder step stays short.
// 128 lanes per matrix: with 242 VGPRs a 256-lane workgroup leaves room for 512 of C4's 1200 matrices at a time
// (2.3 rounds of a latency-bound kernel), 128 lanes for 1024 (1.2 rounds): step lengths 14.2 -> 10.7 ms (64: 13.3)
#ifndef SDPB_TRI_T
#define SDPB_TRI_T 128
#endif
constexpr int TRI_T = SDPB_TRI_T;
#ifndef SDPB_TRI_WAVES
#define SDPB_TRI_WAVES 2
#endif
// (A two-level exact sum through a limb-major image, as in k_gemv_n, was measured here and is slower:
// 4.74 instead of 4.61 ms per launch; every lane needs the result, and the kernel is bound by the
// wavefront-instructions it issues, not by the depth of this tree.)
// Round 5: every sum of the kernel has ONE shape whatever T is (round-4 advisor: lambda_min depended on the team width,
// which is picked from the CU count and from how many matrices a rank owns, in its last bits): the dot products of a
// Householder step are taken by the first TRI_V lanes (term k by lane k mod TRI_V, a tree over TRI_V leaves -- adding the
// exact zeros of idle lanes changes nothing), and the team width of the row products depends on the row length only.
// (Measured and dropped, profiles/r05_tridiag_lanes.txt: an odd word stride for the tree's LDS image -- Mw<18> is 20 words,
// 32 lanes fall on 8 banks -- turns its 16-byte LDS accesses into 4-byte ones and costs 0.8 ms per iteration; the
// conflicts it would remove are 19 M of 95 M LDS-active cycles summed over 256 CUs, i.e. LDS is busy 3 % of this kernel.)
constexpr int TRI_V = 64;
template <int NL> struct TriSlot
{
  Mw<NL> v;
#ifdef SDPB_TRI_PAD
  uint32_t pad[(sizeof(Mw<NL>) / 4) % 2 == 0 ? 1 : 2]; // the measured variant: odd stride in words
#endif
};
template <int NL> __device__ Mw<NL> tri_reduce_sum(const Mw<NL> &v, TriSlot<NL> *sm)
{
  // v: the contributions of the lanes t < TRI_V (other lanes pass anything); every lane gets the sum
  const int t = threadIdx.x;
  if(t < TRI_V)
    sm[t].v = v;
  __syncthreads();
  for(int s = TRI_V / 2; s > 0; s >>= 1)
    {
      if(t < s)
        sm[t].v = mw::add(sm[t].v, sm[t + s].v);
      __syncthreads();
    }
  const Mw<NL> r = sm[0].v;
  __syncthreads();
  return r;
}
// T lanes per matrix (64 ... 512), chosen per SIZE BUCKET of the batch and by how many matrices the rank owns (host:
// Solver::enqueue_min_eigenvalues): blockIdx -> matrix through `ids`.  The bits of the result do not depend on T.
template <int NL, int T>
__global__ void __launch_bounds__(T, T <= 256 ? SDPB_TRI_WAVES : 1024 / T) k_tridiag(Batch A, Batch D, Batch E, const int *ids)
{
  const int q = ids[blockIdx.x];
  const MatDesc d = A.d[q];
  const size_t od = (size_t)D.d[q].off, oe = (size_t)E.d[q].off;
  const int n = d.rows, t = threadIdx.x;
  if(n == 0)
    return;
  __shared__ TriSlot<NL> sm[T];
  __shared__ Mw<NL> s_hinv;
  for(int i = n - 1; i >= 1; --i)
    {
      const int l = i - 1;
      if(l == 0)
        {
          if(t == 0)
            mw::store<NL>(E.p, oe + i, mat_ld<NL>(A, d, i, l));
          continue;
        }
      Acc<NL> hp = mw::acc_zero<NL>();
      if(t < TRI_V)
        for(int k = t; k <= l; k += TRI_V)
          {
            const Mw<NL> a = mat_ld<NL>(A, d, i, k);
            mw::acc_fma(hp, a, a);
          }
      const Mw<NL> h0 = tri_reduce_sum<NL>(mw::acc_result(hp), sm);
      if(mw::is_zero(h0))
        {
          if(t == 0)
            mw::store<NL>(E.p, oe + i, mat_ld<NL>(A, d, i, l));
          continue;
        }
#if defined(__HIP_DEVICE_COMPILE__)
      // the square root and the reciprocal of a Householder step are ONE number each — 11 + 4 us on one lane at 576
      // bits, the longest link of the step: the first wavefront computes them together (mw_wave.hpp, same bits)
      if(t < 64)
        {
          const Mw<NL> f = mat_ld<NL>(A, d, i, l);
          Mw<NL> g = mw::wv::sqrt<NL>(h0, 0);
          if(!f.neg)
            g = mw::neg(g);
          const Mw<NL> h = mw::sub(h0, mw::mul(f, g));
          const Mw<NL> hinv1 = mw::wv::rcp<NL>(h, 0);
          if(t == 0)
            {
              mw::store<NL>(E.p, oe + i, g);
              mat_st<NL>(A, d, i, l, mw::sub(f, g));
              s_hinv = hinv1;
            }
        }
#else
      if(t == 0)
        {
          const Mw<NL> f = mat_ld<NL>(A, d, i, l);
          Mw<NL> g = mw::sqrt(h0);
          if(!f.neg)
            g = mw::neg(g);
          mw::store<NL>(E.p, oe + i, g);
          const Mw<NL> h = mw::sub(h0, mw::mul(f, g));
          mat_st<NL>(A, d, i, l, mw::sub(f, g));
          s_hinv = mw::rcp(h);
        }
#endif
      __syncthreads();
      const Mw<NL> hinv = s_hinv;
      // e[j] = (A_sub u)_j / h with teams of G lanes per row j; G from the row length alone (not from T)
      const int w = l + 1;
      int G = 128 / w;
      G = G < 1 ? 1 : (G > 8 ? 8 : G);
      const int teams = T / G, g = t % G;
      for(int j0 = 0; j0 < w; j0 += teams)
        {
          const int j = j0 + t / G;
          Acc<NL> acc = mw::acc_zero<NL>();
          if(j < w && t / G < teams)
            for(int k = g; k < w; k += G)
              {
                const Mw<NL> ajk = k <= j ? mat_ld<NL>(A, d, j, k) : mat_ld<NL>(A, d, k, j);
                mw::acc_fma(acc, ajk, mat_ld<NL>(A, d, i, k));
              }
          sm[t].v = mw::acc_result(acc);
          __syncthreads();
          if(g == 0 && j < w && t / G < teams)
            {
              Mw<NL> sum = sm[t].v;
              for(int gg = 1; gg < G; ++gg)
                sum = mw::add(sum, sm[t + gg].v);
              mw::store<NL>(E.p, oe + j, mw::mul(sum, hinv));
            }
          __syncthreads();
        }
      // f = sum_j e_j a_ij, the terms dealt to the first TRI_V lanes like those of h0
      Acc<NL> fp = mw::acc_zero<NL>();
      if(t < TRI_V)
        for(int j = t; j < w; j += TRI_V)
          mw::acc_fma(fp, mw::load<NL>(E.p, oe + j), mat_ld<NL>(A, d, i, j));
      const Mw<NL> f = tri_reduce_sum<NL>(mw::acc_result(fp), sm);
      const Mw<NL> hh = mw::mul_2exp(mw::mul(f, hinv), -1);
      for(int j = t; j < w; j += T)
        mw::store<NL>(E.p, oe + j, mw::sub(mw::load<NL>(E.p, oe + j), mw::mul(hh, mat_ld<NL>(A, d, i, j))));
      __syncthreads();
      // rank-2 update of the leading w x w lower triangle (packed index -> (j,k), k <= j)
      const int cnt = w * (w + 1) / 2;
      for(int idx = t; idx < cnt; idx += T)
        {
          int j = 0;
          while((j + 1) * (j + 2) / 2 <= idx)
            ++j;
          const int k = idx - j * (j + 1) / 2;
          Acc<NL> v = mw::acc_zero<NL>();
          mw::acc_add(v, mat_ld<NL>(A, d, j, k));
          mw::acc_fms(v, mat_ld<NL>(A, d, i, j), mw::load<NL>(E.p, oe + k));
          mw::acc_fms(v, mw::load<NL>(E.p, oe + j), mat_ld<NL>(A, d, i, k));
          mat_st<NL>(A, d, j, k, mw::acc_result(v));
        }
      __syncthreads();
    }
  __syncthreads();
  for(int i = t; i < n; i += T)
    mw::store<NL>(D.p, od + i, mat_ld<NL>(A, d, i, i));
}

// Newton iteration for lambda_min of the tridiagonal (D, E2 = offdiag^2) from below,
// working at WL limbs (operands are narrowed on load).  Monotone from below in exact
// arithmetic; quadratically convergent.
// Round 5: the leading principal minors p_i = det(T_i - x) and their derivatives by the three-term recurrence
//   p_{i+1} = (d_i - x) p_i - e_i^2 p_{i-1},   p'_{i+1} = (d_i - x) p'_i - p_i - e_i^2 p'_{i-1}
// -- four products and two normalisations per row instead of a reciprocal (five to seven products deep) and three
// products; ONE reciprocal per Newton step.  The exponent of Mw is an int32: the minors of a 40 x 40 matrix cannot leave
// it.  x below the spectrum <=> every p_i > 0.  Each step of the recurrence is one rounding of (d_i - x) and of e_i^2,
// as in the quotient form.  k_tridiag_min 2.37 -> 1.99 ms per launch on C4 (profiles/r05_tridiag_min.txt: 0.68 ms of it
// are the fp64 bisection, 0.58 / 0.34 / 0.40 ms the 6-, 10- and 18-limb rungs); -DSDPB_TRIMIN_POLY=0: the quotient form.
#ifndef SDPB_TRIMIN_POLY
#define SDPB_TRIMIN_POLY 1
#endif
template <int WL, int NL>
__device__ void tridiag_newton(const Batch &D, const Batch &E, size_t od, size_t oe, int n, Mw<WL> &x, double span, int emax,
                               int backoff_bits, bool &clustered)
{
  // clustered (in/out): a narrower rung of the ladder met slow convergence -- then a first step that is already small says
  // nothing about the distance to go (it is that distance over m), and the rung iterates until two steps show the ratio
  const Mw<WL> minus_one = mw::from_i32<WL>(-1);
  Mw<WL> prev = x;
  // A CLUSTER of m eigenvalues at the bottom of the spectrum, tighter than the distance still to go, makes the Newton step
  // 1/m of that distance: linear convergence with ratio 1 - 1/m (round 6: half of the spectrum within 2^-70 of lambda_min kept
  // the iteration at 2^-73 when its cap of 100 steps ran out).  Two consecutive steps give the ratio and with it the
  // multiplicity the iteration SEES from where it stands: a step of `mul` Newton steps towards a root of multiplicity M shrinks
  // the next step by 1 - mul / M.  The step M delta undershoots the cluster only by the second-order term (quadratic
  // convergence towards the cluster); a multiplicity that was overestimated -- the cluster has a width, and from nearer by
  // fewer of its eigenvalues count -- shows as a non-positive minor at the next evaluation: the step is taken again from the
  // same point with three quarters of the multiplier, and that cap stays (what the iteration sees only decreases on the way).
  // All thresholds are relative to the larger of |x| and the scale of the matrix (2^emax): after the shift of k_tridiag_min
  // lambda_min can be tiny against the entries, and the noise of the minors is relative to the entries.
  Mw<WL> pd = mw::zero<WL>(), base = x, base_delta = mw::zero<WL>();
  bool have_pd = false;
  int pmul = 1, mul = 1, mcap = n; // multiplier of the previous step, of the step under test, and the cap
  bool accel_pending = false;
  for(int it = 0; it < 100; ++it)
    {
      Mw<WL> S = mw::zero<WL>(), inv = mw::zero<WL>(), tq = mw::zero<WL>();
      bool overshoot = false;
#if SDPB_TRIMIN_POLY
      {
        Mw<WL> pm = mw::from_i32<WL>(1), pmm = mw::zero<WL>(), dpm = mw::zero<WL>(), dpmm = mw::zero<WL>();
        for(int i = 0; i < n; ++i)
          {
            const Mw<WL> di = mw::sub(mw::narrow<WL, NL>(mw::load<NL>(D.p, od + i)), x);
            Acc<WL> ap = mw::acc_zero<WL>(), ad = mw::acc_zero<WL>();
            mw::acc_fma(ap, di, pm);
            mw::acc_fma(ad, di, dpm);
            mw::acc_add(ad, pm, 1u);
            if(i >= 1)
              {
                const Mw<WL> e2 = mw::narrow<WL, NL>(mw::load<NL>(E.p, oe + i));
                mw::acc_fms(ap, e2, pmm);
                mw::acc_fms(ad, e2, dpmm);
              }
            const Mw<WL> pi = mw::acc_result(ap);
            if(pi.neg || mw::is_zero(pi))
              {
                overshoot = true;
                break;
              }
            pmm = pm;
            dpmm = dpm;
            pm = pi;
            dpm = mw::acc_result(ad);
          }
        if(!overshoot && !mw::is_zero(dpm))
          S = mw::mul(dpm, mw::rcp(pm)); // p'/p = sum_i q_i'/q_i
      }
#else
      for(int i = 0; i < n; ++i)
        {
          Mw<WL> qi = mw::sub(mw::narrow<WL, NL>(mw::load<NL>(D.p, od + i)), x), qp = minus_one;
          if(i >= 1)
            {
              const Mw<WL> u = mw::mul(mw::narrow<WL, NL>(mw::load<NL>(E.p, oe + i)), inv);
              qi = mw::sub(qi, u);
              qp = mw::add(minus_one, mw::mul(u, tq));
            }
          if(qi.neg || mw::is_zero(qi))
            {
              overshoot = true;
              break;
            }
          inv = mw::rcp(qi);
          tq = mw::mul(qp, inv);
          S = mw::add(S, tq);
        }
#endif
#ifdef SDPB_TRACE_TRIMIN
      if(overshoot)
        printf("WL=%d it=%d overshoot accel_pending=%d x.e=%d\n", WL, it, (int)accel_pending, x.e);
#endif
      if(overshoot && accel_pending)
        {
          // the multiplicity was overestimated: again from the same point with a smaller multiplier (1: the plain Newton step,
          // which cannot cross the root)
          mcap = mul - (mul / 4 > 1 ? mul / 4 : 1);
          mul = mcap;
          accel_pending = mul >= 2;
          x = mw::add(base, mul >= 2 ? mw::mul(base_delta, mw::from_u32<WL>((uint32_t)mul)) : base_delta);
          prev = base;
          pd = base_delta;
          pmul = mul;
          have_pd = true;
          continue;
        }
      accel_pending = false;
      if(overshoot)
        {
          // Newton from below never crosses the root in exact arithmetic: a non-positive
          // pivot after a successful step means x sits within rounding noise of
          // lambda_min; keep the last point that was below.
          if(it > 0 && mw::cmp(x, prev) != 0)
            {
              x = prev;
              return;
            }
          // the start was not below the spectrum: retreat, 256x further on every retry
          Mw<WL> back = mw::abs(x);
          if(mw::is_zero(back) || back.e < emax)
            back = mw::mul_2exp(mw::from_double<WL>(span), emax);
          back.e -= backoff_bits - 8 * it;
          x = mw::sub(x, back);
          prev = x;
          have_pd = false;
          continue;
        }
      if(mw::is_zero(S))
        return;
      const Mw<WL> delta = mw::neg(mw::rcp(S));
      // ratio of this step to the previous one: ~ 0 while the convergence is quadratic, 1 - pmul / M before a cluster of M
      double ratio = 0.0;
      if(have_pd && !mw::is_zero(pd) && !mw::is_zero(delta) && delta.e - pd.e > -60)
        {
          ratio = mw::to_double(mw::mul_2exp(delta, -delta.e)) / mw::to_double(mw::mul_2exp(pd, -pd.e));
          const int de = delta.e - pd.e;
          for(int k = 0; k < (de < 0 ? -de : de); ++k)
            ratio *= de < 0 ? 0.5 : 2.0;
        }
      // the multiplicity seen from here
      int m = 1;
      if(have_pd)
        {
          const double est = (double)pmul / (1.0 - (ratio < 0.999 ? ratio : 0.999));
          m = est >= (double)n ? n : (int)(est + 0.5);
          m = m > mcap ? mcap : m;
          m = m < 1 ? 1 : m;
        }
      const bool slow = ratio > 0.25 || (clustered && !have_pd);
      if(ratio > 0.25)
        clustered = true;
      prev = x;
      const Mw<WL> step = m >= 2 ? mw::mul(delta, mw::from_u32<WL>((uint32_t)m)) : delta;
      const Mw<WL> xn = mw::add(x, step);
      const int ref = xn.e > emax ? xn.e : emax;
#ifdef SDPB_TRACE_TRIMIN
      printf("WL=%d it=%d x.e=%d delta.e=%d ref=%d ratio=%g m=%d mcap=%d slow=%d clustered=%d have_pd=%d\n", WL, it, x.e, delta.e, ref, ratio, m, mcap,
             (int)slow, (int)clustered, (int)have_pd);
#endif
      // quadratic convergence: a step below 2^-(16WL+8) of the scale leaves an error ~ step^2; while the steps still shrink
      // slowly only a step below the resolution of the mantissa ends the iteration
      if(mw::is_zero(step) || mw::is_zero(xn) || step.e < ref - (32 * WL - 6) || (!slow && step.e < ref - (16 * WL + 8)))
        {
          x = xn;
          return;
        }
      base = x;
      base_delta = delta;
      mul = m;
      accel_pending = m >= 2;
      x = xn;
      pd = delta;
      pmul = m;
      have_pd = true;
    }
}

// The Newton iteration on a ladder of widths: NL <- NL/2+1 <- ... down to about six limbs (192 bits: the fp64 bisection
// delivers ~50 bits, two steps fill the first rung; narrower rungs would add their own rounding of D and E, which is
// relative to the LARGEST entry, to a lambda_min that may be small against it).  Each rung ends below lambda_min of
// its own rounded matrix; the next one steps 256 of the previous rung's ulps down and needs one step.  Cost in
// limb^2 units at 576 bits: 2*36 + 100 + 324 instead of 3*100 + 324; at 832 bits 2*64 + 196 + 676 instead of
// 4*196 + 676.
template <int WL, int NL> struct TridiagLadder
{
  static constexpr int H = WL / 2 + 1;
  static constexpr int MINW = NL / 2 + 1 < 6 ? NL / 2 + 1 : 6;
  static __device__ void run(const Batch &D, const Batch &E, size_t od, size_t oe, int n, const Mw<NL> &start, Mw<WL> &x, double span, int emax,
                             bool &clustered)
  {
    if constexpr(H >= MINW && H < WL)
      {
        Mw<H> xs;
        TridiagLadder<H, NL>::run(D, E, od, oe, n, start, xs, span, emax, clustered);
        // step a few ulps of the narrower rung down so that this one starts from below
        Mw<H> ulps = mw::abs(xs);
        if(mw::is_zero(ulps) || ulps.e < emax)
          ulps = mw::mul_2exp(mw::from_u32<H>(1), emax); // the noise of a rung is relative to the entries of the matrix
        ulps.e -= 32 * H - 8;
        xs = mw::sub(xs, ulps);
        x = mw::widen<WL, H>(xs);
        tridiag_newton<WL, NL>(D, E, od, oe, n, x, span, emax, 32 * H - 16, clustered);
      }
    else
      {
        if constexpr(WL < NL)
          x = mw::narrow<WL, NL>(start);
        else
          x = start;
        tridiag_newton<WL, NL>(D, E, od, oe, n, x, span, emax, 24, clustered);
      }
  }
};

// Stage 2: lambda_min of each tridiagonal matrix, one lane per matrix.  fa/fb2 are
// fp64 work arrays laid out like D/E.  E is overwritten by its square.
template <int NL>
__global__ void __launch_bounds__(EIG_T) k_tridiag_min(Batch D, Batch E, double *fa, double *fb2, mw::Ptr lam)
{
  const int q = blockIdx.x * EIG_T + threadIdx.x;
  if(q >= D.count)
    return;
  const size_t od = (size_t)D.d[q].off, oe = (size_t)E.d[q].off;
  const int n = D.d[q].rows;
  if(n == 0)
    {
      Mw<NL> big = mw::from_u32<NL>(1);
      big.e = 1 << 28;
      mw::store<NL>(lam, q, big);
      return;
    }
  if(n == 1)
    {
      mw::store<NL>(lam, q, mw::load<NL>(D.p, od));
      return;
    }
  double *a = fa + od, *b2 = fb2 + od;
  // Shift by the first diagonal entry: lambda_min(T) = c + lambda_min(T - c I).  Near the end of a convergent run every
  // eigenvalue of L^-1 dX L^-T sits within 2^-40 of -(1 - beta) (step.cxx:202-206: both step lengths come out as
  // gamma / (1 - beta)): against the spectrum of T the Newton iteration on det(T - x) sees ONE root of multiplicity n and
  // converges linearly with ratio (n - 1) / n -- it ran into its iteration cap with the fp64 start (2^-48) still in place
  // (found by the strictly feasible fixture of round 6: tests/golden/synthetic/C4f_x0.25_to_termination.json, iteration 81).
  // T - c I has the same eigenvectors and its own scale: the cluster is resolved by the fp64 bisection and Newton is
  // quadratic again.  The subtraction is exact to 2^-(32 NL) of the entries, which is all lambda_min can be known to.
  const Mw<NL> shift = mw::load<NL>(D.p, od);
  for(int i = 0; i < n; ++i)
    mw::store<NL>(D.p, od + i, mw::sub(mw::load<NL>(D.p, od + i), shift));
  // common binary scale for the fp64 image; E := E^2
  int emax = mw::EZERO;
  for(int i = 0; i < n; ++i)
    {
      const int e1 = mw::load<NL>(D.p, od + i).e;
      emax = e1 > emax ? e1 : emax;
      if(i >= 1)
        {
          const int e2 = mw::load<NL>(E.p, oe + i).e;
          emax = e2 > emax ? e2 : emax;
        }
    }
  if(emax == mw::EZERO)
    {
      mw::store<NL>(lam, q, shift); // T = c I
      return;
    }
  for(int i = 0; i < n; ++i)
    {
      a[i] = mw::to_double(mw::mul_2exp(mw::load<NL>(D.p, od + i), -emax));
      double b = 0.0;
      if(i >= 1)
        {
          const Mw<NL> ei = mw::load<NL>(E.p, oe + i);
          b = mw::to_double(mw::mul_2exp(ei, -emax));
          mw::store<NL>(E.p, oe + i, mw::mul(ei, ei));
        }
      b2[i] = b * b;
    }
  // fp64: Gershgorin bounds + Sturm bisection for the smallest eigenvalue
  double lo = 1e300, hi = -1e300;
  for(int i = 0; i < n; ++i)
    {
      const double r = (i >= 1 ? mw::host_device_sqrt(b2[i]) : 0.0) + (i + 1 < n ? mw::host_device_sqrt(b2[i + 1]) : 0.0);
      lo = a[i] - r < lo ? a[i] - r : lo;
      hi = a[i] + r > hi ? a[i] + r : hi;
    }
  const double span = (hi - lo) > 0 ? (hi - lo) : 1.0;
  double blo = lo - 1e-9 * span, bhi = hi + 1e-9 * span;
  for(int it = 0; it < 80; ++it)
    {
      const double mid = 0.5 * (blo + bhi);
      if(mid == blo || mid == bhi)
        break; // the interval is one ulp wide: further steps would leave blo and bhi as they are (same bits, ~25 steps fewer)
      if(sturm_count_f64(a, b2, n, mid) >= 1)
        bhi = mid;
      else
        blo = mid;
    }
  // safely below lambda_min of the exact tridiagonal matrix
  Mw<NL> x = mw::mul_2exp(mw::from_double<NL>(blo - 1e-10 * span - 1e-300), emax);
  // multi-word Newton on p(x) = det(T - x): x += -1 / sum_i q_i'/q_i, on a ladder of mantissa widths (TridiagLadder):
  // quadratic convergence doubles the correct bits per step, so every step runs at the width it can fill
  const Mw<NL> start = x;
  bool clustered = false;
  TridiagLadder<NL, NL>::run(D, E, od, oe, n, start, x, span, emax, clustered);
  mw::store<NL>(lam, q, mw::add(x, shift));
}

// (max diag / min diag) per matrix: cholesky_condition_number.hxx:8-37 without the
// final square (done on the host for the winner only).
// one wavefront per matrix: lanes scan the diagonal with stride 64, then an LDS tree
constexpr int DR_T = 64;
template <int NL> __global__ void __launch_bounds__(DR_T) k_diag_ratio(Batch L, mw::Ptr out, size_t out_off)
{
  const int q = blockIdx.x, t = threadIdx.x;
  const MatDesc d = L.d[q];
  __shared__ Mw<NL> smx[DR_T], smn[DR_T];
  __shared__ int shas[DR_T];
  bool has = t < d.rows;
  Mw<NL> mx = mw::zero<NL>(), mn = mw::zero<NL>();
  if(has)
    {
      mx = mat_ld<NL>(L, d, t, t);
      mn = mx;
      for(int i = t + DR_T; i < d.rows; i += DR_T)
        {
          const Mw<NL> v = mat_ld<NL>(L, d, i, i);
          mx = mw::max(mx, v);
          mn = mw::min(mn, v);
        }
    }
  smx[t] = mx;
  smn[t] = mn;
  shas[t] = has ? 1 : 0;
  __syncthreads();
  for(int s = DR_T / 2; s > 0; s >>= 1)
    {
      if(t < s && shas[t + s])
        {
          smx[t] = shas[t] ? mw::max(smx[t], smx[t + s]) : smx[t + s];
          smn[t] = shas[t] ? mw::min(smn[t], smn[t + s]) : smn[t + s];
          shas[t] = 1;
        }
      __syncthreads();
    }
  if(t == 0)
    mw::store<NL>(out, out_off + q, d.rows > 0 ? mw::div(smx[0], smn[0]) : mw::zero<NL>());
}

// ---- scalars that stay on the device -------------------------------------------
// A host scalar becomes a kernel argument (no H2D copy, stream ordered).
template <int NL> __global__ void k_store_scalar(mw::Ptr p, size_t idx, Mw<NL> v)
{
  if(blockIdx.x == 0 && threadIdx.x == 0)
    mw::store<NL>(p, idx, v);
}

template <int UNUSED = 0> __global__ void k_store_words2(uint32_t *p, uint32_t w0, uint32_t w1)
{
  if(blockIdx.x == 0 && threadIdx.x == 0)
    {
      p[0] = w0;
      p[1] = w1;
    }
}
template <int UNUSED = 0> __global__ void k_store_words4(uint32_t *p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
  if(blockIdx.x == 0 && threadIdx.x == 0)
    {
      p[0] = w0;
      p[1] = w1;
      p[2] = w2;
      p[3] = w3;
    }
}

// Result block of one rank: `nslots` numbers (limb-major, stride nslots) followed by
// X_EXTRA raw words.  The host reads the whole block with ONE copy per synchronisation
// point; with several ranks the blocks are all-gathered and k_combine_slots applies one
// operation per slot, in rank order, so every rank ends with identical bits.
enum XOp : int
{
  X_KEEP = 0, // rank-local or replicated: untouched
  X_SUM,
  X_MAX,
  X_MIN,
  X_ARGMAX // value of the first rank holding the maximum; extra word XW_TAG follows the winner
};
constexpr int X_MAXSLOTS = 24;
constexpr int X_EXTRA = 7;
enum XWord : int
{
  XW_FAIL = 0, // smallest failure tag (0xffffffff = none): every rank raises the same error
  XW_STOP,     // rank 0's wall-clock / signal decision (compute_feasible_and_termination.cxx broadcasts rank 0's)
  XW_OR,       // OR of per-rank flags (SIGTERM received anywhere, run.cxx:332-336)
  XW_TAG,      // which matrix holds the largest Cholesky condition number
  // Collective-sequence self-check: every rank hashes the (kind, bytes, root) of every collective it has
  // enqueued since the communicator was attached (solver.hpp: note_collective) and deposits the 64-bit hash
  // here before each all-gather of the result block; k_combine_slots compares the hashes of all ranks and
  // writes 1 + the first rank that disagrees with rank 0 into XW_SEQBAD (0 = all agree), so that every rank
  // raises the same error at the same synchronisation point instead of computing on mismatched messages.
  XW_SEQ_LO,
  XW_SEQ_HI,
  XW_SEQBAD
};
struct XOps
{
  int op[X_MAXSLOTS];
};
template <int NL> __global__ void __launch_bounds__(64) k_combine_slots(const uint32_t *recv, int world, int nslots, XOps ops, uint32_t *local)
{
  const size_t blk = (size_t)(NL + 1) * nslots + X_EXTRA, xoff = (size_t)(NL + 1) * nslots;
  const int s = threadIdx.x;
  if(s < nslots && ops.op[s] != X_KEEP)
    {
      const int op = ops.op[s];
      Mw<NL> r = mw::load<NL>(mw::CPtr(recv, (size_t)nslots), s);
      int win = 0;
      for(int k = 1; k < world; ++k)
        {
          const Mw<NL> v = mw::load<NL>(mw::CPtr(recv + (size_t)k * blk, (size_t)nslots), s);
          if(op == X_SUM)
            r = mw::add(r, v);
          else if(op == X_MAX)
            r = mw::max(r, v);
          else if(op == X_MIN)
            r = mw::min(r, v);
          else if(mw::lt(r, v))
            {
              r = v;
              win = k;
            }
        }
      mw::store<NL>(mw::Ptr{local, (size_t)nslots}, s, r);
      if(op == X_ARGMAX)
        local[xoff + XW_TAG] = recv[(size_t)win * blk + xoff + XW_TAG];
    }
  if(s == nslots)
    {
      uint32_t f = 0xffffffffu, o = 0;
      for(int k = 0; k < world; ++k)
        {
          const uint32_t fk = recv[(size_t)k * blk + xoff + XW_FAIL];
          f = fk < f ? fk : f;
          o |= recv[(size_t)k * blk + xoff + XW_OR];
        }
      uint32_t bad = 0;
      for(int k = world - 1; k >= 1; --k)
        if(recv[(size_t)k * blk + xoff + XW_SEQ_LO] != recv[xoff + XW_SEQ_LO] || recv[(size_t)k * blk + xoff + XW_SEQ_HI] != recv[xoff + XW_SEQ_HI])
          bad = (uint32_t)k + 1u;
      local[xoff + XW_FAIL] = f;
      local[xoff + XW_STOP] = recv[xoff + XW_STOP];
      local[xoff + XW_OR] = o;
      local[xoff + XW_SEQBAD] = bad;
    }
}
// out[i] = sum over ranks (rank order) of the gathered n-vectors; out has stride n
template <int NL> __global__ void __launch_bounds__(WG) k_combine_vec(const uint32_t *recv, int world, int n, mw::Ptr out)
{
  const int i = blockIdx.x * WG + threadIdx.x;
  if(i >= n)
    return;
  const size_t blk = (size_t)(NL + 1) * n;
  Mw<NL> r = mw::load<NL>(mw::CPtr(recv, (size_t)n), i);
  for(int k = 1; k < world; ++k)
    r = mw::add(r, mw::load<NL>(mw::CPtr(recv + (size_t)k * blk, (size_t)n), i));
  mw::store<NL>(out, i, r);
}
// out = base (+/-) out, element-wise (gemv_t_all after the cross-rank sum)
template <int NL> __global__ void __launch_bounds__(WG) k_base_plus_signed(mw::Ptr out, mw::CPtr base, int has_base, int sign, int n)
{
  const int i = blockIdx.x * WG + threadIdx.x;
  if(i >= n)
    return;
  Mw<NL> s = mw::load<NL>(out, i);
  if(sign < 0)
    s = mw::neg(s);
  if(has_base)
    s = mw::add(mw::load<NL>(base, i), s);
  mw::store<NL>(out, i, s);
}

// Failure tags: stage << 27 | index, smallest wins (earliest stage, lowest block), so the error
// that is raised does not depend on which rank or lane saw it first.  Stages follow the order
// of the reference's throws: X, Y (cholesky_decomposition.cxx:22-25), S_j (compute_Q.cxx:36-38),
// the normalised-Q diagonal (compute_Q.cxx:65-91), Cholesky(Q) (initialize_schur_complement_solver.cxx:99).
enum FailStage : unsigned
{
  FAIL_X = 0,
  FAIL_Y,
  FAIL_S,
  FAIL_QDIAG,
  FAIL_Q
};
template <int UNUSED = 0> __global__ void __launch_bounds__(WG) k_fail_tags(const int *flags, int count, unsigned stage, int per_block, const BlockDesc *blk, uint32_t *tag)
{
  const int q = blockIdx.x * WG + threadIdx.x;
  if(q >= count || !flags[q])
    return;
  const unsigned code = (unsigned)blk[q / per_block].global_index * (unsigned)per_block + (unsigned)(q % per_block);
  atomicMin(tag, stage << 27 | code);
}
// qflags[0] = Cholesky(Q) failed, qflags[1] = 1 + first bad diagonal entry of the normalised Q
template <int UNUSED = 0> __global__ void k_fail_tags_q(const int *qflags, uint32_t *tag)
{
  if(blockIdx.x || threadIdx.x)
    return;
  if(qflags[1])
    atomicMin(tag, (unsigned)FAIL_QDIAG << 27 | (unsigned)(qflags[1] - 1));
  else if(qflags[0])
    atomicMin(tag, (unsigned)FAIL_Q << 27);
}

// Largest (max diag / min diag) over the Cholesky factors of S_j, X_jb, Y_jb in the reference's
// scan order (update_cond_numbers.hxx:16-110: block, then S, X_0, Y_0, X_1, Y_1; the first
// maximum wins).  ratio: [0,J1) S, [J1,3J1) X, [3J1,5J1) Y (k_diag_ratio).  Writes the value to
// res[slot] and global_block*8 + kind*2 + parity + 1 (kind 0 S, 1 X, 2 Y; 0 = none) to *tagword.
template <int NL>
__global__ void __launch_bounds__(WG) k_cond_best(mw::CPtr ratio, int Jl, const BlockDesc *blk, mw::Ptr res, size_t slot, uint32_t *tagword)
{
  const int J1 = Jl > 0 ? Jl : 1, t = threadIdx.x;
  constexpr int NONE = 0x7fffffff;
  Mw<NL> best = mw::zero<NL>();
  int bc = NONE;
  for(int c = t; c < 5 * Jl; c += WG)
    {
      const int l = c / 5, k = c % 5;
      const size_t pos = k == 0 ? (size_t)l : ((k & 1) ? (size_t)J1 + 2 * l + (k == 3) : (size_t)3 * J1 + 2 * l + (k == 4));
      const Mw<NL> v = mw::load<NL>(ratio, pos);
      if(!mw::is_zero(v) && (bc == NONE || mw::lt(best, v)))
        {
          best = v;
          bc = c;
        }
    }
  __shared__ Mw<NL> sv[WG];
  __shared__ int sc[WG];
  sv[t] = best;
  sc[t] = bc;
  __syncthreads();
  for(int s = WG / 2; s > 0; s >>= 1)
    {
      if(t < s && sc[t + s] != NONE)
        {
          const int c = mw::cmp(sv[t], sv[t + s]);
          if(sc[t] == NONE || c < 0 || (c == 0 && sc[t + s] < sc[t]))
            {
              sv[t] = sv[t + s];
              sc[t] = sc[t + s];
            }
        }
      __syncthreads();
    }
  if(t == 0)
    {
      mw::store<NL>(res, slot, sc[0] == NONE ? mw::zero<NL>() : sv[0]);
      uint32_t tag = 0;
      if(sc[0] != NONE)
        {
          const int l = sc[0] / 5, k = sc[0] % 5;
          const int kind = k == 0 ? 0 : ((k & 1) ? 1 : 2), parity = (k == 3 || k == 4) ? 1 : 0;
          tag = (uint32_t)blk[l].global_index * 8u + (uint32_t)kind * 2u + (uint32_t)parity + 1u;
        }
      *tagword = tag;
    }
}

// ---- multi-GPU exchange images ----------------------------------------------
// 32-bit two's-complement limbs widened to u64 lanes so that an integer SUM
// all-reduce (RCCL ncclSum on uint64) adds the partial Q' of every GPU exactly;
// k_narrow_carry propagates the deferred carries afterwards.  
template <int UNUSED = 0> __global__ void __launch_bounds__(WG) k_widen_u64(const uint32_t *in, size_t count, unsigned long long *out)
{
  for(size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < count; i += (size_t)gridDim.x * WG)
    out[i] = in[i];
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(WG)
  k_narrow_carry(const unsigned long long *in, size_t elems, int planes, uint32_t *out)
{
  for(size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < elems; i += (size_t)gridDim.x * WG)
    {
      unsigned long long carry = 0;
      for(int k = 0; k < planes; ++k)
        {
          const unsigned long long s = in[(size_t)k * elems + i] + carry;
          out[(size_t)k * elems + i] = (uint32_t)s;
          carry = s >> 32;
        }
    }
}
// The same pair for the Q' accumulators: only the lower triangle (i >= j) and the N column sums
// behind it carry information, so only those N(N+1)/2 + N entries per plane travel (half the
// message).  Packed entry of (i, j): j N - j (j - 1) / 2 + (i - j); column sum i: N(N+1)/2 + i.
// grid (cdiv(N, WG), N + 1): blockIdx.y = column j, j == N is the row of column sums.
// Columns [c0, c1) only (a chunk of the chased Q'): the message holds their packed entries, then -- with_sums -- the N
// column sums; grid (cdiv(N, WG), c1 - c0 + with_sums): blockIdx.y = c1 - c0 is the row of column sums.
MW_HD size_t tri_packed_offset(int N, int j) { return (size_t)j * N - (size_t)j * (size_t)(j > 0 ? j - 1 : 0) / 2; }
template <int UNUSED = 0>
__global__ void __launch_bounds__(WG)
  k_widen_tri_u64(const uint32_t *acc, size_t acc_stride, int N, int planes, unsigned long long *out, int c0, int c1, int with_sums)
{
  const int i = blockIdx.x * WG + threadIdx.x, j = c0 + (int)blockIdx.y;
  const bool sums = j >= c1;
  if(i >= N || (!sums && i < j))
    return;
  const size_t tri = tri_packed_offset(N, c1) - tri_packed_offset(N, c0), T = tri + (with_sums ? (size_t)N : 0);
  const size_t src = !sums ? (size_t)i + (size_t)j * N : (size_t)N * N + i;
  const size_t dst = !sums ? tri_packed_offset(N, j) - tri_packed_offset(N, c0) + (size_t)(i - j) : tri + i;
  for(int k = 0; k < planes; ++k)
    out[(size_t)k * T + dst] = acc[(size_t)k * acc_stride + src];
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(WG)
  k_narrow_tri_carry(const unsigned long long *in, int N, int planes, uint32_t *acc, size_t acc_stride, int c0, int c1, int with_sums)
{
  const int i = blockIdx.x * WG + threadIdx.x, j = c0 + (int)blockIdx.y;
  const bool sums = j >= c1;
  if(i >= N || (!sums && i < j))
    return;
  const size_t tri = tri_packed_offset(N, c1) - tri_packed_offset(N, c0), T = tri + (with_sums ? (size_t)N : 0);
  const size_t dst = !sums ? (size_t)i + (size_t)j * N : (size_t)N * N + i;
  const size_t src = !sums ? tri_packed_offset(N, j) - tri_packed_offset(N, c0) + (size_t)(i - j) : tri + i;
  unsigned long long carry = 0;
  for(int k = 0; k < planes; ++k)
    {
      const unsigned long long s = in[(size_t)k * T + src] + carry;
      acc[(size_t)k * acc_stride + dst] = (uint32_t)s;
      carry = s >> 32;
    }
}
} // namespace sdpb
