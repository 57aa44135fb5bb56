// ============================================================================
// oracle/sdpb_oracle.cpp — TEST INFRASTRUCTURE ONLY (the parity checker).
//
// A CPU restatement, on GMP `mpf` (the scalar type behind El::BigFloat), of the
// reference's interior-point iteration: the loop body of SDP_Solver::run()
// (src/sdp_solve/SDP_Solver/run/run.cxx:380-467) and SDP_Solver::step()
// (src/sdp_solve/SDP_Solver/run/step/step.cxx:51-229).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
// The product (sdpb_amd/, include/) never links, imports or calls it.
//
// The arithmetic the reference uses lives in third-party libraries that are not
// in /root/reference: the Elemental fork (El::Cholesky/Trsm/Gemm/Syrk/
// HermitianEig; Dockerfile:24-30, no version pin) over GMP mpf.  This file
// restates the *published* algorithms of those calls (unblocked right-looking
// Cholesky, forward/back substitution, triple-loop Gemm) on the same mpf scalar
// type, and is pinned against the reference's own golden traces
// (test/data/end-to-end_tests/*/output/out/iterations.json, out.txt) by
// tests/test_oracle_golden.py at the reference's own tolerance 2^-99
// (test/src/integration_tests/cases/end-to-end.test.cxx:27).
//
// Each function cites the reference file:line it follows.
// ============================================================================
#include <gmp.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
// Independent units (SDP blocks, matrix columns, output entries) are spread over the host
// cores with OpenMP.  Only loops whose iterations write disjoint outputs are parallel and
// every sum keeps its sequential order, so the results are bit-identical for any thread
// count (tests/test_oracle_golden.py runs the goldens at 1 and at all threads).
// ORACLE_THREADS / orc_set_threads() choose the count (default: all cores).
template <class Fn> void parallel_for(int count, Fn fn)
{
  std::string err;
  int err_at = -1;
#pragma omp parallel for schedule(dynamic, 1)
  for(int i = 0; i < count; ++i)
    {
      try
        {
          fn(i);
        }
      catch(std::exception &e)
        {
#pragma omp critical(oracle_error)
          if(err_at < 0 || i < err_at)
            {
              err_at = i;
              err = e.what();
            }
        }
    }
  if(err_at >= 0)
    throw std::runtime_error(err);
}

// ---------------------------------------------------------------------------
// Minimal RAII mpf wrapper (El::BigFloat == mpf at the default precision,
// src/sdpb_util/Environment.cxx:29-36 -> mpf_set_default_prec).
// ---------------------------------------------------------------------------
struct F
{
  mpf_t v;
  F() { mpf_init(v); }
  F(const F &o)
  {
    mpf_init2(v, mpf_get_prec(o.v));
    mpf_set(v, o.v);
  }
  explicit F(double d)
  {
    mpf_init(v);
    mpf_set_d(v, d);
  }
  explicit F(long i)
  {
    mpf_init(v);
    mpf_set_si(v, i);
  }
  F &operator=(const F &o)
  {
    if(this != &o)
      mpf_set(v, o.v);
    return *this;
  }
  ~F() { mpf_clear(v); }
  bool is_zero() const { return mpf_sgn(v) == 0; }
};

inline F operator+(const F &a, const F &b)
{
  F r;
  mpf_add(r.v, a.v, b.v);
  return r;
}
inline F operator-(const F &a, const F &b)
{
  F r;
  mpf_sub(r.v, a.v, b.v);
  return r;
}
inline F operator*(const F &a, const F &b)
{
  F r;
  mpf_mul(r.v, a.v, b.v);
  return r;
}
inline F operator/(const F &a, const F &b)
{
  F r;
  mpf_div(r.v, a.v, b.v);
  return r;
}
inline F operator-(const F &a)
{
  F r;
  mpf_neg(r.v, a.v);
  return r;
}
inline F &operator+=(F &a, const F &b)
{
  mpf_add(a.v, a.v, b.v);
  return a;
}
inline F &operator-=(F &a, const F &b)
{
  mpf_sub(a.v, a.v, b.v);
  return a;
}
inline F &operator*=(F &a, const F &b)
{
  mpf_mul(a.v, a.v, b.v);
  return a;
}
inline F &operator/=(F &a, const F &b)
{
  mpf_div(a.v, a.v, b.v);
  return a;
}
inline bool operator<(const F &a, const F &b) { return mpf_cmp(a.v, b.v) < 0; }
inline bool operator>(const F &a, const F &b) { return mpf_cmp(a.v, b.v) > 0; }
inline bool operator==(const F &a, const F &b)
{
  return mpf_cmp(a.v, b.v) == 0;
}
inline F fabs_(const F &a)
{
  F r;
  mpf_abs(r.v, a.v);
  return r;
}
inline F fsqrt(const F &a)
{
  F r;
  mpf_sqrt(r.v, a.v);
  return r;
}
inline F fmax_(const F &a, const F &b) { return a < b ? b : a; }
inline F fmin_(const F &a, const F &b) { return a < b ? a : b; }
// a += b*c with one temporary (same rounding as `a += b*c` on BigFloat)
inline void fma_(F &a, const F &b, const F &c, F &tmp)
{
  mpf_mul(tmp.v, b.v, c.v);
  mpf_add(a.v, a.v, tmp.v);
}
inline void fms_(F &a, const F &b, const F &c, F &tmp)
{
  mpf_mul(tmp.v, b.v, c.v);
  mpf_sub(a.v, a.v, tmp.v);
}

F from_str(const char *s, mp_bitcnt_t prec = 0)
{
  F r;
  if(prec)
    mpf_set_prec(r.v, prec);
  if(mpf_set_str(r.v, s, 10) != 0)
    throw std::runtime_error(std::string("oracle: bad number '") + s + "'");
  return r;
}

std::string to_str(const F &a)
{
  // enough digits to round-trip the full mantissa (prec+1 limbs)
  const size_t digits
    = (size_t)((mpf_get_prec(a.v) + 64) * 0.30103) + 3;
  mp_exp_t e;
  char *s = mpf_get_str(nullptr, &e, 10, digits, a.v);
  std::string m(s);
  free(s);
  if(m.empty())
    return "0";
  std::string out;
  size_t p = 0;
  if(m[0] == '-')
    {
      out = "-";
      p = 1;
    }
  out += "0." + m.substr(p) + "e" + std::to_string((long)e);
  return out;
}

// ---------------------------------------------------------------------------
// Dense column-major matrix of BigFloat (El::Matrix<El::BigFloat> layout).
// ---------------------------------------------------------------------------
struct Mat
{
  int h = 0, w = 0;
  std::vector<F> a;
  Mat() {}
  Mat(int h_, int w_) : h(h_), w(w_), a((size_t)h_ * w_) {}
  F &operator()(int i, int j) { return a[(size_t)j * h + i]; }
  const F &operator()(int i, int j) const { return a[(size_t)j * h + i]; }
  void zero()
  {
    for(auto &x : a)
      mpf_set_ui(x.v, 0);
  }
};

struct Z
{
  mpz_t v;
  Z() { mpz_init(v); }
  Z(const Z &o) { mpz_init_set(v, o.v); }
  Z &operator=(const Z &o)
  {
    mpz_set(v, o.v);
    return *this;
  }
  ~Z() { mpz_clear(v); }
};

struct NonPD : std::runtime_error
{
  NonPD() : std::runtime_error("A was not numerically HPD") {}
};

// El::Cholesky(LOWER, A): unblocked right-looking variant (all blocks on this
// path are smaller than Elemental's default blocksize, so the blocked driver
// reduces to it).  Called at cholesky_decomposition.cxx:17, compute_Q.cxx:31.
void cholesky_lower(Mat &A)
{
  const int n = A.h;
  F tmp;
  for(int j = 0; j < n; ++j)
    {
      if(mpf_sgn(A(j, j).v) <= 0)
        throw NonPD();
      A(j, j) = fsqrt(A(j, j));
      for(int i = j + 1; i < n; ++i)
        A(i, j) /= A(j, j);
      for(int c = j + 1; c < n; ++c)
        for(int r = c; r < n; ++r)
          fms_(A(r, c), A(r, j), A(c, j), tmp);
    }
  for(int j = 1; j < n; ++j)
    for(int i = 0; i < j; ++i)
      mpf_set_ui(A(i, j).v, 0);
}

// El::Cholesky(UPPER, Q): Q = U^T U, upper triangle referenced
// (initialize_schur_complement_solver.cxx:98).
void cholesky_upper(Mat &A)
{
  const int n = A.h;
  for(int j = 0; j < n; ++j)
    {
      if(mpf_sgn(A(j, j).v) <= 0)
        throw NonPD();
      A(j, j) = fsqrt(A(j, j));
      for(int c = j + 1; c < n; ++c)
        A(j, c) /= A(j, j);
#pragma omp parallel for schedule(static, 8) if(n - j > 64)
      for(int c = j + 1; c < n; ++c)
        {
          F t;
          for(int r = j + 1; r <= c; ++r)
            fms_(A(r, c), A(j, r), A(j, c), t);
        }
    }
  for(int j = 0; j < n; ++j)
    for(int i = j + 1; i < n; ++i)
      mpf_set_ui(A(i, j).v, 0);
}

// El::Trsm(LEFT, LOWER, NORMAL, NON_UNIT, 1, L, B): B := L^{-1} B
void trsm_lln(const Mat &L, Mat &B)
{
  F tmp;
  for(int c = 0; c < B.w; ++c)
    for(int i = 0; i < B.h; ++i)
      {
        for(int k = 0; k < i; ++k)
          fms_(B(i, c), L(i, k), B(k, c), tmp);
        B(i, c) /= L(i, i);
      }
}
// El::Trsm(LEFT, LOWER, TRANSPOSE, NON_UNIT, 1, L, B): B := L^{-T} B
void trsm_llt(const Mat &L, Mat &B)
{
  F tmp;
  for(int c = 0; c < B.w; ++c)
    for(int i = B.h - 1; i >= 0; --i)
      {
        for(int k = i + 1; k < B.h; ++k)
          fms_(B(i, c), L(k, i), B(k, c), tmp);
        B(i, c) /= L(i, i);
      }
}
// El::Trsm(RIGHT, LOWER, TRANSPOSE, NON_UNIT, 1, L, A): A := A L^{-T}
void trsm_rlt(const Mat &L, Mat &A)
{
  F tmp;
  for(int r = 0; r < A.h; ++r)
    for(int j = 0; j < A.w; ++j)
      {
        for(int k = 0; k < j; ++k)
          fms_(A(r, j), A(r, k), L(j, k), tmp);
        A(r, j) /= L(j, j);
      }
}
// U upper: solve U^T z = v then U w = z  (El::cholesky::SolveAfter(UPPER,...),
// solve_schur_complement_equation.cxx:64-65)
void solve_after_upper(const Mat &U, std::vector<F> &v)
{
  const int n = U.h;
  F tmp;
  for(int i = 0; i < n; ++i)
    {
      for(int k = 0; k < i; ++k)
        fms_(v[i], U(k, i), v[k], tmp);
      v[i] /= U(i, i);
    }
  for(int i = n - 1; i >= 0; --i)
    {
      for(int k = i + 1; k < n; ++k)
        fms_(v[i], U(i, k), v[k], tmp);
      v[i] /= U(i, i);
    }
}

// C := alpha*A*B + beta*C   (El::Gemm NORMAL,NORMAL; scale_multiply_add.cxx:4-13)
void gemm_nn(const F &alpha, const Mat &A, const Mat &B, const F &beta, Mat &C)
{
  F acc, tmp;
  const bool beta0 = beta.is_zero();
  for(int j = 0; j < C.w; ++j)
    for(int i = 0; i < C.h; ++i)
      {
        mpf_set_ui(acc.v, 0);
        for(int k = 0; k < A.w; ++k)
          fma_(acc, A(i, k), B(k, j), tmp);
        acc *= alpha;
        if(beta0)
          C(i, j) = acc;
        else
          {
            C(i, j) *= beta;
            C(i, j) += acc;
          }
      }
}

// Smallest eigenvalue of a symmetric matrix (min_eigenvalue.cxx:8-33 calls
// El::HermitianEig and takes El::Min; only the value is used).  Like Elemental this
// reduces to tridiagonal form with Householder reflections (4n^3/3 flops) and then
// finds the eigenvalues of the tridiagonal matrix — here with the implicit QL
// iteration (the published EISPACK tred1/tql1 algorithms; Elemental's
// divide-and-conquer yields the same eigenvalues to working precision).
F min_eigenvalue_sym(Mat A)
{
  const int n = A.h;
  F result;
  if(n == 0)
    return result; // caller skips empty blocks
  if(n == 1)
    return A(0, 0);
  const mp_bitcnt_t prec = mpf_get_prec(A(0, 0).v);
  std::vector<F> d(n), e(n);
  F f, g, h, hh, tmp, zero(0L), one(1L), two(2L);
  // Householder reduction, lower triangle referenced
  for(int i = n - 1; i >= 1; --i)
    {
      const int l = i - 1;
      mpf_set_ui(h.v, 0);
      if(l > 0)
        {
          for(int k = 0; k <= l; ++k)
            fma_(h, A(i, k), A(i, k), tmp);
          if(h.is_zero())
            e[i] = A(i, l);
          else
            {
              f = A(i, l);
              g = fsqrt(h);
              if(mpf_sgn(f.v) >= 0)
                g = -g;
              e[i] = g;
              h -= f * g;
              A(i, l) = f - g;
              mpf_set_ui(f.v, 0);
              for(int j = 0; j <= l; ++j)
                {
                  mpf_set_ui(g.v, 0);
                  for(int k = 0; k <= j; ++k)
                    fma_(g, A(j, k), A(i, k), tmp);
                  for(int k = j + 1; k <= l; ++k)
                    fma_(g, A(k, j), A(i, k), tmp);
                  e[j] = g / h;
                  fma_(f, e[j], A(i, j), tmp);
                }
              hh = f / (h + h);
              for(int j = 0; j <= l; ++j)
                {
                  f = A(i, j);
                  g = e[j] - hh * f;
                  e[j] = g;
                  for(int k = 0; k <= j; ++k)
                    {
                      fms_(A(j, k), f, e[k], tmp);
                      fms_(A(j, k), g, A(i, k), tmp);
                    }
                }
            }
        }
      else
        e[i] = A(i, l);
    }
  for(int i = 0; i < n; ++i)
    d[i] = A(i, i);
  // implicit QL on (d, e)
  for(int i = 1; i < n; ++i)
    e[i - 1] = e[i];
  mpf_set_ui(e[n - 1].v, 0);
  F eps(1L), dd, r, sn, c, p, bb;
  mpf_div_2exp(eps.v, eps.v, prec + 8);
  for(int l = 0; l < n; ++l)
    {
      int iter = 0, m;
      do
        {
          for(m = l; m < n - 1; ++m)
            {
              dd = fabs_(d[m]) + fabs_(d[m + 1]);
              tmp = eps * dd;
              if(!(fabs_(e[m]) > tmp))
                break;
            }
          if(m != l)
            {
              if(++iter > 100000)
                throw std::runtime_error("oracle: tridiagonal QL did not converge");
              g = (d[l + 1] - d[l]) / (two * e[l]);
              r = fsqrt(g * g + one);
              if(mpf_sgn(g.v) < 0)
                r = -r;
              g = d[m] - d[l] + e[l] / (g + r);
              sn = one;
              c = one;
              mpf_set_ui(p.v, 0);
              int i;
              bool underflow = false;
              for(i = m - 1; i >= l; --i)
                {
                  f = sn * e[i];
                  bb = c * e[i];
                  r = fsqrt(f * f + g * g);
                  e[i + 1] = r;
                  if(r.is_zero())
                    {
                      d[i + 1] -= p;
                      mpf_set_ui(e[m].v, 0);
                      underflow = true;
                      break;
                    }
                  sn = f / r;
                  c = g / r;
                  g = d[i + 1] - p;
                  r = (d[i] - g) * sn + two * c * bb;
                  p = sn * r;
                  d[i + 1] = g + p;
                  g = c * r - bb;
                }
              if(underflow)
                continue;
              d[l] -= p;
              e[l] = g;
              mpf_set_ui(e[m].v, 0);
            }
        }
      while(m != l);
    }
  result = d[0];
  for(int i = 1; i < n; ++i)
    if(d[i] < result)
      result = d[i];
  return result;
}

// ---------------------------------------------------------------------------
// Solver parameters (src/sdp_solve/Solver_Parameters/Solver_Parameters.cxx:
// 10-157 defaults).  The reference builds its defaults with
// El::BigFloat("0.3",10) *before* --precision is applied, i.e. at GMP's
// default 64-bit precision (3 limbs); the golden traces show it
// (1d/output/out/iterations.json iteration 1: beta = 0.2999...98725e-58).
// `param_prec_bits` reproduces that.
// ---------------------------------------------------------------------------
struct Params
{
  F duality_gap_threshold, primal_error_threshold, dual_error_threshold,
    initial_matrix_scale_primal, initial_matrix_scale_dual,
    feasible_centering_parameter, infeasible_centering_parameter,
    step_length_reduction, max_complementarity, min_primal_step,
    min_dual_step;
  long max_iterations = 500;
  bool find_primal_feasible = false, find_dual_feasible = false,
       detect_primal_feasible_jump = false, detect_dual_feasible_jump = false;
};

enum Terminate
{
  NotTerminated = -1,
  PrimalDualOptimal = 0,
  PrimalFeasible,
  DualFeasible,
  PrimalFeasibleJumpDetected,
  DualFeasibleJumpDetected,
  MaxIterationsExceeded,
  MaxRuntimeExceeded,
  MaxComplementarityExceeded,
  PrimalStepTooSmall,
  DualStepTooSmall
};
const char *terminate_names[] = {"found primal-dual optimal solution",
                                 "found primal feasible solution",
                                 "found dual feasible solution",
                                 "primal feasible jump detected",
                                 "dual feasible jump detected",
                                 "maxIterations exceeded",
                                 "maxRuntime exceeded",
                                 "maxComplementarity exceeded",
                                 "primal step too small",
                                 "dual step too small"};

// ---------------------------------------------------------------------------
struct Block
{
  int m = 0, K = 0, P = 0; // dim, num_points, schur block size (Block_Info.hxx:54-58)
  int rows[2] = {0, 0};    // bilinear basis heights (Block_Info.hxx:110-114)
  int n[2] = {0, 0};       // psd block sizes (Block_Info.hxx:86-96)
  Mat bases[2];            // rows[b] x K  (SDP.hxx:84)
  Mat bases_block[2];      // I_m (x) bases  (set_bases_blocks.cxx:3-22)
  Mat B;                   // P x N (free_var_matrix)
  std::vector<F> c;        // P (primal_objective_c)
  // solver state (SDP_Solver.hxx:28-74)
  std::vector<F> x, dual_residues, dx;
  Mat X[2], Y[2], primal_residues[2], Xc[2], Yc[2], dX[2], dY[2];
  Mat AXinv[2], AY[2]; // (mK)x(mK) full pairing matrices; tiles are views
  Mat S, L, Poff;      // schur complement, its Cholesky, schur_off_diagonal
  Mat minusXY[2];
};

struct Oracle
{
  int prec = 0, J = 0, N = 0;
  std::vector<Block> blk;
  std::vector<F> b, y, dy, primal_residue_p;
  F objective_const;
  Params par;
  Mat Q;
  double seconds_syrk_Q = 0; // wall time spent in syrk_Q so far (bench.py's cpu_baseline leg reports the stage)
  long total_psd_rows = 0;
  // per-iteration outputs (print_iteration.cxx:77-108)
  F primal_objective, dual_objective, duality_gap, primal_error_P,
    primal_error_p, dual_error, R_error, mu, beta_corrector,
    primal_step_length, dual_step_length, Q_cond_number,
    max_block_cond_number;
  std::string max_block_cond_number_name;
  long iteration = 0;
  int terminate_reason = NotTerminated;
  std::string error;
  std::string strbuf;
  F primal_error() const { return fmax_(primal_error_P, primal_error_p); }
};

// tile (cb, rb) of a pairing matrix as in compute_A_X_inv.cxx:39-56:
// A_X_inv[parity][j][column_block][row_block] = View(M, column_offset,
// row_offset) — i.e. rows start at column_block*K, cols at row_block*K.
inline const F &AX_tile(const Block &bl, int parity, int cb, int rb, int r, int c)
{
  return bl.AXinv[parity](cb * bl.K + r, rb * bl.K + c);
}
// compute_A_Y.cxx:47-64 stores the *transpose* of that view.
inline const F &AY_tile(const Block &bl, int parity, int cb, int rb, int r, int c)
{
  return bl.AY[parity](cb * bl.K + c, rb * bl.K + r);
}

// compute_objectives.cxx:6-29, dot.cxx:4-22
void compute_objectives(Oracle &o)
{
  F sum, tmp, bsum;
  for(auto &bl : o.blk)
    {
      F local; // Dotu per block, then accumulated (dot.cxx:13)
      for(int p = 0; p < bl.P; ++p)
        fma_(local, bl.c[p], bl.x[p], tmp);
      sum += local;
    }
  o.primal_objective = o.objective_const + sum;
  F d;
  for(int n = 0; n < o.N; ++n)
    fma_(d, o.b[n], o.y[n], tmp);
  o.dual_objective = o.objective_const + d;
  F denom = fmax_(fabs_(o.primal_objective) + fabs_(o.dual_objective), F(1L));
  o.duality_gap = fabs_(o.primal_objective - o.dual_objective) / denom;
}

// cholesky_decomposition.cxx:5-28
void cholesky_decomposition(Oracle &o, bool isX)
{
  parallel_for(2 * o.J, [&](int jb) {
    const int j = jb / 2, b = jb % 2;
      {
        Block &bl = o.blk[j];
        Mat &L = isX ? bl.Xc[b] : bl.Yc[b];
        L = isX ? bl.X[b] : bl.Y[b];
        try
          {
            cholesky_lower(L);
          }
        catch(NonPD &e)
          {
            std::ostringstream ss;
            ss << "Error when computing Cholesky decomposition of "
                  "Block_Diagonal_Matrix "
               << (isX ? "X" : "Y") << ", block index = " << j
               << ", parity = " << b << ": " << e.what();
            throw std::runtime_error(ss.str());
          }
      }
  });
}

// compute_A_X_inv.cxx:6-58 and compute_A_Y.cxx:16-66
void compute_bilinear_pairings(Oracle &o)
{
  parallel_for(2 * o.J, [&](int jb) {
    Block &bl = o.blk[jb / 2];
    const int b = jb % 2;
    F tmp;
      {
        const int q = bl.m * bl.K, n = bl.n[b];
        // A_X_inv = (L^{-1} E)^T (L^{-1} E), Syrk LOWER then MakeSymmetric
        Mat T(bl.bases_block[b]);
        trsm_lln(bl.Xc[b], T);
        Mat &AX = bl.AXinv[b];
        AX = Mat(q, q);
        for(int j = 0; j < q; ++j)
          for(int i = j; i < q; ++i)
            {
              F acc;
              for(int k = 0; k < n; ++k)
                fma_(acc, T(k, i), T(k, j), tmp);
              AX(i, j) = acc;
              AX(j, i) = acc;
            }
        // A_Y = E^T (Y E), then MakeSymmetric(LOWER)
        Mat YQ(n, q), &AY = bl.AY[b];
        gemm_nn(F(1L), bl.Y[b], bl.bases_block[b], F(0L), YQ);
        AY = Mat(q, q);
        for(int j = 0; j < q; ++j)
          for(int i = j; i < q; ++i)
            {
              F acc;
              for(int k = 0; k < n; ++k)
                fma_(acc, bl.bases_block[b](k, i), YQ(k, j), tmp);
              AY(i, j) = acc;
              AY(j, i) = acc;
            }
      }
  });
}

// compute_dual_residues_and_error.cxx:7-66
void compute_dual_residues_and_error(Oracle &o)
{
  std::vector<F> block_max(o.J);
  parallel_for(o.J, [&](int jj) {
    Block &bl = o.blk[jj];
    F tmp, &local_max = block_max[jj];
    {
      for(auto &v : bl.dual_residues)
        mpf_set_ui(v.v, 0);
      for(int b = 0; b < 2; ++b)
        for(int cb = 0; cb < bl.m; ++cb)
          for(int rb = 0; rb <= cb; ++rb)
            {
              const int off = (cb * (cb + 1) / 2 + rb) * bl.K;
              for(int k = 0; k < bl.K; ++k)
                bl.dual_residues[off + k] -= AY_tile(bl, b, cb, rb, k, k);
            }
      // dual_residues -= B y ; += c
      for(int p = 0; p < bl.P; ++p)
        {
          F acc;
          for(int n = 0; n < o.N; ++n)
            fma_(acc, bl.B(p, n), o.y[n], tmp);
          bl.dual_residues[p] -= acc;
          bl.dual_residues[p] += bl.c[p];
          local_max = fmax_(local_max, fabs_(bl.dual_residues[p]));
        }
    }
  });
  F local_max; // a maximum is exact: any order gives the same value
  for(auto &v : block_max)
    local_max = fmax_(local_max, v);
  o.dual_error = local_max;
}

// constraint_matrix_weighted_sum.cxx:14-66 : result = sum_p a[p] A_p
void constraint_matrix_weighted_sum(Oracle &o, bool use_dx, bool into_dX)
{
  F half;
  mpf_set_d(half.v, 0.5);
  parallel_for(o.J, [&](int jj) {
    Block &bl = o.blk[jj];
    F tmp, t2;
    {
      const std::vector<F> &a = use_dx ? bl.dx : bl.x;
      for(int b = 0; b < 2; ++b)
        {
          Mat &R = into_dX ? bl.dX[b] : bl.primal_residues[b];
          R.zero();
          const int rs = bl.rows[b];
          for(int cb = 0; cb < bl.m; ++cb)
            for(int rb = 0; rb <= cb; ++rb)
              {
                const int voff = (cb * (cb + 1) / 2 + rb) * bl.K;
                // scaled_bases = bases * diag(sub_vector); result_sub =
                // alpha * bases * scaled_bases^T at (row_offset, column_offset)
                for(int j = 0; j < rs; ++j)
                  for(int i = 0; i < rs; ++i)
                    {
                      F acc;
                      for(int k = 0; k < bl.K; ++k)
                        {
                          mpf_mul(tmp.v, bl.bases[b](j, k).v, a[voff + k].v);
                          fma_(acc, bl.bases[b](i, k), tmp, t2);
                        }
                      if(cb != rb)
                        acc *= half;
                      R(rb * rs + i, cb * rs + j) = acc;
                    }
              }
          if(bl.m > 1) // MakeSymmetric(UPPER)
            for(int j = 0; j < R.w; ++j)
              for(int i = j + 1; i < R.h; ++i)
                R(i, j) = R(j, i);
        }
    }
  });
}

// compute_primal_residues_and_error_P_Ax_X.cxx:5-14
void compute_primal_residues_P(Oracle &o)
{
  constraint_matrix_weighted_sum(o, false, false);
  F mx;
  for(auto &bl : o.blk)
    for(int b = 0; b < 2; ++b)
      for(size_t i = 0; i < bl.primal_residues[b].a.size(); ++i)
        {
          bl.primal_residues[b].a[i] -= bl.X[b].a[i];
          mx = fmax_(mx, fabs_(bl.primal_residues[b].a[i]));
        }
  o.primal_error_P = mx;
}

// compute_primal_residues_and_error_p_b_Bx.cxx:9-86 : p = b - B^T x
void compute_primal_residue_p(Oracle &o)
{
  std::vector<F> local(o.N);
  // every entry n visits the blocks in order, as the serial block-outer loop does
  parallel_for(o.N, [&](int n) {
    F tmp;
    for(size_t j = 0; j < o.blk.size(); ++j)
      {
        Block &bl = o.blk[j];
          {
            F acc; // Gemv(TRANSPOSE, -1, B, x, 0, block)
            for(int p = 0; p < bl.P; ++p)
              fma_(acc, bl.B(p, n), bl.x[p], tmp);
            acc = -acc;
            if(j == 0)
              acc += o.b[n];
            local[n] += acc;
          }
      }
  });
  F mx;
  for(int n = 0; n < o.N; ++n)
    {
      o.primal_residue_p[n] = local[n];
      mx = fmax_(mx, fabs_(local[n]));
    }
  o.primal_error_p = mx;
}

// compute_feasible_and_termination.cxx:4-71 (time-based reasons omitted: the
// oracle has no wall-clock limit)
bool compute_feasible_and_termination(Oracle &o, bool &feasible)
{
  const Params &p = o.par;
  const bool dualf = o.dual_error < p.dual_error_threshold,
             primf = o.primal_error() < p.primal_error_threshold;
  feasible = primf && dualf;
  const bool optimal = o.duality_gap < p.duality_gap_threshold;
  const F one(1L);
  if(feasible && optimal)
    o.terminate_reason = PrimalDualOptimal;
  else if(dualf && p.find_dual_feasible)
    o.terminate_reason = DualFeasible;
  else if(primf && p.find_primal_feasible)
    o.terminate_reason = PrimalFeasible;
  else if(o.dual_step_length == one && p.detect_dual_feasible_jump)
    o.terminate_reason = DualFeasibleJumpDetected;
  else if(o.primal_step_length == one && p.detect_primal_feasible_jump)
    o.terminate_reason = PrimalFeasibleJumpDetected;
  else if(o.iteration > p.max_iterations)
    o.terminate_reason = MaxIterationsExceeded;
  else if(o.iteration > 1 && o.primal_step_length < p.min_primal_step)
    o.terminate_reason = PrimalStepTooSmall;
  else if(o.iteration > 1 && o.dual_step_length < p.min_dual_step)
    o.terminate_reason = DualStepTooSmall;
  else
    return false;
  return true;
}

// compute_schur_complement.cxx:15-125
void compute_schur_complement(Oracle &o)
{
  const F four(4L);
  parallel_for(o.J, [&](int jj) {
    Block &bl = o.blk[jj];
    F element, product;
    {
      const int K = bl.K, m = bl.m;
      bl.S = Mat(bl.P, bl.P);
      for(int c0 = 0; c0 < m; ++c0)
        for(int r0 = 0; r0 <= c0; ++r0)
          {
            const int roff = (c0 * (c0 + 1) / 2 + r0) * K;
            for(int c1 = 0; c1 < m; ++c1)
              for(int r1 = 0; r1 <= c1; ++r1)
                {
                  const int coff = (c1 * (c1 + 1) / 2 + r1) * K;
                  for(int row = 0; row < K; ++row)
                    for(int col = 0; col < K; ++col)
                      {
                        mpf_set_ui(element.v, 0);
                        for(int b = 0; b < 2; ++b)
                          {
                            product = AX_tile(bl, b, c0, r1, row, col);
                            product *= AY_tile(bl, b, c1, r0, row, col);
                            element += product;
                            product = AX_tile(bl, b, r0, r1, row, col);
                            product *= AY_tile(bl, b, c1, c0, row, col);
                            element += product;
                            product = AX_tile(bl, b, c0, c1, row, col);
                            product *= AY_tile(bl, b, r1, r0, row, col);
                            element += product;
                            product = AX_tile(bl, b, r0, c1, row, col);
                            product *= AY_tile(bl, b, r1, c0, row, col);
                            element += product;
                          }
                        element /= four;
                        bl.S(roff + row, coff + col) = element;
                      }
                }
          }
      // MakeSymmetric(LOWER)
      for(int j = 0; j < bl.P; ++j)
        for(int i = 0; i < j; ++i)
          bl.S(i, j) = bl.S(j, i);
    }
  });
}

// compute_Q.cxx:9-61
void initialize_schur_off_diagonal(Oracle &o)
{
  parallel_for(o.J, [&](int j) {
    {
      Block &bl = o.blk[j];
      bl.L = bl.S;
      try
        {
          cholesky_lower(bl.L);
        }
      catch(NonPD &e)
        {
          std::ostringstream ss;
          ss << "Error when computing Cholesky decomposition of block_" << j
             << ": " << e.what();
          throw std::runtime_error(ss.str());
        }
      bl.Poff = bl.B;
      trsm_lln(bl.L, bl.Poff);
    }
  });
}

// compute_Q.cxx:94-132 with Matrix_Normalizer.cxx:75-137,174-192,210-264 and
// the *semantics* of bigint_syrk_blas (bigint_syrk_blas.cxx:183-302): the
// exact integer P'^T P' of the truncated, normalised, 2^p-shifted P'
// (fmpz_BigFloat_convert.hxx:13 -> fmpz_set_mpf truncates toward zero).
void syrk_Q_body(Oracle &o);
void syrk_Q(Oracle &o)
{
  const auto t0 = std::chrono::steady_clock::now();
  syrk_Q_body(o);
  o.seconds_syrk_Q += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
void syrk_Q_body(Oracle &o)
{
  const int N = o.N;
  // GMP reports the rounded-up precision (El::gmp::Precision(), compute_Q.cxx:107)
  const mp_bitcnt_t p = mpf_get_default_prec();
  std::vector<F> norms2(N), norms(N);
  parallel_for(N, [&](int n) {
    F tmp;
    for(auto &bl : o.blk)
      for(int r = 0; r < bl.P; ++r)
        fma_(norms2[n], bl.Poff(r, n), bl.Poff(r, n), tmp);
    norms[n] = fsqrt(norms2[n]);
  });

  // integer image of P'
  std::vector<std::vector<Z>> cols(N);
  size_t Ptot = 0;
  for(auto &bl : o.blk)
    Ptot += bl.P;
  parallel_for(N, [&](int n) {
    F tmp;
    {
      cols[n].resize(Ptot);
      size_t r0 = 0;
      for(auto &bl : o.blk)
        {
          for(int r = 0; r < bl.P; ++r)
            {
              if(norms[n].is_zero())
                {
                  // Matrix_Normalizer.cxx:183 skips zero-norm columns
                  mpz_set_f(cols[n][r0 + r].v, bl.Poff(r, n).v);
                  continue;
                }
              tmp = bl.Poff(r, n) / norms[n];
              mpf_mul_2exp(tmp.v, tmp.v, p);
              mpz_set_f(cols[n][r0 + r].v, tmp.v);
            }
          r0 += bl.P;
        }
    }
  });
  o.Q = Mat(N, N);
  F one(1L), eps(1L);
  mpf_div_2exp(eps.v, eps.v, p / 2);
  // columns from the last to the first: the long ones start first (dynamic schedule)
  parallel_for(N, [&](int jr) {
    const int j = N - 1 - jr;
    Z accz;
    mpz_ptr acc = accz.v;
    for(int i = 0; i <= j; ++i)
      {
        mpz_set_ui(acc, 0);
        for(size_t r = 0; r < Ptot; ++r)
          mpz_addmul(acc, cols[i][r].v, cols[j][r].v);
        F q;
        mpf_set_z(q.v, acc);
        mpf_div_2exp(q.v, q.v, 2 * p);
        if(i == j && !norms[i].is_zero())
          {
            // check_normalized_Q_diagonal, compute_Q.cxx:65-91
            F diff = fabs_(q - one);
            if(!(diff < eps))
              throw std::runtime_error(
                "Normalized Q should have ones on diagonal");
          }
        // restore_Q, Matrix_Normalizer.cxx:245-264
        q = q * norms[i] * norms[j];
        o.Q(i, j) = q;
      }
  });
  // restore_P (Matrix_Normalizer.cxx:210-227) returns P to (P'>>p)*norm; the
  // oracle keeps the un-normalised P, which differs from that by the
  // truncation of P' only (relative 2^-p) — inside every stated tolerance.
}

// update_cond_numbers.hxx:16-110, cholesky_condition_number.hxx:8-37
F cholesky_condition_number(const Mat &L)
{
  if(L.h == 0)
    return F(0L);
  F mx = L(0, 0), mn = L(0, 0);
  for(int i = 1; i < L.h; ++i)
    {
      mx = fmax_(mx, L(i, i));
      mn = fmin_(mn, L(i, i));
    }
  F r = mx / mn;
  return r * r;
}
void update_cond_numbers(Oracle &o)
{
  o.Q_cond_number = cholesky_condition_number(o.Q);
  mpf_set_ui(o.max_block_cond_number.v, 0);
  o.max_block_cond_number_name = "";
  for(int j = 0; j < o.J; ++j)
    {
      Block &bl = o.blk[j];
      F c = cholesky_condition_number(bl.L);
      if(o.max_block_cond_number < c)
        {
          o.max_block_cond_number = c;
          o.max_block_cond_number_name
            = "schur_complement_cholesky.block_" + std::to_string(j);
        }
      for(int b = 0; b < 2; ++b)
        {
          c = cholesky_condition_number(bl.Xc[b]);
          if(o.max_block_cond_number < c)
            {
              o.max_block_cond_number = c;
              o.max_block_cond_number_name = "X_cholesky.block_"
                                             + std::to_string(j) + "_"
                                             + std::to_string(b);
            }
          c = cholesky_condition_number(bl.Yc[b]);
          if(o.max_block_cond_number < c)
            {
              o.max_block_cond_number = c;
              o.max_block_cond_number_name = "Y_cholesky.block_"
                                             + std::to_string(j) + "_"
                                             + std::to_string(b);
            }
        }
    }
}

// cholesky_solve.cxx:4-13 : X := L^{-T} L^{-1} X
void cholesky_solve(const Mat &L, Mat &X)
{
  trsm_lln(L, X);
  trsm_llt(L, X);
}
// Block_Diagonal_Matrix.hxx:95-109
void symmetrize(Mat &A)
{
  F half;
  mpf_set_d(half.v, 0.5);
  for(auto &v : A.a)
    v *= half;
  for(int j = 0; j < A.w; ++j)
    for(int i = j; i < A.h; ++i)
      {
        F s = A(i, j) + A(j, i);
        A(i, j) = s;
        A(j, i) = s;
      }
}

// compute_schur_RHS.cxx:21-86
void compute_schur_RHS(Oracle &o, std::vector<Mat> &Z /* 2 per block */)
{
  parallel_for(o.J, [&](int j) {
    F tmp;
    {
      Block &bl = o.blk[j];
      for(int p = 0; p < bl.P; ++p)
        bl.dx[p] = -bl.dual_residues[p];
      for(int b = 0; b < 2; ++b)
        {
          const int rs = bl.rows[b];
          const Mat &Zb = Z[2 * j + b];
          for(int cb = 0; cb < bl.m; ++cb)
            for(int rb = 0; rb <= cb; ++rb)
              {
                const int off = (cb * (cb + 1) / 2 + rb) * bl.K;
                // Z_times_q = Z_sub * bases ; q_Z_q = Hadamard ; column sums
                for(int k = 0; k < bl.K; ++k)
                  {
                    F colsum;
                    for(int i = 0; i < rs; ++i)
                      {
                        F zq;
                        for(int l = 0; l < rs; ++l)
                          fma_(zq, Zb(rb * rs + i, cb * rs + l),
                               bl.bases[b](l, k), tmp);
                        fma_(colsum, zq, bl.bases[b](i, k), tmp);
                      }
                    bl.dx[off + k] -= colsum;
                  }
              }
        }
    }
  });
}

// solve_schur_complement_equation.cxx:16-79
void solve_schur_complement_equation(Oracle &o)
{
  std::vector<F> dy_sum(o.N);
  parallel_for(o.J, [&](int j) {
    Block &bl = o.blk[j];
    F tmp;
    // dx = L^{-1} dx
    for(int i = 0; i < bl.P; ++i)
      {
        for(int k = 0; k < i; ++k)
          fms_(bl.dx[i], bl.L(i, k), bl.dx[k], tmp);
        bl.dx[i] /= bl.L(i, i);
      }
  });
  // every entry n visits the blocks in order, as the serial block-outer loop does
  parallel_for(o.N, [&](int n) {
    F tmp;
    for(size_t j = 0; j < o.blk.size(); ++j)
      {
        Block &bl = o.blk[j];
      // dy_block = dy - P^T dx ; summed over blocks.  In the reference every
      // block holds a copy of dy (= primal_residue_p, which is non-zero in
      // block 0 plus the B^T x pieces — already summed here), so the sum of
      // the per-block copies is sum_j(-P_j^T dx_j) + sum_j dy_j.
          {
            F acc;
            for(int p = 0; p < bl.P; ++p)
              fma_(acc, bl.Poff(p, n), bl.dx[p], tmp);
            dy_sum[n] -= acc;
          }
      }
  });
  for(int n = 0; n < o.N; ++n)
    o.dy[n] = o.dy[n] + dy_sum[n];
  solve_after_upper(o.Q, o.dy);
  parallel_for(o.J, [&](int jj) {
    Block &bl = o.blk[jj];
    F tmp;
    {
      for(int p = 0; p < bl.P; ++p)
        {
          F acc;
          for(int n = 0; n < o.N; ++n)
            fma_(acc, bl.Poff(p, n), o.dy[n], tmp);
          bl.dx[p] += acc;
        }
      // dx = L^{-T} dx
      for(int i = bl.P - 1; i >= 0; --i)
        {
          for(int k = i + 1; k < bl.P; ++k)
            fms_(bl.dx[i], bl.L(k, i), bl.dx[k], tmp);
          bl.dx[i] /= bl.L(i, i);
        }
    }
  });
}

// compute_search_direction.cxx:44-90
void compute_search_direction(Oracle &o, const F &beta, bool corrector)
{
  const F one(1L), zero(0L), minus_one(-1L);
  std::vector<Mat> R(2 * o.J), Z(2 * o.J);
  F bm = beta * o.mu;
  parallel_for(2 * o.J, [&](int jb) {
    const int j = jb / 2, b = jb % 2;
      {
        Block &bl = o.blk[j];
        Mat &Rb = R[2 * j + b];
        Rb = bl.minusXY[b];
        if(corrector)
          gemm_nn(minus_one, bl.dX[b], bl.dY[b], one, Rb);
        for(int i = 0; i < Rb.h; ++i)
          Rb(i, i) += bm;
        Mat &Zb = Z[2 * j + b];
        Zb = Mat(bl.n[b], bl.n[b]);
        gemm_nn(one, bl.primal_residues[b], bl.Y[b], zero, Zb);
        for(size_t i = 0; i < Zb.a.size(); ++i)
          Zb.a[i] -= Rb.a[i];
        cholesky_solve(bl.Xc[b], Zb);
        symmetrize(Zb);
      }
  });
  compute_schur_RHS(o, Z);
  // In the reference each block's dy starts as that block's primal_residue_p
  // (compute_search_direction.cxx:74); their sum over blocks is the global
  // residue p = b - B^T x held here.
  for(int n = 0; n < o.N; ++n)
    o.dy[n] = o.primal_residue_p[n];
  solve_schur_complement_equation(o);
  constraint_matrix_weighted_sum(o, true, true);
  parallel_for(2 * o.J, [&](int jb) {
    const int j = jb / 2, b = jb % 2;
      {
        Block &bl = o.blk[j];
        for(size_t i = 0; i < bl.dX[b].a.size(); ++i)
          bl.dX[b].a[i] += bl.primal_residues[b].a[i];
        Mat &dYb = bl.dY[b];
        gemm_nn(one, bl.dX[b], bl.Y[b], zero, dYb);
        for(size_t i = 0; i < dYb.a.size(); ++i)
          dYb.a[i] -= R[2 * j + b].a[i];
        cholesky_solve(bl.Xc[b], dYb);
        symmetrize(dYb);
        for(auto &v : dYb.a)
          v = -v;
      }
  });
}

// corrector_centering_parameter.cxx:12-31, frobenius_product_of_sums.cxx:6-31
F corrector_centering_parameter(Oracle &o, bool feasible)
{
  std::vector<F> locals(2 * o.J);
  parallel_for(2 * o.J, [&](int jb) {
    Block &bl = o.blk[jb / 2];
    const int b = jb % 2;
    F tmp, &local = locals[jb];
    for(size_t i = 0; i < bl.X[b].a.size(); ++i)
      {
        F xs = bl.X[b].a[i] + bl.dX[b].a[i];
        F ys = bl.Y[b].a[i] + bl.dY[b].a[i];
        fma_(local, xs, ys, tmp);
      }
  });
  F sum;
  for(auto &local : locals) // block order, as in the serial loop
    sum += local;
  F r = sum / (o.mu * F(o.total_psd_rows));
  F beta = (r < F(1L)) ? r * r : r;
  if(feasible)
    return fmin_(fmax_(o.par.feasible_centering_parameter, beta), F(1L));
  return fmax_(o.par.infeasible_centering_parameter, beta);
}

// step_length.cxx:27-46, lower_triangular_inverse_congruence.cxx:4-16,
// min_eigenvalue.cxx:8-33
F step_length(Oracle &o, bool primal)
{
  bool have = false;
  F lambda;
  std::vector<F> evs(2 * o.J);
  parallel_for(2 * o.J, [&](int jb) {
    Block &bl = o.blk[jb / 2];
    const int b = jb % 2;
    if(bl.n[b] == 0)
      return;
    Mat A(primal ? bl.dX[b] : bl.dY[b]);
    const Mat &L = primal ? bl.Xc[b] : bl.Yc[b];
    trsm_rlt(L, A);
    trsm_lln(L, A);
    evs[jb] = min_eigenvalue_sym(A);
  });
  for(int jb = 0; jb < 2 * o.J; ++jb)
    {
      if(o.blk[jb / 2].n[jb % 2] == 0)
        continue;
      const F &ev = evs[jb];
      if(!have || ev < lambda)
        {
          lambda = ev;
          have = true;
        }
    }
  const F &gamma = o.par.step_length_reduction;
  if(lambda > -gamma)
    return F(1L);
  return -gamma / lambda;
}

// step.cxx:51-229 ; returns true if terminate_now (mu > maxComplementarity)
bool step(Oracle &o, bool feasible)
{
  compute_schur_complement(o);
  initialize_schur_off_diagonal(o);
  syrk_Q(o);
  try
    {
      cholesky_upper(o.Q);
    }
  catch(NonPD &e)
    {
      throw std::runtime_error(std::string("Error when computing Cholesky(Q): ")
                               + e.what());
    }
  const F minus_one(-1L), zero(0L);
  F trace;
  std::vector<F> traces(2 * o.J);
  parallel_for(2 * o.J, [&](int jb) {
    Block &bl = o.blk[jb / 2];
    const int b = jb % 2;
    bl.minusXY[b] = Mat(bl.n[b], bl.n[b]);
    gemm_nn(minus_one, bl.X[b], bl.Y[b], zero, bl.minusXY[b]);
    F &t = traces[jb]; // El::Trace per block then accumulated
    for(int i = 0; i < bl.n[b]; ++i)
      t += bl.minusXY[b](i, i);
  });
  for(auto &t : traces)
    trace += t;
  o.mu = -trace / F(o.total_psd_rows);
  if(o.mu > o.par.max_complementarity)
    return true;
  // compute_R_error.hxx:9-29
  F rerr;
  for(auto &bl : o.blk)
    for(int b = 0; b < 2; ++b)
      for(int j = 0; j < bl.n[b]; ++j)
        for(int i = 0; i < bl.n[b]; ++i)
          {
            F v = bl.minusXY[b](i, j);
            if(i == j)
              v += o.mu;
            rerr = fmax_(rerr, fabs_(v));
          }
  o.R_error = rerr;

  // predictor_centering_parameter.cxx:4-9
  F beta_predictor
    = feasible ? F(0L) : o.par.infeasible_centering_parameter;
  compute_search_direction(o, beta_predictor, false);
  o.beta_corrector = corrector_centering_parameter(o, feasible);
  compute_search_direction(o, o.beta_corrector, true);
  update_cond_numbers(o);

  o.primal_step_length = step_length(o, true);
  o.dual_step_length = step_length(o, false);
  if(feasible)
    {
      o.primal_step_length = fmin_(o.primal_step_length, o.dual_step_length);
      o.dual_step_length = o.primal_step_length;
    }
  F tmp;
  for(auto &bl : o.blk)
    {
      for(int p = 0; p < bl.P; ++p)
        fma_(bl.x[p], o.primal_step_length, bl.dx[p], tmp);
      for(int b = 0; b < 2; ++b)
        for(size_t i = 0; i < bl.X[b].a.size(); ++i)
          {
            // dX *= step ; X += dX (step.cxx:214-216)
            bl.dX[b].a[i] *= o.primal_step_length;
            bl.X[b].a[i] += bl.dX[b].a[i];
            bl.dY[b].a[i] *= o.dual_step_length;
            bl.Y[b].a[i] += bl.dY[b].a[i];
          }
    }
  for(int n = 0; n < o.N; ++n)
    fma_(o.y[n], o.dual_step_length, o.dy[n], tmp);
  return false;
}

void parse_list(const char *txt, std::vector<F> &out, size_t expect,
                const char *what)
{
  out.clear();
  out.reserve(expect);
  const char *p = txt;
  std::string tok;
  while(*p)
    {
      while(*p == ' ' || *p == '\n' || *p == '\t' || *p == ',')
        ++p;
      if(!*p)
        break;
      const char *q = p;
      while(*q && *q != ' ' && *q != '\n' && *q != '\t' && *q != ',')
        ++q;
      tok.assign(p, q - p);
      out.push_back(from_str(tok.c_str()));
      p = q;
    }
  if(out.size() != expect)
    throw std::runtime_error(std::string("oracle: wrong element count for ")
                             + what + ": got " + std::to_string(out.size())
                             + " expected " + std::to_string(expect));
}
} // namespace

// ===========================================================================
// C API (ctypes).  Numbers cross as decimal strings.
// ===========================================================================
#define ORC_TRY(o) try {
#define ORC_CATCH(o)                                                          \
  }                                                                           \
  catch(std::exception & e)                                                   \
  {                                                                           \
    (o)->error = e.what();                                                    \
    return 1;                                                                 \
  }                                                                           \
  return 0;

extern "C" {

// number of host threads the block/column loops use (0 = all cores); returns the count in effect
int orc_set_threads(int n)
{
#ifdef _OPENMP
  if(n > 0)
    omp_set_num_threads(n);
  else if(const char *e = getenv("ORACLE_THREADS"))
    {
      if(atoi(e) > 0)
        omp_set_num_threads(atoi(e));
    }
  else
    omp_set_num_threads(omp_get_num_procs());
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

// GMP's default precision is process-global and mpf_init reads it (as does syrk_Q's truncation point,
// compute_Q.cxx:107 = mpf_get_default_prec).  Every entry point therefore re-establishes the precision of ITS
// oracle first: oracles of different precisions may be alive at once without contaminating each other
// (round-5 review: a 512-bit oracle created before a 1536-bit one silently computed at 1536 bits).
static inline Oracle *enter(void *h)
{
  Oracle *o = static_cast<Oracle *>(h);
  mpf_set_default_prec(o->prec);
  return o;
}

void *orc_create(int precision_bits, int J, const int *dims,
                 const int *num_points, int N)
{
  mpf_set_default_prec(precision_bits);
  Oracle *o = new Oracle;
  o->prec = precision_bits;
  o->J = J;
  o->N = N;
  o->blk.resize(J);
  o->b.resize(N);
  o->y.resize(N);
  o->dy.resize(N);
  o->primal_residue_p.resize(N);
  for(int j = 0; j < J; ++j)
    {
      Block &bl = o->blk[j];
      bl.m = dims[j];
      bl.K = num_points[j];
      bl.P = bl.K * bl.m * (bl.m + 1) / 2;
      const int d = bl.K - 1;
      bl.rows[0] = d / 2 + 1;
      bl.rows[1] = (d + 1) / 2;
      bl.n[0] = bl.m * ((bl.K + 1) / 2);
      bl.n[1] = bl.m * bl.K - bl.n[0];
      o->total_psd_rows += bl.n[0] + bl.n[1];
      bl.x.resize(bl.P);
      bl.dx.resize(bl.P);
      bl.dual_residues.resize(bl.P);
      bl.c.resize(bl.P);
      for(int b = 0; b < 2; ++b)
        {
          bl.X[b] = Mat(bl.n[b], bl.n[b]);
          bl.Y[b] = Mat(bl.n[b], bl.n[b]);
          bl.dX[b] = Mat(bl.n[b], bl.n[b]);
          bl.dY[b] = Mat(bl.n[b], bl.n[b]);
          bl.primal_residues[b] = Mat(bl.n[b], bl.n[b]);
        }
    }
  // Solver_Parameters.cxx defaults, parsed at GMP's initial 64-bit default
  // precision like the reference (see struct Params).
  Params &p = o->par;
  const mp_bitcnt_t pp = 64;
  p.duality_gap_threshold = from_str("1e-30", pp);
  p.primal_error_threshold = from_str("1e-30", pp);
  p.dual_error_threshold = from_str("1e-30", pp);
  p.initial_matrix_scale_primal = from_str("1e20", pp);
  p.initial_matrix_scale_dual = from_str("1e20", pp);
  p.feasible_centering_parameter = from_str("0.1", pp);
  p.infeasible_centering_parameter = from_str("0.3", pp);
  p.step_length_reduction = from_str("0.7", pp);
  p.max_complementarity = from_str("1e100", pp);
  p.min_primal_step = from_str("0", pp);
  p.min_dual_step = from_str("0", pp);
  return o;
}

void orc_destroy(void *h) { delete static_cast<Oracle *>(h); }

const char *orc_last_error(void *h)
{
  return static_cast<Oracle *>(h)->error.c_str();
}

// name in {dualityGapThreshold, primalErrorThreshold, dualErrorThreshold,
// initialMatrixScalePrimal, initialMatrixScaleDual, feasibleCenteringParameter,
// infeasibleCenteringParameter, stepLengthReduction, maxComplementarity,
// minPrimalStep, minDualStep}; prec_bits=0 -> working precision.
int orc_set_param(void *h, const char *name, const char *value, int prec_bits)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  Params &p = o->par;
  const std::string n(name);
  // the parameter keeps the working precision as its storage precision, but
  // its *value* is the one obtained by parsing at prec_bits
  F parsed = from_str(value, prec_bits ? prec_bits : 0);
  F v;
  mpf_set(v.v, parsed.v);
  if(n == "dualityGapThreshold") p.duality_gap_threshold = v;
  else if(n == "primalErrorThreshold") p.primal_error_threshold = v;
  else if(n == "dualErrorThreshold") p.dual_error_threshold = v;
  else if(n == "initialMatrixScalePrimal") p.initial_matrix_scale_primal = v;
  else if(n == "initialMatrixScaleDual") p.initial_matrix_scale_dual = v;
  else if(n == "feasibleCenteringParameter") p.feasible_centering_parameter = v;
  else if(n == "infeasibleCenteringParameter") p.infeasible_centering_parameter = v;
  else if(n == "stepLengthReduction") p.step_length_reduction = v;
  else if(n == "maxComplementarity") p.max_complementarity = v;
  else if(n == "minPrimalStep") p.min_primal_step = v;
  else if(n == "minDualStep") p.min_dual_step = v;
  else throw std::runtime_error("oracle: unknown parameter " + n);
  ORC_CATCH(o)
}

int orc_set_flags(void *h, long max_iterations, int find_primal_feasible,
                  int find_dual_feasible, int detect_primal_feasible_jump,
                  int detect_dual_feasible_jump)
{
  Oracle *o = enter(h);
  o->par.max_iterations = max_iterations;
  o->par.find_primal_feasible = find_primal_feasible;
  o->par.find_dual_feasible = find_dual_feasible;
  o->par.detect_primal_feasible_jump = detect_primal_feasible_jump;
  o->par.detect_dual_feasible_jump = detect_dual_feasible_jump;
  return 0;
}

// Text blobs are whitespace-separated decimals in the JSON's row-major order
// (Json_Block_Data_Parser.hxx:26-36): bases_*[row][k], B[p][n], c[p].
int orc_set_block(void *h, int j, const char *bases_even,
                  const char *bases_odd, const char *B, const char *c)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  Block &bl = o->blk.at(j);
  std::vector<F> tmp;
  const char *src[2] = {bases_even, bases_odd};
  for(int b = 0; b < 2; ++b)
    {
      parse_list(src[b], tmp, (size_t)bl.rows[b] * bl.K, "bilinear_bases");
      bl.bases[b] = Mat(bl.rows[b], bl.K);
      for(int r = 0; r < bl.rows[b]; ++r)
        for(int k = 0; k < bl.K; ++k)
          bl.bases[b](r, k) = tmp[(size_t)r * bl.K + k];
      // set_bases_blocks.cxx:3-22
      Mat &E = bl.bases_block[b];
      E = Mat(bl.n[b], bl.m * bl.K);
      for(int row = 0; row < E.h; ++row)
        for(int col = 0; col < E.w; ++col)
          if(row / bl.rows[b] == col / bl.K)
            E(row, col) = bl.bases[b](row % bl.rows[b], col % bl.K);
    }
  parse_list(B, tmp, (size_t)bl.P * o->N, "B");
  bl.B = Mat(bl.P, o->N);
  for(int p = 0; p < bl.P; ++p)
    for(int n = 0; n < o->N; ++n)
      bl.B(p, n) = tmp[(size_t)p * o->N + n];
  parse_list(c, bl.c, bl.P, "c");
  ORC_CATCH(o)
}

// Same block with B (row-major P x N) and c given as doubles: exact for the dyadic
// rationals of the synthetic generator (mpf_set_d is exact), no decimal round trip.
int orc_set_block_f64(void *h, int j, const char *bases_even,
                      const char *bases_odd, const double *B, const double *c)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  Block &bl = o->blk.at(j);
  // bases through the text path (shared code), B and c below
  {
    std::vector<F> tmp;
    const char *src[2] = {bases_even, bases_odd};
    for(int b = 0; b < 2; ++b)
      {
        parse_list(src[b], tmp, (size_t)bl.rows[b] * bl.K, "bilinear_bases");
        bl.bases[b] = Mat(bl.rows[b], bl.K);
        for(int r = 0; r < bl.rows[b]; ++r)
          for(int k = 0; k < bl.K; ++k)
            bl.bases[b](r, k) = tmp[(size_t)r * bl.K + k];
        Mat &E = bl.bases_block[b];
        E = Mat(bl.n[b], bl.m * bl.K);
        for(int row = 0; row < E.h; ++row)
          for(int col = 0; col < E.w; ++col)
            if(row / bl.rows[b] == col / bl.K)
              E(row, col) = bl.bases[b](row % bl.rows[b], col % bl.K);
      }
  }
  bl.B = Mat(bl.P, o->N);
  for(int p = 0; p < bl.P; ++p)
    for(int n = 0; n < o->N; ++n)
      mpf_set_d(bl.B(p, n).v, B[(size_t)p * o->N + n]);
  bl.c.resize(bl.P);
  for(int p = 0; p < bl.P; ++p)
    mpf_set_d(bl.c[p].v, c[p]);
  ORC_CATCH(o)
}

// c of block j as decimals (after orc_set_block_f64: the feasible synthetic family has dyadic B but a c that
// is no double, sdpb_amd/synthetic.py)
int orc_set_block_c(void *h, int j, const char *c)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  Block &bl = o->blk.at(j);
  parse_list(c, bl.c, bl.P, "c");
  ORC_CATCH(o)
}

int orc_set_objective(void *h, const char *b, const char *constant)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  parse_list(b, o->b, o->N, "b");
  o->objective_const = from_str(constant);
  ORC_CATCH(o)
}

// SDP_Solver.cxx:23-38 : x=0, y=0, X = Omega_p I, Y = Omega_d I
int orc_init_state(void *h)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  for(auto &bl : o->blk)
    {
      for(auto &v : bl.x)
        mpf_set_ui(v.v, 0);
      for(int b = 0; b < 2; ++b)
        {
          bl.X[b].zero();
          bl.Y[b].zero();
          for(int i = 0; i < bl.n[b]; ++i)
            {
              bl.X[b](i, i) = o->par.initial_matrix_scale_primal;
              bl.Y[b](i, i) = o->par.initial_matrix_scale_dual;
            }
        }
    }
  for(auto &v : o->y)
    mpf_set_ui(v.v, 0);
  o->iteration = 0;
  o->terminate_reason = NotTerminated;
  mpf_set_ui(o->primal_step_length.v, 0);
  mpf_set_ui(o->dual_step_length.v, 0);
  ORC_CATCH(o)
}

// One pass of run.cxx:322-467.  *terminated = 1 when the loop would `break`
// (reason in orc_terminate_reason); the iteration's scalars are then the ones
// save_solution would print.  Returns nonzero on error (orc_last_error).
int orc_iterate(void *h, int *terminated)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  *terminated = 0;
  o->iteration += 1;
  compute_objectives(*o);
  cholesky_decomposition(*o, true);
  cholesky_decomposition(*o, false);
  compute_bilinear_pairings(*o);
  compute_dual_residues_and_error(*o);
  compute_primal_residues_P(*o);
  compute_primal_residue_p(*o);
  bool feasible;
  if(compute_feasible_and_termination(*o, feasible))
    {
      *terminated = 1;
      return 0;
    }
  if(step(*o, feasible))
    {
      o->terminate_reason = MaxComplementarityExceeded;
      *terminated = 1;
      return 0;
    }
  ORC_CATCH(o)
}

// ---- the Schur-solver hook (what approx_objective / outer_limits reuse) ----------
// approx_objective/setup_solver.cxx:204-220 and outer_limits/compute_optimal/
// compute_optimal.cxx:188-215 call, on a loaded solution (X, Y):
//   cholesky_decomposition(X), cholesky_decomposition(Y), compute_bilinear_pairings,
//   initialize_schur_complement_solver (compute_schur_complement, L_j = chol(S_j),
//   P_j = L_j^{-1} B_j, Q = sum P_j^T P_j, Cholesky(UPPER, Q))
// and then solve_schur_complement_equation with their own right-hand sides.  The product
// exposes that as sdpb_hip_schur_solver_init / sdpb_hip_schur_solve; this is the checker.
int orc_schur_solver_init(void *h)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  cholesky_decomposition(*o, true);
  cholesky_decomposition(*o, false);
  compute_bilinear_pairings(*o);
  compute_schur_complement(*o);
  initialize_schur_off_diagonal(*o);
  syrk_Q(*o);
  try
    {
      cholesky_upper(o->Q);
    }
  catch(NonPD &e)
    {
      throw std::runtime_error(std::string("Error when computing Cholesky(Q): ") + e.what());
    }
  ORC_CATCH(o)
}
// solve_schur_complement_equation.cxx:16-79 with the solver above: in dx (per block) and dy
// (set with orc_set_array), out the solution in the same arrays
int orc_schur_solve(void *h)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  solve_schur_complement_equation(*o);
  ORC_CATCH(o)
}
// which in {"x","X","y","Y","dx","dy"}: column-major decimals (state injection / right-hand sides)
int orc_set_array(void *h, const char *which, int j, int parity, const char *txt)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  const std::string w(which);
  std::vector<F> v;
  auto into = [&](std::vector<F> &dst) {
    parse_list(txt, v, dst.size(), which);
    for(size_t i = 0; i < dst.size(); ++i)
      dst[i] = v[i];
  };
  if(w == "y") into(o->y);
  else if(w == "dy")
    {
      o->dy.resize(o->N);
      into(o->dy);
    }
  else
    {
      Block &bl = o->blk.at(j);
      if(w == "x") into(bl.x);
      else if(w == "dx")
        {
          bl.dx.resize(bl.P);
          into(bl.dx);
        }
      else if(w == "X") into(bl.X[parity ? 1 : 0].a);
      else if(w == "Y") into(bl.Y[parity ? 1 : 0].a);
      else throw std::runtime_error("orc_set_array: unknown array " + w);
    }
  ORC_CATCH(o)
}

int orc_terminate_reason(void *h)
{
  return static_cast<Oracle *>(h)->terminate_reason;
}
const char *orc_terminate_string(void *h)
{
  int r = static_cast<Oracle *>(h)->terminate_reason;
  return r < 0 ? "" : terminate_names[r];
}

// scalar names follow iterations.json keys (print_iteration.cxx:91-104) and
// out.txt keys (save_solution.cxx:32-37)
const char *orc_get_scalar(void *h, const char *name)
{
  Oracle *o = enter(h);
  const std::string n(name);
  const F *v = nullptr;
  F pe;
  if(n == "mu") v = &o->mu;
  else if(n == "P-obj" || n == "primalObjective") v = &o->primal_objective;
  else if(n == "D-obj" || n == "dualObjective") v = &o->dual_objective;
  else if(n == "gap" || n == "dualityGap") v = &o->duality_gap;
  else if(n == "P-err") v = &o->primal_error_P;
  else if(n == "p-err") v = &o->primal_error_p;
  else if(n == "D-err" || n == "dualError") v = &o->dual_error;
  else if(n == "R-err") v = &o->R_error;
  else if(n == "P-step") v = &o->primal_step_length;
  else if(n == "D-step") v = &o->dual_step_length;
  else if(n == "beta") v = &o->beta_corrector;
  else if(n == "Q_cond_number") v = &o->Q_cond_number;
  else if(n == "max_block_cond_number") v = &o->max_block_cond_number;
  else if(n == "primalError") { pe = o->primal_error(); v = &pe; }
  else if(n == "block_name") { o->strbuf = o->max_block_cond_number_name; return o->strbuf.c_str(); }
  else if(n == "seconds.syrk_Q") { o->strbuf = std::to_string(o->seconds_syrk_Q); return o->strbuf.c_str(); }
  else { o->strbuf = ""; return o->strbuf.c_str(); }
  o->strbuf = to_str(*v);
  return o->strbuf.c_str();
}

// which in {"x","y","X","Y","dx","dy","dX","dY","Q","S","L","P","AXinv","AY",
// "dual_residues","primal_residues","primal_residue_p"}; column-major,
// whitespace separated.  j/parity ignored where not applicable.
const char *orc_get_array(void *h, const char *which, int j, int parity)
{
  Oracle *o = enter(h);
  const std::string w(which);
  std::ostringstream ss;
  auto dumpv = [&](const std::vector<F> &v) {
    for(auto &e : v)
      ss << to_str(e) << "\n";
  };
  auto dumpm = [&](const Mat &m) { dumpv(m.a); };
  if(w == "y") dumpv(o->y);
  else if(w == "dy") dumpv(o->dy);
  else if(w == "primal_residue_p") dumpv(o->primal_residue_p);
  else if(w == "Q") dumpm(o->Q);
  else
    {
      Block &bl = o->blk.at(j);
      if(w == "x") dumpv(bl.x);
      else if(w == "dx") dumpv(bl.dx);
      else if(w == "dual_residues") dumpv(bl.dual_residues);
      else if(w == "X") dumpm(bl.X[parity]);
      else if(w == "Y") dumpm(bl.Y[parity]);
      else if(w == "dX") dumpm(bl.dX[parity]);
      else if(w == "dY") dumpm(bl.dY[parity]);
      else if(w == "primal_residues") dumpm(bl.primal_residues[parity]);
      else if(w == "S") dumpm(bl.S);
      else if(w == "L") dumpm(bl.L);
      else if(w == "P") dumpm(bl.Poff);
      else if(w == "AXinv") dumpm(bl.AXinv[parity]);
      else if(w == "AY") dumpm(bl.AY[parity]);
      else if(w == "Xc") dumpm(bl.Xc[parity]);
      else if(w == "Yc") dumpm(bl.Yc[parity]);
    }
  o->strbuf = ss.str();
  return o->strbuf.c_str();
}

// ---- raw mpf_t records (tests of the product's binary number path) ----------
// One number = 2 + limbs64 words copied straight from the mpf_t: _mp_size, _mp_exp, _mp_d
// (the low limbs are dropped when the mpf_t holds more than limbs64).
static void put_record(const F &f, uint64_t *rec, int limbs64)
{
  const long size = f.v->_mp_size, n = size < 0 ? -size : size, keep = n < limbs64 ? n : limbs64;
  for(int i = 0; i < limbs64 + 2; ++i)
    rec[i] = 0;
  rec[0] = (uint64_t)(long long)(size < 0 ? -keep : keep);
  rec[1] = (uint64_t)(long long)f.v->_mp_exp;
  for(long i = 0; i < keep; ++i)
    rec[2 + i] = f.v->_mp_d[n - keep + i];
}
// which: "bases_even", "bases_odd" (row-major rows x K), "B" (row-major P x N), "c", "b", "constant",
// or any orc_get_array name (column-major).  Returns the element count; writes when capacity allows.
long orc_get_records(void *h, const char *which, int j, int parity, int limbs64, uint64_t *out, long capacity)
{
  Oracle *o = enter(h);
  const std::string w(which);
  std::vector<const F *> v;
  auto rowmajor = [&](const Mat &m) {
    for(int r = 0; r < m.h; ++r)
      for(int c = 0; c < m.w; ++c)
        v.push_back(&m(r, c));
  };
  auto colmajor = [&](const Mat &m) {
    for(auto &e : m.a)
      v.push_back(&e);
  };
  auto vec = [&](const std::vector<F> &x) {
    for(auto &e : x)
      v.push_back(&e);
  };
  if(w == "b") vec(o->b);
  else if(w == "constant") v.push_back(&o->objective_const);
  else if(w == "y") vec(o->y);
  else if(w == "dy") vec(o->dy);
  else
    {
      Block &bl = o->blk.at(j);
      if(w == "bases_even") rowmajor(bl.bases[0]);
      else if(w == "bases_odd") rowmajor(bl.bases[1]);
      else if(w == "B") rowmajor(bl.B);
      else if(w == "c") vec(bl.c);
      else if(w == "x") vec(bl.x);
      else if(w == "dx") vec(bl.dx);
      else if(w == "X") colmajor(bl.X[parity]);
      else if(w == "Y") colmajor(bl.Y[parity]);
      else return -1;
    }
  if(out && capacity >= (long)v.size())
    for(size_t i = 0; i < v.size(); ++i)
      put_record(*v[i], out + i * (size_t)(limbs64 + 2), limbs64);
  return (long)v.size();
}

// The inverse of orc_get_records for the solver state (x, X, y, Y): bit-exact restore of a state saved as records
// (the generator of the full-size fixtures continues an interrupted run of hours from it:
// tests/golden/synthetic/make_synthetic_golden.py).  A record must fit the mpf_t's allocation (_mp_prec + 1 limbs), which
// it does when it was written by an oracle of the same precision with limbs64 >= that count.
int orc_set_records(void *h, const char *which, int j, int parity, int limbs64, const uint64_t *in, long count)
{
  Oracle *o = enter(h);
  ORC_TRY(o)
  const std::string w(which);
  std::vector<F> *dst = nullptr;
  if(w == "y") dst = &o->y;
  else
    {
      Block &bl = o->blk.at(j);
      if(w == "x") dst = &bl.x;
      else if(w == "X") dst = &bl.X[parity ? 1 : 0].a;
      else if(w == "Y") dst = &bl.Y[parity ? 1 : 0].a;
      else throw std::runtime_error("orc_set_records: unknown array " + w);
    }
  if((long)dst->size() != count)
    throw std::runtime_error("orc_set_records: wrong element count for " + w);
  for(long i = 0; i < count; ++i)
    {
      const uint64_t *rec = in + (size_t)i * (size_t)(limbs64 + 2);
      F &f = (*dst)[(size_t)i];
      const long size = (long)(long long)rec[0], n = size < 0 ? -size : size;
      if(n > limbs64 || n > (long)f.v->_mp_prec + 1)
        throw std::runtime_error("orc_set_records: a record is wider than the oracle's mantissa");
      for(long k = 0; k < n; ++k)
        f.v->_mp_d[k] = rec[2 + k];
      f.v->_mp_size = (int)size;
      f.v->_mp_exp = (mp_exp_t)(long long)rec[1];
    }
  ORC_CATCH(o)
}

// ---- kernel-level oracles (calculate_matrix_square.test.cxx recipe) --------
// Exact integer syrk: inputs are P' as decimal *integers* (rows x cols,
// column-major), output upper triangle of Q' = P'^T P' as decimal integers
// (column-major N x N, lower part zero).
const char *orc_int_syrk(void *h, int rows, int cols, const char *Ptxt)
{
  Oracle *o = enter(h);
  std::vector<Z> P((size_t)rows * cols);
  std::istringstream in(Ptxt);
  std::string tok;
  for(auto &z : P)
    {
      in >> tok;
      mpz_set_str(z.v, tok.c_str(), 10);
    }
  mpz_t acc;
  mpz_init(acc);
  std::ostringstream ss;
  for(int j = 0; j < cols; ++j)
    for(int i = 0; i < cols; ++i)
      {
        mpz_set_ui(acc, 0);
        if(i <= j)
          for(int r = 0; r < rows; ++r)
            mpz_addmul(acc, P[(size_t)i * rows + r].v, P[(size_t)j * rows + r].v);
        char *s = mpz_get_str(nullptr, 10, acc);
        ss << s << "\n";
        free(s);
      }
  mpz_clear(acc);
  o->strbuf = ss.str();
  return o->strbuf.c_str();
}

// The value GMP gives a decimal string when parsed at `prec_bits` (the reference parses its
// Solver_Parameters before --precision is applied), as an exact decimal expansion.
const char *orc_parse_exact(void *h, const char *value, int prec_bits)
{
  Oracle *o = enter(h);
  F v = from_str(value, prec_bits ? prec_bits : 0);
  // exact rational n / 2^k (mpq_set_f is exact); the caller expands it in decimal
  mpq_t q;
  mpq_init(q);
  mpq_set_f(q, v.v);
  char *num = mpz_get_str(nullptr, 10, mpq_numref(q));
  const unsigned long k = mpz_sizeinbase(mpq_denref(q), 2) - 1; // denominator is a power of two
  o->strbuf = std::string(num) + " " + std::to_string(k);
  free(num);
  mpq_clear(q);
  return o->strbuf.c_str();
}

// Scalar mpf ops for arithmetic-level parity tests.  op in
// {"add","sub","mul","div","sqrt"}; operands/results decimal strings.
const char *orc_scalar_op(void *h, const char *op, const char *a, const char *b)
{
  Oracle *o = enter(h);
  F x = from_str(a), y = from_str(b), r;
  const std::string s(op);
  if(s == "add") r = x + y;
  else if(s == "sub") r = x - y;
  else if(s == "mul") r = x * y;
  else if(s == "div") r = x / y;
  else if(s == "sqrt") r = fsqrt(x);
  o->strbuf = to_str(r);
  return o->strbuf.c_str();
}

// min_eigenvalue.cxx:8-33 on one symmetric n x n matrix given column-major (the tests aim clustered spectra at it)
const char *orc_min_eigenvalue(void *h, int n, const char *txt)
{
  Oracle *o = enter(h);
  std::vector<F> v;
  parse_list(txt, v, (size_t)n * n, "A");
  Mat A(n, n);
  for(int j = 0; j < n; ++j)
    for(int i = 0; i < n; ++i)
      A(i, j) = v[(size_t)i + (size_t)j * n];
  o->strbuf = to_str(min_eigenvalue_sym(A));
  return o->strbuf.c_str();
}

// ---------------------------------------------------------------------------
// Steps 2 and 4 of the reference's OWN algorithm for Q' = P'^T P' on GMP (CPU-baseline leg of bench.py; the
// description and step 3, one dsyrk per prime, are in oracle/bigint_syrk_blas.py):
//   step 2  compute_block_residues.cxx:223-330 (_fmpz_multi_mod_precomp / fmpz_get_nmod, fmpz_mul_blas_util.hxx:71,84):
//           every entry of P' modulo every prime, centred, as fp64 -- here mpz_fdiv_ui per (entry, prime), OpenMP over entries
//   step 4  restore_bigint_from_residues.hxx:28 (fmpz_multi_CRT_ui, sign = 1), restore_and_reduce.cxx:64-77, the comb of
//           fmpz/Fmpz_Comb.cxx:75-109: x = sum_q (r_q mod p_q) C_q mod M, C_q = (M / p_q) ((M / p_q)^-1 mod p_q), the
//           representative of smallest magnitude -- here mpz_addmul_ui over the primes, one mpz_fdiv_r, OpenMP over outputs
// FLINT's versions use remainder / product trees; these are the plain compiled forms of the same maps.
// ---------------------------------------------------------------------------
// txt != nullptr: `count` decimal integers (exactness tests); else the pseudo-random entries first ... first + count of a
// stream |v| < 2^bits fixed by `seed` (generated before the clock starts).  out: np x count doubles.  seconds: wall time of the reduction alone.
int orc_refq_residues(const char *txt, long count, int bits, unsigned long seed, unsigned long first, const unsigned long *primes, int np,
                      double *out, double *seconds)
{
  try
    {
      std::vector<Z> v((size_t)count);
      if(txt)
        {
          std::istringstream in(txt);
          std::string tok;
          for(auto &z : v)
            {
              if(!(in >> tok) || mpz_set_str(z.v, tok.c_str(), 10) != 0)
                return 4;
            }
        }
      else
        {
          // counter-based: entry (first + i) is the same number whatever the chunking and the thread count
          const int nl = (bits + 63) / 64;
#pragma omp parallel for schedule(static)
          for(long i = 0; i < count; ++i)
            {
              uint64_t limb[64];
              for(int k = 0; k < nl && k < 64; ++k)
                {
                  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)(first + (unsigned long)i) * 64u + (uint64_t)k + 1u);
                  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                  limb[k] = z ^ (z >> 31);
                }
              if(bits % 64)
                limb[nl - 1] &= (~0ull) >> (64 - bits % 64);
              mpz_import(v[(size_t)i].v, (size_t)nl, -1, 8, 0, 0, limb);
              if(limb[0] & 1u)
                mpz_neg(v[(size_t)i].v, v[(size_t)i].v);
            }
        }
      const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(static)
      for(long i = 0; i < count; ++i)
        for(int q = 0; q < np; ++q)
          {
            const unsigned long p = primes[q];
            const unsigned long r = mpz_fdiv_ui(v[(size_t)i].v, p); // in [0, p), sign respected (floor division)
            out[(size_t)q * (size_t)count + (size_t)i] = r > p / 2 ? (double)r - (double)p : (double)r;
          }
      if(seconds)
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      return 0;
    }
  catch(...)
    {
      return 4;
    }
}
// prod: np x n doubles (exact integers: the dsyrk sums of centred residues).  txt_out != nullptr: the n results as decimal
// integers separated by newlines (tests); always: xor of the low limbs of all results in *checksum (keeps the work alive).
int orc_refq_crt(const double *prod, long n, const unsigned long *primes, int np, char *txt_out, size_t txt_cap, size_t *needed,
                 unsigned long *checksum, double *seconds)
{
  try
    {
      Z M;
      mpz_set_ui(M.v, 1);
      for(int q = 0; q < np; ++q)
        mpz_mul_ui(M.v, M.v, primes[q]);
      std::vector<Z> C((size_t)np);
      for(int q = 0; q < np; ++q)
        {
          Z Mq, inv, pq;
          mpz_divexact_ui(Mq.v, M.v, primes[q]);
          mpz_set_ui(pq.v, primes[q]);
          if(!mpz_invert(inv.v, Mq.v, pq.v))
            return 4;
          mpz_mul(C[(size_t)q].v, Mq.v, inv.v);
        }
      Z half;
      mpz_fdiv_q_2exp(half.v, M.v, 1);
      std::vector<Z> res(txt_out ? (size_t)n : 0);
      unsigned long sum = 0;
      const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel reduction(^ : sum)
      {
        Z x;
#pragma omp for schedule(static)
        for(long i = 0; i < n; ++i)
          {
            mpz_set_ui(x.v, 0);
            for(int q = 0; q < np; ++q)
              {
                const long long p = (long long)primes[q];
                long long r = (long long)prod[(size_t)q * (size_t)n + (size_t)i] % p; // |prod| < 2^53: exact
                if(r < 0)
                  r += p;
                mpz_addmul_ui(x.v, C[(size_t)q].v, (unsigned long)r);
              }
            mpz_fdiv_r(x.v, x.v, M.v);
            if(mpz_cmp(x.v, half.v) > 0)
              mpz_sub(x.v, x.v, M.v);
            sum ^= mpz_getlimbn(x.v, 0) * (mpz_size(x.v) ? 1ul : 0ul);
            if(txt_out)
              mpz_set(res[(size_t)i].v, x.v);
          }
      }
      if(seconds)
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if(checksum)
        *checksum = sum;
      if(txt_out || needed)
        {
          std::string s;
          for(auto &z : res)
            {
              char *t = mpz_get_str(nullptr, 10, z.v);
              s += t;
              s += "\n";
              free(t);
            }
          if(needed)
            *needed = s.size() + 1;
          if(txt_out && txt_cap >= s.size() + 1)
            std::memcpy(txt_out, s.c_str(), s.size() + 1);
          else if(txt_out)
            return 5;
        }
      return 0;
    }
  catch(...)
    {
      return 4;
    }
}
} // extern "C"
