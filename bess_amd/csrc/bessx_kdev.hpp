#ifndef BESSX_KDEV_HPP
#define BESSX_KDEV_HPP
// bessx_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind libbessx.so.
//
// Kernel inventory (K-numbers are SURVEY.md section 2.3; reference lines are under /root/reference):
//   k_transpose_in   upload helper: row-major host chunk -> column-major padded X
//   k_col_stats / k_col_scale   K11  Normalize*/add_weight      src/normalize.cpp:20-85, src/Data.h:70-77
//   k_xtv            K1/K2  d = X^T v (and sum x^2 h)            src/Algorithm.h:1109,1236,1240-1246
//   k_score_*        sacrifice scores bd                        src/Algorithm.h:1112-1126,1238-1260
//   k_topk           K4  max_k                                  src/utilities.cpp:179-188
//   k_gram           K6  X_A^T diag(w) X_A on fp64 MFMA         src/Algorithm.h:1134,1171,1199,1299
//   k_gram_reduce    fixed-order sum of the row-slab partials
//   k_chol           K7  Cholesky + triangular solves, one WG   src/Algorithm.h:1134 (QR), :1171 (LDLT)
//   k_resid_lm       r = m*(y - X_A b_A - c), SSE (train / test) src/Algorithm.h:1109, src/Metric.h:147,190
//   k_fit_begin / k_commit   Algorithm::fit bookkeeping          src/Algorithm.h:141-170
//
// Conventions: X is column-major with leading dimension ld (a multiple of 1024 rows when
// n >= 4096), pad rows are zero.  All reductions use a fixed tree: results are bitwise
// reproducible run to run (no floating-point atomics anywhere).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "bessx_dev.h"

namespace bessx {
#ifdef BESSX_KTRACE
// development aid: start time stamp (100 MHz wall clock) of every traced kernel, in launch order
__device__ unsigned long long g_ktrace[1 << 16];
__device__ unsigned int g_ktrace_n;
#define KT(id)                                                                   \
  do {                                                                           \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                \
      const unsigned n_ = atomicAdd(&g_ktrace_n, 1u);                            \
      if (n_ < (1u << 16)) g_ktrace[n_] = (wall_clock64() << 8) | (unsigned)(id); \
    }                                                                            \
  } while (0)
#else
#define KT(id)
#endif
#ifdef BESSX_KTRACE
__device__ unsigned long long g_phase[32];
#define PH_BEGIN() unsigned long long tph_ = wall_clock64()
#define PH(i)                                                   \
  do {                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if (threadIdx.x == 0) {                                     \
      unsigned long long now_ = wall_clock64();                 \
      atomicAdd(&g_phase[i], now_ - tph_);                      \
      tph_ = now_;                                              \
    }                                                           \
  } while (0)
#define PH_COUNT()                                          \
  do {                                                        \
    if (threadIdx.x == 0) atomicAdd(&g_phase[31], 1ull);      \
  } while (0)
#else
#define PH_BEGIN()
#define PH(i)
#define PH_COUNT()
#endif

constexpr int COV_R = 32;   // right-hand-side columns per panel group (two MFMA tiles)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int TOPK_E = 32;  // keys per thread
constexpr int CH_W = 8;                                                   // waves
constexpr int CH_MT = 16;                                                 // max tile rows
constexpr int CH_LDT = 17;                                                // padded tile row stride (doubles)
constexpr int GRP_MAX = 16;  // largest group size built

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Wave-wide reductions without LDS permutes: four DPP exchanges inside each row of 16 lanes (mirror, half mirror, the
// two quad swaps: every lane of a row ends with the row's result), then the four rows through the scalar unit
// (v_readlane).  A __shfl_xor butterfly costs one LDS round trip per stage and 32-bit word.
#define BESSX_DPP32(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false)
__device__ __forceinline__ double dpp_f64(double v, const int stage) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (stage) {
    case 0: lo = BESSX_DPP32(lo, 0x140); hi = BESSX_DPP32(hi, 0x140); break;  // row_mirror
    case 1: lo = BESSX_DPP32(lo, 0x141); hi = BESSX_DPP32(hi, 0x141); break;  // row_half_mirror
    case 2: lo = BESSX_DPP32(lo, 0x4E); hi = BESSX_DPP32(hi, 0x4E); break;    // quad_perm [2,3,0,1]
    default: lo = BESSX_DPP32(lo, 0xB1); hi = BESSX_DPP32(hi, 0xB1); break;   // quad_perm [1,0,3,2]
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int dpp_i32(int v, const int stage) {
  switch (stage) {
    case 0: return BESSX_DPP32(v, 0x140);
    case 1: return BESSX_DPP32(v, 0x141);
    case 2: return BESSX_DPP32(v, 0x4E);
    default: return BESSX_DPP32(v, 0xB1);
  }
}
__device__ __forceinline__ double readlane_f64(double v, const int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// minimum of mn and maximum of mx over the wave, in every lane (min / max are exact: the order does not matter)
__device__ __forceinline__ void wave_min_max(double &mn, double &mx) {
#pragma unroll
  for (int st = 0; st < 4; st++) {
    mn = fmin(mn, dpp_f64(mn, st));
    mx = fmax(mx, dpp_f64(mx, st));
  }
  mn = fmin(fmin(readlane_f64(mn, 0), readlane_f64(mn, 16)), fmin(readlane_f64(mn, 32), readlane_f64(mn, 48)));
  mx = fmax(fmax(readlane_f64(mx, 0), readlane_f64(mx, 16)), fmax(readlane_f64(mx, 32), readlane_f64(mx, 48)));
}
// arg-max of (key, lower index on ties) over the wave, in every lane: a total order, so the result does not depend on
// the order of the comparisons
__device__ __forceinline__ void wave_argmax(unsigned long long &best, int &besti) {
  auto take = [&](unsigned long long ok, int oi) {
    if (ok > best || (ok == best && oi < besti)) {
      best = ok;
      besti = oi;
    }
  };
#pragma unroll
  for (int st = 0; st < 4; st++) {
    const double od = dpp_f64(__longlong_as_double((long long)best), st);
    const int oi = dpp_i32(besti, st);
    take((unsigned long long)__double_as_longlong(od), oi);
  }
  const double bd_ = __longlong_as_double((long long)best);
  unsigned long long rk[4];
  int ri[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    rk[q] = (unsigned long long)__double_as_longlong(readlane_f64(bd_, 16 * q));
    ri[q] = __builtin_amdgcn_readlane(besti, 16 * q);
  }
  best = rk[0];
  besti = ri[0];
#pragma unroll
  for (int q = 1; q < 4; q++) take(rk[q], ri[q]);
}

// Copy of the result block (control words, loss sums, the first kcopy coefficients and indices) from `src` to `dst`
// by the whole workgroup.  The block was (partly) written by this very kernel, so the reads go to L2 (relaxed
// agent-scope atomic loads), but -- unlike volatile accesses, which the compiler keeps in program order and waits
// for one by one -- they are all in flight before the first store.
__device__ __forceinline__ void copy_result_block(const PubArgs &pa, const unsigned char *src, unsigned char *dst) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const unsigned long long *d8 = reinterpret_cast<const unsigned long long *>(src);
  unsigned long long *h8 = reinterpret_cast<unsigned long long *>(dst);
  const int *d4 = reinterpret_cast<const int *>(src + pa.off_a);
  int *h4 = reinterpret_cast<int *>(dst + pa.off_a);
  auto ld8 = [](const unsigned long long *q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld4 = [](const int *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // one element of every region per thread and round (the regions are a few hundred elements at most)
  const int nc = pa.ctrl_bytes / 8;
  const int rounds = (max(max(nc, pa.n_sse), pa.kcopy) + nt - 1) / nt;
  for (int r = 0; r < rounds; r++) {
    const int i = r * nt + tid;
    const bool c = i < nc, e = i < pa.n_sse, k = i < pa.kcopy;
    const unsigned long long vc = c ? ld8(d8 + i) : 0ull, ve = e ? ld8(d8 + pa.off_sse / 8 + i) : 0ull;
    const unsigned long long vb = k ? ld8(d8 + pa.off_b / 8 + i) : 0ull;
    const int va = k ? ld4(d4 + i) : 0;
    if (c) h8[i] = vc;
    if (e) h8[pa.off_sse / 8 + i] = ve;
    if (k) h8[pa.off_b / 8 + i] = vb;
    if (k) h4[i] = va;
  }
}

// The work of k_publish as the tail of another single-block kernel: every thread of the block calls it after a
// barrier that follows the last write to the result block.
__device__ __forceinline__ void publish_body(const PubArgs &pa) {
  const int tid = threadIdx.x;
  if (tid == 0 && pa.count_ptr != nullptr) pa.seq_host[1] = (unsigned long long)pa.count_ptr[0];
  copy_result_block(pa, pa.dev, pa.host);
  // release: every thread's system-scope fence, the barrier, then the flag.  The flag store itself can be relaxed --
  // a release fence followed by a relaxed atomic store is a release operation on it (and a second system-scope
  // release by thread 0 would be one more round trip to host memory)
  // (the full fence is needed: with only s_waitcnt vmcnt(0) per wave the host saw stale blocks at once, tools/soak.py)
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(pa.seq_host, pa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tail of the last kernel of a chained batch: only a device-side snapshot of what a publication would copy; the
// publication itself rides on the next launch of the chain (second workgroup of k_topk) or a k_publish launch.
__device__ __forceinline__ void snapshot_body(const PubArgs &pa) {
  copy_result_block(pa, pa.dev, pa.snap);
  if (threadIdx.x == 0)
    *reinterpret_cast<int *>(pa.snap + pa.snap_count_off) = pa.count_ptr != nullptr ? pa.count_ptr[0] : 0;
}

// Block-wide sum for 256-thread blocks, fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double *sm /*>=4*/) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((sm[0] + sm[1]) + sm[2]) + sm[3];
  __syncthreads();
  return r;
}

// ---- device functions shared by several translation units (lifted out of their families' files)
__device__ __forceinline__ void tile_of(int t, int &I, int &J) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) i++;
  while (i * (i + 1) / 2 > t) i--;
  I = i;
  J = t - i * (i + 1) / 2;
}

// broadcast the value lane `src` (compile-time constant after unrolling) holds to the whole wave via SGPRs
__device__ __forceinline__ double bcast_lane(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// fz.G != nullptr (covariance mode of the LM fit): the Gram entries are gathered from the column cache
// (G[row A_a, column slot_of[A_b]]) instead of read from Gt, and the kernel ends with the work of k_commit -- two
// launches less per PDAS iteration.
// position of entry (row, col) of a 16 x 16 tile in the fp64-MFMA accumulator layout; index of tile (I, J), I >= J
__device__ __forceinline__ int tile_elem(int row, int col) { return ((((row & 3) << 4) | col) << 2) | (row >> 2); }

__device__ __forceinline__ size_t tile_id(int I, int J) { return (size_t)I * (I + 1) / 2 + J; }

// End of a PDAS iteration: beta <- 0; beta[A] = beta_A; A_list.col(l) = A; stop if A == A_list.col(ll), ll < l.
// sol holds the solved coefficients; with an intercept (GLM) sol[0] is coef0 and sol[1..] the slopes.
// body of k_commit for any block size (also the tail of the fused k_chol of the covariance mode)
__device__ __forceinline__ void commit_body(FitCtrl *__restrict__ ctrl, int slot, int T0,
                                            const int *__restrict__ A_new, const double *__restrict__ sol,
                                            int has_intercept, int wait_chain, int *__restrict__ A_cur,
                                            double *__restrict__ b_cur, double *__restrict__ beta_dense,
                                            int *__restrict__ hist, double *__restrict__ hist_beta,
                                            double *__restrict__ hist_coef0, int hist_stride, int *same_any_sh,
                                            unsigned char *__restrict__ inA) {
  // inA (optional): membership flags of the current active set, kept for the repeated-set shortcut of k_cov_d
  const int nt = blockDim.x;
  const int l = slot;
  if (ctrl->same_prev) {
    // A == A_list.col(l-1): the re-fit reproduces the current coefficients; record and stop
    for (int i = threadIdx.x; i < T0; i += nt) {
      hist[(size_t)l * hist_stride + i] = A_cur[i];
      hist_beta[(size_t)l * hist_stride + i] = b_cur[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      hist_coef0[l] = ctrl->coef0;
      ctrl->l = l;
      ctrl->done = 1;
      ctrl->d_fresh = 1;  // the score-pass sums in memory belong to the final coefficients
    }
    return;
  }
  if (wait_chain && !ctrl->irls_done) return;  // IRLS / Newton chain still running: the host re-issues
  const int kc = ctrl->k_cur;
  if (threadIdx.x == 0) *same_any_sh = 0;
  for (int i = threadIdx.x; i < kc; i += nt) {
    beta_dense[A_cur[i]] = 0.0;
    if (inA != nullptr) inA[A_cur[i]] = 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T0; i += nt) {
    int a = A_new[i];
    double b = sol[i + (has_intercept ? 1 : 0)];
    A_cur[i] = a;
    b_cur[i] = b;
    beta_dense[a] = b;
    if (inA != nullptr) inA[a] = 1;
    hist[(size_t)l * hist_stride + i] = a;
    hist_beta[(size_t)l * hist_stride + i] = b;
  }
  __syncthreads();
  // compare with every earlier column (including the all-zero column 0)
  for (int ll = 0; ll < l; ll++) {
    int diff = 0;
    for (int i = threadIdx.x; i < T0; i += nt) diff |= (hist[(size_t)ll * hist_stride + i] != A_new[i]);
    diff = __syncthreads_or(diff);
    if (!diff && threadIdx.x == 0) *same_any_sh = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (has_intercept) ctrl->coef0 = sol[0];
    hist_coef0[l] = ctrl->coef0;
    ctrl->k_cur = T0;
    ctrl->l = l;
    ctrl->done = *same_any_sh;
    ctrl->d_fresh = 0;  // coefficients changed after the last score pass
    ctrl->irls_done = 0;
    ctrl->irls_last = ctrl->irls_steps;
    ctrl->irls_steps = 0;
  }
}

// Rank-revealing fallback of the k x k solves (k_chol, k_ldlt_fallback's successor).  The reference solves its normal
// equations with factorizations that survive a singular or indefinite matrix: column-pivoted Householder QR of the LM
// Gram (src/Algorithm.h:1131-1135), Eigen's LDLT -- diagonal pivoting, a zero pivot gives a zero coefficient -- for the
// IRLS and Newton systems (:1171, :1199, :1299, :1473).  The fast kernels here (Cholesky in registers, conjugate
// gradients) assume positive definite; when a pivot collapses (exactly dependent columns: duplicates both in the active
// set, more columns than independent rows) or turns negative (the Cox Newton matrix under a large ridge, whose sign the
// reference has as written), the system is solved again by LDL^T WITH DIAGONAL PIVOTING on a dense copy in global
// memory: at every step the largest remaining |diagonal| (first of equals, like Eigen's maxCoeff) is the pivot; a
// pivot below 1e-11 of the largest diagonal counts as zero: its unknown is set to 0 and dropped -- the basic solution
// of the consistent system, the coefficient of a duplicated column going to the copy that is eliminated first.
// (Eigen's own tests are "exactly zero" / eps^2-relative: on exactly dependent columns whether they fire is decided by
// rounding, so the reference's numbers there are not reproducible by any other arithmetic; see DESIGN.md.)
// One workgroup, O(m^3 / NT) global-memory steps: rare and small.  A: m x m, both triangles, leading dimension m;
// b: right-hand side in, solution out; dv / perm / zf: m entries of scratch each.  Returns nothing; non-finite
// results are the caller's to flag.
template <int NT>
__device__ void sym_pivoted_solve(double *__restrict__ A, int m, double *__restrict__ b, double *__restrict__ dv,
                                  int *__restrict__ perm, int *__restrict__ zf) {
  __shared__ double s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  __shared__ double s_scal;
  __shared__ int s_piv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the largest |diagonal| and, at every step, the first index of the largest remaining one
  auto arg_absmax_diag = [&](int from) {
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int j = from + tid; j < m; j += NT) {
      const double v = fabs(A[(size_t)j * m + j]);
      if (v > bv || (v == bv && j < bi)) {
        bv = v;
        bi = j;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    __syncthreads();
    if (lane == 0) {
      s_v[wave] = bv;
      s_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      double v = s_v[0];
      int i = s_i[0];
      for (int w = 1; w < NT / 64; w++)
        if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) {
          v = s_v[w];
          i = s_i[w];
        }
      s_scal = v;
      s_piv = i == 0x7fffffff ? from : i;  // (all NaN: any pivot will do, the result is flagged non-finite)
    }
    __syncthreads();
  };
  arg_absmax_diag(0);
  const double tol = 1e-11 * s_scal;
  for (int k = 0; k < m; k++) {
    arg_absmax_diag(k);
    const int piv = s_piv;
    if (tid == 0) perm[k] = piv;
    if (piv != k) {  // symmetric exchange of k and piv: rows, then columns
      for (int c = tid; c < m; c += NT) {
        const double t0 = A[(size_t)c * m + k];
        A[(size_t)c * m + k] = A[(size_t)c * m + piv];
        A[(size_t)c * m + piv] = t0;
      }
      __syncthreads();
      for (int r = tid; r < m; r += NT) {
        const double t0 = A[(size_t)k * m + r];
        A[(size_t)k * m + r] = A[(size_t)piv * m + r];
        A[(size_t)piv * m + r] = t0;
      }
      if (tid == 0) {
        const double t0 = b[k];
        b[k] = b[piv];
        b[piv] = t0;
      }
      __syncthreads();
    }
    const double d = A[(size_t)k * m + k];
    const bool dead = !(fabs(d) > tol);
    if (tid == 0) {
      dv[k] = dead ? 0.0 : d;
      zf[k] = dead ? 1 : 0;
    }
    const int r = m - k - 1;
    if (dead) {
      for (int i = k + 1 + tid; i < m; i += NT) A[(size_t)k * m + i] = 0.0;  // no coupling through a dropped unknown
      __syncthreads();
      continue;
    }
    for (int i = k + 1 + tid; i < m; i += NT) A[(size_t)k * m + i] = A[(size_t)k * m + i] / d;  // column k: l_ik
    __syncthreads();
    // trailing block (both triangles): a_ij -= l_ik a_kj, a_kj still unscaled in row k
    for (long idx = tid; idx < (long)r * r; idx += NT) {
      const int i = k + 1 + (int)(idx % r), j = k + 1 + (int)(idx / r);
      A[(size_t)j * m + i] -= A[(size_t)k * m + i] * A[(size_t)j * m + k];
    }
    __syncthreads();
  }
  // P b -> L^-1 -> D^+ -> L^-T -> P^T
  for (int k = 0; k < m; k++) {
    const double bk = b[k];
    __syncthreads();
    for (int i = k + 1 + tid; i < m; i += NT) b[i] -= A[(size_t)k * m + i] * bk;
    __syncthreads();
  }
  for (int i = tid; i < m; i += NT) b[i] = zf[i] ? 0.0 : b[i] / dv[i];
  __syncthreads();
  for (int k = m - 1; k >= 0; k--) {
    double part = 0.0;
    for (int i = k + 1 + tid; i < m; i += NT) part += A[(size_t)k * m + i] * b[i];
    part = wave_sum(part);
    __syncthreads();
    if (lane == 0) s_v[wave] = part;
    __syncthreads();
    if (tid == 0) {
      double t0 = 0.0;
      for (int w = 0; w < NT / 64; w++) t0 += s_v[w];
      b[k] -= t0;
    }
    __syncthreads();
  }
  if (tid == 0)
    for (int k = m - 1; k >= 0; k--) {
      const int pv = perm[k];
      if (pv != k) {
        const double t0 = b[k];
        b[k] = b[pv];
        b[pv] = t0;
      }
    }
  __syncthreads();
}

// IRLS step t, convergence test (single block).  Logistic (:1181): |ll0 - ll1| / (0.1 + |ll1|) < 1e-6, result =
// iterate BEFORE the last solve, at most 30 tests; Poisson (:1315): |ll0 - ll1| / |0.1 + ll0| < 1e-6 with
// ll0 = 1e5 initially, result = the latest iterate, at most 50 solves.  On exit with irls_done the
// result sits in bprev (what k_commit reads).  Returns (block-uniform) whether the chain has ended.
template <int NT>
__device__ __forceinline__ bool irls_check_body(FitCtrl *__restrict__ ctrl, int t, int fam,
                                                const double *__restrict__ llpart, int nblk, int m,
                                                double *__restrict__ bcur, double *__restrict__ bprev) {
  __shared__ double sm[NT / 64];
  __shared__ int fin;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += NT) s += llpart[b];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = 0.0;
    for (int q = 0; q < NT / 64; q++) s += sm[q];
    int done = 0;
    if (fam == 2) {
      if (t == 0) {
        ctrl->ll0 = s;
      } else {
        if (fabs(ctrl->ll0 - s) / (0.1 + fabs(s)) < 1e-6) done = 1;  // result: bprev (iterate before last solve)
        if (!done) ctrl->ll0 = s;
      }
      fin = done ? 2 : ((t == 30) ? 1 : 0);  // 2: keep bprev; 1: bprev <- bcur then stop; 0: bprev <- bcur, go on
    } else {
      if (t == 0) {
        ctrl->ll0 = 1e5;
      } else {
        if (fabs(ctrl->ll0 - s) / fabs(0.1 + ctrl->ll0) < 1e-6) done = 1;
        if (!done) ctrl->ll0 = s;
      }
      fin = (done || t == 50) ? 1 : 0;  // result is always the latest iterate
    }
  }
  __syncthreads();
  const int f = fin;
  if (f != 2)
    for (int i = threadIdx.x; i < m; i += NT) bprev[i] = bcur[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    ctrl->irls_steps = t + 1;
    if (f != 0) ctrl->irls_done = 1;
  }
  return f != 0;
}

// ------------------------------------------------------------------------------------------
// GLM families (logistic: src/Algorithm.h:1138-1264, src/logistic.cpp:15-59; Poisson: :1266-1368).
// FAM = 2 logistic, 3 Poisson.  Rows are handled two per thread (16-byte loads) like k_resid_lm.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double clampv(double v, double c) { return v > c ? c : (v < -c ? -c : v); }

// linear predictor of two consecutive rows over k columns (optionally offset by one for an intercept slot)
__device__ __forceinline__ d2 lin_pred2(const double *__restrict__ X, long ld, long i, const int *__restrict__ A,
                                        const double *__restrict__ b, int k) {
  d2 acc0 = d2{0.0, 0.0}, acc1 = d2{0.0, 0.0}, acc2 = d2{0.0, 0.0}, acc3 = d2{0.0, 0.0};
  int a = 0;
  for (; a + 4 <= k; a += 4) {
    const d2 x0 = *reinterpret_cast<const d2 *>(X + (size_t)A[a] * ld + i);
    const d2 x1 = *reinterpret_cast<const d2 *>(X + (size_t)A[a + 1] * ld + i);
    const d2 x2 = *reinterpret_cast<const d2 *>(X + (size_t)A[a + 2] * ld + i);
    const d2 x3 = *reinterpret_cast<const d2 *>(X + (size_t)A[a + 3] * ld + i);
    acc0 += x0 * b[a];
    acc1 += x1 * b[a + 1];
    acc2 += x2 * b[a + 2];
    acc3 += x3 * b[a + 3];
  }
  for (; a < k; a++) acc0 += *reinterpret_cast<const d2 *>(X + (size_t)A[a] * ld + i) * b[a];
  return (acc0 + acc1) + (acc2 + acc3);
}

__device__ __forceinline__ void block_pair_sum_128(double a, double b, double *out2) {
  __shared__ double smp[2][2];
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    smp[threadIdx.x >> 6][0] = a;
    smp[threadIdx.x >> 6][1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out2[0] = smp[0][0] + smp[1][0];
    out2[1] = smp[0][1] + smp[1][1];
  }
}

// ------------------------------------------------------------------------------------------
// K6: Gram of the active panel on the fp64 matrix cores.
//   G[a][b] = sum_i w_i * c_a[i] * c_b[i]   for the mp = 16*mt "Gram columns" c_0..c_{mp-1}
// (pointers in colptr: columns of X, the all-ones / working-response vectors of the IRLS design,
// or a zero vector for padding).  Only tiles (I,J), J <= I, are formed.
//
// One wave = one task (tile row I, up to GRAM_JC tiles J0..J0+nJ-1, row slab s); waves are fully
// independent (no LDS, no barriers).  v_mfma_f64_16x16x4_f64 sums over 4 "k" rows per issue; which
// physical row a k-slot means is free as long as A and B agree, so lane (c = lane&15, q = lane>>4)
// loads the 4 consecutive rows row0+4q..row0+4q+3 of its column (32 contiguous bytes, 128 B per
// column per 16-row step) and feeds element t of that vector to the t-th of 4 MFMAs.
// C/D layout of the f64 MFMA: lane l, reg r holds D[row = (l>>4) + 4r][col = l&15]
// (cdna guide section 3: "f64 MFMA does NOT use the f32 maps").
// ------------------------------------------------------------------------------------------
// A Gram column is named by an int: idx >= 0 -> column idx of X; idx < 0 -> column (-idx-1) of the
// auxiliary matrix aux (same ld): aux column 0 = zeros (padding), 1 = ones on the data rows (intercept),
// 2 = the IRLS working response.  Both bases are kernel arguments, so the loads stay global_load.
__device__ __forceinline__ const double *gram_col(const double *__restrict__ X, const double *__restrict__ aux,
                                                  long ld, int idx) {
  return idx >= 0 ? X + (size_t)idx * ld : aux + (size_t)(-idx - 1) * ld;
}

// meta: [0] columns cached, [3] cache generation, [4] a dependent pair is cached (the request itself -- how many
// columns are missing -- lives in the fit's own control block, cov_nmiss: row sets that share their fills share meta).
// slot > 0: the request is the new active set of a PDAS iteration.  The kernel first does what k_gram_cols does in
// the streaming form (repeated active set -> same_prev), then looks the columns up.  If some are missing the fit is
// PARKED (cov_stall = 1, l = -1 - l: every gated kernel of this and the following slots falls through) and the host,
// which sees the flag in its next read-back, issues the fill and the rest of the slot.  slot == 0: start of a fit,
// the request is the initial support and the host has already queued a fill for it.
template <int NT>
__device__ void cov_need_body(const int *__restrict__ list, int len, const double *__restrict__ bd,
                              double *__restrict__ bd2, int p, int *__restrict__ slot_of, int *__restrict__ meta, int C,
                              int *__restrict__ fcols, FitCtrl *__restrict__ ctrl, int slot,
                              const int *__restrict__ A_cur, bool known_diff = false, bool no_restart = false) {
  __shared__ int wsum[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (slot > 0 && known_diff) {
    // (the caller has just built the list as A_cur plus one column: it differs, no need to compare)
    if (tid == 0) ctrl->same_prev = 0;
  } else if (slot > 0) {
    int diff = 1;
    if (ctrl->l >= 1 && ctrl->k_cur == len) {
      diff = 0;
      for (int i = tid; i < len; i += NT) diff |= (list[i] != A_cur[i]);
    }
    diff = __syncthreads_or(diff);
    if (tid == 0) ctrl->same_prev = diff ? 0 : 1;
    if (!diff) return;  // A == A_list.col(l-1): nothing to solve, nothing to look up
  }
  int count = meta[0];
  bool restart = count + len + COV_R > C;  // no room: start the cache over (uniform branch)
  if (restart) {
    // (that test counts the columns of the set that ARE cached as well; before the cache is given up, count the ones
    // really missing -- with a cache that holds every column of the design the answer is always "there is room")
    int missing = 0;
    for (int base = 0; base < len; base += NT) {
      const int i = base + tid;
      const int col = i < len ? list[i] : -1;
      missing += __syncthreads_count(col >= 0 && slot_of[col] < 0);
    }
    restart = count + missing + COV_R > C;
  }
  if (restart && no_restart && slot > 0) {
    // fold chains running side by side share the slot map: nobody rewrites it under the others.  Park the fit
    // (cov_stall = 4); the host starts the cache over when every chain is quiet (k_cov_fill_union)
    if (tid == 0) {
      ctrl->cov_nmiss = 0;
      ctrl->cov_stall = 4;
      ctrl->l = -1 - ctrl->l;
    }
    return;
  }
  if (restart) {
    for (int j = tid; j < p; j += NT) slot_of[j] = -1;
    count = 0;
    __syncthreads();
  }
  int nm = 0;
  if (!restart && len <= NT) {
    // the usual outcome, every column cached: one vote instead of the scan below
    const int col = tid < len ? list[tid] : -1;
    const int miss = (col >= 0 && slot_of[col] < 0) ? 1 : 0;
    if (!__syncthreads_or(miss)) {
      if (tid == 0) ctrl->cov_nmiss = 0;
      return;
    }
  }
  for (int base = 0; base < len; base += NT) {
    const int i = base + tid;
    const int col = i < len ? list[i] : -1;
    const int miss = (col >= 0 && slot_of[col] < 0) ? 1 : 0;
    int inc = miss;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
      off += (w < wave) ? wsum[w] : 0;
      tot += wsum[w];
    }
    if (miss) fcols[nm + off + inc - 1] = col;
    nm += tot;
    __syncthreads();
  }
  const bool spec = nm > 0 && bd != nullptr;
  if (tid == 0) {
    if (restart) {
      meta[0] = 0;  // (otherwise untouched: a background fill may be adding columns concurrently)
      meta[3] += 1;  // cache generation: slot numbers start over
      meta[4] = 0;   // no cached columns, no dependent pairs
    }
    ctrl->cov_nmiss = nm;  // (the request lives in the fit's own control block: row sets may share meta)
    if (nm > 0 && slot > 0) {
      ctrl->cov_stall = 1;
      ctrl->l = -1 - ctrl->l;
    }
  }
  if (spec) {
    for (int j = tid; j < p; j += NT) bd2[j] = slot_of[j] >= 0 ? -1.0 : bd[j];
    __syncthreads();
    for (int i = tid; i < nm; i += NT) bd2[fcols[i]] = -1.0;
  }
}

#define LAUNCH_CHECK()                         \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return e__;         \
  } while (0)

}  // namespace bessx
#endif  // BESSX_KDEV_HPP
