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

// ------------------------------------------------------------------------------------------
// upload: transpose a row-major chunk (rows x p) into column-major X[:, r0 : r0+rows]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_transpose_in(const double *__restrict__ src, int rows, int p,
                                                      double *__restrict__ X, long ld, long r0) {
  __shared__ double tile[64][65];
  int bj = blockIdx.x * 64, bi = blockIdx.y * 64;
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows of 64 per pass
  for (int r = ty; r < 64; r += 4) {
    int i = bi + r, j = bj + tx;
    tile[r][tx] = (i < rows && j < p) ? src[(size_t)i * p + j] : 0.0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    int j = bj + c, i = bi + tx;
    if (j < p && i < rows) X[(size_t)j * ld + r0 + i] = tile[tx][c];
  }
}

// ------------------------------------------------------------------------------------------
// K11: column statistics and rescale.  One 256-thread block per column.
//   mode 1/2 (data_type 1,2): mean_j = (w . x_j)/n, centre, norm_j = sqrt(w . x_j^2), x_j <- sqrt(n) x_j / norm_j
//   mode 3 (data_type 3): no centring.  Then (LM) every row is multiplied by sqrt(w_i).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_col_normalize(double *__restrict__ X, long ld, int n, int p,
                                                       const double *__restrict__ w, int centre, int do_scale,
                                                       int add_weight, double *__restrict__ x_mean,
                                                       double *__restrict__ x_norm) {
  __shared__ double sm[4];
  __shared__ double bc;
  int j = blockIdx.x;
  double *c = X + (size_t)j * ld;
  double mean = 0.0, nm = 1.0;
  if (do_scale) {
    if (centre) {
      double s = 0.0;
      for (int i = threadIdx.x; i < n; i += 256) s += w[i] * c[i];
      s = block_sum_256(s, sm);
      if (threadIdx.x == 0) bc = s / (double)n;
      __syncthreads();
      mean = bc;
      __syncthreads();
    }
    double s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
      double v = c[i] - mean;
      s2 += w[i] * (v * v);
    }
    s2 = block_sum_256(s2, sm);
    if (threadIdx.x == 0) bc = sqrt(s2);
    __syncthreads();
    nm = bc;
    if (threadIdx.x == 0) {
      x_mean[j] = mean;
      x_norm[j] = nm;
    }
  }
  double sn = sqrt((double)n);
  for (int i = threadIdx.x; i < n; i += 256) {
    double v = c[i];
    if (do_scale) v = sn * (v - mean) / nm;
    if (add_weight) v = v * sqrt(w[i]);
    c[i] = v;
  }
}

// y statistics for data_type 1: y_mean = (y . w)/n ; y <- y - y_mean ; (LM) y <- y*sqrt(w).  Single block.
__global__ void __launch_bounds__(256) k_y_prepare(double *__restrict__ y, int n, const double *__restrict__ w,
                                                   int centre, int add_weight, double *__restrict__ y_mean) {
  __shared__ double sm[4];
  __shared__ double bc;
  double mean = 0.0;
  if (centre) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += y[i] * w[i];
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) bc = s / (double)n;
    __syncthreads();
    mean = bc;
  }
  if (threadIdx.x == 0) *y_mean = mean;
  for (int i = threadIdx.x; i < n; i += 256) {
    double v = y[i] - mean;
    if (add_weight) v = v * sqrt(w[i]);
    y[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// K1 / K2: streaming X^T v.  The dominant kernel of every PDAS iteration (HBM bound).
//
// Work decomposition: a task = one wave x (row block rb of 128*U rows) x (CG consecutive columns).
// The wave keeps its 128*U-row slice of v in registers (2*U doubles per lane), streams the CG column
// slices with 16-byte loads straight into registers (no LDS: every byte of X is used exactly once,
// cdna guide "GEMV: load straight to VGPRs"), keeps one accumulator per column and folds the 64 lanes
// with a butterfly that halves the live accumulators at each step (CG + 2 shuffles instead of 6*CG).
// part[rb][j] receives the row-block partial; k_score_* adds the row blocks in fixed order.
// Gate: runs only while ctrl says the PDAS loop of this fit has not converged (see k_commit).
// ------------------------------------------------------------------------------------------
template <int U, int CG, bool TWO, bool NT = true>
__global__ void __launch_bounds__(256) k_xtv(const double *__restrict__ X, long ld, int p, int nrb,
                                             const double *__restrict__ v, const double *__restrict__ v2,
                                             double *__restrict__ part, double *__restrict__ part2,
                                             const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ncg = (p + CG - 1) / CG;
  const long cg = wid / nrb;
  const int rb = (int)(wid - cg * nrb);
  if (cg >= ncg) return;
  const long row0 = (long)rb * (128 * U) + lane * 2;
  d2 vr[U], vr2[TWO ? U : 1];
#pragma unroll
  for (int u = 0; u < U; u++) {
    vr[u] = *reinterpret_cast<const d2 *>(v + row0 + u * 128);
    if (TWO) vr2[u] = *reinterpret_cast<const d2 *>(v2 + row0 + u * 128);
  }
  double acc[CG], acc2[TWO ? CG : 1];
  const int j0 = (int)cg * CG;
#pragma unroll
  for (int c = 0; c < CG; c++) {
    int j = j0 + c;
    j = j < p ? j : p - 1;  // tail group: recompute the last column, result discarded below
    const double *col = X + (size_t)j * ld + row0;
    d2 xv[U];
#pragma unroll
    for (int u = 0; u < U; u++)
      xv[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2 *>(col + u * 128))
                 : *reinterpret_cast<const d2 *>(col + u * 128);
    double a = 0.0, a2 = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      a = fma(xv[u].x, vr[u].x, a);
      a = fma(xv[u].y, vr[u].y, a);
      if (TWO) {
        a2 = fma(xv[u].x * xv[u].x, vr2[u].x, a2);
        a2 = fma(xv[u].y * xv[u].y, vr2[u].y, a2);
      }
    }
    acc[c] = a;
    if (TWO) acc2[c] = a2;
  }
  // butterfly: after the step with offset o the lanes with (lane & o) == 0 keep the lower half of
  // the surviving columns.  CG is a power of two <= 16, so steps use offsets 32,16,8,4 (then 2,1 plain).
  int colbits = 0;
#pragma unroll
  for (int h = CG / 2, o = 32; h >= 1; h >>= 1, o >>= 1) {
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; i++) {
      double keep = up ? acc[i + h] : acc[i];
      double send = up ? acc[i] : acc[i + h];
      acc[i] = keep + __shfl_xor(send, o);
      if (TWO) {
        double keep2 = up ? acc2[i + h] : acc2[i];
        double send2 = up ? acc2[i] : acc2[i + h];
        acc2[i] = keep2 + __shfl_xor(send2, o);
      }
    }
    colbits += up ? h : 0;
  }
  constexpr int REM = 64 / CG;  // lanes still holding partials of the same column
  double s = acc[0], s2 = TWO ? acc2[0] : 0.0;
#pragma unroll
  for (int o = REM / 2; o >= 1; o >>= 1) {
    s += __shfl_xor(s, o);
    if (TWO) s2 += __shfl_xor(s2, o);
  }
  const int j = j0 + colbits;
  if ((lane & (REM - 1)) == 0 && j < p) {
    part[(size_t)rb * p + j] = s;
    if (TWO) part2[(size_t)rb * p + j] = s2;
  }
}

// ------------------------------------------------------------------------------------------
// sacrifice scores.  LM: src/Algorithm.h:1109-1126 with 1x1 Phi (src/utilities.cpp:142-151,167-177).
//   d_j = (sum_rb part[rb][j]) / n_t - 2 lambda beta_j ; phi_j = sqrt(2 lambda + xtx_j / n_t)
//   bd_j = (phi_j beta_j + d_j / phi_j)^2 ; always_select -> DBL_MAX
// GLM (logistic / Poisson), src/Algorithm.h:1236-1260, 1341-1364: d_j = s1 - 2 lambda beta_j,
//   phi_j = sqrt(s2 + 2 lambda).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_score(const double *__restrict__ part, const double *__restrict__ part2,
                                               int nrb, int p, const double *__restrict__ beta_dense,
                                               const double *__restrict__ xtx, double n_t, double lambda, int glm,
                                               const unsigned char *__restrict__ always, double *__restrict__ bd,
                                               const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  // 64 columns per block; the 4 thread groups each add every 4th row block, then the 4 group sums
  // are added in group order (fixed tree).
  __shared__ double sm1[4][65], sm2[4][65];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (j < p)
    for (int rb = g; rb < nrb; rb += 4) {
      s1 += part[(size_t)rb * p + j];
      if (glm) s2 += part2[(size_t)rb * p + j];
    }
  sm1[g][cl] = s1;
  sm2[g][cl] = s2;
  __syncthreads();
  if (g != 0 || j >= p) return;
  s1 = ((sm1[0][cl] + sm1[1][cl]) + sm1[2][cl]) + sm1[3][cl];
  s2 = ((sm2[0][cl] + sm2[1][cl]) + sm2[2][cl]) + sm2[3][cl];
  double b = beta_dense[j], d, phi;
  if (glm) {
    d = s1 - 2.0 * lambda * b;
    phi = sqrt(s2 + 2.0 * lambda);
  } else {
    d = s1 / n_t - 2.0 * lambda * b;
    phi = sqrt(2.0 * lambda + xtx[j] / n_t);
  }
  double inv = 1.0 / phi;
  double t = phi * b + inv * d;
  double v = t * t;
  if (always != nullptr && always[j]) v = DBL_MAX;
  bd[j] = v;
}

// ------------------------------------------------------------------------------------------
// K4: max_k (src/utilities.cpp:179-188).  One 1024-thread workgroup selects the k largest of
// len <= 32768 non-negative doubles (ties -> lower index) and writes their indices ascending.
// Keys are the raw bit patterns (monotone for non-negative doubles; NaN sorts above +inf).
// The threshold (k-th largest key) is found bit by bit with ballot/popcount counting, the
// selection is a block-wide ordered compaction.  idx_in (optional) gives the original index of
// every element (second level of the two-level selection for len > 32768).
// ------------------------------------------------------------------------------------------
constexpr int TOPK_E = 32;  // keys per thread

__device__ __forceinline__ unsigned long long score_key(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? 0ull : b;  // -0.0 / negative (never produced) -> smallest
}

template <int NT>
__device__ void cov_need_body(const int *__restrict__ list, int len, const double *__restrict__ bd,
                              double *__restrict__ bd2, int p, int *__restrict__ slot_of, int *__restrict__ meta, int C,
                              int *__restrict__ fcols, FitCtrl *__restrict__ ctrl, int slot,
                              const int *__restrict__ A_cur, bool known_diff = false, bool no_restart = false);

__device__ __forceinline__ void commit_body(FitCtrl *__restrict__ ctrl, int slot, int T0,
                                            const int *__restrict__ A_new, const double *__restrict__ sol,
                                            int has_intercept, int wait_chain, int *__restrict__ A_cur,
                                            double *__restrict__ b_cur, double *__restrict__ beta_dense,
                                            int *__restrict__ hist, double *__restrict__ hist_beta,
                                            double *__restrict__ hist_coef0, int hist_stride, int *same_any_sh,
                                            unsigned char *__restrict__ inA);

// The repeated-set shortcut of the covariance form (the caller has seen ctrl->fast_same): k_cov_d left, per block of 32
// columns, the smallest score inside the current active set and the largest outside it.  If every inside score beats
// every outside score (and the set has the wanted size, and this is not the first iteration of the fit) the selection
// returns the same set, A == A_list.col(l-1): nothing to search, nothing to look up -- the iteration is recorded
// (the record-and-stop branch of k_commit) and, when this launch ends a chained batch, the result block snapshotted.
// Returns true when the slot is settled.
template <int NT>
__device__ __forceinline__ bool repeated_set_body(const TopkNeed &nd, int k, int *__restrict__ out, int slot) {
  constexpr int NWV = NT / 64;
#ifdef BESSX_KTRACE
  unsigned long long tph_ = wall_clock64();  // (stamps g_phase[16..20], count [28])
#endif
  PH(16);
  __shared__ double rmn[NWV], rmx[NWV];
  double mn = DBL_MAX, mx = -1.0;
  for (int b = threadIdx.x; b < nd.nbmm; b += NT) {
    mn = fmin(mn, nd.bmm[2 * b]);
    mx = fmax(mx, nd.bmm[2 * b + 1]);
  }
  wave_min_max(mn, mx);
  PH(17);
  if ((threadIdx.x & 63) == 0) {
    rmn[threadIdx.x >> 6] = mn;
    rmx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  mn = rmn[0];
  mx = rmx[0];
#pragma unroll
  for (int w = 1; w < NWV; w++) {
    mn = fmin(mn, rmn[w]);
    mx = fmax(mx, rmx[w]);
  }
  const bool same = nd.ctrl->l >= 1 && nd.ctrl->k_cur == k && mn > mx;  // uniform
  __syncthreads();
  if (threadIdx.x == 0) {
    nd.ctrl->fast_same = 0;
    if (same) nd.ctrl->same_prev = 1;
  }
  PH(18);
  if (!same) return false;
  if (nd.commit_on) {
    // (otherwise the solve kernel queued behind this one records the iteration)
    __shared__ int same_any_sh;
    __syncthreads();
    commit_body(nd.ctrl, slot, k, out, nullptr, 0, 0, nd.cm_A_cur, nd.cm_b_cur, nd.cm_beta_dense, nd.cm_hist,
                nd.cm_hist_beta, nd.cm_hist_coef0, nd.cm_hist_stride, &same_any_sh, nd.cm_inA);
    PH(19);
    if (nd.snap.on == 2) {  // the fit has ended: its snapshot for the deferred publication, here and now
      if (threadIdx.x == 0) nd.ctrl->snap_seq = nd.snap.seq;
      __syncthreads();
      snapshot_body(nd.snap);
    }
    PH(20);
#ifdef BESSX_KTRACE
    if (threadIdx.x == 0) atomicAdd(&g_phase[28], 1ull);
#endif
  }
  return true;
}

// nd.slot_of != nullptr (covariance form of the LM fit, single chunk): the kernel ends with the work of k_cov_need on
// the indices it has just selected -- one launch less per PDAS iteration.
// The selection as a device function of an NT-thread block (NT threads = 16 waves: k_topk; 8 waves: the first phase
// of k_sel_cgr, which goes on to solve the selected system in the same launch).  EB = keys per thread this instance
// can hold (bucket of ceil(len / NT)).
template <int EB, int NT>
__device__ __forceinline__ void topk_body(const double *__restrict__ score, const int *__restrict__ idx_in,
                                          int len_total, int chunk, int k, int *__restrict__ out,
                                          int *__restrict__ out_count, const FitCtrl *ctrl, int slot,
                                          const int *__restrict__ run_flag, const TopkNeed &nd) {
  constexpr int NWV = NT / 64;
  if (nd.cont_on) {
    // k_fit_continue(chained) as the prologue of the first kernel of the chained fit: it only starts if the fit
    // before it (serial cont_parent) ended here on a repeated set with fresh score sums
    FitCtrl *c = nd.ctrl;
    const bool go = c->serial == nd.cont_parent && c->done && c->d_fresh && c->l >= 0 && !c->cov_stall && !c->info;
    if (!go) return;  // uniform
    for (int i = threadIdx.x; i < k; i += NT) nd.cm_hist[i] = 0;
    __syncthreads();  // every thread has read the old block
    if (threadIdx.x == 0) {
      c->done = 0;
      c->l = 0;
      c->T0 = k;
      c->irls_done = 0;
      c->irls_steps = 0;
      c->info = 0;
      c->same_prev = 0;
      c->d_fresh = 0;
      c->cov_nfill = 0;
      c->cov_stall = 0;
      c->cov_groups = 0;
      c->cov_miss = 0;
      c->cov_nmiss = 0;
      c->sse_valid = 0;
      c->fast_same = 0;
      c->serial = nd.cont_serial;
    }
    __syncthreads();
  } else if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) {
    return;
  }
  if (run_flag != nullptr && *run_flag == 0) return;
  __shared__ int wcnt[2][NWV];
  if (nd.slot_of != nullptr && nd.inc1 && nd.ctrl->l == 0 && nd.ctrl->k_cur + 1 == k &&
      (gridDim.x == 1 || nd.pub.on)) {
    // First iteration of a fit chained behind a fit of size k-1 whose last iteration confirmed A_cur = max_k(bd, k-1)
    // on exactly these scores: max_k(bd, k) is A_cur plus the best score outside it (ties -> lower index, the same
    // total order).  An arg-max instead of a selection.
    __shared__ unsigned long long bk[NWV];
    __shared__ int bi[NWV];
    PH_BEGIN();
    // this thread's element of the old active set, its cache slot and the cache's column count: loads that do not
    // depend on the arg-max, in flight while it runs (used when the old set fits one round of the block)
    const bool one_round = k - 1 <= NT;
    const int a_mine = (one_round && (int)threadIdx.x < k - 1) ? nd.A_cur[threadIdx.x] : -1;
    const int sl_mine = a_mine >= 0 ? nd.slot_of[a_mine] : 0;
    const int cnt_cache = one_round ? nd.meta[0] : 0;
    unsigned long long best = 0ull;
    int besti = 0x7fffffff;
    bool have = false;
    int dupl = 0;  // this thread has met its maximum more than once
    auto feed = [&](unsigned long long key, int i) {
      if (!have || key > best) {
        best = key;
        besti = i;
        have = true;
        dupl = 0;
      } else if (key == best) {
        dupl = 1;
        if (i < besti) besti = i;
      }
    };
    if (nd.bmm != nullptr && nd.bmm_fresh) {
      // the k_cov_d that produced these scores left, per block of 32 columns, the largest score outside the active
      // set and its column: the arg-max over p scores is the arg-max over p / 32 block maxima (same total order)
      for (int b = threadIdx.x; b < nd.nbmm; b += NT) {
        const double v = nd.bmm[2 * b + 1];
        if (v >= 0.0) feed(score_key(v), (int)nd.bmm[2 * nd.nbmm + b]);
      }
    } else {
      // all loads first (independent), then the comparisons: a load behind a branch per element serialises
      double sc[EB];
      unsigned char ia[EB];
#pragma unroll
      for (int e = 0; e < EB; e++) {
        const int i = threadIdx.x + e * NT;
        const bool in = i < len_total;
        sc[e] = in ? score[i] : 0.0;
        ia[e] = in ? nd.inA[i] : (unsigned char)1;
      }
#pragma unroll
      for (int e = 0; e < EB; e++) {
        const int i = threadIdx.x + e * NT;
        if (!ia[e]) feed(score_key(sc[e]), i);
      }
    }
    PH(0);
    const unsigned long long mybest = best;  // this thread's own maximum, before the reductions
    const int mybesti = besti;
    wave_argmax(best, besti);
    if ((threadIdx.x & 63) == 0) {
      bk[threadIdx.x >> 6] = best;
      bi[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    best = bk[0];
    besti = bi[0];
#pragma unroll
    for (int w = 1; w < NWV; w++)
      if (bk[w] > best || (bk[w] == best && bi[w] < besti)) {
        best = bk[w];
        besti = bi[w];
      }
    {
      // Is the maximum outside the set UNIQUE?  If a second column ties with it, the k-th and the (k + 1)-th largest
      // score are equal and the arg-max (lower index) is not what the reference's nth_element returns: leave the
      // shortcut, the full search below finds the tie and parks the fit for the exact selection.  A thread holds the
      // maxima of the entries it looked at: a tie shows as another thread's (or another entry's) equal key; from the
      // block maxima of k_cov_d only one entry per 32 columns is seen, so the winner's own 32-column block is read too.
      int dup = (have && mybest == best && (mybesti != besti || dupl)) ? 1 : 0;
      if (!dup && threadIdx.x < 32) {
        const int j = (besti & ~31) + (int)threadIdx.x;
        if (j < len_total && j != besti && !nd.inA[j] && score_key(score[j]) == best) dup = 1;
      }
      if (__syncthreads_or(dup)) goto full_search;  // uniform
    }
    PH(1);
    // ordered insertion of besti into the sorted A_cur; its position = number of smaller elements, counted by the
    // block (a binary search by one thread is a chain of dependent global loads)
    const int *A_old = nd.A_cur;
    int lo = 0;
    if (one_round) {
      // the list goes out and the cache lookup is answered from registers: no re-read of what was just stored
      const int sl_new = nd.slot_of[besti];
      int smaller = 0;
      if (a_mine >= 0) {
        out[threadIdx.x + (a_mine > besti ? 1 : 0)] = a_mine;
        smaller = a_mine < besti ? 1 : 0;
      }
      lo = __syncthreads_count(smaller);
      const int miss = ((a_mine >= 0 && sl_mine < 0) || sl_new < 0) ? 1 : 0;
      const bool any_miss = __syncthreads_or(miss) != 0;
      if (threadIdx.x == 0) {
        out[lo] = besti;
        nd.ctrl->fast_same = 0;
      }
      PH(2);
      if (!any_miss && !(cnt_cache + k + COV_R > nd.C) && slot > 0) {
        // every column cached, no restart: what cov_need_body would conclude (the set differs from A_cur by construction)
        if (threadIdx.x == 0) {
          nd.ctrl->same_prev = 0;
          nd.ctrl->cov_nmiss = 0;
        }
        __syncthreads();
        PH(3);
        PH_COUNT();
        return;
      }
      __syncthreads();
      cov_need_body<NT>(out, k, nd.bd, nd.bd2, nd.p, nd.slot_of, nd.meta, nd.C, nd.fcols, nd.ctrl, slot, nd.A_cur, true,
                      nd.no_restart != 0);
      PH(3);
      PH_COUNT();
      return;
    }
    for (int base = 0; base < k - 1; base += NT) {  // uniform trip count
      const int i = base + threadIdx.x;
      int smaller = 0;
      if (i < k - 1) {
        const int a = A_old[i];
        out[i + (a > besti ? 1 : 0)] = a;
        smaller = a < besti ? 1 : 0;
      }
      lo += __syncthreads_count(smaller);
    }
    if (threadIdx.x == 0) {
      out[lo] = besti;
      nd.ctrl->fast_same = 0;
    }
    __syncthreads();
    PH(2);
    cov_need_body<NT>(out, k, nd.bd, nd.bd2, nd.p, nd.slot_of, nd.meta, nd.C, nd.fcols, nd.ctrl, slot, nd.A_cur, true,
                      nd.no_restart != 0);
    PH(3);
    PH_COUNT();
    return;
  }
full_search:
  if (nd.slot_of != nullptr && nd.ctrl->fast_same) {
    if (repeated_set_body<NT>(nd, k, out, slot)) return;
  }
  __shared__ int wsum[NWV];
  __shared__ int wsum2[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * chunk;
  const int len = min(chunk, len_total - base);
  const int kk = min(k, len);
  const int E = (len + NT - 1) / NT;  // <= EB
  const int e0 = tid * E;
  unsigned long long key[EB];
#pragma unroll
  for (int e = 0; e < EB; e++) {
    int i = e0 + e;
    bool ok = e < E && i < len;
    int src = ok ? (idx_in ? idx_in[base + i] : base + i) : 0;
    key[e] = ok ? score_key(score[src]) : 0ull;
  }
  // Threshold search: build T bit by bit, keeping count(key >= T) >= kk.  As soon as the count is
  // EXACTLY kk the set {key >= T} is the answer and the remaining bits need not be resolved (for
  // continuous scores that happens ~log2(len) bits below the leading bit).  Padding keys are 0 and
  // never count because a candidate is always >= 1.
  unsigned long long T = 0ull;
  int par = 0;
  for (int bit = 62; bit >= 0; bit--) {
    const unsigned long long cand = T | (1ull << bit);
    int c = 0;
#pragma unroll
    for (int e = 0; e < EB; e++) c += __popcll(__ballot(key[e] >= cand));
    if (lane == 0) wcnt[par][wave] = c;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) tot += wcnt[par][w];
    if (tot >= kk) T = cand;
    par ^= 1;
    if (tot == kk) break;  // uniform: every thread computed the same tot
  }
  // per-thread counts of keys > T and == T (valid elements only; T == 0 means "everything ties at 0")
  int ngt = 0, neq = 0;
#pragma unroll
  for (int e = 0; e < EB; e++) {
    bool ok = e < E && (e0 + e) < len;
    ngt += (ok && key[e] > T) ? 1 : 0;
    neq += (ok && key[e] == T) ? 1 : 0;
  }
  // exclusive block scans of neq and (later) of the selected count
  auto block_excl_scan = [&](int v, int *ws, int &total) -> int {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
      int sv = ws[w];
      off += (w < wave) ? sv : 0;
      tot += sv;
    }
    total = tot;
    __syncthreads();
    return off + inc - v;
  };
  int tot_eq, tot_gt;
  (void)block_excl_scan(ngt, wsum, tot_gt);
  const int need_eq = kk - tot_gt;  // how many ties to take, lowest indices first
  int eq_before = block_excl_scan(neq, wsum2, tot_eq);
  if (nd.slot_of != nullptr && tot_eq > need_eq && kk < len) {
    // Covariance form (this launch goes on to the cache lookup and, in k_sel_cgr, to the solve): the k-th and the
    // (k + 1)-th largest score are equal, so the set is the one std::nth_element's moves leave (k_topk_ties), not the
    // lower indices.  Park the fit (cov_stall = 3: every queued kernel falls through); the host redoes this slot's
    // selection exactly and issues the rest of the slot.
    if (tid == 0) {
      nd.ctrl->cov_stall = 3;
      nd.ctrl->l = -1 - nd.ctrl->l;
      nd.ctrl->fast_same = 0;
    }
    return;  // uniform
  }
  int take_eq = min(max(need_eq - eq_before, 0), neq);
  int nsel = ngt + take_eq, tot_sel;
  int pos = block_excl_scan(nsel, wsum, tot_sel);
  int *o = out + (size_t)blockIdx.x * k;
  int eq_seen = 0;
#pragma unroll
  for (int e = 0; e < EB; e++) {
    bool ok = e < E && (e0 + e) < len;
    bool sel = ok && key[e] > T;
    if (ok && key[e] == T) {
      sel = eq_seen < take_eq;
      eq_seen++;
    }
    if (sel) {
      o[pos] = idx_in ? idx_in[base + e0 + e] : base + e0 + e;
      pos++;
    }
  }
  // more keys equal to the threshold than the selection can take: the k-th and the (k + 1)-th largest score are EQUAL.
  // The reference's std::nth_element then keeps whichever of the tied indices its partition steps leave in front
  // (implementation-defined, not the lower index taken above): flag it, k_topk_ties redoes the selection move by move
  if (tid == 0 && out_count != nullptr && tot_eq > need_eq && kk < len) out_count[0] = 1;
  if (nd.slot_of != nullptr) {
    __syncthreads();  // the selected indices are visible to the whole block
    cov_need_body<NT>(out, k, nd.bd, nd.bd2, nd.p, nd.slot_of, nd.meta, nd.C, nd.fcols, nd.ctrl, slot, nd.A_cur, false,
                      nd.no_restart != 0);
  }
}

template <int EB>  // keys per thread this instance can hold (bucket of ceil(len / 1024))
__global__ void __launch_bounds__(1024) k_topk(const double *__restrict__ score, const int *__restrict__ idx_in,
                                               int len_total, int chunk, int k, int *__restrict__ out,
                                               int *__restrict__ out_count, const FitCtrl *ctrl, int slot,
                                               const int *__restrict__ run_flag, const TopkNeed nd) {
  KT(1);
  if (nd.pub.on && blockIdx.x == 1) {  // second workgroup: publishes the parent fit's snapshot, nothing else
    publish_body(nd.pub);
    return;
  }
  topk_body<EB, 1024>(score, idx_in, len_total, chunk, k, out, out_count, ctrl, slot, run_flag, nd);
}

// max_k when the k-th and the (k + 1)-th largest score are equal (duplicated columns, 0/1 designs): the reference's
// std::nth_element (libstdc++, GCC 11: __introselect = median-of-3 pivot to the front, Hoare-style unguarded partition,
// insertion sort of the last <= 3) decides which tied indices land in the first k positions, so its moves are redone
// here on the index array 0 .. len-1, comparator comp(i, j) = score[i] > score[j] (src/utilities.cpp:179-188).  One
// 1024-thread block; the partition of a range, sequential in the library, is done in parallel from its definition:
// the scan from the left stops at the positions whose element is NOT greater than the pivot (in ascending order
// L_0 < L_1 < ...), the scan from the right at those whose element is NOT smaller (descending R_0 > R_1 > ...), the
// t-th exchange swaps positions L_t and R_t while L_t < R_t -- up to there neither scan has met a position written by
// an earlier exchange, so both lists are read off the unmodified range -- and the partition returns where the left scan
// stands once the scans have met: min(L_t, R_{t-1}) (the last exchange left a stop at R_{t-1}).  work = 3 len ints (index array, L list / selection flags, R list).  Runs only when the selection kernel
// has raised flag[0]; clears it.  The heap-select branch of the introselect (depth limit reached) is done by one thread.
__global__ void __launch_bounds__(1024) k_topk_ties(const double *__restrict__ score, int len, int k,
                                                    int *__restrict__ out, int *__restrict__ flag,
                                                    int *__restrict__ work, const FitCtrl *ctrl, int slot,
                                                    const int *__restrict__ run_flag) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  if (run_flag != nullptr && *run_flag == 0) return;
  if (flag[0] == 0) return;
  constexpr int NT = 1024, NWV = NT / 64;
  __shared__ int wsum[NWV];
  __shared__ int sh_first, sh_last;
  __shared__ double sh_piv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *idx = work, *Lp = work + len, *Rp = work + 2 * (size_t)len;
  for (int i = tid; i < len; i += NT) idx[i] = i;
  if (tid == 0) {
    sh_first = 0;
    sh_last = len;
  }
  __syncthreads();
  // exclusive rank of this thread's flag among the block's flags (thread order), and the block total
  auto block_rank = [&](bool f, int &total) -> int {
    const unsigned long long m = __ballot(f);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
      off += w < wave ? wsum[w] : 0;
      tot += wsum[w];
    }
    total = tot;
    __syncthreads();
    return off + before;
  };
  int depth = 0;
  while ((len >> (depth + 1)) > 0) depth++;  // std::__lg(len)
  depth *= 2;
  const int nth = k;
  while (true) {
    const int first = sh_first, last = sh_last;
    if (last - first <= 3) break;
    if (depth == 0) {
      // depth limit exhausted: std::__heap_select(first, nth + 1, last) and iter_swap(first, nth), move by move (one
      // thread: rare, and the range is what 2 floor(log2 len) partitions have left)
      if (tid == 0) {
        int *f = idx + first;
        const long hl = (long)nth + 1 - first;
        auto adjust = [&](long hole, int value) {  // std::__adjust_heap incl. __push_heap
          const long top = hole;
          long child = hole;
          while (child < (hl - 1) / 2) {
            child = 2 * (child + 1);
            if (score[f[child]] > score[f[child - 1]]) child--;
            f[hole] = f[child];
            hole = child;
          }
          if ((hl & 1) == 0 && child == (hl - 2) / 2) {
            child = 2 * (child + 1);
            f[hole] = f[child - 1];
            hole = child - 1;
          }
          long parent = (hole - 1) / 2;
          while (hole > top && score[f[parent]] > score[value]) {
            f[hole] = f[parent];
            hole = parent;
            parent = (hole - 1) / 2;
          }
          f[hole] = value;
        };
        if (hl >= 2)
          for (long parent = (hl - 2) / 2;; parent--) {  // std::__make_heap
            adjust(parent, f[parent]);
            if (parent == 0) break;
          }
        for (int i = nth + 1; i < last; i++)
          if (score[idx[i]] > score[f[0]]) {  // std::__pop_heap(first, middle, i)
            const int value = idx[i];
            idx[i] = f[0];
            adjust(0, value);
          }
        const int t = idx[first];
        idx[first] = idx[nth];
        idx[nth] = t;
        sh_first = sh_last = first;  // (nothing left for the insertion sort)
      }
      __syncthreads();
      break;
    }
    depth--;
    if (tid == 0) {
      // __move_median_to_first(first, first + 1, mid, last - 1)
      const int mid = first + (last - first) / 2;
      int *r = idx + first, *a = idx + first + 1, *b = idx + mid, *c = idx + last - 1;
      const double sa = score[*a], sb = score[*b], sc = score[*c];
      int *pick;
      if (sa > sb) pick = (sb > sc) ? b : ((sa > sc) ? c : a);
      else pick = (sa > sc) ? a : ((sb > sc) ? c : b);
      const int t = *r;
      *r = *pick;
      *pick = t;
      sh_piv = score[*r];
    }
    __syncthreads();
    const double sp = sh_piv;
    const int lo = first + 1, hi = last;  // __unguarded_partition(first + 1, last, pivot = first)
    int totL = 0, totR = 0;
    for (int base = lo; base < hi; base += NT) {  // stops of the scan from the left, ascending
      const int i = base + tid;
      const bool f = i < hi && !(score[idx[i]] > sp);
      int tot;
      const int r = block_rank(f, tot);
      if (f) Lp[totL + r] = i;
      totL += tot;
    }
    for (int base = hi - 1; base >= lo; base -= NT) {  // stops of the scan from the right, descending
      const int i = base - tid;
      const bool f = i >= lo && !(sp > score[idx[i]]);
      int tot;
      const int r = block_rank(f, tot);
      if (f) Rp[totR + r] = i;
      totR += tot;
    }
    __syncthreads();
    // number of exchanges = number of t with L_t < R_t (the predicate is monotone in t)
    const int tm = min(totL, totR);
    int cnt = 0;
    for (int base = 0; base < tm; base += NT) {
      const int t = base + tid;
      int tot;
      (void)block_rank(t < tm && Lp[t] < Rp[t], tot);
      cnt += tot;
    }
    // where the scan from the left stands when the scans have met: at its next stop of the unmodified range, or at
    // the position of the last exchange (which now holds an element that is not greater than the pivot), whichever
    // comes first
    int cut = cnt < totL ? Lp[cnt] : 0x7fffffff;
    if (cnt >= 1) cut = min(cut, Rp[cnt - 1]);
    if (cut == 0x7fffffff) {  // cannot happen: the median-of-3 pivot guards the scan
      if (tid == 0) {
        flag[1] = 1;
        flag[0] = 0;
      }
      return;
    }
    __syncthreads();
    for (int t = tid; t < cnt; t += NT) {
      const int a = Lp[t], b = Rp[t], va = idx[a], vb = idx[b];
      idx[a] = vb;
      idx[b] = va;
    }
    if (tid == 0) {
      if (cut <= nth) sh_first = cut;
      else sh_last = cut;
    }
    __syncthreads();
  }
  if (tid == 0) {
    // __insertion_sort of the last <= 3 elements
    const int first = sh_first, last = sh_last;
    for (int i = first + 1; i < last; i++) {
      const int val = idx[i];
      const double sv = score[val];
      int j = i;
      if (sv > score[idx[first]]) {
        for (; j > first; j--) idx[j] = idx[j - 1];
        idx[first] = val;
      } else {
        while (sv > score[idx[j - 1]]) {
          idx[j] = idx[j - 1];
          j--;
        }
        idx[j] = val;
      }
    }
  }
  __syncthreads();
  // std::sort(ind, ind + k): the selected indices ascending -- membership flags, then an ordered compaction
  for (int i = tid; i < len; i += NT) Lp[i] = 0;
  __syncthreads();
  for (int i = tid; i < k; i += NT) Lp[idx[i]] = 1;
  __syncthreads();
  int done = 0;
  for (int base = 0; base < len; base += NT) {
    const int i = base + tid;
    const bool f = i < len && Lp[i] != 0;
    int tot;
    const int r = block_rank(f, tot);
    if (f) out[done + r] = i;
    done += tot;
  }
  if (tid == 0) flag[0] = 0;
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

template <int NJ, bool WEIGHTED>
__device__ __forceinline__ void gram_body(const double *__restrict__ X, const double *__restrict__ aux, long ld,
                                          const int *__restrict__ cols, const double *__restrict__ w,
                                          const GramTask tk, long r_begin, long r_end, double *__restrict__ out,
                                          int tile_base) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, q = lane >> 4;
  const double *pa = gram_col(X, aux, ld, cols[tk.I * 16 + c]) + 4 * q;
  const double *pb[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; jj++) pb[jj] = gram_col(X, aux, ld, cols[(tk.J0 + jj) * 16 + c]) + 4 * q;
  d4 acc[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; jj++) acc[jj] = d4{0.0, 0.0, 0.0, 0.0};
  // Software pipeline: the loads of the next 16 rows are issued before the MFMAs of the current ones.  A slab is a
  // few hundred rows and a wave often has its SIMD to itself, so nothing else hides the load latency.
  d2 a0, a1, w0 = d2{1.0, 1.0}, w1 = d2{1.0, 1.0}, b0[NJ], b1[NJ];
  auto load = [&](long r, d2 &xa0, d2 &xa1, d2 &xw0, d2 &xw1, d2 (&xb0)[NJ], d2 (&xb1)[NJ]) {
    xa0 = *reinterpret_cast<const d2 *>(pa + r);
    xa1 = *reinterpret_cast<const d2 *>(pa + r + 2);
    if (WEIGHTED) {
      xw0 = *reinterpret_cast<const d2 *>(w + r + 4 * q);
      xw1 = *reinterpret_cast<const d2 *>(w + r + 4 * q + 2);
    }
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
      xb0[jj] = *reinterpret_cast<const d2 *>(pb[jj] + r);
      xb1[jj] = *reinterpret_cast<const d2 *>(pb[jj] + r + 2);
    }
  };
  if (r_begin < r_end) load(r_begin, a0, a1, w0, w1, b0, b1);
  for (long r = r_begin; r < r_end; r += 16) {
    d2 na0 = a0, na1 = a1, nw0 = w0, nw1 = w1, nb0[NJ], nb1[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
      nb0[jj] = b0[jj];
      nb1[jj] = b1[jj];
    }
    if (r + 16 < r_end) load(r + 16, na0, na1, nw0, nw1, nb0, nb1);
    double ax = a0.x, ay = a0.y, az = a1.x, aw = a1.y;
    if (WEIGHTED) {
      ax *= w0.x;
      ay *= w0.y;
      az *= w1.x;
      aw *= w1.y;
    }
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
      acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b0[jj].x, acc[jj], 0, 0, 0);
      acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b0[jj].y, acc[jj], 0, 0, 0);
      acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(az, b1[jj].x, acc[jj], 0, 0, 0);
      acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, b1[jj].y, acc[jj], 0, 0, 0);
    }
    a0 = na0;
    a1 = na1;
    w0 = nw0;
    w1 = nw1;
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
      b0[jj] = nb0[jj];
      b1[jj] = nb1[jj];
    }
  }
#pragma unroll
  for (int jj = 0; jj < NJ; jj++) {
    const int t = tk.I * (tk.I + 1) / 2 + tk.J0 + jj - tile_base;
    *reinterpret_cast<d4 *>(out + (size_t)t * 256 + lane * 4) = acc[jj];
  }
}

template <bool WEIGHTED>
__global__ void __launch_bounds__(256) k_gram(const double *__restrict__ X, const double *__restrict__ aux, long ld,
                                              const int *__restrict__ cols, const double *__restrict__ w,
                                              int rows_per_slab, const GramTask *__restrict__ tasks, int ntask,
                                              int nslab, double *__restrict__ part, int ntiles,
                                              const FitCtrl *__restrict__ ctrl, int slot, int gate_mode,
                                              int tile_base) {
  if (ctrl != nullptr) {
    if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
    if ((gate_mode == 1 || gate_mode == 2) && ctrl->irls_done) return;
    if (gate_mode == 3 && !ctrl->gram_full) return;  // LM: whole Gram only when the cache cannot be used
    if (gate_mode == 4 && ctrl->gram_full) return;   // LM: new rows only
  }
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int slab = (int)(wid / ntask);
  if (slab >= nslab) return;
  const GramTask tk = tasks[wid - (long)slab * ntask];
  const long r_begin = (long)slab * rows_per_slab;
  const long r_end = min(r_begin + rows_per_slab, ld);
  double *out = part + (size_t)slab * ntiles * 256;
  switch (tk.nJ) {  // wave-uniform: tasks are cut into runs of 8, 4, 2 or 1 tiles of one tile row
    case 8: gram_body<8, WEIGHTED>(X, aux, ld, cols, w, tk, r_begin, r_end, out, tile_base); break;
    case 4: gram_body<4, WEIGHTED>(X, aux, ld, cols, w, tk, r_begin, r_end, out, tile_base); break;
    case 2: gram_body<2, WEIGHTED>(X, aux, ld, cols, w, tk, r_begin, r_end, out, tile_base); break;
    default: gram_body<1, WEIGHTED>(X, aux, ld, cols, w, tk, r_begin, r_end, out, tile_base); break;
  }
}

// Sum the row-slab partials: Gt[t][e] = sum_s part[s][t][e].  A block owns 16 consecutive elements; its 16 thread
// groups each add every 16th slab, then the 16 group sums are added in group order (fixed tree).
// K6, LDS-staged form (full lower triangle, mt <= 16).  k_gram's waves each read "their" tile row plus a run of
// other tile columns straight from memory, so a column tile of X_A is fetched once per task that touches it:
// (tasks + tiles) / mt times, 41 / 7 at k = 100 -- from MALL / HBM, since X_A (90 MB at n = 100k, k = 100) is far
// beyond L2.  Here one block owns a row slab and ALL tiles: 64 rows of every active column are staged once into LDS
// (coalesced 16-byte loads, next chunk in flight while the current one is multiplied), the waves read their MFMA
// operands from LDS.  Traffic = the active columns once; the rest is MFMA time.  Same output layout as k_gram.
__device__ __forceinline__ void tile_of(int t, int &I, int &J);
template <int NW, int TPW, int NPASS, int GL_RB, bool WEIGHTED>
__global__ void __launch_bounds__(64 * NW) k_gram_lds(const double *__restrict__ X, const double *__restrict__ aux,
                                                      long ld, const int *__restrict__ cols,
                                                      const double *__restrict__ w, int rows_per_slab, int nslab,
                                                      int mt, double *__restrict__ part, int ntiles,
                                                      const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  if (ctrl != nullptr) {
    if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
    if ((gate_mode == 1 || gate_mode == 2) && ctrl->irls_done) return;
    if (gate_mode == 3 && !ctrl->gram_full) return;
    if (gate_mode == 4 && ctrl->gram_full) return;
  }
  extern __shared__ double smem[];  // [mp][GL_LD], then the weights of the chunk
  constexpr int GL_LD = GL_RB + 2, TPC = GL_RB / 2;  // padded column stride; threads per column (a row pair each)
  constexpr int NT = 64 * NW, CPP = NT / TPC;         // columns staged per pass
  const int mp = mt * 16;
  double *wch = smem + (size_t)mp * GL_LD;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ru = tid % TPC, cb = tid / TPC;
  const int slab = blockIdx.x;
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int nchunk = (int)((r_end - r_begin + GL_RB - 1) / GL_RB);
  // column pointers: kept in registers, except in the largest instance where the accumulators need them (there the
  // column index is re-read per chunk: one cached load against 64 rows of MFMA work)
  constexpr bool PTRS = TPW <= 10;
  const double *src[PTRS ? NPASS : 1];
  if (PTRS) {
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int col = i * CPP + cb;
      src[i] = gram_col(X, aux, ld, col < mp ? cols[col] : cols[0]) + 2 * ru;
    }
  }
  d2 st[NPASS], wst = d2{0.0, 0.0};
  auto load = [&](long r0) {
    const bool in = r0 + 2 * ru < r_end;  // slabs end on multiples of 16 rows: a row pair is in or out as a whole
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      st[i] = d2{0.0, 0.0};
      if (in && i * CPP + cb < mp) {
        const double *q_ = PTRS ? src[i] : gram_col(X, aux, ld, cols[i * CPP + cb]) + 2 * ru;
        st[i] = *reinterpret_cast<const d2 *>(q_ + r0);
      }
    }
    if (WEIGHTED && tid < TPC) wst = in ? *reinterpret_cast<const d2 *>(w + r0 + 2 * ru) : d2{0.0, 0.0};
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int col = i * CPP + cb;
      if (col < mp) *reinterpret_cast<d2 *>(smem + (size_t)col * GL_LD + 2 * ru) = st[i];
    }
    if (WEIGHTED && tid < TPC) *reinterpret_cast<d2 *>(wch + 2 * ru) = wst;
  };
  int tI[TPW], tJ[TPW];
  d4 acc[TPW];
#pragma unroll
  for (int ts = 0; ts < TPW; ts++) {
    const int t = wv + NW * ts;
    int I = -1, J = -1;
    if (t < ntiles) tile_of(t, I, J);
    tI[ts] = __builtin_amdgcn_readfirstlane(I);
    tJ[ts] = __builtin_amdgcn_readfirstlane(J);
    acc[ts] = d4{0.0, 0.0, 0.0, 0.0};
  }
  auto compute = [&]() {
#pragma unroll
    for (int ts = 0; ts < TPW; ts++) {
      if (tI[ts] >= 0) {  // wave-uniform
        const double *pa = smem + (size_t)(tI[ts] * 16 + c) * GL_LD + 4 * q;
        const double *pb = smem + (size_t)(tJ[ts] * 16 + c) * GL_LD + 4 * q;
#pragma unroll
        for (int sx = 0; sx < GL_RB / 16; sx++) {
          const d2 a0 = *reinterpret_cast<const d2 *>(pa + 16 * sx), a1 = *reinterpret_cast<const d2 *>(pa + 16 * sx + 2);
          const d2 b0 = *reinterpret_cast<const d2 *>(pb + 16 * sx), b1 = *reinterpret_cast<const d2 *>(pb + 16 * sx + 2);
          double ax = a0.x, ay = a0.y, az = a1.x, aw = a1.y;
          if (WEIGHTED) {
            const d2 w0 = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q);
            const d2 w1 = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q + 2);
            ax *= w0.x;
            ay *= w0.y;
            az *= w1.x;
            aw *= w1.y;
          }
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b0.x, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b0.y, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(az, b1.x, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, b1.y, acc[ts], 0, 0, 0);
        }
      }
    }
  };
  if (nchunk > 0) {
    load(r_begin);
    store();
    if (nchunk > 1) load(r_begin + GL_RB);
    __syncthreads();
    for (int k = 0; k < nchunk; k++) {
      compute();
      __syncthreads();
      if (k + 1 < nchunk) store();
      __syncthreads();
      if (k + 2 < nchunk) load(r_begin + (long)(k + 2) * GL_RB);
    }
  }
  double *out = part + (size_t)slab * ntiles * 256;
#pragma unroll
  for (int ts = 0; ts < TPW; ts++)
    if (tI[ts] >= 0) *reinterpret_cast<d4 *>(out + (size_t)(wv + NW * ts) * 256 + lane * 4) = acc[ts];
}

__device__ __forceinline__ double clampv(double v, double c);
// IRLS step t of the GLM restricted fits (logistic :1148-1204, Poisson :1273-1322) in ONE pass over the active
// columns: linear predictor, working weights, working response, log-likelihood terms AND the slab's weighted Gram
// [1, X_A, z]^T diag(W w mask) [1, X_A, z] -- what k_glm_irls_prep + k_gram_lds did in two launches and two reads of
// X_A (round 2; an earlier fusion, k_gram_irls, did the per-row work chunk by chunk between the barriers of the
// staging pipeline and was no faster).  Here one 512-thread block owns a row slab (about one slab per compute unit)
// and walks it in groups of NCH 64-row chunks; per group it
//   (A) has ALL the group's loads in flight at once (thread (ru, cb): row pair ru of every chunk, columns cb, cb + 16,
//       ...; coalesced 16-byte loads, NCH x NPASS per thread) and KEEPS them in registers -- the loads of the next
//       group are issued chunk by chunk as the registers are emptied into the tile in (C), under the products;
//   (B) forms the linear predictor from those registers (per-thread partial over its columns, the two column halves
//       of a wave folded by one cross-lane add, the 8 waves through LDS), then EVERY row of the slab gets its
//       exp / log / division at once, one row per thread -- not 64 rows at a time between barriers;
//   (C) stages chunk after chunk from the registers into the LDS tile (no second read of X_A), the working response
//       as the last column, and multiplies on the fp64 matrix cores exactly like k_gram_lds.
// Output: the slab partials of all tiles (k_gram_reduce adds them up) and the slab's log-likelihood term (the
// convergence test at the head of k_chol adds those up).  Same arithmetic per row as k_glm_irls_prep; the linear
// predictor is summed in a different (fixed) order.
template <int NPASS, int NCH, int TPW, int FAM>
__global__ void __launch_bounds__(512) k_irls_gram(const double *__restrict__ X, const double *__restrict__ aux,
                                                   long ld, int n, const int *__restrict__ cols,
                                                   const double *__restrict__ y, const double *__restrict__ w,
                                                   const double *__restrict__ mask, int rows_per_slab, int mt,
                                                   double *__restrict__ part, int ntiles,
                                                   const FitCtrl *__restrict__ ctrl, int slot, int t, int T0,
                                                   const double *__restrict__ bcur, double *__restrict__ llpart,
                                                   int wfloor) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev || ctrl->irls_done || ctrl->irls_steps != t) return;
  constexpr int RB = 64, GL_LD = RB + 2, TPC = RB / 2, NW = 8, CPP = 64 * NW / TPC, ROWS = NCH * RB;
  extern __shared__ double smem[];  // [tile: mp x GL_LD | etap: NW x ROWS (phase B only)] Wl[ROWS] zl[ROWS] bet[mp]
  const int mp = mt * 16;
  const size_t tile_doubles = (size_t)mp * GL_LD > (size_t)NW * ROWS ? (size_t)mp * GL_LD : (size_t)NW * ROWS;
  double *etap = smem;
  double *Wl = smem + tile_doubles, *zl = Wl + ROWS, *bet = zl + ROWS;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ru = tid % TPC, cb = tid / TPC;
  const int slab = blockIdx.x;
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int ngroup = (int)((r_end - r_begin + ROWS - 1) / ROWS);  // the slab in groups of NCH chunks
  for (int i = tid; i < mp; i += 64 * NW) bet[i] = i <= T0 ? bcur[i] : 0.0;
  // (A) every load of a group in flight at once.  Column mp - 1 is the working response: formed here, never read.
  int cidx[NPASS];  // column of X (>= 0), auxiliary column (-1, -2), or none (INT_MIN): the pointer is formed per load
#pragma unroll
  for (int i = 0; i < NPASS; i++) {
    const int col = i * CPP + cb;
    cidx[i] = (col < mp - 1) ? cols[col] : INT_MIN;
  }
  d2 st[NCH][NPASS];
  auto load_chunk = [&](int ch, long g_begin) {
    const long r0 = g_begin + (long)ch * RB + 2 * ru;
    const bool in = r0 < r_end;  // ld is a multiple of 16: a row pair is in or out as a whole
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      st[ch][i] = d2{0.0, 0.0};
      if (in && cidx[i] != INT_MIN) st[ch][i] = *reinterpret_cast<const d2 *>(gram_col(X, aux, ld, cidx[i]) + r0);
    }
  };
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) load_chunk(ch, r_begin);
  int tI[TPW], tJ[TPW];
  d4 acc[TPW];
#pragma unroll
  for (int ts = 0; ts < TPW; ts++) {
    const int tt = wv + NW * ts;
    int I = -1, J = -1;
    if (tt < ntiles) tile_of(tt, I, J);
    tI[ts] = __builtin_amdgcn_readfirstlane(I);
    tJ[ts] = __builtin_amdgcn_readfirstlane(J);
    acc[ts] = d4{0.0, 0.0, 0.0, 0.0};
  }
  double ll = 0.0;
  for (int g = 0; g < ngroup; g++) {
    const long g_begin = r_begin + (long)g * ROWS;
    const int nch = (int)((min(g_begin + ROWS, r_end) - g_begin + RB - 1) / RB);
    double yy = 0.0, ww = 0.0, mm = 0.0;
    const long myrow = g_begin + tid;
    const bool rin = tid < ROWS && myrow < (long)n && myrow < r_end;  // (a group may reach beyond a short slab)
    if (rin) {
      yy = y[myrow];
      ww = w[myrow];
      mm = mask != nullptr ? mask[myrow] : 1.0;
    }
    __syncthreads();  // bet is there; the previous group's products have let go of the tile
    // (B) linear predictor of every row of the group
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      d2 e = d2{0.0, 0.0};
#pragma unroll
      for (int i = 0; i < NPASS; i++) {
        const int col = i * CPP + cb;
        e += st[ch][i] * (col < mp ? bet[col] : 0.0);
      }
      e.x += __shfl_xor(e.x, 32);  // the wave holds two column classes (cb even / odd) of the same row pairs
      e.y += __shfl_xor(e.y, 32);
      if (lane < 32) *reinterpret_cast<d2 *>(etap + (size_t)wv * ROWS + ch * RB + 2 * ru) = e;
    }
    __syncthreads();
    if (tid < ROWS) {
      double eta = ((etap[tid] + etap[ROWS + tid]) + (etap[2 * ROWS + tid] + etap[3 * ROWS + tid])) +
                   ((etap[4 * ROWS + tid] + etap[5 * ROWS + tid]) + (etap[6 * ROWS + tid] + etap[7 * ROWS + tid]));
      double Wt = 0.0, zt = 0.0;
      if (rin) {
        if (FAM == 2) {
          const double e = exp(clampv(eta, 30.0)), Pi = e / (1.0 + e);
          ll += (yy * log(Pi) + (1.0 - yy) * log(1.0 - Pi)) * ww * mm;
          double W = Pi * (1.0 - Pi);
          if (t > 0 && wfloor && W < 0.001) W = 0.001;  // (logit_fit of the screening has no floor, src/logistic.cpp:60-160)
          zt = eta + (yy - Pi) / W;
          Wt = W * ww * mm;
        } else {
          double e;
          if (t == 0) {
            e = exp(eta);
          } else {
            eta = clampv(eta, 30.0);
            e = exp(eta);
            if (e < 0.001) e = 0.001;
            ll += (yy * eta - e) * ww * mm;
          }
          zt = eta + (yy - e) / e;
          Wt = e * ww * mm;
        }
      }
      Wl[tid] = Wt;
      zl[tid] = zt;
    }
    __syncthreads();  // (etap is dead from here on: the tile takes its place)
    // (C) the group's Gram, chunk by chunk from the registers; a chunk's registers are refilled with the same chunk
    // of the NEXT group as soon as it is in the tile, so those loads fly under the products
#pragma unroll 1
    for (int ch = 0; ch < nch; ch++) {
      if (ch > 0) __syncthreads();
#pragma unroll
      for (int c2 = 0; c2 < NCH; c2++) {
        if (ch == c2) {  // block-uniform: the register array is indexed by a constant
#pragma unroll
          for (int i = 0; i < NPASS; i++) {
            const int col = i * CPP + cb;
            if (col < mp - 1) *reinterpret_cast<d2 *>(smem + (size_t)col * GL_LD + 2 * ru) = st[c2][i];
          }
          if (g + 1 < ngroup) load_chunk(c2, g_begin + ROWS);
        }
      }
      if (cb == (mp - 1) % CPP)
        *reinterpret_cast<d2 *>(smem + (size_t)(mp - 1) * GL_LD + 2 * ru) = *reinterpret_cast<const d2 *>(zl + ch * RB + 2 * ru);
      __syncthreads();
      const double *wch = Wl + ch * RB;
#pragma unroll
      for (int ts = 0; ts < TPW; ts++) {
        if (tI[ts] >= 0) {  // wave-uniform
          const double *pa = smem + (size_t)(tI[ts] * 16 + c) * GL_LD + 4 * q;
          const double *pb = smem + (size_t)(tJ[ts] * 16 + c) * GL_LD + 4 * q;
#pragma unroll
          for (int sx = 0; sx < RB / 16; sx++) {
            const d2 a0 = *reinterpret_cast<const d2 *>(pa + 16 * sx), a1 = *reinterpret_cast<const d2 *>(pa + 16 * sx + 2);
            const d2 b0 = *reinterpret_cast<const d2 *>(pb + 16 * sx), b1 = *reinterpret_cast<const d2 *>(pb + 16 * sx + 2);
            const d2 w0 = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q);
            const d2 w1 = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q + 2);
            acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x * w0.x, b0.x, acc[ts], 0, 0, 0);
            acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y * w0.y, b0.y, acc[ts], 0, 0, 0);
            acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x * w1.x, b1.x, acc[ts], 0, 0, 0);
            acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y * w1.y, b1.y, acc[ts], 0, 0, 0);
          }
        }
      }
    }
  }
  {  // log-likelihood terms of this slab (the convergence test adds the slabs up)
    __shared__ double llw[NW];
    ll = wave_sum(ll);
    if (lane == 0) llw[wv] = ll;
    __syncthreads();
    if (tid == 0)
      llpart[slab] = ((llw[0] + llw[1]) + (llw[2] + llw[3])) + ((llw[4] + llw[5]) + (llw[6] + llw[7]));
  }
  double *out = part + (size_t)slab * ntiles * 256;
#pragma unroll
  for (int ts = 0; ts < TPW; ts++)
    if (tI[ts] >= 0) *reinterpret_cast<d4 *>(out + (size_t)(wv + NW * ts) * 256 + lane * 4) = acc[ts];
}


__global__ void __launch_bounds__(256) k_gram_reduce(const double *__restrict__ part, int nslab, int ntiles,
                                                     double *__restrict__ Gt, const FitCtrl *__restrict__ ctrl,
                                                     int slot, int gate_mode) {
  if (ctrl != nullptr) {
    if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
    if ((gate_mode == 1 || gate_mode == 2) && ctrl->irls_done) return;
    if (gate_mode == 3 && !ctrl->gram_full) return;
    if (gate_mode == 4 && ctrl->gram_full) return;
  }
  __shared__ double sm[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t tot = (size_t)ntiles * 256;
  const size_t e = (size_t)blockIdx.x * 16 + el;
  double s = 0.0;
  if (e < tot)
    for (int sl = g; sl < nslab; sl += 16) s += part[(size_t)sl * tot + e];
  sm[g][el] = s;
  __syncthreads();
  if (g == 0 && e < tot) {
    double t = sm[0][el];
#pragma unroll
    for (int q = 1; q < 16; q++) t += sm[q][el];
    Gt[e] = t;
  }
}

// ------------------------------------------------------------------------------------------
// Incremental Gram for the LM fit.  X_A^T diag(mask) X_A depends only on the columns and the row set, never on
// the coefficients, and consecutive active sets of a warm-started path differ in a column or two.  Every row set
// keeps the Gram of its last solved active set (dense, symmetric, 256 x 256, double buffered) with its sorted
// column list.  k_gram_plan maps the new active set onto it; if at most 16 columns are new only their rows are
// formed (one extra tile row of k_gram) and k_gram_assemble builds the MFMA-layout tiles for k_chol from cache +
// new rows; otherwise the whole Gram is formed as before.  Either way the cache is refreshed.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gram_plan(const int *__restrict__ A_new, int T0, int mp,
                                                   const int *__restrict__ Ac, const int *__restrict__ meta,
                                                   int *__restrict__ src, int *__restrict__ cols,
                                                   FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kc = meta[0];
  int pos = -1, isnew = 0;
  if (tid < T0) {
    const int a = A_new[tid];
    int lo = 0, hi = kc - 1;
    while (lo <= hi) {
      int mid = (lo + hi) >> 1, v = Ac[mid];
      if (v == a) {
        pos = mid;
        break;
      }
      if (v < a)
        lo = mid + 1;
      else
        hi = mid - 1;
    }
    isnew = pos < 0 ? 1 : 0;
  }
  int inc = isnew;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int off = 0, nn = 0;
  for (int w = 0; w < 4; w++) {
    off += (w < wave) ? wsum[w] : 0;
    nn += wsum[w];
  }
  const int q = off + inc - isnew;  // rank of this column among the new ones
  if (tid < T0) {
    src[tid] = isnew ? -1 - q : pos;
    if (isnew && q < 16) cols[mp + q] = A_new[tid];
  }
  if (tid >= nn && tid < 16) cols[mp + tid] = -1;  // unused rows of the extra tile read the zero column
  if (tid == 0) ctrl->gram_full = (kc == 0 || nn > 16) ? 1 : 0;
}

// Build the tiles k_chol reads (incremental case) and refresh the cache (both cases).
__global__ void __launch_bounds__(256) k_gram_assemble(double *__restrict__ Gt, const double *__restrict__ Rt,
                                                       const int *__restrict__ src, int T0, double *gbuf0,
                                                       double *gbuf1, const int *__restrict__ meta,
                                                       const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
  // meta[1] names the current buffer; it only flips when k_gram_cache_commit really ran (device-side truth)
  const double *Gold = meta[1] ? gbuf1 : gbuf0;
  double *Gnew = meta[1] ? gbuf0 : gbuf1;
  const int t = blockIdx.x, lane = threadIdx.x >> 2, r = threadIdx.x & 3;
  int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  const int J = t - I * (I + 1) / 2;
  const int a = I * 16 + (lane >> 4) + 4 * r, b = J * 16 + (lane & 15);
  const size_t e = (size_t)t * 256 + lane * 4 + r;
  if (a >= T0 || b >= T0) {
    if (!ctrl->gram_full) Gt[e] = 0.0;
    return;
  }
  double v;
  if (ctrl->gram_full) {
    v = Gt[e];
  } else {
    const int sa = src[a], sb = src[b];
    if (sa >= 0 && sb >= 0) {
      v = Gold[(size_t)sb * 256 + sa];
    } else {
      // element (new row q, column c of the new active set) of the extra tile row, MFMA C/D layout
      const int q = sa < 0 ? -1 - sa : -1 - sb, c = sa < 0 ? b : a;
      v = Rt[(size_t)(c >> 4) * 256 + (((q & 3) << 4) + (c & 15)) * 4 + (q >> 2)];
    }
    Gt[e] = v;
  }
  Gnew[(size_t)b * 256 + a] = v;
  Gnew[(size_t)a * 256 + b] = v;
}

__global__ void __launch_bounds__(256) k_gram_cache_commit(const int *__restrict__ A_new, int T0,
                                                           int *__restrict__ Ac, int *__restrict__ meta,
                                                           const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
  for (int i = threadIdx.x; i < T0; i += 256) Ac[i] = A_new[i];
  if (threadIdx.x == 0) {
    meta[0] = T0;
    meta[1] ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// K7: Cholesky factorisation and both triangular solves of the (m x m) normal equations in ONE
// 512-thread workgroup, m + 1 <= 16*mt <= 256.
//
// Data placement: the lower triangle lives in REGISTERS for the whole factorisation, as 16x16 tiles
// in the f64-MFMA accumulator layout, dealt round-robin to the 8 waves (<= 17 tiles = 136 VGPRs per
// lane).  Per block column b: the owners publish the panel tiles (I,b) to LDS; wave 0 factors the
// 16x16 diagonal block in LDS; one thread per sub-diagonal row does the 16-step substitution against
// it; every wave then updates its own trailing tiles with 4 MFMAs per tile, operands read from the
// LDS panel.  The right-hand side rides along as row mp-1 of the matrix, so the forward solve is the
// factorisation itself; the backward solve walks the block columns in reverse with the L tiles still
// in registers (per-tile 16-vector products folded with two shuffles).
// ------------------------------------------------------------------------------------------
constexpr int CH_W = 8;                                                   // waves
constexpr int CH_MT = 16;                                                 // max tile rows
constexpr int CH_LDT = 17;                                                // padded tile row stride (doubles)

// broadcast the value lane `src` (compile-time constant after unrolling) holds to the whole wave via SGPRs
__device__ __forceinline__ double bcast_lane(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// broadcast lane j (0..15, a constant after unrolling) of every 16-lane row to the whole row: DPP row_newbcast,
// two VALU moves per double, no trip through SGPRs
__device__ __forceinline__ int dpp_row_share(int v, int j) {
  switch (j & 15) {
    case 0: return __builtin_amdgcn_update_dpp(0, v, 0x150, 0xF, 0xF, false);
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x151, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x152, 0xF, 0xF, false);
    case 3: return __builtin_amdgcn_update_dpp(0, v, 0x153, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x154, 0xF, 0xF, false);
    case 5: return __builtin_amdgcn_update_dpp(0, v, 0x155, 0xF, 0xF, false);
    case 6: return __builtin_amdgcn_update_dpp(0, v, 0x156, 0xF, 0xF, false);
    case 7: return __builtin_amdgcn_update_dpp(0, v, 0x157, 0xF, 0xF, false);
    case 8: return __builtin_amdgcn_update_dpp(0, v, 0x158, 0xF, 0xF, false);
    case 9: return __builtin_amdgcn_update_dpp(0, v, 0x159, 0xF, 0xF, false);
    case 10: return __builtin_amdgcn_update_dpp(0, v, 0x15A, 0xF, 0xF, false);
    case 11: return __builtin_amdgcn_update_dpp(0, v, 0x15B, 0xF, 0xF, false);
    case 12: return __builtin_amdgcn_update_dpp(0, v, 0x15C, 0xF, 0xF, false);
    case 13: return __builtin_amdgcn_update_dpp(0, v, 0x15D, 0xF, 0xF, false);
    case 14: return __builtin_amdgcn_update_dpp(0, v, 0x15E, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xF, 0xF, false);
  }
}
__device__ __forceinline__ double row_bcast16(double v, int j) {
  return __hiloint2double(dpp_row_share(__double2hiint(v), j), dpp_row_share(__double2loint(v), j));
}
// sum over the 16 lanes of a DPP row, result in every lane: mirror, half mirror, then the two quad swaps
__device__ __forceinline__ double dpp_mov64(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false); break;  // row_mirror
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); break;  // quad_perm [1,0,3,2]
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double v) {
  v += dpp_mov64(v, 0);
  v += dpp_mov64(v, 1);
  v += dpp_mov64(v, 2);
  v += dpp_mov64(v, 3);
  return v;
}
// element idx (lane-dependent) of a register array without dynamic indexing
__device__ __forceinline__ double bcast_pick16(const double (&a)[16], int idx) {
  double r = a[0];
#pragma unroll
  for (int i = 1; i < 16; i++) r = (idx == i) ? a[i] : r;
  return r;
}

__device__ __forceinline__ void tile_of(int t, int &I, int &J) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) i++;
  while (i * (i + 1) / 2 > t) i--;
  I = i;
  J = t - i * (i + 1) / 2;
}

__device__ __forceinline__ void commit_body(FitCtrl *__restrict__ ctrl, int slot, int T0,
                                            const int *__restrict__ A_new, const double *__restrict__ sol,
                                            int has_intercept, int wait_chain, int *__restrict__ A_cur,
                                            double *__restrict__ b_cur, double *__restrict__ beta_dense,
                                            int *__restrict__ hist, double *__restrict__ hist_beta,
                                            double *__restrict__ hist_coef0, int hist_stride, int *same_any_sh,
                                            unsigned char *__restrict__ inA);

// fz.G != nullptr (covariance mode of the LM fit): the Gram entries are gathered from the column cache
// (G[row A_a, column slot_of[A_b]]) instead of read from Gt, and the kernel ends with the work of k_commit -- two
// launches less per PDAS iteration.
// position of entry (row, col) of a 16 x 16 tile in the fp64-MFMA accumulator layout; index of tile (I, J), I >= J
__device__ __forceinline__ int tile_elem(int row, int col) { return ((((row & 3) << 4) | col) << 2) | (row >> 2); }
__device__ __forceinline__ size_t tile_id(int I, int J) { return (size_t)I * (I + 1) / 2 + J; }

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

template <int NT>
__device__ __forceinline__ bool irls_check_body(FitCtrl *__restrict__ ctrl, int t, int fam,
                                                const double *__restrict__ llpart, int nblk, int m,
                                                double *__restrict__ bcur, double *__restrict__ bprev);

template <int CH_SLOTS>  // register tiles per wave: 5 (mt <= 8), 7 (<= 10), 10 (<= 12), 14 (<= 14), 17 (<= 16)
__global__ void __launch_bounds__(512) k_chol(const double *__restrict__ Gt, int m, int mt, double ridge,
                                              int ridge_skip0, const double *__restrict__ rhs,
                                              const int *__restrict__ rhs_gather, double *__restrict__ sol,
                                              int *__restrict__ info, const FitCtrl *__restrict__ ctrl, int slot,
                                              int gate_mode, const CholFuse fz, const IrlsChk ck) {
  __shared__ int same_any_sh;
  if (ctrl != nullptr) {
    if (ctrl->done || ctrl->l != slot - 1) return;
    if (ctrl->same_prev) {
      if (fz.G != nullptr)
        commit_body(fz.ctrl, slot, fz.T0, rhs_gather, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist,
                    fz.hist_beta, fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
      return;
    }
    if ((gate_mode == 1 || gate_mode == 2) && ctrl->irls_done) return;
  }
  if (ck.on) {
    // IRLS step t, convergence test first (what k_glm_irls_check does as its own launch): the log-likelihood terms
    // of the iterate this Gram was formed at are in ck.llpart.  Converged (or out of steps): nothing is solved.
    if (ck.ctrl->irls_steps != ck.t) return;
    if (irls_check_body<512>(ck.ctrl, ck.t, ck.fam, ck.llpart, ck.nblk, ck.m, ck.bcur, ck.bprev)) return;
  }
  // LDS images are addressed by integer offsets.  The single-wave phases below pass data between
  // lanes through LDS; WAVE_SYNC orders them (LDS executes one wave's DS operations in order; the
  // fence keeps the compiler from moving accesses across it and drains lgkmcnt).
  constexpr int TS = 16 * CH_LDT;             // doubles per padded tile
  __shared__ double Psh[2 * CH_MT * TS];      // panel tiles, double buffered
  __shared__ double Lsh[CH_MT * TS];          // factored diagonal blocks
  __shared__ double z[CH_MT * 16];            // right-hand side / solution
  __shared__ double Rsh[CH_MT * 16];          // reciprocals of the diagonal of L
  __shared__ double dorig[CH_MT * 16];        // the diagonal as loaded (a pivot below 1e-11 of it: rank-deficient system)
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile indices live in SGPRs
  const int lc = lane & 15, lq = lane >> 4;
  const int mp = mt * 16, ntiles = mt * (mt + 1) / 2;
  d4 acc[CH_SLOTS];
  int tI[CH_SLOTS], tJ[CH_SLOTS];
  __shared__ int sA[CH_MT * 16], sS[CH_MT * 16];  // gather mode: active columns and their cache slots
  if (fz.G != nullptr) {
    for (int i = tid; i < m; i += 512) {
      const int a = rhs_gather[i];
      sA[i] = a;
      sS[i] = fz.slot_of[a];
    }
    __syncthreads();
  }
  // tiles are dealt round-robin: slot s of this wave is tile t = 8 s + wave of the packed lower triangle.  All
  // loads are issued first (they land in the accumulators), the fix-ups follow.
  {
    int ti, tj;
    tile_of(wave, ti, tj);
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      const bool have = s * CH_W + wave < ntiles;
      tI[s] = __builtin_amdgcn_readfirstlane(have ? ti : -1);  // wave-uniform: keep the tile indices in SGPRs
      tJ[s] = __builtin_amdgcn_readfirstlane(have ? tj : -1);
      tj += CH_W;
      while (tj > ti) {
        tj -= ti + 1;
        ti++;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < CH_SLOTS; s++) {
    acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    if (tI[s] >= 0) {
      if (fz.G != nullptr) {
        const int col = tJ[s] * 16 + lc;
        const int sl = col < m ? sS[col] : 0;
        if (sl < 0) fz.ctrl->cov_miss = 1;  // must not happen: the active columns were cached before this launch
        const double *gcol = fz.G + (size_t)(sl < 0 ? 0 : sl) * fz.p;
        double gv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = tI[s] * 16 + lq + 4 * r;
          gv[r] = (row < m && col < m && sl >= 0) ? gcol[sA[row]] : 0.0;
        }
        acc[s] = d4{gv[0], gv[1], gv[2], gv[3]};
      } else {
        acc[s] = *reinterpret_cast<const d4 *>(Gt + (size_t)(s * CH_W + wave) * 256 + lane * 4);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < CH_SLOTS; s++) {
    if (tI[s] >= 0) {
      double gv[4] = {acc[s].x, acc[s].y, acc[s].z, acc[s].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int row = tI[s] * 16 + lq + 4 * r, col = tJ[s] * 16 + lc;
        double v = gv[r];
        if (row == col && row < m && !(ridge_skip0 && row == 0)) v += ridge;
        if (rhs != nullptr) {
          // right-hand side supplied separately (LM: gathered X^T y): it becomes row mp-1
          if (row >= m || col >= m) v = (row == col) ? 1.0 : 0.0;
          if (row == mp - 1 && col < m) v = rhs[rhs_gather ? rhs_gather[col] : col];
        } else {
          // right-hand side is Gram column mp-1 already (IRLS); only fix the padding
          bool rpad = row >= m && row != mp - 1, cpad = col >= m && col != mp - 1;
          if (rpad || cpad) v = (row == col) ? 1.0 : 0.0;
          if (row == mp - 1 && col == mp - 1) v = 1.0;
        }
        gv[r] = v;
        if (row == col) dorig[row] = v;
      }
      acc[s] = d4{gv[0], gv[1], gv[2], gv[3]};
    }
  }
  for (int b = 0; b < mt; b++) {
    const int pb = (b & 1) * CH_MT * TS;  // panel buffer base
    // 1. publish panel tiles (I, b)
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++)
      if (tJ[s] == b) {
        const int o = pb + tI[s] * TS + lq * CH_LDT + lc;
        Psh[o] = acc[s].x;
        Psh[o + 4 * CH_LDT] = acc[s].y;
        Psh[o + 8 * CH_LDT] = acc[s].z;
        Psh[o + 12 * CH_LDT] = acc[s].w;
      }
    __syncthreads();
    // 2+3. EVERY wave factors the 16x16 diagonal block redundantly in registers (lane holds row lane&15;
    // pivots and multipliers are broadcast inside each 16-lane row with DPP row_newbcast, so the 16-step chain needs
    // no LDS round trips, no SGPR traffic and no barrier before the substitution; 1/sqrt(pivot) comes from v_rsq_f64
    // + two Newton steps instead of a square root and a division), then does the 16-step substitution
    // x * Lbb^T = p for its own rows with the L entries broadcast the same way.  Wave 0 also stores Lbb (for the backward solve and the owner of tile (b,b)).
    {
      if constexpr (CH_SLOTS <= 10) {
        // DPP form: fewer instructions, but the broadcast values live in VGPRs -- affordable up to 10 tile slots
        const int D = pb + b * TS, rr = lane & 15;
        double Lr[16], rinv_mine = 0.0;  // lane j keeps 1 / L[j][j]
  #pragma unroll
        for (int c = 0; c < 16; c++) Lr[c] = Psh[D + rr * CH_LDT + c];
  #pragma unroll
        for (int j = 0; j < 16; j++) {
          const double pjj = row_bcast16(Lr[j], j);
          // 1/sqrt(pjj): hardware estimate + two Newton steps; sqrt(pjj) from it with one correction
          double r = __builtin_amdgcn_rsq(pjj);
          r = fma(0.5 * r, fma(-pjj * r, r, 1.0), r);
          r = fma(0.5 * r, fma(-pjj * r, r, 1.0), r);
          double d = pjj * r;
          d = fma(fma(-d, d, pjj), 0.5 * r, d);
          if (rr == j) rinv_mine = r;
          if (wave == 0 && lane == j) Rsh[b * 16 + j] = r;
          const double lij = (rr == j) ? d : Lr[j] * r;  // rows < j hold unused upper-triangle values
          Lr[j] = lij;
  #pragma unroll
          for (int c = j + 1; c < 16; c++) Lr[c] = fma(-lij, row_bcast16(lij, c), Lr[c]);
          __builtin_amdgcn_sched_barrier(0);  // keep the broadcasts of later steps from being hoisted (VGPR pressure)
        }
        if (wave == 0 && lane < 16) {
  #pragma unroll
          for (int c = 0; c < 16; c++) {
            Psh[D + rr * CH_LDT + c] = Lr[c];
            Lsh[b * TS + rr * CH_LDT + c] = Lr[c];
          }
        }
        const int nrows = (mt - b - 1) * 16;
        if (tid < nrows) {  // a multiple of 16: every 16-lane row is either fully active or idle
          const int pr = pb + (b + 1 + (tid >> 4)) * TS + (tid & 15) * CH_LDT;
          double x[16];
  #pragma unroll
          for (int j = 0; j < 16; j++) x[j] = Psh[pr + j];
  #pragma unroll
          for (int j = 0; j < 16; j++) {
            double sacc = x[j];
  #pragma unroll
            for (int t = 0; t < j; t++) sacc = fma(-x[t], row_bcast16(Lr[t], j), sacc);  // L[j][t] lives in lane j
            x[j] = sacc * row_bcast16(rinv_mine, j);
            __builtin_amdgcn_sched_barrier(0);
          }
  #pragma unroll
          for (int j = 0; j < 16; j++) Psh[pr + j] = x[j];
        }
      } else {
        // 17 tile slots leave no VGPRs to spare: broadcasts go through SGPRs (v_readlane), sqrt + reciprocal
        const int D = pb + b * TS, rr = lane & 15;
        double Lr[16], rinv[16];
  #pragma unroll
        for (int c = 0; c < 16; c++) Lr[c] = Psh[D + rr * CH_LDT + c];
  #pragma unroll
        for (int j = 0; j < 16; j++) {
          const double pjj = bcast_lane(Lr[j], j);
          const double d = sqrt(pjj);
          rinv[j] = bcast_lane(1.0 / d, 0);  // wave-uniform: keep it in SGPRs
          if (wave == 0 && lane == j) Rsh[b * 16 + j] = rinv[j];
          const double lij = (rr == j) ? d : Lr[j] * rinv[j];  // rows < j hold unused upper-triangle values
          Lr[j] = lij;
  #pragma unroll
          for (int c = j + 1; c < 16; c++) Lr[c] = fma(-lij, bcast_lane(lij, c), Lr[c]);
        }
        if (wave == 0 && lane < 16) {
  #pragma unroll
          for (int c = 0; c < 16; c++) {
            Psh[D + rr * CH_LDT + c] = Lr[c];
            Lsh[b * TS + rr * CH_LDT + c] = Lr[c];
          }
        }
        const int nrows = (mt - b - 1) * 16;
        if (tid < nrows) {
          const int pr = pb + (b + 1 + (tid >> 4)) * TS + (tid & 15) * CH_LDT;
          double x[16];
  #pragma unroll
          for (int j = 0; j < 16; j++) x[j] = Psh[pr + j];
  #pragma unroll
          for (int j = 0; j < 16; j++) {
            double sacc = x[j];
  #pragma unroll
            for (int t = 0; t < j; t++) sacc = fma(-x[t], bcast_lane(Lr[t], j), sacc);  // L[j][t] lives in lane j
            x[j] = sacc * rinv[j];
          }
  #pragma unroll
          for (int j = 0; j < 16; j++) Psh[pr + j] = x[j];
        }
      }
    }
    __syncthreads();
    // 4. owners take the finished panel tile back; everyone updates its trailing tiles
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      if (tJ[s] == b) {
        const int o = pb + tI[s] * TS + lq * CH_LDT + lc;
        acc[s] = d4{Psh[o], Psh[o + 4 * CH_LDT], Psh[o + 8 * CH_LDT], Psh[o + 12 * CH_LDT]};
      } else if (tJ[s] > b) {
        const int oi = pb + tI[s] * TS + lc * CH_LDT + lq;
        const int oj = pb + tJ[s] * TS + lc * CH_LDT + lq;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
          double av = -Psh[oi + k4 * 4], bv = Psh[oj + k4 * 4];
          acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[s], 0, 0, 0);
        }
      }
    }
    // no barrier: the next step publishes into the other panel buffer, and three barriers separate
    // this step's reads from the next write of this buffer.
  }
  // ---- backward solve  L^T x = y, y = row mp-1 of L (lanes 48..63, reg 3 of the tiles (mt-1, J))
#pragma unroll
  for (int s = 0; s < CH_SLOTS; s++)
    if (tI[s] == mt - 1 && lq == 3) z[tJ[s] * 16 + lc] = acc[s].w;
  __syncthreads();
  if (tid == 0) z[mp - 1] = 0.0;  // the augmented row itself is not an unknown
  __syncthreads();
  for (int b = mt - 1; b >= 0; b--) {
    if (wave == 0) {
      // lane i (< 16) holds z_i and COLUMN i of Lbb, so L[j][i] is a compile-time register of lane i
      const int zz = b * 16, L = b * TS, ci = lane & 15;
      double Lc[16], zi = z[zz + ci];
#pragma unroll
      for (int r = 0; r < 16; r++) Lc[r] = Lsh[L + r * CH_LDT + ci];
      const double ri = Rsh[zz + ci];
#pragma unroll
      for (int j = 15; j >= 0; j--) {
        double xj = bcast_lane(zi, j) * bcast_lane(ri, j);
        if (b == mt - 1 && j == 15) xj = 0.0;  // the augmented row is not an unknown
        zi = (ci == j) ? xj : ((ci < j) ? fma(-Lc[j], xj, zi) : zi);
      }
      if (lane < 16) z[zz + ci] = zi;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      if (tI[s] == b && tJ[s] < b) {
        const int zb = b * 16 + lq;
        double v = acc[s].x * z[zb] + acc[s].y * z[zb + 4] + acc[s].z * z[zb + 8] + acc[s].w * z[zb + 12];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) z[tJ[s] * 16 + lc] = z[tJ[s] * 16 + lc] - v;
      }
    }
    __syncthreads();
  }
  bool bad = false;
  if (tid < m) {
    double v = z[tid];
    sol[tid] = v;
    bad = !(fabs(v) <= DBL_MAX);
  }
  // rank-deficient to working precision?  1 / L_jj is still in LDS (Rsh) and so is the diagonal as it was loaded: a
  // pivot below 1e-11 of its own diagonal entry means exactly dependent columns (duplicates both in the active set,
  // more columns than independent rows) -- the quotient of two rounding errors would follow.  Such a system, and one
  // whose solution is not finite (negative pivot), is left to the pivoted solve (k_sym_fallback): info = 2.
  if (tid < m) {
    const double rinv = Rsh[tid];
    bad = bad || !(1.0 > 1e-11 * dorig[tid] * rinv * rinv);
  }
  const bool failed = __syncthreads_or(bad) != 0;
  if (failed && tid == 0 && info != nullptr) *info = 2;
  if (failed) return;  // (nothing is committed: k_sym_fallback solves and commits, or the host reports the error)
  if (fz.G != nullptr) {  // sol is visible to the whole block after the barrier above
    // the loss of this solve comes from a residual pass (k_resid_lm): only k_cg has the true residual of the
    // normal equations at hand that makes the solved-system formula an identity
    if (tid == 0) fz.ctrl->sse_valid = 0;
    commit_body(fz.ctrl, slot, fz.T0, rhs_gather, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
                fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
  }
#undef WAVE_SYNC
}

// The pivoted solve behind a k_chol launch that gave up (info = 2): same arguments, same gate; the matrix is read again
// (tiles Gt, or gathered from the Gram column cache like k_chol does), laid out densely in fz.fb_work and solved by
// sym_pivoted_solve.  Success clears info (and, in the covariance form, does the commit k_chol skipped); a result
// that is still not finite leaves info = 1 for the host.  One workgroup; falls through in ~2 us when there is nothing
// to repair.
__global__ void __launch_bounds__(512) k_sym_fallback(const double *__restrict__ Gt, int m, int mt, double ridge,
                                                      int ridge_skip0, const double *__restrict__ rhs,
                                                      const int *__restrict__ rhs_gather, double *__restrict__ sol,
                                                      int *__restrict__ info, const FitCtrl *__restrict__ ctrl,
                                                      int slot, const CholFuse fz) {
  if (info == nullptr || *info != 2) return;
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  __shared__ int same_any_sh;
  __shared__ int sA[CH_MT * 16], sS[CH_MT * 16];
  const int tid = threadIdx.x;
  if (fz.G != nullptr) {
    for (int i = tid; i < m; i += 512) {
      const int a = rhs_gather[i];
      sA[i] = a;
      sS[i] = max(fz.slot_of[a], 0);
    }
    __syncthreads();
  }
  double *A = fz.fb_work, *bb = A + 256 * 256, *dvv = bb + 256;
  int *perm = reinterpret_cast<int *>(dvv + 256), *zf = perm + 256;
  for (int idx = tid; idx < m * m; idx += 512) {
    const int i = idx % m, j = idx / m;
    double v;
    if (fz.G != nullptr) {
      v = fz.G[(size_t)sS[j] * fz.p + sA[i]];
    } else {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      v = Gt[tile_id(hi >> 4, lo >> 4) * 256 + tile_elem(hi & 15, lo & 15)];
    }
    if (i == j && !(ridge_skip0 && i == 0)) v += ridge;
    A[idx] = v;
  }
  for (int i = tid; i < m; i += 512)
    bb[i] = rhs != nullptr ? rhs[rhs_gather ? rhs_gather[i] : i]
                           : Gt[tile_id(mt - 1, i >> 4) * 256 + tile_elem(15, i & 15)];  // IRLS: Gram column mp - 1
  __syncthreads();
  sym_pivoted_solve<512>(A, m, bb, dvv, perm, zf);
  bool bad = false;
  if (tid < m) {
    const double v = bb[tid];
    sol[tid] = v;
    bad = !(fabs(v) <= DBL_MAX);
  }
  const bool failed = __syncthreads_or(bad) != 0;
  if (tid == 0) *info = failed ? 1 : 0;
  if (failed) return;
  if (fz.G != nullptr) {
    if (tid == 0) fz.ctrl->sse_valid = 0;
    commit_body(fz.ctrl, slot, fz.T0, rhs_gather, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
                fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
  }
}

// ------------------------------------------------------------------------------------------
// K7', covariance form of the LM fit: the k x k normal equations by conjugate gradients instead of a factorisation.
//
// After normalisation the Gram of an active set is n (I + small), and the solve is warm-started from the previous
// coefficients (beta_dense[A_new]: exact for the columns that stay, 0 for the new ones), so the residual drops to
// rounding level in a handful of steps, each one a symmetric k x k matrix-vector product with the tiles held in
// registers (same tile dealing and gather as k_chol) -- no sequential pivot chain.  Every reduction has a fixed
// order.  The iterate is accepted only if the TRUE residual |q - (G + ridge I) x| <= 1e-13 |q| (recomputed, not the
// recurrence); otherwise (ill-conditioned design, iteration cap) the fit is parked with cov_stall = 2 and the host
// issues the Cholesky kernel for this slot.  Ends with the loss terms and k_commit's work like the fused k_chol.
// ------------------------------------------------------------------------------------------
// NW = waves of the workgroup: 8, or 1 for systems of at most 64 unknowns (no block barriers at all then).
#ifdef BESSX_CG_PROFILE
__device__ unsigned long long g_cg_prof[16];
#define CGP(i)                                                         \
  do {                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    if (threadIdx.x == 0) {                                            \
      unsigned long long now_ = wall_clock64();                        \
      atomicAdd(&g_cg_prof[i], now_ - tprev_);                         \
      tprev_ = now_;                                                   \
    }                                                                  \
  } while (0)
// inside the loop: accumulated in registers of thread 0 (an atomic per stamp would be waited for by the next one)
#define CGL_DECL() unsigned long long lacc_[4] = {0ull, 0ull, 0ull, 0ull}
#define CGL(i)                                        \
  do {                                                \
    if (threadIdx.x == 0) {                           \
      unsigned long long now_ = wall_clock64();       \
      lacc_[i] += now_ - tprev_;                      \
      tprev_ = now_;                                  \
    }                                                 \
  } while (0)
#define CGL_FLUSH()                                                        \
  do {                                                                     \
    if (threadIdx.x == 0)                                                  \
      for (int q_ = 0; q_ < 4; q_++) atomicAdd(&g_cg_prof[7 + q_], lacc_[q_]); \
  } while (0)
#else
#define CGP(i)
#define CGL_DECL()
#define CGL(i)
#define CGL_FLUSH()
#endif
template <int CH_SLOTS, int NW>
__device__ __forceinline__ void cg_body(int m, int mt, double ridge, const double *__restrict__ rhs,
                                        const int *__restrict__ A_new, double *__restrict__ sol,
                                        const FitCtrl *__restrict__ ctrl, int slot, const CholFuse &fz, int maxit,
                                        const double tol) {
  __shared__ int same_any_sh;
  if (ctrl->done || ctrl->l != slot - 1) return;
  if (ctrl->same_prev) {
    commit_body(fz.ctrl, slot, fz.T0, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
                fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
    return;
  }
  if (fz.dep != nullptr && *fz.dep != 0) {  // exactly dependent columns cached: k_chol's pivot test decides (see cgr_body)
    if (threadIdx.x == 0) {
      fz.ctrl->cov_stall = 2;
      fz.ctrl->l = -1 - fz.ctrl->l;
    }
    return;
  }
#ifdef BESSX_CG_PROFILE
  unsigned long long tprev_ = wall_clock64();
  if (threadIdx.x == 0) atomicAdd(&g_cg_prof[15], 1ull);
#endif
  // one wave: LDS traffic of a wave is in order, a fence keeps the compiler from reordering it
#define CG_SYNC()                                            \
  do {                                                       \
    if (NW == 1) {                                           \
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
      __builtin_amdgcn_wave_barrier();                       \
    } else {                                                 \
      __syncthreads();                                       \
    }                                                        \
  } while (0)
  __shared__ int sA[CH_MT * 16], sS[CH_MT * 16];
  __shared__ double pv[CH_MT * 16];
  __shared__ double yw[2][NW][CH_MT * 16];
  __shared__ double red[2][NW], red2[2][NW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, lq = lane >> 4;
  const int mp = mt * 16, ntiles = mt * (mt + 1) / 2;
  for (int i = tid; i < mp; i += 64 * NW) {
    const int a = i < m ? A_new[i] : 0;
    sA[i] = a;
    sS[i] = i < m ? fz.slot_of[a] : 0;
  }
  CG_SYNC();
  CGP(0);
  d4 acc[CH_SLOTS];
  int tI[CH_SLOTS], tJ[CH_SLOTS];
  {
    int ti, tj;
    tile_of(wave, ti, tj);
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      const bool have = s * NW + wave < ntiles;
      tI[s] = __builtin_amdgcn_readfirstlane(have ? ti : -1);
      tJ[s] = __builtin_amdgcn_readfirstlane(have ? tj : -1);
      tj += NW;
      while (tj > ti) {
        tj -= ti + 1;
        ti++;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < CH_SLOTS; s++) {
    acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    if (tI[s] >= 0) {
      const int col = tJ[s] * 16 + lc;
      const int sl = col < m ? sS[col] : 0;
      if (sl < 0) fz.ctrl->cov_miss = 1;
      const double *gcol = fz.G + (size_t)(sl < 0 ? 0 : sl) * fz.p;
      double gv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = tI[s] * 16 + lq + 4 * r;
        gv[r] = (row < m && col < m && sl >= 0) ? gcol[sA[row]] : 0.0;
      }
      acc[s] = d4{gv[0], gv[1], gv[2], gv[3]};
    }
  }
  CGP(1);
  const bool own = tid < m;  // thread t owns element t of every k-vector
  // block-wide dot products (two at once), fixed order; every thread gets the values.  One barrier: the buffers
  // alternate, and a buffer is rewritten only after another barrier has been passed by everybody.
  int rb = 0;
  auto dot2 = [&](double a, double b, double c, double d, double &ab, double &cd) {
    // wave sums: the 16 lanes of a row in the VALU (DPP), then the four rows
    double v = row_sum16(own ? a * b : 0.0), u = row_sum16(own ? c * d : 0.0);
    v += __shfl_xor(v, 16);
    u += __shfl_xor(u, 16);
    v += __shfl_xor(v, 32);
    u += __shfl_xor(u, 32);
    if (lane == 0) {
      red[rb][wave] = v;
      red2[rb][wave] = u;
    }
    CG_SYNC();
    double t = red[rb][0], t2 = red2[rb][0];
#pragma unroll
    for (int w = 1; w < NW; w++) {
      t += red[rb][w];
      t2 += red2[rb][w];
    }
    rb ^= 1;
    ab = t;
    cd = t2;
  };
  auto dot = [&](double a, double b) -> double {
    double x1, x2;
    dot2(a, b, 0.0, 0.0, x1, x2);
    return x1;
  };
  // y = G v for the vector v (one element per owning thread); symmetric product from the lower-triangle tiles.
  // Two barriers: the per-wave partial results alternate between two buffers.
  int yb = 0;
  auto matvec = [&](double v) -> double {
    if (tid < mp) pv[tid] = own ? v : 0.0;
    double(*yy)[CH_MT * 16] = yw[yb];
#pragma unroll
    for (int q = 0; q < CH_MT * 16 / 64; q++) yy[wave][lane + 64 * q] = 0.0;
    CG_SYNC();
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      if (tI[s] >= 0) {
        const int I = tI[s], J = tJ[s];
        // (a) rows of tile row I: y[I*16 + lq + 4r] += sum_lc T[r] * v[J*16 + lc]; the sum over the 16 lanes of a
        // row stays in the VALU (DPP mirror / quad permutes), lane lc = r of each row then owns output row r
        const double pj = pv[J * 16 + lc];
        const double v0 = row_sum16(acc[s].x * pj), v1 = row_sum16(acc[s].y * pj);
        const double v2 = row_sum16(acc[s].z * pj), v3 = row_sum16(acc[s].w * pj);
        if (lc < 4) {
          const double mine = lc == 0 ? v0 : (lc == 1 ? v1 : (lc == 2 ? v2 : v3));
          yy[wave][I * 16 + lq + 4 * lc] += mine;
        }
        // (b) columns of the tile (off-diagonal tiles only): y[J*16 + lc] += sum_rows T[row][lc] * v[I*16 + row]
        if (I != J) {
          const int ib = I * 16 + lq;
          double w0 = acc[s].x * pv[ib] + acc[s].y * pv[ib + 4] + acc[s].z * pv[ib + 8] + acc[s].w * pv[ib + 12];
          w0 += __shfl_xor(w0, 16);
          w0 += __shfl_xor(w0, 32);
          if (lq == 0) yy[wave][J * 16 + lc] += w0;
        }
      }
    }
    CG_SYNC();
    double y = 0.0;
    if (tid < mp) {
#pragma unroll
      for (int w = 0; w < NW; w++) y += yy[w][tid];
    }
    yb ^= 1;
    return y;
  };
  const double q_t = own ? rhs[sA[tid]] : 0.0;
  // warm start: previous coefficients of the columns that stay; a column that enters starts from its one-variable
  // update d_j / (G_jj + ridge) (d = X^T r of the current coefficients is what the scores were made of)
  double x_t = 0.0;
  if (own) {
    x_t = fz.beta_dense[sA[tid]];
    if (x_t == 0.0 && fz.d != nullptr && sS[tid] >= 0)
      x_t = fz.d[sA[tid]] / (fz.G[(size_t)sS[tid] * fz.p + sA[tid]] + ridge);
  }
  CGP(2);
  const double qq = dot(q_t, q_t);
  double r_t = q_t - (matvec(x_t) + ridge * x_t);
  double p_t = r_t;
  double rs = dot(r_t, r_t);
  CGP(3);
  bool ok = false;
  int it = 0;
  for (int round = 0; round < 3 && !ok; round++) {
    for (; it < maxit && rs > 1e-4 * tol * tol * qq; it++) {  // recurrence residual target: |r| <= tol / 100 |q|
      const double ap = matvec(p_t) + ridge * p_t;
      const double alpha = rs / dot(p_t, ap);
      x_t = fma(alpha, p_t, x_t);
      r_t = fma(-alpha, ap, r_t);
      const double rs_new = dot(r_t, r_t);
      p_t = fma(rs_new / rs, p_t, r_t);
      rs = rs_new;
    }
    // the recurrence drifts: accept only on the recomputed residual |q - (G + ridge I) x| <= 1e-13 |q|
    r_t = q_t - (matvec(x_t) + ridge * x_t);
    rs = dot(r_t, r_t);
    ok = rs <= tol * tol * qq;  // also catches NaN (singular / indefinite matrix): the comparison fails
    p_t = r_t;
    if (it >= maxit) break;
  }
  CGP(4);
#ifdef BESSX_CG_PROFILE
  if (threadIdx.x == 0) atomicAdd(&g_cg_prof[14], (unsigned long long)it);
#endif
  if (!ok) {
    if (tid == 0) {  // park the fit: the host issues the Cholesky kernel for this slot
      fz.ctrl->cov_stall = 2;
      fz.ctrl->l = -1 - fz.ctrl->l;
    }
    return;
  }
  if (own) sol[tid] = x_t;
  {
    // loss terms: |y - X b|^2 = y.y - b.(q + rho) - ridge |b|^2 with rho = q - (G + ridge I) b the residual just
    // recomputed (an identity, not an approximation).  It is only handed to the host when the cancellation is
    // harmless: the sum of the magnitudes of the terms times the unit roundoff stays below 1e-10 of the result.
    // Error of the evaluation: rounding of the three terms (their magnitudes a3, yy) and of the cached Gram entries
    // themselves, which enters through b' dG b ~ eps * max diag(G) * |b|^2 (dominant when collinear columns blow
    // the coefficients up).
    const double qr = q_t + r_t;
    const double a1 = dot(x_t, qr), a2 = dot(x_t, x_t), a3 = dot(fabs(x_t), fabs(qr));
    double gd = (own && sS[tid] >= 0) ? fz.G[(size_t)sS[tid] * fz.p + sA[tid]] : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) gd = fmax(gd, __shfl_xor(gd, o));
    if (lane == 0) red[rb][wave] = gd;
    CG_SYNC();
    gd = red[rb][0];
#pragma unroll
    for (int w = 1; w < NW; w++) gd = fmax(gd, red[rb][w]);
    rb ^= 1;
    if (tid == 0) {
      const double tr = fz.yy - a1 - ridge * a2;
      fz.ctrl->sse_dot = a1;
      fz.ctrl->sse_nrm = a2;
      fz.ctrl->sse_valid =
          (tr > 1e-6 * fz.yy && 4e-16 * (a3 + (ridge + gd) * a2 + fz.yy) <= 1e-10 * tr) ? 1 : 0;
    }
  }
  CG_SYNC();
  CGP(5);
  commit_body(fz.ctrl, slot, fz.T0, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
              fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
  CGP(6);
}

#undef CG_SYNC

template <int CH_SLOTS, int NW>
__global__ void __launch_bounds__(64 * NW) k_cg(int m, int mt, double ridge, const double *__restrict__ rhs,
                                            const int *__restrict__ A_new, double *__restrict__ sol,
                                            const FitCtrl *__restrict__ ctrl, int slot, const CholFuse fz, int maxit,
                                            const double tol) {
  KT(3);
  cg_body<CH_SLOTS, NW>(m, mt, ridge, rhs, A_new, sol, ctrl, slot, fz, maxit, tol);
  if (fz.pub.on) {  // last kernel of a batch of slots: publish (or snapshot) the result block, whatever the body did
    __syncthreads();
    if (fz.pub.on == 2) {
      if (fz.ctrl->snap_seq != fz.pub.seq) snapshot_body(fz.pub);  // (else the selection kernel has taken it already)
    } else {
      publish_body(fz.pub);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K7'', the same conjugate-gradient solve with the matrix dealt by ROWS instead of MFMA tiles (systems of at most
// 64 * RPT rows and 8 * NCW columns, i.e. up to 192 unknowns; larger ones keep k_cg).
//
// Wave w holds columns [w * nc, (w + 1) * nc) of ALL rows: lane l owns rows l, l + 64, ... (RPT of them), NCW matrix
// elements per row in registers.  A product y = G v is then: v to LDS, barrier, every lane multiplies its row
// segments with the broadcast v_j (no cross-lane traffic at all), the 8 segment partials go to LDS, barrier, every
// wave sums the 8 partials of all rows in fixed order.  Because every wave ends up with the WHOLE vector (x, r, p
// are kept replicated in all 8 waves), the dot products are wave-local DPP reductions -- no third barrier, and every
// wave computes bit-identical scalars.  Two barriers per CG step instead of four, no 16-lane row sums per tile.
// Same warm start, same acceptance test on the recomputed residual, same loss identity and commit as k_cg.
// ------------------------------------------------------------------------------------------
template <int RPT, int NCW>
__device__ __forceinline__ void cgr_body(int m, int nc, double ridge, const double *__restrict__ rhs,
                                         const int *__restrict__ A_new, double *__restrict__ sol,
                                         const FitCtrl *__restrict__ ctrl, int slot, const CholFuse &fz, int maxit,
                                         const double tol) {
  __shared__ int same_any_sh;
  if (ctrl->done || ctrl->l != slot - 1) return;
  if (ctrl->same_prev) {
    commit_body(fz.ctrl, slot, fz.T0, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
                fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
    return;
  }
  if (fz.dep != nullptr && *fz.dep != 0) {
    // exactly dependent columns are cached for this row set (k_cov_compact): a system that holds such a pair is for
    // the factorisation with the pivot test (k_chol -> sym_pivoted_solve), not for an iteration that would quietly
    // return one of its many solutions.  Parked like a solve that missed its residual target.
    if (threadIdx.x == 0) {
      fz.ctrl->cov_stall = 2;
      fz.ctrl->l = -1 - fz.ctrl->l;
    }
    return;
  }
#ifdef BESSX_CG_PROFILE
  unsigned long long tprev_ = wall_clock64();
  if (threadIdx.x == 0) atomicAdd(&g_cg_prof[15], 1ull);
#endif
  constexpr int R = 64 * RPT;
  __shared__ int sA[R], sS[R];
  __shared__ double pv[R], qs[R], xs[R], gds[R];
  // partial products of the 8 waves, two buffers in turn (a step's sums may still be read by a slow wave while a
  // fast one writes the next step's: one barrier per product is then enough), and one copy of the multiplied vector
  // per wave (every wave holds it: no barrier between writing and reading one's own copy)
  __shared__ double part[2][8][R];
  __shared__ double pvw[8][R];
  int pbuf = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int off_gs = 0;  // a column of this system outside the slot-indexed copy GS?
  for (int i = tid; i < R; i += 512) {
    const int a = i < m ? A_new[i] : 0;
    const int sl = i < m ? fz.slot_of[a] : 0;
    sA[i] = a;
    sS[i] = sl;
    off_gs |= (sl < 0 || sl >= fz.CS) ? 1 : 0;
  }
  const bool all_gs = !__syncthreads_or(off_gs) && fz.GS != nullptr;  // uniform
  CGP(0);
  const int c0 = wave * nc;
  // right-hand side, warm start (previous coefficients of the columns that stay; a column that enters starts from
  // its one-variable update d_j / (G_jj + ridge)) and the diagonal: independent loads, issued ahead of the gather
  double q0[RPT], x0[RPT], gd0[RPT];
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int row = lane + 64 * r;
    const bool own = wave == 0 && row < m;
    const int a = own ? sA[row] : 0, sl = own ? sS[row] : -1;
    const double qv = own ? rhs[a] : 0.0, bv = own ? fz.beta_dense[a] : 0.0;
    const double dv = (own && fz.d != nullptr) ? fz.d[a] : 0.0;
    const double gv = sl >= 0 ? fz.G[(size_t)sl * fz.p + a] : 0.0;
    q0[r] = qv;
    gd0[r] = gv;
    x0[r] = (own && bv == 0.0 && fz.d != nullptr && sl >= 0) ? dv / (gv + ridge) : bv;
  }
  double g[RPT][NCW];
  {
    // every index this thread needs first (LDS), then all the loads back to back: an index read or a branch between
    // two loads would put an LDS / branch latency in front of every one of them
    int srr[RPT], saa[RPT], slv[NCW];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int row = lane + 64 * r;
      srr[r] = row < m ? sS[row] : -1;
      saa[r] = row < m ? sA[row] : -1;
    }
#pragma unroll
    for (int c = 0; c < NCW; c++) slv[c] = sS[min(c0 + c, R - 1)];
    if (all_gs) {
      // the usual case: every column inside the slot-indexed copy GS.  One base per column, one index per row, and no
      // zeroing behind the loads (a value that is only needed under a uniform condition gets its load sunk into a
      // branch, with a wait behind every load -- 13 us of gather instead of 4): columns outside the system read a word
      // that holds 0.0, rows outside it read some finite Gram entry and are dropped from the products (matvec_pv)
      int rs_[RPT];
#pragma unroll
      for (int r = 0; r < RPT; r++) rs_[r] = max(srr[r], 0);
#pragma unroll
      for (int c = 0; c < NCW; c++) {
        const bool cok = c < nc && c0 + c < m;
        const double *colb = cok ? fz.GS + (size_t)slv[c] * fz.CS : fz.zero;  // wave-uniform
#pragma unroll
        for (int r = 0; r < RPT; r++) g[r][c] = colb[cok ? rs_[r] : 0];
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCW; c++) {
        const int col = c0 + c;
        const bool cok = c < nc && col < m;
        // (kept in a vector register: with readfirstlane on these NCW uniform values the kernel had 210 scalar-register
        // spills and that build produced wrong matrices)
        const int sl = cok ? slv[c] : 0;  // wave-uniform
        if (cok && sl < 0) fz.ctrl->cov_miss = 1;
        const bool cgo = cok && sl >= 0;
        const double *gcol = fz.G + (size_t)(sl < 0 ? 0 : sl) * fz.p;
        const bool small = fz.GS != nullptr && sl < fz.CS;  // uniform
        const double *gsrow = small ? fz.GS + (size_t)(sl < 0 ? 0 : sl) * fz.CS : fz.G;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
          const bool ok = cgo && saa[r] >= 0;
          // both slots inside the slot-indexed copy (L2-resident, neighbouring rows share lines): read it from there
          const bool gs = small && srr[r] >= 0 && srr[r] < fz.CS;
          const double *src = !ok ? fz.G : (gs ? gsrow + srr[r] : gcol + saa[r]);
          const double v = *src;  // unconditional (a valid address either way)
          g[r][c] = ok ? v : 0.0;
        }
      }
    }
  }
  CGP(1);
  // wave-local sum over all R rows (every wave holds every row): fixed order, identical in all waves
  auto wsum = [&](double v) -> double {
    v = row_sum16(v);  // the same value in the 16 lanes of a row
    // (r0 + r1) + (r2 + r3) like the xor-16 / xor-32 exchange, but through the scalar unit instead of two LDS permutes
    auto rl = [](double x, int l) -> double {
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                              __builtin_amdgcn_readlane(__double2loint(x), l));
    };
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
  };
  // reciprocal to full precision for the step lengths: hardware estimate + two Newton steps (a division costs more,
  // and alpha / beta only steer the iteration)
  auto frcp = [](double v) -> double {
    double y = __builtin_amdgcn_rcp(v);
    y = fma(fma(-v, y, 1.0), y, y);
    y = fma(fma(-v, y, 1.0), y, y);
    return y;
  };
  auto dot = [&](const double (&a)[RPT], const double (&b)[RPT]) -> double {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < RPT; r++) t = fma(a[r], b[r], t);
    return wsum(t);
  };
  // y = G v for the vector in pv (visible to the block)
  auto matvec_pv = [&](double (&y)[RPT]) {
    double acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) acc[r] = 0.0;
    // the broadcast reads of v_j in groups of 8, all of a group issued before its products; no branch per column (the
    // matrix entries of columns >= nc are zero, and c0 + c < R because nc <= NCW)
#pragma unroll
    for (int cb = 0; cb < NCW; cb += 8) {
      double vj[8];
#pragma unroll
      for (int c = 0; c < 8; c++)
        if (cb + c < NCW) vj[c] = pv[c0 + cb + c];
#pragma unroll
      for (int c = 0; c < 8; c++)
        if (cb + c < NCW) {
#pragma unroll
          for (int r = 0; r < RPT; r++) acc[r] = fma(g[r][cb + c], vj[c], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) part[pbuf][wave][lane + 64 * r] = acc[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      double t = part[pbuf][0][lane + 64 * r];
#pragma unroll
      for (int w = 1; w < 8; w++) t += part[pbuf][w][lane + 64 * r];
      y[r] = lane + 64 * r < m ? t : 0.0;  // (rows outside the system may hold any finite matrix entries)
    }
    pbuf ^= 1;
  };
  // y = G v for a vector every wave holds in registers (the search direction): through the wave's own LDS copy
  auto matvec = [&](const double (&v)[RPT], double (&y)[RPT]) {
    double *mine = pvw[wave];
#pragma unroll
    for (int r = 0; r < RPT; r++) mine[lane + 64 * r] = v[r];
    double acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) acc[r] = 0.0;
#pragma unroll
    for (int cb = 0; cb < NCW; cb += 8) {
      double vj[8];
#pragma unroll
      for (int c = 0; c < 8; c++)
        if (cb + c < NCW) vj[c] = mine[c0 + cb + c];
#pragma unroll
      for (int c = 0; c < 8; c++)
        if (cb + c < NCW) {
#pragma unroll
          for (int r = 0; r < RPT; r++) acc[r] = fma(g[r][cb + c], vj[c], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) part[pbuf][wave][lane + 64 * r] = acc[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      double t = part[pbuf][0][lane + 64 * r];
#pragma unroll
      for (int w = 1; w < 8; w++) t += part[pbuf][w][lane + 64 * r];
      y[r] = lane + 64 * r < m ? t : 0.0;
    }
    pbuf ^= 1;
  };
  // q, x and the diagonal live in LDS (x is advanced by wave 0 only): the registers belong to the matrix
  double r_t[RPT], p_t[RPT], ap[RPT];
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int row = lane + 64 * r;
      qs[row] = q0[r];
      xs[row] = x0[r];
      gds[row] = gd0[r];
      pv[row] = x0[r];
    }
  }
  __syncthreads();
  CGP(2);
  double qq;
  {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < RPT; r++) t = fma(qs[lane + 64 * r], qs[lane + 64 * r], t);
    qq = wsum(t);
  }
  matvec_pv(ap);  // pv = x
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int row = lane + 64 * r;
    r_t[r] = qs[row] - (ap[r] + ridge * xs[row]);
    p_t[r] = r_t[r];
  }
  double rs = dot(r_t, r_t);
  CGP(3);
  bool ok = false;
  int it = 0;
  CGL_DECL();
  for (int round = 0; round < 3 && !ok; round++) {
    for (; it < maxit && rs > 1e-4 * tol * tol * qq; it++) {  // recurrence residual target: |r| <= tol / 100 |q|
      CGL(0);
      matvec(p_t, ap);
      CGL(1);
#pragma unroll
      for (int r = 0; r < RPT; r++) ap[r] = fma(ridge, p_t[r], ap[r]);
      const double alpha = rs * frcp(dot(p_t, ap));
      CGL(2);
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < RPT; r++) xs[lane + 64 * r] = fma(alpha, p_t[r], xs[lane + 64 * r]);
      }
#pragma unroll
      for (int r = 0; r < RPT; r++) r_t[r] = fma(-alpha, ap[r], r_t[r]);
      // (|r|^2 by the update formula |r|^2 - 2 alpha r.Ap + alpha^2 |Ap|^2, which would fold the step's sums into one
      // reduction, was tried: its rounding error is amplified by |r_old|^2 / |r_new|^2 ~ 300 at every step and the
      // recurrence is useless after five steps)
      const double rs_new = dot(r_t, r_t);
      const double bt = rs_new * frcp(rs);
#pragma unroll
      for (int r = 0; r < RPT; r++) p_t[r] = fma(bt, p_t[r], r_t[r]);
      rs = rs_new;
      CGL(3);
    }
    // the recurrence drifts: accept only on the recomputed residual |q - (G + ridge I) x| <= tol |q|
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < RPT; r++) pv[lane + 64 * r] = xs[lane + 64 * r];
    }
    __syncthreads();
    matvec_pv(ap);
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int row = lane + 64 * r;
      r_t[r] = qs[row] - (ap[r] + ridge * xs[row]);
      p_t[r] = r_t[r];
    }
    rs = dot(r_t, r_t);
    ok = rs <= tol * tol * qq;  // also catches NaN (singular / indefinite matrix): the comparison fails
    if (it >= maxit) break;
  }
  CGL_FLUSH();
  CGP(4);
#ifdef BESSX_CG_PROFILE
  if (threadIdx.x == 0) atomicAdd(&g_cg_prof[14], (unsigned long long)it);
#endif
  if (!ok) {
    if (tid == 0) {  // park the fit: the host issues the Cholesky kernel for this slot
      fz.ctrl->cov_stall = 2;
      fz.ctrl->l = -1 - fz.ctrl->l;
    }
    return;
  }
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < RPT; r++)
      if (lane + 64 * r < m) sol[lane + 64 * r] = xs[lane + 64 * r];
  }
  if (wave == 0) {
    // loss terms, as in k_cg: |y - X b|^2 = y.y - b.(q + rho) - ridge |b|^2 with rho the residual just recomputed
    double t1 = 0.0, t2 = 0.0, t3 = 0.0, gd = 0.0;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int row = lane + 64 * r;
      const double x = xs[row], qr = qs[row] + r_t[r];
      t1 = fma(x, qr, t1);
      t2 = fma(x, x, t2);
      t3 = fma(fabs(x), fabs(qr), t3);
      gd = fmax(gd, gds[row]);
    }
    const double a1 = wsum(t1), a2 = wsum(t2), a3 = wsum(t3);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) gd = fmax(gd, __shfl_xor(gd, o));
    if (tid == 0) {
      const double tr = fz.yy - a1 - ridge * a2;
      fz.ctrl->sse_dot = a1;
      fz.ctrl->sse_nrm = a2;
      fz.ctrl->sse_valid =
          (tr > 1e-6 * fz.yy && 4e-16 * (a3 + (ridge + gd) * a2 + fz.yy) <= 1e-10 * tr) ? 1 : 0;
    }
  }
  __syncthreads();
  CGP(5);
  commit_body(fz.ctrl, slot, fz.T0, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
              fz.hist_coef0, fz.hist_stride, &same_any_sh, fz.inA);
  CGP(6);
}

template <int RPT, int NCW>
__global__ void __launch_bounds__(512) k_cgr(int m, int nc, double ridge, const double *__restrict__ rhs,
                                             const int *__restrict__ A_new, double *__restrict__ sol,
                                             const FitCtrl *__restrict__ ctrl, int slot, const CholFuse fz, int maxit,
                                             const double tol) {
  KT(2);
  cgr_body<RPT, NCW>(m, nc, ridge, rhs, A_new, sol, ctrl, slot, fz, maxit, tol);
  if (fz.pub.on) {  // last kernel of a batch of slots: publish (or snapshot) the result block, whatever the body did
    __syncthreads();
    if (fz.pub.on == 2) {
      if (fz.ctrl->snap_seq != fz.pub.seq) snapshot_body(fz.pub);  // (else the selection kernel has taken it already)
    } else {
      publish_body(fz.pub);
    }
  }
}

// One PDAS iteration of the covariance form behind its GEMV in ONE launch: the selection (topk_body on 8 waves: the
// chained-fit prologue, the arg-max / repeated-set shortcuts or the full search, the cache lookup) and then, in the
// same workgroup, the gather + conjugate-gradient solve + commit of k_cgr.  The two phases talk through the control
// block exactly as the two launches did (every gate is re-read from memory after the barrier), so the results are
// those of k_topk followed by k_cgr -- without the second launch, its fall-through when the selection has already
// settled the slot, and the boundary between them (tools/ktrace.py: 581 + 581 launches per 200-candidate path).
// blockIdx.x == 1 (only when nd.pub.on): the deferred publication of the parent fit, as in k_topk.
// (Round 2 also built a solve from a maintained inverse carried between active sets by bordering updates: correct,
// 31 us per solve against 30 -- DESIGN.md 3a -- and removed in round 3.)
template <int EB, int RPT, int NCW>
__global__ void __launch_bounds__(512) k_sel_cgr(const double *__restrict__ score, int len, int k, int *out,
                                                 const FitCtrl *ctrl, int slot, const TopkNeed nd, int nc, double ridge,
                                                 const double *__restrict__ rhs, double *sol, const CholFuse fz,
                                                 int maxit, const double tol) {
  KT(14);
  if (nd.pub.on && blockIdx.x == 1) {
    publish_body(nd.pub);
    return;
  }
#ifdef BESSX_KTRACE
  const unsigned long long kt0_ = wall_clock64();
#endif
  {
    // The commonest short launch -- the selection that only confirms the active set and ends the fit -- first and in
    // one piece: a launch lands on a compute unit whose instruction cache has none of this kernel, and the same steps
    // spread over the selection, the solve body and the tail were a string of instruction-fetch misses
    // (block 0: 4.0 -> 3.6 us, tools/ktrace.py).
    if (!nd.cont_on && nd.commit_on && nd.slot_of != nullptr && !ctrl->done && ctrl->l == slot - 1 && ctrl->l >= 1 &&
        nd.ctrl->fast_same) {  // uniform
      if (repeated_set_body<512>(nd, k, out, slot)) {
        if (fz.pub.on) {  // last kernel of a batch of slots, as at the end of this kernel
          __syncthreads();
          if (fz.pub.on == 2) {
            if (fz.ctrl->snap_seq != fz.pub.seq) snapshot_body(fz.pub);
          } else {
            publish_body(fz.pub);
          }
        }
#ifdef BESSX_KTRACE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
          atomicAdd(&g_phase[8], wall_clock64() - kt0_);
          atomicAdd(&g_phase[14], 1ull);
        }
        KT(15);
#endif
        return;
      }
    }
  }
  topk_body<EB, 512>(score, nullptr, len, len, k, out, nullptr, ctrl, slot, nullptr, nd);
  __syncthreads();  // the selection's writes (A_new, the control block, a commit) are visible to the whole block
#ifdef BESSX_KTRACE
  const unsigned long long kt1_ = wall_clock64();
  unsigned long long kt2_ = kt1_;
#endif
  cgr_body<RPT, NCW>(k, nc, ridge, rhs, out, sol, ctrl, slot, fz, maxit, tol);
#ifdef BESSX_KTRACE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  kt2_ = wall_clock64();
#endif
  if (fz.pub.on) {  // last kernel of a batch of slots: publish (or snapshot) the result block, whatever the body did
    __syncthreads();
    if (fz.pub.on == 2) {
      if (fz.ctrl->snap_seq != fz.pub.seq) snapshot_body(fz.pub);  // (else the selection phase has taken it already)
    } else {
      publish_body(fz.pub);
    }
  }
#ifdef BESSX_KTRACE
  {  // block 0's time in the selection, the solve body and the tail, by what the launch did
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      const unsigned long long kt3_ = wall_clock64();
      const int cls = ctrl->same_prev ? 0 : 1;
      atomicAdd(&g_phase[8 + 3 * cls], kt1_ - kt0_);
      atomicAdd(&g_phase[9 + 3 * cls], kt2_ - kt1_);
      atomicAdd(&g_phase[10 + 3 * cls], kt3_ - kt2_);
      atomicAdd(&g_phase[14 + cls], 1ull);
    }
  }
  KT(15);  // end of block 0: what follows until the next kernel's start stamp is boundary / idle time
#endif
}

// ------------------------------------------------------------------------------------------
// K7, large systems (m + 1 > 256): blocked right-looking Cholesky on the tile-layout matrix in global memory
// (it stays L2 resident), one small launch per phase of a block column.  Correct for any size; the in-register
// kernel above is the fast path for the BASELINE sizes (k <= 254).  Same augmented-row trick for the right-hand side.
// ------------------------------------------------------------------------------------------

#define BIG_GATE(ctrl, slot, gate_mode)                                              \
  if ((ctrl) != nullptr) {                                                           \
    if ((ctrl)->done || (ctrl)->l != (slot)-1 || (ctrl)->same_prev) return;          \
    if (((gate_mode) == 1 || (gate_mode) == 2) && (ctrl)->irls_done) return;         \
  }

// apply ridge / padding / right-hand-side row to the reduced Gram tiles in place (what k_chol does while loading)
__global__ void __launch_bounds__(256) k_bc_prepare(double *__restrict__ Gt, int m, int mt, double ridge,
                                                    int ridge_skip0, const double *__restrict__ rhs,
                                                    const int *__restrict__ rhs_gather,
                                                    const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int t = blockIdx.x, lane = threadIdx.x >> 2, r = threadIdx.x & 3, mp = mt * 16;
  int I, J;
  tile_of(t, I, J);
  const int row = I * 16 + (lane >> 4) + 4 * r, col = J * 16 + (lane & 15);
  double v = Gt[(size_t)t * 256 + threadIdx.x];
  if (row == col && row < m && !(ridge_skip0 && row == 0)) v += ridge;
  if (rhs != nullptr) {
    if (row >= m || col >= m) v = (row == col) ? 1.0 : 0.0;
    if (row == mp - 1 && col < m) v = rhs[rhs_gather ? rhs_gather[col] : col];
  } else {
    bool rpad = row >= m && row != mp - 1, cpad = col >= m && col != mp - 1;
    if (rpad || cpad) v = (row == col) ? 1.0 : 0.0;
    if (row == mp - 1 && col == mp - 1) v = 1.0;
  }
  Gt[(size_t)t * 256 + threadIdx.x] = v;
}

// block column b: every wave factors tile (b,b) redundantly in registers, wave I-b then substitutes its panel tile
// (I,b) (16 rows, one lane each); the wave of I == b stores the factor and the reciprocal diagonal.
__global__ void __launch_bounds__(64) k_bc_panel(double *__restrict__ Gt, double *__restrict__ rdiag, int mt, int b,
                                                 const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int lane = threadIdx.x, rr = lane & 15, I = b + blockIdx.x;
  const double *D = Gt + tile_id(b, b) * 256;
  double Lr[16], rinv[16];
#pragma unroll
  for (int c = 0; c < 16; c++) Lr[c] = D[tile_elem(rr, c)];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const double d = sqrt(bcast_lane(Lr[j], j));
    rinv[j] = bcast_lane(1.0 / d, 0);
    const double lij = (rr == j) ? d : Lr[j] * rinv[j];
    Lr[j] = lij;
#pragma unroll
    for (int c = j + 1; c < 16; c++) Lr[c] = fma(-lij, bcast_lane(lij, c), Lr[c]);
    if (I == b && lane == j) rdiag[b * 16 + j] = rinv[j];
  }
  double *T = Gt + tile_id(I, b) * 256;
  if (I == b) {
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; c++) T[tile_elem(rr, c)] = Lr[c];
    }
    return;
  }
  double x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = T[tile_elem(rr, j)];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    double sacc = x[j];
#pragma unroll
    for (int t = 0; t < j; t++) sacc = fma(-x[t], bcast_lane(Lr[t], j), sacc);
    x[j] = sacc * rinv[j];
  }
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; j++) T[tile_elem(rr, j)] = x[j];
  }
}

// trailing update of step b: tile (I,J) -= L(I,b) L(J,b)^T for b < J <= I, one wave per tile, 4 fp64 MFMAs
__global__ void __launch_bounds__(64) k_bc_update(double *__restrict__ Gt, int mt, int b,
                                                  const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int lane = threadIdx.x, lc = lane & 15, lq = lane >> 4;
  int Ir, Jr;
  tile_of(blockIdx.x, Ir, Jr);  // enumerate the lower triangle of the trailing (mt-b-1) x (mt-b-1) tile grid
  const int I = b + 1 + Ir, J = b + 1 + Jr;
  const double *A = Gt + tile_id(I, b) * 256, *B = Gt + tile_id(J, b) * 256;
  double *C = Gt + tile_id(I, J) * 256;
  d4 acc = *reinterpret_cast<const d4 *>(C + lane * 4);
#pragma unroll
  for (int k4 = 0; k4 < 4; k4++) {
    double av = -A[tile_elem(lc, k4 * 4 + lq)], bv = B[tile_elem(lc, k4 * 4 + lq)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  *reinterpret_cast<d4 *>(C + lane * 4) = acc;
}

// backward solve, block b: z_b <- Lbb^{-T} z_b, then z_c -= L(b,c)^T z_b for c < b.  One 512-thread block.
// step == mt - 1 first copies y (= row mp-1 of L) into z.
__global__ void __launch_bounds__(512) k_bc_back(const double *__restrict__ Gt, const double *__restrict__ rdiag,
                                                 double *__restrict__ z, int mt, int b, int m,
                                                 double *__restrict__ sol, int *__restrict__ info,
                                                 const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mp = mt * 16;
  if (b == mt - 1) {
    for (int c = tid; c < mp; c += 512) {
      const int J = c >> 4;
      z[c] = (c == mp - 1) ? 0.0 : Gt[tile_id(mt - 1, J) * 256 + tile_elem(15, c & 15)];
    }
    __syncthreads();
  }
  if (wave == 0) {
    const double *D = Gt + tile_id(b, b) * 256;
    const int ci = lane & 15;
    double Lc[16], zi = z[b * 16 + ci];
#pragma unroll
    for (int r = 0; r < 16; r++) Lc[r] = D[tile_elem(r, ci)];
    const double ri = rdiag[b * 16 + ci];
#pragma unroll
    for (int j = 15; j >= 0; j--) {
      double xj = bcast_lane(zi, j) * bcast_lane(ri, j);
      if (b == mt - 1 && j == 15) xj = 0.0;
      zi = (ci == j) ? xj : ((ci < j) ? fma(-Lc[j], xj, zi) : zi);
    }
    if (lane < 16) z[b * 16 + ci] = zi;
  }
  __syncthreads();
  const int lc = lane & 15, lq = lane >> 4;
  for (int c = wave; c < b; c += 8) {
    const d4 t = *reinterpret_cast<const d4 *>(Gt + tile_id(b, c) * 256 + lane * 4);
    const int zb = b * 16 + lq;
    double v = t.x * z[zb] + t.y * z[zb + 4] + t.z * z[zb + 8] + t.w * z[zb + 12];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lq == 0) z[c * 16 + lc] -= v;
  }
  if (b == 0) {
    __syncthreads();
    bool bad = false;
    for (int i = tid; i < m; i += 512) {
      double v = z[i];
      sol[i] = v;
      bad |= !(fabs(v) <= DBL_MAX);
    }
    if (__syncthreads_or(bad) && tid == 0 && info != nullptr) *info = 1;
  }
}

// ------------------------------------------------------------------------------------------
// Algorithm::fit bookkeeping (src/Algorithm.h:141-170) on the device.
// ------------------------------------------------------------------------------------------
// Start of a fit: beta <- beta_init (sparse), A_list.col(0) = 0, l = 0.
__global__ void __launch_bounds__(256) k_fit_begin(FitCtrl *__restrict__ ctrl, int T0, int k_init,
                                                   const int *__restrict__ init_idx,
                                                   const double *__restrict__ init_val, double coef0_init,
                                                   int *__restrict__ A_cur, double *__restrict__ b_cur,
                                                   double *__restrict__ beta_dense, int *__restrict__ hist,
                                                   unsigned char *__restrict__ inA) {
  KT(13);
  // beta_dense (and inA) were zeroed by memset nodes just before this launch
  for (int i = threadIdx.x; i < k_init; i += 256) {
    A_cur[i] = init_idx[i];
    b_cur[i] = init_val[i];
    beta_dense[init_idx[i]] = init_val[i];
    if (inA != nullptr) inA[init_idx[i]] = 1;
  }
  for (int i = threadIdx.x; i < T0; i += 256) hist[i] = 0;
  if (threadIdx.x == 0) {
    ctrl->done = 0;
    ctrl->l = 0;
    ctrl->T0 = T0;
    ctrl->k_cur = k_init;
    ctrl->coef0 = coef0_init;
    ctrl->irls_done = 0;
    ctrl->irls_steps = 0;
    ctrl->info = 0;
    ctrl->same_prev = 0;
    ctrl->d_fresh = 0;
    ctrl->cov_nfill = 0;
    ctrl->cov_stall = 0;
    ctrl->cov_groups = 0;
    ctrl->cov_miss = 0;
    ctrl->cov_nmiss = 0;
    ctrl->sse_valid = 0;
    ctrl->fast_same = 0;
  }
}

// Start of a fit whose initial coefficients ARE the device state left by the previous fit (warm-start chain on
// one row set): nothing to upload, only the loop bookkeeping is reset.
// chained = 1: queued BEHIND the previous fit of a warm-start chain before the host has seen its result; it only
// starts if that fit has ended on a repeated active set with its score-pass sums fresh (exactly the condition under
// which the host would have issued it), otherwise it -- and with it all its slots -- does nothing.
__global__ void __launch_bounds__(256) k_fit_continue(FitCtrl *__restrict__ ctrl, int T0, int *__restrict__ hist,
                                                      int serial, int chained, int parent) {
  KT(8);
  if (chained && !(ctrl->serial == parent && ctrl->done && ctrl->d_fresh && ctrl->l >= 0 && !ctrl->cov_stall &&
                   !ctrl->info))
    return;
  for (int i = threadIdx.x; i < T0; i += 256) hist[i] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctrl->done = 0;
    ctrl->l = 0;
    ctrl->T0 = T0;
    ctrl->irls_done = 0;
    ctrl->irls_steps = 0;
    ctrl->info = 0;
    ctrl->same_prev = 0;
    ctrl->d_fresh = 0;
    ctrl->cov_nfill = 0;
    ctrl->cov_stall = 0;
    ctrl->cov_groups = 0;
    ctrl->cov_miss = 0;
    ctrl->cov_nmiss = 0;
    ctrl->sse_valid = 0;
    ctrl->fast_same = 0;
    ctrl->serial = serial;
  }
}

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

__global__ void __launch_bounds__(256) k_commit(FitCtrl *__restrict__ ctrl, int slot, int T0,
                                                const int *__restrict__ A_new, const double *__restrict__ sol,
                                                int has_intercept, int wait_chain, int *__restrict__ A_cur,
                                                double *__restrict__ b_cur, double *__restrict__ beta_dense,
                                                int *__restrict__ hist, double *__restrict__ hist_beta,
                                                double *__restrict__ hist_coef0, int hist_stride,
                                                unsigned char *__restrict__ inA) {
  if (ctrl->done || ctrl->l != slot - 1) return;
  __shared__ int same_any;
  commit_body(ctrl, slot, T0, A_new, sol, has_intercept, wait_chain, A_cur, b_cur, beta_dense, hist, hist_beta,
              hist_coef0, hist_stride, &same_any, inA);
}

// ------------------------------------------------------------------------------------------
// LM residual for the current beta: e_i = y_i - sum_a X[i,A_a] b_a - coef0 ; r_i = mask_i * e_i,
// plus the two sums of squares needed by LmMetric (src/Metric.h:147 train, :190 CV test):
//   sse[2*blk] = sum mask_i e_i^2, sse[2*blk+1] = sum (1-mask_i) e_i^2   (pad rows excluded).
// when = slot  -> runs iff this slot's k_commit ran (ctrl->l == slot);  when = 0 -> start of fit.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_resid_lm(const double *__restrict__ X, long ld, int n,
                                                  const double *__restrict__ y, const double *__restrict__ mask,
                                                  const FitCtrl *__restrict__ ctrl, int when,
                                                  const int *__restrict__ A_cur, const double *__restrict__ b_cur,
                                                  double *__restrict__ r, double *__restrict__ sse, int mode,
                                                  int kc_given, double c0_given) {
  // mode 0 (streaming score pass: r feeds the next pass): after every commit; same_prev: beta, r, sums unchanged.
  // mode 1 (covariance updates: only the sums of squares of the FINAL coefficients are needed): when the fit ended
  // in this slot; mode 2: the fit ran out of iterations without ending (host issues it after the last slot).
  if (mode == 0 && (ctrl->l != when || (when > 0 && ctrl->same_prev))) return;
  if (mode == 1 && (ctrl->l != when || !ctrl->done)) return;
  if (mode == 2 && (ctrl->l != when || ctrl->done)) return;
  // mode 3: unconditional (host issues it for the rare fit whose loss cannot be taken from the solved system)
  __shared__ d2 part[3][128];
  __shared__ double sm[2][2];
  const int kc = mode == 3 ? kc_given : ctrl->k_cur;  // mode 3: coefficients handed in by the host
  const double c0 = mode == 3 ? c0_given : ctrl->coef0;
  // block b owns rows [256 b, 256 b + 256) (ld is a multiple of 128 rows): 128 row threads x 2 rows, times 4 thread
  // groups that each take every 4th active column; the 4 partial sums are added in group order (fixed tree)
  const int rt = threadIdx.x & 127, g = threadIdx.x >> 7;
  const long i = ((long)blockIdx.x * 128 + rt) * 2;
  d2 acc = d2{0.0, 0.0};
  if (i < ld)
    for (int a = g; a < kc; a += 4) acc += *reinterpret_cast<const d2 *>(X + (size_t)A_cur[a] * ld + i) * b_cur[a];
  if (g > 0) part[g - 1][rt] = acc;
  __syncthreads();
  double s_tr = 0.0, s_te = 0.0;
  if (g == 0 && i < ld) {
    const d2 sx = ((acc + part[0][rt]) + part[1][rt]) + part[2][rt];
    const d2 yv = *reinterpret_cast<const d2 *>(y + i);
    d2 mk = mask ? *reinterpret_cast<const d2 *>(mask + i) : d2{1.0, 1.0};
    const bool in0 = i < n, in1 = i + 1 < n;  // pad rows: y = 0, x = 0, but coef0 must not leak into r
    d2 e = d2{in0 ? yv.x - sx.x - c0 : 0.0, in1 ? yv.y - sx.y - c0 : 0.0};
    if (!in0) mk.x = 0.0;
    if (!in1) mk.y = 0.0;
    *reinterpret_cast<d2 *>(r + i) = mk * e;
    s_tr = mk.x * e.x * e.x + mk.y * e.y * e.y;
    s_te = (in0 ? (1.0 - mk.x) * e.x * e.x : 0.0) + (in1 ? (1.0 - mk.y) * e.y * e.y : 0.0);
  }
  s_tr = wave_sum(s_tr);
  s_te = wave_sum(s_te);
  if (g == 0 && (threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6][0] = s_tr;
    sm[threadIdx.x >> 6][1] = s_te;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sse[2 * blockIdx.x] = sm[0][0] + sm[1][0];
    sse[2 * blockIdx.x + 1] = sm[0][1] + sm[1][1];
  }
}

// ------------------------------------------------------------------------------------------
// Group selection (group size > 1; GroupPdas* with real groups).
// Per group g (columns c0 .. c0+s-1): bd_g = || Phi_g beta_g + Phi_g^{-1} d_g ||^2 / s with Phi_g = sqrtm(M_g),
//   LM:  M_g = 2 lambda I + X_g^T X_g / n_t (src/utilities.cpp:142-151), d = X^T r / n_t - 2 lambda beta
//   GLM: M_g = X_g^T diag(h) X_g + 2 lambda I, d = X^T g - 2 lambda beta   (src/Algorithm.h:1238-1257, 1342-1361)
// k_group_moments forms the s x s blocks (and optionally X_g^T w2) in one pass over the group's columns;
// k_group_score takes the symmetric square root by a Jacobi eigen-decomposition, one thread per group.
// ------------------------------------------------------------------------------------------
constexpr int GRP_MAX = 16;  // largest group size built

template <int S>
__global__ void __launch_bounds__(256) k_group_moments(const double *__restrict__ X, long ld, int n,
                                                       const double *__restrict__ w1, const double *__restrict__ w2,
                                                       const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                       const int *__restrict__ goff, double *__restrict__ mblk,
                                                       double *__restrict__ dcol, int cshift) {
  // cshift: X holds a panel of columns starting at global column cshift (the Cox group branch forms the suffix
  // sums of a panel at a time); mblk / dcol stay indexed by the global group / column
  __shared__ double sm[4];
  const int g = blockIdx.x, s = gsz[g], c0 = gidx[g];
  if (s > S) return;  // (uniform) wider groups: k_group_moments_big
  const int cx = c0 - cshift;
  double acc[S * (S + 1) / 2], dacc[S];
#pragma unroll
  for (int q = 0; q < S * (S + 1) / 2; q++) acc[q] = 0.0;
#pragma unroll
  for (int u = 0; u < S; u++) dacc[u] = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    double xv[S];
#pragma unroll
    for (int u = 0; u < S; u++) xv[u] = u < s ? X[(size_t)(cx + u) * ld + i] : 0.0;
    const double a = w1 ? w1[i] : 1.0, b = w2 ? w2[i] : 0.0;
    int q = 0;
#pragma unroll
    for (int u = 0; u < S; u++) {
      const double xa = xv[u] * a;
      dacc[u] = fma(xv[u], b, dacc[u]);
#pragma unroll
      for (int v = 0; v <= u; v++) {
        acc[q] = fma(xa, xv[v], acc[q]);
        q++;
      }
    }
  }
  int q = 0;
#pragma unroll
  for (int u = 0; u < S; u++) {
    double dv = block_sum_256(dacc[u], sm);
    if (threadIdx.x == 0 && u < s && w2 != nullptr) dcol[c0 + u] = dv;
#pragma unroll
    for (int v = 0; v <= u; v++) {
      double mv = block_sum_256(acc[q++], sm);
      if (threadIdx.x == 0 && u < s) {
        mblk[goff[g] + v * s + u] = mv;
        mblk[goff[g] + u * s + v] = mv;
      }
    }
  }
}

// The sacrifice of one group from its s x s moment block: Phi = sqrtm(block), score = |Phi beta + Phi^-1 d|^2 / s
// (src/Algorithm.h:1112-1123, :1238-1257; Phi / invPhi, src/utilities.cpp:142-177) by a cyclic Jacobi diagonalisation.
// SC = compile-time width (the loops unroll and the s x s arrays live in registers: 25 + 25 doubles at 5 columns) or 0
// = any width up to GRP_MAX with run-time loops (the arrays then sit in scratch memory: 1.9 ms per launch for 2000
// groups of 5 in round 3, 70 % of a grouped LM path).
template <int SC>
__device__ __forceinline__ double group_sacrifice(int s_rt, double *__restrict__ a, double *__restrict__ v,
                                                  const double *__restrict__ bv, const double *__restrict__ dv) {
  const int s = SC > 0 ? SC : s_rt;
  constexpr int UF = SC > 0 ? SC : 1;  // (run-time widths: no unrolling)
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, dg = 0.0;
#pragma unroll UF
    for (int i = 0; i < s; i++)
#pragma unroll UF
      for (int j = 0; j < s; j++) {
        double e = a[j * s + i];
        if (i != j) off += e * e;
        else dg += e * e;
      }
    if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll UF
    for (int i = 0; i < s - 1; i++)
#pragma unroll UF
      for (int j = i + 1; j < s; j++) {
        const double apq = a[j * s + i];
        if (apq == 0.0) continue;
        const double theta = (a[j * s + j] - a[i * s + i]) / (2.0 * apq);
        const double tq = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tq * tq + 1.0), sn = tq * c;
#pragma unroll UF
        for (int k = 0; k < s; k++) {
          const double akp = a[i * s + k], akq = a[j * s + k];
          a[i * s + k] = c * akp - sn * akq;
          a[j * s + k] = sn * akp + c * akq;
        }
#pragma unroll UF
        for (int k = 0; k < s; k++) {
          const double apk = a[k * s + i], aqk = a[k * s + j];
          a[k * s + i] = c * apk - sn * aqk;
          a[k * s + j] = sn * apk + c * aqk;
        }
#pragma unroll UF
        for (int k = 0; k < s; k++) {
          const double vkp = v[i * s + k], vkq = v[j * s + k];
          v[i * s + k] = c * vkp - sn * vkq;
          v[j * s + k] = sn * vkp + c * vkq;
        }
      }
  }
  double t[SC > 0 ? SC : GRP_MAX];
#pragma unroll UF
  for (int i = 0; i < s; i++) t[i] = 0.0;
#pragma unroll UF
  for (int k = 0; k < s; k++) {
    double pb = 0.0, pd = 0.0;
#pragma unroll UF
    for (int j = 0; j < s; j++) {
      pb += v[k * s + j] * bv[j];
      pd += v[k * s + j] * dv[j];
    }
    const double sq = sqrt(a[k * s + k]), coef = sq * pb + pd / sq;
#pragma unroll UF
    for (int i = 0; i < s; i++) t[i] += v[k * s + i] * coef;
  }
  double ss = 0.0;
#pragma unroll UF
  for (int i = 0; i < s; i++) ss += t[i] * t[i];
  return ss / (double)s;
}

template <int SC>
__device__ __forceinline__ double group_score_one(int g, int s_rt, int c0, const int *__restrict__ goff,
                                                  const double *__restrict__ mblk, const double *__restrict__ dcol,
                                                  const double *__restrict__ part, int nrb, int p, int lm, double n_t,
                                                  double lambda, const double *__restrict__ beta_dense) {
  constexpr int SM = SC > 0 ? SC : GRP_MAX;
  constexpr int UF = SC > 0 ? SC : 1;
  const int s = SC > 0 ? SC : s_rt;
  double a[SM * SM], v[SM * SM], dv[SM], bv[SM];
#pragma unroll UF
  for (int u = 0; u < s; u++) {
    double d;
    if (lm) {
      double sacc = 0.0;
      for (int rb = 0; rb < nrb; rb++) sacc += part[(size_t)rb * p + c0 + u];
      d = sacc / n_t;
    } else {
      d = dcol[c0 + u];
    }
    bv[u] = beta_dense[c0 + u];
    dv[u] = d - 2.0 * lambda * bv[u];
#pragma unroll UF
    for (int w = 0; w < s; w++) {
      double m = mblk[goff[g] + w * s + u];
      if (lm) m = m / n_t;
      if (u == w) m += 2.0 * lambda;
      a[w * s + u] = m;
      v[w * s + u] = (u == w) ? 1.0 : 0.0;
    }
  }
  if (s == 1) {
    const double phi = sqrt(a[0]), inv = 1.0 / phi, tt = phi * bv[0] + inv * dv[0];
    return tt * tt;
  }
  return group_sacrifice<SC>(s, a, v, bv, dv);
}

// lm != 0: dcol is taken from the score-pass partials (sum over row blocks / n_t); else dcol holds X^T g already.
__global__ void __launch_bounds__(64) k_group_score(int N, const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                    const int *__restrict__ goff, const double *__restrict__ mblk,
                                                    const double *__restrict__ dcol, const double *__restrict__ part,
                                                    int nrb, int p, int lm, double n_t, double lambda,
                                                    const double *__restrict__ beta_dense,
                                                    const unsigned char *__restrict__ always,
                                                    double *__restrict__ bd, const FitCtrl *__restrict__ ctrl,
                                                    int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;  // (a speculative slot of a fit that has ended)
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= N) return;
  const int s = gsz[g], c0 = gidx[g];
  if (s > GRP_MAX) return;  // wider groups: k_group_score_big
  double res;
#define GS_CASE(S) \
  case S: res = group_score_one<S>(g, s, c0, goff, mblk, dcol, part, nrb, p, lm, n_t, lambda, beta_dense); break
  switch (s) {
    GS_CASE(1);
    GS_CASE(2);
    GS_CASE(3);
    GS_CASE(4);
    GS_CASE(5);
    GS_CASE(6);
    GS_CASE(7);
    GS_CASE(8);
    default: res = group_score_one<0>(g, s, c0, goff, mblk, dcol, part, nrb, p, lm, n_t, lambda, beta_dense);
  }
#undef GS_CASE
  if (always != nullptr && always[g]) res = DBL_MAX;
  bd[g] = res;
}

// ---- groups wider than GRP_MAX columns (any width the session's k x k capacity allows) ------------------------
// Moments: one block per (group, 8 x 8 tile of its s x s block); 64 + 8 register accumulators per thread over the
// rows, fixed-order block sums.  The tile index t enumerates the lower triangle of the ceil(s / 8)^2 tile grid.
constexpr int GB_T = 8;
__global__ void __launch_bounds__(256) k_group_moments_big(const double *__restrict__ X, long ld, int n,
                                                           const double *__restrict__ w1,
                                                           const double *__restrict__ w2,
                                                           const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                           const int *__restrict__ goff, double *__restrict__ mblk,
                                                           double *__restrict__ dcol, int cshift) {
  __shared__ double sm[4];
  const int g = blockIdx.x, s = gsz[g], c0 = gidx[g];
  if (s <= GRP_MAX) return;
  const int nt = (s + GB_T - 1) / GB_T;
  int tu = 0, t = blockIdx.y;
  if (t >= nt * (nt + 1) / 2) return;
  while (t > tu) {  // row tu of the triangle holds tu + 1 tiles
    t -= tu + 1;
    tu++;
  }
  const int tv = t, u0 = tu * GB_T, v0 = tv * GB_T;
  const double *xu = X + (size_t)(c0 - cshift + u0) * ld, *xv = X + (size_t)(c0 - cshift + v0) * ld;
  double acc[GB_T][GB_T], dacc[GB_T];
#pragma unroll
  for (int a = 0; a < GB_T; a++) {
    dacc[a] = 0.0;
#pragma unroll
    for (int b = 0; b < GB_T; b++) acc[a][b] = 0.0;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    double cu[GB_T], cv[GB_T];
#pragma unroll
    for (int a = 0; a < GB_T; a++) {
      cu[a] = u0 + a < s ? xu[(size_t)a * ld + i] : 0.0;
      cv[a] = v0 + a < s ? xv[(size_t)a * ld + i] : 0.0;
    }
    const double wa = w1 ? w1[i] : 1.0, wb = w2 ? w2[i] : 0.0;
#pragma unroll
    for (int a = 0; a < GB_T; a++) {
      const double ua = cu[a] * wa;
      dacc[a] = fma(cu[a], wb, dacc[a]);
#pragma unroll
      for (int b = 0; b < GB_T; b++) acc[a][b] = fma(ua, cv[b], acc[a][b]);
    }
  }
#pragma unroll
  for (int a = 0; a < GB_T; a++) {
    if (tv == 0 && w2 != nullptr) {  // X_g^T w2 once per tile row (uniform)
      const double dv = block_sum_256(dacc[a], sm);
      if (threadIdx.x == 0 && u0 + a < s) dcol[c0 + u0 + a] = dv;
    }
#pragma unroll
    for (int b = 0; b < GB_T; b++) {
      const double mv = block_sum_256(acc[a][b], sm);
      if (threadIdx.x == 0 && u0 + a < s && v0 + b < s) {
        mblk[goff[g] + (v0 + b) * s + (u0 + a)] = mv;
        mblk[goff[g] + (u0 + a) * s + (v0 + b)] = mv;
      }
    }
  }
}

// Score of a wide group WITHOUT the matrix square root: with M = Phi^2 = L L^T (Cholesky),
//   || Phi b + Phi^{-1} d ||^2 = b'Mb + 2 b'd + d'M^{-1}d = || L^T b + L^{-1} d ||^2,
// a sum of squares again (no cancellation).  One block per group: left-looking column Cholesky in a global work
// copy W (L2-resident), forward substitution, then the column sums of L against b.
__global__ void __launch_bounds__(256) k_group_score_big(int N, const int *__restrict__ gidx,
                                                         const int *__restrict__ gsz, const int *__restrict__ goff,
                                                         const double *__restrict__ mblk,
                                                         const double *__restrict__ dcol,
                                                         const double *__restrict__ part, int nrb, int p, int lm,
                                                         double n_t, double lambda,
                                                         const double *__restrict__ beta_dense,
                                                         const unsigned char *__restrict__ always,
                                                         double *__restrict__ work, double *__restrict__ zwork,
                                                         double *__restrict__ bd, const FitCtrl *__restrict__ ctrl,
                                                         int slot) {
  __shared__ double sm[4];
  __shared__ double piv;
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  const int g = blockIdx.x, s = gsz[g], c0 = gidx[g], tid = threadIdx.x;
  if (s <= GRP_MAX) return;
  double *W = work + goff[g];          // s x s, column-major: W[j * s + i] = element (i, j)
  double *z = zwork + c0, *bv = zwork + p + c0;  // right-hand side / solution and beta of this group
  for (int u = tid; u < s; u += 256) {
    double d;
    if (lm) {
      double sacc = 0.0;
      for (int rb = 0; rb < nrb; rb++) sacc += part[(size_t)rb * p + c0 + u];
      d = sacc / n_t;
    } else {
      d = dcol[c0 + u];
    }
    const double b = beta_dense[c0 + u];
    bv[u] = b;
    z[u] = d - 2.0 * lambda * b;
  }
  for (int e = tid; e < s * s; e += 256) {
    const int i = e % s, j = e / s;
    double m = mblk[goff[g] + e];
    if (lm) m = m / n_t;
    if (i == j) m += 2.0 * lambda;
    W[e] = m;
  }
  __syncthreads();
  for (int j = 0; j < s; j++) {
    // column j: W[i][j] -= sum_{k<j} L[i][k] L[j][k] for i >= j (each thread its own rows: no conflicts)
    for (int i = j + tid; i < s; i += 256) {
      double v = W[(size_t)j * s + i];
      for (int k = 0; k < j; k++) v = fma(-W[(size_t)k * s + i], W[(size_t)k * s + j], v);
      W[(size_t)j * s + i] = v;
    }
    __syncthreads();
    if (tid == 0) piv = sqrt(W[(size_t)j * s + j]);
    __syncthreads();
    const double rp = 1.0 / piv;
    for (int i = j + tid; i < s; i += 256) W[(size_t)j * s + i] = (i == j) ? piv : W[(size_t)j * s + i] * rp;
    __syncthreads();
  }
  // forward substitution z <- L^{-1} z (column oriented)
  for (int j = 0; j < s; j++) {
    if (tid == 0) z[j] = z[j] / W[(size_t)j * s + j];
    __syncthreads();
    const double zj = z[j];
    for (int i = j + 1 + tid; i < s; i += 256) z[i] = fma(-W[(size_t)j * s + i], zj, z[i]);
    __syncthreads();
  }
  // t_j = sum_{i>=j} L[i][j] b_i + z_j; result = sum t_j^2 / s
  double acc = 0.0;
  for (int j = tid; j < s; j += 256) {
    double t = z[j];
    for (int i = j; i < s; i++) t = fma(W[(size_t)j * s + i], bv[i], t);
    acc = fma(t, t, acc);
  }
  acc = block_sum_256(acc, sm);
  if (tid == 0) {
    double res = acc / (double)s;
    if (always != nullptr && always[g]) res = DBL_MAX;
    bd[g] = res;
  }
}

// Screening with groups, LM (src/screening.cpp:44-48 on a group: beta = argmin |y - X_g b|, no intercept, no weights):
// with M = X_g^T X_g and d = X_g^T y from the group-moment kernels, beta = M^{-1} d by a Cholesky in a global work copy
// (left-looking, as k_group_score_big) + forward and backward substitution; score = |beta|^2 / s.  One block per
// group, any width.  A singular block (an all-zero column) gives a non-finite coefficient like the reference's QR
// solve, which divides by the zero pivot: ranked first (+inf), never a NaN key.
__global__ void __launch_bounds__(256) k_group_lsq_score(int N, const int *__restrict__ gidx,
                                                         const int *__restrict__ gsz, const int *__restrict__ goff,
                                                         const double *__restrict__ mblk,
                                                         const double *__restrict__ dcol,
                                                         const unsigned char *__restrict__ always,
                                                         double *__restrict__ work, double *__restrict__ zwork,
                                                         double *__restrict__ score) {
  __shared__ double sm[4];
  __shared__ double piv;
  const int g = blockIdx.x, s = gsz[g], c0 = gidx[g], tid = threadIdx.x;
  if (always != nullptr && always[g]) {
    if (tid == 0) score[g] = DBL_MAX;
    return;
  }
  double *W = work + goff[g], *z = zwork + c0;
  for (int u = tid; u < s; u += 256) z[u] = dcol[c0 + u];
  for (int e = tid; e < s * s; e += 256) W[e] = mblk[goff[g] + e];
  __syncthreads();
  for (int j = 0; j < s; j++) {
    for (int i = j + tid; i < s; i += 256) {
      double v = W[(size_t)j * s + i];
      for (int k = 0; k < j; k++) v = fma(-W[(size_t)k * s + i], W[(size_t)k * s + j], v);
      W[(size_t)j * s + i] = v;
    }
    __syncthreads();
    if (tid == 0) {
      // a column that depends exactly on the ones before it (a duplicate inside the group, more columns than rows):
      // its pivot collapses against its own sum of squares.  The reference's column-pivoted QR gives such a column
      // the coefficient 0 (src/screening.cpp:44-48); here it is dropped the same way: unit pivot, no coupling, and a
      // zero right-hand side entry below.  (An all-zero column keeps its 0 / 0: the group then ranks first, as the
      // reference's division by the zero pivot makes it, tests/test_limits_gpu.py.)
      const double d0 = mblk[goff[g] + (size_t)j * s + j], vj = W[(size_t)j * s + j];
      const bool dead = d0 > 0.0 && !(vj > 1e-11 * d0);
      piv = dead ? -1.0 : sqrt(vj);
    }
    __syncthreads();
    const bool dead = piv < 0.0;
    const double rp = dead ? 0.0 : 1.0 / piv;
    // (a dropped column is marked by a negative diagonal entry: its unknown is 0 in both substitutions)
    for (int i = j + tid; i < s; i += 256) W[(size_t)j * s + i] = (i == j) ? (dead ? -1.0 : piv) : W[(size_t)j * s + i] * rp;
    __syncthreads();
  }
  for (int j = 0; j < s; j++) {  // L y = d
    if (tid == 0) z[j] = W[(size_t)j * s + j] < 0.0 ? 0.0 : z[j] / W[(size_t)j * s + j];
    __syncthreads();
    const double zj = z[j];
    for (int i = j + 1 + tid; i < s; i += 256) z[i] = fma(-W[(size_t)j * s + i], zj, z[i]);
    __syncthreads();
  }
  for (int j = s - 1; j >= 0; j--) {  // L^T b = y
    if (tid == 0) z[j] = W[(size_t)j * s + j] < 0.0 ? 0.0 : z[j] / W[(size_t)j * s + j];
    __syncthreads();
    const double zj = z[j];
    for (int i = tid; i < j; i += 256) z[i] = fma(-W[(size_t)i * s + j], zj, z[i]);
    __syncthreads();
  }
  double acc = 0.0;
  for (int u = tid; u < s; u += 256) acc = fma(z[u], z[u], acc);
  acc = block_sum_256(acc, sm);
  if (tid == 0) {
    const double v = acc / (double)s;
    score[g] = (v == v) ? v : HUGE_VAL;
  }
}

// commit of a group-mode iteration: history on the T0 group ids, coefficients on the K expanded columns
__global__ void __launch_bounds__(256) k_commit_group(FitCtrl *__restrict__ ctrl, int slot, int T0,
                                                      const int *__restrict__ G_new, int K,
                                                      const int *__restrict__ cols, const double *__restrict__ sol,
                                                      int has_intercept, int wait_chain, int *__restrict__ A_cur,
                                                      double *__restrict__ b_cur, double *__restrict__ beta_dense,
                                                      int *__restrict__ hist, double *__restrict__ hist_beta,
                                                      double *__restrict__ hist_coef0, int hist_stride) {
  if (ctrl->done || ctrl->l != slot - 1) return;
  if (wait_chain && !ctrl->irls_done) return;
  __shared__ int same_any;
  const int l = slot, kc = ctrl->k_cur;
  if (threadIdx.x == 0) same_any = 0;
  for (int i = threadIdx.x; i < kc; i += 256) beta_dense[A_cur[i]] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += 256) {
    const int a = cols[i];
    const double b = sol[i + (has_intercept ? 1 : 0)];
    A_cur[i] = a;
    b_cur[i] = b;
    beta_dense[a] = b;
    hist_beta[(size_t)l * hist_stride + i] = b;
  }
  for (int i = threadIdx.x; i < T0; i += 256) hist[(size_t)l * hist_stride + i] = G_new[i];
  __syncthreads();
  for (int ll = 0; ll < l; ll++) {
    int diff = 0;
    for (int i = threadIdx.x; i < T0; i += 256) diff |= (hist[(size_t)ll * hist_stride + i] != G_new[i]);
    diff = __syncthreads_or(diff);
    if (!diff && threadIdx.x == 0) same_any = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (has_intercept) ctrl->coef0 = sol[0];
    hist_coef0[l] = ctrl->coef0;
    ctrl->k_cur = K;
    ctrl->l = l;
    ctrl->done = same_any;
    ctrl->d_fresh = 0;
    ctrl->irls_done = 0;
    ctrl->irls_last = ctrl->irls_steps;
    ctrl->irls_steps = 0;
  }
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

// get_A front half: gradient / curvature weights of the CURRENT coefficients, plus the loss sums.
//   logistic (:1223-1235): eta clamp +-30, pr = e/(e+1), g = w (y - pr), h = w pr (1 - pr)
//   Poisson  (:1338-1340): no clamp, g = (y - e) w, h = e w                 (training rows only: x mask)
// stats[2 blk]   = sum over ALL rows of the train_loss summand (src/Metric.h:266-290, :426-440 / poisson.cpp:15-45)
// stats[2 blk+1] = sum over the fold's TEST rows of the CV summand (:338-351 clamp +-25; :489)
template <int FAM>
__global__ void __launch_bounds__(128) k_glm_eta_gh(const double *__restrict__ X, long ld, int n,
                                                    const double *__restrict__ y, const double *__restrict__ w,
                                                    const double *__restrict__ mask,
                                                    const double *__restrict__ logfact,
                                                    const FitCtrl *__restrict__ ctrl, int when,
                                                    const int *__restrict__ A_cur, const double *__restrict__ b_cur,
                                                    double *__restrict__ g, double *__restrict__ h,
                                                    double *__restrict__ stats) {
  if (ctrl->l != when || (when > 0 && ctrl->same_prev)) return;
  const int kc = ctrl->k_cur;
  const double c0 = ctrl->coef0;
  const long i = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  double s_all = 0.0, s_te = 0.0;
  if (i < ld) {
    const d2 sx = lin_pred2(X, ld, i, A_cur, b_cur, kc);
    const d2 yv = *reinterpret_cast<const d2 *>(y + i), wv = *reinterpret_cast<const d2 *>(w + i);
    const d2 mk = mask ? *reinterpret_cast<const d2 *>(mask + i) : d2{1.0, 1.0};
    double gg[2], hh[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bool in = i + q < n;
      const double eta = (q ? sx.y : sx.x) + c0, yy = q ? yv.y : yv.x, ww = q ? wv.y : wv.x, mm = q ? mk.y : mk.x;
      double gq = 0.0, hq = 0.0;
      if (in) {
        if (FAM == 2) {
          double e = exp(clampv(eta, 30.0)), pr = e / (e + 1.0);
          gq = ww * (yy - pr) * mm;
          hq = ww * pr * (1.0 - pr) * mm;
          s_all += ww * (yy * log(pr) + (1.0 - yy) * log(1.0 - pr));
          if (mm == 0.0) {
            double e2 = exp(clampv(eta, 25.0)), p2 = e2 / (e2 + 1.0);
            s_te += ww * (yy * log(p2) + (1.0 - yy) * log(1.0 - p2));
          }
        } else {
          double e = exp(eta);
          gq = (yy - e) * ww * mm;
          hq = e * ww * mm;
          double v = clampv(eta, 30.0), sv = (yy * v - exp(v) - logfact[i + q]) * ww;
          s_all += sv;
          if (mm == 0.0) s_te += sv;
        }
      }
      gg[q] = gq;
      hh[q] = hq;
    }
    *reinterpret_cast<d2 *>(g + i) = d2{gg[0], gg[1]};
    *reinterpret_cast<d2 *>(h + i) = d2{hh[0], hh[1]};
  }
  block_pair_sum_128(s_all, s_te, stats + 2 * blockIdx.x);
}

// IRLS step t, front half: working weights and response at the iterate bcur on the design [1, X_Anew].
//   logistic (:1160-1166 for t = 0, :1177-1194 for t >= 1): Pi = sigma(clamp eta); W = Pi(1-Pi), floored at
//     0.001 only for t >= 1; z = eta + (y - Pi)/W with the UN-clamped eta; ll = sum w [y log Pi + (1-y) log(1-Pi)]
//   Poisson (:1286-1314): t = 0 uses eta, exp(eta) as they are; t >= 1 clamps eta to +-30 and floors
//     exp(eta) at 0.001; W = e w; z = eta + (y - e)/e; ll = sum w (y eta - e)
// Writes Wv = W * w * mask (0 on pad rows), z, and the per-block log-likelihood partial.
template <int FAM>
__global__ void __launch_bounds__(128) k_glm_irls_prep(const double *__restrict__ X, long ld, int n,
                                                       const double *__restrict__ y, const double *__restrict__ w,
                                                       const double *__restrict__ mask,
                                                       const FitCtrl *__restrict__ ctrl, int slot, int t,
                                                       const int *__restrict__ A_new, int T0,
                                                       const double *__restrict__ bcur, double *__restrict__ Wv,
                                                       double *__restrict__ z, double *__restrict__ llpart,
                                                       int wfloor) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev || ctrl->irls_done || ctrl->irls_steps != t) return;
  const long i = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  double ll = 0.0;
  if (i < ld) {
    const d2 sx = lin_pred2(X, ld, i, A_new, bcur + 1, T0);
    const double b0 = bcur[0];
    const d2 yv = *reinterpret_cast<const d2 *>(y + i), wv = *reinterpret_cast<const d2 *>(w + i);
    const d2 mk = mask ? *reinterpret_cast<const d2 *>(mask + i) : d2{1.0, 1.0};
    double Wq[2], zq[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bool in = i + q < n;
      const double yy = q ? yv.y : yv.x, ww = q ? wv.y : wv.x, mm = q ? mk.y : mk.x;
      double eta = (q ? sx.y : sx.x) + b0, Wt = 0.0, zt = 0.0;
      if (in) {
        if (FAM == 2) {
          double e = exp(clampv(eta, 30.0)), Pi = e / (1.0 + e);
          ll += (yy * log(Pi) + (1.0 - yy) * log(1.0 - Pi)) * ww * mm;
          double W = Pi * (1.0 - Pi);
          if (t > 0 && wfloor && W < 0.001) W = 0.001;
          zt = eta + (yy - Pi) / W;
          Wt = W * ww * mm;
        } else {
          double e;
          if (t == 0) {
            e = exp(eta);
          } else {
            eta = clampv(eta, 30.0);
            e = exp(eta);
            if (e < 0.001) e = 0.001;
            ll += (yy * eta - e) * ww * mm;
          }
          zt = eta + (yy - e) / e;
          Wt = e * ww * mm;
        }
      }
      Wq[q] = Wt;
      zq[q] = zt;
    }
    *reinterpret_cast<d2 *>(Wv + i) = d2{Wq[0], Wq[1]};
    *reinterpret_cast<d2 *>(z + i) = d2{zq[0], zq[1]};
  }
  __shared__ double sm[2];
  ll = wave_sum(ll);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ll;
  __syncthreads();
  if (threadIdx.x == 0) llpart[blockIdx.x] = sm[0] + sm[1];
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

__global__ void __launch_bounds__(256) k_glm_irls_check(FitCtrl *__restrict__ ctrl, int slot, int t, int fam,
                                                        const double *__restrict__ llpart, int nblk, int m,
                                                        double *__restrict__ bcur, double *__restrict__ bprev) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev || ctrl->irls_done || ctrl->irls_steps != t) return;
  irls_check_body<256>(ctrl, t, fam, llpart, nblk, m, bcur, bprev);
}

// start of the sub-model fit of one PDAS iteration: logistic starts from zero (:1155), Poisson from
// (coef0, 0) (:1283-1285 after Algorithm::fit zeroed beta_A, :157)
__global__ void __launch_bounds__(256) k_glm_irls_begin(const FitCtrl *__restrict__ ctrl, int slot, int fam, int m,
                                                        double *__restrict__ bcur, double *__restrict__ bprev) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
  for (int i = threadIdx.x; i < m; i += 256) {
    double v = (fam == 3 && i == 0) ? ctrl->coef0 : 0.0;
    bcur[i] = v;
    bprev[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Cox proportional hazards (rows sorted by time, y = status).  GroupPdasCox, src/Algorithm.h:1370-1650;
// loglik_cox, src/coxph.cpp:16-40.  The reference builds risk-set sums with a dense n x n triangular
// matrix (:1386) and two n x p temporaries (:1576-1577); here they are suffix scans.
// ------------------------------------------------------------------------------------------

// Scans over the n rows (suffix: out_i = sum_{j >= i} in_j; prefix otherwise) in two multi-block launches:
// *_tot forms the total of every 1024-element block (256 threads x 4 consecutive elements in scan order), *_apply
// adds the totals of the blocks before it IN BLOCK ORDER (the carry), rescans its block and writes.  Fixed summation
// order, no atomics, no waiting on other blocks: bitwise reproducible.  (A single block walking all n rows took
// 65 us at n = 100 000; this is ~10 us.)
constexpr int SC_T = 256, SC_E = 4, SC_B = SC_T * SC_E;

__device__ __forceinline__ bool cox_scan_gate_closed(const FitCtrl *ctrl, int gate, int slot, int t) {
  if (ctrl == nullptr) return false;
  if (gate == 1) return ctrl->l != slot || (slot > 0 && ctrl->same_prev);  // state pass after commit `slot`
  if (gate == 2)  // Newton step t
    return ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev || ctrl->irls_done || ctrl->irls_steps != t - 1;
  return false;
}

// exclusive offset of this thread's total among the 256 threads of the block (thread order = scan order)
__device__ __forceinline__ double block_excl_256(double t, double *sm /*>=4*/, double *btot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double inc = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double tt = __shfl_up(inc, o);
    if (lane >= o) inc += tt;
  }
  if (lane == 63) sm[wave] = inc;
  __syncthreads();
  double off = 0.0;
  for (int w = 0; w < wave; w++) off += sm[w];
  if (btot != nullptr) *btot = ((sm[0] + sm[1]) + sm[2]) + sm[3];
  __syncthreads();
  return off + inc - t;
}

__global__ void __launch_bounds__(SC_T) k_scan3_tot(const double *__restrict__ in0, const double *__restrict__ in1,
                                                    const double *__restrict__ in2, long n, int suffix, int nvec,
                                                    double *__restrict__ scr, const FitCtrl *__restrict__ ctrl,
                                                    int gate, int slot, int t) {
  if (cox_scan_gate_closed(ctrl, gate, slot, t)) return;
  __shared__ double sm[4];
  const double *in[3] = {in0, in1, in2};
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  for (int v = 0; v < nvec; v++) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < SC_E; q++) {
      const long r = r0 + q;
      if (r < n) s += in[v][suffix ? n - 1 - r : r];
    }
    double bt;
    (void)block_excl_256(s, sm, &bt);
    if (threadIdx.x == 0) scr[(size_t)v * gridDim.x + blockIdx.x] = bt;
  }
}

// recip (optional) receives 1 / out0
__global__ void __launch_bounds__(SC_T) k_scan3_apply(const double *__restrict__ in0, const double *__restrict__ in1,
                                                      const double *__restrict__ in2, double *__restrict__ out0,
                                                      double *__restrict__ out1, double *__restrict__ out2,
                                                      double *__restrict__ recip, long n, int suffix, int nvec,
                                                      const double *__restrict__ scr,
                                                      const FitCtrl *__restrict__ ctrl, int gate, int slot, int t) {
  if (cox_scan_gate_closed(ctrl, gate, slot, t)) return;
  __shared__ double sm[4];
  const double *in[3] = {in0, in1, in2};
  double *out[3] = {out0, out1, out2};
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  for (int v = 0; v < nvec; v++) {
    double carry = 0.0;
    for (int j = 0; j < (int)blockIdx.x; j++) carry += scr[(size_t)v * gridDim.x + j];
    double x[SC_E], tt = 0.0;
#pragma unroll
    for (int q = 0; q < SC_E; q++) {
      const long r = r0 + q;
      x[q] = r < n ? in[v][suffix ? n - 1 - r : r] : 0.0;
      tt += x[q];
    }
    double s = carry + block_excl_256(tt, sm, nullptr);
#pragma unroll
    for (int q = 0; q < SC_E; q++) {
      const long r = r0 + q;
      if (r < n) {
        const long i = suffix ? n - 1 - r : r;
        s += x[q];
        out[v][i] = s;
        // rows after the last training row of a CV fold have an empty risk set: keep their reciprocal finite
        if (v == 0 && recip != nullptr) recip[i] = s != 0.0 ? 1.0 / s : 0.0;
      }
    }
  }
}

// State pass for the CURRENT coefficients: e = exp(clamp(x beta)); TH = w e mask (get_A theta, :1587),
// ET = e (1 - mask) (test rows of a CV fold), EW = w [delta != 0] mask (:1621-1630), WD = w delta mask (:1429).
__global__ void __launch_bounds__(128) k_cox_eta(const double *__restrict__ X, long ld, int n,
                                                 const double *__restrict__ y, const double *__restrict__ w,
                                                 const double *__restrict__ mask, const FitCtrl *__restrict__ ctrl,
                                                 int when, const int *__restrict__ A_cur,
                                                 const double *__restrict__ b_cur, double *__restrict__ E,
                                                 double *__restrict__ TH, double *__restrict__ ET,
                                                 double *__restrict__ EW, double *__restrict__ WD) {
  if (ctrl->l != when || (when > 0 && ctrl->same_prev)) return;
  const long i = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  if (i >= ld) return;
  const d2 sx = lin_pred2(X, ld, i, A_cur, b_cur, ctrl->k_cur);
  const d2 yv = *reinterpret_cast<const d2 *>(y + i), wv = *reinterpret_cast<const d2 *>(w + i);
  const d2 mk = mask ? *reinterpret_cast<const d2 *>(mask + i) : d2{1.0, 1.0};
  double e[2], th[2], et[2], ew[2], wd[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const bool in = i + q < n;
    const double eta = q ? sx.y : sx.x, yy = q ? yv.y : yv.x, ww = q ? wv.y : wv.x, mm = q ? mk.y : mk.x;
    const double ex = in ? exp(clampv(eta, 30.0)) : 0.0;
    e[q] = ex;
    th[q] = ww * ex * mm;
    et[q] = in ? ex * (1.0 - mm) : 0.0;
    ew[q] = (in && yy != 0.0) ? ww * mm : 0.0;
    wd[q] = in ? ww * yy * mm : 0.0;
  }
  *reinterpret_cast<d2 *>(E + i) = d2{e[0], e[1]};
  *reinterpret_cast<d2 *>(TH + i) = d2{th[0], th[1]};
  *reinterpret_cast<d2 *>(ET + i) = d2{et[0], et[1]};
  *reinterpret_cast<d2 *>(EW + i) = d2{ew[0], ew[1]};
  *reinterpret_cast<d2 *>(WD + i) = d2{wd[0], wd[1]};
}

// loss sums: stats[2b] = sum_all w delta log(e / S_all) (CoxMetric::train_loss, src/Metric.h:565-568),
// stats[2b+1] = the same over the fold's test rows with the test-row risk sets (:609)
__global__ void __launch_bounds__(128) k_cox_loss(long ld, int n, const double *__restrict__ y,
                                                  const double *__restrict__ w, const double *__restrict__ mask,
                                                  const FitCtrl *__restrict__ ctrl, int when,
                                                  const double *__restrict__ E, const double *__restrict__ SALL,
                                                  const double *__restrict__ STEST, double *__restrict__ stats) {
  if (ctrl->l != when || (when > 0 && ctrl->same_prev)) return;
  const long i0 = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  double s_all = 0.0, s_te = 0.0;
  for (int q = 0; q < 2; q++) {
    long i = i0 + q;
    if (i < n && y[i] != 0.0) {
      double t = w[i] * y[i];
      s_all += t * log(E[i] / SALL[i]);
      if (mask != nullptr && mask[i] == 0.0) s_te += t * log(E[i] / STEST[i]);
    }
  }
  block_pair_sum_128(s_all, s_te, stats + 2 * blockIdx.x);
}

// carries for the second pass: part[rb][j] <- sum_{rb' > rb} part[rb'][j] (both accumulators)
__global__ void __launch_bounds__(256) k_cox_carry(double *__restrict__ part, double *__restrict__ part2, int nrb,
                                                   int p, const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  double a1 = 0.0, a2 = 0.0;
  for (int rb = nrb - 1; rb >= 0; rb--) {
    double t1 = part[(size_t)rb * p + j], t2 = part2[(size_t)rb * p + j];
    part[(size_t)rb * p + j] = a1;
    part2[(size_t)rb * p + j] = a2;
    a1 += t1;
    a2 += t2;
  }
}

// K3: second pass of the Cox score: per-column suffix scans inside one row block with the carry of the
// later blocks, accumulating  l1 = sum_i ew_i (x_ij - a_ij),  l2 = sum_i ew_i (b_ij - a_ij^2)  where
// a_ij = (sum_{i'>=i} theta x)/S0_i, b_ij = (sum_{i'>=i} theta x^2)/S0_i   (src/Algorithm.h:1593-1630).
// A wave owns 64 columns x one row block.  32-row sub-tiles are loaded coalesced (256 contiguous bytes per
// column), transposed through a private LDS tile, and then every lane walks ITS column row by row from the
// bottom, so the scan needs no cross-lane traffic; theta, 1/S0 and ew are wave-uniform per row.
constexpr int CS_ROWS = 32, CS_RS = 33;  // sub-tile rows, padded LDS row stride (doubles)
template <int U>
__global__ void __launch_bounds__(256) k_cox_colscan(const double *__restrict__ X, long ld, int p, int nrb,
                                                     const double *__restrict__ TH, const double *__restrict__ RS0,
                                                     const double *__restrict__ EW, double *__restrict__ part,
                                                     double *__restrict__ part2, const FitCtrl *__restrict__ ctrl,
                                                     int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  __shared__ double tile[4][64 * CS_RS];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wid = (long)blockIdx.x * 4 + wv;
  const int ncg = (p + 63) / 64;
  const long cg = wid / nrb;
  const int rb = (int)(wid - cg * nrb);
  if (cg >= ncg) return;
  const int j0 = (int)cg * 64;
  const int jmine = min(j0 + lane, p - 1);
  double a1 = part[(size_t)rb * p + jmine], a2 = part2[(size_t)rb * p + jmine];
  double l1 = 0.0, l2 = 0.0;
  constexpr int NSUB = 128 * U / CS_ROWS;
  const long rbase = (long)rb * (128 * U);
  const int c4 = lane >> 4, seg = lane & 15;
  d2 nxt[16];
  auto load_sub = [&](int sub) {
    const long r0 = rbase + (long)sub * CS_ROWS + 2 * seg;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      int j = min(j0 + 4 * it + c4, p - 1);
      nxt[it] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(X + (size_t)j * ld + r0));
    }
  };
  load_sub(NSUB - 1);
  for (int sub = NSUB - 1; sub >= 0; sub--) {
    // registers -> private LDS tile [column][row]
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int o = (4 * it + c4) * CS_RS + 2 * seg;
      tile[wv][o] = nxt[it].x;
      tile[wv][o + 1] = nxt[it].y;
    }
    if (sub > 0) load_sub(sub - 1);  // next tile's loads fly while this one is scanned
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const long r0 = rbase + (long)sub * CS_ROWS;
    for (int r = CS_ROWS - 1; r >= 0; r--) {
      const double th = TH[r0 + r], ew = EW[r0 + r];  // wave-uniform
      const double x = tile[wv][lane * CS_RS + r];
      const double t = th * x;
      a1 += t;
      a2 += x * t;
      if (ew != 0.0) {
        const double rs = RS0[r0 + r];
        const double q1 = a1 * rs;
        l1 += (x - q1) * ew;
        l2 += (a2 * rs - q1 * q1) * ew;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (j0 + lane < p) {
    part[(size_t)rb * p + j0 + lane] = l1;
    part2[(size_t)rb * p + j0 + lane] = l2;
  }
}

// ---- K3, one-pass form of the Cox score --------------------------------------------------------------------
// The same sums with the order of summation exchanged so that X is read ONCE and no carry has to be known while it
// is read.  With S1_j(i) = sum_{l>=i} theta_l x_lj, rs = 1/S0, ew as above:
//   sum_i ew_i a_ij           = sum_l u_l x_lj,      u_l = theta_l C1_l,  C1_l = sum_{i<=l} ew_i rs_i   (prefix scan)
//   sum_i ew_i S2_j(i) rs_i   = sum_l u_l x_lj^2
//   l1 sum = sum_l x_lj (ew_l - u_l),      l2 sum = sum_l u_l x_lj^2 - Q_j,   Q_j = sum_i c2_i S1_j(i)^2, c2 = ew rs^2
// and inside row block b, S1_j(i) = loc_j(i) + car_j(b) (suffix sum within the block + total of the later blocks):
//   Q_j = sum_b [ P2_j(b) + 2 car_j(b) P1_j(b) + car_j(b)^2 P0(b) ],
//   P2 = sum_{i in b} c2_i loc_j(i)^2,  P1 = sum c2_i loc_j(i),  P0 = sum c2_i,  car_j(b) = sum_{b' > b} T_j(b').
// k_cox_uv prepares u, v = ew - u, c2 (n-vectors, after the S0 scan and the prefix scan of ew rs); k_cox_score1p
// walks every column of a row block bottom-up exactly like k_cox_colscan and leaves T, P1, P2 and the two plain
// sums per (block, column) plus P0 per block; k_cox_score_1p folds the blocks (carry in block order).
__global__ void __launch_bounds__(256) k_cox_c1(long ld, const double *__restrict__ EW, const double *__restrict__ RS0,
                                                double *__restrict__ C1in, const FitCtrl *__restrict__ ctrl,
                                                int when) {
  if (ctrl->l != when || (when > 0 && ctrl->same_prev)) return;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  const double ew = EW[i];
  C1in[i] = ew != 0.0 ? ew * RS0[i] : 0.0;
}

__global__ void __launch_bounds__(256) k_cox_uv(long ld, const double *__restrict__ EW, const double *__restrict__ RS0,
                                                const double *__restrict__ TH, const double *__restrict__ C1,
                                                double *__restrict__ CU, double *__restrict__ CV,
                                                double *__restrict__ C2, const FitCtrl *__restrict__ ctrl, int when) {
  if (ctrl->l != when || (when > 0 && ctrl->same_prev)) return;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  const double ew = EW[i], th = TH[i];
  const double u = th != 0.0 ? th * C1[i] : 0.0;
  const double rs = ew != 0.0 ? RS0[i] : 0.0;
  CU[i] = u;
  CV[i] = ew - u;
  C2[i] = ew * rs * rs;
}

// out: 5 arrays of nrb x p (T, P1, P2, sum x v, sum u x^2) followed by P0[nrb]
// MAP: which (column group, row block) a wave takes.  0: consecutive waves walk the row blocks of one column group;
// 1: consecutive waves take consecutive column groups of ONE row block -- the four n-vectors of that row block (32 KB)
// are then shared by everything in flight at a time instead of being fetched again by every column group.
// Round 4, full size (tools/cox_score_bench.py, rocprofv3 --pmc FETCH_SIZE): MAP = 1 fetches 31.5 GB per 32 GB pass
// where MAP = 0 fetches 35.2 GB (the re-read n-vectors), at the SAME 5.1-5.3 ms per pass (0.75-0.79 of 8 TB/s) -- the
// re-reads were wasted traffic, not what holds the kernel; the four vectors interleaved per row (one 32-byte scalar load
// instead of four) measured 0.73-0.78 and were dropped.  MAP = 1 is what the solver runs.
template <int U, int MAP>
__global__ void __launch_bounds__(256) k_cox_score1p(const double *__restrict__ X, long ld, int p, int nrb,
                                                     const double *__restrict__ TH, const double *__restrict__ CU,
                                                     const double *__restrict__ CV, const double *__restrict__ C2,
                                                     double *__restrict__ out, const FitCtrl *__restrict__ ctrl,
                                                     int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  __shared__ double tile[4][64 * CS_RS];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wid = (long)blockIdx.x * 4 + wv;
  const int ncg = (p + 63) / 64;
  const long cg = MAP == 0 ? wid / nrb : wid % ncg;
  const int rb = MAP == 0 ? (int)(wid - cg * nrb) : (int)(wid / ncg);
  if (cg >= ncg || rb >= nrb) return;
  const int j0 = (int)cg * 64;
  double loc = 0.0, g1 = 0.0, g2 = 0.0, p1 = 0.0, p2 = 0.0, p0 = 0.0;
  constexpr int NSUB = 128 * U / CS_ROWS;
  const long rbase = (long)rb * (128 * U);
  const int c4 = lane >> 4, seg = lane & 15;
  d2 nxt[16];
  auto load_sub = [&](int sub) {
    const long r0 = rbase + (long)sub * CS_ROWS + 2 * seg;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      int j = min(j0 + 4 * it + c4, p - 1);
      nxt[it] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(X + (size_t)j * ld + r0));
    }
  };
  load_sub(NSUB - 1);
  for (int sub = NSUB - 1; sub >= 0; sub--) {
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int o = (4 * it + c4) * CS_RS + 2 * seg;
      tile[wv][o] = nxt[it].x;
      tile[wv][o + 1] = nxt[it].y;
    }
    if (sub > 0) load_sub(sub - 1);  // next tile's loads fly while this one is scanned
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const long r0 = rbase + (long)sub * CS_ROWS;
    for (int r = CS_ROWS - 1; r >= 0; r--) {
      // wave-uniform (scalar loads; staging them through LDS measured the same)
      const double th = TH[r0 + r], u = CU[r0 + r], v = CV[r0 + r], c2 = C2[r0 + r];
      const double x = tile[wv][lane * CS_RS + r];
      loc = fma(th, x, loc);
      g1 = fma(x, v, g1);
      g2 = fma(u * x, x, g2);
      const double m = c2 * loc;  // c2 = 0 on rows without an event: no branch needed
      p1 += m;
      p2 = fma(m, loc, p2);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  const size_t plane = (size_t)nrb * p;
  if (j0 + lane < p) {
    const size_t o = (size_t)rb * p + j0 + lane;
    out[o] = loc;
    out[plane + o] = p1;
    out[2 * plane + o] = p2;
    out[3 * plane + o] = g1;
    out[4 * plane + o] = g2;
  }
  if (cg == 0) {  // P0 of this row block, by the wave of its first column group (fixed order)
    for (int r = lane; r < 128 * U; r += 64) p0 += C2[rbase + r];
    p0 = wave_sum(p0);
    if (lane == 0) out[5 * plane + rb] = p0;
  }
}

// Folds the row blocks of k_cox_score1p's sums (carry in block order).  A block = 64 columns x 8 chunks of row blocks
// (one wave per chunk; a thread walks ITS chunk of ITS column from the bottom); inside a chunk the carry is
// car = off + lc with off = the total of the later chunks, so a chunk leaves A = sum [P2 + lc (2 P1 + lc P0)],
// B = sum [2 P1 + 2 lc P0], C = sum P0 and its total T, and Q = sum over chunks (last to first) A + off (B + off C)
// -- fixed order, same for every launch.  (One thread per column over all nrb blocks kept 79 of 256 CUs busy with a
// 196-step loop: 81 us per PDAS iteration at p = 20 000; this form: ~10 us.)
__global__ void __launch_bounds__(512) k_cox_score_1p(const double *__restrict__ part, int nrb, int p,
                                                      const double *__restrict__ beta_dense, double lambda,
                                                      const unsigned char *__restrict__ always,
                                                      double *__restrict__ bd, const FitCtrl *__restrict__ ctrl,
                                                      int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  __shared__ double sh[8][6][64];
  const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const size_t plane = (size_t)nrb * p;
  const int per = (nrb + 7) / 8, lo = min(nrb, c * per), hi = min(nrb, lo + per);
  double lc = 0.0, A = 0.0, B = 0.0, C = 0.0, s1 = 0.0, s2 = 0.0;
  if (j < p) {
    for (int rb = hi - 1; rb >= lo; rb--) {
      const size_t o = (size_t)rb * p + j;
      const double p0 = part[5 * plane + rb], t = part[o], p1 = part[plane + o], p2 = part[2 * plane + o];
      A += p2 + lc * (2.0 * p1 + lc * p0);
      B += 2.0 * (p1 + lc * p0);
      C += p0;
      lc += t;
      s1 += part[3 * plane + o];
      s2 += part[4 * plane + o];
    }
  }
  sh[c][0][lane] = A;
  sh[c][1][lane] = B;
  sh[c][2][lane] = C;
  sh[c][3][lane] = lc;
  sh[c][4][lane] = s1;
  sh[c][5][lane] = s2;
  __syncthreads();
  if (c != 0 || j >= p) return;
  double off = 0.0, q = 0.0;
  s1 = 0.0;
  s2 = 0.0;
  for (int w = 7; w >= 0; w--) {
    q += sh[w][0][lane] + off * (sh[w][1][lane] + off * sh[w][2][lane]);
    off += sh[w][3][lane];
    s1 += sh[w][4][lane];
    s2 += sh[w][5][lane];
  }
  s2 -= q;
  const double b = beta_dense[j];
  const double l1 = -s1 + 2.0 * lambda * b, l2 = s2 + 2.0 * lambda;
  const double d = -l1 / l2;
  double v = fabs(b + d) * sqrt(l2);
  if (always != nullptr && always[j]) v = DBL_MAX;
  bd[j] = v;
}

// Cox sacrifice score (:1629-1634): l1 = -sum + 2 lambda beta, l2 = sum + 2 lambda, bd = |beta - l1/l2| sqrt(l2)
__global__ void __launch_bounds__(256) k_cox_score(const double *__restrict__ part, const double *__restrict__ part2,
                                                   int nrb, int p, const double *__restrict__ beta_dense,
                                                   double lambda, const unsigned char *__restrict__ always,
                                                   double *__restrict__ bd, const FitCtrl *__restrict__ ctrl,
                                                   int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  double s1 = 0.0, s2 = 0.0;
  for (int rb = 0; rb < nrb; rb++) {
    s1 += part[(size_t)rb * p + j];
    s2 += part2[(size_t)rb * p + j];
  }
  const double b = beta_dense[j];
  const double l1 = -s1 + 2.0 * lambda * b, l2 = s2 + 2.0 * lambda;
  const double d = -l1 / l2;
  double v = fabs(b + d) * sqrt(l2);
  if (always != nullptr && always[j]) v = DBL_MAX;
  bd[j] = v;
}

// ---- Newton iteration of the restricted fit (:1377-1490) ------------------------------------
#define COX_NEWTON_GATE(ctrl, slot, t) \
  ((ctrl)->done || (ctrl)->l != (slot)-1 || (ctrl)->same_prev || (ctrl)->irls_done || (ctrl)->irls_steps != (t)-1)

__global__ void __launch_bounds__(256) k_cox_newton_begin(FitCtrl *__restrict__ ctrl, int slot, int k, int mp,
                                                          double *__restrict__ b0, int *__restrict__ idcols) {
  if (ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev) return;
  for (int i = threadIdx.x; i < k; i += 256) b0[i] = 0.0;
  for (int i = threadIdx.x; i < mp; i += 256) idcols[i] = i < k ? i : -1;  // Gram columns of M, zero padding
  if (threadIdx.x == 0) {
    ctrl->ll0 = 1e5;  // :1393
    ctrl->ls_m = 0;
  }
}

// eta0 = X_A b0, theta = exp(clamp eta0) on the training rows (no weights here, :1415-1423)
__global__ void __launch_bounds__(128) k_cox_fit_eta(const double *__restrict__ X, long ld, int n,
                                                     const double *__restrict__ mask,
                                                     const FitCtrl *__restrict__ ctrl, int slot, int t,
                                                     const int *__restrict__ A_new, int k,
                                                     const double *__restrict__ b0, double *__restrict__ ETA0,
                                                     double *__restrict__ THF, double clampc) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  const long i = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  if (i >= ld) return;
  const d2 sx = lin_pred2(X, ld, i, A_new, b0, k);
  const d2 mk = mask ? *reinterpret_cast<const d2 *>(mask + i) : d2{1.0, 1.0};
  *reinterpret_cast<d2 *>(ETA0 + i) = sx;
  *reinterpret_cast<d2 *>(THF + i) =
      d2{i < n ? exp(clampv(sx.x, clampc)) * mk.x : 0.0, i + 1 < n ? exp(clampv(sx.y, clampc)) * mk.y : 0.0};
}

// C_i = prefix sum of w delta / S0 ;  VG = w delta - theta C (so that g = X_A^T VG, :1429) ; WG1 = theta C
// (weights of the first Hessian Gram).  Same two-launch scan; the apply kernel covers the pad rows too (zeros).
__global__ void __launch_bounds__(SC_T) k_cox_cscan_tot(const double *__restrict__ WD, const double *__restrict__ RS0F,
                                                        long n, double *__restrict__ scr,
                                                        const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = r0 + q;
    if (i < n && WD[i] != 0.0) s += WD[i] * RS0F[i];
  }
  double bt;
  (void)block_excl_256(s, sm, &bt);
  if (threadIdx.x == 0) scr[blockIdx.x] = bt;
}

__global__ void __launch_bounds__(SC_T) k_cox_cscan_apply(const double *__restrict__ WD,
                                                          const double *__restrict__ RS0F,
                                                          const double *__restrict__ THF, double *__restrict__ VG,
                                                          double *__restrict__ WG1, long n, long ld,
                                                          const double *__restrict__ scr,
                                                          const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double carry = 0.0;
  for (int j = 0; j < (int)blockIdx.x; j++) carry += scr[j];
  double x[SC_E], tt = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = r0 + q;
    x[q] = (i < n && WD[i] != 0.0) ? WD[i] * RS0F[i] : 0.0;
    tt += x[q];
  }
  double s = carry + block_excl_256(tt, sm, nullptr);
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = r0 + q;
    if (i < n) {
      s += x[q];
      const double tc = THF[i] * s;
      VG[i] = WD[i] - tc;
      WG1[i] = tc;
    } else if (i < ld) {
      VG[i] = 0.0;
      WG1[i] = 0.0;
    }
  }
}

// M[:, a] = suffix(theta x_a) / S0 (the n x k matrix S1/S0, :1426-1428) and g_a = x_a . VG + 2 lambda b0_a (:1429).
// Grid (row blocks, active columns); scr holds per column the block totals [a][b] followed by the partial dot
// products [k + a][b].
__global__ void __launch_bounds__(SC_T) k_cox_M_tot(const double *__restrict__ X, long ld, long n,
                                                    const int *__restrict__ A_new, const double *__restrict__ THF,
                                                    const double *__restrict__ VG, double *__restrict__ scr,
                                                    const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (ctrl != nullptr && COX_NEWTON_GATE(ctrl, slot, t)) return;  // ctrl == nullptr: ungated (group branch of get_A)
  __shared__ double sm[4];
  const int a = blockIdx.y, k = gridDim.y, nb = gridDim.x;
  const double *x = X + (size_t)A_new[a] * ld;
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double s = 0.0, gs = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long r = r0 + q;
    if (r < n) {
      const long i = n - 1 - r;
      s += THF[i] * x[i];
      gs += x[i] * VG[i];
    }
  }
  double bt, bg;
  (void)block_excl_256(s, sm, &bt);
  (void)block_excl_256(gs, sm, &bg);
  if (threadIdx.x == 0) {
    scr[(size_t)a * nb + blockIdx.x] = bt;
    scr[(size_t)(k + a) * nb + blockIdx.x] = bg;
  }
}

__global__ void __launch_bounds__(SC_T) k_cox_M_apply(const double *__restrict__ X, long ld, long n,
                                                      const int *__restrict__ A_new, const double *__restrict__ THF,
                                                      const double *__restrict__ RS0F, const double *__restrict__ b0,
                                                      double lambda, const double *__restrict__ scr,
                                                      double *__restrict__ M, double *__restrict__ g,
                                                      const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (ctrl != nullptr && COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const int a = blockIdx.y, k = gridDim.y, nb = gridDim.x;
  const double *x = X + (size_t)A_new[a] * ld;
  double *m = M + (size_t)a * ld;
  const long r0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double carry = 0.0;
  for (int j = 0; j < (int)blockIdx.x; j++) carry += scr[(size_t)a * nb + j];
  double v[SC_E], tt = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long r = r0 + q;
    v[q] = r < n ? THF[n - 1 - r] * x[n - 1 - r] : 0.0;
    tt += v[q];
  }
  double s = carry + block_excl_256(tt, sm, nullptr);
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long r = r0 + q;
    if (r < n) {
      s += v[q];
      m[n - 1 - r] = s * RS0F[n - 1 - r];
    }
  }
  if (blockIdx.x == 0) {
    for (long i = n + threadIdx.x; i < ld; i += SC_T) m[i] = 0.0;
    if (threadIdx.x == 0) {
      double gg = 0.0;
      for (int j = 0; j < nb; j++) gg += scr[(size_t)(k + a) * nb + j];
      g[a] = gg + 2.0 * lambda * b0[a];
    }
  }
}

// ---- Newton step, the n-vector work in three launches (one-pass Hessian form) -------------------------------------
// theta, its suffix sums S0, C = prefix sums of w delta / S0, and the row weights of the two Grams used to take five
// launches (linear predictor, k_scan3_tot / _apply, k_cox_cscan_tot / _apply).  With ONE partition of the rows into
// 1024-row blocks for both scans the block totals of one scan are produced by the kernel that applies the previous one:
//   k_cox_nvecA: eta0 (carried from the line search), theta, block totals of theta
//   k_cox_nvecB: S0 = suffix sums (carry: the totals of the later blocks), 1 / S0, block totals of w delta / S0
//   k_cox_nvecC: C = prefix sums (carry: the totals of the earlier blocks), theta C, w delta - theta C, and the
//                bookkeeping vectors of k_cox_hess
// Fixed summation order (thread-strided partial sums of the block totals, butterflies, the waves in order; inside a
// block the scan of block_excl_256), no atomics.
__device__ __forceinline__ double block_sum_of_totals(const double *__restrict__ tot, int lo, int hi, double *sm4) {
  double s = 0.0;
  for (int j = lo + (int)threadIdx.x; j < hi; j += SC_T) s += tot[j];
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = s;
  __syncthreads();
  const double r = ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(SC_T) k_cox_nvecA(long ld, int n, const double *__restrict__ mask,
                                                    const FitCtrl *__restrict__ ctrl, int slot, int t,
                                                    const double *__restrict__ UD, double *__restrict__ ETA0,
                                                    double *__restrict__ THF, double clampc,
                                                    double *__restrict__ totA) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const long i0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double step = 0.0;
  if (t > 1) {
    const int m = ctrl->ls_m;
    step = m == 1 ? 0.5 : (m == 2 ? 0.25 : (m == 3 ? 0.125 : (m == 4 ? 0.0625 : 0.03125)));
  }
  double tt = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = i0 + q;
    if (i < ld) {
      const double e = t > 1 ? ETA0[i] + UD[i] * step : 0.0;
      const double th = i < n ? exp(clampv(e, clampc)) * (mask ? mask[i] : 1.0) : 0.0;
      ETA0[i] = e;
      THF[i] = th;
      tt += th;
    }
  }
  double bt;
  (void)block_excl_256(tt, sm, &bt);
  if (threadIdx.x == 0) totA[blockIdx.x] = bt;
}

__global__ void __launch_bounds__(SC_T) k_cox_nvecB(long ld, int n, const double *__restrict__ THF,
                                                    const double *__restrict__ WD, const double *__restrict__ totA,
                                                    double *__restrict__ S0F, double *__restrict__ RS0F,
                                                    double *__restrict__ totB, const FitCtrl *__restrict__ ctrl,
                                                    int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const int nb = gridDim.x;
  const double carry = block_sum_of_totals(totA, (int)blockIdx.x + 1, nb, sm);
  // thread order = scan order: the block's rows from the last to the first
  const long top = (long)blockIdx.x * SC_B + SC_B - 1 - (long)threadIdx.x * SC_E;
  double th[SC_E], tt = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = top - q;
    th[q] = i < n ? THF[i] : 0.0;
    tt += th[q];
  }
  double s = carry + block_excl_256(tt, sm, nullptr);
  double xs = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = top - q;
    if (i < n) {
      s += th[q];
      const double r = s != 0.0 ? 1.0 / s : 0.0;  // (rows behind the last training row of a fold: empty risk set)
      S0F[i] = s;
      RS0F[i] = r;
      const double wd = WD[i];
      if (wd != 0.0) xs += wd * r;
    }
  }
  double bt;
  (void)block_excl_256(xs, sm, &bt);
  if (threadIdx.x == 0) totB[blockIdx.x] = bt;
}

__global__ void __launch_bounds__(SC_T) k_cox_nvecC(long ld, int n, const double *__restrict__ THF,
                                                    const double *__restrict__ WD, const double *__restrict__ RS0F,
                                                    const double *__restrict__ totB, double *__restrict__ VG,
                                                    double *__restrict__ WG1, double *__restrict__ RC,
                                                    double *__restrict__ CW, const FitCtrl *__restrict__ ctrl,
                                                    int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const double carry = block_sum_of_totals(totB, 0, (int)blockIdx.x, sm);
  const long i0 = (long)blockIdx.x * SC_B + (long)threadIdx.x * SC_E;
  double x[SC_E], rs[SC_E], tt = 0.0;
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = i0 + q;
    rs[q] = i < n ? RS0F[i] : 0.0;
    x[q] = (i < n && WD[i] != 0.0) ? WD[i] * rs[q] : 0.0;
    tt += x[q];
  }
  double s = carry + block_excl_256(tt, sm, nullptr);
#pragma unroll
  for (int q = 0; q < SC_E; q++) {
    const long i = i0 + q;
    if (i < n) {
      s += x[q];
      const double tc = THF[i] * s, vg = WD[i] - tc;
      VG[i] = vg;
      WG1[i] = tc;
      // the gradient rides in the first Gram of k_cox_hess as the column VG / WG1 (a row with VG != 0 is an event row or
      // lies behind one, so its theta C is positive); CW = w delta / S0^2, the row weights of the second Gram
      RC[i] = tc != 0.0 ? vg / tc : 0.0;
      CW[i] = x[q] * rs[q];
    } else if (i < ld) {
      VG[i] = 0.0;
      WG1[i] = 0.0;
      RC[i] = 0.0;
      CW[i] = 0.0;
    }
  }
}

// ---- Newton step, one-pass Hessian ----------------------------------------------------------------------------------
// -h = X_A^T diag(theta C) X_A - M^T diag(w delta) M with M_i = S1_i / S0_i, S1_i = sum_{l >= i} theta_l x_l (:1458-1470).
// Round 2 materialised M (two scan launches over the n x k active columns, k_cox_M_tot / k_cox_M_apply), then formed
// two Grams from two more reads (X_A, M).  Here ONE kernel reads X_A once: with c_i = w_i delta_i / S0_i^2 the second
// term is sum_i c_i S1_i S1_i^T, and inside a row slab b S1_i = L_i + car_b with L_i the suffix sum over the slab's own
// rows and car_b the total of the later slabs, so
//     sum_{i in b} c_i S1_i S1_i^T = L_b^T diag(c) L_b + car_b p1_b^T + p1_b car_b^T + P0_b car_b car_b^T,
//     p1_b = sum c_i L_i,  P0_b = sum c_i
// -- no carry has to be known while X is read (the same exchange as the one-pass score, k_cox_score1p).  A block owns a
// slab and walks its 64-row chunks bottom-up: chunk into the LDS tile, first Gram on the matrix cores (weights theta C),
// then the tile is turned IN PLACE into the suffix sums (one wave per column, the 64 rows of the chunk in its 64
// lanes: DPP shifts inside the 16-lane rows, v_readlane across them, the running column sum of the later chunks from
// LDS), second Gram on the same tile (weights c).  One spare column (index k) carries the bookkeeping through the
// products: in the first Gram it holds (w delta - theta C) / (theta C), so row k of that Gram is the gradient
// X_A^T (w delta - theta C) (:1429); in the second it holds ones, so row k is p1_b and its diagonal entry P0_b.
// k_cox_car then forms the slab carries, k_cox_hess_reduce sums the slabs, applies the three carry terms and writes
// the tiles k_chol takes plus the gradient.
__device__ __forceinline__ double dpp_row_shl_f64(double v, const int n) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (n) {  // lane i of a 16-lane row receives lane i + n of the same row, 0.0 beyond the row
    case 1: lo = BESSX_DPP32(lo, 0x101); hi = BESSX_DPP32(hi, 0x101); break;
    case 2: lo = BESSX_DPP32(lo, 0x102); hi = BESSX_DPP32(hi, 0x102); break;
    case 4: lo = BESSX_DPP32(lo, 0x104); hi = BESSX_DPP32(hi, 0x104); break;
    default: lo = BESSX_DPP32(lo, 0x108); hi = BESSX_DPP32(hi, 0x108); break;
  }
  return __hiloint2double(hi, lo);
}
// out_l = sum_{l' >= l} v_l' over the 64 lanes of the wave (fixed order)
__device__ __forceinline__ double wave_suffix_scan(double v, int lane) {
  v += dpp_row_shl_f64(v, 1);
  v += dpp_row_shl_f64(v, 2);
  v += dpp_row_shl_f64(v, 4);
  v += dpp_row_shl_f64(v, 8);
  const double t1 = readlane_f64(v, 16), t2 = readlane_f64(v, 32), t3 = readlane_f64(v, 48);
  const int q = lane >> 4;
  const double add = q == 0 ? (t1 + (t2 + t3)) : (q == 1 ? (t2 + t3) : (q == 2 ? t3 : 0.0));
  return v + add;
}

template <int TPW, int NPASS>
__global__ void __launch_bounds__(512) k_cox_hess(const double *__restrict__ X, const double *__restrict__ aux, long ld,
                                                  const int *__restrict__ cols, const double *__restrict__ WG1,
                                                  const double *__restrict__ CW, const double *__restrict__ THF,
                                                  int rows_per_slab, int mt, int k, double *__restrict__ part1,
                                                  double *__restrict__ part2, double *__restrict__ HT, int ntiles,
                                                  const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  constexpr int RB = 64, GL_LD = RB + 2, TPC = RB / 2, NW = 8, CPP = 64 * NW / TPC;
  extern __shared__ double smem[];  // tile [mp][GL_LD], then w1[RB] w2[RB] th[RB] w2c[RB] run[mp]
  const int mp = mt * 16;
  double *w1 = smem + (size_t)mp * GL_LD, *w2 = w1 + RB, *th = w2 + RB, *w2c = th + RB, *run = w2c + RB;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ru = tid % TPC, cb = tid / TPC;
  const int slab = blockIdx.x;
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int nchunk = (int)((r_end - r_begin + RB - 1) / RB);
  for (int i = tid; i < mp; i += 64 * NW) run[i] = 0.0;
  int cidx[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; i++) {
    const int col = i * CPP + cb;
    cidx[i] = col < mp ? cols[col] : INT_MIN;
  }
  d2 st[NPASS], v1 = d2{0.0, 0.0}, v2 = d2{0.0, 0.0}, v3 = d2{0.0, 0.0};
  auto load = [&](int kc) {
    const long r0 = r_begin + (long)kc * RB + 2 * ru;
    const bool in = r0 < r_end;  // slabs end on multiples of 16 rows: a row pair is in or out as a whole
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      st[i] = d2{0.0, 0.0};
      if (in && cidx[i] != INT_MIN) st[i] = *reinterpret_cast<const d2 *>(gram_col(X, aux, ld, cidx[i]) + r0);
    }
    if (tid < TPC) {
      v1 = in ? *reinterpret_cast<const d2 *>(WG1 + r0) : d2{0.0, 0.0};
      v2 = in ? *reinterpret_cast<const d2 *>(CW + r0) : d2{0.0, 0.0};
      v3 = in ? *reinterpret_cast<const d2 *>(THF + r0) : d2{0.0, 0.0};
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int col = i * CPP + cb;
      if (col < mp) *reinterpret_cast<d2 *>(smem + (size_t)col * GL_LD + 2 * ru) = st[i];
    }
    if (tid < TPC) {
      *reinterpret_cast<d2 *>(w1 + 2 * ru) = v1;
      *reinterpret_cast<d2 *>(w2 + 2 * ru) = v2;
      *reinterpret_cast<d2 *>(th + 2 * ru) = v3;
    }
  };
  int tI[TPW], tJ[TPW];
  d4 acc1[TPW], acc2[TPW];
#pragma unroll
  for (int ts = 0; ts < TPW; ts++) {
    const int tt = wv + NW * ts;
    int I = -1, J = -1;
    if (tt < ntiles) tile_of(tt, I, J);
    tI[ts] = __builtin_amdgcn_readfirstlane(I);
    tJ[ts] = __builtin_amdgcn_readfirstlane(J);
    acc1[ts] = d4{0.0, 0.0, 0.0, 0.0};
    acc2[ts] = d4{0.0, 0.0, 0.0, 0.0};
  }
  auto products = [&](d4 (&acc)[TPW], const double *wch, int nsx) {
#pragma unroll
    for (int ts = 0; ts < TPW; ts++) {
      if (tI[ts] >= 0) {  // wave-uniform
        const double *pa = smem + (size_t)(tI[ts] * 16 + c) * GL_LD + 4 * q;
        const double *pb = smem + (size_t)(tJ[ts] * 16 + c) * GL_LD + 4 * q;
#pragma unroll
        for (int sx = 0; sx < RB / 16; sx++) {
          if (sx >= nsx) break;  // block-uniform
          const d2 a0 = *reinterpret_cast<const d2 *>(pa + 16 * sx), a1 = *reinterpret_cast<const d2 *>(pa + 16 * sx + 2);
          const d2 b0 = *reinterpret_cast<const d2 *>(pb + 16 * sx), b1 = *reinterpret_cast<const d2 *>(pb + 16 * sx + 2);
          const d2 w0 = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q);
          const d2 wq = *reinterpret_cast<const d2 *>(wch + 16 * sx + 4 * q + 2);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x * w0.x, b0.x, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y * w0.y, b0.y, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x * wq.x, b1.x, acc[ts], 0, 0, 0);
          acc[ts] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y * wq.y, b1.y, acc[ts], 0, 0, 0);
        }
      }
    }
  };
  if (nchunk > 0) load(nchunk - 1);
  __syncthreads();  // run[] is zero
  for (int kc = nchunk - 1; kc >= 0; kc--) {  // bottom-up: the suffix sums run from the slab's last row
    store();
    if (kc > 0) load(kc - 1);
    __syncthreads();
    products(acc1, w1, RB / 16);
    __syncthreads();
    // the second Gram has non-zero weights on the event rows only (c_i = w_i delta_i / S0_i^2): the suffix sums are
    // written COMPACTED to the top of the tile, event rows first in their order, and the product walks only the
    // 16-row steps they fill (about half of them)
    const double cw = w2[lane];
    const unsigned long long evm = __ballot(cw != 0.0);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(evm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)evm, 0u));
    const int nev = __popcll(evm);  // the same in every wave
    const bool ev = cw != 0.0;
    for (int col = wv; col < k; col += NW) {
      double *tc = smem + (size_t)col * GL_LD;
      const double s = wave_suffix_scan(th[lane] * tc[lane], lane) + run[col];
      if (lane == 0) run[col] = s;
      if (ev) tc[rank] = s;
    }
    if (wv == (k & (NW - 1))) {
      smem[(size_t)k * GL_LD + lane] = 1.0;
      w2c[lane] = 0.0;
      if (ev) w2c[rank] = cw;  // (one wave: in-order LDS, the zero fill lands first)
    }
    __syncthreads();
    products(acc2, w2c, (nev + 15) >> 4);
    __syncthreads();
  }
  double *o1 = part1 + (size_t)slab * ntiles * 256, *o2 = part2 + (size_t)slab * ntiles * 256;
#pragma unroll
  for (int ts = 0; ts < TPW; ts++)
    if (tI[ts] >= 0) {
      *reinterpret_cast<d4 *>(o1 + (size_t)(wv + NW * ts) * 256 + lane * 4) = acc1[ts];
      *reinterpret_cast<d4 *>(o2 + (size_t)(wv + NW * ts) * 256 + lane * 4) = acc2[ts];
    }
  for (int i = tid; i < mp; i += 64 * NW) HT[(size_t)slab * mp + i] = i < k ? run[i] : 0.0;
}

// slab carries car_b = sum_{b' > b} T_b' (T_b = the slab's own column totals of theta x) and q_b = p1_b + P0_b car_b / 2:
// the three carry terms of a slab are car_b q_b^T + q_b car_b^T.  One block per column, one thread per slab (at most
// 256 slabs: cox_hess_slab_rows), the slabs scanned last to first in a fixed order.
__global__ void __launch_bounds__(256) k_cox_car(const double *__restrict__ HT, const double *__restrict__ part2,
                                                 int nslab, int mt, int k, int ntiles, double *__restrict__ CAR,
                                                 double *__restrict__ Q, const FitCtrl *__restrict__ ctrl, int slot,
                                                 int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const int mp = mt * 16, col = blockIdx.x;
  const int b = nslab - 1 - (int)threadIdx.x;  // thread order = scan order: the last slab first
  const size_t e1 = tile_id(k >> 4, col >> 4) * 256 + tile_elem(k & 15, col & 15);
  const size_t e0 = tile_id(k >> 4, k >> 4) * 256 + tile_elem(k & 15, k & 15);
  double tb = 0.0, p1 = 0.0, P0 = 0.0;
  if (b >= 0) {
    const double *pb = part2 + (size_t)b * ntiles * 256;
    tb = HT[(size_t)b * mp + col];
    p1 = col < k ? pb[e1] : 0.0;
    P0 = pb[e0];
  }
  const double car = block_excl_256(tb, sm, nullptr);
  if (b >= 0) {
    CAR[(size_t)b * mp + col] = car;
    Q[(size_t)b * mp + col] = p1 + 0.5 * P0 * car;
  }
}

// Gt = sum_b [G1_b - P2_b - car_b q_b^T - q_b car_b^T] on the k x k block (fixed order: 16 groups of slabs, then the
// groups), g = row k of sum_b G1_b + 2 lambda beta0 (:1429; the sign of the ridge terms as the reference has them).
__global__ void __launch_bounds__(256) k_cox_hess_reduce(const double *__restrict__ part1,
                                                         const double *__restrict__ part2,
                                                         const double *__restrict__ CAR, const double *__restrict__ Q,
                                                         int nslab, int ntiles, int mp, int k, double lambda,
                                                         const double *__restrict__ b0, double *__restrict__ Gt,
                                                         double *__restrict__ g, const FitCtrl *__restrict__ ctrl,
                                                         int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[16][17];
  const int el = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const size_t tot = (size_t)ntiles * 256;
  const size_t e = (size_t)blockIdx.x * 16 + el;
  double s = 0.0;
  int i = -1, j = -1;
  if (e < tot) {
    int I, J;
    tile_of((int)(e >> 8), I, J);
    const int ee = (int)(e & 255), ln = ee >> 2, reg = ee & 3;
    i = 16 * I + (ln >> 4) + 4 * reg;
    j = 16 * J + (ln & 15);
    const bool inner = i < k && j < k, grad = i == k;
    for (int sl = gq; sl < nslab; sl += 16) {
      double v = part1[(size_t)sl * tot + e];
      if (!grad) v -= part2[(size_t)sl * tot + e];
      if (inner) {
        const double *cr = CAR + (size_t)sl * mp, *qr = Q + (size_t)sl * mp;
        v -= cr[i] * qr[j] + qr[i] * cr[j];
      }
      s += v;
    }
  }
  sm[gq][el] = s;
  __syncthreads();
  if (gq == 0 && e < tot) {
    double tsum = sm[0][el];
#pragma unroll
    for (int r = 1; r < 16; r++) tsum += sm[r][el];
    Gt[e] = tsum;
    if (i == k && j < k) g[j] = tsum + 2.0 * lambda * b0[j];
  }
}

__global__ void __launch_bounds__(256) k_tile_sub(double *__restrict__ a, const double *__restrict__ b, long n,
                                                  const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = a[i] - b[i];
}

// Fallback of the Newton solve: the system (G1 - G2 - 2 lambda I) u = g is positive definite unless the ridge term,
// whose sign the reference has as written (src/Algorithm.h:1429, 1471), outweighs the information matrix -- then
// the Cholesky kernel reports a non-positive pivot (info = 1).  The reference solves with LDL^T and does not care;
// this kernel does the same (un-pivoted LDL^T on a dense copy, one workgroup, global memory: rare and small), only
// when info says the Cholesky kernel gave up, and clears info when it succeeds.
__global__ void __launch_bounds__(256) k_ldlt_fallback(const double *__restrict__ Gt, int m, double ridge,
                                                       const double *__restrict__ rhs, double *__restrict__ sol,
                                                       int *__restrict__ info, const FitCtrl *__restrict__ ctrl,
                                                       int slot, int t, double *__restrict__ work) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  if (*info == 0) return;
  __shared__ double col[256], zz[256];
  __shared__ double dj_sh;
  __shared__ int bad_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *A = work;              // m x m, column-major, lower triangle used
  if (tid == 0) bad_sh = 0;
  for (int idx = tid; idx < m * m; idx += 256) {
    const int i = idx % m, j = idx / m;
    if (i >= j) {
      double v = Gt[tile_id(i >> 4, j >> 4) * 256 + tile_elem(i & 15, j & 15)];
      if (i == j) v += ridge;
      A[idx] = v;
    }
  }
  for (int i = tid; i < m; i += 256) zz[i] = rhs[i];
  __syncthreads();
  for (int j = 0; j < m; j++) {
    if (tid == 0) {
      const double d = A[(size_t)j * m + j];
      dj_sh = d;
      if (!(fabs(d) > 0.0) || !isfinite(d)) bad_sh = 1;
    }
    __syncthreads();
    const double dj = dj_sh;
    if (bad_sh) break;  // uniform
    for (int i = j + 1 + tid; i < m; i += 256) {
      const double a = A[(size_t)j * m + i];
      col[i] = a;                       // a_ij = l_ij d_j
      A[(size_t)j * m + i] = a / dj;    // l_ij
    }
    __syncthreads();
    // trailing update of the lower triangle: a_il -= l_ij d_j l_lj = a_ij * (a_lj / d_j)
    for (int l = j + 1 + wave; l < m; l += 4) {
      const double f = col[l] / dj;
      for (int i = l + lane; i < m; i += 64) A[(size_t)l * m + i] -= col[i] * f;
    }
    __syncthreads();
  }
  if (bad_sh) return;  // info stays 1: the host reports BESSX_ERR_NUMERIC
  // forward substitution L z = g, then z / d, then L^T u = z
  for (int j = 0; j < m; j++) {
    const double zj = zz[j];
    for (int i = j + 1 + tid; i < m; i += 256) zz[i] -= A[(size_t)j * m + i] * zj;
    __syncthreads();
  }
  for (int i = tid; i < m; i += 256) zz[i] /= A[(size_t)i * m + i];
  __syncthreads();
  for (int j = m - 1; j >= 0; j--) {
    // u_j = z_j - sum_{i > j} l_ij u_i
    double part = 0.0;
    for (int i = j + 1 + tid; i < m; i += 256) part += A[(size_t)j * m + i] * zz[i];
    part = block_sum_256(part, col);
    if (tid == 0) zz[j] -= part;
    __syncthreads();
  }
  for (int i = tid; i < m; i += 256) sol[i] = zz[i];
  if (tid == 0) *info = 0;
}

// UD = X_A u  (direction of the linear predictor; beta1 = beta0 + 0.5^m u with u = -h^{-1} g, :1473-1474)
__global__ void __launch_bounds__(128) k_cox_dir(const double *__restrict__ X, long ld,
                                                 const FitCtrl *__restrict__ ctrl, int slot, int t,
                                                 const int *__restrict__ A_new, int k, const double *__restrict__ u,
                                                 double *__restrict__ UD) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  const long i = ((long)blockIdx.x * 128 + threadIdx.x) * 2;
  if (i >= ld) return;
  *reinterpret_cast<d2 *>(UD + i) = lin_pred2(X, ld, i, A_new, u, k);
}

// trial point of the step halving: theta1 = exp(clamp(eta0 + 0.5^m UD)) on the training rows
// Step-halving line search of one Newton step (:1474-1481), all five trial points beta0 + 0.5^m u (m = 1..5) at once:
// the partial log-likelihood of trial m is sum_i w_i delta_i log(theta_i / S0_i) with theta = exp(clamp(eta0 + 0.5^m ud))
// and S0 its suffix sum.  Same two-launch scan as above over five vectors that are never stored (theta is recomputed
// in the second launch); k_cox_ls5_check then applies the reference's rule -- m = 1; while (ll0 > ll1 && m < 5) m++ --
// and finishes the Newton step (:1482-1487).  3 launches per Newton step instead of 26.
__device__ __forceinline__ double cox_trial_theta(double eta0, double ud, double mk, int m) {
  const double step = m == 1 ? 0.5 : (m == 2 ? 0.25 : (m == 3 ? 0.125 : (m == 4 ? 0.0625 : 0.03125)));
  return exp(clampv(eta0 + step * ud, 30.0)) * mk;
}

constexpr int LS_E = 2, LS_B = SC_T * LS_E;  // rows per thread / per block of the line-search scans (n / 512 blocks)

__global__ void __launch_bounds__(SC_T) k_cox_ls5_tot(long n, const double *__restrict__ mask,
                                                      const double *__restrict__ ETA0, const double *__restrict__ UD,
                                                      double *__restrict__ scr, const FitCtrl *__restrict__ ctrl,
                                                      int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  const long r0 = (long)blockIdx.x * LS_B + (long)threadIdx.x * LS_E;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < LS_E; q++) {
    const long r = r0 + q;
    if (r < n) {
      const long i = n - 1 - r;
      const double e0 = ETA0[i], ud = UD[i], mk = mask ? mask[i] : 1.0;
#pragma unroll
      for (int m = 1; m <= 5; m++) s[m - 1] += cox_trial_theta(e0, ud, mk, m);
    }
  }
#pragma unroll
  for (int m = 0; m < 5; m++) {
    double bt;
    (void)block_excl_256(s[m], sm, &bt);
    if (threadIdx.x == 0) scr[(size_t)m * gridDim.x + blockIdx.x] = bt;
  }
}

__global__ void __launch_bounds__(SC_T) k_cox_ls5_apply(long n, const double *__restrict__ mask,
                                                        const double *__restrict__ ETA0,
                                                        const double *__restrict__ UD, const double *__restrict__ WD,
                                                        const double *__restrict__ scr, double *__restrict__ llp,
                                                        const FitCtrl *__restrict__ ctrl, int slot, int t) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  __shared__ double cw[5][4];
  const int nb = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * LS_B + (long)threadIdx.x * LS_E;
  // carries: the totals of the blocks before this one, summed by the whole block (fixed order: thread-strided
  // partial sums, wave butterflies, the four waves in order)
  double cs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += SC_T)
#pragma unroll
    for (int m = 0; m < 5; m++) cs[m] += scr[(size_t)m * nb + j];
  double th[5][LS_E], wd[LS_E];
#pragma unroll
  for (int q = 0; q < LS_E; q++) {
    const long r = r0 + q;
    wd[q] = 0.0;
#pragma unroll
    for (int m = 0; m < 5; m++) th[m][q] = 0.0;
    if (r < n) {
      const long i = n - 1 - r;
      const double e0 = ETA0[i], ud = UD[i], mk = mask ? mask[i] : 1.0;
      wd[q] = WD[i];
#pragma unroll
      for (int m = 1; m <= 5; m++) th[m - 1][q] = cox_trial_theta(e0, ud, mk, m);
    }
  }
#pragma unroll
  for (int m = 0; m < 5; m++) {
    cs[m] = wave_sum(cs[m]);
    if (lane == 0) cw[m][wave] = cs[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 5; m++) {
    const double carry = ((cw[m][0] + cw[m][1]) + cw[m][2]) + cw[m][3];
    double tt = 0.0;
#pragma unroll
    for (int q = 0; q < LS_E; q++) tt += th[m][q];
    double sfx = carry + block_excl_256(tt, sm, nullptr);
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < LS_E; q++) {
      sfx += th[m][q];
      if (wd[q] != 0.0) v += wd[q] * log(th[m][q] / sfx);
    }
    double bt;
    (void)block_excl_256(v, sm, &bt);
    if (threadIdx.x == 0) llp[(size_t)m * nb + blockIdx.x] = bt;
  }
}

// line-search decision + end of Newton step t: stop if the relative change is < 1e-5, else beta0 <- beta1
__global__ void __launch_bounds__(256) k_cox_ls5_check(FitCtrl *__restrict__ ctrl, int slot, int t, int k,
                                                       const double *__restrict__ llp, int nblk,
                                                       double *__restrict__ b0, const double *__restrict__ u) {
  if (COX_NEWTON_GATE(ctrl, slot, t)) return;
  __shared__ double sm[4];
  __shared__ double ll1s[5];
  for (int m = 0; m < 5; m++) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += llp[(size_t)m * nblk + b];
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) ll1s[m] = s;
  }
  __syncthreads();
  const double ll0 = ctrl->ll0;
  int m = 1;
  while (ll0 > ll1s[m - 1] && m < 5) m++;
  const double ll1 = ll1s[m - 1];
  const bool conv = fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-5;
  const double step = pow(0.5, (double)m);
  if (!conv)
    for (int i = threadIdx.x; i < k; i += 256) b0[i] = b0[i] + step * u[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    ctrl->ll1 = ll1;
    ctrl->ls_m = m;
    if (!conv) ctrl->ll0 = ll1;
    ctrl->irls_steps = t;
    if (conv || t == 30) ctrl->irls_done = 1;
  }
}

// ------------------------------------------------------------------------------------------
// Screening (SIS), src/screening.cpp:26-105, singleton groups: marginal fit per column on the RAW data.
//   LM:       score_j = (x_j.y / x_j.x_j)^2 from one two-accumulator score pass (k_xtv) -> k_screen_score_lm
//   logistic: logit_fit (src/logistic.cpp:61-157): 2-parameter IRLS per column; one block per column and pass,
//             the block also solves the 2x2 system and applies the convergence rule, so no host round trips.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_screen_score_lm(const double *__restrict__ sxy, const double *__restrict__ sxx,
                                                         int p, const unsigned char *__restrict__ always,
                                                         double *__restrict__ score) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  // an all-zero column: the reference's colPivHouseholderQr (Eigen 3.3.4) counts its zero pivot as non-zero and
  // solve() divides by it (src/screening.cpp:44-48): beta = +-inf, the column ranks FIRST (measured on the compiled
  // reference, tests/test_limits_gpu.py).  0 / 0 here would be a NaN key; +inf states the same rank explicitly.
  const double b = sxx[j] > 0.0 ? sxy[j] / sxx[j] : HUGE_VAL;
  const double v = b * b;
  score[j] = (always != nullptr && always[j]) ? DBL_MAX : ((v == v) ? v : HUGE_VAL);
}

// state per column: st[0..1] = beta0, st[2..3] = beta1, st[4] = ll0; done[j] != 0 once converged.
__global__ void __launch_bounds__(256) k_screen_logit_pass(const double *__restrict__ X, long ld, int n,
                                                           const double *__restrict__ y, const double *__restrict__ w,
                                                           int t, double *__restrict__ state, int *__restrict__ done) {
  const int j = blockIdx.x;
  if (done[j]) return;
  __shared__ double sm[4];
  double *st = state + (size_t)j * 5;
  const double *x = X + (size_t)j * ld;
  const double ba = t == 0 ? st[0] : st[2], bb = t == 0 ? st[1] : st[3];
  double ll = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, t0 = 0.0, t1 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double xi = x[i], yi = y[i], wi = w[i];
    const double eta = ba + xi * bb, e = exp(clampv(eta, 30.0)), Pi = e / (1.0 + e);
    ll += (yi * log(Pi) + (1.0 - yi) * log(1.0 - Pi)) * wi;
    double W = Pi * (1.0 - Pi);
    const double z = eta + (yi - Pi) / W;
    W = W * wi;
    s0 += W;
    s1 += W * xi;
    s2 += (W * xi) * xi;
    t0 += W * z;
    t1 += (W * xi) * z;
  }
  ll = block_sum_256(ll, sm);
  s0 = block_sum_256(s0, sm);
  s1 = block_sum_256(s1, sm);
  s2 = block_sum_256(s2, sm);
  t0 = block_sum_256(t0, sm);
  t1 = block_sum_256(t1, sm);
  if (threadIdx.x == 0) {
    if (t == 0) {
      st[4] = ll;
    } else {
      if (fabs(st[4] - ll) / (0.1 + fabs(ll)) < 1e-6) {
        done[j] = 1;  // result: beta0, the iterate before the last solve
        return;
      }
      st[0] = st[2];
      st[1] = st[3];
      st[4] = ll;
    }
    const double det = s0 * s2 - s1 * s1;
    st[2] = (s2 * t0 - s1 * t1) / det;
    st[3] = (s0 * t1 - s1 * t0) / det;
  }
}

__global__ void __launch_bounds__(256) k_screen_score_logit(const double *__restrict__ state, int p,
                                                            const unsigned char *__restrict__ always,
                                                            double *__restrict__ score) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  const double b = state[(size_t)j * 5 + 1];
  const double v = b * b;  // a degenerate column (singular 2 x 2 system) must not rank first: non-finite -> 0
  score[j] = (always != nullptr && always[j]) ? DBL_MAX : ((v <= DBL_MAX) ? v : 0.0);
}

// Screening with groups, logistic (logit_fit, src/logistic.cpp:60-160, on the columns of one group): IRLS on [1, X_g]
// from zero, no floor on the weights, stop when |ll0 - ll1| / (0.1 + |ll1|) < 1e-6, at most 1 + 30 solves, result =
// the iterate BEFORE the last solve.  One block per group and IRLS step (like k_screen_logit_pass for single columns):
// the block forms X^T W X (lower triangle, intercept first), X^T W z and the log-likelihood of the current iterate in
// registers over the rows, thread 0 applies the stopping rule and solves the (s + 1) x (s + 1) system (LDL^T).
// Groups of at most SGL_MAX columns.  state per group: beta0[SGL_MAX + 1], beta1[SGL_MAX + 1], ll0.
constexpr int SGL_MAX = 8;
constexpr int SGL_ST = 2 * (SGL_MAX + 1) + 1;
__global__ void __launch_bounds__(256) k_screen_logit_group(const double *__restrict__ X, long ld, int n,
                                                            const double *__restrict__ y, const double *__restrict__ w,
                                                            const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                            int t, double *__restrict__ state, int *__restrict__ done) {
  constexpr int M = SGL_MAX + 1, NT = M * (M + 1) / 2;
  const int g = blockIdx.x;
  if (done[g] || gsz[g] > SGL_MAX) return;  // (wider groups: the restricted-fit chain of a sub-session, host side)
  __shared__ double sm[4];
  __shared__ double bsh[M];
  const int s = gsz[g], m = s + 1;
  double *st = state + (size_t)g * SGL_ST;
  const double *x = X + (size_t)gidx[g] * ld;
  if (threadIdx.x < M) bsh[threadIdx.x] = threadIdx.x < m ? (t == 0 ? st[threadIdx.x] : st[M + threadIdx.x]) : 0.0;
  __syncthreads();
  double S[NT], tv[M], ll = 0.0;
#pragma unroll
  for (int a = 0; a < NT; a++) S[a] = 0.0;
#pragma unroll
  for (int a = 0; a < M; a++) tv[a] = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    double v[M];
    v[0] = 1.0;
    double eta = bsh[0];
#pragma unroll
    for (int u = 1; u < M; u++) {
      v[u] = u < m ? x[(size_t)(u - 1) * ld + i] : 0.0;
      eta = fma(v[u], bsh[u], eta);
    }
    const double yi = y[i], wi = w[i];
    const double e = exp(clampv(eta, 30.0)), Pi = e / (1.0 + e);
    ll += (yi * log(Pi) + (1.0 - yi) * log(1.0 - Pi)) * wi;
    double W = Pi * (1.0 - Pi);
    const double z = eta + (yi - Pi) / W;
    W = W * wi;
    int q = 0;
#pragma unroll
    for (int a = 0; a < M; a++) {
      const double wa = W * v[a];
      tv[a] = fma(wa, z, tv[a]);
#pragma unroll
      for (int b = 0; b <= a; b++) {
        S[q] = fma(wa, v[b], S[q]);
        q++;
      }
    }
  }
  ll = block_sum_256(ll, sm);
#pragma unroll
  for (int a = 0; a < NT; a++) S[a] = block_sum_256(S[a], sm);
#pragma unroll
  for (int a = 0; a < M; a++) tv[a] = block_sum_256(tv[a], sm);
  if (threadIdx.x == 0) {
    if (t == 0) {
      st[2 * M] = ll;
    } else {
      if (fabs(st[2 * M] - ll) / (0.1 + fabs(ll)) < 1e-6) {
        done[g] = 1;  // result: beta0, the iterate before the last solve
        return;
      }
      for (int a = 0; a < m; a++) st[a] = st[M + a];
      st[2 * M] = ll;
    }
    // (s + 1) x (s + 1) solve, un-pivoted LDL^T on the lower triangle S[a (a + 1) / 2 + b]
    double L[M][M], D[M], xs[M];
    for (int j = 0; j < m; j++) {
      double dj = S[j * (j + 1) / 2 + j];
      for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k] * D[k];
      D[j] = dj;
      for (int i = j + 1; i < m; i++) {
        double v = S[i * (i + 1) / 2 + j];
        for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
        L[i][j] = v / dj;
      }
    }
    for (int i = 0; i < m; i++) {
      double v = tv[i];
      for (int k = 0; k < i; k++) v -= L[i][k] * xs[k];
      xs[i] = v;
    }
    for (int i = 0; i < m; i++) xs[i] = xs[i] / D[i];
    for (int i = m - 1; i >= 0; i--) {
      double v = xs[i];
      for (int k = i + 1; k < m; k++) v -= L[k][i] * xs[k];
      xs[i] = v;
    }
    for (int a = 0; a < m; a++) st[M + a] = xs[a];
  }
}

__global__ void __launch_bounds__(256) k_screen_score_logit_group(const double *__restrict__ state, int N,
                                                                  const int *__restrict__ gsz,
                                                                  const unsigned char *__restrict__ always,
                                                                  double *__restrict__ score) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  const double *st = state + (size_t)g * SGL_ST;
  const int s = gsz[g];
  if (s > SGL_MAX) return;  // (scored by the host from a sub-session's fit)
  double acc = 0.0;
  for (int u = 1; u <= s; u++) acc += st[u] * st[u];
  const double v = acc / (double)s;  // coef_norm, src/screening.cpp:60
  score[g] = (always != nullptr && always[g]) ? DBL_MAX : ((v <= DBL_MAX) ? v : 0.0);
}

// Cox marginal fit, cox_fit (src/coxph.cpp:97-172) on one column: the whole damped Newton loop of a column runs in
// one block; the risk-set sums are block scans over the rows taken from the last (rows are sorted by time).
__device__ __forceinline__ double wave_scan_incl(double v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v;
}

__device__ double screen_cox_ll(const double *__restrict__ x, const double *__restrict__ st,
                                const double *__restrict__ w, int n, double b, double *sm /*>=8*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double carry = 0.0, s = 0.0;
  for (int base = 0; base < n; base += 256) {
    const int r = base + threadIdx.x, i = n - 1 - r;
    const double e = r < n ? exp(clampv(x[i] * b, 30.0)) : 0.0;
    double c = wave_scan_incl(e);
    if (lane == 63) sm[wv] = c;
    __syncthreads();
    double pre = carry;
    for (int q = 0; q < wv; q++) pre += sm[q];
    carry += ((sm[0] + sm[1]) + sm[2]) + sm[3];
    c += pre;
    if (r < n) s += (log(e / c) * st[i]) * w[i];
    __syncthreads();
  }
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) sm[4] = s;
  __syncthreads();
  s = sm[4];
  __syncthreads();
  return s;
}

// Screening with groups, Cox (cox_fit, src/coxph.cpp:42-108, on the columns of one group): the damped Newton loop of
// k_screen_cox with an s-vector gradient and an s x s information matrix -- one block per group, the risk-set sums
// S0, S1_u, S2_uv as 1 + s + s (s + 1) / 2 block scans over the rows taken from the last (rows are sorted by time).
// Groups of at most SCG_MAX columns.
constexpr int SCG_MAX = 4;
constexpr int SCG_NS = 1 + SCG_MAX + SCG_MAX * (SCG_MAX + 1) / 2;

__device__ double screen_cox_ll_group(const double *__restrict__ x, long ld, int s, const double *__restrict__ st,
                                      const double *__restrict__ w, int n, const double *b, double *sm /*>=8*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double carry = 0.0, acc = 0.0;
  for (int base = 0; base < n; base += 256) {
    const int r = base + threadIdx.x, i = n - 1 - r;
    double eta = 0.0;
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++)
      if (u < s && r < n) eta = fma(x[(size_t)u * ld + i], b[u], eta);
    const double e = r < n ? exp(clampv(eta, 30.0)) : 0.0;
    double c = wave_scan_incl(e);
    if (lane == 63) sm[wv] = c;
    __syncthreads();
    double pre = carry;
    for (int q = 0; q < wv; q++) pre += sm[q];
    carry += ((sm[0] + sm[1]) + sm[2]) + sm[3];
    c += pre;
    if (r < n) acc += (log(e / c) * st[i]) * w[i];
    __syncthreads();
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) sm[4] = acc;
  __syncthreads();
  acc = sm[4];
  __syncthreads();
  return acc;
}

__global__ void __launch_bounds__(256) k_screen_cox_group(const double *__restrict__ X, long ld, int n,
                                                          const double *__restrict__ st, const double *__restrict__ w,
                                                          const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                          const unsigned char *__restrict__ always,
                                                          double *__restrict__ score) {
  const int g = blockIdx.x;
  if (gsz[g] > SCG_MAX) return;  // (wider groups: the Newton chain of a sub-session, host side)
  if (always != nullptr && always[g]) {
    if (threadIdx.x == 0) score[g] = DBL_MAX;
    return;
  }
  __shared__ double sm[4 * SCG_NS + 16];
  __shared__ double dsh[SCG_MAX];
  const int s = gsz[g];
  const double *x = X + (size_t)gidx[g] * ld;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double b0[SCG_MAX], b1[SCG_MAX], ll0 = 1e5;
#pragma unroll
  for (int u = 0; u < SCG_MAX; u++) b0[u] = 0.0;
  for (int l = 1; l <= 30; l++) {
    double carry[SCG_NS], gr[SCG_MAX], H[SCG_MAX * (SCG_MAX + 1) / 2];
#pragma unroll
    for (int k = 0; k < SCG_NS; k++) carry[k] = 0.0;
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++) gr[u] = 0.0;
#pragma unroll
    for (int k = 0; k < SCG_MAX * (SCG_MAX + 1) / 2; k++) H[k] = 0.0;
    for (int base = 0; base < n; base += 256) {
      const int r = base + threadIdx.x, i = n - 1 - r;
      double xi[SCG_MAX], eta = 0.0;
#pragma unroll
      for (int u = 0; u < SCG_MAX; u++) {
        xi[u] = (u < s && r < n) ? x[(size_t)u * ld + i] : 0.0;
        eta = fma(xi[u], b0[u], eta);
      }
      const double th = r < n ? exp(clampv(eta, 50.0)) : 0.0;
      double a[SCG_NS];
      a[0] = th;
      int q = 1 + SCG_MAX;
#pragma unroll
      for (int u = 0; u < SCG_MAX; u++) {
        a[1 + u] = th * xi[u];
#pragma unroll
        for (int v = 0; v <= u; v++) {
          a[q] = (th * xi[u]) * xi[v];
          q++;
        }
      }
#pragma unroll
      for (int k = 0; k < SCG_NS; k++) {
        a[k] = wave_scan_incl(a[k]);
        if (lane == 63) sm[4 * k + wv] = a[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < SCG_NS; k++) {
        double pre = carry[k];
        for (int w4 = 0; w4 < wv; w4++) pre += sm[4 * k + w4];
        carry[k] += ((sm[4 * k] + sm[4 * k + 1]) + sm[4 * k + 2]) + sm[4 * k + 3];
        a[k] += pre;
      }
      if (r < n) {
        const double ws = w[i] * st[i], r0 = 1.0 / a[0];
        double q1[SCG_MAX];
        int qq = 1 + SCG_MAX, hk = 0;
#pragma unroll
        for (int u = 0; u < SCG_MAX; u++) {
          q1[u] = a[1 + u] * r0;
          gr[u] += (xi[u] - q1[u]) * ws;
#pragma unroll
          for (int v = 0; v <= u; v++) {
            H[hk] += (a[qq] * r0 - q1[u] * q1[v]) * ws;
            hk++;
            qq++;
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++) gr[u] = block_sum_256(gr[u], sm);
#pragma unroll
    for (int k = 0; k < SCG_MAX * (SCG_MAX + 1) / 2; k++) H[k] = block_sum_256(H[k], sm);
    if (threadIdx.x == 0) {
      // d = h^{-1} g with h = -H (src/coxph.cpp:85-91): H x = g by an un-pivoted LDL^T, d = -x
      double L[SCG_MAX][SCG_MAX], D[SCG_MAX], xs[SCG_MAX];
      for (int j = 0; j < s; j++) {
        double dj = H[j * (j + 1) / 2 + j];
        for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k] * D[k];
        D[j] = dj;
        for (int i = j + 1; i < s; i++) {
          double v = H[i * (i + 1) / 2 + j];
          for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
          L[i][j] = v / dj;
        }
      }
      for (int i = 0; i < s; i++) {
        double v = gr[i];
        for (int k = 0; k < i; k++) v -= L[i][k] * xs[k];
        xs[i] = v;
      }
      for (int i = 0; i < s; i++) xs[i] = xs[i] / D[i];
      for (int i = s - 1; i >= 0; i--) {
        double v = xs[i];
        for (int k = i + 1; k < s; k++) v -= L[k][i] * xs[k];
        xs[i] = v;
      }
      for (int u = 0; u < SCG_MAX; u++) dsh[u] = u < s ? -xs[u] : 0.0;
    }
    __syncthreads();
    double d[SCG_MAX];
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++) d[u] = dsh[u];
    __syncthreads();
    int m = 1;
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++) b1[u] = b0[u] - 0.5 * d[u];
    double ll1 = screen_cox_ll_group(x, ld, s, st, w, n, b1, sm);
    while (ll0 > ll1 && m < 5) {
      m = m + 1;
      const double f = pow(0.5, (double)m);
#pragma unroll
      for (int u = 0; u < SCG_MAX; u++) b1[u] = b0[u] - f * d[u];
      ll1 = screen_cox_ll_group(x, ld, s, st, w, n, b1, sm);
    }
    if (fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-5) break;
#pragma unroll
    for (int u = 0; u < SCG_MAX; u++) b0[u] = b1[u];
    ll0 = ll1;
  }
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int u = 0; u < s; u++) acc += b0[u] * b0[u];
    const double v = acc / (double)s;
    score[g] = (v <= DBL_MAX) ? v : 0.0;  // non-finite (degenerate group) -> ranks last
  }
}

__global__ void __launch_bounds__(256) k_screen_cox(const double *__restrict__ X, long ld, int n,
                                                    const double *__restrict__ st, const double *__restrict__ w,
                                                    const unsigned char *__restrict__ always,
                                                    double *__restrict__ score) {
  const int j = blockIdx.x;
  if (always != nullptr && always[j]) {
    if (threadIdx.x == 0) score[j] = DBL_MAX;
    return;
  }
  __shared__ double sm[16];
  const double *x = X + (size_t)j * ld;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double b0 = 0.0, ll0 = 1e5;
  for (int l = 1; l <= 30; l++) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, g = 0.0, h = 0.0;
    for (int base = 0; base < n; base += 256) {
      const int r = base + threadIdx.x, i = n - 1 - r;
      const double xi = r < n ? x[i] : 0.0;
      const double th = r < n ? exp(clampv(xi * b0, 50.0)) : 0.0;
      double a0 = wave_scan_incl(th), a1 = wave_scan_incl(th * xi), a2 = wave_scan_incl((th * xi) * xi);
      if (lane == 63) {
        sm[wv] = a0;
        sm[4 + wv] = a1;
        sm[8 + wv] = a2;
      }
      __syncthreads();
      double p0 = c0, p1 = c1, p2 = c2;
      for (int q = 0; q < wv; q++) {
        p0 += sm[q];
        p1 += sm[4 + q];
        p2 += sm[8 + q];
      }
      c0 += ((sm[0] + sm[1]) + sm[2]) + sm[3];
      c1 += ((sm[4] + sm[5]) + sm[6]) + sm[7];
      c2 += ((sm[8] + sm[9]) + sm[10]) + sm[11];
      a0 += p0;
      a1 += p1;
      a2 += p2;
      if (r < n) {
        const double q1 = a1 / a0, ws = w[i] * st[i];
        g += (xi - q1) * ws;
        h += (a2 / a0 - q1 * q1) * ws;
      }
      __syncthreads();
    }
    g = block_sum_256(g, sm);
    h = block_sum_256(h, sm);
    if (threadIdx.x == 0) sm[12] = g / (-h);
    __syncthreads();
    const double d = sm[12];
    __syncthreads();
    int m = 1;
    double b1 = b0 - 0.5 * d;
    double ll1 = screen_cox_ll(x, st, w, n, b1, sm);
    while (ll0 > ll1 && m < 5) {
      m = m + 1;
      b1 = b0 - pow(0.5, (double)m) * d;
      ll1 = screen_cox_ll(x, st, w, n, b1, sm);
    }
    if (fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-5) break;
    b0 = b1;
    ll0 = ll1;
  }
  if (threadIdx.x == 0) {
    const double v = b0 * b0;
    score[j] = (v <= DBL_MAX) ? v : 0.0;  // non-finite (degenerate column) -> ranks last
  }
}

// X2[:, q] = X[:, A[q]]   (x_A of src/screening.cpp:82-87)
__global__ void __launch_bounds__(256) k_gather_cols(const double *__restrict__ X, long ld, const int *__restrict__ A,
                                                     double *__restrict__ X2) {
  const double *src = X + (size_t)A[blockIdx.y] * ld;
  double *dst = X2 + (size_t)blockIdx.y * ld;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i < ld) *reinterpret_cast<d2 *>(dst + i) = *reinterpret_cast<const d2 *>(src + i);
}

// column sums of squares / cross products on a masked row set: out[j] = sum_i m_i x_ij^2 (xtx) --
// group_XTX for 1x1 groups (src/utilities.cpp:153-165, src/Metric.h:108-129) -- via k_xtv with
// v2 = mask; and X^T (m*y) via k_xtv with v = m*y.  Helper: v_out = a * b elementwise (or copy).
__global__ void __launch_bounds__(256) k_vec_mul(const double *__restrict__ a, const double *__restrict__ b, long n,
                                                 double *__restrict__ out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = b ? a[i] * b[i] : a[i];
}

__global__ void __launch_bounds__(256) k_part_sum(const double *__restrict__ part, int nrb, int p,
                                                  double *__restrict__ out) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p) return;
  double s = 0.0;
  for (int rb = 0; rb < nrb; rb++) s += part[(size_t)rb * p + j];
  out[j] = s;
}

__global__ void __launch_bounds__(256) k_fill(double *__restrict__ a, long n, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = v;
}

// Gram column table: optional intercept, the new active columns, zero padding, optional working response.
// It also sets ctrl->same_prev: the new active set equals the one of the previous PDAS iteration of this fit
// (l >= 1).  Then the restricted fit would reproduce the current coefficients bit for bit (same columns, same
// rows, same lambda), so Gram, solve and residual are skipped and k_commit only records the iteration.
__global__ void __launch_bounds__(256) k_gram_cols(const int *__restrict__ A_new, int T0, int mp, int intercept,
                                                   int rhs_col, int *__restrict__ cols, FitCtrl *__restrict__ ctrl,
                                                   int slot, const int *__restrict__ A_cur, int allow_skip) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  if (ctrl != nullptr) {
    int diff = 1;
    if (allow_skip && ctrl->l >= 1 && ctrl->k_cur == T0) {
      diff = 0;
      for (int i = threadIdx.x; i < T0; i += 256) diff |= (A_new[i] != A_cur[i]);
    }
    diff = __syncthreads_or(diff);
    if (threadIdx.x == 0) ctrl->same_prev = diff ? 0 : 1;
  }
  // layout: [ones?] A_new[0..T0) zero padding ... [working response at mp-1 ?]
  for (int i = threadIdx.x; i < mp; i += 256) {
    int v = -1;  // aux column 0: zeros
    int a = i - intercept;
    if (intercept && i == 0) v = -2;
    else if (a >= 0 && a < T0) v = A_new[a];
    if (rhs_col == 1 && i == mp - 1) v = -3;
    if (rhs_col == 2 && a == T0) v = -3;  // (Cox, one-pass Hessian: the bookkeeping column right behind the active ones)
    cols[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Covariance-update form of the LM score pass.
//
// get_A (src/Algorithm.h:1097-1127) needs d = X^T (m (y - X_A b_A)) for ALL p columns at every PDAS iteration; the
// streaming form (k_xtv) reads the whole of X for it.  But d = X^T(m y) - sum_{a in A} (X^T diag(m) x_a) b_a, and
// the vectors g_a = X^T diag(m) x_a depend only on the column a and the row set.  Every row set keeps a cache
// G[:, slot] of those p-vectors; a PDAS iteration whose active columns are all cached costs one p x |A| GEMV over
// G (k_cov_d) and a gather of the |A| x |A| Gram for the solve (k_cov_gram) -- X is not read at all.  Missing columns
// are formed 32 at a time by ONE pass over X on the fp64 matrix cores (k_cov_panel: X^T diag(m) X_S, S = the missing
// columns plus the best-scoring uncached ones, which are the likeliest to enter next), so a warm-started path
// streams X a dozen times instead of once per iteration.
//
//   k_cov_need       which columns of the wanted set are not cached; masked score copy for the speculation
//   k_topk           (run only on a miss) the 32 best uncached columns
//   k_cov_fill_list  final fill list, cache slots; parks the fit if the list exceeds what the slot's panel covers
//   k_cov_panel      part[slab][j tile][rhs tile] = X_j^T diag(m) X_S on a row slab (MFMA f64 16x16x4)
//   k_cov_reduce     fixed-order sum over slabs, scatter into G
//   k_cov_d / k_cov_gram   the GEMV and the Gram gather
// ------------------------------------------------------------------------------------------
constexpr int COV_NJ = 4;   // streamed 16-column tiles per wave

__device__ __forceinline__ bool cov_gate(const FitCtrl *ctrl, int slot) {
  if (ctrl->done) return false;
  if (slot == 0) return ctrl->l == 0;  // start of a fit
  return ctrl->l == slot - 1 && !ctrl->same_prev;
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
                              const int *__restrict__ A_cur, bool known_diff, bool no_restart) {
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

__global__ void __launch_bounds__(256) k_cov_need(const int *__restrict__ list, int len,
                                                  const double *__restrict__ bd, double *__restrict__ bd2, int p,
                                                  int *__restrict__ slot_of, int *__restrict__ meta, int C,
                                                  int *__restrict__ fcols, FitCtrl *__restrict__ ctrl, int slot,
                                                  const int *__restrict__ A_cur, int no_restart) {
  KT(12);
  if (ctrl->done || (slot == 0 ? ctrl->l != 0 : ctrl->l != slot - 1)) return;
  cov_need_body<256>(list, len, bd, bd2, p, slot_of, meta, C, fcols, ctrl, slot, A_cur, false, no_restart != 0);
}

// Final fill list: the missing columns, then speculative ones (the best-scoring uncached columns, `extras`) up to
// the next multiple of 32 that leaves room for at least 16 of them; cache slots are handed out here.
// parked = 1: issued by the host for a parked fit; 0: start of a fit (slot 0).
__global__ void __launch_bounds__(256) k_cov_fill_list(int *__restrict__ fcols, const int *__restrict__ extras,
                                                       const double *__restrict__ bd2, int *__restrict__ slot_of,
                                                       int *__restrict__ meta, FitCtrl *__restrict__ ctrl,
                                                       int parked, int spec_max, int spec) {
  KT(10);
  if (parked ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int nm = ctrl->cov_nmiss;  // left by the lookup of this fit (k_cov_need / cov_need_body)
  const int tid = threadIdx.x;
  if (nm == 0) {
    if (tid == 0) ctrl->cov_nfill = 0;
    return;
  }
  const int count = meta[0];
  // spec: the lookup left a masked copy of the scores (bd2) and the host ran the selection of `extras` on it.
  // spec_max = 32: the list is rounded up to the next multiple of 32 that leaves room for >= 16 speculative columns;
  // spec_max = 64 (pair panel kernel: two groups per pass over X): to the next multiple of 64 with room for >= 32
  const int room = spec ? min(((nm + spec_max / 2 + spec_max - 1) / spec_max) * spec_max - nm, spec_max) : 0;
  __shared__ int s_ne;
  if (tid < 64) {
    const bool valid = spec && tid < spec_max && bd2[extras[tid]] >= 0.0;  // a genuine uncached column
    const unsigned long long bal = __ballot(valid);
    const int rank = __popcll(bal & ((1ull << tid) - 1ull));
    if (valid && rank < room) fcols[nm + rank] = extras[tid];
    if (tid == 0) s_ne = min((int)__popcll(bal), room);
  }
  __syncthreads();
  const int tot = nm + s_ne, padded = (tot + COV_R - 1) / COV_R * COV_R;
  for (int i = tot + tid; i < padded; i += 256) fcols[i] = -1;
  for (int i = tid; i < tot; i += 256) slot_of[fcols[i]] = count + i;
  if (tid == 0) {
    meta[0] = count + tot;
    ctrl->cov_nfill = padded;
    ctrl->cov_groups += padded / COV_R;
  }
}

__global__ void k_cov_resume(FitCtrl *__restrict__ ctrl) {
  KT(11);
  if (ctrl->cov_stall) {
    ctrl->cov_stall = 0;
    ctrl->l = -1 - ctrl->l;
  }
}

// Fold chains side by side (CV row sets, shared fills; bessx_cv.cpp: fold_fits_side_by_side): ONE fill for every chain
// that is parked on a cache miss (cov_stall = 1) or on a full cache (4).  Runs while every chain is quiet.  The wanted
// sets (u.list: the new active set of every parked chain; after `restart` -- decided by the host, which knows the
// column count -- also the current active set of every chain that is in the middle of a fit) are looked up again
// here against the slot map as it is NOW: a column two folds miss gets one slot, and after a restart every wanted
// column is missing.  Then the best uncached columns of ONE chain's scores (extras / bd2: the masked copy its lookup left)
// fill the list up by the rule of k_cov_fill_list.  fill_ctrl gates the panel / reduce / compact launches of the fill
// (cov_stall = 1, cov_nfill) and carries the column count (k_cur) and the pass count (cov_groups) back to the host.
__global__ void __launch_bounds__(256) k_cov_fill_union(const CovUnion u, int restart, const int *__restrict__ extras,
                                                        const double *__restrict__ bd2, int spec_max, int spec_min,
                                                        int *__restrict__ slot_of, int *__restrict__ meta, int p,
                                                        int *__restrict__ fcols, FitCtrl *__restrict__ fill_ctrl, int C) {
  __shared__ int wsum[4];
  __shared__ int s_ne;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (restart == 2) {
    // decided here: the due lists' missing columns (a column two lists miss counts twice: an upper bound) against
    // the room that is left
    int ub = 0;
    for (int f = 0; f < u.nf; f++) {
      if (u.on_restart[f]) continue;
      const int *__restrict__ list = u.list[f];
      for (int base = 0; base < u.len[f]; base += 256) {
        const int i = base + tid;
        const int col = i < u.len[f] ? list[i] : -1;
        ub += __syncthreads_count(col >= 0 && slot_of[col] < 0);
      }
    }
    restart = (meta[0] + ub + COV_R > C) ? 1 : 0;
    __syncthreads();
  }
  if (restart) {
    for (int j = tid; j < p; j += 256) slot_of[j] = -1;
    __syncthreads();
  }
  const int count = restart ? 0 : meta[0];
  int nm = 0;
  for (int f = 0; f < u.nf; f++) {  // uniform
    if (u.on_restart[f] && !restart) continue;
    const int *__restrict__ list = u.list[f];
    const int len = u.len[f];
    for (int base = 0; base < len; base += 256) {
      const int i = base + tid;
      const int col = i < len ? list[i] : -1;
      const int miss = (col >= 0 && slot_of[col] < 0) ? 1 : 0;
      const unsigned long long bal = __ballot(miss);
      const int rank = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wave] = __popcll(bal);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        off += (w < wave) ? wsum[w] : 0;
        tot += wsum[w];
      }
      if (miss) {
        fcols[nm + off + rank] = col;
        slot_of[col] = count + nm + off + rank;
      }
      nm += tot;
      __syncthreads();  // (the slots handed out are visible to the lookups of the next chunk / chain)
    }
  }
  // (spec_min: the list is rounded up to the next multiple of spec_max that leaves room for that many speculative columns)
  const int room = (extras != nullptr && nm > 0) ? min(((nm + spec_min + spec_max - 1) / spec_max) * spec_max - nm, spec_max) : 0;
  if (tid == 0) s_ne = 0;
  __syncthreads();
  if (tid < 64 && room > 0) {
    const int col = tid < spec_max ? extras[tid] : -1;
    const bool valid = col >= 0 && bd2[col] >= 0.0 && slot_of[col] < 0;  // still a genuine uncached column
    const unsigned long long bal = __ballot(valid);
    const int rank = __popcll(bal & ((1ull << tid) - 1ull));
    if (valid && rank < room) fcols[nm + rank] = col;
    if (tid == 0) s_ne = min((int)__popcll(bal), room);
  }
  __syncthreads();
  const int tot = nm + s_ne, padded = (tot + COV_R - 1) / COV_R * COV_R;
  for (int i = tot + tid; i < padded; i += 256) fcols[i] = -1;
  for (int i = nm + tid; i < tot; i += 256) slot_of[fcols[i]] = count + i;
  if (tid == 0) {
    meta[0] = count + tot;
    if (restart) {
      meta[3] += 1;
      meta[4] = 0;
    }
    fill_ctrl->cov_stall = 1;
    fill_ctrl->cov_nfill = padded;
    fill_ctrl->cov_groups += padded / COV_R;
    fill_ctrl->k_cur = count + tot;
    fill_ctrl->cov_nmiss = restart;  // (1: the cache was started over by this fill)
  }
}

// The panel kernel: one BLOCK (4 waves) = 64 streamed columns x 32 right-hand-side columns on one row slab; big = 1:
// issued by the host for a parked fit (no slot gate), covers groups g0 .. g0+ngroups-1.
// Global loads are coalesced the way the streaming score pass does it -- a wave instruction reads 512 contiguous
// bytes of each of two columns (64 rows) -- into registers, then to an LDS tile [column][row] (row stride padded to
// 66 doubles: conflict-free 16-byte reads in the MFMA operand layout).  Wave w multiplies streamed tile w with both
// right-hand-side tiles.  (Round 2 measured this design against direct-to-register loads, a double-buffered tile, LDS-DMA
// staging with 64- and 32-row chunks and a copy of X in the MFMA operand layout: DESIGN.md 3a; only the two kernels
// that won are kept -- this one, and the pair kernel for launches of two groups.)
constexpr int CP_RB = 64;            // rows per chunk
#ifndef CP_PAD
#define CP_PAD 2
#endif
constexpr int CP_LD = CP_RB + CP_PAD;  // padded row stride of a column in LDS (doubles)
constexpr int CP_COLS = 64 + COV_R;  // columns staged per chunk
// The loads run TWO chunks ahead (two register stages, one LDS tile, two barriers per chunk): more bytes in flight per
// CU at the LDS footprint of one tile (2-3 blocks per CU).
template <bool MASKED>
__global__ void __launch_bounds__(256) k_cov_panel_lds2(const double *__restrict__ X, const double *__restrict__ aux,
                                                        long ld, int p, const double *__restrict__ mask,
                                                        const int *__restrict__ fcols, int g0, int ngroups,
                                                        int rows_per_slab, int nslab, int njg,
                                                        double *__restrict__ part, const FitCtrl *__restrict__ ctrl,
                                                        int big) {
  KT(5);
  if (big ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int nfill = ctrl->cov_nfill;
  const long per_group = (long)nslab * njg;
  const int gl = (int)(blockIdx.x / per_group);
  if (gl >= ngroups || (g0 + gl) * COV_R >= nfill) return;
  const int rem = (int)(blockIdx.x - (long)gl * per_group);
  const int slab = rem / njg, jg = rem - slab * njg;
  extern __shared__ double smem[];  // [CP_COLS][CP_LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, q = lane >> 4;
  const int ru = tid & 31, cbase = tid >> 5;
  const double *src[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const int cc = i * 8 + cbase;
    int col;
    if (cc < 64) {
      const int j = jg * 64 + cc;
      col = j < p ? j : -1;
    } else {
      col = fcols[(g0 + gl) * COV_R + cc - 64];
    }
    src[i] = gram_col(X, aux, ld, col) + 2 * ru;
  }
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int nchunk = (int)((r_end - r_begin + CP_RB - 1) / CP_RB);
  d2 stA[12], stB[12], mA, mB;
#define CP_LOAD(st, ms, r)                                                                                     \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; i++) st[i] =                                                      \
        __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src[i] + (r)));                                \
    _Pragma("unroll") for (int i = 8; i < 12; i++) st[i] = *reinterpret_cast<const d2 *>(src[i] + (r));        \
    if (MASKED) ms = *reinterpret_cast<const d2 *>(mask + (r) + 2 * ru);                                       \
  } while (0)
#define CP_STORE(st, ms)                                                                                       \
  do {                                                                                                         \
    double *dst = smem + 2 * ru;                                                                               \
    _Pragma("unroll") for (int i = 0; i < 12; i++) {                                                           \
      d2 v = st[i];                                                                                            \
      if (MASKED && i >= 8) v = v * ms;                                                                        \
      *reinterpret_cast<d2 *>(dst + (size_t)(i * 8 + cbase) * CP_LD) = v;                                      \
    }                                                                                                          \
  } while (0)
  d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
  const double *pa = smem + (size_t)(wv * 16 + c) * CP_LD + 4 * q;
  const double *pb0 = smem + (size_t)(64 + c) * CP_LD + 4 * q, *pb1 = smem + (size_t)(80 + c) * CP_LD + 4 * q;
  auto compute = [&]() {
#pragma unroll
    for (int s = 0; s < CP_RB / 16; s++) {
      const d2 a0 = *reinterpret_cast<const d2 *>(pa + 16 * s), a1 = *reinterpret_cast<const d2 *>(pa + 16 * s + 2);
      const d2 x0 = *reinterpret_cast<const d2 *>(pb0 + 16 * s), x1 = *reinterpret_cast<const d2 *>(pb0 + 16 * s + 2);
      const d2 y0 = *reinterpret_cast<const d2 *>(pb1 + 16 * s), y1 = *reinterpret_cast<const d2 *>(pb1 + 16 * s + 2);
      const double ax = a0.x, ay = a0.y, az = a1.x, aw = a1.y;
      const double b0x = x0.x, b0y = x0.y, b0z = x1.x, b0w = x1.y;
      const double b1x = y0.x, b1y = y0.y, b1z = y1.x, b1w = y1.y;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b0x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b1x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b0y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b1y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(az, b0z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(az, b1z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, b0w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, b1w, acc1, 0, 0, 0);
    }
  };
  // LDS = chunk k, stage A = chunk k+1, stage B = chunk k+2 (in flight)
  CP_LOAD(stA, mA, r_begin);
  CP_STORE(stA, mA);
  if (nchunk > 1) CP_LOAD(stA, mA, r_begin + CP_RB);
  if (nchunk > 2) CP_LOAD(stB, mB, r_begin + 2 * CP_RB);
  __syncthreads();
  for (int k = 0; k < nchunk; k += 2) {
    compute();
    __syncthreads();
    if (k + 1 < nchunk) CP_STORE(stA, mA);
    __syncthreads();
    if (k + 3 < nchunk) CP_LOAD(stA, mA, r_begin + (long)(k + 3) * CP_RB);
    if (k + 1 >= nchunk) break;
    compute();
    __syncthreads();
    if (k + 2 < nchunk) CP_STORE(stB, mB);
    __syncthreads();
    if (k + 4 < nchunk) CP_LOAD(stB, mB, r_begin + (long)(k + 4) * CP_RB);
  }
#undef CP_LOAD
#undef CP_STORE
  const size_t tiles_per_slab = (size_t)njg * COV_NJ * 2;
  double *out = part + (((size_t)gl * nslab + slab) * tiles_per_slab + (size_t)(jg * COV_NJ + wv) * 2) * 256;
  *reinterpret_cast<d4 *>(out + lane * 4) = acc0;
  *reinterpret_cast<d4 *>(out + 256 + lane * 4) = acc1;
}

// The panel kernel for a PAIR of 32-column groups: 64 right-hand-side columns against the same 64 streamed columns,
// X read ONCE for both groups.  At 32 right-hand-side columns the kernel sits between its two roofs (8 flop per
// streamed byte: 0.65 of HBM, 0.54 of the fp64 matrix cores, neither saturated because the per-chunk overheads --
// barriers, staging stores, operand reads -- are paid per 32 KB of X); at 64 the same overheads buy twice the matrix
// work, the kernel is bound by the fp64 MFMA rate (16 flop per streamed byte) and a path needs about half the passes
// over X.  Same staging scheme as k_cov_panel_lds2 (coalesced 16-byte loads two chunks ahead, one LDS tile
// [column][row + pad]); wave w multiplies streamed tile w with the four right-hand-side tiles.  If the second group
// is beyond the fill list (decided on the device) the block does the work of the 32-column kernel.
constexpr int CP2_COLS = 64 + 2 * COV_R;  // columns staged per chunk
// TWO: both groups of the pair are in the fill list (decided on the device, one branch at kernel entry -- inside the
// loop it would split the matrix-core instruction stream).  The operand reads of row step s + 1 are issued before the
// MFMAs of step s (two operand register sets): left to the compiler, every step started with its LDS reads and a
// full wait, exposing the LDS latency eight times per chunk.
template <bool MASKED, bool TWO>
__device__ __forceinline__ void cov_pair_body(const double *__restrict__ X, const double *__restrict__ aux, long ld,
                                              int p, const double *__restrict__ mask, const int *__restrict__ fcols,
                                              int g0, int rows_per_slab, int nslab, int njg,
                                              double *__restrict__ part, double *smem) {
  constexpr int NT = TWO ? 4 : 2;   // right-hand-side tiles
  constexpr int NL = TWO ? 16 : 12;  // staged columns / 8 = loads per thread and chunk
  const int slab = blockIdx.x / njg, jg = blockIdx.x - slab * njg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, q = lane >> 4;
  const int ru = tid & 31, cbase = tid >> 5;
  // streamed columns jg * 64 + cbase + 8 i: one pointer and a uniform stride (a column beyond p re-reads the last
  // existing one of its thread: its products land in rows >= p, which the reduce kernel never stores); right-hand-side
  // columns: one pointer each
  const int jc = min(jg * 64 + cbase, p - 1);
  const double *sx = X + (size_t)jc * ld + 2 * ru;
  const long sstride = 8 * ld;
  const int ilim = jg * 64 + cbase < p ? (p - 1 - (jg * 64 + cbase)) / 8 : 0;  // last i whose column exists
  const double *src[NL - 8];
#pragma unroll
  for (int i = 0; i < NL - 8; i++) src[i] = gram_col(X, aux, ld, fcols[g0 * COV_R + i * 8 + cbase]) + 2 * ru;
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int nchunk = (int)((r_end - r_begin + CP_RB - 1) / CP_RB);
  d2 st[NL], ms;
  auto load_chunk = [&](long r) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      st[i] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(sx + min(i, ilim) * sstride + r));
#pragma unroll
    for (int i = 8; i < NL; i++) st[i] = *reinterpret_cast<const d2 *>(src[i - 8] + r);
    if (MASKED) ms = *reinterpret_cast<const d2 *>(mask + r + 2 * ru);
  };
  auto store_chunk = [&]() {
    double *dst = smem + 2 * ru;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      d2 v = st[i];
      if (MASKED && i >= 8) v = v * ms;
      *reinterpret_cast<d2 *>(dst + (size_t)(i * 8 + cbase) * CP_LD) = v;
    }
  };
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  const double *pa = smem + (size_t)(wv * 16 + c) * CP_LD + 4 * q;
  const double *pb = smem + (size_t)(64 + c) * CP_LD + 4 * q;  // right-hand-side tile t at pb + t * 16 * CP_LD
  struct Ops {
    d2 a0, a1, b0[NT], b1[NT];
  };
  auto read_ops = [&](int s, Ops &o) {
    o.a0 = *reinterpret_cast<const d2 *>(pa + 16 * s);
    o.a1 = *reinterpret_cast<const d2 *>(pa + 16 * s + 2);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      o.b0[t] = *reinterpret_cast<const d2 *>(pb + (size_t)t * 16 * CP_LD + 16 * s);
      o.b1[t] = *reinterpret_cast<const d2 *>(pb + (size_t)t * 16 * CP_LD + 16 * s + 2);
    }
  };
  auto mfma_ops = [&](const Ops &o) {
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0.x, o.b0[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0.y, o.b0[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1.x, o.b1[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1.y, o.b1[t].y, acc[t], 0, 0, 0);
  };
  auto compute = [&]() {
    Ops oa, ob;
    read_ops(0, oa);
#pragma unroll
    for (int s = 0; s < CP_RB / 16; s += 2) {
      read_ops(s + 1, ob);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads back in front of their use)
      mfma_ops(oa);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < CP_RB / 16) read_ops(s + 2, oa);
      __builtin_amdgcn_sched_barrier(0);
      mfma_ops(ob);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // LDS = chunk k; the registers hold chunk k + 1, loaded while chunk k is multiplied (one stage: the matrix work of
  // a chunk is twice that of the 32-column kernel, and two 240-register waves per SIMD would not fit)
  load_chunk(r_begin);
  store_chunk();
  if (nchunk > 1) load_chunk(r_begin + CP_RB);
  __syncthreads();
#ifndef PAIR_DBG
#define PAIR_DBG 0
#endif
  for (int k = 0; k < nchunk; k++) {
    compute();
    if (PAIR_DBG != 2) __syncthreads();
    if (PAIR_DBG != 1 && PAIR_DBG != 3 && k + 1 < nchunk) store_chunk();
    if (PAIR_DBG != 2) __syncthreads();
    if (PAIR_DBG != 1 && k + 2 < nchunk) load_chunk(r_begin + (long)(k + 2) * CP_RB);
  }
  // the partial-sum layout of the 32-column kernels: [group][slab][tile pair] -- the reduce kernel is unchanged
  const size_t tiles_per_slab = (size_t)njg * COV_NJ * 2;
  double *out = part + ((size_t)slab * tiles_per_slab + (size_t)(jg * COV_NJ + wv) * 2) * 256;
  *reinterpret_cast<d4 *>(out + lane * 4) = acc[0];
  *reinterpret_cast<d4 *>(out + 256 + lane * 4) = acc[1];
  if (TWO) {
    double *out2 = out + (size_t)nslab * tiles_per_slab * 256;
    *reinterpret_cast<d4 *>(out2 + lane * 4) = acc[NT - 2];
    *reinterpret_cast<d4 *>(out2 + 256 + lane * 4) = acc[NT - 1];
  }
}

template <bool MASKED>
__global__ void __launch_bounds__(256) k_cov_panel_pair(const double *__restrict__ X, const double *__restrict__ aux,
                                                        long ld, int p, const double *__restrict__ mask,
                                                        const int *__restrict__ fcols, int g0, int rows_per_slab,
                                                        int nslab, int njg, double *__restrict__ part,
                                                        const FitCtrl *__restrict__ ctrl, int big) {
  KT(5);
  if (big ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int nfill = ctrl->cov_nfill;
  if (g0 * COV_R >= nfill) return;
  extern __shared__ double smem[];  // [CP2_COLS][CP_LD]
  if ((g0 + 1) * COV_R < nfill)     // uniform
    cov_pair_body<MASKED, true>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
  else
    cov_pair_body<MASKED, false>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
}

// G[j, slot_of[col]] = sum over slabs (fixed order); grid (tiles of one group, groups)
__device__ __forceinline__ void cov_reduce_body(const double *__restrict__ part, int g0, int ngroups, int nslab,
                                                int njg, int p, const int *__restrict__ fcols,
                                                const int *__restrict__ slot_of, double *__restrict__ G,
                                                const FitCtrl *__restrict__ ctrl, int big,
                                                int ex_lo, int ex_hi) {
  if (big ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int gl = blockIdx.y;
  if (gl >= ngroups || (g0 + gl) * COV_R >= ctrl->cov_nfill) return;
  const size_t tiles_per_slab = (size_t)njg * COV_NJ * 2;
  const int tile = blockIdx.x, e = threadIdx.x;
  double s = 0.0;
  // slabs [ex_lo, ex_hi) are left out: on the fold-major copy of X (shared fills of the CV row sets) they are the
  // rows of the fold whose TRAINING rows this cache belongs to
  for (int sl = 0; sl < nslab; sl++)
    if (sl < ex_lo || sl >= ex_hi) s += part[(((size_t)gl * nslab + sl) * tiles_per_slab + tile) * 256 + e];
  const int jt = tile >> 1, ni = tile & 1, lane = e >> 2, reg = e & 3;
  const int j = jt * 16 + (lane >> 4) + 4 * reg;
  const int ci = (g0 + gl) * COV_R + ni * 16 + (lane & 15);
  const int col = fcols[ci];
  if (j < p && col >= 0) G[(size_t)slot_of[col] * p + j] = s;
}

__global__ void __launch_bounds__(256) k_cov_reduce(const double *__restrict__ part, int g0, int ngroups, int nslab,
                                                    int njg, int p, const int *__restrict__ fcols,
                                                    const int *__restrict__ slot_of, double *__restrict__ G,
                                                    const FitCtrl *__restrict__ ctrl, int slot, int big,
                                                    int ex_lo, int ex_hi) {
  KT(6);
  cov_reduce_body(part, g0, ngroups, nslab, njg, p, fcols, slot_of, G, ctrl, big, ex_lo, ex_hi);
}

// ... for every row set of a cross-validation at once (shared fills: blockIdx.z = row set; all of them cache the same
// columns under the same slots, each leaves its own fold's slabs out)
__global__ void __launch_bounds__(256) k_cov_reduce_sets(const double *__restrict__ part, int g0, int ngroups, int nslab,
                                                         int njg, int p, const int *__restrict__ fcols,
                                                         const int *__restrict__ slot_of, const CovRowSets rs,
                                                         const FitCtrl *__restrict__ ctrl, int big) {
  KT(6);
  const int r = blockIdx.z;
  cov_reduce_body(part, g0, ngroups, nslab, njg, p, fcols, slot_of, rs.G[r], ctrl, big, rs.ex_lo[r], rs.ex_hi[r]);
}

// After a fill: the Gram entries between the columns just cached and every cached column, written into the small
// slot-indexed matrix GS (both triangles) that the row-dealt solve gathers from -- a few hundred KB that stay in L2,
// instead of k^2 reads scattered over the p x C cache.
__device__ __forceinline__ void cov_compact_body(const double *__restrict__ G, int p,
                                                 const int *__restrict__ slot_of,
                                                 const int *__restrict__ fcols, int g0, double *__restrict__ GS,
                                                 int CS, const FitCtrl *__restrict__ ctrl, int big,
                                                 const double *__restrict__ xtx, int *__restrict__ meta) {
  if (big ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int gl = blockIdx.y;
  if ((g0 + gl) * COV_R >= ctrl->cov_nfill) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int t = j < p ? slot_of[j] : -1;
  if (t < 0) return;
  const double dj = xtx != nullptr ? xtx[j] : 0.0;
  for (int c = 0; c < COV_R; c++) {
    const int col = fcols[(g0 + gl) * COV_R + c];
    if (col < 0) continue;
    const int sc = slot_of[col];
    if (sc < 0) continue;
    const double v = G[(size_t)sc * p + j];
    // two cached columns that are exactly dependent (duplicates, mirror images: |x_j . x_col| = |x_j| |x_col|): a system
    // that holds both is singular but consistent -- conjugate gradients would split the coefficient between them where
    // the reference's pivoted factorisation gives it to one.  meta[4] tells the solve kernels to leave such row sets
    // to k_chol, whose pivot test routes them to the pivoted solve (sym_pivoted_solve).
    if (xtx != nullptr && j != col && v * v >= (1.0 - 2e-11) * dj * xtx[col] && dj > 0.0) meta[4] = 1;
    if (sc < CS && t < CS) {
      GS[(size_t)sc * CS + t] = v;
      GS[(size_t)t * CS + sc] = v;
    }
  }
}

__global__ void __launch_bounds__(256) k_cov_compact(const double *__restrict__ G, int p,
                                                     const int *__restrict__ slot_of,
                                                     const int *__restrict__ fcols, int g0, double *__restrict__ GS,
                                                     int CS, const FitCtrl *__restrict__ ctrl, int big,
                                                     const double *__restrict__ xtx, int *__restrict__ meta) {
  KT(7);
  cov_compact_body(G, p, slot_of, fcols, g0, GS, CS, ctrl, big, xtx, meta);
}

__global__ void __launch_bounds__(256) k_cov_compact_sets(const CovRowSets rs, int p, const int *__restrict__ slot_of,
                                                          const int *__restrict__ fcols, int g0, int CS,
                                                          const FitCtrl *__restrict__ ctrl, int big,
                                                          int *__restrict__ meta) {
  KT(7);
  const int r = blockIdx.z;
  cov_compact_body(rs.G[r], p, slot_of, fcols, g0, rs.GS[r], CS, ctrl, big, rs.xtx[r], meta);
}


// d_j = (X^T m y)_j - sum_i G[j, slot(A_i)] b_i ; 64 columns per block, the sum over i cut in 4 interleaved parts.
// The sacrifice score of k_score (LM branch) is formed in the same kernel: bd_j = (phi b_j + d_j / phi)^2 with
// d_j / n_t - 2 lambda b_j and phi = sqrt(2 lambda + x_j.x_j / n_t).
__global__ void __launch_bounds__(256) k_cov_d(const double *__restrict__ G, int p, const int *__restrict__ slot_of,
                                               const double *__restrict__ xty, const int *__restrict__ A_cur,
                                               const double *__restrict__ b_cur, double *__restrict__ d_out,
                                               const double *__restrict__ beta_dense, const double *__restrict__ xtx,
                                               double n_t, double lambda, const unsigned char *__restrict__ always,
                                               double *__restrict__ bd, const unsigned char *__restrict__ inA,
                                               double *__restrict__ bmm, const FitCtrl *__restrict__ ctrl, int slot) {
  KT(4);
  // what the epilogue needs of this block's 32 columns does not depend on the control block: these loads are in
  // flight while the gate below waits for its own (one round trip less on the block's critical path)
  const int jj = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jj;
  const bool epi = g == 0 && j < p;
  const double e_xty = epi ? xty[j] : 0.0, e_b = epi ? beta_dense[j] : 0.0, e_xtx = epi ? xtx[j] : 1.0;
  const unsigned char e_in = epi ? inA[j] : (unsigned char)0;
  const unsigned char e_al = (epi && always != nullptr) ? always[j] : (unsigned char)0;
  if (ctrl->done || ctrl->l != slot - 1) return;
  // 32 columns x 8 thread groups per block; group g adds the active columns i = g, g+8, ... (two interleaved
  // accumulators), the 8 partial sums are added in group order: a fixed summation tree.
  constexpr int CHUNK = 512;  // active columns staged per round: cache slots and coefficients go through LDS
  __shared__ int s_sl[CHUNK];
  __shared__ double s_b[CHUNK];
  __shared__ double sm[8][32];
  const int kc = ctrl->k_cur;
  double acc0 = 0.0, acc1 = 0.0;
  for (int base = 0; base < kc; base += CHUNK) {
    const int cnt = min(CHUNK, kc - base);
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const int sl = slot_of[A_cur[base + i]];
      if (sl < 0) const_cast<FitCtrl *>(ctrl)->cov_miss = 1;  // must not happen: active columns are cached before use
      s_sl[i] = sl;
      s_b[i] = sl < 0 ? 0.0 : b_cur[base + i];
    }
    __syncthreads();
    if (j < p) {
      // 8 cache entries in flight per thread before the first product (a loop of "two loads, wait, two products" pays
      // the L2 latency once per pair); entries beyond cnt read slot 0 and meet a zero coefficient.  Same summation
      // order as before: i = g, g + 16, ... into acc0, i = g + 8, g + 24, ... into acc1.
      for (int i0 = g; i0 < cnt; i0 += 64) {
        double gv[8], bv[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int i = i0 + 8 * t;
          const bool in = i < cnt;
          gv[t] = G[(size_t)(in ? max(s_sl[i], 0) : 0) * p + j];
          bv[t] = in ? s_b[i] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          if (i0 + 8 * t < cnt) acc0 = fma(gv[t], bv[t], acc0);
          if (i0 + 8 * (t + 1) < cnt) acc1 = fma(gv[t + 1], bv[t + 1], acc1);
        }
      }
    }
    __syncthreads();
  }
  sm[g][jj] = acc0 + acc1;
  __syncthreads();
  if (g == 0 && j < p) {
    double t = sm[0][jj];
#pragma unroll
    for (int q = 1; q < 8; q++) t += sm[q][jj];
    const double s1 = e_xty - t;
    d_out[j] = s1;
    const double b = e_b;
    const double d = s1 / n_t - 2.0 * lambda * b;
    const double phi = sqrt(2.0 * lambda + e_xtx / n_t);
    const double inv = 1.0 / phi;
    const double tt = phi * b + inv * d;
    double v = tt * tt;
    if (e_al) v = DBL_MAX;
    bd[j] = v;
    // repeated-set shortcut: smallest score inside the current active set, largest outside, per block
    sm[1][jj] = e_in ? v : DBL_MAX;
    sm[2][jj] = e_in ? -1.0 : v;
  } else if (g == 0) {
    sm[1][jj] = DBL_MAX;
    sm[2][jj] = -1.0;
  }
  __syncthreads();
  // If every score of the current active set beats every score outside it, max_k returns the same set.  Each block
  // leaves its two extremes in bmm; the selection kernel that follows combines them (min / max are exact, so the
  // order does not matter) and skips its search when the test holds.  fast_same = 1 marks bmm as fresh.
  // (third section of bmm: the column of that largest outside score, lowest index on ties -- the arg-max selection
  // of a fit chained one size up reads the block maxima instead of all p scores)
  if (threadIdx.x == 0) {
    double mn = DBL_MAX, mx = -1.0;
    int mi = 0x7fffffff;
    for (int q = 0; q < 32; q++) {
      mn = fmin(mn, sm[1][q]);
      if (sm[2][q] > mx) {
        mx = sm[2][q];
        mi = blockIdx.x * 32 + q;
      }
    }
    bmm[2 * blockIdx.x] = mn;
    bmm[2 * blockIdx.x + 1] = mx;
    bmm[2 * gridDim.x + blockIdx.x] = (double)mi;
    if (blockIdx.x == 0) const_cast<FitCtrl *>(ctrl)->fast_same = 1;
  }
}

// Gram tiles of the new active set in the layout k_chol / k_bc_* read (see k_gram_assemble)
__global__ void __launch_bounds__(256) k_cov_gram(const double *__restrict__ G, int p,
                                                  const int *__restrict__ slot_of, const int *__restrict__ A_new,
                                                  int T0, double *__restrict__ Gt, int *__restrict__ meta,
                                                  const FitCtrl *__restrict__ ctrl, int slot) {
  if (!cov_gate(ctrl, slot)) return;
  const int t = blockIdx.x, lane = threadIdx.x >> 2, r = threadIdx.x & 3;
  int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  const int J = t - I * (I + 1) / 2;
  const int a = I * 16 + (lane >> 4) + 4 * r, b = J * 16 + (lane & 15);
  double v = 0.0;
  if (a < T0 && b < T0) {
    const int sl = slot_of[A_new[b]];
    if (sl >= 0)
      v = G[(size_t)sl * p + A_new[a]];
    else
      const_cast<FitCtrl *>(ctrl)->cov_miss = 1;
  }
  Gt[(size_t)t * 256 + lane * 4 + r] = v;
}

// ------------------------------------------------------------------------------------------
// Hand the result block of a fit to the host without a copy engine round trip: the block (control words, the
// per-256-row sums of squares, the first kcopy coefficients and indices) is written straight into pinned host
// memory at the same offsets, then a sequence number is released at system scope; the host spins on it.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_publish(const unsigned char *__restrict__ dev, unsigned char *host,
                                                 int ctrl_bytes, size_t off_sse, int n_sse, size_t off_b, size_t off_a,
                                                 int kcopy, unsigned long long *seq_host, unsigned long long seq,
                                                 const int *__restrict__ count_ptr) {
  KT(9);
  const int tid = threadIdx.x;
  // (covariance form) columns in the Gram column cache right now, next to the sequence number
  if (tid == 0 && count_ptr != nullptr) seq_host[1] = (unsigned long long)count_ptr[0];
  const unsigned long long *d8 = reinterpret_cast<const unsigned long long *>(dev);
  unsigned long long *h8 = reinterpret_cast<unsigned long long *>(host);
  for (int i = tid; i < ctrl_bytes / 8; i += 256) h8[i] = d8[i];
  for (int i = tid; i < n_sse; i += 256) h8[off_sse / 8 + i] = d8[off_sse / 8 + i];
  for (int i = tid; i < kcopy; i += 256) h8[off_b / 8 + i] = d8[off_b / 8 + i];
  const int *d4 = reinterpret_cast<const int *>(dev + off_a);
  int *h4 = reinterpret_cast<int *>(host + off_a);
  for (int i = tid; i < kcopy; i += 256) h4[i] = d4[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// out[0] = sum_i a_i b_i (b may be null: sum a_i^2... no: sum a_i a_i), one block, fixed order
__global__ void __launch_bounds__(256) k_dot(const double *__restrict__ a, const double *__restrict__ b, long n,
                                             double *__restrict__ out) {
  __shared__ double sm[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) s = fma(a[i], b ? b[i] : a[i], s);
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = s;
}

// streaming copy used to measure the practical HBM ceiling
__global__ void __launch_bounds__(256) k_copy(const d2 *__restrict__ src, d2 *__restrict__ dst, long n2) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long stride = (long)gridDim.x * 256;
  for (; i < n2; i += stride) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                         \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return e__;         \
  } while (0)

hipError_t launch_transpose_in(const double *src, int rows, int p, double *X, long ld, long r0, hipStream_t st) {
  dim3 grid((p + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(k_transpose_in, grid, dim3(256), 0, st, src, rows, p, X, ld, r0);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_normalize(double *X, long ld, int n, int p, double *y, const double *w, int data_type,
                            int is_normal, int add_weight, double *x_mean, double *x_norm, double *y_mean,
                            hipStream_t st) {
  int centre = is_normal && (data_type == 1 || data_type == 2);
  if (is_normal || add_weight) {
    hipLaunchKernelGGL(k_col_normalize, dim3(p), dim3(256), 0, st, X, ld, n, p, w, centre, is_normal, add_weight,
                       x_mean, x_norm);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_y_prepare, dim3(1), dim3(256), 0, st, y, n, w, (int)(is_normal && data_type == 1), add_weight,
                     y_mean);
  LAUNCH_CHECK();
  return hipSuccess;
}

int xtv_rows_per_block(int U) { return 128 * U; }

template <int U, bool TWO>
static hipError_t launch_xtv_t(const double *X, long ld, int p, const double *v, const double *v2, double *part,
                               double *part2, const FitCtrl *ctrl, int slot, hipStream_t st) {
  constexpr int CG = 16;
  int nrb = (int)(ld / (128 * U));
  long nwaves = (long)nrb * ((p + CG - 1) / CG);
  int nblk = (int)((nwaves + 3) / 4);
  hipLaunchKernelGGL((k_xtv<U, CG, TWO>), dim3(nblk), dim3(256), 0, st, X, ld, p, nrb, v, v2, part, part2, ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_xtv(const double *X, long ld, int p, int U, const double *v, const double *v2, double *part,
                      double *part2, const FitCtrl *ctrl, int slot, hipStream_t st) {
  bool two = v2 != nullptr;
  switch (U) {
    case 8: return two ? launch_xtv_t<8, true>(X, ld, p, v, v2, part, part2, ctrl, slot, st)
                       : launch_xtv_t<8, false>(X, ld, p, v, v2, part, part2, ctrl, slot, st);
    case 4: return two ? launch_xtv_t<4, true>(X, ld, p, v, v2, part, part2, ctrl, slot, st)
                       : launch_xtv_t<4, false>(X, ld, p, v, v2, part, part2, ctrl, slot, st);
    case 2: return two ? launch_xtv_t<2, true>(X, ld, p, v, v2, part, part2, ctrl, slot, st)
                       : launch_xtv_t<2, false>(X, ld, p, v, v2, part, part2, ctrl, slot, st);
    default: return two ? launch_xtv_t<1, true>(X, ld, p, v, v2, part, part2, ctrl, slot, st)
                        : launch_xtv_t<1, false>(X, ld, p, v, v2, part, part2, ctrl, slot, st);
  }
}

// tuning aid: run one geometry variant of the score pass (U rows-per-lane factor, CG columns per wave,
// nontemporal or plain loads) on its own
template <int U, int CG, bool NT>
static hipError_t launch_xtv_variant_t(const double *X, long ld, int p, const double *v, double *part, hipStream_t st) {
  int nrb = (int)(ld / (128 * U));
  long nwaves = (long)nrb * ((p + CG - 1) / CG);
  int nblk = (int)((nwaves + 3) / 4);
  hipLaunchKernelGGL((k_xtv<U, CG, false, NT>), dim3(nblk), dim3(256), 0, st, X, ld, p, nrb, v, (const double *)nullptr,
                     part, (double *)nullptr, (const FitCtrl *)nullptr, 0);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_xtv_variant(int variant, const double *X, long ld, int p, const double *v, double *part,
                              hipStream_t st) {
  switch (variant) {
    case 0: return launch_xtv_variant_t<8, 16, true>(X, ld, p, v, part, st);
    case 1: return launch_xtv_variant_t<8, 16, false>(X, ld, p, v, part, st);
    case 2: return launch_xtv_variant_t<8, 8, true>(X, ld, p, v, part, st);
    case 3: return launch_xtv_variant_t<8, 8, false>(X, ld, p, v, part, st);
    case 4: return launch_xtv_variant_t<4, 16, true>(X, ld, p, v, part, st);
    case 5: return launch_xtv_variant_t<4, 16, false>(X, ld, p, v, part, st);
    case 6: return launch_xtv_variant_t<4, 8, true>(X, ld, p, v, part, st);
    case 7: return launch_xtv_variant_t<2, 16, true>(X, ld, p, v, part, st);
    case 8: return launch_xtv_variant_t<8, 4, true>(X, ld, p, v, part, st);
    case 9: return launch_xtv_variant_t<4, 4, true>(X, ld, p, v, part, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_score(const double *part, const double *part2, int nrb, int p, const double *beta_dense,
                        const double *xtx, double n_t, double lambda, int glm, const unsigned char *always,
                        double *bd, const FitCtrl *ctrl, int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_score, dim3((p + 63) / 64), dim3(256), 0, st, part, part2, nrb, p, beta_dense, xtx, n_t,
                     lambda, glm, always, bd, ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

// two-level selection: chunks of <= 32768 scores each keep their k best, a final block selects from the
// concatenated candidates (already in ascending index order).  cand must hold nchunk*k ints.
void topk_set_variant(int) {}  // reserved: one selection kernel exists

static hipError_t launch_topk_one(int nblk, const double *score, const int *idx_in, int len, int chunk, int k,
                                  int *out, const FitCtrl *ctrl, int slot, hipStream_t st,
                                  const int *run_flag = nullptr, const TopkNeed *need = nullptr,
                                  int *tie_flag = nullptr) {
  TopkNeed nd = {};
  if (need) nd = *need;
  if (nd.pub.on) {
    if (nblk != 1) return hipErrorInvalidValue;  // the publisher rides on single-chunk selections only
    nblk = 2;
  }
  const int per = (std::min(len, chunk) + 1023) / 1024;
#define TOPK_GO(EB)                                                                                              \
  hipLaunchKernelGGL(k_topk<EB>, dim3(nblk), dim3(1024), 0, st, score, idx_in, len, chunk, k, out, tie_flag, \
                     ctrl, slot, run_flag, nd)
  if (per <= 2)
    TOPK_GO(2);
  else if (per <= 4)
    TOPK_GO(4);
  else if (per <= 6)
    TOPK_GO(6);
  else if (per <= 8)
    TOPK_GO(8);
  else if (per <= 10)
    TOPK_GO(10);
  else if (per <= 12)
    TOPK_GO(12);
  else if (per <= 16)
    TOPK_GO(16);
  else if (per <= 20)
    TOPK_GO(20);
  else if (per <= 24)
    TOPK_GO(24);
  else
    TOPK_GO(32);
#undef TOPK_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

bool topk_can_fuse_need(int len) { return len <= 1024 * TOPK_E; }

// the exact selection behind a tie (k_topk_ties); tie->flag is raised by the selection kernels, tie->work = 3 len ints
static hipError_t launch_topk_ties(const double *score, int len, int k, int *out, const TopkTie *tie,
                                   const FitCtrl *ctrl, int slot, hipStream_t st, const int *run_flag) {
  hipLaunchKernelGGL(k_topk_ties, dim3(1), dim3(1024), 0, st, score, len, k, out, tie->flag, tie->work, ctrl, slot,
                     run_flag);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_topk(const double *score, int len, int k, int *out, int *cand, const FitCtrl *ctrl, int slot,
                       hipStream_t st, const int *run_flag, const TopkNeed *need, const TopkTie *tie) {
  const int chunk = 1024 * TOPK_E;
  if (need != nullptr && need->slot_of != nullptr) tie = nullptr;  // (fused flavours of the covariance form: see launch_sel_cgr)
  int *tflag = tie != nullptr ? tie->flag : nullptr;
  if (len <= chunk) {
    hipError_t e = launch_topk_one(1, score, nullptr, len, chunk, k, out, ctrl, slot, st, run_flag, need, tflag);
    if (e == hipSuccess && tie != nullptr) e = launch_topk_ties(score, len, k, out, tie, ctrl, slot, st, run_flag);
    return e;
  }
  if (need != nullptr) return hipErrorInvalidValue;  // callers check topk_can_fuse_need()
  int nchunk = (len + chunk - 1) / chunk;
  long ncand = (long)nchunk * k;
  if (ncand > chunk || k > chunk) return hipErrorInvalidValue;  // would need a third level
  // the chunks are BALANCED (ceil(len / nchunk) scores each, the last at most nchunk - 1 fewer): a chunk shorter
  // than k would leave holes in cand, and with full 32768-wide chunks whether that happens depended on
  // len mod 32768 (p = 32769 allowed k = 1 only).  topk_supported() checks the last chunk still holds >= k scores.
  const int bal = (len + nchunk - 1) / nchunk;
  // (a tie inside one chunk need not be one of the whole selection; redoing the selection exactly is right either way)
  hipError_t e = launch_topk_one(nchunk, score, nullptr, len, bal, k, cand, ctrl, slot, st, run_flag, nullptr, tflag);
  if (e != hipSuccess) return e;
  e = launch_topk_one(1, score, cand, (int)ncand, chunk, k, out, ctrl, slot, st, run_flag, nullptr, tflag);
  if (e == hipSuccess && tie != nullptr) e = launch_topk_ties(score, len, k, out, tie, ctrl, slot, st, run_flag);
  return e;
}

bool topk_supported(int len, int k) {
  const int chunk = 1024 * TOPK_E;
  if (len <= chunk) return k <= len;
  int nchunk = (len + chunk - 1) / chunk;
  const int bal = (len + nchunk - 1) / nchunk;
  int last = len - (nchunk - 1) * bal;
  return k <= last && (long)nchunk * k <= chunk;
}

static int g_gram_variant = 1;  // 1 = LDS-staged kernel where it applies (default), 0 = k_gram throughout
void gram_set_variant(int v) { g_gram_variant = v; }
// the LDS-staged kernel forms whole lower triangles of at most 16 tile rows (the callers size the slabs for it:
// gram_lds_slabs())
bool gram_lds_applies(int ntiles, int tile_base) {
  return g_gram_variant == 1 && tile_base == 0 && ntiles <= 16 * 17 / 2;
}
hipError_t gram_lds_prepare() {
  // dynamic LDS beyond 64 KB has to be requested once per kernel instance
  hipError_t e = hipSuccess;
  const int big = (12 * 16 * 66 + 8 * 64 + 12 * 16) * (int)sizeof(double);
#define GL_ATTR(K) \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, big)
  GL_ATTR((k_gram_lds<4, 9, 16, 64, true>));
  GL_ATTR((k_gram_lds<4, 9, 16, 64, false>));
  GL_ATTR((k_gram_lds<8, 10, 12, 64, true>));
  GL_ATTR((k_gram_lds<8, 10, 12, 64, false>));
  GL_ATTR((k_gram_lds<8, 17, 8, 32, true>));
  GL_ATTR((k_gram_lds<8, 17, 8, 32, false>));
  GL_ATTR((k_irls_gram<2, 8, 1, 2>));
  GL_ATTR((k_irls_gram<2, 8, 1, 3>));
  GL_ATTR((k_irls_gram<3, 8, 1, 2>));
  GL_ATTR((k_irls_gram<3, 8, 1, 3>));
  GL_ATTR((k_irls_gram<4, 6, 2, 2>));
  GL_ATTR((k_irls_gram<4, 6, 2, 3>));
  GL_ATTR((k_irls_gram<5, 5, 2, 2>));
  GL_ATTR((k_irls_gram<5, 5, 2, 3>));
  GL_ATTR((k_irls_gram<6, 4, 3, 2>));
  GL_ATTR((k_irls_gram<6, 4, 3, 3>));
  GL_ATTR((k_irls_gram<7, 3, 4, 2>));
  GL_ATTR((k_irls_gram<7, 3, 4, 3>));
  GL_ATTR((k_irls_gram<8, 2, 5, 2>));
  GL_ATTR((k_irls_gram<8, 2, 5, 3>));
#undef GL_ATTR
  return e;
}

// fused IRLS step (k_irls_gram): slab partials of the weighted Gram + the slabs' log-likelihood terms.  The caller
// follows it with k_gram_reduce and k_chol (whose head is the convergence test).  Slabs are NCH 64-row chunks: 8 chunks
// up to 4 tile rows, 4 beyond (the registers that hold the slab between the two uses bound NCH x tile rows).
// rows per slab: about one slab per compute unit (whole 64-row chunks), at least one group of chunks
// rows per slab: about one slab per compute unit, whole 64-row chunks
int irls_gram_slab_rows(int mt, long ld) {
  (void)mt;
  return (int)(((ld + 255) / 256 + 63) / 64 * 64);
}
bool irls_gram_applies(int mt) { return g_gram_variant == 1 && mt >= 1 && mt <= 8; }
hipError_t launch_irls_gram(int fam, const double *X, const double *aux, long ld, int n, const int *cols,
                            const double *y, const double *w, const double *mask, int nslab, int mt, double *part,
                            int ntiles, const FitCtrl *ctrl, int slot, int t, int T0, const double *bcur,
                            double *llpart, hipStream_t st, int wfloor) {
  if (!irls_gram_applies(mt) || T0 + 2 > mt * 16) return hipErrorInvalidValue;  // intercept, T0 columns, ..., z last
  const int rows = irls_gram_slab_rows(mt, ld);
  if ((long)nslab * rows < ld) return hipErrorInvalidValue;
#define IG_GO(NP_, NCH_, TPW_, FAM_)                                                                               \
  do {                                                                                                             \
    const size_t tile = std::max((size_t)mt * 16 * 66, (size_t)8 * NCH_ * 64);                                     \
    const size_t lds = (tile + 2 * (size_t)NCH_ * 64 + (size_t)mt * 16) * sizeof(double);                          \
    hipLaunchKernelGGL((k_irls_gram<NP_, NCH_, TPW_, FAM_>), dim3(nslab), dim3(512), lds, st, X, aux, ld, n, cols, \
                       y, w, mask, rows, mt, part, ntiles, ctrl, slot, t, T0, bcur, llpart, wfloor);               \
  } while (0)
#define IG_FAM(NP_, NCH_, TPW_) \
  if (fam == 2)                 \
    IG_GO(NP_, NCH_, TPW_, 2);  \
  else                          \
    IG_GO(NP_, NCH_, TPW_, 3)
  switch (mt) {
    case 1:
    case 2: IG_FAM(2, 8, 1); break;
    case 3: IG_FAM(3, 8, 1); break;
    case 4: IG_FAM(4, 6, 2); break;
    case 5: IG_FAM(5, 5, 2); break;
    case 6: IG_FAM(6, 4, 3); break;
    case 7: IG_FAM(7, 3, 4); break;
    default: IG_FAM(8, 2, 5); break;
  }
#undef IG_FAM
#undef IG_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_gram_reduce(const double *part, int nslab, int ntiles, double *Gt, const FitCtrl *ctrl, int slot,
                              int gate_mode, hipStream_t st) {
  hipLaunchKernelGGL(k_gram_reduce, dim3((ntiles * 256 + 15) / 16), dim3(256), 0, st, part, nslab, ntiles, Gt, ctrl,
                     slot, gate_mode);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_gram(const double *X, const double *aux, long ld, const int *cols, const double *w,
                       int rows_per_slab, const GramTask *tasks, int ntask, int nslab, double *part, int ntiles,
                       double *Gt, const FitCtrl *ctrl, int slot, int gate_mode, hipStream_t st, int tile_base) {
  // (gate modes 3 / 4 are the cached LM Gram: a whole-triangle launch that nearly always falls through its gate next to
  // an incremental one -- the fall-through of a large-LDS block costs more than it could ever save there)
  if (gram_lds_applies(ntiles, tile_base) && gate_mode != 3 && gate_mode != 4) {
    int mt = 1;
    while (mt * (mt + 1) / 2 < ntiles) mt++;
#define GL_GO(NW_, TPW_, NP_, RB_)                                                                                  \
  do {                                                                                                              \
    const size_t lds = ((size_t)mt * 16 * (RB_ + 2) + RB_) * sizeof(double);                                        \
    if (w)                                                                                                          \
      hipLaunchKernelGGL((k_gram_lds<NW_, TPW_, NP_, RB_, true>), dim3(nslab), dim3(64 * NW_), lds, st, X, aux, ld, \
                         cols, w, rows_per_slab, nslab, mt, part, ntiles, ctrl, slot, gate_mode);                   \
    else                                                                                                            \
      hipLaunchKernelGGL((k_gram_lds<NW_, TPW_, NP_, RB_, false>), dim3(nslab), dim3(64 * NW_), lds, st, X, aux, ld, \
                         cols, w, rows_per_slab, nslab, mt, part, ntiles, ctrl, slot, gate_mode);                   \
  } while (0)
    if (mt <= 8)
      GL_GO(4, 9, 16, 64);
    else if (mt <= 12)
      GL_GO(8, 10, 12, 64);
    else
      GL_GO(8, 17, 8, 32);  // 32-row chunks: the accumulators leave no room for a 64-row prefetch
#undef GL_GO
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gram_reduce, dim3((ntiles * 256 + 15) / 16), dim3(256), 0, st, part, nslab, ntiles, Gt, ctrl,
                       slot, gate_mode);
    LAUNCH_CHECK();
    return hipSuccess;
  }
  long nwaves = (long)ntask * nslab;
  int nblk = (int)((nwaves + 3) / 4);
  if (w)
    hipLaunchKernelGGL(k_gram<true>, dim3(nblk), dim3(256), 0, st, X, aux, ld, cols, w, rows_per_slab, tasks, ntask,
                       nslab, part, ntiles, ctrl, slot, gate_mode, tile_base);
  else
    hipLaunchKernelGGL(k_gram<false>, dim3(nblk), dim3(256), 0, st, X, aux, ld, cols, w, rows_per_slab, tasks, ntask,
                       nslab, part, ntiles, ctrl, slot, gate_mode, tile_base);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gram_reduce, dim3((ntiles * 256 + 15) / 16), dim3(256), 0, st, part, nslab, ntiles, Gt, ctrl,
                     slot, gate_mode);
  LAUNCH_CHECK();
  return hipSuccess;
}

// LM Gram with the per-row-set cache.  tasks_full / tasks_inc: task lists for the whole lower triangle and for the
// extra tile row (I = mt); gbuf[0/1]: the two dense cache buffers, meta[1] says which one is current.
hipError_t launch_gram_lm_cached(const double *X, const double *aux, long ld, int *cols, const double *w,
                                 const int *A_new, int T0, int mt, const GramTask *tasks_full, int ntask_full,
                                 int rps_full, int nslab_full, const GramTask *tasks_inc, int ntask_inc, int rps_inc,
                                 int nslab_inc, double *part, double *Gt, double *Rt, int *src, double *gbuf0,
                                 double *gbuf1, int *Ac, int *meta, FitCtrl *ctrl, int slot, hipStream_t st) {
  const int mp = mt * 16, ntiles = mt * (mt + 1) / 2;
  hipLaunchKernelGGL(k_gram_plan, dim3(1), dim3(256), 0, st, A_new, T0, mp, (const int *)Ac, (const int *)meta, src,
                     cols, ctrl, slot);
  LAUNCH_CHECK();
  hipError_t e = launch_gram(X, aux, ld, cols, w, rps_full, tasks_full, ntask_full, nslab_full, part, ntiles, Gt, ctrl,
                             slot, 3, st, 0);
  if (e != hipSuccess) return e;
  e = launch_gram(X, aux, ld, cols, w, rps_inc, tasks_inc, ntask_inc, nslab_inc, part, mt, Rt, ctrl, slot, 4, st,
                  ntiles);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gram_assemble, dim3(ntiles), dim3(256), 0, st, Gt, (const double *)Rt, (const int *)src, T0,
                     gbuf0, gbuf1, (const int *)meta, (const FitCtrl *)ctrl, slot);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gram_cache_commit, dim3(1), dim3(256), 0, st, A_new, T0, Ac, meta, (const FitCtrl *)ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_chol(const double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                       const int *rhs_gather, double *sol, int *info, const FitCtrl *ctrl, int slot, int gate_mode,
                       hipStream_t st, const CholFuse *fuse, const IrlsChk *chk) {
  if (mt < 1 || m + 1 > mt * 16) return hipErrorInvalidValue;
  if (mt > CH_MT) return hipErrorInvalidValue;  // callers route larger systems to launch_chol_big
  CholFuse fz = {};
  if (fuse) fz = *fuse;
  IrlsChk ck = {};
  if (chk) ck = *chk;
  // register tiles per wave = ceil(mt (mt + 1) / 2 / 8): the smallest instance that fits (fewer live accumulators)
#define CHOL_GO(S)                                                                                                   \
  hipLaunchKernelGGL(k_chol<S>, dim3(1), dim3(512), 0, st, Gt, m, mt, ridge, ridge_skip0, rhs, rhs_gather, sol, info, \
                     ctrl, slot, gate_mode, fz, ck)
  if (mt <= 8)
    CHOL_GO(5);
  else if (mt <= 10)
    CHOL_GO(7);
  else if (mt <= 12)
    CHOL_GO(10);
  else if (mt <= 14)
    CHOL_GO(14);
  else
    CHOL_GO(17);
#undef CHOL_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_sym_fallback(const double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                               const int *rhs_gather, double *sol, int *info, const FitCtrl *ctrl, int slot,
                               hipStream_t st, const CholFuse *fuse) {
  if (mt < 1 || mt > CH_MT || fuse == nullptr || fuse->fb_work == nullptr || info == nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_sym_fallback, dim3(1), dim3(512), 0, st, Gt, m, mt, ridge, ridge_skip0, rhs, rhs_gather, sol,
                     info, ctrl, slot, *fuse);
  LAUNCH_CHECK();
  return hipSuccess;
}

// Conjugate-gradient solve of the covariance form (k_cg); falls back to k_chol through the parked-fit protocol.
#ifdef BESSX_CG_PROFILE
extern "C" __attribute__((visibility("default"))) int bessx_debug_cg_profile(unsigned long long *out, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cg_prof), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_cg_prof), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
#ifdef BESSX_KTRACE
extern "C" __attribute__((visibility("default"))) int bessx_debug_phase(unsigned long long *out, int reset) {
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
extern "C" __attribute__((visibility("default"))) int bessx_debug_ktrace(unsigned long long *out, int cap, int reset) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ktrace_n), sizeof(n)) != hipSuccess) return -1;
  const unsigned int m = n < (unsigned)cap ? n : (unsigned)cap;
  if (m && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktrace), (size_t)m * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ktrace_n), &z, sizeof(z)) != hipSuccess) return -1;
  }
  return (int)m;
}
#endif
hipError_t launch_cg(int m, int mt, double ridge, const double *rhs, const int *A_new, double *sol, const FitCtrl *ctrl,
                     int slot, const CholFuse *fuse, int maxit, hipStream_t st, double tol, bool by_rows) {
  if (mt < 1 || m > mt * 16 || mt > CH_MT || fuse == nullptr) return hipErrorInvalidValue;
  const CholFuse fz = *fuse;
  if (by_rows && m <= 208) {
    const int nc = (m + 7) / 8;
#define CGR_GO(RP, NW_)                                                                                           \
  hipLaunchKernelGGL((k_cgr<RP, NW_>), dim3(1), dim3(512), 0, st, m, nc, ridge, rhs, A_new, sol, ctrl, slot, fz, \
                     maxit, tol)
    if (m <= 64)
      CGR_GO(1, 8);
    else if (m <= 128)
      CGR_GO(2, 16);
    else if (m <= 192)
      CGR_GO(3, 24);
    else
      CGR_GO(4, 26);
#undef CGR_GO
    LAUNCH_CHECK();
    return hipSuccess;
  }
#define CG_GO(S, W)                                                                                                  \
  hipLaunchKernelGGL((k_cg<S, W>), dim3(1), dim3(64 * W), 0, st, m, mt, ridge, rhs, A_new, sol, ctrl, slot, fz, maxit, \
                     tol)
  // (a one-wave instance for <= 64 unknowns, CG_GO(10, 1), was measured slower: the gather and the tile loop
  // serialise)
  if (mt <= 8)
    CG_GO(5, 8);
  else if (mt <= 10)
    CG_GO(7, 8);
  else if (mt <= 12)
    CG_GO(10, 8);
  else if (mt <= 14)
    CG_GO(14, 8);
  else
    CG_GO(17, 8);
#undef CG_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

// selection + solve in one launch (k_sel_cgr): scores in one chunk of at most 32768, systems of at most 208 unknowns
bool sel_cgr_applies(int len, int m) { return len <= 512 * 64 && m >= 1 && m <= 208; }
hipError_t launch_sel_cgr(const double *score, int len, int k, int *A_new, const FitCtrl *ctrl, int slot,
                          const TopkNeed *need, double ridge, const double *rhs, double *sol, const CholFuse *fuse,
                          int maxit, hipStream_t st, double tol) {
  if (!sel_cgr_applies(len, k) || need == nullptr || fuse == nullptr) return hipErrorInvalidValue;
  const TopkNeed nd = *need;
  const CholFuse fz = *fuse;
  const int nblk = nd.pub.on ? 2 : 1, nc = (k + 7) / 8, per = (len + 511) / 512;
#define SC_GO(EB, RP, NW_)                                                                                      \
  hipLaunchKernelGGL((k_sel_cgr<EB, RP, NW_>), dim3(nblk), dim3(512), 0, st, score, len, k, A_new, ctrl, slot, nd, nc, \
                     ridge, rhs, sol, fz, maxit, tol)
#define SC_BY_M(EB)          \
  do {                       \
    if (k <= 64)             \
      SC_GO(EB, 1, 8);       \
    else if (k <= 128)       \
      SC_GO(EB, 2, 16);      \
    else if (k <= 192)       \
      SC_GO(EB, 3, 24);      \
    else                     \
      SC_GO(EB, 4, 26);      \
  } while (0)
  if (per <= 8)
    SC_BY_M(8);
  else if (per <= 24)
    SC_BY_M(24);
  else
    SC_BY_M(64);
#undef SC_BY_M
#undef SC_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

// Blocked Cholesky for m + 1 > 256.  Gt is overwritten by its factor; rdiag (>= 16*mt) and z (>= 16*mt) are work space.
hipError_t launch_chol_big(double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                           const int *rhs_gather, double *sol, int *info, double *rdiag, double *z,
                           const FitCtrl *ctrl, int slot, int gate_mode, hipStream_t st) {
  if (mt < 1 || m + 1 > mt * 16) return hipErrorInvalidValue;
  const int ntiles = mt * (mt + 1) / 2;
  hipLaunchKernelGGL(k_bc_prepare, dim3(ntiles), dim3(256), 0, st, Gt, m, mt, ridge, ridge_skip0, rhs, rhs_gather,
                     ctrl, slot, gate_mode);
  LAUNCH_CHECK();
  for (int b = 0; b < mt; b++) {
    hipLaunchKernelGGL(k_bc_panel, dim3(mt - b), dim3(64), 0, st, Gt, rdiag, mt, b, ctrl, slot, gate_mode);
    LAUNCH_CHECK();
    const int nt = (mt - b - 1) * (mt - b) / 2;
    if (nt > 0) {
      hipLaunchKernelGGL(k_bc_update, dim3(nt), dim3(64), 0, st, Gt, mt, b, ctrl, slot, gate_mode);
      LAUNCH_CHECK();
    }
  }
  for (int b = mt - 1; b >= 0; b--) {
    hipLaunchKernelGGL(k_bc_back, dim3(1), dim3(512), 0, st, (const double *)Gt, (const double *)rdiag, z, mt, b, m,
                       sol, info, ctrl, slot, gate_mode);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

hipError_t launch_fit_begin(FitCtrl *ctrl, int T0, int k_init, const int *init_idx, const double *init_val,
                            double coef0_init, int *A_cur, double *b_cur, double *beta_dense, int p, int *hist,
                            hipStream_t st, unsigned char *inA) {
  hipError_t e = hipMemsetAsync(beta_dense, 0, (size_t)p * sizeof(double), st);
  if (e == hipSuccess && inA != nullptr) e = hipMemsetAsync(inA, 0, (size_t)p, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fit_begin, dim3(1), dim3(256), 0, st, ctrl, T0, k_init, init_idx, init_val, coef0_init, A_cur,
                     b_cur, beta_dense, hist, inA);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_fit_continue(FitCtrl *ctrl, int T0, int *hist, hipStream_t st, int serial, int chained,
                               int parent) {
  hipLaunchKernelGGL(k_fit_continue, dim3(1), dim3(256), 0, st, ctrl, T0, hist, serial, chained, parent);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_commit(FitCtrl *ctrl, int slot, int T0, const int *A_new, const double *sol, int has_intercept,
                         int wait_chain, int *A_cur, double *b_cur, double *beta_dense, int *hist, double *hist_beta,
                         double *hist_coef0, int hist_stride, hipStream_t st, unsigned char *inA) {
  hipLaunchKernelGGL(k_commit, dim3(1), dim3(256), 0, st, ctrl, slot, T0, A_new, sol, has_intercept, wait_chain, A_cur,
                     b_cur, beta_dense, hist, hist_beta, hist_coef0, hist_stride, inA);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_resid_lm(const double *X, long ld, int n, const double *y, const double *mask,
                           const FitCtrl *ctrl, int when, const int *A_cur, const double *b_cur, double *r,
                           double *sse, hipStream_t st, int mode, int kc_given, double c0_given) {
  int nblk = (int)((ld + 255) / 256);
  hipLaunchKernelGGL(k_resid_lm, dim3(nblk), dim3(512), 0, st, X, ld, n, y, mask, ctrl, when, A_cur, b_cur, r, sse,
                     mode, kc_given, c0_given);
  LAUNCH_CHECK();
  return hipSuccess;
}

template <int FAM>
static hipError_t launch_glm_eta_gh_t(const double *X, long ld, int n, const double *y, const double *w,
                                      const double *mask, const double *logfact, const FitCtrl *ctrl, int when,
                                      const int *A_cur, const double *b_cur, double *g, double *h, double *stats,
                                      hipStream_t st) {
  int nblk = (int)((ld + 255) / 256);
  hipLaunchKernelGGL(k_glm_eta_gh<FAM>, dim3(nblk), dim3(128), 0, st, X, ld, n, y, w, mask, logfact, ctrl, when,
                     A_cur, b_cur, g, h, stats);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_glm_eta_gh(int fam, const double *X, long ld, int n, const double *y, const double *w,
                             const double *mask, const double *logfact, const FitCtrl *ctrl, int when,
                             const int *A_cur, const double *b_cur, double *g, double *h, double *stats,
                             hipStream_t st) {
  return fam == 2 ? launch_glm_eta_gh_t<2>(X, ld, n, y, w, mask, logfact, ctrl, when, A_cur, b_cur, g, h, stats, st)
                  : launch_glm_eta_gh_t<3>(X, ld, n, y, w, mask, logfact, ctrl, when, A_cur, b_cur, g, h, stats, st);
}

hipError_t launch_glm_irls_begin(const FitCtrl *ctrl, int slot, int fam, int m, double *bcur, double *bprev,
                                 hipStream_t st) {
  hipLaunchKernelGGL(k_glm_irls_begin, dim3(1), dim3(256), 0, st, ctrl, slot, fam, m, bcur, bprev);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_glm_irls_prep(int fam, const double *X, long ld, int n, const double *y, const double *w,
                                const double *mask, const FitCtrl *ctrl, int slot, int t, const int *A_new, int T0,
                                const double *bcur, double *Wv, double *z, double *llpart, hipStream_t st,
                                int wfloor) {
  int nblk = (int)((ld + 255) / 256);
  if (fam == 2)
    hipLaunchKernelGGL(k_glm_irls_prep<2>, dim3(nblk), dim3(128), 0, st, X, ld, n, y, w, mask, ctrl, slot, t, A_new,
                       T0, bcur, Wv, z, llpart, wfloor);
  else
    hipLaunchKernelGGL(k_glm_irls_prep<3>, dim3(nblk), dim3(128), 0, st, X, ld, n, y, w, mask, ctrl, slot, t, A_new,
                       T0, bcur, Wv, z, llpart, wfloor);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_glm_irls_check(FitCtrl *ctrl, int slot, int t, int fam, const double *llpart, int nblk, int m,
                                 double *bcur, double *bprev, hipStream_t st) {
  hipLaunchKernelGGL(k_glm_irls_check, dim3(1), dim3(256), 0, st, ctrl, slot, t, fam, llpart, nblk, m, bcur, bprev);
  LAUNCH_CHECK();
  return hipSuccess;
}

static hipError_t launch_scan3(const double *in0, const double *in1, const double *in2, double *out0, double *out1,
                               double *out2, double *recip, long n, int suffix, int nvec, double *scr,
                               const FitCtrl *ctrl, int gate, int slot, int t, hipStream_t st) {
  const int nb = (int)((n + SC_B - 1) / SC_B);
  hipLaunchKernelGGL(k_scan3_tot, dim3(nb), dim3(SC_T), 0, st, in0, in1, in2, n, suffix, nvec, scr, ctrl, gate, slot, t);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan3_apply, dim3(nb), dim3(SC_T), 0, st, in0, in1, in2, out0, out1, out2, recip, n, suffix,
                     nvec, (const double *)scr, ctrl, gate, slot, t);
  LAUNCH_CHECK();
  return hipSuccess;
}

size_t cox_scan_scratch_doubles(long ld, int kmax) {
  return (size_t)(2 * std::max(kmax, 8) + 16) * (size_t)((ld + SC_B - 1) / SC_B);
}

hipError_t launch_cox_state(const double *X, long ld, int n, const double *y, const double *w, const double *mask,
                            const FitCtrl *ctrl, int when, const int *A_cur, const double *b_cur, CoxBufs cb,
                            double *stats, hipStream_t st) {
  int nblk = (int)((ld + 255) / 256);
  hipLaunchKernelGGL(k_cox_eta, dim3(nblk), dim3(128), 0, st, X, ld, n, y, w, mask, ctrl, when, A_cur, b_cur, cb.E,
                     cb.TH, cb.ET, cb.EW, cb.WD);
  LAUNCH_CHECK();
  {
    hipError_t es = launch_scan3(cb.TH, cb.E, cb.ET, cb.S0, cb.SALL, cb.STEST, cb.RS0, (long)n, 1, mask ? 3 : 2, cb.SCR,
                                 ctrl, 1, when, 0, st);
    if (es != hipSuccess) return es;
  }
  hipLaunchKernelGGL(k_cox_loss, dim3(nblk), dim3(128), 0, st, ld, n, y, w, mask, ctrl, when, (const double *)cb.E,
                     (const double *)cb.SALL, (const double *)cb.STEST, stats);
  LAUNCH_CHECK();
  if (cb.one_pass || cb.need_uv) {  // vectors of the one-pass score (k_cox_score1p) / of the group branch of get_A
    const int nb2 = (int)((ld + 255) / 256);
    hipLaunchKernelGGL(k_cox_c1, dim3(nb2), dim3(256), 0, st, ld, (const double *)cb.EW, (const double *)cb.RS0, cb.CV,
                       ctrl, when);  // CV holds ew / S0 until k_cox_uv overwrites it
    LAUNCH_CHECK();
    hipError_t es = launch_scan3(cb.CV, nullptr, nullptr, cb.C1, nullptr, nullptr, nullptr, (long)n, 0, 1, cb.SCR, ctrl,
                                 1, when, 0, st);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(k_cox_uv, dim3(nb2), dim3(256), 0, st, ld, (const double *)cb.EW, (const double *)cb.RS0,
                       (const double *)cb.TH, (const double *)cb.C1, cb.CU, cb.CV, cb.C2, ctrl, when);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

static int g_cox_score_variant = 1;  // the wave -> (column group, row block) map (tools/cox_score_bench.py measures both)
void cox_score_set_variant(int v) { g_cox_score_variant = v & 1; }

hipError_t launch_cox_score_pass(const double *X, long ld, int p, int U, int nrb, CoxBufs cb, double *part,
                                 double *part2, const FitCtrl *ctrl, int slot, hipStream_t st) {
  if (cb.one_pass) {
    long nw = (long)nrb * ((p + 63) / 64);
    int nb = (int)((nw + 3) / 4);
#define CS1_GO3(UU, MM)                                                                                        \
  hipLaunchKernelGGL((k_cox_score1p<UU, MM>), dim3(nb), dim3(256), 0, st, X, ld, p, nrb, (const double *)cb.TH, \
                     (const double *)cb.CU, (const double *)cb.CV, (const double *)cb.C2, part, ctrl, slot)
#define CS1_GO(UU)               \
  if (g_cox_score_variant)       \
    CS1_GO3(UU, 1);              \
  else                           \
    CS1_GO3(UU, 0)
    switch (U) {
      case 8: CS1_GO(8); break;
      case 4: CS1_GO(4); break;
      case 2: CS1_GO(2); break;
      default: CS1_GO(1); break;
    }
#undef CS1_GO
#undef CS1_GO3
    LAUNCH_CHECK();
    return hipSuccess;
  }
  hipError_t e = launch_xtv(X, ld, p, U, cb.TH, cb.TH, part, part2, ctrl, slot, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cox_carry, dim3((p + 255) / 256), dim3(256), 0, st, part, part2, nrb, p, ctrl, slot);
  LAUNCH_CHECK();
  long nwaves = (long)nrb * ((p + 63) / 64);
  int nblk = (int)((nwaves + 3) / 4);
#define CS_GO(UU)                                                                                            \
  hipLaunchKernelGGL(k_cox_colscan<UU>, dim3(nblk), dim3(256), 0, st, X, ld, p, nrb, (const double *)cb.TH, \
                     (const double *)cb.RS0, (const double *)cb.EW, part, part2, ctrl, slot)
  switch (U) {
    case 8: CS_GO(8); break;
    case 4: CS_GO(4); break;
    case 2: CS_GO(2); break;
    default: CS_GO(1); break;
  }
#undef CS_GO
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cox_score(const double *part, const double *part2, int nrb, int p, const double *beta_dense,
                            double lambda, const unsigned char *always, double *bd, const FitCtrl *ctrl, int slot,
                            hipStream_t st) {
  if (part2 == nullptr)  // one-pass layout (k_cox_score1p)
    hipLaunchKernelGGL(k_cox_score_1p, dim3((p + 63) / 64), dim3(512), 0, st, part, nrb, p, beta_dense, lambda,
                       always, bd, ctrl, slot);
  else
    hipLaunchKernelGGL(k_cox_score, dim3((p + 255) / 256), dim3(256), 0, st, part, part2, nrb, p, beta_dense, lambda,
                       always, bd, ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cox_newton_begin(FitCtrl *ctrl, int slot, int k, CoxBufs cb, int *idcols, hipStream_t st) {
  const int mp = (k + 1 + 15) / 16 * 16;
  hipLaunchKernelGGL(k_cox_newton_begin, dim3(1), dim3(256), 0, st, ctrl, slot, k, mp, cb.b0, idcols);
  LAUNCH_CHECK();
  return hipSuccess;
}

// slab geometry of the one-pass Hessian kernel: about one slab per compute unit, whole 64-row chunks
int cox_hess_slab_rows(long ld) { return (int)(((ld + 255) / 256 + 63) / 64 * 64); }
bool cox_hess_applies(int mt) { return mt >= 1 && mt <= 10; }  // (beyond: two accumulator sets no longer fit the registers)
hipError_t cox_hess_prepare() {
  hipError_t e = hipSuccess;
  const int big = (10 * 16 * 66 + 4 * 64 + 10 * 16) * (int)sizeof(double);
#define CH_ATTR(K) \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, big)
  CH_ATTR((k_cox_hess<2, 4>));
  CH_ATTR((k_cox_hess<5, 8>));
  CH_ATTR((k_cox_hess<7, 10>));
#undef CH_ATTR
  return e;
}

hipError_t launch_cox_newton_step(const double *X, const double *aux, long ld, int n, const double *mask,
                                  FitCtrl *ctrl, int slot, int t, const int *A_new, int k, double lambda,
                                  const int *gcols, const int *idcols, int mt, const GramTask *tasks, int ntask,
                                  int rps, int nslab, double *gpart, int ntiles, double *Gt, CoxBufs cb,
                                  hipStream_t st, double *rdiag, double *zbig) {
  const int nb2 = (int)((ld + 255) / 256);
  const bool fused = cb.hess_fused && cox_hess_applies(mt);
  if (fused) {
    // the n-vector work in three launches; SCR: block totals of theta, then of w delta / S0
    const int nbl = (int)((ld + SC_B - 1) / SC_B);
    double *totA = cb.SCR, *totB = cb.SCR + nbl;
    hipLaunchKernelGGL(k_cox_nvecA, dim3(nbl), dim3(SC_T), 0, st, ld, n, mask, (const FitCtrl *)ctrl, slot, t,
                       (const double *)cb.UD, cb.ETA0, cb.THF, cb.fit_clamp, totA);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cox_nvecB, dim3(nbl), dim3(SC_T), 0, st, ld, n, (const double *)cb.THF, (const double *)cb.WD,
                       (const double *)totA, cb.S0F, cb.RS0F, totB, (const FitCtrl *)ctrl, slot, t);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cox_nvecC, dim3(nbl), dim3(SC_T), 0, st, ld, n, (const double *)cb.THF, (const double *)cb.WD,
                       (const double *)cb.RS0F, (const double *)totB, cb.VG, cb.WG1, const_cast<double *>(aux) + 2 * ld,
                       cb.CW, (const FitCtrl *)ctrl, slot, t);
    LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(k_cox_fit_eta, dim3(nb2), dim3(128), 0, st, X, ld, n, mask, (const FitCtrl *)ctrl, slot, t,
                       A_new, k, (const double *)cb.b0, cb.ETA0, cb.THF, cb.fit_clamp);
    LAUNCH_CHECK();
    hipError_t es = launch_scan3(cb.THF, nullptr, nullptr, cb.S0F, nullptr, nullptr, cb.RS0F, (long)n, 1, 1, cb.SCR,
                                 (const FitCtrl *)ctrl, 2, slot, t, st);
    if (es != hipSuccess) return es;
  }
  {
    const int nbn = (int)(((long)n + SC_B - 1) / SC_B), nbl = (int)((ld + SC_B - 1) / SC_B);
    if (!fused) {
      hipLaunchKernelGGL(k_cox_cscan_tot, dim3(nbl), dim3(SC_T), 0, st, (const double *)cb.WD, (const double *)cb.RS0F,
                         (long)n, cb.SCR, (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(k_cox_cscan_apply, dim3(nbl), dim3(SC_T), 0, st, (const double *)cb.WD, (const double *)cb.RS0F,
                         (const double *)cb.THF, cb.VG, cb.WG1, (long)n, ld, (const double *)cb.SCR,
                         (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
    }
    if (fused) {
      const int hrows = cox_hess_slab_rows(ld), hns = (int)((ld + hrows - 1) / hrows), mp = mt * 16;
      const size_t lds = ((size_t)mp * 66 + 4 * 64 + mp) * sizeof(double);
      if (mt <= 4)
        hipLaunchKernelGGL((k_cox_hess<2, 4>), dim3(hns), dim3(512), lds, st, X, aux, ld, gcols, (const double *)cb.WG1,
                           (const double *)cb.CW, (const double *)cb.THF, hrows, mt, k, gpart, cb.HP2, cb.HT, ntiles,
                           (const FitCtrl *)ctrl, slot, t);
      else if (mt <= 8)
        hipLaunchKernelGGL((k_cox_hess<5, 8>), dim3(hns), dim3(512), lds, st, X, aux, ld, gcols, (const double *)cb.WG1,
                           (const double *)cb.CW, (const double *)cb.THF, hrows, mt, k, gpart, cb.HP2, cb.HT, ntiles,
                           (const FitCtrl *)ctrl, slot, t);
      else
        hipLaunchKernelGGL((k_cox_hess<7, 10>), dim3(hns), dim3(512), lds, st, X, aux, ld, gcols,
                           (const double *)cb.WG1, (const double *)cb.CW, (const double *)cb.THF, hrows, mt, k, gpart,
                           cb.HP2, cb.HT, ntiles, (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
      if (hns > 256) return hipErrorInvalidValue;
      hipLaunchKernelGGL(k_cox_car, dim3(mp), dim3(256), 0, st, (const double *)cb.HT,
                         (const double *)cb.HP2, hns, mt, k, ntiles, cb.CAR, cb.HQ, (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(k_cox_hess_reduce, dim3((ntiles * 256 + 15) / 16), dim3(256), 0, st, (const double *)gpart,
                         (const double *)cb.HP2, (const double *)cb.CAR, (const double *)cb.HQ, hns, ntiles, mp, k,
                         lambda, (const double *)cb.b0, Gt, cb.g, (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(k_cox_M_tot, dim3(nbn, k), dim3(SC_T), 0, st, X, ld, (long)n, A_new, (const double *)cb.THF,
                         (const double *)cb.VG, cb.SCR, (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(k_cox_M_apply, dim3(nbn, k), dim3(SC_T), 0, st, X, ld, (long)n, A_new, (const double *)cb.THF,
                         (const double *)cb.RS0F, (const double *)cb.b0, lambda, (const double *)cb.SCR, cb.M, cb.g,
                         (const FitCtrl *)ctrl, slot, t);
      LAUNCH_CHECK();
    }
  }
  hipError_t e = hipSuccess;
  if (!fused) {
    // Hessian: -h = X_A^T diag(theta C) X_A - M^T diag(w delta) M  (from :1458-1470)
    e = launch_gram(X, aux, ld, gcols, cb.WG1, rps, tasks, ntask, nslab, gpart, ntiles, Gt, ctrl, slot, 2, st, 0);
    if (e != hipSuccess) return e;
    e = launch_gram(cb.M, aux, ld, idcols, cb.WD, rps, tasks, ntask, nslab, gpart, ntiles, cb.Gt2, ctrl, slot, 2, st, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tile_sub, dim3((ntiles * 256 + 255) / 256), dim3(256), 0, st, Gt, (const double *)cb.Gt2,
                       (long)ntiles * 256, (const FitCtrl *)ctrl, slot, t);
    LAUNCH_CHECK();
  }
  // lambda = 0: a collapsed pivot (exactly dependent active columns) goes to the pivoted solve inside k_chol; with a
  // ridge the matrix G1 - G2 - 2 lambda I can be indefinite and the un-pivoted LDL^T below follows the oracle
  CholFuse fbz = {};
  fbz.fb_work = lambda == 0.0 ? cb.ldl_work : nullptr;
  e = mt <= CH_MT ? launch_chol(Gt, k, mt, -2.0 * lambda, 0, cb.g, nullptr, cb.u, &ctrl->info, ctrl, slot, 2, st, &fbz)
                  : launch_chol_big(Gt, k, mt, -2.0 * lambda, 0, cb.g, nullptr, cb.u, &ctrl->info, rdiag, zbig, ctrl, slot,
                                    2, st);
  if (e != hipSuccess) return e;
  if (e == hipSuccess && mt <= CH_MT && lambda == 0.0 && cb.ldl_work != nullptr)
    e = launch_sym_fallback(Gt, k, mt, 0.0, 0, cb.g, nullptr, cb.u, &ctrl->info, ctrl, slot, st, &fbz);
  if (e != hipSuccess) return e;
  if (mt <= CH_MT && lambda != 0.0 && cb.ldl_work != nullptr) {  // (k_chol leaves Gt untouched: it works in registers)
    hipLaunchKernelGGL(k_ldlt_fallback, dim3(1), dim3(256), 0, st, (const double *)Gt, k, -2.0 * lambda,
                       (const double *)cb.g, cb.u, &ctrl->info, (const FitCtrl *)ctrl, slot, t, cb.ldl_work);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_cox_dir, dim3(nb2), dim3(128), 0, st, X, ld, (const FitCtrl *)ctrl, slot, t, A_new, k,
                     (const double *)cb.u, cb.UD);
  LAUNCH_CHECK();
  {
    const int nbs = (int)(((long)n + LS_B - 1) / LS_B);
    hipLaunchKernelGGL(k_cox_ls5_tot, dim3(nbs), dim3(SC_T), 0, st, (long)n, mask, (const double *)cb.ETA0,
                       (const double *)cb.UD, cb.SCR, (const FitCtrl *)ctrl, slot, t);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cox_ls5_apply, dim3(nbs), dim3(SC_T), 0, st, (long)n, mask, (const double *)cb.ETA0,
                       (const double *)cb.UD, (const double *)cb.WD, (const double *)cb.SCR, cb.SCR + (size_t)5 * nbs,
                       (const FitCtrl *)ctrl, slot, t);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cox_ls5_check, dim3(1), dim3(256), 0, st, ctrl, slot, t, k, (const double *)(cb.SCR + (size_t)5 * nbs),
                       nbs, cb.b0, (const double *)cb.u);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

hipError_t launch_group_moments(int smax, const double *X, long ld, int n, const double *w1, const double *w2, int N,
                                const int *gidx, const int *gsz, const int *goff, double *mblk, double *dcol,
                                hipStream_t st, int cshift) {
#define GM_GO(S)                                                                                                  \
  hipLaunchKernelGGL(k_group_moments<S>, dim3(N), dim3(256), 0, st, X, ld, n, w1, w2, gidx, gsz, goff, mblk, dcol, \
                     cshift)
  if (smax <= 2)
    GM_GO(2);
  else if (smax <= 4)
    GM_GO(4);
  else if (smax <= 8)
    GM_GO(8);
  else
    GM_GO(16);
#undef GM_GO
  LAUNCH_CHECK();
  if (smax > GRP_MAX) {  // the wide groups, tile by tile (blocks of narrow groups / surplus tiles return at once)
    const int nt = (smax + GB_T - 1) / GB_T;
    hipLaunchKernelGGL(k_group_moments_big, dim3(N, nt * (nt + 1) / 2), dim3(256), 0, st, X, ld, n, w1, w2, gidx, gsz,
                       goff, mblk, dcol, cshift);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

hipError_t launch_group_lsq_score(int N, const int *gidx, const int *gsz, const int *goff, const double *mblk,
                                  const double *dcol, const unsigned char *always, double *work, double *zwork,
                                  double *score, hipStream_t st) {
  hipLaunchKernelGGL(k_group_lsq_score, dim3(N), dim3(256), 0, st, N, gidx, gsz, goff, mblk, dcol, always, work, zwork,
                     score);
  LAUNCH_CHECK();
  return hipSuccess;
}

__global__ void __launch_bounds__(256) k_iota(int *__restrict__ a, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void __launch_bounds__(256) k_vec_sub(double *__restrict__ a, const double *__restrict__ b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] -= b[i];
}
hipError_t launch_iota(int *a, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
  LAUNCH_CHECK();
  return hipSuccess;
}

// GroupPdasCox::get_A, group branch (src/Algorithm.h:1497-1568) without the n x n Hessian.  With theta = w exp(eta),
// S0 its suffix sums, c2_i = sum_{k<=i} y_k w_k / S0_k:  h = diag(c2 theta) - [c3(min(i,j)) theta_i theta_j], and
// sum_{i,j} x_i c3(min(i,j)) theta_i theta_j x_j^T = sum_m (y_m w_m / S0_m^2) S1(m) S1(m)^T with S1 the suffix sums
// of theta x (exchange of the order of summation, as in the singleton branch).  So per group
//   X_g^T h X_g = X_g^T diag(u) X_g - M_g^T diag(y w) M_g,   u = theta c2,  M = S1 / S0,
//   d = X^T (y w - u) - 2 lambda beta   (:1547-1548),
// i.e. two passes of k_group_moments: over X with (u, y w - u), and over the suffix-sum matrix M of one panel of
// whole groups (at most `mcols` columns) at a time with weights y w.  u and y w - u are the vectors CU and CV of the
// one-pass score (status in {0, 1}: w [delta != 0] = w delta).
hipError_t launch_cox_group_moments(const double *X, long ld, int n, int p, CoxBufs cb, const int *allcols, int mcols,
                                    int smax, int N, const int *gidx_h, const int *gsz_h, const int *gidx,
                                    const int *gsz, const int *goff, long mblk_len, double *mblk, double *mblk2,
                                    double *dcol, hipStream_t st) {
  hipError_t e = launch_group_moments(smax, X, ld, n, cb.CU, cb.CV, N, gidx, gsz, goff, mblk, dcol, st, 0);
  if (e != hipSuccess) return e;
  const int nb = (int)((n + SC_B - 1) / SC_B);
  for (int g0 = 0; g0 < N;) {
    int g1 = g0, np = 0;
    while (g1 < N && np + gsz_h[g1] <= mcols) np += gsz_h[g1++];
    if (g1 == g0) return hipErrorInvalidValue;  // a group wider than the work space
    const int j0 = gidx_h[g0];
    hipLaunchKernelGGL(k_cox_M_tot, dim3(nb, np), dim3(SC_T), 0, st, X, ld, (long)n, allcols + j0,
                       (const double *)cb.TH, (const double *)cb.VG, cb.SCR, (const FitCtrl *)nullptr, 0, 0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cox_M_apply, dim3(nb, np), dim3(SC_T), 0, st, X, ld, (long)n, allcols + j0,
                       (const double *)cb.TH, (const double *)cb.RS0, (const double *)cb.b0, 0.0,
                       (const double *)cb.SCR, cb.M, cb.g, (const FitCtrl *)nullptr, 0, 0);
    LAUNCH_CHECK();
    e = launch_group_moments(smax, cb.M, ld, n, cb.WD, nullptr, g1 - g0, gidx + g0, gsz + g0, goff + g0, mblk2, nullptr,
                             st, j0);
    if (e != hipSuccess) return e;
    g0 = g1;
  }
  hipLaunchKernelGGL(k_vec_sub, dim3((int)((mblk_len + 255) / 256)), dim3(256), 0, st, mblk, (const double *)mblk2,
                     mblk_len);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_group_score(int N, const int *gidx, const int *gsz, const int *goff, const double *mblk,
                              const double *dcol, const double *part, int nrb, int p, int lm, double n_t,
                              double lambda, const double *beta_dense, const unsigned char *always, double *bd,
                              hipStream_t st, int smax, double *work, double *zwork, const FitCtrl *ctrl, int slot) {
  hipLaunchKernelGGL(k_group_score, dim3((N + 63) / 64), dim3(64), 0, st, N, gidx, gsz, goff, mblk, dcol, part, nrb, p,
                     lm, n_t, lambda, beta_dense, always, bd, ctrl, slot);
  LAUNCH_CHECK();
  if (smax > GRP_MAX) {
    if (work == nullptr || zwork == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_group_score_big, dim3(N), dim3(256), 0, st, N, gidx, gsz, goff, mblk, dcol, part, nrb, p, lm,
                       n_t, lambda, beta_dense, always, work, zwork, bd, ctrl, slot);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

// find_ind (src/utilities.cpp:113-130) on the device for groups of ONE width gs: the T0 selected groups (ascending) ->
// their T0 * gs columns, in order (all p columns when every group is selected: the same formula).  Gated like the
// kernels around it, so that a PDAS iteration of a grouped fit needs no host round trip between selection and fit.
__global__ void __launch_bounds__(256) k_group_expand(const int *__restrict__ G_sel, int T0, int gs,
                                                      const int *__restrict__ gidx, int *__restrict__ cols,
                                                      const FitCtrl *__restrict__ ctrl, int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T0 * gs) return;
  cols[i] = gidx[G_sel[i / gs]] + i % gs;
}

hipError_t launch_group_expand(const int *G_sel, int T0, int gs, const int *gidx, int *cols, const FitCtrl *ctrl,
                               int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_group_expand, dim3((T0 * gs + 255) / 256), dim3(256), 0, st, G_sel, T0, gs, gidx, cols, ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_commit_group(FitCtrl *ctrl, int slot, int T0, const int *G_new, int K, const int *cols,
                               const double *sol, int has_intercept, int wait_chain, int *A_cur, double *b_cur,
                               double *beta_dense, int *hist, double *hist_beta, double *hist_coef0, int hist_stride,
                               hipStream_t st) {
  hipLaunchKernelGGL(k_commit_group, dim3(1), dim3(256), 0, st, ctrl, slot, T0, G_new, K, cols, sol, has_intercept,
                     wait_chain, A_cur, b_cur, beta_dense, hist, hist_beta, hist_coef0, hist_stride);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_screen_score_lm(const double *sxy, const double *sxx, int p, const unsigned char *always,
                                  double *score, hipStream_t st) {
  hipLaunchKernelGGL(k_screen_score_lm, dim3((p + 255) / 256), dim3(256), 0, st, sxy, sxx, p, always, score);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_screen_logit(const double *X, long ld, int n, int p, const double *y, const double *w,
                               double *state, int *done, const unsigned char *always, double *score, hipStream_t st) {
  hipError_t e = hipMemsetAsync(state, 0, (size_t)p * 5 * sizeof(double), st);
  if (e == hipSuccess) e = hipMemsetAsync(done, 0, (size_t)p * sizeof(int), st);
  if (e != hipSuccess) return e;
  for (int t = 0; t <= 30; t++) {  // the solve before the loop + 30 loop iterations (src/logistic.cpp:135-155)
    hipLaunchKernelGGL(k_screen_logit_pass, dim3(p), dim3(256), 0, st, X, ld, n, y, w, t, state, done);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_screen_score_logit, dim3((p + 255) / 256), dim3(256), 0, st, (const double *)state, p, always,
                     score);
  LAUNCH_CHECK();
  return hipSuccess;
}

bool screen_logit_group_supported(int gmax) { return gmax <= SGL_MAX; }
size_t screen_logit_group_state_doubles(int N) { return (size_t)N * SGL_ST; }
hipError_t launch_screen_logit_group(const double *X, long ld, int n, int N, const int *gidx, const int *gsz,
                                     const double *y, const double *w, double *state, int *done,
                                     const unsigned char *always, double *score, hipStream_t st) {
  hipError_t e = hipMemsetAsync(state, 0, (size_t)N * SGL_ST * sizeof(double), st);
  if (e == hipSuccess) e = hipMemsetAsync(done, 0, (size_t)N * sizeof(int), st);
  if (e != hipSuccess) return e;
  for (int t = 0; t <= 30; t++) {  // the solve before the loop + 30 loop iterations (src/logistic.cpp:135-155)
    hipLaunchKernelGGL(k_screen_logit_group, dim3(N), dim3(256), 0, st, X, ld, n, y, w, gidx, gsz, t, state, done);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_screen_score_logit_group, dim3((N + 255) / 256), dim3(256), 0, st, (const double *)state, N, gsz,
                     always, score);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_screen_cox(const double *X, long ld, int n, int p, const double *st_, const double *w,
                             const unsigned char *always, double *score, hipStream_t st) {
  hipLaunchKernelGGL(k_screen_cox, dim3(p), dim3(256), 0, st, X, ld, n, st_, w, always, score);
  LAUNCH_CHECK();
  return hipSuccess;
}

bool screen_cox_group_supported(int gmax) { return gmax <= SCG_MAX; }
hipError_t launch_screen_cox_group(const double *X, long ld, int n, int N, const int *gidx, const int *gsz,
                                   const double *st_, const double *w, const unsigned char *always, double *score,
                                   hipStream_t st) {
  hipLaunchKernelGGL(k_screen_cox_group, dim3(N), dim3(256), 0, st, X, ld, n, st_, w, gidx, gsz, always, score);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_gather_cols(const double *X, long ld, const int *A, int pnew, double *X2, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((ld / 2 + 255) / 256), pnew), dim3(256), 0, st, X, ld, A, X2);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_vec_mul(const double *a, const double *b, long n, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_vec_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n, out);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_part_sum(const double *part, int nrb, int p, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_part_sum, dim3((p + 255) / 256), dim3(256), 0, st, part, nrb, p, out);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_fill(double *a, long n, double v, hipStream_t st) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, n, v);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_gram_cols(const int *A_new, int T0, int mp, int intercept, int rhs_col, int *cols, FitCtrl *ctrl,
                            int slot, const int *A_cur, int allow_skip, hipStream_t st) {
  hipLaunchKernelGGL(k_gram_cols, dim3(1), dim3(256), 0, st, A_new, T0, mp, intercept, rhs_col, cols, ctrl, slot,
                     A_cur, allow_skip);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_need(const int *list, int len, const double *bd, double *bd2, int p, int *slot_of, int *meta,
                           int C, int *fcols, FitCtrl *ctrl, int slot, const int *A_cur, hipStream_t st,
                           int no_restart) {
  hipLaunchKernelGGL(k_cov_need, dim3(1), dim3(256), 0, st, list, len, bd, bd2, p, slot_of, meta, C, fcols, ctrl, slot,
                     A_cur, no_restart);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_fill_list(int *fcols, const int *extras, const double *bd2, int *slot_of, int *meta,
                                FitCtrl *ctrl, int parked, hipStream_t st, int spec_max, int spec) {
  hipLaunchKernelGGL(k_cov_fill_list, dim3(1), dim3(256), 0, st, fcols, extras, bd2, slot_of, meta, ctrl, parked,
                     spec_max, spec);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_resume(FitCtrl *ctrl, hipStream_t st) {
  hipLaunchKernelGGL(k_cov_resume, dim3(1), dim3(1), 0, st, ctrl);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_fill_union(const CovUnion &u, int restart, const int *extras, const double *bd2, int spec_max,
                                 int spec_min, int *slot_of, int *meta, int p, int *fcols, FitCtrl *fill_ctrl,
                                 hipStream_t st, int C) {
  if (u.nf < 1 || u.nf > 8 || spec_max > 64 || spec_min < 0 || spec_min > spec_max) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_cov_fill_union, dim3(1), dim3(256), 0, st, u, restart, extras, bd2, spec_max, spec_min, slot_of, meta, p,
                     fcols, fill_ctrl, C);
  LAUNCH_CHECK();
  return hipSuccess;
}

int cov_streamed_tiles_per_wave() { return COV_NJ; }

hipError_t launch_cov_panel(const double *X, const double *aux, long ld, int p, const double *mask, const int *fcols,
                            int g0, int ngroups, int rows_per_slab, int nslab, double *part, const FitCtrl *ctrl,
                            int parked, hipStream_t st, int variant) {
  const int pt = (p + 15) / 16, njg = (pt + COV_NJ - 1) / COV_NJ;
  if (variant == 4 && ngroups <= 2) {
    // one block per (slab, 64-column group) for BOTH groups of the launch: X streamed once
    const size_t lds2 = (size_t)CP2_COLS * CP_LD * sizeof(double);
    const long nb2 = (long)nslab * njg;
    if (mask)
      hipLaunchKernelGGL(k_cov_panel_pair<true>, dim3((unsigned)nb2), dim3(256), lds2, st, X, aux, ld, p, mask, fcols,
                         g0, rows_per_slab, nslab, njg, part, ctrl, parked);
    else
      hipLaunchKernelGGL(k_cov_panel_pair<false>, dim3((unsigned)nb2), dim3(256), lds2, st, X, aux, ld, p, mask, fcols,
                         g0, rows_per_slab, nslab, njg, part, ctrl, parked);
    LAUNCH_CHECK();
    return hipSuccess;
  }
  // one block per (group, slab, 64-column group)
  const size_t lds = (size_t)CP_COLS * CP_LD * sizeof(double);
  const long nblk = (long)ngroups * nslab * njg;
  if (mask)
    hipLaunchKernelGGL(k_cov_panel_lds2<true>, dim3((unsigned)nblk), dim3(256), lds, st, X, aux, ld, p, mask, fcols, g0,
                       ngroups, rows_per_slab, nslab, njg, part, ctrl, parked);
  else
    hipLaunchKernelGGL(k_cov_panel_lds2<false>, dim3((unsigned)nblk), dim3(256), lds, st, X, aux, ld, p, mask, fcols,
                       g0, ngroups, rows_per_slab, nslab, njg, part, ctrl, parked);
  LAUNCH_CHECK();
  return hipSuccess;
}

// one-time opt-in to more than 64 KB of dynamic LDS for the staged panel kernel
hipError_t cov_panel_prepare() {
  hipError_t e = hipSuccess;
  const int lds2 = (int)((size_t)CP2_COLS * CP_LD * sizeof(double));
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_panel_pair<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_panel_pair<false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
}

hipError_t launch_cov_reduce(const double *part, int p, const int *fcols, const int *slot_of, double *G, int g0,
                             int ngroups, int nslab, const FitCtrl *ctrl, int parked, hipStream_t st, int ex_lo,
                             int ex_hi) {
  const int pt = (p + 15) / 16, njg = (pt + COV_NJ - 1) / COV_NJ;
  hipLaunchKernelGGL(k_cov_reduce, dim3(njg * COV_NJ * 2, ngroups), dim3(256), 0, st, part, g0, ngroups, nslab, njg, p,
                     fcols, slot_of, G, ctrl, 0, parked, ex_lo, ex_hi);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_reduce_compact_sets(const double *part, int p, const int *fcols, const int *slot_of, int *meta,
                                          const CovRowSets &rs, int g0, int ngroups, int nslab, int CS,
                                          const FitCtrl *ctrl, int parked, hipStream_t st) {
  if (rs.nr < 1 || rs.nr > 9) return hipErrorInvalidValue;
  const int pt = (p + 15) / 16, njg = (pt + COV_NJ - 1) / COV_NJ;
  hipLaunchKernelGGL(k_cov_reduce_sets, dim3(njg * COV_NJ * 2, ngroups, rs.nr), dim3(256), 0, st, part, g0, ngroups,
                     nslab, njg, p, fcols, slot_of, rs, ctrl, parked);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cov_compact_sets, dim3((p + 255) / 256, ngroups, rs.nr), dim3(256), 0, st, rs, p, slot_of, fcols,
                     g0, CS, ctrl, parked, meta);
  LAUNCH_CHECK();
  return hipSuccess;
}

// Xp[j][r] = X[j][perm[r]] (0 where perm[r] < 0): the fold-major copy of X for the shared fills of the CV row sets
__global__ void __launch_bounds__(256) k_rows_permute(const double *__restrict__ X, long ld, const int *__restrict__ perm,
                                                      long ldp, double *__restrict__ Xp) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= ldp) return;
  const int o = perm[r];
  Xp[(size_t)blockIdx.y * ldp + r] = o >= 0 ? X[(size_t)blockIdx.y * ld + o] : 0.0;
}
hipError_t launch_rows_permute(const double *X, long ld, int p, const int *perm, long ldp, double *Xp, hipStream_t st) {
  for (int j0 = 0; j0 < p; j0 += 32768) {  // grid.y limit
    const int nj = std::min(32768, p - j0);
    hipLaunchKernelGGL(k_rows_permute, dim3((unsigned)((ldp + 255) / 256), nj), dim3(256), 0, st, X + (size_t)j0 * ld, ld,
                       perm, ldp, Xp + (size_t)j0 * ldp);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}

hipError_t launch_cov_compact(const double *G, int p, const int *slot_of, const int *fcols, int g0, int ngroups,
                              double *GS, int CS, const FitCtrl *ctrl, int parked, hipStream_t st, const double *xtx,
                              int *meta) {
  hipLaunchKernelGGL(k_cov_compact, dim3((p + 255) / 256, ngroups), dim3(256), 0, st, G, p, slot_of, fcols, g0, GS, CS,
                     ctrl, parked, xtx, meta);
  LAUNCH_CHECK();
  return hipSuccess;
}


hipError_t launch_cov_d(const double *G, int p, const int *slot_of, const double *xty, const int *A_cur,
                        const double *b_cur, double *d_out, const double *beta_dense, const double *xtx, double n_t,
                        double lambda, const unsigned char *always, double *bd, const unsigned char *inA, double *bmm,
                        const FitCtrl *ctrl, int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_cov_d, dim3((p + 31) / 32), dim3(256), 0, st, G, p, slot_of, xty, A_cur, b_cur, d_out,
                     beta_dense, xtx, n_t, lambda, always, bd, inA, bmm, ctrl, slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_gram(const double *G, int p, const int *slot_of, const int *A_new, int T0, int mt, double *Gt,
                           int *meta, const FitCtrl *ctrl, int slot, hipStream_t st) {
  hipLaunchKernelGGL(k_cov_gram, dim3(mt * (mt + 1) / 2), dim3(256), 0, st, G, p, slot_of, A_new, T0, Gt, meta, ctrl,
                     slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_publish(const unsigned char *dev, unsigned char *host, int ctrl_bytes, size_t off_sse, int n_sse,
                          size_t off_b, size_t off_a, int kcopy, unsigned long long *seq_host, unsigned long long seq,
                          hipStream_t st, const int *count_ptr) {
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st, dev, host, ctrl_bytes, off_sse, n_sse, off_b, off_a, kcopy,
                     seq_host, seq, count_ptr);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_dot(const double *a, const double *b, long n, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_dot, dim3(1), dim3(256), 0, st, a, b, n, out);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_copy(const double *src, double *dst, long n, hipStream_t st) {
  hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const d2 *>(src),
                     reinterpret_cast<d2 *>(dst), n / 2);
  LAUNCH_CHECK();
  return hipSuccess;
}

}  // namespace bessx
