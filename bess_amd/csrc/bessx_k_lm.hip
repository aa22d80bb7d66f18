// bessx_k_lm.hip -- data preparation, the streaming score pass, the Gram kernels of the restricted fits, Algorithm::fit bookkeeping,
// the LM residual, publication of the result block (+ their launchers)
#include <atomic>

#include "bessx_kdev.hpp"

namespace bessx {

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
// K1 / K2 for SEVERAL chains in one pass over X (round 6).  Chunk chains of the streaming forms (LM score_mode 1,
// logistic, Poisson) each need X^T v for a vector of their own; issued per chain, every one of them streams the whole of
// X.  Here a launch carries up to XTV_MC_MAX (v, v2, part, part2, ctrl, slot) sets: a column slice of X is loaded ONCE
// into registers and every active chain's accumulators are formed from it.
//
// Bitwise the same sums as k_xtv: the same row blocks of 128*U rows, the same rows per lane (row0 + u*128 + {0,1}), the
// same fma chain per lane and column (u ascending, .x then .y from 0), the same butterfly over the 64 lanes (offsets
// 32, 16, 8, 4, 2, 1 -- k_xtv's exchange steps and its plain steps add the same pairs, the number of columns a wave keeps
// only changes which lane ends up with which column), the same part[rb][j] layout per chain.
//
// Block = 16 waves on ONE row block, XTV_MC_CPW consecutive columns each; the active chains' v (and v2) slices of that
// row block are staged in LDS once per block (8 KB per chain and vector at U = 8, against the 4 MB of X the block streams)
// and a wave re-reads a chain's slice once per PAIR of columns (8 ds_read_b128): registers hold two columns of X, the
// slices of the chain(s) being summed and their accumulators -- whatever the number of chains.  1024 threads per block
// cap the kernel at 128 registers: four waves per SIMD hide the load latency (no software prefetch -- the first version
// kept the next pair in flight in 64 more registers, two waves per SIMD, and the compiler's vmcnt(0) in front of the chain
// loop made every pair wait for its own loads: 0.25 ms per chain and pass on configs[1], nothing shared).  Without the
// second accumulator two chains are summed at a time: their dependent shuffle chains (6 ds_bpermute steps each) overlap.
// Gate per chain like k_xtv's; a launch whose chains have all converged falls through after the gate loads.
// ------------------------------------------------------------------------------------------
// cross-lane moves of a double on the vector ALU (no LDS): a DPP row operation on both halves, and gfx950's lane swaps
template <int CTRL>
__device__ __forceinline__ double xl_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second: the sum of the two
// results is a[l] + a[l + 32] in the lower half and b[l - 32] + b[l] in the upper half
__device__ __forceinline__ double xl_add_swap32(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
// v_permlane16_swap exchanges the odd rows (of 16 lanes) of its first operand with the even rows of its second: even rows
// end with a[l] + a[l + 16], odd rows with b[l - 16] + b[l]
__device__ __forceinline__ double xl_add_swap16(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}

constexpr int XTV_MC_CPW = 32;    // columns per wave
constexpr int XTV_MC_WAVES = 16;  // waves per block (512 columns)

template <int U, bool TWO>
__global__ void __launch_bounds__(64 * XTV_MC_WAVES) k_xtv_mc(const double *__restrict__ X, long ld, int p, int nrb, XtvMc a) {
  // (16-byte alignment of the slices: behind a 40-byte static array the dynamic region started 8 bytes off and every
  // ds_read_b128 was a misaligned access -- 15 B/clock of LDS bandwidth per compute unit instead of 128)
  extern __shared__ __align__(16) double sv[];  // [active chain][TWO ? 2 : 1][128 * U]
  __shared__ __align__(16) int act[XTV_MC_MAX + 4];
  __shared__ __align__(16) const double *vsrc[2 * XTV_MC_MAX];  // where every staged slice comes from
  constexpr int RB = 128 * U, VPC = TWO ? 2 : 1, NT = 64 * XTV_MC_WAVES;
  if (threadIdx.x == 0) {
    int na = 0;
    for (int c = 0; c < a.nc; c++) {
      const FitCtrl *ct = a.ctrl[c];
      if (ct == nullptr || (!ct->done && ct->l == a.slot[c] - 1)) {
        vsrc[na * VPC] = a.v[c];
        if (TWO) vsrc[na * VPC + 1] = a.v2[c];
        act[na++] = c;
      }
    }
    act[XTV_MC_MAX] = na;
    if (blockIdx.x == 0 && a.ran != nullptr) *a.ran = na;
  }
  __syncthreads();
  const int na = act[XTV_MC_MAX];
  if (na == 0) return;
  const int nsp = (p + XTV_MC_WAVES * XTV_MC_CPW - 1) / (XTV_MC_WAVES * XTV_MC_CPW);
  const int rb = blockIdx.x / nsp, sp = blockIdx.x - rb * nsp;
  const long rbase = (long)rb * RB;
  {
    // stage the slices: all of a thread's loads first (independent, in flight together), then its LDS stores
    constexpr int PER = RB / 2, SPT = NT / PER;            // d2 per slice; slices per sweep of the block
    constexpr int SWEEPS = (2 * XTV_MC_MAX + SPT - 1) / SPT;
    const int sl0 = threadIdx.x / PER, o = threadIdx.x - sl0 * PER, nsl = na * VPC;
    d2 tmp[SWEEPS];
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
      const int sl = sl0 + k * SPT;
      if (sl < nsl) tmp[k] = *reinterpret_cast<const d2 *>(vsrc[sl] + rbase + 2 * o);
    }
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
      const int sl = sl0 + k * SPT;
      if (sl < nsl) reinterpret_cast<d2 *>(sv)[(size_t)sl * PER + o] = tmp[k];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = (sp * XTV_MC_WAVES + wv) * XTV_MC_CPW;
  if (j0 >= p) return;
  const long row0 = rbase + lane * 2;
  const bool up = (lane & 32) != 0;
  const bool writer = (lane & 31) == 0;
  // sums of one chain's slice against the two columns in registers, folded over the lanes: column 0 ends in the lower
  // half of the wave, column 1 in the upper half.  The same pairs as k_xtv's butterfly (offsets 32, 16, 8, 4, 2, 1), but
  // on the vector ALU alone: gfx950's v_permlane32_swap / v_permlane16_swap for the two steps across rows of 16 lanes,
  // DPP row operations for the four inside a row -- the LDS pipeline is left to the slices' reads (12 ds_bpermute_b32
  // per chain and pair of columns otherwise: measured 2-4 % slower at 8 chains, tools/xtv_multi_bench.py).
  auto fold = [&](double c0, double c1) {
    double s = xl_add_swap32(c0, c1);          // lanes < 32: c0[l] + c0[l + 32]; lanes >= 32: c1[l - 32] + c1[l]
    s = xl_add_swap16(s, s);                   // + the value 16 lanes away
    s += xl_dpp<0x128>(s);                     // xor 8: row_ror:8
    s += xl_dpp<0x1b>(xl_dpp<0x141>(s));       // xor 4 = (xor 7 = row_half_mirror) then (xor 3 = quad_perm [3,2,1,0])
    s += xl_dpp<0x4e>(s);                      // xor 2: quad_perm [2,3,0,1]
    s += xl_dpp<0xb1>(s);                      // xor 1: quad_perm [1,0,3,2]
    return s;
  };
  for (int g = 0; g < XTV_MC_CPW && j0 + g < p; g += 2) {
    const int ja = j0 + g, jb = ja + 1 < p ? ja + 1 : p - 1;  // tail: the last column again, discarded below
    const double *pa = X + (size_t)ja * ld + row0, *pb = X + (size_t)jb * ld + row0;
    d2 xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; u++) xa[u] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(pa + u * 128));
#pragma unroll
    for (int u = 0; u < U; u++) xb[u] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(pb + u * 128));
    const int j = ja + (up ? 1 : 0);
    const bool wr = writer && j < p;
    if (TWO) {
      for (int w = 0; w < na; w++) {
        const double *vs = sv + (size_t)w * 2 * RB + lane * 2;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
          const d2 vr = *reinterpret_cast<const d2 *>(vs + u * 128), vr2 = *reinterpret_cast<const d2 *>(vs + RB + u * 128);
          a0 = fma(xa[u].x, vr.x, a0);
          a0 = fma(xa[u].y, vr.y, a0);
          a1 = fma(xb[u].x, vr.x, a1);
          a1 = fma(xb[u].y, vr.y, a1);
          b0 = fma(xa[u].x * xa[u].x, vr2.x, b0);
          b0 = fma(xa[u].y * xa[u].y, vr2.y, b0);
          b1 = fma(xb[u].x * xb[u].x, vr2.x, b1);
          b1 = fma(xb[u].y * xb[u].y, vr2.y, b1);
        }
        const double s = fold(a0, a1), s2 = fold(b0, b1);
        if (wr) {
          const int c = act[w];
          a.part[c][(size_t)rb * p + j] = s;
          a.part2[c][(size_t)rb * p + j] = s2;
        }
      }
    } else {
      for (int w = 0; w < na; w += 2) {
        const bool second = w + 1 < na;
        const double *vs = sv + (size_t)w * RB + lane * 2;
        const double *vt = second ? vs + RB : vs;  // (odd count: the last chain twice, the copy discarded)
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
          const d2 vr = *reinterpret_cast<const d2 *>(vs + u * 128), wr2 = *reinterpret_cast<const d2 *>(vt + u * 128);
          a0 = fma(xa[u].x, vr.x, a0);
          a0 = fma(xa[u].y, vr.y, a0);
          a1 = fma(xb[u].x, vr.x, a1);
          a1 = fma(xb[u].y, vr.y, a1);
          b0 = fma(xa[u].x, wr2.x, b0);
          b0 = fma(xa[u].y, wr2.y, b0);
          b1 = fma(xb[u].x, wr2.x, b1);
          b1 = fma(xb[u].y, wr2.y, b1);
        }
        const double s = fold(a0, a1), t = fold(b0, b1);
        if (wr) {
          a.part[act[w]][(size_t)rb * p + j] = s;
          if (second) a.part[act[w + 1]][(size_t)rb * p + j] = t;
        }
      }
    }
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
// Algorithm::fit bookkeeping (src/Algorithm.h:141-170) on the device.
// ------------------------------------------------------------------------------------------
// Start of a fit: beta <- beta_init (sparse), A_list.col(0) = 0, l = 0.
__global__ void __launch_bounds__(256) k_fit_begin(FitCtrl *__restrict__ ctrl, int T0, int k_init,
                                                   const int *__restrict__ init_idx,
                                                   const double *__restrict__ init_val, double coef0_init,
                                                   int *__restrict__ A_cur, double *__restrict__ b_cur,
                                                   double *__restrict__ beta_dense, int *__restrict__ hist,
                                                   unsigned char *__restrict__ inA, int serial) {
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
    // (the fit chained behind this one recognises its parent by this number, like behind a fit opened by
    // k_fit_continue: without it the first fit of a chain was never followed on the device)
    if (serial > 0) ctrl->serial = serial;
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

template <int U, bool TWO>
static hipError_t launch_xtv_mc_t(const double *X, long ld, int p, const XtvMc &a, hipStream_t st) {
  const int nrb = (int)(ld / (128 * U));
  const int nsp = (p + XTV_MC_WAVES * XTV_MC_CPW - 1) / (XTV_MC_WAVES * XTV_MC_CPW);
  const size_t lds = (size_t)a.nc * (TWO ? 2 : 1) * 128 * U * sizeof(double);
  // dynamic LDS beyond 64 KB has to be requested once per kernel instance AND device (a process may hold sessions on
  // several devices; the flags are only ever set, a repeated request is harmless)
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xtv_mc<U, TWO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((size_t)XTV_MC_MAX * (TWO ? 2 : 1) * 128 * U * sizeof(double)));
    if (e != hipSuccess) return e;
    attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_xtv_mc<U, TWO>), dim3(nrb * nsp), dim3(64 * XTV_MC_WAVES), lds, st, X, ld, p, nrb, a);
  LAUNCH_CHECK();
  return hipSuccess;
}

// X^T v for up to XTV_MC_MAX chains in one pass over X (k_xtv_mc); two = every chain brings a second vector
hipError_t launch_xtv_mc(const double *X, long ld, int p, int U, const XtvMc &a, bool two, hipStream_t st) {
  if (a.nc < 1 || a.nc > XTV_MC_MAX) return hipErrorInvalidValue;
  switch (U) {
    case 8: return two ? launch_xtv_mc_t<8, true>(X, ld, p, a, st) : launch_xtv_mc_t<8, false>(X, ld, p, a, st);
    case 4: return two ? launch_xtv_mc_t<4, true>(X, ld, p, a, st) : launch_xtv_mc_t<4, false>(X, ld, p, a, st);
    case 2: return two ? launch_xtv_mc_t<2, true>(X, ld, p, a, st) : launch_xtv_mc_t<2, false>(X, ld, p, a, st);
    default: return two ? launch_xtv_mc_t<1, true>(X, ld, p, a, st) : launch_xtv_mc_t<1, false>(X, ld, p, a, st);
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


hipError_t launch_fit_begin(FitCtrl *ctrl, int T0, int k_init, const int *init_idx, const double *init_val,
                            double coef0_init, int *A_cur, double *b_cur, double *beta_dense, int p, int *hist,
                            hipStream_t st, unsigned char *inA, int serial) {
  hipError_t e = hipMemsetAsync(beta_dense, 0, (size_t)p * sizeof(double), st);
  if (e == hipSuccess && inA != nullptr) e = hipMemsetAsync(inA, 0, (size_t)p, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fit_begin, dim3(1), dim3(256), 0, st, ctrl, T0, k_init, init_idx, init_val, coef0_init, A_cur,
                     b_cur, beta_dense, hist, inA, serial);
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
