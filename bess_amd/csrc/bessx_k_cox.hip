// bessx_k_cox.hip -- Cox proportional hazards: state, one-pass score, Newton step, line search (+ their launchers)
#include "bessx_kdev.hpp"

namespace bessx {

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
// RW: rows per sub-tile (32, or 16: half the LDS per wave -- four waves per SIMD instead of two -- and 128-byte instead of
// 256-byte runs per column and load instruction).
template <int U, int MAP, int RW, int COLS>
__global__ void __launch_bounds__(256) k_cox_score1p(const double *__restrict__ X, long ld, int p, int nrb,
                                                     const double *__restrict__ TH, const double *__restrict__ CU,
                                                     const double *__restrict__ CV, const double *__restrict__ C2,
                                                     double *__restrict__ out, const FitCtrl *__restrict__ ctrl,
                                                     int slot) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;
  constexpr int RS = RW + 1, NSEG = RW / 2, CPI = 64 / NSEG, NIT = COLS / CPI;
  __shared__ double tile[4][COLS * RS];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wid = (long)blockIdx.x * 4 + wv;
  const int ncg = (p + COLS - 1) / COLS;
  const long cg = MAP == 0 ? wid / nrb : wid % ncg;
  const int rb = MAP == 0 ? (int)(wid - cg * nrb) : (int)(wid / ncg);
  if (cg >= ncg || rb >= nrb) return;
  const int j0 = (int)cg * COLS;
  double loc = 0.0, g1 = 0.0, g2 = 0.0, p1 = 0.0, p2 = 0.0, p0 = 0.0;
  constexpr int NSUB = 128 * U / RW;
  const long rbase = (long)rb * (128 * U);
  const int c4 = lane / NSEG, seg = lane % NSEG;
  d2 nxt[NIT];
  auto load_sub = [&](int sub) {
    const long r0 = rbase + (long)sub * RW + 2 * seg;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int j = min(j0 + CPI * it + c4, p - 1);
      nxt[it] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(X + (size_t)j * ld + r0));
    }
  };
  load_sub(NSUB - 1);
  for (int sub = NSUB - 1; sub >= 0; sub--) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int o = (CPI * it + c4) * RS + 2 * seg;
      tile[wv][o] = nxt[it].x;
      tile[wv][o + 1] = nxt[it].y;
    }
    if (sub > 0) load_sub(sub - 1);  // next tile's loads fly while this one is scanned
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const long r0 = rbase + (long)sub * RW;
    for (int r = RW - 1; r >= 0; r--) {
      // wave-uniform (scalar loads; staging them through LDS measured the same)
      const double th = TH[r0 + r], u = CU[r0 + r], v = CV[r0 + r], c2 = C2[r0 + r];
      const double x = tile[wv][(COLS == 64 ? lane : (lane & (COLS - 1))) * RS + r];
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
  if (lane < COLS && j0 + lane < p) {
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

// The same pass for up to COX_MC_MAX chains at once (round 6, chunk chains that share their passes over X): the tile of X
// is staged in LDS once and walked once per ACTIVE chain -- per chain exactly the walk above (rows bottom-up, the same
// fma order, the same five sums per (row block, column) and P0 per row block), so every chain's sums are bitwise those
// of a launch of its own.  The extra walks read the tile from LDS (16 KB per chain and tile against 128 B/clock) and cost
// ~7 vector instructions per element: four chains keep the vector ALU at ~70 % of what the HBM stream allows.
// (Every chain's vectors are kernel arguments of their own, `const double *__restrict__`, as in k_cox_score1p.)
#define COX_MC_CHAIN_ARGS(i)                                                                                         \
  const double *__restrict__ TH##i, const double *__restrict__ CU##i, const double *__restrict__ CV##i,              \
      const double *__restrict__ C2##i, double *__restrict__ out##i, const FitCtrl *__restrict__ ctrl##i, int slot##i
template <int U, int NC>
__global__ void __launch_bounds__(256) k_cox_score1p_mc(const double *__restrict__ X, long ld, int p, int nrb,
                                                        COX_MC_CHAIN_ARGS(0), COX_MC_CHAIN_ARGS(1), COX_MC_CHAIN_ARGS(2),
                                                        COX_MC_CHAIN_ARGS(3), COX_MC_CHAIN_ARGS(4), COX_MC_CHAIN_ARGS(5),
                                                        int *__restrict__ ran) {
  constexpr int RW = 16, COLS = 64, RS = RW + 1, NSEG = RW / 2, CPI = 64 / NSEG, NIT = COLS / CPI;
  __shared__ double tile[4][COLS * RS];
  __shared__ __align__(16) double vecs[4][NC][RW][4];  // per wave and chain: (theta, u, v, c2) of the sub-tile's rows
  struct Chains {  // (indexed by unrolled constants only)
    const double *TH[6], *CU[6], *CV[6], *C2[6];
    double *out[6];
    const FitCtrl *ctrl[6];
    int slot[6];
  };
  const Chains a = {{TH0, TH1, TH2, TH3, TH4, TH5}, {CU0, CU1, CU2, CU3, CU4, CU5}, {CV0, CV1, CV2, CV3, CV4, CV5},
                    {C20, C21, C22, C23, C24, C25}, {out0, out1, out2, out3, out4, out5},
                    {ctrl0, ctrl1, ctrl2, ctrl3, ctrl4, ctrl5}, {slot0, slot1, slot2, slot3, slot4, slot5}};
  bool on[NC];
  int na = 0;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const FitCtrl *ct = a.ctrl[c];
    on[c] = ct == nullptr || (!ct->done && ct->l == a.slot[c] - 1);
    na += on[c] ? 1 : 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && ran != nullptr) *ran = na;
  if (na == 0) return;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wid = (long)blockIdx.x * 4 + wv;
  const int ncg = (p + COLS - 1) / COLS;
  const long cg = wid % ncg;
  const int rb = (int)(wid / ncg);
  if (cg >= ncg || rb >= nrb) return;
  const int j0 = (int)cg * COLS;
  double loc[NC], g1[NC], g2[NC], p1[NC], p2[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) loc[c] = g1[c] = g2[c] = p1[c] = p2[c] = 0.0;
  constexpr int NSUB = 128 * U / RW;
  const long rbase = (long)rb * (128 * U);
  const int c4 = lane / NSEG, seg = lane % NSEG;
  // the sub-tile's rows of the chains' four n-vectors travel WITH the tile: lane l < 32 brings rows 2 (l % 8), + 1 of
  // vector l / 8 of every chain (one 16-byte load per chain), stored row-major (theta, u, v, c2) per row, so that the walk
  // reads a row's four values with two broadcast ds_read_b128 -- per-row scalar loads (k_cox_score1p) keep one chain's
  // walk under the tile's load time but not three (5.3 / 5.8 / 9.4 / 11.8 ms per pass with 1 / 2 / 3 / 4 chains)
  const int vq = (lane >> 3) & 3, vr = 2 * (lane & 7);
  d2 nxt[NIT], nv[NC];
  auto load_sub = [&](int sub) {
    const long r0 = rbase + (long)sub * RW;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const double *t0 = a.TH[c], *t1 = a.CU[c], *t2 = a.CV[c], *t3 = a.C2[c];
      const double *src = vq == 0 ? t0 : (vq == 1 ? t1 : (vq == 2 ? t2 : t3));
      nv[c] = *reinterpret_cast<const d2 *>(src + r0 + vr);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int j = min(j0 + CPI * it + c4, p - 1);
      nxt[it] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(X + (size_t)j * ld + r0 + 2 * seg));
    }
  };
  load_sub(NSUB - 1);
  for (int sub = NSUB - 1; sub >= 0; sub--) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int o = (CPI * it + c4) * RS + 2 * seg;
      tile[wv][o] = nxt[it].x;
      tile[wv][o + 1] = nxt[it].y;
    }
    if (lane < 32) {
#pragma unroll
      for (int c = 0; c < NC; c++) {
        vecs[wv][c][vr][vq] = nv[c].x;
        vecs[wv][c][vr + 1][vq] = nv[c].y;
      }
    }
    if (sub > 0) load_sub(sub - 1);  // next tile's loads fly while this one is walked
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // rows outside, chains inside: the NC walks are independent dependency chains (a row's update is three dependent
    // fp64 operations), so they hide each other's latency, and a row's value is read from the tile once.  Every chain of
    // the launch is walked -- a chain whose gate is closed costs its arithmetic on whatever its vectors hold and is not
    // written below.
    for (int r = RW - 1; r >= 0; r--) {
      const double x = tile[wv][lane * RS + r];
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const d2 tu = *reinterpret_cast<const d2 *>(&vecs[wv][c][r][0]), vc = *reinterpret_cast<const d2 *>(&vecs[wv][c][r][2]);
        const double th = tu.x, u = tu.y, v = vc.x, c2 = vc.y;
        loc[c] = fma(th, x, loc[c]);
        g1[c] = fma(x, v, g1[c]);
        g2[c] = fma(u * x, x, g2[c]);
        const double m = c2 * loc[c];
        p1[c] += m;
        p2[c] = fma(m, loc[c], p2[c]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  const size_t plane = (size_t)nrb * p;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    if (!on[c]) continue;
    double *out = a.out[c];
    if (j0 + lane < p) {
      const size_t o = (size_t)rb * p + j0 + lane;
      out[o] = loc[c];
      out[plane + o] = p1[c];
      out[2 * plane + o] = p2[c];
      out[3 * plane + o] = g1[c];
      out[4 * plane + o] = g2[c];
    }
    if (cg == 0) {  // P0 of this row block, by the wave of its first column group (fixed order)
      double p0 = 0.0;
      for (int r = lane; r < 128 * U; r += 64) p0 += a.C2[c][rbase + r];
      p0 = wave_sum(p0);
      if (lane == 0) out[5 * plane + rb] = p0;
    }
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

static int g_cox_score_variant = 1;  // the wave -> (column group, row block) map (1 = round 4, 0 = round 3)
void cox_score_set_variant(int v) { g_cox_score_variant = v & 1; }

// the one-pass Cox score of up to COX_MC_MAX chains in ONE pass over X (k_cox_score1p_mc)
template <int U>
static hipError_t launch_cox_score1p_mc_u(const double *X, long ld, int p, int nrb, const CoxMc &a, hipStream_t st) {
  const long nw = (long)nrb * ((p + 63) / 64);
  const int nb = (int)((nw + 3) / 4);
#define COX_MC_PASS(i) a.TH[i], a.CU[i], a.CV[i], a.C2[i], a.out[i], a.ctrl[i], a.slot[i]
#define COX_MC_GO(NCC)                                                                                                \
  hipLaunchKernelGGL((k_cox_score1p_mc<U, NCC>), dim3(nb), dim3(256), 0, st, X, ld, p, nrb, COX_MC_PASS(0), COX_MC_PASS(1), \
                     COX_MC_PASS(2), COX_MC_PASS(3), COX_MC_PASS(4), COX_MC_PASS(5), a.ran)
  switch (a.nc) {
    case 1: COX_MC_GO(1); break;
    case 2: COX_MC_GO(2); break;
    case 3: COX_MC_GO(3); break;
    case 4: COX_MC_GO(4); break;
    case 5: COX_MC_GO(5); break;
    default: COX_MC_GO(6); break;
  }
#undef COX_MC_GO
#undef COX_MC_PASS
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cox_score1p_mc(const double *X, long ld, int p, int U, int nrb, const CoxMc &a, hipStream_t st) {
  if (a.nc < 1 || a.nc > COX_MC_MAX) return hipErrorInvalidValue;
  switch (U) {
    case 8: return launch_cox_score1p_mc_u<8>(X, ld, p, nrb, a, st);
    case 4: return launch_cox_score1p_mc_u<4>(X, ld, p, nrb, a, st);
    case 2: return launch_cox_score1p_mc_u<2>(X, ld, p, nrb, a, st);
    default: return launch_cox_score1p_mc_u<1>(X, ld, p, nrb, a, st);
  }
}

hipError_t launch_cox_score_pass(const double *X, long ld, int p, int U, int nrb, CoxBufs cb, double *part,
                                 double *part2, const FitCtrl *ctrl, int slot, hipStream_t st) {
  if (cb.one_pass) {
    long nw = (long)nrb * ((p + 63) / 64);
    int nb = (int)((nw + 3) / 4);
    // (RW = 16 -- half the LDS per wave, four waves per SIMD -- and COLS = 32 with RW = 64 -- 512-byte runs per column
    // and load instruction, half the lanes idle in the walk -- were measured in round 4 at 0.77-0.79 of 8 TB/s like this
    // geometry: neither occupancy nor run length is what holds the pass at 6.2-6.4 TB/s; only <.., 32, 64> is instantiated)
#define CS1_GO3(UU, MM)                                                                                                \
  hipLaunchKernelGGL((k_cox_score1p<UU, MM, 32, 64>), dim3(nb), dim3(256), 0, st, X, ld, p, nrb, (const double *)cb.TH, \
                     (const double *)cb.CU, (const double *)cb.CV, (const double *)cb.C2, part, ctrl, slot)
#define CS1_GO(UU)         \
  if (g_cox_score_variant) \
    CS1_GO3(UU, 1);        \
  else                     \
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


__global__ void __launch_bounds__(256) k_vec_sub(double *__restrict__ a, const double *__restrict__ b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] -= b[i];
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


}  // namespace bessx
