// bessx_k_glm.hip -- group selection, the GLM families (logistic, Poisson), screening (+ their launchers)
#include "bessx_kdev.hpp"

namespace bessx {

// ------------------------------------------------------------------------------------------
// Group selection (group size > 1; GroupPdas* with real groups).
// Per group g (columns c0 .. c0+s-1): bd_g = || Phi_g beta_g + Phi_g^{-1} d_g ||^2 / s with Phi_g = sqrtm(M_g),
//   LM:  M_g = 2 lambda I + X_g^T X_g / n_t (src/utilities.cpp:142-151), d = X^T r / n_t - 2 lambda beta
//   GLM: M_g = X_g^T diag(h) X_g + 2 lambda I, d = X^T g - 2 lambda beta   (src/Algorithm.h:1238-1257, 1342-1361)
// k_group_moments forms the s x s blocks (and optionally X_g^T w2) in one pass over the group's columns;
// k_group_score takes the symmetric square root by a Jacobi eigen-decomposition, one thread per group.
// ------------------------------------------------------------------------------------------

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
// eig_mode: 0 = diagonalise and use; 1 = diagonalise, STORE eigenvectors and eigenvalues (ev / el) and use; 2 = LOAD them
// instead of diagonalising.  For LM the block depends on the row set and lambda only, not on the coefficients: the
// Jacobi sweeps -- 270 us per launch for 2000 groups of 5, one group per thread -- run once per (row set, lambda) and
// every PDAS iteration after it loads 30 numbers per group.  Same arithmetic, stored: bit-identical scores.
template <int SC>
__device__ __forceinline__ double group_sacrifice(int s_rt, double *__restrict__ a, double *__restrict__ v,
                                                  const double *__restrict__ bv, const double *__restrict__ dv,
                                                  int eig_mode = 0, double *__restrict__ ev = nullptr,
                                                  double *__restrict__ el = nullptr) {
  const int s = SC > 0 ? SC : s_rt;
  constexpr int UF = SC > 0 ? SC : 1;  // (run-time widths: no unrolling)
  if (eig_mode == 2) {
#pragma unroll UF
    for (int k = 0; k < s; k++) {
      a[k * s + k] = el[k];
#pragma unroll UF
      for (int j = 0; j < s; j++) v[k * s + j] = ev[k * s + j];
    }
  }
  for (int sweep = 0; sweep < (eig_mode == 2 ? 0 : 60); sweep++) {
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
  if (eig_mode == 1) {
#pragma unroll UF
    for (int k = 0; k < s; k++) {
      el[k] = a[k * s + k];
#pragma unroll UF
      for (int j = 0; j < s; j++) ev[k * s + j] = v[k * s + j];
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
                                                  double lambda, const double *__restrict__ beta_dense, int eig_mode,
                                                  double *__restrict__ eig_v, double *__restrict__ eig_l) {
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
  return group_sacrifice<SC>(s, a, v, bv, dv, eig_mode, eig_v ? eig_v + goff[g] : nullptr, eig_l ? eig_l + c0 : nullptr);
}

// lm != 0: dcol is taken from the score-pass partials (sum over row blocks / n_t); else dcol holds X^T g already.
__global__ void __launch_bounds__(64) k_group_score(int N, const int *__restrict__ gidx, const int *__restrict__ gsz,
                                                    const int *__restrict__ goff, const double *__restrict__ mblk,
                                                    const double *__restrict__ dcol, const double *__restrict__ part,
                                                    int nrb, int p, int lm, double n_t, double lambda,
                                                    const double *__restrict__ beta_dense,
                                                    const unsigned char *__restrict__ always,
                                                    double *__restrict__ bd, const FitCtrl *__restrict__ ctrl,
                                                    int slot, int eig_mode, double *__restrict__ eig_v,
                                                    double *__restrict__ eig_l) {
  if (ctrl != nullptr && (ctrl->done || ctrl->l != slot - 1)) return;  // (a speculative slot of a fit that has ended)
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= N) return;
  const int s = gsz[g], c0 = gidx[g];
  if (s > GRP_MAX) return;  // wider groups: k_group_score_big
  double res;
#define GS_CASE(S) \
  case S: res = group_score_one<S>(g, s, c0, goff, mblk, dcol, part, nrb, p, lm, n_t, lambda, beta_dense, eig_mode, eig_v, \
                                   eig_l); break
  switch (s) {
    GS_CASE(1);
    GS_CASE(2);
    GS_CASE(3);
    GS_CASE(4);
    GS_CASE(5);
    GS_CASE(6);
    GS_CASE(7);
    GS_CASE(8);
    default: res = group_score_one<0>(g, s, c0, goff, mblk, dcol, part, nrb, p, lm, n_t, lambda, beta_dense, eig_mode, eig_v,
                                      eig_l);
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
hipError_t launch_iota(int *a, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
  LAUNCH_CHECK();
  return hipSuccess;
}


hipError_t launch_group_score(int N, const int *gidx, const int *gsz, const int *goff, const double *mblk,
                              const double *dcol, const double *part, int nrb, int p, int lm, double n_t,
                              double lambda, const double *beta_dense, const unsigned char *always, double *bd,
                              hipStream_t st, int smax, double *work, double *zwork, const FitCtrl *ctrl, int slot,
                              int eig_mode, double *eig_v, double *eig_l) {
  hipLaunchKernelGGL(k_group_score, dim3((N + 63) / 64), dim3(64), 0, st, N, gidx, gsz, goff, mblk, dcol, part, nrb, p,
                     lm, n_t, lambda, beta_dense, always, bd, ctrl, slot, eig_mode, eig_v, eig_l);
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


}  // namespace bessx
