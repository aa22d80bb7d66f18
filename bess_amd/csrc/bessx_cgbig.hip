// bessx_cgbig.hip -- the restricted LM fit of the covariance form for MORE than 254 active columns.
//
// primary_model_fit (src/Algorithm.h:1131-1135) solves (X_A^T X_A + lambda I) b = X_A^T y.  Up to 254 unknowns the system
// lives in the registers of one workgroup (k_cgr / k_cg / k_chol, bessx_kernels.hip); beyond that, round 1-3 ran a blocked
// right-looking Cholesky in global memory -- three small launches per 16-column block, 384 launches at k = 2046, and the
// reference's DEFAULT sequence 1..min(p, n / log n) lives almost entirely there (python/bess/linear.py:285-287).  After
// normalisation the Gram matrix of an active set is n (I + small): conjugate gradients reach rounding level in a few
// dozen steps, and a step is ONE product with the k x k matrix -- work for the whole chip, not for one compute unit.
//
//   k_cgb_gather   dense symmetric copy A[j][i] = G[slot(a_j)][a_i] of the cached Gram entries (coalesced writes)
//   k_cgb_init     r = q - (A + lambda I) x0 with x0 = the previous coefficients on the new set; |q|^2
//   k_cgb_step     ONE launch per CG step, no grid barrier: every workgroup redoes the O(k) vector work of the step
//                  for the whole vector (same data, same order: bit-identical scalars everywhere) and multiplies ITS
//                  8 rows of A with the new search direction; vectors and scalars are double-buffered by step parity
//                  so that nobody overwrites what a slower workgroup of the same launch still reads
//   k_cgb_resid    the TRUE residual q - (A + lambda I) x of the accepted iterate (the recurrence drifts)
//   k_cgb_accept   |residual| <= tol |q| -> the solution; else the fit is parked (cov_stall = 2) and the host issues
//                  the blocked Cholesky for the slot, exactly like the small systems' hand-over
//
// Every kernel is gated by the fit's control block like the slot kernels around it; steps beyond convergence fall through.
#include <hip/hip_runtime.h>

#include "bessx_dev.h"

namespace bessx {

namespace {

constexpr int CGB_ROWS = 8;      // rows of A per workgroup (2 per wave)
constexpr int CGB_REG = CGB_MAX_K / 256;  // vector elements a thread keeps in registers

__device__ __forceinline__ bool cgb_gate(const FitCtrl *ctrl, int slot) {
  return !(ctrl->done || ctrl->l != slot - 1 || ctrl->same_prev);
}

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// fixed-order sum over the 256 threads of the block, result in every thread
__device__ __forceinline__ double bsum256(double v, double *sm /* >= 5 */) {
  v = wsum64(v);
  __syncthreads();  // (sm may still be read from the previous use)
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__global__ void __launch_bounds__(256) k_cgb_gather(const double *__restrict__ G, int p, const int *__restrict__ slot_of,
                                                    const int *__restrict__ A_new, int k, int lda,
                                                    double *__restrict__ A, FitCtrl *__restrict__ ctrl, int slot) {
  if (!cgb_gate(ctrl, slot)) return;
  const int j = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const int sl = slot_of[A_new[j]];
  double v = 0.0;
  if (sl >= 0)
    v = G[(size_t)sl * p + A_new[i]];
  else
    ctrl->cov_miss = 1;
  A[(size_t)j * lda + i] = v;  // (symmetric: row j of A = cached Gram column of a_j at the rows a_i)
}

// the rows [8 wg, 8 wg + 8) of A times the vector in LDS; lane 0 of wave w ends with rows 2 w and 2 w + 1
__device__ __forceinline__ void rows_times(const double *__restrict__ A, int lda, int k, const double *ps, int row0,
                                           double out[2]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int row = row0 + 2 * wave + h;
    double acc = 0.0;
    if (row < k) {
      const double *__restrict__ ar = A + (size_t)row * lda;
      for (int j = lane; j < k; j += 64) acc = fma(ar[j], ps[j], acc);
    }
    out[h] = wsum64(acc);
  }
}

__global__ void __launch_bounds__(256) k_cgb_init(const double *__restrict__ A, int lda, int k, double ridge,
                                                  const double *__restrict__ xty, const int *__restrict__ A_new,
                                                  const double *__restrict__ beta_dense, CgbWork w,
                                                  const int *__restrict__ meta, FitCtrl *__restrict__ ctrl, int slot) {
  if (!cgb_gate(ctrl, slot)) return;
  __shared__ double ps[CGB_MAX_K];
  __shared__ double sm[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, row0 = wg * CGB_ROWS;
  for (int i = tid; i < k; i += 256) ps[i] = beta_dense[A_new[i]];
  __syncthreads();
  double ax[2];
  rows_times(A, lda, k, ps, row0, ax);
  double qq = 0.0;
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int row = row0 + 2 * wave + h;
      if (row < k) {
        const double q = xty[A_new[row]];
        w.q[row] = q;
        w.x[row] = ps[row];
        w.r[1][row] = q - (ax[h] + ridge * ps[row]);  // "r of step -1": step 0 takes it over with alpha = 0
        w.p[1][row] = 0.0;
        w.ap[1][row] = 0.0;
        qq = fma(q, q, qq);
      }
    }
    sm[wave] = qq;
  }
  __syncthreads();
  if (tid == 0) {
    w.part_qq[wg] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    if (wg == 0) {
      w.st->done_step = 0x7fffffff;
      w.st->steps = 0;
      ctrl->sse_valid = 0;  // (the loss terms of an earlier solve do not belong to this slot's coefficients)
      w.st->rr[0] = w.st->rr[1] = 1.0;
      // two exactly dependent columns are cached: a singular but consistent system must go to the pivoted route
      // (k_cov_compact's flag, see cov_compact_body) -- park the fit, the host issues the Cholesky for the slot
      if (meta != nullptr && meta[4]) {
        ctrl->cov_stall = 2;
        ctrl->l = -1 - ctrl->l;
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_cgb_step(const double *__restrict__ A, int lda, int k, double ridge, int t,
                                                  int nwg, double tol, CgbWork w, const FitCtrl *__restrict__ ctrl,
                                                  int slot) {
  if (!cgb_gate(ctrl, slot)) return;
  if (w.st->done_step < t) return;  // converged in an earlier launch (written before this launch began)
  __shared__ double ps[CGB_MAX_K];
  __shared__ double sm[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, row0 = wg * CGB_ROWS;
  const int cur = t & 1, prv = cur ^ 1;
  // alpha of the previous step from its p.Ap (summed over the workgroups' parts in a fixed order, by everybody)
  double alpha = 0.0, rr_prev = 1.0;
  if (t > 0) {
    double s = 0.0;
    for (int i = tid; i < nwg; i += 256) s += w.part_pq[prv][i];
    const double pq = bsum256(s, sm);
    rr_prev = w.st->rr[prv];
    alpha = rr_prev / pq;
  }
  double qq;
  {
    double s = 0.0;
    for (int i = tid; i < nwg; i += 256) s += w.part_qq[i];
    qq = bsum256(s, sm);
  }
  // r_t = r_{t-1} - alpha A p_{t-1} for the WHOLE vector (kept in registers), |r_t|^2
  double rl[CGB_REG];
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < CGB_REG; e++) {
    const int i = tid + 256 * e;
    rl[e] = 0.0;
    if (i < k) {
      rl[e] = fma(-alpha, w.ap[prv][i], w.r[prv][i]);
      acc = fma(rl[e], rl[e], acc);
    }
  }
  const double rr = bsum256(acc, sm);
  // x_t of the workgroup's own rows
  if (t > 0 && tid < CGB_ROWS && row0 + tid < k) w.x[row0 + tid] = fma(alpha, w.p[prv][row0 + tid], w.x[row0 + tid]);
  // recurrence target |r| <= tol / 100 |q| (the accepted iterate is checked on its true residual, k_cgb_resid);
  // a NaN ends the iteration too and fails that check
  if (!(rr > 1e-4 * tol * tol * qq)) {
    if (wg == 0 && tid == 0) {
      w.st->done_step = t;
      w.st->steps = t;
      w.st->qq = qq;
    }
    return;
  }
  const double beta = t > 0 ? rr / rr_prev : 0.0;
#pragma unroll
  for (int e = 0; e < CGB_REG; e++) {
    const int i = tid + 256 * e;
    if (i < k) {
      const double pv = fma(beta, w.p[prv][i], rl[e]);
      ps[i] = pv;
      if (i >= row0 && i < row0 + CGB_ROWS) {  // this workgroup's rows of the new vectors
        w.r[cur][i] = rl[e];
        w.p[cur][i] = pv;
      }
    }
  }
  __syncthreads();
  double ax[2];
  rows_times(A, lda, k, ps, row0, ax);
  double pq = 0.0;
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int row = row0 + 2 * wave + h;
      if (row < k) {
        const double v = fma(ridge, ps[row], ax[h]);
        w.ap[cur][row] = v;
        pq = fma(ps[row], v, pq);
      }
    }
    sm[4 + wave] = pq;
  }
  __syncthreads();
  if (tid == 0) {
    w.part_pq[cur][wg] = ((sm[4] + sm[5]) + sm[6]) + sm[7];
    if (wg == 0) {
      w.st->rr[cur] = rr;
      w.st->qq = qq;
      w.st->steps = t + 1;
    }
  }
}

__global__ void __launch_bounds__(256) k_cgb_resid(const double *__restrict__ A, int lda, int k, double ridge, CgbWork w,
                                                   const FitCtrl *__restrict__ ctrl, int slot) {
  if (!cgb_gate(ctrl, slot)) return;
  __shared__ double ps[CGB_MAX_K];
  __shared__ double sm[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, row0 = wg * CGB_ROWS;
  for (int i = tid; i < k; i += 256) ps[i] = w.x[i];
  __syncthreads();
  double ax[2];
  rows_times(A, lda, k, ps, row0, ax);
  double ss = 0.0;
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int row = row0 + 2 * wave + h;
      if (row < k) {
        const double rho = w.q[row] - (ax[h] + ridge * ps[row]);
        w.r[0][row] = rho;  // (the step buffers are free now: k_cgb_accept forms the loss terms from it)
        ss = fma(rho, rho, ss);
      }
    }
    sm[wave] = ss;
  }
  __syncthreads();
  if (tid == 0) w.part_pq[0][wg] = ((sm[0] + sm[1]) + sm[2]) + sm[3];  // (the step buffers are free now)
}

__global__ void __launch_bounds__(256) k_cgb_accept(const double *__restrict__ A, int lda, int k, int nwg, double tol,
                                                    double ridge, double yy, CgbWork w, double *__restrict__ sol,
                                                    FitCtrl *__restrict__ ctrl, int slot) {
  if (!cgb_gate(ctrl, slot)) return;
  __shared__ double sm[8];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int i = tid; i < nwg; i += 256) s += w.part_pq[0][i];
  const double rho2 = bsum256(s, sm);
  const bool ok = w.st->done_step != 0x7fffffff && rho2 <= tol * tol * w.st->qq;  // (a NaN fails the comparison)
  if (!ok) {
    if (tid == 0) {  // park the fit: the host issues the blocked Cholesky for this slot
      ctrl->cov_stall = 2;
      ctrl->l = -1 - ctrl->l;
    }
    return;
  }
  // the solution, and the loss without a pass over X (as in the small systems' solve, cgr_body):
  // |y - X b|^2 = y.y - b.(q + rho) - ridge |b|^2 with rho the residual of the normal equations just recomputed
  double t1 = 0.0, t2 = 0.0, t3 = 0.0, gd = 0.0;
  for (int i = tid; i < k; i += 256) {
    const double x = w.x[i], qr = w.q[i] + w.r[0][i];
    sol[i] = x;
    t1 = fma(x, qr, t1);
    t2 = fma(x, x, t2);
    t3 = fma(fabs(x), fabs(qr), t3);
    gd = fmax(gd, A[(size_t)i * lda + i]);
  }
  const double a1 = bsum256(t1, sm), a2 = bsum256(t2, sm), a3 = bsum256(t3, sm);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) gd = fmax(gd, __shfl_xor(gd, o));
  __syncthreads();
  if ((tid & 63) == 0) sm[tid >> 6] = gd;
  __syncthreads();
  gd = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
  if (tid == 0) {
    const double tr = yy - a1 - ridge * a2;
    ctrl->sse_dot = a1;
    ctrl->sse_nrm = a2;
    ctrl->sse_valid = (tr > 1e-6 * yy && 4e-16 * (a3 + (ridge + gd) * a2 + yy) <= 1e-10 * tr) ? 1 : 0;
    ctrl->irls_steps = w.st->steps;  // (the commit moves it to irls_last: the host sizes its next batch of step launches
                                     // from it)
  }
}

}  // namespace

#define LAUNCH_CHECK()                  \
  do {                                  \
    hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) return e__;  \
  } while (0)

size_t cgb_work_doubles(int kcap) {
  const size_t kc = (size_t)(kcap + 15) / 16 * 16, nwg = (kc + CGB_ROWS - 1) / CGB_ROWS;
  return kc * kc + 8 * kc + 3 * nwg + 16;
}

CgbWork cgb_carve(double *base, int kcap) {
  const size_t kc = (size_t)(kcap + 15) / 16 * 16, nwg = (kc + CGB_ROWS - 1) / CGB_ROWS;
  CgbWork w;
  double *q = base + kc * kc;
  w.x = q;
  w.q = q + kc;
  w.r[0] = q + 2 * kc;
  w.r[1] = q + 3 * kc;
  w.p[0] = q + 4 * kc;
  w.p[1] = q + 5 * kc;
  w.ap[0] = q + 6 * kc;
  w.ap[1] = q + 7 * kc;
  double *t = q + 8 * kc;
  w.part_pq[0] = t;
  w.part_pq[1] = t + nwg;
  w.part_qq = t + 2 * nwg;
  w.st = reinterpret_cast<CgbState *>(t + 3 * nwg);
  return w;
}

// One solve: gather, start, `nsteps` step launches (those beyond convergence fall through), true residual, verdict.
hipError_t launch_cg_big(const double *G, int p, const int *slot_of, const int *meta, const int *A_new, int k,
                         double ridge, const double *xty, const double *beta_dense, double *work, int kcap, double *sol,
                         FitCtrl *ctrl, int slot, int nsteps, double tol, double yy, hipStream_t st) {
  if (k < 1 || k > kcap || kcap > CGB_MAX_K || nsteps < 1) return hipErrorInvalidValue;
  const int lda = (kcap + 15) / 16 * 16;
  const int nwg = (k + CGB_ROWS - 1) / CGB_ROWS;
  const CgbWork w = cgb_carve(work, kcap);
  double *A = work;
  hipLaunchKernelGGL(k_cgb_gather, dim3((k + 255) / 256, k), dim3(256), 0, st, G, p, slot_of, A_new, k, lda, A, ctrl, slot);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cgb_init, dim3(nwg), dim3(256), 0, st, (const double *)A, lda, k, ridge, xty, A_new, beta_dense, w,
                     meta, ctrl, slot);
  LAUNCH_CHECK();
  for (int t = 0; t < nsteps; t++) {
    hipLaunchKernelGGL(k_cgb_step, dim3(nwg), dim3(256), 0, st, (const double *)A, lda, k, ridge, t, nwg, tol, w,
                       (const FitCtrl *)ctrl, slot);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_cgb_resid, dim3(nwg), dim3(256), 0, st, (const double *)A, lda, k, ridge, w, (const FitCtrl *)ctrl,
                     slot);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cgb_accept, dim3(1), dim3(256), 0, st, (const double *)A, lda, k, nwg, tol, ridge, yy, w, sol, ctrl,
                     slot);
  LAUNCH_CHECK();
  return hipSuccess;
}

}  // namespace bessx
