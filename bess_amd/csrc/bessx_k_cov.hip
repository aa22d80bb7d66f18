// bessx_k_cov.hip -- the covariance-update form of the LM score pass: Gram column cache, panel kernels, fills (+ their launchers)
#include "bessx_kdev.hpp"

namespace bessx {

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

// Diagnostic build only (make prof DEFS=-DBESSX_PANEL_CLOCK, tools/panel_bench.py): the shader clock the panel kernels
// really run at = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz), block 0 of every launch prints it.
// MI355X_MICROARCH.md, "DVFS give-back": MFMA-dense loops are held well under 2.4 GHz.
#ifdef BESSX_PANEL_CLOCK
#define PCLK_BEGIN()                                                  \
  const unsigned long long pclk_t0_ = __builtin_amdgcn_s_memtime();   \
  const unsigned long long pclk_r0_ = __builtin_amdgcn_s_memrealtime()
#define PCLK_END(tag)                                                                                           \
  do {                                                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
    if (blockIdx.x == 7 && threadIdx.x == 0) {                                                                  \
      const unsigned long long dt_ = __builtin_amdgcn_s_memtime() - pclk_t0_;                                   \
      const unsigned long long dr_ = __builtin_amdgcn_s_memrealtime() - pclk_r0_;                               \
      printf("PCLK %s cycles %llu ticks100MHz %llu MHz %.0f\n", tag, dt_, dr_, dr_ ? 100.0 * (double)dt_ / (double)dr_ : 0.0); \
    }                                                                                                           \
  } while (0)
#else
#define PCLK_BEGIN()
#define PCLK_END(tag)
#endif

__device__ __forceinline__ bool cov_gate(const FitCtrl *ctrl, int slot) {
  if (ctrl->done) return false;
  if (slot == 0) return ctrl->l == 0;  // start of a fit
  return ctrl->l == slot - 1 && !ctrl->same_prev;
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
                                                       int parked, int spec_max, int spec, int spec_min) {
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
  // spec_max = 32: the list is rounded up to the next multiple of 32 that leaves room for >= spec_min speculative columns
  // (round 5: 8, before 16 -- the second group of a two-group launch costs 0.46 ms where a launch of its own costs 0.78,
  // so 50 missing columns are better served by 64 now and 64 at the next miss than by 96 + 32 + 32: DESIGN.md 3a);
  // spec_max = 64 (pair panel kernel: two groups per pass over X): to the next multiple of 64
  // spec = 2 (round 5; caches that are never started over): the host selected 64 extras, so that the list is FULL up to
  // its multiple of 32 whatever nm is (with 32 extras a list of 50 missing columns was rounded to 96 with 14 empty
  // places -- columns of a pass over X that computed nothing)
  const int pool = spec == 2 ? 2 * COV_R : spec_max;
  const int room = spec ? min(((nm + spec_min + spec_max - 1) / spec_max) * spec_max - nm, pool) : 0;
  __shared__ int s_ne;
  if (tid < 64) {
    const int col = (spec && tid < pool) ? extras[tid] : -1;
    const double sc = col >= 0 ? bd2[col] : -1.0;
    // a genuine uncached column (slot_of here is the WRITER's map: with staged fills -- chunk chains side by side --
    // a column another chain cached after this chain's look-up is dropped as well)
    const bool valid = col >= 0 && sc >= 0.0 && slot_of[col] < 0;
    const unsigned long long bal = __ballot(valid);
    // `extras` comes from the selection kernel in ascending COLUMN order; of a pool larger than the room the best by
    // SCORE are wanted (ties: lower column): rank of this lane's column among the valid ones
    int rank = 0;
    if (pool > spec_max) {
      for (int u = 0; u < 64; u++) {
        const double su = __shfl(sc, u);
        const int cu = __shfl(col, u);
        const bool vu = (bal >> u) & 1ull;
        if (vu && (su > sc || (su == sc && cu < col))) rank++;
      }
    } else {
      rank = __popcll(bal & ((1ull << tid) - 1ull));
    }
    if (valid && rank < room) fcols[nm + rank] = col;
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

// Staged fills (chunk chains side by side, bessx_sync.h: FillRendezvous, concurrent rounds): the launches of a fill work
// on the writer's slot map; the readers' map gets the new entries here, behind the last launch that writes the columns.
__global__ void __launch_bounds__(256) k_cov_publish_slots(const int *__restrict__ fcols, const int *__restrict__ slot_w,
                                                           int *__restrict__ slot_of, const FitCtrl *__restrict__ ctrl) {
  if (!ctrl->cov_stall) return;
  const int nf = ctrl->cov_nfill;
  for (int i = threadIdx.x; i < nf; i += 256) {
    const int col = fcols[i];
    if (col >= 0) slot_of[col] = slot_w[col];
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
  PCLK_BEGIN();
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
  PCLK_END("lds2");
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
  PCLK_BEGIN();
  if ((g0 + 1) * COV_R < nfill)     // uniform
    cov_pair_body<MASKED, true>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
  else
    cov_pair_body<MASKED, false>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
  PCLK_END("pair");
}

// ------------------------------------------------------------------------------------------
// Round 5: the panel pass as ONE 8-wave workgroup per compute unit over 128 streamed columns, two LDS tiles, one barrier
// per chunk (k_cov_panel_dp).
//
// What the counters and the in-kernel clock said about the two kernels above (profiles/r05_panel_*, README): HBM
// traffic is the algorithmic 1.03 x, the LDS is 35 % busy, the waves wait to issue 70-76 % of their time -- and the
// matrix pipe is busy 0.77 / 0.81 of the cycles that REALLY pass: the chip holds its clock at 1.5-2.1 GHz under these
// kernels (s_memtime against s_memrealtime; the 32-column kernel, which moves more bytes per flop, clocks LOWER than
// the pair kernel), which is where the distance to the 2.4 GHz peaks comes from.  What a kernel can still change is the
// data it moves per flop -- every 64-column block stages its own copy of the right-hand-side columns from L2 (half as
// many bytes again as X itself at 32 columns, as many again at 64) -- and the phases a workgroup stands still in.  Here:
//   * 128 streamed columns per workgroup: half the right-hand-side traffic (L2 -> LDS) per byte of X;
//   * 8 waves, wave w multiplies streamed tile w with every right-hand-side tile: two waves per SIMD, as before;
//   * 32-row chunks, TWO LDS tiles: while chunk k is multiplied out of one, chunk k + 1 is written into the other --
//     the stores sit between the matrix instructions of the chunk's first row step, the global loads of chunk k + 3
//     between those of the second -- and ONE barrier per chunk (the kernels above: multiply | barrier | stage | barrier);
//   * operands of the next row step are read while the current one multiplies; loads run two to three chunks ahead in
//     two register stages; every staging operation is unconditional (a branch around one costs the compiler its count
//     of the loads in flight).
// NT = 2: one 32-column group per pass, NT = 4: a pair.  Same partial-sum layout as the kernels above (k_cov_reduce is
// unchanged).  104 KB of LDS at NT = 4, 87 KB at NT = 2.
// ------------------------------------------------------------------------------------------
constexpr int DP_RB = 32;            // rows per chunk
constexpr int DP_LD = DP_RB + CP_PAD;  // padded row stride of a column in LDS (doubles)
constexpr int DP_SC = 128;           // streamed columns per workgroup
template <bool MASKED, int NT>
__device__ __forceinline__ void cov_dp_body(const double *__restrict__ X, const double *__restrict__ aux, long ld,
                                            int p, const double *__restrict__ mask, const int *__restrict__ fcols,
                                            int g0, int rows_per_slab, int nslab, int njg,
                                            double *__restrict__ part, double *smem) {
  constexpr int NS = DP_SC / 32;                // loads of streamed columns per thread and chunk
  constexpr int NL = NS + NT / 2;               // ... and of right-hand-side columns behind them
  constexpr int COLS = DP_SC + 16 * NT;         // columns staged per chunk
  constexpr size_t BUF = (size_t)COLS * DP_LD;  // doubles per LDS tile
  const int njg2 = (njg + 1) / 2;               // workgroups per slab (128 streamed columns each)
  const int slab = blockIdx.x / njg2, jb = blockIdx.x - slab * njg2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, q = lane >> 4;
  const int ru = tid & 15, cbase = tid >> 4;    // 16 threads x 2 rows cover a chunk of one column; 32 columns per load
  // streamed columns jb * 128 + cbase + 32 i (a column beyond p re-reads the last existing one of its thread: its
  // products land in rows >= p, which the reduce kernel never stores)
  const int jc = min(jb * DP_SC + cbase, p - 1);
  const double *sx = X + (size_t)jc * ld + 2 * ru;
  const long sstride = 32 * ld;
  const int ilim = jb * DP_SC + cbase < p ? (p - 1 - (jb * DP_SC + cbase)) / 32 : 0;  // last i whose column exists
  const double *src[NT / 2];
#pragma unroll
  for (int i = 0; i < NT / 2; i++) src[i] = gram_col(X, aux, ld, fcols[g0 * COV_R + i * 32 + cbase]) + 2 * ru;
  const long r_begin = (long)slab * rows_per_slab, r_end = min(r_begin + rows_per_slab, ld);
  const int nchunk = (int)((r_end - r_begin + DP_RB - 1) / DP_RB);
  d2 stA[NL], stB[NL], mA = d2{1.0, 1.0}, mB = d2{1.0, 1.0};
#define DP_LOAD1(st, ms, r, i)                                                                                  \
  do {                                                                                                          \
    if ((i) < NS)                                                                                               \
      st[i] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(sx + min((i), ilim) * sstride + (r)));    \
    else                                                                                                        \
      st[i] = *reinterpret_cast<const d2 *>(src[(i) >= NS ? (i)-NS : 0] + (r));                                 \
    if (MASKED && (i) == NL - 1) ms = *reinterpret_cast<const d2 *>(mask + (r) + 2 * ru);                       \
  } while (0)
#define DP_STORE1(buf, st, ms, i)                                                                               \
  do {                                                                                                          \
    d2 v__ = st[i];                                                                                             \
    if (MASKED && (i) >= NS) v__ = v__ * ms;                                                                    \
    *reinterpret_cast<d2 *>((buf) + 2 * ru + (size_t)((i)*32 + cbase) * DP_LD) = v__;                           \
  } while (0)
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  const size_t offa = (size_t)(wv * 16 + c) * DP_LD + 4 * q, offb = (size_t)(DP_SC + c) * DP_LD + 4 * q;
  struct Ops {
    d2 a0, a1, b0[NT], b1[NT];
  };
  auto read_ops = [&](const double *buf, int st, Ops &o) {
    o.a0 = *reinterpret_cast<const d2 *>(buf + offa + 16 * st);
    o.a1 = *reinterpret_cast<const d2 *>(buf + offa + 16 * st + 2);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      o.b0[t] = *reinterpret_cast<const d2 *>(buf + offb + (size_t)t * 16 * DP_LD + 16 * st);
      o.b1[t] = *reinterpret_cast<const d2 *>(buf + offb + (size_t)t * 16 * DP_LD + 16 * st + 2);
    }
  };
  // One chunk: DP_RB / 16 row steps of 4 x NT matrix instructions; staging operation n (0 .. 2 NL - 1: the NL stores of
  // the next chunk into the other tile, then the NL loads of the chunk three ahead into the stage just stored) goes
  // behind the matrix instructions, spread evenly.
#define DP_CHUNK(cur, nxt, st, ms, rload)                                                                       \
  do {                                                                                                          \
    Ops oa__, ob__;                                                                                             \
    read_ops(cur, 0, oa__);                                                                                     \
    int n__ = 0;                                                                                                \
    _Pragma("unroll") for (int s__ = 0; s__ < DP_RB / 16; s__++) {                                            \
      Ops &o__ = (s__ & 1) ? ob__ : oa__;                                                                       \
      if (s__ + 1 < DP_RB / 16) read_ops(cur, s__ + 1, (s__ & 1) ? oa__ : ob__);                                \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      _Pragma("unroll") for (int j__ = 0; j__ < 4; j__++) {                                                   \
        const double a__ = j__ == 0 ? o__.a0.x : (j__ == 1 ? o__.a0.y : (j__ == 2 ? o__.a1.x : o__.a1.y));       \
        _Pragma("unroll") for (int t__ = 0; t__ < NT; t__++) {                                                \
          const double b__ = j__ == 0 ? o__.b0[t__].x : (j__ == 1 ? o__.b0[t__].y : (j__ == 2 ? o__.b1[t__].x : o__.b1[t__].y)); \
          acc[t__] = __builtin_amdgcn_mfma_f64_16x16x4f64(a__, b__, acc[t__], 0, 0, 0);                         \
        }                                                                                                       \
        /* staging operations due after this group of NT matrix instructions */                                 \
        const int due__ = (2 * NL * (s__ * 4 + j__ + 1)) / (4 * (DP_RB / 16));                                  \
        _Pragma("unroll") for (int k__ = 0; k__ < 2 * NL; k__++) {                                            \
          if (k__ >= n__ && k__ < due__) {                                                                      \
            if (k__ < NL)                                                                                       \
              DP_STORE1(nxt, st, ms, k__);                                                                      \
            else                                                                                                \
              DP_LOAD1(st, ms, rload, k__ - NL);                                                                \
          }                                                                                                     \
        }                                                                                                       \
        n__ = due__;                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)
  double *buf0 = smem, *buf1 = smem + BUF;
  // a load beyond the slab's last chunk re-reads that last chunk (cache hits); the tile it is stored into is never multiplied
  const long r_last = r_begin + (long)(nchunk - 1) * DP_RB;
  // prologue: chunk 0 -> stage A -> tile 0; chunk 1 -> stage B; chunk 2 -> stage A
#pragma unroll
  for (int i = 0; i < NL; i++) DP_LOAD1(stA, mA, r_begin, i);
#pragma unroll
  for (int i = 0; i < NL; i++) DP_STORE1(buf0, stA, mA, i);
  {
    const long r1 = min(r_begin + DP_RB, r_last), r2 = min(r_begin + 2 * DP_RB, r_last);
#pragma unroll
    for (int i = 0; i < NL; i++) DP_LOAD1(stB, mB, r1, i);
#pragma unroll
    for (int i = 0; i < NL; i++) DP_LOAD1(stA, mA, r2, i);
  }
  __syncthreads();
  // chunk k is in tile k & 1; chunk k + 1 waits in a register stage (odd chunks in B, even ones in A); during chunk k
  // that stage is stored into the other tile and reloaded with chunk k + 3
  for (int k = 0; k < nchunk; k += 2) {
    {
      const long rl = min(r_begin + (long)(k + 3) * DP_RB, r_last);
      DP_CHUNK(buf0, buf1, stB, mB, rl);
    }
    __syncthreads();
    if (k + 1 >= nchunk) break;
    {
      const long rl = min(r_begin + (long)(k + 4) * DP_RB, r_last);
      DP_CHUNK(buf1, buf0, stA, mA, rl);
    }
    __syncthreads();
  }
#undef DP_CHUNK
#undef DP_LOAD1
#undef DP_STORE1
  // streamed tile wv of this workgroup = tile wv & 3 of the 64-column group 2 jb + (wv >> 2) in the layout of the
  // kernels above; the last workgroup's second half may lie beyond the last group
  const int jg = 2 * jb + (wv >> 2);
  if (jg < njg) {
    const size_t tiles_per_slab = (size_t)njg * COV_NJ * 2;
    double *out = part + ((size_t)slab * tiles_per_slab + (size_t)(jg * COV_NJ + (wv & 3)) * 2) * 256;
    *reinterpret_cast<d4 *>(out + lane * 4) = acc[0];
    *reinterpret_cast<d4 *>(out + 256 + lane * 4) = acc[1];
    if (NT == 4) {
      double *out2 = out + (size_t)nslab * tiles_per_slab * 256;
      *reinterpret_cast<d4 *>(out2 + lane * 4) = acc[NT - 2];
      *reinterpret_cast<d4 *>(out2 + 256 + lane * 4) = acc[NT - 1];
    }
  }
}

// grid: one 512-thread block per (slab, 128 streamed columns); ngroups = 1: group g0 only (NT = 2), 2: the pair g0,
// g0 + 1 -- or group g0 alone if the second one is beyond the fill list (decided on the device)
template <bool MASKED>
__global__ void __launch_bounds__(512, 1) k_cov_panel_dp(const double *__restrict__ X, const double *__restrict__ aux,
                                                         long ld, int p, const double *__restrict__ mask,
                                                         const int *__restrict__ fcols, int g0, int ngroups,
                                                         int rows_per_slab, int nslab, int njg,
                                                         double *__restrict__ part, const FitCtrl *__restrict__ ctrl,
                                                         int big) {
  KT(5);
  if (big ? !ctrl->cov_stall : (ctrl->done || ctrl->l != 0)) return;
  const int nfill = ctrl->cov_nfill;
  if (g0 * COV_R >= nfill) return;
  extern __shared__ double smem[];
  PCLK_BEGIN();
  if (ngroups >= 2 && (g0 + 1) * COV_R < nfill)  // uniform
    cov_dp_body<MASKED, 4>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
  else
    cov_dp_body<MASKED, 2>(X, aux, ld, p, mask, fcols, g0, rows_per_slab, nslab, njg, part, smem);
  PCLK_END("dp");
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
// (slot < 0: the caller has checked the gate itself -- the merged launches over chunk chains, k_mc_cov_d)
__device__ __forceinline__ void cov_d_body(const double *__restrict__ G, int p, const int *__restrict__ slot_of,
                                           const double *__restrict__ xty, const int *__restrict__ A_cur,
                                           const double *__restrict__ b_cur, double *__restrict__ d_out,
                                           const double *__restrict__ beta_dense, const double *__restrict__ xtx,
                                           double n_t, double lambda, const unsigned char *__restrict__ always,
                                           double *__restrict__ bd, const unsigned char *__restrict__ inA,
                                           double *__restrict__ bmm, const FitCtrl *__restrict__ ctrl, int slot) {
  // what the epilogue needs of this block's 32 columns does not depend on the control block: these loads are in
  // flight while the gate below waits for its own (one round trip less on the block's critical path)
  const int jj = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jj;
  const bool epi = g == 0 && j < p;
  const double e_xty = epi ? xty[j] : 0.0, e_b = epi ? beta_dense[j] : 0.0, e_xtx = epi ? xtx[j] : 1.0;
  const unsigned char e_in = epi ? inA[j] : (unsigned char)0;
  const unsigned char e_al = (epi && always != nullptr) ? always[j] : (unsigned char)0;
  if (slot >= 0 && (ctrl->done || ctrl->l != slot - 1)) return;
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

__global__ void __launch_bounds__(256) k_cov_d(const double *__restrict__ G, int p, const int *__restrict__ slot_of,
                                               const double *__restrict__ xty, const int *__restrict__ A_cur,
                                               const double *__restrict__ b_cur, double *__restrict__ d_out,
                                               const double *__restrict__ beta_dense, const double *__restrict__ xtx,
                                               double n_t, double lambda, const unsigned char *__restrict__ always,
                                               double *__restrict__ bd, const unsigned char *__restrict__ inA,
                                               double *__restrict__ bmm, const FitCtrl *__restrict__ ctrl, int slot) {
  KT(4);
  cov_d_body(G, p, slot_of, xty, A_cur, b_cur, d_out, beta_dense, xtx, n_t, lambda, always, bd, inA, bmm, ctrl, slot);
}

// The score pass of every chunk chain whose coefficients changed since its last one, in ONE launch (bessx_dev.h, McChain):
// blockIdx.y = chain, blockIdx.x = its block of 32 columns.  Chains that are finished, parked or whose scores are still
// those of their coefficients fall through.
__global__ void __launch_bounds__(256) k_mc_cov_d(const McChain *__restrict__ chains) {
  const McChain &ch = chains[blockIdx.y];
  const McState *st = ch.state;
  if (st->finished || st->parked || !st->need_d) return;
  const FitCtrl *ctrl = ch.nd.ctrl;
  if (ctrl->done || ctrl->l < 0) return;
  cov_d_body(ch.G, ch.p, ch.nd.slot_of, ch.xty, ch.nd.A_cur, ch.fz.b_cur, ch.d_out, ch.fz.beta_dense, ch.xtx, ch.n_t,
             ch.lambda, ch.always, ch.bd, ch.nd.inA, ch.bmm, ctrl, -1);
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


hipError_t launch_cov_need(const int *list, int len, const double *bd, double *bd2, int p, int *slot_of, int *meta,
                           int C, int *fcols, FitCtrl *ctrl, int slot, const int *A_cur, hipStream_t st,
                           int no_restart) {
  hipLaunchKernelGGL(k_cov_need, dim3(1), dim3(256), 0, st, list, len, bd, bd2, p, slot_of, meta, C, fcols, ctrl, slot,
                     A_cur, no_restart);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_publish_slots(const int *fcols, const int *slot_w, int *slot_of, const FitCtrl *ctrl, hipStream_t st) {
  hipLaunchKernelGGL(k_cov_publish_slots, dim3(1), dim3(256), 0, st, fcols, slot_w, slot_of, ctrl);
  LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cov_fill_list(int *fcols, const int *extras, const double *bd2, int *slot_of, int *meta,
                                FitCtrl *ctrl, int parked, hipStream_t st, int spec_max, int spec, int spec_min) {
  if (spec_min <= 0) spec_min = spec_max / 2;
  hipLaunchKernelGGL(k_cov_fill_list, dim3(1), dim3(256), 0, st, fcols, extras, bd2, slot_of, meta, ctrl, parked,
                     spec_max, spec, spec_min);
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
  if (variant == 5 && ngroups <= 2) {
    // one 8-wave block per (slab, 128-column group): k_cov_panel_dp (one or both groups in one pass over X)
    const size_t ldsd = (size_t)2 * (DP_SC + 64) * DP_LD * sizeof(double);
    const long nbd = (long)nslab * ((njg + 1) / 2);
    if (mask)
      hipLaunchKernelGGL(k_cov_panel_dp<true>, dim3((unsigned)nbd), dim3(512), ldsd, st, X, aux, ld, p, mask, fcols, g0,
                         ngroups, rows_per_slab, nslab, njg, part, ctrl, parked);
    else
      hipLaunchKernelGGL(k_cov_panel_dp<false>, dim3((unsigned)nbd), dim3(512), ldsd, st, X, aux, ld, p, mask, fcols, g0,
                         ngroups, rows_per_slab, nslab, njg, part, ctrl, parked);
    LAUNCH_CHECK();
    return hipSuccess;
  }
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
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_panel_pair<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
  if (e != hipSuccess) return e;
  const int ldsd = (int)((size_t)2 * (DP_SC + 64) * DP_LD * sizeof(double));
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_panel_dp<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, ldsd);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_panel_dp<false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, ldsd);
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

hipError_t launch_mc_cov_d(const McChain *chains, int nchains, int p, hipStream_t st) {
  hipLaunchKernelGGL(k_mc_cov_d, dim3((p + 31) / 32, nchains), dim3(256), 0, st, chains);
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


}  // namespace bessx
