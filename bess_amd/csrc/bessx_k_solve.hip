// bessx_k_solve.hip -- max_k (selection) and the k x k solves: Cholesky, conjugate gradients by tiles and by rows, the fused selection +
// solve, the blocked Cholesky for large systems (+ their launchers)
#include "bessx_kdev.hpp"

namespace bessx {

// ------------------------------------------------------------------------------------------
// K4: max_k (src/utilities.cpp:179-188).  One 1024-thread workgroup selects the k largest of
// len <= 32768 non-negative doubles (ties -> lower index) and writes their indices ascending.
// Keys are the raw bit patterns (monotone for non-negative doubles; NaN sorts above +inf).
// The threshold (k-th largest key) is found bit by bit with ballot/popcount counting, the
// selection is a block-wide ordered compaction.  idx_in (optional) gives the original index of
// every element (second level of the two-level selection for len > 32768).
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long score_key(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? 0ull : b;  // -0.0 / negative (never produced) -> smallest
}



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
  // (m, the size of the system, is the sparsity level fz.T0 of the fit: the merged launches over chunk chains hand one
  // CholFuse per chain over for all its candidates and the level with m)
  if (ctrl->same_prev) {
    commit_body(fz.ctrl, slot, m, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
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
  commit_body(fz.ctrl, slot, m, A_new, sol, 0, 0, fz.A_cur, fz.b_cur, fz.beta_dense, fz.hist, fz.hist_beta,
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
// Merged launches over chunk chains (bessx_dev.h, McChain): one workgroup per chain does what the host's slot protocol
// and k_sel_cgr did for it -- record a finished candidate, open the next one (k_fit_continue), select, look the columns
// up, solve, commit -- with the very bodies of k_sel_cgr (topk_body, cgr_body: same instances per level, same gates
// through the control block), so a chain's arithmetic is the single chain's, operation by operation.
// ------------------------------------------------------------------------------------------
// (the selection and the four instances of the solve as functions of their own: inlined into one kernel they shared one
// register allocation and spilled 0.7-1.2 KB per thread)
template <int EB>
__device__ __noinline__ void mc_topk_call(const double *score, int p, int k, int *out, const FitCtrl *ctrl, int slot,
                                          const TopkNeed *nd) {
  topk_body<EB, 512>(score, nullptr, p, p, k, out, nullptr, ctrl, slot, nullptr, *nd);
}
template <int RPT, int NCW>
__device__ __noinline__ void mc_cgr_call(int m, int nc, double ridge, const double *rhs, const int *A_new, double *sol,
                                         const FitCtrl *ctrl, int slot, const CholFuse *fz, int maxit, double tol) {
  cgr_body<RPT, NCW>(m, nc, ridge, rhs, A_new, sol, ctrl, slot, *fz, maxit, tol);
}

template <int EB>
__global__ void __launch_bounds__(512) k_mc_sel_cgr(const McChain *__restrict__ chains) {
  // (grid = (1, chains): the selection body takes blockIdx.x for the chunk of the scores it works on -- always 0 here)
  const McChain &ch = chains[blockIdx.y];
  McState *st = ch.state;
  const int tid = threadIdx.x;
  if (st->finished || st->parked) return;  // uniform
  // (nd / nd1 and fz are read where they lie -- uniform addresses, scalar loads; thread-private copies of the two
  // structures went to scratch memory and every field read became a private-memory load: 59 us per launch)
  const TopkNeed &nd = ch.nd;
  const CholFuse &fz = ch.fz;
  FitCtrl *ctrl = nd.ctrl;
  {
    // the score pass of this round has run for this chain (k_mc_cov_d, the launch before): the scores are fresh
    const int nd_ = st->need_d;
    __syncthreads();
    if (nd_ && ctrl->done == 0 && tid == 0) st->need_d = 0;
    __syncthreads();
    if (nd_ && ctrl->done) {
      // (a fit that ended on a cycle right after a solve: its scores were never asked for; fall into the loop below)
    }
  }
  for (int pass = 0; pass < 4; pass++) {
    __syncthreads();
    const int resume = st->resume;  // uniform
    if (!resume) {
      if (ctrl->done) {
        // ---- the candidate is finished: its record, then the next one (k_fit_continue on the device state)
        const int cand = st->cand, T0 = ctrl->T0, W = ch.width;
        for (int i = tid; i < W; i += 512) {
          ch.rec_A[(size_t)cand * W + i] = i < T0 ? fz.A_cur[i] : -1;
          ch.rec_b[(size_t)cand * W + i] = i < T0 ? fz.b_cur[i] : 0.0;
        }
        const int fresh = (ctrl->d_fresh && !ctrl->info && !ctrl->cov_miss) ? 1 : 0;
        const int bad = (ctrl->info || ctrl->cov_miss) ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
          ch.rec_i[cand * MC_REC_I + 0] = T0;
          ch.rec_i[cand * MC_REC_I + 1] = ctrl->l;
          ch.rec_i[cand * MC_REC_I + 2] = ctrl->sse_valid;
          ch.rec_i[cand * MC_REC_I + 3] = 1;
          ch.rec_d[cand * MC_REC_D + 0] = ctrl->coef0;
          ch.rec_d[cand * MC_REC_D + 1] = ctrl->sse_dot;
          ch.rec_d[cand * MC_REC_D + 2] = ctrl->sse_nrm;
        }
        if (bad) {  // (the record is not to be trusted: the host redoes this candidate)
          if (tid == 0) {
            ch.rec_i[cand * MC_REC_I + 3] = 0;
            st->finished = 2;
            st->why = 2;
          }
          return;
        }
        if (cand + 1 >= st->ncand) {
          if (tid == 0) {
            st->cand = cand + 1;
            st->finished = 1;
          }
          return;
        }
        const int Tn = ch.seq[cand + 1];
        for (int i = tid; i < Tn; i += 512) nd.cm_hist[i] = 0;
        __syncthreads();
        if (tid == 0) {
          st->cand = cand + 1;
          st->prev_fresh = fresh;
          st->prev_T0 = T0;
          ctrl->done = 0;
          ctrl->l = 0;
          ctrl->T0 = Tn;
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
          ctrl->serial = ctrl->serial + 1;
          if (!fresh) st->need_d = 1;
        }
        __syncthreads();
        if (!fresh) return;  // the coefficients are newer than the scores: the next score pass first
      } else if (st->need_d) {
        return;  // (nothing to select on: the score pass of the next round comes first)
      }
      const int T0 = ctrl->T0;
      const int slot = ctrl->l + 1;
      if (slot > ch.max_iter || ctrl->l < 0) {  // out of iterations (or a state this kernel does not know): the host's
        if (tid == 0) {
          st->finished = 2;
          st->why = 1;
        }
        return;
      }
      // ---- the selection of this slot: k_topk's body with the cache lookup (repeated set, arg-max start, full search)
      // (nd1 = nd with the arg-max flags set: the first selection of a candidate one level above a predecessor that
      // ended on a repeated set -- the scores are those its last iteration confirmed A_cur on)
      const bool first = slot == 1 && st->prev_fresh && T0 == st->prev_T0 + 1;  // uniform
      mc_topk_call<EB>(ch.bd, ch.p, T0, ch.A_new, ctrl, slot, first ? &ch.nd1 : &ch.nd);
      __syncthreads();
      if (ctrl->l < 0) {  // parked: missing columns (1), a tie at the selection boundary (3), a full cache (4)
        if (tid == 0) st->parked = ctrl->cov_stall;
        return;
      }
    } else {
      __syncthreads();
      if (tid == 0) st->resume = 0;
    }
    // ---- the solve (or the record-and-stop of a repeated set) and the commit: k_cgr's body, the instance k_sel_cgr
    // launches for this level
    {
      const int T0 = ctrl->T0, slot = ctrl->l + 1, nc = (T0 + 7) / 8;
      if (T0 <= 64)
        mc_cgr_call<1, 8>(T0, nc, ch.lambda, ch.xty, ch.A_new, ch.sol, ctrl, slot, &ch.fz, ch.maxit, ch.tol);
      else if (T0 <= 128)
        mc_cgr_call<2, 16>(T0, nc, ch.lambda, ch.xty, ch.A_new, ch.sol, ctrl, slot, &ch.fz, ch.maxit, ch.tol);
      else if (T0 <= 192)
        mc_cgr_call<3, 24>(T0, nc, ch.lambda, ch.xty, ch.A_new, ch.sol, ctrl, slot, &ch.fz, ch.maxit, ch.tol);
      else
        mc_cgr_call<4, 26>(T0, nc, ch.lambda, ch.xty, ch.A_new, ch.sol, ctrl, slot, &ch.fz, ch.maxit, ch.tol);
    }
    __syncthreads();
    if (ctrl->l < 0) {  // the solve parked the fit (2: residual target missed / dependent columns): the host's Cholesky
      if (tid == 0) st->parked = ctrl->cov_stall;
      return;
    }
    if (!ctrl->done) {  // a solve was committed and the fit goes on: the next score pass
      if (tid == 0) {
        st->need_d = 1;
        st->solves += 1;
      }
      return;
    }
    // the fit has ended (a repeated set: the scores stand; or a cycle after a solve): next pass records it
  }
}

// all chains' states and control blocks into pinned memory: [McState (64 B) | FitCtrl (128 B)] per chain, then the
// sequence number (system-scope release, as k_publish)
__global__ void __launch_bounds__(64) k_mc_status(const McChain *__restrict__ chains, int nchains, unsigned char *host,
                                                  unsigned long long *seq_host, unsigned long long seq) {
  const int tid = threadIdx.x;
  for (int c = 0; c < nchains; c++) {
    const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(chains[c].state);
    const unsigned long long *c8 = reinterpret_cast<const unsigned long long *>(chains[c].nd.ctrl);
    unsigned long long *h8 = reinterpret_cast<unsigned long long *>(host + (size_t)c * 192);
    if (tid < 8) h8[tid] = s8[tid];
    if (tid >= 8 && tid < 24) h8[tid] = c8[tid - 8];
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_mc_resume(const McChain *__restrict__ chains, int chain) {
  McState *st = chains[chain].state;
  FitCtrl *ctrl = chains[chain].nd.ctrl;
  if (ctrl->cov_stall) {
    ctrl->cov_stall = 0;
    ctrl->l = -1 - ctrl->l;
  }
  st->parked = 0;
  st->resume = 1;
}

__global__ void k_mc_stop(const McChain *__restrict__ chains, int chain) {
  McState *st = chains[chain].state;
  if (!st->finished) {
    st->finished = 2;
    st->why = 3;
  }
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

// Cholesky of the 16 x 16 diagonal tile D in the registers of one wave: lane (any of the four with lane & 15 == rr) ends
// with row rr of L in Lr and the reciprocal diagonal in rinv.
__device__ __forceinline__ void bc_factor_diag(const double *__restrict__ D, int rr, double (&Lr)[16], double (&rinv)[16]) {
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
  }
}

// block column b, the tiles BELOW the diagonal: every wave factors tile (b,b) redundantly in registers and substitutes
// its panel tile (I,b), I = b + 1 + blockIdx.x (16 rows, one lane each).  Tile (b,b) is only READ here: its factor is
// stored by the extra block of k_bc_update, the next launch.  (Until round 5 the wave of I == b stored the factor in
// place in THIS launch while the other waves were still reading the unfactored tile -- a race between workgroups that
// showed up as a logistic fit at k = 260 whose first IRLS solves were off: coefficients 6e-5 away on one fresh box in two,
// tests/test_lm_gpu.py::test_logistic_beyond_register_solver.)
__global__ void __launch_bounds__(64) k_bc_panel(double *__restrict__ Gt, int mt, int b,
                                                 const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int lane = threadIdx.x, rr = lane & 15, I = b + 1 + blockIdx.x;
  double Lr[16], rinv[16];
  bc_factor_diag(Gt + tile_id(b, b) * 256, rr, Lr, rinv);
  double *T = Gt + tile_id(I, b) * 256;
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

// trailing update of step b: tile (I,J) -= L(I,b) L(J,b)^T for b < J <= I, one wave per tile, 4 fp64 MFMAs.
// The LAST block (blockIdx.x == number of trailing tiles) stores the factor of the diagonal tile (b,b) and its
// reciprocal diagonal: nothing in this launch reads that tile, and every reader of the unfactored one (k_bc_panel of
// step b) has finished.
__global__ void __launch_bounds__(64) k_bc_update(double *__restrict__ Gt, double *__restrict__ rdiag, int mt, int b,
                                                  int ntrail, const FitCtrl *__restrict__ ctrl, int slot, int gate_mode) {
  BIG_GATE(ctrl, slot, gate_mode);
  const int lane = threadIdx.x, lc = lane & 15, lq = lane >> 4;
  if ((int)blockIdx.x == ntrail) {
    double Lr[16], rinv[16];
    double *D = Gt + tile_id(b, b) * 256;
    bc_factor_diag(D, lc, Lr, rinv);
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; c++) D[tile_elem(lc, c)] = Lr[c];
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (lane == j) rdiag[b * 16 + j] = rinv[j];  // (rinv is uniform over the lanes)
    }
    return;
  }
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

// merged launches over chunk chains: scores in one chunk of at most 32768, levels within k_sel_cgr's range
bool mc_applies(int p, int kmax) { return sel_cgr_applies(p, kmax) && topk_can_fuse_need(p); }
hipError_t launch_mc_sel_cgr(const McChain *chains, int nchains, int p, hipStream_t st) {
  const int per = (p + 511) / 512;
  if (per <= 8)
    hipLaunchKernelGGL((k_mc_sel_cgr<8>), dim3(1, nchains), dim3(512), 0, st, chains);
  else if (per <= 24)
    hipLaunchKernelGGL((k_mc_sel_cgr<24>), dim3(1, nchains), dim3(512), 0, st, chains);
  else
    hipLaunchKernelGGL((k_mc_sel_cgr<64>), dim3(1, nchains), dim3(512), 0, st, chains);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_mc_status(const McChain *chains, int nchains, unsigned char *host, unsigned long long *seq_host,
                            unsigned long long seq, hipStream_t st) {
  hipLaunchKernelGGL(k_mc_status, dim3(1), dim3(64), 0, st, chains, nchains, host, seq_host, seq);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_mc_resume(const McChain *chains, int chain, hipStream_t st) {
  hipLaunchKernelGGL(k_mc_resume, dim3(1), dim3(1), 0, st, chains, chain);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_mc_stop(const McChain *chains, int chain, hipStream_t st) {
  hipLaunchKernelGGL(k_mc_stop, dim3(1), dim3(1), 0, st, chains, chain);
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
    if (mt - b - 1 > 0) {
      hipLaunchKernelGGL(k_bc_panel, dim3(mt - b - 1), dim3(64), 0, st, Gt, mt, b, ctrl, slot, gate_mode);
      LAUNCH_CHECK();
    }
    // the trailing tiles + one block that stores the factor of tile (b,b)
    const int nt = (mt - b - 1) * (mt - b) / 2;
    hipLaunchKernelGGL(k_bc_update, dim3(nt + 1), dim3(64), 0, st, Gt, rdiag, mt, b, nt, ctrl, slot, gate_mode);
    LAUNCH_CHECK();
  }
  for (int b = mt - 1; b >= 0; b--) {
    hipLaunchKernelGGL(k_bc_back, dim3(1), dim3(512), 0, st, (const double *)Gt, (const double *)rdiag, z, mt, b, m,
                       sol, info, ctrl, slot, gate_mode);
    LAUNCH_CHECK();
  }
  return hipSuccess;
}


}  // namespace bessx
