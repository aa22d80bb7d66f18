// bessx_fit.cpp -- Algorithm::fit (src/Algorithm.h:113-171) as speculatively queued, device-gated PDAS iterations: the slot
// enqueue functions of every family, the parked-fit protocol of the covariance form, publication of the result block
#include "bessx_host.h"

namespace bessx {

// The pass over X of one PDAS iteration -- K1 (LM, streaming form), K2 (logistic / Poisson: two accumulators), K3 (Cox
// score) -- for slot `slot` of the fit on row set rs.  A chunk chain whose owner runs shared passes hands its vector set
// to the owner's next multi-chain launch (bessx_kchunks.cpp: SharedPass; timed there); everybody else launches on its own
// stream, timed by an event pair when the statistics are on.  kind: 0 LM, 1 GLM, 2 Cox.
static int score_pass(bessx_session *s, int kind, int rs, int slot, std::vector<std::pair<size_t, bool>> &k1_pairs) {
  if (rs == 0 && shared_pass_applies(s)) {
    if (s->timing) k1_pairs.push_back({(size_t)-1, false});
    if (kind == 2 && s->cox.one_pass)
      return shared_pass_submit(s, nullptr, nullptr, s->part_rs[rs], nullptr, &s->cox, s->ctrl, slot);
    if (kind != 2)
      return shared_pass_submit(s, s->r_rs[rs], kind == 1 ? s->h_rs[rs] : nullptr, s->part_rs[rs],
                                kind == 1 ? s->part2_rs[rs] : nullptr, nullptr, s->ctrl, slot);
  }
  hipEvent_t ea = nullptr, eb = nullptr;
  if (int rc = k1_begin(s, &ea, &eb)) return rc;
  hipError_t e;
  if (kind == 2)
    e = launch_cox_score_pass(s->X, s->ld, s->p, s->U, s->nrb, s->cox, s->part_rs[rs], s->part2_rs[rs], s->ctrl, slot, s->st);
  else
    e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], kind == 1 ? s->h_rs[rs] : nullptr, s->part_rs[rs],
                   kind == 1 ? s->part2_rs[rs] : nullptr, s->ctrl, slot, s->st);
  if (s->timing && e == hipSuccess) {
    e = hipEventRecord(eb, s->st);
    k1_pairs.push_back({s->ev_used - 2, false});
  }
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("score pass: ") + hipGetErrorString(e));
  return 0;
}

// --------------------------------------------------------------------------------------------
// Algorithm::fit (src/Algorithm.h:113-171), LM: GroupPdasLm::get_A / primary_model_fit (:1097-1135)
// --------------------------------------------------------------------------------------------
// part: 0 = the whole PDAS iteration; 1 = its LIGHT part -- score pass, scores, selection, the comparison with the
// previous active set and a commit that only records a REPEATED set (the normal end of a fit: 2 launches behind the
// selection instead of the 9 of Gram, solve, commit and residual, which would all fall through); 2 = the rest of an
// iteration whose light part found a NEW set (the host sees that the slot was not committed and queues it).
int enqueue_lm_slot(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                           std::vector<std::pair<size_t, bool>> &k1_pairs, int part) {
  const int mt = (T0 + 1 + 15) / 16, mp = mt * 16;
  const int ntiles = mt * (mt + 1) / 2;
  const GramTask *tasks_full = nullptr;
  int ntask = 0, rps, nslab;
  if (int rc = gram_tasks_for(s, mt, &tasks_full, &ntask)) return rc;
  gram_geometry(s, ntask, &rps, &nslab, ntiles, mt > 16);  // (the cached LM Gram keeps k_gram: gates 3 / 4)
  if ((size_t)nslab * ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
  hipError_t e = hipSuccess;
  if (part != 2) {
    if (!skip_k1) {
      // skip_k1: the partial sums of this row set were computed from exactly the coefficients this fit
      // starts from (the previous fit ended on a repeated active set) -- get_A would recompute them bit for bit.
      if (int rc = score_pass(s, 0, rs, slot, k1_pairs)) return rc;
    } else if (s->timing) {
      k1_pairs.push_back({(size_t)-1, false});
    }
    if (e == hipSuccess)
      e = launch_score(s->part_rs[rs], nullptr, s->nrb, s->p, s->beta_dense, s->xtx[rs], (double)s->n_train[rs],
                       lambda, 0, s->always, s->bd, s->ctrl, slot, s->st);
    if (e == hipSuccess) e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
    if (e == hipSuccess) e = launch_gram_cols(s->A_new, T0, mp, 0, 0, s->gcols, s->ctrl, slot, s->A_cur, 1, s->st);
  }
  if (part == 1) {
    // (wait_chain = 1: k_commit records a repeated set and otherwise stands back -- irls_done is never set for LM)
    if (e == hipSuccess)
      e = launch_commit(s->ctrl, slot, T0, s->A_new, s->sol, 0, 1, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                        s->hist_beta, s->hist_coef0, s->hist_stride, s->st, s->inA);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_lm_slot (light): ") + hipGetErrorString(e));
    return 0;
  }
  if (e == hipSuccess && mt > 16) {
    // beyond the register-resident solver: whole Gram every time, blocked Cholesky in global memory
    e = launch_gram(s->X, s->aux, s->ld, s->gcols, s->mask[rs], rps, tasks_full, ntask, nslab, s->gpart, ntiles, s->Gt,
                    s->ctrl, slot, 0, s->st, 0);
    if (e == hipSuccess)
      e = launch_chol_big(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->rdiag, s->zbig,
                          s->ctrl, slot, 0, s->st);
  } else if (e == hipSuccess) {
    int rps_i, nslab_i;
    const int ntask_i = s->gtask_inc_cnt[mt];
    gram_geometry(s, ntask_i * 3, &rps_i, &nslab_i);  // a third of the usual wave count: the extra row is cheap
    bessx_session::GramCache &gc = s->gcache[rs];
    e = launch_gram_lm_cached(s->X, s->aux, s->ld, s->gcols, s->mask[rs], s->A_new, T0, mt,
                              tasks_full, ntask, rps, nslab, s->gtasks + s->gtask_inc_off[mt],
                              ntask_i, rps_i, nslab_i, s->gpart, s->Gt, s->Rt, s->gsrc, gc.g0, gc.g1, gc.A, gc.meta,
                              s->ctrl, slot, s->st);
    const CholFuse fbz = chol_fallback_only(s);
    if (e == hipSuccess)
      e = launch_chol(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->ctrl, slot, 0, s->st,
                      &fbz);
    if (e == hipSuccess)  // (exactly dependent active columns: the pivoted solve; falls through otherwise)
      e = launch_sym_fallback(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->ctrl, slot, s->st,
                              &fbz);
  }
  if (e == hipSuccess)
    e = launch_commit(s->ctrl, slot, T0, s->A_new, s->sol, 0, 0, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                      s->hist_beta, s->hist_coef0, s->hist_stride, s->st, s->inA);
  if (e == hipSuccess)
    e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->r_rs[rs],
                        s->sse, s->st);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_lm_slot: ") + hipGetErrorString(e));
  return 0;
}


// --------------------------------------------------------------------------------------------
// GLM families: GroupPdasLogistic / GroupPdasPoisson get_A + primary_model_fit
// (src/Algorithm.h:1148-1263, :1273-1367).  One PDAS iteration = score pass (two accumulators) -> top-k
// -> IRLS on [1, X_A] as a chain of (prep, check, weighted Gram, Cholesky) steps that stops itself on the
// device -> commit -> gradient / loss pass for the new coefficients.
// --------------------------------------------------------------------------------------------
// --------------------------------------------------------------------------------------------
// LM slot in covariance-update mode: the same PDAS iteration, with X^T r taken from the cached Gram columns.
// --------------------------------------------------------------------------------------------
// Form the Gram columns of the fill list, 2 groups of 32 columns per launch pair (the slab partials of a launch
// share one work space).  parked = 1: for a parked fit, 0: start of a fit.  The panel kernel is the one kernel
// of this mode that reads X: its launches are timed like the streaming score pass (k1_*).
// Two groups in one launch (a fill of more than 32 columns: cold starts, the chunks of a sharded path): the pair panel
// kernel forms both in ONE pass over X (1.36 ms against 2 x 0.76 ms); single groups keep the default.
int panel_variant_for(const bessx_session *s, int ng) {
  // 3 = k_cov_panel_lds2 (one 32-column group per block), 4 = k_cov_panel_pair (two groups per pass),
  // 5 = k_cov_panel_dp (round 5: one workgroup per compute unit, two LDS tiles, software-pipelined; one or two groups)
  if (s->panel_variant == 5) return 5;
  return ng == 2 ? 4 : 3;
}

// gfirst / compact: a cooperative prefill (bessx_session_cov_prefill_*) forms only SOME groups of the list here and
// fills the slot-indexed Gram GS once every group is in (its own and the ones imported from the other ranks)
// slot_map: the map the reduction and the compaction place the new columns by (staged fills of chunk chains: the
// writer's map; else the cache's own)
int enqueue_cov_fill(bessx_session *s, int rs, int ngroups, int parked, const FitCtrl *gate,
                            int gfirst, bool compact, const int *slot_map) {
  bessx_session::CovCache &cv = s->cov[rs];
  const FitCtrl *gc = gate ? gate : s->ctrl;  // whose cov_stall / cov_nfill the launches look at
  const int *smap = slot_map ? slot_map : cv.slot_of;
  for (int g0 = gfirst; g0 < gfirst + ngroups; g0 += COV_SLOT_GROUPS) {
    const int ng = std::min(COV_SLOT_GROUPS, gfirst + ngroups - g0);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (int rc = k1_begin(s, &ea, &eb)) return rc;
    hipError_t e = hipSuccess;
    if (s->cv_shared) {
      // one unmasked pass over the fold-major copy serves every row set: the columns enter ALL caches (same slots)
      const int nsl_all = s->K * s->cvp_nsl;
      e = launch_cov_panel(s->Xp, s->zp, s->ldp, s->p, nullptr, s->cov_fcols, g0, ng, s->cvp_rps, nsl_all, s->cvp_part,
                           gc, parked, s->st, panel_variant_for(s, ng));
      if (s->timing && e == hipSuccess) {
        e = hipEventRecord(eb, s->st);
        s->cov_timed.push_back({s->ev_used - 2, g0});
      }
      if (s->K + 1 <= 9 && e == hipSuccess) {
        // every row set in one reduce and one compact launch (they share the slot map and the list)
        CovRowSets rsets = {};
        rsets.nr = s->K + 1;
        for (int r = 0; r <= s->K; r++) {
          rsets.G[r] = s->cov[r].G;
          rsets.GS[r] = s->cov[r].GS;
          rsets.xtx[r] = s->xtx[r];
          rsets.ex_lo[r] = r == 0 ? 0 : (r - 1) * s->cvp_nsl;  // fold r-1's own rows out
          rsets.ex_hi[r] = r == 0 ? 0 : r * s->cvp_nsl;
        }
        e = launch_cov_reduce_compact_sets(s->cvp_part, s->p, s->cov_fcols, s->cov[0].slot_of, s->cov[0].meta, rsets, g0, ng,
                                           nsl_all, s->cov_cs, gc, parked, s->st);
      }
      for (int r = 0; r <= s->K && e == hipSuccess && s->K + 1 > 9; r++) {
        bessx_session::CovCache &cr = s->cov[r];
        const int lo = r == 0 ? 0 : (r - 1) * s->cvp_nsl, hi = r == 0 ? 0 : r * s->cvp_nsl;  // fold r-1's own rows out
        e = launch_cov_reduce(s->cvp_part, s->p, s->cov_fcols, cr.slot_of, cr.G, g0, ng, nsl_all, gc, parked, s->st,
                              lo, hi);
        if (e == hipSuccess)
          e = launch_cov_compact(cr.G, s->p, cr.slot_of, s->cov_fcols, g0, ng, cr.GS, s->cov_cs, gc, parked, s->st,
                                 s->xtx[r], cr.meta);
      }
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov panel (shared): ") + hipGetErrorString(e));
      continue;
    }
    e = launch_cov_panel(s->X, s->aux, s->ld, s->p, s->mask[rs], s->cov_fcols, g0, ng, s->cov_rps,
                         s->cov_nslab, s->cov_part, gc, parked, s->st, panel_variant_for(s, ng));
    if (s->timing && e == hipSuccess) {
      e = hipEventRecord(eb, s->st);
      s->cov_timed.push_back({s->ev_used - 2, g0});
    }
    if (e == hipSuccess)
      e = launch_cov_reduce(s->cov_part, s->p, s->cov_fcols, smap, cv.G, g0, ng, s->cov_nslab, gc, parked,
                            s->st);
    if (e == hipSuccess && compact)  // entries between cached columns, by slot: what the solve gathers from
      e = launch_cov_compact(cv.G, s->p, smap, s->cov_fcols, g0, ng, cv.GS, s->cov_cs, gc, parked, s->st, s->xtx[rs],
                             cv.meta);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov panel: ") + hipGetErrorString(e));
  }
  return 0;
}

// the capacity the device-side lookup (cov_need_body: "count + len + 32 > C -> start the cache over") is told: a fill
// can add up to cov_spec speculative columns beyond the requested ones
int cov_C_dev(const bessx_session *s) { return s->cov_C - (s->cov_spec - COV_R); }

// Fusions of the small kernels around a slot (test hook fuse=0 turns them off):
//  * pub: the slot closes a batch -- its solve kernel publishes the result block itself (*pub_fused = true) instead
//    of a k_publish launch behind it;
//  * the selection kernel records a repeated active set itself (TopkNeed.commit_on) and opens a chained fit
//    (TopkNeed.cont_on) instead of a k_fit_continue launch in front of it.
struct SlotFuse {
  const PubArgs *pub = nullptr;
  bool pub_fused = false;
  bool pub_snapshot = false;           // the tail only snapshots the block (chained batch): publication deferred
  const PubArgs *attach = nullptr;     // deferred publication of the parent fit, for the first selection kernel
  bool attached = false;
  int cont_serial = 0, cont_parent = 0;
  bool cont = false, cont_fused = false;
};

// arguments of the solve of a covariance-form slot (k_cg / k_cgr / k_chol with the gather and the commit fused in)
CholFuse cov_fuse_args(bessx_session *s, int rs, int T0, bool force_chol, SlotFuse *sf) {
  bessx_session::CovCache &cv = s->cov[rs];
  CholFuse fz = {cv.G,          cv.slot_of, s->p,         T0,           s->ctrl,        s->A_cur, s->b_cur,
                 s->beta_dense, s->hist,    s->hist_beta, s->hist_coef0, s->hist_stride, s->inA,   s->yy_h[rs],
                 s->part_rs[rs], cv.GS, s->cov_cs,
                 cv.zero, PubArgs{}, s->fb_work, cv.meta + 4};
  // the last kernel of the batch publishes: only when nothing follows the solve in this slot (all rows, k_cg)
  if (sf && sf->pub && s->fuse && s->cov_cg && !force_chol && rs == 0) {
    fz.pub = *sf->pub;
    if (sf->pub_snapshot) fz.pub.on = 2;
    sf->pub_fused = true;
  }
  return fz;
}

// work space of the large-system conjugate gradients (bessx_cgbig.hip), on first use: 134 MB at 4096 unknowns
int cgb_reserve(bessx_session *s) {
  bessx_session *owner = s->parent ? s->parent : s;
  if (!owner->cgb_work) {
    const int kcap = std::min(owner->cap, CGB_MAX_K);
    if (dmalloc(&owner->cgb_work, cgb_work_doubles(kcap)) != hipSuccess) {
      (void)hipGetLastError();
      owner->cgb_work = nullptr;
      return 1;  // (no memory: the blocked Cholesky does it)
    }
    owner->cgb_cap = kcap;
  }
  s->cgb_work = owner->cgb_work;
  s->cgb_cap = owner->cgb_cap;
  return 0;
}

// solve + commit + residual of a slot whose active columns are all cached
int enqueue_cov_tail(bessx_session *s, int slot, int T0, double lambda, int rs, bool force_chol,
                            SlotFuse *sf) {
  const int mt = (T0 + 1 + 15) / 16;
  bessx_session::CovCache &cv = s->cov[rs];
  hipError_t e = hipSuccess;
  bool used_cgb = false;
  if (mt > 16 && s->cov_cg && !force_chol && T0 <= CGB_MAX_K && cgb_reserve(s) == 0) {
    // beyond the register-resident solvers: conjugate gradients over the whole chip, one launch per step
    // (bessx_cgbig.hip); an iterate whose true residual misses the target parks the fit (cov_stall = 2) and the
    // blocked Cholesky below finishes the slot (force_chol)
    e = launch_cg_big(cv.G, s->p, cv.slot_of, cv.meta, s->A_new, T0, lambda, s->xty[rs], s->beta_dense, s->cgb_work,
                      s->cgb_cap, s->sol, s->ctrl, slot, s->cgb_guess, s->cg_tol, s->yy_h[rs], s->st);
    used_cgb = e == hipSuccess;
    if (e == hipSuccess)
      e = launch_commit(s->ctrl, slot, T0, s->A_new, s->sol, 0, 0, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                        s->hist_beta, s->hist_coef0, s->hist_stride, s->st, s->inA);
  } else if (mt > 16) {
    e = launch_cov_gram(cv.G, s->p, cv.slot_of, s->A_new, T0, mt, s->Gt, cv.meta, s->ctrl, slot, s->st);
    if (e == hipSuccess)
      e = launch_chol_big(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->rdiag, s->zbig,
                          s->ctrl, slot, 0, s->st);
    if (e == hipSuccess)
      e = launch_commit(s->ctrl, slot, T0, s->A_new, s->sol, 0, 0, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                        s->hist_beta, s->hist_coef0, s->hist_stride, s->st, s->inA);
  } else {
    // one launch: Gram gathered from the cache while loading, the solve, then the commit.  The solve is conjugate
    // gradients warm-started from the previous coefficients (k_cg); if its true residual does not reach 1e-13 it
    // parks the fit (cov_stall = 2) and the Cholesky kernel is issued for the slot (force_chol).
    CholFuse fz = cov_fuse_args(s, rs, T0, force_chol, sf);
    if (s->cov_cg && !force_chol)
      e = launch_cg(T0, (T0 + 15) / 16, lambda, s->xty[rs], s->A_new, s->sol, s->ctrl, slot, &fz, 64, s->st, s->cg_tol,
                    s->cg_by_rows);
    else {
      e = launch_chol(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->ctrl, slot, 0, s->st,
                      &fz);
      if (e == hipSuccess)  // (its pivot test failed: pivoted solve + the commit k_chol skipped)
        e = launch_sym_fallback(s->Gt, T0, mt, lambda, 0, s->xty[rs], s->A_new, s->sol, &s->ctrl->info, s->ctrl, slot,
                                s->st, &fz);
    }
  }
  // CV row sets need the sums of squares over the test rows too: one pass over the active columns for the final
  // coefficients (runs iff the fit ended here).  On all rows the loss comes from the solved system (k_chol).
  // (the large-system conjugate gradients hand the loss terms over like the small systems' solve)
  if (e == hipSuccess && (rs != 0 || (mt > 16 && !used_cgb) || !s->cov_cg))
    e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->r_rs[rs],
                        s->sse, s->st, 1);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_cov_tail: ") + hipGetErrorString(e));
  return 0;
}

bool cov_speculates(const bessx_session *s) { return topk_supported(s->p, s->cov_spec) && s->p >= 2 * s->cov_spec; }

// skip_d: d of exactly the starting coefficients is in memory (previous fit of the chain); scores_ok: so are the
// sacrifice scores (same lambda), nothing to recompute before the selection.
// grow1: ... and the previous fit (same row set, the last thing the device ran) had sparsity level T0 - 1 and ended
// with A_cur = max_k of these very scores: the first selection is A_cur plus one arg-max (k_topk).
int enqueue_lm_slot_cov(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_d,
                               bool scores_ok, bool grow1, SlotFuse *sf) {
  bessx_session::CovCache &cv = s->cov[rs];
  hipError_t e = hipSuccess;
  if (skip_d && scores_ok) {
    // the last k_cov_d of the previous fit left bd for these coefficients and this lambda
  } else if (!skip_d)  // d and the sacrifice scores in one kernel
  {
    e = launch_cov_d(cv.G, s->p, cv.slot_of, s->xty[rs], s->A_cur, s->b_cur, s->part_rs[rs], s->beta_dense, s->xtx[rs],
                     (double)s->n_train[rs], lambda, s->always, s->bd, s->inA, s->cov_bmm, s->ctrl, slot, s->st);
    s->bmm_owner = rs;
  }
  else {  // d of exactly these coefficients is in memory (previous fit of the chain); lambda may have changed
    e = launch_score(s->part_rs[rs], nullptr, 1, s->p, s->beta_dense, s->xtx[rs], (double)s->n_train[rs], lambda, 0,
                     s->always, s->bd, s->ctrl, slot, s->st);
    s->bmm_owner = -1;  // the block maxima no longer belong to the scores in bd
  }
  // top-k, then the repeated-set test + cache lookup (parks the fit when a column of A_new is not cached) -- in the
  // same launch when the scores fit one chunk of the selection kernel
  if (e == hipSuccess && topk_can_fuse_need(s->p)) {
    TopkNeed nd = {cov_speculates(s) ? s->bd : nullptr, s->bd2, s->p, cov_C_dev(s), cv.slot_of, cv.meta, s->cov_fcols,
                   s->ctrl, s->A_cur, s->cov_bmm, (s->p + 31) / 32, s->inA, (slot == 1 && grow1) ? 1 : 0,
                   (slot == 1 && grow1 && s->bmm_owner == rs) ? 1 : 0};
    nd.cm_A_cur = s->A_cur;
    nd.cm_b_cur = s->b_cur;
    nd.cm_beta_dense = s->beta_dense;
    nd.cm_hist = s->hist;
    nd.cm_hist_beta = s->hist_beta;
    nd.cm_hist_coef0 = s->hist_coef0;
    nd.cm_hist_stride = s->hist_stride;
    nd.cm_inA = s->inA;
    nd.commit_on = (s->fuse && (T0 + 1 + 15) / 16 <= 16) ? 1 : 0;  // (beyond: launch_commit does it, unfused)
    nd.no_restart = s->cov_no_restart ? 1 : 0;
    if (sf && sf->cont && s->fuse && slot == 1 && skip_d && scores_ok) {
      // nothing runs before the selection in this slot: it opens the chained fit itself
      nd.cont_on = 1;
      nd.cont_serial = sf->cont_serial;
      nd.cont_parent = sf->cont_parent;
      sf->cont_fused = true;
      if (sf->attach) {  // ... and its second workgroup publishes the parent's snapshot meanwhile
        nd.pub = *sf->attach;
        sf->attached = true;
      }
    }
    if (sf && sf->pub && sf->pub_snapshot && s->fuse && nd.commit_on && s->cov_cg && rs == 0) {
      nd.snap = *sf->pub;  // last slot of a chained batch: a repeated set is recorded AND snapshotted here
      nd.snap.on = 2;
    }
    if (s->fuse_sel && s->fuse && s->cov_cg && s->cg_by_rows && sel_cgr_applies(s->p, T0)) {  // (k_sel_cgr solves by rows)
      // selection and solve of this slot in ONE launch (k_sel_cgr): same phases, same control-block protocol
      CholFuse fz = cov_fuse_args(s, rs, T0, false, sf);
      e = launch_sel_cgr(s->bd, s->p, T0, s->A_new, s->ctrl, slot, &nd, lambda, s->xty[rs], s->sol, &fz, 64, s->st,
                         s->cg_tol);
      if (e == hipSuccess && rs != 0)  // CV row sets: sums of squares over the test rows for the final coefficients
        e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->r_rs[rs],
                            s->sse, s->st, 1);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_lm_slot_cov (fused): ") + hipGetErrorString(e));
      return 0;
    }
    e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, &nd);
  } else if (e == hipSuccess) {
    e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
    if (e == hipSuccess)
      e = launch_cov_need(s->A_new, T0, cov_speculates(s) ? s->bd : nullptr, s->bd2, s->p, cv.slot_of, cv.meta, cov_C_dev(s),
                          s->cov_fcols, s->ctrl, slot, s->A_cur, s->st, s->cov_no_restart ? 1 : 0);
  }
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_lm_slot_cov: ") + hipGetErrorString(e));
  return enqueue_cov_tail(s, slot, T0, lambda, rs, false, sf);
}

// A parked fit (hc = the control block just read back): fill list, Gram columns, wake-up, rest of the slot.
int cov_unpark(bessx_session *s, const FitCtrl *hc, int T0, double lambda, int rs, int *next_slot) {
  bessx_session::CovCache &cv = s->cov[rs];
  const int stalled = -1 - hc->l + 1, nm = hc->cov_nmiss;
  if (hc->cov_stall == 2) {
    // the conjugate-gradient solve did not reach its residual target: Cholesky for this slot
    s->cov_cg_fallbacks++;
    if ((T0 + 1 + 15) / 16 > 16) s->cgb_guess = 64;  // (large system: perhaps only short of step launches)
    HIPX(launch_cov_resume(s->ctrl, s->st));
    if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs, true)) return rc;
    *next_slot = stalled + 1;
    return 0;
  }
  if (hc->cov_stall == 3) {
    // equal scores at the selection boundary (duplicated columns, 0/1 designs): the fused selection parked the fit; the
    // slot is redone unfused -- plain selection, the exact tie rule (k_topk_ties: the moves of the reference's
    // std::nth_element, src/utilities.cpp:179-188), cache lookup, then the solve
    s->cov_tie_rescues++;
    HIPX(launch_cov_resume(s->ctrl, s->st));
    hipError_t e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, stalled, s->st, nullptr, nullptr, &s->tie);
    if (e == hipSuccess)
      e = launch_cov_need(s->A_new, T0, cov_speculates(s) ? s->bd : nullptr, s->bd2, s->p, cv.slot_of, cv.meta,
                          cov_C_dev(s), s->cov_fcols, s->ctrl, stalled, s->A_cur, s->st, s->cov_no_restart ? 1 : 0);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov_unpark (tie): ") + hipGetErrorString(e));
    if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs)) return rc;
    *next_slot = stalled + 1;
    return 0;
  }
  const bool spec = cov_speculates(s);
  hipError_t e = hipSuccess;
  // a chunk chain (bessx_kchunks.cpp) fills the shared cache only while every other chain stands still; if another chain
  // filled meanwhile the look-up is simply redone (its columns may be there now)
  struct FillGuard {
    bessx_session *c = nullptr;
    bool filled = false;
    ~FillGuard() {
      if (c) kchains_fill_end(c, filled);
    }
  } guard;
  // staged fills (the chains of this round do not stand still for each other): the launches of the fill work on the
  // writer's slot map, the readers' map gets the entries with the last launch
  const bool staged = s->kch_owner && kchains_staged(s) && cv.slot_w != nullptr;
  int *smap = staged ? cv.slot_w : cv.slot_of;
  if (s->kch_owner) {
    kchains_log(s, "parked: asks for the cache", T0, nm);
    const int waited = kchains_fill_begin(s);
    if (waited < 0) return fail(BESSX_ERR_HIP, "chunk chains: the fill rendezvous was abandoned");
    guard.c = s;
    kchains_log(s, waited ? "has the cache: looks again" : "has the cache: fills", T0, nm);
    if (waited > 0) {
      HIPX(launch_cov_resume(s->ctrl, s->st));
      e = launch_cov_need(s->A_new, T0, spec ? s->bd : nullptr, s->bd2, s->p, cv.slot_of, cv.meta, cov_C_dev(s),
                          s->cov_fcols, s->ctrl, stalled, s->A_cur, s->st, 1);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov_unpark (chunk chain): ") + hipGetErrorString(e));
      if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs)) return rc;
      *next_slot = stalled + 1;
      return 0;
    }
  }
  if (s->fill_hook && rs == 0 && !s->parent && spec) {
    // shared wide fill (bessx_session_set_fill_hook): the missing columns, then the uncached columns the scores of this
    // iteration rank highest (bd2: cached and missing ones already at -1), fill_hook_width columns in all
    const int p = s->p;
    std::vector<int> miss((size_t)nm), slot_h((size_t)p);
    std::vector<double> sc((size_t)p);
    HIPX(hipMemcpyAsync(miss.data(), s->cov_fcols, (size_t)nm * sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPX(hipMemcpyAsync(sc.data(), s->bd2, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, s->st));
    HIPX(hipMemcpyAsync(slot_h.data(), cv.slot_of, (size_t)p * sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPX(hipStreamSynchronize(s->st));
    std::vector<int> order;
    order.reserve((size_t)p);
    for (int j = 0; j < p; j++)
      if (slot_h[j] < 0 && sc[j] >= 0.0) order.push_back(j);
    int total = std::max(s->fill_hook_width, (nm + COV_R - 1) / COV_R * COV_R);
    if (nm + (int)order.size() < total) total = (nm + (int)order.size()) / COV_R * COV_R;
    if (total >= nm && total >= COV_R && total <= s->capA) {
      const int extra = total - nm;
      std::partial_sort(order.begin(), order.begin() + extra, order.end(), [&](int a, int b) {
        return sc[a] > sc[b] || (sc[a] == sc[b] && a < b);  // (ties by column: the same list on every rank)
      });
      std::vector<int> list(miss);
      list.insert(list.end(), order.begin(), order.begin() + extra);
      if (int rc = prefill_begin(s, list.data(), total, 2)) return rc;
      if (s->fill_hook(s->fill_hook_user, total / COV_R) != 0)
        return fail(BESSX_ERR_ARG, "the fill hook reported a failure");
      if (s->prefill_cols != 0) return fail(BESSX_ERR_ARG, "the fill hook returned without bessx_session_cov_prefill_end");
      s->shared_wide_fills++;
      HIPX(launch_cov_resume(s->ctrl, s->st));
      if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs)) return rc;
      *next_slot = stalled + 1;
      return 0;
    }
    // (fewer uncached columns than one group: the private fill below)
  }
  // a cache that holds every column is never started over: the extras are taken from the 64 best uncached columns, so
  // that the list is full up to its multiple of 32 (k_cov_fill_list, spec = 2)
  const bool wide = spec && s->cov_C >= (s->p + 31) / 32 * 32 + COV_R && topk_supported(s->p, 2 * COV_R) && s->p >= 4 * COV_R;
  const int pool = wide ? 2 * COV_R : s->cov_spec;
  if (spec) e = launch_topk(s->bd2, s->p, pool, s->cov_extras, s->cand, nullptr, 0, s->st);
  if (e == hipSuccess)
    e = launch_cov_fill_list(s->cov_fcols, s->cov_extras, s->bd2, smap, cv.meta, s->ctrl, 1, s->st, s->cov_spec,
                             spec ? (wide ? 2 : 1) : 0, s->cov_spec_min);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov_unpark: ") + hipGetErrorString(e));
  // upper bound of the list length (the device drops speculative columns that turn out to be cached already)
  const int room = spec ? std::min(((nm + s->cov_spec_min + s->cov_spec - 1) / s->cov_spec) * s->cov_spec - nm, pool) : 0;
  const int ngroups = (nm + room + COV_R - 1) / COV_R;
  if (test_hook("panel_log")) std::fprintf(stderr, "[bessx] fill: level %d, %d missing, list of up to %d -> %d group(s)%s\n", T0, nm, nm + room, ngroups, s->kch_owner ? " (chunk chain)" : "");
  // staged fills run on the owner's fill stream where there is one (it leaves some compute units to the other chains'
  // kernels): the list first, on this chain's stream
  hipStream_t own_st = s->st;
  hipStream_t fill_st = staged ? s->kch_owner->kch_fill_st : nullptr;
  if (fill_st && !s->kch_ev && hipEventCreateWithFlags(&s->kch_ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    s->kch_ev = nullptr;
    fill_st = nullptr;
  }
  if (fill_st) {
    HIPX(hipEventRecord(s->kch_ev, own_st));
    HIPX(hipStreamWaitEvent(fill_st, s->kch_ev, 0));
    s->st = fill_st;
  }
  int frc = enqueue_cov_fill(s, rs, ngroups, 1, nullptr, 0, true, smap);
  hipError_t fe = hipSuccess;
  if (frc == 0 && staged) fe = launch_cov_publish_slots(s->cov_fcols, cv.slot_w, cv.slot_of, s->ctrl, s->st);
  s->st = own_st;
  if (frc) return frc;
  HIPX(fe);
  if (s->kch_owner) {
    HIPX(hipStreamSynchronize(fill_st ? fill_st : s->st));  // the new columns are in memory before any other chain moves again
    s->kch_owner->kch_chunk_fills++;    // (under the rendezvous: one writer)
    guard.filled = true;
    kchains_log(s, "fill done", ngroups, nm);
  }
  HIPX(launch_cov_resume(s->ctrl, s->st));
  if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs)) return rc;
  *next_slot = stalled + 1;
  return 0;
}

int glm_geometry(bessx_session *s, int T0, int *mt, int *mp, int *ntask, int *ntiles, int *rps, int *nslab) {
  *mt = (T0 + 2 + 15) / 16;  // intercept + T0 columns + the working response
  *mp = *mt * 16;
  const GramTask *tk = nullptr;
  if (int rc = gram_tasks_for(s, *mt, &tk, ntask)) return rc;
  *ntiles = *mt * (*mt + 1) / 2;
  gram_geometry(s, *ntask, rps, nslab, *ntiles);
  if ((size_t)*nslab * *ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
  return 0;
}

int enqueue_glm_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs) {
  const int fam = s->model_type;
  int mt, mp, ntask, ntiles, rps, nslab;
  if (int rc = glm_geometry(s, T0, &mt, &mp, &ntask, &ntiles, &rps, &nslab)) return rc;
  hipError_t e = hipSuccess;
  if (!skip_k1) {
    if (int rc = score_pass(s, 1, rs, slot, k1_pairs)) return rc;
  } else if (s->timing) {
    k1_pairs.push_back({(size_t)-1, false});
  }
  if (e == hipSuccess)
    e = launch_score(s->part_rs[rs], s->part2_rs[rs], s->nrb, s->p, s->beta_dense, nullptr, (double)s->n_train[rs],
                     lambda, 1, s->always, s->bd, s->ctrl, slot, s->st);
  if (e == hipSuccess) e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
  // a repeated active set reproduces the logistic fit (cold start); Poisson restarts from the new intercept
  if (e == hipSuccess)
    e = launch_gram_cols(s->A_new, T0, mp, 1, 1, s->gcols, s->ctrl, slot, s->A_cur, fam == 2 ? 1 : 0, s->st);
  if (e == hipSuccess) e = launch_glm_irls_begin(s->ctrl, slot, fam, T0 + 1, s->bcur, s->bprev, s->st);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_glm_head: ") + hipGetErrorString(e));
  return 0;
}

int enqueue_glm_irls_step(bessx_session *s, int slot, int t, int T0, double lambda, int rs) {
  const int fam = s->model_type;
  int mt, mp, ntask, ntiles, rps, nslab;
  if (int rc = glm_geometry(s, T0, &mt, &mp, &ntask, &ntiles, &rps, &nslab)) return rc;
  double *z = s->aux + 2 * s->ld;
  if (s->irls_fuse && irls_gram_applies(mt)) {
    // two launches less and one pass over the active columns instead of two: linear predictor, weights and working
    // response are formed inside the Gram kernel (k_irls_gram), the convergence test at the head of the solve
    const int rows = irls_gram_slab_rows(mt, s->ld);
    const int ns = (int)((s->ld + rows - 1) / rows);
    if ((size_t)ns * ntiles * 256 <= s->gpart_elems && (size_t)ns <= s->llpart_cap) {
      hipError_t e = launch_irls_gram(fam, s->X, s->aux, s->ld, s->n, s->gcols, s->y, s->w, s->mask[rs], ns, mt,
                                      s->gpart, ntiles, s->ctrl, slot, t, T0, s->bcur, s->llpart, s->st, s->irls_wfloor);
      if (e == hipSuccess) e = launch_gram_reduce(s->gpart, ns, ntiles, s->Gt, s->ctrl, slot, 1, s->st);
      const IrlsChk ck = {1, s->ctrl, t, fam, s->llpart, ns, T0 + 1, s->bcur, s->bprev};
      const CholFuse fbz = chol_fallback_only(s);
      if (e == hipSuccess)
        e = launch_chol(s->Gt, T0 + 1, mt, 2.0 * lambda, 1, nullptr, nullptr, s->bcur, &s->ctrl->info, s->ctrl, slot, 1,
                        s->st, &fbz, &ck);
      if (e == hipSuccess && s->glm_fallback)
        e = launch_sym_fallback(s->Gt, T0 + 1, mt, 2.0 * lambda, 1, nullptr, nullptr, s->bcur, &s->ctrl->info, s->ctrl, slot,
                                s->st, &fbz);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_glm_irls_step: ") + hipGetErrorString(e));
      return 0;
    }
  }
  hipError_t e = launch_glm_irls_prep(fam, s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->ctrl, slot, t, s->A_new, T0,
                                      s->bcur, s->Wv, z, s->llpart, s->st, s->irls_wfloor);
  if (e == hipSuccess)
    e = launch_glm_irls_check(s->ctrl, slot, t, fam, s->llpart, s->n_sse_blk, T0 + 1, s->bcur, s->bprev, s->st);
  const GramTask *tk = nullptr;
  int ntk = 0;
  if (int rc = gram_tasks_for(s, mt, &tk, &ntk)) return rc;
  if (e == hipSuccess)
    e = launch_gram(s->X, s->aux, s->ld, s->gcols, s->Wv, rps, tk, ntask, nslab, s->gpart, ntiles, s->Gt, s->ctrl,
                    slot, 1, s->st);
  const CholFuse fbz = chol_fallback_only(s);
  if (e == hipSuccess)
    e = mt <= 16 ? launch_chol(s->Gt, T0 + 1, mt, 2.0 * lambda, 1, nullptr, nullptr, s->bcur, &s->ctrl->info, s->ctrl,
                               slot, 1, s->st, &fbz)
                 : launch_chol_big(s->Gt, T0 + 1, mt, 2.0 * lambda, 1, nullptr, nullptr, s->bcur, &s->ctrl->info,
                                   s->rdiag, s->zbig, s->ctrl, slot, 1, s->st);
  if (e == hipSuccess && mt <= 16 && s->glm_fallback)
    e = launch_sym_fallback(s->Gt, T0 + 1, mt, 2.0 * lambda, 1, nullptr, nullptr, s->bcur, &s->ctrl->info, s->ctrl, slot,
                            s->st, &fbz);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_glm_irls_step: ") + hipGetErrorString(e));
  return 0;
}

int enqueue_glm_tail(bessx_session *s, int slot, int T0, int rs) {
  hipError_t e = launch_commit(s->ctrl, slot, T0, s->A_new, s->bprev, 1, 1, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                               s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
  if (e == hipSuccess)
    e = launch_glm_eta_gh(s->model_type, s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->logfact, s->ctrl, slot,
                          s->A_cur, s->b_cur, s->r_rs[rs], s->h_rs[rs], s->sse, s->st);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_glm_tail: ") + hipGetErrorString(e));
  return 0;
}

// --------------------------------------------------------------------------------------------
// Cox: GroupPdasCox::get_A (two passes over X: block sums, then per-column suffix scans with carries) and
// primary_model_fit (damped Newton with step halving, everything gated on the device).
// --------------------------------------------------------------------------------------------
// Cox work space for sparsity levels above 254: the n x k matrix M = S1/S0, the second Gram and the scan scratch grow
// to the tile-rounded size of the level asked for (once; the largest level of a path comes first only by luck, so the
// growth is geometric).
int cox_reserve(bessx_session *s, int T0) {
  const size_t need = (size_t)((T0 + 1 + 15) / 16) * 16;
  if (need <= s->cox_M_cols) return 0;
  size_t cols = std::min<size_t>((size_t)s->capA, std::max(need, 2 * s->cox_M_cols));
  CoxBufs &c = s->cox;
  HIPX(hipStreamSynchronize(s->st));
  auto regrow = [&](double **ptr, size_t count) -> hipError_t {
    for (auto &q : s->cox_allocs)
      if (q == *ptr) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        hipError_t e = dmalloc(ptr, count);
        q = *ptr;
        if (e == hipSuccess) e = hipMemset(*ptr, 0, count * sizeof(double));
        return e;
      }
    return hipErrorInvalidValue;
  };
  const size_t mt = cols / 16;
  HIPX(regrow(&c.M, (size_t)s->ld * cols));
  HIPX(regrow(&c.Gt2, mt * (mt + 1) / 2 * 256));
  HIPX(regrow(&c.SCR, cox_scan_scratch_doubles(s->ld, (int)cols)));
  s->cox_M_cols = cols;
  return 0;
}

int enqueue_cox_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs) {
  const int mt = (T0 + 1 + 15) / 16, mp = mt * 16;
  hipError_t e = hipSuccess;
  if (!skip_k1) {
    if (int rc = score_pass(s, 2, rs, slot, k1_pairs)) return rc;
  } else if (s->timing) {
    k1_pairs.push_back({(size_t)-1, false});
  }
  if (e == hipSuccess)
    e = launch_cox_score(s->part_rs[rs], s->cox.one_pass ? nullptr : s->part2_rs[rs], s->nrb, s->p, s->beta_dense,
                         lambda, s->always, s->bd, s->ctrl, slot, s->st);
  if (e == hipSuccess) e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
  // (one-pass Hessian: the column behind the active ones is the bookkeeping column of k_cox_hess)
  const int aux_col = (s->cox.hess_fused && cox_hess_applies(mt)) ? 2 : 0;
  if (e == hipSuccess) e = launch_gram_cols(s->A_new, T0, mp, 0, aux_col, s->gcols, s->ctrl, slot, s->A_cur, 1, s->st);
  if (e == hipSuccess) e = launch_cox_newton_begin(s->ctrl, slot, T0, s->cox, s->idcols, s->st);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_cox_head: ") + hipGetErrorString(e));
  return 0;
}

int enqueue_cox_newton(bessx_session *s, int slot, int t, int T0, double lambda, int rs) {
  const int mt = (T0 + 1 + 15) / 16;
  const GramTask *tk = nullptr;
  int ntask = 0;
  if (int rc = gram_tasks_for(s, mt, &tk, &ntask)) return rc;
  const int ntiles = mt * (mt + 1) / 2;
  int rps, nslab;
  gram_geometry(s, ntask, &rps, &nslab, ntiles);
  if ((size_t)nslab * ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
  hipError_t e = launch_cox_newton_step(s->X, s->aux, s->ld, s->n, s->mask[rs], s->ctrl, slot, t, s->A_new, T0, lambda,
                                        s->gcols, s->idcols, mt, tk, ntask, rps, nslab, s->gpart, ntiles, s->Gt, s->cox,
                                        s->st, s->rdiag, s->zbig);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_cox_newton: ") + hipGetErrorString(e));
  return 0;
}

int enqueue_cox_tail(bessx_session *s, int slot, int T0, int rs) {
  hipError_t e = launch_commit(s->ctrl, slot, T0, s->A_new, s->cox.b0, 0, 1, s->A_cur, s->b_cur, s->beta_dense, s->hist,
                               s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
  if (e == hipSuccess)
    e = launch_cox_state(s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->cox,
                         s->sse, s->st);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("enqueue_cox_tail: ") + hipGetErrorString(e));
  return 0;
}

// Results of the kernels queued so far.  kcopy >= 0: only the first kcopy coefficients / indices are wanted and
// the block is published by a kernel into pinned memory (k_publish) while the host spins on its sequence number --
// no copy engine, no interrupt.  kcopy < 0 (or test hook publish=0): plain asynchronous copy + stream synchronisation.
// what a publication of the result block into pinned buffer `buf` copies; takes the next sequence number
PubArgs publish_args(bessx_session *s, int kcopy, int buf, unsigned long long *seq) {
  *seq = ++s->pub_seq;
  PubArgs pa = {s->resblk,
                s->res_buf[buf],
                128,
                (size_t)((unsigned char *)s->sse - s->resblk),
                2 * s->n_sse_blk,
                (size_t)((unsigned char *)s->b_cur - s->resblk),
                (size_t)((unsigned char *)s->A_cur - s->resblk),
                std::min(kcopy, s->capA),
                s->pub_flag + 8 * buf,
                *seq,
                s->cov_mode ? s->cov[0].meta : nullptr,
                1,
                s->snap[buf],
                s->res_bytes};
  return pa;
}

// the publication that follows a snapshot: same target and sequence number, source = the snapshot
PubArgs publish_from_snapshot(const PubArgs &tail) {
  PubArgs pa = tail;
  pa.on = 1;
  pa.dev = tail.snap;
  pa.count_ptr = reinterpret_cast<const int *>(tail.snap + tail.snap_count_off);
  return pa;
}

// a deferred publication nobody has attached to a launch: issue it as a launch of its own
int publish_flush(bessx_session *s) {
  if (!s->pend_on) return 0;
  s->pend_on = false;
  const PubArgs &pa = s->pend;
  HIPX(launch_publish(pa.dev, pa.host, pa.ctrl_bytes, pa.off_sse, pa.n_sse, pa.off_b, pa.off_a, pa.kcopy, pa.seq_host,
                      pa.seq, s->st, pa.count_ptr));
  return 0;
}

int publish_launch(bessx_session *s, const PubArgs &pa) {
  HIPX(launch_publish(pa.dev, pa.host, pa.ctrl_bytes, pa.off_sse, pa.n_sse, pa.off_b, pa.off_a, pa.kcopy, pa.seq_host,
                      pa.seq, s->st, pa.count_ptr));
  return 0;
}

int publish_enqueue(bessx_session *s, int kcopy, int buf, unsigned long long *seq) {
  return publish_launch(s, publish_args(s, kcopy, buf, seq));
}

int publish_wait(bessx_session *s, int buf, unsigned long long want) {
  volatile unsigned long long *flag = s->pub_flag + 8 * buf;
  s->res_h = s->res_buf[buf];
  s->dbg_waits++;
  if (*flag >= want) s->dbg_waits_ready++;  // the result was there already: the device is ahead of the host
  std::chrono::steady_clock::time_point t0;
  bool timed = false;
  for (unsigned spins = 1;; spins++) {
    if (*flag >= want) break;
    // the wall clock every 2^14 spins (~0.5 ms; reading it is ~20 ns and touches nothing the device sees)
    if ((spins & 0x3fff) == 0) {
      const auto now = std::chrono::steady_clock::now();
      if (!timed) {
        t0 = now;
        timed = true;
      } else if (std::chrono::duration<double>(now - t0).count() > s->wait_deadline_s) {
        const hipError_t q = hipStreamQuery(s->st);
        if (q == hipSuccess && *flag >= want) break;
        return fail(BESSX_ERR_HIP, "no result block from the device within " + std::to_string(s->wait_deadline_s) +
                                       " s (BESSX_WAIT_TIMEOUT_S); stream status: " + hipGetErrorString(q) +
                                       " -- the session can only be destroyed now");
      }
    }
    // (rarely: a hipStreamQuery puts a marker with a system-scope release behind the last queued kernel, and the
    // kernel after it then starts ~4 us late -- once per fit when the query ran every 1024 spins, tools/ktrace.py)
    if ((spins & 0xfffff) == 0) {
      hipError_t q = hipStreamQuery(s->st);
      if (q == hipSuccess) {
        if (*flag >= want) break;
        return fail(BESSX_ERR_HIP, "read_results: the published result block did not become visible");
      }
      if (q != hipErrorNotReady) return fail(BESSX_ERR_HIP, std::string("read_results: ") + hipGetErrorString(q));
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

int read_results(bessx_session *s, int kcopy) {
  if (kcopy < 0 || !s->publish) {
    s->res_h = s->res_buf[0];
    HIPX(hipMemcpyAsync(s->res_h, s->resblk, s->res_bytes, hipMemcpyDeviceToHost, s->st));
    HIPX(hipStreamSynchronize(s->st));
    return 0;
  }
  unsigned long long want = 0;
  if (int rc = publish_enqueue(s, kcopy, 0, &want)) return rc;
  return publish_wait(s, 0, want);
}

// --------------------------------------------------------------------------------------------
// Group mode (some group has more than one column): Algorithm::fit with per-group sacrifices.  The host reads the
// selected group ids back after the top-k to expand them into columns (find_ind, src/utilities.cpp:113-130), so
// this loop synchronises twice per PDAS iteration and uses none of the speculative / cached fast paths.
// --------------------------------------------------------------------------------------------
// LM with groups: the diagonalisation of every group's block for (row set, lambda), formed on first use by one ungated
// launch of the score kernel in its store mode (the scores it writes are overwritten by the first real launch)
static int group_eig_ensure(bessx_session *s, int rs, double lambda) {
  if (!s->geig_on || s->gmax > GRP_EIG_MAX) return 1;  // (wider groups take the Cholesky-form score kernel: nothing to keep)
  if ((int)s->geig_v_rs.size() <= rs) {
    s->geig_v_rs.resize((size_t)rs + 1, nullptr);
    s->geig_l_rs.resize((size_t)rs + 1, nullptr);
    s->geig_lambda.resize((size_t)rs + 1, 0.0);
    s->geig_valid.resize((size_t)rs + 1, 0);
  }
  if (!s->geig_v_rs[rs]) {
    if (dmalloc(&s->geig_v_rs[rs], (size_t)s->goff_h[s->N]) != hipSuccess ||
        dmalloc(&s->geig_l_rs[rs], (size_t)s->p) != hipSuccess) {
      (void)hipGetLastError();
      return 1;  // (no memory: diagonalise at every iteration as before)
    }
  }
  if (s->geig_valid[rs] && s->geig_lambda[rs] == lambda) return 0;
  // (part = xty of the row set, one "row block": any finite d does -- only the stored decomposition is kept)
  HIPX(launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->xty[rs], 1, s->p, 1,
                          (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork, s->zwork,
                          nullptr, 0, 1, s->geig_v_rs[rs], s->geig_l_rs[rs]));
  s->geig_valid[rs] = 1;
  s->geig_lambda[rs] = lambda;
  return 0;
}

// Grouped LM in the covariance form: a fit parked on missing Gram columns (cov_stall = 1).  The fill list is built on the
// host -- the missing columns, then the uncached columns of the groups this iteration's sacrifices rank highest, whole
// groups first (a group enters the active set with all its columns), 64 columns in all: one pass of the pair kernel --
// and takes the next free slots of the cache (prefill_begin, in-fit form).  Then the wake-up; the caller queues the rest
// of the stalled slot.
static int grouped_cov_fill(bessx_session *s, const FitCtrl *hc, int T0) {
  (void)T0;
  bessx_session::CovCache &cv = s->cov[0];
  const int p = s->p, N = s->N, nm = hc->cov_nmiss;
  if (nm < 1 || nm > s->capA) return fail(BESSX_ERR_NUMERIC, "internal error: parked grouped fit without a request");
  std::vector<int> list((size_t)nm), slot_h((size_t)p);
  std::vector<double> sc((size_t)N);
  HIPX(hipMemcpyAsync(list.data(), s->cov_fcols, (size_t)nm * sizeof(int), hipMemcpyDeviceToHost, s->st));
  HIPX(hipMemcpyAsync(sc.data(), s->bd, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, s->st));
  HIPX(hipMemcpyAsync(slot_h.data(), cv.slot_of, (size_t)p * sizeof(int), hipMemcpyDeviceToHost, s->st));
  HIPX(hipStreamSynchronize(s->st));
  for (int c : list) slot_h[(size_t)c] = 0;  // (listed: not a candidate for the rest of the list)
  std::vector<int> cand;  // groups with uncached columns, best sacrifice first (ties by group number)
  for (int g = 0; g < N; g++) {
    bool open = false;
    for (int j = 0; j < s->gsz_h[g] && !open; j++) open = slot_h[(size_t)s->gidx_h[g] + j] < 0;
    if (open) cand.push_back(g);
  }
  const int want = std::max(2 * COV_R, (nm + COV_R - 1) / COV_R * COV_R);
  std::sort(cand.begin(), cand.end(), [&](int a, int b) { return sc[a] > sc[b] || (sc[a] == sc[b] && a < b); });
  for (size_t q = 0; q < cand.size() && (int)list.size() < want; q++) {
    const int g = cand[q];
    for (int j = 0; j < s->gsz_h[g] && (int)list.size() < want; j++)
      if (slot_h[(size_t)s->gidx_h[g] + j] < 0) list.push_back(s->gidx_h[g] + j);
  }
  const int total = (int)list.size() / COV_R * COV_R;
  if (total < nm || total > s->capA) {
    // hardly an uncached column left to round the list up to whole 32-column groups (or an over-long request): the
    // missing columns alone, listed on the device as the ungrouped fit's private fill does (last group partly filled)
    HIPX(launch_cov_fill_list(s->cov_fcols, s->cov_extras, s->bd2, cv.slot_of, cv.meta, s->ctrl, 1, s->st, s->cov_spec, 0));
    if (int rc = enqueue_cov_fill(s, 0, (nm + COV_R - 1) / COV_R, 1)) return rc;
    if (s->timing) {
      HIPX(hipStreamSynchronize(s->st));
      if (int rc = cov_collect(s, nm)) return rc;
    }
    HIPX(launch_cov_resume(s->ctrl, s->st));
    return 0;
  }
  if (int rc = prefill_begin(s, list.data(), total, 2)) return rc;
  if (int rc = enqueue_cov_fill(s, 0, total / COV_R, 1, s->fill_ctrl, 0, false)) return rc;
  s->cov_panel_groups += total / COV_R;
  HIPX(launch_cov_compact(cv.G, p, cv.slot_of, s->cov_fcols, 0, total / COV_R, cv.GS, s->cov_cs, s->fill_ctrl, 1, s->st,
                          s->xtx[0], cv.meta));
  s->prefill_cols = 0;
  if (s->timing) {  // (the panel launches' event pairs belong to THIS list)
    HIPX(hipStreamSynchronize(s->st));
    if (int rc = cov_collect(s, total)) return rc;
  }
  HIPX(launch_cov_resume(s->ctrl, s->st));
  return 0;
}

int algorithm_fit_grouped(bessx_session *s) {
  const int T0 = s->sparsity_level, rs = s->cur_rows, fam = s->model_type;
  const double lambda = s->lambda_level;
  if (T0 < 1 || T0 > s->N) return fail(BESSX_ERR_ARG, "sparsity level (number of groups) outside [1, number of groups]");
  if (fam == 4 && !(s->algorithm_type == 2 || s->algorithm_type == 3))
    return fail(BESSX_ERR_UNSUPPORTED, "Cox with groups of size > 1 exists only for algorithm_type 2 / 3 (the group "
                                       "branch of GroupPdasCox::get_A)");
  const bool cox = fam == 4;
  if (!topk_supported(s->N, T0)) return fail(BESSX_ERR_UNSUPPORTED, "top-k selection: too many groups for this sparsity level");
  const bool glm = fam != 1;
  // beta <- beta_init
  const int k_init = (int)s->beta_init.idx.size();
  if (k_init > s->cap) return fail(BESSX_ERR_ARG, "initial support too large");
  // LM, groups of one width, all rows, covariance form (bessx_session_create decides): no residual, no pass over X
  // per PDAS iteration -- d = X^T y - G_A beta_A from the cached Gram columns of the active groups' columns
  const bool gcov = !glm && s->cov_mode && rs == 0 && s->g_uniform > 0 && !s->trace.on &&
                    (long)T0 * s->g_uniform + 2 <= (long)s->capA && k_init <= s->capA;
  // ... and a fit that starts from exactly the model the previous fit of this row set left on the device (the next
  // candidate of a warm-start path) needs no upload and no look-up of its initial support: those columns were the
  // active set a moment ago
  bessx_session::RsCache &cc = s->cache[rs];
  const bool cont = gcov && cc.valid && cc.cov_layout && s->dev_state_rs == rs && cc.coef0 == s->coef0_init &&
                    cc.beta.idx == s->beta_init.idx && cc.beta.val == s->beta_init.val;
  cc.valid = cc.model_only = false;
  if (!cont) {
    int *st_idx = reinterpret_cast<int *>(s->stage_h);
    double *st_val = reinterpret_cast<double *>(s->stage_h + (size_t)s->capA * sizeof(int));
    for (int i = 0; i < k_init; i++) {
      st_idx[i] = s->beta_init.idx[i];
      st_val[i] = s->beta_init.val[i];
    }
    if (k_init) {
      HIPX(hipMemcpyAsync(s->init_idx_d, st_idx, k_init * sizeof(int), hipMemcpyHostToDevice, s->st));
      HIPX(hipMemcpyAsync(s->init_val_d, st_val, k_init * sizeof(double), hipMemcpyHostToDevice, s->st));
    }
  }
  hipError_t e = cont ? launch_fit_continue(s->ctrl, T0, s->hist, s->st, ++s->fit_serial, 0, 0)
                      : launch_fit_begin(s->ctrl, T0, k_init, s->init_idx_d, s->init_val_d, s->coef0_init, s->A_cur,
                                         s->b_cur, s->beta_dense, s->p, s->hist, s->st);
  if (e == hipSuccess && cont) {
    // nothing else to set up
  } else if (e == hipSuccess && gcov) {
    if (k_init > 0) {
      // the first score pass multiplies the cached Gram columns of the initial support: form the missing ones
      bessx_session::CovCache &cv = s->cov[0];
      e = launch_cov_need(s->A_cur, k_init, nullptr, s->bd2, s->p, cv.slot_of, cv.meta, cov_C_dev(s), s->cov_fcols,
                          s->ctrl, 0, s->A_cur, s->st);
      if (e == hipSuccess)
        e = launch_cov_fill_list(s->cov_fcols, s->cov_extras, s->bd2, cv.slot_of, cv.meta, s->ctrl, 0, s->st, s->cov_spec, 0);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group cov begin: ") + hipGetErrorString(e));
      if (int rc = enqueue_cov_fill(s, 0, (k_init + COV_R - 1) / COV_R, 0)) return rc;
    }
  } else if (e == hipSuccess) {
    if (!glm)
      e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, 0, s->A_cur, s->b_cur, s->r_rs[rs], s->sse,
                          s->st);
    else if (cox)
      e = launch_cox_state(s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->ctrl, 0, s->A_cur, s->b_cur, s->cox, s->sse,
                           s->st);
    else
      e = launch_glm_eta_gh(fam, s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->logfact, s->ctrl, 0, s->A_cur,
                            s->b_cur, s->r_rs[rs], s->h_rs[rs], s->sse, s->st);
  }
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group fit begin: ") + hipGetErrorString(e));
  if (cox) s->cox_state_rs = -1;  // the ungrouped path's per-row-set reuse does not apply here
  s->dev_state_rs = -1;
  const FitCtrl *hc = reinterpret_cast<const FitCtrl *>(s->res_h);
  std::vector<int> G(T0), cols;
  std::vector<std::vector<int>> cols_hist;
  int slot = 1;
  // LM, every group of the same width: the selected groups are expanded to columns ON THE DEVICE (k_group_expand), so
  // the number of active columns is known up front, the PDAS iterations are queued two at a time as gated slots like
  // the ungrouped fit's, and the host reads ONE result block per batch -- one round trip for a warm-started fit that
  // ends within two iterations (round 3: two synchronisations per iteration).  Ragged groups, the traced path and the
  // other families keep the host-side expansion below.
  if (!glm && s->g_uniform > 0 && !s->trace.on && (long)T0 * s->g_uniform + 2 <= (long)s->capA) {
    const int gs = s->g_uniform, K = T0 * gs;
    const int mt = (K + 1 + 15) / 16, mp = mt * 16, ntiles = mt * (mt + 1) / 2;
    const GramTask *tk = nullptr;
    int ntask = 0, rps, nslab;
    if (int rc = gram_tasks_for(s, mt, &tk, &ntask)) return rc;
    gram_geometry(s, ntask, &rps, &nslab, ntiles);
    if ((size_t)nslab * ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
    const CholFuse fbz = chol_fallback_only(s);
    std::vector<std::pair<size_t, bool>> k1_pairs;
    const bool eig = group_eig_ensure(s, rs, lambda) == 0;
    // solve + commit of a slot whose columns are all cached (covariance form), and the sums of squares if the fit ends
    auto cov_rest = [&](int sl) -> hipError_t {
      bessx_session::CovCache &cv = s->cov[0];
      hipError_t q = launch_cov_gram(cv.G, s->p, cv.slot_of, s->gcols_new, K, mt, s->Gt, cv.meta, s->ctrl, sl, s->st);
      if (q == hipSuccess)
        q = mt <= 16 ? launch_chol(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl, sl,
                                   0, s->st, &fbz)
                     : launch_chol_big(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info,
                                       s->rdiag, s->zbig, s->ctrl, sl, 0, s->st);
      if (q == hipSuccess && mt <= 16)
        q = launch_sym_fallback(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl, sl,
                                s->st, &fbz);
      if (q == hipSuccess)
        q = launch_commit_group(s->ctrl, sl, T0, s->A_new, K, s->gcols_new, s->sol, 0, 0, s->A_cur, s->b_cur,
                                s->beta_dense, s->hist, s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
      if (q == hipSuccess)
        q = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, sl, s->A_cur, s->b_cur, s->r_rs[rs], s->sse,
                            s->st, 1);
      return q;
    };
    while (gcov && slot <= s->max_iter) {
      bessx_session::CovCache &cv = s->cov[0];
      for (int b = 0; b < 2 && slot <= s->max_iter; b++, slot++) {
        // d (and, unused here, the singleton scores: into bd2) from the cache; the group sacrifices from d
        e = launch_cov_d(cv.G, s->p, cv.slot_of, s->xty[rs], s->A_cur, s->b_cur, s->dcol, s->beta_dense, s->xtx[rs],
                         (double)s->n_train[rs], lambda, s->always, s->bd2, s->inA, s->cov_bmm, s->ctrl, slot, s->st);
        if (e == hipSuccess)
          e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->dcol, 1, s->p, 1,
                                 (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork,
                                 s->zwork, s->ctrl, slot, eig ? 2 : 0, eig ? s->geig_v_rs[rs] : nullptr,
                                 eig ? s->geig_l_rs[rs] : nullptr);
        if (e == hipSuccess)
          e = launch_topk(s->bd, s->N, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
        if (e == hipSuccess) e = launch_group_expand(s->A_new, T0, gs, s->gidx, s->gcols_new, s->ctrl, slot, s->st);
        // repeated set (same_prev) + cache lookup of the expanded columns: parks the fit when one is missing
        if (e == hipSuccess)
          e = launch_cov_need(s->gcols_new, K, nullptr, s->bd2, s->p, cv.slot_of, cv.meta, cov_C_dev(s), s->cov_fcols,
                              s->ctrl, slot, s->A_cur, s->st, 1);
        if (e == hipSuccess) e = cov_rest(slot);
        if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group fit (covariance form): ") + hipGetErrorString(e));
      }
      if (int rc = read_results(s)) return rc;
      if (int rc = cov_collect(s, hc->cov_nfill)) return rc;
      if (hc->cov_miss) return fail(BESSX_ERR_NUMERIC, "internal error: an active column was missing from the Gram column cache");
      if (hc->cov_stall == 1) {
        const int stalled = -1 - hc->l + 1;
        if (int rc = grouped_cov_fill(s, hc, T0)) return rc;
        e = cov_rest(stalled);
        if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group fit (after the fill): ") + hipGetErrorString(e));
        slot = stalled + 1;
        if (slot > s->max_iter) {  // (the stalled slot was the last one: its result)
          if (int rc = read_results(s)) return rc;
          if (int rc = cov_collect(s, hc->cov_nfill)) return rc;
        }
        continue;
      }
      if (hc->cov_stall) return fail(BESSX_ERR_NUMERIC, "internal error: unexpected stall of a grouped fit");
      if (hc->done) break;
    }
    if (gcov) s->cov_panel_groups += hc->cov_groups;  // (fills gated on the fit's own control block: initial support, fallback)
    if (gcov && !hc->done) {
      // out of iterations: the sums of squares of the last coefficients have not been formed yet
      HIPX(launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, hc->l, s->A_cur, s->b_cur, s->r_rs[rs], s->sse,
                           s->st, 2));
      if (int rc = read_results(s)) return rc;
    }
    while (!gcov && slot <= s->max_iter) {
      const int first = slot;
      for (int b = 0; b < 2 && slot <= s->max_iter; b++, slot++) {
        hipEvent_t ea = nullptr, eb = nullptr;
        if (int rc = k1_begin(s, &ea, &eb)) return rc;
        e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], nullptr, s->part_rs[rs], nullptr, s->ctrl, slot, s->st);
        if (s->timing && e == hipSuccess) {
          e = hipEventRecord(eb, s->st);
          k1_pairs.push_back({s->ev_used - 2, false});
        }
        // (the row blocks of X^T r summed once, coalesced over the columns, instead of by every group's thread: same order
        // of summation, 296 -> 25 us per iteration at 2000 groups, 49 row blocks)
        if (e == hipSuccess) e = launch_part_sum(s->part_rs[rs], s->nrb, s->p, s->dcol, s->st);
        if (e == hipSuccess)
          e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->dcol, 1, s->p, 1,
                                 (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork,
                                 s->zwork, s->ctrl, slot, eig ? 2 : 0, eig ? s->geig_v_rs[rs] : nullptr,
                                 eig ? s->geig_l_rs[rs] : nullptr);
        if (e == hipSuccess)
          e = launch_topk(s->bd, s->N, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
        if (e == hipSuccess) e = launch_group_expand(s->A_new, T0, gs, s->gidx, s->gcols_new, s->ctrl, slot, s->st);
        if (e == hipSuccess) e = launch_gram_cols(s->gcols_new, K, mp, 0, 0, s->gcols, s->ctrl, slot, s->A_cur, 0, s->st);
        if (e == hipSuccess)
          e = launch_gram(s->X, s->aux, s->ld, s->gcols, s->mask[rs], rps, tk, ntask, nslab, s->gpart, ntiles, s->Gt,
                          s->ctrl, slot, 0, s->st, 0);
        if (e == hipSuccess)
          e = mt <= 16 ? launch_chol(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl,
                                     slot, 0, s->st, &fbz)
                       : launch_chol_big(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info,
                                         s->rdiag, s->zbig, s->ctrl, slot, 0, s->st);
        if (e == hipSuccess && mt <= 16)
          e = launch_sym_fallback(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl, slot,
                                  s->st, &fbz);
        if (e == hipSuccess)
          e = launch_commit_group(s->ctrl, slot, T0, s->A_new, K, s->gcols_new, s->sol, 0, 0, s->A_cur, s->b_cur,
                                  s->beta_dense, s->hist, s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
        if (e == hipSuccess)
          e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->r_rs[rs],
                              s->sse, s->st);
        if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group fit (device expansion): ") + hipGetErrorString(e));
      }
      if (int rc = read_results(s)) return rc;
      for (size_t i = 0; i < k1_pairs.size(); i++) k1_pairs[i].second = (first + (int)i) <= hc->l;
      if (int rc = k1_collect(s, k1_pairs)) return rc;
      k1_pairs.clear();
      if (hc->done) break;
    }
    slot = s->max_iter + 1;  // (skip the host-side loop below)
  }
  for (; slot <= s->max_iter; slot++) {
    // ---- get_A: per-group sacrifices and top-k over the groups
    if (!glm) {
      e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], nullptr, s->part_rs[rs], nullptr, nullptr, 0, s->st);
      if (e == hipSuccess) e = launch_part_sum(s->part_rs[rs], s->nrb, s->p, s->dcol, s->st);
      const bool eig = e == hipSuccess && group_eig_ensure(s, rs, lambda) == 0;
      if (e == hipSuccess)
        e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->dcol, 1, s->p, 1,
                               (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork,
                               s->zwork, nullptr, 0, eig ? 2 : 0, eig ? s->geig_v_rs[rs] : nullptr,
                               eig ? s->geig_l_rs[rs] : nullptr);
    } else if (cox) {
      // X_g^T h X_g without the n x n Hessian of src/Algorithm.h:1536-1546 (launch_cox_group_moments)
      e = launch_cox_group_moments(s->X, s->ld, s->n, s->p, s->cox, s->allcols, (int)std::min<size_t>(s->cox_M_cols, 256),
                                   s->gmax, s->N, s->gidx_h.data(), s->gsz_h.data(), s->gidx, s->gsz, s->goff,
                                   (long)s->goff_h[s->N], s->mblk, s->mblk2, s->dcol, s->st);
      if (e == hipSuccess)
        e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->mblk, s->dcol, nullptr, 0, s->p, 0, 1.0, lambda,
                               s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork, s->zwork);
    } else {
      e = launch_group_moments(s->gmax, s->X, s->ld, s->n, s->h_rs[rs], s->r_rs[rs], s->N, s->gidx, s->gsz, s->goff,
                               s->mblk, s->dcol, s->st);
      if (e == hipSuccess)
        e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->mblk, s->dcol, nullptr, 0, s->p, 0, 1.0, lambda,
                               s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork, s->zwork);
    }
    if (e == hipSuccess) e = launch_topk(s->bd, s->N, T0, s->A_new, s->cand, nullptr, 0, s->st, nullptr, nullptr, &s->tie);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group get_A: ") + hipGetErrorString(e));
    HIPX(hipMemcpyAsync(G.data(), s->A_new, (size_t)T0 * sizeof(int), hipMemcpyDeviceToHost, s->st));
    HIPX(hipStreamSynchronize(s->st));
    // ---- find_ind
    cols.clear();
    if (T0 == s->N) {
      for (int j = 0; j < s->p; j++) cols.push_back(j);
    } else {
      for (int g : G)
        for (int j = 0; j < s->gsz_h[g]; j++) cols.push_back(s->gidx_h[g] + j);
    }
    const int K = (int)cols.size();
    if (K + 2 > s->capA) return fail(BESSX_ERR_ARG, "selected groups span more columns than this session's capacity");
    cols_hist.push_back(cols);
    HIPX(hipMemcpyAsync(s->gcols_new, cols.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, s->st));
    // ---- primary_model_fit on the expanded columns
    if (!glm) {
      const int mt = (K + 1 + 15) / 16, mp = mt * 16, ntiles = mt * (mt + 1) / 2;
      const GramTask *tk = nullptr;
      int ntask = 0, rps, nslab;
      if (int rc = gram_tasks_for(s, mt, &tk, &ntask)) return rc;
      gram_geometry(s, ntask, &rps, &nslab, ntiles);
      const CholFuse fbz = chol_fallback_only(s);
      e = launch_gram_cols(s->gcols_new, K, mp, 0, 0, s->gcols, s->ctrl, slot, s->A_cur, 0, s->st);
      if (e == hipSuccess)
        e = launch_gram(s->X, s->aux, s->ld, s->gcols, s->mask[rs], rps, tk, ntask, nslab, s->gpart, ntiles, s->Gt,
                        s->ctrl, slot, 0, s->st, 0);
      if (e == hipSuccess)
        e = mt <= 16 ? launch_chol(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl,
                                   slot, 0, s->st, &fbz)
                     : launch_chol_big(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info,
                                       s->rdiag, s->zbig, s->ctrl, slot, 0, s->st);
      if (e == hipSuccess && mt <= 16)  // (wide groups on few rows: more columns than independent rows)
        e = launch_sym_fallback(s->Gt, K, mt, lambda, 0, s->xty[rs], s->gcols_new, s->sol, &s->ctrl->info, s->ctrl, slot,
                                s->st, &fbz);
      if (e == hipSuccess)
        e = launch_commit_group(s->ctrl, slot, T0, s->A_new, K, s->gcols_new, s->sol, 0, 0, s->A_cur, s->b_cur,
                                s->beta_dense, s->hist, s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
      if (e == hipSuccess)
        e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->r_rs[rs],
                            s->sse, s->st);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group fit: ") + hipGetErrorString(e));
      if (int rc = read_results(s)) return rc;
    } else if (cox) {
      // GroupPdasCox::primary_model_fit on the expanded columns: the Newton chain of the ungrouped path
      if (int rc = cox_reserve(s, K)) return rc;
      const int mp = (K + 1 + 15) / 16 * 16;
      e = launch_gram_cols(s->gcols_new, K, mp, 0, (s->cox.hess_fused && cox_hess_applies(mp / 16)) ? 2 : 0, s->gcols,
                           s->ctrl, slot, s->A_cur, 0, s->st);
      if (e == hipSuccess) e = launch_cox_newton_begin(s->ctrl, slot, K, s->cox, s->idcols, s->st);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group cox begin: ") + hipGetErrorString(e));
      const int tmax = 30;
      int t = 1;  // Newton steps are numbered from 1 (:1411)
      int *saved = s->A_new;
      s->A_new = s->gcols_new;  // the step kernels take (column list, count)
      int rc = 0;
      while (true) {
        int upto = std::min(tmax, t + std::max(2, s->irls_guess) - 1);
        for (; t <= upto && rc == 0; t++) rc = enqueue_cox_newton(s, slot, t, K, lambda, rs);
        if (rc) break;
        e = launch_commit_group(s->ctrl, slot, T0, saved, K, s->gcols_new, s->cox.b0, 0, 1, s->A_cur, s->b_cur,
                                s->beta_dense, s->hist, s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
        if (e == hipSuccess)
          e = launch_cox_state(s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->ctrl, slot, s->A_cur, s->b_cur, s->cox,
                               s->sse, s->st);
        if (e != hipSuccess) {
          rc = fail(BESSX_ERR_HIP, std::string("group cox tail: ") + hipGetErrorString(e));
          break;
        }
        rc = read_results(s);
        if (rc || hc->l == slot) break;
        if (t > tmax) {
          rc = fail(BESSX_ERR_NUMERIC, "Newton chain did not terminate");
          break;
        }
      }
      s->A_new = saved;
      if (rc) return rc;
      if (hc->irls_last > 0) s->irls_guess = std::min(tmax + 1, hc->irls_last + 1);
    } else {
      int mt, mp, ntask, ntiles, rps, nslab;
      if (int rc = glm_geometry(s, K, &mt, &mp, &ntask, &ntiles, &rps, &nslab)) return rc;
      e = launch_gram_cols(s->gcols_new, K, mp, 1, 1, s->gcols, s->ctrl, slot, s->A_cur, 0, s->st);
      if (e == hipSuccess) e = launch_glm_irls_begin(s->ctrl, slot, fam, K + 1, s->bcur, s->bprev, s->st);
      if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("group irls begin: ") + hipGetErrorString(e));
      const int tmax = fam == 2 ? 30 : 50;
      int t = 0;
      // the IRLS step kernels take (column list, count): hand them the expanded columns through A_new's slot
      int *saved = s->A_new;
      s->A_new = s->gcols_new;
      int rc = 0;
      while (true) {
        int upto = std::min(tmax, t + std::max(2, s->irls_guess) - 1);
        for (; t <= upto && rc == 0; t++) rc = enqueue_glm_irls_step(s, slot, t, K, lambda, rs);
        if (rc) break;
        e = launch_commit_group(s->ctrl, slot, T0, saved, K, s->gcols_new, s->bprev, 1, 1, s->A_cur, s->b_cur,
                                s->beta_dense, s->hist, s->hist_beta, s->hist_coef0, s->hist_stride, s->st);
        if (e == hipSuccess)
          e = launch_glm_eta_gh(fam, s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->logfact, s->ctrl, slot, s->A_cur,
                                s->b_cur, s->r_rs[rs], s->h_rs[rs], s->sse, s->st);
        if (e != hipSuccess) {
          rc = fail(BESSX_ERR_HIP, std::string("group glm tail: ") + hipGetErrorString(e));
          break;
        }
        rc = read_results(s);
        if (rc || hc->l == slot) break;
        if (t > tmax) {
          rc = fail(BESSX_ERR_NUMERIC, "IRLS chain did not terminate");
          break;
        }
      }
      s->A_new = saved;
      if (rc) return rc;
      if (hc->irls_last > 0) s->irls_guess = std::min(tmax + 1, hc->irls_last + 1);
    }
    if (hc->done) break;
  }
  if (hc->info == 2 && glm && !cox && !s->glm_fallback) {  // (see algorithm_fit)
    s->glm_fallback = true;
    HIPX(hipStreamSynchronize(s->st));
    HIPX(hipMemsetAsync(&s->ctrl->info, 0, sizeof(int), s->st));
    s->cache[rs].valid = s->cache[rs].model_only = false;
    s->dev_state_rs = -1;
    return algorithm_fit_grouped(s);
  }
  if (hc->info) return fail(BESSX_ERR_NUMERIC, "non-finite value in the k x k solve (singular Gram matrix?)");
  const int K = hc->k_cur;
  const double *sse_h = reinterpret_cast<const double *>(s->res_h + ((unsigned char *)s->sse - s->resblk));
  const double *b_h = reinterpret_cast<const double *>(s->res_h + ((unsigned char *)s->b_cur - s->resblk));
  const int *a_h = reinterpret_cast<const int *>(s->res_h + ((unsigned char *)s->A_cur - s->resblk));
  s->beta.idx.assign(a_h, a_h + K);
  s->beta.val.assign(b_h, b_h + K);
  s->coef0 = hc->coef0;
  s->l = hc->done ? hc->l : s->max_iter + 1;
  double tr = 0.0, te = 0.0;
  for (int b = 0; b < s->n_sse_blk; b++) {
    tr += sse_h[2 * b];
    te += sse_h[2 * b + 1];
  }
  s->sse_train = tr;
  s->sse_test = te;
  s->n_fits += 1;
  s->n_iters += hc->l;
  if (gcov && hc->done) {  // the device holds exactly this model: a fit that starts from it continues (see `cont`)
    cc.valid = true;
    cc.cov_layout = true;
    cc.lambda = lambda;
    cc.T0 = T0;
    cc.beta = s->beta;
    cc.coef0 = s->coef0;
    s->dev_state_rs = rs;
  }
  if (s->trace.on) {
    const int L = hc->l;
    std::vector<double> hb((size_t)(L + 1) * s->hist_stride), hc0(L + 1);
    HIPX(hipMemcpy(hb.data(), s->hist_beta, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPX(hipMemcpy(hc0.data(), s->hist_coef0, hc0.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int it = 1; it <= L; it++) {
      const std::vector<int> &cc = cols_hist[it - 1];
      s->trace.meta.push_back(it);
      s->trace.meta.push_back(T0);
      s->trace.meta.push_back(s->n_train[rs]);
      s->trace.meta.push_back((int)s->trace.a_flat.size());
      for (size_t i = 0; i < cc.size(); i++) {
        s->trace.a_flat.push_back(cc[i]);
        s->trace.beta_flat.push_back(hb[(size_t)it * s->hist_stride + i]);
      }
      s->trace.coef0_calls.push_back(hc0[it]);
    }
  }
  return 0;
}

// Queue the first batch of the fit the path function announced (hint) behind the fit `parent`: it starts on the
// device only if that fit ends on a repeated active set with fresh score sums (k_fit_continue, chained).
int enqueue_chained(bessx_session *s, const bessx_session::Hint &hint, int rs, int parent, int buf, int batch,
                           double parent_lambda, int parent_T0) {
  const int Tn = hint.T0;
  if (!(hint.on && s->chain && s->publish && s->warm_start && !s->trace.on && rs == 0 && s->cov_mode && Tn >= 1 &&
        Tn <= s->cap && Tn + COV_R + s->cov_spec <= s->cov_C && topk_supported(s->p, Tn)))
    return 0;
  bessx_session::Ahead &ah = s->ahead;
  s->chain_queued++;
  ah.armed = true;
  ah.T0 = Tn;
  ah.lambda = hint.lambda;
  ah.rs = rs;
  ah.serial = ++s->fit_serial;
  ah.buf = buf;
  // the first selection can open the fit itself when no kernel precedes it in slot 1 (same lambda: the scores stand)
  const bool cont_fusable = s->fuse && hint.lambda == parent_lambda && topk_can_fuse_need(s->p);
  // a deferred publication of the parent: rides on this fit's first kernel if that kernel opens the fit itself,
  // otherwise it goes out now
  PubArgs parent_pub = {};
  const bool have_parent_pub = s->pend_on && cont_fusable && s->defer_pub;
  if (have_parent_pub) {
    parent_pub = s->pend;
    s->pend_on = false;
  } else if (int rc = publish_flush(s)) {
    return rc;
  }
  if (!cont_fusable) HIPX(launch_fit_continue(s->ctrl, Tn, s->hist, s->st, ah.serial, 1, parent));
  const PubArgs pa = publish_args(s, Tn, buf, &ah.seq);
  bool published = false, snapshotted = false;
  for (int b = 0, sl = 1; b < batch && sl <= s->max_iter; b++, sl++) {
    SlotFuse sf;
    const bool last = b + 1 == batch || sl == s->max_iter;
    if (last) {
      sf.pub = &pa;
      sf.pub_snapshot = s->defer_pub;  // this batch's own result: snapshot now, publish with the next launch
    }
    if (sl == 1 && cont_fusable) {
      sf.cont = true;
      sf.cont_serial = ah.serial;
      sf.cont_parent = parent;
      if (have_parent_pub) sf.attach = &parent_pub;
    }
    if (int rc = enqueue_lm_slot_cov(s, sl, Tn, hint.lambda, rs, sl == 1, hint.lambda == parent_lambda,
                                     hint.lambda == parent_lambda && Tn == parent_T0 + 1, &sf))
      return rc;
    if (sl == 1 && cont_fusable && !sf.cont_fused) return fail(BESSX_ERR_HIP, "internal: chained fit was not opened");
    if (sl == 1 && have_parent_pub && !sf.attached) return fail(BESSX_ERR_HIP, "internal: deferred publication lost");
    published = published || sf.pub_fused;
    snapshotted = snapshotted || (sf.pub_fused && sf.pub_snapshot);
  }
  if (snapshotted) {  // the publication of this batch is pending: the next chained launch or publish_flush() issues it
    s->pend = publish_from_snapshot(pa);
    s->pend_on = true;
    return 0;
  }
  return published ? 0 : publish_launch(s, pa);
}

// One Algorithm::fit with the state set by the update_* style members of the session.

int algorithm_fit(bessx_session *s) {
  if (s->grouped) return algorithm_fit_grouped(s);
  const int T0 = s->sparsity_level, rs = s->cur_rows;
  const double lambda = s->lambda_level;
  if (rs != 0 && !s->fold_ctx.empty()) fold_contexts_invalidate(s);  // a fold fitted on the parent's own state
  if (T0 < 1 || T0 > s->cap)
    return fail(BESSX_ERR_ARG, "sparsity level " + std::to_string(T0) + " outside [1, " + std::to_string(s->cap) +
                                   "]: a session holds work space for min(p, max(2046, bessx_problem.max_sparsity))"
                                   " active columns, max_sparsity <= " + std::to_string(T0_HARD));
  if (s->model_type == 4)
    if (int rc = cox_reserve(s, T0)) return rc;
  if (!topk_supported(s->p, T0)) return fail(BESSX_ERR_UNSUPPORTED, "top-k selection: p too large for this sparsity level");
  const bool glm = s->model_type != 1;  // sub-model fit is an iteration chain (IRLS or Newton)
  const bool cox = s->model_type == 4;
  // warm start: this->beta = beta_init; this->coef0 = coef0_init (src/Algorithm.h:147-148)
  const int k_init = (int)s->beta_init.idx.size();
  if (k_init > s->cap) return fail(BESSX_ERR_ARG, "initial support too large");
  // Reuse across fits: when this fit starts from exactly the coefficients the last fit on this row set
  // ended with, and that fit ended on a repeated active set, the residual and the score-pass sums in
  // memory are the ones get_A would recompute (src/Algorithm.h:1109 depends only on beta, coef0 and the rows).
  bessx_session::RsCache &cc = s->cache[rs];
  // (Cox keeps its state vectors once per session, not per row set, so it only reuses within one row set.)
  // covariance-update form of the score pass for this fit (LM; the cache must be able to hold the active set)
  const bool cov = s->cov_mode && !glm && T0 + COV_R + s->cov_spec <= s->cov_C && k_init + COV_R + s->cov_spec <= s->cov_C;
  bool use_cache = cc.valid && cc.coef0 == s->coef0_init && cc.beta.idx == s->beta_init.idx &&
                   cc.beta.val == s->beta_init.val && (!cox || s->cox_state_rs == rs) && (glm || cc.cov_layout == cov);
  if (cox) s->cox_state_rs = rs;
  // Covariance form: the previous fit of the chain ended on a cycle of active sets (src/Algorithm.h: A equal to an EARLIER
  // column of A_list) -- its last solve changed the coefficients after the last score pass.  The device still holds
  // exactly the model this fit starts from: no upload, no look-up of its columns, only the scores are formed again
  // (round 5; before, such a fit was set up from the host's copy: two copies, k_fit_begin and the six launches of a
  // slot-0 fill that finds nothing missing -- 19 times per configs[1] path).
  bool model_kept = cov && !use_cache && cc.model_only && cc.cov_layout && s->dev_state_rs == rs &&
                          cc.coef0 == s->coef0_init && cc.beta.idx == s->beta_init.idx && cc.beta.val == s->beta_init.val &&
                          !s->trace.on && test_hook("model_kept") == nullptr;
  cc.valid = cc.model_only = false;
  // A chained fit may already be queued (or finished) behind the previous one: it is this fit if the path function
  // asked for exactly what it announced; otherwise the device state can no longer be trusted to be the previous
  // fit's result and everything is set up again from the host's copy.
  const bessx_session::Hint hint = s->hint;
  s->hint.on = false;
  bool ahead_hit = false;
  int my_buf = 0, ahead_serial = 0;
  if (s->ahead.armed) {
    s->ahead.armed = false;
    const bessx_session::Ahead mine = s->ahead;
    if (cov && use_cache && s->dev_state_rs == rs && mine.T0 == T0 && mine.lambda == lambda && mine.rs == rs &&
        !s->trace.on) {
      // keep the chain going: the fit after this one goes in before this one's result is awaited
      if (int rc = enqueue_chained(s, hint, rs, mine.serial, mine.buf ^ 1, 2, lambda, T0)) return rc;
      // this fit's result may still be a snapshot waiting for a launch to carry it (nothing was chained behind it)
      if (!s->ahead.armed)
        if (int rc = publish_flush(s)) return rc;
      if (int rc = publish_wait(s, mine.buf, mine.seq)) return rc;
      // serial mismatch: the device did not start this fit (its gate failed); the state is still the previous
      // fit's, the fit chained behind it cannot have started either
      ahead_hit = reinterpret_cast<const FitCtrl *>(s->res_h)->serial == mine.serial;
      if (!ahead_hit) {
        s->ahead.armed = false;
        s->pend_on = false;  // (the fit queued behind it cannot start: its snapshot is never asked for)
      }
      (ahead_hit ? s->chain_hits : s->chain_dead)++;
      my_buf = mine.buf;
      ahead_serial = mine.serial;
    } else {
      HIPX(hipStreamSynchronize(s->st));
      s->pend_on = false;
      s->dev_state_rs = -1;
      use_cache = false;
      model_kept = false;  // (the device state is no longer trusted: the model is set up again from the host's copy)
      s->chain_mismatch++;
    }
  }
  const int my_serial = ahead_hit ? ahead_serial : ++s->fit_serial;
  // bd is one buffer for all row sets: it still holds this row set's scores only if its previous fit was the last
  // thing the device ran (the condition of the upload-free start below)
  const bool scores_ok = cov && use_cache && s->dev_state_rs == rs && cc.cov_layout && cc.lambda == lambda;
  hipError_t e = hipSuccess;
  if (ahead_hit) {
    // nothing to queue: the first batch of this fit is running or done
  } else if ((use_cache && s->dev_state_rs == rs) || model_kept) {
    // the device still holds exactly these coefficients (previous fit of the chain): no upload, no re-initialisation
    e = launch_fit_continue(s->ctrl, T0, s->hist, s->st, my_serial, 0);
  } else {
    int *st_idx = reinterpret_cast<int *>(s->stage_h);
    double *st_val = reinterpret_cast<double *>(s->stage_h + (size_t)s->capA * sizeof(int));
    for (int i = 0; i < k_init; i++) {
      st_idx[i] = s->beta_init.idx[i];
      st_val[i] = s->beta_init.val[i];
    }
    if (k_init) {
      HIPX(hipMemcpyAsync(s->init_idx_d, st_idx, k_init * sizeof(int), hipMemcpyHostToDevice, s->st));
      HIPX(hipMemcpyAsync(s->init_val_d, st_val, k_init * sizeof(double), hipMemcpyHostToDevice, s->st));
    }
    e = launch_fit_begin(s->ctrl, T0, k_init, s->init_idx_d, s->init_val_d, s->coef0_init, s->A_cur, s->b_cur,
                         s->beta_dense, s->p, s->hist, s->st, s->inA, my_serial);
  }
  s->dev_state_rs = rs;
  if (e == hipSuccess && !use_cache && !cov) {
    if (!glm)
      e = launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, 0, s->A_cur, s->b_cur, s->r_rs[rs], s->sse,
                          s->st);
    else if (cox)
      e = launch_cox_state(s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->ctrl, 0, s->A_cur, s->b_cur, s->cox,
                           s->sse, s->st);
    else
      e = launch_glm_eta_gh(s->model_type, s->X, s->ld, s->n, s->y, s->w, s->mask[rs], s->logfact, s->ctrl, 0,
                            s->A_cur, s->b_cur, s->r_rs[rs], s->h_rs[rs], s->sse, s->st);
  }
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("fit begin: ") + hipGetErrorString(e));
  if (cov && !use_cache && !model_kept && k_init > 0) {
    // the first score pass multiplies the cached Gram columns of the initial support: form the missing ones
    bessx_session::CovCache &cv = s->cov[rs];
    e = launch_cov_need(s->A_cur, k_init, nullptr, s->bd2, s->p, cv.slot_of, cv.meta, cov_C_dev(s), s->cov_fcols, s->ctrl, 0,
                        s->A_cur, s->st);
    if (e == hipSuccess)
      e = launch_cov_fill_list(s->cov_fcols, s->cov_extras, s->bd2, cv.slot_of, cv.meta, s->ctrl, 0, s->st, s->cov_spec, 0);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov begin: ") + hipGetErrorString(e));
    if (int rc = enqueue_cov_fill(s, rs, (k_init + COV_R - 1) / COV_R, 0)) return rc;
  }

  const FitCtrl *hc = reinterpret_cast<const FitCtrl *>(s->res_h);
  int slot = 1, batch = 2;  // warm-started fits usually stop after 2 iterations
  std::vector<std::pair<size_t, bool>> k1_pairs;
  bool have_results = ahead_hit;
  if (ahead_hit) slot = 1 + std::min(batch, s->max_iter);
  while (!glm && cov) {
    if (!have_results) {
      const bool first_batch = slot == 1;
      unsigned long long seq = 0;
      PubArgs pa = {};
      if (s->publish) pa = publish_args(s, T0, my_buf, &seq);
      bool published = false;
      if (s->kch_owner && kchains_staged(s)) s->kch_gen_seen = kchains_generation(s);  // (the look-ups queued below run after these fills)
      for (int b = 0; b < batch && slot <= s->max_iter; b++, slot++) {
        SlotFuse sf;
        if (s->publish && (b + 1 == batch || slot == s->max_iter)) sf.pub = &pa;
        if (int rc = enqueue_lm_slot_cov(s, slot, T0, lambda, rs, use_cache && slot == 1, scores_ok,
                                         scores_ok && cc.T0 + 1 == T0, &sf))
          return rc;
        published = published || sf.pub_fused;
      }
      if (!s->publish) {
        if (int rc = read_results(s, T0)) return rc;
      } else {
        if (!published)
          if (int rc = publish_launch(s, pa)) return rc;
        // chain the announced next fit of the warm-start path behind this one before waiting for this one
        if (first_batch) {
          const auto tq = std::chrono::steady_clock::now();
          if (int rc = enqueue_chained(s, hint, rs, my_serial, my_buf ^ 1, batch, lambda, T0)) return rc;
          s->dbg_enq_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq).count();
        }
        if (int rc = publish_wait(s, my_buf, seq)) return rc;
      }
    }
    have_results = false;
    hc = reinterpret_cast<const FitCtrl *>(s->res_h);
    if (int rc = cov_collect(s, hc->cov_nfill)) return rc;
    // the chained fit only starts if this one ended here with fresh score sums
    if (s->ahead.armed && !(hc->done && hc->d_fresh && !hc->info)) {
      s->ahead.armed = false;
      s->pend_on = false;
    }
    if (hc->cov_stall) {
      if (int rc = cov_unpark(s, hc, T0, lambda, rs, &slot)) return rc;
      continue;
    }
    if (hc->done || slot > s->max_iter) break;
  }
  const bool cgb_fit = cov && (T0 + 1 + 15) / 16 > 16 && s->cov_cg && T0 <= CGB_MAX_K && s->cgb_work != nullptr;
  if (cov && !hc->done && (rs != 0 || ((T0 + 1 + 15) / 16 > 16 && !cgb_fit) || !s->cov_cg)) {
    // out of iterations: the sums of squares of the last coefficients have not been formed yet
    HIPX(launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, hc->l, s->A_cur, s->b_cur, s->r_rs[rs], s->sse,
                         s->st, 2));
    if (int rc = read_results(s, T0)) return rc;
    hc = reinterpret_cast<const FitCtrl *>(s->res_h);
  }
  if (cov) {
    s->cov_panel_groups += hc->cov_groups;
    if (hc->cov_miss) return fail(BESSX_ERR_NUMERIC, "internal error: an active column was missing from the Gram column cache");
    // large systems: as many conjugate-gradient step launches per solve as the last solve took, and a few
    if ((T0 + 1 + 15) / 16 > 16 && hc->irls_last > 0) s->cgb_guess = std::max(12, std::min(64, hc->irls_last + 8));
  }
  int pending_heavy = 0;  // the slot whose light part ran and found a NEW active set
  while (!glm && !cov) {
    int first = slot;
    if (s->light_confirm && s->max_iter >= 2) {
      // one whole iteration (or the rest of one whose light part did not commit) + the LIGHT part of the next one: a
      // warm-started fit changes its set in the first iteration and confirms it in the second
      std::vector<int> which;  // PDAS iteration of every pass queued in this batch
      if (pending_heavy) {
        if (int rc = enqueue_lm_slot(s, pending_heavy, T0, lambda, rs, false, k1_pairs, 2)) return rc;
        pending_heavy = 0;
      } else {
        if (int rc = enqueue_lm_slot(s, slot, T0, lambda, rs, use_cache && slot == 1, k1_pairs, 0)) return rc;
        which.push_back(slot++);
      }
      int light = 0;
      if (slot <= s->max_iter) {
        if (int rc = enqueue_lm_slot(s, slot, T0, lambda, rs, false, k1_pairs, 1)) return rc;
        light = slot;
        which.push_back(slot++);
      }
      if (int rc = read_results(s, T0)) return rc;
      // a pass ran iff its iteration was committed -- or is the light part that found a new set (l one short, not done)
      for (size_t i = 0; i < k1_pairs.size() && i < which.size(); i++)
        k1_pairs[i].second = which[i] <= hc->l || (which[i] == hc->l + 1 && which[i] == light && !hc->done);
      if (int rc = k1_collect(s, k1_pairs)) return rc;
      k1_pairs.clear();
      if (hc->done) break;
      if (light && hc->l < light) {
        pending_heavy = light;  // (also at light == max_iter: the iteration still has to be finished)
        continue;
      }
      if (slot > s->max_iter) break;
      continue;
    }
    for (int b = 0; b < batch && slot <= s->max_iter; b++, slot++)
      if (int rc = enqueue_lm_slot(s, slot, T0, lambda, rs, use_cache && slot == 1, k1_pairs, 0)) return rc;
    if (int rc = read_results(s, T0)) return rc;
    // slots first..l really ran K1; later ones fell through their gate
    for (size_t i = 0; i < k1_pairs.size(); i++) k1_pairs[i].second = (first + (int)i) <= hc->l;
    if (int rc = k1_collect(s, k1_pairs)) return rc;
    k1_pairs.clear();
    if (hc->done || slot > s->max_iter) break;
    batch = 2;
  }
  while (glm && slot <= s->max_iter) {
    // one PDAS iteration per round: the IRLS chain is enqueued in guessed batches and stops itself
    const int tmax = s->model_type == 2 ? 30 : (cox ? 30 : 50);
    if (cox) {
      if (int rc = enqueue_cox_head(s, slot, T0, lambda, rs, use_cache && slot == 1, k1_pairs)) return rc;
    } else {
      if (int rc = enqueue_glm_head(s, slot, T0, lambda, rs, use_cache && slot == 1, k1_pairs)) return rc;
    }
    int t = cox ? 1 : 0, steps_used = 0;  // IRLS steps are numbered from 0, Newton steps from 1 (:1411)
    // Round 6, the CONFIRMING iteration without its sub-model chain.  From the second PDAS iteration of a fit on, the new
    // active set usually equals the previous one (the normal end of a fit, src/Algorithm.h:164-170; logistic and Cox
    // re-fits of a repeated set reproduce the coefficients: same_prev) -- and the guessed batch of IRLS / Newton steps
    // queued behind the head then falls through launch by launch: ~20 launches per logistic iteration, ~150 per Cox
    // iteration, each a few microseconds of the command processor that the OTHER chains' kernels wait behind (the device
    // advances about 2.5 chains of small kernels at a time).  So the tail goes in right behind the head: if the set
    // repeated it commits (k_commit's record-and-stop branch) and the iteration is over after 2 launches; if not, it
    // falls through (the chain has not run: irls_done = 0), one host round trip is spent, and the steps follow as before.
    // (Poisson has no repeated-set shortcut -- its IRLS restarts from the updated intercept.)
    if (slot >= 2 && (cox || s->model_type == 2) && s->light_confirm) {
      if (int rc = cox ? enqueue_cox_tail(s, slot, T0, rs) : enqueue_glm_tail(s, slot, T0, rs)) return rc;
      if (int rc = read_results(s, T0)) return rc;
      if (hc->l == slot) {
        for (size_t i = 0; i < k1_pairs.size(); i++) k1_pairs[i].second = true;
        if (int rc = k1_collect(s, k1_pairs)) return rc;
        k1_pairs.clear();
        slot++;
        if (hc->done) break;
        continue;
      }
    }
    while (true) {
      int upto = std::min(tmax, t + std::max(2, s->irls_guess) - 1);
      for (; t <= upto; t++)
        if (int rc = cox ? enqueue_cox_newton(s, slot, t, T0, lambda, rs) : enqueue_glm_irls_step(s, slot, t, T0, lambda, rs))
          return rc;
      if (int rc = cox ? enqueue_cox_tail(s, slot, T0, rs) : enqueue_glm_tail(s, slot, T0, rs)) return rc;
      if (int rc = read_results(s, T0)) return rc;
      if (hc->l == slot) {  // committed (IRLS finished, or the active set repeated)
        steps_used = hc->irls_last;
        break;
      }
      if (t > tmax) return fail(BESSX_ERR_NUMERIC, "IRLS chain did not terminate");
    }
    for (size_t i = 0; i < k1_pairs.size(); i++) k1_pairs[i].second = true;
    if (int rc = k1_collect(s, k1_pairs)) return rc;
    k1_pairs.clear();
    if (steps_used > 0) s->irls_guess = std::min(tmax + 1, steps_used + 1);
    s->n_submodel_steps += steps_used;
    slot++;
    if (hc->done) break;
  }
  if (hc->info == 2 && glm && !cox && !s->glm_fallback) {
    // an IRLS system of this fit was rank-deficient to working precision (exactly dependent active columns) and its
    // k_chol stood back: from now on the chain carries the pivoted solve behind every k_chol (a fall-through launch
    // per step that sessions without such data never pay), and this fit is redone with it
    s->glm_fallback = true;
    HIPX(hipStreamSynchronize(s->st));
    HIPX(hipMemsetAsync(&s->ctrl->info, 0, sizeof(int), s->st));
    s->cache[rs].valid = s->cache[rs].model_only = false;
    s->dev_state_rs = -1;
    return algorithm_fit(s);
  }
  if (hc->info) return fail(BESSX_ERR_NUMERIC, "non-finite value in the k x k solve (singular Gram matrix?)");
  // results
  const double *sse_h = reinterpret_cast<const double *>(s->res_h + ((unsigned char *)s->sse - s->resblk));
  const double *b_h = reinterpret_cast<const double *>(s->res_h + ((unsigned char *)s->b_cur - s->resblk));
  const int *a_h = reinterpret_cast<const int *>(s->res_h + ((unsigned char *)s->A_cur - s->resblk));
  s->beta.idx.assign(a_h, a_h + T0);
  s->beta.val.assign(b_h, b_h + T0);
  s->coef0 = hc->coef0;
  s->l = hc->done ? hc->l : s->max_iter + 1;
  double tr = 0.0, te = 0.0;
  const int mt_fit = (T0 + 1 + 15) / 16;
  if (cov && rs == 0 && (mt_fit <= 16 || cgb_fit) && s->cov_cg) {
    // all rows, covariance form, solve by k_cg: no residual was formed.  |y - X beta|^2 = y.y - beta.(q + rho) -
    // lambda |beta|^2 with rho the residual of the normal equations (k_cg hands over both dot products, and clears
    // sse_valid when the cancellation is not harmless).  If the last solve came from the Cholesky fallback, or the
    // difference cancels badly (near-perfect or ill-conditioned fit), one pass over the active columns with the
    // final coefficients gives the sum directly.
    const double yy = s->yy_h[0];
    tr = yy - hc->sse_dot - lambda * hc->sse_nrm;
    if (!hc->sse_valid || !(tr > 1e-6 * yy)) {
      HIPX(hipStreamSynchronize(s->st));  // a chained fit may be running on the device state: use the host's copy
      int *st_idx = reinterpret_cast<int *>(s->stage_h);
      double *st_val = reinterpret_cast<double *>(s->stage_h + (size_t)s->capA * sizeof(int));
      for (int i = 0; i < T0; i++) {
        st_idx[i] = s->beta.idx[i];
        st_val[i] = s->beta.val[i];
      }
      HIPX(hipMemcpy(s->init_idx_d, st_idx, T0 * sizeof(int), hipMemcpyHostToDevice));
      HIPX(hipMemcpy(s->init_val_d, st_val, T0 * sizeof(double), hipMemcpyHostToDevice));
      HIPX(launch_resid_lm(s->X, s->ld, s->n, s->y, s->mask[rs], s->ctrl, 0, s->init_idx_d, s->init_val_d, s->tmpv,
                           s->sse, s->st, 3, T0, s->coef0));
      std::vector<double> part((size_t)2 * s->n_sse_blk);
      HIPX(hipMemcpyAsync(part.data(), s->sse, part.size() * sizeof(double), hipMemcpyDeviceToHost, s->st));
      HIPX(hipStreamSynchronize(s->st));
      tr = 0.0;
      for (int b = 0; b < s->n_sse_blk; b++) tr += part[2 * b];
    }
  } else {
    for (int b = 0; b < s->n_sse_blk; b++) {
      tr += sse_h[2 * b];
      te += sse_h[2 * b + 1];
    }
  }
  s->sse_train = tr;
  s->sse_test = te;
  if (test_hook("fit_log"))
    std::fprintf(stderr, "[bessx] fit: level %d, %d iterations, started %s, scores kept %d, ended done=%d d_fresh=%d info=%d, slots queued %d\n", T0,
                 hc->l, ahead_hit ? "chained on the device" : (use_cache && s->dev_state_rs == rs ? "by k_fit_continue" : (model_kept ? "by k_fit_continue, scores formed again" : "from the host's copy")),
                 scores_ok ? 1 : 0, hc->done, hc->d_fresh, hc->info, slot - 1);
  cc.valid = hc->done && hc->d_fresh;
  cc.model_only = cov && hc->done && !hc->d_fresh && !hc->info && !hc->cov_miss;
  cc.cov_layout = cov;
  cc.lambda = lambda;
  cc.T0 = T0;
  cc.beta = s->beta;
  cc.coef0 = s->coef0;
  s->n_fits += 1;
  s->n_iters += hc->l;
  if (s->trace.on) {
    const int L = hc->l;
    std::vector<int> hh((size_t)(L + 1) * s->hist_stride);
    std::vector<double> hb((size_t)(L + 1) * s->hist_stride), hc0(L + 1);
    HIPX(hipMemcpy(hh.data(), s->hist, hh.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPX(hipMemcpy(hb.data(), s->hist_beta, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPX(hipMemcpy(hc0.data(), s->hist_coef0, hc0.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int it = 1; it <= L; it++) {
      s->trace.meta.push_back(it);
      s->trace.meta.push_back(T0);
      s->trace.meta.push_back(s->n_train[rs]);
      s->trace.meta.push_back((int)s->trace.a_flat.size());
      for (int i = 0; i < T0; i++) {
        s->trace.a_flat.push_back(hh[(size_t)it * s->hist_stride + i]);
        s->trace.beta_flat.push_back(hb[(size_t)it * s->hist_stride + i]);
      }
      s->trace.coef0_calls.push_back(hc0[it]);
    }
  }
  return 0;
}


}  // namespace bessx

