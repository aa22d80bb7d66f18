// bessx_cv.cpp -- Metric (src/Metric.h): loss / IC formulas and test_loss, the K fold fits of a CV evaluation side by side
#include "bessx_host.h"

namespace bessx {

// --------------------------------------------------------------------------------------------
// The K fold fits of a CV evaluation side by side (Metric::test_loss, src/Metric.h:150-195; see
// bessx_session::fold_ctx).  Every fold context runs the same slots algorithm_fit() would queue for it -- warm start
// from the fold's previous coefficients (:177-188), batches of two PDAS iterations, the parked-fit protocol -- on its
// own stream; the host enqueues a batch for every chain, waits for all of them, and serves every chain that is parked
// on missing Gram columns with ONE fill while all chains are quiet.
// --------------------------------------------------------------------------------------------
void fold_contexts_invalidate(bessx_session *s) {
  for (bessx_session *c : s->fold_ctx) {
    for (auto &cc : c->cache) cc.valid = cc.model_only = false;
    c->dev_state_rs = -1;
  }
}

bool side_by_side_applies(const bessx_session *s, int T0) {
  if (s->parent || s->K < 1 || s->fold_ctx.size() != (size_t)s->K) return false;
  if (s->trace.on || s->model_type != 1 || s->grouped || !s->cov_mode || !s->cv_shared) return false;
  if (T0 < 1 || T0 > s->cap || (T0 + 1 + 15) / 16 > 16) return false;  // (the fused selection + solve launches)
  if (!topk_supported(s->p, T0) || !topk_can_fuse_need(s->p) || !sel_cgr_applies(s->p, T0)) return false;
  // every chain's sets -- the support it starts from AND the one it is heading for -- must fit a cache that has just
  // been started over, together, and the list of one fill its buffer
  long need = 0;
  for (int k = 0; k < s->K; k++)
    need += std::max(T0, s->warm_start ? (int)s->cv_init[k].idx.size() : (int)s->beta_init.idx.size());
  if (need + s->cov_spec + COV_R > (long)cov_C_dev(s)) return false;
  if (need + 2 * s->cov_spec + COV_R > (long)s->capA + 4 * COV_R) return false;
  return true;
}

// `only`: the folds to fit (ascending; nullptr = all K) -- a rank of a fold-sharded CV path fits its own folds
// (bessx_session_cv_eval); `per_fold`: test loss of every fitted fold, in the order of `only`.  *out = their mean.
int fold_fits_side_by_side(bessx_session *s, double *out, const std::vector<int> *only,
                                  double *per_fold) {
  const int K = s->K, T0 = s->sparsity_level, p = s->p;
  const double lambda = s->lambda_level;
  enum Todo { NONE, START, RESUME, UNPARK };
  struct Chain {
    bessx_session *c = nullptr;
    int rs = 0, slot = 1, k_init = 0, serial = 0;
    int prev_T0 = 0;  // sparsity level of the fit whose state the chain's device buffers hold (use_cache)
    bool use_cache = false, scores_ok = false, grow1 = false, active = true, wait_fill = false;
    Todo todo = START;
    unsigned long long seq = 0;
    const FitCtrl *hc = nullptr;
    int rc = 0;
    std::string err;
  };
  std::vector<Chain> ch((size_t)K);
  std::vector<char> sel((size_t)K, only ? 0 : 1);
  if (only)
    for (int k : *only) sel[(size_t)k] = 1;
  int nsel = 0;
  for (char v : sel) nsel += v;
  auto quiet = [&]() {
    for (bessx_session *c : s->fold_ctx) (void)hipStreamSynchronize(c->st);
    (void)hipStreamSynchronize(s->st);
  };
#define SBS(expr)                  \
  do {                             \
    int rc__ = (expr);             \
    if (rc__) {                    \
      std::string keep__ = g_err;  \
      quiet();                     \
      fold_contexts_invalidate(s); \
      g_err = keep__;              \
      return rc__;                 \
    }                              \
  } while (0)
#define SBSH(expr)                                                                                  \
  do {                                                                                              \
    hipError_t e__ = (expr);                                                                        \
    if (e__ != hipSuccess)                                                                          \
      SBS(fail(BESSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)));                 \
  } while (0)
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tick = [&](int which, std::chrono::steady_clock::time_point &t0) {
    const auto t1 = tnow();
    s->sbs_t[which] += std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
  };
  auto tm = tnow();
  for (int r = 1; r <= K; r++) s->cache[r].valid = s->cache[r].model_only = false;  // the fold row sets' state now lives in the contexts
  if (s->dev_state_rs > 0) s->dev_state_rs = -1;
  // ---- start of the K fits: Algorithm::fit up to its first iteration (src/Algorithm.h:147-160)
  // the opening of one chain's fit (its own stream): warm start from the device state or from the uploaded support
  auto open_fit = [&](Chain &q) -> int {
    bessx_session *c = q.c;
    if (q.use_cache) {
      HIPX(launch_fit_continue(c->ctrl, T0, c->hist, c->st, q.serial, 0));
      return 0;
    }
    int *st_idx = reinterpret_cast<int *>(c->stage_h);
    double *st_val = reinterpret_cast<double *>(c->stage_h + (size_t)c->capA * sizeof(int));
    for (int i = 0; i < q.k_init; i++) {
      st_idx[i] = c->beta_init.idx[i];
      st_val[i] = c->beta_init.val[i];
    }
    if (q.k_init) {
      HIPX(hipMemcpyAsync(c->init_idx_d, st_idx, q.k_init * sizeof(int), hipMemcpyHostToDevice, c->st));
      HIPX(hipMemcpyAsync(c->init_val_d, st_val, q.k_init * sizeof(double), hipMemcpyHostToDevice, c->st));
    }
    HIPX(launch_fit_begin(c->ctrl, T0, q.k_init, c->init_idx_d, c->init_val_d, c->coef0_init, c->A_cur, c->b_cur,
                          c->beta_dense, p, c->hist, c->st, c->inA));
    return 0;
  };
  std::vector<int> openers;
  for (int k = 0; k < K; k++) {
    Chain &q = ch[k];
    bessx_session *c = q.c = s->fold_ctx[k];
    const int rs = q.rs = k + 1;
    if (!sel[k]) {
      q.active = false;
      continue;
    }
    c->sparsity_level = T0;
    c->lambda_level = lambda;
    c->cur_rows = rs;
    c->beta_init = s->warm_start ? s->cv_init[k] : s->beta_init;  // update_beta_init(cv_initial_model_param.row(k))
    c->coef0_init = s->coef0_init;
    q.k_init = (int)c->beta_init.idx.size();
    if (q.k_init > c->cap) SBS(fail(BESSX_ERR_ARG, "initial support too large"));
    bessx_session::RsCache &cc = c->cache[rs];
    q.use_cache = cc.valid && cc.coef0 == c->coef0_init && cc.beta.idx == c->beta_init.idx &&
                  cc.beta.val == c->beta_init.val && cc.cov_layout && c->dev_state_rs == rs;
    cc.valid = cc.model_only = false;
    q.prev_T0 = cc.T0;
    q.serial = ++c->fit_serial;
    q.scores_ok = q.use_cache && cc.lambda == lambda;
    q.grow1 = q.scores_ok && cc.T0 + 1 == T0;
    c->dev_state_rs = rs;
    if (!q.use_cache && q.k_init > 0) {
      SBS(open_fit(q));  // (fit_begin leaves the initial support in A_cur)
      q.todo = NONE;
      openers.push_back(k);
    }
  }
  if (!openers.empty()) {
    // The first score pass of a chain that starts from an uploaded support multiplies the cached Gram columns of that
    // support: form the missing ones -- for ALL such chains in ONE fill on the shared slot map, before any chain reads
    // it.  Whether the cache has to be started over is decided once, by the fill's own list kernel, from the cache's
    // occupancy and the columns these supports miss; when it is, the chains that continue from their device state get
    // their current columns back in the same fill (their next selection only looks the ENTERING columns up).  (Until round 3 every chain ran its own
    // slot-0 lookup one after another, and a later chain's restart could evict what an earlier one had just filled.)
    for (int k : openers) SBSH(hipStreamSynchronize(ch[k].c->st));
    CovUnion u = {};
    long ub = 0;
    for (int k = 0; k < K; k++) {
      Chain &q = ch[k];
      if (!q.active) continue;
      const bool opener = !q.use_cache && q.k_init > 0;
      if (opener || (q.use_cache && q.prev_T0 > 0)) {
        u.list[u.nf] = q.c->A_cur;
        u.len[u.nf] = opener ? q.k_init : q.prev_T0;
        u.on_restart[u.nf++] = opener ? 0 : 1;  // (a chain that continues from its device state: cached unless started over)
        ub += opener ? q.k_init : q.prev_T0;
      }
    }
    // (restart = 2: the kernel starts the cache over iff the openers' missing columns do not fit what is left)
    SBSH(launch_cov_fill_union(u, 2, nullptr, nullptr, s->cov_spec, 0, s->cov[0].slot_of, s->cov[0].meta, p, s->cov_fcols,
                               s->fill_ctrl, s->st, cov_C_dev(s)));
    SBSH(hipMemcpyAsync(s->fill_ctrl_h, s->fill_ctrl, sizeof(FitCtrl), hipMemcpyDeviceToHost, s->st));
    SBSH(hipStreamSynchronize(s->st));
    s->cov_panel_groups += s->fill_ctrl_h->cov_groups - s->fill_groups_seen;
    s->fill_groups_seen = s->fill_ctrl_h->cov_groups;
    const int ngroups = s->fill_ctrl_h->cov_nfill / COV_R;
    if (ngroups > (ub + COV_R - 1) / COV_R) SBS(fail(BESSX_ERR_NUMERIC, "internal error: opening fill list longer than its bound"));
    if (ngroups > 0) {
      SBS(enqueue_cov_fill(s, 0, ngroups, 1, s->fill_ctrl));
      SBSH(hipEventRecord(s->ev_fill, s->st));
      s->cv_union_fills++;
      for (Chain &q : ch) q.wait_fill = q.active;
      if (s->timing) {
        SBSH(hipStreamSynchronize(s->st));
        SBS(cov_collect(s, s->fill_ctrl_h->cov_nfill));
      }
    }
  }
  if (!s->fold_pool) {
    s->fold_pool = new FoldPool();
    {
      const int dev = s->device;
      s->fold_pool->start(K - 1, [dev] { (void)hipSetDevice(dev); });
    }
  }
  tick(0, tm);
  // ---- lock-step rounds
  // one chain's share of a round, on its own stream (runs on its own host thread): what the previous read-back asked
  // for (wake a parked fit up and finish its slot), then the next batch of two PDAS iterations and the publication
  auto chain_round = [&](int k) {
    Chain &q = ch[k];
    if (!q.active) return;
    bessx_session *c = q.c;
    auto body = [&]() -> int {
      if (q.wait_fill) HIPX(hipStreamWaitEvent(c->st, s->ev_fill, 0));  // nobody reads the caches before the fill is in
      q.wait_fill = false;
      if (q.todo == START) {
        if (int rc = open_fit(q)) return rc;
      } else if (q.todo == RESUME) {
        const int stalled = -1 - q.hc->l + 1;
        HIPX(launch_cov_resume(c->ctrl, c->st));
        if (int rc = enqueue_cov_tail(c, stalled, T0, lambda, q.rs)) return rc;
        q.slot = stalled + 1;
      } else if (q.todo == UNPARK) {
        if (int rc = cov_unpark(c, q.hc, T0, lambda, q.rs, &q.slot)) return rc;  // 2: Cholesky for the slot; 3: the exact tie rule
      }
      q.todo = NONE;
      for (int b = 0; b < 2 && q.slot <= c->max_iter; b++, q.slot++)
        if (int rc = enqueue_lm_slot_cov(c, q.slot, T0, lambda, q.rs, q.use_cache && q.slot == 1, q.scores_ok, q.grow1, nullptr))
          return rc;
      return publish_enqueue(c, T0, 0, &q.seq);  // the result block goes to pinned memory by a kernel of the chain
    };
    q.rc = body();
    if (q.rc) q.err = g_err;  // (the message is thread-local)
  };
  int remaining = nsel;
  while (remaining > 0) {
    s->cv_rounds++;
    if (!s->fold_pool->run(chain_round, s->wait_deadline_s))
      SBS(fail(BESSX_ERR_HIP, "the host threads of the fold chains did not return from queueing their launches within " +
                                  std::to_string(s->wait_deadline_s) + " s (BESSX_WAIT_TIMEOUT_S) -- the session can only "
                                  "be destroyed now"));
    for (Chain &q : ch)
      if (q.active && q.rc) {
        g_err = q.err;
        SBS(q.rc);
      }
    tick(1, tm);
    for (Chain &q : ch) {
      if (!q.active) continue;
      SBS(publish_wait(q.c, 0, q.seq));
      q.hc = reinterpret_cast<const FitCtrl *>(q.c->res_h);
    }
    tick(2, tm);
    // chains parked on missing columns (1) or on a full cache (4): one fill for all of them, now that every chain is quiet
    bool filled = false;
    {
      int n_parked = 0, sum_nm = 0;
      bool any_full = false;
      Chain *spec_src = nullptr;
      for (Chain &q : ch) {
        if (!q.active) continue;
        const int stl = q.hc->cov_stall;
        if (stl != 1 && stl != 4) continue;
        n_parked++;
        any_full = any_full || stl == 4;
        if (stl == 1) {
          sum_nm += q.hc->cov_nmiss;
          if (!spec_src && cov_speculates(q.c)) spec_src = &q;
        }
      }
      if (n_parked > 0) {
        int meta_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        SBSH(hipMemcpyAsync(meta_h, s->cov[0].meta, sizeof(meta_h), hipMemcpyDeviceToHost, s->st));
        SBSH(hipStreamSynchronize(s->st));
        const bool restart = any_full || meta_h[0] + sum_nm + s->cov_spec + COV_R > cov_C_dev(s);
        CovUnion u = {};
        int ub = 0;
        for (Chain &q : ch) {
          if (!q.active) continue;
          const int stl = q.hc->cov_stall;
          if (stl == 1 || stl == 4 || (restart && stl == 2)) {
            u.list[u.nf] = q.c->A_new;  // the set this chain's parked slot is about to solve on
            u.len[u.nf++] = T0;
            ub += (restart || stl != 1) ? T0 : q.hc->cov_nmiss;
          } else if (restart && stl == 0 && !q.hc->done && q.slot <= q.c->max_iter) {
            u.list[u.nf] = q.c->A_cur;  // in the middle of a fit: its next score pass multiplies these columns
            u.len[u.nf++] = T0;
            ub += T0;
          }
        }
        const int spec_min = s->cov_spec / 2;
        if (spec_src) {
          bessx_session *cs = spec_src->c;
          SBSH(launch_topk(cs->bd2, p, s->cov_spec, cs->cov_extras, cs->cand, nullptr, 0, cs->st));
          SBSH(hipEventRecord(s->ev_ctx, cs->st));
          SBSH(hipStreamWaitEvent(s->st, s->ev_ctx, 0));
        }
        SBSH(launch_cov_fill_union(u, restart ? 1 : 0, spec_src ? spec_src->c->cov_extras : nullptr,
                                   spec_src ? spec_src->c->bd2 : nullptr, s->cov_spec, spec_min, s->cov[0].slot_of, s->cov[0].meta, p,
                                   s->cov_fcols, s->fill_ctrl, s->st));
        // the list's real length (columns two folds miss are listed once) decides how many groups are formed: the
        // pair kernel the host would pick for two groups costs 1.8 passes even when the second group is empty
        SBSH(hipMemcpyAsync(s->fill_ctrl_h, s->fill_ctrl, sizeof(FitCtrl), hipMemcpyDeviceToHost, s->st));
        SBSH(hipStreamSynchronize(s->st));
        s->cov_panel_groups += s->fill_ctrl_h->cov_groups - s->fill_groups_seen;
        s->fill_groups_seen = s->fill_ctrl_h->cov_groups;
        const int ngroups = s->fill_ctrl_h->cov_nfill / COV_R;
        if (const char *ev = std::getenv("BESSX_DEBUG"))
          if (std::atoi(ev) >= 2)
            std::fprintf(stderr, "[sbs] union fill: %d chains parked, sum of their missing columns %d, list %d columns, "
                         "%d cached after it%s\n", n_parked, sum_nm, s->fill_ctrl_h->cov_nfill, s->fill_ctrl_h->k_cur,
                         restart ? " (cache started over)" : "");
        if (ngroups > (ub + (spec_src ? s->cov_spec : 0) + 2 * COV_R - 1) / COV_R)
          SBS(fail(BESSX_ERR_NUMERIC, "internal error: union fill list longer than its bound"));
        SBS(enqueue_cov_fill(s, 0, ngroups, 1, s->fill_ctrl));
        SBSH(hipEventRecord(s->ev_fill, s->st));
        s->cv_union_fills++;
        filled = true;
        if (s->timing) {
          SBSH(hipStreamSynchronize(s->st));
          SBS(cov_collect(s, s->fill_ctrl_h->cov_nfill));
        }
      }
    }
    tick(3, tm);
    for (Chain &q : ch) {
      if (!q.active) continue;
      const FitCtrl *hc = q.hc;
      q.wait_fill = filled;
      if (hc->cov_stall == 1 || hc->cov_stall == 4) {
        q.todo = RESUME;
      } else if (hc->cov_stall) {
        q.todo = UNPARK;
      } else if (hc->done || q.slot > q.c->max_iter) {
        q.active = false;
        remaining--;
      }
    }
    tick(4, tm);
    static const bool verbose = [] {
      const char *ev = std::getenv("BESSX_DEBUG");
      return ev && std::atoi(ev) >= 2;
    }();
    if (verbose) {
      static double last[6] = {0, 0, 0, 0, 0, 0};
      int act = 0;
      for (Chain &q : ch) act += q.active ? 1 : 0;
      std::fprintf(stderr, "[sbs] T0 %d round: enqueue %.0f us, wait %.0f, fill %.0f (%s), still active %d\n", T0,
                   (s->sbs_t[1] - last[1]) * 1e6, (s->sbs_t[2] - last[2]) * 1e6, (s->sbs_t[3] - last[3]) * 1e6,
                   filled ? "union fill" : "-", act);
      for (int i = 0; i < 6; i++) last[i] = s->sbs_t[i];
    }
  }
  // ---- results (the tail of algorithm_fit), in fold order
  int k_last = K - 1;
  while (k_last > 0 && !sel[k_last]) k_last--;
  if (s->warm_start) s->beta_init = s->cv_init[k_last];  // (update_beta_init of the last fold: its warm start, not its result)
  double acc = 0.0;
  int n_out = 0;
  for (int k = 0; k < K; k++) {
    if (!sel[k]) continue;
    Chain &q = ch[k];
    bessx_session *c = q.c;
    const FitCtrl *hc = q.hc;
    if (!hc->done) {
      // out of iterations: the sums of squares of the last coefficients have not been formed yet
      SBSH(launch_resid_lm(c->X, c->ld, c->n, c->y, c->mask[q.rs], c->ctrl, hc->l, c->A_cur, c->b_cur, c->r_rs[q.rs], c->sse,
                           c->st, 2));
      SBS(read_results(c, T0));
      hc = reinterpret_cast<const FitCtrl *>(c->res_h);
    }
    s->cov_panel_groups += hc->cov_groups;
    if (hc->cov_miss)
      SBS(fail(BESSX_ERR_NUMERIC, "internal error: an active column was missing from the Gram column cache"));
    if (hc->info) SBS(fail(BESSX_ERR_NUMERIC, "non-finite value in the k x k solve (singular Gram matrix?)"));
    const double *sse_h = reinterpret_cast<const double *>(c->res_h + ((unsigned char *)c->sse - c->resblk));
    const double *b_h = reinterpret_cast<const double *>(c->res_h + ((unsigned char *)c->b_cur - c->resblk));
    const int *a_h = reinterpret_cast<const int *>(c->res_h + ((unsigned char *)c->A_cur - c->resblk));
    c->beta.idx.assign(a_h, a_h + T0);
    c->beta.val.assign(b_h, b_h + T0);
    c->coef0 = hc->coef0;
    c->l = hc->done ? hc->l : c->max_iter + 1;
    double tr = 0.0, te = 0.0;
    for (int b = 0; b < c->n_sse_blk; b++) {
      tr += sse_h[2 * b];
      te += sse_h[2 * b + 1];
    }
    c->sse_train = tr;
    c->sse_test = te;
    bessx_session::RsCache &cc = c->cache[q.rs];
    cc.valid = hc->done && hc->d_fresh;
    cc.cov_layout = true;
    cc.lambda = lambda;
    cc.T0 = T0;
    cc.beta = c->beta;
    cc.coef0 = c->coef0;
    s->n_fits += 1;
    s->n_iters += hc->l;
    s->cov_cg_fallbacks += c->cov_cg_fallbacks;
    s->cov_tie_rescues += c->cov_tie_rescues;
    c->cov_cg_fallbacks = c->cov_tie_rescues = 0;
    if (s->warm_start) s->cv_init[k] = c->beta;
    const double tl = c->sse_test / (double)(2 * s->n_test[k]);  // src/Metric.h:190
    if (per_fold) per_fold[n_out] = tl;
    n_out++;
    acc += tl;
  }
  // what Algorithm holds after the loop of test_loss: the LAST fold's fit (path.cpp reads it, :314-319)
  const bessx_session *last = s->fold_ctx[k_last];
  s->beta = last->beta;
  s->coef0 = last->coef0;
  s->l = last->l;
  s->sse_train = last->sse_train;
  s->sse_test = last->sse_test;
  s->cur_rows = k_last + 1;
  *out = acc / (double)nsel;
  tick(5, tm);
#undef SBS
#undef SBSH
  return 0;
}

// --------------------------------------------------------------------------------------------
// Metric (src/Metric.h).  Values come from sums the residual kernel already produced.
// --------------------------------------------------------------------------------------------
double metric_train_loss_value(const bessx_session *s) {
  // LmMetric::train_loss, src/Metric.h:145-148: ||y - X beta||^2 / n on ALL rows (train + test rows of the mask)
  if (s->model_type == 1) return (s->sse_train + s->sse_test) / (double)s->n;
  // Logistic / Poisson / Cox train_loss, src/Metric.h:266-290, :426-440, :565-568: -2 * (sum over ALL rows),
  // the sum being kept in sse_train
  return -2.0 * s->sse_train;
}

double metric_fold_test_loss(const bessx_session *s, int k) {
  if (s->model_type == 1) return s->sse_test / (double)(2 * s->n_test[k]);  // src/Metric.h:190
  if (s->model_type == 2 || s->model_type == 4) return -2.0 * s->sse_test;  // :349-351 (clamp +-25), :609 Cox
  return -s->sse_test;                                                      // :489 Poisson
}

int metric_train_loss(bessx_session *s, double *out) {
  *out = metric_train_loss_value(s);
  if (s->metric_depth == 0 && s->trace.on) s->trace.loss_calls.push_back(*out);
  return 0;
}

// test_loss under CV: K fold fits, src/Metric.h:150-195
int metric_test_loss(bessx_session *s, double *out) {
  if (side_by_side_applies(s, s->sparsity_level)) return fold_fits_side_by_side(s, out);
  double acc = 0.0;
  for (int k = 0; k < s->K; k++) {
    if (s->warm_start) s->beta_init = s->cv_init[k];  // update_beta_init(cv_initial_model_param.row(k))
    s->cur_rows = k + 1;                               // update_train_mask + update_group_XTX
    if (int rc = algorithm_fit(s)) return rc;
    if (s->warm_start) s->cv_init[k] = s->beta;
    acc += metric_fold_test_loss(s, k);
  }
  *out = acc / (double)s->K;
  return 0;
}

// ic: src/Metric.h:197-256 (LM)
int metric_ic(bessx_session *s, int ic_type, int is_cv, double *out) {
  s->metric_depth++;
  int rc = 0;
  if (is_cv) {
    rc = metric_test_loss(s, out);
  } else {
    // LM picks the group formula by algorithm_type (src/Metric.h:205,230), the others by g_index.size() == p
    // (:365, :504, :624); the group formula uses log(g_num) and group_df = the sparsity level
    const bool gf = s->model_type == 1 ? !(s->algorithm_type == 1 || s->algorithm_type == 5) : (s->N != s->p);
    double n = (double)s->n, p = gf ? (double)s->N : (double)s->p, c = 0.0, loss = metric_train_loss_value(s);
    if (ic_type == 1) c = 2.0;
    if (ic_type == 2) c = std::log(n);
    if (ic_type == 3) c = std::log(p) * std::log(std::log(n));
    if (ic_type == 4) c = std::log(n) + 2.0 * std::log(p);
    // LM: n log(loss) + c T0 (src/Metric.h:205-229); the other families: loss + c T0 (:365-389, :504-528, :624-648)
    const double base = s->model_type == 1 ? n * std::log(loss) : loss;
    *out = (ic_type >= 1 && ic_type <= 4) ? base + c * (double)s->sparsity_level : 0.0;
  }
  s->metric_depth--;
  if (rc == 0 && s->metric_depth == 0 && s->trace.on) s->trace.ic_calls.push_back(*out);
  return rc;
}


}  // namespace bessx

