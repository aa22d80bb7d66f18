// bessx_session.cpp -- the session: what bessCpp builds (Data + Algorithm + Metric, src/bess.cpp:61-165) resident in HBM --
// allocation, upload and normalisation, row sets and CV folds, the caches' life cycle, getters
#include "bessx_host.h"

namespace bessx {

thread_local std::string g_err;
thread_local int g_marginal_fit_variant = 0;



// A fold context (bessx_session::fold_ctx) owns only what a fit writes; everything else is the parent's.
void fold_ctx_free(bessx_session *c) {
  if (!c) return;
  if (c->st) (void)hipStreamSynchronize(c->st);
  void *dev[] = {c->resblk, c->bd, c->bd2, c->beta_dense, c->inA, c->cov_bmm, c->sol, c->A_new, c->cand, c->tie_buf,
                 c->fb_work, c->hist, c->hist_beta, c->hist_coef0, c->Gt, c->init_idx_d, c->init_val_d, c->cov_fcols,
                 c->cov_extras, c->rdiag, c->zbig};
  for (void *q : dev)
    if (q) (void)hipFree(q);
  if (c->res_buf[0]) (void)hipHostFree(c->res_buf[0]);
  if (c->pub_flag) (void)hipHostFree(c->pub_flag);
  if (c->stage_h) (void)hipHostFree(c->stage_h);
  if (c->st) ctx_stream_destroy(c->st);
  delete c;
}

void drop_fold_contexts(bessx_session *s) {
  if (s->fold_pool) {
    s->fold_pool->stop();
    if (!s->fold_pool->broken) delete s->fold_pool;  // (a broken pool's threads may still touch it: leaked on purpose)
    s->fold_pool = nullptr;
  }
  for (bessx_session *c : s->fold_ctx) fold_ctx_free(c);
  s->fold_ctx.clear();
  if (s->fill_ctrl) (void)hipFree(s->fill_ctrl);
  if (s->fill_ctrl_h) (void)hipHostFree(s->fill_ctrl_h);
  if (s->ev_fill) (void)hipEventDestroy(s->ev_fill);
  if (s->ev_ctx) (void)hipEventDestroy(s->ev_ctx);
  s->fill_ctrl = s->fill_ctrl_h = nullptr;
  s->ev_fill = s->ev_ctx = nullptr;
  s->fill_groups_seen = 0;
}

void session_free(bessx_session *s) {
  if (!s) return;
  if (std::getenv("BESSX_DEBUG") && !s->fold_ctx.empty())
    std::fprintf(stderr, "[bessx] fold chains side by side: %lld rounds, %lld union fills; ms in start %.2f, enqueue %.2f, "
                 "wait %.2f, fill %.2f, continue %.2f, results %.2f\n", s->cv_rounds, s->cv_union_fills, s->sbs_t[0] * 1e3,
                 s->sbs_t[1] * 1e3, s->sbs_t[2] * 1e3, s->sbs_t[3] * 1e3, s->sbs_t[4] * 1e3, s->sbs_t[5] * 1e3);
  drop_fold_contexts(s);
  if (std::getenv("BESSX_DEBUG") && s->kch_paths)
    std::fprintf(stderr, "[bessx] chunk chains: %lld paths, %lld stitch refits, %lld fills in the chunk phase; ms per path in "
                 "the coarse chain %.2f, the chunks %.2f, the stitch %.2f\n", s->kch_paths, s->kch_refits, s->kch_chunk_fills,
                 1e3 * s->kch_t[0] / s->kch_paths, 1e3 * s->kch_t[1] / s->kch_paths, 1e3 * s->kch_t[2] / s->kch_paths);
  kchains_free(s);
  if (std::getenv("BESSX_DEBUG"))
    std::fprintf(stderr, "[bessx] chained fits: queued %lld, used %lld, not started %lld, mismatched %lld; "
                 "CG solves handed to Cholesky: %lld; waits for a published block: %lld, of which the block was "
                 "already there %lld; queueing chained fits took %.3f ms\n",
                 s->chain_queued, s->chain_hits, s->chain_dead, s->chain_mismatch, s->cov_cg_fallbacks, s->dbg_waits,
                 s->dbg_waits_ready, s->dbg_enq_s * 1e3);
  (void)hipSetDevice(s->device);
  if (s->st) (void)hipStreamSynchronize(s->st);
  auto F = [](void *q) {
    if (q) (void)hipFree(q);
  };
  F(s->X);
  F(s->y);
  F(s->w);
  F(s->aux);
  F(s->x_mean);
  F(s->x_norm);
  F(s->y_mean_d);
  F(s->always);
  for (auto q : s->mask) F(q);
  for (auto q : s->xtx) F(q);
  for (auto q : s->xty) F(q);
  for (auto q : s->part_rs) F(q);
  for (auto q : s->r_rs) F(q);
  for (auto q : s->part2_rs) F(q);
  for (auto q : s->h_rs) F(q);
  F(s->logfact);
  F(s->Wv);
  F(s->llpart);
  F(s->bcur);
  F(s->bprev);
  F(s->gidx);
  F(s->gsz);
  F(s->goff);
  F(s->gcols_new);
  F(s->mblk);
  F(s->dcol);
  F(s->mblk2);
  F(s->mwork);
  F(s->zwork);
  F(s->allcols);
  for (auto q : s->gxtx_rs) F(q);
  for (auto q : s->geig_v_rs) F(q);
  for (auto q : s->geig_l_rs) F(q);
  for (auto q : s->cox_allocs) F(q);
  F(s->idcols);
  F(s->part2);
  F(s->bd);
  F(s->beta_dense);
  F(s->sol);
  F(s->tmpv);
  F(s->A_new);
  F(s->cand);
  F(s->tie_buf);
  F(s->fb_work);
  F(s->hist);
  F(s->gcols);
  F(s->info);
  F(s->hist_beta);
  F(s->hist_coef0);
  F(s->gtasks);
  for (auto &bt : s->big_tasks) F(bt.second);
  F(s->rdiag);
  F(s->zbig);
  for (auto &c : s->gcache) {
    F(c.g0);
    F(c.g1);
    F(c.A);
    F(c.meta);
  }
  F(s->Xp);
  F(s->zp);
  F(s->cvp_part);
  for (auto &c : s->cov) {
    F(c.G);
    if (!c.shares_map) {
      F(c.slot_of);
      F(c.meta);
    }
    F(c.GS);
    F(c.zero);
  }
  F(s->cov_part);
  F(s->cgb_work);
  F(s->bd2);
  F(s->inA);
  F(s->cov_bmm);
  F(s->cov_fcols);
  F(s->cov_extras);
  F(s->Rt);
  F(s->gsrc);
  F(s->gpart);
  F(s->Gt);
  F(s->init_idx_d);
  F(s->init_val_d);
  F(s->resblk);
  for (auto q : s->res_buf)
    if (q) (void)hipHostFree(q);
  if (s->pub_flag) (void)hipHostFree(s->pub_flag);
  for (auto q : s->snap)
    if (q) (void)hipFree(q);
  if (s->stage_h) (void)hipHostFree(s->stage_h);
  for (auto e : s->ev_pool) (void)hipEventDestroy(e);
  for (auto &ev : s->xtx_ev)
    if (ev && !s->parent) (void)hipEventDestroy(ev);
  if (s->st) ctx_stream_destroy(s->st);
  delete s;
}

// doubles in the score-pass partial sums of one row set
size_t part_elems(const bessx_session *s) {
  const size_t plane = (size_t)s->nrb * (size_t)s->p;
  return (s->model_type == 4 && s->cox.one_pass) ? 5 * plane + (size_t)s->nrb : plane;
}

// timing of the dominant kernel: event pairs on the session stream, resolved lazily
int k1_begin(bessx_session *s, hipEvent_t *a, hipEvent_t *b) {
  if (!s->timing) return 0;
  if (s->ev_used + 2 > s->ev_pool.size()) {
    for (int i = 0; i < 64; i++) {
      hipEvent_t e;
      HIPX(hipEventCreate(&e));
      s->ev_pool.push_back(e);
    }
  }
  *a = s->ev_pool[s->ev_used++];
  *b = s->ev_pool[s->ev_used++];
  HIPX(hipEventRecord(*a, s->st));
  return 0;
}

// after a stream synchronisation: fold the recorded pairs into the statistics.  A pair whose
// kernel fell through its gate (a speculative slot after convergence) is a real launch of
// near-zero work; it is excluded by the `counted` list the caller keeps.
int k1_collect(bessx_session *s, const std::vector<std::pair<size_t, bool>> &pairs) {
  if (!s->timing) return 0;
  for (auto &pr : pairs) {
    if (!pr.second || pr.first == (size_t)-1) continue;
    float ms = 0.f;
    HIPX(hipEventElapsedTime(&ms, s->ev_pool[pr.first], s->ev_pool[pr.first + 1]));
    s->k1_seconds += (double)ms * 1e-3;
    s->k1_launches += 1;
    s->k1_bytes += 8.0 * (double)s->n * (double)s->p;
  }
  s->ev_used = 0;
  return 0;
}

// the same for the panel launches of the covariance mode: a launch covers up to 2 groups of 32 columns and reads X
// ONCE whatever its width (8 n p algorithmic bytes per launch; 2 n p 32 flops per group: the launches are also
// counted by width, bessx_session_counter 25-28); nfill = length of the fill list the launches worked on
int cov_collect(bessx_session *s, int nfill) {
  if (!s->timing) return 0;
  for (auto &pr : s->cov_timed) {
    const int real = std::min(2, std::max(0, nfill / 32 - pr.second));
    if (real == 0) continue;
    float ms = 0.f;
    HIPX(hipEventElapsedTime(&ms, s->ev_pool[pr.first], s->ev_pool[pr.first + 1]));
    s->k1_seconds += (double)ms * 1e-3;
    s->k1_launches += 1;
    s->k1_bytes += 8.0 * (double)s->n * (double)s->p;
    s->panel_w_seconds[real - 1] += (double)ms * 1e-3;
    s->panel_w_launches[real - 1] += 1;
    if (test_hook("panel_log")) std::fprintf(stderr, "[bessx] panel pass: %d group(s) %.3f ms%s\n", real, ms, s->kch_owner ? " (chunk chain)" : "");
  }
  s->cov_timed.clear();
  s->ev_used = 0;
  return 0;
}

int alloc_gram_cache(bessx_session *s) {
  bessx_session::GramCache c;
  hipError_t e = dmalloc(&c.g0, (size_t)256 * 256);
  if (e == hipSuccess) e = dmalloc(&c.g1, (size_t)256 * 256);
  if (e == hipSuccess) e = dmalloc(&c.A, 256);
  if (e == hipSuccess) e = dmalloc(&c.meta, 2);
  if (e == hipSuccess) e = hipMemset(c.meta, 0, 2 * sizeof(int));
  if (e != hipSuccess) {  // nothing half-built is left behind
    (void)hipFree(c.g0);
    (void)hipFree(c.g1);
    (void)hipFree(c.A);
    (void)hipFree(c.meta);
    return fail(BESSX_ERR_HIP, std::string("Gram cache: ") + hipGetErrorString(e));
  }
  s->gcache.push_back(c);
  return 0;
}


int alloc_cov_cache(bessx_session *s, bool share_map) {
  bessx_session::CovCache c;
  hipError_t e = dmalloc(&c.G, (size_t)s->p * s->cov_C);
  if (share_map && !s->cov.empty()) {
    c.slot_of = s->cov[0].slot_of;
    c.meta = s->cov[0].meta;
    c.shares_map = true;
  }
  if (e == hipSuccess && !c.shares_map) e = dmalloc(&c.slot_of, (size_t)s->p);
  if (e == hipSuccess && !c.shares_map) e = dmalloc(&c.meta, 8);
  if (e == hipSuccess) e = dmalloc(&c.GS, (size_t)COV_CS * COV_CS);
  if (e == hipSuccess) e = hipMemset(c.GS, 0, (size_t)COV_CS * COV_CS * sizeof(double));
  if (e == hipSuccess && !c.shares_map) e = hipMemset(c.slot_of, 0xff, (size_t)s->p * sizeof(int));
  if (e == hipSuccess && !c.shares_map) e = hipMemset(c.meta, 0, 8 * sizeof(int));
  if (e == hipSuccess) e = dmalloc(&c.zero, 8);
  if (e == hipSuccess) e = hipMemset(c.zero, 0, 8 * sizeof(double));
  if (e != hipSuccess) {
    (void)hipFree(c.G);
    if (!c.shares_map) {
      (void)hipFree(c.slot_of);
      (void)hipFree(c.meta);
    }
    (void)hipFree(c.GS);
    (void)hipFree(c.zero);
    return fail(BESSX_ERR_HIP, std::string("Gram column cache: ") + hipGetErrorString(e));
  }
  s->cov.push_back(c);
  return 0;
}

// forget every cached quantity that outlives a fit: a path call starts from nothing, like bessCpp
// A stream with a hardware queue of its own (see fold_ctx_create): created with a compute-unit mask that names every
// compute unit of the device.  false: not available (or switched off by the test hook) -- the caller makes an ordinary one.
// leave_out > 0: a stream that may NOT use `leave_out` of the compute units (mask bits 0, stride, 2 stride, ...): the
// stream the chunk chains' fills run on -- the chains' own small kernels find those units free while a panel pass
// occupies every other one (bessx_kchunks.cpp).
// Streams of this kind hold a hardware queue each, outside the runtime's pool: a process (several sessions, several ranks
// rehearsed on one device) must not take them without bound -- beyond OWN_QUEUE_CAP per process the callers get ordinary
// pool streams (hipStreamNonBlocking, the parent's priority), which is what `false` has always meant to them.
// (What such a stream is: hipExtStreamCreateWithCUMask has no flags argument -- the stream is a BLOCKING stream of normal
// priority, i.e. it synchronises with work on the legacy null stream.  The library itself queues nothing on the null
// stream inside a path call; a host that does -- torch's default stream is the null stream -- serialises with the chains.
// INTEGRATION.md section 5.)
// Creating or destroying one costs ~20 ms (a hardware queue; tools/probe/cumask_stream_churn.hip), so a destroyed one is
// kept IDLE for the next session of the process instead (a drop-in call -- bessx_pywrap_bess / bessx_bessCpp -- creates
// a session per call: 4-9 such streams each time, 0.1-0.2 s, was what a repeated call paid for its chains).
struct OwnStream {
  hipStream_t st;
  int device, leave_out, stride;
  bool in_use;
};
static std::mutex g_own_mu;
static std::vector<OwnStream> g_own_streams;  // in use and idle ones: all count against the cap
static long long g_own_created = 0;           // streams of this kind the process has created so far (statistics)
long long ctx_streams_created() {
  std::lock_guard<std::mutex> lk(g_own_mu);
  return g_own_created;
}
static constexpr size_t OWN_QUEUE_CAP = 24;

void ctx_stream_destroy(hipStream_t st) {
  if (!st) return;
  {
    std::lock_guard<std::mutex> lk(g_own_mu);
    for (auto &o : g_own_streams)
      if (o.st == st && o.in_use) {
        (void)hipStreamSynchronize(st);  // (nothing of the old owner is left on it)
        o.in_use = false;
        return;
      }
  }
  (void)hipStreamDestroy(st);
}

bool ctx_stream_create(int device, hipStream_t *st, int leave_out, int stride) {
  const char *hook = test_hook("ctx_streams");
  if (hook && std::string(hook) == "pool") return false;
  {
    size_t cap = OWN_QUEUE_CAP;
    if (const char *ec = test_hook("ctx_streams_cap")) cap = (size_t)std::max(0, std::atoi(ec));
    std::lock_guard<std::mutex> lk(g_own_mu);
    size_t busy = 0;
    for (auto &o : g_own_streams) busy += o.in_use ? 1 : 0;
    if (busy >= cap) return false;
    for (auto &o : g_own_streams)
      if (!o.in_use && o.device == device && o.leave_out == leave_out && o.stride == stride) {
        o.in_use = true;
        *st = o.st;
        return true;
      }
    if (g_own_streams.size() >= std::max(cap, OWN_QUEUE_CAP)) {
      // every slot is taken by streams of another shape: give an idle one back to the runtime
      for (size_t i = 0; i < g_own_streams.size(); i++)
        if (!g_own_streams[i].in_use) {
          (void)hipStreamDestroy(g_own_streams[i].st);
          g_own_streams.erase(g_own_streams.begin() + (long)i);
          break;
        }
      if (g_own_streams.size() >= std::max(cap, OWN_QUEUE_CAP)) return false;
    }
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount <= 0) return false;
  std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
  for (int i = 0; i < leave_out; i++) {
    const int bit = (int)(((long)i * std::max(1, stride)) % prop.multiProcessorCount);
    mask[(size_t)bit / 32] &= ~(1u << (bit % 32));
  }
  if (hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    *st = nullptr;
    return false;
  }
  std::lock_guard<std::mutex> lk(g_own_mu);
  g_own_streams.push_back({*st, device, leave_out, stride, true});
  g_own_created++;
  return true;
}

// ... asked once per process and device whether that works here (the automatic chain count depends on it)
bool ctx_streams_own_queue(int device) {
  static std::mutex mu;
  static std::vector<int> known;  // device -> -1 unknown / 0 / 1
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0) return false;
  if ((int)known.size() <= device) known.resize((size_t)device + 1, -1);
  if (known[(size_t)device] < 0) {
    hipStream_t st = nullptr;
    const bool ok = ctx_stream_create(device, &st);
    if (ok) ctx_stream_destroy(st);
    known[(size_t)device] = ok ? 1 : 0;
  }
  return known[(size_t)device] == 1;
}

// hipStreamSynchronize with the session's deadline (BESSX_WAIT_TIMEOUT_S): polls the stream, yields between polls
int stream_wait_bounded(bessx_session *s, hipStream_t st, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail(BESSX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->wait_deadline_s)
      return fail(BESSX_ERR_HIP, std::string(what) + ": the stream did not drain within " +
                                     std::to_string(s->wait_deadline_s) + " s (BESSX_WAIT_TIMEOUT_S) -- the session can "
                                     "only be destroyed now");
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

int reset_path_caches(bessx_session *s) {
  kchains_quiesce(s);  // (chunk-chain contexts read the caches that are about to be cleared)
  if (s->ahead.armed) {
    s->ahead.armed = false;
    HIPX(hipStreamSynchronize(s->st));
  }
  s->pend_on = false;  // a deferred publication of a fit nobody will ask for
  s->hint.on = false;
  for (auto &c : s->cache) c.valid = c.model_only = false;
  s->dev_state_rs = -1;
  for (bessx_session *c : s->fold_ctx) {
    HIPX(hipStreamSynchronize(c->st));
    for (auto &cc : c->cache) cc.valid = cc.model_only = false;
    c->dev_state_rs = -1;
  }
  for (auto &g : s->gcache) HIPX(hipMemsetAsync(g.meta, 0, 2 * sizeof(int), s->st));
  for (auto &c : s->cov) {
    HIPX(hipMemsetAsync(c.slot_of, 0xff, (size_t)s->p * sizeof(int), s->st));
    HIPX(hipMemsetAsync(c.meta, 0, 8 * sizeof(int), s->st));
  }
  // The maps are shared with the fold and chunk-chain contexts, which run on streams of their own: the clearing must be
  // COMPLETE before any of them looks a column up.  (Round 4 left it queued on s->st.  A path whose first work on this
  // session is a set of fold fits -- a rank of a fold-sharded CV path that owns no full-data fit,
  // bessx_session_cv_eval(want_full = 0) -- then read the previous path's map while the memset was still waiting in
  // its queue, kept solving on slots that were handed out anew, and failed with "an active column was missing from
  // the Gram column cache": once in ~130 two-rank rehearsals on a loaded box, GPUTEST_r04.  Regression test:
  // tests/test_cv_shard_gpu.py::test_cache_reset_is_complete_before_the_fold_chains_read_the_map.)
  // (bounded like every wait of the host on the device: a stream that never comes back fails the call at the session's
  // deadline instead of hanging it, tests/test_deadline_gpu.py)
  return stream_wait_bounded(s, s->st, "clearing the path caches");
}

// k_chol outside the covariance form: only the work space of its pivoted fallback solve rides in the fuse block
CholFuse chol_fallback_only(const bessx_session *s) {
  CholFuse fz = {};
  fz.fb_work = s->fb_work;
  return fz;
}

void build_gram_tasks(int mt, std::vector<GramTask> &out) {
  for (int I = 0; I < mt; I++) {
    int J = 0, left = I + 1;
    for (int run = GRAM_JC; run >= 1; run >>= 1)
      while (left >= run) {
        out.push_back(GramTask{I, J, run, 0});
        J += run;
        left -= run;
      }
  }
}

// task list of the whole lower triangle for any tile count (lists for mt <= 16 are prebuilt)
int gram_tasks_for(bessx_session *s, int mt, const GramTask **tasks, int *ntask) {
  if (mt <= 16) {
    *tasks = s->gtasks + s->gtask_off[mt];
    *ntask = s->gtask_cnt[mt];
    return 0;
  }
  for (size_t i = 0; i < s->big_tasks.size(); i++)
    if (s->big_tasks[i].first == mt) {
      *tasks = s->big_tasks[i].second;
      *ntask = s->big_task_cnt[i];
      return 0;
    }
  std::vector<GramTask> t;
  build_gram_tasks(mt, t);
  GramTask *d = nullptr;
  HIPX(dmalloc(&d, t.size()));
  HIPX(hipMemcpy(d, t.data(), t.size() * sizeof(GramTask), hipMemcpyHostToDevice));
  s->big_tasks.push_back({mt, d});
  s->big_task_cnt.push_back((int)t.size());
  *tasks = d;
  *ntask = (int)t.size();
  return 0;
}

void gram_geometry(const bessx_session *s, int ntask, int *rows_per_slab, int *nslab, int ntiles,
                          bool allow_lds) {
  if (allow_lds && ntiles > 0 && gram_lds_applies(ntiles, 0) && s->ld >= 64) {
    // LDS-staged kernel: one block per slab computes every tile; slabs are whole 64-row chunks, about one block
    // (4 or 8 waves) per CU
    // (4-wave instance, up to 8 tile rows: two blocks fit a CU; measured 512 >= 256 > 128 slabs on configs[2])
    long ns = std::min<long>(ntiles <= 36 ? 512 : 256, s->ld / 64);
    if (s->gpart_elems > 0) ns = std::max<long>(1, std::min<long>(ns, (long)(s->gpart_elems / ((size_t)ntiles * 256))));
    long rps = ((s->ld + ns - 1) / ns + 63) / 64 * 64;
    *rows_per_slab = (int)rps;
    *nslab = (int)((s->ld + rps - 1) / rps);
    return;
  }
  long target = 4096;  // waves wanted in flight: 256 CUs x 4 SIMDs x 2 waves x 2
  long ns = std::max<long>(1, target / std::max(ntask, 1));
  ns = std::min<long>(ns, 192);  // more slabs only make the fixed-order reduction of the partials longer
  ns = std::min<long>(ns, std::max<long>(1, s->ld / 64));
  if (ntiles > 0 && s->gpart_elems > 0)  // the slab partials must fit the workspace
    ns = std::max<long>(1, std::min<long>(ns, (long)(s->gpart_elems / ((size_t)ntiles * 256))));
  long rps = (s->ld + ns - 1) / ns;
  rps = (rps + 15) / 16 * 16;
  ns = (s->ld + rps - 1) / rps;
  *rows_per_slab = (int)rps;
  *nslab = (int)ns;
}

// upload x (row- or column-major host memory) into the padded column-major device matrix
int upload_x(bessx_session *s, const double *x, int col_major) {
  const int n = s->n, p = s->p;
  HIPX(hipMemsetAsync(s->X, 0, (size_t)s->ld * p * sizeof(double), s->st));
  if (col_major) {
    HIPX(hipMemcpy2DAsync(s->X, (size_t)s->ld * sizeof(double), x, (size_t)n * sizeof(double),
                          (size_t)n * sizeof(double), (size_t)p, hipMemcpyHostToDevice, s->st));
    HIPX(hipStreamSynchronize(s->st));
    return 0;
  }
  // row-major: stage chunks of rows and transpose on the device
  size_t chunk_rows = std::max<size_t>(64, ((size_t)256 << 20) / ((size_t)p * sizeof(double)));
  chunk_rows = std::min<size_t>(chunk_rows, (size_t)n);
  double *stage = nullptr;
  HIPX(dmalloc(&stage, chunk_rows * (size_t)p));
  for (size_t r0 = 0; r0 < (size_t)n; r0 += chunk_rows) {
    size_t rows = std::min(chunk_rows, (size_t)n - r0);
    hipError_t e = hipMemcpyAsync(stage, x + r0 * (size_t)p, rows * (size_t)p * sizeof(double),
                                  hipMemcpyHostToDevice, s->st);
    if (e == hipSuccess) e = launch_transpose_in(stage, (int)rows, p, s->X, s->ld, (long)r0, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    if (e != hipSuccess) {
      (void)hipFree(stage);
      return fail(BESSX_ERR_HIP, std::string("upload_x: ") + hipGetErrorString(e));
    }
  }
  HIPX(hipFree(stage));
  return 0;
}

// X^T (m*y) and column sums of squares on a row set (group_XTX for 1x1 groups,
// src/utilities.cpp:153-165 and src/Metric.h:108-129): one pass of the two-accumulator K1 kernel.
// keep_yy: the row set was prepared before and y has not changed (a path call redoing the all-rows pass): the host's
// copy of y.(m y) stands, so nothing is read back and the host does not wait for the pass
int prepare_rowset(bessx_session *s, int rs, bool keep_yy) {
  const double *m = s->mask[rs];
  // tmpv = m*y (or y), v2 = m (or ones on data rows = aux column 1)
  if (launch_vec_mul(s->y, m, s->ld, s->tmpv, s->st) != hipSuccess) return fail(BESSX_ERR_HIP, "vec_mul");
  const double *v2 = m ? m : s->aux + s->ld;
  hipError_t e = launch_xtv(s->X, s->ld, s->p, s->U, s->tmpv, v2, s->part_rs[rs], s->part2, nullptr, 0, s->st);
  if (e == hipSuccess) e = launch_part_sum(s->part_rs[rs], s->nrb, s->p, s->xty[rs], s->st);
  if (e == hipSuccess && !(keep_yy && (int)s->yy_h.size() > rs)) {
    // y . (m y): the loss of an LM fit is y.y - beta.q - ridge |beta|^2 once (G + ridge I) beta = q is solved
    e = launch_dot(s->tmpv, s->y, s->ld, s->bd, s->st);  // bd is scratch here
    double v = 0.0;
    if (e == hipSuccess) e = hipMemcpyAsync(&v, s->bd, sizeof(double), hipMemcpyDeviceToHost, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    if ((int)s->yy_h.size() <= rs) s->yy_h.resize(rs + 1, 0.0);
    s->yy_h[rs] = v;
  }
  if (e == hipSuccess) e = launch_part_sum(s->part2, s->nrb, s->p, s->xtx[rs], s->st);
  if (e == hipSuccess && s->grouped)  // group_XTX blocks, src/utilities.cpp:153-165
    e = launch_group_moments(s->gmax, s->X, s->ld, s->n, m, nullptr, s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs],
                             nullptr, s->st);
  if ((size_t)rs < s->geig_valid.size()) s->geig_valid[(size_t)rs] = 0;  // (the blocks changed: diagonalise them again)
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("prepare_rowset: ") + hipGetErrorString(e));
  return 0;
}



int need_device() {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1)
    return fail(BESSX_ERR_HIP, "no HIP device visible: libbessx has no CPU path");
  return 0;
}

// copy a column-major (n x p, leading dimension ld_in) host matrix into a zero-padded device matrix
int upload_padded(Scratch &sc, const double *x, int n, int p, int ld_in, int U, double **dX, long *ld_out) {
  long rb = 128L * U;
  long ld = ((long)n + rb - 1) / rb * rb;
  HIPX(sc.alloc(dX, (size_t)ld * p));
  HIPX(hipMemset(*dX, 0, (size_t)ld * p * sizeof(double)));
  HIPX(hipMemcpy2D(*dX, (size_t)ld * sizeof(double), x, (size_t)ld_in * sizeof(double), (size_t)n * sizeof(double),
                   (size_t)p, hipMemcpyHostToDevice));
  *ld_out = ld;
  return 0;
}

int upload_vec_padded(Scratch &sc, const double *v, int n, long ld, double **dv) {
  std::vector<double> tmp((size_t)ld, 0.0);
  if (v) std::copy(v, v + n, tmp.begin());
  HIPX(sc.alloc(dv, (size_t)ld));
  HIPX(hipMemcpy(*dv, tmp.data(), (size_t)ld * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// Cox work space (the vectors of the state pass and of the Newton step, the k-sized blocks): the session's, and once
// more for every chunk-chain context of a Cox session (chain_ctx_create)
#define CX(call)                       \
  do {                                 \
    hipError_t e__ = (call);           \
    if (e__ != hipSuccess) return e__; \
  } while (0)
hipError_t cox_alloc(bessx_session *s) {
  const long ld = s->ld;
  const int n = s->n, capA = s->capA;
  {
    auto V = [&](double **dst, size_t count) -> hipError_t {
      hipError_t e = dmalloc(dst, count);
      if (e == hipSuccess) {
        s->cox_allocs.push_back(*dst);
        e = hipMemset(*dst, 0, count * sizeof(double));
      }
      return e;
    };
    CoxBufs &c = s->cox;
    double **vecs[] = {&c.E, &c.TH, &c.ET, &c.S0, &c.RS0, &c.SALL, &c.STEST, &c.EW, &c.WD, &c.ETA0, &c.THF, &c.S0F,
                       &c.RS0F, &c.VG, &c.WG1, &c.UD, &c.TH1, &c.S1};
    for (auto v : vecs) CX(V(v, (size_t)ld));
    double **vecs1[] = {&c.C1, &c.CU, &c.CV, &c.C2};
    for (auto v : vecs1) CX(V(v, (size_t)ld));
    CX(V(&c.ldl_work, CHOL_FB_DOUBLES));
    // k-sized work space: for sparsity levels up to 254 now, grown by cox_reserve() when a larger one is asked for
    s->cox_M_cols = 256;
    CX(V(&c.M, (size_t)ld * s->cox_M_cols));
    CX(V(&c.g, (size_t)capA));
    CX(V(&c.u, (size_t)capA));
    CX(V(&c.b0, (size_t)capA));
    CX(V(&c.Gt2, (size_t)136 * 256));
    CX(V(&c.llpart, (size_t)(n + 255) / 256 + 1));
    CX(V(&c.SCR, cox_scan_scratch_doubles(ld, 256)));
    // one-pass Hessian of the Newton step (k_cox_hess, up to 10 tile rows; test hook cox_hess=2pass: M = S1 / S0
    // materialised and two Gram launches, as in round 2)
    // Small samples keep the two-pass form: it is built like the reference's own formulas (M = S1 / S0, two Grams), so
    // on the ill-conditioned fits small n produces (near-separated risk sets, a ridge that outweighs the information
    // matrix) its rounding follows the reference's more closely -- both forms are accurate to rounding there, but a
    // Newton iteration on such a system amplifies rounding to 1e-4 and beyond (tests/test_cox_gpu.py).
    // test hook cox_hess=1pass forces the one-pass form at any size.
    c.fit_clamp = g_marginal_fit_variant == 2 ? 50.0 : 30.0;
    c.hess_fused = n >= 1024 ? 1 : 0;
    if (const char *ev = test_hook("cox_hess")) c.hess_fused = std::string(ev) == "2pass" ? 0 : (std::string(ev) == "1pass" ? 1 : c.hess_fused);
    if (c.hess_fused) {
      const size_t hrows = (size_t)cox_hess_slab_rows(ld), hns = ((size_t)ld + hrows - 1) / hrows;
      if (hns * 55 * 256 > s->gpart_elems) {
        c.hess_fused = 0;  // (cannot happen with the default workspace: 256 slabs x 55 tiles)
      } else {
        CX(cox_hess_prepare());
        CX(V(&c.CW, (size_t)ld));
        CX(V(&c.HP2, hns * 55 * 256));
        CX(V(&c.HT, hns * 160));
        CX(V(&c.CAR, hns * 160));
        CX(V(&c.HQ, hns * 160));
      }
    }
  }
  return hipSuccess;
}
#undef CX

}  // namespace bessx

extern "C" {


const char *bessx_last_error(void) { return g_err.c_str(); }

int bessx_device_info(char *buf, int buf_len) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) return fail(BESSX_ERR_HIP, "no HIP device visible");
  int dev = 0;
  HIPX(hipGetDevice(&dev));
  hipDeviceProp_t pr;
  HIPX(hipGetDeviceProperties(&pr, dev));
  std::snprintf(buf, (size_t)buf_len, "%s arch=%s CUs=%d LDS/block=%zu HBM=%.1f GiB clock=%d MHz", pr.name,
                pr.gcnArchName, pr.multiProcessorCount, (size_t)pr.sharedMemPerBlock,
                (double)pr.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), pr.clockRate / 1000);
  return BESSX_OK;
}

int bessx_session_create(bessx_session **out, const bessx_problem *pb) {
  if (!out || !pb || !pb->x || !pb->y) return fail(BESSX_ERR_ARG, "null argument");
  if (pb->n < 1 || pb->p < 1) return fail(BESSX_ERR_ARG, "n and p must be positive");
  if (pb->model_type < 1 || pb->model_type > 4) return fail(BESSX_ERR_ARG, "model_type must be 1..4");
  if (pb->algorithm_type != 1 && pb->algorithm_type != 5 && pb->algorithm_type != 2 && pb->algorithm_type != 3)
    return fail(BESSX_ERR_ARG, "algorithm_type must be 1, 2, 3 or 5 (src/bess.cpp:93)");
  if (pb->data_type < 1 || pb->data_type > 3) return fail(BESSX_ERR_ARG, "data_type must be 1..3");
  if (pb->max_iter < 1) return fail(BESSX_ERR_ARG, "max_iter must be >= 1");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1)
    return fail(BESSX_ERR_HIP, "no HIP device visible: libbessx has no CPU path");
  bessx_session *s = new bessx_session();
  int dev = pb->device;
  if (dev < 0) (void)hipGetDevice(&dev);
  s->device = dev;
  auto bail = [&](int rc) {
    std::string keep = g_err;
    session_free(s);
    g_err = keep;
    return rc;
  };
#define TRY(expr)                     \
  do {                                \
    int rc__ = (expr);                \
    if (rc__) return bail(rc__);      \
  } while (0)
#define HIPT(expr)                                                                             \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return bail(fail(BESSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)));    \
  } while (0)
  HIPT(hipSetDevice(dev));
  {
    // The session's own stream: an ordinary stream of the runtime's pool at NORMAL priority -- the level the fit contexts'
    // streams have (hipExtStreamCreateWithCUMask takes no priority; rounds 1-5 created this one at the highest level,
    // which the contexts' streams of round 5 no longer matched).  Not a stream with a queue of its own: creating and
    // destroying one costs ~20 ms (tools/probe/cumask_stream_churn.hip) and a path measured the same either way
    // (10.8-10.9 ms); BESSX_TEST_HOOKS=parent_stream=own / =high select the other two forms.
    const char *eps = test_hook("parent_stream");
    if (!(eps && std::string(eps) == "own" && ctx_stream_create(dev, &s->st))) {
      int lo = 0, hi = 0;
      HIPT(hipDeviceGetStreamPriorityRange(&lo, &hi));
      const int normal = std::max(std::min(lo, hi), std::min(std::max(lo, hi), 0));
      HIPT(hipStreamCreateWithPriority(&s->st, hipStreamDefault, (eps && std::string(eps) == "high") ? hi : normal));
    }
    HIPT(gram_lds_prepare());
    if (const char *ev = test_hook("irls_fuse")) s->irls_fuse = std::string(ev) == "1";
    if (const char *ev = test_hook("light_confirm")) s->light_confirm = std::string(ev) != "0";
    s->irls_wfloor = g_marginal_fit_variant == 1 ? 0 : 1;
  }
  const int n = pb->n;
  s->n = n;
  s->p = pb->p;
  s->p_full = pb->p;
  s->U = n >= 4096 ? 8 : (n >= 2048 ? 4 : (n >= 1024 ? 2 : 1));
  const long rb = 128L * s->U;
  s->ld = ((long)n + rb - 1) / rb * rb;
  s->nrb = (int)(s->ld / rb);
  const long ld = s->ld;
  std::vector<int> always_sel;
  for (int i = 0; i < pb->always_select_len; i++) always_sel.push_back(pb->always_select[i]);
  bool x_ready = false;
  if (pb->is_screening) {
    // screening(), src/screening.cpp:26-105, before anything else touches the data (src/bess.cpp:57-61)
    const int pf = pb->p, ss = pb->screening_size;
    const bool gscr = pb->group_index && pb->group_index_len > 0 && pb->group_index_len != pf;
    // (Poisson is refused below, with or without groups)
    if (gscr) {
      // groups of the original columns (Data::g_index semantics); screening_size and always_select count GROUPS
      const int Ng = pb->group_index_len;
      long long tot = 0;
      for (int g = 0; g < Ng; g++) {
        const int a = pb->group_index[g], b = g + 1 < Ng ? pb->group_index[g + 1] : pf;
        if ((g == 0 && a != 0) || b <= a || b > pf)
          return bail(fail(BESSX_ERR_ARG, "group_index must start at 0 and increase strictly"));
        tot += (long long)(b - a) * (b - a);
      }
      if (tot > 0x7fffffffLL) return bail(fail(BESSX_ERR_UNSUPPORTED, "group blocks exceed 2^31 entries in total"));
    }
    const int nunits = gscr ? pb->group_index_len : pf;  // what is ranked: groups or columns
    if (pb->model_type == 3)
      return bail(fail(BESSX_ERR_UNSUPPORTED, "Poisson screening: poisson_fit is undefined behaviour in the reference (src/poisson.cpp:113)"));
    if (ss < 1 || ss > nunits) return bail(fail(BESSX_ERR_ARG, "screening_size must be in 1..p (1..number of groups)"));
    if (!topk_supported(nunits, ss)) return bail(fail(BESSX_ERR_UNSUPPORTED, "screening_size too large for the top-k kernel"));
    std::vector<unsigned char> fl((size_t)pf, 0);
    for (int a : always_sel) {
      if (a < 0 || a >= nunits) return bail(fail(BESSX_ERR_ARG, "always_select index out of range"));
      fl[a] = 1;
    }
    double *Xraw = nullptr, *yw = nullptr, *scr = nullptr;
    int *ibuf = nullptr;
    unsigned char *fl_d = nullptr;
    auto drop = [&]() {
      (void)hipFree(Xraw);
      (void)hipFree(yw);
      (void)hipFree(scr);
      (void)hipFree(ibuf);
      (void)hipFree(fl_d);
      s->X = nullptr;
    };
#define HIPS(expr)                                                                            \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      drop();                                                                                 \
      return bail(fail(BESSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)));   \
    }                                                                                         \
  } while (0)
    HIPS(dmalloc(&Xraw, (size_t)ld * pf));
    s->X = Xraw;
    {
      int rc = upload_x(s, pb->x, pb->x_col_major);
      if (rc) {
        drop();
        return bail(rc);
      }
    }
    // yw: y | weight | ones, padded with zeros;  scr: score | partial sums / per-column solver state
    const size_t scr_len = (size_t)pf * 3 + std::max((size_t)2 * s->nrb * pf, (size_t)5 * pf);
    HIPS(dmalloc(&yw, (size_t)ld * 3));
    HIPS(dmalloc(&scr, scr_len));
    HIPS(dmalloc(&ibuf, (size_t)pf + ss + 32768 + 3 * (size_t)pf + 8));
    HIPS(dmalloc(&fl_d, (size_t)pf));
    HIPS(hipMemcpy(fl_d, fl.data(), (size_t)pf, hipMemcpyHostToDevice));
    {
      std::vector<double> tmp((size_t)ld * 3, 0.0);
      for (int i = 0; i < n; i++) {
        tmp[i] = pb->y[i];
        tmp[(size_t)ld + i] = pb->weight ? pb->weight[i] : 1.0;
        tmp[(size_t)2 * ld + i] = 1.0;
      }
      HIPS(hipMemcpy(yw, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    double *score = scr, *sxy = scr + pf, *sxx = scr + 2 * (size_t)pf, *work = scr + 3 * (size_t)pf;
    int *done = ibuf, *keep = ibuf + pf, *cand = ibuf + pf + ss;
    std::vector<int> g_lo, g_sz;  // grouped screening: first column and width of every original group
    if (gscr) {
      // LM marginal fit of a whole group (src/screening.cpp:44-48): moments X_g^T X_g, X_g^T y, then a Cholesky solve
      const int Ng = nunits;
      std::vector<int> g_off((size_t)Ng + 1, 0);
      int gmax = 1;
      g_lo.resize(Ng);
      g_sz.resize(Ng);
      for (int g = 0; g < Ng; g++) {
        g_lo[g] = pb->group_index[g];
        g_sz[g] = (g + 1 < Ng ? pb->group_index[g + 1] : pf) - g_lo[g];
        g_off[g + 1] = g_off[g] + g_sz[g] * g_sz[g];
        gmax = std::max(gmax, g_sz[g]);
      }
      int *gd = nullptr;
      double *gm = nullptr;
      auto gdrop = [&]() {
        (void)hipFree(gd);
        (void)hipFree(gm);
      };
      hipError_t e = dmalloc(&gd, (size_t)3 * Ng + 1);
      if (e == hipSuccess) e = dmalloc(&gm, (size_t)2 * g_off[Ng] + 3 * (size_t)pf);
      if (e == hipSuccess) e = hipMemcpy(gd, g_lo.data(), (size_t)Ng * sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(gd + Ng, g_sz.data(), (size_t)Ng * sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(gd + 2 * Ng, g_off.data(), ((size_t)Ng + 1) * sizeof(int), hipMemcpyHostToDevice);
      double *mblk = gm, *mwork = gm + g_off[Ng], *dcol = gm + 2 * (size_t)g_off[Ng], *zw = dcol + pf;
      if (pb->model_type == 1) {
        if (e == hipSuccess)
          e = launch_group_moments(gmax, Xraw, ld, n, nullptr, yw, Ng, gd, gd + Ng, gd + 2 * Ng, mblk, dcol, s->st);
        if (e == hipSuccess)
          e = launch_group_lsq_score(Ng, gd, gd + Ng, gd + 2 * Ng, mblk, dcol, fl_d, mwork, zw, score, s->st);
      } else if (pb->model_type == 4) {
        // cox_fit on the columns of a group (src/coxph.cpp:42-108): one block per group up to 4 columns
        // (k_screen_cox_group leaves wider groups alone; their fit follows below)
        if (e == hipSuccess)
          e = launch_screen_cox_group(Xraw, ld, n, Ng, gd, gd + Ng, yw, yw + ld, fl_d, score, s->st);
      } else {
        // logit_fit on the columns of a group (src/logistic.cpp:60-160): one block per group up to 8 columns
        double *gstate = nullptr;
        if (e == hipSuccess) e = dmalloc(&gstate, screen_logit_group_state_doubles(Ng));
        if (e == hipSuccess)
          e = launch_screen_logit_group(Xraw, ld, n, Ng, gd, gd + Ng, yw, yw + ld, gstate, done, fl_d, score, s->st);
        if (e == hipSuccess) e = hipStreamSynchronize(s->st);
        (void)hipFree(gstate);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(s->st);
      if (e == hipSuccess && pb->model_type != 1) {
        // Wider groups: logit_fit / cox_fit ARE the families' restricted fits on the group's columns (cold start,
        // lambda = 0, the same stopping rules) -- logit_fit without the floor of the IRLS weight, cox_fit with the
        // linear predictor clamped at 50 instead of 30 in the Newton direction.  Each such group is fitted by the
        // solver's own chain (IRLS: k_irls_gram + k_chol; Newton: k_cox_hess ...) in a sub-session that holds just its
        // columns, unnormalised, with every column active.
        const bool logit = pb->model_type == 2;
        const int limit = logit ? 8 : 4;
        for (int g = 0; g < Ng && e == hipSuccess; g++) {
          const int gs = g_sz[g];
          if (gs <= limit) continue;
          double sc = DBL_MAX;
          if (!fl[g]) {
            if (logit && n <= gs) {
              gdrop();
              drop();
              return bail(fail(BESSX_ERR_UNSUPPORTED, "logistic screening: a group at least as wide as the sample is "
                               "undefined behaviour in the reference (logit_fit returns n coefficients, "
                               "src/logistic.cpp:62-110, of which screening() reads the last g_size, src/screening.cpp:60)"));
            }
            std::vector<double> xs((size_t)n * gs);
            if (pb->x_col_major) {
              std::memcpy(xs.data(), pb->x + (size_t)g_lo[g] * n, xs.size() * sizeof(double));
            } else {
              for (int i = 0; i < n; i++)
                for (int u = 0; u < gs; u++) xs[(size_t)i * gs + u] = pb->x[(size_t)i * pf + g_lo[g] + u];
            }
            bessx_problem q = {};
            q.n = n;
            q.p = gs;
            q.x = xs.data();
            q.x_col_major = pb->x_col_major;
            q.y = pb->y;
            q.weight = pb->weight;
            q.data_type = logit ? 2 : 3;
            q.is_normal = 0;
            q.model_type = pb->model_type;
            q.algorithm_type = 1;
            q.max_iter = 2;  // the second PDAS iteration repeats the (complete) active set and ends the fit
            q.is_warm_start = 1;
            q.device = s->device;
            bessx_session *sub = nullptr;
            g_marginal_fit_variant = logit ? 1 : 2;
            int rc = bessx_session_create(&sub, &q);
            g_marginal_fit_variant = 0;
            std::vector<int> sup((size_t)gs);
            std::vector<double> bq((size_t)gs);
            if (rc == 0)
              rc = bessx_session_fit(sub, gs, 0.0, -1, nullptr, nullptr, 0, 0.0, sup.data(), bq.data(), nullptr, nullptr,
                                     nullptr, nullptr);
            if (sub) bessx_session_destroy(sub);
            (void)hipSetDevice(s->device);
            if (rc != 0) {
              gdrop();
              drop();
              return bail(rc);
            }
            double acc = 0.0;
            for (int u = 0; u < gs; u++) acc += bq[u] * bq[u];
            const double v = acc / (double)gs;  // coef_norm, src/screening.cpp:60
            sc = (v <= DBL_MAX) ? v : 0.0;
          }
          e = hipMemcpy(score + g, &sc, sizeof(double), hipMemcpyHostToDevice);
        }
      }
      gdrop();
      if (e != hipSuccess) {
        drop();
        return bail(fail(BESSX_ERR_HIP, std::string("group screening: ") + hipGetErrorString(e)));
      }
    } else if (pb->model_type == 1) {
      // beta_j = x_j.y / x_j.x_j: the closed form of lm_fit on one column (src/screening.cpp:44-47), one score pass
      HIPS(launch_xtv(Xraw, ld, pf, s->U, yw, yw + 2 * ld, work, work + (size_t)s->nrb * pf, nullptr, 0, s->st));
      HIPS(launch_part_sum(work, s->nrb, pf, sxy, s->st));
      HIPS(launch_part_sum(work + (size_t)s->nrb * pf, s->nrb, pf, sxx, s->st));
      HIPS(launch_screen_score_lm(sxy, sxx, pf, fl_d, score, s->st));
    } else if (pb->model_type == 2) {
      HIPS(launch_screen_logit(Xraw, ld, n, pf, yw, yw + ld, work, done, fl_d, score, s->st));
    } else {
      HIPS(launch_screen_cox(Xraw, ld, n, pf, yw, yw + ld, fl_d, score, s->st));
    }
    {
      // max_k(coef_norm, screening_size), src/screening.cpp:66: equal marginal scores (duplicated columns) are tied
      int *tflag = ibuf + pf + ss + 32768;
      HIPS(hipMemsetAsync(tflag, 0, 8 * sizeof(int), s->st));
      const TopkTie tie = {tflag, tflag + 8};
      HIPS(launch_topk(score, nunits, ss, keep, cand, nullptr, 0, s->st, nullptr, nullptr, &tie));
    }
    HIPS(hipStreamSynchronize(s->st));
    s->screen_map.assign((size_t)ss, 0);
    HIPS(hipMemcpy(s->screen_map.data(), keep, (size_t)ss * sizeof(int), hipMemcpyDeviceToHost));
    int pk = ss;  // columns kept
    if (gscr) {
      // kept groups -> their columns (ascending), the group index of the kept data, always_select by kept-group rank
      s->screen_groups = s->screen_map;
      s->screen_map.clear();
      s->scr_gidx.clear();
      for (int g : s->screen_groups) {
        s->scr_gidx.push_back((int)s->screen_map.size());
        for (int u = 0; u < g_sz[g]; u++) s->screen_map.push_back(g_lo[g] + u);
      }
      pk = (int)s->screen_map.size();
      (void)hipFree(ibuf);
      ibuf = nullptr;
      HIPS(dmalloc(&ibuf, (size_t)pk));
      keep = ibuf;
      HIPS(hipMemcpy(keep, s->screen_map.data(), (size_t)pk * sizeof(int), hipMemcpyHostToDevice));
    }
    double *X2 = nullptr;
    HIPS(dmalloc(&X2, (size_t)ld * pk));
    {
      hipError_t e = launch_gather_cols(Xraw, ld, keep, pk, X2, s->st);
      if (e == hipSuccess) e = hipStreamSynchronize(s->st);
      if (e != hipSuccess) {
        (void)hipFree(X2);
        drop();
        return bail(fail(BESSX_ERR_HIP, std::string("gather_cols: ") + hipGetErrorString(e)));
      }
    }
    drop();
#undef HIPS
    s->X = X2;
    s->p = pk;
    x_ready = true;
    // always_select re-indexed into the kept columns / groups (src/screening.cpp:90-102)
    const std::vector<int> &ranked = gscr ? s->screen_groups : s->screen_map;
    for (int &a : always_sel) a = (int)(std::lower_bound(ranked.begin(), ranked.end(), a) - ranked.begin());
  }
  const int p = s->p;
  s->data_type = pb->data_type;
  s->is_normal = pb->is_normal ? 1 : 0;
  s->model_type = pb->model_type;
  s->algorithm_type = pb->algorithm_type;
  s->max_iter = pb->max_iter;
  s->warm_start = pb->is_warm_start ? 1 : 0;
  {
    // groups: Data::g_index / g_size / g_num (src/Data.h:59-67)
    // (after screening with groups the session lives on the kept groups: scr_gidx is their group index)
    const bool kept_groups = pb->is_screening && !s->scr_gidx.empty();
    const bool have_groups = kept_groups || (pb->group_index && pb->group_index_len > 0 && !pb->is_screening);
    const int *gsrc = kept_groups ? s->scr_gidx.data() : pb->group_index;
    const int gl = kept_groups ? (int)s->scr_gidx.size() : (have_groups ? pb->group_index_len : p);
    s->N = gl;
    s->gidx_h.resize(gl);
    s->gsz_h.resize(gl);
    s->goff_h.resize(gl + 1);
    s->goff_h[0] = 0;
    for (int g = 0; g < gl; g++) {
      const int a = have_groups ? gsrc[g] : g;
      const int b = g + 1 < gl ? (have_groups ? gsrc[g + 1] : g + 1) : p;
      if ((g == 0 && a != 0) || b <= a || b > p) return bail(fail(BESSX_ERR_ARG, "group_index must start at 0 and increase strictly"));
      s->gidx_h[g] = a;
      s->gsz_h[g] = b - a;
      s->gmax = std::max(s->gmax, b - a);
      const long long nxt = (long long)s->goff_h[g] + (long long)(b - a) * (b - a);
      if (nxt > 0x7fffffffLL) return bail(fail(BESSX_ERR_UNSUPPORTED, "group blocks exceed 2^31 entries in total"));
      s->goff_h[g + 1] = (int)nxt;
    }
    s->grouped = s->gmax > 1;
    s->g_uniform = 0;
    if (s->grouped) {
      bool same = true;
      for (int g = 0; g < gl; g++) same = same && s->gsz_h[g] == s->gmax;
      if (same) s->g_uniform = s->gmax;
      if (const char *ev = test_hook("group_expand"))
        if (std::string(ev) == "host") s->g_uniform = 0;
      if (const char *ev = test_hook("group_eig")) s->geig_on = std::string(ev) != "0";
    }
    // groups of up to 16 columns: register-resident blocks and a Jacobi square root per thread; wider ones: tiled
    // moments and a Cholesky form of the same score (k_group_moments_big / k_group_score_big).  Cox forms the
    // suffix-sum matrix of whole groups in a 256-column panel.
    if (s->gmax > 256 && s->model_type == 4)
      return bail(fail(BESSX_ERR_UNSUPPORTED, "Cox: groups wider than 256 columns are not built"));
    if (s->grouped && s->model_type == 4 && !(pb->algorithm_type == 2 || pb->algorithm_type == 3))
      return bail(fail(BESSX_ERR_UNSUPPORTED, "Cox with groups of size > 1 exists only for algorithm_type 2 / 3 (the "
                                              "group branch of GroupPdasCox::get_A, src/Algorithm.h:1497-1568)"));
  }
  if (!x_ready) HIPT(dmalloc(&s->X, (size_t)ld * p));
  HIPT(dmalloc(&s->y, (size_t)ld));
  HIPT(dmalloc(&s->w, (size_t)ld));
  HIPT(dmalloc(&s->aux, (size_t)ld * 3));
  HIPT(dmalloc(&s->x_mean, (size_t)p));
  HIPT(dmalloc(&s->x_norm, (size_t)p));
  HIPT(dmalloc(&s->y_mean_d, 1));
  HIPT(dmalloc(&s->always, (size_t)p));
  HIPT(dmalloc(&s->tmpv, (size_t)ld));
  HIPT(dmalloc(&s->part2, (size_t)s->nrb * p));
  HIPT(dmalloc(&s->bd, (size_t)p));
  HIPT(dmalloc(&s->beta_dense, (size_t)p));
  if (pb->max_sparsity < 0 || pb->max_sparsity > T0_HARD)
    return bail(fail(BESSX_ERR_ARG, "max_sparsity must be in [0, " + std::to_string(T0_HARD) + "]"));
  s->cap = std::min(p, std::max(T0_CAP, pb->max_sparsity));
  s->capA = (s->cap + 2 + 15) / 16 * 16;
  s->capA = std::max(s->capA, 256);
  s->hist_stride = s->capA;
  const int capA = s->capA, mt_max = capA / 16;
  HIPT(dmalloc(&s->sol, (size_t)capA));
  HIPT(dmalloc(&s->A_new, (size_t)capA));
  HIPT(dmalloc(&s->rdiag, (size_t)capA));
  HIPT(dmalloc(&s->zbig, (size_t)capA));
  HIPT(dmalloc(&s->cand, 32768));
  HIPT(dmalloc(&s->fb_work, CHOL_FB_DOUBLES));
  HIPT(dmalloc(&s->tie_buf, (size_t)3 * p + 8));
  HIPT(hipMemset(s->tie_buf, 0, 8 * sizeof(int)));
  s->tie = TopkTie{s->tie_buf, s->tie_buf + 8};
  HIPT(dmalloc(&s->hist, (size_t)(s->max_iter + 2) * s->hist_stride));
  HIPT(dmalloc(&s->hist_beta, (size_t)(s->max_iter + 2) * s->hist_stride));
  HIPT(dmalloc(&s->hist_coef0, (size_t)(s->max_iter + 2)));
  HIPT(dmalloc(&s->gcols, (size_t)capA + 16));
  HIPT(dmalloc(&s->Rt, (size_t)16 * 256));
  HIPT(dmalloc(&s->gsrc, 256));
  HIPT(dmalloc(&s->init_idx_d, (size_t)capA));
  HIPT(dmalloc(&s->init_val_d, (size_t)capA));
  HIPT(dmalloc(&s->Gt, (size_t)mt_max * (mt_max + 1) / 2 * 256));
  // fp64 partial tiles of the row slabs: 48 MB, or at least 8 slabs of the largest Gram this session can form
  s->gpart_elems = std::max<size_t>((size_t)6 << 20, (size_t)8 * mt_max * (mt_max + 1) / 2 * 256);
  HIPT(dmalloc(&s->gpart, s->gpart_elems));
  // Gram task lists for every tile count
  {
    std::vector<GramTask> all;
    s->gtask_off.assign(17, 0);
    s->gtask_cnt.assign(17, 0);
    s->gtask_inc_off.assign(17, 0);
    s->gtask_inc_cnt.assign(17, 0);
    for (int mt = 1; mt <= 16; mt++) {
      s->gtask_off[mt] = (int)all.size();
      build_gram_tasks(mt, all);
      s->gtask_cnt[mt] = (int)all.size() - s->gtask_off[mt];
      // extra tile row I = mt against the tiles J = 0..mt-1, in runs of 8/4/2/1
      s->gtask_inc_off[mt] = (int)all.size();
      int J = 0, left = mt;
      for (int run = GRAM_JC; run >= 1; run >>= 1)
        while (left >= run) {
          all.push_back(GramTask{mt, J, run, 0});
          J += run;
          left -= run;
        }
      s->gtask_inc_cnt[mt] = (int)all.size() - s->gtask_inc_off[mt];
    }
    HIPT(dmalloc(&s->gtasks, all.size()));
    HIPT(hipMemcpy(s->gtasks, all.data(), all.size() * sizeof(GramTask), hipMemcpyHostToDevice));
  }
  // result block
  s->n_sse_blk = (int)((ld + 255) / 256);
  {
    size_t off = 0;
    size_t o_ctrl = off;
    off += 128;
    size_t o_sse = off;
    off += (size_t)2 * s->n_sse_blk * sizeof(double);
    size_t o_b = off;
    off += (size_t)capA * sizeof(double);
    size_t o_a = off;
    off += (size_t)capA * sizeof(int);
    s->res_bytes = off;
    HIPT(hipMalloc(reinterpret_cast<void **>(&s->resblk), off));
    HIPT(hipMemset(s->resblk, 0, off));
    s->ctrl = reinterpret_cast<FitCtrl *>(s->resblk + o_ctrl);
    s->sse = reinterpret_cast<double *>(s->resblk + o_sse);
    s->b_cur = reinterpret_cast<double *>(s->resblk + o_b);
    s->A_cur = reinterpret_cast<int *>(s->resblk + o_a);
    for (int b = 0; b < 2; b++) {
      HIPT(hipHostMalloc(reinterpret_cast<void **>(&s->res_buf[b]), off));
      std::memset(s->res_buf[b], 0, off);
    }
    s->res_h = s->res_buf[0];
    for (int b = 0; b < 2; b++) {
      HIPT(hipMalloc(reinterpret_cast<void **>(&s->snap[b]), off + 64));
      HIPT(hipMemset(s->snap[b], 0, off + 64));
    }
    HIPT(hipHostMalloc(reinterpret_cast<void **>(&s->pub_flag), 128));
    s->pub_flag[0] = 0ull;
    s->pub_flag[8] = 0ull;  // second buffer's flag, its own cache line
    if (const char *ev = test_hook("publish")) s->publish = std::atoi(ev) != 0;
    if (const char *ev = std::getenv("BESSX_WAIT_TIMEOUT_S")) s->wait_deadline_s = std::max(0.001, std::atof(ev));
    if (const char *ev = test_hook("chain")) s->chain = std::atoi(ev) != 0;
    if (const char *ev = std::getenv("BESSX_KPATH_CHAINS")) s->kpath_chains = std::max(0, std::atoi(ev));
    if (!s->publish) s->chain = false;
    HIPT(hipHostMalloc(reinterpret_cast<void **>(&s->stage_h), (size_t)capA * (sizeof(int) + sizeof(double))));
  }
  static_assert(sizeof(FitCtrl) <= 128, "FitCtrl must fit its slot of the result block");
  // data
  if (!x_ready) TRY(upload_x(s, pb->x, pb->x_col_major));
  {
    std::vector<double> tmp((size_t)ld, 0.0);
    std::copy(pb->y, pb->y + n, tmp.begin());
    HIPT(hipMemcpy(s->y, tmp.data(), (size_t)ld * sizeof(double), hipMemcpyHostToDevice));
    std::fill(tmp.begin(), tmp.end(), 0.0);
    for (int i = 0; i < n; i++) tmp[i] = pb->weight ? pb->weight[i] : 1.0;
    HIPT(hipMemcpy(s->w, tmp.data(), (size_t)ld * sizeof(double), hipMemcpyHostToDevice));
    // aux: column 0 zeros, column 1 ones on the data rows, column 2 working response
    HIPT(hipMemset(s->aux, 0, (size_t)ld * 3 * sizeof(double)));
    std::fill(tmp.begin(), tmp.end(), 0.0);
    std::fill(tmp.begin(), tmp.begin() + n, 1.0);
    HIPT(hipMemcpy(s->aux + ld, tmp.data(), (size_t)ld * sizeof(double), hipMemcpyHostToDevice));
    std::vector<unsigned char> fl((size_t)p, 0);
    for (int a : always_sel) {
      if (a < 0 || a >= s->N) return bail(fail(BESSX_ERR_ARG, "always_select index out of range"));
      fl[a] = 1;
    }
    HIPT(hipMemcpy(s->always, fl.data(), (size_t)p, hipMemcpyHostToDevice));
  }
  HIPT(hipMemset(s->x_mean, 0, (size_t)p * sizeof(double)));
  HIPT(hipMemset(s->x_norm, 0, (size_t)p * sizeof(double)));
  {
    // Data::normalize + add_weight (LM only, src/bess.cpp:97)
    hipError_t e = launch_normalize(s->X, ld, n, p, s->y, s->w, s->data_type, s->is_normal, s->model_type == 1,
                                    s->x_mean, s->x_norm, s->y_mean_d, s->st);
    if (e != hipSuccess) return bail(fail(BESSX_ERR_HIP, std::string("normalize: ") + hipGetErrorString(e)));
    HIPT(hipStreamSynchronize(s->st));
    s->x_mean_h.assign((size_t)p, 0.0);
    s->x_norm_h.assign((size_t)p, 0.0);
    HIPT(hipMemcpy(s->x_mean_h.data(), s->x_mean, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    HIPT(hipMemcpy(s->x_norm_h.data(), s->x_norm, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    HIPT(hipMemcpy(&s->y_mean_h, s->y_mean_d, sizeof(double), hipMemcpyDeviceToHost));
    {
      // Data::get_nullloss, src/Data.h:120-130, on the response as Data holds it after normalize() (centred by the
      // weighted mean for data_type 1) and, for the linear model, add_weight() (rows times sqrt(w), src/bess.cpp:97)
      double acc = 0.0, wsum = 0.0;
      for (int i = 0; i < n; i++) {
        const double wi = pb->weight ? pb->weight[i] : 1.0;
        const double yi = pb->y[i] - (pb->data_type == 1 && pb->is_normal ? s->y_mean_h : 0.0);
        acc += (pb->model_type == 1 ? wi : 1.0) * yi * yi;
        wsum += wi;
      }
      s->nullloss = pb->data_type == 1 ? acc / (double)n : 2.0 * std::log(2.0) * wsum;
    }
  }
  // row set 0: all rows
  s->mask.push_back(nullptr);
  s->n_train.push_back(n);
  double *q = nullptr;
  HIPT(dmalloc(&q, (size_t)p));
  s->xtx.push_back(q);
  HIPT(dmalloc(&q, (size_t)p));
  s->xty.push_back(q);
  // Cox: the score pass reads X once and leaves five partial sums per (row block, column) + one per block
  // (k_cox_score1p); the test hook cox_score=2pass keeps the totals / carry / rescan form (two reads of X)
  if (s->model_type == 4) {
    const char *ev = test_hook("cox_score");
    s->cox.one_pass = !(ev && std::string(ev) == "2pass");
    s->cox.need_uv = s->grouped ? 1 : 0;
  }
  HIPT(dmalloc(&q, part_elems(s)));
  s->part_rs.push_back(q);
  HIPT(dmalloc(&q, (size_t)ld));
  HIPT(hipMemset(q, 0, (size_t)ld * sizeof(double)));
  s->r_rs.push_back(q);
  HIPT(dmalloc(&q, (size_t)s->nrb * p));
  s->part2_rs.push_back(q);
  HIPT(dmalloc(&q, (size_t)ld));
  HIPT(hipMemset(q, 0, (size_t)ld * sizeof(double)));
  s->h_rs.push_back(q);
  HIPT(dmalloc(&s->Wv, (size_t)ld));
  HIPT(hipMemset(s->Wv, 0, (size_t)ld * sizeof(double)));
  s->llpart_cap = (size_t)std::max(s->n_sse_blk, 1024);  // (also one entry per row slab of k_irls_gram)
  HIPT(dmalloc(&s->llpart, s->llpart_cap));
  HIPT(dmalloc(&s->bcur, (size_t)capA + 16));
  HIPT(dmalloc(&s->bprev, (size_t)capA + 16));
  HIPT(dmalloc(&s->logfact, (size_t)ld));
  {
    // sum_{j=1..y} log j per row, the loop of loglik_poisson (src/poisson.cpp:27-41); only Poisson reads it
    std::vector<double> lf((size_t)ld, 0.0);
    if (s->model_type == 3)
      for (int i = 0; i < n; i++) {
        double t = 0.0;
        if (pb->y[i] != 1.0)
          for (double j = 1.0; j <= pb->y[i]; j = j + 1.0) t = t + std::log(j);
        lf[i] = t;
      }
    HIPT(hipMemcpy(s->logfact, lf.data(), (size_t)ld * sizeof(double), hipMemcpyHostToDevice));
  }
  s->cache.assign(1, bessx_session::RsCache());
  if (s->grouped) {
    HIPT(dmalloc(&s->gidx, (size_t)s->N));
    HIPT(dmalloc(&s->gsz, (size_t)s->N));
    HIPT(dmalloc(&s->goff, (size_t)s->N + 1));
    HIPT(hipMemcpy(s->gidx, s->gidx_h.data(), (size_t)s->N * sizeof(int), hipMemcpyHostToDevice));
    HIPT(hipMemcpy(s->gsz, s->gsz_h.data(), (size_t)s->N * sizeof(int), hipMemcpyHostToDevice));
    HIPT(hipMemcpy(s->goff, s->goff_h.data(), ((size_t)s->N + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPT(dmalloc(&s->gcols_new, (size_t)s->capA));
    HIPT(dmalloc(&s->mblk, (size_t)s->goff_h[s->N]));
    HIPT(dmalloc(&s->dcol, (size_t)p));
    if (s->gmax > 16) {
      HIPT(dmalloc(&s->mwork, (size_t)s->goff_h[s->N]));
      HIPT(dmalloc(&s->zwork, (size_t)2 * p));
    }
    if (s->model_type == 4) {
      HIPT(dmalloc(&s->mblk2, (size_t)s->goff_h[s->N]));
      HIPT(dmalloc(&s->allcols, (size_t)p));
      HIPT(launch_iota(s->allcols, p, s->st));
    }
    HIPT(dmalloc(&q, (size_t)s->goff_h[s->N]));
    s->gxtx_rs.push_back(q);
  }
  TRY(alloc_gram_cache(s));
  {
    // covariance-update mode: LM with singleton groups, unless the caller or BESSX_SCORE_MODE asks for the
    // streaming form (1) -- 2 insists on it (error if it cannot be set up), 0 = automatic
    int mode = pb->score_mode;
    if (mode == 0)
      if (const char *ev = std::getenv("BESSX_SCORE_MODE")) mode = std::atoi(ev);
    if (mode < 0 || mode > 2) return bail(fail(BESSX_ERR_ARG, "score_mode must be 0 (auto), 1 (streaming) or 2 (covariance)"));
    // (groups: every group of one width <= 16 columns -- the fits whose selected groups are expanded to columns on the
    // device --, all rows, and only with a cache that holds every column of the design)
    const bool eligible = s->model_type == 1 && (!s->grouped || (s->g_uniform > 0 && s->gmax <= GRP_EIG_MAX));
    if (mode == 2 && !eligible)
      return bail(fail(BESSX_ERR_ARG, "covariance score mode exists for LM with singleton groups or groups of one width "
                                      "(at most 16 columns) only"));
    if (eligible && mode != 1) {
      // capacity: every column if p is small, else a few active sets' worth, within 1 GiB per row set
      // (fills of exactly two groups take the pair panel kernel, one pass for 64 columns; the switches that made every
      // fill speculate 64 columns, or never paired, measured at parity in rounds 2-3 and are gone)
      s->cov_spec = COV_R;
      if (const char *ev = test_hook("cov_spec_min")) s->cov_spec_min = std::max(1, std::min(COV_R - 1, std::atoi(ev)));  // (16: rounds 2-4)
      // capacity: EVERY column when that fits 2 GiB per row set (p <= ~16000: a path then forms a column at most once and
      // the cache is never started over -- at 2560 columns the reference's default sequence 1..min(p, n / log n) at
      // n = 25000, p = 3000 restarted it 1237 times and streamed X 73 000 times, round 4); else 2560 columns within 2 GiB
      const long all = ((long)p + 31) / 32 * 32 + COV_R + s->cov_spec;
      const long budget = (((long)2 << 30) / ((long)p * 8)) / 32 * 32;
      long C = all <= budget ? all : std::min<long>(2560, budget);
      if (const char *ev = test_hook("cov_cap"))  // a small cache exercises the restart path
        C = std::min<long>(C, std::max(0, std::atoi(ev)) / 32 * 32);
      if (s->grouped && C < all) C = 0;  // (grouped fits never start the cache over)
      if (C >= 2 * COV_R + s->cov_spec) {
        s->cov_mode = true;
        s->cov_C = (int)C;
        const int pt = (p + 15) / 16, njg = (pt + cov_streamed_tiles_per_wave() - 1) / cov_streamed_tiles_per_wave();
        // row slabs: two 256-thread blocks of the panel kernel share a CU (50 KB of LDS each), so pick the slab count
        // whose block count wastes the least of the last round of 512 blocks
        // the fills' kernel: k_cov_panel_dp (round 5; the default since the chunk chains' coarse phase became a dense
        // run of passes, where it is 3-4 % faster per pass: DESIGN.md 13); test hook panel=lds: k_cov_panel_lds2 / _pair
        s->panel_variant = 5;
        if (const char *ev = test_hook("panel")) s->panel_variant = std::string(ev) == "dp" ? 5 : 0;
        long ns = 1, rps = ld;
        long panel_blocks = 0;
        {
          // (k_cov_panel_dp: one 8-wave block per compute unit, 128 streamed columns per block)
          const bool dp = s->panel_variant == 5;
          const long conc = dp ? 256 : 512;  // blocks resident at a time
          double best = 1e300;
          const long ns_max = std::max<long>(1, std::min<long>(64, ld / 256));
          for (long t = 1; t <= ns_max; t++) {
            const long r = ((ld + t - 1) / t + 63) / 64 * 64, used = (ld + r - 1) / r;
            const long blocks = (long)(dp ? (njg + 1) / 2 : njg) * used;
            const double cost = (double)((blocks + conc - 1) / conc) * (double)r * (blocks < conc ? 2.0 : 1.0);
            if (cost < best) {
              best = cost;
              ns = used;
              rps = r;
              panel_blocks = blocks;
            }
          }
        }
        s->cov_panel_blocks = (int)std::min<long>(panel_blocks, 1 << 30);
        s->cov_rps = (int)rps;
        s->cov_nslab = (int)ns;
        HIPT(dmalloc(&s->cov_part, (size_t)COV_SLOT_GROUPS * ns * njg * cov_streamed_tiles_per_wave() * 2 * 256));
        HIPT(dmalloc(&s->bd2, (size_t)p));
        HIPT(dmalloc(&s->inA, (size_t)p));
        HIPT(hipMemset(s->inA, 0, (size_t)p));
        HIPT(dmalloc(&s->cov_bmm, (size_t)3 * ((p + 31) / 32)));
        HIPT(dmalloc(&s->cov_fcols, (size_t)s->capA + 4 * COV_R));
        HIPT(dmalloc(&s->cov_extras, (size_t)2 * COV_R));
        TRY(alloc_cov_cache(s));
        HIPT(cov_panel_prepare());
        if (const char *ev = test_hook("cov_solver")) s->cov_cg = std::string(ev) != "chol";
        if (const char *ev = test_hook("fuse")) s->fuse = std::string(ev) != "0";
        if (const char *ev = test_hook("fuse_sel")) s->fuse_sel = std::string(ev) != "0";
        if (const char *ev = test_hook("defer_publish")) s->defer_pub = std::string(ev) != "0";
        if (!s->fuse) s->defer_pub = false;
        if (const char *ev = test_hook("cg_layout")) s->cg_by_rows = std::string(ev) != "tiles";
        if (const char *ev = test_hook("cov_cs")) s->cov_cs = std::min(COV_CS, std::max(1, std::atoi(ev)));
      } else if (mode == 2) {
        return bail(fail(BESSX_ERR_ARG, "covariance score mode: p too large for the Gram column cache"));
      }
    }
  }
  HIPT(dmalloc(&s->idcols, (size_t)capA + 16));
  if (s->model_type == 4) HIPT(cox_alloc(s));
  if (s->model_type == 1) {
    // group_XTX of the all-rows set (X^T y, diag(X^T X): one pass over X, src/path.cpp:37).  The session needs it for
    // bessx_session_fit callers; every cold path call redoes it inside the call like the reference (run_path)
    TRY(prepare_rowset(s, 0));
    if (const char *ev = test_hook("path_group_xtx")) s->path_group_xtx = std::string(ev) != "0";
  }
  HIPT(hipStreamSynchronize(s->st));
#undef TRY
#undef HIPT
  *out = s;
  return BESSX_OK;
}

void bessx_session_destroy(bessx_session *s) { session_free(s); }

// Free everything bessx_session_set_cv allocated for the folds (row sets 1..K); every vector is walked by its own
// length, so this is safe on the partly built state an allocation failure leaves behind.
static void drop_folds(bessx_session *s) {
  drop_fold_contexts(s);
  auto drop = [](std::vector<double *> &v) {
    for (size_t i = 1; i < v.size(); i++) (void)hipFree(v[i]);
    if (!v.empty()) v.resize(1);
  };
  drop(s->mask);
  drop(s->xtx);
  drop(s->xty);
  drop(s->part_rs);
  drop(s->r_rs);
  drop(s->part2_rs);
  drop(s->h_rs);
  drop(s->gxtx_rs);
  drop(s->geig_v_rs);
  drop(s->geig_l_rs);
  if (s->geig_valid.size() > 1) s->geig_valid.resize(1);
  if (s->geig_lambda.size() > 1) s->geig_lambda.resize(1);
  for (size_t i = 1; i < s->gcache.size(); i++) {
    (void)hipFree(s->gcache[i].g0);
    (void)hipFree(s->gcache[i].g1);
    (void)hipFree(s->gcache[i].A);
    (void)hipFree(s->gcache[i].meta);
  }
  if (!s->gcache.empty()) s->gcache.resize(1);
  (void)hipFree(s->Xp);
  (void)hipFree(s->zp);
  (void)hipFree(s->cvp_part);
  s->Xp = s->zp = s->cvp_part = nullptr;
  s->cv_shared = false;
  for (size_t i = 1; i < s->cov.size(); i++) {
    (void)hipFree(s->cov[i].G);
    if (!s->cov[i].shares_map) {
      (void)hipFree(s->cov[i].slot_of);
      (void)hipFree(s->cov[i].meta);
    }
    (void)hipFree(s->cov[i].GS);
    (void)hipFree(s->cov[i].zero);
  }
  if (!s->cov.empty()) s->cov.resize(1);
  if (!s->n_train.empty()) s->n_train.resize(1);
  s->n_test.clear();
  s->cv_init.clear();
  s->cv_fold.clear();
  s->K = 0;
}

// Context of row set rs for the fold chains that run side by side (see bessx_session::fold_ctx): a copy of the parent
// that borrows its data and caches and owns the state a fit writes.  Same capacities as the parent, so every enqueue
// function of the covariance form works on it unchanged.
static int fold_ctx_create(bessx_session *ps, int rs, bessx_session **out) {
  bessx_session *c = new bessx_session(*ps);
  c->parent = ps;
  c->fold_pool = nullptr;
  c->fold_ctx.clear();
  c->fill_ctrl = c->fill_ctrl_h = nullptr;
  c->ev_fill = c->ev_ctx = nullptr;
  c->ev_pool.clear();
  c->ev_used = 0;
  c->timing = false;
  c->trace = Trace();
  c->cov_timed.clear();
  c->cox_allocs.clear();
  c->publish = false;  // results by an asynchronous copy of the block: the driver waits for all chains at once
  c->chain = false;
  c->defer_pub = false;
  c->cov_no_restart = true;
  c->hint = bessx_session::Hint();
  c->ahead = bessx_session::Ahead();
  c->pend_on = false;
  c->cache.assign(ps->cache.size(), bessx_session::RsCache());
  c->dev_state_rs = -1;
  c->bmm_owner = -1;
  c->fit_serial = 0;
  c->cur_rows = rs;
  c->n_fits = c->n_iters = 0;
  c->cov_cg_fallbacks = c->cov_tie_rescues = c->cov_panel_groups = 0;
  c->chain_queued = c->chain_hits = c->chain_dead = c->chain_mismatch = 0;
  c->dbg_waits = c->dbg_waits_ready = 0;
  c->pub_flag = nullptr;  // (allocated below: the chains hand their result blocks over by k_publish)
  c->pub_seq = 0;
  c->snap[0] = c->snap[1] = nullptr;
  c->res_buf[0] = c->res_buf[1] = nullptr;
  c->res_h = nullptr;
  c->stage_h = nullptr;
  c->st = nullptr;
  // owned device buffers: cleared first so that a failure half way frees only what this function allocated
  c->resblk = nullptr;
  c->bd = c->bd2 = c->beta_dense = c->cov_bmm = c->sol = c->fb_work = c->hist_beta = c->hist_coef0 = c->Gt = nullptr;
  c->init_val_d = c->rdiag = c->zbig = nullptr;
  c->inA = nullptr;
  c->A_new = c->cand = c->tie_buf = c->hist = c->init_idx_d = c->cov_fcols = c->cov_extras = nullptr;
  const int p = ps->p, capA = ps->capA, mt_max = capA / 16;
  hipError_t e = hipSuccess;
  {
    // A stream with a hardware queue OF ITS OWN, whatever the process' queue budget: the HIP runtime multiplexes ordinary
    // streams onto a pool of GPU_MAX_HW_QUEUES (default 4) hardware queues per priority level -- chains that share a
    // queue wait for each other, which is why round 4 asked for 8 through the environment at import -- but a stream
    // created with a compute-unit mask is given a queue outside that pool.  The mask names every compute unit of the
    // device, so it restricts nothing.  BESSX_TEST_HOOKS=ctx_streams=pool: ordinary streams, round 4's form.
    // (The chains share one priority level: spread over the levels, which have their own pools of hardware queues,
    // the chains on the lower levels ran 2-8 x slower per kernel and the path no faster.)
    const bool own_queue = ctx_stream_create(ps->device, &c->st);
    if (!own_queue) {
      int lo = 0, hi = 0;
      e = hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, hi);
    }
    c->own_hw_queue = own_queue;
  }
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->resblk), ps->res_bytes);
  if (e == hipSuccess) e = hipMemset(c->resblk, 0, ps->res_bytes);
  if (e == hipSuccess) {
    c->ctrl = reinterpret_cast<FitCtrl *>(c->resblk + ((unsigned char *)ps->ctrl - ps->resblk));
    c->sse = reinterpret_cast<double *>(c->resblk + ((unsigned char *)ps->sse - ps->resblk));
    c->b_cur = reinterpret_cast<double *>(c->resblk + ((unsigned char *)ps->b_cur - ps->resblk));
    c->A_cur = reinterpret_cast<int *>(c->resblk + ((unsigned char *)ps->A_cur - ps->resblk));
    e = hipHostMalloc(reinterpret_cast<void **>(&c->res_buf[0]), ps->res_bytes);
  }
  if (e == hipSuccess) {
    std::memset(c->res_buf[0], 0, ps->res_bytes);
    c->res_h = c->res_buf[0];
    e = hipHostMalloc(reinterpret_cast<void **>(&c->stage_h), (size_t)capA * (sizeof(int) + sizeof(double)));
  }
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->pub_flag), 128);
  if (e == hipSuccess) c->pub_flag[0] = c->pub_flag[8] = 0ull;
  if (e == hipSuccess) e = dmalloc(&c->bd, (size_t)p);
  if (e == hipSuccess) e = dmalloc(&c->bd2, (size_t)p);
  if (e == hipSuccess) e = dmalloc(&c->beta_dense, (size_t)p);
  if (e == hipSuccess) e = hipMemset(c->beta_dense, 0, (size_t)p * sizeof(double));
  if (e == hipSuccess) e = dmalloc(&c->inA, (size_t)p);
  if (e == hipSuccess) e = hipMemset(c->inA, 0, (size_t)p);
  if (e == hipSuccess) e = dmalloc(&c->cov_bmm, (size_t)3 * ((p + 31) / 32));
  if (e == hipSuccess) e = dmalloc(&c->sol, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->A_new, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->rdiag, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->zbig, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->cand, 32768);
  if (e == hipSuccess) e = dmalloc(&c->fb_work, CHOL_FB_DOUBLES);
  if (e == hipSuccess) e = dmalloc(&c->tie_buf, (size_t)3 * p + 8);
  if (e == hipSuccess) e = hipMemset(c->tie_buf, 0, 8 * sizeof(int));
  if (e == hipSuccess) c->tie = TopkTie{c->tie_buf, c->tie_buf + 8};
  if (e == hipSuccess) e = dmalloc(&c->hist, (size_t)(ps->max_iter + 2) * ps->hist_stride);
  if (e == hipSuccess) e = dmalloc(&c->hist_beta, (size_t)(ps->max_iter + 2) * ps->hist_stride);
  if (e == hipSuccess) e = dmalloc(&c->hist_coef0, (size_t)(ps->max_iter + 2));
  if (e == hipSuccess) e = dmalloc(&c->init_idx_d, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->init_val_d, (size_t)capA);
  if (e == hipSuccess) e = dmalloc(&c->Gt, (size_t)mt_max * (mt_max + 1) / 2 * 256);
  if (e == hipSuccess) e = dmalloc(&c->cov_fcols, (size_t)capA + 4 * COV_R);
  if (e == hipSuccess) e = dmalloc(&c->cov_extras, (size_t)2 * COV_R);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fold_ctx_free(c);
    return fail(BESSX_ERR_HIP, std::string("fold context: ") + hipGetErrorString(e));
  }
  *out = c;
  return 0;
}

// A chunk chain's fit context (bessx_session::kch): a fold context of the ALL-ROWS row set that runs the parent's whole
// fast path -- results handed over by k_publish, the next candidate chained on the device -- on the parent's cache.
// Beyond a fold context it owns the second result buffer, the snapshots of the deferred publication and the per-row-set
// vectors a fit of row set 0 writes (d, the residual, the work vector of the loss fallback).
}  // extern "C"
namespace bessx {
int chain_ctx_create(bessx_session *ps, bessx_session **out) {
  bessx_session *c = nullptr;
  if (int rc = fold_ctx_create(ps, 0, &c)) return rc;
  c->kch_owner = ps;
  c->kch = nullptr;
  c->kch_index = -1;
  c->kch_ev = nullptr;
  c->kch_fill_st = nullptr;
  c->kch_slot_w = nullptr;
  c->kch_gen_seen = 0;
  c->publish = ps->publish;
  c->chain = ps->chain;
  c->defer_pub = ps->defer_pub;
  c->cov_no_restart = true;  // (the cache holds every column: never needed; and nobody rewrites the map under the others)
  c->fill_hook = nullptr;
  c->tmpv = nullptr;
  c->part_rs[0] = nullptr;
  c->r_rs[0] = nullptr;
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&c->res_buf[1]), ps->res_bytes);
  if (e == hipSuccess) std::memset(c->res_buf[1], 0, ps->res_bytes);
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    e = hipMalloc(reinterpret_cast<void **>(&c->snap[b]), ps->res_bytes + 64);
    if (e == hipSuccess) e = hipMemset(c->snap[b], 0, ps->res_bytes + 64);
  }
  if (e == hipSuccess) e = dmalloc(&c->tmpv, (size_t)ps->ld);
  if (e == hipSuccess) e = dmalloc(&c->part_rs[0], part_elems(ps));
  if (e == hipSuccess) e = dmalloc(&c->r_rs[0], (size_t)ps->ld);
  if (e == hipSuccess) e = hipMemset(c->r_rs[0], 0, (size_t)ps->ld * sizeof(double));
  if (e == hipSuccess && ps->model_type == 1 && !ps->cov_mode) {
    // LM in the streaming form of the score pass (round 5): what enqueue_lm_slot writes besides -- the column list and
    // the slab partials of the Gram kernels, the incremental Gram's row buffer and source map, and the Gram cache of
    // the all-rows row set (two 256 x 256 buffers, the cached column list, its meta words)
    c->ctx_allocs.clear();
    auto own_d = [&](double **dst, size_t count) -> hipError_t {
      *dst = nullptr;
      hipError_t q = dmalloc(dst, count);
      if (q != hipSuccess) return q;
      c->ctx_allocs.push_back(*dst);
      return hipMemset(*dst, 0, count * sizeof(double));
    };
    auto own_n = [&](int **dst, size_t count) -> hipError_t {
      *dst = nullptr;
      hipError_t q = dmalloc(dst, count);
      if (q != hipSuccess) return q;
      c->ctx_allocs.push_back(*dst);
      return hipMemset(*dst, 0, count * sizeof(int));
    };
    e = own_d(&c->gpart, ps->gpart_elems);
    if (e == hipSuccess) e = own_n(&c->gcols, (size_t)ps->capA + 16);
    if (e == hipSuccess) e = own_d(&c->Rt, (size_t)16 * 256);
    if (e == hipSuccess) e = own_n(&c->gsrc, 256);
    if (e == hipSuccess && !c->gcache.empty()) {
      bessx_session::GramCache g;
      e = own_d(&g.g0, (size_t)256 * 256);
      if (e == hipSuccess) e = own_d(&g.g1, (size_t)256 * 256);
      if (e == hipSuccess) e = own_n(&g.A, 256);
      if (e == hipSuccess) e = own_n(&g.meta, 2);
      if (e == hipSuccess) c->gcache[0] = g;
    }
  }
  if (e == hipSuccess && ps->model_type != 1) {
    // the IRLS / Newton families: what their fits write besides -- curvature sums and weights, the IRLS vectors, the
    // auxiliary columns (column 2 is the working response), the slab partials and column lists of the Gram kernels
    c->ctx_allocs.clear();
    auto own = [&](double **dst, size_t count, const double *copy_of) -> hipError_t {
      *dst = nullptr;
      hipError_t q = dmalloc(dst, count);
      if (q != hipSuccess) return q;
      c->ctx_allocs.push_back(*dst);
      return copy_of ? hipMemcpy(*dst, copy_of, count * sizeof(double), hipMemcpyDeviceToDevice)
                     : hipMemset(*dst, 0, count * sizeof(double));
    };
    auto own_i = [&](int **dst, size_t count) -> hipError_t {
      *dst = nullptr;
      hipError_t q = dmalloc(dst, count);
      if (q != hipSuccess) return q;
      c->ctx_allocs.push_back(*dst);
      return hipMemset(*dst, 0, count * sizeof(int));
    };
    const size_t ld = (size_t)ps->ld;
    e = own(&c->part2_rs[0], (size_t)ps->nrb * ps->p, nullptr);
    if (e == hipSuccess) e = own(&c->h_rs[0], ld, nullptr);
    if (e == hipSuccess) e = own(&c->Wv, ld, nullptr);
    if (e == hipSuccess) e = own(&c->llpart, ps->llpart_cap, nullptr);
    if (e == hipSuccess) e = own(&c->bcur, (size_t)ps->capA + 16, nullptr);
    if (e == hipSuccess) e = own(&c->bprev, (size_t)ps->capA + 16, nullptr);
    if (e == hipSuccess) e = own(&c->aux, 3 * ld, ps->aux);
    if (e == hipSuccess) e = own(&c->gpart, ps->gpart_elems, nullptr);
    if (e == hipSuccess) e = own_i(&c->gcols, (size_t)ps->capA + 16);
    if (e == hipSuccess) e = own_i(&c->idcols, (size_t)ps->capA + 16);
    if (e == hipSuccess && ps->idcols)
      e = hipMemcpy(c->idcols, ps->idcols, ((size_t)ps->capA + 16) * sizeof(int), hipMemcpyDeviceToDevice);
    if (e == hipSuccess && ps->model_type == 4) {
      CoxBufs keep = ps->cox;
      c->cox = CoxBufs();
      c->cox.one_pass = keep.one_pass;
      c->cox.need_uv = keep.need_uv;
      c->cox_allocs.clear();
      e = cox_alloc(c);  // (sets hess_fused / fit_clamp as the session's creation did)
      c->cox.hess_fused = c->cox.hess_fused && keep.hess_fused;
      c->cox.fit_clamp = keep.fit_clamp;
      c->cox_state_rs = -1;
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    chain_ctx_free(c);
    return fail(BESSX_ERR_HIP, std::string("chunk chain context: ") + hipGetErrorString(e));
  }
  *out = c;
  return 0;
}

void chain_ctx_free(bessx_session *c) {
  if (!c) return;
  if (c->st) (void)hipStreamSynchronize(c->st);
  if (c->kch_ev) (void)hipEventDestroy(c->kch_ev);
  c->kch_ev = nullptr;
  if (c->res_buf[1]) (void)hipHostFree(c->res_buf[1]);
  for (int b = 0; b < 2; b++)
    if (c->snap[b]) (void)hipFree(c->snap[b]);
  if (c->tmpv) (void)hipFree(c->tmpv);
  if (c->part_rs[0]) (void)hipFree(c->part_rs[0]);
  if (c->r_rs[0]) (void)hipFree(c->r_rs[0]);
  for (void *q : c->ctx_allocs) (void)hipFree(q);
  for (void *q : c->cox_allocs) (void)hipFree(q);
  c->ctx_allocs.clear();
  c->cox_allocs.clear();
  c->res_buf[1] = nullptr;
  c->snap[0] = c->snap[1] = nullptr;
  fold_ctx_free(c);
}
}  // namespace bessx
extern "C" {

int bessx_session_set_cv(bessx_session *s, int K, const int *fold_id, unsigned seed) {
  if (!s || K < 2 || K > s->n) return fail(BESSX_ERR_ARG, "set_cv: bad arguments");
  HIPX(hipSetDevice(s->device));
  const int n = s->n, p = s->p;
  std::vector<int> fold((size_t)n);
  if (fold_id) {
    for (int i = 0; i < n; i++) {
      if (fold_id[i] < 0 || fold_id[i] >= K) return fail(BESSX_ERR_ARG, "set_cv: fold id out of range");
      fold[i] = fold_id[i];
    }
  } else {
    // Metric::set_cv_train_test_mask, src/Metric.h:49-78, with a reproducible generator
    std::vector<int> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 g(seed);
    std::shuffle(perm.begin(), perm.end(), g);
    int size = n / K;
    for (int k = 0; k < K; k++) {
      int b = k * size, e = (k == K - 1) ? n : (k + 1) * size;
      for (int i = b; i < e; i++) fold[perm[i]] = k;
    }
  }
  drop_folds(s);
  s->cache.assign(K + 1, bessx_session::RsCache());
  s->n_train.resize(1);
  s->n_test.assign(K, 0);
  s->K = K;
  s->cv_init.assign(K, SparseVec());
  s->cv_fold = fold;
  // shared fills (see bessx_session::cv_shared): LM in the covariance form, no background fills
  bool share = s->cov_mode && s->model_type == 1 && !s->grouped;
  if (const char *ev = test_hook("cv_shared")) share = share && std::string(ev) != "0";
  // The shared fills need a fold-major second copy of X (every fold padded to whole row slabs).  Whether they are used
  // is settled HERE, before any cache is created as a sharer of row set 0's slot map: the copy must not be much larger
  // than X (many small folds pad badly: K <= n is accepted) and its allocations must succeed -- otherwise the masked
  // per-row-set fills, which need nothing extra, stay in place instead of set_cv failing.
  int sh_nsl = 0, sh_rps = 0;
  long sh_ldp = 0;
  if (share) {
    std::vector<int> cnt((size_t)K, 0);
    for (int i = 0; i < n; i++) cnt[fold[i]]++;
    const int nmax = *std::max_element(cnt.begin(), cnt.end());
    sh_nsl = std::max(1, (nmax + 1023) / 2048);
    sh_rps = ((nmax + sh_nsl - 1) / sh_nsl + 63) / 64 * 64;
    sh_ldp = (long)sh_nsl * sh_rps * K;
    const int pt = (p + 15) / 16, njg = (pt + cov_streamed_tiles_per_wave() - 1) / cov_streamed_tiles_per_wave();
    hipError_t e = sh_ldp * 2 > s->ld * 3 ? hipErrorOutOfMemory : hipSuccess;  // more than 1.5 x the rows of X
    if (e == hipSuccess) e = dmalloc(&s->Xp, (size_t)sh_ldp * p);
    if (e == hipSuccess) e = dmalloc(&s->zp, (size_t)sh_ldp);
    if (e == hipSuccess)
      e = dmalloc(&s->cvp_part, (size_t)COV_SLOT_GROUPS * K * sh_nsl * njg * cov_streamed_tiles_per_wave() * 2 * 256);
    if (e != hipSuccess) {
      (void)hipGetLastError();  // (an allocation failure is not an error of this call)
      (void)hipFree(s->Xp);
      (void)hipFree(s->zp);
      (void)hipFree(s->cvp_part);
      s->Xp = s->zp = s->cvp_part = nullptr;
      share = false;
    }
  }
  std::vector<double> m((size_t)s->ld);
  for (int k = 0; k < K; k++) {
    std::fill(m.begin(), m.end(), 0.0);
    int nt = 0;
    for (int i = 0; i < n; i++) {
      if (fold[i] != k) {
        m[i] = 1.0;
        nt++;
      }
    }
    s->n_test[k] = n - nt;
    if (nt < 1 || n - nt < 1) {
      drop_folds(s);
      return fail(BESSX_ERR_ARG, "set_cv: empty train or test fold");
    }
    // every buffer is handed to its vector as soon as it exists: a failure further down leaves nothing unowned, and
    // drop_folds() (which walks every vector by its own length) returns the session to the no-CV state
#define CVX(expr)                                                                            \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      drop_folds(s);                                                                         \
      return fail(BESSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
    }                                                                                        \
  } while (0)
    auto grow = [&](std::vector<double *> &v, size_t count, bool zero) -> hipError_t {
      double *q = nullptr;
      hipError_t e = dmalloc(&q, count);
      if (e != hipSuccess) return e;
      v.push_back(q);
      return zero ? hipMemset(q, 0, count * sizeof(double)) : hipSuccess;
    };
    CVX(grow(s->mask, (size_t)s->ld, false));
    CVX(hipMemcpy(s->mask.back(), m.data(), (size_t)s->ld * sizeof(double), hipMemcpyHostToDevice));
    CVX(grow(s->xtx, (size_t)p, false));
    CVX(grow(s->xty, (size_t)p, false));
    CVX(grow(s->part_rs, part_elems(s), false));
    CVX(grow(s->r_rs, (size_t)s->ld, true));
    CVX(grow(s->part2_rs, (size_t)s->nrb * p, false));
    CVX(grow(s->h_rs, (size_t)s->ld, true));
    int rc = alloc_gram_cache(s);
    if (rc == 0 && s->cov_mode && !s->grouped) rc = alloc_cov_cache(s, share);  // (grouped: the all-rows fits only)
    if (rc) {
      drop_folds(s);
      return rc;
    }
    if (s->grouped) CVX(grow(s->gxtx_rs, (size_t)s->goff_h[s->N], false));
    s->n_train.push_back(nt);
    if (s->model_type == 1)
      if (int rc2 = prepare_rowset(s, k + 1)) {
        drop_folds(s);
        return rc2;
      }
#undef CVX
  }
  if (share) {
    // fold-major copy: fold k's test rows (ascending) padded with zero rows to cvp_nsl whole slabs of cvp_rps rows
    const long seg = (long)sh_nsl * sh_rps, ldp = sh_ldp;
    std::vector<int> perm((size_t)ldp, -1), fill((size_t)K, 0);
    for (int i = 0; i < n; i++) perm[(size_t)fold[i] * seg + fill[fold[i]]++] = i;
    int *dperm = nullptr;
    hipError_t e = dmalloc(&dperm, (size_t)ldp);
    if (e == hipSuccess) e = hipMemcpy(dperm, perm.data(), (size_t)ldp * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(s->zp, 0, (size_t)ldp * sizeof(double), s->st);
    if (e == hipSuccess) e = launch_rows_permute(s->X, s->ld, p, dperm, ldp, s->Xp, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    (void)hipFree(dperm);
    if (e != hipSuccess) {
      drop_folds(s);
      return fail(BESSX_ERR_HIP, std::string("set_cv (fold-major copy): ") + hipGetErrorString(e));
    }
    s->ldp = ldp;
    s->cvp_rps = sh_rps;
    s->cvp_nsl = sh_nsl;
    s->cv_shared = true;
    // every cache now holds the same columns: start them (and the shared slot map) from empty
    if (int rc = reset_path_caches(s)) {
      drop_folds(s);
      return rc;
    }
    // one fit context per fold: the K chains of a CV evaluation run side by side (fold_fits_side_by_side).  A failed
    // allocation leaves the folds on the parent's own state.
    bool sbs = s->cv_side_by_side && K <= 8 && s->publish && s->fuse && s->cov_cg && s->cg_by_rows && s->fuse_sel;
    if (const char *ev = test_hook("cv_side_by_side")) sbs = sbs && std::string(ev) != "0";
    if (sbs) {
      hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->fill_ctrl), sizeof(FitCtrl));
      if (e == hipSuccess) e = hipMemset(s->fill_ctrl, 0, sizeof(FitCtrl));
      if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s->fill_ctrl_h), sizeof(FitCtrl));
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fill, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_ctx, hipEventDisableTiming);
      for (int k = 0; k < K && e == hipSuccess; k++) {
        bessx_session *c = nullptr;
        if (fold_ctx_create(s, k + 1, &c) != 0) {
          e = hipErrorOutOfMemory;
          break;
        }
        s->fold_ctx.push_back(c);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        drop_fold_contexts(s);
        s->cv_ctx_dropped++;
      } else {
        // The host threads, the hardware queues of the chains' streams and (5.5 ms per stream, one after another:
        // measured) whatever the runtime sets up at a stream's first launch of the slot kernels come into being at their
        // first use -- 30-40 ms on the first evaluation of a path.  Use them once here, where the data is set up anyway:
        // one PDAS slot and a publication per chain, all falling through their gates (slot 5 of a fit that has not begun).
        s->fold_pool = new FoldPool();
        {
      const int dev = s->device;
      s->fold_pool->start(K - 1, [dev] { (void)hipSetDevice(dev); });
    }
        std::vector<hipError_t> we((size_t)K, hipSuccess);
        // (a sparsity level the side-by-side driver would really run on this session: the largest one <= 77 that
        // side_by_side_applies() accepts -- launch geometry and LDS sizes follow from it; none: no warm-up)
        int warm_T0 = 0;
        for (int t = std::min(77, s->cap); t >= 1 && !warm_T0; t--)
          if (side_by_side_applies(s, t)) warm_T0 = t;
        const bool ran = warm_T0 == 0 || s->fold_pool->run([&](int k) {
          bessx_session *c = s->fold_ctx[k];
          unsigned long long seq = 0;
          if (enqueue_lm_slot_cov(c, 5, warm_T0, 0.0, k + 1, false, false, false, nullptr) != 0 ||
              publish_enqueue(c, 1, 0, &seq) != 0)
            we[k] = hipErrorUnknown;
          if (we[k] == hipSuccess) we[k] = hipStreamSynchronize(c->st);
        }, s->wait_deadline_s);
        if (!ran) we[0] = hipErrorUnknown;
        for (hipError_t w : we)
          if (w != hipSuccess) {
            (void)hipGetLastError();
            drop_fold_contexts(s);
            s->cv_ctx_dropped++;  // (visible through bessx_session_counter(s, 11): the folds then run one after another)
            break;
          }
      }
    }
  }
  HIPX(hipStreamSynchronize(s->st));
  return BESSX_OK;
}

int bessx_session_get_cv_folds(const bessx_session *s, int *fold_id) {
  if (!s || !fold_id) return fail(BESSX_ERR_ARG, "null argument");
  if (s->K < 2 || (int)s->cv_fold.size() != s->n) return fail(BESSX_ERR_ARG, "no cross-validation folds set");
  std::copy(s->cv_fold.begin(), s->cv_fold.end(), fold_id);
  return BESSX_OK;
}

int bessx_session_trace_enable(bessx_session *s, int on) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  s->trace.on = on != 0;
  return BESSX_OK;
}

int bessx_session_trace_size(bessx_session *s, int which) {
  if (!s) return -1;
  switch (which) {
    case 0: return (int)s->trace.meta.size();
    case 1: return (int)s->trace.a_flat.size();
    case 2: return (int)s->trace.beta_flat.size();
    case 3: return (int)s->trace.coef0_calls.size();
    case 4: return (int)s->trace.loss_calls.size();
    case 5: return (int)s->trace.ic_calls.size();
  }
  return -1;
}

int bessx_session_trace_copy_int(bessx_session *s, int which, int *out) {
  if (!s || !out) return fail(BESSX_ERR_ARG, "null argument");
  const std::vector<int> &v = which == 0 ? s->trace.meta : s->trace.a_flat;
  if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(int));
  return BESSX_OK;
}

int bessx_session_trace_copy_double(bessx_session *s, int which, double *out) {
  if (!s || !out) return fail(BESSX_ERR_ARG, "null argument");
  const std::vector<double> *v = &s->trace.beta_flat;
  if (which == 3) v = &s->trace.coef0_calls;
  if (which == 4) v = &s->trace.loss_calls;
  if (which == 5) v = &s->trace.ic_calls;
  if (!v->empty()) std::memcpy(out, v->data(), v->size() * sizeof(double));
  return BESSX_OK;
}

int bessx_session_get_normalization(bessx_session *s, double *x_mean, double *x_norm, double *y_mean) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (x_mean) std::copy(s->x_mean_h.begin(), s->x_mean_h.end(), x_mean);
  if (x_norm) std::copy(s->x_norm_h.begin(), s->x_norm_h.end(), x_norm);
  if (y_mean) *y_mean = s->y_mean_h;
  return BESSX_OK;
}

int bessx_session_enable_kernel_timing(bessx_session *s, int on) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  s->timing = on != 0;
  return BESSX_OK;
}

int bessx_session_submodel_steps(bessx_session *s, int reset, long long *steps) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (steps) *steps = s->n_submodel_steps;
  if (reset) s->n_submodel_steps = 0;
  return BESSX_OK;
}

int bessx_session_score_pass_stats(bessx_session *s, int reset, double *seconds, long long *launches,
                                   double *algorithmic_bytes) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (seconds) *seconds = s->k1_seconds;
  if (launches) *launches = s->k1_launches;
  if (algorithmic_bytes) *algorithmic_bytes = s->k1_bytes;
  if (reset) {
    s->k1_seconds = 0.0;
    s->k1_launches = 0;
    s->k1_bytes = 0.0;
    s->panel_w_seconds[0] = s->panel_w_seconds[1] = 0.0;
    s->panel_w_launches[0] = s->panel_w_launches[1] = 0;
  }
  return BESSX_OK;
}

int bessx_session_score_mode(const bessx_session *s) { return s && s->cov_mode ? 2 : 1; }

long long bessx_session_counter(const bessx_session *s, int which) {
  if (!s) return -1;
  switch (which) {
    case 0: return s->chain_hits;
    case 1: return s->cov_cg_fallbacks;
    case 2: return s->cov_panel_groups;
    case 3: return s->chain_queued;
    case 4: return 0;  // (background fills: measured slower in round 2 and removed)
    case 5:
    case 6: return 0;  // (solves from a maintained inverse: measured at parity in round 2 and removed)
    case 7: return s->cv_rounds;
    case 8: return s->cv_union_fills;
    case 9: return s->cov_tie_rescues;
    case 11: return s->cv_ctx_dropped;
    case 12: return (long long)s->fold_ctx.size();
    case 13: return s->shared_wide_fills;
    case 14: return s->kch_paths;
    case 15: return s->kch_refits;
    case 16: return s->kch_chunk_fills;
    case 17: return s->kch_last_chains;
    case 18: return s->kch_giveups;
    case 19: {
      if (s->xtx_ev_pending && s->xtx_ev[0] && s->xtx_ev[1]) {
        float ms = 0.f;
        bessx_session *w = const_cast<bessx_session *>(s);
        if (hipEventSynchronize(s->xtx_ev[1]) == hipSuccess && hipEventElapsedTime(&ms, s->xtx_ev[0], s->xtx_ev[1]) == hipSuccess)
          w->group_xtx_ns = (long long)(1e6 * (double)ms);
        w->xtx_ev_pending = false;
      }
      return s->group_xtx_ns;
    }
    case 20: return s->kch_merged;
    case 21: return s->kch_takeovers;
    case 22: return (long long)(1e6 * s->kch_t[0]);
    case 23: return (long long)(1e6 * s->kch_t[1]);
    case 24: return (long long)(1e6 * s->kch_t[2]);
    case 25: return s->panel_w_launches[0];
    case 26: return (long long)(1e9 * s->panel_w_seconds[0]);
    case 27: return s->panel_w_launches[1];
    case 28: return (long long)(1e9 * s->panel_w_seconds[1]);
    case 29: return s->sp_launches;
    case 30: return s->sp_chain_slots;
    case 31: return s->sp_partial;
    case 32: return ctx_streams_created();
    case 10: {  // times the Gram column cache of the all-rows row set was started over since the last path started
      if (s->cov.empty()) return 0;
      int m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->st) != hipSuccess ||
          hipMemcpy(m, s->cov[0].meta, sizeof(m), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
      return m[3];
    }
    default: return -1;
  }
}

int bessx_session_get_screening(const bessx_session *s, int *columns, int cap) {
  if (!s) return 0;
  for (int j = 0; j < s->p && j < cap && columns; j++) columns[j] = caller_col(s, j);
  return s->p;
}

int bessx_session_get_screening_groups(const bessx_session *s, int *groups, int cap) {
  if (!s) return 0;
  const int cnt = (int)s->screen_groups.size();
  for (int g = 0; g < cnt && g < cap && groups; g++) groups[g] = s->screen_groups[g];
  return cnt;
}

int bessx_session_reset_caches(bessx_session *s) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  HIPX(hipSetDevice(s->device));
  for (auto &v : s->cv_init) v.clear();
  return reset_path_caches(s);
}

int bessx_session_fit_width(const bessx_session *s, int T0) {
  if (!s || T0 < 1 || T0 > s->N) return -1;
  if (!s->grouped) return T0;
  std::vector<int> sz(s->gsz_h);
  std::partial_sort(sz.begin(), sz.begin() + T0, sz.end(), std::greater<int>());
  long w = 0;
  for (int i = 0; i < T0; i++) w += sz[i];
  return (int)std::min<long>(w, s->p);
}


}  // extern "C"
