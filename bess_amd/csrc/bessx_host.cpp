// bessx_host.cpp -- host side of libbessx.so: the session (Data + Algorithm + Metric of the reference's
// bessCpp, resident on the GPU), Algorithm::fit as speculatively enqueued device iterations, the
// path drivers, and the extern "C" ABI of include/bessx.h.
//
// The control flow mirrors the reference so that the two can be read side by side
// (/root/reference): Algorithm::fit src/Algorithm.h:113-171, Metric::{train_loss,test_loss,ic}
// src/Metric.h:138-676, sequential_path / gs_path src/path.cpp:25-389, bessCpp src/bess.cpp:37-214.
// All arithmetic on n- or p-sized data happens in the HIP kernels of bessx_kernels.hip; nothing here
// falls back to a CPU computation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bessx.h"
#include "bessx_dev.h"

namespace bessx {

static thread_local std::string g_err;
// set while a session is created for the marginal fit of one wide group of the screening (screening(),
// src/screening.cpp:42-63): 1 = logit_fit (no weight floor), 2 = cox_fit (linear predictor clamped at 50)
static thread_local int g_marginal_fit_variant = 0;

static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPX(expr)                                                                                  \
  do {                                                                                              \
    hipError_t e__ = (expr);                                                                        \
    if (e__ != hipSuccess)                                                                          \
      return fail(BESSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " (" __FILE__ \
                                                                                      ":" +          \
                                     std::to_string(__LINE__) + ")");                               \
  } while (0)

struct SparseVec {
  std::vector<int> idx;
  std::vector<double> val;
  void clear() {
    idx.clear();
    val.clear();
  }
};

struct Trace {
  bool on = false;
  std::vector<int> meta, a_flat;
  std::vector<double> beta_flat, coef0_calls, loss_calls, ic_calls;
  void clear() {
    meta.clear();
    a_flat.clear();
    beta_flat.clear();
    coef0_calls.clear();
    loss_calls.clear();
    ic_calls.clear();
  }
};

static constexpr int T0_FAST = 254;  // fast path: m + 1 <= 256 lives in the registers of k_chol (with an intercept)
static constexpr int T0_CAP = 2046;   // default capacity of a session (bessx_problem.max_sparsity = 0): m + 2 <= 2048
static constexpr int T0_HARD = 16382;  // largest capacity a session can be created with: m + 2 <= 16384

}  // namespace bessx

using namespace bessx;

// Host threads that queue the chains' launches: launches that alternate between streams cost the host ~10 us each
// (measured: 35 launches per round, 12 ms per path of configs[3]); one thread per chain queues its 7 on its own stream
// while the others do the same.  Workers spin for a job for a while after the last one, then block on a condition
// variable (an idle session holds no core).  The spin is ~4 ms where the host has cores to spare (longer than the
// longest gap inside a path -- a union fill of three groups is 2.5 ms; with 1 ms the workers slept through the fills and
// configs[3] took 32.9 instead of 29.4 ms) and ~0.2 ms where K spinning threads per session would oversubscribe it
// (fewer than 4 hardware threads per chain: several ranks or sessions per host); BESSX_POOL_SPIN_US overrides.
// The caller's wait for its workers is bounded: spin, then sleep on a condition variable, and give up at the
// session's deadline (a worker stuck inside a HIP call) -- the pool is then marked broken and never joined.
struct FoldPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  unsigned ticket = 0;  // (under mu) number of the current job
  std::atomic<unsigned> ticket_hint{0};  // ... its copy for the spinning phase
  std::atomic<int> pending{0};
  bool quit = false, broken = false;
  std::function<void(int)> job;
  int device = 0;
  int spin_iters = 200000;  // pauses of ~40-50 cycles
  void worker(int k) {
    (void)hipSetDevice(device);
    unsigned seen = 0;
    for (;;) {
      bool got = false;
      for (int spin = 0; spin < spin_iters && !got; spin++) {
        got = ticket_hint.load(std::memory_order_acquire) != seen;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      std::function<void(int)> mine;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || ticket != seen; });
        if (quit) return;
        seen = ticket;
        mine = job;
      }
      mine(k);
      if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(mu);  // (the caller may be asleep on cv_done)
        cv_done.notify_all();
      }
    }
  }
  void start(int nworkers, int dev) {
    device = dev;
    const unsigned hw = std::thread::hardware_concurrency();
    spin_iters = (hw >= 4u * (unsigned)(nworkers + 1)) ? 200000 : 10000;
    if (const char *ev = std::getenv("BESSX_POOL_SPIN_US")) spin_iters = std::max(0, std::atoi(ev)) * 50;
    for (int k = 1; k <= nworkers; k++) th.emplace_back([this, k] { worker(k); });
  }
  // runs fn(0) on the caller and fn(1..nworkers) on the workers; true when all are done, false when the workers did
  // not finish within deadline_s (the pool is then broken: its threads may still be inside fn)
  bool run(const std::function<void(int)> &fn, double deadline_s) {
    if (broken) return false;
    {
      std::lock_guard<std::mutex> lk(mu);
      job = fn;
      pending.store((int)th.size(), std::memory_order_relaxed);
      ticket++;
      ticket_hint.store(ticket, std::memory_order_release);
    }
    cv.notify_all();
    fn(0);
    for (int spin = 0; spin < 400000; spin++) {  // ~8 ms: the workers queue a handful of launches each
      if (pending.load(std::memory_order_acquire) == 0) return true;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    std::unique_lock<std::mutex> lk(mu);
    const bool ok = cv_done.wait_for(lk, std::chrono::duration<double>(deadline_s),
                                     [&] { return pending.load(std::memory_order_acquire) == 0; });
    if (!ok) broken = true;
    return ok;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    for (auto &t : th) {
      if (broken)
        t.detach();  // a worker that never came back from a HIP call cannot be joined
      else
        t.join();
    }
    th.clear();
  }
};

struct bessx_session {
  int p_full = 0;                 // columns of the caller's x (p = columns kept by the screening)
  std::vector<int> screen_map;    // kept column -> original column; empty without screening
  // problem
  int n = 0, p = 0;
  long ld = 0;
  int U = 1, nrb = 0;
  int data_type = 1, is_normal = 1, model_type = 1, algorithm_type = 1, max_iter = 20, warm_start = 1;
  int device = 0;
  hipStream_t st = nullptr;
  // device data
  double *X = nullptr, *y = nullptr, *w = nullptr, *aux = nullptr;
  double *x_mean = nullptr, *x_norm = nullptr, *y_mean_d = nullptr;
  unsigned char *always = nullptr;
  // row sets: index 0 = all rows, 1..K = CV training rows of fold k-1
  std::vector<double *> mask, xtx, xty;
  std::vector<double *> part_rs, r_rs;  // per row set: score-pass partial sums and residual of its last fit
  std::vector<double *> part2_rs, h_rs; // GLM: curvature partial sums and curvature weights (r_rs holds g)
  double *logfact = nullptr;            // Poisson: sum_{j<=y_i} log j (src/poisson.cpp:27-41)
  double *Wv = nullptr, *llpart = nullptr, *bcur = nullptr, *bprev = nullptr;  // IRLS work space
  int irls_guess = 8;
  // groups (Data::g_index / g_size, src/Data.h:59-67); grouped == some group has more than one column
  bool grouped = false;
  int N = 0, gmax = 1;
  int g_uniform = 0;  // width of every group when they all have the same one (> 1), else 0: find_ind on the device
  std::vector<int> gidx_h, gsz_h, goff_h;
  int *gidx = nullptr, *gsz = nullptr, *goff = nullptr, *gcols_new = nullptr;
  double *mblk = nullptr, *dcol = nullptr;
  double *mblk2 = nullptr;  // Cox with groups: second term of the per-group blocks
  double *mwork = nullptr, *zwork = nullptr;  // groups wider than 16 columns: Cholesky work copy of the blocks, 2 p vector
  int *allcols = nullptr;   // 0 .. p-1 (column lists of the panels of the Cox group branch)
  std::vector<double *> gxtx_rs;  // per row set: X_g^T diag(mask) X_g blocks (LM)
  int cox_state_rs = -1;
  int dev_state_rs = -1;                // row set of the fit whose final coefficients sit in A_cur/b_cur/beta_dense
  CoxBufs cox = {};                     // Cox work space (model_type 4 only)
  std::vector<void *> cox_allocs;
  int *idcols = nullptr;
  struct RsCache {
    bool valid = false;  // part_rs / r_rs belong to exactly (beta, coef0) below
    bool cov_layout = false;  // part_rs holds d itself (covariance mode), not row-block partial sums
    double lambda = 0.0;      // covariance mode: the scores in bd were formed with this lambda
    int T0 = 0;               // ... by a fit of this sparsity level
    SparseVec beta;
    double coef0 = 0.0;
  };
  std::vector<RsCache> cache;
  std::vector<int> n_train;
  std::vector<double> yy_h;  // per row set: sum m_i y_i^2 of the prepared response (LM loss from the solved system)
  int K = 0;
  // work space
  double *part2 = nullptr, *bd = nullptr, *beta_dense = nullptr, *sol = nullptr;
  double *tmpv = nullptr;
  int *A_new = nullptr, *cand = nullptr, *hist = nullptr, *gcols = nullptr, *info = nullptr;
  double *fb_work = nullptr;  // dense work space of the pivoted fallback solve inside k_chol (sym_pivoted_solve)
  int *tie_buf = nullptr;  // [2 flags | 3 p work ints] of the exact selection behind a score tie (k_topk_ties)
  TopkTie tie = {nullptr, nullptr};
  double *hist_beta = nullptr, *hist_coef0 = nullptr;
  int cap = 0;          // largest sparsity level this session accepts: min(p, T0_CAP)
  int capA = 0;         // array length for T0-sized buffers: cap + 2 rounded up to a tile multiple
  int hist_stride = 0;
  double *rdiag = nullptr, *zbig = nullptr;  // work space of the blocked Cholesky
  std::vector<std::pair<int, GramTask *>> big_tasks;  // task lists for mt > 16, built on demand
  std::vector<int> big_task_cnt;
  size_t cox_M_cols = 0;
  GramTask *gtasks = nullptr;
  std::vector<int> gtask_off, gtask_cnt;  // per mt
  std::vector<int> gtask_inc_off, gtask_inc_cnt;  // per mt: tasks of the extra tile row (incremental LM Gram)
  struct GramCache {
    double *g0 = nullptr, *g1 = nullptr;
    int *A = nullptr, *meta = nullptr;
  };
  std::vector<GramCache> gcache;  // per row set
  // covariance-update mode of the LM score pass (see the k_cov_* kernels): per row set a cache of p-vectors
  // X^T diag(mask) x_a for the columns met so far
  struct CovCache {
    double *G = nullptr;
    int *slot_of = nullptr, *meta = nullptr;
    double *GS = nullptr;  // COV_CS x COV_CS: Gram entries between cached columns, indexed by cache slot (L2-sized)
    double *zero = nullptr;  // a few words that hold 0.0 (CholFuse::zero)
    bool shares_map = false;  // slot_of / meta are row set 0's (shared fills: every row set caches the same columns)
  };
  std::vector<CovCache> cov;
  long long dbg_waits = 0, dbg_waits_ready = 0;  // BESSX_DEBUG: waits for a published block / already there on arrival
  double dbg_enq_s = 0.0;                          // ... seconds spent queueing chained fits
  bool cov_mode = false;
  int cov_cs = 512;        // slots covered by the slot-indexed Gram GS (BESSX_COV_CS <= 512: test hook for the mixed gather)
  double cg_tol = 1e-13;   // accepted relative residual of the conjugate-gradient solve (BESSX_CG_TOL)
  int cov_spec = 32;       // most speculative columns per fill: 64 with the pair panel kernel (variant 4), else 32
  bool fuse_sel = true;    // selection + solve of a slot in one launch, k_sel_cgr (BESSX_FUSE_SEL=0: two launches)
  bool cg_by_rows = true;  // row-dealt kernel k_cgr for systems of up to 208 unknowns (BESSX_CG_LAYOUT=tiles: k_cg)
  // GLM IRLS step in three launches instead of five: linear predictor, weights, working response and the slab Gram
  // in ONE pass over the active columns (k_irls_gram), the reduction, then the convergence test at the head of the
  // solve.  (Round 2's k_gram_irls did the per-row work 64 rows at a time between the barriers of the staging pipeline
  // and lost, 0.180 s against 0.175 s on configs[2]; it is gone.)
  bool irls_fuse = true;   // GLM IRLS step as k_irls_gram + k_gram_reduce + k_chol (BESSX_IRLS_FUSE=0: the five-launch step)
  bool glm_fallback = false;  // the IRLS chain carries the pivoted fallback solve behind every k_chol (set, and the
                              // fit redone, the first time a k_chol of this session meets a rank-deficient system)
  int irls_wfloor = 1;     // floor of the logistic IRLS weight inside the loop (src/Algorithm.h:1188-1192); 0 in the
                           // sub-sessions that run logit_fit for the screening of wide groups (src/logistic.cpp:60-160)
  size_t llpart_cap = 0;
  long long n_submodel_steps = 0;  // IRLS / Newton steps taken since the last reset (bessx_session_submodel_steps)
  bool defer_pub = true;   // chained fits publish through a snapshot + the next launch (BESSX_DEFER_PUBLISH=0: in the tail)
  bool fuse = true;  // small-kernel fusions of the covariance form (SlotFuse); BESSX_FUSE=0 turns them off
  bool cov_cg = true;          // solve by k_cg (falls back to k_chol per slot); BESSX_COV_SOLVER=chol switches it off
  long long cov_cg_fallbacks = 0;
  long long cov_tie_rescues = 0;  // slots redone with the exact tie rule (cov_stall = 3)
  int cov_C = 0;              // cache capacity in columns
  int cov_rps = 0, cov_nslab = 0;
  // shared fills of the CV row sets (LM, covariance form): a fold-major copy Xp of X (rows regrouped by test fold,
  // every fold padded to whole row slabs) lets ONE unmasked pass of the panel kernel serve all K + 1 row sets -- the
  // slab partials of every fold but k sum to fold k's training-row Gram columns, all slabs to the full-data ones
  bool cv_shared = false;
  double *Xp = nullptr, *zp = nullptr, *cvp_part = nullptr;
  long ldp = 0;
  int cvp_rps = 0, cvp_nsl = 0;  // rows per slab, slabs per fold (fold k owns slabs [k * cvp_nsl, (k + 1) * cvp_nsl))
  // The K fold chains of a CV evaluation side by side (LM, covariance form, shared fills; Metric::test_loss,
  // src/Metric.h:150-195, fits the folds one after another, but fold k's fit depends on nothing the others produce):
  // every fold has a CONTEXT of its own -- a bessx_session that borrows the parent's data, row-set vectors and Gram
  // column caches and owns what a fit writes (stream, control / result block, scores, selection and solve work space,
  // host-side warm-start state).  fold_fits_side_by_side() drives them in lock step; whenever chains are parked on
  // missing columns ONE fill (k_cov_fill_union + a pass over the fold-major copy) serves all of them, issued while
  // every chain is quiet, so nobody reads the shared slot map while it is rewritten.  BESSX_CV_SIDE_BY_SIDE=0: the
  // folds are fitted one after another on the parent's own state (round 2's form).
  bessx_session *parent = nullptr;          // set in a fold context
  std::vector<bessx_session *> fold_ctx;    // [k]: context of row set k + 1 (empty: folds run on the parent)
  bool cv_side_by_side = true;
  bool cov_no_restart = false;              // fold context: a full cache parks the fit (cov_stall = 4), the host restarts it
  FitCtrl *fill_ctrl = nullptr;             // gate + statistics block of the union fills (device)
  FitCtrl *fill_ctrl_h = nullptr;           // ... its pinned host copy
  hipEvent_t ev_fill = nullptr, ev_ctx = nullptr;
  long long cv_union_fills = 0, cv_rounds = 0;
  long long cv_ctx_dropped = 0;             // times the fold contexts were given up (allocation / launch failure at set_cv)
  int fill_groups_seen = 0;                 // fill_ctrl->cov_groups already added to cov_panel_groups
  FoldPool *fold_pool = nullptr;            // host threads that queue the chains' launches (one per chain)
  // Every wait of the host on the device (the spin on a published result block, the wait for the chains' host threads)
  // gives up after this many seconds of wall clock and returns BESSX_ERR_HIP with the stream's status: a wedged kernel
  // must not hang the caller at 100 % of a core.  BESSX_WAIT_TIMEOUT_S (read at session creation) overrides; a session
  // that timed out still has work queued on the device and can only be destroyed.
  double wait_deadline_s = 30.0;
  double sbs_t[6] = {0, 0, 0, 0, 0, 0};     // BESSX_DEBUG: seconds in start / enqueue / wait / fill / continue / results
  bool cov_pair_auto = true;  // launches of two groups use the pair panel kernel (BESSX_PANEL_PAIR_AUTO=0: never)
  int cov_variant = 3;        // panel kernel: 3 = one 32-column group per block (k_cov_panel_lds2), two-group launches by the
                              // pair kernel; 4 = the pair kernel whenever it applies, fills speculate up to 64 columns
                              // (BESSX_PANEL_VARIANT=4; measured at parity on configs[1])
  double *cov_part = nullptr, *bd2 = nullptr;
  unsigned char *inA = nullptr;        // 1 for the columns of the current active set
  double *cov_bmm = nullptr;           // per-block min / max of k_cov_d's repeated-set shortcut (+ arg-max columns)
  int bmm_owner = -1;                  // row set of the k_cov_d launch that wrote cov_bmm last
  int *cov_fcols = nullptr, *cov_extras = nullptr;
  long long cov_panel_groups = 0;  // 32-column panel passes over X really executed (host statistics)
  int prefill_cols = 0;            // columns listed by bessx_session_cov_prefill_begin (0: no prefill in progress)
  double *cgb_work = nullptr;      // large-system conjugate gradients (bessx_cgbig.hip): dense matrix + vectors, on first use
  int cgb_cap = 0, cgb_guess = 40; // ... unknowns it holds; step launches queued per solve (adapted to the steps the last took)
  std::vector<std::pair<size_t, int>> cov_timed;  // (event index, first group) of the timed panel launches
  double *Rt = nullptr;
  int *gsrc = nullptr;
  double *gpart = nullptr, *Gt = nullptr;
  size_t gpart_elems = 0;
  int *init_idx_d = nullptr;
  double *init_val_d = nullptr;
  // result block: one D2H copy per host round trip
  unsigned char *resblk = nullptr;
  size_t res_bytes = 0;
  FitCtrl *ctrl = nullptr;
  double *sse = nullptr, *b_cur = nullptr;
  int *A_cur = nullptr;
  int n_sse_blk = 0;
  unsigned char *res_h = nullptr;    // pinned
  bool publish = true;               // results handed over by k_publish (else: asynchronous copy + synchronise)
  unsigned long long *pub_flag = nullptr, pub_seq = 0;  // pinned sequence numbers k_publish releases (one per buffer)
  unsigned char *res_buf[2] = {nullptr, nullptr};       // the two pinned result blocks; res_h points at the current one
  // Deferred publication (chained fits): the last kernel of a chained batch leaves a device snapshot of the result
  // block in snap[buf]; `pend` is the publication that has to follow it -- attached to the first kernel of the next
  // chained fit (second workgroup) or issued as a k_publish launch before the host waits for it.
  unsigned char *snap[2] = {nullptr, nullptr};
  bool pend_on = false;
  PubArgs pend = {};
  // Chained warm-start fits (covariance mode): the path function announces the fit that will follow (hint); the
  // first batch of that fit is queued behind the current one before the host waits for the current result.
  struct Hint {
    bool on = false;
    int T0 = 0;
    double lambda = 0.0;
  } hint;
  struct Ahead {
    bool armed = false;
    int T0 = 0, rs = 0, serial = 0, buf = 0;
    double lambda = 0.0;
    unsigned long long seq = 0;
  } ahead;
  bool chain = true;   // BESSX_CHAIN=0 switches the chaining off
  long long chain_queued = 0, chain_hits = 0, chain_dead = 0, chain_mismatch = 0;
  int fit_serial = 0;
  unsigned char *stage_h = nullptr;  // pinned staging for init vectors
  // host statistics
  std::vector<double> x_mean_h, x_norm_h;
  std::vector<int> cv_fold;  // test fold of every row (bessx_session_get_cv_folds)
  std::vector<int> screen_groups, scr_gidx;  // screening with groups: kept original groups; group index of the kept data
  double y_mean_h = 0.0;
  double nullloss = 0.0;  // Data::get_nullloss (src/Data.h:120-130)
  // Algorithm state (reference member names in comments)
  SparseVec beta;                 // Algorithm::beta
  double coef0 = 0.0;             // Algorithm::coef0
  SparseVec beta_init;            // Algorithm::beta_init
  double coef0_init = 0.0;        // Algorithm::coef0_init
  int sparsity_level = 0;         // Algorithm::sparsity_level
  double lambda_level = 0.0;      // Algorithm::lambda_level
  int cur_rows = 0;               // Algorithm::train_mask (0 = all rows, k+1 = fold k)
  int l = 0;                      // Algorithm::l
  double sse_train = 0.0, sse_test = 0.0;  // of the last fit
  std::vector<SparseVec> cv_init; // Metric::cv_initial_model_param
  std::vector<int> n_test;
  // instrumentation
  Trace trace;
  int metric_depth = 0;
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  double k1_seconds = 0.0, k1_bytes = 0.0;
  long long k1_launches = 0;
  long long n_fits = 0, n_iters = 0;
};

namespace bessx {

template <class T>
static hipError_t dmalloc(T **ptr, size_t count) {
  return hipMalloc(reinterpret_cast<void **>(ptr), std::max<size_t>(count, 1) * sizeof(T));
}

// A fold context (bessx_session::fold_ctx) owns only what a fit writes; everything else is the parent's.
static void fold_ctx_free(bessx_session *c) {
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
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

static void drop_fold_contexts(bessx_session *s) {
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

static void session_free(bessx_session *s) {
  if (!s) return;
  if (std::getenv("BESSX_DEBUG") && !s->fold_ctx.empty())
    std::fprintf(stderr, "[bessx] fold chains side by side: %lld rounds, %lld union fills; ms in start %.2f, enqueue %.2f, "
                 "wait %.2f, fill %.2f, continue %.2f, results %.2f\n", s->cv_rounds, s->cv_union_fills, s->sbs_t[0] * 1e3,
                 s->sbs_t[1] * 1e3, s->sbs_t[2] * 1e3, s->sbs_t[3] * 1e3, s->sbs_t[4] * 1e3, s->sbs_t[5] * 1e3);
  drop_fold_contexts(s);
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
  if (s->st) (void)hipStreamDestroy(s->st);
  delete s;
}

// doubles in the score-pass partial sums of one row set
static size_t part_elems(const bessx_session *s) {
  const size_t plane = (size_t)s->nrb * (size_t)s->p;
  return (s->model_type == 4 && s->cox.one_pass) ? 5 * plane + (size_t)s->nrb : plane;
}

// timing of the dominant kernel: event pairs on the session stream, resolved lazily
static int k1_begin(bessx_session *s, hipEvent_t *a, hipEvent_t *b) {
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
static int k1_collect(bessx_session *s, const std::vector<std::pair<size_t, bool>> &pairs) {
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

// the same for the panel launches of the covariance mode: a launch covers up to 2 groups of 32 columns, each group
// is one pass over X; nfill = length of the fill list the launches worked on
static int cov_collect(bessx_session *s, int nfill) {
  if (!s->timing) return 0;
  for (auto &pr : s->cov_timed) {
    const int real = std::min(2, std::max(0, nfill / 32 - pr.second));
    if (real == 0) continue;
    float ms = 0.f;
    HIPX(hipEventElapsedTime(&ms, s->ev_pool[pr.first], s->ev_pool[pr.first + 1]));
    s->k1_seconds += (double)ms * 1e-3;
    s->k1_launches += 1;
    s->k1_bytes += 8.0 * (double)s->n * (double)s->p * real;
  }
  s->cov_timed.clear();
  s->ev_used = 0;
  return 0;
}

static int alloc_gram_cache(bessx_session *s) {
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

static constexpr int COV_R = 32;        // columns per panel group (matches the kernels)
static constexpr int COV_SLOT_GROUPS = 2;  // groups an ordinary PDAS slot launches
static constexpr int COV_CS = 512;         // side of the slot-indexed Gram of the cached columns (2 MiB)

static int alloc_cov_cache(bessx_session *s, bool share_map = false) {
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
static int reset_path_caches(bessx_session *s) {
  if (s->ahead.armed) {
    s->ahead.armed = false;
    HIPX(hipStreamSynchronize(s->st));
  }
  s->pend_on = false;  // a deferred publication of a fit nobody will ask for
  s->hint.on = false;
  for (auto &c : s->cache) c.valid = false;
  s->dev_state_rs = -1;
  for (bessx_session *c : s->fold_ctx) {
    HIPX(hipStreamSynchronize(c->st));
    for (auto &cc : c->cache) cc.valid = false;
    c->dev_state_rs = -1;
  }
  for (auto &g : s->gcache) HIPX(hipMemsetAsync(g.meta, 0, 2 * sizeof(int), s->st));
  for (auto &c : s->cov) {
    HIPX(hipMemsetAsync(c.slot_of, 0xff, (size_t)s->p * sizeof(int), s->st));
    HIPX(hipMemsetAsync(c.meta, 0, 8 * sizeof(int), s->st));
  }
  return 0;
}

// k_chol outside the covariance form: only the work space of its pivoted fallback solve rides in the fuse block
static CholFuse chol_fallback_only(const bessx_session *s) {
  CholFuse fz = {};
  fz.fb_work = s->fb_work;
  return fz;
}

static void build_gram_tasks(int mt, std::vector<GramTask> &out) {
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
static int gram_tasks_for(bessx_session *s, int mt, const GramTask **tasks, int *ntask) {
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

static void gram_geometry(const bessx_session *s, int ntask, int *rows_per_slab, int *nslab, int ntiles = 0,
                          bool allow_lds = true) {
  if (allow_lds && ntiles > 0 && gram_lds_applies(ntiles, 0) && s->ld >= 64) {
    // LDS-staged kernel: one block per slab computes every tile; slabs are whole 64-row chunks, about one block
    // (4 or 8 waves) per CU
    // (4-wave instance, up to 8 tile rows: two blocks fit a CU; measured 512 >= 256 > 128 slabs on configs[2])
    static const long want = [] {
      const char *ev = std::getenv("BESSX_GRAM_SLABS");
      return ev ? std::max(1L, std::atol(ev)) : 0L;
    }();
    long ns = std::min<long>(want > 0 ? want : (ntiles <= 36 ? 512 : 256), s->ld / 64);
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
static int upload_x(bessx_session *s, const double *x, int col_major) {
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
static int prepare_rowset(bessx_session *s, int rs) {
  const double *m = s->mask[rs];
  // tmpv = m*y (or y), v2 = m (or ones on data rows = aux column 1)
  if (launch_vec_mul(s->y, m, s->ld, s->tmpv, s->st) != hipSuccess) return fail(BESSX_ERR_HIP, "vec_mul");
  const double *v2 = m ? m : s->aux + s->ld;
  hipError_t e = launch_xtv(s->X, s->ld, s->p, s->U, s->tmpv, v2, s->part_rs[rs], s->part2, nullptr, 0, s->st);
  if (e == hipSuccess) e = launch_part_sum(s->part_rs[rs], s->nrb, s->p, s->xty[rs], s->st);
  if (e == hipSuccess) {
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
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("prepare_rowset: ") + hipGetErrorString(e));
  return 0;
}

// --------------------------------------------------------------------------------------------
// Algorithm::fit (src/Algorithm.h:113-171), LM: GroupPdasLm::get_A / primary_model_fit (:1097-1135)
// --------------------------------------------------------------------------------------------
static int enqueue_lm_slot(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                           std::vector<std::pair<size_t, bool>> &k1_pairs) {
  const int mt = (T0 + 1 + 15) / 16, mp = mt * 16;
  const int ntiles = mt * (mt + 1) / 2;
  const GramTask *tasks_full = nullptr;
  int ntask = 0, rps, nslab;
  if (int rc = gram_tasks_for(s, mt, &tasks_full, &ntask)) return rc;
  gram_geometry(s, ntask, &rps, &nslab, ntiles, mt > 16);  // (the cached LM Gram keeps k_gram: gates 3 / 4)
  if ((size_t)nslab * ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
  hipError_t e = hipSuccess;
  if (!skip_k1) {
    // skip_k1: the partial sums of this row set were computed from exactly the coefficients this fit
    // starts from (the previous fit ended on a repeated active set) -- get_A would recompute them bit for bit.
    hipEvent_t ea = nullptr, eb = nullptr;
    if (int rc = k1_begin(s, &ea, &eb)) return rc;
    e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], nullptr, s->part_rs[rs], nullptr, s->ctrl, slot, s->st);
    if (s->timing && e == hipSuccess) {
      e = hipEventRecord(eb, s->st);
      k1_pairs.push_back({s->ev_used - 2, false});
    }
  } else if (s->timing) {
    k1_pairs.push_back({(size_t)-1, false});
  }
  if (e == hipSuccess)
    e = launch_score(s->part_rs[rs], nullptr, s->nrb, s->p, s->beta_dense, s->xtx[rs], (double)s->n_train[rs],
                     lambda, 0, s->always, s->bd, s->ctrl, slot, s->st);
  if (e == hipSuccess) e = launch_topk(s->bd, s->p, T0, s->A_new, s->cand, s->ctrl, slot, s->st, nullptr, nullptr, &s->tie);
  if (e == hipSuccess) e = launch_gram_cols(s->A_new, T0, mp, 0, 0, s->gcols, s->ctrl, slot, s->A_cur, 1, s->st);
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
static int panel_variant_for(const bessx_session *s, int ng) {
  return (s->cov_variant == 3 && ng == 2 && s->cov_pair_auto) ? 4 : s->cov_variant;
}

// gfirst / compact: a cooperative prefill (bessx_session_cov_prefill_*) forms only SOME groups of the list here and
// fills the slot-indexed Gram GS once every group is in (its own and the ones imported from the other ranks)
static int enqueue_cov_fill(bessx_session *s, int rs, int ngroups, int parked, const FitCtrl *gate = nullptr,
                            int gfirst = 0, bool compact = true) {
  bessx_session::CovCache &cv = s->cov[rs];
  const FitCtrl *gc = gate ? gate : s->ctrl;  // whose cov_stall / cov_nfill the launches look at
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
      e = launch_cov_reduce(s->cov_part, s->p, s->cov_fcols, cv.slot_of, cv.G, g0, ng, s->cov_nslab, gc, parked,
                            s->st);
    if (e == hipSuccess && compact)  // entries between cached columns, by slot: what the solve gathers from
      e = launch_cov_compact(cv.G, s->p, cv.slot_of, s->cov_fcols, g0, ng, cv.GS, s->cov_cs, gc, parked, s->st, s->xtx[rs],
                             cv.meta);
    if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov panel: ") + hipGetErrorString(e));
  }
  return 0;
}

static bool cov_speculates(const bessx_session *s);
// the capacity the device-side lookup (cov_need_body: "count + len + 32 > C -> start the cache over") is told: a fill
// can add up to cov_spec speculative columns beyond the requested ones
static int cov_C_dev(const bessx_session *s) { return s->cov_C - (s->cov_spec - COV_R); }

// Fusions of the small kernels around a slot (BESSX_FUSE=0 turns them off):
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
static CholFuse cov_fuse_args(bessx_session *s, int rs, int T0, bool force_chol, SlotFuse *sf) {
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
static int cgb_reserve(bessx_session *s) {
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
static int enqueue_cov_tail(bessx_session *s, int slot, int T0, double lambda, int rs, bool force_chol = false,
                            SlotFuse *sf = nullptr) {
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

static bool cov_speculates(const bessx_session *s) { return topk_supported(s->p, s->cov_spec) && s->p >= 2 * s->cov_spec; }

// skip_d: d of exactly the starting coefficients is in memory (previous fit of the chain); scores_ok: so are the
// sacrifice scores (same lambda), nothing to recompute before the selection.
// grow1: ... and the previous fit (same row set, the last thing the device ran) had sparsity level T0 - 1 and ended
// with A_cur = max_k of these very scores: the first selection is A_cur plus one arg-max (k_topk).
static int enqueue_lm_slot_cov(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_d,
                               bool scores_ok = false, bool grow1 = false, SlotFuse *sf = nullptr) {
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
static int cov_unpark(bessx_session *s, const FitCtrl *hc, int T0, double lambda, int rs, int *next_slot) {
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
  if (spec) e = launch_topk(s->bd2, s->p, s->cov_spec, s->cov_extras, s->cand, nullptr, 0, s->st);
  if (e == hipSuccess)
    e = launch_cov_fill_list(s->cov_fcols, s->cov_extras, s->bd2, cv.slot_of, cv.meta, s->ctrl, 1, s->st, s->cov_spec,
                             spec ? 1 : 0);
  if (e != hipSuccess) return fail(BESSX_ERR_HIP, std::string("cov_unpark: ") + hipGetErrorString(e));
  // upper bound of the list length (the device drops speculative columns that turn out to be cached already)
  const int room = spec ? std::min(((nm + s->cov_spec / 2 + s->cov_spec - 1) / s->cov_spec) * s->cov_spec - nm, s->cov_spec) : 0;
  const int ngroups = (nm + room + COV_R - 1) / COV_R;
  if (int rc = enqueue_cov_fill(s, rs, ngroups, 1)) return rc;
  HIPX(launch_cov_resume(s->ctrl, s->st));
  if (int rc = enqueue_cov_tail(s, stalled, T0, lambda, rs)) return rc;
  *next_slot = stalled + 1;
  return 0;
}

static int glm_geometry(bessx_session *s, int T0, int *mt, int *mp, int *ntask, int *ntiles, int *rps, int *nslab) {
  *mt = (T0 + 2 + 15) / 16;  // intercept + T0 columns + the working response
  *mp = *mt * 16;
  const GramTask *tk = nullptr;
  if (int rc = gram_tasks_for(s, *mt, &tk, ntask)) return rc;
  *ntiles = *mt * (*mt + 1) / 2;
  gram_geometry(s, *ntask, rps, nslab, *ntiles);
  if ((size_t)*nslab * *ntiles * 256 > s->gpart_elems) return fail(BESSX_ERR_ARG, "gram workspace too small");
  return 0;
}

static int enqueue_glm_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs) {
  const int fam = s->model_type;
  int mt, mp, ntask, ntiles, rps, nslab;
  if (int rc = glm_geometry(s, T0, &mt, &mp, &ntask, &ntiles, &rps, &nslab)) return rc;
  hipError_t e = hipSuccess;
  if (!skip_k1) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (int rc = k1_begin(s, &ea, &eb)) return rc;
    e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], s->h_rs[rs], s->part_rs[rs], s->part2_rs[rs], s->ctrl, slot,
                   s->st);
    if (s->timing && e == hipSuccess) {
      e = hipEventRecord(eb, s->st);
      k1_pairs.push_back({s->ev_used - 2, false});
    }
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

static int enqueue_glm_irls_step(bessx_session *s, int slot, int t, int T0, double lambda, int rs) {
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

static int enqueue_glm_tail(bessx_session *s, int slot, int T0, int rs) {
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
static int cox_reserve(bessx_session *s, int T0) {
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

static int enqueue_cox_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs) {
  const int mt = (T0 + 1 + 15) / 16, mp = mt * 16;
  hipError_t e = hipSuccess;
  if (!skip_k1) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (int rc = k1_begin(s, &ea, &eb)) return rc;
    e = launch_cox_score_pass(s->X, s->ld, s->p, s->U, s->nrb, s->cox, s->part_rs[rs], s->part2_rs[rs], s->ctrl, slot,
                              s->st);
    if (s->timing && e == hipSuccess) {
      e = hipEventRecord(eb, s->st);
      k1_pairs.push_back({s->ev_used - 2, false});
    }
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

static int enqueue_cox_newton(bessx_session *s, int slot, int t, int T0, double lambda, int rs) {
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

static int enqueue_cox_tail(bessx_session *s, int slot, int T0, int rs) {
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
// no copy engine, no interrupt.  kcopy < 0 (or BESSX_PUBLISH=0): plain asynchronous copy + stream synchronisation.
// what a publication of the result block into pinned buffer `buf` copies; takes the next sequence number
static PubArgs publish_args(bessx_session *s, int kcopy, int buf, unsigned long long *seq) {
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
static PubArgs publish_from_snapshot(const PubArgs &tail) {
  PubArgs pa = tail;
  pa.on = 1;
  pa.dev = tail.snap;
  pa.count_ptr = reinterpret_cast<const int *>(tail.snap + tail.snap_count_off);
  return pa;
}

// a deferred publication nobody has attached to a launch: issue it as a launch of its own
static int publish_flush(bessx_session *s) {
  if (!s->pend_on) return 0;
  s->pend_on = false;
  const PubArgs &pa = s->pend;
  HIPX(launch_publish(pa.dev, pa.host, pa.ctrl_bytes, pa.off_sse, pa.n_sse, pa.off_b, pa.off_a, pa.kcopy, pa.seq_host,
                      pa.seq, s->st, pa.count_ptr));
  return 0;
}

static int publish_launch(bessx_session *s, const PubArgs &pa) {
  HIPX(launch_publish(pa.dev, pa.host, pa.ctrl_bytes, pa.off_sse, pa.n_sse, pa.off_b, pa.off_a, pa.kcopy, pa.seq_host,
                      pa.seq, s->st, pa.count_ptr));
  return 0;
}

static int publish_enqueue(bessx_session *s, int kcopy, int buf, unsigned long long *seq) {
  return publish_launch(s, publish_args(s, kcopy, buf, seq));
}

static int publish_wait(bessx_session *s, int buf, unsigned long long want) {
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

static int read_results(bessx_session *s, int kcopy = -1) {
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
static int read_results(bessx_session *s, int kcopy);

// --------------------------------------------------------------------------------------------
// Group mode (some group has more than one column): Algorithm::fit with per-group sacrifices.  The host reads the
// selected group ids back after the top-k to expand them into columns (find_ind, src/utilities.cpp:113-130), so
// this loop synchronises twice per PDAS iteration and uses none of the speculative / cached fast paths.
// --------------------------------------------------------------------------------------------
static int algorithm_fit_grouped(bessx_session *s) {
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
  hipError_t e = launch_fit_begin(s->ctrl, T0, k_init, s->init_idx_d, s->init_val_d, s->coef0_init, s->A_cur, s->b_cur,
                                  s->beta_dense, s->p, s->hist, s->st);
  if (e == hipSuccess) {
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
  s->cache[rs].valid = false;
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
    while (slot <= s->max_iter) {
      const int first = slot;
      for (int b = 0; b < 2 && slot <= s->max_iter; b++, slot++) {
        hipEvent_t ea = nullptr, eb = nullptr;
        if (int rc = k1_begin(s, &ea, &eb)) return rc;
        e = launch_xtv(s->X, s->ld, s->p, s->U, s->r_rs[rs], nullptr, s->part_rs[rs], nullptr, s->ctrl, slot, s->st);
        if (s->timing && e == hipSuccess) {
          e = hipEventRecord(eb, s->st);
          k1_pairs.push_back({s->ev_used - 2, false});
        }
        if (e == hipSuccess)
          e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->part_rs[rs], s->nrb, s->p, 1,
                                 (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork,
                                 s->zwork, s->ctrl, slot);
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
      if (e == hipSuccess)
        e = launch_group_score(s->N, s->gidx, s->gsz, s->goff, s->gxtx_rs[rs], nullptr, s->part_rs[rs], s->nrb, s->p, 1,
                               (double)s->n_train[rs], lambda, s->beta_dense, s->always, s->bd, s->st, s->gmax, s->mwork,
                               s->zwork);
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
    s->cache[rs].valid = false;
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
static int enqueue_chained(bessx_session *s, const bessx_session::Hint &hint, int rs, int parent, int buf, int batch,
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
static void fold_contexts_invalidate(bessx_session *s);

static int algorithm_fit(bessx_session *s) {
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
  cc.valid = false;
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
  } else if (use_cache && s->dev_state_rs == rs) {
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
                         s->beta_dense, s->p, s->hist, s->st, s->inA);
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
  if (cov && !use_cache && k_init > 0) {
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
  while (!glm && !cov) {
    int first = slot;
    for (int b = 0; b < batch && slot <= s->max_iter; b++, slot++)
      if (int rc = enqueue_lm_slot(s, slot, T0, lambda, rs, use_cache && slot == 1, k1_pairs)) return rc;
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
    s->cache[rs].valid = false;
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
  cc.valid = hc->done && hc->d_fresh;
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

// --------------------------------------------------------------------------------------------
// The K fold fits of a CV evaluation side by side (Metric::test_loss, src/Metric.h:150-195; see
// bessx_session::fold_ctx).  Every fold context runs the same slots algorithm_fit() would queue for it -- warm start
// from the fold's previous coefficients (:177-188), batches of two PDAS iterations, the parked-fit protocol -- on its
// own stream; the host enqueues a batch for every chain, waits for all of them, and serves every chain that is parked
// on missing Gram columns with ONE fill while all chains are quiet.
// --------------------------------------------------------------------------------------------
static void fold_contexts_invalidate(bessx_session *s) {
  for (bessx_session *c : s->fold_ctx) {
    for (auto &cc : c->cache) cc.valid = false;
    c->dev_state_rs = -1;
  }
}

static bool side_by_side_applies(const bessx_session *s, int T0) {
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
static int fold_fits_side_by_side(bessx_session *s, double *out, const std::vector<int> *only = nullptr,
                                  double *per_fold = nullptr) {
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
  for (int r = 1; r <= K; r++) s->cache[r].valid = false;  // the fold row sets' state now lives in the contexts
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
    cc.valid = false;
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
    s->fold_pool->start(K - 1, s->device);
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
        static const int spec_min_env = [] {
          const char *ev = std::getenv("BESSX_CV_SPEC_MIN");
          return ev ? std::atoi(ev) : -1;
        }();
        const int spec_min = spec_min_env >= 0 ? std::min(spec_min_env, s->cov_spec) : s->cov_spec / 2;
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
static double metric_train_loss_value(const bessx_session *s) {
  // LmMetric::train_loss, src/Metric.h:145-148: ||y - X beta||^2 / n on ALL rows (train + test rows of the mask)
  if (s->model_type == 1) return (s->sse_train + s->sse_test) / (double)s->n;
  // Logistic / Poisson / Cox train_loss, src/Metric.h:266-290, :426-440, :565-568: -2 * (sum over ALL rows),
  // the sum being kept in sse_train
  return -2.0 * s->sse_train;
}

static double metric_fold_test_loss(const bessx_session *s, int k) {
  if (s->model_type == 1) return s->sse_test / (double)(2 * s->n_test[k]);  // src/Metric.h:190
  if (s->model_type == 2 || s->model_type == 4) return -2.0 * s->sse_test;  // :349-351 (clamp +-25), :609 Cox
  return -s->sse_test;                                                      // :489 Poisson
}

static int metric_train_loss(bessx_session *s, double *out) {
  *out = metric_train_loss_value(s);
  if (s->metric_depth == 0 && s->trace.on) s->trace.loss_calls.push_back(*out);
  return 0;
}

// test_loss under CV: K fold fits, src/Metric.h:150-195
static int metric_test_loss(bessx_session *s, double *out) {
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
static int metric_ic(bessx_session *s, int ic_type, int is_cv, double *out) {
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

// --------------------------------------------------------------------------------------------
// paths (src/path.cpp)
// --------------------------------------------------------------------------------------------
struct Candidate {
  int T0;
  double lambda;
  SparseVec beta;
  double coef0, loss, ic;
  int iters;
};

static void denormalize(const bessx_session *s, SparseVec &b, double &coef0, bool gs_variant) {
  // src/path.cpp:76-110 (sequential) and :330-342 (golden section: data_type 3 also takes the "else")
  if (!s->is_normal) return;
  double dot = 0.0, sn = std::sqrt((double)s->n);
  for (size_t i = 0; i < b.idx.size(); i++) {
    b.val[i] = sn * b.val[i] / s->x_norm_h[b.idx[i]];
    dot += b.val[i] * s->x_mean_h[b.idx[i]];
  }
  if (s->data_type == 1)
    coef0 = s->y_mean_h - dot;
  else if (s->data_type == 2 || gs_variant)
    coef0 = coef0 - dot;
}

static int run_fit(bessx_session *s, int T0, double lambda, const SparseVec &beta_init, double coef0_init) {
  s->cur_rows = 0;  // update_train_mask(full_mask) + update_group_XTX(full_group_XTX)
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init = beta_init;
  s->coef0_init = coef0_init;
  return algorithm_fit(s);
}

static inline int caller_col(const bessx_session *s, int j) { return s->screen_map.empty() ? j : s->screen_map[j]; }

static void store_candidate(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant) {
  int i = res->n_candidates++;
  if (i >= res->capacity) return;
  SparseVec b = c.beta;
  double c0 = c.coef0;
  denormalize(s, b, c0, gs_variant);
  if (res->cand_T0) res->cand_T0[i] = c.T0;
  if (res->cand_lambda) res->cand_lambda[i] = c.lambda;
  if (res->cand_iters) res->cand_iters[i] = c.iters;
  if (res->cand_train_loss) res->cand_train_loss[i] = c.loss;
  if (res->cand_ic) res->cand_ic[i] = c.ic;
  if (res->cand_coef0) res->cand_coef0[i] = c0;
  for (int j = 0; j < res->max_T0; j++) {
    bool has = j < (int)b.idx.size();
    if (res->cand_support) res->cand_support[(size_t)i * res->max_T0 + j] = has ? caller_col(s, b.idx[j]) : -1;
    if (res->cand_beta) res->cand_beta[(size_t)i * res->max_T0 + j] = has ? b.val[j] : 0.0;
  }
}

static void store_best(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant) {
  SparseVec b = c.beta;
  double c0 = c.coef0;
  denormalize(s, b, c0, gs_variant);
  if (res->beta) {
    std::fill(res->beta, res->beta + s->p_full, 0.0);  // beta_screening_A of src/bess.cpp:186-197
    for (size_t i = 0; i < b.idx.size(); i++) res->beta[caller_col(s, b.idx[i])] = b.val[i];
  }
  res->coef0 = c0;
  res->train_loss = c.loss;
  res->ic = c.ic;
  res->lambda = c.lambda;
  res->best_T0 = c.T0;
  res->best_iters = c.iters;
}

// does candidate c equal row `row` of the chain's stop table (supports in the caller's numbering, -1 padded;
// coefficients de-normalised like cand_beta, compared to stop_rtol when given)?
static bool chain_row_matches(const bessx_session *s, const bessx_path_chain *ch, int row, const Candidate &c) {
  if (!ch->stop_support || row >= ch->stop_rows) return false;
  const int *want = ch->stop_support + (size_t)row * ch->stop_row_len;
  const int k = (int)c.beta.idx.size();
  if (k > ch->stop_row_len) return false;
  for (int j = 0; j < ch->stop_row_len; j++)
    if (want[j] != (j < k ? caller_col(s, c.beta.idx[j]) : -1)) return false;
  if (ch->stop_beta) {
    SparseVec b = c.beta;
    double c0 = c.coef0;
    denormalize(s, b, c0, false);
    const double *wb = ch->stop_beta + (size_t)row * ch->stop_row_len;
    for (int j = 0; j < k; j++)
      if (!(std::fabs(b.val[j] - wb[j]) <= ch->stop_rtol * std::max(std::fabs(b.val[j]), std::fabs(wb[j])))) return false;
  }
  return true;
}

static int sequential_path(bessx_session *s, const int *seq, int ns, const double *lam, int nl, int ic_type,
                           int is_cv, bessx_path_result *res, bessx_path_chain *chain = nullptr) {
  // src/path.cpp:25-132
  SparseVec beta_init;
  double coef0_init = 0.0;
  if (chain) {
    // the warm-start chain of :60-64 continued from a model another process holds (update_beta_init /
    // update_coef0_init, src/Algorithm.h:85-93, before the first candidate)
    for (int i = 0; i < chain->init_len; i++) {
      if (chain->init_idx[i] < 0 || chain->init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "chain: init index out of range");
      beta_init.idx.push_back(chain->init_idx[i]);
      beta_init.val.push_back(chain->init_val[i]);
    }
    coef0_init = chain->init_coef0;
    chain->stopped_at = -1;
    chain->last_len = 0;
    chain->last_coef0 = 0.0;
  }
  std::vector<Candidate> grid((size_t)ns * nl);
  std::vector<char> have((size_t)ns * nl, 0);
  bool stop = false;
  for (int i = 0; i < ns && !stop; i++) {
    int step = (i % 2 == 0) ? 1 : -1;
    for (int j = (i % 2 == 0) ? 0 : nl - 1; j < nl && j >= 0 && !stop; j += step) {
      {
        // announce the fit that follows in the snake order (src/path.cpp:36-50): it can be chained on the device
        int jn = j + step, in = i;
        if (jn < 0 || jn >= nl) {
          in = i + 1;
          jn = (in % 2 == 0) ? 0 : nl - 1;
        }
        s->hint.on = in < ns && !is_cv;
        if (s->hint.on) {
          s->hint.T0 = seq[in];
          s->hint.lambda = lam[jn];
        }
      }
      if (int rc = run_fit(s, seq[i], lam[j], beta_init, coef0_init)) return rc;
      if (s->warm_start) {
        beta_init = s->beta;
        coef0_init = s->coef0;
      }
      Candidate &c = grid[(size_t)j * ns + i];
      have[(size_t)j * ns + i] = 1;
      c.T0 = seq[i];
      c.lambda = lam[j];
      c.beta = s->beta;
      c.coef0 = s->coef0;
      c.iters = s->l;
      if (int rc = metric_train_loss(s, &c.loss)) return rc;
      if (int rc = metric_ic(s, ic_type, is_cv, &c.ic)) return rc;
      store_candidate(s, res, c, false);
      if (chain && chain_row_matches(s, chain, res->n_candidates - 1, c)) {
        // from here on the chain the caller already holds IS this chain: same model, same successor
        chain->stopped_at = res->n_candidates - 1;
        stop = true;
      }
    }
  }
  size_t best = 0;  // minCoeff over the column-major (ns x nl) matrix: first minimum in storage order
  while (best < grid.size() && !have[best]) best++;
  for (size_t q = best; q < grid.size(); q++)
    if (have[q] && grid[q].ic < grid[best].ic) best = q;
  store_best(s, res, grid[best], false);
  if (chain) {
    // Algorithm::beta / coef0 as the path would hand them to the next candidate (normalised scale)
    chain->last_len = (int)beta_init.idx.size();
    chain->last_coef0 = coef0_init;
    for (int i = 0; i < chain->last_len && i < chain->last_cap; i++) {
      if (chain->last_idx) chain->last_idx[i] = beta_init.idx[i];
      if (chain->last_val) chain->last_val[i] = beta_init.val[i];
    }
  }
  return 0;
}

static int gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res) {
  // src/path.cpp:134-389; lambda stays at its constructor default 0
  SparseVec beta_init;
  double coef0_init = 0.0;
  int Tmin = s_min, Tmax = s_max;
  int T1 = (int)std::round(0.618 * Tmin + 0.382 * Tmax), T2 = (int)std::round(0.382 * Tmin + 0.618 * Tmax);
  double ic1 = 0, ic2 = 0, icT1 = 0, icT2 = 0;
  auto fit_point = [&](int T, double *ic_first, double *ic_second) -> int {
    if (int rc = run_fit(s, T, 0.0, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    Candidate c;
    c.T0 = T;
    c.lambda = 0.0;
    c.beta = s->beta;
    c.coef0 = s->coef0;
    c.iters = s->l;
    if (int rc = metric_train_loss(s, &c.loss)) return rc;
    if (int rc = metric_ic(s, ic_type, is_cv, &c.ic)) return rc;
    store_candidate(s, res, c, true);
    *ic_first = c.ic;
    if (ic_second)
      if (int rc = metric_ic(s, ic_type, is_cv, ic_second)) return rc;  // evaluated twice, :204+:210 etc.
    return 0;
  };
  if (int rc = fit_point(T1, &ic1, nullptr)) return rc;
  icT1 = ic1;
  if (int rc = fit_point(T2, &ic2, &icT2)) return rc;
  while (T1 != T2) {
    if (icT1 < icT2) {
      Tmax = T2;
      T2 = T1;
      ic2 = ic1;
      icT2 = ic1;
      T1 = (int)std::round(0.618 * Tmin + 0.382 * Tmax);
      if (int rc = fit_point(T1, &ic1, &icT1)) return rc;
    } else {
      Tmin = T1;
      T1 = T2;
      ic1 = ic2;
      icT1 = ic2;
      T2 = (int)std::round(0.382 * Tmin + 0.618 * Tmax);
      if (int rc = fit_point(T2, &ic2, &icT2)) return rc;
    }
  }
  Candidate best;
  best.T0 = 0;
  best.lambda = 0.0;
  best.coef0 = 0.0;
  best.loss = 0.0;
  best.ic = DBL_MAX;
  best.iters = 0;
  for (int T = Tmin; T <= Tmax; T++) {
    if (int rc = run_fit(s, T, 0.0, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    int iters_full = s->l;
    double v;
    if (int rc = metric_ic(s, ic_type, is_cv, &v)) return rc;
    if (v < best.ic) {
      // read AFTER ic(): under CV these are the last fold's fit, src/path.cpp:314-319
      best.T0 = T;
      best.beta = s->beta;
      best.coef0 = s->coef0;
      if (int rc = metric_train_loss(s, &best.loss)) return rc;
      best.ic = v;
      best.iters = iters_full;
      store_candidate(s, res, best, true);
    }
  }
  store_best(s, res, best, true);
  return 0;
}

// --------------------------------------------------------------------------------------------
// Powell path for the L0L2 / bsrr types: pgs_path with golden-section or sequential line searches over
// (s, log lambda), src/path.cpp:391-1309.  Pure host control over algorithm_fit(); the ic_sequence matrix the
// reference fills on the side (returned only by the R build as ic_mat) is not kept.
// --------------------------------------------------------------------------------------------
namespace powell {

static int sgn(double a) { return a > 0 ? 1 : (a < 0 ? -1 : 0); }
static double det2(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }

static bool line_intersection(double l1[2][2], double l2[2][2], double out[2]) {  // :414-440
  double xd[2] = {l1[0][0] - l1[1][0], l2[0][0] - l2[1][0]}, yd[2] = {l1[0][1] - l1[1][1], l2[0][1] - l2[1][1]};
  double div = det2(xd, yd);
  if (div == 0) return false;
  double d[2] = {det2(l1[0], l1[1]), det2(l2[0], l2[1])};
  out[0] = det2(d, xd) / div;
  out[1] = det2(d, yd) / div;
  return true;
}

static void cal_intersections(const double p[2], const double u[2], int s_min, int s_max, double lmin, double lmax,
                              double a[2], double b[2]) {  // :445-577
  double l0[2][2] = {{p[0], p[1]}, {p[0] + u[0], p[1] + u[1]}};
  double ls[4][2][2] = {{{(double)s_min, lmin}, {(double)s_min, lmax}},
                        {{(double)s_max, lmin}, {(double)s_max, lmax}},
                        {{(double)s_min, lmin}, {(double)s_max, lmin}},
                        {{(double)s_min, lmax}, {(double)s_max, lmax}}};
  double is[4][2];
  bool ok[4];
  for (int i = 0; i < 4; i++) ok[i] = line_intersection(l0, ls[i], is[i]);
  for (int i = 0; i < 4; i++)
    if (ok[i] && (is[i][0] < s_min - 0.0001 || is[i][0] > s_max + 0.0001 || is[i][1] < lmin - 0.001 ||
                  is[i][1] > lmax + 0.001))
      ok[i] = false;
  for (int i = 0; i < 4; i++)
    if (ok[i])
      for (int j = i + 1; j < 4; j++)
        if (ok[j] && std::fabs(is[i][0] - is[j][0]) < 0.0001 && std::fabs(is[i][1] - is[j][1]) < 0.0001) ok[j] = false;
  int j = 0;
  for (int i = 0; i < 4; i++)
    if (ok[i]) {
      if (j == 2) j += 1;
      if (j == 1) {
        b[0] = is[i][0];
        b[1] = is[i][1];
        j += 1;
      }
      if (j == 0) {
        a[0] = is[i][0];
        a[1] = is[i][1];
        j += 1;
      }
    }
}

struct Search {
  bessx_session *s;
  int ic_type, is_cv;
  SparseVec beta_init;
  double coef0_init = 0.0;
  int fit(int T0, double lambda) {
    if (int rc = run_fit(s, T0, lambda, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    return 0;
  }
};

struct Point {  // what a line search reports back
  SparseVec beta;
  double coef0 = 0, loss = 0, ic = 0;
};

#define PW_TRY(expr)              \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)

// golden_section_search, :579-935
static int golden_section_search(bessx_session *s, int ic_type, int is_cv, const double p[2], const double u[2],
                                 int s_min, int s_max, double lmin, double lmax, double best_arg[2], Point &out) {
  Search ps{s, ic_type, is_cv, SparseVec(), 0.0};
  SparseVec bt1, bt2;
  double lt1 = 0, lt2 = 0, c01 = 0, c02 = 0, closs, dloss, a[2] = {0, 0}, b[2] = {0, 0}, c[2], d[2], h[2];
  const double s_tol = 2, ltol = (lmax - lmin) / 200;
  const double invphi = (std::pow(5, 0.5) - 1.0) / 2.0, invphi2 = (3.0 - std::pow(5, 0.5)) / 2.0;
  cal_intersections(p, u, s_min, s_max, lmin, lmax, a, b);
  h[0] = b[0] - a[0];
  h[1] = b[1] - a[1];
  c[0] = a[0] + invphi2 * h[0];
  c[1] = a[1] + invphi2 * h[1];
  d[0] = a[0] + invphi * h[0];
  d[1] = a[1] + invphi * h[1];
  if (h[0] > 0.0001) {
    c[0] = (int)c[0];
    d[0] = std::ceil(d[0]);
  } else if (h[0] < -0.0001) {
    c[0] = std::ceil(c[0]);
    d[0] = (int)d[0];
  } else {
    c[0] = std::round(c[0]);
    d[0] = std::round(d[0]);
  }
  PW_TRY(ps.fit((int)c[0], std::exp(c[1])));
  PW_TRY(metric_ic(s, ic_type, is_cv, &closs));
  c01 = s->coef0;
  bt1 = s->beta;
  PW_TRY(metric_train_loss(s, &lt1));
  PW_TRY(ps.fit((int)d[0], std::exp(d[1])));
  PW_TRY(metric_ic(s, ic_type, is_cv, &dloss));
  c02 = s->coef0;
  bt2 = s->beta;
  PW_TRY(metric_train_loss(s, &lt2));
  int tt = 0;
  for (;;) {
    if ((std::fabs((invphi2 - invphi) * h[0]) <= s_tol && std::fabs((invphi2 - invphi) * h[1]) < ltol) || tt == 50) {
      double min_loss, tmp;
      if (closs < dloss) {
        best_arg[0] = c[0];
        best_arg[1] = c[1];
        min_loss = closs;
        out.beta = bt1;
        out.coef0 = c01;
        out.ic = closs;
        out.loss = lt1;
      } else {
        best_arg[0] = d[0];
        best_arg[1] = d[1];
        min_loss = dloss;
        out.beta = bt2;
        out.coef0 = c02;
        out.ic = dloss;
        out.loss = lt2;
      }
      for (int i = 1; i < std::fabs((invphi2 - invphi) * h[0]); i++) {
        PW_TRY(ps.fit((int)(c[0] + sgn(h[0]) * i), std::exp(c[1])));
        PW_TRY(metric_ic(s, ic_type, is_cv, &tmp));
        if (tmp < min_loss) {
          best_arg[0] = c[0] + sgn(h[0]) * i;
          best_arg[1] = c[1];
          min_loss = tmp;
          out.beta = s->beta;
          out.coef0 = s->coef0;
          PW_TRY(metric_train_loss(s, &out.loss));
          out.ic = min_loss;
        }
      }
      return 0;
    }
    if (tt >= 100) return 0;
    tt++;
    if (closs < dloss) {
      // the stored model of d (bt2, c02, lt2) is deliberately NOT moved along with the point (:762-766)
      b[0] = d[0];
      b[1] = d[1];
      d[0] = c[0];
      d[1] = c[1];
      dloss = closs;
      h[0] = b[0] - a[0];
      h[1] = b[1] - a[1];
      c[0] = a[0] + invphi2 * h[0];
      c[1] = a[1] + invphi2 * h[1];
      c[0] = h[0] > 0.0001 ? (double)(int)c[0] : (h[0] < -0.0001 ? std::ceil(c[0]) : std::round(c[0]));
      PW_TRY(ps.fit((int)c[0], std::exp(c[1])));
      PW_TRY(metric_ic(s, ic_type, is_cv, &closs));
      c01 = s->coef0;
      bt1 = s->beta;
      PW_TRY(metric_train_loss(s, &lt1));
    } else {
      a[0] = c[0];
      a[1] = c[1];
      c[0] = d[0];
      c[1] = d[1];
      closs = dloss;
      h[0] = b[0] - a[0];
      h[1] = b[1] - a[1];
      d[0] = a[0] + invphi * h[0];
      d[1] = a[1] + invphi * h[1];
      d[0] = h[0] > 0.0001 ? std::ceil(d[0]) : (h[0] < -0.0001 ? (double)(int)d[0] : std::round(d[0]));
      PW_TRY(ps.fit((int)d[0], std::exp(d[1])));
      PW_TRY(metric_ic(s, ic_type, is_cv, &dloss));
      c02 = s->coef0;
      bt2 = s->beta;
      PW_TRY(metric_train_loss(s, &lt2));
    }
  }
}

static int gdc_int(int a, int b) {  // GDC, :937-953
  int Max = a > b ? a : b, Min = (a == Max) ? b : a, z = Min;
  while (Max % Min != 0) {
    z = Max % Min;
    Max = Min;
    Min = z;
  }
  return z;
}

// seq_search, :954-1137 (u is normalised in place like the reference does)
static int seq_search(bessx_session *s, int ic_type, int is_cv, double p[2], double u[2], int s_min, int s_max,
                      double lmin, double lmax, double best_arg[2], Point &out, int nlambda) {
  Search ps{s, ic_type, is_cv, SparseVec(), 0.0};
  const double d_lambda = (lmax - lmin) / (nlambda - 1);
  const size_t cap = (size_t)(s_max - s_min + 1) * nlambda + 2;
  int k_lambda = (int)std::fabs(std::round(u[1] / d_lambda));
  if (std::fabs(u[0]) != 1 && k_lambda != 1) {
    if (k_lambda == 0 && u[0] != 0) {
      u[0] = u[0] / std::fabs(u[0]);
    } else if (u[0] == 0 && k_lambda != 0) {
      u[1] = u[1] / k_lambda;
    } else if (!(k_lambda == 0 && (int)u[0] == 0)) {  // the reference divides by zero there
      int g = gdc_int(k_lambda, std::abs((int)u[0]));
      if (g) {
        u[0] = std::round(u[0] / g);
        u[1] = u[1] / g;
      }
    }
  }
  std::vector<Point> f1, f2;
  auto eval = [&](int T0, double lambda, std::vector<Point> &dst) -> int {
    PW_TRY(ps.fit(T0, lambda));
    Point q;
    PW_TRY(metric_ic(s, ic_type, is_cv, &q.ic));
    q.beta = s->beta;
    q.coef0 = s->coef0;
    PW_TRY(metric_train_loss(s, &q.loss));
    dst.push_back(q);
    return 0;
  };
  PW_TRY(eval((int)(p[0]), std::exp(p[1]), f1));
  f2.push_back(f1[0]);
  SparseVec beta_warm = ps.beta_init;
  double coef0_warm = ps.coef0_init;
  for (int i = 1; (p[0] + i * u[0] <= s_max) && (p[1] + i * u[1] <= lmax + d_lambda * 1e-4) &&
                  (p[0] + i * u[0] >= s_min) && (p[1] + i * u[1] >= lmin - d_lambda * 1e-4) && f1.size() < cap;
       i++)
    PW_TRY(eval((int)(p[0] + i * u[0]), std::exp(p[1] + i * u[1]), f1));
  ps.beta_init = beta_warm;
  ps.coef0_init = coef0_warm;
  for (int j = 1; (p[0] - j * u[0] <= s_max) && (p[1] - j * u[1] <= lmax + d_lambda * 1e-4) &&
                  (p[0] - j * u[0] >= s_min) && (p[1] - j * u[1] >= lmin - d_lambda * 1e-4) && f2.size() < cap;
       j++)
    PW_TRY(eval((int)(p[0] - j * u[0]), std::exp(p[1] - j * u[1]), f2));
  size_t m1 = 0, m2 = 0;
  for (size_t q = 1; q < f1.size(); q++)
    if (f1[q].ic < f1[m1].ic) m1 = q;
  for (size_t q = 1; q < f2.size(); q++)
    if (f2[q].ic < f2[m2].ic) m2 = q;
  int minpos;
  if (f1[m1].ic < f2[m2].ic) {
    minpos = (int)m1;
    out = f1[m1];
  } else {
    minpos = -(int)m2;
    out = f2[m2];
  }
  best_arg[0] = p[0] + minpos * u[0];
  best_arg[1] = p[1] + minpos * u[1];
  return 0;
}

}  // namespace powell

// pgs_path, :1138-1309
static int pgs_path(bessx_session *s, int s_min, int s_max, double lmin, double lmax, int powell_path, int nlambda,
                    int ic_type, int is_cv, bessx_path_result *res) {
  using namespace powell;
  if (powell_path == 1) nlambda = 100;
  double P[3][2] = {{(double)s_min, lmin}, {0, 0}, {0, 0}};
  double U[2][2] = {{0., (lmax - lmin) / (nlambda - 1)}, {1., 0.}};
  std::vector<Candidate> all;
  Point pt;
  auto search = [&](double *pin, double *uu, double *pout) -> int {
    return powell_path == 1 ? golden_section_search(s, ic_type, is_cv, pin, uu, s_min, s_max, lmin, lmax, pout, pt)
                            : seq_search(s, ic_type, is_cv, pin, uu, s_min, s_max, lmin, lmax, pout, pt, nlambda);
  };
  auto record = [&](size_t idx, double lam) {
    if (all.size() <= idx) all.resize(idx + 1);
    Candidate &c = all[idx];
    c.T0 = (int)pt.beta.idx.size();
    c.lambda = lam;
    c.beta = pt.beta;
    c.coef0 = pt.coef0;
    c.loss = pt.loss;
    c.ic = pt.ic;
    c.iters = 0;
  };
  int ttt = 0;
  PW_TRY(search(P[0], U[1], P[0]));
  record(ttt, std::exp(P[0][1]));
  while (ttt < 11) {
    ttt++;
    for (int i = 0; i < 2; i++) {
      PW_TRY(search(P[i], U[i], P[i + 1]));
      record(ttt, std::exp(P[i + 1][1]));
      ttt++;
    }
    U[0][0] = U[1][0];
    U[0][1] = U[1][1];
    U[1][0] = P[2][0] - P[0][0];
    U[1][1] = P[2][1] - P[0][1];
    if (!(std::fabs(U[1][0]) <= 0.0001 && std::fabs(U[1][1]) <= 0.0001) && ttt < 11) {
      PW_TRY(search(P[0], U[1], P[0]));
      record(ttt, std::exp(P[0][1]));
    } else {
      // final fit at P[0]; beta_init / coef0_init are whatever the last search left in the algorithm (:1221-1225)
      s->cur_rows = 0;
      s->sparsity_level = (int)P[0][0];
      s->lambda_level = std::exp(P[0][1]);
      PW_TRY(algorithm_fit(s));
      pt.beta = s->beta;
      pt.coef0 = s->coef0;
      PW_TRY(metric_train_loss(s, &pt.loss));
      PW_TRY(metric_ic(s, ic_type, is_cv, &pt.ic));
      record(ttt, std::exp(P[0][1]));
      all[ttt].iters = s->l;
      ttt++;
      size_t mi = 0;
      for (size_t k = 1; k < (size_t)ttt; k++)
        if (all[k].ic < all[mi].ic) mi = k;
      if (all[mi].ic == all[ttt - 1].ic) mi = ttt - 1;
      for (int k = 0; k < ttt; k++) store_candidate(s, res, all[k], false);
      store_best(s, res, all[mi], false);
      return 0;
    }
  }
  return fail(BESSX_ERR_NUMERIC, "powell end wrong (src/path.cpp:1298-1308)");
}
#undef PW_TRY

struct PgsArgs {
  double lmin, lmax;
  int powell_path, nlambda;
};

// the part of reset_path_caches() a path needs that continues on the caches of the previous call
static int settle_device_chain(bessx_session *s) {
  if (s->ahead.armed) {
    s->ahead.armed = false;
    HIPX(hipStreamSynchronize(s->st));
  }
  s->pend_on = false;
  s->hint.on = false;
  return 0;
}

static int run_path(bessx_session *s, bool gs, const int *seq, int ns, const double *lam, int nl, int s_min,
                    int s_max, int ic_type, int is_cv, bessx_path_result *res, const PgsArgs *pgs = nullptr,
                    bessx_path_chain *chain = nullptr) {
  if (!s || !res) return fail(BESSX_ERR_ARG, "null session or result");
  if (is_cv && s->K < 2) return fail(BESSX_ERR_ARG, "is_cv needs bessx_session_set_cv first");
  HIPX(hipSetDevice(s->device));
  s->trace.clear();
  s->metric_depth = 0;
  for (auto &v : s->cv_init) v.clear();
  res->n_candidates = 0;
  s->n_fits = 0;
  s->n_iters = 0;
  auto t0 = std::chrono::steady_clock::now();
  // a path call starts cold, like a bessCpp call -- unless it continues the job of the previous call (chain->keep_caches:
  // the Gram columns and score sums in memory depend on the data only and stay valid)
  if (int rc0 = (chain && chain->keep_caches) ? settle_device_chain(s) : reset_path_caches(s)) return rc0;
  int rc = pgs  ? pgs_path(s, s_min, s_max, pgs->lmin, pgs->lmax, pgs->powell_path, pgs->nlambda, ic_type, is_cv, res)
           : gs ? gs_path(s, s_min, s_max, ic_type, is_cv, res)
                : sequential_path(s, seq, ns, lam, nl, ic_type, is_cv, res, chain);
  auto t1 = std::chrono::steady_clock::now();
  res->device_seconds = std::chrono::duration<double>(t1 - t0).count();
  res->n_fits = s->n_fits;
  res->n_pdas_iters = s->n_iters;
  return rc;
}

}  // namespace bessx

namespace {
struct Scratch {
  std::vector<void *> ptrs;
  ~Scratch() {
    for (void *q : ptrs) (void)hipFree(q);
  }
  template <class T>
  hipError_t alloc(T **out, size_t count) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(out), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*out);
    return e;
  }
};

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
}  // namespace

// ==============================================================================================
// extern "C"
// ==============================================================================================
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
    int lo = 0, hi = 0;
    HIPT(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPT(hipStreamCreateWithPriority(&s->st, hipStreamDefault, hi));
    HIPT(gram_lds_prepare());
    if (const char *ev = std::getenv("BESSX_IRLS_FUSE")) s->irls_fuse = std::string(ev) == "1";
    s->irls_wfloor = g_marginal_fit_variant == 1 ? 0 : 1;
    if (const char *ev = std::getenv("BESSX_GRAM")) gram_set_variant(std::string(ev) == "direct" ? 0 : 1);
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
      if (std::getenv("BESSX_GROUP_EXPAND") && std::string(std::getenv("BESSX_GROUP_EXPAND")) == "host") s->g_uniform = 0;
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
    if (const char *ev = std::getenv("BESSX_PUBLISH")) s->publish = std::atoi(ev) != 0;
    if (const char *ev = std::getenv("BESSX_WAIT_TIMEOUT_S")) s->wait_deadline_s = std::max(0.001, std::atof(ev));
    if (const char *ev = std::getenv("BESSX_CHAIN")) s->chain = std::atoi(ev) != 0;
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
  // (k_cox_score1p); BESSX_COX_SCORE=2pass keeps the totals / carry / rescan form (two reads of X)
  if (s->model_type == 4) {
    const char *ev = std::getenv("BESSX_COX_SCORE");
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
    const bool eligible = s->model_type == 1 && !s->grouped;
    if (mode == 2 && !eligible)
      return bail(fail(BESSX_ERR_ARG, "covariance score mode exists for LM with singleton groups only"));
    if (eligible && mode != 1) {
      // capacity: every column if p is small, else a few active sets' worth, within 1 GiB per row set
      if (const char *ev = std::getenv("BESSX_PANEL_VARIANT")) s->cov_variant = std::atoi(ev) == 4 ? 4 : 3;
      if (const char *ev = std::getenv("BESSX_PANEL_PAIR_AUTO")) s->cov_pair_auto = std::string(ev) != "0";
      // the pair kernel (variant 4) forms two 32-column groups per pass over X: fills then speculate up to 64 columns
      s->cov_spec = (s->cov_variant == 4 && p >= 4 * COV_R && topk_supported(p, 2 * COV_R)) ? 2 * COV_R : COV_R;
      // capacity: EVERY column when that fits 2 GiB per row set (p <= ~16000: a path then forms a column at most once and
      // the cache is never started over -- at 2560 columns the reference's default sequence 1..min(p, n / log n) at
      // n = 25000, p = 3000 restarted it 1237 times and streamed X 73 000 times, round 4); else 2560 columns within 2 GiB
      const long all = ((long)p + 31) / 32 * 32 + COV_R + s->cov_spec;
      const long budget = (((long)2 << 30) / ((long)p * 8)) / 32 * 32;
      long C = all <= budget ? all : std::min<long>(2560, budget);
      if (const char *ev = std::getenv("BESSX_COV_CAP"))  // test hook: a small cache exercises the restart path
        C = std::min<long>(C, std::max(0, std::atoi(ev)) / 32 * 32);
      if (C >= 2 * COV_R + s->cov_spec) {
        s->cov_mode = true;
        s->cov_C = (int)C;
        const int pt = (p + 15) / 16, njg = (pt + cov_streamed_tiles_per_wave() - 1) / cov_streamed_tiles_per_wave();
        // row slabs: two 256-thread blocks of the panel kernel share a CU (50 KB of LDS each), so pick the slab count
        // whose block count wastes the least of the last round of 512 blocks
        long ns = 1, rps = ld;
        {
          const long conc = 512;  // blocks resident at a time
          double best = 1e300;
          const long ns_max = std::max<long>(1, std::min<long>(64, ld / 256));
          for (long t = 1; t <= ns_max; t++) {
            const long r = ((ld + t - 1) / t + 63) / 64 * 64, used = (ld + r - 1) / r;
            const long blocks = (long)njg * used;
            const double cost = (double)((blocks + conc - 1) / conc) * (double)r * (blocks < conc ? 2.0 : 1.0);
            if (cost < best) {
              best = cost;
              ns = used;
              rps = r;
            }
          }
        }
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
        if (const char *ev = std::getenv("BESSX_COV_SOLVER")) s->cov_cg = std::string(ev) != "chol";
        if (const char *ev = std::getenv("BESSX_FUSE")) s->fuse = std::string(ev) != "0";
        if (const char *ev = std::getenv("BESSX_FUSE_SEL")) s->fuse_sel = std::string(ev) != "0";
        if (const char *ev = std::getenv("BESSX_DEFER_PUBLISH")) s->defer_pub = std::string(ev) != "0";
        if (!s->fuse) s->defer_pub = false;
        if (const char *ev = std::getenv("BESSX_CG_LAYOUT")) s->cg_by_rows = std::string(ev) != "tiles";
        if (const char *ev = std::getenv("BESSX_COV_CS")) s->cov_cs = std::min(COV_CS, std::max(1, std::atoi(ev)));
        if (const char *ev = std::getenv("BESSX_CG_TOL")) {
          const double v = std::atof(ev);
          if (v >= 1e-15 && v <= 1e-6) s->cg_tol = v;
        }
      } else if (mode == 2) {
        return bail(fail(BESSX_ERR_ARG, "covariance score mode: p too large for the Gram column cache"));
      }
    }
  }
  HIPT(dmalloc(&s->idcols, (size_t)capA + 16));
  if (s->model_type == 4) {
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
    for (auto v : vecs) HIPT(V(v, (size_t)ld));
    double **vecs1[] = {&c.C1, &c.CU, &c.CV, &c.C2};
    for (auto v : vecs1) HIPT(V(v, (size_t)ld));
    HIPT(V(&c.ldl_work, CHOL_FB_DOUBLES));
    // k-sized work space: for sparsity levels up to 254 now, grown by cox_reserve() when a larger one is asked for
    s->cox_M_cols = 256;
    HIPT(V(&c.M, (size_t)ld * s->cox_M_cols));
    HIPT(V(&c.g, (size_t)capA));
    HIPT(V(&c.u, (size_t)capA));
    HIPT(V(&c.b0, (size_t)capA));
    HIPT(V(&c.Gt2, (size_t)136 * 256));
    HIPT(V(&c.llpart, (size_t)(n + 255) / 256 + 1));
    HIPT(V(&c.SCR, cox_scan_scratch_doubles(ld, 256)));
    // one-pass Hessian of the Newton step (k_cox_hess, up to 10 tile rows; BESSX_COX_HESS=2pass: M = S1 / S0
    // materialised and two Gram launches, as in round 2)
    // Small samples keep the two-pass form: it is built like the reference's own formulas (M = S1 / S0, two Grams), so
    // on the ill-conditioned fits small n produces (near-separated risk sets, a ridge that outweighs the information
    // matrix) its rounding follows the reference's more closely -- both forms are accurate to rounding there, but a
    // Newton iteration on such a system amplifies rounding to 1e-4 and beyond (tests/test_cox_gpu.py).
    // BESSX_COX_HESS=1pass forces the one-pass form at any size.
    c.fit_clamp = g_marginal_fit_variant == 2 ? 50.0 : 30.0;
    c.hess_fused = n >= 1024 ? 1 : 0;
    if (const char *ev = std::getenv("BESSX_COX_HESS")) c.hess_fused = std::string(ev) == "2pass" ? 0 : (std::string(ev) == "1pass" ? 1 : c.hess_fused);
    if (c.hess_fused) {
      const size_t hrows = (size_t)cox_hess_slab_rows(ld), hns = ((size_t)ld + hrows - 1) / hrows;
      if (hns * 55 * 256 > s->gpart_elems) {
        c.hess_fused = 0;  // (cannot happen with the default workspace: 256 slabs x 55 tiles)
      } else {
        HIPT(cox_hess_prepare());
        HIPT(V(&c.CW, (size_t)ld));
        HIPT(V(&c.HP2, hns * 55 * 256));
        HIPT(V(&c.HT, hns * 160));
        HIPT(V(&c.CAR, hns * 160));
        HIPT(V(&c.HQ, hns * 160));
      }
    }
  }
  if (s->model_type == 1) TRY(prepare_rowset(s, 0));
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
    int lo = 0, hi = 0;
    e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    // (the chains share the parent's priority level: spread over the levels, which have their own pools of hardware
    // queues, the chains on the lower levels ran 2-8 x slower per kernel and the path no faster)
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, hi);
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
  if (const char *ev = std::getenv("BESSX_CV_SHARED")) share = share && std::string(ev) != "0";
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
    if (rc == 0 && s->cov_mode) rc = alloc_cov_cache(s, share);
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
    if (const char *ev = std::getenv("BESSX_CV_SIDE_BY_SIDE")) sbs = sbs && std::string(ev) != "0";
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
        s->fold_pool->start(K - 1, s->device);
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

int bessx_session_sequential_path(bessx_session *s, const int *sequence, int sequence_len, const double *lambda_seq,
                                  int lambda_len, int ic_type, int is_cv, bessx_path_result *res) {
  if (!sequence || sequence_len < 1 || !lambda_seq || lambda_len < 1)
    return fail(BESSX_ERR_ARG, "sequential_path: empty sequence");
  return run_path(s, false, sequence, sequence_len, lambda_seq, lambda_len, 0, 0, ic_type, is_cv, res);
}

int bessx_session_sequential_path_chain(bessx_session *s, const int *sequence, int sequence_len,
                                        const double *lambda_seq, int lambda_len, int ic_type, int is_cv,
                                        bessx_path_chain *chain, bessx_path_result *res) {
  if (!sequence || sequence_len < 1 || !lambda_seq || lambda_len < 1)
    return fail(BESSX_ERR_ARG, "sequential_path: empty sequence");
  if (!chain) return fail(BESSX_ERR_ARG, "sequential_path_chain: null chain");
  if (chain->init_len < 0 || (chain->init_len > 0 && (!chain->init_idx || !chain->init_val)))
    return fail(BESSX_ERR_ARG, "sequential_path_chain: bad initial model");
  if (is_cv && chain->init_len > 0)
    return fail(BESSX_ERR_UNSUPPORTED, "sequential_path_chain: under CV a chain would need the folds' models as well");
  if (chain->stop_support && (chain->stop_rows < 0 || chain->stop_row_len < 1))
    return fail(BESSX_ERR_ARG, "sequential_path_chain: bad stop table");
  return run_path(s, false, sequence, sequence_len, lambda_seq, lambda_len, 0, 0, ic_type, is_cv, res, nullptr, chain);
}

int bessx_session_gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res) {
  if (s_min < 1 || s_max < s_min) return fail(BESSX_ERR_ARG, "gs_path: need 1 <= s_min <= s_max");
  return run_path(s, true, nullptr, 0, nullptr, 0, s_min, s_max, ic_type, is_cv, res);
}

int bessx_session_pgs_path(bessx_session *s, int s_min, int s_max, double lambda_min, double lambda_max, int n_lambda,
                           int powell_path, int ic_type, int is_cv, bessx_path_result *res) {
  if (s_min < 1 || s_max < s_min) return fail(BESSX_ERR_ARG, "pgs_path: need 1 <= s_min <= s_max");
  if (powell_path != 1 && powell_path != 2) return fail(BESSX_ERR_ARG, "pgs_path: powell_path must be 1 or 2");
  if (powell_path == 2 && n_lambda < 2) return fail(BESSX_ERR_ARG, "pgs_path: n_lambda must be >= 2");
  // bessCpp, src/bess.cpp:176-177
  PgsArgs a{std::log(std::max(lambda_min, 1e-5)), std::log(std::max(lambda_max, 1e-5)), powell_path, n_lambda};
  return run_path(s, true, nullptr, 0, nullptr, 0, s_min, s_max, ic_type, is_cv, res, &a);
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

int bessx_session_fit(bessx_session *s, int T0, double lambda, int fold, const int *init_idx, const double *init_val,
                      int init_len, double init_coef0, int *support, double *beta, double *coef0, int *iters,
                      double *train_loss, double *test_loss) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (fold >= s->K) return fail(BESSX_ERR_ARG, "fold index out of range");
  const int width = bessx_session_fit_width(s, T0);
  if (width < 0) return fail(BESSX_ERR_ARG, "sparsity level outside [1, number of groups]");
  HIPX(hipSetDevice(s->device));
  s->cur_rows = fold < 0 ? 0 : fold + 1;
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init.clear();
  for (int i = 0; i < init_len; i++) {
    if (init_idx[i] < 0 || init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "init index out of range");
    s->beta_init.idx.push_back(init_idx[i]);
    s->beta_init.val.push_back(init_val[i]);
  }
  s->coef0_init = init_coef0;
  if (int rc = algorithm_fit(s)) return rc;
  // (groups of size > 1: T0 counts groups, the fit returns the columns of the selected groups -- at most `width`)
  const int got = (int)s->beta.idx.size();
  if (got > width) return fail(BESSX_ERR_NUMERIC, "internal error: more columns selected than bessx_session_fit_width allows");
  for (int i = 0; i < width; i++) {
    if (support) support[i] = i < got ? s->beta.idx[i] : -1;
    if (beta) beta[i] = i < got ? s->beta.val[i] : 0.0;
  }
  if (coef0) *coef0 = s->coef0;
  if (iters) *iters = s->l;
  if (train_loss) *train_loss = metric_train_loss_value(s);
  if (test_loss) *test_loss = fold < 0 ? 0.0 : metric_fold_test_loss(s, fold);
  return BESSX_OK;
}

// One evaluation of a cross-validated candidate for a SUBSET of the folds (and, optionally, the full-data fit in front
// of them): what a rank of a fold-sharded path owns (bess_amd/dist.py).  The fold fits take the library's own route --
// the chains side by side with union fills where that applies, one after another on the session's state otherwise --
// and the session's own per-fold warm starts (Metric::cv_initial_model_param).
int bessx_session_cv_eval(bessx_session *s, int T0, double lambda, int want_full, const int *init_idx,
                          const double *init_val, int init_len, double init_coef0, const int *folds, int n_folds,
                          int *support, double *beta, double *coef0, int *iters, double *train_loss, double *test_loss) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (n_folds < 0 || (n_folds > 0 && !folds)) return fail(BESSX_ERR_ARG, "cv_eval: bad fold list");
  if (n_folds > 0 && s->K < 2) return fail(BESSX_ERR_ARG, "cv_eval needs bessx_session_set_cv first");
  for (int i = 0; i < n_folds; i++)
    if (folds[i] < 0 || folds[i] >= s->K || (i > 0 && folds[i] <= folds[i - 1]))
      return fail(BESSX_ERR_ARG, "cv_eval: folds must be ascending indices in [0, K)");
  const int width = bessx_session_fit_width(s, T0);
  if (width < 0) return fail(BESSX_ERR_ARG, "sparsity level outside [1, number of groups]");
  HIPX(hipSetDevice(s->device));
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init.clear();
  for (int i = 0; i < init_len; i++) {
    if (init_idx[i] < 0 || init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "init index out of range");
    s->beta_init.idx.push_back(init_idx[i]);
    s->beta_init.val.push_back(init_val[i]);
  }
  s->coef0_init = init_coef0;
  int rec = 0;
  auto put = [&](const SparseVec &b, double c0, int l, double tr, double te) -> int {
    const int got = (int)b.idx.size();
    if (got > width) return fail(BESSX_ERR_NUMERIC, "internal error: more columns selected than bessx_session_fit_width allows");
    for (int i = 0; i < width; i++) {
      if (support) support[(size_t)rec * width + i] = i < got ? b.idx[i] : -1;
      if (beta) beta[(size_t)rec * width + i] = i < got ? b.val[i] : 0.0;
    }
    if (coef0) coef0[rec] = c0;
    if (iters) iters[rec] = l;
    if (train_loss) train_loss[rec] = tr;
    if (test_loss) test_loss[rec] = te;
    rec++;
    return 0;
  };
  if (want_full) {
    s->cur_rows = 0;
    if (int rc = algorithm_fit(s)) return rc;
    if (int rc = put(s->beta, s->coef0, s->l, metric_train_loss_value(s), 0.0)) return rc;
  }
  if (n_folds == 0) return BESSX_OK;
  const std::vector<int> only(folds, folds + n_folds);
  s->metric_depth++;
  int rc = 0;
  if (side_by_side_applies(s, T0)) {
    std::vector<double> tl((size_t)n_folds);
    double mean = 0.0;
    rc = fold_fits_side_by_side(s, &mean, &only, tl.data());
    for (int i = 0; i < n_folds && rc == 0; i++) {
      const bessx_session *c = s->fold_ctx[(size_t)only[i]];
      rc = put(c->beta, c->coef0, c->l, metric_train_loss_value(c), tl[i]);
    }
  } else {
    const SparseVec keep = s->beta_init;
    for (int i = 0; i < n_folds && rc == 0; i++) {
      const int k = only[i];
      s->beta_init = s->warm_start ? s->cv_init[k] : keep;  // update_beta_init(cv_initial_model_param.row(k))
      s->cur_rows = k + 1;                                  // update_train_mask + update_group_XTX
      rc = algorithm_fit(s);
      if (rc) break;
      if (s->warm_start) s->cv_init[k] = s->beta;
      rc = put(s->beta, s->coef0, s->l, metric_train_loss_value(s), metric_fold_test_loss(s, k));
    }
  }
  s->metric_depth--;
  return rc;
}

// ----------------------------------------------------------------------------------------------
// Cooperative prefill of the Gram column cache (LM, covariance form, all rows): the ranks of a k-path run share the
// passes over X that every chunk's cold start would otherwise repeat (bess_amd/dist.py, cooperative_prefill).  All ranks
// list the same columns -- slots are handed out in list order on a cache started over, so slot numbers agree across
// ranks -- each forms its share of the 32-column groups, the p x 32 blocks travel (RCCL all-gather), every rank
// imports the others' and fills the slot-indexed Gram once.  Cache contents only: no result depends on it.
// ----------------------------------------------------------------------------------------------
static int prefill_ready(bessx_session *s) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (!s->cov_mode || s->model_type != 1 || s->grouped)
    return fail(BESSX_ERR_UNSUPPORTED, "cov_prefill: the session does not run the covariance form of the LM score pass");
  if (s->cv_shared) return fail(BESSX_ERR_UNSUPPORTED, "cov_prefill: not offered on sessions with cross-validation folds");
  HIPX(hipSetDevice(s->device));
  if (!s->fill_ctrl) {
    HIPX(hipMalloc(reinterpret_cast<void **>(&s->fill_ctrl), sizeof(FitCtrl)));
    HIPX(hipMemset(s->fill_ctrl, 0, sizeof(FitCtrl)));
    HIPX(hipHostMalloc(reinterpret_cast<void **>(&s->fill_ctrl_h), sizeof(FitCtrl)));
  }
  return 0;
}

int bessx_session_marginal_scores(bessx_session *s, double *bd) {
  if (!s || !bd) return fail(BESSX_ERR_ARG, "null argument");
  if (s->model_type != 1 || s->grouped) return fail(BESSX_ERR_UNSUPPORTED, "marginal_scores: LM with singleton groups");
  HIPX(hipSetDevice(s->device));
  // get_A at beta = 0, lambda = 0 (src/Algorithm.h:1109-1123): d_j = x_j . y / n, bd_j = (d_j / phi_j)^2, phi_j^2 = x_j . x_j / n
  std::vector<double> xty((size_t)s->p), xtx((size_t)s->p);
  HIPX(hipStreamSynchronize(s->st));
  HIPX(hipMemcpy(xty.data(), s->xty[0], xty.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPX(hipMemcpy(xtx.data(), s->xtx[0], xtx.size() * sizeof(double), hipMemcpyDeviceToHost));
  const double n = (double)s->n_train[0];
  for (int j = 0; j < s->p; j++) {
    const double phi = std::sqrt(xtx[j] / n), d = xty[j] / n;
    const double t = d * (1.0 / phi);
    bd[j] = t * t;
  }
  return BESSX_OK;
}

int bessx_session_cov_prefill_begin(bessx_session *s, const int *cols, int ncols) {
  if (int rc = prefill_ready(s)) return rc;
  if (!cols || ncols < 1 || ncols % COV_R != 0) return fail(BESSX_ERR_ARG, "cov_prefill: need a multiple of 32 columns");
  if (ncols + s->cov_spec + COV_R > cov_C_dev(s) || ncols > s->capA)
    return fail(BESSX_ERR_ARG, "cov_prefill: the list does not fit the Gram column cache");
  std::vector<char> seen((size_t)s->p, 0);
  for (int i = 0; i < ncols; i++) {
    if (cols[i] < 0 || cols[i] >= s->p || seen[(size_t)cols[i]]) return fail(BESSX_ERR_ARG, "cov_prefill: bad column list");
    seen[(size_t)cols[i]] = 1;
  }
  if (int rc = reset_path_caches(s)) return rc;
  int *st_idx = reinterpret_cast<int *>(s->stage_h);
  std::copy(cols, cols + ncols, st_idx);
  HIPX(hipMemcpyAsync(s->init_idx_d, st_idx, (size_t)ncols * sizeof(int), hipMemcpyHostToDevice, s->st));
  CovUnion u = {};
  u.nf = 1;
  u.list[0] = s->init_idx_d;
  u.len[0] = ncols;
  HIPX(launch_cov_fill_union(u, 1, nullptr, nullptr, s->cov_spec, 0, s->cov[0].slot_of, s->cov[0].meta, s->p, s->cov_fcols,
                             s->fill_ctrl, s->st));
  HIPX(hipStreamSynchronize(s->st));  // (the staging buffer is free again)
  s->prefill_cols = ncols;
  return BESSX_OK;
}

int bessx_session_cov_prefill_compute(bessx_session *s, int g0, int ngroups) {
  if (int rc = prefill_ready(s)) return rc;
  if (g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  if (ngroups == 0) return BESSX_OK;
  if (int rc = enqueue_cov_fill(s, 0, ngroups, 1, s->fill_ctrl, g0, false)) return rc;
  s->cov_panel_groups += ngroups;
  HIPX(hipStreamSynchronize(s->st));
  return cov_collect(s, s->prefill_cols);
}

int bessx_session_cov_prefill_export(bessx_session *s, int g0, int ngroups, double *dst, int dst_on_device) {
  if (int rc = prefill_ready(s)) return rc;
  if (!dst || g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  const size_t cnt = (size_t)ngroups * COV_R * s->p;
  HIPX(hipMemcpyAsync(dst, s->cov[0].G + (size_t)g0 * COV_R * s->p, cnt * sizeof(double),
                      dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->st));
  HIPX(hipStreamSynchronize(s->st));
  return BESSX_OK;
}

int bessx_session_cov_prefill_import(bessx_session *s, int g0, int ngroups, const double *src, int src_on_device) {
  if (int rc = prefill_ready(s)) return rc;
  if (!src || g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  const size_t cnt = (size_t)ngroups * COV_R * s->p;
  HIPX(hipMemcpyAsync(s->cov[0].G + (size_t)g0 * COV_R * s->p, src, cnt * sizeof(double),
                      src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st));
  HIPX(hipStreamSynchronize(s->st));  // (the caller's buffer is free again when the call returns)
  return BESSX_OK;
}

int bessx_session_cov_prefill_end(bessx_session *s) {
  if (int rc = prefill_ready(s)) return rc;
  if (s->prefill_cols < COV_R) return fail(BESSX_ERR_ARG, "cov_prefill_end without cov_prefill_begin");
  bessx_session::CovCache &cv = s->cov[0];
  HIPX(launch_cov_compact(cv.G, s->p, cv.slot_of, s->cov_fcols, 0, s->prefill_cols / COV_R, cv.GS, s->cov_cs, s->fill_ctrl, 1,
                          s->st, s->xtx[0], cv.meta));
  HIPX(hipStreamSynchronize(s->st));
  s->prefill_cols = 0;
  return BESSX_OK;
}

static void debug_sleep_cb(void *ms) {
  std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)ms));
}

int bessx_session_debug_block_stream(bessx_session *s, int milliseconds) {
  if (!s || milliseconds < 0) return fail(BESSX_ERR_ARG, "bad argument");
  HIPX(hipSetDevice(s->device));
  HIPX(hipLaunchHostFunc(s->st, debug_sleep_cb, reinterpret_cast<void *>((intptr_t)milliseconds)));
  return BESSX_OK;
}

// ----------------------------------------------------------------------------------------------
// pywrap_bess drop-in (src/bess.cpp:218-281 -> bessCpp :37-214)
// ----------------------------------------------------------------------------------------------
int bessx_pywrap_bess(double *x, int x_row, int x_col, double *y, int y_len, int data_type, double *weight,
                      int weight_len, int is_normal, int algorithm_type, int model_type, int max_iter,
                      int exchange_num, int path_type, int is_warm_start, int ic_type, int is_cv, int K, int *gindex,
                      int gindex_len, double *state, int state_len, int *sequence, int sequence_len,
                      double *lambda_sequence, int lambda_sequence_len, int s_min, int s_max, int K_max,
                      double epsilon, double lambda_min, double lambda_max, int n_lambda, int is_screening,
                      int screening_size, int powell_path, int *always_select, int always_select_len, double tao,
                      double *beta_out, int beta_out_len, double *coef0_out, int coef0_out_len,
                      double *train_loss_out, int train_loss_out_len, double *ic_out, int ic_out_len,
                      double *nullloss_out, double *aic_out, int aic_out_len, double *bic_out, int bic_out_len,
                      double *gic_out, int gic_out_len, int *A_out, int A_out_len, int *l_out) {
  (void)exchange_num; (void)state; (void)state_len; (void)K_max; (void)epsilon;
  (void)tao;  // dead on the live reference paths
  (void)coef0_out_len; (void)train_loss_out_len; (void)ic_out_len;
  if (!x || !y || !beta_out || !coef0_out || !train_loss_out || !ic_out) return fail(BESSX_ERR_ARG, "null argument");
  if (y_len != x_row || (weight && weight_len != x_row)) return fail(BESSX_ERR_ARG, "length of y / weight != rows of x");
  if (beta_out_len < x_col) return fail(BESSX_ERR_ARG, "beta_out too short");
  if (!gindex || gindex_len < 1 || gindex_len > x_col) return fail(BESSX_ERR_ARG, "bad group index");
  bessx_problem pb;
  std::memset(&pb, 0, sizeof(pb));
  pb.n = x_row;
  pb.p = x_col;
  pb.x = x;
  pb.x_col_major = 0;
  pb.y = y;
  pb.weight = weight;
  pb.data_type = data_type;
  pb.is_normal = is_normal;
  pb.model_type = model_type;
  pb.algorithm_type = algorithm_type;
  pb.max_iter = max_iter;
  pb.is_warm_start = is_warm_start;
  pb.always_select = always_select;
  pb.always_select_len = always_select_len;
  pb.device = -1;
  pb.group_index = gindex;
  pb.group_index_len = gindex_len;
  pb.is_screening = is_screening ? 1 : 0;
  pb.screening_size = screening_size;
  {
    // the work space is sized for the largest active set the path can ask for (levels count groups)
    long top = path_type == 1 ? 0 : s_max;
    if (path_type == 1)
      for (int i = 0; i < sequence_len; i++) top = std::max<long>(top, sequence ? sequence[i] : 0);
    long gmax = 1;
    for (int g = 0; g < gindex_len; g++)
      gmax = std::max<long>(gmax, (g + 1 < gindex_len ? gindex[g + 1] : x_col) - gindex[g]);
    pb.max_sparsity = (int)std::min<long>(std::min<long>(top * gmax, x_col), T0_HARD);
  }
  bessx_session *s = nullptr;
  if (int rc = bessx_session_create(&s, &pb)) return rc;
  int rc = 0;
  if (is_cv) rc = bessx_session_set_cv(s, K, nullptr, 123u);
  bessx_path_result res;
  std::memset(&res, 0, sizeof(res));
  res.beta = beta_out;
  if (rc == 0) {
    if (path_type == 1)
      rc = bessx_session_sequential_path(s, sequence, sequence_len, lambda_sequence, lambda_sequence_len, ic_type,
                                         is_cv, &res);
    else if (algorithm_type == 5 || algorithm_type == 3)  // src/bess.cpp:174-180
      rc = bessx_session_pgs_path(s, s_min, s_max, lambda_min, lambda_max, n_lambda, powell_path, ic_type, is_cv, &res);
    else
      rc = bessx_session_gs_path(s, s_min, s_max, ic_type, is_cv, &res);
  }
  if (rc == 0) {
    *coef0_out = res.coef0;
    *train_loss_out = res.train_loss;
    *ic_out = res.ic;
    if (nullloss_out) *nullloss_out = s->nullloss;
    if (aic_out && aic_out_len > 0) aic_out[0] = 0.0;
    if (bic_out && bic_out_len > 0) bic_out[0] = 0.0;
    if (gic_out && gic_out_len > 0) gic_out[0] = 0.0;
    if (A_out) {
      int k = 0;
      for (int j = 0; j < x_col && k < A_out_len; j++)
        if (beta_out[j] != 0.0) A_out[k++] = j;
      for (; k < A_out_len; k++) A_out[k] = -1;
    }
    if (l_out) *l_out = res.best_iters;
  }
  std::string keep = g_err;
  bessx_session_destroy(s);
  g_err = keep;
  return rc;
}

// ----------------------------------------------------------------------------------------------
// bessCpp drop-in for the R package (src/bess.h:20-33; R/src/RcppExports.cpp:10-48): see include/bessx.h 1b
// ----------------------------------------------------------------------------------------------
int bessx_bessCpp(const double *x, int n, int p, const double *y, int data_type, const double *weight, int is_normal,
                  int algorithm_type, int model_type, int max_iter, int exchange_num, int path_type,
                  int is_warm_start, int ic_type, int is_cv, int K, const double *state, int state_len,
                  const int *sequence, int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                  int K_max, double epsilon, double lambda_min, double lambda_max, int nlambda, int is_screening,
                  int screening_size, int powell_path, const int *g_index, int g_index_len, const int *always_select,
                  int always_select_len, double tao, bessx_r_result *res) {
  (void)exchange_num; (void)state; (void)state_len; (void)K_max; (void)epsilon; (void)tao;  // dead in the reference too
  if (!x || !y || !res || !res->beta) return fail(BESSX_ERR_ARG, "bessCpp: null argument");
  if (!g_index || g_index_len < 1 || g_index_len > p) return fail(BESSX_ERR_ARG, "bessCpp: bad group index");
  const bool seqp = path_type == 1;
  const bool powell = !seqp && (algorithm_type == 5 || algorithm_type == 3);  // src/bess.cpp:174-180
  if (seqp && (!sequence || sequence_len < 1 || !lambda_seq || lambda_len < 1))
    return fail(BESSX_ERR_ARG, "bessCpp: empty sequence / lambda_seq");
  bessx_problem pb;
  std::memset(&pb, 0, sizeof(pb));
  pb.n = n;
  pb.p = p;
  pb.x = x;
  pb.x_col_major = 1;
  pb.y = y;
  pb.weight = weight;
  pb.data_type = data_type;
  pb.is_normal = is_normal;
  pb.model_type = model_type;
  pb.algorithm_type = algorithm_type;
  pb.max_iter = max_iter;
  pb.is_warm_start = is_warm_start;
  pb.always_select = always_select;
  pb.always_select_len = always_select_len;
  pb.device = -1;
  pb.group_index = g_index;
  pb.group_index_len = g_index_len;
  pb.is_screening = is_screening ? 1 : 0;
  pb.screening_size = screening_size;
  long gmax = 1, top = seqp ? 0 : s_max;
  for (int g = 0; g < g_index_len; g++) gmax = std::max<long>(gmax, (g + 1 < g_index_len ? g_index[g + 1] : p) - g_index[g]);
  if (seqp)
    for (int i = 0; i < sequence_len; i++) top = std::max<long>(top, sequence[i]);
  pb.max_sparsity = (int)std::min<long>(std::min<long>(top * gmax, p), T0_HARD);
  bessx_session *s = nullptr;
  if (int rc = bessx_session_create(&s, &pb)) return rc;
  auto done = [&](int rc) {
    std::string keep = g_err;
    bessx_session_destroy(s);
    g_err = keep;
    return rc;
  };
  if (is_cv)
    if (int rc = bessx_session_set_cv(s, K, nullptr, 123u)) return done(rc);
  const int cap = seqp ? sequence_len * lambda_len : (powell ? 128 : 2 * (s_max - s_min + 1) + 64);
  const int maxT = (int)std::max<long>(1, std::min<long>(p, std::max<long>(top, 1) * gmax));
  std::vector<double> c_ic((size_t)cap), c_loss((size_t)cap), c_c0((size_t)cap), c_beta((size_t)cap * maxT);
  std::vector<int> c_sup((size_t)cap * maxT, -1), c_T0((size_t)cap);
  std::vector<double> c_lam((size_t)cap);
  bessx_path_result r;
  std::memset(&r, 0, sizeof(r));
  r.beta = res->beta;
  r.capacity = cap;
  r.max_T0 = maxT;
  r.cand_T0 = c_T0.data();
  r.cand_lambda = c_lam.data();
  r.cand_ic = c_ic.data();
  r.cand_train_loss = c_loss.data();
  r.cand_coef0 = c_c0.data();
  r.cand_beta = c_beta.data();
  r.cand_support = c_sup.data();
  int rc = seqp     ? bessx_session_sequential_path(s, sequence, sequence_len, lambda_seq, lambda_len, ic_type, is_cv, &r)
           : powell ? bessx_session_pgs_path(s, s_min, s_max, lambda_min, lambda_max, nlambda, powell_path, ic_type,
                                             is_cv, &r)
                    : bessx_session_gs_path(s, s_min, s_max, ic_type, is_cv, &r);
  if (rc) return done(rc);
  if (is_screening && res->screening_A) {
    // screening_A of src/screening.cpp:68: kept columns, or kept GROUPS when the groups have more than one column
    if (bessx_session_get_screening_groups(s, res->screening_A, screening_size) == 0)
      bessx_session_get_screening(s, res->screening_A, screening_size);
  }
  res->coef0 = r.coef0;
  res->train_loss = r.train_loss;
  res->ic = r.ic;
  res->lambda = r.lambda;
  const int nc = std::min(r.n_candidates, cap);
  res->n_all = nc;
  // candidates arrive in evaluation order; the sequential path's order is the snake of src/path.cpp:50
  std::vector<int> where((size_t)nc);
  if (seqp) {
    int c = 0;
    for (int i = 0; i < sequence_len; i++) {
      const int step = (i % 2 == 0) ? 1 : -1;
      for (int j = (i % 2 == 0) ? 0 : lambda_len - 1; j < lambda_len && j >= 0 && c < nc; j += step)
        where[c++] = j * sequence_len + i;
    }
  } else {
    for (int c = 0; c < nc; c++) where[c] = c;
  }
  const int wr = std::min(nc, res->all_capacity);
  if (res->beta_all) std::fill(res->beta_all, res->beta_all + (size_t)p * std::max(res->all_capacity, 0), 0.0);
  for (int c = 0; c < nc; c++) {
    const int q = where[c];
    if (q >= wr) continue;
    if (res->coef0_all) res->coef0_all[q] = c_c0[c];
    if (res->train_loss_all) res->train_loss_all[q] = c_loss[c];
    if (res->ic_all) res->ic_all[q] = c_ic[c];
    if (res->beta_all)
      for (int t = 0; t < maxT && c_sup[(size_t)c * maxT + t] >= 0; t++)
        res->beta_all[(size_t)q * p + c_sup[(size_t)c * maxT + t]] = c_beta[(size_t)c * maxT + t];
  }
  return done(BESSX_OK);
}

// ----------------------------------------------------------------------------------------------
// single-kernel entry points for parity tests
// ----------------------------------------------------------------------------------------------
int bessx_op_xtv(const double *x, int n, int p, int ld, const double *v, const double *v2, double *out,
                 double *out2) {
  if (int rc = need_device()) return rc;
  if (!x || !v || !out || n < 1 || p < 1 || ld < n) return fail(BESSX_ERR_ARG, "op_xtv: bad arguments");
  Scratch sc;
  const int U = n >= 4096 ? 8 : (n >= 2048 ? 4 : (n >= 1024 ? 2 : 1));
  double *dX, *dv, *dv2 = nullptr, *part, *part2 = nullptr, *dout;
  long ldd;
  if (int rc = upload_padded(sc, x, n, p, ld, U, &dX, &ldd)) return rc;
  if (int rc = upload_vec_padded(sc, v, n, ldd, &dv)) return rc;
  if (v2)
    if (int rc = upload_vec_padded(sc, v2, n, ldd, &dv2)) return rc;
  int nrb = (int)(ldd / (128L * U));
  HIPX(sc.alloc(&part, (size_t)nrb * p));
  HIPX(sc.alloc(&part2, (size_t)nrb * p));
  HIPX(sc.alloc(&dout, (size_t)p));
  HIPX(launch_xtv(dX, ldd, p, U, dv, dv2, part, part2, nullptr, 0, nullptr));
  HIPX(launch_part_sum(part, nrb, p, dout, nullptr));
  HIPX(hipMemcpy(out, dout, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  if (v2 && out2) {
    HIPX(launch_part_sum(part2, nrb, p, dout, nullptr));
    HIPX(hipMemcpy(out2, dout, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  }
  return BESSX_OK;
}

int bessx_op_topk(const double *score, int len, int k, int *out_idx) {
  if (int rc = need_device()) return rc;
  if (!score || !out_idx || len < 1 || k < 0 || k > len) return fail(BESSX_ERR_ARG, "op_topk: bad arguments");
  if (k == 0) return BESSX_OK;
  if (!topk_supported(len, k)) return fail(BESSX_ERR_UNSUPPORTED, "op_topk: len / k combination needs a third level");
  Scratch sc;
  double *ds;
  int *dout, *dcand;
  HIPX(sc.alloc(&ds, (size_t)len));
  HIPX(sc.alloc(&dout, (size_t)k));
  int *dtie;
  HIPX(sc.alloc(&dcand, (size_t)32768));
  HIPX(sc.alloc(&dtie, (size_t)3 * len + 8));
  HIPX(hipMemset(dtie, 0, 8 * sizeof(int)));
  HIPX(hipMemcpy(ds, score, (size_t)len * sizeof(double), hipMemcpyHostToDevice));
  const TopkTie tie = {dtie, dtie + 8};
  HIPX(launch_topk(ds, len, k, dout, dcand, nullptr, 0, nullptr, nullptr, nullptr, &tie));
  HIPX(hipMemcpy(out_idx, dout, (size_t)k * sizeof(int), hipMemcpyDeviceToHost));
  return BESSX_OK;
}

int bessx_op_topk_bench(int len, int k, int variant, int repeats, double *avg_us) {
  if (int rc = need_device()) return rc;
  if (len < 1 || k < 1 || k > len || repeats < 1 || !avg_us) return fail(BESSX_ERR_ARG, "op_topk_bench: bad arguments");
  if (!topk_supported(len, k)) return fail(BESSX_ERR_UNSUPPORTED, "op_topk_bench: len / k combination needs a third level");
  Scratch sc;
  double *ds;
  int *dout, *dcand;
  HIPX(sc.alloc(&ds, (size_t)len));
  HIPX(sc.alloc(&dout, (size_t)k));
  HIPX(sc.alloc(&dcand, (size_t)32768));
  std::vector<double> h((size_t)len);
  std::mt19937_64 g(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  for (auto &v : h) {
    const double z = nd(g);
    v = z * z;
  }
  HIPX(hipMemcpy(ds, h.data(), (size_t)len * sizeof(double), hipMemcpyHostToDevice));
  topk_set_variant(variant);
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  hipError_t e = launch_topk(ds, len, k, dout, dcand, nullptr, 0, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
  for (int i = 0; i < repeats && e == hipSuccess; i++) e = launch_topk(ds, len, k, dout, dcand, nullptr, 0, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  topk_set_variant(1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIPX(e);
  *avg_us = 1e3 * (double)ms / repeats;
  return BESSX_OK;
}

int bessx_op_gram(const double *x, int n, int p, int ld, const int *cols, int m, const double *w, double *out) {
  if (int rc = need_device()) return rc;
  if (!x || !cols || !out || n < 1 || p < 1 || ld < n || m < 1 || m > T0_CAP) return fail(BESSX_ERR_ARG, "op_gram: bad arguments");
  for (int i = 0; i < m; i++)
    if (cols[i] < 0 || cols[i] >= p) return fail(BESSX_ERR_ARG, "op_gram: column index out of range");
  Scratch sc;
  const int U = 1;
  HIPX(gram_lds_prepare());
  if (const char *ev = std::getenv("BESSX_GRAM")) gram_set_variant(std::string(ev) == "direct" ? 0 : 1);
  double *dX, *dw = nullptr, *daux, *gpart, *Gt;
  long ldd;
  if (int rc = upload_padded(sc, x, n, p, ld, U, &dX, &ldd)) return rc;
  if (w)
    if (int rc = upload_vec_padded(sc, w, n, ldd, &dw)) return rc;
  HIPX(sc.alloc(&daux, (size_t)ldd * 3));
  HIPX(hipMemset(daux, 0, (size_t)ldd * 3 * sizeof(double)));
  const int mt = (m + 15) / 16, mp = mt * 16, ntiles = mt * (mt + 1) / 2;
  std::vector<int> hc(mp, -1);
  std::copy(cols, cols + m, hc.begin());
  int *dcols;
  HIPX(sc.alloc(&dcols, (size_t)mp));
  HIPX(hipMemcpy(dcols, hc.data(), (size_t)mp * sizeof(int), hipMemcpyHostToDevice));
  std::vector<GramTask> tasks;
  build_gram_tasks(mt, tasks);
  GramTask *dt;
  HIPX(sc.alloc(&dt, tasks.size()));
  HIPX(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(GramTask), hipMemcpyHostToDevice));
  bessx_session fake;
  fake.ld = ldd;
  int rps, nslab;
  gram_geometry(&fake, (int)tasks.size(), &rps, &nslab);
  HIPX(sc.alloc(&gpart, (size_t)nslab * ntiles * 256));
  HIPX(sc.alloc(&Gt, (size_t)ntiles * 256));
  HIPX(launch_gram(dX, daux, ldd, dcols, dw, rps, dt, (int)tasks.size(), nslab, gpart, ntiles, Gt, nullptr, 0, 0,
                   nullptr, 0));
  std::vector<double> ht((size_t)ntiles * 256);
  HIPX(hipMemcpy(ht.data(), Gt, ht.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int I = 0; I < mt; I++)
    for (int J = 0; J <= I; J++) {
      int t = I * (I + 1) / 2 + J;
      for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 4; r++) {
          int row = I * 16 + (lane >> 4) + 4 * r, col = J * 16 + (lane & 15);
          if (row < m && col < m) {
            double v = ht[(size_t)t * 256 + lane * 4 + r];
            out[(size_t)col * m + row] = v;
            if (I != J) out[(size_t)row * m + col] = v;
          }
        }
    }
  return BESSX_OK;
}

int bessx_op_chol_solve(const double *a, int m, const double *b, double *sol) {
  if (int rc = need_device()) return rc;
  if (!a || !b || !sol || m < 1 || m > T0_CAP) return fail(BESSX_ERR_ARG, "op_chol_solve: need 1 <= m <= 2046");
  Scratch sc;
  const int mt = (m + 1 + 15) / 16, ntiles = mt * (mt + 1) / 2;
  std::vector<double> ht((size_t)ntiles * 256, 0.0);
  for (int I = 0; I < mt; I++)
    for (int J = 0; J <= I; J++) {
      int t = I * (I + 1) / 2 + J;
      for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 4; r++) {
          int row = I * 16 + (lane >> 4) + 4 * r, col = J * 16 + (lane & 15);
          if (row < m && col < m) ht[(size_t)t * 256 + lane * 4 + r] = a[(size_t)col * m + row];
        }
    }
  double *Gt, *drhs, *dsol;
  int *dinfo;
  HIPX(sc.alloc(&Gt, ht.size()));
  HIPX(sc.alloc(&drhs, (size_t)m));
  HIPX(sc.alloc(&dsol, (size_t)m));
  HIPX(sc.alloc(&dinfo, 1));
  HIPX(hipMemset(dinfo, 0, sizeof(int)));
  HIPX(hipMemcpy(Gt, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPX(hipMemcpy(drhs, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
  if (mt <= 16) {
    double *dfb;
    HIPX(sc.alloc(&dfb, CHOL_FB_DOUBLES));
    CholFuse fbz = {};
    fbz.fb_work = dfb;  // (a singular / indefinite matrix goes to the pivoted solve, like in the fits)
    HIPX(launch_chol(Gt, m, mt, 0.0, 0, drhs, nullptr, dsol, dinfo, nullptr, 0, 0, nullptr, &fbz));
    HIPX(launch_sym_fallback(Gt, m, mt, 0.0, 0, drhs, nullptr, dsol, dinfo, nullptr, 0, nullptr, &fbz));
  } else {
    double *rd, *zz;
    HIPX(sc.alloc(&rd, (size_t)mt * 16));
    HIPX(sc.alloc(&zz, (size_t)mt * 16));
    HIPX(launch_chol_big(Gt, m, mt, 0.0, 0, drhs, nullptr, dsol, dinfo, rd, zz, nullptr, 0, 0, nullptr));
  }
  HIPX(hipMemcpy(sol, dsol, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
  int info = 0;
  HIPX(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost));
  if (info) return fail(BESSX_ERR_NUMERIC, "op_chol_solve: non-finite solution");
  return BESSX_OK;
}

int bessx_op_chol_bench(int m, int repeats, double *avg_us) {
  if (int rc = need_device()) return rc;
  if (m < 1 || m > T0_FAST || repeats < 1 || !avg_us) return fail(BESSX_ERR_ARG, "op_chol_bench: need 1 <= m <= 254");
  Scratch sc;
  const int mt = (m + 1 + 15) / 16, ntiles = mt * (mt + 1) / 2;
  // a well conditioned matrix: 4 I + small symmetric off-diagonal entries
  std::vector<double> ht((size_t)ntiles * 256, 0.0);
  for (int I = 0; I < mt; I++)
    for (int J = 0; J <= I; J++) {
      int t = I * (I + 1) / 2 + J;
      for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 4; r++) {
          int row = I * 16 + (lane >> 4) + 4 * r, col = J * 16 + (lane & 15);
          if (row < m && col < m)
            ht[(size_t)t * 256 + lane * 4 + r] = row == col ? 4.0 : 0.01 * std::cos(0.37 * (row + 1) * (col + 1));
        }
    }
  double *Gt, *drhs, *dsol;
  int *dinfo;
  HIPX(sc.alloc(&Gt, ht.size()));
  HIPX(sc.alloc(&drhs, (size_t)m));
  HIPX(sc.alloc(&dsol, (size_t)m));
  HIPX(sc.alloc(&dinfo, 1));
  HIPX(hipMemset(dinfo, 0, sizeof(int)));
  HIPX(hipMemcpy(Gt, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPX(launch_fill(drhs, m, 1.0, nullptr));
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  hipError_t e = launch_chol(Gt, m, mt, 0.0, 0, drhs, nullptr, dsol, dinfo, nullptr, 0, 0, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
  for (int i = 0; i < repeats && e == hipSuccess; i++)
    e = launch_chol(Gt, m, mt, 0.0, 0, drhs, nullptr, dsol, dinfo, nullptr, 0, 0, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIPX(e);
  *avg_us = 1e3 * (double)ms / repeats;
  return BESSX_OK;
}

int bessx_op_normalize(double *x, int n, int p, double *y, const double *weight, int data_type, int is_normal,
                       int add_weight, double *x_mean, double *x_norm, double *y_mean) {
  if (int rc = need_device()) return rc;
  if (!x || !y || !weight || n < 1 || p < 1) return fail(BESSX_ERR_ARG, "op_normalize: bad arguments");
  Scratch sc;
  double *dX, *dy, *dw, *dm, *dn, *dym;
  long ldd;
  if (int rc = upload_padded(sc, x, n, p, n, 1, &dX, &ldd)) return rc;
  if (int rc = upload_vec_padded(sc, y, n, ldd, &dy)) return rc;
  if (int rc = upload_vec_padded(sc, weight, n, ldd, &dw)) return rc;
  HIPX(sc.alloc(&dm, (size_t)p));
  HIPX(sc.alloc(&dn, (size_t)p));
  HIPX(sc.alloc(&dym, 1));
  HIPX(hipMemset(dm, 0, (size_t)p * sizeof(double)));
  HIPX(hipMemset(dn, 0, (size_t)p * sizeof(double)));
  HIPX(launch_normalize(dX, ldd, n, p, dy, dw, data_type, is_normal, add_weight, dm, dn, dym, nullptr));
  HIPX(hipMemcpy2D(x, (size_t)n * sizeof(double), dX, (size_t)ldd * sizeof(double), (size_t)n * sizeof(double),
                   (size_t)p, hipMemcpyDeviceToHost));
  HIPX(hipMemcpy(y, dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  if (x_mean) HIPX(hipMemcpy(x_mean, dm, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  if (x_norm) HIPX(hipMemcpy(x_norm, dn, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  if (y_mean) HIPX(hipMemcpy(y_mean, dym, sizeof(double), hipMemcpyDeviceToHost));
  return BESSX_OK;
}

int bessx_op_xtv_bench(int n, int p, int variant, int repeats, double *gbps, double *avg_ms) {
  if (int rc = need_device()) return rc;
  if (n < 1 || p < 1 || repeats < 1 || !gbps) return fail(BESSX_ERR_ARG, "op_xtv_bench: bad arguments");
  Scratch sc;
  const long ld = ((long)n + 1023) / 1024 * 1024;  // valid for every variant (multiple of 128*U)
  double *dX, *dv, *part;
  HIPX(sc.alloc(&dX, (size_t)ld * p));
  HIPX(sc.alloc(&dv, (size_t)ld));
  HIPX(sc.alloc(&part, (size_t)(ld / 128) * p));
  HIPX(launch_fill(dX, ld * (long)p, 1.0, nullptr));
  HIPX(launch_fill(dv, ld, 0.5, nullptr));
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  HIPX(launch_xtv_variant(variant, dX, ld, p, dv, part, nullptr));
  HIPX(hipEventRecord(e0, nullptr));
  for (int i = 0; i < repeats; i++) HIPX(launch_xtv_variant(variant, dX, ld, p, dv, part, nullptr));
  HIPX(hipEventRecord(e1, nullptr));
  HIPX(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPX(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = 8.0 * (double)n * (double)p * repeats / ((double)ms * 1e-3) / 1e9;
  if (avg_ms) *avg_ms = ms / repeats;
  return BESSX_OK;
}

int bessx_op_cox_score_bench(int n, int p, int variant, int repeats, double *gbps, double *avg_ms) {
  if (int rc = need_device()) return rc;
  if (n < 1 || p < 1 || repeats < 1 || !gbps) return fail(BESSX_ERR_ARG, "op_cox_score_bench: bad arguments");
  Scratch sc;
  const long ld = ((long)n + 1023) / 1024 * 1024;
  const int nrb = (int)(ld / 1024);
  double *dX, *vec, *out;
  HIPX(sc.alloc(&dX, (size_t)ld * p));
  HIPX(sc.alloc(&vec, (size_t)ld * 4));
  HIPX(sc.alloc(&out, (size_t)5 * nrb * p + nrb));
  HIPX(launch_fill(dX, ld * (long)p, 1.0, nullptr));
  HIPX(launch_fill(vec, ld * 4, 0.5, nullptr));
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  // variant 1 (what the solver runs): consecutive waves take consecutive column groups of one row block; 0 (round 3):
  // the row blocks of one column group
  if (variant < 0 || variant > 1) return fail(BESSX_ERR_ARG, "op_cox_score_bench: variant 0 or 1");
  cox_score_set_variant(variant);
  CoxBufs cb = {};
  cb.one_pass = 1;
  cb.TH = vec;
  cb.CU = vec + ld;
  cb.CV = vec + 2 * ld;
  cb.C2 = vec + 3 * ld;
  HIPX(launch_cox_score_pass(dX, ld, p, 8, nrb, cb, out, nullptr, nullptr, 0, nullptr));
  HIPX(hipEventRecord(e0, nullptr));
  for (int i = 0; i < repeats; i++) HIPX(launch_cox_score_pass(dX, ld, p, 8, nrb, cb, out, nullptr, nullptr, 0, nullptr));
  HIPX(hipEventRecord(e1, nullptr));
  HIPX(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPX(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = 8.0 * (double)n * (double)p * repeats / ((double)ms * 1e-3) / 1e9;
  if (avg_ms) *avg_ms = ms / repeats;
  cox_score_set_variant(1);
  return BESSX_OK;
}

int bessx_op_stream_copy_gbps(long long bytes, int repeats, double *gbps) {
  if (int rc = need_device()) return rc;
  if (bytes < (1 << 20) || repeats < 1 || !gbps) return fail(BESSX_ERR_ARG, "op_stream_copy: bad arguments");
  Scratch sc;
  double *a, *b;
  size_t n = (size_t)bytes / 16 * 2;
  HIPX(sc.alloc(&a, n));
  HIPX(sc.alloc(&b, n));
  HIPX(hipMemset(a, 1, n * sizeof(double)));
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  HIPX(launch_copy(a, b, (long)n, nullptr));
  HIPX(hipEventRecord(e0, nullptr));
  for (int i = 0; i < repeats; i++) HIPX(launch_copy(a, b, (long)n, nullptr));
  HIPX(hipEventRecord(e1, nullptr));
  HIPX(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPX(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = 2.0 * (double)n * 8.0 * repeats / ((double)ms * 1e-3) / 1e9;
  return BESSX_OK;
}

}  // extern "C"
