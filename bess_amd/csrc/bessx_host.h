#ifndef BESSX_HOST_H
#define BESSX_HOST_H
// bessx_host.h -- internal header of the host side of libbessx.so (bessx_session.cpp, bessx_fit.cpp, bessx_cv.cpp,
// bessx_paths.cpp, bessx_abi.cpp): the session (Data + Algorithm + Metric of the reference's
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
#include "bessx_sync.h"

namespace bessx {

extern thread_local std::string g_err;  // (defined in bessx_session.cpp)
// set while a session is created for the marginal fit of one wide group of the screening (screening(),
// src/screening.cpp:42-63): 1 = logit_fit (no weight floor), 2 = cox_fit (linear predictor clamped at 50)
extern thread_local int g_marginal_fit_variant;

// Test hooks.  ONE environment variable, BESSX_TEST_HOOKS = "name=value,name=value,...", read when a session is created:
// the tests use it to force the fallback paths the library keeps anyway (an unfused launch sequence, the Cholesky behind
// the conjugate gradients, the two-pass Cox forms, a small Gram column cache, ...) so that they are compared with the
// defaults.  Not product configuration -- those are BESSX_SCORE_MODE, BESSX_WAIT_TIMEOUT_S, BESSX_POOL_SPIN_US and
// BESSX_DEBUG.  Returns the value of `name`, or nullptr.
inline const char *test_hook(const char *name) {
  static thread_local std::string val;
  const char *ev = std::getenv("BESSX_TEST_HOOKS");
  if (!ev) return nullptr;
  const std::string all(ev), key = std::string(name) + "=";
  size_t pos = 0;
  while (pos < all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    if (all.compare(pos, key.size(), key) == 0) {
      val = all.substr(pos + key.size(), end - pos - key.size());
      return val.c_str();
    }
    pos = end + 1;
  }
  return nullptr;
}

inline int fail(int code, const std::string &msg) {
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

namespace bessx {
struct KChains;
}

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
  // LM: the diagonalisation of every group's block 2 lambda I + X_g^T X_g / n (k_group_score's Jacobi sweeps) depends on
  // the row set and lambda only: kept per row set (eigenvectors, eigenvalues) with the lambda it was formed for
  std::vector<double *> geig_v_rs, geig_l_rs;
  std::vector<double> geig_lambda;
  std::vector<char> geig_valid;
  bool geig_on = true;  // (test hook group_eig=0: diagonalise at every iteration, round 3's form)
  int cox_state_rs = -1;
  int dev_state_rs = -1;                // row set of the fit whose final coefficients sit in A_cur/b_cur/beta_dense
  CoxBufs cox = {};                     // Cox work space (model_type 4 only)
  std::vector<void *> cox_allocs;
  int *idcols = nullptr;
  struct RsCache {
    bool valid = false;  // part_rs / r_rs belong to exactly (beta, coef0) below
    bool model_only = false;  // (covariance form) the device holds (beta, coef0) below, but the score sums are those of
                              // the coefficients BEFORE the last solve: a fit that ended on a cycle of active sets
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
    int *slot_w = nullptr;  // chunk chains with staged fills: the WRITER's slot map (slot_of is what the readers see)
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
  int cov_spec_min = 8;    // a private fill's list is rounded up to the multiple of 32 that leaves room for this many (test hook cov_spec_min)
  bool fuse_sel = true;    // selection + solve of a slot in one launch, k_sel_cgr (test hook fuse_sel=0: two launches)
  bool cg_by_rows = true;  // row-dealt kernel k_cgr for systems of up to 208 unknowns (test hook cg_layout=tiles: k_cg)
  // GLM IRLS step in three launches instead of five: linear predictor, weights, working response and the slab Gram
  // in ONE pass over the active columns (k_irls_gram), the reduction, then the convergence test at the head of the
  // solve.  (Round 2's k_gram_irls did the per-row work 64 rows at a time between the barriers of the staging pipeline
  // and lost, 0.180 s against 0.175 s on configs[2]; it is gone.)
  bool irls_fuse = true;   // GLM IRLS step as k_irls_gram + k_gram_reduce + k_chol (test hook irls_fuse=0: the five-launch step)
  bool glm_fallback = false;  // the IRLS chain carries the pivoted fallback solve behind every k_chol (set, and the
                              // fit redone, the first time a k_chol of this session meets a rank-deficient system)
  int irls_wfloor = 1;     // floor of the logistic IRLS weight inside the loop (src/Algorithm.h:1188-1192); 0 in the
                           // sub-sessions that run logit_fit for the screening of wide groups (src/logistic.cpp:60-160)
  size_t llpart_cap = 0;
  long long n_submodel_steps = 0;  // IRLS / Newton steps taken since the last reset (bessx_session_submodel_steps)
  bool defer_pub = true;   // chained fits publish through a snapshot + the next launch (test hook defer_publish=0: in the tail)
  bool fuse = true;  // small-kernel fusions of the covariance form (SlotFuse); test hook fuse=0 turns them off
  bool cov_cg = true;          // solve by k_cg (falls back to k_chol per slot); test hook cov_solver=chol switches it off
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
  // every chain is quiet, so nobody reads the shared slot map while it is rewritten.  test hook cv_side_by_side=0: the
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
  double *cov_part = nullptr, *bd2 = nullptr;
  unsigned char *inA = nullptr;        // 1 for the columns of the current active set
  double *cov_bmm = nullptr;           // per-block min / max of k_cov_d's repeated-set shortcut (+ arg-max columns)
  int bmm_owner = -1;                  // row set of the k_cov_d launch that wrote cov_bmm last
  int *cov_fcols = nullptr, *cov_extras = nullptr;
  long long cov_panel_groups = 0;  // 32-column panel passes over X really executed (host statistics)
  // The sequential path of ONE problem as several chunk chains at once (bessx_kchunks.cpp): the stitched k-path of the
  // multi-GPU run inside one device.  A coarse warm-start chain over the chunk boundaries fills the Gram column cache
  // and hands every chunk its starting model; the chunks then run side by side, each on a fit context of its own
  // (stream, control block, scores, solve work space) that READS the one cache of the all-rows row set; a chain that
  // needs a column the cache lacks fills it while every other chain stands still (KChains: safe points between
  // candidates); the chunks are stitched into the single chain exactly as bess_amd.dist.StitchedKPath does it.
  std::vector<void *> ctx_allocs;       // chain context: device buffers it owns beyond a fold context's
  bessx::KChains *kch = nullptr;        // parent: contexts, host threads, the fill rendezvous (created at first use)
  bessx_session *kch_owner = nullptr;   // chain context: the session whose cache it reads
  int kpath_chains = 0;                 // 0 = automatic, 1 = one chain (off), C >= 2 = that many chunk chains
  int kch_index = -1;                   // (chain context) its place among the owner's contexts
  unsigned long long kch_gen_seen = 0;  // (chain context) completed fills when this chain last queued a look-up
  bool light_confirm = true;            // GLM / Cox: the tail right behind the head of a PDAS iteration >= 2 (bessx_fit.cpp)
  bool kch_fill_tried = false;          // the fill stream of the staged fills was asked for once (there or not)
  bool kch_sp_member = false;           // (chain context) its thread takes part in the owner's shared passes over X
  int kch_sp_group = 0;                 // ... in this group of chains (the groups alternate on the pass stream)
  long long sp_launches = 0, sp_chain_slots = 0, sp_partial = 0;  // shared passes: launches, open gates in them, batches cut short
  int *kch_slot_w = nullptr;            // (owner) the writer's slot map of staged fills, p ints
  hipStream_t kch_fill_st = nullptr;    // (owner) the stream the chains' staged fills run on (some compute units left out)
  hipEvent_t kch_ev = nullptr;          // (chain context) orders its fill list in front of the fill on kch_fill_st
  long long kch_merged = 0, kch_takeovers = 0;  // chunk phases run as merged launches; chains the host had to finish
  long long kch_paths = 0, kch_refits = 0, kch_chunk_fills = 0;  // paths run chunked, stitch refits, fills in the chunk phase
  int kch_last_chains = 0;              // chains of the last chunked path
  bool kch_auto_off = false;            // the chunks of a path did not merge with the chain: the automatic choice is one chain
  int panel_variant = 0;                // 5: k_cov_panel_dp for the fills (default; test hook panel=lds: 0, the round-2 kernels)
  int cov_panel_blocks = 0;             // workgroups of one panel pass with the slab count chosen at creation
  bool own_hw_queue = false;            // (fit contexts) the context's stream has a hardware queue outside the runtime's pool
  long long group_xtx_ns = 0;           // device time of the last all-rows group_XTX pass inside a path call (LM, timing on)
  hipEvent_t xtx_ev[2] = {nullptr, nullptr};  // around the last such pass (read when counter 19 is asked for)
  bool xtx_ev_pending = false;
  bool path_group_xtx = true;           // every cold path call redoes the all-rows group_XTX pass (src/path.cpp:37)
  long long kch_giveups = 0;            // paths whose stitch ran out of budget (the rest was walked as one chain)
  double kch_t[3] = {0, 0, 0};          // BESSX_DEBUG: seconds in the coarse chain / the chunks / the stitch
  bessx_fill_hook fill_hook = nullptr;  // shared wide fills of a parked fit (bessx_session_set_fill_hook)
  void *fill_hook_user = nullptr;
  int fill_hook_width = 0;
  long long shared_wide_fills = 0;
  int prefill_cols = 0;            // columns listed by bessx_session_cov_prefill_begin (0: no prefill in progress)
  int prefill_base = 0;            // ... first cache slot of that list (0, or the occupancy cov_prefill_extend found)
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
  bool chain = true;   // test hook chain=0 switches the chaining off
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
  // panel launches of the covariance form by width (index 0: one 32-column group, 1: two): X is read ONCE per launch
  // whatever the width, the flops (2 n p 32 per group) double -- two different roofs (HBM / fp64 matrix cores)
  double panel_w_seconds[2] = {0.0, 0.0};
  long long panel_w_launches[2] = {0, 0};
  long long n_fits = 0, n_iters = 0;
};

namespace bessx {

static constexpr int COV_R = 32;        // columns per panel group (matches the kernels)
static constexpr int COV_SLOT_GROUPS = 2;  // groups an ordinary PDAS slot launches
static constexpr int COV_CS = 512;         // side of the slot-indexed Gram of the cached columns (2 MiB)

// Device memory comes back with whatever it last held: on a fresh box mostly zeros, in a long-lived process the bytes of
// earlier sessions.  BESSX_TEST_HOOKS=poison=1 fills every allocation with 0xFF bytes (a NaN for a double, -1 for an int),
// so that a kernel that reads a buffer nobody wrote shows up in the tests instead of on somebody's machine.
inline bool poison_allocations() {  // (read at every allocation: allocations are rare, and a test switches it per session)
  const char *v = test_hook("poison");
  return v && std::string(v) == "1";
}

template <class T>
hipError_t dmalloc(T **ptr, size_t count) {
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void **>(ptr), bytes);
  if (e == hipSuccess && poison_allocations()) e = hipMemset(*ptr, 0xff, bytes);
  return e;
}

struct Scratch {
  std::vector<void *> ptrs;
  ~Scratch() {
    for (void *q : ptrs) (void)hipFree(q);
  }
  template <class T>
  hipError_t alloc(T **out, size_t count) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(out), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess && poison_allocations()) e = hipMemset(*out, 0xff, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*out);
    return e;
  }
};

inline int caller_col(const bessx_session *s, int j) { return s->screen_map.empty() ? j : s->screen_map[j]; }

struct SlotFuse;
struct Candidate {  // one (s, lambda) of a path: the full-data fit's model (normalised scale) and its criteria
  int T0 = 0;
  double lambda = 0.0;
  SparseVec beta;
  double coef0 = 0.0, loss = 0.0, ic = 0.0;
  int iters = 0;
};
struct PgsArgs;

// ---- bessx_session.cpp / bessx_fit.cpp / bessx_cv.cpp / bessx_paths.cpp: what they call across files
void fold_ctx_free(bessx_session *c);
void drop_fold_contexts(bessx_session *s);
void session_free(bessx_session *s);
size_t part_elems(const bessx_session *s);
int k1_begin(bessx_session *s, hipEvent_t *a, hipEvent_t *b);
int k1_collect(bessx_session *s, const std::vector<std::pair<size_t, bool>> &pairs);
int cov_collect(bessx_session *s, int nfill);
int alloc_gram_cache(bessx_session *s);
int alloc_cov_cache(bessx_session *s, bool share_map = false);
int reset_path_caches(bessx_session *s);
CholFuse chol_fallback_only(const bessx_session *s);
void build_gram_tasks(int mt, std::vector<GramTask> &out);
int gram_tasks_for(bessx_session *s, int mt, const GramTask **tasks, int *ntask);
void gram_geometry(const bessx_session *s, int ntask, int *rows_per_slab, int *nslab, int ntiles = 0,
                          bool allow_lds = true);
int upload_x(bessx_session *s, const double *x, int col_major);
int prepare_rowset(bessx_session *s, int rs, bool keep_yy = false);
int cov_C_dev(const bessx_session *s);
bool cov_speculates(const bessx_session *s);
int enqueue_lm_slot(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                           std::vector<std::pair<size_t, bool>> &k1_pairs, int part = 0);
int panel_variant_for(const bessx_session *s, int ng);
int enqueue_cov_fill(bessx_session *s, int rs, int ngroups, int parked, const FitCtrl *gate = nullptr,
                            int gfirst = 0, bool compact = true, const int *slot_map = nullptr);
CholFuse cov_fuse_args(bessx_session *s, int rs, int T0, bool force_chol, SlotFuse *sf);
int cgb_reserve(bessx_session *s);
int enqueue_cov_tail(bessx_session *s, int slot, int T0, double lambda, int rs, bool force_chol = false,
                            SlotFuse *sf = nullptr);
int enqueue_lm_slot_cov(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_d,
                               bool scores_ok = false, bool grow1 = false, SlotFuse *sf = nullptr);
int cov_unpark(bessx_session *s, const FitCtrl *hc, int T0, double lambda, int rs, int *next_slot);
// bessx_kchunks.cpp
bool kchunks_apply(const bessx_session *s, const int *seq, int ns, int nl, int is_cv, const bessx_path_chain *chain);
int sequential_path_chunked(bessx_session *s, const int *seq, int ns, double lambda, int ic_type, bessx_path_result *res,
                            bessx_path_chain *chain = nullptr);
void kchains_safe_point(bessx_session *c);   // chain context, between candidates: stand still while another chain fills
int kchains_fill_begin(bessx_session *c);    // chain context parked on missing columns: wait until it alone runs
void kchains_fill_end(bessx_session *c, bool filled = true);
void kchains_log(bessx_session *c, const char *what, int a, int b);  // (test hook kchunks_log=1)
void kchains_progress(bessx_session *c, int n);  // chain context: n candidates of its chunk are stored
bool kchains_staged(const bessx_session *c);               // this round's fills are staged (nobody stands still)
unsigned long long kchains_generation(bessx_session *c);   // completed fills of the owner's chains so far
bool shared_pass_applies(const bessx_session *c);  // chain context whose passes over X go into the owner's multi-chain launches
int shared_pass_submit(bessx_session *c, const double *v, const double *v2, double *part, double *part2,
                       const CoxBufs *cox, const FitCtrl *ctrl, int slot);
void kchains_free(bessx_session *s);
void kchains_quiesce(bessx_session *s);
int kchunks_prepare(bessx_session *s, int ns, bool link, bool link_warm = false);
hipError_t cox_alloc(bessx_session *s);                        // bessx_session.cpp
int chain_ctx_create(bessx_session *ps, bessx_session **out);  // bessx_session.cpp
void chain_ctx_free(bessx_session *c);
int prefill_begin(bessx_session *s, const int *cols, int ncols, int append);  // bessx_paths.cpp
int glm_geometry(bessx_session *s, int T0, int *mt, int *mp, int *ntask, int *ntiles, int *rps, int *nslab);
int enqueue_glm_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs);
int enqueue_glm_irls_step(bessx_session *s, int slot, int t, int T0, double lambda, int rs);
int enqueue_glm_tail(bessx_session *s, int slot, int T0, int rs);
int cox_reserve(bessx_session *s, int T0);
int enqueue_cox_head(bessx_session *s, int slot, int T0, double lambda, int rs, bool skip_k1,
                            std::vector<std::pair<size_t, bool>> &k1_pairs);
int enqueue_cox_newton(bessx_session *s, int slot, int t, int T0, double lambda, int rs);
int enqueue_cox_tail(bessx_session *s, int slot, int T0, int rs);
PubArgs publish_args(bessx_session *s, int kcopy, int buf, unsigned long long *seq);
PubArgs publish_from_snapshot(const PubArgs &tail);
int publish_flush(bessx_session *s);
int publish_launch(bessx_session *s, const PubArgs &pa);
int publish_enqueue(bessx_session *s, int kcopy, int buf, unsigned long long *seq);
int publish_wait(bessx_session *s, int buf, unsigned long long want);
bool ctx_stream_create(int device, hipStream_t *st, int leave_out = 0, int stride = 1);
long long ctx_streams_created();
void ctx_stream_destroy(hipStream_t st);  // (a stream of ctx_stream_create, or any other: the own-queue count is kept)  // a stream with a hardware queue outside the runtime's pool
bool ctx_streams_own_queue(int device);              // ... does that work on this device (asked once per process)
int stream_wait_bounded(bessx_session *s, hipStream_t st, const char *what);  // hipStreamSynchronize with the deadline
int read_results(bessx_session *s, int kcopy = -1);
int algorithm_fit_grouped(bessx_session *s);
int enqueue_chained(bessx_session *s, const bessx_session::Hint &hint, int rs, int parent, int buf, int batch,
                           double parent_lambda, int parent_T0);
int algorithm_fit(bessx_session *s);
void fold_contexts_invalidate(bessx_session *s);
bool side_by_side_applies(const bessx_session *s, int T0);
int fold_fits_side_by_side(bessx_session *s, double *out, const std::vector<int> *only = nullptr,
                                  double *per_fold = nullptr);
double metric_train_loss_value(const bessx_session *s);
double metric_fold_test_loss(const bessx_session *s, int k);
int metric_train_loss(bessx_session *s, double *out);
int metric_test_loss(bessx_session *s, double *out);
int metric_ic(bessx_session *s, int ic_type, int is_cv, double *out);
void denormalize(const bessx_session *s, SparseVec &b, double &coef0, bool gs_variant);
int run_fit(bessx_session *s, int T0, double lambda, const SparseVec &beta_init, double coef0_init);
void store_candidate(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant);
void store_best(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant);
bool chain_row_matches(const bessx_session *s, const bessx_path_chain *ch, int row, const Candidate &c);
int sequential_path(bessx_session *s, const int *seq, int ns, const double *lam, int nl, int ic_type,
                           int is_cv, bessx_path_result *res, bessx_path_chain *chain = nullptr);
int gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res);
int pgs_path(bessx_session *s, int s_min, int s_max, double lmin, double lmax, int powell_path, int nlambda,
                    int ic_type, int is_cv, bessx_path_result *res);
int settle_device_chain(bessx_session *s);
int run_path(bessx_session *s, bool gs, const int *seq, int ns, const double *lam, int nl, int s_min,
                    int s_max, int ic_type, int is_cv, bessx_path_result *res, const PgsArgs *pgs = nullptr,
                    bessx_path_chain *chain = nullptr);
int need_device();
int upload_padded(Scratch &sc, const double *x, int n, int p, int ld_in, int U, double **dX, long *ld_out);
int upload_vec_padded(Scratch &sc, const double *v, int n, long ld, double **dv);

}  // namespace bessx

#endif  // BESSX_HOST_H
