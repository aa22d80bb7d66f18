// bessx_dev.h -- internal interface between the HIP kernels (bessx_kernels.hip) and the host-side
// solver (bessx_session.cpp, bessx_fit.cpp, bessx_cv.cpp, bessx_paths.cpp, bessx_abi.cpp).  Not part of the C ABI (that is include/bessx.h).
#ifndef BESSX_DEV_H
#define BESSX_DEV_H

#include <hip/hip_runtime.h>

namespace bessx {

// Device-resident control block of one Algorithm::fit (src/Algorithm.h:113-171).  The host enqueues
// PDAS iteration "slots" speculatively; every kernel of slot s runs only if l == s-1 && !done, so
// slots issued after convergence fall through in a few microseconds without a host round trip.
struct FitCtrl {
  int done;        // active set repeated: Algorithm::fit returned
  int l;           // PDAS iterations committed so far (Algorithm::l)
  int T0;          // sparsity level of this fit
  int k_cur;       // support size of the current beta
  int irls_done;   // GLM sub-model fit converged (per PDAS iteration)
  int irls_steps;  // IRLS / Newton steps taken in the current PDAS iteration
  int info;        // non-zero: a k x k solve produced a non-finite value
  int same_prev;   // this slot's active set equals the previous iteration's: solve + residual skipped
  double coef0;    // current intercept
  double ll0;      // GLM: log-likelihood of the previous iterate
  int d_fresh;     // the score-pass partial sums in memory were computed from the CURRENT coefficients
  int irls_last;   // IRLS steps the last committed sub-model fit took (host sizes its next batch from it)
  // Cox Newton line search (src/Algorithm.h:1474-1481)
  int fast_same;   // covariance form: k_cov_d found the selection unchanged (consumed by the next k_topk)
  int ls_m;        // accepted exponent m (step 0.5^m)
  double ll1;      // partial log-likelihood at the trial point
  int gram_full;   // LM Gram cache: 1 = form the whole Gram this slot, 0 = only the rows of the new columns
  // covariance-update mode (LM): see the k_cov_* kernels
  int cov_nfill;   // length of the current fill list (a multiple of 32): missing + speculative columns
  int cov_stall;   // the new active set has uncached columns: the fit is parked (l = -1 - l) until the host has
                   // issued the fill
  int cov_groups;  // 32-column panel groups (passes over X) this fit has formed so far (host statistics)
  int cov_miss;    // internal error flag: an active column was not in the Gram column cache
  int cov_nmiss;   // columns of the requested set that are not cached (set by k_cov_need)
  int serial;      // which fit this block describes (set by k_fit_continue; the host checks it for chained fits)
  int sse_valid;   // sse_dot / sse_nrm belong to the current coefficients (set by the fused k_chol)
  double sse_dot;  // beta . X^T(m y) of the last solve
  double sse_nrm;  // |beta|^2 of the last solve
  unsigned long long snap_seq;  // sequence number of the last device snapshot of this block (deferred publication)
};
static_assert(sizeof(FitCtrl) <= 128, "the published control block is 128 bytes");

constexpr int GRAM_JC = 8;  // most tiles of one tile row handled by one wave of k_gram (runs of 8/4/2/1)
struct GramTask {
  int I, J0, nJ, pad_;
};

hipError_t launch_transpose_in(const double *src, int rows, int p, double *X, long ld, long r0, hipStream_t st);
hipError_t launch_normalize(double *X, long ld, int n, int p, double *y, const double *w, int data_type,
                            int is_normal, int add_weight, double *x_mean, double *x_norm, double *y_mean,
                            hipStream_t st);
hipError_t launch_xtv(const double *X, long ld, int p, int U, const double *v, const double *v2, double *part,
                      double *part2, const FitCtrl *ctrl, int slot, hipStream_t st);
hipError_t launch_xtv_variant(int variant, const double *X, long ld, int p, const double *v, double *part,
                              hipStream_t st);
// One pass over X for several chains' score vectors (k_xtv_mc): per chain the arguments of launch_xtv; ran (optional)
// receives the number of chains whose gate was open (0: the launch fell through).
constexpr int XTV_MC_MAX = 8;
struct XtvMc {
  const double *v[XTV_MC_MAX], *v2[XTV_MC_MAX];
  double *part[XTV_MC_MAX], *part2[XTV_MC_MAX];
  const FitCtrl *ctrl[XTV_MC_MAX];
  int slot[XTV_MC_MAX];
  int nc;
  int *ran;
};
hipError_t launch_xtv_mc(const double *X, long ld, int p, int U, const XtvMc &a, bool two, hipStream_t st);
hipError_t launch_score(const double *part, const double *part2, int nrb, int p, const double *beta_dense,
                        const double *xtx, double n_t, double lambda, int glm, const unsigned char *always,
                        double *bd, const FitCtrl *ctrl, int slot, hipStream_t st);
// What k_publish copies into the pinned result block; the last kernel of a batch of slots can do it itself (on = 1).
struct PubArgs {
  const unsigned char *dev;
  unsigned char *host;
  int ctrl_bytes;
  size_t off_sse;
  int n_sse;
  size_t off_b, off_a;
  int kcopy;
  unsigned long long *seq_host;
  unsigned long long seq;
  const int *count_ptr;
  int on;               // 1: publish now (copy dev -> host, release the sequence number);  2: only take a snapshot of
                        // the block into `snap` (device) -- the publication follows from there, off the critical path
  unsigned char *snap;  // device staging block (same offsets as dev; the cache's column count at snap_count_off)
  size_t snap_count_off;
};
// k_topk2 can end with the work of k_cov_need (covariance form of the LM fit) when the scores fit one chunk
struct TopkNeed {
  const double *bd;
  double *bd2;
  int p;
  int C;
  int *slot_of, *meta, *fcols;
  FitCtrl *ctrl;
  const int *A_cur;
  const double *bmm;  // per-block (min inside, max outside) scores left by k_cov_d, nbmm pairs
  int nbmm;
  const unsigned char *inA;  // membership flags of the current active set
  int inc1;                  // the scores are those on which the previous fit (one size smaller) ended: arg-max path
  int bmm_fresh;             // ... and bmm (block maxima + their columns) comes from the k_cov_d that made them
  // deferred publication of the PARENT fit's result block (pub.on = 1, pub.dev = its snapshot): done by a second
  // workgroup of this launch, concurrently with the selection -- the two system-scope round trips of a publication
  // are then not part of the chain's critical path
  PubArgs pub;
  // ... and this batch's own snapshot (snap.on = 2), taken right where a repeated active set is recorded, so that the
  // solve kernel behind this launch has nothing left to do
  PubArgs snap;
  // fused k_fit_continue(chained): this launch opens a fit chained behind fit `cont_parent` (cont_on = 1)
  int cont_on, cont_serial, cont_parent;
  // fused commit of a repeated active set (the record-and-stop branch of k_commit): commit_on = 1
  int commit_on;
  int no_restart;  // fold chains side by side: a full cache parks the fit (cov_stall = 4) instead of starting it over
  int *cm_A_cur;
  double *cm_b_cur, *cm_beta_dense;
  int *cm_hist;
  double *cm_hist_beta, *cm_hist_coef0;
  int cm_hist_stride;
  unsigned char *cm_inA;
};
bool topk_can_fuse_need(int len);
// exact std::nth_element semantics behind a tie at the selection boundary (k_topk_ties): flag = 2 ints (zeroed once:
// [0] raised by the selection kernels and cleared by k_topk_ties, [1] its heap-select branch was needed), work = 3 len ints
struct TopkTie {
  int *flag;
  int *work;
};
hipError_t launch_topk(const double *score, int len, int k, int *out, int *cand, const FitCtrl *ctrl, int slot,
                       hipStream_t st, const int *run_flag = nullptr, const TopkNeed *need = nullptr,
                       const TopkTie *tie = nullptr);
bool topk_supported(int len, int k);
void topk_set_variant(int v);  // test / benchmark hook: 0 = bit-by-bit search, 1 = radix search (default)
void gram_set_variant(int v);  // 1 = LDS-staged Gram kernel where it applies (default), 0 = k_gram throughout
bool gram_lds_applies(int ntiles, int tile_base);
hipError_t gram_lds_prepare();
// the IRLS step of the GLM fits in one pass over the active columns (k_irls_gram): up to 8 tile rows
bool irls_gram_applies(int mt);
int irls_gram_slab_rows(int mt, long ld);
hipError_t launch_irls_gram(int fam, const double *X, const double *aux, long ld, int n, const int *cols,
                            const double *y, const double *w, const double *mask, int nslab, int mt, double *part,
                            int ntiles, const FitCtrl *ctrl, int slot, int t, int T0, const double *bcur,
                            double *llpart, hipStream_t st, int wfloor = 1);
hipError_t launch_gram_reduce(const double *part, int nslab, int ntiles, double *Gt, const FitCtrl *ctrl, int slot,
                              int gate_mode, hipStream_t st);
hipError_t launch_gram(const double *X, const double *aux, long ld, const int *cols, const double *w,
                       int rows_per_slab, const GramTask *tasks, int ntask, int nslab, double *part, int ntiles,
                       double *Gt, const FitCtrl *ctrl, int slot, int gate_mode, hipStream_t st, int tile_base = 0);
hipError_t launch_gram_lm_cached(const double *X, const double *aux, long ld, int *cols, const double *w,
                                 const int *A_new, int T0, int mt, const GramTask *tasks_full, int ntask_full,
                                 int rps_full, int nslab_full, const GramTask *tasks_inc, int ntask_inc, int rps_inc,
                                 int nslab_inc, double *part, double *Gt, double *Rt, int *src, double *gbuf0,
                                 double *gbuf1, int *Ac, int *meta, FitCtrl *ctrl, int slot, hipStream_t st);
// k_chol in the covariance mode of the LM fit: Gram gathered from the column cache, k_commit's work at the end
struct CholFuse {
  const double *G;
  const int *slot_of;
  int p;
  int T0;
  FitCtrl *ctrl;
  int *A_cur;
  double *b_cur, *beta_dense;
  int *hist;
  double *hist_beta, *hist_coef0;
  int hist_stride;
  unsigned char *inA;  // membership flags of the active set (repeated-set shortcut of k_cov_d)
  double yy;           // y.(m y) of the row set (loss from the solved system, k_cg)
  const double *d;     // X^T (m r) of the current coefficients (start value of entering columns, k_cg)
  const double *GS;    // slot-indexed Gram of the cached columns (CS x CS), or nullptr (k_cgr gathers from it)
  int CS;
  const double *zero;  // a word that holds 0.0 (what k_cgr reads for matrix columns outside its system)
  PubArgs pub;         // pub.on: this launch closes a batch of slots and publishes the result block (k_cg / k_cgr)
  double *fb_work;     // k_chol: work space of the pivoted fallback solve (256 x 256 + 1024 doubles), or nullptr
  const int *dep;      // k_cg / k_cgr: != 0 when exactly dependent columns are cached for the row set (CovCache meta[4])
};
constexpr size_t CHOL_FB_DOUBLES = 256 * 256 + 1024;
// the IRLS convergence test at the head of k_chol (otherwise its own launch, k_glm_irls_check): on = 1
struct IrlsChk {
  int on;
  FitCtrl *ctrl;
  int t, fam;
  const double *llpart;  // log-likelihood terms of the iterate the Gram was formed at, nblk of them
  int nblk, m;           // m = coefficients incl. the intercept
  double *bcur, *bprev;
};
hipError_t launch_chol(const double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                       const int *rhs_gather, double *sol, int *info, const FitCtrl *ctrl, int slot, int gate_mode,
                       hipStream_t st, const CholFuse *fuse = nullptr, const IrlsChk *chk = nullptr);
// the pivoted solve behind a launch_chol whose kernel gave up (info = 2); fuse->fb_work is required
hipError_t launch_sym_fallback(const double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                               const int *rhs_gather, double *sol, int *info, const FitCtrl *ctrl, int slot,
                               hipStream_t st, const CholFuse *fuse);
// tol: accepted relative residual |q - (G + ridge I) x| <= tol |q|; by_rows: systems of up to 208 unknowns use the
// row-dealt kernel (k_cgr), otherwise / beyond the tile-dealt one (k_cg)
hipError_t launch_cg(int m, int mt, double ridge, const double *rhs, const int *A_new, double *sol, const FitCtrl *ctrl,
                     int slot, const CholFuse *fuse, int maxit, hipStream_t st, double tol = 1e-13, bool by_rows = true);
// the selection of a covariance-form slot and the solve behind it in one launch (k_sel_cgr)
bool sel_cgr_applies(int len, int m);
hipError_t launch_sel_cgr(const double *score, int len, int k, int *A_new, const FitCtrl *ctrl, int slot,
                          const TopkNeed *need, double ridge, const double *rhs, double *sol, const CholFuse *fuse,
                          int maxit, hipStream_t st, double tol);
// ---------------------------------------------------------------------------------------------------------------
// Merged launches over the chunk chains of one sequential path (round 5, bessx_kchunks.cpp: mc_run_chunks).  The C chains
// of the chunk phase used to run on a stream and a host thread each; the device runs about 2.5 single-workgroup kernels
// of different streams at a time (tools/probe/launch_rate.hip), so four chains got 2.4 x one chain's rate.  Here every
// chain is a workgroup (or a slice of the grid) of the SAME launch on ONE stream, and the sequencing the host did per
// chain -- which candidate, which PDAS iteration, is the fit over, open the next one -- lives in device memory:
//   k_mc_cov_d    (grid: column blocks x chains)  d and the sacrifice scores of every chain whose coefficients changed
//   k_mc_sel_cgr  (grid: chains)                  per chain: [record the finished candidate, open the next one,] the
//                                                 selection (repeated set / arg-max / full search + cache lookup), the
//                                                 solve, the commit -- the bodies of k_sel_cgr, gate by gate
// One pair of launches = one PDAS iteration WITH a solve for every chain; the confirming iteration of a candidate and the
// first selection of the next one ride in the same k_mc_sel_cgr.  The host queues pairs ahead, reads all chains' states
// back in one block (k_mc_status), serves parked chains with ONE union fill, and takes a chain over (the proven per-
// context path) where anything unusual turns up: a tie at the selection boundary, a solve that missed its target, a
// fit out of iterations, a loss that has to be recomputed.
// ---------------------------------------------------------------------------------------------------------------
struct McState {   // per chain, device memory: written by k_mc_sel_cgr / k_mc_resume only
  int cand;        // index within the chunk of the candidate being fitted
  int ncand;       // candidates of the chunk
  int need_d;      // the coefficients changed since the last score pass: k_mc_cov_d has work for this chain
  int finished;    // 1: every candidate recorded; 2: stopped, the host takes the chain over from candidate `cand`
  int parked;      // cov_stall of a parked fit (1 missing columns, 2 solve, 3 tie, 4 full cache); 0: running
  int resume;      // set with the wake-up after a fill: this slot's solve is still to run
  int prev_fresh;  // the previous candidate ended on a repeated set with fresh scores (arg-max start of the next one)
  int prev_T0;
  int solves;      // statistics: solves committed
  int why;         // finished == 2: 1 out of iterations, 2 info / cov_miss, 3 parked with a code the host does not serve
  int pad_[6];
};
static_assert(sizeof(McState) == 64, "McState is copied to the host in 64-byte records");
constexpr int MC_REC_I = 4;  // per candidate: T0, PDAS iterations, sse_valid, done
constexpr int MC_REC_D = 4;  // per candidate: coef0, sse_dot, sse_nrm, (unused)
struct McChain {   // one chain of a merged run: device memory, constant while the run lasts
  McState *state;
  const int *seq;  // the chunk's sparsity levels (device), state->ncand of them
  int width;       // row length of rec_A / rec_b (largest level of the path)
  int max_iter, p;
  // score pass from the cached Gram columns (the arguments of k_cov_d)
  const double *G, *xty, *xtx;
  double n_t, lambda;
  const unsigned char *always;
  double *d_out, *bd, *bmm;
  // selection and solve (the arguments of k_sel_cgr as enqueue_lm_slot_cov fills them; per candidate only the level and
  // the arg-max flags change)
  TopkNeed nd, nd1;  // nd1: the same with inc1 = bmm_fresh = 1 (arg-max start of a candidate one level up)
  CholFuse fz;       // (its T0 is not read: k_cgr's body takes the level from the size of the system)
  int *A_new;
  double *sol;
  double tol;
  int maxit;
  // per-candidate records
  int *rec_i;
  double *rec_d;
  int *rec_A;
  double *rec_b;
};
bool mc_applies(int p, int kmax);
hipError_t launch_mc_cov_d(const McChain *chains, int nchains, int p, hipStream_t st);
hipError_t launch_mc_sel_cgr(const McChain *chains, int nchains, int p, hipStream_t st);
// every chain's state and control block into pinned memory (192 bytes per chain), then the sequence number
hipError_t launch_mc_status(const McChain *chains, int nchains, unsigned char *host, unsigned long long *seq_host,
                            unsigned long long seq, hipStream_t st);
// after the fill that served it: wake the parked fit up (k_cov_resume) and mark its solve as due
hipError_t launch_mc_resume(const McChain *chains, int chain, hipStream_t st);
// stop a chain where it stands (the host takes it over)
hipError_t launch_mc_stop(const McChain *chains, int chain, hipStream_t st);

hipError_t launch_chol_big(double *Gt, int m, int mt, double ridge, int ridge_skip0, const double *rhs,
                           const int *rhs_gather, double *sol, int *info, double *rdiag, double *z,
                           const FitCtrl *ctrl, int slot, int gate_mode, hipStream_t st);
hipError_t launch_fit_begin(FitCtrl *ctrl, int T0, int k_init, const int *init_idx, const double *init_val,
                            double coef0_init, int *A_cur, double *b_cur, double *beta_dense, int p, int *hist,
                            hipStream_t st, unsigned char *inA = nullptr, int serial = 0);
hipError_t launch_fit_continue(FitCtrl *ctrl, int T0, int *hist, hipStream_t st, int serial = 0, int chained = 0,
                               int parent = 0);
hipError_t launch_commit(FitCtrl *ctrl, int slot, int T0, const int *A_new, const double *sol, int has_intercept,
                         int wait_chain, int *A_cur, double *b_cur, double *beta_dense, int *hist, double *hist_beta,
                         double *hist_coef0, int hist_stride, hipStream_t st, unsigned char *inA = nullptr);
hipError_t launch_resid_lm(const double *X, long ld, int n, const double *y, const double *mask,
                           const FitCtrl *ctrl, int when, const int *A_cur, const double *b_cur, double *r,
                           double *sse, hipStream_t st, int mode = 0, int kc_given = 0, double c0_given = 0.0);
hipError_t launch_dot(const double *a, const double *b, long n, double *out, hipStream_t st);
hipError_t launch_glm_eta_gh(int fam, const double *X, long ld, int n, const double *y, const double *w,
                             const double *mask, const double *logfact, const FitCtrl *ctrl, int when,
                             const int *A_cur, const double *b_cur, double *g, double *h, double *stats,
                             hipStream_t st);
hipError_t launch_glm_irls_begin(const FitCtrl *ctrl, int slot, int fam, int m, double *bcur, double *bprev,
                                 hipStream_t st);
hipError_t launch_glm_irls_prep(int fam, const double *X, long ld, int n, const double *y, const double *w,
                                const double *mask, const FitCtrl *ctrl, int slot, int t, const int *A_new, int T0,
                                const double *bcur, double *Wv, double *z, double *llpart, hipStream_t st,
                                int wfloor = 1);
hipError_t launch_glm_irls_check(FitCtrl *ctrl, int slot, int t, int fam, const double *llpart, int nblk, int m,
                                 double *bcur, double *bprev, hipStream_t st);
// Cox (src/Algorithm.h:1370-1650, src/coxph.cpp:16-40)
struct CoxBufs {
  double *E, *TH, *ET, *S0, *RS0, *SALL, *STEST;      // state pass: exp(eta), w*exp*mask, test-row exp, scans
  double *EW, *WD;                                     // w*[delta!=0]*mask (get_A), w*delta*mask (fit)
  double *ETA0, *THF, *S0F, *RS0F, *VG, *WG1, *UD, *TH1, *S1;  // Newton work vectors
  double *M;                                           // n x k matrix S1/S0 (ld x 256)
  double *g, *u, *b0;                                  // k-vectors: gradient, Newton direction, iterate
  double *Gt2;                                         // second Gram (M^T diag(w delta) M) in tile layout
  double *llpart;
  double *SCR;                                         // block totals of the multi-block scans
  double *C1, *CU, *CV, *C2;                           // one-pass score: prefix sums of ew/S0, u, ew - u, ew/S0^2
  int one_pass;                                        // score pass reads X once (k_cox_score1p)
  int need_uv;                                         // CU / CV wanted although the score is not one-pass (groups)
  double *ldl_work;                                    // 256 x 256: dense copy for the LDL^T fallback of the Newton solve
  // one-pass Hessian of the Newton step (k_cox_hess): weights of the second Gram, its slab partials, per slab the
  // column totals of theta x, the slab carries and the vectors q_b
  int hess_fused;
  double *CW, *HP2, *HT, *CAR, *HQ;
  double fit_clamp;  // clamp of the linear predictor in the Newton step: 30 (src/Algorithm.h:1417-1422); cox_fit of the
                     // screening uses 50 (src/coxph.cpp:65-71)
};
// One pass over X for several chains' one-pass Cox scores (k_cox_score1p_mc): per chain the four n-vectors of
// launch_cox_score_pass (one_pass form) and its output planes; ran (optional) receives the number of open gates.
constexpr int COX_MC_MAX = 6;
struct CoxMc {
  const double *TH[COX_MC_MAX], *CU[COX_MC_MAX], *CV[COX_MC_MAX], *C2[COX_MC_MAX];
  double *out[COX_MC_MAX];
  const FitCtrl *ctrl[COX_MC_MAX];
  int slot[COX_MC_MAX];
  int nc;
  int *ran;
};
hipError_t launch_cox_score1p_mc(const double *X, long ld, int p, int U, int nrb, const CoxMc &a, hipStream_t st);
void cox_score_set_variant(int v);  // bench hook: the wave -> (column group, row block) map of k_cox_score1p (1 = default)
int cox_hess_slab_rows(long ld);
bool cox_hess_applies(int mt);
hipError_t cox_hess_prepare();
size_t cox_scan_scratch_doubles(long ld, int kmax);
hipError_t launch_cox_state(const double *X, long ld, int n, const double *y, const double *w, const double *mask,
                            const FitCtrl *ctrl, int when, const int *A_cur, const double *b_cur, CoxBufs cb,
                            double *stats, hipStream_t st);
hipError_t launch_cox_score_pass(const double *X, long ld, int p, int U, int nrb, CoxBufs cb, double *part,
                                 double *part2, const FitCtrl *ctrl, int slot, hipStream_t st);
hipError_t launch_cox_score(const double *part, const double *part2, int nrb, int p, const double *beta_dense,
                            double lambda, const unsigned char *always, double *bd, const FitCtrl *ctrl, int slot,
                            hipStream_t st);
hipError_t launch_cox_newton_begin(FitCtrl *ctrl, int slot, int k, CoxBufs cb, int *idcols, hipStream_t st);
hipError_t launch_cox_newton_step(const double *X, const double *aux, long ld, int n, const double *mask,
                                  FitCtrl *ctrl, int slot, int t, const int *A_new, int k, double lambda,
                                  const int *gcols, const int *idcols, int mt, const GramTask *tasks, int ntask,
                                  int rps, int nslab, double *gpart, int ntiles, double *Gt, CoxBufs cb,
                                  hipStream_t st, double *rdiag, double *zbig);
hipError_t launch_group_moments(int smax, const double *X, long ld, int n, const double *w1, const double *w2, int N,
                                const int *gidx, const int *gsz, const int *goff, double *mblk, double *dcol,
                                hipStream_t st, int cshift = 0);
hipError_t launch_iota(int *a, int n, hipStream_t st);
// screening with groups, LM: score_g = |argmin_b |y - X_g b||^2 / size(g) from the group moments
hipError_t launch_group_lsq_score(int N, const int *gidx, const int *gsz, const int *goff, const double *mblk,
                                  const double *dcol, const unsigned char *always, double *work, double *zwork,
                                  double *score, hipStream_t st);
// screening with groups, logistic: per-group IRLS (groups of at most 8 columns)
bool screen_logit_group_supported(int gmax);
size_t screen_logit_group_state_doubles(int N);
hipError_t launch_screen_logit_group(const double *X, long ld, int n, int N, const int *gidx, const int *gsz,
                                     const double *y, const double *w, double *state, int *done,
                                     const unsigned char *always, double *score, hipStream_t st);
// screening with groups, Cox: per-group damped Newton (groups of at most 4 columns)
bool screen_cox_group_supported(int gmax);
hipError_t launch_screen_cox_group(const double *X, long ld, int n, int N, const int *gidx, const int *gsz,
                                   const double *st_, const double *w, const unsigned char *always, double *score,
                                   hipStream_t st);
hipError_t launch_cox_group_moments(const double *X, long ld, int n, int p, CoxBufs cb, const int *allcols, int mcols,
                                    int smax, int N, const int *gidx_h, const int *gsz_h, const int *gidx,
                                    const int *gsz, const int *goff, long mblk_len, double *mblk, double *mblk2,
                                    double *dcol, hipStream_t st);
hipError_t launch_group_score(int N, const int *gidx, const int *gsz, const int *goff, const double *mblk,
                              const double *dcol, const double *part, int nrb, int p, int lm, double n_t,
                              double lambda, const double *beta_dense, const unsigned char *always, double *bd,
                              hipStream_t st, int smax = 1, double *work = nullptr, double *zwork = nullptr,
                              const FitCtrl *ctrl = nullptr, int slot = 0, int eig_mode = 0, double *eig_v = nullptr,
                              double *eig_l = nullptr);
constexpr int GRP_EIG_MAX = 16;  // widest group of the register-resident score kernel (GRP_MAX of bessx_kdev.hpp)
// find_ind on the device for groups of one width (k_group_expand)
hipError_t launch_group_expand(const int *G_sel, int T0, int gs, const int *gidx, int *cols, const FitCtrl *ctrl,
                               int slot, hipStream_t st);
hipError_t launch_commit_group(FitCtrl *ctrl, int slot, int T0, const int *G_new, int K, const int *cols,
                               const double *sol, int has_intercept, int wait_chain, int *A_cur, double *b_cur,
                               double *beta_dense, int *hist, double *hist_beta, double *hist_coef0, int hist_stride,
                               hipStream_t st);
hipError_t launch_screen_score_lm(const double *sxy, const double *sxx, int p, const unsigned char *always,
                                  double *score, hipStream_t st);
hipError_t launch_screen_logit(const double *X, long ld, int n, int p, const double *y, const double *w,
                               double *state, int *done, const unsigned char *always, double *score, hipStream_t st);
hipError_t launch_screen_cox(const double *X, long ld, int n, int p, const double *st_, const double *w,
                             const unsigned char *always, double *score, hipStream_t st);
hipError_t launch_gather_cols(const double *X, long ld, const int *A, int pnew, double *X2, hipStream_t st);
// covariance-update mode (LM)
hipError_t launch_cov_need(const int *list, int len, const double *bd, double *bd2, int p, int *slot_of, int *meta,
                           int C, int *fcols, FitCtrl *ctrl, int slot, const int *A_cur, hipStream_t st,
                           int no_restart = 0);
// spec: the lookup formed the masked score copy bd2 and `extras` holds its best columns (0: only the missing columns)
hipError_t launch_cov_publish_slots(const int *fcols, const int *slot_w, int *slot_of, const FitCtrl *ctrl, hipStream_t st);
hipError_t launch_cov_fill_list(int *fcols, const int *extras, const double *bd2, int *slot_of, int *meta,
                                FitCtrl *ctrl, int parked, hipStream_t st, int spec_max, int spec, int spec_min = 0);
hipError_t launch_cov_resume(FitCtrl *ctrl, hipStream_t st);
// the row sets of a cross-validation that share their fills (one launch reduces / compacts for all of them)
struct CovRowSets {
  int nr;
  double *G[9], *GS[9];
  const double *xtx[9];
  int ex_lo[9], ex_hi[9];  // slabs of the fold-major copy this row set leaves out (its own fold's rows)
};
hipError_t launch_cov_reduce_compact_sets(const double *part, int p, const int *fcols, const int *slot_of, int *meta,
                                          const CovRowSets &rs, int g0, int ngroups, int nslab, int CS,
                                          const FitCtrl *ctrl, int parked, hipStream_t st);
// Conjugate gradients for the LM systems of the covariance form beyond the register-resident solvers (bessx_cgbig.hip):
// one launch per step, every workgroup multiplies its 8 rows of the dense k x k copy of the cached Gram entries.
constexpr int CGB_MAX_K = 4096;  // the search direction lives in LDS (32 KB), 16 vector elements per thread in registers
struct CgbState {
  double rr[2];   // |r|^2 of the last two steps (by step parity)
  double qq;      // |q|^2
  int done_step;  // step at which the recurrence residual reached its target (0x7fffffff: not yet)
  int steps;      // steps taken
};
struct CgbWork {
  double *x, *q, *r[2], *p[2], *ap[2], *part_pq[2], *part_qq;
  CgbState *st;
};
size_t cgb_work_doubles(int kcap);  // work space (the dense matrix first) for systems of up to kcap unknowns
hipError_t launch_cg_big(const double *G, int p, const int *slot_of, const int *meta, const int *A_new, int k,
                         double ridge, const double *xty, const double *beta_dense, double *work, int kcap, double *sol,
                         FitCtrl *ctrl, int slot, int nsteps, double tol, double yy, hipStream_t st);
// one fill for several parked fits that share a slot map (k_cov_fill_union)
struct CovUnion {
  int nf;
  const int *list[8];  // the column sets that have to be cached when the fill is done
  int len[8];
  int on_restart[8];   // 1: this set is already cached -- it is listed only if the cache is started over
};
// restart: 0 keep the cache, 1 start it over, 2 decide on the device -- start over iff the columns the due lists miss
// (counted per list: an upper bound) do not fit the C-column cache; fill_ctrl->cov_nmiss tells which it was
hipError_t launch_cov_fill_union(const CovUnion &u, int restart, const int *extras, const double *bd2, int spec_max,
                                 int spec_min, int *slot_of, int *meta, int p, int *fcols, FitCtrl *fill_ctrl,
                                 hipStream_t st, int C = 0);
int cov_streamed_tiles_per_wave();
hipError_t launch_cov_panel(const double *X, const double *aux, long ld, int p, const double *mask, const int *fcols,
                            int g0, int ngroups, int rows_per_slab, int nslab, double *part, const FitCtrl *ctrl,
                            int parked, hipStream_t st, int variant = 3);
hipError_t cov_panel_prepare();
hipError_t launch_cov_reduce(const double *part, int p, const int *fcols, const int *slot_of, double *G, int g0,
                             int ngroups, int nslab, const FitCtrl *ctrl, int parked, hipStream_t st,
                             int ex_lo = 0, int ex_hi = 0);
hipError_t launch_rows_permute(const double *X, long ld, int p, const int *perm, long ldp, double *Xp, hipStream_t st);
hipError_t launch_cov_compact(const double *G, int p, const int *slot_of, const int *fcols, int g0, int ngroups,
                              double *GS, int CS, const FitCtrl *ctrl, int parked, hipStream_t st,
                              const double *xtx = nullptr, int *meta = nullptr);
// background (speculative) fill on a second stream
hipError_t launch_cov_d(const double *G, int p, const int *slot_of, const double *xty, const int *A_cur,
                        const double *b_cur, double *d_out, const double *beta_dense, const double *xtx, double n_t,
                        double lambda, const unsigned char *always, double *bd, const unsigned char *inA, double *bmm,
                        const FitCtrl *ctrl, int slot, hipStream_t st);
hipError_t launch_cov_gram(const double *G, int p, const int *slot_of, const int *A_new, int T0, int mt, double *Gt,
                           int *meta, const FitCtrl *ctrl, int slot, hipStream_t st);
hipError_t launch_publish(const unsigned char *dev, unsigned char *host, int ctrl_bytes, size_t off_sse, int n_sse,
                          size_t off_b, size_t off_a, int kcopy, unsigned long long *seq_host, unsigned long long seq,
                          hipStream_t st, const int *count_ptr = nullptr);
hipError_t launch_vec_mul(const double *a, const double *b, long n, double *out, hipStream_t st);
hipError_t launch_part_sum(const double *part, int nrb, int p, double *out, hipStream_t st);
hipError_t launch_fill(double *a, long n, double v, hipStream_t st);
hipError_t launch_gram_cols(const int *A_new, int T0, int mp, int intercept, int rhs_col, int *cols, FitCtrl *ctrl,
                            int slot, const int *A_cur, int allow_skip, hipStream_t st);
hipError_t launch_copy(const double *src, double *dst, long n, hipStream_t st);

}  // namespace bessx
#endif
