/* include/bessx.h -- C ABI of libbessx.so, the MI355X-native PDAS best-subset solver.
 *
 * This is the drop-in boundary for the hot path of Mamba413/bess (reference paths are
 * relative to /root/reference).  Plain pointers and sizes only; no C++/torch types.
 * Every entry point returns 0 (BESSX_OK) or an error code; bessx_last_error() gives the
 * message for the calling thread.  The library never falls back to a CPU path: if no
 * HIP device is usable every compute entry point fails with BESSX_ERR_HIP.
 *
 * Layers (each cites the reference interface it replaces):
 *   1. bessx_pywrap_bess      <- pywrap_bess, src/bess.h:35-51 (what SWIG binds, python/src/bess.i:17-30)
 *   2. bessx_session_*        <- bessCpp, src/bess.h:20-33: Data + Algorithm* + Metric* set-up
 *                                (src/bess.cpp:61-165) kept resident on the GPU, then
 *                                sequential_path / gs_path / pgs_path (src/path.h:22-44)
 *   3. bessx_session_fit      <- Algorithm::fit + the update_* setters, src/Algorithm.h:77-171
 *   4. bessx_op_*             <- single Eigen call sites of the hot loop;
 *                                exported so that every HIP kernel can be parity-tested alone.
 */
#ifndef BESSX_H
#define BESSX_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  BESSX_OK = 0,
  BESSX_ERR_ARG = 1,         /* invalid argument (the reference would crash or read out of bounds) */
  BESSX_ERR_HIP = 2,         /* HIP runtime / device failure, or no device */
  BESSX_ERR_UNSUPPORTED = 3, /* valid for the reference, not built here (see bessx_problem: screening with groups / Poisson) */
  BESSX_ERR_NUMERIC = 4      /* non-finite pivot in a k x k solve */
};

const char *bessx_last_error(void);

/* Device / build information: writes a NUL-terminated description (device name, CU count,
 * code-object arch) into buf.  Fails with BESSX_ERR_HIP when no GPU is visible. */
int bessx_device_info(char *buf, int buf_len);

/* ---------------------------------------------------------------------------------------
 * 1. Drop-in for pywrap_bess (src/bess.h:35-51, src/bess.cpp:218-281).
 *    Identical argument list (bool -> int).  x is row-major x_row * x_col.  Writes
 *    beta_out[0..x_col), *coef0_out, *train_loss_out, *ic_out exactly like the reference
 *    (src/bess.cpp:277-280); additionally fills the slots the reference leaves
 *    uninitialised: *nullloss_out (Data::get_nullloss, src/Data.h:120-130: |y|^2 / n of the
 *    normalised response for data_type 1, else 2 log 2 * sum(weight)),
 *    A_out[0..k) = selected support, *l_out = PDAS iterations of the selected candidate;
 *    aic/bic/gic_out are set to 0.
 *    Differences, all documented in INTEGRATION.md: is_cv draws folds from a fixed-seed
 *    generator instead of std::random_device; invalid codes return BESSX_ERR_ARG instead of
 *    dereferencing a null Algorithm* (src/bess.cpp:93-116).
 * ------------------------------------------------------------------------------------- */
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
                      double *gic_out, int gic_out_len, int *A_out, int A_out_len, int *l_out);

/* ---------------------------------------------------------------------------------------
 * 1b. Drop-in for bessCpp (src/bess.h:20-33, src/bess.cpp:37-214), the C++ entry the R package binds
 *     (R/src/RcppExports.cpp:10-48).  Same 30 arguments with Eigen objects unpacked into pointer + length
 *     (bool -> int); x is COLUMN-major n x p, as Eigen::MatrixXd and R matrices are.  The outputs are the named
 *     entries of the list the R build returns (src/path.cpp:116-123 sequential, :376-380 golden section,
 *     + screening_A, src/bess.cpp:207), flattened:
 *       sequential_path: candidate q = j * sequence_len + i for lambda j and size i -- beta_all = lambda_len blocks of
 *         p x sequence_len (column-major), coef0_all / train_loss_all = lambda_len blocks of sequence_len, ic_all =
 *         sequence_len x lambda_len column-major;  n_all = sequence_len * lambda_len.
 *       gs_path / pgs_path: candidate q = evaluation order (every new golden-section point, then every improvement
 *         of the final sweep; Powell: the best point of every line search and the final re-fit); n_all = count.
 *     All coefficients are de-normalised like the R build's (src/path.cpp:76-110, :330-373).  R/src/bess_amd_shim.cpp
 *     is the Rcpp wrapper around this function.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  double *beta;            /* p (caller-allocated) */
  double coef0, train_loss, ic, lambda;
  int all_capacity;        /* in: candidates the *_all arrays can hold (beta_all: p * all_capacity doubles) */
  int n_all;               /* out: candidates the path produced (written: min(n_all, all_capacity)) */
  double *beta_all, *coef0_all, *train_loss_all, *ic_all; /* caller-allocated, any may be NULL */
  int *screening_A;        /* screening_size entries (0-based original columns) when is_screening, may be NULL */
} bessx_r_result;

int bessx_bessCpp(const double *x, int n, int p, const double *y, int data_type, const double *weight, int is_normal,
                  int algorithm_type, int model_type, int max_iter, int exchange_num, int path_type,
                  int is_warm_start, int ic_type, int is_cv, int K, const double *state, int state_len,
                  const int *sequence, int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                  int K_max, double epsilon, double lambda_min, double lambda_max, int nlambda, int is_screening,
                  int screening_size, int powell_path, const int *g_index, int g_index_len, const int *always_select,
                  int always_select_len, double tao, bessx_r_result *res);

/* ---------------------------------------------------------------------------------------
 * 2. Session: the state bessCpp builds (src/bess.cpp:61-165), resident in HBM.
 * ------------------------------------------------------------------------------------- */
typedef struct bessx_session bessx_session;

typedef struct {
  int n, p;
  const double *x;   /* host pointer */
  int x_col_major;   /* 0: row-major n x p (NumPy / pywrap_bess); 1: column-major (R / bessCpp MatrixXd) */
  const double *y;   /* n; for Cox: status, rows already sorted by time (python/bess/linear.py:257-263) */
  const double *weight; /* n, or NULL for all ones */
  int data_type;     /* 1 centre x,y + scale; 2 centre x + scale; 3 scale only (src/Data.h:79-93) */
  int is_normal;
  int model_type;    /* 1 LM, 2 logistic, 3 Poisson, 4 Cox (src/bess.cpp:95-111) */
  int algorithm_type;/* 1 PDAS, 5 L0L2 (same code path, lambda from the path); 2/3 need groups of size 1 */
  int max_iter;      /* PDAS iterations per fit (Algorithm::max_iter) */
  int is_warm_start;
  const int *always_select; /* indices kept in every active set (Algorithm::always_select), may be NULL */
  int always_select_len;
  int device;        /* HIP device ordinal, or -1 for the current device */
  const int *group_index; /* Data::g_index (src/Data.h:59-67): first column of every group, ascending from 0; NULL or
                             length p = every column its own group.  With groups, sparsity levels and always_select
                             count / name GROUPS.  Any group width the session's capacity (max_sparsity) holds -- up to 16 columns
                             a register-resident block per group, wider ones a tiled path; Cox: at most 256 columns per
                             group.  Cox with real groups needs algorithm_type 2 / 3
                             (the group branch of GroupPdasCox::get_A, src/Algorithm.h:1497-1568). */
  int group_index_len;
  int is_screening;   /* sure independence screening before the path (screening(), src/screening.cpp:26-105; called at
                         src/bess.cpp:57-61): keep the screening_size columns with the largest squared marginal
                         coefficient on the raw data plus always_select.  LM, logistic and Cox; with groups of
                         size > 1 (any width) screening_size and always_select count / name GROUPS, the marginal fit
                         is the model's fit on the whole group, and the coefficients are written to the columns they
                         belong to -- the reference misplaces them in this case, src/bess.cpp:195-198.  Poisson is
                         refused (the reference's poisson_fit is undefined behaviour there, src/poisson.cpp:113), and so
                         is a logistic group at least as wide as the sample (logit_fit's n <= p branch returns n
                         coefficients of which screening() reads the last g_size: out of bounds).
                         The session then lives on the kept columns: sparsity levels, traces,
                         bessx_session_fit and bessx_session_get_normalization index them 0..screening_size-1
                         (bessx_session_get_screening gives the map); every bessx_path_result is written in the
                         ORIGINAL column numbering, like src/bess.cpp:186-209. */
  int screening_size;
  int score_mode;     /* how the LM score pass X^T r of get_A is evaluated: 0 = automatic
                         (environment variable BESSX_SCORE_MODE, else covariance updates when they apply),
                         1 = streaming: every PDAS iteration reads X once, 2 = covariance updates: cached Gram
                         columns X^T x_a, X is read only when a new column enters.  Same results either way up to
                         summation order. */
  int max_sparsity;   /* largest number of active COLUMNS any fit of this session will be asked for (sparsity level x
                         largest group size); sizes the k x k work space.  0 = default: min(p, 2046).  Up to 16382;
                         levels beyond 254 use the blocked Cholesky in global memory.  bessx_pywrap_bess derives it
                         from sequence / s_max, so the reference's default sequence 1..min(p, n / log n)
                         (python/bess/linear.py:285-287) runs as it is. */
} bessx_problem;

int bessx_session_create(bessx_session **out, const bessx_problem *prob);
void bessx_session_destroy(bessx_session *s);
/* screening_A of src/screening.cpp:68: original column of every kept column (ascending).  Returns the number of
 * kept columns (= p when the session was created without screening, map = identity); writes min(count, cap). */
int bessx_session_get_screening(const bessx_session *s, int *columns, int cap);
/* Screening with groups of size > 1: the kept ORIGINAL group numbers (ascending) = screening_A of the
 * reference for that case; 0 when the session was not screened by groups.  Returns the count, writes min(count, cap). */
int bessx_session_get_screening_groups(const bessx_session *s, int *groups, int cap);
/* 1 = streaming score pass, 2 = covariance updates: what bessx_problem.score_mode resolved to for this session. */
int bessx_session_score_mode(const bessx_session *s);
/* Diagnostics of the covariance form since the session was created: which = 0 fits that ran chained behind their
 * predecessor, 1 conjugate-gradient solves handed to the Cholesky kernel, 2 passes over X (32-column panel groups),
 * 3 chained fits queued; 4-6 always 0 (background fills and the maintained inverse of round 2: measured, removed);
 * 7 rounds in which the fold fits of cross-validation ran side by side (one fit context per fold), 8 fills that served
 * several parked folds at once, 9 PDAS iterations redone with the exact tie rule, 10 times the Gram column cache was
 * started over since the last path call started, 11 times bessx_session_set_cv had to give the per-fold fit contexts up
 * (allocation or launch failure: the fold fits then run one after another -- slower, same results), 12 fold contexts
 * alive now (K when the fold fits of a CV evaluation run side by side, else 0), 13 fills of a parked fit that went
 * through the fill hook (bessx_session_set_fill_hook), 14 sequential paths run as chunk chains side by side
 * (BESSX_KPATH_CHAINS), 15 candidates re-fitted by their stitching, 16 fills of the shared cache
 * during the chunk phase, 17 chains of the last such path, 18 paths whose stitch gave up (a chunk's refit did not meet its
 * own chain within its budget: the rest of the path was walked as one chain, and the automatic choice of this session is
 * one chain from then on), 19 device nanoseconds of the last all-rows group_XTX pass inside a path call (LM; kernel timing
 * on), 20 chunk phases
 * that ran as merged launches on one stream (round 5), 21 chains of such phases the host had to finish through the
 * per-context path (a tie at the selection boundary, a solve handed to the Cholesky kernel, ...), 22-24 microseconds the
 * chunked paths spent in their coarse chain / chunk phase / stitch (host clock, summed); round 6: 25 / 26 timed panel
 * launches of ONE 32-column group and their nanoseconds, 27 / 28 the same for launches of TWO groups (one read of X each;
 * reset with bessx_session_score_pass_stats), 29 multi-chain launches of the chunk chains' shared passes over X
 * (fall-through launches included), 30 vector sets those launches served (kernel timing on), 31 batches launched before
 * every chain had arrived (the 50 ms timeout), 32 streams with a hardware queue of their own the PROCESS has created so
 * far (they are recycled across sessions).  -1 for an unknown id. */
long long bessx_session_counter(const bessx_session *s, int which);

/* Metric::set_cv_train_test_mask + cal_cv_group_XTX (src/Metric.h:49-129).  fold_id[i] in [0,K)
 * gives the test fold of row i; fold_id == NULL draws a permutation from mt19937(seed) and cuts
 * it into K contiguous chunks exactly as src/Metric.h:66-78 does. */
int bessx_session_set_cv(bessx_session *s, int K, const int *fold_id, unsigned seed);
/* The test fold of every row as bessx_session_set_cv fixed it (given or drawn): fold_id[0..n).  Lets a caller hand
 * the very same folds to another implementation (the tests feed them to the oracle). */
int bessx_session_get_cv_folds(const bessx_session *s, int *fold_id);

/* Results of a path run.  All pointers are caller-allocated; any of the *_all pointers may be
 * NULL to skip that output.  Candidates are stored in evaluation order. */
typedef struct {
  double *beta;       /* p: best model, de-normalised (src/path.cpp:76-131, :330-388) */
  double coef0, train_loss, ic, lambda;
  int best_T0, best_iters;
  int capacity;       /* candidate slots available in the arrays below */
  int n_candidates;   /* out: candidates evaluated (may exceed capacity; extra ones are not stored) */
  int *cand_T0;       /* capacity */
  double *cand_lambda;/* capacity */
  int *cand_iters;    /* capacity: PDAS iterations (Algorithm::l) of the full-data fit */
  double *cand_train_loss, *cand_ic, *cand_coef0; /* capacity each; coef0 and beta are de-normalised
                                                     like beta_all / coef0_all of the R build */
  int *cand_support;  /* capacity * max_T0 (row per candidate, ascending, -1 padded) */
  double *cand_beta;  /* capacity * max_T0: coefficients matching cand_support */
  int max_T0;         /* row length of cand_support / cand_beta */
  double device_seconds; /* out: wall time of the path on the device side (host clock around the loop) */
  long long n_fits, n_pdas_iters; /* out: Algorithm::fit calls (incl. CV folds) and get_A calls */
} bessx_path_result;

/* sequential_path (src/path.cpp:25-132): sizes x lambdas in snake order, warm-start chain. */
int bessx_session_sequential_path(bessx_session *s, const int *sequence, int sequence_len,
                                  const double *lambda_seq, int lambda_len, int ic_type, int is_cv,
                                  bessx_path_result *res);
/* sequential_path as ONE LINK of a longer warm-start chain (src/path.cpp:60-64: candidate i starts from candidate
 * i-1's model) that several processes walk in pieces -- the k-path chunks of a multi-GPU run (bess_amd/dist.py,
 * StitchedKPath).  in: the model the first candidate starts from = Algorithm::update_beta_init / update_coef0_init
 * (src/Algorithm.h:85-93) with the predecessor's Algorithm::get_beta / get_coef0 (:97-111), NORMALISED scale, column
 * indices of the session; keep_caches != 0: the call continues the job of the previous path call on this session (the
 * cached Gram columns and score sums depend on the data only and stay), 0: it starts cold like every path call.
 * stop_support / stop_beta (optional): candidates the caller already holds for this very sequence, laid out like
 * cand_support / cand_beta of a bessx_path_result (row i = candidate i in evaluation order, stop_row_len entries, -1 / 0
 * padded, caller's column numbering, de-normalised coefficients).  The path stops after the first candidate i whose
 * support equals row i (and whose coefficients agree within stop_rtol when stop_beta is given): from there on the
 * chain the caller holds IS this chain -- same model, hence the same successors.  That candidate is stored;
 * stopped_at = i, or -1 when the whole sequence was walked.  The best model of the result is the best of the
 * candidates evaluated.  out: last_* = the model the NEXT candidate of the chain would start from (normalised;
 * last_len entries, min(last_len, last_cap) written).  Not offered under CV with an initial model (the folds' chains
 * would have to be handed over as well). */
typedef struct {
  const int *init_idx;
  const double *init_val;
  int init_len;
  double init_coef0;
  int keep_caches;
  const int *stop_support;
  const double *stop_beta; /* may be NULL: supports only */
  int stop_rows, stop_row_len;
  double stop_rtol;
  int stopped_at;          /* out */
  int *last_idx;           /* caller-allocated, last_cap entries; may be NULL */
  double *last_val;
  int last_cap;
  int last_len;            /* out */
  double last_coef0;       /* out */
  /* optional LEAD fits (round 6): sparsity levels (ascending, below the link's first) of a warm-start chain that is run
   * on this session IN FRONT of the link, from init_* -- every rank of a multi-GPU k-path walks the same coarse levels
   * the one-GPU path walks (a few fits whose fills bring nearly every Gram column the link will ask for) instead of
   * starting cold at its chunk: no communication, and the link starts from the last lead model.  Their candidates are
   * not returned; NULL / 0: none.  LM only. */
  const int *lead_levels;
  int lead_len;
} bessx_path_chain;
int bessx_session_sequential_path_chain(bessx_session *s, const int *sequence, int sequence_len,
                                        const double *lambda_seq, int lambda_len, int ic_type, int is_cv,
                                        bessx_path_chain *chain, bessx_path_result *res);

/* gs_path (src/path.cpp:134-389): integer golden section on [s_min, s_max] then exhaustive sweep. */
int bessx_session_gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res);

/* pgs_path (src/path.cpp:1138-1309): Powell search over (s, log lambda) for the L0L2 / bsrr types; line searches
 * by golden section (powell_path 1, n_lambda forced to 100) or on the lambda grid (powell_path 2).  lambda_min /
 * lambda_max are floored at 1e-5 as bessCpp does (src/bess.cpp:176-177).  Candidates = the best point of every
 * line search plus the final re-fit; res->lambda receives the chosen lambda. */
int bessx_session_pgs_path(bessx_session *s, int s_min, int s_max, double lambda_min, double lambda_max, int n_lambda,
                           int powell_path, int ic_type, int is_cv, bessx_path_result *res);

/* Optional trace of every PDAS iteration of every fit of the LAST path run, same layout as the
 * oracle's (oracle/bess_oracle.h): which = 0 geta_meta(int x4: l, T0, train_n, offset) 1 a_flat(int)
 * 2 beta_flat(double) 3 coef0_calls(double) 4 loss_calls(double) 5 ic_calls(double). */
int bessx_session_trace_enable(bessx_session *s, int on);
int bessx_session_trace_size(bessx_session *s, int which);
int bessx_session_trace_copy_int(bessx_session *s, int which, int *out);
int bessx_session_trace_copy_double(bessx_session *s, int which, double *out);

/* Normalisation results (Data::x_mean / x_norm / y_mean, src/Data.h:26-29). */
int bessx_session_get_normalization(bessx_session *s, double *x_mean, double *x_norm, double *y_mean);

/* Timing of the dominant kernel (the X^T r score pass, K1) accumulated since the last reset:
 * HIP events recorded on the session's stream around every launch.  Used by bench.py for the
 * roofline line.  seconds = sum of launch durations, launches = count, bytes = algorithmic
 * bytes (8 * n * p per launch; doubled accumulators read the same bytes). */
int bessx_session_score_pass_stats(bessx_session *s, int reset, double *seconds, long long *launches,
                                   double *algorithmic_bytes);
int bessx_session_enable_kernel_timing(bessx_session *s, int on);
/* Steps of the restricted fits' inner iterations taken since the last reset: IRLS solves of the logistic / Poisson
 * fits (src/Algorithm.h:1148-1204, 1273-1322), Newton steps of the Cox fit (:1377-1490); 0 for LM.  A statistic for
 * bench.py (time per step of the chain between two passes over X). */
int bessx_session_submodel_steps(bessx_session *s, int reset, long long *steps);

/* ---------------------------------------------------------------------------------------
 * 3. One Algorithm::fit (src/Algorithm.h:113-171) on the resident data.
 *    fold = -1: all rows; else the training rows of that CV fold (update_train_mask +
 *    update_group_XTX, src/Metric.h:182-183).  init_* is the warm start (update_beta_init /
 *    update_coef0_init) as a sparse vector of COLUMNS.  Outputs: support[W] ascending, beta[W] with
 *    W = bessx_session_fit_width(s, T0): T0 for singleton groups; with groups of size > 1 (T0 counts groups,
 *    Algorithm::fit returns the columns of the selected groups) the widest T0 groups' column count -- entries beyond
 *    the selected columns are -1 / 0.  coef0,
 *    iters (Algorithm::l), train_loss = the family's train_loss on ALL rows (src/Metric.h:145,266,
 *    426,565), test_loss = the family's CV test loss on the fold's test rows (0 when fold < 0).
 * ------------------------------------------------------------------------------------- */
/* Forget everything a previous fit left on the device beyond the data (score sums, cached Gram columns, fold
 * warm starts): the state a path call starts from (a bessCpp call starts from nothing).  A caller that builds its own
 * path out of bessx_session_fit (bess_amd/dist.py) calls this first, so that a repeated path does not reuse work. */
int bessx_session_reset_caches(bessx_session *s);
int bessx_session_fit_width(const bessx_session *s, int T0); /* -1: T0 outside [1, number of groups] */
int bessx_session_fit(bessx_session *s, int T0, double lambda, int fold, const int *init_idx,
                      const double *init_val, int init_len, double init_coef0, int *support, double *beta,
                      double *coef0, int *iters, double *train_loss, double *test_loss);

/* One evaluation of a cross-validated candidate restricted to SOME folds: Metric::test_loss (src/Metric.h:150-195) for
 * the folds listed (ascending), preceded -- want_full != 0 -- by the full-data Algorithm::fit the path runs before it
 * (src/path.cpp:58, :171).  What one rank of a fold-sharded CV path owns (bess_amd.dist.FoldShardedCV): the fold fits
 * take the library's own route (LM covariance form: the chains side by side on their own streams, one fill of the
 * shared Gram column caches for all parked folds), and start from the session's own per-fold warm starts
 * (cv_initial_model_param.row(k), :177-188; cleared by every path call and by bessx_session_reset_caches).  init_* /
 * init_coef0 = update_beta_init / update_coef0_init as the path sets them before the candidate (the full-data chain's
 * previous model; the fold fits read coef0_init, and beta_init when the session was created without warm start).
 * Outputs: want_full + n_folds records in the order [full,] folds[0], folds[1], ...; record r = support / beta
 * [r * W .. (r+1) * W) with W = bessx_session_fit_width(s, T0), coef0[r], iters[r], train_loss[r], test_loss[r]
 * (0 for the full-data fit) -- the fields of bessx_session_fit. */
int bessx_session_cv_eval(bessx_session *s, int T0, double lambda, int want_full, const int *init_idx,
                          const double *init_val, int init_len, double init_coef0, const int *folds, int n_folds,
                          int *support, double *beta, double *coef0, int *iters, double *train_loss, double *test_loss);

/* Cooperative prefill of the Gram column cache (LM, covariance form of the score pass, all rows; not on sessions with
 * CV folds): several sessions that hold the SAME data -- the ranks of a multi-GPU k-path -- share the passes over X that
 * each of their cold starts would repeat.  Every rank lists the same columns (begin: the cache is started over and slot
 * i goes to cols[i], so slot numbers agree across ranks), forms the Gram columns X^T x_c of ITS groups of 32 (compute),
 * hands the p x 32 blocks out (export; group g = 32 * p doubles, column after column) and takes the others' in (import),
 * then fills the slot-indexed Gram between cached columns once (end).  A path call that follows with
 * bessx_path_chain.keep_caches finds the columns cached.  Cache contents only -- no result depends on it; the blocks
 * are bit-identical to what the rank would have formed itself.  bessx_session_marginal_scores: the sacrifice scores of
 * get_A at beta = 0 (src/Algorithm.h:1109-1123), what the first PDAS iteration of a cold fit ranks -- the list is
 * their top M.  *_on_device: the buffer is device memory of this session's device (else host memory). */
int bessx_session_marginal_scores(bessx_session *s, double *bd /* p */);
int bessx_session_cov_prefill_begin(bessx_session *s, const int *cols, int ncols /* multiple of 32, distinct */);
/* A second list on top of the cache as it is (after a prefill, after fits): the columns -- none of them cached yet --
 * take the next free slots in list order; identical on every rank whose session has done identical work so far. */
int bessx_session_cov_prefill_extend(bessx_session *s, const int *cols, int ncols);
/* bd[p]: the sacrifice scores the last fit's last PDAS iteration ranked; slot_of[p]: cache slot of every column, -1 =
 * not cached.  Either may be NULL.  What a caller needs to choose the next columns worth caching. */
int bessx_session_cov_state(bessx_session *s, double *bd, int *slot_of);
int bessx_session_cov_prefill_compute(bessx_session *s, int g0, int ngroups);
int bessx_session_cov_prefill_export(bessx_session *s, int g0, int ngroups, double *dst, int dst_on_device);
int bessx_session_cov_prefill_import(bessx_session *s, int g0, int ngroups, const double *src, int src_on_device);
int bessx_session_cov_prefill_end(bessx_session *s);

/* How many chunk chains bessx_session_sequential_path may run side by side (bessx_kchunks.cpp):
 * 0 = automatic, 1 = one chain, 2..8 = that many where the path qualifies.  The initial value is BESSX_KPATH_CHAINS (else
 * 0).  The candidates returned are the same for every setting. */
int bessx_session_set_kpath_chains(bessx_session *s, int chains);

/* Shared WIDE fills inside a fit that several sessions run identically (the pilot fit of a multi-GPU k-path: same data,
 * same cache, deterministic kernels -- every rank's fit parks at the same PDAS iteration on the same missing columns).
 * With a hook set, a fit of the all-rows row set that parks on missing Gram columns does not form them privately (one
 * pass over X for 32-64 columns, repeated on every rank): the library lists the missing columns followed by the
 * uncached columns the CURRENT sacrifice scores rank highest, `width` columns in all (a multiple of 32; the same list on
 * every rank), hands them slots as bessx_session_cov_prefill_extend does, and calls hook(user, n_groups) -- the caller
 * forms its share of the groups, exchanges the blocks and closes the list (cov_prefill_compute / export / import /
 * end); the fit then goes on.  A non-zero return fails the fit with BESSX_ERR_ARG.  hook = NULL: private fills. */
typedef int (*bessx_fill_hook)(void *user, int n_groups);
int bessx_session_set_fill_hook(bessx_session *s, bessx_fill_hook hook, void *user, int width);

/* Test hook: queue a host function on the session's stream that sleeps for `milliseconds` (negative: that many
 * MICROseconds) -- everything queued behind it waits, as behind a wedged kernel (tests/test_deadline_gpu.py: the waits of the host give up at
 * BESSX_WAIT_TIMEOUT_S instead of spinning for ever). */
int bessx_session_debug_block_stream(bessx_session *s, int milliseconds);

/* ---------------------------------------------------------------------------------------
 * 4. Single-kernel entry points (host buffers in, host buffers out; each call uploads,
 *    launches on the current device and synchronises).  They exist for parity tests.
 * ------------------------------------------------------------------------------------- */
/* K1: out[j] = sum_i x[i,j] * v[i]  (X^T v; src/Algorithm.h:1109,1236,1341).  x column-major, ld >= n.
 * If v2 != NULL also out2[j] = sum_i x[i,j]^2 * v2[i] (K2; src/Algorithm.h:1240-1246). */
int bessx_op_xtv(const double *x, int n, int p, int ld, const double *v, const double *v2, double *out,
                 double *out2);
/* K1 / K2 for nc <= 8 vectors in ONE pass over X (the multi-chain score pass of chunk chains, k_xtv_mc): v, v2 are
 * nc x n (row c = chain c), out, out2 nc x p.  Bitwise the sums of nc bessx_op_xtv calls. */
int bessx_op_xtv_multi(const double *x, int n, int p, int ld, const double *v, const double *v2, int nc, double *out,
                       double *out2);
/* K4: max_k (src/utilities.cpp:179-188): the k largest scores, indices ascending, ties -> lower index. */
int bessx_op_topk(const double *score, int len, int k, int *out_idx);
/* timing of the top-k kernel (k_topk) on synthetic chi-square scores; `variant` is reserved (one kernel exists) */
int bessx_op_topk_bench(int len, int k, int variant, int repeats, double *avg_us);
/* timing of the register-resident Cholesky solve (k_chol) for an m x m system, 1 <= m <= 254 */
int bessx_op_chol_bench(int m, int repeats, double *avg_us);
/* K6: out (m x m, column-major, full symmetric) = X_A^T diag(w) X_A for the m columns cols[] of x;
 * w may be NULL (src/Algorithm.h:1134,1171,1199,1299). */
int bessx_op_gram(const double *x, int n, int p, int ld, const int *cols, int m, const double *w, double *out);
/* K7: solve the SPD system a * sol = b (a m x m column-major), Cholesky in one workgroup
 * (stands in for ColPivHouseholderQR / LDLT solves, src/Algorithm.h:1134,1171,1199,1299,1473). */
int bessx_op_chol_solve(const double *a, int m, const double *b, double *sol);
/* K11: Normalize / Normalize3 / Normalize4 + add_weight (src/normalize.cpp:20-85, src/Data.h:70-77)
 * on a column-major copy of x; returns the transformed x, y and the statistics. */
int bessx_op_normalize(double *x, int n, int p, double *y, const double *weight, int data_type, int is_normal,
                       int add_weight, double *x_mean, double *x_norm, double *y_mean);
/* Tuning aid for K1: run geometry variant `variant` of the score pass `repeats` times on an n x p matrix
 * generated on the device and report algorithmic GB/s (8*n*p bytes per launch) and the mean launch time. */
int bessx_op_xtv_bench(int n, int p, int variant, int repeats, double *gbps, double *avg_ms);
/* ... and the multi-chain score pass: nc vector sets per launch (two != 0: with the second accumulator); GB/s counts the
 * 8*n*p bytes of X once per launch. */
int bessx_op_xtv_multi_bench(int n, int p, int nc, int two, int repeats, double *gbps, double *avg_ms);
/* The same for the one-pass Cox score kernel (k_cox_score1p, 8*n*p bytes per launch); variant 1 = the one the solver runs
 * (wave map of round 4), 0 = round 3; 10 + nc (nc = 1..4) = the multi-chain kernel k_cox_score1p_mc with nc vector sets. */
int bessx_op_cox_score_bench(int n, int p, int variant, int repeats, double *gbps, double *avg_ms);
/* Device-to-device streaming copy rate in GB/s (read+write bytes / time): the measured HBM ceiling
 * quoted next to the spec peak in bench.py. */
int bessx_op_stream_copy_gbps(long long bytes, int repeats, double *gbps);

/* ---------------------------------------------------------------------------------------
 * 5. A communicator for hosts without torch.distributed (round 6): the ONE collective the sharded paths need -- an
 *    all-gather of small fp64 records (the IC / CV curve, 8 B per candidate; the k-chunks' last models; the fold fits'
 *    records: SURVEY 8e) -- on RCCL directly.  One process per GPU; rank 0 calls bessx_comm_unique_id and the host hands
 *    the 128 bytes to the other ranks by whatever it has (a file, MPI, a socket); every rank then calls bessx_comm_init
 *    (collective: returns when all `world` ranks have called it).  RCCL is loaded when the first of these functions is
 *    called (BESSX_ERR_UNSUPPORTED when it cannot be).  bess_amd.dist.BessxComm is the Python face of it.
 * ------------------------------------------------------------------------------------- */
#define BESSX_COMM_ID_BYTES 128
typedef struct bessx_comm bessx_comm;
int bessx_comm_unique_id(unsigned char *id /* BESSX_COMM_ID_BYTES */);
int bessx_comm_init(bessx_comm **out, int rank, int world, const unsigned char *id, int device);
int bessx_comm_rank(const bessx_comm *c);
int bessx_comm_world(const bessx_comm *c);
/* every rank's `count` doubles to every rank: recv[r * count .. (r + 1) * count) = rank r's send; host buffers */
int bessx_comm_allgather_f64(bessx_comm *c, const double *send, int count, double *recv);
void bessx_comm_destroy(bessx_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* BESSX_H */
