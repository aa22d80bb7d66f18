/* oracle/bess_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's PDAS hot path (Mamba413/bess):
 * Algorithm::fit, the per-family get_A / primary_model_fit, the Metric classes and
 * sequential_path / gs_path.  It exists only to CHECK the HIP implementation:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Parity of this file itself is pinned against the compiled reference
 * (oracle/_ref/libbess_ref.so) and the committed golden vectors -- see
 * tests/test_oracle_vs_reference.py and tests/test_oracle_golden.py.
 */
#ifndef BESS_ORACLE_H
#define BESS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same meaning as the arguments of the reference's bessCpp (src/bess.h:20-33).
 * x is row-major n x p (what pywrap_bess receives, src/utilities.cpp:13-25).
 * cv_fold_id: fold index in [0,K) per row; required when is_cv != 0 (the reference's
 * own folds come from std::random_device, src/Metric.h:57-59, so they cannot be
 * reproduced -- both sides are fed the same folds instead).
 * Returns 0 on success, non-zero on invalid arguments. */
int bess_oracle_run(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                    int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                    int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                    int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                    const int *always_select, int always_len, double *beta_out, double *coef0_out,
                    double *train_loss_out, double *ic_out);

/* As above plus the Powell-path arguments of bessCpp (src/bess.cpp:174-180, pgs_path src/path.cpp:1138-1309):
 * path_type 3 = Powell search over (s, log lambda) with golden-section (powell_path 1) or sequential
 * (powell_path 2) line searches; lambda_out (may be NULL) receives the chosen lambda. */
int bess_oracle_run2(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                     int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                     int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                     int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                     double lambda_min, double lambda_max, int nlambda, int powell_path, const int *always_select,
                     int always_len, double *beta_out, double *coef0_out, double *train_loss_out, double *ic_out,
                     double *lambda_out);

/* As bess_oracle_run2 plus the group structure (Data::g_index, src/Data.h:59-67): g_index[g] = first column of
 * group g, NULL = singleton groups.  Sparsity levels and always_select then count / name GROUPS; the trace stores
 * the expanded column list of every iteration.  Cox takes the group branch of its get_A (src/Algorithm.h:1497-1568,
 * the explicit n x n Hessian: small n only) when algorithm_type is 2 or 3, like the reference. */
int bess_oracle_run3(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                     int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                     int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                     int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                     double lambda_min, double lambda_max, int nlambda, int powell_path, const int *g_index, int g_len,
                     const int *always_select, int always_len, double *beta_out, double *coef0_out,
                     double *train_loss_out, double *ic_out, double *lambda_out);

/* Trace of the last bess_oracle_run (same layout as oracle/ref_harness.cpp):
 * which: 0 geta_meta (int, 4 per get_A call: l, T0, train_n, offset into a_flat)
 *        1 a_flat (int)   2 beta_flat (double)   3 coef0_calls (double)
 *        4 loss_calls (double)   5 ic_calls (double) */
int bess_oracle_trace_size(int which);
void bess_oracle_trace_copy_int(int which, int *out);
void bess_oracle_trace_copy_double(int which, double *out);

/* screening(), src/screening.cpp:26-105, singleton groups, model_type 1 (LM), 2 (logistic), 4 (Cox): marginal fit
 * per column of the raw row-major x, the screening_size columns with the largest squared coefficient (ascending
 * indices) are written to screening_A.  Returns 1 for Poisson (undefined behaviour in the reference). */
int bess_oracle_screening(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                          int screening_size, const int *always_select, int always_len, int *screening_A);

/* Small building blocks exposed so that single HIP kernels can be checked in isolation. */

/* Data::normalize (src/Data.h:79-93, src/normalize.cpp:20-85) and, if add_weight, Data::add_weight (src/Data.h:70-77)
 * on a column-major n x p copy, in place; statistics as Data keeps them. */
int bess_oracle_normalize(double *x, int n, int p, double *y, const double *weight, int data_type, int is_normal,
                          int add_weight, double *x_mean, double *x_norm, double *y_mean);

/* max_k (src/utilities.cpp:179-188): indices of the k largest scores, ascending.
 * Ties are broken towards the lower index (the reference's nth_element leaves ties
 * implementation-defined). */
void bess_oracle_max_k(const double *score, int len, int k, int *out);
long bess_oracle_nth_heap_selects(void); /* times max_k took the heap-select branch of std::nth_element */
int bess_oracle_last_heap_select(double *scores, int cap, int *k); /* ... and the input of the last such call */

/* Solve the symmetric system A x = b (A is k x k, column-major, only the lower triangle
 * is read) by an un-pivoted LDL^T.  Returns 0, or 1 if a pivot is exactly zero. */
int bess_oracle_sym_solve(const double *a, int k, const double *b, double *x);

/* wall seconds of the last bess_oracle_run*: set-up (copy, normalise, group_XTX) and the path itself */
void bess_oracle_last_timing(double *setup_s, double *path_s);

#ifdef __cplusplus
}
#endif
#endif
