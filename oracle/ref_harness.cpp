// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A tracing driver around the *unmodified* reference sources that live under
// /root/reference/src.  It is compiled together with those sources (where they
// lie; nothing is copied) by oracle/Makefile into oracle/_ref/libbess_ref.so.
//
// What it adds on top of the reference:
//   * extern "C" entry points with un-mangled names (bess_ref_pywrap forwards
//     verbatim to the reference's pywrap_bess, src/bess.cpp:218-281);
//   * bess_ref_trace(): repeats the ~40 set-up lines of bessCpp
//     (src/bess.cpp:61-165) but instantiates *subclasses* of the reference's
//     GroupPdas* / *Metric classes whose virtual get_A / primary_model_fit /
//     train_loss / ic first call the reference implementation and then record
//     what it returned.  The path functions that are then run are the
//     reference's own sequential_path / gs_path (src/path.cpp:25-389).
//     This yields the active set of EVERY PDAS iteration of EVERY fit (full
//     data and CV folds), which the non-R build of the reference otherwise
//     throws away (it only returns the best model);
//   * deterministic CV folds: the reference draws folds from
//     std::random_device (src/Metric.h:57-59); the harness fills the public
//     members train_mask_list / test_mask_list / group_XTX_list
//     (src/Metric.h:23-26) from a caller-supplied fold id per row instead.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the resulting library.
#include <Eigen/Eigen>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "List.h"
#include "Data.h"
#include "Algorithm.h"
#include "Metric.h"
#include "path.h"
#include "utilities.h"
#include "bess.h"
#include "screening.h"

namespace {

struct Trace {
  // one record per get_A call: {l, T0, train_n, offset into a_flat}
  std::vector<int> geta_meta;
  std::vector<int> a_flat;
  // one record per primary_model_fit call: beta_A (same length as the matching A) + coef0
  std::vector<double> beta_flat;
  std::vector<double> coef0_calls;
  // top-level metric calls made by the path function
  std::vector<double> loss_calls;
  std::vector<double> ic_calls;
  int metric_depth = 0;
  void clear() { *this = Trace(); }
};

Trace g_trace;

// Long full-size runs (tests/golden/make_fullsize_ref.py): BESS_REF_PROGRESS=1 prints one line per fit;
// BESS_REF_BUDGET_S=<seconds> ends the run at the first fit that would START after the budget (thrown from
// get_A at l == 1, caught in bess_ref_trace2, which then returns 2): the trace holds whole fits only and the
// caller labels the golden file as a prefix of the path.
struct BudgetExceeded {};
std::chrono::steady_clock::time_point g_start;
double g_budget_s = 0.0;
int g_progress = 0, g_fit_count = 0;
double elapsed_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_start).count(); }

template <class Base>
struct TracedAlgorithm : public Base {
  TracedAlgorithm(Data &data, int algorithm_type, unsigned int max_iter) : Base(data, algorithm_type, max_iter) {}

  void get_A(Eigen::MatrixXd X, Eigen::VectorXd y, Eigen::VectorXd beta, double coef0, int T0, Eigen::VectorXd weights,
             Eigen::VectorXi index, Eigen::VectorXi gsize, int N, Eigen::VectorXi &A_out) override {
    int train_n = (int)X.rows();
    if (this->l == 1) {
      if (g_budget_s > 0.0 && elapsed_s() > g_budget_s) throw BudgetExceeded();
      g_fit_count++;
      if (g_progress) {
        std::fprintf(stderr, "[ref] fit %d T0=%d train_n=%d t=%.0fs\n", g_fit_count, T0, train_n, elapsed_s());
        std::fflush(stderr);
      }
    }
    Base::get_A(X, y, beta, coef0, T0, weights, index, gsize, N, A_out);
    g_trace.geta_meta.push_back(this->l);
    g_trace.geta_meta.push_back(T0);
    g_trace.geta_meta.push_back(train_n);
    g_trace.geta_meta.push_back((int)g_trace.a_flat.size());
    // store the EXPANDED column list (find_ind, src/utilities.cpp:113-130), which lines up with the coefficient
    // vector primary_model_fit returns; for singleton groups this is A_out itself
    Eigen::VectorXi ind = find_ind(A_out, index, gsize, (int)X.cols(), N);
    for (int i = 0; i < ind.size(); i++) g_trace.a_flat.push_back(ind(i));
  }

  void primary_model_fit(Eigen::MatrixXd X, Eigen::VectorXd y, Eigen::VectorXd weights, Eigen::VectorXd &beta,
                         double &coef0) override {
    Base::primary_model_fit(X, y, weights, beta, coef0);
    for (int i = 0; i < beta.size(); i++) g_trace.beta_flat.push_back(beta(i));
    g_trace.coef0_calls.push_back(coef0);
  }
};

template <class Base>
struct TracedMetric : public Base {
  TracedMetric(int ic_type, bool is_cv, int K) : Base(ic_type, is_cv, K) {}

  double train_loss(Algorithm *algorithm, Data &data) override {
    g_trace.metric_depth++;
    double v = Base::train_loss(algorithm, data);
    g_trace.metric_depth--;
    if (g_trace.metric_depth == 0) g_trace.loss_calls.push_back(v);
    return v;
  }

  double ic(Algorithm *algorithm, Data &data) override {
    g_trace.metric_depth++;
    double v = Base::ic(algorithm, data);
    g_trace.metric_depth--;
    if (g_trace.metric_depth == 0) g_trace.ic_calls.push_back(v);
    return v;
  }
};

}  // namespace

extern "C" {

// Verbatim forwarder to the reference's Python-facing C++ entry (src/bess.h:35-51).
void bess_ref_pywrap(double *x, int x_row, int x_col, double *y, int y_len, int data_type, double *weight,
                     int weight_len, int is_normal, int algorithm_type, int model_type, int max_iter, int exchange_num,
                     int path_type, int is_warm_start, int ic_type, int is_cv, int K, int *gindex, int gindex_len,
                     double *state, int state_len, int *sequence, int sequence_len, double *lambda_sequence,
                     int lambda_sequence_len, int s_min, int s_max, int K_max, double epsilon, double lambda_min,
                     double lambda_max, int n_lambda, int is_screening, int screening_size, int powell_path,
                     int *always_select, int always_select_len, double tao, double *beta_out, int beta_out_len,
                     double *coef0_out, int coef0_out_len, double *train_loss_out, int train_loss_out_len,
                     double *ic_out, int ic_out_len, double *nullloss_out, double *aic_out, int aic_out_len,
                     double *bic_out, int bic_out_len, double *gic_out, int gic_out_len, int *A_out, int A_out_len,
                     int *l_out) {
  pywrap_bess(x, x_row, x_col, y, y_len, data_type, weight, weight_len, is_normal != 0, algorithm_type, model_type,
              max_iter, exchange_num, path_type, is_warm_start != 0, ic_type, is_cv != 0, K, gindex, gindex_len, state,
              state_len, sequence, sequence_len, lambda_sequence, lambda_sequence_len, s_min, s_max, K_max, epsilon,
              lambda_min, lambda_max, n_lambda, is_screening != 0, screening_size, powell_path, always_select,
              always_select_len, tao, beta_out, beta_out_len, coef0_out, coef0_out_len, train_loss_out,
              train_loss_out_len, ic_out, ic_out_len, nullloss_out, aic_out, aic_out_len, bic_out, bic_out_len,
              gic_out, gic_out_len, A_out, A_out_len, l_out);
}

// Run one reference path with tracing.  x is row-major n x p (as pywrap_bess takes it).
// cv_fold_id: NULL -> the reference's own (random) folds; else fold index in [0,K) per row.
// Returns 0 on success.  Best-model outputs are what the reference's path function returns.
int bess_ref_trace2(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                    int is_normal, int algorithm_type, int model_type, int max_iter, int path_type, int is_warm_start,
                    int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence, int sequence_len,
                    const double *lambda_seq, int lambda_len, int s_min, int s_max, double lambda_min,
                    double lambda_max, int nlambda, int powell_path, const int *g_index, int g_len,
                    const int *always_select, int always_len, double *beta_out, double *coef0_out,
                    double *train_loss_out, double *ic_out, double *lambda_out);

int bess_ref_trace(const double *x, int n, int p, const double *y, const double *weight, int data_type, int is_normal,
                   int algorithm_type, int model_type, int max_iter, int path_type, int is_warm_start, int ic_type,
                   int is_cv, int K, const int *cv_fold_id, const int *sequence, int sequence_len,
                   const double *lambda_seq, int lambda_len, int s_min, int s_max, const int *g_index, int g_len,
                   const int *always_select, int always_len, double *beta_out, double *coef0_out,
                   double *train_loss_out, double *ic_out) {
  return bess_ref_trace2(x, n, p, y, weight, data_type, is_normal, algorithm_type, model_type, max_iter, path_type,
                         is_warm_start, ic_type, is_cv, K, cv_fold_id, sequence, sequence_len, lambda_seq, lambda_len,
                         s_min, s_max, 0.0, 0.0, 100, 1, g_index, g_len, always_select, always_len, beta_out,
                         coef0_out, train_loss_out, ic_out, nullptr);
}

// path_type 3: the reference's Powell path pgs_path (src/path.cpp:1138-1309), called exactly as bessCpp does
// (src/bess.cpp:174-180: log(max(lambda, 1e-5))).
int bess_ref_trace2(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                    int is_normal, int algorithm_type, int model_type, int max_iter, int path_type, int is_warm_start,
                    int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence, int sequence_len,
                    const double *lambda_seq, int lambda_len, int s_min, int s_max, double lambda_min,
                    double lambda_max, int nlambda, int powell_path, const int *g_index, int g_len,
                    const int *always_select, int always_len, double *beta_out, double *coef0_out,
                    double *train_loss_out, double *ic_out, double *lambda_out) {
  g_trace.clear();
  g_start = std::chrono::steady_clock::now();
  g_fit_count = 0;
  g_progress = std::getenv("BESS_REF_PROGRESS") != nullptr;
  g_budget_s = std::getenv("BESS_REF_BUDGET_S") ? std::atof(std::getenv("BESS_REF_BUDGET_S")) : 0.0;
  Eigen::MatrixXd X(n, p);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) X(i, j) = x[(size_t)i * p + j];
  Eigen::VectorXd Y(n), W(n);
  for (int i = 0; i < n; i++) {
    Y(i) = y[i];
    W(i) = weight[i];
  }
  Eigen::VectorXi G(g_len), AS(always_len), SEQ(sequence_len);
  for (int i = 0; i < g_len; i++) G(i) = g_index[i];
  for (int i = 0; i < always_len; i++) AS(i) = always_select[i];
  for (int i = 0; i < sequence_len; i++) SEQ(i) = sequence[i];
  Eigen::VectorXd LAM(lambda_len);
  for (int i = 0; i < lambda_len; i++) LAM(i) = lambda_seq[i];

  // --- the set-up of bessCpp, src/bess.cpp:61-165, with traced subclasses ---
  Data data(X, Y, data_type, W, is_normal != 0, G);
  Algorithm *algorithm = nullptr;
  if (model_type == 1) {
    data.add_weight();
    algorithm = new TracedAlgorithm<GroupPdasLm>(data, algorithm_type, max_iter);
  } else if (model_type == 2) {
    algorithm = new TracedAlgorithm<GroupPdasLogistic>(data, algorithm_type, max_iter);
  } else if (model_type == 3) {
    algorithm = new TracedAlgorithm<GroupPdasPoisson>(data, algorithm_type, max_iter);
  } else {
    algorithm = new TracedAlgorithm<GroupPdasCox>(data, algorithm_type, max_iter);
  }
  algorithm->set_warm_start(is_warm_start != 0);
  algorithm->always_select = AS;
  algorithm->tao = 0.;

  Metric *metric = nullptr;
  if (model_type == 1)
    metric = new TracedMetric<LmMetric>(ic_type, is_cv != 0, K);
  else if (model_type == 2)
    metric = new TracedMetric<LogisticMetric>(ic_type, is_cv != 0, K);
  else if (model_type == 3)
    metric = new TracedMetric<PoissonMetric>(ic_type, is_cv != 0, K);
  else
    metric = new TracedMetric<CoxMetric>(ic_type, is_cv != 0, K);

  if (is_cv) {
    if (cv_fold_id == nullptr) {
      metric->set_cv_train_test_mask(data.get_n());
    } else {
      // same shape as Metric::set_cv_train_test_mask produces (src/Metric.h:66-105):
      // sorted test rows per fold, sorted complement as train rows.
      std::vector<Eigen::VectorXi> train_list(K), test_list(K);
      for (int k = 0; k < K; k++) {
        std::vector<int> tr, te;
        for (int i = 0; i < n; i++) (cv_fold_id[i] == k ? te : tr).push_back(i);
        train_list[k] = Eigen::Map<Eigen::VectorXi>(tr.data(), tr.size());
        test_list[k] = Eigen::Map<Eigen::VectorXi>(te.data(), te.size());
      }
      metric->train_mask_list = train_list;
      metric->test_mask_list = test_list;
    }
    metric->set_cv_initial_model_param(K, data.get_p());
    if (model_type == 1) metric->cal_cv_group_XTX(data);
  }

  List result;
  try {
  if (path_type == 1) {
    result = sequential_path(data, algorithm, metric, SEQ, LAM);
  } else if (path_type == 3) {
    double log_lambda_min = log(max(lambda_min, 1e-5));
    double log_lambda_max = log(max(lambda_max, 1e-5));
    result = pgs_path(data, algorithm, metric, s_min, s_max, log_lambda_min, log_lambda_max, powell_path, nlambda);
    if (lambda_out) {
      double lam = 0.0;
      result.get_value_by_name("lambda", lam);
      *lambda_out = lam;
    }
  } else {
    result = gs_path(data, algorithm, metric, s_min, s_max, 0, 0.);
  }
  } catch (const BudgetExceeded &) {
    // whole fits recorded so far stay readable through bess_ref_trace_size/copy; no best model
    g_trace.metric_depth = 0;
    delete algorithm;
    delete metric;
    return 2;
  }

  Eigen::VectorXd beta;
  double coef0, train_loss, ic;
  result.get_value_by_name("beta", beta);
  result.get_value_by_name("coef0", coef0);
  result.get_value_by_name("train_loss", train_loss);
  result.get_value_by_name("ic", ic);
  for (int j = 0; j < p; j++) beta_out[j] = beta(j);
  *coef0_out = coef0;
  *train_loss_out = train_loss;
  *ic_out = ic;
  delete algorithm;
  delete metric;
  return 0;
}

// CPU-baseline timing (bench.py's cpu_baseline leg): one warm-start chain of the reference's own Algorithm / Metric
// objects, driven through their public setters in the order sequential_path uses them (src/path.cpp:48-74:
// update_train_mask, update_sparsity_level, update_lambda_level, update_beta_init, update_coef0_init,
// update_group_XTX, fit(), then train_loss() and ic()), but STARTING from a caller-supplied model (init_idx /
// init_val in the normalised scale Algorithm::beta lives in) so that a segment from the far end of a path (k = 181..)
// can be timed without paying for the 180 candidates before it.  cand_seconds[i] = wall time of candidate i (fit +
// train_loss + ic); stops after the first candidate that ends beyond budget_s.  Set-up (copy + normalise) is
// reported separately, like the GPU figure excludes upload + normalise.
int bess_ref_time_chain(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                        int is_normal, int algorithm_type, int model_type, int max_iter, int ic_type,
                        const int *sequence, int sequence_len, const int *init_idx, const double *init_val,
                        int init_len, double init_coef0, double budget_s, double *cand_seconds, int *cand_iters,
                        int *n_done, double *setup_seconds) {
  auto t_setup = std::chrono::steady_clock::now();
  g_trace.clear();
  g_budget_s = 0.0;
  g_progress = 0;
  Eigen::MatrixXd X(n, p);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) X(i, j) = x[(size_t)i * p + j];
  Eigen::VectorXd Y(n), W(n);
  for (int i = 0; i < n; i++) {
    Y(i) = y[i];
    W(i) = weight[i];
  }
  Eigen::VectorXi G(p);
  for (int j = 0; j < p; j++) G(j) = j;
  Data data(X, Y, data_type, W, is_normal != 0, G);
  Algorithm *algorithm = nullptr;
  Metric *metric = nullptr;
  if (model_type == 1) {
    data.add_weight();
    algorithm = new GroupPdasLm(data, algorithm_type, max_iter);
    metric = new LmMetric(ic_type, false, 5);
  } else if (model_type == 2) {
    algorithm = new GroupPdasLogistic(data, algorithm_type, max_iter);
    metric = new LogisticMetric(ic_type, false, 5);
  } else if (model_type == 3) {
    algorithm = new GroupPdasPoisson(data, algorithm_type, max_iter);
    metric = new PoissonMetric(ic_type, false, 5);
  } else {
    algorithm = new GroupPdasCox(data, algorithm_type, max_iter);
    metric = new CoxMetric(ic_type, false, 5);
  }
  algorithm->set_warm_start(true);
  algorithm->always_select = Eigen::VectorXi(0);
  algorithm->tao = 0.;
  Eigen::VectorXi full_mask(n);
  for (int i = 0; i < n; i++) full_mask(i) = i;
  std::vector<Eigen::MatrixXd> full_group_XTX =
      group_XTX(data.x, data.g_index, data.g_size, data.n, data.p, data.g_num, algorithm->model_type);
  Eigen::VectorXd beta_init = Eigen::VectorXd::Zero(p);
  for (int i = 0; i < init_len; i++) beta_init(init_idx[i]) = init_val[i];
  double coef0_init = init_coef0;
  *setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup).count();
  auto t0 = std::chrono::steady_clock::now();
  *n_done = 0;
  for (int i = 0; i < sequence_len; i++) {
    auto tc = std::chrono::steady_clock::now();
    algorithm->update_train_mask(full_mask);
    algorithm->update_sparsity_level(sequence[i]);
    algorithm->update_lambda_level(0.0);
    algorithm->update_beta_init(beta_init);
    algorithm->update_coef0_init(coef0_init);
    algorithm->update_group_XTX(full_group_XTX);
    algorithm->fit();
    beta_init = algorithm->get_beta();
    coef0_init = algorithm->get_coef0();
    volatile double sink = metric->train_loss(algorithm, data);
    sink = metric->ic(algorithm, data);
    (void)sink;
    auto t1 = std::chrono::steady_clock::now();
    cand_seconds[i] = std::chrono::duration<double>(t1 - tc).count();
    cand_iters[i] = algorithm->get_l();
    *n_done = i + 1;
    if (std::chrono::duration<double>(t1 - t0).count() > budget_s) break;
  }
  delete algorithm;
  delete metric;
  return 0;
}

// The reference's screening() alone (src/screening.cpp:26-105): returns the kept column indices.
int bess_ref_screening_groups(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                              int screening_size, const int *g_index, int g_len, const int *always_select,
                              int always_len, int *screening_A);

int bess_ref_screening(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                       int screening_size, const int *always_select, int always_len, int *screening_A) {
  return bess_ref_screening_groups(x, n, p, y, weight, model_type, screening_size, nullptr, 0, always_select, always_len,
                                   screening_A);
}

// ... with a group index (g_index == NULL: singleton groups): screening_A = kept GROUP numbers
int bess_ref_screening_groups(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                              int screening_size, const int *g_index, int g_len, const int *always_select,
                              int always_len, int *screening_A) {
  Eigen::MatrixXd X(n, p);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) X(i, j) = x[(size_t)i * p + j];
  Eigen::VectorXd Y(n), W(n);
  for (int i = 0; i < n; i++) {
    Y(i) = y[i];
    W(i) = weight[i];
  }
  const int gl = g_index ? g_len : p;
  Eigen::VectorXi G(gl), AS(always_len);
  for (int j = 0; j < gl; j++) G(j) = g_index ? g_index[j] : j;
  for (int i = 0; i < always_len; i++) AS(i) = always_select[i];
  Eigen::VectorXi A = screening(X, Y, W, model_type, screening_size, G, AS);
  for (int i = 0; i < A.size(); i++) screening_A[i] = A(i);
  return 0;
}

// Trace read-out.  which: 0 geta_meta(int) 1 a_flat(int) 2 beta_flat(double) 3 coef0_calls(double)
//                         4 loss_calls(double) 5 ic_calls(double)
// the reference's max_k itself (src/utilities.cpp:179-188): std::nth_element + sort on the index vector
void bess_ref_max_k(const double *score, int len, int k, int *out) {
  Eigen::VectorXd v = Eigen::Map<const Eigen::VectorXd>(score, len);
  Eigen::VectorXi r;
  max_k(v, k, r);
  for (int i = 0; i < k; i++) out[i] = r(i);
}

int bess_ref_trace_size(int which) {
  switch (which) {
    case 0: return (int)g_trace.geta_meta.size();
    case 1: return (int)g_trace.a_flat.size();
    case 2: return (int)g_trace.beta_flat.size();
    case 3: return (int)g_trace.coef0_calls.size();
    case 4: return (int)g_trace.loss_calls.size();
    case 5: return (int)g_trace.ic_calls.size();
  }
  return -1;
}

void bess_ref_trace_copy_int(int which, int *out) {
  const std::vector<int> &v = which == 0 ? g_trace.geta_meta : g_trace.a_flat;
  if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(int));
}

void bess_ref_trace_copy_double(int which, double *out) {
  const std::vector<double> *v = &g_trace.beta_flat;
  if (which == 3) v = &g_trace.coef0_calls;
  if (which == 4) v = &g_trace.loss_calls;
  if (which == 5) v = &g_trace.ic_calls;
  if (!v->empty()) std::memcpy(out, v->data(), v->size() * sizeof(double));
}

}  // extern "C"
