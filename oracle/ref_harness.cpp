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

template <class Base>
struct TracedAlgorithm : public Base {
  TracedAlgorithm(Data &data, int algorithm_type, unsigned int max_iter) : Base(data, algorithm_type, max_iter) {}

  void get_A(Eigen::MatrixXd X, Eigen::VectorXd y, Eigen::VectorXd beta, double coef0, int T0, Eigen::VectorXd weights,
             Eigen::VectorXi index, Eigen::VectorXi gsize, int N, Eigen::VectorXi &A_out) override {
    int train_n = (int)X.rows();
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

// The reference's screening() alone (src/screening.cpp:26-105): returns the kept column indices.
int bess_ref_screening(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                       int screening_size, const int *always_select, int always_len, int *screening_A) {
  Eigen::MatrixXd X(n, p);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) X(i, j) = x[(size_t)i * p + j];
  Eigen::VectorXd Y(n), W(n);
  for (int i = 0; i < n; i++) {
    Y(i) = y[i];
    W(i) = weight[i];
  }
  Eigen::VectorXi G(p), AS(always_len);
  for (int j = 0; j < p; j++) G(j) = j;
  for (int i = 0; i < always_len; i++) AS(i) = always_select[i];
  Eigen::VectorXi A = screening(X, Y, W, model_type, screening_size, G, AS);
  for (int i = 0; i < A.size(); i++) screening_A[i] = A(i);
  return 0;
}

// Trace read-out.  which: 0 geta_meta(int) 1 a_flat(int) 2 beta_flat(double) 3 coef0_calls(double)
//                         4 loss_calls(double) 5 ic_calls(double)
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
