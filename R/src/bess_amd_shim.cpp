// R/src/bess_amd_shim.cpp -- the R package's C++ entry point bessCpp() on top of libbessx.
//
// Replaces the reference's bessCpp (src/bess.h:20-33, body src/bess.cpp:37-214) and its generated glue
// _BeSS_bessCpp (R/src/RcppExports.cpp:10-48): same 30 arguments in the same order, same named list
// (src/path.cpp:116-123 for the sequential path, :376-380 for golden section, + screening_A, src/bess.cpp:207).
// All arithmetic happens behind the C ABI (include/bessx.h, bessx_bessCpp); this file only converts R objects to
// pointers and the flat result arrays back to R objects.  No Eigen: R matrices are column-major already, which is
// what bessx_bessCpp takes.
//
// NOT COMPILED in the build image (no R, no Rcpp headers there): R boundary unpinned by execution.  Everything below
// the conversion layer -- bessx_bessCpp with column-major input and the *_all outputs -- is exercised through ctypes by
// tests/test_r_boundary_gpu.py.
#include <Rcpp.h>

#include <algorithm>
#include <vector>

#include "bessx.h"

// [[Rcpp::export]]
Rcpp::List bessCpp(Rcpp::NumericMatrix x, Rcpp::NumericVector y, int data_type, Rcpp::NumericVector weight,
                   bool is_normal, int algorithm_type, int model_type, int max_iter, int exchange_num, int path_type,
                   bool is_warm_start, int ic_type, bool is_cv, int K, Rcpp::NumericVector state,
                   Rcpp::IntegerVector sequence, Rcpp::NumericVector lambda_seq, int s_min, int s_max, int K_max,
                   double epsilon, double lambda_min, double lambda_max, int nlambda, bool is_screening,
                   int screening_size, int powell_path, Rcpp::IntegerVector g_index, Rcpp::IntegerVector always_select,
                   double tao) {
  const int n = x.nrow(), p = x.ncol();
  if (y.size() != n || weight.size() != n) Rcpp::stop("bessCpp: y and weight must have nrow(x) entries");
  const bool seqp = path_type == 1;
  const int ns = sequence.size(), nl = lambda_seq.size();
  // capacity of the *_all arrays: the sequential grid, or what a golden-section / Powell search can evaluate;
  // the reference's gs_path returns 100 fixed columns (src/path.cpp:151) and so does this wrapper
  const int cap = seqp ? ns * nl : std::max(100, 2 * (s_max - s_min + 1) + 128);
  Rcpp::NumericVector beta(p);
  std::vector<double> beta_all((size_t)p * cap), coef0_all(cap), loss_all(cap), ic_all(cap);
  std::vector<int> kept(is_screening ? std::max(screening_size, 1) : 1);
  bessx_r_result r = {};
  r.beta = beta.begin();
  r.all_capacity = cap;
  r.beta_all = beta_all.data();
  r.coef0_all = coef0_all.data();
  r.train_loss_all = loss_all.data();
  r.ic_all = ic_all.data();
  r.screening_A = kept.data();
  const int rc = bessx_bessCpp(x.begin(), n, p, y.begin(), data_type, weight.begin(), is_normal, algorithm_type,
                               model_type, max_iter, exchange_num, path_type, is_warm_start, ic_type, is_cv, K,
                               state.begin(), (int)state.size(), sequence.begin(), ns, lambda_seq.begin(), nl, s_min,
                               s_max, K_max, epsilon, lambda_min, lambda_max, nlambda, is_screening, screening_size,
                               powell_path, g_index.begin(), (int)g_index.size(), always_select.begin(),
                               (int)always_select.size(), tao, &r);
  if (rc != BESSX_OK) Rcpp::stop(bessx_last_error());
  Rcpp::List out;
  if (seqp) {
    // beta_all: list over lambda of p x length(sequence); coef0_all / train_loss_all: list over lambda of vectors;
    // ic_all: length(sequence) x length(lambda_seq) matrix
    Rcpp::List b_all(nl), c_all(nl), l_all(nl);
    for (int j = 0; j < nl; j++) {
      Rcpp::NumericMatrix bm(p, ns);
      std::copy(beta_all.begin() + (size_t)j * ns * p, beta_all.begin() + (size_t)(j + 1) * ns * p, bm.begin());
      b_all[j] = bm;
      c_all[j] = Rcpp::NumericVector(coef0_all.begin() + (size_t)j * ns, coef0_all.begin() + (size_t)(j + 1) * ns);
      l_all[j] = Rcpp::NumericVector(loss_all.begin() + (size_t)j * ns, loss_all.begin() + (size_t)(j + 1) * ns);
    }
    Rcpp::NumericMatrix icm(ns, nl);
    std::copy(ic_all.begin(), ic_all.begin() + (size_t)ns * nl, icm.begin());
    out = Rcpp::List::create(Rcpp::Named("beta") = beta, Rcpp::Named("coef0") = r.coef0,
                             Rcpp::Named("train_loss") = r.train_loss, Rcpp::Named("ic") = r.ic,
                             Rcpp::Named("lambda") = r.lambda, Rcpp::Named("beta_all") = b_all,
                             Rcpp::Named("coef0_all") = c_all, Rcpp::Named("train_loss_all") = l_all,
                             Rcpp::Named("ic_all") = icm);
  } else {
    const int ncol = 100;  // src/path.cpp:151-154: fixed width, zero beyond the evaluated points
    Rcpp::NumericMatrix bm(p, ncol);
    Rcpp::NumericVector c_all(ncol), l_all(ncol), i_all(ncol);
    const int k = std::min(std::min(r.n_all, cap), ncol);
    std::copy(beta_all.begin(), beta_all.begin() + (size_t)k * p, bm.begin());
    std::copy(coef0_all.begin(), coef0_all.begin() + k, c_all.begin());
    std::copy(loss_all.begin(), loss_all.begin() + k, l_all.begin());
    std::copy(ic_all.begin(), ic_all.begin() + k, i_all.begin());
    out = Rcpp::List::create(Rcpp::Named("beta") = beta, Rcpp::Named("coef0") = r.coef0,
                             Rcpp::Named("train_loss") = r.train_loss, Rcpp::Named("ic") = r.ic,
                             Rcpp::Named("lambda") = r.lambda, Rcpp::Named("beta_all") = bm,
                             Rcpp::Named("coef0_all") = c_all, Rcpp::Named("train_loss_all") = l_all,
                             Rcpp::Named("ic_all") = i_all);
  }
  if (is_screening) out["screening_A"] = Rcpp::IntegerVector(kept.begin(), kept.begin() + screening_size);
  return out;
}
