// pybind_cbess.cpp -- pybind11 counterpart of the reference's SWIG module `cbess`
// (/root/reference python/src/bess.i:17-30, generated python/bess/cbess.py:65-66).
//
// Exposes pywrap_bess with the SAME 38 positional arguments bess_base.fit passes
// (python/bess/linear.py:360-375: 30 inputs followed by the eight ARGOUT lengths
// p,1,1,1,1,1,1,p) and the same 10-tuple result (beta, coef0, train_loss, ic, nullloss,
// aic, bic, gic, A_out, l_out).  It only marshals NumPy buffers into the C ABI
// (bessx_pywrap_bess, include/bessx.h); the GIL is released around the solve.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <stdexcept>
#include <string>

#include "bessx.h"

namespace py = pybind11;
using darr = py::array_t<double, py::array::c_style | py::array::forcecast>;
using iarr = py::array_t<int, py::array::c_style | py::array::forcecast>;

static py::tuple pywrap_bess(darr x, darr y, int data_type, darr weight, bool is_normal, int algorithm_type,
                             int model_type, int max_iter, int exchange_num, int path_type, bool is_warm_start,
                             int ic_type, bool is_cv, int K, iarr gindex, darr state, iarr sequence,
                             darr lambda_sequence, int s_min, int s_max, int K_max, double epsilon, double lambda_min,
                             double lambda_max, int n_lambda, bool is_screening, int screening_size, int powell_path,
                             iarr always_select, double tao, int beta_out_len, int coef0_out_len,
                             int train_loss_out_len, int ic_out_len, int aic_out_len, int bic_out_len,
                             int gic_out_len, int A_out_len) {
  if (x.ndim() != 2) throw std::invalid_argument("x must be a 2-d array");
  const int n = (int)x.shape(0), p = (int)x.shape(1);
  darr beta(beta_out_len), coef0(coef0_out_len), loss(train_loss_out_len), ic(ic_out_len), aic(aic_out_len),
      bic(bic_out_len), gic(gic_out_len);
  iarr a_out(A_out_len);
  double nullloss = 0.0;
  int l_out = 0, rc;
  {
    py::gil_scoped_release nogil;
    rc = bessx_pywrap_bess(
        x.mutable_data(), n, p, y.mutable_data(), (int)y.size(), data_type, weight.mutable_data(), (int)weight.size(),
        is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type, is_warm_start, ic_type, is_cv, K,
        gindex.mutable_data(), (int)gindex.size(), state.mutable_data(), (int)state.size(), sequence.mutable_data(),
        (int)sequence.size(), lambda_sequence.mutable_data(), (int)lambda_sequence.size(), s_min, s_max, K_max,
        epsilon, lambda_min, lambda_max, n_lambda, is_screening, screening_size, powell_path,
        always_select.mutable_data(), (int)always_select.size(), tao, beta.mutable_data(), beta_out_len,
        coef0.mutable_data(), coef0_out_len, loss.mutable_data(), train_loss_out_len, ic.mutable_data(), ic_out_len,
        &nullloss, aic.mutable_data(), aic_out_len, bic.mutable_data(), bic_out_len, gic.mutable_data(), gic_out_len,
        a_out.mutable_data(), A_out_len, &l_out);
  }
  if (rc != BESSX_OK) throw std::runtime_error(std::string("libbessx error ") + std::to_string(rc) + ": " + bessx_last_error());
  return py::make_tuple(beta, coef0, loss, ic, nullloss, aic, bic, gic, a_out, l_out);
}

PYBIND11_MODULE(_cbess, m) {
  m.doc() = "pybind11 binding of libbessx (MI355X-native PDAS solver): drop-in for the reference's SWIG module cbess";
  m.def("pywrap_bess", &pywrap_bess, "Same positional signature as bess.cbess.pywrap_bess");
}
