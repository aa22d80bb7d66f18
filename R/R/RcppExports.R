# R/R/RcppExports.R -- the R stub Rcpp::compileAttributes() generates for bessCpp() (R/src/bess_amd_shim.cpp): what
# bess() calls (R/R/bess.R).  Written by hand in the generator's layout (no R in the build image); regenerate with
# Rcpp::compileAttributes("R").

bessCpp <- function(x, y, data_type, weight, is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type, is_warm_start, ic_type, is_cv, K, state, sequence, lambda_seq, s_min, s_max, K_max, epsilon, lambda_min, lambda_max, nlambda, is_screening, screening_size, powell_path, g_index, always_select, tao) {
    .Call(`_BeSSamd_bessCpp`, x, y, data_type, weight, is_normal, algorithm_type, model_type, max_iter, exchange_num, path_type, is_warm_start, ic_type, is_cv, K, state, sequence, lambda_seq, s_min, s_max, K_max, epsilon, lambda_min, lambda_max, nlambda, is_screening, screening_size, powell_path, g_index, always_select, tao)
}
