// bessx_abi.cpp -- the drop-in entry points bessx_pywrap_bess / bessx_bessCpp (src/bess.h:20-51) and the single-kernel ops
#include "bessx_host.h"

extern "C" {

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

// the same sums for nc vectors at once by the multi-chain kernel (k_xtv_mc: one pass over X): row c of v / v2 / out / out2
// belongs to chain c.  Bitwise the results of nc calls of bessx_op_xtv (tests/test_ops_gpu.py).
int bessx_op_xtv_multi(const double *x, int n, int p, int ld, const double *v, const double *v2, int nc, double *out,
                       double *out2) {
  if (int rc = need_device()) return rc;
  if (!x || !v || !out || n < 1 || p < 1 || ld < n || nc < 1 || nc > XTV_MC_MAX || (v2 && !out2))
    return fail(BESSX_ERR_ARG, "op_xtv_multi: bad arguments");
  Scratch sc;
  const int U = n >= 4096 ? 8 : (n >= 2048 ? 4 : (n >= 1024 ? 2 : 1));
  double *dX, *dout;
  long ldd;
  if (int rc = upload_padded(sc, x, n, p, ld, U, &dX, &ldd)) return rc;
  const int nrb = (int)(ldd / (128L * U));
  XtvMc a = {};
  a.nc = nc;
  for (int c = 0; c < nc; c++) {
    double *dv, *dv2 = nullptr, *part, *part2 = nullptr;
    if (int rc = upload_vec_padded(sc, v + (size_t)c * n, n, ldd, &dv)) return rc;
    if (v2)
      if (int rc = upload_vec_padded(sc, v2 + (size_t)c * n, n, ldd, &dv2)) return rc;
    HIPX(sc.alloc(&part, (size_t)nrb * p));
    HIPX(sc.alloc(&part2, (size_t)nrb * p));
    a.v[c] = dv;
    a.v2[c] = dv2;
    a.part[c] = part;
    a.part2[c] = part2;
    a.ctrl[c] = nullptr;
    a.slot[c] = 0;
  }
  HIPX(sc.alloc(&dout, (size_t)p));
  HIPX(launch_xtv_mc(dX, ldd, p, U, a, v2 != nullptr, nullptr));
  for (int c = 0; c < nc; c++) {
    HIPX(launch_part_sum(a.part[c], nrb, p, dout, nullptr));
    HIPX(hipMemcpy(out + (size_t)c * p, dout, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    if (v2) {
      HIPX(launch_part_sum(a.part2[c], nrb, p, dout, nullptr));
      HIPX(hipMemcpy(out2 + (size_t)c * p, dout, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    }
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

int bessx_op_xtv_multi_bench(int n, int p, int nc, int two, int repeats, double *gbps, double *avg_ms) {
  if (int rc = need_device()) return rc;
  if (n < 1 || p < 1 || repeats < 1 || !gbps || nc < 1 || nc > XTV_MC_MAX) return fail(BESSX_ERR_ARG, "op_xtv_multi_bench: bad arguments");
  Scratch sc;
  const long ld = ((long)n + 1023) / 1024 * 1024;
  double *dX, *dv, *part;
  HIPX(sc.alloc(&dX, (size_t)ld * p));
  HIPX(sc.alloc(&dv, (size_t)ld * 2 * nc));
  HIPX(sc.alloc(&part, (size_t)(ld / 1024) * p * 2 * nc));
  HIPX(launch_fill(dX, ld * (long)p, 1.0, nullptr));
  HIPX(launch_fill(dv, ld * 2 * nc, 0.5, nullptr));
  XtvMc a = {};
  a.nc = nc;
  for (int c = 0; c < nc; c++) {
    a.v[c] = dv + (size_t)(2 * c) * ld;
    a.v2[c] = two ? dv + (size_t)(2 * c + 1) * ld : nullptr;
    a.part[c] = part + (size_t)(2 * c) * (ld / 1024) * p;
    a.part2[c] = part + (size_t)(2 * c + 1) * (ld / 1024) * p;
  }
  hipEvent_t e0, e1;
  HIPX(hipEventCreate(&e0));
  HIPX(hipEventCreate(&e1));
  HIPX(launch_xtv_mc(dX, ld, p, 8, a, two != 0, nullptr));
  HIPX(hipEventRecord(e0, nullptr));
  for (int i = 0; i < repeats; i++) HIPX(launch_xtv_mc(dX, ld, p, 8, a, two != 0, nullptr));
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
  // variant 10 + nc: the multi-chain kernel (k_cox_score1p_mc) with nc chains' vector sets per launch
  const int mc = (variant >= 11 && variant <= 10 + COX_MC_MAX) ? variant - 10 : 0;
  if (!mc && (variant < 0 || variant > 1)) return fail(BESSX_ERR_ARG, "op_cox_score_bench: variant 0, 1 or 11..16");
  if (mc) {
    double *vecs, *outs;
    HIPX(sc.alloc(&vecs, (size_t)ld * 4 * mc));
    HIPX(sc.alloc(&outs, ((size_t)5 * nrb * p + nrb) * mc));
    HIPX(launch_fill(vecs, ld * 4 * mc, 0.5, nullptr));
    CoxMc a = {};
    a.nc = mc;
    for (int c = 0; c < mc; c++) {
      a.TH[c] = vecs + (size_t)(4 * c) * ld;
      a.CU[c] = vecs + (size_t)(4 * c + 1) * ld;
      a.CV[c] = vecs + (size_t)(4 * c + 2) * ld;
      a.C2[c] = vecs + (size_t)(4 * c + 3) * ld;
      a.out[c] = outs + ((size_t)5 * nrb * p + nrb) * c;
    }
    HIPX(launch_cox_score1p_mc(dX, ld, p, 8, nrb, a, nullptr));
    HIPX(hipEventRecord(e0, nullptr));
    for (int i = 0; i < repeats; i++) HIPX(launch_cox_score1p_mc(dX, ld, p, 8, nrb, a, nullptr));
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
