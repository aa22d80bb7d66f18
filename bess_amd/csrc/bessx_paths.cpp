// bessx_paths.cpp -- sequential_path / gs_path / pgs_path (src/path.cpp) and the session-level entry points built on them
#include "bessx_host.h"

namespace bessx {

// --------------------------------------------------------------------------------------------
// paths (src/path.cpp)
// --------------------------------------------------------------------------------------------

void denormalize(const bessx_session *s, SparseVec &b, double &coef0, bool gs_variant) {
  // src/path.cpp:76-110 (sequential) and :330-342 (golden section: data_type 3 also takes the "else")
  if (!s->is_normal) return;
  double dot = 0.0, sn = std::sqrt((double)s->n);
  for (size_t i = 0; i < b.idx.size(); i++) {
    b.val[i] = sn * b.val[i] / s->x_norm_h[b.idx[i]];
    dot += b.val[i] * s->x_mean_h[b.idx[i]];
  }
  if (s->data_type == 1)
    coef0 = s->y_mean_h - dot;
  else if (s->data_type == 2 || gs_variant)
    coef0 = coef0 - dot;
}

int run_fit(bessx_session *s, int T0, double lambda, const SparseVec &beta_init, double coef0_init) {
  s->cur_rows = 0;  // update_train_mask(full_mask) + update_group_XTX(full_group_XTX)
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init = beta_init;
  s->coef0_init = coef0_init;
  return algorithm_fit(s);
}


void store_candidate(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant) {
  int i = res->n_candidates++;
  if (i >= res->capacity) return;
  SparseVec b = c.beta;
  double c0 = c.coef0;
  denormalize(s, b, c0, gs_variant);
  if (res->cand_T0) res->cand_T0[i] = c.T0;
  if (res->cand_lambda) res->cand_lambda[i] = c.lambda;
  if (res->cand_iters) res->cand_iters[i] = c.iters;
  if (res->cand_train_loss) res->cand_train_loss[i] = c.loss;
  if (res->cand_ic) res->cand_ic[i] = c.ic;
  if (res->cand_coef0) res->cand_coef0[i] = c0;
  for (int j = 0; j < res->max_T0; j++) {
    bool has = j < (int)b.idx.size();
    if (res->cand_support) res->cand_support[(size_t)i * res->max_T0 + j] = has ? caller_col(s, b.idx[j]) : -1;
    if (res->cand_beta) res->cand_beta[(size_t)i * res->max_T0 + j] = has ? b.val[j] : 0.0;
  }
}

void store_best(bessx_session *s, bessx_path_result *res, const Candidate &c, bool gs_variant) {
  SparseVec b = c.beta;
  double c0 = c.coef0;
  denormalize(s, b, c0, gs_variant);
  if (res->beta) {
    std::fill(res->beta, res->beta + s->p_full, 0.0);  // beta_screening_A of src/bess.cpp:186-197
    for (size_t i = 0; i < b.idx.size(); i++) res->beta[caller_col(s, b.idx[i])] = b.val[i];
  }
  res->coef0 = c0;
  res->train_loss = c.loss;
  res->ic = c.ic;
  res->lambda = c.lambda;
  res->best_T0 = c.T0;
  res->best_iters = c.iters;
}

// does candidate c equal row `row` of the chain's stop table (supports in the caller's numbering, -1 padded;
// coefficients de-normalised like cand_beta, compared to stop_rtol when given)?
bool chain_row_matches(const bessx_session *s, const bessx_path_chain *ch, int row, const Candidate &c) {
  if (!ch->stop_support || row >= ch->stop_rows) return false;
  const int *want = ch->stop_support + (size_t)row * ch->stop_row_len;
  const int k = (int)c.beta.idx.size();
  if (k > ch->stop_row_len) return false;
  for (int j = 0; j < ch->stop_row_len; j++)
    if (want[j] != (j < k ? caller_col(s, c.beta.idx[j]) : -1)) return false;
  if (ch->stop_beta) {
    SparseVec b = c.beta;
    double c0 = c.coef0;
    denormalize(s, b, c0, false);
    const double *wb = ch->stop_beta + (size_t)row * ch->stop_row_len;
    for (int j = 0; j < k; j++)
      if (!(std::fabs(b.val[j] - wb[j]) <= ch->stop_rtol * std::max(std::fabs(b.val[j]), std::fabs(wb[j])))) return false;
  }
  return true;
}

int sequential_path(bessx_session *s, const int *seq, int ns, const double *lam, int nl, int ic_type,
                           int is_cv, bessx_path_result *res, bessx_path_chain *chain) {
  // src/path.cpp:25-132
  SparseVec beta_init;
  double coef0_init = 0.0;
  if (chain) {
    // the warm-start chain of :60-64 continued from a model another process holds (update_beta_init /
    // update_coef0_init, src/Algorithm.h:85-93, before the first candidate)
    for (int i = 0; i < chain->init_len; i++) {
      if (chain->init_idx[i] < 0 || chain->init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "chain: init index out of range");
      beta_init.idx.push_back(chain->init_idx[i]);
      beta_init.val.push_back(chain->init_val[i]);
    }
    coef0_init = chain->init_coef0;
    chain->stopped_at = -1;
    chain->last_len = 0;
    chain->last_coef0 = 0.0;
  }
  std::vector<Candidate> grid((size_t)ns * nl);
  std::vector<char> have((size_t)ns * nl, 0);
  bool stop = false;
  for (int i = 0; i < ns && !stop; i++) {
    int step = (i % 2 == 0) ? 1 : -1;
    for (int j = (i % 2 == 0) ? 0 : nl - 1; j < nl && j >= 0 && !stop; j += step) {
      {
        // announce the fit that follows in the snake order (src/path.cpp:36-50): it can be chained on the device
        int jn = j + step, in = i;
        if (jn < 0 || jn >= nl) {
          in = i + 1;
          jn = (in % 2 == 0) ? 0 : nl - 1;
        }
        s->hint.on = in < ns && !is_cv;
        if (s->hint.on) {
          s->hint.T0 = seq[in];
          s->hint.lambda = lam[jn];
        }
      }
      if (int rc = run_fit(s, seq[i], lam[j], beta_init, coef0_init)) return rc;
      if (s->warm_start) {
        beta_init = s->beta;
        coef0_init = s->coef0;
      }
      Candidate &c = grid[(size_t)j * ns + i];
      have[(size_t)j * ns + i] = 1;
      c.T0 = seq[i];
      c.lambda = lam[j];
      c.beta = s->beta;
      c.coef0 = s->coef0;
      c.iters = s->l;
      if (int rc = metric_train_loss(s, &c.loss)) return rc;
      if (int rc = metric_ic(s, ic_type, is_cv, &c.ic)) return rc;
      store_candidate(s, res, c, false);
      if (s->kch_owner) {
        kchains_progress(s, res->n_candidates);
        kchains_safe_point(s);
      }
      if (chain && chain_row_matches(s, chain, res->n_candidates - 1, c)) {
        // from here on the chain the caller already holds IS this chain: same model, same successor
        chain->stopped_at = res->n_candidates - 1;
        stop = true;
      }
    }
  }
  size_t best = 0;  // minCoeff over the column-major (ns x nl) matrix: first minimum in storage order
  while (best < grid.size() && !have[best]) best++;
  for (size_t q = best; q < grid.size(); q++)
    if (have[q] && grid[q].ic < grid[best].ic) best = q;
  store_best(s, res, grid[best], false);
  if (chain) {
    // Algorithm::beta / coef0 as the path would hand them to the next candidate (normalised scale)
    chain->last_len = (int)beta_init.idx.size();
    chain->last_coef0 = coef0_init;
    for (int i = 0; i < chain->last_len && i < chain->last_cap; i++) {
      if (chain->last_idx) chain->last_idx[i] = beta_init.idx[i];
      if (chain->last_val) chain->last_val[i] = beta_init.val[i];
    }
  }
  return 0;
}

int gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res) {
  // src/path.cpp:134-389; lambda stays at its constructor default 0
  SparseVec beta_init;
  double coef0_init = 0.0;
  int Tmin = s_min, Tmax = s_max;
  int T1 = (int)std::round(0.618 * Tmin + 0.382 * Tmax), T2 = (int)std::round(0.382 * Tmin + 0.618 * Tmax);
  double ic1 = 0, ic2 = 0, icT1 = 0, icT2 = 0;
  auto fit_point = [&](int T, double *ic_first, double *ic_second) -> int {
    if (int rc = run_fit(s, T, 0.0, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    Candidate c;
    c.T0 = T;
    c.lambda = 0.0;
    c.beta = s->beta;
    c.coef0 = s->coef0;
    c.iters = s->l;
    if (int rc = metric_train_loss(s, &c.loss)) return rc;
    if (int rc = metric_ic(s, ic_type, is_cv, &c.ic)) return rc;
    store_candidate(s, res, c, true);
    *ic_first = c.ic;
    if (ic_second)
      if (int rc = metric_ic(s, ic_type, is_cv, ic_second)) return rc;  // evaluated twice, :204+:210 etc.
    return 0;
  };
  if (int rc = fit_point(T1, &ic1, nullptr)) return rc;
  icT1 = ic1;
  if (int rc = fit_point(T2, &ic2, &icT2)) return rc;
  while (T1 != T2) {
    if (icT1 < icT2) {
      Tmax = T2;
      T2 = T1;
      ic2 = ic1;
      icT2 = ic1;
      T1 = (int)std::round(0.618 * Tmin + 0.382 * Tmax);
      if (int rc = fit_point(T1, &ic1, &icT1)) return rc;
    } else {
      Tmin = T1;
      T1 = T2;
      ic1 = ic2;
      icT1 = ic2;
      T2 = (int)std::round(0.382 * Tmin + 0.618 * Tmax);
      if (int rc = fit_point(T2, &ic2, &icT2)) return rc;
    }
  }
  Candidate best;
  best.T0 = 0;
  best.lambda = 0.0;
  best.coef0 = 0.0;
  best.loss = 0.0;
  best.ic = DBL_MAX;
  best.iters = 0;
  for (int T = Tmin; T <= Tmax; T++) {
    if (int rc = run_fit(s, T, 0.0, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    int iters_full = s->l;
    double v;
    if (int rc = metric_ic(s, ic_type, is_cv, &v)) return rc;
    if (v < best.ic) {
      // read AFTER ic(): under CV these are the last fold's fit, src/path.cpp:314-319
      best.T0 = T;
      best.beta = s->beta;
      best.coef0 = s->coef0;
      if (int rc = metric_train_loss(s, &best.loss)) return rc;
      best.ic = v;
      best.iters = iters_full;
      store_candidate(s, res, best, true);
    }
  }
  store_best(s, res, best, true);
  return 0;
}

// --------------------------------------------------------------------------------------------
// Powell path for the L0L2 / bsrr types: pgs_path with golden-section or sequential line searches over
// (s, log lambda), src/path.cpp:391-1309.  Pure host control over algorithm_fit(); the ic_sequence matrix the
// reference fills on the side (returned only by the R build as ic_mat) is not kept.
// --------------------------------------------------------------------------------------------
namespace powell {

static int sgn(double a) { return a > 0 ? 1 : (a < 0 ? -1 : 0); }
static double det2(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }

static bool line_intersection(double l1[2][2], double l2[2][2], double out[2]) {  // :414-440
  double xd[2] = {l1[0][0] - l1[1][0], l2[0][0] - l2[1][0]}, yd[2] = {l1[0][1] - l1[1][1], l2[0][1] - l2[1][1]};
  double div = det2(xd, yd);
  if (div == 0) return false;
  double d[2] = {det2(l1[0], l1[1]), det2(l2[0], l2[1])};
  out[0] = det2(d, xd) / div;
  out[1] = det2(d, yd) / div;
  return true;
}

static void cal_intersections(const double p[2], const double u[2], int s_min, int s_max, double lmin, double lmax,
                              double a[2], double b[2]) {  // :445-577
  double l0[2][2] = {{p[0], p[1]}, {p[0] + u[0], p[1] + u[1]}};
  double ls[4][2][2] = {{{(double)s_min, lmin}, {(double)s_min, lmax}},
                        {{(double)s_max, lmin}, {(double)s_max, lmax}},
                        {{(double)s_min, lmin}, {(double)s_max, lmin}},
                        {{(double)s_min, lmax}, {(double)s_max, lmax}}};
  double is[4][2];
  bool ok[4];
  for (int i = 0; i < 4; i++) ok[i] = line_intersection(l0, ls[i], is[i]);
  for (int i = 0; i < 4; i++)
    if (ok[i] && (is[i][0] < s_min - 0.0001 || is[i][0] > s_max + 0.0001 || is[i][1] < lmin - 0.001 ||
                  is[i][1] > lmax + 0.001))
      ok[i] = false;
  for (int i = 0; i < 4; i++)
    if (ok[i])
      for (int j = i + 1; j < 4; j++)
        if (ok[j] && std::fabs(is[i][0] - is[j][0]) < 0.0001 && std::fabs(is[i][1] - is[j][1]) < 0.0001) ok[j] = false;
  int j = 0;
  for (int i = 0; i < 4; i++)
    if (ok[i]) {
      if (j == 2) j += 1;
      if (j == 1) {
        b[0] = is[i][0];
        b[1] = is[i][1];
        j += 1;
      }
      if (j == 0) {
        a[0] = is[i][0];
        a[1] = is[i][1];
        j += 1;
      }
    }
}

struct Search {
  bessx_session *s;
  int ic_type, is_cv;
  SparseVec beta_init;
  double coef0_init = 0.0;
  int fit(int T0, double lambda) {
    if (int rc = run_fit(s, T0, lambda, beta_init, coef0_init)) return rc;
    if (s->warm_start) {
      beta_init = s->beta;
      coef0_init = s->coef0;
    }
    return 0;
  }
};

struct Point {  // what a line search reports back
  SparseVec beta;
  double coef0 = 0, loss = 0, ic = 0;
};

#define PW_TRY(expr)              \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)

// golden_section_search, :579-935
static int golden_section_search(bessx_session *s, int ic_type, int is_cv, const double p[2], const double u[2],
                                 int s_min, int s_max, double lmin, double lmax, double best_arg[2], Point &out) {
  Search ps{s, ic_type, is_cv, SparseVec(), 0.0};
  SparseVec bt1, bt2;
  double lt1 = 0, lt2 = 0, c01 = 0, c02 = 0, closs, dloss, a[2] = {0, 0}, b[2] = {0, 0}, c[2], d[2], h[2];
  const double s_tol = 2, ltol = (lmax - lmin) / 200;
  const double invphi = (std::pow(5, 0.5) - 1.0) / 2.0, invphi2 = (3.0 - std::pow(5, 0.5)) / 2.0;
  cal_intersections(p, u, s_min, s_max, lmin, lmax, a, b);
  h[0] = b[0] - a[0];
  h[1] = b[1] - a[1];
  c[0] = a[0] + invphi2 * h[0];
  c[1] = a[1] + invphi2 * h[1];
  d[0] = a[0] + invphi * h[0];
  d[1] = a[1] + invphi * h[1];
  if (h[0] > 0.0001) {
    c[0] = (int)c[0];
    d[0] = std::ceil(d[0]);
  } else if (h[0] < -0.0001) {
    c[0] = std::ceil(c[0]);
    d[0] = (int)d[0];
  } else {
    c[0] = std::round(c[0]);
    d[0] = std::round(d[0]);
  }
  PW_TRY(ps.fit((int)c[0], std::exp(c[1])));
  PW_TRY(metric_ic(s, ic_type, is_cv, &closs));
  c01 = s->coef0;
  bt1 = s->beta;
  PW_TRY(metric_train_loss(s, &lt1));
  PW_TRY(ps.fit((int)d[0], std::exp(d[1])));
  PW_TRY(metric_ic(s, ic_type, is_cv, &dloss));
  c02 = s->coef0;
  bt2 = s->beta;
  PW_TRY(metric_train_loss(s, &lt2));
  int tt = 0;
  for (;;) {
    if ((std::fabs((invphi2 - invphi) * h[0]) <= s_tol && std::fabs((invphi2 - invphi) * h[1]) < ltol) || tt == 50) {
      double min_loss, tmp;
      if (closs < dloss) {
        best_arg[0] = c[0];
        best_arg[1] = c[1];
        min_loss = closs;
        out.beta = bt1;
        out.coef0 = c01;
        out.ic = closs;
        out.loss = lt1;
      } else {
        best_arg[0] = d[0];
        best_arg[1] = d[1];
        min_loss = dloss;
        out.beta = bt2;
        out.coef0 = c02;
        out.ic = dloss;
        out.loss = lt2;
      }
      for (int i = 1; i < std::fabs((invphi2 - invphi) * h[0]); i++) {
        PW_TRY(ps.fit((int)(c[0] + sgn(h[0]) * i), std::exp(c[1])));
        PW_TRY(metric_ic(s, ic_type, is_cv, &tmp));
        if (tmp < min_loss) {
          best_arg[0] = c[0] + sgn(h[0]) * i;
          best_arg[1] = c[1];
          min_loss = tmp;
          out.beta = s->beta;
          out.coef0 = s->coef0;
          PW_TRY(metric_train_loss(s, &out.loss));
          out.ic = min_loss;
        }
      }
      return 0;
    }
    if (tt >= 100) return 0;
    tt++;
    if (closs < dloss) {
      // the stored model of d (bt2, c02, lt2) is deliberately NOT moved along with the point (:762-766)
      b[0] = d[0];
      b[1] = d[1];
      d[0] = c[0];
      d[1] = c[1];
      dloss = closs;
      h[0] = b[0] - a[0];
      h[1] = b[1] - a[1];
      c[0] = a[0] + invphi2 * h[0];
      c[1] = a[1] + invphi2 * h[1];
      c[0] = h[0] > 0.0001 ? (double)(int)c[0] : (h[0] < -0.0001 ? std::ceil(c[0]) : std::round(c[0]));
      PW_TRY(ps.fit((int)c[0], std::exp(c[1])));
      PW_TRY(metric_ic(s, ic_type, is_cv, &closs));
      c01 = s->coef0;
      bt1 = s->beta;
      PW_TRY(metric_train_loss(s, &lt1));
    } else {
      a[0] = c[0];
      a[1] = c[1];
      c[0] = d[0];
      c[1] = d[1];
      closs = dloss;
      h[0] = b[0] - a[0];
      h[1] = b[1] - a[1];
      d[0] = a[0] + invphi * h[0];
      d[1] = a[1] + invphi * h[1];
      d[0] = h[0] > 0.0001 ? std::ceil(d[0]) : (h[0] < -0.0001 ? (double)(int)d[0] : std::round(d[0]));
      PW_TRY(ps.fit((int)d[0], std::exp(d[1])));
      PW_TRY(metric_ic(s, ic_type, is_cv, &dloss));
      c02 = s->coef0;
      bt2 = s->beta;
      PW_TRY(metric_train_loss(s, &lt2));
    }
  }
}

static int gdc_int(int a, int b) {  // GDC, :937-953
  int Max = a > b ? a : b, Min = (a == Max) ? b : a, z = Min;
  while (Max % Min != 0) {
    z = Max % Min;
    Max = Min;
    Min = z;
  }
  return z;
}

// seq_search, :954-1137 (u is normalised in place like the reference does)
static int seq_search(bessx_session *s, int ic_type, int is_cv, double p[2], double u[2], int s_min, int s_max,
                      double lmin, double lmax, double best_arg[2], Point &out, int nlambda) {
  Search ps{s, ic_type, is_cv, SparseVec(), 0.0};
  const double d_lambda = (lmax - lmin) / (nlambda - 1);
  const size_t cap = (size_t)(s_max - s_min + 1) * nlambda + 2;
  int k_lambda = (int)std::fabs(std::round(u[1] / d_lambda));
  if (std::fabs(u[0]) != 1 && k_lambda != 1) {
    if (k_lambda == 0 && u[0] != 0) {
      u[0] = u[0] / std::fabs(u[0]);
    } else if (u[0] == 0 && k_lambda != 0) {
      u[1] = u[1] / k_lambda;
    } else if (!(k_lambda == 0 && (int)u[0] == 0)) {  // the reference divides by zero there
      int g = gdc_int(k_lambda, std::abs((int)u[0]));
      if (g) {
        u[0] = std::round(u[0] / g);
        u[1] = u[1] / g;
      }
    }
  }
  std::vector<Point> f1, f2;
  auto eval = [&](int T0, double lambda, std::vector<Point> &dst) -> int {
    PW_TRY(ps.fit(T0, lambda));
    Point q;
    PW_TRY(metric_ic(s, ic_type, is_cv, &q.ic));
    q.beta = s->beta;
    q.coef0 = s->coef0;
    PW_TRY(metric_train_loss(s, &q.loss));
    dst.push_back(q);
    return 0;
  };
  PW_TRY(eval((int)(p[0]), std::exp(p[1]), f1));
  f2.push_back(f1[0]);
  SparseVec beta_warm = ps.beta_init;
  double coef0_warm = ps.coef0_init;
  for (int i = 1; (p[0] + i * u[0] <= s_max) && (p[1] + i * u[1] <= lmax + d_lambda * 1e-4) &&
                  (p[0] + i * u[0] >= s_min) && (p[1] + i * u[1] >= lmin - d_lambda * 1e-4) && f1.size() < cap;
       i++)
    PW_TRY(eval((int)(p[0] + i * u[0]), std::exp(p[1] + i * u[1]), f1));
  ps.beta_init = beta_warm;
  ps.coef0_init = coef0_warm;
  for (int j = 1; (p[0] - j * u[0] <= s_max) && (p[1] - j * u[1] <= lmax + d_lambda * 1e-4) &&
                  (p[0] - j * u[0] >= s_min) && (p[1] - j * u[1] >= lmin - d_lambda * 1e-4) && f2.size() < cap;
       j++)
    PW_TRY(eval((int)(p[0] - j * u[0]), std::exp(p[1] - j * u[1]), f2));
  size_t m1 = 0, m2 = 0;
  for (size_t q = 1; q < f1.size(); q++)
    if (f1[q].ic < f1[m1].ic) m1 = q;
  for (size_t q = 1; q < f2.size(); q++)
    if (f2[q].ic < f2[m2].ic) m2 = q;
  int minpos;
  if (f1[m1].ic < f2[m2].ic) {
    minpos = (int)m1;
    out = f1[m1];
  } else {
    minpos = -(int)m2;
    out = f2[m2];
  }
  best_arg[0] = p[0] + minpos * u[0];
  best_arg[1] = p[1] + minpos * u[1];
  return 0;
}

}  // namespace powell

// pgs_path, :1138-1309
int pgs_path(bessx_session *s, int s_min, int s_max, double lmin, double lmax, int powell_path, int nlambda,
                    int ic_type, int is_cv, bessx_path_result *res) {
  using namespace powell;
  if (powell_path == 1) nlambda = 100;
  double P[3][2] = {{(double)s_min, lmin}, {0, 0}, {0, 0}};
  double U[2][2] = {{0., (lmax - lmin) / (nlambda - 1)}, {1., 0.}};
  std::vector<Candidate> all;
  Point pt;
  auto search = [&](double *pin, double *uu, double *pout) -> int {
    return powell_path == 1 ? golden_section_search(s, ic_type, is_cv, pin, uu, s_min, s_max, lmin, lmax, pout, pt)
                            : seq_search(s, ic_type, is_cv, pin, uu, s_min, s_max, lmin, lmax, pout, pt, nlambda);
  };
  auto record = [&](size_t idx, double lam) {
    if (all.size() <= idx) all.resize(idx + 1);
    Candidate &c = all[idx];
    c.T0 = (int)pt.beta.idx.size();
    c.lambda = lam;
    c.beta = pt.beta;
    c.coef0 = pt.coef0;
    c.loss = pt.loss;
    c.ic = pt.ic;
    c.iters = 0;
  };
  int ttt = 0;
  PW_TRY(search(P[0], U[1], P[0]));
  record(ttt, std::exp(P[0][1]));
  while (ttt < 11) {
    ttt++;
    for (int i = 0; i < 2; i++) {
      PW_TRY(search(P[i], U[i], P[i + 1]));
      record(ttt, std::exp(P[i + 1][1]));
      ttt++;
    }
    U[0][0] = U[1][0];
    U[0][1] = U[1][1];
    U[1][0] = P[2][0] - P[0][0];
    U[1][1] = P[2][1] - P[0][1];
    if (!(std::fabs(U[1][0]) <= 0.0001 && std::fabs(U[1][1]) <= 0.0001) && ttt < 11) {
      PW_TRY(search(P[0], U[1], P[0]));
      record(ttt, std::exp(P[0][1]));
    } else {
      // final fit at P[0]; beta_init / coef0_init are whatever the last search left in the algorithm (:1221-1225)
      s->cur_rows = 0;
      s->sparsity_level = (int)P[0][0];
      s->lambda_level = std::exp(P[0][1]);
      PW_TRY(algorithm_fit(s));
      pt.beta = s->beta;
      pt.coef0 = s->coef0;
      PW_TRY(metric_train_loss(s, &pt.loss));
      PW_TRY(metric_ic(s, ic_type, is_cv, &pt.ic));
      record(ttt, std::exp(P[0][1]));
      all[ttt].iters = s->l;
      ttt++;
      size_t mi = 0;
      for (size_t k = 1; k < (size_t)ttt; k++)
        if (all[k].ic < all[mi].ic) mi = k;
      if (all[mi].ic == all[ttt - 1].ic) mi = ttt - 1;
      for (int k = 0; k < ttt; k++) store_candidate(s, res, all[k], false);
      store_best(s, res, all[mi], false);
      return 0;
    }
  }
  return fail(BESSX_ERR_NUMERIC, "powell end wrong (src/path.cpp:1298-1308)");
}
#undef PW_TRY

struct PgsArgs {
  double lmin, lmax;
  int powell_path, nlambda;
};

// the part of reset_path_caches() a path needs that continues on the caches of the previous call
int settle_device_chain(bessx_session *s) {
  kchains_quiesce(s);
  if (s->ahead.armed) {
    s->ahead.armed = false;
    HIPX(hipStreamSynchronize(s->st));
  }
  s->pend_on = false;
  s->hint.on = false;
  return 0;
}

int run_path(bessx_session *s, bool gs, const int *seq, int ns, const double *lam, int nl, int s_min,
                    int s_max, int ic_type, int is_cv, bessx_path_result *res, const PgsArgs *pgs,
                    bessx_path_chain *chain) {
  if (!s || !res) return fail(BESSX_ERR_ARG, "null session or result");
  if (is_cv && s->K < 2) return fail(BESSX_ERR_ARG, "is_cv needs bessx_session_set_cv first");
  HIPX(hipSetDevice(s->device));
  s->trace.clear();
  s->metric_depth = 0;
  for (auto &v : s->cv_init) v.clear();
  res->n_candidates = 0;
  s->n_fits = 0;
  s->n_iters = 0;
  auto t0 = std::chrono::steady_clock::now();
  // a path call starts cold, like a bessCpp call -- unless it continues the job of the previous call (chain->keep_caches:
  // the Gram columns and score sums in memory depend on the data only and stay valid)
  if (int rc0 = (chain && chain->keep_caches) ? settle_device_chain(s) : reset_path_caches(s)) return rc0;
  if (s->model_type == 1 && !(chain && chain->keep_caches) && s->path_group_xtx) {
    // group_XTX of the all-rows set INSIDE the path call, where the reference has it (src/path.cpp:37, :139, :588): X^T y
    // and diag(X^T X) in one pass over X.  The result is the one the session already holds (same kernels, same order of
    // summation), so a path pays the reference's pass without changing a bit; BESSX_TEST_HOOKS=path_group_xtx=0 leaves
    // it to session creation (rounds 1-5).  Device time of the last one: bessx_session_counter 19.
    // (its result is the one the session holds -- nothing is read back, the host goes on queueing the first fit while the
    // pass runs; with the kernel timing on, two events bracket it and counter 19 reads them when asked)
    bool timed = s->timing;
    for (int i = 0; i < 2 && timed; i++)
      if (!s->xtx_ev[i] && hipEventCreate(&s->xtx_ev[i]) != hipSuccess) {
        (void)hipGetLastError();
        s->xtx_ev[i] = nullptr;
        timed = false;
      }
    if (timed) (void)hipEventRecord(s->xtx_ev[0], s->st);
    const int rc_x = prepare_rowset(s, 0, true);
    if (timed && rc_x == 0 && hipEventRecord(s->xtx_ev[1], s->st) == hipSuccess) s->xtx_ev_pending = true;
    // (covariance form: everything that reads X^T y / diag(X^T X) next runs on this stream -- the coarse chain -- and
    // the chunk chains start behind a synchronisation of it.  Streaming form: chunk chains on streams of their own read
    // them at once, so the pass is waited for -- 0.6 ms in a path of 80)
    if (rc_x == 0 && !s->cov_mode) HIPX(hipStreamSynchronize(s->st));
    if (rc_x) return rc_x;
  }
  // Lead fits of a link (bessx_path_chain.lead_levels): a coarse warm-start chain on this session in front of the link,
  // which then starts from its last model on the cache the lead fits have filled
  bessx_path_chain lead_link;
  std::vector<int> lead_idx;
  std::vector<double> lead_val;
  bessx_path_chain *caller_chain = chain;
  if (chain && chain->lead_len > 0 && chain->lead_levels) {
    if (s->model_type != 1 || gs || pgs || is_cv || chain->stop_support)
      return fail(BESSX_ERR_UNSUPPORTED, "lead fits: LM sequential links without a stop table only");
    SparseVec init;
    for (int i = 0; i < chain->init_len; i++) {
      if (chain->init_idx[i] < 0 || chain->init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "chain: init index out of range");
      init.idx.push_back(chain->init_idx[i]);
      init.val.push_back(chain->init_val[i]);
    }
    double c0 = chain->init_coef0;
    int prev = 0;
    for (int m = 0; m < chain->lead_len; m++) {
      const int T = chain->lead_levels[m];
      if (T <= prev || T < 1 || T >= seq[0] || T > s->cap) return fail(BESSX_ERR_ARG, "lead fits: levels must ascend below the link's first");
      prev = T;
      if (int rc = run_fit(s, T, lam[0], init, c0)) return rc;
      init = s->beta;
      c0 = s->coef0;
    }
    if (int rc = settle_device_chain(s)) return rc;
    lead_idx = init.idx;
    lead_val = init.val;
    lead_link = *chain;
    lead_link.init_idx = lead_idx.data();
    lead_link.init_val = lead_val.data();
    lead_link.init_len = (int)lead_idx.size();
    lead_link.init_coef0 = c0;
    lead_link.keep_caches = 1;
    lead_link.lead_levels = nullptr;
    lead_link.lead_len = 0;
    chain = &lead_link;
    s->n_fits = 0;
    s->n_iters = 0;
  }
  // (chunk chains where the path qualifies and their contexts can be had, else the one chain: same candidates)
  const bool chunked = !pgs && !gs && kchunks_apply(s, seq, ns, nl, is_cv, chain) && kchunks_prepare(s, ns, chain != nullptr, chain && chain->init_len > 0 && chain->keep_caches) == 0;
  int rc = pgs  ? pgs_path(s, s_min, s_max, pgs->lmin, pgs->lmax, pgs->powell_path, pgs->nlambda, ic_type, is_cv, res)
           : gs ? gs_path(s, s_min, s_max, ic_type, is_cv, res)
           : chunked ? sequential_path_chunked(s, seq, ns, lam[0], ic_type, res, chain)
                     : sequential_path(s, seq, ns, lam, nl, ic_type, is_cv, res, chain);
  if (chain == &lead_link) {  // the link's outputs belong to the caller's structure
    caller_chain->stopped_at = lead_link.stopped_at;
    caller_chain->last_len = lead_link.last_len;
    caller_chain->last_coef0 = lead_link.last_coef0;
  }
  auto t1 = std::chrono::steady_clock::now();
  res->device_seconds = std::chrono::duration<double>(t1 - t0).count();
  res->n_fits = s->n_fits;
  res->n_pdas_iters = s->n_iters;
  return rc;
}


}  // namespace bessx

extern "C" {

int bessx_session_sequential_path(bessx_session *s, const int *sequence, int sequence_len, const double *lambda_seq,
                                  int lambda_len, int ic_type, int is_cv, bessx_path_result *res) {
  if (!sequence || sequence_len < 1 || !lambda_seq || lambda_len < 1)
    return fail(BESSX_ERR_ARG, "sequential_path: empty sequence");
  return run_path(s, false, sequence, sequence_len, lambda_seq, lambda_len, 0, 0, ic_type, is_cv, res);
}

int bessx_session_sequential_path_chain(bessx_session *s, const int *sequence, int sequence_len,
                                        const double *lambda_seq, int lambda_len, int ic_type, int is_cv,
                                        bessx_path_chain *chain, bessx_path_result *res) {
  if (!sequence || sequence_len < 1 || !lambda_seq || lambda_len < 1)
    return fail(BESSX_ERR_ARG, "sequential_path: empty sequence");
  if (!chain) return fail(BESSX_ERR_ARG, "sequential_path_chain: null chain");
  if (chain->init_len < 0 || (chain->init_len > 0 && (!chain->init_idx || !chain->init_val)))
    return fail(BESSX_ERR_ARG, "sequential_path_chain: bad initial model");
  if (is_cv && chain->init_len > 0)
    return fail(BESSX_ERR_UNSUPPORTED, "sequential_path_chain: under CV a chain would need the folds' models as well");
  if (chain->stop_support && (chain->stop_rows < 0 || chain->stop_row_len < 1))
    return fail(BESSX_ERR_ARG, "sequential_path_chain: bad stop table");
  return run_path(s, false, sequence, sequence_len, lambda_seq, lambda_len, 0, 0, ic_type, is_cv, res, nullptr, chain);
}

int bessx_session_gs_path(bessx_session *s, int s_min, int s_max, int ic_type, int is_cv, bessx_path_result *res) {
  if (s_min < 1 || s_max < s_min) return fail(BESSX_ERR_ARG, "gs_path: need 1 <= s_min <= s_max");
  return run_path(s, true, nullptr, 0, nullptr, 0, s_min, s_max, ic_type, is_cv, res);
}

int bessx_session_pgs_path(bessx_session *s, int s_min, int s_max, double lambda_min, double lambda_max, int n_lambda,
                           int powell_path, int ic_type, int is_cv, bessx_path_result *res) {
  if (s_min < 1 || s_max < s_min) return fail(BESSX_ERR_ARG, "pgs_path: need 1 <= s_min <= s_max");
  if (powell_path != 1 && powell_path != 2) return fail(BESSX_ERR_ARG, "pgs_path: powell_path must be 1 or 2");
  if (powell_path == 2 && n_lambda < 2) return fail(BESSX_ERR_ARG, "pgs_path: n_lambda must be >= 2");
  // bessCpp, src/bess.cpp:176-177
  PgsArgs a{std::log(std::max(lambda_min, 1e-5)), std::log(std::max(lambda_max, 1e-5)), powell_path, n_lambda};
  return run_path(s, true, nullptr, 0, nullptr, 0, s_min, s_max, ic_type, is_cv, res, &a);
}

int bessx_session_fit(bessx_session *s, int T0, double lambda, int fold, const int *init_idx, const double *init_val,
                      int init_len, double init_coef0, int *support, double *beta, double *coef0, int *iters,
                      double *train_loss, double *test_loss) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (fold >= s->K) return fail(BESSX_ERR_ARG, "fold index out of range");
  const int width = bessx_session_fit_width(s, T0);
  if (width < 0) return fail(BESSX_ERR_ARG, "sparsity level outside [1, number of groups]");
  HIPX(hipSetDevice(s->device));
  s->cur_rows = fold < 0 ? 0 : fold + 1;
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init.clear();
  for (int i = 0; i < init_len; i++) {
    if (init_idx[i] < 0 || init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "init index out of range");
    s->beta_init.idx.push_back(init_idx[i]);
    s->beta_init.val.push_back(init_val[i]);
  }
  s->coef0_init = init_coef0;
  if (int rc = algorithm_fit(s)) return rc;
  // (groups of size > 1: T0 counts groups, the fit returns the columns of the selected groups -- at most `width`)
  const int got = (int)s->beta.idx.size();
  if (got > width) return fail(BESSX_ERR_NUMERIC, "internal error: more columns selected than bessx_session_fit_width allows");
  for (int i = 0; i < width; i++) {
    if (support) support[i] = i < got ? s->beta.idx[i] : -1;
    if (beta) beta[i] = i < got ? s->beta.val[i] : 0.0;
  }
  if (coef0) *coef0 = s->coef0;
  if (iters) *iters = s->l;
  if (train_loss) *train_loss = metric_train_loss_value(s);
  if (test_loss) *test_loss = fold < 0 ? 0.0 : metric_fold_test_loss(s, fold);
  return BESSX_OK;
}

// One evaluation of a cross-validated candidate for a SUBSET of the folds (and, optionally, the full-data fit in front
// of them): what a rank of a fold-sharded path owns (bess_amd/dist.py).  The fold fits take the library's own route --
// the chains side by side with union fills where that applies, one after another on the session's state otherwise --
// and the session's own per-fold warm starts (Metric::cv_initial_model_param).
int bessx_session_cv_eval(bessx_session *s, int T0, double lambda, int want_full, const int *init_idx,
                          const double *init_val, int init_len, double init_coef0, const int *folds, int n_folds,
                          int *support, double *beta, double *coef0, int *iters, double *train_loss, double *test_loss) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (n_folds < 0 || (n_folds > 0 && !folds)) return fail(BESSX_ERR_ARG, "cv_eval: bad fold list");
  if (n_folds > 0 && s->K < 2) return fail(BESSX_ERR_ARG, "cv_eval needs bessx_session_set_cv first");
  for (int i = 0; i < n_folds; i++)
    if (folds[i] < 0 || folds[i] >= s->K || (i > 0 && folds[i] <= folds[i - 1]))
      return fail(BESSX_ERR_ARG, "cv_eval: folds must be ascending indices in [0, K)");
  const int width = bessx_session_fit_width(s, T0);
  if (width < 0) return fail(BESSX_ERR_ARG, "sparsity level outside [1, number of groups]");
  HIPX(hipSetDevice(s->device));
  s->sparsity_level = T0;
  s->lambda_level = lambda;
  s->beta_init.clear();
  for (int i = 0; i < init_len; i++) {
    if (init_idx[i] < 0 || init_idx[i] >= s->p) return fail(BESSX_ERR_ARG, "init index out of range");
    s->beta_init.idx.push_back(init_idx[i]);
    s->beta_init.val.push_back(init_val[i]);
  }
  s->coef0_init = init_coef0;
  int rec = 0;
  auto put = [&](const SparseVec &b, double c0, int l, double tr, double te) -> int {
    const int got = (int)b.idx.size();
    if (got > width) return fail(BESSX_ERR_NUMERIC, "internal error: more columns selected than bessx_session_fit_width allows");
    for (int i = 0; i < width; i++) {
      if (support) support[(size_t)rec * width + i] = i < got ? b.idx[i] : -1;
      if (beta) beta[(size_t)rec * width + i] = i < got ? b.val[i] : 0.0;
    }
    if (coef0) coef0[rec] = c0;
    if (iters) iters[rec] = l;
    if (train_loss) train_loss[rec] = tr;
    if (test_loss) test_loss[rec] = te;
    rec++;
    return 0;
  };
  if (want_full) {
    s->cur_rows = 0;
    if (int rc = algorithm_fit(s)) return rc;
    if (int rc = put(s->beta, s->coef0, s->l, metric_train_loss_value(s), 0.0)) return rc;
  }
  if (n_folds == 0) return BESSX_OK;
  const std::vector<int> only(folds, folds + n_folds);
  s->metric_depth++;
  int rc = 0;
  if (side_by_side_applies(s, T0)) {
    std::vector<double> tl((size_t)n_folds);
    double mean = 0.0;
    rc = fold_fits_side_by_side(s, &mean, &only, tl.data());
    for (int i = 0; i < n_folds && rc == 0; i++) {
      const bessx_session *c = s->fold_ctx[(size_t)only[i]];
      rc = put(c->beta, c->coef0, c->l, metric_train_loss_value(c), tl[i]);
    }
  } else {
    const SparseVec keep = s->beta_init;
    for (int i = 0; i < n_folds && rc == 0; i++) {
      const int k = only[i];
      s->beta_init = s->warm_start ? s->cv_init[k] : keep;  // update_beta_init(cv_initial_model_param.row(k))
      s->cur_rows = k + 1;                                  // update_train_mask + update_group_XTX
      rc = algorithm_fit(s);
      if (rc) break;
      if (s->warm_start) s->cv_init[k] = s->beta;
      rc = put(s->beta, s->coef0, s->l, metric_train_loss_value(s), metric_fold_test_loss(s, k));
    }
  }
  s->metric_depth--;
  return rc;
}

// ----------------------------------------------------------------------------------------------
// Cooperative prefill of the Gram column cache (LM, covariance form, all rows): the ranks of a k-path run share the
// passes over X that every chunk's cold start would otherwise repeat (bess_amd/dist.py, cooperative_prefill).  All ranks
// list the same columns -- slots are handed out in list order on a cache started over, so slot numbers agree across
// ranks -- each forms its share of the 32-column groups, the p x 32 blocks travel (RCCL all-gather), every rank
// imports the others' and fills the slot-indexed Gram once.  Cache contents only: no result depends on it.
// ----------------------------------------------------------------------------------------------
static int prefill_ready(bessx_session *s) {
  if (!s) return fail(BESSX_ERR_ARG, "null session");
  if (!s->cov_mode || s->model_type != 1)
    return fail(BESSX_ERR_UNSUPPORTED, "cov_prefill: the session does not run the covariance form of the LM score pass");
  if (s->cv_shared) return fail(BESSX_ERR_UNSUPPORTED, "cov_prefill: not offered on sessions with cross-validation folds");
  HIPX(hipSetDevice(s->device));
  if (!s->fill_ctrl) {
    HIPX(hipMalloc(reinterpret_cast<void **>(&s->fill_ctrl), sizeof(FitCtrl)));
    HIPX(hipMemset(s->fill_ctrl, 0, sizeof(FitCtrl)));
    HIPX(hipHostMalloc(reinterpret_cast<void **>(&s->fill_ctrl_h), sizeof(FitCtrl)));
  }
  return 0;
}

int bessx_session_marginal_scores(bessx_session *s, double *bd) {
  if (!s || !bd) return fail(BESSX_ERR_ARG, "null argument");
  if (s->model_type != 1 || s->grouped) return fail(BESSX_ERR_UNSUPPORTED, "marginal_scores: LM with singleton groups");
  HIPX(hipSetDevice(s->device));
  // get_A at beta = 0, lambda = 0 (src/Algorithm.h:1109-1123): d_j = x_j . y / n, bd_j = (d_j / phi_j)^2, phi_j^2 = x_j . x_j / n
  std::vector<double> xty((size_t)s->p), xtx((size_t)s->p);
  HIPX(hipStreamSynchronize(s->st));
  HIPX(hipMemcpy(xty.data(), s->xty[0], xty.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPX(hipMemcpy(xtx.data(), s->xtx[0], xtx.size() * sizeof(double), hipMemcpyDeviceToHost));
  const double n = (double)s->n_train[0];
  for (int j = 0; j < s->p; j++) {
    const double phi = std::sqrt(xtx[j] / n), d = xty[j] / n;
    const double t = d * (1.0 / phi);
    bd[j] = t * t;
  }
  return BESSX_OK;
}

}  // extern "C"
namespace bessx {
// append = 0: the cache is started over and slot i goes to cols[i]; 1: the columns (all of them uncached) take the next
// free slots of the cache as it is -- the same slots on every rank whose session has done the same work so far
// 2: like 1 from inside a parked fit (the shared wide fills): the fit's host-side state stays as it is
int prefill_begin(bessx_session *s, const int *cols, int ncols, int append) {
  if (int rc = prefill_ready(s)) return rc;
  if (!cols || ncols < 1 || ncols % COV_R != 0) return fail(BESSX_ERR_ARG, "cov_prefill: need a multiple of 32 columns");
  std::vector<char> seen((size_t)s->p, 0);
  for (int i = 0; i < ncols; i++) {
    if (cols[i] < 0 || cols[i] >= s->p || seen[(size_t)cols[i]]) return fail(BESSX_ERR_ARG, "cov_prefill: bad column list");
    seen[(size_t)cols[i]] = 1;
  }
  int base = 0;
  if (append) {
    if (append == 1)
      if (int rc = settle_device_chain(s)) return rc;
    int meta_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPX(hipMemcpyAsync(meta_h, s->cov[0].meta, sizeof(meta_h), hipMemcpyDeviceToHost, s->st));
    HIPX(hipStreamSynchronize(s->st));
    base = meta_h[0];
    if (append == 1) {
      for (auto &c : s->cache) c.valid = c.model_only = false;  // (a fit that follows starts from uploaded coefficients)
      s->dev_state_rs = -1;
    }
  } else if (int rc = reset_path_caches(s)) {
    return rc;
  }
  if (base + ncols + s->cov_spec + COV_R > cov_C_dev(s) || ncols > s->capA)
    return fail(BESSX_ERR_ARG, "cov_prefill: the list does not fit the Gram column cache");
  int *st_idx = reinterpret_cast<int *>(s->stage_h);
  std::copy(cols, cols + ncols, st_idx);
  HIPX(hipMemcpyAsync(s->init_idx_d, st_idx, (size_t)ncols * sizeof(int), hipMemcpyHostToDevice, s->st));
  CovUnion u = {};
  u.nf = 1;
  u.list[0] = s->init_idx_d;
  u.len[0] = ncols;
  HIPX(launch_cov_fill_union(u, append ? 0 : 1, nullptr, nullptr, s->cov_spec, 0, s->cov[0].slot_of, s->cov[0].meta, s->p,
                             s->cov_fcols, s->fill_ctrl, s->st));
  HIPX(hipMemcpyAsync(s->fill_ctrl_h, s->fill_ctrl, sizeof(FitCtrl), hipMemcpyDeviceToHost, s->st));
  HIPX(hipStreamSynchronize(s->st));  // (the staging buffer is free again)
  if (s->fill_ctrl_h->cov_nfill != ncols)
    return fail(BESSX_ERR_ARG, "cov_prefill_extend: a listed column is cached already (the slots would not line up across ranks)");
  s->prefill_cols = ncols;
  s->prefill_base = base;
  return BESSX_OK;
}
}  // namespace bessx
extern "C" {

int bessx_session_cov_prefill_begin(bessx_session *s, const int *cols, int ncols) { return prefill_begin(s, cols, ncols, 0); }
int bessx_session_cov_prefill_extend(bessx_session *s, const int *cols, int ncols) { return prefill_begin(s, cols, ncols, 1); }

// The sacrifice scores of the last PDAS iteration of the last fit (bd) and the slot of every column in the Gram column
// cache of the all-rows row set (-1: not cached): what a caller needs to pick the next columns worth caching.
int bessx_session_cov_state(bessx_session *s, double *bd, int *slot_of) {
  if (int rc = prefill_ready(s)) return rc;
  HIPX(hipStreamSynchronize(s->st));
  if (bd) HIPX(hipMemcpy(bd, s->bd, (size_t)s->p * sizeof(double), hipMemcpyDeviceToHost));
  if (slot_of) HIPX(hipMemcpy(slot_of, s->cov[0].slot_of, (size_t)s->p * sizeof(int), hipMemcpyDeviceToHost));
  return BESSX_OK;
}

int bessx_session_cov_prefill_compute(bessx_session *s, int g0, int ngroups) {
  if (int rc = prefill_ready(s)) return rc;
  if (g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  if (ngroups == 0) return BESSX_OK;
  if (int rc = enqueue_cov_fill(s, 0, ngroups, 1, s->fill_ctrl, g0, false)) return rc;
  s->cov_panel_groups += ngroups;
  HIPX(hipStreamSynchronize(s->st));
  return cov_collect(s, s->prefill_cols);
}

int bessx_session_cov_prefill_export(bessx_session *s, int g0, int ngroups, double *dst, int dst_on_device) {
  if (int rc = prefill_ready(s)) return rc;
  if (!dst || g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  const size_t cnt = (size_t)ngroups * COV_R * s->p;
  HIPX(hipMemcpyAsync(dst, s->cov[0].G + ((size_t)s->prefill_base + (size_t)g0 * COV_R) * s->p, cnt * sizeof(double),
                      dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->st));
  HIPX(hipStreamSynchronize(s->st));
  return BESSX_OK;
}

int bessx_session_cov_prefill_import(bessx_session *s, int g0, int ngroups, const double *src, int src_on_device) {
  if (int rc = prefill_ready(s)) return rc;
  if (!src || g0 < 0 || ngroups < 0 || (g0 + ngroups) * COV_R > s->prefill_cols) return fail(BESSX_ERR_ARG, "cov_prefill: group range");
  const size_t cnt = (size_t)ngroups * COV_R * s->p;
  HIPX(hipMemcpyAsync(s->cov[0].G + ((size_t)s->prefill_base + (size_t)g0 * COV_R) * s->p, src, cnt * sizeof(double),
                      src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st));
  HIPX(hipStreamSynchronize(s->st));  // (the caller's buffer is free again when the call returns)
  return BESSX_OK;
}

int bessx_session_cov_prefill_end(bessx_session *s) {
  if (int rc = prefill_ready(s)) return rc;
  if (s->prefill_cols < COV_R) return fail(BESSX_ERR_ARG, "cov_prefill_end without cov_prefill_begin");
  bessx_session::CovCache &cv = s->cov[0];
  HIPX(launch_cov_compact(cv.G, s->p, cv.slot_of, s->cov_fcols, 0, s->prefill_cols / COV_R, cv.GS, s->cov_cs, s->fill_ctrl, 1,
                          s->st, s->xtx[0], cv.meta));
  HIPX(hipStreamSynchronize(s->st));
  s->prefill_cols = 0;
  return BESSX_OK;
}

int bessx_session_set_kpath_chains(bessx_session *s, int chains) {
  if (!s || chains < 0 || chains > 8) return fail(BESSX_ERR_ARG, "set_kpath_chains: 0 (automatic) .. 8");
  s->kpath_chains = chains;
  s->kch_auto_off = false;  // (an explicit choice, 0 included, is tried afresh)
  return BESSX_OK;
}

int bessx_session_set_fill_hook(bessx_session *s, bessx_fill_hook hook, void *user, int width) {
  if (int rc = prefill_ready(s)) return rc;
  if (hook && (width < COV_R || width % COV_R != 0 || width > s->capA))
    return fail(BESSX_ERR_ARG, "set_fill_hook: width must be a multiple of 32 within the session's sparsity capacity");
  s->fill_hook = hook;
  s->fill_hook_user = user;
  s->fill_hook_width = hook ? width : 0;
  return BESSX_OK;
}

static void debug_sleep_cb(void *us) {
  std::this_thread::sleep_for(std::chrono::microseconds((long)(intptr_t)us));
}

// (milliseconds < 0: -milliseconds MICROseconds -- windows of the length of a fit, tests/test_cv_shard_gpu.py)
int bessx_session_debug_block_stream(bessx_session *s, int milliseconds) {
  if (!s || milliseconds < -1000000) return fail(BESSX_ERR_ARG, "bad argument");
  HIPX(hipSetDevice(s->device));
  const long us = milliseconds >= 0 ? 1000L * milliseconds : -(long)milliseconds;
  HIPX(hipLaunchHostFunc(s->st, debug_sleep_cb, reinterpret_cast<void *>((intptr_t)us)));
  return BESSX_OK;
}


}  // extern "C"
