/* oracle/bess_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C99 restatement of the PDAS hot path of Mamba413/bess.  Every function cites
 * the reference lines (under /root/reference) whose arithmetic it follows.  Nothing
 * here is copied: the reference is Eigen/C++ with dense matrix temporaries, this file
 * is scalar loops over the active columns only.  Summation order therefore differs
 * from Eigen's, so coefficients agree to ~1e-12 relative, not bit for bit; selected
 * supports, iteration counts and path decisions agree exactly on the pinned cases.
 *
 * Parity status: PINNED -- checked against the compiled reference in
 * tests/test_oracle_vs_reference.py (when oracle/_ref/libbess_ref.so is present) and
 * against committed golden vectors in tests/test_oracle_golden.py.
 *
 * Scope: singleton groups only (g_index = 0..p-1), model_type 1 (LM), 2 (logistic),
 * 3 (Poisson), 4 (Cox); path_type 1 (sequential, with the lambda snake) and 2 (golden
 * section).  Group sizes > 1, screening and the Powell path are out of scope
 * (SURVEY.md section 8f).
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime */
#include "bess_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* wall time of the last bess_oracle_run*: set-up (copy, normalise, group_XTX) and the path itself -- for bench.py's
 * cpu_baseline leg (kind "port"), which reports them apart like the GPU figure excludes upload + normalise */
static double g_last_setup_s = 0.0, g_last_path_s = 0.0;
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void bess_oracle_last_timing(double *setup_s, double *path_s) {
  if (setup_s) *setup_s = g_last_setup_s;
  if (path_s) *path_s = g_last_path_s;
}

/* ------------------------------------------------------------------ trace */

typedef struct {
  int *v;
  int n, cap;
} ivec;
typedef struct {
  double *v;
  int n, cap;
} dvec;

static void ipush(ivec *a, int x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? 2 * a->cap : 256;
    a->v = (int *)realloc(a->v, (size_t)a->cap * sizeof(int));
  }
  a->v[a->n++] = x;
}
static void dpush(dvec *a, double x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? 2 * a->cap : 256;
    a->v = (double *)realloc(a->v, (size_t)a->cap * sizeof(double));
  }
  a->v[a->n++] = x;
}

static ivec t_meta, t_a;
static dvec t_beta, t_coef0, t_loss, t_ic;

int bess_oracle_trace_size(int which) {
  switch (which) {
    case 0: return t_meta.n;
    case 1: return t_a.n;
    case 2: return t_beta.n;
    case 3: return t_coef0.n;
    case 4: return t_loss.n;
    case 5: return t_ic.n;
  }
  return -1;
}
void bess_oracle_trace_copy_int(int which, int *out) {
  const ivec *a = which == 0 ? &t_meta : &t_a;
  if (a->n) memcpy(out, a->v, (size_t)a->n * sizeof(int));
}
void bess_oracle_trace_copy_double(int which, double *out) {
  const dvec *a = &t_beta;
  if (which == 3) a = &t_coef0;
  if (which == 4) a = &t_loss;
  if (which == 5) a = &t_ic;
  if (a->n) memcpy(out, a->v, (size_t)a->n * sizeof(double));
}

/* ------------------------------------------------------------------ data */

typedef struct {
  int n, p;
  double *x; /* column-major n x p, normalised in place */
  double *y, *w;
  double *x_mean, *x_norm;
  double y_mean;
  int data_type, is_normal;
  /* groups (Data::g_index / g_size / g_num, src/Data.h:59-67): group g owns columns gidx[g] .. gidx[g]+gsz[g]-1 */
  int N, gmax;
  int *gidx, *gsz, *goff; /* goff: offset of the g x g block of group g in block arrays */
} odata;

#define XC(d, j) ((d)->x + (size_t)(j) * (size_t)(d)->n)

/* Normalize / Normalize3 / Normalize4, src/normalize.cpp:20-85, dispatched as in
 * Data::normalize, src/Data.h:79-93. */
static void data_normalize(odata *d) {
  int n = d->n, p = d->p, i, j;
  double sn = sqrt((double)n);
  if (d->data_type == 1 || d->data_type == 2) {
    for (j = 0; j < p; j++) {
      double s = 0.0, *c = XC(d, j);
      for (i = 0; i < n; i++) s += d->w[i] * c[i];
      d->x_mean[j] = s / (double)n;
    }
    if (d->data_type == 1) {
      double s = 0.0;
      for (i = 0; i < n; i++) s += d->y[i] * d->w[i];
      d->y_mean = s / (double)n;
    }
    for (j = 0; j < p; j++) {
      double *c = XC(d, j), m = d->x_mean[j];
      for (i = 0; i < n; i++) c[i] = c[i] - m;
    }
    if (d->data_type == 1)
      for (i = 0; i < n; i++) d->y[i] = d->y[i] - d->y_mean;
  }
  for (j = 0; j < p; j++) {
    double s = 0.0, *c = XC(d, j);
    for (i = 0; i < n; i++) s += d->w[i] * (c[i] * c[i]);
    d->x_norm[j] = sqrt(s);
  }
  for (j = 0; j < p; j++) {
    double *c = XC(d, j), nm = d->x_norm[j];
    for (i = 0; i < n; i++) c[i] = sn * c[i] / nm;
  }
}

/* Data::add_weight, src/Data.h:70-77 (LM only, src/bess.cpp:97). */
static void data_add_weight(odata *d) {
  int i, j;
  for (i = 0; i < d->n; i++) {
    double s = sqrt(d->w[i]);
    for (j = 0; j < d->p; j++) XC(d, j)[i] = XC(d, j)[i] * s;
    d->y[i] = d->y[i] * s;
  }
}

/* Data::normalize + add_weight on a column-major copy, exported so that the device's normalisation kernels can be
 * checked against the same restatement the paths use (tests/test_ops_gpu.py).  x: n x p column-major, in place. */
int bess_oracle_normalize(double *x, int n, int p, double *y, const double *weight, int data_type, int is_normal,
                          int add_weight, double *x_mean, double *x_norm, double *y_mean) {
  odata d;
  int j;
  if (!x || !y || !weight || n < 1 || p < 1) return 1;
  memset(&d, 0, sizeof(d));
  d.n = n;
  d.p = p;
  d.x = x;
  d.y = y;
  d.w = (double *)weight;
  d.x_mean = x_mean;
  d.x_norm = x_norm;
  d.data_type = data_type;
  d.is_normal = is_normal;
  for (j = 0; j < p; j++) {
    x_mean[j] = 0.0;
    x_norm[j] = 0.0;
  }
  if (is_normal) data_normalize(&d);
  if (add_weight) data_add_weight(&d);
  *y_mean = d.y_mean;
  return 0;
}

/* ------------------------------------------------------------------ small linear algebra */

int bess_oracle_sym_solve(const double *a, int k, const double *b, double *x) {
  /* un-pivoted LDL^T of the lower triangle; stands in for Eigen's
   * ColPivHouseholderQR (src/Algorithm.h:1134) and LDLT (:1171,1199,1299,1473) solves
   * of definite k x k systems. */
  int i, j, m, rc = 0;
  double *l = (double *)malloc((size_t)k * (size_t)k * sizeof(double));
  double *dg = (double *)malloc((size_t)k * sizeof(double));
  for (j = 0; j < k; j++) {
    double dj = a[(size_t)j * k + j];
    for (m = 0; m < j; m++) dj -= l[(size_t)m * k + j] * l[(size_t)m * k + j] * dg[m];
    dg[j] = dj;
    if (dj == 0.0) {
      rc = 1;
      dj = DBL_MIN;
    }
    for (i = j + 1; i < k; i++) {
      double s = a[(size_t)j * k + i];
      for (m = 0; m < j; m++) s -= l[(size_t)m * k + i] * l[(size_t)m * k + j] * dg[m];
      l[(size_t)j * k + i] = s / dj;
    }
  }
  for (i = 0; i < k; i++) {
    double s = b[i];
    for (m = 0; m < i; m++) s -= l[(size_t)m * k + i] * x[m];
    x[i] = s;
  }
  for (i = 0; i < k; i++) x[i] = x[i] / dg[i];
  for (i = k - 1; i >= 0; i--) {
    double s = x[i];
    for (m = i + 1; m < k; m++) s -= l[(size_t)i * k + m] * x[m];
    x[i] = s;
  }
  free(l);
  free(dg);
  return rc;
}

static int int_cmp(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* std::nth_element as libstdc++ (GCC 11, bits/stl_algo.h: __introselect, __unguarded_partition_pivot,
 * __move_median_to_first, __unguarded_partition, __insertion_sort) runs it on the index array 0..len-1 with the
 * comparator of max_k, comp(i, j) = vec(i) > vec(j) (src/utilities.cpp:179-188).  With distinct scores any selection
 * returns the same set; with EQUAL scores at the boundary (duplicated columns, 0/1 designs) which of the tied indices
 * land in the first k positions is decided by exactly these moves, so they are restated step by step.  The reference's
 * toolchain is the pinned dependency here: the golden vectors and oracle/_ref are built with g++ 11.4 / libstdc++.
 * The heap-select branch (depth limit 2 floor(log2 len) exhausted) is restated too: small ranges reach it. */
/* (per thread: bess_oracle_max_k alone may be called from several threads; the path functions may not -- their trace
 * buffers are file-scope) */
static __thread const double *nth_vec;
static int nth_comp(int i, int j) { return nth_vec[i] > nth_vec[j]; }
static void nth_swap(int *a, int *b) {
  int t = *a;
  *a = *b;
  *b = t;
}
static void nth_move_median_to_first(int *result, int *a, int *b, int *c) {
  if (nth_comp(*a, *b)) {
    if (nth_comp(*b, *c)) nth_swap(result, b);
    else if (nth_comp(*a, *c)) nth_swap(result, c);
    else nth_swap(result, a);
  } else if (nth_comp(*a, *c)) nth_swap(result, a);
  else if (nth_comp(*b, *c)) nth_swap(result, c);
  else nth_swap(result, b);
}
static int *nth_unguarded_partition(int *first, int *last, int *pivot) {
  for (;;) {
    while (nth_comp(*first, *pivot)) ++first;
    --last;
    while (nth_comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    nth_swap(first, last);
    ++first;
  }
}
static void nth_insertion_sort(int *first, int *last) {
  int *i;
  if (first == last) return;
  for (i = first + 1; i != last; ++i) {
    int val = *i;
    if (nth_comp(val, *first)) {
      memmove(first + 1, first, (size_t)(i - first) * sizeof(int));
      *first = val;
    } else {
      int *cur = i, *next = i - 1;
      while (nth_comp(val, *next)) {
        *cur = *next;
        cur = next;
        --next;
      }
      *cur = val;
    }
  }
}
/* std::__adjust_heap / __push_heap / __make_heap / __pop_heap / __heap_select of libstdc++ (bits/stl_heap.h,
 * bits/stl_algo.h), on the index array with the same comparator: the branch introselect takes when its depth limit is
 * exhausted (small ranges with unlucky pivots reach it: 2 floor(log2 len) partitions may shrink a range by one each). */
static void nth_adjust_heap(int *first, long hole, long len, int value) {
  const long top = hole;
  long child = hole, parent;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (nth_comp(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  parent = (hole - 1) / 2; /* __push_heap */
  while (hole > top && nth_comp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void nth_heap_select(int *first, int *middle, int *last) {
  const long len = middle - first;
  int *i;
  if (len >= 2) { /* __make_heap */
    long parent = (len - 2) / 2;
    for (;;) {
      nth_adjust_heap(first, parent, len, first[parent]);
      if (parent == 0) break;
      parent--;
    }
  }
  for (i = middle; i < last; ++i)
    if (nth_comp(*i, *first)) { /* __pop_heap(first, middle, i) */
      const int value = *i;
      *i = *first;
      nth_adjust_heap(first, 0, len, value);
    }
}
static long nth_heap_selects = 0; /* how often the branch was taken (tests want to know that they reached it) */
long bess_oracle_nth_heap_selects(void) { return nth_heap_selects; }
static void nth_element_libstdcxx(int *first, int *nth, int *last) {
  long depth_limit;
  long len = last - first;
  int lg = 0;
  if (first == last || nth == last) return;
  while ((len >> (lg + 1)) > 0) lg++; /* std::__lg */
  depth_limit = 2L * lg;
  while (last - first > 3) {
    int *mid, *cut;
    if (depth_limit == 0) {
      nth_heap_selects++;
      nth_heap_select(first, nth + 1, last);
      nth_swap(first, nth); /* "place the nth largest element in its final position" */
      return;
    }
    --depth_limit;
    mid = first + (last - first) / 2;
    nth_move_median_to_first(first, first + 1, mid, last - 1);
    cut = nth_unguarded_partition(first + 1, last, first);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  nth_insertion_sort(first, last);
}

/* test hook: the score vector (and k) of the last max_k call that took the heap-select branch */
static double nth_last_hs_scores[4096];
static int nth_last_hs_len = 0, nth_last_hs_k = 0;
int bess_oracle_last_heap_select(double *scores, int cap, int *k) {
  int i;
  for (i = 0; i < nth_last_hs_len && i < cap; i++) scores[i] = nth_last_hs_scores[i];
  *k = nth_last_hs_k;
  return nth_last_hs_len;
}

void bess_oracle_max_k(const double *score, int len, int k, int *out) {
  /* max_k, src/utilities.cpp:179-188: nth_element(ind, ind + k, ind + len, vec(i) > vec(j)); sort(ind, ind + k) */
  int i;
  int *ind = (int *)malloc((size_t)len * sizeof(int));
  const long hs_before = nth_heap_selects;
  for (i = 0; i < len; i++) ind[i] = i;
  nth_vec = score;
  nth_element_libstdcxx(ind, ind + k, ind + len);
  if (nth_heap_selects != hs_before && len <= 4096) {
    memcpy(nth_last_hs_scores, score, (size_t)len * sizeof(double));
    nth_last_hs_len = len;
    nth_last_hs_k = k;
  }
  for (i = 0; i < k; i++) out[i] = ind[i];
  qsort(out, (size_t)k, sizeof(int), int_cmp);
  free(ind);
}

/* ------------------------------------------------------------------ Algorithm */

typedef struct {
  odata *d;
  int model_type, algorithm_type, max_iter, warm_start;
  int T0;
  double lambda;
  const int *rows; /* train mask (sorted row indices) */
  int n_rows;
  double *beta, *beta_init;
  double coef0, coef0_init;
  int l;
  const double *xtx; /* group_XTX of the current training rows (LM) */
  const int *always;
  int n_always;
} oalg;

static double clamp30(double v) {
  if (v > 30.0) return 30.0;
  if (v < -30.0) return -30.0;
  return v;
}

/* eta_i = sum_j x_ij beta_j over the non-zero beta_j, for i in rows */
static void lin_pred(const odata *d, const double *beta, const int *rows, int n_rows, double *eta) {
  int i, j;
  for (i = 0; i < n_rows; i++) eta[i] = 0.0;
  for (j = 0; j < d->p; j++) {
    if (beta[j] != 0.0) {
      const double *c = d->x + (size_t)j * (size_t)d->n;
      double b = beta[j];
      for (i = 0; i < n_rows; i++) eta[i] += c[rows[i]] * b;
    }
  }
}

static void select_top(oalg *a, double *bd, int *A) {
  /* slice_assignment + max_k, src/utilities.cpp:179-199 */
  int i;
  for (i = 0; i < a->n_always; i++) bd[a->always[i]] = DBL_MAX;
  bess_oracle_max_k(bd, a->d->N, a->T0, A);
}

/* symmetric eigen-decomposition by cyclic Jacobi: a (s x s, column-major) -> eigenvalues ev, eigenvectors v */
static void jacobi_eig(double *a, int s, double *ev, double *v) {
  int i, j, k, sweep;
  for (i = 0; i < s; i++)
    for (j = 0; j < s; j++) v[(size_t)j * s + i] = i == j ? 1.0 : 0.0;
  for (sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, dg = 0.0;
    for (i = 0; i < s; i++)
      for (j = 0; j < s; j++) {
        if (i != j) off += a[(size_t)j * s + i] * a[(size_t)j * s + i];
        else dg += a[(size_t)j * s + i] * a[(size_t)j * s + i];
      }
    if (off <= 1e-32 * dg || off == 0.0) break;
    for (i = 0; i < s - 1; i++)
      for (j = i + 1; j < s; j++) {
        double apq = a[(size_t)j * s + i], app, aqq, theta, t, c, sn;
        if (apq == 0.0) continue;
        app = a[(size_t)i * s + i];
        aqq = a[(size_t)j * s + j];
        theta = (aqq - app) / (2.0 * apq);
        t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        c = 1.0 / sqrt(t * t + 1.0);
        sn = t * c;
        for (k = 0; k < s; k++) {
          double akp = a[(size_t)i * s + k], akq = a[(size_t)j * s + k];
          a[(size_t)i * s + k] = c * akp - sn * akq;
          a[(size_t)j * s + k] = sn * akp + c * akq;
        }
        for (k = 0; k < s; k++) {
          double apk = a[(size_t)k * s + i], aqk = a[(size_t)k * s + j];
          a[(size_t)k * s + i] = c * apk - sn * aqk;
          a[(size_t)k * s + j] = sn * apk + c * aqk;
        }
        for (k = 0; k < s; k++) {
          double vkp = v[(size_t)i * s + k], vkq = v[(size_t)j * s + k];
          v[(size_t)i * s + k] = c * vkp - sn * vkq;
          v[(size_t)j * s + k] = sn * vkp + c * vkq;
        }
      }
  }
  for (i = 0; i < s; i++) ev[i] = a[(size_t)i * s + i];
}

/* Per-group sacrifice bd_g = || Phi_g beta_g + Phi_g^{-1} d_g ||^2 / size(g), Phi_g = sqrtm(M_g)
 * (src/Algorithm.h:1112-1123, :1238-1257; Phi / invPhi: src/utilities.cpp:142-151, 167-177).
 * mblk holds the symmetric blocks M_g (column-major, at goff[g]); dcol the per-column d. */
static void group_scores(const oalg *a, const double *mblk, const double *dcol, double *bd) {
  const odata *d = a->d;
  int g, i, j, k, gm = d->gmax;
  double *m = (double *)malloc((size_t)gm * gm * sizeof(double)), *v = (double *)malloc((size_t)gm * gm * sizeof(double));
  double *ev = (double *)malloc((size_t)gm * sizeof(double)), *t = (double *)malloc((size_t)gm * sizeof(double));
  for (g = 0; g < d->N; g++) {
    int s = d->gsz[g], c0 = d->gidx[g];
    if (s == 1) {
      double phi = sqrt(mblk[d->goff[g]]), inv = 1.0 / phi, tt = phi * a->beta[c0] + inv * dcol[c0];
      bd[g] = tt * tt;
      continue;
    }
    memcpy(m, mblk + d->goff[g], (size_t)s * s * sizeof(double));
    jacobi_eig(m, s, ev, v);
    for (i = 0; i < s; i++) t[i] = 0.0;
    for (k = 0; k < s; k++) {
      /* component of beta_g and d_g along eigenvector k */
      double pb = 0.0, pd = 0.0, sq = sqrt(ev[k]), coef;
      for (j = 0; j < s; j++) {
        pb += v[(size_t)k * s + j] * a->beta[c0 + j];
        pd += v[(size_t)k * s + j] * dcol[c0 + j];
      }
      coef = sq * pb + pd / sq;
      for (i = 0; i < s; i++) t[i] += v[(size_t)k * s + i] * coef;
    }
    {
      double ss = 0.0;
      for (i = 0; i < s; i++) ss += t[i] * t[i];
      bd[g] = ss / (double)s;
    }
  }
  free(m);
  free(v);
  free(ev);
  free(t);
}

/* find_ind, src/utilities.cpp:113-130: groups -> columns (all p columns when every group is selected) */
static int expand_groups(const odata *d, const int *A, int T0, int *cols) {
  int i, j, k = 0;
  if (T0 == d->N) {
    for (j = 0; j < d->p; j++) cols[j] = j;
    return d->p;
  }
  for (i = 0; i < T0; i++)
    for (j = 0; j < d->gsz[A[i]]; j++) cols[k++] = d->gidx[A[i]] + j;
  return k;
}

/* GroupPdasLm::get_A, src/Algorithm.h:1097-1129 with Phi / invPhi for 1x1 groups,
 * src/utilities.cpp:142-151, 167-177. */
static void lm_get_A(oalg *a, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, p = d->p, i, j, g, nb = d->goff[d->N];
  double *r = (double *)malloc((size_t)nt * sizeof(double));
  double *dc = (double *)malloc((size_t)p * sizeof(double));
  double *bd = (double *)malloc((size_t)d->N * sizeof(double));
  double *mb = (double *)malloc((size_t)nb * sizeof(double));
  lin_pred(d, a->beta, a->rows, nt, r);
  for (i = 0; i < nt; i++) r[i] = d->y[a->rows[i]] - r[i] - a->coef0;
  for (j = 0; j < p; j++) {
    const double *c = d->x + (size_t)j * (size_t)d->n;
    double s = 0.0;
    for (i = 0; i < nt; i++) s += c[a->rows[i]] * r[i];
    dc[j] = s / (double)nt - 2.0 * a->lambda * a->beta[j];
  }
  /* Phi_g^2 = 2 lambda I + X_g^T X_g / n (src/utilities.cpp:147) */
  for (g = 0; g < d->N; g++) {
    int sz = d->gsz[g];
    for (i = 0; i < sz * sz; i++) mb[d->goff[g] + i] = a->xtx[d->goff[g] + i] / (double)nt;
    for (i = 0; i < sz; i++) mb[d->goff[g] + i * sz + i] += 2.0 * a->lambda;
  }
  group_scores(a, mb, dc, bd);
  select_top(a, bd, A);
  free(r);
  free(dc);
  free(bd);
  free(mb);
}

/* GroupPdasLm::primary_model_fit, src/Algorithm.h:1131-1135 */
static void lm_fit(oalg *a, const int *A, int k, double *bA) {
  const odata *d = a->d;
  int nt = a->n_rows, i, u, v;
  double *g = (double *)malloc((size_t)k * (size_t)k * sizeof(double));
  double *c = (double *)malloc((size_t)k * sizeof(double));
  for (u = 0; u < k; u++) {
    const double *cu = d->x + (size_t)A[u] * (size_t)d->n;
    double s = 0.0;
    for (i = 0; i < nt; i++) s += cu[a->rows[i]] * d->y[a->rows[i]];
    c[u] = s;
    for (v = 0; v <= u; v++) {
      const double *cv = d->x + (size_t)A[v] * (size_t)d->n;
      s = 0.0;
      for (i = 0; i < nt; i++) s += cu[a->rows[i]] * cv[a->rows[i]];
      if (u == v) s += a->lambda;
      g[(size_t)v * k + u] = s;
    }
  }
  bess_oracle_sym_solve(g, k, c, bA);
  free(g);
  free(c);
}

/* Weighted normal equations of an IRLS step on the design [1, X_A]:
 * (2*lambda*diag(0,1,..,1) + Z'WZ) b = Z'W z, src/Algorithm.h:1171,1199 (logistic),
 * :1299 (Poisson). */
static void irls_solve(oalg *a, const int *A, int k, const double *W, const double *z, double *b) {
  const odata *d = a->d;
  int nt = a->n_rows, m = k + 1, i, u, v;
  double *g = (double *)malloc((size_t)m * (size_t)m * sizeof(double));
  double *c = (double *)malloc((size_t)m * sizeof(double));
  for (u = 0; u < m; u++) {
    const double *cu = u ? d->x + (size_t)A[u - 1] * (size_t)d->n : NULL;
    double s = 0.0;
    for (i = 0; i < nt; i++) s += (cu ? cu[a->rows[i]] : 1.0) * W[i] * z[i];
    c[u] = s;
    for (v = 0; v <= u; v++) {
      const double *cv = v ? d->x + (size_t)A[v - 1] * (size_t)d->n : NULL;
      s = 0.0;
      for (i = 0; i < nt; i++) s += (cu ? cu[a->rows[i]] : 1.0) * W[i] * (cv ? cv[a->rows[i]] : 1.0);
      if (u == v && u > 0) s += 2.0 * a->lambda;
      g[(size_t)v * m + u] = s;
    }
  }
  bess_oracle_sym_solve(g, m, c, b);
  free(g);
  free(c);
}

/* eta_i = b[0] + sum_u x_{i,A[u]} b[u+1] */
static void design_eta(const oalg *a, const int *A, int k, const double *b, double *eta) {
  const odata *d = a->d;
  int nt = a->n_rows, i, u;
  for (i = 0; i < nt; i++) eta[i] = 0.0;
  for (u = 0; u < k; u++) {
    const double *c = d->x + (size_t)A[u] * (size_t)d->n;
    for (i = 0; i < nt; i++) eta[i] += c[a->rows[i]] * b[u + 1];
  }
  for (i = 0; i < nt; i++) eta[i] += b[0];
}

/* GroupPdasLogistic::primary_model_fit, src/Algorithm.h:1148-1204; pi(), src/logistic.cpp:15-59.
 * Cold start, <= 1+30 solves, returns the iterate BEFORE the last solve. */
static void logistic_fit(oalg *a, const int *A, int k, double *bA, double *coef0) {
  const odata *d = a->d;
  int nt = a->n_rows, m = k + 1, i, j;
  double *b0 = (double *)calloc((size_t)m, sizeof(double));
  double *b1 = (double *)calloc((size_t)m, sizeof(double));
  double *eta = (double *)malloc((size_t)nt * sizeof(double));
  double *Pi = (double *)malloc((size_t)nt * sizeof(double));
  double *W = (double *)malloc((size_t)nt * sizeof(double));
  double *Z = (double *)malloc((size_t)nt * sizeof(double));
  double ll0 = 0.0, ll1;
  design_eta(a, A, k, b0, eta);
  for (i = 0; i < nt; i++) {
    double e = exp(clamp30(eta[i])), yi = d->y[a->rows[i]];
    Pi[i] = e / (1.0 + e);
    ll0 += (yi * log(Pi[i]) + (1.0 - yi) * log(1.0 - Pi[i])) * d->w[a->rows[i]];
    W[i] = Pi[i] * (1.0 - Pi[i]);
    Z[i] = eta[i] + (yi - Pi[i]) / W[i];
    W[i] = W[i] * d->w[a->rows[i]];
  }
  irls_solve(a, A, k, W, Z, b1);
  for (j = 0; j < 30; j++) {
    design_eta(a, A, k, b1, eta);
    ll1 = 0.0;
    for (i = 0; i < nt; i++) {
      double e = exp(clamp30(eta[i])), yi = d->y[a->rows[i]];
      Pi[i] = e / (1.0 + e);
      ll1 += (yi * log(Pi[i]) + (1.0 - yi) * log(1.0 - Pi[i])) * d->w[a->rows[i]];
    }
    if (fabs(ll0 - ll1) / (0.1 + fabs(ll1)) < 1e-6) break;
    memcpy(b0, b1, (size_t)m * sizeof(double));
    ll0 = ll1;
    for (i = 0; i < nt; i++) {
      double yi = d->y[a->rows[i]];
      W[i] = Pi[i] * (1.0 - Pi[i]);
      if (W[i] < 0.001) W[i] = 0.001;
      Z[i] = eta[i] + (yi - Pi[i]) / W[i]; /* eta is the un-clamped X*beta0, :1193 */
      W[i] = W[i] * d->w[a->rows[i]];
    }
    irls_solve(a, A, k, W, Z, b1);
  }
  for (i = 0; i < k; i++) bA[i] = b0[i + 1];
  *coef0 = b0[0];
  free(b0);
  free(b1);
  free(eta);
  free(Pi);
  free(W);
  free(Z);
}

/* GroupPdasLogistic::get_A, src/Algorithm.h:1206-1263 (1x1 groups) */
/* shared tail of the GLM get_A: d = X^T g - 2 lambda beta, M_g = X_g^T diag(h) X_g + 2 lambda I */
static void glm_scores(oalg *a, const double *g, const double *h, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, p = d->p, i, j, u, v, gg, nb = d->goff[d->N];
  double *dc = (double *)malloc((size_t)p * sizeof(double));
  double *bd = (double *)malloc((size_t)d->N * sizeof(double));
  double *mb = (double *)malloc((size_t)nb * sizeof(double));
  for (j = 0; j < p; j++) {
    const double *c = d->x + (size_t)j * (size_t)d->n;
    double s1 = 0.0;
    for (i = 0; i < nt; i++) s1 += c[a->rows[i]] * g[i];
    dc[j] = s1 - 2.0 * a->lambda * a->beta[j];
  }
  for (gg = 0; gg < d->N; gg++) {
    int sz = d->gsz[gg], c0 = d->gidx[gg];
    for (u = 0; u < sz; u++)
      for (v = 0; v <= u; v++) {
        const double *cu = d->x + (size_t)(c0 + u) * (size_t)d->n, *cv = d->x + (size_t)(c0 + v) * (size_t)d->n;
        double s2 = 0.0;
        for (i = 0; i < nt; i++) s2 += (cu[a->rows[i]] * h[i]) * cv[a->rows[i]];
        if (u == v) s2 += 2.0 * a->lambda;
        mb[d->goff[gg] + (size_t)v * sz + u] = s2;
        mb[d->goff[gg] + (size_t)u * sz + v] = s2;
      }
  }
  group_scores(a, mb, dc, bd);
  select_top(a, bd, A);
  free(dc);
  free(bd);
  free(mb);
}

static void logistic_get_A(oalg *a, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, i;
  double *g = (double *)malloc((size_t)nt * sizeof(double));
  double *h = (double *)malloc((size_t)nt * sizeof(double));
  lin_pred(d, a->beta, a->rows, nt, g);
  for (i = 0; i < nt; i++) {
    double e = exp(clamp30(g[i] + a->coef0)), pr = e / (e + 1.0), wi = d->w[a->rows[i]];
    g[i] = wi * (d->y[a->rows[i]] - pr);
    h[i] = wi * pr * (1.0 - pr);
  }
  glm_scores(a, g, h, A);
  free(g);
  free(h);
}

/* GroupPdasPoisson::primary_model_fit, src/Algorithm.h:1273-1322.  Warm start from
 * (coef0, beta_A) -- but Algorithm::fit zeroes beta_A first (:157), so only coef0
 * survives. */
static void poisson_fit(oalg *a, const int *A, int k, double *bA, double *coef0) {
  const odata *d = a->d;
  int nt = a->n_rows, m = k + 1, i, j;
  double *b0 = (double *)calloc((size_t)m, sizeof(double));
  double *eta = (double *)malloc((size_t)nt * sizeof(double));
  double *ee = (double *)malloc((size_t)nt * sizeof(double));
  double *W = (double *)malloc((size_t)nt * sizeof(double));
  double *Z = (double *)malloc((size_t)nt * sizeof(double));
  double ll0 = 1e5, ll1;
  for (i = 0; i < k; i++) b0[i + 1] = bA[i];
  b0[0] = *coef0;
  design_eta(a, A, k, b0, eta);
  for (i = 0; i < nt; i++) ee[i] = exp(eta[i]);
  for (j = 0; j < 50; j++) {
    for (i = 0; i < nt; i++) {
      W[i] = ee[i] * d->w[a->rows[i]];
      Z[i] = eta[i] + (d->y[a->rows[i]] - ee[i]) / ee[i];
    }
    irls_solve(a, A, k, W, Z, b0);
    design_eta(a, A, k, b0, eta);
    ll1 = 0.0;
    for (i = 0; i < nt; i++) {
      eta[i] = clamp30(eta[i]);
      ee[i] = exp(eta[i]);
      if (ee[i] < 0.001) ee[i] = 0.001;
      ll1 += (d->y[a->rows[i]] * eta[i] - ee[i]) * d->w[a->rows[i]];
    }
    if (fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-6) break;
    ll0 = ll1;
  }
  for (i = 0; i < k; i++) bA[i] = b0[i + 1];
  *coef0 = b0[0];
  free(b0);
  free(eta);
  free(ee);
  free(W);
  free(Z);
}

/* GroupPdasPoisson::get_A, src/Algorithm.h:1324-1367 (no eta clamp) */
static void poisson_get_A(oalg *a, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, i;
  double *g = (double *)malloc((size_t)nt * sizeof(double));
  double *h = (double *)malloc((size_t)nt * sizeof(double));
  lin_pred(d, a->beta, a->rows, nt, g);
  for (i = 0; i < nt; i++) {
    double e = exp(g[i] + a->coef0), wi = d->w[a->rows[i]];
    g[i] = (d->y[a->rows[i]] - e) * wi;
    h[i] = e * wi;
  }
  glm_scores(a, g, h, A);
  free(g);
  free(h);
}

/* loglik_cox, src/coxph.cpp:16-40, on an arbitrary sorted row subset */
static double cox_loglik(const odata *d, const double *eta_in, const int *rows, int nr) {
  int i;
  double cum = 0.0, s = 0.0;
  for (i = nr - 1; i >= 0; i--) {
    double e = exp(clamp30(eta_in[i]));
    cum = (i == nr - 1) ? e : cum + e;
    s += (log(e / cum) * d->y[rows[i]]) * d->w[rows[i]];
  }
  return s;
}

/* GroupPdasCox::primary_model_fit, src/Algorithm.h:1377-1490.  The reference forms the
 * risk-set sums with a dense n x n upper-triangular ones matrix (:1386,1425-1428);
 * here they are reverse running sums -- the same numbers in O(nk^2). */
static void cox_fit(oalg *a, const int *A, int k, double *bA) {
  const odata *d = a->d;
  int nt = a->n_rows, i, u, v, l, m;
  double *b0 = (double *)calloc((size_t)k + 1, sizeof(double));
  double *b1 = (double *)calloc((size_t)k + 1, sizeof(double));
  double *bz = (double *)calloc((size_t)k + 1, sizeof(double)); /* design_eta wants an intercept slot */
  double *eta = (double *)malloc((size_t)nt * sizeof(double));
  double *th = (double *)malloc((size_t)nt * sizeof(double));
  double *s0 = (double *)malloc((size_t)nt * sizeof(double));
  double *xa = (double *)malloc((size_t)nt * (size_t)(k ? k : 1) * sizeof(double)); /* S1/S0, column-major */
  double *g = (double *)malloc((size_t)(k ? k : 1) * sizeof(double));
  double *h = (double *)malloc((size_t)(k ? k : 1) * (size_t)(k ? k : 1) * sizeof(double));
  double *dd = (double *)malloc((size_t)(k ? k : 1) * sizeof(double));
  double ll0 = 1e5, ll1;
  for (l = 1; l <= 30; l++) {
    double step;
    for (u = 0; u < k; u++) bz[u + 1] = b0[u];
    design_eta(a, A, k, bz, eta);
    for (i = 0; i < nt; i++) th[i] = exp(clamp30(eta[i]));
    for (i = nt - 1; i >= 0; i--) s0[i] = (i == nt - 1) ? th[i] : s0[i + 1] + th[i];
    for (u = 0; u < k; u++) {
      const double *cu = d->x + (size_t)A[u] * (size_t)d->n;
      double acc = 0.0, gs = 0.0;
      for (i = nt - 1; i >= 0; i--) {
        acc += cu[a->rows[i]] * th[i];
        xa[(size_t)u * nt + i] = acc / s0[i];
      }
      for (i = 0; i < nt; i++)
        gs += (cu[a->rows[i]] - xa[(size_t)u * nt + i]) * (d->w[a->rows[i]] * d->y[a->rows[i]]);
      g[u] = gs + 2.0 * a->lambda * b0[u];
    }
    for (u = 0; u < k; u++) {
      const double *cu = d->x + (size_t)A[u] * (size_t)d->n;
      for (v = u; v < k; v++) {
        const double *cv = d->x + (size_t)A[v] * (size_t)d->n;
        double acc = 0.0, hs = 0.0;
        for (i = nt - 1; i >= 0; i--) {
          acc += (th[i] * cu[a->rows[i]]) * cv[a->rows[i]];
          hs += (acc / s0[i] - xa[(size_t)u * nt + i] * xa[(size_t)v * nt + i]) *
                (d->w[a->rows[i]] * d->y[a->rows[i]]);
        }
        hs = -hs;
        if (u == v) hs += 2.0 * a->lambda;
        h[(size_t)u * k + v] = hs;
        h[(size_t)v * k + u] = hs;
      }
    }
    bess_oracle_sym_solve(h, k, g, dd);
    m = 1;
    step = 0.5;
    for (u = 0; u < k; u++) b1[u] = b0[u] - step * dd[u];
    for (u = 0; u < k; u++) bz[u + 1] = b1[u];
    design_eta(a, A, k, bz, eta);
    ll1 = cox_loglik(d, eta, a->rows, nt);
    while (ll0 > ll1 && m < 5) {
      m = m + 1;
      step = pow(0.5, (double)m);
      for (u = 0; u < k; u++) b1[u] = b0[u] - step * dd[u];
      for (u = 0; u < k; u++) bz[u + 1] = b1[u];
      design_eta(a, A, k, bz, eta);
      ll1 = cox_loglik(d, eta, a->rows, nt);
    }
    if (fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-5) break;
    memcpy(b0, b1, (size_t)k * sizeof(double));
    ll0 = ll1;
  }
  for (u = 0; u < k; u++) bA[u] = b0[u];
  free(b0);
  free(b1);
  free(bz);
  free(eta);
  free(th);
  free(s0);
  free(xa);
  free(g);
  free(h);
  free(dd);
}

/* GroupPdasCox::get_A, algorithm_type 1/5 branch, src/Algorithm.h:1569-1640 */
static void cox_get_A(oalg *a, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, p = d->p, i, j;
  double *th = (double *)malloc((size_t)nt * sizeof(double));
  double *s0 = (double *)malloc((size_t)nt * sizeof(double));
  double *bd = (double *)malloc((size_t)p * sizeof(double));
  lin_pred(d, a->beta, a->rows, nt, th);
  for (i = 0; i < nt; i++) th[i] = d->w[a->rows[i]] * exp(clamp30(th[i]));
  for (i = nt - 1; i >= 0; i--) s0[i] = (i == nt - 1) ? th[i] : s0[i + 1] + th[i];
  for (j = 0; j < p; j++) {
    const double *c = d->x + (size_t)j * (size_t)d->n;
    double a1 = 0.0, a2 = 0.0, l1 = 0.0, l2 = 0.0, dj, t;
    for (i = nt - 1; i >= 0; i--) {
      double xv = c[a->rows[i]], xt = th[i] * xv, q1, q2;
      a1 += xt;
      a2 += xv * xt;
      if (d->y[a->rows[i]] != 0.0) {
        q1 = a1 / s0[i];
        q2 = a2 / s0[i] - q1 * q1;
        l1 += (xv - q1) * d->w[a->rows[i]];
        l2 += q2 * d->w[a->rows[i]];
      }
    }
    l1 = -l1 + 2.0 * a->lambda * a->beta[j];
    l2 = l2 + 2.0 * a->lambda;
    dj = -l1 / l2;
    t = fabs(a->beta[j] + dj);
    bd[j] = t * sqrt(l2);
  }
  select_top(a, bd, A);
  free(th);
  free(s0);
  free(bd);
}

/* GroupPdasCox::get_A, algorithm_type 2/3 branch, src/Algorithm.h:1497-1568: the n x n Hessian h of the negative
 * partial log-likelihood with respect to the linear predictor is built exactly as the reference does
 * (h(i,j) = -cum_theta3(min(i,j)) theta_i theta_j off the diagonal, cum_theta2(i) theta_i added on it), then the
 * per-group blocks X_g^T h X_g + 2 lambda I, their square roots and the sacrifices as in the other families. */
static void cox_get_A_group(oalg *a, int *A) {
  const odata *d = a->d;
  int nt = a->n_rows, p = d->p, i, j, u, v, gg, nb = d->goff[d->N];
  double *th = (double *)malloc((size_t)nt * sizeof(double));
  double *c1 = (double *)malloc((size_t)nt * sizeof(double));
  double *c2 = (double *)malloc((size_t)nt * sizeof(double));
  double *c3 = (double *)malloc((size_t)nt * sizeof(double));
  double *g = (double *)malloc((size_t)nt * sizeof(double));
  double *h = (double *)malloc((size_t)nt * (size_t)nt * sizeof(double));
  double *hx = (double *)malloc((size_t)nt * sizeof(double));
  double *dc = (double *)malloc((size_t)p * sizeof(double));
  double *bd = (double *)malloc((size_t)d->N * sizeof(double));
  double *mb = (double *)malloc((size_t)nb * sizeof(double));
  lin_pred(d, a->beta, a->rows, nt, th);
  for (i = 0; i < nt; i++) th[i] = d->w[a->rows[i]] * exp(clamp30(th[i])); /* :1512-1520 */
  for (i = nt - 1; i >= 0; i--) c1[i] = (i == nt - 1) ? th[i] : c1[i + 1] + th[i]; /* cum_theta */
  for (i = 0; i < nt; i++) {
    double yw = d->y[a->rows[i]] * d->w[a->rows[i]];
    c2[i] = yw / c1[i] + (i ? c2[i - 1] : 0.0);          /* cum_theta2, :1526-1530 */
    c3[i] = yw / pow(c1[i], 2) + (i ? c3[i - 1] : 0.0);  /* cum_theta3, :1531-1535 */
  }
  for (i = 0; i < nt; i++)
    for (j = i; j < nt; j++) { /* :1536-1545: upper triangle from row i, mirrored */
      double val = -c3[i] * th[i] * th[j];
      h[(size_t)i * nt + j] = val;
      h[(size_t)j * nt + i] = val;
    }
  for (i = 0; i < nt; i++) h[(size_t)i * nt + i] += c2[i] * th[i]; /* :1546 */
  for (i = 0; i < nt; i++) g[i] = d->w[a->rows[i]] * d->y[a->rows[i]] - c2[i] * th[i]; /* :1547 */
  for (j = 0; j < p; j++) {
    const double *c = d->x + (size_t)j * (size_t)d->n;
    double s1 = 0.0;
    for (i = 0; i < nt; i++) s1 += c[a->rows[i]] * g[i];
    dc[j] = s1 - 2.0 * a->lambda * a->beta[j]; /* :1548 */
  }
  for (gg = 0; gg < d->N; gg++) { /* :1549-1558 */
    int sz = d->gsz[gg], c0 = d->gidx[gg];
    for (v = 0; v < sz; v++) {
      const double *cv = d->x + (size_t)(c0 + v) * (size_t)d->n;
      for (i = 0; i < nt; i++) { /* hx = h x_v */
        double s2 = 0.0;
        const double *hr = h + (size_t)i * nt;
        for (j = 0; j < nt; j++) s2 += hr[j] * cv[a->rows[j]];
        hx[i] = s2;
      }
      for (u = v; u < sz; u++) {
        const double *cu = d->x + (size_t)(c0 + u) * (size_t)d->n;
        double s2 = 0.0;
        for (i = 0; i < nt; i++) s2 += cu[a->rows[i]] * hx[i];
        if (u == v) s2 += 2.0 * a->lambda;
        mb[d->goff[gg] + (size_t)v * sz + u] = s2;
        mb[d->goff[gg] + (size_t)u * sz + v] = s2;
      }
    }
  }
  group_scores(a, mb, dc, bd);
  select_top(a, bd, A);
  free(th);
  free(c1);
  free(c2);
  free(c3);
  free(g);
  free(h);
  free(hx);
  free(dc);
  free(bd);
  free(mb);
}

/* Algorithm::fit, src/Algorithm.h:113-171 */
static void alg_fit(oalg *a) {
  int T0 = a->T0, p = a->d->p, i, ll, l, K;
  int *A = (int *)calloc((size_t)(T0 ? T0 : 1), sizeof(int));
  int *Alist = (int *)calloc((size_t)(T0 ? T0 : 1) * (size_t)(a->max_iter + 2), sizeof(int));
  int *cols = (int *)malloc((size_t)p * sizeof(int));
  double *bA = (double *)malloc((size_t)p * sizeof(double));
  memcpy(a->beta, a->beta_init, (size_t)p * sizeof(double));
  a->coef0 = a->coef0_init;
  for (l = 1; l <= a->max_iter; l++) {
    int off, same = 0;
    a->l = l;
    if (a->model_type == 1)
      lm_get_A(a, A);
    else if (a->model_type == 2)
      logistic_get_A(a, A);
    else if (a->model_type == 3)
      poisson_get_A(a, A);
    else if (a->algorithm_type == 2 || a->algorithm_type == 3)
      cox_get_A_group(a, A);
    else
      cox_get_A(a, A);
    K = expand_groups(a->d, A, T0, cols); /* find_ind + X_seg, :155-156 */
    /* the trace stores the EXPANDED column list (aligned with the coefficients of the fit) */
    ipush(&t_meta, l);
    ipush(&t_meta, T0);
    ipush(&t_meta, a->n_rows);
    off = t_a.n;
    ipush(&t_meta, off);
    for (i = 0; i < K; i++) ipush(&t_a, cols[i]);
    memcpy(Alist + (size_t)l * T0, A, (size_t)T0 * sizeof(int));
    for (i = 0; i < K; i++) bA[i] = 0.0;
    if (a->model_type == 1)
      lm_fit(a, cols, K, bA);
    else if (a->model_type == 2)
      logistic_fit(a, cols, K, bA, &a->coef0);
    else if (a->model_type == 3)
      poisson_fit(a, cols, K, bA, &a->coef0);
    else
      cox_fit(a, cols, K, bA);
    for (i = 0; i < K; i++) dpush(&t_beta, bA[i]);
    dpush(&t_coef0, a->coef0);
    for (i = 0; i < p; i++) a->beta[i] = 0.0;
    for (i = 0; i < K; i++) a->beta[cols[i]] = bA[i];
    for (ll = 0; ll < l && !same; ll++) same = memcmp(A, Alist + (size_t)ll * T0, (size_t)T0 * sizeof(int)) == 0;
    if (same) break;
  }
  if (l > a->max_iter) a->l = a->max_iter + 1;
  free(A);
  free(Alist);
  free(cols);
  free(bA);
}

/* ------------------------------------------------------------------ Metric */

typedef struct {
  int ic_type, is_cv, K;
  int **train, **test;
  int *n_train, *n_test;
  double **cv_init; /* K x p, cv_initial_model_param, src/Metric.h:22,39-47 */
  double **cv_xtx;  /* K x p, cal_cv_group_XTX, src/Metric.h:108-129 (LM) */
  int depth;
} ometric;

/* loss of (beta, coef0) on a row subset: the four train_loss / test_loss bodies,
 * src/Metric.h:145-148,190 (LM), :266-290,338-351 (logistic), :426-440,489 (Poisson via
 * loglik_poisson, src/poisson.cpp:15-45), :565-568,609 (Cox). */
static double subset_loss(const oalg *a, const int *rows, int nr, int is_test) {
  const odata *d = a->d;
  int i;
  double s = 0.0;
  double *eta = (double *)malloc((size_t)(nr ? nr : 1) * sizeof(double));
  lin_pred(d, a->beta, rows, nr, eta);
  if (a->model_type == 1) {
    for (i = 0; i < nr; i++) {
      double r = d->y[rows[i]] - eta[i];
      s += r * r;
    }
    s = is_test ? s / (double)(2 * nr) : s / (double)nr;
  } else if (a->model_type == 2) {
    double cl = is_test ? 25.0 : 30.0;
    for (i = 0; i < nr; i++) {
      double v = eta[i] + a->coef0, e, pr, yi = d->y[rows[i]];
      if (v > cl) v = cl;
      if (v < -cl) v = -cl;
      e = exp(v);
      pr = e / (e + 1.0);
      s += d->w[rows[i]] * (yi * log(pr) + (1.0 - yi) * log(1.0 - pr));
    }
    s = -2.0 * s;
  } else if (a->model_type == 3) {
    for (i = 0; i < nr; i++) {
      double v = clamp30(eta[i] + a->coef0), yi = d->y[rows[i]], t = 0.0, jj;
      if (yi != 1.0)
        for (jj = 1.0; jj <= yi; jj = jj + 1.0) t = t + log(jj);
      s += (yi * v - exp(v) - t) * d->w[rows[i]];
    }
    s = is_test ? -s : -2.0 * s;
  } else {
    s = -2.0 * cox_loglik(d, eta, rows, nr);
  }
  free(eta);
  return s;
}

static int *g_full_rows; /* 0..n-1 */

static double metric_train_loss(ometric *m, oalg *a) {
  double v = subset_loss(a, g_full_rows, a->d->n, 0);
  if (m->depth == 0) dpush(&t_loss, v);
  return v;
}

/* test_loss with CV: src/Metric.h:150-195 (LM), :292-355, :442-494, :570-614 */
static double metric_test_loss(ometric *m, oalg *a) {
  int k, j, p = a->d->p;
  double s = 0.0;
  for (k = 0; k < m->K; k++) {
    if (a->warm_start) memcpy(a->beta_init, m->cv_init[k], (size_t)p * sizeof(double));
    a->rows = m->train[k];
    a->n_rows = m->n_train[k];
    if (a->model_type == 1) a->xtx = m->cv_xtx[k];
    alg_fit(a);
    if (a->warm_start)
      for (j = 0; j < p; j++) m->cv_init[k][j] = a->beta[j];
    s += subset_loss(a, m->test[k], m->n_test[k], 1);
  }
  return s / (double)m->K;
}

/* ic: src/Metric.h:197-256 (LM: n*log(loss) + c*T0), :357-416, :496-555, :616-675 (loss + c*T0) */
static double metric_ic(ometric *m, oalg *a) {
  double v, n = (double)a->d->n, p = (double)a->d->p, c = 0.0, loss;
  m->depth++;
  if (m->is_cv) {
    v = metric_test_loss(m, a);
  } else if (m->ic_type < 1 || m->ic_type > 4) {
    v = 0.0;
  } else {
    /* LM picks the group formula by algorithm_type (src/Metric.h:205,230), the other families by
     * g_index.size() == p (:365, :504, :624); the group formula uses log(g_num) and group_df = sparsity level */
    int grouped = a->model_type == 1 ? !(a->algorithm_type == 1 || a->algorithm_type == 5) : (a->d->N != a->d->p);
    double pp = grouped ? (double)a->d->N : p;
    loss = metric_train_loss(m, a);
    if (m->ic_type == 1) c = 2.0;
    if (m->ic_type == 2) c = log(n);
    if (m->ic_type == 3) c = log(pp) * log(log(n));
    if (m->ic_type == 4) c = log(n) + 2.0 * log(pp);
    v = (a->model_type == 1 ? n * log(loss) : loss) + c * (double)a->T0;
  }
  m->depth--;
  if (m->depth == 0) dpush(&t_ic, v);
  return v;
}

/* ------------------------------------------------------------------ paths */

typedef struct {
  double *beta; /* p */
  double coef0, loss, ic;
} opoint;

static void denorm(const odata *d, double *beta, double *coef0) {
  /* src/path.cpp:76-110 and :330-373 */
  int j;
  double dot = 0.0, sn = sqrt((double)d->n);
  if (!d->is_normal) return;
  for (j = 0; j < d->p; j++) {
    beta[j] = sn * beta[j] / d->x_norm[j];
    dot += beta[j] * d->x_mean[j];
  }
  if (d->data_type == 1)
    *coef0 = d->y_mean - dot;
  else if (d->data_type == 2)
    *coef0 = *coef0 - dot;
}

static void run_fit(oalg *a, int T0, double lambda, const double *beta_init, double coef0_init, const double *xtx) {
  a->rows = g_full_rows;
  a->n_rows = a->d->n;
  a->T0 = T0;
  a->lambda = lambda;
  memcpy(a->beta_init, beta_init, (size_t)a->d->p * sizeof(double));
  a->coef0_init = coef0_init;
  a->xtx = xtx;
  alg_fit(a);
}

/* sequential_path, src/path.cpp:25-132 */
static void seq_path(oalg *a, ometric *m, const double *xtx, const int *seq, int ns, const double *lam, int nl,
                     opoint *best) {
  int p = a->d->p, i, j, bi = 0, bj = 0;
  double *beta_init = (double *)calloc((size_t)p, sizeof(double)), coef0_init = 0.0;
  double *betas = (double *)calloc((size_t)p * (size_t)ns * (size_t)nl, sizeof(double));
  double *c0 = (double *)calloc((size_t)ns * (size_t)nl, sizeof(double));
  double *ls = (double *)calloc((size_t)ns * (size_t)nl, sizeof(double));
  double *ic = (double *)calloc((size_t)ns * (size_t)nl, sizeof(double));
  for (i = 0; i < ns; i++) {
    int step = (i % 2 == 0) ? 1 : -1;
    for (j = (i % 2 == 0) ? 0 : nl - 1; j < nl && j >= 0; j += step) {
      size_t q = (size_t)j * ns + i;
      run_fit(a, seq[i], lam[j], beta_init, coef0_init, xtx);
      if (a->warm_start) {
        memcpy(beta_init, a->beta, (size_t)p * sizeof(double));
        coef0_init = a->coef0;
      }
      memcpy(betas + q * p, a->beta, (size_t)p * sizeof(double));
      c0[q] = a->coef0;
      ls[q] = metric_train_loss(m, a);
      ic[q] = metric_ic(m, a);
    }
  }
  /* minCoeff over a column-major (ns x nl) matrix: first minimum in storage order */
  for (j = 0; j < nl; j++)
    for (i = 0; i < ns; i++)
      if (ic[(size_t)j * ns + i] < ic[(size_t)bj * ns + bi]) {
        bi = i;
        bj = j;
      }
  memcpy(best->beta, betas + ((size_t)bj * ns + bi) * p, (size_t)p * sizeof(double));
  best->coef0 = c0[(size_t)bj * ns + bi];
  best->loss = ls[(size_t)bj * ns + bi];
  best->ic = ic[(size_t)bj * ns + bi];
  denorm(a->d, best->beta, &best->coef0);
  free(beta_init);
  free(betas);
  free(c0);
  free(ls);
  free(ic);
}

static int iround(double v) { return (int)round(v); }

/* gs_path, src/path.cpp:134-389.  lambda is never set there (stays 0). */
static void gs_path(oalg *a, ometric *m, const double *xtx, int s_min, int s_max, opoint *best) {
  int p = a->d->p, Tmin = s_min, Tmax = s_max, T1, T2, T;
  double *beta_init = (double *)calloc((size_t)p, sizeof(double)), coef0_init = 0.0;
  double ic1, ic2, icT1, icT2, best_ic = DBL_MAX;
  T1 = iround(0.618 * Tmin + 0.382 * Tmax);
  T2 = iround(0.382 * Tmin + 0.618 * Tmax);

#define GS_FIT(T)                                                \
  do {                                                           \
    run_fit(a, (T), 0.0, beta_init, coef0_init, xtx);            \
    if (a->warm_start) {                                         \
      memcpy(beta_init, a->beta, (size_t)p * sizeof(double));    \
      coef0_init = a->coef0;                                     \
    }                                                            \
  } while (0)

  GS_FIT(T1);
  metric_train_loss(m, a);
  ic1 = metric_ic(m, a);
  icT1 = ic1;
  GS_FIT(T2);
  metric_train_loss(m, a);
  ic2 = metric_ic(m, a);
  icT2 = metric_ic(m, a); /* evaluated twice, :204 and :210 */
  while (T1 != T2) {
    if (icT1 < icT2) {
      Tmax = T2;
      T2 = T1;
      ic2 = ic1;
      icT2 = ic1;
      T1 = iround(0.618 * Tmin + 0.382 * Tmax);
      GS_FIT(T1);
      metric_train_loss(m, a);
      ic1 = metric_ic(m, a);
      icT1 = metric_ic(m, a);
    } else {
      Tmin = T1;
      T1 = T2;
      ic1 = ic2;
      icT1 = ic2;
      T2 = iround(0.382 * Tmin + 0.618 * Tmax);
      GS_FIT(T2);
      metric_train_loss(m, a);
      ic2 = metric_ic(m, a);
      icT2 = metric_ic(m, a);
    }
  }
  memset(best->beta, 0, (size_t)p * sizeof(double));
  best->coef0 = 0.0;
  best->loss = 0.0;
  for (T = Tmin; T <= Tmax; T++) {
    double v;
    GS_FIT(T);
    v = metric_ic(m, a);
    if (v < best_ic) {
      /* read AFTER ic(): under CV this is the last fold's fit, :314-319 */
      memcpy(best->beta, a->beta, (size_t)p * sizeof(double));
      best->coef0 = a->coef0;
      best->loss = metric_train_loss(m, a);
      best_ic = v;
    }
  }
  best->ic = best_ic;
  /* gs_path de-normalises with "data_type == 1 ... else" (:330-342): data_type 3 also
   * subtracts beta.x_mean, which is zero there (Normalize4 never sets x_mean). */
  if (a->d->is_normal) {
    int j;
    double dot = 0.0, sn = sqrt((double)a->d->n);
    for (j = 0; j < p; j++) {
      best->beta[j] = sn * best->beta[j] / a->d->x_norm[j];
      dot += best->beta[j] * a->d->x_mean[j];
    }
    best->coef0 = a->d->data_type == 1 ? a->d->y_mean - dot : best->coef0 - dot;
  }
  free(beta_init);
#undef GS_FIT
}

/* ------------------------------------------------------------------ Powell path (L0L2 / bsrr)
 * pgs_path, golden_section_search, seq_search and their geometry helpers, src/path.cpp:391-1309.
 * The ic_sequence matrix those functions fill is returned only by the R build (ic_mat) and never read back,
 * so it is not kept here (the reference even indexes it out of range, e.g. :651 uses int(c[0]) as the row
 * of an (s_max - s_min + 1)-row matrix). */

static int sgn(double a) { return a > 0 ? 1 : (a < 0 ? -1 : 0); } /* :391-405 */
static double det2(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }

static void line_intersection(double l1[2][2], double l2[2][2], double out[2], int *ok) { /* :414-440 */
  double xd[2], yd[2], d[2], div;
  xd[0] = l1[0][0] - l1[1][0];
  xd[1] = l2[0][0] - l2[1][0];
  yd[0] = l1[0][1] - l1[1][1];
  yd[1] = l2[0][1] - l2[1][1];
  div = det2(xd, yd);
  if (div == 0) {
    *ok = 0;
    return;
  }
  d[0] = det2(l1[0], l1[1]);
  d[1] = det2(l2[0], l2[1]);
  out[0] = det2(d, xd) / div;
  out[1] = det2(d, yd) / div;
  *ok = 1;
}

static void cal_intersections(const double p[2], const double u[2], int s_min, int s_max, double lmin, double lmax,
                              double a[2], double b[2]) { /* :445-577 */
  double l0[2][2], ls[4][2][2], is[4][2];
  int ok[4], i, j;
  l0[0][0] = p[0];
  l0[0][1] = p[1];
  l0[1][0] = p[0] + u[0];
  l0[1][1] = p[1] + u[1];
  ls[0][0][0] = s_min; ls[0][0][1] = lmin; ls[0][1][0] = s_min; ls[0][1][1] = lmax;
  ls[1][0][0] = s_max; ls[1][0][1] = lmin; ls[1][1][0] = s_max; ls[1][1][1] = lmax;
  ls[2][0][0] = s_min; ls[2][0][1] = lmin; ls[2][1][0] = s_max; ls[2][1][1] = lmin;
  ls[3][0][0] = s_min; ls[3][0][1] = lmax; ls[3][1][0] = s_max; ls[3][1][1] = lmax;
  for (i = 0; i < 4; i++) line_intersection(l0, ls[i], is[i], &ok[i]);
  for (i = 0; i < 4; i++)
    if (ok[i] && ((is[i][0] < s_min - 0.0001) | (is[i][0] > s_max + 0.0001) | (is[i][1] < lmin - 0.001) |
                  (is[i][1] > lmax + 0.001)))
      ok[i] = 0;
  for (i = 0; i < 4; i++)
    if (ok[i])
      for (j = i + 1; j < 4; j++)
        if (ok[j] && fabs(is[i][0] - is[j][0]) < 0.0001 && fabs(is[i][1] - is[j][1]) < 0.0001) ok[j] = 0;
  j = 0;
  for (i = 0; i < 4; i++)
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

typedef struct {
  oalg *a;
  ometric *m;
  const double *xtx;
  double *beta_init; /* the search's own warm-start vector */
  double coef0_init;
} psearch;

static void ps_fit(psearch *ps, int T0, double lambda) {
  run_fit(ps->a, T0, lambda, ps->beta_init, ps->coef0_init, ps->xtx);
  if (ps->a->warm_start) {
    memcpy(ps->beta_init, ps->a->beta, (size_t)ps->a->d->p * sizeof(double));
    ps->coef0_init = ps->a->coef0;
  }
}

/* golden_section_search, :579-935 */
static void golden_section_search(oalg *al, ometric *m, const double *xtx, const double p[2], const double u[2],
                                  int s_min, int s_max, double lmin, double lmax, double best_arg[2], double *beta1,
                                  double *coef01, double *loss1, double *ic1) {
  int P = al->d->p, tt = 0, i;
  psearch ps;
  double *bt1 = (double *)calloc((size_t)P, sizeof(double)), *bt2 = (double *)calloc((size_t)P, sizeof(double));
  double lt1 = 0, lt2 = 0, c01 = 0, c02 = 0, closs, dloss, a[2] = {0, 0}, b[2] = {0, 0}, c[2], d[2], h[2];
  const double s_tol = 2, ltol = (lmax - lmin) / 200;
  const double invphi = (pow(5, 0.5) - 1.0) / 2.0, invphi2 = (3.0 - pow(5, 0.5)) / 2.0;
  ps.a = al;
  ps.m = m;
  ps.xtx = xtx;
  ps.beta_init = (double *)calloc((size_t)P, sizeof(double));
  ps.coef0_init = 0.0;
  cal_intersections(p, u, s_min, s_max, lmin, lmax, a, b);
  h[0] = b[0] - a[0];
  h[1] = b[1] - a[1];
  c[0] = a[0] + invphi2 * h[0];
  c[1] = a[1] + invphi2 * h[1];
  d[0] = a[0] + invphi * h[0];
  d[1] = a[1] + invphi * h[1];
  if (h[0] > 0.0001) {
    c[0] = (int)c[0];
    d[0] = ceil(d[0]);
  } else if (h[0] < -0.0001) {
    c[0] = ceil(c[0]);
    d[0] = (int)d[0];
  } else {
    c[0] = round(c[0]);
    d[0] = round(d[0]);
  }
  ps_fit(&ps, (int)c[0], exp(c[1]));
  closs = metric_ic(m, al);
  c01 = al->coef0;
  memcpy(bt1, al->beta, (size_t)P * sizeof(double));
  lt1 = metric_train_loss(m, al);
  ps_fit(&ps, (int)d[0], exp(d[1]));
  dloss = metric_ic(m, al);
  c02 = al->coef0;
  memcpy(bt2, al->beta, (size_t)P * sizeof(double));
  lt2 = metric_train_loss(m, al);
  for (;;) {
    if ((fabs((invphi2 - invphi) * h[0]) <= s_tol && fabs((invphi2 - invphi) * h[1]) < ltol) || tt == 50) {
      double min_loss, tmp;
      if (closs < dloss) {
        best_arg[0] = c[0];
        best_arg[1] = c[1];
        min_loss = closs;
        memcpy(beta1, bt1, (size_t)P * sizeof(double));
        *coef01 = c01;
        *ic1 = closs;
        *loss1 = lt1;
      } else {
        best_arg[0] = d[0];
        best_arg[1] = d[1];
        min_loss = dloss;
        memcpy(beta1, bt2, (size_t)P * sizeof(double));
        *coef01 = c02;
        *ic1 = dloss;
        *loss1 = lt2;
      }
      for (i = 1; i < fabs((invphi2 - invphi) * h[0]); i++) {
        ps_fit(&ps, (int)(c[0] + sgn(h[0]) * i), exp(c[1]));
        tmp = metric_ic(m, al);
        if (tmp < min_loss) {
          best_arg[0] = c[0] + sgn(h[0]) * i;
          best_arg[1] = c[1];
          min_loss = tmp;
          memcpy(beta1, al->beta, (size_t)P * sizeof(double));
          *coef01 = al->coef0;
          *loss1 = metric_train_loss(m, al);
          *ic1 = min_loss;
        }
      }
      break;
    }
    if (tt >= 100) break;
    tt++;
    if (closs < dloss) {
      b[0] = d[0];
      b[1] = d[1];
      d[0] = c[0];
      d[1] = c[1];
      dloss = closs;
      /* the reference keeps beta_temp2 / coef0_temp2 / train_loss_temp2 of the OLD d here (:762-766) */
      h[0] = b[0] - a[0];
      h[1] = b[1] - a[1];
      c[0] = a[0] + invphi2 * h[0];
      c[1] = a[1] + invphi2 * h[1];
      if (h[0] > 0.0001)
        c[0] = (int)c[0];
      else if (h[0] < -0.0001)
        c[0] = ceil(c[0]);
      else
        c[0] = round(c[0]);
      ps_fit(&ps, (int)c[0], exp(c[1]));
      closs = metric_ic(m, al);
      c01 = al->coef0;
      memcpy(bt1, al->beta, (size_t)P * sizeof(double));
      lt1 = metric_train_loss(m, al);
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
      if (h[0] > 0.0001)
        d[0] = ceil(d[0]);
      else if (h[0] < -0.0001)
        d[0] = (int)d[0];
      else
        d[0] = round(d[0]);
      ps_fit(&ps, (int)d[0], exp(d[1]));
      dloss = metric_ic(m, al);
      c02 = al->coef0;
      memcpy(bt2, al->beta, (size_t)P * sizeof(double));
      lt2 = metric_train_loss(m, al);
    }
  }
  free(bt1);
  free(bt2);
  free(ps.beta_init);
}

static int gdc_int(int a, int b) { /* GDC, :937-953 */
  int Max = a > b ? a : b, Min = (a == Max) ? b : a, z = Min;
  while (Max % Min != 0) {
    z = Max % Min;
    Max = Min;
    Min = z;
  }
  return z;
}

/* seq_search, :954-1137.  u is modified in place like the reference does. */
static void seq_search(oalg *al, ometric *m, const double *xtx, double p[2], double u[2], int s_min, int s_max,
                       double lmin, double lmax, double best_arg[2], double *beta1, double *coef01, double *loss1,
                       double *ic1, int nlambda) {
  int P = al->d->p, i = 0, j = 0, cap = (s_max - s_min + 1) * nlambda + 2, k_lambda, mp1 = 0, mp2 = 0, q, minpos;
  psearch ps;
  double d_lambda = (lmax - lmin) / (nlambda - 1), coef0_warm;
  double *b1 = (double *)calloc((size_t)P * cap, sizeof(double)), *b2 = (double *)calloc((size_t)P * cap, sizeof(double));
  double *c1 = (double *)calloc((size_t)cap, sizeof(double)), *c2 = (double *)calloc((size_t)cap, sizeof(double));
  double *l1 = (double *)calloc((size_t)cap, sizeof(double)), *l2 = (double *)calloc((size_t)cap, sizeof(double));
  double *i1 = (double *)calloc((size_t)cap, sizeof(double)), *i2 = (double *)calloc((size_t)cap, sizeof(double));
  double *beta_warm = (double *)calloc((size_t)P, sizeof(double));
  ps.a = al;
  ps.m = m;
  ps.xtx = xtx;
  ps.beta_init = (double *)calloc((size_t)P, sizeof(double));
  ps.coef0_init = 0.0;
  k_lambda = (int)fabs(round(u[1] / d_lambda));
  if (fabs(u[0]) != 1 && k_lambda != 1) {
    if (k_lambda == 0 && u[0] != 0) {
      u[0] = u[0] / fabs(u[0]);
    } else if (u[0] == 0 && k_lambda != 0) {
      u[1] = u[1] / k_lambda;
    } else if (!(k_lambda == 0 && (int)u[0] == 0)) { /* the reference divides by zero there */
      int g = gdc_int(k_lambda, abs((int)u[0]));
      if (g) {
        u[0] = round(u[0] / g);
        u[1] = u[1] / g;
      }
    }
  }
  ps_fit(&ps, (int)(p[0] + i * u[0]), exp(p[1] + i * u[1]));
  i1[i] = metric_ic(m, al);
  memcpy(b1, al->beta, (size_t)P * sizeof(double));
  c1[i] = al->coef0;
  l1[i] = metric_train_loss(m, al);
  i2[j] = i1[i];
  memcpy(b2, b1, (size_t)P * sizeof(double));
  c2[j] = c1[i];
  l2[j] = l1[i];
  i++;
  j++;
  memcpy(beta_warm, ps.beta_init, (size_t)P * sizeof(double));
  coef0_warm = ps.coef0_init;
  while ((p[0] + i * u[0] <= s_max) && (p[1] + i * u[1] <= lmax + d_lambda * 1e-4) && (p[0] + i * u[0] >= s_min) &&
         (p[1] + i * u[1] >= lmin - d_lambda * 1e-4) && i < cap) {
    ps_fit(&ps, (int)(p[0] + i * u[0]), exp(p[1] + i * u[1]));
    i1[i] = metric_ic(m, al);
    memcpy(b1 + (size_t)i * P, al->beta, (size_t)P * sizeof(double));
    c1[i] = al->coef0;
    l1[i] = metric_train_loss(m, al);
    i++;
  }
  memcpy(ps.beta_init, beta_warm, (size_t)P * sizeof(double));
  ps.coef0_init = coef0_warm;
  while ((p[0] - j * u[0] <= s_max) && (p[1] - j * u[1] <= lmax + d_lambda * 1e-4) && (p[0] - j * u[0] >= s_min) &&
         (p[1] - j * u[1] >= lmin - d_lambda * 1e-4) && j < cap) {
    ps_fit(&ps, (int)(p[0] - j * u[0]), exp(p[1] - j * u[1]));
    i2[j] = metric_ic(m, al);
    memcpy(b2 + (size_t)j * P, al->beta, (size_t)P * sizeof(double));
    c2[j] = al->coef0;
    l2[j] = metric_train_loss(m, al);
    j++;
  }
  for (q = 1; q < i; q++)
    if (i1[q] < i1[mp1]) mp1 = q;
  for (q = 1; q < j; q++)
    if (i2[q] < i2[mp2]) mp2 = q;
  if (i1[mp1] < i2[mp2]) {
    minpos = mp1;
    *ic1 = i1[mp1];
    *loss1 = l1[mp1];
    memcpy(beta1, b1 + (size_t)mp1 * P, (size_t)P * sizeof(double));
    *coef01 = c1[mp1];
  } else {
    minpos = -mp2;
    *ic1 = i2[mp2];
    *loss1 = l2[mp2];
    memcpy(beta1, b2 + (size_t)mp2 * P, (size_t)P * sizeof(double));
    *coef01 = c2[mp2];
  }
  best_arg[0] = p[0] + minpos * u[0];
  best_arg[1] = p[1] + minpos * u[1];
  free(b1); free(b2); free(c1); free(c2); free(l1); free(l2); free(i1); free(i2); free(beta_warm);
  free(ps.beta_init);
}

/* pgs_path, :1138-1309 */
static int pgs_path(oalg *al, ometric *m, const double *xtx, int s_min, int s_max, double lmin, double lmax,
                    int powell_path, int nlambda, opoint *best, double *lambda_out) {
  int P = al->d->p, ttt = 0, i, k, mi = 0, rc = 1;
  double Pp[3][2], U[2][2], ct = 0, lt = 0, it = 0;
  double *bt = (double *)calloc((size_t)P, sizeof(double));
  double *ball = (double *)calloc((size_t)P * 100, sizeof(double));
  double call[100], lall[100], iall[100], lam[100];
  if (powell_path == 1) nlambda = 100;
  Pp[0][0] = (double)s_min;
  Pp[0][1] = lmin;
  U[1][0] = 1.;
  U[1][1] = 0.;
  U[0][0] = 0.;
  U[0][1] = (lmax - lmin) / (nlambda - 1);
#define SEARCH(pin, uu, pout)                                                                                  \
  do {                                                                                                         \
    if (powell_path == 1)                                                                                      \
      golden_section_search(al, m, xtx, pin, uu, s_min, s_max, lmin, lmax, pout, bt, &ct, &lt, &it);           \
    else                                                                                                       \
      seq_search(al, m, xtx, pin, uu, s_min, s_max, lmin, lmax, pout, bt, &ct, &lt, &it, nlambda);             \
  } while (0)
#define RECORD(idx, lamv)                                         \
  do {                                                            \
    memcpy(ball + (size_t)(idx)*P, bt, (size_t)P * sizeof(double)); \
    call[idx] = ct;                                               \
    lall[idx] = lt;                                               \
    iall[idx] = it;                                               \
    lam[idx] = (lamv);                                            \
  } while (0)
  SEARCH(Pp[0], U[1], Pp[0]);
  RECORD(ttt, exp(Pp[0][1]));
  while (ttt < 11) {
    ttt++;
    for (i = 0; i < 2; i++) {
      SEARCH(Pp[i], U[i], Pp[i + 1]);
      RECORD(ttt, exp(Pp[i + 1][1]));
      ttt++;
    }
    U[0][0] = U[1][0];
    U[0][1] = U[1][1];
    U[1][0] = Pp[2][0] - Pp[0][0];
    U[1][1] = Pp[2][1] - Pp[0][1];
    if ((!(fabs(U[1][0]) <= 0.0001 && fabs(U[1][1]) <= 0.0001)) && ttt < 11) {
      SEARCH(Pp[0], U[1], Pp[0]);
      RECORD(ttt, exp(Pp[0][1]));
    } else {
      /* final fit at P[0]; beta_init / coef0_init are whatever the last search left in the algorithm (:1221-1225) */
      al->rows = g_full_rows;
      al->n_rows = al->d->n;
      al->T0 = (int)Pp[0][0];
      al->lambda = exp(Pp[0][1]);
      al->xtx = xtx;
      alg_fit(al);
      memcpy(ball + (size_t)ttt * P, al->beta, (size_t)P * sizeof(double));
      call[ttt] = al->coef0;
      lall[ttt] = metric_train_loss(m, al);
      iall[ttt] = metric_ic(m, al);
      lam[ttt] = exp(Pp[0][1]);
      ttt++;
      for (k = 0; k < ttt; k++) denorm(al->d, ball + (size_t)k * P, &call[k]);
      for (k = 1; k < ttt; k++)
        if (iall[k] < iall[mi]) mi = k;
      if (iall[mi] == iall[ttt - 1]) mi = ttt - 1;
      memcpy(best->beta, ball + (size_t)mi * P, (size_t)P * sizeof(double));
      best->coef0 = call[mi];
      best->loss = lall[mi];
      best->ic = iall[mi];
      *lambda_out = lam[mi];
      rc = 0;
      break;
    }
  }
#undef SEARCH
#undef RECORD
  free(bt);
  free(ball);
  return rc; /* 1: "powell end wrong" (:1298-1308) */
}

/* ------------------------------------------------------------------ screening (SIS)
 * screening(), src/screening.cpp:26-105, for singleton groups: one marginal fit per column on the RAW data (it runs
 * before Data::normalize, src/bess.cpp:57-61), score = coefficient^2, the screening_size best columns are kept.
 *   LM:       least squares without intercept (colPivHouseholderQr of one column, :44)   beta = x.y / x.x
 *   logistic: logit_fit, src/logistic.cpp:61-157 (n > p branch): IRLS with intercept, NO floor on W, returns the
 *             iterate before the last solve
 *   Cox:      cox_fit, src/coxph.cpp:42-109: damped Newton with step halving, clamp +-50, no ridge
 * Poisson is not restated: poisson_fit multiplies two n-vectors as matrices (src/poisson.cpp:113), which is
 * undefined behaviour in the reference build. */
static double screen_logit(const double *x, const double *y, const double *w, int n) {
  double b0[2] = {0, 0}, b1[2] = {0, 0}, ll0 = 0.0, ll1;
  double s0, s1, s2, t0, t1, det;
  int i, j;
  for (j = -1; j < 30; j++) {
    /* j == -1: the solve before the loop (:135-146) */
    const double *b = j < 0 ? b0 : b1;
    ll1 = 0.0;
    s0 = s1 = s2 = t0 = t1 = 0.0;
    for (i = 0; i < n; i++) {
      double eta = b[0] + x[i] * b[1], e = exp(clamp30(eta)), Pi = e / (1.0 + e), W, z;
      ll1 += (y[i] * log(Pi) + (1.0 - y[i]) * log(1.0 - Pi)) * w[i];
      W = Pi * (1.0 - Pi);
      z = eta + (y[i] - Pi) / W;
      W = W * w[i];
      s0 += W;
      s1 += W * x[i];
      s2 += (W * x[i]) * x[i];
      t0 += W * z;
      t1 += (W * x[i]) * z;
    }
    if (j < 0) {
      ll0 = ll1;
    } else {
      if (fabs(ll0 - ll1) / (0.1 + fabs(ll1)) < 1e-6) break;
      b0[0] = b1[0];
      b0[1] = b1[1];
      ll0 = ll1;
    }
    det = s0 * s2 - s1 * s1;
    b1[0] = (s2 * t0 - s1 * t1) / det;
    b1[1] = (s0 * t1 - s1 * t0) / det;
  }
  return b0[1];
}

static double clamp50(double v) { return v > 50 ? 50 : (v < -50 ? -50 : v); }

static double screen_cox_ll(const double *x, const double *st, const double *w, int n, double b) {
  int i;
  double cum = 0.0, s = 0.0;
  for (i = n - 1; i >= 0; i--) {
    double e = exp(clamp30(x[i] * b)); /* loglik_cox clamps at +-30, src/coxph.cpp:20-30 */
    cum = (i == n - 1) ? e : cum + e;
    s += (log(e / cum) * st[i]) * w[i];
  }
  return s;
}

static double screen_cox(const double *x, const double *st, const double *w, int n) {
  double b0 = 0.0, b1 = 0.0, ll0 = 1e5, ll1;
  int i, l, m;
  for (l = 1; l <= 30; l++) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, g = 0.0, h = 0.0, d;
    for (i = n - 1; i >= 0; i--) {
      double th = exp(clamp50(x[i] * b0)), q1;
      a0 += th;
      a1 += th * x[i];
      a2 += (th * x[i]) * x[i];
      q1 = a1 / a0;
      g += (x[i] - q1) * (w[i] * st[i]);
      h += (a2 / a0 - q1 * q1) * (w[i] * st[i]);
    }
    h = -h;
    d = g / h;
    m = 1;
    b1 = b0 - 0.5 * d;
    ll1 = screen_cox_ll(x, st, w, n, b1);
    while (ll0 > ll1 && m < 5) {
      m = m + 1;
      b1 = b0 - pow(0.5, (double)m) * d;
      ll1 = screen_cox_ll(x, st, w, n, b1);
    }
    if (fabs(ll0 - ll1) / fabs(0.1 + ll0) < 1e-5) break;
    b0 = b1;
    ll0 = ll1;
  }
  return b0;
}

int bess_oracle_screening(const double *x, int n, int p, const double *y, const double *weight, int model_type,
                          int screening_size, const int *always_select, int always_len, int *screening_A) {
  double *col = (double *)malloc((size_t)n * sizeof(double)), *score = (double *)malloc((size_t)p * sizeof(double));
  int i, j;
  if (model_type == 3 || screening_size < 1 || screening_size > p) {
    free(col);
    free(score);
    return 1;
  }
  for (j = 0; j < p; j++) {
    double b;
    for (i = 0; i < n; i++) col[i] = x[(size_t)i * p + j];
    if (model_type == 1) {
      double sxy = 0.0, sxx = 0.0;
      for (i = 0; i < n; i++) {
        sxy += col[i] * y[i];
        sxx += col[i] * col[i];
      }
      b = sxy / sxx;
    } else if (model_type == 2) {
      b = screen_logit(col, y, weight, n);
    } else {
      b = screen_cox(col, y, weight, n);
    }
    score[j] = b * b;
  }
  for (i = 0; i < always_len; i++) score[always_select[i]] = DBL_MAX;
  bess_oracle_max_k(score, p, screening_size, screening_A);
  free(col);
  free(score);
  return 0;
}

/* ------------------------------------------------------------------ driver */

/* group_XTX, src/utilities.cpp:153-165: X_g^T X_g on a row subset, one block per group */
static void group_xtx(const odata *d, const int *rows, int nr, double *out) {
  int g, u, v, i;
  for (g = 0; g < d->N; g++) {
    int sz = d->gsz[g], c0 = d->gidx[g];
    for (u = 0; u < sz; u++)
      for (v = 0; v <= u; v++) {
        const double *cu = d->x + (size_t)(c0 + u) * (size_t)d->n, *cv = d->x + (size_t)(c0 + v) * (size_t)d->n;
        double s = 0.0;
        for (i = 0; i < nr; i++) s += cu[rows[i]] * cv[rows[i]];
        out[d->goff[g] + (size_t)v * sz + u] = s;
        out[d->goff[g] + (size_t)u * sz + v] = s;
      }
  }
}

/* bessCpp, src/bess.cpp:37-214 (no screening) */
int bess_oracle_run(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                    int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                    int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                    int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                    const int *always_select, int always_len, double *beta_out, double *coef0_out,
                    double *train_loss_out, double *ic_out) {
  return bess_oracle_run2(x, n, p, y, weight, data_type, is_normal, algorithm_type, model_type, max_iter, path_type,
                          is_warm_start, ic_type, is_cv, K, cv_fold_id, sequence, sequence_len, lambda_seq,
                          lambda_len, s_min, s_max, 0.0, 0.0, 100, 1, always_select, always_len, beta_out, coef0_out,
                          train_loss_out, ic_out, NULL);
}

/* As bess_oracle_run plus the Powell-path arguments of bessCpp (src/bess.cpp:174-180): path_type 3 runs
 * pgs_path with log_lambda = log(max(lambda, 1e-5)). */
int bess_oracle_run2(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                     int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                     int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                     int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                     double lambda_min, double lambda_max, int nlambda, int powell_path, const int *always_select,
                     int always_len, double *beta_out, double *coef0_out, double *train_loss_out, double *ic_out,
                     double *lambda_out) {
  return bess_oracle_run3(x, n, p, y, weight, data_type, is_normal, algorithm_type, model_type, max_iter, path_type,
                          is_warm_start, ic_type, is_cv, K, cv_fold_id, sequence, sequence_len, lambda_seq, lambda_len,
                          s_min, s_max, lambda_min, lambda_max, nlambda, powell_path, NULL, 0, always_select,
                          always_len, beta_out, coef0_out, train_loss_out, ic_out, lambda_out);
}

/* As bess_oracle_run2 plus the group structure: g_index[g] = first column of group g (ascending, g_index[0] = 0),
 * NULL = every column its own group.  Sparsity levels and always_select then count / name groups. */
int bess_oracle_run3(const double *x, int n, int p, const double *y, const double *weight, int data_type,
                     int is_normal, int algorithm_type, int model_type, int max_iter, int path_type,
                     int is_warm_start, int ic_type, int is_cv, int K, const int *cv_fold_id, const int *sequence,
                     int sequence_len, const double *lambda_seq, int lambda_len, int s_min, int s_max,
                     double lambda_min, double lambda_max, int nlambda, int powell_path, const int *g_index, int g_len,
                     const int *always_select, int always_len, double *beta_out, double *coef0_out,
                     double *train_loss_out, double *ic_out, double *lambda_out) {
  odata d;
  int gq;
  oalg a;
  ometric m;
  opoint best;
  double *xtx;
  int i, j, k;
  const double t_enter = now_s();
  double t_path;
  if (n < 1 || p < 1 || model_type < 1 || model_type > 4) return 1;
  if (is_cv && (cv_fold_id == NULL || K < 2)) return 2;
  if (g_index == NULL) g_len = p;
  if (g_len < 1 || g_len > p || (g_index != NULL && g_index[0] != 0)) return 4;
  for (i = 1; g_index != NULL && i < g_len; i++)
    if (g_index[i] <= g_index[i - 1] || g_index[i] >= p) return 4;
  if (model_type == 4 && g_len != p && !(algorithm_type == 2 || algorithm_type == 3))
    return 4; /* Cox with real groups exists only in the group branch of get_A (algorithm_type 2 / 3) */
  if (path_type == 1) {
    for (i = 0; i < sequence_len; i++)
      if (sequence[i] < 0 || sequence[i] > g_len) return 3;
    if (sequence_len < 1 || lambda_len < 1) return 3;
  } else if (s_min < 0 || s_max > g_len || s_min > s_max) {
    return 3;
  }
  if (path_type == 3 && (s_min < 1 || nlambda < 2)) return 3;
  t_meta.n = t_a.n = t_beta.n = t_coef0.n = t_loss.n = t_ic.n = 0;

  d.n = n;
  d.p = p;
  d.data_type = data_type;
  d.is_normal = is_normal;
  d.x = (double *)malloc((size_t)n * (size_t)p * sizeof(double));
  d.y = (double *)malloc((size_t)n * sizeof(double));
  d.w = (double *)malloc((size_t)n * sizeof(double));
  d.x_mean = (double *)calloc((size_t)p, sizeof(double));
  d.x_norm = (double *)calloc((size_t)p, sizeof(double));
  d.y_mean = 0.0;
  d.N = g_len;
  d.gidx = (int *)malloc((size_t)g_len * sizeof(int));
  d.gsz = (int *)malloc((size_t)g_len * sizeof(int));
  d.goff = (int *)malloc((size_t)(g_len + 1) * sizeof(int));
  d.gmax = 1;
  d.goff[0] = 0;
  for (gq = 0; gq < g_len; gq++) {
    d.gidx[gq] = g_index ? g_index[gq] : gq;
    d.gsz[gq] = (gq + 1 < g_len ? (g_index ? g_index[gq + 1] : gq + 1) : p) - d.gidx[gq];
    if (d.gsz[gq] > d.gmax) d.gmax = d.gsz[gq];
    d.goff[gq + 1] = d.goff[gq] + d.gsz[gq] * d.gsz[gq];
  }
  /* row-major -> column-major in 64 x 64 blocks (the plain double loop crawls at BASELINE sizes: 32 GB for configs[4]) */
  for (i = 0; i < n; i += 64) {
    const int ie = i + 64 < n ? i + 64 : n;
    for (j = 0; j < p; j += 64) {
      const int je = j + 64 < p ? j + 64 : p;
      int ii, jj;
      for (jj = j; jj < je; jj++)
        for (ii = i; ii < ie; ii++) d.x[(size_t)jj * n + ii] = x[(size_t)ii * p + jj];
    }
  }
  for (i = 0; i < n; i++) {
    d.y[i] = y[i];
    d.w[i] = weight[i];
  }
  if (is_normal) data_normalize(&d);
  if (model_type == 1) data_add_weight(&d);

  g_full_rows = (int *)malloc((size_t)n * sizeof(int));
  for (i = 0; i < n; i++) g_full_rows[i] = i;

  memset(&a, 0, sizeof(a));
  a.d = &d;
  a.model_type = model_type;
  a.algorithm_type = algorithm_type;
  a.max_iter = max_iter;
  a.warm_start = is_warm_start;
  a.beta = (double *)calloc((size_t)p, sizeof(double));
  a.beta_init = (double *)calloc((size_t)p, sizeof(double));
  a.always = always_select;
  a.n_always = always_len;

  /* group_XTX on the full data, src/path.cpp:37 -> src/utilities.cpp:153-165 (LM only) */
  xtx = (double *)calloc((size_t)d.goff[d.N], sizeof(double));
  if (model_type == 1) group_xtx(&d, g_full_rows, n, xtx);

  memset(&m, 0, sizeof(m));
  m.ic_type = ic_type;
  m.is_cv = is_cv;
  m.K = K;
  if (is_cv) {
    m.train = (int **)calloc((size_t)K, sizeof(int *));
    m.test = (int **)calloc((size_t)K, sizeof(int *));
    m.n_train = (int *)calloc((size_t)K, sizeof(int));
    m.n_test = (int *)calloc((size_t)K, sizeof(int));
    m.cv_init = (double **)calloc((size_t)K, sizeof(double *));
    m.cv_xtx = (double **)calloc((size_t)K, sizeof(double *));
    for (k = 0; k < K; k++) {
      m.train[k] = (int *)malloc((size_t)n * sizeof(int));
      m.test[k] = (int *)malloc((size_t)n * sizeof(int));
      for (i = 0; i < n; i++) {
        if (cv_fold_id[i] == k)
          m.test[k][m.n_test[k]++] = i;
        else
          m.train[k][m.n_train[k]++] = i;
      }
      m.cv_init[k] = (double *)calloc((size_t)p, sizeof(double));
      m.cv_xtx[k] = (double *)calloc((size_t)d.goff[d.N], sizeof(double));
      if (model_type == 1) group_xtx(&d, m.train[k], m.n_train[k], m.cv_xtx[k]);
    }
  }

  best.beta = (double *)calloc((size_t)p, sizeof(double));
  best.coef0 = best.loss = best.ic = 0.0;
  t_path = now_s();
  g_last_setup_s = t_path - t_enter;
  if (path_type == 1) {
    seq_path(&a, &m, xtx, sequence, sequence_len, lambda_seq, lambda_len, &best);
  } else if (path_type == 3) {
    double lam = 0.0;
    double lo = log(lambda_min > 1e-5 ? lambda_min : 1e-5), hi = log(lambda_max > 1e-5 ? lambda_max : 1e-5);
    pgs_path(&a, &m, xtx, s_min, s_max, lo, hi, powell_path, nlambda, &best, &lam);
    if (lambda_out) *lambda_out = lam;
  } else {
    gs_path(&a, &m, xtx, s_min, s_max, &best);
  }

  g_last_path_s = now_s() - t_path;
  memcpy(beta_out, best.beta, (size_t)p * sizeof(double));
  *coef0_out = best.coef0;
  *train_loss_out = best.loss;
  *ic_out = best.ic;

  if (is_cv) {
    for (k = 0; k < K; k++) {
      free(m.train[k]);
      free(m.test[k]);
      free(m.cv_init[k]);
      free(m.cv_xtx[k]);
    }
    free(m.train);
    free(m.test);
    free(m.n_train);
    free(m.n_test);
    free(m.cv_init);
    free(m.cv_xtx);
  }
  free(best.beta);
  free(xtx);
  free(a.beta);
  free(a.beta_init);
  free(g_full_rows);
  g_full_rows = NULL;
  free(d.x);
  free(d.y);
  free(d.w);
  free(d.x_mean);
  free(d.x_norm);
  free(d.gidx);
  free(d.gsz);
  free(d.goff);
  return 0;
}
