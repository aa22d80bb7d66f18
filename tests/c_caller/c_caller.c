/* c_caller.c -- a plain C host of libbessx.so: no Python, no torch, nothing but include/bessx.h and the C ABI.
 * What a C or R host of the library does: create a session on host buffers, run sequential_path, read
 * the candidates.  Built and run by tests/test_c_caller_gpu.py, which compares what it prints with the same path run
 * through the ctypes binding and times it with GPU_MAX_HW_QUEUES unset (the library must not need the variable).
 *   c_caller X.bin y.bin n p kmax repeats      (X row-major n x p doubles, y n doubles)
 * Prints: "support <k> : i0 i1 ..." per candidate, "iters ...", "best <k> <ic>", "ms_per_path <min> <median>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bessx.h"

static double now_ms(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

static double *read_doubles(const char *path, size_t count) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  double *v = (double *)malloc(count * sizeof(double));
  if (v && fread(v, sizeof(double), count, f) != count) {
    free(v);
    v = NULL;
  }
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s X.bin y.bin n p kmax repeats\n", argv[0]);
    return 2;
  }
  const int n = atoi(argv[3]), p = atoi(argv[4]), kmax = atoi(argv[5]), repeats = atoi(argv[6]);
  double *x = read_doubles(argv[1], (size_t)n * (size_t)p), *y = read_doubles(argv[2], (size_t)n);
  if (!x || !y) {
    fprintf(stderr, "cannot read the inputs\n");
    return 2;
  }
  bessx_problem pb;
  memset(&pb, 0, sizeof(pb));
  pb.n = n;
  pb.p = p;
  pb.x = x;
  pb.x_col_major = 0;
  pb.y = y;
  pb.data_type = 1;
  pb.is_normal = 1;
  pb.model_type = 1;
  pb.algorithm_type = 1;
  pb.max_iter = 20;
  pb.is_warm_start = 1;
  pb.device = -1;
  bessx_session *s = NULL;
  if (bessx_session_create(&s, &pb) != BESSX_OK) {
    fprintf(stderr, "bessx_session_create: %s\n", bessx_last_error());
    return 1;
  }
  free(x);
  int *seq = (int *)malloc((size_t)kmax * sizeof(int));
  for (int i = 0; i < kmax; i++) seq[i] = i + 1;
  const double lam = 0.0;
  bessx_path_result r;
  memset(&r, 0, sizeof(r));
  r.beta = (double *)calloc((size_t)p, sizeof(double));
  r.capacity = kmax;
  r.max_T0 = kmax;
  r.cand_T0 = (int *)calloc((size_t)kmax, sizeof(int));
  r.cand_iters = (int *)calloc((size_t)kmax, sizeof(int));
  r.cand_ic = (double *)calloc((size_t)kmax, sizeof(double));
  r.cand_support = (int *)calloc((size_t)kmax * (size_t)kmax, sizeof(int));
  double *ms = (double *)calloc((size_t)(repeats > 0 ? repeats : 1), sizeof(double));
  for (int rep = -2; rep < repeats; rep++) { /* two warm-up paths */
    r.n_candidates = 0;
    const double t0 = now_ms();
    if (bessx_session_sequential_path(s, seq, kmax, &lam, 1, 3, 0, &r) != BESSX_OK) {
      fprintf(stderr, "bessx_session_sequential_path: %s\n", bessx_last_error());
      return 1;
    }
    if (rep >= 0) ms[rep] = now_ms() - t0;
  }
  for (int i = 0; i < kmax && i < r.n_candidates; i++) {
    printf("support %d :", r.cand_T0[i]);
    for (int j = 0; j < r.cand_T0[i]; j++) printf(" %d", r.cand_support[(size_t)i * kmax + j]);
    printf("\n");
  }
  printf("iters");
  for (int i = 0; i < kmax && i < r.n_candidates; i++) printf(" %d", r.cand_iters[i]);
  printf("\nbest %d %.17g\n", r.best_T0, r.ic);
  printf("chains %lld chunked_paths %lld\n", bessx_session_counter(s, 17), bessx_session_counter(s, 14));
  if (repeats > 0) {
    qsort(ms, (size_t)repeats, sizeof(double), cmp_double);
    printf("ms_per_path %.4f %.4f\n", ms[0], ms[repeats / 2]);
  }
  bessx_session_destroy(s);
  return 0;
}
