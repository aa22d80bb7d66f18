// Does creating and destroying streams with a compute-unit mask (hipExtStreamCreateWithCUMask: a hardware queue of their
// own) leak anything?  N rounds of: create K such streams, launch a tiny kernel on each, destroy them; prints the time per
// round, then stays alive for 25 s; meanwhile another process (`./cumask_stream_churn 1 1 child`, started by the shell -- never
// by fork / exec from a process that holds the GPU) creates one such stream and runs a kernel: what the C-host test does
// beside a long pytest process.   hipcc --offload-arch=gfx950 -O2 -o cumask_stream_churn cumask_stream_churn.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>

__global__ void k_touch(int *p) { if (threadIdx.x == 0) p[0] += 1; }

static int one_round(int K, int *buf, const std::vector<uint32_t> &mask) {
  std::vector<hipStream_t> st(K);
  for (int i = 0; i < K; i++)
    if (hipExtStreamCreateWithCUMask(&st[i], (uint32_t)mask.size(), mask.data()) != hipSuccess) return 1;
  for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[i], buf);
  for (int i = 0; i < K; i++) hipStreamSynchronize(st[i]);
  for (int i = 0; i < K; i++) hipStreamDestroy(st[i]);
  return 0;
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 300, K = argc > 2 ? atoi(argv[2]) : 8;
  if (argc > 3) {  // child mode: one stream, one kernel
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    int *buf;
    hipMalloc((void **)&buf, 64);
    auto t0 = std::chrono::steady_clock::now();
    int rc = one_round(1, buf, mask);
    printf("child: rc %d, %.1f ms\n", rc, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  int *buf;
  hipMalloc((void **)&buf, 64);
  for (int r = 0; r < N; r++) {
    auto t0 = std::chrono::steady_clock::now();
    if (one_round(K, buf, mask)) {
      printf("round %d: stream creation failed\n", r);
      break;
    }
    const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (r < 3 || (r + 1) % 50 == 0) printf("round %d (%d streams so far): %.2f ms\n", r + 1, (r + 1) * K, ms);
    fflush(stdout);
  }
  fflush(stdout);
  // stay alive for a while (holding the context, like a long test process): the shell starts `... 1 1 child` beside it
  sleep(argc > 3 ? 0 : 25);
  return 0;
}
