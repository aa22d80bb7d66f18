// How many dependent single-workgroup kernel chains can one process drive at once?  T host threads, a stream each,
// N launches of a kernel that spins ~W microseconds (s_memrealtime, 100 MHz); wall time against T = 1.
//   hipcc --offload-arch=gfx950 -O2 -o launch_rate launch_rate.hip -lpthread ; ./launch_rate [N] [W_us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_spin(unsigned long long *out, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {
  }
  if (threadIdx.x == 0) out[0] += 1;
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1000, W = argc > 2 ? atoi(argv[2]) : 5;
  const int split = argc > 3 ? atoi(argv[3]) : 0;  // 1: every second stream at the lowest priority, the others at the highest
  int plo = 0, phi = 0;
  hipDeviceGetStreamPriorityRange(&plo, &phi);
  for (int T : {1, 2, 4, 8}) {
    std::vector<hipStream_t> st(T);
    std::vector<unsigned long long *> buf(T);
    for (int t = 0; t < T; t++) {
      if (split)
        hipStreamCreateWithPriority(&st[t], hipStreamNonBlocking, (t & 1) ? plo : phi);
      else
        hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
      hipMalloc((void **)&buf[t], 64);
      hipMemset(buf[t], 0, 64);
      for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[t], buf[t], 10);
      hipStreamSynchronize(st[t]);
    }
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
          for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[t], buf[t], W * 100);
          hipStreamSynchronize(st[t]);
        });
      for (auto &x : th) x.join();
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("threads %d: %d launches each of a %d us kernel: %.2f ms wall, %.2f us per launch per chain, %.2f us aggregate\n", T,
           N, W, 1e3 * best, 1e6 * best / N, 1e6 * best / N / T);
    for (int t = 0; t < T; t++) {
      hipStreamDestroy(st[t]);
      hipFree(buf[t]);
    }
  }
  return 0;
}
