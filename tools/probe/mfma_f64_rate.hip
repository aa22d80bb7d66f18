// mfma_f64_rate.hip -- what the fp64 matrix pipe of one SIMD really delivers: a loop of v_mfma_f64_16x16x4_f64 on operands
// that live in registers (no memory, no LDS), W waves per SIMD on every compute unit, for about a millisecond.  Prints,
// per configuration: shader cycles per MFMA and SIMD (s_memtime; the peak of the data sheet assumes 64), the clock the
// chip held meanwhile (s_memtime / s_memrealtime) and the TFLOP/s that follow.  Random operands (zeros run faster:
// MI355X_MICROARCH.md, DVFS).     hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) k_rate(const double *__restrict__ in, double *__restrict__ out, int iters,
                                              unsigned long long *__restrict__ stamps) {
  const int tid = threadIdx.x;
  double a = in[tid], b = in[512 + tid];
  d4 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; t++) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) {
    stamps[2 * ((size_t)blockIdx.x * (blockDim.x / 64) + tid / 64)] = t1 - t0;
    stamps[2 * ((size_t)blockIdx.x * (blockDim.x / 64) + tid / 64) + 1] = r1 - r0;
  }
}

template <int NACC>
static void run(int waves_per_simd, int iters, const double *din, double *dout, unsigned long long *dst, int cus) {
  const int threads = 64 * 4 * waves_per_simd, nw = cus * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {  // the last repetition is the one reported (clocks settled)
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(cus), dim3(threads), 0, 0, din, dout, iters, dst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st((size_t)2 * nw);
  hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double cyc = 0.0, ticks = 0.0;
  for (int w = 0; w < nw; w++) {
    cyc += (double)st[2 * w];
    ticks += (double)st[2 * w + 1];
  }
  cyc /= nw;
  ticks /= nw;
  const double mfma_per_wave = (double)iters * 4 * NACC;
  const double cyc_per_mfma_simd = cyc / (mfma_per_wave * waves_per_simd);
  const double tf = (double)cus * 4 * waves_per_simd * mfma_per_wave * 2048.0 / (ms * 1e-3) / 1e12;
  std::printf("{\"waves_per_simd\": %d, \"accumulators\": %d, \"kernel_ms\": %.3f, \"shader_cycles_per_mfma_and_simd\": %.2f, "
              "\"clock_MHz\": %.0f, \"TFLOPs\": %.1f, \"frac_of_78.6\": %.3f}\n",
              waves_per_simd, NACC, ms, cyc_per_mfma_simd, 100.0 * cyc / ticks, tf, tf / 78.6);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<double> h(1024);
  unsigned long long x = 88172645463325252ull;
  for (auto &v : h) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    v = (double)(x >> 11) / 9007199254740992.0 - 0.5;
  }
  double *din, *dout;
  unsigned long long *dst;
  hipMalloc(&din, 1024 * sizeof(double));
  hipMalloc(&dout, (size_t)cus * 512 * sizeof(double));
  hipMalloc(&dst, (size_t)cus * 8 * 2 * sizeof(unsigned long long));
  hipMemcpy(din, h.data(), 1024 * sizeof(double), hipMemcpyHostToDevice);
  std::printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
  const int iters = 1500;  // x 4 x NACC MFMAs per wave: about a millisecond
  run<2>(1, iters * 2, din, dout, dst, cus);
  run<4>(1, iters, din, dout, dst, cus);
  run<8>(1, iters / 2, din, dout, dst, cus);
  run<2>(2, iters, din, dout, dst, cus);
  run<4>(2, iters / 2, din, dout, dst, cus);
  run<4>(1, iters * 8, din, dout, dst, cus);  // ~8 ms: where the clock ends up under a sustained load
  run<4>(2, iters * 4, din, dout, dst, cus);
  return 0;
}
