// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 when consecutive MFMAs share the accumulator (chains of the
// layer kernels) vs rotate over NACC accumulators, with 1 or 2 waves per SIMD, and with a ds_read between the MFMAs.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/exp/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int GAP>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  __shared__ float sh[4096];
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(i * 0.5f); }
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  float side = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
        if (GAP == 1) asm volatile("s_nop 0");
        if (GAP == 2) asm volatile("s_waitcnt lgkmcnt(0)");
        if (GAP == 3) { side += sh[(threadIdx.x + it) & 4095]; }
      }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = side;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int GAP>
void run(int threads, const char* what) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NACC, GAP>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 256; ++i) avg += h[i];
  avg /= 256;
  const double per_simd = (double)iters * 12 * ((threads / 64 + 3) / 4);      // MFMAs per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, GAP>), dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s threads %4d NACC %d: %.1f cyc per MFMA per SIMD slot | wall %.3f ms -> %.1f TFLOP/s chip\n", what, threads, NACC, avg / per_simd, ms,
         256.0 * (threads / 64) * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 0>(64, "same accumulator, back to back");
  run<1, 0>(128, "same accumulator, back to back");
  run<1, 0>(1024, "same accumulator, back to back");
  run<4, 0>(1024, "4 accumulators");
  run<1, 0>(256, "same accumulator, back to back");
  run<2, 0>(256, "2 accumulators alternating");
  run<4, 0>(256, "4 accumulators");
  run<1, 0>(512, "same accumulator, back to back");
  run<2, 0>(512, "2 accumulators alternating");
  run<1, 1>(256, "same accumulator, s_nop between");
  run<1, 1>(512, "same accumulator, s_nop between");
  run<1, 2>(256, "same accumulator, s_waitcnt between");
  run<1, 2>(512, "same accumulator, s_waitcnt between");
  run<2, 2>(512, "2 accumulators, s_waitcnt between");
  run<1, 3>(512, "same accumulator, ds_read+add between");
  run<2, 3>(512, "2 accumulators, ds_read+add between");
  return 0;
}
