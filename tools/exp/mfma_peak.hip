// Sustained v_mfma_f32_32x32x16_f16 rate with nothing else going on (registers only): the practical MFMA ceiling of the
// box (clock under matrix load included).  hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {   // pseudo-random operand bits: realistic toggling (constant operands draw less power)
    unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 4096.f)); b[i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 4096.f));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    if ((it & 63) == 63)      // keep the accumulators finite and the operands changing
      for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] *= 1e-3f;
    a = -a;
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 3; ++wpc) {          // workgroups (of 4 waves) per CU
    const int grid = 256 * wpc, iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)grid * 4 * iters * 8 * 4 * 32768.0;
      printf("waves/SIMD %d  rep %d  %.2f ms  %.1f TFLOP/s (f16 dense)\n", wpc, rep, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
