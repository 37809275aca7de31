// Matrix-pipe rate of the "f16 main product + block-scaled MX corrections" arithmetic that tests/arith_probe.py sizes against
// the parity gate: per 32-channel slab the f16x3 GEMM issues 6 v_mfma_f32_32x32x16_f16 (h1g1, h1g2, h2g1 x 2 k steps); the
// alternative issues 2 of them (h1g1) + ONE v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64 holds [q(h1)|q(h2)] . [q(g2);q(g1)]
// (one shared scale per 32-wide K block = exactly the MX block).  Registers only, pseudo-random operand bits.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mx_corr_rate.hip -o tools/exp/mx_corr_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// FMT: -1 = three f16 products (today), 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1 for the correction instruction
template <int FMT, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b, a2, b2;
  i32x8 qa, qb;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 4096.f)); b[i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 4096.f));
    a2[i] = a[i] * (_Float16)0.001f; b2[i] = b[i] * (_Float16)0.001f;
    qa[i] = (int)(h & 0x3f3f3f3fu); qb[i] = (int)((h * 2654435761u) & 0x3f3f3f3fu);   // small finite codes in every format
  }
  const int sc = 0x7f7f7f7f;      // e8m0 scale 2^0 in every byte
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {           // 4 slabs per trip
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        if constexpr (FMT < 0) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, acc[j], 0, 0, 0);
        } else {
          acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[j], FMT, FMT, 0, sc, 0, sc);
        }
      }
    }
    if ((it & 63) == 63)
      for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] *= 1e-3f;
    a = -a; qa[0] ^= 0x01010101;
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FMT>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 2, iters = 4000;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FMT, 4>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double slabs = (double)grid * 4 /*waves*/ * iters * 4 * 4 /*acc tiles*/;
  // one "slab-tile" = 32x32 outputs x 32 channels x 3 products x 2 flop in today's accounting
  printf("%-34s %8.2f ms   %.0f G slab-tiles/s   (= %.0f TFLOP/s in f16x3-equivalent flops)\n", name, best, slabs / best / 1e6,
         slabs * 32.0 * 32 * 32 * 3 * 2 / best / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  run<-1>("f16x3 (6 f16 MFMAs per slab)", out);
  run<0>("f16 + MXFP8 correction", out);
  run<2>("f16 + MXFP6 (e2m3) correction", out);
  run<4>("f16 + MXFP4 correction", out);
  return 0;
}
