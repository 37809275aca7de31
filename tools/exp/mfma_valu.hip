// Microbenchmark: do vector instructions of the SAME wave run under its MFMAs?  One wave per SIMD (256 threads), a loop
// of 4 independent v_mfma_f32_32x32x16_f16, each followed by NV vector instructions of one kind:
//   kind 0: v_fma_f32 on ordinary VGPRs   kind 1: v_accvgpr_read + v_fma   kind 2: v_cvt_pk_f16_f32   kind 3: v_pk_mul_f32
//   kind 4: v_exp_f32   kind 5: ds_write_b16   kind 6: v_fma_mix_f32
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/exp/mfma_valu.hip && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  __shared__ unsigned short sh[8192];
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  float agsrc = 1.0f;
  asm volatile("v_accvgpr_write_b32 a200, %0" ::"v"(agsrc));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MFMA) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        float& r = v[n & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(v[(n + 1) & 7]));
        if (KIND == 1) { float tq; asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(tq)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(tq)); }
        if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r) : "v"(v[(n + 1) & 7]));
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[(2 * n) & 6])));
        if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
        if (KIND == 5) asm volatile("ds_write_b16 %0, %1" ::"v"((threadIdx.x * 2 + n * 512) & 16383), "v"(r));
        if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(v[(n + 1) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + sh[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, bool MFMA>
double run() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NV, KIND, MFMA>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 256; ++i) avg += h[i];
  hipFree(out); hipFree(cyc);
  return avg / 256 / (iters * 4.0);
}
template <int KIND>
void row(const char* what) {
  printf("%-34s per (MFMA + NV): NV=0 %.1f | NV=2: %.1f (alone %.1f) | NV=4: %.1f (alone %.1f) | NV=8: %.1f (alone %.1f) | NV=16: %.1f (alone %.1f)\n", what,
         run<0, KIND, true>(), run<2, KIND, true>(), run<2, KIND, false>(), run<4, KIND, true>(), run<4, KIND, false>(),
         run<8, KIND, true>(), run<8, KIND, false>(), run<16, KIND, true>(), run<16, KIND, false>());
}
int main() {
  row<0>("v_fma_f32 (VGPR)");
  row<1>("v_accvgpr_read + v_add");
  row<2>("v_cvt_pk_f16_f32");
  row<3>("v_pk_mul_f32");
  row<4>("v_exp_f32");
  row<5>("ds_write_b16");
  row<6>("v_fma_mix_f32");
  return 0;
}
