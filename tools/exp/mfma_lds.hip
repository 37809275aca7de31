// Microbenchmark: v_mfma_f32_32x32x16_f16 rate when its operands stream from LDS (ds_read_b128, prefetched one group ahead),
// R reads per group of 6 MFMAs, 1 or 2 waves per SIMD, optionally with weight-like global loads.
// hipcc --offload-arch=gfx950 -O3 -o mfma_lds.bin tools/exp/mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int R, int NACC, int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, int stride, const char* wts) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned base = (unsigned)(lane * 16 + wave * 1024);
  u32x4 cur[4], nxt[4];
  for (int r = 0; r < 4; ++r) cur[r] = u32x4{1, 2, 3, (unsigned)r};
  const f16x8 w = __builtin_bit_cast(f16x8, u32x4{0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u});
  u32x4 wb[2][4];
  const char* wl = wts + (wave & 3) * 73728 + lane * 16;      // 4 "channel tiles" of 72 KB, shared by every workgroup
  if (MODE >= 3)
    for (int j = 0; j < 4; ++j) wb[0][j] = *reinterpret_cast<const u32x4*>(wl + j * 1024);
  long long t0 = __builtin_readcyclecounter();
  if (MODE >= 3) {
    // the layer kernels' shape: blocks of 5 tile groups (6 MFMAs each) share 4 weight registers-sets fetched a block ahead
    for (int blk = 0; blk < iters / 5; blk += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int nb = (blk + half + 1) % 18;
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[half ^ 1][j] = *reinterpret_cast<const u32x4*>(wl + nb * 4096 + j * 1024);
#pragma unroll
        for (int g = 0; g < 5; ++g) {
          const unsigned a = (base + (unsigned)(((blk + half) * 5 + g) * stride)) & 0x1ffffu;
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb[half][q & 3]),
                                                                   __builtin_bit_cast(f16x8, cur[q & 3]), acc[g % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < R) {
              nxt[q] = *reinterpret_cast<const u32x4*>(lds + ((a + q * 9472) & 0x1fff0u));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
        }
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    const unsigned a = (base + (unsigned)(it * stride)) & 0x1ffffu;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < R && MODE != 2) nxt[r] = *reinterpret_cast<const u32x4*>(lds + ((a + r * 9472) & 0x1fff0u));
      else nxt[r] = cur[r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (MODE == 1)        // accumulators pinned to AGPRs
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[q % NACC]) : "v"(w), "v"(cur[q & 3]));
      else
        acc[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, cur[q & 3]), acc[q % NACC], 0, 0, 0);
      if (MODE == 2 && q < R) {   // reads spread between the MFMAs instead of in front of them
        nxt[q] = *reinterpret_cast<const u32x4*>(lds + ((a + q * 9472) & 0x1fff0u));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int NACC, int MODE = 0>
void run(int threads, int stride, const char* what) {
  float* out; long long* cyc; char* wts;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&wts, 4 * 73728 + 8192); hipMemset(wts, 0x3c, 4 * 73728 + 8192);
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<R, NACC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipLaunchKernelGGL((k<R, NACC, MODE>), dim3(256), dim3(threads), 147456, 0, out, cyc, iters, stride, wts);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<R, NACC, MODE>), dim3(256), dim3(threads), 147456, 0, out, cyc, iters, stride, wts); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s mode %d waves/SIMD %d R=%d reads per 6 MFMAs, NACC %d: wall %.3f ms -> %.0f TFLOP/s chip\n", what, MODE, threads / 256, R, NACC, ms,
         256.0 * (threads / 64) * iters * 6 * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 1>(512, 16, "no reads");
  run<1, 1>(512, 16, "linear");
  run<2, 1>(512, 16, "linear");
  run<4, 1>(512, 16, "linear");
  run<4, 2>(512, 16, "linear, 2 acc");
  run<4, 1>(256, 16, "linear");
  run<4, 1>(512, 4736, "stride 4736");
  run<4, 1>(512, 768, "stride 768");
  run<4, 1, 1>(512, 16, "AGPR acc");
  run<4, 2, 1>(512, 16, "AGPR acc, 2 acc");
  run<0, 1, 1>(512, 16, "AGPR acc, no reads");
  run<4, 1, 2>(512, 16, "reads between MFMAs");
  run<4, 2, 2>(512, 16, "reads between, 2 acc");
  run<4, 5, 3>(512, 16, "between + weights, 5 acc");
  run<4, 1, 3>(512, 16, "between + weights, 1 acc");
  run<0, 5, 3>(512, 16, "weights only, 5 acc");
  run<4, 5, 3>(256, 16, "1 wave/SIMD: between + weights, 5 acc");
  run<4, 1, 2>(256, 16, "1 wave/SIMD: reads between, 1 acc");
  run<4, 2, 2>(256, 16, "1 wave/SIMD: reads between, 2 acc");
  run<4, 1, 0>(256, 16, "1 wave/SIMD: reads in front");
  return 0;
}
