// Stand-alone check that gfx950 executes the LDS-direct load (global_load_lds_dwordx4, M0 = LDS byte base, lane i -> +16 i)
// as gat_f32.hip uses it.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/ldsdma_test.hip -o tools/exp/ldsdma_test; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each wave copies 64 x 16 B from global to LDS with the LDS-direct load, then the block dumps LDS to out
__global__ void k(const float* __restrict__ in, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // wave w loads rows (2w, 2w+1) of a [rows][128] tile: lane -> row 2w + lane/32, chunk lane%32; source rows are 2048 floats apart
  const float* src = in + (long long)(2 * wave + lane / 32) * 2048 + 4 * (lane % 32);
  const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds + wave * 256));
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(ldsbase) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 256 * 4 * blockDim.x / 64 / 4; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int waves = 4, rows = 2 * waves;
  std::vector<float> h(rows * 2048);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *din, *dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, rows * 128 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), waves * 1024, 0, din, dout);
  std::vector<float> o(rows * 128);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rows; ++r) for (int c = 0; c < 128; ++c) if (o[r * 128 + c] != h[r * 2048 + c]) { if (bad < 5) printf("bad r%d c%d got %f want %f\n", r, c, o[r*128+c], h[r*2048+c]); ++bad; }
  printf("bad=%d\n", bad);
  return bad != 0;
}
