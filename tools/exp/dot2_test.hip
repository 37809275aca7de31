// what does __builtin_amdgcn_fdot2_f32_bf16 compute on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* o) {
  float c = 0.f;
  c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a[threadIdx.x]), __builtin_bit_cast(bf2, b[threadIdx.x]), c, false);
  o[threadIdx.x] = c;
}
static unsigned bf(float x) { unsigned u; memcpy(&u, &x, 4); return u >> 16; }
int main() {
  unsigned ha[4] = {bf(1.0078125f) | (bf(0.f) << 16), bf(1.0078125f) | (bf(1.0078125f) << 16), bf(3.140625f) | (bf(-3.140625f) << 16), bf(1e-20f) | (bf(1e20f) << 16)};
  unsigned hb[4] = {bf(1.0078125f) | (bf(0.f) << 16), bf(1.0078125f) | (bf(1.0078125f) << 16), bf(1.0078125f) | (bf(1.0f) << 16), bf(1e-20f) | (bf(1e-20f) << 16)};
  unsigned *a, *b; float* o;
  hipMalloc(&a, 16); hipMalloc(&b, 16); hipMalloc(&o, 16);
  hipMemcpy(a, ha, 16, hipMemcpyHostToDevice); hipMemcpy(b, hb, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, a, b, o);
  float ho[4]; hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
  printf("got %.9g %.9g %.9g %.9g  (expect 1.01568604 2.03137207 0.0245361328 1)\n", ho[0], ho[1], ho[2], ho[3]);
  return 0;
}
