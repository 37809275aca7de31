// Shared helpers for libmagat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/magat_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MAGAT_WAVE 64
#define MAGAT_NUM_XCD 8

// per-kernel timing hooks (profile.hip); tags are listed in include/magat_hip.h
int magat_prof_begin(int tag, hipStream_t st);
void magat_prof_end(int id, hipStream_t st);

static inline int magat_check_launch() {
  return hipGetLastError() == hipSuccess ? MAGAT_OK : MAGAT_ERR_LAUNCH;
}

static inline size_t magat_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
