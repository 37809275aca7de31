// Shared helpers for libmagat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/magat_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MAGAT_WAVE 64
#define MAGAT_TILE_ROWS 128   // agent-tile height of the tile-major activation layout
// element offset of row m: (m / 128) * tile_stride + (m % 128) * ld
__host__ __device__ __forceinline__ long long magat_row_off(long long m, long long ld, long long tile_stride) {
  return (m >> 7) * tile_stride + (m & 127) * ld;
}
#define MAGAT_NUM_XCD 8

// per-kernel timing hooks (profile.hip); tags are listed in include/magat_hip.h
int magat_prof_begin(int tag, hipStream_t st);
void magat_prof_end(int id, hipStream_t st);

// bf16x6 split-MFMA GEMM (conv_gemm_bf16x6.hip), reached through magat_conv_gemm_f32 when desc->in_fmt == 1
int magat_conv_gemm_bf16x6(const magat_conv_gemm_desc* d, hipStream_t st);
int magat_layer1_fused(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* out,
                       void* ctr, int M, int H, int W, hipStream_t st);   // layer1_fused.hip
size_t magat_layer1_fused_lds(int W);   // 0: the fused kernel does not take this map width
int magat_conv_gemm_f16x3_pair(const magat_conv_gemm_desc* d, hipStream_t st);   // conv_gemm_f16x3_pair.hip
int magat_conv_gemm_f16x3_duo(const magat_conv_gemm_desc* d, hipStream_t st);    // conv_gemm_f16x3_duo.hip
int magat_conv_direct_enabled();   // f16x3 direct kernel on (MAGAT_CONV_DIRECT, default 1)

// packed GAT weights: [Bt NC*G | colbias NC | pad to 4][bf16x3 planes 3*NC*G u16 | pad to 4 floats][f16x2 planes of
// Bt * 2^8: 2*NC*G u16][float 2^-8][pad]: float offset of the f16 block
__host__ __device__ inline size_t magat_gat_f16_block_offset(int NC, int G) {
  const size_t a = (((size_t)NC * (G + 1) + 3) & ~(size_t)3) + ((size_t)3 * NC * G + 1) / 2;
  return (a + 3) & ~(size_t)3;
}

// hoisted GAT maps Z [M][ldz >= NC] = X [M][G] @ Bt^T + colbias from the packed weights (gat_f32.hip): bf16x6 split when
// NC % 32 == 0 and G % 32 == 0, else fp32 MFMA
int magat_gat_maps_gemm(const float* X, const float* packed, float* Z, int M, int G, int NC, int ldz, void* stream,
                        long long ntile_stride = 0);   // > 0: Z in 128-column tiles ntile_stride floats apart (row stride 128)

// fp32 -> three bf16 planes (round-to-nearest-even each)
__device__ __forceinline__ unsigned short magat_bf16_rne(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float magat_bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// sparse-structure GAT kernel (gat_list_f32.hip), driven by magat_gat_forward_packed_f32
int magat_gat_list_capacity(int N, int G, int F);
size_t magat_gat_list_workspace_bytes(int B, int N, int G, int F);
int magat_gat_list_run(const float* X, const void* S, int s_is_f64, const float* Z, const float* bias, float* Y,
                       int ldy, float* A_opt, void* ws, int B, int b0, int N, int G, int K, int P, int mode,
                       int concat, int NC, int qoff, int uoff, int c1off, int c2off, int** over_out, hipStream_t st);

static inline int magat_check_launch() {
  return hipGetLastError() == hipSuccess ? MAGAT_OK : MAGAT_ERR_LAUNCH;
}

static inline size_t magat_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// DPP cross-lane moves run at VALU speed (no LDS crossbar round trip like ds_bpermute / __shfl).
// row_ror:n rotates inside each 16-lane row: four of them all-reduce a row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x121>(v));
  return v;
}
// all-reduce over aligned 8-lane groups: xor 1, xor 2 inside quads, then mirror the half row
__device__ __forceinline__ float oct_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  return v;
}
__device__ __forceinline__ float oct_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// wave-uniform results (4 rows combined through scalar registers)
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
