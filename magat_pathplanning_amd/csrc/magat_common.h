// Shared helpers for libmagat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/magat_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Timing-experiment switches (*_WHATIF_*: a phase of a kernel replaced by a register sink or a constant - wrong results by
// design; tools/whatif_*.sh, tools/csr_layer_bench.py) compile only into an EXPERIMENT build: the source must be compiled
// with -DMAGAT_EXPERIMENT_BUILD, which marks the library (magat_build_flavor() = 1; the Python binding refuses to load it
// unless MAGAT_ALLOW_EXPERIMENT_BUILD=1).  A release build - build_native.build() without --debug - never passes extra
// flags, so a stray environment variable cannot produce a silently wrong libmagat_hip.so.
#if (defined(MAGAT_WHATIF_NO_W) || defined(MAGAT_WHATIF_NO_LDS) || defined(CSR_WHATIF_NODMA) || defined(CSR_WHATIF_NOLOOP) || defined(CSR_WHATIF_NOIDX) || defined(CSR_WHATIF_NOKLOOP) || defined(CSR_WHATIF_NOSTORE) || \
     defined(GM_WHATIF_NOQW) || defined(GM_WHATIF_NOAW) || defined(GM_WHATIF_NOYST) || defined(FUSED_WHATIF_COALESCED) || defined(FUSED_WHATIF_NOSCALAR)) && !defined(MAGAT_EXPERIMENT_BUILD)
#error "*_WHATIF_* timing switches produce wrong results: they compile only with -DMAGAT_EXPERIMENT_BUILD (see magat_common.h)"
#endif
extern "C" int magat_experiment_mark(void);      // options.hip
#ifdef MAGAT_EXPERIMENT_BUILD
namespace { struct MagatExperimentMark { MagatExperimentMark() { magat_experiment_mark(); } } g_magat_experiment_mark; }
#endif

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
void magat_form_note(int id);      // which form a launch took (MAGAT_FORM_*; profile.hip)

// bf16x6 split-MFMA GEMM (conv_gemm_bf16x6.hip), reached through magat_conv_gemm_f32 when desc->in_fmt == 1
int magat_conv_gemm_bf16x6(const magat_conv_gemm_desc* d, hipStream_t st);
int magat_layer1_fused(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* out,
                       void* ctr, int M, int H, int W, hipStream_t st, int* range_flag = nullptr);   // layer1_fused.hip
size_t magat_layer1_fused_lds(int W);
int magat_stem8(const float* x, const float* w0, const float* b0, const float* w1f, const float* b1, void* out, void* ctr,
                int M, int H, int W, hipStream_t st, int* range_flag = nullptr);      // block_fused.hip: the 8-agent-group form   // 0: the fused kernel does not take this map width
// BasicBlock chain kernel (block_fused.hip): layer1.conv2+ds -> layer2.conv1 -> layer2.conv2+ds on 6x6 maps, maps in LDS
size_t magat_block_chain_weight_floats();
size_t magat_block3_weight_floats();
int magat_block_full(const void* in1, const void* in2, const float* wchain, const float* bA, const float* bB, const float* bC,
                     float* out, const float* w3, const float* b1, const float* b2, int M, int* range_flag, hipStream_t st,
                     const float* scales = nullptr,       // 5 device floats replacing the 1 / weight-scale of stages A, B, C, layer3.conv1, conv2
                     int out_gl = 0);                     // 1: pooled map granule-major (when magat_block_full_out_gl())
int magat_block_full_out_gl();
// the same chain with ONE agent per workgroup (block_lat.hip: the latency form of few-agent calls; bit-identical pooled map)
struct magat_lat_head {      // ... with the encoder head (9 x 128 -> 128) and compressMLP (128 -> 128) in its epilogue
  const float* hfrag; const float* cfrag;       // fragment-major f16 planes + [2^-e, 0, 0, 0] (encoder.pack_frag_natural)
  const float* hbias; const float* cbias;
  const float* insc; const float* insc2;        // activation scales of the two layers' inputs (device floats; null / 0 = 1)
  float* feat; int ldfeat; float* comp; int ldcomp;
  int ncomp;                                    // compressMLP's outputs: 32 | 64 | 128 (the published bottleneck widths)
};
struct magat_lat_guard {     // ... and the encoder's range guard inside the same launch (float32 re-computation per agent)
  const float* x;            // raw state maps [M][3][11][11]
  const float* pack; const int64_t* off;      // the encoder pack and its float offsets 0..17 (float32 BN-folded weights)
  int* book;                 // status block of the workspace (magat_guard_book's words)
};
struct magat_lat_stem {      // ... and the stem + layer1.conv1 in front (stem8_kernel's arithmetic): in1 / in2 are not read then
  const float* x;            // raw state maps [M][3][11][11]
  const float* w0; const float* b0;      // stem weights [32][27] and bias (magat_stem8's arguments)
  const float* w1f; const float* b1;     // layer1.conv1 fragment-major + bias
};
int magat_block_lat(const void* in1, const void* in2, const float* wchain, const float* bA, const float* bB, const float* bC,
                    float* out, const float* w3, const float* b1, const float* b2, int M, int* range_flag, hipStream_t st,
                    const float* scales = nullptr, int out_gl = 0, const magat_lat_head* head = nullptr,
                    const magat_lat_guard* guard = nullptr, const magat_lat_stem* stem = nullptr);
int magat_block3(const void* in, float* out, const float* w, const float* b1, const float* b2, int M, int* range_flag,
                 hipStream_t st);
int magat_block_chain(const void* in1, const void* in2, void* out, int out_gl, long long out_pix_stride, long long out_tile,
                      const float* w, const float* bA, const float* bB, const float* bC, int M, int* range_flag,
                      hipStream_t st);
int magat_conv_direct_enabled();   // 1 (the f16x3 GEMM is the register-direct kernel)
// float32 layers of one agent range chained in ONE launch (conv_gemm_f32.hip; the range guard's re-run)
int magat_conv_gemm_chain_f32(const magat_conv_gemm_desc* descs, int n, const int32_t* run_if, int tag, hipStream_t st,
                              int32_t* book = nullptr);

// Library options (options.hip): read from the environment (MAGAT_<NAME>) ONCE, changed at run time through
// magat_set_option - nothing on the launch path calls getenv.
enum MagatOpt {
  MAGAT_OPT_ENC_CHUNK, MAGAT_OPT_CONV_SPLIT, MAGAT_OPT_CONV_PCHAIN,
  MAGAT_OPT_L1_FUSED, MAGAT_OPT_HEAD_SPLITK, MAGAT_OPT_GAT_CHUNK_MB, MAGAT_OPT_GAT_SPLIT, MAGAT_OPT_RANGE_GUARD,
  MAGAT_OPT_BLOCK_FUSED, MAGAT_OPT_CSR_TILED, MAGAT_OPT_HEAD_F16, MAGAT_OPT_GAT_MFMA, MAGAT_OPT_SKINNY, MAGAT_OPT_GAT_PACK,
  MAGAT_OPT_CONV_BNFILL, MAGAT_OPT_HEAD_COMPRESS, MAGAT_OPT_CONV_TM, MAGAT_OPT_CSR_FUSED, MAGAT_OPT_LAT_AGENTS, MAGAT_OPT_GAT_WIDE_FROM, MAGAT_OPT_COUNT
};
int magat_opt(int id);
// hipFuncAttributeMaxDynamicSharedMemorySize, remembered per (kernel slot, device)
#define MAGAT_LDS_SLOTS 192
int magat_ensure_dyn_lds(const void* func, int slot, size_t bytes);
enum MagatLdsSlot {
  MAGAT_LDS_GAT16, MAGAT_LDS_GAT32, MAGAT_LDS_GAT64, MAGAT_LDS_GAT128, MAGAT_LDS_GAT256, MAGAT_LDS_L1FUSED,
  MAGAT_LDS_SIM_GSO_T, MAGAT_LDS_SIM_GSO_F, MAGAT_LDS_SIM_MOVE, MAGAT_LDS_BLOCK_A, MAGAT_LDS_BLOCK_B, MAGAT_LDS_BLOCK_C,
  MAGAT_LDS_SIM_CONN, MAGAT_LDS_CONV_FIRST, MAGAT_LDS_CONV_FIRST11, MAGAT_LDS_GSO_STRUCT, MAGAT_LDS_CSR_TILED_A, MAGAT_LDS_CSR_TILED_B, MAGAT_LDS_CSR_TILED_A16, MAGAT_LDS_CSR_TILED_B16,
  MAGAT_LDS_CSR_TILED_A4, MAGAT_LDS_CSR_TILED_B4, MAGAT_LDS_CSR_TILED_A16_4, MAGAT_LDS_CSR_TILED_B16_4, MAGAT_LDS_BLOCK_B4, MAGAT_LDS_BLOCK_FULL, MAGAT_LDS_BLOCK_FULL_P, MAGAT_LDS_BLOCK_FULL_C, MAGAT_LDS_BLOCK_FULL_C4, MAGAT_LDS_BLOCK_FULL_C5,
  MAGAT_LDS_GATM_0,      // gat_mfma.hip: 24 slots (score mode x shape class x taps x merge)
  MAGAT_LDS_GATM_END = MAGAT_LDS_GATM_0 + 24,
  MAGAT_LDS_STEM8,
  MAGAT_LDS_GATP_0,      // gat_mfma.hip PACK form: 8 slots (score mode x taps x merge)
  MAGAT_LDS_GATP_END = MAGAT_LDS_GATP_0 + 8,
  MAGAT_LDS_GATS_0,      // gat_small.hip: 8 slots (width x taps x merge)
  MAGAT_LDS_GATS_END = MAGAT_LDS_GATS_0 + 8,
  MAGAT_LDS_CSR_FUSED_A,  // gat_csr_fused.hip: score kernel, P = 1 | 2 | 4
  MAGAT_LDS_CSR_FUSED_B = MAGAT_LDS_CSR_FUSED_A + 3,   // hop + tap kernel, 1 | 2 heads per workgroup
  MAGAT_LDS_CSR_FUSED_END = MAGAT_LDS_CSR_FUSED_B + 2,
  MAGAT_LDS_GATD_0 = MAGAT_LDS_CSR_FUSED_END,      // gat_mid.hip: 24 slots (width x taps x row tiles x merge), 24 more for the head-split form, 8 of the 128-wide form
  MAGAT_LDS_GATD_END = MAGAT_LDS_GATD_0 + 56,
  MAGAT_LDS_BLOCK_LAT, MAGAT_LDS_BLOCK_LAT_H, MAGAT_LDS_BLOCK_LAT_S,   // block_lat.hip (chain only / + head / + stem)
  MAGAT_LDS_GAT_SLIM,    // gat_f32.hip: gat_slim_kernel
  MAGAT_LDS_GAT_RERUN_S, MAGAT_LDS_GAT_RERUN_S_END = MAGAT_LDS_GAT_RERUN_S + 5,    // gat_rerun_small_kernel<32 | 64 | 128, without | with the tail>
  MAGAT_LDS_END
};
static_assert(MAGAT_LDS_END <= MAGAT_LDS_SLOTS, "LDS attribute slots");

// packed GAT weights: [Bt NC*G | colbias NC | pad to 4][bf16x3 planes 3*NC*G u16 | pad to 4 floats][f16x2 planes of
// Bt * 2^8: 2*NC*G u16][float 2^-8][pad]: float offset of the f16 block
__host__ __device__ inline size_t magat_gat_f16_block_offset(int NC, int G) {
  const size_t a = (((size_t)NC * (G + 1) + 3) & ~(size_t)3) + ((size_t)3 * NC * G + 1) / 2;
  return (a + 3) & ~(size_t)3;
}

// fragment-major f16x2 planes of Bt * 2^8 in 128-row blocks (G = 128, NC % 128 == 0; gat_mfma.hip): float offset behind the
// row-major planes; NC * G more floats
__host__ __device__ inline size_t magat_gat_frag_offset(int NC, int G) {
  return (magat_gat_f16_block_offset(NC, G) + (size_t)NC * G + 4 + 3) & ~(size_t)3;
}
// bf16-storage CSR layer with the maps inside the graph kernels (gat_csr_fused.hip: KeyQuery, K = 2, G = F = 128, concat):
// the fragment-major bf16 weights sit behind the one-launch kernel's fragments in the packed block (NC * G / 2 more floats)
__host__ __device__ inline size_t magat_gat_csr_fused_offset(int NC, int G) {
  return (magat_gat_frag_offset(NC, G) + (size_t)NC * G + 3) & ~(size_t)3;
}
int magat_gat_csr_fused_supported(int G, int F, int K, int P, int mode, int concat);
int magat_gat_csr_fused_pack(const float* Bt, void* frag_out, int P, hipStream_t st);
size_t magat_gat_csr_fused_order_bytes(int B, int N);
int magat_gat_csr_fused_forward(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr, const int* cscsrc,
                                const int* cscpos, long long nnz, const void* frags, const float* bias, void* Y, int ldy,
                                int y_f32, float* att /* workspace, [nnz][P] */, float* att_opt /* [P][nnz] or null */, int* order, int B, int N,
                                int P, hipStream_t st);
int magat_cast_rows_if(const void* src, void* dst, int to_bf16, long long M, int width, int ld_src, int ld_dst, void* stream,
                       const int32_t* run_if);      // gat_csr_f32.hip: magat_cast_rows, predicated on a device flag (null = always)
// one-launch KeyQuery layer for small graphs and narrow features (gat_small.hip: N <= 32, G = F in {32, 64}, K = 2 | 3)
int magat_gat_small_supported(int N, int G, int F, int K, int mode);
int magat_gat_small_forward(const float* X, int ldx, const void* S, int s_is_f64, const unsigned* rmask_pre, const float* Hs,
                            int NC, const float* bias, float* Y, int ldy, int B, int N, int G, int K, int P, int concat,
                            int* range_flag, hipStream_t st, const float* x_scale);
// one-launch KeyQuery layer for the published widths on graphs of 33 .. 128 agents (gat_mid.hip: G = F in {32, 64}, K = 2 | 3)
int magat_gat_mid_supported(int N, int G, int F, int K, int mode);
int magat_gat_mid_forward(const float* X, int ldx, const void* S, int s_is_f64, const float* Hs, int NC, const float* bias, float* Y,
                          int ldy, int B, int N, int G, int K, int P, int concat, int* range_flag, hipStream_t st,
                          const float* x_scale,
                          float* ypre = nullptr, int ldpre = 0, const float* wfrag = nullptr);      // head-mean scratch rows [B*N][P F]: lets few instances run a workgroup per head
// one-launch KeyQuery layer on the matrix cores (gat_mfma.hip)
int magat_gat_mfma_supported(int N, int G, int F, int K, int mode);
int magat_gat_mfma_forward(const float* X, int ldx, const void* S, int s_is_f64, const unsigned* rmask_pre,
                           const float* packed_frag, const float* bias, float* Y, int ldy, int B, int N, int K, int P,
                           int concat, int* range_flag, hipStream_t st,
                           const float* x_scale = nullptr,       // device float: power-of-two scale of X's planes (null / 0 = 1)
                           int mode = MAGAT_MODE_KEYQUERY,       // GAT_modified / GAT_origin: rank-1 scores (packed_frag: the rank-1 block)
                           const float* kconst = nullptr,        // per head a1 . wb + a2 . wb (device floats), or null
                           float* ypre = nullptr, int ldpre = 0); // head-mean scratch rows [B*N][>= 128 P]: lets few instances run a workgroup per head
int magat_gat_mean_launch(const float* ypre, float* y, long long M, int P, int F, int ldpre, int ldy, hipStream_t st);      // gat_mid.hip

// hoisted GAT maps Z [M][ldz >= NC] = X [M][G] @ Bt^T + colbias from the packed weights (gat_f32.hip): bf16x6 split when
// NC % 32 == 0 and G % 32 == 0, else fp32 MFMA
int magat_gat_maps_gemm(const float* X, const float* packed, float* Z, int M, int G, int NC, int ldz, void* stream,
                        long long ntile_stride = 0,    // > 0: Z in 128-column tiles ntile_stride floats apart (row stride 128)
                        int32_t* status = nullptr,     // range guard status words (device int32[2]) or null: unguarded
                        int force_f32 = 0);            // 1: float32 MFMA kernel (training: Z feeds the backward)

// fp32 -> three bf16 planes (round-to-nearest-even each)
__device__ __forceinline__ unsigned short magat_bf16_rne(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float magat_bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ReLU of the float32 kernels: hands a NaN on like torch.relu does (fmaxf(NaN, 0) would turn it into 0) - the range guard
// sends non-finite inputs to these kernels so that they reach the logits as they do in the reference
__device__ __forceinline__ float magat_relu(float v) { return v < 0.f ? 0.f : v; }

// Range-guard bookkeeping without a launch of its own: the LAST predicated launch that reads a forward's flag calls this from
// every workgroup after its last read of book[0] (also on its early-exit paths); the workgroup that arrives last moves the
// flag to book[2], counts the re-run in book[1] and clears book[0] for the next forward (book[6]: arrival counter, zero
// between launches).  Block-uniform call sites only.
__device__ __forceinline__ void magat_guard_book(int* book) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned prev = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(book) + 6, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == total - 1) {
      const int f = book[0];
      book[2] = f;
      if (f != 0) book[1] += 1;
      book[0] = 0;
      book[6] = 0;
    }
  }
}

// ... and the launch that finds the flag CLEAR (every forward of a sane checkpoint): nothing to count and nothing to clear -
// book[0] is zero already and stays so, only "this forward's flag" becomes 0.  One plain store by one workgroup instead of a
// barrier, an agent-scope atomic with its cache write-back / invalidate per workgroup and the last-arrival test: the no-op
// launches of the guard cost 5-11 us each with that (round 5).
__device__ __forceinline__ void magat_guard_book_idle(int* book) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) book[2] = 0;
}

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
