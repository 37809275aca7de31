// BasicBlock chain kernel: layer1.conv2(+downsample) -> layer2.conv1 -> layer2.conv2(+downsample) of the ResNet encoders
// (reference graphs/models/resnet_pytorch.py:40-73 BasicBlock, :495-524 forward) in ONE launch, with every intermediate
// 6x6 map of a small agent group resident in LDS.  f16x3 arithmetic (conv_gemm_bf16x6.hip): every value is two
// half-precision planes, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation.
//
// Layer by layer these three convolutions moved 3.35 GB through HBM per 51 200-agent step (every map written by one kernel and
// read back - up to nine times - by the next) and ran at 18-33 % of the matrix peak.  Here a workgroup owns EIGHT agents:
//   rows of the implicit GEMMs = (pixel, agent) pairs, 36 x 8 = 288 rows = 9 MFMA row tiles of 32.  The tiles are formed by
//   tap-validity class - 4 x interior (all 9 taps), top / bottom / left / right edge (6 taps each), the 4 corners (9 taps,
//   invalid ones read a zero pixel) - so that a tile skips a tap outright when it falls into the zero padding for all of
//   its pixels: 69 tile-taps are executed per channel tile where 64 are useful (exact skipping needs 32 agents per pixel,
//   whose maps do not fit LDS).
//   LDS map layout [plane][8-channel chunk][pixel slot 0..36][agent][8 halves]: a row's MFMA operand for any tap is one
//   ds_read_b128 at (lane base + compile-time offset); pixel slot 36 is a zero pixel.  Maps: the two 32-channel inputs
//   (layer1.conv1 output, stem stride-2 pixels), layer1's output, layer2.conv1's output (aliasing the dead inputs): 111 KB.
//   Weights are pre-packed fragment-major (encoder.pack_chain_weights): the MFMA row operand of (channel tile, tap, k step,
//   plane) is one contiguous 1 KB block, fetched global -> registers one tap ahead.  No barrier inside a convolution: the
//   maps are read-only while a stage runs, the eight waves drift freely (wave = channel tile x group of row tiles with
//   balanced tap counts); three workgroup barriers per launch.
// Output: layer2's map as f16 plane granules (magat_hip.h in_gl = 2) for layer3.conv1, or float32 row-major tiles for the
// pooled head (ResNetSlim).
#include "magat_common.h"

namespace {
// MAGAT_CHAIN_NT: the one-launch kernel's activation traffic (input maps in, pooled map out) marked non-temporal, so that the
// 1.8 MB of weight fragments every CU of an XCD re-reads per agent group are what its L2 keeps
#ifndef MAGAT_CHAIN_NT
#define MAGAT_CHAIN_NT 3      /* bit 0: the input maps, bit 1: the pooled map */
#endif
#if MAGAT_CHAIN_NT & 1
#define MAGAT_CHAIN_NT_STR " nt"
#else
#define MAGAT_CHAIN_NT_STR ""
#endif
#if MAGAT_CHAIN_NT & 2
#define CHAIN_OUT_STORE(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<f32x4*>(ptr))
#else
#define CHAIN_OUT_STORE(ptr, val) (*reinterpret_cast<f32x4*>(ptr) = (val))
#endif
#include "block_walk.h"
struct ChainParams {
  const char* in1;        // layer1.conv1 output, f16 plane granules, 32 channels: [agent tile][pixel][128 agents x 128 B]
  const char* in2;        // stem output at the stride-2 pixels, same geometry
  char* out;              // layer2 output
  int out_gl;             // 2: f16 plane granules (64 channels); 0: float32 row-major agent tiles
  long long out_pix_stride, out_tile;     // floats
  const char* wA; const char* wB; const char* wC;      // fragment-major f16 weight planes (+ float 2^-e at the end of each)
  const float* bA; const float* bB; const float* bC;   // biases (conv2 + downsample already added)
  const float* sA; const float* sB; const float* sC;     // 2^-e of each weight block (one float behind it)
  int M, groups;
  int* range_flag;
  long long* dbg;         // MAGAT_DEBUG_HOOKS builds only: [groups][8] phase timestamps of wave 0
};

#ifdef MAGAT_DEBUG_HOOKS
#define CHAIN_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long long)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
long long* g_chain_dbg = nullptr;
long long* g_block3_dbg = nullptr;
#else
#define CHAIN_STAMP(i) do { } while (0)
#endif
#ifdef MAGAT_DEBUG_HOOKS
#define FULL_STAMP(i) do { if (l3.dbg && (threadIdx.x & 63) == 0) l3.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FULL_STAMP(i) do { } while (0)
#endif
// the workgroup's core-clock count next to the constant 100 MHz counter, at its start (k = 0) and end (k = 1): row "wave 4" of
// the debug buffer - the clock the chip actually holds under this kernel (tools/chain_phase_probe.py prints it)
#ifdef MAGAT_DEBUG_HOOKS
#define FULL_CLOCKS(k) do { if (l3.dbg && threadIdx.x == 0) { long long* d_ = l3.dbg + ((long long)blockIdx.x * 8 + 4) * 16 + 2 * (k); \
    d_[0] = (long long)__builtin_readcyclecounter(); d_[1] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define FULL_CLOCKS(k) do { } while (0)
#endif
// per-wave phase stamps of the layer3 kernel (debug build): [workgroup][wave 8][16], before and after every barrier
#ifdef MAGAT_DEBUG_HOOKS
#define L3_STAMP(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define L3_STAMP(i) do { } while (0)
#endif


// ---------------------------------------------------------------------------------------------------------------------------
// layer3 of ResNetLarge in one launch: conv1 (64 -> 128) -> conv2 (128 -> 128) + downsample(64 -> 128) -> ReLU -> 2x2 sum-pool.
// Same row tiling as above, 8 agents per workgroup.  The 128-channel intermediate map (148 KB for 8 agents) does not fit LDS
// next to the input, so it is produced and consumed in two 64-channel HALVES: conv1 half h -> LDS, conv2 accumulates its K
// segment over those 64 channels into register accumulators that live across both halves (wave = output channel tile x one
// of two row-tile groups: 69 tile-taps on every SIMD, 80 accumulator registers).  The output never reaches HBM un-pooled:
// ReLU'd values go to an LDS scratch (the two map regions, dead by then) and the AvgPool2d(2) sums - the 1/4 lives in the
// head's weights - are written as [cell 9][agent][128] float32: 0.24 GB per 51 200 agents instead of 0.94 GB written by the
// convolution and 0.94 GB read back by the head.
struct L3Params {
  const char* in;         // layer2 output, f16 plane granules, 64 channels: [agent tile][pixel][128 agents x 256 B]
  float* out;             // pooled float32 [agent tile][cell 9][128 agents][128]
  const char* w1;         // conv1: [ct 4][tap 9][ks 4][plane 2] 1 KB blocks, then float 2^-e
  const char* w2a;        // conv2, K over intermediate channels 0..63:   [ct 4][tap 9][ks 4][plane 2]
  const char* w2b;        // conv2, K over channels 64..127 + residual:   [ct 4][tap 9 | residual][ks 4][plane 2], then float 2^-e
  const float* b1; const float* b2;
  const float* s1; const float* s2;
  int M, groups;
  int* range_flag;
  long long* dbg;
  int out_gl;             // block_full_p_kernel: 1 = the pooled map granule-major, [agent tile][cell 9][128 / 4][128 agents][4]
};



// relu(acc * scale + bias) of a wave's row tiles as f16 plane chunks of an LDS map with COUT channels (channel tile ct)
template <int COUT, int NS>
__device__ __forceinline__ void epi_to_lds(char* lds, int out_off, const int (&tl)[NS], const f32x16 (&acc)[NS], int ct,
                                           const float* bias, float scale, bool rows_ok, bool& clamped) {
  constexpr int PS_OUT = (COUT / 8) * BLK;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bias + 32 * ct + 8 * q + 4 * fh);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (tl[s] < 0) continue;
    const int pix = tile_pix(tl[s], psl);
    float cl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned h1[4], h2[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * ks + e;
        // (one packed fma per pair - these epilogues run with the matrix pipe idle, where v_pk_fma_f32 issues like any other
        //  vector instruction; ReLU is the lower clamp of split2)
        const f32x2 v01 = __builtin_elementwise_fma(f32x2{acc[s][4 * q], acc[s][4 * q + 1]}, f32x2{scale, scale}, f32x2{bq[q][0], bq[q][1]});
        const f32x2 v23 = __builtin_elementwise_fma(f32x2{acc[s][4 * q + 2], acc[s][4 * q + 3]}, f32x2{scale, scale}, f32x2{bq[q][2], bq[q][3]});
        split2(v01[0], v01[1], h1[2 * e], h2[2 * e], cl);
        split2(v23[0], v23[1], h1[2 * e + 1], h2[2 * e + 1], cl);
      }
      char* o = lds + out_off + ((ct * 2 + ks) * 2 + fh) * BLK + pix * PIXB + agent * 16;
      *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
      *reinterpret_cast<u32x4*>(o + PS_OUT) = u32x4{h2[0], h2[1], h2[2], h2[3]};
    }
    clamped |= cl > 65504.f && rows_ok;
  }
}




// ---------------------------------------------------------------------------------------------------------------------------
// layer3, "w4" form: FOUR waves per workgroup, one per SIMD, each with the whole 512-entry register budget (256 arch + 256
// accumulation registers).  conv2: wave = one 32-channel output tile x ALL nine row tiles (144 accumulator registers, live
// across both halves) - every weight fragment is fetched by exactly one wave of the workgroup (the 8-wave form fetches it
// twice for conv2, four times for conv1: the vector-memory return path was 72 % busy).  conv1: wave = (channel tile of the
// half, one of two row-tile groups).  Every wave role is a compile-time list of (tap, k step, row tile) items, so the K walk is
// straight-line code with immediate LDS offsets: no tap loop, no branches, no address arithmetic except for the corner tile;
// the operand reads of item i + 2 sit between the MFMAs of item i (measured, tools/exp/mfma_lds.hip: one wave per SIMD sustains
// 1.9 PF that way, 1.67 PF with the reads in front of the MFMA run), weights come 3 k steps ahead.
__global__ __launch_bounds__(256, 1) void block3_w4_kernel(const L3Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int L_IN = 0, L_MID = MAP64;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  L3_STAMP(11);
  for (int i = t; i < 32 * (PIXB / 4); i += 256)
    *reinterpret_cast<unsigned*>(lds + (i / (PIXB / 4)) * BLK + ZPIX * PIXB + (i % (PIXB / 4)) * 4) = 0u;
  // input map: 16 (plane, chunk) blocks x 36 pixels x 128 B of an agent group = 80 LDS-direct instructions (an expensive
  // instruction to issue, ~150 cycles each): items first, first + step, ... < last
  auto dma_in = [&](int group, int first, int step, int last) {
    const int m0 = group * AG;
    const long long tile_b = (long long)(m0 >> 7) * NPIX * (128 * 64 * 4) + (m0 & 127) * 16;
    for (int item = first; item < last; item += step) {
      const int blk = item / 5, part = item % 5;             // blk = plane * 8 + chunk
      const int pix = part * 8 + (lane >> 3);
      const char* src = p.in + tile_b + (long long)pix * (128 * 64 * 4) + (blk >> 3) * (256 * 64) + (blk & 7) * 2048 +
                        (lane & 7) * 16;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)(L_IN + blk * BLK + part * 8 * PIXB));
      if (pix < NPIX) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  };
  const float s1 = *p.s1, s2 = *p.s2;
  bool clamped = false;
  const int ct1 = wave & 1, rg1 = wave >> 1, ct2 = wave;
  constexpr int BPT1 = 9 * 4 * 2, BPT2A = 9 * 4 * 2, BPT2B = (9 * 4 + 4) * 2;      // 1 KB blocks per channel tile
#ifdef MAGAT_B3W4_ONESHOT
  const int gstride = 1 << 30;
#else
  const int gstride = (int)gridDim.x;       // PERSISTENT: the next group's input streams in under the output phase
#endif
  // conv2's bias: loaded once (a load issued behind the 76 KB input DMA of the output phase would wait for it to land)
  f32x4 bq[4];
  {
    const int fh = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(p.b2 + 32 * ct2 + 8 * q + 4 * fh);
  }
  if ((int)blockIdx.x < p.groups) dma_in(blockIdx.x, wave, 4, 80);
#pragma unroll 1
  for (int group = blockIdx.x; group < p.groups; group += gstride) {
    const bool rows_ok = group * AG + ((lane & 31) & 7) < p.M;
    f32x16 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    L3_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    L3_STAMP(1);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const char* w1 = p.w1 + (size_t)(2 * h + ct1) * BPT1 * 1024;
      const char* w2 = h == 0 ? p.w2a + (size_t)ct2 * BPT2A * 1024 : p.w2b + (size_t)ct2 * BPT2B * 1024;
      if (rg1 == 0) {
        f32x16 a1[W4A::NT];
#pragma unroll
        for (int s = 0; s < W4A::NT; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) a1[s][r] = 0.f;
        walk4<W4A, 4, 0, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_IN, 0, w1, a1, false);
        const int tl[W4A::NT] = {W4A::t[0], W4A::t[1], W4A::t[2], W4A::t[3]};
        epi_to_lds<64, W4A::NT>(lds, L_MID, tl, a1, ct1, p.b1 + 64 * h, s1, rows_ok, clamped);
      } else {
        f32x16 a1[W4B::NT];
#pragma unroll
        for (int s = 0; s < W4B::NT; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) a1[s][r] = 0.f;
        walk4<W4B, 4, 0, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_IN, 0, w1, a1, false);
        const int tl[W4B::NT] = {W4B::t[0], W4B::t[1], W4B::t[2], W4B::t[3], W4B::t[4]};
        epi_to_lds<64, W4B::NT>(lds, L_MID, tl, a1, ct1, p.b1 + 64 * h, s1, rows_ok, clamped);
      }
      L3_STAMP(2 + 4 * h);
      __syncthreads();
      L3_STAMP(3 + 4 * h);
      // conv2: K over these 64 intermediate channels (second half: + the residual 1x1 over the 64 input channels)
      walk4<W4All, 4, 4, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_MID, L_IN, w2, acc, h == 1);
      L3_STAMP(4 + 4 * h);
      __syncthreads();          // MID is rewritten by the next half / becomes scratch; after the second half IN is dead too
      L3_STAMP(5 + 4 * h);
    }
    const bool more = group + gstride < p.groups;
    // ReLU'd output, 64 channels per pass (waves 0-1, then 2-3) -> scratch in the MID region [pixel][agent][64 floats] (quads
    // XOR-swizzled by the row), then the 2x2 sums.  LDS-only barriers: __syncthreads() would wait for the DMA in flight.
    float* S = reinterpret_cast<float*>(lds + L_MID);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      if ((ct2 >> 1) != half) {
        // the two waves with nothing to write in this pass request half of the NEXT group's input instead
        if (more) dma_in(group + gstride, 40 * half + (wave & 1), 2, 40 * half + 40);
      } else {
        const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
          const int pix = tile_pix(W4All::t[s], psl);
          const int row = pix * AG + agent;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[s][4 * q + c] * s2 + bq[q][c], 0.f);
            const int Q = 8 * (ct2 & 1) + 2 * q + fh;
            *reinterpret_cast<f32x4*>(S + row * 64 + ((Q ^ (row & 15)) << 2)) = v;
          }
        }
      }
      L3_LDS_SYNC();
      if (half == 0) L3_STAMP(13); else L3_STAMP(15);
      for (int o = t; o < 9 * AG * 16; o += 256) {
        const int Q = o & 15, agent = (o >> 4) & 7, cell = o >> 7;
        const int cy = cell / 3, cx = cell - 3 * cy;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pix = (2 * cy + (e >> 1)) * 6 + 2 * cx + (e & 1);
          const int row = pix * AG + agent;
          sum += *reinterpret_cast<const f32x4*>(S + row * 64 + ((Q ^ (row & 15)) << 2));
        }
        const int m = group * AG + agent;
        if (m < p.M)
          *reinterpret_cast<f32x4*>(p.out + ((long long)(m >> 7) * 9 + cell) * (128 * 128) + (m & 127) * 128 + 64 * half +
                                    4 * Q) = sum;
      }
      L3_LDS_SYNC();
      if (half == 0) L3_STAMP(14);
    }
    // the scratch ran over the zero pixel slots of the MID blocks
    for (int i = t; i < 16 * (PIXB / 4); i += 256)
      *reinterpret_cast<unsigned*>(lds + L_MID + (i / (PIXB / 4)) * BLK + ZPIX * PIXB + (i % (PIXB / 4)) * 4) = 0u;
    L3_STAMP(10);
  }
  L3_STAMP(12);
  if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
}


// ---- the chain kernel in the four-wave form (see walk4): stage A = one channel tile, the nine row tiles dealt to the four
// waves; stages B and C = (channel tile, one of two row-tile groups) per wave
template <typename TL, int CIN, int C2, int COUT, bool LAST, bool SYNC_BEFORE_EPI = false>
__device__ __forceinline__ void chain_stage4(const ChainParams& p, char* lds, int in_off, int in2_off, int out_off,
                                             const char* wts, int ct, const float* bias, float scale, int group, bool& clamped) {
  constexpr int KSM = CIN / 16, KS2 = C2 / 16, NT = TL::NT;
  constexpr int BPT = (9 * KSM + KS2) * 2;                  // 1 KB weight blocks per channel tile
  constexpr int PS_OUT = (COUT / 8) * BLK;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
  f32x16 acc[NT];
#if defined(MAGAT_WHATIF_TWICE) && defined(MAGAT_DEBUG_HOOKS)
  // timing experiment (results unchanged): the walk runs twice, so that stage time (twice) - stage time (once) = the time of a
  // walk whose code is already in the instruction cache (tools/chain_phase_probe.py; DESIGN.md 4.1 "cold code")
#if MAGAT_WHATIF_TWICE == 2       // ... as two COPIES of the walk's code: the second pass finds its data warm and its code cold
  for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    asm volatile("" ::: "memory");
    if (rep == 0)
      walk4<TL, KSM, KS2, (CIN / 8) * BLK, (C2 > 0 ? C2 / 8 : 1) * BLK, w4_depth(NT, KSM)>(
          lds, in_off, in2_off, wts + (size_t)ct * BPT * 1024, acc, true);
    else
      walk4<TL, KSM, KS2, (CIN / 8) * BLK, (C2 > 0 ? C2 / 8 : 1) * BLK, w4_depth(NT, KSM)>(
          lds, in_off, in2_off, wts + (size_t)ct * BPT * 1024, acc, true);
    asm volatile("" ::: "memory");
  }
#else
#pragma unroll 1
  for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    asm volatile("" ::: "memory");
    walk4<TL, KSM, KS2, (CIN / 8) * BLK, (C2 > 0 ? C2 / 8 : 1) * BLK, w4_depth(NT, KSM)>(
        lds, in_off, in2_off, wts + (size_t)ct * BPT * 1024, acc, true);
  }
#endif
#else
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  walk4<TL, KSM, KS2, (CIN / 8) * BLK, (C2 > 0 ? C2 / 8 : 1) * BLK, w4_depth(NT, KSM)>(
      lds, in_off, in2_off, wts + (size_t)ct * BPT * 1024, acc, true);
#endif
  if (SYNC_BEFORE_EPI) __syncthreads();       // the output overwrites a map that other waves read until their walks end
  // epilogue: as chain_stage
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bias + 32 * ct + 8 * q + 4 * fh);
  const int m = group * AG + agent;
  const bool mok = m < p.M;
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int pix = tile_pix(TL::t[s], psl);
    if (LAST && p.out_gl == 0) {         // float32 row-major agent tiles [tile][pixel][128][COUT]
      if (mok) {
        float* orow = reinterpret_cast<float*>(p.out) + (long long)pix * p.out_pix_stride +
                      magat_row_off(m, COUT, p.out_tile) + 32 * ct + 4 * fh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[s][4 * q + c] * scale + bq[q][c], 0.f);
          *reinterpret_cast<f32x4*>(orow + 8 * q) = v;
        }
      }
      continue;
    }
    float cl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned h1[4], h2[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * ks + e;
        // (one packed fma per pair - these epilogues run with the matrix pipe idle, where v_pk_fma_f32 issues like any other
        //  vector instruction; ReLU is the lower clamp of split2)
        const f32x2 v01 = __builtin_elementwise_fma(f32x2{acc[s][4 * q], acc[s][4 * q + 1]}, f32x2{scale, scale}, f32x2{bq[q][0], bq[q][1]});
        const f32x2 v23 = __builtin_elementwise_fma(f32x2{acc[s][4 * q + 2], acc[s][4 * q + 3]}, f32x2{scale, scale}, f32x2{bq[q][2], bq[q][3]});
        split2(v01[0], v01[1], h1[2 * e], h2[2 * e], cl);
        split2(v23[0], v23[1], h1[2 * e + 1], h2[2 * e + 1], cl);
      }
      const int chunk = (ct * 2 + ks) * 2 + fh;
      if (LAST) {
        if (mok) {
          char* o = p.out + ((long long)pix * p.out_pix_stride + (long long)(m >> 7) * p.out_tile) * 4 + (m & 127) * 16 +
                    (long long)chunk * 2048;
          *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
          *reinterpret_cast<u32x4*>(o + 256 * COUT) = u32x4{h2[0], h2[1], h2[2], h2[3]};
        }
      } else {
        char* o = lds + out_off + chunk * BLK + pix * PIXB + agent * 16;
        *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<u32x4*>(o + PS_OUT) = u32x4{h2[0], h2[1], h2[2], h2[3]};
      }
    }
    clamped |= cl > 65504.f && mok;
  }
}

__global__ __launch_bounds__(256, 1) void block_chain_w4_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 32 * (PIXB / 4); i += 256)
    *reinterpret_cast<unsigned*>(lds + (i / (PIXB / 4)) * BLK + ZPIX * PIXB + (i % (PIXB / 4)) * 4) = 0u;
  auto dma_map = [&](const char* base, int group, int lds_off) {
    const int m0 = group * AG;
    const long long tile_b = (long long)(m0 >> 7) * NPIX * (128 * 32 * 4) + (m0 & 127) * 16;    // bytes: agent tile, agents
    for (int item = wave; item < 8 * 5; item += 4) {
      const int blk = item / 5, part = item % 5;             // blk = plane * 4 + chunk
      const int pix = part * 8 + (lane >> 3);
      const char* src = base + tile_b + (long long)pix * (128 * 32 * 4) + (blk >> 2) * (256 * 32) + (blk & 3) * 2048 +
                        (lane & 7) * 16;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)(lds_off + blk * BLK + part * 8 * PIXB));
      if (pix < NPIX) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  };
  const float sA = *p.sA, sB = *p.sB, sC = *p.sC;
  bool clamped = false;
  const int ct = wave & 1, rg = wave >> 1;
  int group = blockIdx.x;
  if (group < p.groups) dma_map(p.in1, group, LDS_X1N);
  for (; group < p.groups; group += (int)gridDim.x) {
    const bool more = group + (int)gridDim.x < p.groups;
    CHAIN_STAMP(0);
    __syncthreads();            // every wave is done reading the previous group's maps (and the zero pixels are written)
    dma_map(p.in2, group, LDS_X2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this group's main input (issued an iteration ago) + the residual input
    __syncthreads();
    CHAIN_STAMP(1);
    // A: layer1.conv2 (32 -> 32) + downsample(stem stride-2 pixels)      X1 (prefetch region), X2 -> Y
    switch (wave) {
      case 0: chain_stage4<W4P0, 32, 32, 32, false>(p, lds, LDS_X1N, LDS_X2, LDS_Y, p.wA, 0, p.bA, sA, group, clamped); break;
      case 1: chain_stage4<W4P1, 32, 32, 32, false>(p, lds, LDS_X1N, LDS_X2, LDS_Y, p.wA, 0, p.bA, sA, group, clamped); break;
      case 2: chain_stage4<W4P2, 32, 32, 32, false>(p, lds, LDS_X1N, LDS_X2, LDS_Y, p.wA, 0, p.bA, sA, group, clamped); break;
      default: chain_stage4<W4P3, 32, 32, 32, false>(p, lds, LDS_X1N, LDS_X2, LDS_Y, p.wA, 0, p.bA, sA, group, clamped); break;
    }
    CHAIN_STAMP(2);
    __syncthreads();
    CHAIN_STAMP(3);
    if (more) dma_map(p.in1, group + (int)gridDim.x, LDS_X1N);      // lands under stages B and C
    // B: layer2.conv1 (32 -> 64)                                          Y -> Z
    if (rg == 0) chain_stage4<W4A, 32, 0, 64, false>(p, lds, LDS_Y, 0, LDS_Z, p.wB, ct, p.bB, sB, group, clamped);
    else chain_stage4<W4B, 32, 0, 64, false>(p, lds, LDS_Y, 0, LDS_Z, p.wB, ct, p.bB, sB, group, clamped);
    CHAIN_STAMP(4);
    __syncthreads();
    CHAIN_STAMP(5);
    // C: layer2.conv2 (64 -> 64) + downsample(Y)                          Z, Y -> global
    if (rg == 0) chain_stage4<W4A, 64, 32, 64, true>(p, lds, LDS_Z, LDS_Y, 0, p.wC, ct, p.bC, sC, group, clamped);
    else chain_stage4<W4B, 64, 32, 64, true>(p, lds, LDS_Z, LDS_Y, 0, p.wC, ct, p.bC, sC, group, clamped);
    CHAIN_STAMP(6);
  }
  if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
}

// ---- layer1.conv2 -> layer2 -> layer3 -> ReLU -> pool in ONE persistent four-wave kernel: layer2's output map never leaves the
// CU (no 0.47 GB written and 0.6 GB read back per step, no input phase for layer3, no store epilogue for layer2.conv2), and
// both inputs of the NEXT group are requested a whole layer3 ahead (the chain kernel alone waits out an HBM round trip for its
// residual input at the top of every group).  LDS = four 37.9 KB units U0..U3; their roles alternate with the group parity:
//   X1 @ U3', X2 @ U2'   -A->  Y @ U1'   -B->  Z @ (U2', U3')   -C->  layer3's IN @ (U0', U1')     [' = rotated by 2 on odd groups]
//   layer3: IN (U0', U1'), MID (U2', U3'); pooled epilogue scratch in MID; meanwhile the next X1 -> U1', X2 -> U0' (= its U3'', U2'').
struct FullParams { ChainParams c; L3Params l; };



// ---- The one-launch kernel (option BLOCK_FUSED = 2, the default) with the 2 x 2 pooling done in REGISTERS.  The pooled epilogue of the kernel
// above costs 14 k of a group's 162 k cycles with the matrix pipe idle: the 144 accumulators go through an LDS scratch in two
// passes (scratch writes, barrier, pooling reads, barrier, zero-slot repair).  With the slot order of TILE_PIX a lane holds, for
// its 16 channels: one whole corner cell (4 registers of 4 tiles), two half cells whose other halves sit one lane-bit away,
// and a quarter of the centre cell - 11 adds, 4 selects and 4 lane exchanges per channel, no LDS, no barrier; every lane
// then stores pooled values straight from registers.  The next group's inputs land in the MID region (dead once every wave
// has left conv2), so the units keep their roles from group to group (no rotation).
__global__ __launch_bounds__(256, 1) void block_full_p_kernel(const FullParams q) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const ChainParams& p = q.c;
  const L3Params& l3 = q.l;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 32 * (PIXB / 4); i += 256)
    *reinterpret_cast<unsigned*>(lds + (i / (PIXB / 4)) * BLK + ZPIX * PIXB + (i % (PIXB / 4)) * 4) = 0u;
  auto dma_map = [&](const char* base, int group, int lds_off, int first, int step, int last) {
    const int m0 = group * AG;
    const long long tile_b = (long long)(m0 >> 7) * NPIX * (128 * 32 * 4) + (m0 & 127) * 16;    // bytes: agent tile, agents
    for (int item = first; item < last; item += step) {
      const int blk = item / 5, part = item % 5;             // blk = plane * 4 + chunk
      const int pix = part * 8 + (lane >> 3);
      const char* src = base + tile_b + (long long)pix * (128 * 32 * 4) + (blk >> 2) * (256 * 32) + (blk & 3) * 2048 +
                        (lane & 7) * 16;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)(lds_off + blk * BLK + part * 8 * PIXB));
      if (pix < NPIX) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" MAGAT_CHAIN_NT_STR ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  };
  const float sA = *p.sA, sB = *p.sB, sC = *p.sC, s1 = *l3.s1, s2 = *l3.s2;
  bool clamped = false;
  const int ct = wave & 1, rg = wave >> 1, ct2 = wave;
  constexpr int BPT1 = 9 * 4 * 2, BPT2A = 9 * 4 * 2, BPT2B = (9 * 4 + 4) * 2;      // 1 KB blocks per channel tile (layer3)
  f32x4 bq[4];                      // conv2's bias (loaded once: a load behind the input DMA would wait for it)
  {
    const int fh = lane >> 5;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const f32x4*>(l3.b2 + 32 * ct2 + 8 * qd + 4 * fh);
  }
  FULL_CLOCKS(0);
  const int gstride = (int)gridDim.x;
  constexpr int U0 = 0, U1 = MAP32, U2 = 2 * MAP32, U3 = 3 * MAP32;
  if ((int)blockIdx.x < p.groups) {
    dma_map(p.in1, blockIdx.x, U3, wave, 4, 40);
    dma_map(p.in2, blockIdx.x, U2, wave, 4, 40);
  }
#pragma unroll 1
  for (int group = blockIdx.x; group < p.groups; group += gstride) {
    const bool rows_ok = group * AG + ((lane & 31) & 7) < p.M;
    const bool more = group + gstride < p.groups;
    FULL_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this group's inputs (requested during the previous group's epilogue)
    __syncthreads();
    FULL_STAMP(1);
    // A: layer1.conv2 (32 -> 32) + downsample(stem stride-2 pixels)      X1 @ U3, X2 @ U2 -> Y @ U1
    switch (wave) {
      case 0: chain_stage4<W4P0, 32, 32, 32, false>(p, lds, U3, U2, U1, p.wA, 0, p.bA, sA, group, clamped); break;
      case 1: chain_stage4<W4P1, 32, 32, 32, false>(p, lds, U3, U2, U1, p.wA, 0, p.bA, sA, group, clamped); break;
      case 2: chain_stage4<W4P2, 32, 32, 32, false>(p, lds, U3, U2, U1, p.wA, 0, p.bA, sA, group, clamped); break;
      default: chain_stage4<W4P3, 32, 32, 32, false>(p, lds, U3, U2, U1, p.wA, 0, p.bA, sA, group, clamped); break;
    }
    __syncthreads();
    FULL_STAMP(2);
    // B: layer2.conv1 (32 -> 64)                                          Y @ U1 -> Z @ (U2, U3)
    if (rg == 0) chain_stage4<W4I, 32, 0, 64, false>(p, lds, U1, 0, U2, p.wB, ct, p.bB, sB, group, clamped);
    else chain_stage4<W4E, 32, 0, 64, false>(p, lds, U1, 0, U2, p.wB, ct, p.bB, sB, group, clamped);
    __syncthreads();
    FULL_STAMP(3);
    // C: layer2.conv2 (64 -> 64) + downsample(Y)                          Z @ (U2, U3), Y @ U1 -> layer3's input @ (U0, U1)
    if (rg == 0) chain_stage4<W4I, 64, 32, 64, false, true>(p, lds, U2, U1, U0, p.wC, ct, p.bC, sC, group, clamped);
    else chain_stage4<W4E, 64, 32, 64, false, true>(p, lds, U2, U1, U0, p.wC, ct, p.bC, sC, group, clamped);
    __syncthreads();
    FULL_STAMP(4);
    // layer3: IN = (U0, U1), MID = (U2, U3)
    constexpr int L_IN = U0, L_MID = U2;
    const int ct1 = ct, rg1 = rg;
    f32x16 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    auto l3_half = [&](const int h) __attribute__((always_inline)) {
      const char* w2 = h == 0 ? l3.w2a + (size_t)ct2 * BPT2A * 1024 : l3.w2b + (size_t)ct2 * BPT2B * 1024;
      // conv1, output channels 64 h .. 64 h + 63 -> MID (channel tiles 2 h + ct1 of the weight block)
      const char* w1h = l3.w1 + (size_t)(2 * h) * BPT1 * 1024;
      if (rg1 == 0) {
        f32x16 a1[W4I::NT];
#pragma unroll
        for (int s = 0; s < W4I::NT; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) a1[s][r] = 0.f;
        walk4<W4I, 4, 0, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_IN, 0, w1h + (size_t)ct1 * BPT1 * 1024, a1, false);
        if (h == 0) FULL_STAMP(11); else FULL_STAMP(13);
        const int tl[W4I::NT] = {W4I::t[0], W4I::t[1], W4I::t[2], W4I::t[3]};
        epi_to_lds<64, W4I::NT>(lds, L_MID, tl, a1, ct1, l3.b1 + 64 * h, s1, rows_ok, clamped);
      } else {
        f32x16 a1[W4E::NT];
#pragma unroll
        for (int s = 0; s < W4E::NT; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) a1[s][r] = 0.f;
        walk4<W4E, 4, 0, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_IN, 0, w1h + (size_t)ct1 * BPT1 * 1024, a1, false);
        if (h == 0) FULL_STAMP(11); else FULL_STAMP(13);
        const int tl[W4E::NT] = {W4E::t[0], W4E::t[1], W4E::t[2], W4E::t[3], W4E::t[4]};
        epi_to_lds<64, W4E::NT>(lds, L_MID, tl, a1, ct1, l3.b1 + 64 * h, s1, rows_ok, clamped);
      }
      if (h == 0) FULL_STAMP(12); else FULL_STAMP(14);
      __syncthreads();
      FULL_STAMP(5 + 2 * h);
      walk4<W4All, 4, 4, 8 * BLK, 8 * BLK, MAGAT_W4_D>(lds, L_MID, L_IN, w2, acc, h == 1);
      // (after the first half: MID is rewritten by the next conv1; after the second: MID and IN are dead in every wave)
      L3_LDS_SYNC();
      FULL_STAMP(6 + 2 * h);
    };
#if defined(MAGAT_L3_UNROLL_H) && defined(MAGAT_DEBUG_HOOKS)
    // timing experiment (results unchanged): a separate code body per half - the second half then runs code that is NOT in the
    // instruction cache yet, like every other stage of the group loop (DESIGN.md 4.1 "cold code")
    l3_half(0);
    l3_half(1);
#else
#pragma unroll 1
    for (int h = 0; h < 2; ++h) l3_half(h);
#endif
    // the next group's inputs -> the MID region, X1 @ U3, X2 @ U2: in flight under the pooling below
    if (more) {
      dma_map(p.in1, group + gstride, U3, wave, 4, 40);
      dma_map(p.in2, group + gstride, U2, wave, 4, 40);
    }
    FULL_STAMP(9);
    // ---- relu(acc * s2 + bias), 2 x 2 sums in registers, stores.  Accumulator s = tile W4All::t[s]:
    //      0..3 interior tiles Ia Ib Ic Id, 4 C, 5 T (top), 6 B (bottom), 7 L (left), 8 R (right)
    {
      const int fr = lane & 31, fh = lane >> 5, agent = fr & 7;
      const bool lo = (lane >> 3) & 1, hi = (lane >> 4) & 1;       // slot psl = 2 hi + lo
      const int m = group * AG + agent;
      // cells of this lane: its corner cell, the row-edge-middle cell it shares with the lane 8 away, the column-edge-middle
      // cell it shares with the lane 16 away, the centre cell (shared by all four slots)
      const int cellF = 2 * (int)lo + 6 * (int)hi, cell0 = hi ? 1 : 7, cell1 = lo ? 3 : 5;
      // row-major tiles [cell][128 agents][128 channels], or granule-major ones [cell][32 granules][128 agents][4 channels]
      // (magat_hip.h in_gl = 1: what the encoder head's loader reads as 512 contiguous bytes per half wave; here the eight
      // agents of a group make one 128-byte run per store instead of eight 16-byte pieces 512 bytes apart)
      float* ob = l3.out + (long long)(m >> 7) * 9 * (128 * 128) +
                  (l3.out_gl ? (8 * ct2 + fh) * 512 + (m & 127) * 4 : (m & 127) * 128 + 32 * ct2 + 4 * fh);
      const int qstep = l3.out_gl ? 1024 : 8;       // channel quads 2 qd (+ fh) of this wave's 32 channels
      const bool mok = m < p.M;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 vF, v0, v1, v2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * qd + c;
          float v[9];
#pragma unroll
          for (int s = 0; s < 9; ++s) v[s] = magat_relu(__builtin_fmaf(acc[s][r], s2, bq[qd][c]));      // (keeps NaN, like torch.relu)
          const float u = hi ? v[6] : v[5], ux = hi ? v[5] : v[6];
          const float w_ = lo ? v[8] : v[7], wx = lo ? v[7] : v[8];
          vF[c] = (v[4] + v[0]) + (u + w_);
          const float p0 = ux + v[1], p1 = wx + v[2], p2 = v[3];
          v0[c] = p0 + dpp_mov<0x128>(p0);                       // + the lane 8 away (row_ror:8)
          v1[c] = p1 + __shfl_xor(p1, 16, 64);                   // + the lane 16 away
          const float p2b = p2 + dpp_mov<0x128>(p2);
          v2[c] = p2b + __shfl_xor(p2b, 16, 64);
        }
        if (mok) {
          CHAIN_OUT_STORE(ob + (long long)cellF * (128 * 128) + qstep * qd, vF);
          if (!lo) CHAIN_OUT_STORE(ob + (long long)cell0 * (128 * 128) + qstep * qd, v0);
          if (!hi) CHAIN_OUT_STORE(ob + (long long)cell1 * (128 * 128) + qstep * qd, v1);
          if (!lo && !hi) CHAIN_OUT_STORE(ob + (long long)4 * (128 * 128) + qstep * qd, v2);
        }
      }
    }
    FULL_STAMP(10);
  }
  FULL_CLOCKS(1);
  if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
}


}  // namespace

// 1 when magat_block_full can write its pooled map granule-major (out_gl = 1) with the current options
int magat_block_full_out_gl() { return 1; }

// bytes of one stage's fragment-major weight block (without the trailing scale float)
static size_t chain_block_bytes(int cin, int c2, int cout) { return (size_t)(cout / 32) * (9 * (cin / 16) + c2 / 16) * 2 * 1024; }

size_t magat_block_chain_weight_floats() {      // three blocks, each followed by [2^-e, pad x3]
  return (chain_block_bytes(32, 32, 32) + chain_block_bytes(32, 0, 64) + chain_block_bytes(64, 32, 64)) / 4 + 12;
}

// in1 / in2: [ceil(M/128)][36] plane-granule tiles of 32 channels.  w: the three fragment-major weight blocks of
// encoder.pack_chain_weights, each followed by 4 floats [2^-e, 0, 0, 0].  bA / bB / bC: biases of layer1.conv2+downsample,
// layer2.conv1, layer2.conv2+downsample.
// out_gl 2: [ceil(M/128)][36] plane-granule tiles of 64 channels (out_pix_stride = 128*64, out_tile = 36*128*64 floats);
// out_gl 0: float32 row-major tiles with the same strides.
int magat_block_chain(const void* in1, const void* in2, void* out, int out_gl, long long out_pix_stride, long long out_tile,
                      const float* w, const float* bA, const float* bB, const float* bC, int M, int* range_flag,
                      hipStream_t st) {
  if (!in1 || !in2 || !out || !w || !bA || !bB || !bC) return MAGAT_ERR_NULL;
  if (M <= 0 || (out_gl != 0 && out_gl != 2)) return MAGAT_ERR_BAD_SHAPE;
  ChainParams p;
  p.in1 = static_cast<const char*>(in1); p.in2 = static_cast<const char*>(in2); p.out = static_cast<char*>(out);
  p.out_gl = out_gl; p.out_pix_stride = out_pix_stride; p.out_tile = out_tile;
  const char* wb = reinterpret_cast<const char*>(w);
  const size_t nA = chain_block_bytes(32, 32, 32), nB = chain_block_bytes(32, 0, 64), nC = chain_block_bytes(64, 32, 64);
  p.wA = wb; p.wB = wb + nA + 16; p.wC = wb + nA + 16 + nB + 16;
  p.bA = bA; p.bB = bB; p.bC = bC;
  p.sA = reinterpret_cast<const float*>(p.wA + nA);
  p.sB = reinterpret_cast<const float*>(p.wB + nB);
  p.sC = reinterpret_cast<const float*>(p.wC + nC);
  p.M = M; p.groups = (M + AG - 1) / AG;
  p.range_flag = range_flag;
  p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_chain_dbg;
#endif
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&block_chain_w4_kernel), MAGAT_LDS_BLOCK_B4, LDS_TOTAL) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int grid = p.groups < cus ? p.groups : cus;
  const int pid = magat_prof_begin(MAGAT_TAG_BLOCK_CHAIN, st);
  hipLaunchKernelGGL(block_chain_w4_kernel, dim3((unsigned)grid), dim3(256), LDS_TOTAL, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

size_t magat_block3_weight_floats() { return ((size_t)(4 * 72 + 4 * 72 + 4 * 80) * 1024) / 4 + 12; }

// layer3 (+ ReLU + 2x2 sum-pool) of ResNetLarge.  in: [ceil(M/128)][36] plane-granule tiles of 64 channels (layer2's output);
// out: pooled float32 [ceil(M/128)][9][128][128].  w: conv1 block, conv2 first-half block, conv2 second-half block
// (encoder.pack_block3_weights), each followed by 4 floats [2^-e, 0, 0, 0]; b1 / b2: biases of conv1, conv2 + downsample.
int magat_block3(const void* in, float* out, const float* w, const float* b1, const float* b2, int M, int* range_flag,
                 hipStream_t st) {
  if (!in || !out || !w || !b1 || !b2) return MAGAT_ERR_NULL;
  if (M <= 0) return MAGAT_ERR_BAD_SHAPE;
  L3Params p;
  p.in = static_cast<const char*>(in); p.out = out;
  const char* wb = reinterpret_cast<const char*>(w);
  const size_t n1 = (size_t)4 * 72 * 1024, n2a = (size_t)4 * 72 * 1024, n2b = (size_t)4 * 80 * 1024;
  p.w1 = wb; p.w2a = wb + n1 + 16; p.w2b = wb + n1 + 16 + n2a + 16;
  p.s1 = reinterpret_cast<const float*>(p.w1 + n1);
  p.s2 = reinterpret_cast<const float*>(p.w2b + n2b);
  p.b1 = b1; p.b2 = b2;
  p.M = M; p.groups = (M + AG - 1) / AG;
  p.range_flag = range_flag;
  p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_block3_dbg;
#endif
  constexpr size_t lds = 2 * MAP64;
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&block3_w4_kernel), MAGAT_LDS_BLOCK_C, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  const int pid = magat_prof_begin(MAGAT_TAG_BLOCK3, st);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int grid = p.groups < cus ? p.groups : cus;
  hipLaunchKernelGGL(block3_w4_kernel, dim3((unsigned)grid), dim3(256), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

#ifdef MAGAT_DEBUG_HOOKS
extern "C" int magat_chain_set_debug_buffer(long long* dev_buf) { g_chain_dbg = dev_buf; return MAGAT_OK; }
extern "C" int magat_block3_set_debug_buffer(long long* dev_buf) { g_block3_dbg = dev_buf; return MAGAT_OK; }
#endif


// layer1.conv2 -> layer2 -> layer3 -> pool as ONE launch (block_full_p_kernel).  Arguments: those of magat_block_chain (without
// its output) and of magat_block3 (without its input).
int magat_block_full(const void* in1, const void* in2, const float* wchain, const float* bA, const float* bB, const float* bC,
                     float* out, const float* w3, const float* b1, const float* b2, int M, int* range_flag, hipStream_t st,
                     const float* scales, int out_gl) {
  if (!in1 || !in2 || !wchain || !bA || !bB || !bC || !out || !w3 || !b1 || !b2) return MAGAT_ERR_NULL;
  if (M <= 0) return MAGAT_ERR_BAD_SHAPE;
  FullParams q;
  ChainParams& p = q.c;
  p.in1 = static_cast<const char*>(in1); p.in2 = static_cast<const char*>(in2); p.out = nullptr;
  p.out_gl = 2; p.out_pix_stride = 0; p.out_tile = 0;
  const char* wb = reinterpret_cast<const char*>(wchain);
  const size_t nA = chain_block_bytes(32, 32, 32), nB = chain_block_bytes(32, 0, 64), nC = chain_block_bytes(64, 32, 64);
  p.wA = wb; p.wB = wb + nA + 16; p.wC = wb + nA + 16 + nB + 16;
  p.bA = bA; p.bB = bB; p.bC = bC;
  p.sA = reinterpret_cast<const float*>(p.wA + nA);
  p.sB = reinterpret_cast<const float*>(p.wB + nB);
  p.sC = reinterpret_cast<const float*>(p.wC + nC);
  p.M = M; p.groups = (M + AG - 1) / AG;
  p.range_flag = range_flag;
  p.dbg = nullptr;
  L3Params& l = q.l;
  l.in = nullptr; l.out = out;
  const char* w3b = reinterpret_cast<const char*>(w3);
  const size_t n1 = (size_t)4 * 72 * 1024, n2a = (size_t)4 * 72 * 1024, n2b = (size_t)4 * 80 * 1024;
  l.w1 = w3b; l.w2a = w3b + n1 + 16; l.w2b = w3b + n1 + 16 + n2a + 16;
  l.s1 = reinterpret_cast<const float*>(l.w1 + n1);
  l.s2 = reinterpret_cast<const float*>(l.w2b + n2b);
  l.b1 = b1; l.b2 = b2;
  if (scales) { p.sA = scales; p.sB = scales + 1; p.sC = scales + 2; l.s1 = scales + 3; l.s2 = scales + 4; }
  l.M = M; l.groups = p.groups;
  l.range_flag = range_flag;
  l.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  l.dbg = g_block3_dbg;
#endif
  if (out_gl != 0 && out_gl != 1) return MAGAT_ERR_UNSUPPORTED;
  l.out_gl = out_gl;
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&block_full_p_kernel), MAGAT_LDS_BLOCK_FULL_P, LDS_TOTAL) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int grid = p.groups < cus ? p.groups : cus;
  if (p.groups > cus) magat_form_note(MAGAT_FORM_CHAIN_PERSIST);
  const int pid = magat_prof_begin(MAGAT_TAG_BLOCK_FULL, st);
  hipLaunchKernelGGL(block_full_p_kernel, dim3((unsigned)grid), dim3(256), LDS_TOTAL, st, q);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
