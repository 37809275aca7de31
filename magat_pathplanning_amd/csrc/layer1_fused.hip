// conv_first (3 -> 32, 3x3, pad 1, +BN+ReLU) FUSED into layer1.conv1 (32 -> 32, 3x3, stride 2, pad 1, +BN+ReLU) of the
// ResNet encoder (reference graphs/models/resnet_pytorch.py:40-73 BasicBlock, :427-470 stem; decentralplanner_GAT_
// bottleneck.py:90-117), f16x3 arithmetic (two half-precision planes per value, three v_mfma_f32_32x32x16_f16 per
// product, fp32 accumulate - conv_gemm_bf16x6.hip).
//
// Unfused, the 32-channel stem output (121 pixels x 32 floats = 15.5 KB per agent, 793 MB per 51200-agent step) is
// written by one kernel and read back by the next: 0.56 ms of a 4.1 ms step spent on an intermediate that is 13x larger
// than the stem's input and 3.4x larger than layer1.conv1's output.  Here a workgroup owns (128 agents) x (one OUTPUT
// pixel of layer1.conv1): it stages the agents' 5x5x3 input windows in LDS (39 KB), and for every valid tap of the
// stride-2 3x3 window it
//   1. gathers the tap pixel's 27-value im2col row out of LDS and splits it into f16 planes,
//   2. runs the stem as a 32x32x32 f16x3 MFMA product -> D[channel][agent], +bias, ReLU,
//   3. splits D's registers into f16 planes - by construction (plane-granule order, magat_hip.h in_gl = 2) they ARE the
//      activation operand of layer1.conv1's MFMA for this tap - and
//   4. accumulates the tap's 32x32x32 product with the K-permuted layer1.conv1 weights (all nine taps resident in LDS,
//      36 KB, loaded once with the LDS-direct load).
// The stem is recomputed 289/121 = 2.4 times (16 M extra MFMA flops per agent-step against 793 MB of HBM traffic each
// way); no barrier after the staging one.  The stem output at the CENTRE tap is the stride-2 pixel the residual 1x1
// branch of the block reads (resnet_pytorch.py:66-70): it is written out as its own [Ho*Wo][32] plane-granule tensor.
#include <cstdlib>

#include "magat_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct L1Params {
  const float* x;            // (M, 3, H, W)
  const float* w0;           // stem weights [32][27] (BN folded), bias [32]
  const float* b0;
  const unsigned short* w1;  // layer1.conv1 f16 planes [2][32][288], K-permuted per 32-wide slab, then 1 float 2^-e
  const float* b1;
  void* out;                 // [tile][Ho*Wo] plane-granule tiles of 32 channels (layer1.conv1 output)
  void* ctr;                 // same geometry: stem output at pixels (2 oy, 2 ox)
  int M, H, W, Ho, Wo, Mt;
  int* range_flag;           // range guard: OR-ed with 1 when the 16x stem output or the layer1.conv1 output had to be clamped
};

__device__ __forceinline__ void split2(float x, float y, unsigned& p1, unsigned& p2) {
  x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  // residual x - hi as one mixed-precision fma per value (fma(hi, -1, x): exact; the f16 operand is read from its half of the
  // packed register) instead of two conversions and a packed subtract: v_pk_add_f32 does not issue while an MFMA runs
  // (tools/exp/mfma_valu.hip), and this kernel's two waves per SIMD split while the other multiplies
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}

// non-negative inputs: ReLU and the f16 range clamp are the same v_med3
__device__ __forceinline__ void split2_relu(float x, float y, unsigned& p1, unsigned& p2) {
  x = __builtin_amdgcn_fmed3f(x, 0.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, 0.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  // residual x - hi as one mixed-precision fma per value (fma(hi, -1, x): exact; the f16 operand is read from its half of the
  // packed register) instead of two conversions and a packed subtract: v_pk_add_f32 does not issue while an MFMA runs
  // (tools/exp/mfma_valu.hip), and this kernel's two waves per SIMD split while the other multiplies
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}

constexpr float W0_SCALE = 16.f;    // stem weights (and bias) are split as planes of 16 w: the stem output is carried 16x
                                    // too large (exact; saturates beyond 4094) and layer1.conv1's scale undoes it
constexpr int MAXV = 4;             // 16-byte loads per image row: W <= 16 (the LDS windows fit up to W = 15)

// Workgroup = (64 agents) x (one OUTPUT ROW of layer1.conv1), 4 waves: wave -> 32 agents x half of the row's pixels; TWO
// workgroups per CU (78 KB of LDS each at W = 11), so that one stages its windows - an HBM round trip, 70 of the kernel's
// 290 us when exposed - while the other computes (the 128-agent / 8-wave form fitted once per CU).
// LDS: eight taps of the layer1.conv1 weights (32 KB; the centre tap's fragments stay in registers: with all nine the two
// workgroups miss the CU's 160 KB by 4.6 KB) + the agents' 5-row input windows:
//   per agent  [0] | channel c: 5 rows of (W values, 0)      (stride WSTR dwords, odd: agents on distinct banks)
// the trailing zero of a row is also column -1 of the next row (and [0] that of the first), so the stem's zero padding
// needs no index tests.  Every value is stored SPLIT, as one dword (plane-0 half | plane-1 half << 16): a window value is
// used by up to nine stem taps of up to three output pixels, so it is split once here, and a tap builds its MFMA operand
// with one v_perm per two values.  Staging cost is paid once per 128 x Wo output pixels.
constexpr int L1_AG = 64;           // agents per workgroup
__global__ __launch_bounds__(256, 2) void layer1_fused_kernel(const L1Params p) {
  extern __shared__ __attribute__((aligned(1024))) char l1smem[];
  char* const Ws = l1smem;                                           // layer1.conv1 weights: [tap][plane][32 rows][64 B]
  unsigned* const win = reinterpret_cast<unsigned*>(l1smem + 8 * 4096);

  const int RW = p.W + 1, RB = 5 * RW, WSTR = (1 + 3 * RB) | 1;
  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / (2 * p.Ho));      // 128-agent tile of the output layout
  if (mtile >= p.Mt) return;
  const int sub = (slot / p.Ho) & 1;                                 // which 64 agents of the tile
  const int oy = slot % p.Ho;
  const int m0 = mtile * 128 + sub * L1_AG;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int npix = p.Ho * p.Wo;

  // ---- staging -----------------------------------------------------------------------------------------------------
  // layer1.conv1 weights, every tap: 36 pieces of 1 KB (16 rows x 64 B of one plane and tap), LDS-direct
  {
    const char* wb = reinterpret_cast<const char*>(p.w1);
    for (int id = wave; id < 32; id += 4) {              // (slot * 2 + plane) * 2 + row half; slot = tap, skipping tap 4
      const int tap = (id >> 2) + ((id >> 2) >= 4 ? 1 : 0), plane = (id >> 1) & 1, row = (id & 1) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const char* src = wb + ((long long)plane * 32 * 288 + row * 288 + tap * 32 + c * 8) * 2;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Ws + (unsigned)id * 1024u);
      asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  }
  // input windows: image rows 2 oy - 2 .. 2 oy + 2 of the three channels.  One work item = one image row (agent,
  // channel, row): ceil(W/4) 16-byte loads (dword-aligned; the last one clamped back into the row) - per-element
  // 4-byte gathers cost several times the address-coalescer cycles.
  {
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    const int HW = p.H * p.W;
    const int nv = (p.W + 3) / 4;
    constexpr int NIT = (L1_AG * 15 + 255) / 256;        // 4 rows per thread
    f32x4 v4[NIT][MAXV];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int item = t + 256 * i;
      const int a = item / 15, r = item - a * 15;
      const int c = r / 5, wy = r - c * 5;
      const int iy = 2 * oy - 2 + wy;
      const bool ok = item < L1_AG * 15 && m0 + a < p.M && iy >= 0 && iy < p.H;
      const float* row = p.x + (long long)(m0 + a) * 3 * HW + c * HW + iy * p.W;
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        v4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok && j < nv) v4[i][j] = *reinterpret_cast<const f32x4_u*>(row + min(4 * j, p.W - 4));
      }
    }
    for (int i = t; i < L1_AG; i += 256) win[i * WSTR] = 0u;
    // range guard of the INPUT: a NaN / Inf / |x| > 65504 state tensor (the reference would hand the NaN on to the logits,
    // resnet_pytorch.py:40-73 is plain float32) must not become a finite clamp: the negated compare is true for NaN too
    bool xbad = false;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
      for (int j = 0; j < MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xbad |= !(__builtin_fabsf(v4[i][j][e]) <= 65504.f);
    if (xbad && p.range_flag) atomicOr(p.range_flag, 1);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int item = t + 256 * i;
      const int a = item / 15, r = item - a * 15;
      if (item < L1_AG * 15) {
        unsigned* dst = win + a * WSTR + 1 + r * RW;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
          const int sh = 4 * j - min(4 * j, p.W - 4);    // the clamped load starts sh columns early (uniform)
          float val[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            val[e] = v4[i][j][e];                        // sh = 0
            if (sh == 1 && e < 3) val[e] = v4[i][j][e + 1];
            if (sh == 2 && e < 2) val[e] = v4[i][j][e + 2];
            if (sh == 3 && e < 1) val[e] = v4[i][j][e + 3];
          }
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            unsigned p1, p2;
            split2(val[e], val[e + 1], p1, p2);          // (h1a | h1b << 16), (h2a | h2b << 16)
            const int col = 4 * j + e;
            if (j < nv && col < p.W) dst[col] = __builtin_amdgcn_perm(p2, p1, 0x05040100u);          // h1a | h2a << 16
            if (j < nv && col + 1 < p.W) dst[col + 1] = __builtin_amdgcn_perm(p2, p1, 0x07060302u);  // h1b | h2b << 16
          }
        }
        dst[p.W] = 0u;
      }
    }
  }
  // stem weights as this lane's MFMA row-operand fragments (row = channel lane&31, k = 16 ks + 8 (lane>>5) + i), split
  // into f16 planes of 256 w; im2col offsets of the same k values inside a window; biases of the channels this lane ends
  // up holding (4 (lane>>5) + 8 g + c)
  u32x4 wa[2][2];
  int koff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * ks + 8 * fh + i;
      wv[i] = k < 27 ? p.w0[fr * 27 + k] * W0_SCALE : (k == 27 ? p.b0[fr] * W0_SCALE : 0.f);   // slot 27: bias x 1.0
      const int kk = k < 27 ? k : 0;
      koff[ks][i] = (kk / 9) * RB + ((kk % 9) / 3) * RW + (kk % 3);
    }
    unsigned h1[4], h2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(wv[2 * e], wv[2 * e + 1], h1[e], h2[e]);
    wa[ks][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
    wa[ks][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
  }
  // centre tap (ty = tx = 1) of layer1.conv1: this lane's weight fragments straight from global memory (the LDS copy's layout:
  // row fr of the 32 x 288 plane, 16-byte chunk 2 ks + fh of the tap's 32 columns)
  u32x4 fbc[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      fbc[ks][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w1) +
                                                    ((long long)pl * 32 * 288 + fr * 288 + 4 * 32 + (2 * ks + fh) * 8) * 2);
  const float scale1 =
      *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.w1) + 2 * 32 * 288 * 2) * (1.f / W0_SCALE);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // ---- pixels of this wave: 32 agents x half of the output row -----------------------------------------------------------
  const int agent = 32 * (wave & 1) + fr;                // agent inside the workgroup's 64
  const int m = m0 + agent;
  const int half = (p.Wo + 1) / 2;
  const int ox_lo = (wave >> 1) * half, ox_hi = min(p.Wo, ox_lo + half);
  const unsigned* wbase = win + agent * WSTR;
  const int iy0 = 2 * oy - 1;                            // tap (ty, tx) reads stem pixel (iy0 + ty, 2 ox - 1 + tx)
  const int ty0 = iy0 < 0 ? 1 : 0, ty1 = min(3, p.H - iy0);
  constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};     // h1g1 h1g2 h2g1 (activation plane, weight plane)

  u32x4 qkeep[3][2][2];      // stem column 2 ox + 1 (rows iy0 .. iy0 + 2) as layer1.conv1 operand planes: next pixel's tap column 0
  for (int ox = ox_lo; ox < ox_hi; ++ox) {
    const int ix0 = 2 * ox - 1;
    const int tx0 = ix0 < 0 ? 1 : 0, tx1 = min(3, p.W - ix0);
    // plane-granule address of this lane's operand inside a 32-channel tile: + plane * 256 * 32 + ks * 4096
    const long long tile_off = ((long long)mtile * npix + oy * p.Wo + ox) * (128 * 32 * 4) + fh * 2048 + (sub * L1_AG + agent) * 16;
    f32x16 acc1, acc1b;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = acc1b[r] = 0.f;
    bool clamped = false;

    // Taps (ty, tx) of the stride-2 window, fully unrolled with wave-uniform validity tests.  Consecutive output pixels of a
    // wave share a stem column (2 ox + 1 is tap column 2 of pixel ox and tap column 0 of pixel ox + 1): its three stem
    // pixels' operand planes are kept in registers (`qkeep`) instead of being recomputed - 6 of a wave's 27 stem products.
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        if (ty < ty0 || ty >= ty1 || tx < tx0 || tx >= tx1) continue;      // (wave-uniform)
        u32x4 qa[2][2];
        const bool reuse = tx == 0 && ox > ox_lo;                            // (wave-uniform; ox > ox_lo implies ix0 >= 1)
        if (reuse) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) { qa[ks][0] = qkeep[ty][ks][0]; qa[ks][1] = qkeep[ty][ks][1]; }
        } else {
          // window index of the 3x3 patch's corner: row ty (= stem row iy0 + ty - 1), column ix0 + tx - 1 (>= -1)
          const unsigned* wp = wbase + 1 + ty * RW + (ix0 + tx - 1);
          // 1. im2col row of the tap pixel: 16 packed (plane 0 | plane 1) dwords -> the two f16 operand planes by v_perm;
          //    slot k = 27 (a zero-weight padding slot of the 27-wide row) carries the constant 1.0 against the bias
          u32x4 pb[2][2];
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            unsigned pv[8];
  #pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = wp[koff[ks][i]];
            if (ks == 1) pv[3] = fh ? 0x00003C00u : pv[3];
            unsigned h1[4], h2[4];
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              h1[e] = __builtin_amdgcn_perm(pv[2 * e + 1], pv[2 * e], 0x05040100u);
              h2[e] = __builtin_amdgcn_perm(pv[2 * e + 1], pv[2 * e], 0x07060302u);
            }
            pb[ks][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
            pb[ks][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
          }
          // 2. stem (+ bias): D[channel][agent] = 16 (w . x + b)
          f32x16 acc0;
  #pragma unroll
          for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks)
  #pragma unroll
            for (int q = 0; q < 3; ++q)
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ks][PB[q]]),
                                                            __builtin_bit_cast(f16x8, pb[ks][PA[q]]), acc0, 0, 0, 0);
          // 3. ReLU (the lower bound of the f16 clamp) -> f16 planes = layer1.conv1's operand of this tap (quads 2 ks,
          //    2 ks + 1 -> k step ks), still 16x
          {
            float mx = acc0[0];
  #pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc0[r]);
            clamped |= mx > 65504.f;                       // stem output beyond 4094: outside what the 16x form carries
          }
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            unsigned h1[4], h2[4];
  #pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int g = 2 * ks + e;
              split2_relu(acc0[4 * g], acc0[4 * g + 1], h1[2 * e], h2[2 * e]);
              split2_relu(acc0[4 * g + 2], acc0[4 * g + 3], h1[2 * e + 1], h2[2 * e + 1]);
            }
            qa[ks][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
            qa[ks][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
          }
          if (ty == 1 && tx == 1 && m < p.M) {             // stem pixel (2 oy, 2 ox): the residual branch's input, unscaled
            char* o = static_cast<char*>(p.ctr) + tile_off;
  #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              unsigned h1[4], h2[4];
  #pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int g = 2 * ks + e;
                split2_relu(acc0[4 * g] * (1.f / W0_SCALE), acc0[4 * g + 1] * (1.f / W0_SCALE), h1[2 * e], h2[2 * e]);
                split2_relu(acc0[4 * g + 2] * (1.f / W0_SCALE), acc0[4 * g + 3] * (1.f / W0_SCALE), h1[2 * e + 1],
                            h2[2 * e + 1]);
              }
              *reinterpret_cast<u32x4*>(o + ks * 4096) = u32x4{h1[0], h1[1], h1[2], h1[3]};
              *reinterpret_cast<u32x4*>(o + 256 * 32 + ks * 4096) = u32x4{h2[0], h2[1], h2[2], h2[3]};
            }
          }
        }
        if (tx == 2) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) { qkeep[ty][ks][0] = qa[ks][0]; qkeep[ty][ks][1] = qa[ks][1]; }
        }
        // 4. layer1.conv1 tap product
        const int tap = ty * 3 + tx;
        const char* wt = Ws + (tap - (tap > 4 ? 1 : 0)) * 4096;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int c = 2 * ks + fh;
          u32x4 fb[2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            fb[pl] = tap == 4 ? fbc[ks][pl]
                              : *reinterpret_cast<const u32x4*>(wt + pl * 2048 + (fr * 4 + (c ^ ((fr >> 2) & 3))) * 16);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (ks == 0)
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[PB[q]]),
                                                            __builtin_bit_cast(f16x8, qa[ks][PA[q]]), acc1, 0, 0, 0);
            else
              acc1b = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[PB[q]]),
                                                             __builtin_bit_cast(f16x8, qa[ks][PA[q]]), acc1b, 0, 0, 0);
          }
        }
      }

    // ---- layer1.conv1 output of this pixel as f16 plane granules ----------------------------------------------------------
    if (m < p.M) {
      char* o = static_cast<char*>(p.out) + tile_off;
      f32x4 bq1[4];      // (re-read per pixel, an L1 hit: its 16 registers hold the centre tap's weight fragments instead)
#pragma unroll
      for (int g = 0; g < 4; ++g) bq1[g] = *reinterpret_cast<const f32x4*>(p.b1 + 8 * g + 4 * fh);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned h1[4], h2[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int g = 2 * ks + e;
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[c] = fmaxf((acc1[4 * g + c] + acc1b[4 * g + c]) * scale1 + bq1[g][c], 0.f);
            clamped |= v[c] > 65504.f;
          }
          split2(v[0], v[1], h1[2 * e], h2[2 * e]);
          split2(v[2], v[3], h1[2 * e + 1], h2[2 * e + 1]);
        }
        *reinterpret_cast<u32x4*>(o + ks * 4096) = u32x4{h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<u32x4*>(o + 256 * 32 + ks * 4096) = u32x4{h2[0], h2[1], h2[2], h2[3]};
      }
    }
    if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
  }
}

}  // namespace

// LDS bytes of the fused kernel for a W-wide input map; 0 = does not fit / not supported (use the two-kernel path)
size_t magat_layer1_fused_lds(int W) {
  if (W < 4 || W > 4 * MAXV) return 0;
  const size_t wstr = (size_t)((1 + 15 * (W + 1)) | 1);
  const size_t lds = 8 * 4096 + (size_t)L1_AG * wstr * sizeof(float);
  return lds <= 160 * 1024 ? lds : 0;
}

// x (M,3,H,W) -> out, ctr: [ceil(M/128)][Ho*Wo] plane-granule tiles of 32 channels (128*32*4 bytes each), Ho = (H-1)/2+1.
// w1 = f16 planes [2][32][288] of layer1.conv1 (K-permuted copy of the encoder pack) followed by the float 2^-e.
int magat_layer1_fused(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* out,
                       void* ctr, int M, int H, int W, hipStream_t st, int* range_flag) {
  if (!x || !w0 || !b0 || !w1 || !b1 || !out || !ctr) return MAGAT_ERR_NULL;
  if (M <= 0 || H < 3 || W < 4) return MAGAT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(b0) & 15) || (reinterpret_cast<uintptr_t>(b1) & 15) ||
      (reinterpret_cast<uintptr_t>(w1) & 15))
    return MAGAT_ERR_BAD_SHAPE;
  L1Params p;
  p.x = x; p.w0 = w0; p.b0 = b0; p.w1 = reinterpret_cast<const unsigned short*>(w1); p.b1 = b1; p.out = out; p.ctr = ctr;
  p.M = M; p.H = H; p.W = W; p.Ho = (H + 2 - 3) / 2 + 1; p.Wo = (W + 2 - 3) / 2 + 1;
  p.Mt = (M + 127) / 128;
  p.range_flag = range_flag;
  const long long groups = (p.Mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD;
  const long long grid = groups * MAGAT_NUM_XCD * p.Ho * 2;      // two 64-agent workgroups per 128-agent tile and output row
  if (grid <= 0 || grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const size_t lds = magat_layer1_fused_lds(W);                     // 79.1 KB at W = 11: two 4-wave workgroups per CU
  if (lds == 0) return MAGAT_ERR_UNSUPPORTED;
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&layer1_fused_kernel), MAGAT_LDS_L1FUSED, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  const int pid = magat_prof_begin(MAGAT_TAG_CONV_FIRST, st);
  hipLaunchKernelGGL(layer1_fused_kernel, dim3((unsigned)grid), dim3(256), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
