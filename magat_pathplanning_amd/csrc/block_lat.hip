// Latency form of the per-agent encoder - stem -> layer1 -> layer2 -> layer3 -> AvgPool -> fc (+ Flatten + Linear) -> compressMLP
// (reference graphs/models/resnet_pytorch.py:40-73 BasicBlock, :427-524 ResNet; decentralplanner_GAT_bottleneck.py:90-166, 291-302)
// - for the reference's own inference loop, which steps ONE planning instance at a time (configs/dcpGAT_OE_Random.json:58
// "test_batch_size": 1; agents/decentralplannerlocal_OnlineExpert_GAT.py:1030-1055).  The batched kernels give a workgroup EIGHT
// agents (stem8.hip, block_fused.hip: rows of the implicit GEMMs are (pixel, agent) pairs, nine 32-row tiles by tap-validity
// class, 13 428 matrix instructions per group) - that fills the chip from 2 048 agents on and leaves 10 .. 100 agents walking
// 136 k cycles on 2 .. 13 of 256 CUs behind seven launches.  Here a workgroup owns ONE agent and the encoder is ONE launch
// (block_lat_kernel<HEAD, STEM>; option LAT_AGENTS, up to 512 agents):
//   STEM   the stem and layer1.conv1 with stem8_kernel's arithmetic: state maps as f16 planes [c0 c1 c2 1.0] per pixel, one k step
//          = one tap row, the stem map (16 x) in LDS with a zero border, layer1.conv1 over it with the stride-2 geometry, the
//          residual branch's input = its stride-2 pixels x 2^-4;
//   chain  rows = the 36 pixels of the agent's 6 x 6 maps = TWO row tiles (32 + 4 pixels; the second tile is 7/8 padding, and
//          still the walk is 18 tile-taps per k step against 69), every map in LDS with a ZERO BORDER (slot = 7 (y + 1) + (x + 1):
//          one zero column serves as right border of a row and left border of the next), so every tap of every pixel is one
//          ds_read_b128 at a compile-time offset - no validity classes, no zero-pixel selects; 2 460 matrix instructions per
//          agent; maps (f16 plane pairs, [plane][8-channel chunk][64 slots][16 B]): 88 KB, nothing aliased - layer3's
//          128-channel intermediate map fits whole;
//   HEAD   the pooled values never leave the CU: scaled and split as the long-K head's float32 loader splits them, the head's
//          K = 9 x 128 walked in that kernel's order on fragment-major weights (ABI 8), compressMLP on the stored feature rows;
//   guard  a workgroup whose planes clamped recomputes its agent in plain float32 (lat_fallback_f32), the last workgroup books.
// Same weights, same products in the same order per output element (a tap the eight-agent form skips for a whole tile
// contributes exact zeros here), same split-and-store epilogues, and the 2 x 2 pooling adds pair up the pixels of a cell exactly
// as block_full_p_kernel's in-register pooling does (corner cells: the two diagonals; edge-middle cells of the top / bottom row:
// the two columns; the other cells: the two rows): feat / comp are BIT-IDENTICAL to the batched kernels' - a planning instance
// alone and as rows of a 6 400-agent batch give the same logits bit for bit (tests/test_gpu_latency.py).
#include "magat_common.h"

namespace {
#include "block_walk.h"

namespace lat {
constexpr int SLOTS = 64;
constexpr int BLK1 = SLOTS * 16;                  // bytes of one (plane, chunk) block of one agent
constexpr int M32 = 8 * BLK1, M64 = 16 * BLK1;   // 32- / 64-channel map: 2 planes x 4 / 8 chunks
constexpr int L_X1 = 0, L_X2 = M32, L_Y = 2 * M32, L_Z = 3 * M32, L_IN3 = L_Z + M64, L_MA = L_IN3 + M64, L_MB = L_MA + M64,
              L_TOTAL = L_MB + M64;              // 88 KB
constexpr int RP = 7;                             // row pitch in slots
constexpr int MINSH = -(RP + 1);                  // smallest tap shift: lane bases point there, immediates stay non-negative
constexpr int D = 8;                              // weight ring: fragments of k step s + 7 requested under k step s
constexpr int AV = 3;                             // operand ring, as in walk4
constexpr int L_HP = L_X1, HP_PLANE = 4 * 9 * 2 * 32;      // the head's activation planes (HEAD): 2 x 2304 B over the dead X1
constexpr int L_CP = L_X2, CP_PLANE = 8 * 32;              // compressMLP's: 2 x 256 B over the dead X2
// STEM: the stem and layer1.conv1 in the same launch (stem8_kernel's arithmetic for one agent): staged input planes [plane][13 rows]
// [14 columns][c0 c1 c2 1.0] and the 11 x 11 x 32 stem map with a zero border (slot 13 (y + 1) + x + 1) behind the chain's maps
constexpr int S_ROWB = 14 * 8, S_INPL = 13 * S_ROWB, S_RP = 13, S_SLOTS = 176, S_BLK = S_SLOTS * 16, S_DEAD = 170;
constexpr int L_SIN = L_MB + M64, L_SMAP = L_SIN + 3072, L_TOTAL_STEM = L_SMAP + 8 * S_BLK;      // 115 712 B
constexpr int DEADSLOT = 60;                      // where the 28 padding lanes of the second row tile store (slots 0 .. 56 are the map)
}  // namespace lat

struct LatParams {
  const char* in1; const char* in2;               // stem8's outputs: layer1.conv1 map, stem stride-2 pixels (f16 plane granules, 32 ch)
  const char* wA; const char* wB; const char* wC; const char* w1; const char* w2a; const char* w2b;
  const float* bA; const float* bB; const float* bC; const float* b1; const float* b2;
  const float* sA; const float* sB; const float* sC; const float* s1; const float* s2;
  float* out; int out_gl; int M; int* range_flag;
  // HEAD: the encoder head and compressMLP in the epilogue (no pooled map is written then)
  const char* hfrag; const char* cfrag; const float* hbias; const float* cbias; const float* insc; const float* insc2;
  float* feat; float* comp; int ldfeat, ldcomp, ncomp;
  // GUARD (HEAD only): the encoder's range guard inside this launch - raw state maps, float32 BN-folded weights (encoder pack
  // offsets 0..17), the status block of the workspace (book[0] working flag, [1] re-run count, [2] this forward's flag, [6]
  // arrival counter); null = the guard's predicated launches follow as for every other form
  const float* x; const float* pk; long long off[18]; int* book;
  // STEM: stem weights [32][27] + bias (x the activation scale when folded), layer1.conv1 fragment-major + its bias
  const float* w0; const float* b0; const char* w1f; const float* b1c;
  long long* dbg;      // MAGAT_DEBUG_HOOKS builds: [grid][4 waves][16] cycle stamps (tools/lat_phase_probe.py)
};
#ifdef MAGAT_DEBUG_HOOKS
#define LAT_STAMP(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
long long* g_lat_dbg = nullptr;
#else
#define LAT_STAMP(i) do { } while (0)
#endif

// The weight ring of a wave: fragments of the first D - 1 k steps of a walk.  A walk finds them REQUESTED ALREADY - by ring_fill,
// issued behind the previous walk's last MFMA, in front of that stage's epilogue and barrier: the L2 round trip of a stage's first
// fragments (1.5-2 k cycles with one wave per SIMD; eight stages) then hides under the epilogue instead of opening every walk
// (tools/lat_phase_probe.py: 2-3 k cycles per stage beyond its MFMA issue time before, ~1 k after).  Every walk has >= 8 k steps.
typedef u32x4 WRing[lat::D][2];
__device__ __forceinline__ void ring_fill(WRing& w, const char* wbase, unsigned lane16) {
#pragma unroll
  for (int j = 0; j < lat::D - 1; ++j) {
    w[j][0] = *reinterpret_cast<const u32x4*>(wbase + (size_t)j * 2048 + lane16);
    w[j][1] = *reinterpret_cast<const u32x4*>(wbase + (size_t)j * 2048 + (lane16 + 1024u));
  }
}

// value pair -> its two f16 planes, remembering whether a value left +-65504: the instruction sequence of split_pair_f16
// (conv_gemm_bf16x6.hip) - the float32 loaders of the long-K head and of compressMLP, whose planes this kernel reproduces
__device__ __forceinline__ void split_pair_lat(float x, float y, unsigned& p1, unsigned& p2, bool& clamped) {
  clamped |= !(__builtin_fabsf(x) <= 65504.f) | !(__builtin_fabsf(y) <= 65504.f);
  x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}

// signed value pair -> its two f16 planes: split2s of stem8.hip (the stem's state maps and weights)
__device__ __forceinline__ void split2s_lat(float x, float y, unsigned& p1, unsigned& p2) {
  x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}

// K walk of a dense layer on ONE row of activations (the agent's): NSTEP k steps of 16, the activation operand of step s at
// lds + abase + 32 s (second plane PS bytes behind) - every column of the 32-wide tile carries the same row -, this wave's
// 32 output channels' weight fragments at wbase (1 KB per plane and step)
template <int NSTEP, int PS>
__device__ __forceinline__ void walk_lin(char* lds, unsigned abase, const char* wbase, unsigned lane16, f32x16& acc, WRing& w) {
  using namespace lat;
  static_assert(NSTEP >= D - 1, "ring_fill requests D - 1 k steps");
  auto load_w = [&](int step, u32x4 (&b)[2]) {
    b[0] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + lane16);
    b[1] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + (lane16 + 1024u));
  };
  u32x4 av[AV][2];
  auto rd = [&](int i, int pl, u32x4& dst) { dst = *reinterpret_cast<const u32x4*>(lds + abase + (32 * i + pl * PS)); };
#pragma unroll
  for (int j = 0; j < AV - 1; ++j)
    if (j < NSTEP) {
      rd(j, 0, av[j][0]);
      rd(j, 1, av[j][1]);
    }
#pragma clang loop unroll(full)
  for (int i = 0; i < NSTEP; ++i) {
    if (i + D - 1 < NSTEP) load_w(i + D - 1, w[(i + D - 1) % D]);
    constexpr int LA = AV - 1;
    const bool more = i + LA < NSTEP;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[i % D][0]), __builtin_bit_cast(f16x8, av[i % AV][0]),
                                                 acc, 0, 0, 0);
    W4_PIN();
    if (more) { rd(i + LA, 0, av[(i + LA) % AV][0]); W4_PIN(); }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[i % D][1]), __builtin_bit_cast(f16x8, av[i % AV][0]),
                                                 acc, 0, 0, 0);
    W4_PIN();
    if (more) { rd(i + LA, 1, av[(i + LA) % AV][1]); W4_PIN(); }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[i % D][0]), __builtin_bit_cast(f16x8, av[i % AV][1]),
                                                 acc, 0, 0, 0);
    asm volatile("" : "+a"(acc));
    W4_PIN();
  }
}

// The K walk of one wave over NT row tiles: 9 taps x KSM k steps over the map at in_off, then KS2 k steps of the residual 1 x 1
// segment over the map at in2_off (same pixel).  ab[s]: the lane's byte offset of (its slot + MINSH) in chunk fh of a map.
template <int NT, int KSM, int KS2, int PS_IN, int PS_IN2, int RPX = lat::RP, int BLKX = lat::BLK1>
__device__ __forceinline__ void walk1(char* lds, const unsigned (&ab)[NT], int in_off, int in2_off, const char* wbase,
                                      unsigned lane16, f32x16 (&acc)[NT], WRing& w) {
  using namespace lat;
  constexpr int MINSHX = -(RPX + 1);      // (RPX / BLKX: row pitch and block size of the map the taps read: the chain's, or the stem map's)
  constexpr int NMAIN = 9 * KSM, NSTEP = NMAIN + KS2, NI = NSTEP * NT;
  static_assert(NSTEP >= D - 1, "ring_fill requests D - 1 k steps");
  auto load_w = [&](int step, u32x4 (&b)[2]) {
    b[0] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + lane16);
    b[1] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + (lane16 + 1024u));
  };
  u32x4 av[AV][2];
  auto rd = [&](int i, int pl, u32x4& dst) {
    const int step = i / NT, s = i % NT;
    if (step >= NMAIN) {
      const int ks = step - NMAIN;
      dst = *reinterpret_cast<const u32x4*>(lds + (ab[s] + (unsigned)in2_off) + (-MINSHX * 16 + pl * PS_IN2 + ks * 2 * BLK1));
    } else {
      const int tp = step / KSM, ks = step % KSM;
      const int sh = RPX * (tp / 3 - 1) + (tp % 3 - 1) - MINSHX;
      dst = *reinterpret_cast<const u32x4*>(lds + (ab[s] + (unsigned)in_off) + (sh * 16 + pl * PS_IN + ks * 2 * BLKX));
    }
  };
#pragma unroll
  for (int j = 0; j < AV - 1; ++j)
    if (j < NI) {
      rd(j, 0, av[j][0]);
      rd(j, 1, av[j][1]);
    }
#pragma clang loop unroll(full)
  for (int i = 0; i < NI; ++i) {
    const int step = i / NT, s = i % NT;
    if (s == 0 && step + D - 1 < NSTEP) load_w(step + D - 1, w[(step + D - 1) % D]);
    constexpr int LA = AV - 1;
    const bool more = i + LA < NI;
    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][0]),
                                                    __builtin_bit_cast(f16x8, av[i % AV][0]), acc[s], 0, 0, 0);
    W4_PIN();
    if (more) { rd(i + LA, 0, av[(i + LA) % AV][0]); W4_PIN(); }
    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][1]),
                                                    __builtin_bit_cast(f16x8, av[i % AV][0]), acc[s], 0, 0, 0);
    W4_PIN();
    if (more) { rd(i + LA, 1, av[(i + LA) % AV][1]); W4_PIN(); }
    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][0]),
                                                    __builtin_bit_cast(f16x8, av[i % AV][1]), acc[s], 0, 0, 0);
    if (NT == 1) asm volatile("" : "+a"(acc[s]));      // (a lone accumulator chain is otherwise moved behind the whole walk's loads)
    W4_PIN();
  }
}

// one convolution of a wave: walk, then relu(acc * scale + bias) as f16 plane chunks of the output map (COUT channels; this wave's
// 32 channels are chunk pair ct_out of it) - the arithmetic of chain_stage4 / epi_to_lds (block_fused.hip)
template <int NT, int CIN, int C2, int COUT, int RPX = lat::RP, int BLKX = lat::BLK1>
__device__ __forceinline__ void lat_stage(char* lds, const unsigned (&ab)[NT], const unsigned (&sl16)[NT], const bool (&live)[NT],
                                          int in_off, int in2_off, int out_off, const char* wts, int ct_out, const float* bias32,
                                          float scale, unsigned lane16, bool& clamped, WRing& w, const char* wnext) {
  using namespace lat;
  constexpr int KSM = CIN / 16, KS2 = C2 / 16;
  constexpr int PS_OUT = (COUT / 8) * BLK1;
  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  walk1<NT, KSM, KS2, (CIN / 8) * BLKX, (C2 > 0 ? C2 / 8 : 1) * BLK1, RPX, BLKX>(lds, ab, in_off, in2_off, wts, lane16, acc, w);
  ring_fill(w, wnext, lane16);      // the wave's NEXT walk's first fragments: in flight under the epilogue and the barrier below
  const int fh = (int)(lane16 >> 9);
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bias32 + 8 * q + 4 * fh);
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    float cl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned h1[4], h2[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * ks + e;
        const f32x2 v01 = __builtin_elementwise_fma(f32x2{acc[s][4 * q], acc[s][4 * q + 1]}, f32x2{scale, scale}, f32x2{bq[q][0], bq[q][1]});
        const f32x2 v23 = __builtin_elementwise_fma(f32x2{acc[s][4 * q + 2], acc[s][4 * q + 3]}, f32x2{scale, scale}, f32x2{bq[q][2], bq[q][3]});
        split2(v01[0], v01[1], h1[2 * e], h2[2 * e], cl);
        split2(v23[0], v23[1], h1[2 * e + 1], h2[2 * e + 1], cl);
      }
      // (no branch around the stores: the padding lanes of the second tile write to a slot no tap reads - with the stores under
      //  a condition the compiler sinks the whole walk into the conditional block)
      char* o = lds + out_off + ((ct_out * 2 + ks) * 2 + fh) * BLK1 + sl16[s];
      *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
      *reinterpret_cast<u32x4*>(o + PS_OUT) = u32x4{h2[0], h2[1], h2[2], h2[3]};
    }
    clamped |= cl > 65504.f && live[s];
  }
}

// ---- the range guard's float32 re-run of ONE agent inside the latency kernel: when a plane of this agent clamped (or the stem's
// did), the workgroup recomputes the agent from its raw state maps in plain float32 FMAs on the BN-folded float32 weights - the
// layers of enc_run_resnet's float32 pass (encoder_f32.hip), maps in LDS as [channel][pixel].  Slow (a few hundred us) and never
// taken by a sane checkpoint; it replaces two predicated launches per forward, which cost ~15 us of a 95 us step.
// out[co][px] = act(bias[co] + sum_tap sum_c w[co][tap Cin + c] in[c][nb(px, tap)] + sum_c2 w[co][9 Cin + c2] res[c2][rpx])
__device__ __forceinline__ void lat_conv_f32(const float* in, int Cin, int Win, int stride, const float* __restrict__ w, int Krow,
                                             const float* __restrict__ bias, const float* res, int Cres, int Wres, int rstride,
                                             float* out, int Cout) {
  const int inpix = Win * Win, respix = Wres * Wres;
  for (int idx = threadIdx.x; idx < Cout * 36; idx += 256) {
    const int co = idx / 36, px = idx - 36 * co, oy = px / 6, ox = px - 6 * oy;
    const float* wr = w + (long long)co * Krow;
    float acc = bias[co];
    for (int ty = 0; ty < 3; ++ty) {
      const int iy = oy * stride + ty - 1;
      if (iy < 0 || iy >= Win) continue;
      for (int tx = 0; tx < 3; ++tx) {
        const int ix = ox * stride + tx - 1;
        if (ix < 0 || ix >= Win) continue;
        const float* ip = in + iy * Win + ix;
        const float* wp = wr + (ty * 3 + tx) * Cin;
        for (int c = 0; c < Cin; ++c) acc = __builtin_fmaf(wp[c], ip[c * inpix], acc);
      }
    }
    if (res) {
      const float* rp = res + (oy * rstride) * Wres + ox * rstride;
      const float* wp = wr + 9 * Cin;
      for (int c = 0; c < Cres; ++c) acc = __builtin_fmaf(wp[c], rp[c * respix], acc);
    }
    out[idx] = magat_relu(acc);
  }
}

__device__ void lat_fallback_f32(const LatParams& p, int m, float* L) {
  const int t = threadIdx.x;
  float* const XIN = L, *const A0 = L + 384, *const B1 = L + 4256, *const Y1 = L + 5408, *const B2 = L + 6560, *const Y2 = L + 8864,
         *const B3 = L + 11168, *const Y3 = L + 15776, *const PL = L + 20384, *const FT = L + 21536;
  const float* pk = p.pk;
  for (int i = t; i < 363; i += 256) XIN[i] = p.x[(long long)m * 363 + i];
  __syncthreads();
  // stem: conv3x3(3 -> 32, pad 1) + BN + ReLU on 11 x 11; weights [32][c 9 + ty 3 + tx]
  for (int idx = t; idx < 32 * 121; idx += 256) {
    const int co = idx / 121, px = idx - 121 * co, oy = px / 11, ox = px - 11 * oy;
    const float* wr = pk + p.off[0] + co * 27;
    float acc = pk[p.off[1] + co];
    for (int c = 0; c < 3; ++c)
      for (int ty = 0; ty < 3; ++ty) {
        const int iy = oy + ty - 1;
        if (iy < 0 || iy >= 11) continue;
        for (int tx = 0; tx < 3; ++tx) {
          const int ix = ox + tx - 1;
          if (ix < 0 || ix >= 11) continue;
          acc = __builtin_fmaf(wr[c * 9 + ty * 3 + tx], XIN[c * 121 + iy * 11 + ix], acc);
        }
      }
    A0[idx] = magat_relu(acc);
  }
  __syncthreads();
  // layer1: conv1 stride 2, conv2 + downsample (1 x 1, stride 2, over the stem map)
  lat_conv_f32(A0, 32, 11, 2, pk + p.off[2], 288, pk + p.off[3], nullptr, 0, 1, 1, B1, 32);
  __syncthreads();
  lat_conv_f32(B1, 32, 6, 1, pk + p.off[4], 288 + 32, pk + p.off[5], A0, 32, 11, 2, Y1, 32);
  __syncthreads();
  lat_conv_f32(Y1, 32, 6, 1, pk + p.off[6], 288, pk + p.off[7], nullptr, 0, 1, 1, B2, 64);
  __syncthreads();
  lat_conv_f32(B2, 64, 6, 1, pk + p.off[8], 576 + 32, pk + p.off[9], Y1, 32, 6, 1, Y2, 64);
  __syncthreads();
  lat_conv_f32(Y2, 64, 6, 1, pk + p.off[10], 576, pk + p.off[11], nullptr, 0, 1, 1, B3, 128);
  __syncthreads();
  lat_conv_f32(B3, 128, 6, 1, pk + p.off[12], 1152 + 64, pk + p.off[13], Y2, 64, 6, 1, Y3, 128);
  __syncthreads();
  // AvgPool2d(2) as a sum (the 1/4 lives in the head's weights) -> [cell][channel]
  for (int idx = t; idx < 9 * 128; idx += 256) {
    const int cell = idx >> 7, c = idx & 127, cy = cell / 3, cx = cell - 3 * cy;
    const float* y = Y3 + c * 36 + (2 * cy) * 6 + 2 * cx;
    PL[idx] = (y[0] + y[1]) + (y[6] + y[7]);
  }
  __syncthreads();
  if (t < 128) {      // head: fc (+ Flatten + Linear) folded, [n_feat][cell 128 + c]
    const float* wr = pk + p.off[14] + (long long)t * 1152;
    float acc = pk[p.off[15] + t];
    for (int k = 0; k < 1152; ++k) acc = __builtin_fmaf(wr[k], PL[k], acc);
    FT[t] = acc;
    p.feat[(long long)m * p.ldfeat + t] = acc;
  }
  __syncthreads();
  if (t < p.ncomp) {      // compressMLP
    const float* wr = pk + p.off[16] + (long long)t * 128;
    float acc = pk[p.off[17] + t];
    for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(wr[k], FT[k], acc);
    p.comp[(long long)m * p.ldcomp + t] = magat_relu(acc);
  }
}

template <bool HEAD, bool STEM>
__global__ __launch_bounds__(256, 1) void block_lat_kernel(const LatParams p) {
  using namespace lat;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m = blockIdx.x;
  int pre = 0;
  if (HEAD && p.book) pre = __hip_atomic_load(p.book, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned lane16 = (unsigned)lane * 16u;
  // this wave's walks in order and their weight blocks: (STEM: layer1.conv1 -) A (waves 0 / 1 only) - B - C - layer3.conv1 - conv2
  // over MA - conv2 over MB + residual (- HEAD: head - compressMLP).  The first one's ring is requested here, every later one's behind
  // the walk in front of it (ring_fill).
  const char* const wB_ = p.wB + (size_t)(wave & 1) * (18 * 2) * 1024;
  const char* const wC_ = p.wC + (size_t)(wave & 1) * (38 * 2) * 1024;
  const char* const w31_ = p.w1 + (size_t)wave * 72 * 1024;
  const char* const w32a_ = p.w2a + (size_t)wave * 72 * 1024;
  const char* const w32b_ = p.w2b + (size_t)wave * 80 * 1024;
  WRing wr;
  ring_fill(wr, wave < 2 ? (STEM ? p.w1f : p.wA) : wB_, lane16);
  const int fr = lane & 31, fh = lane >> 5;
  // the lane's pixels: tile s, column fr -> position g = 32 s + fr = 4 cell + e in pooled-cell order; the order of a cell's four
  // pixels over its quad of lanes is the pairing of block_full_p_kernel's in-register pooling (file header)
  unsigned abT[2], slT[2];
  bool liveT[2];
  int cellT[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int g = 32 * s + fr;
    const bool lv = g < NPIX;
    const int cell = lv ? g >> 2 : 0, e = g & 3;
    const int cy = cell / 3, cx = cell - 3 * cy;
    const bool corner = cy != 1 && cx != 1, rowmid = cx == 1 && cy != 1;
    const int dy = corner ? (e == 1 || e == 3) : rowmid ? (e & 1) : (e >> 1);
    const int dx = corner ? (e == 1 || e == 2) : rowmid ? (e >> 1) : (e & 1);
    const int slot = lv ? RP * (2 * cy + dy + 1) + (2 * cx + dx + 1) : RP + 1;
    slT[s] = (unsigned)(lv ? slot : DEADSLOT) * 16u;
    abT[s] = (unsigned)(slot + MINSH) * 16u + (unsigned)fh * BLK1;
    liveT[s] = lv;
    cellT[s] = cell;
  }
  const float sA = *p.sA, sB = *p.sB, sC = *p.sC, s1 = *p.s1, s2 = *p.s2;
  // HEAD: everything the late phases read from memory besides their weight rings - activation scales, plane scales, biases - is
  // requested HERE, branch-free, and pinned: next to their uses (where the compiler puts them; the optional scales each in a branch
  // of their own) they were four exposed round trips behind the last walks of a 36 us kernel
  float insc_h = 1.f, insc2_h = 1.f, hs_raw = 0.f, cs_raw = 0.f;
  f32x4 hb_h[4], cb_h[4], b2_h[4];
  if (HEAD) {
    const float i1 = *(p.insc ? p.insc : p.sA), i2 = *(p.insc2 ? p.insc2 : p.sA);
    insc_h = p.insc ? i1 : 1.f;
    insc2_h = p.insc2 ? i2 : 1.f;
    hs_raw = *reinterpret_cast<const float*>(p.hfrag + (size_t)4 * 72 * 2048);
    cs_raw = *reinterpret_cast<const float*>(p.cfrag + (size_t)(p.ncomp / 32) * 8 * 2048);
    const int cw = 32 * wave < p.ncomp ? wave : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hb_h[q] = *reinterpret_cast<const f32x4*>(p.hbias + 32 * wave + 8 * q + 4 * fh);
      cb_h[q] = *reinterpret_cast<const f32x4*>(p.cbias + 32 * cw + 8 * q + 4 * fh);
    }
    if (insc_h == 0.f) insc_h = 1.f;
    if (insc2_h == 0.f) insc2_h = 1.f;
  }
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) b2_h[qd] = *reinterpret_cast<const f32x4*>(p.b2 + 32 * wave + 8 * qd + 4 * fh);
  __builtin_amdgcn_sched_barrier(0);
  bool clamped = false;
  LAT_STAMP(0);
  if (!STEM) {
    // the agent's two input maps (stem8_kernel's outputs): 2 x 36 pixels x 8 (plane, chunk) pieces of 16 B, requested in front of
    // the LDS clear
    u32x4 piece[3];
    unsigned pdst[3];
    const long long tile_b = (long long)(m >> 7) * NPIX * (128 * 32 * 4) + (m & 127) * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = t + 256 * i;
      const int map = id >= 8 * NPIX, r = id - map * 8 * NPIX;
      const int pix = r >> 3, blk = r & 7;
      const int y = pix / 6, x = pix - 6 * y;
      pdst[i] = (unsigned)((map ? L_X2 : L_X1) + blk * BLK1 + (RP * (y + 1) + x + 1) * 16);
      if (id < 16 * NPIX)
        piece[i] = *reinterpret_cast<const u32x4*>((map ? p.in2 : p.in1) + tile_b + (long long)pix * (128 * 32 * 4) +
                                                    (blk >> 2) * (256 * 32) + (blk & 3) * 2048);
    }
    for (int i = t; i < L_TOTAL / 16; i += 256) *reinterpret_cast<u32x4*>(lds + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    L3_LDS_SYNC();
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (t + 256 * i < 16 * NPIX) *reinterpret_cast<u32x4*>(lds + pdst[i]) = piece[i];
    __syncthreads();
  } else {
    // ---- stem (conv 3 -> 32 + BN + ReLU on 11 x 11) + layer1.conv1 (stride 2) in this launch: stem8_kernel's arithmetic for one
    // agent - the state maps as f16 planes [c0 c1 c2 1.0] per pixel with a zero border, one k step = one tap ROW (16 K slots = the
    // four channel slots of four neighbouring pixels, the bias rides against the constant 1.0), the stem map carried 16 x, the
    // residual branch's input = its stride-2 pixels x 2^-4 - so X1 / X2 hold what the two launches hand over, bit for bit
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (t < 121) {
      const float* r = p.x + (long long)m * 363 + t;
      v0 = r[0]; v1 = r[121]; v2 = r[242];
    }
    u32x4 wa[3][2];      // stem weights as this lane's row fragments: row = channel fr, k step = tap row ty (stem8.hip)
    // (24 weights and the bias requested together from clamped addresses, the predicate applied to the value: each in a branch of
    //  its own they were a chain of ten round trips in front of everything else)
    float wraw[3][8];
    const float b0v = p.b0[fr];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int slot = 8 * fh + i, tx = slot >> 2, c = slot & 3;
        wraw[ty][i] = p.w0[fr * 27 + ((tx < 3 && c < 3) ? c * 9 + ty * 3 + tx : 0)];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      float wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int slot = 8 * fh + i, tx = slot >> 2, c = slot & 3;
        float v = (tx < 3 && c < 3) ? wraw[ty][i] : 0.f;
        if (ty == 1 && tx == 1 && c == 3) v = b0v;
        wv[i] = v * 16.f;
      }
      unsigned h1[4], h2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2s_lat(wv[2 * e], wv[2 * e + 1], h1[e], h2[e]);
      wa[ty][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
      wa[ty][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
    }
    const float scale1 = *reinterpret_cast<const float*>(p.w1f + 9 * 2 * 2 * 1024) * (1.f / 16.f);
    for (int i = t; i < L_TOTAL_STEM / 16; i += 256) *reinterpret_cast<u32x4*>(lds + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    L3_LDS_SYNC();
    LAT_STAMP(1);
    if (t < 121) {
      const int y = t / 11, x = t - 11 * y;
      // (range guard of the INPUT: a NaN / Inf / |x| > 65504 entry must not become a finite clamp)
      clamped |= !(__builtin_fabsf(v0) <= 65504.f) || !(__builtin_fabsf(v1) <= 65504.f) || !(__builtin_fabsf(v2) <= 65504.f);
      unsigned h01, l01, h2x, l2x;
      split2s_lat(v0, v1, h01, l01);
      split2s_lat(v2, 1.f, h2x, l2x);
      char* dst = lds + L_SIN + (y + 1) * S_ROWB + (x + 1) * 8;
      *reinterpret_cast<uint2*>(dst) = uint2{h01, h2x};
      *reinterpret_cast<uint2*>(dst + S_INPL) = uint2{l01, l2x};
    }
    L3_LDS_SYNC();
    LAT_STAMP(2);
    {
      // the stem: wave w = pixels 32 w .. 32 w + 31 (pixel 120 repeated behind the map's end: stored to a dead slot)
      const int pixr = 32 * wave + fr, pix = pixr < 121 ? pixr : 120;
      const int y = pix / 11, x = pix - 11 * y;
      const char* ip = lds + L_SIN + y * S_ROWB + (x + 2 * fh) * 8;
      u32x4 xh[3], xl[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(ip + ty * S_ROWB), a1 = *reinterpret_cast<const uint2*>(ip + ty * S_ROWB + 8);
        const uint2 c0 = *reinterpret_cast<const uint2*>(ip + ty * S_ROWB + S_INPL), c1 = *reinterpret_cast<const uint2*>(ip + ty * S_ROWB + S_INPL + 8);
        xh[ty] = u32x4{a0.x, a0.y, a1.x, a1.y};
        xl[ty] = u32x4{c0.x, c0.y, c1.x, c1.y};
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // (w hi, x hi), (w lo, x hi), (w hi, x lo) per tap row; the eight-agent kernel skips the third product for a group whose
      // second planes are all zero - a sum of exact zeros either way
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ty][0]), __builtin_bit_cast(f16x8, xh[ty]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ty][1]), __builtin_bit_cast(f16x8, xh[ty]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ty][0]), __builtin_bit_cast(f16x8, xl[ty]), acc, 0, 0, 0);
      }
      float cl = 0.f;
      const bool lv = pixr < 121;
      char* o = lds + L_SMAP + fh * S_BLK + (lv ? S_RP * (y + 1) + x + 1 : S_DEAD) * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned h1[4], h2[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int q = 2 * ks + e;
          split2(acc[4 * q], acc[4 * q + 1], h1[2 * e], h2[2 * e], cl);
          split2(acc[4 * q + 2], acc[4 * q + 3], h1[2 * e + 1], h2[2 * e + 1], cl);
        }
        *reinterpret_cast<u32x4*>(o + ks * 2 * S_BLK) = u32x4{h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<u32x4*>(o + ks * 2 * S_BLK + 4 * S_BLK) = u32x4{h2[0], h2[1], h2[2], h2[3]};
      }
      clamped |= cl > 65504.f && lv;      // 16 x the stem output beyond the planes' range
    }
    L3_LDS_SYNC();
    LAT_STAMP(3);
    if (wave < 2) {
      // layer1.conv1 over the stem map (stride 2): waves 0 / 1 = row tile 0 / 1 -> X1
      const int g = 32 * wave + fr;
      const bool lv = g < NPIX;
      // (the lane's output pixel (oy, ox) is its pooled-cell-order pixel: slT holds its slot in the chain's maps)
      const int slot7 = (int)((wave ? slT[1] : slT[0]) >> 4);
      const int oy = lv ? slot7 / RP - 1 : 0, ox = lv ? slot7 - RP * (oy + 1) - 1 : 0;
      const unsigned abS[1] = {(unsigned)(S_RP * (2 * oy + 1) + (2 * ox + 1) - (S_RP + 1)) * 16u + (unsigned)fh * S_BLK};
      const unsigned slS[1] = {wave ? slT[1] : slT[0]};
      const bool lvS[1] = {lv};
      lat_stage<1, 32, 0, 32, S_RP, S_BLK>(lds, abS, slS, lvS, L_SMAP, 0, L_X1, p.w1f, 0, p.b1c, scale1, lane16, clamped, wr, p.wA);
    } else {
      // the residual branch's input: the stem map at the 36 stride-2 pixels, planes x 2^-4 -> X2
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      const h2v sc = {(_Float16)0.0625f, (_Float16)0.0625f};
      for (int item = t - 128; item < NPIX * 8; item += 128) {
        const int blk = item & 7, opix = item >> 3;
        const int oy = opix / 6, ox = opix - 6 * oy;
        u32x4 v = *reinterpret_cast<const u32x4*>(lds + L_SMAP + blk * S_BLK + (S_RP * (2 * oy + 1) + 2 * ox + 1) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned hv = v[e];
          v[e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2v, hv) * sc);
        }
        *reinterpret_cast<u32x4*>(lds + L_X2 + blk * BLK1 + (RP * (oy + 1) + ox + 1) * 16) = v;
      }
    }
    L3_LDS_SYNC();
  }
  LAT_STAMP(4);
  const int ct = wave & 1, tl = wave >> 1;
  const unsigned ab1[1] = {tl ? abT[1] : abT[0]}, sl1[1] = {tl ? slT[1] : slT[0]};
  const bool lv1[1] = {tl ? liveT[1] : liveT[0]};
  // A: layer1.conv2 (32 -> 32) + downsample(stem stride-2 pixels)   X1, X2 -> Y      (one channel tile: waves 0 / 1 = row tile 0 / 1)
  if (wave < 2) {
    const unsigned abA[1] = {wave ? abT[1] : abT[0]}, slA[1] = {wave ? slT[1] : slT[0]};
    const bool lvA[1] = {wave ? liveT[1] : liveT[0]};
    lat_stage<1, 32, 32, 32>(lds, abA, slA, lvA, L_X1, L_X2, L_Y, p.wA, 0, p.bA, sA, lane16, clamped, wr, wB_);
  }
  L3_LDS_SYNC();      // (LDS hand-over only: the next walk's ring is in flight)
  LAT_STAMP(5);
  // B: layer2.conv1 (32 -> 64)   Y -> Z      (wave = channel tile x row tile)
  lat_stage<1, 32, 0, 64>(lds, ab1, sl1, lv1, L_Y, 0, L_Z, wB_, ct, p.bB + 32 * ct, sB, lane16, clamped, wr, wC_);
  L3_LDS_SYNC();      // (LDS hand-over only: the next walk's ring is in flight)
  LAT_STAMP(6);
  // C: layer2.conv2 (64 -> 64) + downsample(Y)   Z, Y -> layer3's input
  lat_stage<1, 64, 32, 64>(lds, ab1, sl1, lv1, L_Z, L_Y, L_IN3, wC_, ct, p.bC + 32 * ct, sC, lane16, clamped, wr, w31_);
  L3_LDS_SYNC();      // (LDS hand-over only: the next walk's ring is in flight)
  LAT_STAMP(7);
  // layer3.conv1 (64 -> 128): wave = channel tile, both row tiles; channels 0..63 -> MA, 64..127 -> MB (the two K halves of conv2)
  lat_stage<2, 64, 0, 64>(lds, abT, slT, liveT, L_IN3, 0, wave < 2 ? L_MA : L_MB, w31_, wave & 1, p.b1 + 32 * wave, s1, lane16, clamped,
                          wr, w32a_);
  L3_LDS_SYNC();      // (LDS hand-over only: the next walk's ring is in flight)
  LAT_STAMP(8);
  // layer3.conv2 (128 -> 128) + downsample(layer3's input): K over MA, then over MB + the residual segment - the eight-agent form's
  // order - then relu(acc * s2 + bias), the 2 x 2 sums across each quad of lanes, one store per cell and channel quad
  {
    f32x16 acc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    walk1<2, 4, 0, 8 * BLK1, 8 * BLK1>(lds, abT, L_MA, 0, w32a_, lane16, acc, wr);
    ring_fill(wr, w32b_, lane16);
    walk1<2, 4, 4, 8 * BLK1, 8 * BLK1>(lds, abT, L_MB, L_IN3, w32b_, lane16, acc, wr);
    if (HEAD) ring_fill(wr, p.hfrag + (size_t)wave * 72 * 2048, lane16);
    LAT_STAMP(9);
    f32x4 bq[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) bq[qd] = b2_h[qd];
    float* ob = p.out + (long long)(m >> 7) * 9 * (128 * 128) +
                (p.out_gl ? (8 * wave + fh) * 512 + (m & 127) * 4 : (m & 127) * 128 + 32 * wave + 4 * fh);
    const int qstep = p.out_gl ? 1024 : 8;
    const float insc = insc_h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = magat_relu(__builtin_fmaf(acc[s][4 * qd + c], s2, bq[qd][c]));
          const float pr = a + dpp_mov<0xB1>(a);            // + the lane's partner in its pair (quad_perm [1,0,3,2])
          v[c] = pr + dpp_mov<0x4E>(pr);                    // + the other pair (quad_perm [2,3,0,1])
        }
        const bool owner = liveT[s] && (fr & 3) == 0;
        if (!HEAD) {
          if (owner) *reinterpret_cast<f32x4*>(ob + (long long)cellT[s] * (128 * 128) + qstep * qd) = v;
        } else {
          // the head's activation planes: pooled value x its activation scale, split as the long-K head's float32 loader splits
          // it; operand order [32-channel slab = this wave][cell][k step][lane half][8 halves] in the dead X1 region
          bool cl = false;
          unsigned h1[2], h2[2];
          split_pair_lat(v[0] * insc, v[1] * insc, h1[0], h2[0], cl);
          split_pair_lat(v[2] * insc, v[3] * insc, h1[1], h2[1], cl);
          const unsigned o = (unsigned)(L_HP + ((wave * 9 + cellT[s]) * 2 + (qd >> 1)) * 32 + (qd & 1) * 16 + fh * 8);
          if (owner) {
            *reinterpret_cast<uint2*>(lds + o) = uint2{h1[0], h1[1]};
            *reinterpret_cast<uint2*>(lds + o + HP_PLANE) = uint2{h2[0], h2[1]};
          }
          clamped |= cl && owner;
        }
      }
    }
  }
  if (HEAD) {
    L3_LDS_SYNC();
    LAT_STAMP(10);
    // ---- encoder head: feat = W_head . pooled (K = 9 x 128 in the long-K kernel's order: 32-channel slab outer, cell inner) ----
    const float insc = insc_h, insc2 = insc2_h;
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
    walk_lin<72, HP_PLANE>(lds, (unsigned)(L_HP + fh * 16), p.hfrag + (size_t)wave * 72 * 2048, lane16, hacc, wr);
    if (32 * wave < p.ncomp) ring_fill(wr, p.cfrag + (size_t)wave * 8 * 2048, lane16);
    LAT_STAMP(11);
    {
      const float hs = hs_raw / insc;      // (exact: powers of two)
      bool cl = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = hb_h[q];
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = hacc[4 * q + c] * hs + bb[c];
        if (fr == 0) *reinterpret_cast<f32x4*>(p.feat + (long long)m * p.ldfeat + 32 * wave + 8 * q + 4 * fh) = v;
        // compressMLP's activation planes from the very values stored (x its activation scale), operand order
        // [k step = 2 wave + q / 2][lane half = q % 2][4 fh + c] in the dead X2 region
        unsigned h1[2], h2[2];
        split_pair_lat(v[0] * insc2, v[1] * insc2, h1[0], h2[0], cl);
        split_pair_lat(v[2] * insc2, v[3] * insc2, h1[1], h2[1], cl);
        const unsigned o = (unsigned)(L_CP + (2 * wave + (q >> 1)) * 32 + (q & 1) * 16 + fh * 8);
        if (fr == 0) {
          *reinterpret_cast<uint2*>(lds + o) = uint2{h1[0], h1[1]};
          *reinterpret_cast<uint2*>(lds + o + CP_PLANE) = uint2{h2[0], h2[1]};
        }
      }
      clamped |= cl;
    }
    L3_LDS_SYNC();
    LAT_STAMP(12);
    // ---- compressMLP: comp = relu(W_c . feat + b_c); ncomp = 32 | 64 | 128 outputs: the first ncomp / 32 waves ----
    if (32 * wave < p.ncomp) {
      f32x16 cacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
      walk_lin<8, CP_PLANE>(lds, (unsigned)(L_CP + fh * 16), p.cfrag + (size_t)wave * 8 * 2048, lane16, cacc, wr);
      const float cs = cs_raw / insc2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = cb_h[q];
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = magat_relu(cacc[4 * q + c] * cs + bb[c]);
        if (fr == 0) *reinterpret_cast<f32x4*>(p.comp + (long long)m * p.ldcomp + 32 * wave + 8 * q + 4 * fh) = v;
      }
    }
    LAT_STAMP(13);
    if (p.book) {
      // ---- the encoder's range guard inside this launch (see lat_fallback_f32).  `pre`: the stem's clamp flag (final: the stem
      // ran before this launch; a workgroup of THIS launch that already raised it only makes a later one recompute for nothing)
      const int own = __syncthreads_or(clamped ? 1 : 0);
      if (own | pre) lat_fallback_f32(p, m, reinterpret_cast<float*>(lds));
      if (t == 0) {
        if (own) {
          __hip_atomic_fetch_or(p.book, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        // bookkeeping by the workgroup that arrives last (magat_guard_book's protocol: this is the last reader of the flag)
        const unsigned prev = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.book) + 6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
          const int f = __hip_atomic_load(p.book, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          p.book[2] = f;
          if (f != 0) p.book[1] += 1;
          __hip_atomic_store(p.book, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.book + 6, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      LAT_STAMP(14);
      return;
    }
  }
  if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
}

}  // namespace

#ifdef MAGAT_DEBUG_HOOKS
extern "C" int magat_block_lat_set_debug_buffer(long long* dev_buf) { g_lat_dbg = dev_buf; return MAGAT_OK; }
#endif

static size_t lat_chain_block_bytes(int cin, int c2, int cout) { return (size_t)(cout / 32) * (9 * (cin / 16) + c2 / 16) * 2 * 1024; }

// Arguments: those of magat_block_full (block_fused.hip); one workgroup per agent.
int magat_block_lat(const void* in1, const void* in2, const float* wchain, const float* bA, const float* bB, const float* bC,
                    float* out, const float* w3, const float* b1, const float* b2, int M, int* range_flag, hipStream_t st,
                    const float* scales, int out_gl, const magat_lat_head* head, const magat_lat_guard* guard,
                    const magat_lat_stem* stem) {
  if ((!stem && (!in1 || !in2)) || !wchain || !bA || !bB || !bC || (!head && !out) || !w3 || !b1 || !b2) return MAGAT_ERR_NULL;
  if (M <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (out_gl != 0 && out_gl != 1) return MAGAT_ERR_UNSUPPORTED;
  LatParams p;
  p.in1 = static_cast<const char*>(in1); p.in2 = static_cast<const char*>(in2);
  const char* wb = reinterpret_cast<const char*>(wchain);
  const size_t nA = lat_chain_block_bytes(32, 32, 32), nB = lat_chain_block_bytes(32, 0, 64), nC = lat_chain_block_bytes(64, 32, 64);
  p.wA = wb; p.wB = wb + nA + 16; p.wC = wb + nA + 16 + nB + 16;
  p.bA = bA; p.bB = bB; p.bC = bC;
  p.sA = reinterpret_cast<const float*>(p.wA + nA);
  p.sB = reinterpret_cast<const float*>(p.wB + nB);
  p.sC = reinterpret_cast<const float*>(p.wC + nC);
  const char* w3b = reinterpret_cast<const char*>(w3);
  const size_t n1 = (size_t)4 * 72 * 1024, n2a = (size_t)4 * 72 * 1024, n2b = (size_t)4 * 80 * 1024;
  p.w1 = w3b; p.w2a = w3b + n1 + 16; p.w2b = w3b + n1 + 16 + n2a + 16;
  p.s1 = reinterpret_cast<const float*>(p.w1 + n1);
  p.s2 = reinterpret_cast<const float*>(p.w2b + n2b);
  p.b1 = b1; p.b2 = b2;
  if (scales) { p.sA = scales; p.sB = scales + 1; p.sC = scales + 2; p.s1 = scales + 3; p.s2 = scales + 4; }
  p.out = out; p.out_gl = out_gl; p.M = M; p.range_flag = range_flag;
  p.hfrag = p.cfrag = nullptr; p.hbias = p.cbias = p.insc = p.insc2 = nullptr; p.feat = p.comp = nullptr; p.ldfeat = p.ldcomp = 0; p.ncomp = 0;
  if (head) {
    if (!head->hfrag || !head->cfrag || !head->hbias || !head->cbias || !head->feat || !head->comp) return MAGAT_ERR_NULL;
    if (head->ncomp != 32 && head->ncomp != 64 && head->ncomp != 128) return MAGAT_ERR_UNSUPPORTED;
    if ((head->ldfeat & 3) || (head->ldcomp & 3) || head->ldfeat < 128 || head->ldcomp < head->ncomp ||
        ((reinterpret_cast<uintptr_t>(head->feat) | reinterpret_cast<uintptr_t>(head->comp) |
          reinterpret_cast<uintptr_t>(head->hbias) | reinterpret_cast<uintptr_t>(head->cbias)) & 15))
      return MAGAT_ERR_BAD_SHAPE;
    p.hfrag = reinterpret_cast<const char*>(head->hfrag); p.cfrag = reinterpret_cast<const char*>(head->cfrag);
    p.hbias = head->hbias; p.cbias = head->cbias; p.insc = head->insc; p.insc2 = head->insc2;
    p.feat = head->feat; p.comp = head->comp; p.ldfeat = head->ldfeat; p.ldcomp = head->ldcomp; p.ncomp = head->ncomp;
  }
  p.x = nullptr; p.pk = nullptr; p.book = nullptr; p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_lat_dbg;
#endif
  for (int i = 0; i < 18; ++i) p.off[i] = 0;
  if (guard) {
    if (!head) return MAGAT_ERR_UNSUPPORTED;
    if (!guard->x || !guard->pack || !guard->off || !guard->book) return MAGAT_ERR_NULL;
    p.x = guard->x; p.pk = guard->pack; p.book = guard->book;
    for (int i = 0; i < 18; ++i) p.off[i] = guard->off[i];
    magat_form_note(MAGAT_FORM_GUARD_LAT);
  }
  p.w0 = p.b0 = p.b1c = nullptr; p.w1f = nullptr;
  if (stem) {
    if (!head) return MAGAT_ERR_UNSUPPORTED;
    if (!stem->x || !stem->w0 || !stem->b0 || !stem->w1f || !stem->b1) return MAGAT_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(stem->b1) | reinterpret_cast<uintptr_t>(stem->w1f)) & 15) return MAGAT_ERR_UNSUPPORTED;
    p.x = stem->x; p.w0 = stem->w0; p.b0 = stem->b0; p.w1f = reinterpret_cast<const char*>(stem->w1f); p.b1c = stem->b1;
    magat_form_note(MAGAT_FORM_STEM_LAT);
  }
  const void* fn = !head ? reinterpret_cast<const void*>(&block_lat_kernel<false, false>)
                   : stem ? reinterpret_cast<const void*>(&block_lat_kernel<true, true>)
                          : reinterpret_cast<const void*>(&block_lat_kernel<true, false>);
  const size_t ldsb = stem ? lat::L_TOTAL_STEM : lat::L_TOTAL;
  if (magat_ensure_dyn_lds(fn, !head ? MAGAT_LDS_BLOCK_LAT : stem ? MAGAT_LDS_BLOCK_LAT_S : MAGAT_LDS_BLOCK_LAT_H, ldsb) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  magat_form_note(MAGAT_FORM_CHAIN_LAT);
  if (head) magat_form_note(MAGAT_FORM_HEAD_LAT);
  const int pid = magat_prof_begin(MAGAT_TAG_BLOCK_FULL, st);
  if (!head) hipLaunchKernelGGL((block_lat_kernel<false, false>), dim3((unsigned)M), dim3(256), ldsb, st, p);
  else if (stem) hipLaunchKernelGGL((block_lat_kernel<true, true>), dim3((unsigned)M), dim3(256), ldsb, st, p);
  else hipLaunchKernelGGL((block_lat_kernel<true, false>), dim3((unsigned)M), dim3(256), ldsb, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
