// Stem + layer1.conv1 of the ResNet encoders for groups of eight agents (see the kernel's comment).  Its own source file:
// block_fused.hip is compiled with -amdgpu-mfma-vgpr-form (MFMA results in architectural registers), under which vector
// instructions do NOT run beside the matrix pipe - and this kernel lives on exactly that overlap (the split-and-store of stem
// tile i - 1 between the MFMAs of tile i).
#include "magat_common.h"

namespace {
// MAGAT_STEM8_NT: the kernel's streams (state maps in, the two plane maps out - 472 MB nobody reads before the launch ends)
// marked non-temporal (bit 0 loads, bit 1 stores)
#ifndef MAGAT_STEM8_NT
#define MAGAT_STEM8_NT 0      /* measured: neutral to +2 us (profiles/r04e/ab_nt2.txt); the hooks stay for the next look */
#endif
#if MAGAT_STEM8_NT & 1
#define S8_NT_STR " nt"
#else
#define S8_NT_STR ""
#endif
#if MAGAT_STEM8_NT & 2
#define S8_OUT_STORE(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<u32x4*>(ptr))
#else
#define S8_OUT_STORE(ptr, val) (*reinterpret_cast<u32x4*>(ptr) = (val))
#endif
#include "block_walk.h"

// ---------------------------------------------------------------------------------------------------------------------------
// Stem (conv 3 -> 32, 3x3, pad 1, + BN + ReLU; resnet_pytorch.py:427-470, 495-505) + layer1.conv1 (32 -> 32, 3x3, stride 2,
// pad 1, + BN + ReLU; :40-73) for GROUPS OF EIGHT AGENTS, the organisation of the chain kernel above (round 4; the 64-agent
// row-band kernel of layer1_fused.hip recomputes 18 stem rows for 11 and builds the 27-wide im2col operand of the 3-channel
// stem with 20 vector instructions per MFMA).  Here every stem pixel is computed ONCE, and no vector instruction forms an
// operand:
//   * the state maps of the group are staged as f16 planes [plane][agent][13 rows][14 columns][4 halves] with a zero border
//     (channel 3 of every real pixel = 1.0: the bias rides in the product).  One k step of the stem = one tap ROW: its 16 K
//     slots are the four channel slots of the pixels x - 1, x, x + 1, x + 2 of image row y + ty - 1, i.e. 32 CONTIGUOUS bytes
//     of the staged row - lane half fh reads 16 of them, two ds_read_b64 per plane (9 of 16 slots carry weights: 3 k steps
//     where the dense 27-wide form takes 2, in exchange for an operand that needs no v_perm at all);
//   * the stem output (carried 16x like in layer1_fused.hip: the planes of 16 w are normal numbers) goes to an LDS map in the
//     chain kernels' layout [plane][8-channel chunk][pixel 0..120 | zero slot][agent][16 B] (122 KB), 4 pixels x 8 agents per
//     MFMA row tile;
//   * layer1.conv1 is walk4 over that map with the stride-2 geometry (GeoStem): the same nine row tiles by tap-validity
//     class, fragment-major weights streamed global -> registers (encoder.pack_chain_weights), four waves = the row-tile
//     split of the chain kernel's stage A;
//   * the residual branch's input (the stem at the 36 stride-2 pixels) is copied out of the LDS map, its planes multiplied
//     by 2^-4 (v_pk_mul_f16: exact for normal halves).
// Persistent workgroups (one per CU, 156 KB of LDS), the next group's raw state maps travel global -> LDS (LDS-direct) under the
// walks.  Outputs: out / ctr as magat_layer1_fused writes them (plane-granule tiles of 32 channels).  11 x 11 maps only.
constexpr int S8_ROWB = 14 * 8;                       // bytes per staged image row: 14 pixel slots x 4 halves
constexpr int S8_INAG = 13 * S8_ROWB;                 // 1456 bytes per (plane, agent): agents land on distinct banks
constexpr int S8_INPL = AG * S8_INAG;                 // 11 648
constexpr int S8_IN = 0, S8_RAW = 2 * S8_INPL;        // raw float32 state maps of the NEXT group: 8 x 363 floats
constexpr int S8_RAWB = 11776;
constexpr int S8_S = S8_RAW + S8_RAWB;                // stem map
constexpr int S8_BLK = 122 * PIXB;                    // 15 616 bytes per (plane, chunk) block
constexpr int S8_LOF = S8_S + 8 * S8_BLK;             // 160 000: four words, "wave 4 + i staged a non-zero second plane"
constexpr int S8_TOTAL = S8_LOF + 64;
constexpr float S8_W0 = 16.f;

struct Stem8Params {
  const float* x;            // (M, 3, 11, 11)
  const float* w0;           // stem weights [32][27] (BN folded, k = 9 c + 3 ty + tx), bias [32]
  const float* b0;
  const char* w1;            // layer1.conv1, fragment-major f16 planes [tap 9][ks 2][plane 2] x 1 KB, then float 2^-e
  const float* b1;
  char* out;                 // layer1.conv1 output: [agent tile][36] plane-granule tiles of 32 channels
  char* ctr;                 // stem output at the stride-2 pixels, same geometry
  int M, groups;
  int x_aligned;             // x is 16-byte aligned: the raw maps travel by LDS-direct loads (else by plain dword loads)
  int* range_flag;
  long long* dbg;            // MAGAT_DEBUG_HOOKS builds: [grid][8][16] cycle stamps per wave of the LAST group walked
};
#ifdef MAGAT_DEBUG_HOOKS
#define S8_STAMP(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define S8_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void split2s(float x, float y, unsigned& p1, unsigned& p2) {      // signed values (inputs, weights)
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

template <typename TL>
__device__ __forceinline__ void stem8_conv1(const Stem8Params& p, char* lds, int group, float scale1, bool& clamped) {
  constexpr int NT = TL::NT;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
  // (the bias in front of the walk: requested at the epilogue it would start behind an L2 round trip.  The weight ring filled
  //  in front of the stem -> conv1 barrier was measured too: 181 -> 204 us - the barrier then waits out the fill)
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(p.b1 + 8 * q + 4 * fh);
  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  walk4<TL, 2, 0, 4 * S8_BLK, 4 * S8_BLK, w4_depth(NT, 2), GeoStem>(lds, S8_S, 0, p.w1, acc, false);
  S8_STAMP(4);
  const int m = group * AG + agent;
  const bool mok = m < p.M;
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int pix = tile_pix(TL::t[s], psl);
    float cl = 0.f;
    char* o = p.out + ((long long)(m >> 7) * NPIX + pix) * (128 * 32 * 4) + (m & 127) * 16 + fh * 2048;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned h1[4], h2[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * ks + e;
        const f32x2 v01 = __builtin_elementwise_fma(f32x2{acc[s][4 * q], acc[s][4 * q + 1]}, f32x2{scale1, scale1}, f32x2{bq[q][0], bq[q][1]});
        const f32x2 v23 = __builtin_elementwise_fma(f32x2{acc[s][4 * q + 2], acc[s][4 * q + 3]}, f32x2{scale1, scale1}, f32x2{bq[q][2], bq[q][3]});
        split2(v01[0], v01[1], h1[2 * e], h2[2 * e], cl);
        split2(v23[0], v23[1], h1[2 * e + 1], h2[2 * e + 1], cl);
      }
      if (mok) {
        S8_OUT_STORE(o + ks * 4096, (u32x4{h1[0], h1[1], h1[2], h1[3]}));
        S8_OUT_STORE(o + ks * 4096 + 256 * 32, (u32x4{h2[0], h2[1], h2[2], h2[3]}));
      }
    }
    clamped |= cl > 65504.f && mok;
  }
}

__global__ __launch_bounds__(512, 1) void stem8_kernel(const Stem8Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
  // EIGHT waves, two per SIMD: a lone wave issues in order - one instruction per ~5 cycles, and nothing beside its own MFMAs
  // (measured: the four-wave form of this kernel ran the stem's 9 MFMAs + ~105 vector / LDS instructions per tile as their
  // plain sum) - so the overlap of matrix and vector work has to come from a second wave on the SIMD:
  //   phase 1 (all eight waves): the stem - five row tiles for waves 0-3, three for waves 4-7;
  //   phase 2: waves 0-3 = layer1.conv1 (the nine row tiles dealt as in the chain kernel's stage A: fragment weights are
  //            streamed once per tile list), waves 4-7 = everything that is vector / LDS work only - the NEXT group's raw
  //            maps -> f16 planes, the request for the maps of the group after it, and this group's stride-2 copy.
  // Waves 4-7 own two agents each: a wave requests, waits for and converts only its own agents' raw maps, so no barrier
  // separates the LDS-direct loads from their readers.
  // zero border of the staged planes (and everything else of them), zero pixel slot of the stem map: once per workgroup
  for (int i = t; i < S8_RAW / 16; i += 512) *reinterpret_cast<u32x4*>(lds + i * 16) = u32x4{0u, 0u, 0u, 0u};
  for (int i = t; i < 8 * (PIXB / 4); i += 512)
    *reinterpret_cast<unsigned*>(lds + S8_S + (i / (PIXB / 4)) * S8_BLK + 121 * PIXB + (i % (PIXB / 4)) * 4) = 0u;
  // stem weights as this lane's MFMA row fragments: row = channel fr, k step = tap row ty, K slot 8 fh + i = (pixel column
  // tx = slot / 4, channel slot c = slot % 4); slot (ty 1, tx 1, c 3) carries the bias against the constant 1.0
  // (the 24 weights and the bias are requested together - clamped addresses, the predicate on the value - and pinned: each in a
  //  branch of its own they were a chain of ten round trips at the head of every workgroup)
  u32x4 wa[3][2];
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
      wv[i] = v * S8_W0;
    }
    unsigned h1[4], h2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2s(wv[2 * e], wv[2 * e + 1], h1[e], h2[e]);
    wa[ty][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
    wa[ty][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
  }
  const float scale1 = *reinterpret_cast<const float*>(p.w1 + 9 * 2 * 2 * 1024) * (1.f / S8_W0);
  const long long xfloats = (long long)p.M * 363;
  bool clamped = false;
  const int gstride = (int)gridDim.x;
  const int sw = wave - 4;                       // staging wave 0..3: agents 2 sw, 2 sw + 1 of a group
  // raw state maps of this wave's two agents -> its OWN quarter of RAW (2944 bytes): the 16-byte pieces that overlap bytes
  // [2904 sw, 2904 (sw + 1)) of the group's 11 616 (a piece that straddles two waves' ranges is fetched by both, each into its
  // own buffer: a wave may request the maps of group g + 2 while its neighbour still converts those of g + 1), LDS-direct while
  // the whole group lies inside x; plain loads for the last (partial) group - rows of agents past M become zeros
  const int rb0 = (sw * 2904) & ~15, rb1 = ((sw + 1) * 2904 + 15) & ~15;
  const int rawq = S8_RAW + sw * 2944;                               // this wave's buffer: bytes [rb0, rb1) of the group
  const float* rawf = reinterpret_cast<const float*>(lds + rawq + (sw * 2904 - rb0));      // its first agent's first float
  auto fetch_raw = [&](int group) {
    const long long f0 = (long long)group * (AG * 363);
    if (p.x_aligned && f0 + AG * 363 <= xfloats) {
      const char* src = reinterpret_cast<const char*>(p.x + f0);
      for (int c = 0; c < 3; ++c) {
        const int off = rb0 + c * 1024 + lane * 16;
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)(rawq + c * 1024));
        if (off < rb1)
          asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" S8_NT_STR ::"v"(src + off), "s"(m0v) : "memory", "m0");
      }
    } else {
      for (int i = lane; i < 2 * 363; i += 64) {
        const long long idx = f0 + sw * 726 + i;
        const_cast<float*>(rawf)[i] = idx < xfloats ? p.x[idx] : 0.f;
      }
    }
  };
  // raw float32 maps of this wave's two agents -> f16 planes [plane][agent][row y + 1][column x + 1][c0 c1 c2 1.0]
  // Round 5: the state maps are binary in practice (AgentState.toInputTensor 'Project_G': obstacle / goal / agent channels of 0
  // and 1, dataloader/statetransformer_Guidance.py:185-239) - exact in one f16 plane, the second plane all zeros.  The staging
  // waves note whether ANY second-plane word of the group is non-zero; when none is, the stem skips its third product
  // (w hi x lo = a sum of exact zeros: the same values): 6 MFMAs per tile instead of 9, stem kernel 163-168 -> 156-158 us.
  auto to_planes = [&]() {
    bool xbad = false;
    unsigned lo_any = 0u;
    for (int item = lane; item < 2 * 121; item += 64) {
      const int al = item >= 121 ? 1 : 0, pix = item - 121 * al, a = 2 * sw + al;
      const int y = pix / 11, x = pix - 11 * y;
      const float* r = rawf + al * 363 + pix;
      const float v0 = r[0], v1 = r[121], v2 = r[242];
      // range guard of the INPUT: a NaN / Inf / |x| > 65504 entry must not become a finite clamp (the negated compare is
      // true for NaN too)
      xbad |= !(__builtin_fabsf(v0) <= 65504.f) || !(__builtin_fabsf(v1) <= 65504.f) || !(__builtin_fabsf(v2) <= 65504.f);
      unsigned h01, l01, h2x, l2x;
      split2s(v0, v1, h01, l01);
      split2s(v2, 1.f, h2x, l2x);
      char* dst = lds + S8_IN + a * S8_INAG + (y + 1) * S8_ROWB + (x + 1) * 8;
      *reinterpret_cast<uint2*>(dst) = uint2{h01, h2x};
      *reinterpret_cast<uint2*>(dst + S8_INPL) = uint2{l01, l2x};
      lo_any |= l01 | l2x;
    }
    clamped |= xbad;
    const unsigned long long any = __ballot(lo_any != 0u);
    if (lane == 0) *reinterpret_cast<unsigned*>(lds + S8_LOF + 4 * sw) = any != 0ull ? 1u : 0u;
  };
  __syncthreads();                                 // (the zeroing above, before any plane is written)
  if (wave >= 4 && (int)blockIdx.x < p.groups) {
    fetch_raw(blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    to_planes();
    if ((int)blockIdx.x + gstride < p.groups) fetch_raw(blockIdx.x + gstride);
  }
#pragma unroll 1
  for (int group = blockIdx.x; group < p.groups; group += gstride) {
    S8_STAMP(0);
    L3_LDS_SYNC();        // this group's planes are written; every wave is done with the previous group's stem map
    S8_STAMP(1);
    // ---- phase 1, stem: tiles of 4 pixels x 8 agents (tile 31 does not exist: its
    // lanes repeat pixel 120, as the unused slots of tile 30 do - identical values written twice).  Software-pipelined: the
    // nine MFMAs of tile i alternate with the eight split-and-store pieces of tile i - 1's epilogue, the operand reads of
    // tile i + 1 are issued in front of them (what this wave cannot overlap, the other wave of its SIMD does).
    {
      const bool mok = group * AG + agent < p.M;
      float cl = 0.f;
      const u32x4 lof = *reinterpret_cast<const u32x4*>(lds + S8_LOF);      // (written a whole phase ago, behind two barriers)
      const bool has_lo = __builtin_amdgcn_readfirstlane((int)(lof[0] | lof[1] | lof[2] | lof[3])) != 0;
      // waves 0-3 take FIVE tiles (w, w + 4, .., w + 16), waves 4-7 three (16 + w, 20 + w, 24 + w): the SIMD arbiter is
      // oldest-first, so the lower wave of a pair gets the issue slots and finishes early (3.4 k against 4.6 k cycles with four
      // tiles each)
      const int tbase = wave < 4 ? wave : 16 + wave, ntile = wave < 4 ? 5 : 3;
      auto tile_pix = [&](int k) { return min(4 * (tbase + 4 * k) + psl, 120); };
      auto in_ptr = [&](int pix) {
        const int y = pix / 11, x = pix - 11 * y;
        return lds + S8_IN + agent * S8_INAG + y * S8_ROWB + (x + 2 * fh) * 8;
      };
      auto rd_ops = [&](const char* ip, u32x4 (&xh)[3], u32x4 (&xl)[3]) {
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const uint2 a0 = *reinterpret_cast<const uint2*>(ip + ty * S8_ROWB), a1 = *reinterpret_cast<const uint2*>(ip + ty * S8_ROWB + 8);
          xh[ty] = u32x4{a0.x, a0.y, a1.x, a1.y};
          if (has_lo) {                         // (wave-uniform; an all-zero second plane is neither read nor multiplied)
            const uint2 c0 = *reinterpret_cast<const uint2*>(ip + ty * S8_ROWB + S8_INPL),
                        c1 = *reinterpret_cast<const uint2*>(ip + ty * S8_ROWB + S8_INPL + 8);
            xl[ty] = u32x4{c0.x, c0.y, c1.x, c1.y};
          }
        }
      };
      // MFMA j (0..8) of a tile: tap row j / 3, product j % 3 = (w hi, x hi), (w lo, x hi), (w hi, x lo)
      auto mma = [&](int j, f32x16& acc, const u32x4 (&xh)[3], const u32x4 (&xl)[3]) {
        const int ty = j / 3, q = j % 3;
        if (q == 2 && !has_lo) return;        // (wave-uniform: every second-plane word of the group is zero)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[ty][q == 1 ? 1 : 0]),
                                                     __builtin_bit_cast(f16x8, q == 2 ? xl[ty] : xh[ty]), acc, 0, 0, 0);
      };
      // piece c (0..7) of a tile's epilogue: ReLU (the lower bound of the plane clamp) -> f16 planes of 16 x the stem output;
      // pieces 3 and 7 store chunk 2 ks + fh of the map
      unsigned h1[4], h2[4];
      auto epi = [&](int c, const f32x16& acc, char* o) {
        const int ks = c >> 2, e = (c >> 1) & 1, half = c & 1, q = 2 * ks + e;
        split2(acc[4 * q + 2 * half], acc[4 * q + 2 * half + 1], h1[2 * e + half], h2[2 * e + half], cl);
        if ((c & 3) == 3) {
          *reinterpret_cast<u32x4*>(o + ks * 2 * S8_BLK) = u32x4{h1[0], h1[1], h1[2], h1[3]};
          *reinterpret_cast<u32x4*>(o + ks * 2 * S8_BLK + 4 * S8_BLK) = u32x4{h2[0], h2[1], h2[2], h2[3]};
        }
      };
      f32x16 accA, accB;
      u32x4 xhA[3], xlA[3], xhB[3], xlB[3];
      int pixA = tile_pix(0), pixB = 0;
      rd_ops(in_ptr(pixA), xhA, xlA);
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 9; ++j) mma(j, accA, xhA, xlA);
#pragma unroll 1
      for (int k = 1; k < ntile; k += 2) {
        // tile k (B) under the epilogue of tile k - 1 (A)
        pixB = tile_pix(k);
        rd_ops(in_ptr(pixB), xhB, xlB);
        char* oA = lds + S8_S + fh * S8_BLK + pixA * PIXB + agent * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          mma(j, accB, xhB, xlB);
          W4_PIN();
          if (j < 8) { epi(j, accA, oA); W4_PIN(); }
        }
        // tile k + 1 (A) under the epilogue of tile k (B); the last round has no next tile
        const bool more = k + 1 < ntile;
        char* oB = lds + S8_S + fh * S8_BLK + pixB * PIXB + agent * 16;
        if (more) {
          pixA = tile_pix(k + 1);
          rd_ops(in_ptr(pixA), xhA, xlA);
#pragma unroll
          for (int r = 0; r < 16; ++r) accA[r] = 0.f;
#pragma unroll
          for (int j = 0; j < 9; ++j) {
            mma(j, accA, xhA, xlA);
            W4_PIN();
            if (j < 8) { epi(j, accB, oB); W4_PIN(); }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) epi(j, accB, oB);
        }
      }
      if (ntile & 1) {        // (an odd tile count ends on an A tile whose epilogue is still due)
        char* oA = lds + S8_S + fh * S8_BLK + pixA * PIXB + agent * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) epi(j, accA, oA);
      }
      clamped |= cl > 65504.f && mok;      // 16 x the stem output beyond the planes' range (stem output > 4094)
    }
    S8_STAMP(2);
    L3_LDS_SYNC();
    S8_STAMP(3);
    if (wave < 4) {
      // ---- phase 2, waves 0-3: layer1.conv1 over the stem map (stride 2)
      switch (wave) {
        case 0: stem8_conv1<W4P0>(p, lds, group, scale1, clamped); break;
        case 1: stem8_conv1<W4P1>(p, lds, group, scale1, clamped); break;
        case 2: stem8_conv1<W4P2>(p, lds, group, scale1, clamped); break;
        default: stem8_conv1<W4P3>(p, lds, group, scale1, clamped); break;
      }
      S8_STAMP(5);
    } else {
      // ---- phase 2, waves 4-7: the next group's maps (requested a whole group ago) -> planes; the request for the group after
      // it; this group's stem output at the stride-2 pixels (the block's residual input), unscaled: planes x 2^-4
      if (group + gstride < p.groups) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        to_planes();
        if (group + 2 * gstride < p.groups) fetch_raw(group + 2 * gstride);
      }
      S8_STAMP(4);
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      const h2v sc = {(_Float16)0.0625f, (_Float16)0.0625f};
      for (int item = t - 256; item < NPIX * AG * 4; item += 256) {
        const int a = item & 7, chunk = (item >> 3) & 3, opix = item >> 5;
        const int m = group * AG + a;
        const int oy = opix / 6, sp = 22 * oy + 2 * (opix - 6 * oy);
        const char* src = lds + S8_S + chunk * S8_BLK + sp * PIXB + a * 16;
        u32x4 hi = *reinterpret_cast<const u32x4*>(src), lo = *reinterpret_cast<const u32x4*>(src + 4 * S8_BLK);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned hv = hi[e], lv = lo[e];      // (scalar copies: bit-casting the vector element itself reads element 0)
          hi[e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2v, hv) * sc);
          lo[e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2v, lv) * sc);
        }
        if (m < p.M) {
          char* o = p.ctr + ((long long)(m >> 7) * NPIX + opix) * (128 * 32 * 4) + (m & 127) * 16 + chunk * 2048;
          S8_OUT_STORE(o, hi);
          S8_OUT_STORE(o + 256 * 32, lo);
        }
      }
      S8_STAMP(5);
    }
  }
  if (clamped && p.range_flag) atomicOr(p.range_flag, 1);
}

#ifdef MAGAT_DEBUG_HOOKS
long long* g_stem8_dbg = nullptr;
#endif
}  // namespace

#ifdef MAGAT_DEBUG_HOOKS
extern "C" int magat_stem8_set_debug_buffer(long long* dev_buf) { g_stem8_dbg = dev_buf; return MAGAT_OK; }
#endif

// Stem + layer1.conv1 for groups of eight agents (stem8_kernel): x (M,3,11,11) -> out, ctr as magat_layer1_fused writes them.
// w1f: layer1.conv1 as encoder.pack_chain_weights(rows, 32, 0) (fragment-major planes + [2^-e, 0, 0, 0]).
int magat_stem8(const float* x, const float* w0, const float* b0, const float* w1f, const float* b1, void* out, void* ctr,
                int M, int H, int W, hipStream_t st, int* range_flag) {
  if (!x || !w0 || !b0 || !w1f || !b1 || !out || !ctr) return MAGAT_ERR_NULL;
  if (M <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (H != 11 || W != 11) return MAGAT_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(b1) & 15) || (reinterpret_cast<uintptr_t>(w1f) & 15) || (reinterpret_cast<uintptr_t>(x) & 3))
    return MAGAT_ERR_UNSUPPORTED;
  Stem8Params p;
  p.x = x; p.w0 = w0; p.b0 = b0; p.w1 = reinterpret_cast<const char*>(w1f); p.b1 = b1;
  p.out = static_cast<char*>(out); p.ctr = static_cast<char*>(ctr);
  p.M = M; p.groups = (M + AG - 1) / AG;
  p.x_aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? 1 : 0;      // (a slice of a batch may start at any float)
  p.range_flag = range_flag;
  p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_stem8_dbg;        // (magat_stem8_set_debug_buffer: [grid][8][16])
#endif
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&stem8_kernel), MAGAT_LDS_STEM8, S8_TOTAL) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int grid = p.groups < cus ? p.groups : cus;
  const int pid = magat_prof_begin(MAGAT_TAG_CONV_FIRST, st);
  hipLaunchKernelGGL(stem8_kernel, dim3((unsigned)grid), dim3(512), S8_TOTAL, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
