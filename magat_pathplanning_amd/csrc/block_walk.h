// Shared pieces of the eight-agent-group kernels (block_fused.hip: the BasicBlock chain kernels; stem8.hip: stem +
// layer1.conv1): LDS map geometry, row tiles by tap-validity class, the plane split, and walk4 - the four-wave K walk.
// Included inside each file's anonymous namespace.
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


constexpr int AG = 8;                       // agents per workgroup
constexpr int NPIX = 36, ZPIX = 36;         // 6 x 6 map; pixel slot 36 = zeros
constexpr int PIXB = AG * 16;               // bytes per pixel slot of one (plane, chunk) block: 8 agents x 16 B
constexpr int BLK = (NPIX + 1) * PIXB;      // 4736 bytes per (plane, chunk) block
constexpr int MAP32 = 8 * BLK;              // 32-channel map: 2 planes x 4 chunks
constexpr int MAP64 = 16 * BLK;             // 64-channel map: 2 planes x 8 chunks
// regions: [0, MAP64) Z (stage B output), its second half doubling as the residual input X2 of stage A; Y; the main input X1
constexpr int LDS_X2 = MAP32, LDS_Z = 0, LDS_Y = MAP64, LDS_X1N = MAP64 + MAP32, LDS_TOTAL = MAP64 + 2 * MAP32;
constexpr int TAPBIAS = 7 * PIXB;           // tap shifts are (6 dy + dx) pixel slots in [-7, 7]: biased to stay non-negative

// row tiles by tap-validity class: pixel = 6 y + x
// The order INSIDE a tile (the pixel of slot psl = (lane & 31) >> 3) and the split of the 16 interior pixels over the four
// interior tiles are chosen for the 2 x 2 pooling behind layer3 (block_full_p_kernel): the four pixels of every corner cell
// sit at the SAME slot of four different tiles (C, one row edge, one column edge, interior tile 0) - a lane sums them in
// registers - the edge-middle cells are two in-lane pairs one lane-bit apart (slots {0,1} / {2,3}: DPP row_ror:8; slots
// {0,2} / {1,3}: lanes 16 apart), and the centre cell is interior tile 3.  Slot pairs {0,1} and {2,3} hold pixels of
// opposite parity wherever the class allows it (ds_read_b128 is served 16 lanes = two slots at a time: conflict-free);
// the two column-edge tiles are 2-way, as with any order.
__device__ constexpr int TILE_PIX[9][4] = {{7, 10, 25, 28},  {26, 27, 8, 9},   {16, 13, 22, 19}, {14, 15, 20, 21},
                                           {1, 4, 2, 3},     {32, 33, 31, 34}, {6, 12, 24, 18},  {17, 11, 23, 29},
                                           {0, 5, 30, 35}};
// TILE_PIX[tile][psl] for a tile known at compile time and the lane's slot: a shift of the four bytes as a literal.  (The table
// lookup is a global load - an L2 round trip in front of every walk's first operand read and every epilogue's first store,
// because the thread index they start from is laundered against hoisting; round 4.)
__device__ __forceinline__ int tile_pix(int tile, int psl) {
  const unsigned packed = (unsigned)TILE_PIX[tile][0] | (unsigned)TILE_PIX[tile][1] << 8 | (unsigned)TILE_PIX[tile][2] << 16 |
                          (unsigned)TILE_PIX[tile][3] << 24;
  return (int)((packed >> (8 * psl)) & 0xFFu);
}
// valid taps t = 3 (dy + 1) + (dx + 1) of every pixel of the tile (corners: union; per-lane validity handled by address)
__device__ constexpr int TILE_TAPS[9] = {0x1FF, 0x1FF, 0x1FF, 0x1FF, 0x1F8, 0x03F, 0x1B6, 0x0DB, 0x1FF};
constexpr int T_I0 = 0, T_I1 = 1, T_I2 = 2, T_I3 = 3, T_ET = 4, T_EB = 5, T_EL = 6, T_ER = 7, T_C = 8;
// wave -> row tiles.  32 output channels: one channel tile, waves split the nine row tiles; 64: wave = (channel tile w & 1,
// row group w >> 1) - waves w and w + 4 share a SIMD: their tap counts add up to 33 / 36 per SIMD
// (SIMD sums of MFMAs for 32 -> 32 + residual: 120, 120, 102, 126)
__device__ constexpr int WT32[8][3] = {{T_I0, -1, -1}, {T_I2, -1, -1}, {T_C, -1, -1},  {T_EB, T_EL, -1},
                                       {T_I1, -1, -1}, {T_I3, -1, -1}, {T_ET, -1, -1}, {T_ER, -1, -1}};
__device__ constexpr int WG64[4][3] = {{T_I0, T_I1, -1}, {T_I2, T_I3, -1}, {T_C, T_ET, -1}, {T_EB, T_EL, T_ER}};


// barrier for LDS hand-overs only: __syncthreads() carries s_waitcnt vmcnt(0) in its release fence and would wait for every
// global load in flight (weight prefetches, the next group's input)
#define L3_LDS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)

// value pair -> its two f16 planes; ReLU and the f16 range clamp are the same v_med3.  `vmax` keeps the running maximum of the
// UNclamped values (one v_max3 per pair; the caller compares it with 65504 once per tile for the range guard).
__device__ __forceinline__ void split2(float x, float y, unsigned& p1, unsigned& p2, float& vmax) {
  vmax = fmaxf(fmaxf(vmax, x), y);
  x = __builtin_amdgcn_fmed3f(x, 0.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, 0.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  // residual x - hi: one mixed-precision fma per value (fma(hi, -1, x), exact; the f16 operand read from its half of the
  // packed register) instead of two conversions and a packed subtract.  (v_fma_mixlo_f16 / v_fma_mixhi_f16 - the same fma
  // with the conversion folded in, two instructions per pair instead of three - was measured in round 4: chain kernel 1900 ->
  // 1904 us, stem 165 -> 169 us same-box: not cheaper.)
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}


struct W4Item { signed char tp, ks, s, first; };           // tp = 9: the residual 1x1 segment over in2
struct W4All { static constexpr int NT = 9; static constexpr int t[9] = {T_I0, T_I1, T_I2, T_I3, T_C, T_ET, T_EB, T_EL, T_ER}; };
struct W4A { static constexpr int NT = 4; static constexpr int t[4] = {T_I0, T_I1, T_I2, T_ET}; };           // 33 tile-taps
struct W4B { static constexpr int NT = 5; static constexpr int t[5] = {T_I3, T_C, T_EB, T_EL, T_ER}; };      // 36 tile-taps
// the same nine tiles split the other way round - the FOUR interior tiles (36 tile-taps, 4 epilogues) against the FIVE edge and
// corner tiles (33 tile-taps, 5 epilogues): walk and split-and-store epilogue together are then 0.8 k cycles apart instead of
// 2.3 k (W4A / W4B give the longer walk ALSO the fifth epilogue), which is what the lighter pair of waves waits at the barrier
struct W4I { static constexpr int NT = 4; static constexpr int t[4] = {T_I0, T_I1, T_I2, T_I3}; };
struct W4E { static constexpr int NT = 5; static constexpr int t[5] = {T_C, T_ET, T_EB, T_EL, T_ER}; };
constexpr int W4_TAPS[9] = {0x1FF, 0x1FF, 0x1FF, 0x1FF, 0x1F8, 0x03F, 0x1B6, 0x0DB, 0x1FF};     // = TILE_TAPS, host-visible
template <int NMAX> struct W4Seq { W4Item it[NMAX]; int n, nmain; };
template <typename TL, int KSM, int KS2>
constexpr W4Seq<(9 * KSM + KS2) * TL::NT + 1> w4_seq() {
  W4Seq<(9 * KSM + KS2) * TL::NT + 1> q{};
  int n = 0;
  for (int tp = 0; tp < 9; ++tp)
    for (int ks = 0; ks < KSM; ++ks) {
      bool first = true;
      for (int s = 0; s < TL::NT; ++s)
        if (W4_TAPS[TL::t[s]] >> tp & 1) {
          q.it[n].tp = (signed char)tp; q.it[n].ks = (signed char)ks; q.it[n].s = (signed char)s; q.it[n].first = first;
          first = false;
          ++n;
        }
    }
  q.nmain = n;
  for (int ks = 0; ks < KS2; ++ks)
    for (int s = 0; s < TL::NT; ++s) {
      q.it[n].tp = 9; q.it[n].ks = (signed char)ks; q.it[n].s = (signed char)s; q.it[n].first = s == 0;
      ++n;
    }
  q.n = n;
  return q;
}
// smallest pixel shift (6 dy + dx) over the active taps of a tile: the tile's base address points at that neighbour, so that
// every immediate offset is non-negative
constexpr int w4_minshift(int tile, int rp = 6) {
  int m = 99;
  for (int tp = 0; tp < 9; ++tp)
    if (W4_TAPS[tile] >> tp & 1) {
      const int sh = rp * (tp / 3 - 1) + (tp % 3 - 1);
      if (sh < m) m = sh;
    }
  return m;
}
// Geometry of the LDS map a walk reads.  GeoChain: the 6 x 6 maps of the BasicBlock chain, stride 1 - output pixel pix reads
// input pixel pix + 6 dy + dx.  GeoStem: layer1.conv1 (stride 2, pad 1) over the 11 x 11 stem map of stem8_kernel - output
// pixel (y, x) reads stem pixel (2 y + dy, 2 x + dx); 121 pixel slots + the zero slot per (plane, chunk) block.  The tap
// validity classes of the row tiles are the same (ty = 0 leaves the map for y = 0, ty = 2 for y = 5: 2 * 5 + 1 = 11).
struct GeoChain {
  static constexpr int RP = 6, BLKB = BLK, ZP = ZPIX;
  static __device__ __forceinline__ int base(int pix) { return pix; }
};
struct GeoStem {
  static constexpr int RP = 11, BLKB = 122 * PIXB, ZP = 121;
  static __device__ __forceinline__ int base(int pix) { const int y = pix / 6; return 22 * y + 2 * (pix - 6 * y); }
};

// KSM / KS2: k steps (16 channels) per tap of the main input / of the residual input; PS_IN / PS_IN2: plane strides of the two
// LDS maps; D: weight ring (fetched D - 1 k steps ahead).
#ifndef MAGAT_W4_D
#define MAGAT_W4_D 4
#endif
#ifndef MAGAT_W4_AV
#define MAGAT_W4_AV 3
#endif
#ifdef MAGAT_W4_NOPIN
#define W4_PIN() do { } while (0)
#else
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int w4_depth(int nt, int ksm) { return nt * ksm <= 8 ? 8 : MAGAT_W4_D; }    // short k steps (few MFMAs): deeper weight ring
// first D - 1 k steps of a weight stream into the ring
// (lane16 comes from the caller's laundered thread index: formed from threadIdx here, the D - 1 fragment addresses are
//  invariant in the persistent group loop, get hoisted into 64-bit register pairs, spilled, and re-loaded from scratch in
//  the middle of the fill - each reload's wait sits out the fill's earlier loads)
template <int NSTEP, int D>
__device__ __forceinline__ void w4_fill(const char* wbase, unsigned lane16, u32x4 (&w)[D][2]) {
#pragma unroll
  for (int j = 0; j < D - 1; ++j)
    if (j < NSTEP) {
      w[j][0] = *reinterpret_cast<const u32x4*>(wbase + (size_t)j * 2048 + lane16);
      w[j][1] = *reinterpret_cast<const u32x4*>(wbase + (size_t)j * 2048 + (lane16 + 1024u));
    }
}
// (Filling the ring one stage ahead - before the previous stage's epilogue and an LDS-only barrier - was measured and is
//  slower: chain 608-624 -> 659 us, layer3 1.50-1.52 -> 1.53 ms same-box; the rings of two stages then overlap in registers.)
// NA: length of the caller's accumulator array (>= TL::NT: roles with unlike tile counts share one array and one epilogue);
// rpix (RTPIX): the tiles' pixels of this lane as run-time values - two roles whose tile lists have the same STRUCTURE (tap
// masks, corner flag) then share one body (the compact form of the one-launch kernel, block_fused.hip).
template <typename TL, int KSM, int KS2, int PS_IN, int PS_IN2, int D, typename GEO = GeoChain, int NA = TL::NT, bool RTPIX = false>
__device__ __forceinline__ void walk4(char* lds, int in_off, int in2_off, const char* wbase, f32x16 (&acc)[NA], bool with_res,
                                      const int* rpix = nullptr) {
  static_assert(NA >= TL::NT, "accumulator array shorter than the tile list");
  constexpr int RP = GEO::RP, GBLK = GEO::BLKB;
  u32x4 w[D][2];
  constexpr int NT = TL::NT;
  constexpr auto SQ = w4_seq<TL, KSM, KS2>();
  constexpr int NMAIN = 9 * KSM, NSTEP = NMAIN + KS2;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));        // (keeps the address set-up inside the caller's loop over the halves)
  const int lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int fr = lane & 31, fh = lane >> 5, agent = fr & 7, psl = fr >> 3;
  unsigned ab[NT], b0[NT];
  int cmask = 0;                         // corner tile: the lane's valid taps
  unsigned cab = 0;
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    const int pix = RTPIX ? rpix[s] : tile_pix(TL::t[s], psl);
    ab[s] = (unsigned)(GEO::base(pix) * PIXB + agent * 16 + fh * GBLK);
    b0[s] = ab[s] + (unsigned)(in_off + w4_minshift(TL::t[s], RP) * PIXB);
    if (TL::t[s] == T_C) {
      const int y = pix / 6, x = pix - 6 * y;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        if (y + dy >= 0 && y + dy < 6 && x + dx >= 0 && x + dx < 6) cmask |= 1 << tp;
      }
      cab = ab[s];
    }
  }
  const unsigned az = (unsigned)(GEO::ZP * PIXB + agent * 16 + fh * GBLK + in_off);
  auto load_w = [&](int step, u32x4 (&b)[2]) {
    b[0] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + lane16);
    b[1] = *reinterpret_cast<const u32x4*>(wbase + (size_t)step * 2048 + (lane16 + 1024u));
  };
  constexpr int AV = MAGAT_W4_AV;         // operand ring: reads run AV - 1 items ahead of the MFMAs
  u32x4 av[AV][2];
  auto rd = [&](const W4Item it, int pl, u32x4& dst) {
    const int tile = TL::t[it.s];
    if (it.tp == 9) {                                    // residual: the block input at the same pixel (chain geometry)
      dst = *reinterpret_cast<const u32x4*>(lds + (ab[it.s] + (unsigned)in2_off) + (pl * PS_IN2 + it.ks * 2 * BLK));
    } else if (tile == T_C) {                            // corner pixels: per-lane validity, the others read the zero pixel
      const int sh = RP * (it.tp / 3 - 1) + (it.tp % 3 - 1);
      const unsigned a = (cmask >> it.tp & 1) ? cab + (unsigned)(in_off + sh * PIXB) : az;
      dst = *reinterpret_cast<const u32x4*>(lds + a + (pl * PS_IN + it.ks * 2 * GBLK));
    } else {
      const int sh = RP * (it.tp / 3 - 1) + (it.tp % 3 - 1) - w4_minshift(tile, RP);
      dst = *reinterpret_cast<const u32x4*>(lds + b0[it.s] + (sh * PIXB + pl * PS_IN + it.ks * 2 * GBLK));
    }
  };
  w4_fill<NSTEP, D>(wbase, lane16, w);
#pragma unroll
  for (int j = 0; j < AV - 1; ++j) {
    rd(SQ.it[j], 0, av[j][0]);
    rd(SQ.it[j], 1, av[j][1]);
  }
#pragma clang loop unroll(full)
  for (int i = 0; i < SQ.n; ++i) {
    const W4Item it = SQ.it[i];
    if (KS2 > 0 && i == SQ.nmain && !with_res) break;            // (uniform; the first half of conv2 has no residual)
    const int step = it.tp * KSM + it.ks;              // (residual items: tp = 9)
    if (it.first && step + D - 1 < NSTEP) {
      if (step + D - 1 < NMAIN || with_res) load_w(step + D - 1, w[(step + D - 1) % D]);
    }
    constexpr int LA = AV - 1;
    const bool more = i + LA < SQ.n;
    const bool more_ok = more && (i + LA < SQ.nmain || with_res);
    acc[it.s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][0]),
                                                       __builtin_bit_cast(f16x8, av[i % AV][0]), acc[it.s], 0, 0, 0);
    W4_PIN();
    if (more) { if (more_ok) rd(SQ.it[more ? i + LA : i], 0, av[(i + LA) % AV][0]); W4_PIN(); }
    acc[it.s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][1]),
                                                       __builtin_bit_cast(f16x8, av[i % AV][0]), acc[it.s], 0, 0, 0);
    W4_PIN();
    if (more) { if (more_ok) rd(SQ.it[more ? i + LA : i], 1, av[(i + LA) % AV][1]); W4_PIN(); }
    acc[it.s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[step % D][0]),
                                                       __builtin_bit_cast(f16x8, av[i % AV][1]), acc[it.s], 0, 0, 0);
    W4_PIN();
  }
}


// row-tile split of a one-channel-tile stage over the four waves (chain stage A, layer1.conv1): 18 / 18 / 15 / 18 tile-taps
struct W4P0 { static constexpr int NT = 2; static constexpr int t[2] = {T_I0, T_I1}; };
struct W4P1 { static constexpr int NT = 2; static constexpr int t[2] = {T_I2, T_I3}; };
struct W4P2 { static constexpr int NT = 2; static constexpr int t[2] = {T_C, T_ET}; };
struct W4P3 { static constexpr int NT = 3; static constexpr int t[3] = {T_EB, T_EL, T_ER}; };

