// fp32-accurate segmented-K conv/linear GEMM on the 16-bit matrix cores ("split MFMA": f16x3 and bf16x6 flavours).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TF); v_mfma_f32_32x32x16_{f16,bf16} is 16x faster.  Every fp32
// value is therefore carried as a few 16-bit planes and a product as the partial products that matter:
//   f16x3  (NPL = 2, default): x = h1 + h2, two RNE half-precision planes (22 significand bits); x.w ~= h1g1 + h1g2 + h2g1
//          (dropped h2g2 <= 2^-22 |x.w|): THREE f16 MFMAs per product = 5.3x the fp32-MFMA rate.  fp16's narrow exponent:
//          activation planes are clamped to +-65504 (exact to 1.3e5 through the second plane), tiny residuals go
//          subnormal (absolute error <= 3e-8), weights are pre-scaled by a power of two that the epilogue undoes.
//          Measured against float64 it is as accurate as the fp32 MFMA kernel (tools/conv_bench_split.py).
//   bf16x6 (NPL = 3): x = x1 + x2 + x3, three bf16 planes (24 bits, fp32 exponent range); six products with i + j <= 4
//          (dropped terms <= 2^-24): 2.67x the fp32-MFMA rate.
//   NPL = 1: plain bf16 GEMM (bf16 storage variant of the GAT maps, BASELINE config 5).
// Activations normally stay float32 in HBM and are split by the loader (AF32) - after the wave has issued its MFMAs of
// the previous slab, outside the barrier-to-barrier section; weights are split host-side / by pack_kernel.  Plane
// tensors in HBM (AF32 = false; produced by out_fmt 1 / 3) exist as measured-and-rejected experiments.  Same
// pixel-major / tap-skipping / residual-K-segment structure as conv_gemm_f32.hip.
//
// Tile 128x128x32, 4 waves (2x2, wave tile 64x64).  LDS rows are 32 bf16 = 64 B, unpadded, with the 16-byte
// chunk index XOR-ed by (row>>2)&3: a ds_read_b128 16-lane group (16 consecutive rows, same k chunk) then covers
// all 64 banks exactly once.
#include <cstdlib>
#include <type_traits>

#include "magat_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 128, BK = 32;

struct SplitParams {
  const u16* in;
  const u16* in2;
  const u16* wt;
  const float* bias;
  void* out;
  long long in_pix_stride, in2_pix_stride, out_pix_stride;
  long long in_tile, in2_tile, out_tile;
  long long in_plane, in2_plane, out_plane, wt_plane;
  int M, Mt;
  int Cin, lda, Hin, Win, kH, kW, stride, pad, Hout, Wout;
  int C2, lda2, W2, stride2;
  int Cout, Ktot, ldc, relu;
  int ntn, npix, tag, out_split;
  long long out_nt;         // direct kernel, row-major output: column tile t at out + t * out_nt (0: column 128 t of the row)
  int tepi;                 // direct kernel: row-major float32 output transposed through LDS (always 1 since round 5)
  int korder;               // direct kernel: 1 = channel slab outer, taps inner (always 1 since round 5)
  int in_gl, out_gl;        // direct kernel: granule-major agent tiles [C/4][128][4] for in/in2 resp. out
  const float* acc_scale;   // NPL == 2: device pointer to 1 / (power-of-two weight scale), applied before the bias
  const float* in_scale;    // direct kernel, float32 input: power-of-two activation scale (device float; null or 0 = 1)
  int* range_flag;          // range guard (magat_hip.h): OR-ed with 1 when a value had to be clamped into its f16 / fp8 planes
  // second 1x1 layer computed in the epilogue (direct kernel, FUSE2; magat_conv_gemm_desc.wt2 ...): out2 = act(out . wt2^T + bias2)
  const u16* wt2;           // [2][Cout2][Cout] f16 planes of (weight * 2^e) followed by one float32 2^-e
  const float* bias2;
  float* out2;
  u16* out2h;               // the same rows as RNE bf16 (null: not written), row stride ldc2h
  int ldc2h;
  const float* in_scale2;   // power-of-two scale of `out` on its way into the planes (device float; null or 0 = 1)
  int ldc2, relu2;
};

// the guard's device-side flag: lanes that clamped (normally none: one skipped branch) OR it
__device__ __forceinline__ void report_clamped(int* flag, bool clamped) {
  if (clamped && flag) atomicOr(flag, 1);
}

__device__ __forceinline__ u16 bf16_rne(float v) { return magat_bf16_rne(v); }
__device__ __forceinline__ float bf16_f32(u16 h) { return magat_bf16_f32(h); }

// two floats -> two RNE bf16 packed in one dword (lo = a, hi = b)
// (through the compiler, not as inline asm: an asm statement that reads a register an MFMA has just written gets none of the
//  wait states the hazard recognizer inserts for instructions it knows - round 6, gat_csr_fused.hip)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2_t));
}
// (x, y) -> three packed bf16 pairs with x = x1+x2+x3, y likewise
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt_pk_bf16(x, y);
  const float rx = x - __builtin_bit_cast(float, p1 << 16), ry = y - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = cvt_pk_bf16(rx, ry);
  const float sx = rx - __builtin_bit_cast(float, p2 << 16), sy = ry - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = cvt_pk_bf16(sx, sy);
}

// fp16x2 split ("f16x3": x ~ h1 + h2, two RNE half planes = 22 significand bits; the product keeps h1g1 + h1g2 + h2g1,
// dropping h2g2 <= 2^-22 |xw|): THREE v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones.  fp16 has the
// narrow exponent, so activations are clamped to +-65504 per plane (values up to 1.3e5 stay exact through the second
// plane; larger ones saturate - unreachable for this network) and residuals below 6e-5 go subnormal (absolute error
// <= 3e-8, the fp32 spacing of values near 0.5); weights are pre-scaled by a power of two so that both planes are
// normal numbers (the scale is undone in the epilogue).
__device__ __forceinline__ void split_pair_f16(float x, float y, unsigned& p1, unsigned& p2) {
  // one v_med3_f32 per value (fminf(fmaxf()) costs an extra canonicalising v_max each)
  x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  // residual x - hi: one mixed-precision fma per value (fma(hi, -1, x), exact; the f16 operand read from its half of the
  // packed register) - not two conversions + v_pk_add_f32: packed fp32 instructions do not issue while an MFMA runs
  // (tools/exp/mfma_valu.hip), and the loaders split while other waves of the SIMD multiply
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}
// same, remembering in `clamped` whether a value was outside +-65504 (range guard)
__device__ __forceinline__ void split_pair_f16(float x, float y, unsigned& p1, unsigned& p2, bool& clamped) {
  // (negated compares: true for NaN as well - a NaN activation must take the float32 re-run, which hands it on like the
  // reference does, not come out of the v_med3 clamp as a finite number)
  clamped |= !(__builtin_fabsf(x) <= 65504.f) | !(__builtin_fabsf(y) <= 65504.f);
  split_pair_f16(x, y, p1, p2);
}

// AF32: the activation operands (in, in2) are plain float32 and are split into their three bf16 planes by the
// loader on the way into LDS (no 3-plane tensors in HBM, 2/3 of the activation traffic); weights are always
// pre-split bf16x3.
// BN = 128 / 64 / 32 output channels per tile; waves WGM x WGN, wave tile (128/WGM) x (BN/WGN).
// NPL = 3: the bf16x6 split product.  NPL = 1: plain bf16 storage (in_fmt 3: one bf16 plane for activations AND
// weights, one product) - the bf16 variant of the GAT maps GEMM for BASELINE config 5.
template <bool AF32, int BN, int WGM, int WGN, int NPL = 3>
__global__ __launch_bounds__(256, 3) void conv_gemm_bf16x6_kernel(const SplitParams p) {
  static_assert(NPL == 3 || (NPL == 1 && !AF32) || NPL == 2, "plane count");
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  constexpr int BI = BN >= 64 ? BN / 64 : 1;      // weight-tile 16-byte loads per thread and plane
  __shared__ __attribute__((aligned(16))) u16 lds[NPL * (BM + BN) * 32];   // A planes | B planes
  u16* As = lds;
  u16* Bs = lds + NPL * BM * 32;

  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per_m = p.npix * p.ntn;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / per_m);
  if (mtile >= p.Mt) return;
  const int rem = slot % per_m;
  const int pix = rem / p.ntn, ntile = rem % p.ntn;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int oy = pix / p.Wout, ox = pix % p.Wout;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  const int ty0 = iy0 < 0 ? -iy0 : 0, tx0 = ix0 < 0 ? -ix0 : 0;
  const int ty1 = min(p.kH, p.Hin - iy0), tx1 = min(p.kW, p.Win - ix0);
  const int ntaps = (ty1 - ty0) * (tx1 - tx0);
  const int spt = p.Cin / BK, spt2 = p.C2 / BK;
  const int nslab = ntaps * spt + spt2;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = t >> 2, lchunk = t & 3;          // loader: rows lrow, lrow+64; 16-byte chunk lchunk

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-thread byte offsets (rows past M are clamped: their results are never stored)
  unsigned aoff[2], aoff2[2], boff[2], loff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = min(m0 + lrow + 64 * i, p.M - 1);
    aoff[i] = (unsigned)((magat_row_off(m, p.lda, p.in_tile) + lchunk * 8) * 2);
    aoff2[i] = (unsigned)((magat_row_off(m, p.lda2, p.in2_tile) + lchunk * 8) * 2);
    const int nrow = min(lrow + 64 * i, BN - 1);      // BN = 32: upper half of the threads duplicate row BN-1
    boff[i] = (unsigned)(((long long)(n0 + nrow) * p.Ktot + lchunk * 8) * 2);
    const int row = lrow + 64 * i;
    loff[i] = (unsigned)((row * 4 + (lchunk ^ ((row >> 2) & 3))) * 16);
  }
  const bool bload = lrow < BN;                       // weight-tile loader participation (BN = 32)

  // AF32 loader: row fr0 + 32 i, float4 index fc4 (k = 4 fc4 .. +3): lands in 16-byte chunk fc4>>1, half fc4&1
  const int fr0 = t >> 3, fc4 = t & 7;
  unsigned faoff[4], faoff2[4], floff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + fr0 + 32 * i, p.M - 1);
    faoff[i] = (unsigned)((magat_row_off(m, p.lda, p.in_tile) + fc4 * 4) * 4);
    faoff2[i] = (unsigned)((magat_row_off(m, p.lda2, p.in2_tile) + fc4 * 4) * 4);
    const int row = fr0 + 32 * i;
    floff[i] = (unsigned)((row * 4 + ((fc4 >> 1) ^ ((row >> 2) & 3))) * 16 + (fc4 & 1) * 8);
  }

  // element size of the activation operands in global memory
  constexpr int AE = AF32 ? 4 : 2;
  int cur_ty = ty0, cur_tx = tx0, cur_ks = 0;
  bool cur_main = ntaps > 0;
  auto tap_base = [&](int ty, int tx) -> const char* {
    return reinterpret_cast<const char*>(p.in) + (long long)((iy0 + ty) * p.Win + (ix0 + tx)) * p.in_pix_stride * AE;
  };
  const char* cur_tap = tap_base(ty0, tx0);
  const char* const seg2_base = reinterpret_cast<const char*>(p.in2) +
                                (long long)(oy * p.stride2 * p.W2 + ox * p.stride2) * p.in2_pix_stride * AE;

  u32x4 ra[NPL][2], rb[NPL][BI];
  f32x4 fa32[4];
  auto load_slab = [&]() {
    const bool main_seg = cur_main;
    const int k0 = cur_ks * BK;
    const char* ab;
    long long aplane;
    int bk;
    if (main_seg) {
      ab = cur_tap + (long long)k0 * AE;
      aplane = p.in_plane * 2;
      bk = (cur_ty * p.kW + cur_tx) * p.Cin + k0;
      if (++cur_ks == spt) {
        cur_ks = 0;
        if (++cur_tx == tx1) {
          cur_tx = tx0;
          if (++cur_ty == ty1) cur_main = false;
        }
        if (cur_main) cur_tap = tap_base(cur_ty, cur_tx);
      }
    } else {
      ab = seg2_base + (long long)k0 * AE;
      aplane = p.in2_plane * 2;
      bk = p.kH * p.kW * p.Cin + k0;
      ++cur_ks;
    }
    const char* bb = reinterpret_cast<const char*>(p.wt + bk);
    const long long bplane = p.wt_plane * 2;
    const unsigned selp = main_seg ? 0xffffffffu : 0u;
    if constexpr (AF32) {
      // (selecting between the two offset arrays per element - or per branch - was compiled into scratch-indexed
      // loads on the address path; a masked delta keeps everything in registers)
      const unsigned sel = main_seg ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa32[i] = *reinterpret_cast<const f32x4*>(ab + (faoff2[i] + ((faoff[i] - faoff2[i]) & sel)));
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if constexpr (!AF32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          ra[pl][i] = *reinterpret_cast<const u32x4*>(ab + pl * aplane + (aoff2[i] + ((aoff[i] - aoff2[i]) & selp)));
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) rb[pl][i] = *reinterpret_cast<const u32x4*>(bb + pl * bplane + boff[i]);
    }
  };
  // AF32: the three-plane split of the freshly loaded activations runs right after a wave has issued its MFMAs of
  // the current slab (the loads landed long ago; the VALU work overlaps the wave's own MFMA drain and the other
  // waves' tails) instead of inside the barrier-to-barrier section, which is then only the LDS writes.
  uint2 qs[NPL][4];
  bool clamped = false;
  auto split_regs = [&]() {
    if constexpr (AF32 && NPL == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned q1[2], q2[2], q3[2];
        split_pair(fa32[i][0], fa32[i][1], q1[0], q2[0], q3[0]);
        split_pair(fa32[i][2], fa32[i][3], q1[1], q2[1], q3[1]);
        qs[0][i] = uint2{q1[0], q1[1]};
        qs[1][i] = uint2{q2[0], q2[1]};
        qs[2][i] = uint2{q3[0], q3[1]};
      }
    } else if constexpr (AF32 && NPL == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned q1[2], q2[2];
        split_pair_f16(fa32[i][0], fa32[i][1], q1[0], q2[0], clamped);
        split_pair_f16(fa32[i][2], fa32[i][3], q1[1], q2[1], clamped);
        qs[0][i] = uint2{q1[0], q1[1]};
        qs[1][i] = uint2{q2[0], q2[1]};
      }
    }
  };
  auto store_slab = [&]() {
    char* a = reinterpret_cast<char*>(As);
    char* b = reinterpret_cast<char*>(Bs);
    if constexpr (AF32) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2*>(a + pl * (BM * 64) + floff[i]) = qs[pl][i];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if constexpr (!AF32) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(a + pl * (BM * 64) + loff[i]) = ra[pl][i];
      }
#pragma unroll
      for (int i = 0; i < BI; ++i)
        if (bload) *reinterpret_cast<u32x4*>(b + pl * (BN * 64) + loff[i]) = rb[pl][i];
    }
  };

  // fragment addressing: row = tile row (lane&31), k chunk c = 2*s + (lane>>5)
  const int fr = lane & 31, fh = lane >> 5;
  auto frag = [&](const u16* base, int row, int s) -> u32x4 {
    const int c = 2 * s + fh;
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) +
                                            (row * 4 + (c ^ ((row >> 2) & 3))) * 16);
  };

  if (nslab > 0) {
    load_slab();
    split_regs();
  }
  for (int s = 0; s < nslab; ++s) {
    if (s > 0) __syncthreads();
    store_slab();
    __syncthreads();
    if (s + 1 < nslab) load_slab();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[TM][NPL], fb[TN][NPL];          // 8 packed 16-bit elements each (bf16 or f16 bits)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][pl] = frag(As + pl * BM * 32, wm * WTM + i * 32 + fr, ks);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j][pl] = frag(Bs + pl * BN * 32, wn * WTN + j * 32 + fr, ks);
      }
      // partial products in plane-arrival order (planes are read 0, 1, 2: the first MFMAs start while the later
      // fragments are still in flight; the running fp32 accumulator dwarfs every term of a slab, so adding the small
      // terms first buys no accuracy); the four accumulators are interleaved so consecutive MFMAs never depend on
      // each other.  (Tried without gain: s_setprio(1) around the MFMA cluster (3 % slower); reading both k-steps'
      // fragments before the first MFMA (the scheduler sinks the loads back, same 164 VGPRs, same time); for f16x3,
      // two LDS stages with ONE barrier per slab (64 KB, 2 workgroups per CU): 5-10 % slower than 3 workgroups per CU.)
      //   NPL 3 (bf16x6): x1w1 x1w2 x2w1 x2w2 x1w3 x3w1      NPL 2 (f16x3): h1g1 h1g2 h2g1      NPL 1: one product
      constexpr int NPROD = NPL == 3 ? 6 : (NPL == 2 ? 3 : 1);
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
      for (int q = 0; q < NPROD; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (NPL == 2)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j][PB[q]]),
                                                                 __builtin_bit_cast(f16x8, fa[i][PA[q]]), acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j][PB[q]]),
                                                                  __builtin_bit_cast(bf16x8, fa[i][PA[q]]), acc[i][j], 0, 0, 0);
          }
    }
    if (s + 1 < nslab) split_regs();
  }

  // epilogue: weights are the MFMA row operand -> D[channel][agent]; agent = lane&31,
  // channel = (r&3) + 8*(r>>2) + 4*(lane>>5): four consecutive channels per register quad -> wide stores
  const bool vec = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && (p.out_plane & 3) == 0;
  float acc_scale = 1.f;
  if constexpr (NPL == 2) acc_scale = *p.acc_scale;
  f32x4 bq[TN][4];          // the lane's bias quads, fetched as one batch (not one L2 round trip per quad)
  const bool bvec = p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * WTN + j * 32 + 4 * (lane >> 5) + 8 * q;
      if (bvec) {
        bq[j][q] = *reinterpret_cast<const f32x4*>(p.bias + n);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[j][q][c] = p.bias ? p.bias[n + c] : 0.f;
      }
    }
  if constexpr (NPL == 1 && TM == 2 && TN == 2) {
    if (p.out_split == 2 && vec && (p.ldc & 7) == 0) {
      // One bf16 plane, row-major (the maps of the bf16-storage graph layer: 393 MB per config-5 step): the accumulator layout
      // gives a lane 8-byte pieces of ONE agent's row, i.e. a store instruction scatters 64 pieces over 32 rows 3 KB apart.
      // Transposed through the (now idle) stages - 4 KB per wave and 32-agent pass, 16-byte units XOR-swizzled by the agent -
      // every store instruction writes eight agents' 128-byte runs.
      __syncthreads();                                   // every wave is done reading the stages
      char* const wl = reinterpret_cast<char*>(lds) + wave * 4096;
      const int fr = lane & 31, fh = lane >> 5;
      u16* const outp = static_cast<u16*>(p.out) + (long long)pix * p.out_pix_stride + n0 + wn * WTN;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[i][j][4 * q + c] + bq[j][q][c];
              if (p.relu) v[c] = magat_relu(v[c]);
            }
            const int u = j * 4 + q;                     // 16-byte unit of channels 32 j + 8 q .. + 7 (this lane: half fh)
            *reinterpret_cast<uint2*>(wl + fr * 128 + ((u ^ (fr & 7)) << 4) + 8 * fh) =
                uint2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
          }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int r = st * 8 + (lane >> 3), u = lane & 7;
          const uint4 v = *reinterpret_cast<const uint4*>(wl + r * 128 + ((u ^ (r & 7)) << 4));
          const int m = m0 + wm * WTM + i * 32 + r;
          if (m < p.M) *reinterpret_cast<uint4*>(outp + magat_row_off(m, p.ldc, p.out_tile) + 8 * u) = v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + wn * WTN + j * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + i * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float av = acc[i][j][4 * q + c];
          if constexpr (NPL == 2) av *= acc_scale;
          v[c] = av + bq[j][q][c];
          if (p.relu) v[c] = magat_relu(v[c]);
        }
        const long long o = (long long)pix * p.out_pix_stride + magat_row_off(m, p.ldc, p.out_tile) + n;
        if (p.out_split == 2) {            // one RNE bf16 plane
          u16* dst = static_cast<u16*>(p.out) + o;
          if (vec) {
            *reinterpret_cast<uint2*>(dst) = uint2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[c] = bf16_rne(v[c]);
          }
        } else if (p.out_split == 3) {     // two f16 planes (the operand format of the next f16x3 layer: split once here,
          u16* ob = static_cast<u16*>(p.out);   // not once per tap and slab by every consumer)
          unsigned a1, a2, b1, b2;
          split_pair_f16(v[0], v[1], a1, a2, clamped);
          split_pair_f16(v[2], v[3], b1, b2, clamped);
          if (vec) {
            *reinterpret_cast<uint2*>(ob + o) = uint2{a1, b1};
            *reinterpret_cast<uint2*>(ob + p.out_plane + o) = uint2{a2, b2};
          } else {
            const unsigned w1[2] = {a1, b1}, w2[2] = {a2, b2};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              ob[o + c] = (u16)(w1[c >> 1] >> (16 * (c & 1)));
              ob[p.out_plane + o + c] = (u16)(w2[c >> 1] >> (16 * (c & 1)));
            }
          }
        } else if (p.out_split) {
          u16* ob = static_cast<u16*>(p.out);
          u16 h[3][4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            h[0][c] = bf16_rne(v[c]);
            const float r1 = v[c] - bf16_f32(h[0][c]);
            h[1][c] = bf16_rne(r1);
            h[2][c] = bf16_rne(r1 - bf16_f32(h[1][c]));
          }
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            u16* dst = ob + pl * p.out_plane + o;
            if (vec) {
              uint2 pk;
              pk.x = (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16);
              pk.y = (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16);
              *reinterpret_cast<uint2*>(dst) = pk;
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) dst[c] = h[pl][c];
            }
          }
        } else if (vec) {
          *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) static_cast<float*>(p.out)[o + c] = v[c];
        }
      }
    }
  }
  if constexpr (NPL == 2) report_clamped(p.range_flag, clamped);
}

// ---- f16x3, activations straight into registers ("direct" flavour; in_fmt 4, out_fmt 0) --------------------------------
// The 2x2 kernel above moves every operand byte VGPR -> LDS -> VGPR; on gfx950 the VGPR -> LDS store path runs at only
// ~80 B/clk/CU, so with three f16 MFMAs per product the LDS (656 cycles per 128x128x32 slab) sits next to the MFMA pipe
// (768) and the two barely overlap.  Here the four waves split the tile by ROWS (wave tile 32 x BN): a wave's activation
// fragment is needed by nobody else, so it goes global -> registers -> f16 planes -> MFMA operand and never touches LDS;
// the weight slab is needed by all four waves and travels global -> LDS with the LDS-direct load (no VGPRs, no
// ds_write), double-buffered, ONE barrier per slab with no stores inside it.  LDS traffic per slab: 64 KB of
// ds_read_b128 (256 B/clk) + the 16 KB DMA fill instead of + 32 KB of ds_write.
// TM = 32-agent row groups per wave (workgroup tile 128 TM agents x BN): TM = 2 halves the weight traffic (L2 -> LDS
// fill and ds_read per MFMA) at 2 instead of 3 waves per SIMD.
// PIN: in / in2 arrive as f16 PLANE PAIRS in the plane-granule layout (in_gl = 2, magat_hip.h): the producer's epilogue
// split every value once, the loader here fetches finished 16-byte MFMA operands and does no VALU work at all (with
// float32 input every value is split once per tap it is used by - 7 times on a 6x6 map).
// FUSE2 (BN = 128 = Cout, TM = 1, float32 input, one output pixel, row-major float32 output; round 5): a SECOND 1x1 layer of
// 128 outputs on the workgroup's finished rows - compressMLP behind the encoder head - computed in the epilogue: a wave holds
// its 32 agents' complete 128-wide rows in its accumulators, so the rows become the second layer's activation planes in
// registers (the same scale, split and k order as the layer's own launch: one v_permlane32_swap per register pair turns the
// accumulator layout - a lane half holds channels 8 q + 4 h + c - into the natural k order of an operand), its four weight slabs
// travel through the idle weight stages, 96 more MFMAs per wave.  Bit for bit the result of the two launches it replaces (the
// same products in the same order on the same planes); saves a launch and the re-read of the rows (19 us -> ~8 at c3).
template <int BN, int TM, int INF, bool FUSE2 = false>
__global__ __launch_bounds__(256, (TM == 1 && !FUSE2) ? 3 : 2) void conv_gemm_f16x3_direct_kernel(const SplitParams p) {
  constexpr bool PIN = INF >= 1;
  static_assert(!FUSE2 || (BN == 128 && TM == 1 && INF == 0), "FUSE2: whole 128-wide rows per wave, float32 input");
  constexpr int TN = BN / 32;
  constexpr int STAGE = 2 * BN * 64;                    // bytes per weight stage: two planes of BN rows x 64 B
  // (FUSE2: two more stages - the second layer's first two weight slabs are requested before the first layer's rows are stored
  //  and land under that store; two workgroups per CU either way: 64 KB of LDS, 256 registers - its epilogue holds 64
  //  accumulator and 64 activation-plane registers next to the operands and spilled 56 registers at three waves per SIMD,
  //  which cost more than the launch it saves: 93.6 us against 71.6 + 18.6)
  __shared__ __attribute__((aligned(1024))) char Bs[(FUSE2 ? 4 : 2) * STAGE];

  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per_m = p.npix * p.ntn;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / per_m);
  if (mtile >= p.Mt) return;
  const int rem = slot % per_m;
  const int pix = rem / p.ntn, ntile = rem % p.ntn;
  const int m0 = mtile * (BM * TM), n0 = ntile * BN;
  const int oy = pix / p.Wout, ox = pix % p.Wout;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  const int ty0 = iy0 < 0 ? -iy0 : 0, tx0 = ix0 < 0 ? -ix0 : 0;
  const int ty1 = min(p.kH, p.Hin - iy0), tx1 = min(p.kW, p.Win - ix0);
  const int ntaps = (ty1 - ty0) * (tx1 - tx0);
  const int spt = p.Cin / BK, spt2 = p.C2 / BK;
  const int nslab = ntaps * spt + spt2;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Activation fragment of this lane: row m and 8 k values per k step of every slab (rows past M clamped).
  //   float32 row-major tiles:     floats k = 16 ks + 8 fh .. +7            (one 64-B line per lane pair)
  //   float32 granule-major tiles (in_gl 1, [C/4][128 agents][4 floats]): the same floats, but each 16-byte load sits
  //                                next to the neighbour lanes' (512 contiguous bytes per half wave)
  //   f16 plane granules (PIN, in_gl 2): one finished 16-byte operand per plane and k step
  // Four 16-byte loads per row group and slab in every case, at a + {0, d1, d2, d2 + d1}.
  unsigned aoff[TM], aoff2[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = min(m0 + 32 * (TM * wave + i) + fr, p.M - 1);
    if (PIN) {
      aoff[i] = (unsigned)((mrow >> 7) * p.in_tile * 4 + fh * 2048 + (mrow & 127) * 16);
      aoff2[i] = (unsigned)((mrow >> 7) * p.in2_tile * 4 + fh * 2048 + (mrow & 127) * 16);
    } else if (p.in_gl) {
      aoff[i] = (unsigned)(((mrow >> 7) * p.in_tile + (2 * fh * 128 + (mrow & 127)) * 4) * 4);
      aoff2[i] = (unsigned)(((mrow >> 7) * p.in2_tile + (2 * fh * 128 + (mrow & 127)) * 4) * 4);
    } else {
      aoff[i] = (unsigned)((magat_row_off(mrow, p.lda, p.in_tile) + 8 * fh) * 4);
      aoff2[i] = (unsigned)((magat_row_off(mrow, p.lda2, p.in2_tile) + 8 * fh) * 4);
    }
  }
  // activation scale (float32 input only): a power of two applied on the way into the planes, undone with the weight scale
  float insc = 1.f;
  if constexpr (!PIN) {
    if (p.in_scale) insc = *p.in_scale;
    if (insc == 0.f) insc = 1.f;
  }
  const bool scaled_in = !PIN && insc != 1.f;
  const int kmul = PIN ? 256 : (p.in_gl ? 512 : 4);     // bytes per unit of k0 (k0 % 32 == 0)
  // PIN: d1 = k step (+4096), d2 = plane (+256 C bytes: C differs between in and in2);  float32: d1 = second quad,
  // d2 = k step
  const int d1 = PIN ? 4096 : (p.in_gl ? 2048 : 16);
  const int d2m = PIN ? 256 * p.lda : (p.in_gl ? 8192 : 64), d2s = PIN ? 256 * p.lda2 : d2m;

  // weight slab pieces (1 KB = 16 rows x 64 B of one plane) this wave copies: piece id = wave + 4 i; lane -> LDS bytes
  // [16 lane, +16) of the piece, i.e. row lane/4, chunk slot lane%4, which holds k chunk slot ^ ((row>>2)&3)
  // (32-bit byte offsets - the launcher refuses weight tensors of 4 GB - and the LDS address formed from the wave number at
  // the point of use: the 64-bit offsets + LDS addresses were 12 registers, the kernel sits at the 168-register limit of
  // three waves per SIMD, and what the compiler spilled was re-loaded from scratch inside the K loop: a vector-memory load
  // whose wait also sat out the activation loads in flight)
  unsigned boff[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int id = wave + 4 * i;
    const int plane = id / (BN / 16), row = (id % (BN / 16)) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    boff[i] = (unsigned)(((long long)plane * p.wt_plane + (long long)(n0 + row) * p.Ktot + c * 8) * 2);
  }

  int cur_ty = ty0, cur_tx = tx0, cur_ks = 0;
  bool cur_main = ntaps > 0;
  auto tap_base = [&](int ty, int tx) -> const char* {
    return reinterpret_cast<const char*>(p.in) + (long long)((iy0 + ty) * p.Win + (ix0 + tx)) * p.in_pix_stride * 4;
  };
  const char* cur_tap = tap_base(ty0, tx0);
  const char* const seg2_base = reinterpret_cast<const char*>(p.in2) +
                                (long long)(oy * p.stride2 * p.W2 + ox * p.stride2) * p.in2_pix_stride * 4;
  const char* const wtb = reinterpret_cast<const char*>(p.wt);

  u32x4 fa[TM][4];          // the next slab as loaded: float32 quads, or (PIN) the operands [plane][k step] themselves
  // loads of the next slab: advance() moves the tap cursor and leaves the addresses in na / nb / nd2, load_a(i) fetches
  // row group i into fa, dma(i, stage) sends weight piece i to LDS stage `stage`
  const char* na[TM];
  const char* nb;
  int nd2;
  auto advance = [&]() {
    const bool main_seg = cur_main;
    const int k0 = cur_ks * BK;
    const char* ab;
    int bk;
    if (main_seg) {
      ab = cur_tap + (long long)k0 * kmul;
      bk = (cur_ty * p.kW + cur_tx) * p.Cin + k0;
      if (p.korder) {
        // channel slab OUTER, taps inner: the workgroups of an agent tile (one per output pixel, all on one XCD) then sweep
        // the SAME 32-channel slab of the input map at the same time - the <= 9 re-reads of a (pixel, slab) chunk by
        // neighbouring output pixels fall into a short window and hit in L2.  Tap-major order spreads them over the
        // whole K walk, the tile's input (4.7 MB at layer 3) does not fit the 4 MB L2, and every chunk came back from the
        // fabric 3.6 times (rocprofv3 FETCH_SIZE, profiles/r01f).
        if (++cur_tx == tx1) {
          cur_tx = tx0;
          if (++cur_ty == ty1) {
            cur_ty = ty0;
            if (++cur_ks == spt) { cur_ks = 0; cur_main = false; }
          }
        }
        if (cur_main) cur_tap = tap_base(cur_ty, cur_tx);
      } else if (++cur_ks == spt) {
        cur_ks = 0;
        if (++cur_tx == tx1) {
          cur_tx = tx0;
          if (++cur_ty == ty1) cur_main = false;
        }
        if (cur_main) cur_tap = tap_base(cur_ty, cur_tx);
      }
    } else {
      ab = seg2_base + (long long)k0 * kmul;
      bk = p.kH * p.kW * p.Cin + k0;
      ++cur_ks;
    }
    nb = wtb + (long long)bk * 2;
    const unsigned sel = main_seg ? 0xffffffffu : 0u;
    // (mask arithmetic, not `main_seg ? d2m : d2s`: the compiler turned that select of two by-reference captures into a
    // select of their ADDRESSES - both values went to scratch, and every slab paid a scratch load of the pointer and a flat
    // load of the value, vector-memory loads whose waits also sat out the activation loads in flight)
    nd2 = d2s + (int)((unsigned)(d2m - d2s) & sel);
#pragma unroll
    for (int i = 0; i < TM; ++i) na[i] = ab + (aoff2[i] + ((aoff[i] - aoff2[i]) & sel));
  };
  auto load_a = [&](int i) {
    const char* a = na[i];
    fa[i][0] = *reinterpret_cast<const u32x4*>(a);
    fa[i][1] = *reinterpret_cast<const u32x4*>(a + d1);
    fa[i][2] = *reinterpret_cast<const u32x4*>(a + nd2);
    fa[i][3] = *reinterpret_cast<const u32x4*>(a + nd2 + d1);
  };
  // (the LDS-direct loads always go out AFTER the register loads of the slab: vmcnt retires in order, so the compiler's
  // waits for the activation registers - it cannot see the asm loads - never include the weight fill)
  auto dma = [&](int i, int stage) {
    const char* src = nb + boff[i];
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Bs + (unsigned)(wave + 4 * i) * 1024u +
                                                        (unsigned)stage * (unsigned)STAGE);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
  };
  u32x4 qa[TM][2][2];                                   // [row group][k step][plane]: the lane's 8 k values as packed f16
  bool clamped = false;
  auto take_regs = [&]() {                              // fa -> qa: split float32 quads, or just take the operands
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (PIN) {
        qa[i][0][0] = fa[i][0]; qa[i][1][0] = fa[i][1]; qa[i][0][1] = fa[i][2]; qa[i][1][1] = fa[i][3];
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f32x4 lo = __builtin_bit_cast(f32x4, fa[i][2 * ks]), hi = __builtin_bit_cast(f32x4, fa[i][2 * ks + 1]);
          if (scaled_in) {       // (wave-uniform; single multiplies: v_pk_mul_f32 does not issue under another wave's MFMA)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              asm("v_mul_f32 %0, %1, %2" : "=v"(lo[c]) : "v"(lo[c]), "v"(insc));
              asm("v_mul_f32 %0, %1, %2" : "=v"(hi[c]) : "v"(hi[c]), "v"(insc));
            }
          }
          unsigned h1[4], h2[4];
          split_pair_f16(lo[0], lo[1], h1[0], h2[0], clamped);
          split_pair_f16(lo[2], lo[3], h1[1], h2[1], clamped);
          split_pair_f16(hi[0], hi[1], h1[2], h2[2], clamped);
          split_pair_f16(hi[2], hi[3], h1[3], h2[3], clamped);
          qa[i][ks][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
          qa[i][ks][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
        }
      }
    }
  };
  auto landed = [&]() {            // this wave's pieces are in LDS; then everybody's
    // (the builtin, not inline asm: the compiler's waitcnt bookkeeping then knows that none of ITS loads is pending
    // either - with an opaque asm wait it guarded the reuse of the activation registers with a vmcnt(0) placed right
    // behind the next slab's LDS-direct loads, i.e. every wave sat out the weight fill before its MFMAs)
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt/lgkmcnt untouched
    __syncthreads();
  };

  if (nslab > 0) {
    advance();
#pragma unroll
    for (int i = 0; i < TM; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < TN; ++i) dma(i, 0);
    take_regs();
    landed();
  }
  // MFMAs of slab s.  IL (slabs that have a successor): the next slab's loads are issued BETWEEN the product groups
  // instead of ahead of them - a wave issues in order, so a VMEM / LDS-direct instruction that waits for a queue slot
  // at the top of the body would hold back every MFMA behind it; in the gaps it waits under the wave's own MFMAs.
  auto compute = [&](int s, auto il_tag) {
    constexpr bool IL = decltype(il_tag)::value;
    const char* bst = Bs + (s & 1) * STAGE;
    const int nstage = (s + 1) & 1;
    if constexpr (IL) advance();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fb[TN][2];
      const int c = 2 * ks + fh;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 32 + fr;
          fb[j][pl] = *reinterpret_cast<const u32x4*>(bst + pl * (BN * 64) + (row * 4 + (c ^ ((row >> 2) & 3))) * 16);
        }
      constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};     // h1g1 h1g2 h2g1 (activation plane, weight plane)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j][PB[q]]),
                                                               __builtin_bit_cast(f16x8, qa[i][ks][PA[q]]), acc[i][j],
                                                               0, 0, 0);
        if constexpr (IL) {
          const int g = 3 * ks + q;                     // gap index 0..5: row groups first, then the weight pieces
          __builtin_amdgcn_sched_barrier(0);            // (pin the order: left alone, the scheduler clusters the loads
          if (g < TM) load_a(g);                        //  and guards them with vmcnt waits between the MFMAs)
          constexpr int G0 = TM, NG = 6 - TM, PER = (TN + NG - 1) / NG;
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int piece = (g - G0) * PER + e;
            if (g >= G0 && piece < TN) dma(piece, nstage);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // (float32 input: splitting the next slab's k-step-0 floats under k step 1's MFMAs - their registers are free by
    // then - measured 2-4 % SLOWER on layer 3: the wait for the floats lands inside the MFMA stream.  Other assignments
    // of the loads to the gaps (all activation loads first, weight pieces first, two pieces per gap) are within +-3 %.)
    if constexpr (IL) take_regs();
  };
  {
    for (int s = 0; s + 1 < nslab; ++s) {
      compute(s, std::true_type{});
      landed();
    }
    if (nslab > 0) compute(nslab - 1, std::false_type{});
  }

  if constexpr (!PIN) report_clamped(p.range_flag, clamped);      // float32 input split by the loader
  clamped = false;
  // epilogue: D[channel][agent]; agent = lane&31, channel = (r&3) + 8*(r>>2) + 4*(lane>>5).  The lane's 4 TN bias quads
  // are fetched as ONE batch of 16-byte loads (per-channel conditional loads cost one L2 round trip per quad).
  const bool vec = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
  const float acc_scale = *p.acc_scale / insc;      // (exact: both are powers of two)
  f32x4 bq[TN][4];
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[j][q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 32 + 4 * fh + 8 * q);
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[j][q][c] = p.bias ? p.bias[n0 + j * 32 + 4 * fh + 8 * q + c] : 0.f;
  }
  // Row-major float32 output (the GAT maps' Z, the last conv's map for the pooled head): the accumulator layout gives a
  // lane four 16-byte pieces of ONE agent's row per channel tile, i.e. a store instruction scatters 64 pieces over 32
  // rows.  Transposed through the (now idle) weight stages - one 64 BN-byte region per wave, 16-byte units XOR-swizzled
  // by the agent - every store instruction writes four agents' BN/2-channel runs (256 B each at BN = 128).
  // (a lambda since round 5: the fused second layer stores its rows the same way)
  // on_tile(j, vv): called with the 16 finished values of row group 0's channel tile j (the fused second layer forms its
  // activation planes from them: accumulators and biases are then consumed ONCE, tile by tile)
  auto store_rows_t = [&](f32x16 (&ac)[TM][TN], const float scl, const f32x4 (&bb)[TN][4], const int relu, float* const obase,
                          const int ldo, const long long otile, auto on_tile, u16* const obase16 = nullptr,
                          const int ld16 = 0) __attribute__((always_inline)) {
    constexpr int CP = BN / 2, UP = CP / 4;             // channels / 16-byte units per pass and agent
    constexpr int JP = CP / 32;                          // channel tiles per pass
    // every wave is done reading the weight stages (FUSE2: an LDS-only barrier - __syncthreads() carries vmcnt(0) and would
    // sit out the second layer's weight slabs that are in flight into stages 2 / 3)
    if constexpr (FUSE2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
    char* const wl = Bs + wave * (64 * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + 32 * (TM * wave + i);
#pragma unroll
      for (int ps = 0; ps < TN / JP; ++ps) {
#pragma unroll
        for (int jj = 0; jj < JP; ++jj) {
          const int j = ps * JP + jj;
          float vv[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = ac[i][j][4 * q + c] * scl + bb[j][q][c];
              if (relu) v[c] = magat_relu(v[c]);
              vv[4 * q + c] = v[c];
            }
            const int u = jj * 8 + 2 * q + fh;          // 16-byte unit of channels 32 jj + 8 q + 4 fh .. + 3
            *reinterpret_cast<f32x4*>(wl + fr * (CP * 4) + ((u ^ (fr & (UP - 1))) * 16)) = v;
          }
          if (i == 0) on_tile(j, vv);
        }
#pragma unroll
        for (int st = 0; st < 32 * UP / 64; ++st) {
          const int r = st * (64 / UP) + lane / UP, u = lane % UP;
          const f32x4 v = *reinterpret_cast<const f32x4*>(wl + r * (CP * 4) + ((u ^ (r & (UP - 1))) * 16));
          const int m = mb + r;
          if (m < p.M) {
            *reinterpret_cast<f32x4*>(obase + magat_row_off(m, ldo, otile) + ps * CP + 4 * u) = v;
            if (obase16)      // (the bf16-storage graph layer's input rows: no cast pass behind this launch)
              *reinterpret_cast<uint2*>(obase16 + (long long)m * ld16 + ps * CP + 4 * u) =
                  uint2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
          }
        }
      }
    }
  };
  if constexpr (FUSE2) {
    // ---- the second layer on the finished rows (see the template comment).  First its activation planes - from the very
    // values the store below writes: v = acc * acc_scale + bias (ReLU if the first layer has one), times the second layer's
    // input scale, split exactly as that layer's own float32 loader splits them.
    float insc2 = 1.f;
    if (p.in_scale2) insc2 = *p.in_scale2;
    if (insc2 == 0.f) insc2 = 1.f;
    u32x4 qb[8][2];                                       // [k step of the second layer's K = 128][plane]
    auto planes_of = [&](int j, const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned h1[4], h2[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int q = 2 * ks + e;
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[c] = vv[4 * q + c];
            if (insc2 != 1.f) asm("v_mul_f32 %0, %1, %2" : "=v"(v[c]) : "v"(v[c]), "v"(insc2));
          }
          split_pair_f16(v[0], v[1], h1[2 * e], h2[2 * e], clamped);
          split_pair_f16(v[2], v[3], h1[2 * e + 1], h2[2 * e + 1], clamped);
        }
        // accumulator layout -> operand k order: lane half h holds channels 16 ks + 8 e + 4 h + c in (e, c); an operand's lane
        // half h holds 16 ks + 8 h + i.  Half 0 keeps its e = 0 quad and takes half 1's e = 0 quad as i = 4..7; half 1 takes
        // half 0's e = 1 quad as i = 0..3 and keeps its own: the upper half of the e = 0 registers swaps with the lower half of
        // the e = 1 registers (v_permlane32_swap), dword by dword
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto r1 = __builtin_amdgcn_permlane32_swap(h1[d], h1[2 + d], false, false);
          h1[d] = r1[0]; h1[2 + d] = r1[1];
          const auto r2 = __builtin_amdgcn_permlane32_swap(h2[d], h2[2 + d], false, false);
          h2[d] = r2[0]; h2[2 + d] = r2[1];
        }
        qb[2 * j + ks][0] = u32x4{h1[0], h1[1], h1[2], h1[3]};
        qb[2 * j + ks][1] = u32x4{h2[0], h2[1], h2[2], h2[3]};
      }
    };
    // the second layer's weight slabs 0 / 1 -> stages 2 / 3 (nobody else's: no barrier needed), in flight under the row store
    const char* const w2b = reinterpret_cast<const char*>(p.wt2);
    unsigned boff2[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int id = wave + 4 * i;
      const int plane = id / (BN / 16), row = (id % (BN / 16)) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      boff2[i] = (unsigned)((plane * (BN * BN) + row * BN + c * 8) * 2);
    }
    auto dma2 = [&](int slab, int stage) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const char* src = w2b + slab * 64 + boff2[i];
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Bs + (unsigned)(wave + 4 * i) * 1024u +
                                                            (unsigned)stage * (unsigned)STAGE);
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
      }
    };
    dma2(0, 2);
    dma2(1, 3);
    // the first layer's rows (through the weight stages: a barrier in front, stores behind), the planes formed on the way
    store_rows_t(acc, acc_scale, bq, p.relu, static_cast<float*>(p.out) + (long long)pix * p.out_pix_stride + n0, p.ldc, p.out_tile,
                 planes_of);
    report_clamped(p.range_flag, clamped);
    clamped = false;
    // the second layer's K = 128: four 32-wide slabs - 0 / 1 are in stages 2 / 3 by now, 2 / 3 follow into stages 0 / 1 once
    // every wave is done with the row store's staging reads; ONE more barrier in front of the second pair
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    landed();                                             // slabs 0 / 1 landed for everyone; the staging reads are done
    dma2(2, 0);
    dma2(3, 1);
#pragma unroll
    for (int s2 = 0; s2 < BN / BK; ++s2) {
      if (s2 == 2) landed();
      const char* bst = Bs + ((s2 + 2) & 3) * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = 2 * ks + fh;
        constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
        u32x4 fb[TN][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int row = j * 32 + fr;
            fb[j][pl] = *reinterpret_cast<const u32x4*>(bst + pl * (BN * 64) + (row * 4 + (c ^ ((row >> 2) & 3))) * 16);
          }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j][PB[q]]),
                                                               __builtin_bit_cast(f16x8, qb[2 * s2 + ks][PA[q]]), acc[0][j], 0, 0, 0);
      }
    }
    const float acc_scale2 = *reinterpret_cast<const float*>(w2b + (size_t)2 * BN * BN * 2) / insc2;
    f32x4 bq2[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) bq2[j][q][c] = p.bias2 ? p.bias2[j * 32 + 4 * fh + 8 * q + c] : 0.f;
    store_rows_t(acc, acc_scale2, bq2, p.relu2, p.out2, p.ldc2, (long long)MAGAT_TILE_ROWS * p.ldc2, [](int, const float (&)[16]) {},
                 p.out2h, p.ldc2h);
    return;
  }
  if (TN >= 2 && p.out_gl == 0 && vec && p.tepi) {
    store_rows_t(acc, acc_scale, bq, p.relu,
                 static_cast<float*>(p.out) + (long long)pix * p.out_pix_stride + (p.out_nt ? ntile * p.out_nt : n0), p.ldc,
                 p.out_tile, [](int, const float (&)[16]) {});
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + 32 * (TM * wave + i) + fr;
    if (m >= p.M) continue;
    if (p.out_gl >= 2) {
      // f16 plane granules for the next f16x3 layer: the lane's quads 2 ks, 2 ks + 1 of a 32-channel tile ARE that
      // layer's k-step-ks operand (its weights are packed in this channel order), one 16-byte store per plane
      char* const ob = static_cast<char*>(p.out) + ((long long)pix * p.out_pix_stride + (long long)(m >> 7) * p.out_tile) * 4 +
                       fh * 2048 + (m & 127) * 16;
      const long long oplane = 256LL * p.Cout;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          unsigned h1[4], h2[4];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * ks + e;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[i][j][4 * q + c] * acc_scale + bq[j][q][c];
              if (p.relu) v[c] = magat_relu(v[c]);
            }
            split_pair_f16(v[0], v[1], h1[2 * e], h2[2 * e], clamped);
            split_pair_f16(v[2], v[3], h1[2 * e + 1], h2[2 * e + 1], clamped);
          }
          char* const o = ob + (long long)(((n0 >> 5) + j) * 2 + ks) * 4096;
          *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
          *reinterpret_cast<u32x4*>(o + oplane) = u32x4{h2[0], h2[1], h2[2], h2[3]};
        }
      report_clamped(p.range_flag, clamped);
      clamped = false;
      continue;
    }
    float* const orow = static_cast<float*>(p.out) + (long long)pix * p.out_pix_stride +
                        (p.out_gl ? (m >> 7) * p.out_tile + (m & 127) * 4 : magat_row_off(m, p.ldc, p.out_tile)) +
                        (p.out_nt ? ntile * p.out_nt - n0 : 0);
    const int nmul = p.out_gl ? 128 : 1;                // granule-major: channel quad n/4 is 128 agents x 4 floats away
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + j * 32 + 4 * fh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = acc[i][j][4 * q + c] * acc_scale + bq[j][q][c];
          if (p.relu) v[c] = magat_relu(v[c]);
        }
        if (vec) {
          *reinterpret_cast<f32x4*>(orow + (long long)n * nmul) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) orow[(long long)n * nmul + c] = v[c];
        }
      }
    }
  }
}


}  // namespace

// (f16x3 = the register-direct kernel; its 2x2 LDS-staged form - option CONV_DIRECT = 0 of rounds 1-4 - was removed in round 5)
int magat_conv_direct_enabled() { return 1; }

// in_fmt 5: like 4, but in/in2 arrive as the two f16 planes already (written by a producer with out_fmt 3): no split work.
// in_fmt 4: in/in2 float32 split on load into two f16 planes, wt = [2][Cout][Ktot] f16 planes of (weight * 2^e) followed
// by one float32 2^-e ("f16x3": three f16 MFMAs per product).
// in_fmt 1: in/in2/wt all bf16x3 planes; in_fmt 2: in/in2 float32 (split on load), wt bf16x3 planes; in_fmt 3: in/in2/wt
// ONE bf16 plane each (plain bf16 GEMM, fp32 accumulate).  out_fmt 0 f32, 1 bf16x3 planes, 2 one bf16 plane.
// Cout % 32 == 0, Cin % 32 == 0, C2 % 32 == 0, lda/lda2 % 8 == 0.
int magat_conv_gemm_bf16x6(const magat_conv_gemm_desc* d, hipStream_t st) {
  if (!d->in || !d->wt || !d->out) return MAGAT_ERR_NULL;
  if (d->M <= 0 || d->Cin <= 0 || d->Cout <= 0) return MAGAT_ERR_BAD_SHAPE;
  int BN = d->Cout % 128 == 0 ? 128 : (d->Cout % 64 == 0 ? 64 : 32);
  // few agents and one output pixel (the encoder head at the published batch sizes: 80 workgroups of 128 x 128 walking K = 1152
  // alone on a 256-CU chip): narrower column tiles until the launch fills the chip (option CONV_BNFILL = the workgroup count
  // to reach).  Head at 10 240 / 20 480 agents 46 -> 37 / 58 -> 48 us; narrower than the fill needs is SLOWER (51 200 agents at
  // 64 / 32 columns: 72 -> 87 / 104 us), because every column tile splits the same activations into planes again.  What is
  // left is the K walk itself: 36 slabs x ~1 000 cycles of barrier -> LDS read -> dependent MFMAs -> LDS write whatever the
  // memory does (counters of a 4-slab-prefetch experiment: 321 cycles of average L2 latency, no TLB misses, waves half
  // issuing, a third in barriers - profiles/r04d/head_stream_counters.txt; that kernel, a variant that splits the
  // activations once per workgroup through LDS, and padded tile strides all landed on the same 32-34 us).
  if (d->in_fmt == 4 && d->out_fmt == 0 && !d->out_ntile_stride && !d->out_gl && magat_conv_direct_enabled()) {
    const long long want = magat_opt(MAGAT_OPT_CONV_BNFILL);
    while (BN > 32 && (long long)((d->M + BM - 1) / BM) * d->Hout * d->Wout * (d->Cout / BN) < want) BN >>= 1;
  }
  if ((d->Cout % BN) || (d->Cin % BK) || (d->C2 % BK) || (d->lda % 8) || (d->C2 > 0 && (d->lda2 % 8)) || d->pool)
    return MAGAT_ERR_UNSUPPORTED;
  // second layer in the epilogue (magat_hip.h wt2): whole 128-wide rows per workgroup tile, the transposed row store
  const bool fuse2 = d->wt2 != nullptr;
  if (fuse2 && !(d->in_fmt == 4 && d->out_fmt == 0 && d->in_gl <= 1 && d->out_gl == 0 && !d->out_ntile_stride && BN == 128 &&
                 d->Cout == 128 && d->Cout2 == 128 && d->Hout * d->Wout == 1 && d->out2 && (d->ldc & 3) == 0 && (d->ldc2 & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(d->out) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->out2) & 15) == 0 &&
                 magat_conv_direct_enabled()))
    return MAGAT_ERR_UNSUPPORTED;
  if ((d->in_fmt != 3 && d->in_fmt != 4) || d->out_fmt < 0 || d->out_fmt > 3) return MAGAT_ERR_UNSUPPORTED;
  if (d->in_fmt == 4 && d->out_fmt != 0) return MAGAT_ERR_UNSUPPORTED;                       // f16x3: float32 rows / granules out
  if (d->in_fmt == 3 && d->out_fmt != 0 && d->out_fmt != 2) return MAGAT_ERR_UNSUPPORTED;   // plain bf16: float32 or bf16 out
  if (d->wt_pix_stride || d->ldw) return MAGAT_ERR_UNSUPPORTED;     // float32 kernel only
  if (d->in_fmt >= 4 && d->out_fmt != 0 && d->out_fmt != 3) return MAGAT_ERR_UNSUPPORTED;
  if (d->out_fmt == 3 && d->in_fmt < 4) return MAGAT_ERR_UNSUPPORTED;
  SplitParams p;
  p.in = static_cast<const u16*>(static_cast<const void*>(d->in));
  p.in2 = static_cast<const u16*>(static_cast<const void*>(d->in2));
  p.wt = static_cast<const u16*>(static_cast<const void*>(d->wt));
  p.bias = d->bias;
  p.out = d->out;
  p.in_pix_stride = d->in_pix_stride; p.in2_pix_stride = d->in2_pix_stride; p.out_pix_stride = d->out_pix_stride;
  p.in_plane = d->in_plane_stride; p.in2_plane = d->in2_plane_stride; p.out_plane = d->out_plane_stride;
  p.in_tile = d->in_tile_stride ? d->in_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda;
  p.in2_tile = d->in2_tile_stride ? d->in2_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda2;
  p.out_tile = d->out_tile_stride ? d->out_tile_stride : (long long)MAGAT_TILE_ROWS * d->ldc;
  if ((p.in_tile & 7) || (p.in2_tile & 7)) return MAGAT_ERR_BAD_SHAPE;
  p.M = d->M; p.Cin = d->Cin; p.lda = d->lda; p.Hin = d->Hin; p.Win = d->Win; p.kH = d->kH; p.kW = d->kW;
  p.stride = d->stride; p.pad = d->pad; p.Hout = d->Hout; p.Wout = d->Wout;
  p.C2 = d->C2; p.lda2 = d->lda2; p.W2 = d->W2; p.stride2 = d->stride2;
  p.Cout = d->Cout; p.Ktot = d->kH * d->kW * d->Cin + d->C2; p.ldc = d->ldc; p.relu = d->relu;
  p.wt_plane = (long long)p.Cout * p.Ktot;
  p.npix = d->Hout * d->Wout; p.tag = d->tag; p.out_split = d->out_fmt;
  p.in_gl = d->in_gl; p.out_gl = d->out_gl;
  p.out_nt = d->out_ntile_stride;
  p.range_flag = reinterpret_cast<int*>(d->range_flag);
  p.wt2 = static_cast<const u16*>(d->wt2); p.bias2 = d->bias2; p.out2 = d->out2; p.in_scale2 = d->in_scale2;
  p.ldc2 = d->ldc2; p.relu2 = d->relu2;
  p.out2h = static_cast<u16*>(d->out2_bf16); p.ldc2h = d->ldc2_bf16;
  if (p.out2h && ((p.ldc2h & 3) || (reinterpret_cast<uintptr_t>(p.out2h) & 7))) return MAGAT_ERR_UNSUPPORTED;
  if (p.out_nt && !(d->in_fmt == 4 && d->out_fmt == 0 && d->out_gl == 0 && BN == 128 && magat_conv_direct_enabled()))
    return MAGAT_ERR_UNSUPPORTED;
  p.korder = 1;      // (direct kernel K walk: channel slab outer, taps inner; the tap-major order of round 1 re-fetched every chunk 3.6x)
  p.tepi = 1;        // (row-major float32 output transposed through LDS)
  p.Mt = (p.M + BM - 1) / BM;
  p.ntn = p.Cout / BN;
  if (magat_row_off(p.M, p.lda, p.in_tile) * 4 >= 0xffffffffLL ||
      (p.C2 > 0 && magat_row_off(p.M, p.lda2, p.in2_tile) * 4 >= 0xffffffffLL) ||
      (long long)p.Cout * p.Ktot * 2 >= 0xffffffffLL)
    return MAGAT_ERR_UNSUPPORTED;
  const long long groups = (p.Mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD;
  const long long grid = groups * MAGAT_NUM_XCD * p.npix * p.ntn;
  if (grid <= 0 || grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const int pid = magat_prof_begin(p.tag, st);
  // f16x3 (in_fmt 4): wt = [2][Cout][Ktot] f16 bits followed by one float32 = 1 / weight scale (read by the kernel)
  p.acc_scale = d->acc_scale ? d->acc_scale
                             : reinterpret_cast<const float*>(reinterpret_cast<const char*>(d->wt) + (size_t)2 * p.Cout * p.Ktot * sizeof(u16));
  p.in_scale = d->in_scale;
  // (the activation scale lives in the direct kernel's float32 loader only)
  if (d->in_scale && !(d->in_fmt == 4 && d->out_fmt == 0 && d->in_gl <= 1 && magat_conv_direct_enabled())) return MAGAT_ERR_UNSUPPORTED;
  // (round 5: the bf16x6 flavour - in_fmt 1 / 2, three bf16 planes, six products -, the pre-split f16 input of in_fmt 5 and
  //  the LDS-staged f16x3 form are gone with their users: what is left of the LDS-staged kernel is the plain-bf16 GEMM of the
  //  bf16-storage graph layer, in_fmt 3)
#define MAGAT_SPLIT_LAUNCH(BNV, WM, WN)                                                                              \
  hipLaunchKernelGGL((conv_gemm_bf16x6_kernel<false, BNV, WM, WN, 1>), dim3((unsigned)grid), dim3(256), 0, st, p)
  const int direct = magat_conv_direct_enabled();
  if ((d->in_gl || d->out_gl) && !(d->in_fmt == 4 && d->out_fmt == 0 && direct)) return MAGAT_ERR_UNSUPPORTED;
  if (d->in_gl < 0 || d->in_gl > 2 || d->out_gl < 0 || d->out_gl > 2) return MAGAT_ERR_UNSUPPORTED;
  if (d->in_fmt == 4 && d->out_fmt == 0 && direct) {
    const int tm2 = magat_opt(MAGAT_OPT_CONV_TM);     // 1: one 32-agent row group per wave everywhere
    const bool two = !fuse2 && tm2 >= 2 && (long long)(p.Mt / 2) * p.npix * p.ntn >= 2048;   // enough 256-agent tiles to fill the chip
    const long long mt = two ? (p.Mt + 1) / 2 : p.Mt;
    const long long g2 = (mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD * MAGAT_NUM_XCD * p.npix * p.ntn;
    p.Mt = (int)mt;
#define MAGAT_DIRECT_LAUNCH(BNV)                                                                                     \
  do {                                                                                                              \
    if (two && pin)                                                                                                 \
      hipLaunchKernelGGL((conv_gemm_f16x3_direct_kernel<BNV, 2, 1>), dim3((unsigned)g2), dim3(256), 0, st, p);      \
    else if (two)                                                                                                   \
      hipLaunchKernelGGL((conv_gemm_f16x3_direct_kernel<BNV, 2, 0>), dim3((unsigned)g2), dim3(256), 0, st, p);      \
    else if (pin)                                                                                                   \
      hipLaunchKernelGGL((conv_gemm_f16x3_direct_kernel<BNV, 1, 1>), dim3((unsigned)g2), dim3(256), 0, st, p);      \
    else                                                                                                            \
      hipLaunchKernelGGL((conv_gemm_f16x3_direct_kernel<BNV, 1, 0>), dim3((unsigned)g2), dim3(256), 0, st, p);      \
  } while (0)
    const bool pin = d->in_gl == 2;
    if (fuse2) {
      magat_form_note(MAGAT_FORM_HEAD_COMPRESS);
      hipLaunchKernelGGL((conv_gemm_f16x3_direct_kernel<128, 1, 0, true>), dim3((unsigned)g2), dim3(256), 0, st, p);
    } else if (BN == 128) MAGAT_DIRECT_LAUNCH(128);
    else if (BN == 64) MAGAT_DIRECT_LAUNCH(64);
    else MAGAT_DIRECT_LAUNCH(32);
#undef MAGAT_DIRECT_LAUNCH
  } else if (BN == 128) MAGAT_SPLIT_LAUNCH(128, 2, 2);
  else if (BN == 64) MAGAT_SPLIT_LAUNCH(64, 2, 2);
  else MAGAT_SPLIT_LAUNCH(32, 4, 1);
#undef MAGAT_SPLIT_LAUNCH
  magat_prof_end(pid, st);
  return magat_check_launch();
}
