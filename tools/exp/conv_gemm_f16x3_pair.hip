// f16x3 3x3 / stride-1 conv GEMM, TWO adjacent output pixels per workgroup ("pair" form of the direct kernel in
// conv_gemm_bf16x6.hip; same arithmetic, same plane-granule operands, same LDS-direct weight fills).
//
// Skeleton experiments on the one-pixel kernel showed that its time is set by the activation-operand
// stream from L2 into registers: every input value is fetched once per tap, 7.1 times on a 6x6 map (7.2 GB per layer3.conv2
// launch, 0.57 ms by itself).  Here a workgroup of 8 waves owns 256 agents x the output pixels (y, x), (y, x+1): an input
// pixel of column x-1 .. x+2 is loaded ONCE and multiplied with the weights of the tap it is for EACH of the two outputs
// (columns x and x+1 serve both) - 12 operand loads per channel slab instead of 18, at unchanged weight traffic per MFMA
// (256 agents share every weight slab, as with the one-pixel kernel's 256-agent tiles).  Wave = 32 agents x 128 channels
// x 2 pixels (the same 128 accumulator registers as 64 agents x 1 pixel).
//
// K walk ("steps"): channel slab outer, then tap row, then input column c = 0..3; a step loads one 32-channel operand slab
// per wave and fills one or two weight slabs (LDS slots 0 / 1 of the stage):
//   slot 0 always feeds pixel 0 (tap (ty, c), c = 0..2), slot 1 always pixel 1 (tap (ty, c-1), c = 1..3): static
//   accumulators, the inactive slot of a step is skipped by a uniform branch
// then the residual 1x1 segment (in2), one step per pixel.  Everything else as in the direct kernel: next step's loads
// between the MFMA groups, ONE barrier per step, builtin s_waitcnt.
#include <cstdlib>
#include <type_traits>

#include "magat_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct PairParams {
  const char* in;
  const char* in2;
  const char* wt;            // f16 planes [2][Cout][Ktot] (K-permuted), then one float 2^-e
  const float* bias;
  void* out;
  long long in_pix, in2_pix, out_pix;      // floats
  long long in_tile, in2_tile, out_tile;   // floats
  long long wt_plane;                      // halves: Cout * Ktot
  int M, Mt;
  int Cin, Hin, Win, Hout, Wout, C2, W2, Cout, Ktot, ldc, relu;
  int ntn, npair, out_gl, tag;
  const float* acc_scale;
};

__device__ __forceinline__ void split2(float x, float y, unsigned& p1, unsigned& p2) {
  x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  const f16x2 r = __builtin_convertvector(f32x2{x - (float)h[0], y - (float)h[1]}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  p2 = __builtin_bit_cast(unsigned, r);
}

constexpr int BK = 32;

// <BN, TM, NW>: channel tile, 32-agent row groups per wave, waves per workgroup (32 TM NW agents per workgroup).
//   <128, 1, 8>: 256 agents, wave = 32 agents x 128 channels x 2 pixels (opt-in, one 8-wave workgroup per CU)
//   < 64, 2, 4>: 256 agents, wave = 64 agents x  64 channels x 2 pixels: the same 128 accumulator registers in the
//                proven shape of two free-running 4-wave workgroups per CU - the default of the 64-channel 3x3 layers
template <int BN, int TM, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void conv_gemm_f16x3_pair_kernel(const PairParams p) {
  constexpr int TN = BN / 32;
  constexpr int SLAB = 2 * BN * 64;                      // bytes of one weight slab in LDS: two planes of BN rows x 64 B
  constexpr int NPW = BN / 8 / NW;                       // weight pieces (1 KB) per wave and slab
  static_assert(NPW >= 1 && NPW * NW * 8 == BN, "piece split");
  extern __shared__ __attribute__((aligned(1024))) char Bs[];   // [stage 2][slot 2][SLAB]

  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per_m = p.npair * p.ntn;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / per_m);
  if (mtile >= p.Mt) return;
  const int rem = slot % per_m;
  const int pair = rem / p.ntn, ntile = rem % p.ntn;
  const int hw = p.Wout >> 1;
  const int oy = pair / hw, x0 = 2 * (pair % hw);
  const int m0 = mtile * (32 * TM * NW), n0 = ntile * BN;
  const int ty0 = oy - 1 < 0 ? 1 : 0, ty1 = min(3, p.Hin - (oy - 1));
  const int c0 = x0 - 1 < 0 ? 1 : 0, c1 = min(4, p.Win - (x0 - 1));
  const int spt = p.Cin / BK, spt2 = p.C2 / BK;
  const int nsteps = spt * (ty1 - ty0) * (c1 - c0) + 2 * spt2;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[2][TM][TN];         // [pixel][row group][channel tile]
#pragma unroll
  for (int px = 0; px < 2; ++px)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[px][i][j][r] = 0.f;

  // plane-granule operand address of this lane (magat_hip.h in_gl = 2): + plane * 256 C + k step * 4096 + k0 * 256
  unsigned aoff[TM], aoff2[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = min(m0 + 32 * (TM * wave + i) + fr, p.M - 1);
    aoff[i] = (unsigned)((mrow >> 7) * p.in_tile * 4 + fh * 2048 + (mrow & 127) * 16);
    aoff2[i] = (unsigned)((mrow >> 7) * p.in2_tile * 4 + fh * 2048 + (mrow & 127) * 16);
  }
  const int d2m = 256 * p.Cin, d2s = 256 * p.C2;

  // weight slab pieces (1 KB = 16 rows x 64 B of one plane) this wave copies: piece id = wave + NW i
  long long boff[NPW];
  unsigned bm0[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int id = wave + NW * i;
    const int plane = id / (BN / 16), row = (id % (BN / 16)) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    boff[i] = ((long long)plane * p.wt_plane + (long long)(n0 + row) * p.Ktot + c * 8) * 2;
    bm0[i] = (unsigned)(uintptr_t)Bs + (unsigned)id * 1024u;
  }

  // ---- step cursor ---------------------------------------------------------------------------------------------------
  int cur_seg = spt * (ty1 - ty0) * (c1 - c0) > 0 ? 0 : 1, cur_ks = 0, cur_ty = ty0, cur_c = c0;
  const char* na[TM];        // operand addresses of the next step
  int nd2, nmode, nbk0, nbk1;   // its plane distance, active slots (0: slot 0 = pixel 0 only; 1: both; 2: slot 1 = pixel 1 only), K offsets
  auto advance = [&]() {
    const int k0 = cur_ks * BK;
    if (cur_seg == 0) {
      const char* ab = p.in + ((long long)((oy - 1 + cur_ty) * p.Win + (x0 - 1 + cur_c)) * p.in_pix) * 4 + (long long)k0 * 256;
#pragma unroll
      for (int i = 0; i < TM; ++i) na[i] = ab + aoff[i];
      nd2 = d2m;
      nmode = cur_c == 0 ? 0 : (cur_c == 3 ? 2 : 1);
      nbk0 = (cur_ty * 3 + cur_c) * p.Cin + k0;
      nbk1 = (cur_ty * 3 + (cur_c - 1)) * p.Cin + k0;
      if (++cur_c == c1) {
        cur_c = c0;
        if (++cur_ty == ty1) {
          cur_ty = ty0;
          if (++cur_ks == spt) { cur_ks = 0; cur_seg = 1; }
        }
      }
    } else {
      const int px = cur_seg - 1;
      const char* ab = p.in2 + ((long long)(oy * p.W2 + x0 + px) * p.in2_pix) * 4 + (long long)k0 * 256;
#pragma unroll
      for (int i = 0; i < TM; ++i) na[i] = ab + aoff2[i];
      nd2 = d2s;
      nmode = px == 0 ? 0 : 2;
      nbk0 = 9 * p.Cin + k0;
      nbk1 = nbk0;
      if (++cur_ks == spt2) { cur_ks = 0; ++cur_seg; }
    }
  };
  u32x4 fa[TM][4];              // the next step's operands [row group][plane * 2 + k step] as loaded
  auto load_a = [&](int i) {
    fa[i][0] = *reinterpret_cast<const u32x4*>(na[i]);
    fa[i][1] = *reinterpret_cast<const u32x4*>(na[i] + 4096);
    fa[i][2] = *reinterpret_cast<const u32x4*>(na[i] + nd2);
    fa[i][3] = *reinterpret_cast<const u32x4*>(na[i] + nd2 + 4096);
  };
  // weight piece e (0 .. NPW-1: slot 0; NPW .. 2 NPW-1: slot 1) of the next step into stage `stage`
  auto dma = [&](int e, int stage) {
    const int sl = e / NPW, i = e % NPW;
    const char* src = p.wt + (long long)(sl ? nbk1 : nbk0) * 2 + boff[i];
    const unsigned m0v =
        __builtin_amdgcn_readfirstlane(bm0[i] + (unsigned)(stage * 2 + sl) * (unsigned)SLAB);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
  };
  u32x4 qa[TM][2][2];           // [row group][k step][plane] of the current step
  auto take_regs = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      qa[i][0][0] = fa[i][0]; qa[i][1][0] = fa[i][1]; qa[i][0][1] = fa[i][2]; qa[i][1][1] = fa[i][3];
    }
  };
  auto landed = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    __syncthreads();
  };

  int cmode = 0;
  if (nsteps > 0) {
    advance();
    cmode = nmode;
#pragma unroll
    for (int i = 0; i < TM; ++i) load_a(i);
#pragma unroll
    for (int e = 0; e < NPW; ++e) {
      if (nmode != 2) dma(e, 0);
      if (nmode != 0) dma(NPW + e, 0);
    }
    take_regs();
    landed();
  }
  constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};     // h1g1 h1g2 h2g1 (activation plane, weight plane)
  // MFMAs of step s, the next step's loads between the product groups
  auto compute = [&](int s, auto il_tag) {
    constexpr bool IL = decltype(il_tag)::value;
    const char* bst = Bs + (s & 1) * (2 * SLAB);
    const int nstage = (s + 1) & 1;
    const bool act0 = cmode != 2, act1 = cmode != 0;
    if constexpr (IL) advance();
    int g = 0;                   // gap counter of the step (uniform)
    auto gap = [&]() {
      if constexpr (IL) {
        __builtin_amdgcn_sched_barrier(0);
        // gaps 0 .. TM-1: operand loads; then the slot-0 pieces, then the slot-1 pieces.  The gap counter is a run-time
        // value (slot blocks are skipped by branches), the array indices must not be: indexed by g, the operand buffers
        // and piece offsets were demoted to scratch memory (2.5x slower).
        if constexpr (TM == 1) {
          if (g == 0) load_a(0);
          else if (g <= 2) { if (nmode != 2) dma(g - 1, nstage); }
          else if (g <= 4) { if (nmode != 0) dma(g - 1, nstage); }
        } else {
#pragma unroll
          for (int k = 0; k < TM; ++k)
            if (g == k) load_a(k);
#pragma unroll
          for (int k = 0; k < 2 * NPW; ++k)
            if (g == TM + k && (k < NPW ? nmode != 2 : nmode != 0)) dma(k, nstage);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ++g;
    };
    auto block = [&](int ks, auto px_tag) {
      constexpr int PX = decltype(px_tag)::value;
      const int c = 2 * ks + fh;
      u32x4 fb[TN][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 32 + fr;
          fb[j][pl] = *reinterpret_cast<const u32x4*>(bst + PX * SLAB + pl * (BN * 64) +
                                                      (row * 4 + (c ^ ((row >> 2) & 3))) * 16);
        }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[PX][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j][PB[q]]),
                                                                   __builtin_bit_cast(f16x8, qa[i][ks][PA[q]]),
                                                                   acc[PX][i][j], 0, 0, 0);
        gap();
      }
    };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (act0) block(ks, std::integral_constant<int, 0>{});
      if (act1) block(ks, std::integral_constant<int, 1>{});
    }
    if constexpr (IL) {
      take_regs();
      cmode = nmode;
    }
  };
  for (int s = 0; s + 1 < nsteps; ++s) {
    compute(s, std::true_type{});
    landed();
  }
  if (nsteps > 0) compute(nsteps - 1, std::false_type{});

  // ---- epilogue (per pixel) ------------------------------------------------------------------------------------------------
  const float acc_scale = *p.acc_scale;
  auto bias_of = [&](int j, f32x4 (&bq)[4]) {      // the lane's four bias quads of channel tile j (one batch of loads)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (p.bias) bq[q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 32 + 4 * fh + 8 * q);
      else bq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (p.out_gl == 0) {
    if constexpr (BN == 128 && TM == 1) {
      // row-major float32 (the last conv's map for the pooled head): transposed through the idle weight stages, 8 KB per
      // wave, 16-byte units XOR-swizzled by the agent - every store instruction writes four agents' 64-channel runs
      __syncthreads();
      char* const wl = Bs + wave * 8192;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int pix = oy * p.Wout + x0 + px;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = ps * 2 + jj;
            f32x4 bq[4];
            bias_of(j, bq);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                v[c] = acc[px][0][j][4 * q + c] * acc_scale + bq[q][c];
                if (p.relu) v[c] = fmaxf(v[c], 0.f);
              }
              const int u = jj * 8 + 2 * q + fh;
              *reinterpret_cast<f32x4*>(wl + fr * 256 + ((u ^ (fr & 15)) * 16)) = v;
            }
          }
#pragma unroll
          for (int st = 0; st < 8; ++st) {
            const int r = st * 4 + (lane >> 4), u = lane & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(wl + r * 256 + ((u ^ (r & 15)) * 16));
            const int mm = m0 + 32 * wave + r;
            if (mm < p.M)
              *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + (long long)pix * p.out_pix +
                                        magat_row_off(mm, p.ldc, p.out_tile) + n0 + ps * 64 + 4 * u) = v;
          }
        }
      }
    }
    return;                     // (the host sends row-major outputs of other shapes to the one-pixel kernel)
  }
#pragma unroll
  for (int px = 0; px < 2; ++px) {
    const int pix = oy * p.Wout + x0 + px;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + 32 * (TM * wave + i) + fr;
      if (m >= p.M) continue;
      // f16 plane granules for the next f16x3 layer (out_gl = 2)
      char* const ob = static_cast<char*>(p.out) + ((long long)pix * p.out_pix + (long long)(m >> 7) * p.out_tile) * 4 +
                       fh * 2048 + (m & 127) * 16;
      const long long oplane = 256LL * p.Cout;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 bq[4];
        bias_of(j, bq);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          unsigned h1[4], h2[4];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * ks + e;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[px][i][j][4 * q + c] * acc_scale + bq[q][c];
              if (p.relu) v[c] = fmaxf(v[c], 0.f);
            }
            split2(v[0], v[1], h1[2 * e], h2[2 * e]);
            split2(v[2], v[3], h1[2 * e + 1], h2[2 * e + 1]);
          }
          char* const o = ob + (long long)(((n0 >> 5) + j) * 2 + ks) * 4096;
          *reinterpret_cast<u32x4*>(o) = u32x4{h1[0], h1[1], h1[2], h1[3]};
          *reinterpret_cast<u32x4*>(o + oplane) = u32x4{h2[0], h2[1], h2[2], h2[3]};
        }
      }
    }
  }
}

}  // namespace

// Takes: in_fmt 4, out_fmt 0, in_gl 2, out_gl 0 or 2, 3x3 / stride 1 / pad 1, Hout = Hin, Wout = Win even, Cout % 128 == 0,
// Cin % 32 == 0, C2 % 32 == 0 with stride2 == 1 and W2 == Wout, no pooling.  Returns MAGAT_ERR_UNSUPPORTED otherwise (the
// caller then uses the one-pixel direct kernel).
int magat_conv_gemm_f16x3_pair(const magat_conv_gemm_desc* d, hipStream_t st) {
  if (d->in_fmt != 4 || d->out_fmt != 0 || d->in_gl != 2 || (d->out_gl != 0 && d->out_gl != 2)) return MAGAT_ERR_UNSUPPORTED;
  if (d->kH != 3 || d->kW != 3 || d->stride != 1 || d->pad != 1 || d->Hout != d->Hin || d->Wout != d->Win || (d->Wout & 1) ||
      d->pool)
    return MAGAT_ERR_UNSUPPORTED;
  const int bn = d->Cout % 128 == 0 ? 128 : 64;
  if ((d->Cout % 64) || (d->Cin % 32) || (d->C2 % 32) || (d->C2 > 0 && (d->stride2 != 1 || d->W2 != d->Wout || !d->in2)))
    return MAGAT_ERR_UNSUPPORTED;
  if (bn == 64 && d->out_gl != 2) return MAGAT_ERR_UNSUPPORTED;
  if (d->out_gl == 0 && ((d->ldc & 3) || (reinterpret_cast<uintptr_t>(d->out) & 15))) return MAGAT_ERR_UNSUPPORTED;
  if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15)) return MAGAT_ERR_UNSUPPORTED;
  PairParams p;
  p.in = reinterpret_cast<const char*>(d->in);
  p.in2 = reinterpret_cast<const char*>(d->in2);
  p.wt = reinterpret_cast<const char*>(d->wt);
  p.bias = d->bias;
  p.out = d->out;
  p.in_pix = d->in_pix_stride; p.in2_pix = d->in2_pix_stride; p.out_pix = d->out_pix_stride;
  p.in_tile = d->in_tile_stride ? d->in_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda;
  p.in2_tile = d->in2_tile_stride ? d->in2_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda2;
  p.out_tile = d->out_tile_stride ? d->out_tile_stride : (long long)MAGAT_TILE_ROWS * d->ldc;
  p.M = d->M; p.Mt = (d->M + 255) / 256;
  p.Cin = d->Cin; p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout;
  p.C2 = d->C2; p.W2 = d->W2; p.Cout = d->Cout; p.Ktot = 9 * d->Cin + d->C2; p.ldc = d->ldc; p.relu = d->relu;
  p.wt_plane = (long long)p.Cout * p.Ktot;
  p.ntn = p.Cout / bn; p.npair = d->Hout * (d->Wout / 2); p.out_gl = d->out_gl; p.tag = d->tag;
  p.acc_scale = reinterpret_cast<const float*>(p.wt + (size_t)2 * p.Cout * p.Ktot * 2);
  const long long t128 = (d->M + 127) / 128;
  if (t128 * p.in_tile * 4 >= 0xffffffffLL || (p.C2 > 0 && t128 * p.in2_tile * 4 >= 0xffffffffLL))
    return MAGAT_ERR_UNSUPPORTED;
  const long long groups = (p.Mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD;
  const long long grid = groups * MAGAT_NUM_XCD * p.npair * p.ntn;
  if (grid <= 0 || grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const int pid_dummy = 0; (void)pid_dummy;
  if (bn == 128) {
    constexpr size_t lds = 2 * 2 * (size_t)(2 * 128 * 64);          // 64 KB
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_f16x3_pair_kernel<128, 1, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MAGAT_ERR_LAUNCH;
      attr_set = true;
    }
    const int pid = magat_prof_begin(p.tag, st);
    hipLaunchKernelGGL((conv_gemm_f16x3_pair_kernel<128, 1, 8>), dim3((unsigned)grid), dim3(512), lds, st, p);
    magat_prof_end(pid, st);
    return magat_check_launch();
  }
  constexpr size_t lds64 = 2 * 2 * (size_t)(2 * 64 * 64);            // 32 KB
  const int pid = magat_prof_begin(p.tag, st);
  hipLaunchKernelGGL((conv_gemm_f16x3_pair_kernel<64, 2, 4>), dim3((unsigned)grid), dim3(256), lds64, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
