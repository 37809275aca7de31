// f16x3 conv GEMM, two 256-agent HALVES per workgroup running half a slab apart ("duo" form of the direct kernel in
// conv_gemm_bf16x6.hip; same arithmetic, plane-granule operands, LDS-direct weight fills, K walk and epilogues).
//
// The one-pixel direct kernel runs two independent 4-wave workgroups per CU; the two waves that share a SIMD drift into
// phase (both in their MFMA burst, then both loading), the MFMA pipe is busy 54 % of the time, and every workgroup fetches
// its own copy of each weight slab.  Here ONE 8-wave workgroup owns 512 agents x one output pixel: waves 0-3 (half A) and
// 4-7 (half B) are the two former workgroups, but
//   * they share every weight slab (L2 -> LDS weight traffic and LDS-direct issue per MFMA halved again), and
//   * they are held half a slab period apart by two workgroup barriers per slab: while half A issues its 48 MFMAs per
//     wave from registers + LDS, half B does nothing but issue the loads of its slab after next (activation operands
//     into the buffer it just consumed, its share of the weight slab after next into the free LDS stage) - and vice
//     versa.  A SIMD always holds one wave of each half: one of them is in its MFMA segment at any time.
// Two operand buffers per wave (X / Y, even / odd slabs) are loaded a full period ahead; three LDS weight stages.
#include <cstdlib>
#include <type_traits>

#include "magat_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DuoParams {
  const char* in;
  const char* in2;
  const char* wt;            // f16 planes [2][Cout][Ktot] (K-permuted), then one float 2^-e
  const float* bias;
  void* out;
  long long in_pix, in2_pix, out_pix;      // floats
  long long in_tile, in2_tile, out_tile;   // floats
  long long wt_plane;                      // halves: Cout * Ktot
  int M, Mt;                               // Mt: 512-agent tiles
  int Cin, Hin, Win, kH, kW, stride, pad, Hout, Wout;
  int C2, W2, stride2, Cout, Ktot, ldc, relu;
  int ntn, npix, out_gl, tag;
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

constexpr int BN = 128, TN = 4, BK = 32, TM = 2;
constexpr int STAGE = 2 * BN * 64;                       // bytes of one weight slab in LDS
constexpr int NPW = BN / 64;                             // weight pieces (1 KB) per wave and slab: 16 pieces / 8 waves

__global__ __launch_bounds__(512, 1) void conv_gemm_f16x3_duo_kernel(const DuoParams p) {
  extern __shared__ __attribute__((aligned(1024))) char Bs[];   // 3 stages (64 KB allocated: the epilogue transposes in it)

  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per_m = p.npix * p.ntn;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / per_m);
  if (mtile >= p.Mt) return;
  const int rem = slot % per_m;
  const int pix = rem / p.ntn, ntile = rem % p.ntn;
  const int n0 = ntile * BN;
  const int oy = pix / p.Wout, ox = pix % p.Wout;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  const int ty0 = iy0 < 0 ? -iy0 : 0, tx0 = ix0 < 0 ? -ix0 : 0;
  const int ty1 = min(p.kH, p.Hin - iy0), tx1 = min(p.kW, p.Win - ix0);
  const int ntaps = (ty1 - ty0) * (tx1 - tx0);
  const int spt = p.Cin / BK, spt2 = p.C2 / BK;
  const int nslab = ntaps * spt + spt2;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = wave >> 2;
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = mtile * 512 + 64 * wave;                // first agent of this wave (two 32-agent row groups)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  unsigned aoff[TM], aoff2[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = min(m0 + 32 * i + fr, p.M - 1);
    aoff[i] = (unsigned)((mrow >> 7) * p.in_tile * 4 + fh * 2048 + (mrow & 127) * 16);
    aoff2[i] = (unsigned)((mrow >> 7) * p.in2_tile * 4 + fh * 2048 + (mrow & 127) * 16);
  }
  int d2m = 256 * p.Cin, d2s = 256 * p.C2;
  asm volatile("" : "+s"(d2m), "+s"(d2s));     // (pinned in SGPRs: rematerialised, they came back as a flat load + vmcnt(0) per slab)

  long long boff[NPW];
  unsigned bm0[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int id = wave + 8 * i;
    const int plane = id / (BN / 16), row = (id % (BN / 16)) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    boff[i] = ((long long)plane * p.wt_plane + (long long)(n0 + row) * p.Ktot + c * 8) * 2;
    bm0[i] = (unsigned)(uintptr_t)Bs + (unsigned)id * 1024u;
  }

  // ---- slab cursor: channel slab outer, taps inner, then the residual segment (as the direct kernel with korder = 1) ----
  int cur_ty = ty0, cur_tx = tx0, cur_ks = 0;
  bool cur_main = ntaps > 0;
  const char* const seg2_base = p.in2 + (long long)(oy * p.stride2 * p.W2 + ox * p.stride2) * p.in2_pix * 4;
  const char* na[TM];
  const char* nb;
  int nd2;
  auto advance = [&]() {
    const bool main_seg = cur_main;
    const int k0 = cur_ks * BK;
    const char* ab;
    int bk;
    if (main_seg) {
      ab = p.in + (long long)((iy0 + cur_ty) * p.Win + (ix0 + cur_tx)) * p.in_pix * 4 + (long long)k0 * 256;
      bk = (cur_ty * p.kW + cur_tx) * p.Cin + k0;
      if (++cur_tx == tx1) {
        cur_tx = tx0;
        if (++cur_ty == ty1) {
          cur_ty = ty0;
          if (++cur_ks == spt) { cur_ks = 0; cur_main = false; }
        }
      }
    } else {
      ab = seg2_base + (long long)k0 * 256;
      bk = p.kH * p.kW * p.Cin + k0;
      ++cur_ks;
    }
    nb = p.wt + (long long)bk * 2;
    nd2 = d2s + ((d2m - d2s) & (main_seg ? -1 : 0));
    const unsigned sel = main_seg ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < TM; ++i) na[i] = ab + (aoff2[i] + ((aoff[i] - aoff2[i]) & sel));
  };
  u32x4 X[TM][4], Y[TM][4];      // operands of even / odd slabs: [row group][plane * 2 + k step]
  auto load_into = [&](u32x4 (&buf)[TM][4]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // inline asm, so that EVERY wait for these registers is the explicit one in compute(): compiler-tracked loads made
      // it guard the loop-carried buffers with vmcnt(7..0) inside the MFMA stream and the address temporaries with
      // vmcnt(0) right behind the first weight piece (measured: 16 % slower than the one-pixel kernel)
      const char* a = na[i];
      const char* a2 = a + nd2;
      const char* a1 = a + 4096;
      const char* a3 = a2 + 4096;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[i][0]) : "v"(a) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[i][1]) : "v"(a1) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[i][2]) : "v"(a2) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[i][3]) : "v"(a3) : "memory");
    }
  };
  auto dma_share = [&](int stage) {     // this wave's pieces of the slab advance() just described
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const char* src = nb + boff[i];
      const unsigned m0v = __builtin_amdgcn_readfirstlane(bm0[i] + (unsigned)stage * (unsigned)STAGE);
      asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  };
  constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};     // h1g1 h1g2 h2g1 (activation plane, weight plane)
  auto compute = [&](u32x4 (&buf)[TM][4], int stage) {
    // EVERY load segment issues its 8 operand loads (past the last slab: harmless re-loads), so here everything older
    // than the youngest 8 has to have landed: this buffer's operands and this wave's weight pieces of the preceding
    // segment (they precede those loads; vmcnt retires in order)
    __builtin_amdgcn_s_waitcnt(0x0F78);                   // vmcnt(8)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(buf[i][k]));     // (values are defined from here on)
    const char* bst = Bs + stage * STAGE;
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
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j][PB[q]]),
                                                               __builtin_bit_cast(f16x8, buf[i][PA[q] * 2 + ks]),
                                                               acc[i][j], 0, 0, 0);
    }
  };
  // segment boundary: nothing may be scheduled across it.  The bare s_barrier, NOT __syncthreads(): the latter fences
  // with s_waitcnt vmcnt(0), i.e. every load segment would end by sitting out the loads it has just issued.  What a
  // barrier has to publish here are LDS-direct weight pieces, and their issuing wave has already waited for them
  // (the vmcnt(8) at the start of its next MFMA segment) one barrier before anybody reads them.
  auto seg_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one slab of this half: MFMA segment, then (while the other half computes) the loads of the slab after next
  auto step = [&](int s, u32x4 (&buf)[TM][4]) {
    compute(buf, s % 3);
    seg_barrier();
    // unconditional (no phi copies of registers an asm load is still filling): past the end the cursor stays on the last
    // slab - the extra fill goes to the stage of slab s - 1, which nobody reads any more
    if (s + 2 < nslab) advance();
    dma_share((s + 2) % 3);
    load_into(buf);
    seg_barrier();
  };

  if (nslab > 0) { advance(); dma_share(0); load_into(X); }
  if (nslab > 1) { advance(); dma_share(1); load_into(Y); }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (half == 1) seg_barrier();          // half B idles through segment 0
  for (int s = 0; s < nslab; s += 2) {
    step(s, X);
    if (s + 1 < nslab) step(s + 1, Y);
  }
  if (half == 0) seg_barrier();          // half A idles through the last segment
  // The last load segments' (unused) operand loads may still be in flight: their destination registers must stay reserved
  // until they have landed - a dead asm output is a free register to the compiler, and a late load return into an
  // address temporary faulted (odd slab counts).
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(X[i][k]), "v"(Y[i][k]));

  // ---- epilogue (as the direct kernel) -------------------------------------------------------------------------------------
  const float acc_scale = *p.acc_scale;
  auto bias_of = [&](int j, f32x4 (&bq)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (p.bias) bq[q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 32 + 4 * fh + 8 * q);
      else bq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (p.out_gl == 0) {
    char* const wl = Bs + wave * 8192;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
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
              v[c] = acc[i][j][4 * q + c] * acc_scale + bq[q][c];
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
          const int mm = m0 + 32 * i + r;
          if (mm < p.M)
            *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + (long long)pix * p.out_pix +
                                      magat_row_off(mm, p.ldc, p.out_tile) + n0 + ps * 64 + 4 * u) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + 32 * i + fr;
    if (m >= p.M) continue;
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
            v[c] = acc[i][j][4 * q + c] * acc_scale + bq[q][c];
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

}  // namespace

// Takes: in_fmt 4, out_fmt 0, in_gl 2, out_gl 0 or 2, Cout % 128 == 0, Cin % 32 == 0, C2 % 32 == 0, no pooling.  Returns
// MAGAT_ERR_UNSUPPORTED otherwise (the caller then uses the direct kernel).
int magat_conv_gemm_f16x3_duo(const magat_conv_gemm_desc* d, hipStream_t st) {
  if (d->in_fmt != 4 || d->out_fmt != 0 || d->in_gl != 2 || (d->out_gl != 0 && d->out_gl != 2) || d->pool)
    return MAGAT_ERR_UNSUPPORTED;
  if ((d->Cout % 128) || (d->Cin % 32) || (d->C2 % 32) || (d->C2 > 0 && !d->in2)) return MAGAT_ERR_UNSUPPORTED;
  if (d->out_gl == 0 && ((d->ldc & 3) || (reinterpret_cast<uintptr_t>(d->out) & 15))) return MAGAT_ERR_UNSUPPORTED;
  if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15)) return MAGAT_ERR_UNSUPPORTED;
  DuoParams p;
  p.in = reinterpret_cast<const char*>(d->in);
  p.in2 = reinterpret_cast<const char*>(d->in2);
  p.wt = reinterpret_cast<const char*>(d->wt);
  p.bias = d->bias;
  p.out = d->out;
  p.in_pix = d->in_pix_stride; p.in2_pix = d->in2_pix_stride; p.out_pix = d->out_pix_stride;
  p.in_tile = d->in_tile_stride ? d->in_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda;
  p.in2_tile = d->in2_tile_stride ? d->in2_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda2;
  p.out_tile = d->out_tile_stride ? d->out_tile_stride : (long long)MAGAT_TILE_ROWS * d->ldc;
  p.M = d->M; p.Mt = (d->M + 511) / 512;
  p.Cin = d->Cin; p.Hin = d->Hin; p.Win = d->Win; p.kH = d->kH; p.kW = d->kW; p.stride = d->stride; p.pad = d->pad;
  p.Hout = d->Hout; p.Wout = d->Wout; p.C2 = d->C2; p.W2 = d->W2; p.stride2 = d->stride2;
  p.Cout = d->Cout; p.Ktot = d->kH * d->kW * d->Cin + d->C2; p.ldc = d->ldc; p.relu = d->relu;
  p.wt_plane = (long long)p.Cout * p.Ktot;
  p.ntn = p.Cout / BN; p.npix = d->Hout * d->Wout; p.out_gl = d->out_gl; p.tag = d->tag;
  p.acc_scale = reinterpret_cast<const float*>(p.wt + (size_t)2 * p.Cout * p.Ktot * 2);
  const long long t128 = (d->M + 127) / 128;
  if (t128 * p.in_tile * 4 >= 0xffffffffLL || (p.C2 > 0 && t128 * p.in2_tile * 4 >= 0xffffffffLL))
    return MAGAT_ERR_UNSUPPORTED;
  const long long groups = (p.Mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD;
  const long long grid = groups * MAGAT_NUM_XCD * p.npix * p.ntn;
  if (grid <= 0 || grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  constexpr size_t lds = 64 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_f16x3_duo_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return MAGAT_ERR_LAUNCH;
    attr_set = true;
  }
  const int pid = magat_prof_begin(p.tag, st);
  hipLaunchKernelGGL(conv_gemm_f16x3_duo_kernel, dim3((unsigned)grid), dim3(512), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
