// Segmented-K fp32 NT GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Serves every dense per-agent map of the MAGAT forward:
//   - conv3x3 / conv1x1 + folded BatchNorm (+ residual 1x1 branch as an extra K segment) + ReLU
//     of BasicBlock / ResNet (reference graphs/models/resnet_pytorch.py:40-73, 427-524),
//   - avgpool+fc+Flatten+Linear folded into one 6x6 "valid" conv,
//   - nn.Linear (compressMLP, actionsMLP; decentralplanner_GAT_bottleneck.py:155-166, 221-239),
//   - the GAT layer's hoisted linear maps X @ [W_p | H_{p,k}]^T (graphML.py:1257, 1765-1770).
//
// Activations are pixel-major [pixel][agent][channel]: for one output pixel a 3x3 conv is a
// sum over <= 9 taps of plain [agents x Cin] @ [Cin x Cout] GEMMs whose A rows are contiguous,
// so there is no im2col gather and taps that fall in the zero padding are skipped outright.
//
// Tile: BM x BN x 32, 256 threads = 4 waves, each wave TM x TN MFMA tiles of 32x32.
// LDS rows are padded to 36 floats: a ds_read_b128 16-lane group then touches 16 distinct
// 16-byte slots (9*r mod 16 is a permutation) -> conflict-free.  Inside a 32-wide K slab the
// lane halves own k = 0..15 / 16..31, so each lane fetches its 16 A (or B) operands of a row
// with four ds_read_b128; the MFMA's k-pairing is (s, 16+s), a reordering of the same sum.
#include <cstdlib>

#include "magat_common.h"
#include "skinny_rows.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

struct ConvGemmParams {
  const float* in;
  const float* in2;
  const float* wt;
  const float* bias;
  float* out;
  long long in_pix_stride, in2_pix_stride, out_pix_stride;
  long long in_tile, in2_tile, out_tile;   // agent-tile strides (magat_row_off)
  int M, Mt;
  int Cin, lda, Hin, Win, kH, kW, stride, pad, Hout, Wout;
  int C2, lda2, W2, stride2;
  int Cout, Ktot, ldc, relu;
  int ntn, npix;
  int tag;
  int pool_w;   // POOL: physical input width (input pixel (iy,ix) = sum or max of the 2x2 physical pixels)
  int pool_max; // POOL: 0 = sum (AvgPool2d with 1/4 in the weights), 1 = max (MaxPool2d)
  int dil;      // tap spacing (nn.Conv2d dilation), >= 1
  long long wt_pix;         // per-output-pixel weight offset (floats); 0 = shared weights
  long long out_plane;
  long long out_nt;         // > 0: 128-column tile t of the row-major output lives at out + t * out_nt (row stride ldc = 128)
  const int* run_if;        // range-guard re-run: the kernel does nothing unless *run_if != 0 (null: always runs)
  float* absmax;            // calibration: largest |output| of the launch (atomic max of the float's bits), or null
  int vgrid;                // number of tiles (virtual workgroups); the launch grid is smaller only for predicated re-runs
};

// FULL: M % BM == 0, Cout % BN == 0, Cin % 32 == 0, C2 % 32 == 0 -> the loader has no bounds checks and
// addresses every 16-byte load as (wave-uniform 64-bit base) + (per-thread 32-bit byte offset).
// NBUF = 2: register-staged double buffer, one barrier per slab.  NBUF = 1: single LDS buffer, two barriers
// per slab, half the LDS -> one more workgroup per CU.
template <int BM, int BN, int WGM, int WGN, bool POOL, bool FULL, int NBUF>
__device__ __forceinline__ void conv_gemm_tile(const ConvGemmParams& p, const int bid) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM / 32, BI = BN / 32;  // float4 loads per thread per slab
  __shared__ float lds[NBUF * (BM + BN) * LDS_LD];
  float* As = lds;
  float* Bs = lds + NBUF * BM * LDS_LD;

  // XCD-aware block -> tile map: the dispatcher places block b on XCD b % 8, so give each XCD
  // whole agent tiles (all pixels x all Cout tiles): the tile's inputs stay in that XCD's L2
  // while its <= 9-fold tap re-reads happen.
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per_m = p.npix * p.ntn;
  const int mtile = xcd + MAGAT_NUM_XCD * (slot / per_m);
  if (mtile >= p.Mt) return;
  const int rem = slot % per_m;
  const int pix = rem / p.ntn, ntile = rem % p.ntn;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int oy = pix / p.Wout, ox = pix % p.Wout;
  const float* const wtp = p.wt + (long long)pix * p.wt_pix;

  // valid tap window of this output pixel
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  const int dl = p.dil;      // tap (ty, tx) reads input pixel (iy0 + ty dl, ix0 + tx dl): the taps inside the map are still a range
  const int ty0 = iy0 < 0 ? (-iy0 + dl - 1) / dl : 0, tx0 = ix0 < 0 ? (-ix0 + dl - 1) / dl : 0;
  const int ty1 = min(p.kH, (p.Hin - iy0 + dl - 1) / dl), tx1 = min(p.kW, (p.Win - ix0 + dl - 1) / dl);
  // (window lengths clamped at 0: with stride >= 3 both can be negative for a pixel whose window lies past the map - the input
  //  gradient over a zero-stuffed map - and their product would be a positive tap count that reads past the map)
  const int ntx = max(tx1 - tx0, 0);
  const int ntaps = max(ty1 - ty0, 0) * ntx;
  const int spt = (p.Cin + BK - 1) / BK;
  const int spt2 = (p.C2 + BK - 1) / BK;
  const int nslab = ntaps * spt + spt2;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int c4 = t & 7, r0 = t >> 3;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[AI], rb[BI];
  unsigned aoff[AI], aoff2[AI], boff[BI];   // FULL path: per-thread byte offsets
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    aoff[i] = (unsigned)((magat_row_off(m0 + r0 + 32 * i, p.lda, p.in_tile) + c4 * 4) * 4);
    aoff2[i] = (unsigned)((magat_row_off(m0 + r0 + 32 * i, p.lda2, p.in2_tile) + c4 * 4) * 4);
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) boff[i] = (unsigned)(((long long)(n0 + r0 + 32 * i) * p.Ktot + c4 * 4) * 4);

  // slab cursor (slabs are visited strictly in order, so the tap / channel-slab position is advanced
  // incrementally: no integer divisions on the scalar unit between MFMA bursts)
  int cur_ty = ty0, cur_tx = tx0, cur_ks = 0;
  bool cur_main = ntaps > 0;
  auto tap_base = [&](int ty, int tx) -> const float* {
    if (POOL) return p.in + (long long)(2 * (iy0 + ty * dl) * p.pool_w + 2 * (ix0 + tx * dl)) * p.in_pix_stride;
    return p.in + (long long)((iy0 + ty * dl) * p.Win + (ix0 + tx * dl)) * p.in_pix_stride;
  };
  const float* cur_tap = tap_base(ty0, tx0);
  const float* const seg2_base =
      p.in2 + (long long)(oy * p.stride2 * p.W2 + ox * p.stride2) * p.in2_pix_stride;

  auto load_slab = [&](int s) {
    (void)s;
    const float* abase;
    long long lda, atile;
    int bk, kvalid;
    const bool main_seg = cur_main;
    const int k0 = cur_ks * BK;
    if (main_seg) {
      abase = cur_tap + k0;
      lda = p.lda;
      atile = p.in_tile;
      bk = (cur_ty * p.kW + cur_tx) * p.Cin + k0;
      kvalid = p.Cin - k0;
      if (++cur_ks == spt) {
        cur_ks = 0;
        if (++cur_tx == tx1) {
          cur_tx = tx0;
          if (++cur_ty == ty1) cur_main = false;
        }
        if (cur_main) cur_tap = tap_base(cur_ty, cur_tx);
      }
    } else {
      abase = seg2_base + k0;
      lda = p.lda2;
      atile = p.in2_tile;
      bk = p.kH * p.kW * p.Cin + k0;
      kvalid = p.C2 - k0;
      ++cur_ks;
    }
    if constexpr (FULL) {
      const char* ab = reinterpret_cast<const char*>(abase);
      const char* bb = reinterpret_cast<const char*>(wtp + bk);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const char* src = ab + (main_seg ? aoff[i] : aoff2[i]);
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        if (POOL && main_seg) {
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4 * p.in_pix_stride);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + 4 * (long long)p.pool_w * p.in_pix_stride);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + 4 * (long long)(p.pool_w + 1) * p.in_pix_stride);
          v = p.pool_max ? __builtin_elementwise_max(__builtin_elementwise_max(v, v1), __builtin_elementwise_max(v2, v3))
                         : v + v1 + v2 + v3;
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bb + boff[i]);
      return;
    }
    const bool kok = c4 * 4 < kvalid;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = m0 + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && m < p.M) {
        const float* src = abase + magat_row_off(m, lda, atile) + c4 * 4;
        v = *reinterpret_cast<const f32x4*>(src);
        if (POOL && main_seg) {   // 2x2 pool on load (sum: the 1/4 lives in the weights; or max)
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + p.in_pix_stride);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (long long)p.pool_w * p.in_pix_stride);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (long long)(p.pool_w + 1) * p.in_pix_stride);
          v = p.pool_max ? __builtin_elementwise_max(__builtin_elementwise_max(v, v1), __builtin_elementwise_max(v2, v3))
                         : v + v1 + v2 + v3;
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && n < p.Cout) v = *reinterpret_cast<const f32x4*>(wtp + (long long)n * p.Ktot + bk + c4 * 4);
      rb[i] = v;
    }
  };
  auto store_slab = [&](int buf) {
    float* a = As + buf * BM * LDS_LD;
    float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDS_LD + c4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDS_LD + c4 * 4) = rb[i];
  };

  const int frag_off = (lane & 31) * LDS_LD + 16 * (lane >> 5);
  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDS_LD + (wm * WTM) * LDS_LD + frag_off;
    const float* b = Bs + buf * BN * LDS_LD + (wn * WTN) * LDS_LD + frag_off;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // two halves of the lane's 16 k values: bounds live registers
      f32x4 fa[TM][2], fb[TN][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          fa[i][q] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_LD + (2 * h + q) * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          fb[j][q] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_LD + (2 * h + q) * 4);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q][e], fa[i][q][e], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (NBUF == 2) {
    if (nslab > 0) {
      load_slab(0);
      store_slab(0);
    }
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
      if (s + 1 < nslab) load_slab(s + 1);
      compute(buf);
      if (s + 1 < nslab) store_slab(buf ^ 1);
      __syncthreads();
    }
  } else {
    if (nslab > 0) load_slab(0);
    for (int s = 0; s < nslab; ++s) {
      if (s > 0) __syncthreads();      // every wave is done reading the previous slab
      store_slab(0);
      __syncthreads();
      if (s + 1 < nslab) load_slab(s + 1);   // in flight under this slab's MFMAs
      compute(0);
    }
  }

  // epilogue: bias (+ReLU).  The weight tile is the MFMA's ROW operand, so D[channel][agent]:
  // agent = lane&31, channel = (r&3) + 8*(r>>2) + 4*(lane>>5): registers 4q..4q+3 hold four consecutive channels
  // of one agent -> one 16-byte store (float32) or one 8-byte store per plane (bf16x3) instead of four scalars.
  float* obase = p.out + (long long)pix * p.out_pix_stride;
  unsigned short* sbase = reinterpret_cast<unsigned short*>(p.out) + (long long)pix * p.out_pix_stride;
  const bool vec = FULL && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && (p.out_plane & 3) == 0;
  // the lane's bias values first, as one batch of loads (a conditional load per channel inside the store loop costs
  // one L2 round trip per channel quad)
  float amax = 0.f;
  f32x4 bq[TN][4];
  const bool bvec = p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0 && n0 + BN <= p.Cout;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * WTN + j * 32 + 4 * (lane >> 5) + 8 * q;
      if (bvec) {
        bq[j][q] = *reinterpret_cast<const f32x4*>(p.bias + n);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[j][q][c] = (p.bias && n + c < p.Cout) ? p.bias[n + c] : 0.f;
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
          v[c] = acc[i][j][4 * q + c] + bq[j][q][c];
          if (p.relu) v[c] = magat_relu(v[c]);
          if (n + c < p.Cout) amax = fmaxf(amax, fabsf(v[c]));
        }
        const long long o = magat_row_off(m, p.ldc, p.out_tile) + (p.out_nt ? (long long)(n >> 7) * p.out_nt + (n & 127) : n);
        if (vec) {
          *reinterpret_cast<f32x4*>(obase + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (n + c < p.Cout) obase[o + c] = v[c];
        }
      }
    }
  }
  if (p.absmax) {      // (calibration launches only)
    amax = wave_max(amax);
    if (lane == 0 && amax > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.absmax), __builtin_bit_cast(unsigned, amax));
  }
}

// Normal launches: one workgroup per tile (gridDim.x == vgrid: the loop runs once).  The range guard's predicated re-run
// (run_if set) is launched with a SMALL grid whose workgroups walk the tiles - when the flag is clear, which is the rule,
// a few hundred workgroups return at once instead of tens of thousands being dispatched for nothing.
template <int BM, int BN, int WGM, int WGN, bool POOL, bool FULL, int NBUF, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv_gemm_kernel(const ConvGemmParams p) {
  if (p.run_if && *p.run_if == 0) return;
  for (int bid = blockIdx.x; bid < p.vgrid; bid += gridDim.x) {
    conv_gemm_tile<BM, BN, WGM, WGN, POOL, FULL, NBUF>(p, bid);
    if (bid + (int)gridDim.x < p.vgrid) __syncthreads();     // the next tile reuses the LDS stages
  }
}

// ---- several layers in ONE launch, for the range guard's float32 re-run of the encoder (a dozen predicated launches that
// return at once cost a dispatch each, ~5 us: 1.5 % of a c3 step, 13 % of a batch-1 step).  Every layer's rows are agents: a
// workgroup that computes ALL tiles of layer l for one 128-agent tile has everything layer l + 1 needs for those agents, so
// the layers chain inside the workgroup - no grid-wide barrier - with a release / acquire fence pair around the workgroup
// barrier between two layers (the next layer's loads must not hit lines the vector L1 kept from three layers ago: the map
// buffers rotate).  One generic 128 x 128 tile variant serves every layer (bounds-checked loader; a 32-channel layer wastes
// three quarters of its tile: this path is the exception, not the rule).
constexpr int CHAIN_MAX = 10;
struct ConvChain {
  ConvGemmParams L[CHAIN_MAX];
  int n, Mt;
  const int* run_if;
  int* book;            // range-guard bookkeeping by the last workgroup (magat_guard_book), or null
};

__global__ __launch_bounds__(256) void conv_gemm_chain_kernel(const ConvChain c) {
  if (c.run_if && *c.run_if == 0) {
    if (c.book) magat_guard_book_idle(c.book);
    return;
  }
  for (int mt = blockIdx.x; mt < c.Mt; mt += gridDim.x) {
#pragma unroll 1
    for (int l = 0; l < c.n; ++l) {
      const ConvGemmParams& p = c.L[l];
      const int per_m = p.npix * p.ntn;
#pragma unroll 1
      for (int r = 0; r < per_m; ++r) {
        const int bid = ((mt / MAGAT_NUM_XCD) * per_m + r) * MAGAT_NUM_XCD + (mt % MAGAT_NUM_XCD);      // (conv_gemm_tile's map)
        if (p.pool_w) conv_gemm_tile<128, 128, 2, 2, true, false, 1>(p, bid);
        else conv_gemm_tile<128, 128, 2, 2, false, false, 1>(p, bid);
        __syncthreads();                                  // the next tile reuses the LDS stages
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this layer's stores are out of the CU ...
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ... and the next layer's loads come from L2
    }
  }
  if (c.book) magat_guard_book(c.book);
}

template <int BM, int BN, int WGM, int WGN, bool POOL, bool FULL>
int launch2(ConvGemmParams& p, hipStream_t st, long long grid) {
  const int pid = magat_prof_begin(p.tag, st);
  // Measured on MI355X (tools/conv_bench.py, 51200 agents): the single-buffer form wins everywhere because it
  // fits one more workgroup per CU (l3.conv2: 122 -> 129 TF); capping the 128x128 tile at 128 registers
  // (4 waves/SIMD, 11 spilled VGPRs) adds another 2-3 % (131.6 TF).
  constexpr int MINW = (BM == 128 && BN == 128) ? 4 : 1;
  p.vgrid = (int)grid;
  const unsigned launch_grid = (unsigned)((p.run_if && grid > 512) ? 512 : grid);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WGM, WGN, POOL, FULL, 1, MINW>), dim3(launch_grid), dim3(256), 0, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

template <int BM, int BN, int WGM, int WGN, bool POOL = false>
int launch(ConvGemmParams& p, hipStream_t st) {
  p.Mt = (p.M + BM - 1) / BM;
  p.ntn = (p.Cout + BN - 1) / BN;
  const long long groups = (p.Mt + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD;
  const long long grid = groups * MAGAT_NUM_XCD * p.npix * p.ntn;
  if (grid <= 0 || grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const bool full = p.M % BM == 0 && p.Cout % BN == 0 && p.Cin % BK == 0 && p.C2 % BK == 0 &&
                    magat_row_off(p.M, p.lda, p.in_tile) * 4 < 0xffffffffLL &&
                    (p.C2 == 0 || magat_row_off(p.M, p.lda2, p.in2_tile) * 4 < 0xffffffffLL) &&
                    (long long)p.Cout * p.Ktot * 4 < 0xffffffffLL;
  return full ? launch2<BM, BN, WGM, WGN, POOL, true>(p, st, grid) : launch2<BM, BN, WGM, WGN, POOL, false>(p, st, grid);
}

}  // namespace

// ---- skinny layers (the action head: 640 -> 5 at c3): a GEMM whose output is a handful of columns is a stream of dot products -
// the 32-column MFMA tile spends 27 of its 32 columns on padding and the kernel ran at 0.4-0.6 of the HBM roof.  The row arithmetic
// lives in skinny_rows.h (shared with gat_rerun_small_kernel, which carries the action head of few instances in its launch).
namespace {
typedef MagatSkinnyParams SkinnyParams;
template <int CO>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const SkinnyParams p) {
  extern __shared__ __align__(16) float sk_w[];      // [Ktot / 4][CO][4]
  magat_skinny_stage_weights<CO>(p, sk_w);
  __syncthreads();
  magat_skinny_rows<CO>(p, sk_w, (long long)blockIdx.x * 16, p.M, (long long)gridDim.x * 16);
}

template <int CO>
int skinny_launch(const SkinnyParams& p, int tag, hipStream_t st) {
  const size_t lds = (size_t)(p.Cin + p.C2) * CO * sizeof(float);
  if (lds > 64 * 1024) return MAGAT_ERR_UNSUPPORTED;
  long long blocks = ((long long)p.M + 15) / 16;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const int pid = magat_prof_begin(tag, st);
  hipLaunchKernelGGL(skinny_gemm_kernel<CO>, dim3((unsigned)blocks), dim3(256), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

// takes the layer when it is a plain row-major 1x1 product with at most 8 outputs; MAGAT_ERR_UNSUPPORTED = not this form
int skinny_try(const magat_conv_gemm_desc* d, hipStream_t st) {
  SkinnyParams p;
  if (magat_skinny_params(d, &p) != MAGAT_OK) return MAGAT_ERR_UNSUPPORTED;
  switch (d->Cout) {
    case 1: return skinny_launch<1>(p, d->tag, st);
    case 2: return skinny_launch<2>(p, d->tag, st);
    case 3: return skinny_launch<3>(p, d->tag, st);
    case 4: return skinny_launch<4>(p, d->tag, st);
    case 5: return skinny_launch<5>(p, d->tag, st);
    case 6: return skinny_launch<6>(p, d->tag, st);
    case 7: return skinny_launch<7>(p, d->tag, st);
    default: return skinny_launch<8>(p, d->tag, st);
  }
}
}  // namespace

int magat_skinny_params(const magat_conv_gemm_desc* d, MagatSkinnyParams* out) {
  if (!d || !out) return MAGAT_ERR_NULL;
  if (d->in_fmt != 0 || d->out_fmt != 0 || d->in_gl || d->out_gl || d->Cout > 8 || d->Cout < 1 || d->kH != 1 || d->kW != 1 ||
      d->Hout != 1 || d->Wout != 1 || d->Hin != 1 || d->Win != 1 || d->pool || d->stride != 1 || d->pad != 0 || d->run_if ||
      d->absmax || d->ldw || d->wt_pix_stride || d->in_tile_stride || d->in2_tile_stride || d->out_tile_stride ||
      d->out_ntile_stride || d->in_pix_stride || d->in2_pix_stride || d->out_pix_stride || d->M < 1 ||
      !magat_opt(MAGAT_OPT_SKINNY))
    return MAGAT_ERR_UNSUPPORTED;
  if ((d->Cin & 3) || (d->C2 & 3) || (d->lda & 3) || (d->lda2 & 3) || d->lda < d->Cin || d->ldc < d->Cout || d->C2 < 0 ||
      (d->C2 > 0 && (!d->in2 || d->lda2 < d->C2 || (d->W2 > 1) || d->stride2 > 1)) ||
      ((reinterpret_cast<uintptr_t>(d->wt) | reinterpret_cast<uintptr_t>(d->in) | reinterpret_cast<uintptr_t>(d->in2)) & 15) ||
      (d->bf16_rows & ~3) || ((d->bf16_rows & 2) && d->C2 <= 0) ||
      ((d->bf16_rows & 1) && ((d->Cin & 7) || (d->lda & 7))) || ((d->bf16_rows & 2) && ((d->C2 & 7) || (d->lda2 & 7))))
    return MAGAT_ERR_UNSUPPORTED;
  MagatSkinnyParams& p = *out;
  p.in = static_cast<const float*>(d->in); p.in2 = d->C2 > 0 ? static_cast<const float*>(d->in2) : nullptr;
  p.wt = static_cast<const float*>(d->wt); p.bias = static_cast<const float*>(d->bias); p.out = static_cast<float*>(d->out);
  p.M = d->M; p.Cin = d->Cin; p.C2 = d->C2; p.lda = d->lda; p.lda2 = d->lda2; p.ldc = d->ldc; p.relu = d->relu;
  p.bf16_rows = d->bf16_rows;
  p.Cout = d->Cout;
  return MAGAT_OK;
}

// descriptor -> kernel parameters of the float32 kernel (shape checks included)
static int conv_params_from_desc(const magat_conv_gemm_desc* d, ConvGemmParams& p) {
  if (!d || !d->in || !d->wt || !d->out) return MAGAT_ERR_NULL;
  if (d->in_gl || d->out_gl) return MAGAT_ERR_UNSUPPORTED;   // f16x3 direct kernel only
  if (d->out_ntile_stride && (d->ldc != 128 || (d->Cout & 127) || d->out_fmt != 0)) return MAGAT_ERR_UNSUPPORTED;
  if (d->in_fmt != 0 || d->out_fmt != 0) return MAGAT_ERR_UNSUPPORTED;
  if (d->M <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Hout <= 0 || d->Wout <= 0 || d->kH <= 0 || d->kW <= 0 ||
      d->stride <= 0 || d->pad < 0 || d->C2 < 0)
    return MAGAT_ERR_BAD_SHAPE;
  if ((d->Cin & 3) || (d->C2 & 3) || (d->lda & 3) || d->lda < d->Cin || (d->ldc < d->Cout && !d->out_ntile_stride))
    return MAGAT_ERR_BAD_SHAPE;
  if (d->C2 > 0 && (!d->in2 || (d->lda2 & 3) || d->lda2 < d->C2 || d->stride2 <= 0)) return MAGAT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(d->in) | reinterpret_cast<uintptr_t>(d->wt) |
       reinterpret_cast<uintptr_t>(d->in2)) & 15)
    return MAGAT_ERR_BAD_SHAPE;
  p.in = d->in; p.in2 = d->in2; p.wt = d->wt; p.bias = d->bias; p.out = d->out;
  p.in_pix_stride = d->in_pix_stride; p.in2_pix_stride = d->in2_pix_stride; p.out_pix_stride = d->out_pix_stride;
  p.in_tile = d->in_tile_stride ? d->in_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda;
  p.in2_tile = d->in2_tile_stride ? d->in2_tile_stride : (long long)MAGAT_TILE_ROWS * d->lda2;
  p.out_tile = d->out_tile_stride ? d->out_tile_stride : (long long)MAGAT_TILE_ROWS * d->ldc;
  if ((p.in_tile & 3) || (p.in2_tile & 3)) return MAGAT_ERR_BAD_SHAPE;
  p.M = d->M; p.Cin = d->Cin; p.lda = d->lda; p.Hin = d->Hin; p.Win = d->Win; p.kH = d->kH; p.kW = d->kW;
  p.stride = d->stride; p.pad = d->pad; p.Hout = d->Hout; p.Wout = d->Wout;
  p.C2 = d->C2; p.lda2 = d->lda2; p.W2 = d->W2; p.stride2 = d->stride2;
  p.Cout = d->Cout; p.Ktot = d->kH * d->kW * d->Cin + d->C2; p.ldc = d->ldc; p.relu = d->relu;
  p.absmax = d->absmax;
  if (d->ldw) {
    if (d->ldw < p.Ktot || (d->ldw & 3)) return MAGAT_ERR_BAD_SHAPE;
    p.Ktot = d->ldw;          // weight row stride
  }
  if (d->wt_pix_stride & 3) return MAGAT_ERR_BAD_SHAPE;
  p.wt_pix = d->wt_pix_stride;
  p.npix = d->Hout * d->Wout;
  p.tag = d->tag;
  p.out_plane = d->out_plane_stride;
  p.out_nt = d->out_ntile_stride;
  p.run_if = reinterpret_cast<const int*>(d->run_if);
  p.pool_w = 0;
  p.pool_max = d->pool == 2;
  if (d->dilation < 0 || d->dilation > 64) return MAGAT_ERR_BAD_SHAPE;
  p.dil = d->dilation > 1 ? d->dilation : 1;
  if (d->pool) {   // input map is the 2x2 sum- or max-pool of a physical (2*Hin.. x pool_w) map
    if (d->pool_w < 2 * d->Win || (d->pool != 1 && d->pool != 2)) return MAGAT_ERR_BAD_SHAPE;
    p.pool_w = d->pool_w;
  }
  if ((p.in_pix_stride & 3) || (p.in2_pix_stride & 3)) return MAGAT_ERR_BAD_SHAPE;
  return MAGAT_OK;
}

extern "C" int magat_conv_gemm_f32(const magat_conv_gemm_desc* d, void* stream) {
  if (!d || !d->in || !d->wt || !d->out) return MAGAT_ERR_NULL;
  if (d->bf16_rows && d->in_fmt != 0) return MAGAT_ERR_UNSUPPORTED;
  if (d->wt2 && d->in_fmt != 4) return MAGAT_ERR_UNSUPPORTED;      // (a second layer in the epilogue: f16x3 direct kernel only)
  if (d->dilation > 1 && d->in_fmt != 0) return MAGAT_ERR_UNSUPPORTED;      // (dilated taps: the float32 kernel only)
  if (d->in_fmt >= 1 && d->in_fmt <= 5) return magat_conv_gemm_bf16x6(d, static_cast<hipStream_t>(stream));
  {
    const int src = skinny_try(d, static_cast<hipStream_t>(stream));
    if (src != MAGAT_ERR_UNSUPPORTED) return src;
  }
  if (d->bf16_rows) return MAGAT_ERR_UNSUPPORTED;      // (the streamed form only)
  ConvGemmParams p;
  const int prc = conv_params_from_desc(d, p);
  if (prc != MAGAT_OK) return prc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // few-block launches (head, compressMLP at moderate M): 64-row tiles double the number of workgroups
  const long long blocks128 = (long long)((p.M + 127) / 128) * p.npix * ((p.Cout + 127) / 128);
  const bool small_grid = blocks128 < 1536;
  if (d->pool) return small_grid ? launch<64, 128, 2, 2, true>(p, st) : launch<128, 128, 2, 2, true>(p, st);
  if (p.Cout > 64) return small_grid ? launch<64, 128, 2, 2>(p, st) : launch<128, 128, 2, 2>(p, st);
  if (p.Cout > 32) return launch<128, 64, 2, 2>(p, st);
  return launch<128, 32, 4, 1>(p, st);
}

extern "C" int magat_linear_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M,
                                int N, int K, int relu, void* stream) {
  return magat_linear_tagged_f32(x, ldx, w, b, y, ldy, M, N, K, relu, MAGAT_TAG_UNTAGGED, stream);
}

extern "C" int magat_linear_tagged_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                                       int M, int N, int K, int relu, int tag, void* stream) {
  magat_conv_gemm_desc d = {};
  d.tag = tag;
  d.in = x; d.wt = w; d.bias = b; d.out = y;
  d.M = M; d.Cin = K; d.lda = ldx; d.Hin = d.Win = 1; d.kH = d.kW = 1; d.stride = 1; d.pad = 0;
  d.Hout = d.Wout = 1; d.Cout = N; d.ldc = ldy; d.relu = relu;
  return magat_conv_gemm_f32(&d, stream);
}

// float32 layers chained in one launch (see conv_gemm_chain_kernel): descs[0..n) must share M, every layer reads only what
// earlier layers of the list (or earlier launches) wrote for the SAME agents; run_if as in magat_conv_gemm_desc.
int magat_conv_gemm_chain_f32(const magat_conv_gemm_desc* descs, int n, const int32_t* run_if, int tag, hipStream_t st,
                              int32_t* book) {
  if (!descs || n <= 0 || n > CHAIN_MAX) return MAGAT_ERR_BAD_SHAPE;
  ConvChain c;
  c.n = n;
  c.book = reinterpret_cast<int*>(book);
  c.run_if = reinterpret_cast<const int*>(run_if);
  for (int l = 0; l < n; ++l) {
    if (descs[l].in_fmt != 0 || descs[l].out_fmt != 0 || descs[l].in_gl || descs[l].out_gl || descs[l].out_ntile_stride ||
        descs[l].M != descs[0].M)
      return MAGAT_ERR_UNSUPPORTED;
    const int rc = conv_params_from_desc(&descs[l], c.L[l]);
    if (rc != MAGAT_OK) return rc;
    ConvGemmParams& p = c.L[l];
    p.Mt = (p.M + 127) / 128;
    p.ntn = (p.Cout + 127) / 128;
    p.run_if = nullptr;
    p.vgrid = 0;
  }
  c.Mt = c.L[0].Mt;
  const int grid = c.Mt < 512 ? c.Mt : 512;
  const int pid = magat_prof_begin(tag, st);
  hipLaunchKernelGGL(conv_gemm_chain_kernel, dim3((unsigned)grid), dim3(256), 0, st, c);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
