// GraphFilterBatchAttentional.forward for LARGE SPARSE graphs with bf16 storage, K = 2, KeyQuery, G = F = 128 - BASELINE
// config 5 (1000 agents, comm-radius GSO) - in the REFERENCE'S OWN ORDER: scores, one hop ON THE NODE FEATURES, then the tap
// contraction (graphML.py:1757 `x = x @ aij`, :1768-1770 `z @ h`), with both dense per-agent maps on the matrix cores INSIDE
// the two graph kernels.  The split form of gat_csr_f32.hip writes the maps Z = X [W_p | H_pk]^T (1536 columns: 393 MB at
// config 5) with one launch and reads them back with two; here neither Q nor U ever exists in memory:
//
//   kernel A (scores)   q'_i,p = W_p^T x_i  (MFMA, 32 graph rows per wave, result stays in registers)
//                        e_ij,p = q'_i,p . x_j  for the edges (i,j) of the row, x_j gathered from L2 ONCE for all heads
//                        (v_dot2_f32_bf16: exact products, float32 sums), row softmax  ->  att[p][e]  (float32, CSR order)
//   kernel B (hop+taps) z_j,p = sum over in-edges (i -> j) of att_p(i,j) x_i   (x_i gathered once per head pair, float32 sums)
//                        y_j,p = relu(H_p0 x_j + H_p1 z_j,p + bias)           (MFMA, K = 256, operands = the registers the
//                        hop left behind), bf16 rows (or the same values widened) - the layer's only large write.
//
// Work decomposition: a WAVE owns 32 graph rows; a lane PAIR (l, l + 32) owns one row, lane half h holding the 8-feature
// granules 2m + h (m = 0..7) of every 128-wide vector of that row.  That is exactly the operand layout of
// v_mfma_f32_32x32x16_bf16 (operand k = 16 m + 8 h + 0..7, column = lane % 32), and one v_permlane32_swap per register pair
// turns the accumulator layout (rows 8 j + 4 h + i) into it - so what the matrix cores produce is directly the per-row
// vector of the edge loop, and what the hop accumulates is directly an operand.  No LDS tile of neighbour rows (the tiled
// kernels stream a [N][128 B] slice per (instance, head, pass) and carry partial scores across passes): the 256 KB of an
// instance's X stay L2-resident (all row groups of an instance run on ONE XCD at the same time) and every edge gathers its 256
// bytes once per kernel, shared by all heads.  LDS holds the weights as MFMA fragments (kernel A: 32 KB per head; kernel B:
// 64 KB per head, two heads per workgroup), read-only after the prologue: the persistent loop has no barrier.
// Rows are walked in DEGREE ORDER (csr_rank_kernel: counting sort per instance) so that the 32 rows of a wave have about the
// same number of edges; the groups are dealt to the waves so that every wave sees heavy and light groups.
// One wave per SIMD (all 512 registers: q' of four heads is 128, the hop's accumulators of two heads 128), latency hidden by
// a two-deep software pipeline (indices two slots ahead, rows one slot ahead).
#include "magat_common.h"

typedef unsigned short u16;
typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fbf16x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Int4 {      // four consecutive indices of an edge list: dword-aligned 16-byte load
  int v[4];
};

namespace {

struct FusedParams {
  const u16* X;          // [B*N][128] bf16
  const int* rowptr;     // [B*(N+1)]
  const int* colidx;
  const int* cscptr;     // [B*(N+1)]
  const int* cscsrc;
  const int* cscpos;
  const int* order;      // [2][B][N]: rows by out-degree (descending), rows by in-degree
  float* att;            // [nnz][P]: the P heads of an edge side by side (CSR order)
  const u16* wq;         // [P][4][8][64][8]   fragments of W_p^T:  (feature tile, k step, lane, 8)
  const u16* wh;         // [P][4][16][64][8]  fragments of [H_p0 | H_p1]
  const float* bias;     // [128] or null
  void* Y;
  int ldy, y_f32, act_relu;
  int B, N, P;
  long long nnz;
  long long* dbg;        // FUSED_STAMPS builds: [workgroup][wave][8] cycle totals per phase
};

// Phase stamps (experiment builds, -DFUSED_STAMPS; tools/csr_fused_phases.py): cycles a wave spends in each phase of its steps
#ifdef FUSED_STAMPS
#define STAMP_DECL long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long st_t = __builtin_readcyclecounter(); const long long st_t0 = st_t
#define STAMP(ph) { const long long st_n = __builtin_readcyclecounter(); st_acc[ph] += st_n - st_t; st_t = st_n; }
#define STAMP_OUT(p) { st_acc[7] = __builtin_readcyclecounter() - st_t0; if (p.dbg && lane == 0) for (int i = 0; i < 8; ++i) p.dbg[((long long)blockIdx.x * FUSED_WAVES + wave) * 8 + i] = st_acc[i]; }
#else
#define STAMP_DECL
#define STAMP(ph)
#define STAMP_OUT(p)
#endif

// v_cvt_pk_bf16_f32 (RNE) THROUGH THE COMPILER, not as inline asm: with two waves per SIMD the matrix products leave their
// results in architectural registers (no AGPRs, no v_accvgpr_read in between), and the wait states a vector instruction needs
// behind an MFMA that writes its source are inserted for instructions the compiler knows - an asm statement got none and read
// registers the matrix pipe had not written yet (12 of 12 tests red, run-to-run different results)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, fbf16x2));
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fbf16x2, a), __builtin_bit_cast(fbf16x2, b), c, false);
}
__device__ __forceinline__ f32x16 mfma_bf16(const fu32x4& a, const fu32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fbf16x8, a), __builtin_bit_cast(fbf16x8, b), c, 0, 0, 0);
}
// the upper 32 lanes of `a` trade places with the lower 32 lanes of `b`
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// lower lanes: A(lower) + A(upper), upper lanes: B(lower) + B(upper)
__device__ __forceinline__ float swap_add(float a, float b) {
  unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  swap32(ua, ub);
  return __builtin_bit_cast(float, ua) + __builtin_bit_cast(float, ub);
}
// accumulator tile (rows 8 j + 4 half + i in register 4 j + i, column = lane % 32) -> the two 8-row granules of this lane
// half (granule 2 jj + half, jj = 0, 1) as packed bf16: lower lanes get rows 0-7 and 16-23, upper lanes 8-15 and 24-31
__device__ __forceinline__ void acc_to_granules(const float (&v)[16], fu32x4 (&g)[2]) {
  unsigned pk[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pk[j][0] = pk_bf16(v[4 * j], v[4 * j + 1]);
    pk[j][1] = pk_bf16(v[4 * j + 2], v[4 * j + 3]);
  }
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    swap32(pk[2 * jj][0], pk[2 * jj + 1][0]);
    swap32(pk[2 * jj][1], pk[2 * jj + 1][1]);
    g[jj] = fu32x4{pk[2 * jj][0], pk[2 * jj][1], pk[2 * jj + 1][0], pk[2 * jj + 1][1]};
  }
}

// the lane's eight 16-byte granules of row j (elements 16 m + 8 half ..)
__device__ __forceinline__ void gather_row(const u16* Xb, int j, int goff, fu32x4 (&x)[8]) {
#ifdef FUSED_WHATIF_COALESCED      // timing experiment (wrong results): the four lanes of a quad read 64 contiguous bytes of ONE row
  const int j0 = __builtin_amdgcn_update_dpp(0, j, 0x00, 0xf, 0xf, true);      // quad_perm [0,0,0,0]
  const u16* s = Xb + (long long)j0 * 128 + 8 * (threadIdx.x & 3);
#pragma unroll
  for (int m = 0; m < 8; ++m) x[m] = *reinterpret_cast<const fu32x4*>(s + 32 * (m & 3));
#else
  const u16* s = Xb + (long long)j * 128 + goff;
#pragma unroll
  for (int m = 0; m < 8; ++m) x[m] = *reinterpret_cast<const fu32x4*>(s + 16 * m);
#endif
}

// rows of an instance ranked by edge count (descending): counting sort, 64 bins (63+ edges share the first)
__global__ __launch_bounds__(1024) void csr_rank_kernel(const int* __restrict__ rowptr, const int* __restrict__ cscptr,
                                                        int* __restrict__ order, int B, int N) {
  __shared__ int hist[64], base[64];
  const int t = threadIdx.x;
  const int* ptr = (blockIdx.y ? cscptr : rowptr) + (long long)blockIdx.x * (N + 1);
  int* out = order + ((long long)blockIdx.y * B + blockIdx.x) * N;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int i = t; i < N; i += 1024) {
    const int d = ptr[i + 1] - ptr[i];
    atomicAdd(&hist[63 - (d < 63 ? d : 63)], 1);
  }
  __syncthreads();
  if (t < 64) {
    const int c = hist[t];
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (t >= o) incl += v;
    }
    base[t] = incl - c;
  }
  __syncthreads();
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int i = t; i < N; i += 1024) {
    const int d = ptr[i + 1] - ptr[i];
    const int key = 63 - (d < 63 ? d : 63);
    out[base[key] + atomicAdd(&hist[key], 1)] = i;
  }
}

// Two waves per SIMD (256 registers each): while one wave issues its matrix products the other one's gathers are in flight
// (one wave per SIMD, 512 registers: both kernels 120 us at config 5 - every phase of a wave's step waited for its own loads)
#ifndef FUSED_WAVES_N
#define FUSED_WAVES_N 8
#endif
constexpr int FUSED_WAVES = FUSED_WAVES_N, FUSED_THREADS = 64 * FUSED_WAVES;

// which (instance, 32-row group) a wave works on in step s of its workgroup's persistent loop
struct Item {
  int b, gi;
  bool ok;
};
__device__ __forceinline__ Item fused_item(int u, int s, int wave, int nchunk, int ng, int B) {
  const int xcd = blockIdx.x % MAGAT_NUM_XCD;
  Item it;
  it.b = xcd + MAGAT_NUM_XCD * (u / nchunk);
  const int c = u % nchunk;
  // chunk c of an instance = its groups c, c + nchunk, c + 2 nchunk, .. of the degree ranking (from heavy to light), dealt to
  // the waves in an order that rotates with the step
  it.gi = c + nchunk * ((wave + s) % FUSED_WAVES);
  it.ok = it.b < B && it.gi < ng;
  return it;
}

// ---- kernel A: scores + row softmax, all P heads of a row group
template <int P>
__global__ __launch_bounds__(FUSED_THREADS) void csr_fused_scores_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int HO = P >= 2 ? P / 2 : 1;          // heads whose scores this lane half ends up with
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, half = lane >> 5;
  STAMP_DECL;
  {
    const fu32x4* src = reinterpret_cast<const fu32x4*>(p.wq);
    fu32x4* dst = reinterpret_cast<fu32x4*>(lds);
    for (int i = t; i < P * 2048; i += FUSED_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  STAMP(6)
  const int N = p.N, ng = (N + 31) >> 5, nchunk = (ng + FUSED_WAVES - 1) / FUSED_WAVES;
  const int m0 = blockIdx.x / MAGAT_NUM_XCD, wgs = gridDim.x / MAGAT_NUM_XCD;
  const int items = ((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * nchunk;
  const int goff = 8 * half;
  const char* const frag = lds + lane * 16;
  for (int u = m0, s = 0; u < items; u += wgs, ++s) {
    const Item it = fused_item(u, s, wave, nchunk, ng, p.B);
    if (!it.ok) continue;
    const int b = it.b, pos = it.gi * 32 + r;
    const bool valid = pos < N;
    const int row = p.order[(long long)b * N + (valid ? pos : it.gi * 32)];
    const int* rp = p.rowptr + (long long)b * (N + 1);
    const int e0 = rp[row];
    const int deg = valid ? rp[row + 1] - e0 : 0;
    const u16* Xb = p.X + (long long)b * N * 128;
    // q' = W_p^T x_row on the matrix cores: D[feature][graph row] = sum_g Wfrag[feature][g] X^T[g][row]
    fu32x4 q[P][8];
    {
      fu32x4 xo[8];
      gather_row(Xb, row, goff, xo);
#ifdef FUSED_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(0)
#endif
#pragma unroll
      for (int hp = 0; hp < P; ++hp)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          f32x16 acc = {};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            acc = mfma_bf16(*reinterpret_cast<const fu32x4*>(frag + ((hp * 4 + ft) * 8 + ks) * 1024), xo[ks], acc);
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i];
          fu32x4 g[2];
          acc_to_granules(v, g);
          q[hp][2 * ft] = g[0];
          q[hp][2 * ft + 1] = g[1];
#ifndef FUSED_NO_SB
          if (ft == 3) __builtin_amdgcn_sched_barrier(0);
#endif      // (one head's accumulators live at a time: 64 registers, not 256)
        }
    }
    STAMP(1)
    // edge loop: slot k = the k-th edge of every row of the group.  Both kernels run at the rate a CU gathers random 128-byte lines
    // out of L2 (~16 GB/s per CU whatever the instruction shape: DESIGN.md 4.5), so everything that is NOT a row gather is kept
    // small: neighbour indices four slots per load, the raw scores of the first eight slots in registers (no store + read-back),
    // attention as [edge][head] (one store per lane and slot).
    float mx[HO], sm[HO];
#pragma unroll
    for (int o = 0; o < HO; ++o) {
      mx[o] = -__builtin_inff();
      sm[o] = 0.f;
    }
    const int hbase = P >= 2 ? HO * half : 0;          // first head this lane half owns
    float* const attr = p.att + (long long)e0 * P + hbase;      // att[(e0 + k) * P + head]
    auto score = [&](const fu32x4 (&x)[8], float (&own)[HO]) __attribute__((always_inline)) {
      float sc[P];
#pragma unroll
      for (int hp = 0; hp < P; ++hp) {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          d0 = dot2(q[hp][m][0], x[m][0], d0);
          d1 = dot2(q[hp][m][1], x[m][1], d1);
          d0 = dot2(q[hp][m][2], x[m][2], d0);
          d1 = dot2(q[hp][m][3], x[m][3], d1);
        }
        sc[hp] = d0 + d1;
      }
      if constexpr (P == 4) {
        own[0] = swap_add(sc[0], sc[2]);
        own[1] = swap_add(sc[1], sc[3]);
      } else if constexpr (P == 2) {
        own[0] = swap_add(sc[0], sc[1]);
      } else {
        own[0] = swap_add(sc[0], sc[0]);
      }
    };
    auto track = [&](const float (&own)[HO]) __attribute__((always_inline)) {
#pragma unroll
      for (int o = 0; o < HO; ++o) {
        const float m2 = fmaxf(mx[o], own[o]);
        sm[o] = sm[o] * __expf(mx[o] - m2) + __expf(own[o] - m2);
        mx[o] = m2;
      }
    };
    auto put = [&](int k, const float (&v)[HO]) __attribute__((always_inline)) {
      if constexpr (HO == 2) *reinterpret_cast<f32x2*>(attr + (long long)k * P) = f32x2{v[0], v[1]};
      else if (P >= 2 || half == 0) attr[(long long)k * P] = v[0];
    };
    auto load4 = [&](int k0, int (&c)[4]) __attribute__((always_inline)) {
      if (k0 + 4 <= deg) {
        const Int4 v = *reinterpret_cast<const Int4*>(p.colidx + e0 + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = v.v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = k0 + i < deg ? p.colidx[e0 + k0 + i] : row;
      }
    };
    constexpr int KR = 8;                               // slots whose raw scores stay in registers
    float rawr[KR][HO];
    fu32x4 xa[8], xb[8];
    int cj[4], cn[4];
    load4(0, cj);
    gather_row(Xb, cj[0], goff, xa);
    bool live = true;
#pragma unroll
    for (int bt = 0; bt < KR / 4; ++bt) {
      if (live) load4(4 * bt + 4, cn);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * bt + c;
        if (live && __builtin_amdgcn_ballot_w64(k < deg) == 0ull) live = false;
        if (live) {
          const int nx = c < 3 ? cj[c + 1] : cn[0];
          if ((k & 1) == 0) {
            gather_row(Xb, nx, goff, xb);
            score(xa, rawr[k]);
          } else {
            gather_row(Xb, nx, goff, xa);
            score(xb, rawr[k]);
          }
          if (k < deg) track(rawr[k]);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) cj[c] = cn[c];
    }
    if (live) {      // rows with more than KR edges: the rest through memory (raw score stored, read back below)
      int c1 = cj[1];
      for (int k = KR;; k += 2) {
        if (__builtin_amdgcn_ballot_w64(k < deg) == 0ull) break;
        const int c2 = k + 2 < deg ? p.colidx[e0 + k + 2] : row;
        gather_row(Xb, c1, goff, xb);
        float own[HO];
        score(xa, own);
        if (k < deg) {
          put(k, own);
          track(own);
        }
        if (__builtin_amdgcn_ballot_w64(k + 1 < deg) == 0ull) break;
        c1 = k + 3 < deg ? p.colidx[e0 + k + 3] : row;
        gather_row(Xb, c2, goff, xa);
        score(xb, own);
        if (k + 1 < deg) {
          put(k + 1, own);
          track(own);
        }
      }
    }
    STAMP(2)
    float inv[HO];
#pragma unroll
    for (int o = 0; o < HO; ++o) inv[o] = sm[o] > 0.f ? 1.f / sm[o] : 0.f;
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (k < deg) {
        float v[HO];
#pragma unroll
        for (int o = 0; o < HO; ++o) v[o] = __expf(rawr[k][o] - mx[o]) * inv[o];
        put(k, v);
      }
    if (__builtin_amdgcn_ballot_w64(KR < deg) != 0ull) {
      // (the raw scores' stores are acknowledged before the loads are issued; stored by this very lane, and an agent-scope
      //  load never sees a stale L1 line)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (P >= 2 || half == 0) {
        for (int k = KR; k < deg; k += 4) {
          float raw[4][HO];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int o = 0; o < HO; ++o)
              raw[c][o] = k + c < deg ? __hip_atomic_load(attr + (long long)(k + c) * P + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (k + c < deg) {
              float v[HO];
#pragma unroll
              for (int o = 0; o < HO; ++o) v[o] = __expf(raw[c][o] - mx[o]) * inv[o];
              put(k + c, v);
            }
        }
      }
    }
    STAMP(3)
  }
  STAMP_OUT(p)
}

// ---- kernel B: the hop on X for HP heads of a row group, then y = relu([H0 | H1] [x ; z] + bias) on the matrix cores
template <int HP>
__global__ __launch_bounds__(FUSED_THREADS) void csr_fused_hop_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, half = lane >> 5;
  const int ngrp = p.P / HP;                                     // head groups
  const int m0 = blockIdx.x / MAGAT_NUM_XCD, wgs = gridDim.x / MAGAT_NUM_XCD;
  const int hg = m0 % ngrp, mm = m0 / ngrp, wgs_per = wgs / ngrp;
  STAMP_DECL;
  {
    const fu32x4* src = reinterpret_cast<const fu32x4*>(p.wh) + (long long)hg * HP * 4096;      // 64 KB per head
    fu32x4* dst = reinterpret_cast<fu32x4*>(lds);
    for (int i = t; i < HP * 4096; i += FUSED_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  STAMP(6)
  const int N = p.N, ng = (N + 31) >> 5, nchunk = (ng + FUSED_WAVES - 1) / FUSED_WAVES;
  const int items = ((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * nchunk;
  const int goff = 8 * half;
  const char* const frag = lds + lane * 16;
  const int* const order_in = p.order + (long long)p.B * N;
  for (int u = mm, s = 0; u < items; u += wgs_per, ++s) {
    const Item it = fused_item(u, s, wave, nchunk, ng, p.B);
    if (!it.ok) continue;
    const int b = it.b, pos = it.gi * 32 + r;
    const bool valid = pos < N;
    const int row = order_in[(long long)b * N + (valid ? pos : it.gi * 32)];
    const int* cp = p.cscptr + (long long)b * (N + 1);
    const int s0 = cp[row];
    const int deg = valid ? cp[row + 1] - s0 : 0;
    const u16* Xb = p.X + (long long)b * N * 128;
    const float* const attg = p.att + hg * HP;      // att[pos * P + head]
    f32x2 acc[HP][8][4];
#pragma unroll
    for (int h = 0; h < HP; ++h)
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[h][m][d] = f32x2{0.f, 0.f};
    auto load4i = [&](int k0, int (&src)[4], int (&ps)[4]) __attribute__((always_inline)) {
      if (k0 + 4 <= deg) {
        const Int4 a = *reinterpret_cast<const Int4*>(p.cscsrc + s0 + k0);
        const Int4 c = *reinterpret_cast<const Int4*>(p.cscpos + s0 + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          src[i] = a.v[i];
          ps[i] = c.v[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool on = k0 + i < deg;
          src[i] = on ? p.cscsrc[s0 + k0 + i] : row;
          ps[i] = on ? p.cscpos[s0 + k0 + i] : -1;
        }
      }
    };
    auto ldval = [&](int src, int ps, float (&a)[HP], fu32x4 (&x)[8]) __attribute__((always_inline)) {
      if constexpr (HP == 2) {
        f32x2 v = {0.f, 0.f};
        if (ps >= 0) v = *reinterpret_cast<const f32x2*>(attg + (long long)ps * p.P);
        a[0] = v[0];
        a[1] = v[1];
      } else {
        a[0] = ps >= 0 ? attg[(long long)ps * p.P] : 0.f;
      }
      gather_row(Xb, src, goff, x);
    };
    auto fma_slot = [&](const float (&a)[HP], const fu32x4 (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const f32x2 xv = {__builtin_bit_cast(float, x[m][d] << 16), __builtin_bit_cast(float, x[m][d] & 0xffff0000u)};
#pragma unroll
          for (int h = 0; h < HP; ++h) acc[h][m][d] = __builtin_elementwise_fma(f32x2{a[h], a[h]}, xv, acc[h][m][d]);
        }
    };
    {
      int si[4], sp[4], ni[4], np4[4];
      float aa[HP], ab[HP];
      fu32x4 xa[8], xb[8];
      load4i(0, si, sp);
      ldval(si[0], sp[0], aa, xa);
#ifdef FUSED_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(0)
#endif
      for (int k0 = 0;; k0 += 4) {
        if (__builtin_amdgcn_ballot_w64(k0 < deg) == 0ull) break;
        load4i(k0 + 4, ni, np4);
        ldval(si[1], sp[1], ab, xb);
        fma_slot(aa, xa);
        if (__builtin_amdgcn_ballot_w64(k0 + 1 < deg) == 0ull) break;
        ldval(si[2], sp[2], aa, xa);
        fma_slot(ab, xb);
        if (__builtin_amdgcn_ballot_w64(k0 + 2 < deg) == 0ull) break;
        ldval(si[3], sp[3], ab, xb);
        fma_slot(aa, xa);
        if (__builtin_amdgcn_ballot_w64(k0 + 3 < deg) == 0ull) break;
        ldval(ni[0], np4[0], aa, xa);
        fma_slot(ab, xb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          si[i] = ni[i];
          sp[i] = np4[i];
        }
      }
    }
    STAMP(1)
    // operands: the row itself (k = 0..127) and the hop's result rounded to bf16 (k = 128..255) - both already in operand layout
    fu32x4 xo[8];
    gather_row(Xb, row, goff, xo);
#ifdef FUSED_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(2)
#endif
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      fu32x4 z[8];
#pragma unroll
      for (int m = 0; m < 8; ++m)
        z[m] = fu32x4{pk_bf16(acc[h][m][0][0], acc[h][m][0][1]), pk_bf16(acc[h][m][1][0], acc[h][m][1][1]),
                      pk_bf16(acc[h][m][2][0], acc[h][m][2][1]), pk_bf16(acc[h][m][3][0], acc[h][m][3][1])};
      const int head = hg * HP + h;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        // (the bias quads of this tile, requested in front of its matrix products: read behind them they were eight exposed L2
        //  round trips per step - the tap phase took 22 k cycles per step for 8 k of matrix work)
        f32x4 bqv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bqv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias) bqv[j] = *reinterpret_cast<const f32x4*>(p.bias + 32 * mt + 8 * j + 4 * half);
        }
        f32x16 y = {};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          y = mfma_bf16(*reinterpret_cast<const fu32x4*>(frag + ((h * 4 + mt) * 16 + ks) * 1024), xo[ks], y);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          y = mfma_bf16(*reinterpret_cast<const fu32x4*>(frag + ((h * 4 + mt) * 16 + 8 + ks) * 1024), z[ks], y);
#ifdef FUSED_STAMPS
        { float sink = y[0]; asm volatile("" :: "v"(sink)); }      // (the chain has finished when its first result is readable)
        STAMP(4)
#endif
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float w = y[4 * j + i] + bqv[j][i];
            if (p.act_relu) w = magat_relu(w);
            v[4 * j + i] = w;
          }
        }
        fu32x4 g[2];
        acc_to_granules(v, g);
        if (p.y_f32) {
          if (valid) {
            const long long off = ((long long)b * N + row) * p.ldy + head * 128 + 32 * mt + 8 * half;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              float* yp = static_cast<float*>(p.Y) + off + 16 * jj;
              *reinterpret_cast<f32x4*>(yp) =
                  f32x4{__builtin_bit_cast(float, g[jj][0] << 16), __builtin_bit_cast(float, g[jj][0] & 0xffff0000u),
                        __builtin_bit_cast(float, g[jj][1] << 16), __builtin_bit_cast(float, g[jj][1] & 0xffff0000u)};
              *reinterpret_cast<f32x4*>(yp + 4) =
                  f32x4{__builtin_bit_cast(float, g[jj][2] << 16), __builtin_bit_cast(float, g[jj][2] & 0xffff0000u),
                        __builtin_bit_cast(float, g[jj][3] << 16), __builtin_bit_cast(float, g[jj][3] & 0xffff0000u)};
            }
          }
        } else {
          // bf16 rows through a wave-private 4 KB stage (two column tiles = 128 bytes of 32 rows): a lane's 16-byte pieces are
          // 32 rows apart in memory (64 separate pieces per store instruction, 1024 per step); transposed, eight lanes write a
          // row's 128-byte run.  16-byte units XOR-swizzled by the row pair: conflict-free both ways.  (-2.5 % of the kernel.)
          char* const stg = lds + HP * 65536 + wave * 4096;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int c = 4 * (mt & 1) + 2 * jj + half;
            *reinterpret_cast<fu32x4*>(stg + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = g[jj];
          }
          if (mt & 1) {
            const int rowv = valid ? row : -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int rr = 8 * i + (lane >> 3), ch = lane & 7;
              const fu32x4 val = *reinterpret_cast<const fu32x4*>(stg + rr * 128 + ((ch ^ ((rr >> 1) & 7)) << 4));
              const int rrow = __shfl(rowv, rr, 64);
              if (rrow >= 0)
                *reinterpret_cast<fu32x4*>(static_cast<u16*>(p.Y) + ((long long)b * N + rrow) * p.ldy + head * 128 + 32 * (mt - 1) +
                                           8 * ch) = val;
            }
          }
        }
        STAMP(5)
      }
    }
    STAMP(3)
  }
  STAMP_OUT(p)
}

// weights as MFMA fragments (RNE bf16 of the packed float32 block Bt [NC][128], NC = P 128 + P 2 128): lane l of a fragment
// holds operand row l % 32, k = 16 ks + 8 (l / 32) + 0..7
__global__ void csr_fused_pack_kernel(const float* __restrict__ Bt, u16* __restrict__ out, int P) {
  const int nq = P * 16384, total = P * 49152;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7, l = (idx >> 3) & 63;
    float v;
    if (idx < nq) {             // q'[h'] = sum_g W_p[g][h'] x[g]:  operand row h', k = g;  Bt row (p, g) = W_p[g][:]
      const int ks = (idx >> 9) & 7, ft = (idx >> 12) & 3, hp = idx >> 14;
      v = Bt[(long long)(hp * 128 + 16 * ks + 8 * (l >> 5) + e) * 128 + 32 * ft + (l & 31)];
    } else {                    // y[c] = sum_f H_p0[c][f] x[f] + H_p1[c][f] z[f]:  operand row c, k = tap * 128 + f
      const int j = idx - nq;
      const int ks = (j >> 9) & 15, mt = (j >> 13) & 3, hp = j >> 15;
      const int k = 16 * ks + 8 * (l >> 5) + e;
      v = Bt[(long long)(P * 128 + (hp * 2 + (k >> 7)) * 128 + 32 * mt + (l & 31)) * 128 + (k & 127)];
    }
    out[idx] = magat_bf16_rne(v);
  }
}

// attention [nnz][P] (the kernels' own order) -> [P][nnz] (what magat_gat_forward_*'s att_opt hands to the caller)
__global__ void csr_fused_att_out_kernel(const float* __restrict__ a, float* __restrict__ out, long long nnz, int P) {
  const long long total = nnz * P;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long e = idx / P;
    const int h = (int)(idx - e * P);
    out[(long long)h * nnz + e] = a[idx];
  }
}

int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

}  // namespace

#ifdef FUSED_STAMPS
static long long* g_fused_dbg = nullptr;
extern "C" int magat_csr_fused_set_debug(long long* buf) { g_fused_dbg = buf; return MAGAT_OK; }
#endif

int magat_gat_csr_fused_supported(int G, int F, int K, int P, int mode, int concat) {
  return magat_opt(MAGAT_OPT_CSR_FUSED) && mode == MAGAT_MODE_KEYQUERY && K == 2 && G == 128 && F == 128 && concat &&
         (P == 1 || P == 2 || P == 4);
}

int magat_gat_csr_fused_pack(const float* Bt, void* frag_out, int P, hipStream_t st) {
  hipLaunchKernelGGL(csr_fused_pack_kernel, dim3(192), dim3(256), 0, st, Bt, static_cast<u16*>(frag_out), P);
  return magat_check_launch();
}

size_t magat_gat_csr_fused_order_bytes(int B, int N) { return (size_t)2 * B * N * sizeof(int); }

int magat_gat_csr_fused_forward(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr, const int* cscsrc,
                                const int* cscpos, long long nnz, const void* frags, const float* bias, void* Y, int ldy,
                                int y_f32, float* att, float* att_opt, int* order, int B, int N, int P, hipStream_t st) {
  if (!(P == 1 || P == 2 || P == 4)) return MAGAT_ERR_UNSUPPORTED;
  if ((ldy & (y_f32 ? 3 : 7)) || (reinterpret_cast<uintptr_t>(Y) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
    return MAGAT_ERR_UNSUPPORTED;
  FusedParams p = {};
  p.X = X; p.rowptr = rowptr; p.colidx = colidx; p.cscptr = cscptr; p.cscsrc = cscsrc; p.cscpos = cscpos;
  p.order = order; p.att = att;
  p.wq = static_cast<const u16*>(frags);
  p.wh = p.wq + (size_t)P * 16384;
  p.bias = bias; p.Y = Y; p.ldy = ldy; p.y_f32 = y_f32; p.act_relu = 1;
  p.B = B; p.N = N; p.P = P; p.nnz = nnz;
#ifdef FUSED_STAMPS
  p.dbg = g_fused_dbg;
#endif
  {
    const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);      // (a launch of the layer: counted with its graph kernels)
    hipLaunchKernelGGL(csr_rank_kernel, dim3(B, 2), dim3(1024), 0, st, rowptr, cscptr, order, B, N);
    magat_prof_end(pid, st);
    if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  }
  const int per_xcd = device_cus() / MAGAT_NUM_XCD > 0 ? device_cus() / MAGAT_NUM_XCD : 1;
  const int ng = (N + 31) / 32, nchunk = (ng + FUSED_WAVES - 1) / FUSED_WAVES;
  const int items = ((B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * nchunk;
  {
    const int wgx = items < per_xcd ? items : per_xcd;
    const size_t lds = (size_t)P * 32768;
    const void* fn = P == 4 ? reinterpret_cast<const void*>(&csr_fused_scores_kernel<4>)
                   : P == 2 ? reinterpret_cast<const void*>(&csr_fused_scores_kernel<2>)
                            : reinterpret_cast<const void*>(&csr_fused_scores_kernel<1>);
    if (magat_ensure_dyn_lds(fn, MAGAT_LDS_CSR_FUSED_A + (P == 4 ? 2 : P == 2 ? 1 : 0), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
    const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
    const dim3 grid(MAGAT_NUM_XCD * wgx), block(FUSED_THREADS);
    if (P == 4) hipLaunchKernelGGL((csr_fused_scores_kernel<4>), grid, block, lds, st, p);
    else if (P == 2) hipLaunchKernelGGL((csr_fused_scores_kernel<2>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((csr_fused_scores_kernel<1>), grid, block, lds, st, p);
    magat_prof_end(pid, st);
    if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  }
  {
#ifdef FUSED_STAMPS
    if (p.dbg) p.dbg += 8 * FUSED_WAVES * 4096;      // (the hop kernel's stamps behind the score kernel's)
#endif
    const int HP = P >= 2 ? 2 : 1, ngrp = P / HP;
    int per = per_xcd / ngrp;
    if (per < 1) per = 1;
    const int wgx = ngrp * (items < per ? items : per);
    const size_t lds = (size_t)HP * 65536 + (size_t)FUSED_WAVES * 4096;      // weights + the row-store stage of every wave
    const void* fn = HP == 2 ? reinterpret_cast<const void*>(&csr_fused_hop_kernel<2>)
                             : reinterpret_cast<const void*>(&csr_fused_hop_kernel<1>);
    if (magat_ensure_dyn_lds(fn, MAGAT_LDS_CSR_FUSED_B + (HP == 2 ? 1 : 0), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
    const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
    const dim3 grid(MAGAT_NUM_XCD * wgx), block(FUSED_THREADS);
    if (HP == 2) hipLaunchKernelGGL((csr_fused_hop_kernel<2>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((csr_fused_hop_kernel<1>), grid, block, lds, st, p);
    magat_prof_end(pid, st);
    if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  }
  if (att_opt && nnz > 0) {
    long long blocks = (nnz * P + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(csr_fused_att_out_kernel, dim3((unsigned)blocks), dim3(256), 0, st, att, att_opt, nnz, P);
    if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  }
  magat_form_note(MAGAT_FORM_CSR_FUSED);
  return MAGAT_OK;
}
