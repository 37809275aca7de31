// GraphFilterBatchAttentional.forward for LARGE / SPARSE graphs (CSR GSO, any N): BASELINE config 5
// (1000 agents, comm-radius graph).  Same algebra as gat_f32.hip (reference graphML.py:4636-4671, 1724-1827,
// 1180-1286, 713-823) with the GSO given as its edge structure:
//   rowptr [B*(N+1)] absolute offsets into colidx, colidx[e] = j for the e-th edge i -> j of row i
//   (edge (i,j) present  <=>  |S[b,i,j]| > 1e-9 in the dense form).
// Neighbour rows no longer fit LDS (N*G*4 = 512 KB at N=1000), so the kernels gather feature rows straight
// from global memory with 512-byte coalesced row reads that hit the XCD's L2 (all heads of an instance are
// mapped to one XCD); the per-row work mirrors the dense kernel: 8-lane dot products + DPP reductions for the
// scores, wave-uniform scalar loops for the hops.
//   1. csr_transpose_kernel : CSC view (in-edges of every node, sorted by source, with the CSR position of each
//                             edge) - needed because the reference aggregates over COLUMNS of the row-softmax
//                             (x @ aij, graphML.py:1757)
//   2. Z = X @ [W_p | H_pk]^T on fp32 MFMA (conv_gemm_f32.hip)
//   3. csr_scores_kernel    : e_ij, row softmax -> att[p][e]   (CSR order)
//   4. csr_hop_kernel (K-1x): T <- U_k + A^T T, last one fused with bias / ReLU / concat store
#include <cstdlib>

#include "magat_common.h"

namespace {

typedef unsigned short u16;

// Storage type of the node-feature tensors (X, Z, hop intermediates, Y): float, or bf16 (u16) for the bf16-storage
// variant of BASELINE config 5.  Arithmetic is fp32 either way; attention values stay fp32.
template <typename ST> __device__ __forceinline__ float to_f32(ST v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<u16>(u16 v) { return magat_bf16_f32(v); }
template <typename ST> __device__ __forceinline__ ST from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ u16 from_f32<u16>(float v) { return magat_bf16_rne(v); }
template <typename ST, int V>
struct __attribute__((aligned(sizeof(ST) * V))) Pack {
  ST v[V];
};
template <int V, typename ST>
__device__ __forceinline__ void load_vec(const ST* ptr, float (&o)[V]) {
  const Pack<ST, V> r = *reinterpret_cast<const Pack<ST, V>*>(ptr);
#pragma unroll
  for (int c = 0; c < V; ++c) o[c] = to_f32<ST>(r.v[c]);
}
template <int V, typename ST>
__device__ __forceinline__ void store_vec(ST* ptr, const float (&o)[V]) {
  Pack<ST, V> r;
#pragma unroll
  for (int c = 0; c < V; ++c) r.v[c] = from_f32<ST>(o[c]);
  *reinterpret_cast<Pack<ST, V>*>(ptr) = r;
}

struct CsrParams {
  const void* X;       // [B*N, G]   (ST)
  const void* Z;       // [B*N, NC]  (ST)
  const int* rowptr;   // [B*(N+1)]
  const int* colidx;   // [nnz]
  const int* cscptr;   // [B*(N+1)] (workspace)
  const int* cscsrc;   // [nnz] source node i of each in-edge
  const int* cscpos;   // [nnz] position of that edge in CSR order
  float* att;          // [P][nnz] attention values in CSR order
  const void* Told;    // hop input rows (ST)  (row = (b*N+i), head offset applied by caller)
  void* Tnew;          // (ST)
  const float* bias;
  void* Y;             // (ST)
  int B, N, K, P, mode, concat;
  int NC, qoff, uoff, c1off, c2off, ldy;
  long long nnz;
  int k;               // hop index (U_k added)
  long long told_off;              // element offset of the first Told row inside its buffer
  int told_ld, told_head_stride;   // addressing of Told rows: Told + told_off + (b*N+i)*told_ld + head*told_head_stride
  int last;
  int act_relu;           // apply ReLU in the last hop's store (inference); 0 in training (autograd owns it)
  int y_f32;              // bf16 storage only: Y is a float32 buffer - the last store widens the bf16-ROUNDED result (the values
                          // a bf16 Y would hold; saves the caller's cast kernel)
};

// the layer's result rows: ST, or (bf16 storage, y_f32) the same rounded values as float32
template <int V, typename ST>
__device__ __forceinline__ void store_y(void* Y, int y_f32, long long off, const float (&o)[V]) {
  if (sizeof(ST) == 2 && y_f32) {
    float r[V];
#pragma unroll
    for (int c = 0; c < V; ++c) r[c] = __builtin_bit_cast(float, (unsigned)magat_bf16_rne(o[c]) << 16);
    store_vec<V, float>(static_cast<float*>(Y) + off, r);
  } else {
    store_vec<V, ST>(static_cast<ST*>(Y) + off, o);
  }
}

// ---- 1. CSR -> CSC (per instance), deterministic: counting sort + per-column insertion sort by source
__global__ __launch_bounds__(256) void csr_transpose_kernel(const int* __restrict__ rowptr,
                                                            const int* __restrict__ colidx, int* __restrict__ cscptr,
                                                            int* __restrict__ csctmp, int N) {
  extern __shared__ int cnt[];          // [N+1] counts -> offsets, then [N] cursors
  int* cur = cnt + N + 1;
  const int b = blockIdx.x, t = threadIdx.x;
  const int* rp = rowptr + (long long)b * (N + 1);
  const int e0 = rp[0], e1 = rp[N];
  for (int j = t; j <= N; j += 256) cnt[j] = 0;
  __syncthreads();
  for (int e = e0 + t; e < e1; e += 256) atomicAdd(&cnt[colidx[e] + 1], 1);
  __syncthreads();
  if (t < 64) {                         // exclusive scan by one wave
    int carry = 0;
    for (int base = 0; base <= N; base += 64) {
      const int j = base + t;
      int v = j <= N ? cnt[j] : 0, inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (t >= o) inc += u;
      }
      if (j <= N) cnt[j] = carry + inc;
      carry += __shfl(inc, 63, 64);
    }
  }
  __syncthreads();
  int* cp = cscptr + (long long)b * (N + 1);
  for (int j = t; j <= N; j += 256) cp[j] = e0 + cnt[j];
  for (int j = t; j < N; j += 256) cur[j] = cnt[j];
  __syncthreads();
  for (int e = e0 + t; e < e1; e += 256) {      // unordered parallel fill (slot order depends on timing) ...
    const int slot = e0 + atomicAdd(&cur[colidx[e]], 1);
    csctmp[slot] = e;
  }
}

// ... which csr_sort_columns_kernel turns into a deterministic order: one wave per column rank-sorts the column's
// CSR positions (= source rows ascending) and looks the source row of each in-edge up by bisection in rowptr.
__global__ __launch_bounds__(256) void csr_sort_columns_kernel(const int* __restrict__ rowptr,
                                                               const int* __restrict__ cscptr,
                                                               const int* __restrict__ csctmp, int* __restrict__ cscsrc,
                                                               int* __restrict__ cscpos, int N, long long cols) {
  const int lane = threadIdx.x & 63;
  const long long col = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (col >= cols) return;
  const int b = (int)(col / N), j = (int)(col % N);
  const int* rp = rowptr + (long long)b * (N + 1);
  const int* cp = cscptr + (long long)b * (N + 1);
  const int a = cp[j], d = cp[j + 1] - a;
  for (int t = lane; t < d; t += 64) {
    const int v = csctmp[a + t];
    int rank = 0;
    for (int u = 0; u < d; ++u) rank += csctmp[a + u] < v ? 1 : 0;
    int lo = 0, hi = N - 1;                      // last row i with rp[i] <= v
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (rp[mid] <= v) lo = mid; else hi = mid - 1;
    }
    cscpos[a + rank] = v;
    cscsrc[a + rank] = lo;
  }
}

// ---- 3. scores + row softmax.  8 lanes per row, 32 rows per 256-thread block.
template <int G, typename ST>
__global__ __launch_bounds__(256) void csr_scores_kernel(const CsrParams p) {
  constexpr int GC = G / 4, CP8 = GC / 8 > 0 ? GC / 8 : 1, LE = GC < 8 ? GC : 8;
  const int N = p.N;
  const int rows_per_block = 256 / LE;
  const int tiles = (N + rows_per_block - 1) / rows_per_block;
  // block -> (instance, head, row tile): heads and tiles of an instance stay on one XCD
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per = p.P * tiles;
  const int b = xcd + MAGAT_NUM_XCD * (slot / per);
  if (b >= p.B) return;
  const int head = (slot % per) / tiles, tile = slot % tiles;
  const int t = threadIdx.x, es = t % LE, i = tile * rows_per_block + t / LE;
  if (i >= N) return;
  const int* rp = p.rowptr + (long long)b * (N + 1);
  const int e0 = rp[i], e1 = rp[i + 1];
  if (e1 <= e0) return;
  const ST* Zb = static_cast<const ST*>(p.Z) + (long long)b * N * p.NC;
  float* att = p.att + (long long)head * p.nnz;
  // Row softmax without cross-lane memory traffic: the raw scores are written by one lane and later rewritten
  // by that same lane (program order makes its own stores visible to it); max and sum are carried online in
  // registers, identically in every lane of the group.
  float mx = -__builtin_inff(), sum = 0.f;
  auto online = [&](float d) {
    const float m2 = fmaxf(mx, d);
    sum = sum * __expf(mx - m2) + __expf(d - m2);
    mx = m2;
  };
  if (p.mode == MAGAT_MODE_KEYQUERY) {
    const ST* xr = static_cast<const ST*>(p.X) + ((long long)b * N + i) * G;
    float xi[CP8][4];
#pragma unroll
    for (int q = 0; q < CP8; ++q) load_vec<4, ST>(xr + 4 * (es + LE * q), xi[q]);
    const int qo = p.qoff + head * G;
    for (int e = e0; e < e1; e += 2) {
      const int j0 = p.colidx[e];
      const bool two = e + 1 < e1;
      const int j1 = two ? p.colidx[e + 1] : j0;
      const ST* q0 = Zb + (long long)j0 * p.NC + qo;
      const ST* q1 = Zb + (long long)j1 * p.NC + qo;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int q = 0; q < CP8; ++q) {
        float a0[4], a1[4];
        load_vec<4, ST>(q0 + 4 * (es + LE * q), a0);
        load_vec<4, ST>(q1 + 4 * (es + LE * q), a1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          d0 = fmaf(xi[q][c], a0[c], d0);
          d1 = fmaf(xi[q][c], a1[c], d1);
        }
      }
      if (LE == 8) {
        d0 = oct_sum(d0);
        d1 = oct_sum(d1);
      } else {
#pragma unroll
        for (int o = LE / 2; o > 0; o >>= 1) {
          d0 += __shfl_xor(d0, o, 64);
          d1 += __shfl_xor(d1, o, 64);
        }
      }
      if (es == 0) {
        att[e] = d0;
        if (two) att[e + 1] = d1;
      }
      online(d0);
      if (two) online(d1);
    }
    if (es == 0) {
      const float inv = 1.f / sum;
      for (int e = e0; e < e1; ++e) att[e] = __expf(att[e] - mx) * inv;
    }
  } else {
    const float c2 = to_f32<ST>(Zb[(long long)i * p.NC + p.c2off + head]);
    for (int e = e0 + es; e < e1; e += LE) {
      const float v = to_f32<ST>(Zb[(long long)p.colidx[e] * p.NC + p.c1off + head]) + c2;
      const float l = v > 0.f ? v : 0.2f * v;
      att[e] = l;
      online(l);
    }
    // combine the lanes' (max, sum) pairs
    float gm = mx;
    if (LE == 8) gm = oct_max(gm);
    else
#pragma unroll
      for (int o = LE / 2; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
    float gs = sum > 0.f ? sum * __expf(mx - gm) : 0.f;
    if (LE == 8) gs = oct_sum(gs);
    else
#pragma unroll
      for (int o = LE / 2; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
    const float inv = 1.f / gs;
    for (int e = e0 + es; e < e1; e += LE) att[e] = __expf(att[e] - gm) * inv;
  }
}

// ---- 4. one Horner hop: out[j] = U_k[j] + sum_{in-edges (i -> j)} att[pos] * Told[i]; wave per output row
template <int F, typename ST>
__global__ __launch_bounds__(256) void csr_hop_kernel(const CsrParams p) {
  constexpr int VEC = F >= 64 ? F / 64 : 1;
  constexpr int LANES = F >= 64 ? 64 : F;
  const int N = p.N;
  const int tiles = (N + 3) / 4;         // 4 rows (waves) per block
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int per = p.P * tiles;
  const int b = xcd + MAGAT_NUM_XCD * (slot / per);
  if (b >= p.B) return;
  const int head = (slot % per) / tiles, tile = slot % tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = tile * 4 + wave;
  if (j >= N || lane >= LANES) return;
  const int* cp = p.cscptr + (long long)b * (N + 1);
  const int s0 = cp[j], s1 = cp[j + 1];
  const float* att = p.att + (long long)head * p.nnz;
  const ST* Tb = static_cast<const ST*>(p.Told) + p.told_off + (long long)b * N * p.told_ld +
                 (long long)head * p.told_head_stride + VEC * lane;
  const ST* Zr = static_cast<const ST*>(p.Z) + ((long long)b * N + j) * p.NC + p.uoff + (head * p.K + p.k) * F + VEC * lane;
  float acc[VEC];
  load_vec<VEC, ST>(Zr, acc);
  for (int s = s0; s < s1; s += 2) {
    const int i0 = p.cscsrc[s];
    const float a0 = att[p.cscpos[s]];
    float t0[VEC];
    load_vec<VEC, ST>(Tb + (long long)i0 * p.told_ld, t0);
    if (s + 1 < s1) {
      const int i1 = p.cscsrc[s + 1];
      const float a1 = att[p.cscpos[s + 1]];
      float t1[VEC];
      load_vec<VEC, ST>(Tb + (long long)i1 * p.told_ld, t1);
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a1, t1[c], fmaf(a0, t0[c], acc[c]));
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a0, t0[c], acc[c]);
    }
  }
  if (p.last) {
    if (p.bias) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] += p.bias[VEC * lane + c];
    }
    if (p.act_relu) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] = magat_relu(acc[c]);
    }
    store_y<VEC, ST>(p.Y, p.y_f32, ((long long)b * N + j) * p.ldy + head * F + VEC * lane, acc);
  } else {
    store_vec<VEC, ST>(static_cast<ST*>(p.Tnew) + (((long long)b * N + j) * p.P + head) * F + VEC * lane, acc);
  }
}

// K == 1: Y = U_0 + bias (no graph work)
template <int F, typename ST>
__global__ void csr_k1_kernel(const CsrParams p) {
  const long long total = (long long)p.B * p.N * p.P * (F / 4);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % (F / 4));
    const long long r = idx / (F / 4);
    const int head = (int)(r % p.P);
    const long long m = r / p.P;
    float v[4];
    load_vec<4, ST>(static_cast<const ST*>(p.Z) + m * p.NC + p.uoff + head * F + 4 * c, v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (p.bias) v[q] += p.bias[4 * c + q];
      if (p.act_relu) v[q] = magat_relu(v[q]);
    }
    store_y<4, ST>(p.Y, p.y_f32, m * p.ldy + head * F + 4 * c, v);
  }
}

struct Layout {
  int NC, qoff, uoff, c1off, c2off;
};
Layout layout(int G, int F, int K, int P, int mode) {   // must match pack_layout() in gat_f32.hip
  Layout L;
  if (mode == MAGAT_MODE_KEYQUERY) {
    L.qoff = 0; L.uoff = P * G; L.c1off = L.c2off = 0; L.NC = P * G + P * K * F;
  } else if (mode == MAGAT_MODE_GNN) {
    L.qoff = 0; L.uoff = 0; L.c1off = L.c2off = 0; L.NC = (P * K * F + 31) & ~31;
  } else {
    L.qoff = 0; L.uoff = 0; L.c1off = P * K * F; L.c2off = L.c1off + P; L.NC = (L.c2off + P + 31) & ~31;
  }
  return L;
}

struct WsLayout {
  size_t status, z, cscptr, cscsrc, cscpos, csctmp, att, t0, t1, ytmp, order, total;
};
// the form with the maps inside the graph kernels (gat_csr_fused.hip): bf16 storage + the column view of magat_gso_csr_build
bool csr_fused_form(size_t esz, bool have_csc, int G, int F, int K, int P, int mode, int concat) {
  return esz == 2 && have_csc && magat_gat_csr_fused_supported(G, F, K, P, mode, concat);
}
WsLayout ws_layout(int B, int N, long long nnz, int G, int F, int K, int P, int mode, int concat,
                   size_t esz = sizeof(float), bool have_csc = false) {
  const Layout L = layout(G, F, K, P, mode);
  const bool fused = csr_fused_form(esz, have_csc, G, F, K, P, mode, concat);
  WsLayout w;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += magat_align_up(bytes, 256); return at; };
  w.status = take(256);              // range-guard status words of the maps GEMM (first bytes of the workspace)
  w.z = take(fused ? 0 : (size_t)B * N * L.NC * esz);      // (the fused form has no maps in memory)
  // (the column view: nothing is reserved for it when the caller brings the one magat_gso_csr_build made)
  w.cscptr = take(have_csc ? 0 : (size_t)B * (N + 1) * sizeof(int));
  w.cscsrc = take(have_csc ? 0 : (size_t)nnz * sizeof(int));
  w.cscpos = take(have_csc ? 0 : (size_t)nnz * sizeof(int));
  w.csctmp = take(have_csc ? 0 : (size_t)nnz * sizeof(int));
  w.att = take((size_t)P * nnz * sizeof(float));
  const size_t tb = K > 2 ? (size_t)B * N * P * F * esz : 0;
  w.t0 = take(tb);
  w.t1 = take(K > 3 ? tb : 0);
  w.ytmp = take(concat ? 0 : (size_t)B * N * P * F * esz);
  w.order = take(fused ? magat_gat_csr_fused_order_bytes(B, N) : 0);
  w.total = o;
  return w;
}

// ---- LDS-tiled forms of the score and hop kernels for N <= 1024 (BASELINE config 5: N = 1000) ------------------------------
// The per-edge gathers above read every neighbour row from L2 (256-512 B per edge and head: 3.6 GB per c5 step, the
// kernels sat at 13 % of the HBM roof).  Here one workgroup owns an (instance, head) pair and walks the feature axis in
// passes of 128 BYTES per row (32 fp32 or 64 bf16 features): the pass's [N][128 B] slice of the gathered tensor (Q_p for
// the scores, the hop input for a hop) is streamed into LDS once (LDS-direct loads), and every edge then gathers its
// 16 bytes per lane from LDS.  8 lanes per graph row, 8 rows per wave step.  Partial scores of the passes are accumulated
// in att[] by the lane that owns the edge (edge k of a row belongs to lane k & 7 of the row's group: same-thread
// read-after-write, always coherent); owners prefetch their previous partials as one batch before the edge loop, and the
// value enters the 8-lane reduction of the dot product as an extra term of its owner - no dependent load in the loop.
// LPR = lanes per graph row (8: 128-byte slices, one 1024-thread workgroup per CU at N = 1000; 4: 64-byte slices, 512-thread
// workgroups, two per CU - one streams its slice in while the other computes).  32 edges of a row take the batched path.
constexpr int TILED_EDGES = 32;
template <int LPR>
__device__ __forceinline__ float grp_sum(float d) {
  if (LPR == 8) return oct_sum(d);
  d += __shfl_xor(d, 1, 64);
  d += __shfl_xor(d, 2, 64);
  return d;
}

template <typename ST, int LPR>
__device__ __forceinline__ void tile_dma(char* tile, const ST* src_base, long long row_stride_elems, int N, int t) {
  // rows n = 0..N-1, 16 LPR bytes each from src_base + n * row_stride: one wave instruction moves 64 / LPR rows (1 KB)
  constexpr int RPI = 64 / LPR;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = (int)(blockDim.x >> 6);
  for (int g = wave; g * RPI < N; g += nw) {
    const int n = g * RPI + lane / LPR;
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile + (unsigned)g * 1024u);
    if (n < N) {
      const char* src = reinterpret_cast<const char*>(src_base + (long long)n * row_stride_elems) + (lane & (LPR - 1)) * 16;
      asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    }
  }
}

typedef unsigned tv_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 tv_bf2 __attribute__((ext_vector_type(2)));
template <typename ST> struct TileVec;            // the 16 bytes a lane owns of a tile row, as floats
template <> struct TileVec<float> {
  static constexpr int E = 4;
  static __device__ __forceinline__ float dot(const unsigned (&a)[4], const unsigned (&b)[4]) {     // raw 16-byte pieces
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) d = fmaf(__builtin_bit_cast(float, a[c]), __builtin_bit_cast(float, b[c]), d);
    return d;
  }
  static __device__ __forceinline__ void load(const void* p, float (&o)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
};
template <> struct TileVec<u16> {
  static constexpr int E = 8;
  static __device__ __forceinline__ float dot(const unsigned (&a)[4], const unsigned (&b)[4]) {   // v_dot2c_f32_bf16: exact
    float d = 0.f;                                                                               // products, fp32 accumulation
#pragma unroll
    for (int c = 0; c < 4; ++c)
      d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tv_bf2, a[c]), __builtin_bit_cast(tv_bf2, b[c]), d, false);
    return d;
  }
  static __device__ __forceinline__ void load(const void* p, float (&o)[8]) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = *reinterpret_cast<const u32x4_*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[2 * q] = __builtin_bit_cast(float, v[q] << 16);
      o[2 * q + 1] = __builtin_bit_cast(float, v[q] & 0xffff0000u);
    }
  }
};

// The walk order of an instance's rows: RANKED BY EDGE COUNT (descending), so that the 8 rows of a wave step have (almost) the same
// number of edges - a step runs as many edge slots as its longest row has, and in index order that was ~9.5 slots for the 5.4
// edges per row of config 5.  Counting sort in LDS behind the tile (one thread per row; the order inside a bin is whatever the
// LDS atomics make it - no result depends on it: a row's arithmetic does not know its position), and the (row, first edge,
// edge count) of EVERY position this wave walks - at most 8 steps of 8 rows (N <= 1024, 16 waves) - lands in one register
// triple per lane for the whole kernel: lane 8 s + g holds step s, group g.  (Read inside the step's prefetch, the row pointers
// were a second dependent round trip in front of the column and score loads of every step - round 5.)
constexpr int TILED_SCRATCH = 13 * 1024;      // histogram [64] | rows [1024] | first edges [1024] | edge counts [1024]
__device__ __forceinline__ void tiled_row_order(char* scratch, const int* __restrict__ ptr, int N, int t, int q, int& rowh,
                                                int& e0h, int& degh) {
  int* hist = reinterpret_cast<int*>(scratch);
  int* ord = reinterpret_cast<int*>(scratch + 1024);
  const int lane = t & 63;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  int e0 = 0, d = 0, key = 0, slot = 0;
  if (t < N) {
    e0 = ptr[t];
    d = ptr[t + 1] - e0;
    key = 63 - (d < 63 ? d : 63);
    slot = atomicAdd(&hist[key], 1);
  }
  __syncthreads();
  const int c = hist[lane];          // (every wave scans the 64 bins for itself)
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int base = __shfl(incl - c, key, 64);
  if (t < N) {
    ord[base + slot] = t;
    ord[1024 + base + slot] = e0;
    ord[2048 + base + slot] = d;
  }
  __syncthreads();
  const bool ok = q < N;
  rowh = ok ? ord[q] : 0;
  e0h = ok ? ord[1024 + q] : 0;
  degh = ok ? ord[2048 + q] : 0;
}

template <typename ST, int LPR>
__global__ __launch_bounds__(1024) void csr_tiled_scores_kernel(const CsrParams p, int G) {
  extern __shared__ __attribute__((aligned(1024))) char tile[];
  constexpr int E = TileVec<ST>::E;                 // features per lane and pass
  constexpr int TILE_ROW_BYTES = 16 * LPR, RPS = 64 / LPR;          // bytes of a row slice; graph rows per wave step
  constexpr int FPP = TILE_ROW_BYTES / (int)sizeof(ST);   // features per pass
  constexpr int R = TILED_EDGES / LPR;
  const int N = p.N, NP = G / FPP;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int b = xcd + MAGAT_NUM_XCD * (slot / p.P), head = slot % p.P;
  if (b >= p.B) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, es = lane & (LPR - 1), eg = lane / LPR, gbase = lane & ~(LPR - 1);
  const int rstep = RPS * (int)(blockDim.x >> 6);
  const int* rp = p.rowptr + (long long)b * (N + 1);
  const ST* Zb = static_cast<const ST*>(p.Z) + (long long)b * N * p.NC + p.qoff + head * G;
  const ST* Xb = static_cast<const ST*>(p.X) + (long long)b * N * G;
  float* att = p.att + (long long)head * p.nnz;
  static_assert(LPR == 8, "lane 8 s + g <-> (row step s, row group g)");
  int rowh, e0h, degh;
  tile_dma<ST, LPR>(tile, Zb, p.NC, N, t);      // the first pass's slice streams in while the rows are ranked (scratch: behind the tile)
  tiled_row_order(tile + (size_t)((N + RPS - 1) / RPS) * 1024, rp, N, t, RPS * wave + (lane >> 3) * rstep + (lane & 7), rowh, e0h, degh);
  // everything a row step needs from global memory, requested one step ahead (the loop body then only touches LDS)
  struct Row {
    int e0, deg;
    bool ok;
    unsigned xv[4];          // the lane's 16 bytes of x_i, raw
    int cj[R];
    float pre[R];
  };
  for (int h = 0; h < NP; ++h) {
    const bool first = h == 0, last = h == NP - 1;
    auto fetch = [&](int ib, int sidx, Row& w) {
      w.ok = ib + eg < N;
      const int ir = __shfl(rowh, 8 * sidx + eg, 64);
      w.e0 = __shfl(e0h, 8 * sidx + eg, 64);
      w.deg = __shfl(degh, 8 * sidx + eg, 64);
      {
        const tv_u32x4 v = *reinterpret_cast<const tv_u32x4*>(Xb + (long long)ir * G + h * FPP + es * E);
        w.xv[0] = v[0]; w.xv[1] = v[1]; w.xv[2] = v[2]; w.xv[3] = v[3];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        w.cj[r] = 0;
        w.pre[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        // (rows ranked by edge count: past the first batch of 8 edges most wave steps have nothing to ask for)
        if (r > 0 && __builtin_amdgcn_ballot_w64(LPR * r < w.deg) == 0ull) break;
        const bool mine_ok = es + LPR * r < w.deg;
        w.cj[r] = mine_ok ? p.colidx[w.e0 + es + LPR * r] : 0;
        // (agent-scope load: served by L2.  The value was stored by THIS thread in the previous pass, but a plain load may
        //  hit the copy of the line this CU's L1 still holds from the pass before that)
        w.pre[r] = (!first && mine_ok)
                       ? __hip_atomic_load(att + w.e0 + es + LPR * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
      }
    };
    Row cur, nxt;
    if (RPS * wave < N) fetch(RPS * wave, 0, cur);        // in flight together with the slice
    if (h > 0) {
      __syncthreads();                                // every wave is done with the previous slice
#ifndef CSR_WHATIF_NODMA      // (timing experiments, tools/csr_layer_bench.py: wrong results)
      tile_dma<ST, LPR>(tile, Zb + h * FPP, p.NC, N, t);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef CSR_WHATIF_NOLOOP
    if (p.N > 0) continue;
#endif
    for (int ib = RPS * wave, sidx = 0; ib < N; ib += rstep, ++sidx) {
      if (ib + rstep < N) fetch(ib + rstep, sidx + 1, nxt);
      const int e0 = cur.e0, deg = cur.deg;
      float mine[R];
#pragma unroll
      for (int r = 0; r < R; ++r) mine[r] = 0.f;
      float mx = -__builtin_inff(), sum = 0.f;
      auto online = [&](float d) {
        const float m2 = fmaxf(mx, d);
        sum = sum * __expf(mx - m2) + __expf(d - m2);
        mx = m2;
      };
      auto edge_dot = [&](int j) -> float {
        const tv_u32x4 v = *reinterpret_cast<const tv_u32x4*>(tile + j * TILE_ROW_BYTES + es * 16);
        const unsigned q[4] = {v[0], v[1], v[2], v[3]};
        return TileVec<ST>::dot(cur.xv, q);
      };
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (__builtin_amdgcn_ballot_w64(LPR * r < deg) == 0ull) break;
#pragma unroll
        for (int k = 0; k < LPR; k += 2) {
          // (no row of this wave step has an edge left: at 5.4 edges per row - config 5 - the longest of 8 rows has ~9, and a
          //  full batch of 8 slots for it ran 6-7 empty ones)
          if (k > 0 && __builtin_amdgcn_ballot_w64(LPR * r + k < deg) == 0ull) break;
          const int ka = LPR * r + k, kb = ka + 1;
          const bool va = ka < deg, vb = kb < deg;
          // neighbour index of edge k from its owner lane (past the row's degree: row 0 of the slice, never used)
          const int ja = __shfl(cur.cj[r], gbase + k, 64), jb = __shfl(cur.cj[r], gbase + k + 1, 64);
          float da = edge_dot(ja), db = edge_dot(jb);
          if (es == k) da += cur.pre[r];
          if (es == k + 1) db += cur.pre[r];
          da = grp_sum<LPR>(da);
          db = grp_sum<LPR>(db);
          if (es == k && va) mine[r] = da;
          if (es == k + 1 && vb) mine[r] = db;
        }
      }
      // last pass: the row's softmax over the batched edges from their OWNERS' registers - one maximum and one sum over the
      // 8-lane group (an owner has R candidates) instead of an online update per edge in every lane (two exponentials and a
      // dependent chain per edge: the loop is bound by its vector instructions, not by latency - tools/csr_layer_bench.py)
      if (last) {
        float m = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (es + LPR * r < deg) m = fmaxf(m, mine[r]);
        m = LPR == 8 ? oct_max(m) : fmaxf(fmaxf(m, __shfl_xor(m, 1, 64)), fmaxf(__shfl_xor(m, 2, 64), __shfl_xor(m, 3, 64)));
        float sl = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (es + LPR * r < deg) sl += __expf(mine[r] - m);
        sum = grp_sum<LPR>(sl);
        mx = m;                                           // (rows without edges: -inf and 0, as the online form leaves them)
      }
      // rows with more than 8 R edges: the rest one by one, owner = lane 0 (dependent loads: rare)
      for (int k = LPR * R; k < deg; ++k) {
        float d = edge_dot(p.colidx[e0 + k]);
        if (es == 0 && !first) d += __hip_atomic_load(att + e0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d = grp_sum<LPR>(d);
        if (es == 0) att[e0 + k] = d;
        if (last) online(d);
      }
      if (!last) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (es + LPR * r < deg) att[e0 + es + LPR * r] = mine[r];
      } else {
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (es + LPR * r < deg) att[e0 + es + LPR * r] = __expf(mine[r] - mx) * inv;
        if (es == 0)
          for (int k = LPR * R; k < deg; ++k)
            att[e0 + k] = __expf(__hip_atomic_load(att + e0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - mx) * inv;
      }
      cur = nxt;
    }
  }
}

// one Horner hop out[j] = U_k[j] + sum over in-edges (i -> j) of att * Told[i], Told slice in LDS
template <typename ST, int LPR>
__global__ __launch_bounds__(1024) void csr_tiled_hop_kernel(const CsrParams p, int F) {
  extern __shared__ __attribute__((aligned(1024))) char tile[];
  constexpr int E = TileVec<ST>::E;
  constexpr int TILE_ROW_BYTES = 16 * LPR, RPS = 64 / LPR;
  constexpr int FPP = TILE_ROW_BYTES / (int)sizeof(ST);
  constexpr int R = TILED_EDGES / LPR;
  const int N = p.N, NP = F / FPP;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int b = xcd + MAGAT_NUM_XCD * (slot / p.P), head = slot % p.P;
  if (b >= p.B) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, es = lane & (LPR - 1), eg = lane / LPR, gbase = lane & ~(LPR - 1);
  const int rstep = RPS * (int)(blockDim.x >> 6);
  const int* cp = p.cscptr + (long long)b * (N + 1);
  const float* att = p.att + (long long)head * p.nnz;
  const ST* Tb = static_cast<const ST*>(p.Told) + p.told_off + (long long)b * N * p.told_ld +
                 (long long)head * p.told_head_stride;
  const ST* Ub = static_cast<const ST*>(p.Z) + (long long)b * N * p.NC + p.uoff + (head * p.K + p.k) * F;
  // rows ranked by IN-edge count, their column pointers in registers for the whole kernel (see the score kernel)
  static_assert(LPR == 8, "lane 8 s + g <-> (row step s, row group g)");
  int rowh, s0h, degh;
  tile_dma<ST, LPR>(tile, Tb, p.told_ld, N, t);      // (the first slice under the ranking, as in the score kernel)
  tiled_row_order(tile + (size_t)((N + RPS - 1) / RPS) * 1024, cp, N, t, RPS * wave + (lane >> 3) * rstep + (lane & 7), rowh, s0h, degh);
  // A step's global reads in TWO stages, each requested a whole step before it is needed: the in-edge lists (source row, CSR
  // position) two steps ahead, the weights att[position] - which depend on them - and the U row one step ahead.
  struct Idx {
    int s0, deg, row;
    bool ok;
    int src[R], pos[R];
  };
  struct Val {
    float acc[E];
    float wgt[R];
  };
  for (int h = 0; h < NP; ++h) {
    auto fetch_idx = [&](int jb, int sidx, Idx& w) {
      w.ok = jb + eg < N;
      w.row = __shfl(rowh, 8 * sidx + eg, 64);
      w.s0 = __shfl(s0h, 8 * sidx + eg, 64);
      w.deg = __shfl(degh, 8 * sidx + eg, 64);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        w.src[r] = 0;
        w.pos[r] = -1;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r > 0 && __builtin_amdgcn_ballot_w64(LPR * r < w.deg) == 0ull) break;
        const bool mine_ok = es + LPR * r < w.deg;
#ifdef CSR_WHATIF_NOIDX
        w.src[r] = es + LPR * r; w.pos[r] = mine_ok ? w.s0 + es + LPR * r : -1;
#else
        w.src[r] = mine_ok ? p.cscsrc[w.s0 + es + LPR * r] : 0;
        w.pos[r] = mine_ok ? p.cscpos[w.s0 + es + LPR * r] : -1;
#endif
      }
    };
    auto fetch_val = [&](int jb, const Idx& ix, Val& w) {
      TileVec<ST>::load(Ub + (long long)ix.row * p.NC + h * FPP + es * E, w.acc);
#pragma unroll
      for (int r = 0; r < R; ++r) w.wgt[r] = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r > 0 && __builtin_amdgcn_ballot_w64(LPR * r < ix.deg) == 0ull) break;
        w.wgt[r] = ix.pos[r] >= 0 ? att[ix.pos[r]] : 0.f;
      }
    };
    Idx cur, nxt, nx2;
    Val cv, nv;
    float bv[E];                 // this lane's bias columns of the pass
#pragma unroll
    for (int c = 0; c < E; ++c) bv[c] = (p.last && p.bias) ? p.bias[h * FPP + es * E + c] : 0.f;
    const int jb0 = RPS * wave;
    if (jb0 < N) {
      fetch_idx(jb0, 0, cur);
      if (jb0 + rstep < N) fetch_idx(jb0 + rstep, 1, nxt);
      fetch_val(jb0, cur, cv);
    }
    if (h > 0) {
      __syncthreads();
#ifndef CSR_WHATIF_NODMA
      tile_dma<ST, LPR>(tile, Tb + h * FPP, p.told_ld, N, t);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef CSR_WHATIF_NOLOOP
    if (p.N > 0) continue;
#endif
    for (int jb = jb0, sidx = 0; jb < N; jb += rstep, ++sidx) {
      if (jb + 2 * rstep < N) fetch_idx(jb + 2 * rstep, sidx + 2, nx2);
      if (jb + rstep < N) fetch_val(jb + rstep, nxt, nv);
      const int j = cur.row, deg = cur.deg;
      float acc[E];
#pragma unroll
      for (int c = 0; c < E; ++c) acc[c] = cv.acc[c];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#ifdef CSR_WHATIF_NOKLOOP
        break;
#endif
        if (__builtin_amdgcn_ballot_w64(LPR * r < deg) == 0ull) break;
#pragma unroll
        for (int k = 0; k < LPR; ++k) {
          if (k > 0 && __builtin_amdgcn_ballot_w64(LPR * r + k < deg) == 0ull) break;      // (as in the score loop)
          // (i, a) of edge LPR r + k from its owner lane; edges past the row's degree carry weight 0 and row 0
          const int i = __shfl(cur.src[r], gbase + k, 64);
          const float a = __shfl(cv.wgt[r], gbase + k, 64);
          float tv[E];
          TileVec<ST>::load(tile + i * TILE_ROW_BYTES + es * 16, tv);
#pragma unroll
          for (int c = 0; c < E; ++c) acc[c] = fmaf(a, tv[c], acc[c]);
        }
      }
      for (int k = LPR * R; k < deg; ++k) {
        const int i = p.cscsrc[cur.s0 + k];
        const float a = att[p.cscpos[cur.s0 + k]];
        float tv[E];
        TileVec<ST>::load(tile + i * TILE_ROW_BYTES + es * 16, tv);
#pragma unroll
        for (int c = 0; c < E; ++c) acc[c] = fmaf(a, tv[c], acc[c]);
      }
#ifdef CSR_WHATIF_NOSTORE
      if (acc[0] == 1.2345f && acc[3] == 7.f)
#endif
      if (cur.ok) {
        const int col = h * FPP + es * E;
        if (p.last) {
          if (p.bias) {
#pragma unroll
            for (int c = 0; c < E; ++c) acc[c] += bv[c];
          }
          if (p.act_relu) {
#pragma unroll
            for (int c = 0; c < E; ++c) acc[c] = magat_relu(acc[c]);
          }
          store_y<E, ST>(p.Y, p.y_f32, ((long long)b * N + j) * p.ldy + head * F + col, acc);
        } else {
          store_vec<E, ST>(static_cast<ST*>(p.Tnew) + (((long long)b * N + j) * p.P + head) * F + col, acc);
        }
      }
      cur = nxt; nxt = nx2; cv = nv;
    }
  }
}

template <typename ST>
bool csr_tiled_ok(const CsrParams& p, int width, int which) {      // which: 1 = scores, 2 = hop (option CSR_TILED = bit mask)
  return (magat_opt(MAGAT_OPT_CSR_TILED) & which) && p.N <= 1024 && p.N >= 8 && (width * (int)sizeof(ST)) % 128 == 0;
}
template <typename ST, int LPR>
int launch_tiled(const CsrParams& p, int width, bool scores, hipStream_t st) {
  constexpr int RPS = 64 / LPR;
  const long long grid = (long long)((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * MAGAT_NUM_XCD * p.P;
  if (grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const size_t lds = (size_t)((p.N + RPS - 1) / RPS) * 1024 + TILED_SCRATCH;
  const void* fn = scores ? reinterpret_cast<const void*>(&csr_tiled_scores_kernel<ST, LPR>)
                          : reinterpret_cast<const void*>(&csr_tiled_hop_kernel<ST, LPR>);
  const int slot = (scores ? MAGAT_LDS_CSR_TILED_A : MAGAT_LDS_CSR_TILED_B) + (sizeof(ST) == 2 ? 2 : 0) + (LPR == 4 ? 4 : 0);
  if (magat_ensure_dyn_lds(fn, slot, lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  const int threads = LPR == 4 ? 512 : 1024;
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
  if (scores)
    hipLaunchKernelGGL((csr_tiled_scores_kernel<ST, LPR>), dim3((unsigned)grid), dim3(threads), lds, st, p, width);
  else
    hipLaunchKernelGGL((csr_tiled_hop_kernel<ST, LPR>), dim3((unsigned)grid), dim3(threads), lds, st, p, width);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
template <typename ST>
int run_tiled(const CsrParams& p, int width, bool scores, hipStream_t st) {
  // (the 64-byte-slice form - 512-thread workgroups, two per CU, option CSR_TILED bit 2 of rounds 2-4 - doubled the passes and
  //  their per-row bookkeeping and measured 27 % slower: removed in round 5)
  return launch_tiled<ST, 8>(p, width, scores, st);
}

template <int G, typename ST = float>
int run_scores(const CsrParams& p, hipStream_t st) {
  constexpr int LE = (G / 4) < 8 ? (G / 4) : 8;
  const int rows = 256 / LE, tiles = (p.N + rows - 1) / rows;
  const long long grid = (long long)((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * MAGAT_NUM_XCD * p.P * tiles;
  if (grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
  hipLaunchKernelGGL((csr_scores_kernel<G, ST>), dim3((unsigned)grid), dim3(256), 0, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
template <int F, typename ST = float>
int run_hop(const CsrParams& p, hipStream_t st) {
  const int tiles = (p.N + 3) / 4;
  const long long grid = (long long)((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * MAGAT_NUM_XCD * p.P * tiles;
  if (grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
  hipLaunchKernelGGL((csr_hop_kernel<F, ST>), dim3((unsigned)grid), dim3(256), 0, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}
template <int F, typename ST = float>
int run_k1(const CsrParams& p, hipStream_t st) {
  hipLaunchKernelGGL((csr_k1_kernel<F, ST>), dim3(2048), dim3(256), 0, st, p);
  return magat_check_launch();
}

template <typename ST>
__global__ void head_mean_relu_csr_kernel(const ST* __restrict__ ytmp, void* __restrict__ y, long long M, int P,
                                          int F, int ldy, int y_f32) {
  const int FC = F / 4;
  const long long total = M * FC;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long m = idx / FC;
    const int c = (int)(idx - m * FC);
    float s[4], u[4];
    load_vec<4, ST>(ytmp + m * (long long)P * F + 4 * c, s);
    for (int q = 1; q < P; ++q) {
      load_vec<4, ST>(ytmp + (m * P + q) * (long long)F + 4 * c, u);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += u[e];
    }
    const float fp = (float)P;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = magat_relu(s[e] / fp);
    store_y<4, ST>(y, y_f32, m * ldy + 4 * c, s);
  }
}



#define MAGAT_CSR_DISPATCH(WIDTH, FN, ST)                   \
  switch (WIDTH) {                                         \
    case 16: rc = FN<16, ST>(p, st); break;                \
    case 32: rc = FN<32, ST>(p, st); break;                \
    case 64: rc = FN<64, ST>(p, st); break;                \
    case 128: rc = FN<128, ST>(p, st); break;              \
    default: rc = FN<256, ST>(p, st);                      \
  }

// maps GEMM Z = X @ Bt^T + colbias in the storage type: fp32 (fp32 MFMA / bf16x6 split) or bf16 in, bf16 out
// (one bf16 MFMA product per element pair, fp32 accumulate; weights = plane 0 of the packed bf16x3 block)
template <typename ST>
int csr_maps_gemm(const ST* X, const float* packed, ST* Z, int M, int G, const Layout& L, void* stream, int32_t* status);
template <>
int csr_maps_gemm<float>(const float* X, const float* packed, float* Z, int M, int G, const Layout& L, void* stream,
                         int32_t* status) {
  return magat_gat_maps_gemm(X, packed, Z, M, G, L.NC, L.NC, stream, 0, status);
}
template <>
int csr_maps_gemm<u16>(const u16* X, const float* packed, u16* Z, int M, int G, const Layout& L, void* stream, int32_t*) {
  if ((L.NC % 32) || (G % 32)) return MAGAT_ERR_UNSUPPORTED;
  magat_conv_gemm_desc d = {};
  d.in = reinterpret_cast<const float*>(X);
  d.wt = packed + (((size_t)L.NC * (G + 1) + 3) & ~(size_t)3);     // bf16x3 planes; plane 0 = RNE bf16 of Bt
  d.bias = packed + (size_t)L.NC * G;
  d.out = reinterpret_cast<float*>(Z);
  d.M = M; d.Cin = G; d.lda = G; d.Hin = d.Win = 1; d.kH = d.kW = 1; d.stride = 1; d.Hout = d.Wout = 1;
  d.Cout = L.NC; d.ldc = L.NC; d.tag = MAGAT_TAG_GAT_MAPS; d.in_fmt = 3; d.out_fmt = 2;
  return magat_conv_gemm_f32(&d, stream);
}

template <typename ST>
int csr_forward(const ST* X, const int* rowptr, const int* colidx, long long nnz, const float* packed,
                const float* bias, void* Y, int ldy, float* att_opt, void* workspace, size_t workspace_bytes, int B,
                int N, int G, int F, int K, int P, int mode, int concat, void* stream,
                const float* edge_vals = nullptr, const int* pre_cscptr = nullptr, const int* pre_cscsrc = nullptr,
                const int* pre_cscpos = nullptr, int y_f32 = 0) {
  if (!X || !rowptr || !packed || !Y || (nnz > 0 && !colidx)) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || nnz < 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (mode < MAGAT_MODE_KEYQUERY || mode > MAGAT_MODE_GNN) return MAGAT_ERR_UNSUPPORTED;
  const bool gnn = mode == MAGAT_MODE_GNN;     // fixed edge weights (the GSO values) instead of attention
  if (gnn && (P != 1 || (nnz > 0 && K > 1 && !edge_vals) || (G & 3))) return MAGAT_ERR_BAD_SHAPE;
  if ((!gnn && G != F) || !(F == 16 || F == 32 || F == 64 || F == 128 || F == 256)) return MAGAT_ERR_UNSUPPORTED;
  if ((size_t)(2 * N + 2) * sizeof(int) > 64 * 1024) return MAGAT_ERR_UNSUPPORTED;   // transpose LDS (N <= 8190)
  const int width = concat ? P * F : F;
  if (ldy < width || (ldy & 3)) return MAGAT_ERR_BAD_SHAPE;
  const bool have_csc = pre_cscptr && pre_cscsrc && pre_cscpos;     // made by magat_gso_csr_build (once per GSO)
  const WsLayout w = ws_layout(B, N, nnz, G, F, K, P, mode, concat, sizeof(ST), have_csc);
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < w.total)
    return MAGAT_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const Layout L = layout(G, F, K, P, mode);
  ST* Z = reinterpret_cast<ST*>(ws + w.z);
  int* cscptr = have_csc ? const_cast<int*>(pre_cscptr) : reinterpret_cast<int*>(ws + w.cscptr);
  int* cscsrc = have_csc ? const_cast<int*>(pre_cscsrc) : reinterpret_cast<int*>(ws + w.cscsrc);
  int* cscpos = have_csc ? const_cast<int*>(pre_cscpos) : reinterpret_cast<int*>(ws + w.cscpos);
  float* att = gnn ? const_cast<float*>(edge_vals) : (att_opt ? att_opt : reinterpret_cast<float*>(ws + w.att));
  ST* tbuf[2] = {reinterpret_cast<ST*>(ws + w.t0), reinterpret_cast<ST*>(ws + w.t1)};
  ST* Ytmp = reinterpret_cast<ST*>(ws + w.ytmp);

  if (csr_fused_form(sizeof(ST), have_csc, G, F, K, P, mode, concat)) {
    // maps on the matrix cores inside the score / hop kernels: Q and U never exist in memory (gat_csr_fused.hip)
    return magat_gat_csr_fused_forward(reinterpret_cast<const uint16_t*>(X), rowptr, colidx, cscptr, cscsrc, cscpos, nnz,
                                                packed + magat_gat_csr_fused_offset(L.NC, G), bias, Y, ldy, y_f32,
                                                reinterpret_cast<float*>(ws + w.att), att_opt,
                                                reinterpret_cast<int*>(ws + w.order), B, N, P, st);
    // (MAGAT_ERR_UNSUPPORTED: result rows / bias not 16-byte aligned)
  }
  int rc = csr_maps_gemm<ST>(X, packed, Z, B * N, G, L, stream, reinterpret_cast<int32_t*>(ws + w.status));
  if (rc != MAGAT_OK) return rc;

  CsrParams p = {};
  p.X = X; p.Z = Z; p.rowptr = rowptr; p.colidx = colidx; p.cscptr = cscptr; p.cscsrc = cscsrc; p.cscpos = cscpos;
  p.att = att; p.bias = bias; p.Y = concat ? Y : static_cast<void*>(Ytmp); p.ldy = concat ? ldy : P * F;
  p.y_f32 = concat ? y_f32 : 0;         // (the per-head rows of the mean form stay in the storage type)
  p.B = B; p.N = N; p.K = K; p.P = P; p.mode = mode; p.concat = concat; p.nnz = nnz;
  p.act_relu = gnn ? 0 : concat;        // GraphFilterBatch has no nonlinearity of its own
  p.NC = L.NC; p.qoff = L.qoff; p.uoff = L.uoff; p.c1off = L.c1off; p.c2off = L.c2off;

  if (K == 1 && (!att_opt || gnn)) {
    MAGAT_CSR_DISPATCH(F, run_k1, ST)
    if (rc != MAGAT_OK) return rc;
  } else {
    if (K > 1 && !have_csc) {
      const int pid = magat_prof_begin(MAGAT_TAG_GSO_CSR, st);
      int* csctmp = reinterpret_cast<int*>(ws + w.csctmp);
      hipLaunchKernelGGL(csr_transpose_kernel, dim3(B), dim3(256), (size_t)(2 * N + 2) * sizeof(int), st, rowptr, colidx,
                         cscptr, csctmp, N);
      const long long cols = (long long)B * N;
      hipLaunchKernelGGL(csr_sort_columns_kernel, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, st, rowptr, cscptr,
                         csctmp, cscsrc, cscpos, N, cols);
      magat_prof_end(pid, st);
      if ((rc = magat_check_launch()) != MAGAT_OK) return rc;
    }
    if (!gnn) {
      if (mode == MAGAT_MODE_KEYQUERY && csr_tiled_ok<ST>(p, G, 1)) rc = run_tiled<ST>(p, G, true, st);
      else { MAGAT_CSR_DISPATCH(G, run_scores, ST) }
      if (rc != MAGAT_OK) return rc;
    }
    if (K == 1) {
      MAGAT_CSR_DISPATCH(F, run_k1, ST)
      if (rc != MAGAT_OK) return rc;
    }
    // hops k = K-2 .. 0; the first reads U_{K-1} straight out of Z
    for (int k = K - 2, h = 0; k >= 0; --k, ++h) {
      p.k = k;
      p.last = k == 0;
      if (h == 0) {
        p.Told = Z;                          // row (b*N+i): + i*NC, head: + head*K*F
        p.told_off = L.uoff + (K - 1) * F;
        p.told_ld = L.NC;
        p.told_head_stride = K * F;
      } else {
        p.Told = tbuf[(h - 1) & 1];
        p.told_off = 0;
        p.told_ld = P * F;
        p.told_head_stride = F;
      }
      p.Tnew = tbuf[h & 1];
      if (csr_tiled_ok<ST>(p, F, 2)) rc = run_tiled<ST>(p, F, false, st);
      else { MAGAT_CSR_DISPATCH(F, run_hop, ST) }
      if (rc != MAGAT_OK) return rc;
    }
  }
  if (!concat) {
    const long long M = (long long)B * N;
    long long blocks = (M * (F / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const int pid = magat_prof_begin(MAGAT_TAG_HEAD_MEAN, st);
    hipLaunchKernelGGL((head_mean_relu_csr_kernel<ST>), dim3((unsigned)blocks), dim3(256), 0, st, Ytmp, Y, M, P, F, ldy,
                       y_f32);
    magat_prof_end(pid, st);
    return magat_check_launch();
  }
  return MAGAT_OK;
}

}  // namespace

// float32 <-> bf16 (RNE) row blocks around the bf16-storage layer: src [M][ld_src] -> dst [M][ld_dst], `width` columns
namespace {
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, long long M, int width,
                                     int ld_src, int ld_dst, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;      // (predicated form: behind the encoder's range-guard re-run)
  const int wq = width / 4;
  const long long total = M * wq;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long m = idx / wq;
    const int c = (int)(idx - m * wq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + m * ld_src + c);
    uint2 o;
    o.x = (unsigned)magat_bf16_rne(v[0]) | ((unsigned)magat_bf16_rne(v[1]) << 16);
    o.y = (unsigned)magat_bf16_rne(v[2]) | ((unsigned)magat_bf16_rne(v[3]) << 16);
    *reinterpret_cast<uint2*>(dst + m * ld_dst + c) = o;
  }
}
__global__ void cast_bf16_f32_kernel(const u16* __restrict__ src, float* __restrict__ dst, long long M, int width,
                                     int ld_src, int ld_dst) {
  const int wq = width / 4;
  const long long total = M * wq;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long m = idx / wq;
    const int c = (int)(idx - m * wq) * 4;
    const uint2 v = *reinterpret_cast<const uint2*>(src + m * ld_src + c);
    *reinterpret_cast<f32x4*>(dst + m * ld_dst + c) =
        f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
              __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
  }
}
}  // namespace

int magat_cast_rows_if(const void* src, void* dst, int to_bf16, long long M, int width, int ld_src, int ld_dst, void* stream,
                       const int32_t* run_if);
extern "C" int magat_cast_rows(const void* src, void* dst, int to_bf16, long long M, int width, int ld_src, int ld_dst,
                               void* stream) {
  return magat_cast_rows_if(src, dst, to_bf16, M, width, ld_src, ld_dst, stream, nullptr);
}
// run_if (float32 -> bf16 only): device flag, the launch returns at once when it is zero
int magat_cast_rows_if(const void* src, void* dst, int to_bf16, long long M, int width, int ld_src, int ld_dst, void* stream,
                       const int32_t* run_if) {
  if (!src || !dst) return MAGAT_ERR_NULL;
  if (run_if && !to_bf16) return MAGAT_ERR_UNSUPPORTED;
  if (M <= 0 || width <= 0 || (width & 3) || (ld_src & 3) || (ld_dst & 3) || ld_src < width || ld_dst < width)
    return MAGAT_ERR_BAD_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  long long blocks = (M * (width / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const int pid = magat_prof_begin(run_if ? MAGAT_TAG_UNTAGGED : MAGAT_TAG_GAT_CAST, st);      // (predicated: inside the guard's span)
  if (to_bf16)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)(run_if && blocks > 512 ? 512 : blocks)), dim3(256), 0, st,
                       static_cast<const float*>(src), static_cast<u16*>(dst), M, width, ld_src, ld_dst,
                       reinterpret_cast<const int*>(run_if));
  else
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const u16*>(src),
                       static_cast<float*>(dst), M, width, ld_src, ld_dst);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

extern "C" size_t magat_gat_csr_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode,
                                                int concat) {
  if (B <= 0 || N <= 0 || nnz < 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return 0;
  return ws_layout(B, N, nnz, G, F, K, P, mode, concat).total;
}
extern "C" size_t magat_gat_csr_bf16_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode,
                                                     int concat) {
  if (B <= 0 || N <= 0 || nnz < 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return 0;
  return ws_layout(B, N, nnz, G, F, K, P, mode, concat, sizeof(u16)).total;
}

// workspace of the *_csc_* entry points (the caller brings the column view: no transpose scratch)
extern "C" size_t magat_gat_csc_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode,
                                                int concat, int bf16) {
  if (B <= 0 || N <= 0 || nnz < 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return 0;
  return ws_layout(B, N, nnz, G, F, K, P, mode, concat, bf16 ? sizeof(u16) : sizeof(float), true).total;
}

extern "C" int magat_gat_forward_csr_f32(const float* X, const int* rowptr, const int* colidx, long long nnz,
                                         const float* packed, const float* bias, float* Y, int ldy, float* att_opt,
                                         void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K,
                                         int P, int mode, int concat, void* stream) {
  return csr_forward<float>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, att_opt, workspace, workspace_bytes, B, N, G,
                            F, K, P, mode, concat, stream);
}

extern "C" int magat_gnn_forward_csr_f32(const float* X, const int* rowptr, const int* colidx, const float* vals,
                                         long long nnz, const float* packed, const float* bias, float* Y, int ldy,
                                         void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K,
                                         void* stream) {
  return csr_forward<float>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, nullptr, workspace, workspace_bytes, B, N, G,
                            F, K, 1, MAGAT_MODE_GNN, 1, stream, vals);
}

extern "C" int magat_gat_forward_csr_bf16(const uint16_t* X, const int* rowptr, const int* colidx, long long nnz,
                                          const float* packed, const float* bias, uint16_t* Y, int ldy,
                                          float* att_opt, void* workspace, size_t workspace_bytes, int B, int N, int G,
                                          int F, int K, int P, int mode, int concat, void* stream) {
  return csr_forward<u16>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, att_opt, workspace, workspace_bytes, B, N, G, F,
                          K, P, mode, concat, stream);
}

// Dense GSO -> CSR edge structure (|S| > 1e-9), two calls: count (rowptr via caller-side prefix) is avoided by
// writing per-row degrees first.  deg [B*N] ints.
// edge rule of the reference: |S| > 1e-9 in S's dtype; rule 1 (GAT_origin) |float(S) + delta_ij| > 1e-9f; rule 2
// (GraphFilterBatch multiplies by the values): float(S) != 0
template <typename T>
__device__ __forceinline__ bool gso_edge(T v, bool diag, int self_loops) {
  if (self_loops == 2) return (float)v != 0.f;                          // GraphFilterBatch: every non-zero float(S)
  if (self_loops) return fabsf((float)v + (diag ? 1.f : 0.f)) > 1e-9f;
  return (v < 0 ? -v : v) > (T)1e-9;
}

template <typename T>
__global__ void gso_row_degree_kernel(const T* __restrict__ S, int* __restrict__ deg, int N, long long rows,
                                      int self_loops) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* r = S + row * N;
  const int i = (int)(row % N);
  int c = 0;
  for (int j = lane; j < N; j += 64) c += gso_edge(r[j], j == i, self_loops) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) deg[row] = c;
}
template <typename T>
__global__ void gso_fill_csr_kernel(const T* __restrict__ S, const int* __restrict__ rowstart,
                                    int* __restrict__ colidx, int N, long long rows, int self_loops) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* r = S + row * N;
  const int i = (int)(row % N);
  int base = rowstart[row];
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    bool f = false;
    if (j < N) f = gso_edge(r[j], j == i, self_loops);
    const unsigned long long m = __ballot(f);
    if (f) colidx[base + __popcll(m & ((1ull << lane) - 1ull))] = j;
    base += __popcll(m);
  }
}

extern "C" int magat_gso_row_degrees(const void* S, int s_is_f64, int self_loops, int* deg, int B, int N,
                                     void* stream) {
  if (!S || !deg) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0) return MAGAT_ERR_BAD_SHAPE;
  const long long rows = (long long)B * N;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (s_is_f64)
    hipLaunchKernelGGL(gso_row_degree_kernel<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double*>(S),
                       deg, N, rows, self_loops);
  else
    hipLaunchKernelGGL(gso_row_degree_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(S), deg,
                       N, rows, self_loops);
  return magat_check_launch();
}

extern "C" int magat_gso_fill_csr(const void* S, int s_is_f64, int self_loops, const int* rowstart, int* colidx, int B,
                                  int N, void* stream) {
  if (!S || !rowstart || !colidx) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0) return MAGAT_ERR_BAD_SHAPE;
  const long long rows = (long long)B * N;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (s_is_f64)
    hipLaunchKernelGGL(gso_fill_csr_kernel<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double*>(S),
                       rowstart, colidx, N, rows, self_loops);
  else
    hipLaunchKernelGGL(gso_fill_csr_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(S),
                       rowstart, colidx, N, rows, self_loops);
  return magat_check_launch();
}

namespace {
// ---- GSO -> CSR + CSC in ONE pass over the dense tensor (magat_gso_csr_build) ------------------------------------------
// The dense (B,N,N) GSO of a large instance is big (4 MB per instance at N = 1000): addGSO's in-place scrub, the edge test
// and the row degrees are one streaming read of it (written back only where a value changes), leaving a bit matrix
// (N^2/8 bytes per instance).  A second kernel - one workgroup per instance, a thread per row of the bit matrix - then produces
// everything the CSR kernels need, deterministically (no global atomics; LDS atomics only hand out slots whose order a rank
// placement undoes): rowptr / colidx (ascending j per row), cscptr,
// and per in-edge its source row and CSR position (ascending i per column).
constexpr int GSO_W64_MAX = 16;          // N <= 1024

template <typename T>
__global__ __launch_bounds__(256) void gso_mask_kernel(T* __restrict__ S, unsigned long long* __restrict__ masks,
                                                       int* __restrict__ inst_tot, int N, int W64, long long rows,
                                                       int scrub_nan, int gso_mode, int rule) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  // The row is READ through one pointer and (rarely: addGSO's scrub) WRITTEN through another the compiler cannot relate to it:
  // with the conditional write-back in the same loop as the loads of the same array, every load waited for the store in front
  // of it (the plain row-degree kernel streams the same bytes at 4.5 TB/s, this loop ran at 2.6).  A lane only ever re-writes
  // the element it has just read itself, so no ordering between the two is needed.
  const T* __restrict__ rl = S + row * N;
  T* rs = S + row * N;
  asm volatile("" : "+v"(rs));
  const int i = (int)(row % N);
  int cnt = 0;
  unsigned long long mine = 0ull;      // word `lane` of the row's bit mask: ONE 128-byte store per row behind the loop
  // GSO_MASK_BATCH 64-column steps per batch: the loads of a batch go out before the first value is looked at (one load -> wait
  // -> test per step left the latency to the occupancy alone)
#ifndef GSO_MASK_BATCH
#define GSO_MASK_BATCH 4
#endif
  for (int w0 = 0; w0 < W64; w0 += GSO_MASK_BATCH) {
    T xv[GSO_MASK_BATCH];
#pragma unroll
    for (int u = 0; u < GSO_MASK_BATCH; ++u) {
      const int j = (w0 + u) * 64 + lane;
      xv[u] = (w0 + u < W64 && j < N) ? rl[j] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < GSO_MASK_BATCH; ++u) {
      const int w = w0 + u, j = w * 64 + lane;
      bool f = false;
      if (w < W64 && j < N) {
        T x = xv[u];
        bool dirty = false;
        if (scrub_nan && x != x) { x = (T)0; dirty = true; }
        if (gso_mode == 1 && x > (T)0 && x != (T)1) { x = (T)1; dirty = true; }
        if (dirty) rs[j] = x;
        f = gso_edge(x, j == i, rule);
      }
      const unsigned long long m = __ballot(f);
      cnt += __popcll(m);
      if (lane == w) mine = m;
    }
  }
  if (lane < W64) masks[row * W64 + lane] = mine;
  if (lane == 0) inst_tot[row] = cnt;      // per-ROW degree (summed per instance by gso_totals_kernel: 128 k same-line atomics
}                                          // serialised on one L2 channel took 1 ms)

#ifndef MAGAT_GSO_X4
#define MAGAT_GSO_X4 1
#endif
// float rows with N % 4 == 0 (rows 16-byte aligned): every lane loads FOUR columns per request and all requests of the row
// (N <= 1024: four) go out before the first value is looked at - 16 KB in flight per wave instead of 1 KB per load of the
// scalar form, which left the pass at the latency x occupancy product (3.9 TB/s at config 5).  A lane's four edge flags are a
// nibble at bit 4 (lane % 16) of the 64-column word its 16-lane row covers: the word is the OR over the row (DPP).
__device__ __forceinline__ unsigned gso_dpp_or(unsigned v, const int ctrl_tag) {
  switch (ctrl_tag) {
    case 0: return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    case 1: return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    case 2: return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);    // row_ror:4
    default: return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   // row_ror:8
  }
}
__global__ __launch_bounds__(256) void gso_mask_x4_kernel(float* __restrict__ S, unsigned long long* __restrict__ masks,
                                                          int* __restrict__ inst_tot, int N, int W64, long long rows,
                                                          int scrub_nan, int gso_mode, int rule) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4* __restrict__ rl = reinterpret_cast<const f32x4*>(S + row * N);
  float* rs = S + row * N;                       // (write-back pointer the compiler cannot relate to the loads: see below)
  asm volatile("" : "+v"(rs));
  const int i = (int)(row % N);
  f32x4 xv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j0 = s * 256 + lane * 4;
    xv[s] = j0 < N ? rl[s * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  unsigned long long mine = 0ull;                // word `lane` of the row's bit mask (lanes 0..15)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j0 = s * 256 + lane * 4;
    unsigned nib = 0;
    if (j0 < N) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = xv[s][c];
        bool dirty = false;
        if (scrub_nan && x != x) { x = 0.f; dirty = true; }
        if (gso_mode == 1 && x > 0.f && x != 1.f) { x = 1.f; dirty = true; }
        if (dirty) rs[j0 + c] = x;
        nib |= gso_edge(x, j0 + c == i, rule) ? (1u << c) : 0u;
      }
    }
    const int sh = 4 * (lane & 15);
    unsigned lo = sh < 32 ? nib << sh : 0u, hi = sh >= 32 ? nib << (sh - 32) : 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) { lo = gso_dpp_or(lo, t); hi = gso_dpp_or(hi, t); }
    // every lane of row q = lane / 16 now holds word 4 s + q; lane w < 16 keeps word w: from row w % 4 when w / 4 == s
    const int srcl = 16 * (lane & 3);
    const unsigned glo = (unsigned)__shfl((int)lo, srcl, 64), ghi = (unsigned)__shfl((int)hi, srcl, 64);
    if ((lane >> 2) == s) mine = ((unsigned long long)ghi << 32) | glo;
  }
  int cnt = lane < 16 ? __popcll(mine) : 0;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane < W64) masks[row * W64 + lane] = mine;
  if (lane == 0) inst_tot[row] = cnt;
}

// edge total of every instance: one workgroup per instance over its N row degrees
__global__ __launch_bounds__(256) void gso_totals_kernel(const int* __restrict__ rowdeg, int* __restrict__ inst_tot, int N) {
  __shared__ int part[4];
  const int b = blockIdx.x, t = threadIdx.x;
  int s = 0;
  for (int i = t; i < N; i += 256) s += rowdeg[(size_t)b * N + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) part[t >> 6] = s;
  __syncthreads();
  if (t == 0) inst_tot[b] = part[0] + part[1] + part[2] + part[3];
}

// One workgroup per instance.  Round 5: every phase walks the EDGES - a thread per row over the set bits of ITS row, which it
// holds in registers (16 words; the round-2 kernel kept the whole bit matrix in LDS and counted / filled the columns with a
// thread per column walking all N rows twice: 150-183 us per launch at config 5 with half of the chip's CUs, one per instance,
// tied up beside the encoder for that long).  Column degrees are LDS atomics, an in-edge takes its slot in its column's list
// from an LDS cursor, each column's short list is then sorted by source row (registers), which makes the result the same
// deterministic arrays as before (ascending i per column; ascending j per row comes with the bit order).  The three edge
// arrays are STAGED in LDS (the bit matrix no longer lives there) and leave as whole contiguous runs: written edge by edge,
// the scattered 4-byte stores alone took 70 us.  An instance with more edges than the stage holds writes and sorts through
// global memory instead (reads at agent scope: the entries come from other threads of the workgroup).
// Round 6: the cost per column is BOUNDED whatever the topology.  Staged path: a thread ranks its column's list
// itself only up to GSO_SELF_SORT entries; longer lists (hub columns) are ranked by the whole workgroup, one entry per thread
// (at most stage / 1024 = 12 steps of <= 1024 LDS reads each).  An instance with more edges than the stage holds no longer
// sorts anything: the stage region holds the TRANSPOSED bit matrix instead (LDS atomic ORs, N x N bits = 128 KB at N = 1024),
// and an in-edge's slot is its rank = the set bits of its column below its source row (<= 32 word popcounts) - the arrays are
// written once, in their final order, without a global-memory atomic or a read-back (was: an O(deg^2) selection sort of
// agent-scope loads per column, ~500 k L2 round trips per thread on a dense N = 1000 instance).
constexpr int GSO_SELF_SORT = 32;
constexpr int GSO_STAGE_EDGES = 12 * 1024;      // edges of one instance the LDS stage holds (3 x 4 bytes each: 144 KB; < 2^14)
__global__ __launch_bounds__(1024) void gso_structure_kernel(const unsigned long long* __restrict__ masks,
                                                             int* __restrict__ inst_tot, int* __restrict__ rowptr,
                                                             int* colidx, int* __restrict__ cscptr, int* cscsrc, int* cscpos,
                                                             long long cap, long long* __restrict__ nnz_out, int B, int N,
                                                             int W64, int stage_edges) {
  extern __shared__ __align__(16) int gsi[];
  int* ccnt = gsi;                    // [N] column degrees, then the fill cursors
  int* coff = ccnt + 1024;            // [N] exclusive column offsets (local)
  int* stage = coff + 1024;           // [3][stage_edges]: column of CSR position q | in-edge of slot k as (source row << 14 |
                                      // local CSR position), in arrival order | the same, every column's list sorted
                                      // (stage_edges = min(GSO_STAGE_EDGES, N * N): small graphs ask for little LDS)
  __shared__ int part[17];
  __shared__ int nbig, big[GSO_STAGE_EDGES / GSO_SELF_SORT];      // columns with more than GSO_SELF_SORT in-edges (staged path)
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // absolute offset of this instance = edges of all earlier instances (inst_tot was accumulated by the mask pass)
  int before = 0;
  for (int q = t; q < b; q += 1024) before += inst_tot[q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) part[wave] = before;
  // this thread's row of the bit matrix
  unsigned long long mrow[GSO_W64_MAX];
  {
    const unsigned long long* src = masks + ((size_t)b * N + (t < N ? t : 0)) * W64;
#pragma unroll
    for (int w = 0; w < GSO_W64_MAX; ++w) mrow[w] = (t < N && w < W64) ? src[w] : 0ull;
  }
  ccnt[t] = 0;
  if (t == 0) nbig = 0;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < 16; ++w) base += part[w];
  __syncthreads();
  // block-wide exclusive scan helper over up to 1024 values (one per thread)
  auto block_scan = [&](int v, int* out_total) -> int {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += part[w];
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += part[w];
    __syncthreads();
    *out_total = tot;
    return wbase + inc - v;
  };
  // rows: degrees -> offsets; columns: one LDS atomic per edge
  int rdeg = 0;
#pragma unroll
  for (int w = 0; w < GSO_W64_MAX; ++w) {
    unsigned long long m = mrow[w];
    rdeg += __popcll(m);
    while (m) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1;
      atomicAdd(&ccnt[w * 64 + bit], 1);
    }
  }
  int total = 0;
  const int rex = block_scan(rdeg, &total);      // (its barriers also order the atomics above in front of the reads below)
  const int cdeg = t < N ? ccnt[t] : 0;
  int total2 = 0;
  const int cex = block_scan(cdeg, &total2);
  coff[t] = cex;
  ccnt[t] = 0;
  __syncthreads();
  int* rp = rowptr + (size_t)b * (N + 1);
  int* cp = cscptr + (size_t)b * (N + 1);
  if (t < N) {
    rp[t] = base + rex;
    cp[t] = base + cex;
  }
  if (t == 0) {
    rp[N] = base + total;
    cp[N] = base + total;
  }
  if (b == B - 1 && t == 0 && nnz_out) *nnz_out = (long long)base + total;
  const bool staged = total <= stage_edges;
  if (staged) {
    int* const scol = stage;
    int* const sin = stage + stage_edges;
    int* const sout = stage + 2 * stage_edges;
    // a thread per row over its edges in ascending j: the column index of every CSR position, and the in-edge into the next
    // free slot of column j's list
    {
      int pos = rex;
#pragma unroll
      for (int w = 0; w < GSO_W64_MAX; ++w) {
        unsigned long long m = mrow[w];
        while (m) {
          const int bit = __builtin_ctzll(m);
          m &= m - 1;
          const int j = w * 64 + bit;
          scol[pos] = j;
          sin[coff[j] + atomicAdd(&ccnt[j], 1)] = (t << 14) | pos;
          ++pos;
        }
      }
    }
    __syncthreads();
    // a thread per column: its list ordered by source row - every entry goes to the slot its rank names (the lists are short:
    // 3-5 entries at config 5); a hub column is left to the whole workgroup
    if (t < N) {
      if (cdeg <= GSO_SELF_SORT) {
        for (int e = 0; e < cdeg; ++e) {
          const int w = sin[cex + e];
          int rank = 0;
          for (int f = 0; f < cdeg; ++f) rank += sin[cex + f] < w ? 1 : 0;
          sout[cex + rank] = w;
        }
      } else {
        big[atomicAdd(&nbig, 1)] = t;
      }
    }
    __syncthreads();
    for (int q = 0, nb = nbig; q < nb; ++q) {      // (the list's order is whatever the atomics made it: the result does not know)
      const int j = big[q], cx = coff[j], cd = ccnt[j];
      for (int e = t; e < cd; e += 1024) {
        const int w = sin[cx + e];
        int rank = 0;
        for (int f = 0; f < cd; ++f) rank += sin[cx + f] < w ? 1 : 0;
        sout[cx + rank] = w;
      }
    }
    __syncthreads();
    // the stage leaves as three contiguous runs
    const long long lim = cap - base < total ? cap - base : total;
    for (int q = t; q < lim; q += 1024) {
      const int w = sout[q];
      colidx[base + q] = scol[q];
      cscsrc[base + q] = w >> 14;
      cscpos[base + q] = base + (w & 16383);
    }
  } else {
    // more edges than the stage holds: the stage region takes the transposed bit matrix, an in-edge's slot is its rank
    const int W32 = (N + 31) >> 5;
    unsigned* const colm = reinterpret_cast<unsigned*>(stage);      // [N][W32] bit i of column j's words <=> edge (i -> j)
    for (int q = t; q < N * W32; q += 1024) colm[q] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < GSO_W64_MAX; ++w) {
      unsigned long long m = mrow[w];
      while (m) {
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        atomicOr(&colm[(w * 64 + bit) * W32 + (t >> 5)], 1u << (t & 31));
      }
    }
    __syncthreads();
    {
      int pos = base + rex;
#pragma unroll
      for (int w = 0; w < GSO_W64_MAX; ++w) {
        unsigned long long m = mrow[w];
        while (m) {
          const int bit = __builtin_ctzll(m);
          m &= m - 1;
          const int j = w * 64 + bit;
          if (pos < cap) colidx[pos] = j;
          const unsigned* cw = colm + j * W32;
          int rank = __popc(cw[t >> 5] & ((1u << (t & 31)) - 1u));
          for (int v = 0; v < (t >> 5); ++v) rank += __popc(cw[v]);
          const long long k = (long long)base + coff[j] + rank;
          if (k < cap) {
            cscsrc[k] = t;
            cscpos[k] = pos;
          }
          ++pos;
        }
      }
    }
  }
  // leave inst_tot clear for the next build (every workgroup has read what it needs only after ALL of them pass this
  // point is not guaranteed - so the clearing is done by the NEXT build's memset, see the host code)
}

}  // namespace

extern "C" size_t magat_gso_csr_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0 || N > 64 * GSO_W64_MAX) return 0;
  const size_t w64 = (size_t)(N + 63) / 64;
  return magat_align_up((size_t)B * N * w64 * 8, 256) + magat_align_up((size_t)B * sizeof(int), 256) +
         magat_align_up((size_t)B * N * sizeof(int), 256);
}

extern "C" int magat_gso_csr_build_phase(void* S, int s_is_f64, int scrub_nan, int gso_mode, int edge_rule, int* rowptr,
                                         int* colidx, int* cscptr, int* cscsrc, int* cscpos, long long cap,
                                         long long* nnz_dev, void* workspace, size_t workspace_bytes, int B, int N,
                                         int phase, void* stream) {
  if (!S || !rowptr || !colidx || !cscptr || !cscsrc || !cscpos) return MAGAT_ERR_NULL;
  if (phase < 0 || phase > 2) return MAGAT_ERR_BAD_SHAPE;
  if (B <= 0 || N <= 0 || cap < 0 || edge_rule < 0 || edge_rule > 2 || gso_mode < 0 || gso_mode > 1)
    return MAGAT_ERR_BAD_SHAPE;
  const size_t need = magat_gso_csr_workspace_bytes(B, N);
  if (!need) return MAGAT_ERR_UNSUPPORTED;                      // N > 1024: magat_gso_row_degrees / magat_gso_fill_csr
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < need) return MAGAT_ERR_WORKSPACE;
  const int W64 = (N + 63) / 64;
  // column counters / offsets + the edge stage, sized by the graph (a small N asks for little LDS: several instances per CU);
  // the stage must also hold the transposed bit matrix of an instance with more edges than it stages (N > 110 only)
  const long long nn = (long long)N * N;
  const int stage_edges = (int)(nn < GSO_STAGE_EDGES ? nn : GSO_STAGE_EDGES);
  size_t stage_ints = (size_t)3 * stage_edges;
  if (nn > GSO_STAGE_EDGES && (size_t)N * ((N + 31) / 32) > stage_ints) stage_ints = (size_t)N * ((N + 31) / 32);
  const size_t lds = (size_t)(2 * 1024 + stage_ints) * sizeof(int);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned long long* masks = static_cast<unsigned long long*>(workspace);
  int* inst_tot = reinterpret_cast<int*>(static_cast<char*>(workspace) + magat_align_up((size_t)B * N * W64 * 8, 256));
  int* rowdeg = reinterpret_cast<int*>(reinterpret_cast<char*>(inst_tot) + magat_align_up((size_t)B * sizeof(int), 256));
  const long long rows = (long long)B * N;
  const int pid = magat_prof_begin(MAGAT_TAG_GSO_CSR, st);
  if (phase != 2) {
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (s_is_f64)
      hipLaunchKernelGGL(gso_mask_kernel<double>, dim3(blocks), dim3(256), 0, st, static_cast<double*>(S), masks, rowdeg,
                         N, W64, rows, scrub_nan, gso_mode, edge_rule);
    else if (MAGAT_GSO_X4 && N % 4 == 0 && !(reinterpret_cast<uintptr_t>(S) & 15))
      hipLaunchKernelGGL(gso_mask_x4_kernel, dim3(blocks), dim3(256), 0, st, static_cast<float*>(S), masks, rowdeg, N, W64,
                         rows, scrub_nan, gso_mode, edge_rule);
    else
      hipLaunchKernelGGL(gso_mask_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<float*>(S), masks, rowdeg, N,
                         W64, rows, scrub_nan, gso_mode, edge_rule);
    hipLaunchKernelGGL(gso_totals_kernel, dim3(B), dim3(256), 0, st, rowdeg, inst_tot, N);
  }
  if (phase != 1) {
    if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&gso_structure_kernel), MAGAT_LDS_GSO_STRUCT, lds) != MAGAT_OK)
      return MAGAT_ERR_LAUNCH;
    hipLaunchKernelGGL(gso_structure_kernel, dim3(B), dim3(1024), lds, st, masks, inst_tot, rowptr, colidx, cscptr, cscsrc,
                       cscpos, cap, nnz_dev, B, N, W64, stage_edges);
  }
  magat_prof_end(pid, st);
  return magat_check_launch();
}

extern "C" int magat_gso_csr_build(void* S, int s_is_f64, int scrub_nan, int gso_mode, int edge_rule, int* rowptr,
                                   int* colidx, int* cscptr, int* cscsrc, int* cscpos, long long cap, long long* nnz_dev,
                                   void* workspace, size_t workspace_bytes, int B, int N, void* stream) {
  return magat_gso_csr_build_phase(S, s_is_f64, scrub_nan, gso_mode, edge_rule, rowptr, colidx, cscptr, cscsrc, cscpos, cap,
                                   nnz_dev, workspace, workspace_bytes, B, N, 0, stream);
}

// forward with the CSC view made by magat_gso_csr_build (skips the per-call transpose)
extern "C" int magat_gat_forward_csc_f32(const float* X, const int* rowptr, const int* colidx, const int* cscptr,
                                         const int* cscsrc, const int* cscpos, long long nnz, const float* packed,
                                         const float* bias, float* Y, int ldy, float* att_opt, void* workspace,
                                         size_t workspace_bytes, int B, int N, int G, int F, int K, int P, int mode,
                                         int concat, void* stream) {
  if (!cscptr || !cscsrc || !cscpos) return MAGAT_ERR_NULL;
  return csr_forward<float>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, att_opt, workspace, workspace_bytes, B, N, G,
                            F, K, P, mode, concat, stream, nullptr, cscptr, cscsrc, cscpos);
}
extern "C" int magat_gat_forward_csc_bf16(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr,
                                          const int* cscsrc, const int* cscpos, long long nnz, const float* packed,
                                          const float* bias, uint16_t* Y, int ldy, float* att_opt, void* workspace,
                                          size_t workspace_bytes, int B, int N, int G, int F, int K, int P, int mode,
                                          int concat, void* stream) {
  if (!cscptr || !cscsrc || !cscpos) return MAGAT_ERR_NULL;
  return csr_forward<u16>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, att_opt, workspace, workspace_bytes, B, N, G, F,
                          K, P, mode, concat, stream, nullptr, cscptr, cscsrc, cscpos);
}
extern "C" int magat_gat_forward_csc_bf16_f32out(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr,
                                                 const int* cscsrc, const int* cscpos, long long nnz, const float* packed,
                                                 const float* bias, float* Y, int ldy, float* att_opt, void* workspace,
                                                 size_t workspace_bytes, int B, int N, int G, int F, int K, int P, int mode,
                                                 int concat, void* stream) {
  if (!cscptr || !cscsrc || !cscpos) return MAGAT_ERR_NULL;
  return csr_forward<u16>(X, rowptr, colidx, nnz, packed, bias, Y, ldy, att_opt, workspace, workspace_bytes, B, N, G, F,
                          K, P, mode, concat, stream, nullptr, cscptr, cscsrc, cscpos, 1);
}

// =====================================================================================================
// Training support: forward that keeps what the backward needs, and the
// backward of the graph part of the layer.  The layer is
//     T_{K-1} = U_{K-1};  T_k = U_k + A^T T_{k+1};  Ypre = T_0 + bias;  A = row-softmax(E) on the edges
//     E_ij = x_i . q_j (KeyQuery)  |  lrelu(c1_j + c2_i) (GAT_modified);   [Q | U | c1 c2] = X @ Bt^T + cb
// Given dYpre the kernels below produce dZ (gradient wrt every column of Z) and the direct part of dX;
// the two dense products dX += dZ @ Bt and dBt = dZ^T @ X are plain library GEMMs done by the caller.
//     dT_0 = dYpre;   dT_{k+1} = A dT_k;   dA_ij = sum_k T_{k+1}[i] . dT_k[j];   dU_k = dT_k
//     dE_ij = a_ij (dA_ij - sum_j' a_ij' dA_ij')
//     KeyQuery: dX_i += sum_j dE_ij q_j,  dQ_j = sum_i dE_ij x_i
//     modified: g_ij = dE_ij * lrelu'(c1_j + c2_i),  dc2_i = sum_j g_ij,  dc1_j = sum_i g_ij
// =====================================================================================================
namespace {

struct TrainParams {
  const float* X;
  const float* Z;
  const float* T;        // [(K-2)][M][P*F]   T_k at slot K-2-k  (1 <= k <= K-2)
  const float* att;      // [P][nnz]
  float* datt;           // [P][nnz]  dA, then dE / g in place
  float* dZ;             // [M][NC]
  float* dXd;            // [M][G]
  const int* rowptr; const int* colidx; const int* cscptr; const int* cscsrc; const int* cscpos;
  int B, N, K, P, mode, NC, qoff, uoff, c1off, c2off;
  long long nnz, M;
  int k;                 // hop being differentiated (reads dT_k, writes dT_{k+1})
};

template <int F>
__global__ __launch_bounds__(256) void bwd_hop_kernel(const TrainParams p) {
  constexpr int VEC = F >= 64 ? F / 64 : 1, LANES = F >= 64 ? 64 : F;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  const int N = p.N, tiles = (N + 3) / 4;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD, per = p.P * tiles;
  const int b = xcd + MAGAT_NUM_XCD * (slot / per);
  if (b >= p.B) return;
  const int head = (slot % per) / tiles, tile = slot % tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tile * 4 + wave;
  if (i >= N) return;
  const bool on = lane < LANES;
  const int* rp = p.rowptr + (long long)b * (N + 1);
  const int e0 = rp[i], e1 = rp[i + 1];
  const float* att = p.att + (long long)head * p.nnz;
  float* datt = p.datt + (long long)head * p.nnz;
  const long long row0 = (long long)b * N;
  const int k = p.k, K = p.K;
  // T_{k+1}[i]
  fvec tn;
#pragma unroll
  for (int c = 0; c < VEC; ++c) tn[c] = 0.f;
  if (on) {
    if (k + 1 == K - 1)
      tn = *reinterpret_cast<const fvec*>(p.Z + (row0 + i) * p.NC + p.uoff + (head * K + (K - 1)) * F + VEC * lane);
    else
      tn = *reinterpret_cast<const fvec*>(p.T + ((long long)(K - 2 - (k + 1)) * p.M + row0 + i) * p.P * F + head * F +
                                          VEC * lane);
  }
  const long long ucur = p.uoff + (head * K + k) * F + VEC * lane;       // dT_k lives in dZ's U_k block
  fvec acc;
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
  for (int e = e0; e < e1; ++e) {
    const int j = p.colidx[e];
    const float a = att[e];
    fvec d;
#pragma unroll
    for (int c = 0; c < VEC; ++c) d[c] = 0.f;
    if (on) d = *reinterpret_cast<const fvec*>(p.dZ + (row0 + j) * p.NC + ucur);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      acc[c] = fmaf(a, d[c], acc[c]);
      dot = fmaf(tn[c], d[c], dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) datt[e] = (k == 0 ? 0.f : datt[e]) + dot;
  }
  if (on) *reinterpret_cast<fvec*>(p.dZ + (row0 + i) * p.NC + p.uoff + (head * K + k + 1) * F + VEC * lane) = acc;
}

// softmax backward per row (all heads), then the row-side score gradients
template <int G>
__global__ __launch_bounds__(256) void bwd_scores_rows_kernel(const TrainParams p) {
  constexpr int VEC = G >= 64 ? G / 64 : 1, LANES = G >= 64 ? 64 : G;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  const int N = p.N, tiles = (N + 3) / 4;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int b = xcd + MAGAT_NUM_XCD * (slot / tiles);
  if (b >= p.B) return;
  const int tile = slot % tiles, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tile * 4 + wave;
  if (i >= N) return;
  const bool on = lane < LANES;
  const int* rp = p.rowptr + (long long)b * (N + 1);
  const int e0 = rp[i], e1 = rp[i + 1];
  const long long row0 = (long long)b * N;
  fvec accx;
#pragma unroll
  for (int c = 0; c < VEC; ++c) accx[c] = 0.f;
  for (int head = 0; head < p.P; ++head) {
    const float* att = p.att + (long long)head * p.nnz;
    float* datt = p.datt + (long long)head * p.nnz;
    float s = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) s = fmaf(att[e], datt[e], s);
    s = wave_sum(s);
    if (p.mode == MAGAT_MODE_KEYQUERY) {
      for (int e = e0; e < e1; ++e) {
        const float dE = att[e] * (datt[e] - s);
        if (on) {
          const fvec q = *reinterpret_cast<const fvec*>(p.Z + (row0 + p.colidx[e]) * p.NC + p.qoff + head * G + VEC * lane);
#pragma unroll
          for (int c = 0; c < VEC; ++c) accx[c] = fmaf(dE, q[c], accx[c]);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) datt[e] = dE;
      }
    } else {
      const float c2 = p.Z[(row0 + i) * p.NC + p.c2off + head];
      float g2 = 0.f;
      for (int e = e0 + lane; e < e1; e += 64) {
        const float pre = p.Z[(row0 + p.colidx[e]) * p.NC + p.c1off + head] + c2;
        const float g = att[e] * (datt[e] - s) * (pre > 0.f ? 1.f : 0.2f);
        datt[e] = g;
        g2 += g;
      }
      g2 = wave_sum(g2);
      if (lane == 0) p.dZ[(row0 + i) * p.NC + p.c2off + head] = g2;
    }
  }
  if (p.mode == MAGAT_MODE_KEYQUERY && on) *reinterpret_cast<fvec*>(p.dXd + (row0 + i) * G + VEC * lane) = accx;
}

// column-side score gradients: dQ_j (KeyQuery) or dc1_j (modified), via the CSC view
template <int G>
__global__ __launch_bounds__(256) void bwd_scores_cols_kernel(const TrainParams p) {
  constexpr int VEC = G >= 64 ? G / 64 : 1, LANES = G >= 64 ? 64 : G;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  const int N = p.N, tiles = (N + 3) / 4;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int b = xcd + MAGAT_NUM_XCD * (slot / tiles);
  if (b >= p.B) return;
  const int tile = slot % tiles, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = tile * 4 + wave;
  if (j >= N) return;
  const bool on = lane < LANES;
  const int* cp = p.cscptr + (long long)b * (N + 1);
  const int s0 = cp[j], s1 = cp[j + 1];
  const long long row0 = (long long)b * N;
  for (int head = 0; head < p.P; ++head) {
    const float* dE = p.datt + (long long)head * p.nnz;
    if (p.mode == MAGAT_MODE_KEYQUERY) {
      fvec acc;
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
      for (int s = s0; s < s1; ++s) {
        const float g = dE[p.cscpos[s]];
        if (on) {
          const fvec x = *reinterpret_cast<const fvec*>(p.X + (row0 + p.cscsrc[s]) * G + VEC * lane);
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[c] = fmaf(g, x[c], acc[c]);
        }
      }
      if (on) *reinterpret_cast<fvec*>(p.dZ + (row0 + j) * p.NC + p.qoff + head * G + VEC * lane) = acc;
    } else {
      float g1 = 0.f;
      for (int s = s0 + lane; s < s1; s += 64) g1 += dE[p.cscpos[s]];
      g1 = wave_sum(g1);
      if (lane == 0) p.dZ[(row0 + j) * p.NC + p.c1off + head] = g1;
    }
  }
}

// dU_0 = dYpre
__global__ void bwd_seed_kernel(const float* __restrict__ dY, float* __restrict__ dZ, long long M, int P, int F, int K,
                                int NC, int uoff) {
  const int FC = F / 4;
  const long long total = M * P * FC;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % FC);
    const long long r = idx / FC;
    const int head = (int)(r % P);
    const long long m = r / P;
    *reinterpret_cast<f32x4*>(dZ + m * NC + uoff + (head * K) * F + 4 * c) =
        *reinterpret_cast<const f32x4*>(dY + (m * P + head) * F + 4 * c);
  }
}


// ---- GraphFilterBatch backward (graphML.py:5485-5579 differentiated).  The layer is linear:  Y = b + sum_k A^k X H_k^T
// with A the row operator of "x @ S" (row n gathers S[m][n] X[m]: the CSC view in the forward).  Hence  dU_k = (A^T)^k dY
// - the same hop with the CSR rows of S as gather lists - and the rest is two plain GEMMs of the caller:
// dX = [dU_0 .. dU_{K-1}] Bt,  dH = dU^T X.   dZ [M][K*F]: slice k = dU_k.  One wave per agent row, 4 rows per workgroup.
template <int F>
__global__ __launch_bounds__(256) void gnn_bwd_hop_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                          const float* __restrict__ vals, float* __restrict__ dZ, int B,
                                                          int N, int K, int k) {
  constexpr int VEC = F >= 64 ? F / 64 : 1, LANES = F >= 64 ? 64 : F;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  const int tiles = (N + 3) / 4;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int b = xcd + MAGAT_NUM_XCD * (slot / tiles);
  if (b >= B) return;
  const int tile = slot % tiles, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tile * 4 + wave;
  if (i >= N || lane >= LANES) return;
  const int* rp = rowptr + (long long)b * (N + 1);
  const int e0 = rp[i], e1 = rp[i + 1];
  const long long row0 = (long long)b * N, ld = (long long)K * F;
  const float* src = dZ + (long long)(k - 1) * F + VEC * lane;
  fvec acc;
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
  for (int e = e0; e < e1; ++e) {
    const float a = vals[e];
    const fvec d = *reinterpret_cast<const fvec*>(src + (row0 + colidx[e]) * ld);
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a, d[c], acc[c]);
  }
  *reinterpret_cast<fvec*>(dZ + (row0 + i) * ld + (long long)k * F + VEC * lane) = acc;
}

__global__ void gnn_bwd_seed_kernel(const float* __restrict__ dY, float* __restrict__ dZ, long long M, int F, int K) {
  const int FC = F / 4;
  const long long total = M * FC;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % FC);
    const long long m = idx / FC;
    *reinterpret_cast<f32x4*>(dZ + m * K * F + 4 * c) = *reinterpret_cast<const f32x4*>(dY + m * F + 4 * c);
  }
}

template <int W, typename KFn>
int launch_rows(KFn kern, const TrainParams& p, int per_instance_factor, hipStream_t st) {
  const int tiles = (p.N + 3) / 4;
  const long long grid = (long long)((p.B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * MAGAT_NUM_XCD * per_instance_factor * tiles;
  if (grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, st, p);
  return magat_check_launch();
}

#define MAGAT_WIDTH_SWITCH(W, CALL)        \
  switch (W) {                             \
    case 16: { constexpr int WW = 16; CALL; } break;   \
    case 32: { constexpr int WW = 32; CALL; } break;   \
    case 64: { constexpr int WW = 64; CALL; } break;   \
    case 128: { constexpr int WW = 128; CALL; } break; \
    default: { constexpr int WW = 256; CALL; }         \
  }

}  // namespace

extern "C" int magat_gat_train_forward_f32(const float* X, const int* rowptr, const int* colidx, long long nnz,
                                           const float* packed, const float* bias, float* Ypre, float* att, float* Z,
                                           float* T, int* cscptr, int* cscsrc, int* cscpos, int* csctmp, int B, int N,
                                           int G, int F, int K, int P, int mode, void* stream) {
  if (!X || !rowptr || !packed || !Ypre || !att || !Z || !cscptr || !cscsrc || !cscpos || !csctmp) return MAGAT_ERR_NULL;
  if (nnz > 0 && !colidx) return MAGAT_ERR_NULL;
  if (K > 2 && !T) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || nnz < 0 || K <= 0 || P <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (mode < MAGAT_MODE_KEYQUERY || mode > MAGAT_MODE_GAT_ORIGIN) return MAGAT_ERR_UNSUPPORTED;
  if (G != F || !(G == 16 || G == 32 || G == 64 || G == 128 || G == 256)) return MAGAT_ERR_UNSUPPORTED;
  if ((size_t)(2 * N + 2) * sizeof(int) > 64 * 1024) return MAGAT_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Layout L = layout(G, F, K, P, mode);
  const long long M = (long long)B * N;
  // float32 MFMA maps: Z is kept for the backward, and with caller-owned buffers there is no status word for a guarded
  // split GEMM (the maps are 5 % of a training step either way)
  int rc = magat_gat_maps_gemm(X, packed, Z, (int)M, G, L.NC, L.NC, stream, 0, nullptr, 1);
  if (rc != MAGAT_OK) return rc;
  CsrParams p = {};
  p.X = X; p.Z = Z; p.rowptr = rowptr; p.colidx = colidx; p.cscptr = cscptr; p.cscsrc = cscsrc; p.cscpos = cscpos;
  p.att = att; p.bias = bias; p.Y = Ypre; p.ldy = P * F;
  p.B = B; p.N = N; p.K = K; p.P = P; p.mode = mode; p.concat = 1; p.act_relu = 0; p.nnz = nnz;
  p.NC = L.NC; p.qoff = L.qoff; p.uoff = L.uoff; p.c1off = L.c1off; p.c2off = L.c2off;
  hipLaunchKernelGGL(csr_transpose_kernel, dim3(B), dim3(256), (size_t)(2 * N + 2) * sizeof(int), st, rowptr, colidx,
                     cscptr, csctmp, N);
  hipLaunchKernelGGL(csr_sort_columns_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, rowptr, cscptr, csctmp,
                     cscsrc, cscpos, N, M);
  if ((rc = magat_check_launch()) != MAGAT_OK) return rc;
  MAGAT_WIDTH_SWITCH(G, rc = run_scores<WW>(p, st));
  if (rc != MAGAT_OK) return rc;
  if (K == 1) {
    MAGAT_WIDTH_SWITCH(F, rc = run_k1<WW>(p, st));
    return rc;
  }
  for (int k = K - 2, h = 0; k >= 0; --k, ++h) {
    p.k = k;
    p.last = k == 0;
    if (h == 0) {
      p.Told = Z + L.uoff + (K - 1) * F; p.told_ld = L.NC; p.told_head_stride = K * F;
    } else {
      p.Told = T + (size_t)(h - 1) * M * P * F; p.told_ld = P * F; p.told_head_stride = F;
    }
    p.Tnew = T ? T + (size_t)h * M * P * F : nullptr;     // T_k for k = K-2-h (kept for the backward)
    MAGAT_WIDTH_SWITCH(F, rc = run_hop<WW>(p, st));
    if (rc != MAGAT_OK) return rc;
  }
  return MAGAT_OK;
}

extern "C" int magat_gat_train_backward_f32(const float* dYpre, const float* X, const float* Z, const float* att,
                                            const float* T, const int* rowptr, const int* colidx, const int* cscptr,
                                            const int* cscsrc, const int* cscpos, long long nnz, float* dZ, float* dXd,
                                            float* datt, int B, int N, int G, int F, int K, int P, int mode,
                                            void* stream) {
  if (!dYpre || !X || !Z || !att || !rowptr || !cscptr || !cscsrc || !cscpos || !dZ || !dXd || !datt) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || nnz < 0 || K <= 0 || P <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (G != F || !(G == 16 || G == 32 || G == 64 || G == 128 || G == 256)) return MAGAT_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Layout L = layout(G, F, K, P, mode);
  TrainParams p = {};
  p.X = X; p.Z = Z; p.T = T; p.att = att; p.datt = datt; p.dZ = dZ; p.dXd = dXd;
  p.rowptr = rowptr; p.colidx = colidx; p.cscptr = cscptr; p.cscsrc = cscsrc; p.cscpos = cscpos;
  p.B = B; p.N = N; p.K = K; p.P = P; p.mode = mode; p.NC = L.NC; p.qoff = L.qoff; p.uoff = L.uoff;
  p.c1off = L.c1off; p.c2off = L.c2off; p.nnz = nnz; p.M = (long long)B * N;
  if (hipMemsetAsync(dZ, 0, (size_t)p.M * L.NC * sizeof(float), st) != hipSuccess) return MAGAT_ERR_LAUNCH;
  if (hipMemsetAsync(dXd, 0, (size_t)p.M * G * sizeof(float), st) != hipSuccess) return MAGAT_ERR_LAUNCH;
  {
    long long blocks = (p.M * P * (F / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bwd_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dYpre, dZ, p.M, P, F, K, L.NC, L.uoff);
    int rc = magat_check_launch();
    if (rc != MAGAT_OK) return rc;
  }
  if (K == 1) return MAGAT_OK;     // no graph terms: dZ = [0 | dYpre]
  int rc = MAGAT_OK;
  for (int k = 0; k <= K - 2; ++k) {
    p.k = k;
    MAGAT_WIDTH_SWITCH(F, rc = launch_rows<WW>(bwd_hop_kernel<WW>, p, P, st));
    if (rc != MAGAT_OK) return rc;
  }
  MAGAT_WIDTH_SWITCH(G, rc = launch_rows<WW>(bwd_scores_rows_kernel<WW>, p, 1, st));
  if (rc != MAGAT_OK) return rc;
  MAGAT_WIDTH_SWITCH(G, rc = launch_rows<WW>(bwd_scores_cols_kernel<WW>, p, 1, st));
  return rc;
}

extern "C" int magat_gnn_backward_csr_f32(const float* dY, const int* rowptr, const int* colidx, const float* vals,
                                          long long nnz, float* dZ, int B, int N, int F, int K, void* stream) {
  if (!dY || !rowptr || !dZ) return MAGAT_ERR_NULL;
  if (nnz > 0 && K > 1 && (!colidx || !vals)) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || nnz < 0 || K <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (!(F == 16 || F == 32 || F == 64 || F == 128 || F == 256)) return MAGAT_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)B * N;
  long long blocks = (M * (F / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gnn_bwd_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dY, dZ, M, F, K);
  int rc = magat_check_launch();
  const int tiles = (N + 3) / 4;
  const long long grid = (long long)((B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD) * MAGAT_NUM_XCD * tiles;
  if (grid > 0x7fffffffLL) return MAGAT_ERR_BAD_SHAPE;
  for (int k = 1; k < K && rc == MAGAT_OK; ++k) {
    MAGAT_WIDTH_SWITCH(F, hipLaunchKernelGGL(gnn_bwd_hop_kernel<WW>, dim3((unsigned)grid), dim3(256), 0, st, rowptr, colidx,
                                             vals, dZ, B, N, K, k));
    rc = magat_check_launch();
  }
  return rc;
}
