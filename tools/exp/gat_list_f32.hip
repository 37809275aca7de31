// Sparse-structure ("list") form of the dense-GSO GAT kernel for N <= 128, G = F >= 64.
//
// gat_dense_kernel (gat_f32.hip) keeps a dense N x N attention tile plus two N x G feature tiles in LDS
// (146 KB at N=100, G=128): one workgroup per CU, so its HBM phase and its LDS/ALU phases never overlap, and
// its hop loops walk bitmasks on the scalar unit.  Communication graphs are sparse (degree 5-10), so here the
// graph structure is turned ONCE per instance (shared by all heads) into CSR + CSC lists by gat_struct_kernel,
// and gat_list_kernel needs only ONE N x G tile (Q_p, then the hop operand updated in place through registers)
// + the instance's lists + its attention values in CSC order: 77 KB at N=100 -> two workgroups per CU, whose
// memory and compute phases overlap.  Loops are counted (no bit twiddling); weights/indices of a column are read
// once per row as vectors and broadcast with v_readlane.
// Instances whose edge count exceeds the LDS list capacity are flagged and left to gat_dense_kernel.
//
// STATUS (round 1): correct (the whole GPU suite passes with it enabled) but NOT faster than gat_dense_kernel on
// MI355X - c3: 210 us + 44 us structure pass vs 220 us; c2: 68 + 9 us vs 64 us - because both kernels are bound by
// instruction issue (~100M vs 135M SQ active quad-cycles per launch, rocprofv3), not by occupancy.  It is therefore
// opt-in (MAGAT_GAT_LIST=1) and kept as the starting point for a multi-head-per-wave variant.
#include "magat_common.h"

namespace {

struct ListParams {
  const float* X;      // [B*N, G]
  const void* S;       // [B,N,N]
  const float* Z;      // [B*N, NC]
  const float* bias;
  float* Y;
  float* A_opt;        // [B,P,N,N] (pre-zeroed by the launcher) or null
  unsigned char* gs;   // struct buffers, gs_stride bytes per instance
  int* over;           // [B] 1 = instance exceeds the list capacity -> dense kernel
  int B, N, K, P, mode, concat, s_is_f64;
  int ldy, NC, qoff, uoff, c1off, c2off;
  int cap, gs_stride, b0;
};

// per-instance struct buffer: int nnz, int over, pad[2] | u16 rowptr[N+1] | u16 colptr[N+1] | (4-aligned)
// u8 colidx[cap] | u8 cscsrc[cap] | u16 slot[cap]
__host__ __device__ inline int gs_ptr_bytes(int N) { return ((2 * (N + 1) * 2) + 3) & ~3; }
__host__ __device__ inline int gs_bytes(int N, int cap) { return (16 + gs_ptr_bytes(N) + 4 * cap + 15) & ~15; }

template <typename T>
__device__ __forceinline__ bool edge_of(const T* S, long long idx) {
  const T v = S[idx];
  return (v < 0 ? -v : v) > (T)1e-9;
}

// ---- structure pass: one workgroup (256 threads) per instance
__global__ __launch_bounds__(256) void gat_struct_kernel(const ListParams p) {
  __shared__ unsigned rm[128 * 4], cm[128 * 4];
  __shared__ unsigned short rp[130], cp[130];
  const int N = p.N, b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long sbase = (long long)(p.b0 + b) * N * N;
  for (int i = wave; i < N; i += 4) {
    bool f0 = false, f1 = false;
    if (p.s_is_f64) {
      const double* S = static_cast<const double*>(p.S);
      f0 = lane < N && edge_of(S, sbase + (long long)i * N + lane);
      f1 = lane + 64 < N && edge_of(S, sbase + (long long)i * N + lane + 64);
    } else {
      const float* S = static_cast<const float*>(p.S);
      f0 = lane < N && edge_of(S, sbase + (long long)i * N + lane);
      f1 = lane + 64 < N && edge_of(S, sbase + (long long)i * N + lane + 64);
    }
    const unsigned long long k0 = __ballot(f0), k1 = __ballot(f1);
    if (lane == 0) {
      rm[4 * i] = (unsigned)k0; rm[4 * i + 1] = (unsigned)(k0 >> 32);
      rm[4 * i + 2] = (unsigned)k1; rm[4 * i + 3] = (unsigned)(k1 >> 32);
    }
  }
  __syncthreads();
  for (int j = wave; j < N; j += 4) {      // bit-matrix transpose by ballot
    const bool f0 = lane < N && ((rm[4 * lane + (j >> 5)] >> (j & 31)) & 1u);
    const bool f1 = lane + 64 < N && ((rm[4 * (lane + 64) + (j >> 5)] >> (j & 31)) & 1u);
    const unsigned long long k0 = __ballot(f0), k1 = __ballot(f1);
    if (lane == 0) {
      cm[4 * j] = (unsigned)k0; cm[4 * j + 1] = (unsigned)(k0 >> 32);
      cm[4 * j + 2] = (unsigned)k1; cm[4 * j + 3] = (unsigned)(k1 >> 32);
    }
  }
  __syncthreads();
  if (t == 0 || t == 64) {                  // two tiny serial prefix sums
    const unsigned* m = t == 0 ? rm : cm;
    unsigned short* o = t == 0 ? rp : cp;
    int acc = 0;
    for (int i = 0; i < N; ++i) {
      o[i] = (unsigned short)acc;
      acc += __popc(m[4 * i]) + __popc(m[4 * i + 1]) + __popc(m[4 * i + 2]) + __popc(m[4 * i + 3]);
    }
    o[N] = (unsigned short)acc;
  }
  __syncthreads();
  const int nnz = rp[N];
  unsigned char* g = p.gs + (long long)b * p.gs_stride;
  const bool over = nnz > p.cap;
  if (t == 0) {
    reinterpret_cast<int*>(g)[0] = nnz;
    reinterpret_cast<int*>(g)[1] = over ? 1 : 0;
    p.over[b] = over ? 1 : 0;
  }
  if (over) return;
  unsigned short* grp = reinterpret_cast<unsigned short*>(g + 16);
  unsigned short* gcp = grp + (N + 1);
  unsigned char* gcol = g + 16 + gs_ptr_bytes(N);
  unsigned char* gsrc = gcol + p.cap;
  unsigned short* gslot = reinterpret_cast<unsigned short*>(gsrc + p.cap);
  for (int i = t; i <= N; i += 256) { grp[i] = rp[i]; gcp[i] = cp[i]; }
  if (t < N) {                               // row t: CSR columns + CSC slot of each edge
    int k = rp[t];
    for (int w = 0; w < 4; ++w) {
      unsigned m = rm[4 * t + w];
      while (m) {
        const int j = 32 * w + __builtin_ctz(m);
        m &= m - 1;
        // rank of row t among the sources of column j
        int r = 0;
        const int tw = t >> 5;
        for (int q = 0; q < tw; ++q) r += __popc(cm[4 * j + q]);
        r += __popc(cm[4 * j + tw] & ((1u << (t & 31)) - 1u));
        gcol[k] = (unsigned char)j;
        gslot[k] = (unsigned short)(cp[j] + r);
        ++k;
      }
    }
  } else if (t >= 128 && t - 128 < N) {      // column j: CSC sources, ascending
    const int j = t - 128;
    int k = cp[j];
    for (int w = 0; w < 4; ++w) {
      unsigned m = cm[4 * j + w];
      while (m) {
        gsrc[k++] = (unsigned char)(32 * w + __builtin_ctz(m));
        m &= m - 1;
      }
    }
  }
}

// ---- main kernel: one workgroup per (instance, head)
template <int G>
__global__ __launch_bounds__(512, 4) void gat_list_kernel(const ListParams p) {
  constexpr int F = G, GC = G / 4, CP8 = GC / 8, VEC = F / 64;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = p.N, K = p.K;
  const int bid = blockIdx.x, xcd = bid % MAGAT_NUM_XCD, slot_id = bid / MAGAT_NUM_XCD;
  const int bl = xcd + MAGAT_NUM_XCD * (slot_id / p.P);
  const int head = slot_id % p.P;
  if (bl >= p.B) return;
  if (p.over[bl]) return;                    // too dense for the lists: gat_dense_kernel owns this instance
  const int b = p.b0 + bl;

  float* T = reinterpret_cast<float*>(smem_raw);                 // [N][G]  Q_p, then the hop operand
  float* att = T + N * G;                                        // [cap]   attention in CSC order
  unsigned char* ls = reinterpret_cast<unsigned char*>(att + p.cap);   // struct copy
  const unsigned short* rp = reinterpret_cast<const unsigned short*>(ls + 16);
  const unsigned short* cp = rp + (N + 1);
  const unsigned char* colidx = ls + 16 + gs_ptr_bytes(N);
  const unsigned char* cscsrc = colidx + p.cap;
  const unsigned short* slot = reinterpret_cast<const unsigned short*>(cscsrc + p.cap);

  const int t = threadIdx.x, NT = blockDim.x, lane = t & 63, wave = t >> 6, nwaves = NT >> 6;
  const float* Zb = p.Z + (long long)bl * N * p.NC;
  const float* Xb = p.X + (long long)b * N * G;
  const bool keyquery = p.mode == MAGAT_MODE_KEYQUERY;
  const bool need_att = K > 1 || p.A_opt;

  // ---- stage: lists + Q_p -> LDS
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.gs + (long long)bl * p.gs_stride);
    uint4* dst = reinterpret_cast<uint4*>(ls);
    const int nv = gs_bytes(N, p.cap) / 16;
    for (int i = t; i < nv; i += NT) dst[i] = src[i];
    if (keyquery && need_att) {
      const int qo = p.qoff + head * G;
      for (int idx = t; idx < N * GC; idx += NT) {
        const int n = idx / GC, c = idx % GC;
        *reinterpret_cast<f32x4*>(T + n * G + 4 * c) =
            *reinterpret_cast<const f32x4*>(Zb + (long long)n * p.NC + qo + 4 * c);
      }
    }
  }
  __syncthreads();

  // ---- scores + row softmax: 8 lanes per graph row
  if (need_att) {
    const int es = lane & 7, grp = t >> 3, ngrp = NT >> 3, par = (lane >> 3) & 1;
    for (int i = grp; i < N; i += ngrp) {
      const int e0 = rp[i], e1 = rp[i + 1];
      if (e1 == e0) continue;
      float mx = -__builtin_inff(), sum = 0.f;
      auto online = [&](float d) {
        const float m2 = fmaxf(mx, d);
        sum = sum * __expf(mx - m2) + __expf(d - m2);
        mx = m2;
      };
      if (keyquery) {
        f32x4 xi[CP8];
#pragma unroll
        for (int q = 0; q < CP8; ++q)
          xi[q] = *reinterpret_cast<const f32x4*>(Xb + (long long)i * G + 4 * (es + 8 * (q ^ par)));
#pragma nounroll
        for (int e = e0; e < e1; e += 2) {
          const bool two = e + 1 < e1;
          const int j0 = colidx[e], j1 = two ? colidx[e + 1] : j0;
          float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int q = 0; q < CP8; ++q) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(T + j0 * G + 4 * (es + 8 * (q ^ par)));
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(T + j1 * G + 4 * (es + 8 * (q ^ par)));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              d0 = fmaf(xi[q][c], q0[c], d0);
              d1 = fmaf(xi[q][c], q1[c], d1);
            }
          }
          d0 = oct_sum(d0);
          d1 = oct_sum(d1);
          if (es == 0) {
            att[slot[e]] = d0;
            if (two) att[slot[e + 1]] = d1;
          }
          online(d0);
          if (two) online(d1);
        }
      } else {
        const float c2 = Zb[(long long)i * p.NC + p.c2off + head];
        for (int e = e0 + es; e < e1; e += 8) {
          const float v = Zb[(long long)colidx[e] * p.NC + p.c1off + head] + c2;
          const float l = v > 0.f ? v : 0.2f * v;
          att[slot[e]] = l;
          online(l);
        }
        const float gm = oct_max(mx);
        sum = oct_sum(sum > 0.f ? sum * __expf(mx - gm) : 0.f);
        mx = gm;
      }
      const float inv = 1.f / sum;
      float* ao = p.A_opt ? p.A_opt + (((long long)b * p.P + head) * N + i) * N : nullptr;
      for (int e = e0 + es; e < e1; e += 8) {      // same-wave LDS accesses are ordered: the raw scores are visible
        const int sl = slot[e];
        const float a = __expf(att[sl] - mx) * inv;
        att[sl] = a;
        if (ao) ao[colidx[e]] = a;
      }
    }
  }
  __syncthreads();

  // ---- the deepest hop operand replaces Q_p
  if (K > 1) {
    const int uo = p.uoff + (head * K + (K - 1)) * F;
    for (int idx = t; idx < N * GC; idx += NT) {
      const int n = idx / GC, c = idx % GC;
      *reinterpret_cast<f32x4*>(T + n * F + 4 * c) =
          *reinterpret_cast<const f32x4*>(Zb + (long long)n * p.NC + uo + 4 * c);
    }
    __syncthreads();
  }

  // ---- Horner hops, in place: every wave keeps its rows' new values in registers until all reads are done.
  // The per-wave row buffer is ONE wide vector value indexed dynamically: the backend lowers that to indexed
  // register moves (s_set_gpr_idx), so the row loop stays a real loop (small code, no scratch).
  constexpr int RMAX = 16;                     // rows per wave: N <= 128 with 8 waves, <= 64 with 4, <= 32 with 2
  typedef float tvec __attribute__((ext_vector_type(RMAX * VEC)));
  tvec tn;
  for (int k = K > 1 ? K - 2 : 0; k >= 0; --k) {
    const bool last = k == 0;
    const int uo = p.uoff + (head * K + k) * F;
    fvec bv;
#pragma unroll
    for (int c = 0; c < VEC; ++c) bv[c] = 0.f;
    if (last && p.bias) bv = *reinterpret_cast<const fvec*>(p.bias + VEC * lane);
#pragma nounroll
    for (int r = 0; r < RMAX; ++r) {
      const int j = wave + r * nwaves;
      if (j >= N) break;
      fvec acc = *reinterpret_cast<const fvec*>(Zb + (long long)j * p.NC + uo + VEC * lane);
      if (K > 1) {
        const int s0 = cp[j], s1 = cp[j + 1];
#pragma nounroll
        for (int sb = s0; sb < s1; sb += 64) {    // column's in-edges, 64 at a time: one vector read each
          const int cnt = min(64, s1 - sb);
          const int iv = lane < cnt ? (int)cscsrc[sb + lane] : 0;
          const float av = lane < cnt ? att[sb + lane] : 0.f;
          int u = 0;
#pragma nounroll
          for (; u + 1 < cnt; u += 2) {
            const int i0 = __builtin_amdgcn_readlane(iv, u), i1 = __builtin_amdgcn_readlane(iv, u + 1);
            const float a0 = lane_bcast(av, u), a1 = lane_bcast(av, u + 1);
            const fvec t0 = *reinterpret_cast<const fvec*>(T + i0 * F + VEC * lane);
            const fvec t1 = *reinterpret_cast<const fvec*>(T + i1 * F + VEC * lane);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a1, t1[c], fmaf(a0, t0[c], acc[c]));
          }
          if (u < cnt) {
            const int i0 = __builtin_amdgcn_readlane(iv, u);
            const float a0 = lane_bcast(av, u);
            const fvec t0 = *reinterpret_cast<const fvec*>(T + i0 * F + VEC * lane);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a0, t0[c], acc[c]);
          }
        }
      }
      if (last) {
        fvec res = acc + bv;
        if (p.concat) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) res[c] = fmaxf(res[c], 0.f);
        }
        *reinterpret_cast<fvec*>(p.Y + ((long long)b * N + j) * p.ldy + head * F + VEC * lane) = res;
      } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) tn[r * VEC + c] = acc[c];
      }
    }
    if (last) break;
    __syncthreads();                           // all reads of T done
#pragma nounroll
    for (int r = 0; r < RMAX; ++r) {
      const int j = wave + r * nwaves;
      if (j >= N) break;
      fvec v;
#pragma unroll
      for (int c = 0; c < VEC; ++c) v[c] = tn[r * VEC + c];
      *reinterpret_cast<fvec*>(T + j * F + VEC * lane) = v;
    }
    __syncthreads();
  }
}

int list_threads(int N) { return N <= 32 ? 128 : (N <= 64 ? 256 : 512); }

}  // namespace

// capacity (edges per instance) of the LDS lists; 0 = list path not applicable for this shape
int magat_gat_list_capacity(int N, int G, int F) {
  if (G != F || (G != 64 && G != 128) || N > 128 || N < 1) return 0;
  const long long budget = 80 * 1024 - 256 - (long long)N * G * 4 - 16 - gs_ptr_bytes(N) - 16;
  long long cap = budget / 8;
  if (cap > 4096) cap = 4096;
  const long long all = ((long long)N * N + 63) / 64 * 64;
  if (cap > all) cap = all;
  cap = cap / 64 * 64;
  if (cap < 4LL * N) return 0;
  return (int)cap;
}

size_t magat_gat_list_workspace_bytes(int B, int N, int G, int F) {
  const int cap = magat_gat_list_capacity(N, G, F);
  if (!cap) return 0;
  return magat_align_up((size_t)B * gs_bytes(N, cap), 256) + magat_align_up((size_t)B * sizeof(int), 256);
}

// Runs the structure pass + list kernel for instances [b0, b0+B).  `over` (device, B ints inside ws) tells the
// caller's dense kernel which instances are left to it.  Returns the `over` pointer through over_out.
int magat_gat_list_run(const float* X, const void* S, int s_is_f64, const float* Z, const float* bias, float* Y,
                       int ldy, float* A_opt, void* ws, int B, int b0, int N, int G, int K, int P, int mode,
                       int concat, int NC, int qoff, int uoff, int c1off, int c2off, int** over_out,
                       hipStream_t st) {
  const int cap = magat_gat_list_capacity(N, G, G);
  if (!cap) return MAGAT_ERR_UNSUPPORTED;
  ListParams p;
  p.X = X; p.S = S; p.Z = Z; p.bias = bias; p.Y = Y; p.A_opt = A_opt;
  p.gs = static_cast<unsigned char*>(ws);
  p.gs_stride = gs_bytes(N, cap);
  p.over = reinterpret_cast<int*>(static_cast<char*>(ws) + magat_align_up((size_t)B * p.gs_stride, 256));
  p.B = B; p.N = N; p.K = K; p.P = P; p.mode = mode; p.concat = concat; p.s_is_f64 = s_is_f64;
  p.ldy = ldy; p.NC = NC; p.qoff = qoff; p.uoff = uoff; p.c1off = c1off; p.c2off = c2off;
  p.cap = cap; p.b0 = b0;
  *over_out = p.over;
  int pid = magat_prof_begin(MAGAT_TAG_GAT_PACK, st);
  hipLaunchKernelGGL(gat_struct_kernel, dim3(B), dim3(256), 0, st, p);
  magat_prof_end(pid, st);
  if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  const size_t lds = (size_t)N * G * 4 + (size_t)cap * 4 + gs_bytes(N, cap);
  const int blocks = (B + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD * MAGAT_NUM_XCD * P;
  const int threads = list_threads(N);
  pid = magat_prof_begin(MAGAT_TAG_GAT_GRAPH, st);
  static size_t conf[3] = {0, 0, 0};
  auto go = [&](auto kern, size_t& configured) -> int {
    if (lds > configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return MAGAT_ERR_LAUNCH;
      configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, st, p);
    return MAGAT_OK;
  };
  int rc;
  if (G == 64) rc = go(gat_list_kernel<64>, conf[0]);
  else rc = go(gat_list_kernel<128>, conf[1]);
  magat_prof_end(pid, st);
  if (rc != MAGAT_OK) return rc;
  return magat_check_launch();
}
