// GraphFilterBatchAttentional.forward on gfx950 (reference utils/graphUtils/graphML.py:4636-4671,
// 1724-1827, 1180-1286, 713-823), dense-GSO path (N <= 128).
//
// Algebra (SURVEY.md section 8(a) "verified restatement"), with X[b,n,:] the node-feature rows:
//   M[i,j]   = |S[b,i,j]| > 1e-9
//   KeyQuery : e[i,j] = x_i . q_j,  q_j = W_p x_j
//   modified : e[i,j] = lrelu_0.2(c1_j + c2_i),  c1 = X (W_p^T a1) + a1.wb,  c2 = X (W_p^T a2) + a2.wb
//   A[i,j]   = softmax over the edges of row i (rows without edges are all zero)
//   Y_p      = U_0 + A^T (U_1 + A^T (U_2 + ...)),   U_k = X H_{p,k}^T     (Horner form of sum_k (A^T)^k X H_k^T)
//   concat: out[n, p*F+f] = relu(Y_p[n,f] + bias[f]);  mean: relu(sum_p (Y_p + bias) / P)
//
// Two launches per call (per chunk of instances):
//   1. the dense per-agent maps Z = X @ [W_p | H_{p,k}]^T on the split-MFMA GEMM (f16x3, conv_gemm_bf16x6.hip; fp32 MFMA
//      for shapes it does not take) - they do not shrink with graph sparsity and are ~95 % of the layer's flops;
//   2. gat_dense_kernel below.  Wide path (G, F >= 64): PERSISTENT workgroups - when the LDS tiles allow one workgroup
//      per CU, the grid is one workgroup per CU and each walks its instances and all heads of an instance.  The
//      [N][F] Q_p / U_{K-1} tiles travel global -> LDS with the LDS-direct load (global_load_lds_dwordx4): the next
//      head's (or instance's) first tile streams in during the current head's last hop, U_{K-1} during the score
//      phase.  Edge scores are 8-lane dot products out of LDS (packed FMAs, DPP reductions) over one walk of the row's
//      128-bit edge mask; the softmax runs on the <= 2 scores a lane keeps in registers (dense fallback for rows of more
//      than 16 edges); the K-1 hops gather neighbour rows from LDS with a wave-uniform scalar loop (column aggregation
//      with row-normalised weights - the reference's x @ aij quirk).  Y is written once, already in the (B*N, P*F)
//      layout actionsMLP consumes.  ~3 flop/byte: the kernel whose GB/s is quoted against the HBM roof.
//      Narrow path (G < 64): one workgroup per (instance, head), tiles staged through registers.
#include <cstdlib>
#include <type_traits>

#include "magat_common.h"
#include "skinny_rows.h"

namespace {

// Instrumentation (phase timestamps, phase skipping) exists only in builds made with -DMAGAT_DEBUG_HOOKS
// (python -m magat_pathplanning_amd.build_native --debug -> lib/libmagat_hip_debug.so); the release library has none of it.
#ifdef MAGAT_DEBUG_HOOKS
constexpr bool kDebugHooks = true;
#else
constexpr bool kDebugHooks = false;
#endif

struct GatParams {
  const float* X;   // [B*N, ldx]
  const void* S;    // [B,N,N] f32 or f64
  const float* Z;   // [chunkB*N, NC]  hoisted linear maps of this chunk
  const float* bias;
  float* Y;         // concat: [B*N, ldy] (+ head*F);  mean: Ytmp [B*N, P*F]
  float* A_opt;     // [B,P,N,N] or null
  int B, N, K, P, mode, concat, s_is_f64;
  int ldx, ldy, NC, lda_a;
  int qoff, uoff, c1off, c2off;  // column offsets inside a Z row
  int b0;                        // first instance of this chunk
  long long* dbg;                // optional phase timestamps [blocks][8] (instrumentation; null in production)
  int skip;                      // instrumentation: bit0 skip scores, bit1 skip hops (wrong results; for PMC deltas)
  const int* order;              // when set: slot s of the instance walk processes instance order[s] (balanced assignment)
  const unsigned* rmask_pre;     // when set: [B][N][4] edge masks made by gat_prepare_kernel (the kernel then never reads S)
  long long zts;                 // 0: Z rows are NC wide; > 0: Z is split into 128-column tiles zts floats apart (row stride 128):
                                 // an instance's [N][128] tile is one contiguous run (written so by the maps GEMM)
  float* Ymean;                  // fused head-mean (mean merge, hpb == P): final output [B*N, ldym]; Y is unused then
  int ldym;
  int hpb;                       // heads per workgroup (1, or P: the workgroup walks all heads of its instance and
                                 // loads the next head's Q tile while the current head computes)
  const int* run_if;             // when set: the launch is a no-op unless *run_if != 0 (range-guard re-run of gat_mfma.hip)
  int* book;                     // when set: this launch is the last reader of the guard's flag (magat_guard_book)
};

// diag: 1 on the diagonal when the mode adds self-loops (GAT_origin: S.float() + I, graphML.py:1018)
__device__ __forceinline__ bool is_edge(const void* S, long long idx, int f64, float diag = 0.f) {
  if (diag != 0.f) {
    const float v = f64 ? (float)static_cast<const double*>(S)[idx] : static_cast<const float*>(S)[idx];
    return fabsf(v + diag) > 1e-9f;
  }
  if (f64) return fabs(static_cast<const double*>(S)[idx]) > 1e-9;
  return fabsf(static_cast<const float*>(S)[idx]) > 1e-9f;
}

// Block size is 8 threads per (power-of-two-rounded) node: the wide score phase is exactly one 8-row step per wave, and
// every staging loop of the narrow path is a fixed, fully unrolled handful of 16-byte loads per thread.
template <int G, int F>
__global__ void gat_dense_kernel(const GatParams p) {
  constexpr int RW = G > F ? G : F;
  constexpr int GC = G / 4, FC = F / 4;             // 16-byte chunks per row
  constexpr int LE = GC < 16 ? GC : 16;             // lanes per edge in the score phase
  constexpr int CPL = GC / LE;                      // chunks per lane
  constexpr int EPS = 64 / LE;                      // edges per wave step
  constexpr int LF = FC < 64 ? FC : 64;             // lanes per output row in the hop phase
  constexpr int RPW = 64 / LF;                      // rows per wave step
  constexpr int QG = GC / 8 > 0 ? GC / 8 : 1;       // staged chunks per thread (NT >= 8 N)
  constexpr int QF = FC / 8 > 0 ? FC / 8 : 1;
  constexpr bool WIDE = G >= 64 && F >= 64;         // one wave spans a whole feature row
  constexpr int VEC = WIDE ? F / 64 : 4;            // floats per lane of a feature row in the hop phase
  constexpr int HMAX = WIDE ? 8 : (8 / RPW > 0 ? 8 / RPW : 1);   // hop rows (row-groups) per wave
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  static_assert(FC <= 64, "F <= 256");
  extern __shared__ __align__(16) float smem[];

  const int N = p.N;
  const int bid = blockIdx.x;
  const int xcd = bid % MAGAT_NUM_XCD, slot = bid / MAGAT_NUM_XCD;
  const int hpb = p.hpb, hgroups = p.P / hpb;
  // heads of one instance share an XCD (X_b, S_b in L2).  A workgroup walks instances bl0, bl0 + istride, ...
  // (istride >= B: one instance per workgroup); the first tile of the next instance is prefetched like a next head.
  const int bl0 = xcd + MAGAT_NUM_XCD * (slot / hgroups);
  const int istride = MAGAT_NUM_XCD * ((int)gridDim.x / MAGAT_NUM_XCD / hgroups);
  const int head0 = (slot % hgroups) * hpb;
  if (p.run_if && *p.run_if == 0) {
    if (p.book) magat_guard_book_idle(p.book);
    return;
  }
  if (bl0 >= p.B) {
    if (p.book) magat_guard_book(p.book);
    return;
  }

  float* R0 = smem;                      // Q_p, later hop buffer
  float* R1 = R0 + N * RW;               // X_b during the score phase, then U_{K-1} / hop buffer
  float* A = R1 + N * RW;                // mask, then attention (N x lda_a)
  float* c1s = A + N * p.lda_a;
  float* c2s = c1s + ((N + 3) & ~3);
  int* nbr = reinterpret_cast<int*>(c2s + ((N + 3) & ~3));
  unsigned* rmask = reinterpret_cast<unsigned*>(                                  // [N][4] edge bitmask per row, 16-B aligned
      (reinterpret_cast<uintptr_t>(nbr + 128 * (blockDim.x >> 6)) + 15) & ~static_cast<uintptr_t>(15));

  const int t = threadIdx.x, NT = blockDim.x, lane = t & 63, wave = t >> 6, nwaves = NT >> 6;
  const int K = p.K;
  const bool keyquery = p.mode == MAGAT_MODE_KEYQUERY;
  const bool need_att = K > 1 || p.A_opt;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  long long* dbg = (kDebugHooks && p.dbg) ? p.dbg + (long long)bid * 8 : nullptr;
  const int skip = kDebugHooks ? p.skip : 0;
  if (dbg && t == 0) dbg[0] = clock64();
  // Z addressing: row n, column col of the current instance = Zb + n * zrow + zcol(col)
  const int zrow = p.zts ? 128 : p.NC;
  auto zcol = [&](int col) -> long long { return p.zts ? (long long)(col >> 7) * p.zts + (col & 127) : (long long)col; };
  const float* Zb = p.Z + (long long)bl0 * N * zrow;     // current instance (updated by the instance loop)

  // ---- phase 0 (once per workgroup): the GSO edge masks, this wave's x_i rows, the first head's tiles.
  // WIDE path: the Q_p and U_{K-1} tiles ([N][128] floats, rows NC floats apart in Z) travel global -> LDS with the
  // LDS-direct load (global_load_lds_dwordx4: no destination registers, so nothing for the compiler to guard with
  // s_waitcnt vmcnt(0) and nothing to spill).  A workgroup walks hpb heads; while a head's last hop runs, the next
  // head's Q tile streams into the hop buffer that is no longer read, and that head's U_{K-1} tile streams in during
  // its score phase - only the very first tile's latency is exposed.
  // Narrow path (G < 64, hpb == 1): tiles are staged through registers as before.
  f32x4 qst[QG], xst[QG], ust[QF];
  fvec zerov;
#pragma unroll
  for (int e = 0; e < VEC; ++e) zerov[e] = 0.f;
  // "settle": an empty asm that reads and redefines a register.  The compiler waits for the load that produced the
  // value right here (where waiting is harmless) and afterwards no longer tracks it as an outstanding memory
  // result - otherwise its loop-carried bookkeeping guards later uses with s_waitcnt vmcnt(0), which would also
  // wait for the LDS-direct tile loads in flight.
#define MAGAT_SETTLE_F(x) asm volatile("" : "+v"(x))
  // one wave instruction moves 64 consecutive 16-byte chunks of the tile (LDS address = M0 + 16 * lane)
  auto dma_tile = [&](float* dst, const float* zb, int col_off, int tv) {
    const int lane_ = tv & 63;
    const int wave_ = __builtin_amdgcn_readfirstlane(tv >> 6);
    const int total = N * GC;
    for (int g = wave_; g * 64 < total; g += nwaves) {
      const int idx = g * 64 + lane_;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)dst + (unsigned)g * 1024u);
      if (idx < total) {
        const int n = idx / GC, c = idx % GC;
        const float* src = zb + (long long)n * zrow + zcol(col_off) + 4 * c;
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
      }
    }
  };
  auto tiles_landed = [&]() {      // this wave's outstanding loads are done; then everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  auto issue_q = [&](int head, int tl) {
    const int qo = p.qoff + head * G;
#pragma unroll
    for (int q = 0; q < QG; ++q) {
      const int idx = tl + q * NT, n = idx / GC, c = idx % GC;
      qst[q] = *reinterpret_cast<const f32x4*>(Zb + (long long)(n < N ? n : 0) * zrow + zcol(qo) + 4 * c);
    }
  };
  auto issue_u = [&](int head, int tl) {
    const int uo = p.uoff + (head * K + (K - 1)) * F;
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int idx = tl + q * NT, n = idx / FC, c = idx % FC;
      ust[q] = *reinterpret_cast<const f32x4*>(Zb + (long long)(n < N ? n : 0) * zrow + zcol(uo) + 4 * c);
    }
  };
  float* Rq = R0;     // LDS buffer holding the current head's Q tile (WIDE: alternates between heads)
  float* Ru = R1;     // ... and its U_{K-1} tile
  const int rpw = WIDE ? 1 : RPW;
  // bias slice of this lane (hop-phase lane map), loaded once
  fvec biasv = zerov;
  if (p.bias) biasv = *reinterpret_cast<const fvec*>(p.bias + VEC * (WIDE ? lane : lane % LF));
  if constexpr (WIDE) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) MAGAT_SETTLE_F(biasv[e]);
  }

  // fused head-mean: the workgroup walks all P heads of an instance, every wave owns the same output rows for every
  // head, so sum_p (Y_p + bias) is carried in registers and relu(sum / P) is stored once (graphML.py:4663-4667) -
  // same values and summation order as the separate head_mean_relu_kernel, without the [B*N, P*F] round trip
  const bool fuse_mean = WIDE && p.Ymean != nullptr;
  fvec ysum[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h) ysum[h] = zerov;
  for (int sl_ = bl0; sl_ < p.B; sl_ += istride) {  // ---- instance slots of this workgroup
  const int bl = p.order ? p.order[sl_] : sl_;       // (balanced walk: slots -> instances dealt out by edge count)
  const int b = p.b0 + bl;
  int ti = threadIdx.x;              // laundered per instance (see the note at the head loop)
  asm volatile("" : "+v"(ti));
  const int t = ti, lane = ti & 63, wave = ti >> 6;
  long long t_inst = 0;
  if (dbg && t == 0) t_inst = clock64();
  Zb = p.Z + (long long)bl * N * zrow;
  const float* Xb = p.X + (long long)b * N * p.ldx;
  const long long sbase = (long long)b * N * N;
  // WIDE score phase: 8 lanes per graph row (8 rows per wave step, exactly one step per wave since NT >= 8 N); lane es
  // owns chunks es + 8*(q ^ (eg&1)) of a feature row: odd groups start on the other 128-byte half, so the 16-lane
  // ds_read_b128 service groups never collide.  The x_i row of the group is the same for every head: registers.
  constexpr int CP8 = GC / 8 > 0 ? GC / 8 : 1;
  f32x4 xi[CP8];
  if constexpr (WIDE) {
    if (keyquery && need_att) {
      if (sl_ == bl0) dma_tile(Rq, Zb, p.qoff + head0 * G, t);
      const int es0 = lane & 7, eg0 = lane >> 3, par0 = eg0 & 1;
      const int i = 8 * wave + eg0, ir = i < N ? i : 0;
#pragma unroll
      for (int q = 0; q < CP8; ++q)
        xi[q] = *reinterpret_cast<const f32x4*>(Xb + (long long)ir * p.ldx + 4 * (es0 + 8 * (q ^ par0)));
    }
    if (K > 1 && sl_ == bl0) dma_tile(Ru, Zb, p.uoff + (head0 * K + (K - 1)) * F, t);
  } else if (keyquery && need_att) {
    issue_q(head0, t);
#pragma unroll
    for (int q = 0; q < QG; ++q) {
      const int idx = t + q * NT, n = idx / GC, c = idx % GC;
      xst[q] = zero4;
      if (n < N) xst[q] = *reinterpret_cast<const f32x4*>(Xb + (long long)n * p.ldx + 4 * c);
    }
  }
  if (need_att) {
    if constexpr (WIDE) {   // GSO rows -> 128-bit edge masks (one wave per row, coalesced reads, ballot)
      // every wave owns rows wave, wave + nwaves, ... (at most 8 since NT >= 8 N): all their loads are issued
      // first and the ballots run afterwards - one memory latency per instance instead of one per row
      const float sl = p.mode == MAGAT_MODE_GAT_ORIGIN ? 1.f : 0.f;
      if (p.rmask_pre) {      // made by gat_prepare_kernel: 16 bytes per row instead of the GSO row
        const unsigned* src = p.rmask_pre + (long long)b * N * 4;
        for (int idx = t; idx < 4 * N; idx += NT) rmask[idx] = src[idx];
      }
      auto stage_masks = [&](auto tag) {
        if (p.rmask_pre) return;
        typedef decltype(tag) ST;
        const ST* Sp = static_cast<const ST*>(p.S) + sbase;
        for (int hb = 0; hb < 8; hb += 4) {       // two batches of four rows (register budget)
        if (wave + hb * nwaves >= N) break;
        ST v0[4], v1[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int i = wave + (hb + h) * nwaves, ic = i < N ? i : N - 1;
          v0[h] = Sp[(long long)ic * N + (lane < N ? lane : N - 1)];
          v1[h] = Sp[(long long)ic * N + (lane + 64 < N ? lane + 64 : N - 1)];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int i = wave + (hb + h) * nwaves;
          if (i >= N) break;
          bool f0, f1;
          if (sl != 0.f) {      // GAT_origin: |float(S) + I| > 1e-9f
            f0 = fabsf((float)v0[h] + (lane == i ? sl : 0.f)) > 1e-9f;
            f1 = fabsf((float)v1[h] + (lane + 64 == i ? sl : 0.f)) > 1e-9f;
          } else if (sizeof(ST) == 8) {
            f0 = fabs((double)v0[h]) > 1e-9;
            f1 = fabs((double)v1[h]) > 1e-9;
          } else {
            f0 = fabsf((float)v0[h]) > 1e-9f;
            f1 = fabsf((float)v1[h]) > 1e-9f;
          }
          const unsigned long long k0 = __ballot(f0 && lane < N), k1 = __ballot(f1 && lane + 64 < N);
          if (lane == 0) {
            rmask[4 * i + 0] = (unsigned)k0; rmask[4 * i + 1] = (unsigned)(k0 >> 32);
            rmask[4 * i + 2] = (unsigned)k1; rmask[4 * i + 3] = (unsigned)(k1 >> 32);
          }
        }
        }
      };
      if (p.s_is_f64) stage_masks(double{});
      else stage_masks(float{});
    } else {                // mask -> A (1/0)
      for (int idx = t; idx < N * N; idx += NT) {
        const int i = idx / N, j = idx - i * N;
        const float sl = (p.mode == MAGAT_MODE_GAT_ORIGIN && i == j) ? 1.f : 0.f;
        A[i * p.lda_a + j] = is_edge(p.S, sbase + idx, p.s_is_f64, sl) ? 1.f : 0.f;
      }
    }
  }
  if constexpr (WIDE) {
    if (keyquery && need_att) {
#pragma unroll
      for (int q = 0; q < CP8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) MAGAT_SETTLE_F(xi[q][e]);
    }
  }

  for (int hh = 0; hh < hpb; ++hh) {
  const int head = head0 + hh;
  // the thread index is laundered once per head so every per-thread address / lane-map quantity below is
  // recomputed (a few VALU ops) instead of being hoisted out of the head loop into ~50 long-lived registers
  int tl = threadIdx.x;
  asm volatile("" : "+v"(tl));
  const int t = tl, lane = tl & 63, wave = tl >> 6, wl = wave;
  const int es8 = lane & 7, eg8 = lane >> 3, par8 = eg8 & 1;
  // hop-phase lane map.  WIDE: a wave owns one output row, lane -> VEC consecutive features.
  // otherwise: LF lanes per row (16 B each), RPW rows per wave step.
  const int sub = WIDE ? lane : lane % LF, grp = WIDE ? 0 : lane / LF;
  fvec ucur[HMAX];   // per-head
  if (dbg && t == 0 && hh == hpb - 1 && hh > 0) dbg[0] = clock64();   // instrumentation follows the LAST head
  if (need_att && !keyquery)
    for (int n = t; n < N; n += NT) {
      c1s[n] = Zb[(long long)n * zrow + zcol(p.c1off + head)];
      c2s[n] = Zb[(long long)n * zrow + zcol(p.c2off + head)];
    }
  // U rows of this wave's output rows: one per-lane base pointer, rows nwaves * rpw apart (rows past N re-read row 0
  // of the group: harmless, never stored)
  const int ws_head = __builtin_amdgcn_readfirstlane(wl);
  auto load_urows = [&](fvec (&dst)[HMAX], int kk) {
    // (the lane's first row itself is clamped: with RPW rows per wave step - G, F < 64 - the row groups of the upper waves lie
    //  past N altogether, and for the last instance of the batch past the end of Z: found by tools/exp/fuzz_forward.py as a
    //  memory fault at N = 103, G = 16)
    const int r0 = ws_head * rpw + grp;
    const float* base = Zb + (long long)(r0 < N ? r0 : 0) * zrow + zcol(p.uoff + (head * K + kk) * F) + VEC * sub;
    const long long step = (long long)nwaves * rpw * zrow;
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      const bool ok = (ws_head + h * nwaves) * rpw + grp < N;
      dst[h] = *reinterpret_cast<const fvec*>(ok ? base + h * step : base);
    }
  };
  // the first hop's U rows travel during the score phase.  For the very first head of a workgroup they are issued
  // before the wait for the tiles (everything is in flight together: short score phases, e.g. N = 20, do not cover a
  // second memory latency); for later heads after it (the wait must not include them).
  const bool first_head = hh == 0 && sl_ == bl0;
  if (WIDE && first_head) load_urows(ucur, K > 1 ? K - 2 : 0);
  if constexpr (WIDE) {
    // Q tile (prefetched during the previous head's last hop, or above) is in Rq once every wave's loads are done
    // (waited for at the end of that hop; only a workgroup's very first head waits here); the barrier also retires
    // the previous head's reads of Ru and A
    if (first_head || skip) tiles_landed();
    else __syncthreads();
    if ((hh > 0 || sl_ != bl0) && keyquery && need_att && K > 1)
      dma_tile(Ru, Zb, p.uoff + (head * K + (K - 1)) * F, tl);
  } else {
    if (keyquery && need_att) {
#pragma unroll
      for (int q = 0; q < QG; ++q) {
        const int idx = tl + q * NT, n = idx / GC, c = idx % GC;
        if (n < N) {
          *reinterpret_cast<f32x4*>(R0 + n * G + 4 * c) = qst[q];
          *reinterpret_cast<f32x4*>(R1 + n * G + 4 * c) = xst[q];
        }
      }
    }
    __syncthreads();
    if (K > 1) issue_u(head, tl);     // in flight during the score phase
  }
  if (dbg && t == 0 && hh == hpb - 1) dbg[1] = clock64();
  if (dbg && t == 0 && hh == 0) { dbg[4] += clock64() - t_inst; dbg[5] += 1; }   // instance prologue + first barrier
  if (!(WIDE && first_head)) load_urows(ucur, K > 1 ? K - 2 : 0);

  // ---- phase 1: attention rows out of LDS.
  // WIDE (G >= 64): a 16-lane row of the wave owns one graph row (4 rows per wave step); its lanes walk the
  // row's edge bitmask, each edge costing CPL ds_read_b128 + 4*CPL FMA + 4 DPP adds; the masked softmax then
  // runs with lane = neighbour slot (8 slots per lane), reductions again on DPP.  No ds_bpermute anywhere.
  if (need_att && !(skip & 1)) {
    if constexpr (WIDE) {
      const int es = es8, eg = eg8, par = par8;
      for (int ib = 8 * wave; ib < N; ib += 8 * nwaves) {
        const int i = ib + eg;
        const bool iok = i < N;
        const int ir = iok ? i : 0;
        float* Arow = A + ir * p.lda_a;
        const uint4 mk = *reinterpret_cast<const uint4*>(rmask + 4 * ir);
        unsigned w[4] = {iok ? mk.x : 0u, iok ? mk.y : 0u, iok ? mk.z : 0u, iok ? mk.w : 0u};
        // compact softmax state: edge number e of the row is kept by lane (e & 7) of the group in slot e >> 3 (rows of
        // up to 16 edges); sc / sj = score and neighbour index of the lane's two slots
        float sc0 = -__builtin_inff(), sc1 = -__builtin_inff();
        int sj0 = 0, sj1 = 0, ecount = 0;
        if (keyquery) {
          // one walk over the whole 128-bit edge mask (two 64-bit halves), two edges of this row per trip (both
          // neighbour rows in flight together): the wave runs max over its 8 rows of ceil(degree / 2) trips - walking
          // the four 32-bit words one after the other cost the sum over words of the per-word maxima, about twice that
          unsigned long long ma = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
          unsigned long long mb = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
          {
            while (ma | mb) {
              int j0;
              if (ma) { j0 = __builtin_ctzll(ma); ma &= ma - 1; }
              else { j0 = 64 + __builtin_ctzll(mb); mb &= mb - 1; }
              const bool two = (ma | mb) != 0;
              int j1 = j0;
              if (two) {
                if (ma) { j1 = __builtin_ctzll(ma); ma &= ma - 1; }
                else { j1 = 64 + __builtin_ctzll(mb); mb &= mb - 1; }
              }
              // packed fp32 FMAs (v_pk_fma_f32): even / odd elements accumulate separately, summed at the end
              f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
              for (int q = 0; q < CP8; ++q) {
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(Rq + j0 * G + 4 * (es + 8 * (q ^ par)));
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(Rq + j1 * G + 4 * (es + 8 * (q ^ par)));
                const f32x2 xl = {xi[q][0], xi[q][1]}, xh = {xi[q][2], xi[q][3]};
                p0 = __builtin_elementwise_fma(xl, f32x2{q0[0], q0[1]}, p0);
                p1 = __builtin_elementwise_fma(xl, f32x2{q1[0], q1[1]}, p1);
                p0 = __builtin_elementwise_fma(xh, f32x2{q0[2], q0[3]}, p0);
                p1 = __builtin_elementwise_fma(xh, f32x2{q1[2], q1[3]}, p1);
              }
              float d0 = oct_sum(p0[0] + p0[1]);
              float d1 = oct_sum(p1[0] + p1[1]);
              if (es == 0) {        // raw scores for the dense (fallback) softmax below
                Arow[j0] = d0;
                if (two) Arow[j1] = d1;
              }
              {
                const bool mine = (ecount & 7) == es, hi = ecount >= 8;
                if (mine && !hi) { sc0 = d0; sj0 = j0; }
                if (mine && hi) { sc1 = d0; sj1 = j0; }
                const int e1 = ecount + 1;
                const bool mine1 = two && (e1 & 7) == es, hi1 = e1 >= 8;
                if (mine1 && !hi1) { sc0 = d1; sj0 = j1; }
                if (mine1 && hi1) { sc1 = d1; sj1 = j1; }
                ecount += two ? 2 : 1;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        // Compact softmax (KeyQuery, every row of the wave has <= 16 edges, attention not exported): max / sum over the
        // <= 2 scores per lane, written to the edge positions only.  The zero entries of the row are written once per
        // instance (first head): every head of an instance has the same edge positions.  ~40 VALU ops instead of ~350.
        if (keyquery && !p.A_opt && !__any(ecount > 16)) {
          const float mxc = oct_max(fmaxf(sc0, sc1));
          const float e0 = sc0 > -__builtin_inff() ? __expf(sc0 - mxc) : 0.f;
          const float e1 = sc1 > -__builtin_inff() ? __expf(sc1 - mxc) : 0.f;
          const float sm = oct_sum(e0 + e1);
          const float invc = sm > 0.f ? 1.f / sm : 0.f;
          if (hh == 0 && iok) {
            for (int j = es; j < N; j += 8) Arow[j] = 0.f;
          }
          if (sc0 > -__builtin_inff()) Arow[sj0] = e0 * invc;
          if (sc1 > -__builtin_inff()) Arow[sj1] = e1 * invc;
          continue;
        }
        // masked softmax, lane = neighbour slot j = es + 8*r.  Branch-free: all 16 slots are read back to back
        // (one LDS wait instead of sixteen predicated read+wait blocks) and selected by the edge bits afterwards;
        // slots without an edge hold stale or foreign data that never enters the arithmetic.
        float v[16];
        float mx = -__builtin_inff();
        const float c2 = keyquery ? 0.f : c2s[ir];
        const float* srow = keyquery ? Arow : c1s;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = es + 8 * r;
          v[r] = srow[j < N ? j : N - 1];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = es + 8 * r;
          const bool f = (w[r >> 2] >> (j & 31)) & 1u;      // bits at and beyond N are never set
          float e = v[r];
          if (!keyquery) {
            e += c2;
            e = e > 0.f ? e : 0.2f * e;
          }
          v[r] = f ? e : -__builtin_inff();
          mx = fmaxf(mx, v[r]);
        }
        mx = oct_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = v[r] > -__builtin_inff() ? __expf(v[r] - mx) : 0.f;
          sum += v[r];
        }
        sum = oct_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        float* ao = p.A_opt ? p.A_opt + (((long long)b * p.P + head) * N + ir) * N : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = es + 8 * r;
          if (8 * r < N && iok && j < N) {
            const float a = v[r] * inv;
            Arow[j] = a;
            if (ao) ao[j] = a;
          }
        }
      }
    } else {
    int* nb = nbr + wave * 128;
    const int es = lane % LE, eg = lane / LE;
    for (int i = wave; i < N; i += nwaves) {
      const bool m0 = lane < N && A[i * p.lda_a + lane] != 0.f;
      const bool m1 = lane + 64 < N && A[i * p.lda_a + lane + 64] != 0.f;
      const unsigned long long k0 = __ballot(m0), k1 = __ballot(m1);
      const int deg0 = __popcll(k0), deg = deg0 + __popcll(k1);
      float v0 = 0.f, v1 = 0.f;
      if (keyquery) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (m0) nb[__popcll(k0 & lt)] = lane;
        if (m1) nb[deg0 + __popcll(k1 & lt)] = lane + 64;
        f32x4 xi[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) xi[q] = *reinterpret_cast<const f32x4*>(R1 + i * G + 4 * (es + LE * q));
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < deg; t0 += EPS) {
          const int ei = t0 + eg;
          const bool ok = ei < deg;
          const int j = ok ? nb[ei] : 0;
          float d = 0.f;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(R0 + j * G + 4 * (es + LE * q));
            d = fmaf(xi[q][0], qv[0], d);
            d = fmaf(xi[q][1], qv[1], d);
            d = fmaf(xi[q][2], qv[2], d);
            d = fmaf(xi[q][3], qv[3], d);
          }
          if (LE == 16) {
            d = row16_sum(d);
          } else {
#pragma unroll
            for (int o = LE / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
          }
          if (ok && es == 0) A[i * p.lda_a + j] = d;
        }
        __builtin_amdgcn_wave_barrier();
        if (m0) v0 = A[i * p.lda_a + lane];
        if (m1) v1 = A[i * p.lda_a + lane + 64];
      } else {
        const float c2 = c2s[i];
        if (m0) { const float e = c1s[lane] + c2; v0 = e > 0.f ? e : 0.2f * e; }
        if (m1) { const float e = c1s[lane + 64] + c2; v1 = e > 0.f ? e : 0.2f * e; }
      }
      const float ninf = -__builtin_inff();
      const float mx = wave_max(fmaxf(m0 ? v0 : ninf, m1 ? v1 : ninf));
      const float e0 = m0 ? expf(v0 - mx) : 0.f, e1 = m1 ? expf(v1 - mx) : 0.f;
      const float sum = wave_sum(e0 + e1);
      const float a0 = m0 ? e0 / sum : 0.f, a1 = m1 ? e1 / sum : 0.f;
      if (lane < N) A[i * p.lda_a + lane] = a0;
      if (lane + 64 < N) A[i * p.lda_a + lane + 64] = a1;
      if (p.A_opt) {
        float* ao = p.A_opt + (((long long)b * p.P + head) * N + i) * N;
        if (lane < N) ao[lane] = a0;
        if (lane + 64 < N) ao[lane + 64] = a1;
      }
    }
    }
  }
  if constexpr (WIDE) {
    tiles_landed();     // attention complete, U_{K-1} tile in Ru, first hop's U rows in registers
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
#pragma unroll
      for (int e = 0; e < VEC; ++e) MAGAT_SETTLE_F(ucur[h][e]);
  } else {
    __syncthreads();
  }
  if (dbg && t == 0 && hh == hpb - 1) dbg[2] = clock64();
  if (!WIDE && K > 1) {      // narrow path: X_b is dead, the deepest hop operand takes its place in R1
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int idx = t + q * NT, n = idx / FC, c = idx % FC;
      if (n < N) *reinterpret_cast<f32x4*>(R1 + n * F + 4 * c) = ust[q];
    }
    __syncthreads();
  }

  // ---- phase 2: Horner hops  T <- U_k + A^T T   (k = K-2 .. 0), last one fused with bias/ReLU/store.
  // WIDE: one wave per output row j.  The attention column A[:,j] is read once (2 conflict-free ds_read_b32
  // per lane), its non-zeros become a wave-uniform 128-bit mask (ballot), and the gather loop is scalar:
  // s_ff1 -> v_readlane (weight) -> one ds_read of the neighbour's feature row -> VEC FMAs, two neighbours
  // in flight per iteration.
  const unsigned long long gmask = LF == 64 ? ~0ull : ((1ull << LF) - 1ull);
  float* Rold = Ru;
  float* Rnew = Rq;
  // rows of one hop; LAST is a compile-time tag (std::true_type / false_type): the final hop is a separate copy of
  // this code with no `unext` registers anywhere near it, so the LDS-direct prefetch issued in front of it is not
  // caught by compiler-inserted s_waitcnt vmcnt(0) guards for registers with pending loads
  auto hop_rows = [&](auto last_tag, const float* Rold_, float* Rnew_) {
    constexpr bool last = decltype(last_tag)::value;
    // wave-uniform row bookkeeping lives in SGPRs; per-lane bases are computed once per hop
    const int ws = __builtin_amdgcn_readfirstlane(wl);
    const float* Ac0 = A + (lane < N ? lane : 0) * p.lda_a;            // attention column walkers (lanes i, i + 64)
    const float* Ac1 = A + (lane + 64 < N ? lane + 64 : 0) * p.lda_a;
    const bool lv0 = lane < N, lv1 = lane + 64 < N;
    float* yrow = p.Y + ((long long)b * N + ws * rpw + grp) * p.ldy + head * F + VEC * sub;
    const long long ystep = (long long)nwaves * rpw * p.ldy;
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      const int jb = (ws + h * nwaves) * rpw;
      if (jb >= N) break;
      const int j = jb + grp;
      const bool jok = j < N;
      fvec acc = zerov;
      if (K > 1) {
        if constexpr (WIDE) {
          // attention column j: lanes i and i + 64 (unconditional reads from clamped rows, masked by lv0 / lv1)
          const float c0 = Ac0[j], c1 = Ac1[j];
          const unsigned long long k0 = __ballot(lv0 && c0 != 0.f), k1 = __ballot(lv1 && c1 != 0.f);
          const float* Tl = Rold_ + VEC * lane;
          // (reading up to 8 neighbour rows before the first FMA was tried: 16.5k instead of 10k cycles per hop - the
          // hop is bound by VALU issue, not by LDS latency, and the batched form needs more instructions per row)
          auto gather = [&](unsigned long long km, float cv, int base) {
            while (km) {          // two neighbour rows in flight per trip (weights via v_readlane)
              const int i0 = __builtin_ctzll(km);
              km &= km - 1;
              const float a0 = lane_bcast(cv, i0);
              const fvec t0 = *reinterpret_cast<const fvec*>(Tl + (base + i0) * F);
              if (km) {
                const int i1 = __builtin_ctzll(km);
                km &= km - 1;
                const float a1 = lane_bcast(cv, i1);
                const fvec t1 = *reinterpret_cast<const fvec*>(Tl + (base + i1) * F);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(a1, t1[e], fmaf(a0, t0[e], acc[e]));
              } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(a0, t0[e], acc[e]);
              }
            }
          };
          gather(k0, c0, 0);
          gather(k1, c1, 64);
        } else {
          for (int r0 = 0; r0 < N; r0 += LF) {
            const int i = r0 + sub;
            const float a = (jok && i < N) ? A[i * p.lda_a + j] : 0.f;
            const unsigned long long bal = __ballot(a != 0.f);
            unsigned long long mine = (bal >> (grp * LF)) & gmask;
            while (mine) {
              const int ii = r0 + __builtin_ctzll(mine);
              mine &= mine - 1;
              const float av = A[ii * p.lda_a + j];
              const fvec tv = *reinterpret_cast<const fvec*>(Rold_ + ii * F + VEC * sub);
#pragma unroll
              for (int e = 0; e < VEC; ++e) acc[e] = fmaf(av, tv[e], acc[e]);
            }
          }
        }
      }
      fvec res = ucur[h] + acc;
      if (!jok) continue;
      if constexpr (last) {
        res += biasv;
        if (p.concat) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) res[e] = magat_relu(res[e]);
        }
        if constexpr (WIDE) ucur[h] = res;      // stored after the tile wait below
        else *reinterpret_cast<fvec*>(yrow + h * ystep) = res;
      } else {
        *reinterpret_cast<fvec*>(Rnew_ + j * F + VEC * sub) = res;
      }
    }
    if constexpr (last && WIDE) {
      // The prefetched tile is waited for HERE, while the only outstanding requests are its LDS-direct loads (they had
      // the whole hop to land), and the Y rows are stored afterwards: a wait at the next head's top would also cover
      // these stores' write acknowledgements (loads and stores share vmcnt) - 5-7 k cycles per head.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (fuse_mean) {
        const float invp = 1.f / (float)p.P;      // (the separate kernel divides by P as well: s / fp)
        float* mrow = p.Ymean + ((long long)b * N + ws * rpw + grp) * p.ldym + VEC * sub;
        const long long mstep = (long long)nwaves * rpw * p.ldym;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
          ysum[h] = hh == 0 ? ucur[h] : ysum[h] + ucur[h];
          if (hh == hpb - 1 && (ws + h * nwaves) * rpw + grp < N) {
            fvec o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = magat_relu(ysum[h][e] / (float)p.P);
            *reinterpret_cast<fvec*>(mrow + h * mstep) = o;
          }
        }
        (void)invp;
      } else {
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
          if ((ws + h * nwaves) * rpw + grp < N) *reinterpret_cast<fvec*>(yrow + h * ystep) = ucur[h];
      }
    }
  };
  if (!(skip & 2)) {
    for (int k = K - 2; k >= 1; --k) {      // all hops but the last
      int tk = threadIdx.x;      // laundered per hop: load addresses are computed at the point of use
      asm volatile("" : "+v"(tk));
      fvec unext[HMAX];          // next hop's U rows: in flight during this hop
      load_urows(unext, k - 1);
      hop_rows(std::false_type{}, Rold, Rnew);
#pragma unroll
      for (int h = 0; h < HMAX; ++h) ucur[h] = unext[h];
      __syncthreads();
      if constexpr (WIDE) {
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
#pragma unroll
          for (int e = 0; e < VEC; ++e) MAGAT_SETTLE_F(ucur[h][e]);
      }
      if (dbg && t == 0 && hh == hpb - 1) dbg[3 + (K - 2 - k)] = clock64();
      float* tmp = Rold; Rold = Rnew; Rnew = tmp;
    }
    if constexpr (WIDE) {
      const bool more_heads = hh + 1 < hpb;
      if (more_heads || sl_ + istride < p.B) {
        // Rnew is not read any more: the first tile of the next head (or of the next instance's first head) streams
        // into it during the last hop (every register the rows below consume has been settled: nothing waits for it)
        int tk = threadIdx.x;
        asm volatile("" : "+v"(tk));
        const float* zbn =
            more_heads ? Zb : p.Z + (long long)(p.order ? p.order[sl_ + istride] : sl_ + istride) * N * zrow;
        const int hn = more_heads ? head + 1 : head0;
        if (keyquery && need_att) dma_tile(Rnew, zbn, p.qoff + hn * G, tk);
        else if (K > 1) dma_tile(Rnew, zbn, p.uoff + (hn * K + (K - 1)) * F, tk);
      }
    }
    hop_rows(std::true_type{}, Rold, Rnew);
  }
  if constexpr (WIDE) {      // buffer roles of the next head (its first tile went to Rnew)
    if (keyquery && need_att) { Rq = Rnew; Ru = Rold; }
    else { Ru = Rnew; Rq = Rold; }
  }
  }  // heads
  }  // instances
  if (dbg) {
    __syncthreads();
    if (t == 0) { dbg[6] = clock64(); dbg[7] = wall_clock64(); }
  }
  if (p.book) magat_guard_book(p.book);
}

// ---- The float32 form of the layer where gat_dense_kernel's tiles no longer fit: G = F = 128 on 106 .. 128 agents, KeyQuery - the
// range guard's re-run behind gat_mid.hip's 128-wide form (a launch that returns at once unless the flag is set; when it does run,
// speed is not its job).  From the hoisted maps Z like gat_dense_kernel; a workgroup walks (instance, head) slots with
//   V  [N][129]    Q_p during the scores, then the hop buffer (acc_{k+1})
//   At [N][N | 1]  At[i][j] = a_ij: row i's softmax over its edges j (graphML.py:1262-1286)
// scores e_ij = x_i . Q_p[j]: a wave per row i, lanes j and j + 64; hops acc_k[j] = U_pk[j] + sum_i a_ij acc_{k+1}[i]
// (graphML.py:1744-1775 in Horner form): a wave owns rows j = wave + 4 m, lanes the columns c and c + 64, the new rows stay in
// registers until every wave is done reading the old ones.
__global__ __launch_bounds__(256) void gat_slim_kernel(const GatParams p) {
  extern __shared__ __align__(16) float smem[];
  if (p.run_if && *p.run_if == 0) {
    if (p.book) magat_guard_book_idle(p.book);
    return;
  }
  constexpr int G = 128, F = 128, LDV = 129, MR = 32;
  const int N = p.N, K = p.K, P = p.P, lda = N | 1;
  float* V = smem;
  float* At = V + N * LDV;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int zrow = p.zts ? 128 : p.NC;
  auto zcol = [&](int col) -> long long { return p.zts ? (long long)(col >> 7) * p.zts + (col & 127) : (long long)col; };
  const int c0 = lane, c1 = lane + 64;
  const float bias0 = p.bias ? p.bias[c0] : 0.f, bias1 = p.bias ? p.bias[c1] : 0.f;
  for (int slot = blockIdx.x; slot < p.B * P; slot += gridDim.x) {
    const int bl = slot / P, head = slot % P;
    const long long b = p.b0 + bl;
    const float* Zb = p.Z + (long long)bl * N * zrow;
    const float* Xb = p.X + b * N * p.ldx;
    auto load_tile = [&](int col) {      // V <- columns col .. col + 127 of this instance's Z rows
      const float* src = Zb + zcol(col);
      for (int idx = t; idx < N * 32; idx += 256) {
        const int n = idx >> 5, c4 = idx & 31;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long long)n * zrow + 4 * c4);
        float* d = V + n * LDV + 4 * c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
      }
    };
    __syncthreads();      // every wave is done with the previous slot's V / At
    if (K > 1) {
      load_tile(p.qoff + head * G);
      __syncthreads();
      const int j0 = lane, j1 = lane + 64 < N ? lane + 64 : N - 1;
      const int j0c = j0 < N ? j0 : N - 1;
      for (int i = wave; i < N; i += 4) {
        const float* xi = Xb + (long long)i * p.ldx;
        float e0 = 0.f, e1 = 0.f;
        for (int g = 0; g < G; ++g) {
          const float xg = xi[g];
          e0 = __builtin_fmaf(xg, V[j0c * LDV + g], e0);
          e1 = __builtin_fmaf(xg, V[j1 * LDV + g], e1);
        }
        const long long srow = (b * N + i) * N;
        const bool m0 = j0 < N && is_edge(p.S, srow + j0c, p.s_is_f64);
        const bool m1 = lane + 64 < N && is_edge(p.S, srow + j1, p.s_is_f64);
        float mx = fmaxf(m0 ? e0 : -__builtin_inff(), m1 ? e1 : -__builtin_inff());
        for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float x0 = m0 ? __expf(e0 - mx) : 0.f, x1 = m1 ? __expf(e1 - mx) : 0.f;
        float sum = x0 + x1;
        for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;      // (a row without edges: all zeros, graphML.py:1286)
        if (j0 < N) At[i * lda + j0] = x0 * inv;
        if (lane + 64 < N) At[i * lda + lane + 64] = x1 * inv;
      }
      __syncthreads();      // At complete, Q no longer read
    }
    load_tile(p.uoff + (head * K + (K - 1)) * F);
    __syncthreads();
    float r0[MR], r1[MR];
    for (int k = K - 2; k >= 0; --k) {
      const float* uk = Zb + zcol(p.uoff + (head * K + k) * F);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int j = wave + 4 * m, jc = j < N ? j : N - 1;
        r0[m] = uk[(long long)jc * zrow + c0];
        r1[m] = uk[(long long)jc * zrow + c1];
      }
      for (int i = 0; i < N; ++i) {
        const float v0 = V[i * LDV + c0], v1 = V[i * LDV + c1];
        const float* arow = At + i * lda + wave;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const float a = wave + 4 * m < N ? arow[4 * m] : 0.f;
          r0[m] = __builtin_fmaf(a, v0, r0[m]);
          r1[m] = __builtin_fmaf(a, v1, r1[m]);
        }
      }
      if (k > 0) {
        __syncthreads();      // every wave is done reading acc_{k+1}
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const int j = wave + 4 * m;
          if (j < N) { V[j * LDV + c0] = r0[m]; V[j * LDV + c1] = r1[m]; }
        }
        __syncthreads();
      }
    }
    // output rows: acc_0 + bias; ReLU here (concat) or after the head mean (head_mean_relu_kernel)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int j = wave + 4 * m;
      if (j >= N) continue;
      float y0 = (K > 1 ? r0[m] : V[j * LDV + c0]) + bias0, y1 = (K > 1 ? r1[m] : V[j * LDV + c1]) + bias1;
      if (p.concat) { y0 = magat_relu(y0); y1 = magat_relu(y1); }
      float* yrow = p.Y + (b * N + j) * p.ldy + head * F;
      yrow[c0] = y0;
      yrow[c1] = y1;
    }
  }
  if (p.book) magat_guard_book(p.book);
}


// ---- The range guard's re-run for FEW instances (the closed-loop step of one planning instance: agents/..._GAT.py:1030-1055) as
// ONE launch: behind a one-launch graph kernel the predicated float32 form was two or three launches that return at once - maps
// GEMM, graph kernel (, head mean) - 4.5 us each of a 66 us step.  This kernel is all of them: a workgroup owns an instance, forms
// the maps it needs itself (Q_p and U_pk tiles = X Bt^T + column bias, float32 FMAs from the float32 pack), walks the heads and
// merges them.  KeyQuery, G = F in {32, 64, 128}, N <= 128; gat_slim_kernel's layout: V [N][G + 1], At [N][N | 1].  Speed is not
// its job: it runs when a checkpoint's values left the f16 planes' range.
// TAIL (CO > 0): the skinny float32 layer that consumes the graph layer's rows (the action head, CO outputs; skinny_rows.h) rides
// in the same launch - each workgroup computes it for the rows of its instances, after the re-run if there was one.
template <int G, int CO>
__global__ __launch_bounds__(256) void gat_rerun_small_kernel(const GatParams p, const float* __restrict__ Bt,
                                                              const float* __restrict__ cbias, const MagatSkinnyParams tail) {
  extern __shared__ __align__(16) float smem[];
  // The launch has max(B, row groups of the tail) workgroups.  Flag clear (every forward of a sane checkpoint): all of them share
  // the tail's rows like skinny_gemm_kernel's own launch.  Flag set: workgroup b < B recomputes instance b and then the tail's
  // rows of THAT instance (they depend on nothing else); the others only arrive at the bookkeeping.
  if (p.run_if && *p.run_if == 0) {
    if constexpr (CO > 0) {
      magat_skinny_stage_weights<CO>(tail, smem);
      __syncthreads();
      magat_skinny_rows<CO>(tail, smem, (long long)blockIdx.x * 16, tail.M, (long long)gridDim.x * 16);
    }
    if (p.book) magat_guard_book_idle(p.book);
    return;
  }
  if ((int)blockIdx.x >= p.B) {
    if (p.book) magat_guard_book(p.book);
    return;
  }
  auto run_tail = [&]() {
    if constexpr (CO > 0) {
      __threadfence_block();
      __syncthreads();      // the re-run's rows are written, its LDS is free
      magat_skinny_stage_weights<CO>(tail, smem);
      __syncthreads();
      for (int b = blockIdx.x; b < p.B; b += gridDim.x)
        magat_skinny_rows<CO>(tail, smem, (long long)b * p.N, (long long)(b + 1) * p.N, 16);
    }
  };
  constexpr int F = G, LDV = G + 1, MR = 32, CPL = G > 64 ? 2 : 1;      // columns per lane
  const int N = p.N, K = p.K, P = p.P, lda = N | 1;
  float* V = smem;
  float* At = V + N * LDV;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool cok = lane < G;                       // (G = 32: half the lanes own a column)
  const int c0 = cok ? lane : 0, c1 = lane + 64;   // c1 only when CPL == 2
  const float bias0 = p.bias ? p.bias[c0] : 0.f, bias1 = (p.bias && CPL == 2) ? p.bias[c1] : 0.f;
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    const float* Xb = p.X + (long long)b * N * p.ldx;
    // V <- rows of X times rows col .. col + G - 1 of Bt (+ column bias)
    auto make_tile = [&](int col) {
      for (int idx = t; idx < N * G; idx += 256) {
        const int n = idx / G, c = idx - n * G;
        const float* xr = Xb + (long long)n * p.ldx;
        const float* br = Bt + (long long)(col + c) * G;
        float acc = cbias[col + c];
        for (int g = 0; g < G; g += 4) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + g), bv = *reinterpret_cast<const f32x4*>(br + g);
          acc = __builtin_fmaf(xv[0], bv[0], acc);
          acc = __builtin_fmaf(xv[1], bv[1], acc);
          acc = __builtin_fmaf(xv[2], bv[2], acc);
          acc = __builtin_fmaf(xv[3], bv[3], acc);
        }
        V[n * LDV + c] = acc;
      }
    };
    float s0[MR], s1[MR];      // head mean: running sums of (acc_0 + bias) over the heads, in head order
#pragma unroll
    for (int m = 0; m < MR; ++m) { s0[m] = 0.f; s1[m] = 0.f; }
    for (int head = 0; head < P; ++head) {
      __syncthreads();      // every wave is done with the previous head's V / At
      if (K > 1) {
        make_tile(p.qoff + head * G);
        __syncthreads();
        const int j0c = lane < N ? lane : N - 1, j1 = lane + 64 < N ? lane + 64 : N - 1;
        for (int i = wave; i < N; i += 4) {
          const float* xi = Xb + (long long)i * p.ldx;
          float e0 = 0.f, e1 = 0.f;
          for (int g = 0; g < G; ++g) {
            const float xg = xi[g];
            e0 = __builtin_fmaf(xg, V[j0c * LDV + g], e0);
            e1 = __builtin_fmaf(xg, V[j1 * LDV + g], e1);
          }
          const long long srow = ((long long)b * N + i) * N;
          const bool m0 = lane < N && is_edge(p.S, srow + j0c, p.s_is_f64);
          const bool m1 = lane + 64 < N && is_edge(p.S, srow + j1, p.s_is_f64);
          float mx = fmaxf(m0 ? e0 : -__builtin_inff(), m1 ? e1 : -__builtin_inff());
          for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
          const float x0 = m0 ? __expf(e0 - mx) : 0.f, x1 = m1 ? __expf(e1 - mx) : 0.f;
          float sum = x0 + x1;
          for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
          const float inv = sum > 0.f ? 1.f / sum : 0.f;
          if (lane < N) At[i * lda + lane] = x0 * inv;
          if (lane + 64 < N) At[i * lda + lane + 64] = x1 * inv;
        }
        __syncthreads();
      }
      make_tile(p.uoff + (head * K + (K - 1)) * F);
      __syncthreads();
      float r0[MR], r1[MR];
      for (int k = K - 2; k >= 0; --k) {
        // this wave's rows of U_pk, straight into the accumulators (the same float32 sums make_tile forms)
        const int ucol = p.uoff + (head * K + k) * F;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const int j = wave + 4 * m, jc = j < N ? j : N - 1;
          const float* xr = Xb + (long long)jc * p.ldx;
          float a0 = cbias[ucol + c0], a1 = CPL == 2 ? cbias[ucol + c1] : 0.f;
          const float* b0r = Bt + (long long)(ucol + c0) * G;
          const float* b1r = Bt + (long long)(ucol + (CPL == 2 ? c1 : c0)) * G;
          for (int g = 0; g < G; ++g) {
            const float xg = xr[g];
            a0 = __builtin_fmaf(xg, b0r[g], a0);
            if (CPL == 2) a1 = __builtin_fmaf(xg, b1r[g], a1);
          }
          r0[m] = a0;
          r1[m] = a1;
        }
        for (int i = 0; i < N; ++i) {
          const float v0 = V[i * LDV + c0], v1 = CPL == 2 ? V[i * LDV + c1] : 0.f;
          const float* arow = At + i * lda + wave;
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            const float a = wave + 4 * m < N ? arow[4 * m] : 0.f;
            r0[m] = __builtin_fmaf(a, v0, r0[m]);
            r1[m] = __builtin_fmaf(a, v1, r1[m]);
          }
        }
        if (k > 0) {
          __syncthreads();
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            const int j = wave + 4 * m;
            if (j < N && cok) { V[j * LDV + c0] = r0[m]; if (CPL == 2) V[j * LDV + c1] = r1[m]; }
          }
          __syncthreads();
        }
      }
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int j = wave + 4 * m;
        if (j >= N) continue;
        const float y0 = (K > 1 ? r0[m] : V[j * LDV + c0]) + bias0;
        const float y1 = CPL == 2 ? (K > 1 ? r1[m] : V[j * LDV + c1]) + bias1 : 0.f;
        if (p.concat) {
          float* yrow = p.Ymean + ((long long)b * N + j) * p.ldym + head * F;
          if (cok) yrow[c0] = magat_relu(y0);
          if (CPL == 2) yrow[c1] = magat_relu(y1);
        } else {
          s0[m] += y0;
          s1[m] += y1;
          if (head == P - 1) {      // (graphML.py:4663-4667: mean over the heads, then the nonlinearity)
            float* yrow = p.Ymean + ((long long)b * N + j) * p.ldym;
            if (cok) yrow[c0] = magat_relu(s0[m] / (float)P);
            if (CPL == 2) yrow[c1] = magat_relu(s1[m] / (float)P);
          }
        }
      }
    }
  }
  run_tail();
  if (p.book) magat_guard_book(p.book);
}

constexpr long long GAT_RERUN_SMALL_UNITS = 64;      // instances x heads up to which the re-run is this one launch


// mean over heads then ReLU (graphML.py:4663-4667)
__global__ void head_mean_relu_kernel(const float* __restrict__ ytmp, float* __restrict__ y, long long M, int P,
                                      int F, int ldy, const int* __restrict__ run_if, int* book) {
  if (run_if && *run_if == 0) {
    if (book) magat_guard_book_idle(book);
    return;
  }
  const int FC = F / 4;
  const long long total = M * FC;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long m = idx / FC;
    const int c = (int)(idx - m * FC);
    f32x4 s = *reinterpret_cast<const f32x4*>(ytmp + m * (long long)P * F + 4 * c);
    for (int q = 1; q < P; ++q) s += *reinterpret_cast<const f32x4*>(ytmp + (m * P + q) * (long long)F + 4 * c);
    const float fp = (float)P;
    f32x4 r = {magat_relu(s[0] / fp), magat_relu(s[1] / fp), magat_relu(s[2] / fp), magat_relu(s[3] / fp)};
    *reinterpret_cast<f32x4*>(y + m * ldy + 4 * c) = r;
  }
  if (book) magat_guard_book(book);
}

// ---- weight packing: Bt [NC][G] + column bias [NC]
struct PackLayout {
  int NC, qoff, uoff, c1off, c2off;
};
PackLayout pack_layout(int G, int F, int K, int P, int mode) {
  PackLayout L;
  if (mode == MAGAT_MODE_KEYQUERY) {
    L.qoff = 0;
    L.uoff = P * G;
    L.c1off = L.c2off = 0;
    L.NC = P * G + P * K * F;
  } else if (mode == MAGAT_MODE_GNN) {     // filter taps only
    L.qoff = 0;
    L.uoff = 0;
    L.c1off = L.c2off = 0;
    L.NC = (P * K * F + 31) & ~31;
  } else {
    L.qoff = 0;
    L.uoff = 0;
    L.c1off = P * K * F;
    L.c2off = L.c1off + P;
    L.NC = (L.c2off + P + 31) & ~31;   // multiple of 32: the maps GEMM can always use the bf16 matrix-core tiles
  }
  return L;
}

__global__ void pack_kernel(const float* __restrict__ weight, const float* __restrict__ wbias,
                            const float* __restrict__ mixer, const float* __restrict__ taps,
                            float* __restrict__ packed, int G, int F, int K, int P, int mode, PackLayout L) {
  float* Bt = packed;
  float* cb = packed + (long long)L.NC * G;
  // bf16x3 planes of Bt for the split-MFMA GEMM (raw bf16 bits), 16-byte aligned behind the column bias
  unsigned short* Bs = reinterpret_cast<unsigned short*>(packed + (((long long)L.NC * (G + 1) + 3) & ~3LL));
  unsigned short* Hs = reinterpret_cast<unsigned short*>(packed + magat_gat_f16_block_offset(L.NC, G));
  // the same two planes once more in MFMA-fragment order for gat_mfma.hip (G = 128): 128-row blocks of Bt, per block
  // [32-row tile 4][k step 8][plane 2][lane 64][8 halfs], lane = row % 32 + 32 * (k % 16 / 8)
  // (KeyQuery only: the rank-1 modes' stream at the same offset is written by pack_frag_rank1_kernel, with its own size)
  unsigned short* Fs = (G == 128 && (L.NC & 127) == 0 && mode == MAGAT_MODE_KEYQUERY)
                           ? reinterpret_cast<unsigned short*>(packed + magat_gat_frag_offset(L.NC, G)) : nullptr;
  const long long total = (long long)L.NC * G;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total + L.NC;
       idx += (long long)gridDim.x * blockDim.x) {
    if (idx >= total) {  // column bias
      const int col = (int)(idx - total);
      float v = 0.f;
      if (mode == MAGAT_MODE_GAT_MODIFIED && col >= L.c1off && col < L.c2off + P) {   // GAT_origin has no weight_bias
        const int which = col >= L.c2off, hp = which ? col - L.c2off : col - L.c1off;
        for (int f = 0; f < F; ++f) v = fmaf(mixer[(long long)hp * 2 * F + which * F + f], wbias[hp * F + f], v);
      }
      cb[col] = v;
      continue;
    }
    const int col = (int)(idx / G), g = (int)(idx % G);
    float v = 0.f;
    if (mode == MAGAT_MODE_KEYQUERY && col < L.uoff) {
      v = weight[(long long)col * G + g];  // (P,1,G,G): row p*G+g' = W_p[g',:]
    } else if (col >= L.uoff && col < L.uoff + P * K * F) {
      const int r = col - L.uoff, hp = r / (K * F), k = (r / F) % K, f = r % F;
      if (mode == MAGAT_MODE_GAT_ORIGIN)
        // scalar taps (E=1,K) x W: the reference reshapes permute(0,3,1,2)(W) = (P,G,E,F) straight into (P,F,E,1,G)
        // (graphML.py:1967-1969), so with F == G the filter is W TRANSPOSED: h[p,f,k,g] = h_k * W[p,0,g,f]
        v = taps[k] * weight[((long long)hp * F + g) * G + f];
      else
        v = taps[(((long long)hp * F + f) * K + k) * G + g];  // (P,F,1,K,G)
    } else if (mode != MAGAT_MODE_KEYQUERY && mode != MAGAT_MODE_GNN && col >= L.c1off && col < L.c2off + P) {
      const int which = col >= L.c2off, hp = which ? col - L.c2off : col - L.c1off;
      for (int f = 0; f < F; ++f)
        v = fmaf(mixer[(long long)hp * 2 * F + which * F + f], weight[((long long)hp * F + f) * G + g], v);
    }
    Bt[idx] = v;
    const unsigned short h1 = magat_bf16_rne(v);
    const float r1 = v - magat_bf16_f32(h1);
    const unsigned short h2 = magat_bf16_rne(r1);
    Bs[idx] = h1;
    Bs[total + idx] = h2;
    Bs[2 * total + idx] = magat_bf16_rne(r1 - magat_bf16_f32(h2));
    // f16x2 planes of v * 2^8 (fixed scale: |v| up to 255 representable, residual plane normal down to |v| ~ 5e-4,
    // absolute error floor 1e-10 below that) followed by the inverse scale: the "f16x3" operand of the maps GEMM
    const float vs = v * 256.f;
    const _Float16 g1 = (_Float16)vs;
    const _Float16 g2 = (_Float16)(vs - (float)g1);
    Hs[idx] = __builtin_bit_cast(unsigned short, g1);
    Hs[total + idx] = __builtin_bit_cast(unsigned short, g2);
    if (Fs) {
      const int r = col & 127;
      const long long fo = (long long)(col >> 7) * 32768 + (((r >> 5) * 8 + (g >> 4)) * 2) * 512 +
                           ((r & 31) + 32 * ((g & 15) >> 3)) * 8 + (g & 7);
      Fs[fo] = __builtin_bit_cast(unsigned short, g1);
      Fs[fo + 512] = __builtin_bit_cast(unsigned short, g2);
    }
    if (idx == 0) *reinterpret_cast<float*>(Hs + 2 * total) = 1.f / 256.f;
  }
}

// GAT_modified / GAT_origin at G = F = 128: the weight stream of the one-launch kernel (gat_mfma.hip MODE 1) in the SAME shape
// as KeyQuery's - P blocks of 128 x 128 for G1, then P K tap blocks - as fragment-major f16 planes of 2^8 v.  The G1 block of
// head p holds the two score vectors a1 W_p (row 0) and a2 W_p (row 1), zeros elsewhere; kconst[p] = a1 . wb + a2 . wb.
__global__ void pack_frag_rank1_kernel(const float* __restrict__ weight, const float* __restrict__ wbias,
                                       const float* __restrict__ mixer, const float* __restrict__ taps,
                                       unsigned short* __restrict__ Fs, float* __restrict__ kconst, int K, int P, int mode) {
  constexpr int G = 128, F = 128;
  const long long total = (long long)(P * G + P * K * F) * G;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total + P;
       idx += (long long)gridDim.x * blockDim.x) {
    if (idx >= total) {
      const int hp = (int)(idx - total);
      float v = 0.f;
      if (mode == MAGAT_MODE_GAT_MODIFIED && wbias)
        for (int f = 0; f < F; ++f)
          v = fmaf(mixer[(long long)hp * 2 * F + f] + mixer[(long long)hp * 2 * F + F + f], wbias[hp * F + f], v);
      kconst[hp] = v;
      continue;
    }
    const int col = (int)(idx / G), g = (int)(idx % G);
    float v = 0.f;
    if (col < P * G) {
      const int hp = col / G, r = col % G;
      if (r < 2)
        for (int f = 0; f < F; ++f)
          v = fmaf(mixer[(long long)hp * 2 * F + r * F + f], weight[((long long)hp * F + f) * G + g], v);
    } else {
      const int r = col - P * G, hp = r / (K * F), k = (r / F) % K, f = r % F;
      v = mode == MAGAT_MODE_GAT_ORIGIN ? taps[k] * weight[((long long)hp * F + g) * G + f]
                                        : taps[(((long long)hp * F + f) * K + k) * G + g];
    }
    const float vs = v * 256.f;
    const _Float16 g1 = (_Float16)vs;
    const _Float16 g2 = (_Float16)(vs - (float)g1);
    const int rr = col & 127;
    const long long fo = (long long)(col >> 7) * 32768 + (((rr >> 5) * 8 + (g >> 4)) * 2) * 512 +
                         ((rr & 31) + 32 * ((g & 15) >> 3)) * 8 + (g & 7);
    Fs[fo] = __builtin_bit_cast(unsigned short, g1);
    Fs[fo + 512] = __builtin_bit_cast(unsigned short, g2);
  }
}

static bool gat_rank1_frag(int G, int F, int mode) {
  return G == 128 && F == 128 && (mode == MAGAT_MODE_GAT_MODIFIED || mode == MAGAT_MODE_GAT_ORIGIN);
}

long long* g_gat_dbg = nullptr;   // see magat_gat_set_debug_buffer

bool supported_width(int w) { return w == 16 || w == 32 || w == 64 || w == 128 || w == 256; }

size_t gat_lds_bytes(int N, int G, int F, int nwaves) {
  const int RW = G > F ? G : F;
  const int lda = N | 1;
  return sizeof(float) * (2 * (size_t)N * RW + (size_t)N * lda + 2 * ((N + 3) & ~3)) +
         sizeof(int) * 128 * (size_t)nwaves + sizeof(unsigned) * 4 * (size_t)N + 16;
}

int gat_block_threads(int N) {
  if (N <= 16) return 128;
  if (N <= 32) return 256;
  if (N <= 64) return 512;
  return 1024;
}

// instances per chunk: bounds the hoisted-map intermediate Z (chunk*N*NC floats).  Measured on MI355X
// (c3, B=512): one big launch beats Infinity-Cache-sized chunks (2.02 TB/s vs 1.85 @96 MB vs 1.23 @32 MB),
// so the cap only limits workspace (default 2 GiB).
int gat_chunk_instances(int B, int N, int NC) {
  const double mb = (double)magat_opt(MAGAT_OPT_GAT_CHUNK_MB);
  long long per = (long long)N * NC * 4;
  long long c = (long long)(mb * 1048576.0) / (per > 0 ? per : 1);
  if (c < 8) c = 8;
  if (c > B) c = B;
  return (int)c;
}

template <int G, int F>
int launch_gat(const GatParams& p, int blocks, int threads, size_t lds, hipStream_t st, int tag) {
  constexpr int slot = G == 16 ? MAGAT_LDS_GAT16 : G == 32 ? MAGAT_LDS_GAT32 : G == 64 ? MAGAT_LDS_GAT64
                      : G == 128 ? MAGAT_LDS_GAT128 : MAGAT_LDS_GAT256;
  if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&gat_dense_kernel<G, F>), slot, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  const int pid = magat_prof_begin(tag, st);
  hipLaunchKernelGGL((gat_dense_kernel<G, F>), dim3(blocks), dim3(threads), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

}  // namespace

// Instrumentation only: device buffer of [grid][8] int64 receiving per-workgroup phase timestamps
// (clock64 at entry / after staging / after scores / after each hop / exit, wall_clock64 at exit).
extern "C" int magat_gat_set_debug_buffer(long long* dev_buf) {
  if (!kDebugHooks) return dev_buf ? MAGAT_ERR_UNSUPPORTED : MAGAT_OK;      // release build: no instrumentation
  g_gat_dbg = dev_buf;
  return MAGAT_OK;
}
#ifdef MAGAT_DEBUG_HOOKS
static int g_gat_skip = 0;
extern "C" int magat_gat_set_debug_skip(int mask) { g_gat_skip = mask; return MAGAT_OK; }
#endif

extern "C" size_t magat_gat_packed_floats(int G, int F, int K, int P, int mode) {
  if (G <= 0 || F <= 0 || K <= 0 || P <= 0) return 0;
  const PackLayout L = pack_layout(G, F, K, P, mode);
  if (gat_rank1_frag(G, F, mode))      // + the one-launch kernel's weight stream and the per-head score constants
    return magat_gat_frag_offset(L.NC, G) + (size_t)(P * G + P * K * F) * G + (((size_t)P + 3) & ~(size_t)3);
  if (G == 128 && (L.NC & 127) == 0 && mode == MAGAT_MODE_KEYQUERY)      // + the bf16 fragments of gat_csr_fused.hip (NC * G bf16)
    return magat_gat_csr_fused_offset(L.NC, G) + (size_t)L.NC * G / 2 + 4;
  return magat_gat_f16_block_offset(L.NC, G) + (size_t)L.NC * G + 4;
}

// hoisted dense maps Z = X @ Bt^T + colbias: bf16x6 split-MFMA GEMM when the shape allows, else fp32 MFMA
// row stride of the hoisted-map intermediate Z in the dense path: NC + pad.  NC is a multiple of 128 floats at the
// benchmark shapes (2048 -> 8 KB rows): the tiles a workgroup reads are 512-byte row pieces exactly 8 KB apart, which
// all land in the same HBM channels; a 128-byte skew per row spreads them.
static int gat_zpad() { return 32; }      // row skew of Z (floats)
// float32-MFMA form of the maps (exact fp32 products): the guard's re-run, and the training path
static int gat_maps_gemm_f32(const float* X, const float* packed, float* Z, int M, int G, int NC, int ldz, void* stream,
                             long long ntile_stride, const int32_t* run_if, int tag) {
  magat_conv_gemm_desc d = {};
  d.tag = tag;
  d.in = X; d.wt = packed; d.bias = packed + (size_t)NC * G; d.out = Z;
  d.M = M; d.Cin = G; d.lda = G; d.Hin = d.Win = 1; d.kH = d.kW = 1; d.stride = 1; d.pad = 0;
  d.Hout = d.Wout = 1; d.Cout = NC; d.ldc = ldz;
  if (ntile_stride) { d.ldc = 128; d.out_ntile_stride = ntile_stride; }
  d.run_if = run_if;
  return magat_conv_gemm_f32(&d, stream);
}

__global__ void gat_guard_count_kernel(int* status) {     // see guard_count_kernel (encoder_f32.hip)
  const int f = status[0];
  status[2] = f;
  if (f != 0) status[1] += 1;
  status[0] = 0;
}

// status (device int32[3], may be null): range guard of the f16x3 form - the working flag [0] (zero between forwards) is
// OR-ed by the split GEMM when it had to clamp an X value into its f16 planes, a float32-MFMA GEMM predicated on it re-writes
// Z in the same stream, and a one-thread kernel moves the flag to [2], counts re-runs in [1] and clears [0].  force_f32: skip the split form altogether.
int magat_gat_maps_gemm(const float* X, const float* packed, float* Z, int M, int G, int NC, int ldz, void* stream,
                        long long ntile_stride, int32_t* status, int force_f32) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!force_f32 && magat_opt(MAGAT_OPT_GAT_SPLIT) && NC % 32 == 0 && G % 32 == 0) {
    const int use_f16 = 1;      // (f16x3; the bf16x6 flavour was removed in round 5; the pack's bf16 plane 0 feeds the bf16-storage maps GEMM)
    const bool guard = status && use_f16 && magat_opt(MAGAT_OPT_RANGE_GUARD) != 0;
    magat_conv_gemm_desc d = {};
    d.in = X;
    d.wt = use_f16 ? packed + magat_gat_f16_block_offset(NC, G) : packed + (((size_t)NC * (G + 1) + 3) & ~(size_t)3);
    d.bias = packed + (size_t)NC * G;
    d.out = Z;
    d.M = M; d.Cin = G; d.lda = G; d.Hin = d.Win = 1; d.kH = d.kW = 1; d.stride = 1; d.Hout = d.Wout = 1;
    d.Cout = NC; d.ldc = ldz; d.tag = MAGAT_TAG_GAT_MAPS; d.in_fmt = use_f16 ? 4 : 2;
    if (ntile_stride) { d.ldc = 128; d.out_ntile_stride = ntile_stride; }
    d.range_flag = guard ? status : nullptr;
    // activation scale of the layer input (magat_hip.h "Activation scales"): float word [4] of the status block, written by
    // the caller (0 = none); the direct kernel's float32 loader applies it
    if (status && use_f16 && magat_conv_direct_enabled()) d.in_scale = reinterpret_cast<const float*>(status + 4);
    int rc = magat_conv_gemm_f32(&d, stream);
    if (rc != MAGAT_OK || !guard) return rc;
    const int pid = magat_prof_begin(MAGAT_TAG_RANGE_GUARD, st);
    rc = gat_maps_gemm_f32(X, packed, Z, M, G, NC, ldz, stream, ntile_stride, status, MAGAT_TAG_UNTAGGED);
    if (rc == MAGAT_OK) {
      hipLaunchKernelGGL(gat_guard_count_kernel, dim3(1), dim3(1), 0, st, status);
      if (hipGetLastError() != hipSuccess) rc = MAGAT_ERR_LAUNCH;
    }
    magat_prof_end(pid, st);
    return rc;
  }
  if (ntile_stride && (NC & 127)) return MAGAT_ERR_UNSUPPORTED;
  return gat_maps_gemm_f32(X, packed, Z, M, G, NC, ldz, stream, ntile_stride, nullptr, MAGAT_TAG_GAT_MAPS);
}

extern "C" int magat_gat_pack_weights(const float* weight, const float* weight_bias, const float* mixer,
                                      const float* taps, float* packed, int G, int F, int K, int P, int mode,
                                      void* stream) {
  if ((!weight && mode != MAGAT_MODE_GNN) || !taps || !packed) return MAGAT_ERR_NULL;
  if (mode == MAGAT_MODE_GAT_MODIFIED && (!weight_bias || !mixer)) return MAGAT_ERR_NULL;
  if (mode == MAGAT_MODE_GAT_ORIGIN && !mixer) return MAGAT_ERR_NULL;
  if (G <= 0 || F <= 0 || K <= 0 || P <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (mode < MAGAT_MODE_KEYQUERY || mode > MAGAT_MODE_GNN) return MAGAT_ERR_UNSUPPORTED;
  const PackLayout L = pack_layout(G, F, K, P, mode);
  const long long total = (long long)L.NC * (G + 1);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), weight, weight_bias,
                     mixer, taps, packed, G, F, K, P, mode, L);
  if (G == 128 && F == 128 && K == 2 && mode == MAGAT_MODE_KEYQUERY && (P == 1 || P == 2 || P == 4)) {
    const int rc = magat_gat_csr_fused_pack(packed, packed + magat_gat_csr_fused_offset(L.NC, G), P, static_cast<hipStream_t>(stream));
    if (rc != MAGAT_OK) return rc;
  }
  if (gat_rank1_frag(G, F, mode)) {
    float* frag = packed + magat_gat_frag_offset(L.NC, G);
    hipLaunchKernelGGL(pack_frag_rank1_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), weight, weight_bias,
                       mixer, taps, reinterpret_cast<unsigned short*>(frag), frag + (size_t)(P * G + P * K * F) * G, K, P, mode);
  }
  return magat_check_launch();
}

// 1 when the LDS-resident dense-GSO kernel covers this shape, 0 when the caller must use the CSR entry point
extern "C" int magat_gat_dense_supported(int N, int G, int F) {
  if (N <= 0 || N > 128 || G != F || !supported_width(G)) return 0;
  return gat_lds_bytes(N, G, F, gat_block_threads(N) / 64) <= 160 * 1024 ? 1 : 0;
}

// (The optional GSO plan of rounds 2-5 - edge masks and an edge-count-balanced instance walk made at addGSO time on a side
//  stream, magat_gat_gso_plan - was removed in round 6: opt-in, measured 0.5-1.8 % SLOWER per step at c3, never the default.
//  magat_gat_forward_planned_f32 keeps its signature; its `plan` argument must be NULL.)
constexpr int GAT_PLAN_WALKERS = 256;     // persistent graph-kernel workgroups (one per CU)

constexpr size_t GAT_STATUS_BYTES = 256;   // status block (range guard of the maps GEMM) at the head of the workspace

extern "C" int magat_gat_read_status(const void* workspace, int32_t status_host[2], void* stream) {
  if (!workspace || !status_host) return MAGAT_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t w[3];
  if (hipMemcpyAsync(w, workspace, sizeof(w), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return MAGAT_ERR_LAUNCH;
  status_host[0] = w[2];
  status_host[1] = w[1];
  return MAGAT_OK;
}

extern "C" size_t magat_gat_workspace_bytes(int B, int N, int G, int F, int K, int P, int mode, int concat) {
  if (B <= 0 || N <= 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return 0;
  const PackLayout L = pack_layout(G, F, K, P, mode);
  const int ldz = L.NC + gat_zpad();
  const int chunk = gat_chunk_instances(B, N, ldz);
  size_t bytes = GAT_STATUS_BYTES + magat_align_up((size_t)chunk * N * ldz * sizeof(float), 256);
  if (!concat) bytes += magat_align_up((size_t)B * N * P * F * sizeof(float), 256);
  return bytes;
}

static bool gat_one_launch(int N, int G, int F, int K, int mode, int concat) {
  (void)concat;
  return magat_opt(MAGAT_OPT_GAT_MFMA) && magat_opt(MAGAT_OPT_GAT_SPLIT) &&
         (magat_gat_mfma_supported(N, G, F, K, mode) || magat_gat_small_supported(N, G, F, K, mode) ||
          magat_gat_mid_supported(N, G, F, K, mode));
}

extern "C" int magat_gat_one_launch_supported(int N, int G, int F, int K, int mode, int concat) {
  if (N <= 0 || G != F || !supported_width(G) || N > 128) return 0;
  return gat_one_launch(N, G, F, K, mode, concat) ? 1 : 0;
}

// tail / tail_done: magat_gat_forward_tail_f32 (below)
static int gat_forward_impl(const float* X, const void* S, int s_is_f64, const float* packed, const float* bias, float* Y, int ldy,
                            float* A_opt, void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                            int mode, int concat, const magat_conv_gemm_desc* tail, int* tail_done, void* stream) {
  if (!X || !S || !packed || !Y) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || G <= 0 || F <= 0 || K <= 0 || P <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (mode < MAGAT_MODE_KEYQUERY || mode > MAGAT_MODE_GAT_ORIGIN) return MAGAT_ERR_UNSUPPORTED;
  if (G != F || !supported_width(G) || N > 128) return MAGAT_ERR_UNSUPPORTED;
  const int width = concat ? P * F : F;
  if (ldy < width || (ldy & 3)) return MAGAT_ERR_BAD_SHAPE;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) ||
      workspace_bytes < magat_gat_workspace_bytes(B, N, G, F, K, P, mode, concat))
    return MAGAT_ERR_WORKSPACE;
  const int threads = gat_block_threads(N);
  const size_t lds = gat_lds_bytes(N, G, F, threads / 64);
  // beyond gat_dense_kernel's tiles (128 features: N > 105) only the one-launch form of gat_mid.hip exists, with gat_slim_kernel
  // as the range guard's float32 re-run; no attention tensor there (the CSR kernels serve such requests)
  const bool slim = lds > 160 * 1024;
  if (slim && !(G == 128 && F == 128 && mode == MAGAT_MODE_KEYQUERY && !A_opt && gat_one_launch(N, G, F, K, mode, concat) &&
                (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0))
    return MAGAT_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);

  const PackLayout L = pack_layout(G, F, K, P, mode);
  const int ldz = L.NC + gat_zpad();
  const int chunk = gat_chunk_instances(B, N, ldz);
  int32_t* status = static_cast<int32_t*>(workspace);
  float* Z = reinterpret_cast<float*>(static_cast<char*>(workspace) + GAT_STATUS_BYTES);
  float* Ytmp = reinterpret_cast<float*>(static_cast<char*>(workspace) + GAT_STATUS_BYTES +
                                         magat_align_up((size_t)chunk * N * ldz * sizeof(float), 256));
  GatParams p;
  p.order = nullptr;
  p.rmask_pre = nullptr;
  p.run_if = nullptr;
  p.book = nullptr;
  p.dbg = g_gat_dbg;
  p.skip = 0;
#ifdef MAGAT_DEBUG_HOOKS
  p.skip = g_gat_skip;
#endif
  p.X = X; p.S = S; p.Z = Z; p.bias = bias; p.A_opt = A_opt;
  p.Y = concat ? Y : Ytmp;
  p.N = N; p.K = K; p.P = P; p.mode = mode; p.concat = concat; p.s_is_f64 = s_is_f64;
  p.ldx = G; p.ldy = concat ? ldy : P * F; p.NC = ldz; p.lda_a = N | 1;
  p.qoff = L.qoff; p.uoff = L.uoff; p.c1off = L.c1off; p.c2off = L.c2off;

  // One launch of matrix-core products (gat_mfma.hip) when the shape allows; with the range guard on, the two-launch float32
  // form below follows in the same stream, every launch of it predicated on the flag the fused kernel raises.
  bool rerun_only = false;
  // (the one-launch kernel stores Y in 16-byte pieces: a column block at an odd offset of a wider buffer takes the two-launch form)
  // ... and the small-graph kernel loads X rows as 16-byte pieces)
  if (!A_opt && gat_one_launch(N, G, F, K, mode, concat) && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
      (G == 128 || (reinterpret_cast<uintptr_t>(X) & 15) == 0)) {
    const bool guard = magat_opt(MAGAT_OPT_RANGE_GUARD) != 0;
    const unsigned* masks = nullptr;
    const float* frag = packed + magat_gat_frag_offset(L.NC, G);
    // (128 features: gat_mfma.hip; 32 / 64 features on graphs of at most 32 agents: gat_small.hip, a wave per instance)
    const int rc = G == 128 && magat_gat_mfma_supported(N, G, F, K, mode)
        ? magat_gat_mfma_forward(X, G, S, s_is_f64, masks, frag, bias, Y, ldy, B, N, K, P, concat,
                                 guard ? status : nullptr, st, reinterpret_cast<const float*>(status + 4), mode,
                                 mode == MAGAT_MODE_KEYQUERY ? nullptr : frag + (size_t)(P * G + P * K * F) * G,
                                 concat ? nullptr : Ytmp, P * F)
        : G != 128 && N <= 32
        ? magat_gat_small_forward(X, G, S, s_is_f64, masks, packed + magat_gat_f16_block_offset(L.NC, G), L.NC, bias, Y, ldy, B,
                                  N, G, K, P, concat, guard ? status : nullptr, st, reinterpret_cast<const float*>(status + 4))
        // (32 / 64 features on 33 .. 128 agents, 128 features on 103 .. 128: gat_mid.hip, a workgroup of ceil(N / 32) waves per
        //  instance - round 6)
        : magat_gat_mid_forward(X, G, S, s_is_f64, packed + magat_gat_f16_block_offset(L.NC, G), L.NC, bias, Y, ldy, B, N, G, K, P,
                                concat, guard ? status : nullptr, st, reinterpret_cast<const float*>(status + 4),
                                concat ? nullptr : Ytmp, P * F, G == 128 ? frag : nullptr);
    if (rc != MAGAT_OK || !guard) return rc;
    rerun_only = true;
    p.run_if = status;
    if (mode == MAGAT_MODE_KEYQUERY && (G == 32 || G == 64 || G == 128) && (long long)B * P <= GAT_RERUN_SMALL_UNITS &&
        (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
      // few instances: the whole float32 form as ONE predicated launch (final rows straight into Y, the guard's bookkeeping too)
      p.B = B; p.b0 = 0; p.Ymean = Y; p.ldym = ldy; p.book = reinterpret_cast<int*>(status);
      size_t slds = sizeof(float) * ((size_t)N * (G + 1) + (size_t)N * (N | 1));
      // the layer that reads this one's rows (the action head: five outputs) in the same launch, when the caller handed it over
      MagatSkinnyParams sp{};
      const bool with_tail = tail && magat_skinny_params(tail, &sp) == MAGAT_OK && sp.Cout == 5 && sp.M == B * N &&
                             (size_t)(sp.Cin + sp.C2) * 5 * sizeof(float) <= 64 * 1024;
      if (with_tail && (size_t)(sp.Cin + sp.C2) * 5 * sizeof(float) > slds) slds = (size_t)(sp.Cin + sp.C2) * 5 * sizeof(float);
      const int gi = G == 32 ? 0 : G == 64 ? 1 : 2;
      const void* fns[2][3] = {{reinterpret_cast<const void*>(&gat_rerun_small_kernel<32, 0>), reinterpret_cast<const void*>(&gat_rerun_small_kernel<64, 0>),
                                reinterpret_cast<const void*>(&gat_rerun_small_kernel<128, 0>)},
                               {reinterpret_cast<const void*>(&gat_rerun_small_kernel<32, 5>), reinterpret_cast<const void*>(&gat_rerun_small_kernel<64, 5>),
                                reinterpret_cast<const void*>(&gat_rerun_small_kernel<128, 5>)}};
      if (magat_ensure_dyn_lds(fns[with_tail ? 1 : 0][gi], MAGAT_LDS_GAT_RERUN_S + gi + (with_tail ? 3 : 0), slds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
      const float* cbias = packed + (size_t)L.NC * G;
      const int pid = magat_prof_begin(MAGAT_TAG_RANGE_GUARD, st);
      if (with_tail) {
        const int tg = (B * N + 15) / 16, grid = tg > B ? tg : B;      // (B x heads <= 64, N <= 128: at most 512 workgroups)
        if (G == 32) hipLaunchKernelGGL((gat_rerun_small_kernel<32, 5>), dim3(grid), dim3(256), slds, st, p, packed, cbias, sp);
        else if (G == 64) hipLaunchKernelGGL((gat_rerun_small_kernel<64, 5>), dim3(grid), dim3(256), slds, st, p, packed, cbias, sp);
        else hipLaunchKernelGGL((gat_rerun_small_kernel<128, 5>), dim3(grid), dim3(256), slds, st, p, packed, cbias, sp);
        magat_form_note(MAGAT_FORM_ACTIONS_TAIL);
      } else {
        if (G == 32) hipLaunchKernelGGL((gat_rerun_small_kernel<32, 0>), dim3(B), dim3(256), slds, st, p, packed, cbias, sp);
        else if (G == 64) hipLaunchKernelGGL((gat_rerun_small_kernel<64, 0>), dim3(B), dim3(256), slds, st, p, packed, cbias, sp);
        else hipLaunchKernelGGL((gat_rerun_small_kernel<128, 0>), dim3(B), dim3(256), slds, st, p, packed, cbias, sp);
      }
      magat_prof_end(pid, st);
      if (with_tail && tail_done) *tail_done = 1;
      return magat_check_launch();
    }
  }
  const int hpb_env = 0;      // (heads per workgroup: automatic)
  auto hpb_for = [&](int cb) {
    int h = 1;
    if (G >= 64 && P > 1 && lds > 80 * 1024 && cb >= 256) h = P;
    if (hpb_env > 0 && G >= 64 && P % hpb_env == 0) h = hpb_env;
    return h;
  };
  bool all_fused = !concat && G >= 64 && !slim;
  for (int b0 = 0; b0 < B && all_fused; b0 += chunk) all_fused = hpb_for((B - b0) < chunk ? (B - b0) : chunk) == P;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int cb = (B - b0) < chunk ? (B - b0) : chunk;
    // Z in 128-column tiles ([tile][cb * N][128]: an instance's Q_p / U_pk tile is ONE contiguous N x 128 run for the
    // LDS-direct loads, and every workgroup of the maps GEMM writes one contiguous region) when the dense kernel with
    // 128-wide features consumes it and the f16x3 direct GEMM produces it.
    const bool ztiles = G == 128 && F == 128 && L.NC % 128 == 0 && magat_conv_direct_enabled() &&
                        magat_opt(MAGAT_OPT_GAT_SPLIT);
    p.zts = ztiles ? (long long)cb * N * 128 : 0;
    // (one chunk is the rule; with several, a clamp in an earlier chunk leaves the flag set only until the next chunk's
    // own GEMM clears it - status[1] still counts every re-run)
    int rc = rerun_only ? gat_maps_gemm_f32(X + (size_t)b0 * N * G, packed, Z, cb * N, G, L.NC, ldz, stream, p.zts, status,
                                            MAGAT_TAG_RANGE_GUARD)
                        : magat_gat_maps_gemm(X + (size_t)b0 * N * G, packed, Z, cb * N, G, L.NC, ldz, stream, p.zts, status);
    if (rc != MAGAT_OK) return rc;
    p.B = cb; p.b0 = b0;
    // heads per workgroup: when the LDS tiles allow only one workgroup per CU there is nothing to overlap a
    // workgroup's loads with, so one workgroup walks all P heads of its instance and prefetches the next head's
    // Q tile during the current head's compute; needs enough instances to fill the chip.
    const int hpb = hpb_for(cb);
    p.hpb = hpb;
    // mean merge fused into the graph kernel when one workgroup sees all heads of an instance (decided for the whole
    // call: every chunk must qualify, otherwise the separate kernel merges everything from Ytmp)
    p.Ymean = all_fused ? Y : nullptr;
    p.ldym = ldy;
    // with one workgroup per CU (hpb == P case) the grid is capped at one workgroup per CU and every workgroup walks
    // several instances, prefetching across the instance boundary as well
    int inst_slots = (cb + MAGAT_NUM_XCD - 1) / MAGAT_NUM_XCD * MAGAT_NUM_XCD;
    if (hpb == P && hpb > 1 && inst_slots > GAT_PLAN_WALKERS) inst_slots = GAT_PLAN_WALKERS;
    // the range guard's predicated re-run: a launch that returns at once still pays for every workgroup it dispatches - with one
    // workgroup per (instance, head) that was 4096 dispatches, 76 us per forward at BASELINE config 2 (1024 instances of 20
    // agents; 88 us of a 1.10 ms step went to the guard).  The workgroups walk the instances instead (istride below)
    if (rerun_only && inst_slots > 128) inst_slots = 128;
    // (capping the ordinary launches the same way was measured at the published shape - 4096 workgroups of the narrow graph
    //  kernel, 21.7 us: 23 / 32 / 55 us with 512 / 256 / 128 slots - they are not dispatch-bound)
    const int blocks = inst_slots * (P / hpb);
    p.order = nullptr;
    p.rmask_pre = nullptr;
    const int gtag = rerun_only ? MAGAT_TAG_RANGE_GUARD : MAGAT_TAG_GAT_GRAPH;
    // the re-run's last launch does the guard's bookkeeping: the graph kernel of the last chunk, or the head-mean kernel
    p.book = (rerun_only && b0 + chunk >= B && (concat || all_fused)) ? reinterpret_cast<int*>(status) : nullptr;
    if (slim) {
      const size_t slds = sizeof(float) * ((size_t)N * 129 + (size_t)N * (N | 1));
      if (magat_ensure_dyn_lds(reinterpret_cast<const void*>(&gat_slim_kernel), MAGAT_LDS_GAT_SLIM, slds) != MAGAT_OK)
        return MAGAT_ERR_LAUNCH;
      const int sblocks = cb * P < 128 ? cb * P : 128;      // (a no-op launch pays for every workgroup it dispatches)
      const int pid = magat_prof_begin(gtag, st);
      hipLaunchKernelGGL(gat_slim_kernel, dim3(sblocks), dim3(256), slds, st, p);
      magat_prof_end(pid, st);
      rc = magat_check_launch();
    } else
    switch (G) {
      case 16: rc = launch_gat<16, 16>(p, blocks, threads, lds, st, gtag); break;
      case 32: rc = launch_gat<32, 32>(p, blocks, threads, lds, st, gtag); break;
      case 64: rc = launch_gat<64, 64>(p, blocks, threads, lds, st, gtag); break;
      case 128: rc = launch_gat<128, 128>(p, blocks, threads, lds, st, gtag); break;
      case 256: rc = launch_gat<256, 256>(p, blocks, threads, lds, st, gtag); break;
      default: rc = MAGAT_ERR_UNSUPPORTED;
    }
    if (rc != MAGAT_OK) return rc;
  }
  if (!concat && !all_fused) {
    const long long M = (long long)B * N;
    long long blocks = (M * (F / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const int pid = magat_prof_begin(rerun_only ? MAGAT_TAG_RANGE_GUARD : MAGAT_TAG_HEAD_MEAN, st);
    hipLaunchKernelGGL(head_mean_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Ytmp, Y, M, P, F, ldy, p.run_if,
                       rerun_only ? reinterpret_cast<int*>(status) : nullptr);
    magat_prof_end(pid, st);
    if (magat_check_launch() != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  }
  // (flag -> status[2], re-run count, flag cleared: done by the last workgroup of the re-run's last launch, magat_guard_book)
  return MAGAT_OK;
}

extern "C" int magat_gat_forward_planned_f32(const float* X, const void* S, int s_is_f64, const float* packed,
                                             const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                             size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                             int mode, int concat, const void* plan, void* stream) {
  if (plan) return MAGAT_ERR_UNSUPPORTED;      // (the GSO plan was removed in round 6)
  return gat_forward_impl(X, S, s_is_f64, packed, bias, Y, ldy, A_opt, workspace, workspace_bytes, B, N, G, F, K, P, mode, concat,
                          nullptr, nullptr, stream);
}

// The layer, and - when the launches it takes leave room for it - the skinny float32 layer that reads its rows (`tail`: the action
// head of the planner, magat_conv_gemm_desc of a 1 x 1 product with five outputs over M = B N rows) inside the layer's last launch:
// *tail_done = 1 then, 0 when the caller has to run magat_conv_gemm_f32(tail) itself (always a valid outcome).  Today that is the
// predicated range-guard re-run of few instances (instances x heads <= 64, KeyQuery): the closed-loop step of one planning instance
// loses a launch.  The tail's rows are computed by skinny_rows.h's code either way: the same bits.
extern "C" int magat_gat_forward_tail_f32(const float* X, const void* S, int s_is_f64, const float* packed, const float* bias,
                                          float* Y, int ldy, void* workspace, size_t workspace_bytes, int B, int N, int G, int F,
                                          int K, int P, int mode, int concat, const magat_conv_gemm_desc* tail, int* tail_done,
                                          void* stream) {
  if (tail_done) *tail_done = 0;
  return gat_forward_impl(X, S, s_is_f64, packed, bias, Y, ldy, nullptr, workspace, workspace_bytes, B, N, G, F, K, P, mode, concat,
                          tail, tail_done, stream);
}

extern "C" int magat_gat_forward_packed_f32(const float* X, const void* S, int s_is_f64, const float* packed,
                                            const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                            size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                            int mode, int concat, void* stream) {
  return magat_gat_forward_planned_f32(X, S, s_is_f64, packed, bias, Y, ldy, A_opt, workspace, workspace_bytes, B, N, G,
                                       F, K, P, mode, concat, nullptr, stream);
}

extern "C" int magat_gat_forward_dense_f32(const float* X, const void* S, int s_is_f64, const float* weight,
                                           const float* weight_bias, const float* mixer, const float* taps,
                                           const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                           size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                           int mode, int concat, void* stream) {
  const size_t base = magat_gat_workspace_bytes(B, N, G, F, K, P, mode, concat);
  const size_t pf = magat_gat_packed_floats(G, F, K, P, mode);
  if (base == 0 || pf == 0) return MAGAT_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < base + pf * sizeof(float)) return MAGAT_ERR_WORKSPACE;
  float* packed = reinterpret_cast<float*>(static_cast<char*>(workspace) + base);
  int rc = magat_gat_pack_weights(weight, weight_bias, mixer, taps, packed, G, F, K, P, mode, stream);
  if (rc != MAGAT_OK) return rc;
  return magat_gat_forward_packed_f32(X, S, s_is_f64, packed, bias, Y, ldy, A_opt, workspace, base, B, N, G, F, K,
                                      P, mode, concat, stream);
}

// addGSO's in-place scrub (decentralplanner_GAT_bottleneck.py:272-277)
template <typename T>
__global__ void gso_prepare_kernel(T* S, size_t count, int scrub_nan, int gso_mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T v = S[i];
    if (scrub_nan && v != v) v = (T)0;
    if (gso_mode == 1 && v > (T)0) v = (T)1;
    if (gso_mode == 2) v = (T)1;
    S[i] = v;
  }
}

extern "C" int magat_gso_prepare(void* S, int s_is_f64, size_t count, int scrub_nan, int gso_mode, void* stream) {
  if (!S) return MAGAT_ERR_NULL;
  if (count == 0) return MAGAT_OK;
  size_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (s_is_f64)
    hipLaunchKernelGGL(gso_prepare_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                       static_cast<double*>(S), count, scrub_nan, gso_mode);
  else
    hipLaunchKernelGGL(gso_prepare_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                       static_cast<float*>(S), count, scrub_nan, gso_mode);
  return magat_check_launch();
}

extern "C" int magat_abi_version(void) { return 9; }

extern "C" const char* magat_error_string(int code) {
  switch (code) {
    case MAGAT_OK: return "ok";
    case MAGAT_ERR_BAD_SHAPE: return "bad shape / stride / alignment";
    case MAGAT_ERR_UNSUPPORTED: return "unsupported width, mode or graph size for the gfx950 kernels";
    case MAGAT_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
    case MAGAT_ERR_LAUNCH: return "HIP kernel launch failed";
    case MAGAT_ERR_NULL: return "required pointer is NULL";
    default: return "unknown magat error";
  }
}
