// GraphFilterBatchAttentional.forward (KeyQuery attention, 128 features in and out) as ONE launch in which every step is a
// matrix-core product (reference utils/graphUtils/graphML.py:4636-4671, 1724-1827, 1180-1286, 713-823; algebra in
// gat_f32.hip).  Nothing but X, the GSO and Y crosses HBM: the per-agent maps Z = X [W_p | H_pk]^T that the two-launch
// form writes and reads back (0.84 GB per step at N = 100, B = 512) stay in registers and LDS.
//
// Per planning instance b and head p, with X the [N][128] feature rows (all products are f16x3 split products - two f16
// planes per operand, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation; weights carry a fixed 2^8):
//   G1  Q[j][g]   = sum_f X[j][f] W_p[g][f]                      -> planes Q [j][g]        (graphML.py:1767-1769)
//   G2  E^T[j][i] = sum_g Q[j][g] X[i][g]   (e[i][j] = x_i . q_j), masked row softmax over j IN the accumulator layout
//                   (a lane owns column i: its row of the softmax is 16 MT registers + the partner lane)
//                                                               -> planes A [j][i] * 2^8   (graphML.py:1771-1776)
//   G3  U_k[i][c] = sum_f X[i][f] H_pk[c][f], k = 0..K-1: K accumulator sets, 2^8 U_k    (graphML.py:801-817)
//       planes U^T [c][i] = U_{K-1}
//   G4  for k = K-2 .. 0:  acc_k[j][c] += sum_i A[i][j] U^T[c][i]   (the hop is a dense product: at N = 100 and 5 %
//       density 21 MFMAs per 32 x 32 tile against ~10 k cycles of gathers), planes U^T = acc_k 2^-8 while k > 0
//   Y_p = relu(acc_0 2^-8 + bias)  (concat) or summed over the heads, / P, relu (mean)     (graphML.py:4660-4667)
// Four waves, one per SIMD with the whole register file (the four-wave form of block_fused.hip): wave w owns output
// columns 32 w .. 32 w + 31 of every product - the i tile of E^T (its softmax rows need no other wave), the c tile of U_k
// and of the hops (the U^T rows a wave writes are the rows it reads back).  Weights travel global -> registers as
// fragment-major f16 planes (packed by magat_gat_pack_weights), one 1 KB fragment per (32-row tile, 16-wide k step,
// plane): 2 KB per 3 MT MFMAs per wave.
// LDS (N = 100: 162,240 of 163,840 bytes): X planes [N][2][256 B] (16-byte chunks XOR-swizzled by row), A planes [N][SA],
// U^T planes [128][SA] with SA = 2 KI + 16 (KI = 16 KSI >= N columns; rows 60 or 68 banks apart: conflict-free b128 reads);
// the Q planes live in the U^T region (dead before U^T is written).  That is what bounds N: N <= 101.
// Values outside the f16 range are clamped and reported in range_flag (magat_hip.h "range guard"): the caller re-runs
// the two-launch float32 form when it is set.
#include "magat_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct GatMfmaParams {
  const float* X;             // [B*N][ldx]
  const void* S;              // [B][N][N] f32 | f64 (read only when rmask_pre is null)
  const unsigned* rmask_pre;  // [B][N][4] edge masks of a GSO plan, or null
  const char* wfrag;          // fragment-major weight planes: 64 KB blocks, [P] W_p then [P*K] H_pk
  const float* bias;          // [128] or null
  float* Y;                   // [B*N][ldy]; concat: head p at column 128 p
  int B, N, P, ldx, ldy, concat, s_is_f64;
  int* range_flag;
  long long* dbg;             // MAGAT_DEBUG_HOOKS builds: [grid][4 waves][16] cycle stamps of the last head walked
};

#ifdef MAGAT_DEBUG_HOOKS
#define GM_STAMP(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
long long* g_gat_mfma_dbg = nullptr;
#else
#define GM_STAMP(i) do { } while (0)
#endif

// timing experiments (tools/whatif_gat_mfma.sh; wrong results): a store replaced by a register sink
#define GM_SINK(x) asm volatile("" ::"v"(x))
#ifdef GM_WHATIF_NOQW
#define GM_QW(ptr, val) GM_SINK(val)
#else
#define GM_QW(ptr, val) *reinterpret_cast<unsigned short*>(ptr) = (val)
#endif
#ifdef GM_WHATIF_NOAW
#define GM_AW(ptr, val) GM_SINK(val)
#else
#define GM_AW(ptr, val) *reinterpret_cast<unsigned short*>(ptr) = (val)
#endif
#ifdef GM_WHATIF_NOYST
#define GM_YST(ptr, val) GM_SINK(val)
#else
#define GM_YST(ptr, val) *reinterpret_cast<float*>(ptr) = (val)
#endif

// LDS hand-over barrier without the vmcnt(0) of __syncthreads(): weight fragments and Y stores stay in flight
#define GM_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)

__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one 16-wide k step of NB x MT tiles: the three split products, tile after tile (no two consecutive MFMAs on one accumulator
// when there is more than one tile)
template <int MT, int NB>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[NB][MT], const uint4 (&a)[MT][2], const uint4 (&b)[NB][2]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pa = q == 2 ? 1 : 0, pb = q == 1 ? 1 : 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nb][mt] = mfma16(a[mt][pa], b[nb][pb], acc[nb][mt]);
  }
}

template <int MT>
__device__ __forceinline__ void mma_step_row(f32x16 (&acc)[MT], const uint4 (&a)[MT][2], const uint4 (&b)[2]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pa = q == 2 ? 1 : 0, pb = q == 1 ? 1 : 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(a[mt][pa], b[pb], acc[mt]);
  }
}

// value pair -> its two f16 planes: hi = rne(v) (v_cvt_pk_f16_f32), lo = rne(v - hi) with the residual formed by one
// mixed-precision fma per value (fma(hi, -1, v): exact).  No range clamp: a value beyond the f16 range turns into inf /
// nan planes, and the caller's running maximum `vmax` of |v| raises the range flag, which makes the float32 form
// re-write every output of the launch.
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2) {
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  float rx, ry;      // v - (float)h as fma(h, -1, v), the f16 operand read straight from its half of the packed register
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split2v(float x, float y, unsigned& p1, unsigned& p2, float& vmax) {
  vmax = fmaxf(fmaxf(vmax, fabsf(x)), fabsf(y));
  split_pair(x, y, p1, p2);
}

// (pointer + compile-time constant: the constant lands in the offset field of the ds instruction)
__device__ __forceinline__ uint4 lds128(const char* ptr, int coff) { return *reinterpret_cast<const uint4*>(ptr + coff); }

// MT: 32-row tiles covering the agents (N <= 32 MT); KSI: 16-wide k steps covering them as a contraction index; KT: taps
template <int MT, int KSI, int KT, bool CONCAT>
__global__ __launch_bounds__(256, 1) void gat_mfma_kernel(const GatMfmaParams p) {
  extern __shared__ __align__(16) char lds[];
  constexpr int KI = 16 * KSI, SA = 2 * KI + 16;
  constexpr float kInvScale = 1.f / 256.f;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, h_ = lane >> 5;
  const int N = p.N;
  // LDS map.  X / Q planes: row r at 512 r, hi plane then lo plane (256 B each), 16-byte chunk c of a plane at
  // (c ^ (r & 15)) << 4.  A / U^T planes: rows SA bytes apart, lo plane behind the hi plane.
  const unsigned AO = 512u * N, AP = (unsigned)N * SA;      // A planes [N][SA]
  const unsigned UO = AO + 2 * AP;                          // U^T planes [128][SA]; the Q planes share the region
  constexpr unsigned UP = 128 * SA;
  const unsigned MO = UO + 2 * UP;                          // edge masks [N][4] (staged here when there is no plan)

  // per-lane fragment bases.  Row tiles of the agents (A operand of G1 / G2 / G3 and of the hops; rows past N re-read row
  // N - 1: finite values whose products land in rows nobody stores or in columns the zero entries of A annihilate)
  unsigned rsw_[MT], xsw_[MT], rpa_[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(32 * mt + fr, N - 1);
    rsw_[mt] = row * 512;
    xsw_[mt] = (h_ ^ (row & 15)) << 4;
    rpa_[mt] = AO + row * SA + h_ * 16;
  }
  // fresh copies per phase (laundered: fragment addresses are formed next to their reads, not hoisted per (tile, k step))
#define GM_FRESH(dst, src) unsigned dst[MT]; _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) { dst[mt_] = src[mt_]; asm volatile("" : "+v"(dst[mt_])); }
  const int rowb = min(32 * w + fr, N - 1);        // this wave's i tile as B operand (G2)
  const unsigned rswb_ = rowb * 512, xswb_ = (h_ ^ (rowb & 15)) << 4;
  const int cw_ = 32 * w + fr;                     // this lane's output column (g in G1, i in G2, c in G3 / hops)
  const char* wl = p.wfrag + (size_t)w * 16384 + lane * 16;   // this wave's 32-row tile of a weight block
  const float biasv = p.bias ? p.bias[cw_] : 0.f;
  const bool g2_active = 32 * w < KI;              // waves whose i tile holds columns of A
  float vmax = 0.f;                                // running maximum of |values written to f16 planes|

  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    // ---- instance prologue: X rows -> f16 planes; edge masks
    {
      const float* Xb = p.X + (long long)b * N * p.ldx;
      for (int idx = t; idx < N * 16; idx += 256) {
        const int row = idx >> 4, ch = idx & 15;
        const float* src = Xb + (long long)row * p.ldx + 8 * ch;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        uint4 hi, lo;
        split2v(v0[0], v0[1], hi.x, lo.x, vmax);
        split2v(v0[2], v0[3], hi.y, lo.y, vmax);
        split2v(v1[0], v1[1], hi.z, lo.z, vmax);
        split2v(v1[2], v1[3], hi.w, lo.w, vmax);
        char* dst = lds + (row * 512 + ((ch ^ (row & 15)) << 4));
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 256) = lo;
      }
      if (!p.rmask_pre) {        // GSO rows -> 128-bit edge masks (one wave per row, ballot; |S| > 1e-9 as in gat_f32.hip)
        const long long sbase = (long long)b * N * N;
        for (int i = w; i < N; i += 4) {
          bool f0, f1;
          const int j0 = lane < N ? lane : N - 1, j1 = lane + 64 < N ? lane + 64 : N - 1;
          if (p.s_is_f64) {
            const double* Sp = static_cast<const double*>(p.S) + sbase + (long long)i * N;
            f0 = fabs(Sp[j0]) > 1e-9;
            f1 = fabs(Sp[j1]) > 1e-9;
          } else {
            const float* Sp = static_cast<const float*>(p.S) + sbase + (long long)i * N;
            f0 = fabsf(Sp[j0]) > 1e-9f;
            f1 = fabsf(Sp[j1]) > 1e-9f;
          }
          const unsigned long long k0 = __ballot(f0 && lane < N), k1 = __ballot(f1 && lane + 64 < N);
          if (lane == 0) {
            unsigned* m = reinterpret_cast<unsigned*>(lds + MO) + 4 * i;
            m[0] = (unsigned)k0; m[1] = (unsigned)(k0 >> 32); m[2] = (unsigned)k1; m[3] = (unsigned)(k1 >> 32);
          }
        }
      }
    }
    GM_SYNC();
    // edge mask of this lane's row i = cw (bits j), shifted so that bit (8 (r / 4) + r % 4) is row 32 mt + ... of the tile
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    if (cw_ < N) {
      const uint4 m = p.rmask_pre ? *reinterpret_cast<const uint4*>(p.rmask_pre + ((long long)b * N + cw_) * 4)
                                  : *reinterpret_cast<const uint4*>(lds + MO + 16 * cw_);
      mk[0] = m.x >> (4 * h_); mk[1] = m.y >> (4 * h_); mk[2] = m.z >> (4 * h_); mk[3] = m.w >> (4 * h_);
    }
    f32x16 ysum[CONCAT ? 1 : MT];      // mean merge: sum over the heads of (Y_p + bias), carried in registers
    if constexpr (!CONCAT) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ysum[mt][r] = 0.f;
    }

#pragma unroll 1
    for (int hd = 0; hd < p.P; ++hd) {
      GM_STAMP(0);
      // lane-derived values are laundered per head: the per-element plane addresses below are recomputed where they are
      // used (a few VALU ops) instead of being hoisted out of the head loop into hundreds of long-lived registers
      int cw = cw_, h = h_;
      asm volatile("" : "+v"(cw), "+v"(h));
      // W_p fragments of this wave's g tile: all eight k steps, in flight across the barrier
      uint4 wq[8][2];
      {
        const char* s = wl + (size_t)hd * 65536;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          wq[ks][0] = *reinterpret_cast<const uint4*>(s + ks * 2048);
          wq[ks][1] = *reinterpret_cast<const uint4*>(s + ks * 2048 + 1024);
        }
      }
      if (hd > 0) GM_SYNC();      // the previous head's reads of the U^T / A planes are done
      GM_STAMP(1);
      // ---- G1: Q[j][g]
      {
        GM_FRESH(rsw, rsw_) GM_FRESH(xsw, xsw_)
        f32x16 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.f;
        uint4 a[2][MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const char* ap = lds + (rsw[mt] + xsw[mt]);
          a[0][mt][0] = lds128(ap, 0);
          a[0][mt][1] = lds128(ap, 256);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const char* ap = lds + (rsw[mt] + (xsw[mt] ^ ((ks + 1) << 5)));
              a[(ks + 1) & 1][mt][0] = lds128(ap, 0);
              a[(ks + 1) & 1][mt][1] = lds128(ap, 256);
            }
          }
          const uint4 bb[1][2] = {{wq[ks][0], wq[ks][1]}};
          mma_step<MT, 1>(acc, a[ks & 1], bb);
          __builtin_amdgcn_sched_barrier(0);
        }
        GM_STAMP(11);
        // Q planes [j][g] (chunks swizzled by row): a lane holds column g = cw, rows j = 32 mt + 8 q + 4 h + e.  The swizzle
        // term (j & 15) = (8 (q & 1) + e) | 4 h takes 8 values per lane: 8 base addresses, everything else is an immediate.
        {
          const unsigned gx = ((cw >> 3) ^ (4 * h)) << 4;
          const unsigned qb = UO + 4 * h * 512 + (cw & 7) * 2;
          char* qa[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            unsigned qo = qb + (gx ^ (unsigned)(((c & 4) * 2 + (c & 3)) << 4));
            asm volatile("" : "+v"(qo));      // (kept as 8 registers: not re-formed from its parts at every write)
            qa[c] = lds + qo;
          }
          const f32x2 sc = {kInvScale, kInvScale};
          // all the arithmetic first, as one straight-line block (rows past N: harmless values nobody stores) ...
          unsigned hq[MT][4][2], lq[MT][4][2];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2 v01 = f32x2{acc[0][mt][4 * q], acc[0][mt][4 * q + 1]} * sc;
              const f32x2 v23 = f32x2{acc[0][mt][4 * q + 2], acc[0][mt][4 * q + 3]} * sc;
              split2v(v01[0], v01[1], hq[mt][q][0], lq[mt][q][0], vmax);
              split2v(v23[0], v23[1], hq[mt][q][1], lq[mt][q][1], vmax);
            }
          // ... then the stores, row groups of 8 (4 registers x the two 4-row halves)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int jg = 32 * mt + 8 * q;
              if (jg >= N) break;                                   // (wave-uniform)
              const unsigned short hv[4] = {(unsigned short)hq[mt][q][0], (unsigned short)(hq[mt][q][0] >> 16),
                                            (unsigned short)hq[mt][q][1], (unsigned short)(hq[mt][q][1] >> 16)};
              const unsigned short lv[4] = {(unsigned short)lq[mt][q][0], (unsigned short)(lq[mt][q][0] >> 16),
                                            (unsigned short)lq[mt][q][1], (unsigned short)(lq[mt][q][1] >> 16)};
              if (jg + 8 <= N) {                                    // (wave-uniform: all 8 rows of the group exist)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  char* o = qa[(q & 1) * 4 + e] + (jg + e) * 512;
                  GM_QW(o, hv[e]);
                  GM_QW(o + 256, lv[e]);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  char* o = qa[(q & 1) * 4 + e] + (jg + e) * 512;
                  if (jg + 4 * h + e < N) {
                    *reinterpret_cast<unsigned short*>(o) = hv[e];
                    *reinterpret_cast<unsigned short*>(o + 256) = lv[e];
                  }
                }
              }
            }
        }
      }
      GM_STAMP(2);
      GM_SYNC();
      GM_STAMP(3);
      // ---- G2: E^T[j][i], softmax over j per column i, A planes [j][i] * 2^8
      if (g2_active) {
        GM_FRESH(rsw, rsw_) GM_FRESH(xsw, xsw_)
        f32x16 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.f;
        unsigned rswb = rswb_, xswb = xswb_;
        asm volatile("" : "+v"(rswb), "+v"(xswb));
        const char* qbase = lds + UO;
        uint4 a[2][MT][2], bq[2][1][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const char* ap = qbase + (rsw[mt] + xsw[mt]);
          a[0][mt][0] = lds128(ap, 0);
          a[0][mt][1] = lds128(ap, 256);
        }
        {
          const char* bp = lds + (rswb + xswb);
          bq[0][0][0] = lds128(bp, 0);
          bq[0][0][1] = lds128(bp, 256);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const char* ap = qbase + (rsw[mt] + (xsw[mt] ^ ((ks + 1) << 5)));
              a[(ks + 1) & 1][mt][0] = lds128(ap, 0);
              a[(ks + 1) & 1][mt][1] = lds128(ap, 256);
            }
            const char* bp = lds + (rswb + (xswb ^ ((ks + 1) << 5)));
            bq[(ks + 1) & 1][0][0] = lds128(bp, 0);
            bq[(ks + 1) & 1][0][1] = lds128(bp, 256);
          }
          mma_step<MT, 1>(acc, a[ks & 1], bq[ks & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        GM_STAMP(12);
        // masked softmax of row i = cw over its edges j (graphML.py:1771-1776): in-lane over the MT * 16 accumulator
        // registers, one exchange with the partner lane (the other 4-row halves of the same column).  Entries without an
        // edge become -inf by a bit-field insert under the sign-extended mask bit; exp2(-inf) = 0 needs no select.
        float mx = -__builtin_inff();
        unsigned mkl[4] = {mk[0], mk[1], mk[2], mk[3]};      // (laundered: the per-entry masks are formed here, per head)
        asm volatile("" : "+v"(mkl[0]), "+v"(mkl[1]), "+v"(mkl[2]), "+v"(mkl[3]));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)mkl[mt], 8 * (r >> 2) + (r & 3), 1);
            const float ev = acc[0][mt][r];      // (a scalar copy: bit-casting the vector element itself reads element 0)
            const unsigned u = (__builtin_bit_cast(unsigned, ev) & m) | (0xff800000u & ~m);
            const float em = __builtin_bit_cast(float, u);
            acc[0][mt][r] = em;
            mx = fmaxf(mx, em);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        constexpr float kLog2e = 1.4426950408889634f;
        const float cexp = mx > -__builtin_inff() ? -mx * kLog2e : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[0][mt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[0][mt][r], kLog2e, cexp));
            sum += acc[0][mt][r];
          }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = sum > 0.f ? 256.f / sum : 0.f;
        GM_STAMP(13);
        if (cw < KI) {
          char* ab = lds + (AO + 4 * h * SA + cw * 2);
          char* al = ab + AP;
          const f32x2 sc = {inv, inv};
          unsigned ha[MT][4][2], la[MT][4][2];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2 v01 = f32x2{acc[0][mt][4 * q], acc[0][mt][4 * q + 1]} * sc;
              const f32x2 v23 = f32x2{acc[0][mt][4 * q + 2], acc[0][mt][4 * q + 3]} * sc;
              split_pair(v01[0], v01[1], ha[mt][q][0], la[mt][q][0]);
              split_pair(v23[0], v23[1], ha[mt][q][1], la[mt][q][1]);
            }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int jg = 32 * mt + 8 * q;
              if (jg >= N) break;
              const unsigned short hv[4] = {(unsigned short)ha[mt][q][0], (unsigned short)(ha[mt][q][0] >> 16),
                                            (unsigned short)ha[mt][q][1], (unsigned short)(ha[mt][q][1] >> 16)};
              const unsigned short lv[4] = {(unsigned short)la[mt][q][0], (unsigned short)(la[mt][q][0] >> 16),
                                            (unsigned short)la[mt][q][1], (unsigned short)(la[mt][q][1] >> 16)};
              if (jg + 8 <= N) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  GM_AW(ab + (jg + e) * SA, hv[e]);
                  GM_AW(al + (jg + e) * SA, lv[e]);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (jg + 4 * h + e < N) {
                    *reinterpret_cast<unsigned short*>(ab + (jg + e) * SA) = hv[e];
                    *reinterpret_cast<unsigned short*>(al + (jg + e) * SA) = lv[e];
                  }
              }
            }
        }
      }
      GM_STAMP(4);
      // ---- G3: U_k[i][c] for the K taps, weights H_pk streamed one k step ahead
      f32x16 acc[KT][MT];
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[k][mt][r] = 0.f;
      {
        GM_FRESH(rsw, rsw_) GM_FRESH(xsw, xsw_)
        const char* s = wl + (size_t)(p.P + hd * KT) * 65536;
        uint4 a[2][MT][2], bw[2][KT][2];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          bw[0][k][0] = *reinterpret_cast<const uint4*>(s + (size_t)k * 65536);
          bw[0][k][1] = *reinterpret_cast<const uint4*>(s + (size_t)k * 65536 + 1024);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const char* ap = lds + (rsw[mt] + xsw[mt]);
          a[0][mt][0] = lds128(ap, 0);
          a[0][mt][1] = lds128(ap, 256);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
              bw[(ks + 1) & 1][k][0] = *reinterpret_cast<const uint4*>(s + (size_t)k * 65536 + (ks + 1) * 2048);
              bw[(ks + 1) & 1][k][1] = *reinterpret_cast<const uint4*>(s + (size_t)k * 65536 + (ks + 1) * 2048 + 1024);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const char* ap = lds + (rsw[mt] + (xsw[mt] ^ ((ks + 1) << 5)));
              a[(ks + 1) & 1][mt][0] = lds128(ap, 0);
              a[(ks + 1) & 1][mt][1] = lds128(ap, 256);
            }
          }
          mma_step<MT, KT>(acc, a[ks & 1], bw[ks & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      GM_STAMP(5);
      GM_SYNC();      // Q is dead everywhere, the A planes are complete
      GM_STAMP(6);
      // ---- hops: acc_k += A^T U_{k+1}; the U^T rows of this wave's c tile are written and read by this wave only
#pragma unroll
      for (int k = KT - 2; k >= 0; --k) {
        // U^T planes [c][i] <- acc_{k+1} 2^-8: a lane holds column c = cw, 4 consecutive i per register quad
        char* ub = lds + (UO + cw * SA + h * 8);       // (+ 4 h rows of the quad: 8 bytes)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i0 = 32 * mt + 8 * q;      // + 4 h
            if (i0 < KI) {                        // (KI is a multiple of 16: both 4-row halves are inside or outside)
              uint2 hi, lo;
              const f32x2 sc = {kInvScale, kInvScale};
              const f32x2 v01 = f32x2{acc[k + 1][mt][4 * q], acc[k + 1][mt][4 * q + 1]} * sc;
              const f32x2 v23 = f32x2{acc[k + 1][mt][4 * q + 2], acc[k + 1][mt][4 * q + 3]} * sc;
              split2v(v01[0], v01[1], hi.x, lo.x, vmax);
              split2v(v23[0], v23[1], hi.y, lo.y, vmax);
              *reinterpret_cast<uint2*>(ub + i0 * 2) = hi;
              *reinterpret_cast<uint2*>(ub + (i0 * 2 + UP)) = lo;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own rows: no barrier
        GM_FRESH(rpa, rpa_)
        const char* up = lds + (UO + cw * SA + h * 16);
        const char* apl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) apl[mt] = lds + rpa[mt];
        uint4 a[2][MT][2], bu[2][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          a[0][mt][0] = lds128(apl[mt], 0);
          a[0][mt][1] = lds128(apl[mt] + AP, 0);
        }
        bu[0][0] = lds128(up, 0);
        bu[0][1] = lds128(up, UP);
#pragma unroll
        for (int ks = 0; ks < KSI; ++ks) {
          if (ks + 1 < KSI) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              a[(ks + 1) & 1][mt][0] = lds128(apl[mt], (ks + 1) * 32);
              a[(ks + 1) & 1][mt][1] = lds128(apl[mt] + AP, (ks + 1) * 32);
            }
            bu[(ks + 1) & 1][0] = lds128(up, (ks + 1) * 32);
            bu[(ks + 1) & 1][1] = lds128(up, (ks + 1) * 32 + UP);
          }
          mma_step_row<MT>(acc[k], a[ks & 1], bu[ks & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        GM_STAMP(7 + (KT - 2 - k));
      }
      // ---- epilogue: Y rows j = 32 mt + 8 q + 4 h + e, column c = cw: a wave-uniform row pointer per register and ONE
      // per-lane byte offset (scalar base + 32-bit vector offset addressing)
      {
        const char* ybase = reinterpret_cast<const char*>(p.Y + (long long)b * N * p.ldy + (CONCAT ? hd * 128 : 0));
        const unsigned lbyte = (unsigned)(cw + 4 * h * p.ldy) * 4u;
        const long long rowb = (long long)p.ldy * 4;
        const bool last = hd == p.P - 1;
        const float fp = (float)p.P;
        auto out = [&](int mt, int r) -> float {      // relu as one v_med3 (fmaxf semantics: a NaN gives 0)
          const float v = acc[0][mt][r] * kInvScale + biasv;
          if constexpr (CONCAT) return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());
          ysum[mt][r] = hd == 0 ? v : ysum[mt][r] + v;
          return __builtin_amdgcn_fmed3f(ysum[mt][r] / fp, 0.f, __builtin_inff());
        };
        float oy[MT][16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oy[mt][r] = out(mt, r);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int jg = 32 * mt + 8 * q;
            if (jg >= N) break;
            const float* o = &oy[mt][4 * q];
            if (CONCAT || last) {
              if (jg + 8 <= N) {
#pragma unroll
                for (int e = 0; e < 4; ++e) GM_YST(const_cast<char*>(ybase + (jg + e) * rowb) + lbyte, o[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (jg + 4 * h + e < N)
                    *reinterpret_cast<float*>(const_cast<char*>(ybase + (jg + e) * rowb) + lbyte) = o[e];
              }
            }
          }
      }
      GM_STAMP(10);
    }
  }
  if (p.range_flag && vmax > 65504.f) atomicOr(p.range_flag, 1);
}

size_t gat_mfma_lds(int N, int ksi) {
  const size_t sa = 2 * 16 * (size_t)ksi + 16;
  return 2 * (size_t)N * 256 + 2 * (size_t)N * sa + 2 * 128 * sa + 16 * (size_t)N;
}

// shape classes: (MT, KSI) = (1, 2) N <= 32, (2, 4) N <= 64, (4, 7) N <= 101 (the LDS bound)
int gat_mfma_class(int N) {
  if (N <= 0) return -1;
  const int c = N <= 32 ? 0 : (N <= 64 ? 1 : 2);
  const int ksi = c == 0 ? 2 : (c == 1 ? 4 : 7);
  return (N <= 16 * ksi && gat_mfma_lds(N, ksi) <= 160 * 1024) ? c : -1;
}

template <int MT, int KSI, int KT>
int launch(const GatMfmaParams& p, int slot, hipStream_t st) {
  const size_t lds = gat_mfma_lds(p.N, KSI);
  const void* fn = p.concat ? reinterpret_cast<const void*>(&gat_mfma_kernel<MT, KSI, KT, true>)
                            : reinterpret_cast<const void*>(&gat_mfma_kernel<MT, KSI, KT, false>);
  if (magat_ensure_dyn_lds(fn, slot + (p.concat ? 0 : 6), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  int cus = 256;
  hipDeviceProp_t prop;
  int dev = 0;
  static int cached_cus = 0;
  if (!cached_cus) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cached_cus = prop.multiProcessorCount;
    else
      cached_cus = 256;
  }
  cus = cached_cus;
  const int blocks = p.B < cus ? p.B : cus;
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_LAYER, st);
  if (p.concat) hipLaunchKernelGGL((gat_mfma_kernel<MT, KSI, KT, true>), dim3(blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gat_mfma_kernel<MT, KSI, KT, false>), dim3(blocks), dim3(256), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

}  // namespace

#ifdef MAGAT_DEBUG_HOOKS
extern "C" int magat_gat_mfma_set_debug_buffer(long long* dev_buf) { g_gat_mfma_dbg = dev_buf; return MAGAT_OK; }
#endif

int magat_gat_mfma_supported(int N, int G, int F, int K, int mode) {
  if (mode != MAGAT_MODE_KEYQUERY || G != 128 || F != 128 || (K != 2 && K != 3)) return 0;
  return gat_mfma_class(N) >= 0 ? 1 : 0;
}

int magat_gat_mfma_forward(const float* X, int ldx, const void* S, int s_is_f64, const unsigned* rmask_pre,
                           const float* packed_frag, const float* bias, float* Y, int ldy, int B, int N, int K, int P,
                           int concat, int* range_flag, hipStream_t st) {
  GatMfmaParams p;
  p.X = X; p.ldx = ldx; p.S = S; p.s_is_f64 = s_is_f64; p.rmask_pre = rmask_pre;
  p.wfrag = reinterpret_cast<const char*>(packed_frag);
  p.bias = bias; p.Y = Y; p.ldy = ldy; p.B = B; p.N = N; p.P = P; p.concat = concat; p.range_flag = range_flag;
  p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_gat_mfma_dbg;
#endif
  const int cls = gat_mfma_class(N);
  if (cls == 0) return K == 3 ? launch<1, 2, 3>(p, MAGAT_LDS_GATM_0, st) : launch<1, 2, 2>(p, MAGAT_LDS_GATM_0 + 1, st);
  if (cls == 1) return K == 3 ? launch<2, 4, 3>(p, MAGAT_LDS_GATM_0 + 2, st) : launch<2, 4, 2>(p, MAGAT_LDS_GATM_0 + 3, st);
  if (cls == 2) return K == 3 ? launch<4, 7, 3>(p, MAGAT_LDS_GATM_0 + 4, st) : launch<4, 7, 2>(p, MAGAT_LDS_GATM_0 + 5, st);
  return MAGAT_ERR_UNSUPPORTED;
}
