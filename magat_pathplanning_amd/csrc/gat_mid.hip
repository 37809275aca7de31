// GraphFilterBatchAttentional.forward (KeyQuery attention) as ONE launch of matrix-core products for the PUBLISHED feature widths
// on graphs of 33 .. 128 agents: G = F in {32, 64}, K = 2 | 3 - the released "MAGAT F-32-P4 / B-32-P4" checkpoints
// (scripts/train_DMap.sh:42-46) evaluated on the README's 30 / 40 / 50 / 60 / 100-robot generalisation sets (README.md:372-390;
// reference utils/graphUtils/graphML.py:4636-4671, 1724-1827, 1180-1286).  gat_small.hip covers N <= 32 (a wave per instance),
// gat_mfma.hip G = 128; before round 6 these shapes took the two-launch form (maps GEMM + gat_dense_kernel, Z through HBM:
// 194 us per 512 x 100 agents at F = 32).  Same algebra and f16x3 arithmetic as gat_small.hip (two f16 planes per operand,
// three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; per instance and head):
//   G1  Q[j][g]   = sum_f X[j][f] W_p[g][f]
//   G2  E^T[j][i] = sum_g Q[j][g] X[i][g]; masked softmax over j in the accumulator layout (a lane owns column i) -> A planes
//   G3  U_k[i][c] = sum_f X[i][f] H_pk[c][f],  k = 0..K-1
//   hops acc_k[j][c] += sum_i A[j][i] U^T[c][i],  k = K-2 .. 0  (Horner);  Y_p = relu(acc_0 2^-8 + bias) | head mean
// organised by ROW TILES: a workgroup of NT = ceil(N / 32) waves owns a planning instance, wave w the agents 32 w .. 32 w + 31 -
// as rows j of Q (G1), as columns i of the score tile (G2: its rows of the attention matrix, softmax in registers over all NT
// row tiles), as rows i of U_k (G3) and as output rows j of the hops.  X, Q / U^T (one region: Q is dead when U^T is written)
// and A planes are workgroup LDS (140 KB at N = 128, F = 64; 54 KB at N <= 64: two workgroups per CU); four or five workgroup
// barriers per head.  Weights come straight from the row-major f16 planes of the layer's pack (L2 / L1 hits), the edge masks
// from coalesced row reads of S + ballots.  Values beyond the f16 range raise range_flag: the caller's predicated float32 form
// rewrites the output (its LDS tiles reach N = 128 at these widths).
// G = F = 128 on 103 .. 128 agents (XR form, round 6): every use of the X planes is of the wave's OWN row tile, so they need no LDS
// at all - a lane keeps its operand fragments of row 32 w + lane % 32 in 64 registers, read straight from HBM in that layout.  What
// is left - Q / U^T and A planes - is 139 KB: the one-launch form gat_mfma.hip (162 KB per instance, ends at N = 102) cannot reach.
#include "magat_common.h"


#ifdef MAGAT_DEBUG_HOOKS
// phase stamps (debug builds: tools/exp/mid_phase_probe.py): [workgroup][wave][16] cycle counters of the LAST head a wave ran
#define MID_STAMP(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
static long long* g_gat_mid_dbg = nullptr;
extern "C" int magat_gat_mid_set_debug_buffer(long long* dev_buf) { g_gat_mid_dbg = dev_buf; return MAGAT_OK; }
#else
#define MID_STAMP(i) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct GatMidParams {
  const float* X;             // [B*N][ldx]
  const void* S;              // [B][N][N] f32 | f64
  const unsigned short* Hs;   // f16 planes [2][NC][G] of 2^8 Bt (rows: [P][G] W_p, then [P][K][F] H_pk)
  const float* bias;          // [F] or null
  float* Y;                   // [B*N][ldy]
  int B, N, P, NC, ldx, ldy, s_is_f64;
  int* range_flag;
  const float* x_scale;
  const char* wfrag;          // XR (G = 128): the same planes fragment-major (gat_mfma.hip's stream: 64 KB blocks [P] W_p, [P K] H_pk,
                              // a block = [32-row tile 4][k step 8][plane 2][lane 64][8 halves]) - one coalesced 1 KB read per fragment
  long long* dbg;             // debug builds: phase stamps
  float* Ypre; int ldpre;     // HS, head-mean: the heads' pre-activation outputs [B*N][P F] (the caller's workspace); a small kernel forms the mean
};

__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two floats -> the two f16 planes (value = p1 + p2 to 22 bits).  Compiler-visible conversions only: an asm statement reading a
// register an MFMA has just written gets none of the wait states the hazard recognizer inserts (round 6, gat_csr_fused.hip)
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2) {
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  const f32x2 back = __builtin_convertvector(h, f32x2);
  const f16x2 r = __builtin_convertvector(f32x2{x - back[0], y - back[1]}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split2v(float x, float y, unsigned& p1, unsigned& p2, float& vmax) {
  vmax = fmaxf(fmaxf(vmax, fabsf(x)), fabsf(y));
  split_pair(x, y, p1, p2);
}

// HS (few instances: the batch-1 step of the reference's inference loop): a workgroup per (instance, HEAD) instead of per instance -
// one planning instance of 100 agents then runs on four CUs instead of one.  A head's arithmetic does not change; with head-mean
// the workgroups leave their pre-activation outputs in Ypre and gat_mid_mean_kernel sums them in the loop's order (bit-identical).
template <int F, int KT, int NT, bool CONCAT, bool HS = false, bool XR = false>
__global__ __launch_bounds__(64 * NT) void gat_mid_kernel(const GatMidParams p) {
  extern __shared__ __align__(16) char lds[];
  constexpr int CT = F / 32, KF = F / 16, ROWS = 32 * NT, THREADS = 64 * NT;
  constexpr int RS = 2 * F + 16;             // row stride of the X / Q planes (bytes)
  constexpr int SA = 2 * ROWS + 16;          // row stride of the A / U^T planes (bytes): ROWS columns of halves + 16
  constexpr int XPL = ROWS * RS, APL = ROWS * SA, UPL = F * SA;
  constexpr int XO = 0, QO = XR ? 0 : 2 * XPL, UO = QO;      // (XR: the X planes live in registers)
  constexpr int AO = QO + (2 * XPL > 2 * UPL ? 2 * XPL : 2 * UPL);
  constexpr float kInvScale = 1.f / 256.f;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int N = p.N, G = F;
  float xs = 1.f;
  if (p.x_scale) xs = *p.x_scale;
  if (xs == 0.f) xs = 1.f;
  const float ixs = 1.f / xs;
  const float kOutScale = kInvScale * ixs;
  const float kLog2e = 1.4426950408889634f * ixs * ixs;
  float vmax = 0.f;
  const long long plane = (long long)p.NC * G;      // halves per weight plane
  float biasv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) biasv[ct] = p.bias ? p.bias[32 * ct + fr] : 0.f;
  const int myrow = 32 * w + fr;                    // this lane's agent (row j of Q / A^T, column i of the scores, ...)

  const int ninst = HS ? (int)gridDim.x / p.P : (int)gridDim.x;      // (HS: gridDim.x = instances-in-flight x P)
  const int hlo = HS ? (int)blockIdx.x % p.P : 0;
  int hhi = HS ? hlo + 1 : p.P;
  if constexpr (HS && XR) asm volatile("" : "+s"(hhi));      // (a head LOOP for the compiler too: as straight-line code this form spilled 289 .. 468 registers)
  for (int inst = HS ? (int)blockIdx.x / p.P : (int)blockIdx.x; inst < p.B; inst += ninst) {
    __syncthreads();      // every wave is done with the previous instance's planes
    // ---- X rows -> f16 planes (rows past N: zeros): all waves into LDS, or (XR) every lane its own operand fragments
    uint4 xr[XR ? KF : 1][2];
    if constexpr (XR) {
      const float* xrow = p.X + ((long long)inst * N + (myrow < N ? myrow : N - 1)) * p.ldx + 8 * fh;
#pragma unroll
      for (int ks = 0; ks < KF; ++ks) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(xrow + 16 * ks), v1 = *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4);
        if (myrow >= N) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
        float xv[8] = {v0[0] * xs, v0[1] * xs, v0[2] * xs, v0[3] * xs, v1[0] * xs, v1[1] * xs, v1[2] * xs, v1[3] * xs};
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) bad |= !(fabsf(xv[e]) <= 65504.f);
        if (bad) vmax = __builtin_inff();
        split2v(xv[0], xv[1], xr[ks][0].x, xr[ks][1].x, vmax);
        split2v(xv[2], xv[3], xr[ks][0].y, xr[ks][1].y, vmax);
        split2v(xv[4], xv[5], xr[ks][0].z, xr[ks][1].z, vmax);
        split2v(xv[6], xv[7], xr[ks][0].w, xr[ks][1].w, vmax);
      }
    } else {
      const float* Xb = p.X + (long long)inst * N * p.ldx;
      for (int idx = t; idx < ROWS * (F / 8); idx += THREADS) {
        const int row = idx / (F / 8), ch = idx % (F / 8);
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (row < N) {
          v0 = *reinterpret_cast<const f32x4*>(Xb + (long long)row * p.ldx + 8 * ch);
          v1 = *reinterpret_cast<const f32x4*>(Xb + (long long)row * p.ldx + 8 * ch + 4);
        }
        float xv[8] = {v0[0] * xs, v0[1] * xs, v0[2] * xs, v0[3] * xs, v1[0] * xs, v1[1] * xs, v1[2] * xs, v1[3] * xs};
        bool bad = false;      // (a NaN must raise the flag too: fmaxf drops it from the running maximum)
#pragma unroll
        for (int e = 0; e < 8; ++e) bad |= !(fabsf(xv[e]) <= 65504.f);
        if (bad) vmax = __builtin_inff();
        uint4 hi, lo;
        split2v(xv[0], xv[1], hi.x, lo.x, vmax);
        split2v(xv[2], xv[3], hi.y, lo.y, vmax);
        split2v(xv[4], xv[5], hi.z, lo.z, vmax);
        split2v(xv[6], xv[7], hi.w, lo.w, vmax);
        char* dst = lds + XO + row * RS + ch * 16;
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + XPL) = lo;
      }
    }
    // ---- edge masks of this wave's 32 rows: bit j of word j / 32 <=> |S[i][j]| > 1e-9 (graphML.py:1274-1276; NaN: no edge).
    // A row is read by the whole wave (lane = column, 256-byte runs), the bits come from a ballot; lane (fr, .) keeps row fr's
    unsigned mk[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) mk[jt] = 0u;
    // (eight rows per batch: their loads are in flight together - one row at a time was a serial chain of 64 L2 round trips.  The
    //  scheduling barrier keeps them so: without it the compiler moved every load next to its ballot again, one round trip a row)
    //  The float32 | float64 choice is made ONCE around the loop: inside it, every load sat in its own branch diamond.)
    constexpr int HC = (ROWS + 63) / 64;
    auto build_masks = [&](auto tag) __attribute__((always_inline)) {
      typedef decltype(tag) ST;
      const ST* Sp = static_cast<const ST*>(p.S) + (long long)inst * N * N;
#pragma unroll 1
      for (int il0 = 0; il0 < 32; il0 += 8) {
        ST sv[8][HC];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = 32 * w + il0 + u;
#pragma unroll
          for (int hc = 0; hc < HC; ++hc) {
            const int j = 64 * hc + lane;
            // (branch-free: clamped addresses, the predicate applied to the loaded value - the batch's loads issue back to back)
            sv[u][hc] = Sp[(long long)(i < N ? i : N - 1) * N + (j < N ? j : N - 1)];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int hc = 0; hc < HC; ++hc) {
            const int i = 32 * w + il0 + u, j = 64 * hc + lane;
            const bool e_ = (sv[u][hc] < (ST)0 ? -sv[u][hc] : sv[u][hc]) > (ST)1e-9 && i < N && j < N;      // (NaN: no edge)
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(e_);
            if (fr == il0 + u) {
              if (2 * hc < NT) mk[2 * hc] = (unsigned)bal;
              if (2 * hc + 1 < NT) mk[2 * hc + 1] = (unsigned)(bal >> 32);
            }
          }
      }
    };
    if (p.s_is_f64) build_masks(double{});
    else build_masks(float{});
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) mk[jt] >>= 4 * fh;      // bit (8 (r / 4) + r % 4) = row j of accumulator register r
    __syncthreads();      // X planes complete

    float ysum[(CONCAT || HS || XR) ? 1 : CT][16];      // (XR: the heads' outputs go through Ypre like the head-split form's)
    if constexpr (!CONCAT && !HS && !XR) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ysum[ct][r] = 0.f;
    }
#pragma unroll 1
    for (int hd = hlo; hd < hhi; ++hd) {
      MID_STAMP(0);
      // this lane's 16-byte pieces of weight row (base + lane % 32): k step ks, plane pl at + pl * plane + 16 ks + 8 fh halves
      auto wfrag = [&](long long row0, int ks, int pl) __attribute__((always_inline)) {
        if constexpr (XR)      // (row0 = 128 block + 32 tile: blocks and tiles are 64 KB / 16 KB apart)
          return *reinterpret_cast<const uint4*>(p.wfrag + row0 * 512 + (2 * ks + pl) * 1024 + lane * 16);
        else
          return *reinterpret_cast<const uint4*>(p.Hs + pl * plane + (row0 + fr) * G + 16 * ks + 8 * fh);
      };
      // operand rows of tile `tile` of the X planes
      // (only ever the wave's own tile - which is why XR can keep them in registers)
      auto xfrag = [&](int tile, int ks, int pl) __attribute__((always_inline)) {
        if constexpr (XR) return xr[ks][pl];
        else return *reinterpret_cast<const uint4*>(lds + XO + pl * XPL + (32 * tile + fr) * RS + (16 * ks + 8 * fh) * 2);
      };
      // ---- G1 (operands swapped): Q^T tile of this wave's agents, lane = agent row j, register quads = 4 consecutive columns g.
      // The weight fragments of a whole batch of column tiles (all of them; one at 128 features: 64 registers) are requested
      // before the first product: one exposed L2 round trip per batch - requested k step by k step in front of their three
      // matrix instructions, a wave (alone on its SIMD) sat out one per k step: 20 k cycles for 96 instructions at 128 features
      constexpr int WB = F == 128 ? 1 : CT;      // column tiles per weight batch
#pragma unroll
      for (int cb = 0; cb < CT; cb += WB) {
        uint4 wb[WB][KF][2];
#pragma unroll
        for (int c_ = 0; c_ < WB; ++c_)
#pragma unroll
          for (int ks = 0; ks < KF; ++ks) {
            wb[c_][ks][0] = wfrag((long long)hd * G + 32 * (cb + c_), ks, 0);
            wb[c_][ks][1] = wfrag((long long)hd * G + 32 * (cb + c_), ks, 1);
          }
        __builtin_amdgcn_sched_barrier(0);      // (the requests stay in front of the products: the scheduler sinks them otherwise)
#pragma unroll
        for (int c_ = 0; c_ < WB; ++c_) {
          const int ct = cb + c_;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KF; ++ks) {
            const uint4 w0 = wb[c_][ks][0], w1 = wb[c_][ks][1], x0 = xfrag(w, ks, 0), x1 = xfrag(w, ks, 1);
            acc = mfma16(w0, x0, acc);
            acc = mfma16(w1, x0, acc);
            acc = mfma16(w0, x1, acc);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 hi, lo;
            split2v(acc[4 * q] * kInvScale, acc[4 * q + 1] * kInvScale, hi.x, lo.x, vmax);
            split2v(acc[4 * q + 2] * kInvScale, acc[4 * q + 3] * kInvScale, hi.y, lo.y, vmax);
            char* o = lds + QO + myrow * RS + (32 * ct + 8 * q + 4 * fh) * 2;
            *reinterpret_cast<uint2*>(o) = hi;
            *reinterpret_cast<uint2*>(o + XPL) = lo;
          }
        }
      }
      MID_STAMP(1);
      __syncthreads();      // Q planes complete (and every wave is past the previous head's reads of the U^T region)
      MID_STAMP(2);
      // 32 features on more than 64 agents (one wave per SIMD whatever its registers: the LDS decides): the K taps' fragments -
      // 4 K register quads - are requested here, behind the barrier (which drains the request counter), and fly under G2, the
      // softmax and the A planes
      constexpr bool WEARLY = !XR && F == 32 && NT >= 3;
      uint4 wt[WEARLY ? KT : 1][CT][KF][2];
      if constexpr (WEARLY) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
              const long long row0 = (long long)p.P * G + ((long long)hd * KT + k) * F + 32 * ct;
              wt[k][ct][ks][0] = wfrag(row0, ks, 0);
              wt[k][ct][ks][1] = wfrag(row0, ks, 1);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- G2: E^T[j][i] = sum_g Q[j][g] X[i][g] for this wave's columns i and ALL row tiles j; lane = column i, registers = rows j
      f32x16 e[NT];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) e[jt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
          const char* qp = lds + QO + (32 * jt + fr) * RS + (16 * ks + 8 * fh) * 2;
          const uint4 q0 = *reinterpret_cast<const uint4*>(qp), q1 = *reinterpret_cast<const uint4*>(qp + XPL);
          const uint4 x0 = xfrag(w, ks, 0), x1 = xfrag(w, ks, 1);
          e[jt] = mfma16(q0, x0, e[jt]);
          e[jt] = mfma16(q0, x1, e[jt]);
          e[jt] = mfma16(q1, x0, e[jt]);
        }
      }
      MID_STAMP(3);
      // masked softmax of row i over its edges j (in-lane over the NT tiles + the partner lane), A planes [j][i] * 2^8
      float mx = -__builtin_inff();
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)mk[jt], 8 * (r >> 2) + (r & 3), 1);
          const float ev = e[jt][r];      // (a scalar copy: bit-casting the vector element itself reads element 0)
          const float em = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ev) & m) | (0xff800000u & ~m));
          e[jt][r] = em;
          mx = fmaxf(mx, em);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float cexp = mx > -__builtin_inff() ? -mx * kLog2e : 0.f;
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          e[jt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[jt][r], kLog2e, cexp));
          sum += e[jt][r];
        }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = sum > 0.f ? 256.f / sum : 0.f;
      MID_STAMP(4);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned ha[2], la[2];
          split_pair(e[jt][4 * q] * inv, e[jt][4 * q + 1] * inv, ha[0], la[0]);
          split_pair(e[jt][4 * q + 2] * inv, e[jt][4 * q + 3] * inv, ha[1], la[1]);
          char* o = lds + AO + (32 * jt + 8 * q + 4 * fh) * SA + myrow * 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<unsigned short*>(o + c * SA) = (unsigned short)(ha[c >> 1] >> (16 * (c & 1)));
            *reinterpret_cast<unsigned short*>(o + c * SA + APL) = (unsigned short)(la[c >> 1] >> (16 * (c & 1)));
          }
        }
      MID_STAMP(5);
      // ---- G3: U_k[i][c] for this wave's agents i and the K taps (lane = column c, registers = rows i).  XR: a tap's product is
      // formed when its accumulators are first needed (tap K - 1 here, tap k in front of hop k) - the same chain of products into
      // the same accumulators, but 64 instead of 192 of them live at 128 features and three taps
      f32x16 acc[KT][CT];
      auto g3 = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < CT; cb += WB) {
          uint4 wb[WB][KF][2];      // (a batch of weight fragments up front, like G1)
#pragma unroll
          for (int c_ = 0; c_ < WB; ++c_)
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
              const long long row0 = (long long)p.P * G + ((long long)hd * KT + k) * F + 32 * (cb + c_);
              if constexpr (WEARLY) { wb[c_][ks][0] = wt[k][cb + c_][ks][0]; wb[c_][ks][1] = wt[k][cb + c_][ks][1]; }
              else { wb[c_][ks][0] = wfrag(row0, ks, 0); wb[c_][ks][1] = wfrag(row0, ks, 1); }
            }
          if constexpr (!WEARLY) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c_ = 0; c_ < WB; ++c_) {
            const int ct = cb + c_;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][ct][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
              const uint4 w0 = wb[c_][ks][0], w1 = wb[c_][ks][1], x0 = xfrag(w, ks, 0), x1 = xfrag(w, ks, 1);
              acc[k][ct] = mfma16(x0, w0, acc[k][ct]);
              acc[k][ct] = mfma16(x0, w1, acc[k][ct]);
              acc[k][ct] = mfma16(x1, w0, acc[k][ct]);
            }
          }
        }
      };
      if constexpr (XR) g3(KT - 1);
      else {
#pragma unroll
        for (int k = 0; k < KT; ++k) g3(k);
      }
      MID_STAMP(6);
      __syncthreads();      // A planes complete; every wave is done reading the Q planes (the U^T planes take their place)
      MID_STAMP(7);
      // ---- hops (Horner): acc_k[j-tile w] += A[j][all i] U_{k+1}[all i]; the U^T planes [c][i] are rewritten from acc_{k+1}
#pragma unroll
      for (int k = KT - 2; k >= 0; --k) {
        if (k < KT - 2) __syncthreads();      // every wave is done reading the previous hop's U^T planes
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 hi, lo;
            split2v(acc[k + 1][ct][4 * q] * kInvScale, acc[k + 1][ct][4 * q + 1] * kInvScale, hi.x, lo.x, vmax);
            split2v(acc[k + 1][ct][4 * q + 2] * kInvScale, acc[k + 1][ct][4 * q + 3] * kInvScale, hi.y, lo.y, vmax);
            char* o = lds + UO + (32 * ct + fr) * SA + (32 * w + 8 * q + 4 * fh) * 2;
            *reinterpret_cast<uint2*>(o) = hi;
            *reinterpret_cast<uint2*>(o + UPL) = lo;
          }
        if constexpr (XR) g3(k);
        __syncthreads();      // U^T planes complete
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) {
          const char* ap = lds + AO + myrow * SA + (16 * ks + 8 * fh) * 2;
          const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + APL);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const char* up = lds + UO + (32 * ct + fr) * SA + (16 * ks + 8 * fh) * 2;
            const uint4 u0 = *reinterpret_cast<const uint4*>(up), u1 = *reinterpret_cast<const uint4*>(up + UPL);
            acc[k][ct] = mfma16(a0, u0, acc[k][ct]);
            acc[k][ct] = mfma16(a0, u1, acc[k][ct]);
            acc[k][ct] = mfma16(a1, u0, acc[k][ct]);
          }
        }
      }
      MID_STAMP(8);
      // ---- epilogue: lane = column c, registers = rows j of this wave's tile
      float* yb = p.Y + (long long)inst * N * p.ldy;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * w + 8 * (r >> 2) + 4 * fh + (r & 3);
          const float v = __builtin_fmaf(acc[0][ct][r], kOutScale, biasv[ct]);
          if constexpr (CONCAT) {
            if (j < N) yb[(long long)j * p.ldy + hd * F + 32 * ct + fr] = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());
          } else if constexpr (HS || XR) {
            if (j < N) p.Ypre[((long long)inst * N + j) * p.ldpre + hd * F + 32 * ct + fr] = v;
          } else {
            ysum[ct][r] += v;      // (graphML.py:4663-4667: mean over the heads, then ReLU)
            if (hd == p.P - 1 && j < N)
              yb[(long long)j * p.ldy + 32 * ct + fr] = __builtin_amdgcn_fmed3f(ysum[ct][r] / (float)p.P, 0.f, __builtin_inff());
          }
        }
      MID_STAMP(9);
      __syncthreads();      // every wave is done with this head's U^T / A planes (the next head's G1 writes the Q planes)
      MID_STAMP(10);
    }
  }
  if (p.range_flag && vmax > 65504.f) atomicOr(p.range_flag, 1);
}

// head mean of the head-split form: y = relu((((0 + v_0) + v_1) + ..) / P) - the non-split kernel's sum, in its order
__global__ __launch_bounds__(256) void gat_mid_mean_kernel(const float* __restrict__ ypre, float* __restrict__ y, long long M, int P, int F,
                                                           int ldpre, int ldy) {
  const long long total = M * F;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    const long long m = idx / F;
    const int c = (int)(idx - m * F);
    float s = 0.f;
    for (int q = 0; q < P; ++q) s += ypre[m * ldpre + q * F + c];
    y[m * ldy + c] = __builtin_amdgcn_fmed3f(s / (float)P, 0.f, __builtin_inff());
  }
}

template <int F, int NT, bool XR = false>
constexpr size_t mid_lds() {
  constexpr size_t ROWS = 32 * NT, RS = 2 * F + 16, SA = 2 * ROWS + 16;
  constexpr size_t q = 2 * ROWS * RS, u = 2 * F * SA;
  return (XR ? 0 : 2 * ROWS * RS) + (q > u ? q : u) + 2 * ROWS * SA;
}

constexpr long long MID_HS_MAX_WG = 64;      // head split while instances x heads stay below this many workgroups

// slot: LDS-attribute slots of the four kernels of this (width, taps, row tiles) class - concat, mean, and the two head-split forms
template <int F, int KT, int NT, bool XR = false>
int launch_mid(const GatMidParams& p, int concat, int slot, int slot_hs, hipStream_t st) {
  constexpr size_t lds = mid_lds<F, NT, XR>();
  static_assert(lds <= 160 * 1024, "LDS");
  const void* fn = concat ? reinterpret_cast<const void*>(&gat_mid_kernel<F, KT, NT, true, false, XR>)
                          : reinterpret_cast<const void*>(&gat_mid_kernel<F, KT, NT, false, false, XR>);
  if (magat_ensure_dyn_lds(fn, slot + (concat ? 0 : 1), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  long long per_cu = (160 * 1024) / (long long)lds;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  long long blocks = p.B;
  if (blocks > cus * per_cu) blocks = cus * per_cu;
  // few instances (the batch-1 step): a workgroup per (instance, head); head-mean then needs the caller's scratch rows
  const bool hs = p.P > 1 && (long long)p.B * p.P <= MID_HS_MAX_WG && (concat || p.Ypre);
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_LAYER, st);
  if (hs) {
    const void* fh = concat ? reinterpret_cast<const void*>(&gat_mid_kernel<F, KT, NT, true, true, XR>)
                            : reinterpret_cast<const void*>(&gat_mid_kernel<F, KT, NT, false, true, XR>);
    if (magat_ensure_dyn_lds(fh, slot_hs + (concat ? 0 : 1), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
    const unsigned g = (unsigned)(p.B * p.P);
    if (concat) hipLaunchKernelGGL((gat_mid_kernel<F, KT, NT, true, true, XR>), dim3(g), dim3(64 * NT), lds, st, p);
    else {
      hipLaunchKernelGGL((gat_mid_kernel<F, KT, NT, false, true, XR>), dim3(g), dim3(64 * NT), lds, st, p);
      const long long M = (long long)p.B * p.N;
      hipLaunchKernelGGL(gat_mid_mean_kernel, dim3((unsigned)((M * F + 255) / 256)), dim3(256), 0, st, p.Ypre, p.Y, M, p.P, F, p.ldpre, p.ldy);
    }
    magat_form_note(MAGAT_FORM_GAT_HSPLIT);
  } else if (concat) hipLaunchKernelGGL((gat_mid_kernel<F, KT, NT, true, false, XR>), dim3((unsigned)blocks), dim3(64 * NT), lds, st, p);
  else {
    hipLaunchKernelGGL((gat_mid_kernel<F, KT, NT, false, false, XR>), dim3((unsigned)blocks), dim3(64 * NT), lds, st, p);
    if (XR) {      // (its head mean: the same sum in the same order, from the pre-activation rows)
      const long long M = (long long)p.B * p.N;
      long long mb = (M * F + 255) / 256;
      if (mb > 4096) mb = 4096;
      hipLaunchKernelGGL(gat_mid_mean_kernel, dim3((unsigned)mb), dim3(256), 0, st, p.Ypre, p.Y, M, p.P, F, p.ldpre, p.ldy);
    }
  }
  magat_prof_end(pid, st);
  magat_form_note(MAGAT_FORM_GAT_MID);
  return magat_check_launch();
}

template <int F, int KT>
int launch_mid_nt(const GatMidParams& p, int concat, int slot, hipStream_t st) {
  const int nt = (p.N + 31) / 32;
  if (nt == 2) return launch_mid<F, KT, 2>(p, concat, slot, slot + 24, st);
  if (nt == 3) return launch_mid<F, KT, 3>(p, concat, slot + 2, slot + 26, st);
  return launch_mid<F, KT, 4>(p, concat, slot + 4, slot + 28, st);
}

}  // namespace

// head mean of pre-activation rows [M][ldpre >= P F] (gat_mfma.hip's head-split form merges its heads with it too)
int magat_gat_mean_launch(const float* ypre, float* y, long long M, int P, int F, int ldpre, int ldy, hipStream_t st) {
  long long mb = (M * F + 255) / 256;
  if (mb > 4096) mb = 4096;
  hipLaunchKernelGGL(gat_mid_mean_kernel, dim3((unsigned)mb), dim3(256), 0, st, ypre, y, M, P, F, ldpre, ldy);
  return magat_check_launch();
}

int magat_gat_mid_supported(int N, int G, int F, int K, int mode) {
  if (mode != MAGAT_MODE_KEYQUERY || G != F || (K != 2 && K != 3) || N > 128) return 0;
  if (G == 128) {      // (gat_mfma.hip up to 102 agents; option GAT_WIDE_FROM: where this form takes over from the two launches)
    const int from = magat_opt(MAGAT_OPT_GAT_WIDE_FROM);
    return N >= (from < 103 ? 103 : from);
  }
  return N >= 33 && (G == 32 || G == 64);
}

// Hs: the f16 planes [2][NC][G] of the layer's pack (packed + magat_gat_f16_block_offset(NC, G))
int magat_gat_mid_forward(const float* X, int ldx, const void* S, int s_is_f64, const float* Hs, int NC, const float* bias, float* Y,
                          int ldy, int B, int N, int G, int K, int P, int concat, int* range_flag, hipStream_t st,
                          const float* x_scale, float* ypre, int ldpre, const float* wfrag) {
  // (any N of the four-row-tile class is accepted here for G = 128; WHERE this form takes over is the dispatcher's decision,
  //  magat_gat_mid_supported / option GAT_WIDE_FROM)
  const bool wide_ok = G == 128 && N >= 97 && N <= 128 && (K == 2 || K == 3);
  if (!wide_ok && !magat_gat_mid_supported(N, G, G, K, MAGAT_MODE_KEYQUERY)) return MAGAT_ERR_UNSUPPORTED;
  if ((ldx & 3) || (reinterpret_cast<uintptr_t>(X) & 15)) return MAGAT_ERR_UNSUPPORTED;
  GatMidParams p;
  p.X = X; p.S = S; p.Hs = reinterpret_cast<const unsigned short*>(Hs); p.bias = bias; p.Y = Y;
  p.B = B; p.N = N; p.P = P; p.NC = NC; p.ldx = ldx; p.ldy = ldy; p.s_is_f64 = s_is_f64;
  p.range_flag = range_flag; p.x_scale = x_scale;
  p.dbg = nullptr;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_gat_mid_dbg;
#endif
  p.Ypre = (ypre && ldpre >= P * G) ? ypre : nullptr; p.ldpre = ldpre;
  p.wfrag = reinterpret_cast<const char*>(wfrag);
  if (G == 128 && !wfrag) return MAGAT_ERR_NULL;
  if (G == 128 && !concat && !p.Ypre) return MAGAT_ERR_WORKSPACE;      // (the 128-wide form always merges heads through Ypre)
  // LDS-attribute slots: 6 per (width, taps) pair (three tile counts x two merges), 24 more for their head-split forms, then the
  // eight of the 128-wide form (taps x merge, + head split)
  if (G == 128)
    return K == 3 ? launch_mid<128, 3, 4, true>(p, concat, MAGAT_LDS_GATD_0 + 48, MAGAT_LDS_GATD_0 + 52, st)
                  : launch_mid<128, 2, 4, true>(p, concat, MAGAT_LDS_GATD_0 + 50, MAGAT_LDS_GATD_0 + 54, st);
  if (G == 32) return K == 3 ? launch_mid_nt<32, 3>(p, concat, MAGAT_LDS_GATD_0, st) : launch_mid_nt<32, 2>(p, concat, MAGAT_LDS_GATD_0 + 6, st);
  return K == 3 ? launch_mid_nt<64, 3>(p, concat, MAGAT_LDS_GATD_0 + 12, st) : launch_mid_nt<64, 2>(p, concat, MAGAT_LDS_GATD_0 + 18, st);
}
