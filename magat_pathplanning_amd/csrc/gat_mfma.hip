// GraphFilterBatchAttentional.forward (KeyQuery attention, 128 features in and out) as ONE launch in which every step is a
// matrix-core product (reference utils/graphUtils/graphML.py:4636-4671, 1724-1827, 1180-1286, 713-823; algebra in
// gat_f32.hip).  Nothing but X, the GSO and Y crosses HBM: the per-agent maps Z = X [W_p | H_pk]^T that the two-launch
// form writes and reads back (0.84 GB per step at N = 100, B = 512) stay in registers and LDS.
//
// Per planning instance b and head p, with X the [N][128] feature rows (all products are f16x3 split products - two f16
// planes per operand, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation; weights carry a fixed 2^8):
//   G1  Q[j][g]   = sum_f X[j][f] W_p[g][f]                      -> planes Q [j][g]        (graphML.py:1767-1769)
//       (issued with swapped operands since round 4: the tile is Q^T and the planes are stored without a transpose)
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
// LDS (N = 100: 162,560 of 163,840 bytes): X planes [N][2][256 B] (16-byte chunks XOR-swizzled by row), A planes [N][SA],
// U^T planes [128][SA] with SA = 2 KI + 16 (KI = 16 KSI >= N columns; rows 60 or 68 banks apart: conflict-free b128 reads);
// the Q planes live in the U^T region (dead before U^T is written).  That is what bounds N: N <= 102.
// G3's tap products are not a phase of their own: their MFMAs are issued between the vector instructions that turn the G1 /
// G2 / hop accumulators into planes (see the kernel).  Values outside the f16 range are NOT clamped: the planes turn to
// inf / nan, the running maximum raises range_flag (magat_hip.h "range guard"), and the caller's predicated two-launch
// float32 form re-writes every output of the launch.
#include <type_traits>

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
  int B, N, P, ldx, ldy, concat, s_is_f64;       // (PACK: B = number of PACKS of four instances, Binst = instances)
  int Binst;
  int hsplit;                 // 1, or P (small batches): a workgroup per (instance, head) instead of per instance
  float* Ypre; int ldpre;     // head mean + hsplit: the heads' pre-activation rows [B*N][ldpre] (column 128 p); magat_gat_mean_launch merges them
  int* range_flag;
  const float* kconst;        // rank-1 score modes: per head a1 . wb + a2 . wb (GAT_modified; zeros for GAT_origin), or null
  int origin;                 // GAT_origin: the edge rule is |float(S) + I| > 1e-9 (graphML.py:1018)
  const float* x_scale;       // power-of-two activation scale of the X planes (device float; null or 0 = 1): Q, U and the
                              // accumulators then carry it once, the scores twice (undone in the softmax exponent and in Y)
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
#define GM_QW(ptr, val) do { const uint2 v_ = (val); GM_SINK(v_.x); GM_SINK(v_.y); } while (0)
#else
#define GM_QW(ptr, val) *reinterpret_cast<uint2*>(ptr) = (val)
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

// A lone wave issues in order: vector work only runs under the matrix pipe when MFMAs and vector instructions ALTERNATE in
// the instruction stream.  GM_PIN fences the scheduler: what is written between two fences stays between them.
#define GM_PIN() __builtin_amdgcn_sched_barrier(0)

// LDS hand-over barrier without the vmcnt(0) of __syncthreads(): weight fragments and Y stores stay in flight
#define GM_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)

__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one 16-wide k step of a row of MT tiles: the three split products, tile after tile (no two consecutive MFMAs on one
// accumulator when there is more than one tile)
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
__device__ __forceinline__ float mul1(float a, float b) {      // one v_mul_f32 the vectorizer cannot pair into a packed op
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split2v(float x, float y, unsigned& p1, unsigned& p2, float& vmax) {
  vmax = fmaxf(fmaxf(vmax, fabsf(x)), fabsf(y));
  split_pair(x, y, p1, p2);
}

// 4 x 4 transpose inside every quad of lanes: afterwards a[r] of quad lane l holds what was a[l] of quad lane r (two butterfly
// stages of one select + one DPP move + two selects per register pair)
__device__ __forceinline__ void quad_transpose4(float (&a)[4], int lane) {
  const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
  for (int r0 = 0; r0 < 4; r0 += 2) {
    const float send = odd ? a[r0] : a[r0 + 1];
    const float recv = dpp_mov<0xB1>(send);                // quad_perm [1,0,3,2]
    a[r0] = odd ? recv : a[r0];
    a[r0 + 1] = odd ? a[r0 + 1] : recv;
  }
#pragma unroll
  for (int r0 = 0; r0 < 2; ++r0) {
    const float send = hi ? a[r0] : a[r0 + 2];
    const float recv = dpp_mov<0x4E>(send);                // quad_perm [2,3,0,1]
    a[r0] = hi ? recv : a[r0];
    a[r0 + 2] = hi ? a[r0 + 2] : recv;
  }
}

// (pointer + compile-time constant: the constant lands in the offset field of the ds instruction)
__device__ __forceinline__ uint4 lds128(const char* ptr, int coff) { return *reinterpret_cast<const uint4*>(ptr + coff); }

// MT: 32-row tiles covering the agents (N <= 32 MT); KSI: 16-wide k steps covering them as a contraction index; KT: taps
// MODE 0: KeyQuery scores e_ij = x_i . W_p x_j (G1 + G2 products).  MODE 1: the rank-1 scores of GAT_modified / GAT_origin,
// e_ij = lrelu_0.2(c1_j + c2_i + k_p) with c1 = (a1 W_p) . x, c2 = (a2 W_p) . x (graphML.py:777-796, 1024-1037): the weight
// block of G1 holds the two vectors a1 W_p, a2 W_p as its rows 0 and 1 (magat_gat_pack_weights), so G1 and the Q planes are
// the KeyQuery code unchanged and hand c1 / c2 over as columns 0 / 1 of Q; G2's product is replaced by 16 LDS reads per lane.
// PACK (round 4; N <= 32): FOUR planning instances per pass, instance s in the 32-row slot s of the 128 rows (MT = 4, KSI = 2).
// The per-agent products (G1, G3: rows are agents, the contraction runs over features) take the four slots as their four row
// tiles - every weight fragment is streamed once per pack instead of once per instance, which is what bounded small graphs -
// and the graph products keep to the diagonal blocks: wave w scores instance w only (G2: one tile instead of MT x 4), the A
// planes hold [128 rows j][32 local columns i], and a hop of slot s contracts over that instance's two k steps.  An instance's
// arithmetic does not depend on its slot (same products, same k order, same intra-step positions as the unpacked N <= 32
// kernel): packed, unpacked and differently sharded batches agree bit for bit.
template <int MT, int KSI, int KT, bool CONCAT, int MODE, bool PACK = false>
__global__ __launch_bounds__(256, 1) void gat_mfma_kernel(const GatMfmaParams p) {
  extern __shared__ __align__(16) char lds[];
  static_assert(!PACK || (MT == 4 && KSI == 2), "PACK: four 32-row slots");
  constexpr int KI = 16 * KSI, SA = 2 * KI + 16;
  constexpr int KIU = PACK ? 128 : KI, SAU = 2 * KIU + 16;      // U^T planes [128 c][KIU columns i]; A planes [rows j][KI]
  constexpr int MTS = PACK ? 1 : MT;                            // row tiles of a wave's score tile column (PACK: its own slot)
  constexpr float kInvScale = 1.f / 256.f;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, h_ = lane >> 5;
  const int N = p.N;
  // LDS map.  X / Q planes: row r at 512 r, hi plane then lo plane (256 B each), 16-byte chunk c of a plane at
  // (c ^ (r & 15)) << 4.  A / U^T planes: rows SA bytes apart, lo plane behind the hi plane.
  const int NR = PACK ? 128 : N;                            // rows of the X planes
  const int R8 = PACK ? 128 : (N + 7) & ~7;                 // rows of the Q / A planes: whole groups of 8 are stored
  const unsigned AO = 512u * NR, AP = (unsigned)R8 * SA;    // A planes [R8][SA]
  const unsigned UO = AO + 2 * AP;                          // U^T planes [128][SAU]; the Q planes [R8][512] share the region
  constexpr unsigned UP = 128 * SAU;
  const unsigned MO = AO;      // edge masks [N][4], staged when there is no plan: read into registers behind the prologue
                               // barrier, dead before the first A plane is written (two barriers later)

  // per-lane fragment bases.  Row tiles of the agents (A operand of G1 / G2 / G3 and of the hops; rows past N re-read row
  // N - 1: finite values whose products land in rows nobody stores or in columns the zero entries of A annihilate)
  // Formed afresh from the (laundered) lane number wherever a phase needs them - a handful of vector instructions per phase
  // instead of 3 MT registers that live across the whole kernel (and were spilled):
  //   GM_ROWS(rsw, xsw): byte offset of row min(32 mt + lane % 32, N - 1) in the X / Q planes and its swizzle term
  //   GM_AROWS(rpa):     the same row in the A planes (+ 16 bytes for the upper lane half)
#define GM_ROWS(rsw, xsw) unsigned rsw[MT], xsw[MT]; { int l_ = lane; asm volatile("" : "+v"(l_)); _Pragma("unroll") \
    for (int mt_ = 0; mt_ < MT; ++mt_) { const int row_ = PACK ? 32 * mt_ + (l_ & 31) : min(32 * mt_ + (l_ & 31), N - 1); rsw[mt_] = row_ * 512; \
      xsw[mt_] = ((l_ >> 5) ^ (row_ & 15)) << 4; } }
#define GM_AROWS(rpa) unsigned rpa[MT]; { int l_ = lane; asm volatile("" : "+v"(l_)); _Pragma("unroll") \
    for (int mt_ = 0; mt_ < MT; ++mt_) rpa[mt_] = AO + (PACK ? 32 * mt_ + (l_ & 31) : min(32 * mt_ + (l_ & 31), N - 1)) * SA + (l_ >> 5) * 16; }
  const int cw_ = 32 * w + fr;                     // this lane's output column (g in G1, i in G2, c in G3 / hops)
  const char* wl = p.wfrag + (size_t)w * 16384 + lane * 16;   // this wave's 32-row tile of a weight block
  const float biasv = p.bias ? p.bias[cw_] : 0.f;
  const bool g2_active = PACK || 32 * w < KI;      // waves whose i tile holds columns of A (PACK: every wave scores its slot)
  float vmax = 0.f;                                // running maximum of |values written to f16 planes|
  float xs = 1.f;
  if (p.x_scale) xs = *p.x_scale;
  if (xs == 0.f) xs = 1.f;
  const float ixs = 1.f / xs;                      // (powers of two: exact)
  const float kOutScale = kInvScale * ixs;         // Y = acc_0 2^-8 / xs + bias
  const float kLog2eS = 1.4426950408889634f * ixs * ixs;      // scores carry xs^2

  // Weight fragments: ONE ring of four register pairs over the static stream of a head - W_p k steps 0..7, then the eight
  // fragments of tap K - 1, tap K - 2 (, tap 0) - continued into the next head (the stream is the same for every instance:
  // the last head prefetches head 0's).  A fragment is requested when the k step four ahead of it retires its slot: three
  // k steps (>= 1100 cycles) of slack for an L2 hit, 8 registers per fragment instead of a whole tap held a phase ahead.
  constexpr int FPH = 8 * (1 + KT);      // fragments per head (a multiple of the ring depth)
  // this workgroup's heads [hlo, hhi) and instances b0, b0 + bstride, ... (hsplit = P: one head, for batches that would
  // leave most of the chip idle - the closed-loop step of a single planning instance)
  const int hpw = p.P / p.hsplit, hlo = ((int)blockIdx.x % p.hsplit) * hpw, hhi = hlo + hpw;
  const int b0 = (int)blockIdx.x / p.hsplit, bstride = (int)gridDim.x / p.hsplit;
  uint4 wr[4][2];
  auto wr_load = [&](int hd_, int f) __attribute__((always_inline)) {      // f: static position in the stream of head hd_ (f >= FPH: the next head's)
    int hh = hd_;
    if (f >= FPH) { hh = hd_ + 1 == hhi ? hlo : hd_ + 1; f -= FPH; }
    const int ks = f & 7;
    const char* s = f < 8 ? wl + (size_t)hh * 65536
                          : wl + (size_t)(p.P + hh * KT + (KT - 1 - (f - 8) / 8)) * 65536;
    wr[f & 3][0] = *reinterpret_cast<const uint4*>(s + ks * 2048);
    wr[f & 3][1] = *reinterpret_cast<const uint4*>(s + ks * 2048 + 1024);
  };
#pragma unroll
  for (int f = 0; f < 4; ++f) wr_load(hlo, f);

  for (int b = b0; b < p.B; b += bstride) {
    // ---- instance prologue: X rows -> f16 planes; edge masks
    GM_STAMP(14);
    if (b != b0) GM_SYNC();      // (the previous instance's last hops still read the X planes: tap 0 runs under them)
    {
      // The X rows travel global -> LDS with LDS-direct loads (no destination registers: every request of the instance is in
      // flight at once; as register loads the compiler spilled each value as it arrived - the register file holds the whole
      // kernel's long-lived state here - and the loads ran one memory round trip at a time: 22 k cycles per instance), into
      // the A / U^T regions that nothing uses yet, 2 KB behind the masks; then float32 -> planes out of LDS.
      int t = threadIdx.x;
      asm volatile("" : "+v"(t));      // (per instance: the address arithmetic below is not hoisted into long-lived registers)
      const int lane_ = t & 63;
      // (PACK: b counts packs; instance 4 b + s fills rows 32 s .. 32 s + N - 1, rows that do not exist become zero planes)
      const float* Xb = p.X + (long long)b * (PACK ? 4 : 1) * N * p.ldx;
      const unsigned xst = AO + 2048;
      for (int c = w; c * 64 < NR * 32; c += 4) {          // 64 chunks of 16 bytes per wave instruction; 32 chunks per row
        const int ch = c * 64 + lane_;
        const int row = ch >> 5;
        const long long arow = PACK ? (long long)(row >> 5) * N + (row & 31) : row;      // agent row behind Xb
        const bool rok = PACK ? ((row & 31) < N && 4 * b + (row >> 5) < p.Binst) : ch < N * 32;
        const float* src = Xb + arow * p.ldx + (ch & 31) * 4;
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + xst + (unsigned)c * 1024u);
        if (rok) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
      }
      // (the masks are formed while the X rows are in flight: their first batch of GSO rows shares that round trip)
      if (!p.rmask_pre) {        // GSO rows -> 128-bit edge masks (one wave per row, ballot; |S| > 1e-9 as in gat_f32.hip)
        // rows w, w + 4, ...: eight rows' loads are issued before the first ballot (one memory latency per batch, not per row)
        auto stage_pack = [&](auto tag) __attribute__((always_inline)) {      // PACK: row R = 32 s + i of the pack; 32-bit masks over the instance's own columns
          typedef decltype(tag) ST;
          const int j0 = (lane_ & 31) < N ? (lane_ & 31) : N - 1;
          for (int R0 = w; R0 < 128; R0 += 32) {
            ST v0[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int R = R0 + 4 * r, sl = R >> 5, i = min(R & 31, N - 1);
              const int inst = min(4 * b + sl, p.Binst - 1);
              v0[r] = (static_cast<const ST*>(p.S) + (long long)inst * N * N)[(long long)i * N + j0];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int R = R0 + 4 * r, sl = R >> 5, i = R & 31;
              bool f0 = sizeof(ST) == 8 ? fabs((double)v0[r]) > 1e-9 : fabsf((float)v0[r]) > 1e-9f;
              if (MODE == 1 && p.origin) f0 = fabsf((float)v0[r] + ((lane_ & 31) == i ? 1.f : 0.f)) > 1e-9f;
              const unsigned long long k0 = __ballot(f0 && lane_ < N);
              if (lane_ == 0) {
                unsigned* m = reinterpret_cast<unsigned*>(lds + MO) + 4 * R;
                m[0] = (i < N && 4 * b + sl < p.Binst) ? (unsigned)k0 : 0u; m[1] = 0u; m[2] = 0u; m[3] = 0u;
              }
            }
          }
        };
        auto stage = [&](auto tag) {
          typedef decltype(tag) ST;
          const ST* Sp = static_cast<const ST*>(p.S) + (long long)b * N * N;
          const int j0 = lane_ < N ? lane_ : N - 1, j1 = lane_ + 64 < N ? lane_ + 64 : N - 1;
          for (int i0 = w; i0 < N; i0 += 32) {
            ST v0[8], v1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int i = i0 + 4 * r < N ? i0 + 4 * r : N - 1;
              v0[r] = Sp[(long long)i * N + j0];
              v1[r] = Sp[(long long)i * N + j1];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int i = i0 + 4 * r;
              bool f0 = sizeof(ST) == 8 ? fabs((double)v0[r]) > 1e-9 : fabsf((float)v0[r]) > 1e-9f;
              bool f1 = sizeof(ST) == 8 ? fabs((double)v1[r]) > 1e-9 : fabsf((float)v1[r]) > 1e-9f;
              if (MODE == 1 && p.origin) {      // self-loops added in float32: |float(S) + I| > 1e-9
                f0 = fabsf((float)v0[r] + (lane_ == i ? 1.f : 0.f)) > 1e-9f;
                f1 = fabsf((float)v1[r] + (lane_ + 64 == i ? 1.f : 0.f)) > 1e-9f;
              }
              const unsigned long long k0 = __ballot(f0 && lane_ < N), k1 = __ballot(f1 && lane_ + 64 < N);
              if (lane_ == 0 && i < N) {
                unsigned* m = reinterpret_cast<unsigned*>(lds + MO) + 4 * i;
                m[0] = (unsigned)k0; m[1] = (unsigned)(k0 >> 32); m[2] = (unsigned)k1; m[3] = (unsigned)(k1 >> 32);
              }
            }
          }
        };
        if constexpr (PACK) {
          if (p.s_is_f64) stage_pack(double{});
          else stage_pack(float{});
        } else {
          if (p.s_is_f64) stage(double{});
          else stage(float{});
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GM_SYNC();
      for (int idx = t; idx < NR * 16; idx += 256) {
        const int row = idx >> 4, ch = idx & 15;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + xst + idx * 32);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + xst + idx * 32 + 16);
        if (PACK && !((row & 31) < N && 4 * b + (row >> 5) < p.Binst)) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};      // (nothing was fetched)
        // the activation scale first (a power of two: exact), then the planes.  A NaN must raise the flag too - fmaxf drops it
        // from the running maximum; everything later is made of these planes and finite weights
        float xv[8] = {v0[0] * xs, v0[1] * xs, v0[2] * xs, v0[3] * xs, v1[0] * xs, v1[1] * xs, v1[2] * xs, v1[3] * xs};
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) bad |= !(fabsf(xv[e]) <= 65504.f);
        if (bad) vmax = __builtin_inff();
        uint4 hi, lo;
        split2v(xv[0], xv[1], hi.x, lo.x, vmax);
        split2v(xv[2], xv[3], hi.y, lo.y, vmax);
        split2v(xv[4], xv[5], hi.z, lo.z, vmax);
        split2v(xv[6], xv[7], hi.w, lo.w, vmax);
        char* dst = lds + (row * 512 + ((ch ^ (row & 15)) << 4));
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 256) = lo;
      }
    }
    GM_SYNC();
    GM_STAMP(15);
    // edge mask of this lane's row i = cw (bits j), shifted so that bit (8 (r / 4) + r % 4) is row 32 mt + ... of the tile
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    if constexpr (PACK) {      // row i = lane % 32 of instance 4 b + w; its mask covers the instance's own (<= 32) columns
      if (fr < N && 4 * b + w < p.Binst) {
        const unsigned m = p.rmask_pre ? p.rmask_pre[((long long)(4 * b + w) * N + fr) * 4]
                                       : *reinterpret_cast<const unsigned*>(lds + MO + 16 * cw_);
        mk[0] = m >> (4 * h_);
      }
    } else if (cw_ < N) {
      const uint4 m = p.rmask_pre ? *reinterpret_cast<const uint4*>(p.rmask_pre + ((long long)b * N + cw_) * 4)
                                  : *reinterpret_cast<const uint4*>(lds + MO + 16 * cw_);
      mk[0] = m.x >> (4 * h_); mk[1] = m.y >> (4 * h_); mk[2] = m.z >> (4 * h_); mk[3] = m.w >> (4 * h_);
    }
#pragma unroll 1
    for (int hd = hlo; hd < hhi; ++hd) {
      GM_STAMP(0);
      // lane-derived values are laundered per head: the per-element plane addresses below are recomputed where they are
      // used (a few VALU ops) instead of being hoisted out of the head loop into hundreds of long-lived registers
      int cw = cw_, h = h_;
      asm volatile("" : "+v"(cw), "+v"(h));
      // ---- G3 state.  U_k[i][c] for the K taps needs nothing but the X planes and the weights, so its MFMAs are issued
      // in slices BETWEEN the vector instructions that turn accumulators into planes (a lone wave per SIMD issues in order:
      // the conversions hide under the matrix pipe only when the two streams alternate).  One tap at a time (the
      // accumulator file holds 256 registers: all K taps next to the G1 / G2 tiles do not fit): tap K - 1 under the Q
      // planes, tap K - 2 under the softmax and the A planes, tap 0 (K = 3) under the U^T planes of the hops.
      f32x16 acc[KT][MT];      // (a tap's tiles are zeroed by its first slice: they occupy registers from there on only)
      uint4 a3[MT][2];
      GM_ROWS(rsw3, xsw3)
      constexpr int TAP_PER = 3 * MT, TAP_ALL = 8 * TAP_PER;
      // X fragments of a tap: ONE register set.  Within a k step the products run hi*hi, hi*lo, lo*hi over the row tiles, so
      // tile mt's hi plane is free after the second product and its lo plane after the third: each is re-loaded for the
      // next k step right behind its last MFMA (a whole product, >= 128 cycles, ahead of its next use).
      auto tap_load_a1 = [&](int ks, int mt, int pl) __attribute__((always_inline)) {
        const char* ap = lds + (rsw3[mt] + (xsw3[mt] ^ (unsigned)(ks << 5)));
        a3[mt][pl] = lds128(ap, 256 * pl);
      };
      // item i of tap k's static MFMA sequence (k step, product, row tile); i is a constant wherever this is called
      auto tap_one = [&](int k, int i) __attribute__((always_inline)) {
        const int ks = i / TAP_PER, r = i % TAP_PER;
        const int f = 8 + 8 * (KT - 1 - k) + ks;      // this k step's fragment in the weight stream
        if (i == 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            tap_load_a1(0, mt, 0);
            tap_load_a1(0, mt, 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[k][mt][e] = 0.f;
          }
        }
        const int q = r / MT, mt = r % MT;
        const int pa = q == 2 ? 1 : 0, pb = q == 1 ? 1 : 0;
        acc[k][mt] = mfma16(a3[mt][pa], wr[f & 3][pb], acc[k][mt]);
        if (ks + 1 < 8 && q >= 1) tap_load_a1(ks + 1, mt, q - 1);
        if (r == TAP_PER - 1) wr_load(hd, f + 4);      // (the k step's last MFMA: its slot takes the fragment four ahead)
      };
      auto tap_upto4 = [&](int k, int lo, int hi) __attribute__((always_inline)) {      // items lo .. hi - 1, hi - lo <= 4
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (lo + j < hi) tap_one(k, lo + j);
      };
      if (hd > hlo) GM_SYNC();      // the previous head's reads of the U^T / A planes are done
      GM_STAMP(1);
      // ---- G1: Q[j][g]
      {
        GM_ROWS(rsw, xsw)
        f32x16 accq[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) accq[0][mt][r] = 0.f;
        // (X fragments single-buffered like the taps': a plane is re-loaded for the next k step behind its last MFMA)
        uint4 a[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const char* ap = lds + (rsw[mt] + xsw[mt]);
          a[mt][0] = lds128(ap, 0);
          a[mt][1] = lds128(ap, 256);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              // (operands SWAPPED: the tile is Q^T - a lane holds agent row j = 32 mt + lane % 32 and FOUR CONSECUTIVE columns
              //  g per register quad, i.e. 8 contiguous bytes of a Q plane row: the planes are stored as they are, no transpose)
              accq[0][mt] = mfma16(wr[ks & 3][q == 1 ? 1 : 0], a[mt][q == 2 ? 1 : 0], accq[0][mt]);
              if (ks + 1 < 8 && q >= 1)
                a[mt][q - 1] = lds128(lds + (rsw[mt] + (xsw[mt] ^ (unsigned)((ks + 1) << 5))), 256 * (q - 1));
            }
          wr_load(hd, ks + 4);
          GM_PIN();
        }
        GM_STAMP(11);
        // Q planes [j][g] (16-byte chunks swizzled by row).  The G1 tile is Q^T: a lane holds row j = 32 mt + lane % 32 and, in
        // register quad q, columns g = 32 w + 8 q + 4 h + (0..3) - half of chunk 4 w + q of its row: ONE 8-byte store per plane
        // and quad (round 4; the untransposed tile needed eight 2-byte stores, 16 cycles each with four waves writing).  Rows
        // past the stored groups of 8 go to a dump row behind the planes (dead space of the U^T region until the hops).
        {
          const int fr_ = cw & 31;
          unsigned jb[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int j = 32 * mt + fr_;
            jb[mt] = UO + (unsigned)(j < R8 ? j : R8) * 512u + 8u * h;
          }
          unsigned qx[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            qx[q] = (unsigned)(((4 * w + q) ^ (fr_ & 15)) << 4);
            asm volatile("" : "+v"(qx[q]));      // (kept as 4 registers: not re-formed from its parts at every write)
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              // this quad's arithmetic alternating with the six MFMAs of tap K - 1 that fall to it
              constexpr int TPG = TAP_ALL / (MT * 4);      // = 6
              const int t0 = (mt * 4 + q) * TPG;
              unsigned hq[2], lq[2];
              tap_one(KT - 1, t0);
              // (the weights' 2^8 leaves here; single multiplies on purpose: v_pk_mul_f32 does not run under an MFMA of the same
              // wave - tools/exp/mfma_valu.hip)
              const float v01[2] = {mul1(accq[0][mt][4 * q], kInvScale), mul1(accq[0][mt][4 * q + 1], kInvScale)};
              GM_PIN();
              tap_one(KT - 1, t0 + 1);
              split2v(v01[0], v01[1], hq[0], lq[0], vmax);
              GM_PIN();
              tap_one(KT - 1, t0 + 2);
              const float v23[2] = {mul1(accq[0][mt][4 * q + 2], kInvScale), mul1(accq[0][mt][4 * q + 3], kInvScale)};
              GM_PIN();
              tap_one(KT - 1, t0 + 3);
              split2v(v23[0], v23[1], hq[1], lq[1], vmax);
              GM_PIN();
              tap_one(KT - 1, t0 + 4);
              char* o = lds + (jb[mt] + qx[q]);
              GM_PIN();
              tap_one(KT - 1, t0 + 5);
              GM_PIN();
              GM_QW(o, (uint2{hq[0], hq[1]}));
              GM_QW(o + 256, (uint2{lq[0], lq[1]}));
            }
        }
      }
      GM_STAMP(2);
      GM_SYNC();
      GM_STAMP(3);
      // rank-1 modes: columns 0 / 1 of Q are c1 / c2 (times the input scale): as float32 vectors behind the Q planes (the U^T
      // region holds 8 KB more than the planes need until the hops), rows past N read the last stored row (masked anyway)
      const unsigned CF = UO + (unsigned)R8 * 512u;
      if constexpr (MODE == 1) {
        for (int j = t; j < 128; j += 256) {
          const int row = min(j, R8 - 1);
          const char* qp = lds + UO + row * 512 + ((row & 15) << 4);
          const _Float16 h0 = *reinterpret_cast<const _Float16*>(qp), h1 = *reinterpret_cast<const _Float16*>(qp + 2);
          const _Float16 l0 = *reinterpret_cast<const _Float16*>(qp + 256), l1 = *reinterpret_cast<const _Float16*>(qp + 258);
          *reinterpret_cast<float*>(lds + CF + 4 * j) = (float)h0 + (float)l0;
          *reinterpret_cast<float*>(lds + CF + 512 + 4 * j) = (float)h1 + (float)l1;
        }
        GM_SYNC();
      }
      // ---- G2: E^T[j][i], softmax over j per column i, A planes [j][i] * 2^8
      if (g2_active) {
        // (rows j of the score tiles: every row tile of the instance - or, PACK, the one slot this wave scores: tile w)
        unsigned rsw[MTS], xsw[MTS];
        {
          int l_ = lane;
          asm volatile("" : "+v"(l_));
#pragma unroll
          for (int mt_ = 0; mt_ < MTS; ++mt_) {
            const int row_ = PACK ? 32 * w + (l_ & 31) : min(32 * mt_ + (l_ & 31), N - 1);
            rsw[mt_] = row_ * 512;
            xsw[mt_] = ((l_ >> 5) ^ (row_ & 15)) << 4;
          }
        }
        constexpr int TR = MT / MTS;      // MFMAs of tap K - 2 per interleave point (PACK: four times fewer points)
        auto tap_run = [&](int lo) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < TR; ++i) tap_one(KT - 2, lo + i);
        };
        f32x16 acce[1][MTS];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acce[0][mt][r] = 0.f;
        if constexpr (MODE == 1) {
          const float kc = p.kconst ? p.kconst[hd] : 0.f;
          const float c2i = *reinterpret_cast<const float*>(lds + CF + 512 + 4 * min(cw, 127)) * ixs + kc;
#pragma unroll
          for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 c1 = *reinterpret_cast<const f32x4*>(lds + CF + 4 * (32 * (PACK ? w : mt) + 8 * q + 4 * h));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float ev = __builtin_fmaf(c1[e], ixs, c2i);
                acce[0][mt][4 * q + e] = fmaxf(ev, 0.2f * ev);      // LeakyReLU(0.2)
              }
            }
        }
        if constexpr (MODE == 0) {
        const int rowb = PACK ? cw : min(32 * w + (cw & 31), N - 1);      // this wave's i tile as B operand
        const unsigned rswb = rowb * 512, xswb = (h ^ (rowb & 15)) << 4;
        const char* qbase = lds + UO;
        uint4 a[MTS][2], bq[2][2];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) {
          const char* ap = qbase + (rsw[mt] + xsw[mt]);
          a[mt][0] = lds128(ap, 0);
          a[mt][1] = lds128(ap, 256);
        }
        {
          const char* bp = lds + (rswb + xswb);
          bq[0][0] = lds128(bp, 0);
          bq[0][1] = lds128(bp, 256);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) {
            const char* bp = lds + (rswb + (xswb ^ (unsigned)((ks + 1) << 5)));
            bq[(ks + 1) & 1][0] = lds128(bp, 0);
            bq[(ks + 1) & 1][1] = lds128(bp, 256);
          }
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
              acce[0][mt] = mfma16(a[mt][q == 2 ? 1 : 0], bq[ks & 1][q == 1 ? 1 : 0], acce[0][mt]);
              if (ks + 1 < 8 && q >= 1)
                a[mt][q - 1] = lds128(qbase + (rsw[mt] + (xsw[mt] ^ (unsigned)((ks + 1) << 5))), 256 * (q - 1));
            }
          GM_PIN();
        }
        }      // (MODE == 0)
        GM_STAMP(12);
        // masked softmax of row i = cw over its edges j (graphML.py:1771-1776): in-lane over the MT * 16 accumulator
        // registers, one exchange with the partner lane (the other 4-row halves of the same column).  Entries without an
        // edge become -inf by a bit-field insert under the sign-extended mask bit; exp2(-inf) = 0 needs no select.
        // tap K - 2 under the softmax: TAP_ALL = 24 MT MFMAs = one per four entries of the two passes + four per row group
        float mx = -__builtin_inff();
        unsigned mkl[4] = {mk[0], mk[1], mk[2], mk[3]};      // (laundered: the per-entry masks are formed here, per head)
        asm volatile("" : "+v"(mkl[0]), "+v"(mkl[1]), "+v"(mkl[2]), "+v"(mkl[3]));
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {      // (one MFMA of tap K - 2 per four entries)
            tap_run(TR * (mt * 4 + q));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * q + e;
              const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)mkl[mt], 8 * (r >> 2) + (r & 3), 1);
              const float ev = acce[0][mt][r];      // (a scalar copy: bit-casting the vector element itself reads element 0)
              const unsigned u = (__builtin_bit_cast(unsigned, ev) & m) | (0xff800000u & ~m);
              const float em = __builtin_bit_cast(float, u);
              acce[0][mt][r] = em;
              mx = fmaxf(mx, em);
            }
            GM_PIN();
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float kLog2e = MODE == 1 ? 1.4426950408889634f : kLog2eS;      // (the rank-1 scores are formed at their true scale)
        const float cexp = mx > -__builtin_inff() ? -mx * kLog2e : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            tap_run(4 * MT + TR * (mt * 4 + q));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * q + e;
              acce[0][mt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(acce[0][mt][r], kLog2e, cexp));
              sum += acce[0][mt][r];
            }
            GM_PIN();
          }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = sum > 0.f ? 256.f / sum : 0.f;      // (A planes carry 2^8: the hops add to accumulators that hold 2^8 U_k)
        GM_STAMP(13);
        {
          // (lanes past the KI columns of A store into the 8 pad columns of the row: nothing reads them)
          // (PACK: rows 32 w + ... of the pack, local column i = lane % 32)
          char* abp = lds + (PACK ? AO + (32 * w + 4 * h) * SA + (cw & 31) * 2 : AO + 4 * h * SA + min(cw, KI + 7) * 2);
          char* alp = abp + AP;
#pragma unroll
          for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int t0 = 8 * MT + TR * (mt * 4 + q) * 4;      // four (PACK: sixteen) MFMAs of tap K - 2 per row group
              unsigned ha[2], la[2];
              tap_run(t0);
              // (single multiplies on purpose: v_pk_mul_f32 does not run under an MFMA of the same wave - tools/exp/mfma_valu.hip)
              const float v01[2] = {mul1(acce[0][mt][4 * q], inv), mul1(acce[0][mt][4 * q + 1], inv)};
              const float v23[2] = {mul1(acce[0][mt][4 * q + 2], inv), mul1(acce[0][mt][4 * q + 3], inv)};
              GM_PIN();
              tap_run(t0 + TR);
              split_pair(v01[0], v01[1], ha[0], la[0]);
              GM_PIN();
              tap_run(t0 + 2 * TR);
              split_pair(v23[0], v23[1], ha[1], la[1]);
              GM_PIN();
              tap_run(t0 + 3 * TR);
              const int jg = 32 * mt + 8 * q;
              const unsigned short hv[4] = {(unsigned short)ha[0], (unsigned short)(ha[0] >> 16), (unsigned short)ha[1],
                                            (unsigned short)(ha[1] >> 16)};
              const unsigned short lv[4] = {(unsigned short)la[0], (unsigned short)(la[0] >> 16), (unsigned short)la[1],
                                            (unsigned short)(la[1] >> 16)};
              GM_PIN();
              if (PACK || jg < N) {                                 // (wave-uniform; PACK: every row of the slot - rows past N are zeros)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  GM_AW(abp + (jg + e) * SA, hv[e]);
                  GM_AW(alp + (jg + e) * SA, lv[e]);
                }
              }
            }
        }
      } else {      // (a wave without columns of A: the second half of G3 on its own)
#pragma unroll
        for (int i = 0; i < TAP_ALL; ++i) tap_one(KT - 2, i);
      }
      GM_STAMP(4);
      GM_STAMP(5);
      GM_SYNC();      // Q is dead everywhere, the A planes are complete
      GM_STAMP(6);
      // ---- hops: acc_k += A^T U_{k+1}; the U^T rows of this wave's c tile are written and read by this wave only
      auto hop = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // U^T planes [c][i] <- acc_{k+1} 2^-8: a lane holds column c = cw, 4 consecutive i per register quad
        char* ub = lds + (UO + cw * SAU + h * 8);      // (+ 4 h rows of the quad: 8 bytes)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i0 = 32 * mt + 8 * q;      // + 4 h
            if (i0 < KIU) {                       // (KIU is a multiple of 16: both 4-row halves are inside or outside)
              // (K = 3: tap 0, half under each hop's plane conversion; the last hop's product needs it complete)
              constexpr int NG2 = KIU / 4;     // row groups of both hops
              const int gi = (k == 1 ? 0 : KIU / 8) + mt * 4 + q;
              const int u0 = TAP_ALL * gi / NG2, u1 = TAP_ALL * (gi + 1) / NG2, um = (u0 + u1) / 2;
              uint2 hi, lo;
              if constexpr (KT == 3) tap_upto4(0, u0, um);
              const float v01[2] = {mul1(acc[k + 1][mt][4 * q], kInvScale), mul1(acc[k + 1][mt][4 * q + 1], kInvScale)};
              split2v(v01[0], v01[1], hi.x, lo.x, vmax);
              GM_PIN();
              if constexpr (KT == 3) tap_upto4(0, um, u1);
              const float v23[2] = {mul1(acc[k + 1][mt][4 * q + 2], kInvScale), mul1(acc[k + 1][mt][4 * q + 3], kInvScale)};
              split2v(v23[0], v23[1], hi.y, lo.y, vmax);
              GM_PIN();
              *reinterpret_cast<uint2*>(ub + i0 * 2) = hi;
              *reinterpret_cast<uint2*>(ub + (i0 * 2 + UP)) = lo;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own rows: no barrier
        GM_AROWS(rpa)
        const char* up = lds + (UO + cw * SAU + h * 16);
        const char* apl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) apl[mt] = lds + rpa[mt];
        if constexpr (PACK) {
          // block-diagonal hop: slot mt contracts over its own instance - A rows 32 mt + .. (32 local columns) against columns
          // 32 mt .. 32 mt + 31 of the U^T rows: two k steps per slot, the products in the order of the unpacked kernel
          uint4 a[MT][2], bu[MT][2];
#pragma unroll
          for (int ks = 0; ks < KSI; ++ks) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              a[mt][0] = lds128(apl[mt], ks * 32);
              a[mt][1] = lds128(apl[mt] + AP, ks * 32);
              bu[mt][0] = lds128(up, 64 * mt + 32 * ks);
              bu[mt][1] = lds128(up, 64 * mt + 32 * ks + UP);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[k][mt] = mfma16(a[mt][q == 2 ? 1 : 0], bu[mt][q == 1 ? 1 : 0], acc[k][mt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
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
        }      // (!PACK)
        GM_STAMP(7 + (KT - 2 - k));
      };
      if constexpr (KT == 3) hop(std::integral_constant<int, 1>{});
      hop(std::integral_constant<int, 0>{});
      // ---- epilogue: a lane holds column c = cw of rows j = 32 mt + 8 q + 4 h + e (e = 0..3).  Stored as they are, that is 64
      // four-byte stores per lane and head (4.1 k cycles: the address path takes a store instruction at ~64 cycles whatever its
      // width).  A 4 x 4 transpose inside every quad of lanes turns (4 rows x 1 column) per lane into (1 row x 4 columns):
      // 16 sixteen-byte stores, each row still a 128-byte run per instruction.
      {
        const int fq = cw & 3;                                   // this lane's row of the 4 x 4 block after the transpose
        // (PACK: slot mt is instance 4 b + mt, its rows the N agents of that instance)
        const char* ybase = reinterpret_cast<const char*>(p.Y + (long long)b * (PACK ? 4 : 1) * N * p.ldy + (CONCAT ? hd * 128 : 0));
        const unsigned lbyte = (unsigned)((cw & ~3) + (4 * h + fq) * p.ldy) * 4u;
        const long long rowb = (long long)p.ldy * 4;
        const bool last = hd == p.P - 1;
        const float fp = (float)p.P;
        // concat: relu as one v_med3 (fmaxf semantics: a NaN gives 0).  Head mean: the running sum over the heads of
        // (Y_p + bias) lives in Y itself - every element is owned by one lane of one workgroup, so it is a plain
        // read-add-write of an L2-resident row (64 registers of running sums next to the taps' accumulators spilled) -
        // and the last head stores relu(sum / P)      (graphML.py:4663-4667; the same summation order as before)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int jg = PACK ? 8 * q : 32 * mt + 8 * q;        // first row of the quad group inside its instance
            if (jg >= N) break;                                   // (wave-uniform)
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[0][mt][4 * q + e], kOutScale, biasv);
            quad_transpose4(v, lane);
            const bool rok = jg + 4 * h + fq < N && (!PACK || 4 * b + mt < p.Binst);
            f32x4* dst = reinterpret_cast<f32x4*>(const_cast<char*>(ybase + ((PACK ? mt * N : 0) + jg) * rowb) + lbyte);
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if constexpr (CONCAT) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_fmed3f(o[e], 0.f, __builtin_inff());
            } else if (p.hsplit > 1) {
              // head split: this workgroup owns ONE head - its rows (+ bias, no ReLU) go to the scratch rows, column 128 hd;
              // the mean kernel sums them in head order (the same sum as the read-add-write below)
              dst = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.Ypre + (long long)b * N * p.ldpre + hd * 128) +
                                             (long long)jg * p.ldpre * 4 + (unsigned)((cw & ~3) + (4 * h + fq) * p.ldpre) * 4u);
            } else {
              if (hd > 0 && rok) o += *dst;
              if (last) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_fmed3f(o[e] / fp, 0.f, __builtin_inff());
              }
            }
#ifdef GM_WHATIF_NOYST
            GM_SINK(o[0]);
#else
            if (rok) *dst = o;
#endif
          }
      }
      GM_STAMP(10);
    }
  }
  if (p.range_flag && vmax > 65504.f) atomicOr(p.range_flag, 1);
}

size_t gat_mfma_lds(int N, int ksi) {
  const size_t sa = 2 * 16 * (size_t)ksi + 16, r8 = ((size_t)N + 7) & ~(size_t)7;
  return 2 * (size_t)N * 256 + 2 * r8 * sa + 2 * 128 * sa;
}
// PACK form: X planes [128], A planes [128][80], U^T planes [128][272]
constexpr size_t kGatPackLds = 2 * 128 * 256 + 2 * 128 * 80 + 2 * 128 * 272;

// shape classes: (MT, KSI) = (1, 2) N <= 32, (2, 4) N <= 64, (4, 7) N <= 102 (the LDS bound)
int gat_mfma_class(int N) {
  if (N <= 0) return -1;
  const int c = N <= 32 ? 0 : (N <= 64 ? 1 : 2);
  const int ksi = c == 0 ? 2 : (c == 1 ? 4 : 7);
  return (N <= 16 * ksi && gat_mfma_lds(N, ksi) <= 160 * 1024) ? c : -1;
}

// four instances per pass (N <= 32; see the kernel): one workgroup per pack, every head in it
template <int KT, int MODE>
int launch_pack(const GatMfmaParams& p, int slot, hipStream_t st) {
  const void* fn = p.concat ? reinterpret_cast<const void*>(&gat_mfma_kernel<4, 2, KT, true, MODE, true>)
                            : reinterpret_cast<const void*>(&gat_mfma_kernel<4, 2, KT, false, MODE, true>);
  if (magat_ensure_dyn_lds(fn, slot, kGatPackLds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  GatMfmaParams q = p;
  q.Binst = p.B;
  q.B = (p.B + 3) / 4;
  q.hsplit = 1;
  const int blocks = q.B < cus ? q.B : cus;
  magat_form_note(MAGAT_FORM_GAT_PACK);
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_LAYER, st);
  if (p.concat) hipLaunchKernelGGL((gat_mfma_kernel<4, 2, KT, true, MODE, true>), dim3(blocks), dim3(256), kGatPackLds, st, q);
  else hipLaunchKernelGGL((gat_mfma_kernel<4, 2, KT, false, MODE, true>), dim3(blocks), dim3(256), kGatPackLds, st, q);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

template <int MT, int KSI, int KT, int MODE>
int launch(const GatMfmaParams& p, int slot, hipStream_t st) {
  const size_t lds = gat_mfma_lds(p.N, KSI);
  const void* fn = p.concat ? reinterpret_cast<const void*>(&gat_mfma_kernel<MT, KSI, KT, true, MODE>)
                            : reinterpret_cast<const void*>(&gat_mfma_kernel<MT, KSI, KT, false, MODE>);
  if (magat_ensure_dyn_lds(fn, slot + (p.concat ? 0 : 6) + 12 * MODE, lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  // compute units of the CURRENT device (a cheap attribute query, no cached process-wide value: one process may drive several)
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  GatMfmaParams q = p;
  // A workgroup per (instance, head) pays the instance prologue once per head (12 k + 36 k cycles per unit against 12 k + 36 k P
  // per instance), but fills a chip that B instances alone leave idle: the cheaper of the two round counts (concat only: the
  // head mean accumulates in one workgroup's own rows of Y).
  const long long unsplit = (long long)((p.B + cus - 1) / cus) * (12 + 36 * p.P);
  const long long split = (long long)(((long long)p.B * p.P + cus - 1) / cus) * (12 + 36);
  q.hsplit = ((p.concat || p.Ypre) && p.P > 1 && split < unsplit) ? p.P : 1;      // (head mean: through the caller's scratch rows)
  const long long units = q.hsplit > 1 ? (long long)p.B * p.P : p.B;
  const int blocks = (int)(units < (long long)cus * q.hsplit ? units : (long long)cus * q.hsplit);
  if (q.hsplit > 1) magat_form_note(MAGAT_FORM_GAT_HSPLIT);
  if (units > blocks) magat_form_note(MAGAT_FORM_GAT_PERSIST);
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_LAYER, st);
  if (p.concat) hipLaunchKernelGGL((gat_mfma_kernel<MT, KSI, KT, true, MODE>), dim3(blocks), dim3(256), lds, st, q);
  else {
    hipLaunchKernelGGL((gat_mfma_kernel<MT, KSI, KT, false, MODE>), dim3(blocks), dim3(256), lds, st, q);
    // head split + head mean: relu(mean over the heads) of the scratch rows, summed in head order like the unsplit form's
    // read-add-write of Y (graphML.py:4663-4667)
    if (q.hsplit > 1 && magat_gat_mean_launch(p.Ypre, p.Y, (long long)p.B * p.N, p.P, 128, p.ldpre, p.ldy, st) != MAGAT_OK) {
      magat_prof_end(pid, st);
      return MAGAT_ERR_LAUNCH;
    }
  }
  magat_prof_end(pid, st);
  return magat_check_launch();
}

template <int MODE>
int launch_class(const GatMfmaParams& p, int K, hipStream_t st) {
  const int cls = gat_mfma_class(p.N);
  // N <= 32: four instances per pass once the batch fills the chip that way (option GAT_PACK, default 1; below that a workgroup
  // per instance - or per (instance, head) - has the shorter critical path: the closed-loop step of a single instance).  Packed
  // and unpacked forms produce the same bits for every instance.
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int pack = magat_opt(MAGAT_OPT_GAT_PACK);
  if (cls == 0 && pack && (pack >= 2 || (long long)p.B >= 2LL * cus)) {
    const int slot = MAGAT_LDS_GATP_0 + 4 * MODE + (K == 3 ? 0 : 2) + (p.concat ? 0 : 1);
    return K == 3 ? launch_pack<3, MODE>(p, slot, st) : launch_pack<2, MODE>(p, slot, st);
  }
  if (cls == 0) return K == 3 ? launch<1, 2, 3, MODE>(p, MAGAT_LDS_GATM_0, st) : launch<1, 2, 2, MODE>(p, MAGAT_LDS_GATM_0 + 1, st);
  if (cls == 1) return K == 3 ? launch<2, 4, 3, MODE>(p, MAGAT_LDS_GATM_0 + 2, st) : launch<2, 4, 2, MODE>(p, MAGAT_LDS_GATM_0 + 3, st);
  if (cls == 2) return K == 3 ? launch<4, 7, 3, MODE>(p, MAGAT_LDS_GATM_0 + 4, st) : launch<4, 7, 2, MODE>(p, MAGAT_LDS_GATM_0 + 5, st);
  return MAGAT_ERR_UNSUPPORTED;
}

}  // namespace

#ifdef MAGAT_DEBUG_HOOKS
extern "C" int magat_gat_mfma_set_debug_buffer(long long* dev_buf) { g_gat_mfma_dbg = dev_buf; return MAGAT_OK; }
#endif

int magat_gat_mfma_supported(int N, int G, int F, int K, int mode) {
  if (mode < MAGAT_MODE_KEYQUERY || mode > MAGAT_MODE_GAT_ORIGIN || G != 128 || F != 128 || (K != 2 && K != 3)) return 0;
  return gat_mfma_class(N) >= 0 ? 1 : 0;
}

int magat_gat_mfma_forward(const float* X, int ldx, const void* S, int s_is_f64, const unsigned* rmask_pre,
                           const float* packed_frag, const float* bias, float* Y, int ldy, int B, int N, int K, int P,
                           int concat, int* range_flag, hipStream_t st, const float* x_scale, int mode, const float* kconst,
                           float* ypre, int ldpre) {
  GatMfmaParams p;
  p.Ypre = (!concat && ypre && ldpre >= 128 * P) ? ypre : nullptr; p.ldpre = ldpre;
  p.x_scale = x_scale;
  p.kconst = kconst; p.origin = mode == MAGAT_MODE_GAT_ORIGIN ? 1 : 0;
  p.X = X; p.ldx = ldx; p.S = S; p.s_is_f64 = s_is_f64; p.rmask_pre = rmask_pre;
  p.wfrag = reinterpret_cast<const char*>(packed_frag);
  p.bias = bias; p.Y = Y; p.ldy = ldy; p.B = B; p.N = N; p.P = P; p.concat = concat; p.range_flag = range_flag;
  p.dbg = nullptr;
  p.hsplit = 1;
  p.Binst = B;
#ifdef MAGAT_DEBUG_HOOKS
  p.dbg = g_gat_mfma_dbg;
#endif
  return mode == MAGAT_MODE_KEYQUERY ? launch_class<0>(p, K, st) : launch_class<1>(p, K, st);
}
