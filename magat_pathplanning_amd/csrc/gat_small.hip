// GraphFilterBatchAttentional.forward (KeyQuery attention) for SMALL graphs and NARROW features as one launch of matrix-core
// products: N <= 32 agents, G = F in {32, 64}, K = 2 | 3 - the published MAGAT settings (scripts/train_DMap.sh:42-46: 10 agents,
// bottleneckFeature 32, four heads; reference utils/graphUtils/graphML.py:4636-4671, 1724-1827, 1180-1286).  Same algebra and the
// same f16x3 arithmetic as gat_mfma.hip (two f16 planes per operand, three v_mfma_f32_32x32x16_f16 per product, fp32
// accumulation; per instance b and head p):
//   G1  Q[j][g]   = sum_f X[j][f] W_p[g][f]           (operands swapped: the tile is Q^T, the planes are stored as they are)
//   G2  E^T[j][i] = sum_g Q[j][g] X[i][g]; masked softmax over j in the accumulator layout (a lane owns column i) -> A planes
//   G3  U_k[i][c] = sum_f X[i][f] H_pk[c][f],  k = 0..K-1
//   hops acc_k[j][c] += sum_i A[j][i] U^T[c][i],  k = K-2 .. 0  (Horner);  Y_p = relu(acc_0 2^-8 + bias) | head mean
// but organised the other way round: a graph of <= 32 agents is ONE 32-row tile, so A WAVE OWNS A PLANNING INSTANCE - its X, Q, A
// and U^T planes are wave-private LDS (20-34 KB per wave), no workgroup barrier exists in the kernel, and the head mean is summed
// in the wave's registers.  Weights are read straight from the row-major f16 planes of the layer's pack (magat_gat_pack_weights:
// [2][NC][G] halves of 2^8 Bt: a lane's MFMA row operand is 16 contiguous bytes of a weight row; 16 KB per head at G = 32, L2 /
// L1 hits).  Values beyond the f16 range raise range_flag and the caller's predicated two-launch float32 form rewrites the output.
#include "magat_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct GatSmallParams {
  const float* X;             // [B*N][ldx]
  const void* S;              // [B][N][N] f32 | f64
  const unsigned* rmask_pre;  // [B][N][4] edge masks of a GSO plan (word 0 used), or null
  const unsigned short* Hs;   // f16 planes [2][NC][G] of 2^8 Bt (rows: [P][G] W_p, then [P][K][F] H_pk)
  const float* bias;          // [F] or null
  float* Y;                   // [B*N][ldy]
  int B, N, P, NC, ldx, ldy, s_is_f64;
  int* range_flag;
  const float* x_scale;
};

__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2) {
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(p1), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(p1), "v"(y));
  const f16x2 r = __builtin_convertvector(f32x2{rx, ry}, f16x2);
  p2 = __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split2v(float x, float y, unsigned& p1, unsigned& p2, float& vmax) {
  vmax = fmaxf(fmaxf(vmax, fabsf(x)), fabsf(y));
  split_pair(x, y, p1, p2);
}

template <int F, int KT, bool CONCAT>
__global__ __launch_bounds__(256, F == 32 ? 2 : 1) void gat_small_kernel(const GatSmallParams p) {      // (32 features: two waves per SIMD - a second workgroup per CU at large batches)
  extern __shared__ __align__(16) char lds_all[];
  constexpr int CT = F / 32, KF = F / 16;
  constexpr int RS = 2 * F + 16;             // row stride of the X / Q planes (bytes): rows land 20 / 36 banks apart
  constexpr int SA = 80;                     // row stride of the A / U^T planes: 32 columns + 16 bytes
  constexpr int XO = 0, QO = 64 * RS, AO = 128 * RS, UO = AO + 64 * SA, WLDS = UO + 2 * F * SA;
  constexpr float kInvScale = 1.f / 256.f;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  char* const lds = lds_all + w * WLDS;      // everything below is private to this wave
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

  for (int inst = (int)blockIdx.x * 4 + w; inst < p.B; inst += (int)gridDim.x * 4) {
    // ---- X rows -> f16 planes (rows past N: zeros)
    {
      const float* Xb = p.X + (long long)inst * N * p.ldx;
      for (int idx = lane; idx < 32 * (F / 8); idx += 64) {
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
        *reinterpret_cast<uint4*>(dst + 32 * RS) = lo;
      }
    }
    // ---- edge mask of this lane's row i = lane % 32 (bits j), |S| > 1e-9 (graphML.py:1274-1276; NaN entries are no edges)
    unsigned mk = 0u;
    if (p.rmask_pre) {
      if (fr < N) mk = p.rmask_pre[((long long)inst * N + fr) * 4];
    } else {
      // lane (fr, fh): row i = fr, columns j = 16 fh .. 16 fh + 15 of it; the two halves are OR-ed
      // (the sixteen entries are requested together - clamped addresses, the predicate on the loaded value - and pinned in front
      //  of the tests: as a loop with a run-time bound they were sixteen round trips, one after the other)
      unsigned bits = 0u;
      auto row_bits = [&](auto tag) __attribute__((always_inline)) {
        typedef decltype(tag) ST;
        const ST* Sp = static_cast<const ST*>(p.S) + ((long long)inst * N + (fr < N ? fr : N - 1)) * N;
        ST sv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) sv[u] = Sp[16 * fh + u < N ? 16 * fh + u : N - 1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const bool e_ = (sv[u] < (ST)0 ? -sv[u] : sv[u]) > (ST)1e-9 && fr < N && 16 * fh + u < N;      // (NaN: no edge)
          bits |= (e_ ? 1u : 0u) << (16 * fh + u);
        }
      };
      if (p.s_is_f64) row_bits(double{});
      else row_bits(float{});
      mk = bits | (unsigned)__shfl_xor((int)bits, 32, 64);
    }
    mk >>= 4 * fh;      // bit (8 (r / 4) + r % 4) = row j of accumulator register r

    float ysum[CT][16];
    if constexpr (!CONCAT) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ysum[ct][r] = 0.f;
    }
#pragma unroll 1
    for (int hd = 0; hd < p.P; ++hd) {
      // this lane's 16-byte pieces of weight row (base + lane % 32): k step ks, plane pl at + pl * plane + 16 ks + 8 fh halves
      auto wfrag = [&](long long row0, int ks, int pl) __attribute__((always_inline)) {
        return *reinterpret_cast<const uint4*>(p.Hs + pl * plane + (row0 + fr) * G + 16 * ks + 8 * fh);
      };
      auto xfrag = [&](int ks, int pl) __attribute__((always_inline)) {
        return *reinterpret_cast<const uint4*>(lds + XO + pl * 32 * RS + fr * RS + (16 * ks + 8 * fh) * 2);
      };
      // ---- G1 (operands swapped): Q^T tile, lane = agent row j, register quads = 4 consecutive columns g -> Q planes [j][g]
      // (a product's weight fragments are requested together and pinned in front of its matrix instructions: the compiler moves
      //  every request next to its use otherwise - one exposed L2 round trip per k step at one or two waves per SIMD)
      uint4 wq[CT][KF][2];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
          wq[ct][ks][0] = wfrag((long long)hd * G + 32 * ct, ks, 0);
          wq[ct][ks][1] = wfrag((long long)hd * G + 32 * ct, ks, 1);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
          const uint4 w0 = wq[ct][ks][0], w1 = wq[ct][ks][1], x0 = xfrag(ks, 0), x1 = xfrag(ks, 1);
          acc = mfma16(w0, x0, acc);
          acc = mfma16(w1, x0, acc);
          acc = mfma16(w0, x1, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 hi, lo;
          split2v(acc[4 * q] * kInvScale, acc[4 * q + 1] * kInvScale, hi.x, lo.x, vmax);
          split2v(acc[4 * q + 2] * kInvScale, acc[4 * q + 3] * kInvScale, hi.y, lo.y, vmax);
          char* o = lds + QO + fr * RS + (32 * ct + 8 * q + 4 * fh) * 2;
          *reinterpret_cast<uint2*>(o) = hi;
          *reinterpret_cast<uint2*>(o + 32 * RS) = lo;
        }
      }
      // 32 features: the K taps' fragments (4 K registers quads) are requested here, in flight under G2 and the softmax
      constexpr bool WEARLY = false;      // (requesting the taps in front of G2 cost the second wave per SIMD: 256 + registers)
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
      // ---- G2: E^T[j][i] = sum_g Q[j][g] X[i][g]; lane = column i, registers = rows j
      f32x16 e;
#pragma unroll
      for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KF; ++ks) {
        const char* qp = lds + QO + fr * RS + (16 * ks + 8 * fh) * 2;
        const uint4 q0 = *reinterpret_cast<const uint4*>(qp), q1 = *reinterpret_cast<const uint4*>(qp + 32 * RS);
        const uint4 x0 = xfrag(ks, 0), x1 = xfrag(ks, 1);
        e = mfma16(q0, x0, e);
        e = mfma16(q0, x1, e);
        e = mfma16(q1, x0, e);
      }
      // masked softmax of row i over its edges j (in-lane + the partner lane), A planes [j][i] * 2^8
      float mx = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)mk, 8 * (r >> 2) + (r & 3), 1);
        const float ev = e[r];      // (a scalar copy: bit-casting the vector element itself reads element 0)
        const float em = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ev) & m) | (0xff800000u & ~m));
        e[r] = em;
        mx = fmaxf(mx, em);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float cexp = mx > -__builtin_inff() ? -mx * kLog2e : 0.f;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[r], kLog2e, cexp));
        sum += e[r];
      }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = sum > 0.f ? 256.f / sum : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned ha[2], la[2];
        split_pair(e[4 * q] * inv, e[4 * q + 1] * inv, ha[0], la[0]);
        split_pair(e[4 * q + 2] * inv, e[4 * q + 3] * inv, ha[1], la[1]);
        char* o = lds + AO + (8 * q + 4 * fh) * SA + fr * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<unsigned short*>(o + c * SA) = (unsigned short)(ha[c >> 1] >> (16 * (c & 1)));
          *reinterpret_cast<unsigned short*>(o + c * SA + 32 * SA) = (unsigned short)(la[c >> 1] >> (16 * (c & 1)));
        }
      }
      // ---- G3: U_k[i][c] for the K taps (lane = column c, registers = rows i)
      f32x16 acc[KT][CT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        uint4 wk[WEARLY ? 1 : CT][KF][2];      // (wider layers: a tap's fragments together, in front of its products)
        if constexpr (!WEARLY) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
              const long long row0 = (long long)p.P * G + ((long long)hd * KT + k) * F + 32 * ct;
              wk[ct][ks][0] = wfrag(row0, ks, 0);
              wk[ct][ks][1] = wfrag(row0, ks, 1);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[k][ct][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KF; ++ks) {
            uint4 w0, w1;
            if constexpr (WEARLY) { w0 = wt[k][ct][ks][0]; w1 = wt[k][ct][ks][1]; }
            else { w0 = wk[ct][ks][0]; w1 = wk[ct][ks][1]; }
            const uint4 x0 = xfrag(ks, 0), x1 = xfrag(ks, 1);
            acc[k][ct] = mfma16(x0, w0, acc[k][ct]);
            acc[k][ct] = mfma16(x0, w1, acc[k][ct]);
            acc[k][ct] = mfma16(x1, w0, acc[k][ct]);
          }
        }
      }
      // ---- hops (Horner): acc_k += A U_{k+1}; the U^T planes [c][i] are rewritten from acc_{k+1}
#pragma unroll
      for (int k = KT - 2; k >= 0; --k) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 hi, lo;
            split2v(acc[k + 1][ct][4 * q] * kInvScale, acc[k + 1][ct][4 * q + 1] * kInvScale, hi.x, lo.x, vmax);
            split2v(acc[k + 1][ct][4 * q + 2] * kInvScale, acc[k + 1][ct][4 * q + 3] * kInvScale, hi.y, lo.y, vmax);
            char* o = lds + UO + (32 * ct + fr) * SA + (8 * q + 4 * fh) * 2;
            *reinterpret_cast<uint2*>(o) = hi;
            *reinterpret_cast<uint2*>(o + F * SA) = lo;
          }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const char* ap = lds + AO + fr * SA + (16 * ks + 8 * fh) * 2;
          const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + 32 * SA);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const char* up = lds + UO + (32 * ct + fr) * SA + (16 * ks + 8 * fh) * 2;
            const uint4 u0 = *reinterpret_cast<const uint4*>(up), u1 = *reinterpret_cast<const uint4*>(up + F * SA);
            acc[k][ct] = mfma16(a0, u0, acc[k][ct]);
            acc[k][ct] = mfma16(a0, u1, acc[k][ct]);
            acc[k][ct] = mfma16(a1, u0, acc[k][ct]);
          }
        }
      }
      // ---- epilogue: lane = column c, registers = rows j
      float* yb = p.Y + (long long)inst * N * p.ldy;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 8 * (r >> 2) + 4 * fh + (r & 3);
          const float v = __builtin_fmaf(acc[0][ct][r], kOutScale, biasv[ct]);
          if constexpr (CONCAT) {
            if (j < N) yb[(long long)j * p.ldy + hd * F + 32 * ct + fr] = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());
          } else {
            ysum[ct][r] += v;      // (graphML.py:4663-4667: mean over the heads, then ReLU)
            if (hd == p.P - 1 && j < N)
              yb[(long long)j * p.ldy + 32 * ct + fr] = __builtin_amdgcn_fmed3f(ysum[ct][r] / (float)p.P, 0.f, __builtin_inff());
          }
        }
    }
  }
  if (p.range_flag && vmax > 65504.f) atomicOr(p.range_flag, 1);
}

template <int F, int KT>
int launch_small(const GatSmallParams& p, int concat, int slot, hipStream_t st) {
  constexpr size_t wlds = 128 * (2 * F + 16) + 64 * 80 + 2 * F * 80;
  const size_t lds = 4 * wlds;
  const void* fn = concat ? reinterpret_cast<const void*>(&gat_small_kernel<F, KT, true>)
                          : reinterpret_cast<const void*>(&gat_small_kernel<F, KT, false>);
  if (magat_ensure_dyn_lds(fn, slot + (concat ? 0 : 1), lds) != MAGAT_OK) return MAGAT_ERR_LAUNCH;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  // a wave per instance, four instances per workgroup
  long long blocks = ((long long)p.B + 3) / 4;
  const long long cap = (long long)cus * (lds <= 80 * 1024 ? 2 : 1);
  if (blocks > cap) blocks = cap;
  const int pid = magat_prof_begin(MAGAT_TAG_GAT_LAYER, st);
  if (concat) hipLaunchKernelGGL((gat_small_kernel<F, KT, true>), dim3((unsigned)blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gat_small_kernel<F, KT, false>), dim3((unsigned)blocks), dim3(256), lds, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

}  // namespace

int magat_gat_small_supported(int N, int G, int F, int K, int mode) {
  return mode == MAGAT_MODE_KEYQUERY && N >= 1 && N <= 32 && G == F && (G == 32 || G == 64) && (K == 2 || K == 3);
}

// Hs: the f16 planes [2][NC][G] of the layer's pack (packed + magat_gat_f16_block_offset(NC, G))
int magat_gat_small_forward(const float* X, int ldx, const void* S, int s_is_f64, const unsigned* rmask_pre, const float* Hs,
                            int NC, const float* bias, float* Y, int ldy, int B, int N, int G, int K, int P, int concat,
                            int* range_flag, hipStream_t st, const float* x_scale) {
  if (!magat_gat_small_supported(N, G, G, K, MAGAT_MODE_KEYQUERY)) return MAGAT_ERR_UNSUPPORTED;
  if ((ldx & 3) || (reinterpret_cast<uintptr_t>(X) & 15)) return MAGAT_ERR_UNSUPPORTED;
  GatSmallParams p;
  p.X = X; p.S = S; p.rmask_pre = rmask_pre; p.Hs = reinterpret_cast<const unsigned short*>(Hs); p.bias = bias; p.Y = Y;
  p.B = B; p.N = N; p.P = P; p.NC = NC; p.ldx = ldx; p.ldy = ldy; p.s_is_f64 = s_is_f64;
  p.range_flag = range_flag; p.x_scale = x_scale;
  if (G == 32) return K == 3 ? launch_small<32, 3>(p, concat, MAGAT_LDS_GATS_0, st) : launch_small<32, 2>(p, concat, MAGAT_LDS_GATS_0 + 2, st);
  return K == 3 ? launch_small<64, 3>(p, concat, MAGAT_LDS_GATS_0 + 4, st) : launch_small<64, 2>(p, concat, MAGAT_LDS_GATS_0 + 6, st);
}
