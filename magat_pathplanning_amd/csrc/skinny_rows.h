// The skinny float32 layer (conv_gemm_f32.hip: a 1 x 1 product with at most 8 outputs - the action head, 640 -> 5 at c3) as
// device code two kernels share: skinny_gemm_kernel (a launch of its own) and gat_rerun_small_kernel (gat_f32.hip: the action head
// of few instances rides in the graph layer's predicated re-run launch - one launch less in the closed-loop step of one planning
// instance).  16 lanes own a row: every lane takes the 16-byte pieces l, l + 16, ... of the row (in, then in2), a batch of them in
// flight before the first is used, multiplies them with the CO weight rows out of LDS ([chunk][CO] float4s) in float32 FMAs, and
// the 16 partial sums meet in a DPP row reduction.  The summation order of a row is fixed (chunk order inside a lane, then the
// reduction tree) and does not depend on which workgroup or kernel computes it: bit-identical everywhere.
#pragma once
#include "magat_common.h"

struct MagatSkinnyParams {
  const float* in;
  const float* in2;
  const float* wt;     // [CO][Ktot]
  const float* bias;
  float* out;
  int M, Cin, C2, lda, lda2, ldc, relu;
  int bf16_rows;       // bit 0: in rows are bf16 (8-byte chunks of 4 values), bit 1: in2 rows
  int Cout;
};

// descriptor -> parameters when the layer is this form (MAGAT_ERR_UNSUPPORTED otherwise); conv_gemm_f32.hip
int magat_skinny_params(const magat_conv_gemm_desc* d, MagatSkinnyParams* out);

#ifdef SKINNY_BATCH_OVERRIDE
constexpr int MAGAT_SKINNY_BATCH = SKINNY_BATCH_OVERRIDE;
#else
constexpr int MAGAT_SKINNY_BATCH = 8;      // chunks per lane in flight
#endif

// weights -> LDS [Ktot / 4][CO][4]; every thread of the 256-thread workgroup; the caller synchronises afterwards
template <int CO>
__device__ __forceinline__ void magat_skinny_stage_weights(const MagatSkinnyParams& p, float* sk_w) {
  const int nq = (p.Cin >> 2) + (p.C2 >> 2), Ktot = 4 * nq;
  for (int idx = threadIdx.x; idx < nq * CO; idx += 256) {
    const int q = idx / CO, c = idx - q * CO;
    *reinterpret_cast<f32x4*>(sk_w + (size_t)idx * 4) = *reinterpret_cast<const f32x4*>(p.wt + (size_t)c * Ktot + 4 * q);
  }
}

// rows m0 + (thread / 16) for m0 = m_first, m_first + m_step, ... < m_end (m_step a multiple of 16; 256 threads)
template <int CO>
__device__ __forceinline__ void magat_skinny_rows(const MagatSkinnyParams& p, const float* sk_w, long long m_first, long long m_end,
                                                  long long m_step) {
  const int q1 = p.Cin >> 2;
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;      // 16 rows per workgroup step
  float bv[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) bv[c] = p.bias ? p.bias[c] : 0.f;
  // a lane's unit of work is a 16-byte PIECE of the row: one chunk of 4 float32 values, or two chunks of a bf16 row
  const bool b1 = p.bf16_rows & 1, b2 = (p.bf16_rows >> 1) & 1;
  const int np1 = b1 ? p.Cin >> 3 : q1, np2 = b2 ? p.C2 >> 3 : p.C2 >> 2, np = np1 + np2;
  for (long long m0 = m_first; m0 < m_end; m0 += m_step) {
    const long long m = m0 + grp;
    const bool ok = m < m_end;
    const long long mr = ok ? m : m_end - 1;
    const char* r1 = reinterpret_cast<const char*>(p.in) + mr * p.lda * (b1 ? 2 : 4);
    const char* r2 = reinterpret_cast<const char*>(p.in2) + mr * p.lda2 * (b2 ? 2 : 4) - 16LL * np1;   // (piece pc >= np1 at r2 + 16 pc)
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    auto fma4 = [&](const f32x4& xv, int q) {
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sk_w + ((size_t)q * CO + c) * 4);
        acc[c] = __builtin_fmaf(xv[3], w[3], __builtin_fmaf(xv[2], w[2], __builtin_fmaf(xv[1], w[1],
                 __builtin_fmaf(xv[0], w[0], acc[c]))));
      }
    };
    for (int pb = l16; pb < np; pb += 16 * MAGAT_SKINNY_BATCH) {
      uint4 x[MAGAT_SKINNY_BATCH];
#pragma unroll
      for (int j = 0; j < MAGAT_SKINNY_BATCH; ++j) {
        const int pq = pb + 16 * j;
        const int pc = pq < np ? pq : l16;      // (past the row: piece l16 again, dropped below)
        x[j] = *reinterpret_cast<const uint4*>((pc < np1 ? r1 : r2) + 16LL * pc);
      }
#pragma unroll
      for (int j = 0; j < MAGAT_SKINNY_BATCH; ++j) {
        const int pq = pb + 16 * j;
        if (pq < np) {
          const bool first = pq < np1;
          if (first ? b1 : b2) {      // eight bf16 values: chunks q, q + 1
            const int q = first ? 2 * pq : q1 + 2 * (pq - np1);
            fma4(f32x4{__builtin_bit_cast(float, x[j].x << 16), __builtin_bit_cast(float, x[j].x & 0xffff0000u),
                       __builtin_bit_cast(float, x[j].y << 16), __builtin_bit_cast(float, x[j].y & 0xffff0000u)}, q);
            fma4(f32x4{__builtin_bit_cast(float, x[j].z << 16), __builtin_bit_cast(float, x[j].z & 0xffff0000u),
                       __builtin_bit_cast(float, x[j].w << 16), __builtin_bit_cast(float, x[j].w & 0xffff0000u)}, q + 1);
          } else {
            fma4(__builtin_bit_cast(f32x4, x[j]), first ? pq : q1 + (pq - np1));
          }
        }
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float s = row16_sum(acc[c]) + bv[c];
      if (l16 == c) mine = s;
    }
    if (ok && l16 < CO) p.out[m * p.ldc + l16] = p.relu ? magat_relu(mine) : mine;
  }
}
