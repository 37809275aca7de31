// Training of the per-agent CNN on the HIP path: the weight gradient of a convolution over pixel-major activations.
//
// The reference trains the whole module through autograd (agents/decentralplannerlocal_OnlineExpert_GAT.py:556-567); the
// convolutions of resnet_pytorch.py:40-73, 427-524 are 96 % of a training step's FLOPs.  Forward (training mode) and the
// input gradient are the float32 implicit-GEMM kernel of conv_gemm_f32.hip (magat_conv_gemm_f32: the input gradient of a
// stride-1 convolution IS a convolution of dY with the taps mirrored and the channel roles swapped; a strided one is the same
// over the zero-stuffed dY) - this file adds the one product that kernel cannot express, the WEIGHT gradient
//     dW[co][tap][ci] = sum over output pixels o and agents m of  dY[o][m][co] * X[in(o, tap)][m][ci],
// a GEMM whose contraction runs over (pixel, agent) pairs - tens of thousands of rows - and whose result is at most
// 128 x 1152.  One wave per (output-channel tile, input-channel tile, tap, agent chunk): v_mfma_f32_32x32x2_f32 with the
// two contraction rows of an instruction = two agents of one pixel (lanes 0-31 / 32-63 read 128 contiguous bytes of one
// agent's channels each: no transposes, no LDS), 2 x 2 register tiles of 32 x 32 where the channel counts allow it.  The agent
// chunks write partial sums; the caller adds them up in a fixed order (deterministic gradients, no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/magat_hip.h"
#include "magat_common.h"

namespace {

struct WgradParams {
  const float* x;
  const float* dy;
  float* part;
  long long x_pix_stride, dy_pix_stride;
  int M, Cin, lda, Cout, ldc;
  int Hin, Win, Hout, Wout, kH, kW, stride, pad;
  int cin_out;                 // channels of the weight tensor itself (<= Cin: the stem's fourth channel is padding)
  int chunks, mc;              // chunks = pixel groups x agent ranges; agents per range (even)
  int cm, pc;                  // agent ranges; output pixels per pixel group
  int tiles_ci;                // input-channel tiles of 32 * NCI
};

template <int NCO, int NCI>
__global__ __launch_bounds__(64) void wgrad_kernel(const WgradParams p) {
  const int lane = threadIdx.x;
  const int col = lane & 31, half = lane >> 5;
  int b = blockIdx.x;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int taps = p.kH * p.kW;
  const int tap = b % taps; b /= taps;
  const int chunk = b % p.chunks;
  const int tco = b / p.chunks;
  const int co0 = tco * 32 * NCO, ci0 = tci * 32 * NCI;
  const int ty = tap / p.kW, tx = tap - ty * p.kW;
  const int m0 = (chunk % p.cm) * p.mc;
  const int m1 = m0 + p.mc < p.M ? m0 + p.mc : p.M;
  const int o0 = (chunk / p.cm) * p.pc;
  const int npix = p.Hout * p.Wout;
  const int o1 = o0 + p.pc < npix ? o0 + p.pc : npix;
  f32x16 acc[NCO][NCI];
#pragma unroll
  for (int i = 0; i < NCO; ++i)
#pragma unroll
    for (int j = 0; j < NCI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bool cok[NCI];
#pragma unroll
  for (int j = 0; j < NCI; ++j) cok[j] = ci0 + 32 * j + col < p.Cin;        // (the stem's 3 -> 4 channels: a partial tile)
  for (int o = o0; o < o1; ++o) {
    const int oy = o / p.Wout, ox = o - oy * p.Wout;
    const int iy = oy * p.stride - p.pad + ty;
    if (iy < 0 || iy >= p.Hin) continue;
    {
      const int ix = ox * p.stride - p.pad + tx;
      if (ix < 0 || ix >= p.Win) continue;
      const float* dyp = p.dy + (long long)(oy * p.Wout + ox) * p.dy_pix_stride + co0 + col;
      const float* xp = p.x + (long long)(iy * p.Win + ix) * p.x_pix_stride + ci0 + col;
      // four row pairs per step: their 4 * (NCO + NCI) loads are in flight together (a contraction row pair = two agents)
      constexpr int U = 4;
      for (int mm = m0; mm < m1; mm += 2 * U) {
        float a[U][NCO], bb[U][NCI];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = mm + 2 * u + half;
          const bool ok = m < m1;
#pragma unroll
          for (int i = 0; i < NCO; ++i) a[u][i] = ok ? dyp[(long long)m * p.ldc + 32 * i] : 0.f;
#pragma unroll
          for (int j = 0; j < NCI; ++j) bb[u][j] = ok && cok[j] ? xp[(long long)m * p.lda + 32 * j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < NCO; ++i)
#pragma unroll
            for (int j = 0; j < NCI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], bb[u][j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // part[chunk][co][ci][tap]: torch's (Cout, Cin, kH, kW) layout per chunk - the caller's sum over the chunks IS the gradient
  // tensor (the stores are 4 * taps bytes apart: a few KB per wave, once)
  float* out = p.part + (long long)chunk * p.Cout * taps * p.cin_out;
#pragma unroll
  for (int i = 0; i < NCO; ++i)
#pragma unroll
    for (int j = 0; j < NCI; ++j) {
      const int ci = ci0 + 32 * j + col;
      if (ci >= p.cin_out) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
        out[((long long)co * p.cin_out + ci) * taps + tap] = acc[i][j][r];
      }
    }
}

}  // namespace

// chunks of the contraction: agent ranges of >= 64 agents x groups of output pixels, about four waves per SIMD of the chip
static void wgrad_chunks(int M, int Cin, int Cout, int kH, int kW, int npix, int* cm, int* mc, int* cpix, int* pc) {
  const int nco = Cout % 64 == 0 ? 2 : 1, nci = Cin % 64 == 0 ? 2 : 1;
  const long long waves = (long long)(Cout / (32 * nco)) * ((Cin + 32 * nci - 1) / (32 * nci)) * kH * kW;
  long long want = (4096 + waves - 1) / waves;
  if (want < 1) want = 1;
  long long a = (M + 63) / 64;                               // agent ranges
  if (a > want) a = want;
  int len = (int)(((M + a - 1) / a + 1) & ~1LL);             // agents per range, even
  a = (M + len - 1) / len;
  long long g = (want + a - 1) / a;                          // pixel groups
  if (g > npix) g = npix;
  if (g < 1) g = 1;
  int per = (int)((npix + g - 1) / g);
  g = (npix + per - 1) / per;
  *cm = (int)a; *mc = len; *cpix = (int)g; *pc = per;
}

size_t magat_conv_wgrad_workspace_floats(int M, int Cin, int cin_w, int Cout, int kH, int kW, int npix) {
  if (M <= 0 || Cin <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || npix <= 0 || cin_w <= 0) return 0;
  int cm, mc, cpix, pc;
  wgrad_chunks(M, Cin, Cout, kH, kW, npix, &cm, &mc, &cpix, &pc);
  return (size_t)cm * cpix * Cout * kH * kW * cin_w;
}

// x: [Hin*Win][M][lda >= Cin] float32 pixel-major (pixel stride x_pix_stride floats); dy: [Hout*Wout][M][ldc >= Cout];
// part: magat_conv_wgrad_workspace_floats(...) floats = [chunks][Cout][cin_w][kH][kW] partial sums in torch's weight layout
// (cin_w <= Cin: the channels the weight tensor has; the caller sums over the first axis, *chunks_out tells how many).
// Cout a multiple of 32.
int magat_conv_wgrad_f32(const float* x, long long x_pix_stride, int lda, const float* dy, long long dy_pix_stride, int ldc,
                         float* part, int* chunks_out, int M, int Cin, int cin_w, int Cout, int Hin, int Win, int Hout, int Wout,
                         int kH, int kW, int stride, int pad, void* stream) {
  if (!x || !dy || !part || !chunks_out) return MAGAT_ERR_NULL;
  if (M <= 0 || Cin <= 0 || Cout <= 0 || Cout % 32 || lda < Cin || ldc < Cout || kH <= 0 || kW <= 0 || stride <= 0 || pad < 0 ||
      Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || cin_w <= 0 || cin_w > Cin)
    return MAGAT_ERR_BAD_SHAPE;
  const int nco = Cout % 64 == 0 ? 2 : 1, nci = Cin % 64 == 0 ? 2 : 1;
  WgradParams p;
  p.x = x; p.dy = dy; p.part = part;
  p.x_pix_stride = x_pix_stride; p.dy_pix_stride = dy_pix_stride;
  p.M = M; p.Cin = Cin; p.lda = lda; p.Cout = Cout; p.ldc = ldc; p.cin_out = cin_w;
  p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.kH = kH; p.kW = kW; p.stride = stride; p.pad = pad;
  p.tiles_ci = (Cin + 32 * nci - 1) / (32 * nci);
  const int tiles_co = Cout / (32 * nco);
  int cpix;
  wgrad_chunks(M, Cin, Cout, kH, kW, Hout * Wout, &p.cm, &p.mc, &cpix, &p.pc);
  p.chunks = p.cm * cpix;
  *chunks_out = p.chunks;
  const unsigned grid = (unsigned)((long long)tiles_co * p.chunks * kH * kW * p.tiles_ci);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pid = magat_prof_begin(MAGAT_TAG_CONV_WGRAD, st);
  if (nco == 2 && nci == 2) hipLaunchKernelGGL((wgrad_kernel<2, 2>), dim3(grid), dim3(64), 0, st, p);
  else if (nco == 2) hipLaunchKernelGGL((wgrad_kernel<2, 1>), dim3(grid), dim3(64), 0, st, p);
  else if (nci == 2) hipLaunchKernelGGL((wgrad_kernel<1, 2>), dim3(grid), dim3(64), 0, st, p);
  else hipLaunchKernelGGL((wgrad_kernel<1, 1>), dim3(grid), dim3(64), 0, st, p);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------------
// BatchNorm in TRAINING mode over pixel-major rows [rows = pixels * agents][C] (nn.BatchNorm2d of resnet_pytorch.py:42-58 with
// batch statistics): torch's own kernels for a 2-D (rows, C) input cost 35 us per reduction at 23 040 x 128 (a quarter of a
// training step's kernel time at the reference's batch size).  Here: one streaming pass per reduction - every thread owns a
// 16-byte channel quad and walks rows, a workgroup folds its row lanes through LDS and writes one partial per channel, a
// C-thread kernel adds the partials in double - and one elementwise pass per direction, ReLU fused on request.
//   forward   mean, biased var over the rows; y = [relu]((x - mean) * invstd * gamma + beta); running statistics updated
//             as nn.BatchNorm does (unbiased variance, momentum factor)
//   backward  dyr = dy * (y > 0) if relu;  dbeta = sum dyr;  dgamma = sum dyr * xhat;
//             dx = gamma * invstd * (dyr - dbeta / rows - xhat * dgamma / rows)
namespace {

constexpr int BN_THREADS = 256;

struct BnParams {
  const float* x;
  const float* y;            // forward output (ReLU mask in the backward), or null
  const float* dy;
  float* out;                // forward: y; backward: dx
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* mean;               // saved statistics [C]
  float* invstd;
  float* dgamma;
  float* dbeta;
  float* part;               // [blocks][2][C]
  long long rows;
  int C, blocks, rows_per_block, relu;
  float momentum, eps;
};

// partial sums of (a, b) per channel over this block's rows:  MODE 0: a = x, b = x * x;  MODE 1: a = dyr, b = dyr * xhat
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(const BnParams p) {
  __shared__ float red[2][BN_THREADS * 4];
  const int quads = p.C >> 2, lanes = BN_THREADS / quads;
  const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
  const long long r0 = (long long)blockIdx.x * p.rows_per_block;
  const long long r1 = r0 + p.rows_per_block < p.rows ? r0 + p.rows_per_block : p.rows;
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1 && rl < lanes) {
    mu = *reinterpret_cast<const f32x4*>(p.mean + 4 * q);
    is = *reinterpret_cast<const f32x4*>(p.invstd + 4 * q);
  }
  // MODE 0: shifted sums.  var = E[x^2] - mean^2 in float32 partials loses ~1e-7 mean^2 / var of relative accuracy when
  // |mean| >> std (a conv bias in front of the BatchNorm, drifting activations); with a per-channel pivot taken from the data
  // itself (row 0: within a few std of the mean) the sums are of (x - pivot) and nothing cancels.  torch uses Welford.
  if (MODE == 0 && rl < lanes) mu = *reinterpret_cast<const f32x4*>(p.x + 4 * q);
  if (rl < lanes)
    for (long long r = r0 + rl; r < r1; r += lanes) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + r * p.C + 4 * q);
      if (MODE == 0) {
        const f32x4 d0 = xv - mu;
        sa += d0;
        sb += d0 * d0;
      } else {
        f32x4 d = *reinterpret_cast<const f32x4*>(p.dy + r * p.C + 4 * q);
        if (p.relu) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(p.y + r * p.C + 4 * q);
#pragma unroll
          for (int c = 0; c < 4; ++c) d[c] = yv[c] > 0.f ? d[c] : 0.f;
        }
        sa += d;
        sb += d * ((xv - mu) * is);
      }
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    red[0][threadIdx.x * 4 + c] = sa[c];
    red[1][threadIdx.x * 4 + c] = sb[c];
  }
  __syncthreads();
  // thread t < C folds channel t over the row lanes (fixed order)
  if ((int)threadIdx.x < p.C) {
    const int ch = threadIdx.x, qq = ch >> 2, cc = ch & 3;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < lanes; ++l) {
      a += red[0][(l * quads + qq) * 4 + cc];
      b += red[1][(l * quads + qq) * 4 + cc];
    }
    p.part[((long long)blockIdx.x * 2 + 0) * p.C + ch] = a;
    p.part[((long long)blockIdx.x * 2 + 1) * p.C + ch] = b;
  }
}

// sums of one channel's block partials: a wave per channel, every lane walks blocks / 64 of them, folded in double in a fixed
// order (8 lanes per channel in 4 workgroups took 19 us at 512 blocks: a chain of dependent loads)
__device__ __forceinline__ bool bn_fold(const BnParams& p, int& ch, double& s, double& s2) {
  __shared__ double red[2][64];
  ch = blockIdx.x;
  const int l = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int blk = l; blk < p.blocks; blk += 64) {
    a += (double)p.part[((long long)blk * 2 + 0) * p.C + ch];
    b += (double)p.part[((long long)blk * 2 + 1) * p.C + ch];
  }
  red[0][l] = a;
  red[1][l] = b;
  __syncthreads();
  if (l != 0) return false;
  s = 0.0; s2 = 0.0;
  for (int k = 0; k < 64; ++k) {
    s += red[0][k];
    s2 += red[1][k];
  }
  return true;
}

__global__ __launch_bounds__(64) void bn_finalize_fwd_kernel(const BnParams p) {
  int ch;
  double s, s2;
  if (!bn_fold(p, ch, s, s2)) return;
  const double n = (double)p.rows;
  const double ms = s / n;                    // mean of (x - pivot), pivot = row 0 of the channel (bn_reduce_kernel)
  const double m = (double)p.x[ch] + ms;
  double var = s2 / n - ms * ms;
  if (var < 0.0) var = 0.0;
  p.mean[ch] = (float)m;
  p.invstd[ch] = (float)(1.0 / sqrt(var + (double)p.eps));
  if (p.running_mean) p.running_mean[ch] = (1.f - p.momentum) * p.running_mean[ch] + p.momentum * (float)m;
  if (p.running_var) {
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    p.running_var[ch] = (1.f - p.momentum) * p.running_var[ch] + p.momentum * (float)unb;
  }
}

__global__ __launch_bounds__(64) void bn_finalize_bwd_kernel(const BnParams p) {
  int ch;
  double s, s2;
  if (!bn_fold(p, ch, s, s2)) return;
  p.dbeta[ch] = (float)s;
  p.dgamma[ch] = (float)s2;
}

// MODE 0: y = [relu]((x - mean) * invstd * gamma + beta);  MODE 1: dx
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const BnParams p) {
  const int quads = p.C >> 2;
  const long long total = p.rows * quads;
  const float inv_n = 1.f / (float)p.rows;
  for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * BN_THREADS) {
    const int q = (int)(i % quads);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + 4 * q);
    const f32x4 is = *reinterpret_cast<const f32x4*>(p.invstd + 4 * q);
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 4 * q);
    f32x4 o;
    if (MODE == 0) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.beta + 4 * q);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = (xv[c] - mu[c]) * is[c] * g[c] + b[c];
        o[c] = p.relu ? magat_relu(v) : v;
      }
    } else {
      f32x4 d = *reinterpret_cast<const f32x4*>(p.dy + i * 4);
      if (p.relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(p.y + i * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = yv[c] > 0.f ? d[c] : 0.f;
      }
      const f32x4 db = *reinterpret_cast<const f32x4*>(p.dbeta + 4 * q);
      const f32x4 dg = *reinterpret_cast<const f32x4*>(p.dgamma + 4 * q);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xh = (xv[c] - mu[c]) * is[c];
        o[c] = g[c] * is[c] * (d[c] - db[c] * inv_n - xh * dg[c] * inv_n);
      }
    }
    *reinterpret_cast<f32x4*>(p.out + i * 4) = o;
  }
}

int bn_blocks(long long rows, int C, int* rpb) {
  const int lanes = BN_THREADS / (C >> 2);
  long long b = (rows + (long long)lanes * 8 - 1) / ((long long)lanes * 8);      // ~8 rows per thread
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  *rpb = (int)((rows + b - 1) / b);
  return (int)((rows + *rpb - 1) / *rpb);
}

}  // namespace

size_t magat_bn_train_workspace_floats(long long rows, int C) {
  if (rows <= 0 || C <= 0 || C % 4 || C > 256 || BN_THREADS % (C >> 2)) return 0;
  int rpb;
  return (size_t)bn_blocks(rows, C, &rpb) * 2 * C;
}

int magat_bn_train_forward_f32(const float* x, float* y, long long rows, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int relu, float* save_mean,
                               float* save_invstd, float* workspace, void* stream) {
  if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !workspace) return MAGAT_ERR_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || C > 256) return MAGAT_ERR_BAD_SHAPE;
  if (BN_THREADS % (C >> 2)) return MAGAT_ERR_UNSUPPORTED;      // (channel quads must divide the workgroup: C = 4, 8, .. 256 powers of two x 4)
  BnParams p = {};
  p.x = x; p.out = y; p.gamma = gamma; p.beta = beta; p.running_mean = running_mean; p.running_var = running_var;
  p.mean = save_mean; p.invstd = save_invstd; p.part = workspace; p.rows = rows; p.C = C; p.relu = relu;
  p.momentum = momentum; p.eps = eps;
  p.blocks = bn_blocks(rows, C, &p.rows_per_block);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(p.blocks), dim3(BN_THREADS), 0, st, p);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, st, p);
  const long long total = rows * (C >> 2);
  long long grid = (total + BN_THREADS - 1) / BN_THREADS;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL((bn_apply_kernel<0>), dim3((unsigned)grid), dim3(BN_THREADS), 0, st, p);
  return magat_check_launch();
}

int magat_bn_train_backward_f32(const float* x, const float* y, const float* dy, float* dx, long long rows, int C,
                                const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dgamma,
                                float* dbeta, float* workspace, void* stream) {
  if (!x || !dy || !dx || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !workspace || (relu && !y))
    return MAGAT_ERR_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || C > 256) return MAGAT_ERR_BAD_SHAPE;
  if (BN_THREADS % (C >> 2)) return MAGAT_ERR_UNSUPPORTED;
  BnParams p = {};
  p.x = x; p.y = y; p.dy = dy; p.out = dx; p.gamma = gamma; p.mean = const_cast<float*>(save_mean);
  p.invstd = const_cast<float*>(save_invstd); p.dgamma = dgamma; p.dbeta = dbeta; p.part = workspace; p.rows = rows; p.C = C;
  p.relu = relu;
  p.blocks = bn_blocks(rows, C, &p.rows_per_block);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(p.blocks), dim3(BN_THREADS), 0, st, p);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, st, p);
  const long long total = rows * (C >> 2);
  long long grid = (total + BN_THREADS - 1) / BN_THREADS;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL((bn_apply_kernel<1>), dim3((unsigned)grid), dim3(BN_THREADS), 0, st, p);
  return magat_check_launch();
}
