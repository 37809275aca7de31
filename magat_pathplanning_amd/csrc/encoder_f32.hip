// Per-agent CNN encoder + compressMLP on gfx950 (reference decentralplanner_GAT_bottleneck.py:90-166,
// 291-302; graphs/models/resnet_pytorch.py:40-73, 334-524), inference form: BatchNorm folded into the
// conv weights host-side (fp64 fold, fp32 store), residual 1x1 branch appended as an extra K segment,
// avgpool+fc(+Flatten+Linear) folded into one "valid" conv.  All GEMM-shaped layers run on the fp32
// MFMA kernel of conv_gemm_f32.hip; only the 3-channel first conv is a direct VALU kernel.
#include <cstdlib>

#include "magat_common.h"

namespace {

// conv3x3(3->32, pad 1)+BN+ReLU on (M,3,H,W) NCHW -> pixel-major [H*W][M][32], as an MFMA GEMM with the
// 27-wide im2col row padded to K=32.  A workgroup stages 32 agents' zero-padded images in LDS
// ((H+2)x(W+2) per channel, odd agent stride -> conflict-free column reads); each wave then walks
// output pixels: its 32x32 tile is (32 agents) x (32 channels) at one pixel, i.e. 4 KB of contiguous
// output, from 16 v_mfma_f32_32x32x2_f32 whose A operands are plain ds_read_b32 gathers.
// HC/WC: compile-time map size (0 = runtime H, W).  With constants every index split is a multiply-shift and the
// staging loop unrolls into 12 back-to-back 16-byte loads per thread (the 32 agents' inputs are one contiguous
// 46 KB run), instead of 46 dependent load->divide->store rounds.
template <int HC, int WC>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int M, int Hr, int Wr, long long out_pix_stride,
                                                         long long out_tile_stride, long long out_plane, int out_gl,
                                                         int* range_flag, const int* run_if, int BH, float* absmax) {
  extern __shared__ float img[];
  if (run_if && *run_if == 0) return;        // range-guard re-run: nothing to do unless the split path clamped
  const int H = HC ? HC : Hr, W = WC ? WC : Wr;
  // Row bands (maps whose 32 padded images do not fit the LDS, e.g. 19 x 19 at FOV 17): blockIdx.y owns output rows
  // [ybase, ybase + bh) and stages input rows ybase - 1 .. ybase + bh only.  BH = H (one band) for the usual sizes.
  const int BHe = HC ? HC : BH;
  const int ybase = HC ? 0 : (int)blockIdx.y * BHe;
  const int bh = min(BHe, H - ybase);
  const int PW = W + 2, PHW = (BHe + 2) * PW;
  const int PS = (3 * PHW) | 1;            // odd per-agent stride
  const int HW = H * W;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float amax = 0.f;          // calibration launches: largest stem output of this wave
  // (grid-stride over the 32-agent blocks: one block per workgroup in ordinary launches; the range guard's predicated re-run
  //  is launched with a capped grid - a launch that returns at once still pays for every workgroup it dispatches, and these
  //  carry 65 KB of LDS each: 1600 .. 4000 of them were 10 .. 26 us per forward)
  for (int blk = blockIdx.x; blk * 32 < M; blk += gridDim.x) {
  if (blk != (int)blockIdx.x) __syncthreads();
  const int m0 = blk * 32;
  for (int i = t; i < 32 * PS; i += 256) img[i] = 0.f;
  __syncthreads();
  const int per = 3 * HW;
  if constexpr (HC != 0 && (32 * 3 * HC * WC) % 4 == 0) {
    constexpr int NV = 32 * 3 * HC * WC / 4;          // float4s in the block's contiguous input run
    constexpr int IT = (NV + 255) / 256;
    const long long navail = ((long long)(M - m0) * per) / 4;   // run may be cut short by the batch end
    const f32x4* src = reinterpret_cast<const f32x4*>(x + (long long)m0 * per);
    f32x4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i4 = t + 256 * k;
      v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i4 < NV && i4 < navail) v[k] = src[i4];
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i4 = t + 256 * k;
      if (i4 < NV && i4 < navail) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * i4 + e;
          const int a = f / per, r = f - a * per;
          const int c = r / HW, q = r - c * HW;
          const int y = q / W, xx = q - y * W;
          img[a * PS + c * PHW + (y + 1) * PW + xx + 1] = v[k][e];
        }
      }
    }
    // a batch end that is not float4-aligned leaves < 4 trailing floats: scalar tail
    const long long done = navail * 4, want = (long long)(M - m0 < 32 ? M - m0 : 32) * per;
    for (long long f = done + t; f < want; f += 256) {
      const int a = (int)(f / per), r = (int)(f - (long long)a * per);
      const int c = r / HW, q = r - c * HW;
      const int y = q / W, xx = q - y * W;
      img[a * PS + c * PHW + (y + 1) * PW + xx + 1] = x[(long long)m0 * per + f];
    }
  } else {
    const int brows = bh + 2, bper = 3 * brows * W;      // staged rows per channel: global rows ybase - 1 + ry
    for (int i = t; i < 32 * bper; i += 256) {
      const int a = i / bper, r = i - a * bper;
      const int c = r / (brows * W), q = r - c * brows * W;
      const int ry = q / W, xx = q - ry * W;
      const int y = ybase - 1 + ry;
      if (m0 + a < M && y >= 0 && y < H)
        img[a * PS + c * PHW + ry * PW + xx + 1] = x[(long long)(m0 + a) * per + c * HW + y * W + xx];
    }
  }
  // B operand (weights) and per-k LDS tap offsets for this lane half: k = s + 16*(lane>>5)
  float bw[16];
  int koff[16];
  const int co = lane & 31;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = s + 16 * (lane >> 5);
    const bool kok = k < 27;
    bw[s] = kok ? wt[co * 27 + k] : 0.f;
    const int kk = kok ? k : 0;
    koff[s] = (kk / 9) * PHW + ((kk % 9) / 3) * PW + (kk % 3);
  }
  f32x4 bch[4];     // bias of the channels this lane stores: 8q + 4*(lane>>5) + 0..3
#pragma unroll
  for (int q = 0; q < 4; ++q) bch[q] = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4 * (lane >> 5));
  __syncthreads();
  const int abase = (lane & 31) * PS;
  for (int lpix = wave; lpix < bh * W; lpix += 4) {
    const int oy = lpix / W, ox = lpix - oy * W;
    const int base = abase + oy * PW + ox;
    const int pix = (ybase + oy) * W + ox;      // the output pixel in the whole map
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a = img[base + koff[s]];
      if (s + 16 * (lane >> 5) >= 27) a = 0.f;
      // operands swapped (weights as the row operand): D[channel][agent], so a lane ends up with four runs of
      // 4 consecutive channels of ONE agent -> 16-byte stores instead of 4-byte ones
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[s], a, acc, 0, 0, 0);
    }
    const int agent = m0 + (lane & 31);
    if (agent < M) {
      float* o = out + (long long)pix * out_pix_stride + magat_row_off(agent, 32, out_tile_stride) + 4 * (lane >> 5);
      // granule-major tile ([8 channel quads][128 agents][4], magat_hip.h in_gl/out_gl): quad 2q + (lane>>5) of agent
      // a sits next to its neighbour agents' -> 512-byte runs per half wave and store
      float* og = out + (long long)pix * out_pix_stride + (long long)(agent >> 7) * out_tile_stride +
                  ((lane >> 5) * 128 + (agent & 127)) * 4;
      if (out_gl == 2) {
        // f16 plane granules (magat_hip.h out_gl = 2): quads 2 ks, 2 ks + 1 form the next layer's k-step-ks operand
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        char* ob = reinterpret_cast<char*>(out) + ((long long)pix * out_pix_stride + (long long)(agent >> 7) * out_tile_stride) * 4 +
                   (lane >> 5) * 2048 + (agent & 127) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          unsigned w1[4], w2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {        // pairs (2e, 2e+1) of the 8 values of quads 2 ks, 2 ks + 1
            const int q = 2 * ks + (e >> 1), c = 2 * (e & 1);
            const float a0 = magat_relu(acc[4 * q + c] + bch[q][c]), b0 = magat_relu(acc[4 * q + c + 1] + bch[q][c + 1]);
            if ((!(a0 <= 65504.f) || !(b0 <= 65504.f)) && range_flag) atomicOr(range_flag, 1);      // range guard (rare)
            const float a = __builtin_amdgcn_fmed3f(a0, -65504.f, 65504.f);
            const float b2 = __builtin_amdgcn_fmed3f(b0, -65504.f, 65504.f);
            const h2 h = __builtin_convertvector(f2{a, b2}, h2);
            const h2 r = __builtin_convertvector(f2{a - (float)h[0], b2 - (float)h[1]}, h2);
            w1[e] = __builtin_bit_cast(unsigned, h);
            w2[e] = __builtin_bit_cast(unsigned, r);
          }
          *reinterpret_cast<u4*>(ob + ks * 4096) = u4{w1[0], w1[1], w1[2], w1[3]};
          *reinterpret_cast<u4*>(ob + 256 * 32 + ks * 4096) = u4{w2[0], w2[1], w2[2], w2[3]};
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = magat_relu(acc[4 * q + c] + bch[q][c]);
        if (absmax) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        if (out_gl) {
          *reinterpret_cast<f32x4*>(og + q * 1024) = v;
        } else if (out_plane == 0) {
          *reinterpret_cast<f32x4*>(o + 8 * q) = v;
        } else {      // two f16 planes (operand format of the f16x3 convs, conv_gemm_bf16x6.hip in_fmt 5)
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          typedef float f2 __attribute__((ext_vector_type(2)));
          unsigned short* ob = reinterpret_cast<unsigned short*>(out) + (o - out) + 8 * q;
          unsigned w1[2], w2[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float a = fminf(v[2 * c], 65504.f), b2 = fminf(v[2 * c + 1], 65504.f);
            const h2 h = __builtin_convertvector(f2{a, b2}, h2);
            const h2 r = __builtin_convertvector(f2{a - (float)h[0], b2 - (float)h[1]}, h2);
            w1[c] = __builtin_bit_cast(unsigned, h);
            w2[c] = __builtin_bit_cast(unsigned, r);
          }
          *reinterpret_cast<uint2*>(ob) = uint2{w1[0], w1[1]};
          *reinterpret_cast<uint2*>(ob + out_plane) = uint2{w2[0], w2[1]};
        }
      }
    }
  }
  }      // (32-agent blocks)
  if (absmax) {
    amax = wave_max(amax);
    if (lane == 0 && amax > 0.f) atomicMax(reinterpret_cast<unsigned*>(absmax), __builtin_bit_cast(unsigned, amax));
  }
}

int enc_chunk_agents(int M) {
  long long c = magat_opt(MAGAT_OPT_ENC_CHUNK);
  if (c < 128) c = 128;
  if (c > M) c = M;
  return (int)c;
}

struct BlockShape {
  int cin, cout, stride;
};


// feat[m][n] = bias[n] + sum over the pooled cells (fixed order) of part[c][m][n]: the second half of the split-K head
__global__ __launch_bounds__(256) void head_sum_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                       float* __restrict__ feat, int ldfeat, int M, int nf, int cells,
                                                       const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q = nf / 4;
  if (i >= (long long)M * q) return;
  const int m = (int)(i / q), n = (int)(i % q) * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(bias + n);
  const float* src = part + (long long)m * nf + n;
  for (int c = 0; c < cells; ++c) acc += *reinterpret_cast<const f32x4*>(src + (long long)c * M * nf);
  *reinterpret_cast<f32x4*>(feat + (long long)m * ldfeat + n) = acc;
}

// range guard bookkeeping at the end of a forward: status[2] = this forward's flag, status[1] += 1 when the float32 re-run
// happened, and the working flag status[0] is cleared for the next forward (no memset launch per forward)
__global__ void guard_count_kernel(int* status) {
  const int f = status[0];
  status[2] = f;
  if (f != 0) status[1] += 1;
  status[0] = 0;
}

}  // namespace

static int conv_first_launch(const float* x, const float* wt, const float* bias, float* out, int M, int H, int W,
                             long long pix_stride, long long tile_stride, void* stream, long long out_plane = 0,
                             int out_gl = 0, int* range_flag = nullptr, const int* run_if = nullptr, int tag = MAGAT_TAG_CONV_FIRST,
                             float* absmax = nullptr);

extern "C" int magat_encoder_stem_block_f32(const magat_encoder_desc* d, const float* x, void* out, void* ctr, int M, int form,
                                            int32_t* range_flag, void* stream) {
  if (!d || !d->pack || !x || !out || !ctr) return MAGAT_ERR_NULL;
  if (M <= 0 || form < 0 || form > 2) return MAGAT_ERR_BAD_SHAPE;
  if (d->variant != 0 && d->variant != 1) return MAGAT_ERR_UNSUPPORTED;
  if (d->off[30] == 0) return MAGAT_ERR_UNSUPPORTED;        // (the pack holds no plane-granule weight copies)
  const float* pk = d->pack;
  const int H = d->H, W = d->W;
  const float* sp = d->scaled_off > 0 ? pk + d->scaled_off : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool can8 = H == 11 && W == 11 && d->l1frag_off > 0;
  if (form == 0) form = (can8 && magat_opt(MAGAT_OPT_L1_FUSED) >= 2) ? 2 : 1;
  if (form == 2) {
    if (!can8) return MAGAT_ERR_UNSUPPORTED;
    return magat_stem8(x, sp ? sp : pk + d->off[0], sp ? sp + 864 : pk + d->off[1], pk + d->l1frag_off,
                       sp ? sp + 896 : pk + d->off[3], out, ctr, M, H, W, st, reinterpret_cast<int*>(range_flag));
  }
  if (magat_layer1_fused_lds(W) == 0) return MAGAT_ERR_UNSUPPORTED;
  // (the K-permuted f16 copy of layer1.conv1 sits behind the plain one: two planes + one scale float, padded to 4 floats)
  const int64_t permuted = ((int64_t)32 * 9 * 32 + 1 + 3) & ~3LL;
  return magat_layer1_fused(x, sp ? sp : pk + d->off[0], sp ? sp + 864 : pk + d->off[1], pk + d->off[24] + permuted,
                            sp ? sp + 896 : pk + d->off[3], out, ctr, M, H, W, st, reinterpret_cast<int*>(range_flag));
}

extern "C" int magat_conv_first_f32(const float* x, const float* wt, const float* bias, float* out, int M, int H,
                                    int W, void* stream) {
  return conv_first_launch(x, wt, bias, out, M, H, W, (long long)M * 32, (long long)MAGAT_TILE_ROWS * 32, stream);
}

extern "C" int magat_conv_first_tiled_f32(const float* x, const float* wt, const float* bias, float* out, int M, int H,
                                          int W, void* stream) {
  return conv_first_launch(x, wt, bias, out, M, H, W, (long long)MAGAT_TILE_ROWS * 32,
                           (long long)H * W * MAGAT_TILE_ROWS * 32, stream);
}

static int conv_first_launch(const float* x, const float* wt, const float* bias, float* out, int M, int H, int W,
                             long long pix_stride, long long tile_stride, void* stream, long long out_plane, int out_gl,
                             int* range_flag, const int* run_if, int tag, float* absmax) {
  if (!x || !wt || !bias || !out) return MAGAT_ERR_NULL;
  if (M <= 0 || H <= 0 || W <= 0) return MAGAT_ERR_BAD_SHAPE;
  int BH = H;      // rows per band: the whole map when its 32 padded images fit the LDS
  auto lds_for = [&](int bh) { return sizeof(float) * 32 * (size_t)((3 * (bh + 2) * (W + 2)) | 1); };
  while (BH > 1 && lds_for(BH) > 160 * 1024) BH = (BH + 1) / 2;
  const size_t lds = lds_for(BH);
  if (lds > 160 * 1024) return MAGAT_ERR_UNSUPPORTED;
  const int bands = (H + BH - 1) / BH;
  const bool c11 = H == 11 && W == 11 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (lds > 64 * 1024 &&
      magat_ensure_dyn_lds(c11 ? reinterpret_cast<const void*>(&conv_first_kernel<11, 11>)
                               : reinterpret_cast<const void*>(&conv_first_kernel<0, 0>),
                           c11 ? MAGAT_LDS_CONV_FIRST11 : MAGAT_LDS_CONV_FIRST, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  int blocks = (M + 31) / 32;
  if (run_if && blocks > 256) blocks = 256;      // predicated re-run: a capped grid that walks the blocks (see the kernel)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pid = magat_prof_begin(tag, st);
  if (c11)
    hipLaunchKernelGGL((conv_first_kernel<11, 11>), dim3(blocks), dim3(256), lds, st, x, wt, bias, out, M, H, W,
                       pix_stride, tile_stride, out_plane, out_gl, range_flag, run_if, H, absmax);
  else
    hipLaunchKernelGGL((conv_first_kernel<0, 0>), dim3(blocks, bands), dim3(256), lds, st, x, wt, bias, out, M, H, W,
                       pix_stride, tile_stride, out_plane, out_gl, range_flag, run_if, BH, absmax);
  magat_prof_end(pid, st);
  return magat_check_launch();
}

// Block convolutions on the split-MFMA kernels (split weights in the pack).  Option CONV_SPLIT = bit mask of
// BasicBlocks that use them (bit l = layer l+1); default 7 = all three; 0 keeps every layer on the fp32 MFMA kernel.
static int enc_split_mask(const magat_encoder_desc* d, int v) {
  int m = 0;
  for (int l = 0; l < 3; ++l)
    if ((v >> l & 1) && d->off[24 + 2 * l] > 0 && d->off[25 + 2 * l] > 0) m |= 1 << l;
  return m;
}

// (Split flavour of those layers: f16x3 - two f16 planes, three v_mfma_f32_32x32x16_f16 per product, in_fmt 4; a layer is in the
//  mask only when the pack carries its f16 weight planes.)

// floats per agent of one rotating activation buffer
static size_t enc_buf_floats_per_agent(const magat_encoder_desc* d) {
  if (d->variant >= 2) return (size_t)d->H * d->W * 32;    // plain CNNs: the first map is the largest
  const int Ho = (d->H + 2 - 3) / 2 + 1, Wo = (d->W + 2 - 3) / 2 + 1;
  const size_t a0 = (size_t)d->H * d->W * 32;
  const size_t a3 = (size_t)Ho * Wo * (d->variant == 0 ? 128 : 64);
  return a0 > a3 ? a0 : a3;
}

constexpr size_t ENC_STATUS_BYTES = 256;     // status block at the head of the workspace (magat_hip.h)

extern "C" size_t magat_encoder_workspace_bytes(const magat_encoder_desc* d, int M) {
  if (!d || M <= 0) return 0;
  const int mc = (enc_chunk_agents(M) + MAGAT_TILE_ROWS - 1) / MAGAT_TILE_ROWS * MAGAT_TILE_ROWS;   // whole agent tiles
  return ENC_STATUS_BYTES + 3 * magat_align_up(enc_buf_floats_per_agent(d) * (size_t)mc * sizeof(float), 256);
}

extern "C" int magat_encoder_read_status(const void* workspace, int32_t status_host[2], void* stream) {
  if (!workspace || !status_host) return MAGAT_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t w[3];
  if (hipMemcpyAsync(w, workspace, sizeof(w), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return MAGAT_ERR_LAUNCH;
  status_host[0] = w[2];      // the last forward's flag
  status_host[1] = w[1];      // re-run count
  return MAGAT_OK;
}

// y = act(x @ w^T + b) on the float32 kernel with the guard's predicate
static int enc_linear(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int N, int K,
                      int relu, int tag, const int32_t* run_if, void* stream, float* absmax = nullptr) {
  magat_conv_gemm_desc d = {};
  d.tag = tag;
  d.in = x; d.wt = w; d.bias = b; d.out = y;
  d.M = M; d.Cin = K; d.lda = ldx; d.Hin = d.Win = 1; d.kH = d.kW = 1; d.stride = 1; d.pad = 0;
  d.Hout = d.Wout = 1; d.Cout = N; d.ldc = ldy; d.relu = relu;
  d.run_if = run_if;
  d.absmax = absmax;
  return magat_conv_gemm_f32(&d, stream);
}

// One pass of the ResNet encoders (variant 0 / 1) over all agents with the BasicBlocks in `split` on the split-MFMA
// kernels.  range_flag: where those kernels report a clamp (null: not tracked).  run_if: predicate of the float32
// re-run (every launch of the pass returns immediately unless *run_if != 0; only valid with split == 0).
static int enc_run_resnet(const magat_encoder_desc* d, const float* x, float* feat, int ldfeat, float* comp, int ldcomp,
                          float* bufbase, int M, void* stream, int split, int32_t* range_flag, const int32_t* run_if,
                          float* absmax = nullptr, int32_t* book = nullptr, bool* booked = nullptr, bool* self_guarded = nullptr) {
  const int H = d->H, W = d->W;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int mc = enc_chunk_agents(M);
  const int mcp = (mc + MAGAT_TILE_ROWS - 1) / MAGAT_TILE_ROWS * MAGAT_TILE_ROWS;
  const size_t bstride = magat_align_up(enc_buf_floats_per_agent(d) * (size_t)mcp * sizeof(float), 256) / sizeof(float);
  // activations are TILE-major: [agent tile][pixel][128][C]
  auto pixs = [](int c) { return (int64_t)MAGAT_TILE_ROWS * c; };
  auto tiles = [](int npix, int c) { return (int64_t)npix * MAGAT_TILE_ROWS * c; };
  float* buf[3] = {bufbase, bufbase + bstride, bufbase + 2 * bstride};
  const float* pk = d->pack;
  const BlockShape shapes[3] = {{32, 32, 2}, {32, 64, 1}, {64, 128, 1}};
  const int nblocks = d->variant == 0 ? 3 : 2;
  const bool rerun = run_if != nullptr;
  auto tagof = [&](int t) { return rerun ? MAGAT_TAG_UNTAGGED : t; };     // the re-run is timed as ONE span by the caller
  // (ABI 7) compressMLP's rows as bf16 too; the guard's re-run and the calibration pass leave them to their callers
  unsigned short* const comp16 = (rerun || absmax) ? nullptr : static_cast<unsigned short*>(d->comp_bf16);
  hipStream_t st = static_cast<hipStream_t>(stream);

  // Granule-major activation tiles ([C/4][128 agents][4], magat_hip.h in_gl/out_gl) between the layers when every
  // BasicBlock conv runs on the f16x3 direct kernel: its one-lane-per-agent fragment loads and epilogue stores are then
  // 512-byte runs.  The last conv2 writes row-major tiles again for the pooled head (fp32 MFMA kernel).
  const bool gl = split == (1 << nblocks) - 1 && magat_conv_direct_enabled();      // (mask bit l set <=> layer l has its f16 planes)
  // ... and, when the pack carries the K-permuted weight copies (off[30]), as f16 PLANE granules (in_gl/out_gl = 2): every
  // activation is split into its two half-precision planes once, by the epilogue that produces it, instead of once per
  // tap by every consumer's loader.  Option CONV_PCHAIN=0 keeps float32 granules.
  int lay = gl ? 1 : 0;
  if (gl && d->off[30] != 0 && magat_opt(MAGAT_OPT_CONV_PCHAIN)) lay = 2;
  // float offset of the permuted copy behind an f16 weight block of cout x ktot weights (two planes + one scale float,
  // padded to 4 floats)
  auto permuted = [&](int cout, int ktot) { return lay == 2 ? (int64_t)(((int64_t)cout * ktot + 1 + 3) & ~3LL) : 0; };
  // the guard's re-run: every float32 layer behind the stem goes into ONE predicated launch (magat_conv_gemm_chain_f32)
  const bool chained = rerun;
  for (int m0 = 0; m0 < M; m0 += mc) {
    const int mm = (M - m0) < mc ? (M - m0) : mc;
    magat_conv_gemm_desc chain[10];
    int nchain = 0;
    auto run_or_chain = [&](const magat_conv_gemm_desc& g) -> int {
      if (chained && nchain < 10) { chain[nchain] = g; chain[nchain].run_if = nullptr; ++nchain; return MAGAT_OK; }
      return magat_conv_gemm_f32(&g, stream);
    };
    // Plane chain: the stem and layer1.conv1 run as ONE kernel (layer1_fused.hip) - the 121-pixel stem output never
    // reaches HBM; buf[0] receives only its 36 stride-2 pixels, the input of the block's residual 1x1 branch.
    // Option L1_FUSED=0 keeps the two launches.
    const bool fused1 = lay == 2 && magat_layer1_fused_lds(W) != 0 && magat_opt(MAGAT_OPT_L1_FUSED) != 0;
    // the whole chain in two launches (fused stem + the merged chain kernel): the path the activation scales are folded for
    const bool full_path = fused1 && d->chain_off > 0 && Ho == 6 && Wo == 6 && nblocks == 3 && d->chain3_off > 0 &&
                           magat_opt(MAGAT_OPT_BLOCK_FUSED) >= 2;
    const float* sp = (full_path && d->scaled_off > 0) ? pk + d->scaled_off : nullptr;      // activation-scale block
    int rc;
    // ... as the eight-agent-group kernel (block_fused.hip stem8_kernel: every stem pixel once, no im2col instructions) when
    // the map is 11 x 11 and the pack holds layer1.conv1 fragment-major (option L1_FUSED = 2, the default)
    const bool stem8 = fused1 && H == 11 && W == 11 && d->l1frag_off > 0 && magat_opt(MAGAT_OPT_L1_FUSED) >= 2;
    // latency form of a few-agent call (option LAT_AGENTS; block_lat.hip): ONE launch for the whole encoder - stem, layer1.conv1,
    // the chain, head, compressMLP and the range guard with one agent per workgroup.  Decided here because it takes the stem along.
    const int Mform0 = d->form_agents > 0 ? d->form_agents : M;
    const bool lat_all = full_path && stem8 && !rerun && !absmax && Mform0 <= magat_opt(MAGAT_OPT_LAT_AGENTS) && mm == M &&
                         d->headfrag_off > 0 && d->compfrag_off > 0 && d->n_feat == 128 &&
                         (d->n_comp == 128 || d->n_comp == 64 || d->n_comp == 32) && comp && split && magat_opt(MAGAT_OPT_HEAD_F16);
    if (lat_all)
      rc = MAGAT_OK;
    else if (stem8)
      rc = magat_stem8(x + (size_t)m0 * 3 * H * W, sp ? sp : pk + d->off[0], sp ? sp + 864 : pk + d->off[1],
                       pk + d->l1frag_off, sp ? sp + 896 : pk + d->off[3], buf[1], buf[0], mm, H, W, st, range_flag);
    else if (fused1)
      rc = magat_layer1_fused(x + (size_t)m0 * 3 * H * W, sp ? sp : pk + d->off[0], sp ? sp + 864 : pk + d->off[1],
                              pk + d->off[24] + permuted(32, 9 * 32), sp ? sp + 896 : pk + d->off[3], buf[1], buf[0], mm, H, W,
                              st, range_flag);
    else
      rc = conv_first_launch(x + (size_t)m0 * 3 * H * W, pk + d->off[0], pk + d->off[1], buf[0], mm, H, W,
                             (long long)MAGAT_TILE_ROWS * 32, (long long)H * W * MAGAT_TILE_ROWS * 32, stream, 0, lay,
                             range_flag, run_if, tagof(MAGAT_TAG_CONV_FIRST), absmax);
    if (rc != MAGAT_OK) return rc;
    bool head_done = false;          // head + compressMLP ran in the chain kernel's epilogue (latency form)
    bool compress_done = false;      // compressMLP rode in the head's epilogue
    bool comp16_done = false;        // ... and wrote the bf16 rows (desc.comp_bf16) too
    int cur = 0;              // buffer holding the block input
    int hin = H, win = W;
    int lstart = 0;
    bool pooled_in = false;     // buf[cur] already holds the 2x2-pooled map [tile][(hin/2)(win/2)][128][clast]
    // BasicBlock chain kernel (block_fused.hip): layer1.conv2+downsample -> layer2.conv1 -> layer2.conv2+downsample in one
    // launch with the 6x6 maps of an 8-agent group in LDS (3.35 GB of HBM traffic per 51 200 agents become 0.94 GB).
    // Needs the fused stem's plane-granule outputs; option BLOCK_FUSED=0 keeps the layer-by-layer kernels.
    // The pooled map goes to the head granule-major ([cell][128 / 4][128 agents][4 floats]) when the head is the f16x3 direct
    // GEMM: its loader then reads 512 contiguous bytes per half wave instead of 32 bytes out of every agent's 512-byte row, and
    // the chain kernel stores 128-byte runs instead of 16-byte pieces.  (Not for the few-agent form of the head, which splits K
    // by pooled cell on the float32 kernel.)
    const int clast_ = shapes[nblocks - 1].cout;
    // (the agent count the head's form is chosen on: the whole call's, or - a shard of a larger batch - the global one)
    const int Mform = d->form_agents > 0 ? d->form_agents : M;
    const bool head_splitk = !absmax && !chained && Mform <= magat_opt(MAGAT_OPT_HEAD_SPLITK) && (clast_ & 3) == 0 &&
                             (d->n_feat & 3) == 0 && (size_t)9 * d->n_feat <= enc_buf_floats_per_agent(d);
    const bool head_gl = full_path && !rerun && split && d->head16_off > 0 && (clast_ % 32) == 0 && (d->n_feat % 32) == 0 &&
                         magat_opt(MAGAT_OPT_HEAD_F16) && magat_conv_direct_enabled() && !head_splitk &&
                         magat_block_full_out_gl();
    if (full_path) {
      // both chain kernels as ONE launch: layer2's output map stays in LDS as layer3's input - with eight agents per workgroup, or,
      // for the few agents of a batch-1 step (option LAT_AGENTS, chosen on the global agent count like the head's form), one
      // agent per workgroup (block_lat.hip: the same pooled map bit for bit, 2 460 instead of 13 428 matrix instructions deep)
      const bool lat = !rerun && Mform <= magat_opt(MAGAT_OPT_LAT_AGENTS);
      // ... which then runs the encoder head and compressMLP in its epilogue as well (ABI 8 fragment-major weights): the products
      // of the long-K f16x3 head and of the f16x3 compressMLP in their order - feat / comp bit-identical to the batched forms
      magat_lat_head lh = {};
      if (lat && d->headfrag_off > 0 && d->compfrag_off > 0 && d->n_feat == 128 &&
          (d->n_comp == 128 || d->n_comp == 64 || d->n_comp == 32) && comp && split && magat_opt(MAGAT_OPT_HEAD_F16)) {
        lh.hfrag = pk + d->headfrag_off; lh.cfrag = pk + d->compfrag_off;
        lh.hbias = pk + d->off[15]; lh.cbias = pk + d->off[17];
        lh.insc = d->scaled_off > 0 ? pk + d->scaled_off + 1349 : nullptr;
        lh.insc2 = d->scaled_off > 0 ? pk + d->scaled_off + 1350 : nullptr;
        lh.feat = feat + (size_t)m0 * ldfeat; lh.ldfeat = ldfeat;
        lh.comp = comp + (size_t)m0 * ldcomp; lh.ldcomp = ldcomp; lh.ncomp = d->n_comp;
        head_done = true;
      }
      // ... and the encoder's range guard too: a workgroup whose planes clamped (or the stem's did) recomputes its agent in
      // float32 from the raw state maps, the last workgroup does the guard's bookkeeping - no predicated launches behind it
      magat_lat_stem ls = {};
      if (lat_all && head_done) {
        ls.x = x + (size_t)m0 * 3 * H * W; ls.w0 = sp ? sp : pk + d->off[0]; ls.b0 = sp ? sp + 864 : pk + d->off[1];
        ls.w1f = pk + d->l1frag_off; ls.b1 = sp ? sp + 896 : pk + d->off[3];
      }
      magat_lat_guard lg = {};
      const bool inguard = head_done && range_flag && self_guarded && mm == M && H == 11 && W == 11;
      if (inguard) {
        lg.x = x + (size_t)m0 * 3 * H * W; lg.pack = pk; lg.off = d->off; lg.book = reinterpret_cast<int*>(range_flag);
      }
      if (lat)
        rc = magat_block_lat(buf[1], buf[0], pk + d->chain_off, sp ? sp + 928 : pk + d->off[5], sp ? sp + 960 : pk + d->off[7],
                             sp ? sp + 1024 : pk + d->off[9], buf[2], pk + d->chain3_off, sp ? sp + 1088 : pk + d->off[11],
                             sp ? sp + 1216 : pk + d->off[13], mm, reinterpret_cast<int*>(range_flag), st,
                             sp ? sp + 1344 : nullptr, head_gl ? 1 : 0, head_done ? &lh : nullptr, inguard ? &lg : nullptr,
                             (lat_all && head_done) ? &ls : nullptr);
      if (lat && rc == MAGAT_OK && inguard) *self_guarded = true;
      else
      rc = magat_block_full(buf[1], buf[0], pk + d->chain_off, sp ? sp + 928 : pk + d->off[5], sp ? sp + 960 : pk + d->off[7],
                            sp ? sp + 1024 : pk + d->off[9], buf[2], pk + d->chain3_off, sp ? sp + 1088 : pk + d->off[11],
                            sp ? sp + 1216 : pk + d->off[13], mm, reinterpret_cast<int*>(range_flag), st, sp ? sp + 1344 : nullptr,
                            head_gl ? 1 : 0);
      if (rc != MAGAT_OK) return rc;
      cur = 2; hin = Ho; win = Wo; lstart = 3; pooled_in = true;
    } else if (fused1 && d->chain_off > 0 && Ho == 6 && Wo == 6 && nblocks >= 2 && magat_opt(MAGAT_OPT_BLOCK_FUSED)) {
      rc = magat_block_chain(buf[1], buf[0], buf[2], nblocks == 2 ? 0 : 2, pixs(64), tiles(Ho * Wo, 64), pk + d->chain_off,
                             pk + d->off[5], pk + d->off[7], pk + d->off[9], mm, reinterpret_cast<int*>(range_flag), st);
      if (rc != MAGAT_OK) return rc;
      cur = 2; hin = Ho; win = Wo; lstart = 2;
      // ... and layer3 + ReLU + AvgPool2d(2) as one more launch: the head then reads 9 pooled cells instead of 36 pixels
      if (nblocks == 3 && d->chain3_off > 0) {
        rc = magat_block3(buf[2], buf[0], pk + d->chain3_off, pk + d->off[11], pk + d->off[13], mm,
                          reinterpret_cast<int*>(range_flag), st);
        if (rc != MAGAT_OK) return rc;
        cur = 0; lstart = 3; pooled_in = true;
      }
    }
    for (int l = lstart; l < nblocks; ++l) {
      const BlockShape s = shapes[l];
      const int hout = s.stride == 2 ? Ho : hin, wout = s.stride == 2 ? Wo : win;
      const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
      magat_conv_gemm_desc g = {};
      // conv1 + bn1 + relu
      g.in = buf[cur]; g.wt = pk + d->off[2 + 4 * l]; g.bias = pk + d->off[3 + 4 * l]; g.out = buf[mid];
      g.in_pix_stride = pixs(s.cin); g.out_pix_stride = pixs(s.cout);
      g.in_tile_stride = tiles(hin * win, s.cin); g.out_tile_stride = tiles(hout * wout, s.cout);
      g.M = mm; g.Cin = s.cin; g.lda = s.cin; g.Hin = hin; g.Win = win; g.kH = g.kW = 3; g.stride = s.stride;
      g.pad = 1; g.Hout = hout; g.Wout = wout; g.Cout = s.cout; g.ldc = s.cout; g.relu = 1;
      g.tag = tagof(MAGAT_TAG_BLOCK_CONV + 2 * l);
      g.range_flag = range_flag; g.run_if = run_if;
      if (absmax) g.absmax = absmax + 1 + 2 * l;
      if (split >> l & 1) {      // split-MFMA kernel: float32 activations split by its loader, pre-split weights
        g.in_fmt = 4; g.wt = pk + d->off[24 + 2 * l];
      }
      g.in_gl = g.out_gl = lay;
      if (lay == 2) g.wt += permuted(s.cout, 9 * s.cin);
      if (!(fused1 && l == 0)) {
        rc = run_or_chain(g);
        if (rc != MAGAT_OK) return rc;
      }
      // conv2 + bn2 + (1x1 strided downsample + bn) + relu
      magat_conv_gemm_desc h = {};
      h.in = buf[mid]; h.in2 = buf[cur]; h.wt = pk + d->off[4 + 4 * l]; h.bias = pk + d->off[5 + 4 * l];
      h.out = buf[nxt];
      h.in_pix_stride = pixs(s.cout); h.in2_pix_stride = pixs(s.cin); h.out_pix_stride = pixs(s.cout);
      h.in_tile_stride = tiles(hout * wout, s.cout); h.in2_tile_stride = tiles(hin * win, s.cin);
      h.out_tile_stride = tiles(hout * wout, s.cout);
      h.M = mm; h.Cin = s.cout; h.lda = s.cout; h.Hin = hout; h.Win = wout; h.kH = h.kW = 3; h.stride = 1; h.pad = 1;
      h.Hout = hout; h.Wout = wout; h.C2 = s.cin; h.lda2 = s.cin; h.W2 = win; h.stride2 = s.stride;
      h.Cout = s.cout; h.ldc = s.cout; h.relu = 1;
      h.tag = tagof(MAGAT_TAG_BLOCK_CONV + 2 * l + 1);
      h.range_flag = range_flag; h.run_if = run_if;
      if (absmax) h.absmax = absmax + 2 + 2 * l;
      if (split >> l & 1) {
        h.in_fmt = 4; h.wt = pk + d->off[25 + 2 * l];
      }
      h.in_gl = lay; h.out_gl = l + 1 < nblocks ? lay : 0;
      if (lay == 2) h.wt += permuted(s.cout, 9 * s.cout + s.cin);
      if (fused1 && l == 0) {    // the residual branch reads the stem's stride-2 pixels, stored as an Ho x Wo map
        h.in2_tile_stride = tiles(hout * wout, s.cin); h.W2 = wout; h.stride2 = 1;
      }
      rc = run_or_chain(h);
      if (rc != MAGAT_OK) return rc;
      cur = nxt; hin = hout; win = wout;
    }
    // head: AvgPool2d(2) (sum-pool on load, 1/4 in the weights) + fc(+Flatten+Linear) folded into one
    // (hin/2 x win/2) valid conv over the pooled map -> [mm][n_feat]
    const int clast = shapes[nblocks - 1].cout;
    magat_conv_gemm_desc g = {};
    g.in = buf[cur]; g.wt = pk + d->off[14]; g.bias = pk + d->off[15];
    g.out = feat + (size_t)m0 * ldfeat;
    g.in_pix_stride = pixs(clast); g.in_tile_stride = tiles(pooled_in ? (hin / 2) * (win / 2) : hin * win, clast);
    g.M = mm; g.Cin = clast; g.lda = clast; g.Hin = hin / 2; g.Win = win / 2;
    g.kH = hin / 2; g.kW = win / 2; g.stride = 1; g.pad = 0; g.Hout = g.Wout = 1; g.Cout = d->n_feat; g.ldc = ldfeat;
    g.relu = 0;
    if (!pooled_in) { g.pool = 1; g.pool_w = win; }
    g.tag = tagof(MAGAT_TAG_HEAD);
    g.run_if = run_if;
    if (absmax) g.absmax = absmax + 7;
    // Few agents (the closed-loop batch-1 step): one workgroup per 64 agents would walk all (hin/2)(win/2) clast of K alone
    // (83 us at 100 agents).  Split K by pooled cell instead: every cell is its own 1x1 "output pixel" with its slice of the
    // weight rows (wt_pix_stride / ldw), the partial products land in a free map buffer, a small kernel sums them in
    // a fixed order and adds the bias.  Option HEAD_SPLITK = largest agent count that takes this form (0 = never) - the
    // count of the whole call (M), not of the chunk: the short last chunk of a 70 100-agent batch must sum in the same
    // order as its other chunks, or the batch would differ in the last bit from the same agents presented as shards.
    const int cells = (hin / 2) * (win / 2);
    const int split_max = magat_opt(MAGAT_OPT_HEAD_SPLITK);
    if (head_done) {
      compress_done = true;
    } else if (!absmax && !chained && cells > 1 && Mform <= split_max && (clast & 3) == 0 && (d->n_feat & 3) == 0 &&
        (size_t)cells * d->n_feat <= enc_buf_floats_per_agent(d)) {     // the partials must fit one map buffer
      float* part = buf[(cur + 1) % 3];                 // [cells][mm][n_feat]
      if (!rerun) magat_form_note(MAGAT_FORM_HEAD_SPLITK);
      g.out = part; g.bias = nullptr; g.ldc = d->n_feat;
      g.out_pix_stride = (long long)mm * d->n_feat;
      g.kH = g.kW = 1; g.Hout = hin / 2; g.Wout = win / 2;
      g.wt_pix_stride = clast; g.ldw = cells * clast;
      const int pid = magat_prof_begin(tagof(MAGAT_TAG_HEAD), st);
      g.tag = MAGAT_TAG_UNTAGGED;
      rc = magat_conv_gemm_f32(&g, stream);
      if (rc == MAGAT_OK) {
        const long long total4 = (long long)mm * (d->n_feat / 4);
        hipLaunchKernelGGL(head_sum_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, part,
                           pk + d->off[15], feat + (size_t)m0 * ldfeat, ldfeat, mm, d->n_feat, cells,
                           reinterpret_cast<const int*>(run_if));
        if (hipGetLastError() != hipSuccess) rc = MAGAT_ERR_LAUNCH;
      }
      magat_prof_end(pid, st);
    } else {
      // many agents, pooled input (no pooling on load): a plain valid convolution - f16x3 split products like the blocks
      if (pooled_in && split && d->head16_off > 0 && (clast % 32) == 0 && (d->n_feat % 32) == 0 &&
          magat_opt(MAGAT_OPT_HEAD_F16)) {
        g.in_fmt = 4; g.wt = pk + d->head16_off; g.range_flag = range_flag; g.run_if = nullptr;
        if (d->scaled_off > 0) g.in_scale = pk + d->scaled_off + 1349;
        if (head_gl) g.in_gl = 1;
        magat_form_note(MAGAT_FORM_HEAD_LONGK);
        // compressMLP in the head's epilogue (round 5; option HEAD_COMPRESS): the 128-wide feature rows are complete in the
        // workgroup's registers, so the second layer runs on them there - one launch instead of two, bit for bit the same
        // comp.  Declined (MAGAT_ERR_UNSUPPORTED, nothing launched) when the head's column tile was narrowed for a small
        // batch: the two launches follow as before.
        if (!rerun && comp && d->n_comp == 128 && d->n_feat == 128 && d->comp16_off > 0 && magat_opt(MAGAT_OPT_HEAD_COMPRESS)) {
          magat_conv_gemm_desc f = g;
          f.wt2 = pk + d->comp16_off; f.bias2 = pk + d->off[17]; f.out2 = comp + (size_t)m0 * ldcomp;
          f.Cout2 = d->n_comp; f.ldc2 = ldcomp; f.relu2 = 1;
          f.in_scale2 = d->scaled_off > 0 ? pk + d->scaled_off + 1350 : nullptr;
          if (comp16) {      // (ABI 7: the bf16 rows of the bf16-storage graph layer from the same epilogue)
            f.out2_bf16 = comp16 + (size_t)m0 * d->n_comp;
            f.ldc2_bf16 = d->n_comp;
          }
          rc = magat_conv_gemm_f32(&f, stream);
          if (rc == MAGAT_OK) { compress_done = true; comp16_done = comp16 != nullptr; }
          else if (rc != MAGAT_ERR_UNSUPPORTED) return rc;
        }
      }
      if (!compress_done) rc = g.in_fmt == 0 ? run_or_chain(g) : magat_conv_gemm_f32(&g, stream);
    }
    if (rc != MAGAT_OK) return rc;
    if (d->n_comp > 0 && !compress_done) {
      // compressMLP: f16x3 split products on the direct kernel when the head ran that way (large pooled batches), else
      // float32 MFMA (and always in the guard's re-run)
      if (!rerun && pooled_in && split && d->comp16_off > 0 && (d->n_feat % 32) == 0 && (d->n_comp % 32) == 0 &&
          magat_opt(MAGAT_OPT_HEAD_F16)) {
        magat_conv_gemm_desc c = {};
        c.tag = tagof(MAGAT_TAG_COMPRESS);
        c.in = feat + (size_t)m0 * ldfeat; c.wt = pk + d->comp16_off; c.bias = pk + d->off[17];
        c.out = comp + (size_t)m0 * ldcomp;
        c.M = mm; c.Cin = d->n_feat; c.lda = ldfeat; c.Hin = c.Win = 1; c.kH = c.kW = 1; c.stride = 1; c.pad = 0;
        c.Hout = c.Wout = 1; c.Cout = d->n_comp; c.ldc = ldcomp; c.relu = 1;
        c.in_fmt = 4; c.range_flag = range_flag;
        if (d->scaled_off > 0) c.in_scale = pk + d->scaled_off + 1350;
        rc = magat_conv_gemm_f32(&c, stream);
      } else if (chained) {
        magat_conv_gemm_desc c = {};
        c.in = feat + (size_t)m0 * ldfeat; c.wt = pk + d->off[16]; c.bias = pk + d->off[17]; c.out = comp + (size_t)m0 * ldcomp;
        c.M = mm; c.Cin = d->n_feat; c.lda = ldfeat; c.Hin = c.Win = 1; c.kH = c.kW = 1; c.stride = 1; c.pad = 0;
        c.Hout = c.Wout = 1; c.Cout = d->n_comp; c.ldc = ldcomp; c.relu = 1;
        rc = run_or_chain(c);
      } else {
        rc = enc_linear(feat + (size_t)m0 * ldfeat, ldfeat, pk + d->off[16], pk + d->off[17], comp + (size_t)m0 * ldcomp,
                        ldcomp, mm, d->n_comp, d->n_feat, 1, tagof(MAGAT_TAG_COMPRESS), run_if, stream, absmax ? absmax + 8 : nullptr);
      }
      if (rc != MAGAT_OK) return rc;
    }
    if (comp16 && !comp16_done && !chained && d->n_comp > 0 && (d->n_comp & 3) == 0) {
      // comp came from a launch of its own (small batches, float32 forms): one cast pass over this chunk's rows
      rc = magat_cast_rows(comp + (size_t)m0 * ldcomp, comp16 + (size_t)m0 * d->n_comp, 1, mm, d->n_comp, ldcomp, d->n_comp, stream);
      if (rc != MAGAT_OK) return rc;
    }
    if (nchain > 0) {
      // (the last chunk's chained launch is the last reader of the guard's flag: its last workgroup does the bookkeeping)
      const bool last = book && m0 + mc >= M;
      rc = magat_conv_gemm_chain_f32(chain, nchain, run_if, MAGAT_TAG_UNTAGGED, st, last ? book : nullptr);
      if (rc != MAGAT_OK) return rc;
      if (last && booked) *booked = true;
    }
  }
  return MAGAT_OK;
}

extern "C" int magat_encoder_forward_f32(const magat_encoder_desc* d, const float* x, float* feat, int ldfeat,
                                         float* comp, int ldcomp, void* workspace, size_t workspace_bytes, int M,
                                         void* stream) {
  if (!d || !x || !feat || !d->pack) return MAGAT_ERR_NULL;
  if (M <= 0 || d->H < 3 || d->W < 3 || d->n_feat <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (d->variant < 0 || d->variant > 4) return MAGAT_ERR_UNSUPPORTED;
  if (d->n_comp > 0 && !comp) return MAGAT_ERR_NULL;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) ||
      workspace_bytes < magat_encoder_workspace_bytes(d, M))
    return MAGAT_ERR_WORKSPACE;
  const int H = d->H, W = d->W;
  const int mc = enc_chunk_agents(M);
  const int mcp = (mc + MAGAT_TILE_ROWS - 1) / MAGAT_TILE_ROWS * MAGAT_TILE_ROWS;
  const size_t bstride = magat_align_up(enc_buf_floats_per_agent(d) * (size_t)mcp * sizeof(float), 256) / sizeof(float);
  int32_t* status = static_cast<int32_t*>(workspace);
  float* bufbase = reinterpret_cast<float*>(static_cast<char*>(workspace) + ENC_STATUS_BYTES);
  auto pixs = [](int c) { return (int64_t)MAGAT_TILE_ROWS * c; };
  auto tiles = [](int npix, int c) { return (int64_t)npix * MAGAT_TILE_ROWS * c; };
  float* buf[3] = {bufbase, bufbase + bstride, bufbase + 2 * bstride};
  const float* pk = d->pack;
  hipStream_t st = static_cast<hipStream_t>(stream);

  if (d->variant >= 2) {
    // Plain CNNs (float32 kernels only): conv-BN-ReLU stacks with MaxPool2d(2) behind some layers, the pool folded into the NEXT
    // layer's loader (or, behind the last layer, into a pooled 1x1 GEMM with identity weights).
    //   2: CNN_mode Default - 5 layers, pools behind layers 0, 2, 4;
    //   3 / 4 (ABI 8): the dilated CNNs of DecentralPlannerNet (use_dilated_version 1 / 2; decentralplanner.py:57-86, 138-162):
    //          5 (4) layers with dilation = padding = 1 3 1 3 (1), pools behind layers 1 and 3
    const int nl = d->variant == 4 ? 4 : 5;
    const int chans[6] = {3, 32, 32, 64, 64, 128};
    const int dils[5] = {1, d->variant == 2 ? 1 : 3, 1, d->variant == 2 ? 1 : 3, 1};
    const bool pool_after[5] = {d->variant == 2, d->variant != 2, d->variant == 2, d->variant != 2, d->variant == 2};
    const int clast = chans[nl];
    if (d->n_feat <= 0 || d->n_feat % clast) return MAGAT_ERR_BAD_SHAPE;
    for (int m0 = 0; m0 < M; m0 += mc) {
      const int mm = (M - m0) < mc ? (M - m0) : mc;
      int rc = magat_conv_first_tiled_f32(x + (size_t)m0 * 3 * H * W, pk + d->off[0], pk + d->off[1], buf[0], mm, H, W,
                                          stream);
      if (rc != MAGAT_OK) return rc;
      int cur = 0, hp = H, wp = W;          // physical size of the map in buf[cur]
      const bool last_pooled = pool_after[nl - 1];
      for (int l = 1; l < nl; ++l) {
        const bool pooled = pool_after[l - 1];                // the previous layer was followed by a pool
        const int hin = pooled ? hp / 2 : hp, win = pooled ? wp / 2 : wp;
        const bool to_feat = l == nl - 1 && !last_pooled;      // the last map IS the feature row: [cell][channel]
        magat_conv_gemm_desc g = {};
        g.in = buf[cur]; g.wt = pk + d->off[2 + 2 * (l - 1)]; g.bias = pk + d->off[3 + 2 * (l - 1)];
        g.out = to_feat ? feat + (size_t)m0 * ldfeat : buf[(cur + 1) % 3];
        g.in_pix_stride = pixs(chans[l]); g.out_pix_stride = to_feat ? chans[l + 1] : pixs(chans[l + 1]);
        g.in_tile_stride = tiles(hp * wp, chans[l]); g.out_tile_stride = to_feat ? 0 : tiles(hin * win, chans[l + 1]);
        g.M = mm; g.Cin = chans[l]; g.lda = chans[l]; g.Hin = hin; g.Win = win; g.kH = g.kW = 3; g.stride = 1;
        g.pad = dils[l]; g.dilation = dils[l];
        g.Hout = hin; g.Wout = win; g.Cout = chans[l + 1]; g.ldc = to_feat ? ldfeat : chans[l + 1]; g.relu = 1;
        g.tag = MAGAT_TAG_BLOCK_CONV + (l - 1);
        if (pooled) { g.pool = 2; g.pool_w = wp; }
        rc = magat_conv_gemm_f32(&g, stream);
        if (rc != MAGAT_OK) return rc;
        cur = (cur + 1) % 3; hp = hin; wp = win;
      }
      const int hf = last_pooled ? hp / 2 : hp, wf = last_pooled ? wp / 2 : wp;
      if (hf < 1 || wf < 1 || d->n_feat != clast * hf * wf) return MAGAT_ERR_BAD_SHAPE;
      if (last_pooled) {
        // final MaxPool2d(2) -> feat [mm][hf wf][clast]: a pooled 1x1 GEMM with identity weights, one output pixel per pooled
        // cell.  At the reference's FOV = 9 (11 x 11 maps) the Default CNN ends on one cell; at other map sizes (and for the
        // dilated variants) the features are (cell, channel)-ordered here where the reference's Flatten is (channel, cell)-
        // ordered - encoder.fold_default_cnn / fold_dilated_cnn permute the columns of every weight that reads them
        magat_conv_gemm_desc g = {};
        g.in = buf[cur]; g.wt = pk + d->off[14]; g.out = feat + (size_t)m0 * ldfeat;
        g.in_pix_stride = pixs(clast); g.in_tile_stride = tiles(hp * wp, clast);
        g.M = mm; g.Cin = clast; g.lda = clast; g.Hin = hf; g.Win = wf;
        g.kH = 1; g.kW = 1; g.stride = 1; g.pad = 0; g.Hout = hf; g.Wout = wf; g.Cout = clast; g.ldc = ldfeat;
        g.out_pix_stride = clast;
        g.pool = 2; g.pool_w = wp; g.tag = MAGAT_TAG_HEAD;
        rc = magat_conv_gemm_f32(&g, stream);
        if (rc != MAGAT_OK) return rc;
      }
      if (d->n_comp > 0) {
        rc = magat_linear_tagged_f32(feat + (size_t)m0 * ldfeat, ldfeat, pk + d->off[16], pk + d->off[17],
                                     comp + (size_t)m0 * ldcomp, ldcomp, mm, d->n_comp, d->n_feat, 1,
                                     MAGAT_TAG_COMPRESS, stream);
        if (rc != MAGAT_OK) return rc;
      }
    }
    if (d->comp_bf16 && d->n_comp > 0 && (d->n_comp & 3) == 0)
      return magat_cast_rows(comp, d->comp_bf16, 1, M, d->n_comp, ldcomp, d->n_comp, stream);
    return MAGAT_OK;
  }

  // ResNet encoders.  With the range guard on (default) the split-arithmetic pass reports clamps into status[0] and a
  // float32-MFMA pass of the whole encoder follows in the same stream, every launch of it predicated on that flag: when no
  // value left the f16 planes' range (always, for sane checkpoints) those launches return at once; when one did, feat / comp
  // are recomputed in true fp32 before anything downstream reads them.
  const int split = enc_split_mask(d, magat_opt(MAGAT_OPT_CONV_SPLIT));
  const bool guard = split != 0 && magat_opt(MAGAT_OPT_RANGE_GUARD) != 0;
  bool self_guarded = false;      // the latency form guarded itself (block_lat.hip): nothing to re-run, nothing to book
  int rc = enc_run_resnet(d, x, feat, ldfeat, comp, ldcomp, bufbase, M, stream, split, guard ? status : nullptr, nullptr, nullptr,
                          nullptr, nullptr, &self_guarded);
  if (rc != MAGAT_OK || !guard) return rc;
  if (self_guarded) {
    if (d->comp_bf16 && d->n_comp > 0 && (d->n_comp & 3) == 0)
      rc = magat_cast_rows_if(comp, d->comp_bf16, 1, M, d->n_comp, ldcomp, d->n_comp, stream, status + 2);
    return rc;
  }
  const int pid = magat_prof_begin(MAGAT_TAG_RANGE_GUARD, st);
  bool booked = false;
  rc = enc_run_resnet(d, x, feat, ldfeat, comp, ldcomp, bufbase, M, stream, 0, nullptr, status, nullptr, status, &booked);
  if (rc == MAGAT_OK && !booked) {
    hipLaunchKernelGGL(guard_count_kernel, dim3(1), dim3(1), 0, st, status);
    if (hipGetLastError() != hipSuccess) rc = MAGAT_ERR_LAUNCH;
  }
  // the bf16 rows follow a re-run (status[2] = this forward's flag by now): a launch that returns at once otherwise
  if (rc == MAGAT_OK && d->comp_bf16 && d->n_comp > 0 && (d->n_comp & 3) == 0)
    rc = magat_cast_rows_if(comp, d->comp_bf16, 1, M, d->n_comp, ldcomp, d->n_comp, stream, status + 2);
  magat_prof_end(pid, st);
  return rc;
}

extern "C" int magat_encoder_calibrate_f32(const magat_encoder_desc* d, const float* x, float* feat, int ldfeat, float* comp,
                                           int ldcomp, void* workspace, size_t workspace_bytes, int M, float* absmax,
                                           void* stream) {
  if (!d || !x || !feat || !d->pack || !absmax) return MAGAT_ERR_NULL;
  if (M <= 0 || d->H < 3 || d->W < 3 || d->n_feat <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (d->variant != 0 && d->variant != 1) return MAGAT_ERR_UNSUPPORTED;      // (the Default CNN runs in float32: nothing to scale)
  if (d->n_comp > 0 && !comp) return MAGAT_ERR_NULL;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < magat_encoder_workspace_bytes(d, M))
    return MAGAT_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(absmax, 0, 16 * sizeof(float), st) != hipSuccess) return MAGAT_ERR_LAUNCH;
  float* bufbase = reinterpret_cast<float*>(static_cast<char*>(workspace) + ENC_STATUS_BYTES);
  return enc_run_resnet(d, x, feat, ldfeat, comp, ldcomp, bufbase, M, stream, 0, nullptr, nullptr, absmax);
}
