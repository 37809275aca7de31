// Library options: every tunable of libmagat_hip.so lives in ONE table that is filled from the environment once (first
// use) and can be changed at run time through the C ABI (magat_set_option) - the launch path never calls getenv.
// Also: the per-device cache of kernel attributes (hipFuncSetAttribute is per device) and the encoder status word.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "magat_common.h"

namespace {

struct OptEntry {
  const char* name;
  int def;
};

// order = enum MagatOpt (magat_common.h)
const OptEntry kOpts[MAGAT_OPT_COUNT] = {
    {"ENC_CHUNK", 65536},    // agents per encoder pass (workspace bound)
    {"CONV_SPLIT", 7},       // bit l: BasicBlock l+1 on the split-MFMA kernels (0 = fp32 MFMA everywhere: the strict-float32 form)
    {"CONV_PCHAIN", 1},      // f16 plane-granule activation chain between the BasicBlock convolutions
    {"L1_FUSED", 2},         // stem + layer1.conv1 as one kernel: 2 = eight-agent groups (stem8_kernel; 11 x 11 maps), 1 = 64-agent row
                             // bands (the form of every other map size), 0 = two launches
    {"HEAD_SPLITK", 5120},   // largest agent count for which the encoder head splits K by pooled cell (measured crossover against the
                             // f16x3 direct kernel at 4000-6000 agents); chosen on magat_encoder_desc.form_agents when that is set
    {"GAT_CHUNK_MB", 2048},  // cap of the two-launch graph layer's hoisted-map intermediate Z
    {"GAT_SPLIT", 1},        // GAT maps on the split-MFMA GEMM (0 = fp32 MFMA)
    {"RANGE_GUARD", 1},      // split-arithmetic range guard: overflow flag + stream-ordered fp32 re-run
    {"BLOCK_FUSED", 2},      // BasicBlock chain with the 6x6 maps of eight agents in LDS (four waves x 512 registers, static K walk):
                             // 2 = layer1.conv2 -> layer2 -> layer3 -> pool as ONE launch (block_full_p_kernel), 1 = two launches
                             // (layer1.conv2 + layer2; layer3 + pool), 0 = one launch per convolution
    {"CSR_TILED", 3},        // CSR path, N <= 1024: LDS-tiled kernels (bit 0 scores, bit 1 hops) instead of L2 gathers
    {"HEAD_F16", 1},         // encoder head on the f16x3 split kernel when its input is the pooled map of the layer3 kernel
    {"GAT_MFMA", 1},         // graph layer with 128 features, N <= 102, K = 2 | 3 as ONE launch of matrix-core products (gat_mfma.hip)
    {"SKINNY", 1},           // float32 1x1 layers with at most 8 outputs (the action head) as streamed dot products, not MFMA tiles
    {"GAT_PACK", 1},         // one-launch graph layer, N <= 32: four planning instances per pass (1: when the batch fills the chip
                             // that way; 2: always; 0: never)
    {"CONV_BNFILL", 256},    // f16x3 direct kernel: narrow the output-channel tile (128 -> 64 -> 32) until the launch has at
                             // least this many workgroups (0: never).  Column tiling does not touch any element's summation order:
                             // bit-identical
    {"HEAD_COMPRESS", 1},    // compressMLP computed in the encoder head's epilogue (one launch; needs the 128-column head tile:
                             // batches whose head tile was narrowed by CONV_BNFILL keep the two launches; bit-identical)
    {"CONV_TM", 2},          // direct kernel: 32-agent row groups per wave (2 = 256-agent tiles once they fill the chip; 1 = never:
                             // the reference form the 256-agent tiles are tested against, bit-identical)
    {"CSR_FUSED", 1},        // bf16-storage CSR layer, KeyQuery, K = 2, G = F = 128, concat: the maps on the matrix cores INSIDE the two
                             // graph kernels, hop on X (gat_csr_fused.hip); 0 = maps GEMM + tiled score / hop kernels
    {"LAT_AGENTS", 512},     // largest agent count (magat_encoder_desc.form_agents when set) that takes the LATENCY forms of the encoder:
                             // the BasicBlock chain with one agent per workgroup (block_lat.hip; bit-identical pooled map);
                             // 0 = never (the eight-agent-group kernels at every size)
    {"GAT_WIDE_FROM", 103},  // graph layer with 128 features: first agent count that takes the row-tile one-launch kernel
                             // (gat_mid.hip, X fragments in registers) instead of what follows gat_mfma.hip's N <= 102: the two-launch
                             // form up to 105 agents (0.36 against 0.25 ms per 512 instances), the CSR kernels above (0.76 against
                             // 0.27 ms at 128 agents).  Values below 103 mean 103, above 128 never
};

int g_val[MAGAT_OPT_COUNT];
int g_start[MAGAT_OPT_COUNT];      // the value the process STARTED with: the compile-time default, or the environment's
std::once_flag g_once;

void init_opts() {
  for (int i = 0; i < MAGAT_OPT_COUNT; ++i) {
    g_val[i] = kOpts[i].def;
    char key[64] = "MAGAT_";
    strncat(key, kOpts[i].name, sizeof(key) - 7);
    const char* e = getenv(key);
    if (e && *e) g_val[i] = atoi(e);
    g_start[i] = g_val[i];
  }
}

}  // namespace

int magat_opt(int id) {
  std::call_once(g_once, init_opts);
  return (id >= 0 && id < MAGAT_OPT_COUNT) ? g_val[id] : 0;
}

static int g_experiment_build = 0;
extern "C" int magat_experiment_mark(void) { g_experiment_build = 1; return 0; }
extern "C" int magat_build_flavor(void) { return g_experiment_build; }

extern "C" int magat_set_option(const char* name, int value) {
  if (!name) return MAGAT_ERR_NULL;
  std::call_once(g_once, init_opts);
  if (!strncmp(name, "MAGAT_", 6)) name += 6;
  for (int i = 0; i < MAGAT_OPT_COUNT; ++i)
    if (!strcmp(name, kOpts[i].name)) {
      g_val[i] = value;
      return MAGAT_OK;
    }
  return MAGAT_ERR_UNSUPPORTED;
}

extern "C" int magat_get_option(const char* name, int* value) {
  if (!name || !value) return MAGAT_ERR_NULL;
  std::call_once(g_once, init_opts);
  if (!strncmp(name, "MAGAT_", 6)) name += 6;
  for (int i = 0; i < MAGAT_OPT_COUNT; ++i)
    if (!strcmp(name, kOpts[i].name)) {
      *value = g_val[i];
      return MAGAT_OK;
    }
  return MAGAT_ERR_UNSUPPORTED;
}

extern "C" int magat_reset_option(const char* name) {
  if (!name) return MAGAT_ERR_NULL;
  std::call_once(g_once, init_opts);
  if (!strncmp(name, "MAGAT_", 6)) name += 6;
  for (int i = 0; i < MAGAT_OPT_COUNT; ++i)
    if (!strcmp(name, kOpts[i].name)) {
      g_val[i] = g_start[i];      // (the environment's value when MAGAT_<NAME> was set at start-up, not the compile-time default:
                                  //  a deployment's setting survives a test or tool that flips the option and resets it)
      return MAGAT_OK;
    }
  return MAGAT_ERR_UNSUPPORTED;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember, per (kernel slot, device), the
// largest size configured so far.  `slot` is a small integer the caller owns (one per kernel instantiation).
int magat_ensure_dyn_lds(const void* func, int slot, size_t bytes) {
  constexpr int kMaxDev = 16;
  static std::atomic<size_t> configured[MAGAT_LDS_SLOTS][kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MAGAT_ERR_LAUNCH;
  if (slot < 0 || slot >= MAGAT_LDS_SLOTS || dev < 0 || dev >= kMaxDev) {
    return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
               ? MAGAT_OK : MAGAT_ERR_LAUNCH;
  }
  if (bytes <= configured[slot][dev].load(std::memory_order_acquire)) return MAGAT_OK;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return MAGAT_ERR_LAUNCH;
  size_t cur = configured[slot][dev].load(std::memory_order_relaxed);
  while (cur < bytes && !configured[slot][dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {
  }
  return MAGAT_OK;
}
