// Optional per-kernel timing hooks (bench.py's roofline leg): when enabled, every kernel launch made by
// the library is bracketed by hipEvents recorded on the launch stream; magat_profile_collect() (after the
// caller has synchronised) folds them into per-tag totals.  Disabled by default: zero cost on the hot path.
#include <mutex>
#include <vector>

#include "magat_common.h"

namespace {
struct Span {
  int tag;
  hipEvent_t a, b;
};
bool g_enabled = false;
std::vector<Span> g_pool;   // created lazily, reused
size_t g_used = 0;
double g_total[MAGAT_PROF_TAGS];
long long g_count[MAGAT_PROF_TAGS];
std::mutex g_mu;
constexpr size_t kMaxSpans = 1 << 16;
}  // namespace

int magat_prof_begin(int tag, hipStream_t st) {
  if (!g_enabled || tag == MAGAT_TAG_UNTAGGED) return -1;      // untagged launches are not timed (a tagged span may enclose them)
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_used >= kMaxSpans) return -1;
  if (g_used == g_pool.size()) {
    Span s;
    s.tag = 0;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
    g_pool.push_back(s);
  }
  const int id = (int)g_used++;
  g_pool[id].tag = tag;
  hipEventRecord(g_pool[id].a, st);
  return id;
}

void magat_prof_end(int id, hipStream_t st) {
  if (id < 0) return;
  hipEventRecord(g_pool[id].b, st);
}

// Pre-creates `spans` event pairs so that a timed region does not pay hipEventCreate (~5 us each) for its first use.
extern "C" int magat_profile_reserve(int spans) {
  std::lock_guard<std::mutex> lk(g_mu);
  while (g_pool.size() < (size_t)spans && g_pool.size() < kMaxSpans) {
    Span s;
    s.tag = 0;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return MAGAT_ERR_LAUNCH;
    g_pool.push_back(s);
  }
  return MAGAT_OK;
}

extern "C" int magat_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on != 0;
  return MAGAT_OK;
}

// Folds all finished spans into the per-tag totals.  Caller must have synchronised the streams.
extern "C" int magat_profile_collect(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < g_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_pool[i].a, g_pool[i].b) == hipSuccess) {
      const int t = g_pool[i].tag;
      if (t >= 0 && t < MAGAT_PROF_TAGS) {
        g_total[t] += ms;
        g_count[t] += 1;
      }
    }
  }
  g_used = 0;
  return MAGAT_OK;
}

extern "C" int magat_profile_read(int tag, long long* count, double* total_ms) {
  if (tag < 0 || tag >= MAGAT_PROF_TAGS || !count || !total_ms) return MAGAT_ERR_BAD_SHAPE;
  std::lock_guard<std::mutex> lk(g_mu);
  *count = g_count[tag];
  *total_ms = g_total[tag];
  return MAGAT_OK;
}

extern "C" int magat_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < MAGAT_PROF_TAGS; ++i) {
    g_total[i] = 0.0;
    g_count[i] = 0;
  }
  g_used = 0;
  return MAGAT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// What the chip SUSTAINS on v_mfma_f32_32x32x16_f16 with nothing else going on (registers only, operand bits toggling): the
// clock the chip holds under matrix load is part of the figure - on the MI355X boxes of this project it is 1.5-1.6 PFLOP/s,
// 62 % of the 2.5 PFLOP/s the 2.4 GHz peak clock would give (profiles/r01f/mfma_peak.txt; the chain kernel itself runs at
// 1.67-1.77 GHz, tools/chain_phase_probe.py).  bench.py reports a kernel's ISSUED matrix-core rate against this measured
// ceiling next to the nominal one.
namespace {
typedef _Float16 mp_f16x8 __attribute__((ext_vector_type(8)));
typedef float mp_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_sustained_kernel(float* out, int iters) {
  mp_f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  mp_f16x8 a, b;
  for (int i = 0; i < 8; ++i) {      // pseudo-random operand bits: realistic toggling (constant operands draw less power)
    unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 4096.f));
    b[i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 4096.f));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    if ((it & 63) == 63)
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] *= 1e-3f;
    a = -a;
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

// *tflops = the f16 dense matrix-core rate of one ~ms_target ms launch of the loop above on the current device (one wave per
// SIMD on every CU; synchronises the stream).  scratch: 256 * CUs floats.
extern "C" int magat_mfma_sustained_f16(double* tflops, float* scratch, int ms_target, void* stream) {
  if (!tflops || !scratch) return MAGAT_ERR_NULL;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MAGAT_ERR_LAUNCH;
  // 32 MFMAs of 32 cycles per iteration and SIMD: ~1700 iterations per ms at 1.75 GHz
  const int iters = (ms_target > 0 ? ms_target : 5) * 1700;
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(mfma_sustained_kernel, dim3((unsigned)cus), dim3(256), 0, st, scratch, iters);
  hipEventRecord(e1, st);
  int rc = magat_check_launch();
  float ms = 0.f;
  if (rc == MAGAT_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f))
    rc = MAGAT_ERR_LAUNCH;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc != MAGAT_OK) return rc;
  *tflops = (double)cus * 4 * iters * 32 * 32768.0 / ((double)ms * 1e9);
  return MAGAT_OK;
}
