// Optional per-kernel timing hooks (bench.py's roofline leg): when enabled, every kernel launch made by
// the library is bracketed by hipEvents recorded on the launch stream; magat_profile_collect() (after the
// caller has synchronised) folds them into per-tag totals.  Disabled by default: zero cost on the hot path.
#include <algorithm>
#include <atomic>
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
__global__ __launch_bounds__(256) void mfma_sustained_kernel(float* out, int iters, long long* stamps) {
  // core-clock counter next to the constant 100 MHz counter, first and last instruction of the wave: the clock the chip HELD
  const long long c0 = (long long)__builtin_readcyclecounter(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
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
  if (stamps && threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = (long long)__builtin_readcyclecounter() - c0;
    stamps[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime() - r0;
  }
}
}  // namespace

// *tflops = the f16 dense matrix-core rate of one ~ms_target ms launch of the loop above on the current device (one wave per
// SIMD on every CU; synchronises the stream).  scratch: 256 * CUs floats.
// (ABI 6) the same measurement with its own evidence: *clock_mhz = the core clock the chip held inside that launch (median over
// the CUs of core-clock cycles / 100 MHz ticks, both read by the kernel itself), *per_clk = flop per clock and SIMD the rate
// amounts to at that clock - 1024 is one v_mfma_f32_32x32x16_f16 (32768 flop) issued every 32 cycles, i.e. a matrix pipe that
// never idles: sustained = clock x per_clk x SIMDs, so "x of the sustained rate" is a statement about ISSUE, the gap to the
// nominal 2.5 PFLOP/s a statement about the clock.  scratch: 256 * CUs floats + 2 * CUs int64 behind them.
extern "C" int magat_mfma_sustained_f16_ex(double* tflops, double* clock_mhz, double* per_clk, float* scratch, int ms_target,
                                           void* stream);
extern "C" int magat_mfma_sustained_f16(double* tflops, float* scratch, int ms_target, void* stream) {
  return magat_mfma_sustained_f16_ex(tflops, nullptr, nullptr, scratch, ms_target, stream);
}
extern "C" int magat_mfma_sustained_f16_ex(double* tflops, double* clock_mhz, double* per_clk, float* scratch, int ms_target,
                                           void* stream) {
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
  // (the plain entry point's scratch holds the float results only: no stamps there)
  long long* stamps = (clock_mhz || per_clk) ? reinterpret_cast<long long*>(scratch + (size_t)256 * cus) : nullptr;
  hipLaunchKernelGGL(mfma_sustained_kernel, dim3((unsigned)cus), dim3(256), 0, st, scratch, iters, stamps);
  hipEventRecord(e1, st);
  int rc = magat_check_launch();
  float ms = 0.f;
  if (rc == MAGAT_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f))
    rc = MAGAT_ERR_LAUNCH;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc != MAGAT_OK) return rc;
  *tflops = (double)cus * 4 * iters * 32 * 32768.0 / ((double)ms * 1e9);
  if (stamps) {
    std::vector<long long> h(2 * (size_t)cus);
    if (hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return MAGAT_ERR_LAUNCH;
    std::vector<double> mhz;
    for (int i = 0; i < cus; ++i)
      if (h[2 * i + 1] > 0) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    if (mhz.empty()) return MAGAT_ERR_LAUNCH;
    std::sort(mhz.begin(), mhz.end());
    const double med = mhz[mhz.size() / 2];
    if (clock_mhz) *clock_mhz = med;
    if (per_clk) *per_clk = *tflops * 1e12 / (med * 1e6 * (double)cus * 4);
  }
  return MAGAT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Which FORM of a kernel a launch took (host-side counters, bumped where the launcher decides; tests assert on them next to the
// per-tag launch counts: the forms below exist only at benchmark sizes and must not be swapped silently).
namespace { std::atomic<long long> g_forms[MAGAT_FORMS]; }      // relaxed counters: no lock on the launch path
void magat_form_note(int id) {
  if (id >= 0 && id < MAGAT_FORMS) g_forms[id].fetch_add(1, std::memory_order_relaxed);
}
extern "C" long long magat_form_count(int id) {
  if (id < 0 || id >= MAGAT_FORMS) return -1;
  return g_forms[id].load(std::memory_order_relaxed);
}
extern "C" int magat_form_reset(void) {
  for (int i = 0; i < MAGAT_FORMS; ++i) g_forms[i].store(0, std::memory_order_relaxed);
  return MAGAT_OK;
}
