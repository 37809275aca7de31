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
