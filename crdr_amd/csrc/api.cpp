// Library-level entry points: version, architecture, thread-local error string.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "crdr_hip.h"

namespace crdr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace crdr

namespace crdr {
struct ProfRec { int kind; double flops; hipEvent_t e0, e1; };
static std::mutex g_pm;
static std::vector<ProfRec> g_recs;
static bool g_prof = false;
bool profile_on() { return g_prof; }
void* profile_begin(hipStream_t s) {
  if (!g_prof) return nullptr;
  hipEvent_t* ev = new hipEvent_t[2];
  if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) { delete[] ev; return nullptr; }
  (void)hipEventRecord(ev[0], s);
  return ev;
}
void profile_end(int kind, double flops, void* token, hipStream_t s) {
  if (!token) return;
  hipEvent_t* ev = static_cast<hipEvent_t*>(token);
  (void)hipEventRecord(ev[1], s);
  std::lock_guard<std::mutex> lk(g_pm);
  g_recs.push_back({kind, flops, ev[0], ev[1]});
  delete[] ev;
}
}  // namespace crdr

extern "C" void crdr_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(crdr::g_pm);
  crdr::g_prof = on != 0;
}
// sums over the recorded launches of `kind` (0 = conv forward / input-gradient, 1 = weight-gradient) and clears them
extern "C" int crdr_profile_read(int kind, double* flops, double* ms, long long* launches) {
  std::lock_guard<std::mutex> lk(crdr::g_pm);
  double f = 0, t = 0;
  long long n = 0;
  std::vector<crdr::ProfRec> keep;
  for (auto& r : crdr::g_recs) {
    if (r.kind != kind) { keep.push_back(r); continue; }
    (void)hipEventSynchronize(r.e1);
    float dt = 0.f;
    if (hipEventElapsedTime(&dt, r.e0, r.e1) == hipSuccess) { f += r.flops; t += dt; ++n; }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  crdr::g_recs.swap(keep);
  if (flops) *flops = f;
  if (ms) *ms = t;
  if (launches) *launches = n;
  return 0;
}

extern "C" const char* crdr_last_error(void) { return crdr::g_err; }
extern "C" int crdr_version(void) { return 600; }
extern "C" const char* crdr_arch(void) { return "gfx950"; }
