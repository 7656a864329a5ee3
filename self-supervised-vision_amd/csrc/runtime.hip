// Library runtime: error string, version, per-class HIP-event timing, fill.
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <string.h>

static thread_local char g_err[512] = "";

void ssv_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ssv_version(void) { return 120; }   // 1.2x: round 6, ssv_conv_desc carries the arithmetic (SSV_ARITH_BF16X3) and the pre-split weight planes; earlier: 1.1x round 5 entry points
extern "C" const char* ssv_last_error(void) { return g_err; }
#ifndef SSV_SRC_SHA16
#define SSV_SRC_SHA16 "unknown"
#endif
extern "C" const char* ssv_source_sha16(void) { return SSV_SRC_SHA16; }

extern "C" int ssv_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return prop.multiProcessorCount;
}

// ---- profiling ------------------------------------------------------------------------------
// The ONLY process-wide mutable state of the library besides the thread-local error string, and off by default: when disabled an
// entry point pays one relaxed atomic load.  When enabled, the record list is guarded by a mutex, and the begin / end pair of one
// launch is matched through a thread-local slot, so concurrent callers (one host thread per stream) do not corrupt each other.
struct ProfRec { int cls; hipEvent_t a, b; bool closed; };
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;          // records of the current collection window (under g_prof_mu)
static std::vector<ProfRec> g_free;          // recycled event pairs (under g_prof_mu)
static uint64_t g_gen = 0;                   // collection window number (under g_prof_mu): bumped whenever g_recs is recycled
// this thread's record between begin and end: its index AND the window it belongs to - a collect / reset from another thread in
// between recycles the list, and a stale index must not close somebody else's record
static thread_local int g_open = -1;
static thread_local uint64_t g_open_gen = 0;

void ssv_prof_begin(int cls, hipStream_t s) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfRec r;
  if (!g_free.empty()) { r = g_free.back(); g_free.pop_back(); }
  else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
  r.cls = cls;
  r.closed = false;
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  g_open = (int)g_recs.size() - 1;
  g_open_gen = g_gen;
}
void ssv_prof_end(int cls, hipStream_t s) {
  if (g_open < 0) return;
  (void)cls;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (g_open_gen == g_gen && g_open < (int)g_recs.size()) {
    (void)hipEventRecord(g_recs[g_open].b, s);
    g_recs[g_open].closed = true;
  }
  g_open = -1;
}
extern "C" int ssv_prof_enable(int on) { g_prof_on.store(on != 0, std::memory_order_relaxed); return SSV_OK; }
static int prof_reset_locked() {
  for (auto& r : g_recs) g_free.push_back(r);
  g_recs.clear();
  g_gen += 1;
  return SSV_OK;
}
extern "C" int ssv_prof_reset(void) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_open = -1;
  return prof_reset_locked();
}
extern "C" int ssv_prof_collect(double* ms, int64_t* n) {
  SSV_REQUIRE(ms && n, "ssv_prof_collect: null pointer");
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (int i = 0; i < SSV_PROF_NCLASS; ++i) { ms[i] = 0.0; n[i] = 0; }
  for (auto& r : g_recs) {
    if (!r.closed) continue;                 // begun on another thread and not ended yet: its end event was never recorded
    if (hipEventSynchronize(r.b) != hipSuccess) SSV_FAIL(SSV_ERR_LAUNCH, "prof: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) SSV_FAIL(SSV_ERR_LAUNCH, "prof: elapsed failed");
    ms[r.cls] += t; n[r.cls] += 1;
  }
  g_open = -1;
  return prof_reset_locked();
}

// ---- fill -----------------------------------------------------------------------------------
__global__ void fill_k(int64_t n, float* p, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  f32x4 v4 = {v, v, v, v};
  for (int64_t j = i; j < n4; j += stride) reinterpret_cast<f32x4*>(p)[j] = v4;
  for (int64_t j = (n4 << 2) + i; j < n; j += stride) p[j] = v;
}
extern "C" int ssv_fill(int64_t n, float* p, float value, void* stream) {
  if (n <= 0) return SSV_OK;
  SSV_REQUIRE(p != nullptr && ((uintptr_t)p & 15) == 0, "ssv_fill: null or unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  int64_t blocks = cdiv64(n >> 2 ? n >> 2 : n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_k, dim3((unsigned)blocks), dim3(256), 0, s, n, p, value);
  SSV_CHECK_LAUNCH("ssv_fill");
  return SSV_OK;
}
