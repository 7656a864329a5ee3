// Shared host/device helpers for libssv_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ssv_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error reporting ------------------------------------------------------------------------
void ssv_set_error(const char* fmt, ...);
#define SSV_FAIL(code, ...) do { ssv_set_error(__VA_ARGS__); return (code); } while (0)
#define SSV_REQUIRE(cond, ...) do { if (!(cond)) SSV_FAIL(SSV_ERR_INVALID, __VA_ARGS__); } while (0)
#define SSV_CHECK_LAUNCH(what) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) SSV_FAIL(SSV_ERR_LAUNCH, "%s: %s", (what), hipGetErrorString(e_)); } while (0)

// ---- per-class event timing (ssv_prof_*) ---------------------------------------------------
void ssv_prof_begin(int cls, hipStream_t s);
void ssv_prof_end(int cls, hipStream_t s);
struct ProfScope {
  int cls; hipStream_t s;
  ProfScope(int c, hipStream_t st) : cls(c), s(st) { ssv_prof_begin(cls, s); }
  ~ProfScope() { ssv_prof_end(cls, s); }
};

// ---- exact unsigned division by a runtime constant (n < 2^31) ----------------------------------
struct FastDiv { uint32_t mul, shr, d; };
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d ? d : 1;
  if (f.d == 1) { f.mul = 0; f.shr = 0; return f; }
  uint32_t lg = 31 - __builtin_clz(f.d);
  if (f.d & (f.d - 1)) lg += 1;                       // ceil(log2 d)
  uint32_t p = 31 + lg;
  f.mul = (uint32_t)((((uint64_t)1 << p) + f.d - 1) / f.d);
  f.shr = p - 32;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
