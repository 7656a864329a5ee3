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

// ---- nn.GELU() in its erf form (networks/vit.py:38, models/dino.py:30-33) --------------------------------------------------------
// erf in ~20 VALU instructions (the libm erff costs ~50 and branches): x * P5(x^2) for |x| < 1 (relative error 2e-7), 1 - u P4(u) exp(-x^2) with
// u = 1 / (1 + 0.932 |x|) beyond (Abramowitz-Stegun 7.1.26's form, coefficients re-fitted: 5e-10 in exact arithmetic); 1.3e-7 absolute in fp32 over
// all x (tools/probe/fast_erf_check.py) - the rounding class of 0.5 * v * (1 + erf(.)) whatever erf is used.  Every GELU of the library - the
// stand-alone kernels of vit.hip and the GEMM epilogues of conv_mfma.hip - goes through these, so fused and unfused forms agree bit for bit.
// -DSSV_LIBM_ERF=1 (diagnostic builds): the libm erff / expf.
#ifndef SSV_LIBM_ERF
#define SSV_LIBM_ERF 0
#endif
__device__ __forceinline__ float ssv_erf(float x) {
#if SSV_LIBM_ERF
  return erff(x);
#else
  const float z = x * x, t = __builtin_fabsf(x);
  float a = -0.0005631421809084713f;
  a = __builtin_fmaf(a, z, 0.004917551297694445f);
  a = __builtin_fmaf(a, z, -0.02671131119132042f);
  a = __builtin_fmaf(a, z, 0.11280179768800735f);
  a = __builtin_fmaf(a, z, -0.37612324953079224f);
  a = __builtin_fmaf(a, z, 1.1283791065216064f);
  a *= x;
  const float u = __builtin_amdgcn_rcpf(__builtin_fmaf(0.932012140750885f, t, 1.f));
  float b = 0.3885830342769623f;
  b = __builtin_fmaf(b, u, -0.9805029630661011f);
  b = __builtin_fmaf(b, u, 0.5967519283294678f);
  b = __builtin_fmaf(b, u, 0.47274520993232727f);
  b = __builtin_fmaf(b, u, 0.529606282711029f);
  const float e = __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
  b = __builtin_fmaf(-(b * u), e, 1.f);
  return t < 1.f ? a : __builtin_copysignf(b, x);
#endif
}
// standard normal cdf and pdf at v
__device__ __forceinline__ float ssv_norm_cdf(float v) { return 0.5f * (1.f + ssv_erf(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float ssv_norm_pdf(float v) {
#if SSV_LIBM_ERF
  return 0.39894228040143267794f * expf(-0.5f * v * v);
#else
  return 0.39894228040143267794f * __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f);
#endif
}
__device__ __forceinline__ float ssv_gelu(float v) { return 0.5f * v * (1.f + ssv_erf(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float ssv_gelu_grad(float v) { return ssv_norm_cdf(v) + v * ssv_norm_pdf(v); }
