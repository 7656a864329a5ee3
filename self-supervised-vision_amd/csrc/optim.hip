// Optimizer-side streaming kernels over the flat parameter arena (one launch per step, 20 B/param).
#include "common.h"

namespace {

// optim.SGD(momentum, nesterov=True, weight_decay): g' = g + wd*p; buf = g' (first step) or
// momentum*buf + g'; p -= lr*(g' + momentum*buf).
// hyper != NULL (ssv_sgd_nesterov_dev): lr, weight decay, momentum and the first-step flag are read from DEVICE memory - nothing in the launch changes when a
// schedule moves them, so one captured HIP graph of the step serves the whole run (ssv_amd/graph.py); the arithmetic is the same expression on the same floats.
template <bool TWO>
__global__ void __launch_bounds__(256)
sgd_nesterov_k(int64_t n, float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ g2, float* __restrict__ buf,
               float lr, float wd, float mom, int first, const float* __restrict__ hyper) {
  if (hyper) { lr = hyper[0]; wd = hyper[1]; mom = hyper[2]; first = hyper[3] != 0.f; }
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 graw = reinterpret_cast<const f32x4*>(g)[i];
    if constexpr (TWO) graw += reinterpret_cast<const f32x4*>(g2)[i];      // view-0 slab + view-1 slab, fixed order
    f32x4 gv = graw + wd * pv;
    f32x4 bv = first ? gv : mom * reinterpret_cast<f32x4*>(buf)[i] + gv;
    reinterpret_cast<f32x4*>(buf)[i] = bv;
    reinterpret_cast<f32x4*>(p)[i] = pv - lr * (gv + mom * bv);
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pv = p[i];
    const float gv = (TWO ? g[i] + g2[i] : g[i]) + wd * pv;
    const float bv = first ? gv : mom * buf[i] + gv;
    buf[i] = bv;
    p[i] = pv - lr * (gv + mom * bv);
  }
}

// optim.SGD(momentum, weight_decay, nesterov flag) - the plain form used by the linear probe (utils/eval_utils.py:42)
__global__ void __launch_bounds__(256)
sgd_k(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float lr, float wd, float mom, int nesterov, int first) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pv = p[i];
    const float gv = g[i] + wd * pv;
    const float bv = first ? gv : mom * buf[i] + gv;
    buf[i] = bv;
    p[i] = pv - lr * (nesterov ? gv + mom * bv : bv);
  }
}

__global__ void __launch_bounds__(256)
ema_k(int64_t n, float* __restrict__ t, const float* __restrict__ o, float tau) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float om = 1.0f - tau;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(t)[i] = tau * reinterpret_cast<f32x4*>(t)[i] + om * reinterpret_cast<const f32x4*>(o)[i];
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) t[i] = tau * t[i] + om * o[i];
}

__global__ void __launch_bounds__(256)
add_k(int64_t n, float* __restrict__ dst, const float* __restrict__ src) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<f32x4*>(dst)[i] + reinterpret_cast<const f32x4*>(src)[i];
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

unsigned grid_for(int64_t n) {
  int64_t b = cdiv64((n >> 2) ? (n >> 2) : n, 256);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ssv_sgd_nesterov(int64_t n, float* p, const float* g, const float* g2, float* buf, float lr, float weight_decay,
                                float momentum, int first_step, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && buf, "ssv_sgd_nesterov: bad arguments");
  SSV_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)g2 | (uintptr_t)buf) & 15) == 0, "ssv_sgd_nesterov: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  if (g2) hipLaunchKernelGGL((sgd_nesterov_k<true>), dim3(grid_for(n)), dim3(256), 0, s, n, p, g, g2, buf, lr, weight_decay, momentum, first_step, (const float*)nullptr);
  else    hipLaunchKernelGGL((sgd_nesterov_k<false>), dim3(grid_for(n)), dim3(256), 0, s, n, p, g, g2, buf, lr, weight_decay, momentum, first_step, (const float*)nullptr);
  SSV_CHECK_LAUNCH("ssv_sgd_nesterov");
  return SSV_OK;
}

extern "C" int ssv_sgd_nesterov_dev(int64_t n, float* p, const float* g, const float* g2, float* buf, const float* hyper, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && buf && hyper, "ssv_sgd_nesterov_dev: bad arguments");
  SSV_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)g2 | (uintptr_t)buf | (uintptr_t)hyper) & 15) == 0, "ssv_sgd_nesterov_dev: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  if (g2) hipLaunchKernelGGL((sgd_nesterov_k<true>), dim3(grid_for(n)), dim3(256), 0, s, n, p, g, g2, buf, 0.f, 0.f, 0.f, 0, hyper);
  else    hipLaunchKernelGGL((sgd_nesterov_k<false>), dim3(grid_for(n)), dim3(256), 0, s, n, p, g, g2, buf, 0.f, 0.f, 0.f, 0, hyper);
  SSV_CHECK_LAUNCH("ssv_sgd_nesterov_dev");
  return SSV_OK;
}

extern "C" int ssv_ema(int64_t n, float* target, const float* online, float tau, void* stream) {
  SSV_REQUIRE(n > 0 && target && online, "ssv_ema: bad arguments");
  SSV_REQUIRE((((uintptr_t)target | (uintptr_t)online) & 15) == 0, "ssv_ema: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  hipLaunchKernelGGL(ema_k, dim3(grid_for(n)), dim3(256), 0, s, n, target, online, tau);
  SSV_CHECK_LAUNCH("ssv_ema");
  return SSV_OK;
}

extern "C" int ssv_add(int64_t n, float* dst, const float* src, void* stream) {
  SSV_REQUIRE(n > 0 && dst && src, "ssv_add: bad arguments");
  SSV_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "ssv_add: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(add_k, dim3(grid_for(n)), dim3(256), 0, s, n, dst, src);
  SSV_CHECK_LAUNCH("ssv_add");
  return SSV_OK;
}

extern "C" int ssv_sgd(int64_t n, float* p, const float* g, float* buf, float lr, float weight_decay, float momentum, int nesterov,
                       int first_step, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && buf, "ssv_sgd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  hipLaunchKernelGGL(sgd_k, dim3(grid_for(n)), dim3(256), 0, s, n, p, g, buf, lr, weight_decay, momentum, nesterov, first_step);
  SSV_CHECK_LAUNCH("ssv_sgd");
  return SSV_OK;
}
