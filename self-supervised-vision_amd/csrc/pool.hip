// Pooling and layout kernels (HBM-streaming, float4 along the NHWC channel dimension).
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// MaxPool2d(kernel 3, stride 2, pad 1).  The window is scanned r-major; a strictly greater value
// (or a NaN) replaces the running max, so ties resolve to the first element like ATen.  argmax keeps
// the window slot r*3+s (0..8) per output element.
__global__ void __launch_bounds__(256)
maxpool_fwd_k(int N, int H, int W, int C, int Ho, int Wo, const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ am) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 - 1 + s;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 v = ld4(x + (((size_t)n * H + hi) * W + wi) * C + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first[e] || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; first[e] = false; }
      }
    }
    st4(y + (size_t)i * 4, best);
    uchar4 a; a.x = (uint8_t)bi[0]; a.y = (uint8_t)bi[1]; a.z = (uint8_t)bi[2]; a.w = (uint8_t)bi[3];
    reinterpret_cast<uchar4*>(am)[i] = a;
  }
}

// gather form of the backward (no atomics, deterministic): an input pixel (hi,wi) sits in at most
// 2x2 output windows; it receives dy of those whose argmax slot points back at it.
__global__ void __launch_bounds__(256)
maxpool_bwd_k(int N, int H, int W, int C, int Ho, int Wo, const float* __restrict__ dy, const uint8_t* __restrict__ am, float* __restrict__ dx) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int wi = (int)(t % W); t /= W;
    const int hi = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int th = hi + 1 - r;              // = 2*ho
      if (th < 0 || (th & 1)) continue;
      const int ho = th >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tw = wi + 1 - s;
        if (tw < 0 || (tw & 1)) continue;
        const int wo = tw >> 1;
        if (wo >= Wo) continue;
        const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C4 + c4;
        const uchar4 a = reinterpret_cast<const uchar4*>(am)[o];
        const f32x4 d = ld4(dy + o * 4);
        const int slot = r * 3 + s;
        if (a.x == slot) g[0] += d[0];
        if (a.y == slot) g[1] += d[1];
        if (a.z == slot) g[2] += d[2];
        if (a.w == slot) g[3] += d[3];
      }
    }
    st4(dx + (size_t)i * 4, g);
  }
}

__global__ void __launch_bounds__(256)
gap_fwd_k(int N, int HW, int C, const float* __restrict__ x, float* __restrict__ y) {
  const int C4 = C / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * C4) return;
  const int c4 = (int)(i % C4);
  const int n = (int)(i / C4);
  const float* p = x + (size_t)n * HW * C + 4 * c4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < HW; ++j) s += ld4(p + (size_t)j * C);
  st4(y + (size_t)i * 4, s / (float)HW);
}

__global__ void __launch_bounds__(256)
gap_bwd_k(int N, int HW, int C, const float* __restrict__ dy, float* __restrict__ dx) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * HW * C4;
  const float inv = 1.f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int n = (int)(i / ((int64_t)HW * C4));
    st4(dx + (size_t)i * 4, ld4(dy + ((size_t)n * C4 + c4) * 4) * inv);
  }
}

// NCHW <-> NHWC.  The only tensor that crosses the reference boundary in NCHW is the 3-channel
// image batch; a thread owns one pixel: C strided coalesced reads, one contiguous C-float write.
__global__ void __launch_bounds__(256)
nchw_to_nhwc_k(int N, int C, int HW, const float* __restrict__ in, float* __restrict__ out) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / HW, px = i - n * HW;
    const float* src = in + (size_t)n * C * HW + px;
    float* dst = out + (size_t)i * C;
    for (int c = 0; c < C; ++c) dst[c] = src[(size_t)c * HW];
  }
}
__global__ void __launch_bounds__(256)
nhwc_to_nchw_k(int N, int C, int HW, const float* __restrict__ in, float* __restrict__ out) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / HW, px = i - n * HW;
    const float* src = in + (size_t)i * C;
    float* dst = out + (size_t)n * C * HW + px;
    for (int c = 0; c < C; ++c) dst[(size_t)c * HW] = src[c];
  }
}

unsigned stream_grid(int64_t work) {
  int64_t b = cdiv64(work, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ssv_maxpool3x3s2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, float* y, uint8_t* argmax, void* stream) {
  SSV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ssv_maxpool3x3s2_fwd: bad shape (C %% 4 == 0 required)");
  SSV_REQUIRE(x && y && argmax, "ssv_maxpool3x3s2_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_POOL, s);
  hipLaunchKernelGGL(maxpool_fwd_k, dim3(stream_grid((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, s, N, H, W, C, Ho, Wo, x, y, argmax);
  SSV_CHECK_LAUNCH("ssv_maxpool3x3s2_fwd");
  return SSV_OK;
}

extern "C" int ssv_maxpool3x3s2_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dy, const uint8_t* argmax, float* dx, void* stream) {
  SSV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ssv_maxpool3x3s2_bwd: bad shape (C %% 4 == 0 required)");
  SSV_REQUIRE(dy && dx && argmax, "ssv_maxpool3x3s2_bwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_POOL, s);
  hipLaunchKernelGGL(maxpool_bwd_k, dim3(stream_grid((int64_t)N * H * W * (C / 4))), dim3(256), 0, s, N, H, W, C, Ho, Wo, dy, argmax, dx);
  SSV_CHECK_LAUNCH("ssv_maxpool3x3s2_bwd");
  return SSV_OK;
}

extern "C" int ssv_gap_fwd(int32_t N, int32_t HW, int32_t C, const float* x, float* y, void* stream) {
  SSV_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && x && y, "ssv_gap_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_POOL, s);
  hipLaunchKernelGGL(gap_fwd_k, dim3((unsigned)cdiv64((int64_t)N * (C / 4), 256)), dim3(256), 0, s, N, HW, C, x, y);
  SSV_CHECK_LAUNCH("ssv_gap_fwd");
  return SSV_OK;
}

extern "C" int ssv_gap_bwd(int32_t N, int32_t HW, int32_t C, const float* dy, float* dx, void* stream) {
  SSV_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && dy && dx, "ssv_gap_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_POOL, s);
  hipLaunchKernelGGL(gap_bwd_k, dim3(stream_grid((int64_t)N * HW * (C / 4))), dim3(256), 0, s, N, HW, C, dy, dx);
  SSV_CHECK_LAUNCH("ssv_gap_bwd");
  return SSV_OK;
}

namespace {

// wt[c][R-1-r][S-1-s][k] = w[k][r][s][c]: the filter bank of the convolution that computes a stride-1 dgrad (transposed in
// channels, rotated by 180 degrees); 32 x 32 tiles through LDS so both the read and the write are coalesced
__global__ void __launch_bounds__(256) filter_transpose_k(int K, int RS, int C, const float* __restrict__ w, float* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, c = c0 + tx;
    tile[i][tx] = (k < K && c < C) ? w[((int64_t)k * RS + tap) * C + c] : 0.f;
  }
  __syncthreads();
  const int tap_t = RS - 1 - tap;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, k = k0 + tx;
    if (c < C && k < K) wt[((int64_t)c * RS + tap_t) * K + k] = tile[tx][i];
  }
}

// Grouped convolution (ResNeXt, networks/resnet.py:8-10,57): the block-diagonal filter bank as a dense one, zeros off the diagonal.
// dense[k][tap][c] = (c / Cg == k / Kg) ? grouped[k][tap][c % Cg] : 0;   extract is the inverse gather (with optional +=).
__global__ void __launch_bounds__(256) group_expand_k(int64_t total, int RS, int Cg, int Kg, int groups, const float* __restrict__ wg, float* __restrict__ wd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int C = Cg * groups;
  const int c = (int)(i % C);
  const int64_t kt = i / C;                                   // k * RS + tap
  const int k = (int)(kt / RS);
  const int g = k / Kg;
  wd[i] = (c / Cg == g) ? wg[kt * Cg + (c - g * Cg)] : 0.f;
}
__global__ void __launch_bounds__(256) group_extract_k(int64_t total, int RS, int Cg, int Kg, int groups, const float* __restrict__ dwd, float* __restrict__ dwg, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over the grouped tensor
  if (i >= total) return;
  const int cg = (int)(i % Cg);
  const int64_t kt = i / Cg;
  const int k = (int)(kt / RS);
  const int g = k / Kg;
  const float v = dwd[kt * (int64_t)(Cg * groups) + g * Cg + cg];
  dwg[i] = accumulate ? dwg[i] + v : v;
}

// out[p][0..cout) = in[p][0..cin) followed by zeros (cout > cin): image batches and stem filters 3 -> 4 channels
__global__ void __launch_bounds__(256) pad_channels_k(int64_t npix, int cin, int cout, const float* __restrict__ in, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * cout) return;
  const int64_t pix = i / cout;
  const int c = (int)(i - pix * cout);
  out[i] = c < cin ? in[pix * cin + c] : 0.f;
}
// out[p][0..cout) (+)= in[p][0..cout) of a cin-wide tensor (cout < cin): the gradient of the padded stem filter back to 3 channels
__global__ void __launch_bounds__(256) unpad_channels_k(int64_t npix, int cin, int cout, const float* __restrict__ in, float* __restrict__ out, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * cout) return;
  const int64_t pix = i / cout;
  const int c = (int)(i - pix * cout);
  const float v = in[pix * cin + c];
  out[i] = accumulate ? out[i] + v : v;
}

}  // namespace

extern "C" int ssv_pad_channels(int64_t npix, int32_t cin, int32_t cout, const float* in, float* out, int32_t accumulate, void* stream) {
  SSV_REQUIRE(npix > 0 && cin > 0 && cout > 0 && cin != cout && in && out, "ssv_pad_channels: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int64_t total = npix * cout;
  if (cout > cin) hipLaunchKernelGGL(pad_channels_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, npix, cin, cout, in, out);
  else hipLaunchKernelGGL(unpad_channels_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, npix, cin, cout, in, out, accumulate);
  SSV_CHECK_LAUNCH("ssv_pad_channels");
  return SSV_OK;
}

extern "C" int ssv_group_expand(int32_t K, int32_t R, int32_t S, int32_t Cg, int32_t groups, const float* wg, float* wd, void* stream) {
  SSV_REQUIRE(K > 0 && R > 0 && S > 0 && Cg > 0 && groups > 0 && K % groups == 0 && wg && wd, "ssv_group_expand: bad arguments (K %% groups == 0)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int64_t total = (int64_t)K * R * S * Cg * groups;
  hipLaunchKernelGGL(group_expand_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, total, R * S, Cg, K / groups, groups, wg, wd);
  SSV_CHECK_LAUNCH("ssv_group_expand");
  return SSV_OK;
}

extern "C" int ssv_group_extract(int32_t K, int32_t R, int32_t S, int32_t Cg, int32_t groups, const float* dwd, float* dwg, int32_t accumulate, void* stream) {
  SSV_REQUIRE(K > 0 && R > 0 && S > 0 && Cg > 0 && groups > 0 && K % groups == 0 && dwd && dwg, "ssv_group_extract: bad arguments (K %% groups == 0)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int64_t total = (int64_t)K * R * S * Cg;
  hipLaunchKernelGGL(group_extract_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, total, R * S, Cg, K / groups, groups, dwd, dwg, accumulate);
  SSV_CHECK_LAUNCH("ssv_group_extract");
  return SSV_OK;
}

extern "C" int ssv_filter_transpose(int32_t K, int32_t R, int32_t S, int32_t C, const float* w, float* wt, void* stream) {
  SSV_REQUIRE(K > 0 && R > 0 && S > 0 && C > 0 && w && wt && w != wt, "ssv_filter_transpose: bad arguments");
  SSV_REQUIRE(R * S <= 65535 && cdiv(K, 32) <= 65535, "ssv_filter_transpose: filter too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(filter_transpose_k, dim3(cdiv(C, 32), cdiv(K, 32), R * S), dim3(256), 0, s, K, R * S, C, w, wt);
  SSV_CHECK_LAUNCH("ssv_filter_transpose");
  return SSV_OK;
}

extern "C" int ssv_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, const float* in, float* out, void* stream) {
  SSV_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && in && out, "ssv_nchw_to_nhwc: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(nchw_to_nhwc_k, dim3(stream_grid((int64_t)N * H * W)), dim3(256), 0, s, N, C, H * W, in, out);
  SSV_CHECK_LAUNCH("ssv_nchw_to_nhwc");
  return SSV_OK;
}

extern "C" int ssv_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float* in, float* out, void* stream) {
  SSV_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && in && out, "ssv_nhwc_to_nchw: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(nhwc_to_nchw_k, dim3(stream_grid((int64_t)N * H * W)), dim3(256), 0, s, N, C, H * W, in, out);
  SSV_CHECK_LAUNCH("ssv_nhwc_to_nchw");
  return SSV_OK;
}
