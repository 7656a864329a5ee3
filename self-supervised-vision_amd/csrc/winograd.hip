// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the deep stages (networks/resnet.py:7-10, 56-58: conv3x3, padding 1).
//
// These layers are MFMA-bound at exact fp32 (102 - 128 TFLOP/s of a 157 TFLOP/s pipe): the only way past that roof without changing the
// arithmetic type is fewer multiplies.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per
// (input channel, output channel) instead of 36:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        V = B^T d B   U = G g G^T   M[xi][nu] = sum_c V[xi][nu][c] U[xi][nu][c]
//
// i.e. 16 independent GEMMs  M_p [T x K] = V_p [T x C] . U_p^T [C x K]  (p = 4 xi + nu, T = N * ceil(H/2) * ceil(W/2) tiles), which run as ONE
// batched launch of the implicit-GEMM kernel (conv_mfma.hip, blockIdx.y = p).  The transforms are streaming kernels (this file):
//
//   wino_filter_k   g [K][3][3][C] (OHWI)      -> U [16][K][C]                      once per weight and step
//   wino_input_k    x [N][H][W][C] (+ fused BatchNorm + ReLU of the producer)  -> V [16][T][C]
//   wino_output_k   M [16][T][K]               -> y [N][H][W][K]  (+ BatchNorm statistics partials | + ReLU gate and its partial sums)
//   wino_dy_k       dy [N][H][W][K]            -> dM [16][T][K] = A dY A^T          weight gradient: dU_p = dM_p^T . V_p  (V kept from the forward)
//   wino_dfilter_k  dU [16][K][C]              -> dg (+)= G^T dU G
//
// All arithmetic is fp32; the transforms use only additions and multiplications by 1/2 (exact), so the result differs from the direct
// convolution by summation order and by the rounding of the transformed operands (measured against fp64 next to the direct kernel:
// tools/probe_winograd.py, profiles/r03_probe_winograd.txt).
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- filter: U[p][k][c] = (G g G^T)[xi][nu],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] -------------------------------------
__global__ void __launch_bounds__(256)
wino_filter_k(int K, int C, const float* __restrict__ g, float* __restrict__ U) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (k, c)
  if (idx >= (int64_t)K * C) return;
  const int k = (int)(idx / C), c = (int)(idx - (int64_t)k * C);
  float w[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) w[r][s] = g[((size_t)k * 9 + r * 3 + s) * C + c];
  float t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = w[0][s];
    t[1][s] = 0.5f * (w[0][s] + w[1][s] + w[2][s]);
    t[2][s] = 0.5f * (w[0][s] - w[1][s] + w[2][s]);
    t[3][s] = w[2][s];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
    const size_t o = (size_t)(a * 4) * K * C + (size_t)k * C + c;
    U[o] = u0; U[o + (size_t)K * C] = u1; U[o + 2 * (size_t)K * C] = u2; U[o + 3 * (size_t)K * C] = u3;
  }
}

// dg[k][r][s][c] (+)= (G^T dU G)[r][s]
__global__ void __launch_bounds__(256)
wino_dfilter_k(int K, int C, const float* __restrict__ dU, float* __restrict__ dg, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)K * C) return;
  const int k = (int)(idx / C), c = (int)(idx - (int64_t)k * C);
  float u[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) u[a][b] = dU[(size_t)(a * 4 + b) * K * C + (size_t)k * C + c];
  float t[3][4];                       // G^T u: rows
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
    t[1][b] = 0.5f * (u[1][b] - u[2][b]);
    t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float d0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]), d1 = 0.5f * (t[r][1] - t[r][2]), d2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    float* o = dg + ((size_t)k * 9 + r * 3) * C + c;
    if (accumulate) { o[0] += d0; o[C] += d1; o[2 * (size_t)C] += d2; }
    else { o[0] = d0; o[C] = d1; o[2 * (size_t)C] = d2; }
  }
}

// ---- input: V[p][t][c] = (B^T d B)[xi][nu] over the 4x4 patch of tile t (rows 2i-1 .. 2i+2, zero outside the image) -----------
// One thread = one tile x 4 channels; the lanes of a wave run along the channels, so every load / store is a whole contiguous row segment.
// XF: x is the producer's raw conv output and the operand is relu(x * scale[c] + shift[c]) - same fmaf / fmaxf as bn_apply_k; padding stays 0.
template <bool XF>
__global__ void __launch_bounds__(256)
wino_input_k(int N, int H, int W, int C, int th, int tw, const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
             float* __restrict__ V, int64_t T) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * C4) return;
  const int64_t t = idx / C4;
  const int c = (int)(idx - t * C4) * 4;
  const int n = (int)(t / (th * tw));
  const int r = (int)(t - (int64_t)n * th * tw);
  const int i = r / tw, j = r - i * tw;
  f32x4 scv = {1.f, 1.f, 1.f, 1.f}, shv = {0.f, 0.f, 0.f, 0.f};
  if constexpr (XF) { scv = ld4(sc + c); shv = ld4(sh + c); }
  f32x4 d[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int hi = 2 * i - 1 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int wi = 2 * j - 1 + b;
      const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v = ld4(x + (((size_t)n * H + hi) * W + wi) * C + c);
        if constexpr (XF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], scv[e], shv[e]), 0.f);
        }
      }
      d[a][b] = v;
    }
  }
  f32x4 m[4][4];                       // B^T d
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    m[0][b] = d[0][b] - d[2][b];
    m[1][b] = d[1][b] + d[2][b];
    m[2][b] = d[2][b] - d[1][b];
    m[3][b] = d[1][b] - d[3][b];
  }
  const size_t ps = (size_t)T * C;
  float* o = V + (size_t)t * C + c;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    st4(o + (size_t)(a * 4 + 0) * ps, m[a][0] - m[a][2]);
    st4(o + (size_t)(a * 4 + 1) * ps, m[a][1] + m[a][2]);
    st4(o + (size_t)(a * 4 + 2) * ps, m[a][2] - m[a][1]);
    st4(o + (size_t)(a * 4 + 3) * ps, m[a][1] - m[a][3]);
  }
}

// ---- weight gradient operand: dM[p][t][k] = (A dY A^T)[xi][nu],  A^T = [[1,1,1,0],[0,1,-1,-1]] ----------------------------------
__global__ void __launch_bounds__(256)
wino_dy_k(int N, int H, int W, int K, int th, int tw, const float* __restrict__ dy, float* __restrict__ dM, int64_t T) {
  const int K4 = K >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * K4) return;
  const int64_t t = idx / K4;
  const int k = (int)(idx - t * K4) * 4;
  const int n = (int)(t / (th * tw));
  const int r = (int)(t - (int64_t)n * th * tw);
  const int i = r / tw, j = r - i * tw;
  f32x4 y[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ho = 2 * i + a, wo = 2 * j + b;
      y[a][b] = (ho < H && wo < W) ? ld4(dy + (((size_t)n * H + ho) * W + wo) * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // A y : rows [y0, y0 + y1, y0 - y1, -y1]
  f32x4 s[4][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) { s[0][b] = y[0][b]; s[1][b] = y[0][b] + y[1][b]; s[2][b] = y[0][b] - y[1][b]; s[3][b] = -y[1][b]; }
  const size_t ps = (size_t)T * K;
  float* o = dM + (size_t)t * K + k;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    st4(o + (size_t)(a * 4 + 0) * ps, s[a][0]);
    st4(o + (size_t)(a * 4 + 1) * ps, s[a][0] + s[a][1]);
    st4(o + (size_t)(a * 4 + 2) * ps, s[a][0] - s[a][1]);
    st4(o + (size_t)(a * 4 + 3) * ps, -s[a][1]);
  }
}

// ---- output: y = A^T M A per tile; one workgroup = one GROUP of 16 consecutive tiles x a run of <= 1024 channels ----------------
// MODE 0 plain | 1 statistics: the group's output rows reduced per channel to (mean, centred sum of squares), one partial per group - the layout
// ssv_bn_stats_finalize takes with rows_per_group = ssv_wino_stats_rows_per_group (every group must hold the same number of rows) | 2 gate: y is the gradient w.r.t. a
// BatchNorm + ReLU output whose input is gx: store g = (gx * gscale + gshift > 0) ? y : 0 and the group's sums of g and g * xhat | 3 the same
// with the ReLU bit taken from the forward's byte mask (one byte per four channels, ssv_bn_apply).
constexpr int WG_TILES = 16;
template <int MODE>
__global__ void __launch_bounds__(256)
wino_output_k(int N, int H, int W, int K, int th, int tw, const float* __restrict__ M, float* __restrict__ y, int64_t T,
              float* __restrict__ p0, float* __restrict__ p1, const float* __restrict__ gx, const float* __restrict__ gscale,
              const float* __restrict__ gshift, const float* __restrict__ gmean, const float* __restrict__ ginvstd, const uint8_t* __restrict__ gmask) {
  __shared__ float red[3][256 * 4];
  const int K4 = K >> 2;
  const int L = K4 < 256 ? K4 : 256;          // lanes along the channels
  const int TPP = 256 / L;                    // tiles per pass
  const int tid = threadIdx.x;
  const int lc = tid % L, ts = tid / L;
  const int k = (blockIdx.y * 256 + lc) * 4;
  const bool kok = k < K && ts < TPP;
  const int64_t g0 = (int64_t)blockIdx.x * WG_TILES;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, piv = {0.f, 0.f, 0.f, 0.f};
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f}, gs = {0.f, 0.f, 0.f, 0.f}, gh = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;
  if constexpr (MODE >= 2) { if (kok) { mu = ld4(gmean + k); is = ld4(ginvstd + k); } }
  if constexpr (MODE == 2) { if (kok) { gs = ld4(gscale + k); gh = ld4(gshift + k); } }
  const size_t ps = (size_t)T * K;
  for (int p = 0; p < WG_TILES; p += TPP) {
    const int64_t t = g0 + p + ts;
    if (!kok || t >= T) continue;
    const int n = (int)(t / (th * tw));
    const int r = (int)(t - (int64_t)n * th * tw);
    const int i = r / tw, j = r - i * tw;
    const float* mp = M + (size_t)t * K + k;
    f32x4 m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[a][b] = ld4(mp + (size_t)(a * 4 + b) * ps);
    f32x4 s[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { s[0][b] = m[0][b] + m[1][b] + m[2][b]; s[1][b] = m[1][b] - m[2][b] - m[3][b]; }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const f32x4 o0 = s[a][0] + s[a][1] + s[a][2], o1 = s[a][1] - s[a][2] - s[a][3];
      const int ho = 2 * i + a;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int wo = 2 * j + b;
        if (ho >= H || wo >= W) continue;
        f32x4 v = b == 0 ? o0 : o1;
        const size_t off = (((size_t)n * H + ho) * W + wo) * K + k;
        if constexpr (MODE >= 2) {
          const f32x4 xv = ld4(gx + off);
          if constexpr (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xv[e], gs[e], gh[e]) > 0.f ? v[e] : 0.f;
          } else {
            const unsigned bits = gmask[off >> 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bits >> e) & 1u ? v[e] : 0.f;
          }
          s1 += v;
          s2 += v * ((xv - mu) * is);
        }
        if constexpr (MODE == 1) {
          if (cnt == 0.f) piv = v;
          const f32x4 dv = v - piv;
          s1 += dv; s2 += dv * dv; cnt += 1.f;
        }
        st4(y + off, v);
      }
    }
  }
  if constexpr (MODE == 0) return;
  // merge the TPP partial results per channel in fixed order (thread ts == 0 of each channel lane)
  if constexpr (MODE == 1) {
    // this thread's (count, mean, M2)
    f32x4 mean = piv, m2 = {0.f, 0.f, 0.f, 0.f};
    if (cnt > 0.f) { mean = piv + s1 / cnt; m2 = s2 - s1 * s1 / cnt; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tid * 4 + e] = mean[e]; red[1][tid * 4 + e] = m2[e]; }
    red[2][tid * 4] = cnt;
    __syncthreads();
    if (ts == 0 && kok) {
      f32x4 am = mean, a2 = m2;
      float an = cnt;
      for (int q = 1; q < TPP; ++q) {
        const int o = (q * L + lc) * 4;
        const float bn = red[2][o];
        if (bn == 0.f) continue;
        const float tot = an + bn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = red[0][o + e] - am[e];
          a2[e] += red[1][o + e] + dlt * dlt * (an * bn / tot);
          am[e] += dlt * (bn / tot);
        }
        an = tot;
      }
      st4(p0 + (size_t)blockIdx.x * K + k, am);
      st4(p1 + (size_t)blockIdx.x * K + k, a2);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tid * 4 + e] = s1[e]; red[1][tid * 4 + e] = s2[e]; }
    __syncthreads();
    if (ts == 0 && kok) {
      f32x4 a1 = s1, a2 = s2;
      for (int q = 1; q < TPP; ++q) {
        const int o = (q * L + lc) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1[e] += red[0][o + e]; a2[e] += red[1][o + e]; }
      }
      st4(p0 + (size_t)blockIdx.x * K + k, a1);
      st4(p1 + (size_t)blockIdx.x * K + k, a2);
    }
  }
}

int check_shape(int N, int H, int W, int C, const char* who) {
  SSV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape (channels %% 4 == 0 required)", who);
  SSV_REQUIRE((int64_t)N * H * W * C < (1ll << 31) * 4, "%s: tensor too large", who);
  return SSV_OK;
}

}  // namespace

extern "C" int64_t ssv_wino_tiles(int32_t N, int32_t H, int32_t W) {
  return (N > 0 && H > 0 && W > 0) ? (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) : 0;
}

extern "C" int64_t ssv_wino_groups(int32_t N, int32_t H, int32_t W) { return cdiv64(ssv_wino_tiles(N, H, W), WG_TILES); }

// Output rows every statistics partial of ssv_wino_output_transform summarises (the rows_per_group to hand ssv_bn_stats_finalize), or 0 when the
// map does not partition evenly: 64 when every tile is whole (H, W even: 16 tiles x 4 rows); H * W when one image is exactly one group of 16
// tiles (7x7 and 8x7 .. maps: ceil(H/2) * ceil(W/2) == 16).
extern "C" int32_t ssv_wino_stats_rows_per_group(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  if (H % 2 == 0 && W % 2 == 0) return 4 * WG_TILES;
  if (((H + 1) / 2) * ((W + 1) / 2) == WG_TILES) return H * W;
  return 0;
}

extern "C" int ssv_wino_filter_transform(int32_t K, int32_t C, const float* w, float* U, void* stream) {
  SSV_REQUIRE(K > 0 && C > 0 && w && U, "ssv_wino_filter_transform: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(wino_filter_k, dim3((unsigned)cdiv64((int64_t)K * C, 256)), dim3(256), 0, s, K, C, w, U);
  SSV_CHECK_LAUNCH("ssv_wino_filter_transform");
  return SSV_OK;
}

extern "C" int ssv_wino_filter_grad(int32_t K, int32_t C, const float* dU, float* dw, int accumulate, void* stream) {
  SSV_REQUIRE(K > 0 && C > 0 && dU && dw, "ssv_wino_filter_grad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  hipLaunchKernelGGL(wino_dfilter_k, dim3((unsigned)cdiv64((int64_t)K * C, 256)), dim3(256), 0, s, K, C, dU, dw, accumulate);
  SSV_CHECK_LAUNCH("ssv_wino_filter_grad");
  return SSV_OK;
}

extern "C" int ssv_wino_input_transform(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, const float* in_scale, const float* in_shift,
                                        float* V, void* stream) {
  if (int rc = check_shape(N, H, W, C, "ssv_wino_input_transform")) return rc;
  SSV_REQUIRE(x && V && (in_scale == nullptr) == (in_shift == nullptr), "ssv_wino_input_transform: bad pointers");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)V | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0, "ssv_wino_input_transform: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const int64_t T = (int64_t)N * th * tw;
  const dim3 grid((unsigned)cdiv64(T * (C / 4), 256));
  if (in_scale) hipLaunchKernelGGL(wino_input_k<true>, grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T);
  else hipLaunchKernelGGL(wino_input_k<false>, grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T);
  SSV_CHECK_LAUNCH("ssv_wino_input_transform");
  return SSV_OK;
}

extern "C" int ssv_wino_dy_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, float* dM, void* stream) {
  if (int rc = check_shape(N, H, W, K, "ssv_wino_dy_transform")) return rc;
  SSV_REQUIRE(dy && dM && (((uintptr_t)dy | (uintptr_t)dM) & 15) == 0, "ssv_wino_dy_transform: null or unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const int64_t T = (int64_t)N * th * tw;
  hipLaunchKernelGGL(wino_dy_k, dim3((unsigned)cdiv64(T * (K / 4), 256)), dim3(256), 0, s, N, H, W, K, th, tw, dy, dM, T);
  SSV_CHECK_LAUNCH("ssv_wino_dy_transform");
  return SSV_OK;
}

// y = A^T M A.  Optional (at most one): statistics partials (pmean, pm2: [ssv_wino_groups][K], 64 rows per group - H and W must be even) or a
// ReLU gate with its partial sums (gate->x / mean / invstd / psum_g / psum_gx: [ssv_wino_groups][K]; byte mask or scale + shift; no second target).
extern "C" int ssv_wino_output_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* M, float* y, float* pmean, float* pm2,
                                         const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_shape(N, H, W, K, "ssv_wino_output_transform")) return rc;
  SSV_REQUIRE(M && y && (((uintptr_t)M | (uintptr_t)y | (uintptr_t)pmean | (uintptr_t)pm2) & 15) == 0, "ssv_wino_output_transform: null or unaligned pointer");
  SSV_REQUIRE((pmean == nullptr) == (pm2 == nullptr), "ssv_wino_output_transform: pmean / pm2 must both be given or both NULL");
  SSV_REQUIRE(!(pmean && gate), "ssv_wino_output_transform: statistics and gate are exclusive");
  SSV_REQUIRE(!pmean || ssv_wino_stats_rows_per_group(N, H, W) > 0, "ssv_wino_output_transform: the statistics epilogue needs whole tiles (H, W even) or one image per group (got %d x %d)", H, W);
  SSV_REQUIRE(K % 4 == 0 && (K / 4 <= 256 ? 256 % (K / 4) == 0 : (K / 4) % 256 == 0), "ssv_wino_output_transform: K / 4 must divide or be a multiple of 256 (got K=%d)", K);
  if (gate) {
    SSV_REQUIRE(gate->x && gate->mean && gate->invstd && gate->psum_g && gate->psum_gx && !gate->x2 &&
                ((gate->mask != nullptr) != (gate->scale != nullptr && gate->shift != nullptr)) && ((gate->scale == nullptr) == (gate->shift == nullptr)),
                "ssv_wino_output_transform: the gate carries x, mean, invstd, psum_g, psum_gx and either the byte mask or scale + shift (no second target)");
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(gate ? SSV_PROF_CONV_DGRAD : SSV_PROF_CONV_FWD, s);
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const int64_t T = (int64_t)N * th * tw;
  const dim3 grid((unsigned)cdiv64(T, WG_TILES), (unsigned)cdiv(K / 4, 256));
  const float* nf = nullptr;
  const uint8_t* nb = nullptr;
  if (gate && gate->mask) hipLaunchKernelGGL(wino_output_k<3>, grid, dim3(256), 0, s, N, H, W, K, th, tw, M, y, T, gate->psum_g, gate->psum_gx, gate->x, nf, nf, gate->mean, gate->invstd, gate->mask);
  else if (gate) hipLaunchKernelGGL(wino_output_k<2>, grid, dim3(256), 0, s, N, H, W, K, th, tw, M, y, T, gate->psum_g, gate->psum_gx, gate->x, gate->scale, gate->shift, gate->mean, gate->invstd, nb);
  else if (pmean) hipLaunchKernelGGL(wino_output_k<1>, grid, dim3(256), 0, s, N, H, W, K, th, tw, M, y, T, pmean, pm2, nf, nf, nf, nf, nf, nb);
  else hipLaunchKernelGGL(wino_output_k<0>, grid, dim3(256), 0, s, N, H, W, K, th, tw, M, y, T, (float*)nullptr, (float*)nullptr, nf, nf, nf, nf, nf, nb);
  SSV_CHECK_LAUNCH("ssv_wino_output_transform");
  return SSV_OK;
}
