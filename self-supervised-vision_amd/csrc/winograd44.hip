// Winograd F(4x4, 3x3) for the forward, the data gradient and (round 5) the weight gradient of the stride-1 3x3 convolutions of the deep stages
// (networks/resnet.py:7-10, 56-58).
//
// F(2x2, 3x3) (winograd.hip) runs these layers with 2.25x fewer multiplies than the direct convolution; F(4x4, 3x3) computes a 4x4 output tile from a 6x6
// input tile with 36 multiplies per channel pair instead of 144 - 4x fewer (3.06x on 14x14 / 7x7 maps, whose tiles of 4 cover 16 / 8) - and its transformed
// input is 2.25x the input instead of 4x:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A     36 GEMMs  M_p [T x K] = V_p [T x C] . U_p^T [C x K],  T = N * ceil(H/4) * ceil(W/4): ONE batched launch (ssv_gemm_batched)
//
// Interpolation points {0, 1, -1, 1/2, -2, inf} - chosen for the smallest fp32 error among the sets tried (tools/probe/wino44_numerics.py): the transforms multiply
// by constants up to 8 and 16/15, so the result is NOT as close to fp64 as F(2x2)'s (whose transforms only add and halve).  Measured against an fp64 convolution
// (tools/probe_winograd44.py, profiles/r04_probe_winograd44.txt): forward / data gradient 2-3x the direct kernel's error, inside the go / no-go bar of 3x.  The
// WEIGHT gradient through F(4x4) is 3.6-4.4e-6 from fp64 with a plain fp32 accumulation over the tiles (round 4 kept it on F(2x2) for that: the input transform can
// still leave that second operand, V2, from the same pass over x) and 1.2e-6 - F(2x2)'s own error - with BLOCKED accumulation (round 5: ssv_gemm_batched_wgrad_blocked,
// profiles/r05_probe_winograd44_wgrad.txt): it now runs on V, the forward's own transformed input, and no second operand is written.
//
//   wino44_filter_k   g [K][3][3][C] (OHWI)   -> U [36][K][C]
//   wino44_input_k    x [N][H][W][C] (+ fused BatchNorm + ReLU)  -> V [36][T][C]  (+ V2 [16][T2][C])
//   wino44_output_k   M [36][T][K]            -> y [N][H][W][K]  (+ statistics partials | + ReLU gate and its partial sums)
//   wino44_dy_k       dy [N][H][W][K]         -> dM [36][T][K] = A dY A^T         weight gradient: dU_p = dM_p^T . V_p, dg (+)= G^T dU G (wino44_dfilter_k)
//   wino44_dy_both_k  dy (or g, x, BatchNorm-backward coefficients) -> dM AND the data gradient's transformed input, one pass
#include "common.h"

namespace {

template <int VW> using vecf = float __attribute__((ext_vector_type(VW)));
template <int VW> __device__ __forceinline__ vecf<VW> ldv(const float* p) { return *reinterpret_cast<const vecf<VW>*>(p); }
template <int VW> __device__ __forceinline__ void stv(float* p, vecf<VW> v) { *reinterpret_cast<vecf<VW>*>(p) = v; }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// B^T (6x6), G (6x3), A^T (4x6) for the points {0, 1, -1, 1/2, -2, inf}:
//   B^T = [ 1 -3/2 -2  3/2  1  0 ]   G = [   1      0      0   ]   A^T = [ 1  1  1   1    1  0 ]
//         [ 0  -1  1/2 5/2  1  0 ]       [  1/3    1/3    1/3  ]         [ 0  1 -1  1/2  -2  0 ]
//         [ 0   1 -5/2 1/2  1  0 ]       [ -1/3    1/3   -1/3  ]         [ 0  1  1  1/4   4  0 ]
//         [ 0  -2  -1   2   1  0 ]       [ -16/15 -8/15  -4/15 ]         [ 0  1 -1  1/8  -8  1 ]
//         [ 0  1/2 -1 -1/2  1  0 ]       [  1/15  -2/15   4/15 ]
//         [ 0   1 -3/2 -2  3/2 1 ]       [   0      0      1   ]
template <class T> __device__ __forceinline__ void bt6(const T (&d)[6], T (&v)[6]) {
  v[0] = d[0] - 1.5f * d[1] - 2.f * d[2] + 1.5f * d[3] + d[4];
  v[1] = -d[1] + 0.5f * d[2] + 2.5f * d[3] + d[4];
  v[2] = d[1] - 2.5f * d[2] + 0.5f * d[3] + d[4];
  v[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  v[4] = 0.5f * d[1] - d[2] - 0.5f * d[3] + d[4];
  v[5] = d[1] - 1.5f * d[2] - 2.f * d[3] + 1.5f * d[4] + d[5];
}
template <class T> __device__ __forceinline__ void at6(const T (&m)[6], T (&y)[4]) {
  y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  y[1] = m[1] - m[2] + 0.5f * m[3] - 2.f * m[4];
  y[2] = m[1] + m[2] + 0.25f * m[3] + 4.f * m[4];
  y[3] = m[1] - m[2] + 0.125f * m[3] - 8.f * m[4] + m[5];
}
__device__ __forceinline__ void g3(const float (&g)[3], float (&u)[6]) {
  u[0] = g[0];
  u[1] = (g[0] + g[1] + g[2]) * (1.f / 3.f);
  u[2] = (-g[0] + g[1] - g[2]) * (1.f / 3.f);
  u[3] = (-16.f * g[0] - 8.f * g[1] - 4.f * g[2]) * (1.f / 15.f);
  u[4] = (g[0] - 2.f * g[1] + 4.f * g[2]) * (1.f / 15.f);
  u[5] = g[2];
}

// A (6x4) = (A^T)^T applied to a 4-vector: the weight gradient's dY transform  dM = A dY A^T
template <class T> __device__ __forceinline__ void a6(const T (&y)[4], T (&m)[6]) {
  m[0] = y[0];
  m[1] = y[0] + y[1] + y[2] + y[3];
  m[2] = y[0] - y[1] + y[2] - y[3];
  m[3] = y[0] + 0.5f * y[1] + 0.25f * y[2] + 0.125f * y[3];
  m[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
  m[5] = y[3];
}
// G^T (3x6) applied to a 6-vector: the filter gradient's back transform  dg = G^T dU G
__device__ __forceinline__ void gt3(const float (&u)[6], float (&r)[3]) {
  r[0] = u[0] + (u[1] - u[2]) * (1.f / 3.f) + (u[4] - 16.f * u[3]) * (1.f / 15.f);
  r[1] = (u[1] + u[2]) * (1.f / 3.f) - (8.f * u[3] + 2.f * u[4]) * (1.f / 15.f);
  r[2] = (u[1] - u[2]) * (1.f / 3.f) + (4.f * u[4] - 4.f * u[3]) * (1.f / 15.f) + u[5];
}

// ---- filter: U[p][k][c] = (G g G^T)[xi][nu] --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
wino44_filter_k(int K, int C, const float* __restrict__ g, float* __restrict__ U) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (k, c)
  if (idx >= (int64_t)K * C) return;
  const int k = (int)(idx / C), c = (int)(idx - (int64_t)k * C);
  float t[6][3];                                   // G g : per filter column s
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float col[3], u[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) col[r] = g[((size_t)k * 9 + r * 3 + s) * C + c];
    g3(col, u);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][s] = u[a];
  }
  const size_t ps = (size_t)K * C;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float u[6];
    g3(t[a], u);
#pragma unroll
    for (int b = 0; b < 6; ++b) U[(size_t)(a * 6 + b) * ps + (size_t)k * C + c] = u[b];
  }
}

// dg[k][r][s][c] (+)= (G^T dU G)[r][s]
__global__ void __launch_bounds__(256)
wino44_dfilter_k(int K, int C, const float* __restrict__ dU, float* __restrict__ dg, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)K * C) return;
  const int k = (int)(idx / C), c = (int)(idx - (int64_t)k * C);
  const size_t ps = (size_t)K * C;
  float t[3][6];                                   // G^T dU : per transformed column b
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    float col[6], r[3];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = dU[(size_t)(a * 6 + b) * ps + (size_t)k * C + c];
    gt3(col, r);
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q][b] = r[q];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float d[3];
    gt3(t[r], d);
    float* o = dg + ((size_t)k * 9 + r * 3) * C + c;
    if (accumulate) { o[0] += d[0]; o[C] += d[1]; o[2 * (size_t)C] += d[2]; }
    else { o[0] = d[0]; o[C] = d[1]; o[2 * (size_t)C] = d[2]; }
  }
}

// ---- weight gradient operand: dM[p][t][k] = (A dY A^T)[xi][nu] over the 4x4 OUTPUT tile t (rows 4i .. 4i+3; zero past the edge of the map) ------------------------
// One thread = one tile x 4 channels, lanes along the channels.  dU_p = dM_p^T . V_p with the V the forward's input transform left (ssv_gemm_batched_wgrad, 36 products).
__global__ void __launch_bounds__(256)
wino44_dy_k(int N, int H, int W, int K, int th, int tw, const float* __restrict__ dy, float* __restrict__ dM, int64_t T) {
  const int K4 = K >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * K4) return;
  const int64_t t = idx / K4;
  const int k = (int)(idx - t * K4) * 4;
  const int n = (int)(t / (th * tw));
  const int r = (int)(t - (int64_t)n * th * tw);
  const int i = r / tw, j = r - i * tw;
  f32x4 y[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ho = 4 * i + a, wo = 4 * j + b;
      y[a][b] = (ho < H && wo < W) ? ld4(dy + (((size_t)n * H + ho) * W + wo) * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  f32x4 s[6][4];                                   // A y, column by column
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x4 col[4], m[6];
#pragma unroll
    for (int a = 0; a < 4; ++a) col[a] = y[a][b];
    a6(col, m);
#pragma unroll
    for (int a = 0; a < 6; ++a) s[a][b] = m[a];
  }
  const size_t ps = (size_t)T * K;
  float* o = dM + (size_t)t * K + k;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x4 m[6];
    a6(s[a], m);
#pragma unroll
    for (int b = 0; b < 6; ++b) st4(o + (size_t)(a * 6 + b) * ps, m[b]);
  }
}

// ---- BOTH operands the backward of a layer takes from its output gradient, from ONE pass over it: Vd = B^T P B of the 6x6 patch (the data gradient's transformed
// input, wino44_input_k<.., false, false>'s arithmetic bit for bit) and dM = A Y A^T of the patch's inner 4x4 tile (wino44_dy_k's, bit for bit).
// DYF: the output gradient is never written - it is formed per element from the BatchNorm backward's operands, dy = A[k] g + B[k] (x - mean[k]) + D[k]
// (coef = [A | mean | B | D], ssv_bn_bwd_coef; the same fmaf form as the implicit-GEMM kernels' formed-on-load operand), zero outside the image.
template <int VW, bool DYF>
__global__ void __launch_bounds__(256)
wino44_dy_both_k(int N, int H, int W, int K, int th, int tw, const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ coef,
                 float* __restrict__ Vd, float* __restrict__ dM, int64_t T) {
  using vt = vecf<VW>;
  const int KV = K / VW;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * KV) return;
  const int64_t t = idx / KV;
  const int k = (int)(idx - t * KV) * VW;
  const int n = (int)(t / (th * tw));
  const int r = (int)(t - (int64_t)n * th * tw);
  const int i = r / tw, j = r - i * tw;
  vt cA = {}, cM = {}, cB = {}, cD = {};
  if constexpr (DYF) { cA = ldv<VW>(coef + k); cM = ldv<VW>(coef + (size_t)K + k); cB = ldv<VW>(coef + 2 * (size_t)K + k); cD = ldv<VW>(coef + 3 * (size_t)K + k); }
  vt d[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int hi = 4 * i - 1 + a;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int wi = 4 * j - 1 + b;
      vt v = {};
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
        const size_t off = (((size_t)n * H + hi) * W + wi) * K + k;
        v = ldv<VW>(dy + off);
        if constexpr (DYF) {
          const vt xv = ldv<VW>(x + off);
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = __builtin_fmaf(v[e], cA[e], __builtin_fmaf(xv[e] - cM[e], cB[e], cD[e]));
        }
      }
      d[a][b] = v;
    }
  }
  const size_t ps = (size_t)T * K;
  {                                                // the weight gradient's operand: A Y A^T, Y = d[1..4][1..4]
    vt s[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      vt col[4], m[6];
#pragma unroll
      for (int a = 0; a < 4; ++a) col[a] = d[1 + a][1 + b];
      a6(col, m);
#pragma unroll
      for (int a = 0; a < 6; ++a) s[a][b] = m[a];
    }
    float* o = dM + (size_t)t * K + k;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      vt m[6];
      a6(s[a], m);
#pragma unroll
      for (int b = 0; b < 6; ++b) stv<VW>(o + (size_t)(a * 6 + b) * ps, m[b]);
    }
  }
  // the data gradient's operand: B^T d B, column by column in place, then row by row
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    vt col[6], v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = d[a][b];
    bt6(col, v);
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a][b] = v[a];
  }
  float* o = Vd + (size_t)t * K + k;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    vt v[6];
    bt6(d[a], v);
#pragma unroll
    for (int b = 0; b < 6; ++b) stv<VW>(o + (size_t)(a * 6 + b) * ps, v[b]);
  }
}

// ---- input: V[p][t][c] = (B^T d B)[xi][nu] over the 6x6 patch of tile t (rows 4i-1 .. 4i+4, zero outside the image) ---------------------------------
// One thread = one tile x VW channels, lanes along the channels.  XF: x is the producer's raw conv output and the operand is relu(x * scale[c] + shift[c])
// (the fmaf / fmaxf of bn_apply_k; padding stays 0).  BOTH: the four F(2x2) input tiles that lie inside this patch - tiles (2i + a, 2j + b) of winograd.hip's
// tiling, patch rows 2a .. 2a + 3 - are transformed too and written to V2 [16][T2][C] with EXACTLY wino_input_k's arithmetic (the weight gradient's operand).
template <int VW, bool XF, bool BOTH>
__global__ void __launch_bounds__(256)
wino44_input_k(int N, int H, int W, int C, int th, int tw, const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
               float* __restrict__ V, int64_t T, float* __restrict__ V2, int th2, int tw2, int64_t T2) {
  using vt = vecf<VW>;
  const int CV = C / VW;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * CV) return;
  const int64_t t = idx / CV;
  const int c = (int)(idx - t * CV) * VW;
  const int n = (int)(t / (th * tw));
  const int r = (int)(t - (int64_t)n * th * tw);
  const int i = r / tw, j = r - i * tw;
  vt scv, shv;
  if constexpr (XF) { scv = ldv<VW>(sc + c); shv = ldv<VW>(sh + c); }
  vt d[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int hi = 4 * i - 1 + a;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int wi = 4 * j - 1 + b;
      vt v = {};
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
        v = ldv<VW>(x + (((size_t)n * H + hi) * W + wi) * C + c);
        if constexpr (XF) {
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], scv[e], shv[e]), 0.f);
        }
      }
      d[a][b] = v;
    }
  }
  if constexpr (BOTH) {
    const size_t ps2 = (size_t)T2 * C;
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int i2 = 2 * i + a2, j2 = 2 * j + b2;
        if (i2 >= th2 || j2 >= tw2) continue;                 // a ragged map: this 2x2 tile does not exist (14 = 3 whole tiles of 4 + 2)
        vt m[4][4];                                           // B2^T d (winograd.hip: rows d0 - d2, d1 + d2, d2 - d1, d1 - d3)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const vt d0 = d[2 * a2][2 * b2 + b], d1 = d[2 * a2 + 1][2 * b2 + b], d2 = d[2 * a2 + 2][2 * b2 + b], d3 = d[2 * a2 + 3][2 * b2 + b];
          m[0][b] = d0 - d2; m[1][b] = d1 + d2; m[2][b] = d2 - d1; m[3][b] = d1 - d3;
        }
        float* o = V2 + ((size_t)((int64_t)n * th2 + i2) * tw2 + j2) * C + c;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          stv<VW>(o + (size_t)(a * 4 + 0) * ps2, m[a][0] - m[a][2]);
          stv<VW>(o + (size_t)(a * 4 + 1) * ps2, m[a][1] + m[a][2]);
          stv<VW>(o + (size_t)(a * 4 + 2) * ps2, m[a][2] - m[a][1]);
          stv<VW>(o + (size_t)(a * 4 + 3) * ps2, m[a][1] - m[a][3]);
        }
      }
    }
  }
  // B^T d, column by column, in place
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    vt col[6], v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = d[a][b];
    bt6(col, v);
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a][b] = v[a];
  }
  const size_t ps = (size_t)T * C;
  float* o = V + (size_t)t * C + c;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    vt v[6];
    bt6(d[a], v);
#pragma unroll
    for (int b = 0; b < 6; ++b) stv<VW>(o + (size_t)(a * 6 + b) * ps, v[b]);
  }
}

// ---- output: y = A^T M A per tile; one workgroup = one GROUP of tiles x a run of <= 1024 channels ----------------------------------------------------------
// A group is one ROW of tiles of one image (ROWG: 4 x W output pixels when H % 4 == 0) or one whole image (H x W output pixels): the statistics partials need
// every group to hold the same number of rows, the gate's partial sums do not.  MODE as wino_output_k: 0 plain | 1 statistics (mean, centred sum of squares per
// group) | 2 gate recomputed from gx * gscale + gshift > 0 | 3 gate from the byte mask; 2 / 3 leave sum g and sum g * xhat per group.
constexpr int OVW = 2;                        // channels per thread of the output transform: 36 loaded values + 24 of the first stage stay in registers at 2+ waves / SIMD
template <int MODE>
__global__ void __launch_bounds__(256)
wino44_output_k(int N, int H, int W, int K, int th, int tw, int rowg, const float* __restrict__ M, float* __restrict__ y, int64_t T,
                float* __restrict__ p0, float* __restrict__ p1, const float* __restrict__ gx, const float* __restrict__ gscale,
                const float* __restrict__ gshift, const float* __restrict__ gmean, const float* __restrict__ ginvstd, const uint8_t* __restrict__ gmask) {
  using vt = vecf<OVW>;
  __shared__ float red[3][256 * OVW];
  const int KV = K / OVW;
  const int L = KV < 256 ? KV : 256;          // lanes along the channels
  const int TPP = 256 / L;                    // tiles per pass
  const int tid = threadIdx.x;
  const int lc = tid % L, ts = tid / L;
  const int k = (blockIdx.y * 256 + lc) * OVW;
  const bool kok = k < K && ts < TPP;
  const int gtiles = rowg ? tw : th * tw;                              // tiles of this group
  const int64_t t0 = (int64_t)blockIdx.x * gtiles;                     // groups are consecutive runs of the tile order (n, i, j)
  vt s1 = {}, s2 = {}, piv = {}, mu = {}, is = {}, gs = {}, gh = {};
  float cnt = 0.f;
  if constexpr (MODE >= 2) { if (kok) { mu = ldv<OVW>(gmean + k); is = ldv<OVW>(ginvstd + k); } }
  if constexpr (MODE == 2) { if (kok) { gs = ldv<OVW>(gscale + k); gh = ldv<OVW>(gshift + k); } }
  const size_t ps = (size_t)T * K;
  for (int p = 0; p < gtiles; p += TPP) {
    const int64_t t = t0 + p + ts;
    if (!kok || p + ts >= gtiles || t >= T) continue;
    const int n = (int)(t / (th * tw));
    const int r = (int)(t - (int64_t)n * th * tw);
    const int i = r / tw, j = r - i * tw;
    const float* mp = M + (size_t)t * K + k;
    // the gate's operands of all 16 output pixels are requested BEFORE the 36 transformed values are consumed: one round trip instead of one per output row
    vt gxv[MODE >= 2 ? 16 : 1];
    unsigned gbits[MODE == 3 ? 16 : 1];
    if constexpr (MODE >= 2) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int ho = min(4 * i + a, H - 1), wo = min(4 * j + b, W - 1);          // clamped: a pixel past the edge is loaded but never used
          const size_t off = (((size_t)n * H + ho) * W + wo) * K + k;
          gxv[a * 4 + b] = ldv<OVW>(gx + off);
          if constexpr (MODE == 3) gbits[a * 4 + b] = gmask[off >> 2] >> (unsigned)(off & 3);          // one byte per four channels
        }
    }
    vt s[4][6];                                // A^T m, column by column
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      vt col[6], o[4];
#pragma unroll
      for (int a = 0; a < 6; ++a) col[a] = ldv<OVW>(mp + (size_t)(a * 6 + b) * ps);
      at6(col, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) s[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ho = 4 * i + a;
      vt o[4];
      at6(s[a], o);
      if (ho >= H) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int wo = 4 * j + b;
        if (wo >= W) continue;
        vt v = o[b];
        const size_t off = (((size_t)n * H + ho) * W + wo) * K + k;
        if constexpr (MODE >= 2) {
          const vt xv = gxv[a * 4 + b];
          if constexpr (MODE == 2) {
#pragma unroll
            for (int e = 0; e < OVW; ++e) v[e] = __builtin_fmaf(xv[e], gs[e], gh[e]) > 0.f ? v[e] : 0.f;
          } else {
            const unsigned bits = gbits[a * 4 + b];
#pragma unroll
            for (int e = 0; e < OVW; ++e) v[e] = (bits >> e) & 1u ? v[e] : 0.f;
          }
          s1 += v;
          s2 += v * ((xv - mu) * is);
        }
        if constexpr (MODE == 1) {
          if (cnt == 0.f) piv = v;
          const vt dv = v - piv;
          s1 += dv; s2 += dv * dv; cnt += 1.f;
        }
        stv<OVW>(y + off, v);
      }
    }
  }
  if constexpr (MODE == 0) return;
  // merge the TPP partial results per channel in fixed order (thread ts == 0 of each channel lane), as wino_output_k
  if constexpr (MODE == 1) {
    vt mean = piv, m2 = {};
    if (cnt > 0.f) { mean = piv + s1 / cnt; m2 = s2 - s1 * s1 / cnt; }
#pragma unroll
    for (int e = 0; e < OVW; ++e) { red[0][tid * OVW + e] = mean[e]; red[1][tid * OVW + e] = m2[e]; }
    red[2][tid * OVW] = cnt;
    __syncthreads();
    if (ts == 0 && kok) {
      vt am = mean, a2 = m2;
      float an = cnt;
      for (int q = 1; q < TPP; ++q) {
        const int o = (q * L + lc) * OVW;
        const float bn = red[2][o];
        if (bn == 0.f) continue;
        const float tot = an + bn;
#pragma unroll
        for (int e = 0; e < OVW; ++e) {
          const float dlt = red[0][o + e] - am[e];
          a2[e] += red[1][o + e] + dlt * dlt * (an * bn / tot);
          am[e] += dlt * (bn / tot);
        }
        an = tot;
      }
      stv<OVW>(p0 + (size_t)blockIdx.x * K + k, am);
      stv<OVW>(p1 + (size_t)blockIdx.x * K + k, a2);
    }
  } else {
#pragma unroll
    for (int e = 0; e < OVW; ++e) { red[0][tid * OVW + e] = s1[e]; red[1][tid * OVW + e] = s2[e]; }
    __syncthreads();
    if (ts == 0 && kok) {
      vt a1 = s1, a2 = s2;
      for (int q = 1; q < TPP; ++q) {
        const int o = (q * L + lc) * OVW;
#pragma unroll
        for (int e = 0; e < OVW; ++e) { a1[e] += red[0][o + e]; a2[e] += red[1][o + e]; }
      }
      stv<OVW>(p0 + (size_t)blockIdx.x * K + k, a1);
      stv<OVW>(p1 + (size_t)blockIdx.x * K + k, a2);
    }
  }
}

int check_shape44(int N, int H, int W, int C, const char* who) {
  SSV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape (channels %% 4 == 0 required)", who);
  SSV_REQUIRE((int64_t)N * H * W * C < (1ll << 31) * 4, "%s: tensor too large", who);
  return SSV_OK;
}

}  // namespace

extern "C" int64_t ssv_wino44_tiles(int32_t N, int32_t H, int32_t W) {
  return (N > 0 && H > 0 && W > 0) ? (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4) : 0;
}

// Groups of the output transform's partials and the output rows each one holds.  Statistics (stats != 0) need equal groups: one ROW of tiles (4 x W pixels) when
// H % 4 == 0, else one image (H x W pixels); the gate's sums take the row groups always.
extern "C" int64_t ssv_wino44_groups(int32_t N, int32_t H, int32_t W, int32_t stats) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return (!stats || H % 4 == 0) ? (int64_t)N * ((H + 3) / 4) : (int64_t)N;
}
extern "C" int64_t ssv_wino44_stats_rows_per_group(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return H % 4 == 0 ? (int64_t)4 * W : (int64_t)H * W;
}

extern "C" int ssv_wino44_filter_transform(int32_t K, int32_t C, const float* w, float* U, void* stream) {
  SSV_REQUIRE(K > 0 && C > 0 && w && U, "ssv_wino44_filter_transform: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(wino44_filter_k, dim3((unsigned)cdiv64((int64_t)K * C, 256)), dim3(256), 0, s, K, C, w, U);
  SSV_CHECK_LAUNCH("ssv_wino44_filter_transform");
  return SSV_OK;
}

// dw (+)= G^T dU G for dU [36][K][C] (the 36 products of ssv_gemm_batched_wgrad on V and the transformed output gradient)
extern "C" int ssv_wino44_filter_grad(int32_t K, int32_t C, const float* dU, float* dw, int accumulate, void* stream) {
  SSV_REQUIRE(K > 0 && C > 0 && dU && dw, "ssv_wino44_filter_grad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  hipLaunchKernelGGL(wino44_dfilter_k, dim3((unsigned)cdiv64((int64_t)K * C, 256)), dim3(256), 0, s, K, C, dU, dw, accumulate);
  SSV_CHECK_LAUNCH("ssv_wino44_filter_grad");
  return SSV_OK;
}

// dM [36][ssv_wino44_tiles][K] = A dY A^T per 4x4 output tile: the weight gradient's second operand
extern "C" int ssv_wino44_dy_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, float* dM, void* stream) {
  if (int rc = check_shape44(N, H, W, K, "ssv_wino44_dy_transform")) return rc;
  SSV_REQUIRE(dy && dM && (((uintptr_t)dy | (uintptr_t)dM) & 15) == 0, "ssv_wino44_dy_transform: null or unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t T = (int64_t)N * th * tw;
  hipLaunchKernelGGL(wino44_dy_k, dim3((unsigned)cdiv64(T * (K / 4), 256)), dim3(256), 0, s, N, H, W, K, th, tw, dy, dM, T);
  SSV_CHECK_LAUNCH("ssv_wino44_dy_transform");
  return SSV_OK;
}

// Vd [36][T][K] = B^T P B (what ssv_wino44_input_transform(dy) writes) and dM [36][T][K] = A dY A^T (what ssv_wino44_dy_transform writes) from ONE pass over the
// output gradient.  dyin != NULL: dy is g, the gradient w.r.t. the BatchNorm output behind this convolution, and the output gradient is formed on load from
// (g, dyin->x, dyin->coef [4][K]) - the BatchNorm backward's element-wise pass never runs for this layer.
extern "C" int ssv_wino44_dy_transform_both(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, const ssv_bn_dyin* dyin, float* Vd, float* dM, void* stream) {
  if (int rc = check_shape44(N, H, W, K, "ssv_wino44_dy_transform_both")) return rc;
  SSV_REQUIRE(dy && Vd && dM && (((uintptr_t)dy | (uintptr_t)Vd | (uintptr_t)dM) & 15) == 0, "ssv_wino44_dy_transform_both: null or unaligned pointer");
  SSV_REQUIRE(!dyin || (dyin->x && dyin->coef && (((uintptr_t)dyin->x | (uintptr_t)dyin->coef) & 15) == 0), "ssv_wino44_dy_transform_both: the formed-on-load operand needs x and coef (16-byte aligned)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t T = (int64_t)N * th * tw;
  const dim3 grid((unsigned)cdiv64(T * (K / 2), 256));
  if (dyin) hipLaunchKernelGGL((wino44_dy_both_k<2, true>), grid, dim3(256), 0, s, N, H, W, K, th, tw, dy, dyin->x, dyin->coef, Vd, dM, T);
  else hipLaunchKernelGGL((wino44_dy_both_k<2, false>), grid, dim3(256), 0, s, N, H, W, K, th, tw, dy, (const float*)nullptr, (const float*)nullptr, Vd, dM, T);
  SSV_CHECK_LAUNCH("ssv_wino44_dy_transform_both");
  return SSV_OK;
}

// V [36][ssv_wino44_tiles][C] (+ V2 [16][ssv_wino_tiles][C] when V2 != NULL: the F(2x2) transformed input of the same activation, for the weight gradient)
extern "C" int ssv_wino44_input_transform(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, const float* in_scale, const float* in_shift,
                                          float* V, float* V2, void* stream) {
  if (int rc = check_shape44(N, H, W, C, "ssv_wino44_input_transform")) return rc;
  SSV_REQUIRE(x && V && (in_scale == nullptr) == (in_shift == nullptr), "ssv_wino44_input_transform: bad pointers");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)V | (uintptr_t)V2 | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0, "ssv_wino44_input_transform: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  const int th = (H + 3) / 4, tw = (W + 3) / 4, th2 = (H + 1) / 2, tw2 = (W + 1) / 2;
  const int64_t T = (int64_t)N * th * tw, T2 = (int64_t)N * th2 * tw2;
  if (V2) {                                  // two transformed inputs from one patch: 2 channels per thread keep the patch + both sets of temporaries in registers
    const dim3 grid((unsigned)cdiv64(T * (C / 2), 256));
    if (in_scale) hipLaunchKernelGGL((wino44_input_k<2, true, true>), grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T, V2, th2, tw2, T2);
    else hipLaunchKernelGGL((wino44_input_k<2, false, true>), grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T, V2, th2, tw2, T2);
  } else {
    const dim3 grid((unsigned)cdiv64(T * (C / 4), 256));
    if (in_scale) hipLaunchKernelGGL((wino44_input_k<4, true, false>), grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T, (float*)nullptr, th2, tw2, T2);
    else hipLaunchKernelGGL((wino44_input_k<4, false, false>), grid, dim3(256), 0, s, N, H, W, C, th, tw, x, in_scale, in_shift, V, T, (float*)nullptr, th2, tw2, T2);
  }
  SSV_CHECK_LAUNCH("ssv_wino44_input_transform");
  return SSV_OK;
}

// y = A^T M A.  Optional (at most one): statistics partials (pmean, pm2: [ssv_wino44_groups(.., 1)][K], ssv_wino44_stats_rows_per_group rows each) or a ReLU gate with
// its partial sums (gate->psum_g / psum_gx: [ssv_wino44_groups(.., 0)][K]; byte mask or scale + shift; no second target).
extern "C" int ssv_wino44_output_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* M, float* y, float* pmean, float* pm2,
                                           const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_shape44(N, H, W, K, "ssv_wino44_output_transform")) return rc;
  SSV_REQUIRE(M && y && (((uintptr_t)M | (uintptr_t)y | (uintptr_t)pmean | (uintptr_t)pm2) & 15) == 0, "ssv_wino44_output_transform: null or unaligned pointer");
  SSV_REQUIRE((pmean == nullptr) == (pm2 == nullptr), "ssv_wino44_output_transform: pmean / pm2 must both be given or both NULL");
  SSV_REQUIRE(!(pmean && gate), "ssv_wino44_output_transform: statistics and gate are exclusive");
  SSV_REQUIRE(K % 4 == 0 && (K / OVW <= 256 ? 256 % (K / OVW) == 0 : (K / OVW) % 256 == 0), "ssv_wino44_output_transform: K / 2 must divide or be a multiple of 256 (got K=%d)", K);
  if (gate) {
    SSV_REQUIRE(gate->x && gate->mean && gate->invstd && gate->psum_g && gate->psum_gx && !gate->x2 &&
                ((gate->mask != nullptr) != (gate->scale != nullptr && gate->shift != nullptr)) && ((gate->scale == nullptr) == (gate->shift == nullptr)),
                "ssv_wino44_output_transform: the gate carries x, mean, invstd, psum_g, psum_gx and either the byte mask or scale + shift (no second target)");
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(gate ? SSV_PROF_CONV_DGRAD : SSV_PROF_CONV_FWD, s);
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t T = (int64_t)N * th * tw;
  const int rowg = (!pmean || H % 4 == 0) ? 1 : 0;
  const dim3 grid((unsigned)ssv_wino44_groups(N, H, W, pmean ? 1 : 0), (unsigned)cdiv(K / OVW, 256));
  const float* nf = nullptr;
  const uint8_t* nb = nullptr;
  if (gate && gate->mask) hipLaunchKernelGGL(wino44_output_k<3>, grid, dim3(256), 0, s, N, H, W, K, th, tw, rowg, M, y, T, gate->psum_g, gate->psum_gx, gate->x, nf, nf, gate->mean, gate->invstd, gate->mask);
  else if (gate) hipLaunchKernelGGL(wino44_output_k<2>, grid, dim3(256), 0, s, N, H, W, K, th, tw, rowg, M, y, T, gate->psum_g, gate->psum_gx, gate->x, gate->scale, gate->shift, gate->mean, gate->invstd, nb);
  else if (pmean) hipLaunchKernelGGL(wino44_output_k<1>, grid, dim3(256), 0, s, N, H, W, K, th, tw, rowg, M, y, T, pmean, pm2, nf, nf, nf, nf, nf, nb);
  else hipLaunchKernelGGL(wino44_output_k<0>, grid, dim3(256), 0, s, N, H, W, K, th, tw, rowg, M, y, T, (float*)nullptr, (float*)nullptr, nf, nf, nf, nf, nf, nb);
  SSV_CHECK_LAUNCH("ssv_wino44_output_transform");
  return SSV_OK;
}
