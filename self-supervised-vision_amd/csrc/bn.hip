// Training-mode BatchNorm over the rows of an [M][C] fp32 matrix (NHWC activations: M = N*H*W),
// fused with ReLU and the residual add, plus the column-sum used for Linear bias gradients.
// All of it is HBM-streaming work: float4 per lane along C (coalesced rows), every thread keeps
// its channel group fixed and walks rows, per-block partials are merged in a FIXED order
// (deterministic) with double precision in the finalize kernels.
//
// Statistics: per block shifted sums  S1 = sum(x-K), S2 = sum((x-K)^2)  with K = first row of the
// block (kills the E[x^2]-E[x]^2 cancellation), turned into (mean_b, M2_b) and merged with Chan's
// formula in double - the same accumulation width ATen's CPU batch_norm uses (acc_type<float> = double).
#include "common.h"

namespace {

struct BnPlan { int C4, CT, RT, GY, rpb, nblk; };

BnPlan bn_plan(int64_t M, int C) {
  BnPlan p;
  p.C4 = C / 4;
  p.CT = p.C4 < 256 ? p.C4 : 256;
  p.RT = 256 / p.CT;
  p.GY = cdiv(p.C4, p.CT);
  int64_t rpb = cdiv64(M, 1024);
  const int64_t min_rows = (int64_t)p.RT * 4;
  if (rpb < min_rows) rpb = min_rows;
  p.rpb = (int)rpb;
  p.nblk = (int)cdiv64(M, rpb);
  return p;
}

// Grid of the pure element-wise passes (bn_apply_k, bn_bwd_apply_k): their result does not depend on the partition, so they get a
// grid of their own - about SSV_APPLY_WGS workgroups in all, each walking a long run of rows.
// SSV_APPLY_* / SSV_EXP_SKIP_APPLY: compile-time switches of DIAGNOSTIC builds only (tools/probe/build_variant.sh; the shipped Makefile defines
// none): grid size, rows in flight and register cap of the element-wise passes, and the timing-only what-if that skips them
// (profiles/r02_experiments_step_time.txt).
// register cap of the element-wise passes (amdgpu_num_vgpr(n) caps the unified file at 2n on gfx950)
#ifdef SSV_APPLY_VGPR
#define SSV_APPLY_ATTR __attribute__((amdgpu_num_vgpr(SSV_APPLY_VGPR)))
#else
#define SSV_APPLY_ATTR
#endif
#ifndef SSV_APPLY_UNROLL
#define SSV_APPLY_UNROLL 4
#endif
#ifndef SSV_APPLY_WGS
#define SSV_APPLY_WGS 1024
#endif
struct ApplyGrid { int rpb, nblk; };
ApplyGrid apply_grid(const BnPlan& p, int64_t M) {
  int64_t blocks = SSV_APPLY_WGS / p.GY;
  if (blocks < 1) blocks = 1;
  int64_t rpb = cdiv64(M, blocks);
  const int64_t min_rows = (int64_t)p.RT * 4;
  if (rpb < min_rows) rpb = min_rows;
  ApplyGrid g;
  g.rpb = (int)rpb;
  g.nblk = (int)cdiv64(M, rpb);
  return g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- forward statistics ----------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bn_stats_k(int64_t M, int C, int CT, int RT, int rpb, const float* __restrict__ x, float* __restrict__ pmean, float* __restrict__ pm2) {
  __shared__ f32x4 sm1[256], sm2[256];
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  const bool active = rt < RT && c4 < C / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, K = {0, 0, 0, 0};
  if (active) {
    K = ld4(x + r0 * C + 4 * c4);
    int64_t r = r0 + rt;
    for (; r + 3 * RT < r1; r += 4 * RT) {          // 4 independent 16-B loads in flight per lane
      const f32x4 v0 = ld4(x + r * C + 4 * c4) - K, v1 = ld4(x + (r + RT) * C + 4 * c4) - K;
      const f32x4 v2 = ld4(x + (r + 2 * RT) * C + 4 * c4) - K, v3 = ld4(x + (r + 3 * RT) * C + 4 * c4) - K;
      s1 += (v0 + v1) + (v2 + v3); s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    for (; r < r1; r += RT) {
      const f32x4 v = ld4(x + r * C + 4 * c4) - K;
      s1 += v; s2 += v * v;
    }
  }
  sm1[tid] = s1; sm2[tid] = s2;
  __syncthreads();
  if (active && rt == 0) {
    for (int j = 1; j < RT; ++j) { s1 += sm1[j * CT + ct]; s2 += sm2[j * CT + ct]; }
    const float n = (float)(r1 - r0);
    const f32x4 mean = K + s1 / n;
    const f32x4 m2 = s2 - s1 * s1 / n;
    st4(pmean + (size_t)blockIdx.x * C + 4 * c4, mean);
    st4(pm2 + (size_t)blockIdx.x * C + 4 * c4, m2);
  }
}

// Merge of the per-block (mean_b, M2_b): 8 channels x 32 groups of partial blocks per workgroup, so that even the
// narrowest layer (C = 64) gets 8 workgroups and a lane walks only nblk/32 partials (4 loads in flight); two
// division-free passes in double, fixed summation order:
//   mean = sum_b n_b*mean_b / M ;  M2 = sum_b (M2_b + n_b*(mean_b-mean)^2)
constexpr int FIN_C = 8, FIN_G = 32;

template <class F>
__device__ __forceinline__ double fin_group_sum(int b0, int b1, F&& term) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = b0;
  for (; b + 3 < b1; b += 4) { a0 += term(b); a1 += term(b + 1); a2 += term(b + 2); a3 += term(b + 3); }
  for (; b < b1; ++b) a0 += term(b);
  return (a0 + a1) + (a2 + a3);
}

__global__ void __launch_bounds__(256)
bn_stats_finalize_k(int64_t M, int C, int rpb, int nblk, const float* __restrict__ pmean, const float* __restrict__ pm2,
                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* nbt,
                    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double sm[FIN_G][FIN_C];
  const int cl = threadIdx.x % FIN_C, grp = threadIdx.x / FIN_C;
  const int c = blockIdx.x * FIN_C + cl;
  const bool ok = c < C;
  const int per = (nblk + FIN_G - 1) / FIN_G;
  const int b0 = min(grp * per, nblk), b1 = min(b0 + per, nblk);
  const double last_n = (double)(M - (int64_t)(nblk - 1) * rpb), full_n = (double)rpb;
  double acc = 0.0;
  if (ok) acc = fin_group_sum(b0, b1, [&](int b) { return (b == nblk - 1 ? last_n : full_n) * (double)pmean[(size_t)b * C + c]; });
  sm[grp][cl] = acc;
  __syncthreads();
  double mean = 0.0;
  for (int g = 0; g < FIN_G; ++g) mean += sm[g][cl];
  mean /= (double)M;
  __syncthreads();
  acc = 0.0;
  if (ok) acc = fin_group_sum(b0, b1, [&](int b) {
    const double d = (double)pmean[(size_t)b * C + c] - mean;
    return (double)pm2[(size_t)b * C + c] + (b == nblk - 1 ? last_n : full_n) * d * d; });
  sm[grp][cl] = acc;
  __syncthreads();
  if (grp == 0 && ok) {
    double m2 = 0.0;
    for (int g = 0; g < FIN_G; ++g) m2 += sm[g][cl];
    const double var = m2 / (double)M;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    save_mean[c] = fmean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - fmean * sc;
    if (running_mean) {
      const float unbiased = (float)(M > 1 ? m2 / (double)(M - 1) : var);
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (nbt && c == 0) *nbt += 1;
  }
}

// With RELU the kernel can also write the ReLU mask, one byte per float4 (bit e = element e was positive): the backward
// then reads 1 byte instead of the 16 bytes of y for each group of four elements.
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
  f32x4 r;
  r[0] = __builtin_fmaf(a[0], b[0], c[0]); r[1] = __builtin_fmaf(a[1], b[1], c[1]);
  r[2] = __builtin_fmaf(a[2], b[2], c[2]); r[3] = __builtin_fmaf(a[3], b[3], c[3]);
  return r;
}

// RES: 0 none, 1 add a materialised residual, 2 add a residual that is itself a BatchNorm of a raw conv output
// (res * rscale + rshift: the projection shortcut's BatchNorm folded into the block's closing kernel)
template <bool RELU, int RES>
__global__ void __launch_bounds__(256) SSV_APPLY_ATTR
bn_apply_k(int64_t M, int C, int CT, int RT, int rpb, const float* __restrict__ x, const float* __restrict__ scale,
           const float* __restrict__ shift, const float* __restrict__ res, const float* __restrict__ rscale, const float* __restrict__ rshift,
           float* __restrict__ y, uint8_t* __restrict__ mask) {
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  if (rt >= RT || c4 >= C / 4) return;
  const f32x4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4);
  f32x4 rsc = {0, 0, 0, 0}, rsh = {0, 0, 0, 0};
  if constexpr (RES == 2) { rsc = ld4(rscale + 4 * c4); rsh = ld4(rshift + 4 * c4); }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  auto body = [&](int64_t r) {
    const size_t o = (size_t)r * C + 4 * c4;
    f32x4 v = fma4(ld4(x + o), sc, sh);
    if constexpr (RES == 1) v += ld4(res + o);
    if constexpr (RES == 2) v += fma4(ld4(res + o), rsc, rsh);
    if constexpr (RELU) {
      if (mask) mask[o >> 2] = (uint8_t)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    st4(y + o, v);
  };
  int64_t r = r0 + rt;
#if SSV_APPLY_UNROLL == 4
  for (; r + 3 * RT < r1; r += 4 * RT) { body(r); body(r + RT); body(r + 2 * RT); body(r + 3 * RT); }
#else
  for (; r + RT < r1; r += 2 * RT) { body(r); body(r + RT); }
#endif
  for (; r < r1; r += RT) body(r);
}

// ---- backward ----------------------------------------------------------------------------------
template <bool RELU>
__device__ __forceinline__ f32x4 masked(f32x4 g, f32x4 yv) {
  if constexpr (RELU) {
    g[0] = yv[0] > 0.f ? g[0] : 0.f; g[1] = yv[1] > 0.f ? g[1] : 0.f;
    g[2] = yv[2] > 0.f ? g[2] : 0.f; g[3] = yv[3] > 0.f ? g[3] : 0.f;
  }
  return g;
}
// g masked by ReLU: from the byte mask when the forward wrote one, else from the sign of y
__device__ __forceinline__ f32x4 relu_grad(f32x4 g, const float* __restrict__ y, const uint8_t* __restrict__ mask, size_t o) {
  if (mask) {
    const unsigned b = mask[o >> 2];
    g[0] = (b & 1u) ? g[0] : 0.f; g[1] = (b & 2u) ? g[1] : 0.f; g[2] = (b & 4u) ? g[2] : 0.f; g[3] = (b & 8u) ? g[3] : 0.f;
    return g;
  }
  return masked<true>(g, ld4(y + o));
}

// RELU: 0 no ReLU behind this BatchNorm, 1 mask from the byte mask / the sign of y, 2 mask recomputed as x * scale + shift > 0
// (the fused path never materialises y nor a mask for conv -> BN -> ReLU -> conv chains; scale / shift are the forward's own
// floats, so the recomputed mask is the forward's ReLU bit for bit)
template <int RELU>
__device__ __forceinline__ f32x4 bwd_gate(f32x4 g, f32x4 xv, f32x4 sc, f32x4 sh, const float* __restrict__ y, const uint8_t* __restrict__ mask, size_t o) {
  if constexpr (RELU == 1) return relu_grad(g, y, mask, o);
  if constexpr (RELU == 2) return masked<true>(g, fma4(xv, sc, sh));
  return g;
}

template <int RELU>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_k(int64_t M, int C, int CT, int RT, int rpb, const float* __restrict__ dy, const float* __restrict__ y,
                const uint8_t* __restrict__ mask, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ psg, float* __restrict__ psgx) {
  __shared__ f32x4 sm1[256], sm2[256];
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  const bool active = rt < RT && c4 < C / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (active) {
    const f32x4 mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
    f32x4 sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
    if constexpr (RELU == 2) { sc = ld4(scale + 4 * c4); sh = ld4(shift + 4 * c4); }
    auto body = [&](int64_t r) {
      const size_t o = (size_t)r * C + 4 * c4;
      const f32x4 xv = ld4(x + o);
      const f32x4 g = bwd_gate<RELU>(ld4(dy + o), xv, sc, sh, y, mask, o);
      const f32x4 xh = (xv - mu) * is;
      s1 += g; s2 += g * xh;
    };
    int64_t r = r0 + rt;
    for (; r + 3 * RT < r1; r += 4 * RT) { body(r); body(r + RT); body(r + 2 * RT); body(r + 3 * RT); }
    for (; r < r1; r += RT) body(r);
  }
  sm1[tid] = s1; sm2[tid] = s2;
  __syncthreads();
  if (active && rt == 0) {
    for (int j = 1; j < RT; ++j) { s1 += sm1[j * CT + ct]; s2 += sm2[j * CT + ct]; }
    st4(psg + (size_t)blockIdx.x * C + 4 * c4, s1);
    st4(psgx + (size_t)blockIdx.x * C + 4 * c4, s2);
  }
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_k(int64_t M, int C, int nblk, const float* __restrict__ psg, const float* __restrict__ psgx,
                  float* dgamma, float* dbeta, int accumulate, float* __restrict__ k1, float* __restrict__ k2,
                  const float* __restrict__ gamma = nullptr, const float* __restrict__ mean = nullptr, const float* __restrict__ invstd = nullptr,
                  float* __restrict__ coef = nullptr) {
  __shared__ double s1[FIN_G][FIN_C], s2[FIN_G][FIN_C];
  const int cl = threadIdx.x % FIN_C, grp = threadIdx.x / FIN_C;
  const int c = blockIdx.x * FIN_C + cl;
  const bool ok = c < C;
  const int per = (nblk + FIN_G - 1) / FIN_G;
  const int b0 = min(grp * per, nblk), b1 = min(b0 + per, nblk);
  double sg = 0.0, sgx = 0.0;
  if (ok) {
    sg = fin_group_sum(b0, b1, [&](int b) { return (double)psg[(size_t)b * C + c]; });
    sgx = fin_group_sum(b0, b1, [&](int b) { return (double)psgx[(size_t)b * C + c]; });
  }
  s1[grp][cl] = sg; s2[grp][cl] = sgx;
  __syncthreads();
  if (grp == 0 && ok) {
    sg = 0.0; sgx = 0.0;
    for (int g = 0; g < FIN_G; ++g) { sg += s1[g][cl]; sgx += s2[g][cl]; }
    if (k1) { k1[c] = (float)(sg / (double)M); k2[c] = (float)(sgx / (double)M); }
    if (coef) {      // dx = A * g + B * (x - mean) + D  (ssv_bn_dyin: the consumer convolutions form dx while they stage it)
      const double gi = (double)gamma[c] * (double)invstd[c];
      coef[c] = (float)gi;
      coef[C + c] = mean[c];
      coef[2 * C + c] = (float)(-gi * (double)invstd[c] * (sgx / (double)M));
      coef[3 * C + c] = (float)(-gi * (sg / (double)M));
    }
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sgx;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sg;
  }
}

template <int RELU, bool DRES>
__global__ void __launch_bounds__(256) SSV_APPLY_ATTR
bn_bwd_apply_k(int64_t M, int C, int CT, int RT, int rpb, const float* __restrict__ dy, const float* __restrict__ y,
               const uint8_t* __restrict__ mask, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
               const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
               const float* __restrict__ k1, const float* __restrict__ k2, float* __restrict__ dx, float* __restrict__ dres) {
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  if (rt >= RT || c4 >= C / 4) return;
  const f32x4 mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
  const f32x4 gi = ld4(gamma + 4 * c4) * is, a1 = ld4(k1 + 4 * c4), a2 = ld4(k2 + 4 * c4);
  f32x4 sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
  if constexpr (RELU == 2) { sc = ld4(scale + 4 * c4); sh = ld4(shift + 4 * c4); }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  auto body = [&](int64_t r) {
    const size_t o = (size_t)r * C + 4 * c4;
    const f32x4 xv = ld4(x + o);
    const f32x4 g = bwd_gate<RELU>(ld4(dy + o), xv, sc, sh, y, mask, o);
    const f32x4 xh = (xv - mu) * is;
    st4(dx + o, gi * (g - a1 - xh * a2));
    if constexpr (DRES) st4(dres + o, g);
  };
  int64_t r = r0 + rt;
#if SSV_APPLY_UNROLL == 4
  for (; r + 3 * RT < r1; r += 4 * RT) { body(r); body(r + RT); body(r + 2 * RT); body(r + 3 * RT); }
#else
  for (; r + RT < r1; r += 2 * RT) { body(r); body(r + RT); }
#endif
  for (; r < r1; r += RT) body(r);
}

// ---- column sum --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
colsum_partial_k(int64_t M, int C, int CT, int RT, int rpb, const float* __restrict__ x, float* __restrict__ ps) {
  __shared__ f32x4 sm1[256];
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  const bool active = rt < RT && c4 < C / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  f32x4 s1 = {0, 0, 0, 0};
  if (active) for (int64_t r = r0 + rt; r < r1; r += RT) s1 += ld4(x + (size_t)r * C + 4 * c4);
  sm1[tid] = s1;
  __syncthreads();
  if (active && rt == 0) {
    for (int j = 1; j < RT; ++j) s1 += sm1[j * CT + ct];
    st4(ps + (size_t)blockIdx.x * C + 4 * c4, s1);
  }
}
__global__ void __launch_bounds__(256)
colsum_finalize_k(int C, int nblk, const float* __restrict__ ps, float* out, int accumulate) {
  __shared__ double s1[FIN_G][FIN_C];
  const int cl = threadIdx.x % FIN_C, grp = threadIdx.x / FIN_C;
  const int c = blockIdx.x * FIN_C + cl;
  const bool ok = c < C;
  const int per = (nblk + FIN_G - 1) / FIN_G;
  const int b0 = min(grp * per, nblk), b1 = min(b0 + per, nblk);
  double s = 0.0;
  if (ok) s = fin_group_sum(b0, b1, [&](int b) { return (double)ps[(size_t)b * C + c]; });
  s1[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && ok) {
    s = 0.0;
    for (int g = 0; g < FIN_G; ++g) s += s1[g][cl];
    out[c] = (accumulate ? out[c] : 0.f) + (float)s;
  }
}

int check_mc(int64_t M, int C, const char* who) {
  SSV_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "%s: need M > 0, C > 0, C %% 4 == 0 (M=%lld C=%d)", who, (long long)M, C);
  SSV_REQUIRE(M < (1ll << 31), "%s: M too large", who);
  return SSV_OK;
}

}  // namespace

extern "C" size_t ssv_bn_workspace_bytes(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0 || C % 4) return 0;
  const BnPlan p = bn_plan(M, C);
  return ((size_t)2 * p.nblk * C + 2 * (size_t)C) * sizeof(float);
}

namespace {
// Merge `factor` consecutive fine partials (rows-per-block rpb, the last one ragged) into one coarse partial per channel, in double
// and in fixed order, so that the finalize kernel walks M / (rpb * factor) partials instead of M / rpb.
__global__ void __launch_bounds__(256)
bn_partials_coarsen_k(int64_t M, int C, int rpb, int nblk, int factor, const float* __restrict__ pmean, const float* __restrict__ pm2,
                      float* __restrict__ cmean, float* __restrict__ cm2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int j = blockIdx.y;
  if (c >= C) return;
  const int b0 = j * factor, b1 = min(b0 + factor, nblk);
  const double last_n = (double)(M - (int64_t)(nblk - 1) * rpb), full_n = (double)rpb;
  double tot = 0.0, sum = 0.0;
  for (int b = b0; b < b1; ++b) { const double n = b == nblk - 1 ? last_n : full_n; tot += n; sum += n * (double)pmean[(size_t)b * C + c]; }
  const double mean = sum / tot;
  double m2 = 0.0;
  for (int b = b0; b < b1; ++b) {
    const double n = b == nblk - 1 ? last_n : full_n, d = (double)pmean[(size_t)b * C + c] - mean;
    m2 += (double)pm2[(size_t)b * C + c] + n * d * d;
  }
  cmean[(size_t)j * C + c] = (float)mean;
  cm2[(size_t)j * C + c] = (float)m2;
}

// partial sums of the backward: plain sums, same two-level scheme
__global__ void __launch_bounds__(256)
bn_sums_coarsen_k(int C, int nblk, int factor, const float* __restrict__ p1, const float* __restrict__ p2, float* __restrict__ c1, float* __restrict__ c2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int j = blockIdx.y;
  if (c >= C) return;
  const int b0 = j * factor, b1 = min(b0 + factor, nblk);
  double a = 0.0, b_ = 0.0;
  for (int b = b0; b < b1; ++b) { a += (double)p1[(size_t)b * C + c]; b_ += (double)p2[(size_t)b * C + c]; }
  c1[(size_t)j * C + c] = (float)a;
  c2[(size_t)j * C + c] = (float)b_;
}

void launch_apply(const BnPlan& p, int64_t M, int C, const float* x, const float* scale, const float* shift, const float* res,
                  const float* rscale, const float* rshift, int relu, float* y, uint8_t* mask, hipStream_t s) {
  const ApplyGrid ag = apply_grid(p, M);
  const dim3 grid(ag.nblk, p.GY);
  const int mode = res ? (rscale ? 2 : 1) : 0;
#ifdef SSV_EXP_SKIP_APPLY
  return;
#endif
#define APPLY(R_, M_) hipLaunchKernelGGL((bn_apply_k<R_, M_>), grid, dim3(256), 0, s, M, C, p.CT, p.RT, ag.rpb, x, scale, shift, res, rscale, rshift, y, mask)
  if (relu) { if (mode == 2) APPLY(true, 2); else if (mode == 1) APPLY(true, 1); else APPLY(true, 0); }
  else      { if (mode == 2) APPLY(false, 2); else if (mode == 1) APPLY(false, 1); else APPLY(false, 0); }
#undef APPLY
}

// statistics partials (rows_per_group rows each, the last ragged) -> mean / invstd / scale / shift (+ running statistics)
int launch_finalize(int64_t M, int C, const float* pmean, const float* pm2, int rpb, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* nbt, float* save_mean, float* save_invstd, float* scale, float* shift,
                    float* coarse, size_t coarse_floats, hipStream_t s) {
  int nblk = (int)cdiv64(M, rpb);
  if (nblk > 2048) {                                          // two-level merge: >= 32 fine partials -> one coarse one, then the usual finalize
    const int cap = (int)(coarse_floats / ((size_t)2 * C));
    const int factor = cap > 0 && cdiv(nblk, 32) > cap ? cdiv(nblk, cap) : 32, ncoarse = cdiv(nblk, factor);
    SSV_REQUIRE((size_t)2 * ncoarse * C <= coarse_floats, "bn finalize: workspace too small for the coarse partials");
    float* cmean = coarse;
    float* cm2 = cmean + (size_t)ncoarse * C;
    hipLaunchKernelGGL(bn_partials_coarsen_k, dim3(cdiv(C, 256), ncoarse), dim3(256), 0, s, M, C, rpb, nblk, factor, pmean, pm2, cmean, cm2);
    pmean = cmean; pm2 = cm2; rpb *= factor; nblk = ncoarse;
  }
  hipLaunchKernelGGL(bn_stats_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, rpb, nblk, pmean, pm2,
                     gamma, beta, eps, momentum, running_mean, running_var, nbt, save_mean, save_invstd, scale, shift);
  return SSV_OK;
}
}  // namespace

extern "C" int ssv_bn_train_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                                const float* residual, int relu, float eps, float momentum,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* y, uint8_t* relu_mask, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_train_fwd")) return rc;
  SSV_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && ws, "ssv_bn_train_fwd: null pointer");
  SSV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "ssv_bn_train_fwd: running_mean/var must both be given or both NULL");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_train_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_FWD, s);
  const BnPlan p = bn_plan(M, C);
  float* pmean = (float*)ws;
  float* pm2 = pmean + (size_t)p.nblk * C;
  float* scale = pm2 + (size_t)p.nblk * C;
  float* shift = scale + C;
  const dim3 grid(p.nblk, p.GY);
  hipLaunchKernelGGL(bn_stats_k, grid, dim3(256), 0, s, M, C, p.CT, p.RT, p.rpb, x, pmean, pm2);
  hipLaunchKernelGGL(bn_stats_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, p.rpb, p.nblk, (const float*)pmean, (const float*)pm2,
                     gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, scale, shift);
  launch_apply(p, M, C, x, scale, shift, residual, nullptr, nullptr, relu, y, relu_mask, s);
  SSV_CHECK_LAUNCH("ssv_bn_train_fwd");
  return SSV_OK;
}

// Same as ssv_bn_train_fwd, but the statistics partials come from the producer (ssv_conv2d_fwd_stats): no pass over x for them.
extern "C" int ssv_bn_train_fwd_partials(int64_t M, int32_t C, const float* x, const float* pmean, const float* pm2, int32_t rows_per_group,
                                         const float* gamma, const float* beta, const float* residual, int relu, float eps, float momentum,
                                         float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                         float* y, uint8_t* relu_mask, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_train_fwd_partials")) return rc;
  SSV_REQUIRE(x && pmean && pm2 && rows_per_group > 0 && gamma && beta && y && save_mean && save_invstd && ws, "ssv_bn_train_fwd_partials: bad arguments");
  SSV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "ssv_bn_train_fwd_partials: running_mean/var must both be given or both NULL");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_train_fwd_partials: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_FWD, s);
  const BnPlan p = bn_plan(M, C);
  float* scale = (float*)ws;
  float* shift = scale + C;
  if (int rc = launch_finalize(M, C, pmean, pm2, rows_per_group, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                               save_mean, save_invstd, scale, shift, shift + C, (size_t)2 * p.nblk * C, s)) return rc;
  launch_apply(p, M, C, x, scale, shift, residual, nullptr, nullptr, relu, y, relu_mask, s);
  SSV_CHECK_LAUNCH("ssv_bn_train_fwd_partials");
  return SSV_OK;
}

// ---- the fused path's pieces: statistics -> affine, and the apply on its own ---------------------------------------------------
extern "C" int ssv_bn_stats_finalize(int64_t M, int32_t C, const float* pmean, const float* pm2, int32_t rows_per_group,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     float* save_mean, float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_stats_finalize")) return rc;
  SSV_REQUIRE(pmean && pm2 && rows_per_group > 0 && gamma && beta && save_mean && save_invstd && scale && shift && ws, "ssv_bn_stats_finalize: bad arguments");
  SSV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "ssv_bn_stats_finalize: running_mean/var must both be given or both NULL");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_stats_finalize: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_FWD, s);
  const BnPlan p = bn_plan(M, C);
  if (int rc = launch_finalize(M, C, pmean, pm2, rows_per_group, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                               save_mean, save_invstd, scale, shift, (float*)ws, (size_t)2 * p.nblk * C, s)) return rc;
  SSV_CHECK_LAUNCH("ssv_bn_stats_finalize");
  return SSV_OK;
}

extern "C" int ssv_bn_apply(int64_t M, int32_t C, const float* x, const float* scale, const float* shift,
                            const float* residual, const float* res_scale, const float* res_shift, int relu,
                            float* y, uint8_t* relu_mask, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_apply")) return rc;
  SSV_REQUIRE(x && scale && shift && y, "ssv_bn_apply: null pointer");
  SSV_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || residual), "ssv_bn_apply: res_scale / res_shift go together and need a residual");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_FWD, s);
  launch_apply(bn_plan(M, C), M, C, x, scale, shift, residual, res_scale, res_shift, relu, y, relu_mask, s);
  SSV_CHECK_LAUNCH("ssv_bn_apply");
  return SSV_OK;
}

namespace {
// relu_mode: 0 none, 1 byte mask / sign of y, 2 recomputed from x * scale + shift
int launch_bwd(int64_t M, int C, const float* dy, const float* y, const uint8_t* relu_mask, const float* x, const float* gamma,
               const float* save_mean, const float* save_invstd, const float* scale, const float* shift, int relu_mode,
               float* dx, float* dresidual, float* dgamma, float* dbeta, int accumulate, void* ws, hipStream_t s) {
  const BnPlan p = bn_plan(M, C);
  float* psg = (float*)ws;
  float* psgx = psg + (size_t)p.nblk * C;
  float* k1 = psgx + (size_t)p.nblk * C;
  float* k2 = k1 + C;
  const dim3 grid(p.nblk, p.GY);
#define REDUCE(R_) hipLaunchKernelGGL((bn_bwd_reduce_k<R_>), grid, dim3(256), 0, s, M, C, p.CT, p.RT, p.rpb, dy, y, relu_mask, x, save_mean, save_invstd, scale, shift, psg, psgx)
  if (relu_mode == 2) REDUCE(2); else if (relu_mode == 1) REDUCE(1); else REDUCE(0);
#undef REDUCE
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, p.nblk, (const float*)psg, (const float*)psgx,
                     dgamma, dbeta, accumulate, k1, k2);
  const float* ck1 = k1; const float* ck2 = k2;
  const ApplyGrid ag = apply_grid(p, M);
  const dim3 agrid(ag.nblk, p.GY);
#ifdef SSV_EXP_SKIP_APPLY
  return SSV_OK;
#endif
#define BAPPLY(R_, D_) hipLaunchKernelGGL((bn_bwd_apply_k<R_, D_>), agrid, dim3(256), 0, s, M, C, p.CT, p.RT, ag.rpb, dy, y, relu_mask, x, gamma, save_mean, save_invstd, scale, shift, ck1, ck2, dx, dresidual)
  if (dresidual) { if (relu_mode == 2) BAPPLY(2, true); else if (relu_mode == 1) BAPPLY(1, true); else BAPPLY(0, true); }
  else           { if (relu_mode == 2) BAPPLY(2, false); else if (relu_mode == 1) BAPPLY(1, false); else BAPPLY(0, false); }
#undef BAPPLY
  return SSV_OK;
}
}  // namespace

extern "C" int ssv_bn_train_bwd(int64_t M, int32_t C, const float* dy, const float* y, const uint8_t* relu_mask, const float* x,
                                const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                                float* dx, float* dresidual, float* dgamma, float* dbeta, int accumulate,
                                void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_train_bwd")) return rc;
  SSV_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && ws, "ssv_bn_train_bwd: null pointer");
  SSV_REQUIRE(!relu || y || relu_mask, "ssv_bn_train_bwd: the ReLU mask needs y or relu_mask");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_train_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_BWD, s);
  launch_bwd(M, C, dy, y, relu_mask, x, gamma, save_mean, save_invstd, nullptr, nullptr, relu ? 1 : 0, dx, dresidual, dgamma, dbeta, accumulate, ws, s);
  SSV_CHECK_LAUNCH("ssv_bn_train_bwd");
  return SSV_OK;
}

// BatchNorm + ReLU backward of the fused path: neither y nor a mask exists; the ReLU gate is x * scale + shift > 0
extern "C" int ssv_bn_relu_bwd_affine(int64_t M, int32_t C, const float* dy, const float* x, const float* gamma,
                                      const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                                      float* dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_relu_bwd_affine")) return rc;
  SSV_REQUIRE(dy && x && gamma && save_mean && save_invstd && scale && shift && dx && ws, "ssv_bn_relu_bwd_affine: null pointer");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_relu_bwd_affine: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_BWD, s);
  launch_bwd(M, C, dy, nullptr, nullptr, x, gamma, save_mean, save_invstd, scale, shift, 2, dx, nullptr, dgamma, dbeta, accumulate, ws, s);
  SSV_CHECK_LAUNCH("ssv_bn_relu_bwd_affine");
  return SSV_OK;
}

extern "C" int ssv_bn_bwd_from_partials(int64_t M, int32_t C, const float* g, const float* x, const float* gamma,
                                        const float* save_mean, const float* save_invstd, const float* psum_g, const float* psum_gx, int64_t groups,
                                        float* dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_bwd_from_partials")) return rc;
  SSV_REQUIRE(g && x && gamma && save_mean && save_invstd && psum_g && psum_gx && groups > 0 && groups < (1ll << 31) && dx && ws, "ssv_bn_bwd_from_partials: bad arguments");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_bwd_from_partials: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_BWD, s);
  const BnPlan p = bn_plan(M, C);
  float* k1 = (float*)ws;
  float* k2 = k1 + C;
  int nblk = (int)groups;
  if (nblk > 2048) {                                          // two-level merge, as in the forward
    const int cap = p.nblk;                                   // the workspace holds 2 * p.nblk * C floats behind k1 / k2
    const int factor = cdiv(nblk, 32) > cap ? cdiv(nblk, cap) : 32, ncoarse = cdiv(nblk, factor);
    float* c1 = k2 + C;
    float* c2 = c1 + (size_t)ncoarse * C;
    hipLaunchKernelGGL(bn_sums_coarsen_k, dim3(cdiv(C, 256), ncoarse), dim3(256), 0, s, C, nblk, factor, psum_g, psum_gx, c1, c2);
    psum_g = c1; psum_gx = c2; nblk = ncoarse;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, nblk, psum_g, psum_gx, dgamma, dbeta, accumulate, k1, k2);
  const ApplyGrid ag = apply_grid(p, M);
  const dim3 grid(ag.nblk, p.GY);
#ifdef SSV_EXP_SKIP_APPLY
  return SSV_OK;
#endif
  hipLaunchKernelGGL((bn_bwd_apply_k<0, false>), grid, dim3(256), 0, s, M, C, p.CT, p.RT, ag.rpb, g, (const float*)nullptr, (const uint8_t*)nullptr, x, gamma,
                     save_mean, save_invstd, (const float*)nullptr, (const float*)nullptr, (const float*)k1, (const float*)k2, dx, (float*)nullptr);
  SSV_CHECK_LAUNCH("ssv_bn_bwd_from_partials");
  return SSV_OK;
}

// The same merge without the apply pass: the coefficients of dx = A * g + B * (x - mean) + D per channel, for consumers that form dx while
// they stage it (ssv_conv2d_fwd_dyin, ssv_conv2d_wgrad_dyin).  coef: [4][C] floats = A | mean | B | D.
extern "C" int ssv_bn_bwd_coef(int64_t M, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd,
                               const float* psum_g, const float* psum_gx, int64_t groups, float* coef, float* dgamma, float* dbeta, int accumulate,
                               void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_bn_bwd_coef")) return rc;
  SSV_REQUIRE(gamma && save_mean && save_invstd && psum_g && psum_gx && groups > 0 && groups < (1ll << 31) && coef && ws, "ssv_bn_bwd_coef: bad arguments");
  SSV_REQUIRE((((uintptr_t)coef | (uintptr_t)psum_g | (uintptr_t)psum_gx) & 15) == 0, "ssv_bn_bwd_coef: pointers must be 16-byte aligned");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_bwd_coef: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_BWD, s);
  const BnPlan p = bn_plan(M, C);
  float* k2 = (float*)ws + C;
  int nblk = (int)groups;
  if (nblk > 2048) {                                          // two-level merge, as in ssv_bn_bwd_from_partials
    const int cap = p.nblk;
    const int factor = cdiv(nblk, 32) > cap ? cdiv(nblk, cap) : 32, ncoarse = cdiv(nblk, factor);
    float* c1 = k2 + C;
    float* c2 = c1 + (size_t)ncoarse * C;
    hipLaunchKernelGGL(bn_sums_coarsen_k, dim3(cdiv(C, 256), ncoarse), dim3(256), 0, s, C, nblk, factor, psum_g, psum_gx, c1, c2);
    psum_g = c1; psum_gx = c2; nblk = ncoarse;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, nblk, psum_g, psum_gx, dgamma, dbeta, accumulate,
                     (float*)nullptr, (float*)nullptr, gamma, save_mean, save_invstd, coef);
  SSV_CHECK_LAUNCH("ssv_bn_bwd_coef");
  return SSV_OK;
}

// ---- the image stem: BatchNorm + ReLU + MaxPool2d(3, 2, 1) as one pass each way (networks/resnet.py:147-148) ----------------
// forward : pooled = max over the window of relu(y * scale + shift); y is read once, the 112^2 activation is never written
// backward: g[pixel] = relu'(a) * sum of dpool over the (at most 2 x 2) windows whose argmax is this pixel - formed on the fly
//           in the BatchNorm backward's reduction AND apply passes (same partition, same order as bn_bwd_reduce_k / _apply_k, so the
//           result is bit-identical to maxpool_bwd -> bn_train_bwd), instead of materialising dy of the 112^2 map twice.
namespace {
__global__ void __launch_bounds__(256)
bn_relu_maxpool_fwd_k(int N, int H, int W, int C, int Ho, int Wo, const float* __restrict__ y, const float* __restrict__ scale,
                      const float* __restrict__ shift, float* __restrict__ out, uint8_t* __restrict__ am, float* __restrict__ xmax) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const f32x4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 braw = {0.f, 0.f, 0.f, 0.f};                       // the raw conv output at the arg-max pixel (xmax: the backward's reduction reads it)
    int bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int wi = wo * 2 - 1 + s_;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 raw = ld4(y + (((size_t)n * H + hi) * W + wi) * C + 4 * c4);
        f32x4 v = fma4(raw, sc, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fmaxf(v[e], 0.f);                                                     // the ReLU of bn_apply_k
          if (first[e] || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; braw[e] = raw[e]; bi[e] = r * 3 + s_; first[e] = false; }   // ties: first, like ATen
        }
      }
    }
    st4(out + (size_t)i * 4, best);
    uchar4 a; a.x = (uint8_t)bi[0]; a.y = (uint8_t)bi[1]; a.z = (uint8_t)bi[2]; a.w = (uint8_t)bi[3];
    reinterpret_cast<uchar4*>(am)[i] = a;
    if (xmax) st4(xmax + (size_t)i * 4, braw);
  }
}

// gradient w.r.t. the BatchNorm output of pixel (n, hi, wi), ReLU included.  A pixel sits in at most 2 x 2 windows: along each axis the
// window (c + 1) >> 1 with slot (c + 1) & 1, and for odd c also the window before it with slot 2.  The four candidates are visited in
// the order of maxpool_bwd_k (slot-row major), so the sum is bit-identical.
__device__ __forceinline__ f32x4 pool_gate_grad(int n, int hi, int wi, int Ho, int Wo, int C4, int c4, const float* __restrict__ dpool,
                                                const uint8_t* __restrict__ am, f32x4 a) {
  const int hA = (hi + 1) >> 1, rA = (hi + 1) & 1, wA = (wi + 1) >> 1, sA = (wi + 1) & 1;
  // (window, slot) per axis in increasing slot order: [A (slot 0/1)], then [A - 1 (slot 2)] when the coordinate is odd
  const int hw[2] = {hA, hA - 1}, hs[2] = {rA, 2}, ww[2] = {wA, wA - 1}, wsl[2] = {sA, 2};
  const bool hv[2] = {hA < Ho, (hi & 1) != 0}, wv[2] = {wA < Wo, (wi & 1) != 0};
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (hv[i] && wv[j]) {
        const size_t o = (((size_t)n * Ho + hw[i]) * Wo + ww[j]) * C4 + c4;
        const uchar4 m = reinterpret_cast<const uchar4*>(am)[o];
        const f32x4 d = ld4(dpool + o * 4);
        const int slot = hs[i] * 3 + wsl[j];
        g[0] += m.x == slot ? d[0] : 0.f; g[1] += m.y == slot ? d[1] : 0.f;
        g[2] += m.z == slot ? d[2] : 0.f; g[3] += m.w == slot ? d[3] : 0.f;
      }
    }
  return masked<true>(g, a);
}

template <bool APPLY>
__global__ void __launch_bounds__(256)
bn_pool_bwd_k(int64_t M, int C, int CT, int RT, int rpb, int H, int W, int Ho, int Wo, const float* __restrict__ dpool, const uint8_t* __restrict__ am,
              const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
              const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ k1, const float* __restrict__ k2,
              float* __restrict__ psg, float* __restrict__ psgx, float* __restrict__ dx) {
  __shared__ f32x4 sm1[APPLY ? 1 : 256], sm2[APPLY ? 1 : 256];
  const int tid = threadIdx.x;
  const int ct = tid % CT, rt = tid / CT;
  const int c4 = blockIdx.y * CT + ct;
  const bool active = rt < RT && c4 < C / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < M ? r0 + rpb : M;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (active) {
    const f32x4 mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4), sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4);
    f32x4 gi = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if constexpr (APPLY) { gi = ld4(gamma + 4 * c4) * is; a1 = ld4(k1 + 4 * c4); a2 = ld4(k2 + 4 * c4); }
    // (n, hi, wi) of this thread's current row, advanced by RT pixels per visit without divisions (RT <= W is checked by the host)
    int64_t r = r0 + rt;
    int wi = (int)(r % W);
    int hi = (int)((r / W) % H), n = (int)(r / ((int64_t)W * H));
    auto body = [&](int64_t rr) {
      const size_t o = (size_t)rr * C + 4 * c4;
      const f32x4 xv = ld4(x + o);
      const f32x4 g = pool_gate_grad(n, hi, wi, Ho, Wo, C / 4, c4, dpool, am, fma4(xv, sc, sh));
      const f32x4 xh = (xv - mu) * is;
      if constexpr (APPLY) st4(dx + o, gi * (g - a1 - xh * a2));
      else { s1 += g; s2 += g * xh; }
      wi += RT;
      if (wi >= W) { wi -= W; if (++hi == H) { hi = 0; ++n; } }
    };
    for (; r + 3 * RT < r1; r += 4 * RT) { body(r); body(r + RT); body(r + 2 * RT); body(r + 3 * RT); }
    for (; r < r1; r += RT) body(r);
  }
  if constexpr (!APPLY) {
    sm1[tid] = s1; sm2[tid] = s2;
    __syncthreads();
    if (active && rt == 0) {
      for (int j = 1; j < RT; ++j) { s1 += sm1[j * CT + ct]; s2 += sm2[j * CT + ct]; }
      st4(psg + (size_t)blockIdx.x * C + 4 * c4, s1);
      st4(psgx + (size_t)blockIdx.x * C + 4 * c4, s2);
    }
  }
}
}  // namespace

extern "C" int ssv_bn_relu_maxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* y, const float* scale, const float* shift,
                                       float* out, uint8_t* argmax, float* xmax, void* stream) {
  SSV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ssv_bn_relu_maxpool_fwd: bad shape (C %% 4 == 0 required)");
  SSV_REQUIRE(y && scale && shift && out && argmax, "ssv_bn_relu_maxpool_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_FWD, s);
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_k, dim3((unsigned)blocks), dim3(256), 0, s, N, H, W, C, Ho, Wo, y, scale, shift, out, argmax, xmax);
  SSV_CHECK_LAUNCH("ssv_bn_relu_maxpool_fwd");
  return SSV_OK;
}

extern "C" int ssv_bn_relu_maxpool_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dpool, const uint8_t* argmax, const float* y, const float* xmax,
                                       const float* gamma, const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                                       float* dy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  const int64_t M = (int64_t)N * H * W;
  if (int rc = check_mc(M, C, "ssv_bn_relu_maxpool_bwd")) return rc;
  SSV_REQUIRE(N > 0 && H > 0 && W > 0, "ssv_bn_relu_maxpool_bwd: bad shape");
  SSV_REQUIRE(bn_plan((int64_t)N * H * W, C).RT <= W, "ssv_bn_relu_maxpool_bwd: the feature map is narrower than the row stride of the kernel (W=%d)", W);
  SSV_REQUIRE(dpool && argmax && y && gamma && save_mean && save_invstd && scale && shift && dy && ws, "ssv_bn_relu_maxpool_bwd: null pointer");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_bn_relu_maxpool_bwd: workspace too small");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_BN_BWD, s);
  const BnPlan p = bn_plan(M, C);
  float* psg = (float*)ws;
  float* psgx = psg + (size_t)p.nblk * C;
  float* k1 = psgx + (size_t)p.nblk * C;
  float* k2 = k1 + C;
  const dim3 grid(p.nblk, p.GY);
  int nred = p.nblk;
  if (xmax && bn_plan((int64_t)N * Ho * Wo, C).nblk > p.nblk) xmax = nullptr;      // the pooled plan's partials must fit the workspace laid out above (they do but for tiny maps)
  if (xmax) {
    // Reduction at the POOLED resolution (round 4): sum g and sum g * xhat over the 112^2 map are sums over the pooled positions - a pixel's g is the sum of
    // the dpool of the windows whose arg-max it is, gated by the ReLU bit of that pixel, and the forward kept the raw conv output of every window's arg-max
    // pixel (xmax) - so the pass reads dpool and xmax (2 x 1/4 of the map) instead of walking the full-resolution conv output: the generic reduction
    // kernel with the recomputed gate, on [N*Ho*Wo][C].  Same terms as the full-resolution walk, grouped per window instead of per pixel (rounding-level).
    const int64_t Mp = (int64_t)N * Ho * Wo;
    const BnPlan pp = bn_plan(Mp, C);
    nred = pp.nblk;
    hipLaunchKernelGGL((bn_bwd_reduce_k<2>), dim3(pp.nblk, pp.GY), dim3(256), 0, s, Mp, C, pp.CT, pp.RT, pp.rpb, dpool, (const float*)nullptr,
                       (const uint8_t*)nullptr, xmax, save_mean, save_invstd, scale, shift, psg, psgx);
  } else {
    hipLaunchKernelGGL((bn_pool_bwd_k<false>), grid, dim3(256), 0, s, M, C, p.CT, p.RT, p.rpb, H, W, Ho, Wo, dpool, argmax, y, gamma, save_mean, save_invstd,
                       scale, shift, (const float*)nullptr, (const float*)nullptr, psg, psgx, (float*)nullptr);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, M, C, nred, (const float*)psg, (const float*)psgx,
                     dgamma, dbeta, accumulate, k1, k2);
  hipLaunchKernelGGL((bn_pool_bwd_k<true>), grid, dim3(256), 0, s, M, C, p.CT, p.RT, p.rpb, H, W, Ho, Wo, dpool, argmax, y, gamma, save_mean, save_invstd,
                     scale, shift, (const float*)k1, (const float*)k2, (float*)nullptr, (float*)nullptr, dy);
  SSV_CHECK_LAUNCH("ssv_bn_relu_maxpool_bwd");
  return SSV_OK;
}

extern "C" int ssv_colsum(int64_t M, int32_t C, const float* x, float* out, int accumulate,
                          void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_mc(M, C, "ssv_colsum")) return rc;
  SSV_REQUIRE(x && out && ws, "ssv_colsum: null pointer");
  if (ws_bytes < ssv_bn_workspace_bytes(M, C)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_colsum: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const BnPlan p = bn_plan(M, C);
  float* part = (float*)ws;
  hipLaunchKernelGGL(colsum_partial_k, dim3(p.nblk, p.GY), dim3(256), 0, s, M, C, p.CT, p.RT, p.rpb, x, part);
  hipLaunchKernelGGL(colsum_finalize_k, dim3(cdiv(C, FIN_C)), dim3(256), 0, s, C, p.nblk, (const float*)part, out, accumulate);
  SSV_CHECK_LAUNCH("ssv_colsum");
  return SSV_OK;
}
