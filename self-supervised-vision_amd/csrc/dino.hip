// DINO-specific streaming kernels (HBM-bound; one pass over their operands):
//   * DinoLoss (utils/losses.py:74-89) with its gradient w.r.t. the student logits, and the teacher-centre EMA (models/dino.py:136-141)
//   * weight normalisation of the output layer (nn.utils.weight_norm, models/dino.py:35): w = g * v / ||v||_row
//   * AdamW over a flat arena with the reference's element-wise gradient clamp (models/dino.py:76-79) fused in.
#include "common.h"

namespace {

__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

// tsum[b][k] = sum over the two teacher views g of softmax((teacher[b][g] - center) / temp_t)[k]
__global__ void __launch_bounds__(256) dino_teacher_k(int K, const float* __restrict__ teacher, const float* __restrict__ center,
                                                      float inv_temp_t, float* __restrict__ tsum) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  for (int g = 0; g < 2; ++g) {
    const float* t = teacher + ((int64_t)b * 2 + g) * K;
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, (t[k] - center[k]) * inv_temp_t);
    mx = block_max(mx, sh);
    float sm = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) sm += expf((t[k] - center[k]) * inv_temp_t - mx);
    sm = block_sum(sm, sh);
    const float inv = 1.f / sm;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float p = expf((t[k] - center[k]) * inv_temp_t - mx) * inv;
      float* o = tsum + (int64_t)b * K + k;
      *o = g ? *o + p : p;
    }
  }
}

// one block per student row (b, v): log-softmax, partial loss, gradient
__global__ void __launch_bounds__(256) dino_student_k(int V, int K, const float* __restrict__ student, const float* __restrict__ tsum,
                                                      float inv_temp_s, float gscale, float* __restrict__ dstudent, double* __restrict__ partial) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.x;
  const int b = (int)(row / V);
  const float* srow = student + row * K;
  const float* trow = tsum + (int64_t)b * K;
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, srow[k] * inv_temp_s);
  mx = block_max(mx, sh);
  float sm = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) sm += expf(srow[k] * inv_temp_s - mx);
  sm = block_sum(sm, sh);
  const float lse = mx + logf(sm);
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float logp = srow[k] * inv_temp_s - lse;
    const float tt = trow[k];
    acc -= tt * logp;
    dstudent[row * K + k] = gscale * (2.f * expf(logp) - tt);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[row] = (double)acc;
}

__global__ void dino_loss_reduce_k(int64_t rows, const double* __restrict__ partial, float scale, float* __restrict__ loss, int accumulate) {
  __shared__ double sh[256];
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < rows; i += 256) a += partial[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { const float v = (float)(sh[0] * (double)scale); *loss = accumulate ? *loss + v : v; }
}

__global__ void dino_center_k(int K, int rows1, const float* __restrict__ t1, int rows2, const float* __restrict__ t2, float momentum,
                              float* __restrict__ center) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double acc = 0.0;
  for (int r = 0; r < rows1; ++r) acc += (double)t1[(int64_t)r * K + k];
  for (int r = 0; r < rows2; ++r) acc += (double)t2[(int64_t)r * K + k];
  const float mean = (float)(acc / (double)(rows1 + rows2));
  center[k] = momentum * center[k] + (1.f - momentum) * mean;
}

// ---- weight norm: one wave per output row
__global__ void __launch_bounds__(256) wn_fwd_k(int rows, int cols, const float* __restrict__ g, const float* __restrict__ v,
                                                float* __restrict__ w, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* vr = v + (int64_t)r * cols;
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) s += vr[i] * vr[i];
  const float inv = 1.f / sqrtf(wave_sum(s));
  const float f = g[r] * inv;
  for (int i = lane; i < cols; i += 64) w[(int64_t)r * cols + i] = vr[i] * f;
  if (lane == 0) inv_norm[r] = inv;
}
__global__ void __launch_bounds__(256) wn_bwd_k(int rows, int cols, const float* __restrict__ dw, const float* __restrict__ g,
                                                const float* __restrict__ v, const float* __restrict__ inv_norm,
                                                float* __restrict__ dg, float* __restrict__ dv, int accumulate) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* vr = v + (int64_t)r * cols;
  const float* dr = dw + (int64_t)r * cols;
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) s += dr[i] * vr[i];
  s = wave_sum(s);
  const float inv = inv_norm[r], gg = g[r];
  const float dgr = s * inv;                                  // dL/dg = (dw . v) / ||v||
  const float a = gg * inv, bcoef = gg * s * inv * inv * inv; // dv = g/||v|| dw - g (dw.v)/||v||^3 v
  for (int i = lane; i < cols; i += 64) {
    const float o = a * dr[i] - bcoef * vr[i];
    float* p = dv + (int64_t)r * cols + i;
    *p = accumulate ? *p + o : o;
  }
  if (lane == 0) dg[r] = accumulate ? dg[r] + dgr : dgr;
}

// ---- AdamW (torch.optim.AdamW semantics) with optional second gradient slab and element-wise clamp
__global__ void adamw_k(int64_t n, float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ g2,
                        float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2, float eps, float wd,
                        float bc1, float rsqrt_bc2, const float* __restrict__ bc_dev, float clip) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bc_dev) { bc1 = bc_dev[0]; rsqrt_bc2 = bc_dev[1]; }      // ssv_adamw_counted: the bias corrections of a step count that lives in device memory
  if (lr < 0.f) { lr = bc_dev[2]; wd = bc_dev[3]; }            // ssv_adamw_counted_dev: learning rate and weight decay live there too
  float gi = g[i];
  if (g2) gi += g2[i];
  if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

// one thread: step += 1, then the two bias-correction factors of that step (the arithmetic of ssv_adamw's host side, in double)
__global__ void adamw_tick_k(int64_t* __restrict__ step, float b1, float b2, float* __restrict__ bc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    bc[0] = (float)bc1;
    bc[1] = (float)(1.0 / sqrt(bc2));
  }
}

}  // namespace

extern "C" size_t ssv_dino_loss_workspace_bytes(int32_t bs, int32_t V, int32_t K) {
  if (bs <= 0 || V <= 0 || K <= 0) return 0;
  return (size_t)bs * K * sizeof(float) + (size_t)bs * V * sizeof(double) + 64;
}

extern "C" int ssv_dino_loss(int32_t bs, int32_t V, int32_t K, const float* teacher, const float* student, const float* center,
                             float temp_s, float temp_t, float weight, float* loss, int32_t accumulate_loss, float* dstudent,
                             void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(bs > 0 && V > 0 && K > 0 && teacher && student && center && loss && dstudent && ws, "ssv_dino_loss: bad arguments");
  SSV_REQUIRE(temp_s > 0.f && temp_t > 0.f, "ssv_dino_loss: temperatures must be positive");
  SSV_REQUIRE(ws_bytes >= ssv_dino_loss_workspace_bytes(bs, V, K), "ssv_dino_loss: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  float* tsum = (float*)ws;
  double* partial = (double*)((char*)ws + (((size_t)bs * K * sizeof(float) + 63) & ~(size_t)63));
  const float mean_scale = weight / (float)((int64_t)bs * V);
  hipLaunchKernelGGL(dino_teacher_k, dim3(bs), dim3(256), 0, s, K, teacher, center, 1.f / temp_t, tsum);
  SSV_CHECK_LAUNCH("dino_teacher_k");
  hipLaunchKernelGGL(dino_student_k, dim3(bs * V), dim3(256), 0, s, V, K, student, tsum, 1.f / temp_s, mean_scale / temp_s, dstudent, partial);
  SSV_CHECK_LAUNCH("dino_student_k");
  hipLaunchKernelGGL(dino_loss_reduce_k, dim3(1), dim3(256), 0, s, (int64_t)bs * V, partial, mean_scale, loss, accumulate_loss);
  SSV_CHECK_LAUNCH("dino_loss_reduce_k");
  return SSV_OK;
}

extern "C" int ssv_dino_center_update(int32_t K, int32_t rows1, const float* t1, int32_t rows2, const float* t2, float momentum,
                                      float* center, void* stream) {
  SSV_REQUIRE(K > 0 && rows1 > 0 && rows2 >= 0 && t1 && (rows2 == 0 || t2) && center, "ssv_dino_center_update: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(dino_center_k, dim3(cdiv(K, 256)), dim3(256), 0, s, K, rows1, t1, rows2, t2, momentum, center);
  SSV_CHECK_LAUNCH("dino_center_k");
  return SSV_OK;
}

extern "C" int ssv_weightnorm_fwd(int32_t rows, int32_t cols, const float* g, const float* v, float* w, float* inv_norm, void* stream) {
  SSV_REQUIRE(rows > 0 && cols > 0 && g && v && w && inv_norm, "ssv_weightnorm_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(wn_fwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, cols, g, v, w, inv_norm);
  SSV_CHECK_LAUNCH("wn_fwd_k");
  return SSV_OK;
}

extern "C" int ssv_weightnorm_bwd(int32_t rows, int32_t cols, const float* dw, const float* g, const float* v, const float* inv_norm,
                                  float* dg, float* dv, int32_t accumulate, void* stream) {
  SSV_REQUIRE(rows > 0 && cols > 0 && dw && g && v && inv_norm && dg && dv, "ssv_weightnorm_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(wn_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, cols, dw, g, v, inv_norm, dg, dv, accumulate);
  SSV_CHECK_LAUNCH("wn_bwd_k");
  return SSV_OK;
}

extern "C" int ssv_adamw(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int64_t step, float clip, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && m && v && step >= 1, "ssv_adamw: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, n, p, g, g2, m, v, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)(1.0 / sqrt(bc2)), (const float*)nullptr, clip);
  SSV_CHECK_LAUNCH("adamw_k");
  return SSV_OK;
}

// The same update with the step count in DEVICE memory: *step_dev is incremented and the bias corrections of the new count are formed on the device (bc_dev: two
// floats of scratch the caller owns), so no argument of the launch changes from step to step - what a step replayed as a HIP graph needs (ssv_amd/graph.py).
extern "C" int ssv_adamw_counted(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int64_t* step_dev, float* bc_dev, float clip, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && m && v && step_dev && bc_dev, "ssv_adamw_counted: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  hipLaunchKernelGGL(adamw_tick_k, dim3(1), dim3(64), 0, s, step_dev, beta1, beta2, bc_dev);
  hipLaunchKernelGGL(adamw_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, n, p, g, g2, m, v, lr, beta1, beta2, eps, weight_decay,
                     1.f, 1.f, (const float*)bc_dev, clip);
  SSV_CHECK_LAUNCH("adamw_k(counted)");
  return SSV_OK;
}

// ... and with the learning rate and the weight decay in device memory as well (bc_dev[2], bc_dev[3], written by the host when a schedule moves them): one captured
// graph of the step then serves every epoch.
extern "C" int ssv_adamw_counted_dev(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float beta1, float beta2,
                                     float eps, int64_t* step_dev, float* bc_dev, float clip, void* stream) {
  SSV_REQUIRE(n > 0 && p && g && m && v && step_dev && bc_dev, "ssv_adamw_counted_dev: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_OPTIM, s);
  hipLaunchKernelGGL(adamw_tick_k, dim3(1), dim3(64), 0, s, step_dev, beta1, beta2, bc_dev);
  hipLaunchKernelGGL(adamw_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, n, p, g, g2, m, v, -1.f, beta1, beta2, eps, 0.f,
                     1.f, 1.f, (const float*)bc_dev, clip);
  SSV_CHECK_LAUNCH("adamw_k(counted, device hyper-parameters)");
  return SSV_OK;
}
