// Loss-side kernels of the sibling two-view algorithms (they share every encoder / head kernel with SimCLR and BYOL):
//   SimSiamLoss   utils/losses.py:145-152   mean negative dot product of unit vectors, both pairs in one pass
//   RelicLoss     utils/losses.py:195-201   the invariance term on the DIAGONAL logits (the contrastive part is ssv_ntxent_*)
//   MocoLoss      utils/losses.py:49-71     cross-entropy over [positive | queue] logits, given the N x K queue products
//   MemoryBank    models/moco.py:24-41      ring-buffer push of L2-normalised keys
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
negdot_pair_k(int64_t n, const float* __restrict__ o1, const float* __restrict__ o2, const float* __restrict__ t1, const float* __restrict__ t2,
              float scale, float* __restrict__ do1, float* __restrict__ do2, double* __restrict__ part) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    s += (double)(o1[i] * t2[i]) + (double)(o2[i] * t1[i]);
    do1[i] = -scale * t2[i]; do2[i] = -scale * t1[i];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ void finish_sum_k(int nblk, const double* __restrict__ part, float scale, float* __restrict__ out, int accumulate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[i];
    const float v = (float)(s * (double)scale);
    *out = accumulate ? *out + v : v;
  }
}
int nblocks_for(int64_t n) { int64_t b = cdiv64(n, 1024); return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }

// ---- ReLIC invariance term -------------------------------------------------------------------------
// a[n] = zi[n].zo[n] / T, b[n] = zj[n].zo[n] / T   (one wave per row)
__global__ void __launch_bounds__(256) relic_dots_k(int N, int D, const float* __restrict__ zi, const float* __restrict__ zj,
                                                    const float* __restrict__ zo, float inv_temp, float* __restrict__ a, float* __restrict__ b) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  float sa = 0.f, sb = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float o = zo[(int64_t)r * D + d];
    sa += zi[(int64_t)r * D + d] * o; sb += zj[(int64_t)r * D + d] * o;
  }
  sa = wave_sum(sa); sb = wave_sum(sb);
  if (lane == 0) { a[r] = sa * inv_temp; b[r] = sb * inv_temp; }
}
// one block: p = softmax(a), lq = log_softmax(b), q = exp(lq); kl = sum q (lq - p);
//   d kl / d a_m = p_m (g_m - sum_n p_n g_n), g = -q;      d kl / d b_m = h_m - q_m sum_n h_n, h_n = q_n (lq_n - p_n + 1)
__global__ void __launch_bounds__(1024) relic_kl_k(int N, const float* __restrict__ a, const float* __restrict__ b, float alpha,
                                                   float* __restrict__ da, float* __restrict__ db, float* __restrict__ loss, int accumulate) {
  __shared__ double red[1024];
  auto block_reduce = [&](double v, bool is_max) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] = is_max ? fmax(red[threadIdx.x], red[threadIdx.x + o]) : red[threadIdx.x] + red[threadIdx.x + o];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double ma = -1e300, mb = -1e300;
  for (int n = threadIdx.x; n < N; n += 1024) { ma = fmax(ma, (double)a[n]); mb = fmax(mb, (double)b[n]); }
  ma = block_reduce(ma, true); mb = block_reduce(mb, true);
  double sa = 0.0, sb = 0.0;
  for (int n = threadIdx.x; n < N; n += 1024) { sa += exp((double)a[n] - ma); sb += exp((double)b[n] - mb); }
  sa = block_reduce(sa, false); sb = block_reduce(sb, false);
  const double lsb = mb + log(sb);
  double kl = 0.0, pg = 0.0, hs = 0.0;
  for (int n = threadIdx.x; n < N; n += 1024) {
    const double p = exp((double)a[n] - ma) / sa, lq = (double)b[n] - lsb, q = exp(lq);
    kl += q * (lq - p); pg += p * (-q); hs += q * (lq - p + 1.0);
  }
  kl = block_reduce(kl, false); pg = block_reduce(pg, false); hs = block_reduce(hs, false);
  for (int n = threadIdx.x; n < N; n += 1024) {
    const double p = exp((double)a[n] - ma) / sa, lq = (double)b[n] - lsb, q = exp(lq);
    da[n] = (float)((double)alpha * p * (-q - pg));
    db[n] = (float)((double)alpha * (q * (lq - p + 1.0) - q * hs));
  }
  if (threadIdx.x == 0) { const float v = (float)((double)alpha * kl); *loss = accumulate ? *loss + v : v; }
}
__global__ void __launch_bounds__(256) relic_apply_k(int N, int D, const float* __restrict__ zi, const float* __restrict__ zj,
                                                     const float* __restrict__ zo, const float* __restrict__ da, const float* __restrict__ db,
                                                     float inv_temp, float* __restrict__ dzi, float* __restrict__ dzj, float* __restrict__ dzo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * D) return;
  const int r = (int)(i / D);
  const float ga = da[r] * inv_temp, gb = db[r] * inv_temp;
  dzi[i] = ga * zo[i];
  dzj[i] = gb * zo[i];
  dzo[i] = ga * zi[i] + gb * zj[i];
}

// ---- MoCo -----------------------------------------------------------------------------------------
// one block per query row: logits = [q.k / T | neg[0..K) / T]; loss partial; neg row <- d loss / d(raw product) = softmax / (N T)
__global__ void __launch_bounds__(256) moco_rows_k(int N, int D, int K, int ldk, const float* __restrict__ q, const float* __restrict__ k,
                                                   float* __restrict__ neg, float inv_temp, float* __restrict__ dq_init, double* __restrict__ part) {
  __shared__ float sh[4];
  const int r = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto bsum = [&](float v) { v = wave_sum(v); __syncthreads(); if (lane == 0) sh[w] = v; __syncthreads(); return (sh[0] + sh[1]) + (sh[2] + sh[3]); };
  auto bmax = [&](float v) { v = wave_max(v); __syncthreads(); if (lane == 0) sh[w] = v; __syncthreads(); return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])); };
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) dot += q[(int64_t)r * D + d] * k[(int64_t)r * D + d];
  const float pos = bsum(dot) * inv_temp;
  float* row = neg + (int64_t)r * ldk;
  float mx = pos;
  for (int j = threadIdx.x; j < K; j += 256) mx = fmaxf(mx, row[j] * inv_temp);
  mx = bmax(mx);
  float sm = 0.f;
  for (int j = threadIdx.x; j < K; j += 256) sm += expf(row[j] * inv_temp - mx);
  sm = bsum(sm) + expf(pos - mx);
  const float lse = mx + logf(sm), g = inv_temp / (float)N;
  for (int j = threadIdx.x; j < ldk; j += 256) row[j] = j < K ? expf(row[j] * inv_temp - lse) * g : 0.f;
  const float cpos = (expf(pos - lse) - 1.f) * g;
  for (int d = threadIdx.x; d < D; d += 256) dq_init[(int64_t)r * D + d] = cpos * k[(int64_t)r * D + d];      // the positive's share of dq
  if (threadIdx.x == 0) part[r] = (double)(lse - pos);
}
__global__ void moco_loss_sum_k(int N, const double* __restrict__ part, float* __restrict__ loss) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *loss = (float)(sm[0] / (double)N);
}

// bank[(ptr + i) % K] = keys[i] / max(||keys[i]||, eps), i in [first, n): the rows a sequential push of n keys leaves behind
// ptr_dev != NULL (ssv_queue_push_counted): the write position is read from device memory (and advanced by queue_advance_k behind this kernel), so the launch
// carries no argument that changes from step to step - the step can be replayed as a HIP graph.
__global__ void __launch_bounds__(256) queue_push_k(int K, int D, float* __restrict__ bank, int ptr, const int* __restrict__ ptr_dev, int first, int n,
                                                    const float* __restrict__ keys, float eps) {
  const int lane = threadIdx.x & 63, i = first + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  if (ptr_dev) ptr = *ptr_dev;
  const float* src = keys + (int64_t)i * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += src[d] * src[d];
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
  float* dst = bank + (int64_t)((ptr + i) % K) * D;
  for (int d = lane; d < D; d += 64) dst[d] = src[d] * inv;
}

__global__ void queue_advance_k(int K, int n, int* __restrict__ ptr_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ptr_dev = (*ptr_dev + n) % K;
}

}  // namespace

extern "C" int ssv_negdot_pair_fwd_bwd(int64_t n, const float* o1, const float* o2, const float* t1, const float* t2, float scale,
                                       float* loss, float* do1, float* do2, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(n > 0 && o1 && o2 && t1 && t2 && loss && do1 && do2 && ws, "ssv_negdot_pair_fwd_bwd: bad arguments");
  const int nb = nblocks_for(n);
  SSV_REQUIRE(ws_bytes >= (size_t)nb * sizeof(double), "ssv_negdot_pair_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(negdot_pair_k, dim3(nb), dim3(256), 0, s, n, o1, o2, t1, t2, scale, do1, do2, (double*)ws);
  hipLaunchKernelGGL(finish_sum_k, dim3(1), dim3(64), 0, s, nb, (const double*)ws, -scale, loss, 0);
  SSV_CHECK_LAUNCH("ssv_negdot_pair_fwd_bwd");
  return SSV_OK;
}

extern "C" size_t ssv_relic_kl_workspace_bytes(int32_t N) { return N > 0 ? (size_t)N * 4 * sizeof(float) : 0; }

extern "C" int ssv_relic_kl_fwd_bwd(int32_t N, int32_t D, const float* zi, const float* zj, const float* zo, float inv_temp, float alpha,
                                    float* loss, int32_t accumulate_loss, float* dzi, float* dzj, float* dzo,
                                    void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(N > 0 && D > 0 && zi && zj && zo && loss && dzi && dzj && dzo && ws, "ssv_relic_kl_fwd_bwd: bad arguments");
  SSV_REQUIRE(ws_bytes >= ssv_relic_kl_workspace_bytes(N), "ssv_relic_kl_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  float* a = (float*)ws; float* b = a + N; float* da = b + N; float* db = da + N;
  hipLaunchKernelGGL(relic_dots_k, dim3(cdiv(N, 4)), dim3(256), 0, s, N, D, zi, zj, zo, inv_temp, a, b);
  hipLaunchKernelGGL(relic_kl_k, dim3(1), dim3(1024), 0, s, N, (const float*)a, (const float*)b, alpha, da, db, loss, accumulate_loss);
  hipLaunchKernelGGL(relic_apply_k, dim3((unsigned)cdiv64((int64_t)N * D, 256)), dim3(256), 0, s, N, D, zi, zj, zo, (const float*)da, (const float*)db,
                     inv_temp, dzi, dzj, dzo);
  SSV_CHECK_LAUNCH("ssv_relic_kl_fwd_bwd");
  return SSV_OK;
}

extern "C" int ssv_moco_loss_fwd_bwd(int32_t N, int32_t D, int32_t K, int32_t ldk, const float* q, const float* k, float* neg, float inv_temp,
                                     float* loss, float* dq_init, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(N > 0 && D > 0 && K > 0 && ldk >= K && q && k && neg && loss && dq_init && ws, "ssv_moco_loss_fwd_bwd: bad arguments");
  SSV_REQUIRE(ws_bytes >= (size_t)N * sizeof(double), "ssv_moco_loss_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(moco_rows_k, dim3(N), dim3(256), 0, s, N, D, K, ldk, q, k, neg, inv_temp, dq_init, (double*)ws);
  hipLaunchKernelGGL(moco_loss_sum_k, dim3(1), dim3(256), 0, s, N, (const double*)ws, loss);
  SSV_CHECK_LAUNCH("ssv_moco_loss_fwd_bwd");
  return SSV_OK;
}

extern "C" int ssv_queue_push(int32_t K, int32_t D, float* bank, int32_t ptr, int32_t n, const float* keys, float eps, void* stream) {
  SSV_REQUIRE(K > 0 && D > 0 && bank && ptr >= 0 && ptr < K && n > 0 && keys, "ssv_queue_push: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int first = n > K ? n - K : 0;                        // earlier rows would be overwritten by later ones of the same push
  hipLaunchKernelGGL(queue_push_k, dim3(cdiv(n - first, 4)), dim3(256), 0, s, K, D, bank, ptr, (const int*)nullptr, first, n, keys, eps);
  SSV_CHECK_LAUNCH("ssv_queue_push");
  return SSV_OK;
}

// The same push with the queue pointer in DEVICE memory: rows go to (*ptr_dev + i) % K, then *ptr_dev = (*ptr_dev + n) % K (models/moco.py:31-36's pointer walk).
extern "C" int ssv_queue_push_counted(int32_t K, int32_t D, float* bank, int32_t* ptr_dev, int32_t n, const float* keys, float eps, void* stream) {
  SSV_REQUIRE(K > 0 && D > 0 && bank && ptr_dev && n > 0 && keys, "ssv_queue_push_counted: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int first = n > K ? n - K : 0;
  hipLaunchKernelGGL(queue_push_k, dim3(cdiv(n - first, 4)), dim3(256), 0, s, K, D, bank, 0, (const int*)ptr_dev, first, n, keys, eps);
  hipLaunchKernelGGL(queue_advance_k, dim3(1), dim3(64), 0, s, K, n, ptr_dev);
  SSV_CHECK_LAUNCH("ssv_queue_push_counted");
  return SSV_OK;
}
