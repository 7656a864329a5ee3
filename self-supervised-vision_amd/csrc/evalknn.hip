// k-nearest-neighbour label agreement on the GPU (reference utils/eval_utils.py:13-21, which goes through
// faiss.IndexFlatIP = exact inner-product search):  S = Z Z^T by row chunks on the fp32-MFMA implicit-GEMM kernel
// (a 1x1 "convolution" of the chunk with the whole feature matrix as the filter bank), then one wavefront per query row
// streams its row of S keeping a sorted top-(k+1) per lane in registers; the 64 lists are merged with k+1 rounds of a
// wave arg-max.  The best hit is dropped (the reference drops column 0 blindly - normally the query itself) and the labels
// of the next k are compared with the query's.  Ordering: similarity descending, ties by ascending index.
#include "common.h"
#include <limits.h>

namespace {

template <int CAP>
__global__ void __launch_bounds__(256) knn_agree_k(const float* __restrict__ S, int64_t lds, int n, int rows, int row0,
                                                   const int32_t* __restrict__ labels, int k,
                                                   unsigned long long* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;                                   // whole wave leaves together
  const float* s = S + (int64_t)r * lds;
  float val[CAP];
  int idx[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i) { val[i] = -INFINITY; idx[i] = INT_MAX; }
  auto offer = [&](float v, int j) {
    if (v > val[CAP - 1]) {                                 // strict: an equal value with a larger index never displaces
      val[CAP - 1] = v; idx[CAP - 1] = j;
#pragma unroll
      for (int i = CAP - 1; i > 0; --i) {
        if (val[i] > val[i - 1]) {
          const float tv = val[i]; val[i] = val[i - 1]; val[i - 1] = tv;
          const int ti = idx[i]; idx[i] = idx[i - 1]; idx[i - 1] = ti;
        }
      }
    }
  };
  // each lane owns 4 consecutive columns per 256-column stripe (16-byte loads, two stripes in flight); within a lane the
  // columns arrive in ascending order, which is what the tie rule needs
  int j0 = 0;
  if ((lds & 3) == 0) {
    const int nvec = n & ~511;
    for (; j0 < nvec; j0 += 512) {
      const f32x4 a = *(const f32x4*)(s + j0 + lane * 4);
      const f32x4 b = *(const f32x4*)(s + j0 + 256 + lane * 4);
      const float thr = val[CAP - 1];
      if (fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))) > thr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(a[e], j0 + lane * 4 + e);
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(b[e], j0 + 256 + lane * 4 + e);
      }
    }
  }
  for (int j = j0 + lane; j < n; j += 64) offer(s[j], j);   // tail (and unaligned rows): one column per lane per pass
  const int own = labels[row0 + r];
  unsigned agree = 0;
  for (int t = 0; t <= k; ++t) {
    float bv = val[0];
    int bi = idx[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (bi == INT_MAX) break;                               // fewer than k+1 candidates
    if (idx[0] == bi) {                                     // the winning lane pops its head
#pragma unroll
      for (int i = 0; i < CAP - 1; ++i) { val[i] = val[i + 1]; idx[i] = idx[i + 1]; }
      val[CAP - 1] = -INFINITY; idx[CAP - 1] = INT_MAX;
    }
    if (t > 0) agree += (labels[bi] == own) ? 1u : 0u;
  }
  if (lane == 0 && agree) atomicAdd(count, (unsigned long long)agree);
}

__global__ void zero_count_k(unsigned long long* c) { *c = 0ull; }

// ---- linear probe: NLLLoss(log_softmax(logits)) + accuracy (utils/eval_utils.py:52-54), one wavefront per row ----------------
// stats[0] += sum of per-row losses, stats[1] += number of rows whose arg-max is the label; dlogits = (softmax - onehot) * gscale
__global__ void __launch_bounds__(256) softmax_ce_k(int N, int C, int ld, const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                    float gscale, float* __restrict__ dlogits, float* __restrict__ row_loss, int32_t* __restrict__ row_hit) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  const float* row = logits + (int64_t)r * ld;
  float mx = -INFINITY;
  int arg = 0;
  for (int c = lane; c < C; c += 64) { const float v = row[c]; if (v > mx) { mx = v; arg = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ov > mx || (ov == mx && oa < arg)) { mx = ov; arg = oa; }          // first maximum, like argmax
  }
  float sm = 0.f;
  for (int c = lane; c < C; c += 64) sm += expf(row[c] - mx);
  sm = wave_sum(sm);
  const float lse = mx + logf(sm);
  const int y = labels[r];
  if (dlogits) {
    for (int c = lane; c < ld; c += 64)
      dlogits[(int64_t)r * ld + c] = c < C ? (expf(row[c] - lse) - (c == y ? 1.f : 0.f)) * gscale : 0.f;
  }
  // a label outside [0, C) never reads past the row: its loss is NaN, which poisons the reported mean (the host wrapper refuses
  // such labels up front; NLLLoss in the reference raises)
  if (lane == 0) { row_loss[r] = (unsigned)y < (unsigned)C ? lse - row[y] : __builtin_nanf(""); row_hit[r] = arg == y ? 1 : 0; }
}
__global__ void ce_reduce_k(int N, const float* __restrict__ row_loss, const int32_t* __restrict__ row_hit, float* __restrict__ stats) {
  __shared__ double sl[256];
  __shared__ int sh[256];
  double a = 0.0;
  int h = 0;
  for (int i = threadIdx.x; i < N; i += 256) { a += (double)row_loss[i]; h += row_hit[i]; }
  sl[threadIdx.x] = a; sh[threadIdx.x] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sh[threadIdx.x] += sh[threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) { stats[0] = (float)(sl[0] / (double)N); stats[1] = (float)sh[0] / (float)N; }
}

int64_t chunk_rows(int64_t n) {
  // one chunk of S must satisfy the convolution's element limit (< 2^29 - 2^22 outputs) and stay cache/HBM friendly
  int64_t r = ((1ll << 29) - (1ll << 23)) / n;
  if (r > 4096) r = 4096;
  if (r > n) r = n;
  return r < 1 ? 1 : r;
}

}  // namespace

extern "C" size_t ssv_knn_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)(chunk_rows(n) * n) * sizeof(float) + 256;
}

extern "C" int ssv_knn_label_agreement(int64_t n, int32_t d, const float* z, const int32_t* labels, int32_t k,
                                       unsigned long long* count, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(n >= 2 && n < (1ll << 31) / 2 && d > 0 && d % 4 == 0, "ssv_knn_label_agreement: need 2 <= n < 2^30 and d %% 4 == 0 (got n=%lld d=%d)", (long long)n, d);
  SSV_REQUIRE(k >= 1 && k <= 63 && k < n, "ssv_knn_label_agreement: need 1 <= k <= min(63, n-1) (got %d)", k);
  SSV_REQUIRE(z && labels && count && ws, "ssv_knn_label_agreement: null pointer");
  SSV_REQUIRE(ws_bytes >= ssv_knn_workspace_bytes(n), "ssv_knn_label_agreement: workspace too small (%zu < %zu)", ws_bytes, ssv_knn_workspace_bytes(n));
  SSV_REQUIRE((((uintptr_t)z | (uintptr_t)ws) & 15) == 0, "ssv_knn_label_agreement: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* S = (float*)ws;
  const int64_t cr = chunk_rows(n);
  {
    ProfScope ps(SSV_PROF_MISC, s);
    hipLaunchKernelGGL(zero_count_k, dim3(1), dim3(1), 0, s, count);
  }
  for (int64_t r0 = 0; r0 < n; r0 += cr) {
    const int rows = (int)((n - r0 < cr) ? (n - r0) : cr);
    ssv_conv_desc cd = {};            // fp32-MFMA arithmetic: the reference's search is exact inner products, and integer-valued features must count bit-exactly
    cd.N = rows; cd.H = 1; cd.W = 1; cd.C = d; cd.K = (int32_t)n; cd.R = 1; cd.S = 1; cd.stride = 1; cd.pad = 0; cd.Ho = 1; cd.Wo = 1;
    if (int rc = ssv_conv2d_fwd(&cd, z + r0 * d, z, nullptr, nullptr, S, stream)) return rc;      // S[rows, n] = Z[r0:r0+rows] Z^T
    ProfScope ps(SSV_PROF_MISC, s);
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (k <= 20)
      hipLaunchKernelGGL(knn_agree_k<21>, grid, dim3(256), 0, s, S, n, (int)n, rows, (int)r0, labels, k, count);
    else
      hipLaunchKernelGGL(knn_agree_k<64>, grid, dim3(256), 0, s, S, n, (int)n, rows, (int)r0, labels, k, count);
    SSV_CHECK_LAUNCH("knn_agree_k");
  }
  return SSV_OK;
}

extern "C" size_t ssv_softmax_ce_workspace_bytes(int32_t N) { return N > 0 ? (size_t)N * 8 : 0; }

extern "C" int ssv_softmax_ce_fwd_bwd(int32_t N, int32_t C, int32_t ld, const float* logits, const int32_t* labels, float* stats,
                                      float* dlogits, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(N > 0 && C > 0 && ld >= C && logits && labels && stats && ws, "ssv_softmax_ce_fwd_bwd: bad arguments");
  SSV_REQUIRE(ws_bytes >= ssv_softmax_ce_workspace_bytes(N), "ssv_softmax_ce_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  float* row_loss = (float*)ws;
  int32_t* row_hit = (int32_t*)(row_loss + N);
  hipLaunchKernelGGL(softmax_ce_k, dim3(cdiv(N, 4)), dim3(256), 0, s, N, C, ld, logits, labels, 1.f / (float)N, dlogits, row_loss, row_hit);
  hipLaunchKernelGGL(ce_reduce_k, dim3(1), dim3(256), 0, s, N, (const float*)row_loss, (const int32_t*)row_hit, stats);
  SSV_CHECK_LAUNCH("ssv_softmax_ce_fwd_bwd");
  return SSV_OK;
}
