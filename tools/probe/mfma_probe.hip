// Ablation probe: what does one wave per SIMD lose around a 16-MFMA (32x32x2 f32) group?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i & 7) * 0.01f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a[2][4] = {{1.f, 2.f, 3.f, 4.f}, {1.5f, 2.5f, 3.5f, 4.5f}}, b[2][4] = {{.1f, .2f, .3f, .4f}, {.5f, .6f, .7f, .8f}};
  const float* gp = g + (size_t)blockIdx.x * 4096 + tid * 4;
  f32x4 st[8];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const f32x4*>(gp + ((it * 8 + i) & 63) * 65536);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE >= 1) {
        const int row = (wave >> 1) * 64 + (lane & 31), col = (wave & 1) * 64 + (lane & 31);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&lds[row * 36 + ks * 8 + 4 * (lane >> 5)]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&lds[(row + 32) * 36 + ks * 8 + 4 * (lane >> 5)]);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lds[(128 + col) * 36 + ks * 8 + 4 * (lane >> 5)]);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&lds[(128 + col + 32) * 36 + ks * 8 + 4 * (lane >> 5)]);
        for (int t = 0; t < 4; ++t) { a[0][t] = v0[t]; a[1][t] = v1[t]; b[0][t] = w0[t]; b[1][t] = w1[t]; }
      }
      if (MODE >= 3 && ks == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&lds[256 * 36 + ((tid >> 3) + 32 * i) * 36 + (tid & 7) * 4]) = st[i];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 1][t], b[j & 1][t], acc[j], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> float run(const float* g, float* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, g, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, g, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)64 * 65536 * 4 + 256 * 4096 * 4 * 16); hipMalloc(&out, 2048 * 256 * 4);
  hipMemset(g, 0, (size_t)64 * 65536 * 4 + 256 * 4096 * 4 * 16);
  const int iters = 72;   // 72 k-tiles x 64 MFMA = 4608 MFMA per wave
  const double flop_per_block = 4.0 * 4608 * 2 * 32 * 32 * 2;
  for (int blocks : {256, 512, 768}) {
    float t0 = run<0>(g, out, blocks, iters), t1 = run<1>(g, out, blocks, iters), t2 = run<2>(g, out, blocks, iters), t3 = run<3>(g, out, blocks, iters);
    printf("blocks %4d: mfma only %.3f ms (%.1f TF) | +lds frags %.3f (%.1f) | +barrier %.3f (%.1f) | +global+ds_write %.3f (%.1f)\n", blocks,
           t0, blocks * flop_per_block / t0 / 1e9, t1, blocks * flop_per_block / t1 / 1e9, t2, blocks * flop_per_block / t2 / 1e9, t3, blocks * flop_per_block / t3 / 1e9);
  }
  return 0;
}
