// How fast do DEPENDENT v_mfma_f32_32x32x2_f32 chains issue?  NCH independent accumulators per wave (1 = every MFMA waits for the one before), WPS waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_chain_probe.hip -o tools/probe/bin/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[NCH];
  for (int j = 0; j < NCH; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = a0 + i + threadIdx.x; b[i] = b0 - i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 32 / NCH; ++t)
#pragma unroll
      for (int j = 0; j < NCH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + j) & 7], b[(t * 3 + j) & 7], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NCH; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH> float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 400;                       // 400 x 32 MFMAs per wave
  for (int wps : {1, 2, 3, 4}) {               // one 256-thread block = one wave on each SIMD of a CU; wps blocks per CU resident (tiny kernels: up to 8 fit)
    const int blocks = 256 * wps;
    const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
    float t1 = run<1>(out, blocks, iters), t2 = run<2>(out, blocks, iters), t4 = run<4>(out, blocks, iters);
    printf("waves/SIMD %d: 1 chain %.3f ms (%.1f TF) | 2 chains %.3f (%.1f) | 4 chains %.3f (%.1f)\n", wps, t1, flop / t1 / 1e9, t2, flop / t2 / 1e9, t4, flop / t4 / 1e9);
  }
  return 0;
}
