// What the bf16 matrix pipe sustains on RANDOM operands (the clock the chip holds under that load): v_mfma_f32_32x32x16_bf16 from registers, 4 accumulators per wave,
// 1 / 2 waves per SIMD, against v_mfma_f32_32x32x2_f32 in the same loop shape.  Decides what an fp32 GEMM by bf16 splitting (gemm_split_bf16_probe.hip) can reach:
// TERMS bf16 MFMAs per fp32 product -> sustained bf16 rate / TERMS.
// hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probe/mfma_bf16_rate_probe.hip -o tools/probe/bin/mfma_bf16_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool BF16>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // random operands: bf16 values in [1, 2) x random sign with random mantissas (each 16-bit half: sign | exponent 127 | 7 random mantissa bits)
  u32x4 ra[4], rb[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 4; ++e) {
      const unsigned h1 = hash(threadIdx.x * 977u + blockIdx.x * 131u + i * 17u + e), h2 = hash(h1 + 12345u);
      ra[i][e] = (h1 & 0x807F807Fu) | 0x3F803F80u;
      rb[i][e] = (h2 & 0x807F807Fu) | 0x3F803F80u;
    }
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = __uint_as_float((hash(threadIdx.x + i * 7919u) & 0x807FFFFFu) | 0x3F800000u); fb[i] = __uint_as_float((hash(threadIdx.x * 31u + i) & 0x807FFFFFu) | 0x3F800000u); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (BF16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(t + j) & 3]), __builtin_bit_cast(bf16x8, rb[(t * 3 + j) & 3]), acc[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(t + j) & 7], fb[(t * 3 + j) & 7], acc[j], 0, 0, 0);
      }
    // keep the sums bounded (and the data toggling): scale the accumulators down now and then
    if ((it & 63) == 63)
#pragma unroll
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] *= 1e-3f;
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool BF16> float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<BF16>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<BF16>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int wps : {1, 2}) {
    const int blocks = 256 * wps;
    const int it_b = 4000, it_f = 500;
    const float tb = run<true>(out, blocks, it_b), tf = run<false>(out, blocks, it_f);
    const double fb = (double)blocks * 4 * it_b * 32 * (2.0 * 32 * 32 * 16), ff = (double)blocks * 4 * it_f * 32 * (2.0 * 32 * 32 * 2);
    printf("waves/SIMD %d, random operands from registers: bf16 32x32x16 %.3f ms = %.0f TFLOP/s (/6 = %.0f, /9 = %.0f) | f32 32x32x2 %.3f ms = %.1f TFLOP/s\n",
           wps, tb, fb / tb / 1e9, fb / tb / 1e9 / 6, fb / tb / 1e9 / 9, tf, ff / tf / 1e9);
  }
  return 0;
}
