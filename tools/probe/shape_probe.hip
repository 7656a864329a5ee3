// Which fp32 MFMA shape does the chip run faster BY WALL CLOCK on random data: v_mfma_f32_32x32x2_f32 or v_mfma_f32_16x16x4_f32?
// Both deliver 64 FLOP / cycle / SIMD; MI355X_MICROARCH.md ("DVFS give-back", item 7) reports that the clock the chip holds under an
// MFMA-dense loop can depend on the shape.  Same wave tile (64x64 outputs per wave, 4 waves = 128x128 per workgroup), every operand
// re-read from LDS with ds_read_b128 (the ROWK image of csrc/conv_mfma.hip), one barrier per 32-deep k-tile, random operands.
// Reports TFLOP/s by wall clock and the in-kernel clock (s_memtime / s_memrealtime, 100 MHz reference) per shape and occupancy.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/shape_probe.hip -o /tmp/shape_probe && /tmp/shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDT = 36;

template <int SHAPE, int WGPC>
__global__ void __launch_bounds__(256, WGPC) probe(const float* __restrict__ g, float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[256 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256 * LDT; i += 256) lds[i] = g[(blockIdx.x & 63) * 256 * LDT + i];
  __syncthreads();
  const int wr0 = (wave >> 1) * 64, wc0 = 128 + (wave & 1) * 64;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, h = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const f32x4*>(&lds[(wr0 + 32 * i + l31) * LDT + ks * 8 + 4 * h]);
          b[i] = *reinterpret_cast<const f32x4*>(&lds[(wc0 + 32 * i + l31) * LDT + ks * 8 + 4 * h]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int l15 = lane & 15, q = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const f32x4*>(&lds[(wr0 + 16 * i + l15) * LDT + ks * 16 + 4 * q]);
          b[i] = *reinterpret_cast<const f32x4*>(&lds[(wc0 + 16 * i + l15) * LDT + ks * 16 + 4 * q]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  if (tid == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0;
  }
  out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int WGPC> void run(const float* g, float* out, unsigned long long* clk, int iters, double seconds) {
  const int blocks = 256 * WGPC;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SHAPE, WGPC>), dim3(blocks), dim3(256), 0, 0, g, out, clk, iters);
  hipDeviceSynchronize();
  // warm the chip into its steady clock: back-to-back launches for `seconds`, then time 5
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<SHAPE, WGPC>), dim3(blocks), dim3(256), 0, 0, g, out, clk, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float one; hipEventElapsedTime(&one, e0, e1);
  const int warm = std::max(1, (int)(seconds * 1000.0 / one));
  for (int r = 0; r < warm; ++r) hipLaunchKernelGGL((probe<SHAPE, WGPC>), dim3(blocks), dim3(256), 0, 0, g, out, clk, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<SHAPE, WGPC>), dim3(blocks), dim3(256), 0, 0, g, out, clk, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz(blocks);
  for (int b = 0; b < blocks; ++b) ghz[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;
  std::sort(ghz.begin(), ghz.end());
  const double flop = (double)blocks * 4.0 * iters * 64.0 * 4096.0;     // 64 x (32x32x2) MFMA-equivalents per wave and k-tile
  printf("shape %2d, %d workgroups/CU: %.3f ms  %.1f TFLOP/s  in-kernel clock median %.3f GHz (min %.3f max %.3f)\n", SHAPE, WGPC, ms, flop / ms / 1e9,
         ghz[blocks / 2], ghz.front(), ghz.back());
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  float *g, *out; unsigned long long* clk;
  const size_t n = (size_t)64 * 256 * LDT;
  hipMalloc(&g, n * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 2 * 1024 * 8);
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
  const int iters = 4000;        // ~7 ms per workgroup-round at 1 workgroup/CU
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 1>(g, out, clk, iters, seconds); run<16, 1>(g, out, clk, iters, seconds);
    run<32, 2>(g, out, clk, iters, seconds); run<16, 2>(g, out, clk, iters, seconds);
    run<32, 3>(g, out, clk, iters, seconds); run<16, 3>(g, out, clk, iters, seconds);
  }
  return 0;
}
