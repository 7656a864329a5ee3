// Can the fp32 VALU (v_pk_fma_f32, the same 64 FLOP / cycle / SIMD peak as the fp32 MFMA) add GEMM throughput BESIDE the matrix pipe?
// An MFMA occupies the matrix pipe for 64 cycles but the issue port for 4, so other waves' VALU instructions can issue meanwhile.
// Workgroup = 4 "matrix" waves (each a 64x64 tile on v_mfma_f32_32x32x2_f32, operands from LDS by ds_read_b128, as csrc/conv_mfma.hip)
// + NV "vector" waves (each a 32x64 tile by outer products on v_pk_fma_f32: per k 4 A values + 8 B values per lane -> 16 packed FMAs,
// operands from the same LDS images), one barrier per 32-deep k-tile, operands resident in LDS (no global traffic: the upper bound).
// Reports wall time, the TFLOP/s of each kind and the in-kernel clock for MFMA only, VALU only and both.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/hybrid_probe.hip -o tools/probe/bin/hybrid_probe && tools/probe/bin/hybrid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDT = 36;

// MW matrix waves (0 or 4) + NV vector waves; A image rows 0..127, B image rows 128..255 (+64 extra B rows for the vector waves)
template <int MW, int NV>
__global__ void __launch_bounds__((MW + NV) * 64) hybrid(const float* __restrict__ g, float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[320 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 320 * LDT; i += (MW + NV) * 64) lds[i] = g[(blockIdx.x & 31) * 320 * LDT + i];
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if (wave < MW) {
    const int wr0 = (wave >> 1) * 64, wc0 = 128 + (wave & 1) * 64;
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
    // vector wave v: rows 32v..32v+31 of A x the 64 extra B rows (256..319).  Lane (ly = lane >> 3, lx = lane & 7) owns a 4 x 8 patch:
    // rows 32v + 4*ly + {0..3}, columns 8*lx + {0..7} -> per k: 4 A values, 8 B values, 16 packed FMAs
    const int v = wave - MW;
    const int ly = lane >> 3, lx = lane & 7;
    const float* Ap = &lds[(32 * v + 4 * ly) * LDT];
    const float* Bp = &lds[(256 + 8 * lx) * LDT];
    f32x2 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {            // 4 k values per step: operands as ds_read_b128 along k
        f32x4 a[4], b[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ap + i * LDT + 4 * k4);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bp + j * LDT + 4 * k4);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x2 bb = {b[2 * j][t], b[2 * j + 1][t]};
              const f32x2 aa = {a[i][t], a[i][t]};
              acc[i][j] = __builtin_elementwise_fma(aa, bb, acc[i][j]);
            }
      }
      __syncthreads();
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1];
  }
  if (tid == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0;
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int MW, int NV> void run(const float* g, float* out, unsigned long long* clk, int iters, int wgpc) {
  const int blocks = 256 * wgpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 30; ++r) hipLaunchKernelGGL((hybrid<MW, NV>), dim3(blocks), dim3((MW + NV) * 64), 0, 0, g, out, clk, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((hybrid<MW, NV>), dim3(blocks), dim3((MW + NV) * 64), 0, 0, g, out, clk, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<unsigned long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz(blocks);
  for (int b = 0; b < blocks; ++b) ghz[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;
  std::sort(ghz.begin(), ghz.end());
  const double fm = (double)blocks * MW * iters * 64.0 * 4096.0;                 // 64 MFMAs of 4096 FLOP per matrix wave and k-tile
  const double fv = (double)blocks * NV * iters * 32.0 * 64.0 * 32.0 * 2.0;      // 32 x 64 outputs x 32 k x 2 FLOP per vector wave and k-tile
  printf("%d matrix + %d vector waves, %d workgroups/CU: %.3f ms | MFMA %.1f + VALU %.1f = %.1f TFLOP/s | clock %.3f GHz\n", MW, NV, wgpc, ms,
         fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9, ghz[blocks / 2]);
}

int main() {
  float *g, *out; unsigned long long* clk;
  const size_t n = (size_t)32 * 320 * LDT;
  (void)hipMalloc(&g, n * 4); (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&clk, 2 * 1024 * 8);
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  (void)hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int wgpc : {1, 2}) {
    run<4, 0>(g, out, clk, iters, wgpc);
    run<0, 4>(g, out, clk, iters, wgpc);
    run<4, 2>(g, out, clk, iters, wgpc);
    run<4, 4>(g, out, clk, iters, wgpc);
  }
  return 0;
}
