// Probe (round 6): which main-loop STRUCTURE carries the fp32-accurate product on the bf16 matrix pipe (3 bf16 pieces per operand, 6 piece products, fp32 accumulate)
// closest to the bare compute loop's 249 TFLOP/s?  Y[rows][K] = A[rows][C] . W[K][C]^T, A fp32 in HBM (split while it is staged), W split ONCE into three bf16 planes
// Wp[3][K][C] ahead of the launch (what the library does once per step and weight).  One kernel template, instantiated per candidate:
//   BM x BN x BK tile, NSTAGE LDS stages (1 = the library's register-prefetch loop with two barriers per k-tile; 2 = write tile k + 1 to the other stage after the ONE
//   barrier of tile k, loads of tile k + 2 in flight), M16 = v_mfma_f32_16x16x32_bf16 instead of 32x32x16, WGPC = resident workgroups per CU compiled for,
//   NOSPLIT = timing what-if (every plane of A is its first conversion: no residual arithmetic; results wrong).
// LDS rows are UNPADDED (BK bf16 = 64 or 32 bytes) with the 16-byte slots XOR-swizzled by the row, so the 128 x 128 x 32 tile is 48 KB (three workgroups per CU).
// Prints ms, TFLOP/s of 2 * rows * C * K and the relative l2 error against fp64 on 16 sampled rows.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/gemm_split_tuned_probe.hip -o tools/probe/bin/gemm_split_tuned_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int OOB_OFF = (int)0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000); }
__device__ __forceinline__ u32x4 bload4(rsrc_t rs, int voff, int soff) { return (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0); }

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// 16-byte slot swizzle of an unpadded LDS row (conflict-free ds_read_b128 over the instruction's four 16-lane groups; derivation in the round-6 design notes)
template <int BK, bool M16>
__device__ __forceinline__ int swz(int row) {
  if constexpr (BK == 32) return M16 ? (((row >> 3) & 1) * 3) : ((row >> 2) & 3);
  else return (row >> 3) & 1;
}

// a = p0 + p1 + p2 (round to nearest even; residuals exact): four floats -> three pairs of packed dwords
template <bool NOSPLIT>
__device__ __forceinline__ void split4(const f32x4& v, u32x2 (&pl)[3]) {
  f32x2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf16x2)), p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf16x2));
    if (q < 2 && !NOSPLIT) {
      x0 -= f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
      x1 -= f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
    }
    pl[q] = u32x2{p0, p1};
  }
}

template <int BM, int BN, int BK, int NSTAGE, bool M16, int WGPC, bool NOSPLIT, int ACCM = 0, int PF = 1>
__global__ void __launch_bounds__(256, WGPC)
ksplit(int rows, int C, int K, int tiles_n, const float* __restrict__ A, const unsigned short* __restrict__ Wp, float* __restrict__ Y,
       long long bsA, long long bsW, long long bsY) {
  constexpr int NT = 256, WGM = 2, WGN = 2;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MT = M16 ? 16 : 32;
  constexpr int TM = WTM / MT, TN = WTN / MT;
  constexpr int ROWB = BK * 2;
  constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
  constexpr int STAGE_BYTES = 3 * (A_PLANE + B_PLANE);
  constexpr int AP = BM * BK / 4 / NT;                 // float4 of A per thread and k-tile
  constexpr int BP = BN * BK / 8 / NT;                 // 16-byte pieces of one W plane per thread and k-tile
  static_assert(AP >= 1 && BP >= 1, "tile too small for 256 threads");
  static_assert(!M16 || BK == 32, "16x16x32 contracts 32 per instruction");
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  A += (size_t)blockIdx.y * bsA; Wp += (size_t)blockIdx.y * bsW; Y += (size_t)blockIdx.y * bsY;
  const rsrc_t ra_ = make_rsrc(A, (unsigned)((size_t)rows * C * 4));
  const rsrc_t rw_ = make_rsrc(Wp, (unsigned)((size_t)3 * K * C * 2));

  // staging offsets
  int aoff[AP], alds[AP], boff[BP], blds[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int idx = tid + NT * i, row = idx / (BK / 4), k4 = idx % (BK / 4);
    aoff[i] = (m0 + row < rows) ? ((m0 + row) * C + k4 * 4) * 4 : OOB_OFF;
    alds[i] = row * ROWB + (((k4 >> 1) ^ swz<BK, M16>(row)) << 4) + (k4 & 1) * 8;
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int idx = tid + NT * i, row = idx / (BK / 8), k8 = idx % (BK / 8);
    boff[i] = (n0 + row < K) ? ((n0 + row) * C + k8 * 8) * 2 : OOB_OFF;
    blds[i] = row * ROWB + ((k8 ^ swz<BK, M16>(row)) << 4);
  }
  u32x4 ra[AP], rb[3][BP], ra2[PF == 2 ? AP : 1], rb2[PF == 2 ? 3 : 1][BP];      // PF 2: a second register stage, loads issued TWO k-tiles ahead
  const int plane_bytes = K * C * 2;
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = bload4(ra_, aoff[i], kt * BK * 4);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[q][i] = bload4(rw_, boff[i] == OOB_OFF ? OOB_OFF : boff[i] + q * plane_bytes, kt * BK * 2);
  };
  auto store_tile = [&](int stage) {
    unsigned char* base = lds + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      u32x2 pl[3];
      split4<NOSPLIT>(__builtin_bit_cast(f32x4, ra[i]), pl);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(base + q * A_PLANE + alds[i]) = pl[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < BP; ++i) *reinterpret_cast<u32x4*>(base + 3 * A_PLANE + q * B_PLANE + blds[i]) = rb[q][i];
  };

  // fragment addresses (byte offsets inside a plane)
  using acc_t = typename std::conditional<M16, f32x4, f32x16>::type;
  // ACCM: 0 one accumulator, smallest terms first; 1 TWO accumulators (a0b0 alone in the main one, the five small terms in a second one, added at the end);
  //       2 one accumulator, largest term first; 3 nine terms, one accumulator
  acc_t acc[TM][TN], lo[ACCM == 1 ? TM : 1][ACCM == 1 ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[i][j] = acc_t{}; if constexpr (ACCM == 1) lo[i][j] = acc_t{}; }
  const int fr = M16 ? (lane & 15) : (lane & 31), fh = M16 ? (lane >> 4) : (lane >> 5);
  auto mma_tile = [&](int stage) {
    const unsigned char* base = lds + stage * STAGE_BYTES;
    constexpr int NSLAB = M16 ? 1 : BK / 16;
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
      const int slot = M16 ? fh : (BK == 32 ? 2 * s + fh : fh);
      bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wm + MT * t + fr;
          fa[q][t] = *reinterpret_cast<const bf16x8*>(base + q * A_PLANE + row * ROWB + ((slot ^ swz<BK, M16>(row)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wn + MT * t + fr;
          fb[q][t] = *reinterpret_cast<const bf16x8*>(base + 3 * A_PLANE + q * B_PLANE + row * ROWB + ((slot ^ swz<BK, M16>(row)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // D[channel][row] += w-plane P . a-plane Q; smallest terms first
          if constexpr (M16) {
#define MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[P][j], fa[Q][i], acc[i][j], 0, 0, 0)
#define ML(P, Q) lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[P][j], fa[Q][i], lo[i][j], 0, 0, 0)
            if constexpr (ACCM == 1) { ML(2, 0); ML(0, 2); ML(1, 1); ML(1, 0); ML(0, 1); MM(0, 0); }
            else if constexpr (ACCM == 2) { MM(0, 0); MM(0, 1); MM(1, 0); MM(1, 1); MM(0, 2); MM(2, 0); }
            else if constexpr (ACCM == 3) { MM(2, 2); MM(2, 1); MM(1, 2); MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0); }
            else { MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0); }
#undef MM
#undef ML
          } else {
#define MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][j], fa[Q][i], acc[i][j], 0, 0, 0)
            MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0);
#undef MM
          }
        }
    }
  };

  const int nkt = C / BK;
  if constexpr (NSTAGE == 1 && PF == 2) {
    auto load2 = [&](int kt) {
#pragma unroll
      for (int i = 0; i < AP; ++i) ra2[i] = bload4(ra_, aoff[i], kt * BK * 4);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < BP; ++i) rb2[q][i] = bload4(rw_, boff[i] == OOB_OFF ? OOB_OFF : boff[i] + q * plane_bytes, kt * BK * 2);
    };
    auto swap_in = [&]() {
#pragma unroll
      for (int i = 0; i < AP; ++i) ra[i] = ra2[i];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < BP; ++i) rb[q][i] = rb2[q][i];
    };
    load_tile(0);
    store_tile(0);
    load_tile(1);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      load2(kt + 2);                            // two tiles ahead: a tile's bytes are in flight for two k-tiles of MFMAs
      mma_tile(0);
      __syncthreads();
      store_tile(0);                            // tile kt + 1 (loaded one iteration ago)
      swap_in();
      __syncthreads();
    }
  } else if constexpr (NSTAGE == 1) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      load_tile(kt + 1);                       // past the last tile: in-buffer junk or hardware zeros, stored and never read
      mma_tile(0);
      __syncthreads();
      store_tile(0);
      __syncthreads();
    }
  } else {
    load_tile(0);
    store_tile(0);
    load_tile(1);
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();                         // tile kt is complete in stage kt & 1; every reader of the other stage (tile kt - 1) is done
      store_tile((kt + 1) & 1);
      load_tile(kt + 2);
      mma_tile(kt & 1);
    }
  }

  // epilogue: the MFMA took the w fragment as its A operand, so a lane holds consecutive channels of ONE output row
  if constexpr (ACCM == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += lo[i][j];
  }
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wm + 16 * i + fr;
      if (row >= rows) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + 16 * j + 4 * fh;
        if (col < K) *reinterpret_cast<f32x4*>(Y + (size_t)row * K + col) = acc[i][j];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wm + 32 * i + fr;
      if (row >= rows) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = n0 + wn + 32 * j + 8 * q + 4 * fh;
          if (col < K) {
            f32x4 v;
            if constexpr (!M16) v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(Y + (size_t)row * K + col) = v;
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ unsigned short bf16_rne(float x) { unsigned u = __float_as_uint(x); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
__global__ void presplit_k(size_t n, const float* __restrict__ x, unsigned short* __restrict__ p0, unsigned short* __restrict__ p1, unsigned short* __restrict__ p2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const unsigned short a = bf16_rne(v); const float r1 = v - __uint_as_float((unsigned)a << 16);
  const unsigned short b = bf16_rne(r1); const float r2 = r1 - __uint_as_float((unsigned)b << 16);
  p0[i] = a; p1[i] = b; p2[i] = bf16_rne(r2);
}
// deterministic pseudo-normal fill (sum of four uniforms, variance 1), optionally rectified
__global__ void fill_k(size_t n, float* x, unsigned seed, float scale, int nonneg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
  float acc = 0.f;
  for (int t = 0; t < 4; ++t) { s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32; acc += (float)(s & 0xFFFFFF) / 16777216.f - 0.5f; }
  float v = acc * 1.7320508f * scale;
  if (nonneg) v = v > 0.f ? v : 0.f;
  x[i] = v;
}

template <typename F> static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

struct Shape { const char* name; int batch; long long rows; int C, K; int nonneg; };

int main(int argc, char** argv) {
  const Shape shapes[] = {
    {"4096^3", 1, 4096, 4096, 4096, 0},
    {"4096^3 non-negative A", 1, 4096, 4096, 4096, 1},
    {"long contraction, non-negative A", 1, 2048, 32768, 256, 1},
    {"r50 56x56 1x1 64->256", 1, 1605632, 64, 256, 1},
    {"r50 56x56 1x1 256->64", 1, 1605632, 256, 64, 1},
    {"r50 28x28 1x1 512->128", 1, 401408, 512, 128, 1},
    {"r50 28x28 1x1 128->512", 1, 401408, 128, 512, 1},
    {"r50 14x14 1x1 1024->256", 1, 100352, 1024, 256, 1},
    {"r50 14x14 1x1 256->1024", 1, 100352, 256, 1024, 1},
    {"r50 7x7 1x1 2048->512", 1, 25088, 2048, 512, 1},
    {"r50 7x7 1x1 512->2048", 1, 25088, 512, 2048, 1},
    {"wino 36 x [25088x128].[128x128]", 36, 25088, 128, 128, 0},
    {"wino 36 x [8192x256].[256x256]", 36, 8192, 256, 256, 0},
    {"wino 36 x [2048x512].[512x512]", 36, 2048, 512, 512, 0},
    {"vit 50432 x 384 -> 1152", 1, 50432, 384, 1152, 0},
    {"vit 50432 x 384 -> 1536", 1, 50432, 384, 1536, 0},
    {"vit 50432 x 1536 -> 384", 1, 50432, 1536, 384, 0},
  };
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  for (int si = 0; si < (int)(sizeof(shapes) / sizeof(shapes[0])); ++si) {
    if (only >= 0 && si != only) continue;
    const Shape& sh = shapes[si];
    const size_t na = (size_t)sh.batch * sh.rows * sh.C, nw = (size_t)sh.batch * sh.K * sh.C, ny = (size_t)sh.batch * sh.rows * sh.K;
    float *A, *W, *Y; unsigned short* Wp;
    hipMalloc(&A, na * 4); hipMalloc(&W, nw * 4); hipMalloc(&Y, ny * 4); hipMalloc(&Wp, (size_t)3 * nw * 2);
    hipLaunchKernelGGL(fill_k, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, na, A, 17u, 1.f, sh.nonneg);
    hipLaunchKernelGGL(fill_k, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, nw, W, 91u, 0.05f, 0);
    // planes per batch element: [3][K][C] (bsW = 3 * K * C)
    for (int b = 0; b < sh.batch; ++b) {
      const size_t n1 = (size_t)sh.K * sh.C;
      unsigned short* base = Wp + (size_t)b * 3 * n1;
      hipLaunchKernelGGL(presplit_k, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, 0, n1, W + (size_t)b * n1, base, base + n1, base + 2 * n1);
    }
    hipDeviceSynchronize();
    // fp64 reference: 16 sampled rows of batch element 0 (and of the LAST batch element when batched)
    const int nrows = 16;
    const int bref = sh.batch - 1;
    std::vector<long long> which(nrows);
    for (int i = 0; i < nrows; ++i) which[i] = (long long)(((unsigned long long)i * 2654435761ull + 12345) % (unsigned long long)sh.rows);
    which[nrows - 1] = sh.rows - 1;
    std::vector<float> ha((size_t)nrows * sh.C), hw((size_t)sh.K * sh.C);
    for (int i = 0; i < nrows; ++i) hipMemcpy(&ha[(size_t)i * sh.C], A + ((size_t)bref * sh.rows + which[i]) * sh.C, (size_t)sh.C * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hw.data(), W + (size_t)bref * sh.K * sh.C, hw.size() * 4, hipMemcpyDeviceToHost);
    std::vector<double> ref((size_t)nrows * sh.K);
    for (int i = 0; i < nrows; ++i)
      for (int j = 0; j < sh.K; ++j) {
        double s = 0;
        for (int k = 0; k < sh.C; ++k) s += (double)ha[(size_t)i * sh.C + k] * (double)hw[(size_t)j * sh.C + k];
        ref[(size_t)i * sh.K + j] = s;
      }
    const double flop = 2.0 * sh.batch * sh.rows * sh.C * sh.K;
    const double bytes = 4.0 * (na + ny) + 6.0 * nw;
    printf("## %s: %d x [%lld x %d] . [%d x %d]^T  (%.1f GFLOP, %.2f GB = %.3f ms at 6.29 TB/s)\n", sh.name, sh.batch, sh.rows, sh.C, sh.K, sh.C, flop / 1e9, bytes / 1e9,
           bytes / 6.29e9);
    std::vector<float> hy((size_t)sh.K);
    auto report = [&](const char* name, float ms) {
      double num = 0, den = 0, bias = 0, mag = 0;
      for (int i = 0; i < nrows; ++i) {
        hipMemcpy(hy.data(), Y + ((size_t)bref * sh.rows + which[i]) * sh.K, (size_t)sh.K * 4, hipMemcpyDeviceToHost);
        for (int j = 0; j < sh.K; ++j) { const double r = ref[(size_t)i * sh.K + j], d = (double)hy[j] - r; num += d * d; den += r * r; bias += d * (r > 0 ? 1 : -1); mag += fabs(r); }
      }
      printf("  %-44s %8.3f ms  %7.1f TFLOP/s   error %.3e  signed (toward larger magnitude) %.2e\n", name, ms, flop / ms / 1e9, sqrt(num / den), bias / mag);
      fflush(stdout);
    };
#define RUN(NAME, BM_, BN_, BK_, NS_, M16_, WGPC_, NOSPLIT_, ...) do {                                                                                   \
      if (sh.C % BK_ == 0) {                                                                                                                          \
        hipMemset(Y, 0xff, ny * 4);                                                                                                                   \
        const int tiles_n = (sh.K + BN_ - 1) / BN_;                                                                                                   \
        const long long tiles_m = (sh.rows + BM_ - 1) / BM_;                                                                                          \
        const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)sh.batch);                                                                           \
        auto launch = [&]() { hipLaunchKernelGGL((ksplit<BM_, BN_, BK_, NS_, M16_, WGPC_, NOSPLIT_, ##__VA_ARGS__>), grid, dim3(256), 0, 0, (int)sh.rows, sh.C, sh.K, \
                                                 tiles_n, A, Wp, Y, (long long)sh.rows * sh.C, (long long)3 * sh.K * sh.C, (long long)sh.rows * sh.K); }; \
        const float ms = time_ms(launch, 10);                                                                                                          \
        hipError_t e = hipGetLastError();                                                                                                              \
        if (e != hipSuccess) printf("  %-44s launch error: %s\n", NAME, hipGetErrorString(e)); else report(NAME, ms);                                  \
      } } while (0)
    if (argc > 2 && atoi(argv[2]) == 1) {          // accumulate-mode study (16x16x32): what the way the six terms enter the accumulator does to the error
      RUN("128x128x32 16x16x32 wgpc2, 1 acc small first", 128, 128, 32, 1, true, 2, false, 0);
      RUN("128x128x32 16x16x32 wgpc2, 1 acc large first", 128, 128, 32, 1, true, 2, false, 2);
      RUN("128x128x32 16x16x32 wgpc2, 2 accumulators", 128, 128, 32, 1, true, 2, false, 1);
      RUN("128x128x32 16x16x32 wgpc2, 9 terms", 128, 128, 32, 1, true, 2, false, 3);
      RUN("256x128x32 16x16x32 wgpc1, 2 accumulators", 256, 128, 32, 1, true, 1, false, 1);
      RUN("128x128x32 2 acc, loads 2 tiles ahead, wgpc2", 128, 128, 32, 1, true, 2, false, 1, 2);
      RUN("128x128x32 1 acc, wgpc3", 128, 128, 32, 1, true, 3, false, 0);
      RUN("128x128x32 1 acc, loads 2 tiles ahead, wgpc3", 128, 128, 32, 1, true, 3, false, 0, 2);
      RUN("128x128x32 1 acc, loads 2 tiles ahead, wgpc2", 128, 128, 32, 1, true, 2, false, 0, 2);
    } else {
    RUN("128x128x32 1-stage 32x32x16 wgpc2", 128, 128, 32, 1, false, 2, false);
    RUN("128x128x32 1-stage 32x32x16 wgpc3", 128, 128, 32, 1, false, 3, false);
    RUN("128x128x32 1-stage 16x16x32 wgpc2", 128, 128, 32, 1, true, 2, false);
    RUN("128x128x32 1-stage 16x16x32 wgpc3", 128, 128, 32, 1, true, 3, false);
    RUN("128x128x16 2-stage 32x32x16 wgpc3", 128, 128, 16, 2, false, 3, false);
    RUN("128x128x16 2-stage 32x32x16 wgpc2", 128, 128, 16, 2, false, 2, false);
    RUN("128x128x32 2-stage 32x32x16 wgpc1", 128, 128, 32, 2, false, 1, false);
    RUN("128x128x32 2-stage 16x16x32 wgpc1", 128, 128, 32, 2, true, 1, false);
    RUN("256x128x32 1-stage 32x32x16 wgpc2", 256, 128, 32, 1, false, 2, false);
    RUN("256x128x32 1-stage 16x16x32 wgpc2", 256, 128, 32, 1, true, 2, false);
    RUN("256x128x32 2-stage 32x32x16 wgpc1", 256, 128, 32, 2, false, 1, false);
    RUN("256x128x32 2-stage 16x16x32 wgpc1", 256, 128, 32, 2, true, 1, false);
    RUN("what-if no residuals: 128x128x32 1-st wgpc3", 128, 128, 32, 1, false, 3, true);
    RUN("what-if no residuals: 128x128x16 2-st wgpc3", 128, 128, 16, 2, false, 3, true);
    RUN("what-if no residuals: 256x128x32 1-st wgpc2", 256, 128, 32, 1, false, 2, true);
    }
    hipFree(A); hipFree(W); hipFree(Y); hipFree(Wp);
  }
  return 0;
}
