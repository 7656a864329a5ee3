// fp32 batched GEMM on the BF16 matrix pipe by operand splitting (round 5; OPT-IN: ops.SPLIT_BF16_TERMS / SSV_SPLIT_BF16, default off - the shipped step runs every
// product on v_mfma_f32_32x32x2_f32).  For the transformed-domain products of the Winograd layers (csrc/winograd*.hip): y[b][rows][K] = a[b][rows][C] . w[b][K][C]^T (+ bias[K] + addend[b][rows][K]: the plain 1x1 / Linear products' epilogue).
//
//   a = a1 + a2 + a3,  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)   (round to nearest even; the residuals are exact fp32 subtractions; 3 x 8 significant
//   bits cover the 24 of an fp32), the same for w;  a . w = sum of ai . wj with every ai . wj exact (8 x 8 bits) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//   TERMS = 6 keeps a1w1, a1w2, a2w1, a2w2, a1w3, a3w1 (what is dropped is <= 2^-23 of the product: the size of ONE fp32 rounding of it); TERMS = 9 keeps all.
// The bf16 MFMA sums 16 products per instruction before it rounds into the accumulator (the fp32 form: 2), so the accumulation error is no larger: measured against fp64,
// 6 and 9 terms give the SAME error, at or below the fp32 MFMA kernel's (tools/probe/gemm_split_bf16_probe.hip: 1.01e-6 vs 1.15e-6 at K = 4096; tests/test_gpu_winograd44.py).
// Rate: the bf16 pipe sustains 1.7-1.9 PFLOP/s on random operands (tools/probe/mfma_bf16_rate_probe.hip), i.e. 280-315 TFLOP/s of fp32 products at 6 terms against
// 135-148 for v_mfma_f32_32x32x2_f32 in the same loop.
//
// Tiling: workgroup 128 rows x 128 channels x 32 of the contraction, four waves (2 x 2) of 64 x 64.  Both operands are split while they go from registers to LDS (three
// bf16 planes each, row stride 80 bytes: conflict-free ds_read_b128).  The MFMA takes the w fragment as its A operand and the a fragment as B, so a lane ends up with FOUR
// CONSECUTIVE channels of one output row per register quad: 16-byte stores without a transpose.
#include "common.h"

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int SLD = 40;                                        // bf16 elements per LDS row: 32 + 8 of padding

template <int TERMS>
__global__ void __launch_bounds__(256, 2)
gemm_batched_split_k(int rows, int C, int K, int tiles_n, const float* __restrict__ a, const float* __restrict__ w, float* __restrict__ y,
                     const float* __restrict__ bias, const float* __restrict__ addend) {
  static_assert(TERMS == 6 || TERMS == 9, "6 or 9 terms");
  __shared__ __attribute__((aligned(16))) unsigned short sa[3][SBM * SLD], sb[3][SBN * SLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = (blockIdx.x / tiles_n) * SBM, n0 = (blockIdx.x % tiles_n) * SBN;
  a += (size_t)blockIdx.y * rows * C;
  w += (size_t)blockIdx.y * K * C;
  y += (size_t)blockIdx.y * rows * K;
  if (addend) addend += (size_t)blockIdx.y * rows * K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // staging: a 128 x 32 fp32 tile = 1,024 float4; thread t takes float4 t + 256 i: row = idx >> 3, k4 = (idx & 7) * 4.  Rows past the end repeat the last row (their
  // results are never stored).
  const float* pa[4];
  const float* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
    pa[i] = a + (size_t)min(m0 + row, rows - 1) * C + k4;
    pb[i] = w + (size_t)min(n0 + row, K - 1) * C + k4;
  }
  f32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = *(const f32x4*)(pa[i] + k0); rb[i] = *(const f32x4*)(pb[i] + k0); }
  };
  auto split_store = [&](const f32x4& v, unsigned short (*s)[SBM * SLD], int off) {
    f32x2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf16x2)), p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf16x2));
      if (q < 2) {
        x0 -= f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
        x1 -= f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
      }
      *(uint2*)(&s[q][off]) = make_uint2(p0, p1);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, off = (idx >> 3) * SLD + (idx & 7) * 4;
      split_store(ra[i], sa, off);
      split_store(rb[i], sb, off);
    }
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int k0 = 0; k0 < C; k0 += SBK) {
    const bool more = k0 + SBK < C;
    if (more) gload(k0 + SBK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                // two slabs of 16 along the contraction
      s16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[q][t] = *(const s16x8*)(&sa[q][(wm + 32 * t + r) * SLD + 16 * s + 8 * h]);
          fb[q][t] = *(const s16x8*)(&sb[q][(wn + 32 * t + r) * SLD + 16 * s + 8 * h]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // D[channel][row] += w-plane P . a-plane Q; smallest terms first
#define SSV_MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[P][j]), __builtin_bit_cast(bf16x8, fa[Q][i]), acc[i][j], 0, 0, 0)
          if (TERMS == 9) { SSV_MM(2, 2); SSV_MM(2, 1); SSV_MM(1, 2); }
          SSV_MM(2, 0); SSV_MM(0, 2); SSV_MM(1, 1);
          SSV_MM(1, 0); SSV_MM(0, 1);
          SSV_MM(0, 0);
#undef SSV_MM
        }
    }
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
  // lane (r, h) of tile (i, j): output row m0 + wm + 32 i + r, channels n0 + wn + 32 j + 8 q + 4 h + {0..3} in registers 4 q .. 4 q + 3
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wm + 32 * i + r;
    if (row >= rows) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn + 32 * j + 8 * q + 4 * h;
        if (col < K) {
          f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          if (bias) v += *(const f32x4*)(bias + col);                   // y = a . w^T + bias[channel] + addend[row][channel], as ssv_conv2d_fwd's epilogue
          if (addend) v += *(const f32x4*)(addend + (size_t)row * K + col);
          *(f32x4*)(y + (size_t)row * K + col) = v;
        }
      }
  }
}

}  // namespace

extern "C" int ssv_gemm_batched_split(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* a, const float* w, float* y,
                                      const float* bias, const float* addend, int32_t terms, void* stream) {
  SSV_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && rows < (1ll << 31) && C > 0 && K > 0, "ssv_gemm_batched_split: bad shape");
  SSV_REQUIRE(C % 32 == 0 && K % 4 == 0, "ssv_gemm_batched_split: needs C %% 32 == 0 and K %% 4 == 0 (got C=%d K=%d)", C, K);
  SSV_REQUIRE(terms == 6 || terms == 9, "ssv_gemm_batched_split: terms must be 6 or 9 (got %d)", terms);
  SSV_REQUIRE(a && w && y && (((uintptr_t)a | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)addend) & 15) == 0, "ssv_gemm_batched_split: null or unaligned pointer");
  SSV_REQUIRE(!bias || batch == 1, "ssv_gemm_batched_split: a bias goes with one product (batch 1)");
  const int64_t tiles_m = cdiv64(rows, SBM);
  const int tiles_n = cdiv(K, SBN);
  SSV_REQUIRE(tiles_m * tiles_n < (1ll << 31), "ssv_gemm_batched_split: grid too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)batch);
  if (terms == 6) hipLaunchKernelGGL(gemm_batched_split_k<6>, grid, dim3(256), 0, s, (int)rows, C, K, tiles_n, a, w, y, bias, addend);
  else hipLaunchKernelGGL(gemm_batched_split_k<9>, grid, dim3(256), 0, s, (int)rows, C, K, tiles_n, a, w, y, bias, addend);
  SSV_CHECK_LAUNCH("ssv_gemm_batched_split");
  return SSV_OK;
}
