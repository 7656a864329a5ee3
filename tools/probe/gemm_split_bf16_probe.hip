// Probe (round 5, for the next round's decision): an fp32 GEMM on the BF16 matrix pipe by operand splitting.
//   a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)  (3 x 8 significant bits = the 24 of an fp32), the same for b;
//   a * b = sum of ai * bj, every ai * bj EXACT in the fp32 accumulator's input (8 x 8 bits); TERMS = 6 keeps the products down to 2^-16 relative
//   (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1), TERMS = 9 all of them, TERMS = 3 down to 2^-8 (a1b1, a1b2, a2b1), TERMS = 1 is plain bf16.
// v_mfma_f32_32x32x16_bf16 runs at 16x the per-clock rate of v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md, matrix cores), so 6 / 9 terms are worth 2.67x / 1.78x the
// fp32 pipe per clock - before the clock the chip holds under bf16 load, the split arithmetic (VALU) and 1.5x the LDS bytes per k.
// What it prints, per term count and for the fp32-MFMA kernel of the same tiling: ms, TFLOP/s of the fp32 GEMM's 2MNK, relative l2 error against fp64 on sampled rows.
//   C[M][N] = sum_k A[M][K] * B[N][K]   (both K-contiguous, fp32 in HBM, standard-normal entries)
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/probe/gemm_split_bf16_probe.hip -o tools/probe/bin/gemm_split_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));

#ifndef WHATIF
#define WHATIF 0     // timing what-ifs (results wrong): 1 no residual arithmetic (all planes = the first conversion), 2 = 1 + only plane 0 written to LDS, 4 no restaging / barriers after the first tile
#endif
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDH = 40;                 // bf16 elements per LDS row (32 + 8 pad: 80 bytes, odd multiple of 16 -> conflict-free ds_read_b128 over 16 rows)
constexpr int LDF = 36;                 // floats per LDS row of the fp32 kernel

__device__ __forceinline__ unsigned short bf16_rne(float x) {          // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// ---------------------------------------------------------------------------------------------------------------- split kernel
template <int TERMS>
__global__ void __launch_bounds__(256, 2) gemm_split_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  constexpr int NP = TERMS == 1 ? 1 : (TERMS == 3 ? 2 : 3);           // planes kept per operand
  __shared__ __attribute__((aligned(16))) unsigned short sa[NP][BM * LDH], sb[NP][BN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // staging: a 128 x 32 fp32 tile = 1024 float4; thread t takes float4 t, t + 256, t + 512, t + 768: row = idx >> 3, k4 = (idx & 7) * 4
  f32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
      ra[i] = *(const f32x4*)(A + (size_t)(m0 + row) * K + k0 + k4);
      rb[i] = *(const f32x4*)(B + (size_t)(n0 + row) * K + k0 + k4);
    }
  };
  auto split_store = [&](const f32x4& v, unsigned short (*s)[BM * LDH], int off) {
    // hardware conversions (v_cvt_pk_bf16_f32: round to nearest even, two values per instruction); the residuals are exact fp32 subtractions
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
    unsigned pl[3][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const bf2 h0 = __builtin_convertvector(x0, bf2), h1 = __builtin_convertvector(x1, bf2);
      pl[q][0] = __builtin_bit_cast(unsigned, h0); pl[q][1] = __builtin_bit_cast(unsigned, h1);
      if (q + 1 < NP && !(WHATIF & 3)) {
        x0 -= f2{__uint_as_float(pl[q][0] << 16), __uint_as_float(pl[q][0] & 0xFFFF0000u)};
        x1 -= f2{__uint_as_float(pl[q][1] << 16), __uint_as_float(pl[q][1] & 0xFFFF0000u)};
      }
      if (!(WHATIF & 2) || q == 0) *(uint2*)(&s[q][off]) = make_uint2(pl[q][0], pl[q][1]);
      if ((WHATIF & 3) && q == 0 && NP > 1) { if (WHATIF & 2) break; }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
      split_store(ra[i], sa, row * LDH + k4);
      split_store(rb[i], sb, row * LDH + k4);
    }
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more && !(WHATIF & 4)) gload(k0 + BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                     // two k-slabs of 16
      bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[q][t] = *(const bf16x8*)(&sa[q][(wm + 32 * t + r) * LDH + 16 * s + 8 * h]);
          fb[q][t] = *(const bf16x8*)(&sb[q][(wn + 32 * t + r) * LDH + 16 * s + 8 * h]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
#define MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, fa[P][i]), __builtin_bit_cast(bf16x8n, fb[Q][j]), acc[i][j], 0, 0, 0)
          if (TERMS == 9) { MM(2, 2); MM(2, 1); MM(1, 2); }
          if (TERMS >= 6) { MM(2, 0); MM(0, 2); MM(1, 1); }
          if (TERMS >= 3) { MM(1, 0); MM(0, 1); }
          MM(0, 0);
#undef MM
        }
    }
    if (WHATIF & 4) continue;
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn + 32 * j + r;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------------------------- B (the weights) split ONCE, ahead of the GEMM
__global__ void presplit_k(size_t n, const float* __restrict__ x, unsigned short* __restrict__ p0, unsigned short* __restrict__ p1, unsigned short* __restrict__ p2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const unsigned short a = bf16_rne(v); const float r1 = v - bf16_to_f32(a);
  const unsigned short b = bf16_rne(r1); const float r2 = r1 - bf16_to_f32(b);
  p0[i] = a; p1[i] = b; p2[i] = bf16_rne(r2);
}

// the two-workgroups-per-CU kernel with B read as three bf16 planes [3][N][K] (16-byte pieces straight to LDS): half of the split's vector work is gone
template <int TERMS>
__global__ void __launch_bounds__(256, 2) gemm_split_preb_k(int M, int N, int K, const float* __restrict__ A, const unsigned short* __restrict__ Bp, float* __restrict__ C) {
  constexpr int NP = 3;
  __shared__ __attribute__((aligned(16))) unsigned short sa[NP][BM * LDH], sb[NP][BN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 ra[4];
  bf16x8 rb[NP][2];                                                   // per plane: 128 x 32 bf16 = 512 pieces of 8; thread t takes pieces t, t + 256: row = idx >> 2, k8 = (idx & 3) * 8
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
      ra[i] = *(const f32x4*)(A + (size_t)(m0 + row) * K + k0 + k4);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, k8 = (idx & 3) * 8;
        rb[q][i] = *(const bf16x8*)(Bp + ((size_t)q * N + n0 + row) * K + k0 + k8);
      }
  };
  auto split_store = [&](const f32x4& v, int off) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf2)), p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf2));
      if (q + 1 < NP) {
        x0 -= f2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
        x1 -= f2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
      }
      *(uint2*)(&sa[q][off]) = make_uint2(p0, p1);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int idx = tid + 256 * i; split_store(ra[i], (idx >> 3) * LDH + (idx & 7) * 4); }
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int idx = tid + 256 * i; *(bf16x8*)(&sb[q][(idx >> 2) * LDH + (idx & 3) * 8]) = rb[q][i]; }
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) gload(k0 + BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[q][t] = *(const bf16x8*)(&sa[q][(wm + 32 * t + r) * LDH + 16 * s + 8 * h]);
          fb[q][t] = *(const bf16x8*)(&sb[q][(wn + 32 * t + r) * LDH + 16 * s + 8 * h]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#define MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, fa[P][i]), __builtin_bit_cast(bf16x8n, fb[Q][j]), acc[i][j], 0, 0, 0)
          if (TERMS == 9) { MM(2, 2); MM(2, 1); MM(1, 2); }
          MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0);
#undef MM
        }
    }
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn + 32 * j + r;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------------------------- pipelined split kernel
// ONE workgroup per CU (one wave per SIMD, up to 512 registers), LDS double-buffered (2 x 61 KB), ONE barrier per k-tile.  Two register stages: while the MFMAs of tile k
// run from LDS buffer k & 1, the rows of tile k + 1 (loaded a whole iteration earlier) are split and written to buffer (k + 1) & 1 in eight pieces placed between the MFMA
// groups, and the loads of tile k + 2 are in flight.
template <int TERMS>
__global__ void __launch_bounds__(256, 1) gemm_split_pipe_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  constexpr int NP = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];          // [2 buffers][A | B][3 planes][128 * LDH]
  auto plane = [&](int buf, int ab, int q) { return lds + (size_t)((buf * 2 + ab) * NP + q) * (BM * LDH); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* pa[4];
  const float* pb[4];
  int off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
    pa[i] = A + (size_t)(m0 + row) * K + k4;
    pb[i] = B + (size_t)(n0 + row) * K + k4;
    off[i] = row * LDH + k4;
  }
  f32x4 st[2][8];                                                      // two stages x (4 A + 4 B) float4
  auto gload = [&](f32x4* s, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] = *(const f32x4*)(pa[i] + k0); s[4 + i] = *(const f32x4*)(pb[i] + k0); }
  };
  auto split_store = [&](const f32x4& v, int buf, int ab, int o) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf2)), p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf2));
      if (q + 1 < NP) {
        x0 -= f2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
        x1 -= f2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
      }
      *(uint2*)(plane(buf, ab, q) + o) = make_uint2(p0, p1);
    }
  };
  // one k-tile: MFMAs from buffer `buf`, the stage `nx` (tile k + 1) split into buffer buf ^ 1 in 8 pieces between the MFMA groups
  auto tile = [&](int buf, const f32x4* nx, auto next_tag) {
    constexpr bool has_next = decltype(next_tag)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[q][t] = *(const bf16x8*)(plane(buf, 0, q) + (wm + 32 * t + r) * LDH + 16 * s + 8 * h);
          fb[q][t] = *(const bf16x8*)(plane(buf, 1, q) + (wn + 32 * t + r) * LDH + 16 * s + 8 * h);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#define MM(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, fa[P][i]), __builtin_bit_cast(bf16x8n, fb[Q][j]), acc[i][j], 0, 0, 0)
          if (TERMS == 9) { MM(2, 2); MM(2, 1); MM(1, 2); }
          MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0);
#undef MM
          const int piece = (s * 2 + i) * 2 + j;                       // 0 .. 7: one staged float4 per MFMA group
          if (has_next) {
            split_store(nx[piece], buf ^ 1, piece >> 2, off[piece & 3]);
#ifndef NO_SCHED_GROUPS
            // issue order asked of the scheduler: one MFMA, four vector instructions, ... - the split's ~24 instructions ride in the MFMAs' shadows (one wave per SIMD:
            // nothing else can fill them), then the three LDS writes
#pragma unroll
            for (int g = 0; g < (TERMS == 9 ? 9 : 6); ++g) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 9 ? 3 : 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
#endif
          }
        }
    }
  };
  const int nk = K / BK;
  gload(st[0], 0);
  if (nk > 1) gload(st[1], BK);
#pragma unroll
  for (int p = 0; p < 8; ++p) split_store(st[0][p], 0, p >> 2, off[p & 3]);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) gload(st[0], (kt + 2) * BK);                      // stage 0 is free: its tile is in LDS buffer 0
    if (kt + 1 < nk) tile(0, st[1], std::true_type{}); else tile(0, st[1], std::false_type{});
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) gload(st[1], (kt + 3) * BK);
    if (kt + 2 < nk) tile(1, st[0], std::true_type{}); else tile(1, st[0], std::false_type{});
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn + 32 * j + r;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------------------------- fp32 MFMA kernel, same tiling
__global__ void __launch_bounds__(256, 2) gemm_f32_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) float sa[BM * LDF], sb[BN * LDF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
      ra[i] = *(const f32x4*)(A + (size_t)(m0 + row) * K + k0 + k4);
      rb[i] = *(const f32x4*)(B + (size_t)(n0 + row) * K + k0 + k4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = (idx & 7) * 4;
      *(f32x4*)(&sa[row * LDF + k4]) = ra[i];
      *(f32x4*)(&sb[row * LDF + k4]) = rb[i];
    }
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) gload(k0 + BK);
    float fa[2][16], fb[2][16];                                       // contraction order k = 16 h + s: 16 consecutive floats per lane and tile
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 va = *(const f32x4*)(&sa[(wm + 32 * t + r) * LDF + 16 * h + 4 * q]);
        const f32x4 vb = *(const f32x4*)(&sb[(wn + 32 * t + r) * LDF + 16 * h + 4 * q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { fa[t][4 * q + e] = va[e]; fb[t][4 * q + e] = vb[e]; }
      }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn + 32 * j + r;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------------------------- host
static double rel_l2(const std::vector<float>& c, const std::vector<double>& ref, int rows, int N, const std::vector<int>& which) {
  double num = 0, den = 0;
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < N; ++j) {
      const double d = (double)c[(size_t)which[i] * N + j] - ref[(size_t)i * N + j];
      num += d * d; den += ref[(size_t)i * N + j] * ref[(size_t)i * N + j];
    }
  return sqrt(num / den);
}

template <typename F> static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int nonneg = argc > 4 ? atoi(argv[4]) : 0;                    // 1: A is non-negative (a ReLU output, as the conv layers' activations are)
  if (M % BM || N % BN || K % BK) { printf("M, N multiples of 128, K of 32\n"); return 1; }
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  auto normal = [&]() { const double u = rnd() + 1e-300, v = rnd(); return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v)); };
  for (auto& x : ha) { x = normal(); if (nonneg) x = x > 0 ? x : 0.f; }
  for (auto& x : hb) x = normal() * 0.05f;
  float *A, *B, *C;
  hipMalloc(&A, ha.size() * 4); hipMalloc(&B, hb.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  // fp64 reference on 16 sampled rows
  const int rows = 16;
  std::vector<int> which(rows);
  for (int i = 0; i < rows; ++i) which[i] = (int)((size_t)i * 2654435761u % M);
  std::vector<double> ref((size_t)rows * N);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0;
      const float* a = &ha[(size_t)which[i] * K]; const float* b = &hb[(size_t)j * K];
      for (int k = 0; k < K; ++k) s += (double)a[k] * (double)b[k];
      ref[(size_t)i * N + j] = s;
    }
  std::vector<float> hc((size_t)M * N);
  const dim3 grid(N / BN, M / BM), block(256);
  const double flop = 2.0 * M * N * K;
  printf("C[%d][%d] = A[%d][%d] B[%d][%d]^T, fp32 operands in HBM (A %s), tile 128 x 128 x 32, 4 waves, 2 workgroups per CU, error = relative l2 against fp64 on %d rows\n",
         M, N, M, K, N, K, nonneg ? "non-negative" : "normal", rows);
  auto report = [&](const char* name, float ms) {
    hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
    printf("  %-34s %8.3f ms  %7.1f TFLOP/s (of 2MNK)   error %.3e\n", name, ms, flop / ms / 1e9, rel_l2(hc, ref, rows, N, which));
  };
  report("fp32 MFMA (32x32x2 f32)", time_ms([&]() { hipLaunchKernelGGL(gemm_f32_k, grid, block, 0, 0, M, N, K, A, B, C); }, 10));
  report("bf16 split, 9 terms", time_ms([&]() { hipLaunchKernelGGL(gemm_split_k<9>, grid, block, 0, 0, M, N, K, A, B, C); }, 10));
  report("bf16 split, 6 terms", time_ms([&]() { hipLaunchKernelGGL(gemm_split_k<6>, grid, block, 0, 0, M, N, K, A, B, C); }, 10));
  {
    const size_t lds_bytes = (size_t)2 * 2 * 3 * BM * LDH * 2;
    hipFuncSetAttribute((const void*)gemm_split_pipe_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipFuncSetAttribute((const void*)gemm_split_pipe_k<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    report("bf16 split, 6 terms, pipelined", time_ms([&]() { hipLaunchKernelGGL(gemm_split_pipe_k<6>, grid, block, lds_bytes, 0, M, N, K, A, B, C); }, 10));
    report("bf16 split, 9 terms, pipelined", time_ms([&]() { hipLaunchKernelGGL(gemm_split_pipe_k<9>, grid, block, lds_bytes, 0, M, N, K, A, B, C); }, 10));
  }
  {
    unsigned short* Bp; hipMalloc(&Bp, (size_t)3 * N * K * 2);
    const size_t nb = (size_t)N * K;
    const float tp = time_ms([&]() { hipLaunchKernelGGL(presplit_k, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, nb, B, Bp, Bp + nb, Bp + 2 * nb); }, 5);
    printf("  (splitting B once: %.3f ms)\n", tp);
    report("bf16 split, 6 terms, B pre-split", time_ms([&]() { hipLaunchKernelGGL(gemm_split_preb_k<6>, grid, block, 0, 0, M, N, K, A, Bp, C); }, 10));
    report("bf16 split, 9 terms, B pre-split", time_ms([&]() { hipLaunchKernelGGL(gemm_split_preb_k<9>, grid, block, 0, 0, M, N, K, A, Bp, C); }, 10));
    hipFree(Bp);
  }
  report("bf16 split, 3 terms", time_ms([&]() { hipLaunchKernelGGL(gemm_split_k<3>, grid, block, 0, 0, M, N, K, A, B, C); }, 10));
  report("plain bf16 (1 term)", time_ms([&]() { hipLaunchKernelGGL(gemm_split_k<1>, grid, block, 0, 0, M, N, K, A, B, C); }, 10));
  return 0;
}
