// Two questions about the conv / Linear implicit-GEMM main loop (csrc/conv_mfma.hip), answered on a plain GEMM  C[M][N] = A[M][K] . B[N][K]^T
// (the 1x1-convolution forward form: both operands k-contiguous, ROWK LDS images, ds_read_b128 fragments, v_mfma_f32_32x32x2_f32):
//
//  1. Which main-loop structure is fastest by wall clock on random data?
//       S1  one LDS stage, two barriers per k-tile, register prefetch, 3 workgroups / CU   (the shipped structure)
//       S2  two LDS stages, ONE barrier per k-tile, register prefetch                       (2 workgroups / CU at BK 32, 3 at BK 16)
//  2. How much of an HBM-bound streaming kernel (BatchNorm-apply like: y = x * a + b, 16 B per lane) can run BESIDE the GEMM on a second
//     stream, as a function of the registers each kernel takes (whether the streaming waves fit next to the resident GEMM workgroups)?
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe/gemm_probe.hip -o tools/probe/bin/gemm_probe && tools/probe/bin/gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int OOB_OFF = (int)0x80000000u;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ rsrc_t make_rsrc(const float* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bload4(rsrc_t rs, int voff, int soff) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

struct Frags { float a[2][4], b[2][4]; };
template <int LDT>
__device__ __forceinline__ void load_frags(Frags& f, const float* As, const float* Bs, int wr0, int wc0, int lane, int ks) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(&As[(wr0 + t * 32 + l31) * LDT + ks * 8 + 4 * h]);
    f.a[t][0] = v[0]; f.a[t][1] = v[1]; f.a[t][2] = v[2]; f.a[t][3] = v[3];
    const f32x4 w = *reinterpret_cast<const f32x4*>(&Bs[(wc0 + t * 32 + l31) * LDT + ks * 8 + 4 * h]);
    f.b[t][0] = w[0]; f.b[t][1] = w[1]; f.b[t][2] = w[2]; f.b[t][3] = w[3];
  }
}
__device__ __forceinline__ void mma_frags(const Frags& f, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][t], f.b[j][t], acc[i][j], 0, 0, 0);
}

// 128 x 128 output tile, 4 waves (2 x 2), each 64 x 64 = 2 x 2 accumulators of 32 x 32
// PAD: extra VGPRs kept live across the main loop, to give the probe the register footprint of the real convolution kernels (152-168)
template <int STAGES, int BK, int WGPC, int PAD = 0>
__global__ void __launch_bounds__(256, WGPC) gemm_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  constexpr int LDT = BK + 4, STAGE = 256 * LDT;
  constexpr int CH = BK / 4, RPP = 256 / CH, AP = 128 / RPP;     // float4 per row, rows per pass, passes per operand
  __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
  const int NT = N / 128;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * 128, n0 = nt * 128;
  const int chunk = (tid % CH) * 4, rsub = tid / CH;
  const rsrc_t ra_ = make_rsrc(A, (unsigned)M * K * 4u), rb_ = make_rsrc(B, (unsigned)N * K * 4u);
  int aoff[AP], boff[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    aoff[i] = ((m0 + rsub + RPP * i) * K + chunk) * 4;
    boff[i] = ((n0 + rsub + RPP * i) * K + chunk) * 4;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float pad[PAD > 0 ? PAD : 1];
#pragma unroll
  for (int i = 0; i < PAD; ++i) pad[i] = A[tid + i];
  f32x4 ra[AP], rb[AP];
  int k0 = 0;
  auto load_tile = [&]() {
    const int so = k0 * 4;
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = bload4(ra_, aoff[i], so);
#pragma unroll
    for (int i = 0; i < AP; ++i) rb[i] = bload4(rb_, boff[i], so);
    k0 += BK;           // tiles past K read out of range of neither operand's last rows... (harmless: never stored)
  };
  auto store_tile = [&](float* st) {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&st[(rsub + RPP * i) * LDT + chunk]) = ra[i];
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&st[(128 + rsub + RPP * i) * LDT + chunk]) = rb[i];
  };
  const int nkt = K / BK;
  constexpr int NS = BK / 8;
  Frags fr;
  if constexpr (STAGES == 1) {
    load_tile(); store_tile(smem); __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
      for (int i = 0; i < PAD; ++i) asm volatile("" : "+v"(pad[i]));
      __builtin_amdgcn_sched_barrier(0);
      load_frags<LDT>(fr, smem, smem + 128 * LDT, wr0, wc0, lane, 0);
      load_tile();
      mma_frags(fr, acc);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 2 * AP; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x8, 16 - 2 * AP, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 1; ks < NS; ++ks) { load_frags<LDT>(fr, smem, smem + 128 * LDT, wr0, wc0, lane, ks); mma_frags(fr, acc); }
      __syncthreads();
      if (kt + 1 < nkt) { store_tile(smem); __syncthreads(); }
    }
  } else {
    // two stages: tile kt is read from stage kt&1 while tile kt+1 goes registers -> stage (kt+1)&1 and tile kt+2 global -> registers
    load_tile(); store_tile(smem); load_tile(); __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const float* cur = smem + (kt & 1) * STAGE;
      float* nxt = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
      for (int i = 0; i < PAD; ++i) asm volatile("" : "+v"(pad[i]));
      __builtin_amdgcn_sched_barrier(0);
      load_frags<LDT>(fr, cur, cur + 128 * LDT, wr0, wc0, lane, 0);
      store_tile(nxt);                       // registers of tile kt+1 -> the other stage (its readers passed the last barrier)
      mma_frags(fr, acc);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 2 * AP; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x8, 16 - 2 * AP, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags<LDT>(fr, cur, cur + 128 * LDT, wr0, wc0, lane, 1);
      load_tile();                           // tile kt+2 -> registers
      mma_frags(fr, acc);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 2 * AP; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x8, 16 - 2 * AP, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 2; ks < NS; ++ks) { load_frags<LDT>(fr, cur, cur + 128 * LDT, wr0, wc0, lane, ks); mma_frags(fr, acc); }
      __syncthreads();
    }
  }
  if constexpr (PAD > 0) {
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < PAD; ++i) ps += pad[i];
    if (ps == 123.456f) acc[0][0][0] += ps;
  }
  // epilogue: plain 4-byte stores (acc register j of lane l = row (j&3)+8*(j>>2)+4*(l>>5), column l&31 of its 32x32 tile)
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wc0 + j * 32 + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// ---- LDS-DMA staging (round 4): the same 128 x 128 tile and fragment reads, but the operands go global -> LDS directly
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write issue slots).  One wave-instruction writes 1 KiB of LDS contiguously (wave-uniform
// base + lane * 16 B) = 8 rows of a [row][32 k] image, so the image cannot be padded; bank conflicts of the ds_read_b128 fragment reads are
// avoided by an XOR swizzle of the 16-byte chunks of a row, applied on the SOURCE address: slot (row, s) holds chunk s ^ ((row >> 1) & 7).
//   G1  one LDS stage (32 KB), 3 workgroups / CU: load -> wait -> barrier -> MFMAs -> barrier (no overlap inside a workgroup)
//   G2  two LDS stages (64 KB), 2 workgroups / CU: tile kt+1 lands while tile kt is multiplied, one barrier per k-tile
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
template <int STAGES, int WGPC>
__global__ void __launch_bounds__(256, WGPC) gemm_glds_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  constexpr int BK = 32, STAGE = 256 * BK;
  __shared__ __attribute__((aligned(1024))) float smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
  const int NT = N / 128;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * 128, n0 = nt * 128;
  // DMA pass p (0..7) fills stage rows p*32 .. p*32+31 (0..127 = A tile, 128..255 = B tile); wave w takes rows p*32 + 8w .. +7, lane j row j>>3, slot j&7
  const float* src[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = p * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    src[p] = p < 4 ? A + (size_t)(m0 + row) * K + c * 4 : B + (size_t)(n0 + row - 128) * K + c * 4;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto issue = [&](int kt, float* st) {
    const int k0 = kt * BK;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(src[p] + k0), (lds_void_t*)(st + (p * 32 + wave * 8) * BK), 16, 0, 0);
  };
  auto frags = [&](Frags& f, const float* st, int ks) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ra = wr0 + t * 32 + l31, rb = 128 + wc0 + t * 32 + l31;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&st[ra * BK + 4 * ((ks * 2 + h) ^ ((ra >> 1) & 7))]);
      f.a[t][0] = v[0]; f.a[t][1] = v[1]; f.a[t][2] = v[2]; f.a[t][3] = v[3];
      const f32x4 w = *reinterpret_cast<const f32x4*>(&st[rb * BK + 4 * ((ks * 2 + h) ^ ((rb >> 1) & 7))]);
      f.b[t][0] = w[0]; f.b[t][1] = w[1]; f.b[t][2] = w[2]; f.b[t][3] = w[3];
    }
  };
  const int nkt = K / BK;
  Frags fr;
  if constexpr (STAGES == 1) {
    for (int kt = 0; kt < nkt; ++kt) {
      issue(kt, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { frags(fr, smem, ks); mma_frags(fr, acc); }
      __syncthreads();
    }
  } else {
    issue(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const float* cur = smem + (kt & 1) * STAGE;
      float* nxt = smem + ((kt + 1) & 1) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      frags(fr, cur, 0);
      issue(kt + 1 < nkt ? kt + 1 : kt, nxt);        // the last iteration re-loads a valid tile into the spare stage (never read)
      mma_frags(fr, acc);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) { frags(fr, cur, ks); mma_frags(fr, acc); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wc0 + j * 32 + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// HBM-bound streaming kernel: y = x * a[c] + b[c] over an [R][C] matrix, 16 B per lane, UNROLL rows in flight per lane
template <int UNROLL>
__global__ void __launch_bounds__(256) stream_k(long long n4, const f32x4* __restrict__ x, f32x4* __restrict__ y, float a, float b) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) y[i + u * stride] = v[u] * a + b;
  }
  for (; i < n4; i += stride) y[i] = x[i] * a + b;
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  const int M = 98304, N = 256, K = 1024;              // 768 x 2 tiles of 128 x 128: whole rounds at 2 and at 3 workgroups per CU
  const double gflop = 2.0 * M * N * K / 1e9;
  float *A, *B, *C, *X, *Y;
  const long long n4 = 401408LL * 1024 / 4;            // 1.64 GB in, 1.64 GB out per pass
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&X, n4 * 16)); CK(hipMalloc(&Y, n4 * 16));
  {
    std::vector<float> h((size_t)M * K);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 777, (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Y, 0, n4 * 16));
  }
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  const int grid = (M / 128) * (N / 128);
  const int REP = 40, SREP = 12, NV = 11;
  const bool gemm_only = getenv("GEMM_ONLY") != nullptr;      // round 4: only the main-loop table (the co-residency part is round 2's)

  auto run_gemm = [&](int v, hipStream_t s) {
    switch (v) {
      case 0: hipLaunchKernelGGL((gemm_k<1, 32, 3>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 1: hipLaunchKernelGGL((gemm_k<2, 32, 2>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 2: hipLaunchKernelGGL((gemm_k<2, 16, 3>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 3: hipLaunchKernelGGL((gemm_k<1, 32, 2>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 4: hipLaunchKernelGGL((gemm_k<1, 16, 3>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 5: hipLaunchKernelGGL((gemm_k<1, 32, 3, 50>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 6: hipLaunchKernelGGL((gemm_k<1, 32, 3, 34>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 7: hipLaunchKernelGGL((gemm_k<2, 32, 2, 48>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 8: hipLaunchKernelGGL((gemm_glds_k<1, 3>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 9: hipLaunchKernelGGL((gemm_glds_k<2, 2>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
      case 10: hipLaunchKernelGGL((gemm_glds_k<1, 4>), dim3(grid), dim3(256), 0, s, M, N, K, A, B, C); break;
    }
  };
  const char* gname[] = {"S1 bk32 3wg/cu (shipped)", "S2 bk32 2wg/cu", "S2 bk16 3wg/cu", "S1 bk32 2wg/cu", "S1 bk16 3wg/cu", "S1 bk32 3wg/cu ~160 regs", "S1 bk32 3wg/cu ~144 regs", "S2 bk32 2wg/cu ~192 regs", "G1 LDS-DMA 1 stage 3wg/cu", "G2 LDS-DMA 2 stages 2wg/cu", "G1 LDS-DMA 1 stage 4wg/cu"};
  auto run_stream = [&](int u, int wgs, hipStream_t s) {
    switch (u) {
      case 1: hipLaunchKernelGGL(stream_k<1>, dim3(wgs), dim3(256), 0, s, n4, (const f32x4*)X, (f32x4*)Y, 1.0001f, 0.5f); break;
      case 2: hipLaunchKernelGGL(stream_k<2>, dim3(wgs), dim3(256), 0, s, n4, (const f32x4*)X, (f32x4*)Y, 1.0001f, 0.5f); break;
      case 4: hipLaunchKernelGGL(stream_k<4>, dim3(wgs), dim3(256), 0, s, n4, (const f32x4*)X, (f32x4*)Y, 1.0001f, 0.5f); break;
      case 8: hipLaunchKernelGGL(stream_k<8>, dim3(wgs), dim3(256), 0, s, n4, (const f32x4*)X, (f32x4*)Y, 1.0001f, 0.5f); break;
    }
  };

  // correctness spot check of every GEMM variant against variant 0 (same k order => bitwise equal)
  std::vector<float> ref(4096), got(4096);
  for (int v = 0; v < NV; ++v) {
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    run_gemm(v, s1); CK(hipStreamSynchronize(s1));
    CK(hipMemcpy(v == 0 ? ref.data() : got.data(), C + (size_t)12345 * N, 4096 * 4, hipMemcpyDeviceToHost));
    if (v > 0) { int bad = 0; for (int i = 0; i < 4096; ++i) bad += ref[i] != got[i]; printf("variant %d vs 0: %d mismatches of 4096\n", v, bad); }
  }

  float tg[NV];
  for (int v = 0; v < NV; ++v) {
    for (int r = 0; r < 200; ++r) run_gemm(v, s1);          // warm the clocks (~0.1 s)
    CK(hipEventRecord(e0, s1));
    for (int r = 0; r < REP; ++r) run_gemm(v, s1);
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    tg[v] = elapsed(e0, e1) / REP;
    printf("GEMM %-26s alone: %.3f ms  %.1f TFLOP/s\n", gname[v], tg[v], gflop / tg[v]);
  }
  if (gemm_only) return 0;
  const double gb = 2.0 * n4 * 16 / 1e9;
  float ts[9][4097 / 256 + 1] = {};
  for (int u : {1, 2, 4, 8})
    for (int wgs : {256, 512, 1024, 4096}) {
      for (int r = 0; r < 3; ++r) run_stream(u, wgs, s2);
      CK(hipEventRecord(f0, s2));
      for (int r = 0; r < SREP; ++r) run_stream(u, wgs, s2);
      CK(hipEventRecord(f1, s2)); CK(hipEventSynchronize(f1));
      const float t = elapsed(f0, f1) / SREP;
      ts[u][wgs / 256] = t;
      printf("stream unroll %d, %4d workgroups alone: %.3f ms  %.2f TB/s\n", u, wgs, t, gb / t / 1e3);
    }
  // together: REP GEMMs on s1 next to SREP streaming passes on s2, started together; wall = until both are done
  for (int v : {0, 5, 6, 1, 7}) {
    for (int u : {1, 2, 4, 8}) {
      for (int wgs : {256, 1024}) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
        CK(hipEventRecord(f0, s2));
        for (int r = 0; r < REP; ++r) run_gemm(v, s1);
        for (int r = 0; r < SREP; ++r) run_stream(u, wgs, s2);
        CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2));
        CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
        const float t1 = elapsed(e0, e1), t2 = elapsed(f0, f1);
        printf("together: GEMM %-26s %.1f ms (alone %.1f) | stream u%d %4d wgs %.1f ms (alone %.1f) | wall %.1f vs sum-alone %.1f\n", gname[v], t1, tg[v] * REP, u, wgs, t2,
               ts[u][wgs / 256] * SREP, t1 > t2 ? t1 : t2, tg[v] * REP + ts[u][wgs / 256] * SREP);
      }
    }
  }
  return 0;
}
