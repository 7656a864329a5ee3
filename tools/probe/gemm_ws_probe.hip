// Producer / consumer specialisation of the implicit-GEMM main loop (round 4 probe).
//
// Round 3 (x3) and round 4 (profiles/r04_whatif_pieces.txt) measured that a convolution costs its MFMA time at the fp32 peak PLUS most of everything else (staging
// loads, ds_writes, barriers, epilogue), although three workgroups are resident per CU: inside one wave the phases are serial, and the residents do not interleave well.
// Here the two kinds of work live in DIFFERENT waves of one workgroup:
//   consumers (4 waves, one per SIMD)  : ds_read fragments + v_mfma_f32_32x32x2_f32 only, then the epilogue stores of their 64 x 64 sub-tile;
//   producers (4 waves, one per SIMD)  : buffer loads global -> VGPR -> ds_write into a ring of NS LDS stages, running ahead of the consumers - across tile
//                                        boundaries too: the workgroups are PERSISTENT (one per resident slot, each walks a strided list of tiles), so the next
//                                        tile's operands land while the consumers are still storing the previous tile.
// Hand-over through monotonic LDS counters per stage (full[s]: producer waves that have written use u of stage s; done[s]: consumer waves that have read it), polled
// with s_sleep; every poll loop is BOUNDED and a time-out raises an abort flag that makes every later wait fall through, so the kernel always terminates.
// Same 128 x 128 x 32 tile, LDS image, fragment reads and k order as the shipped loop (S1): results are bitwise equal to it.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe/gemm_ws_probe.hip -o tools/probe/bin/gemm_ws_probe && tools/probe/bin/gemm_ws_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
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

constexpr int BK = 32, LDT = BK + 4, STAGE = 256 * LDT;          // floats per stage: A rows 0..127, B rows 128..255

struct Frags { float a[2][4], b[2][4]; };
__device__ __forceinline__ void load_frags(Frags& f, const float* st, int wr0, int wc0, int lane, int ks) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(&st[(wr0 + t * 32 + l31) * LDT + ks * 8 + 4 * h]);
    f.a[t][0] = v[0]; f.a[t][1] = v[1]; f.a[t][2] = v[2]; f.a[t][3] = v[3];
    const f32x4 w = *reinterpret_cast<const f32x4*>(&st[(128 + wc0 + t * 32 + l31) * LDT + ks * 8 + 4 * h]);
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

// ---- reference: the shipped structure (one LDS stage, register prefetch, two barriers per k-tile, 3 workgroups / CU) --------------------------------------------
__global__ void __launch_bounds__(256, 3) gemm_s1_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) float smem[STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
  const int NT = N / 128;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * 128, n0 = nt * 128;
  const int chunk = (tid & 7) * 4, rsub = tid >> 3;
  const rsrc_t ra_ = make_rsrc(A, (unsigned)M * K * 4u), rb_ = make_rsrc(B, (unsigned)N * K * 4u);
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { aoff[i] = ((m0 + rsub + 32 * i) * K + chunk) * 4; boff[i] = ((n0 + rsub + 32 * i) * K + chunk) * 4; }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 ra[4], rb[4];
  int k0 = 0;
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(ra_, aoff[i], k0 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = bload4(rb_, boff[i], k0 * 4);
    k0 += BK;
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&smem[(rsub + 32 * i) * LDT + chunk]) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&smem[(128 + rsub + 32 * i) * LDT + chunk]) = rb[i];
  };
  const int nkt = K / BK;
  Frags fr;
  load_tile(); store_tile(); __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    __builtin_amdgcn_sched_barrier(0);
    load_frags(fr, smem, wr0, wc0, lane, 0);
    load_tile();
    mma_frags(fr, acc);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) { load_frags(fr, smem, wr0, wc0, lane, ks); mma_frags(fr, acc); }
    __syncthreads();
    if (kt + 1 < nkt) { store_tile(); __syncthreads(); }
  }
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(size_t)(m0 + wr0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + n0 + wc0 + j * 32 + l31] = acc[i][j][r];
}

// ---- producer / consumer specialisation ------------------------------------------------------------------------------------------------------------------------
constexpr unsigned SPIN_LIMIT = 1u << 18;      // polls (each >= ~64 cycles with s_sleep 1): a healthy wait is a few hundred polls at most

__device__ __forceinline__ void wait_ge(volatile unsigned* ctr, unsigned target, volatile unsigned* abort_flag) {
  unsigned n = 0;
  while (true) {
    if (*ctr >= target) return;
    if (*abort_flag) return;
    if (++n > SPIN_LIMIT) { *abort_flag = 1u; return; }
    __builtin_amdgcn_s_sleep(1);
  }
}
#define ORDER() asm volatile("" ::: "memory")      // compiler barrier: no LDS access of the stage moves across the hand-over

template <int NS, int WGPC>
__global__ void __launch_bounds__(512, WGPC) gemm_ws_k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int* __restrict__ err) {
  __shared__ __attribute__((aligned(16))) float smem[NS * STAGE];
  __shared__ unsigned full[NS], done[NS], abort_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NS) { full[tid] = 0u; done[tid] = 0u; }
  if (tid == 0) abort_flag = 0u;
  __syncthreads();
  const int NT = N / 128, total = (M / 128) * NT, nkt = K / BK;
  // persistent schedule: XCD x (blockIdx & 7) owns a contiguous chunk of the tile list (tiles sharing an operand panel meet in one L2), its workgroups stride through it
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nl = gridDim.x >> 3;
  const int per = total >> 3, rem = total & 7;
  const int cbeg = xcd * per + (xcd < rem ? xcd : rem), clen = per + (xcd < rem ? 1 : 0);
  unsigned it = 0;                                   // k-tiles handed over so far (the same sequence on both sides)
  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pt = tid - 256;                        // 0 .. 255: the loader's thread index, as in the shipped loop
    const int chunk = (pt & 7) * 4, rsub = pt >> 3;
    const rsrc_t ra_ = make_rsrc(A, (unsigned)M * K * 4u), rb_ = make_rsrc(B, (unsigned)N * K * 4u);
    for (int j = local; j < clen; j += nl) {
      const int tile = cbeg + j, mt = tile / NT, nt = tile - mt * NT;
      int aoff[4], boff[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { aoff[i] = ((mt * 128 + rsub + 32 * i) * K + chunk) * 4; boff[i] = ((nt * 128 + rsub + 32 * i) * K + chunk) * 4; }
      for (int kt = 0; kt < nkt; ++kt, ++it) {
        f32x4 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = bload4(ra_, aoff[i], kt * BK * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = bload4(rb_, boff[i], kt * BK * 4);
        const unsigned s = it % NS, u = it / NS;
        wait_ge(&done[s], 4u * u, &abort_flag);      // the four consumer waves have read the previous use of this stage
        ORDER();
        float* st = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&st[(rsub + 32 * i) * LDT + chunk]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&st[(128 + rsub + 32 * i) * LDT + chunk]) = rb[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&full[s], 1u);
      }
    }
  } else {
    // ------------------------------------------------------------------ consumers
    const int wr0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
    const int l31 = lane & 31, h = lane >> 5;
    for (int j = local; j < clen; j += nl) {
      const int tile = cbeg + j, mt = tile / NT, nt = tile - mt * NT;
      f32x16 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
      for (int kt = 0; kt < nkt; ++kt, ++it) {
        const unsigned s = it % NS, u = it / NS;
        wait_ge(&full[s], 4u * (u + 1u), &abort_flag);
        ORDER();
        const float* st = smem + s * STAGE;
        Frags fr[2];                                  // fragments one substep ahead of the MFMAs that consume them
        load_frags(fr[0], st, wr0, wc0, lane, 0);
#define STEP(CUR, NXT, KS) do { load_frags(fr[NXT], st, wr0, wc0, lane, KS); mma_frags(fr[CUR], acc); \
                                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x8, 16, 0); } while (0)
        STEP(0, 1, 1);
        STEP(1, 0, 2);
        STEP(0, 1, 3);
#undef STEP
        mma_frags(fr[1], acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&done[s], 1u);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            C[(size_t)(mt * 128 + wr0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + nt * 128 + wc0 + jj * 32 + l31] = acc[i][jj][r];
    }
  }
  __syncthreads();
  if (tid == 0 && abort_flag) atomicAdd(err, 1);
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{98304, 256, 1024, "1x1 1024 -> 256 at 14x14, 768 x 2 tiles (whole rounds)"}, {100352, 1024, 256, "1x1 256 -> 1024 at 14x14 (the layer of r04_whatif_pieces), 8 k-tiles"},
                          {100352, 256, 1024, "1x1 1024 -> 256 at 14x14, 32 k-tiles"}, {100864, 1152, 384, "ViT q/k/v projection, 12 k-tiles"}};
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    const double gflop = 2.0 * M * N * K / 1e9;
    float *A, *B, *C, *Cref;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cref, (size_t)M * N * 4));
    {
      std::vector<float> h((size_t)M * K);
      unsigned s = 12345u;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
      CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      std::vector<float> hb((size_t)N * K);
      for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
      CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    }
    const int tiles = (M / 128) * (N / 128);
    auto run = [&](int v, float* out) {
      switch (v) {
        case 0: hipLaunchKernelGGL(gemm_s1_k, dim3(tiles), dim3(256), 0, 0, M, N, K, A, B, out); break;
        case 1: hipLaunchKernelGGL((gemm_ws_k<2, 2>), dim3(cus * 2), dim3(512), 0, 0, M, N, K, A, B, out, err); break;
        case 2: hipLaunchKernelGGL((gemm_ws_k<3, 1>), dim3(cus), dim3(512), 0, 0, M, N, K, A, B, out, err); break;
        case 3: hipLaunchKernelGGL((gemm_ws_k<4, 1>), dim3(cus), dim3(512), 0, 0, M, N, K, A, B, out, err); break;
      }
    };
    const char* names[] = {"S1 shipped loop, 3 wg/CU", "WS 2 stages, 2 persistent wg/CU", "WS 3 stages, 1 persistent wg/CU", "WS 4 stages, 1 persistent wg/CU"};
    printf("%d x %d x %d (%s), %d tiles\n", M, N, K, sh.what, tiles);
    CK(hipMemset(Cref, 0, (size_t)M * N * 4));
    run(0, Cref); CK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)1 << 16), got((size_t)1 << 16);
    for (int v = 0; v < 4; ++v) {
      if (v > 0) {
        CK(hipMemset(C, 0, (size_t)M * N * 4));
        run(v, C); CK(hipDeviceSynchronize());
        long bad = 0;
        for (size_t off : {(size_t)0, (size_t)M * N / 2 - 4096, (size_t)M * N - ((size_t)1 << 16)}) {
          CK(hipMemcpy(ref.data(), Cref + off, ref.size() * 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(got.data(), C + off, got.size() * 4, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
        }
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (bad || herr) { printf("  %-34s WRONG: %ld mismatches, %d workgroups timed out\n", names[v], bad, herr); CK(hipMemset(err, 0, 4)); continue; }
      }
      for (int r = 0; r < 100; ++r) run(v, C);                      // warm the clocks
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 30; ++r) run(v, C);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      const float t = elapsed(e0, e1) / 30;
      printf("  %-34s %.3f ms  %6.1f TFLOP/s\n", names[v], t, gflop / t);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Cref));
  }
  return 0;
}
