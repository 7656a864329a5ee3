// Convolution (and Linear) as implicit GEMM on the fp32 matrix cores of gfx950.
//
// All three products of a conv layer run on v_mfma_f32_32x32x2_f32 (exact fp32: each result is a
// k-ordered fmaf chain), NHWC activations, OHWI filters:
//
//   fwd   : Y [N*Ho*Wo][K]   = im2col(X) [.][R*S*C] . W^T          rows gathered from X
//   dgrad : dX[N*H*W ][C]    = col2im-gather(dY) [.][taps*K] . W    rows grouped by stride-parity class so
//                                                                   that only the taps that hit are walked
//   wgrad : dW[K][R*S*C]     = dY^T [K][M] . im2col(X) [M][R*S*C]   split over M, fixed-order reduce
//
// Block = 256 threads = 4 waves (one per SIMD), 128x128 (or 256x64 / 64x128) output tile, K-step 32 (16 when the
// contraction is not a multiple of 32), each wave owns a (TM x TN) grid of 32x32 accumulators.  Operands are staged
// global -> VGPR -> LDS with the next tile's loads issued under the current tile's MFMAs (register prefetch, one LDS
// stage, ~37 KB) so that 3 blocks are resident per CU and hide each other's barriers.
//
// BatchNorm fusion (XF variants): the INPUT of a convolution that follows conv -> BN -> ReLU is never materialised;
// the producer's raw output is staged and relu(x * scale[c] + shift[c]) is applied on the way into LDS (forward A
// operand, wgrad X operand), with padding taps kept at zero.  The per-channel scale / shift come from ssv_bn_stats_finalize.
//
// LDS operand images:
//   ROWK  [row][BK k + 4 pad]   - row-major, k contiguous (source is k-contiguous: NHWC channels / OHWI);
//                                fragments are ds_read_b128: lane l takes row (l&31), k = 4*(l>>5)..+3,
//                                the 4 values feed 4 consecutive MFMAs.  Row stride 20 floats makes the
//                                16-lane b128 groups conflict-free.
//   KROW  [BK k][rows]         - k-major (source is row-contiguous: dY / X rows for wgrad, W rows for dgrad);
//                                fragments are ds_read_b32, 32 consecutive banks per half-wave.
// Both operands of one MFMA always take the same k = 8*ks + 4*(l>>5) + t, so any mix is consistent.
#include "common.h"
#include "split_bf16.h"
#include <string.h>
#include <type_traits>

// Compile-time switches of DIAGNOSTIC builds only (tools/probe/build_variant.sh builds a side library with them; the shipped Makefile defines
// none, and nothing here reads the environment): the what-ifs and A/Bs they served are tabulated in profiles/r02_experiments_step_time.txt.
#ifndef SSV_EXP_LDS_PAD
#define SSV_EXP_LDS_PAD 0   // diagnostic builds only: extra LDS floats per forward / data-gradient workgroup (forces a lower occupancy)
#endif
#ifdef SSV_CONV_VGPR          // diagnostic builds only: register cap of the forward / data-gradient kernels (amdgpu_num_vgpr(n) caps the unified file at 2n)
#define SSV_CONV_ATTR __attribute__((amdgpu_num_vgpr(SSV_CONV_VGPR)))
#else
#define SSV_CONV_ATTR
#endif
#ifndef SSV_WHATIF
#define SSV_WHATIF 0        // diagnostic builds only (TIMING what-ifs, results are wrong): bit 0 = the forward kernel's main loop issues no MFMAs, bit 1 = its
#endif                      // epilogue stores nothing, bit 2 = its activation operand is read through an empty descriptor (every load returns zero, no traffic)
#ifndef SSV_EXP_STAGGER
#define SSV_EXP_STAGGER 0   // diagnostic builds only: see conv_fwd_k
#endif
#ifndef SSV_EXP_STAGGER_N
#define SSV_EXP_STAGGER_N 3
#endif
#ifndef SSV_EXP_WGPRIO
#define SSV_EXP_WGPRIO 0    // diagnostic builds only: 1 = every workgroup of the forward kernel takes a STATIC issue priority from its arrival order on its CU
#endif                      // (consecutive arrivals get 0, 1, 2 -> s_setprio 0, 1, 3): does asymmetry between the residents break their lock-step?
#ifndef SSV_EXP_PRIO
#define SSV_EXP_PRIO 0      // diagnostic builds only: s_setprio level of a wave while it is in the MFMA part of a k-tile (0 = never raised, the shipped behaviour)
#endif
#ifndef SSV_CONV_WGPC
#define SSV_CONV_WGPC 3     // resident workgroups per CU the forward / data-gradient kernels are compiled for
#endif

namespace {

#if SSV_EXP_WGPRIO
__device__ unsigned g_cu_arrivals[4096];
__device__ __forceinline__ void wg_static_priority() {
  __shared__ int s_prio;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // HW_REG_XCC_ID [3:0]
    const unsigned idx = ((xcc & 15) << 8) | ((hw >> 8) & 0xff);
    s_prio = (int)(atomicAdd(&g_cu_arrivals[idx & 4095], 1u) % 3u);
  }
  __syncthreads();
  const int pr = s_prio;
  if (pr == 2) __builtin_amdgcn_s_setprio(3);
  else if (pr == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
#endif

constexpr int GBK = 16;   // K-step of the generic (scalar gather) path and the granularity of wgrad row chunks

// blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous run of logical tiles so that tiles
// sharing an operand panel hit the same L2.  Bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
#ifdef SSV_NO_XCD_REMAP
  return orig;
#endif
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

struct ConvKP {
  int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
  int M;              // fwd/wgrad: N*Ho*Wo
  int RSC;            // R*S*C
  FastDiv dHoWo, dWo, dC, dS, dWp;   // dWp: W + 2 (the halo loader's records per staged row)
  float* aux_out;     // fused-activation variants only: forward also writes gelu(y) here; statistics variant: pmean
  float* aux_out2;    // statistics variant: pm2
  const float* aux_in;  //                                   dgrad multiplies by gelu'(aux_in) before the addend
  const float* xf_scale;  // XF variants: per-input-channel affine of the fused BatchNorm + ReLU applied to the staged input
  const float* xf_shift;
  ssv_bn_gate gate;       // GATE variants (this launch computes the gradient w.r.t. a BatchNorm + ReLU output): see epilogue_vec
  const float* dyin_x;    // DYF variants: the staged operand is the BatchNorm backward's dx = A[k] * g + B[k] * (x - mean[k]) + D[k], formed
  const float* dyin_coef; //   on load from the gated gradient g (the operand pointer) and the BatchNorm's input x; coef = A | mean | B | D
  // SUM variants (OPM 2): the staged operand is the closing activation of a residual unit, a = relu(x * scale + shift + res * rscale + rshift),
  // formed on load from the unit's last conv output x (the operand pointer) and its shortcut; the column-tile-0 workgroups also WRITE it
  const float* sum_res; const float* sum_scale; const float* sum_shift; const float* sum_rscale; const float* sum_rshift;
  float* sum_out; uint8_t* sum_mask;
  // batched launches (blockIdx.y = batch index: the 16 GEMMs of a Winograd convolution): element strides of the three operands, 0 otherwise
  long long bs_a, bs_b, bs_o;
  // compact addend (ADDS2 variants): the addend is [N][add_H2][add_W2][K] and belongs to the output pixels with even (h, w) - the data gradient
  // of a 1x1 / stride-2 projection shortcut, computed as a dense GEMM on the subsampled grid instead of a full-resolution tensor of 3/4 zeros
  int add_H2, add_W2;
  // block-diagonal filter banks (a grouped convolution expanded to dense form, ssv_group_expand): input / output channels per group, 0 = dense.
  // Output columns [n0, n1) only depend on the input channels of the groups they fall into, so a tile's contraction runs over that channel range
  // instead of all of C (fwd: channels, dgrad: output channels), and a weight-gradient tile whose rows and columns share no group is skipped.
  int Cg, Kg;
  // SSV_ARITH_BF16X3 (the SP kernel variants, csrc/split_bf16.h): the weight operand pre-split into three bf16 planes [3][K][RSC] (batched launches:
  // [3][batch][K][RSC], ONE ssv_split_planes call over all the batch's filters); NULL in launches whose operands are both activations.  wp_stride: elements per plane
  const unsigned short* w_planes;
  long long wp_stride;
};

// [lo, hi) of the contraction channels (multiples of `bk`, hi capped at `ctot`) that the output columns [n0, n1) of a block-diagonal product can see:
// the columns come in groups of `og`, each fed by `ig` consecutive contraction channels
__device__ __forceinline__ void group_span(int n0, int n1, int og, int ig, int bk, int ctot, int& lo, int& hi) {
  const int glo = n0 / og, ghi = (n1 - 1) / og;
  lo = glo * ig / bk * bk;
  hi = min(ctot, ((ghi + 1) * ig + bk - 1) / bk * bk);
}

// The halo loader (conv_fwd_k, C4 == 3): staged records (one pixel x 16 channels, 64 bytes at an 80-byte stride: ds_read_b128 of 16 consecutive pixels' fragments
// then hit 16 distinct bank quads) a workgroup may hold PER STAGE - its 256 output pixels' rows plus one above and below, each with a padding record left and
// right - and three records of zeros that out-of-image row taps are pointed at (one per column tap).  464 = 8 rows x 58 records (56 x 56 maps).  Two stages.
constexpr int HALO_REC = 464;
constexpr int HALO_RS = 20;                                   // record stride in floats
constexpr int HALO_STAGE = (HALO_REC + 3) * HALO_RS;          // floats per stage
constexpr int HALO_FLOATS = 2 * HALO_STAGE;
// most records a 256-pixel tile of a W-wide map can need (the tile may start anywhere in a row)
static inline int halo_records(int W) { return ((256 + W - 2) / W + 1 + 2) * (W + 2); }

constexpr int XF_MAXC = 1024;   // input channels an XF forward kernel keeps (scale, shift) in LDS for

// erf-form GELU and its derivative: the shared definitions of common.h (the stand-alone kernels of vit.hip use the same ones)
__device__ __forceinline__ float gelu_f(float v) { return ssv_gelu(v); }
__device__ __forceinline__ float gelu_grad_f(float v) { return ssv_gelu_grad(v); }

// Buffer loads: a 128-bit resource descriptor (base, byte size) in SGPRs + a 32-bit per-lane byte offset + a uniform
// SGPR byte offset.  Anything out of [0, size) reads as ZERO in hardware, so padding taps, ragged tile edges and rows
// past the end of a tensor cost one select on the OFFSET (or nothing) instead of exec-mask branches or data selects.
// Needs tensors below 2 GiB (offsets stay non-negative ints; OOB_OFF + any in-range offset is still out of range).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int OOB_OFF = (int)0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(rsrc_t rs, int voff_bytes, int soff_bytes) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, soff_bytes, 0));
}

// 16-byte loads at 4-byte-aligned offsets (the row-taps stem: windows of 3 S floats start at any pixel).  A window may straddle the START or the
// END of the tensor - the first / last pixels of the whole batch - where a buffer load that is partly out of range returns zeros for ALL of its
// dwords.  straddle_off() moves such a load to the nearest offset that is wholly inside and reports how many floats the wanted window lies
// from it (rot in -3 .. 3); straddle_fix() shifts the elements accordingly (elements from outside the tensor become 0).  rot != 0 is rare
// (a handful of lanes of the first and last workgroup), so callers branch on it.
__device__ __forceinline__ int straddle_off(int off, int bytes, int& rot) {
  rot = 0;
  if (off == OOB_OFF) return OOB_OFF;
  if (off <= -16 || off >= bytes) return OOB_OFF;                  // wholly outside
  const int c = off < 0 ? 0 : (off > bytes - 16 ? bytes - 16 : off);
  rot = (off - c) >> 2;                                            // off and c are multiples of 4
  return c;
}
__device__ __forceinline__ f32x4 straddle_fix(f32x4 v, int rot) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int sidx = e + rot;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t = sidx == q ? v[q] : t;
    o[e] = t;
  }
  return o;
}

template <int TM, int TN, bool A_ROWK, bool B_ROWK, int LDA, int LDB, int BK>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wr0, int wc0, int lane, f32x16 (&acc)[TM][TN]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < BK / 8; ++ks) {
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if constexpr (A_ROWK) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&As[(wr0 + tm * 32 + l31) * LDA + ks * 8 + 4 * h]);
        a[tm][0] = v[0]; a[tm][1] = v[1]; a[tm][2] = v[2]; a[tm][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[tm][t] = As[(ks * 8 + 4 * h + t) * LDA + wr0 + tm * 32 + l31];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if constexpr (B_ROWK) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[(wc0 + tn * 32 + l31) * LDB + ks * 8 + 4 * h]);
        b[tn][0] = v[0]; b[tn][1] = v[1]; b[tn][2] = v[2]; b[tn][3] = v[3];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[tn][t] = Bs[(ks * 8 + 4 * h + t) * LDB + wc0 + tn * 32 + l31];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], b[tn][t], acc[tm][tn], 0, 0, 0);
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

#ifdef SSV_STAMP
__device__ unsigned long long g_stamps[8];
#endif

template <int TM, int TN> struct Frags { float a[TM][4], b[TN][4]; };

// fragments of k-substep ks (8 k-values) of the staged tile: lane l takes k = 8*ks + 4*(l>>5) + t, t = 0..3
template <int TM, int TN, bool A_ROWK, bool B_ROWK, int LDA, int LDB>
__device__ __forceinline__ void load_frags(Frags<TM, TN>& f, const float* __restrict__ As, const float* __restrict__ Bs,
                                           int wr0, int wc0, int lane, int ks) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if constexpr (A_ROWK) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&As[(wr0 + tm * 32 + l31) * LDA + ks * 8 + 4 * h]);
      f.a[tm][0] = v[0]; f.a[tm][1] = v[1]; f.a[tm][2] = v[2]; f.a[tm][3] = v[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f.a[tm][t] = As[(ks * 8 + 4 * h + t) * LDA + wr0 + tm * 32 + l31];
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    if constexpr (B_ROWK) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[(wc0 + tn * 32 + l31) * LDB + ks * 8 + 4 * h]);
      f.b[tn][0] = v[0]; f.b[tn][1] = v[1]; f.b[tn][2] = v[2]; f.b[tn][3] = v[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f.b[tn][t] = Bs[(ks * 8 + 4 * h + t) * LDB + wc0 + tn * 32 + l31];
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void mma_frags(const Frags<TM, TN>& f, f32x16 (&acc)[TM][TN]) {
#if SSV_WHATIF & 1
  // what-if: the fragments are still read from LDS (kept alive by one add per fragment), no matrix instruction is issued
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[tm][tn][t] += f.a[tm][t] + f.b[tn][t];
#else
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][t], f.b[tn][t], acc[tm][tn], 0, 0, 0);
#endif
}

// compile-time interleave of one MFMA group with N memory instructions of kind `mask` (LLVM sched groups:
// 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write): MFMA, mem, MFMA, mem, ... then the remaining MFMAs.
// A VMEM / DS-write wave-instruction holds the wave's issue port for ~50-70 cycles - about one fp32 MFMA (64 cycles
// on the matrix pipe) - so alternating them keeps the pipe fed while the wave issues its memory traffic.
#define SSV_INTERLEAVE(NMFMA, NMEM, MASK)                                            \
  do {                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < (NMEM) && i_ < (NMFMA); ++i_) {         \
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                               \
      __builtin_amdgcn_sched_group_barrier((MASK), 1, 0);                            \
    }                                                                                \
    if ((NMFMA) > (NMEM)) __builtin_amdgcn_sched_group_barrier(0x8, (NMFMA) - (NMEM), 0); \
  } while (0)

// K loop over staged tiles: ONE LDS stage, two barriers per tile (small LDS footprint -> 3 workgroups per CU hide each other's
// barriers).  load_tile() issues the next tile's global loads into registers, store_tile(0) writes those registers to LDS.
// xform_tile(): register-to-register transform of the prefetched tile (fused BatchNorm input / BatchNorm-backward operand), run right before
// the tile is stored to LDS.  (Measured: issuing it under the last substep's MFMAs instead is SLOWER, 254.3 -> 256.0 ms per step - its
// s_waitcnt for the prefetched loads then stalls that wave's MFMA stream one substep early.)
struct NoXform { __device__ __forceinline__ void operator()() const {} };
// FLUSH > 0 (the blocked weight gradient, ssv_gemm_batched_wgrad_blocked): two-level accumulation - every FLUSH k-tiles the MFMA accumulators are added into a
// second set and cleared, so no fp32 chain of products is longer than FLUSH * BK rows (a chain's rounding error grows with its length); the second set sums
// nkt / FLUSH block results.  Costs TM * TN * 16 registers and as many v_add_f32 per FLUSH k-tiles.
template <int TM, int TN, bool A_ROWK, bool B_ROWK, int LDA, int LDB, int BK, int NLD, int FLUSH = 0, class LoadTile, class StoreTile, class Xform = NoXform>
__device__ __forceinline__ void k_loop(int nkt, const float* As, const float* Bs, int wr0, int wc0, int lane,
                                       f32x16 (&acc)[TM][TN], LoadTile&& load_tile, StoreTile&& store_tile, Xform&& xform_tile = NoXform()) {
  if (nkt <= 0) return;
  f32x16 hi[FLUSH > 0 ? TM : 1][FLUSH > 0 ? TN : 1];
  if constexpr (FLUSH > 0) zero_acc<TM, TN>(hi);
  constexpr int NS = BK / 8;
  constexpr int NM = 4 * TM * TN;            // MFMAs per substep
  load_tile();
  xform_tile();
  store_tile(0);
  __syncthreads();
  {
#ifdef SSV_STAMP   // diagnostic build only (tools/): where does one k-tile spend its cycles?  Never in the shipped library.
    unsigned long long t_ld = 0, t_mma = 0, t_b1 = 0, t_st = 0, t_b2 = 0;
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); var += now_ - last_; last_ = now_; } while (0)
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define STAMP(var) do {} while (0)
#endif
#ifdef SSV_STAMP
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;
      if (more) load_tile();                 // next tile's global loads fly under this tile's MFMAs
      STAMP(t_ld);
      mma_ktile<TM, TN, A_ROWK, B_ROWK, LDA, LDB, BK>(As, Bs, wr0, wc0, lane, acc);
      STAMP(t_mma);
      __syncthreads();
      STAMP(t_b1);
      if (more) { xform_tile(); store_tile(0); STAMP(t_st); __syncthreads(); STAMP(t_b2); }
    }
#else
    // Branch-free body (one scheduling region per tile): the next tile's NLD buffer loads are issued one per MFMA of
    // the first substep instead of in front of the MFMAs (a VMEM wave-instruction holds the issue port ~65 cycles).
    // The loads issued during the last tile fall outside the operands and are never stored (buffer loads cannot fault).
    Frags<TM, TN> fr;
    for (int kt = 0; kt < nkt; ++kt) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SSV_EXP_PRIO > 0) __builtin_amdgcn_s_setprio(SSV_EXP_PRIO);
      load_frags<TM, TN, A_ROWK, B_ROWK, LDA, LDB>(fr, As, Bs, wr0, wc0, lane, 0);
      load_tile();
      mma_frags<TM, TN>(fr, acc);
      if (NLD > 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, A_ROWK && B_ROWK ? TM + TN : (A_ROWK ? TM + 4 * TN : 4 * (TM + TN)), 0);
        SSV_INTERLEAVE(NM, NLD, 0x20);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 1; ks < NS; ++ks) {
        load_frags<TM, TN, A_ROWK, B_ROWK, LDA, LDB>(fr, As, Bs, wr0, wc0, lane, ks);
        mma_frags<TM, TN>(fr, acc);
      }
      if constexpr (SSV_EXP_PRIO > 0) __builtin_amdgcn_s_setprio(0);
      if constexpr (FLUSH > 0) {
        if ((kt + 1) % FLUSH == 0) {               // uniform
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              hi[tm][tn] += acc[tm][tn];
#pragma unroll
              for (int j = 0; j < 16; ++j) acc[tm][tn][j] = 0.f;
            }
        }
      }
      __syncthreads();
      if (kt + 1 < nkt) { xform_tile(); store_tile(0); __syncthreads(); }
    }
    if constexpr (FLUSH > 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = hi[tm][tn] + acc[tm][tn];
    }
#endif
#ifdef SSV_STAMP
    if (threadIdx.x == 0) {
      atomicAdd(&g_stamps[0], t_ld); atomicAdd(&g_stamps[1], t_mma); atomicAdd(&g_stamps[2], t_b1);
      atomicAdd(&g_stamps[3], t_st); atomicAdd(&g_stamps[4], t_b2); atomicAdd(&g_stamps[5], (unsigned long long)nkt);
    }
#endif
#undef STAMP
  }
}

// Two LDS stages, ONE barrier per k-tile (round 4 experiment S2, plain GEMM-shaped launches with long k-loops only): tile kt is multiplied out of stage
// kt & 1 while tile kt + 1 goes registers -> the other stage (its readers passed the last barrier) and tile kt + 2 global -> registers; the ds_writes are
// interleaved one per MFMA of the first substep, the buffer loads one per MFMA of the second.  2 x 36.9 KB of LDS: 2 workgroups per CU.
template <int TM, int TN, int LDT, int BK, int NLD, int STAGE_FLOATS, class LoadTile, class StoreTile>
__device__ __forceinline__ void k_loop2(int nkt, const float* As, const float* Bs, int wr0, int wc0, int lane,
                                        f32x16 (&acc)[TM][TN], LoadTile&& load_tile, StoreTile&& store_tile) {
  if (nkt <= 0) return;
  constexpr int NS = BK / 8;
  constexpr int NM = 4 * TM * TN;
  static_assert(NS >= 2, "two substeps carry the staging traffic");
  load_tile();
  store_tile(0);
  load_tile();
  __syncthreads();
  Frags<TM, TN> fr;
  for (int kt = 0; kt < nkt; ++kt) {
    const float* a = As + (kt & 1) * STAGE_FLOATS;
    const float* b = Bs + (kt & 1) * STAGE_FLOATS;
    __builtin_amdgcn_sched_barrier(0);
    load_frags<TM, TN, true, true, LDT, LDT>(fr, a, b, wr0, wc0, lane, 0);
    store_tile((kt + 1) & 1);                 // registers of tile kt + 1 (zeros past the last tile: never read)
    mma_frags<TM, TN>(fr, acc);
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    SSV_INTERLEAVE(NM, NLD, 0x200);
    __builtin_amdgcn_sched_barrier(0);
    load_frags<TM, TN, true, true, LDT, LDT>(fr, a, b, wr0, wc0, lane, 1);
    load_tile();                              // tile kt + 2 -> registers
    mma_frags<TM, TN>(fr, acc);
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    SSV_INTERLEAVE(NM, NLD, 0x20);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 2; ks < NS; ++ks) {
      load_frags<TM, TN, true, true, LDT, LDT>(fr, a, b, wr0, wc0, lane, ks);
      mma_frags<TM, TN>(fr, acc);
    }
    __syncthreads();
  }
}

// The k-loop of the SSV_ARITH_BF16X3 variants (csrc/split_bf16.h): the same register-prefetch loop as k_loop - ONE LDS stage, two barriers per k-tile, the next
// tile's global loads in flight under this tile's MFMAs - over LDS images of three bf16 planes per operand.  A k-tile is 32 of the contraction = one
// v_mfma_f32_16x16x32_bf16 per (16 x 16 tile, piece product): 6 x TM16 x TN16 per wave.  KROWA / KROWB: that operand's image is k-major (both in the weight
// gradient, the weights in the strided data gradient; transposed reads).
// The structure was chosen by measurement (tools/probe/gemm_split_tuned_probe.hip, profiles/r06_probe_split_tuned.txt): two LDS stages with one barrier per tile lose
// at every occupancy the LDS allows (106-160 against 175-185 TFLOP/s at 4096^3), the 16 x 16 x 32 instruction beats 32 x 32 x 16 by 6-15 % on every shape (the chip
// holds a higher clock under it), and the split's vector work hides under the other resident workgroups' MFMAs (a build without the residual arithmetic: +0-1 %).
template <int TM16, int TN16, bool KROWA, bool KROWB, int RBA, int RBB, int FLUSH = 0, int JSPLIT = 1, bool DUAL = false, class LoadTile, class StoreTile, class Xform = NoXform>
__device__ __forceinline__ void k_loop_split(int nkt, const unsigned char* As, int a_plane, const unsigned char* Bs, int b_plane, int wr0, int wc0, int lane,
                                             f32x4 (&acc)[TM16][TN16], LoadTile&& load_tile, StoreTile&& store_tile, Xform&& xform_tile = NoXform()) {
  if (nkt <= 0) return;
  f32x4 hi[FLUSH > 0 ? TM16 : 1][FLUSH > 0 ? TN16 : 1];
  if constexpr (FLUSH > 0) {
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j) hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // DUAL: the five small piece products in a second accumulator set (splitbf::mma6x2) - weight gradients and forward products with a long contraction
  // (SP == 2: R * S * C > SP_DUAL_FROM), where six roundings per 32 products in ONE accumulator end 5-7 % over the fp32-MFMA kernel's error
  f32x4 lo[DUAL ? TM16 : 1][DUAL ? TN16 : 1];
  if constexpr (DUAL) {
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j) lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  load_tile();
  xform_tile();
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    load_tile();                               // past the last tile: out-of-range offsets read zeros (or in-buffer rows never stored)
    // JSPLIT > 1: the column fragments of TN16 / JSPLIT tiles at a time, the row fragments re-read per group (tried for the register-bound variants, round 6: the
    // compiler keeps MORE registers live that way - 56 instead of 36 spilled dwords in the closing-sum variant - so every launch uses 1)
    constexpr int TJ = TN16 / JSPLIT;
    static_assert(TN16 % JSPLIT == 0, "column tiles per group");
#pragma unroll
    for (int jg = 0; jg < JSPLIT; ++jg) {
      bf16x8 fb[TJ][3];
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if constexpr (KROWB) fb[j][q] = splitbf::krow_frag<RBB>(Bs + q * b_plane, wc0 + 16 * (jg * TJ + j), lane);
          else fb[j][q] = splitbf::rowk_frag(Bs + q * b_plane, wc0 + 16 * (jg * TJ + j), lane);
        }
#pragma unroll
      for (int i = 0; i < TM16; ++i) {
        bf16x8 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if constexpr (KROWA) fa[q] = splitbf::krow_frag<RBA>(As + q * a_plane, wr0 + 16 * i, lane);
          else fa[q] = splitbf::rowk_frag(As + q * a_plane, wr0 + 16 * i, lane);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if constexpr (DUAL) splitbf::mma6x2(acc[i][jg * TJ + j], lo[i][jg * TJ + j], fb[j], fa);
          else splitbf::mma6(acc[i][jg * TJ + j], fb[j], fa);
        }
      }
    }
    if constexpr (FLUSH > 0) {
      if ((kt + 1) % FLUSH == 0) {               // uniform
#pragma unroll
        for (int i = 0; i < TM16; ++i)
#pragma unroll
          for (int j = 0; j < TN16; ++j) { hi[i][j] += acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
    __syncthreads();
    if (kt + 1 < nkt) { xform_tile(); store_tile(0); __syncthreads(); }
  }
#pragma unroll
  for (int i = 0; i < TM16; ++i)
#pragma unroll
    for (int j = 0; j < TN16; ++j) {
      if constexpr (FLUSH > 0) acc[i][j] = hi[i][j] + acc[i][j];
      if constexpr (DUAL) acc[i][j] += lo[i][j];
    }
}

// Vectorised epilogue: a wave's accumulators hold one COLUMN per lane (32 consecutive columns per 32x32 tile), so a direct
// store is 4 bytes per lane and 16*TM*TN instructions.  Staging 32-row slabs of the wave's tile through its own LDS
// region turns that into 16-byte-per-lane stores of whole 256-byte row segments (4x fewer store instructions, and the
// bias / residual addend are read 16 bytes per lane too).  LDS traffic of one wave is in order, so no barrier is needed
// inside; the caller has passed the barrier that ends the K loop.  row_off(r) gives the element offset of output row r of
// the wave tile, or -1.
//
// Everything that touches global memory here is a BUFFER operation on a (base, size) descriptor with a per-lane byte offset: rows past
// the tensor and columns past its width get the out-of-range offset, whose loads return zero and whose stores are dropped by the
// hardware.  So the body is straight-line code - no exec-mask region per row - and the loads a row needs (residual addend, gate
// operands) are issued for a whole batch of rows before the first of their stores.  With a branch per row and plain pointers (the
// addend may alias the output when accumulating in place) every row was a dependent HBM round trip: the unit-input gradients of
// layer1 / layer2 ran latency-bound at 35-60 TFLOP/s.  The aliasing that exists is "this lane reads the element it then writes" - a
// data dependency the issue order cannot break.
//
// STATS: also reduce, per output column, the rows this wave stores (a group of TM*32 consecutive output rows) to (mean, centred sum
// of squares) - shifted sums around the group's first row, then a shuffle over the lanes that share a column - and write them as
// one partial of the BatchNorm that follows (layout of bn_stats_finalize_k with rows-per-block = TM*32).
// GATE 3 = GATE 2 plus a SECOND reduction target: the gated gradient g of a unit's closing activation is also the gradient w.r.t. the output
// of the projection shortcut's BatchNorm (no ReLU there), so sum g * xhat2 against that BatchNorm's input (bn.x2 / mean2 / invstd2 ->
// bn.psum_gx2; its sum g is psum_g) comes out of the same epilogue and the shortcut's backward needs no reduction pass either.
// GATE (1: the ReLU bit recomputed as x * scale + shift > 0, 2: the ReLU bit from the forward's byte mask): the tile is the gradient
// w.r.t. the OUTPUT of a BatchNorm (+ residual) + ReLU whose input x is bn.x.  The value stored is g = relu'(.) * (acc + addend), and
// the wave leaves, per column, the two sums the BatchNorm backward needs over its rows - sum g and sum g * xhat - as one partial
// (bn.psum_g / bn.psum_gx row `group`): the backward's reduction pass over (dy, mask, x) disappears, at the price of reading x here.
// EPI: 1 = also write gelu(v) to out_act (Linear + GELU forward), 2 = multiply by gelu'(gate) before the addend (its backward),
//      3 = write gelu(v) ONLY (the same forward when no backward will ask for the pre-activation),
//      4 = write gelu'(v) to out and gelu(v) to out_act (the forward when a backward WILL run: nothing in the backward of fc1 -> GELU -> fc2 reads the
//          pre-activation itself, only gelu' of it - taking it here, where the cdf is at hand for gelu(v), leaves the backward a plain multiply),
//      5 = multiply by the stored factor `gate` (= gelu'(h) written by 4) before the addend.
__device__ __forceinline__ void bstore4(rsrc_t rs, int voff_bytes, f32x4 v) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
#if SSV_WHATIF & 2
  if (v[0] == 1.2345e38f)      // what-if: (practically) never true - the value stays live, no store reaches memory
#endif
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rs, voff_bytes, 0, 0);
}

// AddOff: where the addend of output row r lives when it is NOT laid out like the output: a functor (row of the wave tile) -> element offset
// of that row in the addend tensor, or -1 for "this row has no addend" (the compact gradient of a stride-2 shortcut: only rows with even
// (h, w) carry one).  SameOff = the addend has the output's layout.
struct SameOff {};
// one 32-row slab (slab tm) of a wave's accumulators -> its LDS staging area [32][TN * 32 + 4]: the fp32-MFMA tiles hold a column per lane (scalar writes), the
// 16 x 16 tiles of the bf16 piece products (splitbf::mma6: column operand first) four consecutive columns of one row per lane (16-byte writes)
template <int TM, int TN>
__device__ __forceinline__ void stage_slab(const f32x16 (&acc)[TM][TN], float* __restrict__ ep, int tm, int lane) {
  constexpr int LDE = TN * 32 + 4;
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int j = 0; j < 16; ++j) ep[((j & 3) + 8 * (j >> 2) + 4 * h) * LDE + tn * 32 + l31] = acc[tm][tn][j];
}
template <int TM, int TN>
__device__ __forceinline__ void stage_slab(const f32x4 (&acc)[2 * TM][2 * TN], float* __restrict__ ep, int tm, int lane) {
  constexpr int LDE = TN * 32 + 4;
  const int fr = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int jj = 0; jj < 2 * TN; ++jj) *reinterpret_cast<f32x4*>(&ep[(16 * ii + fr) * LDE + 16 * jj + 4 * kg]) = acc[2 * tm + ii][jj];
}
template <bool SP, class A, class B> __device__ __forceinline__ auto& sel_acc(A& a, B& b) { if constexpr (SP) return b; else return a; }

template <int TM, int TN, int EPI = 0, bool STATS = false, int GATE = 0, class Acc, class RowOff, class AddOff = SameOff>
__device__ __forceinline__ void epilogue_vec(const Acc& acc, float* __restrict__ ep, int lane, int col0, int ncols,
                                             const float* __restrict__ bias, const float* addend, float* out, long long out_elems, RowOff&& row_off,
                                             float* out_act = nullptr, const float* gate = nullptr,
                                             float* pmean = nullptr, float* pm2 = nullptr, int group_rows = 0,
                                             const ssv_bn_gate* bn = nullptr, long long group = 0, AddOff add_off = AddOff(), long long add_elems = 0) {
  constexpr int LDE = TN * 32 + 4, C4 = TN * 8, RPI = 64 / C4, NP = 32 / RPI;
  constexpr bool GMASK = GATE >= 2, GX2 = GATE == 3;
  // rows per batch: the loads of a batch are in flight together.  GATE 1 lives in kernels compiled for 3 workgroups per CU (register budget: 2 rows);
  // the two-target gate (GATE 3) is compiled for 2 per CU and uses the room for 4 rows in flight (r03 x1: 1.43 -> 1.34 ms on the 56x56 unit input)
  constexpr int HB0 = GATE == 1 ? 2 : 4;
  constexpr int HB = NP < HB0 ? NP : HB0;
  const int r_in = lane / C4, c4 = lane % C4;
  const int gcol = col0 + c4 * 4;
  const bool cok = gcol < ncols && r_in < RPI;          // TN = 3 (C4 = 24 lanes per row): the last 16 lanes of a wave have no row in a pass
  const unsigned bytes = (unsigned)(out_elems * 4);
  const rsrc_t r_out = make_rsrc(out, bytes);
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 st_p = {0.f, 0.f, 0.f, 0.f}, st_s1 = {0.f, 0.f, 0.f, 0.f}, st_s2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 g_mu = {0.f, 0.f, 0.f, 0.f}, g_is = {0.f, 0.f, 0.f, 0.f}, g_sc = {0.f, 0.f, 0.f, 0.f}, g_sh = {0.f, 0.f, 0.f, 0.f};
  f32x4 g_mu2 = {0.f, 0.f, 0.f, 0.f}, g_is2 = {0.f, 0.f, 0.f, 0.f}, st_s3 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (GATE != 0) {
    if (cok) {
      g_mu = *reinterpret_cast<const f32x4*>(bn->mean + gcol); g_is = *reinterpret_cast<const f32x4*>(bn->invstd + gcol);
      if constexpr (GATE == 1) { g_sc = *reinterpret_cast<const f32x4*>(bn->scale + gcol); g_sh = *reinterpret_cast<const f32x4*>(bn->shift + gcol); }
      if constexpr (GX2) { g_mu2 = *reinterpret_cast<const f32x4*>(bn->mean2 + gcol); g_is2 = *reinterpret_cast<const f32x4*>(bn->invstd2 + gcol); }
    }
  }
  if (bias && cok) b4 = *reinterpret_cast<const f32x4*>(bias + gcol);

  const bool ADD = addend != nullptr;            // uniform
  // The row loop exists twice - with and without the addend loads - and the uniform branch picks one: gfx9 counts loads and stores on ONE in-order
  // counter, so waiting for ANY load issued after a store also waits for that store to drain.  With the addend test inside the loop the compiler put an
  // s_waitcnt vmcnt(0) at every join, and an epilogue WITHOUT an addend (every forward convolution, the Winograd products, the weight gradients)
  // stored four rows, waited for them to reach memory, stored the next four ...  The copy without loads has nothing to wait for (found while looking
  // for what makes a 64 -> 256 1x1 layer take the SUM of its MFMA and HBM times, r03 x3 - it was not this: the layer times did not move).
  auto rows = [&](auto add_tag) {
    constexpr int ADDM = decltype(add_tag)::value;      // 0 / 1: no addend / addend (compile time); 2: decided per launch by the uniform test (gated epilogues)
    constexpr bool ADD_SAME = std::is_same<AddOff, SameOff>::value;
    const rsrc_t r_add = make_rsrc(ADD ? addend : out, ADD_SAME ? bytes : (unsigned)(add_elems * 4));
    const rsrc_t r_gx = make_rsrc(GATE != 0 ? bn->x : out, bytes);
    const rsrc_t r_gm = make_rsrc(GMASK ? reinterpret_cast<const float*>(bn->mask) : out, bytes / 16);
    const rsrc_t r_gx2 = make_rsrc(GX2 ? bn->x2 : out, bytes);
    const rsrc_t r_gate = make_rsrc((EPI == 2 || EPI == 5) ? gate : out, bytes);
    const rsrc_t r_act = make_rsrc((EPI == 1 || EPI == 4) ? out_act : out, bytes);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      stage_slab<TM, TN>(acc, ep, tm, lane);
      if constexpr (STATS) { if (tm == 0) st_p = *reinterpret_cast<const f32x4*>(&ep[c4 * 4]); }     // pivot: first row of the group
      // Rows go through in steps of HB: a step issues its loads (addend, gate operands: in flight together), then computes and stores.  (Issuing the NEXT
      // step's loads before this step's stores - so that waiting for them does not also wait for the stores on gfx9's single in-order counter - was
      // measured level in the step, r03 e11, and is not built.)
      constexpr int HP = HB;
      constexpr int NSTEP = NP / HP;
      int voff[1][HP];
      f32x4 av[1][HP], xv[1][HP], gv[1][HP], xv2[1][GX2 ? HP : 1];
      unsigned mb[1][HP];
      auto issue = [&](int st, int sl) {
#pragma unroll
        for (int i = 0; i < HP; ++i) {
          const long long off = row_off(tm * 32 + (st * HP + i) * RPI + r_in);
          voff[sl][i] = (off >= 0 && cok) ? (int)(off + gcol) * 4 : OOB_OFF;        // tensors stay below 2^29 elements (check_desc)
          int aoff = voff[sl][i];
          if constexpr (!ADD_SAME) {
            const long long ao = add_off(tm * 32 + (st * HP + i) * RPI + r_in);
            aoff = (ao >= 0 && voff[sl][i] != OOB_OFF) ? (int)(ao + gcol) * 4 : OOB_OFF;
          }
          if constexpr (ADDM == 2) { if (ADD) av[sl][i] = bload4(r_add, aoff, 0); else av[sl][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          else if constexpr (ADDM == 1) av[sl][i] = bload4(r_add, aoff, 0);
          else av[sl][i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (EPI == 2 || EPI == 5) gv[sl][i] = bload4(r_gate, voff[sl][i], 0);
          if constexpr (GATE != 0) xv[sl][i] = bload4(r_gx, voff[sl][i], 0);
          if constexpr (GMASK) mb[sl][i] = __builtin_amdgcn_raw_buffer_load_b8(r_gm, voff[sl][i] == OOB_OFF ? OOB_OFF : voff[sl][i] >> 4, 0, 0);
          if constexpr (GX2) xv2[sl][i] = bload4(r_gx2, voff[sl][i], 0);
        }
      };
      auto finish = [&](int st, int sl) {
#pragma unroll
        for (int i = 0; i < HP; ++i) {
          const int r = (st * HP + i) * RPI + r_in;
          f32x4 v = *reinterpret_cast<const f32x4*>(&ep[r * LDE + c4 * 4]);
          const bool ok = voff[sl][i] != OOB_OFF;
          if constexpr (STATS) {
            f32x4 dv = v - st_p;
            dv[0] = ok ? dv[0] : 0.f; dv[1] = ok ? dv[1] : 0.f; dv[2] = ok ? dv[2] : 0.f; dv[3] = ok ? dv[3] : 0.f;
            st_s1 += dv; st_s2 += dv * dv;
          }
          v += b4;
          if constexpr (EPI == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(gv[sl][i][e]);
          }
          if constexpr (EPI == 5) v *= gv[sl][i];
          v += av[sl][i];
          if constexpr (GATE != 0) {
            if constexpr (GATE == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xv[sl][i][e], g_sc[e], g_sh[e]) > 0.f ? v[e] : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (mb[sl][i] >> e) & 1u ? v[e] : 0.f;
            }
            // an out-of-range row / column contributes nothing: its v is garbage but its mask byte read as 0 (GATE 2) or its x as 0
            // (GATE 1: the gate is then shift > 0, so force the product to zero explicitly)
            f32x4 gvv = v;
            if constexpr (GATE == 1) { gvv[0] = ok ? v[0] : 0.f; gvv[1] = ok ? v[1] : 0.f; gvv[2] = ok ? v[2] : 0.f; gvv[3] = ok ? v[3] : 0.f; }
            st_s1 += gvv;
            st_s2 += gvv * ((xv[sl][i] - g_mu) * g_is);
            if constexpr (GX2) st_s3 += gvv * ((xv2[sl][i] - g_mu2) * g_is2);      // an out-of-range row has gvv == 0 (mask byte 0)
          }
          if constexpr (EPI == 3) {          // Linear + GELU where only the activation is wanted (a forward without a backward): one store
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
          }
          if constexpr (EPI == 4) {          // gelu'(v) in v's place, gelu(v) beside it: one cdf serves both
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float cdf = ssv_norm_cdf(v[e]);
              a[e] = 0.5f * v[e] * (2.f * cdf);                    // = 0.5 * v * (1 + erf(.)): the bits of gelu_f
              v[e] = cdf + v[e] * ssv_norm_pdf(v[e]);              // the bits of gelu_grad_f
            }
            bstore4(r_act, voff[sl][i], a);
          }
          bstore4(r_out, voff[sl][i], v);
          if constexpr (EPI == 1) {
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = gelu_f(v[e]);
            bstore4(r_act, voff[sl][i], a);
          }
        }
      };
      {
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
          __builtin_amdgcn_sched_barrier(0);       // one step of rows in flight at a time (register budget of 3 workgroups per CU)
          issue(st, 0);
          finish(st, 0);
        }
      }
    }
  };
  // (the gated / gelu' epilogues always load: one copy with the uniform test inside - two copies of THAT loop cost 224-300 B / lane of scratch)
  if constexpr (GATE != 0 || EPI == 2 || EPI == 5) rows(std::integral_constant<int, 2>{});
  else if (ADD) rows(std::integral_constant<int, 1>{});
  else rows(std::integral_constant<int, 0>{});

  if constexpr (STATS) {
#pragma unroll
    for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { st_s1[e] += __shfl_xor(st_s1[e], o, 64); st_s2[e] += __shfl_xor(st_s2[e], o, 64); }
    }
    if (r_in == 0 && cok && group_rows > 0) {
      const float n = (float)group_rows;
      *reinterpret_cast<f32x4*>(pmean + gcol) = st_p + st_s1 / n;
      *reinterpret_cast<f32x4*>(pm2 + gcol) = st_s2 - st_s1 * st_s1 / n;
    }
  }
  if constexpr (GATE != 0) {
#pragma unroll
    for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st_s1[e] += __shfl_xor(st_s1[e], o, 64); st_s2[e] += __shfl_xor(st_s2[e], o, 64);
        if constexpr (GX2) st_s3[e] += __shfl_xor(st_s3[e], o, 64);
      }
    }
    if (r_in == 0 && cok) {       // every wave writes its partial, zeros included: the finalize kernel reads all of them
      *reinterpret_cast<f32x4*>(bn->psum_g + group * ncols + gcol) = st_s1;
      *reinterpret_cast<f32x4*>(bn->psum_gx + group * ncols + gcol) = st_s2;
      if constexpr (GX2) *reinterpret_cast<f32x4*>(bn->psum_gx2 + group * ncols + gcol) = st_s3;
    }
  }
}

// =============================================================================================
// forward
// =============================================================================================
// DYF (1x1 / stride 1 / no padding only - the data gradient of a 1x1 convolution run as a forward convolution): the A operand is the
// second half of the BatchNorm backward, dx = A[k] * g + B[k] * (x - mean[k]) + D[k], formed while it is staged - the element-wise apply
// pass over (g, x) -> dx of that BatchNorm disappears (it ran AT the HBM roofline and overlapped with nothing: 23 ms of a 257 ms step).
//
// OPM 2 (SUM; same geometry - the first convolution of the NEXT residual unit): the A operand is the closing activation of the previous unit,
// a = relu(bn3(x) + shortcut) (networks/resnet.py:73-74), formed while it is staged from the unit's last conv output and its shortcut
// (a materialised tensor, or the raw projection-shortcut output with its own BatchNorm affine).  The tensor itself is still needed (next
// residual add, weight gradient, backward mask), so the workgroups of column tile 0 also store it and its ReLU byte mask: the stand-alone
// element-wise pass (2 reads + 1 write at the HBM roofline, overlapped with nothing) becomes one extra read and one write inside a convolution.
template <int BM, int BN, int WGM, int WGN, int BK, bool VEC, int EPI = 0, bool STATS = false, int C4 = 0, bool XF = false, int GATE = 0, int OPM = 0, bool ADDS2 = false, bool S2 = false, int SP = 0>
// Resident workgroups per CU the variant is compiled for: 3 (they hide each other's barriers, loads and epilogues) wherever the registers allow.
// The formed-on-load operands carry a second staged stream (ra2) and their per-channel coefficients: 188 - 236 VGPRs, i.e. 2 per CU - except
// the BatchNorm-backward operand on the 128 x 128 tile, which fits 168 with five spilled dwords in the epilogue (r03 x1: 4 - 5 % faster on the
// 28x28 layers); the two-target gate (GATE 3) needs 218.  (kernel_resources.py lists every variant; r03_experiments_step_time.txt the A/Bs.)
// SP > 0 (SSV_ARITH_BF16X3, csrc/split_bf16.h; the float4 path only; 1 = one accumulator per tile, 2 = two - see k_loop_split): the same launch on the bf16 matrix pipe - the A operand is split into three bf16 planes while it is
// staged (after its formed-on-load transform), the weights arrive pre-split (p.w_planes), the main loop is k_loop_split and the accumulators are 16 x 16 tiles; every
// epilogue is the fp32 variant's.  LDS: 192 bytes per staged row = 48 KB on the 128 x 128 tile (the fp32 image: 36.9 KB); narrow outputs (K < 128) take a 128 x 64
// tile (36 KB, half the accumulators) where the fp32 variants take 256 x 64.
__global__ void __launch_bounds__(256, SP ? ((OPM != 0 || GATE == 3 || (SP == 2 && BN == 128)) ? 2 : 3) : ((OPM == 2 || GATE == 3 || S2 || C4 == 3) ? 2 : (OPM == 1 ? ((BM == 128 && BN == 128) ? 3 : 2) : SSV_CONV_WGPC))) SSV_CONV_ATTR
conv_fwd_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
           const float* addend, float* y) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int LDT = BK + 4;                   // ROWK row stride: 16-lane b128 read groups hit 16 distinct 16-B slots
  constexpr int STAGE = (BM + BN) * LDT;
  static_assert(!XF || (VEC && C4 == 0), "the fused-input variant is the float4 path");
  constexpr bool DYF = OPM == 1, SUM = OPM == 2;
  static_assert(OPM == 0 || (VEC && C4 == 0 && !XF), "the formed-on-load operands are the float4 path");
  constexpr int EP_FLOATS = 4 * 32 * (TN * 32 + 4);     // the vectorised epilogue's staging area (one 32-row slab per wave)
  static_assert(!S2 || (VEC && C4 == 0 && !XF && OPM == 0), "the two-stage loop serves the plain float4 path");
  static_assert(C4 != 3 || (VEC && BM == 256 && BK == 32 && TM == 2 && !XF && OPM == 0 && !S2 && EP_FLOATS <= HALO_FLOATS), "the halo loader serves the 256-row tile of the plain float4 path");
  static_assert(!SP || (VEC && C4 == 0 && !S2 && BK == 32), "the bf16-piece variants are the float4 path with K-step 32");
  constexpr int SP_FLOATS = (BM + BN) * 48;             // three planes of 64-byte rows per operand
  constexpr int SMEM = SP ? (SP_FLOATS > EP_FLOATS ? SP_FLOATS : EP_FLOATS)
                          : (C4 == 3 ? HALO_FLOATS : (S2 ? 2 * STAGE : ((VEC && (EPI || STATS || GATE != 0 || OPM != 0) && EP_FLOATS > STAGE) ? EP_FLOATS : STAGE)));
  static_assert(!S2 || 2 * STAGE >= EP_FLOATS, "epilogue staging must fit the two stages");
  __shared__ __attribute__((aligned(16))) float smem[SMEM + SSV_EXP_LDS_PAD];
  __shared__ __attribute__((aligned(16))) float xfs[(XF && !SP) ? 2 * XF_MAXC : 4];    // [scale | shift] of the fused input BatchNorm (SP: read per k-tile, the LDS is the planes')
  float* As = smem;
  float* Bs = smem + BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave / WGN) * (BM / WGM), wc0 = (wave % WGN) * (BN / WGN);
  const int NT = (p.K + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  x += (size_t)blockIdx.y * p.bs_a; w += (size_t)blockIdx.y * p.bs_b; y += (size_t)blockIdx.y * p.bs_o;      // batched GEMMs (uniform; 0 otherwise)
  if (addend) addend += (size_t)blockIdx.y * p.bs_o;

  f32x16 acc[SP ? 1 : TM][SP ? 1 : TN];
  f32x4 acc16[SP ? 2 * TM : 1][SP ? 2 * TN : 1];       // SP: 16 x 16 tiles of the bf16 piece products
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else zero_acc<SP ? 1 : TM, SP ? 1 : TN>(acc);
#if SSV_EXP_WGPRIO
  wg_static_priority();
#endif
#if SSV_EXP_STAGGER
  // diagnostic builds only: the workgroups of the first resident round start SSV_EXP_STAGGER x 3.4 us apart (by their position in that round), so that the
  // workgroups sharing a CU are not in the same phase (prologue / main loop / epilogue) for the rest of the launch
  if (blockIdx.x < 256u * SSV_EXP_STAGGER_N) {
    const int ph = (int)(blockIdx.x >> 8) % SSV_EXP_STAGGER_N;
    for (int i = 0; i < ph * SSV_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  if constexpr (VEC && C4 == 2) {
    // ---- the image stem on the UNPADDED 3-channel input: for one filter row r the S taps x 3 channels of an output pixel are 3 S CONTIGUOUS
    //      floats of the image row (NHWC), so a k-tile = one filter row = 3 S (21) floats padded to BK (24) with zero WEIGHTS: the contraction
    //      is 7 x 24 = 168 columns for 147 real ones (the 4-channel form: 224 for 147, plus the padding pass over the images).  One thread stages
    //      one output row: six 16-byte loads at a 12-byte-aligned address; pixels left / right of the image row (real floats of the neighbouring
    //      row sit there) are zeroed per element, rows above / below and rows past M by the out-of-range offset. ----
    static_assert(BK == 24 && BM == 256, "one staged output row per thread, six float4 per row");
    constexpr int CHN = BK / 4;
    constexpr int BF = (BN * CHN + 255) / 256;                     // weight float4s per thread
    const rsrc_t rx = make_rsrc(x, (SSV_WHATIF & 4) ? 0u : (unsigned)p.N * p.H * p.W * 12u);
    const rsrc_t rw = make_rsrc(w, (unsigned)p.K * p.R * BK * 4u);
    int hi0, rowbase;
    unsigned wmask = 0;                                            // bit px: tap column px of this output pixel lies inside the image row
    {
      const int m = m0 + tid;
      const bool ok = m < p.M;
      const uint32_t mm = ok ? (uint32_t)m : 0u;
      const uint32_t n = fdiv(mm, p.dHoWo);
      const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
      const uint32_t ho = fdiv(rem, p.dWo);
      const uint32_t wo = rem - ho * (uint32_t)p.Wo;
      hi0 = ok ? (int)ho * p.stride - p.pad : -(1 << 20);
      const int wi0 = (int)wo * p.stride - p.pad;
      rowbase = ok ? (((int)n * p.H + hi0) * p.W + wi0) * 12 : 0;
#pragma unroll
      for (int px = 0; px < 8; ++px) wmask |= (unsigned)((px < p.S) & ((unsigned)(wi0 + px) < (unsigned)p.W)) << px;
    }
    int boff[BF];
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      const int f = tid + 256 * i, kf = f / CHN, ch = f - kf * CHN;
      boff[i] = (f < BN * CHN && n0 + kf < p.K) ? ((n0 + kf) * p.R * BK + ch * 4) * 4 : OOB_OFF;
    }
    int lr = 0;
    f32x4 ra[CHN], rb[BF];
    const int xbytes = p.N * p.H * p.W * 12;
    int rots = 0;                                                  // 4 bits per float4: rot + 8 where the load was moved (straddle_off), else 0
    auto load_tile = [&]() {
      const bool hv = (unsigned)(hi0 + lr) < (unsigned)p.H;        // filter rows past R read zero weights (offset past the weight tensor's rows is OOB)
      const int base = rowbase + lr * p.W * 12;
      rots = 0;
#pragma unroll
      for (int j = 0; j < CHN; ++j) {
        int rot;
        ra[j] = bload4(rx, straddle_off((hv && lr < p.R) ? base + j * 16 : OOB_OFF, xbytes, rot), 0);
        rots |= (rot != 0 ? rot + 8 : 0) << (4 * j);
      }
#pragma unroll
      for (int i = 0; i < BF; ++i) rb[i] = bload4(rw, (lr < p.R) ? boff[i] : OOB_OFF, lr * BK * 4);
      lr += 1;
    };
    auto xform_tile = [&]() {
      if (rots != 0) {                                             // the first / last pixels of the whole batch only
#pragma unroll
        for (int j = 0; j < CHN; ++j) { const int q = (rots >> (4 * j)) & 15; if (q) ra[j] = straddle_fix(ra[j], q - 8); }
      }
#pragma unroll
      for (int j = 0; j < CHN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[j][e] = ((wmask >> ((4 * j + e) / 3)) & 1u) ? ra[j][e] : 0.f;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
      for (int j = 0; j < CHN; ++j) *reinterpret_cast<f32x4*>(&As[tid * LDT + 4 * j]) = ra[j];
#pragma unroll
      for (int i = 0; i < BF; ++i) {
        const int f = tid + 256 * i, kf = f / CHN, ch = f - kf * CHN;
        if (f < BN * CHN) *reinterpret_cast<f32x4*>(&Bs[kf * LDT + ch * 4]) = rb[i];
      }
    };
    k_loop<TM, TN, true, true, LDT, LDT, BK, CHN + BF>(p.R, As, Bs, wr0, wc0, lane, acc, load_tile, store_tile, xform_tile);
  } else if constexpr (VEC && C4 == 1) {
    // ---- C == 4 (the image stem with its 3 channels padded to 4): one 16-byte load = one filter TAP of one pixel, a k-tile = BK/4
    //      consecutive taps.  Lane group tl owns tap kt*CH + tl of the tile: (r, s) by one fast division, bounds per staged row. ----
    constexpr int CH = BK / 4, RPP = 256 / CH;
    constexpr int AP = BM / RPP, BP = BN / RPP;
    const int tl = tid % CH, rsub = tid / CH;
    const rsrc_t rx = make_rsrc(x, (unsigned)p.N * p.H * p.W * 16u);
    const rsrc_t rw = make_rsrc(w, (unsigned)p.K * p.RSC * 4u);
    int hi0[AP], wi0[AP], aoff[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + rsub + RPP * i;
      const bool ok = m < p.M;
      const uint32_t mm = ok ? (uint32_t)m : 0u;
      const uint32_t n = fdiv(mm, p.dHoWo);
      const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
      const uint32_t ho = fdiv(rem, p.dWo);
      const uint32_t wo = rem - ho * (uint32_t)p.Wo;
      hi0[i] = ok ? (int)ho * p.stride - p.pad : -(1 << 20);
      wi0[i] = (int)wo * p.stride - p.pad;
      aoff[i] = (((int)n * p.H + hi0[i]) * p.W + wi0[i]) * 16;
    }
    int boff[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int ko = n0 + rsub + RPP * i;
      boff[i] = ko < p.K ? ko * p.RSC * 4 : OOB_OFF;
    }
    const int ntap = p.R * p.S;
    int lt = tl;                                               // this lane group's tap in the current tile
    f32x4 ra[AP], rb[BP];
    auto load_tile = [&]() {
      const bool tv = lt < ntap;
      const int r = (int)fdiv((uint32_t)(tv ? lt : 0), p.dS);
      const int sx = (tv ? lt : 0) - r * p.S;
      const int toff_x = (r * p.W + sx) * 16, toff_w = lt * 16;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = tv & ((unsigned)(hi0[i] + r) < (unsigned)p.H) & ((unsigned)(wi0[i] + sx) < (unsigned)p.W);
        ra[i] = bload4(rx, ok ? aoff[i] + toff_x : OOB_OFF, 0);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = bload4(rw, (tv && boff[i] != OOB_OFF) ? boff[i] + toff_w : OOB_OFF, 0);
      lt += CH;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
      for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[(rsub + RPP * i) * LDT + tl * 4]) = ra[i];
#pragma unroll
      for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[(rsub + RPP * i) * LDT + tl * 4]) = rb[i];
    };
    k_loop<TM, TN, true, true, LDT, LDT, BK, AP + BP>((ntap + CH - 1) / CH, As, Bs, wr0, wc0, lane, acc, load_tile, store_tile);
  } else if constexpr (VEC && C4 == 3) {
    // ---- 3x3 / stride 1 / padding 1 with C % 32 == 0 on the 256 x 64 tile (networks/resnet.py:7-10,56-58: conv2 of the 64-channel units; its data gradient is the
    //      same product with the rotated filter): HALO staging.  The generic loader below stages every pixel NINE times per channel chunk (once per tap: 18 k-tiles
    //      of predicated loads, LDS writes and two barriers each for 64 channels); here the pixels the tile's 256 outputs can see - their image rows plus one above
    //      and below, a zero record left and right of each row - go to LDS ONCE per 16-channel chunk as 64-byte records, and the nine taps are nine record offsets
    //      of the same staged data.  Two LDS stages: the next chunk's global loads are issued before the current chunk's 288 MFMAs per wave and written to the
    //      other stage after them - ONE barrier per chunk, none inside it, and the loads of all resident workgroups spread over the matrix phase instead of
    //      arriving as one burst (one stage of 32 channels, measured: the chip's workgroups run in lockstep, so their staging bursts and their MFMA phases
    //      alternated instead of overlapping - 107 TFLOP/s, the generic loader's figure).  A lane's A fragment is the 8 consecutive channels 8 * half .. of its
    //      pixel's record (two ds_read_b128 at immediate offsets of one address register per row tap), its B fragment the same 8 channels of its output column's
    //      filter tap, read straight from the (L2-resident) filter into registers - the weights never touch LDS; both are fetched one tap (32 MFMAs) ahead,
    //      one memory instruction per MFMA.  Rows of a neighbouring image that sit above / below an image's first / last row are real data in LDS (their own
    //      pixels need them): a lane whose pixel's tap row falls outside its image reads the records of zeros instead. ----
    const int WP = p.W + 2;
    const rsrc_t rx = make_rsrc(x, (SSV_WHATIF & 4) ? 0u : (unsigned)p.N * p.H * p.W * p.C * 4u);
    const rsrc_t rw = make_rsrc(w, (unsigned)p.K * p.RSC * 4u);
    const int l31 = lane & 31, half = lane >> 5;
    const int g0 = (int)fdiv((uint32_t)m0, p.dWo);                          // running row index n * H + ho of the tile's first pixel (uniform)
    const int nrec = ((int)fdiv((uint32_t)(min(m0 + BM, p.M) - 1), p.dWo) - g0 + 3) * WP;      // staged records (<= HALO_REC: checked by the launcher)
    // per lane and row tap dr: LDS float offset (within a stage) of its pixels' fragment in the centre column, or of the middle zero record where the row is
    // outside the image; the stage, the column tap and the 16-byte piece are immediate offsets of the ds_read
    int abase[TM][3];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const uint32_t m = (uint32_t)min(m0 + wr0 + tm * 32 + l31, p.M - 1);  // rows past M repeat the last pixel: never stored nor counted
      const uint32_t g = fdiv(m, p.dWo), n = fdiv(m, p.dHoWo);
      const int hrow = (int)(g - n * (uint32_t)p.H);
      const int P0 = ((int)g - g0 + 1) * WP + (int)(m - g * (uint32_t)p.W) + 1;
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr)
        abase[tm][dr + 1] = (((unsigned)(hrow + dr) < (unsigned)p.H) ? P0 + dr * WP : HALO_REC + 1) * HALO_RS + 8 * half;
    }
    int wofs[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int ko = n0 + wc0 + tn * 32 + l31;
      wofs[tn] = ko < p.K ? (ko * p.RSC + 8 * half) * 4 : OOB_OFF;
    }
    constexpr int NLD = (HALO_REC * 4 + 255) / 256;
    int soff[NLD];                                                          // byte offset of this thread's staged float4s in x (channel chunk 0)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i, rec = e >> 2, slot = e & 3;
      const int j = (int)fdiv((uint32_t)rec, p.dWp), col = rec - j * WP, g = g0 - 1 + j;
      const bool ok = (rec < nrec) & (col >= 1) & (col <= p.W) & ((unsigned)g < (unsigned)(p.N * p.H));
      soff[i] = ok ? ((g * p.W + col - 1) * p.C + slot * 4) * 4 : OOB_OFF;
    }
    f32x4 st[NLD];
    auto issue_stage = [&](int c0) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) st[i] = bload4(rx, soff[i], c0 * 4);
    };
    auto write_stage = [&](int stage) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i, rec = e >> 2, slot = e & 3;
        if (rec < nrec) *reinterpret_cast<f32x4*>(&smem[stage * HALO_STAGE + rec * HALO_RS + slot * 4]) = st[i];
      }
    };
    f32x4 bq[2][TN][2], aq[2][TM][2];
    auto load_b = [&](f32x4 (&b)[TN][2], int tap, int c0) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[tn][q] = bload4(rw, ((SSV_WHATIF & 8) ? 0 : wofs[tn]) + q * 16, (tap * p.C + c0) * 4);      // (OOB_OFF + 16 is still out of range)
    };
    auto load_a = [&](f32x4 (&a)[TM][2], int tap, int stage) {
      const int dr = tap / 3, dc = tap % 3 - 1;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int q = 0; q < 2; ++q) a[tm][q] = *reinterpret_cast<const f32x4*>(&smem[abase[tm][dr] + stage * HALO_STAGE + dc * HALO_RS + 4 * q]);
    };
    issue_stage(0);
    load_b(bq[0], 0, 0);
    for (int i = tid; i < 2 * 3 * HALO_RS; i += 256) smem[(i / (3 * HALO_RS)) * HALO_STAGE + HALO_REC * HALO_RS + i % (3 * HALO_RS)] = 0.f;
    write_stage(0);
    __syncthreads();
    const int nch = p.C / 16;                                               // even: C % 32 == 0
#pragma unroll 1
    for (int ch2 = 0; ch2 < nch; ch2 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                                         // chunk ch2 + u lives in stage u
        const int c0 = (ch2 + u) * 16;
        const bool more = ch2 + u + 1 < nch;
        if (more) issue_stage(c0 + 16);
        load_a(aq[(u * 9) & 1], 0, u);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          constexpr int NMF = 8 * TM * TN;                                  // MFMAs per tap
          const int cur = (u * 9 + tap) & 1;
          __builtin_amdgcn_sched_barrier(0);
          if (tap < 8) { load_b(bq[cur ^ 1], tap + 1, c0); load_a(aq[cur ^ 1], tap + 1, u); }
          else if (more) load_b(bq[cur ^ 1], 0, c0 + 16);                   // the next chunk's first filter tap (its A fragments wait for the stage)
#pragma unroll
          for (int ks = 0; ks < ((SSV_WHATIF & 1) ? 2 : 8); ++ks)          // (timing what-if: a quarter of the MFMAs, every fragment load still consumed)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][tm][(SSV_WHATIF & 1) ? ks : (ks >> 2)][ks & 3], bq[cur][tn][(SSV_WHATIF & 1) ? ks : (ks >> 2)][ks & 3], acc[tm][tn], 0, 0, 0);
          if (tap < 8 && !(SSV_WHATIF & 1)) {
#pragma unroll
            for (int i_ = 0; i_ < 2 * TN; ++i_) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
#pragma unroll
            for (int i_ = 0; i_ < 2 * TM; ++i_) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x8, NMF - 2 * TN - 2 * TM, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) write_stage(u ^ 1);                                       // its last readers passed the previous chunk's barrier
        __syncthreads();
      }
    }
  } else if constexpr (VEC) {
    // ---- C % BK == 0: every k-tile lies inside one filter tap; float4 staging, BK/4 lanes per row ----
    constexpr int CH = BK / 4, RPP = 256 / CH;
    constexpr int AP = BM / RPP, BP = BN / RPP;
    const int chunk = (tid % CH) * 4, rsub = tid / CH;
    // Loader state per staged row: byte offset of (n, hi0, wi0, chunk) in x, top-left input coordinate.  Per tile the
    // tap adds ONE uniform offset; invalid rows / taps get the out-of-range offset and come back as zeros.
    const rsrc_t rx = make_rsrc(x, (SSV_WHATIF & 4) ? 0u : (unsigned)p.N * p.H * p.W * p.C * 4u);
    const rsrc_t rw = make_rsrc(w, (unsigned)p.K * p.RSC * 4u);
    int hi0[AP], wi0[AP], aoff[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + rsub + RPP * i;
      const bool ok = m < p.M;
      const uint32_t mm = ok ? (uint32_t)m : 0u;
      const uint32_t n = fdiv(mm, p.dHoWo);
      const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
      const uint32_t ho = fdiv(rem, p.dWo);
      const uint32_t wo = rem - ho * (uint32_t)p.Wo;
      hi0[i] = ok ? (int)ho * p.stride - p.pad : -(1 << 20);         // a row past M never passes the bounds test
      wi0[i] = (int)wo * p.stride - p.pad;
      aoff[i] = ((((int)n * p.H + hi0[i]) * p.W + wi0[i]) * p.C + chunk) * 4;
      if (!ok) aoff[i] = OOB_OFF;
    }
    int boff[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int ko = n0 + rsub + RPP * i;
      boff[i] = ko < p.K ? (ko * p.RSC + chunk) * 4 : OOB_OFF;
    }
    const bool inb = p.pad == 0 && p.R == 1 && p.S == 1;     // 1x1 / no padding: every tap of a valid row is in bounds
    int lr = 0, ls = 0, lc0 = 0;   // loader position (tap r, s, first channel)
    int ktiles = p.RSC / BK;
    if constexpr (!XF && OPM == 0) {
      if (p.Kg != 0) {             // block-diagonal filter bank (uniform): only the channels of this column tile's groups
        int c_hi;
        group_span(n0, min(n0 + BN, p.K), p.Kg, p.Cg, BK, p.C, lc0, c_hi);
        ktiles = p.R * p.S * ((c_hi - lc0) / BK);
      }
    }
    f32x4 ra[AP], rb[SP ? 1 : BP];
    f32x4 ra2[OPM ? AP : 1], co[(OPM || (XF && SP)) ? 4 : 1];
    // SP: the weight tile arrives as three bf16 planes, 16-byte pieces (8 k of one filter row) straight to LDS: BN * 4 pieces per plane and k-tile
    constexpr int BPP = SP ? (BN * 4 + 255) / 256 : 1;
    u32x4 rbp[SP ? 3 : 1][BPP];
    int boffp[BPP], bldsp[BPP];
    const long long wplane = p.wp_stride;                               // elements per plane (all batch elements)
    const rsrc_t rwp = make_rsrc(SP ? (const void*)(p.w_planes + (size_t)blockIdx.y * p.bs_b) : (const void*)w, SP ? (unsigned)((2 * wplane + (long long)p.K * p.RSC) * 2) : 16u);
    if constexpr (SP) {
#pragma unroll
      for (int i = 0; i < BPP; ++i) {
        const int idx = tid + 256 * i, br = idx >> 2, k8 = idx & 3;
        boffp[i] = (idx < BN * 4 && n0 + br < p.K) ? ((n0 + br) * p.RSC + k8 * 8) * 2 : OOB_OFF;
        bldsp[i] = splitbf::rowk_off(br, k8);
      }
    }
    const rsrc_t rxs0 = make_rsrc((XF && SP) ? p.xf_scale : x, (XF && SP) ? (unsigned)p.C * 4u : 16u), rxs1 = make_rsrc((XF && SP) ? p.xf_shift : x, (XF && SP) ? (unsigned)p.C * 4u : 16u);
    const rsrc_t rx2 = make_rsrc(DYF ? p.dyin_x : (SUM ? p.sum_res : x), (unsigned)p.N * p.H * p.W * p.C * 4u);
    const rsrc_t rco = make_rsrc(DYF ? p.dyin_coef : x, DYF ? (unsigned)p.C * 16u : 16u);
    const rsrc_t rs0 = make_rsrc(SUM ? p.sum_scale : x, (unsigned)p.C * 4u), rs1 = make_rsrc(SUM ? p.sum_shift : x, (unsigned)p.C * 4u);
    const bool raff = SUM && p.sum_rscale != nullptr;           // uniform: the shortcut carries its own BatchNorm affine
    const rsrc_t rs2 = make_rsrc(raff ? p.sum_rscale : x, (unsigned)p.C * 4u), rs3 = make_rsrc(raff ? p.sum_rshift : x, (unsigned)p.C * 4u);
    const rsrc_t rso = make_rsrc(SUM ? p.sum_out : y, (unsigned)p.N * p.H * p.W * p.C * 4u);
    const rsrc_t rsm = make_rsrc(SUM && p.sum_mask ? reinterpret_cast<float*>(p.sum_mask) : y, (unsigned)p.N * p.H * p.W * (p.C / 4));
    int st_off = 0;                // SUM: uniform byte offset of the tile being transformed (the loader has moved on by then)
    int xf_ok = 0, xf_c = 0;       // XF: which of the staged rows hold real pixels (bit i), first channel of this thread's float4
    if constexpr (XF && !SP) {
      for (int c = tid; c < p.C; c += 256) { xfs[c] = p.xf_scale[c]; xfs[XF_MAXC + c] = p.xf_shift[c]; }
      __syncthreads();
    }
    auto load_tile = [&]() {
      const int toff_x = ((lr * p.W + ls) * p.C + lc0) * 4;    // uniform
      const int toff_w = ((lr * p.S + ls) * p.C + lc0) * 4;
      if constexpr (XF) { xf_ok = 0; xf_c = lc0 + chunk; }
      if constexpr (XF && SP) { co[0] = bload4(rxs0, chunk * 4, lc0 * 4); co[1] = bload4(rxs1, chunk * 4, lc0 * 4); }
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = inb | (((unsigned)(hi0[i] + lr) < (unsigned)p.H) & ((unsigned)(wi0[i] + ls) < (unsigned)p.W));
        ra[i] = bload4(rx, ok ? aoff[i] + toff_x : OOB_OFF, 0);
        if constexpr (OPM != 0) ra2[i] = bload4(rx2, ok ? aoff[i] + toff_x : OOB_OFF, 0);
        if constexpr (XF) xf_ok |= (int)ok << i;
      }
      if constexpr (DYF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) co[j] = bload4(rco, (j * p.C + chunk) * 4, lc0 * 4);
      }
      if constexpr (SUM) {
        st_off = toff_x;
        co[0] = bload4(rs0, chunk * 4, lc0 * 4); co[1] = bload4(rs1, chunk * 4, lc0 * 4);
        if (raff) { co[2] = bload4(rs2, chunk * 4, lc0 * 4); co[3] = bload4(rs3, chunk * 4, lc0 * 4); }
        else { co[2] = f32x4{1.f, 1.f, 1.f, 1.f}; co[3] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      if constexpr (SP) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < BPP; ++i)
            rbp[q][i] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rwp, boffp[i] == OOB_OFF ? OOB_OFF : boffp[i] + q * (int)(wplane * 2), toff_w >> 1, 0);
      } else {
#pragma unroll
        for (int i = 0; i < BP; ++i) rb[i] = bload4(rw, boff[i], toff_w);
      }
      // k order = (channel chunk, r, s) with the TAPS innermost: consecutive k-tiles re-read the same pixels' 128-byte channel
      // slice shifted by one tap, so the re-reads hit in L2 (with the channel chunk innermost every tap came back from HBM / the
      // infinity cache: 8-10x the input bytes on the 3x3 layers).  Straight-line advance (selects, no branches) keeps the k-loop
      // body a single scheduling region.
      ls += 1;
      const bool w1 = ls == p.S;
      ls = w1 ? 0 : ls;
      lr += w1 ? 1 : 0;
      const bool w2 = lr == p.R;
      lr = w2 ? 0 : lr;
      lc0 += w2 ? BK : 0;
    };
    auto xform_tile = [&]() {
      if constexpr (XF) {
        // the staged values are the producer's raw conv output: apply its BatchNorm + ReLU here (same fmaf / fmaxf as bn_apply_k,
        // so the operand is bit-identical to the materialised activation); padding taps stay exactly zero.  A row past M keeps
        // relu(shift) - harmless, its output row is never stored nor counted.
        f32x4 sc, sh;
        if constexpr (SP) { sc = co[0]; sh = co[1]; }
        else { sc = *reinterpret_cast<const f32x4*>(&xfs[xf_c]); sh = *reinterpret_cast<const f32x4*>(&xfs[XF_MAXC + xf_c]); }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          // ReLU and "a padding tap stays zero" in ONE instruction per element: clamp to [0, cap] with cap = +inf for a real pixel and
          // 0 for a padding tap (v_med3_f32) - the staged transform costs 2 VALU per element instead of 3
          const float cap = (inb || ((xf_ok >> i) & 1)) ? __builtin_inff() : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(__builtin_fmaf(ra[i][e], sc[e], sh[e]), 0.f, cap);
        }
      } else if constexpr (DYF) {
        // a row past M holds D - mean * B instead of zero: harmless, its output row is never stored nor counted
#pragma unroll
        for (int i = 0; i < AP; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_fmaf(ra[i][e], co[0][e], __builtin_fmaf(ra2[i][e] - co[1][e], co[2][e], co[3][e]));
        }
      } else if constexpr (SUM) {
        // the arithmetic of bn_apply_k (fma(x, scale, shift) + [fma(res, rscale, rshift) | res], mask bit = v > 0, fmaxf): bit-identical to the
        // materialised activation.  A row past M stages relu(shift + rshift) - harmless, see above - and its stores are dropped (out of range).
        const bool wr = nt == 0;                               // uniform: every element of a is staged by exactly one workgroup of column tile 0
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          f32x4 v;
          unsigned bits = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = __builtin_fmaf(ra[i][e], co[0][e], co[1][e]) + __builtin_fmaf(ra2[i][e], co[2][e], co[3][e]);
            bits |= (v[e] > 0.f ? 1u : 0u) << e;
            v[e] = fmaxf(v[e], 0.f);
          }
          ra[i] = v;
          if (wr) {
            const int off = aoff[i] == OOB_OFF ? OOB_OFF : aoff[i] + st_off;
            bstore4(rso, off, v);
            if (p.sum_mask) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, rsm, off == OOB_OFF ? OOB_OFF : off >> 4, 0, 0);
          }
        }
      }
    };
    constexpr int A_PLANE = BM * splitbf::ROWB, B_PLANE = BN * splitbf::ROWB;
    unsigned char* const Asb = reinterpret_cast<unsigned char*>(smem);
    unsigned char* const Bsb = Asb + 3 * A_PLANE;
    auto store_tile = [&](int buf) {
      if constexpr (SP) {
#pragma unroll
        for (int i = 0; i < AP; ++i) splitbf::rowk_store(Asb, A_PLANE, rsub + RPP * i, tid % CH, ra[i]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < BPP; ++i)
            if (BN * 4 % 256 == 0 || tid + 256 * i < BN * 4) *reinterpret_cast<u32x4*>(Bsb + q * B_PLANE + bldsp[i]) = rbp[q][i];
        return;
      }
      const int so = S2 ? buf * STAGE : 0;
#pragma unroll
      for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[so + (rsub + RPP * i) * LDT + chunk]) = ra[i];
#pragma unroll
      for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[so + (rsub + RPP * i) * LDT + chunk]) = rb[i];
    };
    if constexpr (SP) k_loop_split<2 * TM, 2 * TN, false, false, 128, 128, 0, 1, SP == 2>(ktiles, Asb, A_PLANE, Bsb, B_PLANE, wr0, wc0, lane, acc16, load_tile, store_tile, xform_tile);
    else if constexpr (S2) k_loop2<TM, TN, LDT, BK, AP + BP, STAGE>(ktiles, As, Bs, wr0, wc0, lane, acc, load_tile, store_tile);
    else k_loop<TM, TN, true, true, LDT, LDT, BK, AP + BP + (OPM ? AP + 4 : 0)>(ktiles, As, Bs, wr0, wc0, lane, acc, load_tile, store_tile, xform_tile);
  } else {
    // ---- generic gather (any C; used by the 3-channel stem): scalar staging, k -> (r,s,c) per element ----
    static_assert(BK == GBK, "generic path is BK=16");
    constexpr int AE = BM / 16, BE = BN / 16;
    const int kk = tid & 15, rsub = tid >> 4;
    int hw0[AE];        // packed (hi0 & 0xffff) | (wi0 << 16)
    int nimg[AE];
#pragma unroll
    for (int i = 0; i < AE; ++i) {
      const int m = m0 + rsub + 16 * i;
      if (m < p.M) {
        const uint32_t n = fdiv((uint32_t)m, p.dHoWo);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Ho * p.Wo);
        const uint32_t ho = fdiv(rem, p.dWo);
        const uint32_t wo = rem - ho * (uint32_t)p.Wo;
        const int h0 = (int)ho * p.stride - p.pad, w0 = (int)wo * p.stride - p.pad;
        hw0[i] = (h0 & 0xffff) | (w0 << 16);
        nimg[i] = (int)n;
      } else { hw0[i] = 0; nimg[i] = -1; }
    }
    float ra[AE], rb[BE];
    int lk = kk;
    auto load_tile = [&]() {
      const bool kok = lk < p.RSC;
      const uint32_t tap = fdiv((uint32_t)(kok ? lk : 0), p.dC);
      const int c = (kok ? lk : 0) - (int)tap * p.C;
      const uint32_t r = fdiv(tap, p.dS);
      const int s = (int)tap - (int)r * p.S;
#pragma unroll
      for (int i = 0; i < AE; ++i) {
        const int hi = (int)(short)(hw0[i] & 0xffff) + (int)r, wi = (hw0[i] >> 16) + s;
        float v = 0.f;
        if (kok && nimg[i] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          v = x[(((size_t)nimg[i] * p.H + hi) * p.W + wi) * p.C + c];
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BE; ++i) {
        const int ko = n0 + rsub + 16 * i;
        rb[i] = (kok && ko < p.K) ? w[(size_t)ko * p.RSC + lk] : 0.f;
      }
      lk += BK;
    };
    auto store_tile = [&]() {
#pragma unroll
      for (int i = 0; i < AE; ++i) As[(rsub + 16 * i) * LDT + kk] = ra[i];
#pragma unroll
      for (int i = 0; i < BE; ++i) Bs[(rsub + 16 * i) * LDT + kk] = rb[i];
    };
    const int nkt = (p.RSC + BK - 1) / BK;
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;
      if (more) load_tile();
      mma_ktile<TM, TN, true, true, LDT, LDT, BK>(As, Bs, wr0, wc0, lane, acc);
      __syncthreads();
      if (more) { store_tile(); __syncthreads(); }
    }
  }

  // ---- epilogue: acc reg j of lane l is (row (j&3)+8*(j>>2)+4*(l>>5), col l&31) of its 32x32 tile ----
  static_assert(!(EPI || STATS) || (VEC && EP_FLOATS <= SMEM), "the fused-activation / statistics forward needs the vectorised epilogue");
  if constexpr (VEC && EP_FLOATS <= SMEM) {
    if ((p.K & 3) == 0) {             // 16-byte stores need K % 4 == 0 (uniform): whole rows segments through LDS
      const int rbase = m0 + wr0;
      if constexpr (STATS) {
        // this wave's rows rbase .. rbase + TM*32 are group g of the statistics partials (rows-per-group = TM*32 = 64)
        constexpr int GR = TM * 32;
        const int g = rbase / GR;
        const int valid = min(GR, p.M - rbase);
        epilogue_vec<TM, TN, 0, true>(sel_acc<(SP != 0)>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.K, bias, addend, y, (long long)p.M * p.K,
                                          [&](int r) -> long long { const int m = rbase + r; return m < p.M ? (long long)m * p.K : -1; },
                                          nullptr, nullptr, p.aux_out + (size_t)g * p.K, p.aux_out2 + (size_t)g * p.K, valid > 0 ? valid : 0);
      } else if constexpr (GATE != 0) {
        static_assert(BM / WGM == 64, "gate partials are per 64 output rows");
        auto row_off = [&](int r) -> long long { const int m = rbase + r; return m < p.M ? (long long)m * p.K : -1; };
        if constexpr (ADDS2) {
          static_assert(GATE != 0, "the compact stride-2 addend rides on the gated epilogues");
          auto add_off = [&](int r) -> long long {            // output row -> row of the compact addend, or -1 (odd h or w: no contribution)
            const int m = rbase + r;
            if (m >= p.M) return -1;
            const uint32_t n = fdiv((uint32_t)m, p.dHoWo);
            const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Ho * p.Wo);
            const uint32_t ho = fdiv(rem, p.dWo);
            const uint32_t wo = rem - ho * (uint32_t)p.Wo;
            if ((ho | wo) & 1u) return -1;
            return ((long long)((int)n * p.add_H2 + (int)(ho >> 1)) * p.add_W2 + (int)(wo >> 1)) * p.K;
          };
          epilogue_vec<TM, TN, 0, false, GATE>(sel_acc<(SP != 0)>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.K, bias, addend, y, (long long)p.M * p.K, row_off,
                                                   nullptr, nullptr, nullptr, nullptr, 0, &p.gate, (long long)(rbase / 64), add_off,
                                                   (long long)p.N * p.add_H2 * p.add_W2 * p.K);
        } else {
          epilogue_vec<TM, TN, 0, false, GATE>(sel_acc<(SP != 0)>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.K, bias, addend, y, (long long)p.M * p.K, row_off,
                                                   nullptr, nullptr, nullptr, nullptr, 0, &p.gate, (long long)(rbase / 64));
        }
      } else {
        epilogue_vec<TM, TN, EPI>(sel_acc<(SP != 0)>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.K, bias, addend, y, (long long)p.M * p.K,
                                  [&](int r) -> long long { const int m = rbase + r; return m < p.M ? (long long)m * p.K : -1; },
                                  (EPI == 1 || EPI == 4) ? p.aux_out : nullptr, (EPI == 2 || EPI == 5) ? p.aux_in : nullptr);
      }
      return;
    }
  }
  static_assert(GATE == 0 || (VEC && EP_FLOATS <= SMEM), "the gated epilogue is the vectorised one");
  if constexpr (SP) return;            // K % 4 == 0 is a launch precondition of the bf16-piece variants
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wc0 + tn * 32 + l31;
    if (col >= p.K) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = m0 + wr0 + tm * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
        if (row < p.M) {
          const size_t o = (size_t)row * p.K + col;
          float v = acc[tm][tn][j] + bv;
          if (addend) v += addend[o];
          y[o] = v;
        }
      }
    }
  }
}

// =============================================================================================
// dgrad: rows are the input pixels of ONE stride-parity class (blockIdx.y); only taps with
// (ph + pad - r) % stride == 0 contribute to that class, with ho = hq + (ph + pad - r)/stride.
// =============================================================================================
// SP (SSV_ARITH_BF16X3; the 128 x 128 tile with K-step 32): dY rows are split while they are staged (ROWK planes), the weights arrive pre-split (p.w_planes) and keep their
// k-major image ([32 output channels][input channels] per plane), read by transposed reads.
template <int BM, int BN, int WGM, int WGN, int BK, bool EPI = false, int GATE = 0, bool SP = false>
__global__ void __launch_bounds__(256, SP ? 2 : SSV_CONV_WGPC) SSV_CONV_ATTR      // 3 workgroups per CU: they hide each other's barriers, loads and epilogues
conv_dgrad_k(ConvKP p, const float* __restrict__ dy, const float* __restrict__ w, const float* addend, float* dx) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int LDT = BK + 4;
  static_assert(!SP || (BK == 32 && BN == 128), "the bf16-piece data gradient: K-step 32, 128 input channels per tile");
  constexpr int A_FLOATS = BM * LDT, B_FLOATS = BK * BN, STAGE = SP ? 48 * (BM + BN) : A_FLOATS + B_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[STAGE + SSV_EXP_LDS_PAD];
  __shared__ unsigned rowpix[BM];
  __shared__ int taps[64 * 3];      // (dho, dwo, tapoff) per valid tap
  __shared__ int ntaps_s;
  float* As = smem;
  float* Bs = smem + A_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave / WGN) * (BM / WGM), wc0 = (wave % WGN) * (BN / WGN);
  const int st = p.stride;
  const int ph = blockIdx.y / st, pw = blockIdx.y - ph * st;
  const int Hq = ph < p.H ? (p.H - ph + st - 1) / st : 0;
  const int Wq = pw < p.W ? (p.W - pw + st - 1) / st : 0;
  const int Mc = p.N * Hq * Wq;
  const int NT = (p.C + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= Mc) {
    if constexpr (GATE != 0) {       // a tile past this class's rows: its gate partials must still read as zeros
      const long long grp = ((long long)blockIdx.y * (gridDim.x / NT) + mt) * (BM / 64) + (wr0 / 64);
      const int col = n0 + wc0 + lane;                          // BN / WGN columns per wave (64 or 128): lanes stride 64
      for (int c = col; c < min(n0 + wc0 + BN / WGN, p.C); c += 64) { p.gate.psum_g[grp * p.C + c] = 0.f; p.gate.psum_gx[grp * p.C + c] = 0.f; }
    }
    return;
  }

  if (tid == 0) {
    int n = 0;
    for (int r = 0; r < p.R; ++r) {
      const int dr = ph + p.pad - r;
      if (((dr % st) + st) % st != 0) continue;
      for (int s = 0; s < p.S; ++s) {
        const int ds = pw + p.pad - s;
        if (((ds % st) + st) % st != 0) continue;
        // exact division also for negative multiples of st
        taps[3 * n + 0] = dr >= 0 ? dr / st : -((-dr) / st);
        taps[3 * n + 1] = ds >= 0 ? ds / st : -((-ds) / st);
        taps[3 * n + 2] = (r * p.S + s) * p.C;
        ++n;
      }
    }
    ntaps_s = n;
  }
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    unsigned pix = 0xffffffffu;
    if (m < Mc) {
      const int n = m / (Hq * Wq);
      const int rem = m - n * Hq * Wq;
      const int hq = rem / Wq, wq = rem - hq * Wq;
      pix = (unsigned)((n * p.H + hq * st + ph) * p.W + wq * st + pw);
    }
    rowpix[r] = pix;
  }
  __syncthreads();
  const int ntaps = ntaps_s;

  f32x16 acc[SP ? 1 : TM][SP ? 1 : TN];
  f32x4 acc16[SP ? 2 * TM : 1][SP ? 2 * TN : 1];
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else zero_acc<SP ? 1 : TM, SP ? 1 : TN>(acc);

  constexpr int CH = BK / 4, RPP = 256 / CH;
  constexpr int AP = BM / RPP;           // A passes: RPP rows x CH float4 per pass
  constexpr int BCV = BN / 4;            // float4 per B row
  constexpr int BRP = 256 / BCV;         // B rows per pass
  constexpr int BP = BK / BRP;           // B passes
  const int chunk = (tid % CH) * 4, rsub = tid / CH;
  // buffer-load loader (see the forward kernel): byte offset of (n, hq, wq, chunk) in dY per staged row, one uniform
  // offset per (tap, k-tile); rows outside the class / taps outside dY read as zeros
  const rsrc_t rdy = make_rsrc(dy, (unsigned)p.N * p.Ho * p.Wo * p.K * 4u);
  const rsrc_t rw = make_rsrc(w, (unsigned)p.K * p.RSC * 4u);
  int hq_[AP], wq_[AP], aoff[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + rsub + RPP * i;
    const bool ok = m < Mc;
    const int mm = ok ? m : 0;
    const int n = mm / (Hq * Wq);
    const int rem = mm - n * Hq * Wq;
    hq_[i] = ok ? rem / Wq : -(1 << 20);
    wq_[i] = rem - (rem / Wq) * Wq;
    aoff[i] = ok ? (((n * p.Ho + hq_[i]) * p.Wo + wq_[i]) * p.K + chunk) * 4 : OOB_OFF;
  }
  const int bcol = (tid % BCV) * 4, brow = tid / BCV;
  const int boff = n0 + bcol < p.C ? (brow * p.RSC + n0 + bcol) * 4 : OOB_OFF;
  const bool inb = p.pad == 0 && p.R == 1 && p.S == 1;       // 1x1: the single tap maps every class pixel onto a valid dY pixel

  int ti = 0, lk0 = 0;                   // loader position: tap index, first output channel
  int kspan = p.K;
  if (p.Cg != 0) {                       // block-diagonal filter bank (uniform): only the output channels of this column tile's groups
    int k_hi;
    group_span(n0, min(n0 + BN, p.C), p.Cg, p.Kg, BK, p.K, lk0, k_hi);
    kspan = k_hi - lk0;
  }
  int dho = 0, dwo = 0, tapoff = 0;
  if (ntaps > 0) { dho = taps[0]; dwo = taps[1]; tapoff = taps[2]; }
  f32x4 ra[AP], rb[SP ? 1 : BP];
  // SP: the [32 k][BN c] weight tile of every plane in 16-byte pieces (8 input channels of one output channel's tap): 32 * BN / 8 pieces per plane
  constexpr int BPP = SP ? 32 * (BN / 8) / 256 : 1;
  u32x4 rbp[SP ? 3 : 1][BPP];
  int boffp[BPP], bldsp[BPP];
  const rsrc_t rwp = make_rsrc(SP ? (const void*)p.w_planes : (const void*)w, SP ? (unsigned)(3 * p.wp_stride * 2) : 16u);
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < BPP; ++i) {
      const int idx = tid + 256 * i, kr = idx / (BN / 8), c8 = idx % (BN / 8);
      boffp[i] = n0 + c8 * 8 < p.C ? (kr * p.RSC + n0 + c8 * 8) * 2 : OOB_OFF;
      bldsp[i] = splitbf::krow_off<BN * 2>(kr, c8);
    }
  }
  auto load_tile = [&]() {
    const int toff_a = ((dho * p.Wo + dwo) * p.K + lk0) * 4;   // uniform (may be negative: folded into the lane offset)
    const int toff_b = (lk0 * p.RSC + tapoff) * 4;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = inb | (((unsigned)(hq_[i] + dho) < (unsigned)p.Ho) & ((unsigned)(wq_[i] + dwo) < (unsigned)p.Wo));
      ra[i] = bload4(rdy, ok ? aoff[i] + toff_a : OOB_OFF, 0);
    }
    if constexpr (SP) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < BPP; ++i)
          rbp[q][i] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rwp, boffp[i] == OOB_OFF ? OOB_OFF : boffp[i] + q * (int)(p.wp_stride * 2), toff_b >> 1, 0);
    } else {
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = bload4(rw, boff + BRP * i * p.RSC * 4, toff_b);
    }
    // straight-line advance to the next (k-tile, tap) with the TAPS innermost: consecutive tiles re-read the same dY pixels'
    // channel slice one tap over and hit in L2 (see the forward kernel)
    ti += 1;
    const bool w1 = ti >= ntaps;
    ti = w1 ? 0 : ti;
    lk0 += w1 ? BK : 0;
    dho = taps[3 * ti]; dwo = taps[3 * ti + 1]; tapoff = taps[3 * ti + 2];
  };
  constexpr int A_PLANE = BM * splitbf::ROWB, B_PLANE = 32 * BN * 2;
  unsigned char* const Asb = reinterpret_cast<unsigned char*>(smem);
  unsigned char* const Bsb = Asb + 3 * A_PLANE;
  auto store_tile = [&](int buf) {
    if constexpr (SP) {
#pragma unroll
      for (int i = 0; i < AP; ++i) splitbf::rowk_store(Asb, A_PLANE, rsub + RPP * i, tid % CH, ra[i]);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < BPP; ++i) *reinterpret_cast<u32x4*>(Bsb + q * B_PLANE + bldsp[i]) = rbp[q][i];
      return;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[(rsub + RPP * i) * LDT + chunk]) = ra[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow + BRP * i) * BN + bcol]) = rb[i];
  };
  if constexpr (SP) k_loop_split<2 * TM, 2 * TN, false, true, 128, BN * 2>(ntaps * (kspan / BK), Asb, A_PLANE, Bsb, B_PLANE, wr0, wc0, lane, acc16, load_tile, store_tile);
  else k_loop<TM, TN, true, false, LDT, BN, BK, AP + BP>(ntaps * (kspan / BK), As, Bs, wr0, wc0, lane, acc, load_tile, store_tile);

  constexpr int EP_FLOATS = 4 * 32 * (TN * 32 + 4);
  static_assert(!EPI || EP_FLOATS <= STAGE, "the fused-activation dgrad needs the vectorised epilogue");
  static_assert(GATE == 0 || (EP_FLOATS <= STAGE && BM / WGM == 64), "the gated epilogue is the vectorised one, partials per 64 rows");
  if constexpr (EP_FLOATS <= STAGE) {       // C % 4 == 0 is a precondition of this kernel
    if constexpr (GATE != 0) {
      // partial index: (parity class, row tile, 64-row group of the wave) - the grid has the same number of row tiles for every class
      const long long grp = ((long long)blockIdx.y * (gridDim.x / NT) + mt) * (BM / 64) + (wr0 / 64);
      epilogue_vec<TM, TN, 0, false, GATE>(sel_acc<SP>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.C, nullptr, addend, dx, (long long)p.N * p.H * p.W * p.C,
                                               [&](int r) -> long long { const unsigned pix = rowpix[wr0 + r]; return pix != 0xffffffffu ? (long long)pix * p.C : -1; },
                                               nullptr, nullptr, nullptr, nullptr, 0, &p.gate, grp);
    } else {
      epilogue_vec<TM, TN, EPI ? 2 : 0>(sel_acc<SP>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, n0 + wc0, p.C, nullptr, addend, dx, (long long)p.N * p.H * p.W * p.C,
                                        [&](int r) -> long long { const unsigned pix = rowpix[wr0 + r]; return pix != 0xffffffffu ? (long long)pix * p.C : -1; },
                                        nullptr, EPI ? p.aux_in : nullptr);
    }
    return;
  }
  if constexpr (SP) return;
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wc0 + tn * 32 + l31;
    if (col >= p.C) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned pix = rowpix[wr0 + tm * 32 + (j & 3) + 8 * (j >> 2) + 4 * h];
        if (pix != 0xffffffffu) {
          const size_t o = (size_t)pix * p.C + col;
          float v = acc[tm][tn][j];
          if (addend) v += addend[o];
          dx[o] = v;
        }
      }
    }
  }
}

// =============================================================================================
// wgrad: partial[split][K][RSC] over a chunk of the N*Ho*Wo contraction
// =============================================================================================
// GATHER: 3 ROWS (the 3-channel image stem, unpadded: column j = 24 r + e is float e of the 3 S contiguous floats that filter row r sees at an output pixel;
//           the output is dW' [K][R][24] with the columns e >= 3 S zero - see conv_fwd_k's row-taps loader),
//         0 generic (fastdiv per staged row and tile), 1 LIN (1x1 / stride 1 / no padding: X row m is x + m*C),
//         2 S1 (stride 1, any filter/padding: X row of tap (r,s) is x + (m + (r-pad)*W + (s-pad))*C, validity from
//           (ho, wo) kept incrementally per staged row - no division in the loop)
// XF: x is the producer's RAW conv output; the X operand is relu(x * xf_scale[c] + xf_shift[c]) (the activation the forward never
//     materialised), formed on the way into LDS.  A thread's four channels are loop constants, so are its scale / shift registers.
// DYF (LIN gather only): the dY operand is formed on load from (g, x of the BatchNorm behind this convolution, coefficients) - see conv_fwd_k.
// BIAS (LIN gather): the workgroups of column tile 0 also sum the dY rows they stage - the bias gradient's column sums, one partial row per split
//       (p.aux_out [nsplit][K]), reduced with the weight gradient's slabs in the same fixed order: the stand-alone column-sum pass over dY disappears.
// FLUSH: two-level accumulation, see k_loop (the blocked batched weight gradient only).
// SP (SSV_ARITH_BF16X3, csrc/split_bf16.h; float4 gathers only): BOTH operands are activations, so both are split into three bf16 planes while they are staged; the LDS
//     images stay k-major ([32 pixels][channels], as the rows lie in HBM) and the fragments come out of them by transposed reads (ds_read_b64_tr_b16).
template <int BM, int BN, int WGM, int WGN, int BK, bool VECB, int GATHER, bool XF = false, bool DYF = false, bool BIAS = false, int FLUSH = 0, bool SP = false>
__global__ void __launch_bounds__(256, (SP && BM == 128 && BN == 128) ? 2 : 1)
conv_wgrad_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int chunk_rows, int tiles) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  static_assert(!SP || (VECB && GATHER != 3 && BK == 32), "the bf16-piece variants: float4 gathers, K-step 32");
  constexpr int STAGE = SP ? 48 * (BM + BN) : BK * (BM + BN);   // SP: three planes of [32][channels] bf16 per operand
  constexpr int EP_FLOATS = 4 * 32 * (TN * 32 + 4);            // the vectorised epilogue's staging area (one 32-row slab per wave)
  static_assert(!XF || VECB, "the fused-input variant is the float4 path");
  static_assert(!DYF || (VECB && GATHER == 1), "the BatchNorm-backward operand: 1x1 / stride 1 layers");
  __shared__ __attribute__((aligned(16))) float smem[STAGE > EP_FLOATS ? STAGE : EP_FLOATS];
  float* As = smem;               // [BK][BM]
  float* Bs = smem + BK * BM;     // [BK][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave / WGN) * (BM / WGM), wc0 = (wave % WGN) * (BN / WGN);
  const int NCOL = GATHER == 3 ? p.R * 24 : p.RSC;      // columns of the weight-gradient matrix this launch produces
  const int JT = (NCOL + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);     // tiles of one row-slab are neighbours: they share dY / X rows in L2
  const int split = bid / tiles, tile = bid - split * tiles;
  const int it = tile / JT, jt = tile - it * JT;
  const int i0 = it * BM, j0 = jt * BN;
  const int ms = split * chunk_rows;
  const int me = min(ms + chunk_rows, p.M);
  x += (size_t)blockIdx.y * p.bs_a; dy += (size_t)blockIdx.y * p.bs_b; partial += (size_t)blockIdx.y * p.bs_o;     // batched weight gradients
  if constexpr (GATHER != 3) {
    if (p.Kg != 0 && j0 / p.C == (min(j0 + BN, NCOL) - 1) / p.C) {     // block-diagonal bank, column tile inside one tap (uniform)
      const int c0 = j0 % p.C, c1 = c0 + min(BN, NCOL - j0);
      // rows i0.. are output channels, columns input channels c0..c1 of one tap: a tile that no group touches is identically zero
      // and is NOT written - ssv_group_extract reads the diagonal blocks only
      if ((min(i0 + BM, p.K) - 1) / p.Kg < c0 / p.Cg || (c1 - 1) / p.Cg < i0 / p.Kg) return;
    }
  }

  f32x16 acc[SP ? 1 : TM][SP ? 1 : TN];
  f32x4 acc16[SP ? 2 * TM : 1][SP ? 2 * TN : 1];
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else zero_acc<SP ? 1 : TM, SP ? 1 : TN>(acc);

  // A = dY rows (float4 along output channels)
  constexpr int ACV = BM / 4, ARP = 256 / ACV, AP = BK / ARP;
  const int acol = (tid % ACV) * 4, arow = tid / ACV;
  const bool acok = i0 + acol < p.K;
  // B = gathered X (float4 along input channels when C % 4 == 0, scalar otherwise)
  constexpr int BCV = VECB ? BN / 4 : BN, BRP = GATHER == 3 ? 1 : 256 / BCV, BP = GATHER == 3 ? (BK * BCV + 255) / 256 : BK / BRP;
  const int bcol = VECB ? (tid % BCV) * 4 : (tid % BCV), brow = tid / BCV;
  const int j = j0 + bcol;
  const bool jok = j < p.RSC;
  const uint32_t tap = fdiv((uint32_t)(jok ? j : 0), p.dC);
  const int cj = (jok ? j : 0) - (int)tap * p.C;
  const int rj = (int)fdiv(tap, p.dS);
  const int sj = (int)tap - rj * p.S;

  // buffer-load loaders: the row index m only enters through a UNIFORM byte offset (mcur * row pitch), the per-lane offsets
  // are loop constants; rows past the end of the tensor read as zeros (a chunk is a whole number of K-steps, so rows
  // >= me exist only at the tensor end), invalid columns carry the out-of-range offset.
  const rsrc_t rdy = make_rsrc(dy, (unsigned)p.M * p.K * 4u);
  const rsrc_t rx = make_rsrc(x, (unsigned)p.N * p.H * p.W * p.C * 4u);
  int mcur = ms;
  f32x4 ra[AP];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};          // BIAS: this thread's four dY columns summed over the rows it stages
  f32x4 rbv[VECB ? BP : 1];
  float rbs[VECB ? 1 : BP];
  int avoff[AP], bvoff[BP];
  f32x4 ra2[DYF ? AP : 1], dco[DYF ? 4 : 1];
  const rsrc_t rdx = make_rsrc(DYF ? p.dyin_x : dy, (unsigned)p.M * p.K * 4u);
  int a_ok = 0;
  if constexpr (DYF) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dco[j] = acok ? *reinterpret_cast<const f32x4*>(p.dyin_coef + (size_t)j * p.K + i0 + acol) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int xf_ok = 0;
  f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (XF) { if (jok) { xsc = *reinterpret_cast<const f32x4*>(p.xf_scale + cj); xsh = *reinterpret_cast<const f32x4*>(p.xf_shift + cj); } }
#pragma unroll
  for (int i = 0; i < AP; ++i) avoff[i] = acok ? ((arow + ARP * i) * p.K + i0 + acol) * 4 : OOB_OFF;
#pragma unroll
  for (int i = 0; i < BP; ++i) bvoff[i] = jok ? ((brow + BRP * i) * p.C + cj) * 4 : OOB_OFF;
  // S1 state: (ho, wo) of each staged row, byte offset of (row, tap shift, channel) relative to row 0 of the tile
  int s1_ho[BP], s1_wo[BP], s1_base[BP];
  const int s1_qw = BK / p.Wo, s1_rw = BK - s1_qw * p.Wo;
  if constexpr (GATHER == 2) {
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const uint32_t mm = (uint32_t)min(ms + brow + BRP * i, p.M - 1);
      const uint32_t n = fdiv(mm, p.dHoWo);
      const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
      s1_ho[i] = (int)fdiv(rem, p.dWo);
      s1_wo[i] = (int)rem - s1_ho[i] * p.Wo;
      s1_base[i] = ((brow + BRP * i + (rj - p.pad) * p.W + (sj - p.pad)) * p.C + cj) * 4;
    }
  }
  // ROWS state: float4 f = tid + 256 i of the [BK][BN] tile: staged row f / BCV, column 4 (f % BCV) = 24 r + e0
  int rw_row[GATHER == 3 ? BP : 1], rw_r[GATHER == 3 ? BP : 1], rw_e0[GATHER == 3 ? BP : 1];
  unsigned rw_mask[GATHER == 3 ? BP : 1];
  int rw_rot[GATHER == 3 ? BP : 1];
  if constexpr (GATHER == 3) {
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int f = tid + 256 * i, col = j0 + (f % BCV) * 4;
      rw_row[i] = f / BCV;
      rw_r[i] = col / 24;
      rw_e0[i] = col - rw_r[i] * 24;
      if (f >= BK * BCV || col >= NCOL) rw_r[i] = -1;            // not a column / row of this tile
      rw_mask[i] = 0;
      rw_rot[i] = 0;
    }
  }
  auto load_tile = [&]() {
    const int soff_a = mcur * p.K * 4;
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = bload4(rdy, avoff[i], soff_a);
    if constexpr (GATHER == 3) {
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int m = mcur + rw_row[i];
        bool ok = rw_r[i] >= 0 && m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t n = fdiv(mm, p.dHoWo);
        const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
        const uint32_t ho = fdiv(rem, p.dWo);
        const uint32_t wo = rem - ho * (uint32_t)p.Wo;
        const int hi = (int)ho * p.stride - p.pad + rw_r[i], wi0 = (int)wo * p.stride - p.pad;
        ok = ok & ((unsigned)hi < (unsigned)p.H);
        int rot;
        rbv[i] = bload4(rx, straddle_off(ok ? (((int)n * p.H + hi) * p.W + wi0) * 12 + rw_e0[i] * 4 : OOB_OFF, p.N * p.H * p.W * 12, rot), 0);
        rw_rot[i] = rot;
        unsigned mk = 0;                                          // element e is pixel (e0 + e) / 3 of the window: inside the image row and a real tap?
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int px = (rw_e0[i] + e) / 3; mk |= (unsigned)((px < p.S) & ((unsigned)(wi0 + px) < (unsigned)p.W)) << e; }
        rw_mask[i] = mk;
      }
      mcur += BK;
      return;
    }
    if constexpr (DYF) {
      a_ok = 0;
#pragma unroll
      for (int i = 0; i < AP; ++i) { ra2[i] = bload4(rdx, avoff[i], soff_a); a_ok |= (int)(mcur + arow + ARP * i < p.M) << i; }
    }
    if constexpr (GATHER == 1) {
      const int soff_b = mcur * p.C * 4;
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        if constexpr (VECB) rbv[i] = bload4(rx, bvoff[i], soff_b);
        else rbs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, bvoff[i], soff_b, 0));
      }
    } else if constexpr (GATHER == 2) {
      const int moff = mcur * p.C * 4;
      if constexpr (XF) xf_ok = 0;
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int m = mcur + brow + BRP * i;
        const bool ok = jok & (m < p.M) & ((unsigned)(s1_ho[i] + rj - p.pad) < (unsigned)p.H) & ((unsigned)(s1_wo[i] + sj - p.pad) < (unsigned)p.W);
        if constexpr (XF) xf_ok |= (int)ok << i;
        const int off = ok ? s1_base[i] + moff : OOB_OFF;
        if constexpr (VECB) rbv[i] = bload4(rx, off, 0);
        else rbs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
        // advance (ho, wo) by BK output pixels: BK = qW*Wo + rW with a single carry, then wrap ho per image
        int wo = s1_wo[i] + s1_rw;
        const int carry = wo >= p.Wo ? 1 : 0;
        wo -= carry ? p.Wo : 0;
        int ho = s1_ho[i] + s1_qw + carry;
        ho -= ho >= p.Ho ? p.Ho : 0;
        s1_wo[i] = wo; s1_ho[i] = ho;
      }
    } else {
      if constexpr (XF) xf_ok = 0;
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int m = mcur + brow + BRP * i;
        bool ok = jok & (m < p.M);
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t n = fdiv(mm, p.dHoWo);
        const uint32_t rem = mm - n * (uint32_t)(p.Ho * p.Wo);
        const uint32_t ho = fdiv(rem, p.dWo);
        const uint32_t wo = rem - ho * (uint32_t)p.Wo;
        const int hi = (int)ho * p.stride - p.pad + rj, wi = (int)wo * p.stride - p.pad + sj;
        ok = ok & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
        if constexpr (XF) xf_ok |= (int)ok << i;
        const int off = ok ? ((((int)n * p.H + hi) * p.W + wi) * p.C + cj) * 4 : OOB_OFF;
        if constexpr (VECB) rbv[i] = bload4(rx, off, 0);
        else rbs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      }
    }
    mcur += BK;
  };
  auto xform_tile = [&]() {
    if constexpr (GATHER == 3) {
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        if (rw_rot[i] != 0) rbv[i] = straddle_fix(rbv[i], rw_rot[i]);       // the first / last pixels of the whole batch only
#pragma unroll
        for (int e = 0; e < 4; ++e) rbv[i][e] = ((rw_mask[i] >> e) & 1u) ? rbv[i][e] : 0.f;
      }
    }
    if constexpr (DYF) {         // rows past M must contribute nothing (g and x read as zeros there, the affine form would not)
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = (a_ok >> i) & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[i][e] = ok ? __builtin_fmaf(ra[i][e], dco[0][e], __builtin_fmaf(ra2[i][e] - dco[1][e], dco[2][e], dco[3][e])) : 0.f;
      }
    }
    if constexpr (XF) {
      // same fmaf / fmaxf as bn_apply_k: bit-identical to the materialised activation; padding taps (and, in the LIN mode, rows
      // past M, whose dY rows are zero anyway) must not pick up relu(shift)
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const float cap = (GATHER == 1 || ((xf_ok >> i) & 1)) ? __builtin_inff() : 0.f;      // ReLU + zero padding tap as one clamp (v_med3_f32)
#pragma unroll
        for (int e = 0; e < 4; ++e) rbv[i][e] = __builtin_amdgcn_fmed3f(__builtin_fmaf(rbv[i][e], xsc[e], xsh[e]), 0.f, cap);
      }
    }
  };
  constexpr int A_PLANE = 32 * BM * 2, B_PLANE = 32 * BN * 2;            // SP: bytes of one plane of the dY / X image
  unsigned char* const Asb = reinterpret_cast<unsigned char*>(smem);
  unsigned char* const Bsb = Asb + 3 * A_PLANE;
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      if constexpr (SP) splitbf::krow_store<BM * 2>(Asb, A_PLANE, arow + ARP * i, acol, ra[i]);
      else *reinterpret_cast<f32x4*>(&As[(arow + ARP * i) * BM + acol]) = ra[i];
    }
    if constexpr (BIAS) {
      if (jt == 0) {
#pragma unroll
        for (int i = 0; i < AP; ++i) bsum += ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      if constexpr (GATHER == 3) {
        const int f = tid + 256 * i;
        if (f < BK * BCV) *reinterpret_cast<f32x4*>(&Bs[(f / BCV) * BN + (f % BCV) * 4]) = rbv[i];
      } else if constexpr (SP) splitbf::krow_store<BN * 2>(Bsb, B_PLANE, brow + BRP * i, bcol, rbv[i]);
      else if constexpr (VECB) *reinterpret_cast<f32x4*>(&Bs[(brow + BRP * i) * BN + bcol]) = rbv[i];
      else Bs[(brow + BRP * i) * BN + bcol] = rbs[i];
    }
  };
  if constexpr (SP) k_loop_split<2 * TM, 2 * TN, true, true, BM * 2, BN * 2, FLUSH, 1, true>((me - ms + BK - 1) / BK, Asb, A_PLANE, Bsb, B_PLANE, wr0, wc0, lane, acc16, load_tile, store_tile, xform_tile);
  else k_loop<TM, TN, false, false, BM, BN, BK, (VECB ? AP + BP + (DYF ? AP : 0) : 0), FLUSH>((me - ms + BK - 1) / BK, As, Bs, wr0, wc0, lane, acc, load_tile, store_tile, xform_tile);

  if constexpr (BIAS) {
    static_assert(STAGE >= (256 / (BM / 4)) * BM, "the bias partials are folded through the stage buffer");
    if (jt == 0) {               // uniform: fold the ARP row groups of each column in fixed order, one partial row per split
      __syncthreads();
      *reinterpret_cast<f32x4*>(&smem[arow * BM + acol]) = bsum;
      __syncthreads();
      if (tid < BM && i0 + tid < p.K) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < ARP; ++r) t += smem[r * BM + tid];
        p.aux_out[(size_t)split * p.K + i0 + tid] = t;
      }
      __syncthreads();
    }
  }
  float* out = partial + (size_t)split * p.K * NCOL;
  if ((NCOL & 3) == 0) {       // whole 16-byte row segments through the wave's LDS slab (4x fewer store instructions), as in the forward kernel
    const int rbase = i0 + wr0;
    epilogue_vec<TM, TN>(sel_acc<SP>(acc, acc16), smem + wave * (32 * (TN * 32 + 4)), lane, j0 + wc0, NCOL, nullptr, nullptr, out, (long long)p.K * NCOL,
                         [&](int r) -> long long { const int row = rbase + r; return row < p.K ? (long long)row * NCOL : -1; });
    return;
  }
  if constexpr (SP) return;            // NCOL % 4 == 0 follows from C % 4 == 0, a launch precondition of the bf16-piece variants
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = j0 + wc0 + tn * 32 + l31;
    if (col >= p.RSC) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int row = i0 + wr0 + tm * 32 + (jj & 3) + 8 * (jj >> 2) + 4 * h;
        if (row < p.K) out[(size_t)row * p.RSC + col] = acc[tm][tn][jj];
      }
    }
  }
}

// dw (+)= sum over split slabs, fixed order: 64 consecutive elements x 4 groups of slabs per workgroup
__global__ void __launch_bounds__(256)
wgrad_reduce_k(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ dw, int accumulate) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  partial += (size_t)blockIdx.y * nsplit * n; dw += (size_t)blockIdx.y * n;          // batched weight gradients: [batch][split][n] -> [batch][n]
  const int per = (nsplit + 3) / 4;
  const int k0 = grp * per, k1 = min(k0 + per, nsplit);
  float s = 0.f;
  if (i < n) {
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      const float a = partial[(size_t)k * n + i], b = partial[(size_t)(k + 1) * n + i];
      const float c = partial[(size_t)(k + 2) * n + i], d = partial[(size_t)(k + 3) * n + i];
      s += (a + b) + (c + d);
    }
    for (; k < k1; ++k) s += partial[(size_t)k * n + i];
  }
  sm[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < n) dw[i] = (accumulate ? dw[i] : 0.f) + ((sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]));
}

// ---------------------------------------------------------------------------------------------
int check_desc(const ssv_conv_desc* d, const char* who) {
  SSV_REQUIRE(d != nullptr, "%s: null descriptor", who);
  SSV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0,
              "%s: non-positive dimension", who);
  const int Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
  SSV_REQUIRE(Ho == d->Ho && Wo == d->Wo, "%s: Ho/Wo (%d,%d) inconsistent with input/filter/stride/pad (expect %d,%d)", who, d->Ho, d->Wo, Ho, Wo);
  SSV_REQUIRE((int64_t)d->N * d->H * d->W < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo < (1ll << 31), "%s: too many pixels", who);
  SSV_REQUIRE((int64_t)d->R * d->S * d->C < (1 << 24) && d->R * d->S <= 64 && d->H < 32768 && d->W < 32768, "%s: filter too large", who);
  // the kernels address activations and filters through buffer descriptors with non-negative 32-bit byte offsets
  SSV_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (1ll << 29) - (1 << 22) && (int64_t)d->N * d->Ho * d->Wo * d->K < (1ll << 29) - (1 << 22) &&
              (int64_t)d->K * d->R * d->S * d->C < (1ll << 29), "%s: tensor of 2 GiB or more - split the batch", who);
  return SSV_OK;
}

ConvKP make_kp(const ssv_conv_desc* d) {
  ConvKP p;
  p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.R = d->R; p.S = d->S;
  p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
  p.M = d->N * d->Ho * d->Wo;
  p.RSC = d->R * d->S * d->C;
  p.dHoWo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.dWo = make_fastdiv((uint32_t)d->Wo);
  p.dC = make_fastdiv((uint32_t)d->C);
  p.dS = make_fastdiv((uint32_t)d->S);
  p.dWp = make_fastdiv((uint32_t)d->W + 2u);
  p.aux_out = nullptr; p.aux_out2 = nullptr; p.aux_in = nullptr; p.xf_scale = nullptr; p.xf_shift = nullptr;
  memset(&p.gate, 0, sizeof(p.gate));
  p.dyin_x = nullptr; p.dyin_coef = nullptr;
  p.sum_res = p.sum_scale = p.sum_shift = p.sum_rscale = p.sum_rshift = nullptr; p.sum_out = nullptr; p.sum_mask = nullptr;
  p.bs_a = p.bs_b = p.bs_o = 0;
  p.add_H2 = p.add_W2 = 0;
  p.Cg = p.Kg = 0;
  p.w_planes = (const unsigned short*)d->w_planes;
  p.wp_stride = (long long)d->K * p.RSC;
  return p;
}

struct WgradPlan { int bm, bn, it, jt, nsplit, chunk; };
WgradPlan plan_wgrad(const ssv_conv_desc* d, int groups = 0) {
  WgradPlan w;
  const int RSC = d->R * d->S * d->C;
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  const bool bd = groups > 1 && d->C % 64 == 0;     // block-diagonal bank: 64 x 64 tiles, of which only those a group touches do any work
  w.bm = (d->K >= 128 && !bd) ? 128 : 64;
  w.bn = (RSC <= 64 || bd) ? 64 : 128;      // 1x1 convs on 64 channels: a 128-wide tile would be half empty
#ifdef SSV_EXP_WGRAD_BN64                   // diagnostic builds only (round 6): the bf16x3 weight gradient on 128 x 64 tiles (half the accumulator registers: 3-4 workgroups per CU)
  if (d->arithmetic == SSV_ARITH_BF16X3 && w.bm == 128) w.bn = 64;
#endif
  w.it = cdiv(d->K, w.bm);
  w.jt = cdiv(RSC, w.bn);
  int tiles = w.it * w.jt;
  if (bd) {                                 // tiles that work: per tap and row tile, the column tiles of the row tile's groups
    const int cg = d->C / groups, kg = d->K / groups;
    const int per_row = cdiv((cdiv(w.bm, kg) + 1) * cg, w.bn) + 1;
    tiles = w.it * d->R * d->S * (per_row < d->C / w.bn ? per_row : d->C / w.bn);
  }
  // one resident round: 3 workgroups per CU for the 128-row kernels, 4 for the 64-row ones.  Round DOWN - one workgroup more
  // than the chip holds costs a whole second round for the stragglers.
  const int slots = w.bm == 128 ? 768 : 1024;
  int64_t ns = slots / tiles;
  const int64_t max_by_rows = cdiv64(M, 256);
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  int64_t chunk = cdiv64(cdiv64(M, ns), 32) * 32;     // whole K-steps for both BK = 16 and 32
  w.chunk = (int)chunk;
  w.nsplit = (int)cdiv64(M, chunk);
  return w;
}

}  // namespace

// ---- launch selection (compile-time variants only: no environment, no global state) ---------------------------
// K-step 32 when the contraction's channel count allows it, else 16; tile by output width.
namespace {


// Does a forward-kernel launch described by d run its bf16-piece variant (SSV_ARITH_BF16X3)?  The float4 path with whole 32-channel k-tiles, 16-byte output rows, a
// pre-split weight operand (a block-diagonal bank included: its tiles contract over their own groups' channels in either arithmetic).
inline bool sp_fwd_ok(const ssv_conv_desc* d, int groups = 0) {
  return d->arithmetic == SSV_ARITH_BF16X3 && d->w_planes != nullptr && (((uintptr_t)d->w_planes) & 15) == 0 && d->C % 32 == 0 && d->K % 4 == 0;
}
// ... and with how many accumulators per tile: one up to a contraction of SP_DUAL_FROM products per output, two beyond (a property of the LAYER - every fused and
// unfused variant of a layer takes the same form, so they stay bit-identical).  Measured on the 53 ResNet-50 layers (tests/test_gpu_split.py): with one accumulator
// the error against fp64 is 0.8-1.0x the fp32-MFMA kernel's up to a contraction of 1,152 and 1.05-1.07x from 2,048 on.
constexpr int SP_DUAL_FROM = 1152;
inline int sp_fwd_mode(const ssv_conv_desc* d, int groups = 0) {
  if (!sp_fwd_ok(d, groups)) return 0;
  return (int64_t)d->R * d->S * d->C > SP_DUAL_FROM ? 2 : 1;
}
// the strided data-gradient kernel: its 128 x 128 tile (C >= 128), whole 32-channel k-tiles of the output channels, the weights pre-split
inline bool sp_dgrad_ok(const ssv_conv_desc* d) {
  return d->arithmetic == SSV_ARITH_BF16X3 && d->w_planes != nullptr && (((uintptr_t)d->w_planes) & 15) == 0 && d->K % 32 == 0 && d->C >= 128 && d->C % 8 == 0;
}
inline bool sp_wgrad_ok(const ssv_conv_desc* d, int groups = 0) {
  return d->arithmetic == SSV_ARITH_BF16X3 && d->C % 4 == 0 && d->K % 4 == 0;
}

// forward family: optional statistics epilogue (pmean / pm2) and optional fused input BatchNorm + ReLU (in_scale / in_shift)
int launch_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias, const float* addend, float* y,
               float* pmean, float* pm2, const float* in_scale, const float* in_shift, hipStream_t s, const ssv_bn_gate* gate = nullptr,
               int add_H2 = 0, int add_W2 = 0, int groups = 0) {
  ConvKP p = make_kp(d);
  p.add_H2 = add_H2; p.add_W2 = add_W2;
  p.aux_out = pmean; p.aux_out2 = pm2; p.xf_scale = in_scale; p.xf_shift = in_shift;
  if (groups > 1) { p.Cg = d->C / groups; p.Kg = d->K / groups; }
  const bool stats = pmean != nullptr, xf = in_scale != nullptr;
  const int sp = sp_fwd_mode(d, groups);
  const bool wide = d->K >= 128 && groups <= 1;               // block-diagonal banks: the 64-column tile sees the fewest foreign groups
  // (1x1 layers with few k-tiles - 64 -> 256 at 56x56 runs at 2.7 TB/s and 69 TFLOP/s, the SUM of its MFMA and HBM times - were tried on a 128 x 64 tile
  //  at 4 / 5 workgroups per CU, on a 64 x 256 tile writing whole 1 KB rows and as a persistent kernel that loads its next tile under the epilogue: no change, r03 x3)
  const unsigned grid = (unsigned)(wide ? cdiv(p.M, 128) * cdiv(d->K, 128) : (sp ? cdiv(p.M, 128) : cdiv(p.M, 256)) * cdiv(d->K, 64));
  // 3x3 / stride 1 / padding 1 on the 256 x 64 tile: the halo loader (conv_fwd_k, C4 == 3) when the tile's rows fit its LDS records.  DIAGNOSTIC BUILDS ONLY
  // (-DSSV_EXP_HALO, round 4): three different main loops for this layer - the generic one, one halo stage of 32 channels, two halo stages of 16 - all land on
  // 104-109 TFLOP/s, because the kernel already keeps the matrix pipe 0.80-0.82 busy and the chip clocks it at 2.0-2.1 GHz under that load
  // (profiles/r04_probe_halo_loader.txt); the shipped library keeps the one generic loader.
#ifdef SSV_EXP_HALO
  const bool halo = !wide && groups <= 1 && !xf && add_H2 == 0 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->C % 32 == 0 && d->K % 4 == 0 &&
                    d->Ho == d->H && d->Wo == d->W && halo_records(d->W) <= HALO_REC;
#endif
  if (gate) {                                                  // C % 32 == 0 checked by the caller
    p.gate = *gate;
#ifdef SSV_EXP_HALO
    if (halo) {
#define FWDGH(G_) hipLaunchKernelGGL((conv_fwd_k<256, 64, 4, 1, 32, true, 0, false, 3, false, G_>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y)
      if (gate->x2) FWDGH(3); else if (gate->mask) FWDGH(2); else FWDGH(1);
#undef FWDGH
      return SSV_OK;
    }
#endif
#define FWDG_(BM_, BN_, WM_, WN_, G_, SP_) \
  hipLaunchKernelGGL((conv_fwd_k<BM_, BN_, WM_, WN_, 32, true, false, false, false, false, G_, 0, false, false, SP_>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y)
#define FWDG(BM_, BN_, WM_, WN_, G_) FWDG_(BM_, BN_, WM_, WN_, G_, 0)
#define FWDG_TILE(G_) do { if (sp == 2) { if (wide) FWDG_(128, 128, 2, 2, G_, 2); else FWDG_(128, 64, 2, 2, G_, 2); } \
                           else if (sp) { if (wide) FWDG_(128, 128, 2, 2, G_, 1); else FWDG_(128, 64, 2, 2, G_, 1); } \
                           else    { if (wide) FWDG(128, 128, 2, 2, G_); else FWDG(256, 64, 4, 1, G_); } } while (0)
    if (add_H2 > 0) {                                          // compact stride-2 addend: wide tile, byte-mask gates (checked by the caller)
#define FWDGS_(G_, SP_) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, false, false, false, false, G_, 0, true, false, SP_>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y)
#define FWDGS(G_) do { if (sp == 2) FWDGS_(G_, 2); else if (sp) FWDGS_(G_, 1); else FWDGS_(G_, 0); } while (0)
      if (gate->x2) FWDGS(3); else FWDGS(2);
#undef FWDGS
#undef FWDGS_
    } else if (gate->x2) FWDG_TILE(3); else if (gate->mask) FWDG_TILE(2); else FWDG_TILE(1);
#undef FWDG_TILE
#undef FWDG
#undef FWDG_
    return SSV_OK;
  }
#define FWD_(BM_, BN_, WM_, WN_, BK_, ST_, C4_, XF_, SP_) \
  hipLaunchKernelGGL((conv_fwd_k<BM_, BN_, WM_, WN_, BK_, true, false, ST_, C4_, XF_, 0, 0, false, false, SP_>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y)
#define FWD(BM_, BN_, WM_, WN_, BK_, ST_, C4_, XF_) FWD_(BM_, BN_, WM_, WN_, BK_, ST_, C4_, XF_, 0)
// the float4 path with K-step 32: either arithmetic
#define FWD_TILE_A(ST_, XF_) do { \
    if (sp == 2) { if (wide) FWD_(128, 128, 2, 2, 32, ST_, 0, XF_, 2); else FWD_(128, 64, 2, 2, 32, ST_, 0, XF_, 2); } \
    else if (sp) { if (wide) FWD_(128, 128, 2, 2, 32, ST_, 0, XF_, 1); else FWD_(128, 64, 2, 2, 32, ST_, 0, XF_, 1); } \
    else    { if (wide) FWD_(128, 128, 2, 2, 32, ST_, 0, XF_, 0); else FWD_(256, 64, 4, 1, 32, ST_, 0, XF_, 0); } } while (0)
#define FWD_TILE(BK_, ST_, C4_, XF_) do { if (wide) FWD(128, 128, 2, 2, BK_, ST_, C4_, XF_); else FWD(256, 64, 4, 1, BK_, ST_, C4_, XF_); } while (0)
#ifdef SSV_EXP_HALO
  if (halo) {
    if (stats) FWD(256, 64, 4, 1, 32, true, 3, false); else FWD(256, 64, 4, 1, 32, false, 3, false);
  } else
#endif
  if (stats && d->C == 4) {                                    // the padded image stem with the statistics epilogue
    FWD_TILE(32, true, true, false);
  } else if (stats || xf) {                                    // C % 32 == 0 checked by the callers
    if (stats && xf) FWD_TILE_A(true, true);
    else if (stats)  FWD_TILE_A(true, false);
    else             FWD_TILE_A(false, true);
  } else if (d->C == 4) {                                      // image stems (3 channels padded to 4): tap-vector gather
    FWD_TILE(32, false, true, false);
  } else if (d->C % 32 == 0) {
#ifdef SSV_EXP_S2      // diagnostic builds only: two-stage main loop for wide plain launches with at least SSV_EXP_S2 k-tiles
    if (wide && groups <= 1 && p.RSC / 32 >= SSV_EXP_S2) {
      hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 0, false, 0, false, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y);
      return SSV_OK;
    }
#endif
    FWD_TILE_A(false, false);
  } else if (d->C % 16 == 0) {
    FWD_TILE(16, false, false, false);
  } else {
    const unsigned g2 = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 64));
    hipLaunchKernelGGL((conv_fwd_k<128, 64, 2, 2, GBK, false>), dim3(g2), dim3(256), 0, s, p, x, w, bias, addend, y);
  }
#undef FWD_TILE
#undef FWD_TILE_A
#undef FWD
#undef FWD_
  return SSV_OK;
}

}  // namespace

extern "C" int ssv_conv2d_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* addend, float* y, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd")) return rc;
  SSV_REQUIRE(x && w && y, "ssv_conv2d_fwd: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "ssv_conv2d_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  launch_fwd(d, x, w, bias, addend, y, nullptr, nullptr, nullptr, nullptr, s);
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd");
  return SSV_OK;
}

// Grouped convolutions (conv3x3(groups = 32) of the ResNeXt bottlenecks, networks/resnet.py:8-10,57) on their DENSE block-diagonal filter bank
// (ssv_group_expand): the same kernels, but every output-column tile contracts only over the channels of the groups it falls into -
// C / 64 (forward, stride-1 data gradient), K / 64 (strided data gradient) times fewer k-tiles than the dense product; exact, the skipped
// products are all against zero weights.  groups must divide C and K.
namespace {
int check_groups(const ssv_conv_desc* d, int groups, const char* who) {
  SSV_REQUIRE(groups >= 1 && d->C % groups == 0 && d->K % groups == 0, "%s: groups = %d must divide C = %d and K = %d", who, groups, d->C, d->K);
  return SSV_OK;
}
}  // namespace

extern "C" int ssv_conv2d_fwd_grouped(const ssv_conv_desc* d, int32_t groups, const float* x, const float* wd, const float* bias,
                                      const float* addend, float* y, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_grouped")) return rc;
  if (int rc = check_groups(d, groups, "ssv_conv2d_fwd_grouped")) return rc;
  SSV_REQUIRE(x && wd && y, "ssv_conv2d_fwd_grouped: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)wd | (uintptr_t)y) & 15) == 0, "ssv_conv2d_fwd_grouped: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  launch_fwd(d, x, wd, bias, addend, y, nullptr, nullptr, nullptr, nullptr, s, nullptr, 0, 0, groups);
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_grouped");
  return SSV_OK;
}

namespace {
int check_gate(const ssv_bn_gate* g, const char* who) {
  SSV_REQUIRE(g && g->x && g->mean && g->invstd && g->psum_g && g->psum_gx, "%s: incomplete gate", who);
  SSV_REQUIRE((g->mask != nullptr) != (g->scale != nullptr && g->shift != nullptr) && (g->scale == nullptr) == (g->shift == nullptr),
              "%s: a gate carries either the byte mask or scale + shift", who);
  SSV_REQUIRE((((uintptr_t)g->x | (uintptr_t)g->mean | (uintptr_t)g->invstd | (uintptr_t)g->psum_g | (uintptr_t)g->psum_gx | (uintptr_t)g->scale | (uintptr_t)g->shift) & 15) == 0,
              "%s: gate pointers must be 16-byte aligned", who);
  if (g->x2 || g->mean2 || g->invstd2 || g->psum_gx2) {
    SSV_REQUIRE(g->x2 && g->mean2 && g->invstd2 && g->psum_gx2 && g->mask, "%s: the second reduction target needs x2, mean2, invstd2, psum_gx2 and the byte-mask gate", who);
    SSV_REQUIRE((((uintptr_t)g->x2 | (uintptr_t)g->mean2 | (uintptr_t)g->invstd2 | (uintptr_t)g->psum_gx2) & 15) == 0, "%s: gate pointers must be 16-byte aligned", who);
  }
  return SSV_OK;
}
}  // namespace

extern "C" int64_t ssv_conv2d_fwd_gate_groups(const ssv_conv_desc* d) {
  if (!d || d->K <= 0) return 0;
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  const int bm = (d->K >= 128 || sp_fwd_ok(d)) ? 128 : 256;           // rows of the launch's tile (the bf16-piece variants: 128 at every width)
  return cdiv64(M, bm) * (bm / 64);
}

extern "C" int ssv_conv2d_fwd_gated(const ssv_conv_desc* d, const float* x, const float* w, const float* addend, float* y,
                                    const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_gated")) return rc;
  if (int rc = check_gate(gate, "ssv_conv2d_fwd_gated")) return rc;
  SSV_REQUIRE(x && w && y, "ssv_conv2d_fwd_gated: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)addend) & 15) == 0, "ssv_conv2d_fwd_gated: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K % 4 == 0, "ssv_conv2d_fwd_gated: needs C %% 32 == 0 and K %% 4 == 0 (got C=%d K=%d)", d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  launch_fwd(d, x, w, nullptr, addend, y, nullptr, nullptr, nullptr, nullptr, s, gate);
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_gated");
  return SSV_OK;
}

// ssv_conv2d_fwd_gated whose addend is the COMPACT data gradient of a stride-2 projection shortcut: addend [N][H2][W2][K] belongs to the output
// pixels with even (h, w) (H2 = ceil(Ho / 2), W2 = ceil(Wo / 2)); every other pixel gets no addend.  K >= 128, byte-mask gate.
extern "C" int ssv_conv2d_fwd_gated_s2add(const ssv_conv_desc* d, const float* x, const float* w, const float* addend, int32_t H2, int32_t W2, float* y,
                                          const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_gated_s2add")) return rc;
  if (int rc = check_gate(gate, "ssv_conv2d_fwd_gated_s2add")) return rc;
  SSV_REQUIRE(x && w && y && addend, "ssv_conv2d_fwd_gated_s2add: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)addend) & 15) == 0, "ssv_conv2d_fwd_gated_s2add: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K % 4 == 0 && d->K >= 128 && gate->mask, "ssv_conv2d_fwd_gated_s2add: needs C %% 32 == 0, K %% 4 == 0, K >= 128 and the byte-mask gate (got C=%d K=%d)", d->C, d->K);
  SSV_REQUIRE(H2 == (d->Ho + 1) / 2 && W2 == (d->Wo + 1) / 2, "ssv_conv2d_fwd_gated_s2add: the compact addend must be ceil(Ho/2) x ceil(Wo/2) (got %d x %d for %d x %d)", H2, W2, d->Ho, d->Wo);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  launch_fwd(d, x, w, nullptr, addend, y, nullptr, nullptr, nullptr, nullptr, s, gate, H2, W2);
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_gated_s2add");
  return SSV_OK;
}

// 1x1 / stride-1 forward convolution whose input operand is the BatchNorm backward's dx, formed on load (the data gradient of a 1x1
// convolution runs here with the transposed filter); optional gate on the output as in ssv_conv2d_fwd_gated.
namespace {
int fwd_dyin_impl(const ssv_conv_desc* d, const float* g, const ssv_bn_dyin* dyin, const float* w, const float* addend, int add_H2, int add_W2, float* y,
                  const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_dyin")) return rc;
  if (gate) { if (int rc = check_gate(gate, "ssv_conv2d_fwd_dyin")) return rc; }
  SSV_REQUIRE(g && w && y && dyin && dyin->x && dyin->coef, "ssv_conv2d_fwd_dyin: null pointer");
  SSV_REQUIRE((((uintptr_t)g | (uintptr_t)w | (uintptr_t)y | (uintptr_t)addend | (uintptr_t)dyin->x | (uintptr_t)dyin->coef) & 15) == 0,
              "ssv_conv2d_fwd_dyin: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->C % 32 == 0 && d->K % 4 == 0,
              "ssv_conv2d_fwd_dyin: a 1x1 / stride-1 / unpadded convolution with C %% 32 == 0 and K %% 4 == 0 (got %dx%d s%d p%d C=%d K=%d)",
              d->R, d->S, d->stride, d->pad, d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.dyin_x = dyin->x; p.dyin_coef = dyin->coef;
  p.add_H2 = add_H2; p.add_W2 = add_W2;
  if (gate) p.gate = *gate;
  const bool wide = d->K >= 128;
  const int sp = sp_fwd_mode(d);
  const unsigned grid = (unsigned)(wide ? cdiv(p.M, 128) * cdiv(d->K, 128) : (sp ? cdiv(p.M, 128) : cdiv(p.M, 256)) * cdiv(d->K, 64));
  const int gm = gate ? (gate->x2 ? 3 : (gate->mask ? 2 : 1)) : 0;
#define FWDD_(BM_, BN_, WM_, WN_, G_, SP_) \
  hipLaunchKernelGGL((conv_fwd_k<BM_, BN_, WM_, WN_, 32, true, false, false, false, false, G_, 1, false, false, SP_>), dim3(grid), dim3(256), 0, s, p, g, w, (const float*)nullptr, addend, y)
#define FWDD(BM_, BN_, WM_, WN_, G_) FWDD_(BM_, BN_, WM_, WN_, G_, 0)
#define FWDD_TILE(G_) do { if (sp == 2) { if (wide) FWDD_(128, 128, 2, 2, G_, 2); else FWDD_(128, 64, 2, 2, G_, 2); } \
                           else if (sp) { if (wide) FWDD_(128, 128, 2, 2, G_, 1); else FWDD_(128, 64, 2, 2, G_, 1); } \
                           else    { if (wide) FWDD(128, 128, 2, 2, G_); else FWDD(256, 64, 4, 1, G_); } } while (0)
  if (add_H2 > 0) {                                          // compact stride-2 addend: wide tile and byte-mask gate checked by the caller
#define FWDDS_(G_, SP_) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, false, false, false, false, G_, 1, true, false, SP_>), dim3(grid), dim3(256), 0, s, p, g, w, (const float*)nullptr, addend, y)
#define FWDDS(G_) do { if (sp == 2) FWDDS_(G_, 2); else if (sp) FWDDS_(G_, 1); else FWDDS_(G_, 0); } while (0)
    if (gm == 3) FWDDS(3); else FWDDS(2);
#undef FWDDS
#undef FWDDS_
  } else if (gm == 3) FWDD_TILE(3); else if (gm == 2) FWDD_TILE(2); else if (gm == 1) FWDD_TILE(1); else FWDD_TILE(0);
#undef FWDD_TILE
#undef FWDD
#undef FWDD_
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_dyin");
  return SSV_OK;
}
}  // namespace

extern "C" int ssv_conv2d_fwd_dyin(const ssv_conv_desc* d, const float* g, const ssv_bn_dyin* dyin, const float* w, const float* addend, float* y,
                                   const ssv_bn_gate* gate, void* stream) {
  return fwd_dyin_impl(d, g, dyin, w, addend, 0, 0, y, gate, stream);
}

// ssv_conv2d_fwd_dyin with the compact stride-2 addend of ssv_conv2d_fwd_gated_s2add (K >= 128, byte-mask gate required)
extern "C" int ssv_conv2d_fwd_dyin_s2add(const ssv_conv_desc* d, const float* g, const ssv_bn_dyin* dyin, const float* w, const float* addend,
                                         int32_t H2, int32_t W2, float* y, const ssv_bn_gate* gate, void* stream) {
  SSV_REQUIRE(d && gate && gate->mask && addend && d->K >= 128, "ssv_conv2d_fwd_dyin_s2add: needs the compact addend, the byte-mask gate and K >= 128");
  SSV_REQUIRE(H2 == (d->Ho + 1) / 2 && W2 == (d->Wo + 1) / 2, "ssv_conv2d_fwd_dyin_s2add: the compact addend must be ceil(Ho/2) x ceil(Wo/2) (got %d x %d for %d x %d)", H2, W2, d->Ho, d->Wo);
  return fwd_dyin_impl(d, g, dyin, w, addend, H2, W2, y, gate, stream);
}

// 1x1 / stride-1 forward convolution (with the statistics epilogue) whose input is the closing activation of the previous residual unit,
// formed on load AND written out: a = relu(x * scale + shift + (res | res * rscale + rshift)), relu mask optional.
extern "C" int ssv_conv2d_fwd_sumin_stats(const ssv_conv_desc* d, const float* x, const float* res, const float* scale, const float* shift,
                                          const float* rscale, const float* rshift, const float* w, float* y, float* pmean, float* pm2,
                                          float* a_out, uint8_t* mask_out, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_sumin_stats")) return rc;
  SSV_REQUIRE(x && res && scale && shift && w && y && pmean && pm2 && a_out, "ssv_conv2d_fwd_sumin_stats: null pointer");
  SSV_REQUIRE((rscale == nullptr) == (rshift == nullptr), "ssv_conv2d_fwd_sumin_stats: rscale / rshift must both be given or both NULL");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)res | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)rscale | (uintptr_t)rshift | (uintptr_t)w | (uintptr_t)y |
                (uintptr_t)pmean | (uintptr_t)pm2 | (uintptr_t)a_out) & 15) == 0, "ssv_conv2d_fwd_sumin_stats: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->C % 32 == 0 && d->K % 4 == 0,
              "ssv_conv2d_fwd_sumin_stats: a 1x1 / stride-1 / unpadded convolution with C %% 32 == 0 and K %% 4 == 0 (got %dx%d s%d p%d C=%d K=%d)",
              d->R, d->S, d->stride, d->pad, d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.aux_out = pmean; p.aux_out2 = pm2;
  p.sum_res = res; p.sum_scale = scale; p.sum_shift = shift; p.sum_rscale = rscale; p.sum_rshift = rshift; p.sum_out = a_out; p.sum_mask = mask_out;
  // (a 128 x 64 tile with K-step 64 - 256-byte row pieces - is 6 % faster for the 256 -> 64 conv1 of the 56x56 stage, r03 x1, but its statistics
  // epilogue sums each 64-row group in another order: the forward would no longer be bit-identical to bn_apply + conv2d_fwd_stats.  Not taken.)
  const bool wide = d->K >= 128;
  const int sp = sp_fwd_mode(d);
  const unsigned grid = wide ? (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128)) : (unsigned)((sp ? cdiv(p.M, 128) : cdiv(p.M, 256)) * cdiv(d->K, 64));
#define FWDS(BM_, BN_, WM_, WN_, SP_) hipLaunchKernelGGL((conv_fwd_k<BM_, BN_, WM_, WN_, 32, true, false, true, false, false, 0, 2, false, false, SP_>), dim3(grid), dim3(256), 0, s, p, x, w, (const float*)nullptr, (const float*)nullptr, y)
  if (sp == 2) { if (wide) FWDS(128, 128, 2, 2, 2); else FWDS(128, 64, 2, 2, 2); }
  else if (sp) { if (wide) FWDS(128, 128, 2, 2, 1); else FWDS(128, 64, 2, 2, 1); }
  else    { if (wide) FWDS(128, 128, 2, 2, 0); else FWDS(256, 64, 4, 1, 0); }
#undef FWDS
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_sumin_stats");
  return SSV_OK;
}

// Forward convolution that also leaves the BatchNorm statistics partials of its output: group g = output rows [64 g, 64 g + 64)
extern "C" int64_t ssv_conv2d_fwd_stats_groups(const ssv_conv_desc* d) {
  return d ? cdiv64((int64_t)d->N * d->Ho * d->Wo, 64) : 0;
}

extern "C" int ssv_conv2d_fwd_stats(const ssv_conv_desc* d, const float* x, const float* w, float* y, float* pmean, float* pm2, void* stream) {
  return ssv_conv2d_fwd_bnrelu_in_stats(d, x, nullptr, nullptr, w, y, pmean, pm2, stream);
}

extern "C" int ssv_conv2d_fwd_bnrelu_in_stats(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift,
                                              const float* w, float* y, float* pmean, float* pm2, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd_bnrelu_in_stats")) return rc;
  SSV_REQUIRE(x && w && y, "ssv_conv2d_fwd_bnrelu_in_stats: null pointer");
  SSV_REQUIRE((pmean == nullptr) == (pm2 == nullptr), "ssv_conv2d_fwd_bnrelu_in_stats: pmean / pm2 must both be given or both NULL");
  SSV_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "ssv_conv2d_fwd_bnrelu_in_stats: in_scale / in_shift must both be given or both NULL");
  SSV_REQUIRE(pmean || in_scale, "ssv_conv2d_fwd_bnrelu_in_stats: neither statistics nor a fused input requested - call ssv_conv2d_fwd");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)pmean | (uintptr_t)pm2 | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0,
              "ssv_conv2d_fwd_bnrelu_in_stats: pointers must be 16-byte aligned");
  SSV_REQUIRE((d->C % 32 == 0 || (d->C == 4 && !in_scale)) && d->K % 4 == 0,
              "ssv_conv2d_fwd_bnrelu_in_stats: needs C %% 32 == 0 (or the 4-channel stem, statistics only) and K %% 4 == 0 (got C=%d K=%d)", d->C, d->K);
  SSV_REQUIRE(!in_scale || d->C <= XF_MAXC, "ssv_conv2d_fwd_bnrelu_in_stats: a fused input supports at most %d channels (got %d)", XF_MAXC, d->C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  launch_fwd(d, x, w, nullptr, nullptr, y, pmean, pm2, in_scale, in_shift, s);
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd_bnrelu_in_stats");
  return SSV_OK;
}

// Linear + GELU with both tensors kept (the pre-activation h for the backward, gelu(h) for the next layer): one GEMM, one
// epilogue; and its counterpart, dgrad with the GELU derivative applied to the product before the addend.
extern "C" int ssv_linear_gelu_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias, float* h, float* act, void* stream) {
  if (int rc = check_desc(d, "ssv_linear_gelu_fwd")) return rc;
  SSV_REQUIRE(x && w && act, "ssv_linear_gelu_fwd: null pointer");      // h == NULL: the pre-activation is not kept (no backward will read it)
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)h | (uintptr_t)act) & 15) == 0, "ssv_linear_gelu_fwd: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K >= 128 && d->K % 4 == 0, "ssv_linear_gelu_fwd: needs C %% 32 == 0, K >= 128, K %% 4 == 0 (got C=%d K=%d)", d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.aux_out = act;
  const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128));
  // K-step 32 only: its LDS stage is what the vectorised epilogue (the one that writes the second tensor) needs
#ifdef SSV_EXP_S2
  if (p.RSC / 32 >= SSV_EXP_S2) {
    if (h) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 1, false, 0, false, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, h);
    else   hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 3, false, 0, false, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, act);
  } else
#endif
#define FWDE(E_, SP_, OUT_) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, E_, false, 0, false, 0, 0, false, false, SP_>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, OUT_)
  const int sp = sp_fwd_mode(d);
  if (sp == 2)  { if (h) FWDE(1, 2, h); else FWDE(3, 2, act); }
  else if (sp)  { if (h) FWDE(1, 1, h); else FWDE(3, 1, act); }
  else          { if (h) FWDE(1, 0, h); else FWDE(3, 0, act); }
#undef FWDE
  SSV_CHECK_LAUNCH("ssv_linear_gelu_fwd");
  return SSV_OK;
}

extern "C" int ssv_conv2d_dgrad_gelu(const ssv_conv_desc* d, const float* dy, const float* w, const float* h, const float* addend,
                                     float* dx, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_dgrad_gelu")) return rc;
  SSV_REQUIRE(dy && w && h && dx, "ssv_conv2d_dgrad_gelu: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)h | (uintptr_t)dx) & 15) == 0, "ssv_conv2d_dgrad_gelu: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % 32 == 0 && d->C % 4 == 0 && d->C >= 128 && d->stride == 1 && d->R == 1 && d->S == 1,
              "ssv_conv2d_dgrad_gelu: a Linear layer with K %% 32 == 0, C %% 4 == 0, C >= 128 (got K=%d C=%d)", d->K, d->C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_DGRAD, s);
  ConvKP p = make_kp(d);
  p.aux_in = h;
  const int64_t Mc = (int64_t)d->N * d->H * d->W;
  const unsigned gx = (unsigned)(cdiv64(Mc, 128) * cdiv(d->C, 128));
  hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2, 32, true>), dim3(gx, 1), dim3(256), 0, s, p, dy, w, addend, dx);
  SSV_CHECK_LAUNCH("ssv_conv2d_dgrad_gelu");
  return SSV_OK;
}

// The same product on the FORWARD kernel: dh = (dy wt^T) * gelu'(h) (+ addend) with wt = w transposed ([C][K], ssv_filter_transpose), so both
// operands are k-contiguous rows (ds_read_b128) as in every other stride-1 data gradient - the dgrad kernel reads the in-place weight image k-major.
// `d` describes the GEMM as launched: N*H*W rows of dy with d->C columns (the Linear's outputs), d->K result columns (its inputs, = h's width).
extern "C" int ssv_linear_fwd_gelugrad(const ssv_conv_desc* d, const float* dy, const float* wt, const float* h, const float* addend,
                                       float* dh, void* stream) {
  if (int rc = check_desc(d, "ssv_linear_fwd_gelugrad")) return rc;
  SSV_REQUIRE(dy && wt && h && dh, "ssv_linear_fwd_gelugrad: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)h | (uintptr_t)dh | (uintptr_t)addend) & 15) == 0, "ssv_linear_fwd_gelugrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K % 4 == 0 && d->K >= 128 && d->stride == 1 && d->R == 1 && d->S == 1 && d->pad == 0,
              "ssv_linear_fwd_gelugrad: a Linear layer with C %% 32 == 0, K %% 4 == 0, K >= 128 (got C=%d K=%d)", d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_DGRAD, s);
  ConvKP p = make_kp(d);
  p.aux_in = h;
  const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128));
#ifdef SSV_EXP_S2
  if (p.RSC / 32 >= SSV_EXP_S2)
    hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 2, false, 0, false, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  else
#endif
  if (sp_fwd_mode(d) == 2) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 2, false, 0, false, 0, 0, false, false, 2>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  else if (sp_fwd_mode(d) == 1) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 2, false, 0, false, 0, 0, false, false, 1>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  else hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 2>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  SSV_CHECK_LAUNCH("ssv_linear_fwd_gelugrad");
  return SSV_OK;
}

// The pair with the derivative taken in the FORWARD (round 4): fc1's epilogue writes dact = gelu'(h) in the pre-activation's place (and act = gelu(h)); the backward of
// fc2 then multiplies the product by the stored factor - no erf / exp in the backward epilogue, and the forward's cdf serves both tensors.  Same bits as the pair above.
extern "C" int ssv_linear_gelu_fwd_dact(const ssv_conv_desc* d, const float* x, const float* w, const float* bias, float* dact, float* act, void* stream) {
  if (int rc = check_desc(d, "ssv_linear_gelu_fwd_dact")) return rc;
  SSV_REQUIRE(x && w && dact && act, "ssv_linear_gelu_fwd_dact: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)dact | (uintptr_t)act) & 15) == 0, "ssv_linear_gelu_fwd_dact: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K >= 128 && d->K % 4 == 0, "ssv_linear_gelu_fwd_dact: needs C %% 32 == 0, K >= 128, K %% 4 == 0 (got C=%d K=%d)", d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.aux_out = act;
  const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128));
  if (sp_fwd_mode(d) == 2) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 4, false, 0, false, 0, 0, false, false, 2>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, dact);
  else if (sp_fwd_mode(d) == 1) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 4, false, 0, false, 0, 0, false, false, 1>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, dact);
  else hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 4>), dim3(grid), dim3(256), 0, s, p, x, w, bias, (const float*)nullptr, dact);
  SSV_CHECK_LAUNCH("ssv_linear_gelu_fwd_dact");
  return SSV_OK;
}

extern "C" int ssv_linear_fwd_mulgrad(const ssv_conv_desc* d, const float* dy, const float* wt, const float* dact, const float* addend,
                                      float* dh, void* stream) {
  if (int rc = check_desc(d, "ssv_linear_fwd_mulgrad")) return rc;
  SSV_REQUIRE(dy && wt && dact && dh, "ssv_linear_fwd_mulgrad: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)dact | (uintptr_t)dh | (uintptr_t)addend) & 15) == 0, "ssv_linear_fwd_mulgrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C % 32 == 0 && d->K % 4 == 0 && d->K >= 128 && d->stride == 1 && d->R == 1 && d->S == 1 && d->pad == 0,
              "ssv_linear_fwd_mulgrad: a Linear layer with C %% 32 == 0, K %% 4 == 0, K >= 128 (got C=%d K=%d)", d->C, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.aux_in = dact;
  const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128));
  if (sp_fwd_mode(d) == 2) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 5, false, 0, false, 0, 0, false, false, 2>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  else if (sp_fwd_mode(d) == 1) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 5, false, 0, false, 0, 0, false, false, 1>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  else hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, 5>), dim3(grid), dim3(256), 0, s, p, dy, wt, (const float*)nullptr, addend, dh);
  SSV_CHECK_LAUNCH("ssv_linear_fwd_mulgrad");
  return SSV_OK;
}

namespace {
int dgrad_impl(const ssv_conv_desc* d, int groups, const float* dy, const float* w, const float* addend, float* dx, void* stream);
}
extern "C" int ssv_conv2d_dgrad(const ssv_conv_desc* d, const float* dy, const float* w, const float* addend,
                                float* dx, void* stream) {
  return dgrad_impl(d, 0, dy, w, addend, dx, stream);
}

// the parity-class data gradient of a grouped convolution on its dense block-diagonal bank (see ssv_conv2d_fwd_grouped)
extern "C" int ssv_conv2d_dgrad_grouped(const ssv_conv_desc* d, int32_t groups, const float* dy, const float* wd, const float* addend,
                                        float* dx, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_dgrad_grouped")) return rc;
  if (int rc = check_groups(d, groups, "ssv_conv2d_dgrad_grouped")) return rc;
  return dgrad_impl(d, groups, dy, wd, addend, dx, stream);
}

namespace {
int dgrad_impl(const ssv_conv_desc* d, int groups, const float* dy, const float* w, const float* addend, float* dx, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_dgrad")) return rc;
  SSV_REQUIRE(dy && w && dx, "ssv_conv2d_dgrad: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0, "ssv_conv2d_dgrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % 16 == 0 && d->C % 4 == 0, "ssv_conv2d_dgrad: needs K %% 16 == 0 and C %% 4 == 0 (got K=%d C=%d)", d->K, d->C);
  SSV_REQUIRE(d->stride <= 8, "ssv_conv2d_dgrad: stride too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_DGRAD, s);
  ConvKP p = make_kp(d);
  if (groups > 1) { p.Cg = d->C / groups; p.Kg = d->K / groups; }
  const bool bk32 = d->K % 32 == 0;
  const int st = d->stride;
  const int Hq = cdiv(d->H, st), Wq = cdiv(d->W, st);            // class (0,0) is the largest
  const int64_t Mc = (int64_t)d->N * Hq * Wq;
  if (d->C >= 128 && groups <= 1) {
    const dim3 g((unsigned)(cdiv64(Mc, 128) * cdiv(d->C, 128)), st * st);
    if (bk32 && sp_dgrad_ok(d)) hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2, 32, false, 0, true>), g, dim3(256), 0, s, p, dy, w, addend, dx);
    else if (bk32) hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2, 32>), g, dim3(256), 0, s, p, dy, w, addend, dx);
    else      hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2, 16>), g, dim3(256), 0, s, p, dy, w, addend, dx);
  } else {
    const dim3 g((unsigned)(cdiv64(Mc, 256) * cdiv(d->C, 64)), st * st);
    if (bk32) hipLaunchKernelGGL((conv_dgrad_k<256, 64, 4, 1, 32>), g, dim3(256), 0, s, p, dy, w, addend, dx);
    else      hipLaunchKernelGGL((conv_dgrad_k<256, 64, 4, 1, 16>), g, dim3(256), 0, s, p, dy, w, addend, dx);
  }
  SSV_CHECK_LAUNCH("ssv_conv2d_dgrad");
  return SSV_OK;
}
}  // namespace

extern "C" int64_t ssv_conv2d_dgrad_gate_groups(const ssv_conv_desc* d) {
  if (!d || d->C <= 0 || d->stride <= 0) return 0;
  const int st = d->stride;
  const int64_t Mc = (int64_t)d->N * cdiv(d->H, st) * cdiv(d->W, st);
  const int bm = d->C >= 128 ? 128 : 256;
  return (int64_t)st * st * cdiv64(Mc, bm) * (bm / 64);
}

extern "C" int ssv_conv2d_dgrad_gated(const ssv_conv_desc* d, const float* dy, const float* w, const float* addend, float* dx,
                                      const ssv_bn_gate* gate, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_dgrad_gated")) return rc;
  if (int rc = check_gate(gate, "ssv_conv2d_dgrad_gated")) return rc;
  SSV_REQUIRE(gate->x2 == nullptr, "ssv_conv2d_dgrad_gated: the second reduction target is a forward-kernel feature (stride-1 data gradients)");
  SSV_REQUIRE(dy && w && dx, "ssv_conv2d_dgrad_gated: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)addend) & 15) == 0, "ssv_conv2d_dgrad_gated: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % 32 == 0 && d->C % 4 == 0 && d->stride <= 8, "ssv_conv2d_dgrad_gated: needs K %% 32 == 0, C %% 4 == 0, stride <= 8 (got K=%d C=%d)", d->K, d->C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_DGRAD, s);
  ConvKP p = make_kp(d);
  p.gate = *gate;
  const int st = d->stride;
  const int64_t Mc = (int64_t)d->N * cdiv(d->H, st) * cdiv(d->W, st);
#define DG(BM_, BN_, WM_, WN_, G_) hipLaunchKernelGGL((conv_dgrad_k<BM_, BN_, WM_, WN_, 32, false, G_>), g, dim3(256), 0, s, p, dy, w, addend, dx)
#define DGS(G_) hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2, 32, false, G_, true>), g, dim3(256), 0, s, p, dy, w, addend, dx)
  if (d->C >= 128) {
    const dim3 g((unsigned)(cdiv64(Mc, 128) * cdiv(d->C, 128)), st * st);
    if (sp_dgrad_ok(d)) { if (gate->mask) DGS(2); else DGS(1); }
    else if (gate->mask) DG(128, 128, 2, 2, 2); else DG(128, 128, 2, 2, 1);
  } else {
    const dim3 g((unsigned)(cdiv64(Mc, 256) * cdiv(d->C, 64)), st * st);
    if (gate->mask) DG(256, 64, 4, 1, 2); else DG(256, 64, 4, 1, 1);
  }
#undef DG
#undef DGS
  SSV_CHECK_LAUNCH("ssv_conv2d_dgrad_gated");
  return SSV_OK;
}

extern "C" size_t ssv_conv2d_wgrad_workspace_bytes(const ssv_conv_desc* d) {
  if (!d || d->K <= 0 || d->C <= 0) return 0;
  const WgradPlan w = plan_wgrad(d);
  return (size_t)w.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
}

extern "C" size_t ssv_conv2d_wgrad_grouped_workspace_bytes(const ssv_conv_desc* d, int32_t groups) {
  if (!d || d->K <= 0 || d->C <= 0 || groups < 1 || d->C % groups || d->K % groups) return 0;
  const WgradPlan w = plan_wgrad(d, groups);
  return (size_t)w.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
}

extern "C" int ssv_conv2d_wgrad(const ssv_conv_desc* d, const float* x, const float* dy, float* dw,
                                int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return ssv_conv2d_wgrad_bnrelu_in(d, x, nullptr, nullptr, dy, dw, accumulate, ws, ws_bytes, stream);
}

namespace {
int wgrad_impl(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* dy, const ssv_bn_dyin* dyin,
               float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream, int groups = 0, float* dbias = nullptr);
}
// Weight AND bias gradient of a Linear / 1x1 / stride-1 / unpadded layer in one pass over dY (nn.Linear backward: dW = dY^T X, db = column sums
// of dY): the weight-gradient workgroups of column tile 0 sum the dY rows they stage anyway.  C % 4 == 0, K % 4 == 0.
extern "C" size_t ssv_conv2d_wgrad_bias_workspace_bytes(const ssv_conv_desc* d) {
  if (!d || d->K <= 0 || d->C <= 0) return 0;
  const WgradPlan w = plan_wgrad(d);
  return (size_t)w.nsplit * d->K * ((size_t)d->R * d->S * d->C + 1) * sizeof(float);
}
extern "C" int ssv_conv2d_wgrad_bias(const ssv_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                                     void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(d && dbias, "ssv_conv2d_wgrad_bias: null pointer");
  SSV_REQUIRE(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->K % 4 == 0 && d->C % 4 == 0,
              "ssv_conv2d_wgrad_bias: a 1x1 / stride-1 / unpadded layer with K %% 4 == 0 and C %% 4 == 0 (got R=%d stride=%d pad=%d K=%d C=%d)",
              d->R, d->stride, d->pad, d->K, d->C);
  SSV_REQUIRE(((uintptr_t)dbias & 15) == 0, "ssv_conv2d_wgrad_bias: pointers must be 16-byte aligned");
  return wgrad_impl(d, x, nullptr, nullptr, dy, nullptr, dw, accumulate, ws, ws_bytes, stream, 0, dbias);
}
// Weight gradient of a grouped convolution in the layout of its dense block-diagonal bank, dwd [K][R][S][C]: ONLY the diagonal blocks (the entries
// ssv_group_extract reads) are defined - tiles that no group touches are skipped and their entries are whatever the workspace held.
extern "C" int ssv_conv2d_wgrad_grouped(const ssv_conv_desc* d, int32_t groups, const float* x, const float* dy, float* dwd, int accumulate,
                                        void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_wgrad_grouped")) return rc;
  if (int rc = check_groups(d, groups, "ssv_conv2d_wgrad_grouped")) return rc;
  return wgrad_impl(d, x, nullptr, nullptr, dy, nullptr, dwd, accumulate, ws, ws_bytes, stream, groups);
}
extern "C" int ssv_conv2d_wgrad_bnrelu_in(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* dy,
                                          float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return wgrad_impl(d, x, in_scale, in_shift, dy, nullptr, dw, accumulate, ws, ws_bytes, stream);
}

// Weight gradient of a 1x1 / stride-1 convolution whose dY operand is the BatchNorm backward's dx, formed on load from (g, dyin).
extern "C" int ssv_conv2d_wgrad_dyin(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* g,
                                     const ssv_bn_dyin* dyin, float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(dyin && dyin->x && dyin->coef, "ssv_conv2d_wgrad_dyin: null operand description");
  SSV_REQUIRE((((uintptr_t)dyin->x | (uintptr_t)dyin->coef) & 15) == 0, "ssv_conv2d_wgrad_dyin: pointers must be 16-byte aligned");
  SSV_REQUIRE(d && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->K % 4 == 0 && d->C % 4 == 0,
              "ssv_conv2d_wgrad_dyin: a 1x1 / stride-1 / unpadded convolution with K %% 4 == 0 and C %% 4 == 0");
  return wgrad_impl(d, x, in_scale, in_shift, g, dyin, dw, accumulate, ws, ws_bytes, stream);
}

namespace {
int wgrad_impl(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* dy, const ssv_bn_dyin* dyin,
               float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream, int groups, float* dbias) {
  if (int rc = check_desc(d, "ssv_conv2d_wgrad")) return rc;
  SSV_REQUIRE(x && dy && dw && ws, "ssv_conv2d_wgrad: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)ws | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15) == 0, "ssv_conv2d_wgrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % 4 == 0, "ssv_conv2d_wgrad: needs K %% 4 == 0 (got %d)", d->K);
  SSV_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "ssv_conv2d_wgrad: in_scale / in_shift must both be given or both NULL");
  const bool xf = in_scale != nullptr;
  SSV_REQUIRE(!xf || d->C % 4 == 0, "ssv_conv2d_wgrad: a fused input needs C %% 4 == 0 (got C=%d)", d->C);
  const WgradPlan wp = plan_wgrad(d, groups);
  const size_t slabs = (size_t)wp.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
  const size_t need = slabs + (dbias ? (size_t)wp.nsplit * d->K * sizeof(float) : 0);      // bias partials [nsplit][K] behind the slabs
  if (ws_bytes < need) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  ConvKP p = make_kp(d);
  p.xf_scale = in_scale; p.xf_shift = in_shift;
  if (dyin) { p.dyin_x = dyin->x; p.dyin_coef = dyin->coef; }
  if (groups > 1) { p.Cg = d->C / groups; p.Kg = d->K / groups; }
  if (dbias) p.aux_out = (float*)((char*)ws + slabs);
  const bool vecb = d->C % 4 == 0;
  const int tiles = wp.it * wp.jt;
  const dim3 grid((unsigned)(tiles * wp.nsplit));
  float* part = (float*)ws;
  // gather mode of the X operand (see conv_wgrad_k): LIN, S1 (needs one carry per K-step: BK/Wo + 1 <= Ho, Ho == H, Wo == W) or generic
  const bool s1ok = d->stride == 1 && d->Ho == d->H && d->Wo == d->W && 32 / d->Wo + 1 <= d->Ho;
  const int gather = (d->R == 1 && d->S == 1 && d->pad == 0 && d->stride == 1) ? 1 : (s1ok ? 2 : 0);
  const bool sp = sp_wgrad_ok(d, groups);
#define WG_LAUNCH_(BM_, BN_, WM_, WN_, B_, V_, G_, X_, SP_) \
  hipLaunchKernelGGL((conv_wgrad_k<BM_, BN_, WM_, WN_, B_, V_, G_, X_, false, false, 0, SP_>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk, tiles)
#define WG_LAUNCH(BM_, BN_, WM_, WN_, B_, V_, G_, X_) WG_LAUNCH_(BM_, BN_, WM_, WN_, B_, V_, G_, X_, false)
#define WG_LAUNCH_A(BM_, BN_, WM_, WN_, G_, X_) do { if (sp) WG_LAUNCH_(BM_, BN_, WM_, WN_, 32, true, G_, X_, true); else WG_LAUNCH_(BM_, BN_, WM_, WN_, 32, true, G_, X_, false); } while (0)
#define WG_DYIN_(BM_, BN_, WM_, WN_, X_, SP_) \
  hipLaunchKernelGGL((conv_wgrad_k<BM_, BN_, WM_, WN_, 32, true, 1, X_, true, false, 0, SP_>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk, tiles)
#define WG_DYIN(BM_, BN_, WM_, WN_, X_) do { if (sp) WG_DYIN_(BM_, BN_, WM_, WN_, X_, true); else WG_DYIN_(BM_, BN_, WM_, WN_, X_, false); } while (0)
#define WG_GATHER(BM_, BN_, WM_, WN_, X_) \
  do { if (gather == 1) WG_LAUNCH_A(BM_, BN_, WM_, WN_, 1, X_); else if (gather == 2) WG_LAUNCH_A(BM_, BN_, WM_, WN_, 2, X_); \
       else WG_LAUNCH_A(BM_, BN_, WM_, WN_, 0, X_); } while (0)
#define WG_BIAS_(BM_, BN_, WM_, WN_, SP_) \
  hipLaunchKernelGGL((conv_wgrad_k<BM_, BN_, WM_, WN_, 32, true, 1, false, false, true, 0, SP_>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk, tiles)
#define WG_BIAS(BM_, BN_, WM_, WN_) do { if (sp) WG_BIAS_(BM_, BN_, WM_, WN_, true); else WG_BIAS_(BM_, BN_, WM_, WN_, false); } while (0)
  if (dbias) {                                     // preconditions checked by ssv_conv2d_wgrad_bias: LIN gather, float4 columns
    if (wp.bm == 128) { if (wp.bn == 64) WG_BIAS(128, 64, 2, 2); else WG_BIAS(128, 128, 2, 2); }
    else              { if (wp.bn == 64) WG_BIAS(64, 64, 2, 2);  else WG_BIAS(64, 128, 1, 4); }
  } else if (dyin) {                                      // preconditions checked by ssv_conv2d_wgrad_dyin: LIN gather, float4 columns
    if (wp.bm == 128) {
      if (wp.bn == 64) { if (xf) WG_DYIN(128, 64, 2, 2, true); else WG_DYIN(128, 64, 2, 2, false); }
      else             { if (xf) WG_DYIN(128, 128, 2, 2, true); else WG_DYIN(128, 128, 2, 2, false); }
    } else {
      if (wp.bn == 64) { if (xf) WG_DYIN(64, 64, 2, 2, true); else WG_DYIN(64, 64, 2, 2, false); }
      else             { if (xf) WG_DYIN(64, 128, 1, 4, true); else WG_DYIN(64, 128, 1, 4, false); }
    }
  } else if (!vecb) {
    if (wp.bm == 128) WG_LAUNCH(128, 128, 2, 2, GBK, false, 0, false);
    else              WG_LAUNCH(64, 128, 1, 4, GBK, false, 0, false);
  } else if (wp.bn == 64) {                       // RSC <= 64 (and C % 4 == 0): 64-wide column tile
    if (wp.bm == 128) { if (xf) WG_GATHER(128, 64, 2, 2, true); else WG_GATHER(128, 64, 2, 2, false); }
    else              { if (xf) WG_GATHER(64, 64, 2, 2, true); else WG_GATHER(64, 64, 2, 2, false); }
  } else if (wp.bm == 128) {
    if (xf) WG_GATHER(128, 128, 2, 2, true); else WG_GATHER(128, 128, 2, 2, false);
  } else {
    if (xf) WG_GATHER(64, 128, 1, 4, true); else WG_GATHER(64, 128, 1, 4, false);
  }
#undef WG_GATHER
#undef WG_DYIN
#undef WG_DYIN_
#undef WG_BIAS
#undef WG_BIAS_
#undef WG_LAUNCH
#undef WG_LAUNCH_A
#undef WG_LAUNCH_
  SSV_CHECK_LAUNCH("ssv_conv2d_wgrad(partial)");
  const int64_t n = (int64_t)d->K * p.RSC;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, (const float*)part, wp.nsplit, n, dw, accumulate);
  SSV_CHECK_LAUNCH("ssv_conv2d_wgrad(reduce)");
  if (dbias) {
    hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(d->K, 64)), dim3(256), 0, s, (const float*)p.aux_out, wp.nsplit, (int64_t)d->K, dbias, accumulate);
    SSV_CHECK_LAUNCH("ssv_conv2d_wgrad_bias(reduce)");
  }
  return SSV_OK;
}
}  // namespace

namespace {
// ---- the stem's forward from image ROWS staged in LDS (round 4; the weight gradient's twin is stem_wgrad_rows_k below) ---------------------------------------------
// A workgroup of seven waves walks pairs (RPI-tuples) of OUTPUT ROWS: the (RPI - 1) stride + R image rows they see go to LDS as they lie in memory (coalesced float4s,
// zeros left / right of a row and for rows outside the image), the whole filter sits in LDS beside them in step order, and the window element (pixel wo of output row
// rr, filter row r, float e) is the LDS float (rr stride + r) ROWF + off + 3 stride wo + e: wave w owns the 32 pixels 32 w .. of the RPI Wo <= 224 pixels and both
// 32-channel blocks; an MFMA step (two window floats e = 2 i, 2 i + 1 of one filter row; 3 S floats padded to an even count with zero weights) is one ds_read_b32 of the
// window at an IMMEDIATE offset of the lane's pixel address and two of the filter.  No gather, no masks, 154 MFMAs per wave between two barriers, 147 -> 154 contraction
// columns instead of the row-taps form's 168.  Statistics: every wave reduces its 32 pixels per channel to (mean, M2) in registers (two passes over its accumulators),
// the seven pairs are merged in LDS (Chan's update, equal counts) - one partial per RPI Wo pixels (ssv_stem_conv_fwd_stats_rows_per_group).
constexpr int SF_ROWS = 9, SF_ROWF = 728, SF_K = 64, SF_NT = 448, SF_MAXSTEPS = 77;
constexpr int SF_NX = (SF_ROWS * (SF_ROWF / 4) + SF_NT - 1) / SF_NT;

static inline int stem_fwd_rows_rpi(const ssv_conv_desc* d) {       // output rows per iteration, 0 = this launch runs on the row-taps kernel
#ifdef SSV_NO_STEM_ROWS
  return 0;
#endif
  const int padl = (d->pad * 3 + 3) / 4 * 4, off = padl - d->pad * 3;
  if (!(d->C == 3 && d->K == SF_K && d->R <= 7 && d->S * 3 <= 24 && d->R * ((d->S * 3 + 1) / 2) <= SF_MAXSTEPS && d->W % 4 == 0 &&
        padl + d->W * 3 <= SF_ROWF && off + (d->Wo - 1) * d->stride * 3 + 24 <= SF_ROWF && (int64_t)d->N * d->H * d->W * 12 < (1ll << 31)))
    return 0;
  int best = 0;
  for (int r = 1; r * d->Wo <= 224; ++r)
    if ((r * d->Wo) % 32 == 0 && d->Ho % r == 0 && (r - 1) * d->stride + d->R <= SF_ROWS) best = r;
  return best * d->Wo >= 192 ? best : 0;                             // at least six of the seven waves have pixels
}

template <bool STATS>
__global__ void __launch_bounds__(SF_NT, 2)
stem_fwd_rows_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ wrows, float* __restrict__ y, float* __restrict__ pmean, float* __restrict__ pm2,
                int rpi, int iters_per_wg) {
  __shared__ __attribute__((aligned(16))) float srows[SF_ROWS * SF_ROWF];
  __shared__ __attribute__((aligned(16))) float sw[SF_MAXSTEPS * 2 * SF_K];            // [step][half][channel]
  __shared__ float sst[STATS ? 2 * 7 * SF_K : 1];                                      // per wave and channel: mean | M2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int padl = (p.pad * 3 + 3) / 4 * 4, off = padl - p.pad * 3, s3 = p.stride * 3, S3 = 3 * p.S, spr = (S3 + 1) >> 1;
  const int w34 = p.W * 3 / 4, nrows = (rpi - 1) * p.stride + p.R, nx = nrows * w34, miter = rpi * p.Wo, nwav = miter >> 5;
  for (int i = tid; i < SF_ROWS * SF_ROWF; i += SF_NT) srows[i] = 0.f;
  for (int i = tid; i < p.R * spr * 2 * SF_K; i += SF_NT) {        // the filter in step order: step = r * spr + i, float e = 2 i + half of filter row r (zero past 3 S)
    const int n = i & (SF_K - 1), hs = i >> 6, h = hs & 1, st = hs >> 1, r = st / spr, e = 2 * (st - r * spr) + h;
    sw[i] = e < S3 ? wrows[(n * p.R + r) * 24 + e] : 0.f;
  }
  const int mi = 32 * wave + l31, rr = mi / p.Wo, wo = mi - rr * p.Wo;      // this lane's pixel of the iteration (waves past nwav idle)
  const int abase = rr * p.stride * SF_ROWF + off + s3 * wo + half;
  const int bbase = half * SF_K + l31;
  int xr[SF_NX], xj[SF_NX];
#pragma unroll
  for (int i = 0; i < SF_NX; ++i) { const int f = tid + SF_NT * i; xr[i] = f < nx ? f / w34 : -1; xj[i] = f - (f / w34) * w34; }
  const int64_t IT = (int64_t)p.N * p.Ho / rpi;
  const int64_t it0 = (int64_t)blockIdx.x * iters_per_wg, it1 = it0 + iters_per_wg < IT ? it0 + iters_per_wg : IT;
  f32x4 rx[SF_NX];
  auto load_rows = [&](int64_t it) {
    const int64_t g = it * rpi;
    const int n = (int)(g / p.Ho), ho = (int)(g - (int64_t)n * p.Ho);
#pragma unroll
    for (int i = 0; i < SF_NX; ++i) {
      const int hi = ho * p.stride - p.pad + xr[i];
      rx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (xr[i] >= 0 && (unsigned)hi < (unsigned)p.H) rx[i] = *reinterpret_cast<const f32x4*>(x + ((int64_t)n * p.H + hi) * p.W * 3 + 4 * xj[i]);
    }
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int i = 0; i < SF_NX; ++i)
      if (xr[i] >= 0) *reinterpret_cast<f32x4*>(&srows[xr[i] * SF_ROWF + padl + 4 * xj[i]]) = rx[i];
  };
  __syncthreads();                                              // zero fill and filter are in place
  if (it0 < it1) load_rows(it0);
  for (int64_t it = it0; it < it1; ++it) {
    store_rows();
    __syncthreads();
    if (it + 1 < it1) load_rows(it + 1);                        // the next iteration's image rows fly under this one's MFMAs
    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[tn][j] = 0.f;
    if (wave < nwav) {
      for (int r = 0; r < p.R; ++r) {
        const float* ar = srows + abase + r * SF_ROWF;
        const float* br = sw + bbase + r * spr * 2 * SF_K;
#pragma unroll 11
        for (int i = 0; i < spr; ++i) {
          const float a = ar[2 * i];
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, br[i * 2 * SF_K + 32 * tn], acc[tn], 0, 0, 0);
        }
      }
      const int64_t m0 = it * miter + 32 * wave;                // first pixel of this wave (output rows of an iteration are contiguous in y)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int j = 0; j < 16; ++j)       // (32 consecutive channels of a pixel per half-wave: 128-byte segments.  Timing what-if 2: no stores - 1.40 -> 1.27 ms at bs 512)
          if (!(SSV_WHATIF & 2) || acc[tn][j] == 12345.f) y[(m0 + (j & 3) + 8 * (j >> 2) + 4 * half) * SF_K + 32 * tn + l31] = acc[tn][j];
        if constexpr (STATS) {
          float sm = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) sm += acc[tn][j];
          sm += __shfl_xor(sm, 32, 64);
          const float mean = sm * (1.f / 32.f);
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) { const float dlt = acc[tn][j] - mean; q += dlt * dlt; }
          q += __shfl_xor(q, 32, 64);
          if (half == 0) { sst[wave * SF_K + 32 * tn + l31] = mean; sst[(7 + wave) * SF_K + 32 * tn + l31] = q; }
        }
      }
    }
    __syncthreads();                                            // every wave has read this iteration's rows; the per-wave statistics are in LDS
    if constexpr (STATS) {
      if (tid < SF_K) {
        float mean = 0.f, m2 = 0.f;
        for (int w = 0; w < nwav; ++w) mean += sst[w * SF_K + tid];
        mean /= (float)nwav;
        for (int w = 0; w < nwav; ++w) { const float dlt = sst[w * SF_K + tid] - mean; m2 += sst[(7 + w) * SF_K + tid] + 32.f * dlt * dlt; }
        pmean[it * SF_K + tid] = mean;
        pm2[it * SF_K + tid] = m2;
      }
    }
  }
}
}  // namespace

extern "C" int64_t ssv_stem_conv_fwd_stats_rows_per_group(const ssv_conv_desc* d) {
  if (!d) return 0;
  const int rpi = stem_fwd_rows_rpi(d);
  return rpi ? (int64_t)rpi * d->Wo : 64;
}

// ---- the 3-channel image stem on the UNPADDED image (networks/resnet.py:96-99, 147): row-taps form ------------------------------------------
// x [N][H][W][3]; wrows [K][R][24] = the filter's rows, 3 S floats each ((s, c) order = OHWI memory) zero-padded to 24; y [N][Ho][Wo][K];
// optional statistics partials: one (mean, M2) per ssv_stem_conv_fwd_stats_rows_per_group(d) output pixels (64, as ssv_conv2d_fwd_stats, on the row-taps kernel).  3 S <= 24, K % 4 == 0.
extern "C" int ssv_stem_conv_fwd(const ssv_conv_desc* d, const float* x, const float* wrows, float* y, float* pmean, float* pm2, void* stream) {
  if (int rc = check_desc(d, "ssv_stem_conv_fwd")) return rc;
  SSV_REQUIRE(x && wrows && y && (pmean == nullptr) == (pm2 == nullptr), "ssv_stem_conv_fwd: bad pointers");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)wrows | (uintptr_t)y | (uintptr_t)pmean | (uintptr_t)pm2) & 15) == 0, "ssv_stem_conv_fwd: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->C == 3 && d->S * 3 <= 24 && d->S <= 8 && d->K % 4 == 0, "ssv_stem_conv_fwd: a 3-channel input, at most 8 taps per filter row, K %% 4 == 0 (got C=%d S=%d K=%d)", d->C, d->S, d->K);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(d);
  p.aux_out = pmean; p.aux_out2 = pm2;
  if (const int rpi = stem_fwd_rows_rpi(d)) {                    // whole output rows from image rows staged in LDS
    const int64_t iters = (int64_t)d->N * d->Ho / rpi;
    const int ipw = (int)cdiv64(iters, 512);                     // one resident round of two workgroups per CU
    const unsigned g = (unsigned)cdiv64(iters, ipw);
    if (pmean) hipLaunchKernelGGL((stem_fwd_rows_k<true>), dim3(g), dim3(SF_NT), 0, s, p, x, wrows, y, pmean, pm2, rpi, ipw);
    else       hipLaunchKernelGGL((stem_fwd_rows_k<false>), dim3(g), dim3(SF_NT), 0, s, p, x, wrows, y, pmean, pm2, rpi, ipw);
    SSV_CHECK_LAUNCH("ssv_stem_conv_fwd(rows)");
    return SSV_OK;
  }
  const unsigned grid = (unsigned)(cdiv(p.M, 256) * cdiv(d->K, 64));
  if (pmean) hipLaunchKernelGGL((conv_fwd_k<256, 64, 4, 1, 24, true, false, true, 2>), dim3(grid), dim3(256), 0, s, p, x, wrows, (const float*)nullptr, (const float*)nullptr, y);
  else       hipLaunchKernelGGL((conv_fwd_k<256, 64, 4, 1, 24, true, false, false, 2>), dim3(grid), dim3(256), 0, s, p, x, wrows, (const float*)nullptr, (const float*)nullptr, y);
  SSV_CHECK_LAUNCH("ssv_stem_conv_fwd");
  return SSV_OK;
}

namespace {
// ---- the stem's weight gradient from image ROWS staged in LDS (round 4) -------------------------------------------------------------------------------------------
// conv_wgrad_k's row-taps gather (GATHER 3) stages a [32 pixels][R x 24] window tile per k-tile with six unaligned, masked 16-byte gathers per thread - index and
// mask arithmetic that keeps the kernel at 0.59 matrix-pipe busy (69 TFLOP/s inside the step, the slowest launch of the ResNet step).  Here a workgroup walks whole
// OUTPUT ROWS: the R image rows an output row sees go to LDS once, as they lie in memory (coalesced float4s; zeros left / right of the row and for rows outside the
// image), with the dY row beside them, and the window element (pixel wo, filter row r, float e) is the LDS float  r * ROWF + off + 3 * stride * wo + e : the B fragment
// of an MFMA step is one ds_read_b32 at an address that advances by a constant per pixel - no gather, no masks, Wo / 2 steps x 3 MFMAs per wave between two barriers.
// Output: one partial slab [K = 64][R x 24] per workgroup (columns e >= 3 S of a filter row are written as zeros), reduced by wgrad_reduce_k like the other form's.
constexpr int SR_R = 7, SR_ROWF = 728, SR_WO = 112, SR_K = 64;      // ROWF = 24 (mod 64): a column block that spans two filter rows reads 32 distinct banks
constexpr int SR_NX = (SR_R * (SR_ROWF / 4) + 255) / 256, SR_NDY = (SR_WO * SR_K / 4 + 255) / 256;

static inline bool stem_rows_ok(const ssv_conv_desc* d) {
#ifdef SSV_NO_STEM_ROWS
  return false;              // diagnostic builds: the gather form everywhere
#endif
  const int padl = (d->pad * 3 + 3) / 4 * 4, off = padl - d->pad * 3;
  return d->C == 3 && d->K == SR_K && d->R <= SR_R && d->S * 3 <= 24 && d->W % 4 == 0 && d->Wo % 2 == 0 && d->Wo <= SR_WO &&
         padl + d->W * 3 <= SR_ROWF && off + (d->Wo - 1) * d->stride * 3 + 24 <= SR_ROWF && (int64_t)d->N * d->H * d->W * 12 < (1ll << 31);
}

// (Measured and rejected: the real window floats only, packed as column 3 S r + e - 147 columns = five 32-column blocks on five waves instead of six on four:
//  1.51 ms against this form's 1.35 although it issues 17 % fewer MFMAs; profiles/r04_probe_stem_rows.txt.)
__global__ void __launch_bounds__(256, 3)
stem_wgrad_rows_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int rows_per_wg) {
  __shared__ __attribute__((aligned(16))) float srows[(SR_R + 1) * SR_ROWF];      // row SR_R stays zero: the columns past R x 24 read it
  __shared__ __attribute__((aligned(16))) float sdy[SR_WO * SR_K];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;                      // this wave: output channels 32 wm .., columns 96 wn ..
  const int padl = (p.pad * 3 + 3) / 4 * 4, off = padl - p.pad * 3, s3 = p.stride * 3;
  const int w34 = p.W * 3 / 4, nx = p.R * w34, ndy = p.Wo * SR_K / 4, NCOL = p.R * 24;
  for (int i = tid; i < (SR_R + 1) * SR_ROWF; i += 256) srows[i] = 0.f;
  int bbase[3];
#pragma unroll
  for (int tn = 0; tn < 3; ++tn) {
    const int col = 96 * wn + 32 * tn + l31, r = col / 24;
    bbase[tn] = (col < NCOL ? r * SR_ROWF + off + (col - 24 * r) : SR_R * SR_ROWF) + half * s3;
  }
  const int abase = half * SR_K + 32 * wm + l31;
  int xr[SR_NX], xj[SR_NX];                                    // this thread's float4s of the staged image rows: filter row, float4 within the row
#pragma unroll
  for (int i = 0; i < SR_NX; ++i) { const int f = tid + 256 * i; xr[i] = f < nx ? f / w34 : -1; xj[i] = f - (f / w34) * w34; }
  const int64_t G = (int64_t)p.N * p.Ho;
  const int64_t g0 = (int64_t)blockIdx.x * rows_per_wg, g1 = g0 + rows_per_wg < G ? g0 + rows_per_wg : G;
  f32x4 rx[SR_NX], rd[SR_NDY];
  auto load_row = [&](int64_t g) {
    const int n = (int)(g / p.Ho), ho = (int)(g - (int64_t)n * p.Ho);
#pragma unroll
    for (int i = 0; i < SR_NX; ++i) {
      const int hi = ho * p.stride - p.pad + xr[i];
      rx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (xr[i] >= 0 && (unsigned)hi < (unsigned)p.H) rx[i] = *reinterpret_cast<const f32x4*>(x + ((int64_t)n * p.H + hi) * p.W * 3 + 4 * xj[i]);
    }
#pragma unroll
    for (int i = 0; i < SR_NDY; ++i) {
      const int f = tid + 256 * i;
      if (f < ndy) rd[i] = *reinterpret_cast<const f32x4*>(dy + g * p.Wo * SR_K + 4 * f);
    }
  };
  auto store_row = [&]() {
#pragma unroll
    for (int i = 0; i < SR_NX; ++i)
      if (xr[i] >= 0) *reinterpret_cast<f32x4*>(&srows[xr[i] * SR_ROWF + padl + 4 * xj[i]]) = rx[i];
#pragma unroll
    for (int i = 0; i < SR_NDY; ++i) {
      const int f = tid + 256 * i;
      if (f < ndy) *reinterpret_cast<f32x4*>(&sdy[4 * f]) = rd[i];
    }
  };
  f32x16 acc[3];
#pragma unroll
  for (int tn = 0; tn < 3; ++tn)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[tn][j] = 0.f;
  __syncthreads();                                              // the zero fill is complete before the first row is stored over it
  if (g0 < g1) load_row(g0);
  const int nsteps = p.Wo >> 1;
  for (int64_t g = g0; g < g1; ++g) {
    store_row();
    __syncthreads();
    if (g + 1 < g1) load_row(g + 1);                            // the next output row's image rows and dY row fly under this row's MFMAs
#pragma unroll 4
    for (int st = 0; st < nsteps; ++st) {
      const float a = sdy[abase + 2 * st * SR_K];
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, srows[bbase[tn] + 2 * st * s3], acc[tn], 0, 0, 0);
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * SR_K * NCOL;
#pragma unroll
  for (int tn = 0; tn < 3; ++tn) {
    const int col = 96 * wn + 32 * tn + l31;
    if (col >= NCOL) continue;
    const bool real = col - 24 * (col / 24) < 3 * p.S;          // the padding floats of a filter row's 24 stay zero in the row-taps layout
#pragma unroll
    for (int j = 0; j < 16; ++j) out[(size_t)(32 * wm + (j & 3) + 8 * (j >> 2) + 4 * half) * NCOL + col] = real ? acc[tn][j] : 0.f;
  }
}

struct StemRowsPlan { int rows_per_wg, nsplit; };
static inline StemRowsPlan stem_rows_plan(const ssv_conv_desc* d) {      // one resident round of 3 workgroups per CU, whole output rows per workgroup
  const int64_t G = (int64_t)d->N * d->Ho;
  StemRowsPlan r;
  r.rows_per_wg = (int)cdiv64(G, 768);
  r.nsplit = (int)cdiv64(G, r.rows_per_wg);
  return r;
}

struct StemWgradPlan { int tiles, nsplit, chunk; };
StemWgradPlan plan_stem_wgrad(const ssv_conv_desc* d) {
  StemWgradPlan w;
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  w.tiles = cdiv(d->K, 64) * cdiv(d->R * 24, 192);
  int64_t ns = 512 / w.tiles;                                    // this variant holds 207 registers: 2 workgroups per CU, one resident round
  const int64_t max_by_rows = cdiv64(M, 256);
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  const int64_t chunk = cdiv64(cdiv64(M, ns), 32) * 32;
  w.chunk = (int)chunk;
  w.nsplit = (int)cdiv64(M, chunk);
  return w;
}
}  // namespace

extern "C" size_t ssv_stem_conv_wgrad_workspace_bytes(const ssv_conv_desc* d) {
  if (!d || d->K <= 0 || d->R <= 0) return 0;
  const size_t slab = (size_t)d->K * d->R * 24 * sizeof(float);
  const size_t gather = (size_t)plan_stem_wgrad(d).nsplit * slab;
  const size_t rows = stem_rows_ok(d) ? (size_t)stem_rows_plan(d).nsplit * slab : 0;
  return gather > rows ? gather : rows;
}

// which kernel ssv_stem_conv_wgrad takes for this shape: output rows per workgroup of the rows-in-LDS kernel, or 0 = the row-taps gather (a count, not a status)
extern "C" int64_t ssv_stem_conv_wgrad_rows_per_group(const ssv_conv_desc* d) {
  if (!d || d->K <= 0 || d->R <= 0 || d->N <= 0) return 0;
  return stem_rows_ok(d) ? (int64_t)stem_rows_plan(d).rows_per_wg : 0;
}

// dwrows [K][R][24] = weight gradient in the row-taps layout (columns >= 3 S are zero); overwritten, not accumulated
extern "C" int ssv_stem_conv_wgrad(const ssv_conv_desc* d, const float* x, const float* dy, float* dwrows, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_desc(d, "ssv_stem_conv_wgrad")) return rc;
  SSV_REQUIRE(x && dy && dwrows && ws && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dwrows | (uintptr_t)ws) & 15) == 0, "ssv_stem_conv_wgrad: null or unaligned pointer");
  SSV_REQUIRE(d->C == 3 && d->S * 3 <= 24 && d->S <= 8 && d->K % 4 == 0, "ssv_stem_conv_wgrad: a 3-channel input, at most 8 taps per filter row, K %% 4 == 0 (got C=%d S=%d K=%d)", d->C, d->S, d->K);
  const StemWgradPlan wp = plan_stem_wgrad(d);
  const size_t need = ssv_stem_conv_wgrad_workspace_bytes(d);
  if (ws_bytes < need) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_stem_conv_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  const ConvKP p = make_kp(d);
  float* part = (float*)ws;
  if (stem_rows_ok(d)) {                                         // whole output rows from image rows staged in LDS
    const StemRowsPlan rp = stem_rows_plan(d);
    hipLaunchKernelGGL(stem_wgrad_rows_k, dim3((unsigned)rp.nsplit), dim3(256), 0, s, p, x, dy, part, rp.rows_per_wg);
    SSV_CHECK_LAUNCH("ssv_stem_conv_wgrad(rows)");
    const int64_t n = (int64_t)d->K * d->R * 24;
    hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, (const float*)part, rp.nsplit, n, dwrows, 0);
    SSV_CHECK_LAUNCH("ssv_stem_conv_wgrad(reduce)");
    return SSV_OK;
  }
  hipLaunchKernelGGL((conv_wgrad_k<64, 192, 2, 2, 32, true, 3>), dim3((unsigned)(wp.tiles * wp.nsplit)), dim3(256), 0, s, p, x, dy, part, wp.chunk, wp.tiles);
  SSV_CHECK_LAUNCH("ssv_stem_conv_wgrad(partial)");
  const int64_t n = (int64_t)d->K * d->R * 24;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, (const float*)part, wp.nsplit, n, dwrows, 0);
  SSV_CHECK_LAUNCH("ssv_stem_conv_wgrad(reduce)");
  return SSV_OK;
}

// ---- batched GEMMs (the 16 transformed-domain products of a Winograd convolution, csrc/winograd.hip) -------------------------------------
// y[b] [rows][K] = a[b] [rows][C] . w[b]^T [K][C] for b < batch, ONE launch (blockIdx.y = b) of the forward kernel's float4 path
extern "C" int ssv_gemm_batched(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* a, const float* w, float* y, void* stream) {
  SSV_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && rows < (1ll << 31) && C > 0 && K > 0, "ssv_gemm_batched: bad shape");
  SSV_REQUIRE(C % 32 == 0 && K % 4 == 0, "ssv_gemm_batched: needs C %% 32 == 0 and K %% 4 == 0 (got C=%d K=%d)", C, K);
  SSV_REQUIRE(a && w && y && (((uintptr_t)a | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "ssv_gemm_batched: null or unaligned pointer");
  ssv_conv_desc d = {(int32_t)rows, 1, 1, C, K, 1, 1, 1, 0, 1, 1};
  if (int rc = check_desc(&d, "ssv_gemm_batched")) return rc;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(&d);
  p.bs_a = rows * C; p.bs_b = (long long)K * C; p.bs_o = rows * K;
  const bool wide = K >= 128;
  // (Tried: ONE resident round of workgroups, each walking a contiguous run of (tile, batch) units as one continuous k-stream, so that a unit's
  // epilogue runs under the next unit's first loads - the products alone got 10-12 % faster (0.533 -> 0.478 ms at 14x14 x 256), the two-stream
  // training step did not move (231.8 vs 231.7 ms, r03 e2): the other view's kernels already fill those bubbles, and a persistent grid shuts
  // them out.  Not kept.)
  const dim3 grid((unsigned)(wide ? cdiv(p.M, 128) * cdiv(K, 128) : cdiv(p.M, 256) * cdiv(K, 64)), (unsigned)batch);
  if (wide) hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, 32, true, false, false, false, false>), grid, dim3(256), 0, s, p, a, w, (const float*)nullptr, (const float*)nullptr, y);
  else      hipLaunchKernelGGL((conv_fwd_k<256, 64, 4, 1, 32, true, false, false, false, false>), grid, dim3(256), 0, s, p, a, w, (const float*)nullptr, (const float*)nullptr, y);
  SSV_CHECK_LAUNCH("ssv_gemm_batched");
  return SSV_OK;
}

// The same products in SSV_ARITH_BF16X3 (csrc/split_bf16.h): the forward kernel's bf16-piece variant on pre-split weights [batch][3][K][C]; bias / addend as in
// ssv_conv2d_fwd's epilogue (the plain 1x1 / Linear products of the ViT, the heads and the stage-exit gradients use batch 1 of it).
extern "C" int ssv_gemm_batched_split(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* a, const void* w_planes, float* y,
                                      const float* bias, const float* addend, void* stream) {
  SSV_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && rows < (1ll << 31) && C > 0 && K > 0, "ssv_gemm_batched_split: bad shape");
  SSV_REQUIRE(C % 32 == 0 && K % 4 == 0, "ssv_gemm_batched_split: needs C %% 32 == 0 and K %% 4 == 0 (got C=%d K=%d)", C, K);
  SSV_REQUIRE(a && w_planes && y && (((uintptr_t)a | (uintptr_t)w_planes | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)addend) & 15) == 0, "ssv_gemm_batched_split: null or unaligned pointer");
  SSV_REQUIRE(!bias || batch == 1, "ssv_gemm_batched_split: a bias goes with one product (batch 1)");
  ssv_conv_desc d = {(int32_t)rows, 1, 1, C, K, 1, 1, 1, 0, 1, 1};
  if (int rc = check_desc(&d, "ssv_gemm_batched_split")) return rc;
  d.arithmetic = SSV_ARITH_BF16X3; d.w_planes = w_planes;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  ConvKP p = make_kp(&d);
  p.bs_a = rows * C; p.bs_b = (long long)K * C; p.bs_o = rows * K;
  p.wp_stride = (long long)batch * K * C;
  SSV_REQUIRE(p.wp_stride * 6 < (1ll << 31), "ssv_gemm_batched_split: weight planes of 2 GiB or more");
  const bool wide = K >= 128;
  const dim3 grid((unsigned)(cdiv(p.M, 128) * cdiv(K, wide ? 128 : 64)), (unsigned)batch);
  const float* w = nullptr;
#define BSP(BN_, SP_) hipLaunchKernelGGL((conv_fwd_k<128, BN_, 2, 2, 32, true, false, false, false, false, 0, 0, false, false, SP_>), grid, dim3(256), 0, s, p, a, w, bias, addend, y)
  if (C > SP_DUAL_FROM) { if (wide) BSP(128, 2); else BSP(64, 2); }
  else                  { if (wide) BSP(128, 1); else BSP(64, 1); }
#undef BSP
  SSV_CHECK_LAUNCH("ssv_gemm_batched_split");
  return SSV_OK;
}

namespace {
// the same fixed-order fold with the slabs accumulated in fp64: the cross-chunk half of the BLOCKED weight gradient (ssv_gemm_batched_wgrad_blocked)
__global__ void __launch_bounds__(256)
wgrad_reduce64_k(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ dw, int accumulate) {
  __shared__ double sm[4][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  partial += (size_t)blockIdx.y * nsplit * n; dw += (size_t)blockIdx.y * n;
  const int per = (nsplit + 3) / 4;
  const int k0 = grp * per, k1 = min(k0 + per, nsplit);
  double s = 0.0;
  if (i < n) for (int k = k0; k < k1; ++k) s += (double)partial[(size_t)k * n + i];
  sm[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < n) dw[i] = (float)((accumulate ? (double)dw[i] : 0.0) + ((sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e])));
}

struct BatchedWgradPlan { int bm, bn, tiles, nsplit, chunk; };
BatchedWgradPlan plan_batched_wgrad(int batch, int64_t rows, int C, int K, int max_chunk = 0) {
  BatchedWgradPlan w;
  w.bm = K >= 128 ? 128 : 64;
  w.bn = C <= 64 ? 64 : 128;
  w.tiles = cdiv(K, w.bm) * cdiv(C, w.bn);
  const int slots = w.bm == 128 ? 768 : 1024;                 // one resident round over ALL batches (round down: see plan_wgrad)
  int64_t ns = slots / ((int64_t)w.tiles * batch);
  const int64_t max_by_rows = cdiv64(rows, 256);
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  int64_t chunk = cdiv64(cdiv64(rows, ns), 32) * 32;
  if (max_chunk > 0 && chunk > max_chunk) {                   // blocked accumulation: no fp32 chain longer than max_chunk rows; equal chunks
    const int64_t nb = cdiv64(rows, (int64_t)max_chunk);
    chunk = cdiv64(cdiv64(rows, nb), 32) * 32;
  }
  w.chunk = (int)chunk;
  w.nsplit = (int)cdiv64(rows, chunk);
  return w;
}
}  // namespace

extern "C" size_t ssv_gemm_batched_wgrad_workspace_bytes(int32_t batch, int64_t rows, int32_t C, int32_t K) {
  if (batch <= 0 || rows <= 0 || C <= 0 || K <= 0) return 0;
  const BatchedWgradPlan w = plan_batched_wgrad(batch, rows, C, K);
  return (size_t)batch * w.nsplit * K * C * sizeof(float);
}
extern "C" size_t ssv_gemm_batched_wgrad_blocked_workspace_bytes(int32_t batch, int64_t rows, int32_t C, int32_t K, int32_t max_chunk_rows) {
  if (batch <= 0 || rows <= 0 || C <= 0 || K <= 0 || (max_chunk_rows != 0 && max_chunk_rows < 32)) return 0;
  const BatchedWgradPlan w = plan_batched_wgrad(batch, rows, C, K, max_chunk_rows);
  return (size_t)batch * w.nsplit * K * C * sizeof(float);
}

// dw[b] [K][C] = dy[b]^T [K][rows] . x[b] [rows][C] for b < batch: one launch of the weight-gradient kernel (split over the rows, fixed-order reduce).
// max_chunk > 0 (ssv_gemm_batched_wgrad_blocked): BLOCKED accumulation - no fp32 accumulator chain runs over more than max_chunk rows, and the slabs are
// folded in fp64 (still fixed order): the error of a long transformed-domain sum (Winograd F(4x4): 25,088 tiles at 28x28 / batch 512) stops growing with its length.
static int gemm_batched_wgrad(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw, int max_chunk, int flush_rows,
                              void* ws, size_t ws_bytes, void* stream, const char* who, bool sp = false) {
  SSV_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && rows < (1ll << 31) && C > 0 && K > 0, "%s: bad shape", who);
  SSV_REQUIRE(C % 4 == 0 && K % 4 == 0, "%s: needs C %% 4 == 0 and K %% 4 == 0 (got C=%d K=%d)", who, C, K);
  SSV_REQUIRE(x && dy && dw && ws && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)ws) & 15) == 0, "%s: null or unaligned pointer", who);
  ssv_conv_desc d = {(int32_t)rows, 1, 1, C, K, 1, 1, 1, 0, 1, 1};
  if (int rc = check_desc(&d, who)) return rc;
  const BatchedWgradPlan wp = plan_batched_wgrad(batch, rows, C, K, max_chunk);
  const size_t need = (size_t)batch * wp.nsplit * K * C * sizeof(float);
  if (ws_bytes < need) SSV_FAIL(SSV_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  ConvKP p = make_kp(&d);
  p.bs_a = rows * C; p.bs_b = rows * K; p.bs_o = (long long)wp.nsplit * K * C;
  float* part = (float*)ws;
  SSV_REQUIRE((int64_t)wp.tiles * wp.nsplit < (1ll << 31), "%s: too many workgroups", who);
  const dim3 grid((unsigned)(wp.tiles * wp.nsplit), (unsigned)batch);
#define BWG_(BM_, BN_, WM_, WN_, FL_, SP_) hipLaunchKernelGGL((conv_wgrad_k<BM_, BN_, WM_, WN_, 32, true, 1, false, false, false, FL_, SP_>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk, wp.tiles)
#define BWG(BM_, BN_, WM_, WN_, FL_) do { if (sp) BWG_(BM_, BN_, WM_, WN_, FL_, true); else BWG_(BM_, BN_, WM_, WN_, FL_, false); } while (0)
  if (flush_rows > 0) {                       // two-level accumulation: blocks of 4 k-tiles = 128 rows
    if (wp.bm == 128) { if (wp.bn == 64) BWG(128, 64, 2, 2, 4); else BWG(128, 128, 2, 2, 4); }
    else              { if (wp.bn == 64) BWG(64, 64, 2, 2, 4); else BWG(64, 128, 1, 4, 4); }
  } else {
    if (wp.bm == 128) { if (wp.bn == 64) BWG(128, 64, 2, 2, 0); else BWG(128, 128, 2, 2, 0); }
    else              { if (wp.bn == 64) BWG(64, 64, 2, 2, 0); else BWG(64, 128, 1, 4, 0); }
  }
#undef BWG
#undef BWG_
  SSV_CHECK_LAUNCH("ssv_gemm_batched_wgrad(partial)");
  const int64_t n = (int64_t)K * C;
  if (max_chunk > 0 || flush_rows > 0) hipLaunchKernelGGL(wgrad_reduce64_k, dim3((unsigned)cdiv64(n, 64), (unsigned)batch), dim3(256), 0, s, (const float*)part, wp.nsplit, n, dw, 0);
  else hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(n, 64), (unsigned)batch), dim3(256), 0, s, (const float*)part, wp.nsplit, n, dw, 0);
  SSV_CHECK_LAUNCH("ssv_gemm_batched_wgrad(reduce)");
  return SSV_OK;
}

extern "C" int ssv_gemm_batched_wgrad(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw,
                                      void* ws, size_t ws_bytes, void* stream) {
  return gemm_batched_wgrad(batch, rows, C, K, x, dy, dw, 0, 0, ws, ws_bytes, stream, "ssv_gemm_batched_wgrad");
}

extern "C" int ssv_gemm_batched_wgrad_blocked(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw,
                                              int32_t max_chunk_rows, int32_t flush_rows, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(max_chunk_rows == 0 || max_chunk_rows >= 32, "ssv_gemm_batched_wgrad_blocked: max_chunk_rows must be 0 (the plain split) or >= 32 (got %d)", max_chunk_rows);
  SSV_REQUIRE(flush_rows == 0 || flush_rows == 128, "ssv_gemm_batched_wgrad_blocked: flush_rows must be 0 or 128 (got %d)", flush_rows);
  SSV_REQUIRE(max_chunk_rows > 0 || flush_rows > 0, "ssv_gemm_batched_wgrad_blocked: nothing blocked - use ssv_gemm_batched_wgrad");
  return gemm_batched_wgrad(batch, rows, C, K, x, dy, dw, max_chunk_rows, flush_rows, ws, ws_bytes, stream, "ssv_gemm_batched_wgrad_blocked");
}

extern "C" int ssv_gemm_batched_wgrad_split(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw,
                                            int32_t max_chunk_rows, int32_t flush_rows, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(max_chunk_rows == 0 || max_chunk_rows >= 32, "ssv_gemm_batched_wgrad_split: max_chunk_rows must be 0 (the plain split) or >= 32 (got %d)", max_chunk_rows);
  SSV_REQUIRE(flush_rows == 0 || flush_rows == 128, "ssv_gemm_batched_wgrad_split: flush_rows must be 0 or 128 (got %d)", flush_rows);
  return gemm_batched_wgrad(batch, rows, C, K, x, dy, dw, max_chunk_rows, flush_rows, ws, ws_bytes, stream, "ssv_gemm_batched_wgrad_split", true);
}

// which arithmetic a launch described by d takes (include/ssv_hip.h)
extern "C" int ssv_conv_arithmetic(const ssv_conv_desc* d, int32_t product) {
  SSV_REQUIRE(d != nullptr && product >= 0 && product <= 2, "ssv_conv_arithmetic: bad arguments");
  if (int rc = check_desc(d, "ssv_conv_arithmetic")) return rc;
  if (product == 0) return sp_fwd_ok(d) ? SSV_ARITH_BF16X3 : SSV_ARITH_F32_MFMA;
  if (product == 2) return (sp_wgrad_ok(d) && !(d->C == 3)) ? SSV_ARITH_BF16X3 : SSV_ARITH_F32_MFMA;
  return sp_dgrad_ok(d) ? SSV_ARITH_BF16X3 : SSV_ARITH_F32_MFMA;      // the strided data-gradient kernel (stride-1 data gradients are forward-kernel launches on the transposed filter)
}

#ifdef SSV_STAMP
extern "C" int ssv_debug_stamps(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
