// fp32 products on the BF16 matrix pipe of gfx950 by operand splitting (arithmetic SSV_ARITH_BF16X3, round 6).
//
//   a = a0 + a1 + a2,  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)     round to nearest even (v_cvt_pk_bf16_f32); the residuals are exact fp32
//   subtractions and 3 x 8 significant bits (+ the pieces' signs) cover the 24 of an fp32: a == a0 + a1 + a2 exactly unless a piece underflows (below).
//   a * b = sum of ai * bj.  Every ai * bj is exact (8 x 8 bits) and summed in fp32 by v_mfma_f32_16x16x32_bf16, which folds 32 products per accumulator
//   rounding (v_mfma_f32_32x32x2_f32 folds 2).  SIX of the nine terms are kept - a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 - what is dropped (a1b2, a2b1, a2b2) is
//   <= 2^-23 |a b| in the worst case and ~2^-27 rms - below ONE fp32 rounding of the product.  Measured against fp64 the 6-term and the 9-term product have the
//   same error: at or below the fp32-MFMA kernel's on every product shape of the networks (mma6 / mma6x2 below; tests/test_gpu_split.py).
//
// Edge magnitudes (decided, tested in tests/test_gpu_split.py::test_edge_magnitudes):
//   * +-0 split to (+-0, 0, 0): products with zero are exact zeros.
//   * |a| < 2^-109 (~1.5e-33): the second / third piece fall below bf16's (= fp32's) exponent range and are rounded into its denormals or to zero, so the
//     operand is carried with fewer than 24 bits (8 at 1e-38).  Nothing in these networks lives there (the smallest gradients of a step are ~1e-12).
//   * |a| >= 3.3895e38 (bf16's largest finite value is below fp32's): a0 rounds to infinity.  Inf and NaN operands: a0 = Inf / NaN and the residual Inf - Inf
//     is NaN, so every output that touches a non-finite operand is NaN (the fp32-MFMA path gives Inf or NaN there).  A step that produced such a value has
//     diverged on either path; the parity suites never see one.
//
// LDS images (per operand THREE planes, one per piece):
//   ROWK  [row][32 k] bf16, 64-byte rows, UNPADDED: the four 16-byte slots of a row are XOR-swizzled by the row so that the ds_read_b128 of the 16x16x32
//         operand (lane l: row l & 15, slot l >> 4) is conflict-free over the instruction's four 16-lane groups.  (128 + 128) rows x 3 planes = 48 KB:
//         three workgroups per CU, as many as the fp32 kernels keep.
//   KROW  [32 k][rows] bf16 (the weight gradient: contraction over pixels, channels contiguous in HBM): 256- or 128-byte rows with the 16-byte chunks
//         XOR-swizzled, read by ds_read_b64_tr_b16 (a 16-lane group fetches 4 k x 16 channels and each lane receives ITS channel's 4 k values).
// Staging: a thread converts the float4 it loaded (after any formed-on-load transform) into the three planes' 8-byte pieces: 22 VALU per float4.
// Pre-split operands (the weights: ssv_split_planes, once per weight and step) arrive as [3][rows][k] bf16 and go to LDS in 16-byte pieces untouched.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace splitbf {

constexpr int BK = 32;                 // contraction per k-tile = one v_mfma_f32_16x16x32_bf16
constexpr int ROWB = 2 * BK;           // bytes of one ROWK row in one plane

// four floats -> the three planes' packed pairs
__device__ __forceinline__ void split4(const f32x4& v, u32x2 (&pl)[3]) {
  f32x2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf16x2)), p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf16x2));
    if (q < 2) {
      x0 -= f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
      x1 -= f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
    }
    pl[q] = u32x2{p0, p1};
  }
}

// ---- ROWK images ---------------------------------------------------------------------------------------------------------------------------------------
// byte offset of 16-byte slot `slot` (0..3) of row `row` inside one plane.  Swizzle f(row) = 3 * bit 3 of the row: over the 16 lanes of a ds_read_b128 group
// - e.g. rows 0-3 and 12-15 at slot 0 with rows 4-11 at slot 1 - the pairs (row & 3, slot ^ f(row)) are all distinct, i.e. 16 distinct 16-byte pieces of the
// 256-byte bank row; the other three groups likewise (all four enumerated in the round-6 notes of DESIGN.md)
__device__ __forceinline__ int rowk_off(int row, int slot) { return row * ROWB + ((slot ^ (((row >> 3) & 1) * 3)) << 4); }

// stage float4 number k4 (0..7: k = 4 k4 .. 4 k4 + 3) of `row` into the three planes (ds_write_b64 each)
__device__ __forceinline__ void rowk_store(unsigned char* img, int plane_bytes, int row, int k4, const f32x4& v) {
  u32x2 pl[3];
  split4(v, pl);
  const int off = rowk_off(row, k4 >> 1) + (k4 & 1) * 8;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(img + q * plane_bytes + off) = pl[q];
}

// the 16x16x32 operand fragment of 16 consecutive rows from `row0`: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7
__device__ __forceinline__ bf16x8 rowk_frag(const unsigned char* plane, int row0, int lane) {
  return *reinterpret_cast<const bf16x8*>(plane + rowk_off(row0 + (lane & 15), lane >> 4));
}

// ---- KROW images ---------------------------------------------------------------------------------------------------------------------------------------
// [k = 0..31][RB bytes] per plane, RB = 2 * channels of the tile (256 or 128); 16-byte chunk c of k-row m lives at chunk c ^ f(m):
//   256-byte rows: f = ((m & 3) << 2) | ((m >> 2) & 3)   (the dual-use image (b) of the programming guide, T10)
//   128-byte rows: f = 2 * (((m >> 1) & 1) | (((m >> 3) & 1) << 1))
// With either, the 8 (k-row, 32-byte) pieces one 32-lane half of a ds_read_b64_tr_b16 touches - rows 4 rd + q and 8 + 4 rd + q, q = 0..3, same 16 channels -
// fall on 8 distinct 32-byte positions of the 256-byte bank row.
template <int RB>
__device__ __forceinline__ int krow_off(int m, int chunk) {
  static_assert(RB == 256 || RB == 128, "KROW rows are 128 or 64 channels");
  if constexpr (RB == 256) return m * 256 + ((chunk ^ (((m & 3) << 2) | ((m >> 2) & 3))) << 4);
  else return m * 128 + ((chunk ^ ((((m >> 1) & 1) | (((m >> 3) & 1) << 1)) << 1)) << 4);
}

// stage a float4 (channels ch .. ch + 3, ch % 4 == 0, of k-row m) into the three planes
template <int RB>
__device__ __forceinline__ void krow_store(unsigned char* img, int plane_bytes, int m, int ch, const f32x4& v) {
  u32x2 pl[3];
  split4(v, pl);
  const int off = krow_off<RB>(m, ch >> 3) + ((ch >> 2) & 1) * 8;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(img + q * plane_bytes + off) = pl[q];
}

// the 16x16x32 operand fragment of channels ch0 .. ch0 + 15 (ch0 % 16 == 0): lane l holds channel ch0 + (l & 15), k = 8 (l >> 4) .. + 7, from two transposed
// reads of 4 k x 16 channels each (lane 4 q + p of a 16-lane group supplies the address of k-row q, channels 4 p .. 4 p + 3).  EXEC must be all ones.
template <int RB>
__device__ __forceinline__ bf16x8 krow_frag(const unsigned char* plane, int ch0, int lane) {
  const int jl = lane & 15, q = jl >> 2, pp = jl & 3, kg = lane >> 4;
  const int chunk = (ch0 >> 3) + (pp >> 1);
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + krow_off<RB>(8 * kg + q, chunk) + 8 * (pp & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + krow_off<RB>(8 * kg + 4 + q, chunk) + 8 * (pp & 1)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- the six piece products of one (output-row fragment, output-column fragment) pair and k-tile ---------------------------------------------------------
// Two forms, chosen per product KIND (never per fused variant: the fused and unfused forms of one layer stay bit-identical):
//   mma6   - all six terms into the running 16 x 16 accumulator, smallest first.  The accumulator takes six roundings per 32 products and those set the error: it sits
//            10-13 % UNDER the fp32-MFMA kernel's on forward-shaped products (9.9e-7 vs 1.15e-6 at a contraction of 4,096), but 12 % OVER it on a weight-gradient
//            shape.  64 accumulator registers per lane on the 128 x 128 tile: three workgroups per CU.  Forward and data-gradient launches.
//   mma6x2 - a0b0 into the running accumulator (ONE rounding per 32 products at the running sum's magnitude) and the five small terms (<= 2^-8 of a0b0 each) into a
//            second one, added in the epilogue: 2.5-3x lower error (3.9e-7 at 4,096), 64 more registers - two workgroups per CU, 12-19 % slower per launch.
//            Weight gradients (their contraction runs over ~10^5 pixels in chunks).
// Measured: tools/probe/gemm_split_tuned_probe.hip, profiles/r06_probe_split_accmode.txt, r06_probe_split_prefetch.txt.  A third form - the six products summed in a
// fresh tile that the vector unit adds to the accumulator - has mma6x2's error but the compiler keeps all 16 fresh tiles live (117-161 spilled dwords): not shipped.
// Every tile has the COLUMN operand as the instruction's A: lane l then holds output row (l & 15) and the four consecutive output columns 4 (l >> 4) .. + 3 -
// 16-byte pieces of whole output rows for the epilogue.
__device__ __forceinline__ void mma6(f32x4& acc, const bf16x8 (&col)[3], const bf16x8 (&row)[3]) {
#define SSV_MM(P, Q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col[P], row[Q], acc, 0, 0, 0)
  SSV_MM(2, 0); SSV_MM(0, 2); SSV_MM(1, 1); SSV_MM(1, 0); SSV_MM(0, 1); SSV_MM(0, 0);
#undef SSV_MM
}
__device__ __forceinline__ void mma6x2(f32x4& acc, f32x4& lo, const bf16x8 (&col)[3], const bf16x8 (&row)[3]) {
#define SSV_ML(P, Q) lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col[P], row[Q], lo, 0, 0, 0)
  SSV_ML(2, 0); SSV_ML(0, 2); SSV_ML(1, 1); SSV_ML(1, 0); SSV_ML(0, 1);
#undef SSV_ML
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col[0], row[0], acc, 0, 0, 0);
}

}  // namespace splitbf
