// ViT encoder pieces of the reference (networks/vit.py) that are not plain GEMMs, for gfx950:
//   * token assembly  [cls | patches] ++ positional embedding (concatenated along the feature axis, :71-82, :101-102)
//   * LayerNorm over the feature axis with a fused addend  (out = f(x) + LayerNorm(x), :22-31 and :43-46)
//   * exact (erf) GELU
//   * multi-head softmax(QK^T/sqrt(d))V on fp32 MFMA, forward + backward, without materialising the T x T matrix.
// Linear layers run on the implicit-GEMM kernels of conv_mfma.hip (a Linear is a 1x1 convolution over M = B*T "pixels").
//
// Attention tiling (wave64, v_mfma_f32_32x32x2_f32, head size 64).  A 32x32 accumulator tile keeps its COLUMN on the lane
// (col = lane & 31) and 16 rows in registers (row = (j&3) + 8*(j>>2) + 4*(lane>>5)).  Every product is therefore arranged
// so that the index a softmax statistic belongs to is the column:
//   forward / dQ pass : S^T = K Q^T          -> query on the lane: running max / sum / log-sum-exp / delta are lane-local;
//                       O^T += V^T P^T       -> the P^T accumulator registers are fed back AS the B operand, step j uses the
//                                               key rows {(j&3)+8(j>>2)+4*half}: no shuffles, no LDS round trip;
//   dK/dV pass        : S  = Q K^T, dP = dO V^T (key on the lane), dV^T += dO^T P, dK^T += Q^T dS the same way.
// The contraction over the head dimension is taken in the order d = 32*half + s, so each lane's operand fragment is 32
// CONSECUTIVE floats of one row (8 x 16-byte loads).
#include "common.h"
#include "split_bf16.h"

namespace {

constexpr int DH = 64;

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ int rowof(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }

// 32 consecutive floats of one row -> registers (optionally scaled)
__device__ __forceinline__ void load_row32(const float* __restrict__ p, float* __restrict__ r, float scale) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = *(const f32x4*)(p + 4 * i);
    r[4 * i + 0] = v[0] * scale; r[4 * i + 1] = v[1] * scale; r[4 * i + 2] = v[2] * scale; r[4 * i + 3] = v[3] * scale;
  }
}

// acc^T[d][col] tile stored as rows of a [T][ld] matrix: lane (col, half) owns 4 consecutive d per register quad
__device__ __forceinline__ void store_tile_T(float* __restrict__ base, int64_t row_stride, int row, bool valid, int half,
                                             const f32x16& lo, const f32x16& hi, float mul) {
  if (!valid) return;
  float* p = base + (int64_t)row * row_stride;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = lo[4 * jj + e] * mul; b[e] = hi[4 * jj + e] * mul; }
    *(f32x4*)(p + 8 * jj + 4 * half) = a;
    *(f32x4*)(p + 32 + 8 * jj + 4 * half) = b;
  }
}

// The same tile through a wave-private 32 x 64 float LDS scratch, so that the global stores are whole 256-byte rows (16 lanes x 16 bytes, four rows per
// instruction) instead of 64 scattered 16-byte pieces: the scattered form is store-ISSUE bound (~7 B / cycle / CU: tools/stamp_attn.py found the 37-token
// backward spending a quarter of its time issuing dQ stores).  Row c keeps its sixteen 16-byte slots XOR-swizzled with c & 15: the transposing writes and the
// row-major reads are both free of bank conflicts without padding (the scratch is exactly one tile).  Rows [0, nrows) of the tile are stored.
__device__ __forceinline__ void store_tile_T_rows(float* __restrict__ scratch, float* __restrict__ base, int64_t row_stride, int row0, int nrows, int lane,
                                                  const f32x16& lo, const f32x16& hi, float mul) {
  const int c = lane & 31, half = lane >> 5;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = lo[4 * jj + e] * mul; b[e] = hi[4 * jj + e] * mul; }
    *(f32x4*)(scratch + c * 64 + (((2 * jj + half) ^ (c & 15)) << 2)) = a;
    *(f32x4*)(scratch + c * 64 + (((8 + 2 * jj + half) ^ (c & 15)) << 2)) = b;
  }
  const int s = lane & 15;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = 4 * k + (lane >> 4);
    const f32x4 v = *(const f32x4*)(scratch + r * 64 + ((s ^ (r & 15)) << 2));
    if (r < nrows) *(f32x4*)(base + (int64_t)(row0 + r) * row_stride + 4 * s) = v;
  }
}

// ------------------------------------------------------------------------------------------------ attention, shared pieces
// A workgroup = NW wavefronts = NW consecutive 32-row tiles of ONE (image, head).  The tiles it streams over (keys/values in the
// forward and dQ passes, queries/dO in the dK/dV pass) are staged once per workgroup into LDS with coalesced 16-byte loads
// (row stride 68 floats: conflict-free ds_read_b128 of a 32-float row fragment) and double-buffered: the global loads of tile
// t+1 are in flight while the MFMAs of tile t run.
// Scores are kept in log2 units (Q is pre-multiplied by scale * log2 e), so every softmax exponential is one v_exp_f32;
// the saved log-sum-exp stays in natural units (converted on store / load).
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int TS = 68;                                        // LDS row stride (floats) of a staged 32 x 64 tile

template <int NW>
struct Stage {                                                // registers holding one thread's share of two 32x64 tiles
  static constexpr int N4 = 512 / (NW * 64);                  // float4 per thread per tile
  f32x4 a[N4], b[N4];
  __device__ __forceinline__ void load(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int row0, int T) {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
      const int64_t row = min(row0 + r, T - 1);
      a[i] = *(const f32x4*)(A + row * lda + c4);
      b[i] = *(const f32x4*)(B + row * ldb + c4);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ sa, float* __restrict__ sb) const {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
      *(f32x4*)(sa + r * TS + c4) = a[i];
      *(f32x4*)(sb + r * TS + c4) = b[i];
    }
  }
};

__device__ __forceinline__ void lds_row32(const float* __restrict__ tile, int row, int half, float* __restrict__ r) {
  const float* p = tile + row * TS + half * 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = *(const f32x4*)(p + 4 * i);
    r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
  }
}

// ------------------------------------------------------------------------------------------------ attention forward
template <int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 3 : 1)      // 4-wave blocks: 3 per CU (139 registers, no spill) - measured -10 % against 2 per CU
attn_fwd_k(int T, int heads, const float* __restrict__ Q, const float* __restrict__ K,
                                                      const float* __restrict__ V, int ld, float scale,
                                                      float* __restrict__ O, int ldo, float* __restrict__ LSE) {
  __shared__ __attribute__((aligned(16))) float skv[4][32 * TS];           // K stages 0, 1 | V stages 0, 1; at the end: one output tile per wave
  float (*sk)[32 * TS] = skv, (*sv)[32 * TS] = skv + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < T;                                 // a trailing wave only helps staging
  const int64_t tok0 = (int64_t)b * T;
  const int qrow = min(q0 + c, T - 1);
  float qreg[32];
  load_row32(Q + (tok0 + qrow) * ld + h * DH + half * 32, qreg, scale * LOG2E);
  f32x16 o_lo = zero16(), o_hi = zero16();
  float m = -INFINITY, l = 0.f;
  const float* kbase = K + tok0 * ld + h * DH;
  const float* vbase = V + tok0 * ld + h * DH;
  Stage<NW> st;
  st.load(kbase, ld, vbase, ld, 0, T);
  st.store(sk[0], sv[0]);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < T;
    if (more) st.load(kbase, ld, vbase, ld, k0 + 32, T);
    if (active) {
      float kreg[32];
      lds_row32(sk[buf], c, half, kreg);
      f32x16 s = zero16();
#pragma unroll
      for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[i], qreg[i], s, 0, 0, 0);
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k0 + rowof(j, half) >= T) s[j] = -INFINITY;
        mt = fmaxf(mt, s[j]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m, mt);
      const float alpha = ex2(m - mn);                        // m = -inf on the first tile -> 0
      float ls = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { s[j] = ex2(s[j] - mn); ls += s[j]; }
      ls += __shfl_xor(ls, 32, 64);
      l = l * alpha + ls;
      m = mn;
#pragma unroll
      for (int j = 0; j < 16; ++j) { o_lo[j] *= alpha; o_hi[j] *= alpha; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k0 + rowof(j, 0) >= T) continue;                   // both keys of this step are padding (wave-uniform): P is 0 there
        const float* vr = sv[buf] + rowof(j, half) * TS;
        o_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[c], s[j], o_lo, 0, 0, 0);
        o_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 + c], s[j], o_hi, 0, 0, 0);
      }
    }
    if (more) st.store(sk[buf ^ 1], sv[buf ^ 1]);
    __syncthreads();
  }
  const bool valid = q0 + c < T;
  // (the loop's last barrier has passed: the four staging buffers are free - 2 x 2 x 32 x 68 floats, of which each wave takes a private 32 x 64 tile)
  static_assert(NW * 2048 <= 4 * 32 * TS, "one output tile per wave in the staging buffers");
  if (active) store_tile_T_rows(&skv[0][0] + wave * 2048, O + tok0 * ldo + h * DH, ldo, q0, min(32, T - q0), lane, o_lo, o_hi, 1.f / l);
  if (valid && half == 0) LSE[((int64_t)b * heads + h) * T + q0 + c] = (m + log2f(l)) * LN2;
}

// ------------------------------------------------------------------------------------------------ attention forward, SSV_ARITH_BF16X3 (csrc/split_bf16.h)
// The same tiling and softmax (S^T = K Q^T with the query on the lane, P^T fed back as the B operand of O^T += V^T P^T), every product as six bf16 piece products on
// v_mfma_f32_32x32x16_bf16: a 32-key tile costs 2 x 24 MFMAs of 32 cycles where the fp32 form issues 2 x 32 of 64.
//   * Q (pre-scaled) is split once per wave into registers: the instruction's B operand wants 8 consecutive d per lane and 16-wide slab, d = 16 s + 8 (lane >> 5) + e.
//   * K and V tiles are split while they are staged: three planes of [32 keys][64 d] bf16 each (128-byte rows, 16-byte chunks XOR-swizzled).  K fragments are row reads
//     (ds_read_b128); V^T fragments are transposed reads (ds_read_b64_tr_b16: a 16-lane group fetches 4 keys x 16 d and every lane receives ITS d's 4 keys).
//   * P stays in the S accumulator's registers: lane (query, half) holds keys (j & 3) + 8 (j >> 2) + 4 half, so registers 8 s' .. 8 s' + 7 are the keys
//     16 s' + 4 half + {0..3} and 16 s' + 8 + 4 half + {0..3} - the contraction index of the 16-key slab s' is taken in THAT order on both operands (the V^T
//     fragment's two transposed reads start at exactly those two key quads), so P is split in place: no shuffle, no LDS round trip, as in the fp32 form.
constexpr int KVP = 32 * 128;                                // bytes of one [32 keys][64 d] bf16 plane
__device__ __forceinline__ int kplane_off(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }      // ds_read_b128 of lanes = keys: 8 distinct chunks per key parity
__device__ __forceinline__ int vplane_off(int key, int chunk) { return key * 128 + ((chunk ^ (((key >> 1) & 1) << 2)) << 4); } // transposed reads: keys q and q + 2 on opposite 64-byte halves

// eight floats -> the three planes' bf16x8
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8 (&pl)[3]) {
  u32x2 pa[3], pb[3];
  splitbf::split4(a, pa);
  splitbf::split4(b, pb);
#pragma unroll
  for (int q = 0; q < 3; ++q) pl[q] = __builtin_bit_cast(bf16x8, u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]});
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2)
attn_fwd_sp_k(int T, int heads, const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, int ld, float scale,
              float* __restrict__ O, int ldo, float* __restrict__ LSE) {
  // [stage][K | V][plane][32 keys x 128 B]; at the end: one fp32 output tile per wave (NW x 8 KB <= 48 KB)
  __shared__ __attribute__((aligned(16))) unsigned char skv[2 * 2 * 3 * KVP];
  static_assert(NW * 8192 <= 2 * 2 * 3 * KVP, "one output tile per wave in the staging buffers");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < T;
  const int64_t tok0 = (int64_t)b * T;
  const int qrow = min(q0 + c, T - 1);
  bf16x8 qf[4][3];                                             // slab s: d = 16 s + 8 half + {0..7}
  {
    const float* qp = Q + (tok0 + qrow) * ld + h * DH + 8 * half;
    const float sc = scale * LOG2E;
#pragma unroll
    for (int s = 0; s < 4; ++s) split8(*(const f32x4*)(qp + 16 * s) * sc, *(const f32x4*)(qp + 16 * s + 4) * sc, qf[s]);
  }
  f32x16 o_lo = zero16(), o_hi = zero16();
  float m = -INFINITY, l = 0.f;
  const float* kbase = K + tok0 * ld + h * DH;
  const float* vbase = V + tok0 * ld + h * DH;
  constexpr int N4 = 512 / (NW * 64);
  f32x4 rk[N4], rv[N4];
  auto load_kv = [&](int row0) {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
      const int64_t row = min(row0 + r, T - 1);
      rk[i] = *(const f32x4*)(kbase + row * ld + c4);
      rv[i] = *(const f32x4*)(vbase + row * ld + c4);
    }
  };
  auto store_kv = [&](int stage) {
    unsigned char* kp = skv + stage * 6 * KVP;
    unsigned char* vp = kp + 3 * KVP;
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = e & 15;          // c4: float4 index within the 64-float row
      u32x2 pk[3], pv[3];
      splitbf::split4(rk[i], pk);
      splitbf::split4(rv[i], pv);
      const int ko = kplane_off(r, c4 >> 1) + (c4 & 1) * 8, vo = vplane_off(r, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) { *(u32x2*)(kp + q * KVP + ko) = pk[q]; *(u32x2*)(vp + q * KVP + vo) = pv[q]; }
    }
  };
  load_kv(0);
  store_kv(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < T;
    if (more) load_kv(k0 + 32);
    if (active) {
      const unsigned char* kp = skv + buf * 6 * KVP;
      const unsigned char* vp = kp + 3 * KVP;
      f32x16 s = zero16();
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        bf16x8 kf[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) kf[q] = *(const bf16x8*)(kp + q * KVP + kplane_off(c, 2 * sl + half));
#define SSV_MM(P, Q_) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[P], qf[sl][Q_], s, 0, 0, 0)
        SSV_MM(2, 0); SSV_MM(0, 2); SSV_MM(1, 1); SSV_MM(1, 0); SSV_MM(0, 1); SSV_MM(0, 0);
#undef SSV_MM
      }
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k0 + rowof(j, half) >= T) s[j] = -INFINITY;
        mt = fmaxf(mt, s[j]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m, mt);
      const float alpha = ex2(m - mn);
      float ls = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { s[j] = ex2(s[j] - mn); ls += s[j]; }
      ls += __shfl_xor(ls, 32, 64);
      l = l * alpha + ls;
      m = mn;
#pragma unroll
      for (int j = 0; j < 16; ++j) { o_lo[j] *= alpha; o_hi[j] *= alpha; }
      // lane -> its V^T rows: d = 32 hh + 16 ((lane >> 4) & 1) + (lane & 15); the lane's address names key quad row (lane & 15) >> 2, d piece (lane & 3)
      const int jl = lane & 15, qq = jl >> 2, pp = jl & 3, dsel = (lane >> 4) & 1;
      typedef __attribute__((address_space(3))) s16x4* lds_p;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        if (k0 + 16 * sp >= T) continue;                       // every key of this slab is padding (uniform): P is 0 there
        bf16x8 pf[3];
        split8(f32x4{s[8 * sp], s[8 * sp + 1], s[8 * sp + 2], s[8 * sp + 3]}, f32x4{s[8 * sp + 4], s[8 * sp + 5], s[8 * sp + 6], s[8 * sp + 7]}, pf);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          bf16x8 vf[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int chunk = 4 * hh + 2 * dsel + (pp >> 1);
            const int key_a = 16 * sp + 4 * half + qq, key_b = key_a + 8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(vp + q * KVP + vplane_off(key_a, chunk) + 8 * (pp & 1)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(vp + q * KVP + vplane_off(key_b, chunk) + 8 * (pp & 1)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            vf[q] = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
          }
          f32x16& o = hh ? o_hi : o_lo;
#define SSV_MM(P, Q_) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[P], pf[Q_], o, 0, 0, 0)
          SSV_MM(2, 0); SSV_MM(0, 2); SSV_MM(1, 1); SSV_MM(1, 0); SSV_MM(0, 1); SSV_MM(0, 0);
#undef SSV_MM
        }
      }
    }
    if (more) store_kv(buf ^ 1);
    __syncthreads();
  }
  const bool valid = q0 + c < T;
  if (active) store_tile_T_rows(reinterpret_cast<float*>(skv) + wave * 2048, O + tok0 * ldo + h * DH, ldo, q0, min(32, T - q0), lane, o_lo, o_hi, 1.f / l);
  if (valid && half == 0) LSE[((int64_t)b * heads + h) * T + q0 + c] = (m + log2f(l)) * LN2;
}

// ------------------------------------------------------------------------------------------------ attention backward: dQ (and delta)
template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dq_k(int T, int heads, const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, int ld, float scale,
                                                         const float* __restrict__ O, const float* __restrict__ dO, int ldo,
                                                         const float* __restrict__ LSE, float* __restrict__ DELTA,
                                                         float* __restrict__ dQ, int ldg) {
  __shared__ float sk[2][32 * TS], sv[2][32 * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < T;
  const int64_t tok0 = (int64_t)b * T;
  const int qrow = min(q0 + c, T - 1);
  float qreg[32], doreg[32];
  load_row32(Q + (tok0 + qrow) * ld + h * DH + half * 32, qreg, scale * LOG2E);
  load_row32(dO + (tok0 + qrow) * ldo + h * DH + half * 32, doreg, 1.f);
  float delta = 0.f;
  {
    const float* op = O + (tok0 + qrow) * ldo + h * DH + half * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = *(const f32x4*)(op + 4 * i);
      delta += v[0] * doreg[4 * i] + v[1] * doreg[4 * i + 1] + v[2] * doreg[4 * i + 2] + v[3] * doreg[4 * i + 3];
    }
    delta += __shfl_xor(delta, 32, 64);
  }
  const int64_t stat = ((int64_t)b * heads + h) * T;
  const float lse = LSE[stat + qrow] * LOG2E;
  const bool valid = q0 + c < T;
  if (valid && half == 0) DELTA[stat + q0 + c] = delta;
  f32x16 g_lo = zero16(), g_hi = zero16();
  const float* kbase = K + tok0 * ld + h * DH;
  const float* vbase = V + tok0 * ld + h * DH;
  Stage<NW> st;
  st.load(kbase, ld, vbase, ld, 0, T);
  st.store(sk[0], sv[0]);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < T;
    if (more) st.load(kbase, ld, vbase, ld, k0 + 32, T);
    if (active) {
      float kreg[32];
      f32x16 s = zero16(), dp = zero16();
      lds_row32(sk[buf], c, half, kreg);
#pragma unroll
      for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[i], qreg[i], s, 0, 0, 0);
      lds_row32(sv[buf], c, half, kreg);
#pragma unroll
      for (int i = 0; i < 32; ++i) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[i], doreg[i], dp, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = (k0 + rowof(j, half) < T) ? ex2(s[j] - lse) : 0.f;
        s[j] = p * (dp[j] - delta);                            // dS^T
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k0 + rowof(j, 0) >= T) continue;                   // padding keys: dS is 0
        const float* kr = sk[buf] + rowof(j, half) * TS;
        g_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[c], s[j], g_lo, 0, 0, 0);
        g_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32 + c], s[j], g_hi, 0, 0, 0);
      }
    }
    if (more) st.store(sk[buf ^ 1], sv[buf ^ 1]);
    __syncthreads();
  }
  store_tile_T(dQ + tok0 * ldg + h * DH, ldg, q0 + c, valid, half, g_lo, g_hi, scale);
}

// ------------------------------------------------------------------------------------------------ attention backward: dK, dV
template <int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_bwd_dkv_k(int T, int heads, const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, int ld, float scale,
                                                          const float* __restrict__ dO, int ldo, const float* __restrict__ LSE,
                                                          const float* __restrict__ DELTA, float* __restrict__ dK, float* __restrict__ dV, int ldg) {
  __shared__ float sq[2][32 * TS], sd[2][32 * TS], sstat[2][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int k0 = (blockIdx.x * NW + wave) * 32, h = blockIdx.y, b = blockIdx.z;
  const bool active = k0 < T;
  const int64_t tok0 = (int64_t)b * T;
  const int krow = min(k0 + c, T - 1);
  const bool kvalid = k0 + c < T;
  float kreg[32], vreg[32];
  load_row32(K + (tok0 + krow) * ld + h * DH + half * 32, kreg, 1.f);
  load_row32(V + (tok0 + krow) * ld + h * DH + half * 32, vreg, 1.f);
  const int64_t stat = ((int64_t)b * heads + h) * T;
  f32x16 dk_lo = zero16(), dk_hi = zero16(), dv_lo = zero16(), dv_hi = zero16();
  const float* qbase = Q + tok0 * ld + h * DH;
  const float* dobase = dO + tok0 * ldo + h * DH;
  Stage<NW> st;
  float st_stat = 0.f;                                        // threads 0..31: lse, 32..63: delta of the staged query rows
  auto load_stat = [&](int q0) {
    if (threadIdx.x < 64) st_stat = (threadIdx.x < 32 ? LSE : DELTA)[stat + min(q0 + (int)(threadIdx.x & 31), T - 1)] * (threadIdx.x < 32 ? LOG2E : 1.f);
  };
  st.load(qbase, ld, dobase, ldo, 0, T);
  load_stat(0);
  st.store(sq[0], sd[0]);
  if (threadIdx.x < 64) sstat[0][threadIdx.x >> 5][threadIdx.x & 31] = st_stat;
  __syncthreads();
  int buf = 0;
  for (int q0 = 0; q0 < T; q0 += 32, buf ^= 1) {
    const bool more = q0 + 32 < T;
    if (more) { st.load(qbase, ld, dobase, ldo, q0 + 32, T); load_stat(q0 + 32); }
    if (active) {
      float qreg[32];
      f32x16 s = zero16(), dp = zero16();
      lds_row32(sq[buf], c, half, qreg);
#pragma unroll
      for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[i] * (scale * LOG2E), kreg[i], s, 0, 0, 0);   // rows = query, col = key
      lds_row32(sd[buf], c, half, qreg);
#pragma unroll
      for (int i = 0; i < 32; ++i) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[i], vreg[i], dp, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = rowof(j, half);
        const float p = (q0 + r < T && kvalid) ? ex2(s[j] - sstat[buf][0][r]) : 0.f;
        s[j] = p;                                              // P
        dp[j] = p * (dp[j] - sstat[buf][1][r]);                // dS
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (q0 + rowof(j, 0) >= T) continue;                   // padding queries: P and dS are 0
        const float* qr = sq[buf] + rowof(j, half) * TS;
        const float* dr = sd[buf] + rowof(j, half) * TS;
        dv_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[c], s[j], dv_lo, 0, 0, 0);
        dv_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[32 + c], s[j], dv_hi, 0, 0, 0);
        dk_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[c], dp[j], dk_lo, 0, 0, 0);
        dk_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[32 + c], dp[j], dk_hi, 0, 0, 0);
      }
    }
    if (more) {
      st.store(sq[buf ^ 1], sd[buf ^ 1]);
      if (threadIdx.x < 64) sstat[buf ^ 1][threadIdx.x >> 5][threadIdx.x & 31] = st_stat;
    }
    __syncthreads();
  }
  store_tile_T(dK + tok0 * ldg + h * DH, ldg, k0 + c, kvalid && active, half, dk_lo, dk_hi, scale);
  store_tile_T(dV + tok0 * ldg + h * DH, ldg, k0 + c, kvalid && active, half, dv_lo, dv_hi, 1.f);
}

// ------------------------------------------------------------------------------------------------ attention backward, ONE pass (T <= 256)
// The two-pass form above recomputes S = Q K^T and dP = dO V^T twice (once per pass): 7 products per (query tile, key tile) pair for the 4 the
// backward needs + 1.  Here ONE workgroup holds every key tile of an (image, head) - wave w owns keys [32 w, 32 w + 32): its V fragment and its
// dK / dV accumulators live in registers, all K rows of the head sit in LDS - and walks the query tiles once: S and dP are computed once,
// dV += P^T dO and dK += dS^T Q accumulate per wave, and the query gradient dQ_i = sum over the key tiles of dS_ij K_j is reduced ACROSS the
// waves through LDS in fixed wave order (deterministic).  dS has the key on the lane and the query on the register index; the dQ product
// contracts over keys, so each wave transposes its 32 x 32 dS tile through its own LDS scratch (written row = query, read back 4 keys per
// ds_read_b128), which afterwards takes the wave's dQ partial (register-major, so that the reduction reads one b128 per wave and group).
// 5 products instead of 7 per tile pair; delta = rowsum(O * dO) is formed while the query rows are staged (no extra pass, DELTA is still
// written for callers that want it).  Per query tile: two barriers (all partials written -> reduce + restage -> next tile).
#ifdef SSV_STAMP_ATTN   // diagnostic build only (tools/stamp_attn.py): cycles of one wave per phase of a query tile.  Never in the shipped library.
__device__ unsigned long long g_attn_stamps[16];
#define ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                       asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); tph[i] += now_ - last_; last_ = now_; } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif
template <int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 4 ? 1 : 2, 2))) attn_bwd_fused_k(int T, int heads, const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, int ld, float scale,
                                                            const float* __restrict__ O, const float* __restrict__ dO, int ldo,
                                                            const float* __restrict__ LSE, float* __restrict__ DELTA,
                                                            float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV, int ldg) {
  __shared__ __attribute__((aligned(16))) float sK[NW * 32 * TS];      // every key row of this (image, head)
  __shared__ __attribute__((aligned(16))) float sq[32 * TS], sd[32 * TS];
  __shared__ __attribute__((aligned(16))) float sstat[2][32];           // lse (log2 units) | delta of the staged query rows
  __shared__ __attribute__((aligned(16))) float scr[NW][2048];          // per wave: dS^T staging (32 x 36), then its dQ partial (32 registers x 64 lanes)
  constexpr int N4 = 512 / (NW * 64);                                   // float4 per thread per staged 32 x 64 tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int k0 = wave * 32;
  const bool active = k0 < T;
  const int nact = (T + 31) >> 5;                                       // waves that own keys
  const int64_t tok0 = (int64_t)b * T;
  const int krow = min(k0 + c, T - 1);
  const bool kvalid = k0 + c < T;
  const int64_t stat = ((int64_t)b * heads + h) * T;
  const float* qbase = Q + tok0 * ld + h * DH;
  const float* kbase = K + tok0 * ld + h * DH;
  const float* obase = O + tok0 * ldo + h * DH;
  const float* dobase = dO + tok0 * ldo + h * DH;
  // all keys -> LDS (rows past T repeat the last row: their probabilities are forced to 0 below)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
    *(f32x4*)(sK + r * TS + c4) = *(const f32x4*)(kbase + (int64_t)min(r, T - 1) * ld + c4);
  }
  float vreg[32];
  load_row32(V + (tok0 + krow) * ld + h * DH + half * 32, vreg, 1.f);
  f32x16 dk_lo = zero16(), dk_hi = zero16(), dv_lo = zero16(), dv_hi = zero16();
  // staging of one query tile: Q and dO rows to registers, delta = sum_d O * dO reduced over the 16 lanes that share a row
  // load_tile only ISSUES the loads (nothing in it depends on a loaded value, so the wave does not wait for them there - the round-3 form computed delta inside it
  // and paid the global latency once per query tile: 9 % of the tile's cycles, tools/stamp_attn.py); delta is formed when the rows are written to LDS.
  // Workgroups of two waves (T <= 64: four float4 per thread and tensor) do not prefetch: the 48 staging registers spilled, and every reload of a spilled
  // value waits for ALL outstanding loads (s_waitcnt vmcnt(0)) - the prefetch was paid in the middle of the MFMA loops instead of hidden under them.
  constexpr bool PREFETCH = NW > 2;
  f32x4 ra[N4], rb[N4], ro[N4];
  float rlse[N4];
  // (two-wave workgroups whose second - last - tile has at most 8 query rows, the 37-token local crops: those rows are one float4 per thread and tensor; they are
  //  fetched with the first tile, so the workgroup pays ONE global round trip instead of two - with 1.5 waves per SIMD resident nothing else hides the second)
  const bool small_next = NW == 2 && T > 32 && T <= 40;       // uniform
  auto load_tile = [&](int q0, int ni = 512 / (NW * 64)) {    // ni: staged float4s per thread and tensor (rows 8 ni.. keep what the previous tile left: finite)
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      if (i >= ni) break;
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
      const int64_t row = min(q0 + r, T - 1);
      ra[i] = *(const f32x4*)(qbase + row * ld + c4);
      rb[i] = *(const f32x4*)(dobase + row * ldo + c4);
      ro[i] = *(const f32x4*)(obase + row * ldo + c4);
      rlse[i] = LSE[stat + row];
    }
  };
  auto store_tile = [&](int q0, int ni = 512 / (NW * 64)) {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      if (i >= ni) break;
      const int e = i * NW * 64 + threadIdx.x, r = e >> 4, c4 = (e & 15) * 4;
      *(f32x4*)(sq + r * TS + c4) = ra[i];
      *(f32x4*)(sd + r * TS + c4) = rb[i];
      float d = ro[i][0] * rb[i][0] + ro[i][1] * rb[i][1] + ro[i][2] * rb[i][2] + ro[i][3] * rb[i][3];
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
      if ((e & 15) == 0) {
        sstat[0][r] = rlse[i] * LOG2E; sstat[1][r] = d;
        if (q0 + r < T) DELTA[stat + q0 + r] = d;
      }
    }
  };
  load_tile(0);
  store_tile(0);
  if (small_next) load_tile(32, 1);
  __syncthreads();
  float* myscr = scr[wave];
#ifdef SSV_STAMP_ATTN
  unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  for (int q0 = 0; q0 < T; q0 += 32) {
    const bool more = q0 + 32 < T;
    if (PREFETCH && more) load_tile(q0 + 32);               // next tile's rows fly under this tile's MFMAs
    ASTAMP(0);
    if (active) {
      float fr[32];
      f32x16 s = zero16(), dp = zero16();
      {
        float kf[32];
        lds_row32(sq, c, half, fr);
        lds_row32(sK + k0 * TS, c, half, kf);
#pragma unroll
        for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[i] * (scale * LOG2E), kf[i], s, 0, 0, 0);   // rows = query, col = key
      }
      lds_row32(sd, c, half, fr);
#pragma unroll
      for (int i = 0; i < 32; ++i) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[i], vreg[i], dp, 0, 0, 0);
      ASTAMP(1);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {                         // rows 8 jj + 4 half + {0..3}: their statistics are one ds_read_b128 each
        const f32x4 l4 = *(const f32x4*)&sstat[0][8 * jj + 4 * half], d4 = *(const f32x4*)&sstat[1][8 * jj + 4 * half];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * jj + e, r = rowof(j, half);
          const float p = (q0 + r < T && kvalid) ? ex2(s[j] - l4[e]) : 0.f;
          s[j] = p;                                            // P
          dp[j] = p * (dp[j] - d4[e]);                         // dS
        }
      }
      ASTAMP(2);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (q0 + rowof(j, 0) >= T) continue;                   // padding queries: P and dS are 0
        const float* qr = sq + rowof(j, half) * TS;
        const float* dr = sd + rowof(j, half) * TS;
        dv_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[c], s[j], dv_lo, 0, 0, 0);
        dv_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[32 + c], s[j], dv_hi, 0, 0, 0);
        dk_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[c], dp[j], dk_lo, 0, 0, 0);
        dk_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[32 + c], dp[j], dk_hi, 0, 0, 0);
      }
      ASTAMP(3);
      // dS (query on the register index, key on the lane) -> dS^T (key on the register index, query on the lane) through this wave's scratch
#pragma unroll
      for (int j = 0; j < 16; ++j) myscr[rowof(j, half) * 36 + c] = dp[j];
      f32x16 st_;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 t = *(const f32x4*)(myscr + c * 36 + 8 * jj + 4 * half);
        st_[4 * jj] = t[0]; st_[4 * jj + 1] = t[1]; st_[4 * jj + 2] = t[2]; st_[4 * jj + 3] = t[3];
      }
      ASTAMP(4);
      f32x16 g_lo = zero16(), g_hi = zero16();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k0 + rowof(j, 0) >= T) continue;                   // both keys of this step are padding (wave-uniform): dS is 0 there
        const float* kr = sK + (k0 + rowof(j, half)) * TS;
        g_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[c], st_[j], g_lo, 0, 0, 0);
        g_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32 + c], st_[j], g_hi, 0, 0, 0);
      }
      ASTAMP(5);
      // this wave's dQ partial, group-major: group G = registers 4 G' .. 4 G' + 3 of lo (G < 4) / hi, one b128 per lane and group
#pragma unroll
      for (int G = 0; G < 4; ++G) {                            // slot of (G, lane): G * 64 + (lane ^ d4), d4 = 2 G + half = the 16-byte piece of the query's row
        *(f32x4*)(myscr + (G * 64 + (lane ^ (2 * G + half))) * 4) = f32x4{g_lo[4 * G], g_lo[4 * G + 1], g_lo[4 * G + 2], g_lo[4 * G + 3]};
        *(f32x4*)(myscr + ((G + 4) * 64 + (lane ^ (2 * (G + 4) + half))) * 4) = f32x4{g_hi[4 * G], g_hi[4 * G + 1], g_hi[4 * G + 2], g_hi[4 * G + 3]};
      }
    }
    ASTAMP(6);
    __syncthreads();                                           // every partial is in LDS; nobody reads sq / sd any more
    ASTAMP(7);
    // the next tile's rows go to LDS BEFORE this tile's dQ stores are issued: waiting for the (older) prefetched loads must not also wait for younger stores
    // (vmcnt retires in order; the scattered 16-byte stores take far longer than anything else between the two barriers - tools/stamp_attn.py)
    if (!PREFETCH && more) {
      if (small_next) store_tile(q0 + 32, 1);
      else { load_tile(q0 + 32); store_tile(q0 + 32); }
    } else if (more) store_tile(q0 + 32);
    ASTAMP(11);
    // dQ of this query tile: sum over the key tiles in wave order; thread (query q, piece d4) - sixteen lanes store one whole 256-byte row.  The partials'
    // slots are XOR-swizzled (above) so that these row-major reads hit sixteen distinct bank quads.
    for (int i = threadIdx.x; i < 512; i += NW * 64) {
      const int q = i >> 4, d4 = i & 15;
      const int slot = ((d4 >> 1) * 64 + ((q + 32 * (d4 & 1)) ^ d4)) * 4;
      f32x4 acc = *(const f32x4*)(scr[0] + slot);
      for (int w = 1; w < nact; ++w) acc += *(const f32x4*)(scr[w] + slot);
      if (q0 + q < T) *(f32x4*)(dQ + (tok0 + q0 + q) * ldg + h * DH + 4 * d4) = acc * scale;
    }
    ASTAMP(8);
    __syncthreads();
    ASTAMP(9);
  }
#ifdef SSV_STAMP_ATTN
  if (lane == 0 && (wave == 0 || wave == 5)) {               // one wave of each SIMD pair's halves; [10] counts the stamped (wave, tile) pairs
    for (int i = 0; i < 12; ++i) if (i != 10) atomicAdd(&g_attn_stamps[i], tph[i]);
    atomicAdd(&g_attn_stamps[10], (unsigned long long)((T + 31) / 32));
  }
#endif
  // (past the loop's last barrier every dQ partial has been read: the wave's scratch takes its dK, then its dV tile - whole rows to global memory)
  if (active) {
    store_tile_T_rows(myscr, dK + tok0 * ldg + h * DH, ldg, k0, min(32, T - k0), lane, dk_lo, dk_hi, scale);
    store_tile_T_rows(myscr, dV + tok0 * ldg + h * DH, ldg, k0, min(32, T - k0), lane, dv_lo, dv_hi, 1.f);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wavefront per row, lane owns columns 4*lane + 256*it, the row held in registers: x is read from memory ONCE (rounds 1-3 walked it three times - mean, variance,
// output - and every walk paid a memory latency with one row per wave in flight: 2.3-2.9 TB/s on ViT-S shapes).  Each wave takes LNF_R rows at a time and issues every
// load of the group (x and the addend) before the first reduction, so two rows' worth of bytes per wave are in flight.  Arithmetic and summation order are those of the
// three-pass form (two-pass variance about the mean), so the results are the same bits.
constexpr int LNF_R = 2;               // rows per wave and trip
constexpr int LNF_TRIPS = 2;           // trips per wave: 16 rows per workgroup

template <int NIT>
__global__ void __launch_bounds__(256) ln_fwd_k(int64_t M, int C, const float* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const float* __restrict__ addend, float eps,
                                                float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[NIT], bb[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane * 4 + 256 * it;
    g[it] = f32x4{0.f, 0.f, 0.f, 0.f}; bb[it] = g[it];
    if (i < C) { g[it] = *(const f32x4*)(gamma + i); bb[it] = *(const f32x4*)(beta + i); }
  }
#pragma unroll 1
  for (int trip = 0; trip < LNF_TRIPS; ++trip) {
    const int64_t r0 = (((int64_t)blockIdx.x * LNF_TRIPS + trip) * 4 + wave) * LNF_R;
    if (r0 >= M) return;
    f32x4 v[LNF_R][NIT], a[LNF_R][NIT];
#pragma unroll
    for (int k = 0; k < LNF_R; ++k) {
      const int64_t r = min(r0 + k, M - 1);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = lane * 4 + 256 * it;
        v[k][it] = f32x4{0.f, 0.f, 0.f, 0.f}; a[k][it] = v[k][it];
        if (i < C) {
          v[k][it] = *(const f32x4*)(x + r * C + i);
          if (addend) a[k][it] = *(const f32x4*)(addend + r * C + i);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < LNF_R; ++k) {
      const int64_t r = r0 + k;
      if (r >= M) break;
      float s = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) s += (v[k][it][0] + v[k][it][1]) + (v[k][it][2] + v[k][it][3]);       // columns past C hold zeros
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (lane * 4 + 256 * it < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[k][it][e] - mean; q += d * d; }
        }
      const float invstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = lane * 4 + 256 * it;
        if (i < C) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[k][it][e] - mean) * invstd * g[it][e] + bb[it][e];
          if (addend) o += a[k][it];
          *(f32x4*)(y + r * C + i) = o;
        }
      }
      if (lane == 0) { mean_out[r] = mean; invstd_out[r] = invstd; }
    }
  }
}

constexpr int LN_ROWS = 32;            // rows per block in the backward (4 waves x 8 rows): ~3000 workgroups on ViT-S shapes instead of ~400 (the kernel was latency-bound at 1.5 workgroups per CU)

// The next row's dy / x / statistics / addend are loaded before the current row's two reductions (round 4: one row per wave in flight left the pass at ~2.5 TB/s).
template <int NIT>
__global__ void __launch_bounds__(256) ln_bwd_k(int64_t M, int C, const float* __restrict__ dy, const float* __restrict__ x,
                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                const float* __restrict__ invstd, const float* __restrict__ dx_addend,
                                                float* __restrict__ dx, float* __restrict__ partial /* [blocks][2][C] */) {
  __shared__ float red[4][2][NIT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 ag[NIT], ab[NIT], gm[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane * 4 + 256 * it;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[it][e] = 0.f; ab[it][e] = 0.f; gm[it][e] = 0.f; }
    if (i < C) gm[it] = *(const f32x4*)(gamma + i);
  }
  const int64_t r0 = (int64_t)blockIdx.x * LN_ROWS + wave * (LN_ROWS / 4);
  struct Row { f32x4 d[NIT], v[NIT], a[NIT]; float mu, is; };
  auto fetch = [&](Row& w, int64_t r) {
    r = min(r, M - 1);
    w.mu = mean[r]; w.is = invstd[r];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane * 4 + 256 * it;
      w.d[it] = f32x4{0.f, 0.f, 0.f, 0.f}; w.v[it] = w.d[it]; w.a[it] = w.d[it];
      if (i < C) {
        w.d[it] = *(const f32x4*)(dy + r * C + i); w.v[it] = *(const f32x4*)(x + r * C + i);
        if (dx_addend) w.a[it] = *(const f32x4*)(dx_addend + r * C + i);
      }
    }
  };
  Row cur, nxt;
  fetch(cur, r0);
#pragma unroll 1
  for (int rr = 0; rr < LN_ROWS / 4; ++rr) {
    const int64_t r = r0 + rr;
    if (r >= M) break;
    if (rr + 1 < LN_ROWS / 4) fetch(nxt, r + 1);
    const float mu = cur.mu, is = cur.is;
    f32x4 gv[NIT], xh[NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane * 4 + 256 * it;
      if (i < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[it][e] = (cur.v[it][e] - mu) * is;
          gv[it][e] = cur.d[it][e] * gm[it][e];
          s1 += gv[it][e]; s2 += gv[it][e] * xh[it][e];
          ag[it][e] += cur.d[it][e] * xh[it][e]; ab[it][e] += cur.d[it][e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane * 4 + 256 * it;
      if (i < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = is * (gv[it][e] - s1 - xh[it][e] * s2);
        if (dx_addend) o += cur.a[it];
        *(f32x4*)(dx + r * C + i) = o;
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][0][lane * 4 + 256 * it + e] = ag[it][e];
      red[wave][1][lane * 4 + 256 * it + e] = ab[it][e];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    partial[((int64_t)blockIdx.x * 2 + 0) * C + i] = (red[0][0][i] + red[1][0][i]) + (red[2][0][i] + red[3][0][i]);
    partial[((int64_t)blockIdx.x * 2 + 1) * C + i] = (red[0][1][i] + red[1][1][i]) + (red[2][1][i] + red[3][1][i]);
  }
}

// 8 columns x 32 groups of partial blocks per workgroup (four loads in flight per lane), combined in fixed order: deterministic
__global__ void __launch_bounds__(256) ln_bwd_finalize_k(int nblocks, int C, const float* __restrict__ partial, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate) {
  __shared__ double sh[32][8];
  const int col_in = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + col_in;                      // over 2*C
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int which = 0, col = 0;
  if (i < 2 * C) {
    which = i / C; col = i - which * C;
    const int per = (nblocks + 31) / 32;
    const int b0 = min(grp * per, nblocks), b1 = min(b0 + per, nblocks);
    const float* p = partial + (int64_t)which * C + col;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      a0 += (double)p[(int64_t)b * 2 * C]; a1 += (double)p[(int64_t)(b + 1) * 2 * C];
      a2 += (double)p[(int64_t)(b + 2) * 2 * C]; a3 += (double)p[(int64_t)(b + 3) * 2 * C];
    }
    for (; b < b1; ++b) a0 += (double)p[(int64_t)b * 2 * C];
  }
  sh[grp][col_in] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += sh[g][col_in];
    float* out = which ? dbeta : dgamma;
    out[col] = accumulate ? out[col] + (float)t : (float)t;
  }
}

// ------------------------------------------------------------------------------------------------ GELU (erf form, nn.GELU default)
__global__ void gelu_fwd_k(int64_t n4, const f32x4* __restrict__ x, f32x4* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = x[i];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = ssv_gelu(v[e]);
  y[i] = o;
}
__global__ void gelu_bwd_k(int64_t n4, const f32x4* __restrict__ x, const f32x4* __restrict__ dy, f32x4* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = x[i], g = dy[i];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = g[e] * ssv_gelu_grad(v[e]);
  }
  dx[i] = o;
}

// ------------------------------------------------------------------------------------------------ token assembly
__global__ void vit_embed_fwd_k(int64_t total, int T, int F, int P3, int patch, int H, int W, const float* __restrict__ img,
                                const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ tok) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i % F);
  const int64_t bt = i / F;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  float v;
  if (f >= P3) v = pos[(int64_t)t * (F - P3) + (f - P3)];
  else if (t == 0) v = cls[f];
  else {
    const int pw = W / patch, pidx = t - 1, py = pidx / pw, px = pidx - py * pw;
    const int pp = patch * patch, ch = f / pp, rem = f - ch * pp, kh = rem / patch, kw = rem - kh * patch;
    v = img[((b * H + (py * patch + kh)) * (int64_t)W + (px * patch + kw)) * 3 + ch];      // NHWC image, (c, kh, kw) feature order
  }
  tok[i] = v;
}

// One workgroup per token row, four features per thread (round 4: the per-element form above spent a 64-bit division chain per float and wrote 4-byte stores -
// 1.1 TB/s on the 387 MB of a 512-image global-crop pass; it stays for patch sizes that are not multiples of 4).
// (patch % 4 == 0, patch <= 32, E % 4 == 0.)  The patch's pixels are read as whole float4s of the interleaved image rows (coalesced: a patch row is 3 * patch
// consecutive floats) into LDS and leave it in the (c, kh, kw) feature order; measured forms without the LDS step gather four scalars 12 bytes apart per thread
// and reach 1.7-2.1 TB/s.
constexpr int EMB_MAXP = 32;
__global__ void __launch_bounds__(256) vit_embed_fwd4_k(int T, int F, int P3, int patch, int H, int W, const float* __restrict__ img,
                                                        const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ tok) {
  __shared__ __attribute__((aligned(16))) float sp[3 * EMB_MAXP * EMB_MAXP];
  const int64_t bt = blockIdx.x;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  const int pw = W / patch, pidx = t - 1, py = pidx / pw, px = pidx - py * pw, pp = patch * patch, seg = 3 * patch, seg4 = seg >> 2;
  if (t != 0) {
    for (int i = threadIdx.x; i < (P3 >> 2); i += 256) {
      const int kh = i / seg4, j = i - kh * seg4;
      *(f32x4*)(sp + kh * seg + 4 * j) = *(const f32x4*)(img + ((b * H + (py * patch + kh)) * (int64_t)W + px * patch) * 3 + 4 * j);
    }
    __syncthreads();
  }
  for (int f = 4 * threadIdx.x; f < F; f += 1024) {
    f32x4 v;
    if (f >= P3) v = *(const f32x4*)(pos + (int64_t)t * (F - P3) + (f - P3));
    else if (t == 0) v = *(const f32x4*)(cls + f);
    else {
      const int ch = f / pp, rem = f - ch * pp, kh = rem / patch, kw = rem - kh * patch;      // four consecutive kw of one (c, kh): patch % 4 == 0
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = sp[kh * seg + (kw + e) * 3 + ch];
    }
    *(f32x4*)(tok + bt * F + f) = v;
  }
}

// dcls[f] = sum_b dtok[b][0][f] (f < P3);  dpos[t][e] = sum_b dtok[b][t][P3+e].  32 columns x 8 batch groups per workgroup, four independent double sums per
// thread (b = grp, grp + 8, ... in four interleaved chains), combined in fixed order: deterministic.  (Round 3: one thread per column walked the whole batch in
// one dependent chain - 0.4-0.6 ms per pass for 64-77 MB of reads.)
__global__ void __launch_bounds__(256) vit_embed_bwd_k(int B, int T, int F, int P3, const float* __restrict__ dtok, float* __restrict__ dcls,
                                                       float* __restrict__ dpos, int accumulate) {
  __shared__ double sh[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int E = F - P3;
  const int a = blockIdx.x * 32 + col;                         // active columns: the P3 class-token features, then T x E positional ones
  const bool valid = a < P3 + T * E;
  int t = 0, f = 0;
  if (valid) {
    if (a < P3) f = a;
    else { const int r = a - P3; t = r / E; f = P3 + (r - t * E); }
  }
  const float* p = dtok + (int64_t)t * F + f;
  const int64_t bs = (int64_t)T * F;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (valid) {
    int bb = grp;
    for (; bb + 24 < B; bb += 32) {
      a0 += (double)p[bb * bs]; a1 += (double)p[(bb + 8) * bs]; a2 += (double)p[(bb + 16) * bs]; a3 += (double)p[(bb + 24) * bs];
    }
    for (; bb < B; bb += 8) a0 += (double)p[bb * bs];
  }
  sh[grp][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && valid) {
    double acc = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) acc += sh[g][col];
    float* out = a < P3 ? dcls + f : dpos + (int64_t)t * E + (f - P3);
    *out = accumulate ? *out + (float)acc : (float)acc;
  }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int ssv_attention_fwd(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                                 int32_t ld, float scale, float* o, int32_t ldo, float* lse, void* stream) {
  return ssv_attention_fwd_arith(B, T, heads, dh, q, k, v, ld, scale, o, ldo, lse, SSV_ARITH_F32_MFMA, stream);
}

extern "C" int ssv_attention_fwd_arith(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                                       int32_t ld, float scale, float* o, int32_t ldo, float* lse, int32_t arithmetic, void* stream) {
  SSV_REQUIRE(B > 0 && T > 0 && heads > 0 && q && k && v && o && lse, "ssv_attention_fwd: bad arguments");
  SSV_REQUIRE(arithmetic == SSV_ARITH_F32_MFMA || arithmetic == SSV_ARITH_BF16X3, "ssv_attention_fwd: unknown arithmetic %d", arithmetic);
  SSV_REQUIRE(dh == DH, "ssv_attention_fwd: head size must be %d (got %d)", DH, dh);
  SSV_REQUIRE(ld >= heads * dh && ldo >= heads * dh && ld % 4 == 0 && ldo % 4 == 0, "ssv_attention_fwd: row strides must cover heads*dh and be multiples of 4");
  SSV_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "ssv_attention_fwd: pointers must be 16-byte aligned");
  SSV_REQUIRE(B <= 65535 && heads <= 65535, "ssv_attention_fwd: grid too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_ATTN, s);
  // 37-token local crops run the 64-row tile too: a 16x16x4-tile kernel with one wavefront per (image, head) was measured at 0.161 ms against
  // 0.166 ms here (T 37, 2048 images; profiles/r03_attention_kernels.txt) - both sit at ~2.9 TB/s of q/k/v/o traffic, not on the padded MFMA work.
  if (arithmetic == SSV_ARITH_BF16X3) {
    if (T <= 64) hipLaunchKernelGGL(attn_fwd_sp_k<2>, dim3(cdiv(T, 64), heads, B), dim3(128), 0, s, T, heads, q, k, v, ld, scale, o, ldo, lse);
    else hipLaunchKernelGGL(attn_fwd_sp_k<4>, dim3(cdiv(T, 128), heads, B), dim3(256), 0, s, T, heads, q, k, v, ld, scale, o, ldo, lse);
  } else if (T <= 64) hipLaunchKernelGGL(attn_fwd_k<2>, dim3(cdiv(T, 64), heads, B), dim3(128), 0, s, T, heads, q, k, v, ld, scale, o, ldo, lse);
  else hipLaunchKernelGGL(attn_fwd_k<4>, dim3(cdiv(T, 128), heads, B), dim3(256), 0, s, T, heads, q, k, v, ld, scale, o, ldo, lse);
  SSV_CHECK_LAUNCH("attn_fwd_k");
  return SSV_OK;
}

extern "C" int ssv_attention_bwd(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                                 int32_t ld, float scale, const float* o, const float* dout, int32_t ldo, const float* lse,
                                 float* delta, float* dq, float* dk, float* dv, int32_t ldg, void* stream) {
  SSV_REQUIRE(B > 0 && T > 0 && heads > 0 && q && k && v && o && dout && lse && delta && dq && dk && dv, "ssv_attention_bwd: bad arguments");
  SSV_REQUIRE(dh == DH, "ssv_attention_bwd: head size must be %d (got %d)", DH, dh);
  SSV_REQUIRE(ld >= heads * dh && ldo >= heads * dh && ldg >= heads * dh && ld % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0,
              "ssv_attention_bwd: row strides must cover heads*dh and be multiples of 4");
  SSV_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
              "ssv_attention_bwd: pointers must be 16-byte aligned");
  SSV_REQUIRE(B <= 65535 && heads <= 65535, "ssv_attention_bwd: grid too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_ATTN, s);
#ifndef SSV_ATTN_TWO_PASS          // diagnostic builds only: the two-pass backward for every T
  if (T <= 256) {                   // one pass: a workgroup holds all key tiles of an (image, head), dQ reduced across its waves
    const dim3 grid(1, heads, B);
    if (T <= 64) hipLaunchKernelGGL(attn_bwd_fused_k<2>, grid, dim3(128), 0, s, T, heads, q, k, v, ld, scale, o, dout, ldo, lse, delta, dq, dk, dv, ldg);
    else if (T <= 128) hipLaunchKernelGGL(attn_bwd_fused_k<4>, grid, dim3(256), 0, s, T, heads, q, k, v, ld, scale, o, dout, ldo, lse, delta, dq, dk, dv, ldg);
    else hipLaunchKernelGGL(attn_bwd_fused_k<8>, grid, dim3(512), 0, s, T, heads, q, k, v, ld, scale, o, dout, ldo, lse, delta, dq, dk, dv, ldg);
    SSV_CHECK_LAUNCH("attn_bwd_fused_k");
    return SSV_OK;
  }
#endif
  if (T <= 64) {
    const dim3 grid(cdiv(T, 64), heads, B);
    hipLaunchKernelGGL(attn_bwd_dq_k<2>, grid, dim3(128), 0, s, T, heads, q, k, v, ld, scale, o, dout, ldo, lse, delta, dq, ldg);
    SSV_CHECK_LAUNCH("attn_bwd_dq_k");
    hipLaunchKernelGGL(attn_bwd_dkv_k<2>, grid, dim3(128), 0, s, T, heads, q, k, v, ld, scale, dout, ldo, lse, delta, dk, dv, ldg);
  } else {
    const dim3 grid(cdiv(T, 128), heads, B);
    hipLaunchKernelGGL(attn_bwd_dq_k<4>, grid, dim3(256), 0, s, T, heads, q, k, v, ld, scale, o, dout, ldo, lse, delta, dq, ldg);
    SSV_CHECK_LAUNCH("attn_bwd_dq_k");
    hipLaunchKernelGGL(attn_bwd_dkv_k<4>, grid, dim3(256), 0, s, T, heads, q, k, v, ld, scale, dout, ldo, lse, delta, dk, dv, ldg);
  }
  SSV_CHECK_LAUNCH("attn_bwd_dkv_k");
  return SSV_OK;
}

extern "C" int ssv_layernorm_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta, const float* addend,
                                 float eps, float* y, float* mean, float* invstd, void* stream) {
  SSV_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 2048 && x && gamma && beta && y && mean && invstd, "ssv_layernorm_fwd: bad arguments (C %% 4 == 0, C <= 2048)");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)addend | (uintptr_t)y) & 15) == 0, "ssv_layernorm_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_NORM, s);
  const dim3 grid((unsigned)cdiv64(M, 4 * LNF_R * LNF_TRIPS));
  if (C <= 512) hipLaunchKernelGGL(ln_fwd_k<2>, grid, dim3(256), 0, s, M, C, x, gamma, beta, addend, eps, y, mean, invstd);
  else if (C <= 1024) hipLaunchKernelGGL(ln_fwd_k<4>, grid, dim3(256), 0, s, M, C, x, gamma, beta, addend, eps, y, mean, invstd);
  else hipLaunchKernelGGL(ln_fwd_k<8>, grid, dim3(256), 0, s, M, C, x, gamma, beta, addend, eps, y, mean, invstd);
  SSV_CHECK_LAUNCH("ln_fwd_k");
  return SSV_OK;
}

extern "C" size_t ssv_layernorm_workspace_bytes(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)cdiv64(M, LN_ROWS) * 2 * (size_t)C * sizeof(float);
}

extern "C" int ssv_layernorm_bwd(int64_t M, int32_t C, const float* dy, const float* x, const float* gamma, const float* mean,
                                 const float* invstd, const float* dx_addend, float* dx, float* dgamma, float* dbeta,
                                 int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 2048 && dy && x && gamma && mean && invstd && dx && dgamma && dbeta && ws,
              "ssv_layernorm_bwd: bad arguments (C %% 4 == 0, C <= 2048)");
  SSV_REQUIRE(ws_bytes >= ssv_layernorm_workspace_bytes(M, C), "ssv_layernorm_bwd: workspace too small");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx_addend | (uintptr_t)dx) & 15) == 0, "ssv_layernorm_bwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_NORM, s);
  const int nblocks = (int)cdiv64(M, LN_ROWS);
  float* partial = (float*)ws;
  if (C <= 512) hipLaunchKernelGGL(ln_bwd_k<2>, dim3(nblocks), dim3(256), 0, s, M, C, dy, x, gamma, mean, invstd, dx_addend, dx, partial);
  else if (C <= 1024) hipLaunchKernelGGL(ln_bwd_k<4>, dim3(nblocks), dim3(256), 0, s, M, C, dy, x, gamma, mean, invstd, dx_addend, dx, partial);
  else hipLaunchKernelGGL(ln_bwd_k<8>, dim3(nblocks), dim3(256), 0, s, M, C, dy, x, gamma, mean, invstd, dx_addend, dx, partial);
  SSV_CHECK_LAUNCH("ln_bwd_k");
  hipLaunchKernelGGL(ln_bwd_finalize_k, dim3(cdiv(2 * C, 8)), dim3(256), 0, s, nblocks, C, partial, dgamma, dbeta, accumulate);
  SSV_CHECK_LAUNCH("ln_bwd_finalize_k");
  return SSV_OK;
}

extern "C" int ssv_gelu_fwd(int64_t n, const float* x, float* y, void* stream) {
  SSV_REQUIRE(n > 0 && n % 4 == 0 && x && y && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "ssv_gelu_fwd: bad arguments (n %% 4 == 0, 16-byte aligned)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(gelu_fwd_k, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, n / 4, (const f32x4*)x, (f32x4*)y);
  SSV_CHECK_LAUNCH("gelu_fwd_k");
  return SSV_OK;
}

extern "C" int ssv_gelu_bwd(int64_t n, const float* x, const float* dy, float* dx, void* stream) {
  SSV_REQUIRE(n > 0 && n % 4 == 0 && x && dy && dx && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "ssv_gelu_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(gelu_bwd_k, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, n / 4, (const f32x4*)x, (const f32x4*)dy, (f32x4*)dx);
  SSV_CHECK_LAUNCH("gelu_bwd_k");
  return SSV_OK;
}

extern "C" int ssv_vit_embed_fwd(int32_t B, int32_t H, int32_t W, int32_t patch, int32_t E, const float* img_nhwc,
                                 const float* cls, const float* pos, float* tokens, void* stream) {
  SSV_REQUIRE(B > 0 && H > 0 && W > 0 && patch > 0 && E >= 0 && H % patch == 0 && W % patch == 0 && img_nhwc && cls && pos && tokens,
              "ssv_vit_embed_fwd: bad arguments (image size must be a multiple of the patch size)");
  const int T = (H / patch) * (W / patch) + 1, P3 = 3 * patch * patch, F = P3 + E;
  const int64_t total = (int64_t)B * T * F;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  if (patch % 4 == 0 && patch <= EMB_MAXP && E % 4 == 0 && (((uintptr_t)img_nhwc | (uintptr_t)cls | (uintptr_t)pos | (uintptr_t)tokens) & 15) == 0 && (int64_t)B * T < (1ll << 31))
    hipLaunchKernelGGL(vit_embed_fwd4_k, dim3((unsigned)((int64_t)B * T)), dim3(256), 0, s, T, F, P3, patch, H, W, img_nhwc, cls, pos, tokens);
  else
    hipLaunchKernelGGL(vit_embed_fwd_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, total, T, F, P3, patch, H, W, img_nhwc, cls, pos, tokens);
  SSV_CHECK_LAUNCH("vit_embed_fwd_k");
  return SSV_OK;
}

extern "C" int ssv_vit_embed_bwd(int32_t B, int32_t T, int32_t P3, int32_t E, const float* dtokens, float* dcls, float* dpos,
                                 int32_t accumulate, void* stream) {
  SSV_REQUIRE(B > 0 && T > 0 && P3 > 0 && E >= 0 && dtokens && dcls && dpos, "ssv_vit_embed_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(vit_embed_bwd_k, dim3(cdiv(P3 + T * E, 32)), dim3(256), 0, s, B, T, P3 + E, P3, dtokens, dcls, dpos, accumulate);
  SSV_CHECK_LAUNCH("vit_embed_bwd_k");
  return SSV_OK;
}

#ifdef SSV_STAMP_ATTN
extern "C" int ssv_debug_attn_stamps(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_attn_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
