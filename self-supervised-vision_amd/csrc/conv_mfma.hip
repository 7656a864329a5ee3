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
// Block = 256 threads = 4 waves (one per SIMD), 128x128 (or 256x64 / 64x128) output tile, K-step 16,
// each wave owns a (TM x TN) grid of 32x32 accumulators.  Operands are staged global -> VGPR -> LDS
// with the next tile's loads issued before the current tile's MFMAs (register prefetch, one LDS
// buffer, ~20 KB) so that 3-4 blocks are resident per CU and hide each other's barriers.
//
// LDS operand images:
//   ROWK  [row][16 k + 4 pad]  - row-major, k contiguous (source is k-contiguous: NHWC channels / OHWI);
//                                fragments are ds_read_b128: lane l takes row (l&31), k = 4*(l>>5)..+3,
//                                the 4 values feed 4 consecutive MFMAs.  Row stride 20 floats makes the
//                                16-lane b128 groups conflict-free.
//   KROW  [16 k][rows]         - k-major (source is row-contiguous: dY / X rows for wgrad, W rows for dgrad);
//                                fragments are ds_read_b32, 32 consecutive banks per half-wave.
// Both operands of one MFMA always take the same k = 8*ks + 4*(l>>5) + t, so any mix is consistent.
#include "common.h"

namespace {

constexpr int BK = 16;
constexpr int LDT = 20;   // ROWK row stride (floats)

struct ConvKP {
  int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
  int M;              // fwd/wgrad: N*Ho*Wo
  int RSC;            // R*S*C
  FastDiv dHoWo, dWo, dC, dS;
};

template <int TM, int TN, bool A_ROWK, bool B_ROWK, int LDA, int LDB>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wr0, int wc0, int lane, f32x16 (&acc)[TM][TN]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
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

// =============================================================================================
// forward
// =============================================================================================
template <int BM, int BN, int WGM, int WGN, bool VEC>
__global__ void __launch_bounds__(256)
conv_fwd_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
           const float* addend, float* y) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
  float* As = smem;
  float* Bs = smem + BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave / WGN) * (BM / WGM), wc0 = (wave % WGN) * (BN / WGN);
  const int NT = (p.K + BN - 1) / BN;
  const int mt = blockIdx.x / NT, nt = blockIdx.x - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  if constexpr (VEC) {
    // ---- C % 16 == 0: every 16-wide k-tile lies inside one filter tap; float4 staging ----
    constexpr int AP = BM / 64, BP = BN / 64;
    const int chunk = (tid & 3) * 4, rsub = tid >> 2;
    int hi0[AP], wi0[AP];
    size_t abase[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + rsub + 64 * i;
      if (m < p.M) {
        const uint32_t n = fdiv((uint32_t)m, p.dHoWo);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Ho * p.Wo);
        const uint32_t ho = fdiv(rem, p.dWo);
        const uint32_t wo = rem - ho * (uint32_t)p.Wo;
        hi0[i] = (int)ho * p.stride - p.pad;
        wi0[i] = (int)wo * p.stride - p.pad;
        abase[i] = (size_t)n * p.H * p.W * p.C + chunk;
      } else { hi0[i] = -(1 << 28); wi0[i] = 0; abase[i] = 0; }
    }
    size_t bbase[BP];
    bool bok[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int ko = n0 + rsub + 64 * i;
      bok[i] = ko < p.K;
      bbase[i] = (size_t)(bok[i] ? ko : 0) * p.RSC + chunk;
    }
    int lr = 0, ls = 0, lc0 = 0;   // loader position (tap r, s, first channel)
    f32x4 ra[AP], rb[BP];
    auto load_tile = [&]() {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int hi = hi0[i] + lr, wi = wi0[i] + ls;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(x + abase[i] + ((size_t)hi * p.W + wi) * p.C + lc0);
        ra[i] = v;
      }
      const int tapoff = (lr * p.S + ls) * p.C + lc0;
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bok[i]) v = *reinterpret_cast<const f32x4*>(w + bbase[i] + tapoff);
        rb[i] = v;
      }
      lc0 += BK;
      if (lc0 >= p.C) { lc0 = 0; if (++ls == p.S) { ls = 0; ++lr; } }
    };
    auto store_tile = [&]() {
#pragma unroll
      for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[(rsub + 64 * i) * LDT + chunk]) = ra[i];
#pragma unroll
      for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[(rsub + 64 * i) * LDT + chunk]) = rb[i];
    };
    const int nkt = p.RSC / BK;
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;
      if (more) load_tile();
      mma_ktile<TM, TN, true, true, LDT, LDT>(As, Bs, wr0, wc0, lane, acc);
      __syncthreads();
      if (more) { store_tile(); __syncthreads(); }
    }
  } else {
    // ---- generic gather (any C; used by the 3-channel stem): scalar staging, k -> (r,s,c) per element ----
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
      mma_ktile<TM, TN, true, true, LDT, LDT>(As, Bs, wr0, wc0, lane, acc);
      __syncthreads();
      if (more) { store_tile(); __syncthreads(); }
    }
  }

  // ---- epilogue: acc reg j of lane l is (row (j&3)+8*(j>>2)+4*(l>>5), col l&31) of its 32x32 tile ----
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
template <int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256)
conv_dgrad_k(ConvKP p, const float* __restrict__ dy, const float* __restrict__ w, const float* addend, float* dx) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int A_FLOATS = BM * LDT, B_FLOATS = BK * BN;
  __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
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
  const int mt = blockIdx.x / NT, nt = blockIdx.x - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= Mc) return;

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

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  constexpr int AP = BM / 64;            // A passes: 64 rows x 4 float4 per pass
  constexpr int BCV = BN / 4;            // float4 per B row
  constexpr int BRP = 256 / BCV;         // B rows per pass
  constexpr int BP = BK / BRP;           // B passes
  const int chunk = (tid & 3) * 4, rsub = tid >> 2;
  int hq_[AP], wq_[AP], n_[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + rsub + 64 * i;
    if (m < Mc) {
      const int n = m / (Hq * Wq);
      const int rem = m - n * Hq * Wq;
      hq_[i] = rem / Wq; wq_[i] = rem - hq_[i] * Wq; n_[i] = n;
    } else { hq_[i] = -(1 << 28); wq_[i] = 0; n_[i] = 0; }
  }
  const int bcol = (tid % BCV) * 4, brow = tid / BCV;
  const bool bcok = n0 + bcol < p.C;

  int ti = 0, lk0 = 0;                   // loader position: tap index, first output channel
  int dho = 0, dwo = 0, tapoff = 0;
  if (ntaps > 0) { dho = taps[0]; dwo = taps[1]; tapoff = taps[2]; }
  f32x4 ra[AP], rb[BP];
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int ho = hq_[i] + dho, wo = wq_[i] + dwo;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo)
        v = *reinterpret_cast<const f32x4*>(dy + (((size_t)n_[i] * p.Ho + ho) * p.Wo + wo) * p.K + lk0 + chunk);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int ko = lk0 + brow + BRP * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bcok) v = *reinterpret_cast<const f32x4*>(w + (size_t)ko * p.RSC + tapoff + n0 + bcol);
      rb[i] = v;
    }
    lk0 += BK;
    if (lk0 >= p.K) {
      lk0 = 0; ++ti;
      if (ti < ntaps) { dho = taps[3 * ti]; dwo = taps[3 * ti + 1]; tapoff = taps[3 * ti + 2]; }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[(rsub + 64 * i) * LDT + chunk]) = ra[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow + BRP * i) * BN + bcol]) = rb[i];
  };
  const int nkt = ntaps * (p.K / BK);
  if (nkt > 0) {
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;
      if (more) load_tile();
      mma_ktile<TM, TN, true, false, LDT, BN>(As, Bs, wr0, wc0, lane, acc);
      __syncthreads();
      if (more) { store_tile(); __syncthreads(); }
    }
  }

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
template <int BM, int BN, int WGM, int WGN, bool VECB>
__global__ void __launch_bounds__(256)
conv_wgrad_k(ConvKP p, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int chunk_rows) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  __shared__ __attribute__((aligned(16))) float smem[BK * (BM + BN)];
  float* As = smem;               // [16][BM]
  float* Bs = smem + BK * BM;     // [16][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave / WGN) * (BM / WGM), wc0 = (wave % WGN) * (BN / WGN);
  const int JT = (p.RSC + BN - 1) / BN;
  const int it = blockIdx.x / JT, jt = blockIdx.x - it * JT;
  const int i0 = it * BM, j0 = jt * BN;
  const int split = blockIdx.y;
  const int ms = split * chunk_rows;
  const int me = min(ms + chunk_rows, p.M);

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  // A = dY rows (float4 along output channels)
  constexpr int ACV = BM / 4, ARP = 256 / ACV, AP = BK / ARP;
  const int acol = (tid % ACV) * 4, arow = tid / ACV;
  const bool acok = i0 + acol < p.K;
  // B = gathered X (float4 along input channels when C % 4 == 0, scalar otherwise)
  constexpr int BCV = VECB ? BN / 4 : BN, BRP = 256 / BCV, BP = BK / BRP;
  const int bcol = VECB ? (tid % BCV) * 4 : (tid % BCV), brow = tid / BCV;
  const int j = j0 + bcol;
  const bool jok = j < p.RSC;
  const uint32_t tap = fdiv((uint32_t)(jok ? j : 0), p.dC);
  const int cj = (jok ? j : 0) - (int)tap * p.C;
  const int rj = (int)fdiv(tap, p.dS);
  const int sj = (int)tap - rj * p.S;

  int mcur = ms;
  f32x4 ra[AP];
  f32x4 rbv[VECB ? BP : 1];
  float rbs[VECB ? 1 : BP];
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = mcur + arow + ARP * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (acok && m < me) v = *reinterpret_cast<const f32x4*>(dy + (size_t)m * p.K + i0 + acol);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int m = mcur + brow + BRP * i;
      bool ok = jok && m < me;
      size_t off = 0;
      if (ok) {
        const uint32_t n = fdiv((uint32_t)m, p.dHoWo);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Ho * p.Wo);
        const uint32_t ho = fdiv(rem, p.dWo);
        const uint32_t wo = rem - ho * (uint32_t)p.Wo;
        const int hi = (int)ho * p.stride - p.pad + rj, wi = (int)wo * p.stride - p.pad + sj;
        ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        off = (((size_t)n * p.H + hi) * p.W + wi) * p.C + cj;
      }
      if constexpr (VECB) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(x + off);
        rbv[i] = v;
      } else {
        rbs[i] = ok ? x[off] : 0.f;
      }
    }
    mcur += BK;
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[(arow + ARP * i) * BM + acol]) = ra[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      if constexpr (VECB) *reinterpret_cast<f32x4*>(&Bs[(brow + BRP * i) * BN + bcol]) = rbv[i];
      else Bs[(brow + BRP * i) * BN + bcol] = rbs[i];
    }
  };
  const int nkt = (me - ms + BK - 1) / BK;
  if (nkt > 0) {
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;
      if (more) load_tile();
      mma_ktile<TM, TN, false, false, BM, BN>(As, Bs, wr0, wc0, lane, acc);
      __syncthreads();
      if (more) { store_tile(); __syncthreads(); }
    }
  }

  float* out = partial + (size_t)split * p.K * p.RSC;
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
  return p;
}

struct WgradPlan { int bm, bn, it, jt, nsplit, chunk; };
WgradPlan plan_wgrad(const ssv_conv_desc* d) {
  WgradPlan w;
  const int RSC = d->R * d->S * d->C;
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  w.bm = d->K >= 128 ? 128 : 64;
  w.bn = 128;
  w.it = cdiv(d->K, w.bm);
  w.jt = cdiv(RSC, w.bn);
  const int tiles = w.it * w.jt;
  int64_t ns = cdiv64(768, tiles);            // ~3 workgroups per CU
  const int64_t max_by_rows = cdiv64(M, 256);
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  int64_t chunk = cdiv64(cdiv64(M, ns), BK) * BK;
  w.chunk = (int)chunk;
  w.nsplit = (int)cdiv64(M, chunk);
  return w;
}

}  // namespace

extern "C" int ssv_conv2d_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* addend, float* y, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_fwd")) return rc;
  SSV_REQUIRE(x && w && y, "ssv_conv2d_fwd: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "ssv_conv2d_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_FWD, s);
  const ConvKP p = make_kp(d);
  const bool vec = d->C % BK == 0;
  if (vec) {
    if (d->K >= 128) {
      const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 128));
      hipLaunchKernelGGL((conv_fwd_k<128, 128, 2, 2, true>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y);
    } else {
      const unsigned grid = (unsigned)(cdiv(p.M, 256) * cdiv(d->K, 64));
      hipLaunchKernelGGL((conv_fwd_k<256, 64, 4, 1, true>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y);
    }
  } else {
    const unsigned grid = (unsigned)(cdiv(p.M, 128) * cdiv(d->K, 64));
    hipLaunchKernelGGL((conv_fwd_k<128, 64, 2, 2, false>), dim3(grid), dim3(256), 0, s, p, x, w, bias, addend, y);
  }
  SSV_CHECK_LAUNCH("ssv_conv2d_fwd");
  return SSV_OK;
}

extern "C" int ssv_conv2d_dgrad(const ssv_conv_desc* d, const float* dy, const float* w, const float* addend,
                                float* dx, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_dgrad")) return rc;
  SSV_REQUIRE(dy && w && dx, "ssv_conv2d_dgrad: null pointer");
  SSV_REQUIRE((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0, "ssv_conv2d_dgrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % BK == 0 && d->C % 4 == 0, "ssv_conv2d_dgrad: needs K %% 16 == 0 and C %% 4 == 0 (got K=%d C=%d)", d->K, d->C);
  SSV_REQUIRE(d->stride <= 8, "ssv_conv2d_dgrad: stride too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_DGRAD, s);
  const ConvKP p = make_kp(d);
  const int st = d->stride;
  const int Hq = cdiv(d->H, st), Wq = cdiv(d->W, st);            // class (0,0) is the largest
  const int64_t Mc = (int64_t)d->N * Hq * Wq;
  if (d->C >= 128) {
    const unsigned gx = (unsigned)(cdiv64(Mc, 128) * cdiv(d->C, 128));
    hipLaunchKernelGGL((conv_dgrad_k<128, 128, 2, 2>), dim3(gx, st * st), dim3(256), 0, s, p, dy, w, addend, dx);
  } else {
    const unsigned gx = (unsigned)(cdiv64(Mc, 256) * cdiv(d->C, 64));
    hipLaunchKernelGGL((conv_dgrad_k<256, 64, 4, 1>), dim3(gx, st * st), dim3(256), 0, s, p, dy, w, addend, dx);
  }
  SSV_CHECK_LAUNCH("ssv_conv2d_dgrad");
  return SSV_OK;
}

extern "C" size_t ssv_conv2d_wgrad_workspace_bytes(const ssv_conv_desc* d) {
  if (!d || d->K <= 0 || d->C <= 0) return 0;
  const WgradPlan w = plan_wgrad(d);
  return (size_t)w.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
}

extern "C" int ssv_conv2d_wgrad(const ssv_conv_desc* d, const float* x, const float* dy, float* dw,
                                int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_desc(d, "ssv_conv2d_wgrad")) return rc;
  SSV_REQUIRE(x && dy && dw && ws, "ssv_conv2d_wgrad: null pointer");
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)ws) & 15) == 0, "ssv_conv2d_wgrad: pointers must be 16-byte aligned");
  SSV_REQUIRE(d->K % 4 == 0, "ssv_conv2d_wgrad: needs K %% 4 == 0 (got %d)", d->K);
  const WgradPlan wp = plan_wgrad(d);
  const size_t need = (size_t)wp.nsplit * d->K * d->R * d->S * d->C * sizeof(float);
  if (ws_bytes < need) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_CONV_WGRAD, s);
  const ConvKP p = make_kp(d);
  const bool vecb = d->C % 4 == 0;
  const dim3 grid((unsigned)(wp.it * wp.jt), (unsigned)wp.nsplit);
  float* part = (float*)ws;
  if (wp.bm == 128) {
    if (vecb) hipLaunchKernelGGL((conv_wgrad_k<128, 128, 2, 2, true>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk);
    else      hipLaunchKernelGGL((conv_wgrad_k<128, 128, 2, 2, false>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk);
  } else {
    if (vecb) hipLaunchKernelGGL((conv_wgrad_k<64, 128, 1, 4, true>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk);
    else      hipLaunchKernelGGL((conv_wgrad_k<64, 128, 1, 4, false>), grid, dim3(256), 0, s, p, x, dy, part, wp.chunk);
  }
  SSV_CHECK_LAUNCH("ssv_conv2d_wgrad(partial)");
  const int64_t n = (int64_t)d->K * p.RSC;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, (const float*)part, wp.nsplit, n, dw, accumulate);
  SSV_CHECK_LAUNCH("ssv_conv2d_wgrad(reduce)");
  return SSV_OK;
}
