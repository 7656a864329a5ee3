// Loss-side kernels: row L2 normalisation, NT-Xent (Gram matrix on fp32 MFMA + lane-local online
// log-sum-exp), the BYOL MSE pair, and small scalar helpers.
//
// NT-Xent layout trick: every 32x32 Gram tile is computed TRANSPOSED, T = Zc . Zr^T (A = column
// tile, B = this block's 32 rows), so that the accumulator puts the block's row r on the LANE
// (l & 31) and the 16 columns crow(j,h) = (j&3) + 8*(j>>2) + 4*(l>>5) in the lane's registers:
//   * the row-wise log-sum-exp is lane-local over registers and tiles; the only cross-lane step is
//     one shfl_xor(32) at the end (lanes l and l+32 hold the same row);
//   * in the backward the weight tile W (same layout) is directly the A operand of the second
//     product dZ += W . Zc (A[i = l&31][k = l>>5] = W[r][crow(j,h)] for MFMA step j), so the P
//     matrix never touches LDS or HBM.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ int crow(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// ---- F.normalize ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
l2norm_fwd_k(int rows, int D, const float* __restrict__ z, int normalize, float eps, float* __restrict__ zhat, int ldo, float* __restrict__ inv_norm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* zr = z + (size_t)row * D;
  float inv = 1.f;
  if (normalize) {
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = zr[d]; ss += v * v; }
    ss = wave_sum(ss);
    inv = 1.f / fmaxf(sqrtf(ss), eps);
  }
  for (int d = lane; d < ldo; d += 64) zhat[(size_t)row * ldo + d] = d < D ? zr[d] * inv : 0.f;
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}

__global__ void __launch_bounds__(256)
l2norm_bwd_k(int rows, int D, const float* __restrict__ zhat, int ldz, const float* __restrict__ inv_norm,
             const float* __restrict__ dzhat, int ldd, int normalize, float* __restrict__ dz) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* zh = zhat + (size_t)row * ldz;
  const float* dh = dzhat + (size_t)row * ldd;
  if (!normalize) { for (int d = lane; d < D; d += 64) dz[(size_t)row * D + d] = dh[d]; return; }
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += zh[d] * dh[d];
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int d = lane; d < D; d += 64) dz[(size_t)row * D + d] = (dh[d] - zh[d] * dot) * inv;
}

// ---- NT-Xent ---------------------------------------------------------------------------------------
__device__ __forceinline__ int global_row(int lr, int Bloc, int Nglob, int seg0) {
  return lr < Bloc ? seg0 + lr : Nglob + seg0 + (lr - Bloc);
}

// T = Zc_tile . Zr^T for one 32-column tile; zr[q] are this lane's register-resident row fragments
template <int DQ>
__device__ __forceinline__ f32x16 gram_tile(const float* __restrict__ Z, int ldz, int crow_glob, int h, const f32x4 (&zr)[DQ]) {
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const float* zc = Z + (size_t)crow_glob * ldz + 4 * h;
#pragma unroll
  for (int q = 0; q < DQ; ++q) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(zc + 8 * q);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], zr[q][t], acc, 0, 0, 0);
  }
  return acc;
}

template <int DQ>
__global__ void __launch_bounds__(256)
ntxent_fwd_k(int Nglob, int Bloc, int seg0, int ldz, const float* __restrict__ Z, float inv_temp, float* __restrict__ lse, float* __restrict__ pos,
             int tiles_per_split, float* __restrict__ part) {
  // grid (row blocks of 32, column splits): with gridDim.y > 1 a workgroup sweeps only its own run of column tiles and leaves the
  // running (max, sum, positive) of its rows in part[split][row][3]; ntxent_merge_k folds the splits in split order.
  __shared__ float sm_m[4][32], sm_s[4][32], sm_p[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int R2 = 2 * Nglob, rows_loc = 2 * Bloc;
  const int lr = blockIdx.x * 32 + l31;
  const bool rvalid = lr < rows_loc;
  const int rg = global_row(rvalid ? lr : 0, Bloc, Nglob, seg0);
  const int pg = rg < Nglob ? rg + Nglob : rg - Nglob;
  f32x4 zr[DQ];
#pragma unroll
  for (int q = 0; q < DQ; ++q) zr[q] = *reinterpret_cast<const f32x4*>(Z + (size_t)rg * ldz + 8 * q + 4 * h);

  float m = -INFINITY, s = 0.f, pv = 0.f;
  const int ntile = (R2 + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split, t1 = min(ntile, t0 + tiles_per_split);
  for (int ct = t0 + wave; ct < t1; ct += 4) {
    const int cl = ct * 32 + l31;
    const f32x16 T = gram_tile<DQ>(Z, ldz, cl < R2 ? cl : R2 - 1, h, zr);
    float sv[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cg = ct * 32 + crow(j, h);
      const float v = T[j] * inv_temp;
      if (cg == pg) pv += v;
      sv[j] = (cg < R2 && cg != rg) ? v : -INFINITY;
      tmax = fmaxf(tmax, sv[j]);
    }
    if (tmax > -INFINITY) {
      const float mn = fmaxf(m, tmax);
      float add = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) add += expf(sv[j] - mn);     // exp(-inf) = 0 for masked entries
      s = s * expf(m - mn) + add;
      m = mn;
    }
  }
  // merge the two half-waves (same row, disjoint columns)
  {
    const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(s, 32, 64);
    const float mn = fmaxf(m, m2);
    s = (m > -INFINITY ? s * expf(m - mn) : 0.f) + (m2 > -INFINITY ? s2 * expf(m2 - mn) : 0.f);
    m = mn;
    pv += __shfl_xor(pv, 32, 64);
  }
  if (h == 0) { sm_m[wave][l31] = m; sm_s[wave][l31] = s; sm_p[wave][l31] = pv; }
  __syncthreads();
  if (wave == 0 && h == 0 && rvalid) {
    float mm = sm_m[0][l31], ss = sm_s[0][l31], pp = sm_p[0][l31];
    for (int w = 1; w < 4; ++w) {
      const float m2 = sm_m[w][l31], s2 = sm_s[w][l31];
      const float mn = fmaxf(mm, m2);
      ss = (mm > -INFINITY ? ss * expf(mm - mn) : 0.f) + (m2 > -INFINITY ? s2 * expf(m2 - mn) : 0.f);
      mm = mn;
      pp += sm_p[w][l31];
    }
    if (gridDim.y == 1) {
      lse[lr] = mm + logf(ss);
      pos[lr] = pp;
    } else {
      float* p = part + ((size_t)blockIdx.y * rows_loc + lr) * 3;
      p[0] = mm; p[1] = ss; p[2] = pp;
    }
  }
}

// fold the column splits' (max, sum, positive) of every local row, in split order
__global__ void __launch_bounds__(256)
ntxent_merge_k(int rows_loc, int splits, const float* __restrict__ part, float* __restrict__ lse, float* __restrict__ pos) {
  const int lr = blockIdx.x * 256 + threadIdx.x;
  if (lr >= rows_loc) return;
  float mm = -INFINITY, ss = 0.f, pp = 0.f;
  for (int sp = 0; sp < splits; ++sp) {
    const float* p = part + ((size_t)sp * rows_loc + lr) * 3;
    const float m2 = p[0], s2 = p[1];
    const float mn = fmaxf(mm, m2);
    ss = (mm > -INFINITY ? ss * expf(mm - mn) : 0.f) + (m2 > -INFINITY ? s2 * expf(m2 - mn) : 0.f);
    mm = mn;
    pp += p[2];
  }
  lse[lr] = mm + logf(ss);
  pos[lr] = pp;
}

template <int DQ>
__global__ void __launch_bounds__(256)
ntxent_bwd_k(int Nglob, int Bloc, int seg0, int ldz, const float* __restrict__ Z, const float* __restrict__ lse_all,
             float inv_temp, float gscale, float* __restrict__ dZ, int tiles_per_split, float* __restrict__ part) {
  // grid (row blocks of 32, column splits): with gridDim.y > 1 the workgroup's sum over ITS column tiles goes to part[split][row][ldz]
  // and ntxent_bwd_reduce_k adds the splits in split order, subtracts the positive and scales.
  constexpr int DC = DQ / 4;                 // 32-wide chunks of the embedding dimension
  __shared__ float red[32][DQ * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int R2 = 2 * Nglob, rows_loc = 2 * Bloc;
  const int lr = blockIdx.x * 32 + l31;
  const bool rvalid = lr < rows_loc;
  const int rg = global_row(rvalid ? lr : 0, Bloc, Nglob, seg0);
  f32x4 zr[DQ];
#pragma unroll
  for (int q = 0; q < DQ; ++q) zr[q] = *reinterpret_cast<const f32x4*>(Z + (size_t)rg * ldz + 8 * q + 4 * h);
  const float lse_r = lse_all[rg];

  f32x16 out[DC];
#pragma unroll
  for (int dc = 0; dc < DC; ++dc)
#pragma unroll
    for (int j = 0; j < 16; ++j) out[dc][j] = 0.f;

  const int ntile = (R2 + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split, t1 = min(ntile, t0 + tiles_per_split);
  for (int ct = t0 + wave; ct < t1; ct += 4) {
    const int cl = ct * 32 + l31;
    const f32x16 T = gram_tile<DQ>(Z, ldz, cl < R2 ? cl : R2 - 1, h, zr);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cg = ct * 32 + crow(j, h);
      const int cgc = cg < R2 ? cg : R2 - 1;
      const float sv = T[j] * inv_temp;
      float wgt = 0.f;
      if (cg < R2 && cg != rg && rvalid) wgt = expf(sv - lse_r) + expf(sv - lse_all[cgc]);
      const float* zc = Z + (size_t)cgc * ldz + l31;
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) out[dc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wgt, zc[32 * dc], out[dc], 0, 0, 0);
    }
  }
  // fixed-order cross-wave sum through LDS: out[dc][j] of lane l is (row crow(j,h), col 32*dc + l31)
  for (int turn = 0; turn < 4; ++turn) {
    if (wave == turn) {
#pragma unroll
      for (int dc = 0; dc < DC; ++dc)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float* p = &red[crow(j, h)][32 * dc + l31];
          *p = (turn == 0 ? 0.f : *p) + out[dc][j];
        }
    }
    __syncthreads();
  }
  for (int e = tid; e < 32 * ldz; e += 256) {
    const int r = e / ldz, d = e - r * ldz;
    const int lrow = blockIdx.x * 32 + r;
    if (lrow < rows_loc) {
      if (gridDim.y == 1) {
        const int g = global_row(lrow, Bloc, Nglob, seg0);
        const int pg = g < Nglob ? g + Nglob : g - Nglob;
        dZ[(size_t)lrow * ldz + d] = gscale * (red[r][d] - 2.f * Z[(size_t)pg * ldz + d]);
      } else {
        part[((size_t)blockIdx.y * rows_loc + lrow) * ldz + d] = red[r][d];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
ntxent_bwd_reduce_k(int Nglob, int Bloc, int seg0, int ldz, int splits, const float* __restrict__ Z, const float* __restrict__ part,
                    float gscale, float* __restrict__ dZ) {
  const int rows_loc = 2 * Bloc;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rows_loc * ldz) return;
  const int lrow = e / ldz, d = e - lrow * ldz;
  float acc = 0.f;
  for (int sp = 0; sp < splits; ++sp) acc += part[((size_t)sp * rows_loc + lrow) * ldz + d];
  const int g = global_row(lrow, Bloc, Nglob, seg0);
  const int pg = g < Nglob ? g + Nglob : g - Nglob;
  dZ[e] = gscale * (acc - 2.f * Z[(size_t)pg * ldz + d]);
}

__global__ void __launch_bounds__(256)
ntxent_loss_k(int rows, const float* __restrict__ lse, const float* __restrict__ pos, float scale, float* __restrict__ loss) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += 256) s += (double)lse[i] - (double)pos[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *loss = (float)(sm[0] * (double)scale);
}

// ---- BYOL MSE pair -------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mse_pair_k(int64_t n, const float* __restrict__ o1, const float* __restrict__ o2, const float* __restrict__ t1, const float* __restrict__ t2,
           float inv_count, float* __restrict__ do1, float* __restrict__ do2, double* __restrict__ part) {
  __shared__ double sm[256];
  double s = 0.0;
  const float g = 2.f * inv_count;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = o1[i] - t2[i], b = o2[i] - t1[i];
    s += (double)(a * a) + (double)(b * b);
    do1[i] = g * a; do2[i] = g * b;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ void sum_partials_k(int nblk, const double* __restrict__ part, float scale, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[i];
    *out = (float)(s * (double)scale);
  }
}

// Barlow Twins: from the raw cross-correlation Craw = zi_hat^T zj_hat (D x D) to the loss and dL/dC.
//   C = Craw * inv_b ; L = sum_ij W_ij (C_ij - delta_ij)^2, W = 1 on the diagonal, lambda elsewhere;
//   G_ij = 2 W_ij (C_ij - delta_ij) * inv_b   (the 1/B of C = ./B is folded in, ready for the two GEMMs)
__global__ void __launch_bounds__(256)
barlow_cgrad_k(int D, const float* __restrict__ craw, float inv_b, float lambda, float* __restrict__ G, double* __restrict__ part) {
  __shared__ double sm[256];
  double s = 0.0;
  const int64_t n = (int64_t)D * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
    const bool diag = r == c;
    const float d = craw[i] * inv_b - (diag ? 1.f : 0.f);
    const float w = diag ? 1.f : lambda;
    s += (double)(w * d * d);
    G[i] = 2.f * w * d * inv_b;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

__global__ void scale_k(int64_t n, float* __restrict__ x, const float* __restrict__ factor) {
  const float f = *factor;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= f;
}

int reduce_blocks(int64_t n) {
  int64_t b = cdiv64(n, 1024);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int ssv_l2norm_fwd(int32_t rows, int32_t D, const float* z, int32_t normalize, float eps,
                              float* zhat, int32_t ldo, float* inv_norm, void* stream) {
  SSV_REQUIRE(rows > 0 && D > 0 && ldo >= D && z && zhat, "ssv_l2norm_fwd: bad arguments");
  SSV_REQUIRE(!normalize || inv_norm, "ssv_l2norm_fwd: inv_norm required when normalising");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(l2norm_fwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, D, z, normalize, eps, zhat, ldo, inv_norm);
  SSV_CHECK_LAUNCH("ssv_l2norm_fwd");
  return SSV_OK;
}

extern "C" int ssv_l2norm_bwd(int32_t rows, int32_t D, const float* zhat, int32_t ldz, const float* inv_norm,
                              const float* dzhat, int32_t ldd, int32_t normalize, float* dz, void* stream) {
  SSV_REQUIRE(rows > 0 && D > 0 && ldz >= D && ldd >= D && zhat && dzhat && dz, "ssv_l2norm_bwd: bad arguments");
  SSV_REQUIRE(!normalize || inv_norm, "ssv_l2norm_bwd: inv_norm required when normalising");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(l2norm_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, D, zhat, ldz, inv_norm, dzhat, ldd, normalize, dz);
  SSV_CHECK_LAUNCH("ssv_l2norm_bwd");
  return SSV_OK;
}

static int check_ntxent(int Nglob, int Bloc, int seg0, int ldz, const char* who) {
  SSV_REQUIRE(Nglob > 0 && Bloc > 0 && seg0 >= 0 && seg0 + Bloc <= Nglob, "%s: bad row partition (Nglob=%d Bloc=%d seg0=%d)", who, Nglob, Bloc, seg0);
  SSV_REQUIRE(ldz % 32 == 0 && ldz >= 32 && ldz <= 128, "%s: ldz must be 32, 64, 96 or 128 (got %d)", who, ldz);
  SSV_REQUIRE(Nglob < (1 << 29), "%s: Nglob too large", who);
  return SSV_OK;
}

// how many column splits the loss kernels take at this shape: enough workgroups for two per CU (row blocks x splits >= 512 where the
// columns allow), every split at least 8 column tiles (two per wavefront) so the sweep still amortises the row fragments it keeps in registers
extern "C" int64_t ssv_ntxent_default_splits(int32_t Nglob, int32_t Bloc) {
  if (Nglob <= 0 || Bloc <= 0) return 1;
  const int row_blocks = cdiv(2 * Bloc, 32), ntile = cdiv(2 * Nglob, 32);
  int splits = std::min(512 / std::max(row_blocks, 1), ntile / 8);
  return std::max(1, std::min(splits, 64));
}

extern "C" size_t ssv_ntxent_split_workspace_bytes(int32_t Bloc, int32_t ldz, int32_t splits) {
  if (Bloc <= 0 || ldz <= 0 || splits <= 1) return 0;
  return (size_t)splits * 2 * Bloc * (size_t)std::max(ldz, 3) * sizeof(float);
}

static int check_split(int Nglob, int Bloc, int ldz, int splits, const void* ws, size_t ws_bytes, const char* who) {
  SSV_REQUIRE(splits >= 1 && splits <= 64, "%s: splits must be 1..64 (got %d)", who, splits);
  SSV_REQUIRE(splits <= cdiv(2 * Nglob, 32), "%s: more splits (%d) than column tiles", who, splits);
  if (splits > 1) {
    SSV_REQUIRE(ws, "%s: workspace required with splits > 1", who);
    if (ws_bytes < ssv_ntxent_split_workspace_bytes(Bloc, ldz, splits)) SSV_FAIL(SSV_ERR_WORKSPACE, "%s: workspace too small", who);
  }
  return SSV_OK;
}

extern "C" int ssv_ntxent_fwd_split(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                                    float inv_temp, float* lse, float* pos, int32_t splits, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_ntxent(Nglob, Bloc, seg0, ldz, "ssv_ntxent_fwd")) return rc;
  if (int rc = check_split(Nglob, Bloc, ldz, splits, ws, ws_bytes, "ssv_ntxent_fwd_split")) return rc;
  SSV_REQUIRE(Z && lse && pos && ((uintptr_t)Z & 15) == 0, "ssv_ntxent_fwd: null or unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  const int tps = cdiv(cdiv(2 * Nglob, 32), splits);
  splits = cdiv(cdiv(2 * Nglob, 32), tps);                       // no empty trailing split
  const dim3 grid(cdiv(2 * Bloc, 32), splits);
  float* part = (float*)ws;
  switch (ldz / 8) {
    case 4:  hipLaunchKernelGGL((ntxent_fwd_k<4>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, inv_temp, lse, pos, tps, part); break;
    case 8:  hipLaunchKernelGGL((ntxent_fwd_k<8>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, inv_temp, lse, pos, tps, part); break;
    case 12: hipLaunchKernelGGL((ntxent_fwd_k<12>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, inv_temp, lse, pos, tps, part); break;
    default: hipLaunchKernelGGL((ntxent_fwd_k<16>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, inv_temp, lse, pos, tps, part); break;
  }
  if (splits > 1) hipLaunchKernelGGL(ntxent_merge_k, dim3(cdiv(2 * Bloc, 256)), dim3(256), 0, s, 2 * Bloc, splits, part, lse, pos);
  SSV_CHECK_LAUNCH("ssv_ntxent_fwd");
  return SSV_OK;
}

extern "C" int ssv_ntxent_fwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                              float inv_temp, float* lse, float* pos, void* stream) {
  return ssv_ntxent_fwd_split(Nglob, Bloc, seg0, ldz, Z, inv_temp, lse, pos, 1, nullptr, 0, stream);
}

extern "C" int ssv_ntxent_loss(int32_t rows, const float* lse, const float* pos, float scale, float* loss, void* stream) {
  SSV_REQUIRE(rows > 0 && lse && pos && loss, "ssv_ntxent_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(ntxent_loss_k, dim3(1), dim3(256), 0, s, rows, lse, pos, scale, loss);
  SSV_CHECK_LAUNCH("ssv_ntxent_loss");
  return SSV_OK;
}

extern "C" int ssv_ntxent_bwd_split(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                                    const float* lse_all, float inv_temp, float gscale, float* dZ,
                                    int32_t splits, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_ntxent(Nglob, Bloc, seg0, ldz, "ssv_ntxent_bwd")) return rc;
  if (int rc = check_split(Nglob, Bloc, ldz, splits, ws, ws_bytes, "ssv_ntxent_bwd_split")) return rc;
  SSV_REQUIRE(Z && lse_all && dZ && ((uintptr_t)Z & 15) == 0, "ssv_ntxent_bwd: null or unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  const int tps = cdiv(cdiv(2 * Nglob, 32), splits);
  splits = cdiv(cdiv(2 * Nglob, 32), tps);
  const dim3 grid(cdiv(2 * Bloc, 32), splits);
  float* part = (float*)ws;
  switch (ldz / 8) {
    case 4:  hipLaunchKernelGGL((ntxent_bwd_k<4>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, lse_all, inv_temp, gscale, dZ, tps, part); break;
    case 8:  hipLaunchKernelGGL((ntxent_bwd_k<8>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, lse_all, inv_temp, gscale, dZ, tps, part); break;
    case 12: hipLaunchKernelGGL((ntxent_bwd_k<12>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, lse_all, inv_temp, gscale, dZ, tps, part); break;
    default: hipLaunchKernelGGL((ntxent_bwd_k<16>), grid, dim3(256), 0, s, Nglob, Bloc, seg0, ldz, Z, lse_all, inv_temp, gscale, dZ, tps, part); break;
  }
  if (splits > 1)
    hipLaunchKernelGGL(ntxent_bwd_reduce_k, dim3(cdiv(2 * Bloc * ldz, 256)), dim3(256), 0, s, Nglob, Bloc, seg0, ldz, splits, Z, part, gscale, dZ);
  SSV_CHECK_LAUNCH("ssv_ntxent_bwd");
  return SSV_OK;
}

extern "C" int ssv_ntxent_bwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                              const float* lse_all, float inv_temp, float gscale, float* dZ, void* stream) {
  return ssv_ntxent_bwd_split(Nglob, Bloc, seg0, ldz, Z, lse_all, inv_temp, gscale, dZ, 1, nullptr, 0, stream);
}

// ---- NT-Xent for projection widths beyond the register-resident kernels (ldz > 128): the Gram block S = Z_loc Z_all^T comes from the
//      GEMM kernel (ssv_conv2d_fwd, written to memory), these two kernels do the row work, and dZ = W' Z is a GEMM again. ----------
namespace {
// one wavefront per local row: lse over the columns c != r, and the positive logit
__global__ void __launch_bounds__(256)
ntxent_gram_rows_k(int Nglob, int Bloc, int seg0, int lds, const float* __restrict__ S, float inv_temp, float* __restrict__ lse, float* __restrict__ pos) {
  const int lane = threadIdx.x & 63, lr = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (lr >= 2 * Bloc) return;
  const int R2 = 2 * Nglob, rg = global_row(lr, Bloc, Nglob, seg0), pg = rg < Nglob ? rg + Nglob : rg - Nglob;
  const float* row = S + (size_t)lr * lds;
  float m = -INFINITY;
  for (int c = lane; c < R2; c += 64) if (c != rg) m = fmaxf(m, row[c] * inv_temp);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < R2; c += 64) if (c != rg) sum += expf(row[c] * inv_temp - m);
  sum = wave_sum(sum);
  if (lane == 0) { lse[lr] = m + logf(sum); pos[lr] = row[pg] * inv_temp; }
}
// in place: S[lr][c] <- gscale * (e^{s - lse_r} + e^{s - lse_c} - 2 [c == pos(r)]) for c != r, 0 on the diagonal and on the pad columns
__global__ void __launch_bounds__(256)
ntxent_gram_weights_k(int Nglob, int Bloc, int seg0, int lds, float* __restrict__ S, const float* __restrict__ lse_all, float inv_temp, float gscale) {
  const int lr = blockIdx.y;
  const int R2 = 2 * Nglob, rg = global_row(lr, Bloc, Nglob, seg0), pg = rg < Nglob ? rg + Nglob : rg - Nglob;
  const float lse_r = lse_all[rg];
  float* row = S + (size_t)lr * lds;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < lds; c += gridDim.x * 256) {
    float w = 0.f;
    if (c < R2 && c != rg) {
      const float sv = row[c] * inv_temp;
      w = gscale * (expf(sv - lse_r) + expf(sv - lse_all[c]) - (c == pg ? 2.f : 0.f));
    }
    row[c] = w;
  }
}
}  // namespace

extern "C" int ssv_ntxent_gram_fwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t lds, const float* S, float inv_temp,
                                   float* lse, float* pos, void* stream) {
  SSV_REQUIRE(Nglob > 0 && Bloc > 0 && seg0 >= 0 && seg0 + Bloc <= Nglob && lds >= 2 * Nglob && S && lse && pos, "ssv_ntxent_gram_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(ntxent_gram_rows_k, dim3(cdiv(2 * Bloc, 4)), dim3(256), 0, s, Nglob, Bloc, seg0, lds, S, inv_temp, lse, pos);
  SSV_CHECK_LAUNCH("ssv_ntxent_gram_fwd");
  return SSV_OK;
}

extern "C" int ssv_ntxent_gram_weights(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t lds, float* S, const float* lse_all,
                                       float inv_temp, float gscale, void* stream) {
  SSV_REQUIRE(Nglob > 0 && Bloc > 0 && seg0 >= 0 && seg0 + Bloc <= Nglob && lds >= 2 * Nglob && S && lse_all && 2 * Bloc <= 65535, "ssv_ntxent_gram_weights: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  hipLaunchKernelGGL(ntxent_gram_weights_k, dim3(min(cdiv(lds, 256), 64), 2 * Bloc), dim3(256), 0, s, Nglob, Bloc, seg0, lds, S, lse_all, inv_temp, gscale);
  SSV_CHECK_LAUNCH("ssv_ntxent_gram_weights");
  return SSV_OK;
}

extern "C" size_t ssv_reduce_workspace_bytes(int64_t n) { return (size_t)reduce_blocks(n) * sizeof(double); }

extern "C" int ssv_mse_pair_fwd_bwd(int64_t n, const float* o1, const float* o2, const float* t1, const float* t2,
                                    float inv_count, float* loss, float* do1, float* do2,
                                    void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(n > 0 && o1 && o2 && t1 && t2 && loss && do1 && do2 && ws, "ssv_mse_pair_fwd_bwd: bad arguments");
  if (ws_bytes < ssv_reduce_workspace_bytes(n)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_mse_pair_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  const int nb = reduce_blocks(n);
  hipLaunchKernelGGL(mse_pair_k, dim3(nb), dim3(256), 0, s, n, o1, o2, t1, t2, inv_count, do1, do2, (double*)ws);
  hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(64), 0, s, nb, (const double*)ws, inv_count, loss);
  SSV_CHECK_LAUNCH("ssv_mse_pair_fwd_bwd");
  return SSV_OK;
}

extern "C" int ssv_scale(int64_t n, float* x, const float* factor_dev, void* stream) {
  SSV_REQUIRE(n > 0 && x && factor_dev, "ssv_scale: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  hipLaunchKernelGGL(scale_k, dim3(reduce_blocks(n)), dim3(256), 0, s, n, x, factor_dev);
  SSV_CHECK_LAUNCH("ssv_scale");
  return SSV_OK;
}

extern "C" int ssv_barlow_cgrad(int32_t D, const float* craw, float inv_b, float lambda, float* loss, float* G,
                                void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(D > 0 && craw && loss && G && ws, "ssv_barlow_cgrad: bad arguments");
  const int64_t n = (int64_t)D * D;
  if (ws_bytes < ssv_reduce_workspace_bytes(n)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_barlow_cgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  const int nb = reduce_blocks(n);
  hipLaunchKernelGGL(barlow_cgrad_k, dim3(nb), dim3(256), 0, s, D, craw, inv_b, lambda, G, (double*)ws);
  hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(64), 0, s, nb, (const double*)ws, 1.0f, loss);
  SSV_CHECK_LAUNCH("ssv_barlow_cgrad");
  return SSV_OK;
}
