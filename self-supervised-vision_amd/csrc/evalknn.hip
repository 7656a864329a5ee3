// k-nearest-neighbour label agreement on the GPU (reference utils/eval_utils.py:13-21, which goes through
// faiss.IndexFlatIP = exact inner-product search).  Ordering everywhere: similarity descending, ties by ascending index; the best hit is dropped (the reference drops
// column 0 blindly - normally the query itself) and the labels of the next k are compared with the query's.
//   * SSV_ARITH_BF16X3, d in {32, 64, 128}, k <= 20 (the reference's call: proj_dim 128, k = 20): the search is FUSED into the Gram product, S is never written
//     (namespace fused below, round 6).
//   * otherwise (rounds 3-5):  S = Z Z^T by row chunks on the implicit-GEMM kernel (a 1x1 "convolution" of the chunk with the whole feature matrix as the filter bank),
//     then one wavefront per query row streams its row of S keeping a sorted top-(k+1) per lane in registers; the 64 lists are merged with k+1 rounds of a wave arg-max.
#include "common.h"
#include "split_bf16.h"
#include <limits.h>

namespace {

template <int CAP>
__global__ void __launch_bounds__(256) knn_agree_k(const float* __restrict__ S, int64_t lds, int n, int rows, int row0,
                                                   const int32_t* __restrict__ labels, int k,
                                                   unsigned long long* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;                                   // whole wave leaves together
  const float* s = S + (int64_t)r * lds;
  float val[CAP];
  int idx[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i) { val[i] = -INFINITY; idx[i] = INT_MAX; }
  auto offer = [&](float v, int j) {
    if (v > val[CAP - 1]) {                                 // strict: an equal value with a larger index never displaces
      val[CAP - 1] = v; idx[CAP - 1] = j;
#pragma unroll
      for (int i = CAP - 1; i > 0; --i) {
        if (val[i] > val[i - 1]) {
          const float tv = val[i]; val[i] = val[i - 1]; val[i - 1] = tv;
          const int ti = idx[i]; idx[i] = idx[i - 1]; idx[i - 1] = ti;
        }
      }
    }
  };
  // each lane owns 4 consecutive columns per 256-column stripe (16-byte loads, two stripes in flight); within a lane the
  // columns arrive in ascending order, which is what the tie rule needs
  int j0 = 0;
  if ((lds & 3) == 0) {
    // round 6: FOUR stripes of 256 columns per trip (rounds 3-5: two) - a wave streams its 200 KB row with 4 KB in flight, and the "does anything beat my list"
    // test is one max over 16 values; measured at n = 50,000, d = 128: the selection pass 8.5 -> see bench.py's eval_knn leg
    const int nvec4 = n & ~1023;
    for (; j0 < nvec4; j0 += 1024) {
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = __builtin_nontemporal_load((const f32x4*)(s + j0 + 256 * q + lane * 4));
      const float thr = val[CAP - 1];
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = fmaxf(mx, fmaxf(fmaxf(v[q][0], v[q][1]), fmaxf(v[q][2], v[q][3])));
      if (mx > thr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) offer(v[q][e], j0 + 256 * q + lane * 4 + e);
      }
    }
    const int nvec = n & ~511;
    for (; j0 < nvec; j0 += 512) {
      const f32x4 a = *(const f32x4*)(s + j0 + lane * 4);
      const f32x4 b = *(const f32x4*)(s + j0 + 256 + lane * 4);
      const float thr = val[CAP - 1];
      if (fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))) > thr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(a[e], j0 + lane * 4 + e);
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(b[e], j0 + 256 + lane * 4 + e);
      }
    }
  }
  for (int j = j0 + lane; j < n; j += 64) offer(s[j], j);   // tail (and unaligned rows): one column per lane per pass
  const int own = labels[row0 + r];
  unsigned agree = 0;
  for (int t = 0; t <= k; ++t) {
    float bv = val[0];
    int bi = idx[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (bi == INT_MAX) break;                               // fewer than k+1 candidates
    if (idx[0] == bi) {                                     // the winning lane pops its head
#pragma unroll
      for (int i = 0; i < CAP - 1; ++i) { val[i] = val[i + 1]; idx[i] = idx[i + 1]; }
      val[CAP - 1] = -INFINITY; idx[CAP - 1] = INT_MAX;
    }
    if (t > 0) agree += (labels[bi] == own) ? 1u : 0u;
  }
  if (lane == 0 && agree) atomicAdd(count, (unsigned long long)agree);
}

__global__ void zero_count_k(unsigned long long* c) { *c = 0ull; }

// ---- round 6: the search fused into the Gram product (SSV_ARITH_BF16X3, d in {32, 64, 128}, k <= 20) ---------------------------------------------------------------
// S is never written.  A workgroup owns 128 queries (4 waves x 32) and streams a range of columns (= rows of Z) in tiles of 32 through LDS, split into the three bf16
// planes while they are staged (csrc/split_bf16.h; the queries are split once, into registers: the attention kernel's layout, csrc/vit.hip attn_fwd_sp_k).  Each tile is
// 6 x d/16 v_mfma_f32_32x32x16_bf16 per wave; the accumulator hands lane (query c, half h) the scores of columns (j & 3) + 8 (j >> 2) + 4 h.
//   * Each query's best 21 so far are a sorted list (value descending, index ascending: the reference's order) kept in LDS; its last entry is held in registers as the
//     threshold a score has to beat - ties never flood, and after the first tiles a tile rarely holds a candidate (~21 ln(n / 21) per query over the whole scan).
//   * A tile is screened by one max over the lane's 16 scores; a candidate is appended to a lane-private LDS queue (5 entries).  When a queue is full its wave raises a
//     flag and after the next tile's products ALL FOUR waves drain side by side (a drain by one wave alone would hold the other three at the tile barrier): the list
//     comes into registers, every lane inserts its query's queued entries - one branch-free sorted insertion per entry (v_med3 on the values), 64 lanes at a time - and
//     the list goes back.  Insertion is order-independent, so a run is reproducible bit for bit.  A lane that meets more candidates in one tile than its queue holds
//     (the first tiles of a scan) keeps them in a pending mask, drains at once and re-offers them.
// The vector instructions of the search are NOT hidden behind the other wave's matrix instructions (profiles/r02_probe_mfma_plus_valu.txt: the two share issue slots),
// so the design minimises their count: measured steps in profiles/r06_probe_knn_fused.txt.
// grid.y splits the COLUMNS when the row blocks alone would leave CUs empty (choose_parts); every part writes its 21 best per query ([part][slot][query]) and
// knn_finish_k - one LANE per query - merges the parts with the same insertion and counts the label matches of ranks 1..k.
namespace fused {
constexpr int KEEP = 21, DEPTH = 5;
template <int D> __device__ __forceinline__ int kp_off(int key, int chunk) {          // [32 keys][D] bf16 plane, 16-byte chunks XOR-swizzled over each 256-byte bank row
  constexpr int CH = D / 8, RPW = 16 / CH;
  return key * (2 * D) + ((chunk ^ ((key / RPW) & (CH - 1))) << 4);
}
__device__ __forceinline__ int colof(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8 (&pl)[3]) {
  u32x2 pa[3], pb[3];
  splitbf::split4(a, pa);
  splitbf::split4(b, pb);
#pragma unroll
  for (int q = 0; q < 3; ++q) pl[q] = __builtin_bit_cast(bf16x8, u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]});
}
// a precedes b in the result order (bitwise on purpose: three compares and two mask operations, no short-circuit branches)
__device__ __forceinline__ bool before(float av, int ai, float bv, int bi) { return (av > bv) | ((av == bv) & (ai < bi)); }
// (x, xi) into the sorted list; entries are distinct, empties are (-inf, INT_MAX) and an empty offer changes nothing.  Every new slot is a function of the OLD slots
// i-1 and i only (new v[i] = median(v[i-1], v[i], x)), so the 21 slots update independently: no compare-and-swap chain
__device__ __forceinline__ void insert(float (&lv)[KEEP], int (&li)[KEEP], float x, int xi) {
  bool c[KEEP];
#pragma unroll
  for (int i = 0; i < KEEP; ++i) c[i] = before(x, xi, lv[i], li[i]);
#pragma unroll
  for (int i = KEEP - 1; i >= 1; --i) {
    li[i] = c[i - 1] ? li[i - 1] : (c[i] ? xi : li[i]);
    lv[i] = __builtin_amdgcn_fmed3f(lv[i - 1], lv[i], x);
  }
  li[0] = c[0] ? xi : li[0];
  lv[0] = fmaxf(lv[0], x);
}

template <int D>
__global__ void __launch_bounds__(256, 2)
knn_fused_k(const float* __restrict__ Z, int n, int cols_per_part, float* __restrict__ part_v, int* __restrict__ part_i) {
  constexpr int PL = 32 * 2 * D;                               // bytes of one plane of a column tile
  __shared__ __attribute__((aligned(16))) unsigned char skp[2 * 3 * PL];
  __shared__ u32x2 cb[4][64][DEPTH];                           // lane-private candidate queues: (score bits, column)
  __shared__ int flag[4];
  __shared__ u32x2 ls[4][32][KEEP];                            // the queries' lists between drains (168-byte rows: 32 queries on 32 distinct bank pairs)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const int col_begin = blockIdx.y * cols_per_part, col_end = min(n, col_begin + cols_per_part);
  bf16x8 qf[D / 16][3];                                        // slab s: d = 16 s + 8 half + {0..7}
  {
    const float* qp = Z + (int64_t)min(q0 + c, n - 1) * D + 8 * half;
#pragma unroll
    for (int s = 0; s < D / 16; ++s) split8(*(const f32x4*)(qp + 16 * s), *(const f32x4*)(qp + 16 * s + 4), qf[s]);
  }
  u32x2 (*wq)[DEPTH] = cb[wave];
  u32x2* wl = ls[wave][c];
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) wl[i] = u32x2{__float_as_uint(-INFINITY), (unsigned)INT_MAX};
  }
  if (threadIdx.x < 4) flag[threadIdx.x] = 0;
  float thr_v = -INFINITY;                                     // entry 20 of the query's list at the last drain: what a score has to beat
  int thr_i = INT_MAX, cnt = 0;                                // cnt: entries in this lane's queue
  auto drain = [&]() {
    float lv[KEEP];
    int li[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) { const u32x2 e = wl[i]; lv[i] = __uint_as_float(e[0]); li[i] = (int)e[1]; }
    const auto sw = __builtin_amdgcn_permlane32_swap(cnt, cnt, false, false);          // [0]: the upper half receives the lower half's value; [1]: the lower the upper's
    const int pc = half ? (int)sw[0] : (int)sw[1];
    const int a = half ? pc : cnt, tot = cnt + pc;              // the query's entries: a from its half-0 lane, then the half-1 lane's - the same sequence in both lanes
    for (int t = 0; __ballot(t < tot); ++t) {
      float x = -INFINITY;
      int xi = INT_MAX;
      if (t < tot) {
        const bool second = t >= a;
        const u32x2 e = wq[second ? c + 32 : c][second ? t - a : t];
        x = __uint_as_float(e[0]);
        xi = (int)e[1];
      }
      insert(lv, li, x, xi);
    }
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < KEEP; ++i) wl[i] = u32x2{__float_as_uint(lv[i]), (unsigned)li[i]};
    }
    thr_v = lv[KEEP - 1];
    thr_i = li[KEEP - 1];
    cnt = 0;
  };
  constexpr int N4 = D / 32;                                   // float4 per thread per tile
  f32x4 rk[N4];
  auto load_k = [&](int col0) {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * 256 + threadIdx.x, r = e / (D / 4), c4 = e % (D / 4);
      rk[i] = *(const f32x4*)(Z + (int64_t)min(col0 + r, n - 1) * D + c4 * 4);
    }
  };
  auto store_k = [&](int stage) {
    unsigned char* kp = skp + stage * 3 * PL;
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int e = i * 256 + threadIdx.x, r = e / (D / 4), c4 = e % (D / 4);
      u32x2 pk[3];
      splitbf::split4(rk[i], pk);
      const int off = kp_off<D>(r, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) *(u32x2*)(kp + q * PL + off) = pk[q];
    }
  };
  load_k(col_begin);
  store_k(0);
  __syncthreads();
  int buf = 0;
  int tile = 0;
  for (int k0 = col_begin; k0 < col_end; k0 += 32, buf ^= 1, ++tile) {
    const bool more = k0 + 32 < col_end;
    if (more) load_k(k0 + 32);
    // flag[t % 3]: set during tile t-1 (before its barrier), read during tile t, cleared during tile t+1 (after every read), set again during tile t+2 at the earliest
    if (threadIdx.x == 0) flag[(tile + 2) % 3] = 0;
    const int want = flag[tile % 3];                            // read here, used after the products: the LDS latency hides behind them
    const unsigned char* kp = skp + buf * 3 * PL;
    f32x16 s;
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = 0.f;
    bf16x8 kf[2][3];                                            // the next slab's fragments are read while this slab's six products run
#pragma unroll
    for (int q = 0; q < 3; ++q) kf[0][q] = *(const bf16x8*)(kp + q * PL + kp_off<D>(c, half));
#pragma unroll
    for (int sl = 0; sl < D / 16; ++sl) {
      if (sl + 1 < D / 16) {
#pragma unroll
        for (int q = 0; q < 3; ++q) kf[(sl + 1) & 1][q] = *(const bf16x8*)(kp + q * PL + kp_off<D>(c, 2 * (sl + 1) + half));
      }
#define SSV_MM(P, Q_) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sl & 1][P], qf[sl][Q_], s, 0, 0, 0)
      SSV_MM(2, 0); SSV_MM(0, 2); SSV_MM(1, 1); SSV_MM(1, 0); SSV_MM(0, 1); SSV_MM(0, 0);          // smallest terms first
#undef SSV_MM
      __builtin_amdgcn_sched_barrier(0);
    }
    if (want && __ballot(cnt != 0)) drain();                    // some wave's queue filled up during the last tile: all four waves drain NOW, side by side - drains one
                                                                // wave at a time would each hold the other three at the barrier
    float m4[4];                                                // NaN scores drop out here and below (they never enter, as in the unfused search)
#pragma unroll
    for (int g = 0; g < 4; ++g) m4[g] = fmaxf(fmaxf(s[4 * g], s[4 * g + 1]), fmaxf(s[4 * g + 2], s[4 * g + 3]));
    const float mt = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    if (__ballot(mt >= thr_v)) {
      unsigned pend = 0;                                        // candidates that found the lane's queue full
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (__ballot(m4[g] >= thr_v)) {                         // wave-uniform: most groups of most tiles skip
#pragma unroll
          for (int j = 4 * g; j < 4 * g + 4; ++j) {
            const bool ge = s[j] >= thr_v;
            if (__ballot(ge)) {
              const int col = k0 + colof(j, half);
              const bool in = ge && col < col_end && before(s[j], col, thr_v, thr_i);
              const bool room = cnt < DEPTH;
              if (in && room) { wq[lane][cnt] = u32x2{__float_as_uint(s[j]), (unsigned)col}; ++cnt; }
              pend |= (in && !room) ? 1u << j : 0u;
            }
          }
        }
      }
      while (__ballot(pend != 0)) {                             // rare (the first tiles of a scan): drain, then re-offer up to DEPTH of the lane's pending scores
        drain();
#pragma unroll
        for (int r = 0; r < DEPTH; ++r) {
          const bool has = pend != 0;
          const int j = has ? __builtin_ctz(pend) : 0;
          pend &= pend - 1;
          float v = s[0];
#pragma unroll
          for (int jj = 1; jj < 16; ++jj) v = j == jj ? s[jj] : v;
          const int col = k0 + colof(j, half);
          if (has && before(v, col, thr_v, thr_i)) { wq[lane][cnt] = u32x2{__float_as_uint(v), (unsigned)col}; ++cnt; }
        }
      }
      if (__ballot(cnt == DEPTH) && lane == 0) flag[(tile + 1) % 3] = 1;       // ask for a drain after the next tile's products
    }
    if (more) store_k(buf ^ 1);
    __syncthreads();
  }
  drain();
  if (half == 0 && q0 + c < n) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int64_t o = ((int64_t)blockIdx.y * KEEP + i) * n + q0 + c;
      const u32x2 e = wl[i];
      part_v[o] = __uint_as_float(e[0]);
      part_i[o] = (int)e[1];
    }
  }
}

// one lane per query: the parts' sorted lists merged by the same insertion, then the label matches of ranks 1 .. k (rank 0 is dropped blindly, as the reference does)
__global__ void __launch_bounds__(256) knn_finish_k(int n, int parts, const float* __restrict__ part_v, const int* __restrict__ part_i, const int32_t* __restrict__ labels, int k,
                                                    unsigned long long* __restrict__ count) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  unsigned agree = 0;
  if (q < n) {
    float lv[KEEP];
    int li[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) { lv[i] = part_v[(int64_t)i * n + q]; li[i] = part_i[(int64_t)i * n + q]; }
    for (int e = KEEP; e < parts * KEEP; ++e) insert(lv, li, part_v[(int64_t)e * n + q], part_i[(int64_t)e * n + q]);
    const int own = labels[q];
#pragma unroll
    for (int r = 1; r < KEEP; ++r)
      if (r <= k && li[r] != INT_MAX && labels[li[r]] == own) ++agree;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) agree += __shfl_xor(agree, o, 64);
  if ((threadIdx.x & 63) == 0 && agree) atomicAdd(count, (unsigned long long)agree);
}

// column parts: as many as keep every workgroup resident at once (two per CU), at most 8.  Measured (tools/exp/r06_knn_parts.sh): n = 10,000 (79 row blocks) 1.22 ms
// with one part, 0.61 with six; n = 20,000 1.89 -> 1.23 with three; n = 50,000 (391 row blocks) 4.59 with one, 5.17 with five - a part starts from empty lists and
// its first ~50 tiles are the expensive ones, so a second ROUND of workgroups costs more than the idle quarter of the chip it would fill
static int choose_parts(int64_t n) {
  int cus = 256;
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  int64_t p = 2 * (int64_t)cus / cdiv64(n, 128);
  while (p > 1 && cdiv64(n, p) < 32 * 8) --p;                   // a part scans at least 8 tiles
  return (int)(p < 1 ? 1 : p > 8 ? 8 : p);
}
}  // namespace fused
using fused::knn_finish_k;
using fused::knn_fused_k;

// ---- linear probe: NLLLoss(log_softmax(logits)) + accuracy (utils/eval_utils.py:52-54), one wavefront per row ----------------
// stats[0] += sum of per-row losses, stats[1] += number of rows whose arg-max is the label; dlogits = (softmax - onehot) * gscale
__global__ void __launch_bounds__(256) softmax_ce_k(int N, int C, int ld, const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                    float gscale, float* __restrict__ dlogits, float* __restrict__ row_loss, int32_t* __restrict__ row_hit) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  const float* row = logits + (int64_t)r * ld;
  float mx = -INFINITY;
  int arg = 0;
  for (int c = lane; c < C; c += 64) { const float v = row[c]; if (v > mx) { mx = v; arg = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ov > mx || (ov == mx && oa < arg)) { mx = ov; arg = oa; }          // first maximum, like argmax
  }
  float sm = 0.f;
  for (int c = lane; c < C; c += 64) sm += expf(row[c] - mx);
  sm = wave_sum(sm);
  const float lse = mx + logf(sm);
  const int y = labels[r];
  if (dlogits) {
    for (int c = lane; c < ld; c += 64)
      dlogits[(int64_t)r * ld + c] = c < C ? (expf(row[c] - lse) - (c == y ? 1.f : 0.f)) * gscale : 0.f;
  }
  // a label outside [0, C) never reads past the row: its loss is NaN, which poisons the reported mean (the host wrapper refuses
  // such labels up front; NLLLoss in the reference raises)
  if (lane == 0) { row_loss[r] = (unsigned)y < (unsigned)C ? lse - row[y] : __builtin_nanf(""); row_hit[r] = arg == y ? 1 : 0; }
}
__global__ void ce_reduce_k(int N, const float* __restrict__ row_loss, const int32_t* __restrict__ row_hit, float* __restrict__ stats) {
  __shared__ double sl[256];
  __shared__ int sh[256];
  double a = 0.0;
  int h = 0;
  for (int i = threadIdx.x; i < N; i += 256) { a += (double)row_loss[i]; h += row_hit[i]; }
  sl[threadIdx.x] = a; sh[threadIdx.x] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sh[threadIdx.x] += sh[threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) { stats[0] = (float)(sl[0] / (double)N); stats[1] = (float)sh[0] / (float)N; }
}

int64_t chunk_rows(int64_t n) {
  // one chunk of S must satisfy the convolution's element limit (< 2^29 - 2^22 outputs) and stay cache/HBM friendly
  int64_t r = ((1ll << 29) - (1ll << 23)) / n;
  if (r > 4096) r = 4096;
  if (r > n) r = n;
  return r < 1 ? 1 : r;
}

}  // namespace

extern "C" size_t ssv_knn_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)(chunk_rows(n) * n) * sizeof(float) + 256;
}
// ... plus, for SSV_ARITH_BF16X3, the three bf16 planes of the feature matrix (it is the Gram product's filter bank): 6 n d bytes
extern "C" size_t ssv_knn_workspace_bytes_arith(int64_t n, int32_t d, int32_t arithmetic) {
  if (n <= 0 || d <= 0) return 0;
  const size_t unfused = ssv_knn_workspace_bytes(n) + (arithmetic == SSV_ARITH_BF16X3 ? (size_t)n * d * 6 + 256 : 0);
  const size_t parts = (size_t)n * 8 * fused::KEEP * 8 + 256;                                       // the fused search's per-part lists (<= 8 parts)
  return unfused > parts ? unfused : parts;
}

extern "C" int ssv_knn_label_agreement(int64_t n, int32_t d, const float* z, const int32_t* labels, int32_t k,
                                       unsigned long long* count, void* ws, size_t ws_bytes, void* stream) {
  return ssv_knn_label_agreement_arith(n, d, z, labels, k, count, SSV_ARITH_F32_MFMA, ws, ws_bytes, stream);
}

// The same search with the Gram product in the given arithmetic.  SSV_ARITH_BF16X3: every product of two features as six exact bf16 piece products accumulated in fp32
// (csrc/split_bf16.h) - integer-valued features up to 2^16 still give exact scores (their third pieces are zero, every kept term is exact and so is the sum below 2^24).
extern "C" int ssv_knn_label_agreement_arith(int64_t n, int32_t d, const float* z, const int32_t* labels, int32_t k,
                                             unsigned long long* count, int32_t arithmetic, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(n >= 2 && n < (1ll << 31) / 2 && d > 0 && d % 4 == 0, "ssv_knn_label_agreement: need 2 <= n < 2^30 and d %% 4 == 0 (got n=%lld d=%d)", (long long)n, d);
  SSV_REQUIRE(k >= 1 && k <= 63 && k < n, "ssv_knn_label_agreement: need 1 <= k <= min(63, n-1) (got %d)", k);
  SSV_REQUIRE(z && labels && count && ws, "ssv_knn_label_agreement: null pointer");
  SSV_REQUIRE(arithmetic == SSV_ARITH_F32_MFMA || arithmetic == SSV_ARITH_BF16X3, "ssv_knn_label_agreement: unknown arithmetic %d", arithmetic);
  if (arithmetic == SSV_ARITH_BF16X3 && (d % 32 != 0 || (n * d) % 8 != 0)) arithmetic = SSV_ARITH_F32_MFMA;      // no bf16-piece kernel for this width
  SSV_REQUIRE(ws_bytes >= ssv_knn_workspace_bytes_arith(n, d, arithmetic), "ssv_knn_label_agreement: workspace too small (%zu < %zu)", ws_bytes, ssv_knn_workspace_bytes_arith(n, d, arithmetic));
  SSV_REQUIRE((((uintptr_t)z | (uintptr_t)ws) & 15) == 0, "ssv_knn_label_agreement: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* S = (float*)ws;
  const int64_t cr = chunk_rows(n);
  if (arithmetic == SSV_ARITH_BF16X3 && (d == 32 || d == 64 || d == 128) && k < fused::KEEP) {          // the fused search: S never leaves the registers
    static int cached_parts_n = -1, cached_parts = 1;            // (one device kind per process)
    if (cached_parts_n != (int)n) { cached_parts = fused::choose_parts(n); cached_parts_n = (int)n; }
    int parts = cached_parts;
    int cpp = (int)(cdiv64(cdiv64(n, parts), 32) * 32);
    parts = (int)cdiv64(n, cpp);
    float* pv = (float*)ws;
    int* pi = (int*)(pv + (size_t)parts * n * fused::KEEP);
    ProfScope ps(SSV_PROF_MISC, s);
    hipLaunchKernelGGL(zero_count_k, dim3(1), dim3(1), 0, s, count);
    const dim3 grid((unsigned)cdiv64(n, 128), (unsigned)parts);
    if (d == 128) hipLaunchKernelGGL(knn_fused_k<128>, grid, dim3(256), 0, s, z, (int)n, cpp, pv, pi);
    else if (d == 64) hipLaunchKernelGGL(knn_fused_k<64>, grid, dim3(256), 0, s, z, (int)n, cpp, pv, pi);
    else hipLaunchKernelGGL(knn_fused_k<32>, grid, dim3(256), 0, s, z, (int)n, cpp, pv, pi);
    hipLaunchKernelGGL(knn_finish_k, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, (int)n, parts, (const float*)pv, (const int*)pi, labels, k, count);
    SSV_CHECK_LAUNCH("knn_fused_k");
    return SSV_OK;
  }
  void* planes = nullptr;
  if (arithmetic == SSV_ARITH_BF16X3) {
    planes = (char*)ws + ((ssv_knn_workspace_bytes(n) + 255) & ~(size_t)255);
    if (int rc = ssv_split_planes(n * d, z, planes, stream)) return rc;
  }
  {
    ProfScope ps(SSV_PROF_MISC, s);
    hipLaunchKernelGGL(zero_count_k, dim3(1), dim3(1), 0, s, count);
  }
  for (int64_t r0 = 0; r0 < n; r0 += cr) {
    const int rows = (int)((n - r0 < cr) ? (n - r0) : cr);
    ssv_conv_desc cd = {};
    cd.arithmetic = arithmetic; cd.w_planes = planes;
    cd.N = rows; cd.H = 1; cd.W = 1; cd.C = d; cd.K = (int32_t)n; cd.R = 1; cd.S = 1; cd.stride = 1; cd.pad = 0; cd.Ho = 1; cd.Wo = 1;
    if (int rc = ssv_conv2d_fwd(&cd, z + r0 * d, z, nullptr, nullptr, S, stream)) return rc;      // S[rows, n] = Z[r0:r0+rows] Z^T
    ProfScope ps(SSV_PROF_MISC, s);
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (k <= 20)
      hipLaunchKernelGGL(knn_agree_k<21>, grid, dim3(256), 0, s, S, n, (int)n, rows, (int)r0, labels, k, count);
    else
      hipLaunchKernelGGL(knn_agree_k<64>, grid, dim3(256), 0, s, S, n, (int)n, rows, (int)r0, labels, k, count);
    SSV_CHECK_LAUNCH("knn_agree_k");
  }
  return SSV_OK;
}

extern "C" size_t ssv_softmax_ce_workspace_bytes(int32_t N) { return N > 0 ? (size_t)N * 8 : 0; }

extern "C" int ssv_softmax_ce_fwd_bwd(int32_t N, int32_t C, int32_t ld, const float* logits, const int32_t* labels, float* stats,
                                      float* dlogits, void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(N > 0 && C > 0 && ld >= C && logits && labels && stats && ws, "ssv_softmax_ce_fwd_bwd: bad arguments");
  SSV_REQUIRE(ws_bytes >= ssv_softmax_ce_workspace_bytes(N), "ssv_softmax_ce_fwd_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_LOSS, s);
  float* row_loss = (float*)ws;
  int32_t* row_hit = (int32_t*)(row_loss + N);
  hipLaunchKernelGGL(softmax_ce_k, dim3(cdiv(N, 4)), dim3(256), 0, s, N, C, ld, logits, labels, 1.f / (float)N, dlogits, row_loss, row_hit);
  hipLaunchKernelGGL(ce_reduce_k, dim3(1), dim3(256), 0, s, N, (const float*)row_loss, (const int32_t*)row_hit, stats);
  SSV_CHECK_LAUNCH("ssv_softmax_ce_fwd_bwd");
  return SSV_OK;
}
