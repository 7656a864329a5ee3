// Two-view augmentation on the GPU: uint8 source images resident in HBM -> normalised fp32 NHWC views.
// The chain is the one configs/simclr.yaml:13-29 builds from torchvision transforms (reference
// utils/augmentations.py:113-144, run per sample in DataLoader workers, utils/data_utils.py:68-73):
//   RandomApply(ColorJitter, p) -> RandomGrayscale -> RandomResizedCrop(bilinear) -> HorizontalFlip -> ToTensor -> Normalize
// Integer/byte work, HBM- and L2-bound; no matrix cores.  The arithmetic mirrors Pillow's C kernels bit for bit
// (Blend.c float lerp + truncation, Convert.c rgb2l / rgb2hsv / hsv2rgb, Resample.c triangle filter with 22-bit
// fixed-point coefficients and a uint8 intermediate between the horizontal and the vertical pass); the CPU
// restatement it is tested against (oracle/augment.py) is itself pinned to Pillow.
//
// Three launches per batch:
//   aug_prep_k   : per (view, sample) - integer mean of the luma image in front of the contrast op (a global
//                  statistic of the source image) and the two resampling coefficient tables of the crop;
//   aug_views_k  : one thread per output pixel: <= KMAX x KMAX taps, the colour chain applied to each fetched source pixel;
//   (aug_params_k draws the per-sample parameters from a counter-based Philox4x32-10 stream.)
#include "common.h"

namespace {

constexpr int NPARAM = 16;
constexpr int KMAX = 8;                   // taps per axis: covers down-scaling factors up to 3.5
constexpr int PRECISION_BITS = 32 - 8 - 2;

struct AugCfg { double brightness, contrast, saturation, hue, p_jitter, p_gray, p_flip, scale_min, scale_max, ratio_min, ratio_max; };

// ---- Philox4x32-10 ------------------------------------------------------------------------------
struct Philox {
  uint32_t c[4], k[2], buf[4];
  int left;
  __device__ Philox(uint64_t seed, uint64_t step, uint64_t sample, uint32_t view) {
    k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
    c[0] = (uint32_t)sample; c[1] = ((uint32_t)(sample >> 32) & 0xFFFFu) | ((view & 0xFFFFu) << 16);
    c[2] = (uint32_t)step; c[3] = 0; left = 0;
  }
  __device__ void block() {
    uint32_t x0 = c[0], x1 = c[1], x2 = c[2], x3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t lo0 = 0xD2511F53u * x0, hi0 = __umulhi(0xD2511F53u, x0);
      const uint32_t lo1 = 0xCD9E8D57u * x2, hi1 = __umulhi(0xCD9E8D57u, x2);
      const uint32_t n0 = hi1 ^ x1 ^ k0, n1 = lo1, n2 = hi0 ^ x3 ^ k1, n3 = lo0;
      x0 = n0; x1 = n1; x2 = n2; x3 = n3;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    buf[0] = x0; buf[1] = x1; buf[2] = x2; buf[3] = x3;
    c[3] += 1; left = 4;
  }
  __device__ double uniform() {
    if (left == 0) block();
    const uint32_t v = buf[4 - left];
    --left;
    return (double)v * (1.0 / 4294967296.0);
  }
};

// RandomResizedCrop.get_params: 10 attempts of (area ~ U[scale], log-ratio ~ U[log ratio]), then the centre-crop fallback
__device__ void sample_rrc(Philox& st, int Hs, int Ws, double scale_min, double scale_max, double ratio_min, double ratio_max,
                           int& top, int& left, int& h, int& w) {
  const double area = (double)Hs * (double)Ws;
  top = 0; left = 0; h = Hs; w = Ws;
  bool done = false;
  const double lr0 = log(ratio_min), lr1 = log(ratio_max);
  for (int a = 0; a < 10 && !done; ++a) {
    const double target = area * (scale_min + st.uniform() * (scale_max - scale_min));
    const double ratio = exp(lr0 + st.uniform() * (lr1 - lr0));
    const int ww = (int)floor(sqrt(target * ratio) + 0.5), hh = (int)floor(sqrt(target / ratio) + 0.5);
    const double ui = st.uniform(), uj = st.uniform();
    if (0 < ww && ww <= Ws && 0 < hh && hh <= Hs) {
      w = ww; h = hh; top = (int)(ui * (double)(Hs - hh + 1)); left = (int)(uj * (double)(Ws - ww + 1));
      done = true;
    }
  }
  if (!done) {
    const double in_ratio = (double)Ws / (double)Hs;
    if (in_ratio < ratio_min) { w = Ws; h = (int)floor((double)Ws / ratio_min + 0.5); }
    else if (in_ratio > ratio_max) { h = Hs; w = (int)floor((double)Hs * ratio_max + 0.5); }
    else { w = Ws; h = Hs; }
    top = (Hs - h) / 2; left = (Ws - w) / 2;
  }
}

__global__ void aug_params_k(int B, int Hs, int Ws, AugCfg cfg, uint64_t seed, uint64_t step, const int64_t* __restrict__ ids,
                             int64_t sample0, int nviews, float* __restrict__ params) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nviews * B) return;
  const int view = t / B, b = t - view * B;
  const uint64_t sample = (uint64_t)(ids ? ids[b] : sample0 + b);
  Philox st(seed, step, sample, (uint32_t)view);
  float* p = params + (size_t)t * NPARAM;
  p[0] = st.uniform() < cfg.p_jitter ? 1.f : 0.f;
  int order[4] = {0, 1, 2, 3};
  for (int i = 3; i > 0; --i) {
    const int j = (int)(st.uniform() * (double)(i + 1));
    const int tmp = order[i]; order[i] = order[j]; order[j] = tmp;
  }
  p[1] = (float)order[0]; p[2] = (float)order[1]; p[3] = (float)order[2]; p[4] = (float)order[3];
  const double bl = fmax(0.0, 1.0 - cfg.brightness), cl = fmax(0.0, 1.0 - cfg.contrast), sl = fmax(0.0, 1.0 - cfg.saturation);
  p[5] = (float)(bl + st.uniform() * (1.0 + cfg.brightness - bl));
  p[6] = (float)(cl + st.uniform() * (1.0 + cfg.contrast - cl));
  p[7] = (float)(sl + st.uniform() * (1.0 + cfg.saturation - sl));
  p[8] = (float)(-cfg.hue + st.uniform() * 2.0 * cfg.hue);
  p[9] = st.uniform() < cfg.p_gray ? 1.f : 0.f;
  int top, left, h, w;
  sample_rrc(st, Hs, Ws, cfg.scale_min, cfg.scale_max, cfg.ratio_min, cfg.ratio_max, top, left, h, w);
  p[10] = (float)top; p[11] = (float)left; p[12] = (float)h; p[13] = (float)w;
  p[14] = st.uniform() < cfg.p_flip ? 1.f : 0.f;
  p[15] = 0.f;
}

// ---- Pillow pixel arithmetic ---------------------------------------------------------------------
struct Px { int r, g, b; };

__device__ __forceinline__ int luma(Px v) { return (v.r * 19595 + v.g * 38470 + v.b * 7471 + 0x8000) >> 16; }

// Image.blend(degenerate d, image i, factor a): separate float multiply and add (no FMA), truncation
__device__ __forceinline__ int blend1(int d, int i, float a, bool inside) {
  const float t = __fadd_rn((float)d, __fmul_rn(a, (float)(i - d)));
  if (inside) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ Px blend(Px d, Px i, float a) {
  const bool inside = a >= 0.f && a <= 1.f;
  return Px{blend1(d.r, i.r, a, inside), blend1(d.g, i.g, a, inside), blend1(d.b, i.b, a, inside)};
}

__device__ __forceinline__ int c_round(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5)); }
__device__ __forceinline__ int clip8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ Px hue_shift(Px v, int shift) {
  const int maxc = max(v.r, max(v.g, v.b)), minc = min(v.r, min(v.g, v.b));
  int uh = 0, us = 0;
  if (minc != maxc) {                                   // Convert.c rgb2hsv_row
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - v.r), cr), gc = __fdiv_rn((float)(maxc - v.g), cr), bc = __fdiv_rn((float)(maxc - v.b), cr);
    float h;
    if (v.r == maxc) h = __fsub_rn(bc, gc);
    else if (v.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    const double hd = (double)h / 6.0 + 1.0;
    h = (float)(hd - floor(hd));                       // fmod(x, 1.0) for x in (0, 3)
    uh = clip8i((int)((double)h * 255.0));
    us = clip8i((int)((double)s * 255.0));
  }
  uh = (uh + shift) & 255;
  if (us == 0) return Px{maxc, maxc, maxc};             // Convert.c hsv2rgb
  const double h6 = (double)(float)uh * 6.0 / 255.0;
  const float fi = (float)floor(h6);
  const float f = (float)(h6 - (double)fi);
  const float fs = (float)((double)(float)us / 255.0);
  const double vv = (double)maxc;
  const int p = clip8i(c_round(vv * (1.0 - (double)fs)));
  const int q = clip8i(c_round(vv * (1.0 - (double)fs * (double)f)));
  const int t = clip8i(c_round(vv * (1.0 - (double)fs * (1.0 - (double)f))));
  switch (((int)fi) % 6) {
    case 0: return Px{maxc, t, p};
    case 1: return Px{q, maxc, p};
    case 2: return Px{p, maxc, t};
    case 3: return Px{p, q, maxc};
    case 4: return Px{t, p, maxc};
    default: return Px{maxc, p, q};
  }
}

// ops of the jitter chain in drawn order; `upto_contrast` stops in front of the contrast op (for its mean)
__device__ Px color_chain(Px v, const float* __restrict__ p, int cmean, bool upto_contrast) {
  if (p[0] >= 0.5f) {
#pragma unroll 1
    for (int k = 1; k <= 4; ++k) {
      const int op = (int)p[k];
      if (op == 0) v = blend(Px{0, 0, 0}, v, p[5]);
      else if (op == 1) { if (upto_contrast) return v; v = blend(Px{cmean, cmean, cmean}, v, p[6]); }
      else if (op == 2) { const int l = luma(v); v = blend(Px{l, l, l}, v, p[7]); }
      else v = hue_shift(v, ((int)((double)p[8] * 255.0)) & 255);     // np.uint8(hue_factor * 255): wrap-around
    }
  }
  if (!upto_contrast && p[9] >= 0.5f) { const int l = luma(v); v = Px{l, l, l}; }
  return v;
}

__device__ __forceinline__ Px load_px(const uint8_t* __restrict__ img, int Ws, int y, int x) {
  const uint8_t* q = img + ((size_t)y * Ws + x) * 3;
  return Px{q[0], q[1], q[2]};
}

// Resample.c precompute_coeffs (triangle filter) + normalize_coeffs_8bpc for one output index
__device__ void coeffs_for(int in_size, int out_size, int xx, int* __restrict__ rec /* [2 + KMAX] */) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filterscale, ss = 1.0 / filterscale;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > KMAX) xmax = KMAX;
  double k[KMAX];
  double ww = 0.0;
  for (int x = 0; x < KMAX; ++x) {
    double w = 0.0;
    if (x < xmax) {
      const double a = fabs(((double)(x + xmin) - center + 0.5) * ss);
      w = a < 1.0 ? 1.0 - a : 0.0;
    }
    k[x] = w; ww += w;
  }
  rec[0] = xmin; rec[1] = xmax;
  for (int x = 0; x < KMAX; ++x) {
    double v = k[x];
    if (x < xmax && ww != 0.0) v = v / ww;
    rec[2 + x] = x < xmax ? (int)(v * (double)(1 << PRECISION_BITS) + 0.5) : 0;     // coefficients are >= 0 for this filter
  }
}

// one workgroup per (view, sample): contrast mean + coefficient tables
__global__ void __launch_bounds__(256)
aug_prep_k(int B, int Hs, int Ws, int Ho, int Wo, const uint8_t* __restrict__ src, const int64_t* __restrict__ ids,
           const float* __restrict__ params, int* __restrict__ cmean, int* __restrict__ tabs) {
  __shared__ unsigned long long red[256];
  const int vb = blockIdx.x;                               // view * B + b
  const int b = vb % B;
  const float* p = params + (size_t)vb * NPARAM;
  const int top = (int)p[10], ch = (int)p[12], cw = (int)p[13];
  (void)top;
  int* tab = tabs + (size_t)vb * (Ho + Wo) * (2 + KMAX);
  for (int i = threadIdx.x; i < Ho + Wo; i += 256) {
    if (i < Wo) coeffs_for(cw, Wo, i, tab + (size_t)i * (2 + KMAX));
    else coeffs_for(ch, Ho, i - Wo, tab + (size_t)i * (2 + KMAX));
  }
  bool need = false;
  if (p[0] >= 0.5f) need = true;
  unsigned long long s = 0;
  if (need) {
    const uint8_t* img = src + (size_t)(ids ? ids[b] : b) * Hs * Ws * 3;
    for (int i = threadIdx.x; i < Hs * Ws; i += 256) {
      const uint8_t* q = img + (size_t)i * 3;
      s += (unsigned long long)luma(color_chain(Px{q[0], q[1], q[2]}, p, 0, true));
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) cmean[vb] = need ? (int)((double)red[0] / (double)(Hs * Ws) + 0.5) : 0;
}

__global__ void __launch_bounds__(256)
aug_views_k(int B, int Hs, int Ws, int Ho, int Wo, int nviews, const uint8_t* __restrict__ src, const int64_t* __restrict__ ids,
            const float* __restrict__ params, const int* __restrict__ cmean, const int* __restrict__ tabs,
            float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const int64_t total = (int64_t)nviews * B * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    int64_t t = i / Wo;
    const int yo = (int)(t % Ho);
    const int vb = (int)(t / Ho);
    const int b = vb % B;
    const float* p = params + (size_t)vb * NPARAM;
    const int top = (int)p[10], left = (int)p[11];
    const int xs = p[14] >= 0.5f ? Wo - 1 - xo : xo;
    const int* tab = tabs + (size_t)vb * (Ho + Wo) * (2 + KMAX);
    const int* hx = tab + (size_t)xs * (2 + KMAX);
    const int* vy = tab + (size_t)(Wo + yo) * (2 + KMAX);
    const uint8_t* img = src + (size_t)(ids ? ids[b] : b) * Hs * Ws * 3;
    const int cm = cmean[vb];
    const int xmin = hx[0], xn = hx[1], ymin = vy[0], yn = vy[1];
    int av0 = 1 << (PRECISION_BITS - 1), av1 = av0, av2 = av0;
    for (int ty = 0; ty < yn; ++ty) {
      int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int tx = 0; tx < xn; ++tx) {
        const Px v = color_chain(load_px(img, Ws, top + ymin + ty, left + xmin + tx), p, cm, false);
        const int kx = hx[2 + tx];
        a0 += v.r * kx; a1 += v.g * kx; a2 += v.b * kx;
      }
      const int ky = vy[2 + ty];
      av0 += clip8i(a0 >> PRECISION_BITS) * ky; av1 += clip8i(a1 >> PRECISION_BITS) * ky; av2 += clip8i(a2 >> PRECISION_BITS) * ky;
    }
    float* o = out + (size_t)i * 3;          // ToTensor (/255) then Normalize, IEEE float32 ops
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8i(av0 >> PRECISION_BITS), 255.f), m0), s0);
    o[1] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8i(av1 >> PRECISION_BITS), 255.f), m1), s1);
    o[2] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8i(av2 >> PRECISION_BITS), 255.f), m2), s2);
  }
}

// CenterCrop -> ToTensor -> Normalize (the "img" entry of the batch)
__global__ void __launch_bounds__(256)
center_view_k(int B, int Hs, int Ws, int Ho, int Wo, int top, int left, const uint8_t* __restrict__ src, const int64_t* __restrict__ ids,
              float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const int64_t total = (int64_t)B * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    int64_t t = i / Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const uint8_t* q = src + ((size_t)(ids ? ids[b] : b) * Hs * Ws + (size_t)(top + yo) * Ws + left + xo) * 3;
    float* o = out + (size_t)i * 3;
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)q[0], 255.f), m0), s0);
    o[1] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)q[1], 255.f), m1), s1);
    o[2] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)q[2], 255.f), m2), s2);
  }
}


// ---- MultiCrop (utils/augmentations.py:156-173): RandomResizedCrop(BICUBIC) of an ALREADY normalised float view ----------
// boxes[b][crop] = (top, left, h, w); stream keyed (seed, step, sample, view_base + crop) so it never collides with the
// two-view parameter streams (views 0..15)
__global__ void multicrop_params_k(int B, int Hs, int Ws, int ncrop, int view_base, double scale_min, double scale_max,
                                   double ratio_min, double ratio_max, uint64_t seed, uint64_t step, const int64_t* __restrict__ ids,
                                   int64_t sample0, int32_t* __restrict__ boxes) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * ncrop) return;
  const int b = t / ncrop, crop = t - b * ncrop;
  Philox st(seed, step, (uint64_t)(ids ? ids[b] : sample0 + b), (uint32_t)(view_base + crop));
  int top, left, h, w;
  sample_rrc(st, Hs, Ws, scale_min, scale_max, ratio_min, ratio_max, top, left, h, w);
  boxes[4 * t + 0] = top; boxes[4 * t + 1] = left; boxes[4 * t + 2] = h; boxes[4 * t + 3] = w;
}

// cubic convolution weights, A = -0.75 (upsample_bicubic2d)
__device__ __forceinline__ void cubic_w(float t, float* w) {
  const float A = -0.75f;
  float x = t + 1.f;  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;              w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;        w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;        w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// views [B][Hs][Ws][3] (NHWC) -> out [B][ncrop][Ho][Wo][3]; F.interpolate(mode="bicubic", align_corners=False) of the box,
// source index = scale * (dst + 0.5) - 0.5 (not clamped), taps clamped to the box, no antialias, no value clamp
__global__ void __launch_bounds__(256) multicrop_k(int64_t total, int Hs, int Ws, int ncrop, int Ho, int Wo, const float* __restrict__ views,
                                                   const int32_t* __restrict__ boxes, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % Wo);
  int64_t t = i / Wo;
  const int yo = (int)(t % Ho);
  const int64_t bc = t / Ho;                                  // b * ncrop + crop
  const int64_t b = bc / ncrop;
  const int32_t* bx = boxes + 4 * bc;
  const int top = bx[0], left = bx[1], h = bx[2], w = bx[3];
  const float sy = (float)h / (float)Ho, sx = (float)w / (float)Wo;
  const float ry = sy * ((float)yo + 0.5f) - 0.5f, rx = sx * ((float)xo + 0.5f) - 0.5f;
  const float fy = floorf(ry), fx = floorf(rx);
  float wy[4], wx[4];
  cubic_w(ry - fy, wy);
  cubic_w(rx - fx, wx);
  const int iy = (int)fy, ix = (int)fx;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = top + min(max(iy - 1 + a, 0), h - 1);
    float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int xx = left + min(max(ix - 1 + c, 0), w - 1);
      const float* px = views + ((b * Hs + yy) * (int64_t)Ws + xx) * 3;
      row[0] += px[0] * wx[c]; row[1] += px[1] * wx[c]; row[2] += px[2] * wx[c];
    }
    acc[0] += row[0] * wy[a]; acc[1] += row[1] * wy[a]; acc[2] += row[2] * wy[a];
  }
  float* o = out + i * 3;
  o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
}

unsigned grid_for(int64_t n) {
  int64_t b = cdiv64(n, 256);
  if (b > 8192) b = 8192;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int ssv_augment_params(int32_t B, int32_t Hs, int32_t Ws, int32_t nviews, const ssv_aug_cfg* cfg, uint64_t seed, uint64_t step,
                                  const int64_t* sample_ids, int64_t sample0, float* params, void* stream) {
  SSV_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && nviews > 0 && nviews <= 16 && cfg && params, "ssv_augment_params: bad arguments");
  SSV_REQUIRE(cfg->scale_min > 0 && cfg->scale_max >= cfg->scale_min && cfg->ratio_min > 0 && cfg->ratio_max >= cfg->ratio_min, "ssv_augment_params: bad crop ranges");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_AUG, s);
  AugCfg c{cfg->brightness, cfg->contrast, cfg->saturation, cfg->hue, cfg->p_jitter, cfg->p_gray, cfg->p_flip,
           cfg->scale_min, cfg->scale_max, cfg->ratio_min, cfg->ratio_max};
  hipLaunchKernelGGL(aug_params_k, dim3(cdiv(nviews * B, 64)), dim3(64), 0, s, B, Hs, Ws, c, seed, step, sample_ids, sample0, nviews, params);
  SSV_CHECK_LAUNCH("ssv_augment_params");
  return SSV_OK;
}

extern "C" size_t ssv_augment_workspace_bytes(int32_t B, int32_t nviews, int32_t Ho, int32_t Wo) {
  if (B <= 0 || nviews <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return ((size_t)nviews * B * (1 + (size_t)(Ho + Wo) * (2 + KMAX))) * sizeof(int);
}

extern "C" int ssv_augment_views(int32_t B, int32_t nviews, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                                 const uint8_t* src, const int64_t* sample_ids, const float* params,
                                 const float* mean3_host, const float* std3_host, float* out,
                                 void* ws, size_t ws_bytes, void* stream) {
  SSV_REQUIRE(B > 0 && nviews > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && src && params && mean3_host && std3_host && out && ws,
              "ssv_augment_views: bad arguments");
  SSV_REQUIRE((int64_t)Hs * Ws < (1 << 24), "ssv_augment_views: source image too large");
  SSV_REQUIRE(2 * (Hs + Ho - 1) / Ho + 1 <= KMAX && 2 * (Ws + Wo - 1) / Wo + 1 <= KMAX,
              "ssv_augment_views: down-scaling factor above 3.5 needs more than %d taps", KMAX);
  if (ws_bytes < ssv_augment_workspace_bytes(B, nviews, Ho, Wo)) SSV_FAIL(SSV_ERR_WORKSPACE, "ssv_augment_views: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_AUG, s);
  int* cmean = (int*)ws;
  int* tabs = cmean + (size_t)nviews * B;
  hipLaunchKernelGGL(aug_prep_k, dim3(nviews * B), dim3(256), 0, s, B, Hs, Ws, Ho, Wo, src, sample_ids, params, cmean, tabs);
  hipLaunchKernelGGL(aug_views_k, dim3(grid_for((int64_t)nviews * B * Ho * Wo)), dim3(256), 0, s, B, Hs, Ws, Ho, Wo, nviews, src, sample_ids,
                     params, (const int*)cmean, (const int*)tabs, mean3_host[0], mean3_host[1], mean3_host[2],
                     std3_host[0], std3_host[1], std3_host[2], out);
  SSV_CHECK_LAUNCH("ssv_augment_views");
  return SSV_OK;
}

extern "C" int ssv_center_view(int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo, const uint8_t* src, const int64_t* sample_ids,
                               const float* mean3_host, const float* std3_host, float* out, void* stream) {
  SSV_REQUIRE(B > 0 && Hs >= Ho && Ws >= Wo && Ho > 0 && Wo > 0 && src && mean3_host && std3_host && out, "ssv_center_view: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_AUG, s);
  // torchvision center_crop: int(round((H - h) / 2.)) with Python's round-half-to-even
  auto py_round_half = [](int d) { return (d % 2 == 0) ? d / 2 : ((d / 2) % 2 == 0 ? d / 2 : d / 2 + 1); };
  const int top = py_round_half(Hs - Ho), left = py_round_half(Ws - Wo);
  hipLaunchKernelGGL(center_view_k, dim3(grid_for((int64_t)B * Ho * Wo)), dim3(256), 0, s, B, Hs, Ws, Ho, Wo, top, left, src, sample_ids,
                     mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], out);
  SSV_CHECK_LAUNCH("ssv_center_view");
  return SSV_OK;
}

extern "C" int ssv_multicrop_params(int32_t B, int32_t Hs, int32_t Ws, int32_t ncrop, int32_t view_base, double scale_min, double scale_max,
                                    uint64_t seed, uint64_t step, const int64_t* sample_ids, int64_t sample0, int32_t* boxes, void* stream) {
  SSV_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && ncrop > 0 && view_base >= 16 && view_base + ncrop <= 65536 && boxes, "ssv_multicrop_params: bad arguments (view_base >= 16)");
  SSV_REQUIRE(scale_min > 0 && scale_max >= scale_min, "ssv_multicrop_params: bad scale range");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_AUG, s);
  hipLaunchKernelGGL(multicrop_params_k, dim3(cdiv(B * ncrop, 64)), dim3(64), 0, s, B, Hs, Ws, ncrop, view_base, scale_min, scale_max,
                     3.0 / 4.0, 4.0 / 3.0, seed, step, sample_ids, sample0, boxes);
  SSV_CHECK_LAUNCH("ssv_multicrop_params");
  return SSV_OK;
}

extern "C" int ssv_multicrop(int32_t B, int32_t Hs, int32_t Ws, const float* views_nhwc, int32_t ncrop, const int32_t* boxes,
                             int32_t Ho, int32_t Wo, float* out_nhwc, void* stream) {
  SSV_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && ncrop > 0 && Ho > 0 && Wo > 0 && views_nhwc && boxes && out_nhwc, "ssv_multicrop: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_AUG, s);
  const int64_t total = (int64_t)B * ncrop * Ho * Wo;
  hipLaunchKernelGGL(multicrop_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, total, Hs, Ws, ncrop, Ho, Wo, views_nhwc, boxes, out_nhwc);
  SSV_CHECK_LAUNCH("ssv_multicrop");
  return SSV_OK;
}
