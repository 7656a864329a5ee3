/*
 * ssv_hip.h - C ABI of libssv_hip.so: the MI355X (gfx950) hot path of the two-view
 * self-supervised training step.
 *
 * The reference (NightShade99/Self-Supervised-Vision) is pure Python/PyTorch and has NO
 * FFI of its own: each entry point below replaces the implicit ATen / cuDNN
 * call that a reference line issues, cited as "replaces <file:line>".  INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted;
 *   - activations are NHWC fp32 ([N*H*W][C], C contiguous), filters are OHWI fp32
 *     (= a torch [O,I,H,W] tensor in channels_last memory format);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are only
 *     enqueued, nothing synchronises, nothing is allocated, no pointer is kept after return;
 *   - the caller owns every buffer including `ws` (size from the matching *_workspace_bytes);
 *   - return 0 on success, a negative ssv_status otherwise; ssv_last_error() gives the
 *     thread-local message.  Nothing throws across the ABI.
 */
#ifndef SSV_HIP_H
#define SSV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ssv_status {
  SSV_OK = 0,
  SSV_ERR_INVALID = -1,     /* bad shape / null pointer / unsupported configuration */
  SSV_ERR_WORKSPACE = -2,   /* workspace too small */
  SSV_ERR_LAUNCH = -3       /* hipLaunch / hipGetLastError failure */
} ssv_status;

int ssv_version(void);
const char* ssv_last_error(void);
/* Build identity: the first 16 hex digits of the sha256 over the sources this binary was compiled from (csrc Makefile: the .hip files in
 * link order, common.h, this header).  hipcc output is not bit-reproducible, so a hash of the .so file cannot tell "same kernels, rebuilt"
 * from "other kernels"; this one can.  Profiles record it and bench.py refuses counters measured on another build. */
const char* ssv_source_sha16(void);
/* number of CUs of the current device (0 if no device) - host-side helper for grid sizing */
int ssv_device_cus(void);

/* ---- convolution as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) -----------------
 * replaces nn.Conv2d forward/backward issued at networks/resnet.py:39-40,68-70,147 and,
 * with H=W=R=S=1, nn.Linear at models/simclr.py:34-35, models/byol.py:34, models/barlow.py:31-34 */
typedef struct ssv_conv_desc {
  int32_t N, H, W, C;      /* input  [N,H,W,C]            */
  int32_t K, R, S;         /* filter [K,R,S,C]            */
  int32_t stride, pad;     /* same in both spatial dims   */
  int32_t Ho, Wo;          /* output [N,Ho,Wo,K]          */
  /* ABI 120: HOW the launch multiplies.  SSV_ARITH_F32_MFMA (0): v_mfma_f32_32x32x2_f32 on the fp32 operands.  SSV_ARITH_BF16X3 (6): every fp32 operand as
   * three bf16 pieces, six of the nine piece products (each exact) accumulated in fp32 by v_mfma_f32_16x16x32_bf16 - fp32 inputs, fp32 outputs, error against
   * fp64 at or below the fp32-MFMA kernel's (csrc/split_bf16.h states the arithmetic and its edge cases; tests/test_gpu_split.py measures it on every layer
   * shape of the networks).  It is a REQUEST: launches whose shape has no such kernel (C % 32 != 0 forward products - the image stem -, C % 4 != 0 weight
   * gradients, the strided data-gradient kernel) run on fp32 MFMA; ssv_conv_arithmetic() tells which one a launch takes. */
  int32_t arithmetic;
  int32_t reserved;        /* 0 */
  /* SSV_ARITH_BF16X3, launches with a WEIGHT operand (forward, data gradient): that operand pre-split by ssv_split_planes - [3][K][R*S*C] bf16, the `w` the
   * launch is given, piece by piece - made once per weight and step instead of once per tile.  NULL: the launch runs on fp32 MFMA.  Weight gradients (both
   * operands are activations, split while they are staged) ignore it. */
  const void* w_planes;
} ssv_conv_desc;
enum { SSV_ARITH_F32_MFMA = 0, SSV_ARITH_BF16X3 = 6 };
/* product: 0 forward-kernel launches (ssv_conv2d_fwd* / ssv_linear_*), 1 ssv_conv2d_dgrad*, 2 ssv_conv2d_wgrad*.  Returns the arithmetic the launch described by
 * d will run on (SSV_ARITH_*), or SSV_ERR_INVALID */
int ssv_conv_arithmetic(const ssv_conv_desc* d, int32_t product);
/* planes[q][i], q < 3: the three bf16 pieces of x[i] (round to nearest even, residuals exact): x[i] == p0 + p1 + p2 for |x[i]| in [2^-109, 3.38e38].  n % 8 == 0,
 * 16-byte aligned pointers; planes holds 3 n bf16 (6 n bytes). */
int ssv_split_planes(int64_t n, const float* x, void* planes, void* stream);

/* y = conv(x, w) (+ bias[k]) (+ addend)            bias/addend may be NULL */
int ssv_conv2d_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias,
                   const float* addend, float* y, void* stream);
/* [npix][cin] -> [npix][cout]: zero-pad the channel axis (cout > cin) or drop / accumulate back its first cout channels (cout < cin).
 * The image stem runs with its 3 input channels padded to 4: ssv_conv2d_fwd with C == 4 gathers one filter tap per 16-byte load
 * (nn.Conv2d(3, 64, 7, 2, 3) / (3, 64, 3, 1, 1), networks/resnet.py:96-99), and ssv_conv2d_wgrad takes its float4 path. */
int ssv_pad_channels(int64_t npix, int32_t cin, int32_t cout, const float* in, float* out, int32_t accumulate, void* stream);
/* Grouped convolution (conv3x3(groups=32) of the ResNeXt encoders, networks/resnet.py:8-10,57): the grouped filter bank
 * [K][R][S][C/groups] is expanded to the dense block-diagonal [K][R][S][C] one (zeros elsewhere - exact, they contribute 0) and
 * run through the MFMA kernels (the group-aware entry points below); the bank-layout weight gradient is gathered back (+= when accumulate). */
int ssv_group_expand(int32_t K, int32_t R, int32_t S, int32_t Cg, int32_t groups, const float* wg, float* wd, void* stream);
int ssv_group_extract(int32_t K, int32_t R, int32_t S, int32_t Cg, int32_t groups, const float* dwd, float* dwg, int32_t accumulate, void* stream);

/* Grouped convolutions on the DENSE block-diagonal bank wd [K][R][S][C] made by ssv_group_expand (networks/resnet.py:8-10,57: conv3x3(groups = 32) of
 * the ResNeXt bottlenecks): the kernels of ssv_conv2d_fwd / _dgrad / _wgrad, but every output-column tile contracts only over the channels of the
 * groups it falls into - exact (the skipped products are against zero weights) and C/64 (forward), K/64 (data gradient) times fewer k-tiles.
 * groups must divide C and K.  The stride-1 data gradient is ssv_conv2d_fwd_grouped on the transposed bank (ssv_filter_transpose), like the dense one.
 * ssv_conv2d_wgrad_grouped leaves the weight gradient in the bank's layout with ONLY the diagonal blocks defined (what ssv_group_extract reads). */
int ssv_conv2d_fwd_grouped(const ssv_conv_desc* d, int32_t groups, const float* x, const float* wd, const float* bias, const float* addend,
                           float* y, void* stream);
int ssv_conv2d_dgrad_grouped(const ssv_conv_desc* d, int32_t groups, const float* dy, const float* wd, const float* addend, float* dx, void* stream);
size_t ssv_conv2d_wgrad_grouped_workspace_bytes(const ssv_conv_desc* d, int32_t groups);
int ssv_conv2d_wgrad_grouped(const ssv_conv_desc* d, int32_t groups, const float* x, const float* dy, float* dwd, int accumulate,
                             void* ws, size_t ws_bytes, void* stream);
/* wt[c][R-1-r][S-1-s][k] = w[k][r][s][c].  For stride 1, dgrad(dy, w) == ssv_conv2d_fwd(dy, wt) with pad' = R-1-pad: the host
 * routes stride-1 layers that way (both GEMM operands then stream k-contiguous rows; measured 5-15 % faster than the dgrad kernel) */
int ssv_filter_transpose(int32_t K, int32_t R, int32_t S, int32_t C, const float* w, float* wt, void* stream);
/* y = conv(x, w) and, from the same epilogue, the BatchNorm statistics partials of y: for every group g of 64 consecutive output rows
 * and every channel c, pmean[g][c] = mean and pm2[g][c] = centred sum of squares of that group (shifted sums around the group's first
 * row).  ssv_bn_train_fwd_partials merges them (fixed order, double) instead of reading y again.  Needs C % 32 == 0, K % 4 == 0;
 * groups = ssv_conv2d_fwd_stats_groups(d) = ceil(N*Ho*Wo / 64). */
int64_t ssv_conv2d_fwd_stats_groups(const ssv_conv_desc* d);
int ssv_conv2d_fwd_stats(const ssv_conv_desc* d, const float* x, const float* w, float* y, float* pmean, float* pm2, void* stream);
/* The fused conv -> BatchNorm -> ReLU -> conv chain (networks/resnet.py:39-42,68-73: `out = relu(bn1(conv1(x)))` feeding conv2): the
 * activation between two convolutions is never written.  `x` is the PRODUCER's raw conv output and the kernel applies
 * relu(x * in_scale[c] + in_shift[c]) while staging it (padding taps stay zero); in_scale / in_shift come from
 * ssv_bn_stats_finalize.  pmean / pm2 (both or neither) request the statistics epilogue of ssv_conv2d_fwd_stats; in_scale /
 * in_shift (both or neither) the fused input; at least one of the two.  Needs C % 32 == 0, K % 4 == 0, C <= 1024 with a fused input. */
int ssv_conv2d_fwd_bnrelu_in_stats(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift,
                                   const float* w, float* y, float* pmean, float* pm2, void* stream);
/* ssv_conv2d_wgrad whose x operand is relu(x * in_scale[c] + in_shift[c]) formed on load (same workspace); in_scale == NULL: plain */
int ssv_conv2d_wgrad_bnrelu_in(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* dy,
                               float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* BatchNorm backward, first half, in the epilogue of the convolution that produces the gradient.  When a launch computes the
 * gradient w.r.t. the OUTPUT of a BatchNorm (+ residual) + ReLU (the data gradient of the layer that consumed it:
 * networks/resnet.py:39-44,68-74 run backwards), the gated variants store g = relu'(.) * (conv result + addend) and leave, per group
 * of 64 output rows and per channel, sum g and sum g * xhat (xhat = (x - mean) * invstd) - the two reductions the BatchNorm backward
 * needs; ssv_bn_bwd_from_partials finishes it without a reduction pass over (dy, mask, x).  The ReLU bit comes from the forward's
 * byte mask (`mask`, closing BatchNorm of a residual unit) or is recomputed as x * scale + shift > 0 (`scale` / `shift`, the fused
 * chain whose activation was never written) - exactly one of the two.  psum_g / psum_gx: [groups][channels] floats, groups from the
 * matching *_gate_groups (every group is written, zeros included). */
typedef struct ssv_bn_gate {
  const float* x;               /* the BatchNorm's input: same shape as this launch's output */
  const float* scale;           /* forward affine of that BatchNorm, or NULL when `mask` is given */
  const float* shift;
  const uint8_t* mask;          /* ReLU byte mask of the BatchNorm's output (ssv_bn_train_fwd / ssv_bn_apply), or NULL */
  const float* mean;
  const float* invstd;
  float* psum_g;
  float* psum_gx;
  /* optional second reduction target (all four or none; byte-mask gate, forward-kernel entry points only): g is also the gradient w.r.t. the
   * output of the projection shortcut's BatchNorm (networks/resnet.py:72-74, no ReLU there) whose input is x2 - psum_gx2 receives
   * sum g * xhat2 (its sum g is psum_g), so that BatchNorm's backward needs no reduction pass either */
  const float* x2;
  const float* mean2;
  const float* invstd2;
  float* psum_gx2;
} ssv_bn_gate;
/* ssv_conv2d_fwd with the gate (stride-1 data gradients run on the forward kernel with the transposed filter).  C % 32 == 0, K % 4 == 0 */
int64_t ssv_conv2d_fwd_gate_groups(const ssv_conv_desc* d);
int ssv_conv2d_fwd_gated(const ssv_conv_desc* d, const float* x, const float* w, const float* addend, float* y,
                         const ssv_bn_gate* gate, void* stream);
/* The SECOND half of that backward fused into its consumers (networks/resnet.py:66-75 backwards: conv3 / downsample are 1x1): the
 * BatchNorm backward's dx = A[c] * g + B[c] * (x - mean[c]) + D[c] is formed by the consuming kernels while they stage it, from the gated
 * gradient g, the BatchNorm's input x and coef = [A | mean | B | D] ([4][channels], ssv_bn_bwd_coef) - the element-wise apply pass
 * ssv_bn_bwd_from_partials would run over (g, x) -> dx does not exist.  1x1 / stride-1 / unpadded convolutions only. */
typedef struct ssv_bn_dyin {
  const float* x;               /* input of the BatchNorm whose backward this is: same shape as g */
  const float* coef;            /* [4][channels] from ssv_bn_bwd_coef */
} ssv_bn_dyin;
/* y = conv1x1(dx(g, dyin), w) (+ addend), optional gate on y (NULL = none): the data gradient of a 1x1 convolution, run as a forward
 * convolution with the transposed filter.  d->C = channels of g (% 32 == 0), d->K % 4 == 0 */
int ssv_conv2d_fwd_dyin(const ssv_conv_desc* d, const float* g, const ssv_bn_dyin* dyin, const float* w, const float* addend, float* y,
                        const ssv_bn_gate* gate, void* stream);
/* dw (+)= x (*) dx(g, dyin); x may be a raw conv output with (in_scale, in_shift) as in ssv_conv2d_wgrad_bnrelu_in.  d->K % 4 == 0 */
int ssv_conv2d_wgrad_dyin(const ssv_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* g,
                          const ssv_bn_dyin* dyin, float* dw, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The closing activation of a residual unit, a = relu(bn3(x) + shortcut) (networks/resnet.py:73-74), formed by its first consumer - the 1x1
 * conv1 of the next unit - while it stages it, and written out by that kernel (a_out [N,H,W,C]; mask_out: its ReLU byte mask as in
 * ssv_bn_apply, or NULL): a = relu(x * scale + shift + res) or, with rscale / rshift, + res * rscale + rshift (the projection shortcut's raw
 * output and its BatchNorm affine).  Bit-identical to ssv_bn_apply followed by ssv_conv2d_fwd_stats.  1x1 / stride 1 / unpadded, C % 32 == 0. */
int ssv_conv2d_fwd_sumin_stats(const ssv_conv_desc* d, const float* x, const float* res, const float* scale, const float* shift,
                               const float* rscale, const float* rshift, const float* w, float* y, float* pmean, float* pm2,
                               float* a_out, uint8_t* mask_out, void* stream);
/* The data gradient of a 1x1 / stride-2 projection shortcut (networks/resnet.py:131-135) as a dense GEMM on the subsampled grid: it is computed by
 * ssv_conv2d_fwd on the [N][ceil(H/2)][ceil(W/2)][K] grid and handed, COMPACT, to the unit input's other contribution - conv1's data gradient,
 * which runs on the forward kernel - as its addend: addend [N][H2][W2][K] belongs to the output pixels with even (h, w), every other pixel gets
 * none.  (The full-resolution tensor of three quarters zeros, and its re-read, never exist.)  K >= 128, byte-mask gate. */
int ssv_conv2d_fwd_gated_s2add(const ssv_conv_desc* d, const float* x, const float* w, const float* addend, int32_t H2, int32_t W2, float* y,
                               const ssv_bn_gate* gate, void* stream);
int ssv_conv2d_fwd_dyin_s2add(const ssv_conv_desc* d, const float* g, const ssv_bn_dyin* dyin, const float* w, const float* addend,
                              int32_t H2, int32_t W2, float* y, const ssv_bn_gate* gate, void* stream);
/* ssv_conv2d_dgrad with the gate.  K % 32 == 0, C % 4 == 0 */
int64_t ssv_conv2d_dgrad_gate_groups(const ssv_conv_desc* d);
int ssv_conv2d_dgrad_gated(const ssv_conv_desc* d, const float* dy, const float* w, const float* addend, float* dx,
                           const ssv_bn_gate* gate, void* stream);
/* dx = conv_transpose(dy, w) (+ addend)            addend may alias dx (accumulate) or be NULL */
int ssv_conv2d_dgrad(const ssv_conv_desc* d, const float* dy, const float* w, const float* addend,
                     float* dx, void* stream);
/* dw (+)= x (*) dy     deterministic split-K over N*Ho*Wo through `ws`, then a fixed-order reduce */
size_t ssv_conv2d_wgrad_workspace_bytes(const ssv_conv_desc* d);
int ssv_conv2d_wgrad(const ssv_conv_desc* d, const float* x, const float* dy, float* dw,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);
/* weight AND bias gradient of a Linear / 1x1 / stride-1 / unpadded layer in one pass over dy (nn.Linear backward: dw (+)= dy^T x, dbias (+)= column sums
 * of dy): the weight-gradient workgroups of column tile 0 also sum the dy rows they stage; fixed-order reduction.  C % 4 == 0, K % 4 == 0. */
size_t ssv_conv2d_wgrad_bias_workspace_bytes(const ssv_conv_desc* d);
int ssv_conv2d_wgrad_bias(const ssv_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, int accumulate,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- the 3-channel image stem on the unpadded image (networks/resnet.py:96-99,147: conv7x7/2 or conv3x3/1 on [B,3,H,W]) -------------------------
 * Row-taps form: for one filter row the S taps x 3 channels of an output pixel are 3 S contiguous floats of the NHWC image row, so the
 * contraction runs over R rows of 24 floats (3 S <= 24 real ones, zero weights behind them): 168 columns for the 7x7 stem's 147, no channel
 * padding of the images.  wrows / dwrows [K][R][24]: ssv_pad_channels of the OHWI filter viewed as [K*R][3 S] (and back for the gradient). */
int ssv_stem_conv_fwd(const ssv_conv_desc* d, const float* x /*[N][H][W][3]*/, const float* wrows, float* y, float* pmean, float* pm2, void* stream);
/* Output pixels per statistics partial of ssv_stem_conv_fwd for this shape: 64 (groups = ceil(M / 64), as ssv_conv2d_fwd_stats) or a whole number of output rows
 * (the rows-in-LDS kernel: M is a multiple of it).  pmean / pm2 hold ceil(M / that) rows of K floats. */
int64_t ssv_stem_conv_fwd_stats_rows_per_group(const ssv_conv_desc* d);
size_t ssv_stem_conv_wgrad_workspace_bytes(const ssv_conv_desc* d);
/* Which kernel ssv_stem_conv_wgrad takes for this shape: whole output rows per workgroup of the rows-in-LDS kernel, 0 = the row-taps gather (tools/bench_conv.py labels
 * its rows from this and from ssv_stem_conv_fwd_stats_rows_per_group != 64). */
int64_t ssv_stem_conv_wgrad_rows_per_group(const ssv_conv_desc* d);
int ssv_stem_conv_wgrad(const ssv_conv_desc* d, const float* x, const float* dy, float* dwrows, void* ws, size_t ws_bytes, void* stream);

/* ---- Winograd F(2x2, 3x3) for stride-1 / padding-1 3x3 convolutions (csrc/winograd.hip) -------------------------------------
 * replaces nn.Conv2d(k=3, s=1, p=1) forward / data gradient / weight gradient of networks/resnet.py:7-10,56-58 on the deep stages with
 * 2.25x fewer multiplies, all in fp32:  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 2x2 output tile.
 *   T = ssv_wino_tiles(N,H,W) = N * ceil(H/2) * ceil(W/2) tiles;  transformed operands are [16][T][channels] (position p = 4 xi + nu major)
 *   forward      : U = filter_transform(w)            V = input_transform(x [, in_scale, in_shift])      M = gemm_batched(16, T, C, K, V, U)
 *                  y = output_transform(M [, statistics partials: one per 16 tiles, for ssv_bn_stats_finalize(rows_per_group = ssv_wino_stats_rows_per_group)])
 *   data gradient: the same chain on dy with the transposed, rotated filter (ssv_filter_transpose), optional ReLU gate in the output transform
 *   weight grad. : dM = dy_transform(dy)   dU = gemm_batched_wgrad(16, T, C, K, V, dM)   dw (+)= filter_grad(dU)        (V kept from the forward)
 * Every pointer 16-byte aligned, channels % 4 == 0 (gemm: C % 32 == 0).  Nothing allocates; the caller owns U, V, M, dM, dU. */
int64_t ssv_wino_tiles(int32_t N, int32_t H, int32_t W);
int64_t ssv_wino_groups(int32_t N, int32_t H, int32_t W);           /* workgroups of the output transform = statistics / gate partial groups */
int32_t ssv_wino_stats_rows_per_group(int32_t N, int32_t H, int32_t W);   /* rows per statistics partial: 64 (H, W even), H*W (one image = 16 tiles), 0 = unsupported */
int ssv_wino_filter_transform(int32_t K, int32_t C, const float* w /*[K][3][3][C]*/, float* U /*[16][K][C]*/, void* stream);
int ssv_wino_filter_grad(int32_t K, int32_t C, const float* dU /*[16][K][C]*/, float* dw /*[K][3][3][C]*/, int accumulate, void* stream);
int ssv_wino_input_transform(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, const float* in_scale, const float* in_shift,
                             float* V, void* stream);          /* in_scale / in_shift: x is a raw conv output, operand = relu(x*scale+shift); or both NULL */
int ssv_wino_dy_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, float* dM, void* stream);
int ssv_wino_output_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* M, float* y, float* pmean, float* pm2,
                              const ssv_bn_gate* gate, void* stream);

/* ---- Winograd F(4x4, 3x3) for the forward, the data gradient and (round 5, below) the weight gradient of the same layers (csrc/winograd44.hip) -----
 * 36 multiplies per channel pair and 4x4 output tile instead of 144 (F(2x2): 64), transformed input 2.25x the input instead of 4x; interpolation points
 * {0, 1, -1, 1/2, -2, inf}.  Its transforms multiply by constants (F(2x2)'s only add and halve): forward / data gradient are 2-3x the direct kernel's error
 * against fp64 (profiles/r04_probe_winograd44.txt).  ssv_wino44_input_transform can also leave the F(2x2) transformed input (V2, ssv_wino_input_transform's
 * output) from the same pass over x - round 4's weight gradient ran on it; round 5's runs F(4x4) on V itself (ssv_wino44_dy_transform ... below).
 *   T = ssv_wino44_tiles = N * ceil(H/4) * ceil(W/4); V, M: [36][T][channels]; GEMMs: ssv_gemm_batched(36, T, C, K, V, U, M).
 *   Output-transform partials: one per ROW of tiles (4 x W pixels) - for the statistics only when H % 4 == 0, else one per image (equal groups required):
 *   ssv_wino44_groups(N, H, W, stats), ssv_wino44_stats_rows_per_group (the rows_per_group for ssv_bn_stats_finalize). */
int64_t ssv_wino44_tiles(int32_t N, int32_t H, int32_t W);
int64_t ssv_wino44_groups(int32_t N, int32_t H, int32_t W, int32_t stats);
int64_t ssv_wino44_stats_rows_per_group(int32_t N, int32_t H, int32_t W);
int ssv_wino44_filter_transform(int32_t K, int32_t C, const float* w /*[K][3][3][C]*/, float* U /*[36][K][C]*/, void* stream);
int ssv_wino44_input_transform(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, const float* in_scale, const float* in_shift,
                               float* V /*[36][T][C]*/, float* V2 /*[16][ssv_wino_tiles][C] or NULL*/, void* stream);
int ssv_wino44_output_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* M /*[36][T][K]*/, float* y, float* pmean, float* pm2,
                                const ssv_bn_gate* gate, void* stream);   /* pmean / pm2 [groups][K] or NULL; gate (mask or scale + shift, no x2) or NULL */
/* The WEIGHT gradient through F(4x4) (round 5): dM = A dY A^T per 4x4 output tile [36][T][K], dU_p = dM_p^T . V_p (ssv_gemm_batched_wgrad(36, T, C, K, V, dM, dU) on
 * the V the forward's input transform left - no V2 needed then), dw (+)= G^T dU G.  0.5625x the F(2x2) weight gradient's products, a 2.25x instead of a 4x dY
 * transform.  Its error against fp64 is 1.1 - 1.6e-6 relative (F(2x2): 3 - 4e-7) - a weight gradient's error never crosses a ReLU; the bar it is held to is
 * 2e-6 relative l2 on the ResNet-50 shapes (tests/test_gpu_winograd44.py). */
int ssv_wino44_dy_transform(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, float* dM /*[36][T][K]*/, void* stream);
int ssv_wino44_filter_grad(int32_t K, int32_t C, const float* dU /*[36][K][C]*/, float* dw /*[K][3][3][C]*/, int accumulate, void* stream);
/* Both operands a layer's backward takes from its output gradient in ONE pass over it: Vd = what ssv_wino44_input_transform(dy) writes (the data gradient's transformed
 * input) and dM = what ssv_wino44_dy_transform writes, bit for bit.  dyin != NULL: `dy` is g, the gradient w.r.t. the BatchNorm output behind this convolution, and the
 * output gradient A g + B (x - mean) + D is formed per element on load (ssv_bn_bwd_coef's coefficients; zero outside the image): the BatchNorm backward's element-wise
 * pass over this layer's output - one read of g and x, one write and two reads of dy - does not exist. */
int ssv_wino44_dy_transform_both(int32_t N, int32_t H, int32_t W, int32_t K, const float* dy, const ssv_bn_dyin* dyin /* or NULL */,
                                 float* Vd /*[36][T][K]*/, float* dM /*[36][T][K]*/, void* stream);
/* batched GEMMs on the implicit-GEMM kernels, ONE launch: y[b] = a[b] . w[b]^T   /   dw[b] = dy[b]^T . x[b]   (b < batch) */
int ssv_gemm_batched(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* a /*[batch][rows][C]*/, const float* w /*[batch][K][C]*/,
                     float* y /*[batch][rows][K]*/, void* stream);
/* The same products in SSV_ARITH_BF16X3 (ssv_conv_desc.arithmetic; csrc/split_bf16.h): w_planes = ssv_split_planes(batch * K * C, w) - [3][batch][K][C] bf16, ONE call over all the batch's filters;
 * bias [K] (batch 1 only) and addend [batch][rows][K] ride in the epilogue as ssv_conv2d_fwd's do.  C % 32 == 0, K % 4 == 0.  (It replaces the same F.conv2d /
 * nn.Linear arithmetic, networks/resnet.py:56-58, networks/vit.py:17-19, in the transformed domain or directly.) */
int ssv_gemm_batched_split(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* a /*[batch][rows][C]*/, const void* w_planes /*[3][batch][K][C] bf16*/,
                           float* y /*[batch][rows][K]*/, const float* bias /*[K] or NULL (batch 1)*/, const float* addend /*[batch][rows][K] or NULL*/,
                           void* stream);
/* dw[b] = dy[b]^T . x[b] in SSV_ARITH_BF16X3: both operands split while they are staged.  max_chunk_rows / flush_rows as ssv_gemm_batched_wgrad_blocked (0 / 0: the
 * plain split with the fp32 fold); workspace: ssv_gemm_batched_wgrad_blocked_workspace_bytes(batch, rows, C, K, max_chunk_rows). */
int ssv_gemm_batched_wgrad_split(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw,
                                 int32_t max_chunk_rows, int32_t flush_rows, void* ws, size_t ws_bytes, void* stream);
size_t ssv_gemm_batched_wgrad_workspace_bytes(int32_t batch, int64_t rows, int32_t C, int32_t K);
int ssv_gemm_batched_wgrad(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x /*[batch][rows][C]*/, const float* dy /*[batch][rows][K]*/,
                           float* dw /*[batch][K][C]*/, void* ws, size_t ws_bytes, void* stream);
/* The same products with BLOCKED accumulation, for sums whose rounding error must not grow with their length (the Winograd F(4x4) weight gradient,
 * ops.wino44_conv2d_wgrad).  Two independent means: max_chunk_rows (0 or >= 32) caps the rows one workgroup accumulates (more, shorter row chunks); flush_rows
 * (0 or 128) makes the kernel add its MFMA accumulators into a second register set every 128 rows (no fp32 chain of products longer than that, whatever the chunk).
 * Either way the chunks' slabs are folded in fp64, in fixed order. */
size_t ssv_gemm_batched_wgrad_blocked_workspace_bytes(int32_t batch, int64_t rows, int32_t C, int32_t K, int32_t max_chunk_rows);
int ssv_gemm_batched_wgrad_blocked(int32_t batch, int64_t rows, int32_t C, int32_t K, const float* x, const float* dy, float* dw,
                                   int32_t max_chunk_rows, int32_t flush_rows, void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm (training mode, batch statistics) over rows of an [M][C] matrix ------------
 * replaces nn.BatchNorm2d / nn.BatchNorm1d (+ReLU, + residual add) at networks/resnet.py:39-44,
 * 68-74,147 and models/simclr.py:34-35.  C % 4 == 0.
 * fwd: mean/var per channel (shifted sums per block, Chan merge in double), running stats with
 * unbiased var, y = relu?( (x-mean)*invstd*gamma + beta (+ residual) ).
 * relu_mask (may be NULL; M*C/4 bytes): one byte per group of four channels, bit e set where output e is positive -
 * the backward then reads 1 byte instead of 16 bytes of y (pass the same buffer to ssv_bn_train_bwd, or y). */
size_t ssv_bn_workspace_bytes(int64_t M, int32_t C);
int ssv_bn_train_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta,
                     const float* residual, int relu, float eps, float momentum,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* y, uint8_t* relu_mask, float* save_mean, float* save_invstd,
                     void* ws, size_t ws_bytes, void* stream);
/* bwd: g = dy * (y>0 if relu); dgamma (+)= sum g*xhat; dbeta (+)= sum g;
 * dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dresidual = g if not NULL. */
int ssv_bn_train_fwd_partials(int64_t M, int32_t C, const float* x, const float* pmean, const float* pm2, int32_t rows_per_group,
                              const float* gamma, const float* beta, const float* residual, int relu, float eps, float momentum,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float* y, uint8_t* relu_mask, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream);
int ssv_bn_train_bwd(int64_t M, int32_t C, const float* dy, const float* y, const uint8_t* relu_mask, const float* x,
                     const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                     float* dx, float* dresidual, float* dgamma, float* dbeta, int accumulate,
                     void* ws, size_t ws_bytes, void* stream);
/* Pieces of the fused path.  ssv_bn_stats_finalize: statistics partials (ssv_conv2d_fwd_stats layout) -> save_mean / save_invstd,
 * the affine scale = gamma * invstd, shift = beta - mean * scale that the consumer convolution applies on load, and the running
 * statistics update - BatchNorm without its apply pass.  ssv_bn_apply: y = relu?(x * scale + shift (+ residual)) where the residual
 * is either a materialised tensor (res_scale NULL) or the raw output of the projection shortcut's convolution with ITS BatchNorm
 * folded in (residual * res_scale + res_shift: networks/resnet.py:71-73 `identity = downsample(x); out += identity; relu`).
 * ssv_bn_relu_bwd_affine: ssv_bn_train_bwd for a BatchNorm + ReLU whose output was never written - the ReLU gate is recomputed
 * as x * scale + shift > 0 from the forward's own scale / shift (bit-identical to the forward's decision). */
int ssv_bn_stats_finalize(int64_t M, int32_t C, const float* pmean, const float* pm2, int32_t rows_per_group,
                          const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float* save_mean, float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);
int ssv_bn_apply(int64_t M, int32_t C, const float* x, const float* scale, const float* shift,
                 const float* residual, const float* res_scale, const float* res_shift, int relu,
                 float* y, uint8_t* relu_mask, void* stream);
int ssv_bn_relu_bwd_affine(int64_t M, int32_t C, const float* dy, const float* x, const float* gamma,
                           const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                           float* dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* Second half of the BatchNorm backward behind a gated convolution: g is already relu-gated, psum_g / psum_gx hold the partial sums.
 * dgamma (+)= sum g * xhat, dbeta (+)= sum g, dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); dx must not alias g. */
int ssv_bn_bwd_from_partials(int64_t M, int32_t C, const float* g, const float* x, const float* gamma,
                             const float* save_mean, const float* save_invstd, const float* psum_g, const float* psum_gx, int64_t groups,
                             float* dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same merge WITHOUT the apply pass: dgamma / dbeta as above and coef = [A | mean | B | D] ([4][C]) of
 * dx = A * g + B * (x - mean) + D, for ssv_conv2d_fwd_dyin / ssv_conv2d_wgrad_dyin. */
int ssv_bn_bwd_coef(int64_t M, int32_t C, const float* gamma, const float* save_mean, const float* save_invstd,
                    const float* psum_g, const float* psum_gx, int64_t groups, float* coef, float* dgamma, float* dbeta, int accumulate,
                    void* ws, size_t ws_bytes, void* stream);
/* The image stem's BatchNorm + ReLU + MaxPool2d(3, 2, 1) (networks/resnet.py:147-148 `maxpool(relu(bn1(conv1(x))))`) as one pass each way:
 * forward reads the raw conv output y [N,H,W,C] once and writes only the pooled map [N,Ho,Wo,C] + its argmax slots (scale / shift from
 * ssv_bn_stats_finalize); backward forms the gradient w.r.t. the BatchNorm output on the fly (gather over the <= 2x2 windows that chose the
 * pixel, ReLU gate recomputed) inside the BatchNorm backward's reduction and apply passes - bit-identical to ssv_maxpool3x3s2_bwd followed by
 * ssv_bn_train_bwd, without the two full-resolution intermediates.  Workspace: ssv_bn_workspace_bytes(N*H*W, C). */
int ssv_bn_relu_maxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* y, const float* scale, const float* shift,
                            float* out, uint8_t* argmax, float* xmax, void* stream);
/* xmax (optional, [N][Ho][Wo][C] like out): the forward also leaves the RAW conv output y at every window's arg-max pixel; given back to the backward,
 * its reduction pass (sum g, sum g * xhat) runs over the pooled positions - reading dpool and xmax instead of walking the full-resolution y.  Same terms,
 * grouped per window instead of per pixel: equal to the full-resolution reduction to rounding; NULL on either side = the full-resolution walk. */
int ssv_bn_relu_maxpool_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dpool, const uint8_t* argmax, const float* y, const float* xmax,
                            const float* gamma, const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                            float* dy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* out[c] (+)= sum_m x[m][c]   (bias gradient of nn.Linear); same workspace size as BN */
int ssv_colsum(int64_t M, int32_t C, const float* x, float* out, int accumulate,
               void* ws, size_t ws_bytes, void* stream);

/* ---- pooling / layout --------------------------------------------------------------------
 * replaces nn.MaxPool2d(3,2,1) networks/resnet.py:102,148 and AdaptiveAvgPool2d(1)+Flatten :107-108 */
int ssv_maxpool3x3s2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, float* y,
                         uint8_t* argmax, void* stream);
int ssv_maxpool3x3s2_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dy,
                         const uint8_t* argmax, float* dx, void* stream);
int ssv_gap_fwd(int32_t N, int32_t HW, int32_t C, const float* x, float* y, void* stream);
int ssv_gap_bwd(int32_t N, int32_t HW, int32_t C, const float* dy, float* dx, void* stream);
/* the reference boundary hands NCHW fp32 batches (models/simclr.py:87) */
int ssv_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, const float* in, float* out, void* stream);
int ssv_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float* in, float* out, void* stream);

/* ---- F.normalize(p=2, dim=-1) : utils/losses.py:20-22, models/byol.py:47,59, models/barlow.py:35
 * zhat is written with row stride ldo >= D, columns [D, ldo) zero-filled.  normalize=0 copies. */
int ssv_l2norm_fwd(int32_t rows, int32_t D, const float* z, int32_t normalize, float eps,
                   float* zhat, int32_t ldo, float* inv_norm, void* stream);
/* dz = (dzhat - zhat*(zhat.dzhat)) * inv_norm       (normalize=0: dz = dzhat) */
int ssv_l2norm_bwd(int32_t rows, int32_t D, const float* zhat, int32_t ldz, const float* inv_norm,
                   const float* dzhat, int32_t ldd, int32_t normalize, float* dz, void* stream);

/* ---- NT-Xent : SimclrLoss.forward utils/losses.py:15-46 in Gram / log-sum-exp form --------
 * Z is the (gathered) [2*Nglob][ldz] matrix [zi_all ; zj_all] (ldz % 32 == 0, ldz <= 128,
 * pad columns zero).  This rank owns local rows lr in [0, 2*Bloc): global row
 * r = seg0 + lr (lr < Bloc) or Nglob + seg0 + (lr - Bloc).
 * fwd: lse[lr] = logsumexp_{c != r} Z_r.Z_c * inv_temp ; pos[lr] = Z_r.Z_pos(r) * inv_temp.
 * loss = sum_lr (lse - pos) / (2*Nglob)  (ssv_ntxent_loss; all-reduce SUM across ranks).
 * bwd: needs lse of ALL 2*Nglob rows (all-gathered); writes dZ for the local rows:
 *   dZ_r = gscale * ( sum_{c != r} (e^{S_rc-lse_r} + e^{S_rc-lse_c}) Z_c - 2 Z_pos(r) ),
 *   gscale = dloss * inv_temp / (2*Nglob). */
int ssv_ntxent_fwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                   float inv_temp, float* lse, float* pos, void* stream);
int ssv_ntxent_loss(int32_t rows, const float* lse, const float* pos, float scale, float* loss, void* stream);
int ssv_ntxent_bwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                   const float* lse_all, float inv_temp, float gscale, float* dZ, void* stream);
/* The same two passes with the COLUMN sweep cut into `splits` runs of column tiles, one workgroup per (32 rows, run): a rank of the
 * 8-GPU job owns 2*512 rows against 2*4096 gathered columns (BASELINE config 3) - 32 row blocks alone would leave 7/8 of the CUs idle.
 * Partials go through the caller's workspace (ssv_ntxent_split_workspace_bytes; [splits][2*Bloc][max(ldz,3)] floats) and are folded in
 * split order (deterministic).  splits = 1 is ssv_ntxent_fwd / _bwd (no workspace).  ssv_ntxent_default_splits is the library's own
 * choice for a shape (a count, not a status: two workgroups per CU where the columns allow, >= 8 column tiles per run). */
int64_t ssv_ntxent_default_splits(int32_t Nglob, int32_t Bloc);
size_t ssv_ntxent_split_workspace_bytes(int32_t Bloc, int32_t ldz, int32_t splits);
int ssv_ntxent_fwd_split(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                         float inv_temp, float* lse, float* pos, int32_t splits, void* ws, size_t ws_bytes, void* stream);
int ssv_ntxent_bwd_split(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t ldz, const float* Z,
                         const float* lse_all, float inv_temp, float gscale, float* dZ,
                         int32_t splits, void* ws, size_t ws_bytes, void* stream);
/* The same loss for projection widths beyond the register-resident kernels (ldz > 128; the reference accepts any D): the host forms the
 * Gram block S[2*Bloc][lds] = Z_loc . Z_all^T with ssv_conv2d_fwd (lds = 2*Nglob rounded up to 16, pad columns are products with zero
 * rows), ssv_ntxent_gram_fwd reads lse / pos off it, ssv_ntxent_gram_weights turns it IN PLACE into
 * W'[r][c] = gscale * (e^{S_rc-lse_r} + e^{S_rc-lse_c} - 2 [c = pos(r)]) (0 on the diagonal and the pad), and dZ = W' . Z_all is
 * ssv_conv2d_dgrad - two GEMMs on the MFMA kernels, no width limit. */
int ssv_ntxent_gram_fwd(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t lds, const float* S, float inv_temp,
                        float* lse, float* pos, void* stream);
int ssv_ntxent_gram_weights(int32_t Nglob, int32_t Bloc, int32_t seg0, int32_t lds, float* S, const float* lse_all,
                            float inv_temp, float gscale, void* stream);

/* ---- BYOL loss: nn.MSELoss pair models/byol.py:89,129-130 on [B][D] matrices ---------------
 * loss = (sum (o1-t2)^2 + sum (o2-t1)^2) / (B*D); do1 = 2*(o1-t2)*gscale, do2 likewise. */
int ssv_mse_pair_fwd_bwd(int64_t n, const float* o1, const float* o2, const float* t1, const float* t2,
                         float inv_count, float* loss, float* do1, float* do2,
                         void* ws, size_t ws_bytes, void* stream);
size_t ssv_reduce_workspace_bytes(int64_t n);
/* x[i] *= *factor_dev (factor read on the device: no host sync) - chain-rule scale by an upstream grad */
int ssv_scale(int64_t n, float* x, const float* factor_dev, void* stream);

/* ---- Barlow Twins loss: BarlowLoss.forward utils/losses.py:127-142 --------------------------
 * The column standardisation (unbiased std, no eps) is ssv_bn_train_fwd/bwd with gamma = sqrt((B-1)/B), beta = 0,
 * eps = 0; the D x D cross-correlation and the two gradient products are ssv_conv2d_wgrad / _fwd / _dgrad (MFMA).
 * This entry point is the element-wise middle: from Craw = zi_hat^T zj_hat to loss = sum W o (Craw/B - I)^2 and
 * G = dloss/dCraw-side factor 2 W o (Craw/B - I) / B.  Workspace: ssv_reduce_workspace_bytes(D*D). */
int ssv_barlow_cgrad(int32_t D, const float* craw, float inv_b, float lambda, float* loss, float* G,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- optimizer: optim.SGD(momentum=0.9, nesterov=True, weight_decay) utils/train_utils.py:11-13
 * over a flat arena of n floats.  first_step != 0 seeds buf = g.  g2 (may be NULL) is a second gradient slab that is
 * added to g first: the two views' backward passes run on two HIP streams and accumulate into separate slabs. */
int ssv_sgd_nesterov(int64_t n, float* p, const float* g, const float* g2, float* buf, float lr, float weight_decay,
                     float momentum, int first_step, void* stream);
/* the same update with (lr, weight_decay, momentum, first_step != 0) read from hyper[0..3] in DEVICE memory: the launch of a step replayed as a HIP graph, whose
 * schedules (utils/train_utils.py:30-37, models/simclr.py:77-84) then move device floats instead of kernel arguments */
int ssv_sgd_nesterov_dev(int64_t n, float* p, const float* g, const float* g2, float* buf, const float* hyper /*[4]*/, void* stream);
/* BYOL.momentum_update models/byol.py:120-123: t = tau*t + (1-tau)*o over n floats */
int ssv_ema(int64_t n, float* target, const float* online, float tau, void* stream);
int ssv_fill(int64_t n, float* p, float value, void* stream);
/* dst[i] += src[i]  (gradient accumulation where no producer kernel can fuse it) */
int ssv_add(int64_t n, float* dst, const float* src, void* stream);

/* ---- two-view augmentation (R1): the chain of configs/simclr.yaml:13-29 on the GPU ----------
 * replaces DoubleAugmentedDataset.__getitem__ utils/data_utils.py:68-73 / get_transform
 * utils/augmentations.py:128-144 (torchvision 0.9.1 + Pillow 8.3.1 in DataLoader workers).
 * src: uint8 [nsrc][Hs][Ws][3] (PIL layout) resident in HBM; sample_ids (device int64[B], may be NULL = 0..B-1)
 * selects the rows.  params: [nviews][B][SSV_AUG_NPARAM] float32 records
 *   [0] jitter on, [1..4] op order (0 brightness 1 contrast 2 saturation 3 hue), [5..8] the four factors,
 *   [9] gray on, [10..13] crop top,left,height,width, [14] flip on
 * drawn by ssv_augment_params from Philox4x32-10 keyed (seed; sample id, view, step) with torchvision's
 * distributions, or supplied by the caller.  out: fp32 NHWC [nviews][B][Ho][Wo][3], normalised.
 * Pixel arithmetic is Pillow's, bit for bit (oracle/augment.py).  mean3/std3 are HOST pointers to 3 floats. */
#define SSV_AUG_NPARAM 16
typedef struct ssv_aug_cfg {
  double brightness, contrast, saturation, hue;     /* ColorJitter ranges: factor in [max(0,1-x), 1+x], hue in [-h, h] */
  double p_jitter, p_gray, p_flip;                  /* RandomApply p, RandomGrayscale p, RandomHorizontalFlip p */
  double scale_min, scale_max, ratio_min, ratio_max;/* RandomResizedCrop */
} ssv_aug_cfg;
int ssv_augment_params(int32_t B, int32_t Hs, int32_t Ws, int32_t nviews, const ssv_aug_cfg* cfg, uint64_t seed, uint64_t step,
                       const int64_t* sample_ids, int64_t sample0, float* params, void* stream);
size_t ssv_augment_workspace_bytes(int32_t B, int32_t nviews, int32_t Ho, int32_t Wo);
int ssv_augment_views(int32_t B, int32_t nviews, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo,
                      const uint8_t* src, const int64_t* sample_ids, const float* params,
                      const float* mean3_host, const float* std3_host, float* out,
                      void* ws, size_t ws_bytes, void* stream);
/* CenterCrop -> ToTensor -> Normalize: the "img" entry of the batch (configs/simclr.yaml:24-29) */
int ssv_center_view(int32_t B, int32_t Hs, int32_t Ws, int32_t Ho, int32_t Wo, const uint8_t* src, const int64_t* sample_ids,
                    const float* mean3_host, const float* std3_host, float* out, void* stream);

/* ---- kNN evaluation: compute_neighbor_accuracy utils/eval_utils.py:13-21 (faiss.IndexFlatIP search of every feature
 * vector against the whole set, k+1 hits, the best one dropped, labels of the other k compared with the query's).
 * z [n][d] fp32 (d % 4 == 0), labels [n] int32, 1 <= k <= min(63, n-1).  *count (device, 8 bytes) receives the number of
 * (query, neighbour) pairs with equal labels; accuracy = count / (n*k).  Order: inner product descending, ties by index.
 * S = Z Z^T is formed chunk by chunk in `ws` by the MFMA implicit-GEMM kernel; one wavefront per query keeps the top-(k+1). */
size_t ssv_knn_workspace_bytes(int64_t n);
int ssv_knn_label_agreement(int64_t n, int32_t d, const float* z, const int32_t* labels, int32_t k,
                            unsigned long long* count, void* ws, size_t ws_bytes, void* stream);
/* The same search with the Gram product Z Z^T in the given arithmetic (SSV_ARITH_F32_MFMA | SSV_ARITH_BF16X3; widths without a bf16-piece kernel - d % 32 != 0 - run on
 * fp32 MFMA either way).  Workspace: ssv_knn_workspace_bytes_arith(n, d, arithmetic) (the bf16x3 form also holds the three planes of z). */
size_t ssv_knn_workspace_bytes_arith(int64_t n, int32_t d, int32_t arithmetic);
int ssv_knn_label_agreement_arith(int64_t n, int32_t d, const float* z, const int32_t* labels, int32_t k,
                                  unsigned long long* count, int32_t arithmetic, void* ws, size_t ws_bytes, void* stream);

/* ==== "next" row 1 of the scope table: DINO on the reference's ViT (networks/vit.py, models/dino.py) ====================
 * Linear layers of the encoder and of the projection head are ssv_conv2d_fwd/dgrad/wgrad with H = W = R = S = 1 over
 * N = B*T token rows; what follows are the pieces that are not GEMMs.  All matrices are dense row-major fp32. */

/* EmbeddingLayer + nn.Unfold (networks/vit.py:71-82, :101-102): tokens[b][0] = [cls | pos[0]], tokens[b][1+p] =
 * [patch p of image b in (channel, kh, kw) order | pos[1+p]];  T = (H/patch)(W/patch)+1, row length 3*patch^2 + E.
 * img: [B][H][W][3] (channels-last).  bwd: dcls[f] = sum_b dtokens[b][0][f], dpos[t][e] = sum_b dtokens[b][t][3p^2+e]. */
int ssv_vit_embed_fwd(int32_t B, int32_t H, int32_t W, int32_t patch, int32_t E, const float* img_nhwc,
                      const float* cls, const float* pos, float* tokens, void* stream);
int ssv_vit_embed_bwd(int32_t B, int32_t T, int32_t P3, int32_t E, const float* dtokens, float* dcls, float* dpos,
                      int32_t accumulate, void* stream);

/* nn.LayerNorm over the last axis of [M][C] (networks/vit.py:19,40; biased variance, eps inside the sqrt) with a fused
 * addend: y = LN(x) * gamma + beta (+ addend) - the reference's "f(x) + LayerNorm(x)" residual form (:22-31, :43-46).
 * bwd: dx = LN'(dy) (+ dx_addend), dgamma/dbeta (+)= column sums (per-block partials in `ws`, fixed-order final reduce).
 * C % 4 == 0 and C <= 2048 (a row is held in one wavefront's registers), pointers 16-byte aligned. */
int ssv_layernorm_fwd(int64_t M, int32_t C, const float* x, const float* gamma, const float* beta, const float* addend,
                      float eps, float* y, float* mean, float* invstd, void* stream);
size_t ssv_layernorm_workspace_bytes(int64_t M, int32_t C);
int ssv_layernorm_bwd(int64_t M, int32_t C, const float* dy, const float* x, const float* gamma, const float* mean,
                      const float* invstd, const float* dx_addend, float* dx, float* dgamma, float* dbeta,
                      int32_t accumulate, void* ws, size_t ws_bytes, void* stream);

/* Feedforward.fc1 + nn.GELU in one pass (networks/vit.py:42,45): h = x w^T + bias and act = gelu(h), both written by the GEMM
 * epilogue (h is kept for the backward); and the matching backward of fc2: dx = (dy w) * gelu'(h) (+ addend).  Linear layers only
 * (H = W = R = S = 1); forward: C % 32 == 0, K >= 128; dgrad: K % 32 == 0, C >= 128. */
/* h == NULL: only act is written (a forward nobody differentiates: the teacher / target passes) */
int ssv_linear_gelu_fwd(const ssv_conv_desc* d, const float* x, const float* w, const float* bias, float* h, float* act, void* stream);
int ssv_conv2d_dgrad_gelu(const ssv_conv_desc* d, const float* dy, const float* w, const float* h, const float* addend,
                          float* dx, void* stream);
/* the same backward on the forward kernel, given the TRANSPOSED weights wt [C_in][K_out] of fc2 (ssv_filter_transpose): d describes the GEMM as
 * launched - rows of dy with d->C columns (fc2's outputs, % 32), d->K result columns (fc2's inputs = the width of h, % 4, >= 128) */
int ssv_linear_fwd_gelugrad(const ssv_conv_desc* d, const float* dy, const float* wt, const float* h, const float* addend,
                            float* dh, void* stream);
/* The same pair with the derivative taken in the forward: fc1's epilogue writes dact = gelu'(h) in the pre-activation's place (nothing in the backward of
 * fc1 -> GELU -> fc2 reads h itself) and act = gelu(h); the backward is dh = (dy wt^T) * dact (+ addend) - a plain multiply, no erf / exp in its epilogue.
 * Bit-identical to ssv_linear_gelu_fwd + ssv_linear_fwd_gelugrad (the same cdf / pdf expressions, evaluated once instead of twice). */
int ssv_linear_gelu_fwd_dact(const ssv_conv_desc* d, const float* x, const float* w, const float* bias, float* dact, float* act, void* stream);
int ssv_linear_fwd_mulgrad(const ssv_conv_desc* d, const float* dy, const float* wt, const float* dact, const float* addend,
                           float* dh, void* stream);
/* nn.GELU() (erf form; networks/vit.py:38, models/dino.py:30-33) on n floats, n % 4 == 0.  Every GELU of the library (these two, the epilogues above) evaluates erf
 * by the same fp32 polynomial pair (csrc/common.h::ssv_erf: 1.3e-7 absolute over all x). */
int ssv_gelu_fwd(int64_t n, const float* x, float* y, void* stream);
int ssv_gelu_bwd(int64_t n, const float* x, const float* dy, float* dx, void* stream);

/* MultiheadSelfAttention core (networks/vit.py:24-30): o[b][t][h*dh..] = softmax(q k^T * scale) v per (image, head), without
 * materialising the T x T probabilities (the reference returns them only for visualisation).  q/k/v rows have stride `ld`
 * (so a fused [M][3*hidden] projection can be passed as three pointers), o/dout stride `ldo`, gradients stride `ldg`.
 * dh must be 64.  lse and delta: [B][heads][T] (log-sum-exp of the scaled scores; rowsum(dout * o)). */
int ssv_attention_fwd(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                      int32_t ld, float scale, float* o, int32_t ldo, float* lse, void* stream);
/* ssv_attention_fwd in the given arithmetic (SSV_ARITH_F32_MFMA | SSV_ARITH_BF16X3: Q K^T and P V as six bf16 piece products per fp32 product, csrc/split_bf16.h;
 * softmax statistics and outputs fp32 either way) */
int ssv_attention_fwd_arith(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                            int32_t ld, float scale, float* o, int32_t ldo, float* lse, int32_t arithmetic, void* stream);
int ssv_attention_bwd(int32_t B, int32_t T, int32_t heads, int32_t dh, const float* q, const float* k, const float* v,
                      int32_t ld, float scale, const float* o, const float* dout, int32_t ldo, const float* lse,
                      float* delta, float* dq, float* dk, float* dv, int32_t ldg, void* stream);

/* nn.utils.weight_norm(nn.Linear) (models/dino.py:35): w[r][:] = g[r] * v[r][:] / ||v[r]||; bwd -> dg, dv from dw */
int ssv_weightnorm_fwd(int32_t rows, int32_t cols, const float* g, const float* v, float* w, float* inv_norm, void* stream);
int ssv_weightnorm_bwd(int32_t rows, int32_t cols, const float* dw, const float* g, const float* v, const float* inv_norm,
                       float* dg, float* dv, int32_t accumulate, void* stream);

/* DinoLoss.forward (utils/losses.py:80-89): teacher [bs][2][K], student [bs][V][K], center [K];
 * loss (+)= weight * sum_g mean_{b,v}( - softmax((teacher[b][g]-center)/temp_t) . log_softmax(student[b][v]/temp_s) ),
 * dstudent = d(weight * that)/dstudent.  The caller passes weight 0.5 for each of the two pairs (models/dino.py:161-163). */
size_t ssv_dino_loss_workspace_bytes(int32_t bs, int32_t V, int32_t K);
int ssv_dino_loss(int32_t bs, int32_t V, int32_t K, const float* teacher, const float* student, const float* center,
                  float temp_s, float temp_t, float weight, float* loss, int32_t accumulate_loss, float* dstudent,
                  void* ws, size_t ws_bytes, void* stream);
/* update_teacher_center (models/dino.py:136-141): center = m*center + (1-m)*mean over the rows of t1 and t2 */
int ssv_dino_center_update(int32_t K, int32_t rows1, const float* t1, int32_t rows2, const float* t2, float momentum,
                           float* center, void* stream);

/* optim.AdamW (utils/train_utils.py:17-19) over a flat arena; g2 = optional second gradient slab; clip > 0 applies the
 * reference's clamp hooks torch.clamp(grad, -clip, clip) (models/dino.py:76-79) to the summed gradient first.  step >= 1. */
int ssv_adamw(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float lr, float beta1, float beta2,
              float eps, float weight_decay, int64_t step, float clip, void* stream);
/* The same update with the step count in device memory (*step_dev += 1 first; bc_dev: two floats of caller-owned scratch for the bias corrections): no launch
 * argument changes from step to step, so the step can be replayed as a HIP graph. */
int ssv_adamw_counted(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int64_t* step_dev, float* bc_dev, float clip, void* stream);
/* ssv_adamw_counted with the learning rate and the weight decay read from bc_dev[2], bc_dev[3] (device memory the host rewrites when a schedule moves them): no
 * argument of the launch changes over a whole run - one captured HIP graph of the step serves every epoch (ssv_amd/graph.py). */
int ssv_adamw_counted_dev(int64_t n, float* p, const float* g, const float* g2, float* m, float* v, float beta1, float beta2,
                          float eps, int64_t* step_dev, float* bc_dev /*[4]*/, float clip, void* stream);

/* MultiCrop (utils/augmentations.py:156-173): RandomResizedCrop(scale, ratio 3/4..4/3, BICUBIC) boxes drawn from the Philox
 * stream (seed, step, sample, view_base + crop), view_base >= 16; then crop + bicubic resize (align_corners = False, A = -0.75,
 * taps clamped to the box) of already normalised float views [B][Hs][Ws][3] into [B][ncrop][Ho][Wo][3]. */
int ssv_multicrop_params(int32_t B, int32_t Hs, int32_t Ws, int32_t ncrop, int32_t view_base, double scale_min, double scale_max,
                         uint64_t seed, uint64_t step, const int64_t* sample_ids, int64_t sample0, int32_t* boxes, void* stream);
int ssv_multicrop(int32_t B, int32_t Hs, int32_t Ws, const float* views_nhwc, int32_t ncrop, const int32_t* boxes,
                  int32_t Ho, int32_t Wo, float* out_nhwc, void* stream);

/* ==== "next" row 4: sibling two-view algorithms (they reuse every encoder / head kernel above) ========================= */
/* SimSiamLoss (utils/losses.py:145-152) for both pairs: loss = -scale * (sum o1.t2 + sum o2.t1), do1 = -scale*t2, do2 = -scale*t1
 * (scale = 0.5 / batch for models/simsiam.py:126-127).  Workspace: ssv_reduce_workspace_bytes(n). */
int ssv_negdot_pair_fwd_bwd(int64_t n, const float* o1, const float* o2, const float* t1, const float* t2, float scale,
                            float* loss, float* do1, float* do2, void* ws, size_t ws_bytes, void* stream);
/* RelicLoss invariance term (utils/losses.py:195-201): a = diag(zi zo^T)/T, b = diag(zj zo^T)/T over dense [N][D] matrices,
 * loss (+)= alpha * sum_n exp(lq_n) (lq_n - p_n) with p = softmax(a), lq = log_softmax(b) ACROSS the batch; writes the gradients
 * w.r.t. zi, zj, zo.  (The contrastive half of RelicLoss is ssv_ntxent_*.) */
size_t ssv_relic_kl_workspace_bytes(int32_t N);
int ssv_relic_kl_fwd_bwd(int32_t N, int32_t D, const float* zi, const float* zj, const float* zo, float inv_temp, float alpha,
                         float* loss, int32_t accumulate_loss, float* dzi, float* dzj, float* dzo,
                         void* ws, size_t ws_bytes, void* stream);
/* MocoLoss (utils/losses.py:49-71) given neg[n][j] = q[n].bank[j] (N x K products, row stride ldk, from ssv_conv2d_fwd):
 * loss = mean_n(logsumexp([q.k/T | neg/T]) - q.k/T); neg is overwritten by d loss / d neg (columns K..ldk zeroed) and
 * dq_init[n] = d loss / d(q.k) * k[n]; the caller finishes dq = dq_init + neg . bank with ssv_conv2d_dgrad's addend. */
int ssv_moco_loss_fwd_bwd(int32_t N, int32_t D, int32_t K, int32_t ldk, const float* q, const float* k, float* neg, float inv_temp,
                          float* loss, float* dq_init, void* ws, size_t ws_bytes, void* stream);
/* MemoryBank.add_batch (models/moco.py:32-37): bank[(ptr+i) % K] = keys[i] / max(||keys[i]||, eps) for i < n */
int ssv_queue_push(int32_t K, int32_t D, float* bank, int32_t ptr, int32_t n, const float* keys, float eps, void* stream);
/* The same push with the queue pointer in device memory (read by the kernel, advanced behind it): no launch argument changes from step to step (HIP-graph replay). */
int ssv_queue_push_counted(int32_t K, int32_t D, float* bank, int32_t* ptr_dev, int32_t n, const float* keys, float eps, void* stream);

/* linear probe (utils/eval_utils.py:37-76): NLLLoss(log_softmax(logits)) and accuracy of a [N][ld] logit matrix (first C columns
 * valid) against int32 labels; stats[0] = mean loss, stats[1] = fraction of rows whose arg-max is the label; dlogits (may be NULL)
 * = d(mean loss)/dlogits, padding columns zeroed.  Workspace: ssv_softmax_ce_workspace_bytes(N). */
size_t ssv_softmax_ce_workspace_bytes(int32_t N);
int ssv_softmax_ce_fwd_bwd(int32_t N, int32_t C, int32_t ld, const float* logits, const int32_t* labels, float* stats,
                           float* dlogits, void* ws, size_t ws_bytes, void* stream);
/* optim.SGD(lr, momentum, weight_decay, nesterov) over n floats (the probe's optimiser, utils/eval_utils.py:42) */
int ssv_sgd(int64_t n, float* p, const float* g, float* buf, float lr, float weight_decay, float momentum, int nesterov,
            int first_step, void* stream);

/* ---- per-kernel-class timing with HIP events on the launch stream (bench.py roofline) -------
 * classes: see SSV_PROF_* ; when enabled every entry point brackets its launches with an
 * event pair on `stream`.  ssv_prof_collect synchronises the events (not the device).
 * Threading: this is the library's only process-wide mutable state besides the thread-local error string.  It is OFF by default (an
 * entry point then pays one relaxed atomic load); when on, the record list is guarded by a mutex and a launch's begin / end pair is
 * matched through a thread-local slot, so host threads driving different streams may call concurrently.  The library reads no
 * environment variables and keeps no other state between calls. */
enum { SSV_PROF_CONV_FWD = 0, SSV_PROF_CONV_DGRAD, SSV_PROF_CONV_WGRAD, SSV_PROF_BN_FWD, SSV_PROF_BN_BWD,
       SSV_PROF_POOL, SSV_PROF_LOSS, SSV_PROF_OPTIM, SSV_PROF_AUG, SSV_PROF_MISC, SSV_PROF_ATTN, SSV_PROF_NORM, SSV_PROF_NCLASS };
int ssv_prof_enable(int on);
int ssv_prof_reset(void);
int ssv_prof_collect(double* ms_per_class, int64_t* launches_per_class);   /* HOST arrays [SSV_PROF_NCLASS] */

#ifdef __cplusplus
}
#endif
#endif /* SSV_HIP_H */
