// ssv_split_planes: an fp32 tensor as three bf16 planes (SSV_ARITH_BF16X3, csrc/split_bf16.h) - what the forward / data-gradient launches take as their pre-split
// weight operand: made once per weight (and transposed / Winograd-transformed weight) and step instead of once per staged tile.  HBM-bound: 4 B read + 6 B written
// per element (ResNet-50 + heads, filters and their transposes: 56 M elements = 0.09 ms per step at 6.29 TB/s).
#include "common.h"
#include "split_bf16.h"

namespace {
__global__ void __launch_bounds__(256)
split_planes_k(int64_t n4, const f32x4* __restrict__ x, u32x2* __restrict__ p0, u32x2* __restrict__ p1, u32x2* __restrict__ p2) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    u32x2 pl[3];
    splitbf::split4(x[i], pl);
    p0[i] = pl[0]; p1[i] = pl[1]; p2[i] = pl[2];
  }
}
}  // namespace

extern "C" int ssv_split_planes(int64_t n, const float* x, void* planes, void* stream) {
  SSV_REQUIRE(n > 0 && n % 4 == 0 && x && planes, "ssv_split_planes: bad arguments (n = %lld must be a positive multiple of 4)", (long long)n);
  SSV_REQUIRE((((uintptr_t)x | (uintptr_t)planes) & 15) == 0 && (n * 2) % 16 == 0, "ssv_split_planes: pointers and plane size must be 16-byte aligned (n %% 8 == 0)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SSV_PROF_MISC, s);
  const int64_t n4 = n >> 2;
  int64_t b = cdiv64(n4, 256);
  if (b > 4096) b = 4096;
  unsigned short* pp = (unsigned short*)planes;
  hipLaunchKernelGGL(split_planes_k, dim3((unsigned)b), dim3(256), 0, s, n4, (const f32x4*)x, (u32x2*)pp, (u32x2*)(pp + n), (u32x2*)(pp + 2 * n));
  SSV_CHECK_LAUNCH("ssv_split_planes");
  return SSV_OK;
}
