// Factorization-machine second-order term on an arbitrary (B, F, D) tensor.
//
// Reference: FM.forward torch_rechub/basic/layers.py:313-319
//     square_of_sum = sum_f(x)^2 ; sum_of_square = sum_f(x^2) ; ix = square_of_sum - sum_of_square
//     reduce_sum -> 0.5 * sum_d(ix) (B,1) else 0.5 * ix (B,D)
// (the DeepFM path uses the fused kernel in embed.hip; this one serves FM called on its own, e.g. AFM)
// Roofline: HBM, F*D*4 bytes read per sample.  G = next_pow2(D) lanes per sample, lane = one d.
#include "common.h"

namespace {

template <int G>
__global__ __launch_bounds__(RH_BLOCK) void fm_fwd_kernel(const float* __restrict__ x, int64_t xs, int B, int F,
                                                          int D, int reduce_sum, float* __restrict__ out) {
  RH_CHAIN_PRIO();
  const int lig = threadIdx.x % G;
  int64_t b = (int64_t)blockIdx.x * (RH_BLOCK / G) + threadIdx.x / G;
  const bool live = b < B;
  if (!live) b = B - 1;
  const float* row = x + b * xs;
  float tot = 0.f;
  for (int d = lig; d < D; d += G) {
    float s = 0.f, ss = 0.f;
    for (int f = 0; f < F; ++f) {
      const float v = row[(int64_t)f * D + d];
      s += v;
      ss = fmaf(v, v, ss);
    }
    const float ix = s * s - ss;
    if (!reduce_sum) {
      if (live) out[b * D + d] = 0.5f * ix;
    } else {
      tot += ix;
    }
  }
  if (reduce_sum) {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) tot += __shfl_xor(tot, m, RH_WAVE);
    if (live && lig == 0) out[b] = 0.5f * tot;
  }
}

// d out / d x[b,f,d] = g * (S[b,d] - x[b,f,d]),  g = g_out[b] (reduce_sum) or g_out[b,d]
template <int G>
__global__ __launch_bounds__(RH_BLOCK) void fm_bwd_kernel(const float* __restrict__ x, int64_t xs, int B, int F,
                                                          int D, int reduce_sum, const float* __restrict__ g_out,
                                                          float* __restrict__ g_x, int64_t gxs) {
  RH_CHAIN_PRIO();
  const int lig = threadIdx.x % G;
  const int64_t b = (int64_t)blockIdx.x * (RH_BLOCK / G) + threadIdx.x / G;
  if (b >= B) return;
  const float* row = x + b * xs;
  float* grow = g_x + b * gxs;
  for (int d = lig; d < D; d += G) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += row[(int64_t)f * D + d];
    const float g = reduce_sum ? g_out[b] : g_out[b * D + d];
    for (int f = 0; f < F; ++f) grow[(int64_t)f * D + d] = g * (s - row[(int64_t)f * D + d]);
  }
}

int group_for(int D) {
  int g = 4;
  while (g < D && g < RH_WAVE) g *= 2;
  return g;
}

}  // namespace

extern "C" int rh_fm_fwd(const float* x, int64_t x_stride, int B, int F, int D, int reduce_sum, float* out,
                         void* stream) {
  RH_REQUIRE(x && out && F > 0 && D > 0 && B >= 0, RH_E_BADARG, "rh_fm_fwd: bad arguments");
  if (B == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int G = group_for(D);
  const unsigned grid = (unsigned)(((int64_t)B * G + RH_BLOCK - 1) / RH_BLOCK);
  switch (G) {
    case 4: hipLaunchKernelGGL((fm_fwd_kernel<4>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, out); break;
    case 8: hipLaunchKernelGGL((fm_fwd_kernel<8>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, out); break;
    case 16: hipLaunchKernelGGL((fm_fwd_kernel<16>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, out); break;
    case 32: hipLaunchKernelGGL((fm_fwd_kernel<32>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, out); break;
    default: hipLaunchKernelGGL((fm_fwd_kernel<64>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, out); break;
  }
  RH_LAUNCH_CHECK("rh_fm_fwd");
  return 0;
}

extern "C" int rh_fm_bwd(const float* x, int64_t x_stride, int B, int F, int D, int reduce_sum,
                         const float* g_out, float* g_x, int64_t gx_stride, void* stream) {
  RH_REQUIRE(x && g_out && g_x && F > 0 && D > 0 && B >= 0, RH_E_BADARG, "rh_fm_bwd: bad arguments");
  if (B == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int G = group_for(D);
  const unsigned grid = (unsigned)(((int64_t)B * G + RH_BLOCK - 1) / RH_BLOCK);
  switch (G) {
    case 4: hipLaunchKernelGGL((fm_bwd_kernel<4>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, g_out, g_x, gx_stride); break;
    case 8: hipLaunchKernelGGL((fm_bwd_kernel<8>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, g_out, g_x, gx_stride); break;
    case 16: hipLaunchKernelGGL((fm_bwd_kernel<16>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, g_out, g_x, gx_stride); break;
    case 32: hipLaunchKernelGGL((fm_bwd_kernel<32>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, g_out, g_x, gx_stride); break;
    default: hipLaunchKernelGGL((fm_bwd_kernel<64>), dim3(grid), dim3(RH_BLOCK), 0, s, x, x_stride, B, F, D, reduce_sum, g_out, g_x, gx_stride); break;
  }
  RH_LAUNCH_CHECK("rh_fm_bwd");
  return 0;
}
