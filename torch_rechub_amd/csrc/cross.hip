// CrossNetwork (DCN) forward / backward, one wavefront per sample, row kept in registers.
//
// Reference: CrossNetwork.forward torch_rechub/basic/layers.py:412-420
//     x0 = x;  for l: xw = w_l(x) (Linear(d,1,bias=False));  x = x0 * xw + b_l + x
//
// Roofline: HBM (arithmetic intensity < 1 flop/byte).  Forward reads x once and writes the
// result once (3432 B/sample at d = 429) instead of 3 x (GEMV + 3 elementwise) passes; the
// backward recomputes the L scalars w_l.x_l in registers, so it reads x, g and writes g_x only.
// Parameter gradients are reduced per block (registers -> LDS) into partial rows that the
// caller sums, which keeps the result deterministic (no float atomics).
#include "common.h"

namespace {

constexpr int kWavesPerBlock = RH_BLOCK / RH_WAVE;

// lane owns elements e = lane + 64*k, k < EPL (coalesced 256-byte wavefront accesses)
template <int EPL, int NL>
__global__ __launch_bounds__(RH_BLOCK) void cross_fwd_kernel(const float* __restrict__ x0, int64_t x0_stride,
                                                             const float* __restrict__ x, int64_t x_stride,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int d,
                                                             float* __restrict__ out, int64_t out_stride) {
  RH_CHAIN_PRIO();
  const int lane = threadIdx.x % RH_WAVE;
  const int wave = threadIdx.x / RH_WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  float wv[NL][EPL], bv[NL][EPL];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      wv[l][k] = e < d ? w[(int64_t)l * d + e] : 0.f;
      bv[l][k] = e < d ? bias[(int64_t)l * d + e] : 0.f;
    }
  for (int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wave; s < B; s += nwaves) {
    float a0[EPL], xv[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      a0[k] = e < d ? x0[s * x0_stride + e] : 0.f;
      xv[k] = e < d ? x[s * x_stride + e] : 0.f;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      float p = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) p = fmaf(wv[l][k], xv[k], p);
      const float sc = wave_sum(p);
#pragma unroll
      for (int k = 0; k < EPL; ++k) xv[k] = (a0[k] * sc + bv[l][k]) + xv[k];
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      if (e < d) out[s * out_stride + e] = xv[k];
    }
  }
}

template <int EPL, int NL>
__global__ __launch_bounds__(RH_BLOCK) void cross_bwd_kernel(
    const float* __restrict__ x0, int64_t x0_stride, const float* __restrict__ x, int64_t x_stride,
    const float* __restrict__ w, const float* __restrict__ bias, int B, int d,
    const float* __restrict__ g_out, int64_t g_stride, float* __restrict__ g_x0, float* __restrict__ g_x,
    int64_t gx_stride, int sum_into_gx, float* __restrict__ partials) {
  RH_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [waves][2][NL][EPL*64]
  const int lane = threadIdx.x % RH_WAVE;
  const int wave = threadIdx.x / RH_WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  float wv[NL][EPL], bv[NL][EPL];
  float acc_w[NL][EPL], acc_b[NL][EPL];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      wv[l][k] = e < d ? w[(int64_t)l * d + e] : 0.f;
      bv[l][k] = e < d ? bias[(int64_t)l * d + e] : 0.f;
      acc_w[l][k] = 0.f;
      acc_b[l][k] = 0.f;
    }
  for (int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wave; s < B; s += nwaves) {
    float a0[EPL], xs[NL][EPL], G[EPL], gx0[EPL], sc[NL];
    float xv[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      a0[k] = e < d ? x0[s * x0_stride + e] : 0.f;
      xv[k] = e < d ? x[s * x_stride + e] : 0.f;
      G[k] = e < d ? g_out[s * g_stride + e] : 0.f;
      gx0[k] = 0.f;
    }
    // recompute the forward, keeping every x_l
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      float p = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        xs[l][k] = xv[k];
        p = fmaf(wv[l][k], xv[k], p);
      }
      sc[l] = wave_sum(p);
#pragma unroll
      for (int k = 0; k < EPL; ++k) xv[k] = (a0[k] * sc[l] + bv[l][k]) + xv[k];
    }
    // x_{l+1} = x0 * s_l + b_l + x_l  with s_l = w_l . x_l
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      float p = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) p = fmaf(G[k], a0[k], p);
      const float t = wave_sum(p);  // dL/ds_l
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        acc_b[l][k] += G[k];
        acc_w[l][k] = fmaf(t, xs[l][k], acc_w[l][k]);
        gx0[k] = fmaf(G[k], sc[l], gx0[k]);
        G[k] = fmaf(t, wv[l][k], G[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      if (e < d) {
        if (sum_into_gx) {
          g_x[s * gx_stride + e] = G[k] + gx0[k];
        } else {
          g_x[s * gx_stride + e] = G[k];
          g_x0[s * gx_stride + e] = gx0[k];
        }
      }
    }
  }
  // block reduction of the parameter gradients: wave-private LDS slabs, then a fixed-order sum
  constexpr int SLAB = 2 * NL * EPL * RH_WAVE;
  float* mine = lds + (size_t)wave * SLAB;
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      mine[(0 * NL + l) * EPL * RH_WAVE + k * RH_WAVE + lane] = acc_w[l][k];
      mine[(1 * NL + l) * EPL * RH_WAVE + k * RH_WAVE + lane] = acc_b[l][k];
    }
  __syncthreads();
  float* prow = partials + (size_t)blockIdx.x * 2 * NL * d;
  for (int i = threadIdx.x; i < SLAB; i += RH_BLOCK) {
    const int e = i % (EPL * RH_WAVE);  // = k*64 + lane = element index
    const int wl = i / (EPL * RH_WAVE);  // = which*NL + l
    if (e < d) {
      float sum = lds[i];
#pragma unroll
      for (int wvi = 1; wvi < kWavesPerBlock; ++wvi) sum += lds[(size_t)wvi * SLAB + i];
      prow[(size_t)wl * d + e] = sum;
    }
  }
}

int epl_for(int d) {
  int epl = 1;
  while (epl * RH_WAVE < d) epl *= 2;
  return epl;
}

template <int EPL, int NL>
int launch_fwd(const float* x0, int64_t x0s, const float* x, int64_t xs, const float* w, const float* b,
               int B, int d, float* out, int64_t os, hipStream_t s) {
  int64_t blocks = ((int64_t)B + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL((cross_fwd_kernel<EPL, NL>), dim3((unsigned)blocks), dim3(RH_BLOCK), 0, s, x0, x0s, x,
                     xs, w, b, B, d, out, os);
  return 0;
}

template <int EPL, int NL>
int launch_bwd(const float* x0, int64_t x0s, const float* x, int64_t xs, const float* w, const float* b,
               int B, int d, const float* g, int64_t gs, float* gx0, float* gx, int64_t gxs, int sum_into,
               float* partials, int nblocks, hipStream_t s) {
  const size_t shmem = (size_t)kWavesPerBlock * 2 * NL * EPL * RH_WAVE * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && shmem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_bwd_kernel<EPL, NL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((cross_bwd_kernel<EPL, NL>), dim3((unsigned)nblocks), dim3(RH_BLOCK), shmem, s, x0, x0s,
                     x, xs, w, b, B, d, g, gs, gx0, gx, gxs, sum_into, partials);
  return 0;
}

#define RH_CROSS_DISPATCH(FN, ...)                                   \
  switch (epl * 8 + L) {                                             \
    case 1 * 8 + 1: return FN<1, 1>(__VA_ARGS__);                    \
    case 1 * 8 + 2: return FN<1, 2>(__VA_ARGS__);                    \
    case 1 * 8 + 3: return FN<1, 3>(__VA_ARGS__);                    \
    case 1 * 8 + 4: return FN<1, 4>(__VA_ARGS__);                    \
    case 2 * 8 + 1: return FN<2, 1>(__VA_ARGS__);                    \
    case 2 * 8 + 2: return FN<2, 2>(__VA_ARGS__);                    \
    case 2 * 8 + 3: return FN<2, 3>(__VA_ARGS__);                    \
    case 2 * 8 + 4: return FN<2, 4>(__VA_ARGS__);                    \
    case 4 * 8 + 1: return FN<4, 1>(__VA_ARGS__);                    \
    case 4 * 8 + 2: return FN<4, 2>(__VA_ARGS__);                    \
    case 4 * 8 + 3: return FN<4, 3>(__VA_ARGS__);                    \
    case 4 * 8 + 4: return FN<4, 4>(__VA_ARGS__);                    \
    case 8 * 8 + 1: return FN<8, 1>(__VA_ARGS__);                    \
    case 8 * 8 + 2: return FN<8, 2>(__VA_ARGS__);                    \
    case 8 * 8 + 3: return FN<8, 3>(__VA_ARGS__);                    \
    case 8 * 8 + 4: return FN<8, 4>(__VA_ARGS__);                    \
    case 16 * 8 + 1: return FN<16, 1>(__VA_ARGS__);                  \
    case 16 * 8 + 2: return FN<16, 2>(__VA_ARGS__);                  \
    case 32 * 8 + 1: return FN<32, 1>(__VA_ARGS__);                  \
    default: break;                                                  \
  }

int dispatch_fwd(int epl, int L, const float* x0, int64_t x0s, const float* x, int64_t xs, const float* w,
                 const float* b, int B, int d, float* out, int64_t os, hipStream_t s) {
  RH_CROSS_DISPATCH(launch_fwd, x0, x0s, x, xs, w, b, B, d, out, os, s)
  return RH_E_UNSUPPORTED;
}

int dispatch_bwd(int epl, int L, const float* x0, int64_t x0s, const float* x, int64_t xs, const float* w,
                 const float* b, int B, int d, const float* g, int64_t gs, float* gx0, float* gx, int64_t gxs,
                 int sum_into, float* partials, int nblocks, hipStream_t s) {
  RH_CROSS_DISPATCH(launch_bwd, x0, x0s, x, xs, w, b, B, d, g, gs, gx0, gx, gxs, sum_into, partials, nblocks, s)
  return RH_E_UNSUPPORTED;
}

}  // namespace

// max layers one call can take for feature width d (register budget: EPL*L <= 32)
extern "C" int rh_cross_max_layers(int d) {
  if (d <= 0 || d > 2048) return 0;
  const int epl = epl_for(d);
  if (epl <= 8) return 4;
  if (epl == 16) return 2;
  return 1;
}

extern "C" int rh_cross_fwd(const float* x0, int64_t x0_stride, const float* x, int64_t x_stride,
                            const float* w, const float* b, int B, int d, int L, float* out,
                            int64_t out_stride, void* stream) {
  RH_REQUIRE(x0 && x && w && b && out, RH_E_BADARG, "rh_cross_fwd: null pointer");
  RH_REQUIRE(d > 0 && d <= 2048, RH_E_UNSUPPORTED, "rh_cross_fwd: d=%d unsupported (1..2048)", d);
  RH_REQUIRE(L >= 1 && L <= rh_cross_max_layers(d), RH_E_UNSUPPORTED,
             "rh_cross_fwd: L=%d layers per call unsupported for d=%d (max %d; chain calls)", L, d,
             rh_cross_max_layers(d));
  if (B == 0) return 0;
  int rc = dispatch_fwd(epl_for(d), L, x0, x0_stride, x, x_stride, w, b, B, d, out, out_stride,
                        reinterpret_cast<hipStream_t>(stream));
  if (rc != 0) {
    rh_set_error("rh_cross_fwd: no kernel for d=%d L=%d", d, L);
    return rc;
  }
  RH_LAUNCH_CHECK("rh_cross_fwd");
  return 0;
}

extern "C" int rh_cross_bwd_nblocks(int B) {
  // every wavefront takes >= 4 samples so the (2, L, d) partial row is amortised
  int64_t blocks = ((int64_t)B + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  return (int)blocks;
}

extern "C" int rh_cross_bwd(const float* x0, int64_t x0_stride, const float* x, int64_t x_stride,
                            const float* w, const float* b, int B, int d, int L, const float* g_out,
                            int64_t g_stride, float* g_x0, float* g_x, int64_t gx_stride, int sum_into_gx,
                            float* wb_partials, void* stream) {
  RH_REQUIRE(x0 && x && w && b && g_out && g_x && wb_partials, RH_E_BADARG, "rh_cross_bwd: null pointer");
  RH_REQUIRE(sum_into_gx || g_x0, RH_E_BADARG, "rh_cross_bwd: g_x0 is null but sum_into_gx = 0");
  RH_REQUIRE(d > 0 && d <= 2048, RH_E_UNSUPPORTED, "rh_cross_bwd: d=%d unsupported (1..2048)", d);
  RH_REQUIRE(L >= 1 && L <= rh_cross_max_layers(d), RH_E_UNSUPPORTED,
             "rh_cross_bwd: L=%d layers per call unsupported for d=%d (max %d; chain calls)", L, d,
             rh_cross_max_layers(d));
  const int nblocks = rh_cross_bwd_nblocks(B);
  int rc = dispatch_bwd(epl_for(d), L, x0, x0_stride, x, x_stride, w, b, B, d, g_out, g_stride, g_x0, g_x,
                        gx_stride, sum_into_gx, wb_partials, nblocks, reinterpret_cast<hipStream_t>(stream));
  if (rc != 0) {
    rh_set_error("rh_cross_bwd: no kernel for d=%d L=%d", d, L);
    return rc;
  }
  RH_LAUNCH_CHECK("rh_cross_bwd");
  return 0;
}
