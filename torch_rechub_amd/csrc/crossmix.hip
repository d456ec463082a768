// DCN-v2 cross layers: the memory-bound epilogues around the library GEMMs.
//
// Reference:
//   CrossNetV2.forward   torch_rechub/basic/layers.py:440-444   x <- x0 * (W_l x) + b_l + x
//   CrossNetMix.forward  torch_rechub/basic/layers.py:470-506   per layer, experts e = 1..E:
//        o_e = x0 * (U_e tanh(C_e tanh(V_e^T x_l)) + bias_l);  x_{l+1} = sum_e softmax(g)_e o_e + x_l
// The contractions (W_l x, V^T x, C v, U v) are true dense GEMMs and stay on hipBLASLt; what is left around them is
// pure streaming work that the reference spreads over ~6 (V2) / ~15 (Mix) elementwise kernels per layer and direction.
// One wavefront per sample (d <= 2048), everything for that sample in registers; Hadamard + bias + residual (V2) and
// bias + Hadamard + gate-weighted expert mix + residual (Mix) in one pass, likewise the backward (incl. the per-expert
// gate gradient, a wavefront reduction).  Roofline: HBM.
#include "common.h"

namespace {

constexpr int kWaves = RH_BLOCK / RH_WAVE;
constexpr int kMaxExperts = 16;

unsigned wave_grid(int B) {
  int64_t g = ((int64_t)B + kWaves - 1) / kWaves;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// BWD = false: out = x0*y + b + x.   BWD = true: g_x0 = g*y, g_y = g*x0 (g_x = g and g_b = colsum(g) are the caller's)
template <bool BWD>
__global__ __launch_bounds__(RH_BLOCK) void cross_v2_kernel(const float* __restrict__ x0, const float* __restrict__ y,
                                                            const float* __restrict__ b, const float* __restrict__ x,
                                                            const float* __restrict__ g, int64_t n, int d,
                                                            float* __restrict__ o1, float* __restrict__ o2) {
  RH_CHAIN_PRIO();
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * RH_BLOCK) {
    if (!BWD) {
      o1[i] = fmaf(x0[i], y[i], b[i % d]) + x[i];
    } else {
      const float gi = g[i];
      o1[i] = gi * y[i];
      o2[i] = gi * x0[i];
    }
  }
}

struct MixArgs {
  const float* x0;    // (B, d)
  const float* xl;    // (B, d)
  const float* uv;    // (E, B, d)
  const float* gate;  // (B, E) softmax weights
  const float* bias;  // (d,)
  const float* g;     // (B, d) upstream (backward)
  float* out;         // fwd: (B, d) ; bwd: g_x0 (B, d)
  float* g_uv;        // (E, B, d)
  float* g_gate;      // (B, E)
  int B, d, E;
  float* g_bias_partial;  // (gridDim.x, d) per-block sums of g * x0 * sum_e gate_e, or null (backward)
};

template <int EPL, bool BWD>
__global__ __launch_bounds__(RH_BLOCK) void cross_mix_kernel(const MixArgs a) {
  RH_CHAIN_PRIO();
  const int lane = threadIdx.x % RH_WAVE, wave = threadIdx.x / RH_WAVE;
  const int64_t nw = (int64_t)gridDim.x * kWaves;
  const int d = a.d, E = a.E;
  const int64_t plane = (int64_t)a.B * d;
  float bv[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = lane + RH_WAVE * k;
    bv[k] = e < d ? a.bias[e] : 0.f;
  }
  float gb[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) gb[k] = 0.f;
  for (int64_t s = (int64_t)blockIdx.x * kWaves + wave; s < a.B; s += nw) {
    float a0[EPL], acc[EPL], gv[EPL];
    float gsum = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      a0[k] = e < d ? a.x0[s * d + e] : 0.f;
      acc[k] = 0.f;
      gv[k] = (BWD && e < d) ? a.g[s * d + e] : 0.f;
    }
    for (int ex = 0; ex < E; ++ex) {
      const float gt = a.gate[s * E + ex];
      gsum += gt;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + RH_WAVE * k;
        if (e < d) {
          const float t = a.uv[ex * plane + s * d + e] + bv[k];
          if (!BWD) {
            acc[k] = fmaf(gt, a0[k] * t, acc[k]);
          } else {
            acc[k] = fmaf(gt, t, acc[k]);                       // sum_e gate_e (uv_e + bias)
            dot = fmaf(gv[k] * a0[k], t, dot);                  // d/d gate_e
            a.g_uv[ex * plane + s * d + e] = gv[k] * gt * a0[k];
          }
        }
      }
      if (BWD) {
        dot = wave_sum(dot);
        if (lane == 0) a.g_gate[s * E + ex] = dot;
      }
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      if (e < d) a.out[s * d + e] = BWD ? gv[k] * acc[k] : acc[k] + a.xl[s * d + e];
      if (BWD) gb[k] = fmaf(gv[k] * a0[k], gsum, gb[k]);  // d/d bias = g x0 sum_e gate_e
    }
  }
  if (BWD && a.g_bias_partial != nullptr) {  // the block's 4 wavefronts in fixed order -> one partial row per block
    __shared__ float red[kWaves][RH_WAVE * EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) red[wave][lane + RH_WAVE * k] = gb[k];
    __syncthreads();
    for (int e = threadIdx.x; e < d; e += RH_BLOCK) {
      float v = red[0][e];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) v += red[w][e];
      a.g_bias_partial[(int64_t)blockIdx.x * d + e] = v;
    }
  }
}

int mix_epl(int d) {
  int e = 1;
  while (e * RH_WAVE < d) e *= 2;
  return e;
}

template <bool BWD>
int mix_dispatch(const MixArgs& a, hipStream_t s) {
  const unsigned grid = wave_grid(a.B);
#define RH_MIX(EP) hipLaunchKernelGGL((cross_mix_kernel<EP, BWD>), dim3(grid), dim3(RH_BLOCK), 0, s, a)
  switch (mix_epl(a.d)) {
    case 1: RH_MIX(1); break;
    case 2: RH_MIX(2); break;
    case 4: RH_MIX(4); break;
    case 8: RH_MIX(8); break;
    case 16: RH_MIX(16); break;
    case 32: RH_MIX(32); break;
    default: return RH_E_UNSUPPORTED;
  }
#undef RH_MIX
  return 0;
}

}  // namespace

extern "C" int rh_cross_v2_epilogue_fwd(const float* x0, const float* y, const float* b, const float* x, int B, int d,
                                        float* out, void* stream) {
  RH_REQUIRE(x0 && y && b && x && out && B >= 0 && d >= 1, RH_E_BADARG, "rh_cross_v2_epilogue_fwd: bad arguments");
  if (B == 0) return 0;
  const int64_t n = (int64_t)B * d;
  int64_t grid = (n + RH_BLOCK * 4 - 1) / (RH_BLOCK * 4);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL((cross_v2_kernel<false>), dim3((unsigned)grid), dim3(RH_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), x0, y, b, x, nullptr, n, d, out, nullptr);
  RH_LAUNCH_CHECK("rh_cross_v2_epilogue_fwd");
  return 0;
}

extern "C" int rh_cross_v2_epilogue_bwd(const float* x0, const float* y, const float* g, int B, int d, float* g_x0,
                                        float* g_y, void* stream) {
  RH_REQUIRE(x0 && y && g && g_x0 && g_y && B >= 0 && d >= 1, RH_E_BADARG, "rh_cross_v2_epilogue_bwd: bad arguments");
  if (B == 0) return 0;
  const int64_t n = (int64_t)B * d;
  int64_t grid = (n + RH_BLOCK * 4 - 1) / (RH_BLOCK * 4);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL((cross_v2_kernel<true>), dim3((unsigned)grid), dim3(RH_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), x0, y, nullptr, nullptr, g, n, d, g_x0, g_y);
  RH_LAUNCH_CHECK("rh_cross_v2_epilogue_bwd");
  return 0;
}

extern "C" int rh_cross_mix_epilogue_fwd(const float* x0, const float* xl, const float* uv, const float* gate,
                                         const float* bias, int B, int d, int E, float* out, void* stream) {
  RH_REQUIRE(x0 && xl && uv && gate && bias && out, RH_E_BADARG, "rh_cross_mix_epilogue_fwd: null pointer");
  RH_REQUIRE(d >= 1 && d <= 2048 && E >= 1 && E <= kMaxExperts && B >= 0, RH_E_UNSUPPORTED,
             "rh_cross_mix_epilogue_fwd: d=%d E=%d unsupported (d <= 2048, E <= %d)", d, E, kMaxExperts);
  if (B == 0) return 0;
  MixArgs a{x0, xl, uv, gate, bias, nullptr, out, nullptr, nullptr, B, d, E, nullptr};
  int rc = mix_dispatch<false>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_cross_mix_epilogue_fwd");
  return 0;
}

extern "C" int rh_cross_mix_nblocks(int B) { return (int)wave_grid(B); }

static int mix_bwd_impl(const float* x0, const float* uv, const float* gate, const float* bias, const float* g, int B, int d,
                        int E, float* g_x0, float* g_uv, float* g_gate, float* g_bias_partial, void* stream);

extern "C" int rh_cross_mix_epilogue_bwd(const float* x0, const float* uv, const float* gate, const float* bias,
                                         const float* g, int B, int d, int E, float* g_x0, float* g_uv, float* g_gate,
                                         void* stream) {
  return mix_bwd_impl(x0, uv, gate, bias, g, B, d, E, g_x0, g_uv, g_gate, nullptr, stream);
}

// the same + the bias gradient as rh_cross_mix_nblocks(B) x d per-block partial rows (the caller / rh_pack_grads sums them)
extern "C" int rh_cross_mix_epilogue_bwd_b(const float* x0, const float* uv, const float* gate, const float* bias,
                                           const float* g, int B, int d, int E, float* g_x0, float* g_uv, float* g_gate,
                                           float* g_bias_partial, void* stream) {
  RH_REQUIRE(g_bias_partial != nullptr, RH_E_BADARG, "rh_cross_mix_epilogue_bwd_b: null partial buffer");
  return mix_bwd_impl(x0, uv, gate, bias, g, B, d, E, g_x0, g_uv, g_gate, g_bias_partial, stream);
}

static int mix_bwd_impl(const float* x0, const float* uv, const float* gate, const float* bias, const float* g, int B, int d,
                        int E, float* g_x0, float* g_uv, float* g_gate, float* g_bias_partial, void* stream) {
  RH_REQUIRE(x0 && uv && gate && bias && g && g_x0 && g_uv && g_gate, RH_E_BADARG,
             "rh_cross_mix_epilogue_bwd: null pointer");
  RH_REQUIRE(d >= 1 && d <= 2048 && E >= 1 && E <= kMaxExperts && B >= 0, RH_E_UNSUPPORTED,
             "rh_cross_mix_epilogue_bwd: d=%d E=%d unsupported (d <= 2048, E <= %d)", d, E, kMaxExperts);
  if (B == 0) return 0;
  MixArgs a{x0, nullptr, uv, gate, bias, g, g_x0, g_uv, g_gate, B, d, E, g_bias_partial};
  int rc = mix_dispatch<true>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_cross_mix_epilogue_bwd");
  return 0;
}
