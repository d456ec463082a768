// AUGRU recurrence of DIEN's interest-evolving layer, forward and backward through time, for gfx950.
//
// Reference: AUGRU_Cell.forward / AUGRU.forward, torch_rechub/models/ranking/dien.py:30-36, 60-66 -- a Python loop
// over the T steps, each step 6 matmuls + ~12 elementwise kernels (and as many again in the backward).  Per step:
//     u = sigmoid(x Wu + h Uu + bu)      r = sigmoid(x Wr + h Ur + br)      c = tanh(x Wh + r * (h Uh) + bh)
//     h' = (1 - a u) h + a u c           a = the step's attention weight (0 on padded steps: the state stands still)
// The input halves (x W + b for the 3 gates of all steps) are ONE library GEMM done by the caller (xw, (B, T, 3D));
// what is left is a recurrence with D x 3D state weights: one LANE per sample keeps h in registers, the state weights
// sit in LDS (broadcast reads), the loop over T never leaves the kernel.
// The backward recomputes the gates from the stored states, carries dh in registers and writes the pre-activation
// gradients (B, T, 3D) + the candidate's state-side gradient (B, T, D); the weight gradients are GEMMs over those.
// Bound: neither HBM (B*T*(3D + D)*4 bytes each way, 26 MB at B = 4096, T = 100, D = 16) nor MFMA (a D x 3D matvec
// per sample and step is 768 FMAs at D = 16): it is a latency chain of T dependent steps, ~3 k VALU cycles each.
#include "common.h"

namespace {

static __device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// The state weights sit in LDS and every read of them is a wavefront broadcast.  They are loop invariant, so their
// offset goes through an empty asm once per step: otherwise all D * 3D values are hoisted into VGPRs for the whole loop.
static __device__ __forceinline__ int per_step() {
  int off = 0;
  asm volatile("" : "+v"(off));
  return off;
}

// acc[j] = sum_k h[k] * U[k][j], U (D, 3D) row-major in LDS.  Four output columns at a time; the empty asm at the end
// of a block pins its FMAs there (the optimiser otherwise sinks every FMA below all D * 3D / 4 reads of the unrolled
// nest and keeps the read results live: 230 VGPRs at D = 8, scratch spills from D = 16 on).
template <int D>
static __device__ __forceinline__ void state_product(const float* Us, int off, const float (&h)[D], float (&acc)[3 * D]) {
#pragma unroll
  for (int j4 = 0; j4 < 3 * D / 4; ++j4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float4 w = *reinterpret_cast<const float4*>(&Us[off + k * 3 * D + 4 * j4]);
      a0 = fmaf(h[k], w.x, a0);
      a1 = fmaf(h[k], w.y, a1);
      a2 = fmaf(h[k], w.z, a2);
      a3 = fmaf(h[k], w.w, a3);
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    acc[4 * j4 + 0] = a0, acc[4 * j4 + 1] = a1, acc[4 * j4 + 2] = a2, acc[4 * j4 + 3] = a3;
  }
}

template <int D>
__global__ __launch_bounds__(RH_WAVE) void augru_fwd_kernel(const float* __restrict__ xw, const float* __restrict__ attn,
                                                            const float* __restrict__ U, int B, int T,
                                                            float* __restrict__ h_all) {
  __shared__ __attribute__((aligned(16))) float Us[D * 3 * D];
  for (int i = threadIdx.x; i < D * 3 * D; i += RH_WAVE) Us[i] = U[i];
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * RH_WAVE + threadIdx.x;
  if (b >= B) return;
  const float* xb = xw + b * T * 3 * D;
  const float* ab = attn + b * T;
  float* hb = h_all + b * T * D;
  float h[D];
#pragma unroll
  for (int d = 0; d < D; ++d) h[d] = 0.f;
  for (int t = 0; t < T; ++t) {
    float acc[3 * D];
    state_product<D>(Us, per_step(), h, acc);
    const float a = ab[t];
    const float* x = xb + (int64_t)t * 3 * D;
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
      const float4 xu = gload<float4>(x + 4 * d4), xr = gload<float4>(x + D + 4 * d4), xh = gload<float4>(x + 2 * D + 4 * d4);
      const float xus[4] = {xu.x, xu.y, xu.z, xu.w}, xrs[4] = {xr.x, xr.y, xr.z, xr.w}, xhs[4] = {xh.x, xh.y, xh.z, xh.w};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 4 * d4 + e;
        const float u = sigmoidf_(xus[e] + acc[d]);
        const float r = sigmoidf_(xrs[e] + acc[D + d]);
        const float c = tanhf(xhs[e] + r * acc[2 * D + d]);
        const float g = a * u;
        h[d] = (1.f - g) * h[d] + g * c;
        out[e] = h[d];
      }
      gstore<float4>(hb + (int64_t)t * D + 4 * d4, make_float4(out[0], out[1], out[2], out[3]));
    }
  }
}

template <int D>
__global__ __launch_bounds__(RH_WAVE) void augru_bwd_kernel(const float* __restrict__ xw, const float* __restrict__ attn,
                                                            const float* __restrict__ U, const float* __restrict__ h_all,
                                                            const float* __restrict__ g_hall, int B, int T,
                                                            float* __restrict__ d_xw, float* __restrict__ d_huh,
                                                            float* __restrict__ d_attn) {
  __shared__ __attribute__((aligned(16))) float Us[D * 3 * D];
  for (int i = threadIdx.x; i < D * 3 * D; i += RH_WAVE) Us[i] = U[i];
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * RH_WAVE + threadIdx.x;
  if (b >= B) return;
  const float* xb = xw + b * T * 3 * D;
  const float* ab = attn + b * T;
  const float* hb = h_all + b * T * D;
  const float* gb = g_hall != nullptr ? g_hall + b * T * D : nullptr;
  float dh[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dh[d] = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    float hp[D];
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
      float4 v = f4_zero();
      if (t > 0) v = gload<float4>(hb + (int64_t)(t - 1) * D + 4 * d4);
      hp[4 * d4 + 0] = v.x, hp[4 * d4 + 1] = v.y, hp[4 * d4 + 2] = v.z, hp[4 * d4 + 3] = v.w;
      if (gb != nullptr) {
        const float4 gv = gload<float4>(gb + (int64_t)t * D + 4 * d4);
        dh[4 * d4 + 0] += gv.x, dh[4 * d4 + 1] += gv.y, dh[4 * d4 + 2] += gv.z, dh[4 * d4 + 3] += gv.w;
      }
    }
    float acc[3 * D];
    const int off = per_step();
    state_product<D>(Us, off, hp, acc);
    const float a = ab[t];
    const float* x = xb + (int64_t)t * 3 * D;
    float* dx = d_xw + (b * T + t) * 3 * D;
    float* dq = d_huh + (b * T + t) * D;
    float da = 0.f;
    // the three gradient blocks that flow back through U overwrite acc in place: [d pre_u | d pre_r | d (h Uh)]
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
      const float4 xu = gload<float4>(x + 4 * d4), xr = gload<float4>(x + D + 4 * d4), xh = gload<float4>(x + 2 * D + 4 * d4);
      const float xus[4] = {xu.x, xu.y, xu.z, xu.w}, xrs[4] = {xr.x, xr.y, xr.z, xr.w}, xhs[4] = {xh.x, xh.y, xh.z, xh.w};
      float o_u[4], o_r[4], o_c[4], o_q[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 4 * d4 + e;
        const float u = sigmoidf_(xus[e] + acc[d]);
        const float r = sigmoidf_(xrs[e] + acc[D + d]);
        const float q = acc[2 * D + d];  // (h Uh)_d
        const float c = tanhf(xhs[e] + r * q);
        const float g = a * u;
        const float dg = dh[d] * (c - hp[d]);
        const float dc = dh[d] * g;
        dh[d] = dh[d] * (1.f - g);  // the direct path to h_{t-1}; the paths through U are added below
        da = fmaf(dg, u, da);
        const float dpu = dg * a * u * (1.f - u);
        const float dpc = dc * (1.f - c * c);
        const float dpr = dpc * q * r * (1.f - r);
        const float dqh = dpc * r;
        o_u[e] = dpu, o_r[e] = dpr, o_c[e] = dpc, o_q[e] = dqh;
        acc[d] = dpu, acc[D + d] = dpr, acc[2 * D + d] = dqh;
      }
      gstore<float4>(dx + 4 * d4, make_float4(o_u[0], o_u[1], o_u[2], o_u[3]));
      gstore<float4>(dx + D + 4 * d4, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
      gstore<float4>(dx + 2 * D + 4 * d4, make_float4(o_c[0], o_c[1], o_c[2], o_c[3]));
      gstore<float4>(dq + 4 * d4, make_float4(o_q[0], o_q[1], o_q[2], o_q[3]));
    }
    d_attn[b * T + t] = da;
    // dh_{t-1}[k] += sum_j acc[j] * U[k][j]
#pragma unroll
    for (int k = 0; k < D; ++k) {
      float s = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 3 * D / 4; ++j4) {
        const float4 w = *reinterpret_cast<const float4*>(&Us[off + k * 3 * D + 4 * j4]);
        s = fmaf(acc[4 * j4 + 0], w.x, s);
        s = fmaf(acc[4 * j4 + 1], w.y, s);
        s = fmaf(acc[4 * j4 + 2], w.z, s);
        s = fmaf(acc[4 * j4 + 3], w.w, s);
      }
      asm volatile("" : "+v"(s));
      dh[k] += s;
    }
  }
}

}  // namespace

extern "C" int rh_augru_max_dim(void) { return 32; }

extern "C" int rh_augru_fwd(const float* xw, const float* attn, const float* U, int B, int T, int D, float* h_all,
                            void* stream) {
  RH_REQUIRE(B >= 0 && T >= 1, RH_E_BADARG, "rh_augru_fwd: bad shape B=%d T=%d", B, T);
  RH_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32, RH_E_UNSUPPORTED, "rh_augru_fwd: D=%d (4, 8, 16, 32)", D);
  if (B == 0) return 0;
  RH_REQUIRE(xw && attn && U && h_all, RH_E_BADARG, "rh_augru_fwd: null pointer");
  const dim3 grid((unsigned)((B + RH_WAVE - 1) / RH_WAVE)), block(RH_WAVE);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 4: hipLaunchKernelGGL(augru_fwd_kernel<4>, grid, block, 0, s, xw, attn, U, B, T, h_all); break;
    case 8: hipLaunchKernelGGL(augru_fwd_kernel<8>, grid, block, 0, s, xw, attn, U, B, T, h_all); break;
    case 16: hipLaunchKernelGGL(augru_fwd_kernel<16>, grid, block, 0, s, xw, attn, U, B, T, h_all); break;
    default: hipLaunchKernelGGL(augru_fwd_kernel<32>, grid, block, 0, s, xw, attn, U, B, T, h_all); break;
  }
  RH_LAUNCH_CHECK("rh_augru_fwd");
  return 0;
}

extern "C" int rh_augru_bwd(const float* xw, const float* attn, const float* U, const float* h_all, const float* g_hall,
                            int B, int T, int D, float* d_xw, float* d_huh, float* d_attn, void* stream) {
  RH_REQUIRE(B >= 0 && T >= 1, RH_E_BADARG, "rh_augru_bwd: bad shape B=%d T=%d", B, T);
  RH_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32, RH_E_UNSUPPORTED, "rh_augru_bwd: D=%d (4, 8, 16, 32)", D);
  if (B == 0) return 0;
  RH_REQUIRE(xw && attn && U && h_all && d_xw && d_huh && d_attn, RH_E_BADARG, "rh_augru_bwd: null pointer");
  const dim3 grid((unsigned)((B + RH_WAVE - 1) / RH_WAVE)), block(RH_WAVE);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 4: hipLaunchKernelGGL(augru_bwd_kernel<4>, grid, block, 0, s, xw, attn, U, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    case 8: hipLaunchKernelGGL(augru_bwd_kernel<8>, grid, block, 0, s, xw, attn, U, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    case 16: hipLaunchKernelGGL(augru_bwd_kernel<16>, grid, block, 0, s, xw, attn, U, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    default: hipLaunchKernelGGL(augru_bwd_kernel<32>, grid, block, 0, s, xw, attn, U, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
  }
  RH_LAUNCH_CHECK("rh_augru_bwd");
  return 0;
}
