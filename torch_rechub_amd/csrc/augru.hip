// AUGRU recurrence of DIEN's interest-evolving layer, forward and backward through time, for gfx950.
//
// Reference: AUGRU_Cell.forward / AUGRU.forward, torch_rechub/models/ranking/dien.py:30-36, 60-66 -- a Python loop
// over the T steps, each step 6 matmuls + ~12 elementwise kernels (and as many again in the backward).  Per step:
//     u = sigmoid(x Wu + h Uu + bu)      r = sigmoid(x Wr + h Ur + br)      c = tanh(x Wh + r * (h Uh) + bh)
//     h' = (1 - a u) h + a u c           a = the step's attention weight (0 on padded steps: the state stands still)
// The input halves (x W + b for the 3 gates of all steps) are ONE library GEMM done by the caller (xw, (B, T, 3D));
// what is left is a recurrence with D x 3D state weights: D / 4 lanes share a sample (lane q owns state elements
// 4q .. 4q + 3 and the matching columns of every gate), the state is exchanged with wavefront shuffles once per step,
// the state weights sit in LDS, the loop over T never leaves the kernel.
// The backward recomputes the gates from the stored states, carries dh in registers and writes the pre-activation
// gradients (B, T, 3D) + the candidate's state-side gradient (B, T, D); the weight gradients are GEMMs over those.
// Bound: neither HBM (B*T*(3D + D)*4 bytes each way, 26 MB at B = 4096, T = 100, D = 16) nor MFMA (a D x 3D matvec
// per sample and step is 768 FMAs at D = 16): it is a latency chain of T dependent steps, ~3 k VALU cycles each.
#include "common.h"

namespace {

// exp through the hardware exponent and a hardware reciprocal (1 ulp each): a step is a chain of D dependent gate
// evaluations per sample, and libm's expf / tanhf (range reduction, denormal branches) were most of its latency.
static __device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
static __device__ __forceinline__ float tanhf_(float x) { return fmaf(2.f, sigmoidf_(2.f * x), -1.f); }

// The state weights sit in LDS.  They are loop invariant, so their offset goes through an empty asm once per step:
// otherwise all of them are hoisted into VGPRs for the whole loop.
static __device__ __forceinline__ int per_step() {
  int off = 0;
  asm volatile("" : "+v"(off));
  return off;
}

// Work split: G = D / 4 lanes share a sample; lane q owns the state elements d = 4q .. 4q + 3 and, of every gate,
// the matching 4 columns of U.  (a0..a3) = sum_k h[k] * U[k][col .. col + 3].  The empty asm pins the FMAs to the
// end of the block: the optimiser otherwise sinks every FMA below all reads of the unrolled nest and keeps the read
// results live (230 VGPRs at D = 8 in a first version, scratch spills from D = 16 on).
template <int D>
static __device__ __forceinline__ float4 column_block(const float* Us, int col, const float (&h)[D]) {
  // row D of the LDS copy holds the state-side bias (zeros when there is none)
  const float4 b4 = *reinterpret_cast<const float4*>(&Us[col + D * 3 * D]);
  float a0 = b4.x, a1 = b4.y, a2 = b4.z, a3 = b4.w;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(&Us[col + k * 3 * D]);
    a0 = fmaf(h[k], w.x, a0);
    a1 = fmaf(h[k], w.y, a1);
    a2 = fmaf(h[k], w.z, a2);
    a3 = fmaf(h[k], w.w, a3);
  }
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  return make_float4(a0, a1, a2, a3);
}

// Exchanges inside an aligned group of 4 lanes as DPP quad permutes (a VALU move) instead of ds_bpermute (a trip through
// the LDS crossbar): used when 4 lanes share a sample (D = 16), where a step does 16 (forward) / 34 (backward) of them.
template <int CTRL>
static __device__ __forceinline__ float quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;  // lanes (1,0,3,2) / (2,3,0,1)

// sum over the G lanes of a sample's group; every lane gets the total
template <int G>
static __device__ __forceinline__ float group_sum(float v) {
  if constexpr (G == 4) {
    v += quad_perm<kQuadXor1>(v);
    v += quad_perm<kQuadXor2>(v);
  } else {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, RH_WAVE);
  }
  return v;
}

template <int D>
static __device__ __forceinline__ void load_row(const float* p, float (&h)[D]) {
#pragma unroll
  for (int d4 = 0; d4 < D / 4; ++d4) {
    const float4 v = gload<float4>(p + 4 * d4);
    h[4 * d4 + 0] = v.x, h[4 * d4 + 1] = v.y, h[4 * d4 + 2] = v.z, h[4 * d4 + 3] = v.w;
  }
}

template <int D>
__global__ __launch_bounds__(RH_WAVE) void augru_fwd_kernel(const float* __restrict__ xw, const float* __restrict__ attn,
                                                            const float* __restrict__ U, const float* __restrict__ ub,
                                                            int B, int T, float* __restrict__ h_all) {
  constexpr int G = D / 4;            // lanes per sample
  constexpr int SPW = RH_WAVE / G;    // samples per wavefront
  __shared__ __attribute__((aligned(16))) float Us[(D + 1) * 3 * D];
  for (int i = threadIdx.x; i < D * 3 * D; i += RH_WAVE) Us[i] = U[i];
  for (int i = threadIdx.x; i < 3 * D; i += RH_WAVE) Us[D * 3 * D + i] = ub != nullptr ? ub[i] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x;
  const int q = lane % G;
  const int head = lane - q;  // first lane of the sample's group
  int64_t b = (int64_t)blockIdx.x * SPW + lane / G;
  const bool live = b < B;
  if (!live) b = B - 1;  // keep the group converged for the shuffles; nothing is stored
  const float* xb = xw + b * T * 3 * D + 4 * q;
  const float* ab = attn != nullptr ? attn + b * T : nullptr;  // null: every weight is 1 (a plain GRU update)
  float* hb = h_all + b * T * D + 4 * q;
  float h[D];
#pragma unroll
  for (int d = 0; d < D; ++d) h[d] = 0.f;
  float4 own = f4_zero();
  float4 xu = gload<float4>(xb), xr = gload<float4>(xb + D), xh = gload<float4>(xb + 2 * D);
  float a = ab != nullptr ? ab[0] : 1.f;
  for (int t = 0; t < T; ++t) {
    // the next step's inputs are requested before this step's arithmetic: a step is latency, not bandwidth
    const int tn = t + 1 < T ? t + 1 : t;
    const float4 nxu = gload<float4>(xb + (int64_t)tn * 3 * D), nxr = gload<float4>(xb + (int64_t)tn * 3 * D + D),
                 nxh = gload<float4>(xb + (int64_t)tn * 3 * D + 2 * D);
    const float na = ab != nullptr ? ab[tn] : 1.f;
    const int off = per_step() + 4 * q;
    const float4 su = column_block<D>(Us, off, h), sr = column_block<D>(Us, off + D, h),
                 sh = column_block<D>(Us, off + 2 * D, h);
    const float xus[4] = {xu.x, xu.y, xu.z, xu.w}, xrs[4] = {xr.x, xr.y, xr.z, xr.w}, xhs[4] = {xh.x, xh.y, xh.z, xh.w};
    const float sus[4] = {su.x, su.y, su.z, su.w}, srs[4] = {sr.x, sr.y, sr.z, sr.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
    float hv[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = sigmoidf_(xus[e] + sus[e]);
      const float r = sigmoidf_(xrs[e] + srs[e]);
      const float c = tanhf_(xhs[e] + r * shs[e]);
      const float g = a * u;
      hv[e] = (1.f - g) * hv[e] + g * c;
    }
    own = make_float4(hv[0], hv[1], hv[2], hv[3]);
    if (live) gstore<float4>(hb + (int64_t)t * D, own);
    // every lane of the group needs the whole new state for the next state product
    if constexpr (G == 4) {
      const float o[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[0 + e] = quad_perm<0x00>(o[e]);
        h[4 + e] = quad_perm<0x55>(o[e]);
        h[8 + e] = quad_perm<0xAA>(o[e]);
        h[12 + e] = quad_perm<0xFF>(o[e]);
      }
    } else {
#pragma unroll
      for (int p = 0; p < G; ++p) {
        h[4 * p + 0] = __shfl(own.x, head + p, RH_WAVE);
        h[4 * p + 1] = __shfl(own.y, head + p, RH_WAVE);
        h[4 * p + 2] = __shfl(own.z, head + p, RH_WAVE);
        h[4 * p + 3] = __shfl(own.w, head + p, RH_WAVE);
      }
    }
    xu = nxu, xr = nxr, xh = nxh, a = na;
  }
}

template <int D>
__global__ __launch_bounds__(RH_WAVE) void augru_bwd_kernel(const float* __restrict__ xw, const float* __restrict__ attn,
                                                            const float* __restrict__ U, const float* __restrict__ ub,
                                                            const float* __restrict__ h_all,
                                                            const float* __restrict__ g_hall, int B, int T,
                                                            float* __restrict__ d_xw, float* __restrict__ d_huh,
                                                            float* __restrict__ d_attn) {
  constexpr int G = D / 4;
  constexpr int SPW = RH_WAVE / G;
  __shared__ __attribute__((aligned(16))) float Us[(D + 1) * 3 * D];
  for (int i = threadIdx.x; i < D * 3 * D; i += RH_WAVE) Us[i] = U[i];
  for (int i = threadIdx.x; i < 3 * D; i += RH_WAVE) Us[D * 3 * D + i] = ub != nullptr ? ub[i] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x;
  const int q = lane % G;
  int64_t b = (int64_t)blockIdx.x * SPW + lane / G;
  const bool live = b < B;
  if (!live) b = B - 1;
  const float* xb = xw + b * T * 3 * D + 4 * q;
  const float* ab = attn != nullptr ? attn + b * T : nullptr;
  const float* hb = h_all + b * T * D;
  const float* gb = g_hall != nullptr ? g_hall + b * T * D + 4 * q : nullptr;
  float4 dh = f4_zero();  // gradient of the OWN four state elements
  // As in the forward, the inputs of the next step to be processed (t - 1) are requested before this step's arithmetic:
  // h_{t-2} (whole row: the state product needs all of it), the upstream gradient, the input halves and the weight.
  float hp[D];
  if (T > 1) {
    load_row<D>(hb + (int64_t)(T - 2) * D, hp);
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) hp[d] = 0.f;
  }
  float4 gv = gb != nullptr ? gload<float4>(gb + (int64_t)(T - 1) * D) : f4_zero();
  float4 xu = gload<float4>(xb + (int64_t)(T - 1) * 3 * D), xr = gload<float4>(xb + (int64_t)(T - 1) * 3 * D + D),
         xh = gload<float4>(xb + (int64_t)(T - 1) * 3 * D + 2 * D);
  float a = ab != nullptr ? ab[T - 1] : 1.f;
  for (int t = T - 1; t >= 0; --t) {
    const int tn = t > 0 ? t - 1 : 0;  // the step after this one (clamped: the last prefetch is discarded)
    float nhp[D];
    if (tn > 0) {
      load_row<D>(hb + (int64_t)(tn - 1) * D, nhp);
    } else {
#pragma unroll
      for (int d = 0; d < D; ++d) nhp[d] = 0.f;
    }
    const float4 ngv = gb != nullptr ? gload<float4>(gb + (int64_t)tn * D) : f4_zero();
    const float4 nxu = gload<float4>(xb + (int64_t)tn * 3 * D), nxr = gload<float4>(xb + (int64_t)tn * 3 * D + D),
                 nxh = gload<float4>(xb + (int64_t)tn * 3 * D + 2 * D);
    const float na = ab != nullptr ? ab[tn] : 1.f;
    dh = f4_add(dh, gv);
    const int off = per_step() + 4 * q;
    const float4 su = column_block<D>(Us, off, hp), sr = column_block<D>(Us, off + D, hp),
                 sh = column_block<D>(Us, off + 2 * D, hp);
    const float xus[4] = {xu.x, xu.y, xu.z, xu.w}, xrs[4] = {xr.x, xr.y, xr.z, xr.w}, xhs[4] = {xh.x, xh.y, xh.z, xh.w};
    const float sus[4] = {su.x, su.y, su.z, su.w}, srs[4] = {sr.x, sr.y, sr.z, sr.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
    const float dhs[4] = {dh.x, dh.y, dh.z, dh.w};
    float o_u[4], o_r[4], o_c[4], o_q[4], keep[4];
    float da = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float hpe = hp[0];  // hp[4 q + e] without dynamic register indexing
#pragma unroll
      for (int p = 0; p < G; ++p) hpe = (p == q) ? hp[4 * p + e] : hpe;
      const float u = sigmoidf_(xus[e] + sus[e]);
      const float r = sigmoidf_(xrs[e] + srs[e]);
      const float c = tanhf_(xhs[e] + r * shs[e]);
      const float g = a * u;
      const float dg = dhs[e] * (c - hpe);
      const float dc = dhs[e] * g;
      keep[e] = dhs[e] * (1.f - g);  // the direct path to h_{t-1}; the paths through U are added below
      da = fmaf(dg, u, da);
      o_u[e] = dg * a * u * (1.f - u);
      o_c[e] = dc * (1.f - c * c);
      o_r[e] = o_c[e] * shs[e] * r * (1.f - r);
      o_q[e] = o_c[e] * r;
    }
    da = group_sum<G>(da);
    if (live) {
      float* dx = d_xw + (b * T + t) * 3 * D + 4 * q;
      gstore<float4>(dx, make_float4(o_u[0], o_u[1], o_u[2], o_u[3]));
      gstore<float4>(dx + D, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
      gstore<float4>(dx + 2 * D, make_float4(o_c[0], o_c[1], o_c[2], o_c[3]));
      gstore<float4>(d_huh + (b * T + t) * D + 4 * q, make_float4(o_q[0], o_q[1], o_q[2], o_q[3]));
      if (q == 0 && d_attn != nullptr) d_attn[b * T + t] = da;
    }
    // dh_{t-1}[k] += sum_j v[j] U[k][j], v = [d pre_u | d pre_r | d (h Uh)]: this lane holds 12 of the 3D entries of v,
    // so it forms its share of the sum for EVERY k, and the shares are added across the group (xor butterfly)
    float part[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float4 wu = *reinterpret_cast<const float4*>(&Us[off + k * 3 * D]);
      const float4 wr = *reinterpret_cast<const float4*>(&Us[off + D + k * 3 * D]);
      const float4 wq = *reinterpret_cast<const float4*>(&Us[off + 2 * D + k * 3 * D]);
      float s = o_u[0] * wu.x;
      s = fmaf(o_u[1], wu.y, s), s = fmaf(o_u[2], wu.z, s), s = fmaf(o_u[3], wu.w, s);
      s = fmaf(o_r[0], wr.x, s), s = fmaf(o_r[1], wr.y, s), s = fmaf(o_r[2], wr.z, s), s = fmaf(o_r[3], wr.w, s);
      s = fmaf(o_q[0], wq.x, s), s = fmaf(o_q[1], wq.y, s), s = fmaf(o_q[2], wq.z, s), s = fmaf(o_q[3], wq.w, s);
      asm volatile("" : "+v"(s));
      part[k] = s;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) part[k] = group_sum<G>(part[k]);
    float nd[4] = {keep[0], keep[1], keep[2], keep[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mine = part[e];
#pragma unroll
      for (int p = 0; p < G; ++p) mine = (p == q) ? part[4 * p + e] : mine;
      nd[e] += mine;
    }
    dh = make_float4(nd[0], nd[1], nd[2], nd[3]);
#pragma unroll
    for (int d = 0; d < D; ++d) hp[d] = nhp[d];
    gv = ngv, xu = nxu, xr = nxr, xh = nxh, a = na;
  }
}

}  // namespace

extern "C" int rh_augru_max_dim(void) { return 32; }

extern "C" int rh_augru_fwd(const float* xw, const float* attn, const float* U, const float* state_bias, int B, int T,
                            int D, float* h_all, void* stream) {
  RH_REQUIRE(B >= 0 && T >= 1, RH_E_BADARG, "rh_augru_fwd: bad shape B=%d T=%d", B, T);
  RH_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32, RH_E_UNSUPPORTED, "rh_augru_fwd: D=%d (4, 8, 16, 32)", D);
  if (B == 0) return 0;
  RH_REQUIRE(xw && U && h_all, RH_E_BADARG, "rh_augru_fwd: null pointer");
  const int spw = RH_WAVE / (D / 4);  // samples per wavefront (= per workgroup)
  const dim3 grid((unsigned)((B + spw - 1) / spw)), block(RH_WAVE);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 4: hipLaunchKernelGGL(augru_fwd_kernel<4>, grid, block, 0, s, xw, attn, U, state_bias, B, T, h_all); break;
    case 8: hipLaunchKernelGGL(augru_fwd_kernel<8>, grid, block, 0, s, xw, attn, U, state_bias, B, T, h_all); break;
    case 16: hipLaunchKernelGGL(augru_fwd_kernel<16>, grid, block, 0, s, xw, attn, U, state_bias, B, T, h_all); break;
    default: hipLaunchKernelGGL(augru_fwd_kernel<32>, grid, block, 0, s, xw, attn, U, state_bias, B, T, h_all); break;
  }
  RH_LAUNCH_CHECK("rh_augru_fwd");
  return 0;
}

extern "C" int rh_augru_bwd(const float* xw, const float* attn, const float* U, const float* state_bias,
                            const float* h_all, const float* g_hall, int B, int T, int D, float* d_xw, float* d_huh,
                            float* d_attn, void* stream) {
  RH_REQUIRE(B >= 0 && T >= 1, RH_E_BADARG, "rh_augru_bwd: bad shape B=%d T=%d", B, T);
  RH_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32, RH_E_UNSUPPORTED, "rh_augru_bwd: D=%d (4, 8, 16, 32)", D);
  if (B == 0) return 0;
  RH_REQUIRE(xw && U && h_all && d_xw && d_huh, RH_E_BADARG, "rh_augru_bwd: null pointer");
  RH_REQUIRE(attn == nullptr || d_attn != nullptr, RH_E_BADARG, "rh_augru_bwd: attn without d_attn");
  const int spw = RH_WAVE / (D / 4);  // samples per wavefront (= per workgroup)
  const dim3 grid((unsigned)((B + spw - 1) / spw)), block(RH_WAVE);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 4: hipLaunchKernelGGL(augru_bwd_kernel<4>, grid, block, 0, s, xw, attn, U, state_bias, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    case 8: hipLaunchKernelGGL(augru_bwd_kernel<8>, grid, block, 0, s, xw, attn, U, state_bias, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    case 16: hipLaunchKernelGGL(augru_bwd_kernel<16>, grid, block, 0, s, xw, attn, U, state_bias, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
    default: hipLaunchKernelGGL(augru_bwd_kernel<32>, grid, block, 0, s, xw, attn, U, state_bias, h_all, g_hall, B, T, d_xw, d_huh, d_attn); break;
  }
  RH_LAUNCH_CHECK("rh_augru_bwd");
  return 0;
}
