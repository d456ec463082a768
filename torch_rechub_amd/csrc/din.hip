// DIN hot spots: Dice activation and the two memory-bound ends of the ActivationUnit.
//
// Reference:
//   Dice.forward            torch_rechub/basic/activation.py:15-25   (10 elementwise kernels per call; on the (B*L, 256)
//                           attention tensors these are 31 % + 10 % + 7 % of the reference's DIN step)
//       avg = x.mean(1); var = sum_1((x-avg)^2 + eps); ps = sigmoid((x-avg)/sqrt(var)); out = ps*x + (1-ps)*alpha*x
//       (statistics ACROSS NEURONS per row and a SUM, not the paper's batch statistic: SURVEY Q5)
//   ActivationUnit.forward  torch_rechub/models/ranking/din.py:77-92
//       att_input = cat[t, h, t-h, t*h] over (B, L, 4D)   and   output = (att_weight.unsqueeze(-1) * history).sum(1)
// Roofline: HBM; one pass over the data per kernel (Dice: read x, write out; backward: read x, g, write gx).
// With a BatchNorm1d in front the normalisation is folded into the same passes (see dice_kernel).
#include "common.h"

namespace {

constexpr int kWaves = RH_BLOCK / RH_WAVE;

// One wavefront per row; lane owns elements lane + 64*k (coalesced).  EPL = ceil(C / 64) <= 32.
//
// With a BatchNorm1d in front (MLP of DIN's ActivationUnit: Linear -> BatchNorm1d -> Dice, layers.py:281-287) the
// normalisation is folded in: the kernels read the PRE-BatchNorm activations h and form x = h * scale[c] + shift[c]
// on the fly (scale = gamma * rstd, shift = beta - mean * scale, written by the statistics launch of csrc/mlp.hip), so
// the normalised tensor is never materialised: forward = statistics pass + ONE pass (read h, write out) instead of
// statistics + normalise + Dice; backward = two row passes over (h, g):
//   MODE 2 forms g_x = dL/dx of Dice in registers and accumulates, per lane-owned column, sum g_x and sum g_x * xhat
//          (what BatchNorm's backward needs) -> per-block partial rows (blocks, 2, C), summed by mlp.hip's finalize;
//   MODE 3 recomputes g_x and writes dh = gamma * rstd * (g_x - mean_b(g_x) - xhat * mean_b(g_x * xhat)).
// MODE 0 forward, MODE 1 plain Dice backward (no BatchNorm in front).
struct DiceArgs {
  const float* x;          // (N, C): Dice input, or the pre-BatchNorm activations h when scale != null
  const float* g;          // backward: gradient of the Dice output
  const float* alpha;
  float eps;
  int64_t N;
  int C;
  float* out;              // forward: Dice output; MODE 1: g_x; MODE 3: dh
  float* alpha_partial;    // (blocks,) partial sums of dL/dalpha (MODE 1 and 2)
  const float* scale;      // (C,) BatchNorm folded affine, or null
  const float* shift;
  const float* stat;       // MODE 2/3: (>= 4, C) mean, rstd, sum g_x, sum g_x * xhat
  const float* gamma;
  float* col_partial;      // MODE 2: (blocks, 2, C)
  // HEAD: the Dice output feeds a Linear(C -> 1) (last layer of the ActivationUnit's MLP, layers.py:288) and nothing else.
  // forward: out is (N,) = Dice(x) . w + b, the Dice output itself is never written; backward: g is (N,) = dL/d out and the
  // gradient of the Dice output is the rank-1 g[r] * w[c], formed in registers; MODE 2 also accumulates dL/dw per column.
  const float* head_w;     // (C,)
  const float* head_b;     // (1,) or null
  float* head_partial;     // MODE 2: (blocks, C) partial dL/dw;  alpha_partial is (2, blocks): d alpha, then dL/db
};

// FULL: C == 64 * EPL, no column guards (the widths of the ActivationUnit's MLP are multiples of 64).
template <int EPL, int MODE, bool HEAD, bool FULL>
__global__ __launch_bounds__(RH_BLOCK) void dice_kernel(const DiceArgs a) {
  RH_CHAIN_PRIO();
  __shared__ float red[kWaves];
  extern __shared__ float colred[];  // MODE 2: kWaves * (2 + HEAD) * EPL * 64 floats
  constexpr int NCS = HEAD ? 3 : 2;
  const int lane = threadIdx.x % RH_WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / RH_WAVE);  // rows and their guards are wavefront-uniform
  const int64_t nw = (int64_t)gridDim.x * kWaves;
  const int C = a.C;
  const float alpha = a.alpha[0];
  const float invC = 1.f / (float)C;
  const bool bn = a.scale != nullptr;
  float sc[EPL], sh[EPL];                       // folded BatchNorm affine of the lane's columns
  float mu[EPL], rsd[EPL], gm[EPL], sg[EPL], sgx[EPL];  // MODE 2 / 3
  float cs1[EPL], cs2[EPL];                     // MODE 2: column sums over this wavefront's rows
  float hw[EPL], cs3[EPL];                      // HEAD: the lane's weights of the Linear(C -> 1); MODE 2: its dL/dw sums
  float acc_b = 0.f;                            // HEAD, MODE 2: sum of g over this wavefront's rows (same in every lane)
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = lane + RH_WAVE * k;
    const bool ok = (FULL || e < C);
    sc[k] = (bn && ok) ? a.scale[e] : 1.f;
    sh[k] = (bn && ok) ? a.shift[e] : 0.f;
    cs1[k] = cs2[k] = cs3[k] = 0.f;
    hw[k] = (HEAD && ok) ? a.head_w[e] : 0.f;
    if (MODE >= 2) {
      mu[k] = ok ? a.stat[e] : 0.f;
      rsd[k] = ok ? a.stat[C + e] : 0.f;
      gm[k] = ok ? a.gamma[e] : 0.f;
      if (MODE == 3) {
        const float inv_n = 1.f / (float)a.N;
        sg[k] = ok ? a.stat[2 * C + e] * inv_n : 0.f;
        sgx[k] = ok ? a.stat[3 * C + e] * inv_n : 0.f;
      }
    }
  }
  float acc_alpha = 0.f;
  // the next row of the wavefront is fetched while this one goes through its two dependent wavefront reductions, expf and
  // the second pass over the row (measured: -0.5 % of the DIN step; the statistics pass stays at ~3.5 TB/s of its two input
  // streams -- its bound is the per-row chain of reductions and transcendentals, not the loads in flight)
  float hn[EPL], gn[EPL];
  float gsn = 0.f;
  auto fetch = [&](int64_t r) {
    if (HEAD && MODE != 0) gsn = r < a.N ? a.g[r] : 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      hn[k] = (r < a.N && (FULL || e < C)) ? a.x[r * C + e] : 0.f;
      if (MODE != 0 && !HEAD) gn[k] = (r < a.N && (FULL || e < C)) ? a.g[r * C + e] : 0.f;
    }
  };
  fetch((int64_t)blockIdx.x * kWaves + wave);
  for (int64_t r = (int64_t)blockIdx.x * kWaves + wave; r < a.N; r += nw) {
    float hraw[EPL], v[EPL], gin[EPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      hraw[k] = hn[k];
      if (MODE != 0) gin[k] = HEAD ? gsn * hw[k] : gn[k];
    }
    const float gs = gsn;
    fetch(r + nw);
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      v[k] = (FULL || e < C) ? fmaf(hraw[k], sc[k], sh[k]) : 0.f;
      s += v[k];
    }
    const float avg = wave_sum_dpp(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + RH_WAVE * k;
      const float c = (FULL || e < C) ? v[k] - avg : 0.f;
      q = fmaf(c, c, q);
    }
    const float var = wave_sum_dpp(q) + a.eps * (float)C;
    const float rs = rsqrtf(var);
    if (MODE == 0) {
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + RH_WAVE * k;
        if ((FULL || e < C)) {
          const float ps = __builtin_amdgcn_rcpf(1.f + expf(-(v[k] - avg) * rs));  // v_rcp_f32: 1 ulp
          const float o = ps * v[k] + (1.f - ps) * alpha * v[k];
          if (HEAD) dot = fmaf(o, hw[k], dot);
          else a.out[r * C + e] = o;
        }
      }
      if (HEAD) {
        dot = wave_sum_dpp(dot);
        if (lane == 0) a.out[r] = dot + (a.head_b ? a.head_b[0] : 0.f);
      }
    } else {
      // out = x * (alpha + (1-alpha) ps),  ps = sigmoid(z),  z = c * rs
      float gk[EPL], tk[EPL], psk[EPL];
      float st = 0.f, stc = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + RH_WAVE * k;
        gk[k] = gin[k];
        const float c = v[k] - avg;
        psk[k] = __builtin_amdgcn_rcpf(1.f + expf(-c * rs));
        tk[k] = (FULL || e < C) ? gk[k] * v[k] * (1.f - alpha) * psk[k] * (1.f - psk[k]) : 0.f;  // dL/dz
        st += tk[k];
        stc = fmaf(tk[k], c, stc);
        if (MODE != 3) acc_alpha += (FULL || e < C) ? gk[k] * v[k] * (1.f - psk[k]) : 0.f;
        if (HEAD && MODE == 2) cs3[k] = fmaf(gs, (FULL || e < C) ? v[k] * (alpha + (1.f - alpha) * psk[k]) : 0.f, cs3[k]);
      }
      if (HEAD && MODE == 2) acc_b += gs;
      st = wave_sum_dpp(st);
      stc = wave_sum_dpp(stc);
      const float rs3 = rs * rs * rs;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + RH_WAVE * k;
        if ((FULL || e < C)) {
          const float c = v[k] - avg;
          const float gx = gk[k] * (alpha + (1.f - alpha) * psk[k]) + rs * tk[k] - rs * invC * st - rs3 * c * stc;
          if (MODE == 1) {
            a.out[r * C + e] = gx;
          } else {
            const float xhat = (hraw[k] - mu[k]) * rsd[k];
            if (MODE == 2) {
              cs1[k] += gx;
              cs2[k] = fmaf(gx, xhat, cs2[k]);
            } else {
              a.out[r * C + e] = gm[k] * rsd[k] * (gx - sg[k] - xhat * sgx[k]);
            }
          }
        }
      }
    }
  }
  if (MODE == 1 || MODE == 2) {
    acc_alpha = wave_sum(acc_alpha);
    if (lane == 0) red[wave] = acc_alpha;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < kWaves; ++w) t += red[w];
      a.alpha_partial[blockIdx.x] = t;
    }
    if (HEAD) {
      __syncthreads();
      if (lane == 0) red[wave] = acc_b;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kWaves; ++w) t += red[w];
        a.alpha_partial[gridDim.x + blockIdx.x] = t;
      }
    }
  }
  if (MODE == 2) {
    // the wavefronts of the block are summed in wavefront order: deterministic per-block partial rows
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      colred[(wave * NCS + 0) * EPL * RH_WAVE + k * RH_WAVE + lane] = cs1[k];
      colred[(wave * NCS + 1) * EPL * RH_WAVE + k * RH_WAVE + lane] = cs2[k];
      if (HEAD) colred[(wave * NCS + 2) * EPL * RH_WAVE + k * RH_WAVE + lane] = cs3[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NCS * EPL * RH_WAVE; i += RH_BLOCK) {
      const int which = i / (EPL * RH_WAVE), e = i % (EPL * RH_WAVE);
      if ((FULL || e < C)) {
        float t = 0.f;
        for (int w = 0; w < kWaves; ++w) t += colred[(w * NCS + which) * EPL * RH_WAVE + e];
        if (which < 2) a.col_partial[((int64_t)blockIdx.x * 2 + which) * C + e] = t;
        else a.head_partial[(int64_t)blockIdx.x * C + e] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same four passes for C = 64 / 128 / 256 with LPR = C / 4 lanes per row and 64 / LPR rows per wavefront pass: one
// 16-byte load per lane and row, and the per-row reductions (2 forward, 4-5 backward) run for all rows of the pass in the
// same instructions.  dice_kernel above spends one wavefront per row whatever C is: at C = 128 its fixed per-row work
// (reductions, loop, addressing) bounded it at 1.8 - 2.9 TB/s of its streams.  Rows r = (wavefront pass) * RPW + lane / LPR;
// lane l of a row owns columns 4l .. 4l + 3.
template <int LPR>
static __device__ __forceinline__ float group_sum(float v) {
#define RH_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
  RH_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  RH_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  RH_DPP_ADD(0x141);  // row_half_mirror
  RH_DPP_ADD(0x140);  // row_mirror
#undef RH_DPP_ADD
  if (LPR == 16) return v;
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  if (LPR == 64) return (r0 + r1) + (r2 + r3);
  return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
}

static __device__ __forceinline__ void unpack4(const float4 q, float* v) {
  v[0] = q.x;
  v[1] = q.y;
  v[2] = q.z;
  v[3] = q.w;
}

template <int LPR, int MODE, bool HEAD>
__global__ __launch_bounds__(RH_BLOCK) void dice_vec_kernel(const DiceArgs a) {
  RH_CHAIN_PRIO();
  constexpr int RPW = RH_WAVE / LPR;  // rows per wavefront pass
  constexpr int C = 4 * LPR;
  constexpr int NCS = HEAD ? 3 : 2;
  __shared__ float red[2 * kWaves];
  extern __shared__ float colred[];  // MODE 2: kWaves * NCS * C floats
  const int lane = threadIdx.x % RH_WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / RH_WAVE);
  const int l = lane % LPR, sub = lane / LPR;
  const int64_t np = (int64_t)gridDim.x * kWaves;  // wavefront passes in flight
  const float alpha = a.alpha[0];
  constexpr float invC = 1.f / (float)C;
  const bool bn = a.scale != nullptr;
  float sc[4], sh[4], mu[4], rsd[4], gm[4], sg[4], sgx[4], hw[4];
  float cs1[4], cs2[4], cs3[4];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
  auto col4 = [&](const float* p) { return *reinterpret_cast<const float4*>(p + 4 * l); };
  unpack4(bn ? col4(a.scale) : one4, sc);
  unpack4(bn ? col4(a.shift) : z4, sh);
  unpack4(HEAD ? col4(a.head_w) : z4, hw);
  if (MODE >= 2) {
    unpack4(col4(a.stat), mu);
    unpack4(col4(a.stat + C), rsd);
    unpack4(col4(a.gamma), gm);
    if (MODE == 3) {
      const float inv_n = 1.f / (float)a.N;
      unpack4(col4(a.stat + 2 * C), sg);
      unpack4(col4(a.stat + 3 * C), sgx);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg[k] *= inv_n;
        sgx[k] *= inv_n;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) cs1[k] = cs2[k] = cs3[k] = 0.f;
  float acc_alpha = 0.f, acc_b = 0.f;

  float4 hn = z4, gn = z4;
  float gsn = 0.f;
  auto fetch = [&](int64_t pass) {
    const int64_t r = pass * RPW + sub;
    const bool ok = r < a.N;
    hn = ok ? *reinterpret_cast<const float4*>(a.x + r * C + 4 * l) : z4;
    if (MODE != 0) {
      if (HEAD) gsn = ok ? a.g[r] : 0.f;
      else gn = ok ? *reinterpret_cast<const float4*>(a.g + r * C + 4 * l) : z4;
    }
  };
  const int64_t first = (int64_t)blockIdx.x * kWaves + wave;
  fetch(first);
  for (int64_t pass = first; pass * RPW < a.N; pass += np) {
    const int64_t r = pass * RPW + sub;
    const bool ok = r < a.N;
    float hraw[4], gin[4], v[4];
    unpack4(hn, hraw);
    const float gs = gsn;
    if (MODE != 0) {
      if (HEAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) gin[k] = gs * hw[k];
      } else {
        unpack4(gn, gin);
      }
    }
    fetch(pass + np);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = fmaf(hraw[k], sc[k], sh[k]);
      s += v[k];
    }
    const float avg = group_sum<LPR>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = v[k] - avg;
      q = fmaf(c, c, q);
    }
    const float var = group_sum<LPR>(q) + a.eps * (float)C;
    const float rs = rsqrtf(var);
    if (MODE == 0) {
      float o[4], dot = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ps = __builtin_amdgcn_rcpf(1.f + expf(-(v[k] - avg) * rs));
        o[k] = ps * v[k] + (1.f - ps) * alpha * v[k];
        dot = fmaf(o[k], hw[k], dot);
      }
      if (HEAD) {
        dot = group_sum<LPR>(dot);
        if (ok && l == 0) a.out[r] = dot + (a.head_b ? a.head_b[0] : 0.f);
      } else if (ok) {
        *reinterpret_cast<float4*>(a.out + r * C + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else {
      float tk[4], psk[4];
      float st = 0.f, stc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float c = v[k] - avg;
        psk[k] = __builtin_amdgcn_rcpf(1.f + expf(-c * rs));
        tk[k] = gin[k] * v[k] * (1.f - alpha) * psk[k] * (1.f - psk[k]);  // dL/dz
        st += tk[k];
        stc = fmaf(tk[k], c, stc);
        if (MODE != 3) acc_alpha += gin[k] * v[k] * (1.f - psk[k]);
        if (HEAD && MODE == 2) cs3[k] = fmaf(gs, v[k] * (alpha + (1.f - alpha) * psk[k]), cs3[k]);
      }
      if (HEAD && MODE == 2) acc_b += gs;
      st = group_sum<LPR>(st);
      stc = group_sum<LPR>(stc);
      const float rs3 = rs * rs * rs;
      float gx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float c = v[k] - avg;
        gx[k] = gin[k] * (alpha + (1.f - alpha) * psk[k]) + rs * tk[k] - rs * invC * st - rs3 * c * stc;
        if (MODE >= 2) {
          const float xhat = (hraw[k] - mu[k]) * rsd[k];
          if (MODE == 2) {
            // a row past N has h = 0 and g = 0: tk = 0, st = stc = 0, gx = 0 -- nothing to mask
            cs1[k] += gx[k];
            cs2[k] = fmaf(gx[k], xhat, cs2[k]);
          } else {
            gx[k] = gm[k] * rsd[k] * (gx[k] - sg[k] - xhat * sgx[k]);
          }
        }
      }
      if ((MODE == 1 || MODE == 3) && ok)
        *reinterpret_cast<float4*>(a.out + r * C + 4 * l) = make_float4(gx[0], gx[1], gx[2], gx[3]);
    }
  }
  if (MODE == 1 || MODE == 2) {
    acc_alpha = wave_sum_dpp(acc_alpha);
    if (HEAD) acc_b = wave_sum_dpp(l == 0 ? acc_b : 0.f);  // one lane per row carries that row's g
    if (lane == 0) {
      red[wave] = acc_alpha;
      red[kWaves + wave] = acc_b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f, u = 0.f;
      for (int w = 0; w < kWaves; ++w) {
        t += red[w];
        u += red[kWaves + w];
      }
      a.alpha_partial[blockIdx.x] = t;
      if (HEAD) a.alpha_partial[gridDim.x + blockIdx.x] = u;
    }
  }
  if (MODE == 2) {
    // rows of the wavefront pass first (fixed order), then the wavefronts in order: deterministic per-block partial rows
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      for (int m = LPR; m < RH_WAVE; m <<= 1) {
        cs1[k] += __shfl_xor(cs1[k], m, RH_WAVE);
        cs2[k] += __shfl_xor(cs2[k], m, RH_WAVE);
        if (HEAD) cs3[k] += __shfl_xor(cs3[k], m, RH_WAVE);
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        colred[(wave * NCS + 0) * C + 4 * l + k] = cs1[k];
        colred[(wave * NCS + 1) * C + 4 * l + k] = cs2[k];
        if (HEAD) colred[(wave * NCS + 2) * C + 4 * l + k] = cs3[k];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NCS * C; i += RH_BLOCK) {
      const int which = i / C, e = i % C;
      float t = 0.f;
      for (int w = 0; w < kWaves; ++w) t += colred[(w * NCS + which) * C + e];
      if (which < 2) a.col_partial[((int64_t)blockIdx.x * 2 + which) * C + e] = t;
      else a.head_partial[(int64_t)blockIdx.x * C + e] = t;
    }
  }
}

int g_dice_vec = 16 | 32;  // tuning knob RH_TUNE_DICE_VEC: bit mask of the LPR values dice_vec_kernel serves (C = 4 * LPR)

int dice_epl(int C) {
  int e = 1;
  while (e * RH_WAVE < C) e *= 2;
  return e;
}

unsigned dice_grid(int64_t N, int cap = 256 * 16) {
  int64_t g = (N + kWaves - 1) / kWaves;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}
constexpr int kStatsBlocks = 2048;  // MODE 2: few enough partial rows for the column finalize to combine quickly

template <int E, int MODE, bool HEAD>
void dice_launch(const DiceArgs& a, unsigned grid, hipStream_t s) {
  const size_t lds = MODE == 2 ? (size_t)kWaves * (HEAD ? 3 : 2) * E * RH_WAVE * sizeof(float) : 0;
  if (a.C == E * RH_WAVE) hipLaunchKernelGGL((dice_kernel<E, MODE, HEAD, true>), dim3(grid), dim3(RH_BLOCK), lds, s, a);
  else hipLaunchKernelGGL((dice_kernel<E, MODE, HEAD, false>), dim3(grid), dim3(RH_BLOCK), lds, s, a);
}

template <int LPR, int MODE, bool HEAD>
void dice_vec_launch(const DiceArgs& a, hipStream_t s) {
  // MODE 2: the caller sized its partial rows by rh_bn_dice_stats_blocks(N); blocks without rows write zeros
  const int64_t passes = (a.N + RH_WAVE / LPR - 1) / (RH_WAVE / LPR);
  // MODE 1 likewise: alpha_partial has rh_dice_nblocks(N) entries, every one of them must be written
  const unsigned grid = MODE == 2 ? dice_grid(a.N, kStatsBlocks) : MODE == 1 ? dice_grid(a.N) : dice_grid(passes);
  const size_t lds = MODE == 2 ? (size_t)kWaves * (HEAD ? 3 : 2) * 4 * LPR * sizeof(float) : 0;
  hipLaunchKernelGGL((dice_vec_kernel<LPR, MODE, HEAD>), dim3(grid), dim3(RH_BLOCK), lds, s, a);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MODE, bool HEAD = false>
int dice_dispatch(const DiceArgs& a, hipStream_t s) {
  if ((a.C == 64 || a.C == 128 || a.C == 256) && (g_dice_vec & (a.C / 4)) && aligned16(a.x) && aligned16(a.out) &&
      (HEAD || aligned16(a.g)) && aligned16(a.scale) && aligned16(a.shift) && aligned16(a.stat) && aligned16(a.gamma) &&
      aligned16(a.head_w)) {
    if (a.C == 64) dice_vec_launch<16, MODE, HEAD>(a, s);
    else if (a.C == 128) dice_vec_launch<32, MODE, HEAD>(a, s);
    else dice_vec_launch<64, MODE, HEAD>(a, s);
    return 0;
  }
  const unsigned grid = MODE == 2 ? dice_grid(a.N, kStatsBlocks) : dice_grid(a.N);
  switch (dice_epl(a.C)) {
    case 1: dice_launch<1, MODE, HEAD>(a, grid, s); break;
    case 2: dice_launch<2, MODE, HEAD>(a, grid, s); break;
    case 4: dice_launch<4, MODE, HEAD>(a, grid, s); break;
    case 8: dice_launch<8, MODE, HEAD>(a, grid, s); break;
    case 16:
      // the folded-BatchNorm modes keep 9 values per owned column in registers
      if constexpr (MODE >= 2 || HEAD) return RH_E_UNSUPPORTED;
      else dice_launch<16, MODE, HEAD>(a, grid, s);
      break;
    case 32:
      if constexpr (MODE >= 2 || HEAD) return RH_E_UNSUPPORTED;
      else dice_launch<32, MODE, HEAD>(a, grid, s);
      break;
    default: return RH_E_UNSUPPORTED;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// attention input / weighted pooling: one sample per group of G = LPR*LS lanes (LPR = D/4 lanes per row of 16 bytes,
// LS position groups), positions l = j*LS + ls.
struct AttArgs {
  const float* hist;  // (B, L, D), batch stride hs
  int64_t hs;
  const float* tgt;   // (B, D), batch stride ts
  int64_t ts;
  const float* w;     // (B, L) attention weights (pool)
  const float* g;     // upstream gradient
  float* o1;          // fwd: att_input (B*L, 4D) | pooled (B, D);  bwd: g_hist (B, L, D)
  float* o2;          // bwd: g_tgt (B, D) | g_w (B, L)
  int B, L, D;
};

// MODE 0: att_input fwd, 1: att_input bwd, 2: weighted pooling fwd, 3: weighted pooling bwd
template <int LPR, int LS, int MODE>
__global__ __launch_bounds__(RH_BLOCK) void att_kernel(const AttArgs a) {
  RH_CHAIN_PRIO();
  constexpr int G = LPR * LS;
  const int lig = threadIdx.x % G;
  const int q = lig % LPR, ls = lig / LPR;
  int64_t b = (int64_t)blockIdx.x * (RH_BLOCK / G) + threadIdx.x / G;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int L = a.L, D = a.D;
  const float* hrow = a.hist + b * a.hs + q * 4;
  float4 t = f4_zero();
  if (MODE == 0 || MODE == 1) t = gload<float4>(a.tgt + b * a.ts + q * 4);
  float4 gp = f4_zero();
  if (MODE == 3) gp = gload<float4>(a.g + b * D + q * 4);
  float4 acc = f4_zero();
  for (int l = ls; l < L; l += LS) {
    const float4 h = gload<float4>(hrow + (int64_t)l * D);
    if (MODE == 0) {
      float* o = a.o1 + ((b * L + l) * 4) * D + q * 4;
      if (live) {
        gstore<float4>(o, t);
        gstore<float4>(o + D, h);
        gstore<float4>(o + 2 * D, f4_sub(t, h));
        gstore<float4>(o + 3 * D, make_float4(t.x * h.x, t.y * h.y, t.z * h.z, t.w * h.w));
      }
    } else if (MODE == 1) {
      const float* gi = a.g + ((b * L + l) * 4) * D + q * 4;
      const float4 g0 = gload<float4>(gi), g1 = gload<float4>(gi + D), g2 = gload<float4>(gi + 2 * D),
                   g3 = gload<float4>(gi + 3 * D);
      // att_input = [t, h, t-h, t*h]
      const float4 gh = make_float4(g1.x - g2.x + t.x * g3.x, g1.y - g2.y + t.y * g3.y, g1.z - g2.z + t.z * g3.z,
                                    g1.w - g2.w + t.w * g3.w);
      if (live) gstore<float4>(a.o1 + (b * L + l) * D + q * 4, gh);
      acc = make_float4(acc.x + g0.x + g2.x + h.x * g3.x, acc.y + g0.y + g2.y + h.y * g3.y,
                        acc.z + g0.z + g2.z + h.z * g3.z, acc.w + g0.w + g2.w + h.w * g3.w);
    } else if (MODE == 2) {
      acc = f4_fma(a.w[b * L + l], h, acc);
    } else {
      const float wl = a.w[b * L + l];
      if (live) gstore<float4>(a.o1 + (b * L + l) * D + q * 4, f4_scale(gp, wl));
      float d = f4_dot(gp, h);
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) d += __shfl_xor(d, m, RH_WAVE);
      if (live && q == 0) a.o2[b * L + l] = d;
    }
  }
  if (MODE == 1 || MODE == 2) {
#pragma unroll
    for (int m = LPR; m < G; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
    if (live && ls == 0) gstore<float4>((MODE == 1 ? a.o2 : a.o1) + b * D + q * 4, acc);
  }
}

template <int LPR, int MODE>
int att_launch_ls(const AttArgs& a, int ls, hipStream_t s) {
#define RH_ATT(LSV)                                                                                    \
  {                                                                                                    \
    const unsigned grid = (unsigned)(((int64_t)a.B * LPR * LSV + RH_BLOCK - 1) / RH_BLOCK);            \
    hipLaunchKernelGGL((att_kernel<LPR, LSV, MODE>), dim3(grid), dim3(RH_BLOCK), 0, s, a);             \
    return 0;                                                                                          \
  }
  if constexpr (LPR * 16 <= RH_WAVE) {
    if (ls >= 16) RH_ATT(16)
  }
  if constexpr (LPR * 4 <= RH_WAVE) {
    if (ls >= 4) RH_ATT(4)
  }
  RH_ATT(1)
#undef RH_ATT
}

template <int MODE>
int att_dispatch(const AttArgs& a, hipStream_t s) {
  const int lpr = a.D / 4;
  int ls = 1;
  while (ls < 16 && lpr * ls * 4 <= RH_WAVE && ls * 4 <= a.L) ls *= 4;
  switch (lpr) {
    case 1: return att_launch_ls<1, MODE>(a, ls, s);
    case 2: return att_launch_ls<2, MODE>(a, ls, s);
    case 4: return att_launch_ls<4, MODE>(a, ls, s);
    case 8: return att_launch_ls<8, MODE>(a, ls, s);
    case 16: return att_launch_ls<16, MODE>(a, ls, s);
    case 32: return att_launch_ls<32, MODE>(a, ls, s);
    default: return RH_E_UNSUPPORTED;
  }
}

int att_check(const char* who, int B, int L, int D) {
  RH_REQUIRE(B >= 0 && L >= 1, RH_E_BADARG, "%s: bad shape B=%d L=%d", who, B, L);
  RH_REQUIRE(D > 0 && D % 4 == 0 && D <= 128 && ((D / 4) & (D / 4 - 1)) == 0, RH_E_UNSUPPORTED,
             "%s: embed_dim %d unsupported (need 4,8,16,32,64,128)", who, D);
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// PReLU with one learnable slope (nn.PReLU(), the activation of the DSSM towers: activation_layer("prelu"),
// torch_rechub/basic/activation.py:40-41 -> MLP basic/layers.py:283).  ATen runs the backward as a two-output
// elementwise kernel (26 us at 4096 x 512) + a full reduction for the slope gradient; here one pass each way, the slope
// gradient as per-block partial sums (summed by the caller / the step's packing launch).
namespace {
__global__ __launch_bounds__(RH_BLOCK) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                             int64_t n4, int64_t n, float* __restrict__ out) {
  RH_CHAIN_PRIO();
  const float a = slope[0];
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RH_BLOCK) {
    float4 v = gload<float4>(x + 4 * i);
    v.x = v.x > 0.f ? v.x : a * v.x;
    v.y = v.y > 0.f ? v.y : a * v.y;
    v.z = v.z > 0.f ? v.z : a * v.z;
    v.w = v.w > 0.f ? v.w : a * v.w;
    gstore<float4>(out + 4 * i, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // the last 1..3 elements
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    out[i] = x[i] > 0.f ? x[i] : a * x[i];
  }
}

__global__ __launch_bounds__(RH_BLOCK) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ slope, int64_t n4, int64_t n,
                                                             float* __restrict__ gx, float* __restrict__ partial) {
  RH_CHAIN_PRIO();
  __shared__ float red[RH_BLOCK / RH_WAVE];
  const float a = slope[0];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RH_BLOCK) {
    const float4 v = gload<float4>(x + 4 * i), d = gload<float4>(g + 4 * i);
    float4 o;
    o.x = v.x > 0.f ? d.x : a * d.x;
    o.y = v.y > 0.f ? d.y : a * d.y;
    o.z = v.z > 0.f ? d.z : a * d.z;
    o.w = v.w > 0.f ? d.w : a * d.w;
    acc += (v.x > 0.f ? 0.f : d.x * v.x) + (v.y > 0.f ? 0.f : d.y * v.y) + (v.z > 0.f ? 0.f : d.z * v.z) +
           (v.w > 0.f ? 0.f : d.w * v.w);
    gstore<float4>(gx + 4 * i, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    gx[i] = x[i] > 0.f ? g[i] : a * g[i];
    acc += x[i] > 0.f ? 0.f : g[i] * x[i];
  }
  acc = wave_sum(acc);
  if (threadIdx.x % RH_WAVE == 0) red[threadIdx.x / RH_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

unsigned prelu_grid(int64_t n) {
  int64_t g = (n / 4 + RH_BLOCK - 1) / RH_BLOCK;
  if (g > 1024) g = 1024;
  return (unsigned)(g < 1 ? 1 : g);
}
}  // namespace

extern "C" int rh_prelu_nblocks(int64_t n) { return (int)prelu_grid(n); }

extern "C" int rh_prelu_fwd(const float* x, const float* slope, int64_t n, float* out, void* stream) {
  RH_REQUIRE(x && slope && out && n >= 0, RH_E_BADARG, "rh_prelu_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(prelu_grid(n)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), x,
                     slope, n / 4, n, out);
  RH_LAUNCH_CHECK("rh_prelu_fwd");
  return 0;
}

// gx = g * (x > 0 ? 1 : slope);  partial[rh_prelu_nblocks(n)]: per-block sums of g * x over x <= 0 (d / d slope)
extern "C" int rh_prelu_bwd(const float* x, const float* g, const float* slope, int64_t n, float* gx, float* partial,
                            void* stream) {
  RH_REQUIRE(x && g && slope && gx && partial && n >= 1, RH_E_BADARG, "rh_prelu_bwd: bad arguments");
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(prelu_grid(n)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), x, g,
                     slope, n / 4, n, gx, partial);
  RH_LAUNCH_CHECK("rh_prelu_bwd");
  return 0;
}

extern "C" int rh_din_set_tuning(int key, int value) {
  if (key == RH_TUNE_DICE_VEC) {
    g_dice_vec = value;
    return 0;
  }
  return RH_E_BADARG;
}

extern "C" int rh_dice_nblocks(int64_t N) { return (int)dice_grid(N); }
extern "C" int rh_bn_dice_stats_blocks(int64_t N) { return (int)dice_grid(N, kStatsBlocks); }

extern "C" int rh_dice_fwd(const float* x, const float* alpha, float eps, int64_t N, int C, const float* bn_scale,
                           const float* bn_shift, float* out, void* stream) {
  RH_REQUIRE(x && alpha && out && N >= 0 && C >= 1, RH_E_BADARG, "rh_dice_fwd: bad arguments");
  RH_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), RH_E_BADARG, "rh_dice_fwd: scale and shift go together");
  RH_REQUIRE(C <= 2048, RH_E_UNSUPPORTED, "rh_dice_fwd: %d neurons unsupported (max 2048)", C);
  if (N == 0) return 0;
  DiceArgs a{x, nullptr, alpha, eps, N, C, out, nullptr, bn_scale, bn_shift, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int rc = dice_dispatch<0>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_dice_fwd");
  return 0;
}

extern "C" int rh_dice_bwd(const float* x, const float* g, const float* alpha, float eps, int64_t N, int C, float* gx,
                           float* alpha_partial, void* stream) {
  RH_REQUIRE(x && g && alpha && gx && alpha_partial && N >= 0 && C >= 1, RH_E_BADARG, "rh_dice_bwd: bad arguments");
  RH_REQUIRE(C <= 2048, RH_E_UNSUPPORTED, "rh_dice_bwd: %d neurons unsupported (max 2048)", C);
  DiceArgs a{x, g, alpha, eps, N, C, gx, alpha_partial, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int rc = dice_dispatch<1>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_dice_bwd");
  return 0;
}

extern "C" int rh_bn_dice_bwd_stats(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                                    const float* stat, const float* gamma, float* col_partial, float* alpha_partial,
                                    void* stream) {
  RH_REQUIRE(h && g && alpha && stat && gamma && col_partial && alpha_partial && N >= 1 && C >= 1, RH_E_BADARG,
             "rh_bn_dice_bwd_stats: bad arguments");
  RH_REQUIRE(C <= 512, RH_E_UNSUPPORTED, "rh_bn_dice_bwd_stats: %d neurons unsupported (max 512)", C);
  DiceArgs a{h, g, alpha, eps, N, C, nullptr, alpha_partial, stat + 4 * (int64_t)C, stat + 5 * (int64_t)C, stat, gamma,
             col_partial, nullptr, nullptr, nullptr};
  int rc = dice_dispatch<2>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_bn_dice_bwd_stats");
  return 0;
}

extern "C" int rh_bn_dice_bwd_apply(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                                    const float* stat, const float* gamma, float* dh, void* stream) {
  RH_REQUIRE(h && g && alpha && stat && gamma && dh && N >= 1 && C >= 1, RH_E_BADARG, "rh_bn_dice_bwd_apply: bad arguments");
  RH_REQUIRE(C <= 512, RH_E_UNSUPPORTED, "rh_bn_dice_bwd_apply: %d neurons unsupported (max 512)", C);
  DiceArgs a{h, g, alpha, eps, N, C, dh, nullptr, stat + 4 * (int64_t)C, stat + 5 * (int64_t)C, stat, gamma, nullptr, nullptr, nullptr, nullptr};
  int rc = dice_dispatch<3>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_bn_dice_bwd_apply");
  return 0;
}

extern "C" int rh_bn_dice_head_fwd(const float* h, const float* alpha, float eps, int64_t N, int C, const float* bn_scale,
                                   const float* bn_shift, const float* head_w, const float* head_b, float* out,
                                   void* stream) {
  RH_REQUIRE(h && alpha && bn_scale && bn_shift && head_w && out && N >= 0 && C >= 1, RH_E_BADARG,
             "rh_bn_dice_head_fwd: bad arguments");
  RH_REQUIRE(C <= 512, RH_E_UNSUPPORTED, "rh_bn_dice_head_fwd: %d neurons unsupported (max 512)", C);
  if (N == 0) return 0;
  DiceArgs a{h, nullptr, alpha, eps, N, C, out, nullptr, bn_scale, bn_shift, nullptr, nullptr, nullptr, head_w, head_b,
             nullptr};
  int rc = dice_dispatch<0, true>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_bn_dice_head_fwd");
  return 0;
}

extern "C" int rh_bn_dice_head_bwd_stats(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                                         const float* stat, const float* gamma, const float* head_w, float* col_partial,
                                         float* alpha_partial, float* head_partial, void* stream) {
  RH_REQUIRE(h && g && alpha && stat && gamma && head_w && col_partial && alpha_partial && head_partial && N >= 1 && C >= 1,
             RH_E_BADARG, "rh_bn_dice_head_bwd_stats: bad arguments");
  RH_REQUIRE(C <= 512, RH_E_UNSUPPORTED, "rh_bn_dice_head_bwd_stats: %d neurons unsupported (max 512)", C);
  DiceArgs a{h, g, alpha, eps, N, C, nullptr, alpha_partial, stat + 4 * (int64_t)C, stat + 5 * (int64_t)C, stat, gamma,
             col_partial, head_w, nullptr, head_partial};
  int rc = dice_dispatch<2, true>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_bn_dice_head_bwd_stats");
  return 0;
}

extern "C" int rh_bn_dice_head_bwd_apply(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                                         const float* stat, const float* gamma, const float* head_w, float* dh,
                                         void* stream) {
  RH_REQUIRE(h && g && alpha && stat && gamma && head_w && dh && N >= 1 && C >= 1, RH_E_BADARG,
             "rh_bn_dice_head_bwd_apply: bad arguments");
  RH_REQUIRE(C <= 512, RH_E_UNSUPPORTED, "rh_bn_dice_head_bwd_apply: %d neurons unsupported (max 512)", C);
  DiceArgs a{h, g, alpha, eps, N, C, dh, nullptr, stat + 4 * (int64_t)C, stat + 5 * (int64_t)C, stat, gamma, nullptr, head_w,
             nullptr, nullptr};
  int rc = dice_dispatch<3, true>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_bn_dice_head_bwd_apply");
  return 0;
}

extern "C" int rh_din_att_input_fwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride, int B,
                                    int L, int D, float* out, void* stream) {
  if (int rc = att_check("rh_din_att_input_fwd", B, L, D)) return rc;
  RH_REQUIRE(hist && tgt && out, RH_E_BADARG, "rh_din_att_input_fwd: null pointer");
  if (B == 0) return 0;
  AttArgs a{hist, hist_stride, tgt, tgt_stride, nullptr, nullptr, out, nullptr, B, L, D};
  int rc = att_dispatch<0>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_din_att_input_fwd");
  return 0;
}

extern "C" int rh_din_att_input_bwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride,
                                    const float* g, int B, int L, int D, float* g_hist, float* g_tgt, void* stream) {
  if (int rc = att_check("rh_din_att_input_bwd", B, L, D)) return rc;
  RH_REQUIRE(hist && tgt && g && g_hist && g_tgt, RH_E_BADARG, "rh_din_att_input_bwd: null pointer");
  if (B == 0) return 0;
  AttArgs a{hist, hist_stride, tgt, tgt_stride, nullptr, g, g_hist, g_tgt, B, L, D};
  int rc = att_dispatch<1>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_din_att_input_bwd");
  return 0;
}

extern "C" int rh_din_pool_fwd(const float* hist, int64_t hist_stride, const float* w, int B, int L, int D, float* out,
                               void* stream) {
  if (int rc = att_check("rh_din_pool_fwd", B, L, D)) return rc;
  RH_REQUIRE(hist && w && out, RH_E_BADARG, "rh_din_pool_fwd: null pointer");
  if (B == 0) return 0;
  AttArgs a{hist, hist_stride, nullptr, 0, w, nullptr, out, nullptr, B, L, D};
  int rc = att_dispatch<2>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_din_pool_fwd");
  return 0;
}

extern "C" int rh_din_pool_bwd(const float* hist, int64_t hist_stride, const float* w, const float* g, int B, int L,
                               int D, float* g_hist, float* g_w, void* stream) {
  if (int rc = att_check("rh_din_pool_bwd", B, L, D)) return rc;
  RH_REQUIRE(hist && w && g && g_hist && g_w, RH_E_BADARG, "rh_din_pool_bwd: null pointer");
  if (B == 0) return 0;
  AttArgs a{hist, hist_stride, nullptr, 0, w, g, g_hist, g_w, B, L, D};
  int rc = att_dispatch<3>(a, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  RH_LAUNCH_CHECK("rh_din_pool_bwd");
  return 0;
}
