// In-batch negatives of the two-tower trainer WITHOUT the (B, C) score matrix.
//
// Reference: MatchTrainer.train_one_epoch, in-batch branch, torch_rechub/trainers/match_trainer.py:118-138
//     scores = torch.matmul(user_embedding, item_embedding.t())            # (B, B): every pair
//     neg_indices = inbatch_negative_sampling(scores, neg_ratio, ...)       # K columns per row (utils/match.py:104-145)
//     logits = gather_inbatch_logits(scores, neg_indices)                   # (B, 1 + K): diagonal + gathered (:148-153)
// With random negatives only 1 + K of a row's B scores are ever read (K = 20 of 4096 at configs[4]): the product, the
// zero-filled (B, B) gradient of the gather and the two (B, B)-sized gradient products are 6.4 GFLOP and ~0.5 GB per step
// for 11 MFLOP of wanted dot products.  Here: logits[i, 0] = u_i . v_(row0 + i), logits[i, 1 + k] = u_i . v_neg[i, k]
// straight from the embeddings (one wavefront per row, the gathered item rows are 256-byte coalesced loads), and the
// backward g_u[i] = sum_k g[i, k] v_idx, g_v[idx] += g[i, k] u_i (row-wide float atomics into a zeroed buffer: the rows a
// batch hits are random, so contention is one or two adds per line).  The hard-negative branch (top-k over the whole
// row) needs every score and keeps the product.
// Same numbers as the reference up to the summation order of a D-term dot product.  Roofline: launch latency (B K D 4
// bytes = 21 MB of gathered rows at B = 4096, K = 20, D = 64).
#include "common.h"

namespace {

constexpr int kWaves = RH_BLOCK / RH_WAVE;
constexpr int kMaxPerLane = 16;  // D <= 1024

struct InbatchArgs {
  const float* u;  // (B, D), row stride ldu
  int64_t ldu;
  const float* v;  // (C, D), row stride ldv
  int64_t ldv;
  const int64_t* neg;  // (B, K) columns in [0, C)
  const float* g;      // (B, 1 + K)   backward
  float* logits;       // (B, 1 + K)   forward
  float* g_u;          // (B, D)       backward
  float* g_v;          // (C, D)       backward, zeroed by the caller
  int B, C, D, K, row0;
  int* err;
};

template <int PL, bool BWD>  // PL = ceil(D / 64) floats per lane
__global__ __launch_bounds__(RH_BLOCK) void inbatch_logits_kernel(const InbatchArgs a) {
  RH_CHAIN_PRIO();
  const int lane = threadIdx.x % RH_WAVE, wave = threadIdx.x / RH_WAVE;
  const int K1 = a.K + 1;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * kWaves + wave; i < a.B; i += (int64_t)gridDim.x * kWaves) {
    float uv[PL], gu[PL];
#pragma unroll
    for (int e = 0; e < PL; ++e) {
      const int d = lane + e * RH_WAVE;
      uv[e] = d < a.D ? a.u[i * a.ldu + d] : 0.f;
      gu[e] = 0.f;
    }
    for (int k = 0; k < K1; ++k) {  // wavefront-uniform
      int64_t j = k == 0 ? (int64_t)a.row0 + i : a.neg[i * a.K + k - 1];
      if ((uint64_t)j >= (uint64_t)a.C) {
        bad = true;
        j = 0;
      }
      const float* vr = a.v + j * a.ldv;
      if (!BWD) {
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < PL; ++e) {
          const int d = lane + e * RH_WAVE;
          if (d < a.D) dot = fmaf(uv[e], vr[d], dot);
        }
        dot = wave_sum(dot);
        if (lane == 0) a.logits[i * K1 + k] = dot;
      } else {
        const float gk = a.g[i * K1 + k];
        float* gv = a.g_v + j * a.D;
#pragma unroll
        for (int e = 0; e < PL; ++e) {
          const int d = lane + e * RH_WAVE;
          if (d < a.D) {
            gu[e] = fmaf(gk, vr[d], gu[e]);
            atomicAdd(gv + d, gk * uv[e]);
          }
        }
      }
    }
    if (BWD) {
#pragma unroll
      for (int e = 0; e < PL; ++e) {
        const int d = lane + e * RH_WAVE;
        if (d < a.D) a.g_u[i * a.D + d] = gu[e];
      }
    }
  }
  if (bad && a.err != nullptr) atomicOr(a.err, RH_FLAG_INDEX_OOB);
}

template <bool BWD>
int launch(const InbatchArgs& a, hipStream_t s) {
  int64_t grid = ((int64_t)a.B + kWaves - 1) / kWaves;
  if (grid > 256 * 16) grid = 256 * 16;
  const int pl = (a.D + RH_WAVE - 1) / RH_WAVE;
  const dim3 g((unsigned)grid), b(RH_BLOCK);
  if (pl <= 1) hipLaunchKernelGGL((inbatch_logits_kernel<1, BWD>), g, b, 0, s, a);
  else if (pl <= 2) hipLaunchKernelGGL((inbatch_logits_kernel<2, BWD>), g, b, 0, s, a);
  else if (pl <= 4) hipLaunchKernelGGL((inbatch_logits_kernel<4, BWD>), g, b, 0, s, a);
  else if (pl <= 8) hipLaunchKernelGGL((inbatch_logits_kernel<8, BWD>), g, b, 0, s, a);
  else hipLaunchKernelGGL((inbatch_logits_kernel<kMaxPerLane, BWD>), g, b, 0, s, a);
  return 0;
}

int check(const char* who, int B, int C, int D, int K, int row0) {
  RH_REQUIRE(B >= 1 && C >= 1 && K >= 0 && row0 >= 0 && row0 + B <= C, RH_E_BADARG, "%s: bad shape B=%d C=%d K=%d row0=%d",
             who, B, C, K, row0);
  RH_REQUIRE(D >= 1 && D <= RH_WAVE * kMaxPerLane, RH_E_UNSUPPORTED, "%s: embedding width %d unsupported (max %d)", who, D,
             RH_WAVE * kMaxPerLane);
  return 0;
}

}  // namespace

extern "C" int rh_inbatch_logits_fwd(const float* u, int64_t ldu, const float* v, int64_t ldv, const int64_t* neg, int B,
                                     int C, int D, int K, int row0, float* logits, int32_t* err_flag, void* stream) {
  RH_REQUIRE(u && v && logits && (neg || K == 0) && ldu >= D && ldv >= D, RH_E_BADARG, "rh_inbatch_logits_fwd: bad arguments");
  if (int rc = check("rh_inbatch_logits_fwd", B, C, D, K, row0)) return rc;
  InbatchArgs a{u, ldu, v, ldv, neg, nullptr, logits, nullptr, nullptr, B, C, D, K, row0, err_flag};
  launch<false>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_inbatch_logits_fwd");
  return 0;
}

extern "C" int rh_inbatch_logits_bwd(const float* u, int64_t ldu, const float* v, int64_t ldv, const int64_t* neg,
                                     const float* g, int B, int C, int D, int K, int row0, float* g_u, float* g_v,
                                     void* stream) {
  RH_REQUIRE(u && v && g && g_u && g_v && (neg || K == 0) && ldu >= D && ldv >= D, RH_E_BADARG,
             "rh_inbatch_logits_bwd: bad arguments");
  if (int rc = check("rh_inbatch_logits_bwd", B, C, D, K, row0)) return rc;
  InbatchArgs a{u, ldu, v, ldv, neg, g, nullptr, g_u, g_v, B, C, D, K, row0, nullptr};
  launch<true>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_inbatch_logits_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-wise L2 normalisation of the towers' outputs and the mean cross-entropy over the (B, 1 + K) in-batch logits: the
// reference's F.normalize(.., p=2, dim=1) (models/matching/dssm.py:56,66: norm -> clamp_min(eps) -> expand -> div, and
// its five-kernel backward) and torch.nn.CrossEntropyLoss() (trainers/match_trainer.py:60,136: log_softmax + nll_loss)
// were ~30 ATen launches (~160 us of ~1.2 ms) in the configs[4] step.  One launch each way here.
namespace {

// 16 lanes per row (float4 each, d walked in chunks of 64 floats): y = x / max(||x||, eps); nrm saved for the backward
__global__ __launch_bounds__(RH_BLOCK) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t ldx, int B, int d, float eps,
                                                              float* __restrict__ y, float* __restrict__ nrm) {
  RH_CHAIN_PRIO();
  const int q = threadIdx.x % 16, slot = threadIdx.x / 16;
  for (int64_t r = (int64_t)blockIdx.x * (RH_BLOCK / 16) + slot; r < B; r += (int64_t)gridDim.x * (RH_BLOCK / 16)) {
    float ss = 0.f;
    for (int c = q * 4; c < d; c += 64) {
      const float4 v = gload<float4>(x + r * ldx + c);
      ss += f4_dot(v, v);
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m, RH_WAVE);
    const float n = sqrtf(ss);
    const float inv = 1.f / fmaxf(n, eps);
    for (int c = q * 4; c < d; c += 64) gstore<float4>(y + r * (int64_t)d + c, f4_scale(gload<float4>(x + r * ldx + c), inv));
    if (q == 0) nrm[r] = n;
  }
}

// gx = (g - y (g . y)) / ||x||  where ||x|| > eps;  gx = g / eps where the norm was clamped (a constant denominator)
__global__ __launch_bounds__(RH_BLOCK) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ nrm,
                                                              const float* __restrict__ g, int64_t ldg, int B, int d, float eps,
                                                              float* __restrict__ gx) {
  RH_CHAIN_PRIO();
  const int q = threadIdx.x % 16, slot = threadIdx.x / 16;
  for (int64_t r = (int64_t)blockIdx.x * (RH_BLOCK / 16) + slot; r < B; r += (int64_t)gridDim.x * (RH_BLOCK / 16)) {
    const float n = nrm[r];
    const bool clamped = !(n > eps);
    float dot = 0.f;
    for (int c = q * 4; c < d; c += 64) dot += f4_dot(gload<float4>(g + r * ldg + c), gload<float4>(y + r * (int64_t)d + c));
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) dot += __shfl_xor(dot, m, RH_WAVE);
    const float inv = 1.f / fmaxf(n, eps);
    if (clamped) dot = 0.f;
    for (int c = q * 4; c < d; c += 64) {
      const float4 gv = gload<float4>(g + r * ldg + c), yv = gload<float4>(y + r * (int64_t)d + c);
      gstore<float4>(gx + r * (int64_t)d + c, f4_scale(f4_fma(-dot, yv, gv), inv));
    }
  }
}

// one lane per row (C <= 64 classes: the in-batch logits have 1 + K columns): loss terms lse - x[target] summed per block
__global__ __launch_bounds__(RH_BLOCK) void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ target, int B,
                                                          int C, float* __restrict__ lse, float* __restrict__ partial, int* err) {
  RH_CHAIN_PRIO();
  __shared__ float red[RH_BLOCK / RH_WAVE];
  const int64_t r = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
  float term = 0.f;
  if (r < B) {
    const float* row = x + r * (int64_t)C;
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - m);
    const float l = m + logf(s);
    lse[r] = l;
    int64_t t = target ? target[r] : 0;
    if ((uint64_t)t >= (uint64_t)C) {
      if (err) atomicOr(err, RH_FLAG_INDEX_OOB);
      t = 0;
    }
    term = l - row[t];
  }
  term = wave_sum(term);
  if (threadIdx.x % RH_WAVE == 0) red[threadIdx.x / RH_WAVE] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < RH_BLOCK / RH_WAVE; ++w) tot += red[w];
    partial[blockIdx.x] = tot;
  }
}

__global__ __launch_bounds__(RH_BLOCK) void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ target,
                                                          const float* __restrict__ lse, const float* __restrict__ g_loss, int B,
                                                          int C, float* __restrict__ gx) {
  RH_CHAIN_PRIO();
  const float scale = g_loss[0] / (float)B;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < (int64_t)B * C; i += (int64_t)gridDim.x * RH_BLOCK) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    int64_t t = target ? target[r] : 0;
    if ((uint64_t)t >= (uint64_t)C) t = 0;
    gx[i] = scale * (expf(x[i] - lse[r]) - (c == t ? 1.f : 0.f));
  }
}

}  // namespace

extern "C" int rh_l2norm_fwd(const float* x, int64_t ldx, int B, int d, float eps, float* y, float* nrm, void* stream) {
  RH_REQUIRE(x && y && nrm, RH_E_BADARG, "rh_l2norm_fwd: null pointer");
  RH_REQUIRE(B >= 0 && d >= 4 && d % 4 == 0 && d <= 4096 && ldx >= d && ldx % 4 == 0, RH_E_UNSUPPORTED,
             "rh_l2norm_fwd: B=%d d=%d ldx=%lld unsupported (d, ldx multiples of 4, d <= 4096)", B, d, (long long)ldx);
  if (B == 0) return 0;
  const int rows_per_block = RH_BLOCK / 16;
  int64_t grid = ((int64_t)B + rows_per_block - 1) / rows_per_block;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                     B, d, eps, y, nrm);
  RH_LAUNCH_CHECK("rh_l2norm_fwd");
  return 0;
}

extern "C" int rh_l2norm_bwd(const float* y, const float* nrm, const float* g, int64_t ldg, int B, int d, float eps, float* gx,
                             void* stream) {
  RH_REQUIRE(y && nrm && g && gx, RH_E_BADARG, "rh_l2norm_bwd: null pointer");
  RH_REQUIRE(B >= 0 && d >= 4 && d % 4 == 0 && d <= 4096 && ldg >= d && ldg % 4 == 0, RH_E_UNSUPPORTED,
             "rh_l2norm_bwd: B=%d d=%d ldg=%lld unsupported", B, d, (long long)ldg);
  if (B == 0) return 0;
  const int rows_per_block = RH_BLOCK / 16;
  int64_t grid = ((int64_t)B + rows_per_block - 1) / rows_per_block;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), y, nrm,
                     g, ldg, B, d, eps, gx);
  RH_LAUNCH_CHECK("rh_l2norm_bwd");
  return 0;
}

extern "C" int rh_ce_nblocks(int B) { return (B + RH_BLOCK - 1) / RH_BLOCK; }

extern "C" int rh_ce_fwd(const float* logits, const int64_t* target, int B, int C, float* lse, float* loss_partial,
                         int32_t* err_flag, void* stream) {
  RH_REQUIRE(logits && lse && loss_partial, RH_E_BADARG, "rh_ce_fwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1 && C <= 1024, RH_E_UNSUPPORTED, "rh_ce_fwd: B=%d C=%d unsupported (1 <= C <= 1024)", B, C);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)rh_ce_nblocks(B)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     logits, target, B, C, lse, loss_partial, err_flag);
  RH_LAUNCH_CHECK("rh_ce_fwd");
  return 0;
}

extern "C" int rh_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* g_loss, int B, int C,
                         float* g_logits, void* stream) {
  RH_REQUIRE(logits && lse && g_loss && g_logits, RH_E_BADARG, "rh_ce_bwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1 && C <= 1024, RH_E_UNSUPPORTED, "rh_ce_bwd: B=%d C=%d unsupported", B, C);
  int64_t grid = ((int64_t)B * C + RH_BLOCK - 1) / RH_BLOCK;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), logits, target,
                     lse, g_loss, B, C, g_logits);
  RH_LAUNCH_CHECK("rh_ce_bwd");
  return 0;
}

