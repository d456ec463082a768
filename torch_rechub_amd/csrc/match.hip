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
