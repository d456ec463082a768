// MLP hidden-layer epilogue: BatchNorm1d (training or eval) + ReLU + Dropout fused, forward and backward.
//
// Reference: MLP.__init__/forward torch_rechub/basic/layers.py:276-292
//     layers.append(nn.Linear(input_dim, i_dim)); layers.append(nn.BatchNorm1d(i_dim));
//     layers.append(activation_layer(activation)); layers.append(nn.Dropout(p=dropout))
// The Linear stays a library GEMM (true dense contraction); what follows it is 7 ATen kernels forward
// (batch_norm_collect_statistics, transform_input, running-stat updates, relu, dropout) and 4 backward.  Here:
//   B > 8192 (three launches per direction; B <= 8192 folds the finalize into the apply launch, see below):
//   forward : column partial sums -> finalize (mean, rstd, running stats, num_batches_tracked, dropout counter)
//             -> y = dropout(relu((h - mean) * rstd * gamma + beta))
//   backward: g1 = dy * keep/(1-p) * [bn > 0]; column partial sums of g1, g1*xhat -> finalize (dgamma, dbeta)
//             -> dx = gamma * rstd * (g1 - mean(g1) - xhat * mean(g1 * xhat))
// Roofline: HBM; forward reads h twice and writes y once (12 B/element), backward reads h, dy twice and writes dx.
// Column sums use a per-column shift (row 0) so that E[x^2] - E[x]^2 does not cancel.
// Dropout mask = counter-based hash of (seed, per-call counter, element index): recomputed in the backward, never stored,
// and hipGraph-safe (the counter lives in device memory and is bumped by the finalize kernel).
#include "common.h"

namespace {

constexpr int kRowsPerChunk = 16;   // 256 row chunks at B = 4096: enough workgroups to fill the chip
constexpr int kFinCols = 32;        // finalize: 32 columns x 8 chunk groups per block

struct BnArgs {
  const float* h;      // (B, C) pre-BN activations
  const float* dy;     // backward: gradient of the output
  float* out;          // forward: y ; backward: dx
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  float* partial;      // (nchunks, 2, C)
  float* stat;         // (4, C): mean, rstd, (backward: sum g1, sum g1*xhat)
  float* dgamma;
  float* dbeta;
  int64_t* rng;        // [0] = seed, [1] = running call counter
  int64_t* saved_ctr;  // (1,) counter value used by this call (written forward, read backward)
  int B, C, nchunks;
  int rows_per_chunk;  // rows per partial chunk
  int bookkeep;        // this launch sequence advances the dropout counter / num_batches_tracked (partial or finalize kernel)
  float momentum, eps, p_drop;
  int training;
  int affine_out;      // forward finalize also writes stat rows 4, 5: scale = gamma*rstd, shift = beta - mean*scale
  int relu;            // 1: ReLU after the normalisation (MLP hidden layer); 0: BatchNorm only (Dice follows);
                       // 2: PReLU with ONE slope (nn.PReLU(), the two-tower MLPs): y = bn > 0 ? bn : slope * bn
  const float* slope;  // relu == 2
  float* slope_partial;  // relu == 2, backward: per-block sums of dy * min(bn, 0) (one per workgroup of bn_partial_kernel<1>)
  // rh_bn_finalize_bwd_tail (round 6): further per-block partials of the same statistics pass, summed by the SAME launch
  const float* extra;     // (nchunks, C) -> extra_out (C): a third column sum (the head's weight gradient), or null
  float* extra_out;
  const float* scal;      // (nscal, nchunks) -> scal_out (nscal): scalar sums (Dice alpha, head bias), one more workgroup each
  float* scal_out;
  int nscal;
  int col_blocks;         // workgroups [0, col_blocks) take columns, [col_blocks, col_blocks + nscal) one scalar row each
};

// activation after the normalisation and its derivative factor (relu: 0 none, 1 ReLU, 2 PReLU)
static __device__ __forceinline__ float bn_act(int relu, float sl, float bn) {
  return relu == 1 ? fmaxf(bn, 0.f) : (relu == 2 ? (bn > 0.f ? bn : sl * bn) : bn);
}
static __device__ __forceinline__ float bn_act_grad(int relu, float sl, float bn, float dy) {
  return relu == 1 ? (bn > 0.f ? dy : 0.f) : (relu == 2 ? (bn > 0.f ? dy : sl * dy) : dy);
}

static __device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  return rh_drop_hash(seed, ctr, idx);
}

// thread (rsub, c): column c of the tile, rows rsub, rsub+RS, ... of the chunk.  MODE 0: sums of (x-s), (x-s)^2 with
// s = h[0, c].  MODE 1 (backward): sums of g1 and g1*xhat.
template <int MODE>
__global__ __launch_bounds__(RH_BLOCK) void bn_partial_kernel(const BnArgs a, int CW) {
  RH_CHAIN_PRIO();
  __shared__ float red[2][RH_BLOCK];
  const int RS = RH_BLOCK / CW;
  const int c = blockIdx.x * CW + threadIdx.x % CW;
  const int rsub = threadIdx.x / CW;
  const int r0 = blockIdx.y * a.rows_per_chunk;
  const int r1 = min(r0 + a.rows_per_chunk, a.B);
  if (MODE == 0 && a.bookkeep && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    a.saved_ctr[0] = a.rng[1];  // dropout stream of this call (the apply launch reads saved_ctr)
    a.rng[1] += 1;
    if (a.num_batches_tracked != nullptr) a.num_batches_tracked[0] += 1;
  }
  float s1 = 0.f, s2 = 0.f, sgs = 0.f;
  if (c < a.C) {
    if (MODE == 0) {
      const float shift = a.h[c];
#pragma unroll 8
      for (int r = r0 + rsub; r < r1; r += RS) {
        const float x = a.h[(int64_t)r * a.C + c] - shift;
        s1 += x;
        s2 = fmaf(x, x, s2);
      }
    } else {
      const float mean = a.stat[c], rstd = a.stat[a.C + c];
      const float g = a.gamma[c], bt = a.beta[c];
      const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
      const uint32_t thr = (uint32_t)(a.p_drop * 4294967296.0);
      const uint64_t seed = (uint64_t)a.rng[0], ctr = (uint64_t)a.saved_ctr[0];
      const float sl = a.relu == 2 ? a.slope[0] : 0.f;
#pragma unroll 4
      for (int r = r0 + rsub; r < r1; r += RS) {
        const int64_t i = (int64_t)r * a.C + c;
        const float xhat = (a.h[i] - mean) * rstd;
        const float bn = fmaf(xhat, g, bt);
        float dyv = a.dy[i];
        if (a.p_drop > 0.f) dyv = drop_hash(seed, ctr, (uint64_t)i) >= thr ? dyv * keep_scale : 0.f;
        const float g1 = bn_act_grad(a.relu, sl, bn, dyv);
        s1 += g1;
        s2 = fmaf(g1, xhat, s2);
        if (a.relu == 2) sgs = fmaf(dyv, fminf(bn, 0.f), sgs);  // d/d slope
      }
    }
  }
  if (MODE == 1 && a.relu == 2) {  // block-uniform: the slope gradient of this workgroup's tile
    __shared__ float sred[RH_BLOCK / RH_WAVE];
    const float w = wave_sum(sgs);
    if (threadIdx.x % RH_WAVE == 0) sred[threadIdx.x / RH_WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < RH_BLOCK / RH_WAVE; ++k) tot += sred[k];
      a.slope_partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (rsub == 0 && c < a.C) {
    for (int k = 1; k < RS; ++k) {
      s1 += red[0][k * CW + threadIdx.x];
      s2 += red[1][k * CW + threadIdx.x];
    }
    if (MODE == 0) {
      // forward partials are (chunk sum, chunk M2 about the chunk mean): what Chan's parallel-variance combination
      // takes, and what the GEMM epilogue (csrc/gemm.hip) emits directly from its accumulators
      const float nk = (float)(r1 - r0);
      const float sum = fmaf(nk, a.h[c], s1);
      s2 = fmaxf(s2 - s1 * s1 / nk, 0.f);
      s1 = sum;
    }
    a.partial[((int64_t)blockIdx.y * 2 + 0) * a.C + c] = s1;
    a.partial[((int64_t)blockIdx.y * 2 + 1) * a.C + c] = s2;
  }
}

// Chan et al. combination of per-chunk (sum_k, M2_k, n_k): mean = sum_k sum_k / n, M2 = sum_k [M2_k + n_k (mean_k - mean)^2].
// Two passes over one column's partials by GROUPS cooperating threads (fixed order: deterministic); returns mean and
// the biased variance to every thread of the column.  red: [2][GROUPS][COLS + 1] floats of LDS.
template <int GROUPS, int COLS, int MAXIT>
static __device__ __forceinline__ void chan_combine(const BnArgs& a, int c, int cl, int grp, bool cok, float* red,
                                                    float& mean, float& var) {
  // MAXIT * GROUPS >= nchunks on the fused path: every thread's partials are loaded up front (independent loads, one
  // latency), then both passes run from registers; longer chunk lists (3-launch path) fall back to a second read.
  float ps[MAXIT], pm[MAXIT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int k = grp + i * GROUPS;
    const bool ok = cok && k < a.nchunks;
    ps[i] = ok ? a.partial[((int64_t)k * 2 + 0) * a.C + c] : 0.f;
    pm[i] = ok ? a.partial[((int64_t)k * 2 + 1) * a.C + c] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) s += ps[i];
  if (cok) {
    // (lists longer than MAXIT * GROUPS -- the 6400 slabs of a tile GEMM over 409 600 rows: batches of 8 independent loads, the
    // adds in chunk order; as a load -> add chain this tail cost ~2.5 us per chunk beside a bandwidth-bound neighbour)
    int k = grp + MAXIT * GROUPS;
    for (; k + 7 * GROUPS < a.nchunks; k += 8 * GROUPS) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = a.partial[((int64_t)(k + i * GROUPS) * 2 + 0) * a.C + c];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += t[i];
    }
    for (; k < a.nchunks; k += GROUPS) s += a.partial[((int64_t)k * 2 + 0) * a.C + c];
  }
  red[grp * (COLS + 1) + cl] = s;
  __syncthreads();
  s = red[cl];
#pragma unroll
  for (int q = 1; q < GROUPS; ++q) s += red[q * (COLS + 1) + cl];
  mean = s / (float)a.B;
  __syncthreads();
  const float full = (float)a.rows_per_chunk, inv_full = 1.f / full;
  const int last = a.nchunks - 1;
  const float tail = (float)(a.B - last * a.rows_per_chunk), inv_tail = 1.f / tail;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int k = grp + i * GROUPS;
    if (cok && k < a.nchunks) {
      const float d = ps[i] * (k == last ? inv_tail : inv_full) - mean;
      m2 += fmaf((k == last ? tail : full) * d, d, pm[i]);
    }
  }
  if (cok) {
    int k = grp + MAXIT * GROUPS;
    for (; k + 7 * GROUPS < a.nchunks; k += 8 * GROUPS) {
      float t0[8], t1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t0[i] = a.partial[((int64_t)(k + i * GROUPS) * 2 + 0) * a.C + c];
        t1[i] = a.partial[((int64_t)(k + i * GROUPS) * 2 + 1) * a.C + c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = k + i * GROUPS;
        const float d = t0[i] * (kk == last ? inv_tail : inv_full) - mean;
        m2 += fmaf((kk == last ? tail : full) * d, d, t1[i]);
      }
    }
    for (; k < a.nchunks; k += GROUPS) {
      const float d = a.partial[((int64_t)k * 2 + 0) * a.C + c] * (k == last ? inv_tail : inv_full) - mean;
      m2 += fmaf((k == last ? tail : full) * d, d, a.partial[((int64_t)k * 2 + 1) * a.C + c]);
    }
  }
  red[grp * (COLS + 1) + cl] = m2;
  __syncthreads();
  m2 = red[cl];
#pragma unroll
  for (int q = 1; q < GROUPS; ++q) m2 += red[q * (COLS + 1) + cl];
  var = fmaxf(m2 / (float)a.B, 0.f);
}

// FINCOLS columns x (256 / FINCOLS) chunk groups per block.  The backward of a B * L = 409600-row layer (DIN's attention
// MLP) hands over 2048 partial rows: with 32 columns per block that is 8 blocks walking 256 rows each (68 us measured);
// 4 columns per block = 64 blocks x 32 rows.
template <int MODE, int FINCOLS = kFinCols>
__global__ __launch_bounds__(RH_BLOCK) void bn_finalize_kernel(const BnArgs a) {
  RH_CHAIN_PRIO();
  constexpr int kFinCols = FINCOLS;  // shadows the file-level default inside this kernel
  constexpr int GROUPS = RH_BLOCK / kFinCols;
  __shared__ float red[2 * GROUPS * (kFinCols + 1)];
  const int cl = threadIdx.x % kFinCols;
  const int c = blockIdx.x * kFinCols + cl;
  const int grp = threadIdx.x / kFinCols;
  const bool cok = c < a.C;
  if (MODE == 0 && a.bookkeep && blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.bookkeep != 2) {
      a.saved_ctr[0] = a.rng[1];  // dropout stream of this call
      a.rng[1] += 1;
    }
    if (a.num_batches_tracked != nullptr) a.num_batches_tracked[0] += 1;
  }
  if (MODE == 0) {
    float mean, var;
    // (4-column form = the long lists of B * L-row layers, 2048 chunks = 32 per thread: all of them loaded up front -- past
    // MAXIT the passes are load -> add chains, ~2.5 us per link beside a bandwidth-bound kernel of the other attention unit:
    // 84 us measured in the DIN step for this launch with MAXIT = 16; same summation order)
    chan_combine<GROUPS, kFinCols, (FINCOLS <= 4 ? 32 : 16)>(a, c, cl, grp, cok, red, mean, var);
    if (grp != 0 || !cok) return;
    const float n = (float)a.B;
    const float rstd = rsqrtf(var + a.eps);
    a.stat[c] = mean;
    a.stat[a.C + c] = rstd;
    if (a.affine_out) {
      const float scale = a.gamma[c] * rstd;
      a.stat[4 * a.C + c] = scale;
      a.stat[5 * a.C + c] = fmaf(-mean, scale, a.beta[c]);
    }
    if (a.running_mean != nullptr) {
      const float unbiased = a.B > 1 ? var * (n / (n - 1.f)) : var;
      a.running_mean[c] = fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
      a.running_var[c] = fmaf(a.momentum, unbiased - a.running_var[c], a.running_var[c]);
    }
    return;
  }
  if (a.nscal > 0 && (int)blockIdx.x >= a.col_blocks) {
    // one scalar row: thread-strided sums (8 loads in flight), then the 256 lane sums in lane order -- fixed order
    const float* row = a.scal + (int64_t)((int)blockIdx.x - a.col_blocks) * a.nchunks;
    float v = 0.f;
    int k = threadIdx.x;
    for (; k + 7 * RH_BLOCK < a.nchunks; k += 8 * RH_BLOCK) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = row[k + i * RH_BLOCK];
#pragma unroll
      for (int i = 0; i < 8; ++i) v += t[i];
    }
    for (; k < a.nchunks; k += RH_BLOCK) v += row[k];
    float* rs = red;  // (2 * GROUPS * (kFinCols + 1) >= 256 floats for both instantiations)
    rs[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < RH_WAVE) {
      float w = rs[threadIdx.x];
      for (int q = 1; q < RH_BLOCK / RH_WAVE; ++q) w += rs[q * RH_WAVE + threadIdx.x];
      for (int off = RH_WAVE / 2; off > 0; off >>= 1) w += __shfl_down(w, off, RH_WAVE);
      if (threadIdx.x == 0) a.scal_out[(int)blockIdx.x - a.col_blocks] = w;
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const bool has3 = a.extra != nullptr;
  if (cok) {
    // batches of 8 chunks: 16 independent loads in flight, then the adds in chunk order (the order of the plain loop)
    int k = grp;
    for (; k + 7 * GROUPS < a.nchunks; k += 8 * GROUPS) {
      float t1[8], t2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t1[i] = a.partial[((int64_t)(k + i * GROUPS) * 2 + 0) * a.C + c];
        t2[i] = a.partial[((int64_t)(k + i * GROUPS) * 2 + 1) * a.C + c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s1 += t1[i], s2 += t2[i];
      if (has3) {
        float t3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t3[i] = a.extra[(int64_t)(k + i * GROUPS) * a.C + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) s3 += t3[i];
      }
    }
    for (; k < a.nchunks; k += GROUPS) {
      s1 += a.partial[((int64_t)k * 2 + 0) * a.C + c];
      s2 += a.partial[((int64_t)k * 2 + 1) * a.C + c];
      if (has3) s3 += a.extra[(int64_t)k * a.C + c];
    }
  }
  red[grp * (kFinCols + 1) + cl] = s1;
  red[(GROUPS + grp) * (kFinCols + 1) + cl] = s2;
  __syncthreads();
  if (grp == 0 && cok) {
    s1 = red[cl];
    s2 = red[GROUPS * (kFinCols + 1) + cl];
    for (int g2 = 1; g2 < GROUPS; ++g2) {
      s1 += red[g2 * (kFinCols + 1) + cl];
      s2 += red[(GROUPS + g2) * (kFinCols + 1) + cl];
    }
    a.stat[2 * a.C + c] = s1;  // sum g1        = dbeta
    a.stat[3 * a.C + c] = s2;  // sum g1 * xhat = dgamma
    a.dbeta[c] = s1;
    a.dgamma[c] = s2;
  }
  if (!has3) return;  // (workgroup-uniform)
  __syncthreads();
  red[grp * (kFinCols + 1) + cl] = s3;
  __syncthreads();
  if (grp == 0 && cok) {
    s3 = red[cl];
    for (int g2 = 1; g2 < GROUPS; ++g2) s3 += red[g2 * (kFinCols + 1) + cl];
    a.extra_out[c] = s3;
  }
}

// eval mode of rh_bn_stats_fwd: the folded affine from the running statistics
__global__ __launch_bounds__(RH_BLOCK) void bn_eval_affine_kernel(const BnArgs a) {
  const int c = blockIdx.x * RH_BLOCK + threadIdx.x;
  if (c >= a.C) return;
  const float mean = a.running_mean[c], rstd = rsqrtf(a.running_var[c] + a.eps);
  const float scale = a.gamma[c] * rstd;
  a.stat[c] = mean;
  a.stat[a.C + c] = rstd;
  a.stat[4 * a.C + c] = scale;
  a.stat[5 * a.C + c] = fmaf(-mean, scale, a.beta[c]);
}

// MODE 0 forward apply, MODE 1 backward dx, MODE 2 eval-mode forward (running statistics, no dropout)
template <int MODE>
__global__ __launch_bounds__(RH_BLOCK) void bn_apply_kernel(const BnArgs a) {
  RH_CHAIN_PRIO();
  const int64_t n = (int64_t)a.B * a.C;
  const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t thr = (uint32_t)(a.p_drop * 4294967296.0);
  uint64_t seed = 0, ctr = 0;
  if (MODE != 2 && a.p_drop > 0.f) {
    seed = (uint64_t)a.rng[0];
    ctr = (uint64_t)a.saved_ctr[0];
  }
  const float inv_n = 1.f / (float)a.B;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * RH_BLOCK) {
    const int c = (int)(i % a.C);
    float mean, rstd;
    if (MODE == 2) {
      mean = a.running_mean[c];
      rstd = rsqrtf(a.running_var[c] + a.eps);
    } else {
      mean = a.stat[c];
      rstd = a.stat[a.C + c];
    }
    const float g = a.gamma[c];
    const float xhat = (a.h[i] - mean) * rstd;
    const float bn = fmaf(xhat, g, a.beta[c]);
    const float sl = a.relu == 2 ? a.slope[0] : 0.f;
    if (MODE == 1) {
      float dyv = a.dy[i];
      if (a.p_drop > 0.f) dyv = drop_hash(seed, ctr, (uint64_t)i) >= thr ? dyv * keep_scale : 0.f;
      const float g1 = bn_act_grad(a.relu, sl, bn, dyv);
      const float sg = a.stat[2 * a.C + c], sgx = a.stat[3 * a.C + c];
      a.out[i] = g * rstd * (g1 - sg * inv_n - xhat * (sgx * inv_n));
    } else {
      float y = bn_act(a.relu, sl, bn);
      if (MODE == 0 && a.p_drop > 0.f) y = drop_hash(seed, ctr, (uint64_t)i) >= thr ? y * keep_scale : 0.f;
      a.out[i] = y;
    }
  }
}

// ---- B <= 8192: finalize folded into the apply launch (two launches per direction) ---------------------------------
// 64-row chunks give <= 128 partial rows, few enough for every apply block to re-reduce the partials of ITS 32 columns
// (<= 32 KB from L2) instead of waiting for a finalize launch.  Block = 32 columns x 8 row lanes: 128-byte row segments.
constexpr int kFusedRows = 64;       // rows per partial chunk on this path
constexpr int kFusedMaxChunks = 128;
constexpr int kSlabCols = 32, kSlabLanes = RH_BLOCK / kSlabCols;

template <int MODE>  // 0 forward, 1 backward
__global__ __launch_bounds__(RH_BLOCK) void bn_apply_fin_kernel(const BnArgs a, int rows_per_block) {
  RH_CHAIN_PRIO();
  __shared__ float red[2 * kSlabLanes * (kSlabCols + 1)];
  const int cl = threadIdx.x % kSlabCols, grp = threadIdx.x / kSlabCols;
  const int c = blockIdx.x * kSlabCols + cl;
  const bool cok = c < a.C;
  float s1 = 0.f, s2 = 0.f, mean, rstd;
  if (MODE == 0) {
    float var;
    chan_combine<kSlabLanes, kSlabCols, kFusedMaxChunks / kSlabLanes>(a, c, cl, grp, cok, red, mean, var);
    rstd = rsqrtf(var + a.eps);
    if (cok && blockIdx.y == 0 && grp == 0) {
      const float n = (float)a.B;
      a.stat[c] = mean;
      a.stat[a.C + c] = rstd;
      if (a.running_mean != nullptr) {
        const float unbiased = a.B > 1 ? var * (n / (n - 1.f)) : var;
        a.running_mean[c] = fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
        a.running_var[c] = fmaf(a.momentum, unbiased - a.running_var[c], a.running_var[c]);
      }
    }
  } else {
    {
      constexpr int MAXIT = kFusedMaxChunks / kSlabLanes;  // fixed order: deterministic; loads issued together
      float p1[MAXIT], p2[MAXIT];
#pragma unroll
      for (int i = 0; i < MAXIT; ++i) {
        const int k = grp + i * kSlabLanes;
        const bool ok = cok && k < a.nchunks;
        p1[i] = ok ? a.partial[((int64_t)k * 2 + 0) * a.C + c] : 0.f;
        p2[i] = ok ? a.partial[((int64_t)k * 2 + 1) * a.C + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MAXIT; ++i) {
        s1 += p1[i];
        s2 += p2[i];
      }
    }
    red[grp * (kSlabCols + 1) + cl] = s1;
    red[(kSlabLanes + grp) * (kSlabCols + 1) + cl] = s2;
    __syncthreads();
    s1 = red[cl];
    s2 = red[kSlabLanes * (kSlabCols + 1) + cl];
#pragma unroll
    for (int q = 1; q < kSlabLanes; ++q) {
      s1 += red[q * (kSlabCols + 1) + cl];
      s2 += red[(kSlabLanes + q) * (kSlabCols + 1) + cl];
    }
    if (cok) {
      mean = a.stat[c];
      rstd = a.stat[a.C + c];
      if (blockIdx.y == 0 && grp == 0) {
        a.dbeta[c] = s1;
        a.dgamma[c] = s2;
      }
    }
  }
  const float inv_n = 1.f / (float)a.B;
  const float g = cok ? a.gamma[c] : 0.f, bt = cok ? a.beta[c] : 0.f;
  const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t thr = (uint32_t)(a.p_drop * 4294967296.0);
  uint64_t seed = 0, ctr = 0;
  if (a.p_drop > 0.f) {
    seed = (uint64_t)a.rng[0];
    ctr = (uint64_t)a.saved_ctr[0];
  }
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, a.B);
  const float sg = s1 * inv_n, sgx = s2 * inv_n;
  const float sl = a.relu == 2 ? a.slope[0] : 0.f;
#pragma unroll 4
  for (int r = r0 + grp; cok && r < r1; r += kSlabLanes) {
    const int64_t i = (int64_t)r * a.C + c;
    const float xhat = (a.h[i] - mean) * rstd;
    const float bn = fmaf(xhat, g, bt);
    if (MODE == 1) {
      float dyv = a.dy[i];
      if (a.p_drop > 0.f) dyv = drop_hash(seed, ctr, (uint64_t)i) >= thr ? dyv * keep_scale : 0.f;
      const float g1 = bn_act_grad(a.relu, sl, bn, dyv);
      a.out[i] = g * rstd * (g1 - sg - xhat * sgx);
    } else {
      float y = bn_act(a.relu, sl, bn);
      if (a.p_drop > 0.f) y = drop_hash(seed, ctr, (uint64_t)i) >= thr ? y * keep_scale : 0.f;
      a.out[i] = y;
    }
  }
}

bool fused_path_ok(int B) { return (B + kFusedRows - 1) / kFusedRows <= kFusedMaxChunks; }

// partial_rows > 0: a.partial already holds (sum, M2) per partial_rows-row slab (written by the GEMM in front)
template <int MODE>
void launch_fused(BnArgs a, hipStream_t s, int partial_rows = 0, int pre_chunks = 0) {
  if (pre_chunks > 0) {  // backward: the partial rows exist already (any row partition: they are only summed)
    a.rows_per_chunk = (a.B + pre_chunks - 1) / pre_chunks;
    a.nchunks = pre_chunks;
    a.bookkeep = 0;
  } else if (partial_rows > 0) {
    a.rows_per_chunk = partial_rows;
    a.nchunks = (a.B + partial_rows - 1) / partial_rows;
    a.bookkeep = 0;  // the GEMM that wrote the partials advanced the dropout counter / num_batches_tracked
  } else {
    a.rows_per_chunk = kFusedRows;
    a.nchunks = (a.B + kFusedRows - 1) / kFusedRows;
    a.bookkeep = MODE == 0 ? 1 : 0;
    const int cw = kSlabCols;  // 32 columns x 8 row lanes: C/32 x nchunks blocks, 8 rows per thread
    hipLaunchKernelGGL((bn_partial_kernel<MODE>), dim3((unsigned)((a.C + cw - 1) / cw), (unsigned)a.nchunks),
                       dim3(RH_BLOCK), 0, s, a, cw);
  }
  const int slabs = (a.C + kSlabCols - 1) / kSlabCols;
  int gy = (512 + slabs - 1) / slabs;                 // ~512 blocks in total
  int rpb = (a.B + gy - 1) / gy;
  rpb = (rpb + kSlabLanes - 1) / kSlabLanes * kSlabLanes;
  gy = (a.B + rpb - 1) / rpb;
  hipLaunchKernelGGL((bn_apply_fin_kernel<MODE>), dim3((unsigned)slabs, (unsigned)gy), dim3(RH_BLOCK), 0, s, a, rpb);
}

unsigned apply_grid(int64_t n) {
  int64_t g = (n + RH_BLOCK * 4 - 1) / (RH_BLOCK * 4);
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (unsigned)g;
}

}  // namespace

// rows per partial chunk on the three-launch path: 16 for batch-sized inputs (enough workgroups to fill the chip), growing
// with B so that the finalize launch never has more than ~512 chunks per column to combine (DIN: B * L = 409600 rows)
static void launch_finalize_fwd(const BnArgs& a, hipStream_t s);
static int big_chunk_rows(int B) {
  int r = kRowsPerChunk;
  while ((B + r - 1) / r > 512) r *= 2;
  return r;
}

extern "C" int rh_bn_act_nchunks(int B) { return (B + kRowsPerChunk - 1) / kRowsPerChunk; }  // upper bound (workspace size)

static int bn_fwd_impl(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float p_drop,
                       int training, int64_t* rng, int64_t* saved_ctr, float* partial, int partial_rows, float* stat,
                       float* out, int relu, const float* slope, void* stream);

extern "C" int rh_bn_relu_dropout_fwd(const float* h, int B, int C, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float momentum, float eps, float p_drop, int training, int64_t* rng,
                                      int64_t* saved_ctr, float* partial, int partial_rows, float* stat, float* out,
                                      int relu, void* stream) {
  return bn_fwd_impl(h, B, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, p_drop, training,
                     rng, saved_ctr, partial, partial_rows, stat, out, relu != 0 ? 1 : 0, nullptr, stream);
}

// Linear -> BatchNorm1d -> nn.PReLU() -> Dropout of the two-tower MLPs (reference MLP with activation="prelu",
// examples/matching/run_ml_dssm.py:69-80): rh_bn_relu_dropout_fwd / _bwd with y = bn > 0 ? bn : slope[0] * bn in place of
// the ReLU; the backward also emits, per workgroup of its statistics pass, the partial sums of the slope's gradient
// (slope_partial: rh_bn_prelu_nblocks(B, C) floats, summed by the caller / the step's packing launch).
extern "C" int rh_bn_prelu_dropout_fwd(const float* h, int B, int C, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                       float momentum, float eps, float p_drop, int training, int64_t* rng,
                                       int64_t* saved_ctr, float* partial, int partial_rows, float* stat, float* out,
                                       const float* slope, void* stream) {
  RH_REQUIRE(slope != nullptr, RH_E_BADARG, "rh_bn_prelu_dropout_fwd: slope is null");
  return bn_fwd_impl(h, B, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, p_drop, training,
                     rng, saved_ctr, partial, partial_rows, stat, out, 2, slope, stream);
}

static int bn_fwd_impl(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float p_drop,
                       int training, int64_t* rng, int64_t* saved_ctr, float* partial, int partial_rows, float* stat,
                       float* out, int relu, const float* slope, void* stream) {
  RH_REQUIRE(h && gamma && beta && out, RH_E_BADARG, "rh_bn_relu_dropout_fwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1, RH_E_BADARG, "rh_bn_relu_dropout_fwd: bad shape B=%d C=%d", B, C);
  RH_REQUIRE(p_drop >= 0.f && p_drop < 1.f, RH_E_BADARG, "rh_bn_relu_dropout_fwd: p must be in [0, 1)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnArgs a{};
  a.h = h; a.out = out; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = num_batches_tracked; a.partial = partial; a.stat = stat; a.rng = rng; a.saved_ctr = saved_ctr;
  a.B = B; a.C = C; a.rows_per_chunk = big_chunk_rows(B); a.nchunks = (B + a.rows_per_chunk - 1) / a.rows_per_chunk;
  a.momentum = momentum; a.eps = eps; a.p_drop = p_drop;
  a.training = training;
  a.relu = relu;
  a.slope = slope;
  if (!training) {
    RH_REQUIRE(running_mean && running_var, RH_E_BADARG, "rh_bn_relu_dropout_fwd: eval mode needs running statistics");
    a.p_drop = 0.f;
    hipLaunchKernelGGL((bn_apply_kernel<2>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd(eval)");
    return 0;
  }
  RH_REQUIRE(partial && stat && rng && saved_ctr, RH_E_BADARG, "rh_bn_relu_dropout_fwd: training needs workspaces");
  RH_REQUIRE(partial_rows >= 0, RH_E_BADARG, "rh_bn_relu_dropout_fwd: partial_rows");
  if (partial_rows > 0 ? (B + partial_rows - 1) / partial_rows <= kFusedMaxChunks : fused_path_ok(B)) {
    launch_fused<0>(a, s, partial_rows);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd(fused finalize)");
    return 0;
  }
  a.bookkeep = partial_rows > 0 ? 0 : 1;  // 3-launch path: the finalize launch does the bookkeeping unless the GEMM did
  if (partial_rows > 0) {
    a.rows_per_chunk = partial_rows;
    a.nchunks = (B + partial_rows - 1) / partial_rows;
  } else {
    const int CW = kSlabCols;  // 32 columns x 8 row lanes per block: (C / 32) x nchunks blocks
    const dim3 pg((unsigned)((C + CW - 1) / CW), (unsigned)a.nchunks);
    BnArgs ap = a;
    ap.bookkeep = 0;  // the finalize launch below does it
    hipLaunchKernelGGL((bn_partial_kernel<0>), pg, dim3(RH_BLOCK), 0, s, ap, CW);
  }
  launch_finalize_fwd(a, s);
  hipLaunchKernelGGL((bn_apply_kernel<0>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
  RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd");
  return 0;
}

// Forward finalize (Chan combination of the per-chunk partials): 32 columns x 8 chunk groups per workgroup for the short
// lists of batch-sized inputs; long lists (DIN: 400 - 800 chunks of the 409 600 attention rows) with 8 workgroups walking
// ~100 dependent iterations each took 22 - 29 us -- 4 columns x 64 chunk groups then (C / 4 workgroups).
static void launch_finalize_fwd(const BnArgs& a, hipStream_t s) {
  if (a.nchunks > 256)
    hipLaunchKernelGGL((bn_finalize_kernel<0, 4>), dim3((unsigned)((a.C + 3) / 4)), dim3(RH_BLOCK), 0, s, a);
  else
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3((unsigned)((a.C + kFinCols - 1) / kFinCols)), dim3(RH_BLOCK), 0, s, a);
}

// Statistics only (no apply): for the activations that fold the normalisation into their own pass (csrc/din.hip: Dice).
// stat (6, C): mean, rstd, -, -, scale = gamma * rstd, shift = beta - mean * scale.  eval: running statistics.
extern "C" int rh_bn_stats_fwd(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int training,
                               float* partial, float* stat, void* stream) {
  RH_REQUIRE(h && gamma && beta && stat && B >= 1 && C >= 1, RH_E_BADARG, "rh_bn_stats_fwd: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnArgs a{};
  a.h = h; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = num_batches_tracked; a.partial = partial; a.stat = stat; a.B = B; a.C = C;
  a.momentum = momentum; a.eps = eps; a.training = training; a.affine_out = 1;
  if (!training) {
    RH_REQUIRE(running_mean && running_var, RH_E_BADARG, "rh_bn_stats_fwd: eval mode needs running statistics");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((unsigned)((C + RH_BLOCK - 1) / RH_BLOCK)), dim3(RH_BLOCK), 0, s, a);
    RH_LAUNCH_CHECK("rh_bn_stats_fwd(eval)");
    return 0;
  }
  RH_REQUIRE(partial, RH_E_BADARG, "rh_bn_stats_fwd: training needs the partial workspace");
  a.rows_per_chunk = big_chunk_rows(B);
  a.nchunks = (B + a.rows_per_chunk - 1) / a.rows_per_chunk;
  const dim3 pg((unsigned)((C + kSlabCols - 1) / kSlabCols), (unsigned)a.nchunks);
  hipLaunchKernelGGL((bn_partial_kernel<0>), pg, dim3(RH_BLOCK), 0, s, a, kSlabCols);
  a.bookkeep = 2;  // count the batch; there is no dropout stream to advance
  launch_finalize_fwd(a, s);
  RH_LAUNCH_CHECK("rh_bn_stats_fwd");
  return 0;
}

// The same from per-chunk (sum, M2) partials some producer already emitted (csrc/dinmlp.hip: the epilogue of the fused
// first attention layer): chunk k covers rows [k * rows_per_chunk, (k + 1) * rows_per_chunk) of the B rows.
extern "C" int rh_bn_stats_from_partial(const float* partial, int rows_per_chunk, int B, int C, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, float momentum, float eps, float* stat, void* stream) {
  RH_REQUIRE(partial && gamma && beta && stat && B >= 1 && C >= 1 && rows_per_chunk >= 1, RH_E_BADARG,
             "rh_bn_stats_from_partial: bad arguments");
  BnArgs a{};
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = num_batches_tracked; a.partial = const_cast<float*>(partial); a.stat = stat; a.B = B; a.C = C;
  a.momentum = momentum; a.eps = eps; a.training = 1; a.affine_out = 1;
  a.rows_per_chunk = rows_per_chunk;
  a.nchunks = (B + rows_per_chunk - 1) / rows_per_chunk;
  a.bookkeep = 2;  // count the batch; there is no dropout stream to advance
  launch_finalize_fwd(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_bn_stats_from_partial");
  return 0;
}

// Column sums of (rows, 2, C) partials -> stat rows 2, 3 (sum g, sum g * xhat) and dbeta / dgamma.
// _tail: the same launch also sums the other per-block partials of the statistics pass that wrote `partial` -- a third column
// sum extra (rows, C) -> extra_out (C) (null: none) and nscal scalar rows scal (nscal, rows) -> scal_out (nscal) (0: none) --
// instead of one framework reduction launch each behind it (DIN backward: 3 + 1 launches per attention unit, 5 - 100 us each
// beside the other unit's bandwidth-bound passes).  Fixed summation order.
static int finalize_bwd(const char* who, float* partial, int rows, int C, float* stat, float* dgamma, float* dbeta,
                        const float* extra, float* extra_out, const float* scal, int nscal, float* scal_out, void* stream) {
  RH_REQUIRE(partial && stat && dgamma && dbeta && rows >= 1 && C >= 1, RH_E_BADARG, "%s: bad arguments", who);
  RH_REQUIRE((extra == nullptr) == (extra_out == nullptr), RH_E_BADARG, "%s: extra and extra_out go together", who);
  RH_REQUIRE(nscal >= 0 && nscal <= 8 && (nscal == 0 || (scal && scal_out)), RH_E_BADARG, "%s: %d scalar rows (0..8)", who, nscal);
  BnArgs a{};
  a.partial = partial; a.stat = stat; a.dgamma = dgamma; a.dbeta = dbeta; a.C = C; a.nchunks = rows; a.B = 1;
  a.extra = extra; a.extra_out = extra_out; a.scal = scal; a.scal_out = scal_out; a.nscal = nscal;
  if (rows > 512) {
    a.col_blocks = (C + 3) / 4;
    hipLaunchKernelGGL((bn_finalize_kernel<1, 4>), dim3((unsigned)(a.col_blocks + nscal)), dim3(RH_BLOCK), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
  } else {
    a.col_blocks = (C + kFinCols - 1) / kFinCols;
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3((unsigned)(a.col_blocks + nscal)), dim3(RH_BLOCK), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
  }
  RH_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int rh_bn_finalize_bwd(float* partial, int rows, int C, float* stat, float* dgamma, float* dbeta,
                                  void* stream) {
  return finalize_bwd("rh_bn_finalize_bwd", partial, rows, C, stat, dgamma, dbeta, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int rh_bn_finalize_bwd_tail(float* partial, int rows, int C, float* stat, float* dgamma, float* dbeta,
                                       const float* extra, float* extra_out, const float* scal, int nscal, float* scal_out,
                                       void* stream) {
  return finalize_bwd("rh_bn_finalize_bwd_tail", partial, rows, C, stat, dgamma, dbeta, extra, extra_out, scal, nscal, scal_out,
                      stream);
}

// The backward with the column sums (sum g1, sum g1 * xhat) ALREADY formed by the producer of dy (csrc/linear.hip: the
// output head's backward writes them as nchunks_pre partial rows (2, C) each): one launch (finalize + apply).
extern "C" int rh_bn_relu_dropout_bwd_pre(const float* h, const float* dy, int B, int C, const float* gamma,
                                          const float* beta, float p_drop, const int64_t* rng, const int64_t* saved_ctr,
                                          const float* partial, int nchunks_pre, float* stat, float* dx, float* dgamma,
                                          float* dbeta, int relu, void* stream) {
  RH_REQUIRE(h && dy && gamma && beta && rng && saved_ctr && partial && stat && dx && dgamma && dbeta, RH_E_BADARG,
             "rh_bn_relu_dropout_bwd_pre: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1 && nchunks_pre >= 1 && nchunks_pre <= kFusedMaxChunks, RH_E_UNSUPPORTED,
             "rh_bn_relu_dropout_bwd_pre: %d partial rows (max %d)", nchunks_pre, kFusedMaxChunks);
  BnArgs a{};
  a.h = h; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.partial = const_cast<float*>(partial); a.stat = stat;
  a.dgamma = dgamma; a.dbeta = dbeta; a.rng = const_cast<int64_t*>(rng); a.saved_ctr = const_cast<int64_t*>(saved_ctr);
  a.B = B; a.C = C; a.p_drop = p_drop; a.training = 1; a.relu = relu != 0;
  launch_fused<1>(a, reinterpret_cast<hipStream_t>(stream), 0, nchunks_pre);
  RH_LAUNCH_CHECK("rh_bn_relu_dropout_bwd_pre");
  return 0;
}

static int bn_bwd_impl(const float* h, const float* dy, int B, int C, const float* gamma, const float* beta, float p_drop,
                       const int64_t* rng, const int64_t* saved_ctr, float* partial, float* stat, float* dx, float* dgamma,
                       float* dbeta, int relu, const float* slope, float* slope_partial, void* stream);

extern "C" int rh_bn_relu_dropout_bwd(const float* h, const float* dy, int B, int C, const float* gamma,
                                      const float* beta, float p_drop, const int64_t* rng, const int64_t* saved_ctr,
                                      float* partial, float* stat, float* dx, float* dgamma, float* dbeta, int relu,
                                      void* stream) {
  return bn_bwd_impl(h, dy, B, C, gamma, beta, p_drop, rng, saved_ctr, partial, stat, dx, dgamma, dbeta, relu != 0 ? 1 : 0,
                     nullptr, nullptr, stream);
}

// workgroups of the backward's statistics pass = entries of slope_partial
extern "C" int rh_bn_prelu_nblocks(int B, int C) {
  const int rows = fused_path_ok(B) ? kFusedRows : big_chunk_rows(B);
  return ((C + kSlabCols - 1) / kSlabCols) * ((B + rows - 1) / rows);
}

extern "C" int rh_bn_prelu_dropout_bwd(const float* h, const float* dy, int B, int C, const float* gamma,
                                       const float* beta, float p_drop, const int64_t* rng, const int64_t* saved_ctr,
                                       float* partial, float* stat, float* dx, float* dgamma, float* dbeta,
                                       const float* slope, float* slope_partial, void* stream) {
  RH_REQUIRE(slope && slope_partial, RH_E_BADARG, "rh_bn_prelu_dropout_bwd: null slope pointer");
  return bn_bwd_impl(h, dy, B, C, gamma, beta, p_drop, rng, saved_ctr, partial, stat, dx, dgamma, dbeta, 2, slope,
                     slope_partial, stream);
}

static int bn_bwd_impl(const float* h, const float* dy, int B, int C, const float* gamma, const float* beta, float p_drop,
                       const int64_t* rng, const int64_t* saved_ctr, float* partial, float* stat, float* dx, float* dgamma,
                       float* dbeta, int relu, const float* slope, float* slope_partial, void* stream) {
  RH_REQUIRE(h && dy && gamma && beta && rng && saved_ctr && partial && stat && dx && dgamma && dbeta, RH_E_BADARG,
             "rh_bn_relu_dropout_bwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1, RH_E_BADARG, "rh_bn_relu_dropout_bwd: bad shape B=%d C=%d", B, C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnArgs a{};
  a.h = h; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.partial = partial; a.stat = stat;
  a.dgamma = dgamma; a.dbeta = dbeta; a.rng = const_cast<int64_t*>(rng); a.saved_ctr = const_cast<int64_t*>(saved_ctr);
  a.B = B; a.C = C; a.rows_per_chunk = big_chunk_rows(B); a.nchunks = (B + a.rows_per_chunk - 1) / a.rows_per_chunk;
  a.p_drop = p_drop; a.training = 1;
  a.relu = relu;
  a.slope = slope;
  a.slope_partial = slope_partial;
  if (fused_path_ok(B)) {
    launch_fused<1>(a, s);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_bwd(fused finalize)");
    return 0;
  }
  const int CW = kSlabCols;  // 32 columns x 8 row lanes per block: (C / 32) x nchunks blocks
  const dim3 pg((unsigned)((C + CW - 1) / CW), (unsigned)a.nchunks);
  hipLaunchKernelGGL((bn_partial_kernel<1>), pg, dim3(RH_BLOCK), 0, s, a, CW);
  hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3((unsigned)((C + kFinCols - 1) / kFinCols)), dim3(RH_BLOCK), 0, s, a);
  hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
  RH_LAUNCH_CHECK("rh_bn_relu_dropout_bwd");
  return 0;
}
