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
  int bookkeep;        // partial kernel also advances the dropout counter / num_batches_tracked (fused-finalize path)
  float momentum, eps, p_drop;
  int training;
};

static __device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed + ctr * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// thread (rsub, c): column c of the tile, rows rsub, rsub+RS, ... of the chunk.  MODE 0: sums of (x-s), (x-s)^2 with
// s = h[0, c].  MODE 1 (backward): sums of g1 and g1*xhat.
template <int MODE>
__global__ __launch_bounds__(RH_BLOCK) void bn_partial_kernel(const BnArgs a, int CW) {
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
  float s1 = 0.f, s2 = 0.f;
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
#pragma unroll 4
      for (int r = r0 + rsub; r < r1; r += RS) {
        const int64_t i = (int64_t)r * a.C + c;
        const float xhat = (a.h[i] - mean) * rstd;
        const float bn = fmaf(xhat, g, bt);
        float g1 = bn > 0.f ? a.dy[i] : 0.f;
        if (a.p_drop > 0.f) g1 = drop_hash(seed, ctr, (uint64_t)i) >= thr ? g1 * keep_scale : 0.f;
        s1 += g1;
        s2 = fmaf(g1, xhat, s2);
      }
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
    a.partial[((int64_t)blockIdx.y * 2 + 0) * a.C + c] = s1;
    a.partial[((int64_t)blockIdx.y * 2 + 1) * a.C + c] = s2;
  }
}

template <int MODE>
__global__ __launch_bounds__(RH_BLOCK) void bn_finalize_kernel(const BnArgs a) {
  __shared__ float red[2][RH_BLOCK];
  constexpr int GROUPS = RH_BLOCK / kFinCols;
  const int c = blockIdx.x * kFinCols + threadIdx.x % kFinCols;
  const int grp = threadIdx.x / kFinCols;
  if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    a.saved_ctr[0] = a.rng[1];  // dropout stream of this call
    a.rng[1] += 1;
    if (a.num_batches_tracked != nullptr) a.num_batches_tracked[0] += 1;
  }
  float s1 = 0.f, s2 = 0.f;
  if (c < a.C) {
    // chunk partials are summed in a fixed order (group-strided, then across groups): deterministic
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
    int k = grp;
    for (; k + 3 * GROUPS < a.nchunks; k += 4 * GROUPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t1[u] += a.partial[((int64_t)(k + u * GROUPS) * 2 + 0) * a.C + c];
        t2[u] += a.partial[((int64_t)(k + u * GROUPS) * 2 + 1) * a.C + c];
      }
    }
    for (; k < a.nchunks; k += GROUPS) {
      t1[0] += a.partial[((int64_t)k * 2 + 0) * a.C + c];
      t2[0] += a.partial[((int64_t)k * 2 + 1) * a.C + c];
    }
    s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
    s2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (grp != 0 || c >= a.C) return;
  for (int g2 = 1; g2 < GROUPS; ++g2) {
    s1 += red[0][g2 * kFinCols + threadIdx.x];
    s2 += red[1][g2 * kFinCols + threadIdx.x];
  }
  if (MODE == 0) {
    const float n = (float)a.B;
    const float m1 = s1 / n;
    const float mean = a.h[c] + m1;
    float var = s2 / n - m1 * m1;  // biased variance (normalisation)
    var = var > 0.f ? var : 0.f;
    a.stat[c] = mean;
    a.stat[a.C + c] = rsqrtf(var + a.eps);
    if (a.running_mean != nullptr) {
      const float unbiased = a.B > 1 ? var * (n / (n - 1.f)) : var;
      a.running_mean[c] = fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
      a.running_var[c] = fmaf(a.momentum, unbiased - a.running_var[c], a.running_var[c]);
    }
  } else {
    a.stat[2 * a.C + c] = s1;  // sum g1        = dbeta
    a.stat[3 * a.C + c] = s2;  // sum g1 * xhat = dgamma
    a.dbeta[c] = s1;
    a.dgamma[c] = s2;
  }
}

// MODE 0 forward apply, MODE 1 backward dx, MODE 2 eval-mode forward (running statistics, no dropout)
template <int MODE>
__global__ __launch_bounds__(RH_BLOCK) void bn_apply_kernel(const BnArgs a) {
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
    if (MODE == 1) {
      float g1 = bn > 0.f ? a.dy[i] : 0.f;
      if (a.p_drop > 0.f) g1 = drop_hash(seed, ctr, (uint64_t)i) >= thr ? g1 * keep_scale : 0.f;
      const float sg = a.stat[2 * a.C + c], sgx = a.stat[3 * a.C + c];
      a.out[i] = g * rstd * (g1 - sg * inv_n - xhat * (sgx * inv_n));
    } else {
      float y = bn > 0.f ? bn : 0.f;
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
  __shared__ float red[2][kSlabLanes][kSlabCols + 1];
  const int cl = threadIdx.x % kSlabCols, grp = threadIdx.x / kSlabCols;
  const int c = blockIdx.x * kSlabCols + cl;
  const bool cok = c < a.C;
  float s1 = 0.f, s2 = 0.f;
  if (cok) {
    for (int k = grp; k < a.nchunks; k += kSlabLanes) {  // fixed order: deterministic
      s1 += a.partial[((int64_t)k * 2 + 0) * a.C + c];
      s2 += a.partial[((int64_t)k * 2 + 1) * a.C + c];
    }
  }
  red[0][grp][cl] = s1;
  red[1][grp][cl] = s2;
  __syncthreads();
  s1 = red[0][0][cl];
  s2 = red[1][0][cl];
#pragma unroll
  for (int q = 1; q < kSlabLanes; ++q) {
    s1 += red[0][q][cl];
    s2 += red[1][q][cl];
  }
  if (!cok) return;
  const float n = (float)a.B, inv_n = 1.f / n;
  const float g = a.gamma[c], bt = a.beta[c];
  float mean, rstd;
  if (MODE == 0) {
    const float m1 = s1 / n;
    mean = a.h[c] + m1;
    float var = s2 / n - m1 * m1;
    var = var > 0.f ? var : 0.f;
    rstd = rsqrtf(var + a.eps);
    if (blockIdx.y == 0 && grp == 0) {
      a.stat[c] = mean;
      a.stat[a.C + c] = rstd;
      if (a.running_mean != nullptr) {
        const float unbiased = a.B > 1 ? var * (n / (n - 1.f)) : var;
        a.running_mean[c] = fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
        a.running_var[c] = fmaf(a.momentum, unbiased - a.running_var[c], a.running_var[c]);
      }
    }
  } else {
    mean = a.stat[c];
    rstd = a.stat[a.C + c];
    if (blockIdx.y == 0 && grp == 0) {
      a.dbeta[c] = s1;
      a.dgamma[c] = s2;
    }
  }
  const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t thr = (uint32_t)(a.p_drop * 4294967296.0);
  uint64_t seed = 0, ctr = 0;
  if (a.p_drop > 0.f) {
    seed = (uint64_t)a.rng[0];
    ctr = (uint64_t)a.saved_ctr[0];
  }
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, a.B);
  const float sg = s1 * inv_n, sgx = s2 * inv_n;
#pragma unroll 4
  for (int r = r0 + grp; r < r1; r += kSlabLanes) {
    const int64_t i = (int64_t)r * a.C + c;
    const float xhat = (a.h[i] - mean) * rstd;
    const float bn = fmaf(xhat, g, bt);
    if (MODE == 1) {
      float g1 = bn > 0.f ? a.dy[i] : 0.f;
      if (a.p_drop > 0.f) g1 = drop_hash(seed, ctr, (uint64_t)i) >= thr ? g1 * keep_scale : 0.f;
      a.out[i] = g * rstd * (g1 - sg - xhat * sgx);
    } else {
      float y = bn > 0.f ? bn : 0.f;
      if (a.p_drop > 0.f) y = drop_hash(seed, ctr, (uint64_t)i) >= thr ? y * keep_scale : 0.f;
      a.out[i] = y;
    }
  }
}

bool fused_path_ok(int B) { return (B + kFusedRows - 1) / kFusedRows <= kFusedMaxChunks; }

template <int MODE>
void launch_fused(BnArgs a, hipStream_t s) {
  a.rows_per_chunk = kFusedRows;
  a.nchunks = (a.B + kFusedRows - 1) / kFusedRows;
  a.bookkeep = MODE == 0 ? 1 : 0;
  const int cw = kSlabCols;  // 32 columns x 8 row lanes: C/32 x nchunks blocks, 8 rows per thread
  hipLaunchKernelGGL((bn_partial_kernel<MODE>), dim3((unsigned)((a.C + cw - 1) / cw), (unsigned)a.nchunks), dim3(RH_BLOCK), 0,
                     s, a, cw);
  const int slabs = (a.C + kSlabCols - 1) / kSlabCols;
  int gy = (512 + slabs - 1) / slabs;                 // ~512 blocks in total
  int rpb = (a.B + gy - 1) / gy;
  rpb = (rpb + kSlabLanes - 1) / kSlabLanes * kSlabLanes;
  gy = (a.B + rpb - 1) / rpb;
  hipLaunchKernelGGL((bn_apply_fin_kernel<MODE>), dim3((unsigned)slabs, (unsigned)gy), dim3(RH_BLOCK), 0, s, a, rpb);
}

int col_width(int C) {
  int cw = 32;
  while (cw < C && cw < RH_BLOCK) cw *= 2;
  return cw;
}

unsigned apply_grid(int64_t n) {
  int64_t g = (n + RH_BLOCK * 4 - 1) / (RH_BLOCK * 4);
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (unsigned)g;
}

}  // namespace

extern "C" int rh_bn_act_nchunks(int B) { return (B + kRowsPerChunk - 1) / kRowsPerChunk; }

extern "C" int rh_bn_relu_dropout_fwd(const float* h, int B, int C, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float momentum, float eps, float p_drop, int training, int64_t* rng,
                                      int64_t* saved_ctr, float* partial, float* stat, float* out, void* stream) {
  RH_REQUIRE(h && gamma && beta && out, RH_E_BADARG, "rh_bn_relu_dropout_fwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1, RH_E_BADARG, "rh_bn_relu_dropout_fwd: bad shape B=%d C=%d", B, C);
  RH_REQUIRE(p_drop >= 0.f && p_drop < 1.f, RH_E_BADARG, "rh_bn_relu_dropout_fwd: p must be in [0, 1)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnArgs a{};
  a.h = h; a.out = out; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = num_batches_tracked; a.partial = partial; a.stat = stat; a.rng = rng; a.saved_ctr = saved_ctr;
  a.B = B; a.C = C; a.nchunks = rh_bn_act_nchunks(B); a.rows_per_chunk = kRowsPerChunk; a.momentum = momentum; a.eps = eps; a.p_drop = p_drop;
  a.training = training;
  if (!training) {
    RH_REQUIRE(running_mean && running_var, RH_E_BADARG, "rh_bn_relu_dropout_fwd: eval mode needs running statistics");
    a.p_drop = 0.f;
    hipLaunchKernelGGL((bn_apply_kernel<2>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd(eval)");
    return 0;
  }
  RH_REQUIRE(partial && stat && rng && saved_ctr, RH_E_BADARG, "rh_bn_relu_dropout_fwd: training needs workspaces");
  if (fused_path_ok(B)) {
    launch_fused<0>(a, s);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd(fused finalize)");
    return 0;
  }
  const int CW = col_width(C);
  const dim3 pg((unsigned)((C + CW - 1) / CW), (unsigned)a.nchunks);
  hipLaunchKernelGGL((bn_partial_kernel<0>), pg, dim3(RH_BLOCK), 0, s, a, CW);
  hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3((unsigned)((C + kFinCols - 1) / kFinCols)), dim3(RH_BLOCK), 0, s, a);
  hipLaunchKernelGGL((bn_apply_kernel<0>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
  RH_LAUNCH_CHECK("rh_bn_relu_dropout_fwd");
  return 0;
}

extern "C" int rh_bn_relu_dropout_bwd(const float* h, const float* dy, int B, int C, const float* gamma,
                                      const float* beta, float p_drop, const int64_t* rng, const int64_t* saved_ctr,
                                      float* partial, float* stat, float* dx, float* dgamma, float* dbeta,
                                      void* stream) {
  RH_REQUIRE(h && dy && gamma && beta && rng && saved_ctr && partial && stat && dx && dgamma && dbeta, RH_E_BADARG,
             "rh_bn_relu_dropout_bwd: null pointer");
  RH_REQUIRE(B >= 1 && C >= 1, RH_E_BADARG, "rh_bn_relu_dropout_bwd: bad shape B=%d C=%d", B, C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnArgs a{};
  a.h = h; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.partial = partial; a.stat = stat;
  a.dgamma = dgamma; a.dbeta = dbeta; a.rng = const_cast<int64_t*>(rng); a.saved_ctr = const_cast<int64_t*>(saved_ctr);
  a.B = B; a.C = C; a.nchunks = rh_bn_act_nchunks(B); a.rows_per_chunk = kRowsPerChunk; a.p_drop = p_drop; a.training = 1;
  if (fused_path_ok(B)) {
    launch_fused<1>(a, s);
    RH_LAUNCH_CHECK("rh_bn_relu_dropout_bwd(fused finalize)");
    return 0;
  }
  const int CW = col_width(C);
  const dim3 pg((unsigned)((C + CW - 1) / CW), (unsigned)a.nchunks);
  hipLaunchKernelGGL((bn_partial_kernel<1>), pg, dim3(RH_BLOCK), 0, s, a, CW);
  hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3((unsigned)((C + kFinCols - 1) / kFinCols)), dim3(RH_BLOCK), 0, s, a);
  hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(apply_grid((int64_t)B * C)), dim3(RH_BLOCK), 0, s, a);
  RH_LAUNCH_CHECK("rh_bn_relu_dropout_bwd");
  return 0;
}
