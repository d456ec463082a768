// The MLP's GEMM-shaped and head-shaped pieces that the library does badly at CTR batch sizes.
//
// Reference:
//   MLP (Linear -> BN -> act -> Dropout)* -> Linear(., 1)   torch_rechub/basic/layers.py:254-292
//   DeepFM / WideDeep head: sigmoid(y_linear + y_fm + y_deep)  models/ranking/deepfm.py:39-43, widedeep.py:35-39
//   CTRTrainer loss: torch.nn.BCELoss()(y_pred, y)             trainers/ctr_trainer.py:62, :93-95
//
// 1. linear_wgrad: dW = g^T x (N_out x K_in, reduction over the batch B) and db = colsum(g).  At B = 4096 the output is
//    tiny (256 x 429) and the reduction long, so the library's single-pass kernels leave most CUs idle (26 us each for
//    two of them in the round-1 profile).  Here: split over the batch, f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 ==
//    fmaf chain), operands straight from global memory -- for dW both fragments are batch-major, so lane (i, k) of the
//    A/B fragment reads g[b0+k][n0+i] / x[b0+k][k0+i]: 128-byte coalesced rows, no LDS staging -- partial tiles to a
//    workspace, and a second small launch sums the partials in split order (deterministic).  db falls out
//    of the A fragments for free.  Roofline: f32 MFMA (157 TF).
// 2. head: y = sigmoid(h . w + b + e0 + e1), the (., 1) output layer fused with the wide/FM terms and the sigmoid; the
//    backward gives g_z, g_h, g_w, g_b in ONE launch (three GEMMs with a dimension of 1, two adds, sigmoid and their
//    backward kernels in the reference).  Roofline: HBM (h read once per direction).
// 3. bce: mean binary cross entropy and its backward, log clamped at -100 like torch.nn.BCELoss.
#include "common.h"
#include "wgrad_body.h"

namespace {

using namespace rh_wgrad;

// Two builds of the same body: the default (the compiler takes 78 VGPRs + 128 AGPRs: two wavefronts per SIMD, what a
// B = 4096 launch of ~500 workgroups fills anyway) and one held to 128 registers = four wavefronts per SIMD for the long
// reductions, where the latency of the operand loads is hidden by the other wavefronts (DIN, B * L = 409 600 rows).
__global__ __launch_bounds__(RH_BLOCK) void linear_wgrad_kernel(const WgradArgs a) {
  extern __shared__ float red[];  // kWaves * kPartStride floats
  linear_wgrad_body<false>(a, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

__global__ __launch_bounds__(RH_BLOCK) void linear_wgrad_group_kernel(const WgradGroupArgs ga) {
  extern __shared__ float red[];
  linear_wgrad_group_body<false>(ga, red, (int)blockIdx.x);
}

__global__ __launch_bounds__(RH_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void linear_wgrad_long_kernel(
    const WgradArgs a) {
  extern __shared__ float red[];  // kPartStride floats: the block's tile + its db slice
  linear_wgrad_body<true>(a, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// (the grouped launch in the 128-register build: RH_TUNE_WGRAD_SHORT_FORM = 1, the default)
__global__ __launch_bounds__(RH_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void linear_wgrad_group_long_kernel(
    const WgradGroupArgs ga) {
  extern __shared__ float red[];
  linear_wgrad_group_body<true>(ga, red, (int)blockIdx.x);
}

// Long reductions whose whole output is at most kRowsMaxTiles tiles (round 6; DIN's attention MLP: 409 600 rows, (128, 256) and
// (256, 64)): ONE workgroup per row split computes the WHOLE (N, K) slab -- wavefront w owns tile (w / tiles_k, w % tiles_k) and
// walks every row pair of the split by itself.  In the tile-per-workgroup form above every 64-column slice of g is read once per
// K tile and every slice of x once per N tile, by workgroups that the dispatcher deals to eight different XCDs: rocprofv3
// FETCH_SIZE 1 678 MB for the (128, 256) gradient whose operands are 629 MB, 839 for the (256, 64) one (524).  Here the slices a
// workgroup's wavefronts share come out of its own CU's vector cache / the XCD's L2.  No LDS, no cross-wavefront reduction (each
// output element has one owner); per element the sum runs over the split's rows in order -- another order than the form above
// (four wavefronts' interleaved rows added up afterwards), the same for every launch of this form.
constexpr int kRowsMaxTiles = 8;
// TN x TK: 64 x 64 tiles per WAVEFRONT (1 x 1, 1 x 2 or 2 x 1).  Both forms of the weight gradient load one dword per MFMA
// operand and are bound by the CU's vector-cache bandwidth (55 % of the f32 MFMA peak alone); two tiles per wavefront reuse the
// shared operand from registers: 6 loads per 8 MFMAs instead of 4 per 4.
template <int TN, int TK>
__global__ __launch_bounds__(RH_WAVE * kRowsMaxTiles) void linear_wgrad_rows_kernel(const WgradArgs a, const int groups_k) {
  RH_CHAIN_PRIO();
  constexpr int NA = 2 * TN, NB = 2 * TK;  // 32-column operand slices per wavefront
  const int lane = threadIdx.x % RH_WAVE, wave = threadIdx.x / RH_WAVE;
  const int half = lane >> 5, c = lane & 31;
  const int s = (int)blockIdx.x;
  const int n0 = (wave / groups_k) * (kTile * TN), k0 = (wave % groups_k) * (kTile * TK);
  const int b_lo = s * a.rows_per_split;
  const int b_hi = min(a.B, b_lo + a.rows_per_split);
  // Columns past N / K are clamped to column 0: their products land in rows / columns that are never stored
  const float* ga[NA];
  const float* xb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) ga[i] = a.g + (n0 + 32 * i + c < a.N ? n0 + 32 * i + c : 0) + (int64_t)half * a.ldg;
#pragma unroll
  for (int j = 0; j < NB; ++j) xb[j] = a.x + (k0 + 32 * j + c < a.K ? k0 + 32 * j + c : 0) + (int64_t)half * a.ldx;
  v16f acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = v16f{};
  float bs[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) bs[i] = 0.f;
  constexpr int U = (TN * TK == 1) ? 8 : (TK == 2 ? 6 : 4);  // row pairs in flight, fetched one iteration ahead
  float fa[U][NA], fb[U][NB], qa[U][NA], qb[U][NB];
  auto fetch = [&](int p, float (*A)[NA], float (*B)[NB]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = p + 2 * u;
#pragma unroll
      for (int i = 0; i < NA; ++i) A[u][i] = gload<float>(ga[i] + r * a.ldg);
#pragma unroll
      for (int j = 0; j < NB; ++j) B[u][j] = gload<float>(xb[j] + r * a.ldx);
    }
  };
  auto issue = [&](float (*A)[NA], float (*B)[NB]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u][i], B[u][j], acc[i][j], 0, 0, 0);
        bs[i] += A[u][i];
      }
    }
  };
  constexpr int kStep = 2 * U;
  int p = b_lo;
  const int last_full = b_hi - kStep;  // p <= last_full: rows p .. p + kStep - 1 exist
  if (p <= last_full) {
    fetch(p, fa, fb);
    for (; p + kStep <= last_full; p += kStep) {
      fetch(p + kStep, qa, qb);
      issue(fa, fb);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) fa[u][i] = qa[u][i];
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[u][j] = qb[u][j];
      }
    }
    issue(fa, fb);
    p += kStep;
  }
  for (; p < b_hi; p += 2) {  // ragged end, guarded per row
    const bool ok = p + half < b_hi;
    const int64_t r = ok ? p : b_lo - half;
    const float m = ok ? 1.f : 0.f;
    float ta[NA], tb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ta[i] = gload<float>(ga[i] + r * a.ldg) * m;
#pragma unroll
    for (int j = 0; j < NB; ++j) tb[j] = gload<float>(xb[j] + r * a.ldx) * m;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[i], tb[j], acc[i][j], 0, 0, 0);
      bs[i] += ta[i];
    }
  }
  float* outW = a.direct ? a.dW : a.partial + (int64_t)s * a.N * a.K;
  float* outB = a.direct ? a.db : a.partial + (int64_t)a.S * a.N * a.K + (int64_t)s * a.N;
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int k = k0 + 32 * j + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < a.N && k < a.K) outW[(int64_t)n * a.K + k] = acc[i][j][r];
      }
    }
    const float b2 = bs[i] + __shfl_xor(bs[i], 32);
    if (outB != nullptr && k0 == 0 && half == 0 && n0 + 32 * i + c < a.N) outB[n0 + 32 * i + c] = b2;
  }
}

// Second launch of the split weight gradient: sums the S partial tiles in split order (deterministic).  An in-kernel
// "last block reduces" election needs a device-scope fence per block, which on this 8-XCD part writes back and
// invalidates the XCD's whole L2 (measured: ~140 us for 500 blocks) -- a 3 us launch is the cheaper barrier.
__global__ __launch_bounds__(RH_BLOCK) void wgrad_reduce_kernel(const WgradArgs a) {
  RH_CHAIN_PRIO();
  const int64_t nk = (int64_t)a.N * a.K;
  const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
  const bool is_b = i >= nk;
  if (i >= nk + a.N || (is_b && a.db == nullptr)) return;
  const int64_t stride = is_b ? a.N : nk;
  const float* p0 = is_b ? a.partial + (int64_t)a.S * nk + (i - nk) : a.partial + i;
  float v = 0.f;
  int q = 0;
  for (; q + 4 <= a.S; q += 4) {
    const float t0 = p0[(q + 0) * stride], t1 = p0[(q + 1) * stride];
    const float t2 = p0[(q + 2) * stride], t3 = p0[(q + 3) * stride];
    v = (((v + t0) + t1) + t2) + t3;
  }
  for (; q < a.S; ++q) v += p0[q * stride];
  if (is_b) a.db[i - nk] = v;
  else a.dW[i] = v;
}

// Workgroups a launch aims for.  Short batches (B = 4096): 512, two per CU -- every further split is another (N, K) slab
// written and summed.  Long reductions (DIN's B * L = 409 600 rows): g_long_blocks, so that every SIMD holds enough
// wavefronts to cover the latency of the operand loads (tuning knob RH_TUNE_WGRAD_BLOCKS).
int g_long_blocks = 1024;   // negative: -value workgroups with the default build of the kernel (A/B)
constexpr int kLongRows = 32768;
// RH_TUNE_WGRAD_SHORT_FORM: build used for batch-sized reductions (B < 32768).  0: the round-1 form (78 VGPRs + 128 AGPRs,
// four LDS tiles); 1: the 128-register, one-LDS-tile build of the long reductions.  Beside the optimizer's resident sweep
// (2 wavefronts of 112 registers per SIMD) only ONE wavefront of the 206-register build fits on a SIMD where the launch's
// ~500 workgroups want two: measured 28.5 us alone, 39.5 us beside the sweep (round 4).  Default 1 since the relaxed join
// (optim.py) took the sweep off the step's critical cycle: 0.2622 -> 0.2580 ms per step (same box, two rounds); while the
// sweep's path was the longer one the two builds tied.
int g_short_form = 1;
// RH_TUNE_WGRAD_ROWS_FORM: long reductions of at most kRowsMaxTiles tiles as ONE workgroup per row split (linear_wgrad_rows_kernel);
// value = the workgroups aimed for (default 512: two 512-thread workgroups per CU), 0 = the tile-per-workgroup form.
int g_rows_form = 512;
int g_rows_pair = 1;  // (RH_TUNE_WGRAD_ROWS_FORM given negative: -value workgroups, one tile per wavefront)

static bool wgrad_rows_form(int B, int N, int K) {
  const int tiles = ((N + kTile - 1) / kTile) * ((K + kTile - 1) / kTile);
  return g_rows_form > 0 && g_long_blocks > 0 && B >= kLongRows && tiles >= 2 && tiles <= kRowsMaxTiles;
}

void wgrad_plan(int B, int N, int K, int* tiles_n, int* tiles_k, int* S, int* rps) {
  *tiles_n = (N + kTile - 1) / kTile;
  *tiles_k = (K + kTile - 1) / kTile;
  const int tiles = *tiles_n * *tiles_k;
  if (wgrad_rows_form(B, N, K)) {
    // g_rows_form workgroups of 8 wavefronts = 4 x g_rows_form wavefronts (8 per CU at the default, two per SIMD: what the
    // paired-tile builds' ~220 registers allow); a slab of fewer wavefronts is split further to keep that many in flight
    // ((256, 64) as two wavefronts of 128 x 64: 1024 splits -- with 512 its launch took 228 us alone for 160)
    const bool pair = g_rows_pair && (*tiles_k % 2 == 0 || *tiles_n % 2 == 0);
    const int waves = pair ? tiles / 2 : tiles;
    int s = g_rows_form * kRowsMaxTiles / (2 * waves);
    if (s < g_rows_form) s = g_rows_form;
    const int max_s = (B + 255) / 256;
    if (s > max_s) s = max_s;
    int r = (B + s - 1) / s;
    r = (r + 15) / 16 * 16;
    *rps = r;
    *S = (B + r - 1) / r;
    return;
  }
  int s = (B >= kLongRows ? abs(g_long_blocks) : 512) / tiles;
  const int max_s = (B + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int r = (B + s - 1) / s;
  r = (r + 7) / 8 * 8;
  *rps = r;
  *S = (B + r - 1) / r;
}

// ---------------------------------------------------------------------------------------------
constexpr int kHeadLanes = 16;                       // lanes per row
constexpr int kHeadRows = RH_BLOCK / kHeadLanes;     // rows per block pass
constexpr int kHeadMaxV4 = 16;                       // float4 per lane -> K <= 1024

__device__ __forceinline__ float group_sum16(float v) {
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

__global__ __launch_bounds__(RH_BLOCK) void head_fwd_kernel(const float* __restrict__ h, int64_t ldh,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ e0, const float* __restrict__ e1,
                                                            int B, int K, float* __restrict__ y,
                                                            const float* __restrict__ t, float* __restrict__ loss_partial) {
  RH_CHAIN_PRIO();
  __shared__ float lred[kHeadRows];
  const int sub = threadIdx.x % kHeadLanes, grp = threadIdx.x / kHeadLanes;
  const int nv = K / 4;
  const float b0 = bias ? bias[0] : 0.f;
  float lacc = 0.f;  // this lane group's BCE terms (t != null): -(t log y + (1 - t) log(1 - y)), logs clamped at -100
  for (int64_t row = (int64_t)blockIdx.x * kHeadRows + grp; row < B; row += (int64_t)gridDim.x * kHeadRows) {
    float acc = 0.f;
    for (int v = sub; v < nv; v += kHeadLanes) {
      const float4 hv = gload<float4>(h + row * ldh + 4 * v);
      const float4 wv = gload<float4>(w + 4 * v);
      acc = fmaf(hv.x, wv.x, acc);
      acc = fmaf(hv.y, wv.y, acc);
      acc = fmaf(hv.z, wv.z, acc);
      acc = fmaf(hv.w, wv.w, acc);
    }
    if (sub < K - 4 * nv) acc = fmaf(h[row * ldh + 4 * nv + sub], w[4 * nv + sub], acc);  // K % 4 tail columns
    acc = group_sum16(acc);
    if (sub == 0) {
      float z = acc + b0;
      if (e0) z += e0[row];
      if (e1) z += e1[row];
      const float yv = 1.f / (1.f + __expf(-z));
      y[row] = yv;
      if (t) {
        const float tv = t[row];
        lacc -= tv * fmaxf(logf(yv), -100.f) + (1.f - tv) * fmaxf(log1pf(-yv), -100.f);
      }
    }
  }
  if (loss_partial) {  // fixed order: row groups of the block, then the blocks (rh_step_scalars)
    if (sub == 0) lred[grp] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < kHeadRows; ++r) v += lred[r];
      loss_partial[blockIdx.x] = v;
    }
  }
}

// The output head on the PRE-BatchNorm activations of the last hidden layer (fused MLP chain, round 4):
//   y = sigmoid(dropout(relu(batch_norm(z))) . w + b + e0 + e1)
// Every workgroup combines the per-slab (sum, M2) statistics the GEMM that produced z left behind (Chan's formula, slab
// order, one column per thread, all loads in flight) into mean / rstd in LDS, then walks its rows as head_fwd_kernel does,
// applying normalisation, ReLU and the dropout hash on the way into the dot product: the BatchNorm + ReLU + Dropout pass
// over the last hidden layer and its output tensor are gone.  Workgroup 0 writes mean / rstd (the backward reads them) and
// the running statistics.  Same per-element arithmetic as bn_apply_fin_kernel<0> (csrc/mlp.hip) followed by head_fwd_kernel.
struct HeadBnFwdArgs {
  const float* z;
  int64_t ldz;
  const float* stats;  // (ceil(B / rows), 2, K)
  int rows;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  float momentum, eps, p;
  const int64_t *rng, *ctr;
  float* stat_out;  // (>= 2, K)
  const float *w, *bias, *e0, *e1;
  int B, K;
  float* y;
  const float* t;
  float* loss_partial;
};
constexpr int kHeadBnMaxK = 256;
constexpr int kHeadBnMaxSlabs = 32;  // slabs per thread and round of loads (common.h, rh_combine_slabs)

__global__ __launch_bounds__(RH_BLOCK) void head_bnact_fwd_kernel(const HeadBnFwdArgs a) {
  RH_CHAIN_PRIO();
  __shared__ float lred[kHeadRows];
  __shared__ float cst[4 * kHeadBnMaxK];  // mean, rstd, gamma, beta
  const int K = a.K, B = a.B;
  {
    __shared__ float red[2 * RH_BLOCK];
    const int nslab = (B + a.rows - 1) / a.rows;
    rh_combine_slabs<RH_BLOCK, kHeadBnMaxSlabs>(a.stats, nslab, a.rows, B, K, red, (int)threadIdx.x,
                                                [&](int c, float mean, float var) {
      const float rstd = rsqrtf(var + a.eps);
      cst[c] = mean;
      cst[kHeadBnMaxK + c] = rstd;
      cst[2 * kHeadBnMaxK + c] = a.gamma[c];
      cst[3 * kHeadBnMaxK + c] = a.beta[c];
      if (blockIdx.x == 0) {
        a.stat_out[c] = mean;
        a.stat_out[K + c] = rstd;
        if (a.running_mean != nullptr) {
          const float n = (float)B;
          const float unbiased = B > 1 ? var * (n / (n - 1.f)) : var;
          a.running_mean[c] = fmaf(a.momentum, mean - a.running_mean[c], a.running_mean[c]);
          a.running_var[c] = fmaf(a.momentum, unbiased - a.running_var[c], a.running_var[c]);
        }
      }
    });
  }
  float keep_scale = 1.f;
  uint32_t thr = 0;
  uint64_t seed = 0, ctr = 0;
  if (a.p > 0.f) {
    keep_scale = 1.f / (1.f - a.p);
    thr = (uint32_t)(a.p * 4294967296.0);
    seed = (uint64_t)a.rng[0];
    ctr = (uint64_t)a.ctr[0];
  }
  const int sub = threadIdx.x % kHeadLanes, grp = threadIdx.x / kHeadLanes;
  const int nv = K / 4;
  const float b0 = a.bias ? a.bias[0] : 0.f;
  float lacc = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kHeadRows + grp; row < B; row += (int64_t)gridDim.x * kHeadRows) {
    float acc = 0.f;
    for (int v = sub; v < nv; v += kHeadLanes) {
      const float4 zv = gload<float4>(a.z + row * a.ldz + 4 * v);
      const float4 wv = gload<float4>(a.w + 4 * v);
      const float4 mean = *reinterpret_cast<const float4*>(cst + 4 * v);
      const float4 rstd = *reinterpret_cast<const float4*>(cst + kHeadBnMaxK + 4 * v);
      const float4 gam = *reinterpret_cast<const float4*>(cst + 2 * kHeadBnMaxK + 4 * v);
      const float4 bet = *reinterpret_cast<const float4*>(cst + 3 * kHeadBnMaxK + 4 * v);
      const uint64_t e = (uint64_t)row * (uint64_t)K + (uint64_t)(4 * v);
      auto one = [&](float x, float m, float r, float g, float b, uint64_t idx) -> float {
        float y_ = fmaxf(fmaf((x - m) * r, g, b), 0.f);
        if (a.p > 0.f) y_ = rh_drop_hash(seed, ctr, idx) >= thr ? y_ * keep_scale : 0.f;
        return y_;
      };
      acc = fmaf(one(zv.x, mean.x, rstd.x, gam.x, bet.x, e), wv.x, acc);
      acc = fmaf(one(zv.y, mean.y, rstd.y, gam.y, bet.y, e + 1), wv.y, acc);
      acc = fmaf(one(zv.z, mean.z, rstd.z, gam.z, bet.z, e + 2), wv.z, acc);
      acc = fmaf(one(zv.w, mean.w, rstd.w, gam.w, bet.w, e + 3), wv.w, acc);
    }
    acc = group_sum16(acc);
    if (sub == 0) {
      float zz = acc + b0;
      if (a.e0) zz += a.e0[row];
      if (a.e1) zz += a.e1[row];
      const float yv = 1.f / (1.f + __expf(-zz));
      a.y[row] = yv;
      if (a.t) {
        const float tv = a.t[row];
        lacc -= tv * fmaxf(logf(yv), -100.f) + (1.f - tv) * fmaxf(log1pf(-yv), -100.f);
      }
    }
  }
  if (a.loss_partial) {
    if (sub == 0) lred[grp] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < kHeadRows; ++r) v += lred[r];
      a.loss_partial[blockIdx.x] = v;
    }
  }
}

// The scalar work of one training step (see step_scalars_kernel below); shared with head_bwd_kernel, whose launch can carry it
// as one extra workgroup (round 4): nothing between the head's forward and its backward needs these scalars.
struct StepScalarArgs {
  const float* loss_partial;
  int n;
  float inv_b;
  float* loss;
  double* hyper;
  int64_t* step;
  float* ring;
  int64_t ring_mask;
  int64_t *c0, inc0, mod0, *c1, inc1, mod1;
  int on;  // head_bwd_kernel: 1 = the last workgroup of the grid runs step_scalars_body
};

static __device__ __forceinline__ void step_scalars_body(const StepScalarArgs& q, float* red /* RH_BLOCK / RH_WAVE floats */) {
  if (q.loss_partial != nullptr) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < q.n; i += RH_BLOCK) acc += q.loss_partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x % RH_WAVE == 0) red[threadIdx.x / RH_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) q.loss[0] = (((red[0] + red[1]) + red[2]) + red[3]) * q.inv_b;
  }
  if (threadIdx.x == 0) {
    if (q.hyper != nullptr) {
      const int64_t t = *q.step + 1;
      *q.step = t;
      const double lr = q.hyper[0], b1 = q.hyper[1], b2 = q.hyper[2];
      const double bc1 = 1.0 - pow(b1, (double)t);
      const double bc2 = 1.0 - pow(b2, (double)t);
      q.hyper[8] = lr / bc1;
      q.hyper[9] = sqrt(bc2);
      q.hyper[10] = 1.0 - b1;
      q.hyper[11] = 1.0 - b2;
      q.hyper[12] = (double)t;
      const double A = q.hyper[8] * q.hyper[9], E = q.hyper[3] * q.hyper[9];
      q.hyper[13] = A;
      q.hyper[14] = E;
      if (q.ring != nullptr) {
        q.ring[2 * (t & q.ring_mask) + 0] = (float)A;
        q.ring[2 * (t & q.ring_mask) + 1] = (float)E;
      }
    }
    if (q.c0 != nullptr) {
      int64_t p = *q.c0 + q.inc0;
      if (q.mod0 > 0 && p >= q.mod0) p %= q.mod0;
      *q.c0 = p;
    }
    if (q.c1 != nullptr) {
      int64_t p = *q.c1 + q.inc1;
      if (q.mod1 > 0 && p >= q.mod1) p %= q.mod1;
      *q.c1 = p;
    }
  }
}

struct HeadBwdArgs {
  const float* h;
  int64_t ldh;
  const float* w;
  const float* y;
  const float* g_y;
  const float* t;       // != null: g_y is not read; g_y[row] = g_loss[0] / B * (y - t) / max((1 - y) y, 1e-12)  (BCELoss)
  const float* g_loss;  //          (the arithmetic of rh_bce_bwd, so fused == unfused bit for bit)
  int B, K;
  float* g_h;  // (B, K) contiguous
  float* g_z;  // (B,)
  float* partial;      // (gridDim.x, K + 1)
  float* g_w;          // (K,)
  float* g_b;          // (1,) or null
  // BN = true: h is the output of BatchNorm1d (+ ReLU, + Dropout) over bn_z, and g_h is that layer's whole upstream
  // gradient: its backward column sums (sum g1, sum g1 * xhat; csrc/mlp.hip bn_partial_kernel<1>, same mask arithmetic)
  // are formed here from the rows this workgroup walks anyway -> bn_partial (gridDim.x, 2, K)
  const float* bn_z;
  const float* bn_stat;  // (>= 2, K): mean, rstd
  const float* bn_gamma;
  const float* bn_beta;
  const int64_t* bn_rng;
  const int64_t* bn_ctr;
  float* bn_partial;
  float bn_p;
  int bn_relu;
  StepScalarArgs sc;  // sc.on: the grid has one workgroup more, which runs the step's scalar work instead of rows
};

template <int MAXV, bool BN = false>
__global__ __launch_bounds__(RH_BLOCK) void head_bwd_kernel(const HeadBwdArgs a) {
  RH_CHAIN_PRIO();
  __shared__ float red[kHeadRows][kHeadLanes * 4 + 1];
  __shared__ float gb_red[kHeadRows];
  const int nrow_blocks = (int)gridDim.x - (a.sc.on ? 1 : 0);
  if (a.sc.on && (int)blockIdx.x == nrow_blocks) {
    step_scalars_body(a.sc, gb_red);
    return;
  }
  const int sub = threadIdx.x % kHeadLanes, grp = threadIdx.x / kHeadLanes;
  const int K = a.K, nv = K / 4;
  float4 wacc[MAXV];
  float4 wv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    wacc[i] = f4_zero();
    const int v = sub + i * kHeadLanes;
    wv[i] = v < nv ? gload<float4>(a.w + 4 * v) : f4_zero();
  }
  float4 bmean[BN ? MAXV : 1], brstd[BN ? MAXV : 1], bgam[BN ? MAXV : 1], bbet[BN ? MAXV : 1];
  float4 bs1[BN ? MAXV : 1], bs2[BN ? MAXV : 1];
  float keep_scale = 1.f;
  uint32_t thr = 0;
  uint64_t seed = 0, ctr = 0;
  if (BN) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = sub + i * kHeadLanes;
      bs1[i] = bs2[i] = f4_zero();
      bmean[i] = v < nv ? gload<float4>(a.bn_stat + 4 * v) : f4_zero();
      brstd[i] = v < nv ? gload<float4>(a.bn_stat + K + 4 * v) : f4_zero();
      bgam[i] = v < nv ? gload<float4>(a.bn_gamma + 4 * v) : f4_zero();
      bbet[i] = v < nv ? gload<float4>(a.bn_beta + 4 * v) : f4_zero();
    }
    if (a.bn_p > 0.f) {
      keep_scale = 1.f / (1.f - a.bn_p);
      thr = (uint32_t)(a.bn_p * 4294967296.0);
      seed = (uint64_t)a.bn_rng[0];
      ctr = (uint64_t)a.bn_ctr[0];
    }
  }
  float gb = 0.f;
  // K % 4 tail columns 4 nv + sub (sub < K - 4 nv <= 3): one scalar lane each
  const bool has_tail = sub < K - 4 * nv;
  const int tcol = 4 * nv + sub;
  const float tw = has_tail ? a.w[tcol] : 0.f;
  float tacc = 0.f;
  const float lscale = a.t ? a.g_loss[0] / (float)a.B : 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kHeadRows + grp; row < a.B; row += (int64_t)nrow_blocks * kHeadRows) {
    const float yv = a.y[row];
    // the BCE gradient formed inline (t given) and / or an upstream gradient of y (another consumer of the prediction)
    float gy = a.t ? lscale * (yv - a.t[row]) / fmaxf((1.f - yv) * yv, 1e-12f) : a.g_y[row];
    if (a.t && a.g_y) gy += a.g_y[row];
    const float gz = gy * yv * (1.f - yv);
    if (sub == 0) {
      a.g_z[row] = gz;
      gb += gz;
    }
    if (has_tail) {
      tacc = fmaf(gz, a.h[row * a.ldh + tcol], tacc);
      a.g_h[row * K + tcol] = gz * tw;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = sub + i * kHeadLanes;
      if (v < nv) {
        // BN with h == null (fused MLP chain, round 4): the head's input was never written -- the forward applied
        // BatchNorm + ReLU + Dropout on the way into its dot product -- and is recomputed here from bn_z the same way
        const bool recompute = BN && a.h == nullptr;
        float4 hv = recompute ? f4_zero() : gload<float4>(a.h + row * a.ldh + 4 * v);
        const float4 gh = make_float4(gz * wv[i].x, gz * wv[i].y, gz * wv[i].z, gz * wv[i].w);
        gstore<float4>(a.g_h + row * K + 4 * v, gh);
        if (BN) {
          const float4 zv = gload<float4>(a.bn_z + row * K + 4 * v);
          const uint64_t e0 = (uint64_t)row * K + 4 * v;
          auto one = [&](float z, float mean, float rstd, float gam, float bet, float g, uint64_t e, float& s1, float& s2,
                         float& act) {
            const float xhat = (z - mean) * rstd;
            const float bn = fmaf(xhat, gam, bet);
            float g1 = (!a.bn_relu || bn > 0.f) ? g : 0.f;
            float y_ = a.bn_relu ? fmaxf(bn, 0.f) : bn;
            if (a.bn_p > 0.f) {
              const bool keep = rh_drop_hash(seed, ctr, e) >= thr;
              g1 = keep ? g1 * keep_scale : 0.f;
              y_ = keep ? y_ * keep_scale : 0.f;
            }
            if (recompute) act = y_;
            s1 += g1;
            s2 = fmaf(g1, xhat, s2);
          };
          one(zv.x, bmean[i].x, brstd[i].x, bgam[i].x, bbet[i].x, gh.x, e0 + 0, bs1[i].x, bs2[i].x, hv.x);
          one(zv.y, bmean[i].y, brstd[i].y, bgam[i].y, bbet[i].y, gh.y, e0 + 1, bs1[i].y, bs2[i].y, hv.y);
          one(zv.z, bmean[i].z, brstd[i].z, bgam[i].z, bbet[i].z, gh.z, e0 + 2, bs1[i].z, bs2[i].z, hv.z);
          one(zv.w, bmean[i].w, brstd[i].w, bgam[i].w, bbet[i].w, gh.w, e0 + 3, bs1[i].w, bs2[i].w, hv.w);
        }
        wacc[i].x = fmaf(gz, hv.x, wacc[i].x);
        wacc[i].y = fmaf(gz, hv.y, wacc[i].y);
        wacc[i].z = fmaf(gz, hv.z, wacc[i].z);
        wacc[i].w = fmaf(gz, hv.w, wacc[i].w);
      }
    }
  }
  // block reduction over the kHeadRows row groups, 64 columns (16 lanes x float4) at a time
  float* part = a.partial + (int64_t)blockIdx.x * (K + 1);
  if (sub == 0) gb_red[grp] = gb;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    __syncthreads();
    red[grp][sub * 4 + 0] = wacc[i].x;
    red[grp][sub * 4 + 1] = wacc[i].y;
    red[grp][sub * 4 + 2] = wacc[i].z;
    red[grp][sub * 4 + 3] = wacc[i].w;
    __syncthreads();
    if (threadIdx.x < kHeadLanes * 4) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < kHeadRows; ++r) v += red[r][threadIdx.x];
      const int col = i * kHeadLanes * 4 + threadIdx.x;
      if (col < K) part[col] = v;
    }
    if (BN) {  // the same reduction over the row groups for the two BatchNorm sums of these 64 columns
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float4 src = which == 0 ? bs1[i] : bs2[i];
        __syncthreads();
        red[grp][sub * 4 + 0] = src.x;
        red[grp][sub * 4 + 1] = src.y;
        red[grp][sub * 4 + 2] = src.z;
        red[grp][sub * 4 + 3] = src.w;
        __syncthreads();
        if (threadIdx.x < kHeadLanes * 4) {
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < kHeadRows; ++r) v += red[r][threadIdx.x];
          const int col = i * kHeadLanes * 4 + threadIdx.x;
          if (col < K) a.bn_partial[((int64_t)blockIdx.x * 2 + which) * K + col] = v;
        }
      }
    }
  }
  __syncthreads();
  if (sub < 4) red[grp][sub] = tacc;
  __syncthreads();
  if ((int)threadIdx.x < K - 4 * nv) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) v += red[r][threadIdx.x];
    part[4 * nv + threadIdx.x] = v;
  }
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int r = 0; r < kHeadRows; ++r) v += gb_red[r];
    part[K] = v;
  }
}

// Column sums of a (rows, cols) partial buffer -> out[0 .. split) and out2[0 .. cols - split); the extra last block sums
// the vector v (n,) into vsum.  32 columns x 8 row groups per block: 128-byte coalesced reads, fixed summation order.
constexpr int kCsCols = 32, kCsGroups = RH_BLOCK / kCsCols;

__global__ __launch_bounds__(RH_BLOCK) void colsum_kernel(const float* __restrict__ a, int rows, int cols,
                                                          float* __restrict__ out, int split, float* __restrict__ out2,
                                                          const float* __restrict__ v, int64_t n,
                                                          float* __restrict__ vsum) {
  RH_CHAIN_PRIO();
  __shared__ float red[kCsGroups][kCsCols + 1];
  const int nb_cols = (cols + kCsCols - 1) / kCsCols;
  if ((int)blockIdx.x < nb_cols) {
    const int c = threadIdx.x % kCsCols, grp = threadIdx.x / kCsCols;
    const int col = blockIdx.x * kCsCols + c;
    float acc = 0.f;
    if (col < cols) {
      int r = grp;
      for (; r + 3 * kCsGroups < rows; r += 4 * kCsGroups) {
        const float t0 = a[(int64_t)r * cols + col], t1 = a[(int64_t)(r + kCsGroups) * cols + col];
        const float t2 = a[(int64_t)(r + 2 * kCsGroups) * cols + col], t3 = a[(int64_t)(r + 3 * kCsGroups) * cols + col];
        acc = (((acc + t0) + t1) + t2) + t3;
      }
      for (; r < rows; r += kCsGroups) acc += a[(int64_t)r * cols + col];
    }
    red[grp][c] = acc;
    __syncthreads();
    if (grp == 0 && col < cols) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < kCsGroups; ++q) t += red[q][c];
      if (col < split) {
        out[col] = t;
      } else if (out2) {
        out2[col - split] = t;
      }
    }
  } else {
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += RH_BLOCK) acc += v[i];
    acc = wave_sum(acc);
    if (threadIdx.x % RH_WAVE == 0) red[0][threadIdx.x / RH_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) vsum[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  }
}

void launch_colsum(const float* a, int rows, int cols, float* out, int split, float* out2, const float* v, int64_t n,
                   float* vsum, hipStream_t st) {
  const int nb = (cols + kCsCols - 1) / kCsCols + (v ? 1 : 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(RH_BLOCK), 0, st, a, rows, cols, out, split, out2, v, n, vsum);
}

int head_grid(int B) {
  int g = (B + kHeadRows * 2 - 1) / (kHeadRows * 2);  // >= 2 rows per lane group (4: 64 workgroups at B = 4096, a quarter of the CUs)
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return g;
}

// ---------------------------------------------------------------------------------------------
constexpr int kBceBlock = 1024;

__global__ __launch_bounds__(kBceBlock) void bce_fwd_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                            int64_t B, float* __restrict__ loss) {
  __shared__ float red[kBceBlock / RH_WAVE];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += kBceBlock) {
    const float yv = y[i], tv = t[i];
    const float ly = fmaxf(logf(yv), -100.f), l1y = fmaxf(log1pf(-yv), -100.f);
    acc -= tv * ly + (1.f - tv) * l1y;
  }
  acc = wave_sum(acc);
  if (threadIdx.x % RH_WAVE == 0) red[threadIdx.x / RH_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < kBceBlock / RH_WAVE; ++w) v += red[w];
    loss[0] = v / (float)B;
  }
}

__global__ __launch_bounds__(RH_BLOCK) void bce_bwd_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                           const float* __restrict__ g_loss, int64_t B,
                                                           float* __restrict__ g_y) {
  const float scale = g_loss[0] / (float)B;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < B; i += (int64_t)gridDim.x * RH_BLOCK) {
    const float yv = y[i];
    g_y[i] = scale * (yv - t[i]) / fmaxf((1.f - yv) * yv, 1e-12f);
  }
}

// The scalar work of one training step in ONE launch (was: rh_bce_fwd + rh_adam_prepare + rh_batch_advance):
//   loss[0] = sum(loss_partial[0 .. n)) / B          (mean BCE from the per-block terms of rh_head_loss_fwd)
//   Adam bias corrections of step t + 1 (== rh_adam_prepare: step counter, hyper[8..14], the (A, E) ring)
//   up to two device counters advanced: *c = (*c + inc) % mod (mod == 0: no wrap) -- the loader's batch position, ...
// Each part is skipped when its pointer is null.
__global__ __launch_bounds__(RH_BLOCK) void step_scalars_kernel(const StepScalarArgs q) {
  RH_CHAIN_PRIO();
  __shared__ float red[RH_BLOCK / RH_WAVE];
  step_scalars_body(q, red);
}

}  // namespace

extern "C" int rh_step_scalars(const float* loss_partial, int n_partial, int64_t B, float* loss, double* hyper,
                               int64_t* step, float* ring, int ring_size, int64_t* c0, int64_t inc0, int64_t mod0,
                               int64_t* c1, int64_t inc1, int64_t mod1, void* stream) {
  RH_REQUIRE(loss_partial == nullptr || (loss != nullptr && n_partial >= 1 && B >= 1), RH_E_BADARG,
             "rh_step_scalars: loss_partial needs loss, n_partial >= 1 and B >= 1");
  RH_REQUIRE(hyper == nullptr || step != nullptr, RH_E_BADARG, "rh_step_scalars: hyper without step");
  RH_REQUIRE(ring == nullptr || (ring_size > 0 && (ring_size & (ring_size - 1)) == 0), RH_E_BADARG,
             "rh_step_scalars: ring_size must be a power of two");
  const StepScalarArgs q{loss_partial, n_partial, loss_partial ? 1.f / (float)B : 0.f, loss, hyper, step, ring,
                         (int64_t)(ring_size - 1), c0, inc0, mod0, c1, inc1, mod1, 0};
  hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), q);
  RH_LAUNCH_CHECK("rh_step_scalars");
  return 0;
}

extern "C" int64_t rh_linear_wgrad_workspace(int B, int N, int K) {
  if (B < 1 || N < 1 || K < 1) return 0;
  int tn, tk, S, rps;
  wgrad_plan(B, N, K, &tn, &tk, &S, &rps);
  return (int64_t)S * ((int64_t)N * K + N);
}

extern "C" int rh_linear_wgrad_splits(int B, int N, int K) {
  if (B < 1 || N < 1 || K < 1) return 0;
  int tn, tk, S, rps;
  wgrad_plan(B, N, K, &tn, &tk, &S, &rps);
  return S;
}

extern "C" int rh_linear_wgrad_tiles(int N, int K) { return ((N + kTile - 1) / kTile) * ((K + kTile - 1) / kTile); }

// reduce != 0: dW / db are produced (second small launch when the batch was split).  reduce == 0: only the per-split
// slabs are written -- partial holds S x (N, K) then S x (N,) with S = rh_linear_wgrad_splits(B, N, K) -- and the caller
// sums them (rh_pack_grads); dW / db may be null.
static int wgrad_impl(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K, float* dW, float* db,
                      float* partial, int reduce, void* stream);

extern "C" int rh_linear_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K,
                               float* dW, float* db, float* partial, void* stream) {
  RH_REQUIRE(dW != nullptr, RH_E_BADARG, "rh_linear_wgrad: null pointer");
  return wgrad_impl(g, ldg, x, ldx, B, N, K, dW, db, partial, 1, stream);
}

extern "C" int rh_linear_wgrad_partial(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K,
                                       float* partial, void* stream) {
  return wgrad_impl(g, ldg, x, ldx, B, N, K, nullptr, nullptr, partial, 0, stream);
}

static int wgrad_impl(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K, float* dW, float* db,
                      float* partial, int reduce, void* stream) {
  RH_REQUIRE(g && x && partial, RH_E_BADARG, "rh_linear_wgrad: null pointer");
  RH_REQUIRE(B >= 1 && N >= 1 && K >= 1 && ldg >= N && ldx >= K, RH_E_BADARG,
             "rh_linear_wgrad: bad shape B=%d N=%d K=%d ldg=%lld ldx=%lld", B, N, K, (long long)ldg, (long long)ldx);
  WgradArgs a{g, ldg, x, ldx, B, N, K, 1, B, partial, dW, db, 0};
  int tn, tk;
  wgrad_plan(B, N, K, &tn, &tk, &a.S, &a.rows_per_split);
  a.direct = (reduce && a.S == 1) ? 1 : 0;
  const bool long_form = (B >= kLongRows && g_long_blocks > 0) || (B < kLongRows && g_short_form == 1);
  const size_t lds = (size_t)(long_form ? 1 : kWaves) * kPartStride * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kWaves * kPartStride * sizeof(float)));
    RH_REQUIRE(e == hipSuccess, (int)e, "rh_linear_wgrad: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (wgrad_rows_form(B, N, K)) {
    // two tiles per wavefront along K when the K tiles pair up, else along N (RH_TUNE_WGRAD_ROWS_FORM < 0: one per wavefront)
    if (g_rows_pair && tk % 2 == 0)
      hipLaunchKernelGGL((linear_wgrad_rows_kernel<1, 2>), dim3(a.S), dim3(RH_WAVE * tn * (tk / 2)), 0, st, a, tk / 2);
    else if (g_rows_pair && tn % 2 == 0)
      hipLaunchKernelGGL((linear_wgrad_rows_kernel<2, 1>), dim3(a.S), dim3(RH_WAVE * (tn / 2) * tk), 0, st, a, tk);
    else
      hipLaunchKernelGGL((linear_wgrad_rows_kernel<1, 1>), dim3(a.S), dim3(RH_WAVE * tn * tk), 0, st, a, tk);
  } else if (long_form)
    hipLaunchKernelGGL(linear_wgrad_long_kernel, dim3(tk, tn, a.S), dim3(RH_BLOCK), lds, st, a);
  else
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(tk, tn, a.S), dim3(RH_BLOCK), lds, st, a);
  if (reduce && a.S > 1)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((int64_t)N * K + N + RH_BLOCK - 1) / RH_BLOCK)), dim3(RH_BLOCK),
                       0, st, a);
  RH_LAUNCH_CHECK("rh_linear_wgrad");
  return 0;
}

// (declared in wgrad_body.h: the argument block of a grouped launch, for rh_linear_wgrad_partial_group below and for
// rh_adam_lazy_step_ahead_wgrad of csrc/optim.hip)
int rh_wgrad_group_fill(int n, const float* const* g, const int64_t* ldg, const float* const* x, const int64_t* ldx, const int* B,
                        const int* N, const int* K, float* const* partial, rh_wgrad::WgradGroupArgs* out, const char* who) {
  RH_REQUIRE(n >= 1 && n <= kWgradGroup && g && ldg && x && ldx && B && N && K && partial, RH_E_BADARG,
             "%s: 1 <= n <= %d problems", who, kWgradGroup);
  WgradGroupArgs& ga = *out;
  ga = WgradGroupArgs{};
  ga.n = n;
  ga.prefix[0] = 0;
  for (int i = 0; i < n; ++i) {
    RH_REQUIRE(g[i] && x[i] && partial[i] && B[i] >= 1 && N[i] >= 1 && K[i] >= 1 && ldg[i] >= N[i] && ldx[i] >= K[i], RH_E_BADARG,
               "%s: bad problem %d", who, i);
    RH_REQUIRE(B[i] < kLongRows, RH_E_UNSUPPORTED, "%s: B = %d (long reductions take the single call)", who, B[i]);
    WgradArgs a{g[i], ldg[i], x[i], ldx[i], B[i], N[i], K[i], 1, B[i], partial[i], nullptr, nullptr, 0};
    int tn, tk;
    wgrad_plan(B[i], N[i], K[i], &tn, &tk, &a.S, &a.rows_per_split);
    ga.p[i] = a;
    ga.tiles_k[i] = tk;
    ga.tiles_n[i] = tn;
    ga.prefix[i + 1] = ga.prefix[i] + tk * tn * a.S;
  }
  for (int i = n; i < kWgradGroup; ++i) ga.prefix[i + 1] = ga.prefix[n];
  return 0;
}

// n <= 8 problems of rh_linear_wgrad_partial as one launch: arrays of n entries each (host memory); problem i writes its
// slabs to partial[i] (rh_linear_wgrad_workspace(B[i], N[i], K[i]) floats, S = rh_linear_wgrad_splits(...) as the single call).
extern "C" int rh_linear_wgrad_partial_group(int n, const float* const* g, const int64_t* ldg, const float* const* x,
                                             const int64_t* ldx, const int* B, const int* N, const int* K,
                                             float* const* partial, void* stream) {
  WgradGroupArgs ga;
  const int rc = rh_wgrad_group_fill(n, g, ldg, x, ldx, B, N, K, partial, &ga, "rh_linear_wgrad_partial_group");
  if (rc != 0) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_group_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kWaves * kPartStride * sizeof(float)));
    RH_REQUIRE(e == hipSuccess, (int)e, "rh_linear_wgrad_partial_group: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  if (g_short_form == 1)
    hipLaunchKernelGGL(linear_wgrad_group_long_kernel, dim3((unsigned)ga.prefix[n]), dim3(RH_BLOCK),
                       (size_t)kPartStride * sizeof(float), reinterpret_cast<hipStream_t>(stream), ga);
  else
    hipLaunchKernelGGL(linear_wgrad_group_kernel, dim3((unsigned)ga.prefix[n]), dim3(RH_BLOCK),
                       (size_t)kWaves * kPartStride * sizeof(float), reinterpret_cast<hipStream_t>(stream), ga);
  RH_LAUNCH_CHECK("rh_linear_wgrad_partial_group");
  return 0;
}

extern "C" int rh_linear_set_tuning(int key, int value) {
  if (key == RH_TUNE_WGRAD_BLOCKS) {
    g_long_blocks = value;
    return 0;
  }
  if (key == RH_TUNE_WGRAD_SHORT_FORM) {
    g_short_form = value;
    return 0;
  }
  if (key == RH_TUNE_WGRAD_ROWS_FORM) {
    g_rows_form = value < 0 ? -value : value;
    g_rows_pair = value < 0 ? 0 : 1;
    return 0;
  }
  return RH_E_BADARG;
}

extern "C" int rh_head_nblocks(int B) { return head_grid(B); }

int head_fwd_grid(int B) {
  int grid = (B + kHeadRows - 1) / kHeadRows;
  if (grid > 256 * 8) grid = 256 * 8;
  return grid < 1 ? 1 : grid;
}

extern "C" int rh_head_fwd(const float* h, int64_t ldh, const float* w, const float* bias, const float* e0,
                           const float* e1, int B, int K, float* y, void* stream) {
  RH_REQUIRE(h && w && y, RH_E_BADARG, "rh_head_fwd: null pointer");
  RH_REQUIRE(B >= 0 && K >= 1 && ldh >= K, RH_E_UNSUPPORTED, "rh_head_fwd: bad width K=%d", K);
  if (B == 0) return 0;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(head_fwd_grid(B)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), h,
                     ldh, w, bias, e0, e1, B, K, y, (const float*)nullptr, (float*)nullptr);
  RH_LAUNCH_CHECK("rh_head_fwd");
  return 0;
}

extern "C" int rh_head_loss_nblocks(int B) { return head_fwd_grid(B); }

extern "C" int rh_head_loss_fwd(const float* h, int64_t ldh, const float* w, const float* bias, const float* e0,
                                const float* e1, int B, int K, float* y, const float* t, float* loss_partial,
                                void* stream) {
  RH_REQUIRE(h && w && y && t && loss_partial, RH_E_BADARG, "rh_head_loss_fwd: null pointer");
  RH_REQUIRE(B >= 1 && K >= 1 && ldh >= K, RH_E_UNSUPPORTED, "rh_head_loss_fwd: bad shape B=%d K=%d", B, K);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(head_fwd_grid(B)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), h,
                     ldh, w, bias, e0, e1, B, K, y, t, loss_partial);
  RH_LAUNCH_CHECK("rh_head_loss_fwd");
  return 0;
}

// Output head of the fused MLP chain: see head_bnact_fwd_kernel.  z (B, K) = PRE-BatchNorm activations of the last hidden
// layer, stats (ceil(B / stats_rows), 2, K) the per-slab (sum, M2) of the GEMM that produced it, ctr the dropout counter
// that GEMM drew for this layer.  t / loss_partial as rh_head_loss_fwd (both NULL: no loss terms); loss_partial has
// rh_head_loss_nblocks(B) entries.  K % 4 == 0, K <= 256.
extern "C" int rh_head_bnact_fwd(const float* z, int64_t ldz, const float* stats, int stats_rows, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                 float p_drop, const int64_t* rng, const int64_t* ctr, float* stat_out, const float* w,
                                 const float* bias, const float* e0, const float* e1, int B, int K, float* y, const float* t,
                                 float* loss_partial, void* stream) {
  RH_REQUIRE(z && stats && gamma && beta && rng && ctr && stat_out && w && y, RH_E_BADARG, "rh_head_bnact_fwd: null pointer");
  RH_REQUIRE(B >= 2 && K >= 4 && K % 4 == 0 && K <= kHeadBnMaxK && ldz >= K && stats_rows >= 1, RH_E_UNSUPPORTED,
             "rh_head_bnact_fwd: bad shape B=%d K=%d (K %% 4 == 0, K <= %d)", B, K, kHeadBnMaxK);
  RH_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (t == nullptr) == (loss_partial == nullptr), RH_E_BADARG,
             "rh_head_bnact_fwd: bad arguments");
  RH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), RH_E_BADARG,
             "rh_head_bnact_fwd: running_mean and running_var go together");
  HeadBnFwdArgs a{z, ldz, stats, stats_rows, gamma, beta, running_mean, running_var, momentum, eps, p_drop, rng, ctr,
                  stat_out, w, bias, e0, e1, B, K, y, t, loss_partial};
  hipLaunchKernelGGL(head_bnact_fwd_kernel, dim3(head_fwd_grid(B)), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     a);
  RH_LAUNCH_CHECK("rh_head_bnact_fwd");
  return 0;
}

struct HeadBnArgs {
  const float *z, *stat, *gamma, *beta;
  const int64_t *rng, *ctr;
  float* partial;
  float p;
  int relu;
};
static int head_bwd_impl(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                         const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b,
                         float* partial, void* stream, int reduce = 1, const HeadBnArgs* bn = nullptr,
                         const StepScalarArgs* sc = nullptr);

// rh_head_bwd_bn whose launch also carries the step's scalar work (rh_step_scalars' arguments) as ONE extra workgroup: the
// mean of the BCE terms the head's forward left behind, the Adam bias corrections of the coming optimizer step and the
// device counters.  Nothing between the head's forward and this launch reads those scalars (the loss value is consumed
// after the backward; trainers/ctr_trainer.py:88-99), so the scalar launch of the step disappears (round 4).
extern "C" int rh_head_bwd_bn_scalars(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y,
                                      const float* t, const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w,
                                      float* g_b, float* partial, int reduce, const float* bn_z, const float* bn_stat,
                                      const float* bn_gamma, const float* bn_beta, float bn_p, const int64_t* bn_rng,
                                      const int64_t* bn_ctr, int bn_relu, float* bn_partial, const float* loss_partial,
                                      int n_partial, float* loss, double* hyper, int64_t* step, float* ring, int ring_size,
                                      int64_t* c0, int64_t inc0, int64_t mod0, int64_t* c1, int64_t inc1, int64_t mod1,
                                      void* stream) {
  RH_REQUIRE(g_y != nullptr || (t != nullptr && g_loss != nullptr), RH_E_BADARG,
             "rh_head_bwd_bn_scalars: give g_y and / or (t, g_loss)");
  RH_REQUIRE((t != nullptr) == (g_loss != nullptr), RH_E_BADARG, "rh_head_bwd_bn_scalars: t and g_loss go together");
  RH_REQUIRE(bn_z && bn_stat && bn_gamma && bn_beta && bn_partial && (bn_p <= 0.f || (bn_rng && bn_ctr)), RH_E_BADARG,
             "rh_head_bwd_bn_scalars: null pointer");
  RH_REQUIRE(K % 4 == 0 && bn_p >= 0.f && bn_p < 1.f, RH_E_UNSUPPORTED, "rh_head_bwd_bn_scalars: K=%d p=%g unsupported", K, bn_p);
  RH_REQUIRE(loss_partial == nullptr || (loss != nullptr && n_partial >= 1), RH_E_BADARG,
             "rh_head_bwd_bn_scalars: loss_partial needs loss and n_partial >= 1");
  RH_REQUIRE(hyper == nullptr || step != nullptr, RH_E_BADARG, "rh_head_bwd_bn_scalars: hyper without step");
  RH_REQUIRE(ring == nullptr || (ring_size > 0 && (ring_size & (ring_size - 1)) == 0), RH_E_BADARG,
             "rh_head_bwd_bn_scalars: ring_size must be a power of two");
  const HeadBnArgs bn{bn_z, bn_stat, bn_gamma, bn_beta, bn_rng, bn_ctr, bn_partial, bn_p, bn_relu};
  const StepScalarArgs sc{loss_partial, n_partial, loss_partial ? 1.f / (float)B : 0.f, loss, hyper, step, ring,
                          (int64_t)(ring_size - 1), c0, inc0, mod0, c1, inc1, mod1, 1};
  return head_bwd_impl(h, ldh, w, y, g_y, t, g_loss, B, K, g_h, g_z, g_w, g_b, partial, stream, reduce, &bn, &sc);
}

// rh_head_bwd_ex + the BatchNorm-backward column sums of the hidden layer below the head (h = dropout(relu(bn(bn_z)))
// and the head is its only consumer): bn_partial (rh_head_nblocks(B), 2, K) = per-block (sum g1, sum g1 * xhat), what
// rh_bn_relu_dropout_bwd_pre takes instead of launching its own statistics pass.  K % 4 == 0.
// h may be NULL: the head's input is then recomputed from bn_z (forward = rh_head_bnact_fwd, which never wrote it).
extern "C" int rh_head_bwd_bn(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                              const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b,
                              float* partial, int reduce, const float* bn_z, const float* bn_stat, const float* bn_gamma,
                              const float* bn_beta, float bn_p, const int64_t* bn_rng, const int64_t* bn_ctr, int bn_relu,
                              float* bn_partial, void* stream) {
  RH_REQUIRE(g_y != nullptr || (t != nullptr && g_loss != nullptr), RH_E_BADARG,
             "rh_head_bwd_bn: give g_y and / or (t, g_loss)");
  RH_REQUIRE((t != nullptr) == (g_loss != nullptr), RH_E_BADARG, "rh_head_bwd_bn: t and g_loss go together");
  RH_REQUIRE(bn_z && bn_stat && bn_gamma && bn_beta && bn_partial && (bn_p <= 0.f || (bn_rng && bn_ctr)), RH_E_BADARG,
             "rh_head_bwd_bn: null pointer");
  RH_REQUIRE(K % 4 == 0 && bn_p >= 0.f && bn_p < 1.f, RH_E_UNSUPPORTED, "rh_head_bwd_bn: K=%d p=%g unsupported", K, bn_p);
  const HeadBnArgs bn{bn_z, bn_stat, bn_gamma, bn_beta, bn_rng, bn_ctr, bn_partial, bn_p, bn_relu};
  return head_bwd_impl(h, ldh, w, y, g_y, t, g_loss, B, K, g_h, g_z, g_w, g_b, partial, stream, reduce, &bn);
}

// One entry point for every combination: g_y given (t, g_loss null), the BCE gradient formed inline (g_y null), or both
// (the prediction has a second consumer besides the fused loss: the two gradients of y add);
// reduce == 0: g_w / g_b are not produced, the caller sums the rh_head_nblocks(B) x (K + 1) partial rows (rh_pack_grads).
extern "C" int rh_head_bwd_ex(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                              const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b,
                              float* partial, int reduce, void* stream) {
  RH_REQUIRE(g_y != nullptr || (t != nullptr && g_loss != nullptr), RH_E_BADARG,
             "rh_head_bwd_ex: give g_y and / or (t, g_loss)");
  RH_REQUIRE((t != nullptr) == (g_loss != nullptr), RH_E_BADARG, "rh_head_bwd_ex: t and g_loss go together");
  return head_bwd_impl(h, ldh, w, y, g_y, t, g_loss, B, K, g_h, g_z, g_w, g_b, partial, stream, reduce);
}

extern "C" int rh_head_bwd(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, int B, int K,
                           float* g_h, float* g_z, float* g_w, float* g_b, float* partial, void* stream) {
  RH_REQUIRE(g_y != nullptr, RH_E_BADARG, "rh_head_bwd: null pointer");
  return head_bwd_impl(h, ldh, w, y, g_y, nullptr, nullptr, B, K, g_h, g_z, g_w, g_b, partial, stream);
}

extern "C" int rh_head_loss_bwd(const float* h, int64_t ldh, const float* w, const float* y, const float* t,
                                const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b,
                                float* partial, void* stream) {
  RH_REQUIRE(t && g_loss, RH_E_BADARG, "rh_head_loss_bwd: null pointer");
  return head_bwd_impl(h, ldh, w, y, nullptr, t, g_loss, B, K, g_h, g_z, g_w, g_b, partial, stream);
}

static int head_bwd_impl(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                         const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b,
                         float* partial, void* stream, int reduce, const HeadBnArgs* bn, const StepScalarArgs* sc) {
  RH_REQUIRE((h || bn) && w && y && g_h && g_z && (g_w || !reduce) && partial, RH_E_BADARG, "rh_head_bwd: null pointer");
  RH_REQUIRE(B >= 1 && K >= 1 && K <= 4 * kHeadLanes * kHeadMaxV4 && ldh >= K, RH_E_UNSUPPORTED,
             "rh_head_bwd: K=%d unsupported (1 .. %d)", K, 4 * kHeadLanes * kHeadMaxV4);
  HeadBwdArgs a{h, ldh, w, y, g_y, t, g_loss, B, K, g_h, g_z, partial, g_w, g_b};
  if (sc != nullptr) a.sc = *sc;
  const int need = K < 4 ? 1 : (K / 4 + kHeadLanes - 1) / kHeadLanes;
  const int row_blocks = head_grid(B);
  const dim3 grid(row_blocks + (sc != nullptr ? 1 : 0)), block(RH_BLOCK);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bn != nullptr) {
    a.bn_z = bn->z, a.bn_stat = bn->stat, a.bn_gamma = bn->gamma, a.bn_beta = bn->beta, a.bn_rng = bn->rng;
    a.bn_ctr = bn->ctr, a.bn_partial = bn->partial, a.bn_p = bn->p, a.bn_relu = bn->relu;
    RH_REQUIRE(need <= 4, RH_E_UNSUPPORTED, "rh_head_bwd_bn: K=%d too wide (max %d)", K, 4 * kHeadLanes * 4);
    if (need <= 1) hipLaunchKernelGGL((head_bwd_kernel<1, true>), grid, block, 0, st, a);
    else if (need <= 2) hipLaunchKernelGGL((head_bwd_kernel<2, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((head_bwd_kernel<4, true>), grid, block, 0, st, a);
  } else
  if (need <= 1) hipLaunchKernelGGL(head_bwd_kernel<1>, grid, block, 0, st, a);
  else if (need <= 2) hipLaunchKernelGGL(head_bwd_kernel<2>, grid, block, 0, st, a);
  else if (need <= 4) hipLaunchKernelGGL(head_bwd_kernel<4>, grid, block, 0, st, a);
  else if (need <= 8) hipLaunchKernelGGL(head_bwd_kernel<8>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(head_bwd_kernel<16>, grid, block, 0, st, a);
  if (reduce) launch_colsum(partial, row_blocks, K + 1, g_w, K, g_b, nullptr, 0, nullptr, st);
  RH_LAUNCH_CHECK("rh_head_bwd");
  return 0;
}

extern "C" int rh_colsum(const float* a, int rows, int cols, float* out, const float* v, int64_t n, float* vsum,
                         void* stream) {
  RH_REQUIRE((a && out && rows >= 0 && cols >= 1) || (!a && v), RH_E_BADARG, "rh_colsum: bad arguments");
  RH_REQUIRE(!v || (vsum && n >= 0), RH_E_BADARG, "rh_colsum: vector sum needs an output");
  launch_colsum(a, a ? rows : 0, a ? cols : 0, out, cols, nullptr, v, n, vsum, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_colsum");
  return 0;
}

extern "C" int rh_bce_fwd(const float* y, const float* t, int64_t B, float* loss, void* stream) {
  RH_REQUIRE(y && t && loss && B >= 1, RH_E_BADARG, "rh_bce_fwd: bad arguments");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(kBceBlock), 0, reinterpret_cast<hipStream_t>(stream), y, t, B, loss);
  RH_LAUNCH_CHECK("rh_bce_fwd");
  return 0;
}

extern "C" int rh_bce_bwd(const float* y, const float* t, const float* g_loss, int64_t B, float* g_y, void* stream) {
  RH_REQUIRE(y && t && g_loss && g_y && B >= 1, RH_E_BADARG, "rh_bce_bwd: bad arguments");
  int64_t grid = (B + RH_BLOCK - 1) / RH_BLOCK;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), y, t,
                     g_loss, B, g_y);
  RH_LAUNCH_CHECK("rh_bce_bwd");
  return 0;
}
