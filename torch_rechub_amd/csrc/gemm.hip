// f32 GEMMs of the MLP at CTR batch sizes (M = batch = 4096, N, K <= 512), on the f32 MFMA.
//
// Reference: every nn.Linear of MLP, torch_rechub/basic/layers.py:279,290 -- forward y = x W^T + b and the input
// gradient g_x = g W (aten::addmm / mm).  At M = 4096 the library picks 256x16 / 224x32 macro-tiles that run at
// 20-45 TF (9-21 us per GEMM in the round-1 profile); these shapes are small enough that one 64x64 (or 32x64) tile per
// workgroup fills the chip exactly once, so a plain LDS-tiled kernel on v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered
// fmaf chain) gets close to the 157 TF f32 matrix peak without any tuning database.
//
//   C[M, N] = A[M, K] * op(B) (+ bias[N])
//     B_KMAJOR = true  : B is (N, K) row-major, C = A B^T      (forward: B = weight)
//     B_KMAJOR = false : B is (K, N) row-major, C = A B        (input gradient: A = g, B = weight)
//   Workgroup = WM x WN wavefronts, each owning one 32x32 accumulator tile; K is walked in tiles of 32 through two LDS
//   buffers (global -> registers for tile t+1 is issued before the MFMAs of tile t).  Within a K tile MFMA step s takes
//   the k pair {s, 16 + s}: lane (i, kk) then reads 16 CONSECUTIVE floats of its row -> four ds_read_b128 per operand per
//   tile instead of sixteen ds_read_b32 (the summation order inside a tile is permuted, the result is the same f32 sum
//   up to reassociation).  Rows of A / B are only 4-byte aligned (K = 429): the global loads are dwordx4 in unaligned
//   mode, tails are zero-filled.
//   STATS (forward in front of BatchNorm): the epilogue also emits, per tile-row slab of the output and per column, the
//   slab's sum and its M2 = sum (x - slab mean)^2 computed from the accumulators in registers (two passes, no
//   cancellation) -- the BatchNorm statistics then need no separate pass over h (csrc/mlp.hip combines the slabs with
//   Chan's formula, in slab order: deterministic).
//   PRO (round 4, forward of layer l + 1 of an MLP): A is not read as it is but as the OUTPUT of the hidden layer in front,
//   a = dropout(relu(batch_norm(h))) (torch_rechub/basic/layers.py:281-286), formed on the way from global memory to LDS:
//   every workgroup first combines the per-slab (sum, M2) statistics the producing GEMM's STATS epilogue left behind
//   (Chan's formula in slab order, one column per thread, all loads in flight at once) into mean / rstd of all K
//   columns (LDS), then applies normalisation, ReLU and the counter-hash dropout mask to each float4 it parks in LDS.
//   The BatchNorm + ReLU + Dropout pass of csrc/mlp.hip (read h, write a) and its launch are gone; the workgroups of the
//   first tile column also write a (the weight gradient of this layer needs it) and workgroup 0 the statistics the
//   backward needs and the running statistics.  Same per-element arithmetic as bn_apply_fin_kernel<0>.
//   BNBWD (round 4, input gradient of layer l + 1): the epilogue also forms, per tile-row slab and column, the
//   BatchNorm-backward sums (sum g1, sum g1 * xhat) of layer l from the accumulators (g1 = the tile's gradient under
//   layer l's ReLU / dropout mask, recomputed from h_l) -- what bn_partial_kernel<1> read g and h again for.
// Roofline: f32 MFMA (157 TF); 0.9 GFLOP for the 4096 x 429 x 256 layer = 5.7 us at peak.
#include <type_traits>

#include "common.h"

#ifndef RH_PROBE
#define RH_PROBE 0  // tools/_probe: 1 = no global traffic inside the K loop, 2 = no MFMA
#endif

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;          // K tile
constexpr int kLd = kBK + 4;     // LDS row stride of a k-major tile: 16-byte aligned rows, conflict-light b128 reads

struct GemmArgs {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  const float* bias;  // (N,) or null
  float* C;
  int64_t ldc;
  int M, N, K;
  float* stats;  // STATS: (ceil(M / BM), 2, N): slab sum, slab M2 (BM = rows of the workgroup tile)
  int64_t* bn_rng;        // STATS, optional: the consumer's (seed, call counter) -- see rh_linear_fwd
  int64_t* bn_saved_ctr;
  int64_t* bn_batches;
  // PRO: A = dropout(relu(bn(A))) of the hidden layer in front (K columns); BNBWD: the same layer's description for the
  // epilogue sums (then over the N output columns, h = pro_h)
  const float* pro_stats;   // PRO: (ceil(M / pro_rows), 2, K) per-slab (sum, M2) of A
  int pro_rows;
  const float* pro_gamma;
  const float* pro_beta;
  float* pro_running_mean;  // PRO, optional
  float* pro_running_var;
  float pro_momentum, pro_eps, pro_p;
  const int64_t* pro_rng;   // (seed, .)
  const int64_t* pro_ctr;   // the dropout counter of that layer's call
  float* pro_stat_out;      // PRO: (>= 2, K) mean, rstd (written by workgroup 0); BNBWD: read
  float* pro_act_out;       // PRO, optional: (M, K) the activations a, written by the workgroups of tile column 0
  const float* pro_h;       // BNBWD: (M, N) pre-BatchNorm activations of the layer whose gradient the tile holds
  int64_t ldh;
  float* bwd_partial;       // BNBWD: (ceil(M / BM), 2, N)
  long long* chain_gate;    // rh_linear_fwd_gate: the deferred sweep's gate words (csrc/optim.hip), or null
  // CrossNetV2 (round 5): ep_out != null -> C = the raw product y, ep_out = ep_mul * y + bias + ep_add (x0 * (W x) + b + x,
  // torch_rechub/basic/layers.py:440-444); ep_out == null, ep_add != null -> C = product + ep_add (the residual of the input
  // gradient).  ep_* are (M, N) with leading dimension ld_ep.
  const float* ep_mul;
  const float* ep_add;
  float* ep_out;
  int64_t ld_ep;
};

constexpr int kProMaxSlabs = 32;  // PRO: slabs per thread and round of loads of the statistics prologue (common.h)
constexpr int kProMaxK = 1024;

static __device__ __forceinline__ float4 load4_guard(const float* row, int k, int K, bool row_ok) {
  if (!row_ok || k >= K) return f4_zero();
  if (k + 3 < K) return gload<float4>(row + k);
  float4 v = f4_zero();
  v.x = row[k];
  if (k + 1 < K) v.y = row[k + 1];
  if (k + 2 < K) v.z = row[k + 2];
  return v;
}

// (PRO / BNBWD run BESIDE the optimizer's resident sweep -- 2 wavefronts of 112 registers per SIMD at its default grid -- and
// must fit into what those leave free (288 registers per SIMD), or their workgroups are not placed until the sweep ends:
// measured 194 us for a 26 us kernel with a 256 + 31 register prologue.  Forcing 3 wavefronts per SIMD (168 registers) spills
// ~210 bytes per lane to scratch, so the budget is met by the prologue's shape instead: 219 + 16.)
template <int WM, int WN, bool B_KMAJOR, bool STATS, bool PRO = false, bool BNBWD = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(const GemmArgs a) {
  RH_CHAIN_PRIO();
  constexpr int NT = 64 * WM * WN;      // threads
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int A_V4 = BM * kBK / 4 / NT;              // float4 per thread per A tile
  constexpr int B_V4 = BN * kBK / 4 / NT;              // float4 per thread per B tile
  constexpr int kLdN = BN + 4;                         // row stride of the n-major B tile (B_KMAJOR = false)
  constexpr int A_TILE = BM * kLd;
  constexpr int B_TILE = B_KMAJOR ? BN * kLd : kBK * kLdN;
  __shared__ float lds[2 * (A_TILE + B_TILE)];
  // PRO: mean, rstd, gamma, beta of the K columns of A (zero past K), 4 x kp floats of DYNAMIC LDS (kp = K rounded up to the
  // K tile): the kernel runs beside the optimizer's sweep, whose workgroups are sized so that two of them leave a CU just
  // enough LDS for one workgroup of this kernel at CTR widths (csrc/optim.hip, kDeferredSweepLds)
  extern __shared__ float pro_c[];
  const int kp = ((a.K + kBK - 1) / kBK) * kBK;
  const int tid = threadIdx.x, lane = tid % RH_WAVE, wave = tid / RH_WAVE;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Renumber them so that the tiles_n
  // workgroups sharing one slab of A -- and neighbouring slabs -- sit on the same XCD: A then comes from HBM once.
  int bid = blockIdx.y * gridDim.x + blockIdx.x;
  const int nblk = gridDim.x * gridDim.y;
  if (nblk % 8 == 0) bid = (bid % 8) * (nblk / 8) + bid / 8;
  const int m0 = (bid / (int)gridDim.x) * BM, n0 = (bid % (int)gridDim.x) * BN;
  const int li = lane & 31, kk = lane >> 5;
  // rh_linear_fwd_gate: the LAST workgroup in dispatch order counts a chain start when it starts -- every workgroup of this
  // launch has been placed by then (one per CU at CTR batch sizes), which is the moment the optimizer's deferred sweep may
  // be dispatched beside the chain (stream_gate_kernel, csrc/optim.hip)
  if (a.chain_gate != nullptr && tid == 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) == nblk - 1)
    __hip_atomic_fetch_add(a.chain_gate + 2, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  float4 ra0[A_V4], rb0[B_V4], ra1[A_V4], rb1[B_V4];
  // Per-thread source pointers of its float4 pieces (rows past M / N are clamped: they only feed accumulator rows /
  // columns that are never stored).  Interior tiles use plain vector loads; only the last K tile and workgroups on the
  // N edge of an n-major B take the guarded path.
  const float* pa[A_V4];
  const float* pb[B_V4];
#pragma unroll
  for (int v = 0; v < A_V4; ++v) {
    const int e = tid + v * NT;
    const int row = m0 + e / 8;
    pa[v] = a.A + (int64_t)(row < a.M ? row : 0) * a.lda + (e % 8) * 4;
  }
#pragma unroll
  for (int v = 0; v < B_V4; ++v) {
    const int e = tid + v * NT;
    if (B_KMAJOR) {
      const int row = n0 + e / 8;
      pb[v] = a.B + (int64_t)(row < a.N ? row : 0) * a.ldb + (e % 8) * 4;
    } else {
      pb[v] = a.B + (int64_t)(e / (BN / 4)) * a.ldb + n0 + (e % (BN / 4)) * 4;
    }
  }
  // n-major B on the N edge: only the few float4 pieces that straddle column N take the guarded load
  bool bfull[B_V4];
#pragma unroll
  for (int v = 0; v < B_V4; ++v) bfull[v] = B_KMAJOR || n0 + ((tid + v * NT) % (BN / 4)) * 4 + 3 < a.N;
  float keep_scale = 1.f;
  uint32_t thr = 0;
  uint64_t seed = 0, ctr = 0;
  if (PRO || BNBWD) {
    if (a.pro_p > 0.f) {
      keep_scale = 1.f / (1.f - a.pro_p);
      thr = (uint32_t)(a.pro_p * 4294967296.0);
      seed = (uint64_t)a.pro_rng[0];
      ctr = (uint64_t)a.pro_ctr[0];
    }
  }
  // PRO: one float4 of A (row `row`, columns k .. k + 3) -> the hidden layer's output; columns past K come out as 0
  auto pro_apply = [&](float4 v, int row, int k) -> float4 {
    const float4 mean = *reinterpret_cast<const float4*>(pro_c + k);
    const float4 rstd = *reinterpret_cast<const float4*>(pro_c + kp + k);
    const float4 gam = *reinterpret_cast<const float4*>(pro_c + 2 * kp + k);
    const float4 bet = *reinterpret_cast<const float4*>(pro_c + 3 * kp + k);
    const uint64_t e = (uint64_t)row * (uint64_t)a.K + (uint64_t)k;
    auto one = [&](float x, float m, float r, float g, float b, uint64_t idx) -> float {
      const float xhat = (x - m) * r;
      float y = fmaxf(fmaf(xhat, g, b), 0.f);
      if (a.pro_p > 0.f) y = rh_drop_hash(seed, ctr, idx) >= thr ? y * keep_scale : 0.f;
      return y;
    };
    return make_float4(one(v.x, mean.x, rstd.x, gam.x, bet.x, e), one(v.y, mean.y, rstd.y, gam.y, bet.y, e + 1),
                       one(v.z, mean.z, rstd.z, gam.z, bet.z, e + 2), one(v.w, mean.w, rstd.w, gam.w, bet.w, e + 3));
  };
  auto bload = [&](int v, int k0) -> float4 {
    if (B_KMAJOR) return gload<float4>(pb[v] + k0);
    if (bfull[v]) return gload<float4>(pb[v] + (int64_t)k0 * a.ldb);
    const int c = ((tid + v * NT) % (BN / 4)) * 4;
    return load4_guard(pb[v] + (int64_t)k0 * a.ldb - (n0 + c), n0 + c, a.N, true);
  };
  auto gfetch = [&](int k0, float4* ra, float4* rb) {
    if (k0 + kBK <= a.K) {  // wavefront-uniform
#pragma unroll
      for (int v = 0; v < A_V4; ++v) ra[v] = gload<float4>(pa[v] + k0);
#pragma unroll
      for (int v = 0; v < B_V4; ++v) rb[v] = bload(v, k0);
      return;
    }
#pragma unroll
    for (int v = 0; v < A_V4; ++v) {
      const int e = tid + v * NT;            // float4 index in the (BM x 8) tile
      ra[v] = load4_guard(pa[v] - (e % 8) * 4, k0 + (e % 8) * 4, a.K, true);
    }
#pragma unroll
    for (int v = 0; v < B_V4; ++v) {
      const int e = tid + v * NT;
      if (B_KMAJOR) {
        rb[v] = load4_guard(pb[v] - (e % 8) * 4, k0 + (e % 8) * 4, a.K, true);
      } else {
        const int r = e / (BN / 4), c = (e % (BN / 4)) * 4;  // r = k within the tile, c = n within the tile
        const int k = k0 + r;
        rb[v] = load4_guard(a.B + (int64_t)(k < a.K ? k : 0) * a.ldb, n0 + c, a.N, k < a.K);
      }
    }
  };
  auto lstore = [&](int buf, const float4* ra, const float4* rb, int k0) {
    float* As = lds + buf * (A_TILE + B_TILE);
    float* Bs = As + A_TILE;
#pragma unroll
    for (int v = 0; v < A_V4; ++v) {
      const int e = tid + v * NT;
      const int r = e / 8, c = (e % 8) * 4;
      float4 av = ra[v];
      if (PRO) {
        av = pro_apply(av, m0 + r, k0 + c);
        if (a.pro_act_out != nullptr && n0 == 0 && m0 + r < a.M && k0 + c < a.K)
          gstore<float4>(a.pro_act_out + (int64_t)(m0 + r) * a.K + k0 + c, av);
      }
      *reinterpret_cast<float4*>(As + r * kLd + c) = av;
    }
#pragma unroll
    for (int v = 0; v < B_V4; ++v) {
      const int e = tid + v * NT;
      if (B_KMAJOR) {
        const int r = e / 8, c = (e % 8) * 4;
        *reinterpret_cast<float4*>(Bs + r * kLd + c) = rb[v];
      } else {
        const int r = e / (BN / 4), c = (e % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(Bs + r * kLdN + c) = rb[v];
      }
    }
  };

  // fragment reads of one K tile: 16 floats of A and of B per lane (k pairs {s, 16 + s}, see the header)
  auto lfrag = [&](int buf, float* fa, float* fb) {
    const float* As = lds + buf * (A_TILE + B_TILE) + (wm * 32 + li) * kLd + kk * 16;
    const float* Bs = lds + buf * (A_TILE + B_TILE) + A_TILE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(As + 4 * q);
      fa[4 * q] = v.x, fa[4 * q + 1] = v.y, fa[4 * q + 2] = v.z, fa[4 * q + 3] = v.w;
    }
    if (B_KMAJOR) {
      const float* Br = Bs + (wn * 32 + li) * kLd + kk * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(Br + 4 * q);
        fb[4 * q] = v.x, fb[4 * q + 1] = v.y, fb[4 * q + 2] = v.z, fb[4 * q + 3] = v.w;
      }
    } else {
      const float* Bc = Bs + (kk * 16) * kLdN + wn * 32 + li;
#pragma unroll
      for (int s = 0; s < 16; ++s) fb[s] = Bc[s * kLdN];
    }
  };

  // Software pipeline, one barrier per K tile.  At the top of iteration t: tile t's fragments are in registers, tile
  // t+1 is in LDS, tiles t+2 and t+3 are in flight from global memory (two register sets: a first touch of A costs an
  // HBM round trip, longer than the 16 MFMAs of one tile).  The iteration issues the LDS reads of tile t+1, runs the
  // MFMAs of tile t under them, parks tile t+2 in the LDS buffer tile t came from and fetches tile t+4.
  v16f acc = {};
  const int ntiles = (a.K + kBK - 1) / kBK;
  // BNBWD: this lane's 16 elements of h (the tile of the layer whose gradient the accumulators will hold) are requested
  // now and consumed in the epilogue
  float zpre[BNBWD ? 16 : 1];
  if (BNBWD) {
    const int colz = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + 4 * kk + (r & 3) + 8 * (r >> 2);
      zpre[r] = (colz < a.N && row < a.M) ? a.pro_h[(int64_t)row * a.ldh + colz] : 0.f;
    }
  }
  float fa0[16], fb0[16], fa1[16], fb1[16];
  gfetch(0, ra0, rb0);
  if (ntiles > 1) gfetch(kBK, ra1, rb1);
  // (PRO: the first two tiles are in flight while the statistics of A's columns are combined)
  if (PRO) {
    // statistics of the K columns of A: Chan's combination of the per-slab (sum, M2) pairs in slab order (common.h,
    // the arithmetic of csrc/mlp.hip::chan_combine), all loads of a column pass in flight at once
    const int nslab = (a.M + a.pro_rows - 1) / a.pro_rows;
    for (int c = a.K + tid; c < ((a.K + kBK - 1) / kBK) * kBK; c += NT)  // zero tail: columns past K transform to 0
      pro_c[c] = pro_c[kp + c] = pro_c[2 * kp + c] = pro_c[3 * kp + c] = 0.f;
    rh_combine_slabs<NT, kProMaxSlabs>(a.pro_stats, nslab, a.pro_rows, a.M, a.K, lds, tid, [&](int c, float mean, float var) {
      const float rstd = rsqrtf(var + a.pro_eps);
      pro_c[c] = mean;
      pro_c[kp + c] = rstd;
      pro_c[2 * kp + c] = a.pro_gamma[c];
      pro_c[3 * kp + c] = a.pro_beta[c];
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        a.pro_stat_out[c] = mean;
        a.pro_stat_out[a.K + c] = rstd;
        if (a.pro_running_mean != nullptr) {
          const float n = (float)a.M;
          const float unbiased = a.M > 1 ? var * (n / (n - 1.f)) : var;
          a.pro_running_mean[c] = fmaf(a.pro_momentum, mean - a.pro_running_mean[c], a.pro_running_mean[c]);
          a.pro_running_var[c] = fmaf(a.pro_momentum, unbiased - a.pro_running_var[c], a.pro_running_var[c]);
        }
      }
    });
  }
  lstore(0, ra0, rb0, 0);
  if (ntiles > 1) lstore(1, ra1, rb1, kBK);
  if (RH_PROBE != 1 && ntiles > 2) gfetch(2 * kBK, ra0, rb0);
  if (RH_PROBE != 1 && ntiles > 3) gfetch(3 * kBK, ra1, rb1);
  __syncthreads();
  lfrag(0, fa0, fb0);
  __syncthreads();  // every wavefront has tile 0 in registers before buffer 0 is overwritten below
  // One K tile.  FULL = true: the steady state (every tile it touches exists and is a full interior tile) -- straight-line
  // code, so the compiler's waitcnt insertion sees exactly which loads are outstanding; FULL = false: run-time checks.
  auto step = [&](auto full, int t, float* fa, float* fb, float* fan, float* fbn, float4* ra, float4* rb) {
    constexpr bool FULL = decltype(full)::value;
    // The dependent MFMA chain occupies the wavefront's issue slot for 16 x 64 cycles; the LDS / global work of the
    // iteration is placed INSIDE it in program order (sched_barrier pins it) so that it runs under the MFMAs
    // instead of after them.
    if (FULL || t + 1 < ntiles) lfrag((t + 1) & 1, fan, fbn);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (RH_PROBE == 2) acc[s] += fa[s] * fb[s];
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
      if (RH_PROBE != 1 && s == 3) {
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || t + 2 < ntiles) lstore(t & 1, ra, rb, (t + 2) * kBK);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (RH_PROBE != 1 && s == 7) {
        __builtin_amdgcn_sched_barrier(0);
        if (FULL) {
#pragma unroll
          for (int v = 0; v < A_V4; ++v) ra[v] = gload<float4>(pa[v] + (t + 4) * kBK);
#pragma unroll
          for (int v = 0; v < B_V4; ++v) rb[v] = bload(v, (t + 4) * kBK);
        } else if (t + 4 < ntiles) {
          gfetch((t + 4) * kBK, ra, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  int t = 0;
  const int fast_tiles = a.K / kBK;  // tiles [0, fast_tiles) are full interior tiles
  for (; t + 5 < fast_tiles; t += 2) {
    step(std::true_type{}, t, fa0, fb0, fa1, fb1, ra0, rb0);
    step(std::true_type{}, t + 1, fa1, fb1, fa0, fb0, ra1, rb1);
  }
  for (; t < ntiles; t += 2) {
    step(std::false_type{}, t, fa0, fb0, fa1, fb1, ra0, rb0);
    if (t + 1 < ntiles) step(std::false_type{}, t + 1, fa1, fb1, fa0, fb0, ra1, rb1);
  }

  // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = n0 + wn * 32 + li;
  const bool cok = col < a.N;
  const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
  const int rbase = m0 + wm * 32 + 4 * kk;
  if (a.ep_add != nullptr) {
    // CrossNetV2 epilogues: the (M, N) operands of the Hadamard / residual terms are read here, all 16 (+ 16) loads of a lane
    // in flight together
    float ea[16], em[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      const bool ok = cok && row < a.M;
      ea[r] = ok ? a.ep_add[(int64_t)row * a.ld_ep + col] : 0.f;
      em[r] = (ok && a.ep_out != nullptr) ? a.ep_mul[(int64_t)row * a.ld_ep + col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      if (!(cok && row < a.M)) continue;
      if (a.ep_out != nullptr) {
        a.C[(int64_t)row * a.ldc + col] = acc[r];                                        // y = W x (the backward needs it)
        a.ep_out[(int64_t)row * a.ld_ep + col] = fmaf(em[r], acc[r], bv) + ea[r];        // x0 * y + b + x (cross_v2_kernel)
      } else {
        a.C[(int64_t)row * a.ldc + col] = acc[r] + bv + ea[r];
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] += bv;
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      if (cok && row < a.M) a.C[(int64_t)row * a.ldc + col] = acc[r];
    }
  }
  if (STATS && a.bn_rng != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    // the BatchNorm + Dropout launch that consumes `stats` is a single launch of many blocks: it cannot advance its own
    // call counter without a grid-wide election, so the producer does it (one thread, once per forward)
    a.bn_saved_ctr[0] = a.bn_rng[1];
    a.bn_rng[1] += 1;
    if (a.bn_batches != nullptr) a.bn_batches[0] += 1;
  }
  if (STATS) {
    // slab = the BM rows of this workgroup's tile (rows past M count as absent).  Per wavefront: sum and M2 about its
    // own mean over its 32 rows (two passes over the accumulators); the WM wavefronts of a column are then merged with
    // Chan's formula through LDS (the K-loop buffers are free by now).
    const int nrows = max(0, min(32, a.M - (m0 + wm * 32)));
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (rbase + (r & 3) + 8 * (r >> 2) < a.M) ? acc[r] : 0.f;
    s += __shfl_xor(s, 32);
    const float mean = nrows > 0 ? s / (float)nrows : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[r] - mean;
      m2 += (rbase + (r & 3) + 8 * (r >> 2) < a.M) ? d * d : 0.f;
    }
    m2 += __shfl_xor(m2, 32);
    if (WM == 1) {
      if (kk == 0 && cok && nrows > 0) {
        a.stats[((int64_t)(m0 / BM) * 2 + 0) * a.N + col] = s;
        a.stats[((int64_t)(m0 / BM) * 2 + 1) * a.N + col] = m2;
      }
    } else {
      float* ex = lds;  // [WM][WN * 32][2]
      if (kk == 0) {
        ex[((wm * WN + wn) * 32 + li) * 2 + 0] = s;
        ex[((wm * WN + wn) * 32 + li) * 2 + 1] = m2;
      }
      __syncthreads();
      if (wm == 0 && kk == 0 && cok) {
        float tot = 0.f, ntot = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          tot += ex[((w * WN + wn) * 32 + li) * 2];
          ntot += (float)max(0, min(32, a.M - (m0 + w * 32)));
        }
        const float gmean = tot / ntot;
        float m2t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          const float nw = (float)max(0, min(32, a.M - (m0 + w * 32)));
          if (nw > 0.f) {
            const float d = ex[((w * WN + wn) * 32 + li) * 2] / nw - gmean;
            m2t += fmaf(nw * d, d, ex[((w * WN + wn) * 32 + li) * 2 + 1]);
          }
        }
        a.stats[((int64_t)(m0 / BM) * 2 + 0) * a.N + col] = tot;
        a.stats[((int64_t)(m0 / BM) * 2 + 1) * a.N + col] = m2t;
      }
    }
  }
  if (BNBWD) {
    // BatchNorm-backward sums of the layer whose output gradient this tile is: g1 = g under that layer's ReLU / dropout
    // mask (the arithmetic of bn_partial_kernel<1>), summed over the BM rows of the tile per column; slab order and the
    // order inside a slab are fixed -> deterministic
    float s1 = 0.f, s2 = 0.f;
    if (cok) {
      const float mean = a.pro_stat_out[col], rstd = a.pro_stat_out[a.N + col];
      const float gam = a.pro_gamma[col], bet = a.pro_beta[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < a.M) {
          const float xhat = (zpre[r] - mean) * rstd;
          const float bn = fmaf(xhat, gam, bet);
          float g1 = acc[r];
          if (a.pro_p > 0.f) g1 = rh_drop_hash(seed, ctr, (uint64_t)row * (uint64_t)a.N + (uint64_t)col) >= thr ? g1 * keep_scale : 0.f;
          g1 = bn > 0.f ? g1 : 0.f;
          s1 += g1;
          s2 = fmaf(g1, xhat, s2);
        }
      }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (WM == 1) {
      if (kk == 0 && cok) {
        a.bwd_partial[((int64_t)(m0 / BM) * 2 + 0) * a.N + col] = s1;
        a.bwd_partial[((int64_t)(m0 / BM) * 2 + 1) * a.N + col] = s2;
      }
    } else {
      float* ex = lds;  // [WM][WN * 32][2] (the K-loop buffers are free: the loop ended with a barrier)
      if (kk == 0) {
        ex[((wm * WN + wn) * 32 + li) * 2 + 0] = s1;
        ex[((wm * WN + wn) * 32 + li) * 2 + 1] = s2;
      }
      __syncthreads();
      if (wm == 0 && kk == 0 && cok) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t1 += ex[((w * WN + wn) * 32 + li) * 2];
          t2 += ex[((w * WN + wn) * 32 + li) * 2 + 1];
        }
        a.bwd_partial[((int64_t)(m0 / BM) * 2 + 0) * a.N + col] = t1;
        a.bwd_partial[((int64_t)(m0 / BM) * 2 + 1) * a.N + col] = t2;
      }
    }
  }
}

// one tile per workgroup; 64x64 tiles unless that leaves most of the 256 CUs idle
bool big_tiles(int M, int N) { return (int64_t)((M + 63) / 64) * ((N + 63) / 64) >= 192; }

// (PRO always takes the 64-row tile: half as many workgroups repeat the statistics prologue, and y's own slabs come out
// 64 rows high -- at most 64 of them for the next consumer's prologue at CTR batch sizes)
bool chain_tiles(int M) { return M > 32; }

template <bool B_KMAJOR, bool STATS, bool PRO = false, bool BNBWD = false>
void launch(const GemmArgs& a, hipStream_t s) {
  const unsigned dyn = PRO ? 4u * (unsigned)(((a.K + kBK - 1) / kBK) * kBK) * sizeof(float) : 0u;  // pro_c
  if (PRO ? chain_tiles(a.M) : big_tiles(a.M, a.N)) {
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2, B_KMAJOR, STATS, PRO, BNBWD>), dim3((a.N + 63) / 64, (a.M + 63) / 64), dim3(256), dyn,
                       s, a);
  } else {
    hipLaunchKernelGGL((gemm_f32_kernel<1, 2, B_KMAJOR, STATS, PRO, BNBWD>), dim3((a.N + 63) / 64, (a.M + 31) / 32), dim3(128), dyn,
                       s, a);
  }
}

}  // namespace

extern "C" int rh_gemm_stats_rows(int M, int N) { return big_tiles(M, N) ? 64 : 32; }
extern "C" int rh_gemm_chain_stats_rows(int M) { return chain_tiles(M) ? 64 : 32; }  // slab height of rh_linear_bnact_fwd

static int linear_fwd_impl(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                           float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr, int64_t* bn_batches,
                           int64_t* chain_gate, void* stream);

extern "C" int rh_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N,
                             int K, float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr,
                             int64_t* bn_batches, void* stream) {
  return linear_fwd_impl(x, ldx, w, ldw, bias, M, N, K, y, ldy, stats, bn_rng, bn_saved_ctr, bn_batches, nullptr, stream);
}

// rh_linear_fwd that also counts a CHAIN START in the deferred sweep's gate words (gate[2], see csrc/optim.hip) when its last
// workgroup starts: captured as the first own GEMM of a step's hipGraph, it releases the sweep of the step before into a
// chip on which this launch's workgroups are already placed.
extern "C" int rh_linear_fwd_gate(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N,
                                  int K, float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr,
                                  int64_t* bn_batches, int64_t* chain_gate, void* stream) {
  RH_REQUIRE(chain_gate != nullptr, RH_E_BADARG, "rh_linear_fwd_gate: null gate");
  return linear_fwd_impl(x, ldx, w, ldw, bias, M, N, K, y, ldy, stats, bn_rng, bn_saved_ctr, bn_batches, chain_gate, stream);
}

static int linear_fwd_impl(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                           float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr, int64_t* bn_batches,
                           int64_t* chain_gate, void* stream) {
  RH_REQUIRE(x && w && y, RH_E_BADARG, "rh_linear_fwd: null pointer");
  RH_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldx >= K && ldw >= K && ldy >= N, RH_E_BADARG,
             "rh_linear_fwd: bad shape M=%d N=%d K=%d", M, N, K);
  RH_REQUIRE(bn_rng == nullptr || (stats != nullptr && bn_saved_ctr != nullptr), RH_E_BADARG,
             "rh_linear_fwd: the BatchNorm call counter needs stats and saved_ctr");
  GemmArgs a{};
  a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.bias = bias; a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
  a.stats = stats; a.bn_rng = bn_rng; a.bn_saved_ctr = bn_saved_ctr; a.bn_batches = bn_batches;
  a.chain_gate = reinterpret_cast<long long*>(chain_gate);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stats) launch<true, true>(a, s);
  else launch<true, false>(a, s);
  RH_LAUNCH_CHECK("rh_linear_fwd");
  return 0;
}

// Layer l + 1 of an MLP in ONE launch: y = dropout(relu(batch_norm(h))) W^T + b, where h (M, K) is the PRE-BatchNorm output
// of layer l and pro_stats its per-slab (sum, M2) pairs (the `stats` of the rh_linear_fwd / rh_linear_bnact_fwd call that
// produced h; pro_rows = rh_gemm_stats_rows(M, K)), ctr the dropout counter that call drew for layer l.
// Replaces, per hidden layer: rh_bn_relu_dropout_fwd (BatchNorm1d + ReLU + Dropout, torch_rechub/basic/layers.py:283-286)
// and the read of its output by the next nn.Linear (:282).  Also written: stat_out (>= 2, K) = mean, rstd of layer l (its
// backward reads them), layer l's running statistics, act_out (M, K) = the activations (optional: the weight gradient of
// THIS layer reads them), and -- as rh_linear_fwd -- y's own slab statistics / the next BatchNorm's bookkeeping.
extern "C" int rh_linear_bnact_fwd(const float* h, int64_t ldh, int M, int K, const float* pro_stats, int pro_rows,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float p_drop, const int64_t* rng, const int64_t* ctr,
                                   float* stat_out, float* act_out, const float* w, int64_t ldw, const float* bias, int N,
                                   float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr,
                                   int64_t* bn_batches, void* stream) {
  RH_REQUIRE(h && pro_stats && gamma && beta && rng && ctr && stat_out && w && y, RH_E_BADARG,
             "rh_linear_bnact_fwd: null pointer");
  RH_REQUIRE(M >= 2 && N >= 1 && K >= 4 && K % 4 == 0 && K <= kProMaxK && ldh >= K && ldh % 4 == 0 && ldw >= K && ldy >= N &&
                 pro_rows >= 1,
             RH_E_UNSUPPORTED, "rh_linear_bnact_fwd: bad shape M=%d N=%d K=%d (K %% 4 == 0, K <= %d)", M, N, K, kProMaxK);
  RH_REQUIRE(p_drop >= 0.f && p_drop < 1.f, RH_E_BADARG, "rh_linear_bnact_fwd: p must be in [0, 1)");
  RH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), RH_E_BADARG,
             "rh_linear_bnact_fwd: running_mean and running_var go together");
  RH_REQUIRE(bn_rng == nullptr || (stats != nullptr && bn_saved_ctr != nullptr), RH_E_BADARG,
             "rh_linear_bnact_fwd: the BatchNorm call counter needs stats and saved_ctr");
  GemmArgs a{};
  a.A = h; a.lda = ldh; a.B = w; a.ldb = ldw; a.bias = bias; a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
  a.stats = stats; a.bn_rng = bn_rng; a.bn_saved_ctr = bn_saved_ctr; a.bn_batches = bn_batches;
  a.pro_stats = pro_stats; a.pro_rows = pro_rows; a.pro_gamma = gamma; a.pro_beta = beta;
  a.pro_running_mean = running_mean; a.pro_running_var = running_var; a.pro_momentum = momentum; a.pro_eps = eps;
  a.pro_p = p_drop; a.pro_rng = rng; a.pro_ctr = ctr; a.pro_stat_out = stat_out; a.pro_act_out = act_out;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stats) launch<true, true, true>(a, s);
  else launch<true, false, true>(a, s);
  RH_LAUNCH_CHECK("rh_linear_bnact_fwd");
  return 0;
}

extern "C" int rh_linear_dgrad(const float* g, int64_t ldg, const float* w, int64_t ldw, int M, int N, int K, float* gx,
                               int64_t ldgx, void* stream) {
  // gx (M, K) = g (M, N) w (N, K): a GEMM with reduction length N and output width K
  RH_REQUIRE(g && w && gx, RH_E_BADARG, "rh_linear_dgrad: null pointer");
  RH_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldg >= N && ldw >= K && ldgx >= K, RH_E_BADARG,
             "rh_linear_dgrad: bad shape M=%d N=%d K=%d", M, N, K);
  GemmArgs a{};
  a.A = g; a.lda = ldg; a.B = w; a.ldb = ldw; a.C = gx; a.ldc = ldgx; a.M = M; a.N = K; a.K = N;
  launch<false, false>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_linear_dgrad");
  return 0;
}

// rh_linear_dgrad whose epilogue also forms the BatchNorm-backward column sums of the hidden layer that produced this
// Linear's input: gx (M, K) is the gradient of a = dropout(relu(batch_norm(h))), h (M, K) that layer's pre-BatchNorm
// activations, stat (>= 2, K) its mean / rstd, ctr its dropout counter.  bwd_partial (ceil(M / R), 2, K), R =
// rh_gemm_stats_rows(M, K): per R-row slab (sum g1, sum g1 * xhat) -- what rh_bn_relu_dropout_bwd_pre takes as `partial`
// with nchunks_pre = ceil(M / R).  Replaces the statistics launch of the BatchNorm backward (bn_partial_kernel<1>).
extern "C" int rh_linear_dgrad_bnbwd(const float* g, int64_t ldg, const float* w, int64_t ldw, int M, int N, int K, float* gx,
                                     int64_t ldgx, const float* h, int64_t ldh, const float* stat, const float* gamma,
                                     const float* beta, float p_drop, const int64_t* rng, const int64_t* ctr,
                                     float* bwd_partial, void* stream) {
  RH_REQUIRE(g && w && gx && h && stat && gamma && beta && rng && ctr && bwd_partial, RH_E_BADARG,
             "rh_linear_dgrad_bnbwd: null pointer");
  RH_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldg >= N && ldw >= K && ldgx >= K && ldh >= K, RH_E_BADARG,
             "rh_linear_dgrad_bnbwd: bad shape M=%d N=%d K=%d", M, N, K);
  RH_REQUIRE(p_drop >= 0.f && p_drop < 1.f, RH_E_BADARG, "rh_linear_dgrad_bnbwd: p must be in [0, 1)");
  GemmArgs a{};
  a.A = g; a.lda = ldg; a.B = w; a.ldb = ldw; a.C = gx; a.ldc = ldgx; a.M = M; a.N = K; a.K = N;
  a.pro_h = h; a.ldh = ldh; a.pro_stat_out = const_cast<float*>(stat); a.pro_gamma = gamma; a.pro_beta = beta;
  a.pro_p = p_drop; a.pro_rng = rng; a.pro_ctr = ctr; a.bwd_partial = bwd_partial;
  launch<false, false, false, true>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_linear_dgrad_bnbwd");
  return 0;
}

constexpr int kCrossV2MaxM = 16384, kCrossV2MaxD = 1024;  // = ops.CROSS_V2_MAX_B / CROSS_V2_MAX_D

// One CrossNetV2 layer forward in ONE launch (round 5; reference CrossNetV2.forward, torch_rechub/basic/layers.py:440-444:
// x <- x0 * (W_l x) + b_l + x): y (M, d) = x W^T on the f32-MFMA tile GEMM, and from the accumulators out = x0 * y + b + x.
// Rounds 1-4 ran the product on the library and the Hadamard + bias + residual as a second pass over four (M, d) arrays
// (rh_cross_v2_epilogue_fwd).  y is stored too: the backward's g_x0 = g * y needs it.
extern "C" int rh_cross_v2_fwd(const float* x0, const float* x, const float* w, const float* b, int M, int d, float* y,
                               float* out, void* stream) {
  RH_REQUIRE(x0 && x && w && b && y && out, RH_E_BADARG, "rh_cross_v2_fwd: null pointer");
  // (all operands contiguous: row stride d; the limits of the tile GEMM at batch size, ops.CROSS_V2_MAX_B / _MAX_D)
  RH_REQUIRE(M >= 1 && M <= kCrossV2MaxM && d >= 1 && d <= kCrossV2MaxD, RH_E_UNSUPPORTED,
             "rh_cross_v2_fwd: bad shape M=%d d=%d (1 <= M <= %d, 1 <= d <= %d)", M, d, kCrossV2MaxM, kCrossV2MaxD);
  GemmArgs a{};
  a.A = x; a.lda = d; a.B = w; a.ldb = d; a.bias = b; a.C = y; a.ldc = d; a.M = M; a.N = d; a.K = d;
  a.ep_mul = x0; a.ep_add = x; a.ep_out = out; a.ld_ep = d;
  launch<true, false>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_cross_v2_fwd");
  return 0;
}

// ... and the input gradient of the layer: gx (M, d) = g_y W + g, g_y = g * x0 the gradient of the product (formed by
// rh_cross_v2_epilogue_bwd together with g_x0 = g * y), g the upstream gradient that also reaches x through the residual.
extern "C" int rh_cross_v2_dgrad(const float* g_y, const float* w, const float* g, int M, int d, float* gx, void* stream) {
  RH_REQUIRE(g_y && w && g && gx, RH_E_BADARG, "rh_cross_v2_dgrad: null pointer");
  RH_REQUIRE(M >= 1 && M <= kCrossV2MaxM && d >= 1 && d <= kCrossV2MaxD, RH_E_UNSUPPORTED,
             "rh_cross_v2_dgrad: bad shape M=%d d=%d (1 <= M <= %d, 1 <= d <= %d)", M, d, kCrossV2MaxM, kCrossV2MaxD);
  GemmArgs a{};
  a.A = g_y; a.lda = d; a.B = w; a.ldb = d; a.C = gx; a.ldc = d; a.M = M; a.N = d; a.K = d;
  a.ep_add = g; a.ld_ep = d;
  launch<false, false>(a, reinterpret_cast<hipStream_t>(stream));
  RH_LAUNCH_CHECK("rh_cross_v2_dgrad");
  return 0;
}
