// The split-batch weight gradient dW = g^T x of an nn.Linear (torch_rechub/basic/layers.py:279,290) on the f32 MFMA: the
// workgroup body and its argument blocks, shared by csrc/linear.hip (rh_linear_wgrad*, the grouped launch) and csrc/optim.hip
// (round 6: the MLP chain's grouped weight gradients ride in the optimizer's end-of-step launch, adam_lazy_step_ahead_kernel).
// Description of the kernel: csrc/linear.hip, item 1.
#pragma once
#include "common.h"

namespace rh_wgrad {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kTile = 64;               // output tile edge per block (2 x 2 MFMA tiles of 32 x 32 per wave)
constexpr int kWaves = RH_BLOCK / RH_WAVE;
constexpr int kTileElems = kTile * kTile;
constexpr int kPartStride = kTileElems + kTile;  // tile + its db slice
constexpr int kUnroll = 4;              // row pairs in flight per wave

struct WgradArgs {
  const float* g;  // (B, N), row stride ldg
  int64_t ldg;
  const float* x;  // (B, K), row stride ldx
  int64_t ldx;
  int B, N, K;
  int S, rows_per_split;
  float* partial;      // split s: dW part at partial + s * N * K, db part at partial + S * N * K + s * N
  float* dW;           // (N, K) contiguous
  float* db;           // (N,) or null
  int direct;          // 1: S == 1 and the block writes dW / db itself
};

// LONG: the build for long reductions (see the two kernels below): simple prefetch loop and ONE LDS tile; otherwise the
// round-1 form (ping-pong register sets, one LDS tile per wavefront).  Same sums in the same order either way.
template <bool LONG>
__device__ __forceinline__ void linear_wgrad_body(const WgradArgs& a, float* red, const int bx, const int by, const int s) {
  RH_CHAIN_PRIO();
  const int lane = threadIdx.x % RH_WAVE, wave = threadIdx.x / RH_WAVE;
  const int half = lane >> 5, c = lane & 31;
  // (Measured and dropped: a 1-D launch that puts the tiles of one split on ONE XCD, so that its L2 serves the rows they
  // share -- 945 us against 884 for DIN's four long launches; with x-fastest tiles each XCD streams its own column range.)
  const int k0 = bx * kTile, n0 = by * kTile;
  const int b_lo = s * a.rows_per_split;
  const int b_hi = min(a.B, b_lo + a.rows_per_split);

  // Columns past N / K are clamped to column 0: their products land in tile rows / columns that are never stored, so
  // the inner loop needs no column masks (and stays free of exec-masked loads).
  const int na0 = n0 + c, na1 = n0 + 32 + c, kb0 = k0 + c, kb1 = k0 + 32 + c;
  const float* ga0 = a.g + (na0 < a.N ? na0 : 0) + (int64_t)half * a.ldg;
  const float* ga1 = a.g + (na1 < a.N ? na1 : 0) + (int64_t)half * a.ldg;
  const float* xb0 = a.x + (kb0 < a.K ? kb0 : 0) + (int64_t)half * a.ldx;
  const float* xb1 = a.x + (kb1 < a.K ? kb1 : 0) + (int64_t)half * a.ldx;

  v16f acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
  float bs0 = 0.f, bs1 = 0.f;
  float fa0[kUnroll], fa1[kUnroll], fb0[kUnroll], fb1[kUnroll];
  float qa0[kUnroll], qa1[kUnroll], qb0[kUnroll], qb1[kUnroll];
  // wave w takes row pairs w, w + 4, ...; one iteration = kUnroll pairs = 16 dword loads, fetched one iteration ahead
  auto fetch = [&](int p, float* A0, float* A1, float* B0, float* B1) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = p + 2 * kWaves * u;
      A0[u] = gload<float>(ga0 + r * a.ldg);
      A1[u] = gload<float>(ga1 + r * a.ldg);
      B0[u] = gload<float>(xb0 + r * a.ldx);
      B1[u] = gload<float>(xb1 + r * a.ldx);
    }
  };
  auto issue = [&](const float* A0, const float* A1, const float* B0, const float* B1) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B0[u], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B1[u], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B0[u], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B1[u], acc11, 0, 0, 0);
      bs0 += A0[u];
      bs1 += A1[u];
    }
  };
  constexpr int kStep = 2 * kWaves * kUnroll;
  int p = b_lo + 2 * wave;
  const int last_full = b_hi - (2 * kWaves * (kUnroll - 1) + 2);  // p <= last_full: every row of the iteration exists
  if (p <= last_full) {
    fetch(p, fa0, fa1, fb0, fb1);
    if (LONG) {
      for (; p + kStep <= last_full; p += kStep) {
        fetch(p + kStep, qa0, qa1, qb0, qb1);
        issue(fa0, fa1, fb0, fb1);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          fa0[u] = qa0[u];
          fa1[u] = qa1[u];
          fb0[u] = qb0[u];
          fb1[u] = qb1[u];
        }
      }
      issue(fa0, fa1, fb0, fb1);
      p += kStep;
    } else {
      while (true) {  // ping-pong between the two register sets; the branch conditions are wave-uniform
        if (p + kStep > last_full) {
          issue(fa0, fa1, fb0, fb1);
          p += kStep;
          break;
        }
        fetch(p + kStep, qa0, qa1, qb0, qb1);
        issue(fa0, fa1, fb0, fb1);
        p += kStep;
        if (p + kStep > last_full) {
          issue(qa0, qa1, qb0, qb1);
          p += kStep;
          break;
        }
        fetch(p + kStep, fa0, fa1, fb0, fb1);
        issue(qa0, qa1, qb0, qb1);
        p += kStep;
      }
    }
  }
  // ragged end of the last split: fewer than kStep rows, guarded per row
  for (; p < b_hi; p += 2 * kWaves) {
    const bool ok = p + half < b_hi;
    const int64_t r = ok ? p : b_lo - half;  // (the fragment pointers already carry + half rows)
    const float m = ok ? 1.f : 0.f;
    const float t0 = gload<float>(ga0 + r * a.ldg) * m, t1 = gload<float>(ga1 + r * a.ldg) * m;
    const float t2 = gload<float>(xb0 + r * a.ldx) * m, t3 = gload<float>(xb1 + r * a.ldx) * m;
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(t0, t2, acc00, 0, 0, 0);
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(t0, t3, acc01, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(t1, t2, acc10, 0, 0, 0);
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(t1, t3, acc11, 0, 0, 0);
    bs0 += t0;
    bs1 += t1;
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  // LONG: the four wavefronts add their tiles into ONE LDS tile, one after the other in wavefront order (deterministic, and the
  // same sum ((w0 + w1) + w2) + w3 as four separate tiles summed afterwards).  Four tiles were 66.5 KB per workgroup = two
  // workgroups per CU = two wavefronts per SIMD, too few to hide the operand loads of a long reduction (DIN: 409 600 rows,
  // ~70 TF); one tile is 16.6 KB.
  bs0 += __shfl_xor(bs0, 32);
  bs1 += __shfl_xor(bs1, 32);
  if (!LONG) {
    float* mine = red + wave * kPartStride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      mine[row * kTile + c] = acc00[r];
      mine[row * kTile + 32 + c] = acc01[r];
      mine[(32 + row) * kTile + c] = acc10[r];
      mine[(32 + row) * kTile + 32 + c] = acc11[r];
    }
    if (half == 0) {
      mine[kTileElems + c] = bs0;
      mine[kTileElems + 32 + c] = bs1;
    }
    __syncthreads();
  }
  for (int w = 0; LONG && w < kWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float* q = red + row * kTile + c;
        if (w == 0) {
          q[0] = acc00[r];
          q[32] = acc01[r];
          q[32 * kTile] = acc10[r];
          q[32 * kTile + 32] = acc11[r];
        } else {
          q[0] += acc00[r];
          q[32] += acc01[r];
          q[32 * kTile] += acc10[r];
          q[32 * kTile + 32] += acc11[r];
        }
      }
      if (half == 0) {
        if (w == 0) {
          red[kTileElems + c] = bs0;
          red[kTileElems + 32 + c] = bs1;
        } else {
          red[kTileElems + c] += bs0;
          red[kTileElems + 32 + c] += bs1;
        }
      }
    }
    __syncthreads();
  }
  // Partial results of split s in the layout of the outputs themselves -- (N, K) and (N,) slabs, one per split -- so
  // that summing the splits is a plain slab sum for whoever does it (wgrad_reduce_kernel, or rh_pack_grads fused with
  // the packing of the step's dense gradients).
  float* outW = a.direct ? a.dW : a.partial + (int64_t)s * a.N * a.K;
  float* outB = a.direct ? a.db : a.partial + (int64_t)a.S * a.N * a.K + (int64_t)s * a.N;
  for (int e = threadIdx.x; e < kPartStride; e += RH_BLOCK) {
    float v = red[e];
    if (!LONG) {
#pragma unroll
      for (int w = 1; w < kWaves; ++w) v += red[w * kPartStride + e];
    }
    if (e < kTileElems) {
      const int n = n0 + e / kTile, k = k0 + e % kTile;
      if (n < a.N && k < a.K) outW[(int64_t)n * a.K + k] = v;
    } else if (outB && bx == 0 && n0 + e - kTileElems < a.N) {
      outB[n0 + e - kTileElems] = v;
    }
  }
}

// Several independent weight-gradient problems as ONE launch (round 4): the backward of CrossNetMix leaves two per layer
// (g_UTb = g_Y^T wp, g_VgT = g_PG^T x_l), none of which anything else in the backward waits for -- eight launches of
// 16.5 us each were a sixth of the DCN-v2 step.  Workgroup b belongs to problem i with prefix[i] <= b < prefix[i + 1]; inside
// a problem the workgroups are numbered tile-column fastest, then tile row, then split, as the 3-D grid of the single launch.
constexpr int kWgradGroup = 8;
struct WgradGroupArgs {
  WgradArgs p[kWgradGroup];
  int prefix[kWgradGroup + 1];
  int tiles_k[kWgradGroup], tiles_n[kWgradGroup];
  int n;
};

// workgroup b of a grouped launch -> its problem and tile (numbering as documented above)
template <bool LONG>
__device__ __forceinline__ void linear_wgrad_group_body(const WgradGroupArgs& ga, float* red, const int b) {
  int i = 0;
#pragma unroll
  for (int q = 1; q < kWgradGroup; ++q) i += (q < ga.n && b >= ga.prefix[q]) ? 1 : 0;
  const int local = b - ga.prefix[i];
  const int tk = ga.tiles_k[i], tn = ga.tiles_n[i];
  linear_wgrad_body<LONG>(ga.p[i], red, local % tk, (local / tk) % tn, local / (tk * tn));
}

}  // namespace rh_wgrad

// Host side (csrc/linear.hip): validates n <= 8 batch-sized problems and lays them out as ONE grouped launch -- the plan
// (tiles, splits) of rh_linear_wgrad_partial per problem, hence the same slabs bit for bit whoever launches the group.
int rh_wgrad_group_fill(int n, const float* const* g, const int64_t* ldg, const float* const* x, const int64_t* ldx, const int* B,
                        const int* N, const int* K, float* const* partial, rh_wgrad::WgradGroupArgs* out, const char* who);
