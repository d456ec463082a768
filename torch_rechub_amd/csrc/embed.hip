// Fused multi-field embedding gather + FM + LR (forward, backward, row scatter) for gfx950.
//
// Reference op chains replaced (paths relative to /root/reference):
//   EmbeddingLayer.forward  torch_rechub/basic/layers.py:77-127
//   FM.forward              torch_rechub/basic/layers.py:313-319
//   LR.forward              torch_rechub/basic/layers.py:185-189
//   composed by DeepFM.forward torch_rechub/models/ranking/deepfm.py:34-43
//
// Roofline: HBM.  A D=16 fp32 row is one 64-byte access, so the forward is a random-gather
// kernel: lanes are grouped LPR = D/4 per row (one 16-byte load each), FS row-groups per sample,
// all of a lane's gathers are issued before the first is consumed (index phase -> gather phase ->
// use phase) and the per-sample FM / LR reductions are wavefront xor-shuffles.  The backward is
// field-major so that tiny tables (Criteo has vocab 3, 4, 10 ...) are pre-aggregated in LDS:
// device-scope atomics on one address serialise at ~11 ns each on MI355X, LDS atomics do not.
#include <stdlib.h>

#include "common.h"

namespace {

struct EmbedFwdArgs {
  const int64_t* fdesc;
  const int64_t* idesc;
  const int64_t* ddesc;
  int B, F, D, ND;
  int dense_col;  // first output column of the dense block
  float* out;
  int64_t out_stride;
  const float* lr_w;
  const float* lr_b;
  float* lr_out;
  float* fm_out;
  float* s_out;
  int* err;
};

// One sample is handled by G = LPR*FS lanes: lane (fs, q) gathers dwords [4q, 4q+4) of the rows
// of fields f = j*FS + fs, j = 0..ceil(F/FS)-1.  Groups never straddle a wavefront (G | 64).
template <int LPR, int FS, typename IdxT, bool HAS_LR>
__global__ __launch_bounds__(RH_BLOCK) void embed_fwd_kernel(const EmbedFwdArgs a) {
  RH_CHAIN_PRIO();
  constexpr int G = LPR * FS;
  constexpr int SPB = RH_BLOCK / G;
  constexpr int U = 8;  // gathers in flight per lane
  const int tid = threadIdx.x;
  const int lig = tid % G;
  const int q = lig % LPR;
  const int fs = lig / LPR;
  int64_t b = (int64_t)blockIdx.x * SPB + tid / G;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;  // keep the wavefront converged for the shuffles; results discarded
  const int F = a.F, D = a.D;
  const int nfl = (F + FS - 1) / FS;

  float4 S = f4_zero();
  float Qs = 0.f, Ls = 0.f;
  bool oob_any = false;
  float* orow = a.out + b * a.out_stride + q * 4;
  // The LR weights (F x D floats, the same for every sample) wait in LDS instead of 8 float4 registers per lane: the
  // kernel drops from 144 to 114 VGPRs, i.e. from 3 to 4 wavefronts per SIMD of gathers in flight.
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  if (HAS_LR) {
    for (int i = tid; i < F * LPR; i += RH_BLOCK)
      *reinterpret_cast<float4*>(wlds + 4 * i) = gload<float4>(a.lr_w + 4 * i);
  }
  bool staged = !HAS_LR;

  // dense values appended after the sparse block (layers.py:120 order).  The first column of every lane rides along
  // with the gathers (descriptor with the field descriptors, value with the indices): as a tail loop after the
  // gathers it was two more dependent round trips (measured at B = 65536, 13 dense columns: 49.6 -> 64.7 us).
  const bool dense0 = a.ND > 0 && lig < a.ND;
  const float* dp0 = nullptr;
  int64_t ds0 = 0;
  float dv0 = 0.f;
  if (dense0) {
    dp0 = reinterpret_cast<const float*>(a.ddesc[lig]);
    ds0 = a.ddesc[a.ND + lig];
  }

  for (int j0 = 0; j0 < nfl; j0 += U) {
    int fcl[U];
    bool ok[U];
    const IdxT* ip[U];
    int64_t st[U];
    const float* tab[U];
    int64_t voc[U];
    int col[U];
    // phase 0: descriptors (L1/L2 resident)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = (j0 + u) * FS + fs;
      ok[u] = f < F;
      fcl[u] = ok[u] ? f : F - 1;
      ip[u] = reinterpret_cast<const IdxT*>(a.idesc[fcl[u]]);
      st[u] = a.idesc[F + fcl[u]];
      tab[u] = reinterpret_cast<const float*>(a.fdesc[fcl[u]]);
      voc[u] = a.fdesc[2 * F + fcl[u]];
      col[u] = (int)a.idesc[2 * F + fcl[u]] * D;  // output slot of the field -> column
    }
    // phase 1: indices
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = (int64_t)gload<IdxT>(ip[u] + b * st[u]);
    if (j0 == 0 && dense0) dv0 = gload<float>(dp0 + b * ds0);
    // phase 2: row gathers (16 B per lane), all in flight together
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool oob = (uint64_t)row[u] >= (uint64_t)voc[u];
      oob_any |= (oob && ok[u]);
      const int64_t r = oob ? 0 : row[u];
      v[u] = gload<float4>(tab[u] + r * D + q * 4);
    }
    if (!staged) {  // block-uniform: the staged LR weights become visible (first pass only)
      __syncthreads();
      staged = true;
    }
    // phase 3: accumulate (branch-free selects) + emit
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 vm = ok[u] ? v[u] : f4_zero();
      S = f4_add(S, vm);
      Qs += f4_dot(vm, vm);
      if (HAS_LR) Ls += f4_dot(*reinterpret_cast<const float4*>(wlds + (fcl[u] * LPR + q) * 4), vm);
      if (ok[u] && live) gstore<float4>(orow + col[u], v[u]);
    }
  }

  if (dense0 && live) a.out[b * a.out_stride + a.dense_col + lig] = dv0;
  if (a.ND > G && live) {  // more dense columns than lanes per sample: the rest in a tail loop
    for (int j = lig + G; j < a.ND; j += G) {
      const float* dp = reinterpret_cast<const float*>(a.ddesc[j]);
      const int64_t ds = a.ddesc[a.ND + j];
      a.out[b * a.out_stride + a.dense_col + j] = gload<float>(dp + b * ds);
    }
  }

  // S over the FS row-groups of the sample
#pragma unroll
  for (int m = LPR; m < G; m <<= 1) S = f4_add(S, f4_shfl_xor(S, m));
  float t = f4_dot(S, S);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, RH_WAVE);
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    Qs += __shfl_xor(Qs, m, RH_WAVE);
    Ls += __shfl_xor(Ls, m, RH_WAVE);
  }
  if (live) {
    if (a.s_out != nullptr && fs == 0) gstore<float4>(a.s_out + b * D + q * 4, S);
    if (lig == 0) {
      if (a.fm_out != nullptr) a.fm_out[b] = 0.5f * (t - Qs);
      if (a.lr_out != nullptr) a.lr_out[b] = Ls + (a.lr_b != nullptr ? a.lr_b[0] : 0.f);
    }
  }
  if (oob_any && a.err != nullptr) atomicOr(a.err, RH_FLAG_INDEX_OOB);
}

// Field-uniform kernel: a wavefront holds 64 / LPR samples (LPR lanes each) and walks ITS share of the fields -- the
// WS wavefronts of a workgroup split the fields of the same samples between them -- so the field is wavefront-uniform
// and its descriptors are scalar loads from the constant cache (SGPRs).  In the lane-split kernel above the five
// per-lane descriptor loads per gathered row are 35 of the 56 vector-memory instructions of a wavefront and one more
// dependent round trip; here a wavefront issues ceil(F / WS) index loads, as many row gathers, as many stores, and the
// chain is index -> row -> store.  Measured at B = 65536 (tools/fwd_probe.py, index sets cycled): what matters is the
// number of dependent phases per wavefront (26 fields in 4 phases of 8: 60 us; 1 phase of 26 at 2 wavefronts per SIMD:
// 48 us), hence the split of the fields over wavefronts: ONE phase of <= 8 gathers per lane at full occupancy.
// FM / LR partial sums run in the lane over its fields (field order), across the LPR lanes by xor-shuffle, across the
// WS wavefronts through LDS (wavefront order).
#define RH_CONST __attribute__((address_space(4)))
static __device__ __forceinline__ int64_t desc_at(const int64_t* p, int i) {
  return reinterpret_cast<const RH_CONST int64_t*>(reinterpret_cast<uintptr_t>(p))[i];
}

template <int LPR, int WS, typename IdxT, bool HAS_LR>
__global__ __launch_bounds__(RH_WAVE * WS) void embed_fwd_uniform_kernel(const EmbedFwdArgs a) {
  constexpr int SPW = RH_WAVE / LPR;  // samples per wavefront = per workgroup
  constexpr int NT = RH_WAVE * WS;
  constexpr int U = 8;   // gathers in flight per lane: ONE phase for F <= 8 * WS fields
  constexpr int DU = 4;  // dense columns per lane fetched with the first indices (last wavefront)
  constexpr int D = 4 * LPR;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / RH_WAVE);
  const int lane = tid % RH_WAVE;
  const int q = lane % LPR;
  int64_t b = (int64_t)blockIdx.x * SPW + lane / LPR;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int F = a.F;
  const int per = (F + WS - 1) / WS;  // fields of a wavefront: [fbeg, fend)
  const int fbeg = wave * per < F ? wave * per : F;
  const int fend = fbeg + per < F ? fbeg + per : F;
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  float* wlds = dyn_lds;                                   // HAS_LR: F * D weights
  float* red = dyn_lds + (HAS_LR ? ((F * D + 3) / 4) * 4 : 0);  // WS > 1: [WS][64][6] partial S, Qs, Ls
  if (HAS_LR) {
    for (int i = tid; i < F * LPR; i += NT) *reinterpret_cast<float4*>(wlds + 4 * i) = gload<float4>(a.lr_w + 4 * i);
  }
  float4 S = f4_zero();
  float Qs = 0.f, Ls = 0.f;
  bool oob_any = false;
  float* orow = a.out + b * a.out_stride + q * 4;

  const bool dense_wave = wave == WS - 1;  // the last wavefront has the fewest fields
  float dv[DU];
#pragma unroll
  for (int k = 0; k < DU; ++k) {
    const int j = q + k * LPR;
    dv[k] = 0.f;
    if (dense_wave && j < a.ND) dv[k] = gload<float>(reinterpret_cast<const float*>(a.ddesc[j]) + b * a.ddesc[a.ND + j]);
  }

  int64_t r[U];  // dead once the gathers of the phase are issued: the next phase's indices land in the same registers
  auto load_idx = [&](int f0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      r[u] = 0;
      if (f < fend) {  // wavefront-uniform
        const IdxT* ip = reinterpret_cast<const IdxT*>(desc_at(a.idesc, f));
        r[u] = (int64_t)gload<IdxT>(ip + b * desc_at(a.idesc, F + f));
      }
    }
  };
  auto phase = [&](int f0, bool first) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      v[u] = f4_zero();
      if (f < fend) {
        const float* tab = reinterpret_cast<const float*>(desc_at(a.fdesc, f));
        const bool oob = (uint64_t)r[u] >= (uint64_t)desc_at(a.fdesc, 2 * F + f);
        oob_any |= oob;
        v[u] = gload<float4>(tab + (oob ? 0 : r[u]) * D + q * 4);
      }
    }
    if (f0 + U < fend) load_idx(f0 + U);
    if (HAS_LR && first) __syncthreads();  // the staged LR weights (every wavefront runs the first phase)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      if (f < fend) {
        S = f4_add(S, v[u]);
        Qs += f4_dot(v[u], v[u]);
        if (HAS_LR) Ls += f4_dot(*reinterpret_cast<const float4*>(wlds + (f * LPR + q) * 4), v[u]);
        if (live) gstore<float4>(orow + (int)desc_at(a.idesc, 2 * F + f) * D, v[u]);
      }
    }
  };
  load_idx(fbeg);
  phase(fbeg, true);
#pragma unroll 1
  for (int f0 = fbeg + U; f0 < fend; f0 += U) phase(f0, false);

  if (live && dense_wave) {
#pragma unroll
    for (int k = 0; k < DU; ++k) {
      const int j = q + k * LPR;
      if (j < a.ND) a.out[b * a.out_stride + a.dense_col + j] = dv[k];
    }
    for (int j = q + DU * LPR; j < a.ND; j += LPR)
      a.out[b * a.out_stride + a.dense_col + j] =
          gload<float>(reinterpret_cast<const float*>(a.ddesc[j]) + b * a.ddesc[a.ND + j]);
  }
  if (oob_any && a.err != nullptr) atomicOr(a.err, RH_FLAG_INDEX_OOB);

  if (WS > 1) {
    float* mine = red + (wave * RH_WAVE + lane) * 6;
    mine[0] = S.x, mine[1] = S.y, mine[2] = S.z, mine[3] = S.w;
    mine[4] = Qs;
    mine[5] = Ls;
    __syncthreads();
    if (wave != 0) return;
    S = f4_zero();
    Qs = Ls = 0.f;
#pragma unroll
    for (int w = 0; w < WS; ++w) {
      const float* o = red + (w * RH_WAVE + lane) * 6;
      S = f4_add(S, make_float4(o[0], o[1], o[2], o[3]));
      Qs += o[4];
      Ls += o[5];
    }
  }
  float t = f4_dot(S, S);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) {
    t += __shfl_xor(t, m, RH_WAVE);
    Qs += __shfl_xor(Qs, m, RH_WAVE);
    Ls += __shfl_xor(Ls, m, RH_WAVE);
  }
  if (live) {
    if (a.s_out != nullptr) gstore<float4>(a.s_out + b * D + q * 4, S);
    if (q == 0) {
      if (a.fm_out != nullptr) a.fm_out[b] = 0.5f * (t - Qs);
      if (a.lr_out != nullptr) a.lr_out[b] = Ls + (a.lr_b != nullptr ? a.lr_b[0] : 0.f);
    }
  }
}

int g_fwd_path = 0;  // tuning knob RH_TUNE_FWD_PATH: 0 auto (by batch size), 1 lane-split kernel, 2 field-uniform kernel
// B * F from which the field-uniform kernel wins (rocprofv3 kernel durations, F = 26, D = 16: B = 4096 6.4 vs 8.2 us,
// 8192 9.1 vs 9.5, 16384 14.7 vs 12.2, 65536 59.8 vs 50.3): below, its extra barrier + LDS exchange and the cold
// scalar cache cost more than the descriptor loads it saves
constexpr int64_t kUniformMinLookups = 12288 * 26;
constexpr int kUniformWaves = 4;  // measured at B = 65536: 8 / 4 / 2 / 1 wavefronts per sample group = 58.6 / 50.3 / 52.5 / 58.3 us

template <int LPR, int WS, typename IdxT>
int launch_fwd_uniform(const EmbedFwdArgs& a, hipStream_t s) {
  constexpr int SPW = RH_WAVE / LPR;
  const unsigned grid = (unsigned)((a.B + SPW - 1) / SPW);
  const size_t red = WS > 1 ? (size_t)WS * RH_WAVE * 6 * sizeof(float) : 0;
  if (a.lr_w != nullptr)
    hipLaunchKernelGGL((embed_fwd_uniform_kernel<LPR, WS, IdxT, true>), dim3(grid), dim3(RH_WAVE * WS),
                       (size_t)((a.F * a.D + 3) / 4) * 4 * sizeof(float) + red, s, a);
  else
    hipLaunchKernelGGL((embed_fwd_uniform_kernel<LPR, WS, IdxT, false>), dim3(grid), dim3(RH_WAVE * WS), red, s, a);
  return 0;
}

template <int WS, typename IdxT>
int dispatch_uniform(const EmbedFwdArgs& a, hipStream_t s) {
  switch (a.D / 4) {
    case 1: return launch_fwd_uniform<1, WS, IdxT>(a, s);
    case 2: return launch_fwd_uniform<2, WS, IdxT>(a, s);
    case 4: return launch_fwd_uniform<4, WS, IdxT>(a, s);
    case 8: return launch_fwd_uniform<8, WS, IdxT>(a, s);
    case 16: return launch_fwd_uniform<16, WS, IdxT>(a, s);
    case 32: return launch_fwd_uniform<32, WS, IdxT>(a, s);
    default: return RH_E_UNSUPPORTED;
  }
}

template <int LPR, int FS, typename IdxT>
int launch_fwd(const EmbedFwdArgs& a, hipStream_t s) {
  constexpr int SPB = RH_BLOCK / (LPR * FS);
  const unsigned grid = (unsigned)((a.B + SPB - 1) / SPB);
  if (a.lr_w != nullptr)
    hipLaunchKernelGGL((embed_fwd_kernel<LPR, FS, IdxT, true>), dim3(grid), dim3(RH_BLOCK),
                       (size_t)a.F * a.D * sizeof(float), s, a);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<LPR, FS, IdxT, false>), dim3(grid), dim3(RH_BLOCK), 0, s, a);
  return 0;
}

template <int LPR, typename IdxT>
int dispatch_fs(const EmbedFwdArgs& a, int fs, hipStream_t s) {
  if constexpr (LPR * 8 <= RH_WAVE) {
    if (fs == 8) return launch_fwd<LPR, 8, IdxT>(a, s);
  }
  if constexpr (LPR * 4 <= RH_WAVE) {
    if (fs >= 4) return launch_fwd<LPR, 4, IdxT>(a, s);
  }
  if constexpr (LPR * 2 <= RH_WAVE) {
    if (fs >= 2) return launch_fwd<LPR, 2, IdxT>(a, s);
  }
  return launch_fwd<LPR, 1, IdxT>(a, s);
}

template <typename IdxT>
int dispatch_lpr(const EmbedFwdArgs& a, int fs, hipStream_t s) {
  switch (a.D / 4) {
    case 1: return dispatch_fs<1, IdxT>(a, fs, s);
    case 2: return dispatch_fs<2, IdxT>(a, fs, s);
    case 4: return dispatch_fs<4, IdxT>(a, fs, s);
    case 8: return dispatch_fs<8, IdxT>(a, fs, s);
    case 16: return dispatch_fs<16, IdxT>(a, fs, s);
    case 32: return dispatch_fs<32, IdxT>(a, fs, s);
    default: return RH_E_UNSUPPORTED;
  }
}

// ------------------------------------------------------------------------------------------
struct EmbedBwdArgs {
  const int64_t* fdesc;
  const int64_t* idesc;
  int B, F, D;
  const float* g_out;
  int64_t g_stride;
  const float* emb;
  int64_t emb_stride;
  const float* s_sum;
  const float* g_fm;
  const float* g_lr;
  const float* lr_w;
  float* lr_wgrad;
  float scale;
  float* rows_out;
  const float* rows_in;
  int spb;
  int wide_atomics;
  int path;  // experiment knob: 0 auto, 1 global atomics for every table, 3 no sink, 4 chunk-fastest block order
  int nchunks8;  // sample chunks rounded up to a multiple of 8
  int* err;
};

constexpr int kSweeps = 2;  // phase-B sweeps of the small-table path: covers kSweeps * 4 * (64 / D) rows (32 at D = 16)

// grid = (sample chunks, fields).  SRC 0: compute the gradient row from the upstream gradients,
// 1: read it from rows_in.  SINK 0: scatter-add into the table gradient, 1: write to rows_out.  (Round 3 also built a
// row-list sink -- plain stores of every lookup's row + a hash-linked chain per table row -- and measured it slower at every
// batch size, 21.9 / 47.0 / 159 us against 11.9 / 30.2 / 103 us: removed, DESIGN 3.2 keeps the numbers.)
//
// Small tables (Criteo has vocab 3, 4, 10, 15, 18, 24, 27, 105): every lookup of the batch lands on a few rows, and what
// is expensive on MI355X is contention, measured at B = 65536 (tools/bwd_field_probe.py):
//   * device-scope atomics on ONE cache line serialise at ~11 ns per request (a 3-row table: 1 ms; 105 rows: +9 us);
//   * LDS float atomics (ds_add_f32) retire at ~5 cycles per LANE, conflict-free or not (the 9 tables of <= 512 rows
//     LDS-aggregated: 126 us; with 16 private copies: 84 us);
//   * one float4 accumulator per table row in every lane (one-hot FMA) needs 128 registers: occupancy 2, 62-80 us.
//   * walking the lookups one by one per wavefront (row-split accumulators fed straight from HBM) serialises the
//     memory latency: 116 us.
//   * 32 row-split scalar accumulators per lane selected by compare chains: 64 x 32 selects per wavefront, 143 us
//     (uniform branches around them cost more than the selects).
// So tables of <= kSweeps * 4 * (64 / D) rows (32 at D = 16) are summed WITHOUT atomics in two phases per block:
//   A. the usual layout (one lookup per LPR lanes, 4 per lane in flight: full memory parallelism) computes the
//      gradient rows and parks them in LDS with plain stores, [lookup][D] + the row id;
//   B. every lane owns ONE (row, float) pair per sweep -- lane (rg, d) of wavefront w owns float d of row
//      sweep * 4 RG + w * RG + rg (RG = 64 / D) -- and ALL wavefronts walk ALL parked lookups (LDS reads) adding the
//      float under `row == mine`: O(1) work per lookup, no shuffles, no atomics, no cross-wave hand-off.
// At the end every lane holds the block's complete sum of its (row, float): 64 lanes = RG whole rows, added to the table
// gradient with one coalesced atomic instruction (one request per 64-byte line per block and sweep).
// PRIO: the in-step launch (B <= 8192) is part of the step's latency chain and raises its wave priority (RH_CHAIN_PRIO);
// large batches are bandwidth kernels and keep the plain code -- the s_setprio at the top moves the register allocation
// from 123 to 129 VGPRs (4 -> 3 wavefronts per SIMD), which cost the B = 65536 launch 14 %.
template <int LPR, typename IdxT, int SRC, int SINK, bool PRIO>
__global__ __launch_bounds__(RH_BLOCK) void embed_bwd_kernel(const EmbedBwdArgs a) {
  if (PRIO) RH_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];  // small tables: parked gradient rows + row ids
  __shared__ float4 red[RH_BLOCK / RH_WAVE][32];
  constexpr int LPP = RH_BLOCK / LPR;  // lookups per pass
  constexpr int U = 4;
  // Block -> (sample chunk, field).  The blocks that read the SAME rows of g_out / emb / s_sum (one chunk, all fields:
  // adjacent 64-byte pieces of the same 1.7 KB rows) are made neighbours on ONE XCD (block id % 8 = XCD, measured
  // placement): they run at the same time and share the XCD's L2 lines, instead of each fetching a piece of every
  // line on its own.  a.path == 4 keeps the plain chunk-fastest order for comparison.
  int f, chunk;
  {
    const int L = blockIdx.x, F_ = a.F;
    if (a.path == 4) {
      chunk = L % a.nchunks8;
      f = L / a.nchunks8;
    } else {
      const int xcd = L & 7, within = L >> 3;
      f = within % F_;
      chunk = (within / F_) * 8 + xcd;
    }
  }
  if ((int64_t)chunk * a.spb >= (int64_t)a.B) return;  // padding block (chunk count rounded up to the 8 XCDs)
  const int tid = threadIdx.x;
  const int q = tid % LPR;
  const int slot = tid / LPR;
  const int F = a.F, D = a.D;
  const IdxT* ip = reinterpret_cast<const IdxT*>(a.idesc[f]);
  const int64_t st = a.idesc[F + f];
  float* gtab = reinterpret_cast<float*>(a.fdesc[F + f]);
  const int64_t vocab = a.fdesc[2 * F + f];
  const int64_t pad = a.fdesc[3 * F + f];
  const int col = (int)a.idesc[2 * F + f] * D;  // column of this field in g_out / emb
  const bool has_tab = gtab != nullptr;  // frozen tables (requires_grad = False) carry no gradient buffer
  constexpr bool kSmallOk = (SINK != 1) && (LPR <= 16);
  constexpr int RG = kSmallOk ? RH_WAVE / (4 * LPR) : 1;  // row groups of phase B
  constexpr int NL = LPP * U;                            // lookups per pass
  constexpr int RPS = (RH_BLOCK / RH_WAVE) * RG;         // rows per phase-B sweep
  const bool small = kSmallOk && has_tab && a.path != 1 && vocab <= (int64_t)kSweeps * RPS;  // block-uniform
  float* park = lds;                                                 // [NL][D] gradient rows of the pass
  int* park_row = reinterpret_cast<int*>(lds + NL * 4 * LPR);       // [NL] row id, -1 = dead lookup
  float acc[kSweeps];
#pragma unroll
  for (int k = 0; k < kSweeps; ++k) acc[k] = 0.f;
  const bool has_gout = (SRC == 0) && a.g_out != nullptr;
  const bool has_lr = (SRC == 0) && a.g_lr != nullptr && a.lr_w != nullptr;
  const bool has_fm = (SRC == 0) && a.g_fm != nullptr;
  const bool want_wgrad = (SRC == 0) && a.lr_wgrad != nullptr && a.g_lr != nullptr;
  const float4 w4 =
      has_lr ? gload<float4>(a.lr_w + f * D + q * 4) : f4_zero();
  float4 wacc = f4_zero();
  bool oob_any = false;
  const int64_t b0 = (int64_t)chunk * a.spb;
  const int64_t b1 = (b0 + a.spb < (int64_t)a.B) ? b0 + a.spb : (int64_t)a.B;

  // trip count is uniform over the block: the full-line atomic path below shuffles across the wavefront
  for (int64_t base = b0; base < b1; base += (int64_t)LPP * U) {
    const int64_t bb = base + slot;
    bool ok[U];
    int64_t bc[U], row[U];
    float4 g[U], v[U], s4[U];
    float gl[U], gf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = bb + (int64_t)u * LPP;
      ok[u] = b < b1;
      bc[u] = ok[u] ? b : b1 - 1;
      row[u] = (int64_t)gload<IdxT>(ip + bc[u] * st);
      if (SRC == 1) {
        g[u] = gload<float4>(a.rows_in + (bc[u] * F + f) * D + q * 4);
      } else {
        g[u] = has_gout ? gload<float4>(a.g_out + bc[u] * a.g_stride + col + q * 4)
                        : f4_zero();
        const bool need_v = has_fm || want_wgrad;
        v[u] = need_v ? gload<float4>(a.emb + bc[u] * a.emb_stride + col + q * 4)
                      : f4_zero();
        s4[u] = has_fm ? gload<float4>(a.s_sum + bc[u] * D + q * 4) : f4_zero();
        gl[u] = (has_lr || want_wgrad) ? a.g_lr[bc[u]] : 0.f;
        gf[u] = has_fm ? a.g_fm[bc[u]] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 gr = g[u];
      if (SRC == 0) {
        gr = f4_fma(gl[u], w4, gr);                   // + g_lr * w   (w4 = 0 when LR absent)
        gr = f4_fma(gf[u], f4_sub(s4[u], v[u]), gr);  // + g_fm * (S - v)
        if (want_wgrad && ok[u]) wacc = f4_fma(gl[u], v[u], wacc);
      }
      gr = f4_scale(gr, a.scale);
      if (SINK == 1) {
        if (ok[u]) gstore<float4>(a.rows_out + (bc[u] * F + f) * D + q * 4, gr);
      } else {
        const bool oob = (uint64_t)row[u] >= (uint64_t)vocab;
        oob_any |= (oob && ok[u]);
        const bool live = ok[u] && has_tab && !oob && row[u] != pad && a.path != 3;
        if (kSmallOk && small) {
          const int j = u * LPP + slot;
          *reinterpret_cast<float4*>(park + j * (4 * LPR) + q * 4) = gr;
          if (q == 0) park_row[j] = live ? (int)row[u] : -1;
        } else if (a.path == 6) {
          // TIMING ONLY (tools/bwd_ceiling_probe.py; wrong sums when two lookups meet on a row): the cheapest scatter there
          // is -- one plain 16-byte store per lane, a whole 64-byte row per LPR lanes, no re-layout, no read-modify-write.
          // The ceiling of ANY scheme that writes each lookup's row once (sorted / segmented reductions included: with
          // uniform indices over 10 M-row tables a batch holds next to no duplicate to merge).
          if (live) gstore<float4>(gtab + row[u] * D + q * 4, gr);
        } else if (a.wide_atomics) {
          // Re-lay the wavefront's 256 gradient floats (64/LPR rows x 4*LPR dwords) so that one atomic
          // instruction carries WHOLE rows: 4 requests of 64 contiguous dwords instead of 4 requests that each
          // touch a quarter of every row.  Device-scope float atomics are memory-side RMWs on MI355X: their cost
          // is per request per line, so a full 64-byte row per request is 4x fewer line operations.
          constexpr int DD = 4 * LPR;
          const int lane = tid % RH_WAVE;
          const int row_lo = (int)(row[u] & 0xffffffff), row_hi = (int)(row[u] >> 32);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = j * RH_WAVE + lane;
            const int rsl = e / DD;
            const int dw = e % DD;
            const int src = rsl * LPR + dw / 4;
            const float vx = __shfl(gr.x, src, RH_WAVE), vy = __shfl(gr.y, src, RH_WAVE);
            const float vz = __shfl(gr.z, src, RH_WAVE), vw = __shfl(gr.w, src, RH_WAVE);
            const int c = dw & 3;
            const float val = c == 0 ? vx : (c == 1 ? vy : (c == 2 ? vz : vw));
            const int head = rsl * LPR;
            const int64_t r = ((int64_t)__shfl(row_hi, head, RH_WAVE) << 32) | (uint32_t)__shfl(row_lo, head, RH_WAVE);
            const int flag = __shfl((int)live, head, RH_WAVE);
            if (flag) {
              if (a.path == 5) gstore<float>(gtab + r * DD + dw, val);  // TIMING ONLY: the same requests as plain stores
              else gatomic_add_f32(gtab + r * DD + dw, val);
            }
          }
        } else if (live) {
          gatomic_add_f4(gtab + row[u] * D + q * 4, gr);
        }
      }
    }
    if (kSmallOk && small) {  // phase B (block-uniform branch)
      constexpr int DD = 4 * LPR;
      __syncthreads();
      const int lane = tid % RH_WAVE, wave = tid / RH_WAVE;
      const int d = lane % DD;
      const int myrow = wave * RG + lane / DD;
#pragma unroll 16
      for (int j = 0; j < NL; ++j) {
        const int rb = park_row[j];  // wavefront-uniform (broadcast read)
        const float val = park[j * DD + d];
#pragma unroll
        for (int k = 0; k < kSweeps; ++k) acc[k] += (rb == myrow + k * RPS) ? val : 0.f;
      }
      __syncthreads();  // the next pass overwrites the parked rows
    }
  }
  if (kSmallOk && small) {
    constexpr int DD = 4 * LPR;
    const int lane = tid % RH_WAVE, wave = tid / RH_WAVE;
    const int d = lane % DD;
#pragma unroll
    for (int k = 0; k < kSweeps; ++k) {
      const int r = k * RPS + wave * RG + lane / DD;
      if (r < (int)vocab && acc[k] != 0.f) gatomic_add_f32(gtab + (int64_t)r * DD + d, acc[k]);
    }
  }

  if (want_wgrad) {
    // sum over the lookups of this block that share q: inside the wavefront, then across the 4
#pragma unroll
    for (int m = LPR; m < RH_WAVE; m <<= 1) wacc = f4_add(wacc, f4_shfl_xor(wacc, m));
    const int lane = tid % RH_WAVE, wave = tid / RH_WAVE;
    if (lane < LPR) red[wave][lane] = wacc;
    __syncthreads();
    if (tid < LPR) {
      float4 sum = red[0][tid];
#pragma unroll
      for (int wv = 1; wv < RH_BLOCK / RH_WAVE; ++wv) sum = f4_add(sum, red[wv][tid]);
      gstore<float4>(a.lr_wgrad + ((int64_t)chunk * F + f) * D + tid * 4, sum);
    }
  }
  if (oob_any && a.err != nullptr) atomicOr(a.err, RH_FLAG_INDEX_OOB);
}

int g_bwd_path = 0;        // tuning knob RH_TUNE_BWD_PATH (experiments)

int g_wide_atomics = 1;  // tuning knob RH_TUNE_WIDE_ATOMICS (rh_set_tuning / RECHUB_TUNE=1=0), default on

int wide_atomics_default() { return g_wide_atomics; }

template <int LPR, typename IdxT, int SRC, int SINK>
int launch_bwd(EmbedBwdArgs a, hipStream_t s) {
  const unsigned gx = (unsigned)((a.B + a.spb - 1) / a.spb);
  a.wide_atomics = wide_atomics_default();
  a.path = g_bwd_path;
  // parked gradient rows of one pass (16 KiB) + their row ids
  const size_t shmem = (SINK != 1) ? (size_t)(RH_BLOCK * 4 * 4 + (RH_BLOCK / LPR) * 4) * sizeof(float) : 0;
  a.nchunks8 = (int)((gx + 7) / 8 * 8);
  const dim3 grid((unsigned)a.nchunks8 * (unsigned)a.F);
  if (a.B <= 8192)
    hipLaunchKernelGGL((embed_bwd_kernel<LPR, IdxT, SRC, SINK, true>), grid, dim3(RH_BLOCK), shmem, s, a);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<LPR, IdxT, SRC, SINK, false>), grid, dim3(RH_BLOCK), shmem, s, a);
  return 0;
}

template <typename IdxT, int SRC, int SINK>
int dispatch_bwd(const EmbedBwdArgs& a, hipStream_t s) {
  switch (a.D / 4) {
    case 1: return launch_bwd<1, IdxT, SRC, SINK>(a, s);
    case 2: return launch_bwd<2, IdxT, SRC, SINK>(a, s);
    case 4: return launch_bwd<4, IdxT, SRC, SINK>(a, s);
    case 8: return launch_bwd<8, IdxT, SRC, SINK>(a, s);
    case 16: return launch_bwd<16, IdxT, SRC, SINK>(a, s);
    case 32: return launch_bwd<32, IdxT, SRC, SINK>(a, s);
    default: return RH_E_UNSUPPORTED;
  }
}

int check_common(const char* who, const int64_t* fdesc, const int64_t* idesc, int B, int F, int D) {
  RH_REQUIRE(fdesc != nullptr && idesc != nullptr, RH_E_BADARG, "%s: null descriptor", who);
  RH_REQUIRE(B >= 0 && F > 0, RH_E_BADARG, "%s: bad shape B=%d F=%d", who, B, F);
  RH_REQUIRE(D > 0 && D % 4 == 0 && D <= 128 && ((D / 4) & (D / 4 - 1)) == 0, RH_E_UNSUPPORTED,
             "%s: embed_dim %d unsupported by the fused kernel (need 4,8,16,32,64,128)", who, D);
  RH_REQUIRE(F <= 65535, RH_E_UNSUPPORTED, "%s: too many fields (%d)", who, F);
  return 0;
}

int pick_spb(int spb) {
  if (spb <= 0) return 256;
  return ((spb + 63) / 64) * 64;
}

}  // namespace

extern "C" int rh_embed_fwd(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B, int F,
                            int D, const int64_t* ddesc, int n_dense, int dense_col, float* out,
                            int64_t out_stride,
                            const float* lr_w, const float* lr_b, float* lr_out, float* fm_out,
                            float* s_out, int field_split, int32_t* err_flag, void* stream) {
  if (int rc = check_common("rh_embed_fwd", fdesc, idesc, B, F, D)) return rc;
  RH_REQUIRE(out != nullptr, RH_E_BADARG, "rh_embed_fwd: out is null");
  RH_REQUIRE(dense_col >= 0 && out_stride >= (int64_t)dense_col + n_dense, RH_E_BADARG,
             "rh_embed_fwd: out_stride %lld too small for dense block at column %d (+%d)", (long long)out_stride,
             dense_col, n_dense);
  RH_REQUIRE(n_dense == 0 || ddesc != nullptr, RH_E_BADARG, "rh_embed_fwd: n_dense > 0 but ddesc is null");
  RH_REQUIRE(lr_out == nullptr || lr_w != nullptr, RH_E_BADARG, "rh_embed_fwd: lr_out without lr_w");
  RH_REQUIRE(lr_out == nullptr || (int64_t)F * D <= 12288, RH_E_UNSUPPORTED,
             "rh_embed_fwd: fused LR keeps the F*D = %lld weights in LDS (<= 12288 floats)", (long long)F * D);
  if (B == 0) return 0;
  EmbedFwdArgs a{fdesc, idesc, ddesc, B, F, D, n_dense, dense_col, out, out_stride, lr_w, lr_b, lr_out, fm_out, s_out, err_flag};
  if (lr_out == nullptr) a.lr_w = nullptr;
  int fs = field_split;
  const int lpr = D / 4;
  if (fs <= 0) {
    // smallest split for which a lane's fields fit ONE phase of 8 gathers (ceil(F/fs) <= 8): all of its row
    // loads are in flight together and the sample's stores stay 64*fs bytes wide.  Measured best at every batch
    // size for F = 26, D = 16 (fs = 4: 8.9 / 15.4 / 51.7 us at B = 4096 / 16384 / 65536).
    fs = 1;
    while (fs < 8 && lpr * fs * 2 <= RH_WAVE && (F + fs - 1) / fs > 8) fs *= 2;
  }
  while (fs > 1 && lpr * fs > RH_WAVE) fs /= 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool uniform = field_split <= 0 && (g_fwd_path == 2 || (g_fwd_path == 0 && (int64_t)B * F >= kUniformMinLookups));
  int rc;
  if (uniform) rc = idx_is_i64 ? dispatch_uniform<kUniformWaves, int64_t>(a, s) : dispatch_uniform<kUniformWaves, int32_t>(a, s);
  else rc = idx_is_i64 ? dispatch_lpr<int64_t>(a, fs, s) : dispatch_lpr<int32_t>(a, fs, s);
  if (rc != 0) return rc;
  RH_LAUNCH_CHECK("rh_embed_fwd");
  return 0;
}

extern "C" int rh_embed_bwd(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B, int F,
                            int D, const float* g_out, int64_t g_stride, const float* emb,
                            int64_t emb_stride, const float* s_sum, const float* g_fm,
                            const float* g_lr, const float* lr_w, float* lr_wgrad, float scale, int sink,
                            float* rows_out, int samples_per_block, int32_t* err_flag, void* stream) {
  if (int rc = check_common("rh_embed_bwd", fdesc, idesc, B, F, D)) return rc;
  RH_REQUIRE(sink == 0 || sink == 1, RH_E_BADARG, "rh_embed_bwd: sink must be 0 or 1");
  RH_REQUIRE(sink == 0 || rows_out != nullptr, RH_E_BADARG, "rh_embed_bwd: sink=1 needs rows_out");
  RH_REQUIRE(g_fm == nullptr || (emb != nullptr && s_sum != nullptr), RH_E_BADARG,
             "rh_embed_bwd: g_fm needs emb and s_sum");
  RH_REQUIRE(lr_wgrad == nullptr || (emb != nullptr && g_lr != nullptr), RH_E_BADARG,
             "rh_embed_bwd: lr_wgrad needs emb and g_lr");
  if (B == 0) return 0;
  EmbedBwdArgs a{fdesc, idesc, B, F, D, g_out, g_stride, emb, emb_stride, s_sum, g_fm, g_lr, lr_w,
                 lr_wgrad, scale, rows_out, nullptr, pick_spb(samples_per_block), 0, 0, 0, err_flag};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (sink == 0)
    rc = idx_is_i64 ? dispatch_bwd<int64_t, 0, 0>(a, s) : dispatch_bwd<int32_t, 0, 0>(a, s);
  else
    rc = idx_is_i64 ? dispatch_bwd<int64_t, 0, 1>(a, s) : dispatch_bwd<int32_t, 0, 1>(a, s);
  if (rc != 0) return rc;
  RH_LAUNCH_CHECK("rh_embed_bwd");
  return 0;
}

extern "C" int rh_set_tuning(int key, int value) {
  if (key == RH_TUNE_WIDE_ATOMICS) {
    g_wide_atomics = value != 0;
    return 0;
  }
  if (key == RH_TUNE_BWD_PATH) {
    g_bwd_path = value;
    return 0;
  }
  if (key == RH_TUNE_FWD_PATH) {
    g_fwd_path = value;
    return 0;
  }
  if (key == RH_TUNE_BWD_SPLIT || key == RH_TUNE_BWD_SLABS) return 0;  // retired knobs (kept so old probes still run)
  if (rh_optim_set_tuning(key, value) == 0) return 0;
  if (rh_linear_set_tuning(key, value) == 0) return 0;
  if (rh_din_set_tuning(key, value) == 0) return 0;
  rh_set_error("rh_set_tuning: unknown key %d", key);
  return RH_E_BADARG;
}

extern "C" int rh_embed_bwd_nchunks(int B, int samples_per_block) {
  const int spb = pick_spb(samples_per_block);
  return (B + spb - 1) / spb;
}

extern "C" int rh_embed_scatter_rows(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B,
                                     int F, int D, const float* rows, float scale,
                                     int samples_per_block, int32_t* err_flag, void* stream) {
  if (int rc = check_common("rh_embed_scatter_rows", fdesc, idesc, B, F, D)) return rc;
  RH_REQUIRE(rows != nullptr, RH_E_BADARG, "rh_embed_scatter_rows: rows is null");
  if (B == 0) return 0;
  EmbedBwdArgs a{fdesc, idesc, B, F, D, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                 nullptr, scale, nullptr, rows, pick_spb(samples_per_block), 0, 0, 0, err_flag};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = idx_is_i64 ? dispatch_bwd<int64_t, 1, 0>(a, s) : dispatch_bwd<int32_t, 1, 0>(a, s);
  if (rc != 0) return rc;
  RH_LAUNCH_CHECK("rh_embed_scatter_rows");
  return 0;
}
