// Dense Adam over every embedding table in one launch (torch.optim.Adam semantics, coupled L2).
//
// Reference: CTRTrainer.__init__ / train_one_epoch, torch_rechub/trainers/ctr_trainer.py:59-61,99
//   optimizer_fn(model.parameters(), lr=1e-3, weight_decay=1e-5) -> optimizer.step() walks EVERY
//   row of EVERY table each step: rows absent from the batch still decay their moments
//   and receive the coupled weight-decay gradient wd*p.
//
// Roofline: HBM streaming.  Algorithmic bytes per element: read p, g, m, v + write p, m, v = 28 B
// (7 streams x 4 B; 15.1 GB per step at 33.76 M rows x 16).  The gradient buffer is re-zeroed in
// the same pass (stores only where it was non-zero), so no separate zero_grad / memset pass exists.
#include "common.h"
#include "wgrad_body.h"

namespace {

constexpr int kMaxTensors = 128;
constexpr int kMaxRing = 1024;  // entries of the (A, E) ring kept in LDS by the lazy sweep
int g_sweep_grid = 0;  // tuning knob RH_TUNE_SWEEP_GRID (0 = default 8192 workgroups, the measured best)
// A DEFERRED sweep (rh_adam_lazy_sweep with t_value >= 0) runs on a side stream BESIDE the step's launch chain.  Left at
// its in-line shape (8192 persistent workgroups, every wave slot it can get) it starves the chain: measured on the DeepFM
// step, the chain's kernels ran 2.6x slower under it and the overlap gained nothing.  Its residency is therefore capped:
//   RH_TUNE_DEFERRED_GRID  persistent workgroups of a deferred sweep (default 512 = 2 per CU = 2 wavefronts per SIMD; the
//                          trainer's self-tuning also tries 256 = 1 per CU, which wins under long chains: DCN-v2, B >= 8192)
// Step time by cap (DeepFM, B = 4096, same box, round 3): grid 256 / 384 / 448 / 512 / 576 / 640 / 1024:
// 0.360 / 0.355 / 0.318 / 0.302 / 0.334 / 0.326 / 0.345 ms.  (An LDS-padding cap reached 0.311 ms and was removed: it also
// kept LDS-hungry kernels of the chain off the CU.)
int g_deferred_grid = 512;
int g_gate_ns = 32000;  // RH_TUNE_SWEEP_GATE_NS: rh_adam_sweep_gate, hold-back behind the opening for graphs that count no chain start (round 4: every graph)
int g_stagger_ns = 15000;  // RH_TUNE_SWEEP_STAGGER_NS (untraced landscape, tools/period_hist.py: 12-18 us clean, 9 us 21 % slow steps)

constexpr int kVecPerThread = 4;                                // float4 per thread per stream
constexpr int kChunk4 = RH_BLOCK * kVecPerThread;               // float4 per virtual block

struct AdamArgs {
  const int64_t* tdesc;  // [5*T] p, g, m, v, numel
  const double* hyper;
  int T;
  int zero_grad;
  int64_t total_vblocks;
  int64_t vb_prefix[kMaxTensors + 1];  // virtual-block prefix sum per tensor
};

__global__ void adam_prepare_kernel(double* hyper, int64_t* step, float* ring, int64_t ring_mask) {
  RH_CHAIN_PRIO();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t t = *step + 1;
    *step = t;
    const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2];
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    hyper[8] = lr / bc1;      // step_size
    hyper[9] = sqrt(bc2);     // bias_correction2_sqrt
    hyper[10] = 1.0 - b1;     // lerp weight
    hyper[11] = 1.0 - b2;
    hyper[12] = (double)t;
    // folded form used by every kernel here:  p -= A * m / (sqrt(v) + E)
    //   = step_size * m / (sqrt(v)/sqrt(bc2) + eps)   with A = step_size*sqrt(bc2), E = eps*sqrt(bc2)
    const double A = hyper[8] * hyper[9], E = hyper[3] * hyper[9];
    hyper[13] = A;
    hyper[14] = E;
    if (ring != nullptr) {
      ring[2 * (t & ring_mask) + 0] = (float)A;
      ring[2 * (t & ring_mask) + 1] = (float)E;
    }
  }
}

struct AdamScalars {
  float b1, b2, wd, one_m_b1, one_m_b2;
  float c1wd, c2wd2;  // (1-b1)*wd, (1-b2)*wd^2: the zero-gradient form below
  float A, E;  // per-step: A = lr/(1-b1^t)*sqrt(1-b2^t), E = eps*sqrt(1-b2^t)
};

// One element of torch.optim.Adam (_single_tensor_adam, amsgrad=False, maximize=False):
//   g += wd*p; m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
//   denom = v.sqrt()/bc2_sqrt + eps; p.addcdiv_(m, denom, -step_size)      [ == p -= A*m/(sqrt(v)+E) ]
// The dense and the lazy kernels share THIS function, so their results are bit-identical.
static __device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& h, float A,
                                                 float E) {
  g = fmaf(h.wd, p, g);
  m = fmaf(h.one_m_b1, g - m, m);
  v = fmaf(h.one_m_b2, g * g, v * h.b2);
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE-correct expansions (~12 + ~11 instructions): the lazy
  // replay is ALU-bound on exactly this function.  E > 0 keeps the denominator >= eps*sqrt(1-b2^t) (no 1/0), and a
  // denormal v is far below E^2, so flushing it changes nothing.  Relative error of the update ~2e-7.
  p = fmaf(-A, m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + E), p);
}
// Two elements at a time on 64-bit register pairs: everything but sqrt / rcp is a v_pk_* instruction.  Same operations,
// same order, same rounding as adam_elem, element for element.  Measured on MI355X (rocprofv3 SQ counters on the sweep,
// and variants with sqrt / rcp removed): v_pk_*_f32 issue at 4 cycles per wavefront, v_sqrt_f32 / v_rcp_f32 at 8; the
// replay loop (zero-gradient form below: 16 packed + 8 transcendental + 2 plain ops per 256 element-steps = 136 cycles)
// keeps the VALU ~80 % busy -- the sweep is bound by f32 VALU throughput, half of it the two transcendentals.
typedef float rh_v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void adam_pair(rh_v2f& p, rh_v2f g, rh_v2f& m, rh_v2f& v, const AdamScalars& h,
                                                 float A, float E) {
  g = __builtin_elementwise_fma(rh_v2f{h.wd, h.wd}, p, g);
  m = __builtin_elementwise_fma(rh_v2f{h.one_m_b1, h.one_m_b1}, g - m, m);
  v = __builtin_elementwise_fma(rh_v2f{h.one_m_b2, h.one_m_b2}, g * g, v * h.b2);
  rh_v2f d = rh_v2f{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)} + E;
  d = rh_v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  p = __builtin_elementwise_fma(rh_v2f{-A, -A}, m * d, p);
}
// An element whose data gradient is EXACTLY zero (every replayed step of the lazy kernels; untouched rows in the dense
// pass) has g = wd*p, and the same update needs one operation less when g is never formed:
//   m = b1*m + (1-b1)*wd*p,  v = b2*v + (1-b2)*wd^2*p^2          (8 instead of 9 packed ops + sqrt + rcp)
// Same real-number result as adam_pair with g = 0, different rounding -- so EVERY kernel applies this form to every
// element with a zero gradient and adam_pair to the others (adam_f4 selects per element): lazy == dense stays
// bit-identical, and the replay loop, which the step time hangs on, runs the short form unconditionally.
static __device__ __forceinline__ void adam_pair_zero_g(rh_v2f& p, rh_v2f& m, rh_v2f& v, const AdamScalars& h, float A,
                                                        float E) {
  m = __builtin_elementwise_fma(rh_v2f{h.c1wd, h.c1wd}, p, m * h.b1);
  v = __builtin_elementwise_fma(rh_v2f{h.c2wd2, h.c2wd2}, p * p, v * h.b2);
  rh_v2f d = rh_v2f{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)} + E;
  d = rh_v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  p = __builtin_elementwise_fma(rh_v2f{-A, -A}, m * d, p);
}
static __device__ __forceinline__ void adam_f4_zero_g(float4& P, float4& M, float4& V, const AdamScalars& h, float A,
                                                      float E) {
  rh_v2f p0{P.x, P.y}, p1{P.z, P.w}, m0{M.x, M.y}, m1{M.z, M.w}, v0{V.x, V.y}, v1{V.z, V.w};
  adam_pair_zero_g(p0, m0, v0, h, A, E);
  adam_pair_zero_g(p1, m1, v1, h, A, E);
  P = make_float4(p0.x, p0.y, p1.x, p1.y);
  M = make_float4(m0.x, m0.y, m1.x, m1.y);
  V = make_float4(v0.x, v0.y, v1.x, v1.y);
}
// one step with a gradient that may be zero element-wise
static __device__ __forceinline__ void adam_f4(float4& P, const float4 G, float4& M, float4& V, const AdamScalars& h,
                                               float A, float E) {
  float4 Pz = P, Mz = M, Vz = V;
  adam_f4_zero_g(Pz, Mz, Vz, h, A, E);
  rh_v2f p0{P.x, P.y}, p1{P.z, P.w}, m0{M.x, M.y}, m1{M.z, M.w}, v0{V.x, V.y}, v1{V.z, V.w};
  adam_pair(p0, rh_v2f{G.x, G.y}, m0, v0, h, A, E);
  adam_pair(p1, rh_v2f{G.z, G.w}, m1, v1, h, A, E);
  P = make_float4(G.x == 0.f ? Pz.x : p0.x, G.y == 0.f ? Pz.y : p0.y, G.z == 0.f ? Pz.z : p1.x, G.w == 0.f ? Pz.w : p1.y);
  M = make_float4(G.x == 0.f ? Mz.x : m0.x, G.y == 0.f ? Mz.y : m0.y, G.z == 0.f ? Mz.z : m1.x, G.w == 0.f ? Mz.w : m1.y);
  V = make_float4(G.x == 0.f ? Vz.x : v0.x, G.y == 0.f ? Vz.y : v0.y, G.z == 0.f ? Vz.z : v1.x, G.w == 0.f ? Vz.w : v1.y);
}
static __device__ __forceinline__ AdamScalars load_scalars(const double* hyper) {
  AdamScalars h;
  h.b1 = (float)hyper[1];
  h.b2 = (float)hyper[2];
  h.wd = (float)hyper[4];
  h.one_m_b1 = (float)hyper[10];
  h.one_m_b2 = (float)hyper[11];
  h.c1wd = h.one_m_b1 * h.wd;
  h.c2wd2 = h.one_m_b2 * h.wd * h.wd;
  h.A = (float)hyper[13];
  h.E = (float)hyper[14];
  return h;
}

__global__ __launch_bounds__(RH_BLOCK) void adam_dense_kernel(const AdamArgs a) {
  const AdamScalars h = load_scalars(a.hyper);
  const int T = a.T;
  for (int64_t vb = blockIdx.x; vb < a.total_vblocks; vb += gridDim.x) {
    // binary search: last t with vb_prefix[t] <= vb  (wave-uniform, scalar loads from kernarg)
    int lo = 0, hi = T;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.vb_prefix[mid] <= vb) lo = mid; else hi = mid;
    }
    const int t = lo;
    float* p = reinterpret_cast<float*>(a.tdesc[0 * T + t]);
    float* g = reinterpret_cast<float*>(a.tdesc[1 * T + t]);
    float* m = reinterpret_cast<float*>(a.tdesc[2 * T + t]);
    float* v = reinterpret_cast<float*>(a.tdesc[3 * T + t]);
    const int64_t n4 = a.tdesc[4 * T + t] / 4;
    const int64_t base4 = (vb - a.vb_prefix[t]) * kChunk4;
    float4 P[kVecPerThread], Gv[kVecPerThread], M[kVecPerThread], V[kVecPerThread];
    bool ok[kVecPerThread];
#pragma unroll
    for (int k = 0; k < kVecPerThread; ++k) {
      const int64_t i4 = base4 + (int64_t)k * RH_BLOCK + threadIdx.x;
      ok[k] = i4 < n4;
      const int64_t j4 = ok[k] ? i4 : 0;
      P[k] = gload<float4>(p + j4 * 4);
      Gv[k] = gload<float4>(g + j4 * 4);
      M[k] = gload<float4>(m + j4 * 4);
      V[k] = gload<float4>(v + j4 * 4);
    }
#pragma unroll
    for (int k = 0; k < kVecPerThread; ++k) {
      if (!ok[k]) continue;
      const int64_t i4 = base4 + (int64_t)k * RH_BLOCK + threadIdx.x;
      const bool gnz = Gv[k].x != 0.f || Gv[k].y != 0.f || Gv[k].z != 0.f || Gv[k].w != 0.f;
      adam_f4(P[k], Gv[k], M[k], V[k], h, h.A, h.E);
      gstore<float4>(p + i4 * 4, P[k]);
      gstore<float4>(m + i4 * 4, M[k]);
      gstore<float4>(v + i4 * 4, V[k]);
      if (a.zero_grad && gnz) gstore<float4>(g + i4 * 4, f4_zero());
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Blocked-lazy EXACT Adam.  Adam is element-wise: a row's (p, m, v) depend only on that row's own gradient history,
// and a row that is not in the batch has the KNOWN gradient wd*p.  So a row can be brought up to date later by
// replaying the skipped steps in registers ("replay": g = wd*p with that step's bias corrections from a ring buffer)
// with NO memory traffic — the result is bit-identical to stepping it densely every step (same adam_elem).
//   * touched pass: every row the batch looked up is claimed once (atomicMax on its last-step word), replayed up to
//     step t-1 and then stepped with its data gradient; its gradient row is re-zeroed.
//   * sweep pass: each step a 1/K window of every table is brought up to date, so no row ever lags more than K steps
//     (bounds the replay length and the ring); tables with <= small_rows rows use K = 1 (dense, no atomics).
//   * flush = sweep over all rows (before the weights are read by anything else: eval, state_dict, checkpoint).
// HBM traffic per step falls from 28 B x all elements to ~1/K of that; the arithmetic (one adam_elem per element per
// step) is unchanged and becomes the bound.
// ldesc (device int64 [8*T]): p, g, m, v, last(int32*) pointers, rows, K_t, window rows w_t
struct LazyTouchedArgs {
  const int64_t* ldesc;        // [8*T] as above
  const int64_t* field_table;  // [2*F]: table index of the field (-1: skip), padding_idx (-1: none)
  const int64_t* idesc;        // [>=2*F] index column pointer + stride per field
  const double* hyper;
  const float* ring;
  int ring_mask;
  int T, B, F, spb;
  int* err;
  // ASSEMBLE (round 4): the batch does not exist yet -- this launch also assembles it (rh_batch_gather's work) and reads its
  // own indices straight from the resident dataset: sample b of the batch = dataset row perm[(pos + b) mod N]
  const int64_t* perm;
  const int64_t* pos;
  int64_t N;
  const int64_t* src_sparse;  // (N, Fd)
  int Fd;
  const float* src_dense;     // (N, ND) or null
  int ND;
  const float* src_label;     // (N,) or null
  int64_t* sparse_out;        // (B, Fd): the batch buffer the fields' index columns point into
  float* dense_out;
  float* label_out;
  // LOOK (round 4): further workgroups pre-refresh those lookups of the batches AFTER this one (samples pos + off + B + b,
  // b < look_n) that fall into the window the coming deferred sweep walks; look = samples per lookahead workgroup (0: none)
  int look;
  int look_n;   // samples the lookahead covers (B x depth)
  int64_t off;  // ASSEMBLE: sample b of this pass is dataset position pos + off + b (0: the batch at pos; -B: the one before)
};

struct LazySweepArgs {
  const int64_t* ldesc;
  const double* hyper;
  const float* ring;
  int ring_mask;
  int T;
  int flush;  // 1: window = whole table
  int64_t t_value;  // >= 0: the step this sweep belongs to, by value (deferred sweep); < 0: hyper[12]
  int64_t total_vblocks;
  int64_t vb_prefix[kMaxTensors + 1];
  // merged launch (rh_adam_lazy_step): the first touch_blocks workgroups run the touched-rows step of the batch, the rest
  // the window sweep.  Both claim a lazy row with atomicMax on its last-step word, and whoever claims it replays it AND
  // applies its gradient row (the sweep then reads the gradient of every window row)
  int touch_blocks, touch_chunks, touch_period;
  int touch_i32;  // merged launch: the batch's index columns are int32 (row-sharded tables: localised indices)
  LazyTouchedArgs touch;
};

template <int LPR, typename IdxT, bool REFRESH, bool ASSEMBLE = false, bool LOOK = false, bool GRAD = !REFRESH>
static __device__ __forceinline__ void lazy_touched_body(const LazyTouchedArgs& a, int bx, int f);

// (bx_, gdim_: this workgroup's index and the number of workgroups of the sweep / merged part -- blockIdx.x / gridDim.x unless
// a launch carries more parts: round 4 measured the dense-gradient packing as such a part and dropped it)
template <int LPR, bool MERGED>
static __device__ __forceinline__ void lazy_sweep_body(const LazySweepArgs& a, const int bx_, const int gdim_) {
  constexpr int RPB = RH_BLOCK / LPR;  // rows per block
  constexpr int D = 4 * LPR;
  // merged launch: every touch_period-th workgroup is a touched-rows workgroup (INTERLEAVED with the sweep's: put first,
  // they fill every CU before a sweep workgroup starts and the two parts run one after the other -- measured)
  int64_t bid = bx_, nblk = gdim_;
  if (MERGED) {
    const int bx = bx_, P = a.touch_period;
    const int slot_ = bx / P;
    if (bx % P == 0 && slot_ < a.touch_blocks) {
      if (a.touch_i32)  // (launch-uniform)
        lazy_touched_body<LPR, int, false>(a.touch, slot_ % a.touch_chunks, slot_ / a.touch_chunks);
      else
        lazy_touched_body<LPR, int64_t, false>(a.touch, slot_ % a.touch_chunks, slot_ / a.touch_chunks);
      return;
    }
    bid = bx - (slot_ + 1 < a.touch_blocks ? slot_ + 1 : a.touch_blocks);
    nblk = (int64_t)gdim_ - a.touch_blocks;
  }
  AdamScalars h = load_scalars(a.hyper);
  const int t = a.t_value >= 0 ? (int)a.t_value : (int)a.hyper[12];
  if (a.t_value >= 0) {  // deferred: hyper[12..14] may already belong to the next step; the ring entry of t does not
    h.A = a.ring[2 * (t & a.ring_mask)];
    h.E = a.ring[2 * (t & a.ring_mask) + 1];
  }
  const int T = a.T;
  const int q = threadIdx.x % LPR;
  const int slot = threadIdx.x / LPR;
  // The per-step (A, E) ring lives in LDS for the replay loop: a global load there would make every s_waitcnt vmcnt
  // also wait for the prefetched rows of the next unit (vmcnt retires in order) and serialise memory behind the ALU.
  __shared__ float ring_s[2 * kMaxRing];
  for (int i = threadIdx.x; i < 2 * (a.ring_mask + 1); i += RH_BLOCK) ring_s[i] = a.ring[i];
  __syncthreads();

  // One unit of work = one table row of one virtual block.  The loads of unit n+1 are issued before unit n is
  // replayed, so the (long, pure-ALU) replay of one row hides the HBM latency of the next.
  struct Unit {
    float *p, *g, *m, *v;
    int* last;
    int64_t r;
    int old;
    bool live, with_g, claimed;
    float4 P, M, V, G;
  };
  auto fetch = [&](int64_t vb, Unit& u) {
    u.live = false;
    if (vb >= a.total_vblocks) return;
    int lo = 0, hi = T;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.vb_prefix[mid] <= vb) lo = mid; else hi = mid;
    }
    const int ti = lo;
    u.p = reinterpret_cast<float*>(a.ldesc[0 * T + ti]);
    u.g = reinterpret_cast<float*>(a.ldesc[1 * T + ti]);
    u.m = reinterpret_cast<float*>(a.ldesc[2 * T + ti]);
    u.v = reinterpret_cast<float*>(a.ldesc[3 * T + ti]);
    u.last = reinterpret_cast<int*>(a.ldesc[4 * T + ti]);
    const int64_t rows = a.ldesc[5 * T + ti];
    const int64_t K = a.ldesc[6 * T + ti];
    const int64_t w = a.flush ? rows : a.ldesc[7 * T + ti];
    const int64_t wstart = a.flush ? 0 : ((int64_t)(t - 1) % K) * w;
    const int64_t local = (vb - a.vb_prefix[ti]) * RPB + slot;
    u.r = wstart + local;
    if (local >= w || u.r >= rows) return;
    u.with_g = (K == 1);  // dense tables receive their gradient here; others had it applied when touched
    const bool claim = MERGED && K != 1;
    u.claimed = claim;
    if (claim) {
      // the touched-rows workgroups of this launch may want the same row: claim it (exactly one claimant sees a value
      // < t) and take over its gradient row.  Unconditional (a window row appears once): ONE round trip, in flight
      // together with the row's loads -- a load-then-atomic pair in front of them cost 10 us per launch
      u.old = t;
      if (q == 0) u.old = atomicMax(u.last + u.r, t);
      u.with_g = true;
    } else {
      u.old = gload<int>(u.last + u.r);
    }
    u.P = gload<float4>(u.p + u.r * D + q * 4);
    u.M = gload<float4>(u.m + u.r * D + q * 4);
    u.V = gload<float4>(u.v + u.r * D + q * 4);
    u.G = u.with_g ? gload<float4>(u.g + u.r * D + q * 4) : f4_zero();
    if (claim) u.old = __shfl(u.old, (int)(threadIdx.x % RH_WAVE) - q, RH_WAVE);
    u.live = true;
  };
  auto process = [&](Unit& u) {
    const bool work = u.live && u.old < t;  // old >= t: already stepped by the touched pass
    // The replay loop runs on a wavefront-uniform counter (scalar ALU, ring entry read once per wavefront) from the
    // oldest row of the wavefront; rows that are more recent join later under the exec mask.
    const int first = work ? u.old + 1 : t;
    // The replay runs in SEGMENTS between the steps at which rows of the wavefront join (rows of one window were last
    // swept together; rows the batch touched since join later): inside a segment the set of replaying lanes is fixed
    // (one exec mask) and the step counter is scalar.  The per-iteration form (`if (j >= first)` inside one loop) cost
    // ~22 of 158 cycles per iteration in v_cmp / s_and_saveexec / s_or exec / counter VALU ops.
    int j = wave_min_uniform(first);
    while (j < t) {  // wavefront-uniform
      const int nxt = wave_min_uniform(first > j ? first : t);  // the next joining step, > j
      if (first <= j) {
        // two steps per iteration, both ring entries read (LDS) before the first step's arithmetic: at the 2 wavefronts per
        // SIMD of the deferred form the read's latency is otherwise exposed once per replayed step (same operations in the
        // same order).  Round 3 measured it on the sweep alone (188 -> 172 us at 512 workgroups) with no gain for the step --
        // chain and sweep were co-critical then; with the round-4 chain the sweep's path is the longer one: 0.275 -> 0.269 ms.
        int jj = j;
        for (; jj + 2 <= nxt; jj += 2) {
          const float2 ae0 = *reinterpret_cast<const float2*>(ring_s + 2 * (jj & a.ring_mask));
          const float2 ae1 = *reinterpret_cast<const float2*>(ring_s + 2 * ((jj + 1) & a.ring_mask));
          __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the second read to its use)
          adam_f4_zero_g(u.P, u.M, u.V, h, ae0.x, ae0.y);
          adam_f4_zero_g(u.P, u.M, u.V, h, ae1.x, ae1.y);
        }
        if (jj < nxt) {
          const float2 ae = *reinterpret_cast<const float2*>(ring_s + 2 * (jj & a.ring_mask));
          adam_f4_zero_g(u.P, u.M, u.V, h, ae.x, ae.y);
        }
      }
      j = nxt;
    }
    if (work) {
      // lazy tables never carry a gradient here (their rows got it in the touched pass): short form, no select
      if (u.with_g) adam_f4(u.P, u.G, u.M, u.V, h, h.A, h.E);
      else adam_f4_zero_g(u.P, u.M, u.V, h, h.A, h.E);
      gstore<float4>(u.p + u.r * D + q * 4, u.P);
      gstore<float4>(u.m + u.r * D + q * 4, u.M);
      gstore<float4>(u.v + u.r * D + q * 4, u.V);
      if (u.with_g && (u.G.x != 0.f || u.G.y != 0.f || u.G.z != 0.f || u.G.w != 0.f))
        gstore<float4>(u.g + u.r * D + q * 4, f4_zero());
      if (q == 0 && !(MERGED && u.with_g && u.claimed)) u.last[u.r] = t;
    }
  };
  // two units ping-pong (no register copies): the loads of one are in flight while the other is replayed
  Unit ua, ub;
  fetch(bid, ua);
  for (int64_t vb = bid; vb < a.total_vblocks; vb += 2 * nblk) {
    fetch(vb + nblk, ub);
    process(ua);
    fetch(vb + 2 * nblk, ua);
    process(ub);
  }
}

template <int LPR, bool MERGED = false>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_sweep_kernel(const LazySweepArgs a) {
  lazy_sweep_body<LPR, MERGED>(a, (int)blockIdx.x, (int)gridDim.x);
}

// The DEFERRED window sweep of the lazy tables with VPL float4 per lane (round 5).  The replay of one element is a dependent
// chain per step -- v' -> v_sqrt -> + E -> v_rcp -> m' * d -> fma into p, and p feeds the next step's m', v' -- about 80 cycles
// long against 136 cycles of issue per step for the two packed pairs a lane holds at one float4 per lane.  Beside the step's
// chain the sweep runs at 2 wavefronts per SIMD (its residency cap): the other wavefront covers only part of that latency and
// the loop reaches 0.52 of its VALU ceiling.  Here a row of D floats is held by LPR = D / (4 VPL) lanes with VPL
// float4 EACH (32 contiguous bytes at VPL = 2): 2 VPL independent pair chains per lane at the same wavefront count -- the
// instruction-level parallelism that more wavefronts would buy, without their wave slots, LDS and registers (more resident
// sweep wavefronts starve the chain's 240-register kernels: 768 / 1024 workgroups were 0.333-0.354 ms steps in round 3).
// Same adam_f4_zero_g per float4, same order of steps: the bits of the one-float4 kernel (and of the dense pass).
// Only the lazy tables' window (RH_SWEEP_LAZY_TABLES by value): no gradient rows, no claims -- which is what keeps the two
// ping-pong units inside the 128 registers that leave a SIMD room for the chain.
template <int LPR, int VPL>
static __device__ __forceinline__ void lazy_sweep_wide_body(const LazySweepArgs& a, const int bx_, const int gdim_) {
  constexpr int RPB = RH_BLOCK / LPR;
  constexpr int D = 4 * LPR * VPL;
  const int64_t bid = bx_, nblk = gdim_;
  AdamScalars h = load_scalars(a.hyper);
  const int t = (int)a.t_value;
  h.A = a.ring[2 * (t & a.ring_mask)];
  h.E = a.ring[2 * (t & a.ring_mask) + 1];
  const int T = a.T;
  const int q = threadIdx.x % LPR;
  const int slot = threadIdx.x / LPR;
  __shared__ float ring_s[2 * kMaxRing];
  for (int i = threadIdx.x; i < 2 * (a.ring_mask + 1); i += RH_BLOCK) ring_s[i] = a.ring[i];
  __syncthreads();
  struct Unit {
    float *p, *m, *v;
    int* last;
    int64_t off;  // element offset of this lane's first float4
    int64_t r;
    int old;
    bool live;
    float4 P[VPL], M[VPL], V[VPL];
  };
  auto fetch = [&](int64_t vb, Unit& u) {
    u.live = false;
    if (vb >= a.total_vblocks) return;
    int lo = 0, hi = T;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.vb_prefix[mid] <= vb) lo = mid; else hi = mid;
    }
    const int ti = lo;
    u.p = reinterpret_cast<float*>(a.ldesc[0 * T + ti]);
    u.m = reinterpret_cast<float*>(a.ldesc[2 * T + ti]);
    u.v = reinterpret_cast<float*>(a.ldesc[3 * T + ti]);
    u.last = reinterpret_cast<int*>(a.ldesc[4 * T + ti]);
    const int64_t rows = a.ldesc[5 * T + ti];
    const int64_t K = a.ldesc[6 * T + ti];
    const int64_t w = a.ldesc[7 * T + ti];
    const int64_t wstart = ((int64_t)(t - 1) % K) * w;
    const int64_t local = (vb - a.vb_prefix[ti]) * RPB + slot;
    u.r = wstart + local;
    if (local >= w || u.r >= rows) return;
    u.off = u.r * D + (int64_t)q * (4 * VPL);
    u.old = gload<int>(u.last + u.r);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      u.P[k] = gload<float4>(u.p + u.off + 4 * k);
      u.M[k] = gload<float4>(u.m + u.off + 4 * k);
      u.V[k] = gload<float4>(u.v + u.off + 4 * k);
    }
    u.live = true;
  };
  auto process = [&](Unit& u) {
    const bool work = u.live && u.old < t;
    const int first = work ? u.old + 1 : t;
    int j = wave_min_uniform(first);
    while (j < t) {  // wavefront-uniform segments between the steps at which rows join (see lazy_sweep_body)
      const int nxt = wave_min_uniform(first > j ? first : t);
      if (first <= j) {
        int jj = j;
        for (; jj + 2 <= nxt; jj += 2) {
          const float2 ae0 = *reinterpret_cast<const float2*>(ring_s + 2 * (jj & a.ring_mask));
          const float2 ae1 = *reinterpret_cast<const float2*>(ring_s + 2 * ((jj + 1) & a.ring_mask));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < VPL; ++k) adam_f4_zero_g(u.P[k], u.M[k], u.V[k], h, ae0.x, ae0.y);
#pragma unroll
          for (int k = 0; k < VPL; ++k) adam_f4_zero_g(u.P[k], u.M[k], u.V[k], h, ae1.x, ae1.y);
        }
        if (jj < nxt) {
          const float2 ae = *reinterpret_cast<const float2*>(ring_s + 2 * (jj & a.ring_mask));
#pragma unroll
          for (int k = 0; k < VPL; ++k) adam_f4_zero_g(u.P[k], u.M[k], u.V[k], h, ae.x, ae.y);
        }
      }
      j = nxt;
    }
    if (work) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        adam_f4_zero_g(u.P[k], u.M[k], u.V[k], h, h.A, h.E);
        gstore<float4>(u.p + u.off + 4 * k, u.P[k]);
        gstore<float4>(u.m + u.off + 4 * k, u.M[k]);
        gstore<float4>(u.v + u.off + 4 * k, u.V[k]);
      }
      if (q == 0) u.last[u.r] = t;
    }
  };
  Unit ua, ub;
  fetch(bid, ua);
  for (int64_t vb = bid; vb < a.total_vblocks; vb += 2 * nblk) {
    fetch(vb + nblk, ub);
    process(ua);
    fetch(vb + 2 * nblk, ua);
    process(ub);
  }
}

template <int LPR, int VPL>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_sweep_wide_kernel(const LazySweepArgs a) {
  lazy_sweep_wide_body<LPR, VPL>(a, (int)blockIdx.x, (int)gridDim.x);
}

// One lane waits `ticks` of the constant wall clock (rh_adam_sweep_stagger).
__global__ void stream_delay_kernel(const long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// Gate words (int64[RH_GATE_WORDS]): gate[0] = count of openings, gate[1] = wall clock of the last opening, gate[2] = count of
// CHAIN STARTS (bumped by the last workgroup of the step's first own GEMM when it starts: rh_linear_fwd_gate, csrc/gemm.hip),
// gate[4 + 2 (i & 3)], gate[5 + 2 (i & 3)] = the chain-start count and the wall clock AT opening number i (a ring of four: a
// sweep is released at most three openings late).
// stream_gate_kernel: one lane waits until the count has reached `expected` -- the step's graph has reached its last launch --
// and then for the NEXT step's chain to have started, i.e. gate[2] > its value at that opening: the sweep's workgroups are
// then dispatched while that GEMM's workgroups are already placed, one per CU, and spread evenly over the SIMDs (dispatched
// TOGETHER with a launch of the chain they do not: 125 us instead of 26 for the chain's 240-register kernels in 15-30 % of the
// steps).  Round 4 approximated this point by a wall-clock hold-back behind the opening (22 us: 305 us steps,
// 28 us: 245 us, box-dependent -- VERDICT r04 weak 6); the clock is now only the FALLBACK: without a chain start within
// `ticks` of the opening (the last step of an epoch, a host that is late, a graph without an own GEMM) the sweep goes anyway.
// Gives up after `timeout` ticks without the opening itself and raises *err: a gate nobody opens must not wedge the queue.
__global__ void stream_gate_kernel(const long long* gate, const long long expected, const long long ticks, const long long timeout,
                                   int* err, long long* done_host, const long long done_value) {
  // (rh_adam_sweep_gate_done) Everything enqueued on this stream before this launch has completed -- the sweep of the step
  // before among it: say so in a word of host-mapped memory.  The host then knows how far the sweeps have come WITHOUT an
  // event record between two kernels of this stream (measured: 7.5 us of idle queue per step on what is the step's longer
  // path since round 6) and without a wait packet in front of the chain's graph.
  if (done_host != nullptr) __hip_atomic_store(done_host, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - expected < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout) {
      if (err != nullptr) atomicOr(err, RH_ERR_GATE_TIMEOUT);
      return;
    }
  }
  const int slot = 4 + 2 * (int)(expected & 3);
  const long long base = __hip_atomic_load(gate + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long opened = __hip_atomic_load(gate + slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (__hip_atomic_load(gate + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base < 1 && wall_clock64() - opened < ticks)
    __builtin_amdgcn_s_sleep(4);
}

// the opening (one lane): note the chain-start count and the time under this opening's number, then count it
static __device__ __forceinline__ void gate_open(long long* gate) {
  const long long idx = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;  // (one opener at a time)
  const int slot = 4 + 2 * (int)(idx & 3);
  const long long now = (long long)wall_clock64();
  __hip_atomic_store(gate + slot, __hip_atomic_load(gate + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(gate + slot + 1, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(gate + 1, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(gate, 1ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void stream_gate_open_kernel(long long* gate) { gate_open(gate); }
// rh_adam_sweep_release: counts a chain start that is not one -- the host knows that no further step follows the ones it has
// enqueued (end of an epoch, a synchronisation), so the last deferred sweep need not sit out its fallback
__global__ void stream_gate_release_kernel(long long* gate) {
  __hip_atomic_fetch_add(gate + 2, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// REFRESH: the pre-gather pass -- the rows carry no gradient yet (their gradient rows are zero), so they are neither
// read nor re-zeroed and the closing step is the zero-gradient form too (a quarter less traffic per row)
// LOOK (with REFRESH, ASSEMBLE): the workgroup walks `look` samples of the NEXT batch instead (dataset rows
// perm[(pos + B + b) mod N]) and refreshes only those of their rows that lie in the window the deferred sweep launched behind
// this pass is going to walk.  That sweep then finds every row the next batch reads already stamped and leaves it alone, so
// the next step's refresh (and gather) may run WHILE that sweep is still running: the sweep has to be done only before the
// refresh after that (optim.TableAdam, "relaxed join").  ~ B * F / K rows per step.
// GRAD (default: !REFRESH): the rows may carry a gradient -- it is read, applied in the closing step and re-zeroed.  REFRESH
// with GRAD is the refresh of the NEXT batch inside the end-of-step launch of THIS step (adam_lazy_step_ahead_kernel): a row
// both batches look up is claimed by one of the two passes, and whichever it is applies the gradient.
template <int LPR, typename IdxT, bool REFRESH, bool ASSEMBLE, bool LOOK, bool GRAD>
static __device__ __forceinline__ void lazy_touched_body(const LazyTouchedArgs& a, int bx, int f) {
  constexpr int LPP = RH_BLOCK / LPR;
  constexpr int D = 4 * LPR;
  const int T = a.T, F = a.F;
  const int64_t ti = a.field_table[f];
  if (ti < 0) return;
  const int64_t K = a.ldesc[6 * T + ti];
  if (K == 1) return;  // dense table: stepped (with its gradient) by the sweep pass
  const int64_t pad = a.field_table[F + f];
  float* p = reinterpret_cast<float*>(a.ldesc[0 * T + ti]);
  float* g = reinterpret_cast<float*>(a.ldesc[1 * T + ti]);
  float* m = reinterpret_cast<float*>(a.ldesc[2 * T + ti]);
  float* v = reinterpret_cast<float*>(a.ldesc[3 * T + ti]);
  int* last = reinterpret_cast<int*>(a.ldesc[4 * T + ti]);
  const int64_t rows = a.ldesc[5 * T + ti];
  const IdxT* ip = reinterpret_cast<const IdxT*>(a.idesc[f]);
  const int64_t st = a.idesc[F + f];
  // ASSEMBLE: the field's index column is column `acol` of the batch buffer, i.e. of the dataset rows
  const int64_t acol = ASSEMBLE ? (reinterpret_cast<const int64_t*>(ip) - a.sparse_out) : 0;
  const int64_t apos = ASSEMBLE ? a.pos[0] : 0;
  const AdamScalars h = load_scalars(a.hyper);
  const int t = (int)a.hyper[12];
  const int q = threadIdx.x % LPR;
  const int slot = threadIdx.x / LPR;
  const int lane = threadIdx.x % RH_WAVE;
  const int64_t nsamp = LOOK ? a.look_n : a.B;
  int64_t b0 = (int64_t)bx * (LOOK ? a.look : a.spb);
  int64_t b1 = (b0 + (LOOK ? a.look : a.spb) < nsamp) ? b0 + (LOOK ? a.look : a.spb) : nsamp;
  // REFRESH replays up to K steps per row: the per-step (A, E) ring entries come from LDS, as in the sweep.  Read from
  // global memory inside the replay loop they were one dependent L2 round trip per replayed step (the compiler emits a
  // vector load + s_waitcnt vmcnt per iteration): the pass was bound by that latency, 26.8 us in the DeepFM step.
  __shared__ float ring_t[REFRESH ? 2 * kMaxRing : 2];
  if (REFRESH) {
    for (int i = threadIdx.x; i < 2 * (a.ring_mask + 1); i += RH_BLOCK) ring_t[i] = a.ring[i];
    __syncthreads();
  }
  const float* ring = REFRESH ? ring_t : a.ring;
  // The lookups that carry the field's padding_idx are skipped below (a padded history batch would send a hundred thousand
  // of them to ONE claim word: 221 us instead of 81 for the 204 800 history lookups of configs[4]).  The padding row itself
  // is still kept exact -- zero or not, dense Adam moves it like every other row -- by ONE more pass of the first workgroup
  // of the field whose only live lookup is that row.  (As a separate block of code with its own replay loop in front of
  // this loop it cost the pass 26 -> 48 us in the DeepFM step, where no field has a padding row at all.)
  const bool pad_pass = !LOOK && REFRESH && pad >= 0 && pad < rows && bx == 0;  // block-uniform
  __shared__ int64_t s_look[LOOK ? RH_BLOCK : 1];
  __shared__ int s_nlook;
  const int64_t look_b1 = b1;
  for (int64_t sub = LOOK ? b0 : 0; sub < (LOOK ? look_b1 : 1); sub += RH_BLOCK) {
  if (LOOK) {
    // the samples sub .. sub + 255 of the next batch, one per thread: keep the rows inside the coming sweep's window
    const int64_t w = a.ldesc[7 * T + ti];
    const int64_t ws = ((int64_t)(t - 1) % K) * w;
    if (threadIdx.x == 0) s_nlook = 0;
    __syncthreads();  // (also: the previous round is done with s_look)
    const int64_t b = sub + threadIdx.x;
    if (b < look_b1) {
      const int64_t p = (apos + a.off + (int64_t)a.B + b) % a.N;
      const int64_t r = gload<int64_t>(a.src_sparse + gload<int64_t>(a.perm + p) * a.Fd + acol);
      if (r >= ws && r < ws + w && r < rows && r != pad) s_look[atomicAdd(&s_nlook, 1)] = r;
    }
    __syncthreads();
    b0 = 0;
    b1 = s_nlook;
  }
  const int64_t b_end = b1 + (pad_pass ? LPP : 0);
  for (int64_t base = b0; base < b_end; base += LPP) {  // uniform trip count: the claim is broadcast by shuffle
    const bool extra = base >= b1;
    const int64_t b = base + slot;
    const bool ok = !extra && b < b1;
    int64_t r;
    if (LOOK) {
      r = s_look[ok ? b : 0];
    } else if (ASSEMBLE) {
      int64_t p = apos + a.off + (ok ? b : b1 - 1);
      if (p >= a.N) p %= a.N;
      if (p < 0) p += a.N;
      r = gload<int64_t>(a.src_sparse + gload<int64_t>(a.perm + p) * a.Fd + acol);
    } else {
      r = (int64_t)gload<IdxT>(ip + (ok ? b : b1 - 1) * st);
    }
    bool valid = ok && (uint64_t)r < (uint64_t)rows && r != pad;
    if (extra) {
      r = pad;
      valid = slot == 0;
    }
    int old = t;
    if (valid && q == 0) {
      old = gload<int>(last + r);
      if (old < t) old = atomicMax(last + r, t);  // exactly one claimant sees a value < t
    }
    old = __shfl(old, lane - q, RH_WAVE);
    bool act = valid && old < t;  // this lane group claimed the row
    int64_t rr = r;
    if (REFRESH) {
      // The replay below costs a wavefront the LONGEST lag among its rows (the others idle under the exec mask), and the
      // rows of one pass lag anything from 0 to K steps: with 64 / LPR random rows per wavefront that is ~ K every time.
      // Re-deal the claimed rows of the pass over the lane groups in order of their lag (rank by counting through LDS):
      // the wavefronts then hold rows of similar lag and their maxima add up to ~ 5/8 of what they were.  Which lane
      // group replays a row changes nothing in its arithmetic.
      __shared__ int s_key[LPP];
      __shared__ int s_old[LPP];
      __shared__ int64_t s_row[LPP];
      const int lag = act ? t - old : 0;
      __syncthreads();  // the previous pass is done with the arrays
      if (q == 0) s_key[slot] = lag;
      __syncthreads();
      int rank = 0;
      // (unrolled by 4, not completely: with all 64 keys of a pass preloaded the LPR = 4 kernel took 110 VGPRs = 4 wavefronts
      // per SIMD, and the 6.5 wavefronts per SIMD of the DeepFM launch ran as two rounds of a latency-bound replay)
#pragma unroll 4
      for (int j = 0; j < LPP; ++j) {
        const int kj = s_key[j];
        rank += (kj > lag || (kj == lag && j < slot)) ? 1 : 0;
      }
      // wavefront w of every workgroup lands on the same SIMD of its CU: rotate the sorted order by a workgroup-dependent
      // number of wavefronts, or one SIMD would collect the longest-lag wavefront of every workgroup
      constexpr int kPerWave = RH_WAVE / LPR;
      rank = (rank + (bx % (LPP / kPerWave > 0 ? LPP / kPerWave : 1)) * kPerWave) % LPP;
      if (q == 0) {
        s_row[rank] = r;
        s_old[rank] = act ? old : t;  // t = nothing to do
      }
      __syncthreads();
      rr = s_row[slot];
      old = s_old[slot];
      act = old < t;
    }
    float4 P = f4_zero(), M = f4_zero(), V = f4_zero(), G = f4_zero();
    if (act) {
      P = gload<float4>(p + rr * D + q * 4);
      M = gload<float4>(m + rr * D + q * 4);
      V = gload<float4>(v + rr * D + q * 4);
      if (GRAD) G = gload<float4>(g + rr * D + q * 4);
    }
    // replay in segments between the steps at which rows of the wavefront join (see adam_lazy_sweep_kernel): fixed exec
    // mask and a scalar step counter inside a segment (the ring entry becomes a scalar load)
    const int first = act ? old + 1 : t;
    int j = wave_min_uniform(first);
    while (j < t) {
      const int nxt = wave_min_uniform(first > j ? first : t);
      if (first <= j) {
        for (int jj = j; jj < nxt; ++jj) {
          const float A = ring[2 * (jj & a.ring_mask)], E = ring[2 * (jj & a.ring_mask) + 1];
          adam_f4_zero_g(P, M, V, h, A, E);
        }
      }
      j = nxt;
    }
    if (!act) continue;
    if (GRAD) adam_f4(P, G, M, V, h, h.A, h.E);  // (an all-zero G gives the bits of the zero-gradient form)
    else adam_f4_zero_g(P, M, V, h, h.A, h.E);
    gstore<float4>(p + rr * D + q * 4, P);
    gstore<float4>(m + rr * D + q * 4, M);
    gstore<float4>(v + rr * D + q * 4, V);
    // (REFRESH with GRAD: most rows of the NEXT batch carry no gradient -- their gradient rows are zero already)
    if (GRAD && (!REFRESH || G.x != 0.f || G.y != 0.f || G.z != 0.f || G.w != 0.f)) gstore<float4>(g + rr * D + q * 4, f4_zero());
  }
  }
}

template <int LPR, typename IdxT, bool REFRESH>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_touched_kernel(const LazyTouchedArgs a) {
  RH_CHAIN_PRIO();
  lazy_touched_body<LPR, IdxT, REFRESH>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// Up to kTouchGroup touched-rows passes (one per gather of a step with several: two-tower / sequence models) as ONE launch
// (round 6): the refreshes in front of a step's first gather -- and the touched-rows steps at its end -- are independent passes
// over different index batches of the same table group; as dependent launches the short ones (16 k lookups) cost a launch each
// in front of the step's chain (DSSM: 14 + 51 + 16 us -> the longest).  Workgroups [prefix[i], prefix[i + 1]) run pass i.
constexpr int kTouchGroup = 4;
struct LazyTouchedGroupArgs {
  LazyTouchedArgs rec[kTouchGroup];
  int prefix[kTouchGroup + 1];
  int chunks[kTouchGroup];
  int i64[kTouchGroup];
  int n;
};

template <int LPR, bool REFRESH>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_touched_group_kernel(const LazyTouchedGroupArgs g) {
  RH_CHAIN_PRIO();
  const int bx = (int)blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < kTouchGroup; ++k) i += (k < g.n && bx >= g.prefix[k]) ? 1 : 0;
  const int local = bx - g.prefix[i];
  // (workgroup-uniform branches; a by-value copy keeps the pass's arguments in scalar registers)
#pragma unroll
  for (int k = 0; k < kTouchGroup; ++k) {
    if (k == i) {
      if (g.i64[k]) lazy_touched_body<LPR, int64_t, REFRESH>(g.rec[k], local % g.chunks[k], local / g.chunks[k]);
      else lazy_touched_body<LPR, int, REFRESH>(g.rec[k], local % g.chunks[k], local / g.chunks[k]);
      return;
    }
  }
}

// Batch assembly + pre-gather refresh as ONE launch (round 4; reference: TorchDataset.__getitem__ + default_collate,
// torch_rechub/utils/data.py:14-25,61-83, then the rows optimizer.step() would have left in the tables, trainers/ctr_trainer.py:99).
// Row blockIdx.y < F of the grid: the refresh of field f for its 64-sample chunk, the indices read from the dataset through
// perm (what rh_batch_gather would have written).  Row blockIdx.y == F: the assembly of that chunk into the static batch
// buffers (all sparse columns, dense columns, labels; 16 lanes per sample as batch_gather_kernel) -- nothing in THIS launch
// reads them, so the two parts need no ordering; the short copy runs under the refresh's latency chain.
template <int LPR>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_refresh_assemble_kernel(const LazyTouchedArgs a) {
  RH_CHAIN_PRIO();
  if ((int)blockIdx.y < a.F) {
    lazy_touched_body<LPR, int64_t, true, true>(a, (int)blockIdx.x, (int)blockIdx.y);
    return;
  }
  if ((int)blockIdx.y > a.F) {
    // lookahead rows: R = look / spb fields share one grid row (a lookahead workgroup walks R times the samples)
    const int R = a.look / a.spb;
    const int f = ((int)blockIdx.y - a.F - 1) * R + (int)blockIdx.x % R;
    if (f < a.F) lazy_touched_body<LPR, int64_t, true, true, true>(a, (int)blockIdx.x / R, f);
    return;
  }
  constexpr int G = 16;
  const int lig = threadIdx.x % G;
  const int64_t pos = a.pos[0];
  const int64_t b0 = (int64_t)blockIdx.x * a.spb;
  const int64_t b1 = (b0 + a.spb < (int64_t)a.B) ? b0 + a.spb : (int64_t)a.B;
  for (int64_t b = b0 + threadIdx.x / G; b < b1; b += RH_BLOCK / G) {
    int64_t p = pos + b;
    if (p >= a.N) p %= a.N;
    const int64_t src = a.perm[p];
    for (int j = lig; j < a.Fd; j += G) a.sparse_out[b * a.Fd + j] = a.src_sparse[src * a.Fd + j];
    for (int j = lig; j < a.ND; j += G) a.dense_out[b * a.ND + j] = a.src_dense[src * a.ND + j];
    if (lig == 0 && a.src_label != nullptr) a.label_out[b] = a.src_label[src];
  }
}

// The END of step t and the HEAD of step t + 1 as ONE launch (round 4, relaxed join with the head folded into the previous
// step's graph; reference: optimizer.step() of step t, trainers/ctr_trainer.py:99, then TorchDataset.__getitem__ +
// default_collate of batch t + 1, utils/data.py:14-25,61-83).  Parts, in workgroup order:
//   B  refresh of the lookups of batch t + 1 (dataset positions pos .. pos + B: the step's scalar launch has advanced pos),
//      replay form WITH gradient: a row that batch t also looked up is claimed by one of the two passes (atomicMax on its
//      last-step word, as in the merged touched + sweep launch) and the claimant applies the gradient;
//   A  the touched-rows step of batch t, its indices read from the dataset too (positions pos - B ..): the static batch buffer
//      they were gathered from is being overwritten by part D of this very launch;
//   C  lookahead: the lookups of batches t + 2 .. t + 1 + depth that fall into the window of the sweep launched behind this
//      step (LOOK above);
//   D  assembly of batch t + 1 into the static batch buffers;
//   E  the dense (K = 1) tables' step, as rh_adam_lazy_step_mode(RH_SWEEP_DENSE_TABLES).
// What the strict form ran as three dependent launches with two idle gaps between them (touched rows 22 us, gap, assembly +
// refresh 30 us, gap) overlaps inside one launch; the step's graph then begins with the gather.
struct StepAheadParts {
  int nB, chunksB;  // part B: chunksB x F workgroups of spbB samples
  int spbB;
  int nA, chunksA;  // part A
  int spbA;
  int nC, chunksC;  // part C: chunksC x F workgroups of `look` samples
  int nD;           // part D
  int w_order;      // adam_lazy_step_ahead_wgrad_kernel: how part W is dealt among the others (RH_TUNE_WGRAD_RIDER_ORDER)
  // rh_adam_lazy_step_ahead_touched (round 6, data parallel): part A walks BA rows of another index matrix -- the GATHERED
  // lookups of every rank's batch, columns idescA -- instead of the batch at dataset positions pos - B .. (BA = 0)
  const int64_t* idescA;
  int BA;
};

template <int LPR>
__device__ __forceinline__ void step_ahead_body(const LazySweepArgs& a, const StepAheadParts& parts, int bx, const int nblocks) {
  if (bx < parts.nB) {
    LazyTouchedArgs ta = a.touch;
    ta.spb = parts.spbB;
    ta.off = 0;
    lazy_touched_body<LPR, int64_t, true, true, false, true>(ta, bx % parts.chunksB, bx / parts.chunksB);
    return;
  }
  bx -= parts.nB;
  if (bx < parts.nA) {
    LazyTouchedArgs ta = a.touch;
    ta.spb = parts.spbA;
    if (parts.BA > 0) {  // (launch-uniform) the touched rows come from an index matrix of their own: the plain touched pass
      ta.idesc = parts.idescA;
      ta.B = parts.BA;
      ta.off = 0;
      lazy_touched_body<LPR, int64_t, false, false, false, true>(ta, bx % parts.chunksA, bx / parts.chunksA);
      return;
    }
    ta.off = -(int64_t)ta.B;
    lazy_touched_body<LPR, int64_t, false, true, false, true>(ta, bx % parts.chunksA, bx / parts.chunksA);
    return;
  }
  bx -= parts.nA;
  if (bx < parts.nC) {
    LazyTouchedArgs ta = a.touch;
    ta.off = 0;
    lazy_touched_body<LPR, int64_t, true, true, true, true>(ta, bx % parts.chunksC, bx / parts.chunksC);
    return;
  }
  bx -= parts.nC;
  if (bx < parts.nD) {
    const LazyTouchedArgs& ta = a.touch;
    constexpr int G = 16;
    const int lig = threadIdx.x % G;
    const int64_t pos = ta.pos[0];
    const int64_t b0 = (int64_t)bx * parts.spbB;
    const int64_t b1 = (b0 + parts.spbB < (int64_t)ta.B) ? b0 + parts.spbB : (int64_t)ta.B;
    for (int64_t b = b0 + threadIdx.x / G; b < b1; b += RH_BLOCK / G) {
      int64_t p = pos + b;
      if (p >= ta.N) p %= ta.N;
      const int64_t src = ta.perm[p];
      for (int j = lig; j < ta.Fd; j += G) ta.sparse_out[b * ta.Fd + j] = ta.src_sparse[src * ta.Fd + j];
      for (int j = lig; j < ta.ND; j += G) ta.dense_out[b * ta.ND + j] = ta.src_dense[src * ta.ND + j];
      if (lig == 0 && ta.src_label != nullptr) ta.label_out[b] = ta.src_label[src];
    }
    return;
  }
  bx -= parts.nD;
  lazy_sweep_body<LPR, true>(a, bx, nblocks - parts.nB - parts.nA - parts.nC - parts.nD);
}

template <int LPR>
__global__ __launch_bounds__(RH_BLOCK) void adam_lazy_step_ahead_kernel(const LazySweepArgs a, const StepAheadParts parts) {
  RH_CHAIN_PRIO();
  step_ahead_body<LPR>(a, parts, (int)blockIdx.x, (int)gridDim.x);
}

// ... with part W in front (round 6): the grouped weight gradients of the step's MLP chain (dW_l = g_l^T a_{l-1}, reference:
// the nn.Linear backward inside loss.backward(), trainers/ctr_trainer.py:98) as the FIRST nW workgroups of this launch.
// Nothing on the step's critical chain waits for them -- their slabs are summed by the packing launch BEHIND this one -- so
// they no longer occupy ~32 us of the chain between the last input-gradient GEMM and the gather's backward: their MFMA work
// runs beside the replay arithmetic (VALU) and the dependent row traffic of parts B / A.  Same workgroup body, same split
// plan per problem (rh_wgrad_group_fill) as rh_linear_wgrad_partial_group: the same slabs bit for bit.  Held to 128 registers
// (what both the 108-register parts above and the long-reduction build of the weight gradient fit into).
template <int LPR>
__global__ __launch_bounds__(RH_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void adam_lazy_step_ahead_wgrad_kernel(
    const LazySweepArgs a, const StepAheadParts parts, const rh_wgrad::WgradGroupArgs ga) {
  RH_CHAIN_PRIO();
  extern __shared__ float wred[];  // rh_wgrad::kPartStride floats
  const int nW = ga.prefix[rh_wgrad::kWgradGroup];
  const int nrest = (int)gridDim.x - nW;
  // Workgroup order (beside the optimizer's resident sweep a SIMD has room for two wavefronts of this launch).  Measured, same
  // box, ms per step (profiles/r06_ab_wgrad_rider.txt): no rider 0.2418; W in front of the other parts (w_order 0) 0.2315; dealt
  // alternately with them (1: pair p = (W[p], rest[p]), the order inside a pair flipping every 256 workgroups) 0.2378; W
  // BEHIND them (2, the default) 0.2268 -- the refresh part's workgroups run longest (a replay of up to lazy_k steps per row)
  // and want to be placed first; the short MFMA workgroups then fill the launch's tail.
  int bx = (int)blockIdx.x;
  bool is_w;
  int idx;
  if (parts.w_order == 1) {
    const int m = nW < nrest ? nW : nrest;
    if (bx < 2 * m) {
      is_w = (((bx & 1) ^ ((bx >> 8) & 1)) == 0);
      idx = bx >> 1;
    } else {
      is_w = nW > nrest;
      idx = bx - m;
    }
  } else if (parts.w_order == 2) {
    is_w = bx >= nrest;
    idx = is_w ? bx - nrest : bx;
  } else {
    is_w = bx < nW;
    idx = is_w ? bx : bx - nW;
  }
  if (is_w) {
    rh_wgrad::linear_wgrad_group_body<true>(ga, wred, idx);
    return;
  }
  step_ahead_body<LPR>(a, parts, idx, nrest);
}

int g_rider_order = 2;  // RH_TUNE_WGRAD_RIDER_ORDER: workgroup order of adam_lazy_step_ahead_wgrad_kernel (see there)
int g_sweep_wide = 2;  // RH_TUNE_SWEEP_WIDE: float4 per lane of the deferred lazy-table sweep at embed_dim >= 8 (2 = default; 1 = round-4 kernel)

// the deferred window sweep of the lazy tables, VPL float4 per lane (lazy_sweep_wide_body)
template <int LPR, int VPL>
int launch_sweep_wide(LazySweepArgs& a, const int64_t* h_rows, const int64_t* h_window, hipStream_t s) {
  constexpr int RPB = RH_BLOCK / LPR;
  a.vb_prefix[0] = 0;
  for (int t = 0; t < a.T; ++t) {
    int64_t w = h_window[t] < h_rows[t] ? h_window[t] : h_rows[t];
    if (h_window[t] >= h_rows[t]) w = 0;  // K_t == 1: a dense table, not this launch's
    a.vb_prefix[t + 1] = a.vb_prefix[t] + (w + RPB - 1) / RPB;
  }
  for (int t = a.T + 1; t <= kMaxTensors; ++t) a.vb_prefix[t] = a.vb_prefix[a.T];
  a.total_vblocks = a.vb_prefix[a.T];
  a.touch_blocks = a.touch_chunks = 0;
  a.touch_period = 1;
  if (a.total_vblocks == 0) return 0;
  int64_t grid = a.total_vblocks;
  const int64_t cap = g_deferred_grid > 0 ? g_deferred_grid : (g_sweep_grid > 0 ? g_sweep_grid : 256 * 32);
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((adam_lazy_sweep_wide_kernel<LPR, VPL>), dim3((unsigned)grid), dim3(RH_BLOCK), 0, s, a);
  return 0;
}

template <int LPR>
int launch_sweep(LazySweepArgs& a, int mode, const int64_t* h_rows, const int64_t* h_window, hipStream_t s,
                 const LazyTouchedArgs* touch = nullptr) {
  constexpr int RPB = RH_BLOCK / LPR;
  // (four float4 per lane, measured in round 5: 148 registers -- two such wavefronts leave a SIMD no room for the chain's
  // 235-register GEMM prologue -- 0.2696 ms per step against 0.2408; not built)
  if constexpr (LPR >= 2) {
    if (mode == RH_SWEEP_LAZY_TABLES && a.t_value >= 0 && touch == nullptr && g_sweep_wide >= 2)
      return launch_sweep_wide<LPR / 2, 2>(a, h_rows, h_window, s);
  }
  a.vb_prefix[0] = 0;
  for (int t = 0; t < a.T; ++t) {
    int64_t w = a.flush ? h_rows[t] : (h_window[t] < h_rows[t] ? h_window[t] : h_rows[t]);
    const bool dense_table = h_window[t] >= h_rows[t];  // K_t == 1
    if ((mode == RH_SWEEP_LAZY_TABLES && dense_table) || (mode == RH_SWEEP_DENSE_TABLES && !dense_table)) w = 0;
    a.vb_prefix[t + 1] = a.vb_prefix[t] + (w + RPB - 1) / RPB;
  }
  for (int t = a.T + 1; t <= kMaxTensors; ++t) a.vb_prefix[t] = a.vb_prefix[a.T];
  a.total_vblocks = a.vb_prefix[a.T];
  a.touch_blocks = a.touch_chunks = 0;
  a.touch_period = 1;
  if (touch != nullptr) {
    a.touch = *touch;
    a.touch_chunks = (touch->B + touch->spb - 1) / touch->spb;
    a.touch_blocks = a.touch_chunks * touch->F;
  }
  if (a.total_vblocks == 0 && a.touch_blocks == 0) return 0;
  // persistent-style grid: each workgroup walks several virtual blocks so the prefetch has something to overlap
  int64_t grid = a.total_vblocks;
  const int64_t cap = (a.t_value >= 0 && touch == nullptr && g_deferred_grid > 0) ? g_deferred_grid
                      : (g_sweep_grid > 0 ? g_sweep_grid : 256 * 32);
  if (grid > cap) grid = cap;
  if (touch != nullptr) {
    if (grid < 1) grid = 1;
    // every period-th workgroup is a touched one: (touch_blocks - 1) * period < grid + touch_blocks, so all of them exist;
    // fewer sweep than touched workgroups (small tables): period 1 = touched first
    int64_t period = (grid + a.touch_blocks) / a.touch_blocks;
    if (period > 1 && period % 8 == 0) period -= 1;  // block id mod 8 = XCD: keep the touched workgroups on all of them
    a.touch_period = (int)period;
    hipLaunchKernelGGL((adam_lazy_sweep_kernel<LPR, true>), dim3((unsigned)(grid + a.touch_blocks)), dim3(RH_BLOCK), 0, s,
                       a);
  } else {
    hipLaunchKernelGGL((adam_lazy_sweep_kernel<LPR, false>), dim3((unsigned)grid), dim3(RH_BLOCK), 0, s, a);
  }
  return 0;
}


// ------------------------------------------------------------------------------------------------------------
// Adam for the small dense parameters (MLP / LR / cross weights): ONE launch over all of them, gradients read from
// the packed flat bucket (the tensor the RCCL all-reduce runs on), element-granular so arbitrary numel / 4-byte
// alignment are fine.  Same adam_elem as the tables.  sdesc (device int64 [5*T]): p, m, v pointers, numel, offset of
// the parameter's gradient inside flat_g.
struct AdamSmallArgs {
  const int64_t* sdesc;
  const float* flat_g;
  const double* hyper;
  int T;
  int64_t total_vblocks;
  int64_t vb_prefix[kMaxTensors + 1];
};
constexpr int kSmallChunk = 1024;  // elements per virtual block (4 per thread, strided)

__global__ __launch_bounds__(RH_BLOCK) void adam_small_kernel(const AdamSmallArgs a) {
  RH_CHAIN_PRIO();
  const AdamScalars h = load_scalars(a.hyper);
  const int T = a.T;
  for (int64_t vb = blockIdx.x; vb < a.total_vblocks; vb += gridDim.x) {
    int lo = 0, hi = T;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.vb_prefix[mid] <= vb) lo = mid; else hi = mid;
    }
    const int t = lo;
    float* p = reinterpret_cast<float*>(a.sdesc[0 * T + t]);
    float* m = reinterpret_cast<float*>(a.sdesc[1 * T + t]);
    float* v = reinterpret_cast<float*>(a.sdesc[2 * T + t]);
    const int64_t n = a.sdesc[3 * T + t];
    const float* g = a.flat_g + a.sdesc[4 * T + t];
    const int64_t base = (vb - a.vb_prefix[t]) * kSmallChunk;
#pragma unroll
    for (int k = 0; k < kSmallChunk / RH_BLOCK; ++k) {
      const int64_t i = base + (int64_t)k * RH_BLOCK + threadIdx.x;
      if (i < n) {
        float P = p[i], M = m[i], V = v[i];
        adam_elem(P, g[i], M, V, h, h.A, h.E);
        p[i] = P;
        m[i] = M;
        v[i] = V;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Pack the dense (non-embedding) gradients of one step into the flat bucket the optimizer (and the RCCL all-reduce)
// reads, summing per-block / per-split partial slabs on the way: ONE launch instead of torch.cat + the trailing
// reduction launch of every backward kernel (wgrad_reduce x layers, colsum x 2 in the DeepFM step).
//   dst[off + i] = sum_{r < nparts} src[r * stride + i]  (+ add[i])      i < numel, fixed order r = 0, 1, ...
// nparts == 0 writes zeros (a parameter without gradient, like DDP).  The items travel BY VALUE in the kernel
// arguments: the slabs are temporaries whose addresses are only stable inside a captured hipGraph.
constexpr int kPackItems = 32;
struct PackArgs {
  RhPackItem it[kPackItems];
  int64_t vb_prefix[kPackItems + 1];
  int n;
  float* flat;
  // rh_pack_grads_adam: the Adam step of the parameter the item belongs to, on the gradient element just packed
  // (sdesc as rh_adam_small: [5 * T] p, m, v pointers, numel, flat offset; item i of this launch = parameter base + i)
  const int64_t* sdesc;
  const double* hyper;
  int T, base;
  long long* gate;  // rh_pack_grads_adam_gate: opened (rh_adam_sweep_gate_open's work) by the launch when it STARTS
};

static __device__ __forceinline__ void pack_adam(const PackArgs& a, const AdamScalars& h, int item, int64_t i, float g) {
  const int t = a.base + item;
  float* p = reinterpret_cast<float*>(a.sdesc[0 * a.T + t]);
  float* m = reinterpret_cast<float*>(a.sdesc[1 * a.T + t]);
  float* v = reinterpret_cast<float*>(a.sdesc[2 * a.T + t]);
  float P = p[i], M = m[i], V = v[i];
  adam_elem(P, g, M, V, h, h.A, h.E);
  p[i] = P;
  m[i] = M;
  v[i] = V;
}
constexpr int kPackChunk = RH_BLOCK;  // outputs per virtual block
constexpr int kPackDeep = 32;          // more partial rows than this: the rows are split over the wavefronts

template <bool ADAM>
static __device__ __forceinline__ void pack_body(const PackArgs& a, const int bx_, const int gdim_) {
  __shared__ float red[RH_BLOCK / RH_WAVE];
  AdamScalars h{};
  if (ADAM) h = load_scalars(a.hyper);
  for (int64_t vb = bx_; vb < a.vb_prefix[a.n]; vb += gdim_) {
    int lo = 0, hi = a.n;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.vb_prefix[mid] <= vb) lo = mid; else hi = mid;
    }
    const RhPackItem& it = a.it[lo];
    const float* src = reinterpret_cast<const float*>(it.src);
    const float* add = reinterpret_cast<const float*>(it.add);
    float* dst = a.flat + it.dst_offset;
    if (it.numel < 64 && it.nparts > 64) {
      // tall and thin (a bias: hundreds of partial rows of one float): the block strides over the rows
      __syncthreads();
      for (int64_t e = 0; e < it.numel; ++e) {
        float acc = 0.f;
        for (int64_t r = threadIdx.x; r < it.nparts; r += RH_BLOCK) acc += src[r * it.stride + e];
        acc = wave_sum(acc);
        if (threadIdx.x % RH_WAVE == 0) red[threadIdx.x / RH_WAVE] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
          const float gsum = (((red[0] + red[1]) + red[2]) + red[3]) + (add ? add[e] : 0.f);
          dst[e] = gsum;
          if (ADAM) pack_adam(a, h, lo, e, gsum);
        }
        __syncthreads();
      }
      continue;
    }
    if (it.nparts > kPackDeep) {
      // many partial rows (a per-block bias partial: hundreds of rows of a few hundred floats): one thread per output
      // would walk them as one dependent chain (measured: 86 us for 512 x 429 in the DCN-v2 step).  64 outputs per
      // virtual block, the 4 wavefronts take every 4th row (8 loads in flight each), partials summed in wavefront order.
      const int wave = threadIdx.x / RH_WAVE, lane = threadIdx.x % RH_WAVE;
      const int64_t i = (vb - a.vb_prefix[lo]) * RH_WAVE + lane;
      float v = 0.f;
      if (i < it.numel) {
        int64_t r = wave;
        for (; r + 28 < it.nparts; r += 32) {
          float t[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] = src[(r + 4 * k) * it.stride + i];
#pragma unroll
          for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; r < it.nparts; r += 4) v += src[r * it.stride + i];
      }
      __shared__ float deep[RH_BLOCK];
      __syncthreads();
      deep[threadIdx.x] = v;
      __syncthreads();
      if (wave == 0 && i < it.numel) {
        const float gsum = (((deep[lane] + deep[RH_WAVE + lane]) + deep[2 * RH_WAVE + lane]) + deep[3 * RH_WAVE + lane]) +
                           (add ? add[i] : 0.f);
        dst[i] = gsum;
        if (ADAM) pack_adam(a, h, lo, i, gsum);
      }
      continue;
    }
    const int64_t i = (vb - a.vb_prefix[lo]) * kPackChunk + threadIdx.x;
    if (i < it.numel) {
      float v = 0.f;
      int64_t r = 0;
      for (; r + 4 <= it.nparts; r += 4) {
        const float t0 = src[(r + 0) * it.stride + i], t1 = src[(r + 1) * it.stride + i];
        const float t2 = src[(r + 2) * it.stride + i], t3 = src[(r + 3) * it.stride + i];
        v = (((v + t0) + t1) + t2) + t3;
      }
      for (; r < it.nparts; ++r) v += src[r * it.stride + i];
      if (add) v += add[i];
      dst[i] = v;
      if (ADAM) pack_adam(a, h, lo, i, v);
    }
  }
}

template <bool ADAM>
__global__ __launch_bounds__(RH_BLOCK) void pack_grads_kernel(const PackArgs a) {
  RH_CHAIN_PRIO();
  if (a.gate != nullptr && blockIdx.x == 0 && threadIdx.x == 0) gate_open(a.gate);
  pack_body<ADAM>(a, (int)blockIdx.x, (int)gridDim.x);
}

// fills a.it / a.vb_prefix from the caller's items [base, base + a.n); returns false on a bad item
static bool pack_fill(PackArgs& a, const RhPackItem* items, int base) {
  a.vb_prefix[0] = 0;
  for (int i = 0; i < a.n; ++i) {
    const RhPackItem& it = items[base + i];
    if (!(it.numel >= 0 && it.nparts >= 0 && it.dst_offset >= 0 && (it.nparts == 0 || it.src != 0))) return false;
    a.it[i] = it;
    const bool tall = it.numel < 64 && it.nparts > 64;
    const int64_t per = it.nparts > kPackDeep ? RH_WAVE : kPackChunk;
    a.vb_prefix[i + 1] = a.vb_prefix[i] + (tall ? 1 : (it.numel + per - 1) / per);
  }
  for (int i = a.n; i < kPackItems; ++i) a.vb_prefix[i + 1] = a.vb_prefix[a.n];
  return true;
}

}  // namespace

static int pack_impl(const RhPackItem* items, int n, float* flat, const int64_t* sdesc, const double* hyper, void* stream,
                     int64_t* gate = nullptr) {
  RH_REQUIRE(items != nullptr && flat != nullptr && n >= 0, RH_E_BADARG, "rh_pack_grads: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n; base += kPackItems) {
    PackArgs a;
    a.n = n - base < kPackItems ? n - base : kPackItems;
    a.flat = flat;
    a.sdesc = sdesc;
    a.hyper = hyper;
    a.T = n;
    a.base = base;
    a.gate = base == 0 ? reinterpret_cast<long long*>(gate) : nullptr;
    RH_REQUIRE(pack_fill(a, items, base), RH_E_BADARG, "rh_pack_grads: bad item in [%d, %d)", base, base + a.n);
    if (a.vb_prefix[a.n] == 0) continue;
    int64_t grid = a.vb_prefix[a.n];
    if (grid > 4096) grid = 4096;
    if (sdesc != nullptr) hipLaunchKernelGGL(pack_grads_kernel<true>, dim3((unsigned)grid), dim3(RH_BLOCK), 0, st, a);
    else hipLaunchKernelGGL(pack_grads_kernel<false>, dim3((unsigned)grid), dim3(RH_BLOCK), 0, st, a);
  }
  RH_LAUNCH_CHECK("rh_pack_grads");
  return 0;
}

extern "C" int rh_pack_grads(const RhPackItem* items, int n, float* flat, void* stream) {
  return pack_impl(items, n, flat, nullptr, nullptr, stream);
}

// rh_pack_grads + rh_adam_small in one launch: item i is the gradient of parameter i of sdesc (n = T entries, the
// rh_adam_small layout), hyper holds THIS step's scalars already (rh_step_scalars / rh_adam_prepare ran before).
extern "C" int rh_pack_grads_adam(const RhPackItem* items, int n, float* flat, const int64_t* sdesc, const double* hyper,
                                  void* stream) {
  RH_REQUIRE(sdesc != nullptr && hyper != nullptr, RH_E_BADARG, "rh_pack_grads_adam: null pointer");
  return pack_impl(items, n, flat, sdesc, hyper, stream);
}

// rh_pack_grads_adam whose launch also opens a sweep gate (rh_adam_sweep_gate_open's work) when it starts: placed BEHIND the
// end-of-step table launch in a step's hipGraph, the packing launch announces that launch's end -- one launch fewer on the chain
// than a separate opening (4 us + a launch gap).  The packing and the table launch are independent of each other.
extern "C" int rh_pack_grads_adam_gate(const RhPackItem* items, int n, float* flat, const int64_t* sdesc, const double* hyper,
                                       int64_t* gate, void* stream) {
  RH_REQUIRE(sdesc != nullptr && hyper != nullptr && n >= 1, RH_E_BADARG, "rh_pack_grads_adam_gate: null pointer");
  return pack_impl(items, n, flat, sdesc, hyper, stream, gate);
}

extern "C" int rh_optim_set_tuning(int key, int value) {
  if (key == RH_TUNE_SWEEP_GRID) {
    g_sweep_grid = value;
    return 0;
  }
  if (key == RH_TUNE_DEFERRED_GRID) {
    g_deferred_grid = value;
    return 0;
  }
  if (key == RH_TUNE_SWEEP_STAGGER_NS) {
    g_stagger_ns = value;
    return 0;
  }
  if (key == RH_TUNE_SWEEP_GATE_NS) {
    g_gate_ns = value;
    return 0;
  }
  if (key == RH_TUNE_SWEEP_WIDE) {
    g_sweep_wide = value;
    return 0;
  }
  if (key == RH_TUNE_WGRAD_RIDER_ORDER) {
    g_rider_order = value;
    return 0;
  }
  return RH_E_BADARG;
}

extern "C" int rh_adam_prepare(double* hyper, int64_t* step, float* ring, int ring_size, void* stream) {
  RH_REQUIRE(hyper != nullptr && step != nullptr, RH_E_BADARG, "rh_adam_prepare: null pointer");
  RH_REQUIRE(ring == nullptr || (ring_size > 0 && (ring_size & (ring_size - 1)) == 0), RH_E_BADARG,
             "rh_adam_prepare: ring_size must be a power of two");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), hyper,
                     step, ring, (int64_t)(ring_size - 1));
  RH_LAUNCH_CHECK("rh_adam_prepare");
  return 0;
}

extern "C" int rh_adam_dense(const int64_t* tdesc, int T, const int64_t* h_numel, const double* hyper,
                             int zero_grad, void* stream) {
  RH_REQUIRE(tdesc != nullptr && h_numel != nullptr && hyper != nullptr, RH_E_BADARG,
             "rh_adam_dense: null pointer");
  RH_REQUIRE(T >= 1 && T <= kMaxTensors, RH_E_UNSUPPORTED, "rh_adam_dense: T=%d (max %d tensors per call)", T,
             kMaxTensors);
  AdamArgs a;
  a.tdesc = tdesc;
  a.hyper = hyper;
  a.T = T;
  a.zero_grad = zero_grad;
  a.vb_prefix[0] = 0;
  for (int t = 0; t < T; ++t) {
    RH_REQUIRE(h_numel[t] >= 0 && h_numel[t] % 4 == 0, RH_E_BADARG,
               "rh_adam_dense: numel[%d]=%lld must be a non-negative multiple of 4", t, (long long)h_numel[t]);
    const int64_t n4 = h_numel[t] / 4;
    a.vb_prefix[t + 1] = a.vb_prefix[t] + (n4 + kChunk4 - 1) / kChunk4;
  }
  for (int t = T + 1; t <= kMaxTensors; ++t) a.vb_prefix[t] = a.vb_prefix[T];
  a.total_vblocks = a.vb_prefix[T];
  if (a.total_vblocks == 0) return 0;
  int64_t grid = a.total_vblocks;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  RH_LAUNCH_CHECK("rh_adam_dense");
  return 0;
}

extern "C" int rh_adam_lazy_sweep(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                  const double* hyper, const float* ring, int ring_size, int mode, int64_t t_value,
                                  void* stream) {
  RH_REQUIRE(ldesc && h_rows && h_window && hyper && ring, RH_E_BADARG, "rh_adam_lazy_sweep: null pointer");
  RH_REQUIRE(mode >= RH_SWEEP_WINDOW && mode <= RH_SWEEP_DENSE_TABLES, RH_E_BADARG,
             "rh_adam_lazy_sweep: mode %d", mode);
  RH_REQUIRE(T >= 1 && T <= kMaxTensors, RH_E_UNSUPPORTED, "rh_adam_lazy_sweep: T=%d (max %d)", T, kMaxTensors);
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_sweep: ring_size must be a power of two <= %d", kMaxRing);
  LazySweepArgs a{};
  a.ldesc = ldesc;
  a.hyper = hyper;
  a.ring = ring;
  a.ring_mask = ring_size - 1;
  a.T = T;
  a.flush = mode == RH_SWEEP_FLUSH ? 1 : 0;
  a.t_value = t_value;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = RH_E_UNSUPPORTED;
  switch (D / 4) {
    case 1: rc = launch_sweep<1>(a, mode, h_rows, h_window, s); break;
    case 2: rc = launch_sweep<2>(a, mode, h_rows, h_window, s); break;
    case 4: rc = launch_sweep<4>(a, mode, h_rows, h_window, s); break;
    case 8: rc = launch_sweep<8>(a, mode, h_rows, h_window, s); break;
    case 16: rc = launch_sweep<16>(a, mode, h_rows, h_window, s); break;
    case 32: rc = launch_sweep<32>(a, mode, h_rows, h_window, s); break;
    default: break;
  }
  RH_REQUIRE(rc == 0 && D % 4 == 0, RH_E_UNSUPPORTED, "rh_adam_lazy_sweep: embed_dim %d unsupported", D);
  RH_LAUNCH_CHECK("rh_adam_lazy_sweep");
  return 0;
}

// Relaxed-join step (optim.py): the deferred sweep is released by the event behind the head's refresh and would start within
// a microsecond of the chain's first GEMM (the cross-queue latency happens to equal the gather in front of it).  Dispatched
// TOGETHER, the two launches interleave on the CUs, the sweep's resident wavefronts end up unevenly spread over the SIMDs, and
// in 15-30 % of the steps the chain's 240-register kernels (PRO GEMM, head) found no SIMD with room until sweep workgroups
// retired: 125 us instead of 26 (tools/step_stats.py, profiles/r04_relaxed_join.txt).  Released a few microseconds INTO that
// GEMM -- its workgroups already placed, one per CU -- the sweep spreads evenly: 0 slow steps of 300 at 5 / 7 / 11 us.  This
// one-lane launch in front of the sweep is that hold-back.  (A 55 KB LDS request per sweep workgroup and a per-CU admission
// counter were measured and did not fix it.)
static long long wall_khz() {
  static long long khz = 0;
  if (khz == 0) {
    int dev = 0, rate = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, dev) != hipSuccess ||
        rate <= 0)
      rate = 100000;
    khz = rate;
  }
  return khz;
}

extern "C" int rh_adam_sweep_stagger(void* stream) {
  if (g_stagger_ns <= 0) return 0;
  hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)g_stagger_ns * wall_khz() / 1000000);
  RH_LAUNCH_CHECK("rh_adam_sweep_stagger");
  return 0;
}

// The same hold-back released by a DEVICE word instead of an event (step-ahead form).  rh_adam_sweep_gate_open is captured as
// the LAST launch of the step's graph: one lane counts the opening and notes its time.  rh_adam_sweep_gate occupies `stream`
// until the count has reached `expected` and RH_TUNE_SWEEP_GATE_NS have passed SINCE THAT OPENING.  An event record between
// two graph launches cost the chain ~7 us of idle queue (this launch ~2); and a sweep that is released late -- the one before
// it ran long -- is not held back any further, where a fixed delay behind an event added itself to every sweep and left the
// side queue (delay + 231 us per step) no slack against a 245 us period.  A gate not opened within 2 s gives up and raises
// RH_ERR_GATE_TIMEOUT in *err_flag.
static int sweep_gate_impl(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag, int64_t* done_host,
                           int64_t done_value, void* stream);

extern "C" int rh_adam_sweep_gate(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag, void* stream) {
  return sweep_gate_impl(gate, expected, fallback_ns, err_flag, nullptr, 0, stream);
}

// rh_adam_sweep_gate that first stores done_value into *done_host (DEVICE address of host-mapped memory, rh_host_device_pointer):
// stream order makes that "every launch enqueued on `stream` before this one has completed" -- the caller's count of finished
// sweeps, readable by the host without an event.
extern "C" int rh_adam_sweep_gate_done(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag,
                                       int64_t* done_host, int64_t done_value, void* stream) {
  RH_REQUIRE(done_host != nullptr, RH_E_BADARG, "rh_adam_sweep_gate_done: null done_host");
  return sweep_gate_impl(gate, expected, fallback_ns, err_flag, done_host, done_value, stream);
}

// The device address of host memory allocated pinned + mapped (hipHostMalloc, what torch's pin_memory() uses); fails for
// memory the device cannot reach.
extern "C" int rh_host_device_pointer(void* host, void** device) {
  RH_REQUIRE(host != nullptr && device != nullptr, RH_E_BADARG, "rh_host_device_pointer: null pointer");
  hipError_t e = hipHostGetDevicePointer(device, host, 0);
  RH_REQUIRE(e == hipSuccess, (int)e, "rh_host_device_pointer: %s", hipGetErrorString(e));
  return 0;
}

static int sweep_gate_impl(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag, int64_t* done_host,
                           int64_t done_value, void* stream) {
  RH_REQUIRE(gate != nullptr, RH_E_BADARG, "rh_adam_sweep_gate: null gate");
  // fallback_ns > 0: the caller's graph counts chain starts (rh_linear_fwd_gate) and this is only the safety net behind the
  // opening; 0: RH_TUNE_SWEEP_GATE_NS -- for a graph without an own GEMM in front that hold-back IS the release, as in round 4
  const long long ns = fallback_ns > 0 ? (long long)fallback_ns : (long long)(g_gate_ns > 0 ? g_gate_ns : 0);
  hipLaunchKernelGGL(stream_gate_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(gate), (long long)expected, ns * wall_khz() / 1000000, 2000 * wall_khz(),
                     err_flag, reinterpret_cast<long long*>(done_host), (long long)done_value);
  RH_LAUNCH_CHECK("rh_adam_sweep_gate");
  return 0;
}

extern "C" int rh_adam_sweep_gate_open(int64_t* gate, void* stream) {
  RH_REQUIRE(gate != nullptr, RH_E_BADARG, "rh_adam_sweep_gate_open: null gate");
  hipLaunchKernelGGL(stream_gate_open_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<long long*>(gate));
  RH_LAUNCH_CHECK("rh_adam_sweep_gate_open");
  return 0;
}

extern "C" int rh_adam_sweep_release(int64_t* gate, void* stream) {
  RH_REQUIRE(gate != nullptr, RH_E_BADARG, "rh_adam_sweep_release: null gate");
  hipLaunchKernelGGL(stream_gate_release_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<long long*>(gate));
  RH_LAUNCH_CHECK("rh_adam_sweep_release");
  return 0;
}

extern "C" int rh_adam_lazy_touched(const int64_t* ldesc, int T, const int64_t* field_table, const int64_t* idesc,
                                    int idx_is_i64, int B, int F, int D, const double* hyper, const float* ring,
                                    int ring_size, int samples_per_block, int refresh, int32_t* err_flag,
                                    void* stream) {
  RH_REQUIRE(ldesc && field_table && idesc && hyper && ring, RH_E_BADARG, "rh_adam_lazy_touched: null pointer");
  RH_REQUIRE(T >= 1 && F >= 1 && F <= 65535 && B >= 0, RH_E_BADARG, "rh_adam_lazy_touched: bad shape");
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_touched: ring_size must be a power of two <= %d", kMaxRing);
  if (B == 0) return 0;
  int spb = samples_per_block <= 0 ? 256 : ((samples_per_block + 63) / 64) * 64;
  LazyTouchedArgs a{ldesc, field_table, idesc, hyper, ring, ring_size - 1, T, B, F, spb, err_flag};
  const dim3 grid((unsigned)((B + spb - 1) / spb), (unsigned)F);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define RH_LT(LPR)                                                                                             \
  if (idx_is_i64 && refresh)                                                                                   \
    hipLaunchKernelGGL((adam_lazy_touched_kernel<LPR, int64_t, true>), grid, dim3(RH_BLOCK), 0, s, a);        \
  else if (idx_is_i64)                                                                                         \
    hipLaunchKernelGGL((adam_lazy_touched_kernel<LPR, int64_t, false>), grid, dim3(RH_BLOCK), 0, s, a);       \
  else if (refresh)                                                                                            \
    hipLaunchKernelGGL((adam_lazy_touched_kernel<LPR, int32_t, true>), grid, dim3(RH_BLOCK), 0, s, a);        \
  else                                                                                                         \
    hipLaunchKernelGGL((adam_lazy_touched_kernel<LPR, int32_t, false>), grid, dim3(RH_BLOCK), 0, s, a);
  switch (D / 4) {
    case 1: RH_LT(1) break;
    case 2: RH_LT(2) break;
    case 4: RH_LT(4) break;
    case 8: RH_LT(8) break;
    case 16: RH_LT(16) break;
    case 32: RH_LT(32) break;
    default: rh_set_error("rh_adam_lazy_touched: embed_dim %d unsupported", D); return RH_E_UNSUPPORTED;
  }
#undef RH_LT
  RH_LAUNCH_CHECK("rh_adam_lazy_touched");
  return 0;
}

// n <= 4 rh_adam_lazy_touched passes over ONE table group (ldesc, T, D, hyper, ring) as one launch: arrays of n host entries
// (field_table / idesc: device pointers; idx_is_i64, B, F per pass).  refresh as rh_adam_lazy_touched, for all passes.
extern "C" int rh_adam_lazy_touched_group(const int64_t* ldesc, int T, int n, const int64_t* const* field_table,
                                          const int64_t* const* idesc, const int* idx_is_i64, const int* B, const int* F, int D,
                                          const double* hyper, const float* ring, int ring_size, int samples_per_block,
                                          int refresh, int32_t* err_flag, void* stream) {
  RH_REQUIRE(ldesc && field_table && idesc && idx_is_i64 && B && F && hyper && ring, RH_E_BADARG,
             "rh_adam_lazy_touched_group: null pointer");
  RH_REQUIRE(T >= 1 && n >= 1 && n <= kTouchGroup, RH_E_BADARG, "rh_adam_lazy_touched_group: %d passes (1..%d)", n, kTouchGroup);
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_touched_group: ring_size must be a power of two <= %d", kMaxRing);
  const int spb = samples_per_block <= 0 ? 256 : ((samples_per_block + 63) / 64) * 64;
  LazyTouchedGroupArgs g{};
  g.n = n;
  g.prefix[0] = 0;
  for (int i = 0; i < n; ++i) {
    RH_REQUIRE(field_table[i] && idesc[i] && F[i] >= 1 && F[i] <= 65535 && B[i] >= 0, RH_E_BADARG,
               "rh_adam_lazy_touched_group: bad pass %d", i);
    g.rec[i] = LazyTouchedArgs{ldesc, field_table[i], idesc[i], hyper, ring, ring_size - 1, T, B[i], F[i], spb, err_flag};
    g.chunks[i] = (B[i] + spb - 1) / spb;
    if (g.chunks[i] < 1) g.chunks[i] = 1;  // (an empty pass: its workgroups find no sample)
    g.i64[i] = idx_is_i64[i] ? 1 : 0;
    const int64_t blocks = B[i] > 0 ? (int64_t)g.chunks[i] * F[i] : 0;
    RH_REQUIRE((int64_t)g.prefix[i] + blocks < (1ll << 30), RH_E_UNSUPPORTED, "rh_adam_lazy_touched_group: grid too large");
    g.prefix[i + 1] = g.prefix[i] + (int)blocks;
  }
  for (int i = n; i < kTouchGroup; ++i) g.prefix[i + 1] = g.prefix[n], g.chunks[i] = 1;
  if (g.prefix[n] == 0) return 0;
  const dim3 grid((unsigned)g.prefix[n]);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define RH_LTG(LPR)                                                                                          \
  if (refresh) hipLaunchKernelGGL((adam_lazy_touched_group_kernel<LPR, true>), grid, dim3(RH_BLOCK), 0, s, g); \
  else hipLaunchKernelGGL((adam_lazy_touched_group_kernel<LPR, false>), grid, dim3(RH_BLOCK), 0, s, g);
  switch (D / 4) {
    case 1: RH_LTG(1) break;
    case 2: RH_LTG(2) break;
    case 4: RH_LTG(4) break;
    case 8: RH_LTG(8) break;
    case 16: RH_LTG(16) break;
    case 32: RH_LTG(32) break;
    default: rh_set_error("rh_adam_lazy_touched_group: embed_dim %d unsupported", D); return RH_E_UNSUPPORTED;
  }
#undef RH_LTG
  RH_LAUNCH_CHECK("rh_adam_lazy_touched_group");
  return 0;
}

extern "C" int rh_adam_lazy_refresh_assemble(const int64_t* ldesc, int T, const int64_t* field_table, const int64_t* idesc, int B,
                                             int F, int D, const double* hyper, const float* ring, int ring_size,
                                             int samples_per_block, int32_t* err_flag, const int64_t* perm, const int64_t* pos,
                                             int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND,
                                             const float* label, int64_t* sparse_out, float* dense_out, float* label_out,
                                             int lookahead, void* stream) {
  RH_REQUIRE(ldesc && field_table && idesc && hyper && ring && perm && pos && sparse && sparse_out, RH_E_BADARG,
             "rh_adam_lazy_refresh_assemble: null pointer");
  RH_REQUIRE(T >= 1 && F >= 1 && F <= 65534 && B >= 1 && N >= 1 && Fd >= 1 && ND >= 0, RH_E_BADARG,
             "rh_adam_lazy_refresh_assemble: bad shape");
  RH_REQUIRE(ND == 0 || (dense && dense_out), RH_E_BADARG, "rh_adam_lazy_refresh_assemble: dense pointers null");
  RH_REQUIRE(label == nullptr || label_out != nullptr, RH_E_BADARG, "rh_adam_lazy_refresh_assemble: label_out null");
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_refresh_assemble: ring_size must be a power of two <= %d", kMaxRing);
  const int spb = samples_per_block <= 0 ? 64 : ((samples_per_block + 63) / 64) * 64;
  const int chunks = (B + spb - 1) / spb;
  // lookahead workgroups walk R chunks each (one sample per thread and round), R fields share a grid row
  int R = RH_BLOCK / spb >= 1 ? RH_BLOCK / spb : 1;
  if (chunks % R != 0) R = 1;
  LazyTouchedArgs a{ldesc, field_table, idesc, hyper, ring, ring_size - 1, T, B, F, spb, err_flag,
                    perm, pos, N, sparse, Fd, dense, ND, label, sparse_out, dense_out, label_out, lookahead ? R * spb : 0,
                    lookahead ? B : 0, 0};
  const dim3 grid((unsigned)chunks, (unsigned)(F + 1 + (lookahead ? (F + R - 1) / R : 0)));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D / 4) {
    case 1: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<1>), grid, dim3(RH_BLOCK), 0, s, a); break;
    case 2: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<2>), grid, dim3(RH_BLOCK), 0, s, a); break;
    case 4: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<4>), grid, dim3(RH_BLOCK), 0, s, a); break;
    case 8: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<8>), grid, dim3(RH_BLOCK), 0, s, a); break;
    case 16: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<16>), grid, dim3(RH_BLOCK), 0, s, a); break;
    case 32: hipLaunchKernelGGL((adam_lazy_refresh_assemble_kernel<32>), grid, dim3(RH_BLOCK), 0, s, a); break;
    default: rh_set_error("rh_adam_lazy_refresh_assemble: embed_dim %d unsupported", D); return RH_E_UNSUPPORTED;
  }
  RH_LAUNCH_CHECK("rh_adam_lazy_refresh_assemble");
  return 0;
}

// ONE launch for the end of the step: the touched-rows step of the batch (as rh_adam_lazy_touched, refresh = 0) AND the
// window sweep of every table (as rh_adam_lazy_sweep, RH_SWEEP_WINDOW) -- the first is a short latency-bound pass of a few
// hundred workgroups, the second saturates the vector ALUs: run together, the first hides under the second.  int64 indices.
static int lazy_step_impl(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                          const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                          const int64_t* idesc, int B, int F, int samples_per_block, int32_t* err_flag, int sweep_mode,
                          int idx_is_i64, void* stream) {
  RH_REQUIRE(ldesc && h_rows && h_window && hyper && ring && field_table && idesc, RH_E_BADARG,
             "rh_adam_lazy_step: null pointer");
  RH_REQUIRE(T >= 1 && T <= kMaxTensors && F >= 1 && F <= 65535 && B >= 1, RH_E_BADARG, "rh_adam_lazy_step: bad shape");
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_step: ring_size must be a power of two <= %d", kMaxRing);
  const int spb = samples_per_block <= 0 ? 256 : ((samples_per_block + 63) / 64) * 64;
  LazyTouchedArgs ta{ldesc, field_table, idesc, hyper, ring, ring_size - 1, T, B, F, spb, err_flag};
  LazySweepArgs a{};
  a.ldesc = ldesc;
  a.hyper = hyper;
  a.ring = ring;
  a.ring_mask = ring_size - 1;
  a.T = T;
  a.flush = 0;
  a.t_value = -1;
  a.touch_i32 = idx_is_i64 ? 0 : 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = RH_E_UNSUPPORTED;
  switch (D / 4) {
    case 1: rc = launch_sweep<1>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    case 2: rc = launch_sweep<2>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    case 4: rc = launch_sweep<4>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    case 8: rc = launch_sweep<8>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    case 16: rc = launch_sweep<16>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    case 32: rc = launch_sweep<32>(a, sweep_mode, h_rows, h_window, s, &ta); break;
    default: break;
  }
  RH_REQUIRE(rc == 0 && D % 4 == 0, RH_E_UNSUPPORTED, "rh_adam_lazy_step: embed_dim %d unsupported", D);
  RH_LAUNCH_CHECK("rh_adam_lazy_step");
  return 0;
}

// rh_adam_lazy_step_mode(RH_SWEEP_DENSE_TABLES) of step t + rh_adam_lazy_refresh_assemble of batch t + 1 as ONE launch (see
// adam_lazy_step_ahead_kernel).  B is the size of BOTH batches; pos must already be advanced to batch t + 1; idesc describes
// index columns inside sparse_out (as rh_adam_lazy_refresh_assemble).  look_depth >= 0: batches looked ahead for the coming
// deferred sweep's window.
static int step_ahead_impl(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                           const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                           const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm, const int64_t* pos,
                           int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND, const float* label,
                           int64_t* sparse_out, float* dense_out, float* label_out, int look_depth,
                           const rh_wgrad::WgradGroupArgs* ga, const int64_t* touched_idesc, int touched_B, void* stream);

extern "C" int rh_adam_lazy_step_ahead(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                       const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                       const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm,
                                       const int64_t* pos, int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND,
                                       const float* label, int64_t* sparse_out, float* dense_out, float* label_out,
                                       int look_depth, void* stream) {
  return step_ahead_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, err_flag, perm, pos, N,
                         sparse, Fd, dense, ND, label, sparse_out, dense_out, label_out, look_depth, nullptr, nullptr, 0, stream);
}

// rh_adam_lazy_step_ahead whose touched-rows part walks touched_B rows of ANOTHER index matrix (touched_idesc: F column
// pointers + F strides, int64 indices) -- the gathered lookups of every rank's batch under data parallelism with replicated
// tables -- instead of the batch before the one it assembles.  Everything else as rh_adam_lazy_step_ahead.
extern "C" int rh_adam_lazy_step_ahead_touched(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                               const double* hyper, const float* ring, int ring_size,
                                               const int64_t* field_table, const int64_t* idesc, int B, int F, int32_t* err_flag,
                                               const int64_t* perm, const int64_t* pos, int64_t N, const int64_t* sparse, int Fd,
                                               const float* dense, int ND, const float* label, int64_t* sparse_out,
                                               float* dense_out, float* label_out, int look_depth,
                                               const int64_t* touched_idesc, int touched_B, void* stream) {
  RH_REQUIRE(touched_idesc != nullptr && touched_B >= 1, RH_E_BADARG, "rh_adam_lazy_step_ahead_touched: no touched index matrix");
  return step_ahead_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, err_flag, perm, pos, N,
                         sparse, Fd, dense, ND, label, sparse_out, dense_out, label_out, look_depth, nullptr, touched_idesc,
                         touched_B, stream);
}

// rh_adam_lazy_step_ahead whose launch also carries n <= 8 weight-gradient problems (arrays as rh_linear_wgrad_partial_group:
// problem i writes its split slabs to wpartial[i]) -- see adam_lazy_step_ahead_wgrad_kernel.
extern "C" int rh_adam_lazy_step_ahead_wgrad(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                             const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                             const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm,
                                             const int64_t* pos, int64_t N, const int64_t* sparse, int Fd, const float* dense,
                                             int ND, const float* label, int64_t* sparse_out, float* dense_out, float* label_out,
                                             int look_depth, int wn, const float* const* wg, const int64_t* wldg,
                                             const float* const* wx, const int64_t* wldx, const int* wB, const int* wN,
                                             const int* wK, float* const* wpartial, void* stream) {
  rh_wgrad::WgradGroupArgs ga;
  const int rc = rh_wgrad_group_fill(wn, wg, wldg, wx, wldx, wB, wN, wK, wpartial, &ga, "rh_adam_lazy_step_ahead_wgrad");
  if (rc != 0) return rc;
  return step_ahead_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, err_flag, perm, pos, N,
                         sparse, Fd, dense, ND, label, sparse_out, dense_out, label_out, look_depth, &ga, nullptr, 0, stream);
}

static int step_ahead_impl(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                           const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                           const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm, const int64_t* pos,
                           int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND, const float* label,
                           int64_t* sparse_out, float* dense_out, float* label_out, int look_depth,
                           const rh_wgrad::WgradGroupArgs* ga, const int64_t* touched_idesc, int touched_B, void* stream) {
  RH_REQUIRE(ldesc && h_rows && h_window && hyper && ring && field_table && idesc && perm && pos && sparse && sparse_out,
             RH_E_BADARG, "rh_adam_lazy_step_ahead: null pointer");
  RH_REQUIRE(T >= 1 && T <= kMaxTensors && F >= 1 && F <= 65535 && B >= 1 && N >= B && Fd >= 1 && ND >= 0 && look_depth >= 0 &&
                 look_depth <= 4, RH_E_BADARG, "rh_adam_lazy_step_ahead: bad shape");
  RH_REQUIRE(ND == 0 || (dense && dense_out), RH_E_BADARG, "rh_adam_lazy_step_ahead: dense pointers null");
  RH_REQUIRE(label == nullptr || label_out != nullptr, RH_E_BADARG, "rh_adam_lazy_step_ahead: label_out null");
  RH_REQUIRE(ring_size > 0 && (ring_size & (ring_size - 1)) == 0 && ring_size <= kMaxRing, RH_E_BADARG,
             "rh_adam_lazy_step_ahead: ring_size must be a power of two <= %d", kMaxRing);
  LazySweepArgs a{};
  a.ldesc = ldesc;
  a.hyper = hyper;
  a.ring = ring;
  a.ring_mask = ring_size - 1;
  a.T = T;
  a.flush = 0;
  a.t_value = -1;
  a.vb_prefix[0] = 0;
  for (int t = 0; t < T; ++t) {  // the dense (K = 1) tables only, as launch_sweep(RH_SWEEP_DENSE_TABLES)
    const bool dense_table = h_window[t] >= h_rows[t];
    const int64_t w = dense_table ? h_rows[t] : 0;
    const int rpb = RH_BLOCK / (D / 4);
    a.vb_prefix[t + 1] = a.vb_prefix[t] + (w + rpb - 1) / rpb;
  }
  for (int t = T + 1; t <= kMaxTensors; ++t) a.vb_prefix[t] = a.vb_prefix[T];
  a.total_vblocks = a.vb_prefix[T];
  a.touch_blocks = a.touch_chunks = 0;
  a.touch_period = 1;
  const int look = RH_BLOCK;
  a.touch = LazyTouchedArgs{ldesc, field_table, idesc, hyper, ring, ring_size - 1, T, B, F, 64, err_flag,
                            perm, pos, N, sparse, Fd, dense, ND, label, sparse_out, dense_out, label_out,
                            look, look_depth * B, 0};
  StepAheadParts parts;
  parts.spbB = 64;
  parts.chunksB = (B + parts.spbB - 1) / parts.spbB;
  parts.nB = parts.chunksB * F;
  parts.spbA = 256;
  parts.idescA = touched_idesc;
  parts.BA = touched_idesc != nullptr ? touched_B : 0;
  parts.chunksA = ((parts.BA > 0 ? parts.BA : B) + parts.spbA - 1) / parts.spbA;
  parts.nA = parts.chunksA * F;
  parts.chunksC = (look_depth * B + look - 1) / look;
  parts.nC = parts.chunksC * F;
  parts.nD = parts.chunksB;
  parts.w_order = g_rider_order;
  int64_t sweep_grid = a.total_vblocks;
  const int64_t cap = g_sweep_grid > 0 ? g_sweep_grid : 256 * 32;
  if (sweep_grid > cap) sweep_grid = cap;
  if (sweep_grid < 1) sweep_grid = 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (ga != nullptr) {
    const dim3 gridw((unsigned)(ga->prefix[rh_wgrad::kWgradGroup] + parts.nB + parts.nA + parts.nC + parts.nD + sweep_grid));
    const size_t lds = (size_t)rh_wgrad::kPartStride * sizeof(float);
    switch (D / 4) {
      case 1: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<1>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      case 2: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<2>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      case 4: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<4>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      case 8: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<8>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      case 16: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<16>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      case 32: hipLaunchKernelGGL((adam_lazy_step_ahead_wgrad_kernel<32>), gridw, dim3(RH_BLOCK), lds, s, a, parts, *ga); break;
      default: rh_set_error("rh_adam_lazy_step_ahead_wgrad: embed_dim %d unsupported", D); return RH_E_UNSUPPORTED;
    }
    RH_LAUNCH_CHECK("rh_adam_lazy_step_ahead_wgrad");
    return 0;
  }
  const dim3 grid((unsigned)(parts.nB + parts.nA + parts.nC + parts.nD + sweep_grid));
  switch (D / 4) {
    case 1: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<1>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    case 2: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<2>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    case 4: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<4>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    case 8: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<8>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    case 16: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<16>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    case 32: hipLaunchKernelGGL((adam_lazy_step_ahead_kernel<32>), grid, dim3(RH_BLOCK), 0, s, a, parts); break;
    default: rh_set_error("rh_adam_lazy_step_ahead: embed_dim %d unsupported", D); return RH_E_UNSUPPORTED;
  }
  RH_LAUNCH_CHECK("rh_adam_lazy_step_ahead");
  return 0;
}

extern "C" int rh_adam_small(const int64_t* sdesc, int T, const int64_t* h_numel, const float* flat_g,
                             const double* hyper, void* stream) {
  RH_REQUIRE(sdesc && h_numel && flat_g && hyper, RH_E_BADARG, "rh_adam_small: null pointer");
  RH_REQUIRE(T >= 1 && T <= kMaxTensors, RH_E_UNSUPPORTED, "rh_adam_small: T=%d (max %d tensors per call)", T,
             kMaxTensors);
  AdamSmallArgs a;
  a.sdesc = sdesc;
  a.flat_g = flat_g;
  a.hyper = hyper;
  a.T = T;
  a.vb_prefix[0] = 0;
  for (int t = 0; t < T; ++t) {
    RH_REQUIRE(h_numel[t] >= 0, RH_E_BADARG, "rh_adam_small: negative numel");
    a.vb_prefix[t + 1] = a.vb_prefix[t] + (h_numel[t] + kSmallChunk - 1) / kSmallChunk;
  }
  for (int t = T + 1; t <= kMaxTensors; ++t) a.vb_prefix[t] = a.vb_prefix[T];
  a.total_vblocks = a.vb_prefix[T];
  if (a.total_vblocks == 0) return 0;
  int64_t grid = a.total_vblocks;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adam_small_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     a);
  RH_LAUNCH_CHECK("rh_adam_small");
  return 0;
}

extern "C" int rh_adam_lazy_step(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                 const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                 const int64_t* idesc, int B, int F, int samples_per_block, int32_t* err_flag,
                                 void* stream) {
  return lazy_step_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, samples_per_block,
                        err_flag, RH_SWEEP_WINDOW, 1, stream);
}

// rh_adam_lazy_step with the sweep part restricted as rh_adam_lazy_sweep's `mode`: RH_SWEEP_DENSE_TABLES = the touched-rows
// step + the dense (K = 1) tables only -- the end of a step whose window sweep of the lazy tables is deferred to a side
// stream (torch_rechub_amd/optim.py, deferred form).
extern "C" int rh_adam_lazy_step_mode(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                      const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                      const int64_t* idesc, int B, int F, int samples_per_block, int32_t* err_flag,
                                      int sweep_mode, void* stream) {
  RH_REQUIRE(sweep_mode == RH_SWEEP_WINDOW || sweep_mode == RH_SWEEP_DENSE_TABLES, RH_E_BADARG,
             "rh_adam_lazy_step_mode: sweep_mode %d", sweep_mode);
  return lazy_step_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, samples_per_block,
                        err_flag, sweep_mode, 1, stream);
}

// rh_adam_lazy_step_mode for int32 OR int64 index columns (idx_is_i64 as rh_adam_lazy_touched): the row-sharded step's
// localised indices are int32 -- its touched-rows step and the dense tables' step were two launches (round 6: one).
extern "C" int rh_adam_lazy_step_mode_idx(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                          const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                          const int64_t* idesc, int idx_is_i64, int B, int F, int samples_per_block,
                                          int32_t* err_flag, int sweep_mode, void* stream) {
  RH_REQUIRE(sweep_mode == RH_SWEEP_WINDOW || sweep_mode == RH_SWEEP_DENSE_TABLES, RH_E_BADARG,
             "rh_adam_lazy_step_mode_idx: sweep_mode %d", sweep_mode);
  return lazy_step_impl(ldesc, T, h_rows, h_window, D, hyper, ring, ring_size, field_table, idesc, B, F, samples_per_block,
                        err_flag, sweep_mode, idx_is_i64, stream);
}
