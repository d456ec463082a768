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

namespace {

constexpr int kMaxTensors = 128;
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

__global__ void adam_prepare_kernel(double* hyper, int64_t* step) {
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
  }
}

struct AdamScalars {
  float b2, eps, wd, step_size, bc2_sqrt, one_m_b1, one_m_b2;
};

// One element of torch.optim.Adam (_single_tensor_adam, amsgrad=False, maximize=False):
//   g += wd*p; m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
//   denom = v.sqrt()/bc2_sqrt + eps; p.addcdiv_(m, denom, -step_size)
static __device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& h) {
  g = fmaf(h.wd, p, g);
  m = fmaf(h.one_m_b1, g - m, m);
  v = fmaf(h.one_m_b2, g * g, v * h.b2);
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p = fmaf(-h.step_size, m / denom, p);
}

__global__ __launch_bounds__(RH_BLOCK) void adam_dense_kernel(const AdamArgs a) {
  AdamScalars h;
  h.b2 = (float)a.hyper[2];
  h.eps = (float)a.hyper[3];
  h.wd = (float)a.hyper[4];
  h.step_size = (float)a.hyper[8];
  h.bc2_sqrt = (float)a.hyper[9];
  h.one_m_b1 = (float)a.hyper[10];
  h.one_m_b2 = (float)a.hyper[11];
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
      adam_elem(P[k].x, Gv[k].x, M[k].x, V[k].x, h);
      adam_elem(P[k].y, Gv[k].y, M[k].y, V[k].y, h);
      adam_elem(P[k].z, Gv[k].z, M[k].z, V[k].z, h);
      adam_elem(P[k].w, Gv[k].w, M[k].w, V[k].w, h);
      gstore<float4>(p + i4 * 4, P[k]);
      gstore<float4>(m + i4 * 4, M[k]);
      gstore<float4>(v + i4 * 4, V[k]);
      if (a.zero_grad && gnz) gstore<float4>(g + i4 * 4, f4_zero());
    }
  }
}

}  // namespace

extern "C" int rh_adam_prepare(double* hyper, int64_t* step, void* stream) {
  RH_REQUIRE(hyper != nullptr && step != nullptr, RH_E_BADARG, "rh_adam_prepare: null pointer");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), hyper,
                     step);
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
