// Device-resident minibatch assembly: the columnar dataset lives in HBM, a batch is a gather of
// perm[pos .. pos+B) rows into static buffers (hipGraph friendly: pointers never change).
//
// Reference: TorchDataset.__getitem__ + DataLoader default_collate, torch_rechub/utils/data.py:14-25,61-83
//            and the per-column host->device copies at torch_rechub/trainers/ctr_trainer.py:84-85.
// Roofline: HBM, 2 x (F*8 + ND*4 + 4) bytes per sample (read + write), rows are contiguous.
#include "common.h"

namespace {

// one sample per group of 16 lanes; 8-byte index elements, 4-byte dense elements
__global__ __launch_bounds__(RH_BLOCK) void batch_gather_kernel(const int64_t* __restrict__ perm,
                                                                const int64_t* __restrict__ pos, int64_t N,
                                                                int B, const int64_t* __restrict__ sparse, int F,
                                                                const float* __restrict__ dense, int ND,
                                                                const float* __restrict__ label,
                                                                int64_t* __restrict__ sparse_out,
                                                                float* __restrict__ dense_out,
                                                                float* __restrict__ label_out) {
  constexpr int G = 16;
  const int lig = threadIdx.x % G;
  const int64_t b = (int64_t)blockIdx.x * (RH_BLOCK / G) + threadIdx.x / G;
  if (b >= B) return;
  int64_t p = *pos + b;
  if (p >= N) p %= N;  // wrap: an epoch boundary inside a batch reuses the head of the permutation
  const int64_t src = perm[p];
  for (int j = lig; j < F; j += G) sparse_out[b * F + j] = sparse[src * F + j];
  for (int j = lig; j < ND; j += G) dense_out[b * ND + j] = dense[src * ND + j];
  if (lig == 0 && label != nullptr) label_out[b] = label[src];
}

__global__ void batch_advance_kernel(int64_t* pos, int64_t B, int64_t N) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t p = *pos + B;
    if (N > 0 && p >= N) p %= N;
    *pos = p;
  }
}

}  // namespace

extern "C" int rh_batch_gather(const int64_t* perm, const int64_t* pos, int64_t N, int B, const int64_t* sparse,
                               int F, const float* dense, int ND, const float* label, int64_t* sparse_out,
                               float* dense_out, float* label_out, void* stream) {
  RH_REQUIRE(perm && pos && N > 0, RH_E_BADARG, "rh_batch_gather: null perm/pos or N <= 0");
  RH_REQUIRE(F == 0 || (sparse && sparse_out), RH_E_BADARG, "rh_batch_gather: sparse pointers null");
  RH_REQUIRE(ND == 0 || (dense && dense_out), RH_E_BADARG, "rh_batch_gather: dense pointers null");
  RH_REQUIRE(label == nullptr || label_out != nullptr, RH_E_BADARG, "rh_batch_gather: label_out null");
  if (B <= 0) return 0;
  const unsigned grid = (unsigned)(((int64_t)B + RH_BLOCK / 16 - 1) / (RH_BLOCK / 16));
  hipLaunchKernelGGL(batch_gather_kernel, dim3(grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     perm, pos, N, B, sparse, F, dense, ND, label, sparse_out, dense_out, label_out);
  RH_LAUNCH_CHECK("rh_batch_gather");
  return 0;
}

extern "C" int rh_batch_advance(int64_t* pos, int64_t B, int64_t N, void* stream) {
  RH_REQUIRE(pos != nullptr, RH_E_BADARG, "rh_batch_advance: null pos");
  hipLaunchKernelGGL(batch_advance_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pos, B, N);
  RH_LAUNCH_CHECK("rh_batch_advance");
  return 0;
}
