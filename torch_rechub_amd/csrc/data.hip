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
  RH_CHAIN_PRIO();
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

// In-batch negative sampling: row i draws K distinct columns uniformly from {0..B-1} \ {i} (Floyd's algorithm, one
// thread per row, membership bitmap of the row in LDS).  Counter-based hash RNG (seed, call counter, row, draw): no
// library RNG state, hipGraph-replayable; the caller bumps the counter with rh_batch_advance(rng + 1, 1, 0).
// Reference: inbatch_negative_sampling torch_rechub/utils/match.py:136-145 (Python loop of randperm per row).
static __device__ __forceinline__ uint32_t sample_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed + ctr * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// Rectangular form (cross-rank negatives): the rows are rows [row0, row0 + B) of a (cols x cols) global problem, row r
// draws from the columns {0..cols-1} \ {r}; the RNG is keyed by the GLOBAL row, so the ranks of a job draw what one
// process would draw for the same rows.
__global__ void inbatch_sample_kernel(const int64_t* __restrict__ rng, int B, int cols, int row0, int K,
                                      int words_per_row, int64_t* __restrict__ out) {
  extern __shared__ uint32_t bitmap[];  // [rows per block][words_per_row]
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* mine = bitmap + (size_t)threadIdx.x * words_per_row;
  for (int w = 0; w < words_per_row; ++w) mine[w] = 0u;
  if (row >= B) return;
  const uint64_t seed = (uint64_t)rng[0], ctr = (uint64_t)rng[1];
  const int N = cols - 1;  // candidates: every column but the row's own
  const uint32_t own = (uint32_t)(row0 + row);
  int pos = 0;
  for (int j = N - K; j < N; ++j) {
    const uint32_t t = sample_hash(seed, ctr, (uint64_t)own * (uint64_t)K + (uint64_t)pos) % (uint32_t)(j + 1);
    const bool taken = (mine[t >> 5] >> (t & 31)) & 1u;
    const uint32_t pick = taken ? (uint32_t)j : t;
    mine[pick >> 5] |= 1u << (pick & 31);
    out[(int64_t)row * K + pos] = (int64_t)pick + (pick >= own ? 1 : 0);
    ++pos;
  }
}

}  // namespace

extern "C" int rh_inbatch_sample_rows(const int64_t* rng, int B, int cols, int row0, int K, int64_t* out,
                                      void* stream) {
  RH_REQUIRE(rng && out, RH_E_BADARG, "rh_inbatch_sample: null pointer");
  RH_REQUIRE(B >= 1 && cols >= 2 && row0 >= 0 && row0 + B <= cols, RH_E_BADARG,
             "rh_inbatch_sample: rows [%d, %d) outside the %d columns", row0, row0 + B, cols);
  RH_REQUIRE(K >= 1 && K <= cols - 1, RH_E_BADARG, "rh_inbatch_sample: need 1 <= K <= columns-1 (columns=%d K=%d)",
             cols, K);
  const int words = (cols + 31) / 32;
  int rows = 64;
  while (rows > 1 && (size_t)rows * words * 4 > 48 * 1024) rows /= 2;
  RH_REQUIRE((size_t)rows * words * 4 <= 64 * 1024, RH_E_UNSUPPORTED, "rh_inbatch_sample: %d columns too many", cols);
  const unsigned grid = (unsigned)((B + rows - 1) / rows);
  hipLaunchKernelGGL(inbatch_sample_kernel, dim3(grid), dim3(rows), (size_t)rows * words * 4,
                     reinterpret_cast<hipStream_t>(stream), rng, B, cols, row0, K, words, out);
  RH_LAUNCH_CHECK("rh_inbatch_sample");
  return 0;
}

extern "C" int rh_inbatch_sample(const int64_t* rng, int B, int K, int64_t* out, void* stream) {
  RH_REQUIRE(B >= 2, RH_E_BADARG, "rh_inbatch_sample: need 1 <= K <= B-1 (B=%d K=%d)", B, K);
  return rh_inbatch_sample_rows(rng, B, B, 0, K, out, stream);
}

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
