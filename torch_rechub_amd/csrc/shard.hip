// Row-sharded embedding tables: index localisation.
//
// A table of `vocab` rows is dealt out row by row: global row g lives on rank g % world as local row g / world (hot
// rows of a Zipf-distributed field are low ids; dealing spreads them over all ranks).  Every rank receives the index
// matrix of the WHOLE global batch (all-gather, 4-8 B per lookup) and rewrites it for its own shard: a lookup it owns
// becomes the local row, every other lookup -- and the table's padding row -- becomes the shard's `sink` row, a row of
// zeros that the gather reads as zeros and that the backward / optimizer kernels skip as padding_idx.  The masked
// gathers of all ranks are then summed by a reduce-scatter: exactly one rank contributes a non-zero row per lookup,
// so the sum is the row itself whatever the reduction order.
//
// Reference semantics reproduced: EmbeddingLayer.forward over nn.Embedding(vocab, D, padding_idx)
// (torch_rechub/basic/layers.py:77-127, basic/initializers.py:16-21); an index outside [0, vocab) raises there
// (IndexError / device assert) and sets RH_FLAG_INDEX_OOB here.
// Roofline: HBM, 12-16 B per lookup (read 4/8, write 4); N*F is a few hundred thousand elements.
#include "common.h"

namespace {

template <typename IdxT>
__global__ __launch_bounds__(RH_BLOCK) void shard_localize_kernel(const IdxT* __restrict__ idx, int64_t n, int F,
                                                                  const int64_t* __restrict__ desc, uint32_t world,
                                                                  uint32_t rank, int32_t* __restrict__ local,
                                                                  int* err) {
  RH_CHAIN_PRIO();
  const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int f = (int)(i % F);
  const int64_t vocab = desc[f], pad = desc[F + f], sink = desc[2 * F + f];
  const int64_t g = (int64_t)idx[i];
  int64_t out = sink;
  if ((uint64_t)g >= (uint64_t)vocab) {
    if (err != nullptr) atomicOr(err, RH_FLAG_INDEX_OOB);
  } else if (g != pad) {
    // vocab < 2^31 * world is checked by the host, so the quotient fits; 32-bit division when the id does
    const bool narrow = g <= 0xffffffffll;
    const uint64_t q = narrow ? (uint64_t)((uint32_t)g / world) : (uint64_t)g / world;
    const uint64_t r = narrow ? (uint64_t)((uint32_t)g % world) : (uint64_t)g % world;
    if (r == rank) out = (int64_t)q;
  }
  local[i] = (int32_t)out;
}

// int64 ids -> int32 for the wire, saturating: an id beyond int32 stays out of every table's range (RH_FLAG_INDEX_OOB at the
// localisation), it does not wrap onto a valid row; -1 and below -> -1.  Row r of the (rows, F) matrix is row stride `ld`.
__global__ __launch_bounds__(RH_BLOCK) void shard_narrow_kernel(const int64_t* __restrict__ idx, int64_t ld, int64_t n, int F,
                                                                int32_t* __restrict__ out) {
  RH_CHAIN_PRIO();
  const int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int64_t g = idx[(i / F) * ld + (i % F)];
  out[i] = (int32_t)(g < -1 ? -1 : (g > 2147483647ll ? 2147483647ll : g));
}

}  // namespace

// The (rows, F) int64 index matrix (row stride ld) as the contiguous int32 matrix that travels in the index all-gather of the
// row-sharded lookup -- written straight into this rank's slice of the gather buffer, so that the collective runs in place
// (round 6: `idx.clamp(-1, 2^31 - 1).to(int32)` + the copy into the buffer were three launches of the sharded step's head).
extern "C" int rh_shard_narrow(const int64_t* idx, int64_t ld, int64_t n_rows, int F, int32_t* out, void* stream) {
  RH_REQUIRE(n_rows >= 0 && F >= 1 && ld >= F, RH_E_BADARG, "rh_shard_narrow: bad shape (%lld, %d) ld %lld", (long long)n_rows, F,
             (long long)ld);
  if (n_rows == 0) return 0;
  RH_REQUIRE(idx && out, RH_E_BADARG, "rh_shard_narrow: null pointer");
  const int64_t n = n_rows * F;
  hipLaunchKernelGGL(shard_narrow_kernel, dim3((unsigned)((n + RH_BLOCK - 1) / RH_BLOCK)), dim3(RH_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), idx, ld, n, F, out);
  RH_LAUNCH_CHECK("rh_shard_narrow");
  return 0;
}

extern "C" int rh_shard_localize(const void* idx, int idx_is_i64, int64_t n_rows, int F, const int64_t* desc,
                                 int world, int rank, int32_t* local, int32_t* err_flag, void* stream) {
  RH_REQUIRE(n_rows >= 0 && F >= 1, RH_E_BADARG, "rh_shard_localize: bad shape (%lld, %d)", (long long)n_rows, F);
  RH_REQUIRE(world >= 1 && rank >= 0 && rank < world, RH_E_BADARG, "rh_shard_localize: rank %d of %d", rank, world);
  if (n_rows == 0) return 0;
  RH_REQUIRE(idx && desc && local, RH_E_BADARG, "rh_shard_localize: null pointer");
  const int64_t n = n_rows * F;
  const unsigned grid = (unsigned)((n + RH_BLOCK - 1) / RH_BLOCK);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (idx_is_i64)
    hipLaunchKernelGGL(shard_localize_kernel<int64_t>, dim3(grid), dim3(RH_BLOCK), 0, s,
                       static_cast<const int64_t*>(idx), n, F, desc, (uint32_t)world, (uint32_t)rank, local, err_flag);
  else
    hipLaunchKernelGGL(shard_localize_kernel<int32_t>, dim3(grid), dim3(RH_BLOCK), 0, s,
                       static_cast<const int32_t*>(idx), n, F, desc, (uint32_t)world, (uint32_t)rank, local, err_flag);
  RH_LAUNCH_CHECK("rh_shard_localize");
  return 0;
}
