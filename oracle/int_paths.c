/* Plain-C restatement of the integer / index paths of the CTR hot path -- a second, independent checker next to the
 * numpy oracle (oracle/ctr_oracle.py).  TEST INFRASTRUCTURE ONLY: nothing under torch_rechub_amd/ may link or load it.
 *
 *   o_embedding_gather   EmbeddingLayer.forward's per-field row gather, torch_rechub/basic/layers.py:83,110 (pure copy)
 *   o_batch_gather       TorchDataset.__getitem__ + default_collate over a permutation, torch_rechub/utils/data.py:14-25,61-83
 *   o_shard_localize     the lookup of layers.py:83-99 re-addressed for one rank's row-shard (no reference code; the
 *                        result it must reproduce is the plain gather -- see ctr_oracle.sharded_embedding_gather)
 *   o_inbatch_sample_rows  the HIP sampler's stream (csrc/data.hip): the reference draws randperm(B-1)[:K] per row
 *                        (utils/match.py:136-145) and pins only shape / no-self / distinctness in its tests
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -> oracle/_c/liboracle_int.so)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out (B, F, D) <- tables[f][idx[b, f], :]; returns the number of out-of-range indices (the reference raises). */
int64_t o_embedding_gather(const float* const* tables, const int64_t* vocab, const int64_t* idx, int64_t B, int F, int D,
                           float* out) {
  int64_t bad = 0;
  for (int64_t b = 0; b < B; ++b)
    for (int f = 0; f < F; ++f) {
      const int64_t g = idx[b * F + f];
      float* dst = out + (b * F + f) * D;
      if (g < 0 || g >= vocab[f]) {
        ++bad;
        memset(dst, 0, (size_t)D * sizeof(float));
      } else {
        memcpy(dst, tables[f] + g * D, (size_t)D * sizeof(float));
      }
    }
  return bad;
}

/* rows perm[(pos + b) mod N] of the columnar dataset into the batch buffers */
void o_batch_gather(const int64_t* perm, int64_t pos, int64_t N, int64_t B, const int64_t* sparse, int F, const float* dense,
                    int ND, const float* label, int64_t* sparse_out, float* dense_out, float* label_out) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t src = perm[(pos + b) % N];
    for (int j = 0; j < F; ++j) sparse_out[b * F + j] = sparse[src * F + j];
    for (int j = 0; j < ND; ++j) dense_out[b * ND + j] = dense[src * ND + j];
    if (label) label_out[b] = label[src];
  }
}

/* local[i, f] = g / world if g % world == rank and g != pad[f], else ceil(vocab[f] / world) (the sink row);
 * returns the number of out-of-range ids (mapped to the sink row, like the kernel does while raising its flag). */
int64_t o_shard_localize(const int64_t* idx, int64_t n_rows, int F, const int64_t* vocab, const int64_t* pad, int world,
                         int rank, int32_t* local) {
  int64_t bad = 0;
  for (int64_t i = 0; i < n_rows; ++i)
    for (int f = 0; f < F; ++f) {
      const int64_t g = idx[i * F + f];
      const int64_t sink = (vocab[f] + world - 1) / world;
      int64_t out = sink;
      if (g < 0 || g >= vocab[f])
        ++bad;
      else if (g != pad[f] && g % world == rank)
        out = g / world;
      local[i * F + f] = (int32_t)out;
    }
  return bad;
}

static uint32_t sample_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed + ctr * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

/* out (B, K): global row r = row0 + i draws K distinct columns of {0..cols-1} \ {r} (Floyd's subset algorithm over
 * the cols - 1 other columns, counter-hash stream keyed by (seed, ctr, r * K + draw)).  Returns 0, or -1 on bad sizes. */
int o_inbatch_sample_rows(uint64_t seed, uint64_t ctr, int B, int cols, int row0, int K, int64_t* out) {
  if (B < 1 || cols < 2 || row0 < 0 || row0 + B > cols || K < 1 || K > cols - 1) return -1;
  unsigned char* taken = (unsigned char*)malloc((size_t)cols);
  if (!taken) return -1;
  const int n = cols - 1;
  for (int i = 0; i < B; ++i) {
    const uint32_t own = (uint32_t)(row0 + i);
    memset(taken, 0, (size_t)cols);
    int pos = 0;
    for (int j = n - K; j < n; ++j, ++pos) {
      const uint32_t t = sample_hash(seed, ctr, (uint64_t)own * (uint64_t)K + (uint64_t)pos) % (uint32_t)(j + 1);
      const uint32_t pick = taken[t] ? (uint32_t)j : t;
      taken[pick] = 1;
      out[(int64_t)i * K + pos] = (int64_t)pick + (pick >= own ? 1 : 0);
    }
  }
  free(taken);
  return 0;
}
