// Sequence-feature gather + masked pooling (sum / mean / concat), forward and backward.
//
// Reference: EmbeddingLayer.forward sequence branch torch_rechub/basic/layers.py:86-99
//            InputMask.forward       torch_rechub/basic/layers.py:148-161  (mask = idx != padding_idx, or != -1)
//            SumPooling / AveragePooling / ConcatPooling  torch_rechub/basic/layers.py:204-251
//              sum  = bmm(mask, emb);  mean = sum / (mask.sum + 1e-16);  concat = emb (mask ignored)
//
// Roofline: HBM random gather, L rows of D*4 bytes per sample.  A sample is handled by G = LPR*LS lanes
// (LPR = D/4 lanes per row, LS interleaved position groups) so that all of a lane's row gathers are in
// flight together; the pooling reduction is an xor-shuffle over the LS groups.
#include "common.h"

namespace {

struct SeqArgs {
  const float* table;
  float* grad_table;
  int64_t vocab;
  const void* idx;
  int64_t sb, sl;
  int B, L, D, mode;
  int64_t sentinel, pad;
  float* out;
  const float* g_out;
  int64_t ostride;
  float scale;
  int* err;
};

template <int LPR, int LS, typename IdxT, bool BWD>
__global__ __launch_bounds__(RH_BLOCK) void seq_pool_kernel(const SeqArgs a) {
  RH_CHAIN_PRIO();
  constexpr int G = LPR * LS;
  constexpr int SPB = RH_BLOCK / G;
  constexpr int U = 8;
  const int lig = threadIdx.x % G;
  const int q = lig % LPR;
  const int ls = lig / LPR;
  int64_t b = (int64_t)blockIdx.x * SPB + threadIdx.x / G;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int L = a.L, D = a.D;
  const int nll = (L + LS - 1) / LS;
  const IdxT* ip = reinterpret_cast<const IdxT*>(a.idx) + b * a.sb;
  const bool pooled = a.mode != 2;
  bool oob_any = false;

  if (!BWD) {
    float4 acc = f4_zero();
    float cnt = 0.f;
    for (int j0 = 0; j0 < nll; j0 += U) {
      int64_t row[U];
      bool ok[U];
      int lc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int l = (j0 + u) * LS + ls;
        ok[u] = l < L;
        lc[u] = ok[u] ? l : L - 1;
        row[u] = (int64_t)gload<IdxT>(ip + lc[u] * a.sl);
      }
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool oob = (uint64_t)row[u] >= (uint64_t)a.vocab;
        oob_any |= (oob && ok[u]);
        v[u] = gload<float4>(a.table + (oob ? 0 : row[u]) * D + q * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pooled) {
          const bool use = ok[u] && row[u] != a.sentinel;
          acc = f4_add(acc, use ? v[u] : f4_zero());
          cnt += use ? 1.f : 0.f;
        } else if (ok[u] && live) {
          gstore<float4>(a.out + b * a.ostride + (int64_t)lc[u] * D + q * 4, v[u]);
        }
      }
    }
    if (pooled) {
#pragma unroll
      for (int m = LPR; m < G; m <<= 1) {
        acc = f4_add(acc, f4_shfl_xor(acc, m));
        cnt += __shfl_xor(cnt, m, RH_WAVE);
      }
      if (a.mode == 1) acc = f4_scale(acc, 1.f / (cnt + 1e-16f));
      if (live && ls == 0) gstore<float4>(a.out + b * a.ostride + q * 4, acc);
    }
  } else {
    float inv = 1.f;
    if (a.mode == 1) {
      float cnt = 0.f;
      for (int l = ls; l < L; l += LS) cnt += ((int64_t)gload<IdxT>(ip + l * a.sl) != a.sentinel) ? 1.f : 0.f;
#pragma unroll
      for (int m = LPR; m < G; m <<= 1) cnt += __shfl_xor(cnt, m, RH_WAVE);
      inv = 1.f / (cnt + 1e-16f);
    }
    float4 gp = f4_zero();
    if (pooled) gp = f4_scale(gload<float4>(a.g_out + b * a.ostride + q * 4), a.scale * inv);
    for (int j0 = 0; j0 < nll; j0 += U) {
      int64_t row[U];
      bool ok[U];
      float4 g[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int l = (j0 + u) * LS + ls;
        ok[u] = l < L;
        const int lc = ok[u] ? l : L - 1;
        row[u] = (int64_t)gload<IdxT>(ip + lc * a.sl);
        g[u] = pooled ? gp : f4_scale(gload<float4>(a.g_out + b * a.ostride + (int64_t)lc * D + q * 4), a.scale);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool oob = (uint64_t)row[u] >= (uint64_t)a.vocab;
        oob_any |= (oob && ok[u]);
        const bool masked = pooled && row[u] == a.sentinel;
        if (ok[u] && live && !oob && !masked && row[u] != a.pad) gatomic_add_f4(a.grad_table + row[u] * D + q * 4, g[u]);
      }
    }
  }
  if (oob_any && live && a.err != nullptr) atomicOr(a.err, RH_FLAG_INDEX_OOB);
}

template <int LPR, int LS, typename IdxT, bool BWD>
int launch(const SeqArgs& a, hipStream_t s) {
  constexpr int SPB = RH_BLOCK / (LPR * LS);
  const unsigned grid = (unsigned)((a.B + SPB - 1) / SPB);
  hipLaunchKernelGGL((seq_pool_kernel<LPR, LS, IdxT, BWD>), dim3(grid), dim3(RH_BLOCK), 0, s, a);
  return 0;
}

template <int LPR, typename IdxT, bool BWD>
int dispatch_ls(const SeqArgs& a, int ls, hipStream_t s) {
  if constexpr (LPR * 16 <= RH_WAVE) {
    if (ls >= 16) return launch<LPR, 16, IdxT, BWD>(a, s);
  }
  if constexpr (LPR * 8 <= RH_WAVE) {
    if (ls >= 8) return launch<LPR, 8, IdxT, BWD>(a, s);
  }
  if constexpr (LPR * 4 <= RH_WAVE) {
    if (ls >= 4) return launch<LPR, 4, IdxT, BWD>(a, s);
  }
  if constexpr (LPR * 2 <= RH_WAVE) {
    if (ls >= 2) return launch<LPR, 2, IdxT, BWD>(a, s);
  }
  return launch<LPR, 1, IdxT, BWD>(a, s);
}

template <typename IdxT, bool BWD>
int dispatch(const SeqArgs& a, hipStream_t s) {
  const int lpr = a.D / 4;
  int ls = 1;
  while (ls < 16 && lpr * ls * 2 <= RH_WAVE && ls * 8 < a.L) ls *= 2;
  switch (lpr) {
    case 1: return dispatch_ls<1, IdxT, BWD>(a, ls, s);
    case 2: return dispatch_ls<2, IdxT, BWD>(a, ls, s);
    case 4: return dispatch_ls<4, IdxT, BWD>(a, ls, s);
    case 8: return dispatch_ls<8, IdxT, BWD>(a, ls, s);
    case 16: return dispatch_ls<16, IdxT, BWD>(a, ls, s);
    case 32: return dispatch_ls<32, IdxT, BWD>(a, ls, s);
    default: return RH_E_UNSUPPORTED;
  }
}

int check(const char* who, const void* idx, int B, int L, int D, int mode, int64_t vocab) {
  RH_REQUIRE(idx != nullptr && B >= 0 && L > 0 && vocab > 0, RH_E_BADARG, "%s: bad arguments", who);
  RH_REQUIRE(mode >= 0 && mode <= 2, RH_E_BADARG, "%s: mode %d (0 sum, 1 mean, 2 concat)", who, mode);
  RH_REQUIRE(D > 0 && D % 4 == 0 && D <= 128 && ((D / 4) & (D / 4 - 1)) == 0, RH_E_UNSUPPORTED,
             "%s: embed_dim %d unsupported (need 4,8,16,32,64,128)", who, D);
  return 0;
}

}  // namespace

extern "C" int rh_seq_pool_fwd(const float* table, int64_t vocab, const void* idx, int idx_is_i64,
                               int64_t idx_stride_b, int64_t idx_stride_l, int B, int L, int D, int mode,
                               int64_t mask_sentinel, float* out, int64_t out_stride, int32_t* err_flag,
                               void* stream) {
  if (int rc = check("rh_seq_pool_fwd", idx, B, L, D, mode, vocab)) return rc;
  RH_REQUIRE(table && out, RH_E_BADARG, "rh_seq_pool_fwd: null pointer");
  if (B == 0) return 0;
  SeqArgs a{table, nullptr, vocab, idx, idx_stride_b, idx_stride_l, B, L, D, mode, mask_sentinel, -1, out, nullptr,
            out_stride, 1.f, err_flag};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = idx_is_i64 ? dispatch<int64_t, false>(a, s) : dispatch<int32_t, false>(a, s);
  if (rc != 0) return rc;
  RH_LAUNCH_CHECK("rh_seq_pool_fwd");
  return 0;
}

extern "C" int rh_seq_pool_bwd(float* grad_table, int64_t vocab, const void* idx, int idx_is_i64,
                               int64_t idx_stride_b, int64_t idx_stride_l, int B, int L, int D, int mode,
                               int64_t mask_sentinel, int64_t padding_idx, const float* g_out, int64_t g_stride,
                               float scale, int32_t* err_flag, void* stream) {
  if (int rc = check("rh_seq_pool_bwd", idx, B, L, D, mode, vocab)) return rc;
  RH_REQUIRE(grad_table && g_out, RH_E_BADARG, "rh_seq_pool_bwd: null pointer");
  if (B == 0) return 0;
  SeqArgs a{nullptr, grad_table, vocab, idx, idx_stride_b, idx_stride_l, B, L, D, mode, mask_sentinel, padding_idx,
            nullptr, g_out, g_stride, scale, err_flag};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = idx_is_i64 ? dispatch<int64_t, true>(a, s) : dispatch<int32_t, true>(a, s);
  if (rc != 0) return rc;
  RH_LAUNCH_CHECK("rh_seq_pool_bwd");
  return 0;
}
