// DIN local activation unit, first layer: z = [t, h, t - h, t * h] W^T + b over the B * L history positions, with the
// (B * L, 4 D) operand built IN REGISTERS (never in HBM) and the BatchNorm statistics of z as an epilogue.
//
// Reference: ActivationUnit.forward torch_rechub/models/ranking/din.py:77-92
//     target = target.unsqueeze(1).expand(-1, L, -1)
//     att_input = torch.cat([target, history, target - history, target * history], dim=-1).view(-1, 4 * D)
//     att_weight = self.attention(att_input)          # MLP: Linear(4 D, N) -> BatchNorm1d -> Dice -> ... (layers.py:276-292)
// At configs[3] (B = 4096, L = 100, D = 16, N = 256) that operand is 105 MB per history field written, read back by the
// GEMM, and z (419 MB) is read once more for the BatchNorm statistics.
//
// Layout.  v_mfma_f32_32x32x2_f32 takes A as one float per lane: lane (i = lane % 32, kk = lane / 32) holds A[i][k] for
// the k of its half.  With the k-pair of MFMA step s chosen as {s, 2 D + s}, lane (i, 0) needs [t | h] of row i and lane
// (i, 1) needs [t - h | t * h]: both are built from ONE 64-byte load of the history row and the sample's target row,
// directly into the operand registers -- no LDS, no shuffles.  The weight fragment of a wavefront (its 64 output
// columns x its k half: 2 x 2 D registers) is loaded once and stays in registers for every row tile.
// A workgroup = N / 64 wavefronts on the same 32 rows (different columns; the history loads of the others hit L1) walking
// a chunk of rows; the epilogue adds the bias, stores the 32 x 64 tile (128-byte row segments) and folds the tile into the
// chunk's running (count, mean, M2) per column with Chan's formula -- the (sum, M2) pairs rh_bn_stats_from_partial
// (csrc/mlp.hip) combines, exactly what bn_partial_kernel<0> would have produced from a pass over z.
// Exact f32: the MFMA is a k-ordered fmaf chain (summation order over k permuted, same terms).
// Roofline: the store of z (N * 4 bytes per row: 419 MB at configs[3] -> ~100 us at 4 TB/s); MFMA time 85 us at peak.
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

struct AttL1Args {
  const float* hist;  // (B, L, D), rows contiguous inside a sample, sample stride hs
  int64_t hs;
  const float* tgt;   // (B, D), row stride ts
  int64_t ts;
  const float* W;     // (N, 4 D) row-major (nn.Linear.weight)
  const float* bias;  // (N,) or null
  float* z;           // (B * L, N)
  float* partial;     // (nchunks, 2, N) chunk sum / chunk M2, or null
  int B, L, N;
  int rows_per_chunk;  // multiple of 32
};

template <int D>
__global__ __launch_bounds__(RH_BLOCK) void din_att_l1_kernel(const AttL1Args a) {
  RH_CHAIN_PRIO();
  constexpr int H = 2 * D;  // floats of a lane's operand half = MFMA steps
  const int lane = threadIdx.x % RH_WAVE, wave = threadIdx.x / RH_WAVE;
  const int i = lane & 31, kk = lane >> 5;
  const int64_t R = (int64_t)a.B * a.L;
  const int N = a.N;
  const int col0 = wave * 64 + i;  // this lane's column in the wavefront's first 32-column tile (second: + 32)

  // weight fragment: lane (j, kk) of tile c holds W[wave * 64 + c * 32 + j][kk * H + s], s < H
  float w[2][H];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* wr = a.W + (int64_t)(col0 + c * 32) * (4 * D) + kk * H;
#pragma unroll
    for (int s = 0; s < H; s += 4) {
      const float4 v = gload<float4>(wr + s);
      w[c][s] = v.x, w[c][s + 1] = v.y, w[c][s + 2] = v.z, w[c][s + 3] = v.w;
    }
  }
  float bv[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) bv[c] = a.bias != nullptr ? a.bias[col0 + c * 32] : 0.f;

  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_chunk;
  const int64_t r1 = r0 + a.rows_per_chunk < R ? r0 + a.rows_per_chunk : R;
  // running statistics of the chunk, per column tile (both lanes of a column carry the same values)
  float cnt = 0.f, mean[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f};

  auto load_rows = [&](int64_t base, float* h, float* t) {
    int64_t r = base + i;
    if (r >= R) r = R - 1;  // clamped: the rows past the end are never stored nor counted
    const int64_t b = r / a.L;
    const float* hp = a.hist + b * a.hs + (r - b * a.L) * D;
    const float* tp = a.tgt + b * a.ts;
#pragma unroll
    for (int s = 0; s < D; s += 4) {
      const float4 hv = gload<float4>(hp + s), tv = gload<float4>(tp + s);
      h[s] = hv.x, h[s + 1] = hv.y, h[s + 2] = hv.z, h[s + 3] = hv.w;
      t[s] = tv.x, t[s + 1] = tv.y, t[s + 2] = tv.z, t[s + 3] = tv.w;
    }
  };

  float hn[D], tn[D];
  load_rows(r0, hn, tn);
  for (int64_t base = r0; base < r1; base += 32) {
    float av[H];
#pragma unroll
    for (int s = 0; s < D; ++s) {
      av[s] = kk == 0 ? tn[s] : tn[s] - hn[s];
      av[D + s] = kk == 0 ? hn[s] : tn[s] * hn[s];
    }
    if (base + 32 < r1) load_rows(base + 32, hn, tn);  // the next tile's rows are in flight under this tile's MFMAs
    v16f acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < H; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], w[0][s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], w[1][s], acc[1], 0, 0, 0);
    }
    // C / D map of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    const int64_t rbase = base + 4 * kk;
    const int nrows = (int)(r1 - base < 32 ? r1 - base : 32);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float s1 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[c][q] += bv[c];
        const int64_t row = rbase + (q & 3) + 8 * (q >> 2);
        if (row < r1) {
          a.z[row * N + col0 + c * 32] = acc[c][q];
          s1 += acc[c][q];
        }
      }
      if (a.partial != nullptr) {
        s1 += __shfl_xor(s1, 32, RH_WAVE);
        const float tm = s1 / (float)nrows;
        float t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float dlt = acc[c][q] - tm;
          if (rbase + (q & 3) + 8 * (q >> 2) < r1) t2 = fmaf(dlt, dlt, t2);
        }
        t2 += __shfl_xor(t2, 32, RH_WAVE);
        // Chan: merge (nrows, tm, t2) into the chunk's (cnt, mean, m2)
        const float tot = cnt + (float)nrows;
        const float dlt = tm - mean[c];
        m2[c] += t2 + dlt * dlt * (cnt * (float)nrows / tot);
        mean[c] += dlt * ((float)nrows / tot);
      }
    }
    cnt += (float)nrows;
  }
  if (a.partial != nullptr && kk == 0 && r0 < R) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      a.partial[((int64_t)blockIdx.x * 2 + 0) * N + col0 + c * 32] = mean[c] * cnt;
      a.partial[((int64_t)blockIdx.x * 2 + 1) * N + col0 + c * 32] = m2[c];
    }
  }
}

}  // namespace

extern "C" int rh_din_att_l1_supported(int D, int N) {
  return ((D == 4 || D == 8 || D == 16) && N >= 64 && N <= 256 && N % 64 == 0) ? 1 : 0;
}

// rows per statistics chunk (= per workgroup): ~3 workgroups per CU, a multiple of 32
extern "C" int rh_din_att_l1_chunk_rows(int64_t rows) {
  int64_t per = (rows + 767) / 768;
  per = (per + 31) / 32 * 32;
  return (int)(per < 32 ? 32 : per);
}

extern "C" int rh_din_att_l1_fwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride,
                                 const float* W, const float* bias, int B, int L, int D, int N, float* z, float* partial,
                                 void* stream) {
  RH_REQUIRE(hist && tgt && W && z, RH_E_BADARG, "rh_din_att_l1_fwd: null pointer");
  RH_REQUIRE(rh_din_att_l1_supported(D, N), RH_E_UNSUPPORTED,
             "rh_din_att_l1_fwd: D=%d N=%d unsupported (D in {4, 8, 16}, N a multiple of 64 up to 256)", D, N);
  RH_REQUIRE(B >= 0 && L >= 1 && hist_stride >= (int64_t)L * D && tgt_stride >= D, RH_E_BADARG,
             "rh_din_att_l1_fwd: bad shape B=%d L=%d", B, L);
  if (B == 0) return 0;
  AttL1Args a{hist, hist_stride, tgt, tgt_stride, W, bias, z, partial, B, L, N, 0};
  const int64_t R = (int64_t)B * L;
  a.rows_per_chunk = rh_din_att_l1_chunk_rows(R);
  const unsigned grid = (unsigned)((R + a.rows_per_chunk - 1) / a.rows_per_chunk);
  const dim3 block(RH_WAVE * (N / 64));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 4: hipLaunchKernelGGL((din_att_l1_kernel<4>), dim3(grid), block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((din_att_l1_kernel<8>), dim3(grid), block, 0, s, a); break;
    default: hipLaunchKernelGGL((din_att_l1_kernel<16>), dim3(grid), block, 0, s, a); break;
  }
  RH_LAUNCH_CHECK("rh_din_att_l1_fwd");
  return 0;
}
