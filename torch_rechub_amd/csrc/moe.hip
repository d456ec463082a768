// DCN-v2 mixture of low-rank experts (CrossNetMix) as TWO dense products per layer + three fused passes.
//
// Reference: CrossNetMix.forward torch_rechub/basic/layers.py:470-506 -- per layer l, expert e = 1..E (rank r):
//     g_e = gating_e(x_l);  v = tanh(V_e^T x_l);  v = tanh(C_e v);  o_e = x0 * (U_e v + bias_l)
//     x_{l+1} = sum_e softmax(g)_e o_e + x_l
// i.e. 4 E matmuls on (B, d, 1) column vectors + ~10 E elementwise kernels per layer in the reference.
//
// Restated with KP = E r + E rounded up to a multiple of 4 (E = 4, r = 32: 132):
//     PG (B, KP)  = x_l VgT^T          VgT (KP, d): rows e r + k = V_e[:, k], rows E r + e = gating_e.weight, rest 0
//     mid pass    : v1 = tanh(PG[:, :E r]);  v2_e = tanh(C_e v1_e);  gate = softmax(PG[:, E r : E r + E]);
//                   wp (B, KP) = [gate_e v2_e | sum_e gate_e | 0 ...]
//     Y (B, d)    = wp UTb^T           UTb (d, KP): columns e r + k = U_e[:, k], column E r = bias_l, rest 0
//     x_{l+1}     = x0 * Y + x_l       ( = x0 * (sum_e gate_e U_e v2_e + bias_l sum_e gate_e) + x_l )
// The sum over the experts is the K dimension of the second product, the bias rides along as one more K column, and
// both products are plain (B, 429) x (429, 132) GEMMs (library).  What is in this file: packing the parameters into
// VgT / UTb (one launch for all layers), the mid pass forward / backward (the r x r products per sample and expert
// run on LDS-resident C_e; the backward also accumulates g_C in registers), the residual backward
// (g_Y = g * x0, g_x0 += g * Y), and unpacking the weight-gradient slabs of the two products (rh_linear_wgrad_partial)
// into g_U, g_V, g_C, g_bias, g_gating (one launch for all layers, the slabs summed in split order on the way).
// Roofline: launch latency -- a layer moves ~20 MB at B = 4096; the point is 4 launches per layer instead of ~25.
#include "common.h"

namespace {

constexpr int kMaxLayers = 8;
constexpr int kMaxExperts = 16;

struct MoePackArgs {
  const float* U[kMaxLayers];     // (E, d, r)
  const float* V[kMaxLayers];     // (E, d, r)
  const float* bias[kMaxLayers];  // (d,)
  const float* Wg[kMaxExperts];   // (d,)
  float* VgT;                     // (L, KP, d)
  float* UTb;                     // (L, d, KP)
  int L, E, d, r, KP;
};

__global__ __launch_bounds__(RH_BLOCK) void moe_pack_kernel(const MoePackArgs a) {
  RH_CHAIN_PRIO();
  const int64_t per = (int64_t)a.KP * a.d;
  const int64_t total = 2 * a.L * per;
  const int ER = a.E * a.r;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * RH_BLOCK) {
    const int which = (int)(i / (a.L * per));  // 0: VgT, 1: UTb
    const int64_t o = i % (a.L * per);
    const int l = (int)(o / per);
    const int64_t w = o % per;
    if (which == 0) {
      const int row = (int)(w / a.d), j = (int)(w % a.d);
      float v = 0.f;
      if (row < ER) v = a.V[l][((int64_t)(row / a.r) * a.d + j) * a.r + row % a.r];
      else if (row < ER + a.E) v = a.Wg[row - ER][j];
      a.VgT[o] = v;
    } else {
      const int j = (int)(w / a.KP), c = (int)(w % a.KP);
      float v = 0.f;
      if (c < ER) v = a.U[l][((int64_t)(c / a.r) * a.d + j) * a.r + c % a.r];
      else if (c == ER) v = a.bias[l][j];
      a.UTb[o] = v;
    }
  }
}

struct MoeMidArgs {
  const float* PG;    // (B, KP)            fwd
  const float* C;     // (E, r, r)
  const float* g_wp;  // (B, KP)            bwd
  float* v1;          // (B, E r)           fwd: out, bwd: in
  float* v2;          // (B, E r)
  float* gate;        // (B, E)
  float* wp;          // (B, KP)            fwd out
  float* g_PG;        // (B, KP)            bwd out
  float* gC_partial;  // (gridDim.x, E r r) bwd out
  int B, E, KP;
};

// One thread per (sample slot s, expert e, rank index k); SPB = 256 / (E R) samples per pass of a workgroup.
template <int R>
__global__ __launch_bounds__(RH_BLOCK) void moe_mid_fwd_kernel(const MoeMidArgs a) {
  RH_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int E = a.E, ER = E * R, KP = a.KP;
  const int SPB = RH_BLOCK / ER;
  float* Cs = lds;                       // [E][R][R + 1]
  float* v1s = lds + (E * R * (R + 1) + 3) / 4 * 4;    // [SPB][ER], 16-byte aligned
  for (int i = threadIdx.x; i < E * R * R; i += RH_BLOCK) Cs[(i / R) * (R + 1) + i % R] = a.C[i];
  const int s = threadIdx.x / ER, c = threadIdx.x % ER, e = c / R, k = c % R;
  const bool slot = s < SPB;
  const int64_t groups = ((int64_t)a.B + SPB - 1) / SPB;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t b = grp * SPB + s;
    const bool live = slot && b < a.B;
    float a1 = 0.f;
    if (live) a1 = tanhf(a.PG[b * KP + c]);
    __syncthreads();  // Cs staged (first pass) / the previous pass is done with v1s
    if (slot) v1s[s * ER + c] = a1;
    __syncthreads();
    if (live) {
      const float* crow = Cs + (e * R + k) * (R + 1);
      const float4* vin = reinterpret_cast<const float4*>(v1s + s * ER + e * R);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < R / 4; ++j) {
        const float4 v4 = vin[j];
        acc = fmaf(crow[4 * j + 0], v4.x, acc);
        acc = fmaf(crow[4 * j + 1], v4.y, acc);
        acc = fmaf(crow[4 * j + 2], v4.z, acc);
        acc = fmaf(crow[4 * j + 3], v4.w, acc);
      }
      const float a2 = tanhf(acc);
      // softmax over the E gating scores of the sample (every thread of the sample, redundantly: E <= 16)
      const float* lg = a.PG + b * KP + ER;
      float m = lg[0];
      for (int x = 1; x < E; ++x) m = fmaxf(m, lg[x]);
      float den = 0.f, mine = 0.f;
      for (int x = 0; x < E; ++x) {
        const float ex = expf(lg[x] - m);
        den += ex;
        if (x == e) mine = ex;
      }
      const float ge = mine / den;
      a.v1[b * ER + c] = a1;
      a.v2[b * ER + c] = a2;
      a.wp[b * KP + c] = ge * a2;
      if (c < E) {  // thread c of the sample: gate_c; the first one also the sum of the gates and the zero padding
        const float gc = expf(lg[c] - m) / den;
        a.gate[b * E + c] = gc;
      }
      if (c == 0) {
        float sg = 0.f;
        for (int x = 0; x < E; ++x) sg += expf(lg[x] - m) / den;
        a.wp[b * KP + ER] = sg;
        for (int x = ER + 1; x < KP; ++x) a.wp[b * KP + x] = 0.f;
      }
    }
  }
}

template <int R>
__global__ __launch_bounds__(RH_BLOCK) void moe_mid_bwd_kernel(const MoeMidArgs a) {
  RH_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int E = a.E, ER = E * R, KP = a.KP;
  const int SPB = RH_BLOCK / ER;
  float* Cs = lds;                      // [E][R][R + 1]
  float* v1s = Cs + (E * R * (R + 1) + 3) / 4 * 4;    // [SPB][ER], 16-byte aligned
  float* gcs = v1s + SPB * ER;          // [SPB][ER]   g_c = g_v2 * (1 - v2^2)
  float* ggs = gcs + SPB * ER;          // [SPB][E]    g_gate
  float* red = ggs + SPB * kMaxExperts; // [SPB][ER][R] (after the loop)
  for (int i = threadIdx.x; i < E * R * R; i += RH_BLOCK) Cs[(i / R) * (R + 1) + i % R] = a.C[i];
  const int s = threadIdx.x / ER, c = threadIdx.x % ER, e = c / R, k = c % R;
  const bool slot = s < SPB;
  float acc[R];  // g_C[e][k][:] over the samples of this slot
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j] = 0.f;
  const int64_t groups = ((int64_t)a.B + SPB - 1) / SPB;
  float n1 = 0.f, n2 = 0.f, nw = 0.f, ne = 0.f, ns = 0.f;  // the next pass's inputs, in flight during this pass
  auto fetch = [&](int64_t grp) {
    const int64_t b = grp * SPB + s;
    n1 = n2 = nw = ne = ns = 0.f;
    if (slot && grp < groups && b < a.B) {
      n1 = a.v1[b * ER + c];
      n2 = a.v2[b * ER + c];
      nw = a.g_wp[b * KP + c];
      ne = a.gate[b * E + e];
      ns = a.g_wp[b * KP + ER];
    }
  };
  fetch(blockIdx.x);
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t b = grp * SPB + s;
    const bool live = slot && b < a.B;
    const float a1 = n1, a2 = n2, gw = nw, ge = ne, g_sg = ns;
    fetch(grp + gridDim.x);
    const float gc = gw * ge * (1.f - a2 * a2);
    __syncthreads();  // Cs staged (first pass) / the previous pass is done with the staging arrays
    // g_gate_e = sum_k g_wp[e, k] v2[e, k] + g_(sum of the gates): the R threads of (sample, expert) are R consecutive
    // lanes of one wavefront (R | 64, groups aligned): xor-shuffle sum
    float gg = gw * a2;
#pragma unroll
    for (int m = 1; m < R; m <<= 1) gg += __shfl_xor(gg, m, RH_WAVE);
    if (slot) {
      v1s[s * ER + c] = a1;
      gcs[s * ER + c] = gc;
      if (k == 0) ggs[s * kMaxExperts + e] = gg + g_sg;
    }
    __syncthreads();
    // g_v1[e][k] = sum_k' C[e][k'][k] g_c[e][k']   (this thread's k is the COLUMN here)
    float gv1 = 0.f;
    if (slot) {
      const float4* gin = reinterpret_cast<const float4*>(gcs + s * ER + e * R);
      const float4* vin = reinterpret_cast<const float4*>(v1s + s * ER + e * R);
      const float* ccol = Cs + e * R * (R + 1) + k;
#pragma unroll
      for (int j = 0; j < R / 4; ++j) {
        const float4 g4 = gin[j], v4 = vin[j];
        gv1 = fmaf(ccol[(4 * j + 0) * (R + 1)], g4.x, gv1);
        gv1 = fmaf(ccol[(4 * j + 1) * (R + 1)], g4.y, gv1);
        gv1 = fmaf(ccol[(4 * j + 2) * (R + 1)], g4.z, gv1);
        gv1 = fmaf(ccol[(4 * j + 3) * (R + 1)], g4.w, gv1);
        acc[4 * j + 0] = fmaf(gc, v4.x, acc[4 * j + 0]);
        acc[4 * j + 1] = fmaf(gc, v4.y, acc[4 * j + 1]);
        acc[4 * j + 2] = fmaf(gc, v4.z, acc[4 * j + 2]);
        acc[4 * j + 3] = fmaf(gc, v4.w, acc[4 * j + 3]);
      }
    }
    __syncthreads();  // ggs complete
    if (live) {
      a.g_PG[b * KP + c] = gv1 * (1.f - a1 * a1);
      if (c < KP - ER) {  // the gating columns: softmax backward for expert c, zeros in the padding
        float out = 0.f;
        if (c < E) {
          float dot = 0.f;
          for (int x = 0; x < E; ++x) dot = fmaf(a.gate[b * E + x], ggs[s * kMaxExperts + x], dot);
          out = a.gate[b * E + c] * (ggs[s * kMaxExperts + c] - dot);
        }
        a.g_PG[b * KP + ER + c] = out;
      }
    }
  }
  // g_C partial of this workgroup: the sample slots summed in slot order
  __syncthreads();
  if (slot) {
#pragma unroll
    for (int j = 0; j < R; ++j) red[(s * ER + c) * R + j] = acc[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ER * R; i += RH_BLOCK) {
    float t = 0.f;
    for (int x = 0; x < SPB; ++x) t += red[x * ER * R + i];
    a.gC_partial[(int64_t)blockIdx.x * ER * R + i] = t;
  }
}

// residual backward of x_{l+1} = x0 * Y + x_l:  g_Y = g * x0,  acc (+)= g * Y   (acc: the running gradient of x0)
// mode bit 0: first layer of the backward (acc is written, not accumulated); bit 1: last one -- acc also takes g itself (the
// residual path's gradient), so that the caller's closing product accumulates into acc and IS the gradient of x (= x0 = x_0):
// no copy of g for an out-of-place addmm and no final add (round 6: three launches of the DCN-v2 step less)
__global__ __launch_bounds__(RH_BLOCK) void moe_res_bwd_kernel(const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ x0, int64_t ldx0,
                                                               const float* __restrict__ Y, int B, int d, int mode,
                                                               float* __restrict__ g_Y, float* __restrict__ acc) {
  const bool first = (mode & 1) != 0, fold = (mode & 2) != 0;
  RH_CHAIN_PRIO();
  const int64_t n = (int64_t)B * d;
  for (int64_t i = (int64_t)blockIdx.x * RH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * RH_BLOCK) {
    const int64_t b = i / d;
    const int j = (int)(i % d);
    const float gi = g[b * ldg + j];
    g_Y[i] = gi * x0[b * ldx0 + j];
    const float t = gi * Y[i];
    const float a = first ? t : acc[i] + t;
    acc[i] = fold ? a + gi : a;
  }
}

struct MoeUnpackArgs {
  const float* slabV[kMaxLayers];  // S1 slabs (KP, d): g_VgT = g_PG^T x_l
  const float* slabU[kMaxLayers];  // S2 slabs (d, KP): g_UTb = g_Y^T wp
  const float* gC[kMaxLayers];     // NB partials (E r r)
  int64_t strideV[kMaxLayers], strideU[kMaxLayers];
  int S1[kMaxLayers], S2[kMaxLayers], NB[kMaxLayers];
  float* g_U[kMaxLayers];
  float* g_V[kMaxLayers];
  float* g_bias[kMaxLayers];
  float* g_C[kMaxLayers];
  float* g_Wg[kMaxExperts];
  int L, E, d, r, KP;
};

static __device__ __forceinline__ float sum_parts(const float* p, int64_t stride, int n) {
  float v = 0.f;
  int x = 0;
  for (; x + 4 <= n; x += 4) {
    const float t0 = p[(x + 0) * stride], t1 = p[(x + 1) * stride], t2 = p[(x + 2) * stride], t3 = p[(x + 3) * stride];
    v = (((v + t0) + t1) + t2) + t3;
  }
  for (; x < n; ++x) v += p[x * stride];
  return v;
}

// workgroups [0, cblocks): g_C -- 64 outputs each, the (up to 256) per-workgroup partial rows of the mid backward split
// over the 4 wavefronts, 8 loads in flight, summed in wavefront order (one thread per output would walk them as one
// dependent chain).  The rest: one thread per output of g_U / g_V / g_bias / g_gating over the few wgrad slabs.
__global__ __launch_bounds__(RH_BLOCK) void moe_unpack_kernel(const MoeUnpackArgs a, int cblocks) {
  RH_CHAIN_PRIO();
  const int E = a.E, d = a.d, r = a.r, KP = a.KP, ER = E * r;
  const int64_t nU = (int64_t)E * d * r, nC = (int64_t)E * r * r;
  if ((int)blockIdx.x < cblocks) {
    __shared__ float part[RH_BLOCK];
    const int per = (int)((nC + RH_WAVE - 1) / RH_WAVE);  // workgroups per layer
    const int l = blockIdx.x / per;
    const int wave = threadIdx.x / RH_WAVE, lane = threadIdx.x % RH_WAVE;
    const int64_t o = (int64_t)(blockIdx.x % per) * RH_WAVE + lane;
    float v = 0.f;
    if (o < nC) {
      const float* p = a.gC[l] + o;
      int x = wave;
      for (; x + 28 < a.NB[l]; x += 32) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(int64_t)(x + 4 * k) * nC];
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
      }
      for (; x < a.NB[l]; x += 4) v += p[(int64_t)x * nC];
    }
    part[threadIdx.x] = v;
    __syncthreads();
    if (wave == 0 && o < nC)
      a.g_C[l][o] = ((part[lane] + part[RH_WAVE + lane]) + part[2 * RH_WAVE + lane]) + part[3 * RH_WAVE + lane];
    return;
  }
  const int64_t per_layer = 2 * nU + d;  // g_U, g_V, g_bias
  const int64_t total = a.L * per_layer + (int64_t)E * d;
  const int64_t nthreads = (int64_t)(gridDim.x - cblocks) * RH_BLOCK;
  for (int64_t i = (int64_t)(blockIdx.x - cblocks) * RH_BLOCK + threadIdx.x; i < total; i += nthreads) {
    if (i >= a.L * per_layer) {  // gating weights: shared by the layers, summed over them in layer order
      const int64_t o = i - a.L * per_layer;
      const int e = (int)(o / d), j = (int)(o % d);
      float v = 0.f;
      for (int l = 0; l < a.L; ++l) v += sum_parts(a.slabV[l] + (int64_t)(ER + e) * d + j, a.strideV[l], a.S1[l]);
      a.g_Wg[e][j] = v;
      continue;
    }
    const int l = (int)(i / per_layer);
    int64_t o = i % per_layer;
    if (o < nU) {  // g_U[e][j][k] = g_UTb[j][e r + k]
      const int e = (int)(o / ((int64_t)d * r)), j = (int)((o / r) % d), k = (int)(o % r);
      a.g_U[l][o] = sum_parts(a.slabU[l] + (int64_t)j * KP + e * r + k, a.strideU[l], a.S2[l]);
    } else if ((o -= nU) < nU) {  // g_V[e][j][k] = g_VgT[e r + k][j]; walked along j: the slab rows are read coalesced
      const int row = (int)(o / d), j = (int)(o % d);
      a.g_V[l][((int64_t)(row / r) * d + j) * r + row % r] =
          sum_parts(a.slabV[l] + (int64_t)row * d + j, a.strideV[l], a.S1[l]);
    } else {  // g_bias[j] = g_UTb[j][E r]
      o -= nU;
      a.g_bias[l][o] = sum_parts(a.slabU[l] + o * KP + ER, a.strideU[l], a.S2[l]);
    }
  }
}

// forward: one pass per workgroup while they last (a pass is two dependent round trips + two barriers: ~2.5 us, walking 16
// of them per workgroup was 42 us per layer); backward: every workgroup leaves a g_C partial (E r r floats), so fewer
// workgroups, each prefetching its next pass
int mid_grid(int B, int E, int r, bool bwd) {
  const int spb = RH_BLOCK / (E * r);
  int64_t g = ((int64_t)B + spb - 1) / spb;
  const int64_t cap = bwd ? 512 : 2048;
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

size_t mid_lds(int E, int r, bool bwd) {
  const int ER = E * r, spb = RH_BLOCK / ER;
  size_t n = ((size_t)E * r * (r + 1) + 3) / 4 * 4 + (size_t)spb * ER;
  if (bwd) n += (size_t)spb * ER + (size_t)spb * kMaxExperts + (size_t)spb * ER * r;
  return n * sizeof(float);
}

bool rank_ok(int r) { return r == 4 || r == 8 || r == 16 || r == 32 || r == 64; }

template <int R, bool BWD>
int launch_mid(const MoeMidArgs& a, hipStream_t s) {
  const size_t lds = mid_lds(a.E, R, BWD);
  auto* fn = BWD ? moe_mid_bwd_kernel<R> : moe_mid_fwd_kernel<R>;
  static size_t reserved = 48 * 1024;  // per instantiation: raise the dynamic-LDS limit once per size class
  if (lds > reserved) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    RH_REQUIRE(e == hipSuccess, (int)e, "rh_cross_moe_mid: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    reserved = lds;
  }
  hipLaunchKernelGGL(fn, dim3(mid_grid(a.B, a.E, R, BWD)), dim3(RH_BLOCK), lds, s, a);
  return 0;
}

template <bool BWD>
int dispatch_mid(const MoeMidArgs& a, int r, hipStream_t s) {
  switch (r) {
    case 4: return launch_mid<4, BWD>(a, s);
    case 8: return launch_mid<8, BWD>(a, s);
    case 16: return launch_mid<16, BWD>(a, s);
    case 32: return launch_mid<32, BWD>(a, s);
    case 64: return launch_mid<64, BWD>(a, s);
    default: return RH_E_UNSUPPORTED;
  }
}

int check_shape(const char* who, int E, int d, int r) {
  RH_REQUIRE(E >= 1 && E <= kMaxExperts && d >= 1 && rank_ok(r) && E * r <= RH_BLOCK, RH_E_UNSUPPORTED,
             "%s: E=%d r=%d unsupported (E <= %d, r in {4, 8, 16, 32, 64}, E r <= %d)", who, E, r, kMaxExperts, RH_BLOCK);
  RH_REQUIRE(mid_lds(E, r, true) <= 160 * 1024, RH_E_UNSUPPORTED, "%s: E=%d r=%d needs %zu B of LDS", who, E, r,
             mid_lds(E, r, true));
  return 0;
}

}  // namespace

extern "C" int rh_cross_moe_kp(int E, int r) { return ((E * r + E + 3) / 4) * 4; }

extern "C" int rh_cross_moe_supported(int L, int E, int d, int r) {
  return (L >= 1 && L <= kMaxLayers && E >= 1 && E <= kMaxExperts && d >= 1 && rank_ok(r) && E * r <= RH_BLOCK &&
          mid_lds(E, r, true) <= 160 * 1024) ? 1 : 0;
}

extern "C" int rh_cross_moe_mid_blocks(int B, int E, int r) { return mid_grid(B, E, r, true); }

extern "C" int rh_cross_moe_pack(const float* const* U, const float* const* V, const float* const* bias,
                                 const float* const* Wg, int L, int E, int d, int r, float* VgT, float* UTb, void* stream) {
  RH_REQUIRE(U && V && bias && Wg && VgT && UTb, RH_E_BADARG, "rh_cross_moe_pack: null pointer");
  RH_REQUIRE(L >= 1 && L <= kMaxLayers, RH_E_UNSUPPORTED, "rh_cross_moe_pack: %d layers (max %d)", L, kMaxLayers);
  if (int rc = check_shape("rh_cross_moe_pack", E, d, r)) return rc;
  MoePackArgs a{};
  for (int l = 0; l < L; ++l) a.U[l] = U[l], a.V[l] = V[l], a.bias[l] = bias[l];
  for (int e = 0; e < E; ++e) a.Wg[e] = Wg[e];
  a.VgT = VgT, a.UTb = UTb, a.L = L, a.E = E, a.d = d, a.r = r, a.KP = rh_cross_moe_kp(E, r);
  const int64_t total = 2 * (int64_t)L * a.KP * d;
  int64_t grid = (total + RH_BLOCK - 1) / RH_BLOCK;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(moe_pack_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a);
  RH_LAUNCH_CHECK("rh_cross_moe_pack");
  return 0;
}

extern "C" int rh_cross_moe_mid_fwd(const float* PG, const float* C, int B, int E, int r, float* v1, float* v2, float* gate,
                                    float* wp, void* stream) {
  RH_REQUIRE(PG && C && v1 && v2 && gate && wp, RH_E_BADARG, "rh_cross_moe_mid_fwd: null pointer");
  if (int rc = check_shape("rh_cross_moe_mid_fwd", E, 1, r)) return rc;
  if (B <= 0) return 0;
  MoeMidArgs a{PG, C, nullptr, v1, v2, gate, wp, nullptr, nullptr, B, E, rh_cross_moe_kp(E, r)};
  if (int rc = dispatch_mid<false>(a, r, reinterpret_cast<hipStream_t>(stream))) return rc;
  RH_LAUNCH_CHECK("rh_cross_moe_mid_fwd");
  return 0;
}

extern "C" int rh_cross_moe_mid_bwd(const float* g_wp, const float* v1, const float* v2, const float* gate, const float* C,
                                    int B, int E, int r, float* g_PG, float* gC_partial, void* stream) {
  RH_REQUIRE(g_wp && v1 && v2 && gate && C && g_PG && gC_partial, RH_E_BADARG, "rh_cross_moe_mid_bwd: null pointer");
  if (int rc = check_shape("rh_cross_moe_mid_bwd", E, 1, r)) return rc;
  RH_REQUIRE(B >= 1, RH_E_BADARG, "rh_cross_moe_mid_bwd: empty batch");
  MoeMidArgs a{nullptr, C, g_wp, const_cast<float*>(v1), const_cast<float*>(v2), const_cast<float*>(gate), nullptr, g_PG,
               gC_partial, B, E, rh_cross_moe_kp(E, r)};
  if (int rc = dispatch_mid<true>(a, r, reinterpret_cast<hipStream_t>(stream))) return rc;
  RH_LAUNCH_CHECK("rh_cross_moe_mid_bwd");
  return 0;
}

extern "C" int rh_cross_moe_res_bwd(const float* g, int64_t ldg, const float* x0, int64_t ldx0, const float* Y, int B, int d,
                                    int mode, float* g_Y, float* acc, void* stream) {
  RH_REQUIRE(g && x0 && Y && g_Y && acc, RH_E_BADARG, "rh_cross_moe_res_bwd: null pointer");
  RH_REQUIRE(B >= 0 && d >= 1 && ldg >= d && ldx0 >= d, RH_E_BADARG, "rh_cross_moe_res_bwd: bad shape");
  RH_REQUIRE(mode >= 0 && mode <= 3, RH_E_BADARG, "rh_cross_moe_res_bwd: mode %d (bit 0 first, bit 1 last)", mode);
  if (B == 0) return 0;
  int64_t grid = ((int64_t)B * d + RH_BLOCK - 1) / RH_BLOCK;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(moe_res_bwd_kernel, dim3((unsigned)grid), dim3(RH_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), g,
                     ldg, x0, ldx0, Y, B, d, mode, g_Y, acc);
  RH_LAUNCH_CHECK("rh_cross_moe_res_bwd");
  return 0;
}

extern "C" int rh_cross_moe_unpack(const float* const* slabV, const int* S1, const float* const* slabU, const int* S2,
                                   const float* const* gC, const int* NB, int L, int E, int d, int r, float* const* g_U,
                                   float* const* g_V, float* const* g_bias, float* const* g_C, float* const* g_Wg,
                                   void* stream) {
  RH_REQUIRE(slabV && S1 && slabU && S2 && gC && NB && g_U && g_V && g_bias && g_C && g_Wg, RH_E_BADARG,
             "rh_cross_moe_unpack: null pointer");
  RH_REQUIRE(L >= 1 && L <= kMaxLayers, RH_E_UNSUPPORTED, "rh_cross_moe_unpack: %d layers (max %d)", L, kMaxLayers);
  if (int rc = check_shape("rh_cross_moe_unpack", E, d, r)) return rc;
  MoeUnpackArgs a{};
  a.L = L, a.E = E, a.d = d, a.r = r, a.KP = rh_cross_moe_kp(E, r);
  for (int l = 0; l < L; ++l) {
    a.slabV[l] = slabV[l], a.slabU[l] = slabU[l], a.gC[l] = gC[l];
    a.S1[l] = S1[l], a.S2[l] = S2[l], a.NB[l] = NB[l];
    a.strideV[l] = (int64_t)a.KP * d, a.strideU[l] = (int64_t)d * a.KP;
    a.g_U[l] = g_U[l], a.g_V[l] = g_V[l], a.g_bias[l] = g_bias[l], a.g_C[l] = g_C[l];
  }
  for (int e = 0; e < E; ++e) a.g_Wg[e] = g_Wg[e];
  const int64_t total = (int64_t)L * (2 * (int64_t)E * d * r + d) + (int64_t)E * d;
  int64_t grid = (total + RH_BLOCK - 1) / RH_BLOCK;
  if (grid > 2048) grid = 2048;
  const int cblocks = L * (int)(((int64_t)E * r * r + RH_WAVE - 1) / RH_WAVE);
  hipLaunchKernelGGL(moe_unpack_kernel, dim3((unsigned)(grid + cblocks)), dim3(RH_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), a, cblocks);
  RH_LAUNCH_CHECK("rh_cross_moe_unpack");
  return 0;
}
