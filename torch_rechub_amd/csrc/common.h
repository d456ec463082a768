// Internal helpers shared by the gfx950 kernels of librechub_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rechub_hip.h"

#define RH_WAVE 64
#define RH_BLOCK 256

void rh_set_error(const char* fmt, ...);
extern "C" int rh_optim_set_tuning(int key, int value);
extern "C" int rh_linear_set_tuning(int key, int value);
extern "C" int rh_din_set_tuning(int key, int value);

#define RH_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      rh_set_error(__VA_ARGS__);     \
      return (code);                 \
    }                                \
  } while (0)

// Returns from the enclosing int function with the hipError_t of the last launch, if any.
#define RH_LAUNCH_CHECK(name)                                               \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      rh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return (int)e_;                                                       \
    }                                                                       \
  } while (0)

// Wave priority of the step's latency-bound chain kernels (s_setprio 3 = highest): when the optimizer's VALU-saturating
// window sweep is resident on the same SIMD (deferred sweep on a side stream), the chain's waves win instruction issue.
#define RH_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

// Pointers fetched from descriptor tables are generic to the compiler; these helpers pin them to
// the global address space so the access is a global_load/global_store (vmcnt only), not flat_*.
#define RH_GLOBAL __attribute__((address_space(1)))
template <typename T>
static __device__ __forceinline__ T gload(const void* p) {
  return *reinterpret_cast<const RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(p));
}
template <typename T>
static __device__ __forceinline__ void gstore(void* p, T v) {
  *reinterpret_cast<RH_GLOBAL T*>(reinterpret_cast<uintptr_t>(p)) = v;
}
// 4-byte aligned on purpose: rows of a (B, 429) activation are only dword aligned; the amdhsa target
// runs in unaligned-access mode, so this is still one global_load_dwordx4 / global_store_dwordx4.
typedef float rh_v4f __attribute__((ext_vector_type(4), aligned(4)));
template <>
__device__ __forceinline__ float4 gload<float4>(const void* p) {
  const rh_v4f v = *reinterpret_cast<const RH_GLOBAL rh_v4f*>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
template <>
__device__ __forceinline__ void gstore<float4>(void* p, float4 v) {
  rh_v4f x = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<RH_GLOBAL rh_v4f*>(reinterpret_cast<uintptr_t>(p)) = x;
}

static __device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
static __device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
static __device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
static __device__ __forceinline__ float4 f4_scale(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
// a + s*b
static __device__ __forceinline__ float4 f4_fma(float s, float4 b, float4 a) {
  return make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w));
}
static __device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
static __device__ __forceinline__ float4 f4_shfl_xor(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m, RH_WAVE), __shfl_xor(v.y, m, RH_WAVE),
                     __shfl_xor(v.z, m, RH_WAVE), __shfl_xor(v.w, m, RH_WAVE));
}
// butterfly sum over all 64 lanes of the wavefront; every lane gets the total
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, RH_WAVE);
  return v;
}
// the same total as a wavefront-uniform value, without LDS traffic: four DPP steps inside each row of 16 lanes, one
// v_readlane per row (wave_sum above is six dependent ds_bpermute).  For the kernels whose per-row chain of reductions is
// what bounds them (Dice: up to five reductions per row).  EVERY lane of the wavefront must be active; summation order
// differs from wave_sum's butterfly (last-bit differences).
static __device__ __forceinline__ float wave_sum_dpp(float v) {
#define RH_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
  RH_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  RH_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  RH_DPP_ADD(0x141);  // row_half_mirror
  RH_DPP_ADD(0x140);  // row_mirror
#undef RH_DPP_ADD
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
// minimum over all 64 lanes as a wavefront-uniform (SGPR) value: DPP within each row of 16 lanes (no LDS traffic, unlike
// __shfl_xor = ds_bpermute), then one readlane per row
static __device__ __forceinline__ int wave_min_uniform(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return min(min(r0, r1), min(r2, r3));
}
static __device__ __forceinline__ int wave_max_uniform(int v) { return -wave_min_uniform(-v); }
// hardware fp32 atomic add on global memory (global_atomic_add_f32, result unused)
static __device__ __forceinline__ void gatomic_add_f32(float* p, float v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f32(
      reinterpret_cast<RH_GLOBAL float*>(reinterpret_cast<uintptr_t>(p)), v);
}
static __device__ __forceinline__ void gatomic_add_f4(float* p, float4 v) {
  gatomic_add_f32(p + 0, v.x);
  gatomic_add_f32(p + 1, v.y);
  gatomic_add_f32(p + 2, v.z);
  gatomic_add_f32(p + 3, v.w);
}
// LDS fp32 atomic add (ds_add_f32); p must point into __shared__ memory
#define RH_LDS_ATOMIC_ADD_F4(base, off, v)   \
  do {                                       \
    unsafeAtomicAdd(&(base)[(off) + 0], (v).x); \
    unsafeAtomicAdd(&(base)[(off) + 1], (v).y); \
    unsafeAtomicAdd(&(base)[(off) + 2], (v).z); \
    unsafeAtomicAdd(&(base)[(off) + 3], (v).w); \
  } while (0)

// Per-column batch statistics from per-slab (sum, M2) pairs -- what the STATS epilogue of csrc/gemm.hip writes: slab k holds
// rows [k * rows, (k + 1) * rows) of the M rows, stats[(k * 2 + {0, 1}) * K + c] = (sum, M2 about the slab mean) of column c.
// ONE pass over the slabs, about a shift = the mean of slab 0 (so nothing large cancels: the slab means differ from it by
// ~ std / sqrt(rows)):   d_k = mean_k - shift,  S1 = sum_k n_k d_k,  S2 = sum_k (M2_k + n_k d_k^2)
//                        mean = shift + S1 / M,  M2 = S2 - S1^2 / M           (the parallel-variance identity of Chan et al.)
// in a FIXED order (deterministic).  A workgroup of NT threads walks the K columns min(K, NT) at a time; with K < NT,
// G = NT / K thread groups split a column's slabs (k = g, g + G, ...) and are summed in group order through `red` (2 NT
// floats of LDS).  Loads go out in rounds of MAXS slabs per thread (2 MAXS registers; unconditional, clamped addresses: no
// branch per load).  Consumers of the statistics that run BESIDE the optimizer's resident sweep must stay small in
// registers -- a 256-register variant of this prologue (all 64 slabs of a column in flight at once) could not be placed on
// a SIMD until the sweep had finished (194 us for a 26 us kernel).  emit(c, mean, var) is called by ONE thread per column
// (var = biased).  All NT threads must call this (it synchronises the workgroup).
template <int NT, int MAXS, typename F>
static __device__ __forceinline__ void rh_combine_slabs(const float* __restrict__ stats, int nslab, int rows, int M, int K,
                                                        float* red, int tid, F emit) {
  const int kc = K < NT ? K : NT;       // columns per pass
  const int G = K < NT ? NT / K : 1;    // slab groups per column
  const int cl = tid % kc, g = tid / kc;
  const bool live = g < G;
  const float full = (float)rows, inv_full = 1.f / full;
  const float tail = (float)(M - (nslab - 1) * rows), inv_tail = 1.f / tail;
  for (int c0 = 0; c0 < K; c0 += kc) {  // uniform trip count
    const int c = c0 + cl;
    const bool ok = live && c < K;
    const int cc = c < K ? c : K - 1;
    const float shift = stats[cc] * (nslab == 1 ? inv_tail : inv_full);  // mean of slab 0
    float s1 = 0.f, s2 = 0.f;
    for (int base = 0; base < nslab; base += MAXS * G) {  // uniform trip count
      float ps[MAXS], pm[MAXS];
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const int k = base + g + i * G;
        const int kk = (ok && k < nslab) ? k : 0;
        ps[i] = stats[((int64_t)kk * 2 + 0) * K + cc];
        pm[i] = stats[((int64_t)kk * 2 + 1) * K + cc];
      }
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const int k = base + g + i * G;
        if (ok && k < nslab) {
          const bool last = k == nslab - 1;
          const float n = last ? tail : full;
          const float d = ps[i] * (last ? inv_tail : inv_full) - shift;
          s1 = fmaf(n, d, s1);
          s2 += fmaf(n * d, d, pm[i]);
        }
      }
    }
    __syncthreads();  // (the previous pass is done with red)
    if (live) {
      red[g * kc + cl] = s1;
      red[NT + g * kc + cl] = s2;
    }
    __syncthreads();
    if (ok && g == 0) {
      float t1 = red[cl], t2 = red[NT + cl];
      for (int q = 1; q < G; ++q) {
        t1 += red[q * kc + cl];
        t2 += red[NT + q * kc + cl];
      }
      const float dm = t1 / (float)M;
      emit(c, shift + dm, fmaxf((t2 - t1 * dm) / (float)M, 0.f));
    }
  }
  __syncthreads();
}

// Counter-based dropout hash (seed, call counter, element index) -> 32 random bits; shared by every kernel that applies or
// re-derives a dropout mask (csrc/mlp.hip, the head / GEMM prologues and epilogues of the fused MLP chain), so forward and
// backward -- fused or not -- see the same mask.  Round 4: the 64-bit splitmix form (three 64-bit multiplies = twelve
// quarter-rate v_mul_*_u32, ~250 cycles per element and wavefront) made the mask the most expensive part of the kernels
// that recompute it on an operand load; this one is the `lowbias32` integer finaliser (two 32-bit multiplies, full
// avalanche on sequential inputs) over index ^ key, the key (seed, counter) folded per call: ~80 cycles.
static __device__ __forceinline__ uint32_t rh_drop_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  const uint32_t k0 = (uint32_t)seed ^ ((uint32_t)ctr * 0x9E3779B1u) ^ ((uint32_t)(ctr >> 32) * 0xC2B2AE3Du);  // per call
  const uint32_t k1 = (uint32_t)(seed >> 32);
  uint32_t h = (uint32_t)idx ^ k0 ^ (((uint32_t)(idx >> 32) + k1) * 0x85EBCA77u);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
