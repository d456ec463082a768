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

// Counter-based dropout hash of csrc/mlp.hip (seed, call counter, element index) -> 32 random bits; shared with the head
// backward of csrc/linear.hip, which forms the BatchNorm-backward column sums of the layer below from the same mask.
static __device__ __forceinline__ uint32_t rh_drop_hash(uint64_t seed, uint64_t ctr, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed + ctr * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
