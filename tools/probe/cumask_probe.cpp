// CU-mask probe (no torch):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip cumask_probe.cpp -o cumask_probe
// For several hipExtStreamCreateWithCUMask patterns: which (XCC, SE/SH/CU) slots the workgroups of a launch land on, and
// how a VALU-bound kernel's time scales -- i.e. how mask bits map onto the 8 XCDs x 32 CUs of an MI355X.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <set>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void where_kernel(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xffff) | (a == 7.f ? 1u << 31 : 0);
}

__global__ void valu_kernel(float* out, int iters) {
  float a = threadIdx.x, b = blockIdx.x, c = 1.f, d = 2.f;
  for (int i = 0; i < iters; ++i) {
    a = fmaf(a, 1.0001f, 0.5f);
    b = fmaf(b, 0.9999f, 0.25f);
    c = fmaf(c, 1.0002f, 0.125f);
    d = fmaf(d, 0.9998f, 0.0625f);
  }
  if (a + b + c + d == 12345.f) out[0] = a;
}

int main() {
  const int NW = 8;  // 256 bits
  struct Pat { const char* name; uint32_t m[NW]; };
  std::vector<Pat> pats;
  auto mk = [&](const char* name, auto fn) {
    Pat p;
    p.name = name;
    memset(p.m, 0, sizeof(p.m));
    for (int i = 0; i < 256; ++i)
      if (fn(i)) p.m[i / 32] |= 1u << (i % 32);
    pats.push_back(p);
  };
  mk("all256", [](int i) { return true; });
  mk("first128", [](int i) { return i < 128; });
  mk("first32", [](int i) { return i < 32; });
  mk("first8", [](int i) { return i < 8; });
  mk("even", [](int i) { return i % 2 == 0; });
  mk("mod8<6 (192)", [](int i) { return i % 8 < 6; });
  mk("mod8<5 (160)", [](int i) { return i % 8 < 5; });
  mk("mod8>=6 (64)", [](int i) { return i % 8 >= 6; });
  mk("first192", [](int i) { return i < 192; });
  mk("last64", [](int i) { return i >= 192; });
  uint32_t* d_out;
  float* d_f;
  CK(hipMalloc(&d_out, 1 << 20));
  CK(hipMalloc(&d_f, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto& p : pats) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, NW, p.m));
    const int nb = 8192;
    std::vector<uint32_t> h(nb);
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, st, d_out, 20000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d_out, nb * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> slots;
    int per_xcc[16] = {0};
    std::set<uint32_t> per_xcc_slots[16];
    for (auto v : h) {
      slots.insert(v & 0xfffff);
      per_xcc[(v >> 16) & 0xf]++;
      per_xcc_slots[(v >> 16) & 0xf].insert(v & 0xffff);
    }
    hipLaunchKernelGGL(valu_kernel, dim3(256 * 8), dim3(256), 0, st, d_f, 1000);  // warm
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(valu_kernel, dim3(256 * 16), dim3(256), 0, st, d_f, 100000);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s distinct (xcc,se,sh,cu) = %3zu  valu kernel %.3f ms   CUs per XCC:", p.name, slots.size(), ms);
    for (int x = 0; x < 8; ++x) printf(" %zu", per_xcc_slots[x].size());
    printf("\n");
    CK(hipStreamDestroy(st));
  }
  // two streams with complementary masks running concurrently: does each keep its own CUs?
  {
    Pat a = pats[5], b = pats[7];  // 192 / 64
    hipStream_t sa, sb;
    CK(hipExtStreamCreateWithCUMask(&sa, NW, a.m));
    CK(hipExtStreamCreateWithCUMask(&sb, NW, b.m));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a0, sa));
      hipLaunchKernelGGL(valu_kernel, dim3(256 * 16), dim3(256), 0, sa, d_f, 100000);
      CK(hipEventRecord(a1, sa));
      CK(hipEventRecord(b0, sb));
      hipLaunchKernelGGL(valu_kernel, dim3(256 * 16), dim3(256), 0, sb, d_f, 25000);
      CK(hipEventRecord(b1, sb));
      CK(hipDeviceSynchronize());
    }
    float ma, mb;
    CK(hipEventElapsedTime(&ma, a0, a1));
    CK(hipEventElapsedTime(&mb, b0, b1));
    printf("concurrent: 192-CU stream (full work) %.3f ms, 64-CU stream (1/4 work) %.3f ms\n", ma, mb);
  }
  return 0;
}
