// Standalone timing of csrc/gemm.hip (no torch):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch_rechub_amd/csrc [-DRH_PROBE=1|2] -x hip gemm_probe.cpp -o gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../torch-rechub_amd/csrc/gemm.hip"
void rh_set_error(const char*, ...) {}
int main() {
  const int M = 4096, N = 256, K = 429;
  float *x, *w, *b, *y, *st;
  hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4);
  hipMalloc(&y, (size_t)M * 512 * 4); hipMalloc(&st, (size_t)128 * 2 * 512 * 4);
  hipMemset(x, 0, (size_t)M * K * 4); hipMemset(w, 0, (size_t)N * K * 4); hipMemset(b, 0, N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto fn) {
    for (int i = 0; i < 5; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us\n", name, ms / 200 * 1e3);
  };
  time("fwd 4096x256x32 (1 tile)", [&] { rh_linear_fwd(x, 32, w, 32, b, M, N, 32, y, N, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x256x128 (4 tiles)", [&] { rh_linear_fwd(x, 128, w, 128, b, M, N, 128, y, N, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x256x256 (8 tiles)", [&] { rh_linear_fwd(x, 256, w, 256, b, M, N, 256, y, N, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x256x416 (13 tiles, aligned)", [&] { rh_linear_fwd(x, 416, w, 416, b, M, N, 416, y, N, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x256x429 (+stats)", [&] { rh_linear_fwd(x, K, w, K, b, M, N, K, y, N, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x256x429", [&] { rh_linear_fwd(x, K, w, K, b, M, N, K, y, N, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("fwd 4096x128x256", [&] { rh_linear_fwd(x, 256, w, 256, b, M, 128, 256, y, 128, nullptr, nullptr, nullptr, nullptr, nullptr); });
  time("dgrad 4096x256 -> 429", [&] { rh_linear_dgrad(x, 256, w, 429, M, 256, 429, y, 429, nullptr); });
  time("dgrad 4096x128 -> 256", [&] { rh_linear_dgrad(x, 128, w, 256, M, 128, 256, y, 256, nullptr); });
  return 0;
}
