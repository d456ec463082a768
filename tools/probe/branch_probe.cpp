// Graph-branch probe (no torch):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip branch_probe.cpp -o branch_probe
// Do the parallel branches of a captured hipGraph run concurrently on this runtime?  Two independent kernels of ~T us each
// (64 workgroups: a fraction of the chip), captured as fork / join across two streams; replay time ~T = concurrent, ~2T =
// serialised.  Run it under different runtime settings (e.g. DEBUG_CLR_GRAPH_PACKET_CAPTURE=0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (a == 7.f) out[0] = a;
}
int main() {
  float* d; CK(hipMalloc(&d, 64));
  hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipEvent_t fork, join, t0, t1; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int iters = 120000;
  hipGraph_t g; hipGraphExec_t ge;
  for (int branches = 1; branches <= 2; ++branches) {
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, d, 1000);
    if (branches == 2) {
      CK(hipEventRecord(fork, A)); CK(hipStreamWaitEvent(B, fork, 0));
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, d, iters);
    } else {
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, d, iters);
    }
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, d, iters);
    if (branches == 2) { CK(hipEventRecord(join, B)); CK(hipStreamWaitEvent(A, join, 0)); }
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, d, 1000);
    CK(hipStreamEndCapture(A, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, A));
    CK(hipStreamSynchronize(A));
    CK(hipEventRecord(t0, A));
    const int n = 50;
    for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, A));
    CK(hipEventRecord(t1, A)); CK(hipStreamSynchronize(A));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    printf("%s: %.1f us per replay\n", branches == 2 ? "two long kernels on two captured branches" : "two long kernels in one chain      ", 1e3 * ms / n);
  }
  return 0;
}
