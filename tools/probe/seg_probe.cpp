// Segment-boundary probe (no torch):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip seg_probe.cpp -o seg_probe
// What does a boundary between two hipGraph launches cost on this runtime, and is a short EAGER segment in front of a graph
// cheaper than a short graph?  One "step" = 2 head kernels + 20 body kernels of ~6 us each (one workgroup per CU), replayed
// back to back on one stream:
//   V1  one graph of 22 kernels                      (the in-line form of the step)
//   V2  graph(2) + graph(20)                          (the deferred form: the sweep is forked between the two)
//   V3  2 eager launches + graph(20)
//   V4  22 eager launches
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (a == 7.f) out[0] = a;
}
static hipGraphExec_t capture(hipStream_t s, float* d, int n, int iters) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, iters);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  return ge;
}
int main(int argc, char** argv) {
  float* d; CK(hipMalloc(&d, 64));
  hipStream_t A; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int iters = 250, n = 400;
  hipGraphExec_t g22 = capture(A, d, 22, iters), g2 = capture(A, d, 2, iters), g20 = capture(A, d, 20, iters);
  for (int v = 0; v <= 4; ++v) {
    for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
      CK(hipStreamSynchronize(A));
      CK(hipEventRecord(t0, A));
      for (int i = 0; i < n; ++i) {
        if (v == 0) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, d, iters); continue; }
        if (v == 1) CK(hipGraphLaunch(g22, A));
        if (v == 2) { CK(hipGraphLaunch(g2, A)); CK(hipGraphLaunch(g20, A)); }
        if (v == 3) { for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, d, iters); CK(hipGraphLaunch(g20, A)); }
        if (v == 4) for (int k = 0; k < 22; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, d, iters);
      }
      CK(hipEventRecord(t1, A)); CK(hipStreamSynchronize(A));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) {
        const char* name[] = {"one eager kernel           ", "V1 graph(22)               ", "V2 graph(2) + graph(20)    ", "V3 2 eager + graph(20)     ",
                              "V4 22 eager                "};
        printf("%s %.1f us per step\n", name[v], 1e3 * ms / n);
      }
    }
  }
  return 0;
}
