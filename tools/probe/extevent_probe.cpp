// External-event probe (no torch):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip extevent_probe.cpp -o extevent_probe
// Can ONE captured hipGraph synchronise, on every replay, with eager work on another stream?
//   graph (stream A):  k1 -> RECORD_EXT(e_mid) -> k2 ... ; at its head: WAIT_EXT(e_side)
//   host per step:     hipGraphLaunch(A); hipStreamWaitEvent(B, e_mid); side kernel on B; hipEventRecord(e_side, B)
// The kernels check ordering through a device counter protocol and the loop is timed against the same work without
// any cross-stream dependency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

// state[0] = steps completed by the main chain (k2), state[1] = steps completed by the side kernel, state[2] = errors
__global__ void k1(volatile int64_t* state, int spin) {
  // head of step s (s = state[0]): the side kernel of step s-2 ... must have finished: state[1] >= s - 1
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t s = state[0];
    if (state[1] < s - 1) atomicAdd((unsigned long long*)&state[2], 1ull);
    if (a == 7.f) state[3] = 1;
  }
}
__global__ void k2(volatile int64_t* state, int spin) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    state[0] = state[0] + 1;
    if (a == 7.f) state[3] = 1;
  }
}
// side kernel of step s (by value): k1 of step s must have run (state[4] = last k1 step marker), and side(s-1) done
__global__ void kmark(volatile int64_t* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[4] = state[0];  // step index whose k1 has completed
}
__global__ void kside(volatile int64_t* state, int64_t s, int spin) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0001f, 0.5f);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (state[4] < s) atomicAdd((unsigned long long*)&state[2], 1ull);        // started before step s's record point
    if (state[1] != s) atomicAdd((unsigned long long*)&state[2], 1000ull);    // side kernels out of order
    state[1] = s + 1;
    if (a == 7.f) state[3] = 1;
  }
}

int main() {
  int64_t* state;
  CK(hipMalloc(&state, 64));
  CK(hipMemset(state, 0, 64));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipEvent_t e_mid, e_side;
  CK(hipEventCreateWithFlags(&e_mid, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e_side, hipEventDisableTiming));
  CK(hipEventRecord(e_side, B));  // so that the first wait has something to wait for
  const int spin_main = 20000, spin_side = 60000;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
  CK(hipStreamWaitEvent(A, e_side, hipEventWaitExternal));
  hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, A, state, spin_main);
  hipLaunchKernelGGL(kmark, dim3(1), dim3(64), 0, A, state);
  CK(hipEventRecordWithFlags(e_mid, A, hipEventRecordExternal));
  hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, A, state, spin_main);
  CK(hipStreamEndCapture(A, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  printf("graph nodes: %zu\n", nn);
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const int steps = 200;
  CK(hipEventRecord(t0, A));
  for (int s = 0; s < steps; ++s) {
    CK(hipGraphLaunch(ge, A));
    CK(hipStreamWaitEvent(B, e_mid, 0));
    hipLaunchKernelGGL(kside, dim3(64), dim3(256), 0, B, state, (int64_t)s, spin_side);
    CK(hipEventRecord(e_side, B));
  }
  CK(hipEventRecord(t1, A));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, t0, t1));
  int64_t h[8];
  CK(hipMemcpy(h, state, 64, hipMemcpyDeviceToHost));
  printf("cross-stream loop: %.1f us/step  main steps %ld side steps %ld  ordering errors %ld\n", 1e3 * ms / steps,
         (long)h[0], (long)h[1], (long)h[2]);
  // reference timings: main graph alone, side alone
  CK(hipMemset(state, 0, 64));
  hipGraph_t g2;
  hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, A, state, spin_main);
  hipLaunchKernelGGL(kmark, dim3(1), dim3(64), 0, A, state);
  hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, A, state, spin_main);
  CK(hipStreamEndCapture(A, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  CK(hipEventRecord(t0, A));
  for (int s = 0; s < steps; ++s) CK(hipGraphLaunch(ge2, A));
  CK(hipEventRecord(t1, A));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, t0, t1));
  printf("main graph alone: %.1f us/step\n", 1e3 * ms / steps);
  CK(hipEventRecord(t0, B));
  for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(kside, dim3(64), dim3(256), 0, B, state, (int64_t)s, spin_side);
  CK(hipEventRecord(t1, B));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, t0, t1));
  printf("side kernel alone: %.1f us/step\n", 1e3 * ms / steps);
  return 0;
}
