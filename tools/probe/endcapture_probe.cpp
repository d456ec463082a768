// Minimal reproducer of the hipStreamEndCapture segfault that tools/bitwise_probe.py dp hit in round 5.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/endcapture_probe.cpp -o tools/probe/endcapture_probe
//   endcapture_probe 0   same roles twice:  capture(origin A, fork B), capture(origin A, fork B)      -> fine
//   endcapture_probe 1   roles swapped:     capture(origin A, fork B), capture(origin B, fork A)      -> ?
//   endcapture_probe 2   fork reused as a fork of ANOTHER origin: capture(A, B), capture(C, B)        -> ?
//   endcapture_probe 3   origin reused as a fork of another origin: capture(A, B), capture(C, A)      -> ?
//   endcapture_probe 4   ONE capture: A -> B -> C forked in a chain, C joined back into B (a forked stream waits for an event
//                        of the stream forked from it), B into A
//   endcapture_probe 5   the same chain with C and B both joined back into the origin A
// The backtrace of the Python crash (rocgdb, round 6) is an unbounded recursion hip::Stream::EndCapture() ->
// hip::Stream::EndCapture() -> ...: the runtime ends the capture of every stream in the origin's list of streams that joined
// it, recursively; a stale link left on a stream by an EARLIER capture closes a cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

__global__ void bump(int* p) { atomicAdd(p, 1); }

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1;
  hipStream_t A, B, C;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  int* d;
  CK(hipMalloc(&d, sizeof(int)));
  CK(hipMemset(d, 0, sizeof(int)));
  auto capture = [&](hipStream_t origin, hipStream_t fork, const char* what) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, origin, d);
    CK(hipEventRecord(e1, origin));
    CK(hipStreamWaitEvent(fork, e1, 0));  // fork joins the capture
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, fork, d);
    CK(hipEventRecord(e2, fork));
    CK(hipStreamWaitEvent(origin, e2, 0));  // joined back
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, origin, d);
    printf("%s: ending capture ...\n", what);
    fflush(stdout);
    CK(hipStreamEndCapture(origin, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, origin));
    CK(hipStreamSynchronize(origin));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    printf("%s: ok\n", what);
    fflush(stdout);
  };
  if (mode == 4 || mode == 5) {
    // ONE capture, origin A, two forked streams that wait on EACH OTHER's events: B joins through an event of A, C through an
    // event of B, then B waits for an event recorded on C (the join of C's work into B) before A waits for B.
    // mode 5: the same dependencies with C joined back into the ORIGIN instead (A waits for C, then for B).
    hipEvent_t e3, e4;
    CK(hipEventCreateWithFlags(&e3, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e4, hipEventDisableTiming));
    hipGraph_t g;
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, A, d);
    CK(hipEventRecord(e1, A));
    CK(hipStreamWaitEvent(B, e1, 0));  // B joins (parent A)
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, B, d);
    CK(hipEventRecord(e2, B));
    CK(hipStreamWaitEvent(C, e2, 0));  // C joins (parent B)
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, C, d);
    CK(hipEventRecord(e3, C));
    if (mode == 4) {
      CK(hipStreamWaitEvent(B, e3, 0));  // B, a forked stream, waits for an event of the stream forked from it
      hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, B, d);
      CK(hipEventRecord(e4, B));
      CK(hipStreamWaitEvent(A, e4, 0));
    } else {
      CK(hipStreamWaitEvent(A, e3, 0));
      CK(hipEventRecord(e4, B));
      CK(hipStreamWaitEvent(A, e4, 0));
    }
    printf("mode %d: ending capture ...\n", mode);
    fflush(stdout);
    CK(hipStreamEndCapture(A, &g));
    printf("mode %d: capture ended\n", mode);
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, A));
    CK(hipStreamSynchronize(A));
    int h = 0;
    CK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
    printf("mode %d done, counter %d (expected %d)\n", mode, h, mode == 4 ? 4 : 3);
    return 0;
  }
  capture(A, B, "capture 1 (origin A, fork B)");
  switch (mode) {
    case 0: capture(A, B, "capture 2 (origin A, fork B)"); break;
    case 1: capture(B, A, "capture 2 (origin B, fork A)"); break;
    case 2: capture(C, B, "capture 2 (origin C, fork B)"); break;
    case 3: capture(C, A, "capture 2 (origin C, fork A)"); break;
  }
  int h = 0;
  CK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
  printf("mode %d done, counter %d (expected 6)\n", mode, h);
  return 0;
}
