// Error reporting and ABI version of librechub_hip.so (host-only translation unit).
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "rechub_hip.h"

static thread_local char g_err[512] = "";

void rh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int rh_abi_version(void) { return RH_ABI_VERSION; }
extern "C" const char* rh_last_error(void) { return g_err; }

#define RH_HIP_OK(call, what)                                                \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      rh_set_error("%s: %s", what, hipGetErrorString(e_));                   \
      return (int)e_;                                                        \
    }                                                                        \
  } while (0)

// MI355X: 8 XCDs x 32 CUs; mask bit i addresses CU i / 8 of XCD i % 8 (tools/probe/cumask_probe.cpp).
extern "C" int rh_stream_create_cumask(int cus_per_xcd, int from_top, void** out) {
  if (!out || cus_per_xcd < 1 || cus_per_xcd > 32) {
    rh_set_error("rh_stream_create_cumask: cus_per_xcd=%d outside 1..32", cus_per_xcd);
    return RH_E_BADARG;
  }
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int cu = 0; cu < 32; ++cu) {
    const bool on = from_top ? cu >= 32 - cus_per_xcd : cu < cus_per_xcd;
    if (on) mask[cu / 4] |= 0xffu << (8 * (cu % 4));  // the 8 bits cu * 8 .. cu * 8 + 7 = this CU of every XCD
  }
  hipStream_t st = nullptr;
  RH_HIP_OK(hipExtStreamCreateWithCUMask(&st, 8, mask), "hipExtStreamCreateWithCUMask");
  *out = st;
  return 0;
}
extern "C" int rh_stream_create_priority(int priority, void** out) {
  if (!out) return RH_E_BADARG;
  hipStream_t st = nullptr;
  RH_HIP_OK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority), "hipStreamCreateWithPriority");
  *out = st;
  return 0;
}
extern "C" int rh_stream_destroy(void* stream) {
  RH_HIP_OK(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)), "hipStreamDestroy");
  return 0;
}
extern "C" int rh_event_create(void** out) {
  if (!out) return RH_E_BADARG;
  hipEvent_t e = nullptr;
  RH_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
  *out = e;
  return 0;
}
extern "C" int rh_event_destroy(void* event) {
  RH_HIP_OK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)), "hipEventDestroy");
  return 0;
}
extern "C" int rh_event_record(void* event, void* stream, int external) {
  RH_HIP_OK(hipEventRecordWithFlags(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream),
                                    external ? hipEventRecordExternal : hipEventRecordDefault),
            "hipEventRecordWithFlags");
  return 0;
}
extern "C" int rh_stream_wait_event(void* stream, void* event, int external) {
  RH_HIP_OK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event),
                               external ? hipEventWaitExternal : hipEventWaitDefault),
            "hipStreamWaitEvent");
  return 0;
}
