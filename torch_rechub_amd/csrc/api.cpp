// Error reporting and ABI version of librechub_hip.so (host-only translation unit).
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
