// capi.cpp — host-side plumbing of the C ABI (error string, version).
#include <cstdarg>
#include <cstdio>
#include "../../include/lvdhip.h"

static thread_local char g_err[512] = "";

void lvd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lvdhip_last_error(void) { return g_err; }
extern "C" int lvdhip_version(void) { return 101; }  // 101: lvd_gemm_params.ldrowbias, lvd_ca_dq_params.acc32 / ldacc / acc_mode
