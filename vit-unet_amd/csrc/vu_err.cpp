#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "vu_common.h"

static thread_local char g_err[512] = "";

void vu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* vu_get_error() { return g_err; }

int vu_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vu_set_error("%s: %s", what, hipGetErrorString(e));
    return VU_ELAUNCH;
  }
  return VU_OK;
}
