// Error reporting and ABI version of libflowhigh_hip.so.
#include <stdarg.h>

#include "fh_common.h"

static thread_local char g_err[512] = "";

void fh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fh_last_error(void) { return g_err; }
extern "C" int fh_abi_version(void) { return FH_ABI_VERSION; }
