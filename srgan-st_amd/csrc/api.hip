// Error channel + version of libsrganst.so (SURVEY.md 8b "Error convention").
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

int sst_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

SST_API const char* sst_last_error(void) { return g_err; }
SST_API int sst_version(void) { return 100; }  // 0.1.0
SST_API const char* sst_arch(void) { return "gfx950"; }
