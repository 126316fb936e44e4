// core.hip — version / thread-local error string of libsy11.
#include "common.h"

static thread_local char g_err[512] = "";

void sy11_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sy11_version(void) { return SY11_VERSION; }
extern "C" const char* sy11_last_error(void) { return g_err; }
