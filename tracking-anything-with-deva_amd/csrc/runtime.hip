// Error channel + version of libdeva_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace deva {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace deva

extern "C" int deva_hip_version(void) { return DEVA_HIP_ABI_VERSION; }
extern "C" const char* deva_hip_last_error(void) { return deva::g_err; }
