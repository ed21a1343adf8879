// Error plumbing + version for the C ABI.
#include <stdarg.h>

#include "common.h"

namespace ofa {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return OFA_ERR_LAUNCH;
  }
  return OFA_OK;
}
}  // namespace ofa

extern "C" int ofa_version(void) { return 100; }
extern "C" const char* ofa_last_error(void) { return ofa::g_err; }
