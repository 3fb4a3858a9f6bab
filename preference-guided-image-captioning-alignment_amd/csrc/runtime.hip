// Host-side runtime of the C ABI: version, thread-local last error, launch check.
#include <stdarg.h>

#include "common.h"

namespace pgca {
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
    return PGCA_ERR_LAUNCH;
  }
  return PGCA_OK;
}
}  // namespace pgca

extern "C" int pgca_version(void) { return PGCA_ABI_VERSION; }
extern "C" int pgca_sizeof_gemm_args(void) { return (int)sizeof(pgca_gemm_args); }
extern "C" int pgca_sizeof_skinny_args(void) { return (int)sizeof(pgca_skinny_args); }
extern "C" const char* pgca_last_error(void) { return pgca::g_err; }
