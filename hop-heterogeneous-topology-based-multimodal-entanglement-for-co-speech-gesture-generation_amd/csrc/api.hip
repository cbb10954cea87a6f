// libhopmi: version / error reporting shared by all entry points.
#include "common.h"

namespace hopmi {
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
    return HOPMI_ELAUNCH;
  }
  return HOPMI_OK;
}
}  // namespace hopmi

extern "C" const char* hopmi_version(void) { return "hopmi 0.1 (gfx950)"; }
extern "C" const char* hopmi_last_error(void) { return hopmi::g_err; }
