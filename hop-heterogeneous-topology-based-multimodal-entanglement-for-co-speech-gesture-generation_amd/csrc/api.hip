// libhopmi: version / error reporting shared by all entry points.
#include "common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace hopmi {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Tuning knobs (HOPMI_WN_GRID, ...) are read from the environment ONCE per process and kept in a small table: a getenv
// per launch showed up in host profiles of the training step.  hopmi_reload_env() forgets the table (probes that sweep
// a knob inside one process call it after changing the variable).
namespace {
struct EnvSlot { const char* name; int value; };
constexpr int kEnvSlots = 32;
EnvSlot g_env[kEnvSlots];
std::atomic<int> g_env_n{0};
std::mutex g_env_mu;
}  // namespace

int env_int(const char* name, int dflt) {
  const int n = g_env_n.load(std::memory_order_acquire);
  for (int i = 0; i < n; ++i)
    if (g_env[i].name == name || strcmp(g_env[i].name, name) == 0) return g_env[i].value < 0 ? dflt : g_env[i].value;
  std::lock_guard<std::mutex> lk(g_env_mu);
  const char* e = getenv(name);
  const int v = (e && *e) ? atoi(e) : -1;           // every knob is a non-negative integer; -1 = unset
  const int m = g_env_n.load(std::memory_order_relaxed);
  if (m < kEnvSlots) {
    g_env[m] = EnvSlot{name, v};
    g_env_n.store(m + 1, std::memory_order_release);
  }
  return v < 0 ? dflt : v;
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
extern "C" void hopmi_reload_env(void) {
  std::lock_guard<std::mutex> lk(hopmi::g_env_mu);
  hopmi::g_env_n.store(0, std::memory_order_release);
}

// ---- stream-capture hygiene (hopmi/graph.py): look before ending a capture that an exception interrupted
extern "C" int hopmi_stream_capture_status(void* stream, int* status) {
  if (!status) return HOPMI_EINVAL;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    // a query on an invalidated capture may itself be answered with the invalidation error: that IS the answer
    if (e == hipErrorStreamCaptureInvalidated) { *status = 2; return HOPMI_OK; }
    hopmi::set_error("hipStreamIsCapturing: %s", hipGetErrorString(e));
    return HOPMI_ELAUNCH;
  }
  *status = st == hipStreamCaptureStatusActive ? 1 : st == hipStreamCaptureStatusInvalidated ? 2 : 0;
  return HOPMI_OK;
}

extern "C" int hopmi_stream_capture_abandon(void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  (void)hipGetLastError();                       // (an invalidated capture answers with its error: expected, not sticky)
  if (g) (void)hipGraphDestroy(g);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t q = hipStreamIsCapturing(s, &st);
  (void)hipGetLastError();
  if (q != hipSuccess || st != hipStreamCaptureStatusNone) {
    hopmi::set_error("hipStreamEndCapture: %s; the stream is still in capture mode (%s)", hipGetErrorString(e),
                     q != hipSuccess ? hipGetErrorString(q) : st == hipStreamCaptureStatusActive ? "active" : "invalidated");
    return HOPMI_ELAUNCH;
  }
  return HOPMI_OK;
}
