// Error plumbing of the C ABI (thread-local last-error string).
#include "bf_internal.h"

namespace bf {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace bf

extern "C" {
const char* bf_last_error(void) { return bf::g_last_error.c_str(); }
const char* bf_version(void) { return "bundlefusion_amd 0.1 (gfx950)"; }
int bf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { bf::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return BF_ERR_NO_DEVICE; }
    return n;
}
}
