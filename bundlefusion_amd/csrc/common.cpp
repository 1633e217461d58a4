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
// Host<->device copies and a device-wide fence issued by THIS library's HIP runtime, so that callers which hold
// only raw pointers (ctypes, cgo, JNI ...) order their transfers against the kernels launched through this ABI.
int bf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return BF_OK;
    BF_HIP_TRY(hipDeviceSynchronize());
    BF_HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return BF_OK;
}
int bf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return BF_OK;
    BF_HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    BF_HIP_TRY(hipDeviceSynchronize());
    return BF_OK;
}
int bf_device_synchronize(void) { BF_HIP_TRY(hipDeviceSynchronize()); return BF_OK; }
int bf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { bf::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return BF_ERR_NO_DEVICE; }
    return n;
}
}
