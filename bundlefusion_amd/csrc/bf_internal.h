// Host-side plumbing shared by the C-ABI translation units: status/error handling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/bf_hip.h"

// Diagnostic (tools/first_run_check.py): BF_DEBUG_POISON="<file>[:<line>]=<hex32>[,...]" fills every allocation made at that place (or "*": everywhere) with the
// 32-bit pattern - a result that changes with the pattern was computed from memory nobody had written.  Without the variable: hipMalloc and one getenv per process.  (BF_MALLOC below.)
#include <cstdlib>
#include <cstring>
inline hipError_t bf_debug_malloc(void** p, size_t n, const char* file, int line) {
    const hipError_t e = hipMalloc(p, n);
    static const char* env = getenv("BF_DEBUG_POISON");
    if (e != hipSuccess || !env || n < 4) return e;
    const char* base = strrchr(file, '/'); base = base ? base + 1 : file;
    char key[128]; snprintf(key, sizeof key, "%s:%d=", base, line);
    char keyFile[128]; snprintf(keyFile, sizeof keyFile, "%s=", base);
    const char* hit = strstr(env, key);
    if (!hit) { hit = strstr(env, keyFile); }
    if (!hit) { hit = strstr(env, "*="); }
    if (!hit) return e;
    const unsigned v = (unsigned)strtoul(strchr(hit, '=') + 1, nullptr, 16);
    (void)hipMemsetD32((hipDeviceptr_t)*p, (int)v, n / 4);
    (void)hipDeviceSynchronize();
    return e;
}
// every device allocation of the library goes through this name (an explicitly named wrapper: the HIP API name itself is not redefined)
#define BF_MALLOC(p, n) bf_debug_malloc((void**)(p), (n), __FILE__, __LINE__)


namespace bf {

void set_error(const char* fmt, ...);

#define BF_HIP_TRY(expr)                                                                     \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            bf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return BF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define BF_REQUIRE(cond, msg)                                      \
    do {                                                           \
        if (!(cond)) {                                             \
            bf::set_error("%s: %s", __func__, msg);                \
            return BF_ERR_INVALID_ARG;                             \
        }                                                          \
    } while (0)

inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

}  // namespace bf
