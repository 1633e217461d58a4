// Host-side plumbing shared by the C-ABI translation units: status/error handling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/bf_hip.h"

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
