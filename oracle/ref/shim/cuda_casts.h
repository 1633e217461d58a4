// oracle/ref/shim/cuda_casts.h — TEST INFRASTRUCTURE ONLY; force-included (after every system header) for SiftGPU/ProgramCU.cu only.
// ProgramCU.cu writes `(unsigned int)round(xmax - xmin + 1)` and `int depthx = round(...)`.  On the GPU the float -> integer conversion
// saturates (cvt.rzi: a negative value becomes 0 for an unsigned target, NaN becomes 0); on x86 a negative float converted to unsigned wraps
// to ~4e9, which turns the descriptor / orientation sampling loops of key points whose window lies outside the image into ~2^32 iterations.
// All eight uses of round() in that file convert to int or unsigned int, so round() is given the device conversion semantics here.
#pragma once
#include <cmath>
struct shim_rounded {
    float v;
    operator int() const { return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int)v)); }
    operator unsigned int() const { return !(v > 0.0f) ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (unsigned int)v); }
};
static inline shim_rounded shim_round(float x) { shim_rounded r = {roundf(x)}; return r; }
#define round(x) shim_round(x)
