// Device-side math helpers shared by the gfx950 kernels.  IEEE binary32, built with
// -ffp-contract=off so every kernel's arithmetic is the op-by-op sequence written
// here (bit-comparable with a CPU evaluation of the same sequence).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf {

#define BF_DEV __device__ __forceinline__
#define BF_HD __host__ __device__ __forceinline__

struct f3 { float x, y, z; };
struct i3 { int x, y, z; };

BF_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
BF_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
BF_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
BF_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
BF_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
BF_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
BF_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

#define BF_MINF (-__builtin_huge_valf())
#define BF_PINF (__builtin_huge_valf())

// float -> int32 toward zero, saturating, NaN -> 0 (what v_cvt_i32_f32 does; spelled
// out because an out-of-range fptosi is poison to the optimiser).
BF_HD int f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
// float -> uint32 toward zero, saturating (negative and NaN -> 0), as v_cvt_u32_f32
BF_HD uint32_t f2u(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
BF_HD int sgn(float v) { return (0.0f < v) - (v < 0.0f); }

// row-major 4x4 (reference float4x4 / mat4f memory layout)
struct m44 { float e[16]; };

BF_HD f3 xform(const m44& m, f3 v) {   // implicit w = 1
    return mk3(m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z + m.e[3] * 1.0f,
               m.e[4] * v.x + m.e[5] * v.y + m.e[6] * v.z + m.e[7] * 1.0f,
               m.e[8] * v.x + m.e[9] * v.y + m.e[10] * v.z + m.e[11] * 1.0f);
}
BF_HD f3 rot(const m44& m, f3 v) {
    return mk3(m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z,
               m.e[4] * v.x + m.e[5] * v.y + m.e[6] * v.z,
               m.e[8] * v.x + m.e[9] * v.y + m.e[10] * v.z);
}
BF_HD m44 mul44(const m44& a, const m44& b) {
    m44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.e[i * 4 + j] = a.e[i * 4 + 0] * b.e[0 * 4 + j] + a.e[i * 4 + 1] * b.e[1 * 4 + j] +
                             a.e[i * 4 + 2] * b.e[2 * 4 + j] + a.e[i * 4 + 3] * b.e[3 * 4 + j];
    return r;
}
BF_HD m44 identity44() {
    m44 r;
    for (int i = 0; i < 16; ++i) r.e[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return r;
}
// Cofactor inverse of a general 4x4: adj(r,c) = (-1)^(r+c) det3(minor without row c, col r),
// det3 by the Leibniz sum taken in the fixed term order
//   a00 a11 a22 - a00 a12 a21 - a10 a01 a22 + a10 a02 a21 + a20 a01 a12 - a20 a02 a11,
// then scaled by 1/det (det expanded along row 0).
BF_HD m44 inverse44(const m44& m) {
    m44 adj;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            int rows[3], cols[3], n = 0;
            for (int i = 0; i < 4; ++i) if (i != c) rows[n++] = i;
            n = 0;
            for (int j = 0; j < 4; ++j) if (j != r) cols[n++] = j;
            const float s = ((r + c) & 1) ? -1.0f : 1.0f;
#define BF_A(i, j) m.e[rows[i] * 4 + cols[j]]
            adj.e[r * 4 + c] = (s * BF_A(0, 0)) * BF_A(1, 1) * BF_A(2, 2) - (s * BF_A(0, 0)) * BF_A(1, 2) * BF_A(2, 1) -
                               (s * BF_A(1, 0)) * BF_A(0, 1) * BF_A(2, 2) + (s * BF_A(1, 0)) * BF_A(0, 2) * BF_A(2, 1) +
                               (s * BF_A(2, 0)) * BF_A(0, 1) * BF_A(1, 2) - (s * BF_A(2, 0)) * BF_A(0, 2) * BF_A(1, 1);
#undef BF_A
        }
    const float det = m.e[0] * adj.e[0] + m.e[1] * adj.e[4] + m.e[2] * adj.e[8] + m.e[3] * adj.e[12];
    const float detr = 1.0f / det;
    m44 res;
    for (int i = 0; i < 16; ++i) res.e[i] = adj.e[i] * detr;
    return res;
}

// ---- wave64 reductions (DPP/ds_swizzle via __shfl_xor, width 64) ----
BF_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
BF_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
BF_DEV int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
BF_DEV unsigned wave_max_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

}  // namespace bf
