// TEST INFRASTRUCTURE ONLY — CPU oracle for the BundleFusion hot path.
// Nothing in the product (bundlefusion_amd/, include/) may include, link or call
// anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker.
//
// This oracle is a from-scratch restatement of the reference algorithm text; each
// function cites the file:line it follows (paths relative to /root/reference/FriedLiver/Source).
// PINNING: the reference (niessner/BundleFusion) ships no golden vectors and no tests, and its
// application cannot be built here (CUDA 7 + Windows/DirectX + un-vendored mLib) - but its
// DEVICE code and the HOST classes of this path can: oracle/ref/Makefile compiles them for the host into
// oracle/_ref/libbfref.so and tests/test_ref_pin_cpu.py compares this oracle with it on the same inputs (the host
// classes - CUDAImageManager, OnlineBundler, Bundler, SBA, CUDASolverBundling, CUDACache, TrajectoryManager,
// CUDASceneRepHashSDF, CorrespondenceEvaluator - end to end against tests/oracle_pipeline.py; golden vectors written by
// that build are in tests/golden/).  Pinned per stage:
// the integer maps, SE(3), SVD / Kabsch / greedy Kabsch filter, TSDF operators, image operators
// and the cache frame, the GN/PCG solver, marching cubes, the ray-cast kernel, the whole SiftGPU
// fork (pyramid, detection, descriptors, matcher) and the match-filter chain of SIFTImageManager.cu.
// Each oracle file states what of its stage is pinned and to which bound.
// The reference itself is not bit-reproducible
// (atomic append order, bucket try-locks, -use_fast_math), so wherever it is
// order-dependent the oracle fixes ONE canonical order, documented at the site.
//
// Arithmetic: IEEE-754 binary32, round-to-nearest-even, no FMA contraction
// (build with -ffp-contract=off), correctly rounded / and sqrt.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

// Host threads of the image-space loops (every output pixel is computed independently, so the results do not depend on it).
// Set by or_set_threads (cpu_baseline: 1 thread and all cores); the voxel update takes its thread count per call.
extern int g_threads;

static const float MINF = -std::numeric_limits<float>::infinity();
static const float PINF = std::numeric_limits<float>::infinity();

struct f3 { float x, y, z; };
struct i3 { int x, y, z; };

inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// float -> int, round toward zero.  CUDA cvt.rzi.s32.f32 and gfx950 v_cvt_i32_f32
// both saturate and map NaN to 0; plain C casts are UB there, so spell it out.
inline int f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
// float -> uint32 toward zero: cvt.rzi.u32.f32 / v_cvt_u32_f32 saturate (negative and NaN -> 0)
inline uint32_t f2u(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
// cutil_math.h:31  sign(): (0<v)-(v<0)
inline int sgn(float v) { return (0.0f < v) - (v < 0.0f); }

// Row-major 4x4, the reference's float4x4 (SiftGPU/cuda_SimpleMatrixUtil.h:855).
struct m44 {
    float e[16];
    float& operator()(int r, int c) { return e[r * 4 + c]; }
    float operator()(int r, int c) const { return e[r * 4 + c]; }
    static m44 identity() {
        m44 m;
        for (int i = 0; i < 16; ++i) m.e[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return m;
    }
};
// float4x4 * float3 with implicit w=1   (cuda_SimpleMatrixUtil.h:937-944)
inline f3 xform(const m44& m, f3 v) {
    return {m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z + m.e[3] * 1.0f,
            m.e[4] * v.x + m.e[5] * v.y + m.e[6] * v.z + m.e[7] * 1.0f,
            m.e[8] * v.x + m.e[9] * v.y + m.e[10] * v.z + m.e[11] * 1.0f};
}
// float3x3 part * float3
inline f3 rot(const m44& m, f3 v) {
    return {m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z,
            m.e[4] * v.x + m.e[5] * v.y + m.e[6] * v.z,
            m.e[8] * v.x + m.e[9] * v.y + m.e[10] * v.z};
}
inline m44 mul(const m44& a, const m44& b) {
    m44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.e[i * 4 + j] = a.e[i * 4 + 0] * b.e[0 * 4 + j] + a.e[i * 4 + 1] * b.e[1 * 4 + j] +
                             a.e[i * 4 + 2] * b.e[2 * 4 + j] + a.e[i * 4 + 3] * b.e[3 * 4 + j];
    return r;
}
// General 4x4 inverse by cofactors, float4x4::getInverse (cuda_SimpleMatrixUtil.h:980-1104).
// The reference spells out 16 six-term expansions; every one of them is the Leibniz
// expansion  a00*a11*a22 - a00*a12*a21 - a10*a01*a22 + a10*a02*a21 + a20*a01*a12 - a20*a02*a11
// of the 3x3 minor (rows/cols in increasing order), with the cofactor sign applied
// term-wise (exact), then  det = sum_k e[0][k]*inv[k][0],  res = inv * (1/det).
inline m44 inverse(const m44& m) {
    m44 adj;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            // adj(r,c) = (-1)^(r+c) * minor(remove row c, col r)
            int rows[3], cols[3], n = 0;
            for (int i = 0; i < 4; ++i) if (i != c) rows[n++] = i;
            n = 0;
            for (int j = 0; j < 4; ++j) if (j != r) cols[n++] = j;
            auto a = [&](int i, int j) { return m.e[rows[i] * 4 + cols[j]]; };
            const float s = ((r + c) & 1) ? -1.0f : 1.0f;
            adj.e[r * 4 + c] = (s * a(0, 0)) * a(1, 1) * a(2, 2) - (s * a(0, 0)) * a(1, 2) * a(2, 1) -
                               (s * a(1, 0)) * a(0, 1) * a(2, 2) + (s * a(1, 0)) * a(0, 2) * a(2, 1) +
                               (s * a(2, 0)) * a(0, 1) * a(1, 2) - (s * a(2, 0)) * a(0, 2) * a(1, 1);
        }
    const float det = m.e[0] * adj.e[0] + m.e[1] * adj.e[4] + m.e[2] * adj.e[8] + m.e[3] * adj.e[12];
    const float detr = 1.0f / det;
    m44 res;
    for (int i = 0; i < 16; ++i) res.e[i] = adj.e[i] * detr;
    return res;
}

}  // namespace orc
