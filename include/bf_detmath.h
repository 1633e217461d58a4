/*
 * bf_detmath.h — deterministic single-precision elementary functions.
 *
 * The SIFT front end needs exp, atan2, sin/cos and acos.  libm results differ between a host CPU
 * and the GPU's device library in the last bit, which would make keypoint orientations (floor() of
 * an angle), descriptor bytes and ratio-test decisions platform dependent.  These versions are part
 * of the arithmetic contract of this ABI: fixed sequences of IEEE-754 binary32 +,-,*,/ and sqrt only
 * (no FMA, build with -ffp-contract=off), so every conforming implementation — the gfx950 kernels and
 * the CPU oracle — returns the same bits.  Accuracy is ~2 ulp on the ranges used here, i.e. the same
 * class as the reference's -use_fast_math intrinsics (__expf, __sincosf, atan2f, acosf).
 *
 * Plain C, usable from host and device code.
 */
#ifndef BF_DETMATH_H
#define BF_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BF_DM_FN __host__ __device__ static inline
#else
#define BF_DM_FN static inline
#endif

BF_DM_FN float bf_dm_from_bits(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
BF_DM_FN uint32_t bf_dm_bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }
/* correctly rounded square root on both sides (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt
 * expands this to the IEEE sequence; __fsqrt_rn would map to the 1-ulp native instruction) */
BF_DM_FN float bf_dm_sqrt(float x) { return __builtin_sqrtf(x); }

/* round to nearest integer, ties away from zero, |x| < 2^22 */
BF_DM_FN float bf_dm_round(float x) { return (float)(int)(x >= 0.0f ? x + 0.5f : x - 0.5f); }

/* e^x for x in [-87, 87]; below -87 returns 0 */
BF_DM_FN float bf_dm_exp(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 87.0f) x = 87.0f;
    const float k = bf_dm_round(x * 1.44269504088896341f);
    /* Cody-Waite: ln2 = 0.693145751953125 + 1.42860682030941723212e-6 */
    float r = x - k * 0.693145751953125f;
    r = r - k * 1.42860682030941723212e-6f;
    /* degree-6 Taylor/minimax on [-ln2/2, ln2/2] */
    float p = 1.3888889e-3f;
    p = p * r + 8.3333338e-3f;
    p = p * r + 4.1666668e-2f;
    p = p * r + 1.6666667e-1f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const int ki = (int)k;
    return p * bf_dm_from_bits((uint32_t)(ki + 127) << 23);   /* exact scaling by 2^k, k in [-126,126] */
}

/* atan(a) for a in [0,1] */
BF_DM_FN float bf_dm_atan01(float a) {
    /* reduce to [0, tan(pi/8)] : atan(a) = pi/4 + atan((a-1)/(a+1)) */
    float off = 0.0f;
    if (a > 0.41421356237f) { a = (a - 1.0f) / (a + 1.0f); off = 0.78539816339744831f; }
    const float s = a * a;
    float p = -0.0752896400f;      /* odd minimax, |a| <= 0.4143 */
    p = p * s + 0.1065626393f;
    p = p * s - 0.1420889944f;
    p = p * s + 0.1999355085f;
    p = p * s - 0.3333314528f;
    p = p * s * a + a;
    return off + p;
}

/* atan2(y, x) in (-pi, pi]; atan2(0,0) = 0 */
BF_DM_FN float bf_dm_atan2(float y, float x) {
    const float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float r;
    if (ay <= ax) r = bf_dm_atan01(ay / ax);
    else r = 1.57079632679489662f - bf_dm_atan01(ax / ay);
    if (x < 0.0f) r = 3.14159265358979324f - r;
    return y < 0.0f ? -r : r;
}

/* sin and cos of a, |a| <= ~100 */
BF_DM_FN void bf_dm_sincos(float a, float* s, float* c) {
    const float k = bf_dm_round(a * 0.63661977236758134f);        /* quadrant */
    /* pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8 */
    float r = a - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188e-8f;
    const float z = r * r;
    float sp = -1.9515295891e-4f;
    sp = sp * z + 8.3321608736e-3f;
    sp = sp * z - 1.6666654611e-1f;
    sp = sp * z * r + r;
    float cp = 2.443315711809948e-5f;
    cp = cp * z - 1.388731625493765e-3f;
    cp = cp * z + 4.166664568298827e-2f;
    cp = cp * z * z - 0.5f * z + 1.0f;
    const int q = ((int)k) & 3;
    if (q == 0) { *s = sp; *c = cp; }
    else if (q == 1) { *s = cp; *c = -sp; }
    else if (q == 2) { *s = -sp; *c = -cp; }
    else { *s = -cp; *c = sp; }
}

/* asin(z) for z in [0, 0.5] */
BF_DM_FN float bf_dm_asin05(float z) {
    const float s = z * z;
    float p = 4.2163199048e-2f;
    p = p * s + 2.4181311049e-2f;
    p = p * s + 4.5470025998e-2f;
    p = p * s + 7.4953002686e-2f;
    p = p * s + 1.6666752422e-1f;
    return p * s * z + z;
}

/* acos(x) for x in [-1, 1] */
BF_DM_FN float bf_dm_acos(float x) {
    const float ax = x < 0.0f ? -x : x;
    float r;
    if (ax > 0.5f) {
        const float z = bf_dm_sqrt((1.0f - ax) * 0.5f);
        r = 2.0f * bf_dm_asin05(z);                 /* acos(|x|) */
    } else {
        r = 1.57079632679489662f - bf_dm_asin05(ax);
    }
    return x < 0.0f ? 3.14159265358979324f - r : r;
}

#endif /* BF_DETMATH_H */
