// TEST INFRASTRUCTURE ONLY (see or_common.h) — SE(3)/so(3) exp & log and small fixed-size
// matrices, restating Solver/LieDerivUtil.h:19-307 (device) == PoseHelper.h:215-426 (host twin).
// PINNED to LieDerivUtil.h through oracle/_ref (tests/test_ref_pin_cpu.py::test_lie_maps_within_detmath_bound: the reference calls libm,
// this restatement the fixed sequences of include/bf_detmath.h; the difference is bounded, not zero).
#pragma once
#include "or_common.h"

namespace orc {

struct m33 {
    float e[9];
    float& operator()(int r, int c) { return e[r * 3 + c]; }
    float operator()(int r, int c) const { return e[r * 3 + c]; }
};
inline f3 operator*(const m33& m, f3 v) {
    return {m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z, m.e[3] * v.x + m.e[4] * v.y + m.e[5] * v.z,
            m.e[6] * v.x + m.e[7] * v.y + m.e[8] * v.z};
}
inline m33 mul33(const m33& a, const m33& b) {
    m33 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.e[i * 3 + j] = a.e[i * 3] * b.e[j] + a.e[i * 3 + 1] * b.e[3 + j] + a.e[i * 3 + 2] * b.e[6 + j];
    return r;
}
inline m33 rot33(const m44& m) { return m33{{m.e[0], m.e[1], m.e[2], m.e[4], m.e[5], m.e[6], m.e[8], m.e[9], m.e[10]}}; }
inline f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(f3 a) { return sqrtf(dot(a, a)); }

const float ONE_TWENTIETH = 0.05f;
const float ONE_SIXTH = 0.16666667f;

// LieDerivUtil.h:19-47
inline void rodrigues_so3_exp(f3 w, float A, float B, m33& R) {
    {
        const float wx2 = w.x * w.x, wy2 = w.y * w.y, wz2 = w.z * w.z;
        R(0, 0) = 1.0f - B * (wy2 + wz2);
        R(1, 1) = 1.0f - B * (wx2 + wz2);
        R(2, 2) = 1.0f - B * (wx2 + wy2);
    }
    { const float a = A * w.z, b = B * (w.x * w.y); R(0, 1) = b - a; R(1, 0) = b + a; }
    { const float a = A * w.y, b = B * (w.x * w.z); R(0, 2) = b + a; R(2, 0) = b - a; }
    { const float a = A * w.x, b = B * (w.y * w.z); R(1, 2) = b - a; R(2, 1) = b + a; }
}
// :50-76
inline m33 exp_rotation(f3 w) {
    const float theta_sq = dot(w, w);
    const float theta = sqrtf(theta_sq);
    float A, B;
    if (theta_sq < 1e-8f) { A = 1.0f - ONE_SIXTH * theta_sq; B = 0.5f; }
    else if (theta_sq < 1e-6f) { B = 0.5f - 0.25f * ONE_SIXTH * theta_sq; A = 1.0f - theta_sq * ONE_SIXTH * (1.0f - ONE_TWENTIETH * theta_sq); }
    else { const float inv_theta = 1.0f / theta; A = sinf(theta) * inv_theta; B = (1 - cosf(theta)) * (inv_theta * inv_theta); }
    m33 R;
    rodrigues_so3_exp(w, A, B, R);
    return R;
}
// :79-133
inline f3 ln_rotation(const m33& R) {
    f3 result;
    const float cos_angle = ((R(0, 0) + R(1, 1) + R(2, 2)) - 1.0f) * 0.5f;
    result.x = (R(2, 1) - R(1, 2)) * 0.5f;
    result.y = (R(0, 2) - R(2, 0)) * 0.5f;
    result.z = (R(1, 0) - R(0, 1)) * 0.5f;
    float sin_angle_abs = length(result);
    if (cos_angle > 0.70710678118654752440f) {
        if (sin_angle_abs > 0) result = result * (asinf(sin_angle_abs) / sin_angle_abs);
    } else if (cos_angle > -0.70710678118654752440f) {
        const float angle = acosf(cos_angle);
        result = result * (angle / sin_angle_abs);
    } else {
        const float angle = 3.14159265358979323846f - asinf(sin_angle_abs);
        const float d0 = R(0, 0) - cos_angle, d1 = R(1, 1) - cos_angle, d2 = R(2, 2) - cos_angle;
        f3 r2;
        if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) { r2.x = d0; r2.y = (R(1, 0) + R(0, 1)) * 0.5f; r2.z = (R(0, 2) + R(2, 0)) * 0.5f; }
        else if (fabsf(d1) > fabsf(d2)) { r2.x = (R(1, 0) + R(0, 1)) * 0.5f; r2.y = d1; r2.z = (R(2, 1) + R(1, 2)) * 0.5f; }
        else { r2.x = (R(0, 2) + R(2, 0)) * 0.5f; r2.y = (R(2, 1) + R(1, 2)) * 0.5f; r2.z = d2; }
        if (dot(r2, result) < 0) r2 = r2 * -1.0f;
        result = r2 * (angle / length(r2));
    }
    return result;
}
// :135-158
inline void matrixToPose(const m44& M, f3& rot, f3& trans) {
    const m33 R = rot33(M);
    const f3 t = {M.e[3], M.e[7], M.e[11]};
    rot = ln_rotation(R);
    const float theta = length(rot);
    float shtot = 0.5f;
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    const m33 halfrotator = exp_rotation(rot * -0.5f);
    trans = halfrotator * t;
    if (theta > 0.001f) trans = trans - rot * (dot(t, rot) * (1 - 2 * shtot) / dot(rot, rot));
    else trans = trans - rot * (dot(t, rot) / 24);
    trans = trans * (1.0f / (2 * shtot));
}
// :160-207
inline m44 poseToMatrix(f3 rot, f3 trans) {
    m44 M = m44::identity();
    f3 translation;
    m33 rotation;
    const float theta_sq = dot(rot, rot);
    const float theta = sqrtf(theta_sq);
    float A, B;
    const f3 cr = cross(rot, trans);
    if (theta_sq < 1e-8f) {
        A = 1.0f - ONE_SIXTH * theta_sq; B = 0.5f;
        translation = trans + cr * 0.5f;
    } else {
        float C;
        if (theta_sq < 1e-6f) { C = ONE_SIXTH * (1.0f - ONE_TWENTIETH * theta_sq); A = 1.0f - theta_sq * C; B = 0.5f - 0.25f * ONE_SIXTH * theta_sq; }
        else { const float inv_theta = 1.0f / theta; A = sinf(theta) * inv_theta; B = (1 - cosf(theta)) * (inv_theta * inv_theta); C = (1 - A) * (inv_theta * inv_theta); }
        const f3 w_cross = cross(rot, cr);
        translation = trans + cr * B + w_cross * C;
    }
    rodrigues_so3_exp(rot, A, B, rotation);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M(r, c) = rotation(r, c);
    M(0, 3) = translation.x; M(1, 3) = translation.y; M(2, 3) = translation.z;
    return M;
}
// :301-307
inline void computeLieUpdate(f3 updW, f3 updT, f3 curW, f3 curT, f3& newW, f3& newT) {
    const m44 update = poseToMatrix(updW, updT);
    const m44 cur = poseToMatrix(curW, curT);
    matrixToPose(mul(update, cur), newW, newT);
}
// :231-242
inline f3 dAlpha(f3 p) { return {0.0f, -p.z, p.y}; }
inline f3 dBeta(f3 p) { return {p.z, 0.0f, -p.x}; }
inline f3 dGamma(f3 p) { return {-p.y, p.x, 0.0f}; }

}  // namespace orc
