// SE(3) exponential / logarithm on the device (Lie-algebra pose parametrisation of the bundling
// solver).  Same closed forms as the reference's Solver/LieDerivUtil.h:19-207,301-307 (Rodrigues with
// Taylor branches near 0, three-branch log); written for gfx950 with IEEE sin/cos/asin/acos.
#pragma once
#include "bf_device.h"

namespace bf {

struct m33 { float e[9]; };
BF_HD f3 mul33v(const m33& m, f3 v) {
    return mk3(m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z, m.e[3] * v.x + m.e[4] * v.y + m.e[5] * v.z,
               m.e[6] * v.x + m.e[7] * v.y + m.e[8] * v.z);
}
BF_HD float len3(f3 a) { return sqrtf(dot3(a, a)); }

BF_HD void rodrigues(f3 w, float A, float B, m33& R) {
    const float wx2 = w.x * w.x, wy2 = w.y * w.y, wz2 = w.z * w.z;
    R.e[0] = 1.0f - B * (wy2 + wz2);
    R.e[4] = 1.0f - B * (wx2 + wz2);
    R.e[8] = 1.0f - B * (wx2 + wy2);
    float a = A * w.z, b = B * (w.x * w.y);
    R.e[1] = b - a; R.e[3] = b + a;
    a = A * w.y; b = B * (w.x * w.z);
    R.e[2] = b + a; R.e[6] = b - a;
    a = A * w.x; b = B * (w.y * w.z);
    R.e[5] = b - a; R.e[7] = b + a;
}

BF_HD m33 expRotation(f3 w) {
    const float t2 = dot3(w, w);
    const float t = sqrtf(t2);
    float A, B;
    if (t2 < 1e-8f) { A = 1.0f - 0.16666667f * t2; B = 0.5f; }
    else if (t2 < 1e-6f) { B = 0.5f - 0.25f * 0.16666667f * t2; A = 1.0f - t2 * 0.16666667f * (1.0f - 0.05f * t2); }
    else { const float it = 1.0f / t; A = sinf(t) * it; B = (1 - cosf(t)) * (it * it); }
    m33 R;
    rodrigues(w, A, B, R);
    return R;
}

BF_HD f3 lnRotation(const m33& R) {
    f3 res;
    const float c = ((R.e[0] + R.e[4] + R.e[8]) - 1.0f) * 0.5f;
    res.x = (R.e[7] - R.e[5]) * 0.5f;
    res.y = (R.e[2] - R.e[6]) * 0.5f;
    res.z = (R.e[3] - R.e[1]) * 0.5f;
    const float s = len3(res);
    if (c > 0.70710678118654752440f) {
        if (s > 0) res = res * (asinf(s) / s);
    } else if (c > -0.70710678118654752440f) {
        res = res * (acosf(c) / s);
    } else {
        const float angle = 3.14159265358979323846f - asinf(s);
        const float d0 = R.e[0] - c, d1 = R.e[4] - c, d2 = R.e[8] - c;
        f3 r2;
        if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) r2 = mk3(d0, (R.e[3] + R.e[1]) * 0.5f, (R.e[2] + R.e[6]) * 0.5f);
        else if (fabsf(d1) > fabsf(d2)) r2 = mk3((R.e[3] + R.e[1]) * 0.5f, d1, (R.e[7] + R.e[5]) * 0.5f);
        else r2 = mk3((R.e[2] + R.e[6]) * 0.5f, (R.e[7] + R.e[5]) * 0.5f, d2);
        if (dot3(r2, res) < 0) r2 = r2 * -1.0f;
        res = r2 * (angle / len3(r2));
    }
    return res;
}

BF_HD void matrixToPose(const m44& M, f3& rot, f3& trans) {
    m33 R;
    R.e[0] = M.e[0]; R.e[1] = M.e[1]; R.e[2] = M.e[2]; R.e[3] = M.e[4]; R.e[4] = M.e[5]; R.e[5] = M.e[6]; R.e[6] = M.e[8]; R.e[7] = M.e[9]; R.e[8] = M.e[10];
    const f3 t = mk3(M.e[3], M.e[7], M.e[11]);
    rot = lnRotation(R);
    const float theta = len3(rot);
    float shtot = 0.5f;
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    const m33 half = expRotation(rot * -0.5f);
    trans = mul33v(half, t);
    if (theta > 0.001f) trans = trans - rot * (dot3(t, rot) * (1 - 2 * shtot) / dot3(rot, rot));
    else trans = trans - rot * (dot3(t, rot) / 24);
    trans = trans * (1.0f / (2 * shtot));
}

BF_HD m44 poseToMatrix(f3 rot, f3 trans) {
    m44 M = identity44();
    const float t2 = dot3(rot, rot);
    const float t = sqrtf(t2);
    float A, B;
    f3 tr;
    const f3 cr = cross3(rot, trans);
    if (t2 < 1e-8f) {
        A = 1.0f - 0.16666667f * t2; B = 0.5f;
        tr = trans + cr * 0.5f;
    } else {
        float C;
        if (t2 < 1e-6f) { C = 0.16666667f * (1.0f - 0.05f * t2); A = 1.0f - t2 * C; B = 0.5f - 0.25f * 0.16666667f * t2; }
        else { const float it = 1.0f / t; A = sinf(t) * it; B = (1 - cosf(t)) * (it * it); C = (1 - A) * (it * it); }
        tr = trans + cr * B + cross3(rot, cr) * C;
    }
    m33 R;
    rodrigues(rot, A, B, R);
    M.e[0] = R.e[0]; M.e[1] = R.e[1]; M.e[2] = R.e[2]; M.e[4] = R.e[3]; M.e[5] = R.e[4]; M.e[6] = R.e[5]; M.e[8] = R.e[6]; M.e[9] = R.e[7]; M.e[10] = R.e[8];
    M.e[3] = tr.x; M.e[7] = tr.y; M.e[11] = tr.z;
    return M;
}

BF_HD void lieUpdate(f3 updW, f3 updT, f3 curW, f3 curT, f3& newW, f3& newT) {
    matrixToPose(mul44(poseToMatrix(updW, updT), poseToMatrix(curW, curT)), newW, newT);
}

}  // namespace bf
