// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of descriptor matching and of the
// three match filters + correspondence bookkeeping.
//   SiftGPU/ProgramCU.cu:1634-1936 (dot products, row / column best-2, mutual check),
//   SiftGPU/SiftMatch.cpp:160-196;  SiftGPU/SIFTImageManager.cu:59-143 (sort), :186-263 (Kabsch filter),
//   :318-389 (surface area), :418-585 (dense verify), :610-658 (add residuals), :692-774, :1036-1127;
//   SiftGPU/cuda_kabsch.h:73-211,278-502; SiftGPU/cuda_svd3.h (McAdams et al. 3x3 SVD);
//   SiftGPU/cuda_SVD.h:69-208 (cyclic Jacobi); SiftGPU/cuda_EigenValue.h:9-89; SiftGPU/cuda_surfaceArea.h.
// PINNED through oracle/_ref (tests/test_ref_pin_cpu.py): 3x3 SVD, eigen solver, Kabsch; the matcher (identical index pairs, distances within
// 1e-6: acos); the kernels of SIFTImageManager.cu with their launch configurations - Kabsch filter, surface-area filter (decision flips exactly
// at this file's area), dense verification (flips exactly at this file's error / correspondence fraction), EntryJ rows, VerifyTrajectory verdicts.
// Not pinnable: SortKeyPointMatchesCU_Kernel (its termination flag relies on lock-step warps).  Canonical choices: matches of a pair are emitted in ascending column (current-frame
// key) order before the distance sort (the reference appends with atomicAdd); the sums of the surface-area filter follow the reference's
// 32-lane warpReduceSum tree, other small sums (<= 25 terms) run in index order; the dense-verify sums follow the reference's block
// (per-thread row sums, 32-lane warp trees, adders in ascending thread order); rsqrt is 1/sqrt; acos comes from include/bf_detmath.h.
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <vector>

#include "../include/bf_detmath.h"
#include "../include/bf_hip.h"
#include "or_common.h"

using namespace orc;

namespace {

const int MAX_RAW = 128, MAX_FILT = 25;

inline float rsq(float x) { return 1.0f / sqrtf(x); }
struct f2 { float x, y; };

// ------------------------------------------------------------------ McAdams 3x3 SVD (cuda_svd3.h)
const float GAMMA_ = 5.828427124f, CSTAR = 0.923879532f, SSTAR = 0.3826834323f, SVD_EPS = 1e-6f;

void givensQuat(float a11, float a12, float a22, float& ch, float& sh) {
    ch = 2 * (a11 - a22); sh = a12;
    const bool b = GAMMA_ * sh * sh < ch * ch;
    const float w = rsq(ch * ch + sh * sh);
    ch = b ? w * ch : CSTAR; sh = b ? w * sh : SSTAR;
}
void jacobiConj(int x, int y, int z, float& s11, float& s21, float& s22, float& s31, float& s32, float& s33, float* q) {
    float ch, sh;
    givensQuat(s11, s21, s22, ch, sh);
    const float scale = ch * ch + sh * sh;
    const float a = (ch * ch - sh * sh) / scale, b = (2 * sh * ch) / scale;
    const float t11 = s11, t21 = s21, t22 = s22, t31 = s31, t32 = s32, t33 = s33;
    s11 = a * (a * t11 + b * t21) + b * (a * t21 + b * t22);
    s21 = a * (-b * t11 + a * t21) + b * (-b * t21 + a * t22);
    s22 = -b * (-b * t11 + a * t21) + a * (-b * t21 + a * t22);
    s31 = a * t31 + b * t32; s32 = -b * t31 + a * t32; s33 = t33;
    float tmp[3] = {q[0] * sh, q[1] * sh, q[2] * sh};
    sh *= q[3];
    q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
    q[z] += sh; q[3] -= tmp[z]; q[x] += tmp[y]; q[y] -= tmp[x];
    const float n11 = s22, n21 = s32, n22 = s33, n31 = s21, n32 = s31, n33 = s11;
    s11 = n11; s21 = n21; s22 = n22; s31 = n31; s32 = n32; s33 = n33;
}
inline void cswap(bool c, float& X, float& Y) { float Z = X; X = c ? Y : X; Y = c ? Z : Y; }
inline void cnswap(bool c, float& X, float& Y) { float Z = -X; X = c ? Y : X; Y = c ? Z : Y; }
void qrGivens(float a1, float a2, float& ch, float& sh) {
    const float rho = (a1 * a1 + a2 * a2) * rsq(a1 * a1 + a2 * a2);       // accurateSqrt = x*rsqrt(x)
    sh = rho > SVD_EPS ? a2 : 0;
    ch = fabsf(a1) + fmaxf(rho, SVD_EPS);
    cswap(a1 < 0, sh, ch);
    const float w = rsq(ch * ch + sh * sh);
    ch *= w; sh *= w;
}
// A (row-major 3x3) = U S V^T ; S returned as full 3x3 (upper triangular R of the QR step)
void svd3(const float* A, float* U, float* S, float* V) {
    const float a11 = A[0], a12 = A[1], a13 = A[2], a21 = A[3], a22 = A[4], a23 = A[5], a31 = A[6], a32 = A[7], a33 = A[8];
    float s11 = a11 * a11 + a21 * a21 + a31 * a31, s21 = a12 * a11 + a22 * a21 + a32 * a31, s22 = a12 * a12 + a22 * a22 + a32 * a32;
    float s31 = a13 * a11 + a23 * a21 + a33 * a31, s32 = a13 * a12 + a23 * a22 + a33 * a32, s33 = a13 * a13 + a23 * a23 + a33 * a33;
    float q[4] = {0, 0, 0, 1};
    for (int i = 0; i < 4; ++i) {
        jacobiConj(0, 1, 2, s11, s21, s22, s31, s32, s33, q);
        jacobiConj(1, 2, 0, s11, s21, s22, s31, s32, s33, q);
        jacobiConj(2, 0, 1, s11, s21, s22, s31, s32, s33, q);
    }
    const float w = q[3], x = q[0], y = q[1], z = q[2];
    float v11 = 1 - 2 * (y * y + z * z), v12 = 2 * (x * y - w * z), v13 = 2 * (x * z + w * y);
    float v21 = 2 * (x * y + w * z), v22 = 1 - 2 * (x * x + z * z), v23 = 2 * (y * z - w * x);
    float v31 = 2 * (x * z - w * y), v32 = 2 * (y * z + w * x), v33 = 1 - 2 * (x * x + y * y);
    float b11 = a11 * v11 + a12 * v21 + a13 * v31, b12 = a11 * v12 + a12 * v22 + a13 * v32, b13 = a11 * v13 + a12 * v23 + a13 * v33;
    float b21 = a21 * v11 + a22 * v21 + a23 * v31, b22 = a21 * v12 + a22 * v22 + a23 * v32, b23 = a21 * v13 + a22 * v23 + a23 * v33;
    float b31 = a31 * v11 + a32 * v21 + a33 * v31, b32 = a31 * v12 + a32 * v22 + a33 * v32, b33 = a31 * v13 + a32 * v23 + a33 * v33;
    // sortSingularValues (cuda_svd3.h:229-259) — rho2 uses b23 where b32 is meant; kept as in the reference
    float rho1 = b11 * b11 + b21 * b21 + b31 * b31, rho2 = b12 * b12 + b22 * b22 + b23 * b23, rho3 = b13 * b13 + b23 * b23 + b33 * b33;
    bool c = rho1 < rho2;
    cnswap(c, b11, b12); cnswap(c, v11, v12); cnswap(c, b21, b22); cnswap(c, v21, v22); cnswap(c, b31, b32); cnswap(c, v31, v32); cswap(c, rho1, rho2);
    c = rho1 < rho3;
    cnswap(c, b11, b13); cnswap(c, v11, v13); cnswap(c, b21, b23); cnswap(c, v21, v23); cnswap(c, b31, b33); cnswap(c, v31, v33); cswap(c, rho1, rho3);
    c = rho2 < rho3;
    cnswap(c, b12, b13); cnswap(c, v12, v13); cnswap(c, b22, b23); cnswap(c, v22, v23); cnswap(c, b32, b33); cnswap(c, v32, v33);
    // QRDecomposition :277-337
    float ch1, sh1, ch2, sh2, ch3, sh3, a, b;
    float r11, r12, r13, r21, r22, r23, r31, r32, r33;
    qrGivens(b11, b21, ch1, sh1);
    a = 1 - 2 * sh1 * sh1; b = 2 * ch1 * sh1;
    r11 = a * b11 + b * b21; r12 = a * b12 + b * b22; r13 = a * b13 + b * b23;
    r21 = -b * b11 + a * b21; r22 = -b * b12 + a * b22; r23 = -b * b13 + a * b23;
    r31 = b31; r32 = b32; r33 = b33;
    qrGivens(r11, r31, ch2, sh2);
    a = 1 - 2 * sh2 * sh2; b = 2 * ch2 * sh2;
    b11 = a * r11 + b * r31; b12 = a * r12 + b * r32; b13 = a * r13 + b * r33;
    b21 = r21; b22 = r22; b23 = r23;
    b31 = -b * r11 + a * r31; b32 = -b * r12 + a * r32; b33 = -b * r13 + a * r33;
    qrGivens(b22, b32, ch3, sh3);
    a = 1 - 2 * sh3 * sh3; b = 2 * ch3 * sh3;
    r11 = b11; r12 = b12; r13 = b13;
    r21 = a * b21 + b * b31; r22 = a * b22 + b * b32; r23 = a * b23 + b * b33;
    r31 = -b * b21 + a * b31; r32 = -b * b22 + a * b32; r33 = -b * b23 + a * b33;
    const float sh12 = sh1 * sh1, sh22 = sh2 * sh2, sh32 = sh3 * sh3;
    U[0] = (-1 + 2 * sh12) * (-1 + 2 * sh22);
    U[1] = 4 * ch2 * ch3 * (-1 + 2 * sh12) * sh2 * sh3 + 2 * ch1 * sh1 * (-1 + 2 * sh32);
    U[2] = 4 * ch1 * ch3 * sh1 * sh3 - 2 * ch2 * (-1 + 2 * sh12) * sh2 * (-1 + 2 * sh32);
    U[3] = 2 * ch1 * sh1 * (1 - 2 * sh22);
    U[4] = -8 * ch1 * ch2 * ch3 * sh1 * sh2 * sh3 + (-1 + 2 * sh12) * (-1 + 2 * sh32);
    U[5] = -2 * ch3 * sh3 + 4 * sh1 * (ch3 * sh1 * sh3 + ch1 * ch2 * sh2 * (-1 + 2 * sh32));
    U[6] = 2 * ch2 * sh2;
    U[7] = 2 * ch3 * (1 - 2 * sh22) * sh3;
    U[8] = (-1 + 2 * sh22) * (-1 + 2 * sh32);
    S[0] = r11; S[1] = r12; S[2] = r13; S[3] = r21; S[4] = r22; S[5] = r23; S[6] = r31; S[7] = r32; S[8] = r33;
    V[0] = v11; V[1] = v12; V[2] = v13; V[3] = v21; V[4] = v22; V[5] = v23; V[6] = v31; V[7] = v32; V[8] = v33;
}

inline float det3(const float* m) {   // cuda_SimpleMatrixUtil.h:1544-1559
    return m[0] * m[4] * m[8] + m[1] * m[5] * m[6] + m[2] * m[3] * m[7] - m[6] * m[4] * m[2] - m[7] * m[5] * m[0] - m[8] * m[3] * m[1];
}
inline void mm3(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// computeEigenValues (cuda_EigenValue.h:9-41) of a symmetric 3x3, x >= y >= z
f3 eigenValues3(const float* A) {
    const float PI = 3.14159265f;
    f3 e;
    float p = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (p == 0) { e.x = A[0]; e.y = A[4]; e.z = A[8]; return e; }
    const float q = (A[0] + A[4] + A[8]) / 3.0f;
    p = (A[0] - q) * (A[0] - q) + (A[4] - q) * (A[4] - q) + (A[8] - q) * (A[8] - q) + 2.0f * p;
    p = sqrtf(p / 6.0f);
    float B[9];
    const float ip = 1.0f / p;
    for (int i = 0; i < 9; ++i) B[i] = (A[i] - ((i % 4 == 0) ? q : 0.0f)) * ip;
    const float r = det3(B) / 2.0f;
    float phi;
    if (r <= -1.0f) phi = PI / 3.0f;
    else if (r >= 1) phi = 0;
    else phi = bf_dm_acos(r) / 3.0f;
    float s, c;
    bf_dm_sincos(phi, &s, &c);
    e.x = q + 2.0f * p * c;
    bf_dm_sincos(phi + PI * (2.0f / 3.0f), &s, &c);
    e.z = q + 2.0f * p * c;
    e.y = 3.0f * q - e.x - e.z;
    return e;
}

// cyclic Jacobi of a symmetric 3x3 (cuda_SVD.h:133-208, the classic rotation sweep), then
// MYEIGEN::eigenSystem :69-122: "eigenvector" i is ROW i of the rotation matrix (the columns hold the
// eigenvectors — kept as in the reference), sorted by decreasing |eigenvalue|.
bool eigenSystem3(const float* M, float* evals, float evecs[3][3]) {
    float a[3][3], v[3][3], d[3], b[3], z[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = M[i + 3 * j]; v[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (int i = 0; i < 3; ++i) { b[i] = d[i] = a[i][i]; z[i] = 0.0f; }
    bool ok = false;
    for (int sweep = 1; sweep <= 50; ++sweep) {
        float sm = 0.0f;
        for (int ip = 0; ip < 2; ++ip) for (int iq = ip + 1; iq < 3; ++iq) sm += fabsf(a[ip][iq]);
        if (sm == 0.0f) { ok = true; break; }
        const float tresh = sweep < 4 ? 0.2f * sm / 9 : 0.0f;
        for (int ip = 0; ip < 2; ++ip)
            for (int iq = ip + 1; iq < 3; ++iq) {
                const float g = 100.0f * fabsf(a[ip][iq]);
                if (sweep > 4 && (float)(fabsf(d[ip]) + g) == fabsf(d[ip]) && (float)(fabsf(d[iq]) + g) == fabsf(d[iq])) a[ip][iq] = 0.0f;
                else if (fabsf(a[ip][iq]) > tresh) {
                    float h = d[iq] - d[ip], t;
                    if ((float)(fabsf(h) + g) == fabsf(h)) t = a[ip][iq] / h;
                    else {
                        const float theta = 0.5f * h / a[ip][iq];
                        t = 1.0f / (fabsf(theta) + sqrtf(1.0f + theta * theta));
                        if (theta < 0.0f) t = -t;
                    }
                    const float c = 1.0f / sqrtf(1 + t * t), s = t * c, tau = s / (1.0f + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h;
                    a[ip][iq] = 0.0f;
                    auto rot2 = [&](float (*m)[3], int i, int j, int k, int l) { const float gg = m[i][j], hh = m[k][l]; m[i][j] = gg - s * (hh + gg * tau); m[k][l] = hh + s * (gg - hh * tau); };
                    for (int j = 0; j <= ip - 1; ++j) rot2(a, j, ip, j, iq);
                    for (int j = ip + 1; j <= iq - 1; ++j) rot2(a, ip, j, j, iq);
                    for (int j = iq + 1; j < 3; ++j) rot2(a, ip, j, iq, j);
                    for (int j = 0; j < 3; ++j) rot2(v, j, ip, j, iq);
                }
            }
        for (int i = 0; i < 3; ++i) { b[i] += z[i]; d[i] = b[i]; z[i] = 0.0f; }
    }
    if (!ok) return false;
    for (int i = 0; i < 3; ++i) { evals[i] = d[i]; for (int j = 0; j < 3; ++j) evecs[i][j] = v[i][j]; }
    for (int i = 0; i < 3; ++i) {
        float cur = 0.0f; int idx = -1;
        for (int j = i; j < 3; ++j) if (fabsf(evals[j]) > cur) { cur = fabsf(evals[j]); idx = j; }
        if (idx != i && idx != -1) { std::swap(evals[i], evals[idx]); for (int j = 0; j < 3; ++j) std::swap(evecs[i][j], evecs[idx][j]); }
    }
    return true;
}

struct Key { float x, y, scale, depth; };

inline f3 backProject(const float* Kinv, const Key& k) {      // colorIntrinsicsInv * (depth * (x, y, 1))
    const f3 p = {k.depth * k.x, k.depth * k.y, k.depth * 1.0f};
    return {Kinv[0] * p.x + Kinv[1] * p.y + Kinv[2] * p.z + Kinv[3] * 1.0f, Kinv[4] * p.x + Kinv[5] * p.y + Kinv[6] * p.z + Kinv[7] * 1.0f,
            Kinv[8] * p.x + Kinv[9] * p.y + Kinv[10] * p.z + Kinv[11] * 1.0f};
}

// cuda_kabsch.h:73-211
m44 kabsch(const f3* src, const f3* tgt, unsigned n, f3& evs) {
    f3 p0 = {0, 0, 0}, q0 = {0, 0, 0};
    for (unsigned i = 0; i < n; ++i) { p0 = p0 + src[i]; q0 = q0 + tgt[i]; }
    p0 = p0 / (float)n; q0 = q0 / (float)n;
    float V[9] = {0};
    for (unsigned i = 0; i < n; ++i) {
        const f3 p = src[i] - p0, q = tgt[i] - q0;
        const float pv[3] = {p.x, p.y, p.z}, qv[3] = {q.x, q.y, q.z};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V[r * 3 + c] += pv[r] * qv[c];
    }
    for (int i = 0; i < 9; ++i) V[i] /= (float)n;
    float U[9], S[9], W[9];
    svd3(V, U, S, W);
    float s[3] = {S[0], S[4], S[8]};
    for (int i = 0; i < 3; ++i) if (s[i] < 0.0f) { s[i] *= -1.0f; for (int j = 0; j < 3; ++j) U[j * 3 + i] *= -1.0f; }   // svdAbsEV
    evs = {s[0], s[1], s[2]};
    if (evs.x < evs.y) std::swap(evs.x, evs.y);
    if (evs.y < evs.z) std::swap(evs.y, evs.z);
    if (evs.x < evs.y) std::swap(evs.x, evs.y);
    float Wt[9] = {W[0], W[3], W[6], W[1], W[4], W[7], W[2], W[5], W[8]}, UWt[9];
    mm3(U, Wt, UWt);
    float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (det3(UWt) < 0) I[8] = -1;
    float WI[9], Ut[9] = {U[0], U[3], U[6], U[1], U[4], U[7], U[2], U[5], U[8]}, R[9];
    mm3(W, I, WI);
    mm3(WI, Ut, R);
    m44 ret = m44::identity();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ret(i, j) = R[i * 3 + j];
    ret(0, 3) = q0.x - (R[0] * p0.x + R[1] * p0.y + R[2] * p0.z);
    ret(1, 3) = q0.y - (R[3] * p0.x + R[4] * p0.y + R[5] * p0.z);
    ret(2, 3) = q0.z - (R[6] * p0.x + R[7] * p0.y + R[8] * p0.z);
    return ret;
}

f3 covarianceEig(const f3* pts, unsigned n) {      // covarianceSVD :213-229
    f3 p0 = {0, 0, 0};
    for (unsigned i = 0; i < n; ++i) p0 = p0 + pts[i];
    p0 = p0 / (float)n;
    float V[9] = {0};
    for (unsigned i = 0; i < n; ++i) {
        const f3 p = pts[i] - p0;
        const float pv[3] = {p.x, p.y, p.z};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V[r * 3 + c] += pv[r] * pv[c];
    }
    for (int i = 0; i < 9; ++i) V[i] /= (float)n;
    return eigenValues3(V);
}

struct Sel { uint32_t ix, iy; float dist; };

bool computeReprojection(f3* src, f3* tgt, unsigned n, float* res, m44& T, Sel* sel) {     // :386-419
    f3 ev;
    T = kabsch(src, tgt, n, ev);
    for (unsigned i = 0; i < n; ++i) { const f3 d = xform(T, src[i]) - tgt[i]; res[i] = dot(d, d); }
    for (unsigned i = 0; i < n; ++i)                       // sortKabschResiduals :376-384
        for (unsigned j = i; j < n; ++j)
            if (res[i] > res[j]) { std::swap(res[i], res[j]); std::swap(src[i], src[j]); std::swap(tgt[i], tgt[j]); std::swap(sel[i], sel[j]); }
    const float c1 = ev.x / ev.y;
    f3 e = covarianceEig(src, n);
    const float cp = e.x / e.y;
    e = covarianceEig(tgt, n);
    const float cq = e.x / e.y;
    if (c1 != c1 || cp != cp || cq != cq || fabsf(c1) > 100.0f || fabsf(cp) > 100.0f || fabsf(cq) > 100.0f) return false;
    return true;
}

// computeProjError (SIFTImageManager.cu:418-487), float normals
struct CF { const float* depth; const float* campos; const float* normals; };
void projError(unsigned idx, unsigned W, unsigned H, float distThresh, float normalThresh, const m44& T, const float* K, const CF& in,
               const CF& model, float dmin, float dmax, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    const float* p = in.campos + 4 * idx;
    const float* nin = in.normals + 4 * idx;
    const float dIn = in.depth[idx];
    if (p[0] != MINF && nin[0] != MINF && dIn >= dmin && dIn <= dmax) {
        const float pt[4] = {T.e[0] * p[0] + T.e[1] * p[1] + T.e[2] * p[2] + T.e[3] * p[3], T.e[4] * p[0] + T.e[5] * p[1] + T.e[6] * p[2] + T.e[7] * p[3],
                             T.e[8] * p[0] + T.e[9] * p[1] + T.e[10] * p[2] + T.e[11] * p[3], T.e[12] * p[0] + T.e[13] * p[1] + T.e[14] * p[2] + T.e[15] * p[3]};
        const float nt[3] = {T.e[0] * nin[0] + T.e[1] * nin[1] + T.e[2] * nin[2] + T.e[3] * 0.0f, T.e[4] * nin[0] + T.e[5] * nin[1] + T.e[6] * nin[2] + T.e[7] * 0.0f,
                             T.e[8] * nin[0] + T.e[9] * nin[1] + T.e[10] * nin[2] + T.e[11] * 0.0f};
        const float tx = K[0] * pt[0] + K[1] * pt[1] + K[2] * pt[2] + K[3] * 1.0f, ty = K[4] * pt[0] + K[5] * pt[1] + K[6] * pt[2] + K[7] * 1.0f,
                    tz = K[8] * pt[0] + K[9] * pt[1] + K[10] * pt[2] + K[11] * 1.0f;
        const int sx = f2i(roundf(tx / tz)), sy = f2i(roundf(ty / tz));
        if (sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H) {
            const float* pT = model.campos + 4 * (sy * W + sx);
            const float* nT = model.normals + 4 * (sy * W + sx);
            if (pT[0] != MINF && nT[0] != MINF) {
                const float dx = pt[0] - pT[0], dy = pt[1] - pT[1], dz = pt[2] - pT[2], dw = pt[3] - pT[3];
                const float d = sqrtf(dx * dx + dy * dy + dz * dz + dw * dw);
                const float dN = nt[0] * nT[0] + nt[1] * nT[1] + nt[2] * nT[2];
                const float projDepth = pt[2];
                const float tgtDepth = model.depth[sy * W + sx];
                if (tgtDepth >= dmin && tgtDepth <= dmax) {
                    const bool b = ((tgtDepth != MINF && projDepth < tgtDepth) && d > distThresh);
                    if ((dN >= normalThresh && d <= distThresh) || b) {
                        const float z01 = (pt[2] - dmin) / (dmax - dmin);
                        out[0] = d;
                        out[1] = fmaxf(0.0f, 0.5f * ((1.0f - d / distThresh) + (1.0f - z01)));
                        out[2] = 1.0f;
                    }
                }
            }
        }
    }
}

inline float butterfly64(float* l) {
    for (int o = 32; o > 0; o >>= 1) { float t[64]; for (int i = 0; i < 64; ++i) t[i] = l[i] + l[i ^ o]; memcpy(l, t, sizeof t); }
    return l[0];
}

}  // namespace

extern "C" {

// GetSiftMatch (SiftMatch.cpp:160-196) + SortKeyPointMatchesCU (SIFTImageManager.cu:59-143).
// Returns the match counter (may exceed 128 like the reference's); at most 128 are stored.
int or_sift_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float distmax, float ratiomax, uint32_t off1, uint32_t off2,
                  uint32_t* outIdx, float* outDist, int sortByDistance) {
    if (n1 <= 0 || n2 <= 0) return 0;
    std::vector<int> rowBest(n1, 0), rowNext(n1, 0), rowIdx(n1, -1), colBest(n2, 0), colNext(n2, 0), colIdx(n2, -1);
    std::vector<uint32_t> rowKey(n1, 0), colKey(n2, 0);
    for (int r = 0; r < n1; ++r)
        for (int c = 0; c < n2; ++c) {
            int dotv = 0;
            for (int k = 0; k < 128; ++k) dotv += (int)d1[r * 128 + k] * (int)d2[c * 128 + k];
            // row statistics: RowMatch_Kernel :1772-1831 (ties: lowest (c mod 32, c))
            const uint32_t kr = ((uint32_t)(c & 31) << 16) | (uint32_t)c;
            if (dotv > rowBest[r] || (dotv == rowBest[r] && rowIdx[r] >= 0 && kr < rowKey[r])) {
                if (dotv > rowBest[r]) rowNext[r] = rowBest[r]; else rowNext[r] = std::max(rowNext[r], dotv);
                rowBest[r] = dotv; rowIdx[r] = c; rowKey[r] = kr;
            } else rowNext[r] = std::max(rowNext[r], dotv);
            // column statistics: MultiplyDescriptor :1700-1714 + ColMatch :1852-1896 (ties: lowest (r/4 mod 32, r))
            const uint32_t kc = ((uint32_t)((r >> 2) & 31) << 16) | (uint32_t)r;
            if (dotv > colBest[c] || (dotv == colBest[c] && colIdx[c] >= 0 && kc < colKey[c])) {
                if (dotv > colBest[c]) colNext[c] = colBest[c]; else colNext[c] = std::max(colNext[c], dotv);
                colBest[c] = dotv; colIdx[c] = r; colKey[c] = kc;
            } else colNext[c] = std::max(colNext[c], dotv);
        }
    std::vector<int> rowRes(n1);
    std::vector<float> rowDist(n1);
    for (int r = 0; r < n1; ++r) {
        const float dist = bf_dm_acos(fminf((float)rowBest[r] * 0.000003814697265625f, 1.0f));
        const float distn = bf_dm_acos(fminf((float)rowNext[r] * 0.000003814697265625f, 1.0f));
        rowRes[r] = (dist < distmax) && (dist < distn * ratiomax) ? rowIdx[r] : -1;
        rowDist[r] = dist;
    }
    int count = 0;
    for (int c = 0; c < n2; ++c) {
        const float dist = bf_dm_acos(fminf((float)colBest[c] * 0.000003814697265625f, 1.0f));
        const float distn = bf_dm_acos(fminf((float)colNext[c] * 0.000003814697265625f, 1.0f));
        const int f1 = (dist < distmax) && (dist < distn * ratiomax) ? colIdx[c] : -1;
        if (f1 >= 0 && rowRes[f1] == c) {
            if (count < MAX_RAW) { outIdx[2 * count] = (uint32_t)f1 + off1; outIdx[2 * count + 1] = (uint32_t)c + off2; outDist[count] = rowDist[f1]; }
            count++;
        }
    }
    if (sortByDistance) {      // odd-even transposition with strict '>' == stable ascending sort
        const int n = std::min(count, MAX_RAW);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return outDist[a] < outDist[b]; });
        std::vector<uint32_t> ti(outIdx, outIdx + 2 * n);
        std::vector<float> td(outDist, outDist + n);
        for (int i = 0; i < n; ++i) { outIdx[2 * i] = ti[2 * order[i]]; outIdx[2 * i + 1] = ti[2 * order[i] + 1]; outDist[i] = td[order[i]]; }
    }
    return count;
}

// filterKeyPointMatches (cuda_kabsch.h:422-502).  keys: all keypoints (4 floats each), idx/dist: sorted raw
// matches.  Outputs the filtered matches (<=25) and the 4x4 transform; returns the filtered count.
int or_filter_keypoint_matches(const float* keys4, uint32_t* idx, float* dist, int numRaw, const float* Kinv, int minNumMatches,
                               float maxKabschRes2, float* transform16) {
    const Key* keys = (const Key*)keys4;
    numRaw = std::min(numRaw, MAX_RAW);
    std::vector<Sel> sel(numRaw + MAX_FILT);
    for (int i = 0; i < numRaw; ++i) sel[i] = {idx[2 * i], idx[2 * i + 1], dist[i]};
    f3 src[MAX_FILT], tgt[MAX_FILT];
    float res[MAX_FILT];
    unsigned cur = 0;
    int i = 0;
    float curMaxRes = 100.0f;
    bool validT = false;
    m44 T = m44::identity();
    for (;;) {
        if (i == numRaw || cur >= (unsigned)MAX_FILT) {
            if ((int)cur < minNumMatches || curMaxRes >= maxKabschRes2 || !validT) cur = 0;
            break;
        }
        // addMatch :278-294
        bool add = true;
        {
            const Key& ai = keys[sel[i].ix]; const Key& aj = keys[sel[i].iy];
            for (unsigned k = 0; k < cur; ++k) {
                const Key& ki = keys[sel[k].ix]; const Key& kj = keys[sel[k].iy];
                const float d0 = sqrtf((ai.x - ki.x) * (ai.x - ki.x) + (ai.y - ki.y) * (ai.y - ki.y));
                const float d1 = sqrtf((aj.x - kj.x) * (aj.x - kj.x) + (aj.y - kj.y) * (aj.y - kj.y));
                if (d0 <= 5 || d1 <= 5) { add = false; break; }
            }
        }
        if (add) {
            sel[cur] = sel[i];
            cur++;
            if (cur >= 3) {
                for (unsigned k = 0; k < cur; ++k) { src[k] = backProject(Kinv, keys[sel[k].ix]); tgt[k] = backProject(Kinv, keys[sel[k].iy]); }
                validT = computeReprojection(src, tgt, cur, res, T, sel.data());
                const bool b = validT;
                const m44 prevT = T;
                curMaxRes = res[cur - 1];
                if (curMaxRes > maxKabschRes2) {
                    float lastRes = -1;
                    const int startIdx = (int)cur - 1;
                    for (int k = startIdx; k >= 3; --k) {
                        lastRes = res[k];
                        cur--;
                        validT = computeReprojection(src, tgt, cur, res, T, sel.data());
                        curMaxRes = res[cur - 1];
                        if (cur == 3 && (curMaxRes > maxKabschRes2 || (b && !validT))) {
                            cur++; curMaxRes = lastRes; validT = b; T = prevT;
                            break;
                        }
                        if (curMaxRes < maxKabschRes2) break;
                    }
                }
            }
        }
        i++;
    }
    for (unsigned k = 0; k < cur; ++k) { idx[2 * k] = sel[k].ix; idx[2 * k + 1] = sel[k].iy; dist[k] = sel[k].dist; }
    memcpy(transform16, T.e, 64);
    return (int)cur;
}

// warpReduceSum (cudaUtil.h:25-29) as lane 0 of a 32-lane warp sees it: val += shfl_down(val, 16), 8, 4, 2, 1; lanes >= n hold 0
static float warpTree32(const float* v, int n) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = i < n ? v[i] : 0.0f;
    for (int off = 16; off > 0; off >>= 1)
        for (int i = 0; i < off; ++i) x[i] = x[i] + x[i + off];
    return x[0];
}

// FilterMatchesBySurfaceAreaCU_Kernel (SIFTImageManager.cu:318-389) with computeKeyPointMatchesCovariance / computeCovariance2d /
// computeAreaOrientedBoundingBox2 (cuda_surfaceArea.h:13-131): one 32-thread block per pair, every sum is a warpReduceSum;
// returns 1 if the pair survives
int or_filter_surface_area(const float* keys4, const uint32_t* idx, int n, const float* Kinv, float areaThresh, float* areas2) {
    const Key* keys = (const Key*)keys4;
    float area[2] = {0.0f, 0.0f};
    float t[32];
    for (int which = 0; which < 2; ++which) {
        f3 pts[MAX_FILT];
        for (int i = 0; i < n; ++i) pts[i] = backProject(Kinv, keys[idx[2 * i + which]]);
        f3 mean;
        for (int i = 0; i < n; ++i) t[i] = pts[i].x; mean.x = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = pts[i].y; mean.y = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = pts[i].z; mean.z = warpTree32(t, n);
        mean = mean / (float)n;
        float V[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                for (int i = 0; i < n; ++i) {
                    const f3 p = pts[i] - mean;
                    const float pv[3] = {p.x, p.y, p.z};
                    t[i] = pv[r] * pv[c];
                }
                V[r * 3 + c] = warpTree32(t, n);
            }
        for (int i = 0; i < 9; ++i) V[i] /= (float)n;
        float evals[3], ev[3][3];
        if (!eigenSystem3(V, evals, ev)) continue;
        const f3 ev0 = {ev[0][0], ev[0][1], ev[0][2]}, ev1 = {ev[1][0], ev[1][1], ev[1][2]}, ev2 = {ev[2][0], ev[2][1], ev[2][2]};
        f2 proj[MAX_FILT];
        for (int i = 0; i < n; ++i) {           // projectKeysToPlane, cuda_surfaceArea.h:134-158
            const f3 s = (pts[i] - ev2 * dot(ev2, pts[i] - mean)) - mean;
            proj[i] = {dot(s, ev0), dot(s, ev1)};
        }
        // computeAreaOrientedBoundingBox2 :87-131
        f2 m2;
        for (int i = 0; i < n; ++i) t[i] = proj[i].x; m2.x = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = proj[i].y; m2.y = warpTree32(t, n);
        m2.x /= (float)n; m2.y /= (float)n;
        float c00, c01, c10, c11;
        for (int i = 0; i < n; ++i) { const float a = proj[i].x - m2.x; t[i] = a * a; } c00 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float a = proj[i].x - m2.x, b = proj[i].y - m2.y; t[i] = a * b; } c01 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float a = proj[i].x - m2.x, b = proj[i].y - m2.y; t[i] = b * a; } c10 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float b = proj[i].y - m2.y; t[i] = b * b; } c11 = warpTree32(t, n);
        c00 /= (float)n; c01 /= (float)n; c10 /= (float)n; c11 /= (float)n;
        const float disc = 0.5f * sqrtf((c00 - c11) * (c00 - c11) + 4 * c01 * c01);
        const float l1 = (c00 + c11) / 2 + disc, l2 = (c00 + c11) / 2 - disc;
        auto evec2 = [&](float lam) { float vx = -c01, vy = c00 - lam; const float mag = sqrtf(vx * vx + vy * vy); vx /= mag; vy /= mag; return f2{vx, vy}; };
        f2 a0 = evec2(l1), a1 = evec2(l2);
        auto norm2 = [](f2 v) { const float il = 1.0f / sqrtf(v.x * v.x + v.y * v.y); return f2{v.x * il, v.y * il}; };
        a0 = norm2(a0); a1 = norm2(a1);
        float minx = FLT_MAX, miny = FLT_MAX, maxx = -FLT_MAX, maxy = -FLT_MAX;
        for (int i = 0; i < n; ++i) {
            const float cx = a0.x * proj[i].x + a0.y * proj[i].y, cy = a1.x * proj[i].x + a1.y * proj[i].y;
            minx = fminf(minx, cx); miny = fminf(miny, cy); maxx = fmaxf(maxx, cx); maxy = fmaxf(maxy, cy);
        }
        const float ex = maxx - minx, ey = maxy - miny;
        area[which] = (ex < 0.00001f || ey < 0.00001f) ? 0.0f : ex * ey;
    }
    if (areas2) { areas2[0] = area[0]; areas2[1] = area[1]; }
    return (area[0] < areaThresh && area[1] < areaThresh) ? 0 : 1;
}

// FilterMatchesByDenseVerifyCU_Kernel (:491-585) / VerifyTrajectoryCU_Kernel (:1036-1127); returns 1 if valid.
// The block sum is restated as the reference executes it, not as it was meant: the block is (W, ceil(H/32)) threads, thread (x, ty) sums
// its 32 rows in order, warpReduceSum runs over the 32-lane warps of the LINEAR thread id ty*W + x (a shuffle from a lane outside the
// warp returns the caller's own value), and the threads with threadIdx.x % 32 == 0 add what they hold to the block total - for W = 80
// those are lane 0 of warps 0-2 and lane 16 of warps 2-4, so the upper rows count once and of the lower rows the columns 0-15 count
// three times, 32-47 and 64-79 twice and the rest not at all (pinned against the reference's kernel: tests/test_ref_pin_cpu.py).
// The atomicAdd order of those threads is arbitrary in the reference; here it is ascending thread id.
int or_dense_verify(const float* inDepth, const float* inCampos, const float* inNormals, const float* moDepth, const float* moCampos,
                    const float* moNormals, unsigned W, unsigned H, const float* K16, const float* T16, float distThresh, float normalThresh,
                    float errThresh, float corrThresh, float dmin, float dmax, float* errOut, float* corrOut) {
    m44 T; memcpy(T.e, T16, 64);
    const m44 Tinv = inverse(T);
    const CF in = {inDepth, inCampos, inNormals}, mo = {moDepth, moCampos, moNormals};
    const unsigned rows = (H + 31) / 32, nt = W * rows;
    std::vector<float> loc(3 * (size_t)nt, 0.0f);
    for (unsigned v = 0; v < nt; ++v) {
        const unsigned x = v % W, ty = v / W;
        for (unsigned i = 0; i < 32; ++i) {
            const unsigned y = ty * 32 + i;
            if (y >= H) continue;
            const unsigned idx = y * W + x;
            float a[3], b[3];
            projError(idx, W, H, distThresh, normalThresh, T, K16, in, mo, dmin, dmax, a);
            projError(idx, W, H, distThresh, normalThresh, Tinv, K16, mo, in, dmin, dmax, b);
            for (int k = 0; k < 3; ++k) loc[3 * v + k] += a[k] + b[k];
        }
    }
    for (unsigned w0 = 0; w0 < nt; w0 += 32)                              // warpReduceSum per warp of the linear thread id
        for (int k = 0; k < 3; ++k) {
            float x[32], y[32];
            for (unsigned l = 0; l < 32; ++l) x[l] = w0 + l < nt ? loc[3 * (w0 + l) + k] : 0.0f;
            for (unsigned off = 16; off > 0; off >>= 1) {
                for (unsigned l = 0; l < 32; ++l) { const bool ex = l + off < 32 && w0 + l + off < nt; y[l] = x[l] + (ex ? x[l + off] : x[l]); }
                memcpy(x, y, sizeof x);
            }
            for (unsigned l = 0; l < 32 && w0 + l < nt; ++l) loc[3 * (w0 + l) + k] = x[l];
        }
    float tot[3] = {0.0f, 0.0f, 0.0f};
    for (unsigned v = 0; v < nt; ++v)
        if ((v % W) % 32 == 0)
            for (int k = 0; k < 3; ++k) tot[k] += loc[3 * v + k];
    const float err = tot[0] / tot[1];
    const float corr = 0.5f * tot[2] / (float)(W * H);
    if (errOut) *errOut = err;
    if (corrOut) *corrOut = corr;
    return (corr < corrThresh || err > errThresh || err != err) ? 0 : 1;
}

// AddCurrToResidualsCU_Kernel (:610-658): EntryJ of one filtered match
void or_make_entry(const float* keys4, uint32_t ix, uint32_t iy, uint32_t img_i, uint32_t img_j, const float* Kinv, bf_entry_j* e) {
    const Key* keys = (const Key*)keys4;
    const f3 a = backProject(Kinv, keys[ix]), b = backProject(Kinv, keys[iy]);
    e->imgIdx_i = img_i; e->imgIdx_j = img_j;
    e->pos_i[0] = a.x; e->pos_i[1] = a.y; e->pos_i[2] = a.z;
    e->pos_j[0] = b.x; e->pos_j[1] = b.y; e->pos_j[2] = b.z;
}

void or_svd3(const float* A, float* U, float* S, float* V) { svd3(A, U, S, V); }
void or_kabsch(const float* src3, const float* tgt3, int n, float* T16, float* evs3) {
    f3 ev;
    const m44 T = kabsch((const f3*)src3, (const f3*)tgt3, (unsigned)n, ev);
    memcpy(T16, T.e, 64);
    evs3[0] = ev.x; evs3[1] = ev.y; evs3[2] = ev.z;
}

}  // extern "C"

extern "C" {
void or_inverse44(const float* T16, float* out16) { m44 T; memcpy(T.e, T16, 64); const m44 r = inverse(T); memcpy(out16, r.e, 64); }
void or_mul44(const float* A16, const float* B16, float* out16) {
    m44 A, B; memcpy(A.e, A16, 64); memcpy(B.e, B16, 64);
    const m44 r = mul(A, B); memcpy(out16, r.e, 64);
}
}
