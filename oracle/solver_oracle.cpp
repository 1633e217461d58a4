// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of the bundling solver:
// Gauss-Newton over SE(3) poses (image 0 fixed) with the sparse 3-D point term (matrix-free
// J / J^T) and the dense depth+colour term (explicit 6N x 6N JtJ / Jtr), Jacobi-PCG.
//   Solver/SolverBundling.cu:30-306 (dense system), :511-550 (max residual), :576-614 (energy),
//   :657-749 (high-residual count, GN convergence), :755-1022 (PCG kernels), :1024-1108 (iteration
//   control, early-out 5e-7), :1137-1220 (GN loop, early-out 0.005), :1226-1248 (variable table);
//   Solver/SolverBundlingEquationsLie.h:27-277; Solver/SolverBundlingDenseUtil.h:22-113,229-298,
//   371-424; Solver/ICPUtil.h:14-111; Solver/LieDerivUtil.h; SBA.cu:75-108; CUDACameraUtil.h.
// Sums that the reference forms with float atomics / warp shuffles are formed here in index
// order (the reference's order is non-deterministic).  PINNED to Solver/SolverBundling.cu through oracle/_ref (sparse and sparse+dense problems, poses 1e-4 / 1e-3:
// tests/test_ref_pin_cpu.py::test_solver_vs_reference_kernels, ::test_solver_dense_terms_vs_reference_kernels).
#include <algorithm>
#include <cstdio>
#include <vector>

#include "../include/bf_hip.h"
#include "or_se3.h"

using namespace orc;

namespace {

const float FLOAT_EPSILON = 0.000001f;   // SolverUtil.h:9

struct EntryJ { uint32_t i, j; float pi[3]; float pj[3]; };
inline bool valid(const EntryJ& c) { return c.i != 0xFFFFFFFFu; }
inline f3 v3(const float* p) { return {p[0], p[1], p[2]}; }

struct CacheFrame {     // CUDACachedFrame, CUDACacheUtil.h:10-53
    const float* depth; const float* campos4; const float* intensity; const float* derivs2;
    const uint8_t* normalsU4; const float* normals4;
};

struct Params {         // SolverParameters + thresholds of CUDASolverBundling.cpp:93-100
    float denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax;
    uint32_t denseOverlapCheckSubsampleFactor;
    float weightSparse, weightDenseDepth, weightDenseColor;
    bool useDense, usePairwise;
};

struct Solver {
    uint32_t N, C, maxCorrPerImage;
    EntryJ* corr;
    const int* validImages;
    std::vector<CacheFrame> cache;
    uint32_t W = 0, H = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    Params p;
    std::vector<f3> xRot, xTrans, deltaRot, deltaTrans, rRot, rTrans, zRot, zTrans, pRot, pTrans, ApRot, ApTrans, precRot, precTrans, Jp;
    std::vector<float> rDotzOld;
    std::vector<m44> T, Tinv;
    std::vector<std::vector<int>> varToCorr;
    std::vector<float> JtJ, Jtr, corrCounts;
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    // diagnostics
    std::vector<float> convergence;
    int pcgIterations = 0, gnIterations = 0;
};

// ---- ICPUtil.h bilinear (invalid = MINF in .x) ----
template <int K>
bool bilinear(float x, float y, const float* in, unsigned W, unsigned H, float* out) {
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const float alpha = x - (float)x0, beta = y - (float)y0;
    float s0[K], s1[K], w0 = 0.0f, w1 = 0.0f;
    for (int k = 0; k < K; ++k) s0[k] = s1[k] = 0.0f;
    auto tap = [&](int px, int py, float wgt, float* s, float& w) {
        if ((unsigned)px < W && (unsigned)py < H) {
            const float* v = in + (size_t)K * ((size_t)py * W + px);
            if (v[0] != MINF) { for (int k = 0; k < K; ++k) s[k] += wgt * v[k]; w += wgt; }
        }
    };
    tap(x0, y0, 1.0f - alpha, s0, w0);
    tap(x0 + 1, y0, alpha, s0, w0);
    tap(x0, y0 + 1, 1.0f - alpha, s1, w1);
    tap(x0 + 1, y0 + 1, alpha, s1, w1);
    float ss[K], ww = 0.0f;
    for (int k = 0; k < K; ++k) ss[k] = 0.0f;
    if (w0 > 0.0f) { for (int k = 0; k < K; ++k) ss[k] += (1.0f - beta) * (s0[k] / w0); ww += (1.0f - beta); }
    if (w1 > 0.0f) { for (int k = 0; k < K; ++k) ss[k] += beta * (s1[k] / w1); ww += beta; }
    if (ww > 0.0f) { for (int k = 0; k < K; ++k) out[k] = ss[k] / ww; return true; }
    for (int k = 0; k < K; ++k) out[k] = MINF;
    return false;
}

inline f3 depthToCamera(const Solver& s, int x, int y, float d) {
    const float kx = ((float)x - s.cx) / s.fx, ky = ((float)y - s.cy) / s.fy;
    return {d * kx, d * ky, d};
}
inline void cameraToDepth(const Solver& s, f3 p, float& u, float& v) { u = p.x * s.fx / p.z + s.cx; v = p.y * s.fy / p.z + s.cy; }

// SolverBundlingDenseUtil.h:22-42 (pre-filter, depth only)
bool findDenseCorrDepthOnly(const Solver& s, unsigned idx, const m44& tr, const float* tgtDepth, const float* srcDepth) {
    const unsigned x = idx % s.W, y = idx / s.W;
    const f3 cposj = depthToCamera(s, (int)x, (int)y, srcDepth[idx]);
    if (cposj.z > s.p.denseDepthMin && cposj.z < s.p.denseDepthMax) {
        const f3 q = xform(tr, cposj);
        float u, v;
        cameraToDepth(s, q, u, v);
        const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
        if (tx >= 0 && ty >= 0 && tx < (int)s.W && ty < (int)s.H) {
            const f3 ct = depthToCamera(s, tx, ty, tgtDepth[ty * s.W + tx]);
            if (ct.z > s.p.denseDepthMin && ct.z < s.p.denseDepthMax)
                if (length(q - ct) <= s.p.denseDistThresh) return true;
        }
    }
    return false;
}
// :152-184 (uchar4 normals, depth images) — used for the pair weights
bool findDenseCorrUchar(const Solver& s, unsigned idx, const m44& tr, const CacheFrame& tgt, const CacheFrame& src) {
    const unsigned x = idx % s.W, y = idx / s.W;
    const f3 cposj = depthToCamera(s, (int)x, (int)y, src.depth[idx]);
    if (cposj.z > s.p.denseDepthMin && cposj.z < s.p.denseDepthMax) {
        const uint8_t* nu = src.normalsU4 + 4 * idx;
        if (nu[0] | nu[1] | nu[2] | nu[3]) {
            f3 nrmj = {(float)nu[0] / 255.0f * 2.0f - 1.0f, (float)nu[1] / 255.0f * 2.0f - 1.0f, (float)nu[2] / 255.0f * 2.0f - 1.0f};
            nrmj = rot(tr, nrmj);
            const f3 q = xform(tr, cposj);
            float u, v;
            cameraToDepth(s, q, u, v);
            const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
            if (tx >= 0 && ty >= 0 && tx < (int)s.W && ty < (int)s.H) {
                const f3 ct = depthToCamera(s, tx, ty, tgt.depth[ty * s.W + tx]);
                if (ct.z > s.p.denseDepthMin && ct.z < s.p.denseDepthMax) {
                    const uint8_t* nt = tgt.normalsU4 + 4 * (ty * s.W + tx);
                    if (nt[0] | nt[1] | nt[2] | nt[3]) {
                        const f3 nT = {(float)nt[0] / 255.0f * 2.0f - 1.0f, (float)nt[1] / 255.0f * 2.0f - 1.0f, (float)nt[2] / 255.0f * 2.0f - 1.0f};
                        const float dist = length(q - ct);
                        const float dN = dot(nrmj, nT);
                        if (dN >= s.p.denseNormalThresh && dist <= s.p.denseDistThresh) return true;
                    }
                }
            }
        }
    }
    return false;
}
// :79-113 (camera positions + float normals) — used to build the system
bool findDenseCorrFull(const Solver& s, unsigned idx, const m44& tr, const CacheFrame& tgt, const CacheFrame& src,
                       f3& camPosSrc, f3& camPosSrcToTgt, float& u, float& v, f3& camPosTgt, f3& normalTgt) {
    const float* cp = src.campos4 + 4 * idx;
    if (cp[2] > s.p.denseDepthMin && cp[2] < s.p.denseDepthMax) {
        camPosSrc = {cp[0], cp[1], cp[2]};
        const float* nj = src.normals4 + 4 * idx;
        if (nj[0] != MINF) {
            // transform * float4 normal (w = 0 for valid normals, CUDAImageUtil.cu:429)
            const float n4[4] = {tr.e[0] * nj[0] + tr.e[1] * nj[1] + tr.e[2] * nj[2] + tr.e[3] * nj[3],
                                 tr.e[4] * nj[0] + tr.e[5] * nj[1] + tr.e[6] * nj[2] + tr.e[7] * nj[3],
                                 tr.e[8] * nj[0] + tr.e[9] * nj[1] + tr.e[10] * nj[2] + tr.e[11] * nj[3],
                                 tr.e[12] * nj[0] + tr.e[13] * nj[1] + tr.e[14] * nj[2] + tr.e[15] * nj[3]};
            camPosSrcToTgt = xform(tr, camPosSrc);
            cameraToDepth(s, camPosSrcToTgt, u, v);
            const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
            if (tx >= 0 && ty >= 0 && tx < (int)s.W && ty < (int)s.H) {
                float ci[4];
                bilinear<4>(u, v, tgt.campos4, s.W, s.H, ci);
                if (ci[2] > s.p.denseDepthMin && ci[2] < s.p.denseDepthMax) {
                    camPosTgt = {ci[0], ci[1], ci[2]};
                    float ni[4];
                    bilinear<4>(u, v, tgt.normals4, s.W, s.H, ni);
                    if (ni[0] != MINF) {
                        normalTgt = {ni[0], ni[1], ni[2]};
                        const float dist = length(camPosSrcToTgt - camPosTgt);
                        const float dN = n4[0] * ni[0] + n4[1] * ni[1] + n4[2] * ni[2] + n4[3] * ni[3];
                        if (dN >= s.p.denseNormalThresh && dist <= s.p.denseDistThresh) return true;
                    }
                }
            }
        }
    }
    return false;
}

// SolverBundlingDenseUtil.h:416-424
bool computeAngleDiff(const m44& tr, float thresh) {
    const float il = 1.0f / sqrtf(3.0f);
    const f3 x = {1.0f * il, 1.0f * il, 1.0f * il};
    const f3 v = rot(tr, x);
    const float c = std::min(std::max(dot(x, v), -1.0f), 1.0f);
    return fabsf(acosf(c)) < thresh;
}

// LieDerivUtil.h:247-272: 3x6 = d/d(xi_i) of (Tj^-1 e^xi Ti)^-1 ... evaluated as j0(3x12) * j1(12x6)
void derivI(const m44& A, const m44& D, f3 p, float jac[3][6]) {
    float j0[3][12] = {{0}}, j1[12][6] = {{0}};
    const m44 tr = mul(A, D);
    const f3 pt = {p.x - tr.e[3], p.y - tr.e[7], p.z - tr.e[11]};
    j0[0][0] = pt.x; j0[0][1] = pt.y; j0[0][2] = pt.z;
    j0[1][3] = pt.x; j0[1][4] = pt.y; j0[1][5] = pt.z;
    j0[2][6] = pt.x; j0[2][7] = pt.y; j0[2][8] = pt.z;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { j0[r][c + 9] = -tr(c, r); j1[r + 9][c] = A(r, c); }
    const m33 RA = rot33(A);
    for (int k = 0; k < 4; ++k) {
        const f3 v = {D(0, k), D(1, k), D(2, k)};
        m33 skew = {{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}};
        m33 m = mul33(RA, skew);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) j1[3 * k + r][3 + c] = m(r, c) * -1.0f;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) {
            float acc = 0.0f;
            for (int k = 0; k < 12; ++k) acc += j0[r][k] * j1[k][c];
            jac[r][c] = acc;
        }
}
// :277-295
void derivJ(const m44& A, const m44& D, f3 p, float jac[3][6]) {
    const f3 dr1 = {D(0, 0), D(0, 1), D(0, 2)}, dr2 = {D(1, 0), D(1, 1), D(1, 2)}, dr3 = {D(2, 0), D(2, 1), D(2, 2)};
    const float dtx = D(0, 3), dty = D(1, 3), dtz = D(2, 3);
    float j[3][6] = {{1, 0, 0, 0.0f, dot(p, dr3) + dtz, -(dot(p, dr2) + dty)},
                     {0, 1, 0, -(dot(p, dr3) + dtz), 0.0f, dot(p, dr1) + dtx},
                     {0, 0, 1, dot(p, dr2) + dty, -(dot(p, dr1) + dtx), 0.0f}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) jac[r][c] = A(r, 0) * j[0][c] + A(r, 1) * j[1][c] + A(r, 2) * j[2][c];
}

void buildVarTable(Solver& s) {   // SolverBundling.cu:1226-1248 (index order instead of atomic order)
    s.varToCorr.assign(s.N, {});
    for (uint32_t c = 0; c < s.C; ++c) {
        EntryJ& e = s.corr[c];
        if (!valid(e)) continue;
        if (s.varToCorr[e.i].size() < s.maxCorrPerImage && s.varToCorr[e.j].size() < s.maxCorrPerImage) {
            s.varToCorr[e.i].push_back((int)c);
            s.varToCorr[e.j].push_back((int)c);
        } else { e.i = e.j = 0xFFFFFFFFu; }
    }
}

// ---- BuildDenseSystem, SolverBundling.cu:308-471 ----
bool buildDenseSystem(Solver& s) {
    const uint32_t N = s.N, dim = 6 * N, npix = s.W * s.H;
    s.JtJ.assign((size_t)dim * dim, 0.0f);
    s.Jtr.assign(dim, 0.0f);
    s.pairs.clear();
    // FindImageImageCorr_Kernel :30-79
    const uint32_t sub = s.p.denseOverlapCheckSubsampleFactor;
    const uint32_t subW = s.W / sub;
    auto testPair = [&](uint32_t i, uint32_t j) {
        if (s.validImages[i] == 0 || s.validImages[j] == 0) return;
        const m44 tr = mul(s.Tinv[i], s.T[j]);
        if (!computeAngleDiff(tr, 0.52f)) return;
        int found = 0;
        for (uint32_t t = 0; t < 512; ++t) {
            const uint32_t x = (t % subW) * sub, y = (t / subW) * sub, idx = y * s.W + x;
            if (idx < npix && findDenseCorrDepthOnly(s, idx, tr, s.cache[i].depth, s.cache[j].depth)) found++;
        }
        if (found > 10) s.pairs.push_back({i, j});
    };
    if (s.p.usePairwise) { for (uint32_t i = 0; i < N; ++i) for (uint32_t j = i + 1; j < N; ++j) testPair(i, j); }
    else { for (uint32_t i = 0; i + 1 < N; ++i) testPair(i, i + 1); }
    if (s.pairs.empty()) return false;
    // FindDenseCorrespondences_Kernel :92-160 + WeightDenseCorrespondences_Kernel :162-180
    s.corrCounts.assign(s.pairs.size(), 0.0f);
    for (size_t pi = 0; pi < s.pairs.size(); ++pi) {
        const uint32_t i = s.pairs[pi].first, j = s.pairs[pi].second;
        const m44 tr = mul(s.Tinv[i], s.T[j]);
        int count = 0;
        for (uint32_t idx = 0; idx < npix; ++idx) count += findDenseCorrUchar(s, idx, tr, s.cache[i], s.cache[j]) ? 1 : 0;
        float x = (float)count;
        if (x > 0) { if (x < 800) x = 0; else x = 1.0f / std::min(logf(x), 9.0f); }
        s.corrCounts[pi] = x;
    }
    // BuildDenseSystem_Kernel<depth,color> :182-306 + addToLocalSystem (DenseUtil :229-298)
    const bool useDepth = s.p.weightDenseDepth > 0.0f, useColor = s.p.weightDenseColor > 0.0f;
    auto addLocal = [&](const float* Ji, const float* Jj, uint32_t vi, uint32_t vj, float res, float w) {
        for (uint32_t a = 0; a < 6; ++a) {
            for (uint32_t b = a; b < 6; ++b) {
                if (vi > 0) s.JtJ[(size_t)(vi * 6 + b) * dim + (vi * 6 + a)] += Ji[a] * Ji[b] * w;
                if (vj > 0) s.JtJ[(size_t)(vj * 6 + b) * dim + (vj * 6 + a)] += Jj[a] * Jj[b] * w;
                if (vi > 0 && vj > 0) {
                    s.JtJ[(size_t)(vj * 6 + b) * dim + (vi * 6 + a)] += Ji[a] * Jj[b] * w;
                    if (a != b) s.JtJ[(size_t)(vj * 6 + a) * dim + (vi * 6 + b)] += Ji[b] * Jj[a] * w;
                }
            }
            if (vi > 0) s.Jtr[vi * 6 + a] += Ji[a] * res * w;
            if (vj > 0) s.Jtr[vj * 6 + a] += Jj[a] * res * w;
        }
    };
    for (size_t pi = 0; pi < s.pairs.size(); ++pi) {
        const uint32_t i = s.pairs[pi].first, j = s.pairs[pi].second;
        const float pw = s.corrCounts[pi];
        if (pw == 0.0f) continue;
        const m44 &Ti = s.T[i], &Tj = s.T[j], &TiI = s.Tinv[i], &TjI = s.Tinv[j];
        const m44 tr = mul(TiI, Tj);
        for (uint32_t idx = 0; idx < npix; ++idx) {
            f3 cs, cst, ct, nt;
            float u, v;
            const bool found = findDenseCorrFull(s, idx, tr, s.cache[i], s.cache[j], cs, cst, u, v, ct, nt);
            if (!found) continue;
            if (useDepth) {
                float Ji[6] = {0}, Jj[6] = {0};
                const f3 diff = ct - cst;
                const float res = dot(diff, nt);
                const float w = s.p.weightDenseDepth * pw * powf(std::max(0.0f, 1.0f - ct.z / 2.0f), 2.5f);   // :256
                float jac[3][6];
                if (i > 0) { derivI(TjI, Ti, cs, jac); for (int k = 0; k < 6; ++k) Ji[k] = -(jac[0][k] * nt.x + jac[1][k] * nt.y + jac[2][k] * nt.z); }
                if (j > 0) { derivJ(TiI, Tj, cs, jac); for (int k = 0; k < 6; ++k) Jj[k] = -(jac[0][k] * nt.x + jac[1][k] * nt.y + jac[2][k] * nt.z); }
                addLocal(Ji, Jj, i, j, res, w);
            }
            if (useColor) {
                float dI[2], iT;
                bilinear<2>(u, v, s.cache[i].derivs2, s.W, s.H, dI);
                bilinear<1>(u, v, s.cache[i].intensity, s.W, s.H, &iT);
                const float res = iT - s.cache[j].intensity[idx];
                const bool ok = dI[0] != MINF && fabsf(res) < s.p.denseColorThresh && sqrtf(dI[0] * dI[0] + dI[1] * dI[1]) > s.p.denseColorGradientMin;
                if (!ok) continue;
                float Ji[6] = {0}, Jj[6] = {0};
                // dCameraToScreen (ICPUtil.h:14-26): 2x3
                const float z2 = cst.z * cst.z;
                const float P[2][3] = {{s.fx / cst.z, 0.0f, -s.fx * cst.x / z2}, {0.0f, s.fy / cst.z, -s.fy * cst.y / z2}};
                float jac[3][6];
                auto row = [&](float* J) {
                    for (int k = 0; k < 6; ++k) {
                        const float a0 = P[0][0] * jac[0][k] + P[0][1] * jac[1][k] + P[0][2] * jac[2][k];
                        const float a1 = P[1][0] * jac[0][k] + P[1][1] * jac[1][k] + P[1][2] * jac[2][k];
                        J[k] = dI[0] * a0 + dI[1] * a1;
                    }
                };
                if (i > 0) { derivI(TjI, Ti, cs, jac); row(Ji); }
                if (j > 0) { derivJ(TiI, Tj, cs, jac); row(Jj); }
                const float w = s.p.weightDenseColor * pw * std::max(0.0f, 1.0f - fabsf(res) / (1.15f * s.p.denseColorThresh));
                addLocal(Ji, Jj, i, j, res, w);
            }
        }
    }
    for (uint32_t y = 0; y < dim; ++y)       // FlipJtJ_Kernel :81-91
        for (uint32_t x = y + 1; x < dim; ++x) s.JtJ[(size_t)y * dim + x] = s.JtJ[(size_t)x * dim + y];
    return true;
}

// evalMinusJTFDevice, SolverBundlingEquationsLie.h:63-148
void evalMinusJTF(Solver& s, uint32_t v, f3& resRot, f3& resTrans) {
    f3 rRot = {0, 0, 0}, rTrans = {0, 0, 0}, pRot = {0, 0, 0}, pTrans = {0, 0, 0};
    s.deltaRot[v] = {0, 0, 0};
    s.deltaTrans[v] = {0, 0, 0};
    for (int ci : s.varToCorr[v]) {
        const EntryJ& c = s.corr[ci];
        if (!valid(c)) continue;
        const m44 &TI = s.T[c.i], &TJ = s.T[c.j];
        f3 worldP; float sign = 1;
        if (v != c.i) { sign = -1; worldP = xform(TJ, v3(c.pj)); } else worldP = xform(TI, v3(c.pi));
        const f3 da = dAlpha(worldP), db = dBeta(worldP), dc = dGamma(worldP);
        const f3 r = xform(TI, v3(c.pi)) - xform(TJ, v3(c.pj));
        rRot = rRot + mk3(dot(da, r), dot(db, r), dot(dc, r)) * sign;
        rTrans = rTrans + r * sign;
        pRot = pRot + mk3(dot(da, da), dot(db, db), dot(dc, dc));
        pTrans = pTrans + mk3(1.0f, 1.0f, 1.0f);
    }
    resRot = rRot * -s.p.weightSparse;
    resTrans = rTrans * -s.p.weightSparse;
    if (s.p.useDense) {
        resRot = resRot - mk3(s.Jtr[v * 6 + 3], s.Jtr[v * 6 + 4], s.Jtr[v * 6 + 5]);
        resTrans = resTrans - mk3(s.Jtr[v * 6 + 0], s.Jtr[v * 6 + 1], s.Jtr[v * 6 + 2]);
    }
    auto inv = [](float p) { return p > FLOAT_EPSILON ? 1.0f / p : 1.0f; };
    s.precRot[v] = {inv(pRot.x), inv(pRot.y), inv(pRot.z)};
    s.precTrans[v] = {inv(pTrans.x), inv(pTrans.y), inv(pTrans.z)};
}

inline f3 cmul(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }

// one Gauss-Newton iteration's linear solve; returns number of PCG iterations
int solveLinear(Solver& s, uint32_t nLin) {
    const uint32_t N = s.N;
    // Initialization :755-849
    float rDotz = 0.0f;
    for (uint32_t x = 1; x < N; ++x) {
        f3 rR, rT;
        evalMinusJTF(s, x, rR, rT);
        s.rRot[x] = rR; s.rTrans[x] = rT;
        s.pRot[x] = cmul(s.precRot[x], rR);
        s.pTrans[x] = cmul(s.precTrans[x], rT);
        rDotz += dot(rR, s.pRot[x]) + dot(rT, s.pTrans[x]);
        s.ApRot[x] = {0, 0, 0}; s.ApTrans[x] = {0, 0, 0};
    }
    for (uint32_t x = 1; x < N; ++x) s.rDotzOld[x] = rDotz;
    int it = 0;
    for (uint32_t lin = 0; lin < nLin; ++lin) {
        bool last = (lin == nLin - 1);
        ++it;
        if (s.p.weightSparse > 0.0f) {
            // PCGStep_Kernel0 (applyJDevice :195-228)
            for (uint32_t c = 0; c < s.C; ++c) {
                f3 b = {0, 0, 0};
                const EntryJ& e = s.corr[c];
                if (valid(e)) {
                    if (e.i > 0) {
                        const f3 w = xform(s.T[e.i], v3(e.pi));
                        const f3 pp = s.pRot[e.i];
                        b = b + (dAlpha(w) * pp.x + dBeta(w) * pp.y + dGamma(w) * pp.z + s.pTrans[e.i]);
                    }
                    if (e.j > 0) {
                        const f3 w = xform(s.T[e.j], v3(e.pj));
                        const f3 pp = s.pRot[e.j];
                        b = b - (dAlpha(w) * pp.x + dBeta(w) * pp.y + dGamma(w) * pp.z + s.pTrans[e.j]);
                    }
                    b = b * s.p.weightSparse;
                }
                s.Jp[c] = b;
            }
            // PCGStep_Kernel1a (applyJTDevice :154-193)
            for (uint32_t x = 1; x < N; ++x) {
                f3 oR = {0, 0, 0}, oT = {0, 0, 0};
                for (int ci : s.varToCorr[x]) {
                    const EntryJ& e = s.corr[ci];
                    if (!valid(e)) continue;
                    f3 w; float sign = 1;
                    if (x != e.i) { sign = -1; w = xform(s.T[e.j], v3(e.pj)); } else w = xform(s.T[e.i], v3(e.pi));
                    const f3 jp = s.Jp[ci];
                    oR = oR + mk3(dot(dAlpha(w), jp), dot(dBeta(w), jp), dot(dGamma(w), jp)) * sign;
                    oT = oT + jp * sign;
                }
                s.ApRot[x] = s.ApRot[x] + oR;
                s.ApTrans[x] = s.ApTrans[x] + oT;
            }
        }
        if (s.p.useDense) {   // applyJTJDenseDevice, DenseUtil :371-411
            const uint32_t dim = 6 * N;
            for (uint32_t x = 1; x < N; ++x) {
                f3 oR = {0, 0, 0}, oT = {0, 0, 0};
                for (uint32_t i = 1; i < N; ++i) {
                    const float* A = s.JtJ.data();
                    auto blk = [&](int r0, int c0, f3 v) {
                        f3 o;
                        o.x = A[(size_t)(x * 6 + r0 + 0) * dim + i * 6 + c0 + 0] * v.x + A[(size_t)(x * 6 + r0 + 0) * dim + i * 6 + c0 + 1] * v.y + A[(size_t)(x * 6 + r0 + 0) * dim + i * 6 + c0 + 2] * v.z;
                        o.y = A[(size_t)(x * 6 + r0 + 1) * dim + i * 6 + c0 + 0] * v.x + A[(size_t)(x * 6 + r0 + 1) * dim + i * 6 + c0 + 1] * v.y + A[(size_t)(x * 6 + r0 + 1) * dim + i * 6 + c0 + 2] * v.z;
                        o.z = A[(size_t)(x * 6 + r0 + 2) * dim + i * 6 + c0 + 0] * v.x + A[(size_t)(x * 6 + r0 + 2) * dim + i * 6 + c0 + 1] * v.y + A[(size_t)(x * 6 + r0 + 2) * dim + i * 6 + c0 + 2] * v.z;
                        return o;
                    };
                    oT = oT + (blk(0, 0, s.pTrans[i]) + blk(0, 3, s.pRot[i]));
                    oR = oR + (blk(3, 0, s.pTrans[i]) + blk(3, 3, s.pRot[i]));
                }
                s.ApRot[x] = s.ApRot[x] + oR;
                s.ApTrans[x] = s.ApTrans[x] + oT;
            }
        }
        // PCGStep_Kernel1b :930-946
        float pAp = 0.0f;
        for (uint32_t x = 1; x < N; ++x) pAp += dot(s.pRot[x], s.ApRot[x]) + dot(s.pTrans[x], s.ApTrans[x]);
        // PCGStep_Kernel2 :948-983
        float rzNew = 0.0f;
        for (uint32_t x = 1; x < N; ++x) {
            float alpha = 0.0f;
            if (pAp > FLOAT_EPSILON) alpha = s.rDotzOld[x] / pAp;
            s.deltaRot[x] = s.deltaRot[x] + s.pRot[x] * alpha;
            s.deltaTrans[x] = s.deltaTrans[x] + s.pTrans[x] * alpha;
            s.rRot[x] = s.rRot[x] - s.ApRot[x] * alpha;
            s.rTrans[x] = s.rTrans[x] - s.ApTrans[x] * alpha;
            s.zRot[x] = cmul(s.precRot[x], s.rRot[x]);
            s.zTrans[x] = cmul(s.precTrans[x], s.rTrans[x]);
            rzNew += dot(s.zRot[x], s.rRot[x]) + dot(s.zTrans[x], s.rTrans[x]);
        }
        if (fabsf(pAp) < 5e-7f) last = true;     // :1088-1093 (host reads d_scanAlpha[0] == p.Ap)
        // PCGStep_Kernel3 :985-1022
        for (uint32_t x = 1; x < N; ++x) {
            const float old = s.rDotzOld[x];
            float beta = 0.0f;
            if (old > FLOAT_EPSILON) beta = rzNew / old;
            s.rDotzOld[x] = rzNew;
            s.pRot[x] = s.zRot[x] + s.pRot[x] * beta;
            s.pTrans[x] = s.zTrans[x] + s.pTrans[x] * beta;
            s.ApRot[x] = {0, 0, 0}; s.ApTrans[x] = {0, 0, 0};
            if (last) {
                f3 r, t;
                computeLieUpdate(s.deltaRot[x], s.deltaTrans[x], s.xRot[x], s.xTrans[x], r, t);
                s.xRot[x] = r; s.xTrans[x] = t;
            }
        }
        if (last) break;
    }
    return it;
}

float evalResidual(const Solver& s) {     // EvalResidualDevice :576-593 (poseToMatrix of the current x)
    float sum = 0.0f;
    for (uint32_t c = 0; c < s.C; ++c) {
        const EntryJ& e = s.corr[c];
        if (!valid(e)) continue;
        const m44 TI = poseToMatrix(s.xRot[e.i], s.xTrans[e.i]), TJ = poseToMatrix(s.xRot[e.j], s.xTrans[e.j]);
        const f3 r = xform(TI, v3(e.pi)) - xform(TJ, v3(e.pj));
        sum += s.p.weightSparse * dot(r, r);
    }
    return sum;
}

float absMaxResidual(const Solver& s, const EntryJ& e, float w) {   // evalAbsMaxResidualDevice :27-40
    if (!valid(e)) return 0.0f;
    const m44 TI = poseToMatrix(s.xRot[e.i], s.xTrans[e.i]), TJ = poseToMatrix(s.xRot[e.j], s.xTrans[e.j]);
    const f3 a = xform(TI, v3(e.pi)), b = xform(TJ, v3(e.pj));
    const f3 r = {w * fabsf(a.x - b.x), w * fabsf(a.y - b.y), w * fabsf(a.z - b.z)};
    return std::max(r.z, std::max(r.x, r.y));
}

}  // namespace

extern "C" {

// SBA.cu:75-108 with the Lie-space helpers (GlobalDefines.h:12 USE_LIE_SPACE)
void or_matrices_to_poses(const float* T16, uint32_t n, float* rot3, float* trans3, const int* validImages) {
    for (uint32_t i = 0; i < n; ++i) {
        if (!validImages[i]) continue;
        m44 M; memcpy(M.e, T16 + 16 * i, 64);
        f3 r, t;
        matrixToPose(M, r, t);
        rot3[3 * i] = r.x; rot3[3 * i + 1] = r.y; rot3[3 * i + 2] = r.z;
        trans3[3 * i] = t.x; trans3[3 * i + 1] = t.y; trans3[3 * i + 2] = t.z;
    }
}
void or_poses_to_matrices(const float* rot3, const float* trans3, uint32_t n, float* T16, const int* validImages) {
    for (uint32_t i = 0; i < n; ++i) {
        if (!validImages[i]) continue;
        const m44 M = poseToMatrix(v3(rot3 + 3 * i), v3(trans3 + 3 * i));
        memcpy(T16 + 16 * i, M.e, 64);
    }
}

struct or_solver_args {
    void* corr; uint32_t numCorr; const int* validImages; uint32_t numImages; uint32_t maxCorrPerImage;
    uint32_t nNonLin, nLin;
    const void* cacheFrames;   // array of CacheFrame (6 host pointers each) or NULL
    uint32_t W, H; float fx, fy, cx, cy;
    const float* weightsSparse; const float* weightsDenseDepth; const float* weightsDenseColor;
    int usePairwise;
    float denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax;
    uint32_t denseOverlapCheckSubsampleFactor;
    float* rot3; float* trans3;
    // outputs
    float* convergence;        // nNonLin+1 energies (sparse term, EvalResidual) or NULL
    int* pcgIterations;        // per GN iteration, nNonLin entries, or NULL
    int* gnIterations;
    float* maxResidual; int* maxResidualIndex;
    float* denseJtJ; float* denseJtr;   // optional dumps of the LAST built dense system (6N x 6N, 6N)
    int* numDensePairs;
};

// solveBundlingStub :1137-1220 + CUDASolverBundling::solve :187-284
void or_solver_solve(or_solver_args* a) {
    Solver s;
    s.N = a->numImages; s.C = a->numCorr; s.maxCorrPerImage = a->maxCorrPerImage;
    s.corr = (EntryJ*)a->corr; s.validImages = a->validImages;
    if (a->cacheFrames) {
        const CacheFrame* cf = (const CacheFrame*)a->cacheFrames;
        s.cache.assign(cf, cf + s.N);
        s.W = a->W; s.H = a->H; s.fx = a->fx; s.fy = a->fy; s.cx = a->cx; s.cy = a->cy;
    }
    s.p.denseDistThresh = a->denseDistThresh; s.p.denseNormalThresh = a->denseNormalThresh; s.p.denseColorThresh = a->denseColorThresh;
    s.p.denseColorGradientMin = a->denseColorGradientMin; s.p.denseDepthMin = a->denseDepthMin; s.p.denseDepthMax = a->denseDepthMax;
    s.p.denseOverlapCheckSubsampleFactor = a->denseOverlapCheckSubsampleFactor;
    s.p.usePairwise = a->usePairwise != 0;
    const uint32_t N = s.N;
    s.xRot.resize(N); s.xTrans.resize(N);
    for (uint32_t i = 0; i < N; ++i) { s.xRot[i] = v3(a->rot3 + 3 * i); s.xTrans[i] = v3(a->trans3 + 3 * i); }
    for (auto* v : {&s.deltaRot, &s.deltaTrans, &s.rRot, &s.rTrans, &s.zRot, &s.zTrans, &s.pRot, &s.pTrans, &s.ApRot, &s.ApTrans, &s.precRot, &s.precTrans})
        v->assign(N, f3{0, 0, 0});
    s.Jp.assign(std::max<uint32_t>(s.C, 1), f3{0, 0, 0});
    s.rDotzOld.assign(N, 0.0f);
    s.T.resize(N); s.Tinv.resize(N);
    buildVarTable(s);
    s.p.weightSparse = a->weightsSparse[0];
    if (a->convergence) a->convergence[0] = evalResidual(s);
    int gn = 0;
    for (uint32_t it = 0; it < a->nNonLin; ++it) {
        s.p.weightSparse = a->weightsSparse[it];
        s.p.weightDenseDepth = a->weightsDenseDepth[it];
        s.p.weightDenseColor = a->weightsDenseColor[it];
        s.p.useDense = (s.p.weightDenseDepth > 0 || s.p.weightDenseColor > 0) && !s.cache.empty();
        for (uint32_t i = 0; i < N; ++i) { s.T[i] = poseToMatrix(s.xRot[i], s.xTrans[i]); s.Tinv[i] = inverse(s.T[i]); }   // :1114-1121
        if (s.p.useDense) s.p.useDense = buildDenseSystem(s);
        const int pcg = solveLinear(s, a->nLin);
        if (a->pcgIterations) a->pcgIterations[it] = pcg;
        if (a->convergence) a->convergence[it + 1] = evalResidual(s);
        ++gn;
        if (it + 1 < a->nNonLin) {            // EvalGNConvergence :694-749
            float mx = 0.0f;
            for (uint32_t x = 1; x < N; ++x) {
                if (s.validImages[x] == 0) continue;
                const f3 d = s.deltaRot[x], t = s.deltaTrans[x];
                const float r = std::max(std::max(std::max(fabsf(d.x), fabsf(t.x)), std::max(fabsf(d.y), fabsf(t.y))), std::max(fabsf(d.z), fabsf(t.z)));
                mx = std::max(mx, r);
            }
            if (mx < 0.005f) break;
        }
    }
    if (a->gnIterations) *a->gnIterations = gn;
    for (uint32_t i = 0; i < N; ++i) {
        a->rot3[3 * i] = s.xRot[i].x; a->rot3[3 * i + 1] = s.xRot[i].y; a->rot3[3 * i + 2] = s.xRot[i].z;
        a->trans3[3 * i] = s.xTrans[i].x; a->trans3[3 * i + 1] = s.xTrans[i].y; a->trans3[3 * i + 2] = s.xTrans[i].z;
    }
    if (a->maxResidual) {     // computeMaxResidual, CUDASolverBundling.cpp:313-427 (first maximum in index order)
        float mx = 0.0f; int mi = 0;
        if (s.p.weightSparse > 0.0f)
            for (uint32_t c = 0; c < s.C; ++c) { const float r = absMaxResidual(s, s.corr[c], s.p.weightSparse); if (mx < r) { mx = r; mi = (int)c; } }
        *a->maxResidual = mx; *a->maxResidualIndex = mi;
    }
    if (a->denseJtJ && !s.JtJ.empty()) memcpy(a->denseJtJ, s.JtJ.data(), s.JtJ.size() * 4);
    if (a->denseJtr && !s.Jtr.empty()) memcpy(a->denseJtr, s.Jtr.data(), s.Jtr.size() * 4);
    if (a->numDensePairs) *a->numDensePairs = (int)s.pairs.size();
}

// CUDASolverBundling::useVerification :454-476 (the reference reads an uninitialised weightSparse;
// every sparse weight in SBA.cpp:28-38 is 1.0, which is what is used here)
int or_solver_use_verification(const void* corr, uint32_t numCorr, const float* rot3, const float* trans3, uint32_t numImages,
                               float verifyOptDistThresh, float verifyOptPercentThresh) {
    Solver s;
    s.N = numImages; s.C = numCorr; s.corr = (EntryJ*)corr;
    s.xRot.resize(numImages); s.xTrans.resize(numImages);
    for (uint32_t i = 0; i < numImages; ++i) { s.xRot[i] = v3(rot3 + 3 * i); s.xTrans[i] = v3(trans3 + 3 * i); }
    uint32_t high = 0;
    for (uint32_t c = 0; c < numCorr; ++c) if (absMaxResidual(s, s.corr[c], 1.0f) > verifyOptDistThresh) high++;
    return ((float)high / (float)numCorr >= verifyOptPercentThresh) ? 1 : 0;
}

void or_pose_to_matrix(const float* rot3, const float* trans3, float* T16) { const m44 M = poseToMatrix(v3(rot3), v3(trans3)); memcpy(T16, M.e, 64); }
void or_matrix_to_pose(const float* T16, float* rot3, float* trans3) {
    m44 M; memcpy(M.e, T16, 64);
    f3 r, t; matrixToPose(M, r, t);
    rot3[0] = r.x; rot3[1] = r.y; rot3[2] = r.z; trans3[0] = t.x; trans3[1] = t.y; trans3[2] = t.z;
}

}  // extern "C"
