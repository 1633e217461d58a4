// TEST INFRASTRUCTURE ONLY (see or_common.h) - CPU restatement of the reference's ray cast of the voxel hash.
//
// Follows  DepthSensing/CUDARayCastSDF.cu:17-48 (renderKernel), :86-160 (rayIntervalSplatKernel),
//          DepthSensing/RayCastSDFUtil.h:86-283 (trilinearInterpolationSimpleFastFast with colour, findIntersectionLinear / Bisection,
//          gradientForPoint, cameraToDepthProj, depthProjToCameraZ, traverseCoarseGridSimpleSampleAll),
//          DepthSensing/CameraUtil.cu:665-693 (computeNormalsDevice),
//          DepthSensing/DX11RayIntervalSplatting.cpp:137-216 + Shaders/RayIntervalSplatting.hlsl for what the two draw calls leave in
//          the min / max render targets (nearest / farthest block's camera depth per pixel, -inf where no block projects).
// or_rc_render is pinned against the reference's renderKernel (oracle/_ref, tests/test_ref_pin_cpu.py) on the same interval images;
// the splat replaces a D3D11 rasteriser pass and is a definition of this repository (coverage = pixel centres inside the rectangle).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../include/bf_hip.h"
#include "or_common.h"
#include "or_volume.h"

using namespace orc;

namespace {

const float NINF = -std::numeric_limits<float>::infinity();

struct Rc {
    Vol v;
    const bf_ray_cast_params* p;
};

inline float frac1(float x) { return x - floorf(x); }

bool trilinear(const Vol& v, f3 pos, float& dist, uint8_t col[3]) {
    const float oSet = v.voxelSize;
    const f3 posDual = pos - mk3(oSet / 2.0f, oSet / 2.0f, oSet / 2.0f);
    const f3 pv = pos / v.voxelSize;
    const float wx = frac1(pv.x), wy = frac1(pv.y), wz = frac1(pv.z);
    dist = 0.0f;
    float c[3] = {0.0f, 0.0f, 0.0f};
    const float off[8][3] = {{0, 0, 0}, {oSet, 0, 0}, {0, oSet, 0}, {0, 0, oSet}, {oSet, oSet, 0}, {0, oSet, oSet}, {oSet, 0, oSet}, {oSet, oSet, oSet}};
    const float w[8] = {(1.0f - wx) * (1.0f - wy) * (1.0f - wz), wx * (1.0f - wy) * (1.0f - wz), (1.0f - wx) * wy * (1.0f - wz), (1.0f - wx) * (1.0f - wy) * wz,
                        wx * wy * (1.0f - wz), (1.0f - wx) * wy * wz, wx * (1.0f - wy) * wz, wx * wy * wz};
    for (int k = 0; k < 8; ++k) {
        const Vx s = getVoxel(v, posDual + mk3(off[k][0], off[k][1], off[k][2]));
        if (s.weight == 0) return false;
        dist += w[k] * s.sdf;
        for (int j = 0; j < 3; ++j) c[j] += w[k] * (float)s.c[j];
    }
    for (int j = 0; j < 3; ++j) col[j] = (uint8_t)(f2u(c[j]) & 0xFF);
    return true;
}

bool bisection(const Vol& v, f3 camPos, f3 dir, float d0, float r0, float d1, float r1, float& alpha, uint8_t col[3]) {
    float a = r0, aDist = d0, b = r1, bDist = d1, c = 0.0f;
    for (int i = 0; i < 3; ++i) {
        c = a + (aDist / (aDist - bDist)) * (b - a);
        float cDist;
        if (!trilinear(v, camPos + dir * c, cDist, col)) return false;
        if (aDist * cDist > 0.0f) { a = c; aDist = cDist; }
        else { b = c; bDist = cDist; }
    }
    alpha = c;
    return true;
}

f3 gradientForPoint(const Vol& v, f3 pos) {
    const float vs = v.voxelSize;
    float d[6] = {0, 0, 0, 0, 0, 0}; uint8_t col[3];
    trilinear(v, pos - mk3(0.5f * vs, 0.0f, 0.0f), d[0], col);
    trilinear(v, pos - mk3(0.0f, 0.5f * vs, 0.0f), d[1], col);
    trilinear(v, pos - mk3(0.0f, 0.0f, 0.5f * vs), d[2], col);
    trilinear(v, pos + mk3(0.5f * vs, 0.0f, 0.0f), d[3], col);
    trilinear(v, pos + mk3(0.0f, 0.5f * vs, 0.0f), d[4], col);
    trilinear(v, pos + mk3(0.0f, 0.0f, 0.5f * vs), d[5], col);
    const f3 g = mk3((d[0] - d[3]) / vs, (d[1] - d[4]) / vs, (d[2] - d[5]) / vs);
    const float l = sqrtf(dot(g, g));
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(-g.x / l, -g.y / l, -g.z / l);
}

inline f3 normalize3(f3 v) { const float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }

m44 toM(const float* p) { m44 m; memcpy(&m, p, 64); return m; }

f3 cameraToDepthProj(const bf_ray_cast_params& p, f3 pos) {
    const float px = pos.x * p.fx / pos.z + p.mx, py = pos.y * p.fy / pos.z + p.my;
    f3 r;
    r.x = (2.0f * px - ((float)p.m_width - 1.0f)) / ((float)p.m_width - 1.0f);
    r.y = (((float)p.m_height - 1.0f) - 2.0f * py) / ((float)p.m_height - 1.0f);
    r.z = (pos.z - p.m_minDepth) / (p.m_maxDepth - p.m_minDepth);
    return r;
}

}  // namespace

extern "C" {

// the two render targets after DX11RayIntervalSplatting::rayIntervalSplatting
int or_rc_splat(const bf_hash_entry* compact, uint32_t numOccupied, const bf_hash_params* hp, const bf_depth_camera_params* cam, const bf_ray_cast_params* p,
                float* rayMin, float* rayMax) {
    const uint32_t W = p->m_width, H = p->m_height;
    for (uint32_t i = 0; i < W * H; ++i) { rayMin[i] = NINF; rayMax[i] = NINF; }
    const m44 view = toM(p->m_viewMatrix), rigidInv = toM(hp->m_rigidTransformInverse);
    const float vs = hp->m_virtualVoxelSize;
    for (uint32_t idx = 0; idx < numOccupied; ++idx) {
        const bf_hash_entry& e = compact[idx];
        if (e.ptr == BF_FREE_ENTRY) continue;
        {   // isSDFBlockInCameraFrustumApprox
            f3 w = mk3((float)(e.pos[0] * VBS), (float)(e.pos[1] * VBS), (float)(e.pos[2] * VBS)) * vs;
            const float off = vs * 0.5f * ((float)VBS - 1.0f);
            w = w + mk3(off, off, off);
            const f3 pc = xform(rigidInv, w);
            const float sx = pc.x * cam->fx / pc.z + cam->mx, sy = pc.y * cam->fy / pc.z + cam->my;
            const float wm1 = (float)cam->m_imageWidth - 1.0f, hm1 = (float)cam->m_imageHeight - 1.0f;
            float px = (2.0f * sx - wm1) / wm1, py = (hm1 - 2.0f * sy) / hm1;
            float pz = (pc.z - cam->m_sensorDepthWorldMin) / (cam->m_sensorDepthWorldMax - cam->m_sensorDepthWorldMin);
            px *= 0.95f; py *= 0.95f; pz *= 0.95f;
            if (px < -1.0f || px > 1.0f || py < -1.0f || py > 1.0f || pz < 0.0f || pz > 1.0f) continue;
        }
        const f3 wv = mk3((float)(e.pos[0] * VBS), (float)(e.pos[1] * VBS), (float)(e.pos[2] * VBS)) * vs;
        const float h = vs / 2.0f, ext = (float)VBS * vs;
        const f3 mn = mk3(wv.x - h, wv.y - h, wv.z - h), mx = mk3(mn.x + ext, mn.y + ext, mn.z + ext);
        f3 lo = mk3(INFINITY, INFINITY, INFINITY), hi = mk3(-INFINITY, -INFINITY, -INFINITY);
        for (int c = 0; c < 8; ++c) {
            const f3 q = cameraToDepthProj(*p, xform(view, mk3((c & 1) ? mx.x : mn.x, (c & 2) ? mx.y : mn.y, (c & 4) ? mx.z : mn.z)));
            lo.x = fminf(lo.x, q.x); lo.y = fminf(lo.y, q.y); lo.z = fminf(lo.z, q.z);
            hi.x = fmaxf(hi.x, q.x); hi.y = fmaxf(hi.y, q.y); hi.z = fmaxf(hi.z, q.z);
        }
        const float dMin = lo.z * (p->m_maxDepth - p->m_minDepth) + p->m_minDepth, dMax = hi.z * (p->m_maxDepth - p->m_minDepth) + p->m_minDepth;
        const float X0 = (lo.x * 0.5f + 0.5f) * (float)W, X1 = (hi.x * 0.5f + 0.5f) * (float)W;
        const float Y0 = (1.0f - (hi.y * 0.5f + 0.5f)) * (float)H, Y1 = (1.0f - (lo.y * 0.5f + 0.5f)) * (float)H;
        if (!(X0 < X1) || !(Y0 < Y1)) continue;
        for (uint32_t y = 0; y < H; ++y) {
            const float cy = (float)y + 0.5f;
            if (!(Y0 <= cy && cy < Y1)) continue;
            for (uint32_t x = 0; x < W; ++x) {
                const float cx = (float)x + 0.5f;
                if (!(X0 <= cx && cx < X1)) continue;
                float& a = rayMin[y * W + x]; float& b = rayMax[y * W + x];
                if (a == NINF || dMin < a) a = dMin;
                if (b == NINF || dMax > b) b = dMax;
            }
        }
    }
    return 0;
}

// renderKernel over the whole image (+ computeNormals when gradients are off), from given interval images
int or_rc_render(const bf_hash_entry* hash, const bf_voxel* vox, const bf_hash_params* hp, const bf_ray_cast_params* p, const float* rayMin, const float* rayMax,
                 float* depth, float* depth4, float* normals, float* colors) {
    Vol v = {hash, vox, hp->m_hashNumBuckets, hp->m_hashMaxCollisionLinkedListSize, hp->m_virtualVoxelSize};
    const uint32_t W = p->m_width, H = p->m_height;
    const m44 view = toM(p->m_viewMatrix), viewInv = toM(p->m_viewMatrixInverse);
#pragma omp parallel for schedule(dynamic, 4)
    for (int yy = 0; yy < (int)H; ++yy)
        for (uint32_t x = 0; x < W; ++x) {
            const uint32_t y = (uint32_t)yy, px = y * W + x;
            depth[px] = NINF;
            for (int k = 0; k < 4; ++k) { depth4[4 * px + k] = NINF; normals[4 * px + k] = NINF; colors[4 * px + k] = NINF; }
            const float cx = ((float)x - p->mx) / p->fx, cy = ((float)y - p->my) / p->fy;
            const f3 camDir = normalize3(mk3(1.0f * cx, 1.0f * cy, 1.0f));
            const f3 worldCamPos = xform(viewInv, mk3(0.0f, 0.0f, 0.0f));
            const f3 wd = mk3(viewInv.e[0] * camDir.x + viewInv.e[1] * camDir.y + viewInv.e[2] * camDir.z + viewInv.e[3] * 0.0f,
                              viewInv.e[4] * camDir.x + viewInv.e[5] * camDir.y + viewInv.e[6] * camDir.z + viewInv.e[7] * 0.0f,
                              viewInv.e[8] * camDir.x + viewInv.e[9] * camDir.y + viewInv.e[10] * camDir.z + viewInv.e[11] * 0.0f);
            const f3 worldDir = normalize3(wd);
            float minInterval = rayMin[px], maxInterval = rayMax[px];
            if (minInterval == 0.0f || minInterval == NINF) continue;
            if (maxInterval == 0.0f || maxInterval == NINF) continue;
            minInterval = fmaxf(minInterval, p->m_minDepth);
            maxInterval = fminf(maxInterval, p->m_maxDepth);
            float lastSdf = 0.0f, lastAlpha = 0.0f; uint32_t lastWeight = 0;
            const float depthToRayLength = 1.0f / camDir.z;
            float rayCurrent = depthToRayLength * fmaxf(p->m_minDepth, minInterval);
            const float rayEnd = depthToRayLength * fminf(p->m_maxDepth, maxInterval);
            while (rayCurrent < rayEnd) {
                const f3 cur = worldCamPos + worldDir * rayCurrent;
                float dist; uint8_t col[3];
                if (trilinear(v, cur, dist, col)) {
                    if (lastWeight > 0 && lastSdf > 0.0f && dist < 0.0f) {
                        float alpha = 0.0f; uint8_t col2[3] = {0, 0, 0};
                        const bool ok = bisection(v, worldCamPos, worldDir, lastSdf, lastAlpha, dist, rayCurrent, alpha, col2);
                        const f3 iso = worldCamPos + worldDir * alpha;
                        if (ok && fabsf(lastSdf - dist) < p->m_thresSampleDist && fabsf(dist) < p->m_thresDist) {
                            const float d = alpha / depthToRayLength;
                            depth[px] = d;
                            depth4[4 * px] = d * cx; depth4[4 * px + 1] = d * cy; depth4[4 * px + 2] = d; depth4[4 * px + 3] = 1.0f;
                            colors[4 * px] = (float)col2[0] / 255.f; colors[4 * px + 1] = (float)col2[1] / 255.f; colors[4 * px + 2] = (float)col2[2] / 255.f; colors[4 * px + 3] = 1.0f;
                            if (p->m_useGradients) {
                                const f3 g = gradientForPoint(v, iso);
                                const f3 n = mk3(-g.x, -g.y, -g.z);
                                normals[4 * px] = view.e[0] * n.x + view.e[1] * n.y + view.e[2] * n.z + view.e[3] * 0.0f;
                                normals[4 * px + 1] = view.e[4] * n.x + view.e[5] * n.y + view.e[6] * n.z + view.e[7] * 0.0f;
                                normals[4 * px + 2] = view.e[8] * n.x + view.e[9] * n.y + view.e[10] * n.z + view.e[11] * 0.0f;
                                normals[4 * px + 3] = 1.0f;
                            }
                            break;
                        }
                    }
                    lastSdf = dist; lastAlpha = rayCurrent; lastWeight = 1;
                    rayCurrent += p->m_rayIncrement;
                } else {
                    lastWeight = 0;
                    rayCurrent += p->m_rayIncrement;
                }
            }
        }
    if (!p->m_useGradients) {                                                     // computeNormalsDevice
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                float* o = normals + 4 * (y * W + x);
                o[0] = o[1] = o[2] = o[3] = NINF;
                if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
                    const float *CC = depth4 + 4 * (y * W + x), *PC = depth4 + 4 * ((y + 1) * W + x), *CP = depth4 + 4 * (y * W + x + 1);
                    const float *MC = depth4 + 4 * ((y - 1) * W + x), *CM = depth4 + 4 * (y * W + x - 1);
                    if (CC[0] != NINF && PC[0] != NINF && CP[0] != NINF && MC[0] != NINF && CM[0] != NINF) {
                        const f3 u = mk3(PC[0] - MC[0], PC[1] - MC[1], PC[2] - MC[2]), w = mk3(CP[0] - CM[0], CP[1] - CM[1], CP[2] - CM[2]);
                        const f3 n = mk3(u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x);
                        const float l = sqrtf(dot(n, n));
                        if (l > 0.0f) { const float nl = -l; o[0] = n.x / nl; o[1] = n.y / nl; o[2] = n.z / nl; o[3] = 1.0f; }
                    }
                }
            }
    }
    return 0;
}

}  // extern "C"
