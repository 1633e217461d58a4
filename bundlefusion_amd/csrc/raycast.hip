// Ray cast of the voxel-hash volume for gfx950.  Replaces DepthSensing/CUDARayCastSDF.{h,cpp,cu},
// RayCastSDFUtil.h, CUDARayCastParams.h and DX11RayIntervalSplatting.{h,cpp} + Shaders/RayIntervalSplatting.hlsl behind the
// bf_ray_cast_* C ABI (paths relative to /root/reference/FriedLiver/Source).
//
// A CONSUMER of the volume: it reads the raw arrays bf_scene_get_hash_data() hands out in the reference layout, the frustum list
// (d_hashCompactified, m_numOccupiedBlocks) and HashParams - what CUDARayCastSDF::render reads.
//
//  * Ray-interval splatting.  The reference writes two triangles per SDF block into a D3D11 vertex buffer (rayIntervalSplatKernel,
//    CUDARayCastSDF.cu:86-160) and lets the rasteriser keep, per pixel, the nearest (depth test LESS) / farthest (GREATER) block's
//    camera-space depth: render targets cleared to -inf (DX11CustomRenderTarget.cpp:204-216), no depth clip, pixel shader = the
//    vertex's w (RayIntervalSplatting.hlsl).  Here one wave per block projects the eight corners with the reference's arithmetic and
//    folds the screen rectangle into two images with 32-bit atomic min / max on order-preserving float keys: one compute launch, no
//    graphics interop, the same interval per pixel (the depth test compares the projected depth, which is monotone in the camera
//    depth that is written; where the test clamps to [0,1] the written values lie outside [m_minDepth, m_maxDepth] and renderKernel
//    clamps them to the same bound).  Coverage follows the D3D11 rule - a pixel belongs to the rectangle when its centre (i + 0.5)
//    lies in [left, right) x [top, bottom) - without the rasteriser's 1/256-pixel vertex snapping.
//  * renderKernel / traverseCoarseGridSimpleSampleAll (CUDARayCastSDF.cu:17-60, RayCastSDFUtil.h:222-283): one thread per pixel,
//    the reference's arithmetic operation by operation (ray stepping, 3 bisection steps, trilinear samples of 8 voxels each, optional
//    analytic gradient); -ffp-contract=off keeps every product and sum separately rounded.  normalize() is v * (1 / sqrt(v.v)).
//  * computeNormals (CameraUtil.cu:665-693) and convertDepthFloatToCameraSpaceFloat4 (:386-403) as in the reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "bf_device.h"
#include "bf_internal.h"
#include "bf_volume.h"

using namespace bf;

namespace {

struct RcArgs {
    Vol v;
    const bf_hash_entry* compact; uint32_t numOccupied;
    m44 view, viewInv;              // m_viewMatrix (world -> camera), m_viewMatrixInverse
    m44 rigidInv;                   // c_hashParams.m_rigidTransformInverse (frustum test of the splat)
    bf_depth_camera_params cam;     // c_depthCameraParams (frustum test)
    float mx, my, fx, fy;
    uint32_t W, H;
    float minDepth, maxDepth, rayIncrement, thresSampleDist, thresDist;
    int useGradients;
    uint32_t* minKey; uint32_t* maxKey;
    float *rayMin, *rayMax;
    float* depth; float4* depth4; float4* normals; float4* colors;
};

// order-preserving map float -> uint32 (all finite values and infinities)
BF_DEV uint32_t orderKey(float x) { const uint32_t b = __float_as_uint(x); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
BF_DEV float orderVal(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
constexpr uint32_t KEY_NONE_MIN = 0xFFFFFFFFu, KEY_NONE_MAX = 0u;

__global__ void k_rc_clear(RcArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.W * a.H) { a.minKey[i] = KEY_NONE_MIN; a.maxKey[i] = KEY_NONE_MAX; }
}

// RayCastData::cameraToDepthProj, RayCastSDFUtil.h:207-220
BF_DEV f3 cameraToDepthProj(const RcArgs& a, f3 pos) {
    const float px = pos.x * a.fx / pos.z + a.mx, py = pos.y * a.fy / pos.z + a.my;
    f3 r;
    r.x = (2.0f * px - ((float)a.W - 1.0f)) / ((float)a.W - 1.0f);
    r.y = (((float)a.H - 1.0f) - 2.0f * py) / ((float)a.H - 1.0f);
    r.z = (pos.z - a.minDepth) / (a.maxDepth - a.minDepth);
    return r;
}

// isSDFBlockInCameraFrustumApprox, VoxelUtilHashSDF.h:322-326 + DepthCameraUtil.h:97-142
BF_DEV bool blockInFrustumRc(const RcArgs& a, i3 b) {
    f3 w = mk3((float)(b.x * VOL_BS), (float)(b.y * VOL_BS), (float)(b.z * VOL_BS)) * a.v.voxelSize;
    const float off = a.v.voxelSize * 0.5f * ((float)VOL_BS - 1.0f);
    w = w + mk3(off, off, off);
    const f3 pc = xform(a.rigidInv, w);
    const float sx = pc.x * a.cam.fx / pc.z + a.cam.mx, sy = pc.y * a.cam.fy / pc.z + a.cam.my;
    const float wm1 = (float)a.cam.m_imageWidth - 1.0f, hm1 = (float)a.cam.m_imageHeight - 1.0f;
    float px = (2.0f * sx - wm1) / wm1, py = (hm1 - 2.0f * sy) / hm1;
    float pz = (pc.z - a.cam.m_sensorDepthWorldMin) / (a.cam.m_sensorDepthWorldMax - a.cam.m_sensorDepthWorldMin);
    px *= 0.95f; py *= 0.95f; pz *= 0.95f;
    return !(px < -1.0f || px > 1.0f || py < -1.0f || py > 1.0f || pz < 0.0f || pz > 1.0f);
}

// rayIntervalSplatKernel (CUDARayCastSDF.cu:86-160) + the two draw calls of DX11RayIntervalSplatting::rayIntervalSplatting: one wave per block
__global__ __launch_bounds__(256) void k_rc_splat(RcArgs a) {
    const uint32_t idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= a.numOccupied) return;
    const int4 e = reinterpret_cast<const int4*>(a.compact)[(size_t)idx * 2];
    if (e.w == BF_FREE_ENTRY) return;
    i3 b; b.x = e.x; b.y = e.y; b.z = e.z;
    if (!blockInFrustumRc(a, b)) return;
    const f3 wv = mk3((float)(b.x * VOL_BS), (float)(b.y * VOL_BS), (float)(b.z * VOL_BS)) * a.v.voxelSize;
    const float h = a.v.voxelSize / 2.0f;
    const f3 mn = mk3(wv.x - h, wv.y - h, wv.z - h);
    const float ext = (float)VOL_BS * a.v.voxelSize;
    const f3 mxv = mk3(mn.x + ext, mn.y + ext, mn.z + ext);
    f3 lo = mk3(BF_PINF, BF_PINF, BF_PINF), hi = mk3(BF_MINF, BF_MINF, BF_MINF);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f3 p = cameraToDepthProj(a, xform(a.view, mk3((c & 1) ? mxv.x : mn.x, (c & 2) ? mxv.y : mn.y, (c & 4) ? mxv.z : mn.z)));
        lo.x = fminf(lo.x, p.x); lo.y = fminf(lo.y, p.y); lo.z = fminf(lo.z, p.z);
        hi.x = fmaxf(hi.x, p.x); hi.y = fmaxf(hi.y, p.y); hi.z = fmaxf(hi.z, p.z);
    }
    // depthProjToCameraZ (RayCastSDFUtil.h:196-198) of the nearest / farthest corner
    const float dMin = lo.z * (a.maxDepth - a.minDepth) + a.minDepth, dMax = hi.z * (a.maxDepth - a.minDepth) + a.minDepth;
    // viewport transform: NDC x in [-1,1] -> [0,W], NDC y in [1,-1] -> [0,H]
    const float X0 = (lo.x * 0.5f + 0.5f) * (float)a.W, X1 = (hi.x * 0.5f + 0.5f) * (float)a.W;
    const float Y0 = (1.0f - (hi.y * 0.5f + 0.5f)) * (float)a.H, Y1 = (1.0f - (lo.y * 0.5f + 0.5f)) * (float)a.H;
    if (!(X0 < X1) || !(Y0 < Y1)) return;                     // also rejects NaN
    const int x0 = max(0, f2i(ceilf(X0 - 0.5f))), x1 = min((int)a.W, f2i(ceilf(X1 - 0.5f)));       // centres i + 0.5 in [X0, X1)
    const int y0 = max(0, f2i(ceilf(Y0 - 0.5f))), y1 = min((int)a.H, f2i(ceilf(Y1 - 0.5f)));
    if (x0 >= x1 || y0 >= y1) return;
    const uint32_t kMin = orderKey(dMin), kMax = orderKey(dMax);
    const int w = x1 - x0, n = w * (y1 - y0);
    for (int i = (int)lane; i < n; i += 64) {
        const uint32_t p = (uint32_t)(y0 + i / w) * a.W + (uint32_t)(x0 + i % w);
        atomicMin(&a.minKey[p], kMin);
        atomicMax(&a.maxKey[p], kMax);
    }
}

struct Sample { float sdf; uint32_t color; };
// RayCastData::trilinearInterpolationSimpleFastFast, RayCastSDFUtil.h:97-116 (distance and colour)
BF_DEV bool trilinearRc(const Vol& v, f3 pos, float& dist, uint32_t& color) {
    const float oSet = v.voxelSize;
    const f3 posDual = pos - mk3(oSet / 2.0f, oSet / 2.0f, oSet / 2.0f);
    const f3 pv = pos / v.voxelSize;
    const float wx = fracf_(pv.x), wy = fracf_(pv.y), wz = fracf_(pv.z);
    dist = 0.0f;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    Vx s;
#define BF_RC_TAP(ox, oy, oz, wgt)                                                                         \
    s = getVoxel(v, posDual + mk3(ox, oy, oz));                                                            \
    if (s.weight == 0) return false;                                                                       \
    {                                                                                                      \
        const float ww = (wgt);                                                                            \
        dist += ww * s.sdf;                                                                                \
        cr += ww * (float)(s.color & 0xFF); cg += ww * (float)((s.color >> 8) & 0xFF); cb += ww * (float)((s.color >> 16) & 0xFF); \
    }
    BF_RC_TAP(0.0f, 0.0f, 0.0f, (1.0f - wx) * (1.0f - wy) * (1.0f - wz))
    BF_RC_TAP(oSet, 0.0f, 0.0f, wx * (1.0f - wy) * (1.0f - wz))
    BF_RC_TAP(0.0f, oSet, 0.0f, (1.0f - wx) * wy * (1.0f - wz))
    BF_RC_TAP(0.0f, 0.0f, oSet, (1.0f - wx) * (1.0f - wy) * wz)
    BF_RC_TAP(oSet, oSet, 0.0f, wx * wy * (1.0f - wz))
    BF_RC_TAP(0.0f, oSet, oSet, (1.0f - wx) * wy * wz)
    BF_RC_TAP(oSet, 0.0f, oSet, wx * (1.0f - wy) * wz)
    BF_RC_TAP(oSet, oSet, oSet, wx * wy * wz)
#undef BF_RC_TAP
    // make_uchar3(colorFloat.x, ...): float -> unsigned char conversion (values are convex combinations of bytes)
    color = (f2u(cr) & 0xFF) | ((f2u(cg) & 0xFF) << 8) | ((f2u(cb) & 0xFF) << 16);
    return true;
}

// findIntersectionBisection, RayCastSDFUtil.h:127-150
BF_DEV bool bisection(const Vol& v, f3 camPos, f3 dir, float d0, float r0, float d1, float r1, float& alpha, uint32_t& color) {
    float a = r0, aDist = d0, b = r1, bDist = d1, c = 0.0f;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        c = a + (aDist / (aDist - bDist)) * (b - a);                       // findIntersectionLinear :120-123
        float cDist;
        if (!trilinearRc(v, camPos + dir * c, cDist, color)) return false;
        if (aDist * cDist > 0.0f) { a = c; aDist = cDist; }
        else { b = c; bDist = cDist; }
    }
    alpha = c;
    return true;
}

// gradientForPoint, RayCastSDFUtil.h:153-177
BF_DEV f3 gradientForPoint(const Vol& v, f3 pos) {
    const float vs = v.voxelSize;
    float dp00 = 0, d0p0 = 0, d00p = 0, d100 = 0, d010 = 0, d001 = 0; uint32_t col;
    trilinearRc(v, pos - mk3(0.5f * vs, 0.0f, 0.0f), dp00, col);
    trilinearRc(v, pos - mk3(0.0f, 0.5f * vs, 0.0f), d0p0, col);
    trilinearRc(v, pos - mk3(0.0f, 0.0f, 0.5f * vs), d00p, col);
    trilinearRc(v, pos + mk3(0.5f * vs, 0.0f, 0.0f), d100, col);
    trilinearRc(v, pos + mk3(0.0f, 0.5f * vs, 0.0f), d010, col);
    trilinearRc(v, pos + mk3(0.0f, 0.0f, 0.5f * vs), d001, col);
    const f3 g = mk3((dp00 - d100) / vs, (d0p0 - d010) / vs, (d00p - d001) / vs);
    const float l = sqrtf(dot3(g, g));
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(-g.x / l, -g.y / l, -g.z / l);
}

BF_DEV f3 normalize3(f3 v) { const float inv = 1.0f / sqrtf(dot3(v, v)); return v * inv; }

// renderKernel (CUDARayCastSDF.cu:17-48) + traverseCoarseGridSimpleSampleAll (RayCastSDFUtil.h:222-283)
__global__ __launch_bounds__(64) void k_rc_render(RcArgs a) {
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    if (x >= a.W || y >= a.H) return;
    const uint32_t p = y * a.W + x;
    const float4 minf4 = make_float4(BF_MINF, BF_MINF, BF_MINF, BF_MINF);
    a.depth[p] = BF_MINF; a.depth4[p] = minf4; a.normals[p] = minf4; a.colors[p] = minf4;
    const uint32_t kMin = a.minKey[p], kMax = a.maxKey[p];
    float minInterval = kMin == KEY_NONE_MIN ? BF_MINF : orderVal(kMin);
    float maxInterval = kMax == KEY_NONE_MAX ? BF_MINF : orderVal(kMax);
    a.rayMin[p] = minInterval; a.rayMax[p] = maxInterval;
    const float cx = ((float)x - a.mx) / a.fx, cy = ((float)y - a.my) / a.fy;
    const f3 camDir = normalize3(mk3(1.0f * cx, 1.0f * cy, 1.0f));                         // depthToCamera(x, y, 1.0f)
    const f3 worldCamPos = xform(a.viewInv, mk3(0.0f, 0.0f, 0.0f));
    const f3 wdir = mk3(a.viewInv.e[0] * camDir.x + a.viewInv.e[1] * camDir.y + a.viewInv.e[2] * camDir.z + a.viewInv.e[3] * 0.0f,
                        a.viewInv.e[4] * camDir.x + a.viewInv.e[5] * camDir.y + a.viewInv.e[6] * camDir.z + a.viewInv.e[7] * 0.0f,
                        a.viewInv.e[8] * camDir.x + a.viewInv.e[9] * camDir.y + a.viewInv.e[10] * camDir.z + a.viewInv.e[11] * 0.0f);
    const f3 worldDir = normalize3(wdir);
    if (minInterval == 0.0f || minInterval == BF_MINF) return;
    if (maxInterval == 0.0f || maxInterval == BF_MINF) return;
    minInterval = fmaxf(minInterval, a.minDepth);
    maxInterval = fminf(maxInterval, a.maxDepth);

    float lastSdf = 0.0f, lastAlpha = 0.0f; uint32_t lastWeight = 0;
    const float depthToRayLength = 1.0f / camDir.z;
    float rayCurrent = depthToRayLength * fmaxf(a.minDepth, minInterval);
    const float rayEnd = depthToRayLength * fminf(a.maxDepth, maxInterval);
#pragma unroll 1
    while (rayCurrent < rayEnd) {
        const f3 cur = worldCamPos + worldDir * rayCurrent;
        float dist; uint32_t color;
        if (trilinearRc(a.v, cur, dist, color)) {
            if (lastWeight > 0 && lastSdf > 0.0f && dist < 0.0f) {
                float alpha = 0.0f; uint32_t color2 = 0;
                const bool ok = bisection(a.v, worldCamPos, worldDir, lastSdf, lastAlpha, dist, rayCurrent, alpha, color2);
                const f3 iso = worldCamPos + worldDir * alpha;
                if (ok && fabsf(lastSdf - dist) < a.thresSampleDist) {
                    if (fabsf(dist) < a.thresDist) {
                        const float depth = alpha / depthToRayLength;
                        a.depth[p] = depth;
                        a.depth4[p] = make_float4(depth * cx, depth * cy, depth, 1.0f);
                        a.colors[p] = make_float4((float)(color2 & 0xFF) / 255.f, (float)((color2 >> 8) & 0xFF) / 255.f, (float)((color2 >> 16) & 0xFF) / 255.f, 1.0f);
                        if (a.useGradients) {
                            const f3 g = gradientForPoint(a.v, iso);
                            const f3 nrm = mk3(-g.x, -g.y, -g.z);
                            const m44& V = a.view;
                            a.normals[p] = make_float4(V.e[0] * nrm.x + V.e[1] * nrm.y + V.e[2] * nrm.z + V.e[3] * 0.0f,
                                                       V.e[4] * nrm.x + V.e[5] * nrm.y + V.e[6] * nrm.z + V.e[7] * 0.0f,
                                                       V.e[8] * nrm.x + V.e[9] * nrm.y + V.e[10] * nrm.z + V.e[11] * 0.0f, 1.0f);
                        }
                        return;
                    }
                }
            }
            lastSdf = dist; lastAlpha = rayCurrent; lastWeight = 1;
            rayCurrent += a.rayIncrement;
        } else {
            lastWeight = 0;
            rayCurrent += a.rayIncrement;
        }
    }
}

// computeNormalsDevice, CameraUtil.cu:665-693
__global__ void k_rc_normals(float4* out, const float4* in, uint32_t W, uint32_t H) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    out[y * W + x] = make_float4(BF_MINF, BF_MINF, BF_MINF, BF_MINF);
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        const float4 CC = in[y * W + x], PC = in[(y + 1) * W + x], CP = in[y * W + x + 1], MC = in[(y - 1) * W + x], CM = in[y * W + x - 1];
        if (CC.x != BF_MINF && PC.x != BF_MINF && CP.x != BF_MINF && MC.x != BF_MINF && CM.x != BF_MINF) {
            const f3 u = mk3(PC.x - MC.x, PC.y - MC.y, PC.z - MC.z), w = mk3(CP.x - CM.x, CP.y - CM.y, CP.z - CM.z);
            const f3 n = mk3(u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x);
            const float l = sqrtf(dot3(n, n));
            if (l > 0.0f) { const float nl = -l; out[y * W + x] = make_float4(n.x / nl, n.y / nl, n.z / nl, 1.0f); }
        }
    }
}

// convertDepthFloatToCameraSpaceFloat4Device, CameraUtil.cu:386-403 (kinectDepthToSkeleton with the DEPTH camera's intrinsics)
__global__ void k_rc_to_camera(float4* out, const float* in, bf_depth_camera_params cam, uint32_t W, uint32_t H) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    out[y * W + x] = make_float4(BF_MINF, BF_MINF, BF_MINF, BF_MINF);
    const float depth = in[y * W + x];
    if (depth != BF_MINF) {
        const float fx = ((float)x - cam.mx) / cam.fx, fy = ((float)y - cam.my) / cam.fy;
        out[y * W + x] = make_float4(depth * fx, depth * fy, depth, 1.0f);
    }
}

}  // namespace

struct bf_ray_cast {
    bf_ray_cast_params params;
    hipStream_t stream = nullptr;
    uint32_t capacity = 0;          // pixels allocated
    float *d_depth = nullptr, *d_depth4 = nullptr, *d_normals = nullptr, *d_colors = nullptr, *d_rayMin = nullptr, *d_rayMax = nullptr;
    uint32_t *d_minKey = nullptr, *d_maxKey = nullptr;
};

extern "C" {

int bf_ray_cast_create(const bf_ray_cast_params* p, bf_ray_cast** out) {                     // create(), CUDARayCastSDF.cpp:21-34
    BF_REQUIRE(p && out, "null argument");
    BF_REQUIRE(p->m_width > 0 && p->m_height > 0, "empty ray cast image");
    bf_ray_cast* rc = new bf_ray_cast;
    rc->params = *p;
    const size_t n = (size_t)p->m_width * p->m_height;
    rc->capacity = (uint32_t)n;
    hipError_t e = hipSuccess;
    auto A = [&](void** ptr, size_t bytes) { if (e == hipSuccess) e = BF_MALLOC(ptr, bytes); };
    A((void**)&rc->d_depth, n * 4); A((void**)&rc->d_depth4, n * 16); A((void**)&rc->d_normals, n * 16); A((void**)&rc->d_colors, n * 16);
    A((void**)&rc->d_rayMin, n * 4); A((void**)&rc->d_rayMax, n * 4); A((void**)&rc->d_minKey, n * 4); A((void**)&rc->d_maxKey, n * 4);
    if (e != hipSuccess) { bf_ray_cast_destroy(rc); set_error("hipMalloc failed: %s", hipGetErrorString(e)); return BF_ERR_HIP; }
    *out = rc;
    return BF_OK;
}

int bf_ray_cast_destroy(bf_ray_cast* rc) {
    if (!rc) return BF_OK;
    (void)hipFree(rc->d_depth); (void)hipFree(rc->d_depth4); (void)hipFree(rc->d_normals); (void)hipFree(rc->d_colors);
    (void)hipFree(rc->d_rayMin); (void)hipFree(rc->d_rayMax); (void)hipFree(rc->d_minKey); (void)hipFree(rc->d_maxKey);
    delete rc;
    return BF_OK;
}

int bf_ray_cast_set_stream(bf_ray_cast* rc, void* s) { BF_REQUIRE(rc, "null ray cast"); rc->stream = (hipStream_t)s; return BF_OK; }

// render(hashData, hashParams, lastRigidTransform), CUDARayCastSDF.cpp:42-72 + rayIntervalSplatting :83-98
int bf_ray_cast_render(bf_ray_cast* rc, const bf_hash_data* hd, const bf_hash_params* hp, const bf_depth_camera_params* cam, const float lastRigidTransform[16]) {
    BF_REQUIRE(rc && hd && hp && cam && lastRigidTransform, "null argument");
    BF_REQUIRE(hd->d_hash && hd->d_SDFBlocks && hd->d_hashCompactified, "hash data without arrays");
    bf_ray_cast_params& P = rc->params;
    BF_REQUIRE((size_t)P.m_width * P.m_height <= rc->capacity, "ray cast image larger than the buffers created");
    RcArgs a = {};
    a.v.hash = hd->d_hash; a.v.vox = hd->d_SDFBlocks; a.v.numBuckets = hp->m_hashNumBuckets; a.v.maxChain = hp->m_hashMaxCollisionLinkedListSize;
    a.v.voxelSize = hp->m_virtualVoxelSize;
    a.compact = hd->d_hashCompactified; a.numOccupied = hp->m_numOccupiedBlocks;
    if (hp->m_numOccupiedBlocks != 0) {                                      // rayIntervalSplatting returns early for an empty list: the view stays
        if (P.m_maxNumVertices <= 6 * hp->m_numOccupiedBlocks) { set_error("not enough space for vertex buffer for ray interval splatting"); return BF_ERR_CAPACITY; }
        P.m_numOccupiedSDFBlocks = hp->m_numOccupiedBlocks;
        m44 T; memcpy(T.e, lastRigidTransform, 64);
        const m44 Tinv = inverse44(T);
        memcpy(P.m_viewMatrix, Tinv.e, 64); memcpy(P.m_viewMatrixInverse, T.e, 64);
    }
    memcpy(a.view.e, P.m_viewMatrix, 64); memcpy(a.viewInv.e, P.m_viewMatrixInverse, 64);
    memcpy(a.rigidInv.e, hp->m_rigidTransformInverse, 64);
    a.cam = *cam;
    a.mx = P.mx; a.my = P.my; a.fx = P.fx; a.fy = P.fy; a.W = P.m_width; a.H = P.m_height;
    a.minDepth = P.m_minDepth; a.maxDepth = P.m_maxDepth; a.rayIncrement = P.m_rayIncrement; a.thresSampleDist = P.m_thresSampleDist; a.thresDist = P.m_thresDist;
    a.useGradients = P.m_useGradients;
    a.minKey = rc->d_minKey; a.maxKey = rc->d_maxKey; a.rayMin = rc->d_rayMin; a.rayMax = rc->d_rayMax;
    a.depth = rc->d_depth; a.depth4 = (float4*)rc->d_depth4; a.normals = (float4*)rc->d_normals; a.colors = (float4*)rc->d_colors;
    BF_REQUIRE(a.rayIncrement > 0.0f, "m_rayIncrement must be positive");
    const uint32_t n = a.W * a.H;
    k_rc_clear<<<div_up(n, 256), 256, 0, rc->stream>>>(a);
    if (a.numOccupied) k_rc_splat<<<div_up(a.numOccupied, 4), 256, 0, rc->stream>>>(a);
    k_rc_render<<<dim3(div_up(a.W, 8), div_up(a.H, 8)), 64, 0, rc->stream>>>(a);
    if (!P.m_useGradients) k_rc_normals<<<dim3(div_up(a.W, 16), div_up(a.H, 16)), dim3(16, 16), 0, rc->stream>>>((float4*)rc->d_normals, (const float4*)rc->d_depth4, a.W, a.H);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_ray_cast_get_data(bf_ray_cast* rc, bf_ray_cast_data* out) {                           // getRayCastData
    BF_REQUIRE(rc && out, "null argument");
    out->d_depth = rc->d_depth; out->d_depth4 = rc->d_depth4; out->d_normals = rc->d_normals; out->d_colors = rc->d_colors;
    out->d_rayIntervalSplatMin = rc->d_rayMin; out->d_rayIntervalSplatMax = rc->d_rayMax;
    return BF_OK;
}

int bf_ray_cast_get_params(bf_ray_cast* rc, bf_ray_cast_params* out) { BF_REQUIRE(rc && out, "null argument"); *out = rc->params; return BF_OK; }

int bf_ray_cast_update_min_max(bf_ray_cast* rc, float depthMin, float depthMax) {            // updateRayCastMinMax
    BF_REQUIRE(rc, "null ray cast");
    rc->params.m_minDepth = depthMin; rc->params.m_maxDepth = depthMax;
    return BF_OK;
}

int bf_ray_cast_set_intrinsics(bf_ray_cast* rc, uint32_t width, uint32_t height, const float intrinsics[16]) {      // setRayCastIntrinsics
    BF_REQUIRE(rc && intrinsics, "null argument");
    BF_REQUIRE((size_t)width * height <= rc->capacity && width > 0 && height > 0, "ray cast image larger than the buffers created");
    rc->params.m_width = width; rc->params.m_height = height;
    rc->params.fx = intrinsics[0]; rc->params.fy = intrinsics[5]; rc->params.mx = intrinsics[2]; rc->params.my = intrinsics[6];
    return BF_OK;
}

int bf_ray_cast_convert_to_camera_space(bf_ray_cast* rc, const bf_depth_camera_params* cam) {                     // convertToCameraSpace, .cpp:75-81
    BF_REQUIRE(rc && cam, "null argument");
    const uint32_t W = rc->params.m_width, H = rc->params.m_height;
    k_rc_to_camera<<<dim3(div_up(W, 16), div_up(H, 16)), dim3(16, 16), 0, rc->stream>>>((float4*)rc->d_depth4, rc->d_depth, *cam, W, H);
    if (!rc->params.m_useGradients) k_rc_normals<<<dim3(div_up(W, 16), div_up(H, 16)), dim3(16, 16), 0, rc->stream>>>((float4*)rc->d_normals, (const float4*)rc->d_depth4, W, H);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

}  // extern "C"
