// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of the reference's
// voxel-hashed TSDF: reset / alloc / compactify / integrate / de-integrate / GC.
//
// Follows  DepthSensing/VoxelUtilHashSDF.h:226-826,
//          DepthSensing/CUDASceneRepHashSDF.cu:27-684,
//          DepthSensing/CUDASceneRepHashSDF.h:65-155,328-391,
//          DepthSensing/DepthCameraUtil.h:70-144.
// PINNED to the reference's CUDASceneRepHashSDF.cu / VoxelUtilHashSDF.h through oracle/_ref: voxel bytes, key sets, per-bucket occupancy,
// free-block count and frustum-list sets on three configurations (tests/test_ref_pin_cpu.py::test_tsdf_operators_vs_reference_kernels).
//
// Canonical order (the reference is racy here, VoxelUtilHashSDF.h:604 try-lock,
// CUDASceneRepHashSDF.cu:348 atomic append):
//   * new block keys of one alloc are inserted in ascending (home bucket, packed key);
//     keys that find a free slot in their home bucket are placed first (pass 1), the
//     bucket-full ones walk the collision window afterwards in the same order (pass 2);
//   * the i-th new key consumes heap[heapCounter - i];
//   * GC deletes in ascending (home bucket, packed key) order;
//   * the allocated-block list and the compactified list are in insertion order.
// Any result of this order is one the reference's fixed-point loop can produce.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

#include "../include/bf_hip.h"
#include "or_common.h"

using namespace orc;

namespace {

const int BS = BF_SDF_BLOCK_SIZE;
const int VOX = BS * BS * BS;  // 512

struct AllocRec { uint64_t key; int32_t ptr; uint32_t pad; };

struct Scene {
    bf_hash_params p;
    bf_depth_camera_params cam;   // camera of the last (de)integrate / compactify
    std::vector<bf_hash_entry> hash;
    std::vector<uint32_t> heap;
    uint32_t heapCounter;
    std::vector<bf_voxel> vox;
    std::vector<bf_hash_entry> compact;
    std::vector<uint32_t> compactSrc;
    std::vector<AllocRec> allocList;   // ptr == FREE_ENTRY marks a hole
    uint32_t numIntegrated;
    uint32_t numDropped;
};

// VoxelUtilHashSDF.h:226-234.  NB: m_hashNumBuckets is `unsigned int`, so the `%`
// is evaluated in unsigned arithmetic (usual arithmetic conversions); the
// `if (res < 0)` fix-up in the reference is dead code.
inline uint32_t hashPos(const Scene& s, i3 v) {
    const uint32_t p0 = 73856093u, p1 = 19349669u, p2 = 83492791u;
    uint32_t h = ((uint32_t)v.x * p0) ^ ((uint32_t)v.y * p1) ^ ((uint32_t)v.z * p2);
    return h % s.p.m_hashNumBuckets;
}
inline bool keyable(i3 b) {
    const int L = 1 << 20;
    return b.x >= -L && b.x < L && b.y >= -L && b.y < L && b.z >= -L && b.z < L;
}
inline uint64_t packKey(i3 b) {
    const int L = 1 << 20;
    return ((uint64_t)(uint32_t)(b.z + L) << 42) | ((uint64_t)(uint32_t)(b.y + L) << 21) |
           (uint64_t)(uint32_t)(b.x + L);
}
inline i3 unpackKey(uint64_t k) {
    const int L = 1 << 20;
    return {(int)(k & 0x1FFFFF) - L, (int)((k >> 21) & 0x1FFFFF) - L, (int)((k >> 42) & 0x1FFFFF) - L};
}
// :283-287
inline i3 worldToVirtualVoxelPos(const Scene& s, f3 pos) {
    f3 p = pos / s.p.m_virtualVoxelSize;
    return {f2i(p.x + (float)sgn(p.x) * 0.5f), f2i(p.y + (float)sgn(p.y) * 0.5f),
            f2i(p.z + (float)sgn(p.z) * 0.5f)};
}
// :290-299
inline i3 virtualVoxelPosToSDFBlock(i3 v) {
    if (v.x < 0) v.x -= BS - 1;
    if (v.y < 0) v.y -= BS - 1;
    if (v.z < 0) v.z -= BS - 1;
    return {v.x / BS, v.y / BS, v.z / BS};
}
inline i3 worldToSDFBlock(const Scene& s, f3 w) { return virtualVoxelPosToSDFBlock(worldToVirtualVoxelPos(s, w)); }
// :303-315
inline f3 SDFBlockToWorld(const Scene& s, i3 b) {
    return mk3((float)(b.x * BS), (float)(b.y * BS), (float)(b.z * BS)) * s.p.m_virtualVoxelSize;
}
// DepthCameraUtil.h:70-76, 97-108
inline void cameraToKinectScreenFloat(const bf_depth_camera_params& c, f3 p, float& sx, float& sy) {
    sx = p.x * c.fx / p.z + c.mx;
    sy = p.y * c.fy / p.z + c.my;
}
inline float cameraToKinectProjZ(const bf_depth_camera_params& c, float z) {
    return (z - c.m_sensorDepthWorldMin) / (c.m_sensorDepthWorldMax - c.m_sensorDepthWorldMin);
}
// DepthCameraUtil.h:135-142
inline bool isInCameraFrustumApprox(const bf_depth_camera_params& c, const m44& viewInv, f3 pos) {
    f3 pc = xform(viewInv, pos);
    float sx, sy;
    cameraToKinectScreenFloat(c, pc, sx, sy);
    f3 pr;
    pr.x = (2.0f * sx - ((float)c.m_imageWidth - 1.0f)) / ((float)c.m_imageWidth - 1.0f);
    pr.y = (((float)c.m_imageHeight - 1.0f) - 2.0f * sy) / ((float)c.m_imageHeight - 1.0f);
    pr.z = cameraToKinectProjZ(c, pc.z);
    pr = pr * 0.95f;
    return !(pr.x < -1.0f || pr.x > 1.0f || pr.y < -1.0f || pr.y > 1.0f || pr.z < 0.0f || pr.z > 1.0f);
}
// VoxelUtilHashSDF.h:322-326
inline bool isSDFBlockInCameraFrustumApprox(const Scene& s, const m44& viewInv, i3 b) {
    f3 w = SDFBlockToWorld(s, b);
    const float off = s.p.m_virtualVoxelSize * 0.5f * ((float)BS - 1.0f);
    w = w + mk3(off, off, off);
    return isInCameraFrustumApprox(s.cam, viewInv, w);
}
// DepthCameraUtil.h:113-118
inline f3 kinectDepthToSkeleton(const bf_depth_camera_params& c, uint32_t ux, uint32_t uy, float d) {
    const float x = ((float)ux - c.mx) / c.fx;
    const float y = ((float)uy - c.my) / c.fy;
    return mk3(d * x, d * y, d);
}

// VoxelUtilHashSDF.h:441-485  (returns slot index or -1)
int findEntry(const Scene& s, i3 b) {
    const uint32_t h = hashPos(s, b);
    const uint32_t hp = h * BF_HASH_BUCKET_SIZE;
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        const bf_hash_entry& c = s.hash[hp + j];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) return (int)(hp + j);
    }
    const uint32_t last = (h + 1) * BF_HASH_BUCKET_SIZE - 1;
    const uint32_t total = BF_HASH_BUCKET_SIZE * s.p.m_hashNumBuckets;
    uint32_t i = last;
    for (uint32_t it = 0; it < s.p.m_hashMaxCollisionLinkedListSize; ++it) {
        const bf_hash_entry& c = s.hash[i];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) return (int)i;
        if (c.offset == 0) break;
        i = (last + c.offset) % total;
    }
    return -1;
}

void setRigid(Scene& s, const float* T) {   // CUDASceneRepHashSDF.h:128-134
    m44 m;
    memcpy(m.e, T, 64);
    m44 inv = inverse(m);
    memcpy(s.p.m_rigidTransform, m.e, 64);
    memcpy(s.p.m_rigidTransformInverse, inv.e, 64);
}

// CUDASceneRepHashSDF.cu:165-251 — candidate blocks of one depth pixel.
template <class F>
void ddaPixel(const Scene& s, const m44& T, const m44& Tinv, const float* depth, uint32_t x, uint32_t y, F&& visit) {
    const bf_hash_params& hp = s.p;
    float d = depth[(size_t)y * s.cam.m_imageWidth + x];
    if (d == MINF || d == 0.0f) return;
    if (d >= hp.m_maxIntegrationDistance) return;
    float t = hp.m_truncation + hp.m_truncScale * d;
    float minDepth = std::min(hp.m_maxIntegrationDistance, d - t);
    float maxDepth = std::min(hp.m_maxIntegrationDistance, d + t);
    if (minDepth >= maxDepth) return;
    f3 rayMin = xform(T, kinectDepthToSkeleton(s.cam, x, y, minDepth));
    f3 rayMax = xform(T, kinectDepthToSkeleton(s.cam, x, y, maxDepth));
    f3 dv = rayMax - rayMin;
    const float invLen = 1.0f / sqrtf(dot(dv, dv));   // cutil normalize(): v*rsqrtf(dot)
    f3 rayDir = dv * invLen;
    i3 cur = worldToSDFBlock(s, rayMin);
    i3 end = worldToSDFBlock(s, rayMax);
    f3 step = mk3((float)sgn(rayDir.x), (float)sgn(rayDir.y), (float)sgn(rayDir.z));
    auto c01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
    i3 nb = {cur.x + f2i(c01(step.x)), cur.y + f2i(c01(step.y)), cur.z + f2i(c01(step.z))};
    const float hv = 0.5f * hp.m_virtualVoxelSize;
    f3 boundary = SDFBlockToWorld(s, nb) - mk3(hv, hv, hv);
    f3 tMax = {(boundary.x - rayMin.x) / rayDir.x, (boundary.y - rayMin.y) / rayDir.y, (boundary.z - rayMin.z) / rayDir.z};
    const float bw = (float)BS * hp.m_virtualVoxelSize;   // (step*SDF_BLOCK_SIZE*voxelSize)/rayDir
    f3 tDelta = {(step.x * (float)BS * hp.m_virtualVoxelSize) / rayDir.x,
                 (step.y * (float)BS * hp.m_virtualVoxelSize) / rayDir.y,
                 (step.z * (float)BS * hp.m_virtualVoxelSize) / rayDir.z};
    (void)bw;
    i3 bound = {f2i((float)end.x + step.x), f2i((float)end.y + step.y), f2i((float)end.z + step.z)};
    if (rayDir.x == 0.0f) { tMax.x = PINF; tDelta.x = PINF; }
    if (boundary.x - rayMin.x == 0.0f) { tMax.x = PINF; tDelta.x = PINF; }
    if (rayDir.y == 0.0f) { tMax.y = PINF; tDelta.y = PINF; }
    if (boundary.y - rayMin.y == 0.0f) { tMax.y = PINF; tDelta.y = PINF; }
    if (rayDir.z == 0.0f) { tMax.z = PINF; tDelta.z = PINF; }
    if (boundary.z - rayMin.z == 0.0f) { tMax.z = PINF; tDelta.z = PINF; }
    for (unsigned iter = 0; iter < 1024; ++iter) {
        if (isSDFBlockInCameraFrustumApprox(s, Tinv, cur)) visit(cur);
        if (tMax.x < tMax.y && tMax.x < tMax.z) {
            cur.x = f2i((float)cur.x + step.x);
            if (cur.x == bound.x) return;
            tMax.x += tDelta.x;
        } else if (tMax.z < tMax.y) {
            cur.z = f2i((float)cur.z + step.z);
            if (cur.z == bound.z) return;
            tMax.z += tDelta.z;
        } else {
            cur.y = f2i((float)cur.y + step.y);
            if (cur.y == bound.y) return;
            tMax.y += tDelta.y;
        }
    }
}

struct SortRec { uint32_t bucket; uint64_t key; };
inline bool recLess(const SortRec& a, const SortRec& b) { return a.bucket != b.bucket ? a.bucket < b.bucket : a.key < b.key; }

// CUDASceneRepHashSDF.h:328-352 (alloc fixed point) + VoxelUtilHashSDF.h:549-655 (allocBlock)
void allocBlocks(Scene& s, const float* depth) {
    m44 T, Tinv;
    memcpy(T.e, s.p.m_rigidTransform, 64);
    memcpy(Tinv.e, s.p.m_rigidTransformInverse, 64);
    std::unordered_set<uint64_t> seen;
    std::vector<SortRec> cand;
    for (uint32_t y = 0; y < s.cam.m_imageHeight; ++y)
        for (uint32_t x = 0; x < s.cam.m_imageWidth; ++x)
            ddaPixel(s, T, Tinv, depth, x, y, [&](i3 b) {
                if (!keyable(b)) return;
                uint64_t k = packKey(b);
                if (seen.count(k)) return;
                seen.insert(k);
                if (findEntry(s, b) >= 0) return;
                cand.push_back({hashPos(s, b), k});
            });
    std::sort(cand.begin(), cand.end(), recLess);
    const uint32_t heapFree = s.heapCounter + 1;   // CUDASceneRepHashSDF.h:171
    size_t M = cand.size();
    if (M > heapFree) { s.numDropped += (uint32_t)(M - heapFree); M = heapFree; }
    const uint32_t total = BF_HASH_BUCKET_SIZE * s.p.m_hashNumBuckets;
    const size_t base = s.allocList.size();
    s.allocList.resize(base + M);
    std::vector<size_t> overflow;
    // pass 1: home-bucket slots
    for (size_t i = 0; i < M;) {
        size_t j = i;
        while (j < M && cand[j].bucket == cand[i].bucket) ++j;
        const uint32_t hp = cand[i].bucket * BF_HASH_BUCKET_SIZE;
        uint32_t slot = 0;
        for (size_t k = i; k < j; ++k) {
            while (slot < BF_HASH_BUCKET_SIZE && s.hash[hp + slot].ptr != BF_FREE_ENTRY) ++slot;
            const int32_t ptr = (int32_t)(s.heap[s.heapCounter - k] * VOX);
            s.allocList[base + k] = {cand[k].key, ptr, 0};
            if (slot < BF_HASH_BUCKET_SIZE) {
                i3 b = unpackKey(cand[k].key);
                bf_hash_entry& e = s.hash[hp + slot];
                e.pos[0] = b.x; e.pos[1] = b.y; e.pos[2] = b.z;
                e.offset = 0;            // NO_OFFSET, :608
                e.ptr = ptr;
                ++slot;
            } else {
                overflow.push_back(k);
            }
        }
        i = j;
    }
    // pass 2: collision window, VoxelUtilHashSDF.h:614-654
    std::vector<uint32_t> returned;
    for (size_t k : overflow) {
        const uint32_t h = cand[k].bucket;
        const uint32_t last = (h + 1) * BF_HASH_BUCKET_SIZE - 1;
        bool done = false;
        uint32_t maxIter = 0;
        int offset = 0;
        while (maxIter < s.p.m_hashMaxCollisionLinkedListSize) {
            offset++;
            uint32_t i = (last + (uint32_t)offset) % total;
            if ((offset % BF_HASH_BUCKET_SIZE) == 0) continue;
            if (s.hash[i].ptr == BF_FREE_ENTRY) {
                i3 b = unpackKey(cand[k].key);
                bf_hash_entry& e = s.hash[i];
                e.pos[0] = b.x; e.pos[1] = b.y; e.pos[2] = b.z;
                e.offset = s.hash[last].offset;
                e.ptr = s.allocList[base + k].ptr;
                s.hash[last].offset = (uint32_t)offset;
                done = true;
                break;
            }
            maxIter++;
        }
        if (!done) {
            returned.push_back((uint32_t)s.allocList[base + k].ptr / VOX);
            s.allocList[base + k].ptr = BF_FREE_ENTRY;
            s.numDropped++;
        }
    }
    s.heapCounter -= (uint32_t)M;
    for (uint32_t r : returned) { s.heapCounter++; s.heap[s.heapCounter] = r; }   // appendHeap :542-546
}

// CUDASceneRepHashSDF.cu:324-366 — frustum list (here from the allocated-block list)
void compactify(Scene& s) {
    m44 Tinv;
    memcpy(Tinv.e, s.p.m_rigidTransformInverse, 64);
    s.compact.clear();
    s.compactSrc.clear();
    for (size_t i = 0; i < s.allocList.size(); ++i) {
        const AllocRec& r = s.allocList[i];
        if (r.ptr == BF_FREE_ENTRY) continue;
        i3 b = unpackKey(r.key);
        if (!isSDFBlockInCameraFrustumApprox(s, Tinv, b)) continue;
        bf_hash_entry e;
        memset(&e, 0, sizeof e);
        e.pos[0] = b.x; e.pos[1] = b.y; e.pos[2] = b.z;
        e.ptr = r.ptr;
        e.offset = 0;
        s.compact.push_back(e);
        s.compactSrc.push_back((uint32_t)i);
    }
    s.p.m_numOccupiedBlocks = (uint32_t)s.compact.size();
}

inline float roundHalfAway(float v) { return roundf(v); }   // CUDA round()

// CUDASceneRepHashSDF.cu:420-521, one voxel
template <bool DEINT>
inline void updateVoxel(const Scene& s, const m44& Tinv, const bf_hash_entry& entry, uint32_t i,
                        const float* depthImg, const uint8_t* colorImg, bf_voxel* vox) {
    const bf_hash_params& hp = s.p;
    const bf_depth_camera_params& cam = s.cam;
    const int bx = entry.pos[0] * BS + (int)(i % BS);
    const int by = entry.pos[1] * BS + (int)((i % (BS * BS)) / BS);
    const int bz = entry.pos[2] * BS + (int)(i / (BS * BS));
    f3 pf = mk3((float)bx, (float)by, (float)bz) * hp.m_virtualVoxelSize;
    pf = xform(Tinv, pf);
    float sx, sy;
    cameraToKinectScreenFloat(cam, pf, sx, sy);
    const uint32_t px = (uint32_t)f2i(sx + 0.5f), py = (uint32_t)f2i(sy + 0.5f);
    if (!(px < cam.m_imageWidth && py < cam.m_imageHeight)) return;
    const float depth = depthImg[(size_t)py * cam.m_imageWidth + px];
    float cr = MINF, cg = MINF, cb = MINF;
    if (colorImg) {
        const uint8_t* c = colorImg + 4 * ((size_t)py * cam.m_imageWidth + px);
        cr = (float)c[0]; cg = (float)c[1]; cb = (float)c[2];
    }
    if (!(cr != MINF && depth != MINF)) return;
    if (!(depth < hp.m_maxIntegrationDistance)) return;
    float sdf = depth - pf.z;
    const float trunc = hp.m_truncation + hp.m_truncScale * depth;
    if (!(fabsf(sdf) < trunc)) return;
    if (sdf >= 0.0f) sdf = fminf(trunc, sdf); else sdf = fmaxf(-trunc, sdf);
    const float wUpd = 1.0f;   // :466 "weightUpdate = 1.0f"
    uint8_t ccol[3];
    if (colorImg) { ccol[0] = (uint8_t)f2i(cr); ccol[1] = (uint8_t)f2i(cg); ccol[2] = (uint8_t)f2i(cb); }
    else { ccol[0] = 0; ccol[1] = 255; ccol[2] = 0; }
    bf_voxel& v = vox[(size_t)entry.ptr + i];
    const bf_voxel old = v;
    bf_voxel nv;
    float res[3];
    if (!DEINT) {
        for (int k = 0; k < 3; ++k) {
            float r;
            if (old.weight == 0) r = (float)ccol[k];
            else r = 0.2f * (float)ccol[k] + 0.8f * (float)old.color[k];
            r = roundHalfAway(r);
            res[k] = fmaxf(0.0f, fminf(r, 254.5f));
        }
        nv.color[0] = (uint8_t)f2i(res[0]); nv.color[1] = (uint8_t)f2i(res[1]); nv.color[2] = (uint8_t)f2i(res[2]); nv.color[3] = 255;
        nv.sdf = (sdf * wUpd + old.sdf * old.weight) / (wUpd + old.weight);
        nv.weight = fminf((float)hp.m_integrationWeightMax, wUpd + old.weight);
    } else {
        for (int k = 0; k < 3; ++k) {
            float r = ((float)old.color[k] * old.weight - (float)ccol[k] * wUpd) / (old.weight - wUpd);
            r = roundHalfAway(r);
            res[k] = fmaxf(0.0f, fminf(r, 254.5f));
        }
        nv.color[0] = (uint8_t)f2i(res[0]); nv.color[1] = (uint8_t)f2i(res[1]); nv.color[2] = (uint8_t)f2i(res[2]); nv.color[3] = 255;
        nv.sdf = (old.sdf * old.weight - sdf * wUpd) / (old.weight - wUpd);
        nv.weight = fmaxf(0.0f, old.weight - wUpd);
        if (nv.weight <= 0.001f) { nv.sdf = 0.0f; nv.color[0] = nv.color[1] = nv.color[2] = nv.color[3] = 0; nv.weight = 0.0f; }
    }
    v = nv;
}

template <bool DEINT>
void updateBlocks(Scene& s, const float* depth, const uint8_t* color, int threads) {
    m44 Tinv;
    memcpy(Tinv.e, s.p.m_rigidTransformInverse, 64);
    const long n = (long)s.compact.size();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long b = 0; b < n; ++b)
        for (uint32_t i = 0; i < (uint32_t)VOX; ++i)
            updateVoxel<DEINT>(s, Tinv, s.compact[b], i, depth, color, s.vox.data());
}

// VoxelUtilHashSDF.h:740-826 deleteHashEntryElement (serial: locks always succeed)
bool deleteEntry(Scene& s, i3 b) {
    const uint32_t h = hashPos(s, b);
    const uint32_t hp = h * BF_HASH_BUCKET_SIZE;
    const uint32_t total = BF_HASH_BUCKET_SIZE * s.p.m_hashNumBuckets;
    auto clear = [&](uint32_t i) { s.hash[i].pos[0] = s.hash[i].pos[1] = s.hash[i].pos[2] = 0; s.hash[i].offset = 0; s.hash[i].ptr = BF_FREE_ENTRY; };
    auto appendHeap = [&](int32_t ptr) { s.heapCounter++; s.heap[s.heapCounter] = (uint32_t)ptr / VOX; };
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        const uint32_t i = hp + j;
        bf_hash_entry c = s.hash[i];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) {
            appendHeap(c.ptr);
            if (c.offset != 0) {
                const uint32_t next = (i + c.offset) % total;
                s.hash[i] = s.hash[next];
                clear(next);
            } else {
                clear(i);
            }
            return true;
        }
    }
    const uint32_t last = (h + 1) * BF_HASH_BUCKET_SIZE - 1;
    bf_hash_entry c = s.hash[last];
    uint32_t prev = last;
    uint32_t i = (last + c.offset) % total;
    for (uint32_t it = 0; it < s.p.m_hashMaxCollisionLinkedListSize; ++it) {
        c = s.hash[i];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) {
            appendHeap(c.ptr);
            clear(i);
            s.hash[prev].offset = c.offset;
            return true;
        }
        if (c.offset == 0) return false;
        prev = i;
        i = (last + c.offset) % total;
    }
    return false;
}

// CUDASceneRepHashSDF.h:110-126, .cu:584-668
void garbageCollect(Scene& s) {
    if (s.compact.empty()) return;
    std::vector<SortRec> del;
    std::vector<uint32_t> delSrc;
    for (size_t b = 0; b < s.compact.size(); ++b) {
        const bf_hash_entry& e = s.compact[b];
        uint32_t maxW = 0;
        for (int i = 0; i < VOX; ++i) {
            const float w = s.vox[(size_t)e.ptr + i].weight;
            const uint32_t wi = (uint32_t)f2i(w);   // shared_MaxWeight is uint, .cu:581,606
            maxW = std::max(maxW, wi);
        }
        if (maxW == 0) {
            i3 p = {e.pos[0], e.pos[1], e.pos[2]};
            del.push_back({hashPos(s, p), packKey(p)});
            s.allocList[s.compactSrc[b]].ptr = BF_FREE_ENTRY;
            (void)delSrc;
        }
    }
    std::vector<std::pair<SortRec, int32_t>> order;
    for (auto& d : del) order.push_back({d, 0});
    std::sort(order.begin(), order.end(), [](auto& a, auto& b) { return recLess(a.first, b.first); });
    for (auto& d : order) {
        i3 b = unpackKey(d.first.key);
        int slot = findEntry(s, b);
        if (slot < 0) continue;
        const int32_t ptr = s.hash[slot].ptr;
        if (deleteEntry(s, b)) {
            for (int i = 0; i < VOX; ++i) { bf_voxel& v = s.vox[(size_t)ptr + i]; v.sdf = 0; v.weight = 0; v.color[0] = v.color[1] = v.color[2] = v.color[3] = 0; }
        }
    }
    // stable compaction of the allocated list, then refresh the frustum list
    size_t w = 0;
    for (size_t i = 0; i < s.allocList.size(); ++i)
        if (s.allocList[i].ptr != BF_FREE_ENTRY) s.allocList[w++] = s.allocList[i];
    s.allocList.resize(w);
    compactify(s);
}

void resetScene(Scene& s) {   // CUDASceneRepHashSDF.cu:27-65
    const uint32_t N = s.p.m_numSDFBlocks;
    s.heapCounter = N - 1;
    for (uint32_t i = 0; i < N; ++i) s.heap[i] = N - i - 1;
    memset(s.vox.data(), 0, s.vox.size() * sizeof(bf_voxel));
    for (auto& e : s.hash) { memset(&e, 0, sizeof e); e.ptr = BF_FREE_ENTRY; }
    s.compact.clear();
    s.compactSrc.clear();
    s.allocList.clear();
    s.numIntegrated = 0;
    s.numDropped = 0;
    s.p.m_numOccupiedBlocks = 0;
    m44 I = m44::identity();
    memcpy(s.p.m_rigidTransform, I.e, 64);
    memcpy(s.p.m_rigidTransformInverse, I.e, 64);
}

}  // namespace

extern "C" {

void* or_scene_create(const bf_hash_params* p) {
    Scene* s = new Scene();
    s->p = *p;
    memset(&s->cam, 0, sizeof s->cam);
    s->hash.resize((size_t)p->m_hashNumBuckets * BF_HASH_BUCKET_SIZE);
    s->heap.resize(p->m_numSDFBlocks);
    s->vox.resize((size_t)p->m_numSDFBlocks * VOX);
    resetScene(*s);
    return s;
}
void or_scene_destroy(void* h) { delete (Scene*)h; }
void or_scene_reset(void* h) { resetScene(*(Scene*)h); }

void or_scene_integrate(void* h, const float* T, const float* depth, const uint8_t* color,
                        const bf_depth_camera_params* cam, int threads) {
    Scene& s = *(Scene*)h;
    s.cam = *cam;
    setRigid(s, T);
    allocBlocks(s, depth);
    compactify(s);
    updateBlocks<false>(s, depth, color, threads < 1 ? 1 : threads);
    s.numIntegrated++;
}
void or_scene_deintegrate(void* h, const float* T, const float* depth, const uint8_t* color,
                          const bf_depth_camera_params* cam, int threads) {
    Scene& s = *(Scene*)h;
    s.cam = *cam;
    setRigid(s, T);
    compactify(s);
    updateBlocks<true>(s, depth, color, threads < 1 ? 1 : threads);
    s.numIntegrated--;
}
void or_scene_compactify(void* h, const float* T, const bf_depth_camera_params* cam) {
    Scene& s = *(Scene*)h;
    s.cam = *cam;
    setRigid(s, T);
    compactify(s);
}
void or_scene_garbage_collect(void* h) { garbageCollect(*(Scene*)h); }

// timed pieces for bench.py's cpu_baseline: seconds for the voxel update only
double or_scene_time_update(void* h, const float* depth, const uint8_t* color, int threads, int deint) {
    Scene& s = *(Scene*)h;
    auto t0 = std::chrono::steady_clock::now();
    if (deint) updateBlocks<true>(s, depth, color, threads); else updateBlocks<false>(s, depth, color, threads);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

const bf_hash_entry* or_scene_hash(void* h) { return ((Scene*)h)->hash.data(); }
const uint32_t* or_scene_heap(void* h) { return ((Scene*)h)->heap.data(); }
uint32_t or_scene_heap_counter(void* h) { return ((Scene*)h)->heapCounter; }
const bf_voxel* or_scene_voxels(void* h) { return ((Scene*)h)->vox.data(); }
const bf_hash_entry* or_scene_compactified(void* h) { return ((Scene*)h)->compact.data(); }
uint32_t or_scene_num_occupied(void* h) { return (uint32_t)((Scene*)h)->compact.size(); }
uint32_t or_scene_num_allocated(void* h) {
    uint32_t n = 0;
    for (auto& r : ((Scene*)h)->allocList) n += (r.ptr != BF_FREE_ENTRY);
    return n;
}
uint32_t or_scene_num_dropped(void* h) { return ((Scene*)h)->numDropped; }
void or_scene_params(void* h, bf_hash_params* out) { *out = ((Scene*)h)->p; }

uint32_t or_hash_pos(uint32_t numBuckets, int x, int y, int z) {
    Scene s; s.p.m_hashNumBuckets = numBuckets;
    return hashPos(s, {x, y, z});
}
void or_world_to_block(float voxelSize, const float* w, int* out) {
    Scene s; s.p.m_virtualVoxelSize = voxelSize;
    i3 v = worldToVirtualVoxelPos(s, {w[0], w[1], w[2]});
    i3 b = virtualVoxelPosToSDFBlock(v);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = b.x; out[4] = b.y; out[5] = b.z;
}
void or_mat4_inverse(const float* m, float* out) {
    m44 a; memcpy(a.e, m, 64);
    m44 r = inverse(a);
    memcpy(out, r.e, 64);
}

}  // extern "C"
