// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of the reference's marching-cubes extraction over the voxel hash.
//
// Follows  DepthSensing/MarchingCubesSDFUtil.h:118-253 (extractIsoSurfaceAtPosition, vertexInterp),
//          DepthSensing/RayCastSDFUtil.h:86-116 (frac, trilinearInterpolationSimpleFastFast),
//          DepthSensing/VoxelUtilHashSDF.h:278-299,347-358,407-417,441-485 (voxel / block maps, getVoxel, getHashEntryForSDFBlockPos),
//          DepthSensing/CUDAMarchingCubesSDF.cu:15-28 (one thread per voxel of every occupied hash slot).
// The case tables are an ARGUMENT: tests pass the reference's own Tables.h (through oracle/_ref) to pin this restatement to the
// reference triangle for triangle, and the product's generated tables to check the HIP kernel bit for bit.
// Canonical order (the reference appends with one global atomic): hash slot ascending, voxel index z*64 + y*8 + x ascending, table order.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/bf_hip.h"
#include "or_common.h"
#include "or_volume.h"

using namespace orc;

namespace {

const int BS = BF_SDF_BLOCK_SIZE;
inline float frac1(float x) { return x - floorf(x); }
bool trilinear(const Vol& v, f3 pos, float& dist) {        // RayCastSDFUtil.h:97-116
    const float oSet = v.voxelSize;
    const f3 posDual = pos - mk3(oSet / 2.0f, oSet / 2.0f, oSet / 2.0f);
    const f3 pv = pos / v.voxelSize;
    const float wx = frac1(pv.x), wy = frac1(pv.y), wz = frac1(pv.z);
    dist = 0.0f;
    Vx s;
    s = getVoxel(v, posDual + mk3(0.0f, 0.0f, 0.0f)); if (s.weight == 0) return false; dist += (1.0f - wx) * (1.0f - wy) * (1.0f - wz) * s.sdf;
    s = getVoxel(v, posDual + mk3(oSet, 0.0f, 0.0f)); if (s.weight == 0) return false; dist += wx * (1.0f - wy) * (1.0f - wz) * s.sdf;
    s = getVoxel(v, posDual + mk3(0.0f, oSet, 0.0f)); if (s.weight == 0) return false; dist += (1.0f - wx) * wy * (1.0f - wz) * s.sdf;
    s = getVoxel(v, posDual + mk3(0.0f, 0.0f, oSet)); if (s.weight == 0) return false; dist += (1.0f - wx) * (1.0f - wy) * wz * s.sdf;
    s = getVoxel(v, posDual + mk3(oSet, oSet, 0.0f)); if (s.weight == 0) return false; dist += wx * wy * (1.0f - wz) * s.sdf;
    s = getVoxel(v, posDual + mk3(0.0f, oSet, oSet)); if (s.weight == 0) return false; dist += (1.0f - wx) * wy * wz * s.sdf;
    s = getVoxel(v, posDual + mk3(oSet, 0.0f, oSet)); if (s.weight == 0) return false; dist += wx * (1.0f - wy) * wz * s.sdf;
    s = getVoxel(v, posDual + mk3(oSet, oSet, oSet)); if (s.weight == 0) return false; dist += wx * wy * wz * s.sdf;
    return true;
}
struct Vert { f3 p, c; };
Vert vertexInterp(float iso, f3 p1, f3 p2, float d1, float d2, const uint8_t* c1, const uint8_t* c2) {      // MarchingCubesSDFUtil.h:231-253
    Vert r1; r1.p = p1; r1.c = mk3((float)c1[0], (float)c1[1], (float)c1[2]) / 255.f;
    Vert r2; r2.p = p2; r2.c = mk3((float)c2[0], (float)c2[1], (float)c2[2]) / 255.f;
    if (fabsf(iso - d1) < 0.00001f) return r1;
    if (fabsf(iso - d2) < 0.00001f) return r2;
    if (fabsf(d1 - d2) < 0.00001f) return r1;
    const float mu = (iso - d1) / (d2 - d1);
    Vert r;
    r.p.x = p1.x + mu * (p2.x - p1.x); r.p.y = p1.y + mu * (p2.y - p1.y); r.p.z = p1.z + mu * (p2.z - p1.z);
    r.c.x = (float)((float)c1[0] + mu * (float)((int)c2[0] - (int)c1[0])) / 255.f;
    r.c.y = (float)((float)c1[1] + mu * (float)((int)c2[1] - (int)c1[1])) / 255.f;
    r.c.z = (float)((float)c1[2] + mu * (float)((int)c2[2] - (int)c1[2])) / 255.f;
    return r;
}

}  // namespace

extern "C" {

// returns the number of triangles the volume holds; the first min(that, maxTriangles) are written (18 floats each: 3 x (position, colour))
uint32_t or_mc_extract(const bf_hash_entry* hash, const bf_voxel* vox, const bf_hash_params* hp, float thresh, float thresh2, int boxEnabled, const float* minCorner,
                       const float* maxCorner, const uint16_t* edgeTable, const int8_t* triTable, float* out, uint32_t maxTriangles) {
    Vol v = {hash, vox, hp->m_hashNumBuckets, hp->m_hashMaxCollisionLinkedListSize, hp->m_virtualVoxelSize};
    const uint32_t numSlots = hp->m_hashNumBuckets * BF_HASH_BUCKET_SIZE;
    std::vector<uint32_t> occ;
    for (uint32_t slot = 0; slot < numSlots; ++slot) if (hash[slot].ptr != BF_FREE_ENTRY) occ.push_back(slot);
    std::vector<std::vector<float>> per(occ.size());          // blocks are independent: evaluated in parallel, concatenated in slot order
#pragma omp parallel for schedule(dynamic, 8)
    for (long b = 0; b < (long)occ.size(); ++b) {
        const bf_hash_entry& e = hash[occ[b]];
        std::vector<float>& mine = per[b];
        for (int i = 0; i < BS * BS * BS; ++i) {
            const i3 pi = {e.pos[0] * BS + (i & 7), e.pos[1] * BS + ((i >> 3) & 7), e.pos[2] * BS + (i >> 6)};
            const f3 worldPos = mk3((float)pi.x, (float)pi.y, (float)pi.z) * v.voxelSize;
            if (boxEnabled == 1) {
                if (worldPos.x < minCorner[0] || worldPos.x > maxCorner[0]) continue;
                if (worldPos.y < minCorner[1] || worldPos.y > maxCorner[1]) continue;
                if (worldPos.z < minCorner[2] || worldPos.z > maxCorner[2]) continue;
            }
            const float iso = 0.0f, P = v.voxelSize / 2.0f, M = -P;
            const f3 p000 = worldPos + mk3(M, M, M), p100 = worldPos + mk3(P, M, M), p010 = worldPos + mk3(M, P, M), p001 = worldPos + mk3(M, M, P);
            const f3 p110 = worldPos + mk3(P, P, M), p011 = worldPos + mk3(M, P, P), p101 = worldPos + mk3(P, M, P), p111 = worldPos + mk3(P, P, P);
            float d000, d100, d010, d001, d110, d011, d101, d111;
            const bool v000 = trilinear(v, p000, d000), v100 = trilinear(v, p100, d100), v010 = trilinear(v, p010, d010), v001 = trilinear(v, p001, d001);
            const bool v110 = trilinear(v, p110, d110), v011 = trilinear(v, p011, d011), v101 = trilinear(v, p101, d101), v111 = trilinear(v, p111, d111);
            if (!v000 || !v100 || !v010 || !v001 || !v110 || !v011 || !v101 || !v111) continue;
            uint32_t ci = 0;
            if (d010 < iso) ci += 1;
            if (d110 < iso) ci += 2;
            if (d100 < iso) ci += 4;
            if (d000 < iso) ci += 8;
            if (d011 < iso) ci += 16;
            if (d111 < iso) ci += 32;
            if (d101 < iso) ci += 64;
            if (d001 < iso) ci += 128;
            const float da[8] = {d000, d100, d010, d001, d110, d011, d101, d111};
            bool skip = false;
            for (int k = 0; k < 8 && !skip; ++k)
                for (int l = 0; l < 8; ++l) {
                    if (da[k] * da[l] < 0.0f) { if (fabsf(da[k]) + fabsf(da[l]) > thresh) { skip = true; break; } }
                    else { if (fabsf(da[k] - da[l]) > thresh) { skip = true; break; } }
                }
            for (int k = 0; k < 8 && !skip; ++k) if (fabsf(da[k]) > thresh2) skip = true;
            if (skip) continue;
            const uint32_t em = edgeTable[ci];
            if (em == 0 || em == 255) continue;
            const Vx own = getVoxel(v, worldPos);
            Vert vl[12];
            if (em & 1) vl[0] = vertexInterp(iso, p010, p110, d010, d110, own.c, own.c);
            if (em & 2) vl[1] = vertexInterp(iso, p110, p100, d110, d100, own.c, own.c);
            if (em & 4) vl[2] = vertexInterp(iso, p100, p000, d100, d000, own.c, own.c);
            if (em & 8) vl[3] = vertexInterp(iso, p000, p010, d000, d010, own.c, own.c);
            if (em & 16) vl[4] = vertexInterp(iso, p011, p111, d011, d111, own.c, own.c);
            if (em & 32) vl[5] = vertexInterp(iso, p111, p101, d111, d101, own.c, own.c);
            if (em & 64) vl[6] = vertexInterp(iso, p101, p001, d101, d001, own.c, own.c);
            if (em & 128) vl[7] = vertexInterp(iso, p001, p011, d001, d011, own.c, own.c);
            if (em & 256) vl[8] = vertexInterp(iso, p010, p011, d010, d011, own.c, own.c);
            if (em & 512) vl[9] = vertexInterp(iso, p110, p111, d110, d111, own.c, own.c);
            if (em & 1024) vl[10] = vertexInterp(iso, p100, p101, d100, d101, own.c, own.c);
            if (em & 2048) vl[11] = vertexInterp(iso, p000, p001, d000, d001, own.c, own.c);
            for (int t = 0; triTable[ci * 16 + t] != -1; t += 3)
                for (int k = 0; k < 3; ++k) {
                    const Vert& s = vl[triTable[ci * 16 + t + k]];
                    const float o[6] = {s.p.x, s.p.y, s.p.z, s.c.x, s.c.y, s.c.z};
                    mine.insert(mine.end(), o, o + 6);
                }
        }
    }
    uint32_t n = 0;
    for (const auto& v1 : per)
        for (size_t i = 0; i + 18 <= v1.size(); i += 18) {
            if (n < maxTriangles) memcpy(out + (size_t)n * 18, v1.data() + i, 72);
            ++n;
        }
    return n;
}

}  // extern "C"
