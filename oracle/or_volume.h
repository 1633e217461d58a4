// TEST INFRASTRUCTURE ONLY (see or_common.h) - read-only view of the voxel hash shared by the consumers' restatements (marching cubes,
// ray cast): DepthSensing/VoxelUtilHashSDF.h:226-234 (hash), :278-299 (voxel / block maps), :347-358 (local index), :407-417 (getVoxel),
// :441-485 (getHashEntryForSDFBlockPos).  Paths relative to /root/reference/FriedLiver/Source.
#pragma once
#include <cstdint>
#include <cstring>

#include "../include/bf_hip.h"
#include "or_common.h"

namespace orc {

const int VBS = BF_SDF_BLOCK_SIZE;

struct Vol { const bf_hash_entry* hash; const bf_voxel* vox; uint32_t numBuckets, maxChain; float voxelSize; };

inline uint32_t hashPos(const Vol& v, i3 b) {
    const uint32_t h = ((uint32_t)b.x * 73856093u) ^ ((uint32_t)b.y * 19349669u) ^ ((uint32_t)b.z * 83492791u);
    return h % v.numBuckets;
}
inline i3 worldToVirtualVoxelPos(const Vol& v, f3 pos) {
    const f3 p = pos / v.voxelSize;
    return {f2i(p.x + (float)sgn(p.x) * 0.5f), f2i(p.y + (float)sgn(p.y) * 0.5f), f2i(p.z + (float)sgn(p.z) * 0.5f)};
}
inline i3 voxelToBlock(i3 p) {
    if (p.x < 0) p.x -= VBS - 1;
    if (p.y < 0) p.y -= VBS - 1;
    if (p.z < 0) p.z -= VBS - 1;
    return {p.x / VBS, p.y / VBS, p.z / VBS};
}
inline int localIndex(i3 p) {
    int x = p.x % VBS, y = p.y % VBS, z = p.z % VBS;
    if (x < 0) x += VBS;
    if (y < 0) y += VBS;
    if (z < 0) z += VBS;
    return z * VBS * VBS + y * VBS + x;
}
inline int32_t findBlock(const Vol& v, i3 b) {                    // getHashEntryForSDFBlockPos :441-485
    const uint32_t hp = hashPos(v, b) * BF_HASH_BUCKET_SIZE;
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        const bf_hash_entry& e = v.hash[hp + j];
        if (e.pos[0] == b.x && e.pos[1] == b.y && e.pos[2] == b.z && e.ptr != BF_FREE_ENTRY) return e.ptr;
    }
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1, total = BF_HASH_BUCKET_SIZE * v.numBuckets;
    uint32_t i = last;
    for (uint32_t it = 0; it < v.maxChain; ++it) {
        const bf_hash_entry& e = v.hash[i];
        if (e.pos[0] == b.x && e.pos[1] == b.y && e.pos[2] == b.z && e.ptr != BF_FREE_ENTRY) return e.ptr;
        if (e.offset == 0) break;
        i = (last + e.offset) % total;
    }
    return BF_FREE_ENTRY;
}
struct Vx { float sdf, weight; uint8_t c[4]; };
inline Vx getVoxel(const Vol& v, f3 w) {
    const i3 vp = worldToVirtualVoxelPos(v, w);
    const int32_t ptr = findBlock(v, voxelToBlock(vp));
    Vx r; memset(&r, 0, sizeof r);
    if (ptr != BF_FREE_ENTRY) { const bf_voxel& s = v.vox[(size_t)ptr + localIndex(vp)]; r.sdf = s.sdf; r.weight = s.weight; memcpy(r.c, s.color, 4); }
    return r;
}

}  // namespace orc
