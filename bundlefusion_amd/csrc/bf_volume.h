// Read-only device view of the voxel-hash volume in the reference layout (32-byte HashEntry, 12-byte Voxel, ptr = block * 512):
// what the consumers of the volume (marching cubes, ray cast) share.  Integer maps and look-up follow VoxelUtilHashSDF.h
// (/root/reference/FriedLiver/Source/DepthSensing), cited per function.
#pragma once
#include "bf_device.h"

namespace bf {

constexpr int VOL_BS = BF_SDF_BLOCK_SIZE;

struct Vol {
    const bf_hash_entry* hash; const bf_voxel* vox;
    uint32_t numBuckets, maxChain;
    float voxelSize;
};
BF_DEV uint32_t hashPos(uint32_t numBuckets, i3 b) {          // VoxelUtilHashSDF.h:226-234
    const uint32_t h = ((uint32_t)b.x * 73856093u) ^ ((uint32_t)b.y * 19349669u) ^ ((uint32_t)b.z * 83492791u);
    return h % numBuckets;
}
BF_DEV i3 worldToVirtualVoxelPos(float voxelSize, f3 pos) {   // :283-287
    const f3 p = pos / voxelSize;
    i3 r;
    r.x = f2i(p.x + (float)sgn(p.x) * 0.5f); r.y = f2i(p.y + (float)sgn(p.y) * 0.5f); r.z = f2i(p.z + (float)sgn(p.z) * 0.5f);
    return r;
}
BF_DEV i3 voxelToBlock(i3 v) {                                 // :290-299
    if (v.x < 0) v.x -= VOL_BS - 1;
    if (v.y < 0) v.y -= VOL_BS - 1;
    if (v.z < 0) v.z -= VOL_BS - 1;
    i3 r; r.x = v.x / VOL_BS; r.y = v.y / VOL_BS; r.z = v.z / VOL_BS;
    return r;
}
BF_DEV int localIndex(i3 v) {                                  // virtualVoxelPosToLocalSDFBlockIndex :347-358
    int x = v.x % VOL_BS, y = v.y % VOL_BS, z = v.z % VOL_BS;
    if (x < 0) x += VOL_BS;
    if (y < 0) y += VOL_BS;
    if (z < 0) z += VOL_BS;
    return z * VOL_BS * VOL_BS + y * VOL_BS + x;
}
// getHashEntryForSDFBlockPos :441-485 -> ptr or FREE_ENTRY
BF_DEV int32_t findBlock(const Vol& v, i3 b) {
    const uint32_t h = hashPos(v.numBuckets, b), hp = h * BF_HASH_BUCKET_SIZE;
    const int4* e4 = reinterpret_cast<const int4*>(v.hash);
#pragma unroll
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        const int4 a = e4[(size_t)(hp + j) * 2];
        if (a.x == b.x && a.y == b.y && a.z == b.z && a.w != BF_FREE_ENTRY) return a.w;
    }
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1, total = BF_HASH_BUCKET_SIZE * v.numBuckets;
    uint32_t off = v.hash[last].offset;
    if (off == 0) return BF_FREE_ENTRY;
    uint32_t i = (last + off) % total;
    for (uint32_t it = 1; it < v.maxChain; ++it) {
        const int4 a = e4[(size_t)i * 2];
        if (a.x == b.x && a.y == b.y && a.z == b.z && a.w != BF_FREE_ENTRY) return a.w;
        off = v.hash[i].offset;
        if (off == 0) break;
        i = (last + off) % total;
    }
    return BF_FREE_ENTRY;
}
struct Vx { float sdf, weight; uint32_t color; };
BF_DEV Vx getVoxel(const Vol& v, f3 worldPos) {                // :407-417 (a missing block reads as the zero voxel)
    const i3 vp = worldToVirtualVoxelPos(v.voxelSize, worldPos);
    const int32_t ptr = findBlock(v, voxelToBlock(vp));
    Vx r; r.sdf = 0.0f; r.weight = 0.0f; r.color = 0u;
    if (ptr != BF_FREE_ENTRY) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(v.vox + ((size_t)(uint32_t)ptr + (uint32_t)localIndex(vp)));
        r.sdf = __uint_as_float(p[0]); r.weight = __uint_as_float(p[1]); r.color = p[2];
    }
    return r;
}
BF_DEV float fracf_(float x) { return x - floorf(x); }

}  // namespace bf
