// Voxel-hashed TSDF for gfx950: reset / alloc / compactify / integrate / de-integrate / GC.
// Replaces DepthSensing/CUDASceneRepHashSDF.{h,cu} + VoxelUtilHashSDF.h of the reference
// (paths relative to /root/reference/FriedLiver/Source) behind the bf_scene_* C ABI.
//
// Design:
//  * memory layout of d_hash / d_SDFBlocks / d_heap / d_hashCompactified is the reference's
//    (32-B HashEntry stride, 12-B AoS voxels, ptr = block*512) so downstream consumers of
//    HashDataStruct (ray cast, marching cubes) keep working;
//  * allocation is lock-free and run-to-run deterministic: one pass over the depth pixels
//    collects the missing block keys into a 64-bit CAS de-dup set and scatters them into 256
//    bins by home-bucket range; each bin is sorted in LDS by (home bucket, key) and inserted
//    with rank-based slot / heap assignment; bucket-full keys are chained by one serial tail
//    kernel.  No bucket mutex, no host fixed-point loop, no D2H read-back;
//  * the frustum list is built from an incremental list of allocated blocks (16 B each)
//    instead of scanning all numBuckets*4 hash slots; its length never leaves the device —
//    the voxel-update kernel is a persistent grid-stride loop over a device-side count.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "bf_device.h"
#include "bf_internal.h"
#include "../../include/bf_comm.h"

using namespace bf;

namespace {

constexpr int BS = BF_SDF_BLOCK_SIZE;
constexpr int VOX = BS * BS * BS;
constexpr uint32_t NBINS = 256;        // one LDS sort workgroup per bin (~1 per CU); 1024 lighter bins (256 threads, 17 KB) measured 2 % slower
constexpr uint32_t BINCAP = 4096;      // records per bin (64 KB of LDS when sorting)
constexpr uint32_t OVCAP = 4096;       // bucket-full keys per alloc handled by the tail
constexpr uint32_t TILE = 1024;        // entries per ordered-compaction tile
constexpr uint64_t EMPTY64 = ~0ull;
constexpr int KEYLIM = 1 << 20;

struct AllocRec { uint64_t key; int32_t ptr; uint32_t pad; };          // 16 B
struct BinRec { uint64_t key; uint32_t bucket; uint32_t aux; };        // 16 B

enum Stat { ST_DROPPED = 0, ST_ERROR = 1, ST_STUCK = 2, ST_TICKET = 3, ST_COUNT = 4 };      // ST_STUCK: de-dup slots of keys that found their bin full (released by k_alloc_finish)
enum ErrBits { ERR_BIN_OVERFLOW = 1, ERR_DEDUPE_FULL = 2, ERR_OV_OVERFLOW = 4, ERR_GC_MISSING = 8, ERR_LIST_FULL = 16 };

struct Dev {
    bf_hash_entry* hash;
    uint32_t* heap;
    uint32_t* heapCounter;
    bf_voxel* vox;
    bf_hash_entry* compact;
    uint32_t* compactSrc;
    int32_t* compactCount;
    AllocRec* allocList;
    AllocRec* allocListAlt;
    uint32_t* allocCount;
    uint32_t* allocSnap;       // length of the allocated-block list this operator's compaction covers: written behind the operator's allocation (the next operator's
                               // allocation may append to the list while this one's lists are still being built on another stream)
    uint64_t* dedupe;
    uint32_t dedupeMask;
    BinRec* bins;
    uint32_t* binCount;
    BinRec* overflow;
    uint32_t* overflowCount;
    uint32_t* stuckSlots;
    uint32_t* tileCounts;
    uint32_t* stats;
    unsigned long long* occSum;      // [0] sum of frustum-list lengths per OPERATOR (a fused launch counts its two lists), [1] / [2] sum of list lengths per LAUNCH of the plain / the fused kernel (the union list once) over the timed voxel-update launches
};

struct Frame {          // per-call constants (kernarg => scalar loads)
    m44 T, Tinv;
    bf_depth_camera_params cam;
    uint32_t numBuckets;
    uint32_t maxChain;
    uint32_t numSDFBlocks;
    float voxelSize;
    float maxIntegrationDistance;
    float truncScale;
    float truncation;
    uint32_t shardLo, shardHi;       // owned home-bucket range (whole table when the volume is not sharded)
    float weightMax;
};

// ---------------------------------------------------------------------------------------
// integer maps (bit-exact parity targets)
// ---------------------------------------------------------------------------------------
// VoxelUtilHashSDF.h:226-234; m_hashNumBuckets is unsigned so `%` is an unsigned modulo.
BF_HD uint32_t hashPos(uint32_t numBuckets, i3 v) {
    uint32_t h = ((uint32_t)v.x * 73856093u) ^ ((uint32_t)v.y * 19349669u) ^ ((uint32_t)v.z * 83492791u);
    return h % numBuckets;
}
BF_HD bool keyable(i3 b) {
    return b.x >= -KEYLIM && b.x < KEYLIM && b.y >= -KEYLIM && b.y < KEYLIM && b.z >= -KEYLIM && b.z < KEYLIM;
}
BF_HD uint64_t packKey(i3 b) {
    return ((uint64_t)(uint32_t)(b.z + KEYLIM) << 42) | ((uint64_t)(uint32_t)(b.y + KEYLIM) << 21) |
           (uint64_t)(uint32_t)(b.x + KEYLIM);
}
BF_HD i3 unpackKey(uint64_t k) {
    i3 r;
    r.x = (int)(k & 0x1FFFFF) - KEYLIM;
    r.y = (int)((k >> 21) & 0x1FFFFF) - KEYLIM;
    r.z = (int)((k >> 42) & 0x1FFFFF) - KEYLIM;
    return r;
}
BF_HD i3 worldToVirtualVoxelPos(float voxelSize, f3 pos) {     // :283-287
    f3 p = pos / voxelSize;
    i3 r;
    r.x = f2i(p.x + (float)sgn(p.x) * 0.5f);
    r.y = f2i(p.y + (float)sgn(p.y) * 0.5f);
    r.z = f2i(p.z + (float)sgn(p.z) * 0.5f);
    return r;
}
BF_HD i3 virtualVoxelPosToSDFBlock(i3 v) {                     // :290-299
    if (v.x < 0) v.x -= BS - 1;
    if (v.y < 0) v.y -= BS - 1;
    if (v.z < 0) v.z -= BS - 1;
    i3 r; r.x = v.x / BS; r.y = v.y / BS; r.z = v.z / BS;
    return r;
}
BF_HD f3 SDFBlockToWorld(float voxelSize, i3 b) {
    return mk3((float)(b.x * BS), (float)(b.y * BS), (float)(b.z * BS)) * voxelSize;
}
BF_HD bool blockInFrustumT(const m44& Tinv, const bf_depth_camera_params& cam, float voxelSize, i3 b) {              // :322-326, DepthCameraUtil.h:97-142
    f3 w = SDFBlockToWorld(voxelSize, b);
    const float off = voxelSize * 0.5f * ((float)BS - 1.0f);
    w = w + mk3(off, off, off);
    f3 pc = xform(Tinv, w);
    const float sx = pc.x * cam.fx / pc.z + cam.mx;
    const float sy = pc.y * cam.fy / pc.z + cam.my;
    const float wm1 = (float)cam.m_imageWidth - 1.0f, hm1 = (float)cam.m_imageHeight - 1.0f;
    float px = (2.0f * sx - wm1) / wm1;
    float py = (hm1 - 2.0f * sy) / hm1;
    float pz = (pc.z - cam.m_sensorDepthWorldMin) / (cam.m_sensorDepthWorldMax - cam.m_sensorDepthWorldMin);
    px *= 0.95f; py *= 0.95f; pz *= 0.95f;
    return !(px < -1.0f || px > 1.0f || py < -1.0f || py > 1.0f || pz < 0.0f || pz > 1.0f);
}
BF_HD bool blockInFrustum(const Frame& f, i3 b) { return blockInFrustumT(f.Tinv, f.cam, f.voxelSize, b); }

// read-only lookup, VoxelUtilHashSDF.h:441-485.  The four entries of the home bucket and its chain head are loaded unconditionally
// (one memory round trip instead of up to five dependent ones: next to the voxel kernel of the other stream a round trip costs 2-3 us).
BF_DEV bool blockPresent(const Dev& d, const Frame& f, i3 b, uint32_t h) {
    const uint32_t hp = h * BF_HASH_BUCKET_SIZE;
    const int4* e4 = reinterpret_cast<const int4*>(d.hash);
    int4 a[BF_HASH_BUCKET_SIZE];
#pragma unroll
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) a[j] = e4[(size_t)(hp + j) * 2];                 // pos.xyz, ptr
    const uint32_t lastOffset = d.hash[hp + BF_HASH_BUCKET_SIZE - 1].offset;
    bool hit = false;
#pragma unroll
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) hit = hit || (a[j].x == b.x && a[j].y == b.y && a[j].z == b.z && a[j].w != BF_FREE_ENTRY);
    if (hit) return true;
    if (lastOffset == 0) return false;
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1;
    const uint32_t total = BF_HASH_BUCKET_SIZE * f.numBuckets;
    uint32_t i = (last + lastOffset) % total;
    for (uint32_t it = 1; it < f.maxChain; ++it) {
        const int4 c = e4[(size_t)i * 2];
        const uint32_t off = d.hash[i].offset;
        if (c.x == b.x && c.y == b.y && c.z == b.z && c.w != BF_FREE_ENTRY) return true;
        if (off == 0) break;
        i = (last + off) % total;
    }
    return false;
}

// ---------------------------------------------------------------------------------------
// reset
// ---------------------------------------------------------------------------------------
__global__ void k_reset(Dev d, uint32_t numSDFBlocks, uint32_t numEntries, uint32_t dedupeSize) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = gid; i < numSDFBlocks; i += stride) d.heap[i] = numSDFBlocks - i - 1;   // .cu:38
    uint4* h4 = reinterpret_cast<uint4*>(d.hash);
    for (uint32_t i = gid; i < numEntries; i += stride) {                                       // .cu:47-55
        h4[(size_t)i * 2] = make_uint4(0, 0, 0, (uint32_t)BF_FREE_ENTRY);
        h4[(size_t)i * 2 + 1] = make_uint4(0, 0, 0, 0);
    }
    for (uint32_t i = gid; i < dedupeSize; i += stride) d.dedupe[i] = EMPTY64;
    if (gid < NBINS) d.binCount[gid] = 0;
    if (gid < ST_COUNT) d.stats[gid] = 0;
    if (gid == 0) {
        d.heapCounter[0] = numSDFBlocks - 1;                                                   // .cu:33
        d.compactCount[0] = 0;
        d.allocCount[0] = 0;
        d.overflowCount[0] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// alloc, pass A: depth pixels -> missing block keys -> de-dup set -> bins
//   (CUDASceneRepHashSDF.cu:165-251 DDA;  one wave = one 8x8 pixel tile)
// ---------------------------------------------------------------------------------------
BF_DEV void emitCandidate(const Dev& d, const Frame& f, i3 b) {
    if (!keyable(b)) return;
    const uint32_t h = hashPos(f.numBuckets, b);
    if (h < f.shardLo || h >= f.shardHi) return;      // hash-bucket sharding: this volume owns the home buckets [shardLo, shardHi)
    if (blockPresent(d, f, b, h)) return;
    const uint64_t key = packKey(b);
    // 64-bit CAS claim in the open-addressing de-dup set (linear probing)
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & d.dedupeMask;
    for (uint32_t probe = 0; probe <= d.dedupeMask; ++probe) {
        const unsigned long long old =
            atomicCAS(reinterpret_cast<unsigned long long*>(&d.dedupe[slot]), (unsigned long long)EMPTY64, (unsigned long long)key);
        if (old == key) return;                 // somebody already queued this block
        if (old == EMPTY64) {
            const uint32_t bin = (uint32_t)(((uint64_t)h * NBINS) / f.numBuckets);
            const uint32_t pos = atomicAdd(&d.binCount[bin], 1u);
            if (pos < BINCAP) {
                BinRec r; r.key = key; r.bucket = h; r.aux = slot;
                d.bins[(size_t)bin * BINCAP + pos] = r;
            } else {
                // The slot stays claimed while other lanes are still probing (releasing it here would cut their linear-probe chains
                // and let a key be claimed twice => two heap blocks for one key); k_alloc_finish releases the recorded slots.
                atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW);
                const uint32_t q = atomicAdd(&d.stats[ST_STUCK], 1u);
                if (q < OVCAP) d.stuckSlots[q] = slot;
            }
            return;
        }
        slot = (slot + 1) & d.dedupeMask;
    }
    atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_DEDUPE_FULL);
}

// One wave = one 8x8 pixel tile.  The DDA itself touches no global memory: every block a ray steps through goes into a small LDS set
// of the wave (neighbouring pixels walk through the same 8^3 block most of the time - a block is ~9 pixels wide at 2 m - so a tile
// sees ~30 distinct blocks in ~600 lane-steps; a lane whose left or upper neighbour holds the same block this step does not even try).
// Only the distinct blocks are then tested against the frustum and looked up in the hash table, one lane per block, all look-ups of a
// wave in flight together.  (The first version looked every block up inside the DDA loop: one dependent ~2 us global round trip per
// step made the kernel a 15-deep latency chain, 40-75 us per launch.)  Queuing a block is idempotent and the bins are sorted before
// they are placed, so the hash table does not depend on the order in which candidates are found.
// PREP_PRIO: the allocation / list kernels are a few thousand short, latency-bound waves that share each SIMD with up to seven waves of
// the voxel update of the previous operator (other stream), which saturates VALU issue; at equal wave priority they ran 2-4x slower
// than alone (23 -> 51 us, 22 -> 54 us) and the allocation chain paced the loop.  s_setprio 3 lets them issue first.
constexpr uint32_t WSET = 256;        // slots of a wave's block set (open addressing, load <= 0.5)
constexpr uint32_t WLIST = 128;       // distinct blocks buffered per wave before they are looked up

BF_DEV bool waveSetInsert(unsigned long long* set, uint64_t key) {
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (WSET - 1);
    for (uint32_t probe = 0; probe < WSET; ++probe) {
        const unsigned long long old = atomicCAS(&set[slot], (unsigned long long)EMPTY64, (unsigned long long)key);
        if (old == EMPTY64) return true;
        if (old == key) return false;
        slot = (slot + 1) & (WSET - 1);
    }
    return false;       // cannot happen: the set is flushed at half load
}

// Multi-GPU allocation (bf_scene_alloc_collect / _ingest): with the volume sharded by home bucket every rank would have to
// march ALL pixels to find its own blocks - the part of an operator that does not shrink with the number of GPUs.  Instead each rank marches
// a band of the pixel tiles and COLLECTS the distinct in-frustum block keys it meets (no ownership test, no table lookup), the ranks exchange
// their key lists (one all-gather), and every rank INGESTS all lists: ownership test, table lookup, de-dup set, bins - the second half of
// emitCandidate - followed by the usual placement.  Queuing a key is idempotent and bins are sorted before placement, so the table does not
// depend on who found a block or in which order.
struct Collect { unsigned long long* keys; uint32_t* slots; uint32_t* count; uint32_t capacity; uint32_t tile0, tile1; };
// the march reads every pixel's depth anyway: when the operator's update wants the frame as 8-byte {depth, colour} texels (fast contract), the march writes them
// on the way - one launch less on the allocation stream than a separate interleave pass (texel == null: no)
struct TexelOut { const uint32_t* color; uint2* texel; };

BF_DEV void collectCandidate(const Dev& d, const Collect& c, i3 b) {
    if (!keyable(b)) return;
    const uint64_t key = packKey(b);
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & d.dedupeMask;      // the de-dup set, borrowed: one entry per distinct key of this rank's band
    for (uint32_t probe = 0; probe <= d.dedupeMask; ++probe) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&d.dedupe[slot]), (unsigned long long)EMPTY64, (unsigned long long)key);
        if (old == key) return;
        if (old == EMPTY64) {
            const uint32_t pos = atomicAdd(c.count, 1u);
            if (pos < c.capacity) { c.keys[pos] = key; c.slots[pos] = slot; }
            else { atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); d.dedupe[slot] = EMPTY64; }       // (nobody probes past a key that is being withdrawn: the collect kernel only ever inserts)
            return;
        }
        slot = (slot + 1) & d.dedupeMask;
    }
    atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_DEDUPE_FULL);
}

// Batched allocation (bf_scene_run_batch): ONE march over the frames of all operators of a batch.  The march touches neither the hash table nor the
// bins: every distinct in-frustum block key of operator `op` is claimed in the batch's own key set together with the set of operators that need it.  The
// FIRST of them allocates it in the serial order (operators before it must not see the block, operators after it find it present); the others matter only
// when that allocation fails (heap or collision window exhausted): the serial operators would each try again, and so does the batch's replay.  The table
// look-up, the ownership test and the binning per operator follow in k_batch_bin, behind whatever still has to free table entries (garbage collection).
struct BatchSink {
    unsigned long long* set; uint32_t mask;      // 64-bit keys, open addressing (EMPTY64 = free)
    uint32_t* opMask;                            // per slot: bit k = operator k needs the key
    uint32_t* list; uint32_t* count; uint32_t cap;   // slots claimed by this batch, in arrival order (the order reaches no result: k_batch_place sorts)
    uint32_t* flags;                             // [0] bit 0: a home bucket cannot take all its new keys (k_batch_bin), bit 1: the slot list overflowed
    uint32_t op;
};

BF_DEV void claimCandidate(const Dev& d, const BatchSink& bs, i3 b) {
    if (!keyable(b)) return;
    const uint64_t key = packKey(b);
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & bs.mask;
    for (uint32_t probe = 0; probe <= bs.mask; ++probe) {
        const unsigned long long old = atomicCAS(&bs.set[slot], (unsigned long long)EMPTY64, (unsigned long long)key);
        if (old == EMPTY64) {
            const uint32_t pos = atomicAdd(bs.count, 1u);
            if (pos < bs.cap) bs.list[pos] = slot;
            else { atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); atomicOr(bs.flags, 2u); }      // the set is cleared as a whole behind this batch
        }
        if (old == EMPTY64 || old == key) { atomicOr(&bs.opMask[slot], 1u << bs.op); return; }
        slot = (slot + 1) & bs.mask;
    }
    atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_DEDUPE_FULL);
}

// SINK 0: queue the missing keys this volume owns (emitCandidate), 1: collect the distinct in-frustum keys of a band (collectCandidate), 2: claim for a batch
template <int SINK>
BF_DEV void waveSetFlush(const Dev& d, const Frame& f, const Collect& c, const BatchSink& bs, unsigned long long* set, const unsigned long long* list, uint32_t n, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the list entries were written by other lanes of this wave
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        if (i < n) {
            const i3 b = unpackKey(list[i]);
            if (blockInFrustum(f, b)) { if (SINK == 1) collectCandidate(d, c, b); else if (SINK == 2) claimCandidate(d, bs, b); else emitCandidate(d, f, b); }
        }
    }
    for (uint32_t i = lane; i < WSET; i += 64) set[i] = EMPTY64;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// the march of one 8x8 pixel tile by one wave (set / list: the wave's LDS scratch); marches == false: the tile's texels only (an operator that does not allocate)
template <int SINK>
BF_DEV void marchTile(const Dev& d, const Frame& f, const float* __restrict__ depth, const Collect& c, const TexelOut& tx, const BatchSink& bs, uint32_t tile, bool marches,
                      unsigned long long* set, unsigned long long* list) {
    constexpr bool COLLECT = SINK == 1;
    const uint32_t W = f.cam.m_imageWidth, H = f.cam.m_imageHeight;
    const uint32_t tilesX = (W + 7) / 8;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i = lane; i < WSET; i += 64) set[i] = EMPTY64;
    const uint32_t x = (tile % tilesX) * 8 + (lane & 7);
    const uint32_t y = (tile / tilesX) * 8 + (lane >> 3);
    bool alive = x < W && y < H && (!COLLECT || tile < c.tile1);
    const float dd = alive ? depth[(size_t)y * W + x] : BF_MINF;
    if (tx.texel != nullptr && alive) tx.texel[(size_t)y * W + x] = make_uint2(__float_as_uint(dd), tx.color[(size_t)y * W + x]);
    if (!marches) return;       // wave-uniform
    if (dd == BF_MINF || dd == 0.0f) alive = false;
    if (dd >= f.maxIntegrationDistance) alive = false;
    const float t = f.truncation + f.truncScale * dd;
    const float minDepth = fminf(f.maxIntegrationDistance, dd - t);
    const float maxDepth = fminf(f.maxIntegrationDistance, dd + t);
    if (minDepth >= maxDepth) alive = false;
    const float kx = ((float)x - f.cam.mx) / f.cam.fx;
    const float ky = ((float)y - f.cam.my) / f.cam.fy;
    const f3 rayMin = xform(f.T, mk3(minDepth * kx, minDepth * ky, minDepth));
    const f3 rayMax = xform(f.T, mk3(maxDepth * kx, maxDepth * ky, maxDepth));
    const f3 dv = rayMax - rayMin;
    const float invLen = 1.0f / sqrtf(dot3(dv, dv));
    const f3 rayDir = dv * invLen;
    i3 cur = virtualVoxelPosToSDFBlock(worldToVirtualVoxelPos(f.voxelSize, rayMin));
    const i3 end = virtualVoxelPosToSDFBlock(worldToVirtualVoxelPos(f.voxelSize, rayMax));
    const f3 step = mk3((float)sgn(rayDir.x), (float)sgn(rayDir.y), (float)sgn(rayDir.z));
    i3 nb;
    nb.x = cur.x + f2i(fminf(fmaxf(step.x, 0.0f), 1.0f));
    nb.y = cur.y + f2i(fminf(fmaxf(step.y, 0.0f), 1.0f));
    nb.z = cur.z + f2i(fminf(fmaxf(step.z, 0.0f), 1.0f));
    const float hv = 0.5f * f.voxelSize;
    const f3 boundary = SDFBlockToWorld(f.voxelSize, nb) - mk3(hv, hv, hv);
    f3 tMax = mk3((boundary.x - rayMin.x) / rayDir.x, (boundary.y - rayMin.y) / rayDir.y, (boundary.z - rayMin.z) / rayDir.z);
    f3 tDelta = mk3((step.x * (float)BS * f.voxelSize) / rayDir.x, (step.y * (float)BS * f.voxelSize) / rayDir.y,
                    (step.z * (float)BS * f.voxelSize) / rayDir.z);
    i3 bound;
    bound.x = f2i((float)end.x + step.x);
    bound.y = f2i((float)end.y + step.y);
    bound.z = f2i((float)end.z + step.z);
    if (rayDir.x == 0.0f) { tMax.x = BF_PINF; tDelta.x = BF_PINF; }
    if (boundary.x - rayMin.x == 0.0f) { tMax.x = BF_PINF; tDelta.x = BF_PINF; }
    if (rayDir.y == 0.0f) { tMax.y = BF_PINF; tDelta.y = BF_PINF; }
    if (boundary.y - rayMin.y == 0.0f) { tMax.y = BF_PINF; tDelta.y = BF_PINF; }
    if (rayDir.z == 0.0f) { tMax.z = BF_PINF; tDelta.z = BF_PINF; }
    if (boundary.z - rayMin.z == 0.0f) { tMax.z = BF_PINF; tDelta.z = BF_PINF; }
    uint32_t uniq = 0;                      // wave-uniform: blocks buffered in `list`
    for (unsigned iter = 0; iter < 1024; ++iter) {
        if (!__any((int)alive)) break;
        const int live = alive ? 1 : 0;
        const int lx = __shfl_up(cur.x, 1, 64), ly = __shfl_up(cur.y, 1, 64), lz = __shfl_up(cur.z, 1, 64), ll = __shfl_up(live, 1, 64);
        const int ux = __shfl_up(cur.x, 8, 64), uy = __shfl_up(cur.y, 8, 64), uz = __shfl_up(cur.z, 8, 64), ul = __shfl_up(live, 8, 64);
        const bool sameLeft = (lane & 7) != 0 && ll && lx == cur.x && ly == cur.y && lz == cur.z;
        const bool sameUp = lane >= 8 && ul && ux == cur.x && uy == cur.y && uz == cur.z;
        bool fresh = false;
        uint64_t key = 0;
        if (alive && !sameLeft && !sameUp && keyable(cur)) { key = packKey(cur); fresh = waveSetInsert(set, key); }
        const unsigned long long mask = __ballot((int)fresh);
        if (fresh) list[uniq + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = key;
        uniq += (uint32_t)__popcll(mask);
        if (uniq > WLIST - 64) { waveSetFlush<SINK>(d, f, c, bs, set, list, uniq, lane); uniq = 0; }
        if (alive) {
            if (tMax.x < tMax.y && tMax.x < tMax.z) {
                cur.x = f2i((float)cur.x + step.x);
                if (cur.x == bound.x) alive = false;
                tMax.x += tDelta.x;
            } else if (tMax.z < tMax.y) {
                cur.z = f2i((float)cur.z + step.z);
                if (cur.z == bound.z) alive = false;
                tMax.z += tDelta.z;
            } else {
                cur.y = f2i((float)cur.y + step.y);
                if (cur.y == bound.y) alive = false;
                tMax.y += tDelta.y;
            }
        }
    }
    if (uniq) waveSetFlush<SINK>(d, f, c, bs, set, list, uniq, lane);
}

template <bool COLLECT>
__global__ __launch_bounds__(256) void k_alloc_candidates(Dev d, Frame f, const float* __restrict__ depth, Collect c, TexelOut tx) {
    __builtin_amdgcn_s_setprio(3);          // see PREP_PRIO
    __shared__ unsigned long long setAll[4][WSET];
    __shared__ unsigned long long listAll[4][WLIST];
    if (!COLLECT && blockIdx.x == 0 && threadIdx.x == 0) { d.compactCount[0] = 0; d.compactCount[1] = 0; }      // the operator's list: the placement kernel behind this one appends to it
    const uint32_t tile = (COLLECT ? c.tile0 : 0u) + blockIdx.x * 4 + (threadIdx.x >> 6);
    marchTile<COLLECT ? 1 : 0>(d, f, depth, c, tx, BatchSink{}, tile, true, setAll[threadIdx.x >> 6], listAll[threadIdx.x >> 6]);
}

// after a collect pass: give the borrowed de-dup entries back
__global__ void k_collect_release(Dev d, Collect c) {
    const uint32_t n = min(c.count[0], c.capacity);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d.dedupe[c.slots[i]] = EMPTY64;
}

// ingest one rank's key list: the second half of emitCandidate for every key (ownership, lookup, de-dup set, bin)
__global__ __launch_bounds__(256) void k_alloc_ingest(Dev d, Frame f, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ count, uint32_t capacity) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n = min(count[0], capacity);
    if (count[0] > capacity && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW);      // the collecting rank dropped keys: every rank that ingests its list reports it
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) emitCandidate(d, f, unpackKey(keys[i]));
}

// ---------------------------------------------------------------------------------------
// LDS bitonic sort of (bucket, key, aux) records, ascending by (bucket, key)
// ---------------------------------------------------------------------------------------
struct SortLds {
    uint64_t key[BINCAP];
    uint32_t bucket[BINCAP];
    uint32_t aux[BINCAP];
};

BF_DEV uint32_t nextPow2(uint32_t v) {
    uint32_t p = 64;
    while (p < v) p <<= 1;
    return p;
}

BF_DEV void ldsBitonicSort(SortLds& s, uint32_t npad) {
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                const bool up = ((lo & k) == 0);
                const uint32_t bl = s.bucket[lo], bh = s.bucket[hi];
                const uint64_t kl = s.key[lo], kh = s.key[hi];
                const bool gt = (bl != bh) ? (bl > bh) : (kl > kh);
                if (gt == up) {
                    s.bucket[lo] = bh; s.bucket[hi] = bl;
                    s.key[lo] = kh; s.key[hi] = kl;
                    const uint32_t a = s.aux[lo]; s.aux[lo] = s.aux[hi]; s.aux[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// AGENT: the records were written by OTHER workgroups of this launch (the overflow list in a placement's tail): read at the scope they were stored at.  (Round 5: a
// plain load behind the tail's agent-scope acquire fence can return what this XCD's L2 held before - measured on the matcher's and the key list's last-workgroup
// hand-offs, where 3 of 16 runs of the frame loop gave another trajectory until the finisher's loads became agent-scope loads: profiles/r05_determinism.md.)
template <bool AGENT = false>
BF_DEV uint32_t loadBinSorted(SortLds& s, const BinRec* recs, uint32_t n) {
    const uint32_t npad = nextPow2(n);
    for (uint32_t i = threadIdx.x; i < npad; i += blockDim.x) {
        if (i < n) {
            BinRec r;
            if (AGENT) {
                const unsigned long long* q = reinterpret_cast<const unsigned long long*>(recs + i);
                const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r.key = a; r.bucket = (uint32_t)b; r.aux = (uint32_t)(b >> 32);
            } else r = recs[i];
            s.key[i] = r.key; s.bucket[i] = r.bucket; s.aux[i] = r.aux;
        }
        else { s.key[i] = EMPTY64; s.bucket[i] = 0xFFFFFFFFu; s.aux[i] = 0; }
    }
    __syncthreads();
    ldsBitonicSort(s, npad);
    return npad;
}

// sum over bins b < limit of min(binCount[b], BINCAP); result broadcast to the block
BF_DEV uint32_t binPrefix(const uint32_t* binCount, uint32_t limit, uint32_t* scratch) {
    uint32_t v = 0;
    for (uint32_t b = threadIdx.x; b < limit; b += blockDim.x) v += min(binCount[b], BINCAP);
    v = (uint32_t)wave_sum_i((int)v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) total += scratch[w];
    __syncthreads();
    return total;
}

// ---------------------------------------------------------------------------------------
// the operator's block list (frustum list, or union list of a fused re-integration)
//   LISTS 0: blocks in the frustum of f (replaces compactifyHashAllInOneKernel, .cu:324-366);  LISTS 1: every live block (list maintenance after GC);
//   LISTS 2: blocks in the frustum of f (bit 0 of the flags) or of fo (bit 1)
// ---------------------------------------------------------------------------------------
template <int MODE>
BF_DEV uint32_t keepRec(const Frame& f, const Frame& fo, const AllocRec& r) {
    if (r.ptr == BF_FREE_ENTRY) return 0u;
    if (MODE == 1) return 1u;
    const i3 b = unpackKey(r.key);
    if (MODE == 0) return blockInFrustum(f, b) ? 1u : 0u;
    return (blockInFrustum(f, b) ? 1u : 0u) | (blockInFrustum(fo, b) ? 2u : 0u);
}

BF_DEV void listWrite(const Dev& d, uint32_t pos, uint64_t key, int32_t ptr, uint32_t flags, uint32_t src) {
    const i3 b = unpackKey(key);
    uint4* o4 = reinterpret_cast<uint4*>(d.compact + pos);
    o4[0] = make_uint4((uint32_t)b.x, (uint32_t)b.y, (uint32_t)b.z, (uint32_t)ptr);
    o4[1] = make_uint4(flags, 0u, 0u, 0u);       // the offset field doubles as the frustum flags of a union list
    d.compactSrc[pos] = src;
}

// every lane of the (converged) wave calls this; lanes with keep != 0 append their block: one reservation per wave
template <int MODE>
BF_DEV void listAppendWave(const Dev& d, uint32_t keep, uint64_t key, int32_t ptr, uint32_t src) {
    const unsigned long long m = __ballot(keep != 0u);
    if (m == 0ull) return;                                      // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t ob = 0;
    if (MODE == 2) ob = (uint32_t)wave_sum_i((int)((keep & 1u) + ((keep >> 1) & 1u)));
    uint32_t base = 0;
    if ((int)lane == leader) {
        base = atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), (uint32_t)__popcll(m));
        if (MODE == 2) atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, ob);
    }
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (keep) listWrite(d, base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), key, ptr, MODE == 2 ? keep : 0u, src);
}

// every thread of the (converged) 256-thread workgroup calls this with at most one block: one reservation per workgroup, the order inside the workgroup kept
template <int MODE>
BF_DEV void listAppendBlock(const Dev& d, uint32_t keep, uint64_t key, int32_t ptr, uint32_t src, uint32_t* wscan, uint32_t* sbase) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(keep != 0u);
    const uint32_t c = (uint32_t)__popcll(m);
    if (lane == 0) wscan[wave] = c;
    if (MODE == 2) { const uint32_t ob = (uint32_t)wave_sum_i((int)((keep & 1u) + ((keep >> 1) & 1u))); if (lane == 0 && ob) atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, ob); }
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = wscan[0] + wscan[1] + wscan[2] + wscan[3]; *sbase = tot ? atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), tot) : 0u; }
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; ++w) woff += wscan[w];
    if (keep) listWrite(d, *sbase + woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), key, ptr, MODE == 2 ? keep : 0u, src);
    __syncthreads();
}

// One pass over the tiles tile0, tile0 + tileStep, ... of the allocated-block list's first n entries (a tile = K entries per thread of the workgroup): every
// workgroup filters its tile, reserves a contiguous range of the list with one atomic add and writes its keepers there - tile ranges in arrival order, the
// order inside a tile kept.
template <int MODE, uint32_t K>
BF_DEV void listAppendTiles(const Dev& d, const Frame& f, const Frame& fo, uint32_t n, uint32_t tile0, uint32_t tileStep, uint32_t* wscan, uint32_t* sbase) {
    const uint32_t tileSize = K * 256u;
    const uint32_t numTiles = (n + tileSize - 1) / tileSize;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = tile0; tile < numTiles; tile += tileStep) {
        AllocRec recs[K];
        uint32_t keep[K];
        uint32_t c = 0, ob = 0;
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t i = tile * tileSize + threadIdx.x * K + k;
            keep[k] = 0u;
            if (i < n) { recs[k] = d.allocList[i]; keep[k] = keepRec<MODE>(f, fo, recs[k]); }
            c += keep[k] ? 1u : 0u;
            ob += (keep[k] & 1u) + ((keep[k] >> 1) & 1u);
        }
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
        if (lane == 63) wscan[wave] = incl;
        if (MODE == 2) { ob = (uint32_t)wave_sum_i((int)ob); if (lane == 0 && ob) atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, ob); }
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t tot = wscan[0] + wscan[1] + wscan[2] + wscan[3]; *sbase = tot ? atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), tot) : 0u; }
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wscan[w];
        uint32_t pos = *sbase + woff + incl - c;
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) {
            if (!keep[k]) continue;
            listWrite(d, pos, recs[k].key, recs[k].ptr, MODE == 2 ? keep[k] : 0u, tile * tileSize + threadIdx.x * K + k);
            ++pos;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// alloc, pass B: one workgroup per bin — sort, rank, place into home buckets
//   (serial-equivalent of allocBlock's in-bucket branch, VoxelUtilHashSDF.h:553-612)
// ---------------------------------------------------------------------------------------
// One launch for passes B and C: PLACE_WGS workgroups walk the bins (the usual bin holds a handful of new keys; a full bin - 4096 records,
// first frames of a scan - sorts in ~10 us), and the workgroup that finishes last runs the single-workgroup tail.  (As two kernels of
// 256 and 1 workgroups the passes cost two more dependent launches per operator, each of which queued behind the voxel kernels of the
// other stream for 10-30 us: the allocation chain, not the voxel update, paced the re-integration loop.)
constexpr uint32_t PLACE_WGS = 256;

// write-through (agent scope) 8-byte stores: what the tail workgroup reads of the other workgroups' output goes straight to memory, so
// the hand-off needs no L2 write-back (a release fence writes back the XCD's whole L2, which the voxel kernel of the other stream keeps
// full of dirty lines: measured 30-50 us per launch)
BF_DEV void storeThrough(void* p, uint64_t v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BF_DEV uint64_t pack2(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

template <int LISTS>
BF_DEV void placeBin(const Dev& d, const Frame& f, const Frame& fo, SortLds& s, uint32_t* scratch, int8_t* sel, uint32_t bin) {
    const uint32_t n = min(d.binCount[bin], BINCAP);
    if (n == 0) return;                       // block-uniform
    loadBinSorted(s, d.bins + (size_t)bin * BINCAP, n);
    const uint32_t base = binPrefix(d.binCount, bin, scratch);
    const uint32_t heapC = d.heapCounter[0];
    const uint32_t allocBase = d.allocCount[0];
    // blocks this pass may hand out: the free heap blocks, but never more than the allocated-block list can still record (holes left by
    // exhausted collision windows are compacted only by GC, so live + holes can reach the list capacity before the heap is empty)
    const uint32_t heapFree = min(heapC + 1u, f.numSDFBlocks - min(allocBase, f.numSDFBlocks));
    // phase 1 (reads only): rank among the bin's new keys of the same home bucket -> free slot
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint32_t h = s.bucket[idx];
        uint32_t rank = 0;
        while (rank < BF_HASH_BUCKET_SIZE && idx > rank && s.bucket[idx - rank - 1] == h) ++rank;
        int slot = -1;
        if (rank < BF_HASH_BUCKET_SIZE) {
            uint32_t seen = 0;
#pragma unroll
            for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
                const int32_t p = d.hash[h * BF_HASH_BUCKET_SIZE + j].ptr;
                if (p == BF_FREE_ENTRY) { if (seen == rank && slot < 0) slot = (int)j; ++seen; }
            }
        }
        sel[idx] = (int8_t)slot;
    }
    __syncthreads();
    // phase 2 (writes): disjoint slots, rank-ordered heap consumption.  (Whole waves walk the loop: the list append below is a wave operation.)
    for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t idx = i0 + threadIdx.x;
        uint32_t keep = 0; uint64_t key = 0; int32_t ptr = 0;
        const uint32_t gi = base + idx;
        if (idx < n) {
            key = s.key[idx];
            const uint32_t h = s.bucket[idx];
            d.dedupe[s.aux[idx]] = EMPTY64;
            if (gi >= heapFree) atomicAdd(&d.stats[ST_DROPPED], 1u);                    // heap exhausted
            else {
                ptr = (int32_t)(d.heap[heapC - gi] * (uint32_t)VOX);                    // consumeHeap :536-540
                uint64_t* ar = reinterpret_cast<uint64_t*>(d.allocList + allocBase + gi);      // AllocRec {key, ptr, pad}
                storeThrough(ar, key); storeThrough(ar + 1, pack2((uint32_t)ptr, 0u));
                const int slot = sel[idx];
                if (slot >= 0) {
                    const i3 b = unpackKey(key);
                    uint64_t* e = reinterpret_cast<uint64_t*>(d.hash + ((size_t)h * BF_HASH_BUCKET_SIZE + (uint32_t)slot));   // {pos.xyz, ptr, offset = NO_OFFSET :608, pad}
                    storeThrough(e, pack2((uint32_t)b.x, (uint32_t)b.y)); storeThrough(e + 1, pack2((uint32_t)b.z, (uint32_t)ptr));
                    storeThrough(e + 2, 0ull); storeThrough(e + 3, 0ull);
                    if (LISTS >= 0) { AllocRec r; r.key = key; r.ptr = ptr; r.pad = 0; keep = keepRec<(LISTS < 0 ? 0 : LISTS)>(f, fo, r); }
                } else {                                                                // the tail walks the collision window (and puts the block on the list)
                    const uint32_t ov = atomicAdd(d.overflowCount, 1u);
                    if (ov < OVCAP) { uint64_t* r = reinterpret_cast<uint64_t*>(d.overflow + ov); storeThrough(r, key); storeThrough(r + 1, pack2(h, gi)); }      // BinRec {key, bucket, aux}
                    else atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_OV_OVERFLOW);
                }
            }
        }
        if (LISTS >= 0) listAppendWave<(LISTS < 0 ? 0 : LISTS)>(d, keep, key, ptr, allocBase + gi);
    }
    __syncthreads();                          // s and sel are reused for the next bin
}

// A bin of at most 64 records (every bin once the scan is under way) is placed by ONE wave: bitonic sort in registers (lane
// exchanges, no LDS, no barriers), rank by lane exchange, and two memory round trips in all - {bin records, bin counts, counters} and
// {the four slots of the home bucket, the heap block}.  Same order and same decisions as placeBin.
template <int LISTS>
BF_DEV void placeBinWave(const Dev& d, const Frame& f, const Frame& fo, uint32_t n, BinRec r, uint32_t base, uint32_t heapC, uint32_t allocBase, uint32_t lane) {
    uint32_t bucket = lane < n ? r.bucket : 0xFFFFFFFFu;
    uint64_t key = lane < n ? r.key : EMPTY64;
    uint32_t aux = lane < n ? r.aux : 0u;
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)bucket, (int)j, 64), oa = (uint32_t)__shfl_xor((int)aux, (int)j, 64);
            const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)key, (int)j, 64), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)j, 64);
            const uint64_t ok = (uint64_t)olo | ((uint64_t)ohi << 32);
            const bool wantMin = ((lane & j) == 0) == ((lane & k) == 0);
            const bool ownGreater = bucket != ob ? bucket > ob : key > ok;
            const bool otherGreater = bucket != ob ? ob > bucket : ok > key;
            if (wantMin ? ownGreater : otherGreater) { bucket = ob; key = ok; aux = oa; }
        }
    }
    const uint32_t idx = lane;
    const uint32_t b1 = (uint32_t)__shfl_up((int)bucket, 1, 64), b2 = (uint32_t)__shfl_up((int)bucket, 2, 64);
    const uint32_t b3 = (uint32_t)__shfl_up((int)bucket, 3, 64), b4 = (uint32_t)__shfl_up((int)bucket, 4, 64);
    const uint32_t gi = base + idx;
    uint32_t keep = 0; int32_t ptr = 0;
    if (idx < n) {
        uint32_t rank = 0;
        if (idx >= 1 && b1 == bucket) { rank = 1; if (idx >= 2 && b2 == bucket) { rank = 2; if (idx >= 3 && b3 == bucket) { rank = 3; if (idx >= 4 && b4 == bucket) rank = 4; } } }
        const uint32_t heapFree = min(heapC + 1u, f.numSDFBlocks - min(allocBase, f.numSDFBlocks));
        int32_t p[BF_HASH_BUCKET_SIZE];
#pragma unroll
        for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) p[j] = d.hash[bucket * BF_HASH_BUCKET_SIZE + j].ptr;
        const uint32_t heapBlock = gi < heapFree ? d.heap[heapC - gi] : 0u;                                // consumeHeap :536-540
        int slot = -1;
        if (rank < BF_HASH_BUCKET_SIZE) {
            uint32_t seen = 0;
#pragma unroll
            for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j)
                if (p[j] == BF_FREE_ENTRY) { if (seen == rank && slot < 0) slot = (int)j; ++seen; }
        }
        d.dedupe[aux] = EMPTY64;
        if (gi >= heapFree) atomicAdd(&d.stats[ST_DROPPED], 1u);                                           // heap exhausted
        else {
            ptr = (int32_t)(heapBlock * (uint32_t)VOX);
            uint64_t* ar = reinterpret_cast<uint64_t*>(d.allocList + allocBase + gi);
            storeThrough(ar, key); storeThrough(ar + 1, pack2((uint32_t)ptr, 0u));
            if (slot >= 0) {
                const i3 b = unpackKey(key);
                uint64_t* e = reinterpret_cast<uint64_t*>(d.hash + ((size_t)bucket * BF_HASH_BUCKET_SIZE + (uint32_t)slot));
                storeThrough(e, pack2((uint32_t)b.x, (uint32_t)b.y)); storeThrough(e + 1, pack2((uint32_t)b.z, (uint32_t)ptr));
                storeThrough(e + 2, 0ull); storeThrough(e + 3, 0ull);
                if (LISTS >= 0) { AllocRec ar2; ar2.key = key; ar2.ptr = ptr; ar2.pad = 0; keep = keepRec<(LISTS < 0 ? 0 : LISTS)>(f, fo, ar2); }
            } else {                                                                                       // the tail walks the collision window (and puts the block on the list)
                const uint32_t ov = atomicAdd(d.overflowCount, 1u);
                if (ov < OVCAP) { uint64_t* o = reinterpret_cast<uint64_t*>(d.overflow + ov); storeThrough(o, key); storeThrough(o + 1, pack2(bucket, gi)); }
                else atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_OV_OVERFLOW);
            }
        }
    }
    if (LISTS >= 0) listAppendWave<(LISTS < 0 ? 0 : LISTS)>(d, keep, key, ptr, allocBase + gi);
}

// ---------------------------------------------------------------------------------------
// alloc, pass C: single workgroup tail — bucket-full keys walk the collision window in
// sorted order (VoxelUtilHashSDF.h:614-654), counters are committed, bins are recycled.
// ---------------------------------------------------------------------------------------
// M (keys queued in all bins), heapC, allocBase and stuckAll are inputs of the launch that every workgroup read in its first round trip (nothing changes them before
// this point); the only thing the tail has to fetch is what the other workgroups produced: the number of bucket-full keys.
template <int LISTS>
BF_DEV void placeTail(const Dev& d, const Frame& f, const Frame& fo, SortLds& s, uint32_t M, uint32_t heapC, uint32_t allocBase, uint32_t stuckAll) {
    const uint32_t nov = min(__hip_atomic_load(d.overflowCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), OVCAP);
    if (nov > 0) loadBinSorted<true>(s, d.overflow, nov);
    if (threadIdx.x == 0) {
        const uint32_t listRoom = f.numSDFBlocks - min(allocBase, f.numSDFBlocks);
        const uint32_t heapFree = min(heapC + 1u, listRoom);          // the same limit placeBin applied
        if (M > listRoom && listRoom < heapC + 1u) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_LIST_FULL);
        const uint32_t Mp = min(M, heapFree);
        const uint32_t total = BF_HASH_BUCKET_SIZE * f.numBuckets;
        uint32_t newCounter = heapC - Mp;
        uint32_t dropped = 0;
        for (uint32_t k = 0; k < nov; ++k) {
            const uint32_t h = s.bucket[k];
            const uint32_t gi = s.aux[k];
            const uint32_t last = (h + 1) * BF_HASH_BUCKET_SIZE - 1;
            const int32_t ptr = __hip_atomic_load(&d.allocList[allocBase + gi].ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool done = false;
            uint32_t maxIter = 0;
            int offset = 0;
            while (maxIter < f.maxChain) {
                offset++;
                const uint32_t i = (last + (uint32_t)offset) % total;
                if ((offset % BF_HASH_BUCKET_SIZE) == 0) continue;      // never a bucket's last slot :624
                if (__hip_atomic_load(&d.hash[i].ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == BF_FREE_ENTRY) {
                    const i3 b = unpackKey(s.key[k]);
                    d.hash[i].pos[0] = b.x; d.hash[i].pos[1] = b.y; d.hash[i].pos[2] = b.z;
                    d.hash[i].offset = __hip_atomic_load(&d.hash[last].offset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d.hash[i].ptr = ptr;
                    d.hash[last].offset = (uint32_t)offset;
                    done = true;
                    if (LISTS >= 0) {                                 // (one lane: bucket-full keys are a handful per operator at most)
                        AllocRec r; r.key = s.key[k]; r.ptr = ptr; r.pad = 0;
                        const uint32_t keep = keepRec<(LISTS < 0 ? 0 : LISTS)>(f, fo, r);
                        if (keep) {
                            const uint32_t pos = atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), 1u);
                            if (LISTS == 2) atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, (keep & 1u) + ((keep >> 1) & 1u));
                            listWrite(d, pos, r.key, ptr, LISTS == 2 ? keep : 0u, allocBase + gi);
                        }
                    }
                    break;
                }
                maxIter++;
            }
            if (!done) {                                              // window exhausted: give the block back
                d.allocList[allocBase + gi].ptr = BF_FREE_ENTRY;
                newCounter++;
                d.heap[newCounter] = (uint32_t)ptr / (uint32_t)VOX;  // appendHeap :542-546
                dropped++;
            }
        }
        d.heapCounter[0] = newCounter;
        d.allocCount[0] = allocBase + Mp;
        d.allocSnap[0] = allocBase + Mp;
        if (dropped) atomicAdd(&d.stats[ST_DROPPED], dropped);
        d.overflowCount[0] = 0;
    }
    __syncthreads();
    {   // keys that found their bin full: release their de-dup slots now that nobody probes any more.  Only the first OVCAP of them
        // were recorded; beyond that (ERR_BIN_OVERFLOW is set anyway) the whole set is cleared - at this point every slot of the set
        // belongs to a key that has been placed or dropped, so "empty everywhere" is its correct state.
        const uint32_t stuck = min(stuckAll, OVCAP);
        for (uint32_t q = threadIdx.x; q < stuck; q += blockDim.x) d.dedupe[d.stuckSlots[q]] = EMPTY64;
        if (stuckAll > OVCAP)
            for (uint32_t i = threadIdx.x; i <= d.dedupeMask; i += blockDim.x) d.dedupe[i] = EMPTY64;
        if (stuckAll) { __syncthreads(); if (threadIdx.x == 0) d.stats[ST_STUCK] = 0; }      // block-uniform
    }
    for (uint32_t b = threadIdx.x; b < NBINS; b += blockDim.x) d.binCount[b] = 0;
}

// The hand-off to the last workgroup below relies on gfx94x / gfx950 memory-model details: vmcnt counts stores (s_waitcnt vmcnt(0) drains
// them), agent-scope relaxed atomic stores are write-through (sc1) and a single-lane agent acquire invalidates the CU's L1 for the whole
// workgroup.  There is no portable fallback; the library is built for gfx950 only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_alloc_place: the workgroup hand-off is written for gfx942 / gfx950"
#endif
// LISTS >= 0 (round 4): the same launch also builds the operator's block list - frustum list (0) or union list of a fused re-integration (2).  The list is
// {entries of the allocated-block list as it stood before this placement that pass the frustum test} + {the blocks placed now}: every workgroup, once its bin is
// placed, filters its share of the old entries (nobody writes them during this launch) and appends them; every placed block is appended by the lane that placed it
// (bucket-full keys by the tail).  The list's ORDER is not part of any result (the voxel update treats every block on its own, garbage collection bins and sorts
// what it takes from the list; the reference's own compactify appends with atomicAdd, CUDASceneRepHashSDF.cu:324-366).  The preparation chain of an operator -
// which paces the re-integration loop (profiles/r04_streams_and_queues.md section 3) - is then two launches: march, place + list.  The list counters are zeroed
// by the march (or by the host in front of an exchanged march).
template <int LISTS>
__global__ __launch_bounds__(256) void k_alloc_place(Dev d, Frame f, Frame fo) {
    __shared__ SortLds s;
    __shared__ uint32_t scratch[16];
    __shared__ int8_t sel[BINCAP];
    __shared__ uint32_t lastFlag;
    __shared__ uint32_t bigBin;
    __shared__ uint32_t wscan[4];
    __shared__ uint32_t sbase;
    __shared__ uint32_t tailIn[4];      // M, heapC, allocBase, stuck keys: the launch's inputs the tail needs (read in the first round trip)
    __builtin_amdgcn_s_setprio(3);
    static_assert(PLACE_WGS == NBINS && NBINS == 256, "one workgroup per bin");
    // The kernel is a chain of dependent memory round trips, each 3-6 us next to the voxel update of the previous operator (profiles/r04_pipeline_timeline_window.txt:
    // 50-65 us per launch in the saturated window); everything that does not depend on another workgroup is therefore requested in the FIRST one - the bin, the
    // counters, and the workgroup's tile of the allocated-block list for the list pass below.
    constexpr int LM = LISTS < 0 ? 0 : LISTS;
    AllocRec rec0; rec0.key = 0; rec0.ptr = BF_FREE_ENTRY; rec0.pad = 0;
    uint32_t nOld = 0;
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
    if (LISTS >= 0) { nOld = d.allocCount[0]; if (i0 < f.numSDFBlocks) rec0 = d.allocList[i0]; }
    // One workgroup per bin.  The usual bin holds a handful of new keys and is placed by the workgroup's first wave alone (registers only);
    // a bin with more than 64 keys - first frames of a scan, fast motion, every operator of a 2 mm sweep - is sorted by the whole workgroup
    // in LDS.  (Round 2 gave each of 64 workgroups four bins and walked its big bins one after the other: 409 us per operator on the
    // 1280x960 @2 mm sweep, as long as the voxel update the allocation is supposed to hide behind.)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t bin = blockIdx.x;
    if (wave == 0) {   // first round trip: this bin's count and first 64 records, the counts of all bins, the two counters
        const uint32_t nRaw = d.binCount[bin];
        const uint4 c4 = reinterpret_cast<const uint4*>(d.binCount)[lane];
        const BinRec r = d.bins[(size_t)bin * BINCAP + lane];
        const uint32_t heapC = d.heapCounter[0], allocBase = d.allocCount[0], stuckAll = d.stats[ST_STUCK];
        const uint32_t n = min(nRaw, BINCAP);
        const uint32_t M = (uint32_t)wave_sum_i((int)(min(c4.x, BINCAP) + min(c4.y, BINCAP) + min(c4.z, BINCAP) + min(c4.w, BINCAP)));
        uint32_t pre = 0;
        if (lane * 4 + 0 < bin) pre += min(c4.x, BINCAP);
        if (lane * 4 + 1 < bin) pre += min(c4.y, BINCAP);
        if (lane * 4 + 2 < bin) pre += min(c4.z, BINCAP);
        if (lane * 4 + 3 < bin) pre += min(c4.w, BINCAP);
        const uint32_t base = (uint32_t)wave_sum_i((int)pre);
        if (lane == 0) { bigBin = n > 64 ? 1u : 0u; tailIn[0] = M; tailIn[1] = heapC; tailIn[2] = allocBase; tailIn[3] = stuckAll; }
        if (n > 0 && n <= 64) placeBinWave<LISTS>(d, f, fo, n, r, base, heapC, allocBase, lane);
    }
    __syncthreads();
    if (bigBin) placeBin<LISTS>(d, f, fo, s, scratch, sel, bin);          // block-uniform
    // The list pass over the allocated-block list as it stood before this placement (the tail writes allocCount after every workgroup has been here).  One entry per
    // thread - the record requested above - while that gives every workgroup at most one tile (the ~60 000 allocated blocks of the bench window are 230 tiles over
    // the 256 workgroups), four per thread beyond (262 000 blocks of the 5000-frame stream are one tile per workgroup again).
    if (LISTS >= 0) {
        if (nOld <= gridDim.x * 256u) listAppendBlock<LM>(d, i0 < nOld ? keepRec<LM>(f, fo, rec0) : 0u, rec0.key, rec0.ptr, i0, wscan, &sbase);
        else listAppendTiles<LM, 4>(d, f, fo, nOld, blockIdx.x, gridDim.x, wscan, &sbase);
    }
    // hand-off to the workgroup that arrives last: every wave drains its (write-through) stores, then one lane takes a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&d.stats[ST_TICKET], 1u);
        lastFlag = ticket == gridDim.x - 1 ? 1u : 0u;
        if (lastFlag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // same CU as every other wave of this workgroup: its L1 is clean from here on
    }
    __syncthreads();
    if (!lastFlag) return;
    placeTail<LISTS>(d, f, fo, s, tailIn[0], tailIn[1], tailIn[2], tailIn[3]);
    if (threadIdx.x == 0) d.stats[ST_TICKET] = 0;
}

// ---------------------------------------------------------------------------------------
// list maintenance after a garbage collection: ordered stream compaction of the allocated-block list (holes dropped, order kept; two passes, deterministic)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_count(Dev d) {
    __shared__ uint32_t wsum[4];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n = d.allocCount[0];
    const uint32_t numTiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = tile * TILE + threadIdx.x * 4 + k;
            if (i < n) c += d.allocList[i].ptr != BF_FREE_ENTRY ? 1u : 0u;
        }
        c = (uint32_t)wave_sum_i((int)c);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) d.tileCounts[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_compact_scatter(Dev d) {
    __shared__ uint32_t wsum[4];
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t wscan[4];
    const uint32_t n = d.allocCount[0];
    const uint32_t numTiles = (n + TILE - 1) / TILE;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
        // exclusive prefix of the preceding tiles
        uint32_t v = 0;
        for (uint32_t t = threadIdx.x; t < tile; t += blockDim.x) v += d.tileCounts[t];
        v = (uint32_t)wave_sum_i((int)v);
        if (lane == 0) wsum[wave] = v;
        __syncthreads();
        const uint32_t base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
        AllocRec recs[4];
        bool keep[4];
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = tile * TILE + threadIdx.x * 4 + k;
            keep[k] = false;
            if (i < n) { recs[k] = d.allocList[i]; keep[k] = recs[k].ptr != BF_FREE_ENTRY; }
            c += keep[k] ? 1u : 0u;
        }
        // exclusive scan of c across the block: wave scan + wave totals
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
        if (lane == 63) wscan[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wscan[w];
        const uint32_t tileTotal = wscan[0] + wscan[1] + wscan[2] + wscan[3];
        uint32_t pos = base + woff + incl - c;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (keep[k]) d.allocListAlt[pos++] = recs[k];
        if (tile == numTiles - 1 && threadIdx.x == 0) d.tileCounts[numTiles] = base + tileTotal;      // new list length, committed by k_list_commit
        __syncthreads();
    }
}

// The frustum (MODE 0) or union (MODE 2) list of an operator that does not allocate (a de-integration, an operator behind an external allocation, the frustum
// list after a fused re-integration): one pass over the allocated-block list up to the operator's snapshot.  compactCount[0] / [1] are zeroed by k_alloc_snapshot.
template <int MODE>
__global__ __launch_bounds__(256) void k_compact_append(Dev d, Frame f, Frame fo) {
    __shared__ uint32_t wscan[4];
    __shared__ uint32_t sbase;
    __builtin_amdgcn_s_setprio(3);
    listAppendTiles<MODE, 4>(d, f, fo, d.allocSnap[0], blockIdx.x, gridDim.x, wscan, &sbase);
}

__global__ void k_alloc_snapshot(Dev d) { d.allocSnap[0] = d.allocCount[0]; d.compactCount[0] = 0; d.compactCount[1] = 0; }      // an operator without an allocation of its own

__global__ void k_list_commit(Dev d) {
    const uint32_t n = d.allocCount[0];
    const uint32_t numTiles = (n + TILE - 1) / TILE;
    if (numTiles > 0) d.allocCount[0] = d.tileCounts[numTiles];
}

// ---------------------------------------------------------------------------------------
// voxel update: integrate / de-integrate (CUDASceneRepHashSDF.cu:420-521)
// ---------------------------------------------------------------------------------------
// combineVoxel / its inverse on (sdf, weight, packed colour); DEINT resets a voxel whose weight drops to zero.
//
// Integrate: 0.2 c + 0.8 o with bytes c, o is (c + 4 o) / 5, never within 0.1 of a rounding tie, so roundf equals
// round-to-nearest-even (one v_rndne instead of the five-instruction roundf; all 65536 (c, o) pairs are checked on the CPU in
// tests/test_host_cpu.py).  De-integrate keeps the reference expressions: (o w - c) / (w - 1) has ties and its distance to a tie
// shrinks with the weight (1 / (2 (w - 1))), so no shortcut is safe for long sequences.
template <bool DEINT, class FR>
BF_DEV void voxelApply(const FR& f, float sdf, uchar4 cc, float& vSdf, float& vW, uint32_t& vC) {
    const float c0 = (float)cc.x, c1 = (float)cc.y, c2 = (float)cc.z;
    const float oSdf = vSdf, oW = vW;
    const uint32_t oC = vC;
    const float o0 = (float)(oC & 0xFF), o1 = (float)((oC >> 8) & 0xFF), o2 = (float)((oC >> 16) & 0xFF);
    float nSdf, nW;
    uint32_t nC;
    if (!DEINT) {
        float r0, r1, r2;
        if (oW == 0.0f) { r0 = c0; r1 = c1; r2 = c2; }
        else { r0 = 0.2f * c0 + 0.8f * o0; r1 = 0.2f * c1 + 0.8f * o1; r2 = 0.2f * c2 + 0.8f * o2; }
        r0 = fmaxf(0.0f, fminf(__builtin_rintf(r0), 254.5f));
        r1 = fmaxf(0.0f, fminf(__builtin_rintf(r1), 254.5f));
        r2 = fmaxf(0.0f, fminf(__builtin_rintf(r2), 254.5f));
        nC = (uint32_t)(int)r0 | ((uint32_t)(int)r1 << 8) | ((uint32_t)(int)r2 << 16) | 0xFF000000u;      // r in [0, 254.5] after the clamps, never NaN
        nSdf = (sdf * 1.0f + oSdf * oW) / (1.0f + oW);
        nW = fminf(f.weightMax, 1.0f + oW);
    } else {
        float r0 = (o0 * oW - c0 * 1.0f) / (oW - 1.0f);
        float r1 = (o1 * oW - c1 * 1.0f) / (oW - 1.0f);
        float r2 = (o2 * oW - c2 * 1.0f) / (oW - 1.0f);
        r0 = fmaxf(0.0f, fminf(roundf(r0), 254.5f));
        r1 = fmaxf(0.0f, fminf(roundf(r1), 254.5f));
        r2 = fmaxf(0.0f, fminf(roundf(r2), 254.5f));
        nC = (uint32_t)(int)r0 | ((uint32_t)(int)r1 << 8) | ((uint32_t)(int)r2 << 16) | 0xFF000000u;
        nSdf = (oSdf * oW - sdf * 1.0f) / (oW - 1.0f);
        nW = fmaxf(0.0f, oW - 1.0f);
        if (nW <= 0.001f) { nSdf = 0.0f; nC = 0u; nW = 0.0f; }
    }
    vSdf = nSdf; vW = nW; vC = nC;
}

// ---------------------------------------------------------------------------------------
// voxel update, column form: ONE WAVE per SDF block.  Lane (x, y) owns the z-column of 8 voxels and walks it in 4 steps of 2
// voxels held in the two halves of packed-f32 registers (v_pk_mul/add/fma_f32: two IEEE operations per issue slot).
//
// Why: rocprofv3 SQ counters showed the one-voxel-per-lane kernels of round 1 (512-thread workgroups, removed in round 4) VALU-issue bound
// (profiles/r02_sq_tsdf_update.md: 234 vector instructions per 64 voxels, SQ_ACTIVE_INST_VALU ~ 86 % of the SIMD cycles), not HBM bound.
// Per voxel the arithmetic is the IEEE operation sequence of voxelSampleU / voxelApply, bit for bit, but
//   * the terms of the camera transform that depend on (x, y) only are computed once per column, the z term once per wave;
//   * the two divisions of the projection share one refined reciprocal, and so do the four of a de-integration: the quotient is
//     formed by the instruction sequence the compiler emits for an IEEE f32 division (rcp, 2 FMA Newton step, q = n*r, two
//     residual corrections) minus its v_div_scale / v_div_fmas / v_div_fixup range handling — identical bits whenever that range
//     handling is the identity.  That is guaranteed per block by a wave-uniform bound on the camera-space coordinates (projection)
//     and checked per value where the result is stored (TSDF quotients); anything else takes the literal `/`;
//   * independent operations of the two voxels of a pair are packed.
// A wave whose block fails the bound runs voxelSampleU / voxelApply for its 512 voxels (colExact).
// ---------------------------------------------------------------------------------------
// f2i as the one instruction it describes (v_cvt_i32_f32: toward zero, saturating, NaN -> 0).  The portable spelling in bf_device.h
// costs three compares and three exec-mask branches per conversion in the voxel kernels' inner loop.
// Spelled as a cast (one v_cvt_i32_f32_e32; HIP's __float2int_rz puts a redundant v_trunc_f32 in front of it), never as inline assembly: the compiler pads the
// gfx950 VALU hazards only around instructions it can see.  (Round 5 blamed an asm spelling of this conversion for the batched update's run-to-run differences;
// round 6 measured them in the running loop and found the packed-FP32 code of the projection responsible: profiles/r06_determinism.md.)
BF_DEV int f2iHw(float v) { return (int)v; }
// min / max of values that are never NaN as ONE v_med3_f32: fminf / fmaxf cost two instructions each under the IEEE mode (a v_max_f32 x, x, x to quiet a signalling
// NaN in front of the v_min / v_max) - 72 of the batched update's ~990 vector instructions per operator slot
// (finite outer bounds: with an infinity the compiler folds the median back into min / max)
BF_DEV float minNoNan(float v, float hi) { return __builtin_amdgcn_fmed3f(v, -0x1p126f, hi); }
BF_DEV float maxNoNan(float v, float lo) { return __builtin_amdgcn_fmed3f(v, lo, 0x1p126f); }

typedef float v2f __attribute__((ext_vector_type(2)));
BF_DEV v2f sp2(float a) { v2f r; r.x = a; r.y = a; return r; }
BF_DEV v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct UpdCam {            // the pose-independent part of Frame the voxel update reads (kernarg => SGPRs)
    float fx, fy, mx, my;
    float voxelSize, maxDist, truncScale, truncation, weightMax;
    uint32_t W, H;
};
struct UpdPose {           // rows 0..2 of the world -> camera transform, plus host-computed bounds for the fast path
    float R[9], t[3];
    float rmax;            // max |R_ij|
    uint32_t fastOk;       // pose and camera finite and moderate (see makeUpdPose)
};

// reciprocal refined by one Newton step: the first three operations of the f32 division expansion
BF_DEV v2f rcpRefined(v2f d) {
    v2f r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    const v2f e = pkfma(-d, r, sp2(1.0f));
    return pkfma(e, r, r);
}
// n / d given r = rcpRefined(d): q = n*r and two residual corrections (the rest of the expansion).  Equals the IEEE quotient when
// the hardware's operand scaling would be the identity (d, n and n/d normal and far from the ends of the exponent range).
BF_DEV v2f divRefined(v2f n, v2f d, v2f r) {
    v2f q = n * r;
    v2f e = pkfma(-d, q, n);
    q = pkfma(e, r, q);
    e = pkfma(-d, q, n);
    return pkfma(e, r, q);
}
// the same for a quotient whose bits are stored: values outside the proven range (zero included: the sign of a zero quotient) take `/`
BF_DEV float divStored(float n, float d, float q) {
    if (!(fabsf(n) >= 0x1p-60f && fabsf(n) <= 0x1p60f)) q = n / d;
    return q;
}

struct ColPose { float t0, t1, t2; };      // (R_r0 * x + R_r1 * y) of this lane's column, r = 0..2

BF_DEV ColPose colPose(const UpdPose& f, float xw, float yw) {
    ColPose p;
    p.t0 = f.R[0] * xw + f.R[1] * yw;
    p.t1 = f.R[3] * xw + f.R[4] * yw;
    p.t2 = f.R[6] * xw + f.R[7] * yw;
    return p;
}

// wave-uniform: may every voxel of block e be projected with divRefined?  Camera-space z in [0.01, 4096], |x|, |y| <= 4096, all the
// terms of the transform bounded by 4096 (so that rounding cannot eat the margins), intrinsics moderate (fastOk).
BF_DEV bool blockFast(const UpdCam& c, const UpdPose& f, int4 e) {
    const float s = 8.0f * c.voxelSize;
    const float bx = (float)e.x * s, by = (float)e.y * s, bz = (float)e.z * s;
    const float bmax = fmaxf(fmaxf(fabsf(bx), fabsf(by)), fabsf(bz)) + s;
    const float x0 = f.R[0] * bx + f.R[1] * by + f.R[2] * bz + f.t[0];
    const float y0 = f.R[3] * bx + f.R[4] * by + f.R[5] * bz + f.t[1];
    const float z0 = f.R[6] * bx + f.R[7] * by + f.R[8] * bz + f.t[2];
    const float spr = 3.0f * f.rmax * s;       // |pc(voxel) - pc(corner)| <= (|R_r0| + |R_r1| + |R_r2|) * 7 voxels
    return f.fastOk && bmax * f.rmax <= 4096.0f && z0 - spr >= 0.01f && z0 + spr <= 4096.0f && fabsf(x0) + spr <= 4096.0f && fabsf(y0) + spr <= 4096.0f;
}

struct PairSample { v2f sdf; uint32_t cA, cB; bool okA, okB; };      // truncated signed distances, colour pixels, "voxel is updated"
struct PairAddr { v2f pcz; uint32_t pixA, pixB; bool inA, inB; };       // camera-space z, pixel index, "projects into the image"

// voxelSample for the two voxels (z, z + 1) of this lane's column, first half: projection.  pz = their world z coordinates
BF_DEV PairAddr projectPair(const UpdCam& c, const UpdPose& f, const ColPose& p, v2f pz) {
    const v2f pcx = (sp2(p.t0) + sp2(f.R[2]) * pz) + sp2(f.t[0]);
    const v2f pcy = (sp2(p.t1) + sp2(f.R[5]) * pz) + sp2(f.t[1]);
    PairAddr o;
    o.pcz = (sp2(p.t2) + sp2(f.R[8]) * pz) + sp2(f.t[2]);
    const v2f r = rcpRefined(o.pcz);
    const v2f sx = divRefined(pcx * sp2(c.fx), o.pcz, r) + sp2(c.mx);
    const v2f sy = divRefined(pcy * sp2(c.fy), o.pcz, r) + sp2(c.my);
    const v2f hx = sx + sp2(0.5f), hy = sy + sp2(0.5f);
    const uint32_t pxA = (uint32_t)f2iHw(hx.x), pyA = (uint32_t)f2iHw(hy.x), pxB = (uint32_t)f2iHw(hx.y), pyB = (uint32_t)f2iHw(hy.y);
    o.inA = pxA < c.W && pyA < c.H; o.inB = pxB < c.W && pyB < c.H;
    o.pixA = pyA * c.W + pxA; o.pixB = pyB * c.W + pxB;
    return o;
}
// second half, after the depth (and, fetched alongside it, the colour) of the two pixels arrived
BF_DEV PairSample finishPair(const UpdCam& c, const PairAddr& a, v2f dep, uint32_t cA, uint32_t cB) {
    PairSample o;
    o.cA = cA; o.cB = cB;
    o.sdf = dep - a.pcz;
    const v2f trunc = sp2(c.truncation) + sp2(c.truncScale) * dep;
    o.okA = a.inA && dep.x != BF_MINF && dep.x < c.maxDist && fabsf(o.sdf.x) < trunc.x;
    o.okB = a.inB && dep.y != BF_MINF && dep.y < c.maxDist && fabsf(o.sdf.y) < trunc.y;
    return o;          // |sdf| < trunc already: the reference's clamp to [-trunc, trunc] is the identity
}

// fmaxf(0, fminf(x, 254.5)) as one v_med3_f32.  Equal for every non-NaN x; a NaN (de-integration of a voxel of weight <= 1, whose result
// is reset afterwards) gives 254.5 there and an unspecified member of {0, 254.5} here - both in [0, 254.5], both thrown away.
BF_DEV float clampByte(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 254.5f); }
BF_DEV float byteF(uint32_t c, int k) { return (float)((c >> (8 * k)) & 0xFFu); }
BF_DEV uint32_t packRGB(float r0, float r1, float r2) {       // r in [0, 254.5] after the clamps, never NaN
    return (uint32_t)(int)r0 | ((uint32_t)(int)r1 << 8) | ((uint32_t)(int)r2 << 16) | 0xFF000000u;
}

// voxelApply<true> on the pair (lanes / halves where on* is false keep their voxel)
BF_DEV void deintPair(v2f sdf, uint32_t cA, uint32_t cB, bool onA, bool onB, v2f& vS, v2f& vW, uint32_t& vCA, uint32_t& vCB) {
    const v2f d = vW - sp2(1.0f);
    const v2f r = rcpRefined(d);
    float ra[3], rb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v2f o, c; o.x = byteF(vCA, k); o.y = byteF(vCB, k); c.x = byteF(cA, k); c.y = byteF(cB, k);
        const v2f q = divRefined(o * vW - c, d, r);           // a zero or out-of-range quotient here dies in the clamp / the reset below
        ra[k] = clampByte(roundf(q.x));
        rb[k] = clampByte(roundf(q.y));
    }
    const v2f n = vS * vW - sdf;
    const v2f q = divRefined(n, d, r);
    float sA = divStored(n.x, d.x, q.x), sB = divStored(n.y, d.y, q.y);
    float wA = fmaxf(0.0f, d.x), wB = fmaxf(0.0f, d.y);
    uint32_t nA = packRGB(ra[0], ra[1], ra[2]), nB = packRGB(rb[0], rb[1], rb[2]);
    if (wA <= 0.001f) { sA = 0.0f; nA = 0u; wA = 0.0f; }      // weights <= 1 (d <= 0: rcp gave inf / NaN) end here
    if (wB <= 0.001f) { sB = 0.0f; nB = 0u; wB = 0.0f; }
    if (onA) { vS.x = sA; vW.x = wA; vCA = nA; }
    if (onB) { vS.y = sB; vW.y = wB; vCB = nB; }
}

// voxelApply<false> on the pair
BF_DEV void intPair(float weightMax, v2f sdf, uint32_t cA, uint32_t cB, bool onA, bool onB, v2f& vS, v2f& vW, uint32_t& vCA, uint32_t& vCB) {
    const v2f d = sp2(1.0f) + vW;                               // >= 1
    const v2f r = rcpRefined(d);
    float ra[3], rb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v2f o, c; o.x = byteF(vCA, k); o.y = byteF(vCB, k); c.x = byteF(cA, k); c.y = byteF(cB, k);
        const v2f m = sp2(0.2f) * c + sp2(0.8f) * o;
        const float a = vW.x == 0.0f ? c.x : m.x, b = vW.y == 0.0f ? c.y : m.y;
        ra[k] = clampByte(__builtin_rintf(a));
        rb[k] = clampByte(__builtin_rintf(b));
    }
    const v2f n = sdf + vS * vW;
    const v2f q = divRefined(n, d, r);
    const float sA = divStored(n.x, d.x, q.x), sB = divStored(n.y, d.y, q.y);
    if (onA) { vS.x = sA; vW.x = fminf(weightMax, d.x); vCA = packRGB(ra[0], ra[1], ra[2]); }
    if (onB) { vS.y = sB; vW.y = fminf(weightMax, d.y); vCB = packRGB(rb[0], rb[1], rb[2]); }
}

// voxelSample on (UpdCam, UpdPose): the literal operation sequence for one voxel — the path of blocks outside the proven range
BF_DEV bool voxelSampleU(const UpdCam& c, const UpdPose& f, int4 e, int lx, int ly, int lz, const float* __restrict__ depth, float& sdf, uint32_t& pix) {
    const float x = (float)(e.x * BS + lx) * c.voxelSize, y = (float)(e.y * BS + ly) * c.voxelSize, z = (float)(e.z * BS + lz) * c.voxelSize;
    const float cx = f.R[0] * x + f.R[1] * y + f.R[2] * z + f.t[0] * 1.0f;
    const float cy = f.R[3] * x + f.R[4] * y + f.R[5] * z + f.t[1] * 1.0f;
    const float cz = f.R[6] * x + f.R[7] * y + f.R[8] * z + f.t[2] * 1.0f;
    const float sx = cx * c.fx / cz + c.mx;
    const float sy = cy * c.fy / cz + c.my;
    const uint32_t px = (uint32_t)f2i(sx + 0.5f), py = (uint32_t)f2i(sy + 0.5f);
    if (!(px < c.W && py < c.H)) return false;
    pix = py * c.W + px;
    const float dep = depth[pix];
    if (dep == BF_MINF) return false;
    if (!(dep < c.maxDist)) return false;
    sdf = dep - cz;
    const float trunc = c.truncation + c.truncScale * dep;
    if (!(fabsf(sdf) < trunc)) return false;
    if (sdf >= 0.0f) sdf = fminf(trunc, sdf); else sdf = fmaxf(-trunc, sdf);
    return true;
}

// the literal per-voxel path for one block (512 voxels, 8 per lane)
template <bool DE, bool IN>
BF_DEV void colExact(const Dev& d, const UpdCam& c, const UpdPose& fIn, const UpdPose& fDe, int4 e, uint32_t flags, uint32_t lane, const float* __restrict__ depth,
                     const uchar4* __restrict__ color) {
    const int lx = (int)(lane & 7), ly = (int)(lane >> 3);
#pragma unroll 1
    for (int lz = 0; lz < 8; ++lz) {
        float sdfDe = 0.0f, sdfIn = 0.0f; uint32_t pixDe = 0, pixIn = 0;
        const bool doDe = DE && (flags & 2u) && voxelSampleU(c, fDe, e, lx, ly, lz, depth, sdfDe, pixDe);
        const bool doIn = IN && (flags & 1u) && voxelSampleU(c, fIn, e, lx, ly, lz, depth, sdfIn, pixIn);
        if (!doDe && !doIn) continue;
        uint32_t* vp = reinterpret_cast<uint32_t*>(d.vox + ((size_t)(uint32_t)e.w + (uint32_t)lz * 64u + lane));
        float vSdf = __uint_as_float(vp[0]), vW = __uint_as_float(vp[1]);
        uint32_t vC = vp[2];
        if (doDe) voxelApply<true>(c, sdfDe, color[pixDe], vSdf, vW, vC);
        if (doIn) voxelApply<false>(c, sdfIn, color[pixIn], vSdf, vW, vC);
        vp[0] = __float_as_uint(vSdf); vp[1] = __float_as_uint(vW); vp[2] = vC;
    }
}

template <bool DE, bool IN>
BF_DEV void colFast(const Dev& d, const UpdCam& c, const UpdPose& fIn, const UpdPose& fDe, int4 e, uint32_t flags, uint32_t lane, const float* __restrict__ depth,
                    const uint32_t* __restrict__ color) {
    const float xw = (float)(e.x * BS + (int)(lane & 7)) * c.voxelSize, yw = (float)(e.y * BS + (int)(lane >> 3)) * c.voxelSize;
    const bool useDe = DE && (flags & 2u), useIn = IN && (flags & 1u);       // wave-uniform
    ColPose pDe = {0.0f, 0.0f, 0.0f}, pIn = {0.0f, 0.0f, 0.0f};
    if (useDe) pDe = colPose(fDe, xw, yw);
    if (useIn) pIn = colPose(fIn, xw, yw);
    uint32_t* base = reinterpret_cast<uint32_t*>(d.vox + ((size_t)(uint32_t)e.w + lane));
#pragma unroll 1
    for (int z = 0; z < 8; z += 2) {
        // every load of the pair is issued before the first use: the two voxels (speculatively: ~1 in 4 is not touched), then depth AND
        // colour of the projected pixels of both poses — one memory round trip per pair instead of five
        uint32_t* vpA = base + (size_t)z * 64u * 3u; uint32_t* vpB = vpA + 64u * 3u;
        v2f vS, vW; uint32_t vCA, vCB;
        vS.x = __uint_as_float(vpA[0]); vW.x = __uint_as_float(vpA[1]); vCA = vpA[2];
        vS.y = __uint_as_float(vpB[0]); vW.y = __uint_as_float(vpB[1]); vCB = vpB[2];
        v2f pz; pz.x = (float)(e.z * BS + z) * c.voxelSize; pz.y = (float)(e.z * BS + z + 1) * c.voxelSize;
        PairAddr aDe, aIn;
        aDe.inA = aDe.inB = aIn.inA = aIn.inB = false; aDe.pixA = aDe.pixB = aIn.pixA = aIn.pixB = 0u; aDe.pcz = aIn.pcz = sp2(0.0f);
        if (useDe) aDe = projectPair(c, fDe, pDe, pz);
        if (useIn) aIn = projectPair(c, fIn, pIn, pz);
        v2f dDe = sp2(BF_MINF), dIn = sp2(BF_MINF); uint32_t cDeA = 0u, cDeB = 0u, cInA = 0u, cInB = 0u;
        if (aDe.inA) { dDe.x = depth[aDe.pixA]; cDeA = color[aDe.pixA]; }
        if (aDe.inB) { dDe.y = depth[aDe.pixB]; cDeB = color[aDe.pixB]; }
        if (aIn.inA) { dIn.x = depth[aIn.pixA]; cInA = color[aIn.pixA]; }
        if (aIn.inB) { dIn.y = depth[aIn.pixB]; cInB = color[aIn.pixB]; }
        const PairSample sDe = finishPair(c, aDe, dDe, cDeA, cDeB), sIn = finishPair(c, aIn, dIn, cInA, cInB);
        const bool anyA = sDe.okA || sIn.okA, anyB = sDe.okB || sIn.okB;
        if (!anyA && !anyB) continue;
        if (DE && (sDe.okA || sDe.okB)) deintPair(sDe.sdf, sDe.cA, sDe.cB, sDe.okA, sDe.okB, vS, vW, vCA, vCB);
        if (IN && (sIn.okA || sIn.okB)) intPair(c.weightMax, sIn.sdf, sIn.cA, sIn.cB, sIn.okA, sIn.okB, vS, vW, vCA, vCB);
        if (anyA) { vpA[0] = __float_as_uint(vS.x); vpA[1] = __float_as_uint(vW.x); vpA[2] = vCA; }
        if (anyB) { vpB[0] = __float_as_uint(vS.y); vpB[1] = __float_as_uint(vW.y); vpB[2] = vCB; }
    }
}

// MODE 0 integrate (pose `in`), 1 de-integrate (pose `de`), 2 fused: de-integrate at `de`, then integrate at `in`, over the union
// list with membership flags
template <int MODE>
__global__ __launch_bounds__(256) void k_update_col(Dev d, UpdCam c, UpdPose in, UpdPose de, const float* __restrict__ depth, const uchar4* __restrict__ color,
                                                    int accumulate, int forceExact) {
    if (color == nullptr) return;   // .cu:441-448: without colour data `color.x != MINF` never holds
    constexpr bool DE = MODE != 0, IN = MODE != 1;
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nWaves = gridDim.x * 4u;
    if (accumulate && wave == 0 && lane == 0) {          // block accounting of the timed launches (bench.py); no atomics in the update itself
        if (MODE == 2) { d.occSum[2] += (unsigned long long)n; d.occSum[0] += (unsigned long long)(uint32_t)d.compactCount[1]; }
        else { d.occSum[0] += (unsigned long long)n; d.occSum[1] += (unsigned long long)n; }
    }
    // Grid stride over a list of ~25-36 k blocks with 8192 workgroups of four waves (about one block per wave): workgroups retire at
    // different times and the dispatcher refills the CUs, which balances blocks of unequal cost.  Measured alternatives on the same
    // list: 4096 workgroups +3.5 %, 16384 +12 %, a grid of exactly the resident waves (6 per CU, ~6 blocks per wave) +21 %, blocks
    // handed out through one atomic ticket counter 3x slower (~36 k device-scope atomics on one address serialise at the memory side).
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const int4 e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];                                    // wave-uniform
        const uint32_t flags = MODE == 2 ? reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4] : 3u;
        bool fast = !forceExact;
        if (DE && (flags & 2u)) fast = fast && blockFast(c, de, e);
        if (IN && (flags & 1u)) fast = fast && blockFast(c, in, e);
        if (fast) colFast<DE, IN>(d, c, in, de, e, flags, lane, depth, reinterpret_cast<const uint32_t*>(color));
        else colExact<DE, IN>(d, c, in, de, e, flags, lane, depth, color);
    }
}

// ---------------------------------------------------------------------------------------
// voxel update, arithmetic contract "fast" (bf_scene_set_arith(1) / BF_TSDF_ARITH=fast).
//
// The reference's Release build compiles its CUDA code with -use_fast_math (FriedLiver.vcxproj:124 <FastMath>true</FastMath>): `a / b`
// is the approximate __fdividef, products and sums are contracted to FMAs, denormals flush.  The column kernel above instead evaluates
// every operation of CUDASceneRepHashSDF.cu:425-516 as written, IEEE op by op (that is what makes it bit-comparable with a host build
// of the reference, and VALU-issue bound: profiles/r02_sq_tsdf_update.md, 1696 vector instructions per block of a fused launch).
// This kernel is the same update under the contract the reference's GPU build actually has:
//   * a quotient is numerator * v_rcp_f32(denominator) (1 ulp reciprocal), no refinement, no range handling;
//   * the image coordinates are one FMA chain over the voxel's integer coordinates with the intrinsics folded into the rows of the
//     world -> camera transform on the host (in double, rounded once);
//   * camera-space z keeps the exact kernel's operation order, so the signed distance SAMPLE of a voxel (depth - z) has the same bits
//     in both contracts whenever both pick the same pixel;
//   * colour bytes: v_cvt_f32_ubyteN in, one FMA per channel, v_cvt_pk_u8_f32 out.
// What can differ from the exact contract, and how the tests bound it (tests/test_tsdf_gpu.py, arith = fast): the set of allocated
// blocks, bucket occupancy, heap and every weight are decided by integer / comparison logic on identical samples and stay EXACT;
// sdf within 1e-5 x truncation and colour within 1 LSB; a voxel whose projection lies within ~1e-4 pixel of a pixel
// boundary may sample the neighbouring pixel (as it may between the reference's own fast-math build and a host build of it) - the
// tests recompute the projection in double, exclude exactly those voxels, and bound their share.
// Depth and colour are fetched through buffer descriptors (32-bit offsets, out-of-image lanes read 0 and are masked) - no 64-bit
// address arithmetic, no divergent branches around the loads.
// ---------------------------------------------------------------------------------------
// (64-byte aligned like every kernel argument the fast update reads with wide scalar loads: none of them straddles a 64-byte line - tsdf_batch.h BatchUpdOpApx)
struct alignas(64) ApxCam {
    float mxh, myh;            // principal point + 0.5 (the rounding offset of the pixel index)
    float voxelSize, maxDist, truncScale, truncation, weightMax;
    uint32_t W, H, bytes;      // image size, bytes of one image plane (W * H * 4)
};
struct alignas(64) ApxPose {
    float ax, bx, cx, dx;      // fx * voxelSize * (R00, R01, R02), fx * t0: numerator of the image x coordinate over the integer voxel coordinates
    float ay, by, cy, dy;      // fy * voxelSize * (R10, R11, R12), fy * t1
    float r6, r7, r8, t2;      // third row of the world -> camera transform (camera-space z in the exact kernel's operation order)
};

// v_cvt_pk_u8_f32 saturates to [0, 255]; whether it rounds to nearest-even or truncates is probed once per scene (k_probe_cvt) and
// selects the kernel instantiation: RNE converts the value itself, RTZ the value + 0.5 (folded into the FMA that produces it).
// roundf-then-truncate of the exact contract == nearest, ties away from zero: differs from RNE on exact .5 ties only (1 LSB, inside
// the contract).  Both contracts clamp at 254 ((uint)254.5).
template <bool RNE>
BF_DEV uint32_t packByte(float v, uint32_t sel, uint32_t old) { return __builtin_amdgcn_cvt_pk_u8_f32(minNoNan(v, RNE ? 254.4f : 254.9f), sel, old); }      // (a NaN quotient - weight 1 de-integrated - is discarded by the caller)

struct ApxCol { float nx0, ny0, zc; };       // per lane (column x, y) and pose: image-coordinate numerators at z = 0, R6 * xw + R7 * yw

BF_DEV ApxCol apxCol(const ApxPose& p, float ix, float iy, float xw, float yw) {
    ApxCol c;
    c.nx0 = __builtin_fmaf(p.ax, ix, __builtin_fmaf(p.bx, iy, p.dx));
    c.ny0 = __builtin_fmaf(p.ay, ix, __builtin_fmaf(p.by, iy, p.dy));
    c.zc = p.r6 * xw + p.r7 * yw;
    return c;
}

struct ApxSample { v2f pcz; uint32_t offA, offB; bool inA, inB; };

#ifdef BF_VAR_SCALAR_PROJECT          // diagnostic variant (tools/build_variant.py): the projection as scalar instructions - the second voxel's operations depend (through an
BF_DEV float afterF(float v, float dep) { asm volatile("" : "+v"(v) : "v"(dep)); return v; }      // empty asm) on the first voxel's results, so the compiler cannot pair them into v_pk_*_f32
#endif
BF_DEV ApxSample apxProject(const ApxCam& c, const ApxPose& p, const ApxCol& col, v2f iz, v2f pz, bool use) {
    ApxSample o;
#ifdef BF_VAR_SCALAR_PROJECT          // bit 0 the depth, 1 the numerators, 2 the image coordinates (7: the whole projection)
    v2f nx, ny, r, hx, hy;
    if (BF_VAR_SCALAR_PROJECT & 1) { o.pcz.x = (col.zc + p.r8 * pz.x) + p.t2; o.pcz.y = (col.zc + p.r8 * afterF(pz.y, o.pcz.x)) + p.t2; }
    else o.pcz = (sp2(col.zc) + sp2(p.r8) * pz) + sp2(p.t2);
    if (BF_VAR_SCALAR_PROJECT & 2) {
        nx.x = __builtin_fmaf(p.cx, iz.x, col.nx0); ny.x = __builtin_fmaf(p.cy, iz.x, col.ny0);
        const float izy = afterF(iz.y, nx.x + ny.x);
        nx.y = __builtin_fmaf(p.cx, izy, col.nx0); ny.y = __builtin_fmaf(p.cy, izy, col.ny0);
    } else { nx = pkfma(sp2(p.cx), iz, sp2(col.nx0)); ny = pkfma(sp2(p.cy), iz, sp2(col.ny0)); }
    r.x = __builtin_amdgcn_rcpf(o.pcz.x); r.y = __builtin_amdgcn_rcpf(o.pcz.y);
    if (BF_VAR_SCALAR_PROJECT & 4) {
        hx.x = __builtin_fmaf(nx.x, r.x, c.mxh); hy.x = __builtin_fmaf(ny.x, r.x, c.myh);
        const float ry = afterF(r.y, hx.x + hy.x);
        hx.y = __builtin_fmaf(nx.y, ry, c.mxh); hy.y = __builtin_fmaf(ny.y, ry, c.myh);
    } else { hx = pkfma(nx, r, sp2(c.mxh)); hy = pkfma(ny, r, sp2(c.myh)); }
#else
#ifndef BF_VAR_PAD          // diagnostic variants: idle issue slots (s_nop 4, pinned by scheduling barriers) at one place of the projection - bit 0 in front of it, 1 depth -> reciprocal,
#define BF_VAR_PAD 0        // 2 reciprocal -> image coordinates, 3 image coordinates -> conversion
#endif
#define BF_PAD_AT(bit) do { if (BF_VAR_PAD & (1 << (bit))) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 4"); __builtin_amdgcn_sched_barrier(0); } } while (0)
    BF_PAD_AT(0);
    o.pcz = (sp2(col.zc) + sp2(p.r8) * pz) + sp2(p.t2);
    const v2f nx = pkfma(sp2(p.cx), iz, sp2(col.nx0)), ny = pkfma(sp2(p.cy), iz, sp2(col.ny0));
    BF_PAD_AT(1);
    v2f r; r.x = __builtin_amdgcn_rcpf(o.pcz.x); r.y = __builtin_amdgcn_rcpf(o.pcz.y);
    BF_PAD_AT(2);
    const v2f hx = pkfma(nx, r, sp2(c.mxh)), hy = pkfma(ny, r, sp2(c.myh));
    BF_PAD_AT(3);
#endif
    const uint32_t pxA = (uint32_t)f2iHw(hx.x), pyA = (uint32_t)f2iHw(hy.x), pxB = (uint32_t)f2iHw(hx.y), pyB = (uint32_t)f2iHw(hy.y);
    o.inA = use && pxA < c.W && pyA < c.H; o.inB = use && pxB < c.W && pyB < c.H;
    o.offA = o.inA ? (__umul24(pyA, c.W) + pxA) << 2 : 0xFFFFFFFFu;      // beyond the descriptor's range: the load returns 0
    o.offB = o.inB ? (__umul24(pyB, c.W) + pxB) << 2 : 0xFFFFFFFFu;
    return o;
}

// Per block: what the lanes of the wave need of the list entry (wave-uniform) and of their column (x, y)
struct ApxBlock {
    uint32_t* base;            // voxel (x, y, 0) of this lane
    float kz;                  // (float)(8 * block z)
    ApxCol cDe, cIn;
    bool useDe, useIn;         // wave-uniform: the block lies in the frustum of the old / new pose
};
// One voxel pair (z, z + 1) of a block: the two voxels, and depth + colour of the pixels both poses project them to.
struct ApxPair {
    v2f vS, vW; uint32_t vCA, vCB;
    v2f pczDe, pczIn, dDe, dIn;
    uint32_t kDeA, kDeB, kInA, kInB;
    bool inDeA, inDeB, inInA, inInB;
};

template <bool DE, bool IN>
BF_DEV ApxBlock apxBlock(const Dev& d, const ApxCam& c, const ApxPose& pIn, const ApxPose& pDe, int4 e, uint32_t flags, uint32_t lane) {
    ApxBlock b;
    const float ix = (float)(e.x * BS + (int)(lane & 7)), iy = (float)(e.y * BS + (int)(lane >> 3));
    const float xw = ix * c.voxelSize, yw = iy * c.voxelSize;
    b.useDe = DE && (flags & 2u); b.useIn = IN && (flags & 1u);
    b.cDe = apxCol(pDe, ix, iy, xw, yw);
    b.cIn = apxCol(pIn, ix, iy, xw, yw);
    b.base = reinterpret_cast<uint32_t*>(d.vox + ((size_t)(uint32_t)e.w + lane));
    b.kz = (float)(e.z * BS);
    return b;
}

// depth and colour of one pixel: ONE 8-byte gather from the interleaved image the prep stream builds per operator (k_interleave) - half the
// vector-memory instructions of the two-plane form (two 4-byte gathers; measured 108.0 -> 88.9 us per fused launch, gpurun r03n; removed in round 4)
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct ApxTexel { float dep; uint32_t col; };
BF_DEV ApxTexel apxGather(__amdgpu_buffer_rsrc_t texRes, uint32_t off) {
    ApxTexel r;
    const v2u t = __builtin_amdgcn_raw_buffer_load_b64(texRes, (int)(off << 1), 0, 0);      // 0xFFFFFFFF << 1 stays beyond the range
    r.dep = __uint_as_float(t.x); r.col = t.y;
    return r;
}

__global__ void k_interleave(const float* __restrict__ depth, const uint32_t* __restrict__ color, uint2* __restrict__ texel, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) texel[i] = make_uint2(__float_as_uint(depth[i]), color[i]);
}

// the samples of a pair: projection of its two voxels under both poses and the four texel gathers (no load depends on another load)
template <bool DE, bool IN>
BF_DEV void apxSamples(const ApxCam& c, const ApxPose& pIn, const ApxPose& pDe, const ApxBlock& b, int z, __amdgpu_buffer_rsrc_t texRes, ApxPair& o) {
    v2f iz; iz.x = b.kz + (float)z; iz.y = b.kz + (float)(z + 1);                 // exact small integers
    const v2f pz = iz * sp2(c.voxelSize);
    o.pczDe = o.pczIn = sp2(0.0f); o.dDe = o.dIn = sp2(0.0f); o.kDeA = o.kDeB = o.kInA = o.kInB = 0u;
    o.inDeA = o.inDeB = o.inInA = o.inInB = false;
    if (DE) {
        const ApxSample a = apxProject(c, pDe, b.cDe, iz, pz, b.useDe);
        o.pczDe = a.pcz; o.inDeA = a.inA; o.inDeB = a.inB;
        const ApxTexel tA = apxGather(texRes, a.offA), tB = apxGather(texRes, a.offB);
        o.dDe.x = tA.dep; o.kDeA = tA.col; o.dDe.y = tB.dep; o.kDeB = tB.col;
    }
    if (IN) {
        const ApxSample a = apxProject(c, pIn, b.cIn, iz, pz, b.useIn);
        o.pczIn = a.pcz; o.inInA = a.inA; o.inInB = a.inB;
        const ApxTexel tA = apxGather(texRes, a.offA), tB = apxGather(texRes, a.offB);
        o.dIn.x = tA.dep; o.kInA = tA.col; o.dIn.y = tB.dep; o.kInB = tB.col;
    }
}

// the pair's two voxel slices (768 contiguous bytes each); ldA / ldB wave-uniform: a slice no lane has a valid sample for is not read (its registers
// hold zeros that are computed on and never stored)
BF_DEV void apxLoadVoxels(const ApxBlock& b, int z, ApxPair& o, bool ldA, bool ldB) {
    const uint32_t* vpA = b.base + (size_t)z * 64u * 3u; const uint32_t* vpB = vpA + 64u * 3u;
    o.vS = sp2(0.0f); o.vW = sp2(0.0f); o.vCA = o.vCB = 0u;
    if (ldA) { o.vS.x = __uint_as_float(vpA[0]); o.vW.x = __uint_as_float(vpA[1]); o.vCA = vpA[2]; }
    if (ldB) { o.vS.y = __uint_as_float(vpB[0]); o.vW.y = __uint_as_float(vpB[1]); o.vCB = vpB[2]; }
}

// which voxels of a pair have a valid sample (the conditions of apxStageB, on the same values)
template <bool DE, bool IN>
BF_DEV void apxTouched(const ApxCam& c, const ApxPair& a, bool& anyA, bool& anyB) {
    const v2f sDe = a.dDe - a.pczDe, sIn = a.dIn - a.pczIn;
    const v2f tDe = sp2(c.truncation) + sp2(c.truncScale) * a.dDe, tIn = sp2(c.truncation) + sp2(c.truncScale) * a.dIn;
    const bool okDeA = DE && a.inDeA && a.dDe.x < c.maxDist && fabsf(sDe.x) < tDe.x, okDeB = DE && a.inDeB && a.dDe.y < c.maxDist && fabsf(sDe.y) < tDe.y;
    const bool okInA = IN && a.inInA && a.dIn.x < c.maxDist && fabsf(sIn.x) < tIn.x, okInB = IN && a.inInB && a.dIn.y < c.maxDist && fabsf(sIn.y) < tIn.y;
    anyA = okDeA || okInA; anyB = okDeB || okInB;
}

// Stage B on registers: sample validity, voxelApply<true> and / or voxelApply<false> on the pair's two voxels (vS, vW, vCA, vCB); stA / stB: the voxel has a
// valid sample (its new value is in the registers), otherwise its registers are unchanged.
template <bool DE, bool IN, bool RNE>
BF_DEV void apxCompute(const ApxCam& c, const ApxPair& a, v2f& vS, v2f& vW, uint32_t& vCA, uint32_t& vCB, bool& stA, bool& stB) {
    // sample validity (voxelSample): the depth -inf of an invalid pixel fails |sdf| < trunc by itself; |sdf| < trunc makes the reference's
    // clamp to [-trunc, trunc] the identity.  Truncation in the exact contract's operations: the validity of a sample (hence every
    // weight) does not depend on the contract.
    const v2f sDe = a.dDe - a.pczDe, sIn = a.dIn - a.pczIn;
    const v2f tDe = sp2(c.truncation) + sp2(c.truncScale) * a.dDe, tIn = sp2(c.truncation) + sp2(c.truncScale) * a.dIn;
    const bool okDeA = DE && a.inDeA && a.dDe.x < c.maxDist && fabsf(sDe.x) < tDe.x, okDeB = DE && a.inDeB && a.dDe.y < c.maxDist && fabsf(sDe.y) < tDe.y;
    const bool okInA = IN && a.inInA && a.dIn.x < c.maxDist && fabsf(sIn.x) < tIn.x, okInB = IN && a.inInB && a.dIn.y < c.maxDist && fabsf(sIn.y) < tIn.y;
    stA = okDeA || okInA; stB = okDeB || okInB;
    if (!stA && !stB) return;
    if (DE && (okDeA || okDeB)) {           // voxelApply<true>
        const v2f dd = vW - sp2(1.0f);
            v2f r; r.x = __builtin_amdgcn_rcpf(dd.x); r.y = __builtin_amdgcn_rcpf(dd.y);
            uint32_t nA = 0xFF000000u, nB = 0xFF000000u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v2f o, cc; o.x = byteF(vCA, k); o.y = byteF(vCB, k); cc.x = byteF(a.kDeA, k); cc.y = byteF(a.kDeB, k);
            // roundf of the exact contract = nearest, ties AWAY from zero, and (o w - c) / (w - 1) is an exact tie for every second value at small
            // weights (w = 3: half-integers): the conversion's nearest-EVEN would be 1 LSB off there, and the next operators amplify it.  The
            // bias pushes ties up; a non-tie is at least 1 / (2 (w - 1)) away from one, the quotient's own error is < 7e-5 - exact for w < 2000.
            const v2f q = pkfma(pkfma(o, vW, -cc), r, sp2(RNE ? 0x1p-12f : 0.5f));
            nA = packByte<RNE>(q.x, (uint32_t)k, nA); nB = packByte<RNE>(q.y, (uint32_t)k, nB);
        }
        const v2f s = pkfma(vS, vW, -sDe) * r;
        float sA = s.x, sB = s.y, wA = maxNoNan(dd.x, 0.0f), wB = maxNoNan(dd.y, 0.0f);
        if (wA <= 0.001f) { sA = 0.0f; nA = 0u; wA = 0.0f; }
        if (wB <= 0.001f) { sB = 0.0f; nB = 0u; wB = 0.0f; }
        if (okDeA) { vS.x = sA; vW.x = wA; vCA = nA; }
        if (okDeB) { vS.y = sB; vW.y = wB; vCB = nB; }
    }
    if (IN && (okInA || okInB)) {           // voxelApply<false>
        const v2f dd = sp2(1.0f) + vW;
            v2f r; r.x = __builtin_amdgcn_rcpf(dd.x); r.y = __builtin_amdgcn_rcpf(dd.y);
            v2f ca, cb;                         // colour blend 0.2 new + 0.8 old; a voxel without weight takes the new colour
        ca.x = vW.x == 0.0f ? 1.0f : 0.2f; cb.x = vW.x == 0.0f ? 0.0f : 0.8f;
        ca.y = vW.y == 0.0f ? 1.0f : 0.2f; cb.y = vW.y == 0.0f ? 0.0f : 0.8f;
        uint32_t nA = 0xFF000000u, nB = 0xFF000000u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v2f o, cc; o.x = byteF(vCA, k); o.y = byteF(vCB, k); cc.x = byteF(a.kInA, k); cc.y = byteF(a.kInB, k);
            const v2f m = pkfma(o, cb, RNE ? cc * ca : pkfma(cc, ca, sp2(0.5f)));
            nA = packByte<RNE>(m.x, (uint32_t)k, nA); nB = packByte<RNE>(m.y, (uint32_t)k, nB);
        }
        const v2f s = pkfma(vS, vW, sIn) * r;
        if (okInA) { vS.x = s.x; vW.x = minNoNan(dd.x, c.weightMax); vCA = nA; }
        if (okInB) { vS.y = s.y; vW.y = minNoNan(dd.y, c.weightMax); vCB = nB; }
    }
}

// Stage B of the per-operator kernels: compute, store.
template <bool DE, bool IN, bool RNE>
BF_DEV void apxStageB(const ApxCam& c, const ApxBlock& b, int z, const ApxPair& a) {
    uint32_t* vpA = b.base + (size_t)z * 64u * 3u; uint32_t* vpB = vpA + 64u * 3u;
    v2f vS = a.vS, vW = a.vW; uint32_t vCA = a.vCA, vCB = a.vCB;
    bool stA, stB;
    apxCompute<DE, IN, RNE>(c, a, vS, vW, vCA, vCB, stA, stB);
    if (stA) { vpA[0] = __float_as_uint(vS.x); vpA[1] = __float_as_uint(vW.x); vpA[2] = vCA; }
    if (stB) { vpB[0] = __float_as_uint(vS.y); vpB[1] = __float_as_uint(vW.y); vpB[2] = vCB; }
}

struct ApxEntry { int4 e; uint32_t flags; };
template <int MODE>
BF_DEV ApxEntry apxEntry(const Dev& d, uint32_t blk) {
    ApxEntry r;
    r.e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];                                              // wave-uniform
    r.flags = MODE == 2 ? reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4] : 3u;
    return r;
}

// One wave per block of the frustum (or union) list, lane = (x, y) column, the eight z walked as four pairs.
// A pair's samples are gathered first and each of its two voxel slices is loaded only when some lane of the wave has a valid sample for it - two dependent round
// trips per touched pair, no voxel traffic at all for an untouched slice (about 2.5 of a block's 8 slices are touched by no lane).  Against speculative loads
// beside the samples (one round trip; the form until round 4, removed in round 5): 79.2 -> 73.4 us per fused launch (gpurun r04a, profiles/r04_update_variants.md).
// Measured and withdrawn in round 3 (profiles/r03_lds_footprint.md; the code is in the history): the block's pixel footprint staged through LDS
// (83 us), plus all eight slices in one round trip (79.2), plus per-slice loads behind a second sampling pass (104), whole-row write-backs (79.7),
// stage A of the next pair issued before stage B of the current one (113: 80 VGPRs -> 6 waves), the body held to 80 SGPRs (no change), 4096 / 2048
// workgroups (97 / 106).
template <int MODE, bool RNE>
__global__ __launch_bounds__(256) void k_update_apx(Dev d, ApxCam c, ApxPose in, ApxPose de, const uint2* __restrict__ tex, int hasColor, int accumulate) {
    if (!hasColor) return;          // CUDASceneRepHashSDF.cu:441-448: without colour data `color.x != MINF` never holds
    constexpr bool DE = MODE != 0, IN = MODE != 1;
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), nWaves = gridDim.x * 4u;
    if (accumulate && wave == 0 && lane == 0) {          // block accounting of the timed launches
        if (MODE == 2) { d.occSum[2] += (unsigned long long)n; d.occSum[0] += (unsigned long long)(uint32_t)d.compactCount[1]; }
        else { d.occSum[0] += (unsigned long long)n; d.occSum[1] += (unsigned long long)n; }
    }
    if (wave >= n) return;
    const __amdgpu_buffer_rsrc_t texRes = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(tex), 0, (int)(2u * c.bytes), 0x00020000);
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const ApxEntry en = apxEntry<MODE>(d, blk);
        const ApxBlock cur = apxBlock<DE, IN>(d, c, in, de, en.e, en.flags, lane);
#pragma unroll 1
        for (int z = 0; z < 8; z += 2) {
            ApxPair pa;
            apxSamples<DE, IN>(c, in, de, cur, z, texRes, pa);
            bool anyA, anyB;
            apxTouched<DE, IN>(c, pa, anyA, anyB);
            const bool ldA = __builtin_amdgcn_ballot_w64(anyA) != 0ull, ldB = __builtin_amdgcn_ballot_w64(anyB) != 0ull;      // wave-uniform
            if (!ldA && !ldB) continue;                                               // nothing of this pair is read or written
            apxLoadVoxels(cur, z, pa, ldA, ldB);
            apxStageB<DE, IN, RNE>(c, cur, z, pa);
        }
    }
}


__global__ void k_probe_cvt(uint32_t* out) {
    const float v[8] = {0.5f, 1.5f, 2.5f, 2.7f, 254.4f, 300.0f, -3.0f, 3.49f};
    if (threadIdx.x < 8) out[threadIdx.x] = __builtin_amdgcn_cvt_pk_u8_f32(v[threadIdx.x], 1u, 0xAABBCCDDu);
}

#include "tsdf_batch.h"

// ---------------------------------------------------------------------------------------
// garbage collection (CUDASceneRepHashSDF.cu:584-668, VoxelUtilHashSDF.h:740-826)
// ---------------------------------------------------------------------------------------
// needMask != 0: d.compact is a union list (fused re-integration, batch) whose entries with (flags & needMask) != 0 are exactly the frustum list of the last
// pose - the list the reference's garbageCollect walks (its last compactify) - so the union list is filtered here instead of being re-compacted first
__global__ __launch_bounds__(256) void k_gc_identify(Dev d, Frame f, uint32_t needMask) {
    __shared__ uint32_t wmax[4];
    const uint32_t n = (uint32_t)d.compactCount[0];
    for (uint32_t blk = blockIdx.x; blk < n; blk += gridDim.x) {
        if (needMask != 0u && (reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4] & needMask) == 0u) continue;      // block-uniform
        const int4 e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];
        const bf_voxel* v = d.vox + (size_t)(uint32_t)e.w;
        const uint32_t w0 = (uint32_t)f2i(v[2 * threadIdx.x + 0].weight);      // uint shared_MaxWeight, .cu:581,606
        const uint32_t w1 = (uint32_t)f2i(v[2 * threadIdx.x + 1].weight);
        uint32_t m = wave_max_u(max(w0, w1));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
            if (m == 0) {
                i3 b; b.x = e.x; b.y = e.y; b.z = e.z;
                const uint32_t h = hashPos(f.numBuckets, b);
                const uint32_t bin = (uint32_t)(((uint64_t)h * NBINS) / f.numBuckets);
                const uint32_t pos = atomicAdd(&d.binCount[bin], 1u);
                if (pos < BINCAP) {
                    BinRec r; r.key = packKey(b); r.bucket = h; r.aux = d.compactSrc[blk];
                    d.bins[(size_t)bin * BINCAP + pos] = r;
                } else {
                    atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW);
                }
            }
        }
        __syncthreads();
    }
}

// serial deleteHashEntryElement; returns the freed ptr or FREE_ENTRY if the key is absent
BF_DEV int32_t deleteEntry(const Dev& d, const Frame& f, i3 b, uint32_t h) {
    const uint32_t hp = h * BF_HASH_BUCKET_SIZE;
    const uint32_t total = BF_HASH_BUCKET_SIZE * f.numBuckets;
    uint4* h4 = reinterpret_cast<uint4*>(d.hash);
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        const uint32_t i = hp + j;
        const bf_hash_entry c = d.hash[i];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) {
            if (c.offset != 0) {
                const uint32_t next = (i + c.offset) % total;
                h4[(size_t)i * 2] = h4[(size_t)next * 2];
                h4[(size_t)i * 2 + 1] = h4[(size_t)next * 2 + 1];
                h4[(size_t)next * 2] = make_uint4(0, 0, 0, (uint32_t)BF_FREE_ENTRY);
                h4[(size_t)next * 2 + 1] = make_uint4(0, 0, 0, 0);
            } else {
                h4[(size_t)i * 2] = make_uint4(0, 0, 0, (uint32_t)BF_FREE_ENTRY);
                h4[(size_t)i * 2 + 1] = make_uint4(0, 0, 0, 0);
            }
            return c.ptr;
        }
    }
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1;
    bf_hash_entry c = d.hash[last];
    uint32_t prev = last;
    uint32_t i = (last + c.offset) % total;
    for (uint32_t it = 0; it < f.maxChain; ++it) {
        c = d.hash[i];
        if (c.pos[0] == b.x && c.pos[1] == b.y && c.pos[2] == b.z && c.ptr != BF_FREE_ENTRY) {
            h4[(size_t)i * 2] = make_uint4(0, 0, 0, (uint32_t)BF_FREE_ENTRY);
            h4[(size_t)i * 2 + 1] = make_uint4(0, 0, 0, 0);
            d.hash[prev].offset = c.offset;
            return c.ptr;
        }
        if (c.offset == 0) return BF_FREE_ENTRY;
        prev = i;
        i = (last + c.offset) % total;
    }
    return BF_FREE_ENTRY;
}

__global__ __launch_bounds__(1024) void k_gc_delete(Dev d, Frame f) {
    __shared__ SortLds s;
    __shared__ uint32_t scratch[16];
    const uint32_t bin = blockIdx.x;
    const uint32_t n = min(d.binCount[bin], BINCAP);
    if (n == 0) return;
    loadBinSorted(s, d.bins + (size_t)bin * BINCAP, n);
    const uint32_t base = binPrefix(d.binCount, bin, scratch);
    const uint32_t heapC = d.heapCounter[0];
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint32_t h = s.bucket[idx];
        if (idx > 0 && s.bucket[idx - 1] == h) continue;            // one thread per home bucket
        for (uint32_t k = idx; k < n && s.bucket[k] == h; ++k) {
            const int32_t ptr = deleteEntry(d, f, unpackKey(s.key[k]), h);
            d.allocList[s.aux[k]].ptr = BF_FREE_ENTRY;
            if (ptr == BF_FREE_ENTRY) { atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_GC_MISSING); s.aux[k] = 0xFFFFFFFFu; continue; }
            d.heap[heapC + 1u + base + k] = (uint32_t)ptr / (uint32_t)VOX;   // appendHeap, rank-ordered
            s.aux[k] = (uint32_t)ptr;
        }
    }
    __syncthreads();
    // clear the freed voxel blocks (.cu:662-665): 6144 B = 384 x 16 B each
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t ptr = s.aux[k];
        if (ptr == 0xFFFFFFFFu) continue;
        uint4* v4 = reinterpret_cast<uint4*>(d.vox + (size_t)ptr);
        for (uint32_t t = threadIdx.x; t < (uint32_t)(VOX * 12 / 16); t += blockDim.x) v4[t] = make_uint4(0, 0, 0, 0);
    }
}

__global__ __launch_bounds__(256) void k_gc_finish(Dev d) {
    __shared__ uint32_t scratch[16];
    const uint32_t D = binPrefix(d.binCount, NBINS, scratch);
    if (threadIdx.x == 0) d.heapCounter[0] += D;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < NBINS; b += blockDim.x) d.binCount[b] = 0;
}

}  // namespace

// =========================================================================================
// host side: bf_scene == CUDASceneRepHashSDF
// =========================================================================================
struct bf_scene {
    bf_hash_params params;
    bf_depth_camera_params cam;
    bool haveCam = false;
    Dev d{};
    hipStream_t stream = nullptr;
    uint32_t numIntegrated = 0;
    uint32_t dedupeSize = 0;
    uint32_t gridCompact = 0, gridUpdateCol = 0, gridUpdateColPlain = 0;
    bool forceExactDiv = false;     // k_update_col takes the literal `/` path for every block (BF_TSDF_EXACT_DIV=1; tests)
    bool externalAlloc = false;     // bf_scene_set_external_alloc: integrate / re-integrate do not allocate (the caller ran bf_scene_alloc_collect / _ingest)
    // bf_scene_set_alloc_comm: the operators' own allocation with the ray march divided over the ranks of a communicator - collect on this rank's band of the
    // pixel tiles, ONE all-gather of fixed-size records {count, keys[capacity]} on the allocation stream, ingest of every rank's list, placement
    bf_comm* allocComm = nullptr;
    uint32_t allocCap = 0;          // keys per rank and operator
    uint8_t *d_allocSend = nullptr, *d_allocRecv = nullptr; uint32_t* d_allocSlots = nullptr;
    uint8_t *d_batchSend = nullptr, *d_batchRecv = nullptr;      // the same exchange for a whole batch: records {count, pad, {key, operator mask, pad}[allocCap]}
    int arith = BF_TSDF_ARITH_FAST; // bf_scene_set_arith / BF_TSDF_ARITH: fast (k_update_apx: the contract of the reference's own GPU build; default since round 4) or
                                    // exact (k_update_col: IEEE op by op, bit-comparable with a host build of the reference and with the oracle)
    int cvtRne = -1;                // what v_cvt_pk_u8_f32 does on this device: 1 nearest-even, 0 truncation, -1 not probed yet
    uint2* texel[8] = {}; size_t texelPixels = 0;      // the operator's frame as 8-byte {depth, colour} texels (k_interleave), one per list buffer (NB)
    int32_t* d_hashDecision = nullptr;
    uint32_t shardLo = 0, shardHi = 0xFFFFFFFFu;      // bf_scene_set_shard
    uint32_t opsTimed = 0;          // integrate / de-integrate operations covered by the timed launches (a fused launch counts 2)
    uint32_t imagesTimed = 0;       // frames (depth + colour images) the timed launches sampled: one per operator, n per batch of n
    // Software pipelining of consecutive operators (bf_scene_set_overlap): allocation + frustum compaction of operator n+1 run on
    // the internal `prep` stream while the voxel update of operator n runs on `stream`.  The update never reads the hash table and
    // allocation never touches voxels; the only shared object is the frustum list, which is double-buffered.
    bool overlap = false;
    hipStream_t prep = nullptr;
    // NB list buffers: allocation + list of operator n+1 .. n+NB-1 may be prepared while operator n updates voxels.  (With two buffers the
    // prep stream had to wait for the update two operators back, and the two cross-stream event hops of ~40 us each sat inside the
    // steady-state cycle: period = hop + (prep + update) / 2, profiles/r02_pipeline_timeline.txt: 44 us idle between consecutive updates.)
    static constexpr int NBMAX = 8;
    int NB = 4;                     // list buffers in use (2 .. NBMAX)
    bf_hash_entry* cbuf[NBMAX] = {}; uint32_t* csrc[NBMAX] = {}; int32_t* ccnt[NBMAX] = {};
    int cur = 0;                    // buffer that holds the latest list (== d.compact / d.compactSrc / d.compactCount)
    hipEvent_t evPrep[NBMAX] = {}, evUpd[NBMAX] = {}, evBarrier = nullptr, evTmp = nullptr;
    bool updRecorded[NBMAX] = {};
    bool barrierPending = false;    // the last exclusive section of the main stream has not been waited for by the preparation stream yet
    hipEvent_t pendingEv = nullptr; // bf_scene_wait_event: the next operator's first kernel waits for it
    const uint2* frameTexels = nullptr;   // bf_scene_set_frame_texels: the next operator's frame as interleaved texels, made once when the frame was ingested
    bool compactStale = false;      // d.compact holds a union list (fused re-integration, batch) or nothing usable (behind a garbage collection), not the frustum list of the last pose
    uint32_t gcMask = 0;            // compactStale and != 0: the entries of d.compact with (flags & gcMask) != 0 ARE the frustum list of the last pose (k_gc_identify filters)
    // batched operators (bf_scene_run_batch, tsdf_batch.h)
    BatchDev bd{};
    bool batchReady = false;
    uint2* btexel[NBMAX][BF_SCENE_BATCH_MAX] = {}; size_t btexelPixels = 0;      // the batch's frames as texel images, one set per list buffer (the NB in use)
    // diagnostic BF_DEBUG_VERIFY_BATCH=<file> (tsdf_batch.h k_verify_*): two shadow copies of the voxels, the mismatch log in pinned host memory
    uint32_t updateLds = 0;       // (diagnostic, BF_DEBUG_UPDATE_LDS: unused dynamic LDS per workgroup of the batched fast update = a cap on its workgroups per CU)
    const char* verifyPath = nullptr; bf_voxel* vshadow[2] = {}; VerifyLog* vlog = nullptr; uint32_t vseq = 0;
    // optional HIP-event timing of the voxel-update kernel
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t eventsUsed = 0;
    std::vector<void*> allocations;
};

#define BF_TRY_RC(expr) do { int _rc = (expr); if (_rc != BF_OK) return _rc; } while (0)

namespace {

template <class T>
int devAlloc(bf_scene* s, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = BF_MALLOC(&q, count * sizeof(T));
    if (e != hipSuccess) { set_error("BF_MALLOC(%zu B) failed: %s", count * sizeof(T), hipGetErrorString(e)); return BF_ERR_HIP; }
    s->allocations.push_back(q);
    *p = (T*)q;
    return BF_OK;
}

Frame makeFrame(const bf_scene* s) {
    Frame f;
    memcpy(f.T.e, s->params.m_rigidTransform, 64);
    memcpy(f.Tinv.e, s->params.m_rigidTransformInverse, 64);
    f.cam = s->cam;
    f.numBuckets = s->params.m_hashNumBuckets;
    f.maxChain = s->params.m_hashMaxCollisionLinkedListSize;
    f.numSDFBlocks = s->params.m_numSDFBlocks;
    f.voxelSize = s->params.m_virtualVoxelSize;
    f.maxIntegrationDistance = s->params.m_maxIntegrationDistance;
    f.truncScale = s->params.m_truncScale;
    f.truncation = s->params.m_truncation;
    f.weightMax = (float)s->params.m_integrationWeightMax;
    f.shardLo = s->shardLo; f.shardHi = s->shardHi;
    return f;
}

UpdCam makeUpdCam(const Frame& f) {
    UpdCam u;
    u.fx = f.cam.fx; u.fy = f.cam.fy; u.mx = f.cam.mx; u.my = f.cam.my;
    u.voxelSize = f.voxelSize; u.maxDist = f.maxIntegrationDistance; u.truncScale = f.truncScale; u.truncation = f.truncation; u.weightMax = f.weightMax;
    u.W = f.cam.m_imageWidth; u.H = f.cam.m_imageHeight;
    return u;
}
UpdPose makeUpdPose(const Frame& f) {
    UpdPose u;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) u.R[r * 3 + c] = f.Tinv.e[r * 4 + c]; u.t[r] = f.Tinv.e[r * 4 + 3]; }
    float rmax = 0.0f; bool ok = true;
    for (int i = 0; i < 9; ++i) { ok = ok && std::isfinite(u.R[i]); rmax = std::max(rmax, std::fabs(u.R[i])); }
    for (int i = 0; i < 3; ++i) ok = ok && std::isfinite(u.t[i]) && std::fabs(u.t[i]) <= 4096.0f;
    u.rmax = rmax;
    // the ranges under which the shared-reciprocal quotients of colFast are the IEEE quotients (see the kernel's header comment)
    const UpdCam c = makeUpdCam(f);
    ok = ok && rmax <= 16.0f && c.fx >= 0x1p-10f && c.fx <= 0x1p20f && c.fy >= 0x1p-10f && c.fy <= 0x1p20f && std::isfinite(c.mx) && std::isfinite(c.my) &&
         c.voxelSize >= 1e-6f && c.voxelSize <= 16.0f && c.weightMax >= 1.0f && c.weightMax <= 0x1p39f && std::isfinite(c.truncation) && std::isfinite(c.truncScale) &&
         std::isfinite(c.maxDist);
    u.fastOk = ok ? 1u : 0u;
    return u;
}

ApxCam makeApxCam(const Frame& f) {
    ApxCam u;
    u.mxh = f.cam.mx + 0.5f; u.myh = f.cam.my + 0.5f;
    u.voxelSize = f.voxelSize; u.maxDist = f.maxIntegrationDistance; u.truncScale = f.truncScale; u.truncation = f.truncation; u.weightMax = f.weightMax;
    u.W = f.cam.m_imageWidth; u.H = f.cam.m_imageHeight; u.bytes = u.W * u.H * 4u;
    return u;
}
ApxPose makeApxPose(const Frame& f) {
    ApxPose u;
    const double fx = f.cam.fx, fy = f.cam.fy, vs = f.voxelSize;
    const float* M = f.Tinv.e;
    u.ax = (float)(fx * vs * M[0]); u.bx = (float)(fx * vs * M[1]); u.cx = (float)(fx * vs * M[2]); u.dx = (float)(fx * M[3]);
    u.ay = (float)(fy * vs * M[4]); u.by = (float)(fy * vs * M[5]); u.cy = (float)(fy * vs * M[6]); u.dy = (float)(fy * M[7]);
    u.r6 = M[8]; u.r7 = M[9]; u.r8 = M[10]; u.t2 = M[11];
    return u;
}

// one-time probe of v_cvt_pk_u8_f32 (rounding mode and saturation) on the scene's device
int probeCvt(bf_scene* s) {
    if (s->cvtRne >= 0) return BF_OK;
    uint32_t* d_out = nullptr;
    BF_HIP_TRY(BF_MALLOC((void**)&d_out, 8 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_probe_cvt, dim3(1), dim3(64), 0, s->stream, d_out);
    uint32_t h[8];
    hipError_t e = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(d_out);
    if (e != hipSuccess) { set_error("cvt probe failed: %s", hipGetErrorString(e)); return BF_ERR_HIP; }
    uint32_t b[8];
    for (int i = 0; i < 8; ++i) {
        if ((h[i] & 0xFFFF00FFu) != 0xAABB00DDu) { set_error("v_cvt_pk_u8_f32 probe: unexpected packing 0x%08x", h[i]); return BF_ERR_HIP; }
        b[i] = (h[i] >> 8) & 0xFFu;
    }
    // inputs 0.5 1.5 2.5 2.7 254.4 300 -3 3.49
    const uint32_t rne[8] = {0, 2, 2, 3, 254, 255, 0, 3}, rtz[8] = {0, 1, 2, 2, 254, 255, 0, 3};
    if (memcmp(b, rne, sizeof b) == 0) s->cvtRne = 1;
    else if (memcmp(b, rtz, sizeof b) == 0) s->cvtRne = 0;
    else { set_error("v_cvt_pk_u8_f32 probe: neither nearest-even nor truncation (%u %u %u %u %u %u %u %u)", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]); return BF_ERR_HIP; }
    return BF_OK;
}

template <int MODE>
void launchApx(bf_scene* s, uint32_t grid, const Dev& dv, const ApxCam& c, const ApxPose& in, const ApxPose& de, const uint2* tex, int hasColor, int acc) {
    if (s->cvtRne) hipLaunchKernelGGL((k_update_apx<MODE, true>), dim3(grid), dim3(256), 0, s->stream, dv, c, in, de, tex, hasColor, acc);
    else hipLaunchKernelGGL((k_update_apx<MODE, false>), dim3(grid), dim3(256), 0, s->stream, dv, c, in, de, tex, hasColor, acc);
}

void setLastRigidTransform(bf_scene* s, const float* T) {       // CUDASceneRepHashSDF.h:128-134
    m44 m;
    memcpy(m.e, T, 64);
    const m44 inv = inverse44(m);
    memcpy(s->params.m_rigidTransform, m.e, 64);
    memcpy(s->params.m_rigidTransformInverse, inv.e, 64);
}

Dev devBuf(const bf_scene* s, int b) {
    Dev d = s->d;
    d.compact = s->cbuf[b]; d.compactSrc = s->csrc[b]; d.compactCount = s->ccnt[b]; d.allocSnap = reinterpret_cast<uint32_t*>(s->ccnt[b]) + 2;
    return d;
}
void useBuf(bf_scene* s, int b) { s->cur = b; s->d.compact = s->cbuf[b]; s->d.compactSrc = s->csrc[b]; s->d.compactCount = s->ccnt[b]; s->d.allocSnap = reinterpret_cast<uint32_t*>(s->ccnt[b]) + 2; }

// exclusive section on the main stream: everything issued on `prep` so far happens before, everything issued on `prep` later after
int beginExclusive(bf_scene* s) {
    if (!s->overlap) return BF_OK;
    BF_HIP_TRY(hipEventRecord(s->evTmp, s->prep));
    BF_HIP_TRY(hipStreamWaitEvent(s->stream, s->evTmp, 0));
    return BF_OK;
}
int endExclusive(bf_scene* s) {
    if (!s->overlap) return BF_OK;
    BF_HIP_TRY(hipEventRecord(s->evBarrier, s->stream));
    s->barrierPending = true;
    return BF_OK;
}
int syncAll(bf_scene* s) {
    if (s->prep) BF_HIP_TRY(hipStreamSynchronize(s->prep));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    return BF_OK;
}

int launchCompactify(bf_scene* s) {                              // compactifyHashEntries :355-391 (main stream, current buffer)
    const Frame f = makeFrame(s);
    hipLaunchKernelGGL(k_alloc_snapshot, dim3(1), dim3(1), 0, s->stream, s->d);
    hipLaunchKernelGGL(k_compact_append<0>, dim3(s->gridCompact), dim3(256), 0, s->stream, s->d, f, f);
    BF_HIP_TRY(hipGetLastError());
    s->compactStale = false; s->gcMask = 0;
    return BF_OK;
}
int refreshStaleList(bf_scene* s) {                              // a fused re-integration left a union list behind
    if (!s->compactStale) return BF_OK;
    BF_TRY_RC(beginExclusive(s));
    BF_TRY_RC(launchCompactify(s));
    return endExclusive(s);
}

// alloc :328-352 (dv: the operator's view - its list buffer and snapshot slot) and, in the same launch as the placement, the operator's block list:
// frustum list of f (fused == false) or union list of f and fo
int launchAllocOn(bf_scene* s, hipStream_t st, const Dev& dv, const Frame& f, const Frame& fo, bool fused, const float* d_depth, TexelOut tx) {
    const uint32_t tiles = div_up(s->cam.m_imageWidth, 8) * div_up(s->cam.m_imageHeight, 8);
    if (s->allocComm) {
        // the march divided over the ranks: rank r marches the tiles [r T / G, (r + 1) T / G) and collects the distinct in-frustum keys it
        // meets; the records {count, keys} of all ranks are all-gathered on this stream (RCCL: ring over xGMI; nothing waits for it but the ingest behind
        // it - the allocation stream runs up to three operators ahead of the voxel updates); every rank queues the keys of every list that it owns
        uint32_t world = 1, rank = 0;
        BF_TRY_RC(bf_comm_world(s->allocComm, &world, &rank));
        const uint64_t rec = 8 + 8ull * s->allocCap;
        uint32_t* cnt = reinterpret_cast<uint32_t*>(s->d_allocSend);
        Collect c;
        c.keys = reinterpret_cast<unsigned long long*>(s->d_allocSend + 8); c.slots = s->d_allocSlots; c.count = cnt; c.capacity = s->allocCap;
        c.tile0 = (uint32_t)((uint64_t)tiles * rank / world); c.tile1 = (uint32_t)((uint64_t)tiles * (rank + 1) / world);
        BF_HIP_TRY(hipMemsetAsync(cnt, 0, 8, st));
        BF_HIP_TRY(hipMemsetAsync(dv.compactCount, 0, 8, st));      // the operator's list counters (the local march zeroes them itself)
        if (c.tile1 > c.tile0) hipLaunchKernelGGL(k_alloc_candidates<true>, dim3(div_up(c.tile1 - c.tile0, 4)), dim3(256), 0, st, dv, f, d_depth, c, TexelOut{nullptr, nullptr});
        hipLaunchKernelGGL(k_collect_release, dim3(std::min<uint32_t>(div_up(s->allocCap, 256u), 2048u)), dim3(256), 0, st, dv, c);
        BF_TRY_RC(bf_comm_all_gather(s->allocComm, s->d_allocSend, s->d_allocRecv, rec, st));
        for (uint32_t r = 0; r < world; ++r) {
            const uint8_t* base = s->d_allocRecv + rec * r;
            hipLaunchKernelGGL(k_alloc_ingest, dim3(std::min<uint32_t>(div_up(s->allocCap, 256u), 1024u)), dim3(256), 0, st, dv, f,
                               reinterpret_cast<const unsigned long long*>(base + 8), reinterpret_cast<const uint32_t*>(base), s->allocCap);
        }
    } else
    hipLaunchKernelGGL(k_alloc_candidates<false>, dim3(div_up(tiles, 4)), dim3(256), 0, st, dv, f, d_depth, Collect{}, tx);
    if (fused) hipLaunchKernelGGL(k_alloc_place<2>, dim3(PLACE_WGS), dim3(256), 0, st, dv, f, fo);
    else hipLaunchKernelGGL(k_alloc_place<0>, dim3(PLACE_WGS), dim3(256), 0, st, dv, f, f);
    return BF_OK;
}

// What the allocation stream has to wait for before it may touch the table or read a frame: the event the caller ordered the next operator behind
// (bf_scene_wait_event: the frame's ingest) and the last exclusive section of the main stream (garbage collection, compactify).  Both are consumed here:
// whoever touches the prep stream first - runOperator, or bf_scene_alloc_collect / _ingest / _place when the caller allocates itself - waits, and
// everything issued on that stream afterwards is ordered behind it.
int prepWaits(bf_scene* s, hipStream_t ps) {
    if (s->pendingEv) { BF_HIP_TRY(hipStreamWaitEvent(ps, s->pendingEv, 0)); s->pendingEv = nullptr; }
    if (s->overlap && s->barrierPending) { BF_HIP_TRY(hipStreamWaitEvent(ps, s->evBarrier, 0)); s->barrierPending = false; }
    return BF_OK;
}

// One operator = preparation (allocation + block list) -> voxel update.  kind 0 integrate(f), 1 de-integrate(f), 2 fused: de-integrate(fo) +
// integrate(f).  With overlap enabled the preparation goes to the prep stream and only the update to the main stream.  The preparation of an operator that
// allocates is two launches - the march (which also writes the frame's texels for the fast contract) and the placement, which builds the list on the way;
// an operator that does not allocate takes its snapshot and filters the allocated-block list.
int runOperator(bf_scene* s, int kind, const Frame& f, const Frame& fo, const bf_depth_camera_data* data) {
    const int b = s->overlap ? (s->cur + 1) % s->NB : s->cur;
    hipStream_t ps = s->overlap ? s->prep : s->stream;
    const bool useTexel = s->arith == BF_TSDF_ARITH_FAST && data->d_colorData != nullptr;
    BF_TRY_RC(prepWaits(s, ps));                   // incl. the frame's ingest (bf_scene_wait_event)
    const Dev dv = devBuf(s, b);
    const uint2* opTexels = s->frameTexels;        // the caller's per-frame texel image (consumed by this operator), or the one made here
    s->frameTexels = nullptr;
    const size_t npx = (size_t)s->cam.m_imageWidth * s->cam.m_imageHeight;
    if (useTexel && !opTexels && s->texelPixels < npx) {
        BF_TRY_RC(syncAll(s));
        for (int k = 0; k < bf_scene::NBMAX; ++k) { if (s->texel[k]) (void)hipFree(s->texel[k]); s->texel[k] = nullptr; BF_HIP_TRY(BF_MALLOC((void**)&s->texel[k], npx * sizeof(uint2))); }
        s->texelPixels = npx;
    }
    // The preparation writes the operator's snapshot and list into buffer b (and, fast contract, the frame's texels into texel buffer b): not before the update
    // that used buffer b NB operators ago has finished - which also keeps the preparation at most NB operators ahead of the updates
    if (s->overlap && s->updRecorded[b]) BF_HIP_TRY(hipStreamWaitEvent(ps, s->evUpd[b], 0));
    const bool marches = kind != 1 && !s->externalAlloc;                       // de-integration neither allocates nor frees
    const bool texelsFromMarch = marches && !s->allocComm && useTexel && !opTexels;      // (the divided march covers a band of the image only)
    if (useTexel && !opTexels && !texelsFromMarch) {
        hipLaunchKernelGGL(k_interleave, dim3(std::min<uint32_t>(div_up((uint32_t)npx, 256u), 2048u)), dim3(256), 0, ps, data->d_depthData, reinterpret_cast<const uint32_t*>(data->d_colorData), s->texel[b], (uint32_t)npx);
        opTexels = s->texel[b];
    }
    if (marches) {
        TexelOut tx{nullptr, nullptr};
        if (texelsFromMarch) { tx.color = reinterpret_cast<const uint32_t*>(data->d_colorData); tx.texel = s->texel[b]; opTexels = s->texel[b]; }
        BF_TRY_RC(launchAllocOn(s, ps, dv, f, fo, kind == 2, data->d_depthData, tx));
    } else {
        hipLaunchKernelGGL(k_alloc_snapshot, dim3(1), dim3(1), 0, ps, dv);
        if (kind == 2) hipLaunchKernelGGL(k_compact_append<2>, dim3(s->gridCompact), dim3(256), 0, ps, dv, f, fo);
        else hipLaunchKernelGGL(k_compact_append<0>, dim3(s->gridCompact), dim3(256), 0, ps, dv, f, f);
    }
    if (s->overlap) {
        BF_HIP_TRY(hipEventRecord(s->evPrep[b], ps));
        BF_HIP_TRY(hipStreamWaitEvent(s->stream, s->evPrep[b], 0));
    }
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (s->timing) {
        if (s->eventsUsed == s->events.size()) {
            hipEvent_t e0, e1;
            BF_HIP_TRY(hipEventCreate(&e0));
            BF_HIP_TRY(hipEventCreate(&e1));
            s->events.push_back({e0, e1});
        }
        ev = &s->events[s->eventsUsed++];
        s->opsTimed += kind == 2 ? 2 : 1;
        s->imagesTimed += 1;
        BF_HIP_TRY(hipEventRecord(ev->first, s->stream));
    }
    const uchar4* color = reinterpret_cast<const uchar4*>(data->d_colorData);
    const int acc = s->timing ? 1 : 0;
    if (s->arith == BF_TSDF_ARITH_FAST) {
        const ApxCam ac = makeApxCam(f);
        const ApxPose pin = makeApxPose(f), pde = makeApxPose(kind == 2 ? fo : f);
        const int hasColor = useTexel ? 1 : 0;
        if (kind == 0) launchApx<0>(s, s->gridUpdateColPlain, dv, ac, pin, pde, opTexels, hasColor, acc);
        else if (kind == 1) launchApx<1>(s, s->gridUpdateColPlain, dv, ac, pin, pde, opTexels, hasColor, acc);
        else launchApx<2>(s, s->gridUpdateCol, dv, ac, pin, pde, opTexels, hasColor, acc);
    } else {
        const UpdCam uc = makeUpdCam(f);
        const UpdPose pin = makeUpdPose(f), pde = makeUpdPose(kind == 2 ? fo : f);
        const int fe = s->forceExactDiv ? 1 : 0;
        if (kind == 0) hipLaunchKernelGGL(k_update_col<0>, dim3(s->gridUpdateColPlain), dim3(256), 0, s->stream, dv, uc, pin, pde, data->d_depthData, color, acc, fe);
        else if (kind == 1) hipLaunchKernelGGL(k_update_col<1>, dim3(s->gridUpdateColPlain), dim3(256), 0, s->stream, dv, uc, pin, pde, data->d_depthData, color, acc, fe);
        else hipLaunchKernelGGL(k_update_col<2>, dim3(s->gridUpdateCol), dim3(256), 0, s->stream, dv, uc, pin, pde, data->d_depthData, color, acc, fe);
    }
    if (ev) BF_HIP_TRY(hipEventRecord(ev->second, s->stream));
    if (s->overlap) { BF_HIP_TRY(hipEventRecord(s->evUpd[b], s->stream)); s->updRecorded[b] = true; }
    BF_HIP_TRY(hipGetLastError());
    useBuf(s, b);
    s->compactStale = kind == 2;
    s->gcMask = kind == 2 ? 1u : 0u;      // bit 0 of a union list's flags: the block lies in the frustum of the new pose (keepRec<2>)
    return BF_OK;
}

// ---- batched operators (tsdf_batch.h)
int ensureBatch(bf_scene* s) {
    if (s->batchReady) return BF_OK;
    const size_t N = s->params.m_numSDFBlocks;
    uint32_t ds = 1u << 18;
    while (ds < (1u << 22) && (size_t)ds < 4u * N) ds <<= 1;
    BatchDev& bd = s->bd;
    int rc = BF_OK;
#define A(ptr, cnt) if (rc == BF_OK) rc = devAlloc(s, &(ptr), (cnt))
    A(bd.set, (size_t)ds);
    A(bd.opMask, (size_t)ds);
    A(bd.candList, (size_t)ds / 2);
    A(bd.candCount, 1);
    A(bd.bins, (size_t)BMAX * NBINS * BINCAP);
    A(bd.binCount, (size_t)BMAX * NBINS);
    A(bd.bucketCnt, (size_t)s->params.m_hashNumBuckets);
    A(bd.flags, 4);
#undef A
    if (rc != BF_OK) return rc;
    bd.setMask = ds - 1; bd.candCap = ds / 2;
    hipLaunchKernelGGL(k_batch_reset, dim3(2048), dim3(256), 0, s->overlap ? s->prep : s->stream, bd, s->params.m_hashNumBuckets);      // (the stream the first march follows on)
    BF_HIP_TRY(hipGetLastError());
    s->batchReady = true;
    return BF_OK;
}

constexpr uint32_t VERIFY_CAP = 1u << 16;
int verifyBuffers(bf_scene* s) {
    if (s->vlog) return BF_OK;
    for (int q = 0; q < 2; ++q) BF_HIP_TRY(BF_MALLOC((void**)&s->vshadow[q], (size_t)s->params.m_numSDFBlocks * VOX * sizeof(bf_voxel)));
    BF_HIP_TRY(hipHostMalloc((void**)&s->vlog, sizeof(VerifyLog) + (size_t)VERIFY_CAP * sizeof(VerifyRec), hipHostMallocCoherent));
    memset(s->vlog, 0, sizeof(VerifyLog));
    s->vlog->cap = VERIFY_CAP;
    return BF_OK;
}
void verifyDump(bf_scene* s) {          // (streams drained by the caller) appends {count, cap, batches, blocks} + the records to the file
    if (!s->vlog) return;
    if (FILE* f = fopen(s->verifyPath, "ab")) {
        const uint32_t n = std::min(s->vlog->count, s->vlog->cap);
        fwrite(s->vlog, 16, 1, f);
        fwrite(s->vlog->rec, sizeof(VerifyRec), n, f);
        fclose(f);
    }
}

// A batch of operators in the serial order ops[0], ops[1], ...: one march, one binning, one placement + union list (preparation stream), one voxel update.
int runBatch(bf_scene* s, const bf_scene_batch_op* ops, uint32_t n) {
    BF_TRY_RC(ensureBatch(s));
    const int b = s->overlap ? (s->cur + 1) % s->NB : s->cur;
    hipStream_t ps = s->overlap ? s->prep : s->stream;
    const Dev dv = devBuf(s, b);
    const bool fast = s->arith == BF_TSDF_ARITH_FAST;
    const size_t npx = (size_t)s->cam.m_imageWidth * s->cam.m_imageHeight;
    if (fast && s->btexelPixels < npx) {
        BF_TRY_RC(syncAll(s));
        s->btexelPixels = 0;                    // (a failed allocation below leaves the sets to be rebuilt by the next call)
        for (int q = 0; q < bf_scene::NBMAX; ++q)
            for (uint32_t k = 0; k < BMAX; ++k) {
                if (s->btexel[q][k]) (void)hipFree(s->btexel[q][k]);
                s->btexel[q][k] = nullptr;
                if (q < s->NB) BF_HIP_TRY(BF_MALLOC((void**)&s->btexel[q][k], npx * sizeof(uint2)));
            }
        s->btexelPixels = npx;
    }
    // The march runs on the preparation stream like everything else of the preparation (on the main stream behind the previous batch's update it measured the same
    // frame rate, gpurun r05c / r05f).
    hipStream_t ms = ps;
    for (uint32_t k = 0; k < n; ++k) if (ops[k].wait_event) BF_HIP_TRY(hipStreamWaitEvent(ms, (hipEvent_t)ops[k].wait_event, 0));
    if (s->pendingEv) { BF_HIP_TRY(hipStreamWaitEvent(ms, s->pendingEv, 0)); s->pendingEv = nullptr; }
    if (s->overlap && s->updRecorded[b]) BF_HIP_TRY(hipStreamWaitEvent(ps, s->evUpd[b], 0));      // the update that read list buffer b and its texel set NB batches ago
    s->frameTexels = nullptr;
    // per-operator frames: integration pose (kinds 0, 2), de-integration pose (kinds 1, 2)
    Frame fin[BMAX], fde[BMAX];
    BatchCommon bc;
    BatchMarchArgs ma;
    BatchFrusta fr;
    memset(&ma, 0, sizeof ma); memset(&fr, 0, sizeof fr);
    uint32_t opsCount = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const bf_scene_batch_op& o = ops[k];
        setLastRigidTransform(s, o.T0);
        const Frame f0 = makeFrame(s);
        Frame f1 = f0;
        if (o.kind == 2) { setLastRigidTransform(s, o.T1); f1 = makeFrame(s); }
        fin[k] = o.kind == 2 ? f1 : f0;          // kind 0: integrate at T0; kind 2: integrate at T1
        fde[k] = f0;                             // kind 1: de-integrate at T0; kind 2: de-integrate at T0
        fr.TinvIn[k] = fin[k].Tinv; fr.TinvDe[k] = fde[k].Tinv;
        fr.bits[k] = o.kind == 0 ? 1u : o.kind == 1 ? 2u : 3u;
        BatchMarchOp& m = ma.op[k];
        m.T = fin[k].T; m.Tinv = fin[k].Tinv;
        m.depth = o.data.d_depthData; m.color = reinterpret_cast<const uint32_t*>(o.data.d_colorData);
        m.marches = (o.kind != 1 && !s->externalAlloc) ? 1u : 0u;      // a de-integration neither allocates nor frees
        m.texel = (fast && o.data.d_colorData && !o.d_texels) ? s->btexel[b][k] : nullptr;
        opsCount += o.kind == 2 ? 2u : 1u;
    }
    const uint32_t tiles = div_up(s->cam.m_imageWidth, 8) * div_up(s->cam.m_imageHeight, 8);
    uint32_t world = 1, rank = 0;
    if (s->allocComm) BF_TRY_RC(bf_comm_world(s->allocComm, &world, &rank));
    uint2* texelOf[BMAX];
    for (uint32_t k = 0; k < n; ++k) {
        texelOf[k] = ma.op[k].texel;
        if (s->allocComm && texelOf[k]) {          // the divided march covers a band of the image only: the operator's texel image is made by its own launch
            hipLaunchKernelGGL(k_interleave, dim3(std::min<uint32_t>(div_up((uint32_t)npx, 256u), 2048u)), dim3(256), 0, ms, ops[k].data.d_depthData, reinterpret_cast<const uint32_t*>(ops[k].data.d_colorData), texelOf[k], (uint32_t)npx);
            ma.op[k].texel = nullptr;
        }
    }
    const Frame fl = fin[n - 1];                // (the last pose set above is the last operator's: what a compactify / garbage collection behind the batch refers to)
    bc.cam = s->cam;
    bc.numBuckets = fl.numBuckets; bc.maxChain = fl.maxChain; bc.numSDFBlocks = fl.numSDFBlocks;
    bc.voxelSize = fl.voxelSize; bc.maxIntegrationDistance = fl.maxIntegrationDistance; bc.truncScale = fl.truncScale; bc.truncation = fl.truncation;
    bc.shardLo = fl.shardLo; bc.shardHi = fl.shardHi; bc.nOps = n;
    bc.tile0 = (uint32_t)((uint64_t)tiles * rank / world); bc.tile1 = (uint32_t)((uint64_t)tiles * (rank + 1) / world);
    if (bc.tile1 > bc.tile0) hipLaunchKernelGGL(k_batch_march, dim3(div_up(bc.tile1 - bc.tile0, 4), n), dim3(256), 0, ms, dv, s->bd, bc, ma);
    if (s->allocComm) {          // one all-gather of the batch's {key, operator mask} lists (tsdf_batch.h)
        const uint64_t brec = 8 + sizeof(BatchRec) * (uint64_t)s->allocCap;
        hipLaunchKernelGGL(k_batch_pack, dim3(256), dim3(256), 0, ms, dv, s->bd, reinterpret_cast<uint32_t*>(s->d_batchSend), reinterpret_cast<BatchRec*>(s->d_batchSend + 8), s->allocCap);
        hipLaunchKernelGGL(k_batch_pack_finish, dim3(1), dim3(1), 0, ms, s->bd);
        BF_TRY_RC(bf_comm_all_gather(s->allocComm, s->d_batchSend, s->d_batchRecv, brec, ms));
        for (uint32_t r = 0; r < world; ++r) {
            const uint8_t* base = s->d_batchRecv + brec * r;
            hipLaunchKernelGGL(k_batch_ingest, dim3(256), dim3(256), 0, ms, dv, s->bd, reinterpret_cast<const uint32_t*>(base), reinterpret_cast<const BatchRec*>(base + 8), s->allocCap);
        }
    }
    // the table look-ups wait for whatever frees table entries (the last garbage collection); the march above does not
    if (s->overlap && s->barrierPending) { BF_HIP_TRY(hipStreamWaitEvent(ps, s->evBarrier, 0)); s->barrierPending = false; }
    hipLaunchKernelGGL(k_batch_bin, dim3(1024), dim3(256), 0, ps, dv, s->bd, bc);
    hipLaunchKernelGGL(k_batch_place, dim3(PLACE_WGS), dim3(256), 0, ps, dv, s->bd, bc, fr);
    if (s->overlap) {
        BF_HIP_TRY(hipEventRecord(s->evPrep[b], ps));
        BF_HIP_TRY(hipStreamWaitEvent(s->stream, s->evPrep[b], 0));
    }
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (s->timing) {
        if (s->eventsUsed == s->events.size()) {
            hipEvent_t e0, e1;
            BF_HIP_TRY(hipEventCreate(&e0));
            BF_HIP_TRY(hipEventCreate(&e1));
            s->events.push_back({e0, e1});
        }
        ev = &s->events[s->eventsUsed++];
        s->opsTimed += opsCount;
        s->imagesTimed += n;
        BF_HIP_TRY(hipEventRecord(ev->first, s->stream));
    }
    const int acc = s->timing ? 1 : 0;
    if (fast) {
        BatchUpdApxArgs ua;
        memset(&ua, 0, sizeof ua);
        ua.nOps = n;
        for (uint32_t k = 0; k < n; ++k) {
            ua.op[k].in = makeApxPose(fin[k]); ua.op[k].de = makeApxPose(fde[k]);
            ua.op[k].tex = ops[k].d_texels ? reinterpret_cast<const uint2*>(ops[k].d_texels) : texelOf[k];
            if (ops[k].data.d_colorData) ua.liveMask |= 3u << (2u * k);      // CUDASceneRepHashSDF.cu:441-448: no colour data, no update
        }
        const ApxCam ac = makeApxCam(fl);
        auto update = [&](const Dev& dd, int accumulate) {
            if (s->cvtRne) hipLaunchKernelGGL((k_update_batch_apx<true>), dim3(s->gridUpdateCol), dim3(256), s->updateLds, s->stream, dd, ac, ua, accumulate);
            else hipLaunchKernelGGL((k_update_batch_apx<false>), dim3(s->gridUpdateCol), dim3(256), s->updateLds, s->stream, dd, ac, ua, accumulate);
        };
        if (s->verifyPath) {
            BF_TRY_RC(verifyBuffers(s));
            hipLaunchKernelGGL(k_verify_copy, dim3(s->gridUpdateCol), dim3(256), 0, s->stream, dv, s->vshadow[0], s->vshadow[1]);
        }
        update(dv, acc);
        if (s->verifyPath) {
            for (int q = 0; q < 2; ++q) { Dev dq = dv; dq.vox = s->vshadow[q]; update(dq, 0); }
            hipLaunchKernelGGL(k_verify_compare, dim3(s->gridUpdateCol), dim3(256), 0, s->stream, dv, s->vshadow[0], s->vshadow[1], s->vlog, s->vseq++, n, s->gridUpdateCol * 4u);
        }
    } else {
        BatchUpdColArgs ua;
        memset(&ua, 0, sizeof ua);
        ua.nOps = n;
        for (uint32_t k = 0; k < n; ++k) {
            ua.op[k].in = makeUpdPose(fin[k]); ua.op[k].de = makeUpdPose(fde[k]);
            ua.op[k].depth = ops[k].data.d_depthData; ua.op[k].color = reinterpret_cast<const uchar4*>(ops[k].data.d_colorData);
        }
        hipLaunchKernelGGL(k_update_batch_col, dim3(s->gridUpdateCol), dim3(256), 0, s->stream, dv, makeUpdCam(fl), ua, acc, s->forceExactDiv ? 1 : 0);
    }
    if (ev) BF_HIP_TRY(hipEventRecord(ev->second, s->stream));
    if (s->overlap) { BF_HIP_TRY(hipEventRecord(s->evUpd[b], s->stream)); s->updRecorded[b] = true; }
    BF_HIP_TRY(hipGetLastError());
    useBuf(s, b);
    s->compactStale = true;
    // the frustum list of the LAST pose inside the union list: the blocks with the last operator's bit (integration pose, or the pose of a de-integration)
    s->gcMask = (ops[n - 1].kind == 1 ? 2u : 1u) << (2u * (n - 1));
    for (uint32_t k = 0; k < n; ++k) s->numIntegrated += ops[k].kind == 0 ? 1u : ops[k].kind == 1 ? (uint32_t)-1 : 0u;
    return BF_OK;
}

int checkCall(bf_scene* s, const float* T, const bf_depth_camera_data* data, const bf_depth_camera_params* cam,
              const uint32_t* d_bitMask) {
    BF_REQUIRE(s && T && data && cam, "null argument");
    BF_REQUIRE(data->d_depthData, "d_depthData is null");
    BF_REQUIRE(d_bitMask == nullptr, "chunk streaming (d_bitMask) is not supported: it is disabled for BundleFusion");
    BF_REQUIRE(cam->m_imageWidth > 0 && cam->m_imageHeight > 0, "empty image");
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_scene_create(const bf_hash_params* p, bf_scene** out) {
    BF_REQUIRE(p && out, "null argument");
    BF_REQUIRE(p->m_hashNumBuckets > 0 && p->m_numSDFBlocks > 0, "empty hash / heap");
    BF_REQUIRE(p->m_hashBucketSize == BF_HASH_BUCKET_SIZE, "m_hashBucketSize must be 4 (HASH_BUCKET_SIZE)");
    BF_REQUIRE(p->m_SDFBlockSize == BF_SDF_BLOCK_SIZE, "m_SDFBlockSize must be 8 (SDF_BLOCK_SIZE)");
    BF_REQUIRE((uint64_t)p->m_hashNumBuckets * BF_HASH_BUCKET_SIZE < 0x7FFFFFFFull, "hash too large for 32-bit slot indices");
    BF_REQUIRE((uint64_t)p->m_numSDFBlocks * VOX < 0x7FFFFFFFull, "too many SDF blocks for the reference's int ptr");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device visible"); return BF_ERR_NO_DEVICE; }
    bf_scene* s = new bf_scene();
    s->params = *p;
    const size_t numEntries = (size_t)p->m_hashNumBuckets * BF_HASH_BUCKET_SIZE;
    const size_t N = p->m_numSDFBlocks;
    uint32_t ds = 1u << 16;
    while (ds < 2u * NBINS * BINCAP && ds < 4u * N) ds <<= 1;
    s->dedupeSize = ds;
    int rc = BF_OK;
#define A(ptr, cnt) if (rc == BF_OK) rc = devAlloc(s, &(ptr), (cnt))
    A(s->d.hash, numEntries);
    A(s->d.heap, N);
    A(s->d.heapCounter, 1);
    A(s->d.vox, N * VOX);
    for (int b = 0; b < bf_scene::NBMAX; ++b) { A(s->cbuf[b], N); A(s->csrc[b], N); A(s->ccnt[b], 4); }      // ccnt: [0] list length, [1] operator blocks of a union list (entries in the new frustum + entries in the old one)
    A(s->d.occSum, 3);
    A(s->d.allocList, N);
    A(s->d.allocListAlt, N);
    A(s->d.allocCount, 1);
    A(s->d.dedupe, (size_t)ds);
    A(s->d.bins, (size_t)NBINS * BINCAP);
    A(s->d.binCount, NBINS);
    A(s->d.overflow, OVCAP);
    A(s->d.overflowCount, 1);
    A(s->d.stuckSlots, OVCAP);
    A(s->d.tileCounts, N / TILE + 4);
    A(s->d.stats, ST_COUNT);
    A(s->d_hashDecision, 4);
#undef A
    if (rc != BF_OK) { bf_scene_destroy(s); return rc; }
    s->d.dedupeMask = ds - 1;
    useBuf(s, 0);
    {   // allocation / compaction are short and sit on the critical path of the next voxel update: give them queue priority
        int least = 0, greatest = 0;
        BF_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        BF_HIP_TRY(hipStreamCreateWithPriority(&s->prep, hipStreamNonBlocking, greatest));
    }
    for (int b = 0; b < bf_scene::NBMAX; ++b)
        for (hipEvent_t* e : {&s->evPrep[b], &s->evUpd[b]}) BF_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (hipEvent_t* e : {&s->evBarrier, &s->evTmp})
        BF_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    s->gridCompact = std::min<uint32_t>(std::max<uint32_t>(div_up((uint32_t)N, TILE), 1u), 2048u);
    s->gridUpdateCol = s->gridUpdateColPlain = 8192;      // see k_update_col
    if (const char* e = getenv("BF_TSDF_EXACT_DIV")) s->forceExactDiv = atoi(e) != 0;
    if (const char* e = getenv("BF_DEBUG_VERIFY_BATCH")) s->verifyPath = e;
    if (const char* e = getenv("BF_DEBUG_UPDATE_LDS")) s->updateLds = (uint32_t)atoi(e);
    *out = s;
    int rcReset = bf_scene_reset(s);
    if (rcReset != BF_OK) return rcReset;
    int arith = BF_TSDF_ARITH_FAST;
    if (const char* e = getenv("BF_TSDF_ARITH")) {
        if (strcasecmp(e, "exact") == 0) arith = BF_TSDF_ARITH_EXACT;
        else if (strcasecmp(e, "fast") == 0) arith = BF_TSDF_ARITH_FAST;
        else { set_error("BF_TSDF_ARITH must be 'fast' or 'exact' (got '%s')", e); bf_scene_destroy(s); *out = nullptr; return BF_ERR_INVALID_ARG; }
    }
    return bf_scene_set_arith(s, arith);
}

// ---- multi-GPU allocation: collect on a band of pixel tiles, exchange, ingest (see Collect above)
int bf_scene_set_external_alloc(bf_scene* s, int enable) {
    BF_REQUIRE(s, "null scene");
    BF_TRY_RC(syncAll(s));
    s->externalAlloc = enable != 0;
    return BF_OK;
}

// March the pixel tiles [part * T / parts, (part + 1) * T / parts) of the frame (T = number of 8x8 tiles, row-major: a band of image rows)
// at pose camToWorld and write the distinct in-frustum block keys to d_keys[capacity] (packed 64-bit keys, an opaque format for
// bf_scene_alloc_ingest), their number to d_count[0].  d_slots[capacity] is scratch of the same length.  Runs on the allocation stream
// (the scene's prep stream when operators are software-pipelined); the caller orders its exchange after bf_scene_alloc_sync.
int bf_scene_alloc_collect(bf_scene* s, const float camToWorld[16], const bf_depth_camera_data* data, const bf_depth_camera_params* cam, uint32_t part, uint32_t parts,
                           uint64_t* d_keys, uint32_t* d_slots, uint32_t* d_count, uint32_t capacity) {
    BF_REQUIRE(s && camToWorld && data && cam && d_keys && d_slots && d_count && data->d_depthData, "null argument");
    BF_REQUIRE(parts >= 1 && part < parts && capacity > 0, "bad partition");
    s->cam = *cam; s->haveCam = true;
    setLastRigidTransform(s, camToWorld);
    const Frame f = makeFrame(s);
    hipStream_t st = s->overlap ? s->prep : s->stream;
    BF_TRY_RC(prepWaits(s, st));
    const uint32_t tiles = div_up(cam->m_imageWidth, 8) * div_up(cam->m_imageHeight, 8);
    Collect c;
    c.keys = reinterpret_cast<unsigned long long*>(d_keys); c.slots = d_slots; c.count = d_count; c.capacity = capacity;
    c.tile0 = (uint32_t)((uint64_t)tiles * part / parts); c.tile1 = (uint32_t)((uint64_t)tiles * (part + 1) / parts);
    BF_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), st));
    if (c.tile1 > c.tile0) hipLaunchKernelGGL(k_alloc_candidates<true>, dim3(div_up(c.tile1 - c.tile0, 4)), dim3(256), 0, st, s->d, f, data->d_depthData, c, TexelOut{nullptr, nullptr});
    hipLaunchKernelGGL(k_collect_release, dim3(std::min<uint32_t>(div_up(capacity, 256u), 2048u)), dim3(256), 0, st, s->d, c);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

// Queue every key of one rank's list (d_keys / d_count as written by bf_scene_alloc_collect on that rank) that this volume owns and does
// not hold yet; bf_scene_alloc_place then inserts what was queued.  Call once per rank list, then place once, then the operator.
int bf_scene_alloc_ingest(bf_scene* s, const uint64_t* d_keys, const uint32_t* d_count, uint32_t capacity) {
    BF_REQUIRE(s && d_keys && d_count, "null argument");
    const Frame f = makeFrame(s);
    hipStream_t st = s->overlap ? s->prep : s->stream;
    BF_TRY_RC(prepWaits(s, st));
    hipLaunchKernelGGL(k_alloc_ingest, dim3(std::min<uint32_t>(div_up(capacity, 256u), 1024u)), dim3(256), 0, st, s->d, f, reinterpret_cast<const unsigned long long*>(d_keys), d_count, capacity);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}
int bf_scene_alloc_place(bf_scene* s) {
    BF_REQUIRE(s, "null scene");
    const Frame f = makeFrame(s);
    hipStream_t st = s->overlap ? s->prep : s->stream;
    BF_TRY_RC(prepWaits(s, st));
    hipLaunchKernelGGL(k_alloc_place<-1>, dim3(PLACE_WGS), dim3(256), 0, st, s->d, f, f);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}
int bf_scene_alloc_sync(bf_scene* s) {          // the allocation stream has drained (collect lists are complete, ingest has read its input)
    BF_REQUIRE(s, "null scene");
    BF_HIP_TRY(hipStreamSynchronize(s->overlap ? s->prep : s->stream));
    return BF_OK;
}

// The operators' own allocation over a communicator (see bf_scene::allocComm).  The volume must be sharded consistently (bf_scene_set_shard(rank, world) of the
// same communicator); every rank must run the same operator sequence (the collective is issued inside integrate / re-integrate).  capacity_keys: upper bound
// of the distinct in-frustum blocks one rank's band of pixel tiles meets in one operator (exceeding it raises the scene's error flag, never drops silently).
// comm == null: back to the local march.
int bf_scene_set_alloc_comm(bf_scene* s, bf_comm* comm, uint32_t capacity_keys) {
    BF_REQUIRE(s, "null scene");
    BF_TRY_RC(syncAll(s));
    if (s->d_allocSend) { (void)hipFree(s->d_allocSend); s->d_allocSend = nullptr; }
    if (s->d_allocRecv) { (void)hipFree(s->d_allocRecv); s->d_allocRecv = nullptr; }
    if (s->d_allocSlots) { (void)hipFree(s->d_allocSlots); s->d_allocSlots = nullptr; }
    if (s->d_batchSend) { (void)hipFree(s->d_batchSend); s->d_batchSend = nullptr; }
    if (s->d_batchRecv) { (void)hipFree(s->d_batchRecv); s->d_batchRecv = nullptr; }
    s->allocComm = nullptr; s->allocCap = 0;
    if (!comm) return BF_OK;
    BF_REQUIRE(capacity_keys >= 64, "capacity_keys too small");
    uint32_t world = 1, rank = 0;
    BF_TRY_RC(bf_comm_world(comm, &world, &rank));
    const uint64_t rec = 8 + 8ull * capacity_keys;
    BF_HIP_TRY(BF_MALLOC((void**)&s->d_allocSend, rec));
    BF_HIP_TRY(BF_MALLOC((void**)&s->d_allocRecv, rec * world));
    BF_HIP_TRY(BF_MALLOC((void**)&s->d_allocSlots, sizeof(uint32_t) * capacity_keys));
    const uint64_t brec = 8 + sizeof(BatchRec) * (uint64_t)capacity_keys;
    BF_HIP_TRY(BF_MALLOC((void**)&s->d_batchSend, brec));
    BF_HIP_TRY(BF_MALLOC((void**)&s->d_batchRecv, brec * world));
    s->allocComm = comm; s->allocCap = capacity_keys;
    return BF_OK;
}

// Arithmetic contract of the voxel update (see k_update_apx): BF_TSDF_ARITH_EXACT evaluates CUDASceneRepHashSDF.cu:425-516 IEEE
// operation by operation (bit-comparable with a host build of the reference); BF_TSDF_ARITH_FAST is the contract of the reference's
// own GPU build (-use_fast_math, FriedLiver.vcxproj:124): approximate division, FMA contraction.  Allocation, lists, weights and
// garbage collection are unaffected.
int bf_scene_set_arith(bf_scene* s, int mode) {
    BF_REQUIRE(s, "null scene");
    BF_REQUIRE(mode == BF_TSDF_ARITH_EXACT || mode == BF_TSDF_ARITH_FAST, "unknown arithmetic contract");
    if (mode == BF_TSDF_ARITH_FAST) BF_TRY_RC(probeCvt(s));
    s->arith = mode;
    return BF_OK;
}

int bf_scene_get_arith(bf_scene* s, int* mode) {
    BF_REQUIRE(s && mode, "null argument");
    *mode = s->arith;
    return BF_OK;
}

int bf_scene_destroy(bf_scene* s) {
    if (!s) return BF_OK;
    (void)syncAll(s);
    for (void* q : s->allocations) hipFree(q);
    for (uint2* t : s->texel) if (t) hipFree(t);
    for (auto& set : s->btexel) for (uint2* t : set) if (t) hipFree(t);
    verifyDump(s);
    for (bf_voxel* v : s->vshadow) if (v) hipFree(v);
    if (s->vlog) hipHostFree(s->vlog);
    if (s->d_allocSend) hipFree(s->d_allocSend);
    if (s->d_allocRecv) hipFree(s->d_allocRecv);
    if (s->d_allocSlots) hipFree(s->d_allocSlots);
    if (s->d_batchSend) hipFree(s->d_batchSend);
    if (s->d_batchRecv) hipFree(s->d_batchRecv);
    for (auto& e : s->events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (int b = 0; b < bf_scene::NBMAX; ++b) for (hipEvent_t e : {s->evPrep[b], s->evUpd[b]}) if (e) hipEventDestroy(e);
    for (hipEvent_t e : {s->evBarrier, s->evTmp}) if (e) hipEventDestroy(e);
    if (s->prep) hipStreamDestroy(s->prep);
    delete s;
    return BF_OK;
}

int bf_scene_set_stream(bf_scene* s, void* hip_stream) {
    BF_REQUIRE(s, "null scene");
    BF_TRY_RC(syncAll(s));
    s->stream = (hipStream_t)hip_stream;
    return BF_OK;
}

int bf_scene_set_overlap(bf_scene* s, int enable) {
    BF_REQUIRE(s, "null scene");
    BF_TRY_RC(syncAll(s));
    s->overlap = enable != 0 && !getenv("BF_DEBUG_NO_OVERLAP");      // (diagnostic: the preparation on the main stream, nothing of the volume beside its own update)
    for (bool& u : s->updRecorded) u = false;
    s->barrierPending = false;
    return BF_OK;
}

// The fast contract gathers depth and colour of a sample as ONE 8-byte texel {depth f32, colour RGBX8}.  Without this call every operator interleaves its frame
// itself (one more launch on the allocation stream, per operator); a caller that keeps its frames anyway can interleave each frame once, when it arrives
// (bf_image_interleave_texels), and hand the image to every operator that integrates or de-integrates that frame: `d_texels` (width x height x 8 bytes) is
// consumed by the NEXT operator only.  The pipeline does this when the frames' texel images fit its budget.
int bf_scene_set_frame_texels(bf_scene* s, const void* d_texels) {
    BF_REQUIRE(s, "null scene");
    s->frameTexels = reinterpret_cast<const uint2*>(d_texels);
    return BF_OK;
}
int bf_image_interleave_texels(void* d_texels, const float* d_depth, const uint8_t* d_colorRGBX, uint32_t numPixels, void* hip_stream) {
    BF_REQUIRE(d_texels && d_depth && d_colorRGBX && numPixels > 0, "bad argument");
    hipLaunchKernelGGL(k_interleave, dim3(std::min<uint32_t>(div_up(numPixels, 256u), 2048u)), dim3(256), 0, (hipStream_t)hip_stream, d_depth, reinterpret_cast<const uint32_t*>(d_colorRGBX),
                       reinterpret_cast<uint2*>(d_texels), numPixels);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_scene_wait_event(bf_scene* s, void* hip_event) {
    BF_REQUIRE(s, "null scene");
    s->pendingEv = (hipEvent_t)hip_event;
    return BF_OK;
}

int bf_scene_reset(bf_scene* s) {                                  // CUDASceneRepHashSDF.h:147-155
    BF_REQUIRE(s, "null scene");
    s->numIntegrated = 0;
    const m44 I = identity44();
    memcpy(s->params.m_rigidTransform, I.e, 64);
    memcpy(s->params.m_rigidTransformInverse, I.e, 64);
    s->params.m_numOccupiedBlocks = 0;
    BF_TRY_RC(syncAll(s));
    s->compactStale = false; s->gcMask = 0; s->barrierPending = false; s->pendingEv = nullptr;

    for (bool& u : s->updRecorded) u = false;
    for (int b = 0; b < bf_scene::NBMAX; ++b) BF_HIP_TRY(hipMemsetAsync(s->ccnt[b], 0, 16, s->stream));
    const size_t numEntries = (size_t)s->params.m_hashNumBuckets * BF_HASH_BUCKET_SIZE;
    BF_HIP_TRY(hipMemsetAsync(s->d.vox, 0, (size_t)s->params.m_numSDFBlocks * VOX * sizeof(bf_voxel), s->stream));
    hipLaunchKernelGGL(k_reset, dim3(2048), dim3(256), 0, s->stream, s->d, s->params.m_numSDFBlocks, (uint32_t)numEntries,
                       s->dedupeSize);
    if (s->batchReady) hipLaunchKernelGGL(k_batch_reset, dim3(2048), dim3(256), 0, s->stream, s->bd, s->params.m_hashNumBuckets);
    BF_HIP_TRY(hipGetLastError());
    if (s->prep) BF_HIP_TRY(hipStreamSynchronize(s->stream));      // the next operator's allocation runs on the preparation stream
    return BF_OK;
}

int bf_scene_integrate(bf_scene* s, const float T[16], const bf_depth_camera_data* data,
                       const bf_depth_camera_params* cam, const uint32_t* d_bitMask) {
    int rc = checkCall(s, T, data, cam, d_bitMask);
    if (rc) return rc;
    s->cam = *cam; s->haveCam = true;
    setLastRigidTransform(s, T);
    const Frame f = makeFrame(s);
    if ((rc = runOperator(s, 0, f, f, data))) return rc;
    s->numIntegrated++;
    return BF_OK;
}

int bf_scene_deintegrate(bf_scene* s, const float T[16], const bf_depth_camera_data* data,
                         const bf_depth_camera_params* cam, const uint32_t* d_bitMask) {
    int rc = checkCall(s, T, data, cam, d_bitMask);
    if (rc) return rc;
    s->cam = *cam; s->haveCam = true;
    setLastRigidTransform(s, T);
    const Frame f = makeFrame(s);
    if ((rc = runOperator(s, 1, f, f, data))) return rc;
    s->numIntegrated--;
    return BF_OK;
}

// Hash-bucket sharding: this volume keeps only blocks whose home bucket lies in
// [rank * numBuckets / world, (rank + 1) * numBuckets / world).  Every shard sees every frame and pose; allocation inserts the
// owned keys only, so frustum lists, voxel updates and garbage collection shrink to the shard without further changes and
// without any exchange between shards.  The union of the shards' (key -> voxels) maps is the unsharded volume.
int bf_scene_set_shard(bf_scene* s, uint32_t rank, uint32_t world) {
    BF_REQUIRE(s && world >= 1 && rank < world, "bad shard");
    BF_REQUIRE(s->numIntegrated == 0, "set the shard before the first integration");
    const uint64_t nb = s->params.m_hashNumBuckets;
    s->shardLo = (uint32_t)(nb * rank / world);
    s->shardHi = (uint32_t)(nb * (rank + 1) / world);
    return BF_OK;
}

// De-integrate the frame at oldT and integrate it at newT as one fused pass (MI355X addition; the reference issues the two
// operators back to back, DepthSensing.cpp:882-889).  Bit-identical to bf_scene_deintegrate(oldT) + bf_scene_integrate(newT).
int bf_scene_reintegrate(bf_scene* s, const float oldT[16], const float newT[16], const bf_depth_camera_data* data,
                         const bf_depth_camera_params* cam) {
    int rc = checkCall(s, newT, data, cam, nullptr);
    if (rc) return rc;
    BF_REQUIRE(oldT, "null argument");
    s->cam = *cam; s->haveCam = true;
    setLastRigidTransform(s, oldT);
    const Frame fo = makeFrame(s);
    setLastRigidTransform(s, newT);
    const Frame f = makeFrame(s);
    return runOperator(s, 2, f, fo, data);
}

// A batch of operators in the serial order ops[0] .. ops[n - 1] (MI355X addition; DepthSensing.cpp:854-902 issues them one by one): the same table, heap and
// voxels as bf_scene_integrate / _deintegrate / _reintegrate called in that order, in four launches and one pass over the touched blocks.
int bf_scene_run_batch(bf_scene* s, const bf_scene_batch_op* ops, uint32_t n, const bf_depth_camera_params* cam) {
    BF_REQUIRE(s && ops && cam, "null argument");
    BF_REQUIRE(n >= 1 && n <= BF_SCENE_BATCH_MAX, "1 .. BF_SCENE_BATCH_MAX operators per batch");
    BF_REQUIRE(cam->m_imageWidth > 0 && cam->m_imageHeight > 0, "empty image");
    for (uint32_t k = 0; k < n; ++k) {
        BF_REQUIRE(ops[k].kind >= 0 && ops[k].kind <= 2, "unknown operator kind");
        BF_REQUIRE(ops[k].data.d_depthData, "d_depthData is null");
    }
    s->cam = *cam; s->haveCam = true;
    return runBatch(s, ops, n);
}

int bf_scene_set_last_rigid_transform_and_compactify(bf_scene* s, const float T[16], const bf_depth_camera_params* cam) {
    BF_REQUIRE(s && T && cam, "null argument");
    s->cam = *cam; s->haveCam = true;
    setLastRigidTransform(s, T);
    BF_TRY_RC(beginExclusive(s));
    BF_TRY_RC(launchCompactify(s));
    return endExclusive(s);
}

int bf_scene_garbage_collect(bf_scene* s) {                          // :110-126
    BF_REQUIRE(s, "null scene");
    if (!s->haveCam) return BF_OK;                                  // nothing was ever compactified
    BF_TRY_RC(beginExclusive(s));
    uint32_t needMask = 0;
    if (s->compactStale) { if (s->gcMask) needMask = s->gcMask; else BF_TRY_RC(launchCompactify(s)); }
    const Frame f = makeFrame(s);
    hipLaunchKernelGGL(k_gc_identify, dim3(4096), dim3(256), 0, s->stream, s->d, f, needMask);
    hipLaunchKernelGGL(k_gc_delete, dim3(NBINS), dim3(1024), 0, s->stream, s->d, f);
    hipLaunchKernelGGL(k_gc_finish, dim3(1), dim3(256), 0, s->stream, s->d);
    hipLaunchKernelGGL(k_compact_count, dim3(s->gridCompact), dim3(256), 0, s->stream, s->d);
    hipLaunchKernelGGL(k_compact_scatter, dim3(s->gridCompact), dim3(256), 0, s->stream, s->d);
    hipLaunchKernelGGL(k_list_commit, dim3(1), dim3(1), 0, s->stream, s->d);
    std::swap(s->d.allocList, s->d.allocListAlt);
    BF_HIP_TRY(hipGetLastError());
    // The list in d.compact refers to positions of the allocated-block list before its compaction: nothing usable.  It is rebuilt when somebody needs the
    // frustum list of the last pose (the accessors, a garbage collection with no operator in between); every operator builds its own list anyway.
    s->compactStale = true; s->gcMask = 0;
    return endExclusive(s);
}

int bf_scene_get_hash_data(bf_scene* s, bf_hash_data* out) {
    BF_REQUIRE(s && out, "null argument");
    BF_TRY_RC(refreshStaleList(s));
    BF_TRY_RC(syncAll(s));
    out->d_heap = s->d.heap;
    out->d_heapCounter = s->d.heapCounter;
    out->d_hashDecision = s->d_hashDecision;
    out->d_hashDecisionPrefix = nullptr;
    out->d_hash = s->d.hash;
    out->d_hashCompactified = s->d.compact;
    out->d_hashCompactifiedCounter = s->d.compactCount;
    out->d_SDFBlocks = s->d.vox;
    out->d_hashBucketMutex = nullptr;
    return BF_OK;
}

// Scratch-capacity conditions of the allocation (bin / de-dup set / overflow list full, allocated-block list full) are recorded on the
// device by the asynchronous operators; every accessor that synchronises anyway reports them (bf_scene_integrate itself never waits).
static int checkDeviceErrors(bf_scene* s) {
    uint32_t stats[ST_COUNT];
    BF_HIP_TRY(hipMemcpyAsync(stats, s->d.stats, sizeof stats, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    if (stats[ST_ERROR]) { set_error("TSDF scratch capacity exceeded during an earlier operator (error bits 0x%x: 1 bin, 2 de-dup set, 4 overflow list, 8 GC, 16 block list)", stats[ST_ERROR]); return BF_ERR_CAPACITY; }
    return BF_OK;
}

int bf_scene_get_hash_params(bf_scene* s, bf_hash_params* out) {
    BF_REQUIRE(s && out, "null argument");
    BF_TRY_RC(refreshStaleList(s));
    BF_TRY_RC(syncAll(s));
    int32_t n = 0;
    BF_HIP_TRY(hipMemcpyAsync(&n, s->d.compactCount, 4, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    s->params.m_numOccupiedBlocks = (uint32_t)n;
    *out = s->params;
    return checkDeviceErrors(s);
}

int bf_scene_get_heap_free_count(bf_scene* s, uint32_t* out) {       // :168-172
    BF_REQUIRE(s && out, "null argument");
    BF_TRY_RC(syncAll(s));
    uint32_t c = 0;
    BF_HIP_TRY(hipMemcpyAsync(&c, s->d.heapCounter, 4, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    *out = c + 1u;
    return BF_OK;
}

int bf_scene_get_num_integrated_frames(bf_scene* s, uint32_t* out) {
    BF_REQUIRE(s && out, "null argument");
    *out = s->numIntegrated;
    return BF_OK;
}

int bf_scene_set_last_rigid_transform(bf_scene* s, const float camToWorld[16]) {            // CUDASceneRepHashSDF.h:128-134
    BF_REQUIRE(s && camToWorld, "null argument");
    setLastRigidTransform(s, camToWorld);
    return BF_OK;
}

int bf_scene_get_num_allocated_blocks(bf_scene* s, uint32_t* out) {
    BF_REQUIRE(s && out, "null argument");
    BF_TRY_RC(syncAll(s));
    uint32_t n = 0;
    BF_HIP_TRY(hipMemcpyAsync(&n, s->d.allocCount, 4, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<AllocRec> recs(n);
    if (n) BF_HIP_TRY(hipMemcpy(recs.data(), s->d.allocList, (size_t)n * sizeof(AllocRec), hipMemcpyDeviceToHost));
    uint32_t live = 0;
    for (auto& r : recs) live += (r.ptr != BF_FREE_ENTRY);
    *out = live;
    return BF_OK;
}

int bf_scene_debug_hash(bf_scene* s, uint32_t out[6]) {              // debugHash :179-314
    BF_REQUIRE(s && out, "null argument");
    BF_TRY_RC(syncAll(s));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t numEntries = (size_t)s->params.m_hashNumBuckets * BF_HASH_BUCKET_SIZE;
    const uint32_t N = s->params.m_numSDFBlocks;
    std::vector<bf_hash_entry> hash(numEntries);
    std::vector<uint32_t> heap(N);
    uint32_t heapCounter = 0, stats[ST_COUNT];
    BF_HIP_TRY(hipMemcpy(hash.data(), s->d.hash, numEntries * sizeof(bf_hash_entry), hipMemcpyDeviceToHost));
    BF_HIP_TRY(hipMemcpy(heap.data(), s->d.heap, (size_t)N * 4, hipMemcpyDeviceToHost));
    BF_HIP_TRY(hipMemcpy(&heapCounter, s->d.heapCounter, 4, hipMemcpyDeviceToHost));
    BF_HIP_TRY(hipMemcpy(stats, s->d.stats, sizeof stats, hipMemcpyDeviceToHost));
    const uint32_t nFree = heapCounter + 1u;
    std::vector<uint8_t> state(N, 0);   // 1 = free, 2 = allocated
    uint32_t dupFree = 0;
    for (uint32_t i = 0; i < nFree && i < N; ++i) { if (state[heap[i]] == 1) dupFree++; state[heap[i]] = 1; }
    uint32_t occupied = 0, both = 0, dupKeys = 0;
    std::unordered_set<uint64_t> keys;
    for (size_t i = 0; i < numEntries; ++i) {
        if (hash[i].ptr == BF_FREE_ENTRY) continue;
        occupied++;
        const uint32_t blk = (uint32_t)hash[i].ptr / VOX;
        if (blk < N) { if (state[blk] == 1) both++; else state[blk] = 2; }
        i3 b; b.x = hash[i].pos[0]; b.y = hash[i].pos[1]; b.z = hash[i].pos[2];
        if (!keys.insert(packKey(b)).second) dupKeys++;
    }
    uint32_t leaked = dupFree;
    for (uint32_t i = 0; i < N; ++i) leaked += (state[i] == 0);
    out[0] = occupied; out[1] = nFree; out[2] = dupKeys; out[3] = both; out[4] = leaked; out[5] = stats[ST_DROPPED];
    if (stats[ST_ERROR]) { set_error("TSDF scratch capacity exceeded (error bits 0x%x)", stats[ST_ERROR]); return BF_ERR_CAPACITY; }
    return BF_OK;
}

// test / tooling aid: ptr of each of n blocks (d_pos: n x int3 block coordinates on the device) or FREE_ENTRY - the table as it stands behind everything issued so far
namespace {
__global__ void k_find_blocks(Dev d, uint32_t numBuckets, uint32_t maxChain, const int* __restrict__ pos, uint32_t n, int32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i3 b; b.x = pos[3 * i]; b.y = pos[3 * i + 1]; b.z = pos[3 * i + 2];
    const uint32_t h = hashPos(numBuckets, b), hp = h * BF_HASH_BUCKET_SIZE, total = BF_HASH_BUCKET_SIZE * numBuckets;
    int32_t r = BF_FREE_ENTRY;
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE && r == BF_FREE_ENTRY; ++j) { const bf_hash_entry e = d.hash[hp + j]; if (e.pos[0] == b.x && e.pos[1] == b.y && e.pos[2] == b.z && e.ptr != BF_FREE_ENTRY) r = e.ptr; }
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1;
    uint32_t off = d.hash[last].offset;
    for (uint32_t it = 0; it < maxChain && r == BF_FREE_ENTRY && off != 0; ++it) {
        const bf_hash_entry e = d.hash[(last + off) % total];
        if (e.pos[0] == b.x && e.pos[1] == b.y && e.pos[2] == b.z && e.ptr != BF_FREE_ENTRY) r = e.ptr;
        off = e.offset;
    }
    out[i] = r;
}
}  // namespace
int bf_scene_debug_find_blocks(bf_scene* s, const int32_t* d_pos, uint32_t n, int32_t* d_ptr_out) {
    BF_REQUIRE(s && d_pos && d_ptr_out, "null argument");
    BF_TRY_RC(syncAll(s));
    if (n) hipLaunchKernelGGL(k_find_blocks, dim3(div_up(n, 256u)), dim3(256), 0, s->stream, s->d, s->params.m_hashNumBuckets, s->params.m_hashMaxCollisionLinkedListSize, d_pos, n, d_ptr_out);
    BF_HIP_TRY(hipGetLastError());
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    return BF_OK;
}

int bf_scene_kernel_timing(bf_scene* s, int enable) {
    BF_REQUIRE(s, "null scene");
    s->timing = enable != 0;
    s->eventsUsed = 0;
    s->opsTimed = 0;
    s->imagesTimed = 0;
    BF_HIP_TRY(hipMemsetAsync(s->d.occSum, 0, 3 * sizeof(unsigned long long), s->stream));
    return BF_OK;
}

int bf_scene_kernel_timing_read(bf_scene* s, uint32_t* count, float* total_ms) {
    BF_REQUIRE(s && count && total_ms, "null argument");
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    float tot = 0.0f;
    for (size_t i = 0; i < s->eventsUsed; ++i) {
        float ms = 0.0f;
        BF_HIP_TRY(hipEventElapsedTime(&ms, s->events[i].first, s->events[i].second));
        tot += ms;
    }
    *count = (uint32_t)s->eventsUsed;
    *total_ms = tot;
    s->eventsUsed = 0;
    return BF_OK;
}

int bf_scene_kernel_timing_occupied(bf_scene* s, uint64_t* sumOccupiedBlocks, uint32_t* numOps) {
    return bf_scene_kernel_timing_blocks(s, sumOccupiedBlocks, nullptr, nullptr, numOps);
}

int bf_scene_kernel_timing_images(bf_scene* s, uint32_t* numImages) {
    BF_REQUIRE(s && numImages, "null argument");
    *numImages = s->imagesTimed;
    return BF_OK;
}

int bf_scene_kernel_timing_blocks(bf_scene* s, uint64_t* sumOperatorBlocks, uint64_t* visitedPlain, uint64_t* visitedFused, uint32_t* numOps) {
    BF_REQUIRE(s, "null argument");
    if (numOps) *numOps = s->opsTimed;
    unsigned long long v[3] = {0, 0, 0};
    BF_HIP_TRY(hipMemcpyAsync(v, s->d.occSum, sizeof v, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    if (sumOperatorBlocks) *sumOperatorBlocks = v[0];
    if (visitedPlain) *visitedPlain = v[1];
    if (visitedFused) *visitedFused = v[2];
    return BF_OK;
}

}  // extern "C"
