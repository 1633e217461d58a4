// Batched TSDF operators (bf_scene_run_batch) — included by tsdf.hip inside its anonymous namespace, behind the per-operator kernels.
//
// The reference issues the up to s_maxFrameFixes re-integrations of a frame one after the other, each with its own allocation, compactify and two
// passes over the volume (DepthSensing.cpp:854-902, CUDASceneRepHashSDF.h:65-155).  All operators of a frame are known when its commands are posted,
// and their results only depend on their ORDER, so a batch of B operators runs as
//   k_batch_march  one ray march over the B frames (no table access): every distinct in-frustum block key with the first operator that needs it
//   k_batch_bin    table look-up + ownership + binning per (first operator, home-bucket range) of the missing keys
//   k_batch_place  placement of all B operators' keys - the serial order's slots, heap blocks and list positions - and ONE union list of the batch: every
//                  allocated block that some operator touches, with two membership bits per operator (frustum of its integration / de-integration pose)
//   k_update_batch ONE pass over the union list: a wave loads a block's voxels once, applies the batch's operators in order, writes them back once
// Same table, same heap, same voxel bits as the B operators issued one by one (tests/test_tsdf_batch_gpu.py), with 4 launches instead of 3 B and one trip
// of every touched block through HBM instead of one per operator.

constexpr uint32_t BMAX = BF_SCENE_BATCH_MAX;
static_assert(2 * BMAX <= 32, "two membership bits per operator in the 32-bit flags word of a list entry");
static_assert(BMAX <= 12 && (uint64_t)NBINS * BINCAP <= (1u << 20), "record packing: operator in 4 bits, needing operators in 12, rank among an operator's keys in 20");

struct BatchDev {
    unsigned long long* set; uint32_t setMask;   // the batch's key set (own storage: the per-operator path releases its set key by key)
    uint32_t* opMask;                            // per slot: the operators that need the key (bit k = operator k)
    uint32_t* candList; uint32_t* candCount; uint32_t candCap;
    BinRec* bins;                                // [BMAX][NBINS][BINCAP]
    uint32_t* binCount;                          // [BMAX][NBINS]
    uint32_t* bucketCnt;                         // [numBuckets]: missing keys of this batch per home bucket (k_batch_bin; zeroed again by k_batch_place)
    uint32_t* flags;                             // see BatchSink
};

struct BatchCommon {
    bf_depth_camera_params cam;
    uint32_t numBuckets, maxChain, numSDFBlocks;
    float voxelSize, maxIntegrationDistance, truncScale, truncation;
    uint32_t shardLo, shardHi;
    uint32_t nOps;
    uint32_t tile0, tile1;          // the march's band of 8x8 pixel tiles (the whole image unless the march is divided over the ranks of a communicator)
};
// (kernel arguments read at a run-time index - an operator, a level, a job - are laid out so that no scalar load of them straddles a 64-byte line: see BatchUpdOpApx)
struct alignas(64) BatchMarchOp { m44 T, Tinv; const float* depth; const uint32_t* color; uint2* texel; uint32_t marches; uint32_t pad; };
struct BatchMarchArgs { BatchMarchOp op[BMAX]; };
// bits[k]: bit 0 operator k integrates (pose TinvIn[k]), bit 1 it de-integrates (pose TinvDe[k])
struct alignas(64) BatchFrusta { m44 TinvIn[BMAX], TinvDe[BMAX]; uint32_t bits[BMAX]; };

__global__ void k_batch_reset(BatchDev bd, uint32_t numBuckets) {
    const uint32_t stride = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = gid; i <= bd.setMask; i += stride) { bd.set[i] = EMPTY64; bd.opMask[i] = 0u; }
    for (uint32_t i = gid; i < numBuckets; i += stride) bd.bucketCnt[i] = 0u;
    for (uint32_t i = gid; i < BMAX * NBINS; i += stride) bd.binCount[i] = 0u;
    if (gid == 0) { bd.candCount[0] = 0u; bd.flags[0] = 0u; }
}

// ---------------------------------------------------------------------------------------
// march: grid (tiles / 4, operators)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_batch_march(Dev d, BatchDev bd, BatchCommon c, BatchMarchArgs a) {
    __shared__ unsigned long long setAll[4][WSET];
    __shared__ unsigned long long listAll[4][WLIST];
    const uint32_t op = blockIdx.y;
    const BatchMarchOp& o = a.op[op];
    Frame f;
    f.T = o.T; f.Tinv = o.Tinv; f.cam = c.cam;
    f.numBuckets = c.numBuckets; f.maxChain = c.maxChain; f.numSDFBlocks = c.numSDFBlocks;
    f.voxelSize = c.voxelSize; f.maxIntegrationDistance = c.maxIntegrationDistance; f.truncScale = c.truncScale; f.truncation = c.truncation;
    f.shardLo = c.shardLo; f.shardHi = c.shardHi; f.weightMax = 0.0f;
    BatchSink bs;
    bs.set = bd.set; bs.mask = bd.setMask; bs.opMask = bd.opMask; bs.list = bd.candList; bs.count = bd.candCount; bs.cap = bd.candCap; bs.flags = bd.flags; bs.op = op;
    TexelOut tx; tx.color = o.color; tx.texel = o.texel;
    const uint32_t tile = c.tile0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= c.tile1) return;          // wave-uniform
    marchTile<2>(d, f, o.depth, Collect{}, tx, bs, tile, o.marches != 0u, setAll[threadIdx.x >> 6], listAll[threadIdx.x >> 6]);
}

// ---------------------------------------------------------------------------------------
// the march divided over the ranks of a communicator (bf_scene_set_alloc_comm): every rank marches its band of pixel tiles for ALL operators of the batch, packs the
// distinct keys it met with their operator masks into a fixed-size record {count, {key, mask}[capacity]}, ONE all-gather per batch hands every rank every list, and
// each rank claims all of them in its own key set (the union of the bands' sets with the masks OR-ed: exactly the set the undivided march builds); binning,
// placement and the update follow unchanged.  (Round 5 fell back to one operator at a time here: one all-gather per operator.)
// ---------------------------------------------------------------------------------------
struct BatchRec { unsigned long long key; uint32_t mask, pad; };
__global__ void k_batch_pack(Dev d, BatchDev bd, uint32_t* count, BatchRec* recs, uint32_t capacity) {
    const uint32_t n = min(bd.candCount[0], bd.candCap);
    if (blockIdx.x == 0 && threadIdx.x == 0) { count[0] = min(n, capacity); count[1] = 0u; if (n > capacity) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); }      // raised, never dropped silently
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = bd.candList[i];
        if (i < capacity) { BatchRec r; r.key = bd.set[slot]; r.mask = bd.opMask[slot]; r.pad = 0u; recs[i] = r; }
        bd.set[slot] = EMPTY64; bd.opMask[slot] = 0u;          // the set is refilled from the gathered lists (this rank's own among them)
    }
}
__global__ void k_batch_pack_finish(BatchDev bd) { bd.candCount[0] = 0u; }
__global__ void k_batch_ingest(Dev d, BatchDev bd, const uint32_t* count, const BatchRec* recs, uint32_t capacity) {
    const uint32_t n = min(count[0], capacity);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BatchRec r = recs[i];
        uint32_t slot = (uint32_t)((r.key * 0x9E3779B97F4A7C15ull) >> 40) & bd.setMask;
        bool done = false;
        for (uint32_t probe = 0; probe <= bd.setMask && !done; ++probe) {          // claimCandidate with a whole mask
            const unsigned long long old = atomicCAS(&bd.set[slot], (unsigned long long)EMPTY64, r.key);
            if (old == EMPTY64) {
                const uint32_t pos = atomicAdd(bd.candCount, 1u);
                if (pos < bd.candCap) bd.candList[pos] = slot;
                else { atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); atomicOr(bd.flags, 2u); }
            }
            if (old == EMPTY64 || old == r.key) { atomicOr(&bd.opMask[slot], r.mask); done = true; }
            slot = (slot + 1) & bd.setMask;
        }
        if (!done) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_DEDUPE_FULL);
    }
}

// ---------------------------------------------------------------------------------------
// look-up + binning of the claimed keys (after everything that frees table entries)
// ---------------------------------------------------------------------------------------
// blockPresent, and the number of free slots of the home bucket on the way
BF_DEV bool blockPresentFree(const Dev& d, uint32_t numBuckets, uint32_t maxChain, i3 b, uint32_t h, uint32_t& freeSlots) {
    const uint32_t hp = h * BF_HASH_BUCKET_SIZE;
    const int4* e4 = reinterpret_cast<const int4*>(d.hash);
    int4 a[BF_HASH_BUCKET_SIZE];
#pragma unroll
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) a[j] = e4[(size_t)(hp + j) * 2];
    const uint32_t lastOffset = d.hash[hp + BF_HASH_BUCKET_SIZE - 1].offset;
    bool hit = false;
    freeSlots = 0;
#pragma unroll
    for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
        hit = hit || (a[j].x == b.x && a[j].y == b.y && a[j].z == b.z && a[j].w != BF_FREE_ENTRY);
        freeSlots += a[j].w == BF_FREE_ENTRY ? 1u : 0u;
    }
    if (hit) return true;
    if (lastOffset == 0) return false;
    const uint32_t last = hp + BF_HASH_BUCKET_SIZE - 1;
    const uint32_t total = BF_HASH_BUCKET_SIZE * numBuckets;
    uint32_t i = (last + lastOffset) % total;
    for (uint32_t it = 1; it < maxChain; ++it) {
        const int4 cc = e4[(size_t)i * 2];
        const uint32_t off = d.hash[i].offset;
        if (cc.x == b.x && cc.y == b.y && cc.z == b.z && cc.w != BF_FREE_ENTRY) return true;
        if (off == 0) break;
        i = (last + off) % total;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_batch_bin(Dev d, BatchDev bd, BatchCommon c) {
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.compactCount[0] = 0; d.compactCount[1] = 0; }      // the batch's list: k_batch_place appends to it
    const uint32_t n = min(bd.candCount[0], bd.candCap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = bd.candList[i];
        const uint64_t key = bd.set[slot];
        const uint32_t need = bd.opMask[slot];                      // != 0: the claimant set its bit before the march ended
        const uint32_t op = (uint32_t)__ffs((int)need) - 1u;         // the first operator that needs the block allocates it
        bd.set[slot] = EMPTY64; bd.opMask[slot] = 0u;               // nobody probes the set during this kernel
        const i3 b = unpackKey(key);
        const uint32_t h = hashPos(c.numBuckets, b);
        if (h < c.shardLo || h >= c.shardHi) continue;
        uint32_t freeSlots = 0;
        if (blockPresentFree(d, c.numBuckets, c.maxChain, b, h, freeSlots)) continue;
        const uint32_t bin = (uint32_t)(((uint64_t)h * NBINS) / c.numBuckets);
        const uint32_t pos = atomicAdd(&bd.binCount[op * NBINS + bin], 1u);
        if (pos >= BINCAP) { atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); continue; }
        BinRec r; r.key = key; r.bucket = h; r.aux = (need << 4) | op;      // aux: operator in the low 4 bits, above them every operator that needs the key
        bd.bins[((size_t)(op * NBINS + bin)) * BINCAP + pos] = r;
        // a home bucket that cannot take all its new keys sends some of them through the collision window of OTHER buckets: the one case in which the
        // operators' placements depend on each other beyond the bucket itself - k_batch_place then replays the batch operator by operator
        const uint32_t before = atomicAdd(&bd.bucketCnt[h], 1u);
        if (before + 1u > freeSlots) atomicOr(bd.flags, 1u);
    }
}

// ---------------------------------------------------------------------------------------
// placement of the whole batch + its union list
// ---------------------------------------------------------------------------------------
BF_DEV uint32_t keepMask(const BatchCommon& c, const BatchFrusta& fr, uint64_t key, int32_t ptr, uint32_t birth) {
    if (ptr == BF_FREE_ENTRY) return 0u;
    const i3 b = unpackKey(key);
    uint32_t m = 0u;
    for (uint32_t k = 0; k < c.nOps; ++k) {          // (uniform trip count: the poses come from the kernel arguments through scalar loads)
        if (k < birth) continue;                     // the block does not exist yet when operator k runs
        const uint32_t bits = fr.bits[k];
        if ((bits & 1u) && blockInFrustumT(fr.TinvIn[k], c.cam, c.voxelSize, b)) m |= 1u << (2u * k);
        if ((bits & 2u) && blockInFrustumT(fr.TinvDe[k], c.cam, c.voxelSize, b)) m |= 2u << (2u * k);
    }
    return m;
}

// every lane of the (converged) wave calls this; lanes with keep != 0 append their block with its membership bits
BF_DEV void batchAppendWave(const Dev& d, uint32_t keep, uint64_t key, int32_t ptr, uint32_t src) {
    const unsigned long long m = __ballot(keep != 0u);
    if (m == 0ull) return;                                      // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t ob = (uint32_t)wave_sum_i((int)__popc(keep));
    uint32_t base = 0;
    if ((int)lane == leader) {
        base = atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), (uint32_t)__popcll(m));
        atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, ob);
    }
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (keep) listWrite(d, base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), key, ptr, keep, src);
}
// every thread of the (converged) workgroup calls this with at most one block: one reservation per workgroup
BF_DEV void batchAppendBlock(const Dev& d, uint32_t keep, uint64_t key, int32_t ptr, uint32_t src, uint32_t* wscan, uint32_t* sbase) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(keep != 0u);
    const uint32_t cnt = (uint32_t)__popcll(m);
    const uint32_t ob = (uint32_t)wave_sum_i((int)__popc(keep));
    if (lane == 0) { wscan[wave] = cnt; if (ob) atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, ob); }
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = wscan[0] + wscan[1] + wscan[2] + wscan[3]; *sbase = tot ? atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), tot) : 0u; }
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; ++w) woff += wscan[w];
    if (keep) listWrite(d, *sbase + woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), key, ptr, keep, src);
    __syncthreads();
}

// LDS bitonic sort by (home bucket, operator [low 4 bits of aux], key): inside a bucket the serial order of slot consumption
BF_DEV void ldsBitonicSort3(SortLds& s, uint32_t npad) {
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                const bool up = ((lo & k) == 0);
                const uint32_t bl = s.bucket[lo], bh = s.bucket[hi];
                const uint32_t al = s.aux[lo], ah = s.aux[hi];
                const uint64_t kl = s.key[lo], kh = s.key[hi];
                const bool gt = (bl != bh) ? (bl > bh) : ((al & 15u) != (ah & 15u)) ? ((al & 15u) > (ah & 15u)) : (kl > kh);
                if (gt == up) {
                    s.bucket[lo] = bh; s.bucket[hi] = bl;
                    s.key[lo] = kh; s.key[hi] = kl;
                    s.aux[lo] = ah; s.aux[hi] = al;
                }
            }
            __syncthreads();
        }
    }
}

// the rank-th free slot of home bucket h (rank counted over the batch's new keys of that bucket in serial order), or -1
BF_DEV int freeSlotOfRank(const Dev& d, uint32_t h, uint32_t rank) {
    int slot = -1;
    if (rank < BF_HASH_BUCKET_SIZE) {
        uint32_t seen = 0;
#pragma unroll
        for (uint32_t j = 0; j < BF_HASH_BUCKET_SIZE; ++j) {
            const int32_t p = d.hash[h * BF_HASH_BUCKET_SIZE + j].ptr;
            if (p == BF_FREE_ENTRY) { if (seen == rank && slot < 0) slot = (int)j; ++seen; }
        }
    }
    return slot;
}

BF_DEV void writeEntry(const Dev& d, uint32_t h, int slot, uint64_t key, int32_t ptr) {      // {pos.xyz, ptr, offset = NO_OFFSET, pad}, VoxelUtilHashSDF.h:600-612
    const i3 b = unpackKey(key);
    uint64_t* e = reinterpret_cast<uint64_t*>(d.hash + ((size_t)h * BF_HASH_BUCKET_SIZE + (uint32_t)slot));
    storeThrough(e, pack2((uint32_t)b.x, (uint32_t)b.y)); storeThrough(e + 1, pack2((uint32_t)b.z, (uint32_t)ptr));
    storeThrough(e + 2, 0ull); storeThrough(e + 3, 0ull);
}

struct BatchLds {
    SortLds s;
    int8_t sel[BINCAP];
    uint16_t opIdx[BINCAP];
    uint32_t cnts[BMAX * NBINS];
    uint32_t opM[BMAX], opPre[BMAX], opHeapC[BMAX + 1], opAllocBase[BMAX + 1], opHeapFree[BMAX];
    uint32_t wscan[4], wmaxs[4], sbase, lastFlag, slow, rp[4];
};

// Fast path, one workgroup per bin: the bin's new keys of ALL operators, sorted by (home bucket, operator, key).  Inside a home bucket that is the order in
// which the serial operators consume its free slots; the heap block and the position in the allocated-block list of a key follow from its operator's
// offsets (the operators before it took sum M_j blocks) and its rank among its operator's keys (bins are bucket ranges: the bins before this one come first).
BF_DEV void batchPlaceBin(const Dev& d, const BatchDev& bd, const BatchCommon& c, const BatchFrusta& fr, BatchLds& L, uint32_t bin, uint32_t total) {
    SortLds& s = L.s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t npad = nextPow2(total);
    uint32_t off = 0;
    for (uint32_t k = 0; k < c.nOps; ++k) {
        const uint32_t nk = L.cnts[k * NBINS + bin];
        const BinRec* recs = bd.bins + ((size_t)(k * NBINS + bin)) * BINCAP;
        for (uint32_t i = tid; i < nk; i += 256u) { const BinRec r = recs[i]; s.key[off + i] = r.key; s.bucket[off + i] = r.bucket; s.aux[off + i] = r.aux; }
        off += nk;
    }
    for (uint32_t i = total + tid; i < npad; i += 256u) { s.key[i] = EMPTY64; s.bucket[i] = 0xFFFFFFFFu; s.aux[i] = 0xFFFFFFFFu; }
    __syncthreads();
    ldsBitonicSort3(s, npad);
    // rank of every key among its operator's keys of this bin (sorted order): one block-wide exclusive scan per operator present
    const uint32_t E = (npad + 255u) / 256u, c0 = tid * E;
    for (uint32_t k = 0; k < c.nOps; ++k) {
        if (L.cnts[k * NBINS + bin] == 0u) continue;      // block-uniform
        uint32_t cnt = 0;
        for (uint32_t e = 0; e < E; ++e) { const uint32_t j = c0 + e; if (j < total && (s.aux[j] & 15u) == k) ++cnt; }
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
        if (lane == 63) L.wscan[wave] = incl;
        __syncthreads();
        uint32_t run = incl - cnt;
        for (uint32_t w = 0; w < wave; ++w) run += L.wscan[w];
        for (uint32_t e = 0; e < E; ++e) { const uint32_t j = c0 + e; if (j < total && (s.aux[j] & 15u) == k) L.opIdx[j] = (uint16_t)run++; }
        __syncthreads();
    }
    // slots (reads only: nobody else writes this bin's buckets)
    for (uint32_t idx = tid; idx < total; idx += 256u) {
        const uint32_t h = s.bucket[idx];
        uint32_t rank = 0;
        while (rank < BF_HASH_BUCKET_SIZE && idx > rank && s.bucket[idx - rank - 1] == h) ++rank;
        L.sel[idx] = (int8_t)freeSlotOfRank(d, h, rank);
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < total; i0 += 256u) {          // whole waves walk the loop: the list append is a wave operation
        const uint32_t idx = i0 + tid;
        uint32_t keep = 0, src = 0; uint64_t key = 0; int32_t ptr = 0;
        if (idx < total) {
            key = s.key[idx];
            const uint32_t h = s.bucket[idx], k = s.aux[idx] & 15u;
            const uint32_t gi = L.opPre[k] + (uint32_t)L.opIdx[idx];
            bd.bucketCnt[h] = 0u;
            // heap exhausted: in the serial order operator k drops the key and so does every later operator that needs it (nothing gives a block back inside a
            // batch on this path), each counting its own drop
            if (gi >= L.opHeapFree[k]) atomicAdd(&d.stats[ST_DROPPED], (uint32_t)__popc(s.aux[idx] >> (4u + k)));
            else {
                ptr = (int32_t)(d.heap[L.opHeapC[k] - gi] * (uint32_t)VOX);                                   // consumeHeap :536-540
                src = L.opAllocBase[k] + gi;
                uint64_t* ar = reinterpret_cast<uint64_t*>(d.allocList + src);
                storeThrough(ar, key); storeThrough(ar + 1, pack2((uint32_t)ptr, 0u));
                const int slot = L.sel[idx];
                if (slot >= 0) { writeEntry(d, h, slot, key, ptr); keep = keepMask(c, fr, key, ptr, k); }
                else atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_OV_OVERFLOW);      // cannot happen: k_batch_bin sends such a batch to the replay
            }
        }
        batchAppendWave(d, keep, key, ptr, src);
    }
    __syncthreads();
}

// replay only: operator k could not place the key (heap or collision window exhausted) - the serial order's next operator that needs the block tries again
BF_DEV void batchRequeue(const Dev& d, const BatchDev& bd, const BatchCommon& c, BatchLds& L, uint64_t key, uint32_t h, uint32_t need, uint32_t k) {
    const uint32_t rest = need >> (k + 1u);
    if (rest == 0u) return;
    const uint32_t kn = k + (uint32_t)__ffs((int)rest);
    const uint32_t bin = (uint32_t)(((uint64_t)h * NBINS) / c.numBuckets);
    const uint32_t pos = atomicAdd(&L.cnts[kn * NBINS + bin], 1u);
    if (pos >= BINCAP) { atomicSub(&L.cnts[kn * NBINS + bin], 1u); atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_BIN_OVERFLOW); return; }
    uint64_t* r = reinterpret_cast<uint64_t*>(bd.bins + ((size_t)(kn * NBINS + bin)) * BINCAP + pos);
    storeThrough(r, key); storeThrough(r + 1, pack2(h, (need << 4) | kn));
    atomicAdd(&L.opM[kn], 1u);
}

// Replay (one workgroup, the last one): the batch's operators one after the other, each bin by bin and followed by the walk of its bucket-full keys through
// their collision windows - the serial algorithm of k_alloc_place (placeBin / placeTail) on the batch's bins.  Taken when a home bucket cannot hold all its
// new keys (then a key lands in ANOTHER bucket and may shift what later operators find there) or a bin's keys of all operators exceed the LDS sort.
BF_DEV void batchReplay(const Dev& d, const BatchDev& bd, const BatchCommon& c, const BatchFrusta& fr, BatchLds& L, uint32_t heapC0, uint32_t allocBase0) {
    SortLds& s = L.s;
    const uint32_t tid = threadIdx.x, N = c.numSDFBlocks, totalSlots = BF_HASH_BUCKET_SIZE * c.numBuckets;
    uint32_t hc = heapC0, ab = allocBase0;
    for (uint32_t k = 0; k < c.nOps; ++k) {
        const uint32_t M = L.opM[k];
        if (M == 0u) continue;
        const uint32_t listRoom = N - min(ab, N);
        const uint32_t heapFree = min(hc + 1u, listRoom);
        if (tid == 0 && M > listRoom && listRoom < hc + 1u) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_LIST_FULL);
        const uint32_t Mp = min(M, heapFree);
        if (tid == 0) L.rp[2] = 0u;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t b = 0; b < NBINS; ++b) {
            const uint32_t n = L.cnts[k * NBINS + b];
            if (n == 0u) continue;
            loadBinSorted(s, bd.bins + ((size_t)(k * NBINS + b)) * BINCAP, n);          // by (home bucket, key)
            for (uint32_t idx = tid; idx < n; idx += 256u) {
                const uint32_t h = s.bucket[idx];
                uint32_t rank = 0;
                while (rank < BF_HASH_BUCKET_SIZE && idx > rank && s.bucket[idx - rank - 1] == h) ++rank;
                L.sel[idx] = (int8_t)freeSlotOfRank(d, h, rank);
            }
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
                const uint32_t idx = i0 + tid;
                uint32_t keep = 0, src = 0; uint64_t key = 0; int32_t ptr = 0;
                if (idx < n) {
                    key = s.key[idx];
                    const uint32_t h = s.bucket[idx], gi = base + idx;
                    bd.bucketCnt[h] = 0u;
                    if (gi >= heapFree) { atomicAdd(&d.stats[ST_DROPPED], 1u); batchRequeue(d, bd, c, L, key, h, s.aux[idx] >> 4, k); }
                    else {
                        ptr = (int32_t)(d.heap[hc - gi] * (uint32_t)VOX);
                        src = ab + gi;
                        uint64_t* ar = reinterpret_cast<uint64_t*>(d.allocList + src);
                        storeThrough(ar, key); storeThrough(ar + 1, pack2((uint32_t)ptr, 0u));
                        const int slot = L.sel[idx];
                        if (slot >= 0) { writeEntry(d, h, slot, key, ptr); keep = keepMask(c, fr, key, ptr, k); }
                        else {
                            const uint32_t ov = atomicAdd(&L.rp[2], 1u);
                            if (ov < OVCAP) { uint64_t* o = reinterpret_cast<uint64_t*>(d.overflow + ov); storeThrough(o, key); storeThrough(o + 1, pack2(h, (gi << 12) | ((s.aux[idx] >> 4) & 0xFFFu))); }
                            else atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_OV_OVERFLOW);
                        }
                    }
                }
                batchAppendWave(d, keep, key, ptr, src);
            }
            base += n;
            __threadfence();              // what this bin wrote is what the next bins, the walk below and the next operators read (write-back + L1 invalidate)
            __syncthreads();
        }
        const uint32_t nov = min(L.rp[2], OVCAP);
        if (nov > 0u) loadBinSorted(s, d.overflow, nov);
        if (tid == 0) {                   // VoxelUtilHashSDF.h:614-654, keys in sorted order
            uint32_t newCounter = hc - Mp, dropped = 0;
            for (uint32_t q = 0; q < nov; ++q) {
                const uint32_t h = s.bucket[q], gi = s.aux[q] >> 12, need = s.aux[q] & 0xFFFu;      // (gi < NBINS * BINCAP = 2^20)
                const uint32_t last = (h + 1) * BF_HASH_BUCKET_SIZE - 1;
                const int32_t ptr = d.allocList[ab + gi].ptr;
                bool done = false;
                uint32_t maxIter = 0;
                int offset = 0;
                while (maxIter < c.maxChain) {
                    offset++;
                    const uint32_t i = (last + (uint32_t)offset) % totalSlots;
                    if ((offset % BF_HASH_BUCKET_SIZE) == 0) continue;      // never a bucket's last slot :624
                    if (d.hash[i].ptr == BF_FREE_ENTRY) {
                        const i3 bb = unpackKey(s.key[q]);
                        d.hash[i].pos[0] = bb.x; d.hash[i].pos[1] = bb.y; d.hash[i].pos[2] = bb.z;
                        d.hash[i].offset = d.hash[last].offset;
                        d.hash[i].ptr = ptr;
                        d.hash[last].offset = (uint32_t)offset;
                        done = true;
                        const uint32_t keep = keepMask(c, fr, s.key[q], ptr, k);
                        if (keep) {
                            const uint32_t pos = atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount), 1u);
                            atomicAdd(reinterpret_cast<uint32_t*>(d.compactCount) + 1, (uint32_t)__popc(keep));
                            listWrite(d, pos, s.key[q], ptr, keep, ab + gi);
                        }
                        break;
                    }
                    maxIter++;
                }
                if (!done) {                                                  // window exhausted: give the block back
                    d.allocList[ab + gi].ptr = BF_FREE_ENTRY;
                    newCounter++;
                    d.heap[newCounter] = (uint32_t)ptr / (uint32_t)VOX;       // appendHeap :542-546
                    dropped++;
                    batchRequeue(d, bd, c, L, s.key[q], h, need, k);
                }
            }
            if (dropped) atomicAdd(&d.stats[ST_DROPPED], dropped);
            L.rp[0] = newCounter; L.rp[1] = ab + Mp;
        }
        __threadfence();
        __syncthreads();
        hc = L.rp[0]; ab = L.rp[1];
        __syncthreads();
    }
    if (tid == 0) { L.opHeapC[c.nOps] = hc; L.opAllocBase[c.nOps] = ab; }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_batch_place(Dev d, BatchDev bd, BatchCommon c, BatchFrusta fr) {
    __shared__ BatchLds L;
    __builtin_amdgcn_s_setprio(3);
    static_assert(PLACE_WGS == NBINS, "one workgroup per bin");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, bin = blockIdx.x, nOps = c.nOps, N = c.numSDFBlocks;
    // first round trip: the counts of all (operator, bin) pairs, the counters, the batch's flags, and this thread's entry of the allocated-block list for the list pass
    for (uint32_t i = tid; i < nOps * NBINS; i += 256u) L.cnts[i] = min(bd.binCount[i], BINCAP);
    const uint32_t heapC0 = d.heapCounter[0], allocBase0 = d.allocCount[0], fl = bd.flags[0];
    AllocRec rec0; rec0.key = 0; rec0.ptr = BF_FREE_ENTRY; rec0.pad = 0;
    const uint32_t i0 = bin * 256u + tid;
    if (i0 < N) rec0 = d.allocList[i0];
    __syncthreads();
    for (uint32_t k = wave; k < nOps; k += 4u) {              // keys per operator, and of the bins before this one
        uint32_t tot = 0, pre = 0;
        for (uint32_t j = lane; j < NBINS; j += 64u) { const uint32_t v = L.cnts[k * NBINS + j]; tot += v; if (j < bin) pre += v; }
        tot = (uint32_t)wave_sum_i((int)tot); pre = (uint32_t)wave_sum_i((int)pre);
        if (lane == 0) { L.opM[k] = tot; L.opPre[k] = pre; }
    }
    {   // the fullest bin over all operators (every workgroup computes the same value: the mode needs no agreement protocol)
        uint32_t t = 0;
        for (uint32_t k = 0; k < nOps; ++k) t += L.cnts[k * NBINS + tid];
        t = wave_max_u(t);
        if (lane == 0) L.wmaxs[wave] = t;
    }
    __syncthreads();
    if (tid == 0) {       // the operators' heap / list offsets in the serial order (no key bucket-full: nothing is given back in between)
        uint32_t hc = heapC0, ab = allocBase0;
        for (uint32_t k = 0; k < nOps; ++k) {
            const uint32_t heapFree = min(hc + 1u, N - min(ab, N));
            const uint32_t Mp = min(L.opM[k], heapFree);
            L.opHeapC[k] = hc; L.opAllocBase[k] = ab; L.opHeapFree[k] = heapFree;
            hc -= Mp; ab += Mp;
        }
        L.opHeapC[nOps] = hc; L.opAllocBase[nOps] = ab;
        const uint32_t fullest = max(max(L.wmaxs[0], L.wmaxs[1]), max(L.wmaxs[2], L.wmaxs[3]));
        L.slow = ((fl & 1u) || fullest > BINCAP) ? 1u : 0u;
    }
    __syncthreads();
    const bool slowMode = L.slow != 0u;
    uint32_t total = 0;
    for (uint32_t k = 0; k < nOps; ++k) total += L.cnts[k * NBINS + bin];
    if (!slowMode && total > 0u) batchPlaceBin(d, bd, c, fr, L, bin, total);          // block-uniform
    // the list pass over the allocated-block list as it stood before this batch (blocks that exist for every operator)
    if (allocBase0 <= gridDim.x * 256u) batchAppendBlock(d, i0 < allocBase0 ? keepMask(c, fr, rec0.key, rec0.ptr, 0u) : 0u, rec0.key, rec0.ptr, i0, L.wscan, &L.sbase);
    else {
        for (uint32_t tile = bin; tile * 256u < allocBase0; tile += gridDim.x) {
            const uint32_t i = tile * 256u + tid;
            AllocRec r; r.key = 0; r.ptr = BF_FREE_ENTRY; r.pad = 0;
            if (i < allocBase0) r = d.allocList[i];
            batchAppendBlock(d, keepMask(c, fr, r.key, r.ptr, 0u), r.key, r.ptr, i, L.wscan, &L.sbase);
        }
    }
    // hand-off to the workgroup that arrives last (see k_alloc_place)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t ticket = atomicAdd(&d.stats[ST_TICKET], 1u);
        L.lastFlag = ticket == gridDim.x - 1 ? 1u : 0u;
        if (L.lastFlag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!L.lastFlag) return;
    if (slowMode) batchReplay(d, bd, c, fr, L, heapC0, allocBase0);
    else if (tid == 0) {
        for (uint32_t k = 0; k < nOps; ++k) {
            const uint32_t listRoom = N - min(L.opAllocBase[k], N);
            if (L.opM[k] > listRoom && listRoom < L.opHeapC[k] + 1u) atomicOr(&d.stats[ST_ERROR], (uint32_t)ERR_LIST_FULL);
        }
    }
    if (tid == 0) {
        d.heapCounter[0] = L.opHeapC[nOps];
        d.allocCount[0] = L.opAllocBase[nOps];
        d.allocSnap[0] = L.opAllocBase[nOps];
        d.overflowCount[0] = 0;
        bd.candCount[0] = 0u; bd.flags[0] = 0u;
        d.stats[ST_TICKET] = 0;
    }
    for (uint32_t i = tid; i < nOps * NBINS; i += 256u) bd.binCount[i] = 0u;
    if (fl & 2u)          // the slot list overflowed: claimed slots nobody recorded - at this point every slot of the set belongs to a handled or dropped key
        for (uint32_t i = tid; i <= bd.setMask; i += 256u) { bd.set[i] = EMPTY64; bd.opMask[i] = 0u; }
}

// ---------------------------------------------------------------------------------------
// the batch's voxel update, fast contract: one wave per block of the union list, lane = (x, y) column.  The block's 512 voxels are loaded ONCE into registers
// (24 per lane), the batch's operators are applied in order - each gathers its samples from its own frame's texel image -, and every slice some lane changed
// goes back whole.  Per voxel and operator the operations are k_update_apx's (apxSamples / apxCompute): the same bits as the operators issued one by one.
// ---------------------------------------------------------------------------------------
// 64-byte aligned, each pose in a cache line of its own: the wave reads an operator's constants with scalar loads at a run-time index, once per block and operator,
// and no s_load_dwordx8 / x4 of them may straddle a 64-byte line (with the natural 104-byte stride they did).
struct alignas(64) BatchUpdOpApx { ApxPose in, de; const uint2* tex; };          // (ApxPose is 64-byte aligned itself)
struct BatchUpdApxArgs { BatchUpdOpApx op[BMAX]; uint32_t nOps; uint32_t liveMask; };      // liveMask: membership bits of the operators that update voxels (an operator without colour data does not)

#ifdef BF_VAR_VLOAD
struct KargMirrorApx { Dev d; ApxCam c; BatchUpdApxArgs a; int accumulate; };
BF_DEV BatchUpdOpApx loadOpVector(uint32_t k) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KargMirrorApx, a) + (uint64_t)k * sizeof(BatchUpdOpApx);
    const volatile v4u* src = reinterpret_cast<const volatile v4u*>(ka);
    union { BatchUpdOpApx o; uint32_t w[sizeof(BatchUpdOpApx) / 4]; } u;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(BatchUpdOpApx) / 16; ++i) {
        const v4u v = src[i];
        u.w[4 * i + 0] = __builtin_amdgcn_readfirstlane(v.x); u.w[4 * i + 1] = __builtin_amdgcn_readfirstlane(v.y);
        u.w[4 * i + 2] = __builtin_amdgcn_readfirstlane(v.z); u.w[4 * i + 3] = __builtin_amdgcn_readfirstlane(v.w);
    }
    return u.o;
}
#endif

template <bool RNE>
__global__ __launch_bounds__(256) void k_update_batch_apx(Dev d, ApxCam c, BatchUpdApxArgs a, int accumulate) {
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), nWaves = gridDim.x * 4u;
    if (accumulate && wave == 0 && lane == 0) { d.occSum[2] += (unsigned long long)n; d.occSum[0] += (unsigned long long)(uint32_t)d.compactCount[1]; }
    if (wave >= n) return;
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const int4 e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];                                    // wave-uniform
        const uint32_t mask = reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4] & a.liveMask;
        if (mask == 0u) continue;
        uint32_t* base = reinterpret_cast<uint32_t*>(d.vox + ((size_t)(uint32_t)e.w + lane));
        v2f vS[4], vW[4]; uint32_t vCA[4], vCB[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t* vpA = base + (size_t)(2 * p) * 64u * 3u; const uint32_t* vpB = vpA + 64u * 3u;
            vS[p].x = __uint_as_float(vpA[0]); vW[p].x = __uint_as_float(vpA[1]); vCA[p] = vpA[2];
            vS[p].y = __uint_as_float(vpB[0]); vW[p].y = __uint_as_float(vpB[1]); vCB[p] = vpB[2];
        }
        uint32_t dirty = 0u;
        for (uint32_t k = 0; k < a.nOps; ++k) {
            const uint32_t m = (mask >> (2u * k)) & 3u;                   // wave-uniform: bit 0 the block lies in the frustum of operator k's integration pose, bit 1 of its de-integration pose
            if (m == 0u) continue;
#ifdef BF_VAR_VLOAD          // diagnostic variant: the operator's record through vector loads of the kernel-argument segment + readfirstlane (no scalar load at a run-time index)
            const BatchUpdOpApx o = loadOpVector(k);
#else
            const BatchUpdOpApx& o = a.op[k];
#endif
            const __amdgpu_buffer_rsrc_t texRes = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(o.tex), 0, (int)(2u * c.bytes), 0x00020000);
            const ApxBlock cur = apxBlock<true, true>(d, c, o.in, o.de, e, m, lane);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#ifdef BF_VAR_REVERSE_PAIRS          // diagnostic variant: the pair behind the operator's set-up is the block's LAST one (does the error follow the position in the instruction stream?)
                const int p = 3 - pp;
#else
                const int p = pp;
#endif
                ApxPair pa;
                apxSamples<true, true>(c, o.in, o.de, cur, 2 * p, texRes, pa);
                bool stA, stB;
                apxCompute<true, true, RNE>(c, pa, vS[p], vW[p], vCA[p], vCB[p], stA, stB);
                dirty |= ((stA ? 1u : 0u) << (2 * p)) | ((stB ? 2u : 0u) << (2 * p));
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t* vpA = base + (size_t)(2 * p) * 64u * 3u; uint32_t* vpB = vpA + 64u * 3u;
            if (__builtin_amdgcn_ballot_w64((dirty >> (2 * p)) & 1u) != 0ull) { vpA[0] = __float_as_uint(vS[p].x); vpA[1] = __float_as_uint(vW[p].x); vpA[2] = vCA[p]; }
            if (__builtin_amdgcn_ballot_w64((dirty >> (2 * p + 1)) & 1u) != 0ull) { vpB[0] = __float_as_uint(vS[p].y); vpB[1] = __float_as_uint(vW[p].y); vpB[2] = vCB[p]; }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Diagnostic (BF_DEBUG_VERIFY_BATCH=<file>): every fast-contract batch update runs three times - on the volume and on two shadow copies of the union list's
// blocks taken just before - and the three results are compared voxel by voxel.  The update is a pure function of (block, operators, texel images): any
// difference is an execution error of one of the three launches, recorded with the two other values (the majority is the truth).  The frame loop keeps its
// structure (the other streams keep running beside the three launches), so a 2000-frame stream is ~2000 trials of the real kernel under its real neighbours.
// ---------------------------------------------------------------------------------------
struct VerifyRec { uint32_t seq, blk, vox, which; int32_t ex, ey, ez; uint32_t mask; uint32_t val[3][3]; uint32_t nOps, hwWave, pad[5]; };      // 96 bytes
static_assert(sizeof(VerifyRec) == 96, "record layout (tools/verify_stream.py)");
struct VerifyLog { uint32_t count, cap, batches, blocks; VerifyRec rec[1]; };

__global__ __launch_bounds__(256) void k_verify_copy(Dev d, bf_voxel* s0, bf_voxel* s1) {
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6), nWaves = gridDim.x * 4u;
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const uint32_t ptr = (uint32_t)reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2].w;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(d.vox + ptr);
        uint32_t* a = reinterpret_cast<uint32_t*>(s0 + ptr); uint32_t* b = reinterpret_cast<uint32_t*>(s1 + ptr);
        for (uint32_t i = lane; i < VOX * 3u; i += 64u) { const uint32_t v = src[i]; a[i] = v; b[i] = v; }
    }
}
__global__ __launch_bounds__(256) void k_verify_compare(Dev d, const bf_voxel* s0, const bf_voxel* s1, VerifyLog* log, uint32_t seq, uint32_t nOps, uint32_t nWavesUpd) {
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6), nWaves = gridDim.x * 4u;
    if (wave == 0 && lane == 0) { atomicAdd(&log->batches, 1u); atomicAdd(&log->blocks, n); }
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const int4 e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];
        const uint32_t mask = reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4];
        const uint32_t* r = reinterpret_cast<const uint32_t*>(d.vox + (uint32_t)e.w);
        const uint32_t* a = reinterpret_cast<const uint32_t*>(s0 + (uint32_t)e.w); const uint32_t* b = reinterpret_cast<const uint32_t*>(s1 + (uint32_t)e.w);
        for (uint32_t v = lane; v < VOX; v += 64u) {
            uint32_t x[3][3];
            for (int q = 0; q < 3; ++q) { x[0][q] = r[v * 3 + q]; x[1][q] = a[v * 3 + q]; x[2][q] = b[v * 3 + q]; }
            const bool d01 = x[0][0] != x[1][0] || x[0][1] != x[1][1] || x[0][2] != x[1][2];
            const bool d02 = x[0][0] != x[2][0] || x[0][1] != x[2][1] || x[0][2] != x[2][2];
            const bool d12 = x[1][0] != x[2][0] || x[1][1] != x[2][1] || x[1][2] != x[2][2];
            if (d01 || d02 || d12) {
                const uint32_t pos = atomicAdd(&log->count, 1u);
                if (pos < log->cap) {
                    VerifyRec& o = log->rec[pos];
                    o.seq = seq; o.blk = blk; o.vox = v; o.which = (d01 ? 1u : 0u) | (d02 ? 2u : 0u) | (d12 ? 4u : 0u);
                    o.ex = e.x; o.ey = e.y; o.ez = e.z; o.mask = mask; o.nOps = nOps; o.hwWave = blk % nWavesUpd;
                    for (int c = 0; c < 3; ++c) for (int q = 0; q < 3; ++q) o.val[c][q] = x[c][q];
                }
            }
        }
    }
}

// exact contract: the batch's operators one after the other on the block, each through k_update_col's column form (colFast / colExact) - the voxels travel
// through the wave's own cache lines between operators instead of through registers; same bits as the operators issued one by one
struct alignas(64) BatchUpdOpCol { UpdPose in; float padIn[2]; UpdPose de; float padDe[2]; const float* depth; const uchar4* color; };      // 56-byte poses, one 64-byte line each
struct BatchUpdColArgs { BatchUpdOpCol op[BMAX]; uint32_t nOps; uint32_t pad; };

__global__ __launch_bounds__(256) void k_update_batch_col(Dev d, UpdCam c, BatchUpdColArgs a, int accumulate, int forceExact) {
    const uint32_t n = (uint32_t)d.compactCount[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nWaves = gridDim.x * 4u;
    if (accumulate && wave == 0 && lane == 0) { d.occSum[2] += (unsigned long long)n; d.occSum[0] += (unsigned long long)(uint32_t)d.compactCount[1]; }
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        const int4 e = reinterpret_cast<const int4*>(d.compact)[(size_t)blk * 2];                                    // wave-uniform
        const uint32_t mask = reinterpret_cast<const uint32_t*>(d.compact)[(size_t)blk * 8 + 4];
        for (uint32_t k = 0; k < a.nOps; ++k) {
            const uint32_t m = (mask >> (2u * k)) & 3u;
            const BatchUpdOpCol& o = a.op[k];
            if (m == 0u || o.color == nullptr) continue;      // .cu:441-448: without colour data `color.x != MINF` never holds
            bool fast = !forceExact;
            if (m & 2u) fast = fast && blockFast(c, o.de, e);
            if (m & 1u) fast = fast && blockFast(c, o.in, e);
            if (fast) colFast<true, true>(d, c, o.in, o.de, e, m, lane, o.depth, reinterpret_cast<const uint32_t*>(o.color));
            else colExact<true, true>(d, c, o.in, o.de, e, m, lane, o.depth, o.color);
        }
    }
}
