// Marching-cubes mesh extraction over the voxel hash for gfx950.
// Replaces DepthSensing/CUDAMarchingCubesHashSDF.{h,cpp}, CUDAMarchingCubesSDF.cu, MarchingCubesSDFUtil.h and the part of
// RayCastSDFUtil.h it uses (trilinearInterpolationSimpleFastFast) behind the bf_marching_cubes_* C ABI (paths relative to
// /root/reference/FriedLiver/Source).
//
// It is a CONSUMER of the volume: it reads nothing but the raw arrays bf_scene_get_hash_data() hands out in the reference layout
// (32-byte HashEntry, 12-byte Voxel, ptr = block * 512) plus HashParams — what CUDAMarchingCubesHashSDF::extractIsoSurface reads.
//
// Design:
//  * the reference appends triangles with one global atomic (order differs from run to run); here the occupied hash slots are
//    compacted in slot order, every SDF block counts its triangles, one scan turns the counts into offsets and a second pass writes:
//    triangle order = (hash slot, voxel index z*64 + y*8 + x, table order) on every run;
//  * per cell the arithmetic is the reference's, operation by operation (8 trilinear samples of 8 voxels each, the two threshold
//    gates, vertexInterp);
//  * the case tables are GENERATED (makeTables): on every cube face the cut edges are joined so that the corners below the
//    iso-level are separated (the choice of the published table for ambiguous faces), loops are oriented with those corners to the
//    left seen from outside, and each loop is triangulated as a fan from its lowest edge.  This reproduces the edge table and, for
//    all 256 cases, the same oriented polygon loops as the table the reference ships (Bourke, "Polygonising a scalar field");
//    polygons with more than three vertices may be split along different diagonals — same vertices, same surface patches
//    (tests/test_ref_pin_cpu.py::test_marching_cubes_tables_vs_reference).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "bf_device.h"
#include "bf_internal.h"
#include "bf_volume.h"

using namespace bf;

namespace {

constexpr int BS = BF_SDF_BLOCK_SIZE;
constexpr uint32_t TILE = 1024;

struct McTables { uint16_t edgeMask[256]; int8_t tri[256][16]; uint8_t numTri[256]; };

// ---------------------------------------------------------------------------------------
// case tables
// ---------------------------------------------------------------------------------------
const int CV[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};          // cube corners
const int CE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};  // cube edges
const int CF[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 2, 6, 7}, {0, 3, 7, 4}, {1, 2, 6, 5}};              // faces, corners in cyclic order
const int CFN[6][3] = {{0, 0, -1}, {0, 0, 1}, {0, -1, 0}, {0, 1, 0}, {-1, 0, 0}, {1, 0, 0}};                            // outward normals

int edgeOf(int a, int b) {
    for (int k = 0; k < 12; ++k) if ((CE[k][0] == a && CE[k][1] == b) || (CE[k][0] == b && CE[k][1] == a)) return k;
    return -1;
}

void makeTables(McTables& t) {
    memset(&t, 0, sizeof t);
    for (int cs = 0; cs < 256; ++cs) {
        for (int k = 0; k < 16; ++k) t.tri[cs][k] = -1;
        uint16_t mask = 0;
        for (int k = 0; k < 12; ++k) if (((cs >> CE[k][0]) & 1) != ((cs >> CE[k][1]) & 1)) mask |= (uint16_t)(1u << k);
        t.edgeMask[cs] = mask;
        int next[12]; for (int k = 0; k < 12; ++k) next[k] = -1;
        for (int f = 0; f < 6; ++f) {
            const int* c = CF[f];
            int in[4]; for (int i = 0; i < 4; ++i) in[i] = (cs >> c[i]) & 1;
            int pairs[2][3]; int np = 0;                      // (edge, edge, corner below the iso-level on the inner side)
            int cut[4], nc = 0;
            for (int i = 0; i < 4; ++i) if (in[i] != in[(i + 1) & 3]) cut[nc++] = i;
            if (nc == 2) {
                int inside = -1; for (int i = 0; i < 4; ++i) if (in[i]) { inside = c[i]; break; }
                pairs[np][0] = edgeOf(c[cut[0]], c[(cut[0] + 1) & 3]); pairs[np][1] = edgeOf(c[cut[1]], c[(cut[1] + 1) & 3]); pairs[np][2] = inside; ++np;
            } else if (nc == 4) {                             // ambiguous face: cut off each corner below the iso-level on its own
                for (int i = 0; i < 4; ++i) if (in[i]) { pairs[np][0] = edgeOf(c[(i + 3) & 3], c[i]); pairs[np][1] = edgeOf(c[i], c[(i + 1) & 3]); pairs[np][2] = c[i]; ++np; }
            }
            for (int p = 0; p < np; ++p) {
                const int e1 = pairs[p][0], e2 = pairs[p][1], cc = pairs[p][2];
                double m1[3], m2[3], d[3], q[3];
                for (int k = 0; k < 3; ++k) {
                    m1[k] = 0.5 * (CV[CE[e1][0]][k] + CV[CE[e1][1]][k]); m2[k] = 0.5 * (CV[CE[e2][0]][k] + CV[CE[e2][1]][k]);
                    d[k] = m2[k] - m1[k]; q[k] = CV[cc][k] - m1[k];
                }
                const double cr[3] = {d[1] * q[2] - d[2] * q[1], d[2] * q[0] - d[0] * q[2], d[0] * q[1] - d[1] * q[0]};
                const double s = CFN[f][0] * cr[0] + CFN[f][1] * cr[1] + CFN[f][2] * cr[2];
                if (s > 0) next[e1] = e2; else next[e2] = e1;       // the corner below the iso-level lies to the left, seen from outside
            }
        }
        bool seen[12] = {false};
        int n = 0;
        for (int s0 = 0; s0 < 12; ++s0) {
            if (next[s0] < 0 || seen[s0]) continue;
            int loop[12], len = 0;
            for (int cur = s0; !seen[cur]; cur = next[cur]) { seen[cur] = true; loop[len++] = cur; }      // starts at the loop's lowest edge
            for (int i = 1; i + 1 < len; ++i) { t.tri[cs][n++] = (int8_t)loop[0]; t.tri[cs][n++] = (int8_t)loop[i]; t.tri[cs][n++] = (int8_t)loop[i + 1]; }
        }
        t.numTri[cs] = (uint8_t)(n / 3);
    }
}

// ---------------------------------------------------------------------------------------
// volume access (the reference layout only)
// ---------------------------------------------------------------------------------------
struct McArgs {
    Vol v;
    float thresh, thresh2;
    int boxEnabled; float minCorner[3], maxCorner[3];
    uint32_t maxTriangles;
};

// RayCastData::trilinearInterpolationSimpleFastFast, RayCastSDFUtil.h:97-116 (distance only: marching cubes ignores its colour)
BF_DEV bool trilinear(const Vol& v, f3 pos, float& dist) {
    const float oSet = v.voxelSize;
    const f3 posDual = pos - mk3(oSet / 2.0f, oSet / 2.0f, oSet / 2.0f);
    const f3 pv = pos / v.voxelSize;
    const float wx = fracf_(pv.x), wy = fracf_(pv.y), wz = fracf_(pv.z);
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
BF_DEV Vert vertexInterp(float iso, f3 p1, f3 p2, float d1, float d2, uint32_t col) {      // MarchingCubesSDFUtil.h:231-253 (both ends carry the cell's colour)
    const float c0 = (float)(col & 0xFF), c1 = (float)((col >> 8) & 0xFF), c2 = (float)((col >> 16) & 0xFF);
    Vert r1; r1.p = p1; r1.c = mk3(c0, c1, c2) / 255.0f;
    Vert r2; r2.p = p2; r2.c = r1.c;
    if (fabsf(iso - d1) < 0.00001f) return r1;
    if (fabsf(iso - d2) < 0.00001f) return r2;
    if (fabsf(d1 - d2) < 0.00001f) return r1;
    const float mu = (iso - d1) / (d2 - d1);
    Vert r;
    r.p.x = p1.x + mu * (p2.x - p1.x); r.p.y = p1.y + mu * (p2.y - p1.y); r.p.z = p1.z + mu * (p2.z - p1.z);
    r.c.x = (float)(c0 + mu * (float)(c0 - c0)) / 255.0f; r.c.y = (float)(c1 + mu * (float)(c1 - c1)) / 255.0f; r.c.z = (float)(c2 + mu * (float)(c2 - c2)) / 255.0f;
    return r;
}

// extractIsoSurfaceAtPosition, MarchingCubesSDFUtil.h:118-229.  Returns the number of triangles of the cell at worldPos; EMIT writes them.
template <bool EMIT>
BF_DEV uint32_t cell(const McArgs& a, const McTables* __restrict__ t, f3 worldPos, bf_mc_triangle* out) {
    const Vol& v = a.v;
    if (a.boxEnabled == 1) {
        if (worldPos.x < a.minCorner[0] || worldPos.x > a.maxCorner[0]) return 0;
        if (worldPos.y < a.minCorner[1] || worldPos.y > a.maxCorner[1]) return 0;
        if (worldPos.z < a.minCorner[2] || worldPos.z > a.maxCorner[2]) return 0;
    }
    const float iso = 0.0f;
    const float P = v.voxelSize / 2.0f, M = -P;
    const f3 p000 = worldPos + mk3(M, M, M), p100 = worldPos + mk3(P, M, M), p010 = worldPos + mk3(M, P, M), p001 = worldPos + mk3(M, M, P);
    const f3 p110 = worldPos + mk3(P, P, M), p011 = worldPos + mk3(M, P, P), p101 = worldPos + mk3(P, M, P), p111 = worldPos + mk3(P, P, P);
    float d000, d100, d010, d001, d110, d011, d101, d111;
    // the reference evaluates all eight before testing validity; a sample that fails leaves the cell empty either way
    const bool v000 = trilinear(v, p000, d000), v100 = trilinear(v, p100, d100), v010 = trilinear(v, p010, d010), v001 = trilinear(v, p001, d001);
    const bool v110 = trilinear(v, p110, d110), v011 = trilinear(v, p011, d011), v101 = trilinear(v, p101, d101), v111 = trilinear(v, p111, d111);
    if (!v000 || !v100 || !v010 || !v001 || !v110 || !v011 || !v101 || !v111) return 0;
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
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            if (da[k] * da[l] < 0.0f) { if (fabsf(da[k]) + fabsf(da[l]) > a.thresh) return 0; }
            else { if (fabsf(da[k] - da[l]) > a.thresh) return 0; }
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (fabsf(da[k]) > a.thresh2) return 0;
    const uint32_t em = t->edgeMask[ci];
    if (em == 0 || em == 255) return 0;
    const uint32_t nt = t->numTri[ci];
    if (!EMIT) return nt;
    const Vx own = getVoxel(v, worldPos);
    Vert vl[12];
    if (em & 1) vl[0] = vertexInterp(iso, p010, p110, d010, d110, own.color);
    if (em & 2) vl[1] = vertexInterp(iso, p110, p100, d110, d100, own.color);
    if (em & 4) vl[2] = vertexInterp(iso, p100, p000, d100, d000, own.color);
    if (em & 8) vl[3] = vertexInterp(iso, p000, p010, d000, d010, own.color);
    if (em & 16) vl[4] = vertexInterp(iso, p011, p111, d011, d111, own.color);
    if (em & 32) vl[5] = vertexInterp(iso, p111, p101, d111, d101, own.color);
    if (em & 64) vl[6] = vertexInterp(iso, p101, p001, d101, d001, own.color);
    if (em & 128) vl[7] = vertexInterp(iso, p001, p011, d001, d011, own.color);
    if (em & 256) vl[8] = vertexInterp(iso, p010, p011, d010, d011, own.color);
    if (em & 512) vl[9] = vertexInterp(iso, p110, p111, d110, d111, own.color);
    if (em & 1024) vl[10] = vertexInterp(iso, p100, p101, d100, d101, own.color);
    if (em & 2048) vl[11] = vertexInterp(iso, p000, p001, d000, d001, own.color);
    for (uint32_t i = 0; i < nt; ++i) {
        bf_mc_triangle tr;
        for (int k = 0; k < 3; ++k) {
            const Vert& s = vl[t->tri[ci][3 * i + k]];
            tr.v[k].p[0] = s.p.x; tr.v[k].p[1] = s.p.y; tr.v[k].p[2] = s.p.z; tr.v[k].c[0] = s.c.x; tr.v[k].c[1] = s.c.y; tr.v[k].c[2] = s.c.z;
        }
        out[i] = tr;
    }
    return nt;
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
// occupied hash slots in slot order: per-tile counts, then a scatter with the scanned tile offsets
__global__ __launch_bounds__(256) void k_slots_count(Vol v, uint32_t numSlots, uint32_t* tileCounts) {
    __shared__ uint32_t ws[4];
    const uint32_t tile = blockIdx.x;
    uint32_t c = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint32_t i = tile * TILE + threadIdx.x * 4 + k; if (i < numSlots && v.hash[i].ptr != BF_FREE_ENTRY) ++c; }
    c = (uint32_t)wave_sum_i((int)c);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tileCounts[tile] = ws[0] + ws[1] + ws[2] + ws[3];
}
// single workgroup exclusive scan of n values in place; total -> *total
__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t* a, uint32_t n, uint32_t* total) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < n ? a[i] : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += y; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wtot[w];
        const uint32_t c0 = carry;
        if (i < n) a[i] = c0 + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void k_slots_scatter(Vol v, uint32_t numSlots, const uint32_t* tileOffsets, uint32_t* slots) {
    __shared__ uint32_t wscan[4];
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool occ[4]; uint32_t c = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint32_t i = tile * TILE + threadIdx.x * 4 + k; occ[k] = i < numSlots && v.hash[i].ptr != BF_FREE_ENTRY; c += occ[k] ? 1u : 0u; }
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += y; }
    if (lane == 63) wscan[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; ++w) woff += wscan[w];
    uint32_t pos = tileOffsets[tile] + woff + incl - c;
    for (uint32_t k = 0; k < 4; ++k) if (occ[k]) slots[pos++] = tile * TILE + threadIdx.x * 4 + k;
}

// one 512-thread workgroup per occupied slot (= SDF block), thread = voxel (x fastest): count pass / emit pass
template <bool EMIT>
__global__ __launch_bounds__(512) void k_mc_blocks(McArgs a, const McTables* __restrict__ t, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ numBlocks,
                                                   uint32_t* blockCounts /* COUNT: out totals; EMIT: in exclusive offsets */, bf_mc_triangle* out) {
    __shared__ uint32_t wscan[8];
    const uint32_t nb = numBlocks[0];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const bf_hash_entry e = a.v.hash[slots[b]];
        const uint32_t i = threadIdx.x;
        i3 pi; pi.x = e.pos[0] * BS + (int)(i & 7); pi.y = e.pos[1] * BS + (int)((i >> 3) & 7); pi.z = e.pos[2] * BS + (int)(i >> 6);
        const f3 worldPos = mk3((float)pi.x, (float)pi.y, (float)pi.z) * a.v.voxelSize;      // virtualVoxelPosToWorld :308-310
        bf_mc_triangle tmp[5];
        const uint32_t nt = cell<EMIT>(a, t, worldPos, tmp);
        uint32_t incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += y; }
        if (lane == 63) wscan[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (uint32_t w = 0; w < 8; ++w) { if (w < wave) woff += wscan[w]; tot += wscan[w]; }
        if (!EMIT) { if (threadIdx.x == 0) blockCounts[b] = tot; }
        else {
            uint32_t pos = blockCounts[b] + woff + incl - nt;
            for (uint32_t k = 0; k < nt; ++k, ++pos) if (pos < a.maxTriangles) out[pos] = tmp[k];
        }
        __syncthreads();
    }
}

}  // namespace

// =========================================================================================
// host side: bf_marching_cubes == CUDAMarchingCubesHashSDF
// =========================================================================================
struct bf_marching_cubes {
    bf_marching_cubes_params params;
    hipStream_t stream = nullptr;
    McTables* d_tables = nullptr;
    bf_mc_triangle* d_triangles = nullptr;
    uint32_t *d_tileCounts = nullptr, *d_slots = nullptr, *d_numBlocks = nullptr, *d_blockCounts = nullptr, *d_total = nullptr;
    uint32_t capSlots = 0;
    uint32_t numTriangles = 0;          // of the last extraction (clamped to m_maxNumTriangles)
    uint32_t numTrianglesFound = 0;     // before clamping
    std::vector<bf_mc_triangle> mesh;   // m_meshData: triangles accumulated by copyTrianglesToCPU
};

extern "C" {

int bf_marching_cubes_tables(uint16_t edgeTable[256], int8_t triTable[256 * 16]) {
    BF_REQUIRE(edgeTable && triTable, "null argument");
    McTables t; makeTables(t);
    memcpy(edgeTable, t.edgeMask, sizeof t.edgeMask);
    memcpy(triTable, t.tri, sizeof t.tri);
    return BF_OK;
}

int bf_marching_cubes_create(const bf_marching_cubes_params* p, bf_marching_cubes** out) {      // CUDAMarchingCubesHashSDF::create (.cpp:14-20)
    BF_REQUIRE(p && out && p->m_maxNumTriangles > 0, "bad argument");
    BF_REQUIRE(p->m_sdfBlockSize == BF_SDF_BLOCK_SIZE && p->m_hashBucketSize == BF_HASH_BUCKET_SIZE, "block size 8 / bucket size 4 expected");
    bf_marching_cubes* m = new bf_marching_cubes;
    m->params = *p;
    McTables t; makeTables(t);
    BF_HIP_TRY(BF_MALLOC((void**)&m->d_tables, sizeof t));
    BF_HIP_TRY(hipMemcpy(m->d_tables, &t, sizeof t, hipMemcpyHostToDevice));
    BF_HIP_TRY(BF_MALLOC((void**)&m->d_triangles, sizeof(bf_mc_triangle) * (size_t)p->m_maxNumTriangles));
    BF_HIP_TRY(BF_MALLOC((void**)&m->d_numBlocks, 4));
    BF_HIP_TRY(BF_MALLOC((void**)&m->d_total, 4));
    *out = m;
    return BF_OK;
}

int bf_marching_cubes_destroy(bf_marching_cubes* m) {
    if (!m) return BF_OK;
    (void)hipDeviceSynchronize();
    (void)hipFree(m->d_tables); (void)hipFree(m->d_triangles); (void)hipFree(m->d_numBlocks); (void)hipFree(m->d_total);
    (void)hipFree(m->d_tileCounts); (void)hipFree(m->d_slots); (void)hipFree(m->d_blockCounts);
    delete m;
    return BF_OK;
}

int bf_marching_cubes_set_stream(bf_marching_cubes* m, void* s) { BF_REQUIRE(m, "null argument"); m->stream = (hipStream_t)s; return BF_OK; }

// extractIsoSurface(hashData, hashParams, rayCastData, minCorner, maxCorner, boxEnabled)  .cpp:107-119 (+ copyTrianglesToCPU :27-46)
int bf_marching_cubes_extract(bf_marching_cubes* m, const bf_hash_data* hd, const bf_hash_params* hp, const float minCorner[3], const float maxCorner[3], int boxEnabled) {
    BF_REQUIRE(m && hd && hp && hd->d_hash && hd->d_SDFBlocks, "null argument");
    BF_REQUIRE(hp->m_hashBucketSize == BF_HASH_BUCKET_SIZE && hp->m_SDFBlockSize == BF_SDF_BLOCK_SIZE, "block size 8 / bucket size 4 expected");
    const uint32_t numSlots = hp->m_hashNumBuckets * BF_HASH_BUCKET_SIZE;
    const uint32_t numTiles = div_up(numSlots, TILE);
    if (numSlots > m->capSlots) {
        (void)hipFree(m->d_tileCounts); (void)hipFree(m->d_slots); (void)hipFree(m->d_blockCounts);
        BF_HIP_TRY(BF_MALLOC((void**)&m->d_tileCounts, 4 * (size_t)(numTiles + 1)));
        BF_HIP_TRY(BF_MALLOC((void**)&m->d_slots, 4 * (size_t)numSlots));
        BF_HIP_TRY(BF_MALLOC((void**)&m->d_blockCounts, 4 * (size_t)numSlots));
        m->capSlots = numSlots;
    }
    McArgs a;
    a.v.hash = (const bf_hash_entry*)hd->d_hash; a.v.vox = (const bf_voxel*)hd->d_SDFBlocks;
    a.v.numBuckets = hp->m_hashNumBuckets; a.v.maxChain = hp->m_hashMaxCollisionLinkedListSize; a.v.voxelSize = hp->m_virtualVoxelSize;
    a.thresh = m->params.m_threshMarchingCubes; a.thresh2 = m->params.m_threshMarchingCubes2;
    a.boxEnabled = boxEnabled ? 1 : 0;
    for (int k = 0; k < 3; ++k) { a.minCorner[k] = minCorner ? minCorner[k] : 0.0f; a.maxCorner[k] = maxCorner ? maxCorner[k] : 0.0f; }
    a.maxTriangles = m->params.m_maxNumTriangles;
    hipStream_t st = m->stream;
    hipLaunchKernelGGL(k_slots_count, dim3(numTiles), dim3(256), 0, st, a.v, numSlots, m->d_tileCounts);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, m->d_tileCounts, numTiles, m->d_numBlocks);
    hipLaunchKernelGGL(k_slots_scatter, dim3(numTiles), dim3(256), 0, st, a.v, numSlots, (const uint32_t*)m->d_tileCounts, m->d_slots);
    uint32_t nb = 0;
    BF_HIP_TRY(hipMemcpyAsync(&nb, m->d_numBlocks, 4, hipMemcpyDeviceToHost, st));
    BF_HIP_TRY(hipStreamSynchronize(st));
    m->numTriangles = m->numTrianglesFound = 0;
    if (nb == 0) return BF_OK;
    const uint32_t grid = std::min(nb, 8192u);
    hipLaunchKernelGGL(k_mc_blocks<false>, dim3(grid), dim3(512), 0, st, a, (const McTables*)m->d_tables, (const uint32_t*)m->d_slots, (const uint32_t*)m->d_numBlocks,
                       m->d_blockCounts, (bf_mc_triangle*)nullptr);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, m->d_blockCounts, nb, m->d_total);
    hipLaunchKernelGGL(k_mc_blocks<true>, dim3(grid), dim3(512), 0, st, a, (const McTables*)m->d_tables, (const uint32_t*)m->d_slots, (const uint32_t*)m->d_numBlocks,
                       m->d_blockCounts, m->d_triangles);
    BF_HIP_TRY(hipGetLastError());
    uint32_t total = 0;
    BF_HIP_TRY(hipMemcpyAsync(&total, m->d_total, 4, hipMemcpyDeviceToHost, st));
    BF_HIP_TRY(hipStreamSynchronize(st));
    m->numTrianglesFound = total;
    m->numTriangles = std::min(total, m->params.m_maxNumTriangles);          // like appendTriangle (:262-284): the buffer's capacity bounds the mesh
    const size_t base = m->mesh.size();                                       // copyTrianglesToCPU: append to m_meshData
    m->mesh.resize(base + m->numTriangles);
    if (m->numTriangles) BF_HIP_TRY(hipMemcpy(m->mesh.data() + base, m->d_triangles, sizeof(bf_mc_triangle) * (size_t)m->numTriangles, hipMemcpyDeviceToHost));
    return BF_OK;
}

int bf_marching_cubes_get_triangles_gpu(bf_marching_cubes* m, const bf_mc_triangle** d_triangles, uint32_t* numTriangles, uint32_t* numFound) {
    BF_REQUIRE(m, "null argument");
    if (d_triangles) *d_triangles = m->d_triangles;
    if (numTriangles) *numTriangles = m->numTriangles;
    if (numFound) *numFound = m->numTrianglesFound;
    return BF_OK;
}

int bf_marching_cubes_get_mesh(bf_marching_cubes* m, bf_mc_triangle* h_out, uint32_t capacity, uint32_t* count) {
    BF_REQUIRE(m && count, "null argument");
    *count = (uint32_t)m->mesh.size();
    if (h_out) memcpy(h_out, m->mesh.data(), sizeof(bf_mc_triangle) * std::min<size_t>(capacity, m->mesh.size()));
    return BF_OK;
}

int bf_marching_cubes_clear_mesh_buffer(bf_marching_cubes* m) { BF_REQUIRE(m, "null argument"); m->mesh.clear(); return BF_OK; }

// saveMesh (.cpp:48-105): vertices closer than 1e-5 are merged, duplicate and degenerate faces dropped, an optional rigid transform is
// applied, and the mesh is written as a binary little-endian PLY with per-vertex colours.  (mLib's MeshData / MeshIO are not in the
// tree; the merge is a hash grid of cell 1e-5 with the first vertex of a cell as its representative.)
int bf_marching_cubes_save_mesh(bf_marching_cubes* m, const char* filename, const float transform[16], uint32_t* numVertices, uint32_t* numFaces) {
    BF_REQUIRE(m && filename, "null argument");
    struct Key { int64_t x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
    struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.x * 73856093ll ^ k.y * 19349669ll ^ k.z * 83492791ll); } };
    std::unordered_map<Key, uint32_t, KeyHash> grid;
    std::vector<float> pos, col;
    std::vector<uint32_t> faces;
    const double inv = 1.0 / 0.00001;
    auto vertex = [&](const bf_mc_vertex& v) {
        const Key k = {(int64_t)std::floor(v.p[0] * inv + 0.5), (int64_t)std::floor(v.p[1] * inv + 0.5), (int64_t)std::floor(v.p[2] * inv + 0.5)};
        auto it = grid.find(k);
        if (it != grid.end()) return it->second;
        const uint32_t id = (uint32_t)(pos.size() / 3);
        grid.emplace(k, id);
        for (int c = 0; c < 3; ++c) { pos.push_back(v.p[c]); col.push_back(v.c[c]); }
        return id;
    };
    struct Face { uint32_t a, b, c; };
    // duplicate faces are keyed on the full sorted vertex triple (96 bits): a packed 64-bit key with 21-bit fields would let distinct
    // faces of a mesh with more than 2^21 vertices collide, and the later one would be dropped as a "duplicate" (a hole)
    struct Tri { uint32_t v[3]; bool operator==(const Tri& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; } };
    struct TriHash { size_t operator()(const Tri& t) const { uint64_t h = t.v[0] * 0x9E3779B97F4A7C15ull; h = (h ^ t.v[1]) * 0xC2B2AE3D27D4EB4Full; h = (h ^ t.v[2]) * 0x165667B19E3779F9ull; return (size_t)(h ^ (h >> 29)); } };
    auto canon = [](Face f) { Tri t = {{f.a, f.b, f.c}}; std::sort(t.v, t.v + 3); return t; };
    std::unordered_set<Tri, TriHash> seen;
    for (const bf_mc_triangle& t : m->mesh) {
        const Face f = {vertex(t.v[0]), vertex(t.v[1]), vertex(t.v[2])};
        if (f.a == f.b || f.b == f.c || f.a == f.c) continue;
        if (!seen.insert(canon(f)).second) continue;
        faces.push_back(f.a); faces.push_back(f.b); faces.push_back(f.c);
    }
    if (transform) {
        m44 T; memcpy(T.e, transform, 64);
        for (size_t i = 0; i < pos.size(); i += 3) { const f3 q = xform(T, mk3(pos[i], pos[i + 1], pos[i + 2])); pos[i] = q.x; pos[i + 1] = q.y; pos[i + 2] = q.z; }
    }
    FILE* f = fopen(filename, "wb");
    if (!f) { set_error("cannot open %s", filename); return BF_ERR_INVALID_ARG; }
    const uint32_t nv = (uint32_t)(pos.size() / 3), nf = (uint32_t)(faces.size() / 3);
    fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment bundlefusion_amd marching cubes\nelement vertex %u\nproperty float x\nproperty float y\nproperty float z\n"
               "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nelement face %u\nproperty list uchar int vertex_indices\nend_header\n", nv, nf);
    for (uint32_t i = 0; i < nv; ++i) {
        fwrite(&pos[3 * i], 4, 3, f);
        uint8_t c[4];
        for (int k = 0; k < 3; ++k) c[k] = (uint8_t)std::max(0.0f, std::min(255.0f, col[3 * i + k] * 255.0f + 0.5f));
        c[3] = 255;
        fwrite(c, 1, 4, f);
    }
    for (uint32_t i = 0; i < nf; ++i) { const uint8_t three = 3; fwrite(&three, 1, 1, f); int32_t idx[3] = {(int32_t)faces[3 * i], (int32_t)faces[3 * i + 1], (int32_t)faces[3 * i + 2]}; fwrite(idx, 4, 3, f); }
    fclose(f);
    if (numVertices) *numVertices = nv;
    if (numFaces) *numFaces = nf;
    m->mesh.clear();                                                            // clearMeshBuffer() at the end of saveMesh
    return BF_OK;
}

}  // extern "C"
