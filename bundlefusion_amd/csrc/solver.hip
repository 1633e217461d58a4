// Bundling solver for gfx950: Gauss-Newton over SE(3) poses with a sparse 3-D point term and a dense
// depth(+colour) term, Jacobi-preconditioned CG.  Replaces Solver/CUDASolverBundling.{h,cpp} +
// Solver/SolverBundling.cu + SBA.cu of the reference (paths relative to
// /root/reference/FriedLiver/Source) behind the bf_solver_* C ABI.
//
// Design:
//  * the reference re-applies J and J^T per correspondence in every PCG iteration (6-7 launches, a
//    memset and a blocking D2H per iteration, 450 iterations per global solve).  Here the Jacobians —
//    which are constant during the linear solve — are contracted ONCE per Gauss-Newton iteration into
//    a block-sparse normal matrix: one 6x6 block pair (D_ij = sum J_i^T J_i, O_ij = sum J_i^T J_j) and
//    a 6-vector per DIRECTED image pair that shares correspondences or dense overlap, in CSR order;
//  * every sum is a gather in a fixed order (correspondence index / slot order): no float atomics,
//    run-to-run deterministic, identical on every rank of a multi-GPU job;
//  * the dense term's per-pair 13x13 Gram [J_i | J_j | r]^T W [J_i | J_j | r] over the 80x60 pixels is
//    the one genuine dense contraction: v_mfma_f32_16x16x4_f32 (exact f32 fma chain);
//  * the whole PCG loop (init, <=nLin iterations, the 5e-7 early-out, Lie update, GN convergence test)
//    is ONE persistent single-workgroup kernel: no host round trip inside a solve.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "bf_device.h"
#include "bf_internal.h"
#include "bf_se3.h"

using namespace bf;

namespace {

constexpr float FLOAT_EPSILON = 0.000001f;       // SolverUtil.h:9
constexpr uint32_t NOSLOT = 0xFFFFFFFFu;
constexpr int DENSE_BLK = 120;                   // ii(36) jj(36) ij(36) gi(6) gj(6)

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Flag { FL_DONE = 0, FL_NUM_SLOTS = 1, FL_NUM_PAIRS = 2, FL_LIST_LEN = 3, FL_GN_ITERS = 4, FL_BARRIER_FAIL = 5, FL_PCG_ITERS = 8, FL_COUNT = 40 };

struct Cfg {
    float denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax;
    uint32_t subsample;
    uint32_t W, H;
    float fx, fy, cx, cy;
    float wSparse, wDepth, wColor;
    int usePairwise;
};

struct Dev {
    const bf_entry_j* corr; uint32_t C;
    const int* valid; uint32_t N;
    float* xRot; float* xTrans;
    m44* T; m44* Tinv;
    uint32_t *keyCount, *keyStart, *cursor, *slotOfKey, *denseRaw, *keyPair;
    uint32_t *rowStart, *slotCol, *slotKey, *corrList;
    float *slotD, *slotO, *slotG, *slotP;
    float *diagA, *rhs, *prec;
    float *delta, *r, *p, *Ap;
    uint2* densePairs; float* denseWeight; float* denseBlocks;
    const bf_cached_frame* cache;
    int* flags;
    int* gridBar;                    // one grid-barrier counter per Gauss-Newton iteration (k_pcg_coop)
    float* energies;
    float* maxRes; int* maxIdx; int* highCount;
    int* numEntriesPerRow;
    float* coopVec;                  // k_pcg_coop with vecGlobal: per-workgroup vectors (problems too large for LDS)
    uint32_t* scanRow;               // k_scan_*: three sums (then exclusive bases) per row of the key table
    float* maxResPart; int* maxIdxPart;   // k_max_residual: one (value, index) per workgroup
    uint32_t maxSlots, maxPairs;
};

BF_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
BF_DEV bool validCorr(const bf_entry_j& c) { return c.imgIdx_i != 0xFFFFFFFFu; }

__global__ void k_solve_begin(Dev d) {          // flags and the grid barrier's words cleared (one launch instead of two memsets)
    for (uint32_t k = threadIdx.x; k < (uint32_t)FL_COUNT; k += blockDim.x) d.flags[k] = 0;
    for (uint32_t k = threadIdx.x; k < 32u; k += blockDim.x) d.gridBar[k] = 0;
}

// ------------------------------------------------------------------ poses -> matrices (:1114-1121)
// ... and the iteration's structure counters cleared in the same launch (four hipMemsetAsync per Gauss-Newton iteration until round 6: the chunk's solve job is
// ~120 dependent launches, and what it costs is launches, not kernel time - profiles/r06_loop_schedule.md)
__global__ void k_poses(Dev d, uint32_t numKeys, int useDense) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = i; k < numKeys; k += gridDim.x * blockDim.x) {
        d.keyCount[k] = 0u; d.cursor[k] = 0u;
        if (useDense) { d.denseRaw[k] = 0u; d.keyPair[k] = 0u; }
    }
    if (d.flags[FL_DONE]) return;
    if (i >= d.N) return;
    const m44 M = poseToMatrix(ld3(d.xRot + 3 * i), ld3(d.xTrans + 3 * i));
    d.T[i] = M;
    d.Tinv[i] = inverse44(M);
}

// ------------------------------------------------------------------ structure: directed pair keys
__global__ void k_key_count(Dev d) {
    if (d.flags[FL_DONE]) return;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    const bf_entry_j e = d.corr[c];
    if (!validCorr(e) || e.imgIdx_i >= d.N || e.imgIdx_j >= d.N) return;
    atomicAdd(&d.keyCount[e.imgIdx_i * d.N + e.imgIdx_j], 1u);     // integer atomics: order-independent
    atomicAdd(&d.keyCount[e.imgIdx_j * d.N + e.imgIdx_i], 1u);
}

// d_numEntriesPerRow of the reference's variable -> correspondence table (SolverBundling.cu:1226-1248): the number of
// valid correspondences that touch image i == the sum of row i of the directed key counts
__global__ void k_row_entries(Dev d) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    uint32_t n = 0;
    for (uint32_t j = 0; j < d.N; ++j) n += d.keyCount[i * d.N + j];
    d.numEntriesPerRow[i] = (int)n;
}

// Three-channel exclusive scan over the N*N directed keys (key order = CSR order): list length, slots, dense pairs.  Integer sums, so
// any evaluation order gives the same table; this one reads the key tables coalesced and scales with the number of key frames
// (the single-workgroup version walked N*N / 1024 keys per thread with a stride between lanes: 383 us at N = 450 key frames,
// three times per global solve - profiles/r03_5000_frame_stream.md):
//   k_scan_rows  one wave per row i: the row's three sums
//   k_scan_base  one workgroup: exclusive scan over the N rows, totals -> flags
//   k_scan_fill  one wave per row: 64 keys per step, wave-level exclusive scan on top of the row's base
BF_DEV void scanKey(const Dev& d, int useDense, uint32_t N, uint32_t i, uint32_t j, uint32_t& cnt, uint32_t& dn) {
    cnt = d.keyCount[i * N + j];
    dn = (useDense && i != j) ? d.denseRaw[min(i, j) * N + max(i, j)] : 0u;
}

__global__ __launch_bounds__(64) void k_scan_rows(Dev d, int useDense) {
    if (d.flags[FL_DONE]) return;
    const uint32_t N = d.N, i = blockIdx.x, lane = threadIdx.x;
    uint32_t a = 0, b = 0, c = 0;
    for (uint32_t j = lane; j < N; j += 64) {
        uint32_t cnt, dn;
        scanKey(d, useDense, N, i, j, cnt, dn);
        a += cnt; b += (cnt > 0 || dn) ? 1u : 0u; c += (dn && i < j) ? 1u : 0u;
    }
    a = (uint32_t)wave_sum_i((int)a); b = (uint32_t)wave_sum_i((int)b); c = (uint32_t)wave_sum_i((int)c);
    if (lane == 0) { d.scanRow[i * 3 + 0] = a; d.scanRow[i * 3 + 1] = b; d.scanRow[i * 3 + 2] = c; }
}

__global__ __launch_bounds__(1024) void k_scan_base(Dev d) {
    if (d.flags[FL_DONE]) return;
    __shared__ uint32_t sA[1024], sB[1024], sC[1024];
    const uint32_t N = d.N;
    const uint32_t chunk = (N + blockDim.x - 1) / blockDim.x;
    const uint32_t r0 = min(threadIdx.x * chunk, N), r1 = min(r0 + chunk, N);
    uint32_t a = 0, b = 0, c = 0;
    for (uint32_t r = r0; r < r1; ++r) { a += d.scanRow[r * 3]; b += d.scanRow[r * 3 + 1]; c += d.scanRow[r * 3 + 2]; }
    sA[threadIdx.x] = a; sB[threadIdx.x] = b; sC[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
        uint32_t ta = 0, tb = 0, tc = 0;
        if (threadIdx.x >= off) { ta = sA[threadIdx.x - off]; tb = sB[threadIdx.x - off]; tc = sC[threadIdx.x - off]; }
        __syncthreads();
        sA[threadIdx.x] += ta; sB[threadIdx.x] += tb; sC[threadIdx.x] += tc;
        __syncthreads();
    }
    uint32_t ra = sA[threadIdx.x] - a, rb = sB[threadIdx.x] - b, rc = sC[threadIdx.x] - c;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t ta = d.scanRow[r * 3], tb = d.scanRow[r * 3 + 1], tc = d.scanRow[r * 3 + 2];
        d.scanRow[r * 3] = ra; d.scanRow[r * 3 + 1] = rb; d.scanRow[r * 3 + 2] = rc;      // exclusive bases of row r
        d.rowStart[r] = rb;
        ra += ta; rb += tb; rc += tc;
    }
    if (threadIdx.x == blockDim.x - 1) {
        d.rowStart[N] = min(rb, d.maxSlots);
        d.flags[FL_LIST_LEN] = (int)ra;
        d.flags[FL_NUM_SLOTS] = (int)min(rb, d.maxSlots);
        d.flags[FL_NUM_PAIRS] = (int)min(rc, d.maxPairs);
    }
}

BF_DEV uint32_t waveExScan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if ((int)lane >= o) incl += t; }
    total = (uint32_t)__shfl((int)incl, 63, 64);
    return incl - v;
}

__global__ __launch_bounds__(64) void k_scan_fill(Dev d, int useDense) {
    if (d.flags[FL_DONE]) return;
    const uint32_t N = d.N, i = blockIdx.x, lane = threadIdx.x;
    uint32_t ra = d.scanRow[i * 3], rb = d.scanRow[i * 3 + 1], rc = d.scanRow[i * 3 + 2];
    for (uint32_t j0 = 0; j0 < N; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t cnt = 0, dn = 0;
        if (j < N) scanKey(d, useDense, N, i, j, cnt, dn);
        const uint32_t hasSlot = (j < N && (cnt > 0 || dn)) ? 1u : 0u, hasPair = (j < N && dn && i < j) ? 1u : 0u;
        uint32_t ta, tb, tc;
        const uint32_t ea = waveExScan(cnt, lane, ta), eb = waveExScan(hasSlot, lane, tb), ec = waveExScan(hasPair, lane, tc);
        if (j < N) {
            const uint32_t k = i * N + j;
            d.keyStart[k] = ra + ea;
            if (hasSlot) {
                const uint32_t sl = rb + eb;
                if (sl < d.maxSlots) { d.slotOfKey[k] = sl; d.slotCol[sl] = j; d.slotKey[sl] = k; }
            } else d.slotOfKey[k] = NOSLOT;
            if (hasPair) {
                const uint32_t pr = rc + ec;
                if (pr < d.maxPairs) { d.densePairs[pr] = make_uint2(i, j); d.keyPair[k] = pr + 1; d.keyPair[j * N + i] = pr + 1; }
            }
        }
        ra += ta; rb += tb; rc += tc;
    }
}

__global__ void k_fill(Dev d) {
    if (d.flags[FL_DONE]) return;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    const bf_entry_j e = d.corr[c];
    if (!validCorr(e) || e.imgIdx_i >= d.N || e.imgIdx_j >= d.N) return;
    const uint32_t ka = e.imgIdx_i * d.N + e.imgIdx_j, kb = e.imgIdx_j * d.N + e.imgIdx_i;
    d.corrList[d.keyStart[ka] + atomicAdd(&d.cursor[ka], 1u)] = (c << 1);        // role A: row image == imgIdx_i
    d.corrList[d.keyStart[kb] + atomicAdd(&d.cursor[kb], 1u)] = (c << 1) | 1u;   // role B: row image == imgIdx_j
}

// ------------------------------------------------------------------ dense term
BF_DEV f3 depthToCamera(const Cfg& c, int x, int y, float dep) {           // CUDACameraUtil.h:16-20
    const float kx = ((float)x - c.cx) / c.fx, ky = ((float)y - c.cy) / c.fy;
    return mk3(dep * kx, dep * ky, dep);
}
BF_DEV void cameraToDepth(const Cfg& c, f3 p, float& u, float& v) { u = p.x * c.fx / p.z + c.cx; v = p.y * c.fy / p.z + c.cy; }

template <int K>
BF_DEV bool bilinear(float x, float y, const float* __restrict__ in, uint32_t W, uint32_t H, float* out) {   // ICPUtil.h:28-111
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const float alpha = x - (float)x0, beta = y - (float)y0;
    float s0[K], s1[K], w0 = 0.0f, w1 = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) { s0[k] = 0.0f; s1[k] = 0.0f; }
#define BF_TAP(PX, PY, WGT, S, WS)                                                          \
    if ((uint32_t)(PX) < W && (uint32_t)(PY) < H) {                                         \
        const float* v = in + (size_t)K * ((size_t)(PY) * W + (PX));                        \
        if (v[0] != BF_MINF) { _Pragma("unroll") for (int k = 0; k < K; ++k) S[k] += (WGT) * v[k]; WS += (WGT); } \
    }
    BF_TAP(x0, y0, 1.0f - alpha, s0, w0)
    BF_TAP(x0 + 1, y0, alpha, s0, w0)
    BF_TAP(x0, y0 + 1, 1.0f - alpha, s1, w1)
    BF_TAP(x0 + 1, y0 + 1, alpha, s1, w1)
#undef BF_TAP
    float ss[K], ww = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) ss[k] = 0.0f;
    if (w0 > 0.0f) { _Pragma("unroll") for (int k = 0; k < K; ++k) ss[k] += (1.0f - beta) * (s0[k] / w0); ww += (1.0f - beta); }
    if (w1 > 0.0f) { _Pragma("unroll") for (int k = 0; k < K; ++k) ss[k] += beta * (s1[k] / w1); ww += beta; }
    if (ww > 0.0f) { _Pragma("unroll") for (int k = 0; k < K; ++k) out[k] = ss[k] / ww; return true; }
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = BF_MINF;
    return false;
}

BF_DEV bool angleBelow(const m44& tr, float thresh) {                        // DenseUtil :416-424
    const float il = 1.0f / sqrtf(3.0f);
    const f3 x = mk3(1.0f * il, 1.0f * il, 1.0f * il);
    const f3 v = rot(tr, x);
    const float c = fminf(fmaxf(dot3(x, v), -1.0f), 1.0f);
    return fabsf(acosf(c)) < thresh;
}

// FindImageImageCorr_Kernel (SolverBundling.cu:30-79): one workgroup per candidate pair
__global__ __launch_bounds__(512) void k_dense_overlap(Dev d, Cfg c) {
    if (d.flags[FL_DONE]) return;
    uint32_t i, j;
    if (c.usePairwise) { i = blockIdx.x; j = blockIdx.y; if (i >= j) return; }
    else { i = blockIdx.x; j = i + 1; }
    if (j >= d.N || d.valid[i] == 0 || d.valid[j] == 0) return;
    const m44 tr = mul44(d.Tinv[i], d.T[j]);
    if (!angleBelow(tr, 0.52f)) return;
    __shared__ int found;
    if (threadIdx.x == 0) found = 0;
    __syncthreads();
    const uint32_t subW = c.W / c.subsample;
    const uint32_t x = (threadIdx.x % subW) * c.subsample, y = (threadIdx.x / subW) * c.subsample;
    const uint32_t idx = y * c.W + x;
    bool ok = false;
    if (idx < c.W * c.H) {                                                   // DenseUtil :22-42
        const float* srcDepth = d.cache[j].d_depthDownsampled;
        const float* tgtDepth = d.cache[i].d_depthDownsampled;
        const f3 cp = depthToCamera(c, (int)x, (int)y, srcDepth[idx]);
        if (cp.z > c.denseDepthMin && cp.z < c.denseDepthMax) {
            const f3 q = xform(tr, cp);
            float u, v;
            cameraToDepth(c, q, u, v);
            const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
            if (tx >= 0 && ty >= 0 && tx < (int)c.W && ty < (int)c.H) {
                const f3 ct = depthToCamera(c, tx, ty, tgtDepth[ty * c.W + tx]);
                if (ct.z > c.denseDepthMin && ct.z < c.denseDepthMax && len3(q - ct) <= c.denseDistThresh) ok = true;
            }
        }
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&found, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && found > 10) d.denseRaw[i * d.N + j] = 1u;
}

// FindDenseCorrespondences_Kernel + WeightDenseCorrespondences_Kernel (:92-180): one workgroup per pair
__global__ __launch_bounds__(256) void k_dense_weight(Dev d, Cfg c) {
    if (d.flags[FL_DONE]) return;
    __shared__ int wcount[4];
    const uint32_t nPairs = (uint32_t)d.flags[FL_NUM_PAIRS];
    const uint32_t npix = c.W * c.H;
    for (uint32_t p = blockIdx.x; p < nPairs; p += gridDim.x) {
        const uint2 pr = d.densePairs[p];
        const m44 tr = mul44(d.Tinv[pr.x], d.T[pr.y]);
        const bf_cached_frame tgt = d.cache[pr.x], src = d.cache[pr.y];
        int count = 0;
        for (uint32_t idx = threadIdx.x; idx < npix; idx += blockDim.x) {     // DenseUtil :152-184 (uchar4 normals)
            const uint32_t x = idx % c.W, y = idx / c.W;
            const f3 cp = depthToCamera(c, (int)x, (int)y, src.d_depthDownsampled[idx]);
            if (!(cp.z > c.denseDepthMin && cp.z < c.denseDepthMax)) continue;
            const uchar4 nu = reinterpret_cast<const uchar4*>(src.d_normalsDownsampledUCHAR4)[idx];
            if (!(nu.x | nu.y | nu.z | nu.w)) continue;
            f3 nj = mk3((float)nu.x / 255.0f * 2.0f - 1.0f, (float)nu.y / 255.0f * 2.0f - 1.0f, (float)nu.z / 255.0f * 2.0f - 1.0f);
            nj = rot(tr, nj);
            const f3 q = xform(tr, cp);
            float u, v;
            cameraToDepth(c, q, u, v);
            const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
            if (!(tx >= 0 && ty >= 0 && tx < (int)c.W && ty < (int)c.H)) continue;
            const f3 ct = depthToCamera(c, tx, ty, tgt.d_depthDownsampled[ty * c.W + tx]);
            if (!(ct.z > c.denseDepthMin && ct.z < c.denseDepthMax)) continue;
            const uchar4 nt = reinterpret_cast<const uchar4*>(tgt.d_normalsDownsampledUCHAR4)[ty * c.W + tx];
            if (!(nt.x | nt.y | nt.z | nt.w)) continue;
            const f3 nT = mk3((float)nt.x / 255.0f * 2.0f - 1.0f, (float)nt.y / 255.0f * 2.0f - 1.0f, (float)nt.z / 255.0f * 2.0f - 1.0f);
            if (dot3(nj, nT) >= c.denseNormalThresh && len3(q - ct) <= c.denseDistThresh) count++;
        }
        count = wave_sum_i(count);
        if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = count;
        __syncthreads();
        if (threadIdx.x == 0) {
            float x = (float)(wcount[0] + wcount[1] + wcount[2] + wcount[3]);
            if (x > 0) { if (x < 800) x = 0; else x = 1.0f / fminf(logf(x), 9.0f); }
            d.denseWeight[p] = x;
        }
        __syncthreads();
    }
}

// 3x6 Jacobians of the source point in the target frame (LieDerivUtil.h:247-295), order [t | omega]
BF_DEV void derivI(const m44& A, const m44& D, f3 p, float jac[3][6]) {
    const m44 tr = mul44(A, D);
    const float pt[3] = {p.x - tr.e[3], p.y - tr.e[7], p.z - tr.e[11]};
    // j1 rows 3k..3k+2, cols 3..5 :  -(R_A * skew(D col k));  rows 9..11, cols 0..2 : R_A
    float m[4][3][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float vx = D.e[k], vy = D.e[4 + k], vz = D.e[8 + k];
        const float sk[3][3] = {{0.0f, -vz, vy}, {vz, 0.0f, -vx}, {-vy, vx, 0.0f}};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                m[k][r][cc] = (A.e[r * 4 + 0] * sk[0][cc] + A.e[r * 4 + 1] * sk[1][cc] + A.e[r * 4 + 2] * sk[2][cc]) * -1.0f;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            // translation columns: only j1 rows 9..11 are non-zero
            jac[r][cc] = (-tr.e[0 * 4 + r]) * A.e[0 * 4 + cc] + (-tr.e[1 * 4 + r]) * A.e[1 * 4 + cc] + (-tr.e[2 * 4 + r]) * A.e[2 * 4 + cc];
            // rotation columns: j0 row r is [.. pt at 3r..3r+2 .. | -R^T at 9..11]
            float acc = pt[0] * m[r][0][cc];
            acc += pt[1] * m[r][1][cc];
            acc += pt[2] * m[r][2][cc];
            acc += (-tr.e[0 * 4 + r]) * m[3][0][cc];
            acc += (-tr.e[1 * 4 + r]) * m[3][1][cc];
            acc += (-tr.e[2 * 4 + r]) * m[3][2][cc];
            jac[r][3 + cc] = acc;
        }
    }
}
BF_DEV void derivJ(const m44& A, const m44& D, f3 p, float jac[3][6]) {
    const float a = dot3(p, mk3(D.e[0], D.e[1], D.e[2])) + D.e[3];
    const float b = dot3(p, mk3(D.e[4], D.e[5], D.e[6])) + D.e[7];
    const float c = dot3(p, mk3(D.e[8], D.e[9], D.e[10])) + D.e[11];
    const float j[3][6] = {{1, 0, 0, 0.0f, c, -b}, {0, 1, 0, -c, 0.0f, a}, {0, 0, 1, b, -a, 0.0f}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) jac[r][cc] = A.e[r * 4 + 0] * j[0][cc] + A.e[r * 4 + 1] * j[1][cc] + A.e[r * 4 + 2] * j[2][cc];
}

// BuildDenseSystem_Kernel (:182-306): per pair the 13x13 Gram of rows [J_i | J_j | r] weighted by w.
// 4 waves per pair, each wave: 64 pixels -> LDS transpose -> 16 MFMAs (4 pixel rows per MFMA).
__global__ __launch_bounds__(256) void k_dense_build(Dev d, Cfg c) {
    if (d.flags[FL_DONE]) return;
    __shared__ float tileX[4][64][17];
    __shared__ float tileW[4][64];
    __shared__ float part[4][256];
    const uint32_t nPairs = (uint32_t)d.flags[FL_NUM_PAIRS];
    const uint32_t npix = c.W * c.H;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool useDepth = c.wDepth > 0.0f, useColor = c.wColor > 0.0f;
    for (uint32_t p = blockIdx.x; p < nPairs; p += gridDim.x) {
        const float pw = d.denseWeight[p];
        float* out = d.denseBlocks + (size_t)p * DENSE_BLK;
        if (pw == 0.0f) {
            if (threadIdx.x < DENSE_BLK) out[threadIdx.x] = 0.0f;
            continue;
        }
        const uint2 pr = d.densePairs[p];
        const uint32_t i = pr.x, j = pr.y;
        const m44 Ti = d.T[i], Tj = d.T[j], TiI = d.Tinv[i], TjI = d.Tinv[j];
        const m44 tr = mul44(TiI, Tj);
        const bf_cached_frame tgt = d.cache[i], src = d.cache[j];
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t base = 0; base < npix; base += 256) {      // same trip count for all 4 waves
            const uint32_t idx = base + threadIdx.x;
            float Xd[13], Xc[13], wd = 0.0f, wc = 0.0f;
#pragma unroll
            for (int k = 0; k < 13; ++k) { Xd[k] = 0.0f; Xc[k] = 0.0f; }
            if (idx < npix) {                                                // findDenseCorr, DenseUtil :79-113
                const float4 cp4 = reinterpret_cast<const float4*>(src.d_cameraposDownsampled)[idx];
                if (cp4.z > c.denseDepthMin && cp4.z < c.denseDepthMax) {
                    const f3 cs = mk3(cp4.x, cp4.y, cp4.z);
                    const float4 nj = reinterpret_cast<const float4*>(src.d_normalsDownsampled)[idx];
                    if (nj.x != BF_MINF) {
                        const float n4[4] = {tr.e[0] * nj.x + tr.e[1] * nj.y + tr.e[2] * nj.z + tr.e[3] * nj.w,
                                             tr.e[4] * nj.x + tr.e[5] * nj.y + tr.e[6] * nj.z + tr.e[7] * nj.w,
                                             tr.e[8] * nj.x + tr.e[9] * nj.y + tr.e[10] * nj.z + tr.e[11] * nj.w,
                                             tr.e[12] * nj.x + tr.e[13] * nj.y + tr.e[14] * nj.z + tr.e[15] * nj.w};
                        const f3 cst = xform(tr, cs);
                        float u, v;
                        cameraToDepth(c, cst, u, v);
                        const int tx = f2i(roundf(u)), ty = f2i(roundf(v));
                        if (tx >= 0 && ty >= 0 && tx < (int)c.W && ty < (int)c.H) {
                            float ci[4], ni[4];
                            bilinear<4>(u, v, tgt.d_cameraposDownsampled, c.W, c.H, ci);
                            if (ci[2] > c.denseDepthMin && ci[2] < c.denseDepthMax) {
                                bilinear<4>(u, v, tgt.d_normalsDownsampled, c.W, c.H, ni);
                                if (ni[0] != BF_MINF) {
                                    const f3 ct = mk3(ci[0], ci[1], ci[2]);
                                    const f3 nt = mk3(ni[0], ni[1], ni[2]);
                                    const float dist = len3(cst - ct);
                                    const float dN = n4[0] * ni[0] + n4[1] * ni[1] + n4[2] * ni[2] + n4[3] * ni[3];
                                    if (dN >= c.denseNormalThresh && dist <= c.denseDistThresh) {
                                        float jI[3][6], jJ[3][6];
                                        if (i > 0) derivI(TjI, Ti, cs, jI);
                                        if (j > 0) derivJ(TiI, Tj, cs, jJ);
                                        if (useDepth) {                     // :244-277
                                            const f3 diff = ct - cst;
                                            Xd[12] = dot3(diff, nt);
                                            wd = c.wDepth * pw * powf(fmaxf(0.0f, 1.0f - ct.z / 2.0f), 2.5f);
#pragma unroll
                                            for (int k = 0; k < 6; ++k) {
                                                if (i > 0) Xd[k] = -(jI[0][k] * nt.x + jI[1][k] * nt.y + jI[2][k] * nt.z);
                                                if (j > 0) Xd[6 + k] = -(jJ[0][k] * nt.x + jJ[1][k] * nt.y + jJ[2][k] * nt.z);
                                            }
                                        }
                                        if (useColor) {                     // :278-304
                                            float dI[2], iT;
                                            bilinear<2>(u, v, tgt.d_intensityDerivsDownsampled, c.W, c.H, dI);
                                            bilinear<1>(u, v, tgt.d_intensityDownsampled, c.W, c.H, &iT);
                                            const float res = iT - src.d_intensityDownsampled[idx];
                                            if (dI[0] != BF_MINF && fabsf(res) < c.denseColorThresh &&
                                                sqrtf(dI[0] * dI[0] + dI[1] * dI[1]) > c.denseColorGradientMin) {
                                                const float z2 = cst.z * cst.z;
                                                const float P00 = c.fx / cst.z, P02 = -c.fx * cst.x / z2, P11 = c.fy / cst.z, P12 = -c.fy * cst.y / z2;
                                                Xc[12] = res;
                                                wc = c.wColor * pw * fmaxf(0.0f, 1.0f - fabsf(res) / (1.15f * c.denseColorThresh));
#pragma unroll
                                                for (int k = 0; k < 6; ++k) {
                                                    if (i > 0) Xc[k] = dI[0] * (P00 * jI[0][k] + 0.0f * jI[1][k] + P02 * jI[2][k]) + dI[1] * (0.0f * jI[0][k] + P11 * jI[1][k] + P12 * jI[2][k]);
                                                    if (j > 0) Xc[6 + k] = dI[0] * (P00 * jJ[0][k] + 0.0f * jJ[1][k] + P02 * jJ[2][k]) + dI[1] * (0.0f * jJ[0][k] + P11 * jJ[1][k] + P12 * jJ[2][k]);
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // depth rows then colour rows of these 64 pixels through the matrix core
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 0 && !useDepth) continue;
                if (pass == 1 && !useColor) continue;
#pragma unroll
                for (int k = 0; k < 13; ++k) tileX[wave][lane][k] = pass == 0 ? Xd[k] : Xc[k];
                tileX[wave][lane][13] = 0.0f; tileX[wave][lane][14] = 0.0f; tileX[wave][lane][15] = 0.0f;
                tileW[wave][lane] = pass == 0 ? wd : wc;
                __syncthreads();
#pragma unroll
                for (int g = 0; g < 16; g += 2) {
                    const int r0 = 4 * g + (int)(lane >> 4), r1 = r0 + 4;
                    const float b0 = tileX[wave][r0][lane & 15], b1 = tileX[wave][r1][lane & 15];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(tileW[wave][r0] * b0, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tileW[wave][r1] * b1, b1, acc1, 0, 0, 0);
                }
                __syncthreads();
            }
        }
        // D layout: lane l, reg r -> row 4*(l>>4)+r, col l&15
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
        __syncthreads();
        if (threadIdx.x < DENSE_BLK) {
            const uint32_t t = threadIdx.x;
            uint32_t row, col;
            if (t < 36) { row = t / 6; col = t % 6; }                        // ii
            else if (t < 72) { row = 6 + (t - 36) / 6; col = 6 + (t - 36) % 6; }   // jj
            else if (t < 108) { row = (t - 72) / 6; col = 6 + (t - 72) % 6; }      // ij (rows i, cols j)
            else if (t < 114) { row = t - 108; col = 12; }                    // J_i^T r
            else { row = 6 + (t - 114); col = 12; }                          // J_j^T r
            const uint32_t e = row * 16 + col;
            out[t] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ per directed pair: gather the sparse
// correspondences in ascending index order, add the dense blocks.  One wave per slot.
// Variable order inside a 6-block: [t0 t1 t2 | w0 w1 w2] (the reference's dense layout).
__global__ __launch_bounds__(256) void k_slots(Dev d, Cfg c, int useDense) {
    if (d.flags[FL_DONE]) return;
    const uint32_t nSlots = (uint32_t)d.flags[FL_NUM_SLOTS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wavesPerGrid = gridDim.x * (blockDim.x >> 6);
    for (uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nSlots; s += wavesPerGrid) {
        const uint32_t key = d.slotKey[s];
        const uint32_t start = d.keyStart[key], cnt = d.keyCount[key];
        const uint32_t a = lane / 6, b = lane % 6;      // lanes 0..35: entry (a,b); 36..41: g[lane-36]; 42..45: precond
        float accD = 0.0f, accO = 0.0f;
        uint32_t prev = 0;
        bool first = true;
        for (uint32_t k = 0; k < cnt; ++k) {
            // next entry in ascending (corrIdx, role) order: min over entries > prev
            uint32_t best = 0xFFFFFFFFu;
            for (uint32_t t = lane; t < cnt; t += 64) {
                const uint32_t v = d.corrList[start + t];
                if ((first || v > prev) && v < best) best = v;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
            prev = best; first = false;
            const bf_entry_j e = d.corr[best >> 1];
            const bool roleB = best & 1u;
            const m44 TI = d.T[e.imgIdx_i], TJ = d.T[e.imgIdx_j];
            const f3 wi = xform(TI, mk3(e.pos_i[0], e.pos_i[1], e.pos_i[2]));
            const f3 wj = xform(TJ, mk3(e.pos_j[0], e.pos_j[1], e.pos_j[2]));
            const f3 r = wi - wj;
            const f3 wx = roleB ? wj : wi, wy = roleB ? wi : wj;
            const float sx = roleB ? -1.0f : 1.0f;
            // column q of [I | dAlpha dBeta dGamma] evaluated at w  (LieDerivUtil.h:231-242)
            auto col = [](f3 w, uint32_t q) -> f3 {
                switch (q) {
                    case 0: return mk3(1.0f, 0.0f, 0.0f);
                    case 1: return mk3(0.0f, 1.0f, 0.0f);
                    case 2: return mk3(0.0f, 0.0f, 1.0f);
                    case 3: return mk3(0.0f, -w.z, w.y);
                    case 4: return mk3(w.z, 0.0f, -w.x);
                    default: return mk3(-w.y, w.x, 0.0f);
                }
            };
            if (lane < 36) {
                const f3 ja = col(wx, a);
                accD += dot3(ja, col(wx, b));                 // (s J)^T (s J): sign cancels
                accO += -dot3(ja, col(wy, b));                // J_x^T J_y carries opposite signs
            } else if (lane < 42) {
                accD += sx * dot3(col(wx, lane - 36), r);     // J_x^T r
            } else if (lane < 45) {
                const f3 jq = col(wx, lane - 42 + 3);
                accD += dot3(jq, jq);                         // preconditioner: sum da.da (unweighted, EquationsLie.h:105)
            } else if (lane == 45) {
                accD += 1.0f;                                 // pTrans += (1,1,1)
            }
        }
        // dense contribution of this directed pair
        float dD = 0.0f, dO = 0.0f;
        if (useDense) {
            const uint32_t pp = d.keyPair[key];
            if (pp) {
                const float* blk = d.denseBlocks + (size_t)(pp - 1) * DENSE_BLK;
                const uint32_t i = key / d.N, j = key % d.N;
                const bool fwd = i < j;                        // stored pair is (min,max); blocks are ii, jj, ij
                if (lane < 36) {
                    dD = fwd ? blk[a * 6 + b] : blk[36 + a * 6 + b];
                    dO = fwd ? blk[72 + a * 6 + b] : blk[72 + b * 6 + a];
                } else if (lane < 42) {
                    dD = fwd ? blk[108 + (lane - 36)] : blk[114 + (lane - 36)];
                }
            }
        }
        if (lane < 36) {
            d.slotD[(size_t)s * 36 + lane] = c.wSparse * accD + dD;
            d.slotO[(size_t)s * 36 + lane] = c.wSparse * accO + dO;
        } else if (lane < 42) {
            d.slotG[(size_t)s * 6 + (lane - 36)] = c.wSparse * accD + dD;
        } else if (lane < 46) {
            d.slotP[(size_t)s * 4 + (lane - 42)] = accD;
        }
    }
}

// per image: diagonal block, right-hand side -J^T F, Jacobi preconditioner (EquationsLie.h:63-148)
__global__ __launch_bounds__(256) void k_rows(Dev d) {
    if (d.flags[FL_DONE]) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= d.N) return;
    const uint32_t s0 = d.rowStart[i], s1 = d.rowStart[i + 1];
    float acc = 0.0f;
    for (uint32_t s = s0; s < s1; ++s) {
        if (lane < 36) acc += d.slotD[(size_t)s * 36 + lane];
        else if (lane < 42) acc += d.slotG[(size_t)s * 6 + (lane - 36)];
        else if (lane < 46) acc += d.slotP[(size_t)s * 4 + (lane - 42)];
    }
    if (lane < 36) d.diagA[(size_t)i * 36 + lane] = acc;
    else if (lane < 42) d.rhs[(size_t)i * 6 + (lane - 36)] = -acc;
    else if (lane < 45) d.prec[(size_t)i * 6 + 3 + (lane - 42)] = acc > FLOAT_EPSILON ? 1.0f / acc : 1.0f;   // rotation
    else if (lane == 45) {
        const float v = acc > FLOAT_EPSILON ? 1.0f / acc : 1.0f;
        d.prec[(size_t)i * 6 + 0] = v; d.prec[(size_t)i * 6 + 1] = v; d.prec[(size_t)i * 6 + 2] = v;
    }
}

// ------------------------------------------------------------------ PCG: one persistent workgroup
BF_DEV float blockSum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) t += sh[w];
    return t;
}
BF_DEV float blockMax(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) t = fmaxf(t, sh[w]);
    return t;
}

constexpr size_t PCG_LDS_MAX = 160 * 1024 - 1024;      // dynamic LDS of k_pcg<true> (gfx950: 160 KB per workgroup)
// The CG vectors (p, r, delta, M^-1, Ap) and the row offsets live in LDS when they fit (VEC_LDS; 124 N + 4 bytes, N <= ~1300 key
// frames): every phase of an iteration is then one LDS round trip instead of a store -> barrier -> load trip through L2, and
// what is left per iteration is streaming the off-diagonal blocks once.  The arithmetic and its order are the same in both
// variants.
template <bool VEC_LDS>
__global__ __launch_bounds__(1024) void k_pcg(Dev d, uint32_t nLin, uint32_t gnIter, int lastGN) {
    if (d.flags[FL_DONE]) return;
    extern __shared__ float dynLds[];
    __shared__ float sh[16];
    const uint32_t N = d.N, n6 = 6 * N;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    float* const P = VEC_LDS ? dynLds : d.p;
    float* const R = VEC_LDS ? dynLds + n6 : d.r;
    float* const X = VEC_LDS ? dynLds + 2 * n6 : d.delta;
    float* const AP = VEC_LDS ? dynLds + 3 * n6 : d.Ap;
    const float* const M = VEC_LDS ? dynLds + 4 * n6 : d.prec;
    const uint32_t* const RS = VEC_LDS ? (const uint32_t*)(dynLds + 5 * n6) : d.rowStart;
    if (VEC_LDS) {
        for (uint32_t t = threadIdx.x; t < n6; t += blockDim.x) dynLds[4 * n6 + t] = d.prec[t];
        for (uint32_t t = threadIdx.x; t <= N; t += blockDim.x) ((uint32_t*)(dynLds + 5 * n6))[t] = d.rowStart[t];
        __syncthreads();
    }
    // Initialization (SolverBundling.cu:755-794): r = -J^T F, p = M^-1 r, delta = 0
    float part = 0.0f;
    for (uint32_t t = threadIdx.x; t < n6; t += blockDim.x) {
        const bool var = t >= 6;
        const float rr = var ? d.rhs[t] : 0.0f;
        const float pp = var ? M[t] * rr : 0.0f;
        R[t] = rr; P[t] = pp; X[t] = 0.0f;
        part += rr * pp;
    }
    float rzOld = blockSum(part, sh);
    uint32_t it = 0;
    for (uint32_t lin = 0; lin < nLin; ++lin) {
        bool last = (lin == nLin - 1);
        ++it;
        __syncthreads();
        // Ap = A p : block-row gather (replaces PCGStep_Kernel0/1a/_Dense, :870-928).  A quarter wave (16 lanes) per block row, so 64
        // rows are in flight per pass and a lane has its whole 144-byte block (nine 16-byte loads) in flight at once; key-frame
        // graphs have ~10-30 neighbours per row, which 16 lanes cover in one or two strides.
        {
            const uint32_t q = lane >> 4, l16 = lane & 15;
            for (uint32_t i0 = 1; i0 < N; i0 += 4 * nWaves) {
                const uint32_t i = i0 + wave * 4 + q;
                float acc[6] = {0, 0, 0, 0, 0, 0};
                if (i < N) {
                    const uint32_t s0 = RS[i], s1 = RS[i + 1];
                    for (uint32_t s = s0 + l16; s < s1; s += 16) {
                        const float4* O4 = reinterpret_cast<const float4*>(d.slotO + (size_t)s * 36);     // 144-byte stride: 16-byte aligned
                        const uint32_t col = d.slotCol[s];
                        float O[36];
#pragma unroll
                        for (int v = 0; v < 9; ++v) { const float4 o = O4[v]; O[4 * v] = o.x; O[4 * v + 1] = o.y; O[4 * v + 2] = o.z; O[4 * v + 3] = o.w; }
                        const float* pj = P + 6 * col;
                        const float q0 = pj[0], q1 = pj[1], q2 = pj[2], q3 = pj[3], q4 = pj[4], q5 = pj[5];
#pragma unroll
                        for (int a = 0; a < 6; ++a)
                            acc[a] += ((((O[a * 6 + 0] * q0 + O[a * 6 + 1] * q1) + O[a * 6 + 2] * q2) + O[a * 6 + 3] * q3) + O[a * 6 + 4] * q4) + O[a * 6 + 5] * q5;
                    }
                }
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc[a] += __shfl_xor(acc[a], o, 64);      // butterfly inside the 16-lane group
                if (i < N && l16 < 6) {
                    const float* A = d.diagA + (size_t)i * 36 + l16 * 6;
                    const float* pi = P + 6 * i;
                    float v = 0.0f;
#pragma unroll
                    for (int b = 0; b < 6; ++b) v += A[b] * pi[b];
                    float sel = acc[0];
                    if (l16 == 1) sel = acc[1]; else if (l16 == 2) sel = acc[2]; else if (l16 == 3) sel = acc[3]; else if (l16 == 4) sel = acc[4]; else if (l16 == 5) sel = acc[5];
                    AP[6 * i + l16] = v + sel;
                }
            }
        }
        __syncthreads();
        part = 0.0f;
        for (uint32_t t = 6 + threadIdx.x; t < n6; t += blockDim.x) part += P[t] * AP[t];
        const float pAp = blockSum(part, sh);                                      // PCGStep_Kernel1b
        const float alpha = pAp > FLOAT_EPSILON ? rzOld / pAp : 0.0f;             // PCGStep_Kernel2 :948-983
        part = 0.0f;
        for (uint32_t t = 6 + threadIdx.x; t < n6; t += blockDim.x) {
            X[t] = X[t] + alpha * P[t];
            const float rr = R[t] - alpha * AP[t];
            R[t] = rr;
            part += (M[t] * rr) * rr;
        }
        const float rzNew = blockSum(part, sh);
        if (fabsf(pAp) < 5e-7f) last = true;                                      // :1088-1093
        const float beta = rzOld > FLOAT_EPSILON ? rzNew / rzOld : 0.0f;          // PCGStep_Kernel3 :985-1022
        rzOld = rzNew;
        for (uint32_t t = 6 + threadIdx.x; t < n6; t += blockDim.x) P[t] = M[t] * R[t] + beta * P[t];
        if (last) break;
    }
    __syncthreads();
    // Lie update (computeLieUpdate, LieDerivUtil.h:301-307) + GN convergence (EvalGNConvergence :694-749)
    float mx = 0.0f;
    for (uint32_t i = 1 + threadIdx.x; i < N; i += blockDim.x) {
        const f3 dT = ld3(X + 6 * i), dW = ld3(X + 6 * i + 3);
        f3 nw, nt;
        lieUpdate(dW, dT, ld3(d.xRot + 3 * i), ld3(d.xTrans + 3 * i), nw, nt);
        d.xRot[3 * i] = nw.x; d.xRot[3 * i + 1] = nw.y; d.xRot[3 * i + 2] = nw.z;
        d.xTrans[3 * i] = nt.x; d.xTrans[3 * i + 1] = nt.y; d.xTrans[3 * i + 2] = nt.z;
        if (d.valid[i] != 0) mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(dW.x), fabsf(dT.x)), fmaxf(fabsf(dW.y), fabsf(dT.y))), fmaxf(fabsf(dW.z), fabsf(dT.z))));
    }
    mx = blockMax(mx, sh);
    if (threadIdx.x == 0) {
        d.flags[FL_GN_ITERS] = (int)gnIter + 1;
        d.flags[FL_PCG_ITERS + gnIter] = (int)it;
        if (!lastGN && mx < 0.005f) d.flags[FL_DONE] = 1;                         // :1204-1210
    }
}

// Cooperative variant: G workgroups of 256 threads, each owning a contiguous range of block rows whose off-diagonal blocks,
// column indices and diagonal blocks stay in its LDS for the whole solve (a key-frame graph of N=500 with ~45 neighbours per
// row is 3 MB: one CU streams that in ~25 us per iteration, 64 CUs hold it on chip).  Per iteration a workgroup computes Ap for
// its rows, publishes them (agent-scope stores into a double-buffered exchange vector) and waits on ONE grid barrier; every
// workgroup then reads the whole Ap and carries the CG vectors redundantly in its own LDS - the dot products and updates are
// the same arithmetic in the same order on every workgroup, so alpha, beta and the early-out decision agree bit for bit and
// no second exchange is needed.  The result does not depend on G (row arithmetic is per row, reductions are per 256 threads).
// Exchanged data and the barrier counter use relaxed agent-scope atomics (sc1 accesses: coherent across the 8 XCD L2s) and an
// explicit s_waitcnt instead of fences, so no L2 write-back / invalidate disturbs the voxel kernels running beside it.
constexpr uint32_t COOP_THREADS = 256, COOP_THREADS_SMALL = 1024, COOP_MAX_GROUPS = 64, COOP_ROWS_PER_GROUP = 8, COOP_SPIN_LIMIT = 1u << 22;

BF_DEV void coopPublish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BF_DEV float coopRead(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all workgroups of the launch arrive; returns false when a peer never shows up (never expected: every group is resident
// on its own CU; the bound turns a scheduling surprise into an error instead of a hung queue)
BF_DEV bool gridBarrier(uint32_t* counter, uint32_t target, int* shFail) {
    __builtin_amdgcn_s_waitcnt(0);                   // this wave's published stores have reached the coherence point
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > COOP_SPIN_LIMIT) { *shFail = 1; break; }
        }
    }
    __syncthreads();
    return *shFail == 0;
}

// THREADS: 256; 1024 for ONE workgroup over a small problem (6 N <= 256: every vector element has a thread of its own under both widths and the wave
// partials beyond the first four are zeros, so the sums - and every bit of the result - are those of the 256-thread form): the row products, one wave per row,
// take N / 16 instead of N / 4 trips - the single-workgroup solves of a chunk (11 frames) and of the first ~40 key frames are latency chains of 100 - 150 iterations.
// vecGlobal: the workgroup's private copies of the five vectors and of the row offsets live in global memory (d.coopVec, one region per
// workgroup, L2-resident) instead of LDS - the form for problems whose vectors do not fit (31 N floats: N > ~1300 key frames; round 2 fell
// back to ONE workgroup there, 79 us per iteration at N = 2000).  Same arithmetic, same order.
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void k_pcg_coop(Dev d, uint32_t nLin, uint32_t gnIter, int lastGN, uint32_t ldsFloats, uint32_t vecGlobal) {
    if (d.flags[FL_DONE]) return;
    extern __shared__ __align__(16) float coopLds[];
    float* const dynLds = coopLds;
    __shared__ float sh[16];
    __shared__ int shFail;
    const uint32_t N = d.N, n6 = 6 * N, G = gridDim.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = THREADS >> 6;
    const uint32_t vecFloats = (5 * n6 + (N + 1) + 3) & ~3u;
    float* const vec = vecGlobal ? d.coopVec + (size_t)blockIdx.x * vecFloats : dynLds;
    float* const P = vec;
    float* const R = vec + n6;
    float* const X = vec + 2 * n6;
    float* const AP = vec + 3 * n6;
    float* const M = vec + 4 * n6;
    uint32_t* const RS = reinterpret_cast<uint32_t*>(vec + 5 * n6);
    // rows [r0, r1) belong to this workgroup
    const uint32_t rpg = (N - 1 + G - 1) / G;
    const uint32_t r0 = min(N, 1 + blockIdx.x * rpg), r1 = min(N, r0 + rpg);
    const uint32_t ldsBase = vecGlobal ? 0u : 5 * n6 + (N + 1);
    float* const DG = dynLds + ldsBase;                                        // diagonal blocks of the own rows
    float* const OL = dynLds + ((ldsBase + rpg * 36 + 3) & ~3u);               // off-diagonal blocks (16-byte aligned), then one column index per slot
    if (threadIdx.x == 0) shFail = 0;
    for (uint32_t t = threadIdx.x; t < n6; t += THREADS) M[t] = d.prec[t];
    for (uint32_t t = threadIdx.x; t <= N; t += THREADS) RS[t] = d.rowStart[t];
    for (uint32_t t = threadIdx.x; t < (r1 - r0) * 36; t += THREADS) DG[t] = d.diagA[(size_t)r0 * 36 + t];
    __syncthreads();
    const uint32_t sBase = RS[r0], sEnd = RS[r1];
    const uint32_t ldsSlots = min(sEnd - sBase, (ldsFloats - (uint32_t)(OL - dynLds)) / 37u);
    uint32_t* const CL = reinterpret_cast<uint32_t*>(OL + (size_t)ldsSlots * 36);
    {
        const float4* src = reinterpret_cast<const float4*>(d.slotO + (size_t)sBase * 36);
        float4* dst = reinterpret_cast<float4*>(OL);
        for (uint32_t t = threadIdx.x; t < ldsSlots * 9; t += THREADS) dst[t] = src[t];
        for (uint32_t t = threadIdx.x; t < ldsSlots; t += THREADS) CL[t] = d.slotCol[sBase + t];
    }
    // Initialization (SolverBundling.cu:755-794): r = -J^T F, p = M^-1 r, delta = 0
    float part = 0.0f;
    for (uint32_t t = threadIdx.x; t < n6; t += THREADS) {
        const bool var = t >= 6;
        const float rr = var ? d.rhs[t] : 0.0f;
        const float pp = var ? M[t] * rr : 0.0f;
        R[t] = rr; P[t] = pp; X[t] = 0.0f;
        part += rr * pp;
    }
    float rzOld = blockSum(part, sh);
    float* const xchg[2] = {d.Ap, d.p};                                        // exchange vectors (the single-workgroup kernel's Ap and p)
    uint32_t* const counter = reinterpret_cast<uint32_t*>(d.gridBar) + gnIter;
    uint32_t it = 0;
    for (uint32_t lin = 0; lin < nLin; ++lin) {
        bool last = (lin == nLin - 1);
        ++it;
        __syncthreads();
        float* const out = G > 1 ? xchg[lin & 1] : AP;
        // Ap for the own rows: one wave per row, 8 lanes per off-diagonal block (lane a < 6 = row a of the 6x6 block)
        {
            const uint32_t g8 = lane >> 3, a = lane & 7;
            for (uint32_t i = r0 + wave; i < r1; i += nWaves) {
                const uint32_t s0 = RS[i], s1 = RS[i + 1];
                float acc = 0.0f;
                if (a < 6) {
                    for (uint32_t s = s0 + g8; s < s1; s += 8) {
                        const uint32_t ls = s - sBase;
                        float o0, o1, o2, o3, o4, o5; uint32_t col;
                        if (ls < ldsSlots) {
                            const float* O = OL + (size_t)ls * 36 + a * 6;
                            o0 = O[0]; o1 = O[1]; o2 = O[2]; o3 = O[3]; o4 = O[4]; o5 = O[5]; col = CL[ls];
                        } else {
                            const float* O = d.slotO + (size_t)s * 36 + a * 6;
                            o0 = O[0]; o1 = O[1]; o2 = O[2]; o3 = O[3]; o4 = O[4]; o5 = O[5]; col = d.slotCol[s];
                        }
                        const float* pj = P + 6 * col;
                        acc += ((((o0 * pj[0] + o1 * pj[1]) + o2 * pj[2]) + o3 * pj[3]) + o4 * pj[4]) + o5 * pj[5];
                    }
                }
                acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);
                if (lane < 6) {
                    const float* A = DG + (size_t)(i - r0) * 36 + lane * 6;
                    const float* pi = P + 6 * i;
                    float v = 0.0f;
#pragma unroll
                    for (int b = 0; b < 6; ++b) v += A[b] * pi[b];
                    if (G > 1) coopPublish(out + 6 * i + lane, v + acc); else out[6 * i + lane] = v + acc;
                }
            }
        }
        part = 0.0f;
        if (G > 1) {
            if (!gridBarrier(counter, G * it, &shFail)) { if (threadIdx.x == 0) d.flags[FL_BARRIER_FAIL] = 1; return; }
            for (uint32_t t = 6 + threadIdx.x; t < n6; t += THREADS) { const float ap = coopRead(out + t); AP[t] = ap; part += P[t] * ap; }
        } else {
            __syncthreads();
            for (uint32_t t = 6 + threadIdx.x; t < n6; t += THREADS) part += P[t] * AP[t];
        }
        const float pAp = blockSum(part, sh);                                      // PCGStep_Kernel1b
        const float alpha = pAp > FLOAT_EPSILON ? rzOld / pAp : 0.0f;             // PCGStep_Kernel2 :948-983
        part = 0.0f;
        for (uint32_t t = 6 + threadIdx.x; t < n6; t += THREADS) {           // each thread re-reads its own AP[t]: no barrier needed
            X[t] = X[t] + alpha * P[t];
            const float rr = R[t] - alpha * AP[t];
            R[t] = rr;
            part += (M[t] * rr) * rr;
        }
        const float rzNew = blockSum(part, sh);
        if (fabsf(pAp) < 5e-7f) last = true;                                      // :1088-1093
        const float beta = rzOld > FLOAT_EPSILON ? rzNew / rzOld : 0.0f;          // PCGStep_Kernel3 :985-1022
        rzOld = rzNew;
        for (uint32_t t = 6 + threadIdx.x; t < n6; t += THREADS) P[t] = M[t] * R[t] + beta * P[t];
        if (last) break;
    }
    if (blockIdx.x != 0) return;
    __syncthreads();
    // Lie update (computeLieUpdate, LieDerivUtil.h:301-307) + GN convergence (EvalGNConvergence :694-749)
    float mx = 0.0f;
    for (uint32_t i = 1 + threadIdx.x; i < N; i += THREADS) {
        const f3 dT = ld3(X + 6 * i), dW = ld3(X + 6 * i + 3);
        f3 nw, nt;
        lieUpdate(dW, dT, ld3(d.xRot + 3 * i), ld3(d.xTrans + 3 * i), nw, nt);
        d.xRot[3 * i] = nw.x; d.xRot[3 * i + 1] = nw.y; d.xRot[3 * i + 2] = nw.z;
        d.xTrans[3 * i] = nt.x; d.xTrans[3 * i + 1] = nt.y; d.xTrans[3 * i + 2] = nt.z;
        if (d.valid[i] != 0) mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(dW.x), fabsf(dT.x)), fmaxf(fabsf(dW.y), fabsf(dT.y))), fmaxf(fabsf(dW.z), fabsf(dT.z))));
    }
    mx = blockMax(mx, sh);
    if (threadIdx.x == 0) {
        d.flags[FL_GN_ITERS] = (int)gnIter + 1;
        d.flags[FL_PCG_ITERS + gnIter] = (int)it;
        if (!lastGN && mx < 0.005f) d.flags[FL_DONE] = 1;                         // :1204-1210
    }
}

// ------------------------------------------------------------------ residual analysis
BF_DEV float absMaxResidual(const Dev& d, const bf_entry_j& e, float w) {         // EquationsLie.h:27-40
    if (!validCorr(e)) return 0.0f;
    const m44 TI = poseToMatrix(ld3(d.xRot + 3 * e.imgIdx_i), ld3(d.xTrans + 3 * e.imgIdx_i));
    const m44 TJ = poseToMatrix(ld3(d.xRot + 3 * e.imgIdx_j), ld3(d.xTrans + 3 * e.imgIdx_j));
    const f3 a = xform(TI, mk3(e.pos_i[0], e.pos_i[1], e.pos_i[2])), b = xform(TJ, mk3(e.pos_j[0], e.pos_j[1], e.pos_j[2]));
    return fmaxf(w * fabsf(a.z - b.z), fmaxf(w * fabsf(a.x - b.x), w * fabsf(a.y - b.y)));
}

// EvalMaxResidualDevice (:511-550) + the host max loop of computeMaxResidual: first maximum in index order
// computeMaxResidual: the largest residual and the SMALLEST correspondence index that attains it.  G workgroups reduce interleaved slices
// (a single workgroup took 746 us over the 37 k correspondences of a 450-key-frame problem), the last pass reduces the G partials with the
// same order relation, so the result does not depend on G.
constexpr uint32_t MAXRES_GROUPS = 64;
BF_DEV bool maxResBetter(float o, int oi, float cur, int ci) { return cur < o || (cur == o && oi < ci && o > 0.0f); }

__global__ __launch_bounds__(1024) void k_max_residual(Dev d, float w) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    float best = 0.0f; int bi = 0;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
        const float r = absMaxResidual(d, d.corr[c], w);
        if (best < r) { best = r; bi = (int)c; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (uint32_t s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float o = sv[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
            if (maxResBetter(o, oi, sv[threadIdx.x], si[threadIdx.x])) { sv[threadIdx.x] = o; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { d.maxResPart[blockIdx.x] = sv[0]; d.maxIdxPart[blockIdx.x] = si[0]; }
}

__global__ __launch_bounds__(64) void k_max_residual_final(Dev d, uint32_t G) {
    float best = 0.0f; int bi = 0;
    if (threadIdx.x < G) { best = d.maxResPart[threadIdx.x]; bi = d.maxIdxPart[threadIdx.x]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (maxResBetter(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    if (threadIdx.x == 0) { d.maxRes[0] = best; d.maxIdx[0] = bi; }
}

__global__ void k_count_high(Dev d, float w, float thresh) {                      // CountHighResidualsDevice :657-668
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    bool hi = false;
    if (c < d.C) hi = absMaxResidual(d, d.corr[c], w) > thresh;
    const unsigned long long m = __ballot(hi);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(d.highCount, __popcll(m));
}

// EvalResidualDevice (:576-593), ordered single-workgroup sum
__global__ __launch_bounds__(1024) void k_energy(Dev d, float w, uint32_t slot, int honourDone) {
    if (honourDone && d.flags[FL_DONE] && slot > (uint32_t)d.flags[FL_GN_ITERS]) return;
    __shared__ float sh[16];
    float part = 0.0f;
    for (uint32_t c = threadIdx.x; c < d.C; c += blockDim.x) {
        const bf_entry_j e = d.corr[c];
        if (!validCorr(e)) continue;
        const m44 TI = poseToMatrix(ld3(d.xRot + 3 * e.imgIdx_i), ld3(d.xTrans + 3 * e.imgIdx_i));
        const m44 TJ = poseToMatrix(ld3(d.xRot + 3 * e.imgIdx_j), ld3(d.xTrans + 3 * e.imgIdx_j));
        const f3 r = xform(TI, mk3(e.pos_i[0], e.pos_i[1], e.pos_i[2])) - xform(TJ, mk3(e.pos_j[0], e.pos_j[1], e.pos_j[2]));
        part += w * dot3(r, r);
    }
    const float tot = blockSum(part, sh);
    if (threadIdx.x == 0) d.energies[slot] = tot;
}

// SBA.cu:75-108
__global__ void k_matrices_to_poses(const m44* T, uint32_t n, float* rot3, float* trans3, const int* valid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && valid[i]) {
        f3 r, t;
        matrixToPose(T[i], r, t);
        rot3[3 * i] = r.x; rot3[3 * i + 1] = r.y; rot3[3 * i + 2] = r.z;
        trans3[3 * i] = t.x; trans3[3 * i + 1] = t.y; trans3[3 * i + 2] = t.z;
    }
}
__global__ void k_poses_to_matrices(const float* rot3, const float* trans3, uint32_t n, m44* T, const int* valid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && valid[i]) T[i] = poseToMatrix(ld3(rot3 + 3 * i), ld3(trans3 + 3 * i));
}

// debug: expand the block-sparse dense contribution into the reference's 6N x 6N layout
__global__ void k_expand_dense(Dev d, float* JtJ, float* Jtr) {
    const uint32_t nPairs = (uint32_t)d.flags[FL_NUM_PAIRS];
    const uint32_t dim = 6 * d.N;
    for (uint32_t p = 0; p < nPairs; ++p) {      // serial over pairs: ordered accumulation
        const uint2 pr = d.densePairs[p];
        const float* blk = d.denseBlocks + (size_t)p * DENSE_BLK;
        const uint32_t t = threadIdx.x;
        if (t < 36) {
            const uint32_t a = t / 6, b = t % 6;
            JtJ[(size_t)(pr.x * 6 + a) * dim + pr.x * 6 + b] += blk[a * 6 + b];
            JtJ[(size_t)(pr.y * 6 + a) * dim + pr.y * 6 + b] += blk[36 + a * 6 + b];
            JtJ[(size_t)(pr.x * 6 + a) * dim + pr.y * 6 + b] += blk[72 + a * 6 + b];
            JtJ[(size_t)(pr.y * 6 + b) * dim + pr.x * 6 + a] += blk[72 + a * 6 + b];
        } else if (t < 42) {
            Jtr[pr.x * 6 + (t - 36)] += blk[108 + (t - 36)];
            Jtr[pr.y * 6 + (t - 36)] += blk[114 + (t - 36)];
        }
        __syncthreads();
    }
}

}  // namespace

// =========================================================================================
struct bf_solver {
    bf_solver_config cfg;
    uint32_t maxImages, maxResiduals, maxCorrPerImage;
    Dev d{};
    hipStream_t stream = nullptr;
    std::vector<void*> allocations;
    std::vector<float> convergence;
    float hMaxRes = 0.0f; int hMaxIdx = 0; int hBarrierFail = 0;
    bool pcgWide = true;                 // (BF_PCG_WIDE=0: the 256-thread kernel for every problem, for the A/B)
    int pcgGroups = -1;                  // BF_PCG_GROUPS: -1 automatic, 0 single-workgroup kernel, n forced group count
    uint32_t maxCoopGroups = COOP_MAX_GROUPS;
    bool forceVecGlobal = false;
    size_t coopVecFloats = 0;            // capacity of d.coopVec (k_pcg_coop with the vectors in global memory)
    uint32_t lastN = 0, lastGNrequested = 0;
    float lastWeightSparse = 1.0f;
    bool lastUsedDense = false;
};

namespace {
template <class T>
bool sAlloc(bf_solver* s, T** p, size_t n) {
    void* q = nullptr;
    if (BF_MALLOC(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return false;
    s->allocations.push_back(q);
    *p = (T*)q;
    return true;
}
}  // namespace

extern "C" {

int bf_solver_create(uint32_t maxNumberOfImages, uint32_t maxNumResiduals, const bf_solver_config* cfg, bf_solver** out) {
    BF_REQUIRE(cfg && out, "null argument");
    BF_REQUIRE(maxNumberOfImages >= 2 && maxNumResiduals > 0, "bad capacity");
    bf_solver* s = new bf_solver();
    s->cfg = *cfg;
    s->maxImages = maxNumberOfImages; s->maxResiduals = maxNumResiduals;
    BF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PCG_LDS_MAX));
    BF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg_coop<COOP_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PCG_LDS_MAX));
    BF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pcg_coop<COOP_THREADS_SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PCG_LDS_MAX));
    if (const char* e = getenv("BF_PCG_WIDE")) s->pcgWide = atoi(e) != 0;
    if (const char* e = getenv("BF_PCG_GROUPS")) s->pcgGroups = atoi(e);        // 0: single-workgroup kernel, n > 0: force n groups
    if (const char* e = getenv("BF_PCG_VEC_GLOBAL")) s->forceVecGlobal = atoi(e) != 0;      // tests: the large-N form (vectors in global memory) on a small problem
    {   // every group of the cooperative PCG must be resident at once (each may take a whole CU's LDS): never ask for more than
        // half of the CUs this device (or compute partition) has
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            s->maxCoopGroups = std::max(1u, std::min(COOP_MAX_GROUPS, (uint32_t)cus / 2u));
        else (void)hipGetLastError();                   // keep the default (a full MI355X has 256 CUs)
    }
    s->maxCorrPerImage = std::min(std::max(maxNumResiduals / maxNumberOfImages, 1000u), 4000u);   // .cpp:39
    const size_t N = maxNumberOfImages, M = N * N, C = maxNumResiduals;
    Dev& d = s->d;
    // a directed slot exists per pair that has correspondences (<= 2C) or dense overlap (<= N^2)
    d.maxSlots = (uint32_t)std::min<size_t>(M, 2 * C + (N <= 64 ? M : 4 * N));
    d.maxPairs = (uint32_t)(N * (N - 1) / 2);
    if (N > 64) d.maxPairs = (uint32_t)std::min<size_t>(d.maxPairs, d.maxSlots);
    bool ok = sAlloc(s, &d.T, N) && sAlloc(s, &d.Tinv, N) && sAlloc(s, &d.keyCount, M) && sAlloc(s, &d.keyStart, M) && sAlloc(s, &d.cursor, M) &&
              sAlloc(s, &d.slotOfKey, M) && sAlloc(s, &d.denseRaw, M) && sAlloc(s, &d.keyPair, M) && sAlloc(s, &d.rowStart, N + 1) &&
              sAlloc(s, &d.slotCol, (size_t)d.maxSlots) && sAlloc(s, &d.slotKey, (size_t)d.maxSlots) && sAlloc(s, &d.corrList, 2 * C) &&
              sAlloc(s, &d.slotD, (size_t)d.maxSlots * 36) && sAlloc(s, &d.slotO, (size_t)d.maxSlots * 36) && sAlloc(s, &d.slotG, (size_t)d.maxSlots * 6) &&
              sAlloc(s, &d.slotP, (size_t)d.maxSlots * 4) && sAlloc(s, &d.diagA, N * 36) && sAlloc(s, &d.rhs, N * 6) && sAlloc(s, &d.prec, N * 6) &&
              sAlloc(s, &d.delta, N * 6) && sAlloc(s, &d.r, N * 6) && sAlloc(s, &d.p, N * 6) && sAlloc(s, &d.Ap, N * 6) &&
              sAlloc(s, &d.densePairs, (size_t)d.maxPairs) && sAlloc(s, &d.denseWeight, (size_t)d.maxPairs) &&
              sAlloc(s, &d.denseBlocks, (size_t)d.maxPairs * DENSE_BLK) && sAlloc(s, &d.flags, FL_COUNT) && sAlloc(s, &d.gridBar, 32) && sAlloc(s, &d.energies, 40) &&
              sAlloc(s, &d.maxRes, 1) && sAlloc(s, &d.maxIdx, 1) && sAlloc(s, &d.highCount, 1) && sAlloc(s, &d.numEntriesPerRow, N) &&
              sAlloc(s, &d.scanRow, 3 * (N + 1)) && sAlloc(s, &d.coopVec, s->coopVecFloats = ((31 * N + 4096 <= PCG_LDS_MAX / 4 && !s->forceVecGlobal) ? (size_t)4 : (size_t)s->maxCoopGroups * ((31 * N + 1 + 3) & ~(size_t)3))) && sAlloc(s, &d.maxResPart, MAXRES_GROUPS) && sAlloc(s, &d.maxIdxPart, MAXRES_GROUPS);
    if (!ok) { set_error("bf_solver_create: hipMalloc failed"); bf_solver_destroy(s); return BF_ERR_HIP; }
    (void)hipMemset(d.flags, 0, FL_COUNT * sizeof(int));
    *out = s;
    return BF_OK;
}

int bf_solver_destroy(bf_solver* s) {
    if (!s) return BF_OK;
    (void)hipStreamSynchronize(s->stream);
    for (void* p : s->allocations) (void)hipFree(p);
    delete s;
    return BF_OK;
}

int bf_solver_set_stream(bf_solver* s, void* st) { BF_REQUIRE(s, "null solver"); s->stream = (hipStream_t)st; return BF_OK; }

int bf_solver_solve(bf_solver* s, bf_entry_j* d_corr, uint32_t numCorr, const int32_t* d_valid, uint32_t N, uint32_t nNonLin, uint32_t nLin,
                    const bf_cached_frame* d_cache, uint32_t cw, uint32_t ch, const float K4[4], const float* wS, const float* wDD,
                    const float* wDC, uint32_t numWeights, int usePairwise, float* d_rot, float* d_trans, int rebuildJT, int findMaxResidual,
                    uint32_t revalidateIdx) {
    (void)rebuildJT; (void)revalidateIdx;    // the block structure is rebuilt every Gauss-Newton iteration
    BF_REQUIRE(s && d_valid && d_rot && d_trans && wS && wDD && wDC, "null argument");
    BF_REQUIRE(numCorr == 0 || d_corr, "null correspondences");
    nNonLin = std::min(nNonLin, numWeights);                                  // .cpp:193
    BF_REQUIRE(N > 1 && nNonLin > 0 && nLin > 0, "numberOfImages > 1 && nNonLinearIterations > 0 required");   // MLIB_ASSERT .cpp:194
    BF_REQUIRE(N <= s->maxImages && numCorr <= s->maxResiduals, "problem exceeds the solver's capacity");
    BF_REQUIRE(nNonLin <= 32, "at most 32 non-linear iterations");
    Dev d = s->d;
    d.corr = d_corr; d.C = numCorr; d.valid = d_valid; d.N = N; d.xRot = d_rot; d.xTrans = d_trans; d.cache = d_cache;
    s->d = d;
    Cfg c;
    c.denseDistThresh = s->cfg.denseDistThresh; c.denseNormalThresh = s->cfg.denseNormalThresh; c.denseColorThresh = s->cfg.denseColorThresh;
    c.denseColorGradientMin = s->cfg.denseColorGradientMin; c.denseDepthMin = s->cfg.denseDepthMin; c.denseDepthMax = s->cfg.denseDepthMax;
    c.subsample = s->cfg.denseOverlapCheckSubsampleFactor;
    c.W = cw; c.H = ch;
    if (d_cache) {
        BF_REQUIRE(K4, "cache intrinsics missing");
        BF_REQUIRE(c.subsample > 0 && cw / c.subsample > 8, "denseDepthWidth / subsample factor must exceed 8");   // .cpp:243
        BF_REQUIRE((cw / c.subsample) * (ch / c.subsample) <= 512, "overlap pre-filter samples exceed one workgroup");
        c.fx = K4[0]; c.fy = K4[1]; c.cx = K4[2]; c.cy = K4[3];
    } else { c.fx = c.fy = c.cx = c.cy = 0.0f; }
    c.usePairwise = usePairwise;
    hipStream_t st = s->stream;
    const size_t M = (size_t)N * N;
    hipLaunchKernelGGL(k_solve_begin, dim3(1), dim3(64), 0, st, d);
    const bool record = s->cfg.recordConvergence != 0;
    if (record) hipLaunchKernelGGL(k_energy, dim3(1), dim3(1024), 0, st, d, wS[0], 0u, 0);
    bool anyDense = false, prevDense = false;
    for (uint32_t it = 0; it < nNonLin; ++it) {
        c.wSparse = wS[it]; c.wDepth = wDD[it]; c.wColor = wDC[it];
        const int useDense = (d_cache != nullptr) && (c.wDepth > 0.0f || c.wColor > 0.0f);
        anyDense |= useDense != 0;
        // The block structure (directed key counts, CSR offsets, the keys' correspondence lists) depends on the correspondences and, through the dense pairs, on the
        // poses.  Without a dense term it is the same in every Gauss-Newton iteration: built in the first, kept afterwards (five launches per iteration fewer in the
        // global solves: the chunk's solve job costs launches, not kernel time - profiles/r06_loop_schedule.md).
        const bool reuse = it > 0 && !useDense && !prevDense;
        prevDense = useDense != 0;
        hipLaunchKernelGGL(k_poses, dim3(std::max<uint32_t>(div_up(N, 256), std::min<uint32_t>(div_up((uint32_t)M, 256), 128u))), dim3(256), 0, st, d, reuse ? 0u : (uint32_t)M, useDense);
        if (useDense) {
            const dim3 grid = usePairwise ? dim3(N, N) : dim3(N - 1, 1);
            hipLaunchKernelGGL(k_dense_overlap, grid, dim3(512), 0, st, d, c);
        }
        if (!reuse) {
            if (numCorr) hipLaunchKernelGGL(k_key_count, dim3(div_up(numCorr, 256)), dim3(256), 0, st, d);
            if (it == 0) hipLaunchKernelGGL(k_row_entries, dim3(div_up(N, 64)), dim3(64), 0, st, d);     // table as of solve start (rebuildJT)
            hipLaunchKernelGGL(k_scan_rows, dim3(N), dim3(64), 0, st, d, useDense);
            hipLaunchKernelGGL(k_scan_base, dim3(1), dim3(1024), 0, st, d);
            hipLaunchKernelGGL(k_scan_fill, dim3(N), dim3(64), 0, st, d, useDense);
            if (numCorr) hipLaunchKernelGGL(k_fill, dim3(div_up(numCorr, 256)), dim3(256), 0, st, d);
        }
        if (useDense) {
            const uint32_t g = std::min<uint32_t>(std::max<uint32_t>(N * (N - 1) / 2, 1u), 1024u);
            hipLaunchKernelGGL(k_dense_weight, dim3(g), dim3(256), 0, st, d, c);
            hipLaunchKernelGGL(k_dense_build, dim3(g), dim3(256), 0, st, d, c);
        }
        hipLaunchKernelGGL(k_slots, dim3(std::min<uint32_t>(div_up(d.maxSlots, 4), 2048u)), dim3(256), 0, st, d, c, useDense);
        hipLaunchKernelGGL(k_rows, dim3(div_up(N, 4)), dim3(256), 0, st, d);
        {
            uint32_t G = N <= 32 ? 1u : std::min(s->maxCoopGroups, div_up(N - 1, COOP_ROWS_PER_GROUP));
            if (s->pcgGroups > 0) G = std::min(std::min<uint32_t>((uint32_t)s->pcgGroups, N - 1), s->maxCoopGroups);
            const uint32_t rpg = div_up(N - 1, G);
            size_t baseFloats = ((size_t)31 * N + 1 + (size_t)rpg * 36 + 3) & ~(size_t)3;    // 5 vectors, row offsets, own diagonal blocks
            const uint32_t vecGlobal = (s->forceVecGlobal || baseFloats + 37 * 8 > PCG_LDS_MAX / 4) ? 1u : 0u;      // the vectors do not fit into LDS: keep them in global memory (d.coopVec)
            if (vecGlobal) baseFloats = ((size_t)rpg * 36 + 3) & ~(size_t)3;
            const size_t ldsFloats = std::min<size_t>(PCG_LDS_MAX / 4, baseFloats + (size_t)rpg * std::min<uint32_t>(N - 1, 96u) * 37);
            const bool vecFits = !vecGlobal || (size_t)G * (((size_t)31 * N + 1 + 3) & ~(size_t)3) <= s->coopVecFloats;
            if (s->pcgGroups != 0 && nNonLin <= 32 && vecFits && baseFloats + 37 * 8 <= PCG_LDS_MAX / 4)
            {
                if (s->pcgWide && G == 1 && 6 * N <= COOP_THREADS)
                    hipLaunchKernelGGL(k_pcg_coop<COOP_THREADS_SMALL>, dim3(G), dim3(COOP_THREADS_SMALL), ldsFloats * 4, st, d, nLin, it, (int)(it == nNonLin - 1), (uint32_t)ldsFloats, vecGlobal);
                else
                    hipLaunchKernelGGL(k_pcg_coop<COOP_THREADS>, dim3(G), dim3(COOP_THREADS), ldsFloats * 4, st, d, nLin, it, (int)(it == nNonLin - 1), (uint32_t)ldsFloats, vecGlobal);
            }
            else if ((size_t)N * 124 + 4 <= PCG_LDS_MAX)                              // 5 vectors of 6N floats + N+1 row offsets
                hipLaunchKernelGGL(k_pcg<true>, dim3(1), dim3(1024), (size_t)N * 124 + 4, st, d, nLin, it, (int)(it == nNonLin - 1));
            else hipLaunchKernelGGL(k_pcg<false>, dim3(1), dim3(1024), 0, st, d, nLin, it, (int)(it == nNonLin - 1));
        }
        if (record) hipLaunchKernelGGL(k_energy, dim3(1), dim3(1024), 0, st, d, c.wSparse, it + 1, 1);
    }
    BF_HIP_TRY(hipGetLastError());
    s->lastN = N; s->lastGNrequested = nNonLin; s->lastWeightSparse = wS[nNonLin - 1]; s->lastUsedDense = anyDense;
    if (findMaxResidual) {                                                   // computeMaxResidual .cpp:313-427
        if (s->lastWeightSparse > 0.0f && numCorr > 0) {
            const uint32_t gmr = std::min<uint32_t>(MAXRES_GROUPS, std::max<uint32_t>(1u, div_up(numCorr, 1024u)));
            hipLaunchKernelGGL(k_max_residual, dim3(gmr), dim3(1024), 0, st, d, s->lastWeightSparse);
            hipLaunchKernelGGL(k_max_residual_final, dim3(1), dim3(64), 0, st, d, gmr);
            BF_HIP_TRY(hipMemcpyAsync(&s->hMaxRes, d.maxRes, 4, hipMemcpyDeviceToHost, st));
            BF_HIP_TRY(hipMemcpyAsync(&s->hMaxIdx, d.maxIdx, 4, hipMemcpyDeviceToHost, st));
            BF_HIP_TRY(hipMemcpyAsync(&s->hBarrierFail, d.flags + FL_BARRIER_FAIL, 4, hipMemcpyDeviceToHost, st));
            BF_HIP_TRY(hipStreamSynchronize(st));
            BF_REQUIRE(s->hBarrierFail == 0, "PCG grid barrier timed out (a workgroup of k_pcg_coop was never scheduled)");
        } else { s->hMaxRes = 0.0f; s->hMaxIdx = 0; }
    }
    if (record) {
        int flags[FL_COUNT];
        float en[40];
        BF_HIP_TRY(hipMemcpyAsync(flags, d.flags, sizeof flags, hipMemcpyDeviceToHost, st));
        BF_HIP_TRY(hipMemcpyAsync(en, d.energies, sizeof en, hipMemcpyDeviceToHost, st));
        BF_HIP_TRY(hipStreamSynchronize(st));
        s->convergence.assign(nNonLin + 1, -1.0f);                           // .cpp:202
        for (int k = 0; k <= flags[FL_GN_ITERS] && k <= (int)nNonLin; ++k) s->convergence[k] = en[k];
    }
    return BF_OK;
}

int bf_solver_get_var_to_corr_num_entries_per_row(bf_solver* s, const int32_t** d_out) {
    BF_REQUIRE(s && d_out, "null argument");
    *d_out = s->d.numEntriesPerRow;
    return BF_OK;
}

// The reference's per-image correspondence table holds m_maxCorrPerImage slots per image (CUDASolverBundling.cpp:39); correspondences
// beyond that are invalidated in atomic arrival order ("AT RANDOM", .cpp:195-199, SolverBundling.cu:1226-1248).  The block-sparse
// system here has no such limit and uses every correspondence; this call tells an integrator whether the reference would have
// dropped some in the last solve: the number of images whose correspondence count exceeds the limit, and the limit.
int bf_solver_get_corr_overflow(bf_solver* s, uint32_t* numImagesOverLimit, uint32_t* limit) {
    BF_REQUIRE(s && numImagesOverLimit, "null argument");
    if (limit) *limit = s->maxCorrPerImage;
    *numImagesOverLimit = 0;
    const uint32_t N = s->lastN;
    if (N == 0) return BF_OK;
    std::vector<int> rows(N);                 // the table of the last solve's start (k_row_entries)
    BF_HIP_TRY(hipMemcpyAsync(rows.data(), s->d.numEntriesPerRow, sizeof(int) * N, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int r : rows) if ((uint32_t)std::max(r, 0) > s->maxCorrPerImage) (*numImagesOverLimit)++;
    return BF_OK;
}

int bf_solver_get_max_residual(bf_solver* s, float* mx, int32_t* idx) {
    BF_REQUIRE(s && mx && idx, "null argument");
    *mx = s->hMaxRes; *idx = s->hMaxIdx;
    return BF_OK;
}

int bf_solver_get_max_residual_pair(bf_solver* s, uint32_t curFrame, const bf_entry_j* d_corr, uint32_t imageIndices[2], float* maxRes, int* remove) {
    (void)curFrame;
    BF_REQUIRE(s && d_corr && imageIndices && maxRes && remove, "null argument");
    bf_entry_j e;
    BF_HIP_TRY(hipMemcpyAsync(&e, d_corr + s->hMaxIdx, sizeof e, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    imageIndices[0] = e.imgIdx_i; imageIndices[1] = e.imgIdx_j;
    *maxRes = s->hMaxRes;
    *remove = (!(e.imgIdx_i == 0 && e.imgIdx_j < 10) && s->hMaxRes > s->cfg.optMaxResThresh) ? 1 : 0;   // .cpp:442
    return BF_OK;
}

int bf_solver_use_verification(bf_solver* s, const bf_entry_j* d_corr, uint32_t numCorr, int* out) {
    BF_REQUIRE(s && d_corr && out && numCorr > 0, "bad argument");
    BF_REQUIRE(s->d.xRot && s->d.xTrans, "useVerification before solve");
    Dev d = s->d;
    d.corr = d_corr; d.C = numCorr;
    BF_HIP_TRY(hipMemsetAsync(d.highCount, 0, 4, s->stream));
    // the reference evaluates this with an uninitialised weightSparse (.cpp:456); every sparse weight in
    // SBA.cpp:28-38 is 1.0, which is used here
    hipLaunchKernelGGL(k_count_high, dim3(div_up(numCorr, 256)), dim3(256), 0, s->stream, d, 1.0f, s->cfg.verifyOptDistThresh);
    int cnt = 0;
    BF_HIP_TRY(hipMemcpyAsync(&cnt, d.highCount, 4, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    *out = ((float)cnt / (float)numCorr >= s->cfg.verifyOptPercentThresh) ? 1 : 0;
    return BF_OK;
}

int bf_solver_get_convergence(bf_solver* s, float* out, uint32_t cap, uint32_t* count) {
    BF_REQUIRE(s && out && count, "null argument");
    const uint32_t n = (uint32_t)std::min<size_t>(cap, s->convergence.size());
    for (uint32_t i = 0; i < n; ++i) out[i] = s->convergence[i];
    *count = n;
    return BF_OK;
}

int bf_solver_get_iteration_counts(bf_solver* s, int32_t* out, uint32_t cap) {
    BF_REQUIRE(s && out && cap > 0, "bad argument");
    int flags[FL_COUNT];
    BF_HIP_TRY(hipMemcpyAsync(flags, s->d.flags, sizeof flags, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    BF_REQUIRE(flags[FL_BARRIER_FAIL] == 0, "PCG grid barrier timed out (a workgroup of k_pcg_coop was never scheduled)");
    out[0] = flags[FL_GN_ITERS];
    for (uint32_t k = 1; k < cap && k <= 32; ++k) out[k] = (int)k <= flags[FL_GN_ITERS] ? flags[FL_PCG_ITERS + k - 1] : 0;
    return BF_OK;
}

int bf_solver_debug_dense_system(bf_solver* s, float* hJtJ, float* hJtr, uint32_t N, int32_t* numPairs) {
    BF_REQUIRE(s && hJtJ && hJtr && numPairs && N == s->lastN, "bad argument");
    const size_t dim = 6 * (size_t)N;
    float *dJ = nullptr, *dr = nullptr;
    BF_HIP_TRY(BF_MALLOC(&dJ, dim * dim * 4));
    BF_HIP_TRY(BF_MALLOC(&dr, dim * 4));
    BF_HIP_TRY(hipMemsetAsync(dJ, 0, dim * dim * 4, s->stream));
    BF_HIP_TRY(hipMemsetAsync(dr, 0, dim * 4, s->stream));
    if (s->lastUsedDense) hipLaunchKernelGGL(k_expand_dense, dim3(1), dim3(64), 0, s->stream, s->d, dJ, dr);
    BF_HIP_TRY(hipMemcpyAsync(hJtJ, dJ, dim * dim * 4, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipMemcpyAsync(hJtr, dr, dim * 4, hipMemcpyDeviceToHost, s->stream));
    int flags[FL_COUNT];
    BF_HIP_TRY(hipMemcpyAsync(flags, s->d.flags, sizeof flags, hipMemcpyDeviceToHost, s->stream));
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    *numPairs = s->lastUsedDense ? flags[FL_NUM_PAIRS] : 0;
    (void)hipFree(dJ); (void)hipFree(dr);
    return BF_OK;
}

int bf_convert_matrices_to_poses(const float* d_T, uint32_t n, float* d_rot, float* d_trans, const int32_t* d_valid, void* st) {
    BF_REQUIRE(d_T && d_rot && d_trans && d_valid, "null argument");
    if (n) hipLaunchKernelGGL(k_matrices_to_poses, dim3(div_up(n, 64)), dim3(64), 0, (hipStream_t)st, reinterpret_cast<const m44*>(d_T), n, d_rot, d_trans, d_valid);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}
int bf_convert_poses_to_matrices(const float* d_rot, const float* d_trans, uint32_t n, float* d_T, const int32_t* d_valid, void* st) {
    BF_REQUIRE(d_T && d_rot && d_trans && d_valid, "null argument");
    if (n) hipLaunchKernelGGL(k_poses_to_matrices, dim3(div_up(n, 64)), dim3(64), 0, (hipStream_t)st, d_rot, d_trans, n, reinterpret_cast<m44*>(d_T), d_valid);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

}  // extern "C"
