// SIFT keypoint detection + description for gfx950 (replaces the SiftGPU fork's detection path:
// SiftGPU/ProgramCU.cu, SiftPyramid.cpp, SiftGPU.cpp of the reference; paths relative to
// /root/reference/FriedLiver/Source) behind the bf_sift_* C ABI.
//
// The reference issues ~130 launches and ~15 blocking read-backs per 640x480 frame.  Here:
//  * the 4-octave x 6-level Gaussian pyramid is built by a wavefront schedule of fused H+V LDS-tiled
//    blurs (levels of different octaves that are ready together share a launch: 18 launches, not 48);
//  * DoG values are never stored — the extremum test evaluates them from the Gaussian levels;
//  * keypoint lists, the feature-count limit, the <=2-orientation expansion and all counts stay on the
//    device (single-workgroup bookkeeping kernels); nothing is read back during detection;
//  * histogram / descriptor accumulation uses lane-private bins + a xor-butterfly instead of LDS float
//    atomics, so results are run-to-run deterministic and bit-comparable with the CPU oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/bf_detmath.h"
#include "bf_device.h"
#include "bf_internal.h"

using namespace bf;

namespace {

constexpr int NUM_OCT = 4, DOG_LEVELS = 3, NLEV = 6, NKL = NUM_OCT * DOG_LEVELS;
constexpr int MAX_FW = 33;
constexpr int TILE_W = 32, TILE_H = 32;
constexpr uint32_t MAXRAW = 6144;

struct Taps { int fw[6]; float k[6][MAX_FW]; };

// (jobs and levels are read from the kernel arguments at a run-time index: laid out so that no scalar load of one straddles a 64-byte line - a straddling
// s_load_dwordx8 of run-time-indexed arguments is what gave the batched voxel update its rare run-to-run differences, csrc/tsdf_batch.h BatchUpdOpApx)
struct alignas(64) BlurJob { const float* src; float* dst; int w, h, srcW, mode, filter, blocks; };   // mode 0 blur, 1 down-sample copy
struct BlurJobs { int n; BlurJob j[3]; };

struct alignas(128) LevelInfo {        // one (octave, key level): DoG level j = g[j+1]-g[j], j = 1..3
    const float* g[4];    // gaussian array index j-1 .. j+2   (DoG j-1, j, j+1)
    int w, h, octave, fmax, cap;
    float keyLocScale;
    const float* mag; const float* ang;    // gradient of gaussian level j-1 (array index j)
    float sigma;
};
struct Levels { LevelInfo l[NKL]; };

struct RawKey { int x, y, li; uint32_t ori; };       // ori: two packed 16-bit orientations (65535 = none)
struct Feat { float x, y, s, o; int li; int pad[3]; };

struct SiftDev {
    uint32_t* cand;        // per level candidate keys (row<<16|col), capacity cap
    uint32_t candOff[NKL];
    uint32_t* candCount;   // NKL
    RawKey* raw;           // compact ordered list after limit(0)
    int* counts;           // [0]=numRaw, [1]=numFeat, [2..2+NKL) levelNum after limit0, [16..16+NKL) final level counts, [30]=error
    Feat* feats;
    float* des;            // numFeat*128 floats
    uint32_t* ticket;      // k_keys_finalize: arrivals of its per-level workgroups
};

enum { CNT_RAW = 0, CNT_FEAT = 1, CNT_LEVEL0 = 2, CNT_LEVEL1 = 16, CNT_ERR = 30, CNT_TOTAL = 32 };

// ---------------------------------------------------------------- pyramid
// fused separable blur: tile 32x32 outputs, (32+2r)x(32+2r) inputs, H pass into LDS then V pass.
// taps are applied in index order 0..fw-1 like FilterH/FilterV (ProgramCU.cu:159-264), clamp-to-edge.
#ifndef BF_VAR_BLUR_THREADS
#define BF_VAR_BLUR_THREADS 1024
#endif
constexpr int BLUR_THREADS = BF_VAR_BLUR_THREADS;     // 16 waves per tile: the three phases are latency chains (11 / 7 / 4 trips of a 256-thread group), the device is otherwise idle
__global__ __launch_bounds__(BLUR_THREADS) void k_blur(BlurJobs jobs, Taps taps) {
    __shared__ float tileIn[(TILE_H + MAX_FW - 1) * (TILE_W + MAX_FW - 1)];
    __shared__ float tileH[(TILE_H + MAX_FW - 1) * TILE_W];
    int b = blockIdx.x, ji = 0;
    while (ji < jobs.n - 1 && b >= jobs.j[ji].blocks) { b -= jobs.j[ji].blocks; ++ji; }
    const BlurJob job = jobs.j[ji];
    const int tilesX = (job.w + TILE_W - 1) / TILE_W;
    const int x0 = (b % tilesX) * TILE_W, y0 = (b / tilesX) * TILE_H;
    if (job.mode == 1) {          // DownsampleKernel :330-354
        for (int t = threadIdx.x; t < TILE_W * TILE_H; t += BLUR_THREADS) {
            const int x = x0 + t % TILE_W, y = y0 + t / TILE_W;
            if (x < job.w && y < job.h) job.dst[(size_t)y * job.w + x] = job.src[(size_t)(y << 1) * job.srcW + min(x << 1, job.srcW - 1)];
        }
        return;
    }
    const int fw = taps.fw[job.filter], half = fw >> 1;
    const float* k = taps.k[job.filter];
    const int inW = TILE_W + fw - 1, inH = TILE_H + fw - 1;     // <= 64
    {   // one wave per input row (inW <= 64 columns): no division by a run-time width
        const int ix = threadIdx.x & 63;
        const int sx = min(max(x0 - half + ix, 0), job.w - 1);
        if (ix < inW)
            for (int iy = threadIdx.x >> 6; iy < inH; iy += BLUR_THREADS / 64) {
                const int sy = min(max(y0 - half + iy, 0), job.h - 1);
                tileIn[iy * inW + ix] = job.src[(size_t)sy * job.w + sx];
            }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TILE_W * inH; t += BLUR_THREADS) {
        const int ox = t % TILE_W, iy = t / TILE_W;
        // tileIn column ox+i holds the clamped source column (x0+ox) - half + i
        float v = 0.0f;
        for (int i = 0; i < fw; ++i) v += tileIn[iy * inW + ox + i] * k[i];
        tileH[t] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TILE_W * TILE_H; t += BLUR_THREADS) {
        const int ox = t % TILE_W, oy = t / TILE_W;
        const int gx = x0 + ox, gy = y0 + oy;
        if (gx >= job.w || gy >= job.h) continue;
        float v = 0.0f;
        for (int i = 0; i < fw; ++i) v += tileH[(oy + i) * TILE_W + ox] * k[i];
        job.dst[(size_t)gy * job.w + gx] = v;
    }
}

// gradient magnitude / orientation of a gaussian level (ComputeDOG_Kernel :550-569); linear fetches
// outside the image buffer read 0 like tex1Dfetch
struct alignas(64) GradJob { const float* g; float* mag; float* ang; int w, h, blocks; };
struct GradJobs { GradJob j[NKL]; };
__device__ __forceinline__ void gradBlock(const GradJobs& jobs, int b) {
    int ji = 0;
    while (ji < NKL - 1 && b >= jobs.j[ji].blocks) { b -= jobs.j[ji].blocks; ++ji; }
    const GradJob job = jobs.j[ji];
    const long n = (long)job.w * job.h;
    const long idx = (long)b * 256 + threadIdx.x;
    if (idx >= n) return;
    auto fetch = [&](long i) -> float { return (i < 0 || i >= n) ? 0.0f : job.g[i]; };
    const float dx = fetch(idx + 1) - fetch(idx - 1);
    const float dy = fetch(idx + job.w) - fetch(idx - job.w);
    const float grd = 0.5f * sqrtf(dx * dx + dy * dy);
    job.mag[idx] = grd;
    job.ang[idx] = grd == 0.0f ? 0.0f : bf_dm_atan2(dy, dx);
}

// ---------------------------------------------------------------- keypoint detection (ComputeKEY_Kernel :616-750)
struct DetectCfg { int W, H, depthW, depthH; float depthMin, depthMax, dogThreshold, edgeT; int blocks[NKL]; };

// (the gradient blocks ride in the same launch: they read the finished pyramid like the detection does and are independent of it - one dispatch fewer on the
// detection's queue)
__global__ __launch_bounds__(256) void k_detect(Levels lv, DetectCfg c, SiftDev d, const float* __restrict__ depth, GradJobs grad, int detectBlocks) {
    if ((int)blockIdx.x >= detectBlocks) { gradBlock(grad, (int)blockIdx.x - detectBlocks); return; }
    int b = blockIdx.x, li = 0;
    while (li < NKL - 1 && b >= c.blocks[li]) { b -= c.blocks[li]; ++li; }
    const LevelInfo L = lv.l[li];
    const int w = L.w, h = L.h;
    const int tilesX = (w + 15) / 16;
    const int col = (b % tilesX) * 16 + (threadIdx.x & 15), row = (b / tilesX) * 16 + (threadIdx.x >> 4);
    if (!(row > 0 && col > 0 && row < h - 2 && col < w - 2)) return;
    const long index = (long)row * w + col;
    const int depthx = f2i(roundf((L.keyLocScale * (float)col + 0.5f) * (float)(c.depthW - 1) / (float)(c.W - 1)));
    const int depthy = f2i(roundf((L.keyLocScale * (float)row + 0.5f) * (float)(c.depthH - 1) / (float)(c.H - 1)));
    if (depthx < 0 || depthx >= c.depthW || depthy < 0 || depthy >= c.depthH) return;
    const float dd = depth[(size_t)depthy * c.depthW + depthx];
    if (dd == BF_MINF || dd < c.depthMin || dd > c.depthMax) return;
#define DOG(LEV, IDX) (L.g[(LEV) + 1][(IDX)] - L.g[(LEV)][(IDX)])     /* LEV 0,1,2 = previous, current, next */
    const float v = DOG(1, index);
    if (fabsf(v) <= c.dogThreshold) return;
    float d10 = DOG(1, index - 1), d12 = DOG(1, index + 1);
    float nmax = fmaxf(d10, d12), nmin = fminf(d10, d12);
    if (v <= nmax && v >= nmin) return;
    float r0[3], r2[3], t3[3];
#define CMP3(OUT, LEV, IDX)                                                                        \
    OUT[0] = DOG(LEV, (IDX) - 1); OUT[1] = DOG(LEV, (IDX)); OUT[2] = DOG(LEV, (IDX) + 1);          \
    if (v > nmax) { nmax = fmaxf(nmax, OUT[0]); nmax = fmaxf(nmax, OUT[1]); nmax = fmaxf(nmax, OUT[2]); if (v < nmax) return; } \
    else { nmin = fminf(nmin, OUT[0]); nmin = fminf(nmin, OUT[1]); nmin = fminf(nmin, OUT[2]); if (v > nmin) return; }
    CMP3(r0, 1, index - w)
    CMP3(r2, 1, index + w)
    const float vx2 = v * 2.0f;
    const float fxx = d10 + d12 - vx2, fyy = r0[1] + r2[1] - vx2;
    const float fxy = 0.25f * (r2[2] + r0[0] - r2[0] - r0[2]);
    const float t1 = fxx * fyy - fxy * fxy, t2 = (fxx + fyy) * (fxx + fyy);
    if (t1 <= 0 || t2 > c.edgeT * t1) return;
    CMP3(t3, 0, index - w)
    CMP3(t3, 0, index)
    CMP3(t3, 0, index + w)
    CMP3(t3, 2, index - w)
    CMP3(t3, 2, index)
    CMP3(t3, 2, index + w)
#undef CMP3
#undef DOG
    const uint32_t pos = atomicAdd(&d.candCount[li], 1u);
    if (pos < (uint32_t)L.cap) d.cand[d.candOff[li] + pos] = ((uint32_t)row << 16) | (uint32_t)col;
}

// One workgroup per level sorts that level's candidates by (row, col) (round 5: one workgroup walked the twelve levels one after the other, 70-90 us per frame);
// the workgroup that finishes last keeps the first fmax per level, applies LimitFeatureCount(0) (SiftPyramid.cpp:227-255) and emits the compact ordered key
// list.  Hand-off: write-through stores, drained, ticket, agent acquire (the pattern of k_alloc_place, tsdf.hip).
__global__ __launch_bounds__(1024) void k_keys_finalize(Levels lv, SiftDev d, int featureCountThreshold) {
    __shared__ uint32_t keys[8192];
    __shared__ int levelNum[NKL], levelStart[NKL + 1];
    __shared__ uint32_t lastFlag;
    {
        const int li = blockIdx.x;
        const int cap = lv.l[li].cap;
        const uint32_t cnt = min(d.candCount[li], (uint32_t)cap);
        uint32_t npad = 64;
        while (npad < cnt) npad <<= 1;
        for (uint32_t i = threadIdx.x; i < npad; i += blockDim.x) keys[i] = i < cnt ? d.cand[d.candOff[li] + i] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 2; k <= npad; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const uint32_t a = keys[lo], bb = keys[hi];
                    if ((a > bb) == up) { keys[lo] = bb; keys[hi] = a; }
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) __hip_atomic_store(d.cand + d.candOff[li] + i, keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(d.ticket, 1u);
        lastFlag = t == gridDim.x - 1 ? 1u : 0u;
        if (lastFlag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!lastFlag) return;
    if (threadIdx.x == 0) {
        for (int li = 0; li < NKL; ++li) levelNum[li] = (int)min(min(d.candCount[li], (uint32_t)lv.l[li].cap), (uint32_t)lv.l[li].fmax);
        if (featureCountThreshold > 0) {
            int total = 0;
            for (int i = 0; i < NKL; ++i) total += levelNum[i];
            int i = 0;
            while (i < NKL && total - levelNum[i] > featureCountThreshold) { total -= levelNum[i]; levelNum[i++] = 0; }
        }
        int s = 0;
        for (int i = 0; i < NKL; ++i) { levelStart[i] = s; s += levelNum[i]; d.counts[CNT_LEVEL0 + i] = levelNum[i]; }
        levelStart[NKL] = s;
        d.counts[CNT_RAW] = s;
        d.counts[CNT_ERR] = 0;
        d.ticket[0] = 0u;
    }
    __syncthreads();
    for (int li = 0; li < NKL; ++li)
        for (int i = threadIdx.x; i < levelNum[li]; i += blockDim.x) {
            const uint32_t key = __hip_atomic_load(d.cand + d.candOff[li] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (sorted by the level's workgroup: read at the scope it was stored at)
            RawKey r; r.x = (int)(key & 0xFFFF); r.y = (int)(key >> 16); r.li = li; r.ori = 0xFFFFFFFFu;
            d.raw[levelStart[li] + i] = r;
        }
    if (threadIdx.x < NKL) d.candCount[threadIdx.x] = 0;     // recycle for the next frame
}

// ---------------------------------------------------------------- orientation (ComputeOrientation_Kernel :905-1142)
__global__ __launch_bounds__(64) void k_orientation(Levels lv, SiftDev d) {
    const int idx = blockIdx.x;
    if (idx >= d.counts[CNT_RAW]) return;
    const uint32_t lane = threadIdx.x;
    RawKey key = d.raw[idx];
    const LevelInfo L = lv.l[key.li];
    const int width = L.w, height = L.h;
    const float kx = key.x + 0.5f, ky = key.y + 0.5f, sigma = L.sigma;
    const float gsigma = sigma * 1.5f;
    const float win = fabsf(sigma) * 1.5f * 2.0f;
    const float dist_threshold = win * win + 0.5f;
    const float factor = -0.5f / (gsigma * gsigma);
    const float xmin = fmaxf(1.5f, floorf(kx - win) + 0.5f), ymin = fmaxf(1.5f, floorf(ky - win) + 0.5f);
    const float xmax = fminf(width - 1.5f, floorf(kx + win) + 0.5f), ymax = fminf(height - 1.5f, floorf(ky + win) + 0.5f);
    const uint32_t xlen = f2u(roundf(xmax - xmin + 1)), ylen = f2u(roundf(ymax - ymin + 1));
    const uint32_t num = xlen * ylen;
    float part[36];
#pragma unroll
    for (int b = 0; b < 36; ++b) part[b] = 0.0f;
    for (uint32_t s = lane; s < num; s += 64) {
        const float x = (float)(s % xlen) + xmin, y = (float)(s / xlen) + ymin;
        const float dx = x - kx, dy = y - ky;
        const float sq = dx * dx + dy * dy;
        if (sq < dist_threshold) {
            const size_t pi = (size_t)f2i(y) * width + (size_t)f2i(x);
            const float weight = L.mag[pi] * bf_dm_exp(sq * factor);
            int oidx = f2i(floorf(L.ang[pi] * 5.7295779513082320876798154814105f));
            if (oidx < 0) oidx += 36;
            if (oidx > 35) oidx = 35;
#pragma unroll
            for (int b = 0; b < 36; ++b) if (b == oidx) part[b] += weight;      // register-resident bins
        }
    }
    __shared__ float vote[36], tmpv[36];
#pragma unroll
    for (int b = 0; b < 36; ++b) { const float t = wave_sum(part[b]); if (lane == 0) vote[b] = t; }
    __syncthreads();
    float* src = vote; float* dst = tmpv;
    for (int it = 0; it < 6; ++it) {                          // 6x circular box filter :987-1003
        float v = 0.0f;
        if (lane < 36) v = (src[(lane + 35) % 36] + src[lane] + src[(lane + 1) % 36]) * (float)(1.0 / 3.0);
        __syncthreads();
        if (lane < 36) dst[lane] = v;
        __syncthreads();
        float* t = src; src = dst; dst = t;
    }
    if (lane == 0) {       // <=2 peaks above 0.8 max, parabolic refinement, 16-bit packing
        float maxv = 0.0f;
        for (int t = 0; t < 36; ++t) maxv = fmaxf(maxv, vote[t]);
        const float thr = maxv * 0.8f;
        float rot[2] = {0.0f, 0.0f};
        int ocount = 0, maxIndex = -1;
        for (int pass = 0; pass < 2; ++pass) {
            float bw = -1.0f; int bi = -1;
            for (int cb = 0; cb < 36; ++cb) {
                if (pass == 1 && cb == maxIndex) continue;
                if (vote[cb] > thr && vote[cb] > vote[(cb + 35) % 36] && vote[cb] > vote[(cb + 1) % 36])
                    if (bw < vote[cb]) { bw = vote[cb]; bi = cb; }
            }
            if (bi >= 0) {
                const int m = (bi + 35) % 36, p = (bi + 1) % 36;
                const float di = 0.5f * ((vote[p] - vote[m]) / (2.0f * vote[bi] - vote[p] - vote[m]));
                rot[pass] = (float)bi + di + 0.5f;
                ocount++;
                if (pass == 0) maxIndex = bi;
            } else if (pass == 0) break;
        }
        uint32_t us1 = 65535, us2 = 65535;
        if (ocount > 0) {
            float fr1 = rot[0] / 36.0f; if (fr1 < 0) fr1 += 1.0f;
            us1 = (uint32_t)(f2i(floorf(fr1 * 65535.0f)) & 0xFFFF);
            if (ocount > 1) { float fr2 = rot[1] / 36.0f; if (fr2 < 0) fr2 += 1.0f; us2 = (uint32_t)(f2i(floorf(fr2 * 65535.0f)) & 0xFFFF); }
        }
        d.raw[idx].ori = (us2 << 16) | us1;
    }
}

// single workgroup: ReshapeFeatureList (:1994-2026, ordered) + LimitFeatureCount(1)
__global__ __launch_bounds__(1024) void k_reshape(Levels lv, SiftDev d, float minKeyScale, int featureCountThreshold, int maxFeatures) {
    __shared__ uint8_t nOri[MAXRAW];
    __shared__ int outPos[MAXRAW];
    __shared__ int finalNum[NKL], finalStart[NKL + 1], levelFirstOut[NKL];
    const int nRaw = d.counts[CNT_RAW];
    for (int i = threadIdx.x; i < nRaw; i += blockDim.x) {
        const RawKey r = d.raw[i];
        const LevelInfo& L = lv.l[r.li];
        const uint32_t us1 = r.ori & 0xFFFF, us2 = r.ori >> 16;
        int n = 0;
        if (L.sigma * L.keyLocScale >= minKeyScale && us1 != 65535) { n = 1; if (us2 != 65535 && us2 != us1) n = 2; }
        nOri[i] = (uint8_t)n;
    }
    __syncthreads();
    // Per level: running output index in list order, capped at fmax (the atomic cap of the reference).  With e_i the number of orientations in front of key i
    // inside its level, the capped walk gives out_i = min(e_i, fmax) and n_i' = min(n_i, fmax - out_i): a block-wide exclusive scan (round 5; thread 0 walked
    // the up to 6144 keys alone, 50-90 us per frame).
    {
        __shared__ int waveSum[16];
        __shared__ int levelBase[NKL];       // e of the level's first key in the scan over the whole list
        const int E = (nRaw + (int)blockDim.x - 1) / (int)blockDim.x, c0 = (int)threadIdx.x * E;
        int cnt = 0;
        for (int e = 0; e < E; ++e) { const int i = c0 + e; if (i < nRaw) cnt += nOri[i]; }
        int incl = cnt;
        const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) waveSum[wv] = incl;
        __syncthreads();
        int run = incl - cnt;
        for (int q = 0; q < wv; ++q) run += waveSum[q];
        for (int e = 0; e < E; ++e) { const int i = c0 + e; if (i < nRaw) { outPos[i] = run; run += nOri[i]; } }      // exclusive scan over the whole list
        __syncthreads();
        if (threadIdx.x < NKL) {
            int start = 0;
            for (int q = 0; q < (int)threadIdx.x; ++q) start += d.counts[CNT_LEVEL0 + q];
            const int ln = d.counts[CNT_LEVEL0 + threadIdx.x];
            const int base = ln > 0 ? outPos[start] : 0;
            const int tot = ln > 0 ? outPos[start + ln - 1] + nOri[start + ln - 1] - base : 0;
            levelBase[threadIdx.x] = base;
            finalNum[threadIdx.x] = min(tot, lv.l[threadIdx.x].fmax);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nRaw; i += blockDim.x) {
            const int li = d.raw[i].li, fmax = lv.l[li].fmax;
            const int out = min(outPos[i] - levelBase[li], fmax);
            nOri[i] = (uint8_t)min((int)nOri[i], fmax - out);
            outPos[i] = out;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (featureCountThreshold > 0) {
            int total = 0;
            for (int i = 0; i < NKL; ++i) total += finalNum[i];
            int i = 0;
            while (i < NKL && total - finalNum[i] > featureCountThreshold) { total -= finalNum[i]; finalNum[i++] = 0; }
        }
        int s = 0;
        for (int li = 0; li < NKL; ++li) { finalStart[li] = s; s += finalNum[li]; d.counts[CNT_LEVEL1 + li] = finalNum[li]; }
        finalStart[NKL] = s;
        if (s > maxFeatures) { d.counts[CNT_ERR] = 1; s = maxFeatures; }       // Bundler.cpp:97 "too many keypoints"
        d.counts[CNT_FEAT] = s;
    }
    __syncthreads();
    const float fac = (float)(2.0 * 3.14159265358979323846 / 65535.0);
    const int nFeat = d.counts[CNT_FEAT];
    for (int i = threadIdx.x; i < nRaw; i += blockDim.x) {
        const RawKey r = d.raw[i];
        if (finalNum[r.li] == 0) continue;
        const LevelInfo& L = lv.l[r.li];
        for (int q = 0; q < nOri[i]; ++q) {
            const int o = finalStart[r.li] + outPos[i] + q;
            if (o >= nFeat) continue;
            Feat f;
            f.x = r.x + 0.5f; f.y = r.y + 0.5f; f.s = L.sigma;
            f.o = fac * (float)(q == 0 ? (r.ori & 0xFFFF) : (r.ori >> 16));
            f.li = r.li; f.pad[0] = f.pad[1] = f.pad[2] = 0;
            d.feats[o] = f;
        }
    }
}

// ---------------------------------------------------------------- descriptor (ComputeDescriptor_Kernel :1178-1257)
// one wave per 4x4 cell, 4 cells per workgroup, 4 workgroups per feature
__global__ __launch_bounds__(256) void k_descriptor(Levels lv, SiftDev d) {
    const int ft = blockIdx.x >> 2;
    if (ft >= d.counts[CNT_FEAT]) return;
    const uint32_t lane = threadIdx.x & 63;
    const int cell = ((blockIdx.x & 3) << 2) | (threadIdx.x >> 6);
    const Feat f = d.feats[ft];
    const LevelInfo L = lv.l[f.li];
    const int width = L.w, height = L.h;
    const float spt = fabsf(f.s * 3.0f);
    float s, c;
    bf_dm_sincos(f.o, &s, &c);
    const float anglef = f.o > 3.14159265358979323846f ? f.o - (float)(2.0 * 3.14159265358979323846) : f.o;
    const float cspt = c * spt, sspt = s * spt, crspt = c / spt, srspt = s / spt;
    const float rpi = (float)(4.0 / 3.14159265358979323846);
    const int ix = cell & 3, iy = cell >> 2;
    const float ox = ix - 1.5f, oy = iy - 1.5f;
    const float ptx = cspt * ox - sspt * oy + f.x, pty = cspt * oy + sspt * ox + f.y;
    const float bsz = fabsf(cspt) + fabsf(sspt);
    const float xmin = fmaxf(1.5f, floorf(ptx - bsz) + 0.5f), ymin = fmaxf(1.5f, floorf(pty - bsz) + 0.5f);
    const float xmax = fminf(width - 1.5f, floorf(ptx + bsz) + 0.5f), ymax = fminf(height - 1.5f, floorf(pty + bsz) + 0.5f);
    const uint32_t xlen = f2u(roundf(xmax - xmin + 1)), ylen = f2u(roundf(ymax - ymin + 1));
    const uint32_t size = xlen * ylen;
    float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t si = lane; si < size; si += 64) {
        const float x = (float)(si % xlen) + xmin, y = (float)(si / xlen) + ymin;
        const float dx = x - ptx, dy = y - pty;
        const float nx = crspt * dx + srspt * dy, ny = crspt * dy - srspt * dx;
        const float nxn = fabsf(nx), nyn = fabsf(ny);
        if (nxn < 1.0f && nyn < 1.0f) {
            const size_t pi = (size_t)f2i(y) * width + (size_t)f2i(x);
            const float dnx = nx + ox, dny = ny + oy;
            const float ww = bf_dm_exp(-0.125f * (dnx * dnx + dny * dny));
            const float wx = 1.0f - nxn, wy = 1.0f - nyn;
            const float weight = ww * wx * wy * L.mag[pi];
            float theta = (anglef - L.ang[pi]) * rpi;
            if (theta < 0) theta += 8.0f;
            const float fo = floorf(theta);
            const int fidx = f2i(fo) & 7, fidx1 = (f2i(fo) + 1) & 7;
            const float w1 = (fo + 1.0f - theta) * weight, w2 = (theta - fo) * weight;
#pragma unroll
            for (int b = 0; b < 8; ++b) { if (b == fidx) part[b] += w1; if (b == fidx1) part[b] += w2; }
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) { const float t = wave_sum(part[b]); if (lane == 0) d.des[(size_t)ft * 128 + cell * 8 + b] = t; }
}

// NormalizeDescriptor_Kernel (:1339-1370) + ConvertDescriptorToUChar (:2100) + CreateGlobalKeyPointList (:2049-2081)
// half a wave (32 lanes x 4 values) per feature
__global__ __launch_bounds__(256) void k_desc_finalize(Levels lv, SiftDev d, DetectCfg c, const float* __restrict__ depth, float* outKeys,
                                                       uint8_t* outDescs, int* outCount) {
    const int ft = blockIdx.x * 8 + (threadIdx.x >> 5);
    const uint32_t t = threadIdx.x & 31;
    const int nFeat = d.counts[CNT_FEAT];
    if (blockIdx.x == 0 && threadIdx.x == 0) outCount[0] = d.counts[CNT_ERR] ? -1 : nFeat;
    if (ft >= nFeat) return;
    float4 v = reinterpret_cast<const float4*>(d.des + (size_t)ft * 128)[t];
    float n1 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) n1 += __shfl_xor(n1, o, 32);
    n1 = 1.0f / sqrtf(n1);
    v.x = fminf(0.2f, v.x * n1); v.y = fminf(0.2f, v.y * n1); v.z = fminf(0.2f, v.z * n1); v.w = fminf(0.2f, v.w * n1);
    float n2 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 32);
    n2 = 1.0f / sqrtf(n2);
    v.x *= n2; v.y *= n2; v.z *= n2; v.w *= n2;
    uchar4 u;
    u.x = (unsigned char)f2i(512 * v.x + 0.5f); u.y = (unsigned char)f2i(512 * v.y + 0.5f);
    u.z = (unsigned char)f2i(512 * v.z + 0.5f); u.w = (unsigned char)f2i(512 * v.w + 0.5f);
    reinterpret_cast<uchar4*>(outDescs + (size_t)ft * 128)[t] = u;
    if (t == 0) {
        const Feat f = d.feats[ft];
        const float ks = lv.l[f.li].keyLocScale;
        const float posX = ks * (f.x - 0.5f) + 0.5f, posY = ks * (f.y - 0.5f) + 0.5f;
        const float depthX = posX * (float)(c.depthW - 1) / (float)(c.W - 1), depthY = posY * (float)(c.depthH - 1) / (float)(c.H - 1);
        const int ipx = f2i(roundf(depthX)), ipy = f2i(roundf(depthY));
        float4 k4 = make_float4(posX, posY, ks * f.s, depth[(size_t)ipy * c.depthW + ipx]);
        reinterpret_cast<float4*>(outKeys)[ft] = k4;
    }
}

void makeTapsHost(float sigma, int& fw, float* k) {                // CreateFilterKernel :431-460
    int sz = (int)ceil(4.0f * sigma - 0.5);
    int width = 2 * sz + 1;
    if (width > 33) { sz = 16; width = 33; } else if (width < 5) { sz = 2; width = 5; }
    float rv = 1.0f / (sigma * sigma), ksum = 0;
    for (int i = -sz; i <= sz; ++i) { const float v = expf(-0.5f * i * i * rv); k[i + sz] = v; ksum += v; }
    rv = 1.0f / ksum;
    for (int i = 0; i < width; ++i) k[i] *= rv;
    fw = width;
}

}  // namespace

struct bf_sift {
    int W, H, depthW, depthH;
    float depthMin, depthMax, minKeyScale;
    int featureCountThreshold, maxFeatures;
    hipStream_t stream = nullptr;
    Taps taps;
    Levels levels;
    GradJobs gradJobs; int gradBlocks = 0;
    DetectCfg detect; int detectBlocks = 0;
    std::vector<BlurJobs> schedule;        // the level-by-level pyramid (k_blur)
    float* gauss[NUM_OCT][NLEV];
    float* mag[NUM_OCT][3]; float* ang[NUM_OCT][3];
    SiftDev d{};
    int* d_count = nullptr;
    std::vector<void*> allocations;
};

extern "C" {

int bf_sift_create(uint32_t width, uint32_t height, uint32_t depthWidth, uint32_t depthHeight, uint32_t featureCountThreshold,
                   float depthMin, float depthMax, float minKeyScale, uint32_t maxNumKeysPerImage, bf_sift** out) {
    BF_REQUIRE(out, "null argument");
    BF_REQUIRE(width >= 64 && height >= 64 && (width % 8) == 0 && (height % 8) == 0, "SIFT image must be >= 64x64 and a multiple of 8");
    BF_REQUIRE(maxNumKeysPerImage > 0, "maxNumKeysPerImage must be positive");
    bf_sift* s = new bf_sift();
    s->W = (int)width; s->H = (int)height; s->depthW = (int)depthWidth; s->depthH = (int)depthHeight;
    s->depthMin = depthMin; s->depthMax = depthMax; s->minKeyScale = minKeyScale;
    s->featureCountThreshold = (int)featureCountThreshold; s->maxFeatures = (int)maxNumKeysPerImage;
    // SiftParam::ParseSiftParam (SiftGPU.cpp:126-174): level_min -1, 3 DoG levels, level_max 4
    const float sigma0 = 1.6f * powf(2.0f, 1.0f / DOG_LEVELS);
    const float sigmak = powf(2.0f, 1.0f / DOG_LEVELS);
    const float dsigma0 = sigma0 * sqrtf(1.0f - 1.0f / (sigmak * sigmak));
    const float sa = sigma0 * powf(2.0f, -1.0f / (float)DOG_LEVELS), sb = 0.5f;
    const float initSigma = sa > sb + 0.001 ? sqrtf(sa * sa - sb * sb) : 0.0f;
    memset(&s->taps, 0, sizeof s->taps);
    makeTapsHost(initSigma, s->taps.fw[0], s->taps.k[0]);
    for (int i = 0; i < 5; ++i) makeTapsHost(dsigma0 * powf(sigmak, (float)i), s->taps.fw[i + 1], s->taps.k[i + 1]);
    auto A = [&](void** p, size_t bytes) { if (BF_MALLOC(p, bytes) != hipSuccess) return false; s->allocations.push_back(*p); return true; };
    bool ok = true;
    uint32_t candTotal = 0;
    for (int o = 0; o < NUM_OCT; ++o) {
        const int w = s->W >> o, h = s->H >> o;
        for (int a = 0; a < NLEV; ++a) ok = ok && A((void**)&s->gauss[o][a], (size_t)w * h * 4);
        for (int a = 0; a < 3; ++a) ok = ok && A((void**)&s->mag[o][a], (size_t)w * h * 4) && A((void**)&s->ang[o][a], (size_t)w * h * 4);
    }
    if (!ok) { set_error("bf_sift_create: hipMalloc failed"); bf_sift_destroy(s); return BF_ERR_HIP; }
    int gb = 0, db = 0;
    for (int o = 0; o < NUM_OCT; ++o) {
        const int w = s->W >> o, h = s->H >> o;
        for (int j = 1; j <= 3; ++j) {
            const int li = o * DOG_LEVELS + j - 1;
            LevelInfo& L = s->levels.l[li];
            for (int q = 0; q < 4; ++q) L.g[q] = s->gauss[o][j - 1 + q];
            L.w = w; L.h = h; L.octave = o;
            const int fm = (int)(w * h * 0.005f);
            L.fmax = fm > 4096 ? 4096 : (fm < 32 ? 32 : fm);
            L.cap = std::min(8192, std::max(128, L.fmax * 5));
            L.keyLocScale = (float)(1 << o);
            L.mag = s->mag[o][j - 1]; L.ang = s->ang[o][j - 1];
            L.sigma = sigma0 * powf(2.0f, (float)(j - 1) / (float)DOG_LEVELS);     // GetLevelSigma(j-1)
            s->d.candOff[li] = candTotal; candTotal += (uint32_t)L.cap;
            GradJob& G = s->gradJobs.j[li];
            G.g = s->gauss[o][j]; G.mag = s->mag[o][j - 1]; G.ang = s->ang[o][j - 1]; G.w = w; G.h = h; G.blocks = (w * h + 255) / 256;
            gb += G.blocks;
            s->detect.blocks[li] = ((w + 15) / 16) * ((h + 15) / 16);
            db += s->detect.blocks[li];
        }
    }
    s->gradBlocks = gb; s->detectBlocks = db;
    s->detect.W = s->W; s->detect.H = s->H; s->detect.depthW = s->depthW; s->detect.depthH = s->depthH;
    s->detect.depthMin = depthMin; s->detect.depthMax = depthMax;
    s->detect.dogThreshold = 0.02f / DOG_LEVELS;
    s->detect.edgeT = (10.0f + 1) * (10.0f + 1) / 10.0f;
    ok = A((void**)&s->d.cand, (size_t)candTotal * 4) && A((void**)&s->d.candCount, NKL * 4) && A((void**)&s->d.raw, MAXRAW * sizeof(RawKey)) &&
         A((void**)&s->d.counts, CNT_TOTAL * 4) && A((void**)&s->d.feats, (size_t)(maxNumKeysPerImage + 8) * sizeof(Feat)) &&
         A((void**)&s->d.des, (size_t)(maxNumKeysPerImage + 8) * 128 * 4) && A((void**)&s->d_count, 4) && A((void**)&s->d.ticket, 4);
    if (!ok) { set_error("bf_sift_create: hipMalloc failed"); bf_sift_destroy(s); return BF_ERR_HIP; }
    (void)hipMemset(s->d.ticket, 0, 4);
    (void)hipMemset(s->d.candCount, 0, NKL * 4);
    (void)hipMemset(s->d.counts, 0, CNT_TOTAL * 4);
    // wavefront schedule of the pyramid (SiftPyramid::BuildPyramid, SiftPyramid.cpp:82-145)
    struct J { BlurJob job; int depth; };
    std::vector<J> all;
    int depthOf[NUM_OCT][NLEV];
    for (int o = 0; o < NUM_OCT; ++o) {
        const int w = s->W >> o, h = s->H >> o;
        for (int a = 0; a < NLEV; ++a) {
            BlurJob b; memset(&b, 0, sizeof b);
            b.dst = s->gauss[o][a]; b.w = w; b.h = h;
            b.blocks = ((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H);
            int dep;
            if (a == 0 && o == 0) { b.src = nullptr; b.mode = 0; b.filter = 0; b.srcW = w; dep = 0; }
            else if (a == 0) { b.src = s->gauss[o - 1][3]; b.mode = 1; b.srcW = s->W >> (o - 1); dep = depthOf[o - 1][3] + 1; }
            else { b.src = s->gauss[o][a - 1]; b.mode = 0; b.filter = a; b.srcW = w; dep = depthOf[o][a - 1] + 1; }
            depthOf[o][a] = dep;
            all.push_back({b, dep});
        }
    }
    int maxDepth = 0;
    for (auto& j : all) maxDepth = std::max(maxDepth, j.depth);
    for (int dep = 0; dep <= maxDepth; ++dep) {
        BlurJobs bj; bj.n = 0;
        for (auto& j : all)
            if (j.depth == dep) {
                if (bj.n == 3) { s->schedule.push_back(bj); bj.n = 0; }
                bj.j[bj.n++] = j.job;
            }
        if (bj.n) s->schedule.push_back(bj);
    }
    *out = s;
    return BF_OK;
}

int bf_sift_destroy(bf_sift* s) {
    if (!s) return BF_OK;
    (void)hipStreamSynchronize(s->stream);
    for (void* p : s->allocations) (void)hipFree(p);
    delete s;
    return BF_OK;
}

int bf_sift_set_stream(bf_sift* s, void* st) { BF_REQUIRE(s, "null sift"); s->stream = (hipStream_t)st; return BF_OK; }

// SiftGPU::RunSIFT (SiftGPU.cpp:72-101) + GetKeyPointsAndDescriptorsCUDA (:267-272): fully asynchronous.
// d_keyPoints: float4 (x, y, scale, depth) per feature; d_descs: 128 B per feature; d_numKeys: device int
// (-1 if more than maxNumKeysPerImage features were found: "too many keypoints", Bundler.cpp:97).
int bf_sift_run(bf_sift* s, const float* d_intensity, const float* d_depth, float* d_keyPoints, uint8_t* d_descs, int32_t* d_numKeys) {
    BF_REQUIRE(s && d_intensity && d_depth && d_keyPoints && d_descs && d_numKeys, "null argument");
    hipStream_t st = s->stream;
    for (size_t i = 0; i < s->schedule.size(); ++i) {
        BlurJobs bj = s->schedule[i];
        int blocks = 0;
        for (int k = 0; k < bj.n; ++k) { if (bj.j[k].src == nullptr) bj.j[k].src = d_intensity; blocks += bj.j[k].blocks; }
        hipLaunchKernelGGL(k_blur, dim3(blocks), dim3(BLUR_THREADS), 0, st, bj, s->taps);
    }
    hipLaunchKernelGGL(k_detect, dim3(s->detectBlocks + s->gradBlocks), dim3(256), 0, st, s->levels, s->detect, s->d, d_depth, s->gradJobs, s->detectBlocks);
    hipLaunchKernelGGL(k_keys_finalize, dim3(NKL), dim3(1024), 0, st, s->levels, s->d, s->featureCountThreshold);
    hipLaunchKernelGGL(k_orientation, dim3(2048), dim3(64), 0, st, s->levels, s->d);
    hipLaunchKernelGGL(k_reshape, dim3(1), dim3(1024), 0, st, s->levels, s->d, s->minKeyScale, s->featureCountThreshold, s->maxFeatures);
    hipLaunchKernelGGL(k_descriptor, dim3(4 * (uint32_t)s->maxFeatures), dim3(256), 0, st, s->levels, s->d);
    hipLaunchKernelGGL(k_desc_finalize, dim3(div_up((uint32_t)s->maxFeatures, 8)), dim3(256), 0, st, s->levels, s->d, s->detect, d_depth, d_keyPoints,
                       d_descs, d_numKeys);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

// test hook: copy one gaussian pyramid level (octave, array index 0..5) to the host (syncs)
int bf_sift_debug_level(bf_sift* s, uint32_t octave, uint32_t index, float* h_out) {
    BF_REQUIRE(s && h_out && octave < NUM_OCT && index < NLEV, "bad argument");
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t n = (size_t)(s->W >> octave) * (s->H >> octave);
    BF_HIP_TRY(hipMemcpy(h_out, s->gauss[octave][index], n * 4, hipMemcpyDeviceToHost));
    return BF_OK;
}
// test hook: out[0]=numRaw out[1]=numFeat out[2..14)=level counts after the first limit out[14..26)=final
int bf_sift_debug_counts(bf_sift* s, int32_t out[26]) {
    BF_REQUIRE(s && out, "bad argument");
    int c[CNT_TOTAL];
    BF_HIP_TRY(hipStreamSynchronize(s->stream));
    BF_HIP_TRY(hipMemcpy(c, s->d.counts, sizeof c, hipMemcpyDeviceToHost));
    out[0] = c[CNT_RAW]; out[1] = c[CNT_FEAT];
    for (int i = 0; i < NKL; ++i) { out[2 + i] = c[CNT_LEVEL0 + i]; out[14 + i] = c[CNT_LEVEL1 + i]; }
    return BF_OK;
}

}  // extern "C"
