// Dense frame cache for gfx950 (replaces CUDACache.{h,cpp} + the CUDAImageUtil.cu operators it
// calls; paths relative to /root/reference/FriedLiver/Source) behind the bf_cache_* C ABI.
//
// The reference runs 9 full-resolution kernels per frame (5x5 range-gated Gaussian on depth,
// camera-space positions, normals, three point-sampling resamplers, ...) and then keeps only the
// W x H = 80 x 60 point samples.  Here ONE kernel evaluates exactly those samples: a thread per
// cached pixel computes the filtered depth at its sample position and its 4-neighbourhood (the only
// filtered depths the normal at that sample depends on), i.e. 125 taps instead of a 640x480x25
// sweep (~12x less traffic, 1 launch).  A second kernel (one workgroup per 16x12 tile, chain evaluated in LDS on tile + halo) does the intensity chain
// (sample -> 11x11 Gaussian -> Sobel).  Per-pixel arithmetic and summation order are the reference's
// (taps summed x-outer / y-inner), so results are bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "bf_device.h"
#include "bf_internal.h"

using namespace bf;

namespace {

constexpr int MAX_R = 6;
struct Taps { int r; float w[(2 * MAX_R + 1) * (2 * MAX_R + 1)]; };

struct CacheGeom {
    uint32_t W, H, dw, dh, cw, ch;
    float inv[16];          // inverse of the INPUT depth intrinsics
    float sigmaR;
    int useDepthFilter;
};

BF_DEV void sampleIdx(uint32_t x, uint32_t y, uint32_t ow, uint32_t oh, uint32_t iw, uint32_t ih, uint32_t& xi, uint32_t& yi) {
    const float sw = (float)(iw - 1) / (float)(ow - 1);      // CUDAImageUtil.cu:100-104
    const float sh = (float)(ih - 1) / (float)(oh - 1);
    xi = (uint32_t)f2i((float)x * sw + 0.5f);
    yi = (uint32_t)f2i((float)y * sh + 0.5f);
}

// gaussFilterDepthMapDevice (CUDAImageUtil.cu:759-796) evaluated at one pixel
BF_DEV float filteredDepth(const float* __restrict__ d, const CacheGeom& g, const Taps& t, int x, int y) {
    const float c = d[(size_t)y * g.dw + x];
    if (!g.useDepthFilter) return c;
    if (c == BF_MINF) return BF_MINF;
    float sum = 0.0f, sumW = 0.0f;
    const int r = t.r, n = 2 * r + 1;
    for (int m = x - r; m <= x + r; ++m)
        for (int k = y - r; k <= y + r; ++k)
            if (m >= 0 && k >= 0 && m < (int)g.dw && k < (int)g.dh) {
                const float v = d[(size_t)k * g.dw + m];
                if (v != BF_MINF && fabsf(c - v) < g.sigmaR) {
                    const float w = t.w[(m - x + r) * n + (k - y + r)];
                    sumW += w;
                    sum += w * v;
                }
            }
    return sumW > 0.0f ? sum / sumW : BF_MINF;
}

// convertDepthFloatToCameraSpaceFloat4_Kernel (CUDAImageUtil.cu:367-385)
BF_DEV float4 camPos(const CacheGeom& g, int x, int y, float depth) {
    if (depth == BF_MINF) return make_float4(BF_MINF, BF_MINF, BF_MINF, BF_MINF);
    const float vx = (float)x * depth, vy = (float)y * depth, vz = depth, vw = depth;
    const float cx = g.inv[0] * vx + g.inv[1] * vy + g.inv[2] * vz + g.inv[3] * vw;
    const float cy = g.inv[4] * vx + g.inv[5] * vy + g.inv[6] * vz + g.inv[7] * vw;
    const float cw = g.inv[12] * vx + g.inv[13] * vy + g.inv[14] * vz + g.inv[15] * vw;
    return make_float4(cx, cy, cw, 1.0f);
}

// One output pixel needs up to five filtered depths (its own sample and the four neighbours of the normal's central differences), each a (2r+1)^2 window over the
// full-resolution depth map: five threads per pixel evaluate one each (round 6; one thread evaluated all five, 58 us per frame for 4800 pixels on 19 workgroups),
// the pixel's first thread finishes with the reference's sequence on those values.  filteredDepth is a pure function: same bits.
constexpr int CG_PIX = 64;      // pixels per workgroup (one wave per role)
__global__ __launch_bounds__(CG_PIX * 5) void k_cache_geometry(CacheGeom g, Taps t, const float* __restrict__ depth, bf_cached_frame f) {
    __shared__ float dep[5][CG_PIX];
    const uint32_t px = threadIdx.x, role = threadIdx.y;          // role 0 the pixel itself, 1 (x, y+1), 2 (x+1, y), 3 (x, y-1), 4 (x-1, y)
    const uint32_t idx = blockIdx.x * CG_PIX + px;
    uint32_t xi = 0, yi = 0;
    bool live = idx < g.W * g.H;
    if (live) {
        sampleIdx(idx % g.W, idx / g.W, g.W, g.H, g.dw, g.dh, xi, yi);
        live = xi < g.dw && yi < g.dh;
    }
    const bool interior = live && xi > 0 && xi < g.dw - 1 && yi > 0 && yi < g.dh - 1;
    if (role == 0 ? live : interior) {
        const int ox = role == 2 ? 1 : role == 4 ? -1 : 0, oy = role == 1 ? 1 : role == 3 ? -1 : 0;
        dep[role][px] = filteredDepth(depth, g, t, (int)xi + ox, (int)yi + oy);
    }
    __syncthreads();
    if (role != 0 || !live) return;
    const float dC = dep[0][px];
    f.d_depthDownsampled[idx] = dC;                                           // resampleFloat :93
    const float4 CC = camPos(g, (int)xi, (int)yi, dC);
    reinterpret_cast<float4*>(f.d_cameraposDownsampled)[idx] = CC;           // resampleFloat4 :126
    float4 nrm = make_float4(BF_MINF, BF_MINF, BF_MINF, BF_MINF);            // computeNormals_Kernel :404-433
    if (interior && CC.x != BF_MINF) {
        const float4 PC = camPos(g, (int)xi, (int)yi + 1, dep[1][px]);
        const float4 CP = camPos(g, (int)xi + 1, (int)yi, dep[2][px]);
        const float4 MC = camPos(g, (int)xi, (int)yi - 1, dep[3][px]);
        const float4 CM = camPos(g, (int)xi - 1, (int)yi, dep[4][px]);
        if (PC.x != BF_MINF && CP.x != BF_MINF && MC.x != BF_MINF && CM.x != BF_MINF) {
            const f3 a = mk3(PC.x - MC.x, PC.y - MC.y, PC.z - MC.z);
            const f3 b = mk3(CP.x - CM.x, CP.y - CM.y, CP.z - CM.z);
            const f3 n = cross3(a, b);
            const float l = sqrtf(dot3(n, n));
            if (l > 0.0f) nrm = make_float4(n.x / -l, n.y / -l, n.z / -l, 0.0f);
        }
    }
    reinterpret_cast<float4*>(f.d_normalsDownsampled)[idx] = nrm;
    uchar4 nu = make_uchar4(0, 0, 0, 0);                                      // convertNormalsFloat4ToUCHAR4 :497-514
    if (nrm.x != BF_MINF) {
        nu.x = (unsigned char)f2i(roundf(((nrm.x + 1.0f) / 2.0f) * 255));
        nu.y = (unsigned char)f2i(roundf(((nrm.y + 1.0f) / 2.0f) * 255));
        nu.z = (unsigned char)f2i(roundf(((nrm.z + 1.0f) / 2.0f) * 255));
    }
    reinterpret_cast<uchar4*>(f.d_normalsDownsampledUCHAR4)[idx] = nu;
}

// resampleToIntensity (:224) -> gaussFilterIntensity (:811) -> computeIntensityDerivatives (:260) for one IT_W x IT_H tile of
// the W x H cache image per workgroup.  The chain is evaluated in LDS on the tile plus its halo (Sobel needs 1 ring of
// filtered values, the filter needs r rings of samples), so the tiles are independent and no pass over global memory
// separates the three operators; every value is produced by the same tap order as a whole-image pass.
constexpr int IT_W = 16, IT_H = 12;
__global__ __launch_bounds__(256) void k_cache_intensity(CacheGeom g, Taps t, int useFilter, const uchar4* __restrict__ color, bf_cached_frame f) {
    constexpr int MAXS = (IT_W + 2 * (MAX_R + 1)) * (IT_H + 2 * (MAX_R + 1));
    __shared__ float sSamp[MAXS];
    __shared__ float sFilt[(IT_W + 2) * (IT_H + 2)];
    const int W = (int)g.W, H = (int)g.H;
    const int r = useFilter ? t.r : 0, nt = 2 * r + 1;
    const int x0 = (int)blockIdx.x * IT_W, y0 = (int)blockIdx.y * IT_H;
    const int sw = IT_W + 2 * (r + 1), sh = IT_H + 2 * (r + 1);
    for (int i = (int)threadIdx.x; i < sw * sh; i += 256) {                    // samples (x0-r-1 .. ) of the point-sampled intensity
        const int x = x0 - r - 1 + i % sw, y = y0 - r - 1 + i / sw;
        float v = 0.0f;
        if (x >= 0 && y >= 0 && x < W && y < H) {
            uint32_t xi, yi;
            sampleIdx((uint32_t)x, (uint32_t)y, g.W, g.H, g.cw, g.ch, xi, yi);
            if (xi < g.cw && yi < g.ch) {
                const uchar4 c = color[(size_t)yi * g.cw + xi];
                v = (0.299f * (float)c.x + 0.587f * (float)c.y + 0.114f * (float)c.z) / 255.0f;
            }
        }
        sSamp[i] = v;
    }
    __syncthreads();
    const int fw = IT_W + 2, fh = IT_H + 2;
    for (int i = (int)threadIdx.x; i < fw * fh; i += 256) {                    // filtered values of the tile + 1 ring
        const int x = x0 - 1 + i % fw, y = y0 - 1 + i / fw;
        float v = 0.0f;
        if (x >= 0 && y >= 0 && x < W && y < H) {
            if (useFilter) {
                float sum = 0.0f, sumW = 0.0f;
                for (int m = x - r; m <= x + r; ++m)
                    for (int k = y - r; k <= y + r; ++k)
                        if (m >= 0 && k >= 0 && m < W && k < H) {
                            const float w = t.w[(m - x + r) * nt + (k - y + r)];
                            sumW += w;
                            sum += w * sSamp[(k - (y0 - r - 1)) * sw + (m - (x0 - r - 1))];
                        }
                v = sum / sumW;                                               // sumW > 0: the centre tap is always inside
            } else v = sSamp[(y - (y0 - r - 1)) * sw + (x - (x0 - r - 1))];
            if (x >= x0 && x < x0 + IT_W && y >= y0 && y < y0 + IT_H) f.d_intensityDownsampled[y * W + x] = v;
        }
        sFilt[i] = v;
    }
    __syncthreads();
    float2* out = reinterpret_cast<float2*>(f.d_intensityDerivsDownsampled);
    for (int i = (int)threadIdx.x; i < IT_W * IT_H; i += 256) {
        const int x = x0 + i % IT_W, y = y0 + i / IT_W;
        if (x >= W || y >= H) continue;
        float2 o = make_float2(BF_MINF, BF_MINF);
        if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
#define BF_P(dx, dy) sFilt[(y + (dy) - (y0 - 1)) * fw + (x + (dx) - (x0 - 1))]
            const float p00 = BF_P(-1, -1), p01 = BF_P(-1, 0), p02 = BF_P(-1, 1);
            const float p10 = BF_P(0, -1), p12 = BF_P(0, 1);
            const float p20 = BF_P(1, -1), p21 = BF_P(1, 0), p22 = BF_P(1, 1);
#undef BF_P
            if (!(p00 == BF_MINF || p01 == BF_MINF || p02 == BF_MINF || p10 == BF_MINF || p12 == BF_MINF || p20 == BF_MINF ||
                  p21 == BF_MINF || p22 == BF_MINF)) {
                float u = (-1.0f) * p00 + (1.0f) * p20 + (-2.0f) * p01 + (2.0f) * p21 + (-1.0f) * p02 + (1.0f) * p22;
                u /= 8.0f;
                float v = (-1.0f) * p00 + (-2.0f) * p10 + (-1.0f) * p20 + (1.0f) * p02 + (2.0f) * p12 + (1.0f) * p22;
                v /= 8.0f;
                o = make_float2(u, v);
            }
        }
        out[y * W + x] = o;
    }
}

Taps makeTaps(float sigma) {
    Taps t;
    memset(&t, 0, sizeof t);
    t.r = (int)ceil(2.0 * sigma);
    const int n = 2 * t.r + 1;
    for (int dx = -t.r; dx <= t.r; ++dx)
        for (int dy = -t.r; dy <= t.r; ++dy)
            t.w[(dx + t.r) * n + (dy + t.r)] = expf(-((float)(dx * dx + dy * dy) / (2.0f * sigma * sigma)));
    return t;
}

}  // namespace

struct bf_cache {
    uint32_t W, H, dw, dh, maxImages, current = 0;
    m44 intrinsics, inputIntrinsics, inputIntrinsicsInv;
    float sigmaIntensity, sigmaD, sigmaR;
    Taps tapsDepth, tapsIntensity;
    std::vector<bf_cached_frame> frames;
    bf_cached_frame* d_frames = nullptr;
    float* d_scratch = nullptr;
    std::vector<void*> allocations;
    hipStream_t stream = nullptr;
};

extern "C" {

int bf_cache_create(uint32_t dw, uint32_t dh, uint32_t W, uint32_t H, uint32_t maxNumImages, const float K[16], float colorDownSigma,
                    float depthDownSigmaD, float depthDownSigmaR, bf_cache** out) {
    BF_REQUIRE(K && out, "null argument");
    BF_REQUIRE(W > 1 && H > 1 && dw > 1 && dh > 1 && maxNumImages > 0, "bad sizes");
    BF_REQUIRE(ceil(2.0 * colorDownSigma) <= MAX_R && ceil(2.0 * depthDownSigmaD) <= MAX_R, "filter radius > 6 not supported");
    bf_cache* c = new bf_cache();
    c->W = W; c->H = H; c->dw = dw; c->dh = dh; c->maxImages = maxNumImages;
    memcpy(c->inputIntrinsics.e, K, 64);
    c->intrinsics = c->inputIntrinsics;                                  // CUDACache.cpp:20-24
    c->intrinsics.e[0] *= (float)W / (float)dw;
    c->intrinsics.e[5] *= (float)H / (float)dh;
    c->intrinsics.e[2] *= (float)(W - 1) / (float)(dw - 1);
    c->intrinsics.e[6] *= (float)(H - 1) / (float)(dh - 1);
    c->inputIntrinsicsInv = inverse44(c->inputIntrinsics);
    c->sigmaIntensity = colorDownSigma; c->sigmaD = depthDownSigmaD; c->sigmaR = depthDownSigmaR;
    if (depthDownSigmaD > 0) c->tapsDepth = makeTaps(depthDownSigmaD); else memset(&c->tapsDepth, 0, sizeof(Taps));
    if (colorDownSigma > 0) c->tapsIntensity = makeTaps(colorDownSigma); else memset(&c->tapsIntensity, 0, sizeof(Taps));
    const size_t n = (size_t)W * H;
    // one slab per array type: frame i at offset i*n  (288 GB of HBM: 1200 frames are 300 MB)
    float *depth, *campos, *inten, *derivs, *normals; uint8_t* nu;
    auto A = [&](void** p, size_t bytes) { if (BF_MALLOC(p, bytes) != hipSuccess) return false; c->allocations.push_back(*p); return true; };
    bool ok = A((void**)&depth, n * 4 * maxNumImages) && A((void**)&campos, n * 16 * maxNumImages) && A((void**)&inten, n * 4 * maxNumImages) &&
              A((void**)&derivs, n * 8 * maxNumImages) && A((void**)&nu, n * 4 * maxNumImages) && A((void**)&normals, n * 16 * maxNumImages) &&
              A((void**)&c->d_frames, sizeof(bf_cached_frame) * maxNumImages) && A((void**)&c->d_scratch, n * 4);
    if (!ok) { set_error("bf_cache_create: hipMalloc failed"); bf_cache_destroy(c); return BF_ERR_HIP; }
    c->frames.resize(maxNumImages);
    for (uint32_t i = 0; i < maxNumImages; ++i) {
        bf_cached_frame& f = c->frames[i];
        f.d_depthDownsampled = depth + n * i;
        f.d_cameraposDownsampled = campos + 4 * n * i;
        f.d_intensityDownsampled = inten + n * i;
        f.d_intensityDerivsDownsampled = derivs + 2 * n * i;
        f.d_normalsDownsampledUCHAR4 = nu + 4 * n * i;
        f.d_normalsDownsampled = normals + 4 * n * i;
    }
    BF_HIP_TRY(hipMemcpy(c->d_frames, c->frames.data(), sizeof(bf_cached_frame) * maxNumImages, hipMemcpyHostToDevice));
    *out = c;
    return BF_OK;
}

int bf_cache_destroy(bf_cache* c) {
    if (!c) return BF_OK;
    (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocations) (void)hipFree(p);
    delete c;
    return BF_OK;
}

int bf_cache_set_stream(bf_cache* c, void* s) { BF_REQUIRE(c, "null cache"); c->stream = (hipStream_t)s; return BF_OK; }

int bf_cache_store_frame(bf_cache* c, const float* d_depth, uint32_t dw, uint32_t dh, const uint8_t* d_color, uint32_t cw, uint32_t ch) {
    BF_REQUIRE(c && d_depth && d_color, "null argument");
    BF_REQUIRE(c->current < c->maxImages, "CUDACache reached max # images");
    BF_REQUIRE(dw == c->dw && dh == c->dh, "depth size differs from the cache's input size");
    CacheGeom g;
    g.W = c->W; g.H = c->H; g.dw = dw; g.dh = dh; g.cw = cw; g.ch = ch;
    memcpy(g.inv, c->inputIntrinsicsInv.e, 64);
    g.sigmaR = c->sigmaR;
    g.useDepthFilter = c->sigmaD > 0.0f;
    const bf_cached_frame f = c->frames[c->current];
    const uint32_t n = c->W * c->H;
    hipLaunchKernelGGL(k_cache_geometry, dim3(div_up(n, (uint32_t)CG_PIX)), dim3(CG_PIX, 5), 0, c->stream, g, c->tapsDepth, d_depth, f);
    hipLaunchKernelGGL(k_cache_intensity, dim3(div_up(c->W, IT_W), div_up(c->H, IT_H)), dim3(256), 0, c->stream, g, c->tapsIntensity,
                       (int)(c->sigmaIntensity > 0.0f), reinterpret_cast<const uchar4*>(d_color), f);
    BF_HIP_TRY(hipGetLastError());
    c->current++;
    return BF_OK;
}

int bf_cache_reset(bf_cache* c) { BF_REQUIRE(c, "null cache"); c->current = 0; return BF_OK; }

int bf_cache_copy_cache_frame_from(bf_cache* c, bf_cache* other, uint32_t frameFrom) {
    BF_REQUIRE(c && other, "null cache");
    BF_REQUIRE(c->current < c->maxImages && frameFrom < other->maxImages, "frame index out of range");
    BF_REQUIRE(c->W == other->W && c->H == other->H, "cache geometry differs");
    const size_t n = (size_t)c->W * c->H;
    const bf_cached_frame &d = c->frames[c->current], &s = other->frames[frameFrom];
    BF_HIP_TRY(hipMemcpyAsync(d.d_depthDownsampled, s.d_depthDownsampled, n * 4, hipMemcpyDeviceToDevice, c->stream));
    BF_HIP_TRY(hipMemcpyAsync(d.d_cameraposDownsampled, s.d_cameraposDownsampled, n * 16, hipMemcpyDeviceToDevice, c->stream));
    BF_HIP_TRY(hipMemcpyAsync(d.d_intensityDownsampled, s.d_intensityDownsampled, n * 4, hipMemcpyDeviceToDevice, c->stream));
    BF_HIP_TRY(hipMemcpyAsync(d.d_intensityDerivsDownsampled, s.d_intensityDerivsDownsampled, n * 8, hipMemcpyDeviceToDevice, c->stream));
    BF_HIP_TRY(hipMemcpyAsync(d.d_normalsDownsampledUCHAR4, s.d_normalsDownsampledUCHAR4, n * 4, hipMemcpyDeviceToDevice, c->stream));
    BF_HIP_TRY(hipMemcpyAsync(d.d_normalsDownsampled, s.d_normalsDownsampled, n * 16, hipMemcpyDeviceToDevice, c->stream));
    c->current++;
    return BF_OK;
}

int bf_cache_increment(bf_cache* c) { BF_REQUIRE(c, "null cache"); c->current++; return BF_OK; }
int bf_cache_get_num_frames(bf_cache* c, uint32_t* out) { BF_REQUIRE(c && out, "null argument"); *out = c->current; return BF_OK; }
int bf_cache_set_current_frame(bf_cache* c, uint32_t n) { BF_REQUIRE(c && n <= c->maxImages, "bad frame count"); c->current = n; return BF_OK; }
int bf_cache_get_frames_gpu(bf_cache* c, const bf_cached_frame** d) { BF_REQUIRE(c && d, "null argument"); *d = c->d_frames; return BF_OK; }
int bf_cache_get_frame(bf_cache* c, uint32_t i, bf_cached_frame* out) {
    BF_REQUIRE(c && out && i < c->maxImages, "bad argument");
    *out = c->frames[i];
    return BF_OK;
}
int bf_cache_get_geometry(bf_cache* c, uint32_t* w, uint32_t* h, float k4[4]) {
    BF_REQUIRE(c && w && h && k4, "null argument");
    *w = c->W; *h = c->H;
    k4[0] = c->intrinsics.e[0]; k4[1] = c->intrinsics.e[5]; k4[2] = c->intrinsics.e[2]; k4[3] = c->intrinsics.e[6];
    return BF_OK;
}

}  // extern "C"
