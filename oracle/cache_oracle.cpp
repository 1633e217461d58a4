// TEST INFRASTRUCTURE ONLY (see or_common.h) — CPU restatement of the image operators used on
// the hot path and of CUDACache::storeFrame.
//   CUDAImageUtil.cu:93-124 (resampleFloat), :126-157 (resampleFloat4), :214-245 (resampleToIntensity),
//   :260-300 (computeIntensityDerivatives), :367-385 (convertDepthFloatToCameraSpaceFloat4),
//   :404-433 (computeNormals), :497-514 (convertNormalsFloat4ToUCHAR4), :701-742 (erodeDepthMap),
//   :759-796 (gaussFilterDepthMap), :811-846 (gaussFilterIntensity);  CUDACache.cpp:14-86.
// PINNED to the reference's kernels of CUDAImageUtil.cu through oracle/_ref, bit for bit (tests/test_ref_pin_cpu.py::
// test_ingest_and_resample_kernels, ::test_cache_store_frame_vs_reference_kernels).  The Gaussian taps exp(-(dx^2+dy^2)/(2 sigma^2)) are evaluated once on the
// host with expf (the reference evaluates __expf per tap in the kernel, fast-math).
#include <cstdlib>
#include <vector>

#include "../include/bf_hip.h"
#include "or_common.h"

using namespace orc;

namespace {

struct f4 { float x, y, z, w; };
struct f2 { float x, y; };

inline float gaussD(float sigma, int x, int y) { return expf(-((float)(x * x + y * y) / (2.0f * sigma * sigma))); }

void resampleIdx(unsigned x, unsigned y, unsigned ow, unsigned oh, unsigned iw, unsigned ih, unsigned& xi, unsigned& yi) {
    const float sw = (float)(iw - 1) / (float)(ow - 1);
    const float sh = (float)(ih - 1) / (float)(oh - 1);
    xi = (unsigned)f2i((float)x * sw + 0.5f);
    yi = (unsigned)f2i((float)y * sh + 0.5f);
}

}  // namespace

namespace orc { int g_threads = 1; }

extern "C" {

void or_set_threads(int n) { orc::g_threads = n < 1 ? 1 : n; }
int or_get_threads() { return orc::g_threads; }

void or_erode_depth(float* out, const float* in, int structureSize, int w, int h, float dThresh, float fracReq) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned count = 0;
            const float old = in[y * w + x];
            for (int i = -structureSize; i <= structureSize; ++i)
                for (int j = -structureSize; j <= structureSize; ++j)
                    if (x + j >= 0 && x + j < w && y + i >= 0 && y + i < h) {
                        const float d = in[(y + i) * w + (x + j)];
                        if (d == MINF || d == 0.0f || fabsf(d - old) > dThresh) count++;
                    }
            const unsigned sum = (2 * structureSize + 1) * (2 * structureSize + 1);
            out[y * w + x] = ((float)count / (float)sum >= fracReq) ? MINF : in[y * w + x];
        }
}

void or_gauss_filter_depth(float* out, const float* in, float sigmaD, float sigmaR, unsigned w, unsigned h) {
    const int r = (int)ceil(2.0 * sigmaD);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (int y = 0; y < (int)h; ++y)
        for (int x = 0; x < (int)w; ++x) {
            float sum = 0.0f, sumW = 0.0f;
            out[y * w + x] = MINF;
            const float c = in[y * w + x];
            if (c != MINF)
                for (int m = x - r; m <= x + r; ++m)
                    for (int n = y - r; n <= y + r; ++n)
                        if (m >= 0 && n >= 0 && m < (int)w && n < (int)h) {
                            const float d = in[n * w + m];
                            if (d != MINF && fabsf(c - d) < sigmaR) {
                                const float wt = gaussD(sigmaD, m - x, n - y);
                                sumW += wt;
                                sum += wt * d;
                            }
                        }
            if (sumW > 0.0f) out[y * w + x] = sum / sumW;
        }
}

void or_gauss_filter_intensity(float* out, const float* in, float sigmaD, unsigned w, unsigned h) {
    const int r = (int)ceil(2.0 * sigmaD);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (int y = 0; y < (int)h; ++y)
        for (int x = 0; x < (int)w; ++x) {
            float sum = 0.0f, sumW = 0.0f;
            for (int m = x - r; m <= x + r; ++m)
                for (int n = y - r; n <= y + r; ++n)
                    if (m >= 0 && n >= 0 && m < (int)w && n < (int)h) {
                        const float wt = gaussD(sigmaD, m - x, n - y);
                        sumW += wt;
                        sum += wt * in[n * w + m];
                    }
            if (sumW > 0.0f) out[y * w + x] = sum / sumW;
        }
}

void or_resample_float(float* out, unsigned ow, unsigned oh, const float* in, unsigned iw, unsigned ih) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < oh; ++y)
        for (unsigned x = 0; x < ow; ++x) {
            unsigned xi, yi;
            resampleIdx(x, y, ow, oh, iw, ih, xi, yi);
            if (xi < iw && yi < ih) out[y * ow + x] = in[yi * iw + xi];
        }
}

// resampleUCHAR4_Kernel, CUDAImageUtil.cu:160-177 (colour to the integration resolution, CUDAImageManager.cpp:72-78)
void or_resample_uchar4(uint8_t* out, unsigned ow, unsigned oh, const uint8_t* in, unsigned iw, unsigned ih) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < oh; ++y)
        for (unsigned x = 0; x < ow; ++x) {
            unsigned xi, yi;
            resampleIdx(x, y, ow, oh, iw, ih, xi, yi);
            if (xi < iw && yi < ih) memcpy(out + 4 * ((size_t)y * ow + x), in + 4 * ((size_t)yi * iw + xi), 4);
        }
}

float or_intensity(const uint8_t* c) { return (0.299f * (float)c[0] + 0.587f * (float)c[1] + 0.114f * (float)c[2]) / 255.0f; }

void or_resample_to_intensity(float* out, unsigned ow, unsigned oh, const uint8_t* in, unsigned iw, unsigned ih) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < oh; ++y)
        for (unsigned x = 0; x < ow; ++x) {
            unsigned xi, yi;
            resampleIdx(x, y, ow, oh, iw, ih, xi, yi);
            if (xi < iw && yi < ih) out[y * ow + x] = or_intensity(in + 4 * ((size_t)yi * iw + xi));
        }
}

void or_intensity_derivatives(float* out2, const float* in, unsigned w, unsigned h) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < h; ++y)
        for (unsigned x = 0; x < w; ++x) {
            float* o = out2 + 2 * (y * w + x);
            o[0] = MINF; o[1] = MINF;
            if (x > 0 && x < w - 1 && y > 0 && y < h - 1) {
                const float p00 = in[(y - 1) * w + (x - 1)], p01 = in[y * w + (x - 1)], p02 = in[(y + 1) * w + (x - 1)];
                const float p10 = in[(y - 1) * w + x], p12 = in[(y + 1) * w + x];
                const float p20 = in[(y - 1) * w + (x + 1)], p21 = in[y * w + (x + 1)], p22 = in[(y + 1) * w + (x + 1)];
                if (p00 == MINF || p01 == MINF || p02 == MINF || p10 == MINF || p12 == MINF || p20 == MINF || p21 == MINF || p22 == MINF) continue;
                float u = (-1.0f) * p00 + (1.0f) * p20 + (-2.0f) * p01 + (2.0f) * p21 + (-1.0f) * p02 + (1.0f) * p22;
                u /= 8.0f;
                float v = (-1.0f) * p00 + (-2.0f) * p10 + (-1.0f) * p20 + (1.0f) * p02 + (2.0f) * p12 + (1.0f) * p22;
                v /= 8.0f;
                o[0] = u; o[1] = v;
            }
        }
}

void or_depth_to_campos(float* out4, const float* in, const float* intrinsicsInv, unsigned w, unsigned h) {
    const float* M = intrinsicsInv;
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < h; ++y)
        for (unsigned x = 0; x < w; ++x) {
            float* o = out4 + 4 * (y * w + x);
            o[0] = o[1] = o[2] = o[3] = MINF;
            const float d = in[y * w + x];
            if (d != MINF) {
                const float vx = (float)x * d, vy = (float)y * d, vz = d, vw = d;
                const float cx = M[0] * vx + M[1] * vy + M[2] * vz + M[3] * vw;
                const float cy = M[4] * vx + M[5] * vy + M[6] * vz + M[7] * vw;
                const float cw = M[12] * vx + M[13] * vy + M[14] * vz + M[15] * vw;
                o[0] = cx; o[1] = cy; o[2] = cw; o[3] = 1.0f;
            }
        }
}

void or_compute_normals(float* out4, const float* in4, unsigned w, unsigned h) {
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < h; ++y)
        for (unsigned x = 0; x < w; ++x) {
            float* o = out4 + 4 * (y * w + x);
            o[0] = o[1] = o[2] = o[3] = MINF;
            if (x > 0 && x < w - 1 && y > 0 && y < h - 1) {
                const float* CC = in4 + 4 * (y * w + x);
                const float* PC = in4 + 4 * ((y + 1) * w + x);
                const float* CP = in4 + 4 * (y * w + x + 1);
                const float* MC = in4 + 4 * ((y - 1) * w + x);
                const float* CM = in4 + 4 * (y * w + x - 1);
                if (CC[0] != MINF && PC[0] != MINF && CP[0] != MINF && MC[0] != MINF && CM[0] != MINF) {
                    const f3 a = {PC[0] - MC[0], PC[1] - MC[1], PC[2] - MC[2]};
                    const f3 b = {CP[0] - CM[0], CP[1] - CM[1], CP[2] - CM[2]};
                    const f3 n = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
                    const float l = sqrtf(dot(n, n));
                    if (l > 0.0f) { o[0] = n.x / -l; o[1] = n.y / -l; o[2] = n.z / -l; o[3] = 0.0f; }
                }
            }
        }
}

void or_normals_to_uchar4(uint8_t* out, const float* in4, unsigned w, unsigned h) {
    for (unsigned i = 0; i < w * h; ++i) {
        uint8_t* o = out + 4 * i;
        o[0] = o[1] = o[2] = o[3] = 0;
        const float* p = in4 + 4 * i;
        if (p[0] != MINF)
            for (int k = 0; k < 3; ++k) o[k] = (uint8_t)f2i(roundf(((p[k] + 1.0f) / 2.0f) * 255));
    }
}

// CUDACache::storeFrame (CUDACache.cpp:45-86).  Output: the six arrays of one CUDACachedFrame.
void or_cache_store_frame(const float* depth, unsigned dw, unsigned dh, const uint8_t* color, unsigned cw, unsigned ch,
                          unsigned W, unsigned H, const float* inputIntrinsicsInv, float sigmaIntensity, float sigmaD, float sigmaR,
                          float* depthDown, float* camposDown4, float* intensityDown, float* intensityDerivs2,
                          uint8_t* normalsU4, float* normalsDown4) {
    std::vector<float> filt((size_t)dw * dh), campos((size_t)dw * dh * 4), normals((size_t)dw * dh * 4), inten((size_t)W * H);
    const float* din = depth;
    if (sigmaD > 0.0f) { or_gauss_filter_depth(filt.data(), depth, sigmaD, sigmaR, dw, dh); din = filt.data(); }
    or_depth_to_campos(campos.data(), din, inputIntrinsicsInv, dw, dh);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            unsigned xi, yi;
            resampleIdx(x, y, W, H, dw, dh, xi, yi);
            if (xi < dw && yi < dh) memcpy(camposDown4 + 4 * (y * W + x), campos.data() + 4 * ((size_t)yi * dw + xi), 16);
        }
    or_compute_normals(normals.data(), campos.data(), dw, dh);
#pragma omp parallel for num_threads(orc::g_threads) schedule(static)
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            unsigned xi, yi;
            resampleIdx(x, y, W, H, dw, dh, xi, yi);
            if (xi < dw && yi < dh) memcpy(normalsDown4 + 4 * (y * W + x), normals.data() + 4 * ((size_t)yi * dw + xi), 16);
        }
    or_normals_to_uchar4(normalsU4, normalsDown4, W, H);
    or_resample_float(depthDown, W, H, din, dw, dh);
    or_resample_to_intensity(inten.data(), W, H, color, cw, ch);
    if (sigmaIntensity > 0.0f) or_gauss_filter_intensity(intensityDown, inten.data(), sigmaIntensity, W, H);
    else memcpy(intensityDown, inten.data(), sizeof(float) * W * H);
    or_intensity_derivatives(intensityDerivs2, intensityDown, W, H);
}

}  // extern "C"
