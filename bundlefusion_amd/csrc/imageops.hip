// Frame-ingest image operators for gfx950 (the CUDAImageUtil.cu operators CUDAImageManager::process and
// OnlineBundler::getCurrentFrame call; paths relative to /root/reference/FriedLiver/Source):
//   erodeDepthMap :701-757, gaussFilterDepthMap :759-809, gaussFilterIntensity :811-859,
//   resampleFloat :93-124, resampleUCHAR4 :160-191, resampleToIntensity :201-258.
// Streaming HBM-bound stencils: one thread per output pixel, 64x4 tiles so a wave reads 256 B rows; the
// neighbourhoods (<= 9x9) stay in L1/L2.  Gaussian taps are tabulated on the host (expf once per tap instead of
// once per pixel and tap) and summed in the reference's order (x outer, y inner), which makes the result the
// same IEEE sequence as the CPU restatement.
#include <hip/hip_runtime.h>

#include <cmath>

#include "bf_device.h"
#include "bf_internal.h"

using namespace bf;

namespace {

constexpr int MAX_R = 8;
struct Taps { int r; float w[(2 * MAX_R + 1) * (2 * MAX_R + 1)]; };

Taps makeTaps(float sigma, int r) {
    Taps t;
    t.r = r;
    const int n = 2 * r + 1;
    for (int x = -r; x <= r; ++x)
        for (int y = -r; y <= r; ++y) t.w[(x + r) * n + (y + r)] = expf(-((float)(x * x + y * y) / (2.0f * sigma * sigma)));
    return t;
}

// csrc / cdst1 / cdst2 (optional): a pixel-wise copy of a 4-byte-per-pixel image of the same size riding along (the ingest's colour copies: one launch less per copy)
__global__ __launch_bounds__(256) void k_erode(float* __restrict__ out, const float* __restrict__ in, int s, int w, int h, float dThresh, float fracReq,
                                               const uint32_t* __restrict__ csrc, uint32_t* __restrict__ cdst1, uint32_t* __restrict__ cdst2) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    if (csrc) { const uint32_t c = csrc[y * w + x]; if (cdst1) cdst1[y * w + x] = c; if (cdst2) cdst2[y * w + x] = c; }
    unsigned count = 0;
    const float old = in[y * w + x];
    for (int i = -s; i <= s; ++i)
        for (int j = -s; j <= s; ++j)
            if (x + j >= 0 && x + j < w && y + i >= 0 && y + i < h) {
                const float d = in[(y + i) * w + (x + j)];
                if (d == BF_MINF || d == 0.0f || fabsf(d - old) > dThresh) count++;
            }
    const unsigned sum = (2 * s + 1) * (2 * s + 1);
    out[y * w + x] = ((float)count / (float)sum >= fracReq) ? BF_MINF : old;
}

// out2 (optional): a second copy of the result (the ingest's stored frame)
__global__ __launch_bounds__(256) void k_gauss_depth(float* __restrict__ out, const float* __restrict__ in, Taps t, float sigmaR, int w, int h, float* __restrict__ out2) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int r = t.r, n = 2 * r + 1;
    float sum = 0.0f, sumW = 0.0f, res = BF_MINF;
    const float c = in[y * w + x];
    if (c != BF_MINF)
        for (int m = x - r; m <= x + r; ++m)
            for (int k = y - r; k <= y + r; ++k)
                if (m >= 0 && k >= 0 && m < w && k < h) {
                    const float d = in[k * w + m];
                    if (d != BF_MINF && fabsf(c - d) < sigmaR) {
                        const float wt = t.w[(m - x + r) * n + (k - y + r)];
                        sumW += wt;
                        sum += wt * d;
                    }
                }
    if (sumW > 0.0f) res = sum / sumW;
    out[y * w + x] = res;
    if (out2) out2[y * w + x] = res;
}

__global__ __launch_bounds__(256) void k_gauss_intensity(float* __restrict__ out, const float* __restrict__ in, Taps t, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int r = t.r, n = 2 * r + 1;
    float sum = 0.0f, sumW = 0.0f;
    for (int m = x - r; m <= x + r; ++m)
        for (int k = y - r; k <= y + r; ++k)
            if (m >= 0 && k >= 0 && m < w && k < h) {
                const float wt = t.w[(m - x + r) * n + (k - y + r)];
                sumW += wt;
                sum += wt * in[k * w + m];
            }
    if (sumW > 0.0f) out[y * w + x] = sum / sumW;
}

BF_DEV bool sampleIdx(uint32_t x, uint32_t y, uint32_t ow, uint32_t oh, uint32_t iw, uint32_t ih, uint32_t& xi, uint32_t& yi) {
    const float sw = (float)(iw - 1) / (float)(ow - 1);
    const float sh = (float)(ih - 1) / (float)(oh - 1);
    xi = (uint32_t)f2i((float)x * sw + 0.5f);
    yi = (uint32_t)f2i((float)y * sh + 0.5f);
    return xi < iw && yi < ih;
}

template <class T>
__global__ __launch_bounds__(256) void k_resample(T* __restrict__ out, uint32_t ow, uint32_t oh, const T* __restrict__ in, uint32_t iw, uint32_t ih) {
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ow || y >= oh) return;
    uint32_t xi, yi;
    if (sampleIdx(x, y, ow, oh, iw, ih, xi, yi)) out[y * ow + x] = in[yi * iw + xi];
}

__global__ __launch_bounds__(256) void k_resample_intensity(float* __restrict__ out, uint32_t ow, uint32_t oh, const uchar4* __restrict__ in, uint32_t iw, uint32_t ih) {
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ow || y >= oh) return;
    uint32_t xi, yi;
    if (sampleIdx(x, y, ow, oh, iw, ih, xi, yi)) {
        const uchar4 c = in[yi * iw + xi];
        out[y * ow + x] = ((0.299f * (float)c.x + 0.587f * (float)c.y) + 0.114f * (float)c.z) / 255.0f;      // convertToIntensity :204-207
    }
}

inline dim3 grid2(uint32_t w, uint32_t h) { return dim3(div_up(w, 64), div_up(h, 4)); }

}  // namespace

extern "C" {

int bf_image_erode_depth_map(float* d_output, const float* d_input, int structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq,
                             void* stream) {
    BF_REQUIRE(d_output && d_input && d_output != d_input && structureSize >= 0 && width && height, "bad argument");
    k_erode<<<grid2(width, height), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, d_input, structureSize, (int)width, (int)height, dThresh, fracReq, nullptr, nullptr, nullptr);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

// The erosion with a pixel-wise copy of a 4-byte-per-pixel image of the same size riding along (d_copySrc -> d_copyDst1 and, if not null, d_copyDst2): what
// CUDAImageManager::process does with three launches (CUDAImageManager.cpp:39-60, :66-86) when the frame already lives in device memory.
int bf_image_erode_depth_map_and_copy(float* d_output, const float* d_input, int structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq,
                                      const void* d_copySrc, void* d_copyDst1, void* d_copyDst2, void* stream) {
    BF_REQUIRE(d_output && d_input && d_output != d_input && structureSize >= 0 && width && height && d_copySrc && d_copyDst1, "bad argument");
    k_erode<<<grid2(width, height), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, d_input, structureSize, (int)width, (int)height, dThresh, fracReq,
                                                                           (const uint32_t*)d_copySrc, (uint32_t*)d_copyDst1, (uint32_t*)d_copyDst2);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_image_gauss_filter_depth_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, void* stream) {
    BF_REQUIRE(d_output && d_input && d_output != d_input && width && height, "bad argument");
    const int r = (int)ceil(2.0 * sigmaD);
    BF_REQUIRE(r >= 0 && r <= MAX_R, "sigmaD too large (kernel radius > 8)");
    k_gauss_depth<<<grid2(width, height), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, d_input, makeTaps(sigmaD, r), sigmaR, (int)width, (int)height, nullptr);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

// ... writing the filtered map to a second buffer as well (the ingest's stored frame, CUDAImageManager.cpp:121-136)
int bf_image_gauss_filter_depth_map2(float* d_output, float* d_output2, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, void* stream) {
    BF_REQUIRE(d_output && d_output2 && d_input && d_output != d_input && d_output2 != d_input && width && height, "bad argument");
    const int r = (int)ceil(2.0 * sigmaD);
    BF_REQUIRE(r >= 0 && r <= MAX_R, "sigmaD too large (kernel radius > 8)");
    k_gauss_depth<<<grid2(width, height), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, d_input, makeTaps(sigmaD, r), sigmaR, (int)width, (int)height, d_output2);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_image_gauss_filter_intensity(float* d_output, const float* d_input, float sigmaD, uint32_t width, uint32_t height, void* stream) {
    BF_REQUIRE(d_output && d_input && d_output != d_input && width && height, "bad argument");
    const int r = (int)ceil(2.0 * sigmaD);
    BF_REQUIRE(r >= 0 && r <= MAX_R, "sigmaD too large (kernel radius > 8)");
    k_gauss_intensity<<<grid2(width, height), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, d_input, makeTaps(sigmaD, r), (int)width, (int)height);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_image_resample_float(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const float* d_input, uint32_t inputWidth, uint32_t inputHeight,
                            void* stream) {
    BF_REQUIRE(d_output && d_input && outputWidth > 1 && outputHeight > 1 && inputWidth && inputHeight, "bad argument");
    k_resample<float><<<grid2(outputWidth, outputHeight), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, outputWidth, outputHeight, d_input, inputWidth, inputHeight);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_image_resample_uchar4(uint8_t* d_output, uint32_t outputWidth, uint32_t outputHeight, const uint8_t* d_input, uint32_t inputWidth,
                             uint32_t inputHeight, void* stream) {
    BF_REQUIRE(d_output && d_input && outputWidth > 1 && outputHeight > 1 && inputWidth && inputHeight, "bad argument");
    k_resample<uchar4><<<grid2(outputWidth, outputHeight), dim3(64, 4), 0, (hipStream_t)stream>>>((uchar4*)d_output, outputWidth, outputHeight,
                                                                                                  (const uchar4*)d_input, inputWidth, inputHeight);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_image_resample_to_intensity(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const uint8_t* d_input, uint32_t inputWidth,
                                   uint32_t inputHeight, void* stream) {
    BF_REQUIRE(d_output && d_input && outputWidth > 1 && outputHeight > 1 && inputWidth && inputHeight, "bad argument");
    k_resample_intensity<<<grid2(outputWidth, outputHeight), dim3(64, 4), 0, (hipStream_t)stream>>>(d_output, outputWidth, outputHeight, (const uchar4*)d_input,
                                                                                                    inputWidth, inputHeight);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

}  // extern "C"
