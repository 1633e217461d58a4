// oracle/ref/ref_image.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// C entry points over the REFERENCE's own image kernels, CUDAImageUtil.cu (erodeDepthMapDevice :701, gaussFilterDepthMapDevice
// :759, gaussFilterIntensityDevice :811, resampleFloat_Kernel :93, resampleFloat4_Kernel :126, resampleUCHAR4_Kernel :160,
// resampleToIntensity_Kernel :224, computeIntensityDerivatives_Kernel :260, convertDepthFloatToCameraSpaceFloat4_Kernel :367,
// computeNormals_Kernel :404, convertNormalsFloat4ToUCHAR4_Kernel :497), run through the serial block emulator.
// ref_cache_store_frame sequences them as CUDACache::storeFrame does (CUDACache.cpp:45-86, both normal formats enabled as in
// CUDACacheUtil.h:7-8); ref_ingest as CUDAImageManager::process does (CUDAImageManager.cpp:93-149: two erosions, the range-gated
// Gaussian, resampling to the integration resolution) — those two host classes need mLib and cannot be compiled themselves.
#include "CUDAImageUtil.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)

#include <vector>

extern "C" {

void ref_erode_depth(float* out, const float* in, int structureSize, unsigned w, unsigned h, float dThresh, float fracReq) {
    CUDAImageUtil::erodeDepthMap(out, (float*)in, structureSize, w, h, dThresh, fracReq);
}
void ref_gauss_filter_depth(float* out, const float* in, float sigmaD, float sigmaR, unsigned w, unsigned h) { CUDAImageUtil::gaussFilterDepthMap(out, in, sigmaD, sigmaR, w, h); }
void ref_gauss_filter_intensity(float* out, const float* in, float sigmaD, unsigned w, unsigned h) { CUDAImageUtil::gaussFilterIntensity(out, in, sigmaD, w, h); }
void ref_resample_float(float* out, unsigned ow, unsigned oh, const float* in, unsigned iw, unsigned ih) { CUDAImageUtil::resampleFloat(out, ow, oh, in, iw, ih); }
void ref_resample_uchar4(unsigned char* out, unsigned ow, unsigned oh, const unsigned char* in, unsigned iw, unsigned ih) {
    CUDAImageUtil::resampleUCHAR4((uchar4*)out, ow, oh, (const uchar4*)in, iw, ih);
}
void ref_resample_to_intensity(float* out, unsigned ow, unsigned oh, const unsigned char* in, unsigned iw, unsigned ih) {
    CUDAImageUtil::resampleToIntensity(out, ow, oh, (const uchar4*)in, iw, ih);
}

// CUDAImageManager::process, the device part (.cpp:93-149).  raw: in = sensor depth, out = the buffer after the erosions
// (d_depthInputRaw); filt = d_depthInputFiltered; integ (may be NULL) = the frame stored for integration.
void ref_ingest(float* raw, float* filt, float* integ, unsigned w, unsigned h, unsigned wi, unsigned hi, int erode, int depthFilter, float sigmaD, float sigmaR) {
    if (erode) {
        unsigned int numIter = 2;
        numIter = 2 * ((numIter + 1) / 2);
        for (unsigned int i = 0; i < numIter; i++) {
            if (i % 2 == 0) CUDAImageUtil::erodeDepthMap(filt, raw, 3, w, h, 0.05f, 0.3f);
            else CUDAImageUtil::erodeDepthMap(raw, filt, 3, w, h, 0.05f, 0.3f);
        }
    }
    if (depthFilter) CUDAImageUtil::gaussFilterDepthMap(filt, raw, sigmaD, sigmaR, w, h);
    else CUDAImageUtil::copy<float>(filt, raw, w, h);
    if (!integ) return;
    if (w == wi && h == hi) CUDAImageUtil::copy<float>(integ, filt, wi, hi);
    else CUDAImageUtil::resampleFloat(integ, wi, hi, filt, w, h);
}

// CUDACache::storeFrame (CUDACache.cpp:45-86) into caller arrays (W x H = the cache resolution)
void ref_cache_store_frame(const float* depth, unsigned dw, unsigned dh, const unsigned char* color, unsigned cw, unsigned ch, unsigned W, unsigned H,
                           const float* inputIntrinsicsInv16, float sigmaIntensity, float sigmaD, float sigmaR, float* depthDown, float* camposDown,
                           float* intensityDown, float* derivsDown, unsigned char* normalsU, float* normalsDown) {
    std::vector<float> filterHelper((size_t)dw * dh), intensityHelper((size_t)W * H);
    std::vector<float4> helperCamPos((size_t)dw * dh), helperNormals((size_t)dw * dh);
    const float* d_inputDepth = depth;
    if (sigmaD > 0.0f) {
        CUDAImageUtil::gaussFilterDepthMap(filterHelper.data(), depth, sigmaD, sigmaR, dw, dh);
        d_inputDepth = filterHelper.data();
    }
    CUDAImageUtil::convertDepthFloatToCameraSpaceFloat4(helperCamPos.data(), d_inputDepth, float4x4(inputIntrinsicsInv16), dw, dh);
    CUDAImageUtil::resampleFloat4((float4*)camposDown, W, H, helperCamPos.data(), dw, dh);
    CUDAImageUtil::computeNormals(helperNormals.data(), helperCamPos.data(), dw, dh);
    CUDAImageUtil::resampleFloat4((float4*)normalsDown, W, H, helperNormals.data(), dw, dh);
    CUDAImageUtil::convertNormalsFloat4ToUCHAR4((uchar4*)normalsU, (const float4*)normalsDown, W, H);
    CUDAImageUtil::resampleFloat(depthDown, W, H, d_inputDepth, dw, dh);
    CUDAImageUtil::resampleToIntensity(intensityHelper.data(), W, H, (const uchar4*)color, cw, ch);
    if (sigmaIntensity > 0.0f) CUDAImageUtil::gaussFilterIntensity(intensityDown, intensityHelper.data(), sigmaIntensity, W, H);
    else memcpy(intensityDown, intensityHelper.data(), sizeof(float) * W * H);
    CUDAImageUtil::computeIntensityDerivatives((float2*)derivsDown, intensityDown, W, H);
}

}  // extern "C"
