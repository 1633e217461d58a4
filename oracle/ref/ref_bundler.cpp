// oracle/ref/ref_bundler.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's Bundler (Bundler.cpp compiled as it is: constructor :18-53, detectFeatures :91-101, matchAndFilter :103-249,
// optimize :251-280, storeCachedFrame, copyFrame, isValid, tryRevalidation :309-357, reset, addInvalidFrame, invalidateLastFrame,
// fuseToGlobal) on top of the reference's own SiftGPU fork, SIFTImageManager, CUDACache (CUDACache.cpp, also compiled as it is), SBA,
// CUDASolverBundling and CUDAImageManager.  Stand-ins: mLib's element arithmetic (shim/mlib_standin.h) and an RGBDSensor that carries sizes,
// intrinsics and the frame the test supplies (shim/app/RGBDSensor.h).
#define private public
#define protected public
#include "Bundler.h"
#include "CUDACache.h"
#include "CUDAImageManager.h"
#include "GlobalAppState.h"
#include "SiftGPU/SiftCameraParams.h"
#undef private
#undef protected

extern "C" void updateConstantSiftCameraParams(const SiftCameraParams& params);

struct ref_bundling_state {            // GlobalBundlingState / GlobalAppState values the bundling classes read
    unsigned int maxNumImages, submapSize, widthSIFT, heightSIFT, maxNumKeysPerImage, numLocalNonLinIterations, numLocalLinIterations, numGlobalNonLinIterations,
        numGlobalLinIterations, downsampledWidth, downsampledHeight, minNumMatchesLocal, minNumMatchesGlobal, denseOverlapCheckSubsampleFactor, numOptPerResidualRemoval;
    float verifySiftErrThresh, verifySiftCorrThresh, projCorrDistThres, projCorrNormalThres, projCorrColorThresh, surfAreaPcaThresh, verifyOptErrThresh,
        verifyOptCorrThresh, maxKabschResidual2, minKeyScale, siftMatchThresh, siftMatchRatioMaxLocal, siftMatchRatioMaxGlobal, colorDownSigma, depthDownSigmaD,
        depthDownSigmaR, optMaxResThresh, denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax, sensorDepthMin,
        sensorDepthMax;
    int useComprehensiveFrameInvalidation, useLocalVerify, useLocalDense, erodeSIFTdepth, depthFilter;
    float depthSigmaD, depthSigmaR;
};

extern "C" {

void ref_set_bundling_state(const ref_bundling_state* p) {
    GlobalBundlingState& g = GlobalBundlingState::get();
    g.s_maxNumImages = p->maxNumImages; g.s_submapSize = p->submapSize; g.s_widthSIFT = p->widthSIFT; g.s_heightSIFT = p->heightSIFT;
    g.s_maxNumKeysPerImage = p->maxNumKeysPerImage; g.s_numLocalNonLinIterations = p->numLocalNonLinIterations; g.s_numLocalLinIterations = p->numLocalLinIterations;
    g.s_numGlobalNonLinIterations = p->numGlobalNonLinIterations; g.s_numGlobalLinIterations = p->numGlobalLinIterations;
    g.s_downsampledWidth = p->downsampledWidth; g.s_downsampledHeight = p->downsampledHeight; g.s_minNumMatchesLocal = p->minNumMatchesLocal;
    g.s_minNumMatchesGlobal = p->minNumMatchesGlobal; g.s_denseOverlapCheckSubsampleFactor = p->denseOverlapCheckSubsampleFactor;
    g.s_numOptPerResidualRemoval = p->numOptPerResidualRemoval;
    g.s_verifySiftErrThresh = p->verifySiftErrThresh; g.s_verifySiftCorrThresh = p->verifySiftCorrThresh; g.s_projCorrDistThres = p->projCorrDistThres;
    g.s_projCorrNormalThres = p->projCorrNormalThres; g.s_projCorrColorThresh = p->projCorrColorThresh; g.s_surfAreaPcaThresh = p->surfAreaPcaThresh;
    g.s_verifyOptErrThresh = p->verifyOptErrThresh; g.s_verifyOptCorrThresh = p->verifyOptCorrThresh; g.s_maxKabschResidual2 = p->maxKabschResidual2;
    g.s_minKeyScale = p->minKeyScale; g.s_siftMatchThresh = p->siftMatchThresh; g.s_siftMatchRatioMaxLocal = p->siftMatchRatioMaxLocal;
    g.s_siftMatchRatioMaxGlobal = p->siftMatchRatioMaxGlobal; g.s_colorDownSigma = p->colorDownSigma; g.s_depthDownSigmaD = p->depthDownSigmaD;
    g.s_depthDownSigmaR = p->depthDownSigmaR; g.s_optMaxResThresh = p->optMaxResThresh; g.s_denseDistThresh = p->denseDistThresh;
    g.s_denseNormalThresh = p->denseNormalThresh; g.s_denseColorThresh = p->denseColorThresh; g.s_denseColorGradientMin = p->denseColorGradientMin;
    g.s_denseDepthMin = p->denseDepthMin; g.s_denseDepthMax = p->denseDepthMax;
    g.s_useComprehensiveFrameInvalidation = p->useComprehensiveFrameInvalidation != 0; g.s_useLocalVerify = p->useLocalVerify != 0;
    g.s_useLocalDense = p->useLocalDense != 0; g.s_erodeSIFTdepth = p->erodeSIFTdepth != 0;
    g.s_depthFilter = p->depthFilter != 0; g.s_depthSigmaD = p->depthSigmaD; g.s_depthSigmaR = p->depthSigmaR;
    g.s_enableGlobalTimings = false; g.s_enablePerFrameTimings = false; g.s_verbose = false; g.s_recordSolverConvergence = false;
    GlobalAppState::get().s_sensorDepthMin = p->sensorDepthMin; GlobalAppState::get().s_sensorDepthMax = p->sensorDepthMax;
}

// the SIFT-side camera constants OnlineBundler's constructor uploads (OnlineBundler.cpp:46-55)
void ref_set_sift_camera(unsigned int depthW, unsigned int depthH, unsigned int siftW, unsigned int siftH, const float* siftIntrinsics16, float minKeyScale) {
    SiftCameraParams c;
    c.m_depthWidth = depthW; c.m_depthHeight = depthH; c.m_intensityWidth = siftW; c.m_intensityHeight = siftH;
    c.m_siftIntrinsics = float4x4(siftIntrinsics16); c.m_siftIntrinsicsInv = c.m_siftIntrinsics.getInverse();
    c.m_minKeyScale = minKeyScale;
    updateConstantSiftCameraParams(c);
}

struct ref_bundler { Bundler* b; CUDAImageManager* im; RGBDSensor* sensor; };

ref_bundler* ref_bundler_create(unsigned int maxNumImages, unsigned int maxNumKeysPerImage, const float* siftIntrinsicsInv16, unsigned int depthW, unsigned int depthH,
                                const float* depthIntrinsics16, int isLocal) {
    ref_bundler* h = new ref_bundler;
    h->sensor = new RGBDSensor(depthW, depthH, depthW, depthH, mat4f(depthIntrinsics16), mat4f(depthIntrinsics16));
    h->im = new CUDAImageManager(depthW, depthH, GlobalBundlingState::get().s_widthSIFT, GlobalBundlingState::get().s_heightSIFT, h->sensor, false);
    h->b = new Bundler(maxNumImages, maxNumKeysPerImage, mat4f(siftIntrinsicsInv16), h->im, isLocal != 0);
    return h;
}
void ref_bundler_destroy(ref_bundler* h) { delete h->b; delete h->im; delete h->sensor; delete h; }
void ref_bundler_detect_features(ref_bundler* h, float* intensitySift, const float* depthFilt) { h->b->detectFeatures(intensitySift, depthFilt); }
void ref_bundler_store_cached_frame(ref_bundler* h, unsigned int depthW, unsigned int depthH, const unsigned char* colorRGBX, unsigned int colorW, unsigned int colorH,
                                    const float* depthRaw) {
    h->b->storeCachedFrame(depthW, depthH, (const uchar4*)colorRGBX, colorW, colorH, depthRaw);
}
unsigned int ref_bundler_match_and_filter(ref_bundler* h) { return h->b->matchAndFilter(); }
int ref_bundler_optimize(ref_bundler* h, unsigned int numNonLin, unsigned int numLin, int useVerify, int removeMaxResidual, int isScanDone, int* optRemoved) {
    bool removed = false;
    const bool ok = h->b->optimize(numNonLin, numLin, useVerify != 0, removeMaxResidual != 0, isScanDone != 0, removed);
    *optRemoved = removed ? 1 : 0;
    return ok ? 1 : 0;
}
void ref_bundler_copy_frame(ref_bundler* h, ref_bundler* from, unsigned int frame) { h->b->copyFrame(from->b, frame); }
void ref_bundler_add_invalid_frame(ref_bundler* h) { h->b->addInvalidFrame(); }
void ref_bundler_invalidate_last_frame(ref_bundler* h) { h->b->invalidateLastFrame(); }
void ref_bundler_fuse_to_global(ref_bundler* h, ref_bundler* glob) { h->b->fuseToGlobal(glob->b); }
unsigned int ref_bundler_try_revalidation(ref_bundler* h, unsigned int curGlobalFrame, int isScanDone) { return h->b->tryRevalidation(curGlobalFrame, isScanDone != 0); }
void ref_bundler_reset(ref_bundler* h) { h->b->reset(); }
int ref_bundler_is_valid(ref_bundler* h) { return h->b->isValid() ? 1 : 0; }
unsigned int ref_bundler_num_frames(ref_bundler* h) { return h->b->getNumFrames(); }
unsigned int ref_bundler_curr_frame(ref_bundler* h) { return h->b->getCurrFrameNumber(); }
unsigned int ref_bundler_revalidated_idx(ref_bundler* h) { return h->b->getRevalidatedIdx(); }
void ref_bundler_get_trajectory(ref_bundler* h, float* out16, unsigned int n) { memcpy(out16, h->b->d_trajectory, 64 * (size_t)n); }
void ref_bundler_set_trajectory(ref_bundler* h, const float* in16, unsigned int first, unsigned int n) { memcpy(h->b->d_trajectory + first, in16, 64 * (size_t)n); }
void ref_bundler_get_valid(ref_bundler* h, int* out, unsigned int n) { for (unsigned int i = 0; i < n; ++i) out[i] = h->b->m_siftManager->m_validImages[i]; }
unsigned int ref_bundler_num_keys(ref_bundler* h, unsigned int image) { return h->b->m_siftManager->getNumKeyPointsPerImage(image); }
void ref_bundler_get_keys(ref_bundler* h, unsigned int image, float* keys4, unsigned char* descs128) {
    const SIFTImageGPU& img = h->b->m_siftManager->getImageGPU(image);
    const unsigned int n = h->b->m_siftManager->getNumKeyPointsPerImage(image);
    memcpy(keys4, img.d_keyPoints, sizeof(SIFTKeyPoint) * (size_t)n); memcpy(descs128, img.d_keyPointDescs, sizeof(SIFTKeyPointDesc) * (size_t)n);
}
unsigned int ref_bundler_num_correspondences(ref_bundler* h) { return h->b->m_siftManager->getNumGlobalCorrespondences(); }
void ref_bundler_get_correspondences(ref_bundler* h, void* entryJ, unsigned int n) { memcpy(entryJ, h->b->m_siftManager->d_globMatches, sizeof(EntryJ) * (size_t)n); }
void ref_bundler_get_num_filtered(ref_bundler* h, int* out, unsigned int n) { memcpy(out, h->b->m_siftManager->d_currNumFilteredMatchesPerImagePair, sizeof(int) * (size_t)n); }
void ref_bundler_get_cache_frame(ref_bundler* h, unsigned int i, float* depth, float* campos4, float* intensity, float* derivs2, float* normals4, unsigned int* wh) {
    CUDACache* c = h->b->m_cudaCache;
    const size_t n = (size_t)c->getWidth() * c->getHeight();
    const CUDACachedFrame& f = c->m_cache[i];
    wh[0] = c->getWidth(); wh[1] = c->getHeight();
    memcpy(depth, f.d_depthDownsampled, 4 * n); memcpy(campos4, f.d_cameraposDownsampled, 16 * n); memcpy(intensity, f.d_intensityDownsampled, 4 * n);
    memcpy(derivs2, f.d_intensityDerivsDownsampled, 8 * n); memcpy(normals4, f.d_normalsDownsampled, 16 * n);
}
void* ref_bundler_sift_manager(ref_bundler* h) { return h->b->m_siftManager; }
void* ref_bundler_cuda_cache(ref_bundler* h) { return h->b->m_cudaCache; }
// all key points of the manager as the evaluator sees them (getSIFTKeyPointsDEBUG: the packed array)
unsigned int ref_bundler_get_all_keys(ref_bundler* h, float* keys4, unsigned int capacity) {
    unsigned int n = 0;
    for (unsigned int k : h->b->m_siftManager->m_numKeyPointsPerImage) n += k;
    memcpy(keys4, h->b->m_siftManager->d_keyPoints, sizeof(SIFTKeyPoint) * (size_t)std::min(n, capacity));
    return n;
}
// a ref_siftmgr view (ref_siftmgr.cpp) of the bundler's manager, for the raw / filtered match accessors; free() it
struct ref_siftmgr { SIFTImageManager* m; unsigned int maxImages, maxKeys; CUDACachedFrame* frames; };
ref_siftmgr* ref_bundler_siftmgr_view(ref_bundler* h) { ref_siftmgr* v = (ref_siftmgr*)calloc(1, sizeof(ref_siftmgr)); v->m = h->b->m_siftManager; return v; }
void ref_bundler_cache_intrinsics(ref_bundler* h, float* K16) { memcpy(K16, h->b->m_cudaCache->getIntrinsics().matrix, 64); }

}
