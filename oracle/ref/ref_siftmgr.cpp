// oracle/ref/ref_siftmgr.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's match-filter chain: SiftGPU/SIFTImageManager.cu compiled from where it lies — SortKeyPointMatchesCU,
// FilterKeyPointMatchesCU (cuda_kabsch.h), FilterMatchesBySurfaceAreaCU (cuda_surfaceArea.h, cuda_EigenValue.h),
// FilterMatchesByDenseVerifyCU, AddCurrToResidualsCU, VerifyTrajectoryCU — kernels AND their launch configurations (the member
// functions of class SIFTImageManager are called as they are) - and the host half of the class, SIFTImageManager.cpp (constructor / alloc,
// createSIFTImageGPU / finalizeSIFTImageGPU, computeTracks + fuseToGlobal, filterFrames), compiled by the Makefile from a temporary copy
// whose two includes of GlobalBundlingState.h / GlobalAppState.h (unused by the file; they pull the application's precompiled header) are
// deleted.  Private members are opened with a #define where this glue has to reach state the class keeps to itself - test glue, not product code.
#define private public
#define protected public
#include "SIFTImageManager.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)
#undef private
#undef protected

#include <cstdlib>
#include <cstring>
#include <new>

struct ref_siftmgr {
    SIFTImageManager* m;
    unsigned int maxImages, maxKeys;
    CUDACachedFrame* frames;
};

extern "C" {

ref_siftmgr* ref_siftmgr_create(unsigned int maxImages, unsigned int maxKeysPerImage) {
    ref_siftmgr* h = new ref_siftmgr;
    h->maxImages = maxImages; h->maxKeys = maxKeysPerImage;
    h->m = new SIFTImageManager(maxImages, maxKeysPerImage);             // SIFTImageManager.cpp:7-15, alloc :274-313
    for (unsigned int i = 0; i < maxImages; ++i) h->m->m_validImages[i] = 1;
    h->frames = (CUDACachedFrame*)calloc(maxImages, sizeof(CUDACachedFrame));
    return h;
}

// an image through createSIFTImageGPU / finalizeSIFTImageGPU (keys are PACKED in the reference: image i starts at the prefix sum)
unsigned int ref_siftmgr_add_image(ref_siftmgr* h, const float* keys4, const unsigned char* descs128, unsigned int n) {
    SIFTImageGPU& img = h->m->createSIFTImageGPU();
    memcpy(img.d_keyPoints, keys4, sizeof(SIFTKeyPoint) * (size_t)n);
    memcpy(img.d_keyPointDescs, descs128, sizeof(SIFTKeyPointDesc) * (size_t)n);
    h->m->finalizeSIFTImageGPU(n);
    return h->m->m_numKeyPointsPerImagePrefixSum.back();
}
void ref_siftmgr_set_residuals(ref_siftmgr* h, const void* entryJ, const unsigned int* keyIdx2, unsigned int n) {
    memcpy(h->m->d_globMatches, entryJ, sizeof(EntryJ) * (size_t)n);
    memcpy(h->m->d_globMatchesKeyPointIndices, keyIdx2, sizeof(uint2) * (size_t)n);
    h->m->m_globNumResiduals = n; *h->m->d_globNumResiduals = (int)n;
}
// fuseToGlobal (SIFTImageManager.cpp:411-468) into a fresh global manager; returns the number of fused keys
unsigned int ref_siftmgr_fuse_to_global(ref_siftmgr* h, const float* K16, const float* Kinv16, const float* transforms16, unsigned int maxKeysGlobal, float* outKeys4,
                                        unsigned char* outDescs128) {
    SIFTImageManager global(4, maxKeysGlobal);
    h->m->fuseToGlobal(&global, float4x4(K16), (const float4x4*)transforms16, float4x4(Kinv16));
    const unsigned int n = global.m_numKeyPointsPerImage[0];
    memcpy(outKeys4, global.d_keyPoints, sizeof(SIFTKeyPoint) * (size_t)n);
    memcpy(outDescs128, global.d_keyPointDescs, sizeof(SIFTKeyPointDesc) * (size_t)n);
    return n;
}
// filterFrames (SIFTImageManager.cpp:551-575): returns the last matched frame, *valid receives m_validImages[curFrame]
unsigned int ref_siftmgr_filter_frames(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num, const int* validIn, int* validOut) {
    for (unsigned int i = 0; i < num; ++i) h->m->m_validImages[i] = validIn[i];
    const unsigned int last = h->m->filterFrames(cur, start, num);
    *validOut = h->m->m_validImages[cur];
    return last;
}

void ref_siftmgr_set_keys(ref_siftmgr* h, const float* keys, unsigned int count) { memcpy(h->m->d_keyPoints, keys, sizeof(SIFTKeyPoint) * (size_t)count); }
void ref_siftmgr_set_raw(ref_siftmgr* h, unsigned int pair, int numMatches, const float* dist, const unsigned int* idx2) {
    h->m->d_currNumMatchesPerImagePair[pair] = numMatches;
    memcpy(h->m->d_currMatchDistances + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_RAW, dist, sizeof(float) * MAX_MATCHES_PER_IMAGE_PAIR_RAW);
    memcpy(h->m->d_currMatchKeyPointIndices + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_RAW, idx2, sizeof(uint2) * MAX_MATCHES_PER_IMAGE_PAIR_RAW);
}
void ref_siftmgr_get_raw(ref_siftmgr* h, unsigned int pair, int* numMatches, float* dist, unsigned int* idx2) {
    *numMatches = h->m->d_currNumMatchesPerImagePair[pair];
    memcpy(dist, h->m->d_currMatchDistances + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_RAW, sizeof(float) * MAX_MATCHES_PER_IMAGE_PAIR_RAW);
    memcpy(idx2, h->m->d_currMatchKeyPointIndices + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_RAW, sizeof(uint2) * MAX_MATCHES_PER_IMAGE_PAIR_RAW);
}
void ref_siftmgr_set_filtered(ref_siftmgr* h, unsigned int pair, int numMatches, const float* dist, const unsigned int* idx2, const float* T, const float* Tinv) {
    h->m->d_currNumFilteredMatchesPerImagePair[pair] = numMatches;
    if (dist) memcpy(h->m->d_currFilteredMatchDistances + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED, dist, sizeof(float) * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED);
    if (idx2) memcpy(h->m->d_currFilteredMatchKeyPointIndices + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED, idx2, sizeof(uint2) * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED);
    if (T) memcpy(h->m->d_currFilteredTransforms + pair, T, 64);
    if (Tinv) memcpy(h->m->d_currFilteredTransformsInv + pair, Tinv, 64);
}
void ref_siftmgr_get_filtered(ref_siftmgr* h, unsigned int pair, int* numMatches, float* dist, unsigned int* idx2, float* T, float* Tinv) {
    *numMatches = h->m->d_currNumFilteredMatchesPerImagePair[pair];
    memcpy(dist, h->m->d_currFilteredMatchDistances + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED, sizeof(float) * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED);
    memcpy(idx2, h->m->d_currFilteredMatchKeyPointIndices + (size_t)pair * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED, sizeof(uint2) * MAX_MATCHES_PER_IMAGE_PAIR_FILTERED);
    memcpy(T, h->m->d_currFilteredTransforms + pair, 64); memcpy(Tinv, h->m->d_currFilteredTransformsInv + pair, 64);
}
// a cached frame = the six host arrays of CUDACachedFrame (CUDACacheUtil.h:10-53); they must outlive the calls
void ref_siftmgr_set_cached_frame(ref_siftmgr* h, unsigned int i, float* depth, float* campos4, float* intensity, float* derivs2, unsigned char* normalsU4, float* normals4) {
    CUDACachedFrame& f = h->frames[i];
    f.d_depthDownsampled = depth; f.d_cameraposDownsampled = (float4*)campos4; f.d_intensityDownsampled = intensity;
    f.d_intensityDerivsDownsampled = (float2*)derivs2;
#ifdef CUDACACHE_UCHAR_NORMALS
    f.d_normalsDownsampledUCHAR4 = (uchar4*)normalsU4;
#endif
#ifdef CUDACACHE_FLOAT_NORMALS
    f.d_normalsDownsampled = (float4*)normals4;
#endif
}

void ref_siftmgr_sort(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num) { h->m->SortKeyPointMatchesCU(cur, start, num); }
void ref_siftmgr_filter_keypoint_matches(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num, const float* Kinv16, unsigned int minMatches, float maxRes2) {
    h->m->FilterKeyPointMatchesCU(cur, start, num, float4x4(Kinv16), minMatches, maxRes2);
}
void ref_siftmgr_filter_surface_area(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num, const float* Kinv16, float areaThresh) {
    h->m->FilterMatchesBySurfaceAreaCU(cur, start, num, float4x4(Kinv16), areaThresh);
}
void ref_siftmgr_filter_dense_verify(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num, unsigned int W, unsigned int H, const float* K16,
                                     float distThresh, float normalThresh, float colorThresh, float errThresh, float corrThresh, float dmin, float dmax) {
    h->m->FilterMatchesByDenseVerifyCU(cur, start, num, W, H, float4x4(K16), h->frames, distThresh, normalThresh, colorThresh, errThresh, corrThresh, dmin, dmax);
}
unsigned int ref_siftmgr_add_curr_to_residuals(ref_siftmgr* h, unsigned int cur, unsigned int start, unsigned int num, const float* Kinv16) {
    h->m->AddCurrToResidualsCU(cur, start, num, float4x4(Kinv16));
    return h->m->m_globNumResiduals;
}
void ref_siftmgr_get_residuals(ref_siftmgr* h, void* entryJ, unsigned int* keyIdx2, unsigned int n) {
    memcpy(entryJ, h->m->d_globMatches, sizeof(EntryJ) * (size_t)n);
    memcpy(keyIdx2, h->m->d_globMatchesKeyPointIndices, sizeof(uint2) * (size_t)n);
}
int ref_siftmgr_verify_trajectory(ref_siftmgr* h, unsigned int numImages, const float* traj16, const int* validImages, unsigned int W, unsigned int H, const float* K16,
                                  float distThresh, float normalThresh, float colorThresh, float errThresh, float corrThresh, float dmin, float dmax) {
    for (unsigned int i = 0; i < numImages; ++i) h->m->m_validImages[i] = validImages[i];
    std::vector<float4x4> T(numImages);
    memcpy(T.data(), traj16, 64 * (size_t)numImages);
    return h->m->VerifyTrajectoryCU(numImages, T.data(), W, H, float4x4(K16), h->frames, distThresh, normalThresh, colorThresh, errThresh, corrThresh, dmin, dmax);
}
// CheckForInvalidFramesCU / CheckForInvalidFramesSimpleCU (:746-797) and InvalidateImageToImageCU (:706-720); valid flags in and out
void ref_siftmgr_check_invalid_frames(ref_siftmgr* h, const int* rows, unsigned int numVars, int* valid, int simple) {
    for (unsigned int i = 0; i < numVars; ++i) h->m->m_validImages[i] = valid[i];
    if (simple) h->m->CheckForInvalidFramesSimpleCU(rows, numVars);
    else h->m->CheckForInvalidFramesCU(rows, numVars);
    for (unsigned int i = 0; i < numVars; ++i) valid[i] = h->m->m_validImages[i];
}
void ref_siftmgr_invalidate_image_to_image(ref_siftmgr* h, unsigned int i, unsigned int j) { h->m->InvalidateImageToImageCU(make_uint2(i, j)); }
unsigned int ref_siftmgr_sizeof_entryj() { return (unsigned int)sizeof(EntryJ); }

}  // extern "C"
