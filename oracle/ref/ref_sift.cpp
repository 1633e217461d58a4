// oracle/ref/ref_sift.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's SiftGPU fork, whole: SiftGPU.cpp, SiftPyramid.cpp, GlobalUtil.cpp, CuTexImage.cpp, SiftMatch.cpp and ProgramCU.cu
// (every pyramid / DoG / key-point / orientation / descriptor / matching kernel) compiled from where they lie; the Makefile adds those
// files as separate objects.  This file is the part of Bundler / OnlineBundler that drives them: SiftCameraParams as
// OnlineBundler.cpp:45-55 fills them, SiftGPU::SetParams + InitSiftGPU as Bundler::initSift (Bundler.cpp:57-62), RunSIFT +
// GetKeyPointsAndDescriptorsCUDA as Bundler::detectFeatures (:91-101), SiftMatchGPU::SetDescriptors + GetSiftMatch as
// Bundler::matchAndFilter (:127-135).  Elementary functions (exp, atan2, sincos, acos ...) are glibc's here, CUDA's fast-math intrinsics
// in the reference build and the fixed sequences of include/bf_detmath.h in the oracle / product: stages that call them agree to a
// stated bound, the others bit for bit.
#define private public
#define protected public
#include "GlobalUtil.h"
#include "SiftGPU.h"
#include "SiftPyramid.h"
#include "CuTexImage.h"
#undef private
#undef protected
#include "SiftMatch.h"
#include "SiftCameraParams.h"

SiftCameraParams c_siftCameraParams;                                                   // CUDASiftConstant.cu:4 (__constant__)
extern "C" void updateConstantSiftCameraParams(const SiftCameraParams& p) { c_siftCameraParams = p; }

struct ref_sift {
    SiftGPU* sift;
    SiftMatchGPU* matcher;
    unsigned int w, h, maxKeys;
    SIFTKeyPoint* keys; SIFTKeyPointDesc* descs;
    int* numMatches; float* dist; uint2* idx;
};

extern "C" {

ref_sift* ref_sift_create(unsigned int siftW, unsigned int siftH, unsigned int depthW, unsigned int depthH, const float* K16, const float* Kinv16,
                          unsigned int featureCountThreshold, float depthMin, float depthMax, float minKeyScale, unsigned int maxKeysPerImage) {
    SiftCameraParams p; memset(&p, 0, sizeof p);
    p.m_depthWidth = depthW; p.m_depthHeight = depthH; p.m_intensityWidth = siftW; p.m_intensityHeight = siftH;
    p.m_siftIntrinsics = float4x4(K16); p.m_siftIntrinsicsInv = float4x4(Kinv16);
    p.m_downSampIntrinsics = float4x4(K16); p.m_downSampIntrinsicsInv = float4x4(Kinv16);      // not read by the SIFT kernels
    p.m_minKeyScale = minKeyScale;
    updateConstantSiftCameraParams(p);
    ref_sift* h = new ref_sift;
    h->w = siftW; h->h = siftH; h->maxKeys = maxKeysPerImage;
    h->sift = new SiftGPU;
    h->sift->SetParams(siftW, siftH, false, featureCountThreshold, depthMin, depthMax);
    h->sift->InitSiftGPU();
    h->matcher = new SiftMatchGPU(maxKeysPerImage);
    h->matcher->InitSiftMatch();
    h->keys = (SIFTKeyPoint*)calloc(maxKeysPerImage, sizeof(SIFTKeyPoint));
    h->descs = (SIFTKeyPointDesc*)calloc(maxKeysPerImage, sizeof(SIFTKeyPointDesc));
    h->numMatches = (int*)calloc(1, sizeof(int));
    h->dist = (float*)calloc(MAX_MATCHES_PER_IMAGE_PAIR_RAW, sizeof(float));
    h->idx = (uint2*)calloc(MAX_MATCHES_PER_IMAGE_PAIR_RAW, sizeof(uint2));
    return h;
}

// Bundler::detectFeatures: returns the number of key points; keys4 (x, y, scale, depth) and descs128 receive up to maxKeys entries
int ref_sift_run(ref_sift* h, const float* intensity, const float* depth, float* keys4, unsigned char* descs128) {
    if (!h->sift->RunSIFT(const_cast<float*>(intensity), depth)) return -1;
    SIFTImageGPU img; img.d_keyPoints = h->keys; img.d_keyPointDescs = h->descs;
    const unsigned int n = h->sift->GetKeyPointsAndDescriptorsCUDA(img, depth, h->maxKeys);
    const unsigned int m = n < h->maxKeys ? n : h->maxKeys;
    memcpy(keys4, h->keys, sizeof(SIFTKeyPoint) * m);
    memcpy(descs128, h->descs, sizeof(SIFTKeyPointDesc) * m);
    return (int)n;
}

// Bundler::matchAndFilter :127-135 for one image pair: returns the raw match count; idx2 / dist receive up to 128 entries
int ref_sift_match(ref_sift* h, const unsigned char* descs1, int n1, const unsigned char* descs2, int n2, unsigned int off1, unsigned int off2,
                   float distMax, float ratioMax, unsigned int* idx2, float* dist) {
    ImagePairMatch m; m.d_numMatches = h->numMatches; m.d_distances = h->dist; m.d_keyPointIndices = h->idx;
    *h->numMatches = 0;
    h->matcher->SetDescriptors(0, n1, const_cast<unsigned char*>(descs1));
    h->matcher->SetDescriptors(1, n2, const_cast<unsigned char*>(descs2));
    h->matcher->GetSiftMatch(n1, m, make_uint2(off1, off2), distMax, ratioMax);
    const int n = *h->numMatches;
    const int c = n < (int)MAX_MATCHES_PER_IMAGE_PAIR_RAW ? n : (int)MAX_MATCHES_PER_IMAGE_PAIR_RAW;
    memcpy(idx2, h->idx, sizeof(uint2) * (size_t)(c > 0 ? c : 0));
    memcpy(dist, h->dist, sizeof(float) * (size_t)(c > 0 ? c : 0));
    return n;
}

// the first two stages of SiftPyramid::RunSIFT only (BuildPyramid + DetectKeypoints, SiftPyramid.cpp:148-166): everything up to the raw
// per-level key lists is +, -, *, / arithmetic, so these stages can be compared bit for bit
int ref_sift_detect(ref_sift* h, const float* intensity, const float* depth) {
    SiftPyramid* p = h->sift->_pyramid;
    p->BuildPyramid(const_cast<float*>(intensity));
    p->DetectKeypoints(depth);
    return p->_featureNum;
}
// the next stages of RunSIFT (SiftPyramid.cpp:170-196): LimitFeatureCount(0), GetFeatureOrientations, ReshapeFeatureList, LimitFeatureCount(1)
// SiftPyramid::LimitFeatureCount is declared inline and defined in SiftPyramid.cpp only (no symbol to link against): same statements (:208-236)
static void limitFeatureCount(SiftPyramid* p) {
    if (GlobalUtil::_FeatureCountThreshold <= 0) return;
    if (GlobalUtil::_TruncateMethod == 2) {
        int i = 0, new_feature_num = 0, level_num = p->param._dog_level_num * p->_octave_num;
        for (; new_feature_num < GlobalUtil::_FeatureCountThreshold && i < level_num; ++i) new_feature_num += p->_levelFeatureNum[i];
        for (; i < level_num; ++i) p->_levelFeatureNum[i] = 0;
        if (new_feature_num < p->_featureNum) p->_featureNum = new_feature_num;
    } else {
        int i = 0;
        while (p->_featureNum - p->_levelFeatureNum[i] > GlobalUtil::_FeatureCountThreshold) {
            p->_featureNum -= p->_levelFeatureNum[i];
            p->_levelFeatureNum[i++] = 0;
        }
    }
}
int ref_sift_orient(ref_sift* h) {
    SiftPyramid* p = h->sift->_pyramid;
    limitFeatureCount(p);
    p->GetFeatureOrientations();
    p->ReshapeFeatureList();
    limitFeatureCount(p);
    return p->_featureNum;
}
// final key list of one slot after ref_sift_orient: float4 (x, y, scale, orientation) per feature
int ref_sift_final_keys(ref_sift* h, int slot, float* out4, int capacityKeys) {
    SiftPyramid* p = h->sift->_pyramid;
    const int n = p->_levelFeatureNum[slot];
    const int c = n < capacityKeys ? n : capacityKeys;
    if (c > 0) memcpy(out4, p->_featureTexFinal[slot]._cuData, sizeof(float) * 4 * (size_t)c);
    return n;
}
// raw key list of one (octave, dog level) slot: int4 per key as ComputeKEY_Kernel wrote it
int ref_sift_raw_keys(ref_sift* h, int slot, int* out4, int capacityKeys) {
    SiftPyramid* p = h->sift->_pyramid;
    const int n = p->_levelFeatureNum[slot];
    const int c = n < capacityKeys ? n : capacityKeys;
    if (c > 0) memcpy(out4, p->_featureTexRaw[slot]._cuData, sizeof(int) * 4 * (size_t)c);
    return n;
}

// inspection of the pyramid after ref_sift_run (the classes' private members are opened above for this): image of one level
// (which: 0 Gaussian, 1 DoG, 2 key, 3 gradient, 4 rotation; level index counted from the first stored level of the octave)
int ref_sift_level(ref_sift* h, int octave, int level, int which, float* out, int capacityFloats, int* width, int* height, int* channels) {
    SiftPyramid* p = h->sift->_pyramid;
    CuTexImage* t = p->GetBaseLevel(octave, which) + level;
    *width = t->GetImgWidth(); *height = t->GetImgHeight(); *channels = t->GetImgNumChannels();
    const int n = *width * *height * *channels;
    if (n > capacityFloats || !t->_cuData) return -1;
    memcpy(out, t->_cuData, sizeof(float) * (size_t)n);
    return n;
}
int ref_sift_level_counts(ref_sift* h, int* counts, int capacity) {
    SiftPyramid* p = h->sift->_pyramid;
    const int n = p->_octave_num * p->param._dog_level_num;
    for (int i = 0; i < n && i < capacity; ++i) counts[i] = p->_levelFeatureNum[i];
    return n;
}

}  // extern "C"
