// oracle/ref/ref_sba.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's host side of the bundle adjustment, compiled from where it lies: SBA.cpp (constructor with the weight schedules :19-51,
// align :54-118, alignCUDA :120-139, removeMaxResidualCUDA :167-203) and Solver/CUDASolverBundling.cpp (constructor :24-136, solve
// :186-283, buildVariablesToCorrespondencesTable, computeMaxResidual :312-419, getMaxResidual :421-446, useVerification :448-476) on top
// of the kernels of Solver/SolverBundling.cu and SBA.cu and of the real SIFTImageManager (InvalidateImageToImageCU,
// CheckForInvalidFrames[Simple]CU) and the real CUDACache.  Stand-ins (shim/): mLib's element arithmetic, Timer and ParameterFile; empty
// SiftVisualization.h / cuda_d3d11_interop.h.  The Makefile compiles verbatim temporary copies of the application files so that their
// quoted includes resolve to those stand-ins instead of the application's precompiled header (see its header comment).
#define private public
#define protected public
#include "SBA.h"
#include "CUDACache.h"
#undef private
#undef protected

struct ref_siftmgr { SIFTImageManager* m; unsigned int maxImages, maxKeys; CUDACachedFrame* frames; };      // as in ref_siftmgr.cpp
struct ref_bundling_params {           // the GlobalBundlingState values SBA / CUDASolverBundling read
    unsigned int numLocalNonLinIterations, numGlobalNonLinIterations, submapSize, denseOverlapCheckSubsampleFactor;
    float optMaxResThresh, denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax;
    int useComprehensiveFrameInvalidation, useLocalDense, recordSolverConvergence;
};

extern "C" {

void* ref_sba_create(unsigned int maxImages, unsigned int maxNumResiduals, const ref_bundling_params* p) {
    GlobalBundlingState& g = GlobalBundlingState::get();
    g.s_numLocalNonLinIterations = p->numLocalNonLinIterations; g.s_numGlobalNonLinIterations = p->numGlobalNonLinIterations;
    g.s_submapSize = p->submapSize; g.s_denseOverlapCheckSubsampleFactor = p->denseOverlapCheckSubsampleFactor;
    g.s_optMaxResThresh = p->optMaxResThresh; g.s_denseDistThresh = p->denseDistThresh; g.s_denseNormalThresh = p->denseNormalThresh;
    g.s_denseColorThresh = p->denseColorThresh; g.s_denseColorGradientMin = p->denseColorGradientMin;
    g.s_denseDepthMin = p->denseDepthMin; g.s_denseDepthMax = p->denseDepthMax;
    g.s_useComprehensiveFrameInvalidation = p->useComprehensiveFrameInvalidation != 0; g.s_useLocalDense = p->useLocalDense != 0;
    g.s_recordSolverConvergence = p->recordSolverConvergence != 0;
    g.s_enableGlobalTimings = false; g.s_verbose = false;
    SBA* s = new SBA();
    s->init(maxImages, maxNumResiduals);
    return s;
}
void ref_sba_destroy(void* s) { delete (SBA*)s; }
void ref_sba_set_global_weights(void* s, const float* ws, const float* wd, const float* wc, unsigned int n, int useGlobalDenseOpt) {
    ((SBA*)s)->setGlobalWeights(std::vector<float>(ws, ws + n), std::vector<float>(wd, wd + n), std::vector<float>(wc, wc + n), useGlobalDenseOpt != 0);
}
// the weight schedules the constructor builds (n floats each): which = 0 local / 1 global
unsigned int ref_sba_get_weights(void* s, int which, float* ws, float* wd, float* wc) {
    SBA* a = (SBA*)s;
    const std::vector<float>& S = which ? a->m_globalWeightsSparse : a->m_localWeightsSparse;
    const std::vector<float>& D = which ? a->m_globalWeightsDenseDepth : a->m_localWeightsDenseDepth;
    const std::vector<float>& Cc = which ? a->m_globalWeightsDenseColor : a->m_localWeightsDenseColor;
    for (size_t i = 0; i < S.size(); ++i) { ws[i] = S[i]; wd[i] = D[i]; wc[i] = Cc[i]; }
    return (unsigned int)S.size();
}
// SBA::align on the manager's images and global correspondences.  valid[] goes in (m_validImages, mirrored to the "device" copy) and
// comes out (CheckForInvalidFrames* writes the device copy); transforms in and out; cached frames = those of the manager handle
// (ref_siftmgr_set_cached_frame) when cacheW > 0.  Returns `removed`.
int ref_sba_align(void* s, ref_siftmgr* h, int* valid, unsigned int currentFrame, unsigned int cacheW, unsigned int cacheH, const float* cacheIntrinsics16,
                  float* transforms16, unsigned int maxNumIters, unsigned int numPCGits, int useVerify, int isLocal, int isStart, int isEnd, unsigned int revalidateIdx,
                  float* maxResidualOut, int* useVerificationOut, float* convergenceOut) {
    SBA* a = (SBA*)s;
    SIFTImageManager* m = h->m;
    const unsigned int n = m->getNumImages();
    for (unsigned int i = 0; i < n; ++i) m->m_validImages[i] = valid[i];
    m->updateGPUValidImages();
    m->setCurrentFrame(currentFrame);
    CUDACache* cache = nullptr;
    if (cacheW) {
        cache = new CUDACache(cacheW, cacheH, cacheW, cacheH, n, mat4f(cacheIntrinsics16));      // input size = cached size: intrinsics unchanged
        cache->setCachedFrames(std::vector<CUDACachedFrame>(h->frames, h->frames + n));             // copies the frame data into its own buffers
        cache->setCurrentFrame(n);
    }
    const bool removed = a->align(m, cache, (float4x4*)transforms16, maxNumIters, numPCGits, useVerify != 0, isLocal != 0, false, isStart != 0, isEnd != 0, false, revalidateIdx);
    for (unsigned int i = 0; i < n; ++i) valid[i] = m->d_validImages[i];
    *maxResidualOut = a->getMaxResidual();
    *useVerificationOut = a->useVerification() ? 1 : 0;
    if (convergenceOut) { const std::vector<float>& c = a->m_solver->getConvergenceAnalysis(); for (size_t i = 0; i < c.size(); ++i) convergenceOut[i] = c[i]; }
    delete cache;
    return removed ? 1 : 0;
}

}
