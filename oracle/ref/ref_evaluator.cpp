// oracle/ref/ref_evaluator.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's CorrespondenceEvaluator (CorrespondenceEvaluator.h / .cpp compiled as they are, with -DEVALUATE_SPARSE_CORRESPONDENCES for
// these two translation units only: computeCachedData :10-45, evaluate :47-124, computeCorrespondences :126-252, computeOverlap :254-277,
// computeNormals, computeCameraSpacePositions) on the SIFTImageManager and CUDACache of a reference Bundler (ref_bundler.cpp).  The class works
// on mLib images and vectors: shim/mlib_standin.h supplies them as plain containers / element arithmetic - with three conventions of
// mLib that the stand-in can only assume (they are stated where tests/oracle_eval.py assumes the same): a default-constructed vec3f is
// (0, 0, 0), mat4f * vec4f sums each row left to right, math::round rounds half away from zero.
#define private public
#define protected public
#include "CorrespondenceEvaluator.h"
#include "CUDACache.h"
#undef private
#undef protected

extern "C" {

void* ref_evaluator_create(const float* trajectory16, unsigned int n, const char* logPrefix) {
    std::vector<mat4f> T(n);
    for (unsigned int i = 0; i < n; ++i) T[i] = mat4f(trajectory16 + 16 * (size_t)i);
    return new CorrespondenceEvaluator(T, logPrefix ? logPrefix : "");
}
void ref_evaluator_destroy(void* e) { delete (CorrespondenceEvaluator*)e; }
void ref_evaluator_finish_logging(void* e) { ((CorrespondenceEvaluator*)e)->finishLoggingToFile(); }
// evaluate() on a bundler's key points, current matches and cached frames (ref_bundler_sift_manager / ref_bundler_cuda_cache); out3 = numCorrect, numDetected, numTotal
void ref_evaluator_evaluate(void* e, void* siftManager, void* cudaCache, const float* siftIntrinsicsInv16, int filtered, int recomputeCache, int clearCache, const char* corrType, unsigned int* out3) {
    const CorrEvaluation r = ((CorrespondenceEvaluator*)e)->evaluate((const SIFTImageManager*)siftManager, (const CUDACache*)cudaCache, mat4f(siftIntrinsicsInv16), filtered != 0, recomputeCache != 0,
                                                                     clearCache != 0, corrType);
    out3[0] = r.numCorrect; out3[1] = r.numDetected; out3[2] = r.numTotal;
}
void ref_evaluator_has_gt_overlap(void* e, unsigned char* out, unsigned int n) {
    const std::vector<bool>& v = ((CorrespondenceEvaluator*)e)->m_cacheHasGTCorrByOverlap;
    for (unsigned int i = 0; i < n; ++i) out[i] = i < v.size() && v[i] ? 1 : 0;
}

}
