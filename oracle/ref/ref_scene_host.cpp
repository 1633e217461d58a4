// oracle/ref/ref_scene_host.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's host class of the voxel-hash volume, DepthSensing/CUDASceneRepHashSDF.h (compiled as it is: parametersFromGlobalAppState
// :39-59, integrate :65-83, deIntegrate :85-107, garbageCollect :110-126, setLastRigidTransform[AndCompactify] :128-140, reset :147-155,
// alloc :328-352, compactifyHashEntries :355-391), on the kernels of CUDASceneRepHashSDF.cu that ref_tsdf.cpp carries.  ref_tsdf.cpp
// sequences the same launch wrappers by hand (it predates the mLib stand-in); tests/test_ref_pin_cpu.py requires both to produce the same
// volume byte for byte.  prefixSumStub belongs to the scan-based compaction the header has commented out (:362-376); CUDAScan's
// constructor only allocates.
#define private public
#include "CUDASceneRepHashSDF.h"
#undef private

extern "C" void prefixSumStub(int*, int*, int*, int*, unsigned int, unsigned int, unsigned int, unsigned int, unsigned int) { throw std::runtime_error("prefixSumStub: the scan-based compaction is not part of the path"); }

extern "C" {

// HashParams as DepthSensing.cpp builds them: CUDASceneRepHashSDF::parametersFromGlobalAppState(GlobalAppState::get())
void ref_hash_params_from_global_app_state(unsigned int hashNumBuckets, unsigned int hashMaxCollisionLinkedListSize, unsigned int hashNumSDFBlocks, float SDFVoxelSize,
                                           float SDFMaxIntegrationDistance, float SDFTruncation, float SDFTruncationScale, unsigned int SDFIntegrationWeightSample,
                                           unsigned int SDFIntegrationWeightMax, void* outHashParams) {
    GlobalAppState& g = GlobalAppState::get();
    g.s_hashNumBuckets = hashNumBuckets; g.s_hashMaxCollisionLinkedListSize = hashMaxCollisionLinkedListSize; g.s_hashNumSDFBlocks = hashNumSDFBlocks;
    g.s_SDFVoxelSize = SDFVoxelSize; g.s_SDFMaxIntegrationDistance = SDFMaxIntegrationDistance; g.s_SDFTruncation = SDFTruncation;
    g.s_SDFTruncationScale = SDFTruncationScale; g.s_SDFIntegrationWeightSample = SDFIntegrationWeightSample; g.s_SDFIntegrationWeightMax = SDFIntegrationWeightMax;
    g.s_streamingVoxelExtents = vec3f(1.0f, 1.0f, 1.0f); g.s_streamingGridDimensions = vec3i(257, 257, 257); g.s_streamingMinGridPos = vec3i(-128, -128, -128);
    g.s_streamingInitialChunkListSize = 2000;
    HashParams p = CUDASceneRepHashSDF::parametersFromGlobalAppState(g);
    p.m_numOccupiedBlocks = 0;        // not set by the function (the constructor path sets it in reset())
    memcpy(outHashParams, &p, sizeof p);
}

void* ref_hscene_create(const void* hashParams) {
    HashParams p; memcpy(&p, hashParams, sizeof p);
    GlobalAppState::get().s_garbageCollectionEnabled = true; GlobalAppState::get().s_streamingEnabled = false; GlobalAppState::get().s_timingsDetailledEnabled = false;
    return new CUDASceneRepHashSDF(p);
}
void ref_hscene_destroy(void* s) { delete (CUDASceneRepHashSDF*)s; }
static DepthCameraParams camera(const void* cam) { DepthCameraParams cp; memcpy(&cp, cam, sizeof cp); DepthCameraData::updateParams(cp); return cp; }
void ref_hscene_integrate(void* s, const float* T, const float* depth, const unsigned char* colorRGBX, const void* cam) {
    const DepthCameraParams cp = camera(cam);
    DepthCameraData d(depth, (const uchar4*)colorRGBX);
    ((CUDASceneRepHashSDF*)s)->integrate(mat4f(T), d, cp, nullptr);
}
void ref_hscene_deintegrate(void* s, const float* T, const float* depth, const unsigned char* colorRGBX, const void* cam) {
    const DepthCameraParams cp = camera(cam);
    DepthCameraData d(depth, (const uchar4*)colorRGBX);
    ((CUDASceneRepHashSDF*)s)->deIntegrate(mat4f(T), d, cp, nullptr);
}
void ref_hscene_compactify(void* s, const float* T, const void* cam) { camera(cam); ((CUDASceneRepHashSDF*)s)->setLastRigidTransformAndCompactify(mat4f(T)); }
void ref_hscene_garbage_collect(void* s) { ((CUDASceneRepHashSDF*)s)->garbageCollect(); }
const void* ref_hscene_hash(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashData().d_hash; }
const void* ref_hscene_heap(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashData().d_heap; }
unsigned int ref_hscene_heap_counter(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashData().d_heapCounter[0]; }
const void* ref_hscene_voxels(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashData().d_SDFBlocks; }
const void* ref_hscene_compactified(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashData().d_hashCompactified; }
unsigned int ref_hscene_num_occupied(void* s) { return ((CUDASceneRepHashSDF*)s)->getHashParams().m_numOccupiedBlocks; }
unsigned int ref_hscene_num_integrated(void* s) { return ((CUDASceneRepHashSDF*)s)->getNumIntegratedFrames(); }

}

// CUDARayCastSDF::parametersFromGlobalAppState (CUDARayCastSDF.h:24-52): the static function only - the class's render() drives Direct3D
#include "CUDARayCastSDF.h"
extern "C" void ref_ray_cast_params_from_global_app_state(unsigned int rayCastWidth, unsigned int rayCastHeight, unsigned int integrationWidth, unsigned int integrationHeight,
                                                          float renderDepthMin, float renderDepthMax, float SDFRayIncrementFactor, float SDFTruncation,
                                                          float SDFRayThresSampleDistFactor, float SDFRayThresDistFactor, int SDFUseGradients, unsigned int hashNumSDFBlocks,
                                                          const float* intrinsics16, void* outRayCastParams) {
    GlobalAppState& g = GlobalAppState::get();
    g.s_rayCastWidth = rayCastWidth; g.s_rayCastHeight = rayCastHeight; g.s_integrationWidth = integrationWidth; g.s_integrationHeight = integrationHeight;
    g.s_renderDepthMin = renderDepthMin; g.s_renderDepthMax = renderDepthMax; g.s_SDFRayIncrementFactor = SDFRayIncrementFactor; g.s_SDFTruncation = SDFTruncation;
    g.s_SDFRayThresSampleDistFactor = SDFRayThresSampleDistFactor; g.s_SDFRayThresDistFactor = SDFRayThresDistFactor; g.s_SDFUseGradients = SDFUseGradients != 0;
    g.s_hashNumSDFBlocks = hashNumSDFBlocks;
    const mat4f K(intrinsics16);
    RayCastParams p; memset(&p, 0, sizeof p);
    p = CUDARayCastSDF::parametersFromGlobalAppState(g, K, K.getInverse());
    memcpy(outRayCastParams, &p, sizeof p);
}
extern "C" unsigned int ref_sizeof_ray_cast_params() { return (unsigned int)sizeof(RayCastParams); }
