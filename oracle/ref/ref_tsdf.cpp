// oracle/ref/ref_tsdf.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// C entry points over the REFERENCE's own voxel-hash code, compiled from where it lies:
//   DepthSensing/CUDASceneRepHashSDF.cu   (kernels resetHeap/resetHash/alloc/compactifyHashAllInOne/integrateDepthMap<deIntegrate>/
//                                          garbageCollectIdentify/garbageCollectFree and their extern "C" launch wrappers)
//   DepthSensing/VoxelUtilHashSDF.h       (HashDataStruct: computeHashPos, worldToVirtualVoxelPos, virtualVoxelPosToSDFBlock,
//                                          delinearizeVoxelIndex, allocBlock, getHashEntryForSDFBlockPos, deleteHashEntryElement, ...)
//   DepthSensing/DepthCameraUtil.h, CUDAHashParams.h, CUDADepthCameraParams.h
// through shim/cuda_runtime.h and the serial block emulator emu.h.  This file only sequences the launch wrappers the way
// CUDASceneRepHashSDF.h does (integrate :65-83, deIntegrate :85-107, garbageCollect :110-126, setLastRigidTransform :128-134,
// reset :147-155, alloc :328-352, compactifyHashEntries :355-391).  The host class itself is compiled as well (ref_scene_host.cpp, over the
// mLib stand-in) and produces the same volume byte for byte (tests/test_ref_pin_cpu.py).
// Threads of the emulated launches run in index order, so the physical slot / heap order of the result is ONE of the orders the
// CUDA reference can produce; parity targets are the order-independent quantities.
#include "CUDASceneRepHashSDF.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)

HashParams c_hashParams;
DepthCameraParams c_depthCameraParams;
extern "C" void updateConstantHashParams(const HashParams& p) { c_hashParams = p; }
extern "C" void updateConstantDepthCameraParams(const DepthCameraParams& p) { c_depthCameraParams = p; }

static_assert(sizeof(HashParams) == 224, "HashParams layout");
static_assert(sizeof(HashEntry) == 32 && sizeof(Voxel) == 12, "HashEntry / Voxel layout");

#include "ref_scene.h"

static unsigned int heapFree(ref_scene* s) { return s->data.d_heapCounter[0] + 1; }

static void setLastRigidTransform(ref_scene* s, const float* T) {                 // :128-134
    s->params.m_rigidTransform = float4x4(T);
    s->params.m_rigidTransformInverse = s->params.m_rigidTransform.getInverse();
    s->data.updateParams(s->params);
}
static void compactify(ref_scene* s) {                                             // :355-391
    s->params.m_numOccupiedBlocks = compactifyHashAllInOneCUDA(s->data, s->params);
    s->data.updateParams(s->params);
}

extern "C" {

ref_scene* ref_scene_create(const void* hashParams) {
    ref_scene* s = new ref_scene();
    memcpy(&s->params, hashParams, sizeof(HashParams));
    s->data.allocate(s->params, true);
    s->params.m_rigidTransform.setIdentity(); s->params.m_rigidTransformInverse.setIdentity(); s->params.m_numOccupiedBlocks = 0;
    s->data.updateParams(s->params);
    resetCUDA(s->data, s->params);                                                  // reset() :147-155
    return s;
}
void ref_scene_destroy(ref_scene* s) { if (s) { s->data.free(); delete s; } }

void ref_scene_integrate(ref_scene* s, const float* T, const float* depth, const unsigned char* colorRGBX, const void* cam) {   // :65-83
    DepthCameraParams cp; memcpy(&cp, cam, sizeof cp);
    DepthCameraData::updateParams(cp);
    DepthCameraData d(depth, (const uchar4*)colorRGBX);
    bindInputDepthColorTextures(d, cp.m_imageWidth, cp.m_imageHeight);
    setLastRigidTransform(s, T);
    unsigned int prevFree = heapFree(s);                                            // alloc :328-352
    while (1) {
        resetHashBucketMutexCUDA(s->data, s->params);
        allocCUDA(s->data, s->params, d, cp, nullptr);
        const unsigned int currFree = heapFree(s);
        if (prevFree != currFree) prevFree = currFree; else break;
    }
    compactify(s);
    integrateDepthMapCUDA(s->data, s->params, d, cp);
    s->numIntegrated++;
}

void ref_scene_deintegrate(ref_scene* s, const float* T, const float* depth, const unsigned char* colorRGBX, const void* cam) {  // :85-107
    DepthCameraParams cp; memcpy(&cp, cam, sizeof cp);
    DepthCameraData::updateParams(cp);
    DepthCameraData d(depth, (const uchar4*)colorRGBX);
    bindInputDepthColorTextures(d, cp.m_imageWidth, cp.m_imageHeight);
    setLastRigidTransform(s, T);
    compactify(s);
    deIntegrateDepthMapCUDA(s->data, s->params, d, cp);
    s->numIntegrated--;
}

void ref_scene_compactify(ref_scene* s, const float* T, const void* cam) {          // setLastRigidTransformAndCompactify :136-139
    DepthCameraParams cp; memcpy(&cp, cam, sizeof cp);
    DepthCameraData::updateParams(cp);
    setLastRigidTransform(s, T);
    compactify(s);
}

void ref_scene_garbage_collect(ref_scene* s) {                                      // :110-126
    if (s->params.m_numOccupiedBlocks > 0) {
        garbageCollectIdentifyCUDA(s->data, s->params);
        resetHashBucketMutexCUDA(s->data, s->params);
        garbageCollectFreeCUDA(s->data, s->params);
    }
}

const void* ref_scene_hash(ref_scene* s) { return s->data.d_hash; }
const void* ref_scene_heap(ref_scene* s) { return s->data.d_heap; }
unsigned int ref_scene_heap_counter(ref_scene* s) { return s->data.d_heapCounter[0]; }
const void* ref_scene_voxels(ref_scene* s) { return s->data.d_SDFBlocks; }
const void* ref_scene_compactified(ref_scene* s) { return s->data.d_hashCompactified; }
unsigned int ref_scene_num_occupied(ref_scene* s) { return s->params.m_numOccupiedBlocks; }

// the integer maps, called on the reference's own HashDataStruct (VoxelUtilHashSDF.h:226-344)
unsigned int ref_compute_hash_pos(unsigned int numBuckets, int x, int y, int z) {
    HashParams p = c_hashParams; const HashParams keep = c_hashParams;
    p.m_hashNumBuckets = numBuckets; c_hashParams = p;
    HashDataStruct h;
    const unsigned int r = h.computeHashPos(make_int3(x, y, z));
    c_hashParams = keep;
    return r;
}
void ref_world_to_block(float voxelSize, const float* w, int* out6) {
    HashParams p = c_hashParams; const HashParams keep = c_hashParams;
    p.m_virtualVoxelSize = voxelSize; c_hashParams = p;
    HashDataStruct h;
    const int3 v = h.worldToVirtualVoxelPos(make_float3(w[0], w[1], w[2]));
    const int3 b = h.virtualVoxelPosToSDFBlock(v);
    out6[0] = v.x; out6[1] = v.y; out6[2] = v.z; out6[3] = b.x; out6[4] = b.y; out6[5] = b.z;
    c_hashParams = keep;
}
void ref_delinearize_voxel_index(unsigned int idx, unsigned int* out3) {
    HashDataStruct h;
    const uint3 r = h.delinearizeVoxelIndex(idx);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
unsigned int ref_linearize_voxel_pos(int x, int y, int z) {
    HashDataStruct h;
    return h.linearizeVoxelPos(make_int3(x, y, z));
}
void ref_mat4_inverse(const float* m, float* out) {        // float4x4::getInverse, cuda_SimpleMatrixUtil.h
    const float4x4 r = float4x4(m).getInverse();
    memcpy(out, r.entries, 64);
}

}  // extern "C"
