// Host-level operators of the hot path: CUDAImageManager, Bundler (+ the SBA wrapper), TrajectoryManager, OnlineBundler and
// the headless frame loop, behind the C ABI of include/bf_pipeline.h.  Control flow follows the reference
// (CUDAImageManager.cpp, Bundler.cpp, SBA.cpp, OnlineBundler.cpp/.cu, TrajectoryManager.cpp, DepthSensing.cpp:723-762,
// :854-902, :966-1095; paths relative to /root/reference/FriedLiver/Source); the execution model is the MI355X one:
//  * one HIP stream carries a frame from ingest to integration; the only host read-back per frame is a 96-byte record
//    (last matched frame, validity, #residuals, #keys, SIFT pose) — the reference blocks on ~20 small D2H copies per frame;
//  * all frames stay resident in HBM at integration resolution (the reference keeps them on the host and re-uploads a
//    frame for every integrate / de-integrate), slab-allocated so that ingest never calls hipMalloc in steady state;
//  * bundling reads the ingest buffers in place (the reference's copyToBundling makes three device-to-device copies per
//    frame because its bundler lives on a second GPU).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <limits>
#include <list>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/bf_pipeline.h"
#include "../../include/bf_comm.h"
#include "bf_device.h"
#include "bf_internal.h"
#include "bf_se3.h"

using namespace bf;

#define BF_TRY(expr) do { int _rc = (expr); if (_rc != BF_OK) return _rc; } while (0)

namespace {

const float NINF = -std::numeric_limits<float>::infinity();

m44 toM(const float* p) { m44 m; memcpy(m.e, p, 64); return m; }
m44 minfM() { m44 m; for (int i = 0; i < 16; ++i) m.e[i] = NINF; return m; }

// intrinsics rescale rule of CUDAImageManager.h:162-168 / OnlineBundlerHelper.h:46-50
m44 scaleIntrinsics(const float* K, uint32_t wOut, uint32_t hOut, uint32_t wIn, uint32_t hIn) {
    m44 m = toM(K);
    m.e[0] *= (float)wOut / (float)wIn;
    m.e[5] *= (float)hOut / (float)hIn;
    m.e[2] *= (float)(wOut - 1) / (float)(wIn - 1);
    m.e[6] *= (float)(hOut - 1) / (float)(hIn - 1);
    return m;
}

// ------------------------------------------------------------------------------------------------ OnlineBundler.cu
__global__ void k_sift_transform(uint32_t curFrameIndex, const m44* completeTrajectory, uint32_t lastValidCompleteTransform, m44* siftTrajectory,
                                 uint32_t curFrameIndexAll, const int* numFilt, const m44* filteredTransformsInv, m44* currIntegrateTrans) {   // :5-52
    for (int i = (int)curFrameIndex - 1; i >= 0; i--) {
        if (numFilt[i] > 0) {
            const uint32_t idxPrevSiftKnown = curFrameIndexAll - (curFrameIndex - (uint32_t)i);
            const m44 cur = mul44(siftTrajectory[idxPrevSiftKnown], filteredTransformsInv[i]);
            siftTrajectory[curFrameIndexAll] = cur;
            m44 transform;
            if (lastValidCompleteTransform == 0) transform = cur;
            else if (idxPrevSiftKnown < lastValidCompleteTransform) transform = mul44(completeTrajectory[idxPrevSiftKnown], filteredTransformsInv[i]);
            else {
                const m44 offset = mul44(inverse44(siftTrajectory[lastValidCompleteTransform]), siftTrajectory[idxPrevSiftKnown]);
                transform = mul44(mul44(completeTrajectory[lastValidCompleteTransform], offset), filteredTransformsInv[i]);
            }
            currIntegrateTrans[0] = transform;
            break;
        }
    }
}

// the same for a frame whose chunk-local matching ran elsewhere (chunk-parallel mode): `i` and filteredTransformsInv[i] arrive by value
__global__ void k_sift_transform_ext(uint32_t curFrameIndex, const m44* completeTrajectory, uint32_t lastValidCompleteTransform, m44* siftTrajectory,
                                     uint32_t curFrameIndexAll, int i, m44 filteredTransformInv, m44* currIntegrateTrans) {
    const uint32_t idxPrevSiftKnown = curFrameIndexAll - (curFrameIndex - (uint32_t)i);
    const m44 cur = mul44(siftTrajectory[idxPrevSiftKnown], filteredTransformInv);
    siftTrajectory[curFrameIndexAll] = cur;
    m44 transform;
    if (lastValidCompleteTransform == 0) transform = cur;
    else if (idxPrevSiftKnown < lastValidCompleteTransform) transform = mul44(completeTrajectory[idxPrevSiftKnown], filteredTransformInv);
    else {
        const m44 offset = mul44(inverse44(siftTrajectory[lastValidCompleteTransform]), siftTrajectory[idxPrevSiftKnown]);
        transform = mul44(mul44(completeTrajectory[lastValidCompleteTransform], offset), filteredTransformInv);
    }
    currIntegrateTrans[0] = transform;
}

// chunk worker: which previous frame of the chunk does frame `cur` chain from (k_sift_transform's scan), and its relative transform
__global__ void k_pick_relative(uint32_t cur, const int* numFilt, const m44* filteredTransformsInv, bf_chunk_frame_record* out) {
    out->prevLocal = -1;
    for (int i = (int)cur - 1; i >= 0; i--)
        if (numFilt[i] > 0) {
            out->prevLocal = i;
            for (int k = 0; k < 16; ++k) out->relInv[k] = filteredTransformsInv[i].e[k];
            break;
        }
}

__global__ void k_update_trajectory(const m44* globalTrajectory, m44* completeTrajectory, uint32_t numCompleteTransforms, const m44* localTrajectories,
                                    uint32_t numLocalTransformsPerTrajectory, const int* imageInvalidateList) {      // :73-92
    const uint32_t idxComplete = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t submapSize = numLocalTransformsPerTrajectory - 1;
    if (idxComplete < numCompleteTransforms) {
        const uint32_t idxGlobal = idxComplete / submapSize, idxLocal = idxComplete % submapSize;
        m44 r;
        if (imageInvalidateList[idxComplete] == 0) { for (int i = 0; i < 16; ++i) r.e[i] = BF_MINF; }
        else r = mul44(globalTrajectory[idxGlobal], localTrajectories[idxGlobal * numLocalTransformsPerTrajectory + idxLocal]);
        completeTrajectory[idxComplete] = r;
    }
}

__global__ void k_init_next_global(m44* globalTrajectory, uint32_t numGlobalTransforms, uint32_t initGlobalIdx, const m44* localTrajectories,
                                   uint32_t lastValidLocal, uint32_t numLocalTransformsPerTrajectory) {              // :116-126
    globalTrajectory[numGlobalTransforms] =
        mul44(globalTrajectory[initGlobalIdx], localTrajectories[numGlobalTransforms * numLocalTransformsPerTrajectory - (numLocalTransformsPerTrajectory - lastValidLocal)]);
}

__global__ void k_fill_identity(m44* T, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) T[i] = identity44();
}

}  // namespace

extern "C" {

int bf_compute_sift_transform(const float* d_currFilteredTransformsInv, const int32_t* d_numFilt, const float* d_completeTrajectory,
                              uint32_t lastValidCompleteTransform, float* d_siftTrajectory, uint32_t curFrameIndexAll, uint32_t curFrameIndex,
                              float* d_currIntegrateTrans, void* stream) {
    if (curFrameIndex == 0) return BF_OK;
    BF_REQUIRE(d_currFilteredTransformsInv && d_numFilt && d_completeTrajectory && d_siftTrajectory && d_currIntegrateTrans, "null argument");
    k_sift_transform<<<1, 1, 0, (hipStream_t)stream>>>(curFrameIndex, (const m44*)d_completeTrajectory, lastValidCompleteTransform, (m44*)d_siftTrajectory,
                                                      curFrameIndexAll, d_numFilt, (const m44*)d_currFilteredTransformsInv, (m44*)d_currIntegrateTrans);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_update_trajectory(const float* d_globalTrajectory, uint32_t numGlobalTransforms, float* d_completeTrajectory, uint32_t numCompleteTransforms,
                         const float* d_localTrajectories, uint32_t numLocalTransformsPerTrajectory, uint32_t numLocalTrajectories,
                         const int32_t* d_imageInvalidateList, void* stream) {
    (void)numGlobalTransforms; (void)numLocalTrajectories;
    if (numCompleteTransforms == 0) return BF_OK;
    BF_REQUIRE(d_globalTrajectory && d_completeTrajectory && d_localTrajectories && d_imageInvalidateList && numLocalTransformsPerTrajectory > 1, "bad argument");
    k_update_trajectory<<<div_up(numCompleteTransforms, 128), 128, 0, (hipStream_t)stream>>>((const m44*)d_globalTrajectory, (m44*)d_completeTrajectory,
                                                                                           numCompleteTransforms, (const m44*)d_localTrajectories,
                                                                                           numLocalTransformsPerTrajectory, d_imageInvalidateList);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_init_next_global_transform(float* d_globalTrajectory, uint32_t numGlobalTransforms, uint32_t initGlobalIdx, const float* d_localTrajectories,
                                  uint32_t lastValidLocal, uint32_t numLocalTransformsPerTrajectory, void* stream) {
    BF_REQUIRE(d_globalTrajectory && d_localTrajectories && numGlobalTransforms >= 1, "bad argument");
    k_init_next_global<<<1, 1, 0, (hipStream_t)stream>>>((m44*)d_globalTrajectory, numGlobalTransforms, initGlobalIdx, (const m44*)d_localTrajectories,
                                                        lastValidLocal, numLocalTransformsPerTrajectory);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

}  // extern "C"

// ================================================================================================ CUDAImageManager
struct bf_image_manager {
    bf_rgbd_sensor_desc sensor;
    bf_global_bundling_state gbs;
    uint32_t wInt = 0, hInt = 0, wSIFT = 0, hSIFT = 0;
    bool onGPU = true;
    bool scratch = false;          // storeFramesOnGPU == 2: one frame slot that every process() overwrites (chunk workers keep no history)
    hipStream_t stream = nullptr;
    m44 depthIntrinsics, depthIntrinsicsInv, colorIntrinsics, colorIntrinsicsInv, depthExtrinsics, depthExtrinsicsInv, siftDepthIntrinsics;
    // The ingest buffers at sensor resolution, NSETS sets (frame n uses set n % NSETS): frame n + 1 can be ingested (on its own stream) while the feature detection of frame n
    // still reads frame n's set.  d_depthInputRaw / d_depthInputFiltered / d_colorInput always name the set of the frame ingested last - what
    // CUDAImageManager's members of these names hold after process().  inputGuard[k]: an event of the caller's that the ingest waits for before it
    // overwrites set k (the consumer of the frame that used the set last), or null.
    float *d_depthInputRaw = nullptr, *d_depthInputFiltered = nullptr;
    uint8_t* d_colorInput = nullptr;
    static const uint32_t NSETS = 4;     // frame n uses set n % NSETS (== the staging slot of the frame's detection, BF_STAGE_SLOTS below)
    float *rawSet[NSETS] = {}, *filtSet[NSETS] = {};
    uint8_t* colSet[NSETS] = {};
    hipEvent_t inputGuard[NSETS] = {};
    // frames at integration resolution: slabs of SLAB frames in HBM (onGPU) or host vectors + one staging pair (reference default)
    static const uint32_t SLAB = 256;
    std::vector<float*> depthSlabs; std::vector<uint8_t*> colorSlabs;
    // optional third plane per stored frame: depth and colour interleaved as 8-byte texels (bf_image_manager_set_store_texels), made once at ingest for the
    // voxel update's gathers - every operator on the frame would otherwise interleave it again (11 times per input frame in the saturated loop)
    bool storeTexels = false; std::vector<uint8_t*> texelSlabs;
    std::vector<std::vector<float>> hostDepth; std::vector<std::vector<uint8_t>> hostColor;
    float* d_stageDepth = nullptr; uint8_t* d_stageColor = nullptr;
    int activeDepth = -1, activeColor = -1;
    uint32_t currFrame = 0;
    size_t nInt() const { return (size_t)wInt * hInt; }
};

extern "C" {

int bf_image_manager_create(uint32_t wInt, uint32_t hInt, uint32_t wSIFT, uint32_t hSIFT, const bf_rgbd_sensor_desc* sensor,
                            const bf_global_bundling_state* gbs, int storeFramesOnGPU, bf_image_manager** out) {
    BF_REQUIRE(sensor && gbs && out, "null argument");
    BF_REQUIRE(wInt > 1 && hInt > 1 && sensor->depthWidth > 1 && sensor->depthHeight > 1 && sensor->colorWidth > 1 && sensor->colorHeight > 1, "bad image size");
    bf_image_manager* im = new bf_image_manager;
    im->sensor = *sensor; im->gbs = *gbs;
    im->wInt = wInt; im->hInt = hInt; im->wSIFT = wSIFT; im->hSIFT = hSIFT; im->onGPU = storeFramesOnGPU != 0; im->scratch = storeFramesOnGPU == 2;
    im->siftDepthIntrinsics = toM(sensor->depthIntrinsics);
    im->depthIntrinsics = scaleIntrinsics(sensor->depthIntrinsics, wInt, hInt, sensor->depthWidth, sensor->depthHeight);
    im->depthIntrinsicsInv = inverse44(im->depthIntrinsics);
    im->colorIntrinsics = scaleIntrinsics(sensor->colorIntrinsics, wInt, hInt, sensor->colorWidth, sensor->colorHeight);
    im->colorIntrinsicsInv = inverse44(im->colorIntrinsics);
    im->depthExtrinsics = toM(sensor->depthExtrinsics);
    im->depthExtrinsicsInv = inverse44(im->depthExtrinsics);
    const size_t nd = (size_t)sensor->depthWidth * sensor->depthHeight, nc = (size_t)sensor->colorWidth * sensor->colorHeight;
    for (uint32_t k = 0; k < (im->scratch ? 1u : bf_image_manager::NSETS); ++k) {
        BF_HIP_TRY(BF_MALLOC((void**)&im->rawSet[k], nd * 4));
        BF_HIP_TRY(BF_MALLOC((void**)&im->filtSet[k], nd * 4));
        BF_HIP_TRY(BF_MALLOC((void**)&im->colSet[k], nc * 4));
    }
    if (im->scratch) for (uint32_t k = 1; k < bf_image_manager::NSETS; ++k) { im->rawSet[k] = im->rawSet[0]; im->filtSet[k] = im->filtSet[0]; im->colSet[k] = im->colSet[0]; }
    im->d_depthInputRaw = im->rawSet[0]; im->d_depthInputFiltered = im->filtSet[0]; im->d_colorInput = im->colSet[0];
    if (!im->onGPU) {
        BF_HIP_TRY(BF_MALLOC((void**)&im->d_stageDepth, im->nInt() * 4));
        BF_HIP_TRY(BF_MALLOC((void**)&im->d_stageColor, im->nInt() * 4));
    }
    *out = im;
    return BF_OK;
}

int bf_image_manager_reset(bf_image_manager* im) {
    BF_REQUIRE(im, "null manager");
    (void)hipStreamSynchronize(im->stream);
    for (auto p : im->depthSlabs) (void)hipFree(p);
    for (auto p : im->colorSlabs) (void)hipFree(p);
    for (auto p : im->texelSlabs) (void)hipFree(p);
    im->texelSlabs.clear();
    im->depthSlabs.clear(); im->colorSlabs.clear(); im->hostDepth.clear(); im->hostColor.clear();
    im->activeDepth = im->activeColor = -1;
    im->currFrame = 0;
    return BF_OK;
}

int bf_image_manager_destroy(bf_image_manager* im) {
    if (!im) return BF_OK;
    bf_image_manager_reset(im);
    for (uint32_t k = 0; k < (im->scratch ? 1u : bf_image_manager::NSETS); ++k) { (void)hipFree(im->rawSet[k]); (void)hipFree(im->filtSet[k]); (void)hipFree(im->colSet[k]); }
    (void)hipFree(im->d_stageDepth); (void)hipFree(im->d_stageColor);
    delete im;
    return BF_OK;
}

int bf_image_manager_set_stream(bf_image_manager* im, void* s) { BF_REQUIRE(im, "null manager"); im->stream = (hipStream_t)s; return BF_OK; }
// keep, per stored frame, its depth and colour interleaved as 8-byte texels as well (before the first frame; frames on the GPU only): +8 bytes per pixel and frame
int bf_image_manager_set_store_texels(bf_image_manager* im, int enable) {
    BF_REQUIRE(im, "null manager");
    BF_REQUIRE(im->currFrame == 0, "set_store_texels after the first frame");
    im->storeTexels = enable != 0 && im->onGPU && !im->scratch;
    return BF_OK;
}
int bf_image_manager_get_integrate_frame_texels(bf_image_manager* im, uint32_t frame, const void** d_texels) {
    BF_REQUIRE(im && d_texels, "null argument");
    *d_texels = nullptr;
    if (!im->storeTexels || frame >= im->currFrame || frame / bf_image_manager::SLAB >= im->texelSlabs.size()) return BF_OK;
    *d_texels = im->texelSlabs[frame / bf_image_manager::SLAB] + (size_t)(frame % bf_image_manager::SLAB) * im->nInt() * 8;
    return BF_OK;
}
// see bf_image_manager::inputGuard: set k (0 / 1) is overwritten by frames of parity k; `hip_event` (or null) is recorded by whoever reads a frame's
// input buffers on ANOTHER stream than the ingest's, after its last read
int bf_image_manager_set_input_guard(bf_image_manager* im, uint32_t set, void* hip_event) {
    BF_REQUIRE(im && set < bf_image_manager::NSETS, "bad argument");
    im->inputGuard[set] = (hipEvent_t)hip_event;
    return BF_OK;
}

static int im_process(bf_image_manager* im, const float* depth, const uint8_t* color, hipMemcpyKind kind, int* gotFrame) {
    BF_REQUIRE(im && gotFrame, "null argument");
    *gotFrame = 0;
    if (!depth || !color) return BF_OK;                     // sensor->processDepth()/processColor() returned false
    if (!im->scratch && im->currFrame + 1 > im->gbs.s_maxNumImages * im->gbs.s_submapSize) return BF_OK;      // .cpp:26-29 "reached max #images"
    const bf_rgbd_sensor_desc& sn = im->sensor;
    const size_t nd = (size_t)sn.depthWidth * sn.depthHeight, nc = (size_t)sn.colorWidth * sn.colorHeight, ni = im->nInt();
    hipStream_t st = im->stream;
    const uint32_t f = im->scratch ? 0u : im->currFrame;
    float* frameDepth = nullptr; uint8_t* frameColor = nullptr;
    if (im->onGPU) {
        if (f / bf_image_manager::SLAB >= im->depthSlabs.size()) {
            float* pd = nullptr; uint8_t* pc = nullptr;
            const size_t slots = im->scratch ? 1 : bf_image_manager::SLAB;
            BF_HIP_TRY(BF_MALLOC((void**)&pd, ni * 4 * slots));
            BF_HIP_TRY(BF_MALLOC((void**)&pc, ni * 4 * slots));
            im->depthSlabs.push_back(pd); im->colorSlabs.push_back(pc);
            if (im->storeTexels) { uint8_t* pt = nullptr; BF_HIP_TRY(BF_MALLOC((void**)&pt, ni * 8 * slots)); im->texelSlabs.push_back(pt); }
        }
        frameDepth = im->depthSlabs[f / bf_image_manager::SLAB] + (size_t)(f % bf_image_manager::SLAB) * ni;
        frameColor = im->colorSlabs[f / bf_image_manager::SLAB] + (size_t)(f % bf_image_manager::SLAB) * ni * 4;
    } else {
        im->hostDepth.emplace_back(ni); im->hostColor.emplace_back(ni * 4);
    }
    {   // this frame's input set
        const uint32_t k = im->currFrame % bf_image_manager::NSETS;
        if (im->inputGuard[k]) BF_HIP_TRY(hipStreamWaitEvent(st, im->inputGuard[k], 0));
        im->d_depthInputRaw = im->rawSet[k]; im->d_depthInputFiltered = im->filtSet[k]; im->d_colorInput = im->colSet[k];
    }
    // ---- the frame is in device memory already, sensor and integration resolution coincide, erosion and depth filter on (the frame loop's usual case): the same
    // images in three launches - erosion 1 straight from the caller's depth with the colour copies riding along, erosion 2, depth filter writing the stored frame too
    if (kind == hipMemcpyDeviceToDevice && im->onGPU && !im->storeTexels && nc == nd && sn.colorWidth == im->wInt && sn.colorHeight == im->hInt && sn.depthWidth == im->wInt &&
        sn.depthHeight == im->hInt && im->gbs.s_erodeSIFTdepth && im->gbs.s_depthFilter) {
        BF_TRY(bf_image_erode_depth_map_and_copy(im->d_depthInputFiltered, depth, 3, sn.depthWidth, sn.depthHeight, 0.05f, 0.3f, color, im->d_colorInput, frameColor, st));
        BF_TRY(bf_image_erode_depth_map(im->d_depthInputRaw, im->d_depthInputFiltered, 3, sn.depthWidth, sn.depthHeight, 0.05f, 0.3f, st));
        BF_TRY(bf_image_gauss_filter_depth_map2(im->d_depthInputFiltered, frameDepth, im->d_depthInputRaw, im->gbs.s_depthSigmaD, im->gbs.s_depthSigmaR, sn.depthWidth, sn.depthHeight, st));
        im->currFrame++;
        *gotFrame = 1;
        return BF_OK;
    }
    // ---- colour  (.cpp:39-60)
    BF_HIP_TRY(hipMemcpyAsync(im->d_colorInput, color, nc * 4, kind, st));
    const bool sameC = sn.colorWidth == im->wInt && sn.colorHeight == im->hInt;
    uint8_t* colorDst = im->onGPU ? frameColor : im->d_stageColor;
    if (sameC) BF_HIP_TRY(hipMemcpyAsync(colorDst, im->d_colorInput, ni * 4, hipMemcpyDeviceToDevice, st));
    else BF_TRY(bf_image_resample_uchar4(colorDst, im->wInt, im->hInt, im->d_colorInput, sn.colorWidth, sn.colorHeight, st));
    // ---- depth  (.cpp:66-112): two 7x7 erosions ping-pong (the twice-eroded map ends in d_depthInputRaw), then the range-gated Gaussian
    BF_HIP_TRY(hipMemcpyAsync(im->d_depthInputRaw, depth, nd * 4, kind, st));
    if (im->gbs.s_erodeSIFTdepth) {
        BF_TRY(bf_image_erode_depth_map(im->d_depthInputFiltered, im->d_depthInputRaw, 3, sn.depthWidth, sn.depthHeight, 0.05f, 0.3f, st));
        BF_TRY(bf_image_erode_depth_map(im->d_depthInputRaw, im->d_depthInputFiltered, 3, sn.depthWidth, sn.depthHeight, 0.05f, 0.3f, st));
    }
    if (im->gbs.s_depthFilter) BF_TRY(bf_image_gauss_filter_depth_map(im->d_depthInputFiltered, im->d_depthInputRaw, im->gbs.s_depthSigmaD, im->gbs.s_depthSigmaR,
                                                                     sn.depthWidth, sn.depthHeight, st));
    else BF_HIP_TRY(hipMemcpyAsync(im->d_depthInputFiltered, im->d_depthInputRaw, nd * 4, hipMemcpyDeviceToDevice, st));
    // ---- integration-resolution depth (.cpp:121-147): the filtered map, or the sensor's own samples when erosion is off
    const bool sameD = sn.depthWidth == im->wInt && sn.depthHeight == im->hInt;
    float* depthDst = im->onGPU ? frameDepth : im->d_stageDepth;
    if (sameD) BF_HIP_TRY(hipMemcpyAsync(depthDst, im->gbs.s_erodeSIFTdepth ? im->d_depthInputFiltered : im->d_depthInputRaw, ni * 4, hipMemcpyDeviceToDevice, st));
    else BF_TRY(bf_image_resample_float(depthDst, im->wInt, im->hInt, im->d_depthInputFiltered, sn.depthWidth, sn.depthHeight, st));
    if (im->onGPU && im->storeTexels && f / bf_image_manager::SLAB < im->texelSlabs.size())
        BF_TRY(bf_image_interleave_texels(im->texelSlabs[f / bf_image_manager::SLAB] + (size_t)(f % bf_image_manager::SLAB) * ni * 8, frameDepth, frameColor, (uint32_t)ni, st));
    if (!im->onGPU) {
        BF_HIP_TRY(hipMemcpyAsync(im->hostDepth.back().data(), im->d_stageDepth, ni * 4, hipMemcpyDeviceToHost, st));
        BF_HIP_TRY(hipMemcpyAsync(im->hostColor.back().data(), im->d_stageColor, ni * 4, hipMemcpyDeviceToHost, st));
        BF_HIP_TRY(hipStreamSynchronize(st));
        im->activeDepth = im->activeColor = (int)f;
    }
    im->currFrame++;
    *gotFrame = 1;
    return BF_OK;
}

int bf_image_manager_process(bf_image_manager* im, const float* h_depth, const uint8_t* h_color, int* gotFrame) {
    return im_process(im, h_depth, h_color, hipMemcpyHostToDevice, gotFrame);
}
int bf_image_manager_process_device(bf_image_manager* im, const float* d_depth, const uint8_t* d_color, int* gotFrame) {
    return im_process(im, d_depth, d_color, hipMemcpyDeviceToDevice, gotFrame);
}

int bf_image_manager_copy_to_bundling(bf_image_manager* im, float* d_depthRaw, float* d_depthFilt, uint8_t* d_color) {
    BF_REQUIRE(im && d_depthRaw && d_depthFilt && d_color, "null argument");
    const size_t nd = (size_t)im->sensor.depthWidth * im->sensor.depthHeight, nc = (size_t)im->sensor.colorWidth * im->sensor.colorHeight;
    BF_HIP_TRY(hipMemcpyAsync(d_depthRaw, im->d_depthInputRaw, nd * 4, hipMemcpyDeviceToDevice, im->stream));
    BF_HIP_TRY(hipMemcpyAsync(d_depthFilt, im->d_depthInputFiltered, nd * 4, hipMemcpyDeviceToDevice, im->stream));
    BF_HIP_TRY(hipMemcpyAsync(d_color, im->d_colorInput, nc * 4, hipMemcpyDeviceToDevice, im->stream));
    return BF_OK;
}

int bf_image_manager_get_input_gpu(bf_image_manager* im, const float** raw, const float** filt, const uint8_t** color) {
    BF_REQUIRE(im, "null manager");
    if (raw) *raw = im->d_depthInputRaw;
    if (filt) *filt = im->d_depthInputFiltered;
    if (color) *color = im->d_colorInput;
    return BF_OK;
}

int bf_image_manager_get_integrate_frame_gpu(bf_image_manager* im, uint32_t frame, const float** d_depth, const uint8_t** d_color) {
    BF_REQUIRE(im && d_depth && d_color && frame < im->currFrame, "frame out of range");
    const size_t ni = im->nInt();
    if (im->onGPU) {
        *d_depth = im->depthSlabs[frame / bf_image_manager::SLAB] + (size_t)(frame % bf_image_manager::SLAB) * ni;
        *d_color = im->colorSlabs[frame / bf_image_manager::SLAB] + (size_t)(frame % bf_image_manager::SLAB) * ni * 4;
    } else {                                       // one frame globally valid on the GPU at a time (.h:71-96)
        if (im->activeDepth != (int)frame) { BF_HIP_TRY(hipMemcpyAsync(im->d_stageDepth, im->hostDepth[frame].data(), ni * 4, hipMemcpyHostToDevice, im->stream)); im->activeDepth = (int)frame; }
        if (im->activeColor != (int)frame) { BF_HIP_TRY(hipMemcpyAsync(im->d_stageColor, im->hostColor[frame].data(), ni * 4, hipMemcpyHostToDevice, im->stream)); im->activeColor = (int)frame; }
        *d_depth = im->d_stageDepth; *d_color = im->d_stageColor;
    }
    return BF_OK;
}

// ManagedRGBDInputFrame::getDepthFrameCPU / getColorFrameCPU (CUDAImageManager.h:97-119): host copies of a stored frame at
// integration resolution (the reference keeps every frame on the host; here they live in HBM and are copied out on request)
int bf_image_manager_get_integrate_frame_cpu(bf_image_manager* im, uint32_t frame, float* h_depth, uint8_t* h_color) {
    BF_REQUIRE(im && frame < im->currFrame && (h_depth || h_color), "frame out of range");
    const size_t ni = im->nInt();
    if (!im->onGPU) {
        if (h_depth) memcpy(h_depth, im->hostDepth[frame].data(), ni * 4);
        if (h_color) memcpy(h_color, im->hostColor[frame].data(), ni * 4);
        return BF_OK;
    }
    const float* d_depth = nullptr; const uint8_t* d_color = nullptr;
    BF_TRY(bf_image_manager_get_integrate_frame_gpu(im, frame, &d_depth, &d_color));
    if (h_depth) BF_HIP_TRY(hipMemcpyAsync(h_depth, d_depth, ni * 4, hipMemcpyDeviceToHost, im->stream));
    if (h_color) BF_HIP_TRY(hipMemcpyAsync(h_color, d_color, ni * 4, hipMemcpyDeviceToHost, im->stream));
    BF_HIP_TRY(hipStreamSynchronize(im->stream));
    return BF_OK;
}

int bf_image_manager_get_curr_frame_number(bf_image_manager* im, uint32_t* out) {
    BF_REQUIRE(im && out, "null argument");
    BF_REQUIRE(im->currFrame > 0, "getCurrFrameNumber before the first process()");
    *out = im->currFrame - 1;
    return BF_OK;
}
int bf_image_manager_get_num_frames(bf_image_manager* im, uint32_t* out) { BF_REQUIRE(im && out, "null argument"); *out = im->currFrame; return BF_OK; }
int bf_image_manager_get_integration_size(bf_image_manager* im, uint32_t* w, uint32_t* h) { BF_REQUIRE(im && w && h, "null argument"); *w = im->wInt; *h = im->hInt; return BF_OK; }
int bf_image_manager_get_depth_intrinsics(bf_image_manager* im, float K[16], float Kinv[16]) {
    BF_REQUIRE(im, "null manager");
    if (K) memcpy(K, im->depthIntrinsics.e, 64);
    if (Kinv) memcpy(Kinv, im->depthIntrinsicsInv.e, 64);
    return BF_OK;
}
int bf_image_manager_get_depth_extrinsics(bf_image_manager* im, float E[16], float Einv[16]) {
    BF_REQUIRE(im, "null manager");
    if (E) memcpy(E, im->depthExtrinsics.e, 64);
    if (Einv) memcpy(Einv, im->depthExtrinsicsInv.e, 64);
    return BF_OK;
}
int bf_image_manager_get_sift_depth(bf_image_manager* im, uint32_t* w, uint32_t* h, float K[16]) {
    BF_REQUIRE(im, "null manager");
    if (w) *w = im->sensor.depthWidth;
    if (h) *h = im->sensor.depthHeight;
    if (K) memcpy(K, im->siftDepthIntrinsics.e, 64);
    return BF_OK;
}

}  // extern "C"

// ================================================================================================ Bundler (+ SBA)
struct bf_bundler {
    bf_global_app_state gas; bf_global_bundling_state gbs;
    uint32_t maxImages = 0, maxKeys = 0;
    bool isLocal = true;
    hipStream_t stream = nullptr;
    bf_sift* sift = nullptr; bf_siftmgr* mgr = nullptr; bf_cache* cache = nullptr; bf_solver* solver = nullptr;
    m44 siftIntrinsics, siftIntrinsicsInv;
    m44* d_trajectory = nullptr;
    float *d_xRot = nullptr, *d_xTrans = nullptr;
    int continueRetry = 0;
    uint32_t revalidatedIdx = 0xFFFFFFFFu;
    // SBA state (SBA.cpp:20-51)
    std::vector<float> localWS, localWD, localWC, globalWS, globalWD, globalWC;
    bool useGlobalDenseOpt = false, useLocalDense = true, comprehensive = false, sbaVerify = false;
    float maxResidual = -1.0f;
    uint32_t numSolves = 0;
    std::vector<int> validScratch;
    bf_correspondence_evaluator* corrEvaluator = nullptr;       // Bundler.h:101-103
};

namespace {

int bundlerValid(bf_bundler* b, std::vector<int>& v, uint32_t n) {
    v.assign(std::max<uint32_t>(n, 1), 0);
    return bf_siftmgr_get_valid_images(b->mgr, v.data(), n);
}

int cacheK(bf_bundler* b, uint32_t& w, uint32_t& h, float k4[4], m44& K) {
    BF_TRY(bf_cache_get_geometry(b->cache, &w, &h, k4));
    K = identity44();
    K.e[0] = k4[0]; K.e[5] = k4[1]; K.e[2] = k4[2]; K.e[6] = k4[3];
    return BF_OK;
}

// the asynchronous half of Bundler::matchAndFilter (:103-221): everything up to the read-back of the frame result
int evaluateStage(bf_bundler* b, bool filtered, bool recompute, bool clear, const char* type) {      // Bundler.cpp:145-147,164-166,181-183,202-204
    if (!b->corrEvaluator) return BF_OK;
    const bf_corr_eval_params p = {b->gbs.s_denseDepthMin, b->gbs.s_denseDepthMax, b->gbs.s_projCorrDistThres, b->gbs.s_projCorrNormalThres, b->gbs.s_projCorrColorThresh};
    return bf_correspondence_evaluator_evaluate(b->corrEvaluator, b->mgr, b->cache, b->siftIntrinsicsInv.e, &p, filtered, recompute, clear, type, b->stream, nullptr);
}

// ... in two stages.  The PAIR stage - match, Kabsch filter, surface-area filter, dense verification - computes, per previous image, from the two images
// alone; it runs on the stream and result set bf_siftmgr_set_pair_stage selected.  The COMMIT stage - which previous images count, filterFrames, the
// frame's EntryJ rows - needs the valid flags of all earlier frames and runs on the bundler's stream.
int matchAndFilterPairs(bf_bundler* b, uint32_t& curFrame, uint32_t& startFrame, uint32_t& numFrames) {
    BF_TRY(bf_siftmgr_get_num_images(b->mgr, &numFrames));
    BF_REQUIRE(numFrames > 1, "matchAndFilter needs more than one frame");
    BF_TRY(bf_siftmgr_get_current_frame(b->mgr, &curFrame));
    startFrame = numFrames == curFrame + 1 ? 0 : curFrame + 1;
    const float ratioMax = b->isLocal ? b->gbs.s_siftMatchRatioMaxLocal : b->gbs.s_siftMatchRatioMaxGlobal;
    BF_TRY(bf_siftmgr_update_gpu_valid_images(b->mgr));
    BF_TRY(bf_siftmgr_match(b->mgr, curFrame, startFrame, numFrames, b->gbs.s_siftMatchThresh, ratioMax));
    if (curFrame > 0) {
        const uint32_t minNumMatches = b->isLocal ? b->gbs.s_minNumMatchesLocal : b->gbs.s_minNumMatchesGlobal;
        BF_TRY(evaluateStage(b, false, true, false, "raw"));
        BF_TRY(bf_siftmgr_filter_keypoint_matches(b->mgr, curFrame, startFrame, numFrames, b->siftIntrinsicsInv.e, minNumMatches, b->gbs.s_maxKabschResidual2));
        BF_TRY(evaluateStage(b, true, false, false, "kabsch"));
        BF_TRY(bf_siftmgr_filter_matches_by_surface_area(b->mgr, curFrame, startFrame, numFrames, b->siftIntrinsicsInv.e, b->gbs.s_surfAreaPcaThresh));
        BF_TRY(evaluateStage(b, true, false, false, "sa"));
        uint32_t cw, ch; float k4[4]; m44 K;
        BF_TRY(cacheK(b, cw, ch, k4, K));
        const bf_cached_frame* d_frames = nullptr;
        BF_TRY(bf_cache_get_frames_gpu(b->cache, &d_frames));
        BF_TRY(bf_siftmgr_filter_matches_by_dense_verify(b->mgr, curFrame, startFrame, numFrames, cw, ch, K.e, d_frames, b->gbs.s_projCorrDistThres,
                                                         b->gbs.s_projCorrNormalThres, b->gbs.s_projCorrColorThresh, b->gbs.s_verifySiftErrThresh,
                                                         b->gbs.s_verifySiftCorrThresh, b->gas.s_sensorDepthMin, b->gas.s_sensorDepthMax));
        BF_TRY(evaluateStage(b, true, false, true, "dense"));
    }
    return BF_OK;
}
int matchAndFilterCommit(bf_bundler* b, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, bool speculative) {
    if (curFrame > 0) {
        if (speculative) BF_TRY(bf_siftmgr_commit_pairs(b->mgr, curFrame, startFrame, numFrames));
        BF_TRY(bf_siftmgr_filter_frames_async(b->mgr, curFrame, startFrame, numFrames));
        BF_TRY(bf_siftmgr_add_curr_to_residuals(b->mgr, curFrame, startFrame, numFrames, b->siftIntrinsicsInv.e));
    }
    return BF_OK;
}
int matchAndFilterEnqueue(bf_bundler* b, uint32_t& curFrame, uint32_t& startFrame, uint32_t& numFrames) {
    BF_TRY(matchAndFilterPairs(b, curFrame, startFrame, numFrames));
    return matchAndFilterCommit(b, curFrame, startFrame, numFrames, false);
}

int tryRevalidation(bf_bundler* b, uint32_t curGlobalFrame, bool bIsScanDone, uint32_t* out);

// the host half after the read-back (:223-248)
int matchAndFilterFinish(bf_bundler* b, uint32_t curFrame, uint32_t numFrames, uint32_t* lastMatchedFrame) {
    *lastMatchedFrame = 0xFFFFFFFFu;
    if (curFrame == 0) return BF_OK;
    uint32_t last = 0xFFFFFFFFu; int32_t numKeysCur = 0;
    BF_TRY(bf_siftmgr_sync_frame_result(b->mgr, curFrame, &last, &numKeysCur));
    if (numKeysCur < 0) { set_error("too many keypoints"); return BF_ERR_CAPACITY; }       // Bundler.cpp:98
    if (numKeysCur == 0) return BF_OK;                                                     // :115 (nothing was matched; the frame stays invalid)
    *lastMatchedFrame = last;
    if (!b->isLocal) {
        if (last != 0xFFFFFFFFu && last + 1 != curFrame) {                                  // re-initialise from the last match (:224-227)
            BF_HIP_TRY(hipMemcpyAsync(b->d_trajectory + curFrame, b->d_trajectory + last, sizeof(m44), hipMemcpyDeviceToDevice, b->stream));
            if (curFrame + 1 < b->maxImages) BF_HIP_TRY(hipMemcpyAsync(b->d_trajectory + curFrame + 1, b->d_trajectory + last, sizeof(m44), hipMemcpyDeviceToDevice, b->stream));
        }
        if (curFrame + 1 == numFrames) {
            if (last != 0xFFFFFFFFu) { uint32_t r; BF_TRY(tryRevalidation(b, curFrame, false, &r)); }
            else BF_TRY(bf_siftmgr_add_to_retry_list(b->mgr, curFrame));
        }
    }
    return BF_OK;
}

int matchAndFilter(bf_bundler* b, uint32_t* lastMatchedFrame) {
    uint32_t cur, start, num;
    BF_TRY(matchAndFilterEnqueue(b, cur, start, num));
    return matchAndFilterFinish(b, cur, num, lastMatchedFrame);
}

int tryRevalidation(bf_bundler* b, uint32_t curGlobalFrame, bool bIsScanDone, uint32_t* out) {       // Bundler.cpp:306-352 (USE_RETRY)
    b->revalidatedIdx = 0xFFFFFFFFu;
    *out = 0xFFFFFFFFu;
    if (b->continueRetry < 0) { *out = 0; return BF_OK; }            // "return false" from a function returning unsigned
    uint32_t idx; int found = 0;
    BF_TRY(bf_siftmgr_get_top_retry_image(b->mgr, &idx, &found));
    if (found) {
        if (bIsScanDone) {
            if (b->continueRetry == 0) b->continueRetry = (int)idx;
            else if (b->continueRetry == (int)idx) { b->continueRetry = -1; return BF_OK; }
        }
        BF_TRY(bf_siftmgr_set_current_frame(b->mgr, idx));
        uint32_t lastMatchedGlobal;
        BF_TRY(matchAndFilter(b, &lastMatchedGlobal));
        std::vector<int> v;
        uint32_t n; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &n));
        BF_TRY(bundlerValid(b, v, n));
        if (v[idx] != 0) {
            BF_REQUIRE(lastMatchedGlobal != 0xFFFFFFFFu, "revalidated frame without a match");
            BF_HIP_TRY(hipMemcpyAsync(b->d_trajectory + idx, b->d_trajectory + lastMatchedGlobal, sizeof(m44), hipMemcpyDeviceToDevice, b->stream));
            b->revalidatedIdx = idx;
        } else BF_TRY(bf_siftmgr_add_to_retry_list(b->mgr, idx));
        BF_TRY(bf_siftmgr_set_current_frame(b->mgr, curGlobalFrame));
    }
    *out = b->revalidatedIdx;
    return BF_OK;
}

// SBA::align + alignCUDA + removeMaxResidualCUDA (SBA.cpp:53-204)
int sbaAlign(bf_bundler* b, uint32_t maxNumIters, uint32_t numPCGits, bool useVerify, bool isEnd, uint32_t revalidateIdx, bool* removed) {
    *removed = false;
    b->sbaVerify = false; b->maxResidual = -1.0f;
    const std::vector<float>*wS, *wD, *wC;
    std::vector<float> zeros;
    bool useCache = true;
    if (b->isLocal) {
        wS = &b->localWS;
        if (b->useLocalDense) { wD = &b->localWD; wC = &b->localWC; }
        else { useCache = false; zeros.assign(b->localWD.size(), 0.0f); wD = wC = &zeros; }
    } else {
        wS = &b->globalWS;
        if (!b->useGlobalDenseOpt) { useCache = false; zeros.assign(b->globalWD.size(), 0.0f); wD = wC = &zeros; }
        else { wD = &b->globalWD; wC = &b->globalWC; }
    }
    uint32_t numImages, numCorr;
    BF_TRY(bf_siftmgr_get_num_images(b->mgr, &numImages));
    BF_TRY(bf_siftmgr_get_num_global_correspondences(b->mgr, &numCorr));
    const int32_t* d_valid = nullptr; bf_entry_j* d_corr = nullptr;
    BF_TRY(bf_siftmgr_get_valid_images_gpu(b->mgr, &d_valid));
    BF_TRY(bf_siftmgr_get_global_correspondences_gpu(b->mgr, &d_corr));
    BF_TRY(bf_convert_matrices_to_poses((const float*)b->d_trajectory, numImages, b->d_xRot, b->d_xTrans, d_valid, b->stream));
    uint32_t cw = 0, ch = 0; float k4[4] = {0, 0, 0, 0}; m44 K;
    const bf_cached_frame* d_frames = nullptr;
    if (useCache) { BF_TRY(cacheK(b, cw, ch, k4, K)); BF_TRY(bf_cache_get_frames_gpu(b->cache, &d_frames)); }
    BF_TRY(bf_solver_solve(b->solver, d_corr, numCorr, d_valid, numImages, maxNumIters, numPCGits, d_frames, cw, ch, k4, wS->data(), wD->data(), wC->data(),
                           (uint32_t)wS->size(), 1, b->d_xRot, b->d_xTrans, 1, isEnd ? 1 : 0, revalidateIdx));
    b->numSolves++;
    if (isEnd && wS->front() > 0) {                                            // removeMaxResidualCUDA :168-204
        uint32_t curFrame;
        BF_TRY(bf_siftmgr_get_current_frame(b->mgr, &curFrame));
        if (revalidateIdx != 0xFFFFFFFFu) curFrame = revalidateIdx;
        uint32_t pair[2]; int remove = 0;
        BF_TRY(bf_solver_get_max_residual_pair(b->solver, curFrame, d_corr, pair, &b->maxResidual, &remove));
        if (remove) {
            BF_TRY(bf_siftmgr_invalidate_image_to_image(b->mgr, pair[0], pair[1]));
            const int32_t* d_rows = nullptr;
            BF_TRY(bf_solver_get_var_to_corr_num_entries_per_row(b->solver, &d_rows));
            if (b->comprehensive) BF_TRY(bf_siftmgr_check_for_invalid_frames(b->mgr, d_rows, numImages));
            else BF_TRY(bf_siftmgr_check_for_invalid_frames_simple(b->mgr, d_rows, numImages));
            *removed = true;
        }
    }
    if (useVerify) {
        // SBA.cpp:106-109.  Without correspondences CUDASolverBundling::useVerification (:454-476) evaluates 0 / 0 >= thresh, which
        // is false: no trajectory verification is requested then.
        if (wS->front() > 0) { int v = 0; if (numCorr > 0) BF_TRY(bf_solver_use_verification(b->solver, d_corr, numCorr, &v)); b->sbaVerify = v != 0; }
        else b->sbaVerify = true;
    }
    BF_TRY(bf_convert_poses_to_matrices(b->d_xRot, b->d_xTrans, numImages, (float*)b->d_trajectory, d_valid, b->stream));
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_bundler_create(uint32_t maxNumImages, uint32_t maxNumKeysPerImage, const float siftIntrinsicsInv[16], bf_image_manager* manager, int isLocal,
                      const bf_global_app_state* gas, const bf_global_bundling_state* gbs, bf_bundler** out) {
    BF_REQUIRE(siftIntrinsicsInv && manager && gas && gbs && out && maxNumImages >= 2, "bad argument");
    bf_bundler* b = new bf_bundler;
    b->gas = *gas; b->gbs = *gbs; b->maxImages = maxNumImages; b->maxKeys = maxNumKeysPerImage; b->isLocal = isLocal != 0;
    b->siftIntrinsicsInv = toM(siftIntrinsicsInv);
    b->siftIntrinsics = inverse44(b->siftIntrinsicsInv);
    int rc = BF_OK;
    const bf_rgbd_sensor_desc& sn = manager->sensor;
    if (b->isLocal)              // initSift: SetParams(w, h, false, 150, depthMin, depthMax) (:57-69)
        rc = bf_sift_create(gbs->s_widthSIFT, gbs->s_heightSIFT, sn.depthWidth, sn.depthHeight, 150, gas->s_sensorDepthMin, gas->s_sensorDepthMax, gbs->s_minKeyScale,
                            maxNumKeysPerImage, &b->sift);
    const uint32_t maxNumResiduals = 25 * (maxNumImages * (maxNumImages - 1)) / 2;
    bf_solver_config cfg;
    cfg.optMaxResThresh = gbs->s_optMaxResThresh; cfg.denseDistThresh = gbs->s_denseDistThresh; cfg.denseNormalThresh = gbs->s_denseNormalThresh;
    cfg.denseColorThresh = gbs->s_denseColorThresh; cfg.denseColorGradientMin = gbs->s_denseColorGradientMin; cfg.denseDepthMin = gbs->s_denseDepthMin;
    cfg.denseDepthMax = gbs->s_denseDepthMax; cfg.denseOverlapCheckSubsampleFactor = gbs->s_denseOverlapCheckSubsampleFactor;
    cfg.verifyOptDistThresh = 0.02f; cfg.verifyOptPercentThresh = 0.05f; cfg.recordConvergence = gbs->s_recordSolverConvergence;
    if (!rc) rc = bf_solver_create(maxNumImages, maxNumResiduals, &cfg, &b->solver);
    if (!rc) rc = bf_cache_create(sn.depthWidth, sn.depthHeight, gbs->s_downsampledWidth, gbs->s_downsampledHeight, maxNumImages, manager->siftDepthIntrinsics.e,
                                  gbs->s_colorDownSigma, gbs->s_depthDownSigmaD, gbs->s_depthDownSigmaR, &b->cache);
    if (!rc) rc = bf_siftmgr_create(maxNumImages, maxNumKeysPerImage, &b->mgr);
    if (rc) { bf_bundler_destroy(b); return rc; }
    BF_HIP_TRY(BF_MALLOC((void**)&b->d_trajectory, sizeof(m44) * (maxNumImages + 1)));
    BF_HIP_TRY(BF_MALLOC((void**)&b->d_xRot, sizeof(float) * 3 * maxNumImages));
    BF_HIP_TRY(BF_MALLOC((void**)&b->d_xTrans, sizeof(float) * 3 * maxNumImages));
    k_fill_identity<<<div_up(maxNumImages + 1, 64), 64>>>(b->d_trajectory, maxNumImages + 1);
    BF_HIP_TRY(hipDeviceSynchronize());
    const uint32_t maxNumIts = std::max(gbs->s_numGlobalNonLinIterations, gbs->s_numLocalNonLinIterations);     // SBA.cpp:28-38
    b->localWS.assign(maxNumIts, 1.0f); b->localWD.resize(maxNumIts); b->localWC.assign(maxNumIts, 0.0f);
    for (uint32_t i = 0; i < maxNumIts; ++i) b->localWD[i] = (float)i + 1.0f;
    b->globalWS.assign(maxNumIts, 1.0f); b->globalWD.assign(maxNumIts, 1.0f); b->globalWC.assign(maxNumIts, 0.1f);
    for (uint32_t i = 2; i < maxNumIts; ++i) b->globalWD[i] = (float)i;
    b->useGlobalDenseOpt = false;
    b->comprehensive = gbs->s_useComprehensiveFrameInvalidation != 0;          // SBA.h:31-32
    b->useLocalDense = gbs->s_useLocalDense != 0;
    *out = b;
    return BF_OK;
}

int bf_bundler_destroy(bf_bundler* b) {
    if (!b) return BF_OK;
    bf_sift_destroy(b->sift); bf_solver_destroy(b->solver); bf_cache_destroy(b->cache); bf_siftmgr_destroy(b->mgr);
    bf_correspondence_evaluator_destroy(b->corrEvaluator);
    (void)hipFree(b->d_trajectory); (void)hipFree(b->d_xRot); (void)hipFree(b->d_xTrans);
    delete b;
    return BF_OK;
}

int bf_bundler_set_stream(bf_bundler* b, void* s) {
    BF_REQUIRE(b, "null bundler");
    b->stream = (hipStream_t)s;
    if (b->sift) BF_TRY(bf_sift_set_stream(b->sift, s));
    BF_TRY(bf_solver_set_stream(b->solver, s)); BF_TRY(bf_cache_set_stream(b->cache, s)); BF_TRY(bf_siftmgr_set_stream(b->mgr, s));
    return BF_OK;
}

int bf_bundler_get_trajectory_gpu(bf_bundler* b, float** d) { BF_REQUIRE(b && d, "null argument"); *d = (float*)b->d_trajectory; return BF_OK; }
int bf_bundler_get_valid_images(bf_bundler* b, int32_t* h_out, uint32_t count) { BF_REQUIRE(b, "null bundler"); return bf_siftmgr_get_valid_images(b->mgr, h_out, count); }
int bf_bundler_get_cache_intrinsics(bf_bundler* b, float K[16], float Kinv[16]) {
    BF_REQUIRE(b, "null bundler");
    uint32_t w, h; float k4[4]; m44 M;
    BF_TRY(cacheK(b, w, h, k4, M));
    if (K) memcpy(K, M.e, 64);
    if (Kinv) { const m44 I = inverse44(M); memcpy(Kinv, I.e, 64); }
    return BF_OK;
}
int bf_bundler_get_curr_frame_number(bf_bundler* b, uint32_t* out) { BF_REQUIRE(b, "null bundler"); return bf_siftmgr_get_current_frame(b->mgr, out); }
int bf_bundler_get_num_frames(bf_bundler* b, uint32_t* out) { BF_REQUIRE(b, "null bundler"); return bf_siftmgr_get_num_images(b->mgr, out); }

int bf_bundler_is_valid(bf_bundler* b, int* out) {
    BF_REQUIRE(b && out, "null argument");
    uint32_t n; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &n));
    std::vector<int> v; BF_TRY(bundlerValid(b, v, n));
    *out = 0;
    for (uint32_t i = 1; i < n; ++i) if (v[i] != 0) { *out = 1; break; }
    return BF_OK;
}

int bf_bundler_reset(bf_bundler* b) {
    BF_REQUIRE(b, "null bundler");
    uint32_t n; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &n));
    if (n) k_fill_identity<<<div_up(n, 64), 64, 0, b->stream>>>(b->d_trajectory, n);
    BF_TRY(bf_siftmgr_reset(b->mgr));
    BF_TRY(bf_cache_reset(b->cache));
    return BF_OK;
}

int bf_bundler_detect_features(bf_bundler* b, const float* d_intensitySift, const float* d_inputDepthFilt) {
    BF_REQUIRE(b && b->sift && d_intensitySift && d_inputDepthFilt, "detectFeatures needs a local bundler");
    bf_sift_image_gpu img;
    BF_TRY(bf_siftmgr_create_image(b->mgr, &img));
    BF_TRY(bf_sift_run(b->sift, d_intensitySift, d_inputDepthFilt, (float*)img.d_keyPoints, (uint8_t*)img.d_keyPointDescs, img.d_numKeyPoints));
    return bf_siftmgr_finalize_image(b->mgr, -1);       // the count stays on the device; "too many keypoints" surfaces at the frame read-back
}

int bf_bundler_store_cached_frame(bf_bundler* b, uint32_t dw, uint32_t dh, const uint8_t* d_color, uint32_t cw, uint32_t ch, const float* d_depthRaw) {
    BF_REQUIRE(b, "null bundler");
    return bf_cache_store_frame(b->cache, d_depthRaw, dw, dh, d_color, cw, ch);
}

int bf_bundler_copy_frame(bf_bundler* b, bf_bundler* from, uint32_t frame) {
    BF_REQUIRE(b && from, "null bundler");
    bf_sift_image_gpu next, cur;
    BF_TRY(bf_siftmgr_get_image(from->mgr, frame, &cur));
    BF_TRY(bf_siftmgr_create_image(b->mgr, &next));
    const uint32_t mk = std::min(b->maxKeys, from->maxKeys);
    BF_HIP_TRY(hipMemcpyAsync(next.d_keyPoints, cur.d_keyPoints, sizeof(bf_sift_keypoint) * mk, hipMemcpyDeviceToDevice, b->stream));
    BF_HIP_TRY(hipMemcpyAsync(next.d_keyPointDescs, cur.d_keyPointDescs, sizeof(bf_sift_keypoint_desc) * mk, hipMemcpyDeviceToDevice, b->stream));
    BF_HIP_TRY(hipMemcpyAsync(next.d_numKeyPoints, cur.d_numKeyPoints, sizeof(int32_t), hipMemcpyDeviceToDevice, b->stream));
    BF_TRY(bf_siftmgr_finalize_image(b->mgr, -1));
    return bf_cache_copy_cache_frame_from(b->cache, from->cache, frame);
}

int bf_bundler_add_invalid_frame(bf_bundler* b) {
    BF_REQUIRE(b, "null bundler");
    BF_TRY(bf_cache_increment(b->cache));
    bf_sift_image_gpu img;
    BF_TRY(bf_siftmgr_create_image(b->mgr, &img));
    BF_TRY(bf_siftmgr_finalize_image(b->mgr, 0));
    uint32_t n; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &n));        // initializeNextTransformUnknown (Bundler.h:78-82)
    BF_HIP_TRY(hipMemcpyAsync(b->d_trajectory + n, b->d_trajectory + n - 1, sizeof(m44), hipMemcpyDeviceToDevice, b->stream));
    return BF_OK;
}

int bf_bundler_invalidate_last_frame(bf_bundler* b) {
    BF_REQUIRE(b, "null bundler");
    uint32_t n; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &n));
    if (n <= 1) { set_error("INVALID_FIRST_CHUNK"); return BF_ERR_STATE; }      // the reference writes processed.txt and exits (:377-384)
    return bf_siftmgr_set_valid_image(b->mgr, n - 1, 0);
}

int bf_bundler_get_current_sift_transforms_gpu(bf_bundler* b, const float** d_out) { BF_REQUIRE(b && d_out, "null argument"); return bf_siftmgr_get_filt_transforms_gpu(b->mgr, nullptr, d_out); }
int bf_bundler_get_num_filt_matches_gpu(bf_bundler* b, const int32_t** d_out) { BF_REQUIRE(b && d_out, "null argument"); return bf_siftmgr_get_num_filt_matches_gpu(b->mgr, d_out); }

int bf_bundler_match_and_filter(bf_bundler* b, uint32_t* lastMatchedFrame) {
    BF_REQUIRE(b && lastMatchedFrame, "null argument");
    return matchAndFilter(b, lastMatchedFrame);
}

int bf_bundler_optimize(bf_bundler* b, uint32_t numNonLin, uint32_t numLin, int bUseVerify, int bRemoveMaxResidual, int bIsScanDone, int* bOptRemoved, int* valid) {
    (void)bIsScanDone;
    BF_REQUIRE(b && bOptRemoved && valid, "null argument");
    uint32_t numImages; BF_TRY(bf_siftmgr_get_num_images(b->mgr, &numImages));
    BF_REQUIRE(numImages > 1, "optimize needs more than one image");
    bool removed = false;
    BF_TRY(sbaAlign(b, numNonLin, numLin, bUseVerify != 0, bRemoveMaxResidual != 0, b->revalidatedIdx, &removed));
    *bOptRemoved = removed ? 1 : 0;
    *valid = 1;
    if (b->sbaVerify) {                                                        // Bundler.cpp:259-273
        uint32_t cw, ch; float k4[4]; m44 K;
        BF_TRY(cacheK(b, cw, ch, k4, K));
        const bf_cached_frame* d_frames = nullptr;
        BF_TRY(bf_cache_get_frames_gpu(b->cache, &d_frames));
        int32_t v = 0;
        BF_TRY(bf_siftmgr_verify_trajectory(b->mgr, numImages, (const float*)b->d_trajectory, cw, ch, K.e, d_frames, b->gbs.s_projCorrDistThres,
                                            b->gbs.s_projCorrNormalThres, b->gbs.s_projCorrColorThresh, b->gbs.s_verifyOptErrThresh, b->gbs.s_verifyOptCorrThresh,
                                            0.1f, 3.0f, &v));
        *valid = v > 0 ? 1 : 0;
    }
    return BF_OK;
}

int bf_bundler_set_solve_weights(bf_bundler* b, const float* sparse, const float* denseDepth, const float* denseColor, uint32_t n) {
    BF_REQUIRE(b && sparse && denseDepth && denseColor && n > 0, "bad argument");
    b->globalWS.assign(sparse, sparse + n); b->globalWD.assign(denseDepth, denseDepth + n); b->globalWC.assign(denseColor, denseColor + n);
    b->useGlobalDenseOpt = denseDepth[n - 1] > 0 || denseColor[n - 1] > 0;      // Bundler.h:51-54
    return BF_OK;
}

int bf_bundler_fuse_to_global(bf_bundler* b, bf_bundler* glob) {
    BF_REQUIRE(b && glob, "null bundler");
    BF_TRY(bf_siftmgr_fuse_to_global(b->mgr, glob->mgr, b->siftIntrinsics.e, (const float*)b->d_trajectory, b->siftIntrinsicsInv.e));
    return bf_cache_copy_cache_frame_from(glob->cache, b->cache, 0);
}

int bf_bundler_try_revalidation(bf_bundler* b, uint32_t curGlobalFrame, int bIsScanDone, uint32_t* revalidatedIdx) {
    BF_REQUIRE(b && revalidatedIdx, "null argument");
    return tryRevalidation(b, curGlobalFrame, bIsScanDone != 0, revalidatedIdx);
}
int bf_bundler_get_revalidated_idx(bf_bundler* b, uint32_t* out) { BF_REQUIRE(b && out, "null argument"); *out = b->revalidatedIdx; return BF_OK; }

int bf_bundler_save_sparse_corrs_to_file(bf_bundler* b, const char* filename) {       // UINT64 count + EntryJ[count]
    BF_REQUIRE(b && filename, "null argument");
    uint32_t n; BF_TRY(bf_siftmgr_get_num_global_correspondences(b->mgr, &n));
    if (n == 0) return BF_OK;                                                    // "warning: no sparse correspondences to save"
    bf_entry_j* d_corr = nullptr;
    BF_TRY(bf_siftmgr_get_global_correspondences_gpu(b->mgr, &d_corr));
    std::vector<bf_entry_j> corr(n);
    BF_HIP_TRY(hipMemcpyAsync(corr.data(), d_corr, sizeof(bf_entry_j) * n, hipMemcpyDeviceToHost, b->stream));
    BF_HIP_TRY(hipStreamSynchronize(b->stream));
    std::ofstream out(filename, std::ios::binary);
    if (!out.is_open()) { set_error("cannot open %s", filename); return BF_ERR_INVALID_ARG; }
    const uint64_t cnt = n;
    out.write((const char*)&cnt, sizeof cnt);
    out.write((const char*)corr.data(), sizeof(bf_entry_j) * n);
    return BF_OK;
}

int bf_bundler_initialize_correspondence_evaluator(bf_bundler* b, const float* h_trajectory, uint32_t numTransforms, const char* logFilePrefix) {
    BF_REQUIRE(b, "null bundler");
    bf_correspondence_evaluator_destroy(b->corrEvaluator);
    b->corrEvaluator = nullptr;
    return bf_correspondence_evaluator_create(h_trajectory, numTransforms, logFilePrefix, &b->corrEvaluator);
}
int bf_bundler_finish_correspondence_evaluator_logging(bf_bundler* b) {
    BF_REQUIRE(b, "null bundler");
    return b->corrEvaluator ? bf_correspondence_evaluator_finish_logging_to_file(b->corrEvaluator) : BF_OK;
}
int bf_bundler_get_correspondence_evaluator(bf_bundler* b, bf_correspondence_evaluator** out) { BF_REQUIRE(b && out, "null argument"); *out = b->corrEvaluator; return BF_OK; }
int bf_bundler_get_sift_manager(bf_bundler* b, bf_siftmgr** out) { BF_REQUIRE(b && out, "null argument"); *out = b->mgr; return BF_OK; }
int bf_bundler_get_cache(bf_bundler* b, bf_cache** out) { BF_REQUIRE(b && out, "null argument"); *out = b->cache; return BF_OK; }
int bf_bundler_get_solver(bf_bundler* b, bf_solver** out) { BF_REQUIRE(b && out, "null argument"); *out = b->solver; return BF_OK; }

}  // extern "C"

// ================================================================================================ TrajectoryManager
struct bf_trajectory_manager {
    struct Frame {
        int type; uint32_t frameIdx; m44 integratedTransform; float dist;
        // se(3) coordinates of integratedTransform / of the optimised pose the last time they were needed, and of which matrices
        // (generate_update_lists ranks ALL frames every time the lists run low: at 5000 frames the two logarithms per frame were
        // 1.5 ms per call, 0.8 ms per input frame - the term that made the long stream decay, profiles/r03_5000_frame_stream.md)
        m44 intOf, optOf; f3 ri, ti, ro, to; bool haveInt = false, haveOpt = false;
    };
    struct Cand { Frame* f; uint32_t pos; };
    std::vector<char> picked;            // scratch of generate_update_lists
    std::vector<Cand> candScratch; std::vector<Frame*> restScratch;
    std::vector<m44> optimizedTransforms;
    std::vector<Frame> frames;
    std::vector<Frame*> framesSort;
    uint32_t numAddedFrames = 0, numOptimizedFrames = 0;
    std::list<Frame*> toDeIntegrate, toIntegrate, toReIntegrate;
    uint32_t topNActive = 30;
    float minPoseDistSqrt = 0.0f, featureRescaleRotToTrans = 2.0f;
    m44& opt(const Frame& f) { return optimizedTransforms[f.frameIdx]; }
};

namespace {
void tmInvalidate(bf_trajectory_manager* tm, uint32_t idx) {                       // TrajectoryManager.cpp:190-199
    auto& f = tm->frames[idx];
    if (f.type == BF_TF_INVALID) return;
    const int before = f.type;
    f.type = BF_TF_INVALID;
    if (before == BF_TF_INTEGRATED) tm->toDeIntegrate.push_back(&f);
    // never integrated and no pose any more: nothing to integrate.  (The reference leaves the frame in m_toIntegrateList, and its
    // getTopFromIntegrateList then stops at "ERROR NEED TO CHECK FOR INVALIDATE INTEGRATE LIST ELEMENTS" + assert, :142-146.)
    else if (before == BF_TF_NOT_INTEGRATED_WITH_TRANSFORM) tm->toIntegrate.remove(&f);
}
}  // namespace

extern "C" {

int bf_trajectory_manager_create(uint32_t numMaxImage, uint32_t topNActive, float minPoseDistSqrt, bf_trajectory_manager** out) {
    BF_REQUIRE(out && numMaxImage > 0, "bad argument");
    bf_trajectory_manager* tm = new bf_trajectory_manager;
    tm->optimizedTransforms.assign(numMaxImage, minfM());
    tm->frames.resize(numMaxImage);
    for (uint32_t i = 0; i < numMaxImage; ++i) { tm->frames[i].type = BF_TF_NOT_INTEGRATED_NO_TRANSFORM; tm->frames[i].frameIdx = i; tm->frames[i].integratedTransform = minfM(); tm->frames[i].dist = 0.0f; }
    tm->framesSort.reserve(numMaxImage);
    tm->topNActive = topNActive; tm->minPoseDistSqrt = minPoseDistSqrt;
    *out = tm;
    return BF_OK;
}
int bf_trajectory_manager_destroy(bf_trajectory_manager* tm) { delete tm; return BF_OK; }

int bf_trajectory_manager_add_frame(bf_trajectory_manager* tm, int type, const float transform[16], uint32_t idx) {
    BF_REQUIRE(tm && transform && idx < tm->frames.size(), "frame out of range");
    auto& f = tm->frames[idx];
    f.type = type; f.frameIdx = idx; f.integratedTransform = toM(transform);
    tm->optimizedTransforms[idx] = f.integratedTransform;
    tm->framesSort.push_back(&f);
    tm->numAddedFrames++;
    return BF_OK;
}

int bf_trajectory_manager_update_optimized_transform(bf_trajectory_manager* tm, const float* d_trajectory, uint32_t numFrames, void* stream) {
    BF_REQUIRE(tm && (d_trajectory || numFrames == 0), "null argument");
    tm->numOptimizedFrames = numFrames;
    numFrames = std::min(numFrames, tm->numAddedFrames);
    if (numFrames) {
        BF_HIP_TRY(hipMemcpyAsync(tm->optimizedTransforms.data(), d_trajectory, sizeof(m44) * numFrames, hipMemcpyDeviceToHost, (hipStream_t)stream));
        BF_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return BF_OK;
}

// the same with the trajectory already on the host (updateOptimizedTransform takes a device pointer because the bundler's
// trajectory lives there, TrajectoryManager.cpp:36-45)
int bf_trajectory_manager_update_optimized_transform_host(bf_trajectory_manager* tm, const float* h_trajectory, uint32_t numFrames) {
    BF_REQUIRE(tm && (h_trajectory || numFrames == 0), "null argument");
    tm->numOptimizedFrames = numFrames;
    numFrames = std::min(numFrames, tm->numAddedFrames);
    if (numFrames) memcpy(tm->optimizedTransforms.data(), h_trajectory, sizeof(m44) * numFrames);
    return BF_OK;
}

int bf_trajectory_manager_generate_update_lists(bf_trajectory_manager* tm) {        // :47-111
    BF_REQUIRE(tm, "null manager");
    const uint32_t numFrames = std::min(tm->numOptimizedFrames, tm->numAddedFrames);
    for (uint32_t i = 0; i < numFrames; ++i) {
        auto& f = tm->frames[i];
        const m44& T = tm->opt(f);
        if (T.e[0] == NINF) tmInvalidate(tm, i);
        else {
            if (f.type == BF_TF_NOT_INTEGRATED_NO_TRANSFORM || f.type == BF_TF_INVALID) { f.type = BF_TF_NOT_INTEGRATED_WITH_TRANSFORM; tm->toIntegrate.push_back(&f); }
            // the two logarithms are cached per frame, keyed by the matrix bits they were taken of: same values, computed once per pose change
            if (!f.haveOpt || memcmp(f.optOf.e, T.e, 64) != 0) { matrixToPose(T, f.ro, f.to); f.optOf = T; f.haveOpt = true; }
            if (!f.haveInt || memcmp(f.intOf.e, f.integratedTransform.e, 64) != 0) { matrixToPose(f.integratedTransform, f.ri, f.ti); f.intOf = f.integratedTransform; f.haveInt = true; }
            const f3 ro = f.ro, to = f.to, ri = f.ri, ti = f.ti;
            // PoseHelper::MatrixToPose (USE_LIE_SPACE, PoseHelper.h:332-363) returns (translation part, rotation vector), and the
            // reference rescales elements 0..2 (:70-77): it is the TRANSLATION part that is doubled, whatever the member's name says.
            // Pinned against TrajectoryManager.cpp itself (tests/test_ref_pin_cpu.py).
            const float s = tm->featureRescaleRotToTrans;
            const float d[6] = {ti.x * s - to.x * s, ti.y * s - to.y * s, ti.z * s - to.z * s, ri.x - ro.x, ri.y - ro.y, ri.z - ro.z};
            f.dist = 0.0f;
            for (int k = 0; k < 6; ++k) f.dist += d[k] * d[k];
        }
    }
    // The reference std::sorts all frames (Integrated first, by descending distance) and then reads positions < topNActive only.  Here: the
    // first topNActive positions of the STABLE sort of the current order (ties: earlier position first) by partial selection, the other
    // frames keep their relative order - O(F log topN) instead of O(F log F) per call.  Which frames are selected is identical: a selected
    // frame has type Integrated and a distance > m_minPoseDistSqrt >= 0, and among those the order is strict except for bit-equal
    // distances of different frames, which the position decides like the stable sort of the full array would.
    {
        typedef bf_trajectory_manager::Cand Cand;
        std::vector<Cand>& c = tm->candScratch;            // owned by the manager: no allocation per call
        c.resize(numFrames);
        for (uint32_t i = 0; i < numFrames; ++i) c[i] = {tm->framesSort[i], i};
        const uint32_t K = std::min<uint32_t>(tm->topNActive, numFrames);
        auto before = [](const Cand& l, const Cand& r) {
            const bool li = l.f->type == BF_TF_INTEGRATED, ri = r.f->type == BF_TF_INTEGRATED;
            if (li != ri) return li;
            if (li && l.f->dist != r.f->dist) return l.f->dist > r.f->dist;
            if (li) return l.f->frameIdx < r.f->frameIdx;      // bit-equal distances of two Integrated frames: by frame index, not by a position that depends on the history of earlier calls
            return l.pos < r.pos;
        };
        std::partial_sort(c.begin(), c.begin() + K, c.end(), before);
        tm->picked.assign(numFrames, 0);
        for (uint32_t i = 0; i < K; ++i) tm->picked[c[i].pos] = 1;
        std::vector<bf_trajectory_manager::Frame*>& rest = tm->restScratch;
        rest.clear();
        rest.reserve(numFrames - K);
        for (uint32_t i = 0; i < numFrames; ++i) if (!tm->picked[i]) rest.push_back(tm->framesSort[i]);
        for (uint32_t i = 0; i < K; ++i) tm->framesSort[i] = c[i].f;
        std::copy(rest.begin(), rest.end(), tm->framesSort.begin() + K);
    }
    for (uint32_t i = (uint32_t)tm->toReIntegrate.size(); i < tm->topNActive && i < numFrames; ++i) {
        auto* f = tm->framesSort[i];
        if (f->dist > tm->minPoseDistSqrt && f->type == BF_TF_INTEGRATED) { f->type = BF_TF_REINTEGRATION; tm->toReIntegrate.push_back(f); }
        else break;
    }
    return BF_OK;
}

int bf_trajectory_manager_confirm_integration(bf_trajectory_manager* tm, uint32_t frameIdx) {
    BF_REQUIRE(tm && frameIdx < tm->frames.size(), "frame out of range");
    tm->frames[frameIdx].type = BF_TF_INTEGRATED;
    return BF_OK;
}

int bf_trajectory_manager_get_top_from_reintegrate_list(bf_trajectory_manager* tm, float oldT[16], float newT[16], uint32_t* frameIdx, int* found) {
    BF_REQUIRE(tm && oldT && newT && frameIdx && found, "null argument");
    *found = 0;
    if (tm->toReIntegrate.empty()) return BF_OK;
    while (!tm->toReIntegrate.empty()) {                     // some may have been invalidated in the meantime (:123-134)
        auto* f = tm->toReIntegrate.front();
        const m44 nT = tm->opt(*f);
        memcpy(newT, nT.e, 64); memcpy(oldT, f->integratedTransform.e, 64);
        *frameIdx = f->frameIdx;
        tm->toReIntegrate.pop_front();
        if (nT.e[0] != NINF) { f->integratedTransform = nT; break; }
        // Invalidated while queued.  The reference leaves the frame typed ReIntegration here and promises that it "will be added
        // to the deintegrate list next time" (:133), but invalidateFrame (:191-199) queues a de-integration only for frames typed
        // Integrated, so the geometry at the old pose would stay in the volume for good (and be integrated a second time when
        // the frame becomes valid again).  The frame IS still integrated at its old pose: say so, and the next list update
        // de-integrates it exactly once.  (Deviation from the reference, DESIGN.md "Deviations".)
        f->type = BF_TF_INTEGRATED;
    }
    *found = 1;
    return BF_OK;
}

int bf_trajectory_manager_get_top_from_integrate_list(bf_trajectory_manager* tm, float trans[16], uint32_t* frameIdx, int* found) {
    BF_REQUIRE(tm && trans && frameIdx && found, "null argument");
    *found = 0;
    if (tm->toIntegrate.empty()) return BF_OK;
    while (!tm->toIntegrate.empty()) {
        auto* f = tm->toIntegrate.front();
        BF_REQUIRE(f->type == BF_TF_NOT_INTEGRATED_WITH_TRANSFORM, "integrate list holds an invalidated frame");
        const m44 T = tm->opt(*f);
        tm->toIntegrate.pop_front();
        // the optimisation result that arrived after the list was made took the pose away again: not integrated, Invalid until a
        // pose comes back.  (The reference hands the -inf pose to its consumer, which asserts: DepthSensing.cpp:881.  Integrating it
        // would also leave a NaN in the ranking distance of an Integrated frame, i.e. an unspecified sort order.)
        if (T.e[0] == NINF) { f->type = BF_TF_INVALID; continue; }
        memcpy(trans, T.e, 64);
        *frameIdx = f->frameIdx;
        f->integratedTransform = T;
        *found = 1;
        break;
    }
    return BF_OK;
}

int bf_trajectory_manager_get_top_from_deintegrate_list(bf_trajectory_manager* tm, float trans[16], uint32_t* frameIdx, int* found) {
    BF_REQUIRE(tm && trans && frameIdx && found, "null argument");
    *found = 0;
    if (tm->toDeIntegrate.empty()) return BF_OK;
    auto* f = tm->toDeIntegrate.front();
    memcpy(trans, f->integratedTransform.e, 64);
    *frameIdx = f->frameIdx;
    tm->toDeIntegrate.pop_front();
    *found = 1;
    return BF_OK;
}

int bf_trajectory_manager_get_num_optimized_frames(bf_trajectory_manager* tm, uint32_t* out) { BF_REQUIRE(tm && out, "null argument"); *out = tm->numOptimizedFrames; return BF_OK; }
int bf_trajectory_manager_get_num_added_frames(bf_trajectory_manager* tm, uint32_t* out) { BF_REQUIRE(tm && out, "null argument"); *out = tm->numAddedFrames; return BF_OK; }
int bf_trajectory_manager_get_num_active_operations(bf_trajectory_manager* tm, uint32_t* out) {
    BF_REQUIRE(tm && out, "null argument");
    *out = (uint32_t)(tm->toDeIntegrate.size() + tm->toIntegrate.size() + tm->toReIntegrate.size());
    return BF_OK;
}
int bf_trajectory_manager_get_optimized_transforms(bf_trajectory_manager* tm, float* h_out, uint32_t capacity, uint32_t* count) {    // .h:49-69
    BF_REQUIRE(tm && h_out && count, "null argument");
    const uint32_t n = std::min(std::min(tm->numAddedFrames, tm->numOptimizedFrames), capacity);
    for (uint32_t i = 0; i < n; ++i) {
        const m44 T = tm->frames[i].type == BF_TF_INVALID ? minfM() : tm->optimizedTransforms[i];
        memcpy(h_out + 16 * (size_t)i, T.e, 64);
    }
    *count = n;
    return BF_OK;
}
int bf_trajectory_manager_get_frame(bf_trajectory_manager* tm, uint32_t idx, int* type, float integratedTransform[16], float* dist) {
    BF_REQUIRE(tm && idx < tm->frames.size(), "frame out of range");
    if (type) *type = tm->frames[idx].type;
    if (integratedTransform) memcpy(integratedTransform, tm->frames[idx].integratedTransform.e, 64);
    if (dist) *dist = tm->frames[idx].dist;
    return BF_OK;
}

}  // extern "C"

// ================================================================================================ OnlineBundler
struct bf_online_bundler {
    enum State { DO_NOTHING, PROCESS, INVALIDATE };
    bf_global_app_state gas; bf_global_bundling_state gbs;
    bf_image_manager* im = nullptr;
    hipStream_t stream = nullptr;
    // BundlerInputData (OnlineBundlerHelper.h:7-66): ingest buffers are read in place, only the SIFT intensity image is owned
    uint32_t depthW = 0, depthH = 0, colorW = 0, colorH = 0, widthSIFT = 0, heightSIFT = 0;
    m44 siftIntrinsics, siftIntrinsicsInv;
    float *d_intensitySIFT = nullptr, *d_intensityFilterHelper = nullptr;
    uint32_t submapSize = 10, numOptPerResidualRemoval = 1;
    bf_bundler *local = nullptr, *optLocal = nullptr, *global = nullptr;
    bf_trajectory_manager* tm = nullptr;
    m44 *d_completeTrajectory = nullptr, *d_localTrajectories = nullptr, *d_siftTrajectory = nullptr, *d_currIntegrateTransform = nullptr;
    int* d_imageInvalidateList = nullptr;
    std::vector<std::vector<int>> localTrajectoriesValid;
    std::vector<int> invalidImagesList;
    std::vector<m44> currIntegrateTransform;
    m44* h_pinT = nullptr;                 // PEND pinned slots, indexed by frame % PEND (several processInput calls may be in flight)
    // BundlerState (OnlineBundlerHelper.h:70-109)
    int lastFrameProcessed = -1; bool bLastFrameValid = false;
    int localToSolve = -1, lastLocalSolved = -1;
    uint32_t numFramesPastEnd = 0, numCompleteTransforms = 0, lastValidCompleteTransform = 0;
    bool bGlobalTrackingLost = false;
    State processState = DO_NOTHING;
    bool bUseSolve = true;
    uint32_t totalNumOptLocalFrames = 0;
    uint32_t numLocalSolves = 0, numGlobalSolves = 0;
    // Detect-ahead (bf_online_bundler_detect_ahead): SIFT detection and the dense cache frame of a frame depend on its pixels only,
    // so they can be computed on their own stream into a two-slot staging bundler while the previous frame is still being
    // matched / solved; processInput then commits the staged slot (one copy kernel) instead of detecting.
    bf_bundler* stage = nullptr;
    hipStream_t detectStream = nullptr;
    // a second detection queue (bf_online_bundler_set_second_detect_stream): odd frames detect on it, with their own detector and intensity images - two detections in flight
    hipStream_t detectStream2 = nullptr; bf_sift* sift2 = nullptr; float *d_intensitySIFT2 = nullptr, *d_intensityFilterHelper2 = nullptr;
    static const uint32_t STAGE = 4;      // staging slots: frame n is detected into slot n % STAGE (the loop may run up to STAGE - 1 frames behind its input)
    hipEvent_t evDetect[STAGE] = {}, evStageFree[STAGE] = {}, evCache[STAGE] = {};
    int stagedFrame[STAGE] = {-1, -1, -1, -1};
    // processInput calls in flight (between _begin and _end), oldest first.  Two deep: the matching chain of frame k + 1 is enqueued on the bundling
    // stream BEHIND frame k's before the host waits for frame k's result - everything frame k + 1's chain needs of frame k is device state (key points,
    // validity flags, the SIFT trajectory entry: the fix-up of an invalid frame, OnlineBundler.cpp:215-221, is the kernel k_sift_fixup).
    struct Pend { int phase = 0; uint32_t frame = 0, cur = 0, num = 0; bool lastLocal = false, match = false, swapped = false; bf_bundler* b = nullptr; };
    static constexpr int PEND = 4;         // processInput calls that may be in flight
    Pend pend[PEND];
    int pendHead = 0, pendCount = 0;
    // ---- lagged solve (bf_online_bundler_set_solve_lag): the chunk's solves (optimizeLocal + processGlobal + optimizeGlobal, OnlineBundler.cpp:229-240) run on
    // their own host thread and stream, like the reference's optimiser thread (FriedLiver.cpp:112-123) - but their results become visible at a DEFINED
    // frame: the solves started by chunk-closing frame b are applied when frame b + lag enters processInput (trajectory, last valid transform) and its
    // re-integration scheduling (TrajectoryManager), never earlier and never later, so runs are reproducible and comparable with the oracle loop under the
    // same lag (tests/oracle_pipeline.py: solve_lag).  lag = 0: the reference's serial order (the default).
    uint32_t solveLag = 0;
    hipStream_t sSolve = nullptr;          // stream of the solves: == stream when lag == 0
    m44 *d_completeOut = nullptr, *d_completeShadow = nullptr;      // where k_update_trajectory writes: d_completeTrajectory itself (serial) or the shadow that is swapped in at apply time
    hipEvent_t evChunk = nullptr;          // recorded on the bundling stream behind the chunk's last chain: the solve stream waits for it
    struct Published { bool haveTraj = false; uint32_t numTotal = 0; bool setLastValid = false; uint32_t lastValid = 0; bool haveLost = false, trackingLost = false; std::vector<m44> complete; };
    struct Job {
        bool active = false, bundleApplied = false;
        uint32_t applyAt = 0;              // first frame that sees the results
        std::thread th; int rc = BF_OK; std::string message;
        Published pub;
        bf_bundler* chunk = nullptr;       // the optLocal the job works on (gets its stream back at apply time)
    } job;
    Published* pubTarget = nullptr;        // non-null while a job runs: obPublish stores here instead of applying
    bool chunkClosed = false;              // prepareLocalSolve ran and its solves have not been started yet (main thread only; processState itself belongs to the solves)
    // ---- pair stages of consecutive frames side by side (bf_online_bundler_set_pair_streams): the staged detection of frame k is committed and its match /
    // Kabsch / surface-area / dense-verification kernels run on pair stream k & 1, into the sift manager's result set k & 1, while frame k - 1's are still
    // running on the other stream; the bundling stream only carries the short commit stage (valid flags, filterFrames, EntryJ rows, pose kernel).  The
    // greedy Kabsch filter is a ~0.26 ms chain of dependent 3x3 SVDs on ten waves (profiles/r03_sq_feature_pipeline.md): it cannot be made shorter, but two
    // of them fit beside each other.
    hipStream_t sPair[2] = {nullptr, nullptr};
    hipEvent_t evImage[2] = {nullptr, nullptr}, evPairDone[2] = {nullptr, nullptr}, evPairSetFree[2] = {nullptr, nullptr}, evChunkCopy = nullptr;
    bool pairSetUsed[2] = {false, false};
    hipEvent_t lastImageEvent = nullptr;   // behind the copy of the newest image into m_local (by a pair stream, or the chunk-boundary copy on the bundling stream)
    // chunk-parallel mode (bf_pipeline_process_frame_chunked): the local half of the chunk being closed comes from this package
    const bf_chunk_header* extChunk = nullptr;
    bool isLastLocalFrame(uint32_t curFrame) const { return curFrame >= submapSize && (curFrame % submapSize) == 0; }
    void invalidateImages(uint32_t s, uint32_t e = 0xFFFFFFFFu) { if (e == 0xFFFFFFFFu) invalidImagesList[s] = 0; else for (uint32_t i = s; i < e; ++i) invalidImagesList[i] = 0; }
    void validateImages(uint32_t s) { invalidImagesList[s] = 1; }
};

namespace {

const int ID_MARK_OFFSET = 2;

// chunk-parallel mode: what Bundler::fuseToGlobal (:388-394) leaves in the global bundler — one more image (the fused key points and
// descriptors) and one more cache frame (the chunk's first) — taken from a chunk package instead of m_optLocal
// A package arrives by all-gather from another rank: before anything in it is used as an offset or an index, every field that is
// one is checked against the buffer it came in (`bytes`) and against this bundler's configuration.
int chunkPackageCheck(const bf_chunk_header* h, uint64_t bytes, uint32_t maxKeys, uint32_t cacheW, uint32_t cacheH, uint32_t submapSize) {
    BF_REQUIRE(h && bytes >= sizeof(bf_chunk_header), "chunk package: shorter than its header");
    BF_REQUIRE(h->magic == BF_CHUNK_MAGIC, "chunk package: bad magic (a rank without a chunk in this round sends zeros)");
    BF_REQUIRE(h->totalBytes <= bytes && h->totalBytes >= sizeof(bf_chunk_header), "chunk package: totalBytes exceeds the buffer");
    BF_REQUIRE(h->submapSize == submapSize && h->numFrames >= 1 && h->numFrames <= submapSize + 1 && h->numFrames <= BF_CHUNK_MAX_FRAMES, "chunk package: chunk size differs");
    BF_REQUIRE(h->maxKeys == maxKeys && h->numKeys <= maxKeys, "chunk package: key capacity differs");
    BF_REQUIRE(h->cacheWidth == cacheW && h->cacheHeight == cacheH, "chunk package: cache geometry differs");
    const uint64_t n = (uint64_t)cacheW * cacheH;
    const uint64_t need[8] = {(uint64_t)sizeof(bf_sift_keypoint) * h->numKeys, (uint64_t)128 * h->numKeys, n * 4, n * 16, n * 4, n * 8, n * 4, n * 16};
    const uint64_t off[8] = {h->offKeys, h->offDescs, h->offCache[0], h->offCache[1], h->offCache[2], h->offCache[3], h->offCache[4], h->offCache[5]};
    for (int k = 0; k < 8; ++k)
        BF_REQUIRE(off[k] >= sizeof(bf_chunk_header) && off[k] <= h->totalBytes && need[k] <= h->totalBytes - off[k], "chunk package: a payload section lies outside the package");
    for (uint32_t j = 1; j < h->numFrames; ++j)          // (record 0 is never read: the chunk's first frame is chained by the previous chunk)
        BF_REQUIRE(h->frames[j].prevLocal < (int32_t)j, "chunk package: a frame is chained to a later frame");       // -1 = none; k_sift_transform_ext indexes curFrame - (localIdx - prevLocal)
    return BF_OK;
}

int obAppendKeyFrame(bf_bundler* glob, const bf_chunk_header* h) {
    BF_REQUIRE(h->magic == BF_CHUNK_MAGIC && h->numKeys <= glob->maxKeys, "bad chunk package");
    const uint8_t* base = reinterpret_cast<const uint8_t*>(h);
    bf_sift_image_gpu img;
    BF_TRY(bf_siftmgr_create_image(glob->mgr, &img));
    if (h->numKeys) {
        BF_HIP_TRY(hipMemcpyAsync(img.d_keyPoints, base + h->offKeys, sizeof(bf_sift_keypoint) * h->numKeys, hipMemcpyHostToDevice, glob->stream));
        BF_HIP_TRY(hipMemcpyAsync(img.d_keyPointDescs, base + h->offDescs, (size_t)128 * h->numKeys, hipMemcpyHostToDevice, glob->stream));
    }
    BF_TRY(bf_siftmgr_finalize_image(glob->mgr, (int32_t)h->numKeys));
    uint32_t ci, cw, ch; float k4[4];
    BF_TRY(bf_cache_get_num_frames(glob->cache, &ci));
    BF_TRY(bf_cache_get_geometry(glob->cache, &cw, &ch, k4));
    BF_REQUIRE(cw == h->cacheWidth && ch == h->cacheHeight, "chunk package: cache geometry differs");
    bf_cached_frame cd;
    BF_TRY(bf_cache_get_frame(glob->cache, ci, &cd));
    const size_t n = (size_t)cw * ch;
    void* dst[6] = {cd.d_depthDownsampled, cd.d_cameraposDownsampled, cd.d_intensityDownsampled, cd.d_intensityDerivsDownsampled, cd.d_normalsDownsampledUCHAR4, cd.d_normalsDownsampled};
    const size_t bytes[6] = {n * 4, n * 16, n * 4, n * 8, n * 4, n * 16};
    for (int k = 0; k < 6; ++k) BF_HIP_TRY(hipMemcpyAsync(dst[k], base + h->offCache[k], bytes[k], hipMemcpyHostToDevice, glob->stream));
    BF_TRY(bf_cache_increment(glob->cache));
    BF_HIP_TRY(hipStreamSynchronize(glob->stream));      // like fuseToGlobal: the package memory may be reused after the call
    return BF_OK;
}

// `chunk`: the bundler that holds the chunk being closed (m_local at the time of the call in the reference); swap = false when processInput_begin has
// already exchanged m_local / m_optLocal (it does so for a chunk's last frame, so that the next frame's chain can be enqueued before this frame's result
// has been read back)
int obPrepareLocalSolve(bf_online_bundler* ob, uint32_t curFrame, bool isSequenceEnd, bf_bundler* chunk, bool swap) {            // OnlineBundler.cpp:134-165
    ob->processState = bf_online_bundler::DO_NOTHING;
    uint32_t curLocalIdx = (std::max(curFrame, 1u) - 1) / ob->submapSize;
    if (isSequenceEnd && (curFrame % ob->submapSize) == 0) {
        curLocalIdx++;
        ob->localToSolve = -((int)curLocalIdx + ID_MARK_OFFSET);
        ob->processState = bf_online_bundler::INVALIDATE;
    } else {
        int valid = 0;
        BF_TRY(bf_bundler_is_valid(chunk, &valid));
        if (valid) { ob->localToSolve = (int)curLocalIdx; ob->processState = bf_online_bundler::PROCESS; }
        else { ob->localToSolve = -((int)curLocalIdx + ID_MARK_OFFSET); ob->processState = bf_online_bundler::INVALIDATE; }
    }
    if (swap) std::swap(ob->local, ob->optLocal);
    ob->chunkClosed = true;
    return BF_OK;
}

void obSetTrackingLost(bf_online_bundler* ob, bool v) {
    if (ob->pubTarget) { ob->pubTarget->haveLost = true; ob->pubTarget->trackingLost = v; }
    else ob->bGlobalTrackingLost = v;
}

// what a global optimisation makes visible to the frame loop (OnlineBundler.cpp:394-401): the complete trajectory (already written to d_completeOut by
// obUpdateTrajectory) in the TrajectoryManager, the number of complete transforms and - after a valid solve - the last valid complete transform.
// Serial order: at once.  Inside a lagged job: stored, applied by obApplyBundleSide / obApplyTmSide at the job's frame.
int obPublishTrajectory(bf_online_bundler* ob, uint32_t numTotalFrames, bool setLastValid, uint32_t lastValid) {
    if (ob->pubTarget) {
        bf_online_bundler::Published& P = *ob->pubTarget;
        P.haveTraj = true; P.numTotal = numTotalFrames; P.setLastValid = setLastValid; P.lastValid = lastValid;
        P.complete.resize(numTotalFrames);
        if (numTotalFrames) {
            BF_HIP_TRY(hipMemcpyAsync(P.complete.data(), ob->d_completeOut, sizeof(m44) * numTotalFrames, hipMemcpyDeviceToHost, ob->sSolve));
            BF_HIP_TRY(hipStreamSynchronize(ob->sSolve));
        }
        return BF_OK;
    }
    BF_TRY(bf_trajectory_manager_update_optimized_transform(ob->tm, (const float*)ob->d_completeTrajectory, numTotalFrames, ob->sSolve));
    ob->numCompleteTransforms = numTotalFrames;
    if (setLastValid) ob->lastValidCompleteTransform = lastValid;
    return BF_OK;
}

int obOptimizeLocal(bf_online_bundler* ob, uint32_t numNonLin, uint32_t numLin) {                    // :242-271
    if (ob->processState == bf_online_bundler::DO_NOTHING) return BF_OK;
    const bf_online_bundler::State optLocalState = ob->processState;
    ob->processState = bf_online_bundler::DO_NOTHING;
    uint32_t curLocalIdx = 0xFFFFFFFFu, nOpt;
    if (ob->extChunk) nOpt = ob->extChunk->numFrames;
    else BF_TRY(bf_bundler_get_num_frames(ob->optLocal, &nOpt));
    const uint32_t numLocalFrames = std::min(ob->submapSize, nOpt);
    if (optLocalState == bf_online_bundler::PROCESS) {
        curLocalIdx = (uint32_t)ob->localToSolve;
        int removed = 0, valid = 0;
        if (ob->extChunk) valid = ob->extChunk->solveValid;          // the chunk's worker ran Bundler::optimize
        else BF_TRY(bf_bundler_optimize(ob->optLocal, numNonLin, numLin, ob->gbs.s_useLocalVerify, 0, ob->numFramesPastEnd != 0, &removed, &valid));
        ob->numLocalSolves++;
        if (valid) {
            if (ob->extChunk) BF_HIP_TRY(hipMemcpyAsync(ob->d_localTrajectories + (size_t)(ob->submapSize + 1) * curLocalIdx, ob->extChunk->localTrajectory,
                                                        sizeof(m44) * (ob->submapSize + 1), hipMemcpyHostToDevice, ob->sSolve));
            else BF_HIP_TRY(hipMemcpyAsync(ob->d_localTrajectories + (size_t)(ob->submapSize + 1) * curLocalIdx, ob->optLocal->d_trajectory, sizeof(m44) * (ob->submapSize + 1),
                                      hipMemcpyDeviceToDevice, ob->sSolve));
            ob->processState = bf_online_bundler::PROCESS;
        } else ob->processState = bf_online_bundler::INVALIDATE;
    } else if (optLocalState == bf_online_bundler::INVALIDATE) {
        curLocalIdx = (uint32_t)(-ob->localToSolve - ID_MARK_OFFSET);
        ob->processState = bf_online_bundler::INVALIDATE;
    }
    ob->localToSolve = -1;
    ob->lastLocalSolved = (int)curLocalIdx;
    ob->totalNumOptLocalFrames = ob->submapSize * (uint32_t)ob->lastLocalSolved + numLocalFrames;
    return BF_OK;
}

int obProcessGlobal(bf_online_bundler* ob) {                                                       // :280-361
    const bf_online_bundler::State processState = ob->processState;
    if (processState == bf_online_bundler::DO_NOTHING) {
        if (ob->numFramesPastEnd != 0) {
            uint32_t idx;
            BF_TRY(bf_bundler_try_revalidation(ob->global, (uint32_t)ob->lastLocalSolved, 1, &idx));
            if (idx != 0xFFFFFFFFu && idx < ob->localTrajectoriesValid.size()) {
                const std::vector<int>& validLocal = ob->localTrajectoriesValid[idx];
                for (uint32_t i = 0; i < validLocal.size(); ++i) if (validLocal[i] == 1) ob->validateImages(idx * ob->submapSize + i);
                ob->processState = bf_online_bundler::PROCESS;
            }
        }
        return BF_OK;
    }
    ob->processState = bf_online_bundler::DO_NOTHING;
    if (processState == bf_online_bundler::PROCESS) {
        uint32_t curGlobalFrame, nOpt;
        std::vector<int> validImagesLocal(ob->submapSize + 1, 0);
        if (ob->extChunk) {                          // the fused key frame, the chunk's size and its valid flags come with the package
            BF_TRY(obAppendKeyFrame(ob->global, ob->extChunk));
            nOpt = ob->extChunk->numFrames;
            for (uint32_t i = 0; i <= ob->submapSize && i < BF_CHUNK_MAX_FRAMES; ++i) validImagesLocal[i] = ob->extChunk->validImages[i];
        } else {
            BF_TRY(bf_bundler_fuse_to_global(ob->optLocal, ob->global));
            BF_TRY(bf_bundler_get_num_frames(ob->optLocal, &nOpt));
            BF_TRY(bf_bundler_get_valid_images(ob->optLocal, validImagesLocal.data(), ob->submapSize + 1));
        }
        BF_TRY(bf_bundler_get_curr_frame_number(ob->global, &curGlobalFrame));
        const uint32_t numLocalFrames = std::min(ob->submapSize, nOpt);
        uint32_t lastValidLocal = 0;
        for (int i = (int)nOpt - 1; i >= 0; --i) if (validImagesLocal[i]) { lastValidLocal = (uint32_t)i; break; }
        for (uint32_t i = 0; i < numLocalFrames; ++i) if (validImagesLocal[i] == 0) ob->invalidateImages(curGlobalFrame * ob->submapSize + i);
        ob->localTrajectoriesValid[curGlobalFrame] = validImagesLocal;
        ob->localTrajectoriesValid[curGlobalFrame].resize(numLocalFrames);
        uint32_t nGlob;
        BF_TRY(bf_bundler_get_num_frames(ob->global, &nGlob));
        BF_TRY(bf_init_next_global_transform((float*)ob->global->d_trajectory, nGlob, curGlobalFrame, (const float*)ob->d_localTrajectories, lastValidLocal,
                                             ob->submapSize + 1, ob->sSolve));
        if (!ob->extChunk) BF_TRY(bf_bundler_reset(ob->optLocal));
        if (nGlob > 1) {
            uint32_t lastMatchedGlobal;
            BF_TRY(bf_bundler_match_and_filter(ob->global, &lastMatchedGlobal));
            if (lastMatchedGlobal == 0xFFFFFFFFu) { obSetTrackingLost(ob, true); ob->processState = bf_online_bundler::INVALIDATE; }
            else {
                obSetTrackingLost(ob, false);
                const uint32_t revalidateIdx = ob->global->revalidatedIdx;
                if (revalidateIdx != 0xFFFFFFFFu) {
                    const std::vector<int>& validLocal = ob->localTrajectoriesValid[revalidateIdx];
                    for (uint32_t i = 0; i < validLocal.size(); ++i) if (validLocal[i] == 1) ob->validateImages(revalidateIdx * ob->submapSize + i);
                }
                ob->processState = bf_online_bundler::PROCESS;
            }
        }
    } else if (processState == bf_online_bundler::INVALIDATE) {
        ob->processState = bf_online_bundler::INVALIDATE;
        BF_TRY(bf_bundler_add_invalid_frame(ob->global));
        if (!ob->extChunk) BF_TRY(bf_bundler_reset(ob->optLocal));
        ob->invalidateImages(ob->submapSize * (uint32_t)ob->lastLocalSolved, ob->totalNumOptLocalFrames);
    }
    return BF_OK;
}

int obUpdateTrajectory(bf_online_bundler* ob, uint32_t curFrame) {                                    // :363-371
    if (curFrame) BF_HIP_TRY(hipMemcpyAsync(ob->d_imageInvalidateList, ob->invalidImagesList.data(), sizeof(int) * curFrame, hipMemcpyHostToDevice, ob->sSolve));
    uint32_t nGlob;
    BF_TRY(bf_bundler_get_num_frames(ob->global, &nGlob));
    ob->d_completeOut = ob->pubTarget ? ob->d_completeShadow : ob->d_completeTrajectory;      // a lagged job writes the shadow; it is swapped in at the job's frame
    return bf_update_trajectory((const float*)ob->global->d_trajectory, nGlob, (float*)ob->d_completeOut, curFrame, (const float*)ob->d_localTrajectories,
                                ob->submapSize + 1, nGlob, ob->d_imageInvalidateList, ob->sSolve);
}

int obOptimizeGlobal(bf_online_bundler* ob, uint32_t numNonLin, uint32_t numLin) {                   // :373-408
    const bool isSequenceDone = ob->numFramesPastEnd > 0;
    if (!isSequenceDone && ob->processState == bf_online_bundler::DO_NOTHING) return BF_OK;
    if (ob->lastLocalSolved < 0) return BF_OK;       // MLIB_ASSERT(m_lastLocalSolved >= 0): nothing was ever solved (sequence shorter than one chunk)
    const bf_online_bundler::State state = isSequenceDone ? bf_online_bundler::PROCESS : ob->processState;
    const uint32_t numTotalFrames = ob->totalNumOptLocalFrames;
    if (state == bf_online_bundler::PROCESS) {
        uint32_t nGlob;
        BF_TRY(bf_bundler_get_num_frames(ob->global, &nGlob));
        const uint32_t countNumFrames = (ob->numFramesPastEnd > 0) ? ob->numFramesPastEnd : numTotalFrames / ob->submapSize;
        const bool bRemoveMaxResidual = (countNumFrames % ob->numOptPerResidualRemoval) == (ob->numOptPerResidualRemoval - 1);
        int removed = 0, valid = 1;
        if (nGlob > 1) {                               // MLIB_ASSERT(getNumImages() > 1) in Bundler::optimize: a one-keyframe global problem has nothing to solve
            BF_TRY(bf_bundler_optimize(ob->global, numNonLin, numLin, 0, bRemoveMaxResidual, ob->numFramesPastEnd > 0, &removed, &valid));
            ob->numGlobalSolves++;
        }
        if (removed) {
            std::vector<int> v(nGlob, 0);
            BF_TRY(bf_bundler_get_valid_images(ob->global, v.data(), nGlob));
            for (uint32_t i = 0; i < nGlob; ++i) if (v[i] == 0) ob->invalidateImages(i * ob->submapSize, std::min((i + 1) * ob->submapSize, numTotalFrames));
        }
        BF_TRY(obUpdateTrajectory(ob, numTotalFrames));
        BF_TRY(obPublishTrajectory(ob, numTotalFrames, valid != 0, ob->submapSize * (uint32_t)ob->lastLocalSolved));
    } else if (state == bf_online_bundler::INVALIDATE) {
        BF_TRY(bf_bundler_invalidate_last_frame(ob->global));
        ob->invalidateImages(ob->submapSize * (uint32_t)ob->lastLocalSolved, ob->totalNumOptLocalFrames);
        BF_TRY(obUpdateTrajectory(ob, numTotalFrames));
        BF_TRY(obPublishTrajectory(ob, numTotalFrames, false, 0));
    }
    ob->processState = bf_online_bundler::DO_NOTHING;
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_online_bundler_create(const bf_rgbd_sensor_desc* sensor, bf_image_manager* im, const bf_global_app_state* gas, const bf_global_bundling_state* gbs,
                             bf_online_bundler** out) {
    BF_REQUIRE(sensor && im && gas && gbs && out, "null argument");
    bf_online_bundler* ob = new bf_online_bundler;
    ob->gas = *gas; ob->gbs = *gbs; ob->im = im;
    ob->depthW = sensor->depthWidth; ob->depthH = sensor->depthHeight; ob->colorW = sensor->colorWidth; ob->colorH = sensor->colorHeight;
    ob->widthSIFT = gbs->s_widthSIFT; ob->heightSIFT = gbs->s_heightSIFT;
    ob->siftIntrinsics = scaleIntrinsics(sensor->colorIntrinsics, ob->widthSIFT, ob->heightSIFT, ob->colorW, ob->colorH);
    ob->siftIntrinsicsInv = inverse44(ob->siftIntrinsics);
    ob->submapSize = gbs->s_submapSize; ob->numOptPerResidualRemoval = std::max(gbs->s_numOptPerResidualRemoval, 1u);
    const uint32_t maxNumImages = gbs->s_maxNumImages, S = ob->submapSize;
    int rc = bf_bundler_create(S + 1, gbs->s_maxNumKeysPerImage, ob->siftIntrinsicsInv.e, im, 1, gas, gbs, &ob->local);
    if (!rc) rc = bf_bundler_create(S + 1, gbs->s_maxNumKeysPerImage, ob->siftIntrinsicsInv.e, im, 1, gas, gbs, &ob->optLocal);
    if (!rc) rc = bf_bundler_create(maxNumImages, gbs->s_maxNumKeysPerImage, ob->siftIntrinsicsInv.e, im, 0, gas, gbs, &ob->global);
    if (!rc) rc = bf_trajectory_manager_create(maxNumImages * S, gas->s_topNActive, gas->s_minPoseDistSqrt, &ob->tm);
    if (!rc) rc = bf_bundler_create(bf_online_bundler::STAGE, gbs->s_maxNumKeysPerImage, ob->siftIntrinsicsInv.e, im, 1, gas, gbs, &ob->stage);
    for (uint32_t k = 0; k < bf_online_bundler::STAGE && !rc; ++k) {                     // the staging slots exist from the start (empty images)
        bf_sift_image_gpu img;
        rc = bf_siftmgr_create_image(ob->stage->mgr, &img);
        if (!rc) rc = bf_siftmgr_finalize_image(ob->stage->mgr, 0);
    }
    if (rc) { bf_online_bundler_destroy(ob); return rc; }
    for (uint32_t k = 0; k < bf_online_bundler::STAGE; ++k) {
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evDetect[k], hipEventDisableTiming));
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evCache[k], hipEventDisableTiming));
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evStageFree[k], hipEventDisableTiming));
    }
    const size_t nAll = (size_t)maxNumImages * S;
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_intensitySIFT, sizeof(float) * ob->widthSIFT * ob->heightSIFT));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_intensityFilterHelper, sizeof(float) * ob->widthSIFT * ob->heightSIFT));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_completeTrajectory, sizeof(m44) * nAll));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_localTrajectories, sizeof(m44) * (size_t)maxNumImages * (S + 1)));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_siftTrajectory, sizeof(m44) * nAll));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_currIntegrateTransform, sizeof(m44) * nAll));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_imageInvalidateList, sizeof(int) * nAll));
    BF_HIP_TRY(hipHostMalloc((void**)&ob->h_pinT, bf_online_bundler::PEND * sizeof(m44)));
    BF_HIP_TRY(BF_MALLOC((void**)&ob->d_completeShadow, sizeof(m44) * nAll));
    BF_HIP_TRY(hipMemset(ob->d_completeShadow, 0, sizeof(m44) * nAll));
    BF_HIP_TRY(hipEventCreateWithFlags(&ob->evChunk, hipEventDisableTiming));
    BF_HIP_TRY(hipEventCreateWithFlags(&ob->evChunkCopy, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evImage[k], hipEventDisableTiming));
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evPairDone[k], hipEventDisableTiming));
        BF_HIP_TRY(hipEventCreateWithFlags(&ob->evPairSetFree[k], hipEventDisableTiming));
    }
    k_fill_identity<<<div_up((uint32_t)(maxNumImages * (S + 1)), 64), 64>>>(ob->d_localTrajectories, maxNumImages * (S + 1));
    k_fill_identity<<<1, 64>>>(ob->d_siftTrajectory, 1);
    k_fill_identity<<<1, 64>>>(ob->d_currIntegrateTransform, 1);
    BF_HIP_TRY(hipMemset(ob->d_completeTrajectory, 0, sizeof(m44) * nAll));
    BF_HIP_TRY(hipDeviceSynchronize());
    ob->localTrajectoriesValid.resize(maxNumImages);
    ob->invalidImagesList.assign(nAll, 1);
    ob->currIntegrateTransform.assign(nAll, minfM());
    ob->currIntegrateTransform[0] = identity44();
    *out = ob;
    return BF_OK;
}

int bf_online_bundler_destroy(bf_online_bundler* ob) {
    if (!ob) return BF_OK;
    if (ob->job.th.joinable()) ob->job.th.join();
    (void)hipFree(ob->d_completeShadow);
    if (ob->evChunk) (void)hipEventDestroy(ob->evChunk);
    if (ob->evChunkCopy) (void)hipEventDestroy(ob->evChunkCopy);
    for (int k = 0; k < 2; ++k) for (hipEvent_t e : {ob->evImage[k], ob->evPairDone[k], ob->evPairSetFree[k]}) if (e) (void)hipEventDestroy(e);
    bf_bundler_destroy(ob->local); bf_bundler_destroy(ob->optLocal); bf_bundler_destroy(ob->global); bf_bundler_destroy(ob->stage); bf_trajectory_manager_destroy(ob->tm);
    for (uint32_t k = 0; k < bf_online_bundler::STAGE; ++k) { if (ob->evDetect[k]) (void)hipEventDestroy(ob->evDetect[k]); if (ob->evCache[k]) (void)hipEventDestroy(ob->evCache[k]); if (ob->evStageFree[k]) (void)hipEventDestroy(ob->evStageFree[k]); }
    if (ob->sift2) bf_sift_destroy(ob->sift2);
    (void)hipFree(ob->d_intensitySIFT2); (void)hipFree(ob->d_intensityFilterHelper2);
    (void)hipFree(ob->d_intensitySIFT); (void)hipFree(ob->d_intensityFilterHelper); (void)hipFree(ob->d_completeTrajectory); (void)hipFree(ob->d_localTrajectories);
    (void)hipFree(ob->d_siftTrajectory); (void)hipFree(ob->d_currIntegrateTransform); (void)hipFree(ob->d_imageInvalidateList);
    if (ob->h_pinT) (void)hipHostFree(ob->h_pinT);
    delete ob;
    return BF_OK;
}

int bf_online_bundler_set_stream(bf_online_bundler* ob, void* s) {
    BF_REQUIRE(ob, "null bundler");
    BF_REQUIRE(!ob->job.active && ob->pendCount == 0, "set_stream with work in flight");
    ob->stream = (hipStream_t)s;
    BF_TRY(bf_bundler_set_stream(ob->local, s)); BF_TRY(bf_bundler_set_stream(ob->optLocal, s));
    if (ob->solveLag == 0) { ob->sSolve = ob->stream; BF_TRY(bf_bundler_set_stream(ob->global, s)); }
    return BF_OK;
}

// Lagged solve (see the struct): lag in [1, s_submapSize] frames, on `solveStream` (a stream of the caller's, other than the bundling stream).  lag = 0
// returns to the serial order.  Only between frames, with nothing in flight.
int bf_online_bundler_set_solve_lag(bf_online_bundler* ob, uint32_t lag, void* solveStream) {
    BF_REQUIRE(ob, "null bundler");
    BF_REQUIRE(!ob->job.active && ob->pendCount == 0, "set_solve_lag with work in flight");
    BF_REQUIRE(lag <= ob->submapSize, "the solve lag cannot exceed s_submapSize (the next chunk needs the optimiser's bundler back)");
    BF_REQUIRE(lag == 0 || (solveStream && (hipStream_t)solveStream != ob->stream), "a lagged solve needs its own stream");
    BF_HIP_TRY(hipStreamSynchronize(ob->stream));
    if (ob->sSolve && ob->sSolve != ob->stream) BF_HIP_TRY(hipStreamSynchronize(ob->sSolve));
    ob->solveLag = lag;
    ob->sSolve = lag ? (hipStream_t)solveStream : ob->stream;
    return bf_bundler_set_stream(ob->global, ob->sSolve);
}
// Two streams of the caller's for the pair stages of consecutive frames (see the struct); (null, null): everything on the bundling stream (default).
// Takes effect for frames whose detection was staged (bf_online_bundler_detect_ahead).
int bf_online_bundler_set_pair_streams(bf_online_bundler* ob, void* stream0, void* stream1) {
    BF_REQUIRE(ob, "null bundler");
    BF_REQUIRE(ob->pendCount == 0, "set_pair_streams with a frame in flight");
    BF_REQUIRE((stream0 == nullptr) == (stream1 == nullptr), "both pair streams or none");
    ob->sPair[0] = (hipStream_t)stream0; ob->sPair[1] = (hipStream_t)stream1;
    return BF_OK;
}
int bf_online_bundler_get_solve_lag(bf_online_bundler* ob, uint32_t* lag) { BF_REQUIRE(ob && lag, "null argument"); *lag = ob->solveLag; return BF_OK; }

// processInput (:167-227) in two halves: _begin enqueues everything up to the frame read-back, _end fetches the result and
// finishes the host logic.  A caller may enqueue independent work (the re-integration of old frames on another stream)
// between the two; bf_online_bundler_process_input is simply begin + end.
int bf_online_bundler_set_detect_stream(bf_online_bundler* ob, void* s) {
    BF_REQUIRE(ob, "null bundler");
    ob->detectStream = (hipStream_t)s;
    return bf_bundler_set_stream(ob->stage, s);
}

// Odd frames detect on `s` with a detector of their own: the detection of a frame is ~30 dependent launches that leave most of the device idle, and one queue's worth of
// them (0.85 ms per frame in the frame loop) was what set the loop's period (profiles/r05_loop_trace.md).  Null: back to one queue.
int bf_online_bundler_set_second_detect_stream(bf_online_bundler* ob, void* s) {
    BF_REQUIRE(ob && ob->detectStream, "set_second_detect_stream needs bf_online_bundler_set_detect_stream first");
    ob->detectStream2 = (hipStream_t)s;
    if (!s) return BF_OK;
    if (!ob->sift2) {
        const bf_bundler* b = ob->stage;
        BF_TRY(bf_sift_create(b->gbs.s_widthSIFT, b->gbs.s_heightSIFT, ob->depthW, ob->depthH, 150, b->gas.s_sensorDepthMin, b->gas.s_sensorDepthMax, b->gbs.s_minKeyScale, b->maxKeys, &ob->sift2));
        BF_HIP_TRY(BF_MALLOC((void**)&ob->d_intensitySIFT2, sizeof(float) * ob->widthSIFT * ob->heightSIFT));
        BF_HIP_TRY(BF_MALLOC((void**)&ob->d_intensityFilterHelper2, sizeof(float) * ob->widthSIFT * ob->heightSIFT));
    }
    return bf_sift_set_stream(ob->sift2, s);
}

// Feature detection + dense cache frame of the image manager's current frame, into staging slot (frame % STAGE), on the detect
// stream (which must also be the image manager's stream, so that it is ordered after the ingest).  No bundler state changes.
int bf_online_bundler_detect_ahead(bf_online_bundler* ob) { return bf_online_bundler_detect_ahead_after(ob, nullptr); }

// ... with the ingest on ANOTHER stream than the detect stream: `ingest_event` was recorded behind the frame's ingest.  The detection's own completion
// event of the slot (frame parity) doubles as the guard of the image manager's input set of that parity (bf_image_manager_set_input_guard).
int bf_online_bundler_detect_ahead_after(bf_online_bundler* ob, void* ingest_event) {
    BF_REQUIRE(ob && ob->detectStream, "detect_ahead needs bf_online_bundler_set_detect_stream");
    uint32_t frame;
    BF_TRY(bf_image_manager_get_curr_frame_number(ob->im, &frame));
    const int slot = (int)(frame % bf_online_bundler::STAGE);
    BF_REQUIRE(ob->stagedFrame[slot] < 0, "staging slot still holds an uncommitted frame");
    // The second detection stream is ordered behind the frame's ingest by the ingest event only (and the cache frame then runs on the ingest stream): without an
    // event - bf_online_bundler_detect_ahead(), or an image manager that shares the detection stream - odd frames would detect beside their own ingest.
    BF_REQUIRE(ob->detectStream2 == nullptr || (ingest_event != nullptr && ob->im->stream != ob->detectStream2 && ob->im->stream != ob->detectStream),
               "a second detect stream needs the ingest on a stream of its own and its event (bf_online_bundler_detect_ahead_after)");
    const bool odd = ob->detectStream2 != nullptr && (frame & 1u) != 0u;
    hipStream_t sd = odd ? ob->detectStream2 : ob->detectStream;
    bf_sift* sift = odd ? ob->sift2 : ob->stage->sift;
    float*& intensity = odd ? ob->d_intensitySIFT2 : ob->d_intensitySIFT;
    float*& helper = odd ? ob->d_intensityFilterHelper2 : ob->d_intensityFilterHelper;
    if (ingest_event) BF_HIP_TRY(hipStreamWaitEvent(sd, (hipEvent_t)ingest_event, 0));
    BF_HIP_TRY(hipStreamWaitEvent(sd, ob->evStageFree[slot], 0));          // the commit that last read this slot
    // The dense cache frame needs the ingest's images only, not the detection: with the ingest on its own stream it follows the ingest THERE (two launches, ~115 us),
    // beside the ~30 launches of the detection instead of behind them - the detect stream was the frame loop's saturated queue (profiles/r05_loop_trace.md)
    hipStream_t si = ingest_event ? ob->im->stream : nullptr;
    const bool cacheBeside = si != nullptr && si != sd;
    if (cacheBeside) {
        BF_HIP_TRY(hipStreamWaitEvent(si, ob->evStageFree[slot], 0));
        BF_TRY(bf_cache_set_stream(ob->stage->cache, si));
        BF_TRY(bf_cache_set_current_frame(ob->stage->cache, (uint32_t)slot));
        const int rc = bf_cache_store_frame(ob->stage->cache, ob->im->d_depthInputRaw, ob->depthW, ob->depthH, ob->im->d_colorInput, ob->colorW, ob->colorH);
        BF_TRY(bf_cache_set_stream(ob->stage->cache, sd));
        BF_TRY(rc);
        BF_HIP_TRY(hipEventRecord(ob->evCache[slot], si));
    }
    BF_TRY(bf_image_resample_to_intensity(intensity, ob->widthSIFT, ob->heightSIFT, ob->im->d_colorInput, ob->colorW, ob->colorH, sd));
    if (ob->gas.s_colorFilter) {
        BF_TRY(bf_image_gauss_filter_intensity(helper, intensity, ob->gas.s_colorSigmaD, ob->widthSIFT, ob->heightSIFT, sd));
        std::swap(helper, intensity);
    }
    bf_sift_image_gpu img;
    BF_TRY(bf_siftmgr_get_image(ob->stage->mgr, (uint32_t)slot, &img));
    BF_TRY(bf_sift_run(sift, intensity, ob->im->d_depthInputFiltered, (float*)img.d_keyPoints, (uint8_t*)img.d_keyPointDescs, img.d_numKeyPoints));
    if (cacheBeside) BF_HIP_TRY(hipStreamWaitEvent(sd, ob->evCache[slot], 0));      // evDetect covers both: it is also the guard of the input set both read
    else {
        BF_TRY(bf_cache_set_current_frame(ob->stage->cache, (uint32_t)slot));
        BF_TRY(bf_cache_store_frame(ob->stage->cache, ob->im->d_depthInputRaw, ob->depthW, ob->depthH, ob->im->d_colorInput, ob->colorW, ob->colorH));
    }
    BF_HIP_TRY(hipEventRecord(ob->evDetect[slot], sd));
    ob->stagedFrame[slot] = (int)frame;
    return BF_OK;
}

}  // extern "C"

namespace {

struct alignas(32) CopySeg { uint32_t* dst; const uint32_t* src; uint32_t words; };      // (read at a run-time index: no scalar load of one straddles a 64-byte line)
struct CopySegs { CopySeg s[9]; };
__global__ __launch_bounds__(256) void k_copy_segments(CopySegs j) {
    const CopySeg c = j.s[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < c.words; i += gridDim.x * blockDim.x) c.dst[i] = c.src[i];
}

// the staged detection of `frame` becomes the local bundler's next image (what detectFeatures + storeCachedFrame would have produced)
int obCommitStaged(bf_online_bundler* ob, uint32_t frame, hipStream_t st) {
    const int slot = (int)(frame % bf_online_bundler::STAGE);
    BF_HIP_TRY(hipStreamWaitEvent(st, ob->evDetect[slot], 0));
    bf_bundler *b = ob->local, *from = ob->stage;
    bf_sift_image_gpu src, dst;
    BF_TRY(bf_siftmgr_get_image(from->mgr, (uint32_t)slot, &src));
    BF_TRY(bf_siftmgr_create_image(b->mgr, &dst));
    uint32_t ci;
    BF_TRY(bf_cache_get_num_frames(b->cache, &ci));
    bf_cached_frame cs, cd;
    BF_TRY(bf_cache_get_frame(from->cache, (uint32_t)slot, &cs));
    BF_TRY(bf_cache_get_frame(b->cache, ci, &cd));
    uint32_t cw, ch; float k4[4];
    BF_TRY(bf_cache_get_geometry(b->cache, &cw, &ch, k4));
    const uint32_t n = cw * ch, mk = std::min(b->maxKeys, from->maxKeys);
    CopySegs j;
    j.s[0] = {(uint32_t*)dst.d_keyPoints, (const uint32_t*)src.d_keyPoints, (uint32_t)(sizeof(bf_sift_keypoint) / 4) * mk};
    j.s[1] = {(uint32_t*)dst.d_keyPointDescs, (const uint32_t*)src.d_keyPointDescs, (uint32_t)(sizeof(bf_sift_keypoint_desc) / 4) * mk};
    j.s[2] = {(uint32_t*)dst.d_numKeyPoints, (const uint32_t*)src.d_numKeyPoints, 1u};
    j.s[3] = {(uint32_t*)cd.d_depthDownsampled, (const uint32_t*)cs.d_depthDownsampled, n};
    j.s[4] = {(uint32_t*)cd.d_cameraposDownsampled, (const uint32_t*)cs.d_cameraposDownsampled, 4 * n};
    j.s[5] = {(uint32_t*)cd.d_intensityDownsampled, (const uint32_t*)cs.d_intensityDownsampled, n};
    j.s[6] = {(uint32_t*)cd.d_intensityDerivsDownsampled, (const uint32_t*)cs.d_intensityDerivsDownsampled, 2 * n};
    j.s[7] = {(uint32_t*)cd.d_normalsDownsampledUCHAR4, (const uint32_t*)cs.d_normalsDownsampledUCHAR4, n};
    j.s[8] = {(uint32_t*)cd.d_normalsDownsampled, (const uint32_t*)cs.d_normalsDownsampled, 4 * n};
    hipLaunchKernelGGL(k_copy_segments, dim3(32, 9), dim3(256), 0, st, j);
    BF_HIP_TRY(hipGetLastError());
    BF_TRY(bf_siftmgr_finalize_image(b->mgr, -1));
    BF_TRY(bf_cache_increment(b->cache));
    BF_HIP_TRY(hipEventRecord(ob->evStageFree[slot], st));
    ob->stagedFrame[slot] = -1;
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_online_bundler_process_input_begin(bf_online_bundler* ob) {
    BF_REQUIRE(ob, "null bundler");
    uint32_t curFrame;
    BF_TRY(bf_image_manager_get_curr_frame_number(ob->im, &curFrame));
    return bf_online_bundler_process_input_begin_frame(ob, curFrame);
}

}  // extern "C"

namespace {

// OnlineBundler.cpp:215-221 on the device: a frame without a connection keeps the previous frame's SIFT pose
__global__ void k_sift_fixup(const int32_t* frameResult, m44* siftTrajectory, uint32_t curFrameIndexAll) {
    if (frameResult[1] == 0) siftTrajectory[curFrameIndexAll] = siftTrajectory[curFrameIndexAll - 1];
}

// ---- lagged solve: job = optimizeLocal + processGlobal + optimizeGlobal of one chunk, on the solve stream and its own host thread
int obWaitJob(bf_online_bundler* ob) {                     // the job's work is complete (nothing is applied)
    if (ob->job.th.joinable()) ob->job.th.join();
    if (ob->job.active && ob->job.rc != BF_OK) { set_error("lagged solve: %s", ob->job.message.c_str()); return ob->job.rc; }
    return BF_OK;
}

// what frame `frame`'s processInput sees of a finished job: the complete trajectory and the last valid complete transform (k_sift_transform's inputs)
int obApplyBundleSide(bf_online_bundler* ob, uint32_t frame, bool force) {
    if (!ob->job.active || ob->job.bundleApplied || (!force && frame < ob->job.applyAt)) return BF_OK;
    BF_TRY(obWaitJob(ob));
    const bf_online_bundler::Published& P = ob->job.pub;
    if (P.haveTraj) {
        std::swap(ob->d_completeTrajectory, ob->d_completeShadow);       // the job wrote every entry [0, numTotal) of the shadow; entries beyond are never read
        ob->numCompleteTransforms = P.numTotal;
        if (P.setLastValid) ob->lastValidCompleteTransform = P.lastValid;
    }
    BF_TRY(bf_bundler_set_stream(ob->job.chunk, ob->stream));             // the optimiser's bundler (reset by the job) goes back to the bundling stream
    ob->job.bundleApplied = true;
    return BF_OK;
}
// ... and what its re-integration scheduling sees: the optimised poses in the TrajectoryManager
int obApplyTmSide(bf_online_bundler* ob, uint32_t frame, bool force) {
    if (!ob->job.active || (!force && frame < ob->job.applyAt)) return BF_OK;
    BF_TRY(obApplyBundleSide(ob, frame, true));
    const bf_online_bundler::Published& P = ob->job.pub;
    if (P.haveTraj) BF_TRY(bf_trajectory_manager_update_optimized_transform_host(ob->tm, (const float*)P.complete.data(), P.numTotal));
    if (P.haveLost) ob->bGlobalTrackingLost = P.trackingLost;
    ob->job.active = false;
    return BF_OK;
}

// diagnostic BF_DEBUG_SOLVE_PROFILE=1: wall time of the chunk job's three stages (each closed by a stream synchronisation), printed when the process ends
struct SolveProfile {
    double t[3] = {0, 0, 0}; uint64_t n = 0; bool on = getenv("BF_DEBUG_SOLVE_PROFILE") != nullptr;
    ~SolveProfile() { if (on && n) fprintf(stderr, "solve job: %llu chunks, local solve %.3f ms, fuse + global matching %.3f ms, global solve %.3f ms per chunk\n", (unsigned long long)n, 1e3 * t[0] / n, 1e3 * t[1] / n, 1e3 * t[2] / n); }
};
SolveProfile g_solveProfile;

int obSolves(bf_online_bundler* ob, uint32_t nlLocal, uint32_t linLocal, uint32_t nlGlobal, uint32_t linGlobal) {
    if (!g_solveProfile.on) {
        BF_TRY(obOptimizeLocal(ob, nlLocal, linLocal));
        BF_TRY(obProcessGlobal(ob));
        return obOptimizeGlobal(ob, nlGlobal, linGlobal);
    }
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    BF_TRY(obOptimizeLocal(ob, nlLocal, linLocal)); (void)hipStreamSynchronize(ob->sSolve);
    double t1 = now(); g_solveProfile.t[0] += t1 - t0;
    BF_TRY(obProcessGlobal(ob)); (void)hipStreamSynchronize(ob->sSolve);
    double t2 = now(); g_solveProfile.t[1] += t2 - t1;
    const int rc = obOptimizeGlobal(ob, nlGlobal, linGlobal); (void)hipStreamSynchronize(ob->sSolve);
    g_solveProfile.t[2] += now() - t2; g_solveProfile.n++;
    return rc;
}

int obStartJob(bf_online_bundler* ob, uint32_t frame, uint32_t nlLocal, uint32_t linLocal, uint32_t nlGlobal, uint32_t linGlobal) {
    BF_REQUIRE(!ob->job.active, "the previous chunk's lagged solve has not been applied yet (lag > s_submapSize?)");
    bf_online_bundler::Job& J = ob->job;
    J.active = true; J.bundleApplied = false; J.applyAt = frame + ob->solveLag; J.rc = BF_OK; J.message.clear();
    J.pub = bf_online_bundler::Published();
    J.chunk = ob->optLocal;
    BF_TRY(bf_bundler_set_stream(ob->optLocal, ob->sSolve));
    BF_HIP_TRY(hipStreamWaitEvent(ob->sSolve, ob->evChunk, 0));           // the chunk's chains (bundling stream) before the solve reads the chunk
    ob->pubTarget = &J.pub;
    int dev = 0;
    BF_HIP_TRY(hipGetDevice(&dev));
    J.th = std::thread([ob, dev, nlLocal, linLocal, nlGlobal, linGlobal] {
        (void)hipSetDevice(dev);
        int rc = obSolves(ob, nlLocal, linLocal, nlGlobal, linGlobal);
        if (rc == BF_OK && hipStreamSynchronize(ob->sSolve) != hipSuccess) { set_error("solve stream failed"); rc = BF_ERR_HIP; }
        if (rc != BF_OK) { ob->job.rc = rc; ob->job.message = bf_last_error(); }
        ob->pubTarget = nullptr;
    });
    return BF_OK;
}

}  // namespace

extern "C" {

// processInput for an explicit frame: either the image manager's current frame (detected here, like the reference), or an
// earlier frame whose detection was staged by bf_online_bundler_detect_ahead while newer frames were already ingested.
int bf_online_bundler_process_input_begin_frame(bf_online_bundler* ob, uint32_t curFrame) {
    BF_REQUIRE(ob, "null bundler");
    BF_REQUIRE(ob->pendCount < bf_online_bundler::PEND, "too many processInput calls in flight");
    const bool staged = ob->stagedFrame[curFrame % bf_online_bundler::STAGE] == (int)curFrame;
    {
        uint32_t imFrame;
        BF_TRY(bf_image_manager_get_curr_frame_number(ob->im, &imFrame));
        BF_REQUIRE(staged || imFrame == curFrame, "frame is neither staged nor the image manager's current frame");
    }
    const bool bIsLastLocal = ob->isLastLocalFrame(curFrame);
    bf_online_bundler::Pend& P = ob->pend[(ob->pendHead + ob->pendCount) % bf_online_bundler::PEND];
    P = bf_online_bundler::Pend();
    P.frame = curFrame; P.lastLocal = bIsLastLocal; P.b = ob->local;
    if (curFrame > 0 && ob->lastFrameProcessed == (int)curFrame) {                 // sequence has ended
        BF_REQUIRE(ob->pendCount == 0, "end-of-sequence iteration with a frame still in flight");
        BF_TRY(obApplyTmSide(ob, curFrame, true));                                  // a lagged solve still pending is applied before the first iteration past the end
        if (ob->numFramesPastEnd == 0 && ob->localToSolve == -1) { if (!bIsLastLocal) BF_TRY(obPrepareLocalSolve(ob, curFrame, true, ob->local, true)); }
        const uint32_t numSolveFramesBeforeExit = ob->gas.s_numSolveFramesBeforeExit;
        if (numSolveFramesBeforeExit != 0xFFFFFFFFu) {
            if (ob->numFramesPastEnd == numSolveFramesBeforeExit) {                 // USE_GLOBAL_DENSE_AT_END (GlobalBundlingState.h:9)
                if (ob->lastFrameProcessed < 10000) {
                    ob->gbs.s_numGlobalNonLinIterations = 3;
                    const float sp[3] = {1.0f, 1.0f, 1.0f}, dd[3] = {15.0f, 15.0f, 15.0f}, dc[3] = {0.0f, 0.0f, 0.0f};
                    BF_TRY(bf_bundler_set_solve_weights(ob->global, sp, dd, dc, 3));
                }
            }
            if (ob->numFramesPastEnd == numSolveFramesBeforeExit + 1) ob->bUseSolve = false;       // "stopping solve"
        }
        ob->numFramesPastEnd++;
        P.phase = 2;                                                                // nothing to read back
        ob->pendCount++;
        return BF_OK;
    }
    BF_TRY(obApplyBundleSide(ob, curFrame, false));            // lagged solve: this is the frame that sees the new trajectory
    const int par = (int)(curFrame & 1u);
    const bool spec = staged && ob->sPair[0] != nullptr && ob->local->corrEvaluator == nullptr;       // pair stage on its own stream
    hipStream_t ps = spec ? ob->sPair[par] : ob->stream;
    if (spec) {
        if (ob->pairSetUsed[par]) BF_HIP_TRY(hipStreamWaitEvent(ps, ob->evPairSetFree[par], 0));      // the commit stage that read this result set two frames ago
        if (ob->lastImageEvent) BF_HIP_TRY(hipStreamWaitEvent(ps, ob->lastImageEvent, 0));            // the previous image of the chunk is in place
    }
    if (staged) BF_TRY(obCommitStaged(ob, curFrame, ps));
    else {
        // getCurrentFrame (:106-116): luminance at SIFT resolution straight from the ingest buffer
        BF_TRY(bf_image_resample_to_intensity(ob->d_intensitySIFT, ob->widthSIFT, ob->heightSIFT, ob->im->d_colorInput, ob->colorW, ob->colorH, ob->stream));
        if (ob->gas.s_colorFilter) {
            BF_TRY(bf_image_gauss_filter_intensity(ob->d_intensityFilterHelper, ob->d_intensitySIFT, ob->gas.s_colorSigmaD, ob->widthSIFT, ob->heightSIFT, ob->stream));
            std::swap(ob->d_intensityFilterHelper, ob->d_intensitySIFT);
        }
        BF_TRY(bf_bundler_detect_features(ob->local, ob->d_intensitySIFT, ob->im->d_depthInputFiltered));
        BF_TRY(bf_bundler_store_cached_frame(ob->local, ob->depthW, ob->depthH, ob->im->d_colorInput, ob->colorW, ob->colorH, ob->im->d_depthInputRaw));
    }
    if (spec) { BF_HIP_TRY(hipEventRecord(ob->evImage[par], ps)); ob->lastImageEvent = ob->evImage[par]; }
    else ob->lastImageEvent = nullptr;                                               // everything on the bundling stream: ordered by the stream
    uint32_t curLocalFrame;
    BF_TRY(bf_bundler_get_curr_frame_number(ob->local, &curLocalFrame));
    uint32_t start = 0;
    if (curLocalFrame > 0) {
        // ---- pair stage (matchAndFilter up to the dense verification)
        BF_TRY(bf_siftmgr_set_pair_stage(ob->local->mgr, spec ? (uint32_t)par : 0u, spec ? ps : nullptr, spec ? 1 : 0));
        BF_TRY(matchAndFilterPairs(ob->local, P.cur, start, P.num));
    }
    if (spec) {
        BF_HIP_TRY(hipEventRecord(ob->evPairDone[par], ps));
        BF_HIP_TRY(hipStreamWaitEvent(ob->stream, ob->evPairDone[par], 0));
    }
    if (bIsLastLocal) BF_TRY(bf_bundler_copy_frame(ob->optLocal, ob->local, curLocalFrame));
    if (curLocalFrame > 0) {
        // ---- commit stage + computeCurrentSiftTransform (:118-132) with ONE read-back: the pose kernel is enqueued before the frame
        // result is fetched (it writes nothing when no pair survived the filters, which is exactly the "invalid" case)
        BF_TRY(matchAndFilterCommit(ob->local, P.cur, start, P.num, spec));
        const float* d_Tinv = nullptr; const int32_t* d_nf = nullptr; const int32_t* d_res = nullptr;
        BF_TRY(bf_bundler_get_current_sift_transforms_gpu(ob->local, &d_Tinv));
        BF_TRY(bf_bundler_get_num_filt_matches_gpu(ob->local, &d_nf));
        BF_TRY(bf_siftmgr_get_frame_result_gpu(ob->local->mgr, &d_res));
        BF_TRY(bf_compute_sift_transform(d_Tinv, d_nf, (const float*)ob->d_completeTrajectory, ob->lastValidCompleteTransform, (float*)ob->d_siftTrajectory, curFrame,
                                         curLocalFrame, (float*)(ob->d_currIntegrateTransform + curFrame), ob->stream));
        k_sift_fixup<<<1, 1, 0, ob->stream>>>(d_res, ob->d_siftTrajectory, curFrame);
        BF_HIP_TRY(hipGetLastError());
        BF_HIP_TRY(hipMemcpyAsync(ob->h_pinT + (curFrame % bf_online_bundler::PEND), ob->d_currIntegrateTransform + curFrame, sizeof(m44), hipMemcpyDeviceToHost, ob->stream));
        BF_TRY(bf_siftmgr_prefetch_frame_result(ob->local->mgr));      // the read-back itself is enqueued now; _end only waits for it
        BF_TRY(bf_siftmgr_set_pair_stage(ob->local->mgr, spec ? (uint32_t)par : 0u, nullptr, 0));      // (direct callers of this bundler get the reference's behaviour)
        P.match = true;
    }
    if (spec) { BF_HIP_TRY(hipEventRecord(ob->evPairSetFree[par], ob->stream)); ob->pairSetUsed[par] = true; }
    if (bIsLastLocal) {
        // the chunk is complete on the device: from here on m_local is the next chunk (which starts with a copy of this frame) - the exchange of
        // prepareLocalSolve (:164), done now so that the next frame's chain can be enqueued before this frame's result is back
        BF_HIP_TRY(hipEventRecord(ob->evChunk, ob->stream));
        if (spec) { BF_HIP_TRY(hipEventRecord(ob->evChunkCopy, ob->stream)); ob->lastImageEvent = ob->evChunkCopy; }      // the next chunk's first image: the copy above
        std::swap(ob->local, ob->optLocal);
        P.swapped = true;
    }
    P.phase = 1;
    ob->pendCount++;
    return BF_OK;
}

int bf_online_bundler_process_input_end(bf_online_bundler* ob) {
    BF_REQUIRE(ob, "null bundler");
    BF_REQUIRE(ob->pendCount > 0, "process_input_end without process_input_begin");
    const bf_online_bundler::Pend P = ob->pend[ob->pendHead];
    ob->pendHead = (ob->pendHead + 1) % bf_online_bundler::PEND; ob->pendCount--;
    if (P.phase == 2) return BF_OK;
    const uint32_t curFrame = P.frame;
    ob->bLastFrameValid = true;
    if (P.match) {
        uint32_t last;
        BF_TRY(matchAndFilterFinish(P.b, P.cur, P.num, &last));
        ob->bLastFrameValid = last != 0xFFFFFFFFu;
        if (!ob->bLastFrameValid) ob->currIntegrateTransform[curFrame] = minfM();      // (the SIFT trajectory entry was fixed up on the device: k_sift_fixup)
        else ob->currIntegrateTransform[curFrame] = ob->h_pinT[curFrame % bf_online_bundler::PEND];
    }
    if (P.lastLocal) BF_TRY(obPrepareLocalSolve(ob, curFrame, false, P.b, !P.swapped));
    ob->lastFrameProcessed = (int)curFrame;
    return BF_OK;
}

int bf_online_bundler_process_input(bf_online_bundler* ob) {                                         // :167-227
    BF_TRY(bf_online_bundler_process_input_begin(ob));
    return bf_online_bundler_process_input_end(ob);
}

// `frame`: the frame whose body this is (only used by the lagged mode, to stamp the job)
int bf_online_bundler_process_frame(bf_online_bundler* ob, uint32_t frame, uint32_t nlLocal, uint32_t linLocal, uint32_t nlGlobal, uint32_t linGlobal) {
    BF_REQUIRE(ob, "null bundler");
    if (!ob->bUseSolve) return BF_OK;
    if (ob->solveLag > 0 && ob->numFramesPastEnd == 0 && !ob->extChunk) {
        if (!ob->chunkClosed) return BF_OK;              // process() does nothing between chunk ends (processState is not read here: a running job owns it)
        ob->chunkClosed = false;
        return obStartJob(ob, frame, nlLocal, linLocal, nlGlobal, linGlobal);
    }
    ob->chunkClosed = false;
    if (ob->solveLag > 0) {       // past the end of the sequence the solves run in the serial order (on the solve stream, from this thread)
        BF_TRY(obApplyTmSide(ob, frame, true));
        BF_HIP_TRY(hipStreamSynchronize(ob->stream));
        BF_TRY(obSolves(ob, nlLocal, linLocal, nlGlobal, linGlobal));
        BF_HIP_TRY(hipStreamSynchronize(ob->sSolve));
        return BF_OK;
    }
    return obSolves(ob, nlLocal, linLocal, nlGlobal, linGlobal);
}

int bf_online_bundler_process(bf_online_bundler* ob, uint32_t nlLocal, uint32_t linLocal, uint32_t nlGlobal, uint32_t linGlobal) {
    BF_REQUIRE(ob, "null bundler");
    return bf_online_bundler_process_frame(ob, ob->lastFrameProcessed >= 0 ? (uint32_t)ob->lastFrameProcessed : 0u, nlLocal, linLocal, nlGlobal, linGlobal);
}

// lagged mode: what frame `frame`'s re-integration scheduling has to see (call before the TrajectoryManager is consulted for that frame)
int bf_online_bundler_apply_lagged_solve(bf_online_bundler* ob, uint32_t frame) { BF_REQUIRE(ob, "null bundler"); return obApplyTmSide(ob, frame, false); }
// ... and: the solve thread has finished what it was given (nothing becomes visible earlier than its frame)
int bf_online_bundler_wait_solves(bf_online_bundler* ob) { BF_REQUIRE(ob, "null bundler"); return obWaitJob(ob); }

int bf_online_bundler_get_current_integration_frame(bf_online_bundler* ob, float siftTransform[16], uint32_t* frameIdx, int* bGlobalTrackingLost, int* valid) {
    BF_REQUIRE(ob && siftTransform && frameIdx && bGlobalTrackingLost && valid, "null argument");
    *bGlobalTrackingLost = ob->bGlobalTrackingLost ? 1 : 0;
    if (ob->bLastFrameValid && ob->lastFrameProcessed >= 0) {
        memcpy(siftTransform, ob->currIntegrateTransform[ob->lastFrameProcessed].e, 64);
        *frameIdx = (uint32_t)ob->lastFrameProcessed;
        *valid = 1;
    } else *valid = 0;
    return BF_OK;
}

int bf_online_bundler_get_trajectory_manager(bf_online_bundler* ob, bf_trajectory_manager** out) { BF_REQUIRE(ob && out, "null argument"); *out = ob->tm; return BF_OK; }
int bf_online_bundler_get_curr_processed_frame(bf_online_bundler* ob, int32_t* out) { BF_REQUIRE(ob && out, "null argument"); *out = ob->lastFrameProcessed; return BF_OK; }
int bf_online_bundler_get_bundler(bf_online_bundler* ob, int which, bf_bundler** out) {
    BF_REQUIRE(ob && out && which >= 0 && which <= 2, "bad argument");
    *out = which == 0 ? ob->local : which == 1 ? ob->optLocal : ob->global;
    return BF_OK;
}
int bf_online_bundler_get_complete_trajectory(bf_online_bundler* ob, float* h_out, uint32_t capacity, uint32_t* count) {
    BF_REQUIRE(ob && h_out && count, "null argument");
    const uint32_t n = std::min(ob->numCompleteTransforms, capacity);
    if (n) { BF_HIP_TRY(hipMemcpyAsync(h_out, ob->d_completeTrajectory, sizeof(m44) * n, hipMemcpyDeviceToHost, ob->stream)); BF_HIP_TRY(hipStreamSynchronize(ob->stream)); }
    *count = n;
    return BF_OK;
}
int bf_online_bundler_save_global_sparse_corrs_to_file(bf_online_bundler* ob, const char* filename) { BF_REQUIRE(ob, "null bundler"); return bf_bundler_save_sparse_corrs_to_file(ob->global, filename); }
int bf_online_bundler_initialize_correspondence_evaluator(bf_online_bundler* ob, const float* h_completeTrajectory, uint32_t numFrames, const char* logFilePrefix) {
    BF_REQUIRE(ob && h_completeTrajectory, "null argument");                                  // OnlineBundler.cpp:81-90: "only want global trajectory"
    std::vector<float> keyPoses;
    for (uint32_t i = 0; i < numFrames; i += ob->submapSize) keyPoses.insert(keyPoses.end(), h_completeTrajectory + 16 * (size_t)i, h_completeTrajectory + 16 * (size_t)i + 16);
    return bf_bundler_initialize_correspondence_evaluator(ob->global, keyPoses.data(), (uint32_t)(keyPoses.size() / 16), logFilePrefix);
}
int bf_online_bundler_finish_correspondence_evaluator_logging(bf_online_bundler* ob) {       // :480-487
    BF_REQUIRE(ob, "null bundler");
    BF_TRY(bf_bundler_finish_correspondence_evaluator_logging(ob->global));
    BF_TRY(bf_bundler_finish_correspondence_evaluator_logging(ob->local));
    return bf_bundler_finish_correspondence_evaluator_logging(ob->optLocal);
}

}  // extern "C"

// ================================================================================================ frame loop
struct bf_pipeline {
    bf_global_app_state gas; bf_global_bundling_state gbs; bf_rgbd_sensor_desc sensor;
    bf_image_manager* im = nullptr; bf_online_bundler* ob = nullptr; bf_scene* scene = nullptr;
    bf_depth_camera_params cam;
    // The bundling stream carries matching, filters, pose chaining and (serial order) the solves; the volume stream carries every voxel update and the
    // garbage collection (the volume's own preparation stream: allocation + block lists).  Re-integration of old frames depends only on host-side lists
    // and on frames ingested earlier, so it runs concurrently with the current frame's feature pipeline; integration of the current frame waits for its
    // ingest (evIngest).
    hipStream_t sBundle = nullptr, sVolume = nullptr;
    // Look-ahead: feature detection + dense cache frame of frame k+1 run on the detect stream (its ingest on sIngest) while frame k is matched, filtered,
    // integrated and solved; the body of a frame is executed by a later call (`depth` below) or by the first call that needs its result.  The per-frame
    // work and its order are unchanged, so are the results.
    hipStream_t sDetect = nullptr, sDetect2 = nullptr;      // sDetect2: the queue of the odd frames' detections (one of sPair) or null
    bool lookahead = true;
    // Two frames behind the input (round 4): the call that delivers frame n (1) enqueues the matching chain of frame n - 1 on the bundling stream - behind the
    // chain of frame n - 2, which the previous call enqueued -, (2) ingests and detects frame n, (3) runs the body of frame n - 2: its match result was
    // enqueued a whole call ago, so this thread normally does not wait for the GPU any more (round 3: 0.77 ms per frame of waiting for the chain it had
    // just enqueued).  A chunk's last frame in the serial order is the exception: its solves change what the next frame's chain reads, so its body
    // runs BEFORE that chain is enqueued (in the lagged mode there is no such dependence).
    int deferred = -1;              // frame that has been detected; its matching chain is not enqueued yet
    std::deque<uint32_t> begun;     // frames whose chain is enqueued and whose body has not run, oldest first (at most `depth`)
    uint32_t depth = 2;             // chains in flight across a call boundary + 1 (BF_PIPELINE_DEPTH, 2 .. PEND): the call that delivers frame n enqueues the chain of frame
                                    // n - 1 and runs the body of frame n - depth.  Measured (gpurun r04c, depth 2): a chain's twelve dependent launches take 1.0 - 1.2 ms from
                                    // enqueue to result next to the volume's and the detector's queues - twice the sum of their kernel times - so one call of slack is not enough
    hipStream_t sSolve = nullptr;   // stream of the lagged solves (bf_pipeline_set_solve_lag)
    hipStream_t sIngest = nullptr;  // the ingest filters of frame n + 1 run beside the detection of frame n (several input sets in the image manager)
    hipStream_t sPair[2] = {nullptr, nullptr};      // pair stages of consecutive frames side by side (bf_online_bundler_set_pair_streams)
    static const int NEV = 8;
    hipEvent_t evIngest[NEV] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // ring, indexed by frame
    // The volume stream is fed by its own host thread: the main thread decides WHAT to integrate (TrajectoryManager lists, poses)
    // and posts commands; the worker issues the launches (three per operator + the event operations: 13 us of HIP calls per operator,
    // 12 % of the wall time), so they do not serialize with the ~60 launches of the detect and bundling streams on one CPU thread.
    struct VolCmd { int kind; bf_depth_camera_data data; const void* texels; float T0[16], T1[16]; int waitEv; };   // kind: 0 integrate, 1 de-integrate, 2 fused re-integrate, 3 GC, 4 flush
    // Batched volume operators (round 5, bf_scene_run_batch): the volume thread collects a frame's operators - the integration of the previous frame, which
    // arrives last in that frame's body, and this frame's re-integrations - and issues them as ONE batch when the frame's garbage collection arrives
    // (DepthSensing.cpp:854-902 order kept: ..., integrate(k-1), fixes(k), GC(k), integrate(k), ...); a flush command (every accessor, bf_pipeline_synchronize)
    // issues what is pending.  Per batch: four launches and one pass over the touched blocks instead of 3 launches and one pass per operator.
    bool volBatching = true;
    std::vector<VolCmd> volPending;                // worker-thread only
    static const size_t MAX_QUEUE = 48;          // back-pressure: the volume thread may lag the bundling thread by a few frames at most
    std::thread worker;
    std::mutex mu;
    std::condition_variable cvWork, cvIdle;
    std::deque<VolCmd> queue;
    bool stop = false, busy = false;
    int workerError = BF_OK;
    std::string workerMessage;
    uint32_t numIntegrate = 0, numDeIntegrate = 0;
    double volBusy = 0.0, volCommands = 0.0;      // the volume thread: seconds spent issuing operators (HIP API calls), operators issued (bf_pipeline_get_volume_thread_profile)
    // wall time the calling thread spent in each part of plFrame, accumulated (bf_pipeline_get_host_profile): where the frame loop's
    // critical path lies without a profiler.  [0] enqueue of the previous frame's matching chain, [1] ingest + detection enqueue,
    // [2] re-integration commands, [3] wait for the matching result + host logic, [4] integration command, [5] solves, [6] ingest wait, [7] frames
    double hostProfile[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // diagnostic (BF_PIPELINE_TRACE=<file>): per frame the host times of the chain's enqueue and of the wait for its result, and the GPU times (HIP events on the
    // streams) at which the frame's detection ended and its matching chain started / ended, all on one clock - written when the pipeline is destroyed
    struct TraceRec { uint32_t frame; double hEnq0, hEnq1, hDet0, hDet1, hWait0, hWait1, hBody1; hipEvent_t gDet, gChain0, gChain1; };
    std::string tracePath; std::vector<TraceRec> trace; hipEvent_t traceBase = nullptr; double traceBaseHost = 0.0;
    TraceRec* traceOf(uint32_t frame) { for (size_t i = trace.size(); i-- > 0;) if (trace[i].frame == frame) return &trace[i]; return nullptr; }
    bool timings = false;
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bf_frame_timing last;
};

namespace {

int volExecute(bf_pipeline* p, const bf_pipeline::VolCmd& c) {                                     // DepthSensing.cpp:723-762
    if (c.kind == 3) return bf_scene_garbage_collect(p->scene);
    if (c.waitEv >= 0) BF_TRY(bf_scene_wait_event(p->scene, p->evIngest[c.waitEv]));
    if (c.texels) BF_TRY(bf_scene_set_frame_texels(p->scene, c.texels));
    if (c.kind == 0) return bf_scene_integrate(p->scene, c.T0, &c.data, &p->cam, nullptr);
    if (c.kind == 1) return bf_scene_deintegrate(p->scene, c.T0, &c.data, &p->cam, nullptr);
    return bf_scene_reintegrate(p->scene, c.T0, c.T1, &c.data, &p->cam);
}

int volSubmitPending(bf_pipeline* p) {
    std::vector<bf_pipeline::VolCmd>& q = p->volPending;
    if (q.empty()) return BF_OK;
    bf_scene_batch_op ops[BF_SCENE_BATCH_MAX];
    const uint32_t n = (uint32_t)q.size();
    for (uint32_t k = 0; k < n; ++k) {
        const bf_pipeline::VolCmd& c = q[k];
        ops[k].kind = c.kind; ops[k].reserved = 0;
        memcpy(ops[k].T0, c.T0, 64); memcpy(ops[k].T1, c.kind == 2 ? c.T1 : c.T0, 64);
        ops[k].data = c.data; ops[k].d_texels = c.texels;
        ops[k].wait_event = c.waitEv >= 0 ? (void*)p->evIngest[c.waitEv] : nullptr;
    }
    q.clear();
    return bf_scene_run_batch(p->scene, ops, n, &p->cam);
}

// the volume thread's handling of one command (see bf_pipeline::volBatching)
int volHandle(bf_pipeline* p, const bf_pipeline::VolCmd& c) {
    if (!p->volBatching) return c.kind == 4 ? BF_OK : volExecute(p, c);
    if (c.kind <= 2) {
        p->volPending.push_back(c);
        return p->volPending.size() == BF_SCENE_BATCH_MAX ? volSubmitPending(p) : BF_OK;
    }
    BF_TRY(volSubmitPending(p));
    return c.kind == 3 ? bf_scene_garbage_collect(p->scene) : BF_OK;
}

void volWorker(bf_pipeline* p) {
    for (;;) {
        bf_pipeline::VolCmd c;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cvWork.wait(lk, [p] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;             // stop requested and drained
            c = p->queue.front(); p->queue.pop_front();
            p->busy = true;
        }
        const double tv = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const int rc = volHandle(p, c);
        const double dv = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tv;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->volBusy += dv; p->volCommands += 1.0;
            if (rc != BF_OK && p->workerError == BF_OK) { p->workerError = rc; p->workerMessage = bf_last_error(); }
            p->busy = false;
            p->cvIdle.notify_all();
        }
    }
}

int volPost(bf_pipeline* p, int kind, uint32_t frame, const float* T0, const float* T1, int waitEv) {
    bf_pipeline::VolCmd c;
    c.kind = kind; c.waitEv = waitEv;
    c.data.d_depthData = nullptr; c.data.d_colorData = nullptr;
    c.texels = nullptr;
    if (kind < 3) {
        BF_TRY(bf_image_manager_get_integrate_frame_gpu(p->im, frame, &c.data.d_depthData, &c.data.d_colorData));   // resolved on the calling thread
        BF_TRY(bf_image_manager_get_integrate_frame_texels(p->im, frame, &c.texels));
    }
    if (T0) memcpy(c.T0, T0, 64);
    if (T1) memcpy(c.T1, T1, 64);
    if (p->timings) return volExecute(p, c);        // stage timings are taken with everything issued from the calling thread
    std::unique_lock<std::mutex> lk(p->mu);
    p->cvIdle.wait(lk, [p] { return p->queue.size() < bf_pipeline::MAX_QUEUE || p->workerError != BF_OK; });
    if (p->workerError != BF_OK) { set_error("volume worker: %s", p->workerMessage.c_str()); return p->workerError; }
    p->queue.push_back(c);
    p->cvWork.notify_one();
    return BF_OK;
}

int volDrain(bf_pipeline* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->worker.joinable() && p->workerError == BF_OK) {      // what the volume thread holds back for the next batch is issued now
        bf_pipeline::VolCmd f; memset(&f, 0, sizeof f); f.kind = 4; f.waitEv = -1;
        p->queue.push_back(f);
        p->cvWork.notify_one();
    }
    p->cvIdle.wait(lk, [p] { return p->queue.empty() && !p->busy; });
    if (p->workerError != BF_OK) { set_error("volume worker: %s", p->workerMessage.c_str()); return p->workerError; }
    return BF_OK;
}

int plIntegrate(bf_pipeline* p, uint32_t frameIdx, const float* T, bool de, int waitEv = -1) {
    if (!p->gas.s_integrationEnabled) return BF_OK;
    if (de) p->numDeIntegrate++; else p->numIntegrate++;
    return volPost(p, de ? 1 : 0, frameIdx, T, nullptr, waitEv);
}

int plReintegrate(bf_pipeline* p) {                                                                 // :854-902
    const uint32_t maxPerFrameFixes = p->gas.s_maxFrameFixes;
    bf_trajectory_manager* tm = p->ob->tm;
    uint32_t active;
    BF_TRY(bf_trajectory_manager_get_num_active_operations(tm, &active));
    if (active < maxPerFrameFixes) BF_TRY(bf_trajectory_manager_generate_update_lists(tm));
    for (uint32_t fixes = 0; fixes < maxPerFrameFixes; ++fixes) {
        float newT[16], oldT[16]; uint32_t frameIdx = 0xFFFFFFFFu; int found = 0;
        BF_TRY(bf_trajectory_manager_get_top_from_deintegrate_list(tm, oldT, &frameIdx, &found));
        if (found) { BF_TRY(plIntegrate(p, frameIdx, oldT, true)); continue; }
        BF_TRY(bf_trajectory_manager_get_top_from_integrate_list(tm, newT, &frameIdx, &found));
        if (found) { BF_TRY(plIntegrate(p, frameIdx, newT, false)); BF_TRY(bf_trajectory_manager_confirm_integration(tm, frameIdx)); continue; }
        BF_TRY(bf_trajectory_manager_get_top_from_reintegrate_list(tm, oldT, newT, &frameIdx, &found));
        if (found) {
            if (newT[0] == NINF) continue;          // every candidate was invalidated meanwhile: no volume operation now; get_top re-typed them Integrated, so the next list update de-integrates them at their old poses
            if (p->gas.s_integrationEnabled) {          // deIntegrate(old) + integrate(new) (:885-886) as one fused pass over the volume
                BF_TRY(volPost(p, 2, frameIdx, oldT, newT, -1));
                p->numDeIntegrate++; p->numIntegrate++;
            }
            BF_TRY(bf_trajectory_manager_confirm_integration(tm, frameIdx));
            continue;
        }
        break;
    }
    // the frame boundary closes the volume thread's batch: the garbage collection does, or (collection disabled) a flush command - a batch is one frame's operators
    // whatever the settings, and nothing stays pending across frames
    BF_TRY(volPost(p, p->gas.s_garbageCollectionEnabled ? 3 : 4, 0, nullptr, nullptr, -1));
    return BF_OK;
}

// everything of the frame loop after the ingest, for frame `frame` (got: a new frame, as opposed to an iteration after the
// sequence ended), in two parts so that a caller can put host work between the enqueue of processInput and its read-back
//                                                                                                   :966-1095 (serial branch)
inline double plNow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int plBodyBegin(bf_pipeline* p, uint32_t frame) {
    // ---- processInput: enqueue (bundling stream) ...
    const double t0 = plNow();
    bf_pipeline::TraceRec* tr = p->tracePath.empty() ? nullptr : p->traceOf(frame);
    if (tr) { tr->hEnq0 = t0; (void)hipEventRecord(tr->gChain0, p->sBundle); }
    const int rc = bf_online_bundler_process_input_begin_frame(p->ob, frame);
    if (tr) { (void)hipEventRecord(tr->gChain1, p->sBundle); tr->hEnq1 = plNow(); }
    p->hostProfile[0] += plNow() - t0;
    return rc;
}

int plBodyRest(bf_pipeline* p, uint32_t frame, bool got) {
    hipStream_t sa = p->sBundle, sv = p->sVolume;
    const bool tm = p->timings;
    const int evSlot = got ? (int)(frame % bf_pipeline::NEV) : -1;
    if (got) BF_TRY(bf_online_bundler_apply_lagged_solve(p->ob, frame));      // lagged mode: the frame that sees an earlier chunk's optimised poses
    // ---- fix old frames (volume stream; launches issued by the volume thread), concurrently with the feature pipeline
    double t0 = plNow();
    if (tm) (void)hipEventRecord(p->ev[4], sv);
    BF_TRY(plReintegrate(p));
    if (tm) (void)hipEventRecord(p->ev[5], sv);
    double t1 = plNow(); p->hostProfile[2] += t1 - t0; t0 = t1;
    // ---- ... and its read-back
    bf_pipeline::TraceRec* tr = p->tracePath.empty() ? nullptr : p->traceOf(frame);
    if (tr) tr->hWait0 = plNow();
    BF_TRY(bf_online_bundler_process_input_end(p->ob));
    t1 = plNow(); p->hostProfile[3] += t1 - t0; t0 = t1;
    if (tr) tr->hWait1 = t1;
    if (tm) (void)hipEventRecord(p->ev[2], sa);
    // ---- reconstruction of the current frame (volume stream, after this frame's ingest)
    if (tm) (void)hipEventRecord(p->ev[6], sv);
    if (got) {
        float T[16]; uint32_t frameIdx = 0; int lost = 0, valid = 0;
        BF_TRY(bf_online_bundler_get_current_integration_frame(p->ob, T, &frameIdx, &lost, &valid));
        if (valid && p->gas.s_reconstructionEnabled) {
            BF_TRY(plIntegrate(p, frameIdx, T, false, evSlot));
            BF_TRY(bf_trajectory_manager_add_frame(p->ob->tm, BF_TF_INTEGRATED, T, frame));
        } else {
            const m44 inv = minfM();
            BF_TRY(bf_trajectory_manager_add_frame(p->ob->tm, BF_TF_NOT_INTEGRATED_NO_TRANSFORM, inv.e, frame));
        }
    }
    if (tm) (void)hipEventRecord(p->ev[7], sv);
    t1 = plNow(); p->hostProfile[4] += t1 - t0; t0 = t1;
    // ---- bundling optimisation (bundling stream)
    BF_TRY(bf_online_bundler_process_frame(p->ob, frame, p->gbs.s_numLocalNonLinIterations, p->gbs.s_numLocalLinIterations, p->ob->gbs.s_numGlobalNonLinIterations,
                                           p->gbs.s_numGlobalLinIterations));
    p->hostProfile[5] += plNow() - t0;
    if (got) p->hostProfile[7] += 1.0;
    if (tr) tr->hBody1 = plNow();
    return BF_OK;
}

int plBody(bf_pipeline* p, uint32_t frame, bool got) {
    BF_TRY(plBodyBegin(p, frame));
    return plBodyRest(p, frame, got);
}

// does the body of `frame` have to run before the next frame's chain may be enqueued?  Serial order: yes for a chunk's last frame (its solves write the
// trajectory and the last valid transform the next chain's pose kernel reads, and hand the optimiser's bundler back).  Lagged mode: never.
bool plBodyBeforeNextChain(const bf_pipeline* p, uint32_t frame) {
    return p->ob->solveLag == 0 && p->ob->isLastLocalFrame(frame);
}

int plRestFront(bf_pipeline* p) {
    const uint32_t f = p->begun.front();
    p->begun.pop_front();
    return plBodyRest(p, f, true);
}

// run the bodies of all frames that are still waiting for theirs, in stream order
int plFlush(bf_pipeline* p) {
    while (!p->begun.empty()) BF_TRY(plRestFront(p));
    if (p->deferred >= 0) {
        const uint32_t f = (uint32_t)p->deferred;
        p->deferred = -1;
        BF_TRY(plBody(p, f, true));
    }
    return bf_online_bundler_wait_solves(p->ob);          // a lagged solve has finished (it is APPLIED at its frame, not here)
}

int plFrame(bf_pipeline* p, const float* depth, const uint8_t* color, bool device, bool haveInput, int* gotFrame) {
    hipStream_t sa = p->sBundle, sd = p->sIngest;
    const bool tm = p->timings;
    const bool ahead = p->lookahead && !tm && haveInput;
    if (!ahead) BF_TRY(plFlush(p));
    // ---- the detected frame's matching chain onto the bundling stream (behind the chain of the frame before it, whose result is still outstanding) ...
    if (p->deferred >= 0) {
        if (!p->begun.empty() && plBodyBeforeNextChain(p, p->begun.back())) { while (!p->begun.empty()) BF_TRY(plRestFront(p)); }
        const uint32_t f = (uint32_t)p->deferred;
        p->deferred = -1;
        BF_TRY(plBodyBegin(p, f));
        p->begun.push_back(f);
    }
    // ---- read input (detect stream)
    const double tIn = plNow();
    if (tm) (void)hipEventRecord(p->ev[0], sd);
    int got = 0;
    if (haveInput) BF_TRY(device ? bf_image_manager_process_device(p->im, depth, color, &got) : bf_image_manager_process(p->im, depth, color, &got));
    const uint32_t frame = p->im->currFrame > 0 ? p->im->currFrame - 1 : 0;
    if (got) BF_HIP_TRY(hipEventRecord(p->evIngest[frame % bf_pipeline::NEV], sd));
    if (got && !ahead) BF_HIP_TRY(hipStreamWaitEvent(sa, p->evIngest[frame % bf_pipeline::NEV], 0));      // the frame is detected on the bundling stream, from the ingest buffers
    if (tm) { (void)hipEventRecord(p->ev[1], sd); (void)hipEventRecord(p->ev[8], sa); }
    if (got && !p->tracePath.empty() && p->trace.size() < 4096) {
        bf_pipeline::TraceRec r; memset(&r, 0, sizeof r); r.frame = frame; r.hDet0 = tIn;
        (void)hipEventCreate(&r.gDet); (void)hipEventCreate(&r.gChain0); (void)hipEventCreate(&r.gChain1);
        if (!p->traceBase) { (void)hipEventCreate(&p->traceBase); (void)hipEventRecord(p->traceBase, p->sDetect); (void)hipEventSynchronize(p->traceBase); p->traceBaseHost = plNow(); }
        p->trace.push_back(r);
    }
    if (got && ahead) BF_TRY(bf_online_bundler_detect_ahead_after(p->ob, p->evIngest[frame % bf_pipeline::NEV]));
    if (got && !p->tracePath.empty()) { bf_pipeline::TraceRec* tr = p->traceOf(frame); if (tr) { if (ahead) (void)hipEventRecord(tr->gDet, (p->sDetect2 && (frame & 1u)) ? p->sDetect2 : p->sDetect); tr->hDet1 = plNow(); } }
    p->hostProfile[1] += plNow() - tIn;
    // ---- ... and the body of the oldest frame in flight, whose chain was enqueued by the previous call
    while (p->begun.size() + 1 > p->depth) BF_TRY(plRestFront(p));
    if (ahead && got) p->deferred = (int)frame;
    else if (p->im->currFrame > 0) {
        // no new frame although one was offered (the image manager is full, CUDAImageManager.cpp:22-35): `frame` is the last accepted frame, whose chain was
        // enqueued above and whose body has not run - every frame in flight completes first, then this call is the iteration past the end of the sequence
        BF_TRY(plFlush(p));
        BF_TRY(plBody(p, frame, got != 0));
        if (got) BF_HIP_TRY(hipStreamSynchronize(sa));      // (the detection read the frame's input set on the bundling stream: done before the set's next ingest, two frames on)
    }
    if (tm) {
        (void)hipEventRecord(p->ev[3], sa);
        (void)hipEventSynchronize(p->ev[3]);
        (void)hipEventSynchronize(p->ev[7]);
        memset(&p->last, 0, sizeof p->last);
        (void)hipEventElapsedTime(&p->last.timeSensorProcess, p->ev[0], p->ev[1]);
        (void)hipEventElapsedTime(&p->last.timeSiftDetection, p->ev[8], p->ev[2]);        // SIFT + cache + match + filters + read-back
        (void)hipEventElapsedTime(&p->last.timeSolve, p->ev[2], p->ev[3]);
        (void)hipEventElapsedTime(&p->last.timeReIntegrate, p->ev[4], p->ev[5]);
        (void)hipEventElapsedTime(&p->last.timeReconstruct, p->ev[6], p->ev[7]);
        (void)hipEventElapsedTime(&p->last.timeTotal, p->ev[0], p->ev[3]);
    }
    // the caller's depth / colour buffers (host or device) are free for reuse when this returns, like after the reference's
    // synchronous CUDAImageManager::process: the ingest ran beside the previous frame's body, so this wait is normally over already
    if (got) { const double tw = plNow(); BF_HIP_TRY(hipEventSynchronize(p->evIngest[frame % bf_pipeline::NEV])); p->hostProfile[6] += plNow() - tw; }
    if (gotFrame) *gotFrame = got;
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_pipeline_create(const bf_global_app_state* gas, const bf_global_bundling_state* gbs, const bf_rgbd_sensor_desc* sensor, bf_pipeline** out) {
    BF_REQUIRE(gas && gbs && sensor && out, "null argument");
    BF_REQUIRE(!gas->s_streamingEnabled, "chunk streaming is not part of the BundleFusion path (zParametersDefault.txt:100)");
    bf_pipeline* p = new bf_pipeline;
    p->gas = *gas; p->gbs = *gbs; p->sensor = *sensor;
    memset(&p->last, 0, sizeof p->last);
    int rc = bf_image_manager_create(gas->s_integrationWidth, gas->s_integrationHeight, gbs->s_widthSIFT, gbs->s_heightSIFT, sensor, gbs, 1, &p->im);
    // (bf_image_manager_set_store_texels - a third plane per stored frame, depth and colour interleaved for the fast voxel update - is not used here: measured no
    // gain, 680 vs 685 frames/s, gpurun r04l; since round 5 the batch's march writes the texels of its operators.)
    if (!rc) rc = bf_online_bundler_create(sensor, p->im, gas, gbs, &p->ob);
    bf_hash_params hp;                                          // CUDASceneRepHashSDF::parametersFromGlobalAppState :39-59
    memset(&hp, 0, sizeof hp);
    const m44 I = identity44();
    memcpy(hp.m_rigidTransform, I.e, 64); memcpy(hp.m_rigidTransformInverse, I.e, 64);
    hp.m_hashNumBuckets = gas->s_hashNumBuckets; hp.m_hashBucketSize = 4; hp.m_hashMaxCollisionLinkedListSize = gas->s_hashMaxCollisionLinkedListSize;
    hp.m_SDFBlockSize = 8; hp.m_numSDFBlocks = gas->s_hashNumSDFBlocks; hp.m_virtualVoxelSize = gas->s_SDFVoxelSize;
    hp.m_maxIntegrationDistance = gas->s_SDFMaxIntegrationDistance; hp.m_truncation = gas->s_SDFTruncation; hp.m_truncScale = gas->s_SDFTruncationScale;
    hp.m_integrationWeightSample = gas->s_SDFIntegrationWeightSample; hp.m_integrationWeightMax = gas->s_SDFIntegrationWeightMax;
    for (int i = 0; i < 3; ++i) { hp.m_streamingVoxelExtents[i] = gas->s_streamingVoxelExtents[i]; hp.m_streamingGridDimensions[i] = gas->s_streamingGridDimensions[i]; hp.m_streamingMinGridPos[i] = gas->s_streamingMinGridPos[i]; }
    hp.m_streamingInitialChunkListSize = gas->s_streamingInitialChunkListSize;
    if (!rc) rc = bf_scene_create(&hp, &p->scene);
    if (rc) { bf_pipeline_destroy(p); return rc; }
    p->cam.fx = p->im->depthIntrinsics.e[0]; p->cam.fy = p->im->depthIntrinsics.e[5]; p->cam.mx = p->im->depthIntrinsics.e[2]; p->cam.my = p->im->depthIntrinsics.e[6];
    p->cam.m_sensorDepthWorldMin = gas->s_renderDepthMin; p->cam.m_sensorDepthWorldMax = gas->s_renderDepthMax;      // DepthSensing.cpp:636-643
    p->cam.m_imageWidth = gas->s_integrationWidth; p->cam.m_imageHeight = gas->s_integrationHeight;
    for (auto& e : p->ev) BF_HIP_TRY(hipEventCreate(&e));
    for (auto& e : p->evIngest) BF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BF_HIP_TRY(hipDeviceSynchronize());                                  // creation-time work was issued on the null stream
    {   // the feature pipeline is a chain of small latency-bound kernels with a host read-back at its end: it gets priority over
        // the wide, throughput-bound voxel updates of the volume stream
        int least = 0, greatest = 0;
        BF_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sBundle, hipStreamNonBlocking, greatest));
        // The volume stream may use all but `reserve` compute units (BF_VOLUME_CU_RESERVE, 0: the whole device at the lowest queue priority): the matching chain is
        // ~20 dependent small launches, and behind a batched voxel update that fills every CU each of them waits for waves to retire.  With the frame's operators
        // batched the volume queue is busy about half of the time, so the slower update costs nothing: 873 / 893 / 939 / 944 frames/s at R = 0 / 16 / 32 / 64, the
        // wait for the match result 0.65 -> 0.49 ms (gpurun r06a, one box).  (Round 3, one launch chain per operator and the volume stream the bottleneck: no gain.)
        uint32_t reserve = 32;
        if (const char* e = getenv("BF_VOLUME_CU_RESERVE")) reserve = (uint32_t)std::max(atoi(e), 0);
        if (reserve) {
            hipDeviceProp_t prop; int devId = 0;
            BF_HIP_TRY(hipGetDevice(&devId));
            BF_HIP_TRY(hipGetDeviceProperties(&prop, devId));
            const uint32_t ncu = (uint32_t)prop.multiProcessorCount, use = ncu > reserve ? ncu - reserve : 1u;
            std::vector<uint32_t> mask((ncu + 31u) / 32u, 0u);
            for (uint32_t i = 0; i < use; ++i) mask[i >> 5] |= 1u << (i & 31u);
            BF_HIP_TRY(hipExtStreamCreateWithCUMask(&p->sVolume, (uint32_t)mask.size(), mask.data()));
        } else
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sVolume, hipStreamNonBlocking, least));
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sDetect, hipStreamNonBlocking, greatest));
        // The HIP runtime maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, four by default) by priority class and creation order, and streams
        // that share a queue serialise.  Which streams exist - even idle ones - therefore decides the frame rate: measured on one box (gpurun r04i - r04k) 710
        // frames/s with exactly this set in exactly this order (allocation [the scene's], bundling, volume, detection, solve, ingest, pair x 2), 440 - 660 with
        // any other set tried (a second preparation stream, no solve / pair streams, the ingest on the bundling stream, 2 - 16 queues requested).  Round 6, with the
        // volume on a masked stream and the lagged schedule: eight creation orders of the five streams below gave 937 - 1037 frames/s, this one 985 - 1005 (gpurun r06m / r06n:
        // within the spread; whichever streams share a queue, the waits move and their sum stays - profiles/r06_loop_schedule.md).
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sSolve, hipStreamNonBlocking, greatest));
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sIngest, hipStreamNonBlocking, greatest));
        for (auto& st : p->sPair) BF_HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
    }
    if (const char* e = getenv("BF_PIPELINE_LOOKAHEAD")) p->lookahead = atoi(e) != 0;
    if (const char* e = getenv("BF_PIPELINE_DEPTH")) p->depth = (uint32_t)std::min(std::max(atoi(e), 2), std::min<int>(bf_online_bundler::PEND, (int)bf_online_bundler::STAGE));
    BF_TRY(bf_image_manager_set_stream(p->im, p->sIngest));
    BF_TRY(bf_online_bundler_set_stream(p->ob, p->sBundle));
    BF_TRY(bf_online_bundler_set_detect_stream(p->ob, p->sDetect));
    static_assert(bf_image_manager::NSETS == bf_online_bundler::STAGE, "the image manager's input sets and the staging slots are indexed alike");
    for (uint32_t k = 0; k < bf_image_manager::NSETS; ++k) BF_TRY(bf_image_manager_set_input_guard(p->im, k, p->ob->evDetect[k]));      // the staged detection of the frame that used the set last
    {
        // odd frames detect on a second queue (bf_online_bundler_set_second_detect_stream): the first of the two pair streams.  (The pair stages of consecutive frames
        // on those two streams - bf_online_bundler_set_pair_streams, round 4 - lost: 659 vs 697 frames/s, gpurun r04c; the mode is reachable through that call only.)
        BF_TRY(bf_online_bundler_set_second_detect_stream(p->ob, p->sPair[0])); p->sDetect2 = p->sPair[0];
    }
    BF_TRY(bf_scene_set_stream(p->scene, p->sVolume));
    BF_TRY(bf_scene_set_overlap(p->scene, 1));        // frames are ordered against the volume by evIngest / host synchronisation
    {   // Default schedule (round 6): the reference's - its optimiser runs beside the frame loop on a second thread (FriedLiver.cpp:112-143) - made deterministic: the
        // chunk solves on their own thread and stream, applied exactly `lag` frames behind the frame that closed the chunk.  bf_pipeline_set_solve_lag(p, 0) /
        // BF_PIPELINE_SOLVE_LAG=0: the serial order of DepthSensing.cpp's single-threaded branch (what the oracle loop and the compiled reference loop run).
        uint32_t lag = std::min<uint32_t>(10u, p->ob->submapSize);
        if (const char* e = getenv("BF_PIPELINE_SOLVE_LAG")) lag = (uint32_t)std::max(atoi(e), 0);
        if (lag && p->lookahead && lag < p->depth) lag = 0;
        BF_TRY(bf_pipeline_set_solve_lag(p, lag));
    }
    if (const char* e = getenv("BF_PIPELINE_TRACE")) p->tracePath = e;
    int dev = 0;
    BF_HIP_TRY(hipGetDevice(&dev));
    p->worker = std::thread([p, dev] { (void)hipSetDevice(dev); volWorker(p); });
    *out = p;
    return BF_OK;
}

int bf_pipeline_destroy(bf_pipeline* p) {
    if (!p) return BF_OK;
    if (p->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
        p->cvWork.notify_all();
        p->worker.join();
    }
    (void)hipDeviceSynchronize();
    if (!p->tracePath.empty() && !p->trace.empty()) {
        if (FILE* f = fopen(p->tracePath.c_str(), "a")) {
            fprintf(f, "# frame | host: detect enqueue (begin end) chain enqueue (begin end) wait for the result (begin end) body end | GPU: detection end, chain begin, chain end   [ms since the first traced frame]\n");
            for (auto& r : p->trace) {
                float gd = -1.0f, g0 = -1.0f, g1 = -1.0f;
                if (hipEventQuery(r.gDet) == hipSuccess) (void)hipEventElapsedTime(&gd, p->traceBase, r.gDet);
                if (r.hEnq0 > 0.0) { (void)hipEventElapsedTime(&g0, p->traceBase, r.gChain0); (void)hipEventElapsedTime(&g1, p->traceBase, r.gChain1); }
                const double b = p->traceBaseHost;
                fprintf(f, "%5u | %8.3f %8.3f  %8.3f %8.3f  %8.3f %8.3f  %8.3f | %8.3f %8.3f %8.3f\n", r.frame, 1e3 * (r.hDet0 - b), 1e3 * (r.hDet1 - b), 1e3 * (r.hEnq0 - b), 1e3 * (r.hEnq1 - b),
                        1e3 * (r.hWait0 - b), 1e3 * (r.hWait1 - b), 1e3 * (r.hBody1 - b), gd, g0, g1);
                (void)hipEventDestroy(r.gDet); (void)hipEventDestroy(r.gChain0); (void)hipEventDestroy(r.gChain1);
            }
            fclose(f);
        }
    }
    bf_online_bundler_destroy(p->ob); bf_image_manager_destroy(p->im); bf_scene_destroy(p->scene);
    for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : p->evIngest) if (e) (void)hipEventDestroy(e);
    if (p->sBundle) (void)hipStreamDestroy(p->sBundle);
    if (p->sVolume) (void)hipStreamDestroy(p->sVolume);
    if (p->sDetect) (void)hipStreamDestroy(p->sDetect);
    if (p->sSolve) (void)hipStreamDestroy(p->sSolve);
    if (p->sIngest) (void)hipStreamDestroy(p->sIngest);
    for (auto st : p->sPair) if (st) (void)hipStreamDestroy(st);
    delete p;
    return BF_OK;
}

// Lagged solve (bf_online_bundler_set_solve_lag): the chunk solves leave the frame loop's critical path - they run on their own thread and stream and
// their results are applied exactly `lag` frames after the frame that closed the chunk (1 <= lag <= s_submapSize; 0 = the reference's serial order,
// the default).  The reference's own multi-threaded build runs them on a separate thread with no defined hand-over point (FriedLiver.cpp:112-143).
int bf_pipeline_set_solve_lag(bf_pipeline* p, uint32_t lag) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_REQUIRE(!p->ob->job.active, "a lagged solve is still waiting for its frame");
    // The call that delivers frame n enqueues the chain of frame n - 1 BEFORE it runs the body of frame n - depth (where a chunk's job gets its frame of
    // application): a lag below the depth would ask for a chain that is already enqueued to see the result.  (lag >= depth: exactly `lag` frames, as the oracle
    // loop under the same lag - tests/test_pipeline_gpu.py, lag = depth included.)
    BF_REQUIRE(lag == 0 || !p->lookahead || lag >= p->depth, "the solve lag must be 0 or at least the frame loop's depth (BF_PIPELINE_DEPTH, default 2)");
    if (lag && !p->sSolve) {
        // Not the lowest priority: measured (gpurun r04b) the cooperative PCG - up to 64 workgroups meeting at a grid barrier 450 times per global solve - is
        // then starved by the volume stream's wide launches and a chunk's solves take longer than the ten frames they have
        int least = 0, greatest = 0;
        BF_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        BF_HIP_TRY(hipStreamCreateWithPriority(&p->sSolve, hipStreamNonBlocking, greatest));
    }
    return bf_online_bundler_set_solve_lag(p->ob, lag, p->sSolve);
}
int bf_pipeline_get_solve_lag(bf_pipeline* p, uint32_t* lag) { BF_REQUIRE(p && lag, "null argument"); return bf_online_bundler_get_solve_lag(p->ob, lag); }

int bf_pipeline_set_comm(bf_pipeline* p, bf_comm* comm, uint32_t capacity_keys) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    return bf_scene_set_alloc_comm(p->scene, comm, capacity_keys);
}

// Batched volume operators on / off (on by default; see bf_pipeline::volBatching).  Off: every operator is issued on its own (bf_scene_integrate / _deintegrate /
// _reintegrate), as before round 5 - same volume either way.
int bf_pipeline_set_volume_batching(bf_pipeline* p, int enable) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    p->volBatching = enable != 0;
    return BF_OK;
}

int bf_pipeline_set_volume_shard(bf_pipeline* p, uint32_t rank, uint32_t world) {
    BF_REQUIRE(p, "null pipeline");
    return bf_scene_set_shard(p->scene, rank, world);
}

int bf_pipeline_process_frame(bf_pipeline* p, const float* h_depth, const uint8_t* h_color, int* gotFrame) {
    BF_REQUIRE(p, "null pipeline");
    return plFrame(p, h_depth, h_color, false, true, gotFrame);
}
int bf_pipeline_process_frame_device(bf_pipeline* p, const float* d_depth, const uint8_t* d_color, int* gotFrame) {
    BF_REQUIRE(p, "null pipeline");
    return plFrame(p, d_depth, d_color, true, true, gotFrame);
}
int bf_pipeline_process_end_of_sequence(bf_pipeline* p, uint32_t* numActiveOperations) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFrame(p, nullptr, nullptr, false, false, nullptr));
    if (numActiveOperations) BF_TRY(bf_trajectory_manager_get_num_active_operations(p->ob->tm, numActiveOperations));
    return BF_OK;
}
int bf_pipeline_synchronize(bf_pipeline* p) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    BF_HIP_TRY(hipStreamSynchronize(p->sIngest));
    BF_HIP_TRY(hipStreamSynchronize(p->sDetect));
    for (auto st : p->sPair) if (st) BF_HIP_TRY(hipStreamSynchronize(st));
    BF_HIP_TRY(hipStreamSynchronize(p->sBundle));
    if (p->sSolve) BF_HIP_TRY(hipStreamSynchronize(p->sSolve));
    BF_HIP_TRY(hipStreamSynchronize(p->sVolume));
    return BF_OK;
}
int bf_pipeline_get_scene(bf_pipeline* p, bf_scene** out) {         // the volume thread is drained first: the caller may use the scene directly
    BF_REQUIRE(p && out, "null argument");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    *out = p->scene;
    return BF_OK;
}
int bf_pipeline_get_image_manager(bf_pipeline* p, bf_image_manager** out) { BF_REQUIRE(p && out, "null argument"); *out = p->im; return BF_OK; }
int bf_pipeline_get_online_bundler(bf_pipeline* p, bf_online_bundler** out) {      // every frame handed in so far has been processed when this returns
    BF_REQUIRE(p && out, "null argument");
    BF_TRY(plFlush(p));
    *out = p->ob;
    return BF_OK;
}
int bf_pipeline_get_num_frames(bf_pipeline* p, uint32_t* out) { BF_REQUIRE(p && out, "null argument"); *out = p->im->currFrame; return BF_OK; }
int bf_pipeline_get_integrated_trajectory(bf_pipeline* p, float* h_out, uint32_t capacity, uint32_t* count) {
    BF_REQUIRE(p && h_out && count, "null argument");
    BF_TRY(plFlush(p));
    const uint32_t n = std::min(p->ob->tm->numAddedFrames, capacity);
    for (uint32_t i = 0; i < n; ++i) {
        const auto& f = p->ob->tm->frames[i];
        const bool in = f.type == BF_TF_INTEGRATED || f.type == BF_TF_REINTEGRATION;
        const m44 T = in ? f.integratedTransform : minfM();
        memcpy(h_out + 16 * (size_t)i, T.e, 64);
    }
    *count = n;
    return BF_OK;
}
int bf_pipeline_get_counters(bf_pipeline* p, uint32_t* numIntegrate, uint32_t* numDeIntegrate, uint32_t* numLocalSolves, uint32_t* numGlobalSolves) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    if (numIntegrate) *numIntegrate = p->numIntegrate;
    if (numDeIntegrate) *numDeIntegrate = p->numDeIntegrate;
    if (numLocalSolves) *numLocalSolves = p->ob->numLocalSolves;
    if (numGlobalSolves) *numGlobalSolves = p->ob->numGlobalSolves;
    return BF_OK;
}
int bf_pipeline_get_host_profile(bf_pipeline* p, double out[8], int reset) {
    BF_REQUIRE(p && out, "null argument");
    BF_TRY(plFlush(p));
    for (int i = 0; i < 8; ++i) out[i] = p->hostProfile[i];
    if (reset) for (double& v : p->hostProfile) v = 0.0;
    return BF_OK;
}
// where the volume thread's time goes: seconds it spent inside the operators' HIP API calls and the number of operators it issued since the last reset -
// if busy / wall is near 1, the thread's launch rate, not the GPU, paces the volume
int bf_pipeline_get_volume_thread_profile(bf_pipeline* p, double* busySeconds, double* commands, int reset) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    std::lock_guard<std::mutex> lk(p->mu);
    if (busySeconds) *busySeconds = p->volBusy;
    if (commands) *commands = p->volCommands;
    if (reset) { p->volBusy = 0.0; p->volCommands = 0.0; }
    return BF_OK;
}
int bf_pipeline_enable_timings(bf_pipeline* p, int enable) {
    BF_REQUIRE(p, "null pipeline");
    BF_TRY(plFlush(p));
    BF_TRY(volDrain(p));
    p->timings = enable != 0;
    return BF_OK;
}
int bf_pipeline_get_last_timing(bf_pipeline* p, bf_frame_timing* out) { BF_REQUIRE(p && out, "null argument"); *out = p->last; return BF_OK; }

}  // extern "C"

// ================================================================================================ chunk-parallel bundling
// Stage A — bf_chunk_worker: the chunk-local half of OnlineBundler for ONE local chunk.  Nothing here reads global state: the
// frames of a chunk are ingested, detected, matched and filtered against each other (Bundler::matchAndFilter on m_local), the
// chunk is solved (Bundler::optimize on m_optLocal) and fused into one key frame (fuseToGlobal).  The result is a flat host
// package (bf_chunk_header + payload) that any rank can feed to bf_pipeline_process_frame_chunked.
struct bf_chunk_worker {
    bf_global_app_state gas; bf_global_bundling_state gbs; bf_rgbd_sensor_desc sensor;
    bf_image_manager* im = nullptr;       // ingest only (scratch mode: no frame history)
    bf_bundler* local = nullptr;          // the chunk: s_submapSize + 1 images
    bf_siftmgr* fuseMgr = nullptr;        // one-image target of fuseToGlobal
    float *d_intensity = nullptr, *d_intensityHelper = nullptr;
    bf_chunk_frame_record* d_rec = nullptr;
    m44 siftIntrinsics, siftIntrinsicsInv;
    hipStream_t stream = nullptr;
    uint32_t cacheW = 0, cacheH = 0;
    uint64_t packageBytes = 0, offKeys = 0, offDescs = 0, offCache[6] = {0, 0, 0, 0, 0, 0};
};

extern "C" {

int bf_chunk_worker_create(const bf_global_app_state* gas, const bf_global_bundling_state* gbs, const bf_rgbd_sensor_desc* sensor, bf_chunk_worker** out) {
    BF_REQUIRE(gas && gbs && sensor && out, "null argument");
    BF_REQUIRE(gbs->s_submapSize + 1 <= BF_CHUNK_MAX_FRAMES, "s_submapSize + 1 exceeds BF_CHUNK_MAX_FRAMES");
    bf_chunk_worker* w = new bf_chunk_worker;
    w->gas = *gas; w->gbs = *gbs; w->sensor = *sensor;
    w->siftIntrinsics = scaleIntrinsics(sensor->colorIntrinsics, gbs->s_widthSIFT, gbs->s_heightSIFT, sensor->colorWidth, sensor->colorHeight);
    w->siftIntrinsicsInv = inverse44(w->siftIntrinsics);
    int rc = bf_image_manager_create(gas->s_integrationWidth, gas->s_integrationHeight, gbs->s_widthSIFT, gbs->s_heightSIFT, sensor, gbs, 2, &w->im);
    if (!rc) rc = bf_bundler_create(gbs->s_submapSize + 1, gbs->s_maxNumKeysPerImage, w->siftIntrinsicsInv.e, w->im, 1, gas, gbs, &w->local);
    if (!rc) rc = bf_siftmgr_create(2, gbs->s_maxNumKeysPerImage, &w->fuseMgr);
    if (rc) { bf_chunk_worker_destroy(w); return rc; }
    const size_t nS = (size_t)gbs->s_widthSIFT * gbs->s_heightSIFT;
    BF_HIP_TRY(BF_MALLOC((void**)&w->d_intensity, nS * 4));
    BF_HIP_TRY(BF_MALLOC((void**)&w->d_intensityHelper, nS * 4));
    BF_HIP_TRY(BF_MALLOC((void**)&w->d_rec, sizeof(bf_chunk_frame_record) * BF_CHUNK_MAX_FRAMES));
    float k4[4];
    BF_TRY(bf_cache_get_geometry(w->local->cache, &w->cacheW, &w->cacheH, k4));
    const uint64_t n = (uint64_t)w->cacheW * w->cacheH, mk = gbs->s_maxNumKeysPerImage;
    uint64_t off = (sizeof(bf_chunk_header) + 255) / 256 * 256;
    w->offKeys = off; off += mk * sizeof(bf_sift_keypoint);
    w->offDescs = off; off += mk * 128;
    const uint64_t cb[6] = {n * 4, n * 16, n * 4, n * 8, n * 4, n * 16};
    for (int k = 0; k < 6; ++k) { w->offCache[k] = off; off += cb[k]; }
    w->packageBytes = (off + 255) / 256 * 256;
    BF_HIP_TRY(hipDeviceSynchronize());
    *out = w;
    return BF_OK;
}

int bf_chunk_worker_destroy(bf_chunk_worker* w) {
    if (!w) return BF_OK;
    (void)hipDeviceSynchronize();
    bf_bundler_destroy(w->local); bf_siftmgr_destroy(w->fuseMgr); bf_image_manager_destroy(w->im);
    (void)hipFree(w->d_intensity); (void)hipFree(w->d_intensityHelper); (void)hipFree(w->d_rec);
    delete w;
    return BF_OK;
}

int bf_chunk_worker_set_stream(bf_chunk_worker* w, void* s) {
    BF_REQUIRE(w, "null worker");
    w->stream = (hipStream_t)s;
    BF_TRY(bf_image_manager_set_stream(w->im, s));
    BF_TRY(bf_bundler_set_stream(w->local, s));
    return bf_siftmgr_set_stream(w->fuseMgr, s);
}

int bf_chunk_worker_package_bytes(bf_chunk_worker* w, uint64_t* bytes) { BF_REQUIRE(w && bytes, "null argument"); *bytes = w->packageBytes; return BF_OK; }

int bf_chunk_worker_run(bf_chunk_worker* w, uint32_t chunkIndex, uint32_t numFrames, const float* const* d_depth, const uint8_t* const* d_color, void* h_package) {
    BF_REQUIRE(w && d_depth && d_color && h_package, "null argument");
    BF_REQUIRE(numFrames >= 2 && numFrames <= w->gbs.s_submapSize + 1, "a chunk holds 2 .. s_submapSize + 1 frames");
    const bf_rgbd_sensor_desc& sn = w->sensor;
    hipStream_t st = w->stream;
    bf_chunk_header* h = reinterpret_cast<bf_chunk_header*>(h_package);
    uint8_t* base = reinterpret_cast<uint8_t*>(h_package);
    memset(h, 0, sizeof *h);
    h->magic = BF_CHUNK_MAGIC; h->chunkIndex = chunkIndex; h->numFrames = numFrames; h->submapSize = w->gbs.s_submapSize;
    h->maxKeys = w->gbs.s_maxNumKeysPerImage; h->cacheWidth = w->cacheW; h->cacheHeight = w->cacheH;
    h->offKeys = w->offKeys; h->offDescs = w->offDescs; for (int k = 0; k < 6; ++k) h->offCache[k] = w->offCache[k];
    h->totalBytes = w->packageBytes;
    BF_HIP_TRY(hipMemsetAsync(w->d_rec, 0, sizeof(bf_chunk_frame_record) * BF_CHUNK_MAX_FRAMES, st));
    int validFlags[BF_CHUNK_MAX_FRAMES] = {0};
    for (uint32_t j = 0; j < numFrames; ++j) {
        int got = 0;
        BF_TRY(bf_image_manager_process_device(w->im, d_depth[j], d_color[j], &got));
        BF_REQUIRE(got, "chunk worker: ingest refused a frame");
        // getCurrentFrame (OnlineBundler.cpp:106-116), detectFeatures, storeCachedFrame (:203-206)
        BF_TRY(bf_image_resample_to_intensity(w->d_intensity, w->gbs.s_widthSIFT, w->gbs.s_heightSIFT, w->im->d_colorInput, sn.colorWidth, sn.colorHeight, st));
        if (w->gas.s_colorFilter) {
            BF_TRY(bf_image_gauss_filter_intensity(w->d_intensityHelper, w->d_intensity, w->gas.s_colorSigmaD, w->gbs.s_widthSIFT, w->gbs.s_heightSIFT, st));
            std::swap(w->d_intensityHelper, w->d_intensity);
        }
        BF_TRY(bf_bundler_detect_features(w->local, w->d_intensity, w->im->d_depthInputFiltered));
        BF_TRY(bf_bundler_store_cached_frame(w->local, sn.depthWidth, sn.depthHeight, w->im->d_colorInput, sn.colorWidth, sn.colorHeight, w->im->d_depthInputRaw));
        if (j > 0) {                                            // matchAndFilter + the part of computeCurrentSiftTransform that is chunk-local (:118-132)
            uint32_t cur, start, num;
            BF_TRY(matchAndFilterEnqueue(w->local, cur, start, num));
            const float* d_Tinv = nullptr; const int32_t* d_nf = nullptr;
            BF_TRY(bf_bundler_get_current_sift_transforms_gpu(w->local, &d_Tinv));
            BF_TRY(bf_bundler_get_num_filt_matches_gpu(w->local, &d_nf));
            k_pick_relative<<<1, 1, 0, st>>>(j, d_nf, (const m44*)d_Tinv, w->d_rec + j);
            BF_HIP_TRY(hipGetLastError());
            BF_TRY(bf_siftmgr_prefetch_frame_result(w->local->mgr));
            uint32_t last;
            BF_TRY(matchAndFilterFinish(w->local, cur, num, &last));
            validFlags[j] = last != 0xFFFFFFFFu ? 1 : 0;
        }
    }
    BF_HIP_TRY(hipMemcpyAsync(h->frames, w->d_rec, sizeof(bf_chunk_frame_record) * BF_CHUNK_MAX_FRAMES, hipMemcpyDeviceToHost, st));
    // prepareLocalSolve (:134-165), optimizeLocal (:242-271), fuseToGlobal (:297)
    int valid = 0;
    BF_TRY(bf_bundler_is_valid(w->local, &valid));
    h->chunkValid = valid;
    if (valid) {
        int removed = 0, solveValid = 0;
        BF_TRY(bf_bundler_optimize(w->local, w->gbs.s_numLocalNonLinIterations, w->gbs.s_numLocalLinIterations, w->gbs.s_useLocalVerify, 0, 0, &removed, &solveValid));
        h->solveValid = solveValid;
        if (solveValid) {
            BF_HIP_TRY(hipMemcpyAsync(h->localTrajectory, w->local->d_trajectory, sizeof(m44) * (w->gbs.s_submapSize + 1), hipMemcpyDeviceToHost, st));
            BF_TRY(bf_siftmgr_reset(w->fuseMgr));
            BF_TRY(bf_siftmgr_fuse_to_global(w->local->mgr, w->fuseMgr, w->local->siftIntrinsics.e, (const float*)w->local->d_trajectory, w->local->siftIntrinsicsInv.e));
            bf_sift_image_gpu img;
            BF_TRY(bf_siftmgr_get_image(w->fuseMgr, 0, &img));
            int32_t nk = 0;
            BF_TRY(bf_siftmgr_get_num_keypoints(w->fuseMgr, 0, 1, &nk));
            h->numKeys = (uint32_t)std::max(nk, 0);
            if (h->numKeys) {
                BF_HIP_TRY(hipMemcpyAsync(base + h->offKeys, img.d_keyPoints, sizeof(bf_sift_keypoint) * h->numKeys, hipMemcpyDeviceToHost, st));
                BF_HIP_TRY(hipMemcpyAsync(base + h->offDescs, img.d_keyPointDescs, (size_t)128 * h->numKeys, hipMemcpyDeviceToHost, st));
            }
            bf_cached_frame cf;
            BF_TRY(bf_cache_get_frame(w->local->cache, 0, &cf));
            const size_t n = (size_t)w->cacheW * w->cacheH;
            const void* src[6] = {cf.d_depthDownsampled, cf.d_cameraposDownsampled, cf.d_intensityDownsampled, cf.d_intensityDerivsDownsampled, cf.d_normalsDownsampledUCHAR4, cf.d_normalsDownsampled};
            const size_t bytes[6] = {n * 4, n * 16, n * 4, n * 8, n * 4, n * 16};
            for (int k = 0; k < 6; ++k) BF_HIP_TRY(hipMemcpyAsync(base + h->offCache[k], src[k], bytes[k], hipMemcpyDeviceToHost, st));
        }
        std::vector<int> v(w->gbs.s_submapSize + 1, 0);
        BF_TRY(bf_bundler_get_valid_images(w->local, v.data(), w->gbs.s_submapSize + 1));
        for (uint32_t i = 0; i <= w->gbs.s_submapSize; ++i) h->validImages[i] = v[i];
    }
    BF_HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t j = 0; j < numFrames; ++j) h->frames[j].valid = validFlags[j];
    BF_TRY(bf_bundler_reset(w->local));
    return BF_OK;
}

}  // extern "C"

// Stage B — the global half, replicated on every rank, in stream order
namespace {

// processInput (:167-227) for a frame whose chunk-local results are in `pkg`
int obProcessInputChunked(bf_online_bundler* ob, uint32_t curFrame, const bf_chunk_header* pkg, uint32_t localIdx) {
    BF_REQUIRE(ob->pendCount == 0, "processInput already in flight");
    BF_REQUIRE(ob->solveLag == 0, "chunked mode runs the global half in the serial order (no lagged solve)");
    BF_REQUIRE(pkg && pkg->magic == BF_CHUNK_MAGIC && pkg->submapSize == ob->submapSize, "bad chunk package");
    BF_REQUIRE(localIdx < pkg->numFrames && curFrame == pkg->chunkIndex * ob->submapSize + localIdx, "frame is not frame localIdx of the package's chunk");
    BF_REQUIRE(!(curFrame > 0 && ob->lastFrameProcessed == (int)curFrame), "chunked mode: the sequence end is driven by bf_pipeline_process_end_of_sequence");
    const bool bIsLastLocal = ob->isLastLocalFrame(curFrame);
    ob->bLastFrameValid = true;
    if (localIdx > 0) {
        const bf_chunk_frame_record& rec = pkg->frames[localIdx];
        ob->bLastFrameValid = rec.valid != 0;
        if (rec.valid && rec.prevLocal >= 0) {                  // computeCurrentSiftTransform (:118-132) with the chunk's relative transform
            m44 M; memcpy(M.e, rec.relInv, 64);
            k_sift_transform_ext<<<1, 1, 0, ob->stream>>>(localIdx, ob->d_completeTrajectory, ob->lastValidCompleteTransform, ob->d_siftTrajectory, curFrame, rec.prevLocal, M,
                                                         ob->d_currIntegrateTransform + curFrame);
            BF_HIP_TRY(hipGetLastError());
            BF_HIP_TRY(hipMemcpyAsync(ob->h_pinT, ob->d_currIntegrateTransform + curFrame, sizeof(m44), hipMemcpyDeviceToHost, ob->stream));
            BF_HIP_TRY(hipStreamSynchronize(ob->stream));
            ob->currIntegrateTransform[curFrame] = *ob->h_pinT;
        }
        if (!ob->bLastFrameValid) {
            ob->currIntegrateTransform[curFrame] = minfM();
            BF_HIP_TRY(hipMemcpyAsync(ob->d_siftTrajectory + curFrame, ob->d_siftTrajectory + curFrame - 1, sizeof(m44), hipMemcpyDeviceToDevice, ob->stream));
        }
    }
    if (bIsLastLocal) {                                         // prepareLocalSolve (:134-165) without the bundler swap
        BF_REQUIRE(localIdx + 1 == pkg->numFrames && pkg->numFrames == ob->submapSize + 1, "chunked mode needs complete chunks");
        ob->processState = bf_online_bundler::DO_NOTHING;
        const uint32_t curLocalIdx = (std::max(curFrame, 1u) - 1) / ob->submapSize;
        if (pkg->chunkValid) { ob->localToSolve = (int)curLocalIdx; ob->processState = bf_online_bundler::PROCESS; }
        else { ob->localToSolve = -((int)curLocalIdx + ID_MARK_OFFSET); ob->processState = bf_online_bundler::INVALIDATE; }
    }
    ob->lastFrameProcessed = (int)curFrame;
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_pipeline_process_frame_chunked(bf_pipeline* p, const float* d_depth, const uint8_t* d_color, const void* h_package, uint64_t packageBytes, uint32_t localIdx, int* gotFrame) {
    BF_REQUIRE(p && d_depth && d_color && h_package, "null argument");
    BF_REQUIRE(!p->timings, "per-stage timings are not available in chunked mode");
    {
        uint32_t cw = 0, ch = 0; float k4[4];
        BF_TRY(bf_cache_get_geometry(p->ob->global->cache, &cw, &ch, k4));
        BF_TRY(chunkPackageCheck(reinterpret_cast<const bf_chunk_header*>(h_package), packageBytes, p->ob->global->maxKeys, cw, ch, p->ob->submapSize));
    }
    BF_TRY(plFlush(p));                                         // (a pipeline is driven either serially or chunked; nothing is deferred in chunked mode)
    const bf_chunk_header* pkg = reinterpret_cast<const bf_chunk_header*>(h_package);
    hipStream_t sd = p->sIngest;
    // ---- read input: the ingest filters produce the frame that is integrated (CUDAImageManager::process); no detection on this rank
    int got = 0;
    BF_TRY(bf_image_manager_process_device(p->im, d_depth, d_color, &got));
    if (gotFrame) *gotFrame = got;
    if (!got) return BF_OK;
    const uint32_t frame = p->im->currFrame - 1;
    BF_HIP_TRY(hipEventRecord(p->evIngest[frame % bf_pipeline::NEV], sd));
    // ---- fix old frames, processInput, reconstruction of the current frame, bundling optimisation: plBodyRest with the package in place of m_local / m_optLocal
    p->ob->extChunk = pkg;
    int rc = obProcessInputChunked(p->ob, frame, pkg, localIdx);
    if (rc == BF_OK) {           // phase 2: processInput already complete, _end has nothing to read back
        bf_online_bundler::Pend& P = p->ob->pend[(p->ob->pendHead + p->ob->pendCount) % bf_online_bundler::PEND];
        P = bf_online_bundler::Pend(); P.phase = 2; P.frame = frame;
        p->ob->pendCount++;
        rc = plBodyRest(p, frame, true);
    }
    p->ob->extChunk = nullptr;
    BF_TRY(rc);
    BF_HIP_TRY(hipEventSynchronize(p->evIngest[frame % bf_pipeline::NEV]));
    return BF_OK;
}

}  // extern "C"
