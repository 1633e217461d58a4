// CorrespondenceEvaluator (CorrespondenceEvaluator.h:39-141, CorrespondenceEvaluator.cpp:11-96 and :98-314;
// paths relative to /root/reference/FriedLiver/Source): precision / recall of the image-to-image correspondences of the frame that
// was just matched, against a reference trajectory, after each stage of Bundler::matchAndFilter (raw, kabsch, sa, dense;
// Bundler.cpp:145-204).
//
// The reference does the whole evaluation on the host: per key frame it downloads the 80x60 depth and intensity image of EVERY
// cached frame (2 x 19 KB D2H per frame and stage-0 call) and walks them in scalar loops.  Here the ground-truth overlap test
// (computeOverlap / computeCorrespondences) is one launch over all cached frames - one workgroup per previous frame, both projection
// directions in the same pass, camera-space points and normals recomputed from the cached depth in registers - and only 16 bytes
// of counters per frame come back; the match lists are read back once per call like the reference does.  The arithmetic is the
// reference's host arithmetic, term by term (mLib mat4f * vec4f rows are summed left to right; the library is compiled with
// -ffp-contract=off), so the counters equal those of a scalar restatement bit for bit (tests/test_evaluator_gpu.py).
//
// Kept quirks of the reference (they decide what is counted):
//  * the squared distance of a match is compared with m_maxProjErrorForCorrectCorr = 0.2 (not with its square, .cpp:79);
//  * numMatches[] is read for all images of the manager, including slots the current matching round did not touch (.cpp:62-69);
//  * normals of interior pixels with an invalid neighbour keep the value of PointImage::allocate - mLib's vec3 default constructor,
//    (0,0,0) - so they pass the "!= -inf" tests and fail the normal-angle test (.cpp:271-288).
// Not restated: the debugPrint branch (point clouds / PNGs via FreeImage, .cpp:115-150,167-172,216-222) and the unused
// sumResidual / sumWeight outputs of computeCorrespondences.
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../../include/bf_pipeline.h"
#include "bf_device.h"
#include "bf_internal.h"

using namespace bf;

#define BF_TRY(expr) do { int _rc = (expr); if (_rc != BF_OK) return _rc; } while (0)

namespace {

const float NINF = -std::numeric_limits<float>::infinity();

struct OverlapArgs {
    const bf_cached_frame* frames;
    const m44* curToPrv;            // [numFrames][2]: transform cur -> prev, and its inverse
    uint32_t* counts;               // [numFrames][4]: numCorr0, numValid0 (cur -> prev), numCorr1, numValid1 (prev -> cur)
    uint32_t curFrame, W, H;
    m44 K, Kinv;
    float depthMin, depthMax, distThresh, normalThresh;
};

// computeCameraSpacePositions (.cpp:300-311): (Kinv * vec4(x d, y d, d, d)).xyz, -inf for invalid depth
__device__ __forceinline__ f3 evalCamPos(const float* depth, uint32_t W, int x, int y, const m44& Ki) {
    const float d = depth[(uint32_t)y * W + (uint32_t)x];
    if (d == BF_MINF) return mk3(BF_MINF, BF_MINF, BF_MINF);
    const float vx = (float)x * d, vy = (float)y * d;
    return mk3(Ki.e[0] * vx + Ki.e[1] * vy + Ki.e[2] * d + Ki.e[3] * d,
               Ki.e[4] * vx + Ki.e[5] * vy + Ki.e[6] * d + Ki.e[7] * d,
               Ki.e[8] * vx + Ki.e[9] * vy + Ki.e[10] * d + Ki.e[11] * d);
}

// computeNormals (.cpp:271-298)
__device__ __forceinline__ f3 evalNormal(const float* depth, uint32_t W, uint32_t H, int x, int y, const m44& Ki, f3 CC) {
    if (!(x > 0 && x + 1 < (int)W && y > 0 && y + 1 < (int)H)) return mk3(BF_MINF, BF_MINF, BF_MINF);
    const f3 PC = evalCamPos(depth, W, x, y + 1, Ki), CP = evalCamPos(depth, W, x + 1, y, Ki);
    const f3 MC = evalCamPos(depth, W, x, y - 1, Ki), CM = evalCamPos(depth, W, x - 1, y, Ki);
    if (CC.x != BF_MINF && PC.x != BF_MINF && CP.x != BF_MINF && MC.x != BF_MINF && CM.x != BF_MINF) {
        const f3 a = mk3(PC.x - MC.x, PC.y - MC.y, PC.z - MC.z), b = mk3(CP.x - CM.x, CP.y - CM.y, CP.z - CM.z);
        const f3 n = mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
        const float l = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
        if (l > 0.0f) { const float nl = -l; return mk3(n.x / nl, n.y / nl, n.z / nl); }
        return mk3(BF_MINF, BF_MINF, BF_MINF);
    }
    return mk3(0.0f, 0.0f, 0.0f);            // left at the image's initial value
}

// computeCorrespondences (.cpp:98-224) for one source pixel; returns bit 0 = counted in numValid, bit 1 = counted in numCorr
__device__ __forceinline__ uint32_t evalPixel(const float* depth0, const float* depth1, uint32_t W, uint32_t H, int x, int y, const m44& T, const OverlapArgs& a) {
    const f3 p0 = evalCamPos(depth0, W, x, y, a.Kinv);
    const f3 n0 = evalNormal(depth0, W, H, x, y, a.Kinv, p0);
    if (!(p0.x != BF_MINF && n0.x != BF_MINF)) return 0u;
    uint32_t r = 0u;
    const float d0 = depth0[(uint32_t)y * W + (uint32_t)x];
    if (d0 > a.depthMin && d0 < a.depthMax) r |= 1u;
    const f3 pT = mk3(T.e[0] * p0.x + T.e[1] * p0.y + T.e[2] * p0.z + T.e[3] * 1.0f,
                      T.e[4] * p0.x + T.e[5] * p0.y + T.e[6] * p0.z + T.e[7] * 1.0f,
                      T.e[8] * p0.x + T.e[9] * p0.y + T.e[10] * p0.z + T.e[11] * 1.0f);
    const float pTw = T.e[12] * p0.x + T.e[13] * p0.y + T.e[14] * p0.z + T.e[15] * 1.0f;
    const f3 nT = mk3(T.e[0] * n0.x + T.e[1] * n0.y + T.e[2] * n0.z + T.e[3] * 0.0f,
                      T.e[4] * n0.x + T.e[5] * n0.y + T.e[6] * n0.z + T.e[7] * 0.0f,
                      T.e[8] * n0.x + T.e[9] * n0.y + T.e[10] * n0.z + T.e[11] * 0.0f);
    const float nTw = T.e[12] * n0.x + T.e[13] * n0.y + T.e[14] * n0.z + T.e[15] * 0.0f;
    // cameraToDepth (.h:120-124): mat4f * vec3f = affine product followed by the division by w
    const float qx = a.K.e[0] * pT.x + a.K.e[1] * pT.y + a.K.e[2] * pT.z + a.K.e[3];
    const float qy = a.K.e[4] * pT.x + a.K.e[5] * pT.y + a.K.e[6] * pT.z + a.K.e[7];
    const float qz = a.K.e[8] * pT.x + a.K.e[9] * pT.y + a.K.e[10] * pT.z + a.K.e[11];
    const float qw = a.K.e[12] * pT.x + a.K.e[13] * pT.y + a.K.e[14] * pT.z + a.K.e[15];
    const float hx = qx / qw, hy = qy / qw, hz = qz / qw;
    const int sx = f2i(floorf(hx / hz + 0.5f)), sy = f2i(floorf(hy / hz + 0.5f));     // math::round
    if (sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H) {
        const f3 pt = evalCamPos(depth1, W, sx, sy, a.Kinv);
        const f3 nt = evalNormal(depth1, W, H, sx, sy, a.Kinv, pt);
        if (pt.x != BF_MINF && nt.x != BF_MINF) {
            const float ex = pT.x - pt.x, ey = pT.y - pt.y, ez = pT.z - pt.z, ew = pTw - 1.0f;
            const float d = sqrtf(ex * ex + ey * ey + ez * ez + ew * ew);
            const float dNormal = nT.x * nt.x + nT.y * nt.y + nT.z * nt.z + nTw * 0.0f;
            if (dNormal >= a.normalThresh && d <= a.distThresh) r |= 2u;
        }
    }
    return r;
}

__global__ __launch_bounds__(256) void k_eval_overlap(OverlapArgs a) {
    const uint32_t prv = blockIdx.x;
    __shared__ uint32_t acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0u;
    __syncthreads();
    if (prv != a.curFrame) {
        const float* dCur = a.frames[a.curFrame].d_depthDownsampled;
        const float* dPrv = a.frames[prv].d_depthDownsampled;
        const m44 T0 = a.curToPrv[2 * prv], T1 = a.curToPrv[2 * prv + 1];
        int c0 = 0, v0 = 0, c1 = 0, v1 = 0;
        for (uint32_t i = threadIdx.x; i < a.W * a.H; i += blockDim.x) {
            const int x = (int)(i % a.W), y = (int)(i / a.W);
            const uint32_t r0 = evalPixel(dCur, dPrv, a.W, a.H, x, y, T0, a);
            const uint32_t r1 = evalPixel(dPrv, dCur, a.W, a.H, x, y, T1, a);
            v0 += (int)(r0 & 1u); c0 += (int)(r0 >> 1); v1 += (int)(r1 & 1u); c1 += (int)(r1 >> 1);
        }
        c0 = wave_sum_i(c0); v0 = wave_sum_i(v0); c1 = wave_sum_i(c1); v1 = wave_sum_i(v1);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], (uint32_t)c0); atomicAdd(&acc[1], (uint32_t)v0); atomicAdd(&acc[2], (uint32_t)c1); atomicAdd(&acc[3], (uint32_t)v1); }
    }
    __syncthreads();
    if (threadIdx.x < 4) a.counts[4 * prv + threadIdx.x] = acc[threadIdx.x];
}

// mat4f * vec3f on the host (affine product, then the division by w)
void xformPoint(const m44& m, const float v[3], float out[3]) {
    const float x = m.e[0] * v[0] + m.e[1] * v[1] + m.e[2] * v[2] + m.e[3];
    const float y = m.e[4] * v[0] + m.e[5] * v[1] + m.e[6] * v[2] + m.e[7];
    const float z = m.e[8] * v[0] + m.e[9] * v[1] + m.e[10] * v[2] + m.e[11];
    const float w = m.e[12] * v[0] + m.e[13] * v[1] + m.e[14] * v[2] + m.e[15];
    out[0] = x / w; out[1] = y / w; out[2] = z / w;
}

float precisionOf(const bf_corr_evaluation& e) { return e.numDetected > 0 ? (float)e.numCorrect / (float)e.numDetected : NINF; }
float recallOf(const bf_corr_evaluation& e) { return e.numTotal > 0 ? (float)e.numDetected / (float)e.numTotal : NINF; }

}  // namespace

struct bf_correspondence_evaluator {
    std::vector<m44> referenceTrajectory;
    std::string logFilePrefix;
    float minOverlapThreshForGTCorr = 0.1f, maxProjErrorForCorrectCorr = 0.2f;          // .h:44-46
    std::ofstream outPerFrame, outIncorrect;
    // cached data of the current frame (computeCachedData)
    std::vector<bf_sift_keypoint> cachedKeys;
    std::vector<uint8_t> hasGTCorrByOverlap;
    std::vector<uint32_t> overlapCounts;                       // the 4 counters per frame of the last computeCachedData
    std::map<std::string, bf_corr_evaluation> totals;          // sum of evaluate() results per corrType (not in the reference: it only logs)
    // device scratch
    m44* d_T = nullptr; uint32_t* d_counts = nullptr; uint32_t capacity = 0;
    bool logging() const { return !logFilePrefix.empty(); }
};

namespace {

int evEnsure(bf_correspondence_evaluator* ev, uint32_t n) {
    if (n <= ev->capacity) return BF_OK;
    if (ev->d_T) hipFree(ev->d_T);
    if (ev->d_counts) hipFree(ev->d_counts);
    ev->d_T = nullptr; ev->d_counts = nullptr; ev->capacity = 0;
    const uint32_t cap = std::max<uint32_t>(64, 2 * n);
    BF_HIP_TRY(BF_MALLOC((void**)&ev->d_T, sizeof(m44) * 2 * cap));
    BF_HIP_TRY(BF_MALLOC((void**)&ev->d_counts, sizeof(uint32_t) * 4 * cap));
    ev->capacity = cap;
    return BF_OK;
}

// computeCachedData (.cpp:11-45)
int evComputeCachedData(bf_correspondence_evaluator* ev, bf_siftmgr* mgr, bf_cache* cache, const bf_corr_eval_params& p, hipStream_t stream) {
    uint32_t curFrame, numFrames, maxKeys;
    BF_TRY(bf_siftmgr_get_current_frame(mgr, &curFrame));
    BF_TRY(bf_siftmgr_get_num_images(mgr, &numFrames));
    BF_TRY(bf_siftmgr_get_max_num_keypoints_per_image(mgr, &maxKeys));
    BF_REQUIRE(numFrames <= ev->referenceTrajectory.size() && curFrame < numFrames, "reference trajectory shorter than the number of images");
    const bf_sift_keypoint* d_keys = nullptr;
    BF_TRY(bf_siftmgr_get_keys_gpu(mgr, &d_keys, nullptr, nullptr));
    ev->cachedKeys.resize((size_t)numFrames * maxKeys);
    BF_HIP_TRY(hipMemcpyAsync(ev->cachedKeys.data(), d_keys, sizeof(bf_sift_keypoint) * ev->cachedKeys.size(), hipMemcpyDeviceToHost, stream));

    uint32_t W, H; float k4[4];
    BF_TRY(bf_cache_get_geometry(cache, &W, &H, k4));
    uint32_t numCached;
    BF_TRY(bf_cache_get_num_frames(cache, &numCached));
    BF_REQUIRE(numCached >= numFrames, "cache holds fewer frames than the key-point manager");
    OverlapArgs a = {};
    BF_TRY(bf_cache_get_frames_gpu(cache, &a.frames));
    a.K = identity44(); a.K.e[0] = k4[0]; a.K.e[5] = k4[1]; a.K.e[2] = k4[2]; a.K.e[6] = k4[3];
    a.Kinv = inverse44(a.K);
    a.curFrame = curFrame; a.W = W; a.H = H;
    a.depthMin = p.depthMin; a.depthMax = p.depthMax; a.distThresh = p.distThresh; a.normalThresh = p.normalThresh;
    BF_TRY(evEnsure(ev, numFrames));
    std::vector<m44> T(2 * (size_t)numFrames);
    for (uint32_t i = 0; i < numFrames; ++i) {
        T[2 * i] = mul44(inverse44(ev->referenceTrajectory[i]), ev->referenceTrajectory[curFrame]);      // transformCurToPrv (.cpp:32)
        T[2 * i + 1] = inverse44(T[2 * i]);
    }
    BF_HIP_TRY(hipMemcpyAsync(ev->d_T, T.data(), sizeof(m44) * T.size(), hipMemcpyHostToDevice, stream));
    a.curToPrv = ev->d_T; a.counts = ev->d_counts;
    k_eval_overlap<<<numFrames, 256, 0, stream>>>(a);
    BF_HIP_TRY(hipGetLastError());
    ev->overlapCounts.assign(4 * (size_t)numFrames, 0u);
    BF_HIP_TRY(hipMemcpyAsync(ev->overlapCounts.data(), ev->d_counts, sizeof(uint32_t) * 4 * numFrames, hipMemcpyDeviceToHost, stream));
    BF_HIP_TRY(hipStreamSynchronize(stream));

    if (ev->hasGTCorrByOverlap.size() < numFrames) ev->hasGTCorrByOverlap.resize(numFrames, 0);            // resize(numFrames, false): old entries are kept
    for (uint32_t i = 0; i < numFrames; ++i) {
        if (i == curFrame) continue;
        // computeOverlap (.cpp:226-250)
        const uint32_t* c = &ev->overlapCounts[4 * i];
        uint32_t oc = c[0], ov = c[1];
        const float percent0 = (float)c[0] / (float)c[1];
        if (!(percent0 > ev->minOverlapThreshForGTCorr)) {
            const float percent1 = (float)c[2] / (float)c[3];
            if (!(percent0 > percent1)) { oc = c[2]; ov = c[3]; }
        }
        const float o = (float)oc / (float)ov;
        if (ov > 0 && o > ev->minOverlapThreshForGTCorr) ev->hasGTCorrByOverlap[i] = 1;
    }
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_correspondence_evaluator_create(const float* h_referenceTrajectory, uint32_t numTransforms, const char* logFilePrefix, bf_correspondence_evaluator** out) {
    BF_REQUIRE(out && (h_referenceTrajectory || numTransforms == 0), "null argument");
    bf_correspondence_evaluator* ev = new bf_correspondence_evaluator;
    ev->referenceTrajectory.resize(numTransforms);
    if (numTransforms) memcpy(ev->referenceTrajectory.data(), h_referenceTrajectory, sizeof(m44) * numTransforms);
    ev->logFilePrefix = logFilePrefix ? logFilePrefix : "";
    if (ev->logging()) {                                                                              // .h:47-54
        ev->outPerFrame.open(ev->logFilePrefix + "_frame.csv");
        ev->outIncorrect.open(ev->logFilePrefix + "_wrong.csv");
        if (!ev->outPerFrame.is_open() || !ev->outIncorrect.is_open()) {
            set_error("[CorrespondenceEvaluator] failed to open log file(s): %s", ev->logFilePrefix.c_str());
            delete ev;
            return BF_ERR_INVALID_ARG;
        }
        ev->outPerFrame << "numFrames,curFrame,type,precision,recall,numCorrect,numDetected,numTotal" << std::endl;
        ev->outIncorrect << "numFrames,curFrame,matchFrame,type,err" << std::endl;
    }
    *out = ev;
    return BF_OK;
}

int bf_correspondence_evaluator_destroy(bf_correspondence_evaluator* ev) {
    if (!ev) return BF_OK;
    if (ev->d_T) hipFree(ev->d_T);
    if (ev->d_counts) hipFree(ev->d_counts);
    delete ev;
    return BF_OK;
}

int bf_correspondence_evaluator_finish_logging_to_file(bf_correspondence_evaluator* ev) {               // .h:64-69
    BF_REQUIRE(ev, "null evaluator");
    if (ev->logging()) { ev->outPerFrame.close(); ev->outIncorrect.close(); }
    return BF_OK;
}

// evaluate (.cpp:47-96)
int bf_correspondence_evaluator_evaluate(bf_correspondence_evaluator* ev, bf_siftmgr* mgr, bf_cache* cache, const float siftIntrinsicsInv[16],
                                         const bf_corr_eval_params* params, int filtered, int recomputeCache, int clearCache, const char* corrType,
                                         void* hip_stream, bf_corr_evaluation* out) {
    BF_REQUIRE(ev && mgr && cache && siftIntrinsicsInv && params && corrType, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    uint32_t curFrame, numFrames, maxKeys;
    BF_TRY(bf_siftmgr_get_current_frame(mgr, &curFrame));
    BF_TRY(bf_siftmgr_get_num_images(mgr, &numFrames));
    BF_TRY(bf_siftmgr_get_max_num_keypoints_per_image(mgr, &maxKeys));
    const bool useLog = ev->logging() && ev->outPerFrame.is_open();
    if (recomputeCache) BF_TRY(evComputeCachedData(ev, mgr, cache, *params, stream));
    BF_REQUIRE(ev->hasGTCorrByOverlap.size() >= numFrames && ev->cachedKeys.size() >= (size_t)numFrames * maxKeys && numFrames <= ev->referenceTrajectory.size(),
               "evaluate() without cached data: the first call for a frame needs recomputeCache");

    // getCurrMatchKeyPointIndicesDEBUG (SIFTImageManager.h:236-243)
    const uint32_t offsetVal = filtered ? BF_MAX_MATCHES_PER_IMAGE_PAIR_FILTERED : BF_MAX_MATCHES_PER_IMAGE_PAIR_RAW;
    std::vector<uint32_t> idx((size_t)numFrames * offsetVal * 2);
    std::vector<int32_t> numMatches(numFrames);
    const uint32_t* d_idx = nullptr; const int32_t* d_num = nullptr;
    BF_TRY(bf_siftmgr_get_curr_matches_gpu(mgr, filtered, &d_idx, &d_num));
    BF_HIP_TRY(hipMemcpyAsync(numMatches.data(), d_num, sizeof(int32_t) * numFrames, hipMemcpyDeviceToHost, stream));
    BF_HIP_TRY(hipMemcpyAsync(idx.data(), d_idx, sizeof(uint32_t) * idx.size(), hipMemcpyDeviceToHost, stream));
    BF_HIP_TRY(hipStreamSynchronize(stream));

    const m44 Ki = [&] { m44 m; memcpy(m.e, siftIntrinsicsInv, 64); return m; }();
    bf_corr_evaluation eval = {0, 0, 0};
    for (uint32_t p = 0; p < numFrames; ++p) {
        if (p == curFrame) continue;
        const uint32_t nm = numMatches[p] > 0 ? (uint32_t)numMatches[p] : 0u;
        if (ev->hasGTCorrByOverlap[p]) { eval.numTotal++; if (nm > 0) eval.numDetected++; }
        float maxErr2 = 0.0f;
        for (uint32_t m = 0; m < nm && m < offsetVal; ++m) {
            const uint32_t i0 = idx[2 * ((size_t)offsetVal * p + m)], i1 = idx[2 * ((size_t)offsetVal * p + m) + 1];
            BF_REQUIRE(i0 < ev->cachedKeys.size() && i1 < ev->cachedKeys.size(), "match refers to a key point outside the manager");
            const bf_sift_keypoint &k0 = ev->cachedKeys[i0], &k1 = ev->cachedKeys[i1];
            const float v0[3] = {k0.depth * k0.pos[0], k0.depth * k0.pos[1], k0.depth * 1.0f}, v1[3] = {k1.depth * k1.pos[0], k1.depth * k1.pos[1], k1.depth * 1.0f};
            float cp0[3], cp1[3], w0[3], w1[3];
            xformPoint(Ki, v0, cp0); xformPoint(Ki, v1, cp1);                                                  // depthToCamera (.h:125-128)
            xformPoint(ev->referenceTrajectory[p], cp0, w0); xformPoint(ev->referenceTrajectory[curFrame], cp1, w1);
            const float ex = w0[0] - w1[0], ey = w0[1] - w1[1], ez = w0[2] - w1[2];
            const float err2 = ex * ex + ey * ey + ez * ez;
            if (err2 > maxErr2) maxErr2 = err2;
        }
        if (nm > 0) {
            if (maxErr2 < ev->maxProjErrorForCorrectCorr) { if (ev->hasGTCorrByOverlap[p]) eval.numCorrect++; }
            else if (useLog) ev->outIncorrect << numFrames << "," << curFrame << "," << p << "," << corrType << "," << std::sqrt(maxErr2) << std::endl;
        }
        if (useLog && ev->hasGTCorrByOverlap[p] && nm == 0) ev->outIncorrect << numFrames << "," << curFrame << "," << p << "," << corrType << "," << -1.0f << std::endl;
    }
    if (useLog)
        ev->outPerFrame << numFrames << "," << curFrame << "," << corrType << "," << precisionOf(eval) << "," << recallOf(eval) << "," << eval.numCorrect << ","
                        << eval.numDetected << "," << eval.numTotal << std::endl;
    bf_corr_evaluation& t = ev->totals[corrType];
    t.numCorrect += eval.numCorrect; t.numDetected += eval.numDetected; t.numTotal += eval.numTotal;
    if (clearCache) { ev->cachedKeys.clear(); ev->hasGTCorrByOverlap.clear(); }                                  // clearCachedData (.h:73-77)
    if (out) *out = eval;
    return BF_OK;
}

int bf_correspondence_evaluator_get_total(bf_correspondence_evaluator* ev, const char* corrType, bf_corr_evaluation* out) {
    BF_REQUIRE(ev && corrType && out, "null argument");
    auto it = ev->totals.find(corrType);
    if (it == ev->totals.end()) { out->numCorrect = out->numDetected = out->numTotal = 0; return BF_OK; }
    *out = it->second;
    return BF_OK;
}

int bf_correspondence_evaluator_get_overlap_counts(bf_correspondence_evaluator* ev, uint32_t* h_counts, uint8_t* h_hasGTCorr, uint32_t numFrames) {
    BF_REQUIRE(ev && 4 * (size_t)numFrames <= ev->overlapCounts.size(), "no cached data for that many frames");
    if (h_counts) memcpy(h_counts, ev->overlapCounts.data(), sizeof(uint32_t) * 4 * numFrames);
    if (h_hasGTCorr) {
        BF_REQUIRE(ev->hasGTCorrByOverlap.size() >= numFrames, "cached data was cleared");
        memcpy(h_hasGTCorr, ev->hasGTCorrByOverlap.data(), numFrames);
    }
    return BF_OK;
}

float bf_corr_evaluation_get_precision(const bf_corr_evaluation* e) { return e ? precisionOf(*e) : NINF; }
float bf_corr_evaluation_get_recall(const bf_corr_evaluation* e) { return e ? recallOf(*e) : NINF; }

}  // extern "C"
