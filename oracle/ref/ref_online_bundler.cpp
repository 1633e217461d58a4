// oracle/ref/ref_online_bundler.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's OnlineBundler (OnlineBundler.cpp / OnlineBundler.cu / OnlineBundlerHelper.h compiled as they are): processInput
// :164-238, process = optimizeLocal :252-283 + processGlobal :292-362 + optimizeGlobal :364-416, updateTrajectory, prepareLocalSolve,
// computeCurrentSiftTransform, getCurrentIntegrationFrame - the whole bundling half of the frame loop, on the reference's own Bundler,
// SiftGPU fork, SIFTImageManager, CUDACache, SBA, CUDASolverBundling, TrajectoryManager and CUDAImageManager (the ingest: CUDAImageManager.cpp compiled
// as it is).  Stand-ins: mLib's element arithmetic; an RGBDSensor that carries sizes, intrinsics and the sensor frame the test supplies.
// One frame = set_frame (= CUDAImageManager::process) + process_input + process, the serial order of FriedLiver.cpp:135-143.
#define private public
#define protected public
#include "OnlineBundler.h"
#include "RGBDSensor.h"
#include "CUDAImageManager.h"
#include "Bundler.h"
#include "TrajectoryManager.h"
#undef private
#undef protected

struct ref_app_state { unsigned int topNActive, numSolveFramesBeforeExit; float minPoseDistSqrt, colorSigmaD, colorSigmaR; int colorFilter; };
struct ref_online_bundler { OnlineBundler* ob; RGBDSensor* sensor; CUDAImageManager* im; };

extern "C" {

void ref_set_app_state(const ref_app_state* p) {
    GlobalAppState& a = GlobalAppState::get();
    a.s_topNActive = p->topNActive; a.s_numSolveFramesBeforeExit = p->numSolveFramesBeforeExit; a.s_minPoseDistSqrt = p->minPoseDistSqrt;
    a.s_colorSigmaD = p->colorSigmaD; a.s_colorSigmaR = p->colorSigmaR; a.s_colorFilter = p->colorFilter != 0;
    a.s_sensorIdx = 8;
}

// ref_set_bundling_state (ref_bundler.cpp) and ref_set_app_state first
ref_online_bundler* ref_ob_create(unsigned int depthW, unsigned int depthH, unsigned int colorW, unsigned int colorH, unsigned int integrationW, unsigned int integrationH,
                                  const float* depthIntrinsics16, const float* colorIntrinsics16) {
    ref_online_bundler* h = new ref_online_bundler;
    h->sensor = new RGBDSensor(depthW, depthH, colorW, colorH, mat4f(depthIntrinsics16), mat4f(colorIntrinsics16));
    h->im = new CUDAImageManager(integrationW, integrationH, GlobalBundlingState::get().s_widthSIFT, GlobalBundlingState::get().s_heightSIFT, h->sensor, false);
    h->ob = new OnlineBundler(h->sensor, h->im);
    return h;
}
void ref_ob_destroy(ref_online_bundler* h) { delete h->ob; delete h->im; delete h->sensor; delete h; }
// a new sensor frame through the reference's own ingest, CUDAImageManager::process (CUDAImageManager.cpp:22-158)
int ref_ob_set_frame(ref_online_bundler* h, const float* sensorDepth, const unsigned char* colorRGBX) { h->sensor->setFrame(sensorDepth, colorRGBX); return h->im->process() ? 1 : 0; }
// what the ingest left: the SIFT-side raw / filtered depth (sensor resolution) and the frame stored for integration (integration resolution)
void ref_ob_ingest_outputs(ref_online_bundler* h, float* depthRaw, float* depthFilt, unsigned int frame, float* depthIntegration, unsigned char* colorIntegrationRGBX) {
    const size_t n = (size_t)h->sensor->getDepthWidth() * h->sensor->getDepthHeight(), ni = (size_t)h->im->getIntegrationWidth() * h->im->getIntegrationHeight();
    memcpy(depthRaw, h->im->d_depthInputRaw, 4 * n); memcpy(depthFilt, h->im->d_depthInputFiltered, 4 * n);
    memcpy(depthIntegration, h->im->getIntegrateFrame(frame).getDepthFrameCPU(), 4 * ni);
    memcpy(colorIntegrationRGBX, h->im->getIntegrateFrame(frame).getColorFrameCPU(), 4 * ni);
}
// The depth Gauss filter evaluates exp() with glibc here and with include/bf_detmath.h in the oracle (<= 2 ulp per weight, pinned to 3e-6 on
// its own): the test checks the filtered depth to that bound, then puts the oracle's bits in so that everything downstream can be compared
// bit for bit.
void ref_ob_override_filtered_depth(ref_online_bundler* h, const float* depthFilt) {
    memcpy(h->im->d_depthInputFiltered, depthFilt, 4 * (size_t)h->sensor->getDepthWidth() * h->sensor->getDepthHeight());
}
void ref_ob_integration_intrinsics(ref_online_bundler* h, float* K16) { memcpy(K16, h->im->getDepthIntrinsics().matrix, 64); }
void ref_ob_process_input(ref_online_bundler* h) { h->ob->processInput(); }
void ref_ob_process(ref_online_bundler* h, unsigned int nlLocal, unsigned int linLocal, unsigned int nlGlobal, unsigned int linGlobal) { h->ob->process(nlLocal, linLocal, nlGlobal, linGlobal); }
int ref_ob_current_integration_frame(ref_online_bundler* h, float* T16, unsigned int* frameIdx, int* trackingLost) {
    mat4f T; bool lost = false; unsigned int idx = 0;
    const bool ok = h->ob->getCurrentIntegrationFrame(T, idx, lost);
    memcpy(T16, T.matrix, 64); *frameIdx = idx; *trackingLost = lost ? 1 : 0;
    return ok ? 1 : 0;
}
// state after a frame: [lastFrameProcessed, lastFrameValid, localToSolve, lastLocalSolved, numFramesPastEnd, numCompleteTransfroms, lastValidCompleteTransform,
//                       globalTrackingLost, processState, useSolve, totalNumOptLocalFrames]
void ref_ob_state(ref_online_bundler* h, int* out11) {
    const BundlerState& s = h->ob->m_state;
    out11[0] = s.m_lastFrameProcessed; out11[1] = s.m_bLastFrameValid; out11[2] = s.m_localToSolve; out11[3] = s.m_lastLocalSolved; out11[4] = (int)s.m_numFramesPastEnd;
    out11[5] = (int)s.m_numCompleteTransforms; out11[6] = (int)s.m_lastValidCompleteTransform; out11[7] = s.m_bGlobalTrackingLost; out11[8] = (int)s.m_processState;
    out11[9] = s.m_bUseSolve; out11[10] = (int)s.m_totalNumOptLocalFrames;
}
void ref_ob_complete_trajectory(ref_online_bundler* h, float* out16, unsigned int n) { memcpy(out16, h->ob->d_completeTrajectory, 64 * (size_t)n); }
void ref_ob_sift_trajectory(ref_online_bundler* h, float* out16, unsigned int n) { memcpy(out16, h->ob->d_siftTrajectory, 64 * (size_t)n); }
void ref_ob_local_trajectories(ref_online_bundler* h, float* out16, unsigned int n) { memcpy(out16, h->ob->d_localTrajectories, 64 * (size_t)n); }
void ref_ob_invalid_images_list(ref_online_bundler* h, unsigned int* out, unsigned int n) { for (unsigned int i = 0; i < n; ++i) out[i] = h->ob->m_invalidImagesList[i]; }
// the three bundlers: which = 0 m_local, 1 m_optLocal, 2 m_global (handles usable with the ref_bundler_* accessors; not owned)
struct ref_bundler { Bundler* b; CUDAImageManager* im; RGBDSensor* sensor; };
void ref_ob_bundler(ref_online_bundler* h, int which, ref_bundler* out) { out->b = which == 0 ? h->ob->m_local : (which == 1 ? h->ob->m_optLocal : h->ob->m_global); out->im = h->im; out->sensor = h->sensor; }
void* ref_ob_trajectory_manager(ref_online_bundler* h) { return h->ob->getTrajectoryManager(); }

}
