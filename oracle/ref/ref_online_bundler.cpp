// oracle/ref/ref_online_bundler.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's OnlineBundler (OnlineBundler.cpp / OnlineBundler.cu / OnlineBundlerHelper.h compiled as they are): processInput
// :164-238, process = optimizeLocal :252-283 + processGlobal :292-362 + optimizeGlobal :364-416, updateTrajectory, prepareLocalSolve,
// computeCurrentSiftTransform, getCurrentIntegrationFrame - the whole bundling half of the frame loop, on the reference's own Bundler,
// SiftGPU fork, SIFTImageManager, CUDACache, SBA, CUDASolverBundling and TrajectoryManager.  Stand-ins: mLib's element arithmetic; an
// RGBDSensor that carries sizes and colour intrinsics; a CUDAImageManager that hands over the frame the test supplies (the ingest that
// produces it is pinned at kernel level).  One frame = set_frame + process_input + process, the serial order of FriedLiver.cpp:135-143.
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
ref_online_bundler* ref_ob_create(unsigned int depthW, unsigned int depthH, unsigned int colorW, unsigned int colorH, const float* depthIntrinsics16, const float* colorIntrinsics16) {
    ref_online_bundler* h = new ref_online_bundler;
    h->sensor = new RGBDSensor(depthW, depthH, colorW, colorH, mat4f(colorIntrinsics16));
    h->im = new CUDAImageManager(depthW, depthH, colorW, colorH, mat4f(depthIntrinsics16));
    h->ob = new OnlineBundler(h->sensor, h->im);
    return h;
}
void ref_ob_destroy(ref_online_bundler* h) { delete h->ob; delete h->im; delete h->sensor; delete h; }
void ref_ob_set_frame(ref_online_bundler* h, const float* depthRaw, const float* depthFilt, const unsigned char* colorRGBX) { h->im->setFrame(depthRaw, depthFilt, (const uchar4*)colorRGBX); }
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
struct ref_bundler { Bundler* b; CUDAImageManager* im; };
void ref_ob_bundler(ref_online_bundler* h, int which, ref_bundler* out) { out->b = which == 0 ? h->ob->m_local : (which == 1 ? h->ob->m_optLocal : h->ob->m_global); out->im = h->im; }
void* ref_ob_trajectory_manager(ref_online_bundler* h) { return h->ob->getTrajectoryManager(); }

}
