// oracle/ref/ref_loop.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's frame loop itself, compiled as it is: DepthSensing/DepthSensing.cpp:723-762 (integrate, deIntegrate), :853-902 (reintegrate) and
// :966-1129 (OnD3D11FrameRender: read input, processInput, fix old frames, reconstruction of the current frame, bundling optimisation, the end-of-sequence
// exit) - the Makefile cuts those line ranges out of the file where it lies into DepthSensing_slice.inc (a temporary), and this file supplies what the
// rest of DepthSensing.cpp would have: the globals (the reference's own CUDAImageManager / OnlineBundler / CUDASceneRepHashSDF objects the test created
// through ref_online_bundler.cpp / ref_scene_host.cpp), and stand-ins for the DirectX side (rendering, text, video) and for chunk streaming (off for
// BundleFusion, zParametersDefault.txt:100).  RUN_MULTITHREADED is undefined in front of the slice: the serial order (the one the oracle loop and the product implement).
// Until round 5 these lines were RESTATED by the test that drives the reference's classes (tests/test_ref_pin_cpu.py: ref_reintegrate and the per-frame
// sequence); tests/test_ref_pin_cpu.py::test_compiled_frame_loop_equals_the_restated_one now holds that restatement to the compiled original.
#define private public
#define protected public
#include "GlobalAppState.h"
#include "GlobalBundlingState.h"
#include "ConditionManager.h"
#include "TimingLog.h"
#include "RGBDSensor.h"
#include "CUDAImageManager.h"
#include "OnlineBundler.h"
#include "TrajectoryManager.h"
#include "CUDASceneRepHashSDF.h"
#undef private
#undef protected
#include <vector>

struct ref_online_bundler { OnlineBundler* ob; RGBDSensor* sensor; CUDAImageManager* im; };          // as in ref_online_bundler.cpp

namespace {

// the volume the slice talks to: the reference's host class, with a record of what it was asked to do
struct LoopOp { int kind; float T[16]; int frame; };
struct SceneWithLog {
    CUDASceneRepHashSDF* s = nullptr;
    std::vector<LoopOp> log;
    void note(int kind, const mat4f& T, const DepthCameraData& d);
    void integrate(const mat4f& T, const DepthCameraData& d, const DepthCameraParams& p, unsigned int* bitMask) { note(0, T, d); s->integrate(T, d, p, bitMask); }
    void deIntegrate(const mat4f& T, const DepthCameraData& d, const DepthCameraParams& p, unsigned int* bitMask) { note(1, T, d); s->deIntegrate(T, d, p, bitMask); }
    void garbageCollect() { s->garbageCollect(); }
    unsigned int getHeapFreeCount() { return s->getHeapFreeCount(); }
};
struct CUDASceneRepChunkGrid {               // chunk streaming: never instantiated (g_chunkGrid stays NULL, s_streamingEnabled false)
    void streamOutToCPUPass0GPU(const vec3f&, float, bool, bool) {}
    void streamInToGPUPass1GPU(bool) {}
    unsigned int* getBitMaskGPU() { return nullptr; }
};

CUDAImageManager* g_CudaImageManager = nullptr;
OnlineBundler* g_depthSensingBundler = nullptr;
RGBDSensor* g_depthSensingRGBDSensor = nullptr;
SceneWithLog g_scene;
SceneWithLog* g_sceneRep = &g_scene;
CUDASceneRepChunkGrid* g_chunkGrid = nullptr;
mat4f g_transformWorld = mat4f::identity();
mat4f g_lastRigidTransform = mat4f::identity();
DepthCameraParams g_depthCameraParams;
bool g_renderText = false;
bool g_stopRequested = false;
int g_framesRendered = 0;

// which stored frame the slice just handed over: with the frames kept on the host (the reference's default) getDepthFrameGPU() stages the frame in ONE global
// buffer and remembers whose data it holds (CUDAImageManager.h:71-82)
void SceneWithLog::note(int kind, const mat4f& T, const DepthCameraData& d) {
    LoopOp o; o.kind = kind; memcpy(o.T, T.matrix, 64); o.frame = -1;
    for (size_t f = 0; f < g_CudaImageManager->m_data.size(); ++f) {
        CUDAImageManager::ManagedRGBDInputFrame* fr = &g_CudaImageManager->m_data[f];
        if (CUDAImageManager::ManagedRGBDInputFrame::s_bIsOnGPU ? (fr->m_depthIntegration == d.d_depthData) : (fr == CUDAImageManager::ManagedRGBDInputFrame::s_activeDepthGPU)) { o.frame = (int)f; break; }
    }
    log.push_back(o);
}

// the DirectX side of the frame: nothing to compute
void visualizeFrame(ID3D11DeviceContext*, ID3D11Device*, const mat4f&, bool) { ++g_framesRendered; }
void renderTopDown(ID3D11DeviceContext*, const mat4f&, bool) {}
void RenderText() {}
void StopScanningAndExit(bool aborted = false) { (void)aborted; g_stopRequested = true; }          // DepthSensing.cpp:904-961 extracts the mesh and calls exit(0)

}  // namespace

#ifndef CALLBACK
#define CALLBACK
#endif
#define DXUT_EndPerfEvent() ((void)0)

// GlobalAppState.h:12 defines RUN_MULTITHREADED (bundling on a second thread, hand-shakes through ConditionManager); the file keeps the serial order in the
// #else branches, and that is the order compiled here - the one the oracle loop and the product implement (the threaded hand-shake orders the same calls).
#undef RUN_MULTITHREADED
namespace {
#include "DepthSensing_slice.inc"
}

// GlobalAppState.cpp (the D3D11 query behind WaitForGPU) is not part of the build; the timing switches that call it are off
void GlobalAppState::WaitForGPU() {}

extern "C" {

// the objects of one run: the bundler / image manager / sensor of ref_ob_create, the volume of ref_hscene_create, the integration camera
void ref_loop_bind(ref_online_bundler* h, void* hscene, const void* cam, unsigned int maxFrameFixes) {
    GlobalAppState::get().s_maxFrameFixes = maxFrameFixes;            // (ref_set_app_state carries the switches the bundling classes read; this one is the loop's)
    g_CudaImageManager = h->im; g_depthSensingBundler = h->ob; g_depthSensingRGBDSensor = h->sensor;
    g_scene.s = (CUDASceneRepHashSDF*)hscene; g_scene.log.clear();
    memcpy(&g_depthCameraParams, cam, sizeof g_depthCameraParams);
    DepthCameraData::updateParams(g_depthCameraParams);
    g_transformWorld = mat4f::identity(); g_lastRigidTransform = mat4f::identity();
    g_stopRequested = false; g_framesRendered = 0;
    GlobalAppState::get().s_integrationEnabled = true; GlobalAppState::get().s_reconstructionEnabled = true; GlobalAppState::get().s_streamingEnabled = false;
    GlobalAppState::get().s_binaryDumpSensorUseTrajectory = false; GlobalAppState::get().s_recordData = false; GlobalAppState::get().s_generateVideo = false;
    GlobalAppState::get().s_printTimingsDirectory = "";
    h->sensor->setReceiving(true);
}
// the next sensor frame (what the device / file reader would deliver to processDepth / processColor); receiving = 0: the sequence is over
void ref_loop_set_frame(ref_online_bundler* h, const float* sensorDepth, const unsigned char* colorRGBX) { h->sensor->setFrame(sensorDepth, colorRGBX); h->sensor->setReceiving(true); }
void ref_loop_end_of_sequence(ref_online_bundler* h) { h->sensor->setReceiving(false); }
// ONE call of OnD3D11FrameRender (DepthSensing.cpp:966-1129); returns 1 when the loop asked to stop (StopScanningAndExit)
int ref_loop_frame_render() { OnD3D11FrameRender(nullptr, nullptr, 0.0, 0.0f, nullptr); return g_stopRequested ? 1 : 0; }
// what the slice asked of the volume since ref_loop_bind: kind (0 integrate, 1 de-integrate), transform, and the stored frame it handed over (its index)
unsigned int ref_loop_num_ops() { return (unsigned int)g_scene.log.size(); }
void ref_loop_op(unsigned int i, int* kind, float* T16, int* frame) {
    const LoopOp& o = g_scene.log[i];
    *kind = o.kind; memcpy(T16, o.T, 64); *frame = o.frame;
}
int ref_loop_frames_rendered() { return g_framesRendered; }

}
