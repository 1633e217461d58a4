// Throughput of the DROP-IN CLASS SURFACE (include/bundlefusion/bundlefusion.hpp: the reference's class and method names over the C ABI) on a recorded frame file:
// the serial body of DepthSensing.cpp's OnD3D11FrameRender (:966-1095 with reintegrate :854-902) exactly as examples/headless_driver.cpp writes it - one frame at a
// time, processInput() returning the frame's pose before anything else happens - without the per-frame printf.  bench.py runs it on the stream of its timed window
// and reports the result beside bf_pipeline_*'s (`class_surface`).  With `deferred = 1` the CUDASceneRepHashSDF wrapper collects a frame's integrate / deIntegrate
// calls into one bf_scene_run_batch (setDeferredBatching): the only change a maintainer would make to the loop.
//   class_surface_bench <frames.bin> <W> <H> <frames> <preroll> <voxel> <buckets> <blocks> <fx> <fy> <mx> <my> <deferred>
// frames.bin: per frame W*H float depth (metres, -inf invalid) then W*H*4 bytes RGBX.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "bundlefusion/bundlefusion.hpp"

using namespace bundlefusion;

struct FileSensor : RGBDSensor {
    std::vector<float> depth; std::vector<unsigned char> color; unsigned int frame = 0, numFrames, W, H; int cur = -1;
    FileSensor(const char* path, unsigned int w, unsigned int h, unsigned int n, float fx, float fy, float mx, float my) : numFrames(n), W(w), H(h) {
        std::memset(&m_desc, 0, sizeof m_desc);
        m_desc.depthWidth = m_desc.colorWidth = w; m_desc.depthHeight = m_desc.colorHeight = h;
        const mat4f I = mat4f::identity();
        mat4f K = I; K(0, 0) = fx; K(1, 1) = fy; K(0, 2) = mx; K(1, 2) = my;
        std::memcpy(m_desc.depthIntrinsics, K.m, 64); std::memcpy(m_desc.colorIntrinsics, K.m, 64);
        std::memcpy(m_desc.depthExtrinsics, I.m, 64); std::memcpy(m_desc.colorExtrinsics, I.m, 64);
        depth.resize((size_t)w * h * n); color.resize((size_t)w * h * 4 * n);
        FILE* f = std::fopen(path, "rb");
        if (!f) throw std::runtime_error("cannot open the frame file");
        for (unsigned int i = 0; i < n; ++i)
            if (std::fread(&depth[(size_t)w * h * i], 4, (size_t)w * h, f) != (size_t)w * h || std::fread(&color[(size_t)w * h * 4 * i], 1, (size_t)w * h * 4, f) != (size_t)w * h * 4)
                throw std::runtime_error("frame file too short");
        std::fclose(f);
    }
    bool processDepth() override { if (frame >= numFrames) return false; cur = (int)frame++; return true; }
    bool processColor() override { return true; }
    const float* getDepthFloat() const override { return &depth[(size_t)W * H * cur]; }
    const unsigned char* getColorRGBX() const override { return &color[(size_t)W * H * 4 * cur]; }
};

int main(int argc, char** argv) {
    if (argc < 14) { std::printf("usage: %s frames.bin W H frames preroll voxel buckets blocks fx fy mx my deferred\n", argv[0]); return 2; }
    try {
        const unsigned int W = std::atoi(argv[2]), H = std::atoi(argv[3]), n = std::atoi(argv[4]), pre = std::atoi(argv[5]);
        GlobalAppState& gas = GlobalAppState::get();              // zParametersDefault.txt / zParametersBundlingDefault.txt values, then bench.py's overrides
        GlobalBundlingState& gbs = GlobalBundlingState::get();
        gas.s_integrationWidth = W; gas.s_integrationHeight = H;
        gas.s_SDFVoxelSize = (float)std::atof(argv[6]); gas.s_hashNumBuckets = std::atoi(argv[7]); gas.s_hashNumSDFBlocks = std::atoi(argv[8]);
        gbs.s_maxNumImages = std::max(n / 10 + 8, 16u);
        const bool deferred = std::atoi(argv[13]) != 0;
        FileSensor sensor(argv[1], W, H, n, (float)std::atof(argv[9]), (float)std::atof(argv[10]), (float)std::atof(argv[11]), (float)std::atof(argv[12]));
        CUDAImageManager imageManager(gas.s_integrationWidth, gas.s_integrationHeight, gbs.s_widthSIFT, gbs.s_heightSIFT, &sensor, /*storeFramesOnGPU=*/true);
        OnlineBundler bundler(&sensor, &imageManager);
        CUDASceneRepHashSDF sceneRep(CUDASceneRepHashSDF::parametersFromGlobalAppState(gas));
        sceneRep.setDeferredBatching(deferred);
        DepthCameraParams cam;
        const mat4f K = imageManager.getDepthIntrinsics();
        cam.fx = K(0, 0); cam.fy = K(1, 1); cam.mx = K(0, 2); cam.my = K(1, 2);
        cam.m_sensorDepthWorldMin = gas.s_renderDepthMin; cam.m_sensorDepthWorldMax = gas.s_renderDepthMax;
        cam.m_imageWidth = imageManager.getIntegrationWidth(); cam.m_imageHeight = imageManager.getIntegrationHeight();
        TrajectoryManager* tm = bundler.getTrajectoryManager();
        unsigned int done = 0, tracked = 0, nIn = 0, nDe = 0;
        std::chrono::steady_clock::time_point t0;
        for (;;) {
            if (done == pre) { (void)sceneRep.getHeapFreeCount(); t0 = std::chrono::steady_clock::now(); }      // (flushes and drains the volume's streams)
            if (!imageManager.process()) break;
            bundler.processInput();
            if (tm->getNumActiveOperations() < gas.s_maxFrameFixes) tm->generateUpdateLists();
            for (unsigned int fixes = 0; fixes < gas.s_maxFrameFixes; fixes++) {
                mat4f newT, oldT; unsigned int idx;
                if (tm->getTopFromDeIntegrateList(oldT, idx)) {
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.deIntegrate(oldT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr); nDe++;
                } else if (tm->getTopFromIntegrateList(newT, idx)) {
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.integrate(newT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr); nIn++;
                    tm->confirmIntegration(idx);
                } else if (tm->getTopFromReIntegrateList(oldT, newT, idx)) {
                    if (newT(0, 0) == -std::numeric_limits<float>::infinity()) continue;
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.deIntegrate(oldT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr); nDe++;
                    sceneRep.integrate(newT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr); nIn++;
                    tm->confirmIntegration(idx);
                } else break;
            }
            sceneRep.garbageCollect();
            mat4f T; unsigned int frameIdx; bool lost;
            if (bundler.getCurrentIntegrationFrame(T, frameIdx, lost)) {
                auto f = imageManager.getIntegrateFrame(frameIdx);
                sceneRep.integrate(T, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr); nIn++; tracked++;
                tm->addFrame(TrajectoryManager::TrajectoryFrame::Integrated, T, imageManager.getCurrFrameNumber());
            } else {
                mat4f inv; for (int i = 0; i < 16; ++i) inv.m[i] = -std::numeric_limits<float>::infinity();
                tm->addFrame(TrajectoryManager::TrajectoryFrame::NotIntegrated_NoTransform, inv, imageManager.getCurrFrameNumber());
            }
            bundler.process(gbs.s_numLocalNonLinIterations, gbs.s_numLocalLinIterations, gbs.s_numGlobalNonLinIterations, gbs.s_numGlobalLinIterations);
            done++;
        }
        const unsigned int heapFree = sceneRep.getHeapFreeCount();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"frames_timed\": %u, \"value\": %.3f, \"unit\": \"frames/s\", \"ms_per_frame\": %.4f, \"frames_tracked\": %u, \"integrate\": %u, \"deintegrate\": %u, \"heap_free\": %u, \"deferred_batching\": %s}\n",
                    done - pre, (done - pre) / dt, 1e3 * dt / (done - pre), tracked, nIn, nDe, heapFree, deferred ? "true" : "false");
    } catch (const std::exception& e) {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        return 1;
    }
    return 0;
}
