// Headless driver written against the reference's class names (include/bundlefusion/bundlefusion.hpp):
// the serial body of DepthSensing.cpp's OnD3D11FrameRender without DirectX.  Build:
//   g++ -std=c++17 -I include examples/headless_driver.cpp -L bundlefusion_amd/lib -lbf_hip -Wl,-rpath,$PWD/bundlefusion_amd/lib -o headless_driver
// Run:  ./headless_driver zParametersDefault.txt zParametersBundlingDefault.txt
// With s_sensorIdx = 8 the file named by s_binaryDumpSensorFile is played (SensorDataReader, FriedLiver.cpp:89-94) and the
// optimised trajectory is evaluated against the poses stored in it; any other sensor index feeds a constant-depth dummy sensor.
#include <cstdio>
#include <limits>

#include "bundlefusion/bundlefusion.hpp"

using namespace bundlefusion;

struct DummySensor : RGBDSensor {           // RGBDSensor contract: host float depth (metres, -inf invalid) + RGBX8 colour
    std::vector<float> depth; std::vector<unsigned char> color; unsigned int frame = 0, numFrames;
    DummySensor(unsigned int w, unsigned int h, unsigned int n) : numFrames(n) {
        std::memset(&m_desc, 0, sizeof m_desc);
        m_desc.depthWidth = m_desc.colorWidth = w; m_desc.depthHeight = m_desc.colorHeight = h;
        const mat4f I = mat4f::identity();
        mat4f K = I; K(0, 0) = K(1, 1) = 583.0f * w / 640.0f; K(0, 2) = (w - 1) / 2.0f; K(1, 2) = (h - 1) / 2.0f;
        std::memcpy(m_desc.depthIntrinsics, K.m, 64); std::memcpy(m_desc.colorIntrinsics, K.m, 64);
        std::memcpy(m_desc.depthExtrinsics, I.m, 64); std::memcpy(m_desc.colorExtrinsics, I.m, 64);
        depth.assign((size_t)w * h, 2.0f); color.assign((size_t)w * h * 4, 128);
    }
    bool processDepth() override { return frame++ < numFrames; }
    bool processColor() override { return true; }
    const float* getDepthFloat() const override { return depth.data(); }
    const unsigned char* getColorRGBX() const override { return color.data(); }
};

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: %s zParametersDefault.txt zParametersBundlingDefault.txt\n", argv[0]); return 0; }
    try {
        GlobalAppState::get().readMembers(argv[1]);
        GlobalBundlingState::get().readMembers(argv[2]);
        const GlobalAppState& gas = GlobalAppState::get();
        const GlobalBundlingState& gbs = GlobalBundlingState::get();
        DummySensor dummy(640, 480, 30);
        SensorDataReader reader;
        const bool useFile = gas.s_sensorIdx == 8;
        if (useFile) reader.createFirstConnected();
        RGBDSensor& sensor = useFile ? static_cast<RGBDSensor&>(reader) : static_cast<RGBDSensor&>(dummy);
        CUDAImageManager imageManager(gas.s_integrationWidth, gas.s_integrationHeight, gbs.s_widthSIFT, gbs.s_heightSIFT, &sensor, /*storeFramesOnGPU=*/true);
        OnlineBundler bundler(&sensor, &imageManager);
        CUDASceneRepHashSDF sceneRep(CUDASceneRepHashSDF::parametersFromGlobalAppState(gas));
        DepthCameraParams cam;                                  // DepthSensing.cpp:636-643
        const mat4f K = imageManager.getDepthIntrinsics();
        cam.fx = K(0, 0); cam.fy = K(1, 1); cam.mx = K(0, 2); cam.my = K(1, 2);
        cam.m_sensorDepthWorldMin = gas.s_renderDepthMin; cam.m_sensorDepthWorldMax = gas.s_renderDepthMax;
        cam.m_imageWidth = imageManager.getIntegrationWidth(); cam.m_imageHeight = imageManager.getIntegrationHeight();
        TrajectoryManager* tm = bundler.getTrajectoryManager();
        for (;;) {
            const bool bGotDepth = imageManager.process();
            if (!bGotDepth) break;
            bundler.processInput();
            // reintegrate(): DepthSensing.cpp:854-902
            if (tm->getNumActiveOperations() < gas.s_maxFrameFixes) tm->generateUpdateLists();
            for (unsigned int fixes = 0; fixes < gas.s_maxFrameFixes; fixes++) {
                mat4f newT, oldT; unsigned int idx;
                if (tm->getTopFromDeIntegrateList(oldT, idx)) {
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.deIntegrate(oldT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr);
                } else if (tm->getTopFromIntegrateList(newT, idx)) {
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.integrate(newT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr);
                    tm->confirmIntegration(idx);
                } else if (tm->getTopFromReIntegrateList(oldT, newT, idx)) {
                    if (newT(0, 0) == -std::numeric_limits<float>::infinity()) continue;
                    auto f = imageManager.getIntegrateFrame(idx);
                    sceneRep.deIntegrate(oldT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr);
                    sceneRep.integrate(newT, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr);
                    tm->confirmIntegration(idx);
                } else break;
            }
            sceneRep.garbageCollect();
            mat4f T; unsigned int frameIdx; bool bGlobalTrackingLost;
            if (bundler.getCurrentIntegrationFrame(T, frameIdx, bGlobalTrackingLost)) {
                auto f = imageManager.getIntegrateFrame(frameIdx);
                sceneRep.integrate(T, DepthCameraData(f.getDepthFrameGPU(), f.getColorFrameGPU()), cam, nullptr);
                tm->addFrame(TrajectoryManager::TrajectoryFrame::Integrated, T, imageManager.getCurrFrameNumber());
            } else {
                mat4f inv; for (int i = 0; i < 16; ++i) inv.m[i] = -std::numeric_limits<float>::infinity();
                tm->addFrame(TrajectoryManager::TrajectoryFrame::NotIntegrated_NoTransform, inv, imageManager.getCurrFrameNumber());
            }
            bundler.process(gbs.s_numLocalNonLinIterations, gbs.s_numLocalLinIterations, gbs.s_numGlobalNonLinIterations, gbs.s_numGlobalLinIterations);
            std::printf("<<< [Frame: %u ] %u >>>\n", imageManager.getCurrFrameNumber(), sceneRep.getHeapFreeCount());
        }
        if (useFile) {                                          // DepthSensing.cpp:905-910 (StopScanningAndExit): evaluate against the recorded trajectory
            std::vector<mat4f> trajectory;
            tm->getOptimizedTransforms(trajectory);
            reader.evaluateTrajectory(trajectory);
        }
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
