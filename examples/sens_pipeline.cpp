// Plays a recorded ".sens" file through the whole-loop C entry points (bf_pipeline_*: four streams, volume worker thread,
// one-frame detection look-ahead) and evaluates the optimised trajectory against the poses stored in the file.
// Build:  g++ -std=c++17 -I include examples/sens_pipeline.cpp -L bundlefusion_amd/lib -lbf_hip -Wl,-rpath,$PWD/bundlefusion_amd/lib -o sens_pipeline
// Run:    ./sens_pipeline sequence.sens [zParametersDefault.txt zParametersBundlingDefault.txt]
#include <cstdio>
#include <vector>

#include "bundlefusion/bundlefusion.hpp"

using namespace bundlefusion;

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: %s sequence.sens [zParametersDefault.txt zParametersBundlingDefault.txt]\n", argv[0]); return 0; }
    try {
        GlobalAppState& gas = GlobalAppState::get();
        GlobalBundlingState& gbs = GlobalBundlingState::get();
        if (argc >= 4) { gas.readMembers(argv[2]); gbs.readMembers(argv[3]); }
        SensorDataReader sensor;
        sensor.createFirstConnected(argv[1]);
        if (argc < 4) { gas.s_integrationWidth = sensor.getDepthWidth(); gas.s_integrationHeight = sensor.getDepthHeight(); }
        gas.s_sensorIdx = 8;
        bf_pipeline* p = nullptr;
        check(bf_pipeline_create(&gas, &gbs, &sensor.desc(), &p));
        unsigned int frames = 0;
        while (sensor.processDepth() && sensor.processColor()) {
            int got = 0;
            check(bf_pipeline_process_frame(p, sensor.getDepthFloat(), sensor.getColorRGBX(), &got));     // buffers are free again on return
            if (!got) break;
            ++frames;
        }
        for (int k = 0; k < 5; ++k) check(bf_pipeline_process_end_of_sequence(p, nullptr));               // let the last solves and fixes finish
        check(bf_pipeline_synchronize(p));
        bf_online_bundler* ob = nullptr;
        check(bf_pipeline_get_online_bundler(p, &ob));
        bf_trajectory_manager* tm = nullptr;
        check(bf_online_bundler_get_trajectory_manager(ob, &tm));
        std::vector<mat4f> trajectory(frames);
        uint32_t n = 0;
        if (frames) check(bf_trajectory_manager_get_optimized_transforms(tm, trajectory[0].m, frames, &n));
        trajectory.resize(n);
        uint32_t nInt = 0, nDe = 0, nLocal = 0, nGlobal = 0;
        check(bf_pipeline_get_counters(p, &nInt, &nDe, &nLocal, &nGlobal));
        std::printf("%s: %u frames, %u integrations, %u de-integrations, %u local / %u global solves\n", sensor.getSensorName().c_str(), frames, nInt, nDe, nLocal, nGlobal);
        sensor.evaluateTrajectory(trajectory);
        check(bf_pipeline_destroy(p));
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
