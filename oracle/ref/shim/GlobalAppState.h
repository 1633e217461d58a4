// oracle/ref/shim/GlobalAppState.h — TEST INFRASTRUCTURE ONLY.  The two parameters TrajectoryManager's constructor reads from the
// application singleton (TrajectoryManager.cpp:19-20; the real GlobalAppState.h needs mLib's ParameterFile and the sensor / DirectX
// headers).  Reached only by the build-time copies in the temporary directory: reference files that sit next to the real header keep
// finding that one first.
#ifndef BF_REF_SHIM_GLOBAL_APP_STATE_H
#define BF_REF_SHIM_GLOBAL_APP_STATE_H
class GlobalAppState {
public:
    unsigned int s_topNActive = 30;
    float s_minPoseDistSqrt = 0.0f;
    static GlobalAppState& get() { static GlobalAppState s; return s; }
};
#endif
