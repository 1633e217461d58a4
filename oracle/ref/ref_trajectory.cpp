// oracle/ref/ref_trajectory.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's TrajectoryManager (TrajectoryManager.cpp:8-200: addFrame, updateOptimizedTransform, generateUpdateLists, the three
// getTopFrom*List consumers, confirmIntegration, invalidateFrame) with the se(3) logarithm of PoseHelper.h:275-363 it ranks frames with,
// compiled from where they lie.  mLib is absent from the reference tree; shim/mlib_standin.h supplies its vector / matrix element
// arithmetic, shim/GlobalAppState.h the two parameters the constructor reads.  The Makefile drops the include of CUDAImageManager.h
// from a temporary copy of TrajectoryManager.h (nothing of it is used; it pulls the sensor classes).  cudaMemcpy is the shim's memcpy,
// so updateOptimizedTransform takes a host array.
#include "TrajectoryManager.h"
#include "GlobalAppState.h"

extern "C" {

void* ref_tm_create(unsigned int numMaxImages, unsigned int topNActive, float minPoseDistSqrt) {
    GlobalAppState::get().s_topNActive = topNActive;
    GlobalAppState::get().s_minPoseDistSqrt = minPoseDistSqrt;
    return new TrajectoryManager(numMaxImages);
}
void ref_tm_destroy(void* h) { delete (TrajectoryManager*)h; }
void ref_tm_add_frame(void* h, int type, const float* T16, unsigned int idx) {
    mat4f T; memcpy(T.matrix, T16, 64);
    ((TrajectoryManager*)h)->addFrame((TrajectoryManager::TrajectoryFrame::TYPE)type, T, idx);
}
void ref_tm_update_optimized_transform(void* h, const float* trajectory, unsigned int numFrames) {
    ((TrajectoryManager*)h)->updateOptimizedTransform((const float4x4*)trajectory, numFrames);
}
void ref_tm_generate_update_lists(void* h) { ((TrajectoryManager*)h)->generateUpdateLists(); }
void ref_tm_confirm_integration(void* h, unsigned int idx) { ((TrajectoryManager*)h)->confirmIntegration(idx); }
unsigned int ref_tm_num_active(void* h) { return ((TrajectoryManager*)h)->getNumActiveOperations(); }
int ref_tm_top_reintegrate(void* h, float* oldT, float* newT, unsigned int* idx) {
    mat4f a, b; bool f = ((TrajectoryManager*)h)->getTopFromReIntegrateList(a, b, *idx);
    memcpy(oldT, a.matrix, 64); memcpy(newT, b.matrix, 64); return f;
}
int ref_tm_top_integrate(void* h, float* T, unsigned int* idx) {
    mat4f a; bool f = ((TrajectoryManager*)h)->getTopFromIntegrateList(a, *idx);
    memcpy(T, a.matrix, 64); return f;
}
int ref_tm_top_deintegrate(void* h, float* T, unsigned int* idx) {
    mat4f a; bool f = ((TrajectoryManager*)h)->getTopFromDeIntegrateList(a, *idx);
    memcpy(T, a.matrix, 64); return f;
}
void ref_tm_frame(void* h, unsigned int idx, int* type, float* integrated, float* optimized, float* dist) {
    const auto& f = ((TrajectoryManager*)h)->getFrames()[idx];
    *type = (int)f.type; *dist = f.dist;
    memcpy(integrated, f.integratedTransform.matrix, 64); memcpy(optimized, f.optimizedTransform.matrix, 64);
}
// PoseHelper::MatrixToPose / PoseToMatrix (USE_LIE_SPACE variants, PoseHelper.h:332-426): pose = (translation part, rotation vector)
void ref_pose_matrix_to_pose(const float* T16, float* pose6) {
    mat4f T; memcpy(T.matrix, T16, 64);
    Pose p = PoseHelper::MatrixToPose(T);
    for (int i = 0; i < 6; ++i) pose6[i] = p[i];
}
void ref_pose_pose_to_matrix(const float* pose6, float* T16) {
    Pose p; for (int i = 0; i < 6; ++i) p[i] = pose6[i];
    mat4f T = PoseHelper::PoseToMatrix(p);
    memcpy(T16, T.matrix, 64);
}

}
