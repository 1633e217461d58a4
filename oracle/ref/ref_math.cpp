// oracle/ref/ref_math.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// C entry points over the REFERENCE's own closed-form math, compiled from where it lies (paths under FriedLiver/Source):
//   Solver/LieDerivUtil.h      exp_rotation :50, ln_rotation :79, matrixToPose :135, poseToMatrix :160, evalLie_derivI/J :247/:277,
//                              computeLieUpdate :301
//   SiftGPU/cuda_svd3.h        svd :344 (McAdams et al. 3x3 SVD)
//   SiftGPU/cuda_EigenValue.h  computeEigenValues :9
//   SiftGPU/cuda_kabsch.h      kabsch :73, covarianceSVD :213, ComputeReprojection :380, filterKeyPointMatches :422 (the whole
//                              greedy Kabsch match filter of one image pair)
// Elementary functions (sqrt aside) come from glibc here; the oracle and the product take sin/cos/acos/atan2/exp from
// include/bf_detmath.h, so comparisons through those stages carry the few-ulp bound stated in tests/test_ref_pin_cpu.py.
#include "Solver/LieDerivUtil.h"
#include "SiftGPU/cuda_kabsch.h"

extern "C" {

void ref_exp_rotation(const float* w, float* R9) {
    const float3x3 R = exp_rotation(make_float3(w[0], w[1], w[2]));
    memcpy(R9, R.entries, 36);
}
void ref_ln_rotation(const float* R9, float* w) {
    const float3 r = ln_rotation(float3x3(R9));
    w[0] = r.x; w[1] = r.y; w[2] = r.z;
}
void ref_matrix_to_pose(const float* M16, float* rot, float* trans) {
    float3 r, t;
    matrixToPose(float4x4(M16), r, t);
    rot[0] = r.x; rot[1] = r.y; rot[2] = r.z; trans[0] = t.x; trans[1] = t.y; trans[2] = t.z;
}
void ref_pose_to_matrix(const float* rot, const float* trans, float* M16) {
    const float4x4 M = poseToMatrix(make_float3(rot[0], rot[1], rot[2]), make_float3(trans[0], trans[1], trans[2]));
    memcpy(M16, M.entries, 64);
}
void ref_lie_update(const float* dW, const float* dT, const float* w, const float* t, float* nw, float* nt) {
    float3 a, b;
    computeLieUpdate(make_float3(dW[0], dW[1], dW[2]), make_float3(dT[0], dT[1], dT[2]), make_float3(w[0], w[1], w[2]), make_float3(t[0], t[1], t[2]), a, b);
    nw[0] = a.x; nw[1] = a.y; nw[2] = a.z; nt[0] = b.x; nt[1] = b.y; nt[2] = b.z;
}
void ref_lie_deriv(int which, const float* A16, const float* D16, const float* p, float* out18) {      // 3x6, row-major
    const matNxM<3, 6> J = which == 0 ? evalLie_derivI(float4x4(A16), float4x4(D16), make_float3(p[0], p[1], p[2]))
                                      : evalLie_derivJ(float4x4(A16), float4x4(D16), make_float3(p[0], p[1], p[2]));
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) out18[r * 6 + c] = J(r, c);
}

void ref_svd3(const float* A, float* U, float* S, float* V) {
    svd(A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], A[8],
        U[0], U[1], U[2], U[3], U[4], U[5], U[6], U[7], U[8],
        S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8],
        V[0], V[1], V[2], V[3], V[4], V[5], V[6], V[7], V[8]);
}
void ref_eigenvalues3(const float* A9, float* ev) {
    const float3 e = computeEigenValues(float3x3(A9));
    ev[0] = e.x; ev[1] = e.y; ev[2] = e.z;
}
void ref_kabsch(const float* src, const float* tgt, unsigned int n, float* T16, float* ev) {
    float3 e;
    const float4x4 T = kabsch((volatile float3*)src, (volatile float3*)tgt, n, e);
    memcpy(T16, T.entries, 64);
    ev[0] = e.x; ev[1] = e.y; ev[2] = e.z;
}
// filterKeyPointMatches (cuda_kabsch.h:422-502): idx / dist are overwritten with the kept matches like in the reference
unsigned int ref_filter_keypoint_matches(const float* keys, unsigned int* idx, float* dist, unsigned int numRaw, const float* Kinv16,
                                         unsigned int minMatches, float maxRes2, float* T16) {
    float4x4 T;
    const unsigned int n = filterKeyPointMatches((const SIFTKeyPoint*)keys, (volatile uint2*)idx, (volatile float*)dist, numRaw, T, float4x4(Kinv16),
                                                 minMatches, maxRes2);
    memcpy(T16, T.entries, 64);
    return n;
}

}  // extern "C"
