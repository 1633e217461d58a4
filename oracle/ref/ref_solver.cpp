// oracle/ref/ref_solver.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's Gauss-Newton / PCG bundling solver, Solver/SolverBundling.cu (BuildDenseSystem :308-471, Initialization :796,
// PCGIteration :1024-1108, solveBundlingStub :1137-1220, EvalResidual :595, evalMaxResidual :552, countHighResiduals :670,
// buildVariablesToCorrespondencesTableCUDA :1250) with its device headers SolverBundlingEquationsLie.h, SolverBundlingDenseUtil.h,
// SolverBundlingUtil.h, LieDerivUtil.h, ICPUtil.h, compiled from where they lie and run through the serial block emulator (warp
// reductions are lock-step fibers; float atomics add in thread-index order).  This file allocates and fills SolverInput /
// SolverState / SolverParameters the way CUDASolverBundling does (CUDASolverBundling.cpp:20-110 constructor, :187-284 solve,
// :286-292 buildVariablesToCorrespondencesTable, :454-476 useVerification).  The host class itself is compiled too since (ref_sba.cpp, over an mLib stand-in): on the same problems it
// produces bit-identical poses and energies to this driver (tests/test_ref_pin_cpu.py).
#include "SolverBundling.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)

#include <limits>
#include <vector>

struct ref_solver_params {             // GlobalBundlingState values the constructor reads (CUDASolverBundling.cpp:97-104)
    float denseDistThresh, denseNormalThresh, denseColorThresh, denseColorGradientMin, denseDepthMin, denseDepthMax;
    unsigned int denseOverlapCheckSubsampleFactor;
};
struct ref_cache_frame { float* depth; float* campos; float* intensity; float* derivs; unsigned char* normalsU; float* normals; };

static int g_lastNumDensePairs = -1, g_lastCorrCount = -1; static float g_lastSumResidual = 0.0f;

extern "C" {

// diagnostics of the last ref_solver_solve: overlapping image pairs of the last dense build (d_numDenseOverlappingImages), dense depth correspondences and their
// residual sum (d_corrCount, d_sumResidual)
void ref_solver_last_dense_stats(int* numPairs, int* corrCount, float* sumResidual) { *numPairs = g_lastNumDensePairs; *corrCount = g_lastCorrCount; *sumResidual = g_lastSumResidual; }

// returns the number of Gauss-Newton iterations recorded in convergence[] (nNonLin + 1 entries, -1 where an early out skipped them)
int ref_solver_solve(void* corrEntryJ, unsigned int numCorr, const int* validImages, unsigned int numImages, unsigned int maxImages, unsigned int maxResiduals,
                     unsigned int nNonLin, unsigned int nLin, const float* wSparse, const float* wDenseDepth, const float* wDenseColor, unsigned int numWeights,
                     const ref_cache_frame* cacheFrames, unsigned int cacheW, unsigned int cacheH, const float* cacheIntrinsics4, int usePairwiseDense,
                     const ref_solver_params* gp, float* rot3N, float* trans3N, float* convergence, float* maxResidualOut, int* maxResidualIdxOut,
                     int* numEntriesPerRowOut, int* useVerificationOut) {
    nNonLin = std::min(nNonLin, numWeights);
    const unsigned int nv = maxImages;
    const unsigned int maxCorrPerImage = std::min(std::max(maxResiduals / maxImages, 1000u), 4000u);       // math::clamp, .cpp:39
    SolverState st; memset(&st, 0, sizeof st);
    std::vector<float3> dRot(nv), dTr(nv), rRot(nv), rTr(nv), zRot(nv), zTr(nv), pRot(nv), pTr(nv), Jp(std::max(maxResiduals, 1u)), ApR(nv), ApT(nv), prR(nv), prT(nv);
    std::vector<float> scanAlpha(2), rDotzOld(nv), sumResidual(1), sumResidualColor(1);
    std::vector<int> countHigh(1), corrCount(1), corrCountColor(1), numDense(1);
    std::vector<float> denseJtJ((size_t)36 * nv * nv), denseJtr((size_t)6 * nv);
    const unsigned int maxPairs = maxImages * (maxImages - 1) / 2;
    std::vector<float> denseCorrCounts(std::max(maxPairs, 1u));
    std::vector<uint2> denseOverlap(std::max(maxPairs, 1u));
    std::vector<float4x4> xT(maxImages), xTinv(maxImages);
    st.d_deltaRot = dRot.data(); st.d_deltaTrans = dTr.data(); st.d_rRot = rRot.data(); st.d_rTrans = rTr.data(); st.d_zRot = zRot.data(); st.d_zTrans = zTr.data();
    st.d_pRot = pRot.data(); st.d_pTrans = pTr.data(); st.d_Jp = Jp.data(); st.d_Ap_XRot = ApR.data(); st.d_Ap_XTrans = ApT.data(); st.d_scanAlpha = scanAlpha.data();
    st.d_rDotzOld = rDotzOld.data(); st.d_precondionerRot = prR.data(); st.d_precondionerTrans = prT.data(); st.d_sumResidual = sumResidual.data();
    st.d_countHighResidual = countHigh.data(); st.d_denseJtJ = denseJtJ.data(); st.d_denseJtr = denseJtr.data(); st.d_denseCorrCounts = denseCorrCounts.data();
    st.d_denseOverlappingImages = denseOverlap.data(); st.d_numDenseOverlappingImages = numDense.data(); st.d_corrCount = corrCount.data();
    st.d_corrCountColor = corrCountColor.data(); st.d_sumResidualColor = sumResidualColor.data(); st.d_xTransforms = xT.data(); st.d_xTransformInverses = xTinv.data();
    for (auto* v : {&dRot, &dTr, &rRot, &rTr, &zRot, &zTr, &pRot, &pTr}) memset(v->data(), -1, sizeof(float3) * nv);      // .cpp:205-212
    st.d_xRot = (float3*)rot3N; st.d_xTrans = (float3*)trans3N;
    const unsigned int nBlocks = (std::max(maxResiduals, 1u) + THREADS_PER_BLOCK - 1) / THREADS_PER_BLOCK;
    std::vector<float> dMaxRes(nBlocks), hMaxRes(nBlocks); std::vector<int> dMaxIdx(nBlocks), hMaxIdx(nBlocks);
    SolverStateAnalysis an; an.d_maxResidual = dMaxRes.data(); an.d_maxResidualIndex = dMaxIdx.data(); an.h_maxResidual = hMaxRes.data(); an.h_maxResidualIndex = hMaxIdx.data();

    SolverParameters par; memset(&par, 0, sizeof par);
    par.denseDistThresh = gp->denseDistThresh; par.denseNormalThresh = gp->denseNormalThresh; par.denseColorThresh = gp->denseColorThresh;
    par.denseColorGradientMin = gp->denseColorGradientMin; par.denseDepthMin = gp->denseDepthMin; par.denseDepthMax = gp->denseDepthMax;
    par.denseOverlapCheckSubsampleFactor = gp->denseOverlapCheckSubsampleFactor;
    par.nNonLinearIterations = nNonLin; par.nLinIterations = nLin;
    par.verifyOptDistThresh = 0.02f; par.verifyOptPercentThresh = 0.05f;                                  // .cpp:33-34
    par.highResidualThresh = std::numeric_limits<float>::infinity();
    par.weightSparse = wSparse[0]; par.weightDenseDepth = wDenseDepth[0]; par.weightDenseColor = wDenseColor[0];
    par.useDense = par.weightDenseDepth > 0 || par.weightDenseColor > 0;
    par.useDenseDepthAllPairwise = usePairwiseDense != 0;

    std::vector<int> varToCorr((size_t)maxImages * maxCorrPerImage), numEntriesPerRow(maxImages, 0);
    std::vector<CUDACachedFrame> frames;
    SolverInput in; memset(&in, 0, sizeof in);
    in.d_correspondences = (EntryJ*)corrEntryJ; in.d_variablesToCorrespondences = varToCorr.data(); in.d_numEntriesPerRow = numEntriesPerRow.data();
    in.numberOfImages = numImages; in.numberOfCorrespondences = numCorr; in.maxNumberOfImages = maxImages; in.maxCorrPerImage = maxCorrPerImage;
    in.maxNumDenseImPairs = maxPairs; in.weightsSparse = wSparse; in.weightsDenseDepth = wDenseDepth; in.weightsDenseColor = wDenseColor; in.d_validImages = validImages;
    if (cacheFrames) {
        frames.resize(numImages);
        for (unsigned int i = 0; i < numImages; ++i) {
            frames[i].d_depthDownsampled = cacheFrames[i].depth; frames[i].d_cameraposDownsampled = (float4*)cacheFrames[i].campos;
            frames[i].d_intensityDownsampled = cacheFrames[i].intensity; frames[i].d_intensityDerivsDownsampled = (float2*)cacheFrames[i].derivs;
            frames[i].d_normalsDownsampledUCHAR4 = (uchar4*)cacheFrames[i].normalsU; frames[i].d_normalsDownsampled = (float4*)cacheFrames[i].normals;
        }
        in.d_cacheFrames = frames.data(); in.denseDepthWidth = cacheW; in.denseDepthHeight = cacheH;
        in.intrinsics = make_float4(cacheIntrinsics4[0], cacheIntrinsics4[1], cacheIntrinsics4[2], cacheIntrinsics4[3]);
    } else {
        in.d_cacheFrames = NULL; in.denseDepthWidth = 0; in.denseDepthHeight = 0;
        in.intrinsics = make_float4(-std::numeric_limits<float>::infinity());
    }
    if (numCorr > 0) buildVariablesToCorrespondencesTableCUDA(in.d_correspondences, numCorr, maxCorrPerImage, in.d_variablesToCorrespondences, in.d_numEntriesPerRow, NULL);   // .cpp:286-292
    if (numEntriesPerRowOut) memcpy(numEntriesPerRowOut, numEntriesPerRow.data(), sizeof(int) * numImages);
    if (convergence) for (unsigned int i = 0; i <= nNonLin; ++i) convergence[i] = -1.0f;
    solveBundlingStub(in, st, par, an, convergence, NULL);
    g_lastNumDensePairs = numDense[0]; g_lastCorrCount = corrCount[0]; g_lastSumResidual = sumResidual[0];
    if (maxResidualOut) {                                                    // computeMaxResidual, .cpp:313-349 (weights = 1, 0, 0)
        SolverParameters p2 = par; p2.highResidualThresh = std::numeric_limits<float>::infinity();
        p2.weightSparse = 1.0f; p2.weightDenseDepth = 0.0f; p2.weightDenseColor = 0.0f;
        evalMaxResidual(in, st, an, p2, NULL);
        const unsigned int n = (numCorr + THREADS_PER_BLOCK - 1) / THREADS_PER_BLOCK;
        float mx = -1.0f; int idx = -1;
        for (unsigned int i = 0; i < n; ++i) if (an.d_maxResidual[i] > mx) { mx = an.d_maxResidual[i]; idx = an.d_maxResidualIndex[i]; }
        *maxResidualOut = mx; if (maxResidualIdxOut) *maxResidualIdxOut = idx;
    }
    if (useVerificationOut) {                                                // useVerification, .cpp:454-476
        SolverParameters p3; memset(&p3, 0, sizeof p3);
        p3.nNonLinearIterations = 0; p3.nLinIterations = 0; p3.verifyOptDistThresh = 0.02f; p3.verifyOptPercentThresh = 0.05f;
        SolverInput i3; memset(&i3, 0, sizeof i3);
        i3.d_correspondences = in.d_correspondences; i3.numberOfCorrespondences = numCorr; i3.maxNumberOfImages = maxImages; i3.maxCorrPerImage = maxCorrPerImage;
        const unsigned int numHigh = countHighResiduals(i3, st, p3, NULL);
        *useVerificationOut = ((float)numHigh / numCorr >= p3.verifyOptPercentThresh) ? 1 : 0;
    }
    return (int)nNonLin;
}

}  // extern "C"
