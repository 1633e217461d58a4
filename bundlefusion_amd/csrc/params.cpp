// Parameter files of the reference (zParametersDefault.txt, zParametersBundlingDefault.txt) read into plain structs.
// Grammar as accepted by ml::ParameterFile (mLib, not in the tree — behaviour taken from the two shipped files and the
// X-macro readers GlobalAppState.h:128-136 / GlobalBundlingState.h:90-98): one `name = value;` per line, `//` starts a
// comment, values are booleans (true/false), numbers with an optional f suffix, quoted strings, or space separated
// vectors.  Unknown names are ignored; names that never appear keep the default and are counted as missing.
#include <cstdio>
#include <limits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/bf_pipeline.h"
#include "bf_internal.h"

namespace {

typedef std::map<std::string, std::string> KV;

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

bool parseFile(const char* filename, KV& kv) {
    std::ifstream in(filename);
    if (!in.is_open()) return false;
    std::string line;
    while (std::getline(in, line)) {
        bool inStr = false;
        size_t cut = std::string::npos;
        for (size_t i = 0; i + 1 < line.size(); ++i) {
            if (line[i] == '"') inStr = !inStr;
            if (!inStr && line[i] == '/' && line[i + 1] == '/') { cut = i; break; }
        }
        if (cut != std::string::npos) line = line.substr(0, cut);
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string name = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        const size_t semi = val.find_last_of(';');
        if (semi != std::string::npos) val = trim(val.substr(0, semi));
        if (!name.empty()) kv[name] = val;
    }
    return true;
}

std::vector<std::string> tokens(const std::string& v) {
    std::vector<std::string> out;
    std::istringstream ss(v);
    std::string t;
    while (ss >> t) out.push_back(t);
    return out;
}
float toF(const std::string& t) { return strtof(t.c_str(), nullptr); }      // "0.06f" -> 0.06 (strtof stops at the suffix)
bool toB(const std::string& t) { return t == "true" || t == "1"; }

struct Reader {
    const KV& kv; uint32_t missing = 0;
    explicit Reader(const KV& k) : kv(k) {}
    const std::string* find(const char* n) { auto it = kv.find(n); if (it == kv.end()) { missing++; return nullptr; } return &it->second; }
    void u(const char* n, uint32_t& d) { if (auto* s = find(n)) d = (uint32_t)strtoul(s->c_str(), nullptr, 10); }
    void f(const char* n, float& d) { if (auto* s = find(n)) d = toF(*s); }
    void b(const char* n, int32_t& d) { if (auto* s = find(n)) d = toB(*s) ? 1 : 0; }
    void fv(const char* n, float* d, int k) { if (auto* s = find(n)) { auto t = tokens(*s); for (int i = 0; i < k && i < (int)t.size(); ++i) d[i] = toF(t[i]); } }
    void iv(const char* n, int32_t* d, int k) { if (auto* s = find(n)) { auto t = tokens(*s); for (int i = 0; i < k && i < (int)t.size(); ++i) d[i] = (int32_t)strtol(t[i].c_str(), nullptr, 10); } }
};

void readApp(Reader& r, bf_global_app_state& g) {
    r.u("s_sensorIdx", g.s_sensorIdx);
    r.u("s_integrationWidth", g.s_integrationWidth); r.u("s_integrationHeight", g.s_integrationHeight);
    r.u("s_maxFrameFixes", g.s_maxFrameFixes); r.u("s_topNActive", g.s_topNActive); r.f("s_minPoseDistSqrt", g.s_minPoseDistSqrt);
    r.f("s_sensorDepthMax", g.s_sensorDepthMax); r.f("s_sensorDepthMin", g.s_sensorDepthMin);
    r.f("s_renderDepthMax", g.s_renderDepthMax); r.f("s_renderDepthMin", g.s_renderDepthMin);
    r.u("s_hashNumBuckets", g.s_hashNumBuckets); r.u("s_hashNumSDFBlocks", g.s_hashNumSDFBlocks);
    r.u("s_hashMaxCollisionLinkedListSize", g.s_hashMaxCollisionLinkedListSize);
    r.f("s_SDFVoxelSize", g.s_SDFVoxelSize); r.f("s_SDFTruncation", g.s_SDFTruncation); r.f("s_SDFTruncationScale", g.s_SDFTruncationScale);
    r.f("s_SDFMaxIntegrationDistance", g.s_SDFMaxIntegrationDistance);
    r.u("s_SDFIntegrationWeightSample", g.s_SDFIntegrationWeightSample); r.u("s_SDFIntegrationWeightMax", g.s_SDFIntegrationWeightMax);
    r.f("s_colorSigmaD", g.s_colorSigmaD); r.f("s_colorSigmaR", g.s_colorSigmaR); r.b("s_colorFilter", g.s_colorFilter);
    r.b("s_integrationEnabled", g.s_integrationEnabled); r.b("s_garbageCollectionEnabled", g.s_garbageCollectionEnabled);
    r.b("s_reconstructionEnabled", g.s_reconstructionEnabled); r.b("s_streamingEnabled", g.s_streamingEnabled);
    r.b("s_bUseCameraCalibration", g.s_bUseCameraCalibration); r.b("s_binaryDumpSensorUseTrajectory", g.s_binaryDumpSensorUseTrajectory);
    r.u("s_garbageCollectionStarve", g.s_garbageCollectionStarve);
    r.fv("s_streamingVoxelExtents", g.s_streamingVoxelExtents, 3); r.iv("s_streamingGridDimensions", g.s_streamingGridDimensions, 3);
    r.iv("s_streamingMinGridPos", g.s_streamingMinGridPos, 3); r.u("s_streamingInitialChunkListSize", g.s_streamingInitialChunkListSize);
    r.u("s_rayCastWidth", g.s_rayCastWidth); r.u("s_rayCastHeight", g.s_rayCastHeight);
    r.f("s_SDFRayIncrementFactor", g.s_SDFRayIncrementFactor); r.f("s_SDFRayThresSampleDistFactor", g.s_SDFRayThresSampleDistFactor);
    r.f("s_SDFRayThresDistFactor", g.s_SDFRayThresDistFactor); r.b("s_SDFUseGradients", g.s_SDFUseGradients);
    if (auto* s = r.find("s_numSolveFramesBeforeExit")) g.s_numSolveFramesBeforeExit = (uint32_t)strtol(s->c_str(), nullptr, 10);   // may be -1
    if (auto* s = r.find("s_binaryDumpSensorFile")) {                                    // a quoted string
        std::string v = *s;
        if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
        snprintf(g.s_binaryDumpSensorFile, sizeof g.s_binaryDumpSensorFile, "%s", v.c_str());
    }
}

void readBundling(Reader& r, bf_global_bundling_state& g) {
    r.b("s_enableGlobalTimings", g.s_enableGlobalTimings); r.b("s_enablePerFrameTimings", g.s_enablePerFrameTimings);
    r.u("s_maxNumImages", g.s_maxNumImages); r.u("s_submapSize", g.s_submapSize); r.u("s_widthSIFT", g.s_widthSIFT); r.u("s_heightSIFT", g.s_heightSIFT);
    r.u("s_maxNumKeysPerImage", g.s_maxNumKeysPerImage);
    r.u("s_numLocalNonLinIterations", g.s_numLocalNonLinIterations); r.u("s_numLocalLinIterations", g.s_numLocalLinIterations);
    r.u("s_numGlobalNonLinIterations", g.s_numGlobalNonLinIterations); r.u("s_numGlobalLinIterations", g.s_numGlobalLinIterations);
    r.u("s_downsampledWidth", g.s_downsampledWidth); r.u("s_downsampledHeight", g.s_downsampledHeight);
    r.f("s_verifySiftErrThresh", g.s_verifySiftErrThresh); r.f("s_verifySiftCorrThresh", g.s_verifySiftCorrThresh);
    r.f("s_projCorrDistThres", g.s_projCorrDistThres); r.f("s_projCorrNormalThres", g.s_projCorrNormalThres); r.f("s_projCorrColorThresh", g.s_projCorrColorThresh);
    r.f("s_surfAreaPcaThresh", g.s_surfAreaPcaThresh);
    r.b("s_recordSolverConvergence", g.s_recordSolverConvergence); r.b("s_erodeSIFTdepth", g.s_erodeSIFTdepth);
    r.f("s_verifyOptErrThresh", g.s_verifyOptErrThresh); r.f("s_verifyOptCorrThresh", g.s_verifyOptCorrThresh);
    r.b("s_verbose", g.s_verbose); r.b("s_sendUplinkFeedbackImage", g.s_sendUplinkFeedbackImage);
    r.f("s_depthSigmaD", g.s_depthSigmaD); r.f("s_depthSigmaR", g.s_depthSigmaR); r.b("s_depthFilter", g.s_depthFilter);
    r.u("s_minNumMatchesLocal", g.s_minNumMatchesLocal); r.u("s_minNumMatchesGlobal", g.s_minNumMatchesGlobal);
    r.b("s_useComprehensiveFrameInvalidation", g.s_useComprehensiveFrameInvalidation);
    r.f("s_maxKabschResidual2", g.s_maxKabschResidual2); r.f("s_minKeyScale", g.s_minKeyScale); r.f("s_siftMatchThresh", g.s_siftMatchThresh);
    r.f("s_siftMatchRatioMaxLocal", g.s_siftMatchRatioMaxLocal); r.f("s_siftMatchRatioMaxGlobal", g.s_siftMatchRatioMaxGlobal);
    r.b("s_useLocalVerify", g.s_useLocalVerify); r.b("s_useLocalDense", g.s_useLocalDense);
    r.u("s_numOptPerResidualRemoval", g.s_numOptPerResidualRemoval);
    r.f("s_colorDownSigma", g.s_colorDownSigma); r.f("s_depthDownSigmaD", g.s_depthDownSigmaD); r.f("s_depthDownSigmaR", g.s_depthDownSigmaR);
    r.f("s_optMaxResThresh", g.s_optMaxResThresh); r.f("s_denseDistThresh", g.s_denseDistThresh); r.f("s_denseNormalThresh", g.s_denseNormalThresh);
    r.f("s_denseColorThresh", g.s_denseColorThresh); r.f("s_denseColorGradientMin", g.s_denseColorGradientMin);
    r.f("s_denseDepthMin", g.s_denseDepthMin); r.f("s_denseDepthMax", g.s_denseDepthMax);
    r.u("s_denseOverlapCheckSubsampleFactor", g.s_denseOverlapCheckSubsampleFactor);
}

}  // namespace

extern "C" {

int bf_global_app_state_default(bf_global_app_state* g) {           // zParametersDefault.txt
    BF_REQUIRE(g, "null argument");
    memset(g, 0, sizeof *g);
    g->s_sensorIdx = 7;
    g->s_integrationWidth = 320; g->s_integrationHeight = 240;
    g->s_maxFrameFixes = 10; g->s_topNActive = 30; g->s_minPoseDistSqrt = 0.0f;
    g->s_sensorDepthMax = 4.0f; g->s_sensorDepthMin = 0.1f; g->s_renderDepthMax = 4.0f; g->s_renderDepthMin = 0.1f;
    g->s_hashNumBuckets = 800000; g->s_hashNumSDFBlocks = 200000; g->s_hashMaxCollisionLinkedListSize = 7;
    g->s_SDFVoxelSize = 0.010f; g->s_SDFTruncation = 0.06f; g->s_SDFTruncationScale = 0.02f; g->s_SDFMaxIntegrationDistance = 3.0f;
    g->s_SDFIntegrationWeightSample = 1; g->s_SDFIntegrationWeightMax = 99999999;
    g->s_colorSigmaD = 2.0f; g->s_colorSigmaR = 0.1f; g->s_colorFilter = 0;
    g->s_integrationEnabled = 1; g->s_garbageCollectionEnabled = 1; g->s_reconstructionEnabled = 1; g->s_streamingEnabled = 0;
    g->s_bUseCameraCalibration = 0; g->s_binaryDumpSensorUseTrajectory = 0; g->s_garbageCollectionStarve = 0;
    g->s_streamingVoxelExtents[0] = g->s_streamingVoxelExtents[1] = g->s_streamingVoxelExtents[2] = 1.0f;
    g->s_streamingGridDimensions[0] = g->s_streamingGridDimensions[1] = g->s_streamingGridDimensions[2] = 257;
    g->s_streamingMinGridPos[0] = g->s_streamingMinGridPos[1] = g->s_streamingMinGridPos[2] = -128;
    g->s_streamingInitialChunkListSize = 2000;
    g->s_numSolveFramesBeforeExit = 30;
    g->s_rayCastWidth = 320; g->s_rayCastHeight = 240;
    g->s_SDFRayIncrementFactor = 0.8f; g->s_SDFRayThresSampleDistFactor = 50.5f; g->s_SDFRayThresDistFactor = 50.0f; g->s_SDFUseGradients = 0;
    snprintf(g->s_binaryDumpSensorFile, sizeof g->s_binaryDumpSensorFile, "%s", "../data/sequence.sens");
    return BF_OK;
}

int bf_global_bundling_state_default(bf_global_bundling_state* g) {  // zParametersBundlingDefault.txt
    BF_REQUIRE(g, "null argument");
    memset(g, 0, sizeof *g);
    g->s_maxNumImages = 1200; g->s_submapSize = 10; g->s_widthSIFT = 640; g->s_heightSIFT = 480; g->s_maxNumKeysPerImage = 1024;
    g->s_numLocalNonLinIterations = 2; g->s_numLocalLinIterations = 100; g->s_numGlobalNonLinIterations = 3; g->s_numGlobalLinIterations = 150;
    g->s_downsampledWidth = 80; g->s_downsampledHeight = 60;
    g->s_verifySiftErrThresh = 0.075f; g->s_verifySiftCorrThresh = 0.02f; g->s_projCorrDistThres = 0.15f; g->s_projCorrNormalThres = 0.97f;
    g->s_projCorrColorThresh = 0.1f; g->s_surfAreaPcaThresh = 0.032f;
    g->s_recordSolverConvergence = 0; g->s_erodeSIFTdepth = 1;
    g->s_verifyOptErrThresh = 0.05f; g->s_verifyOptCorrThresh = 0.001f;
    g->s_verbose = 0; g->s_sendUplinkFeedbackImage = 1;
    g->s_depthSigmaD = 2.0f; g->s_depthSigmaR = 0.05f; g->s_depthFilter = 1;
    g->s_minNumMatchesLocal = 5; g->s_minNumMatchesGlobal = 5; g->s_useComprehensiveFrameInvalidation = 1;
    g->s_maxKabschResidual2 = 0.0004f; g->s_minKeyScale = 3.0f; g->s_siftMatchThresh = 0.7f;
    g->s_siftMatchRatioMaxLocal = 0.8f; g->s_siftMatchRatioMaxGlobal = 0.8f;
    g->s_useLocalVerify = 1; g->s_useLocalDense = 1; g->s_numOptPerResidualRemoval = 1;
    g->s_colorDownSigma = 2.5f; g->s_depthDownSigmaD = 1.0f; g->s_depthDownSigmaR = 0.05f;
    g->s_optMaxResThresh = 0.08f; g->s_denseDistThresh = 0.15f; g->s_denseNormalThresh = 0.97f; g->s_denseColorThresh = 0.1f;
    g->s_denseColorGradientMin = 0.005f; g->s_denseDepthMin = 0.5f; g->s_denseDepthMax = 4.0f; g->s_denseOverlapCheckSubsampleFactor = 4;
    return BF_OK;
}

int bf_global_app_state_read(const char* filename, bf_global_app_state* out, uint32_t* numMissing) {
    BF_REQUIRE(filename && out, "null argument");
    KV kv;
    if (!parseFile(filename, kv)) { bf::set_error("cannot open parameter file %s", filename); return BF_ERR_INVALID_ARG; }
    bf_global_app_state_default(out);
    Reader r(kv);
    readApp(r, *out);
    if (numMissing) *numMissing = r.missing;
    return BF_OK;
}

int bf_global_bundling_state_read(const char* filename, bf_global_bundling_state* out, uint32_t* numMissing) {
    BF_REQUIRE(filename && out, "null argument");
    KV kv;
    if (!parseFile(filename, kv)) { bf::set_error("cannot open parameter file %s", filename); return BF_ERR_INVALID_ARG; }
    bf_global_bundling_state_default(out);
    Reader r(kv);
    readBundling(r, *out);
    if (numMissing) *numMissing = r.missing;
    return BF_OK;
}

int bf_write_processed_summary(const char* path, uint32_t heapFreeCount, const float* T, uint32_t numTransforms, int aborted, int* validOut) {      // DepthSensing.cpp:921-957
    BF_REQUIRE(path && (T || numTransforms == 0 || aborted), "null argument");
    FILE* f = fopen(path, "w");
    if (!f) { bf::set_error("cannot write %s", path); return BF_ERR_INVALID_ARG; }
    bool valid = false;
    if (aborted) fprintf(f, "valid = false\nABORTED\n");
    else {
        valid = heapFreeCount >= 800;                                        // "probably a messed up reconstruction (used up all the heap...)"
        uint32_t numValid = 0;
        for (uint32_t i = 0; i < numTransforms; ++i) if (T[16 * (size_t)i] != -std::numeric_limits<float>::infinity()) ++numValid;      // PoseHelper::countNumValidTransforms
        if (numValid < (uint32_t)std::lround(0.5f * (float)numTransforms)) valid = false;
        fprintf(f, "valid = %s\nheapFreeCount = %u\nnumValidOptTransforms = %u\nnumTransforms = %u\n", valid ? "true" : "false", heapFreeCount, numValid, numTransforms);
    }
    fclose(f);
    if (validOut) *validOut = valid ? 1 : 0;
    return BF_OK;
}

int bf_ray_cast_params_from_global_app_state(const bf_global_app_state* gas, const float K[16], bf_ray_cast_params* p) {   // CUDARayCastSDF.h:24-52
    BF_REQUIRE(gas && K && p, "null argument");
    memset(p, 0, sizeof *p);
    float fx = K[0], fy = K[5], mx = K[2], my = K[6];
    if (gas->s_rayCastWidth != gas->s_integrationWidth || gas->s_rayCastHeight != gas->s_integrationHeight) {
        fx *= (float)gas->s_rayCastWidth / (float)gas->s_integrationWidth;
        fy *= (float)gas->s_rayCastHeight / (float)gas->s_integrationHeight;
        mx *= (float)(gas->s_rayCastWidth - 1) / (float)(gas->s_integrationWidth - 1);
        my *= (float)(gas->s_rayCastHeight - 1) / (float)(gas->s_integrationHeight - 1);
    }
    p->m_width = gas->s_rayCastWidth; p->m_height = gas->s_rayCastHeight;
    p->fx = fx; p->fy = fy; p->mx = mx; p->my = my;
    p->m_minDepth = gas->s_renderDepthMin; p->m_maxDepth = gas->s_renderDepthMax;
    p->m_rayIncrement = gas->s_SDFRayIncrementFactor * gas->s_SDFTruncation;
    p->m_thresSampleDist = gas->s_SDFRayThresSampleDistFactor * p->m_rayIncrement;
    p->m_thresDist = gas->s_SDFRayThresDistFactor * p->m_rayIncrement;
    p->m_useGradients = gas->s_SDFUseGradients;
    p->m_maxNumVertices = gas->s_hashNumSDFBlocks * 6;
    for (int i = 0; i < 16; ++i) p->m_viewMatrix[i] = p->m_viewMatrixInverse[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return BF_OK;
}

}  // extern "C"
