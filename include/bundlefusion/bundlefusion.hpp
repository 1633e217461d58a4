// Header-only C++ view of libbf_hip.so with the reference's class names and method signatures (the drop-in
// boundary of SURVEY.md §8b): CUDAImageManager, Bundler, OnlineBundler, TrajectoryManager, CUDASceneRepHashSDF,
// GlobalAppState / GlobalBundlingState.  Every method forwards to the C ABI of bf_hip.h / bf_pipeline.h; errors are
// thrown as std::runtime_error carrying bf_last_error() where the reference throws MLIB_EXCEPTION.
//
// Matrix arguments are `mat4f` = 16 floats, row-major (same memory as ml::mat4f / float4x4).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../bf_pipeline.h"
#include "../bf_sensordata.h"

namespace bundlefusion {

struct mat4f {
    float m[16];
    static mat4f identity() { mat4f r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return r; }
    float& operator()(int r, int c) { return m[r * 4 + c]; }
    float operator()(int r, int c) const { return m[r * 4 + c]; }
    const float* getData() const { return m; }
    float* getData() { return m; }
    // the handful of ml::mat4f operations the frame loop uses (g_transformWorld * transformation, getInverse(), setZero(-inf))
    mat4f operator*(const mat4f& o) const {
        mat4f r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float v = 0.0f; for (int k = 0; k < 4; ++k) v += m[i * 4 + k] * o.m[k * 4 + j]; r.m[i * 4 + j] = v; }
        return r;
    }
    void setZero(float v = 0.0f) { for (float& e : m) e = v; }
    void setIdentity() { *this = identity(); }
    mat4f getTranspose() const { mat4f r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i * 4 + j] = m[j * 4 + i]; return r; }
    mat4f getInverse() const {                                     // general 4x4 inverse by cofactors
        const float* a = m; mat4f r; float* o = r.m;
        o[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
        o[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
        o[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
        o[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
        o[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
        o[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
        o[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
        o[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
        o[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
        o[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
        o[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
        o[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
        o[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
        o[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
        o[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
        o[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
        const float det = a[0] * o[0] + a[1] * o[4] + a[2] * o[8] + a[3] * o[12];
        const float inv = 1.0f / det;
        for (float& e : r.m) e *= inv;
        return r;
    }
};
typedef mat4f float4x4;

inline void check(int rc) { if (rc != BF_OK) throw std::runtime_error(std::string("bundlefusion: ") + bf_last_error()); }

// ---- GlobalAppState / GlobalBundlingState (GlobalAppState.h:106-160, GlobalBundlingState.h:67-120): singletons filled from the two files
class GlobalAppState : public bf_global_app_state {
public:
    static GlobalAppState& get() { static GlobalAppState s; return s; }
    static GlobalAppState& getInstance() { return get(); }
    void readMembers(const std::string& parameterFile) { check(bf_global_app_state_read(parameterFile.c_str(), this, nullptr)); }
private:
    GlobalAppState() { bf_global_app_state_default(this); }
};
class GlobalBundlingState : public bf_global_bundling_state {
public:
    static GlobalBundlingState& get() { static GlobalBundlingState s; return s; }
    static GlobalBundlingState& getInstance() { return get(); }
    void readMembers(const std::string& parameterFile) { check(bf_global_bundling_state_read(parameterFile.c_str(), this, nullptr)); }
private:
    GlobalBundlingState() { bf_global_bundling_state_default(this); }
};

// ---- RGBDSensor accessor contract (RGBDSensor.h:25-61); derive and fill the host buffers in processDepth/processColor
class RGBDSensor {
public:
    virtual ~RGBDSensor() { if (m_recWriter) bf_sensor_data_writer_close(m_recWriter); if (!m_recTmp.empty()) std::remove(m_recTmp.c_str()); }
    virtual bool processDepth() = 0;
    virtual bool processColor() = 0;
    virtual const float* getDepthFloat() const = 0;               // metres, -inf invalid
    virtual const unsigned char* getColorRGBX() const = 0;        // 4 x u8 per pixel
    virtual std::string getSensorName() const { return "RGBDSensor"; }
    unsigned int getDepthWidth() const { return m_desc.depthWidth; }
    unsigned int getDepthHeight() const { return m_desc.depthHeight; }
    unsigned int getColorWidth() const { return m_desc.colorWidth; }
    unsigned int getColorHeight() const { return m_desc.colorHeight; }
    mat4f getDepthIntrinsics() const { mat4f m; std::memcpy(m.m, m_desc.depthIntrinsics, 64); return m; }
    mat4f getColorIntrinsics() const { mat4f m; std::memcpy(m.m, m_desc.colorIntrinsics, 64); return m; }
    mat4f getDepthExtrinsics() const { mat4f m; std::memcpy(m.m, m_desc.depthExtrinsics, 64); return m; }
    mat4f getColorExtrinsics() const { mat4f m; std::memcpy(m.m, m_desc.colorExtrinsics, 64); return m; }
    mat4f getDepthIntrinsicsInv() const { return getDepthIntrinsics().getInverse(); }
    mat4f getColorIntrinsicsInv() const { return getColorIntrinsics().getInverse(); }
    mat4f getDepthExtrinsicsInv() const { return getDepthExtrinsics().getInverse(); }
    mat4f getColorExtrinsicsInv() const { return getColorExtrinsics().getInverse(); }
    const bf_rgbd_sensor_desc& desc() const { return m_desc; }

    // ---- recording (RGBDSensor.cpp:264-312, 353-398, "modern .sens files"): every recordFrame() appends the current depth
    // (u16 = round(1000 * metres), invalid -> 0, zlib) and colour to a temporary .sens; saveRecordedFramesToFile attaches the
    // trajectory, drops frames without a pose and writes the final file.  Colour is JPEG-compressed like the reference's recordings
    // (TYPE_JPEG; baseline 4:4:4, quality 90 - mLib uses stb_image_write, an encoder's bytes are not pinned by anything).
    void recordFrame() {
        if (!m_recWriter) {
            bf_sensor_data_info info; std::memset(&info, 0, sizeof info);
            info.versionNumber = 4;
            std::snprintf(info.sensorName, sizeof info.sensorName, "%s", getSensorName().c_str());
            std::memcpy(info.colorIntrinsic, m_desc.colorIntrinsics, 64); std::memcpy(info.colorExtrinsic, m_desc.colorExtrinsics, 64);
            std::memcpy(info.depthIntrinsic, m_desc.depthIntrinsics, 64); std::memcpy(info.depthExtrinsic, m_desc.depthExtrinsics, 64);
            info.colorCompressionType = BF_SENS_COLOR_JPEG; info.depthCompressionType = BF_SENS_DEPTH_ZLIB_USHORT;
            info.colorWidth = m_desc.colorWidth; info.colorHeight = m_desc.colorHeight; info.depthWidth = m_desc.depthWidth; info.depthHeight = m_desc.depthHeight;
            info.depthShift = 1000.0f;
            m_recTmp = m_recordTmpPrefix + std::to_string((unsigned long long)(uintptr_t)this) + ".rec.sens";
            check(bf_sensor_data_writer_create(m_recTmp.c_str(), &info, &m_recWriter));
        }
        const size_t nd = (size_t)m_desc.depthWidth * m_desc.depthHeight, nc = (size_t)m_desc.colorWidth * m_desc.colorHeight;
        std::vector<uint16_t> depth(nd);
        std::vector<unsigned char> color(nc * 3);
        const float* d = getDepthFloat();
        const unsigned char* c = getColorRGBX();
        for (size_t i = 0; i < nd; ++i) {
            const float v = d[i] * 1000.0f;                                       // (unsigned short) round(m_depthShift * d[i]) :305
            depth[i] = (v > 0.0f && v < 65535.5f) ? (uint16_t)(v + 0.5f) : (uint16_t)0;
        }
        for (size_t i = 0; i < nc; ++i) { color[3 * i] = c[4 * i]; color[3 * i + 1] = c[4 * i + 1]; color[3 * i + 2] = c[4 * i + 2]; }
        uint64_t jpegSize = 0;
        check(bf_encode_jpeg_rgb(color.data(), m_desc.colorWidth, m_desc.colorHeight, 90, nullptr, 0, &jpegSize));
        std::vector<unsigned char> jpeg(jpegSize);
        check(bf_encode_jpeg_rgb(color.data(), m_desc.colorWidth, m_desc.colorHeight, 90, jpeg.data(), jpeg.size(), &jpegSize));
        const mat4f I = mat4f::identity();
        check(bf_sensor_data_writer_add_frame(m_recWriter, I.m, 0, 0, jpeg.data(), jpegSize, depth.data()));
        m_numRecorded++;
    }
    unsigned int getNumRecordedFrames() const { return m_numRecorded; }
    // returns the name actually written: unless overwriteExistingFile, an existing file gets a numeric suffix (:382-396)
    std::string saveRecordedFramesToFile(const std::string& filename, const std::vector<mat4f>& trajectory, bool overwriteExistingFile = false) {
        if (!m_recWriter || trajectory.empty()) return std::string();
        check(bf_sensor_data_writer_close(m_recWriter)); m_recWriter = nullptr;
        std::string actual = filename;
        if (!overwriteExistingFile) {
            const size_t dot = filename.find_last_of('.');
            const std::string base = dot == std::string::npos ? filename : filename.substr(0, dot), ext = dot == std::string::npos ? "" : filename.substr(dot);
            for (unsigned int num = 1; fileExists(actual); ++num) actual = base + std::to_string(num) + ext;
        }
        bf_sensor_data* sd = nullptr;
        check(bf_sensor_data_open(m_recTmp.c_str(), &sd));
        const int rc = bf_sensor_data_save_recorded(sd, actual.c_str(), trajectory[0].m, trajectory.size());
        bf_sensor_data_close(sd);
        std::remove(m_recTmp.c_str()); m_recTmp.clear(); m_numRecorded = 0;
        check(rc);
        return actual;
    }
    void setRecordTempPrefix(const std::string& prefix) { m_recordTmpPrefix = prefix; }   // where the temporary recording lives (default: ./)
protected:
    bf_rgbd_sensor_desc m_desc;
private:
    static bool fileExists(const std::string& f) { if (FILE* fp = std::fopen(f.c_str(), "rb")) { std::fclose(fp); return true; } return false; }
    bf_sensor_data_writer* m_recWriter = nullptr;
    std::string m_recTmp, m_recordTmpPrefix = "./bf_";
    unsigned int m_numRecorded = 0;
};

// ---- the confirmation file of StopScanningAndExit (DepthSensing.cpp:921-957): processed.txt next to the processed sequence; returns `valid`
inline bool writeProcessedSummary(const std::string& path, unsigned int heapFreeCount, const std::vector<mat4f>& optimizedTrajectory, bool aborted = false) {
    int valid = 0;
    check(bf_write_processed_summary(path.c_str(), heapFreeCount, optimizedTrajectory.empty() ? nullptr : optimizedTrajectory[0].m,
                                             (uint32_t)optimizedTrajectory.size(), aborted ? 1 : 0, &valid));
    return valid != 0;
}

// ---- PoseHelper (PoseHelper.h:8-166): trajectory bookkeeping and evaluation on the host
namespace PoseHelper {
inline unsigned int countNumValidTransforms(const std::vector<mat4f>& trajectory) {
    unsigned int count = 0;
    for (const mat4f& T : trajectory) if (T.m[0] != -std::numeric_limits<float>::infinity()) count++;
    return count;
}
// key-frame poses + per-chunk relative poses -> one pose per frame (:19-33)
inline void composeTrajectory(unsigned int submapSize, const std::vector<mat4f>& keys, std::vector<mat4f>& all) {
    std::vector<mat4f> transforms;
    for (unsigned int i = 0; i < keys.size(); i++) {
        const mat4f& key = keys[i];
        transforms.push_back(key);
        const mat4f offset = all[i * submapSize].getInverse();
        const unsigned int num = (unsigned int)std::min((int)submapSize, (int)all.size() - (int)(i * submapSize));
        for (unsigned int s = 1; s < num; s++) transforms.push_back(key * offset * all[i * submapSize + s]);
    }
    all = transforms;
}
inline std::pair<float, unsigned int> evaluateAteRmse(const std::vector<mat4f>& trajectory, const std::vector<mat4f>& referenceTrajectory,
                                                      unsigned int numTransforms = (unsigned int)-1) {
    if (numTransforms == (unsigned int)-1) numTransforms = (unsigned int)std::min(trajectory.size(), referenceTrajectory.size());
    float rmse = 0.0f; uint32_t n = 0;
    check(bf_evaluate_ate_rmse(numTransforms ? trajectory[0].m : nullptr, numTransforms ? referenceTrajectory[0].m : nullptr, numTransforms, &rmse, &n));
    return std::make_pair(rmse, (unsigned int)n);
}
inline mat4f getAlignmentBetweenTrajectories(const std::vector<mat4f>& trajectory, const std::vector<mat4f>& referenceTrajectory,
                                             unsigned int numTransforms = (unsigned int)-1) {
    if (numTransforms == (unsigned int)-1) numTransforms = (unsigned int)std::min(trajectory.size(), referenceTrajectory.size());
    mat4f ret;
    check(bf_trajectory_alignment(numTransforms ? trajectory[0].m : nullptr, numTransforms ? referenceTrajectory[0].m : nullptr, numTransforms, ret.m));
    return ret;
}
inline std::vector<std::pair<unsigned int, float>> evaluateErr2PerImage(const std::vector<mat4f>& trajectory, const std::vector<mat4f>& referenceTrajectory) {
    const uint32_t n = (uint32_t)std::min(trajectory.size(), referenceTrajectory.size());
    std::vector<uint32_t> idx(n ? n : 1); std::vector<float> e(n ? n : 1);
    uint32_t cnt = 0;
    check(bf_evaluate_err2_per_image(n ? trajectory[0].m : nullptr, n ? referenceTrajectory[0].m : nullptr, n, idx.data(), e.data(), &cnt));
    std::vector<std::pair<unsigned int, float>> errors;
    for (uint32_t i = 0; i < cnt; ++i) errors.push_back(std::make_pair((unsigned int)idx[i], e[i]));
    return errors;
}
// "index tx ty tz qx qy qz qw" per valid pose (:148-165)
inline void saveToPoseFile(const std::string& filename, const std::vector<mat4f>& trajectory) {
    std::ofstream s(filename);
    for (unsigned int i = 0; i < trajectory.size(); i++) {
        const mat4f& T = trajectory[i];
        if (T(0, 0) == -std::numeric_limits<float>::infinity()) continue;
        // unit quaternion of the rotation block (largest-component branch, w >= 0 convention of ml::quatf(mat3f) is not pinned: q and -q are the same rotation)
        const float m00 = T(0, 0), m11 = T(1, 1), m22 = T(2, 2), tr = m00 + m11 + m22;
        float qw, qx, qy, qz;
        if (tr > 0.0f) { const float r = std::sqrt(tr + 1.0f) * 2.0f; qw = 0.25f * r; qx = (T(2, 1) - T(1, 2)) / r; qy = (T(0, 2) - T(2, 0)) / r; qz = (T(1, 0) - T(0, 1)) / r; }
        else if (m00 > m11 && m00 > m22) { const float r = std::sqrt(1.0f + m00 - m11 - m22) * 2.0f; qw = (T(2, 1) - T(1, 2)) / r; qx = 0.25f * r; qy = (T(0, 1) + T(1, 0)) / r; qz = (T(0, 2) + T(2, 0)) / r; }
        else if (m11 > m22) { const float r = std::sqrt(1.0f + m11 - m00 - m22) * 2.0f; qw = (T(0, 2) - T(2, 0)) / r; qx = (T(0, 1) + T(1, 0)) / r; qy = 0.25f * r; qz = (T(1, 2) + T(2, 1)) / r; }
        else { const float r = std::sqrt(1.0f + m22 - m00 - m11) * 2.0f; qw = (T(1, 0) - T(0, 1)) / r; qx = (T(0, 2) + T(2, 0)) / r; qy = (T(1, 2) + T(2, 1)) / r; qz = 0.25f * r; }
        s << i << " " << T(0, 3) << " " << T(1, 3) << " " << T(2, 3) << " " << qx << " " << qy << " " << qz << " " << qw << std::endl;
    }
}
}  // namespace PoseHelper

// ---- TimingLog (TimingLog.h:6-283): per-frame timings in the reference's text / "excel" file formats, so that numbers are
// comparable with a CUDA run of the reference.  Fed from bf_frame_timing (bf_pipeline_get_last_timing) or filled directly.
class TimingLog {
public:
    struct FrameTiming {                                               // TimingLog.h:9-57
        double timeSiftDetection = 0, timeSiftMatching = 0, timeMatchFilterKeyPoint = 0, timeMatchFilterSurfaceArea = 0, timeMatchFilterDenseVerify = 0;
        double timeMisc = 0, timeSolve = 0; unsigned int numItersSolve = 0;
        double timeSensorProcess = 0, timeReIntegrate = 0, timeReconstruct = 0, timeVisualize = 0;
        FrameTiming() {}
        explicit FrameTiming(const bf_frame_timing& t) {               // the loop's stage timings; the three filters are one stage here
            timeSiftDetection = t.timeSiftDetection; timeSiftMatching = t.timeSiftMatching; timeMatchFilterKeyPoint = t.timeMatchFilter;
            timeSolve = t.timeSolve; timeSensorProcess = t.timeSensorProcess; timeReIntegrate = t.timeReIntegrate; timeReconstruct = t.timeReconstruct;
        }
        void print(std::ostream* out, bool printDepthSensing) const {
            *out << "\tTime SIFT Detection: " << std::to_string(timeSiftDetection) << "ms" << std::endl;
            *out << "\tTime SIFT Matching: " << std::to_string(timeSiftMatching) << "ms" << std::endl;
            *out << "\tTime Match Filter Key Point: " << std::to_string(timeMatchFilterKeyPoint) << "ms" << std::endl;
            *out << "\tTime Match Filter Surface Area: " << std::to_string(timeMatchFilterSurfaceArea) << "ms" << std::endl;
            *out << "\tTime Match Filter Dense Verify: " << std::to_string(timeMatchFilterDenseVerify) << "ms" << std::endl;
            *out << "\tTime Misc: " << std::to_string(timeMisc) << "ms" << std::endl;
            *out << "\tTime Solve: " << std::to_string(timeSolve) << "ms" << std::endl;
            *out << "\t#iters solve: " << std::to_string(numItersSolve) << std::endl;
            if (printDepthSensing) {
                *out << "\tTime Process Input: " << std::to_string(timeSensorProcess) << "ms" << std::endl;
                *out << "\tTime Re-Integrate: " << std::to_string(timeReIntegrate) << "ms" << std::endl;
                *out << "\tTime Reconstruct: " << std::to_string(timeReconstruct) << std::endl;
                *out << "\tTime Visualize: " << std::to_string(timeVisualize) << std::endl;
            }
        }
    };
    static void init() { resetTimings(); }
    static void destroy() {}
    static void resetTimings() { local().clear(); global().clear(); total().clear(); }
    static void addLocalFrameTiming() { local().push_back(FrameTiming()); }
    static void addGlobalFrameTiming() { global().push_back(FrameTiming()); }
    static void addLocalFrameTiming(const bf_frame_timing& t) { local().push_back(FrameTiming(t)); total().push_back(t.timeTotal); }
    static FrameTiming& getFrameTiming(bool isLocal) { return isLocal ? local().back() : global().back(); }
    static void printAllTimings(const std::string& dir = "./timings/") {   // the directory must exist (the reference creates it with mLib's util)
        printTimings(dir + (total().empty() ? "timingLog.txt" : "timingLogPerFrame.txt"));
        printExcelTimings(dir + "excel");
    }
    static void printTimings(const std::string& filename) {             // TimingLog.h:83-121
        std::ofstream outFile;
        if (!filename.empty()) outFile.open(filename, std::ios::out);
        std::ostream& out = filename.empty() ? std::cout : outFile;
        if (!global().empty()) {
            out << "Global Timings Per Frame:" << std::endl;
            for (unsigned int i = 0; i < global().size(); i++) { out << "[ frame " << i << " ]" << std::endl; global()[i].print(&out, false); }
            out << std::endl << std::endl;
        }
        if (!local().empty()) {
            out << "Local Timings Per Frame:" << std::endl;
            for (unsigned int i = 0; i < local().size(); i++) { out << "[ frame " << i << " ]" << std::endl; local()[i].print(&out, true); }
            out << std::endl << std::endl;
        }
        if (!total().empty()) {
            out << "Total Timings Per Frame:" << std::endl;
            for (unsigned int i = 0; i < total().size(); i++) out << "[ frame " << i << " ] " << total()[i] << " ms" << std::endl;
            out << std::endl << std::endl;
        }
    }
    static void printExcelTimings(const std::string& prefix) {          // TimingLog.h:204-233
        const std::string separator = ",";
        if (!global().empty()) { std::ofstream out(prefix + "_global.txt"); printAverages(&out, separator, global(), false); printExcelTimings(&out, separator, global(), false); }
        if (!local().empty()) { std::ofstream out(prefix + "_local.txt"); printAverages(&out, separator, local(), true); printExcelTimings(&out, separator, local(), true); }
        if (!total().empty()) {
            std::ofstream out(prefix + "_total.txt");
            out << "Per Frame Timings";
            for (unsigned int i = 0; i < total().size(); i++) out << separator << total()[i];
        }
    }
private:
    static std::vector<FrameTiming>& local() { static std::vector<FrameTiming> v; return v; }
    static std::vector<FrameTiming>& global() { static std::vector<FrameTiming> v; return v; }
    static std::vector<double>& total() { static std::vector<double> v; return v; }
    template <class F> static void row(std::ofstream* out, const char* name, const std::string& sep, const std::vector<FrameTiming>& ft, F field) {
        *out << name;
        for (unsigned int i = 0; i < ft.size(); i++) *out << sep << field(ft[i]);
        *out << std::endl;
    }
    static void printExcelTimings(std::ofstream* out, const std::string& sep, const std::vector<FrameTiming>& ft, bool printDepthSensing) {   // :123-162
        row(out, "SIFT Detection", sep, ft, [](const FrameTiming& f) { return f.timeSiftDetection; });
        row(out, "SIFT Matching", sep, ft, [](const FrameTiming& f) { return f.timeSiftMatching; });
        row(out, "Match Filter Key Point", sep, ft, [](const FrameTiming& f) { return f.timeMatchFilterKeyPoint; });
        row(out, "Match Filter Surface Area", sep, ft, [](const FrameTiming& f) { return f.timeMatchFilterSurfaceArea; });
        row(out, "Match Filter Dense Verify", sep, ft, [](const FrameTiming& f) { return f.timeMatchFilterDenseVerify; });
        row(out, "Misc", sep, ft, [](const FrameTiming& f) { return f.timeMisc; });
        row(out, "Solve", sep, ft, [](const FrameTiming& f) { return f.timeSolve; });
        row(out, "Solve #Iters", sep, ft, [](const FrameTiming& f) { return f.numItersSolve; });
        if (printDepthSensing) {
            row(out, "Process Input", sep, ft, [](const FrameTiming& f) { return f.timeSensorProcess; });
            row(out, "Re-Integrate", sep, ft, [](const FrameTiming& f) { return f.timeReIntegrate; });
            row(out, "Reconstruct", sep, ft, [](const FrameTiming& f) { return f.timeReconstruct; });
            row(out, "Visualize", sep, ft, [](const FrameTiming& f) { return f.timeVisualize; });
        }
    }
    template <class F> static void average(std::ofstream* out, const char* name, const std::string& sep, const std::vector<FrameTiming>& ft, F field) {
        double sum = 0.0; unsigned int count = 0;
        for (unsigned int i = 0; i < ft.size(); i++) { sum += field(ft[i]); count++; }
        *out << name << sep << (sum / count) << sep << count << std::endl;
    }
    static void printAverages(std::ofstream* out, const std::string& sep, const std::vector<FrameTiming>& ft, bool printDepthSensing) {        // :164-202
        *out << "Average times:" << std::endl;
        average(out, "SIFT Detection", sep, ft, [](const FrameTiming& f) { return f.timeSiftDetection; });
        average(out, "SIFT Matching", sep, ft, [](const FrameTiming& f) { return f.timeSiftMatching; });
        average(out, "Corr Filter", sep, ft, [](const FrameTiming& f) { return f.timeMatchFilterKeyPoint + f.timeMatchFilterSurfaceArea + f.timeMatchFilterDenseVerify; });
        average(out, "Misc", sep, ft, [](const FrameTiming& f) { return f.timeMisc; });
        average(out, "Solve", sep, ft, [](const FrameTiming& f) { return f.timeSolve; });
        if (printDepthSensing) {
            average(out, "Re-Integrate", sep, ft, [](const FrameTiming& f) { return f.timeReIntegrate; });
            average(out, "Misc", sep, ft, [](const FrameTiming& f) { return f.timeSensorProcess + f.timeReconstruct + f.timeVisualize; });
        }
        *out << std::endl << std::endl;
    }
};

// ---- SensorDataReader (SensorDataReader.h:19-68): plays a recorded ".sens" file through the RGBDSensor contract
class SensorDataReader : public RGBDSensor {
public:
    SensorDataReader() { std::memset(&m_desc, 0, sizeof m_desc); }
    ~SensorDataReader() { releaseData(); }
    SensorDataReader(const SensorDataReader&) = delete;
    // optional PNG / JPEG decoder (the reference decodes inside mLib with stb_image); raw colour needs none
    void setColorDecoder(bf_sens_color_decoder fn, void* user) { m_decoder = fn; m_decoderUser = user; if (m_sd) check(bf_sensor_data_set_color_decoder(m_sd, fn, user)); }
    void createFirstConnected() { createFirstConnected(GlobalAppState::get().s_binaryDumpSensorFile); }         // .cpp:40-83
    void createFirstConnected(const std::string& filename) {
        releaseData();
        check(bf_sensor_data_open(filename.c_str(), &m_sd));
        if (m_decoder) check(bf_sensor_data_set_color_decoder(m_sd, m_decoder, m_decoderUser));
        check(bf_sensor_data_get_info(m_sd, &m_info));
        check(bf_sensor_data_get_sensor_desc(m_sd, &m_desc));
        m_numFrames = (unsigned int)m_info.numFrames;
        const GlobalBundlingState& gbs = GlobalBundlingState::get();
        if (m_numFrames > gbs.s_maxNumImages * gbs.s_submapSize)                                               // :65-67
            throw std::runtime_error("sens file #frames = " + std::to_string(m_numFrames) + ", please change param file to accommodate");
        uint64_t colorBytes = 0, depthBytes = 0;
        if (m_numFrames > 0) check(bf_sensor_data_get_frame_sizes(m_sd, 0, &colorBytes, &depthBytes));
        m_bHasColorData = m_numFrames > 0 && colorBytes > 0;                                                   // :73-78
        m_depth.assign((size_t)m_desc.depthWidth * m_desc.depthHeight, 0.0f);
        m_colorRGBX.assign((size_t)m_desc.colorWidth * m_desc.colorHeight * 4, 0);
        m_currFrame = 0; m_playData = true; m_bIsReceivingFrames = true;
    }
    bool processDepth() override {                                                                              // .cpp:85-122
        if (!m_sd) return false;
        if (m_currFrame >= m_numFrames) { m_playData = false; stopReceivingFrames(); m_currFrame = 0; }        // "binary dump sequence complete"
        if (!m_playData) return false;
        check(bf_sensor_data_read_depth(m_sd, m_currFrame, m_depth.data()));                                   // 0 -> -inf, u16 / depthShift
        if (m_bHasColorData) check(bf_sensor_data_read_color_rgbx(m_sd, m_currFrame, m_colorRGBX.data()));
        m_currFrame++;
        return true;
    }
    bool processColor() override { return true; }                 // everything is done in processDepth (.h:35-38)
    const float* getDepthFloat() const override { return m_depth.data(); }
    const unsigned char* getColorRGBX() const override { return m_colorRGBX.data(); }
    std::string getSensorName() const override { return m_info.sensorName; }
    unsigned int getNumFrames() const { return m_numFrames; }
    bool isReceivingFrames() const { return m_bIsReceivingFrames; }
    void stopReceivingFrames() { m_bIsReceivingFrames = false; }
    mat4f getRigidTransform(int offset) const {                                                                 // .cpp:128-135
        const unsigned int idx = m_currFrame - 1 + offset;
        if (!m_sd || idx >= m_numFrames) throw std::runtime_error("invalid trajectory index " + std::to_string(idx));
        mat4f T; check(bf_sensor_data_get_frame_pose(m_sd, idx, T.m, nullptr, nullptr));
        return T;
    }
    void getTrajectory(std::vector<mat4f>& trajectory) const {                                                  // .cpp:191-201
        trajectory.clear();
        if (!m_sd) return;
        trajectory.resize(m_numFrames);
        for (unsigned int f = 0; f < m_numFrames; f++) {
            check(bf_sensor_data_get_frame_pose(m_sd, f, trajectory[f].m, nullptr, nullptr));
            if (trajectory[f].m[0] == -std::numeric_limits<float>::infinity()) throw std::runtime_error("ERROR invalid transform in reference trajectory");
        }
    }
    void saveToFile(const std::string& filename, const std::vector<mat4f>& trajectory) const {                 // .cpp:152-165
        check(bf_sensor_data_save_with_trajectory(m_sd, filename.c_str(), trajectory.empty() ? nullptr : trajectory[0].m, trajectory.size()));
    }
    std::pair<float, unsigned int> evaluateTrajectory(const std::vector<mat4f>& trajectory) const {            // .cpp:167-189 (prints "ate rmse = ..")
        float rmse = 0.0f; uint32_t n = 0;
        check(bf_sensor_data_evaluate_trajectory(m_sd, trajectory.empty() ? nullptr : trajectory[0].m, trajectory.size(), &rmse, &n));
        std::printf("*********************************\nate rmse = %g, %u\n*********************************\n", rmse, n);
        return std::make_pair(rmse, (unsigned int)n);
    }
private:
    void releaseData() { if (m_sd) bf_sensor_data_close(m_sd); m_sd = nullptr; m_currFrame = 0; m_bHasColorData = false; }
    bf_sensor_data* m_sd = nullptr;
    bf_sensor_data_info m_info;
    bf_sens_color_decoder m_decoder = nullptr; void* m_decoderUser = nullptr;
    std::vector<float> m_depth; std::vector<unsigned char> m_colorRGBX;
    unsigned int m_numFrames = 0, m_currFrame = 0;
    bool m_bHasColorData = false, m_playData = true, m_bIsReceivingFrames = false;
};

// ---- CUDAImageManager (CUDAImageManager.h:10-337)
class CUDAImageManager {
public:
    class ManagedRGBDInputFrame {
    public:
        const float* getDepthFrameGPU() { const float* d; const uint8_t* c; check(bf_image_manager_get_integrate_frame_gpu(m_im, m_idx, &d, &c)); return d; }
        const unsigned char* getColorFrameGPU() { const float* d; const uint8_t* c; check(bf_image_manager_get_integrate_frame_gpu(m_im, m_idx, &d, &c)); return c; }
        const float* getDepthFrameCPU() {                // copied out of HBM on request; valid while this object lives
            uint32_t w, h; check(bf_image_manager_get_integration_size(m_im, &w, &h));
            m_depthCPU.resize((size_t)w * h);
            check(bf_image_manager_get_integrate_frame_cpu(m_im, m_idx, m_depthCPU.data(), nullptr));
            return m_depthCPU.data();
        }
        const unsigned char* getColorFrameCPU() {
            uint32_t w, h; check(bf_image_manager_get_integration_size(m_im, &w, &h));
            m_colorCPU.resize((size_t)w * h * 4);
            check(bf_image_manager_get_integrate_frame_cpu(m_im, m_idx, nullptr, m_colorCPU.data()));
            return m_colorCPU.data();
        }
    private:
        friend class CUDAImageManager;
        bf_image_manager* m_im = nullptr; unsigned int m_idx = 0;
        std::vector<float> m_depthCPU; std::vector<unsigned char> m_colorCPU;
    };
    CUDAImageManager(unsigned int widthIntegration, unsigned int heightIntegration, unsigned int widthSIFT, unsigned int heightSIFT, RGBDSensor* sensor,
                     bool storeFramesOnGPU = false) : m_sensor(sensor) {      // default as CUDAImageManager.h:140; pass true to keep every frame resident in HBM (what bf_pipeline does)
        check(bf_image_manager_create(widthIntegration, heightIntegration, widthSIFT, heightSIFT, &sensor->desc(), &GlobalBundlingState::get(), storeFramesOnGPU, &m_h));
    }
    ~CUDAImageManager() { bf_image_manager_destroy(m_h); }
    CUDAImageManager(const CUDAImageManager&) = delete;
    void reset() { check(bf_image_manager_reset(m_h)); }
    bool process() {
        if (!m_sensor->processDepth()) return false;       // order is important (CUDAImageManager.cpp:24-25)
        if (!m_sensor->processColor()) return false;
        int got = 0;
        check(bf_image_manager_process(m_h, m_sensor->getDepthFloat(), m_sensor->getColorRGBX(), &got));
        return got != 0;
    }
    void copyToBundling(float* d_depthRaw, float* d_depthFilt, unsigned char* d_color) const { check(bf_image_manager_copy_to_bundling(m_h, d_depthRaw, d_depthFilt, d_color)); }
    ManagedRGBDInputFrame getIntegrateFrame(unsigned int frame) { ManagedRGBDInputFrame f; f.m_im = m_h; f.m_idx = frame; return f; }
    ManagedRGBDInputFrame getLastIntegrateFrame() { return getIntegrateFrame(getCurrFrameNumber()); }
    unsigned int getCurrFrameNumber() const { uint32_t n; check(bf_image_manager_get_curr_frame_number(m_h, &n)); return n; }
    unsigned int getIntegrationWidth() const { uint32_t w, h; check(bf_image_manager_get_integration_size(m_h, &w, &h)); return w; }
    unsigned int getIntegrationHeight() const { uint32_t w, h; check(bf_image_manager_get_integration_size(m_h, &w, &h)); return h; }
    mat4f getDepthIntrinsics() const { mat4f k; check(bf_image_manager_get_depth_intrinsics(m_h, k.m, nullptr)); return k; }
    mat4f getDepthIntrinsicsInv() const { mat4f k; check(bf_image_manager_get_depth_intrinsics(m_h, nullptr, k.m)); return k; }
    mat4f getDepthExtrinsics() const { mat4f k; check(bf_image_manager_get_depth_extrinsics(m_h, k.m, nullptr)); return k; }
    mat4f getDepthExtrinsicsInv() const { mat4f k; check(bf_image_manager_get_depth_extrinsics(m_h, nullptr, k.m)); return k; }
    unsigned int getSIFTDepthWidth() const { uint32_t w; check(bf_image_manager_get_sift_depth(m_h, &w, nullptr, nullptr)); return w; }
    unsigned int getSIFTDepthHeight() const { uint32_t h; check(bf_image_manager_get_sift_depth(m_h, nullptr, &h, nullptr)); return h; }
    mat4f getSIFTDepthIntrinsics() const { mat4f k; check(bf_image_manager_get_sift_depth(m_h, nullptr, nullptr, k.m)); return k; }
    bool hasBundlingFrameRdy() const { return m_bHasBundlingFrameRdy; }
    void setBundlingFrameRdy() { m_bHasBundlingFrameRdy = true; }
    void confirmRdyBundlingFrame() { m_bHasBundlingFrameRdy = false; }
    bf_image_manager* handle() const { return m_h; }
    const RGBDSensor* sensor() const { return m_sensor; }
private:
    bf_image_manager* m_h = nullptr;
    RGBDSensor* m_sensor;
    bool m_bHasBundlingFrameRdy = false;
};

// ---- Bundler (Bundler.h:17-104)
class Bundler {
public:
    Bundler(unsigned int maxNumImages, unsigned int maxNumKeysPerImage, const mat4f& siftIntrinsicsInv, const CUDAImageManager* manager, bool isLocal) {
        check(bf_bundler_create(maxNumImages, maxNumKeysPerImage, siftIntrinsicsInv.m, manager->handle(), isLocal, &GlobalAppState::get(), &GlobalBundlingState::get(), &m_h));
        m_own = true;
    }
    explicit Bundler(bf_bundler* borrowed) : m_h(borrowed), m_own(false) {}
    ~Bundler() { if (m_own) bf_bundler_destroy(m_h); }
    Bundler(const Bundler&) = delete;
    float4x4* getTrajectoryGPU() { float* d; check(bf_bundler_get_trajectory_gpu(m_h, &d)); return reinterpret_cast<float4x4*>(d); }
    const std::vector<int>& getValidImages() const {
        uint32_t n; check(bf_bundler_get_num_frames(m_h, &n));
        m_valid.assign(n ? n : 1, 0);
        check(bf_bundler_get_valid_images(m_h, m_valid.data(), n));
        return m_valid;
    }
    void getCacheIntrinsics(float4x4& intrinsics, float4x4& intrinsicsInv) { check(bf_bundler_get_cache_intrinsics(m_h, intrinsics.m, intrinsicsInv.m)); }
    unsigned int getCurrFrameNumber() const { uint32_t n; check(bf_bundler_get_curr_frame_number(m_h, &n)); return n; }
    unsigned int getNumFrames() const { uint32_t n; check(bf_bundler_get_num_frames(m_h, &n)); return n; }
    bool isValid() const { int v; check(bf_bundler_is_valid(m_h, &v)); return v != 0; }
    void reset() { check(bf_bundler_reset(m_h)); }
    void detectFeatures(float* d_intensitySift, const float* d_inputDepthFilt) { check(bf_bundler_detect_features(m_h, d_intensitySift, d_inputDepthFilt)); }
    void storeCachedFrame(unsigned int depthWidth, unsigned int depthHeight, const unsigned char* d_inputColor, unsigned int colorWidth, unsigned int colorHeight,
                          const float* d_inputDepthRaw) { check(bf_bundler_store_cached_frame(m_h, depthWidth, depthHeight, d_inputColor, colorWidth, colorHeight, d_inputDepthRaw)); }
    void copyFrame(const Bundler* b, unsigned int frame) { check(bf_bundler_copy_frame(m_h, b->m_h, frame)); }
    void addInvalidFrame() { check(bf_bundler_add_invalid_frame(m_h)); }
    void invalidateLastFrame() { check(bf_bundler_invalidate_last_frame(m_h)); }
    const float4x4* getCurrentSiftTransformsGPU() const { const float* d; check(bf_bundler_get_current_sift_transforms_gpu(m_h, &d)); return reinterpret_cast<const float4x4*>(d); }
    const int* getNumFiltMatchesGPU() const { const int32_t* d; check(bf_bundler_get_num_filt_matches_gpu(m_h, &d)); return d; }
    unsigned int matchAndFilter() { uint32_t last; check(bf_bundler_match_and_filter(m_h, &last)); return last; }
    bool optimize(unsigned int numNonLinIterations, unsigned int numLinIterations, bool bUseVerify, bool bRemoveMaxResidual, bool bIsScanDone, bool& bOptRemoved) {
        int removed = 0, valid = 0;
        check(bf_bundler_optimize(m_h, numNonLinIterations, numLinIterations, bUseVerify, bRemoveMaxResidual, bIsScanDone, &removed, &valid));
        bOptRemoved = removed != 0;
        return valid != 0;
    }
    void setSolveWeights(const std::vector<float>& sparse, const std::vector<float>& densedepth, const std::vector<float>& densecolor) {
        check(bf_bundler_set_solve_weights(m_h, sparse.data(), densedepth.data(), densecolor.data(), (uint32_t)sparse.size()));
    }
    void fuseToGlobal(Bundler* glob) { check(bf_bundler_fuse_to_global(m_h, glob->m_h)); }
    unsigned int tryRevalidation(unsigned int curGlobalFrame, bool bIsScanDone) { uint32_t r; check(bf_bundler_try_revalidation(m_h, curGlobalFrame, bIsScanDone, &r)); return r; }
    unsigned int getRevalidatedIdx() const { uint32_t r; check(bf_bundler_get_revalidated_idx(m_h, &r)); return r; }
    void saveSparseCorrsToFile(const std::string& filename) const { check(bf_bundler_save_sparse_corrs_to_file(m_h, filename.c_str())); }
    bf_bundler* handle() const { return m_h; }
private:
    bf_bundler* m_h = nullptr;
    bool m_own = false;
    mutable std::vector<int> m_valid;
};

// ---- TrajectoryManager (TrajectoryManager.h:6-116)
class TrajectoryManager {
public:
    struct TrajectoryFrame {
        enum TYPE { Integrated = 0, NotIntegrated_NoTransform = 1, NotIntegrated_WithTransform = 2, Invalid = 3, ReIntegration = 4 };
    };
    explicit TrajectoryManager(bf_trajectory_manager* borrowed) : m_h(borrowed) {}
    void addFrame(TrajectoryFrame::TYPE what, const mat4f& transform, unsigned int idx) { check(bf_trajectory_manager_add_frame(m_h, what, transform.m, idx)); }
    void updateOptimizedTransform(const float4x4* d_trajectory, unsigned int numFrames) {
        check(bf_trajectory_manager_update_optimized_transform(m_h, reinterpret_cast<const float*>(d_trajectory), numFrames, nullptr));
    }
    void generateUpdateLists() { check(bf_trajectory_manager_generate_update_lists(m_h)); }
    void confirmIntegration(unsigned int frameIdx) { check(bf_trajectory_manager_confirm_integration(m_h, frameIdx)); }
    bool getTopFromReIntegrateList(mat4f& oldTransform, mat4f& newTransform, unsigned int& frameIdx) {
        int f; check(bf_trajectory_manager_get_top_from_reintegrate_list(m_h, oldTransform.m, newTransform.m, &frameIdx, &f)); return f != 0;
    }
    bool getTopFromIntegrateList(mat4f& trans, unsigned int& frameIdx) { int f; check(bf_trajectory_manager_get_top_from_integrate_list(m_h, trans.m, &frameIdx, &f)); return f != 0; }
    bool getTopFromDeIntegrateList(mat4f& trans, unsigned int& frameIdx) { int f; check(bf_trajectory_manager_get_top_from_deintegrate_list(m_h, trans.m, &frameIdx, &f)); return f != 0; }
    // the reference serialises updateOptimizedTransform against generateUpdateLists across its two threads (.h:70-77); callers of this
    // layer that use two threads get the same two calls
    void lockUpdateTransforms() { m_mutexUpdateTransforms.lock(); }
    void unlockUpdateTransforms() { m_mutexUpdateTransforms.unlock(); }
    unsigned int getNumOptimizedFrames() const { uint32_t n; check(bf_trajectory_manager_get_num_optimized_frames(m_h, &n)); return n; }
    unsigned int getNumAddedFrames() const { uint32_t n; check(bf_trajectory_manager_get_num_added_frames(m_h, &n)); return n; }
    unsigned int getNumActiveOperations() const { uint32_t n; check(bf_trajectory_manager_get_num_active_operations(m_h, &n)); return n; }
    void getOptimizedTransforms(std::vector<mat4f>& transforms) {
        const unsigned int n = getNumAddedFrames();
        transforms.resize(n ? n : 1);
        uint32_t cnt = 0;
        check(bf_trajectory_manager_get_optimized_transforms(m_h, transforms[0].m, n, &cnt));
        transforms.resize(cnt);
    }
private:
    bf_trajectory_manager* m_h;
    std::mutex m_mutexUpdateTransforms;
};

// ---- OnlineBundler (OnlineBundler.h:10-106)
class OnlineBundler {
public:
    OnlineBundler(const RGBDSensor* sensor, const CUDAImageManager* imageManager) {
        check(bf_online_bundler_create(&sensor->desc(), imageManager->handle(), &GlobalAppState::get(), &GlobalBundlingState::get(), &m_h));
        bf_trajectory_manager* tm; check(bf_online_bundler_get_trajectory_manager(m_h, &tm));
        m_tm = new TrajectoryManager(tm);
    }
    ~OnlineBundler() { delete m_tm; bf_online_bundler_destroy(m_h); }
    OnlineBundler(const OnlineBundler&) = delete;
    bool getCurrentIntegrationFrame(mat4f& siftTransform, unsigned int& frameIdx, bool& bGlobalTrackingLost) {
        int lost = 0, valid = 0;
        check(bf_online_bundler_get_current_integration_frame(m_h, siftTransform.m, &frameIdx, &lost, &valid));
        bGlobalTrackingLost = lost != 0;
        return valid != 0;
    }
    void processInput() { check(bf_online_bundler_process_input(m_h)); }
    void process(unsigned int numNonLinItersLocal, unsigned int numLinItersLocal, unsigned int numNonLinItersGlobal, unsigned int numLinItersGlobal) {
        check(bf_online_bundler_process(m_h, numNonLinItersLocal, numLinItersLocal, numNonLinItersGlobal, numLinItersGlobal));
    }
    TrajectoryManager* getTrajectoryManager() { return m_tm; }
    bool hasProcssedInputFrame() const { return m_bHasProcessedInputFrame; }
    void setProcessedInputFrame() { m_bHasProcessedInputFrame = true; }
    void confirmProcessedInputFrame() { m_bHasProcessedInputFrame = false; }
    void exitBundlingThread() { m_bExitBundlingThread = true; }
    bool getExitBundlingThread() const { return m_bExitBundlingThread; }
    unsigned int getCurrProcessedFrame() const { int32_t f; check(bf_online_bundler_get_curr_processed_frame(m_h, &f)); return (unsigned int)f; }
    void saveGlobalSparseCorrsToFile(const std::string& filename) const { check(bf_online_bundler_save_global_sparse_corrs_to_file(m_h, filename.c_str())); }
    // EVALUATE_SPARSE_CORRESPONDENCES (OnlineBundler.cpp:81-90, :480-487): the reference builds the evaluator in the constructor from the
    // .sens trajectory; here the caller hands the per-frame reference trajectory over once (e.g. SensorDataReader::getTrajectory)
    void initializeCorrespondenceEvaluator(const std::vector<mat4f>& completeTrajectory, const std::string& logFilePrefix = "debug/_corr-evaluation") {
        check(bf_online_bundler_initialize_correspondence_evaluator(m_h, completeTrajectory.empty() ? nullptr : completeTrajectory[0].m, (uint32_t)completeTrajectory.size(), logFilePrefix.c_str()));
    }
    void finishCorrespondenceEvaluatorLogging() { check(bf_online_bundler_finish_correspondence_evaluator_logging(m_h)); }
    bf_online_bundler* handle() const { return m_h; }
private:
    bf_online_bundler* m_h = nullptr;
    TrajectoryManager* m_tm = nullptr;
    bool m_bHasProcessedInputFrame = false, m_bExitBundlingThread = false;
};

// ---- voxel-hash volume (DepthSensing/CUDASceneRepHashSDF.h, CUDAHashParams.h, CUDADepthCameraParams.h, DepthCameraUtil.h)
typedef bf_hash_params HashParams;
typedef bf_depth_camera_params DepthCameraParams;
typedef bf_hash_data HashDataStruct;
struct DepthCameraData {
    DepthCameraData() { d.d_depthData = nullptr; d.d_colorData = nullptr; }
    DepthCameraData(const float* d_depth, const unsigned char* d_color) { d.d_depthData = d_depth; d.d_colorData = d_color; }
    // DepthCameraUtil.h:50-53 uploads the camera parameters to constant memory; here they travel with every integrate() call
    static void updateParams(const DepthCameraParams&) {}
    bf_depth_camera_data d;
};

class CUDASceneRepHashSDF {
public:
    explicit CUDASceneRepHashSDF(const HashParams& params) { check(bf_scene_create(&params, &m_h)); }
    ~CUDASceneRepHashSDF() { try { flush(); } catch (...) {} bf_scene_destroy(m_h); }
    CUDASceneRepHashSDF(const CUDASceneRepHashSDF&) = delete;
    static HashParams parametersFromGlobalAppState(const GlobalAppState& gas) {          // CUDASceneRepHashSDF.h:39-59
        HashParams p;
        std::memset(&p, 0, sizeof p);
        const mat4f I = mat4f::identity();
        std::memcpy(p.m_rigidTransform, I.m, 64); std::memcpy(p.m_rigidTransformInverse, I.m, 64);
        p.m_hashNumBuckets = gas.s_hashNumBuckets; p.m_hashBucketSize = 4; p.m_hashMaxCollisionLinkedListSize = gas.s_hashMaxCollisionLinkedListSize;
        p.m_SDFBlockSize = 8; p.m_numSDFBlocks = gas.s_hashNumSDFBlocks; p.m_virtualVoxelSize = gas.s_SDFVoxelSize;
        p.m_maxIntegrationDistance = gas.s_SDFMaxIntegrationDistance; p.m_truncation = gas.s_SDFTruncation; p.m_truncScale = gas.s_SDFTruncationScale;
        p.m_integrationWeightSample = gas.s_SDFIntegrationWeightSample; p.m_integrationWeightMax = gas.s_SDFIntegrationWeightMax;
        for (int i = 0; i < 3; ++i) { p.m_streamingVoxelExtents[i] = gas.s_streamingVoxelExtents[i]; p.m_streamingGridDimensions[i] = gas.s_streamingGridDimensions[i]; p.m_streamingMinGridPos[i] = gas.s_streamingMinGridPos[i]; }
        p.m_streamingInitialChunkListSize = gas.s_streamingInitialChunkListSize;
        return p;
    }
    void integrate(const mat4f& lastRigidTransform, const DepthCameraData& data, const DepthCameraParams& params, unsigned int* d_bitMask) {
        if (!m_deferred || d_bitMask) { flush(); check(bf_scene_integrate(m_h, lastRigidTransform.m, &data.d, &params, d_bitMask)); return; }
        // deIntegrate(old) directly followed by integrate(new) of the same frame (DepthSensing.cpp:885-886) is one fused re-integration
        if (!m_pending.empty() && m_pending.back().kind == 1 && m_pending.back().data.d_depthData == data.d.d_depthData && m_pending.back().data.d_colorData == data.d.d_colorData && m_lastWasDe) {
            m_pending.back().kind = 2; std::memcpy(m_pending.back().T1, lastRigidTransform.m, 64); m_lastWasDe = false;
        } else push(0, lastRigidTransform, data);
        m_cam = params;
    }
    void deIntegrate(const mat4f& lastRigidTransform, const DepthCameraData& data, const DepthCameraParams& params, unsigned int* d_bitMask) {
        if (!m_deferred || d_bitMask) { flush(); check(bf_scene_deintegrate(m_h, lastRigidTransform.m, &data.d, &params, d_bitMask)); return; }
        push(1, lastRigidTransform, data); m_lastWasDe = true;
        m_cam = params;
    }
    void garbageCollect() { flush(); check(bf_scene_garbage_collect(m_h)); }
    void reset() { m_pending.clear(); check(bf_scene_reset(m_h)); }
    void setLastRigidTransformAndCompactify(const mat4f& lastRigidTransform, const DepthCameraParams& params) {
        flush(); check(bf_scene_set_last_rigid_transform_and_compactify(m_h, lastRigidTransform.m, &params));
    }
    void setLastRigidTransform(const mat4f& lastRigidTransform) { flush(); check(bf_scene_set_last_rigid_transform(m_h, lastRigidTransform.m)); }
    const mat4f getLastRigidTransform() { const HashParams p = getHashParams(); mat4f m; std::memcpy(m.m, p.m_rigidTransform, 64); return m; }
    void debugHash() {                                             // :179-314: the invariants the reference prints, from the device-side check
        flush();
        uint32_t v[6]; check(bf_scene_debug_hash(m_h, v));
        std::printf("number of occupied entries: %u\nfree heap: %u\nduplicate keys: %u\nallocated and free: %u\nleaked blocks: %u\ndropped: %u\n", v[0], v[1], v[2], v[3], v[4], v[5]);
    }
    HashDataStruct getHashData() { flush(); HashDataStruct d; check(bf_scene_get_hash_data(m_h, &d)); return d; }
    HashParams getHashParams() { flush(); HashParams p; check(bf_scene_get_hash_params(m_h, &p)); return p; }
    unsigned int getHeapFreeCount() { flush(); uint32_t n; check(bf_scene_get_heap_free_count(m_h, &n)); return n; }
    unsigned int getNumIntegratedFrames() { flush(); uint32_t n; check(bf_scene_get_num_integrated_frames(m_h, &n)); return n; }
    bf_scene* handle() const { return m_h; }

    // MI355X addition (off by default: the calls above then execute one by one, like the reference's): integrate / deIntegrate calls are COLLECTED and issued as one
    // bf_scene_run_batch - one ray march, one placement, one pass over the touched blocks for up to BF_SCENE_BATCH_MAX operators - when garbageCollect() or any
    // accessor is called: a frame's reintegrate() pass (DepthSensing.cpp:854-902) becomes one batch, as in bf_pipeline.  Same volume bit for bit (the batch is the
    // operators in call order).  Requires the DepthCameraData pointers to stay valid until that flush: true for frames of a CUDAImageManager constructed with
    // storeFramesOnGPU = true (every frame resident), not for the reference's single staging buffer.
    void setDeferredBatching(bool enable) { flush(); m_deferred = enable; }
    void flush() {
        if (m_pending.empty()) return;
        std::vector<bf_scene_batch_op> ops; ops.swap(m_pending);
        m_lastWasDe = false;
        check(bf_scene_run_batch(m_h, ops.data(), (uint32_t)ops.size(), &m_cam));
    }
private:
    void push(int kind, const mat4f& T, const DepthCameraData& data) {
        if (m_pending.size() == BF_SCENE_BATCH_MAX) flush();
        bf_scene_batch_op o; std::memset(&o, 0, sizeof o);
        o.kind = kind; std::memcpy(o.T0, T.m, 64); std::memcpy(o.T1, T.m, 64); o.data = data.d;
        m_pending.push_back(o); m_lastWasDe = false;
    }
    bf_scene* m_h = nullptr;
    bool m_deferred = false, m_lastWasDe = false;
    std::vector<bf_scene_batch_op> m_pending;
    DepthCameraParams m_cam;
};

// ---- consumers of the volume: CUDAMarchingCubesHashSDF (DepthSensing/CUDAMarchingCubesHashSDF.h:8-82) and CUDARayCastSDF (CUDARayCastSDF.h:14-101)
typedef bf_marching_cubes_params MarchingCubesParams;
typedef bf_ray_cast_params RayCastParams;
typedef bf_ray_cast_data RayCastData;
struct vec3f { float x, y, z; vec3f(float a = 0.0f, float b = 0.0f, float c = 0.0f) : x(a), y(b), z(c) {} };

class CUDAMarchingCubesHashSDF {
public:
    explicit CUDAMarchingCubesHashSDF(const MarchingCubesParams& params) : m_params(params) { check(bf_marching_cubes_create(&params, &m_h)); }
    ~CUDAMarchingCubesHashSDF() { bf_marching_cubes_destroy(m_h); }
    CUDAMarchingCubesHashSDF(const CUDAMarchingCubesHashSDF&) = delete;
    // .h:20-29; s_marchingCubesMaxNumTriangles / s_SDFMarchingCubeThreshFactor are not members of bf_global_app_state: pass them
    static MarchingCubesParams parametersFromGlobalAppState(const GlobalAppState& gas, unsigned int marchingCubesMaxNumTriangles = 3000000, float SDFMarchingCubeThreshFactor = 10.0f) {
        MarchingCubesParams p;
        p.m_maxNumTriangles = marchingCubesMaxNumTriangles;
        p.m_threshMarchingCubes = SDFMarchingCubeThreshFactor * gas.s_SDFVoxelSize;
        p.m_threshMarchingCubes2 = SDFMarchingCubeThreshFactor * gas.s_SDFVoxelSize;
        p.m_sdfBlockSize = 8; p.m_hashBucketSize = 4; p.m_hashNumBuckets = gas.s_hashNumBuckets;
        return p;
    }
    void clearMeshBuffer() { check(bf_marching_cubes_clear_mesh_buffer(m_h)); }
    // the RayCastData argument of the reference only carries the sampling helpers (RayCastSDFUtil.h); it is not needed here
    void extractIsoSurface(const HashDataStruct& hashData, const HashParams& hashParams, const RayCastData& /*rayCastData*/,
                           const vec3f& minCorner = vec3f(0.0f, 0.0f, 0.0f), const vec3f& maxCorner = vec3f(0.0f, 0.0f, 0.0f), bool boxEnabled = false) {
        const float mn[3] = {minCorner.x, minCorner.y, minCorner.z}, mx[3] = {maxCorner.x, maxCorner.y, maxCorner.z};
        check(bf_marching_cubes_extract(m_h, &hashData, &hashParams, mn, mx, boxEnabled ? 1 : 0));
    }
    void saveMesh(const std::string& filename, const mat4f* transform = nullptr, bool /*overwriteExistingFile*/ = false) {
        uint32_t nv, nf;
        check(bf_marching_cubes_save_mesh(m_h, filename.c_str(), transform ? transform->m : nullptr, &nv, &nf));
    }
    bf_marching_cubes* handle() const { return m_h; }
private:
    MarchingCubesParams m_params;
    bf_marching_cubes* m_h = nullptr;
};

class CUDARayCastSDF {
public:
    explicit CUDARayCastSDF(const RayCastParams& params) { check(bf_ray_cast_create(&params, &m_h)); }
    ~CUDARayCastSDF() { bf_ray_cast_destroy(m_h); }
    CUDARayCastSDF(const CUDARayCastSDF&) = delete;
    static RayCastParams parametersFromGlobalAppState(const GlobalAppState& gas, const mat4f& intrinsics, const mat4f& /*intrinsicsInv*/) {      // .h:24-52
        RayCastParams p; check(bf_ray_cast_params_from_global_app_state(&gas, intrinsics.m, &p)); return p;
    }
    // render(hashData, hashParams, lastRigidTransform) .cpp:42-72; the depth camera parameters the reference keeps in constant memory travel with the call
    void render(const HashDataStruct& hashData, const HashParams& hashParams, const DepthCameraParams& depthCamera, const mat4f& lastRigidTransform) {
        check(bf_ray_cast_render(m_h, &hashData, &hashParams, &depthCamera, lastRigidTransform.m));
    }
    const RayCastData& getRayCastData() { check(bf_ray_cast_get_data(m_h, &m_data)); return m_data; }
    const RayCastParams& getRayCastParams() { check(bf_ray_cast_get_params(m_h, &m_params)); return m_params; }
    void updateRayCastMinMax(float depthMin, float depthMax) { check(bf_ray_cast_update_min_max(m_h, depthMin, depthMax)); }
    void setRayCastIntrinsics(unsigned int width, unsigned int height, const mat4f& intrinsics, const mat4f& /*intrinsicsInverse*/) {
        check(bf_ray_cast_set_intrinsics(m_h, width, height, intrinsics.m));
    }
    void convertToCameraSpace(const DepthCameraParams& depthCamera) { check(bf_ray_cast_convert_to_camera_space(m_h, &depthCamera)); }
    bf_ray_cast* handle() const { return m_h; }
private:
    bf_ray_cast* m_h = nullptr;
    RayCastData m_data;
    RayCastParams m_params;
};

// ---- CorrespondenceEvaluator (CorrespondenceEvaluator.h:10-141)
struct CorrEvaluation : public bf_corr_evaluation {
    CorrEvaluation() { numCorrect = numDetected = numTotal = 0; }
    float getPrecision() const { return bf_corr_evaluation_get_precision(this); }
    float getRecall() const { return bf_corr_evaluation_get_recall(this); }
    CorrEvaluation& operator+=(const CorrEvaluation& r) { numCorrect += r.numCorrect; numDetected += r.numDetected; numTotal += r.numTotal; return *this; }
};

}  // namespace bundlefusion
