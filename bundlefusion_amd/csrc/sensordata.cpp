// Recorded sequences: the ".sens" container (ml::SensorData version 4) as used by SensorDataReader.cpp:40-128.
// Host-only.  The byte layout is documented in include/bf_sensordata.h; frames are indexed at open time and read on demand.
#include <zlib.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bf_sensordata.h"
#include "bf_internal.h"

using namespace bf;

struct bf_sensor_data {
    FILE* f = nullptr;
    bf_sensor_data_info info;
    struct Frame { float T[16]; uint64_t tsColor, tsDepth, colorSize, depthSize; int64_t colorOffset, depthOffset; };
    std::vector<Frame> frames;
    bf_sens_color_decoder decoder = nullptr;
    void* decoderUser = nullptr;
    std::vector<uint8_t> scratch;
};

struct bf_sensor_data_writer {
    FILE* f = nullptr;
    bf_sensor_data_info info;
    int64_t numFramesOffset = 0;
    uint64_t numFrames = 0;
    std::vector<uint8_t> scratch;
};

namespace {

const uint32_t SENS_VERSION = 4;                 // M_SENS_VERSION_NUMBER of the SensorData revision BundleFusion was written against
const uint64_t IMU_FRAME_BYTES = 5 * 3 * 8 + 8;  // rotationRate, acceleration, magneticField, attitude, gravity (vec3d) + timeStamp

template <class T> bool rd(FILE* f, T* v, size_t n = 1) { return fread(v, sizeof(T), n, f) == n; }
template <class T> bool wr(FILE* f, const T* v, size_t n = 1) { return fwrite(v, sizeof(T), n, f) == n; }

int64_t fileSize(FILE* f) {
    const int64_t cur = ftello(f);
    fseeko(f, 0, SEEK_END);
    const int64_t end = ftello(f);
    fseeko(f, cur, SEEK_SET);
    return end;
}

// no C++ exception leaves the C ABI: allocation failures on absurd (but formally valid) inputs become status codes
template <class F> int guarded(F f) {
    try { return f(); }
    catch (const std::bad_alloc&) { set_error("out of host memory"); return BF_ERR_CAPACITY; }
    catch (const std::exception& e) { set_error("%s", e.what()); return BF_ERR_STATE; }
}

int readBytes(bf_sensor_data* sd, int64_t offset, uint64_t size, uint8_t* out) {
    if (fseeko(sd->f, offset, SEEK_SET) != 0 || fread(out, 1, size, sd->f) != size) { set_error("sens: short read at offset %lld", (long long)offset); return BF_ERR_STATE; }
    return BF_OK;
}

}  // namespace

extern "C" {

int bf_sensor_data_open(const char* filename, bf_sensor_data** out) {
    return guarded([&]() -> int {
    BF_REQUIRE(filename && out, "null argument");
    FILE* f = fopen(filename, "rb");
    if (!f) { set_error("could not open file %s", filename); return BF_ERR_INVALID_ARG; }          // "could not open file" (SensorData::loadFromFile)
    struct Closer { void operator()(bf_sensor_data* p) const { bf_sensor_data_close(p); } };
    std::unique_ptr<bf_sensor_data, Closer> guard(new bf_sensor_data);          // an exception below (bad_alloc) must not leak sd / its FILE*
    bf_sensor_data* sd = guard.get();
    sd->f = f;
    bf_sensor_data_info& h = sd->info;
    memset(&h, 0, sizeof h);
    const int64_t total = fileSize(f);
    bool ok = rd(f, &h.versionNumber);
    if (ok && h.versionNumber != SENS_VERSION) {
        set_error("sens: version %u, expected %u", h.versionNumber, SENS_VERSION);                   // "Invalid file format" / version mismatch
        return BF_ERR_INVALID_ARG;
    }
    uint64_t strLen = 0;
    ok = ok && rd(f, &strLen) && strLen <= (uint64_t)total;
    if (ok) {
        std::string name(strLen, '\0');
        ok = strLen == 0 || rd(f, &name[0], strLen);
        snprintf(h.sensorName, sizeof h.sensorName, "%s", name.c_str());
    }
    ok = ok && rd(f, h.colorIntrinsic, 16) && rd(f, h.colorExtrinsic, 16) && rd(f, h.depthIntrinsic, 16) && rd(f, h.depthExtrinsic, 16) &&
         rd(f, &h.colorCompressionType) && rd(f, &h.depthCompressionType) && rd(f, &h.colorWidth) && rd(f, &h.colorHeight) &&
         rd(f, &h.depthWidth) && rd(f, &h.depthHeight) && rd(f, &h.depthShift) && rd(f, &h.numFrames);
    // a frame record is at least 64 + 4 * 8 = 96 bytes (pose, two time stamps, two sizes): bound numFrames by what the rest of the
    // file can hold before frames.resize() (112 B of bookkeeping per frame) — a small forged header cannot ask for 100x the file size
    const int64_t afterHeader = ftello(f);
    if (!ok || afterHeader < 0 || h.numFrames > (uint64_t)(total - afterHeader) / 96u) { set_error("sens: truncated or invalid header in %s", filename); return BF_ERR_INVALID_ARG; }
    const uint32_t MAXDIM = 1u << 15;                                      // image buffers are sized from these fields
    if (h.depthWidth > MAXDIM || h.depthHeight > MAXDIM || h.colorWidth > MAXDIM || h.colorHeight > MAXDIM || !(h.depthShift > 0.0f) || !std::isfinite(h.depthShift)) {
        set_error("sens: implausible header in %s (depth %ux%u, colour %ux%u, depthShift %g)", filename, h.depthWidth, h.depthHeight, h.colorWidth, h.colorHeight, (double)h.depthShift);
        return BF_ERR_INVALID_ARG;
    }
    sd->frames.resize(h.numFrames);
    for (uint64_t i = 0; i < h.numFrames; ++i) {
        bf_sensor_data::Frame& fr = sd->frames[i];
        ok = rd(f, fr.T, 16) && rd(f, &fr.tsColor) && rd(f, &fr.tsDepth) && rd(f, &fr.colorSize) && rd(f, &fr.depthSize);
        if (ok) {
            fr.colorOffset = ftello(f);
            fr.depthOffset = fr.colorOffset + (int64_t)fr.colorSize;
            ok = fr.colorSize <= (uint64_t)total && fr.depthSize <= (uint64_t)total && fr.depthOffset + (int64_t)fr.depthSize <= total &&
                 fseeko(f, fr.depthOffset + (int64_t)fr.depthSize, SEEK_SET) == 0;
        }
        if (!ok) { set_error("sens: truncated at frame %llu of %llu in %s", (unsigned long long)i, (unsigned long long)h.numFrames, filename); return BF_ERR_INVALID_ARG; }
    }
    uint64_t numIMU = 0;
    if (rd(f, &numIMU)) {                                                   // the IMU block is optional; no multiplication that could wrap
        const int64_t pos = ftello(f);
        if (pos >= 0 && pos <= total && numIMU <= (uint64_t)(total - pos) / IMU_FRAME_BYTES) h.numIMUFrames = numIMU;
    }
    *out = guard.release();
    return BF_OK;
    });
}

int bf_sensor_data_close(bf_sensor_data* sd) {
    if (!sd) return BF_OK;
    if (sd->f) fclose(sd->f);
    delete sd;
    return BF_OK;
}

int bf_sensor_data_get_info(bf_sensor_data* sd, bf_sensor_data_info* out) { BF_REQUIRE(sd && out, "null argument"); *out = sd->info; return BF_OK; }

int bf_sensor_data_get_sensor_desc(bf_sensor_data* sd, bf_rgbd_sensor_desc* out) {                // SensorDataReader.cpp:57-63
    BF_REQUIRE(sd && out, "null argument");
    const bf_sensor_data_info& h = sd->info;
    memset(out, 0, sizeof *out);
    out->depthWidth = h.depthWidth; out->depthHeight = h.depthHeight;
    out->colorWidth = h.colorWidth > 1u ? h.colorWidth : 1u; out->colorHeight = h.colorHeight > 1u ? h.colorHeight : 1u;   // std::max(.., 1u) :57
    // initializeDepth/ColorIntrinsics(fx, fy, mx, my) build a 4x4 from the four scalars (RGBDSensor.cpp:142-147, 161-166); extrinsics are taken whole
    const float* src[2] = {h.depthIntrinsic, h.colorIntrinsic};
    float* dst[2] = {out->depthIntrinsics, out->colorIntrinsics};
    for (int k = 0; k < 2; ++k) {
        float* K = dst[k];
        for (int i = 0; i < 16; ++i) K[i] = 0.0f;
        K[0] = src[k][0]; K[5] = src[k][5]; K[2] = src[k][2]; K[6] = src[k][6]; K[10] = 1.0f; K[15] = 1.0f;
    }
    memcpy(out->depthExtrinsics, h.depthExtrinsic, 64);
    memcpy(out->colorExtrinsics, h.colorExtrinsic, 64);
    return BF_OK;
}

int bf_sensor_data_set_color_decoder(bf_sensor_data* sd, bf_sens_color_decoder fn, void* user) {
    BF_REQUIRE(sd, "null argument");
    sd->decoder = fn; sd->decoderUser = user;
    return BF_OK;
}

int bf_sensor_data_get_frame_pose(bf_sensor_data* sd, uint64_t frame, float T[16], uint64_t* tsColor, uint64_t* tsDepth) {
    BF_REQUIRE(sd && frame < sd->frames.size(), "frame index out of range");
    if (T) memcpy(T, sd->frames[frame].T, 64);
    if (tsColor) *tsColor = sd->frames[frame].tsColor;
    if (tsDepth) *tsDepth = sd->frames[frame].tsDepth;
    return BF_OK;
}

int bf_sensor_data_get_frame_sizes(bf_sensor_data* sd, uint64_t frame, uint64_t* colorSize, uint64_t* depthSize) {
    BF_REQUIRE(sd && frame < sd->frames.size(), "frame index out of range");
    if (colorSize) *colorSize = sd->frames[frame].colorSize;
    if (depthSize) *depthSize = sd->frames[frame].depthSize;
    return BF_OK;
}

int bf_sensor_data_read_depth_raw(bf_sensor_data* sd, uint64_t frame, uint16_t* out) {             // SensorData::decompressDepthAlloc
    return guarded([&]() -> int {
    BF_REQUIRE(sd && out && frame < sd->frames.size(), "bad argument");
    const bf_sensor_data::Frame& fr = sd->frames[frame];
    const uint64_t bytes = (uint64_t)sd->info.depthWidth * sd->info.depthHeight * 2;
    if (sd->info.depthCompressionType == BF_SENS_DEPTH_RAW_USHORT) {
        if (fr.depthSize != bytes) { set_error("sens: frame %llu: raw depth has %llu bytes, expected %llu", (unsigned long long)frame, (unsigned long long)fr.depthSize, (unsigned long long)bytes); return BF_ERR_STATE; }
        return readBytes(sd, fr.depthOffset, bytes, reinterpret_cast<uint8_t*>(out));
    }
    if (sd->info.depthCompressionType == BF_SENS_DEPTH_ZLIB_USHORT) {
        sd->scratch.resize(fr.depthSize);
        const int rc = readBytes(sd, fr.depthOffset, fr.depthSize, sd->scratch.data());
        if (rc) return rc;
        uLongf dstLen = (uLongf)bytes;
        const int z = uncompress(reinterpret_cast<Bytef*>(out), &dstLen, sd->scratch.data(), (uLong)fr.depthSize);
        if (z != Z_OK || dstLen != bytes) { set_error("sens: frame %llu: zlib depth does not inflate to %llu bytes (zlib %d)", (unsigned long long)frame, (unsigned long long)bytes, z); return BF_ERR_STATE; }
        return BF_OK;
    }
    set_error("sens: depth compression type %d is not supported (raw and zlib u16 are)", sd->info.depthCompressionType);   // "unknown depth compression type"
    return BF_ERR_INVALID_ARG;
    });
}

int bf_sensor_data_read_depth(bf_sensor_data* sd, uint64_t frame, float* out) {                    // SensorDataReader::processDepth :98-103
    return guarded([&]() -> int {
    BF_REQUIRE(sd && out && frame < sd->frames.size(), "bad argument");
    const size_t n = (size_t)sd->info.depthWidth * sd->info.depthHeight;
    std::vector<uint16_t> raw(n);
    const int rc = bf_sensor_data_read_depth_raw(sd, frame, raw.data());
    if (rc) return rc;
    const float shift = sd->info.depthShift;
    for (size_t i = 0; i < n; ++i) out[i] = raw[i] == 0 ? -std::numeric_limits<float>::infinity() : (float)raw[i] / shift;
    return BF_OK;
    });
}

int bf_sensor_data_read_color_compressed(bf_sensor_data* sd, uint64_t frame, uint8_t* out, uint64_t capacity, uint64_t* size) {
    BF_REQUIRE(sd && size && frame < sd->frames.size(), "bad argument");
    const bf_sensor_data::Frame& fr = sd->frames[frame];
    *size = fr.colorSize;
    if (!out) return BF_OK;                                                       // size query
    BF_REQUIRE(capacity >= fr.colorSize, "buffer too small");
    return readBytes(sd, fr.colorOffset, fr.colorSize, out);
}

int bf_sensor_data_read_color_rgbx(bf_sensor_data* sd, uint64_t frame, uint8_t* out) {            // decompressColorAlloc + :107-111
    return guarded([&]() -> int {
    BF_REQUIRE(sd && out && frame < sd->frames.size(), "bad argument");
    const bf_sensor_data::Frame& fr = sd->frames[frame];
    const size_t n = (size_t)sd->info.colorWidth * sd->info.colorHeight;
    if (fr.colorSize == 0) { memset(out, 0, std::max<size_t>(n, 1) * 4); return BF_OK; }            // m_bHasColorData == false
    std::vector<uint8_t> rgb(n * 3);
    if (sd->info.colorCompressionType == BF_SENS_COLOR_RAW) {
        if (fr.colorSize != n * 3) { set_error("sens: frame %llu: raw colour has %llu bytes, expected %llu", (unsigned long long)frame, (unsigned long long)fr.colorSize, (unsigned long long)(n * 3)); return BF_ERR_STATE; }
        const int rc = readBytes(sd, fr.colorOffset, fr.colorSize, rgb.data());
        if (rc) return rc;
    } else if (sd->info.colorCompressionType == BF_SENS_COLOR_PNG || sd->info.colorCompressionType == BF_SENS_COLOR_JPEG) {
        sd->scratch.resize(fr.colorSize);
        const int rc = readBytes(sd, fr.colorOffset, fr.colorSize, sd->scratch.data());
        if (rc) return rc;
        if (sd->decoder) {
            if (sd->decoder(sd->decoderUser, sd->scratch.data(), fr.colorSize, sd->info.colorCompressionType, sd->info.colorWidth, sd->info.colorHeight, rgb.data()) != 0) {
                set_error("sens: frame %llu: the colour decoder failed", (unsigned long long)frame);
                return BF_ERR_STATE;
            }
        } else {
            const int drc = bf_decode_color_rgb(sd->scratch.data(), fr.colorSize, sd->info.colorCompressionType, sd->info.colorWidth, sd->info.colorHeight, rgb.data());
            if (drc) return drc;                                                   // message set by the decoder
        }
    } else {
        set_error("sens: colour compression type %d is not supported", sd->info.colorCompressionType);                 // "unknown compression type"
        return BF_ERR_INVALID_ARG;
    }
    for (size_t i = 0; i < n; ++i) { out[4 * i] = rgb[3 * i]; out[4 * i + 1] = rgb[3 * i + 1]; out[4 * i + 2] = rgb[3 * i + 2]; out[4 * i + 3] = 255; }   // vec4uc(vec3uc): w = 255
    return BF_OK;
    });
}

// ------------------------------------------------------------------------------------------------ writer
int bf_sensor_data_writer_create(const char* filename, const bf_sensor_data_info* info, bf_sensor_data_writer** out) {
    BF_REQUIRE(filename && info && out, "null argument");
    BF_REQUIRE(info->depthCompressionType == BF_SENS_DEPTH_RAW_USHORT || info->depthCompressionType == BF_SENS_DEPTH_ZLIB_USHORT, "depth compression must be raw or zlib u16");
    BF_REQUIRE(info->depthWidth > 0 && info->depthHeight > 0 && info->depthShift > 0.0f, "bad depth geometry");
    FILE* f = fopen(filename, "wb");
    if (!f) { set_error("could not open file %s for writing", filename); return BF_ERR_INVALID_ARG; }
    bf_sensor_data_writer* w = new bf_sensor_data_writer;
    w->f = f; w->info = *info;
    const bf_sensor_data_info& h = w->info;
    const uint32_t version = SENS_VERSION;
    const uint64_t strLen = strnlen(h.sensorName, sizeof h.sensorName);
    const uint64_t zero = 0;
    bool ok = wr(f, &version) && wr(f, &strLen) && (strLen == 0 || wr(f, h.sensorName, strLen)) && wr(f, h.colorIntrinsic, 16) && wr(f, h.colorExtrinsic, 16) &&
              wr(f, h.depthIntrinsic, 16) && wr(f, h.depthExtrinsic, 16) && wr(f, &h.colorCompressionType) && wr(f, &h.depthCompressionType) &&
              wr(f, &h.colorWidth) && wr(f, &h.colorHeight) && wr(f, &h.depthWidth) && wr(f, &h.depthHeight) && wr(f, &h.depthShift);
    w->numFramesOffset = ftello(f);
    ok = ok && wr(f, &zero);
    if (!ok) { set_error("sens: write failed (%s)", filename); fclose(f); delete w; return BF_ERR_STATE; }
    *out = w;
    return BF_OK;
}

int bf_sensor_data_writer_add_frame(bf_sensor_data_writer* w, const float T[16], uint64_t tsColor, uint64_t tsDepth, const uint8_t* colorBytes,
                                    uint64_t colorSize, const uint16_t* depth) {
    return guarded([&]() -> int {
    BF_REQUIRE(w && T && depth && (colorBytes || colorSize == 0), "null argument");
    const uint64_t rawBytes = (uint64_t)w->info.depthWidth * w->info.depthHeight * 2;
    const uint8_t* depthBytes = reinterpret_cast<const uint8_t*>(depth);
    uint64_t depthSize = rawBytes;
    if (w->info.depthCompressionType == BF_SENS_DEPTH_ZLIB_USHORT) {
        uLongf bound = compressBound((uLong)rawBytes);
        w->scratch.resize(bound);
        if (compress2(w->scratch.data(), &bound, depthBytes, (uLong)rawBytes, 8) != Z_OK) { set_error("sens: zlib compression failed"); return BF_ERR_STATE; }
        depthBytes = w->scratch.data(); depthSize = bound;
    }
    FILE* f = w->f;
    const bool ok = wr(f, T, 16) && wr(f, &tsColor) && wr(f, &tsDepth) && wr(f, &colorSize) && wr(f, &depthSize) &&
                    (colorSize == 0 || wr(f, colorBytes, colorSize)) && wr(f, depthBytes, depthSize);
    if (!ok) { set_error("sens: write failed at frame %llu", (unsigned long long)w->numFrames); return BF_ERR_STATE; }
    w->numFrames++;
    return BF_OK;
    });
}

int bf_sensor_data_writer_close(bf_sensor_data_writer* w) {
    if (!w) return BF_OK;
    const uint64_t zero = 0;
    bool ok = wr(w->f, &zero);                                                    // no IMU frames
    ok = ok && fseeko(w->f, w->numFramesOffset, SEEK_SET) == 0 && wr(w->f, &w->numFrames);
    ok = (fclose(w->f) == 0) && ok;
    delete w;
    if (!ok) { set_error("sens: finishing the file failed"); return BF_ERR_STATE; }
    return BF_OK;
}

// ------------------------------------------------------------------------------------------------ trajectory I/O and evaluation
static int saveCopy(bf_sensor_data* sd, const char* filename, const float* trajectory, uint64_t numTransforms, bool dropRest) {
    FILE* f = fopen(filename, "wb");
    if (!f) { set_error("could not open file %s for writing", filename); return BF_ERR_INVALID_ARG; }
    const bf_sensor_data_info& h = sd->info;
    const uint32_t version = SENS_VERSION;
    const uint64_t strLen = strnlen(h.sensorName, sizeof h.sensorName), zero = 0;
    const uint64_t numFrames = dropRest ? std::min<uint64_t>(numTransforms, sd->frames.size()) : sd->frames.size();
    bool ok = wr(f, &version) && wr(f, &strLen) && (strLen == 0 || wr(f, h.sensorName, strLen)) && wr(f, h.colorIntrinsic, 16) && wr(f, h.colorExtrinsic, 16) &&
              wr(f, h.depthIntrinsic, 16) && wr(f, h.depthExtrinsic, 16) && wr(f, &h.colorCompressionType) && wr(f, &h.depthCompressionType) &&
              wr(f, &h.colorWidth) && wr(f, &h.colorHeight) && wr(f, &h.depthWidth) && wr(f, &h.depthHeight) && wr(f, &h.depthShift) && wr(f, &numFrames);
    float invalid[16];
    for (float& v : invalid) v = -std::numeric_limits<float>::infinity();
    std::vector<uint8_t> buf;
    for (uint64_t i = 0; ok && i < numFrames; ++i) {
        const bf_sensor_data::Frame& fr = sd->frames[i];
        const float* T = i < numTransforms ? trajectory + 16 * i : invalid;
        buf.resize(fr.colorSize + fr.depthSize);
        if (readBytes(sd, fr.colorOffset, fr.colorSize + fr.depthSize, buf.data()) != BF_OK) { fclose(f); return BF_ERR_STATE; }
        ok = wr(f, T, 16) && wr(f, &fr.tsColor) && wr(f, &fr.tsDepth) && wr(f, &fr.colorSize) && wr(f, &fr.depthSize) && (buf.empty() || wr(f, buf.data(), buf.size()));
    }
    ok = ok && wr(f, &zero);                                                      // IMU frames are not carried over
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("sens: writing %s failed", filename); return BF_ERR_STATE; }
    return BF_OK;
}

int bf_sensor_data_save_with_trajectory(bf_sensor_data* sd, const char* filename, const float* trajectory, uint64_t numTransforms) {
    return guarded([&]() -> int {
    BF_REQUIRE(sd && filename && (trajectory || numTransforms == 0), "null argument");
    return saveCopy(sd, filename, trajectory, numTransforms, false);
    });
}

int bf_sensor_data_save_recorded(bf_sensor_data* sd, const char* filename, const float* trajectory, uint64_t numTransforms) {
    return guarded([&]() -> int {
    BF_REQUIRE(sd && filename && trajectory && numTransforms > 0, "null argument");
    if (numTransforms > sd->frames.size()) { set_error("something went wrong; found more transforms than frames"); return BF_ERR_INVALID_ARG; }   // RGBDSensor.cpp:365
    return saveCopy(sd, filename, trajectory, numTransforms, true);
    });
}

}  // extern "C"

namespace {

// symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (double precision): A = V diag(w) V^T
void jacobiEigen3(double A[3][3], double V[3][3], double w[3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

// rigid alignment R p + t ~ r (Kabsch): R = argmax tr(R H), H = sum (p - pc)(r - rc)^T, via the polar factor of H^T
void kabschAlign(const std::vector<std::array<double, 3>>& p, const std::vector<std::array<double, 3>>& r, double R[3][3], double t[3]) {
    const size_t n = p.size();
    double pc[3] = {0, 0, 0}, rc[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { pc[k] += p[i][k]; rc[k] += r[i][k]; }
    for (int k = 0; k < 3; ++k) { pc[k] /= (double)n; rc[k] /= (double)n; }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[a][b] += (p[i][a] - pc[a]) * (r[i][b] - rc[b]);
    // H = U S V^T  =>  R = V D U^T with D = diag(1, 1, det(V U^T)).  V, S^2 from the eigen-decomposition of H^T H; U = H V S^-1.
    double HtH[3][3], V[3][3], w[3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { HtH[a][b] = 0; for (int k = 0; k < 3; ++k) HtH[a][b] += H[k][a] * H[k][b]; }
    jacobiEigen3(HtH, V, w);
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return w[a] > w[b]; });
    double Vs[3][3], U[3][3], sv[3];
    for (int j = 0; j < 3; ++j) { sv[j] = sqrt(std::max(w[order[j]], 0.0)); for (int k = 0; k < 3; ++k) Vs[k][j] = V[k][order[j]]; }
    for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) { double v = 0; for (int m = 0; m < 3; ++m) v += H[k][m] * Vs[m][j]; U[k][j] = v; }
    // orthonormalise U's columns (Gram-Schmidt); a vanishing singular value (planar / collinear positions) is completed by a cross product
    auto norm = [](double* v) { const double l = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (l > 0) { v[0] /= l; v[1] /= l; v[2] /= l; } return l; };
    double u0[3] = {U[0][0], U[1][0], U[2][0]}, u1[3] = {U[0][1], U[1][1], U[2][1]}, u2[3];
    if (norm(u0) == 0.0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; }
    { const double d = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2]; for (int k = 0; k < 3; ++k) u1[k] -= d * u0[k]; }
    if (norm(u1) < 1e-12 || sv[1] <= 1e-12 * std::max(sv[0], 1e-300)) {          // rank 1: any unit vector orthogonal to u0
        const int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1;
        u1[0] = u0[1] * e[2] - u0[2] * e[1]; u1[1] = u0[2] * e[0] - u0[0] * e[2]; u1[2] = u0[0] * e[1] - u0[1] * e[0];
        norm(u1);                                                                   // collinear positions: any rotation about the line is optimal
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1]; u2[1] = u0[2] * u1[0] - u0[0] * u1[2]; u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    double v0[3] = {Vs[0][0], Vs[1][0], Vs[2][0]}, v1[3] = {Vs[0][1], Vs[1][1], Vs[2][1]}, v2[3];
    v2[0] = v0[1] * v1[2] - v0[2] * v1[1]; v2[1] = v0[2] * v1[0] - v0[0] * v1[2]; v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
    // u2 = u0 x u1 and v2 = v0 x v1 make both bases right-handed, so R = sum v_j u_j^T is a proper rotation; this equals the usual
    // V diag(1, 1, det(V U^T)) U^T (the direction of the smallest singular value is the one that flips when H contains a reflection)
    const double* uu[3] = {u0, u1, u2};
    const double* vv[3] = {v0, v1, v2};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { R[a][b] = 0; for (int j = 0; j < 3; ++j) R[a][b] += vv[j][a] * uu[j][b]; }
    for (int a = 0; a < 3; ++a) { t[a] = rc[a]; for (int b = 0; b < 3; ++b) t[a] -= R[a][b] * pc[b]; }
}

}  // namespace

extern "C" {

static void validPositions(const float* traj, const float* ref, uint32_t n, std::vector<std::array<double, 3>>& pts, std::vector<std::array<double, 3>>& refPts,
                           std::vector<uint32_t>* indices) {
    const float NINF = -std::numeric_limits<float>::infinity();
    auto tr = [](const float* T, uint32_t i, int k) { return (double)T[16 * (size_t)i + 4 * k + 3]; };
    for (uint32_t i = 0; i < n; ++i)
        if (traj[16 * (size_t)i] != NINF && ref[16 * (size_t)i] != NINF) {
            pts.push_back({tr(traj, i, 0), tr(traj, i, 1), tr(traj, i, 2)});
            refPts.push_back({tr(ref, i, 0), tr(ref, i, 1), tr(ref, i, 2)});
            if (indices) indices->push_back(i);
        }
}

int bf_evaluate_ate_rmse(const float* traj, const float* ref, uint32_t numTransforms, float* rmse, uint32_t* numEvaluated) {     // PoseHelper.h:35-79
    return guarded([&]() -> int {
    BF_REQUIRE((traj && ref) || numTransforms == 0, "null argument");
    BF_REQUIRE(rmse && numEvaluated, "null argument");
    const float NINF = -std::numeric_limits<float>::infinity();
    auto tr = [](const float* T, uint32_t i, int k) { return (double)T[16 * (size_t)i + 4 * k + 3]; };
    if (numTransforms < 3) {
        *rmse = NINF; *numEvaluated = numTransforms;
        if (numTransforms == 2) {
            const double l = sqrt(tr(ref, 0, 0) * tr(ref, 0, 0) + tr(ref, 0, 1) * tr(ref, 0, 1) + tr(ref, 0, 2) * tr(ref, 0, 2));
            if (!(l > 0.0001)) {
                double d = 0;
                for (int k = 0; k < 3; ++k) d += (tr(traj, 1, k) - tr(ref, 1, k)) * (tr(traj, 1, k) - tr(ref, 1, k));
                *rmse = (float)sqrt(d);
            }
        }
        return BF_OK;
    }
    std::vector<std::array<double, 3>> pts, refPts;
    validPositions(traj, ref, numTransforms, pts, refPts, nullptr);
    if (pts.empty()) { *rmse = NINF; *numEvaluated = 0; return BF_OK; }                    // "ERROR no points to evaluate"
    double R[3][3], t[3];
    kabschAlign(pts, refPts, R, t);
    double err = 0;
    for (size_t i = 0; i < pts.size(); ++i)
        for (int a = 0; a < 3; ++a) {
            double v = t[a] - refPts[i][a];
            for (int b = 0; b < 3; ++b) v += R[a][b] * pts[i][b];
            err += v * v;
        }
    *rmse = (float)sqrt(err / (double)pts.size());
    *numEvaluated = (uint32_t)pts.size();
    return BF_OK;
    });
}

int bf_trajectory_alignment(const float* traj, const float* ref, uint32_t numTransforms, float align[16]) {                       // PoseHelper.h:81-108
    return guarded([&]() -> int {
    BF_REQUIRE(align && ((traj && ref) || numTransforms == 0), "null argument");
    for (int i = 0; i < 16; ++i) align[i] = -std::numeric_limits<float>::infinity();          // ret.setZero(-inf): "cannot evaluate"
    if (numTransforms < 3) return BF_OK;
    std::vector<std::array<double, 3>> pts, refPts;
    validPositions(traj, ref, numTransforms, pts, refPts, nullptr);
    if (pts.empty()) return BF_OK;
    double R[3][3], t[3];
    kabschAlign(pts, refPts, R, t);
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) align[4 * a + b] = (float)R[a][b]; align[4 * a + 3] = (float)t[a]; }
    align[12] = align[13] = align[14] = 0.0f; align[15] = 1.0f;
    return BF_OK;
    });
}

int bf_evaluate_err2_per_image(const float* traj, const float* ref, uint32_t numTransforms, uint32_t* imageIndices, float* err2, uint32_t* count) {   // :110-146
    return guarded([&]() -> int {
    BF_REQUIRE(imageIndices && err2 && count && ((traj && ref) || numTransforms == 0), "null argument");
    *count = 0;
    auto tr = [](const float* T, uint32_t i, int k) { return (double)T[16 * (size_t)i + 4 * k + 3]; };
    if (numTransforms < 3) {
        if (numTransforms == 2) {
            const double l = sqrt(tr(ref, 0, 0) * tr(ref, 0, 0) + tr(ref, 0, 1) * tr(ref, 0, 1) + tr(ref, 0, 2) * tr(ref, 0, 2));
            if (!(l > 0.0001)) {
                double d = 0;
                for (int k = 0; k < 3; ++k) d += (tr(traj, 1, k) - tr(ref, 1, k)) * (tr(traj, 1, k) - tr(ref, 1, k));
                imageIndices[0] = 0; err2[0] = (float)sqrt(d); *count = 1;                     // the reference stores the distance (not squared) here
            }
        }
        return BF_OK;
    }
    std::vector<std::array<double, 3>> pts, refPts;
    std::vector<uint32_t> idx;
    validPositions(traj, ref, numTransforms, pts, refPts, &idx);
    if (pts.empty()) return BF_OK;
    double R[3][3], t[3];
    kabschAlign(pts, refPts, R, t);
    for (size_t i = 0; i < pts.size(); ++i) {
        double e = 0;
        for (int a = 0; a < 3; ++a) {
            double v = t[a] - refPts[i][a];
            for (int b = 0; b < 3; ++b) v += R[a][b] * pts[i][b];
            e += v * v;
        }
        imageIndices[i] = idx[i]; err2[i] = (float)e;
    }
    *count = (uint32_t)pts.size();
    return BF_OK;
    });
}

int bf_sensor_data_evaluate_trajectory(bf_sensor_data* sd, const float* trajectory, uint64_t numTransforms, float* rmse, uint32_t* numEvaluated) {   // :167-189
    return guarded([&]() -> int {
    BF_REQUIRE(sd && trajectory && rmse && numEvaluated, "null argument");
    const size_t nRef = sd->frames.size();
    BF_REQUIRE(nRef > 0, "no frames");
    // offset = referenceTrajectory.front().getInverse(); reference[i] = offset * reference[i]  (row-major 4x4, general inverse of a rigid pose)
    const float* F = sd->frames[0].T;
    double Rinv[3][3], tinv[3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Rinv[a][b] = F[4 * b + a];
    for (int a = 0; a < 3; ++a) { tinv[a] = 0; for (int b = 0; b < 3; ++b) tinv[a] -= Rinv[a][b] * F[4 * b + 3]; }
    std::vector<float> ref(16 * nRef);
    const float NINF = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < nRef; ++i) {
        const float* T = sd->frames[i].T;
        float* o = ref.data() + 16 * i;
        if (T[0] == NINF) { for (int k = 0; k < 16; ++k) o[k] = NINF; continue; }
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 4; ++b) { double v = b == 3 ? tinv[a] : 0.0; for (int k = 0; k < 3; ++k) v += Rinv[a][k] * T[4 * k + b]; o[4 * a + b] = (float)v; }
        }
        o[12] = o[13] = o[14] = 0.0f; o[15] = 1.0f;
    }
    const uint32_t n = (uint32_t)std::min<uint64_t>(numTransforms, nRef);
    return bf_evaluate_ate_rmse(trajectory, ref.data(), n, rmse, numEvaluated);
    });
}

}  // extern "C"
