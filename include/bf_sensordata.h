/* bf_sensordata.h — recorded RGB-D sequences (".sens") for the BundleFusion path.
 *
 * Replaces, for the frame loop, the reference's `SensorDataReader` (FriedLiver/Source/SensorDataReader.h/.cpp) and the part
 * of mLib's `ml::SensorData` it calls (`loadFromFile`, `RGBDFrameCacheRead::getNext`, `decompressDepthAlloc`,
 * `decompressColorAlloc`; call sites SensorDataReader.cpp:40-83, 85-128).  mLib is a git submodule of the reference that is
 * NOT vendored under /root/reference (external/mLib is empty, .gitmodules pins no commit), so the container format is restated
 * from its published description (the ScanNet "SensorData" reader documents the same byte layout):
 *
 *   u32  version (= 4)
 *   u64  strlen, char[strlen] sensor name
 *   f32[16] colour intrinsic, f32[16] colour extrinsic, f32[16] depth intrinsic, f32[16] depth extrinsic   (row-major mat4f)
 *   i32  colour compression (-1 unknown, 0 raw RGB8, 1 PNG, 2 JPEG)
 *   i32  depth compression  (-1 unknown, 0 raw u16, 1 zlib-compressed u16, 2 "occi" u16)
 *   u32  colourWidth, colourHeight, depthWidth, depthHeight
 *   f32  depthShift          (depth in metres = u16 / depthShift; 0 = invalid)
 *   u64  numFrames, then per frame:
 *        f32[16] cameraToWorld, u64 timeStampColour, u64 timeStampDepth, u64 colourSizeBytes, u64 depthSizeBytes,
 *        colour bytes, depth bytes
 *   u64  numIMUFrames, then per IMU frame 5 x f64[3] + u64 time stamp (128 B)      (may be absent in older files)
 *
 * All functions run on the host only (no GPU needed); status / error conventions as in bf_hip.h.
 */
#ifndef BF_SENSORDATA_H
#define BF_SENSORDATA_H

#include "bf_pipeline.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { BF_SENS_COLOR_UNKNOWN = -1, BF_SENS_COLOR_RAW = 0, BF_SENS_COLOR_PNG = 1, BF_SENS_COLOR_JPEG = 2 };
enum { BF_SENS_DEPTH_UNKNOWN = -1, BF_SENS_DEPTH_RAW_USHORT = 0, BF_SENS_DEPTH_ZLIB_USHORT = 1, BF_SENS_DEPTH_OCCI_USHORT = 2 };

typedef struct bf_sensor_data_info {          /* ml::SensorData header fields */
    uint32_t versionNumber;
    char sensorName[256];
    float colorIntrinsic[16], colorExtrinsic[16], depthIntrinsic[16], depthExtrinsic[16];
    int32_t colorCompressionType, depthCompressionType;
    uint32_t colorWidth, colorHeight, depthWidth, depthHeight;
    float depthShift;
    uint64_t numFrames, numIMUFrames;
} bf_sensor_data_info;

typedef struct bf_sensor_data bf_sensor_data;               /* a .sens file opened for reading */
typedef struct bf_sensor_data_writer bf_sensor_data_writer; /* a .sens file being written      */

/* colour decoder for PNG / JPEG frames: decode `size` bytes into width*height RGB8; return 0 on success.  ml::SensorData decodes
 * with stb_image; this library has a built-in decoder (baseline JPEG, 8-bit non-interlaced PNG: bf_decode_color_rgb) that is used
 * unless the host application installs its own with bf_sensor_data_set_color_decoder (e.g. for progressive JPEG). */
typedef int (*bf_sens_color_decoder)(void* user, const uint8_t* data, uint64_t size, int32_t compressionType, uint32_t width,
                                     uint32_t height, uint8_t* rgbOut);
BF_API int bf_decode_color_rgb(const uint8_t* data, uint64_t size, int32_t compressionType, uint32_t width, uint32_t height, uint8_t* rgbOut);
/* baseline JPEG (4:4:4, image-optimised Huffman tables) of an RGB8 image, for recording colour frames the way the reference does
 * (TYPE_JPEG, RGBDSensor.cpp:276).  out == NULL: only *size is returned (the stream is kept for the following call with a buffer). */
BF_API int bf_encode_jpeg_rgb(const uint8_t* rgb, uint32_t width, uint32_t height, int32_t quality, uint8_t* out, uint64_t capacity, uint64_t* size);

/* SensorData::loadFromFile (frames are indexed and read on demand, so a file larger than host memory can be played) */
BF_API int bf_sensor_data_open(const char* filename, bf_sensor_data** out);
BF_API int bf_sensor_data_close(bf_sensor_data* sd);
BF_API int bf_sensor_data_get_info(bf_sensor_data* sd, bf_sensor_data_info* out);
/* RGBDSensor::init + initializeDepth/ColorIntrinsics/Extrinsics as done by createFirstConnected (SensorDataReader.cpp:57-63) */
BF_API int bf_sensor_data_get_sensor_desc(bf_sensor_data* sd, bf_rgbd_sensor_desc* out);
BF_API int bf_sensor_data_set_color_decoder(bf_sensor_data* sd, bf_sens_color_decoder fn, void* user);
/* RGBDFrame fields */
BF_API int bf_sensor_data_get_frame_pose(bf_sensor_data* sd, uint64_t frame, float cameraToWorld[16], uint64_t* timeStampColor,
                                         uint64_t* timeStampDepth);
BF_API int bf_sensor_data_get_frame_sizes(bf_sensor_data* sd, uint64_t frame, uint64_t* colorSizeBytes, uint64_t* depthSizeBytes);
/* decompressDepthAlloc: depthWidth*depthHeight u16 */
BF_API int bf_sensor_data_read_depth_raw(bf_sensor_data* sd, uint64_t frame, uint16_t* h_out);
/* what SensorDataReader::processDepth hands to the image manager (:98-103): metres, 0 -> -inf */
BF_API int bf_sensor_data_read_depth(bf_sensor_data* sd, uint64_t frame, float* h_depthMetres);
/* the stored colour bytes as they are (for an external decoder) */
BF_API int bf_sensor_data_read_color_compressed(bf_sensor_data* sd, uint64_t frame, uint8_t* h_out, uint64_t capacity, uint64_t* size);
/* decompressColorAlloc + the vec4uc(vec3uc) widening of processDepth (:107-111): colourWidth*colourHeight RGBX, X = 255.
 * A file without colour (size 0) yields zeros like the reference's untouched m_colorRGBX. */
BF_API int bf_sensor_data_read_color_rgbx(bf_sensor_data* sd, uint64_t frame, uint8_t* h_rgbx);

/* SensorData::saveToFile, incrementally.  info: sensorName, calibration, compression types (colour: frames are stored as
 * given; depth: RAW_USHORT or ZLIB_USHORT, compressed here), sizes, depthShift; numFrames is filled in by _close. */
BF_API int bf_sensor_data_writer_create(const char* filename, const bf_sensor_data_info* info, bf_sensor_data_writer** out);
BF_API int bf_sensor_data_writer_add_frame(bf_sensor_data_writer* w, const float cameraToWorld[16], uint64_t timeStampColor,
                                           uint64_t timeStampDepth, const uint8_t* colorBytes, uint64_t colorSizeBytes,
                                           const uint16_t* depth);
BF_API int bf_sensor_data_writer_close(bf_sensor_data_writer* w);

/* SensorDataReader::saveToFile(filename, trajectory) (SensorDataReader.cpp:152-165, "kind of a hack"): the same frames with
 * cameraToWorld replaced by trajectory[i] (16 floats each) for i < numTransforms and by an all -inf matrix for the rest. */
BF_API int bf_sensor_data_save_with_trajectory(bf_sensor_data* sd, const char* filename, const float* trajectory, uint64_t numTransforms);
/* RGBDSensor::saveRecordedFramesToFile (RGBDSensor.cpp:353-398): only the first numTransforms frames, each with its pose; more
 * transforms than frames is an error ("something went wrong; found more transforms than frames"). */
BF_API int bf_sensor_data_save_recorded(bf_sensor_data* sd, const char* filename, const float* trajectory, uint64_t numTransforms);

/* PoseHelper::evaluateAteRmse (PoseHelper.h:35-79): absolute trajectory error after a rigid (Kabsch) alignment of the camera
 * positions; transforms whose first element is -inf are skipped on either side.  *rmse = -inf when fewer than 3 transforms are
 * given (except the reference's special case for 2).  *numEvaluated = number of positions used.  mLib's EigenWrapperf::kabsch is
 * not in the tree: the alignment here is the textbook SVD solution (double precision). */
BF_API int bf_evaluate_ate_rmse(const float* trajectory, const float* referenceTrajectory, uint32_t numTransforms, float* rmse,
                                uint32_t* numEvaluated);
/* PoseHelper::getAlignmentBetweenTrajectories (PoseHelper.h:81-108): the rigid transform itself (all -inf when it cannot be evaluated) */
BF_API int bf_trajectory_alignment(const float* trajectory, const float* referenceTrajectory, uint32_t numTransforms, float align[16]);
/* PoseHelper::evaluateErr2PerImage (:110-146): squared position error of every pose that is valid on both sides, after the alignment;
 * imageIndices / err2 need room for numTransforms entries */
BF_API int bf_evaluate_err2_per_image(const float* trajectory, const float* referenceTrajectory, uint32_t numTransforms, uint32_t* imageIndices,
                                      float* err2, uint32_t* count);
/* SensorDataReader::evaluateTrajectory (:167-189): the file's cameraToWorld poses, re-based so that the first is identity, are
 * the reference. */
BF_API int bf_sensor_data_evaluate_trajectory(bf_sensor_data* sd, const float* trajectory, uint64_t numTransforms, float* rmse,
                                              uint32_t* numEvaluated);

#ifdef __cplusplus
}
#endif
#endif /* BF_SENSORDATA_H */
