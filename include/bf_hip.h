/*
 * bf_hip.h — C ABI of libbf_hip.so, the MI355X (gfx950) implementation of the
 * BundleFusion pose-optimisation + volumetric (re-)integration hot path.
 *
 * Every entry point replaces one method of the reference's operator surface
 * (paths relative to /root/reference/FriedLiver/Source).  Plain pointers and
 * sizes only; all `d_*` pointers are DEVICE pointers on the current HIP device,
 * all `const float m[16]` matrices are row-major 4x4 (the reference's
 * float4x4 / mat4f layout) in HOST memory.
 *
 * Conventions
 *   - every function returns BF_OK (0) or a negative bf_status; the message of
 *     the last failure on the calling thread is available via bf_last_error().
 *     (the reference throws MLIB_EXCEPTION; a C ABI cannot.)
 *   - work is enqueued on the handle's HIP stream (bf_*_set_stream; default is
 *     the NULL stream like the reference).  Functions that return a host value
 *     (counts) synchronise that stream; everything else is asynchronous.
 */
#ifndef BF_HIP_H
#define BF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_API __attribute__((visibility("default")))

typedef enum bf_status {
    BF_OK = 0,
    BF_ERR_INVALID_ARG = -1,
    BF_ERR_HIP = -2,           /* a HIP runtime call failed                      */
    BF_ERR_NO_DEVICE = -3,     /* no gfx950 device / extension unusable          */
    BF_ERR_CAPACITY = -4,      /* a fixed-capacity buffer overflowed             */
    BF_ERR_STATE = -5          /* call order violated (e.g. GC before compactify) */
} bf_status;

BF_API const char* bf_last_error(void);
BF_API const char* bf_version(void);
/* number of visible HIP devices, <0 on error (never falls back to a CPU path) */
BF_API int bf_device_count(void);

/* ------------------------------------------------------------------------- */
/* Voxel-hash TSDF:  DepthSensing/CUDASceneRepHashSDF.h                       */
/* ------------------------------------------------------------------------- */

#define BF_SDF_BLOCK_SIZE 8          /* VoxelUtilHashSDF.h:40 */
#define BF_HASH_BUCKET_SIZE 4        /* VoxelUtilHashSDF.h:41 */
#define BF_LOCK_ENTRY (-1)           /* VoxelUtilHashSDF.h:52 */
#define BF_FREE_ENTRY (-2)           /* VoxelUtilHashSDF.h:53 */

/* HashEntry, VoxelUtilHashSDF.h:56-74: __align__(16), 20 B payload => 32 B stride */
typedef struct __attribute__((aligned(16))) bf_hash_entry {
    int32_t pos[3];      /* SDF-block coordinate                                  */
    int32_t ptr;         /* first voxel of the block in d_SDFBlocks, or FREE_ENTRY */
    uint32_t offset;     /* collision chain link, relative to bucket's last slot  */
    uint32_t _pad[3];
} bf_hash_entry;

/* Voxel, VoxelUtilHashSDF.h:77-98: 12 B, AoS */
typedef struct bf_voxel {
    float sdf;
    float weight;
    uint8_t color[4];
} bf_voxel;

/* HashParams, DepthSensing/CUDAHashParams.h:9-39 (same fields, same order) */
typedef struct bf_hash_params {
    float m_rigidTransform[16];
    float m_rigidTransformInverse[16];
    uint32_t m_hashNumBuckets;
    uint32_t m_hashBucketSize;
    uint32_t m_hashMaxCollisionLinkedListSize;
    uint32_t m_numSDFBlocks;
    int32_t m_SDFBlockSize;
    float m_virtualVoxelSize;
    uint32_t m_numOccupiedBlocks;
    float m_maxIntegrationDistance;
    float m_truncScale;
    float m_truncation;
    uint32_t m_integrationWeightSample;
    uint32_t m_integrationWeightMax;
    float m_streamingVoxelExtents[3];
    int32_t m_streamingGridDimensions[3];
    int32_t m_streamingMinGridPos[3];
    uint32_t m_streamingInitialChunkListSize;
    uint32_t m_dummy[2];
} bf_hash_params;

/* DepthCameraParams, DepthSensing/CUDADepthCameraParams.h:7-19 */
typedef struct bf_depth_camera_params {
    float fx, fy, mx, my;
    uint32_t m_imageWidth, m_imageHeight;
    float m_sensorDepthWorldMin, m_sensorDepthWorldMax;
} bf_depth_camera_params;

/* DepthCameraData, DepthSensing/DepthCameraUtil.h:17-29: borrowed device pointers
 * at integration resolution; depth in metres with -inf = invalid, colour RGBX8. */
typedef struct bf_depth_camera_data {
    const float* d_depthData;
    const uint8_t* d_colorData; /* uchar4 per pixel, may be NULL */
} bf_depth_camera_data;

/* HashDataStruct raw pointers, VoxelUtilHashSDF.h:830-838.  d_hashBucketMutex,
 * d_hashDecisionPrefix are NULL: the gfx950 allocator is lock-free.
 * d_hashCompactified holds m_numSDFBlocks entries (it can never need more). */
typedef struct bf_hash_data {
    uint32_t* d_heap;
    uint32_t* d_heapCounter;
    int32_t* d_hashDecision;
    int32_t* d_hashDecisionPrefix;
    bf_hash_entry* d_hash;
    bf_hash_entry* d_hashCompactified;
    int32_t* d_hashCompactifiedCounter;
    bf_voxel* d_SDFBlocks;
    int32_t* d_hashBucketMutex;
} bf_hash_data;

typedef struct bf_scene bf_scene; /* == class CUDASceneRepHashSDF */

/* CUDASceneRepHashSDF(const HashParams&)            CUDASceneRepHashSDF.h:32,317 */
BF_API int bf_scene_create(const bf_hash_params* params, bf_scene** out);
/* ~CUDASceneRepHashSDF                               CUDASceneRepHashSDF.h:35    */
BF_API int bf_scene_destroy(bf_scene* s);
BF_API int bf_scene_set_stream(bf_scene* s, void* hip_stream);
/* reset()                                            CUDASceneRepHashSDF.h:147   */
BF_API int bf_scene_reset(bf_scene* s);
/* integrate(lastRigidTransform, data, params, d_bitMask)   :65-83
 * d_bitMask (streaming) must be NULL: streaming is disabled for BundleFusion
 * (zParametersDefault.txt:100) and incompatible with de-integration (:89-91). */
BF_API int bf_scene_integrate(bf_scene* s, const float cam_to_world[16],
                              const bf_depth_camera_data* data,
                              const bf_depth_camera_params* cam, const uint32_t* d_bitMask);
/* deIntegrate(...)                                         :85-108               */
BF_API int bf_scene_deintegrate(bf_scene* s, const float cam_to_world[16],
                                const bf_depth_camera_data* data,
                                const bf_depth_camera_params* cam, const uint32_t* d_bitMask);
/* garbageCollect()                                         :110-126              */
BF_API int bf_scene_garbage_collect(bf_scene* s);
/* setLastRigidTransformAndCompactify(T)                    :136-139
 * (needs the camera of the following ray cast / GC for the frustum test)         */
BF_API int bf_scene_set_last_rigid_transform_and_compactify(
    bf_scene* s, const float cam_to_world[16], const bf_depth_camera_params* cam);
/* getHashData() / getHashParams()                          :158-164
 * get_hash_params synchronises and refreshes m_numOccupiedBlocks.               */
BF_API int bf_scene_get_hash_data(bf_scene* s, bf_hash_data* out);
BF_API int bf_scene_get_hash_params(bf_scene* s, bf_hash_params* out);
/* getHeapFreeCount()                                       :168-172 (syncs)      */
BF_API int bf_scene_get_heap_free_count(bf_scene* s, uint32_t* out);
/* getNumIntegratedFrames()                                 :174                  */
BF_API int bf_scene_get_num_integrated_frames(bf_scene* s, uint32_t* out);
/* debugHash() invariants                                   :179-314 (syncs).
 * out[0]=#occupied entries out[1]=#free heap out[2]=#duplicate keys
 * out[3]=#entries whose ptr is also on the free heap out[4]=#leaked blocks
 * out[5]=#blocks dropped by the allocator so far (chain window / heap exhausted) */
BF_API int bf_scene_debug_hash(bf_scene* s, uint32_t out[6]);
/* number of allocated SDF blocks (= occupied hash entries), syncs */
BF_API int bf_scene_get_num_allocated_blocks(bf_scene* s, uint32_t* out);
/* Opt-in HIP-event timing of the voxel-update kernel (integrateDepthMapKernel<>'s
 * replacement) on the scene's stream: enable, run ops, then read {#launches, total ms}
 * (read synchronises the stream and clears the accumulator).                       */
BF_API int bf_scene_kernel_timing(bf_scene* s, int enable);
BF_API int bf_scene_kernel_timing_read(bf_scene* s, uint32_t* count, float* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* BF_HIP_H */
