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
/* Plumbing for hosts that hold raw pointers only: copies / a device-wide fence issued by this library's own HIP
 * runtime (a process may hold more than one copy of libamdhip64; work is only ordered within one of them). */
/* Restrict every thread of the calling process to the CPUs of the NUMA node HIP device `device` is attached to (intersected with the
 * caller's current affinity); threads created later inherit it.  cpulist_out (optional) receives the node's CPU list ("0-63,128-191"),
 * empty if nothing was changed (unknown topology, BF_BIND_NUMA=0).  Launch latency from the remote socket costs ~20 % of the frame rate
 * (DESIGN.md 4.4); the library never changes affinities on its own. */
BF_API int bf_bind_host_threads_to_device(int device, char* cpulist_out, size_t cpulist_len);
BF_API int bf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
BF_API int bf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
BF_API int bf_device_synchronize(void);
/* one stream only / a blocking copy whose direction follows from the pointers (hipMemcpyDefault): for all-gather callbacks of a host language (bf_comm.h) */
BF_API int bf_stream_synchronize(void* hip_stream);
BF_API int bf_memcpy(void* dst, const void* src, size_t bytes);

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
/* Software pipelining of consecutive operators: with overlap enabled, the preparation of operators n+1 .. n+3 (allocation: the ray march and the
 * placement, which also builds the operator's block list - two launches) runs on an internal stream while the voxel update of operator n runs on the
 * scene's stream (four list buffers; the update never reads the hash table, allocation never touches voxels).  Results are unchanged.  The ORDER of
 * d_hashCompactified is unspecified (as in the reference, whose compactify appends with atomicAdd, CUDASceneRepHashSDF.cu:324-366): compare it as a
 * set.  The caller then must
 * order its input frames against the scene with bf_scene_wait_event (or have them complete) instead of relying on stream
 * order: bf_scene_wait_event makes the next operator's first kernel wait for `hip_event`.                               */
BF_API int bf_scene_set_overlap(bf_scene* s, int enable);
BF_API int bf_scene_wait_event(bf_scene* s, void* hip_event);
/* Multi-GPU hash-bucket sharding (one process per GPU): the volume owns the home buckets
 * [rank*numBuckets/world, (rank+1)*numBuckets/world) and allocates / integrates / collects only blocks hashing there.
 * Call before the first integrate; every shard is fed every frame and pose, there is no exchange between shards. */
BF_API int bf_scene_set_shard(bf_scene* s, uint32_t rank, uint32_t world);
/* Multi-GPU allocation (with bf_scene_set_shard): the ray march of the allocation (CUDASceneRepHashSDF.cu:165-251) is the part of an operator
 * that does not shrink when the volume is sharded by home bucket - every shard would march every pixel to find its own blocks.  Instead:
 *   rank r:   bf_scene_alloc_collect(part = r, parts = world)  marches a band of the pixel tiles, writes the distinct in-frustum block keys
 *   all ranks exchange the key lists (one all-gather; the lists are opaque 64-bit keys + a count)
 *   every rank: bf_scene_alloc_ingest(list) for each rank's list (ownership, table lookup, de-dup), bf_scene_alloc_place() once,
 *               then bf_scene_integrate / _reintegrate with bf_scene_set_external_alloc(1) (they skip their own allocation).
 * The resulting table is identical to the one the operator's own allocation builds (tests/test_tsdf_gpu.py).  d_slots is scratch of
 * `capacity` words; everything runs on the scene's allocation stream, bf_scene_alloc_sync waits for it. */
BF_API int bf_scene_set_external_alloc(bf_scene* s, int enable);
BF_API int bf_scene_alloc_collect(bf_scene* s, const float cam_to_world[16], const bf_depth_camera_data* data, const bf_depth_camera_params* cam,
                                  uint32_t part, uint32_t parts, uint64_t* d_keys, uint32_t* d_slots, uint32_t* d_count, uint32_t capacity);
BF_API int bf_scene_alloc_ingest(bf_scene* s, const uint64_t* d_keys, const uint32_t* d_count, uint32_t capacity);
BF_API int bf_scene_alloc_place(bf_scene* s);
BF_API int bf_scene_alloc_sync(bf_scene* s);
/* Arithmetic contract of the voxel update, CUDASceneRepHashSDF.cu:425-516 (integrateDepthMapKernel / deIntegrateDepthMapKernel):
 *   BF_TSDF_ARITH_FAST (default)   the contract of the reference's own Release GPU build (FriedLiver.vcxproj:124 <FastMath>true</FastMath>):
 *                                  approximate division (v_rcp_f32), FMA contraction.  Block set, bucket occupancy, heap and voxel
 *                                  weights are the same as in exact mode; sdf within 1e-5 x truncation, colour within 1 LSB, except
 *                                  voxels projecting within 2e-4 pixel (640x480; 6e-4 at 1280x960; measured 1.1e-4 / 4.8e-4) of a pixel
 *                                  boundary (they may sample the neighbouring pixel) - held against the oracle, tests/test_tsdf_fast_gpu.py.
 *                                  This is the path bench.py measures.
 *   BF_TSDF_ARITH_EXACT            every operation as written, IEEE binary32, no contraction - bit-comparable with a host build of the
 *                                  reference (and with oracle/): the mode of every bit-for-bit test.
 * Environment: BF_TSDF_ARITH=fast|exact selects the mode of every scene created afterwards.  May be switched at any time.
 * (Until round 3 exact was the default and the benchmark measured fast; since round 4 the library default is what is measured.) */
#define BF_TSDF_ARITH_EXACT 0
#define BF_TSDF_ARITH_FAST 1
BF_API int bf_scene_set_arith(bf_scene* s, int mode);
/* MI355X addition (fast contract): a sample's depth and colour are gathered as ONE 8-byte texel {depth f32, colour RGBX8}.  Without _set_frame_texels an
 * operator's own ray march writes the texels of its frame on the way (a de-integration, or an operator behind an external / exchanged allocation,
 * interleaves the frame in a launch of its own).  A caller that keeps its frames can interleave each frame once
 * (bf_image_interleave_texels: numPixels x 8 bytes) and hand that image to the NEXT operator on the frame with _set_frame_texels (consumed by one operator). */
BF_API int bf_scene_set_frame_texels(bf_scene* s, const void* d_texels);
BF_API int bf_image_interleave_texels(void* d_texels, const float* d_depth, const uint8_t* d_colorRGBX, uint32_t numPixels, void* hip_stream);
BF_API int bf_scene_get_arith(bf_scene* s, int* mode);
/* MI355X addition: deIntegrate(oldT) + integrate(newT) of the same frame (DepthSensing.cpp:882-889) as ONE pass over
 * the union of the two frustum lists — each touched voxel is read and written once.  Bit-identical to the two calls. */
BF_API int bf_scene_reintegrate(bf_scene* s, const float old_cam_to_world[16], const float new_cam_to_world[16],
                                const bf_depth_camera_data* data, const bf_depth_camera_params* cam);
/* garbageCollect()                                         :110-126              */
/* A batch of operators (MI355X addition).  DepthSensing.cpp:854-902 issues the up to s_maxFrameFixes re-integrations of a frame one after the other, each
 * with its own allocation, compactify and passes over the volume; bf_scene_run_batch takes them together - kind 0: integrate(T0), 1: deIntegrate(T0),
 * 2: deIntegrate(T0) + integrate(T1) of the same frame - and leaves the hash table, the heap and every voxel exactly as the same calls issued in that order
 * would (CUDASceneRepHashSDF.h:65-155), with one ray march over all frames, one placement, one union block list and ONE pass of the voxel update in which
 * every touched block is loaded once, updated by the batch's operators in order and stored once.  `wait_event` (optional hipEvent_t): the operator's frame is
 * complete when the event is; `d_texels` (optional): the frame as interleaved texels (bf_image_interleave_texels).  cam as in bf_scene_integrate. */
#define BF_SCENE_BATCH_MAX 12
typedef struct bf_scene_batch_op {
    int32_t kind;
    int32_t reserved;
    float T0[16], T1[16];
    bf_depth_camera_data data;
    const void* d_texels;
    void* wait_event;
} bf_scene_batch_op;
BF_API int bf_scene_run_batch(bf_scene* s, const bf_scene_batch_op* ops, uint32_t n, const bf_depth_camera_params* cam);
BF_API int bf_scene_garbage_collect(bf_scene* s);
/* setLastRigidTransformAndCompactify(T)                    :136-139
 * (needs the camera of the following ray cast / GC for the frustum test)         */
BF_API int bf_scene_set_last_rigid_transform_and_compactify(
    bf_scene* s, const float cam_to_world[16], const bf_depth_camera_params* cam);
/* setLastRigidTransform(T)                                 :128-134 (host state only: m_rigidTransform and its inverse) */
BF_API int bf_scene_set_last_rigid_transform(bf_scene* s, const float cam_to_world[16]);
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
/* test / tooling aid: d_ptr_out[i] = ptr of block d_pos[3 i .. 3 i + 2] (device arrays) or FREE_ENTRY, behind everything issued so far (synchronises) */
BF_API int bf_scene_debug_find_blocks(bf_scene* s, const int32_t* d_pos, uint32_t n, int32_t* d_ptr_out);
/* number of allocated SDF blocks (= occupied hash entries), syncs */
BF_API int bf_scene_get_num_allocated_blocks(bf_scene* s, uint32_t* out);
/* Opt-in HIP-event timing of the voxel-update kernel (integrateDepthMapKernel<>'s
 * replacement) on the scene's stream: enable, run ops, then read {#launches, total ms}
 * (read synchronises the stream and clears the accumulator).                       */
BF_API int bf_scene_kernel_timing(bf_scene* s, int enable);
BF_API int bf_scene_kernel_timing_read(bf_scene* s, uint32_t* count, float* total_ms);
/* over the launches timed since bf_scene_kernel_timing(s, 1) (call before _read): the sum of the frustum-list lengths
 * (m_numOccupiedBlocks) of the integrate / de-integrate operations they performed, and the number of those operations
 * (a fused re-integration launch performs two).                                                                     */
BF_API int bf_scene_kernel_timing_occupied(bf_scene* s, uint64_t* sumOccupiedBlocks, uint32_t* numOps);
/* the same plus the sum of the list lengths per LAUNCH, separately for the plain (integrate / de-integrate) and the fused
 * re-integration kernel: a fused launch visits the union of its two frustum lists once (each voxel read and written once), so its
 * algorithmic traffic is unionBlocks * (512*24 + 32) B, not 2 B                                                              */
/* frames (depth + colour images) the timed launches sampled: one per operator, n per batch of n */
BF_API int bf_scene_kernel_timing_images(bf_scene* s, uint32_t* numImages);
BF_API int bf_scene_kernel_timing_blocks(bf_scene* s, uint64_t* sumOperatorBlocks, uint64_t* visitedPlain, uint64_t* visitedFused, uint32_t* numOps);


/* ------------------------------------------------------------------------- */
/* Dense frame cache:  CUDACache.h / CUDACacheUtil.h                          */
/* ------------------------------------------------------------------------- */

/* CUDACachedFrame, CUDACacheUtil.h:10-53 (CUDACACHE_UCHAR_NORMALS and _FLOAT_NORMALS both on) */
typedef struct bf_cached_frame {
    float* d_depthDownsampled;            /* W*H                                  */
    float* d_cameraposDownsampled;        /* float4 per pixel (x,y,z,1) or -inf   */
    float* d_intensityDownsampled;        /* W*H                                  */
    float* d_intensityDerivsDownsampled;  /* float2 per pixel                     */
    uint8_t* d_normalsDownsampledUCHAR4;  /* uchar4 per pixel                     */
    float* d_normalsDownsampled;          /* float4 per pixel (nx,ny,nz,0) or -inf */
} bf_cached_frame;

typedef struct bf_cache bf_cache; /* == class CUDACache */

/* CUDACache(widthDepthInput, heightDepthInput, widthDownSampled, heightDownSampled, maxNumImages,
 *           inputIntrinsics)  CUDACache.cpp:14-43; the three filter sigmas are
 * GlobalBundlingState s_colorDownSigma / s_depthDownSigmaD / s_depthDownSigmaR.               */
BF_API int bf_cache_create(uint32_t widthDepthInput, uint32_t heightDepthInput, uint32_t widthDownSampled,
                           uint32_t heightDownSampled, uint32_t maxNumImages, const float inputIntrinsics[16],
                           float colorDownSigma, float depthDownSigmaD, float depthDownSigmaR, bf_cache** out);
BF_API int bf_cache_destroy(bf_cache* c);
BF_API int bf_cache_set_stream(bf_cache* c, void* hip_stream);
/* storeFrame(d_depth, w, h, d_color, cw, ch)                  CUDACache.cpp:45-86 */
BF_API int bf_cache_store_frame(bf_cache* c, const float* d_depth, uint32_t inputDepthWidth, uint32_t inputDepthHeight,
                                const uint8_t* d_color, uint32_t inputColorWidth, uint32_t inputColorHeight);
BF_API int bf_cache_reset(bf_cache* c);                                   /* CUDACache.h:17 */
BF_API int bf_cache_copy_cache_frame_from(bf_cache* c, bf_cache* other, uint32_t frameFrom); /* :24 */
BF_API int bf_cache_increment(bf_cache* c);                               /* incrementCache :43 */
BF_API int bf_cache_get_num_frames(bf_cache* c, uint32_t* out);
BF_API int bf_cache_set_current_frame(bf_cache* c, uint32_t n);           /* setCurrentFrame (debug) */
/* getCacheFramesGPU(): device array of maxNumImages bf_cached_frame   CUDACache.h:22 */
BF_API int bf_cache_get_frames_gpu(bf_cache* c, const bf_cached_frame** d_frames);
/* host copy of one frame's pointer struct (getCacheFrames()[i]) */
BF_API int bf_cache_get_frame(bf_cache* c, uint32_t i, bf_cached_frame* out);
/* getWidth/getHeight/getIntrinsics: out = {fx, fy, cx, cy} of the down-sampled camera */
BF_API int bf_cache_get_geometry(bf_cache* c, uint32_t* width, uint32_t* height, float intrinsics4[4]);

/* ------------------------------------------------------------------------- */
/* Bundling solver:  Solver/CUDASolverBundling.h, SBA.cu                      */
/* ------------------------------------------------------------------------- */

/* EntryJ, SiftGPU/SIFTImageManager.h:45-60 (32 B) */
typedef struct bf_entry_j {
    uint32_t imgIdx_i, imgIdx_j;   /* 0xFFFFFFFF in imgIdx_i == invalid */
    float pos_i[3];
    float pos_j[3];
} bf_entry_j;

/* thresholds CUDASolverBundling reads from GlobalBundlingState (.cpp:35-36, :93-100) */
typedef struct bf_solver_config {
    float optMaxResThresh;          /* s_optMaxResThresh              */
    float denseDistThresh;          /* s_denseDistThresh              */
    float denseNormalThresh;        /* s_denseNormalThresh            */
    float denseColorThresh;         /* s_denseColorThresh             */
    float denseColorGradientMin;    /* s_denseColorGradientMin        */
    float denseDepthMin;            /* s_denseDepthMin                */
    float denseDepthMax;            /* s_denseDepthMax                */
    uint32_t denseOverlapCheckSubsampleFactor;
    float verifyOptDistThresh;      /* 0.02 (.cpp:35)                 */
    float verifyOptPercentThresh;   /* 0.05 (.cpp:36)                 */
    int32_t recordConvergence;      /* s_recordSolverConvergence      */
} bf_solver_config;

typedef struct bf_solver bf_solver; /* == class CUDASolverBundling */

/* CUDASolverBundling(maxNumberOfImages, maxNumResiduals)      CUDASolverBundling.cpp:24 */
BF_API int bf_solver_create(uint32_t maxNumberOfImages, uint32_t maxNumResiduals, const bf_solver_config* cfg,
                            bf_solver** out);
BF_API int bf_solver_destroy(bf_solver* s);
BF_API int bf_solver_set_stream(bf_solver* s, void* hip_stream);
/* solve(...)                                                   CUDASolverBundling.cpp:187-284
 * d_rot / d_trans: float3 per image (se(3) unknowns), updated in place.  d_cacheFrames may be
 * NULL (dense term off).  weights*: numWeights host floats, one per non-linear iteration.
 * Asynchronous unless findMaxResidual/recordConvergence need host values (one sync at the end). */
BF_API int bf_solver_solve(bf_solver* s, bf_entry_j* d_correspondences, uint32_t numberOfCorrespondences,
                           const int32_t* d_validImages, uint32_t numberOfImages, uint32_t nNonLinearIterations,
                           uint32_t nLinearIterations, const bf_cached_frame* d_cacheFrames, uint32_t cacheWidth,
                           uint32_t cacheHeight, const float cacheIntrinsics4[4], const float* weightsSparse,
                           const float* weightsDenseDepth, const float* weightsDenseColor, uint32_t numWeights,
                           int usePairwiseDense, float* d_rot, float* d_trans, int rebuildJT, int findMaxResidual,
                           uint32_t revalidateIdx);
/* getVarToCorrNumEntriesPerRow(): device int[numberOfImages], #valid correspondences touching each image at the
 * last solve                                                    CUDASolverBundling.h:48 */
BF_API int bf_solver_get_var_to_corr_num_entries_per_row(bf_solver* s, const int32_t** d_out);
/* m_maxCorrPerImage (CUDASolverBundling.cpp:39): the reference invalidates correspondences beyond this many per image in atomic arrival
 * order (.cpp:195-199); this solver uses all of them.  *numImagesOverLimit = images of the last solve whose count exceeds the limit. */
BF_API int bf_solver_get_corr_overflow(bf_solver* s, uint32_t* numImagesOverLimit, uint32_t* limit);
/* getMaxResidual(max, index)                                   CUDASolverBundling.h:37-40 */
BF_API int bf_solver_get_max_residual(bf_solver* s, float* max, int32_t* index);
/* getMaxResidual(curFrame, d_corr, imageIndices, maxRes) -> remove?   .cpp:429-452 */
BF_API int bf_solver_get_max_residual_pair(bf_solver* s, uint32_t curFrame, const bf_entry_j* d_correspondences,
                                           uint32_t imageIndices[2], float* maxRes, int* remove);
/* useVerification(d_corr, n)                                   .cpp:454-476 (syncs) */
BF_API int bf_solver_use_verification(bf_solver* s, const bf_entry_j* d_correspondences, uint32_t numberOfCorrespondences,
                                      int* out);
/* getConvergenceAnalysis(): energies per GN iteration of the last solve (needs recordConvergence) */
BF_API int bf_solver_get_convergence(bf_solver* s, float* out, uint32_t capacity, uint32_t* count);
/* diagnostics of the last solve (syncs): out[0]=#GN iterations run, out[1..]=#PCG iterations of each */
BF_API int bf_solver_get_iteration_counts(bf_solver* s, int32_t* out, uint32_t capacity);
/* dump of the last dense system in the reference's layout (6N x 6N row-major JtJ, 6N Jtr);
 * test hook, syncs.  numPairs = #overlapping image pairs found.                              */
BF_API int bf_solver_debug_dense_system(bf_solver* s, float* h_JtJ, float* h_Jtr, uint32_t numImages, int32_t* numPairs);

/* convertMatricesToPosesCU / convertPosesToMatricesCU          SBA.cu:75-108 (Lie space) */
BF_API int bf_convert_matrices_to_poses(const float* d_transforms, uint32_t numTransforms, float* d_rot, float* d_trans,
                                        const int32_t* d_validImages, void* hip_stream);
BF_API int bf_convert_poses_to_matrices(const float* d_rot, const float* d_trans, uint32_t numImages, float* d_transforms,
                                        const int32_t* d_validImages, void* hip_stream);

/* ------------------------------------------------------------------------- */
/* SIFT detection:  SiftGPU/SiftGPU.h (detection half of the fork)            */
/* ------------------------------------------------------------------------- */

/* SIFTKeyPoint, SiftGPU/SIFTImageManager.h:20-24 (16 B); SIFTKeyPointDesc :26-28 (128 B) */
typedef struct bf_sift_keypoint { float pos[2]; float scale; float depth; } bf_sift_keypoint;
typedef struct bf_sift_keypoint_desc { uint8_t feature[128]; } bf_sift_keypoint_desc;

typedef struct bf_sift bf_sift; /* == class SiftGPU (+ SiftPyramid) */

/* SiftGPU::SetParams(siftWidth, siftHeight, enableTiming, featureCountThreshold, siftDepthMin, siftDepthMax)
 * + InitSiftGPU  (SiftGPU.cpp:224-253; Bundler.cpp:57-62 passes threshold 150 and the sensor depth range);
 * depthWidth/Height and minKeyScale are the SiftCameraParams fields the kernels read
 * (SiftCameraParams.h, OnlineBundler.cpp:46-56).                                                      */
BF_API int bf_sift_create(uint32_t siftWidth, uint32_t siftHeight, uint32_t depthWidth, uint32_t depthHeight,
                          uint32_t featureCountThreshold, float siftDepthMin, float siftDepthMax, float minKeyScale,
                          uint32_t maxNumKeysPerImage, bf_sift** out);
BF_API int bf_sift_destroy(bf_sift* s);
BF_API int bf_sift_set_stream(bf_sift* s, void* hip_stream);
/* RunSIFT(d_intensity, d_depth) + GetKeyPointsAndDescriptorsCUDA(image, d_depth, max)  SiftGPU.cpp:72-101,267-272.
 * Asynchronous; *d_numKeys (device int) receives the feature count, or -1 for "too many keypoints".  */
BF_API int bf_sift_run(bf_sift* s, const float* d_intensity, const float* d_depth, float* d_keyPoints, uint8_t* d_descs,
                       int32_t* d_numKeys);
BF_API int bf_sift_debug_level(bf_sift* s, uint32_t octave, uint32_t index, float* h_out);
BF_API int bf_sift_debug_counts(bf_sift* s, int32_t out[26]);


/* ------------------------------------------------------------------------- */
/* Keypoint / match store, matcher, match filters:                            */
/*   SiftGPU/SIFTImageManager.h, SiftGPU/SiftMatch.h                          */
/* ------------------------------------------------------------------------- */

/* SIFTImageGPU, SIFTImageManager.h:32-36.  d_numKeyPoints is this library's addition: the key count of an
 * image stays on the device (bf_sift_run writes it) so detection -> matching needs no host read-back.
 * Keys of image i start at element i * maxKeyPointsPerImage (fixed stride instead of the reference's packed
 * prefix-sum layout); key indices in matches / correspondences are indices into that array.            */
typedef struct bf_sift_image_gpu {
    bf_sift_keypoint* d_keyPoints;
    bf_sift_keypoint_desc* d_keyPointDescs;
    int32_t* d_numKeyPoints;
} bf_sift_image_gpu;

typedef struct bf_siftmgr bf_siftmgr; /* == class SIFTImageManager (+ the SiftMatchGPU it feeds) */

/* SIFTImageManager(maxImages, maxKeyPointsPerImage)            SIFTImageManager.cpp:7-15, alloc :274-313 */
BF_API int bf_siftmgr_create(uint32_t maxImages, uint32_t maxKeyPointsPerImage, bf_siftmgr** out);
BF_API int bf_siftmgr_destroy(bf_siftmgr* m);
BF_API int bf_siftmgr_set_stream(bf_siftmgr* m, void* hip_stream);
BF_API int bf_siftmgr_reset(bf_siftmgr* m);                                      /* reset()  .h:112-124 */
/* createSIFTImageGPU / finalizeSIFTImageGPU(numKeyPoints)     .cpp:44-75.  numKeyPoints < 0: keep the count the
 * detector wrote to d_numKeyPoints.                                                                   */
BF_API int bf_siftmgr_create_image(bf_siftmgr* m, bf_sift_image_gpu* out);
BF_API int bf_siftmgr_finalize_image(bf_siftmgr* m, int32_t numKeyPoints);
BF_API int bf_siftmgr_get_image(bf_siftmgr* m, uint32_t imageIdx, bf_sift_image_gpu* out);          /* getImageGPU */
BF_API int bf_siftmgr_get_num_images(bf_siftmgr* m, uint32_t* out);
BF_API int bf_siftmgr_get_max_num_keypoints_per_image(bf_siftmgr* m, uint32_t* out);
BF_API int bf_siftmgr_get_current_frame(bf_siftmgr* m, uint32_t* out);
BF_API int bf_siftmgr_set_current_frame(bf_siftmgr* m, uint32_t idx);
/* getNumKeyPointsPerImage for images [first, first+count) — D2H + sync */
BF_API int bf_siftmgr_get_num_keypoints(bf_siftmgr* m, uint32_t first, uint32_t count, int32_t* h_out);

/* The per-pair loop of Bundler::matchAndFilter (Bundler.cpp:117-136): SiftMatchGPU::GetSiftMatch
 * (SiftMatch.cpp:160-196, ProgramCU.cu:1634-1936) of image curFrame against every image in
 * [startFrame, numFrames) \ {curFrame}, followed by SortKeyPointMatchesCU (SIFTImageManager.cu:59-143).
 * Pairs whose previous image is invalid or has no keys get 0 matches.  One launch, asynchronous.        */
BF_API int bf_siftmgr_match(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, float distMax,
                            float ratioMax);
/* FilterKeyPointMatchesCU                                      SIFTImageManager.cu:186-316 */
BF_API int bf_siftmgr_filter_keypoint_matches(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames,
                                              const float siftIntrinsicsInv[16], uint32_t minNumMatches,
                                              float maxKabschRes2);
/* FilterMatchesBySurfaceAreaCU                                 :318-416 */
BF_API int bf_siftmgr_filter_matches_by_surface_area(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame,
                                                     uint32_t numFrames, const float colorIntrinsicsInv[16],
                                                     float areaThresh);
/* FilterMatchesByDenseVerifyCU                                 :418-608 */
BF_API int bf_siftmgr_filter_matches_by_dense_verify(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame,
                                                     uint32_t numFrames, uint32_t imageWidth, uint32_t imageHeight,
                                                     const float intrinsics[16], const bf_cached_frame* d_cachedFrames,
                                                     float distThresh, float normalThresh, float colorThresh,
                                                     float errThresh, float corrThresh, float sensorDepthMin,
                                                     float sensorDepthMax);
/* MI355X addition: the per-pair result arrays (d_currNumMatchesPerImagePair ... d_currFilteredTransformsInv, SIFTImageManager.h:262-276) exist twice.
 * _set_pair_stage selects the set (0 / 1) and the stream (null: the manager's) the pair kernels above are issued on from now on, so that the pair
 * kernels of frame k + 1 can run beside those of frame k; speculative != 0: match does not consult the valid flags of the previous images (they may
 * still be in the making on the manager's stream) and _commit_pairs - on the manager's stream, once those flags are final - clears the pairs
 * Bundler::matchAndFilter would not have matched (Bundler.cpp:126-129).  Ordering between the streams is the caller's.  The accessors of the per-pair
 * arrays name the selected set.  Default (0, null, 0): the reference's behaviour. */
BF_API int bf_siftmgr_set_pair_stage(bf_siftmgr* m, uint32_t set, void* hip_stream, int speculative);
BF_API int bf_siftmgr_commit_pairs(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames);
/* filterFrames                                                 SIFTImageManager.cpp:551-575 (syncs, returns the
 * last matched frame or 0xFFFFFFFF).  The _async form only enqueues the decision; it is read back together with
 * the residual count by bf_siftmgr_sync_frame_result (one D2H per frame).                              */
BF_API int bf_siftmgr_filter_frames(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames,
                                    uint32_t* lastMatchedFrame);
BF_API int bf_siftmgr_filter_frames_async(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames);
/* AddCurrToResidualsCU                                         SIFTImageManager.cu:610-690.  Skipped on the device
 * when curFrame was not connected (Bundler.cpp:218-219); pairs are appended in ascending previous-image order. */
BF_API int bf_siftmgr_add_curr_to_residuals(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames,
                                            const float colorIntrinsicsInv[16]);
BF_API int bf_siftmgr_sync_frame_result(bf_siftmgr* m, uint32_t curFrame, uint32_t* lastMatchedFrame,
                                        int32_t* numKeyPointsCur);
/* optional: enqueue that read-back now (after filter_frames_async / add_curr_to_residuals); sync_frame_result then only waits - for an
 * event behind the copy, not for the stream: up to two read-backs may be in flight, so the next frame's chain can be enqueued on the same
 * stream before the previous frame's result is consumed (results are consumed oldest first). */
BF_API int bf_siftmgr_prefetch_frame_result(bf_siftmgr* m);
/* the device record behind the read-back, for kernels that act on a frame's result without a host round trip:
 * 4 x int32 {lastMatched (-1: none), valid, numResiduals, numKeysCur}, rewritten by every filter_frames_async / add_curr_to_residuals */
BF_API int bf_siftmgr_get_frame_result_gpu(bf_siftmgr* m, const int32_t** d_frameResult);
/* InvalidateImageToImageCU / CheckForInvalidFrames[Simple]CU   :692-795 */
BF_API int bf_siftmgr_invalidate_image_to_image(bf_siftmgr* m, uint32_t imageIdx_i, uint32_t imageIdx_j);
BF_API int bf_siftmgr_check_for_invalid_frames_simple(bf_siftmgr* m, const int32_t* d_varToCorrNumEntriesPerRow,
                                                      uint32_t numVars);
BF_API int bf_siftmgr_check_for_invalid_frames(bf_siftmgr* m, const int32_t* d_varToCorrNumEntriesPerRow,
                                               uint32_t numVars);
/* VerifyTrajectoryCU                                           :1036-1159 */
BF_API int bf_siftmgr_verify_trajectory(bf_siftmgr* m, uint32_t numImages, const float* d_trajectory, uint32_t imageWidth,
                                        uint32_t imageHeight, const float intrinsics[16],
                                        const bf_cached_frame* d_cachedFrames, float distThresh, float normalThresh,
                                        float colorThresh, float errThresh, float corrThresh, float sensorDepthMin,
                                        float sensorDepthMax, int32_t* valid);
/* getValidImages / invalidateFrame / updateGPUValidImages / getValidImagesGPU   .h:157-163 */
BF_API int bf_siftmgr_get_valid_images(bf_siftmgr* m, int32_t* h_out, uint32_t count);
BF_API int bf_siftmgr_set_valid_image(bf_siftmgr* m, uint32_t frame, int32_t valid);
BF_API int bf_siftmgr_update_gpu_valid_images(bf_siftmgr* m);
BF_API int bf_siftmgr_get_valid_images_gpu(bf_siftmgr* m, const int32_t** d_out);
/* getGlobalCorrespondencesGPU / getNumGlobalCorrespondences / setGlobalCorrespondencesDEBUG   .h:182-188,251-253 */
BF_API int bf_siftmgr_get_global_correspondences_gpu(bf_siftmgr* m, bf_entry_j** d_out);
BF_API int bf_siftmgr_get_global_correspondence_keys_gpu(bf_siftmgr* m, const uint32_t** d_out /* uint2 per entry */);
BF_API int bf_siftmgr_get_num_global_correspondences(bf_siftmgr* m, uint32_t* out);
BF_API int bf_siftmgr_set_global_correspondences(bf_siftmgr* m, const bf_entry_j* h_corr, uint32_t n);
/* getFiltTransformsToWorldGPU (= d_currFilteredTransformsInv) / getNumFiltMatchesGPU   .h:254-255 */
BF_API int bf_siftmgr_get_filt_transforms_gpu(bf_siftmgr* m, const float** d_transforms, const float** d_transformsInv);
BF_API int bf_siftmgr_get_num_filt_matches_gpu(bf_siftmgr* m, const int32_t** d_out);
BF_API int bf_siftmgr_get_keys_gpu(bf_siftmgr* m, const bf_sift_keypoint** d_keys, const bf_sift_keypoint_desc** d_descs,
                                   const int32_t** d_numKeys);
#define BF_MAX_MATCHES_PER_IMAGE_PAIR_RAW 128u        /* GlobalDefines.h:8 */
#define BF_MAX_MATCHES_PER_IMAGE_PAIR_FILTERED 25u    /* GlobalDefines.h:9 */
/* getCurrMatchKeyPointIndicesDEBUG without the copy (SIFTImageManager.h:236-243): device views of the current frame's match lists -
 * uint2 key indices [maxImages][128] (raw) or [maxImages][25] (filtered) and the match count per previous image.  A key index is
 * image * maxNumKeyPointsPerImage + key (fixed stride; the reference packs the key points of all images). */
BF_API int bf_siftmgr_get_curr_matches_gpu(bf_siftmgr* m, int filtered, const uint32_t** d_keyPointIndices, const int32_t** d_numMatches);
/* get{Raw,Filt}KeyPointIndicesAndMatchDistancesDEBUG           .h:212-235 (128 / 25 slots are copied) */
BF_API int bf_siftmgr_get_raw_matches(bf_siftmgr* m, uint32_t imagePairIndex, int32_t* numMatches,
                                      uint32_t* h_keyPointIndices, float* h_distances);
BF_API int bf_siftmgr_get_filt_matches(bf_siftmgr* m, uint32_t imagePairIndex, int32_t* numMatches,
                                       uint32_t* h_keyPointIndices, float* h_distances, float* h_transform,
                                       float* h_transformInv);
/* addToRetryList / getTopRetryImage                            .h:263-271 */
BF_API int bf_siftmgr_add_to_retry_list(bf_siftmgr* m, uint32_t idx);
BF_API int bf_siftmgr_get_top_retry_image(bf_siftmgr* m, uint32_t* idx, int* found);
/* fuseToGlobal(global, colorIntrinsics, d_transforms, colorIntrinsicsInv)   SIFTImageManager.cpp:367-476
 * Runs on the device (track building by connected components + the reference's depth-first order per component; the new key frame
 * and its key count are written in HBM, nothing is copied to the host); asynchronous on the manager's stream.
 * bf_siftmgr_fuse_to_global_host is the reference's own form (all key points / descriptors / correspondences to the host, recursive
 * search there, upload) - same results bit for bit.
 * bf_siftmgr_fuse_error: capacity conditions of the device search since creation (none at present: always 0). */
BF_API int bf_siftmgr_fuse_to_global(bf_siftmgr* local, bf_siftmgr* global, const float colorIntrinsics[16],
                                     const float* d_transforms, const float colorIntrinsicsInv[16]);
BF_API int bf_siftmgr_fuse_to_global_host(bf_siftmgr* local, bf_siftmgr* global, const float colorIntrinsics[16],
                                          const float* d_transforms, const float colorIntrinsicsInv[16]);
BF_API int bf_siftmgr_fuse_error(bf_siftmgr* local, int* err);


/* ------------------------------------------------------------------------- */
/* Frame-ingest image operators:  CUDAImageUtil.h / CUDAImageUtil.cu          */
/* (device pointers, asynchronous on hip_stream)                              */
/* ------------------------------------------------------------------------- */
/* erodeDepthMap(d_output, d_input, structureSize, w, h, dThresh, fracReq)     CUDAImageUtil.cu:701-757 */
BF_API int bf_image_erode_depth_map(float* d_output, const float* d_input, int structureSize, uint32_t width, uint32_t height,
                                    float dThresh, float fracReq, void* hip_stream);
/* gaussFilterDepthMap(d_output, d_input, sigmaD, sigmaR, w, h)                :759-809 */
BF_API int bf_image_gauss_filter_depth_map(float* d_output, const float* d_input, float sigmaD, float sigmaR, uint32_t width,
                                           uint32_t height, void* hip_stream);
/* fused forms for an ingest whose frame already lives in device memory: the first erosion carries the colour image's copies along, the depth filter writes
 * the stored frame as well (three launches instead of seven; same images bit for bit) */
BF_API int bf_image_erode_depth_map_and_copy(float* d_output, const float* d_input, int structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq,
                                             const void* d_copySrc, void* d_copyDst1, void* d_copyDst2, void* stream);
BF_API int bf_image_gauss_filter_depth_map2(float* d_output, float* d_output2, const float* d_input, float sigmaD, float sigmaR, uint32_t width, uint32_t height, void* stream);
/* gaussFilterIntensity(d_output, d_input, sigmaD, w, h)                       :811-859 */
BF_API int bf_image_gauss_filter_intensity(float* d_output, const float* d_input, float sigmaD, uint32_t width, uint32_t height,
                                           void* hip_stream);
/* resampleFloat / resampleUCHAR4 / resampleToIntensity                        :93-124, :160-191, :201-258 */
BF_API int bf_image_resample_float(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const float* d_input,
                                   uint32_t inputWidth, uint32_t inputHeight, void* hip_stream);
BF_API int bf_image_resample_uchar4(uint8_t* d_output, uint32_t outputWidth, uint32_t outputHeight, const uint8_t* d_input,
                                    uint32_t inputWidth, uint32_t inputHeight, void* hip_stream);
BF_API int bf_image_resample_to_intensity(float* d_output, uint32_t outputWidth, uint32_t outputHeight, const uint8_t* d_input,
                                          uint32_t inputWidth, uint32_t inputHeight, void* hip_stream);

/* ------------------------------------------------------------------------- */
/* Marching cubes over the voxel hash:                                         */
/* DepthSensing/CUDAMarchingCubesHashSDF.h:8-58, .cpp, CUDAMarchingCubesSDF.cu, */
/* MarchingCubesSDFUtil.h:9-287 (a consumer of getHashData(): it reads the raw  */
/* arrays in the reference layout)                                             */
/* ------------------------------------------------------------------------- */
typedef struct bf_mc_vertex { float p[3]; float c[3]; } bf_mc_vertex;              /* MarchingCubesData::Vertex  :28-32 */
typedef struct bf_mc_triangle { bf_mc_vertex v[3]; } bf_mc_triangle;               /* MarchingCubesData::Triangle :34-39 (72 bytes) */
typedef struct bf_marching_cubes_params {                                          /* MarchingCubesParams :9-22, parametersFromGlobalAppState (.h:20-29) */
    uint32_t m_maxNumTriangles;          /* s_marchingCubesMaxNumTriangles */
    uint32_t m_sdfBlockSize, m_hashNumBuckets, m_hashBucketSize;
    float m_threshMarchingCubes;         /* s_SDFMarchingCubeThreshFactor * s_SDFVoxelSize */
    float m_threshMarchingCubes2;
} bf_marching_cubes_params;
typedef struct bf_marching_cubes bf_marching_cubes;   /* == class CUDAMarchingCubesHashSDF */

BF_API int bf_marching_cubes_create(const bf_marching_cubes_params* p, bf_marching_cubes** out);
BF_API int bf_marching_cubes_destroy(bf_marching_cubes* m);
BF_API int bf_marching_cubes_set_stream(bf_marching_cubes* m, void* hip_stream);
/* extractIsoSurface(hashData, hashParams, rayCastData, minCorner, maxCorner, boxEnabled)  .cpp:107-119; the triangles are also appended
 * to the host mesh buffer (copyTrianglesToCPU :27-46).  Triangle order is deterministic: (hash slot, voxel index, table order). */
BF_API int bf_marching_cubes_extract(bf_marching_cubes* m, const bf_hash_data* hashData, const bf_hash_params* hashParams,
                                     const float minCorner[3], const float maxCorner[3], int boxEnabled);
/* device buffer of the last extraction: *numTriangles <= m_maxNumTriangles were written, *numFound is the number the volume holds */
BF_API int bf_marching_cubes_get_triangles_gpu(bf_marching_cubes* m, const bf_mc_triangle** d_triangles, uint32_t* numTriangles, uint32_t* numFound);
BF_API int bf_marching_cubes_get_mesh(bf_marching_cubes* m, bf_mc_triangle* h_out, uint32_t capacity, uint32_t* count);   /* m_meshData */
BF_API int bf_marching_cubes_clear_mesh_buffer(bf_marching_cubes* m);                                                     /* clearMeshBuffer */
/* saveMesh(filename, transform)  .cpp:48-105: merge vertices closer than 1e-5, drop duplicate / degenerate faces, binary PLY */
BF_API int bf_marching_cubes_save_mesh(bf_marching_cubes* m, const char* filename, const float transform[16], uint32_t* numVertices, uint32_t* numFaces);
/* the generated case tables (edge mask per case, up to 5 triangles of edge indices, -1 terminated), for inspection / tests */
BF_API int bf_marching_cubes_tables(uint16_t edgeTable[256], int8_t triTable[256 * 16]);

/* ------------------------------------------------------------------------- */
/* Ray cast of the volume: class CUDARayCastSDF (DepthSensing/CUDARayCastSDF.h:14-101, .cpp, .cu),        */
/* RayCastSDFUtil.h, DX11RayIntervalSplatting (the rasterised interval splat is a compute pass here)      */
/* ------------------------------------------------------------------------- */
typedef struct bf_ray_cast_params {        /* struct RayCastParams, CUDARayCastParams.h:7-27 (same field order) */
    float m_viewMatrix[16];
    float m_viewMatrixInverse[16];
    float mx, my, fx, fy;                  /* ray cast intrinsics */
    uint32_t m_width, m_height;
    uint32_t m_numOccupiedSDFBlocks;
    uint32_t m_maxNumVertices;             /* kept for the capacity check of rayIntervalSplatting (6 vertices per block) */
    int32_t m_splatMinimum;                /* unused: both intervals are splatted in one pass */
    float m_minDepth, m_maxDepth;
    float m_rayIncrement, m_thresSampleDist, m_thresDist;
    int32_t m_useGradients;
    uint32_t dummy0;
} bf_ray_cast_params;

typedef struct bf_ray_cast_data {          /* struct RayCastData, RayCastSDFUtil.h:33-303: device images of m_width x m_height */
    float* d_depth;                        /* depth along the camera z axis, -inf where no surface was hit */
    float* d_depth4;                       /* float4 camera-space point (x, y, z, 1) or -inf */
    float* d_normals;                      /* float4 camera-space normal (w = 1) or -inf */
    float* d_colors;                       /* float4 rgb in [0,1] (w = 1) or -inf */
    float* d_rayIntervalSplatMin;          /* per-pixel entry / exit depth of the allocated blocks (the two D3D11 render targets), -inf = no block */
    float* d_rayIntervalSplatMax;
} bf_ray_cast_data;

typedef struct bf_ray_cast bf_ray_cast;    /* == class CUDARayCastSDF */
BF_API int bf_ray_cast_create(const bf_ray_cast_params* params, bf_ray_cast** out);          /* CUDARayCastSDF(params) */
BF_API int bf_ray_cast_destroy(bf_ray_cast* rc);
BF_API int bf_ray_cast_set_stream(bf_ray_cast* rc, void* hip_stream);
/* render(hashData, hashParams, lastRigidTransform)  .cpp:42-72: hashData / hashParams as bf_scene_get_hash_data / _get_hash_params
 * return them after an integrate or setLastRigidTransformAndCompactify (frustum list + m_numOccupiedBlocks + m_rigidTransformInverse);
 * depthCamera = the DepthCameraParams the volume was compactified with (c_depthCameraParams of the splat's frustum test).  */
BF_API int bf_ray_cast_render(bf_ray_cast* rc, const bf_hash_data* hashData, const bf_hash_params* hashParams,
                              const bf_depth_camera_params* depthCamera, const float lastRigidTransform[16]);
BF_API int bf_ray_cast_get_data(bf_ray_cast* rc, bf_ray_cast_data* out);                      /* getRayCastData */
BF_API int bf_ray_cast_get_params(bf_ray_cast* rc, bf_ray_cast_params* out);                  /* getRayCastParams */
BF_API int bf_ray_cast_update_min_max(bf_ray_cast* rc, float depthMin, float depthMax);       /* updateRayCastMinMax */
BF_API int bf_ray_cast_set_intrinsics(bf_ray_cast* rc, uint32_t width, uint32_t height, const float intrinsics[16]);   /* setRayCastIntrinsics */
BF_API int bf_ray_cast_convert_to_camera_space(bf_ray_cast* rc, const bf_depth_camera_params* depthCamera);            /* convertToCameraSpace */

#ifdef __cplusplus
}
#endif
#endif /* BF_HIP_H */
