/*
 * bf_pipeline.h — C ABI of the host-level operators of the BundleFusion hot path on MI355X:
 * the parameter files, CUDAImageManager, Bundler, OnlineBundler, TrajectoryManager and the
 * headless per-frame loop.  Kernel-level entry points live in bf_hip.h.
 *
 * Every entry cites the reference interface it replaces (paths relative to
 * /root/reference/FriedLiver/Source).  Pointers named d_* are device pointers on the GPU the object was
 * created on, h_* host pointers; 4x4 matrices are 16 floats, row-major (mat4f / float4x4).
 * All functions return BF_OK (0) or a bf_status error; bf_last_error() holds the message
 * (the reference throws MLIB_EXCEPTION in those places).
 */
#ifndef BF_PIPELINE_H
#define BF_PIPELINE_H

#include "bf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Parameter files:  GlobalAppState.h:24-104, GlobalBundlingState.h:15-65     */
/* (name = value; lines, // comments; parsed like ml::ParameterFile)          */
/* ------------------------------------------------------------------------- */

/* the GlobalAppState fields the hot path reads (rendering / recording / streaming keys are accepted and ignored) */
typedef struct bf_global_app_state {
    uint32_t s_sensorIdx;
    uint32_t s_integrationWidth, s_integrationHeight;
    uint32_t s_maxFrameFixes, s_topNActive;
    float s_minPoseDistSqrt;
    float s_sensorDepthMax, s_sensorDepthMin;
    float s_renderDepthMax, s_renderDepthMin;
    uint32_t s_hashNumBuckets, s_hashNumSDFBlocks, s_hashMaxCollisionLinkedListSize;
    float s_SDFVoxelSize, s_SDFTruncation, s_SDFTruncationScale, s_SDFMaxIntegrationDistance;
    uint32_t s_SDFIntegrationWeightSample, s_SDFIntegrationWeightMax;
    float s_colorSigmaD, s_colorSigmaR;
    int32_t s_colorFilter;
    int32_t s_integrationEnabled, s_garbageCollectionEnabled, s_reconstructionEnabled, s_streamingEnabled;
    int32_t s_bUseCameraCalibration, s_binaryDumpSensorUseTrajectory;
    uint32_t s_garbageCollectionStarve;
    float s_streamingVoxelExtents[3];
    int32_t s_streamingGridDimensions[3], s_streamingMinGridPos[3];
    uint32_t s_streamingInitialChunkListSize;
    uint32_t s_numSolveFramesBeforeExit;
    char s_binaryDumpSensorFile[512];      /* the .sens file played when s_sensorIdx == 8 (SensorDataReader.cpp:44) */
    /* ray cast (CUDARayCastSDF::parametersFromGlobalAppState, CUDARayCastSDF.h:24-52) */
    uint32_t s_rayCastWidth, s_rayCastHeight;
    float s_SDFRayIncrementFactor, s_SDFRayThresSampleDistFactor, s_SDFRayThresDistFactor;
    int32_t s_SDFUseGradients;
} bf_global_app_state;

typedef struct bf_global_bundling_state {
    int32_t s_enableGlobalTimings, s_enablePerFrameTimings;
    uint32_t s_maxNumImages, s_submapSize, s_widthSIFT, s_heightSIFT, s_maxNumKeysPerImage;
    uint32_t s_numLocalNonLinIterations, s_numLocalLinIterations, s_numGlobalNonLinIterations, s_numGlobalLinIterations;
    uint32_t s_downsampledWidth, s_downsampledHeight;
    float s_verifySiftErrThresh, s_verifySiftCorrThresh, s_projCorrDistThres, s_projCorrNormalThres, s_projCorrColorThresh;
    float s_surfAreaPcaThresh;
    int32_t s_recordSolverConvergence, s_erodeSIFTdepth;
    float s_verifyOptErrThresh, s_verifyOptCorrThresh;
    int32_t s_verbose, s_sendUplinkFeedbackImage;
    float s_depthSigmaD, s_depthSigmaR;
    int32_t s_depthFilter;
    uint32_t s_minNumMatchesLocal, s_minNumMatchesGlobal;
    int32_t s_useComprehensiveFrameInvalidation;
    float s_maxKabschResidual2, s_minKeyScale, s_siftMatchThresh, s_siftMatchRatioMaxLocal, s_siftMatchRatioMaxGlobal;
    int32_t s_useLocalVerify, s_useLocalDense;
    uint32_t s_numOptPerResidualRemoval;
    float s_colorDownSigma, s_depthDownSigmaD, s_depthDownSigmaR;
    float s_optMaxResThresh, s_denseDistThresh, s_denseNormalThresh, s_denseColorThresh, s_denseColorGradientMin;
    float s_denseDepthMin, s_denseDepthMax;
    uint32_t s_denseOverlapCheckSubsampleFactor;
} bf_global_bundling_state;

/* defaults = the values of zParametersDefault.txt / zParametersBundlingDefault.txt as shipped */
BF_API int bf_global_app_state_default(bf_global_app_state* out);
BF_API int bf_global_bundling_state_default(bf_global_bundling_state* out);
/* CUDARayCastSDF::parametersFromGlobalAppState(gas, intrinsics, intrinsicsInv)  CUDARayCastSDF.h:24-52: intrinsics = those of the
 * integration images; they are rescaled when the ray cast size differs (same rule as the image manager). */
/* The confirmation file the application writes when a scan ends (StopScanningAndExit, DepthSensing.cpp:921-957): `valid = true|false`
 * (false when fewer than 800 SDF blocks are left on the heap, or fewer than round(0.5 n) of the n optimised transforms are valid, i.e. not
 * -inf), heapFreeCount, numValidOptTransforms, numTransforms; with `aborted`: `valid = false` + `ABORTED` (invalid first chunk).  Host only.
 * h_optimizedTrajectory = TrajectoryManager::getOptimizedTransforms (n x 16 floats). */
BF_API int bf_write_processed_summary(const char* path, uint32_t heapFreeCount, const float* h_optimizedTrajectory, uint32_t numTransforms, int aborted, int* validOut);
BF_API int bf_ray_cast_params_from_global_app_state(const bf_global_app_state* gas, const float intrinsics[16], bf_ray_cast_params* out);
/* GlobalAppState::readMembers(ParameterFile) GlobalAppState.h:128-136 / GlobalBundlingState.h:90-98.  Starts from
 * the defaults; *numMissing (optional) counts fields that the file did not set (the reference warns per field). */
BF_API int bf_global_app_state_read(const char* filename, bf_global_app_state* out, uint32_t* numMissing);
BF_API int bf_global_bundling_state_read(const char* filename, bf_global_bundling_state* out, uint32_t* numMissing);

/* ------------------------------------------------------------------------- */
/* RGBDSensor accessor contract (RGBDSensor.h:25-61) as plain data            */
/* ------------------------------------------------------------------------- */
typedef struct bf_rgbd_sensor_desc {
    uint32_t depthWidth, depthHeight, colorWidth, colorHeight;
    float depthIntrinsics[16], colorIntrinsics[16];
    float depthExtrinsics[16], colorExtrinsics[16];
} bf_rgbd_sensor_desc;

/* ------------------------------------------------------------------------- */
/* CUDAImageManager:  CUDAImageManager.h:10-337, .cpp:22-158                  */
/* ------------------------------------------------------------------------- */
typedef struct bf_image_manager bf_image_manager;

/* CUDAImageManager(widthIntegration, heightIntegration, widthSIFT, heightSIFT, sensor, storeFramesOnGPU)  .h:140-194 */
BF_API int bf_image_manager_create(uint32_t widthIntegration, uint32_t heightIntegration, uint32_t widthSIFT, uint32_t heightSIFT,
                                   const bf_rgbd_sensor_desc* sensor, const bf_global_bundling_state* gbs, int storeFramesOnGPU,
                                   bf_image_manager** out);
BF_API int bf_image_manager_destroy(bf_image_manager* im);
BF_API int bf_image_manager_set_stream(bf_image_manager* im, void* hip_stream);
/* MI355X addition: the ingest buffers at sensor resolution (d_depthInputRaw / d_depthInputFiltered / d_colorInput, CUDAImageManager.h:268-284) exist
 * four times (frame n uses set n % 4), so that later frames can be ingested on their own stream while frame n's buffers are still being read (feature
 * detection on another stream).  `hip_event` (a hipEvent_t, or null) guards set `set` (0 .. 3): process() waits for it before it overwrites the set; the consumer of
 * a frame's buffers records it after its last read.  The accessors always name the set of the frame ingested last. */
BF_API int bf_image_manager_set_input_guard(bf_image_manager* im, uint32_t set, void* hip_event);
/* MI355X addition: keep, per stored frame, depth and colour ALSO interleaved as 8-byte texels (what the fast voxel update gathers, bf_scene_set_frame_texels):
 * made once at ingest instead of once per operator on the frame.  Before the first frame; needs storeFramesOnGPU.  _get_..._texels returns null when off. */
BF_API int bf_image_manager_set_store_texels(bf_image_manager* im, int enable);
BF_API int bf_image_manager_get_integrate_frame_texels(bf_image_manager* im, uint32_t frame, const void** d_texels);
BF_API int bf_image_manager_reset(bf_image_manager* im);
/* process()  .cpp:22-158.  h_depth = sensor->getDepthFloat() (metres, -inf invalid), h_colorRGBX = getColorRGBX().
 * *gotFrame = 0 when the frame capacity (s_maxNumImages * s_submapSize) is reached.
 * The _device form takes the same two images already resident in HBM (no PCIe transfer).          */
BF_API int bf_image_manager_process(bf_image_manager* im, const float* h_depth, const uint8_t* h_colorRGBX, int* gotFrame);
BF_API int bf_image_manager_process_device(bf_image_manager* im, const float* d_depth, const uint8_t* d_colorRGBX, int* gotFrame);
/* copyToBundling(d_depthRaw, d_depthFilt, d_color)  .h:223-227 */
BF_API int bf_image_manager_copy_to_bundling(bf_image_manager* im, float* d_depthRaw, float* d_depthFilt, uint8_t* d_color);
/* the device input buffers themselves (d_depthInputRaw, d_depthInputFiltered, d_colorInput) */
BF_API int bf_image_manager_get_input_gpu(bf_image_manager* im, const float** d_depthRaw, const float** d_depthFilt, const uint8_t** d_color);
/* getIntegrateFrame(i).getDepthFrameGPU() / getColorFrameGPU()  .h:71-96 */
BF_API int bf_image_manager_get_integrate_frame_gpu(bf_image_manager* im, uint32_t frame, const float** d_depth, const uint8_t** d_color);
/* getIntegrateFrame(i).getDepthFrameCPU() / getColorFrameCPU()  .h:97-119: copies into caller buffers (either may be NULL); syncs */
BF_API int bf_image_manager_get_integrate_frame_cpu(bf_image_manager* im, uint32_t frame, float* h_depth, uint8_t* h_colorRGBX);
BF_API int bf_image_manager_get_curr_frame_number(bf_image_manager* im, uint32_t* out);   /* getCurrFrameNumber (after process) */
BF_API int bf_image_manager_get_num_frames(bf_image_manager* im, uint32_t* out);
BF_API int bf_image_manager_get_integration_size(bf_image_manager* im, uint32_t* width, uint32_t* height);
BF_API int bf_image_manager_get_depth_intrinsics(bf_image_manager* im, float intrinsics[16], float intrinsicsInv[16]);
BF_API int bf_image_manager_get_depth_extrinsics(bf_image_manager* im, float extrinsics[16], float extrinsicsInv[16]);
BF_API int bf_image_manager_get_sift_depth(bf_image_manager* im, uint32_t* width, uint32_t* height, float intrinsics[16]);

/* ------------------------------------------------------------------------- */
/* Bundler:  Bundler.h:17-104, Bundler.cpp                                     */
/* ------------------------------------------------------------------------- */
typedef struct bf_bundler bf_bundler;

/* Bundler(maxNumImages, maxNumKeysPerImage, siftIntrinsicsInv, imageManager, isLocal)  Bundler.cpp:19-54 */
BF_API int bf_bundler_create(uint32_t maxNumImages, uint32_t maxNumKeysPerImage, const float siftIntrinsicsInv[16],
                             bf_image_manager* manager, int isLocal, const bf_global_app_state* gas,
                             const bf_global_bundling_state* gbs, bf_bundler** out);
BF_API int bf_bundler_destroy(bf_bundler* b);
BF_API int bf_bundler_set_stream(bf_bundler* b, void* hip_stream);
BF_API int bf_bundler_get_trajectory_gpu(bf_bundler* b, float** d_trajectory);                 /* getTrajectoryGPU */
BF_API int bf_bundler_get_valid_images(bf_bundler* b, int32_t* h_out, uint32_t count);          /* getValidImages */
BF_API int bf_bundler_get_cache_intrinsics(bf_bundler* b, float intrinsics[16], float intrinsicsInv[16]);
BF_API int bf_bundler_get_curr_frame_number(bf_bundler* b, uint32_t* out);
BF_API int bf_bundler_get_num_frames(bf_bundler* b, uint32_t* out);
BF_API int bf_bundler_is_valid(bf_bundler* b, int* out);                                        /* isValid :295-304 */
BF_API int bf_bundler_reset(bf_bundler* b);                                                     /* :354-360 */
BF_API int bf_bundler_detect_features(bf_bundler* b, const float* d_intensitySift, const float* d_inputDepthFilt);  /* :91-101 */
BF_API int bf_bundler_store_cached_frame(bf_bundler* b, uint32_t depthWidth, uint32_t depthHeight, const uint8_t* d_inputColor,
                                         uint32_t colorWidth, uint32_t colorHeight, const float* d_inputDepthRaw);  /* :279-282 */
BF_API int bf_bundler_copy_frame(bf_bundler* b, bf_bundler* from, uint32_t frame);              /* :284-293 */
BF_API int bf_bundler_add_invalid_frame(bf_bundler* b);                                         /* :362-368 */
BF_API int bf_bundler_invalidate_last_frame(bf_bundler* b);                                     /* :375-386 */
BF_API int bf_bundler_get_current_sift_transforms_gpu(bf_bundler* b, const float** d_out);      /* getCurrentSiftTransformsGPU */
BF_API int bf_bundler_get_num_filt_matches_gpu(bf_bundler* b, const int32_t** d_out);
BF_API int bf_bundler_match_and_filter(bf_bundler* b, uint32_t* lastMatchedFrame);              /* :103-249 */
BF_API int bf_bundler_optimize(bf_bundler* b, uint32_t numNonLinIterations, uint32_t numLinIterations, int bUseVerify,
                               int bRemoveMaxResidual, int bIsScanDone, int* bOptRemoved, int* valid);   /* :251-277 */
BF_API int bf_bundler_set_solve_weights(bf_bundler* b, const float* sparse, const float* denseDepth, const float* denseColor, uint32_t n);
BF_API int bf_bundler_fuse_to_global(bf_bundler* b, bf_bundler* glob);                           /* :388-394 */
BF_API int bf_bundler_try_revalidation(bf_bundler* b, uint32_t curGlobalFrame, int bIsScanDone, uint32_t* revalidatedIdx);  /* :306-352 */
BF_API int bf_bundler_get_revalidated_idx(bf_bundler* b, uint32_t* out);
BF_API int bf_bundler_save_sparse_corrs_to_file(bf_bundler* b, const char* filename);           /* :396-409 */
/* the operators a Bundler owns (for inspection) */
BF_API int bf_bundler_get_sift_manager(bf_bundler* b, bf_siftmgr** out);
BF_API int bf_bundler_get_cache(bf_bundler* b, bf_cache** out);
BF_API int bf_bundler_get_solver(bf_bundler* b, bf_solver** out);

/* ------------------------------------------------------------------------- */
/* CorrespondenceEvaluator:  CorrespondenceEvaluator.h:39-141, .cpp           */
/* (compiled in the reference under EVALUATE_SPARSE_CORRESPONDENCES,          */
/*  GlobalBundlingState.h:11; always available here, active once attached)    */
/* ------------------------------------------------------------------------- */
typedef struct bf_corr_evaluation {       /* struct CorrEvaluation  .h:10-37 */
    uint32_t numCorrect;                  /* image pairs with matches whose largest world-space error is below the threshold AND that overlap */
    uint32_t numDetected;                 /* overlapping image pairs for which matches were found */
    uint32_t numTotal;                    /* overlapping image pairs (by the reference trajectory) */
} bf_corr_evaluation;
BF_API float bf_corr_evaluation_get_precision(const bf_corr_evaluation* e);     /* numCorrect / numDetected, -inf without detections */
BF_API float bf_corr_evaluation_get_recall(const bf_corr_evaluation* e);        /* numDetected / numTotal,   -inf without overlaps   */

typedef struct bf_corr_eval_params {      /* what computeCachedData reads from GlobalBundlingState  .cpp:15-19 */
    float depthMin, depthMax;             /* s_denseDepthMin / s_denseDepthMax */
    float distThresh, normalThresh, colorThresh;   /* s_projCorrDistThres / s_projCorrNormalThres / s_projCorrColorThresh (colour is unused, .cpp:198) */
} bf_corr_eval_params;

typedef struct bf_correspondence_evaluator bf_correspondence_evaluator;
/* CorrespondenceEvaluator(referenceTrajectory, logFilePrefix)  .h:42-57: one 4x4 per image of the manager that will be evaluated;
 * logFilePrefix NULL or "" = no files, otherwise <prefix>_frame.csv and <prefix>_wrong.csv are written in the reference's format. */
BF_API int bf_correspondence_evaluator_create(const float* h_referenceTrajectory, uint32_t numTransforms, const char* logFilePrefix,
                                              bf_correspondence_evaluator** out);
BF_API int bf_correspondence_evaluator_destroy(bf_correspondence_evaluator* ev);
/* evaluate(siftManager, cudaCache, siftIntrinsicsInv, filtered, recomputeCache, clearCache, corrType)  .cpp:47-96.
 * Synchronises hip_stream (the stream the manager's matching was enqueued on).  out may be NULL. */
BF_API int bf_correspondence_evaluator_evaluate(bf_correspondence_evaluator* ev, bf_siftmgr* siftManager, bf_cache* cache,
                                                const float siftIntrinsicsInv[16], const bf_corr_eval_params* params, int filtered,
                                                int recomputeCache, int clearCache, const char* corrType, void* hip_stream,
                                                bf_corr_evaluation* out);
BF_API int bf_correspondence_evaluator_finish_logging_to_file(bf_correspondence_evaluator* ev);       /* .h:64-69 */
/* sum of all evaluate() results for one corrType ("raw", "kabsch", "sa", "dense") since creation */
BF_API int bf_correspondence_evaluator_get_total(bf_correspondence_evaluator* ev, const char* corrType, bf_corr_evaluation* out);
/* the counters of the last computeCachedData: per image {numCorr, numValid} cur->prev and prev->cur, and the overlap flags */
BF_API int bf_correspondence_evaluator_get_overlap_counts(bf_correspondence_evaluator* ev, uint32_t* h_counts4, uint8_t* h_hasGTCorr,
                                                          uint32_t numFrames);
/* Bundler::initializeCorrespondenceEvaluator / finishCorrespondenceEvaluatorLogging  Bundler.h:61-67; with an evaluator attached,
 * matchAndFilter evaluates after each stage (Bundler.cpp:145-204), which synchronises the stream four times per frame. */
BF_API int bf_bundler_initialize_correspondence_evaluator(bf_bundler* b, const float* h_trajectory, uint32_t numTransforms, const char* logFilePrefix);
BF_API int bf_bundler_finish_correspondence_evaluator_logging(bf_bundler* b);
BF_API int bf_bundler_get_correspondence_evaluator(bf_bundler* b, bf_correspondence_evaluator** out);   /* NULL when none is attached */

/* ------------------------------------------------------------------------- */
/* TrajectoryManager:  TrajectoryManager.h:6-116, .cpp                        */
/* ------------------------------------------------------------------------- */
typedef struct bf_trajectory_manager bf_trajectory_manager;
enum { BF_TF_INTEGRATED = 0, BF_TF_NOT_INTEGRATED_NO_TRANSFORM = 1, BF_TF_NOT_INTEGRATED_WITH_TRANSFORM = 2, BF_TF_INVALID = 3, BF_TF_REINTEGRATION = 4 };

BF_API int bf_trajectory_manager_create(uint32_t numMaxImage, uint32_t topNActive, float minPoseDistSqrt, bf_trajectory_manager** out);
BF_API int bf_trajectory_manager_destroy(bf_trajectory_manager* tm);
BF_API int bf_trajectory_manager_add_frame(bf_trajectory_manager* tm, int type, const float transform[16], uint32_t idx);
BF_API int bf_trajectory_manager_update_optimized_transform(bf_trajectory_manager* tm, const float* d_trajectory, uint32_t numFrames, void* hip_stream);
BF_API int bf_trajectory_manager_update_optimized_transform_host(bf_trajectory_manager* tm, const float* h_trajectory, uint32_t numFrames);   /* host copy of the trajectory */
BF_API int bf_trajectory_manager_generate_update_lists(bf_trajectory_manager* tm);
BF_API int bf_trajectory_manager_confirm_integration(bf_trajectory_manager* tm, uint32_t frameIdx);
BF_API int bf_trajectory_manager_get_top_from_reintegrate_list(bf_trajectory_manager* tm, float oldTransform[16], float newTransform[16], uint32_t* frameIdx, int* found);
BF_API int bf_trajectory_manager_get_top_from_integrate_list(bf_trajectory_manager* tm, float trans[16], uint32_t* frameIdx, int* found);
BF_API int bf_trajectory_manager_get_top_from_deintegrate_list(bf_trajectory_manager* tm, float trans[16], uint32_t* frameIdx, int* found);
BF_API int bf_trajectory_manager_get_num_optimized_frames(bf_trajectory_manager* tm, uint32_t* out);
BF_API int bf_trajectory_manager_get_num_added_frames(bf_trajectory_manager* tm, uint32_t* out);
BF_API int bf_trajectory_manager_get_num_active_operations(bf_trajectory_manager* tm, uint32_t* out);
BF_API int bf_trajectory_manager_get_optimized_transforms(bf_trajectory_manager* tm, float* h_out, uint32_t capacity, uint32_t* count);
/* getFrames(): type and integrated transform of frame i */
BF_API int bf_trajectory_manager_get_frame(bf_trajectory_manager* tm, uint32_t idx, int* type, float integratedTransform[16], float* dist);

/* ------------------------------------------------------------------------- */
/* OnlineBundler:  OnlineBundler.h:10-106, .cpp, .cu                          */
/* ------------------------------------------------------------------------- */
typedef struct bf_online_bundler bf_online_bundler;

/* OnlineBundler(sensor, imageManager)  OnlineBundler.cpp:31-91 */
BF_API int bf_online_bundler_create(const bf_rgbd_sensor_desc* sensor, bf_image_manager* imageManager, const bf_global_app_state* gas,
                                    const bf_global_bundling_state* gbs, bf_online_bundler** out);
BF_API int bf_online_bundler_destroy(bf_online_bundler* ob);
BF_API int bf_online_bundler_set_stream(bf_online_bundler* ob, void* hip_stream);
BF_API int bf_online_bundler_process_input(bf_online_bundler* ob);                               /* processInput :167-227 */
/* the same in two halves: _begin enqueues all device work of processInput, _end performs its single read-back and the host
 * logic; independent work (e.g. re-integration on another stream) may be enqueued in between.  Up to TWO calls may be in flight
 * (_begin(k), _begin(k+1), _end(k), ...; _end completes the oldest): everything frame k+1's matching chain needs of frame k is
 * device state - the fix-up of an unconnected frame (OnlineBundler.cpp:215-221) is a kernel - so a host loop never has to wait
 * for the chain it has just enqueued.  Exception in the serial order: for a chunk's last frame k, run _end(k) and process()
 * before _begin(k+1) (its solves write what the next chain's pose kernel reads). */
BF_API int bf_online_bundler_process_input_begin(bf_online_bundler* ob);
BF_API int bf_online_bundler_process_input_end(bf_online_bundler* ob);
/* Detect-ahead: feature detection and the dense cache frame depend only on a frame's pixels.  _detect_ahead computes them for the
 * image manager's current frame on `detect stream` (the stream the image manager ingests on) into a two-slot staging area,
 * without touching bundler state; _process_input_begin_frame(frame) later commits that staged frame instead of detecting, even
 * when the image manager has ingested a newer frame meanwhile.  A host loop can so overlap detection of frame k+1 with the
 * matching / solving of frame k (what the reference's bundling thread does with its one-frame lag, OnlineBundler.cpp:167). */
BF_API int bf_online_bundler_set_detect_stream(bf_online_bundler* ob, void* hip_stream);
/* Odd frames detect on a second stream with a detector of their own (two detections in flight); null: one stream again.  After set_detect_stream. */
BF_API int bf_online_bundler_set_second_detect_stream(bf_online_bundler* ob, void* hip_stream);
BF_API int bf_online_bundler_detect_ahead(bf_online_bundler* ob);
/* the same when the ingest runs on ANOTHER stream than the detect stream: `ingest_event` (a hipEvent_t) was recorded behind the frame's ingest */
BF_API int bf_online_bundler_detect_ahead_after(bf_online_bundler* ob, void* ingest_event);
BF_API int bf_online_bundler_process_input_begin_frame(bf_online_bundler* ob, uint32_t frame);
BF_API int bf_online_bundler_process(bf_online_bundler* ob, uint32_t numNonLinItersLocal, uint32_t numLinItersLocal,
                                     uint32_t numNonLinItersGlobal, uint32_t numLinItersGlobal);  /* process :410-416 */
/* process() for the body of an explicit frame (what a loop with frames in flight calls) */
BF_API int bf_online_bundler_process_frame(bf_online_bundler* ob, uint32_t frame, uint32_t numNonLinItersLocal, uint32_t numLinItersLocal,
                                           uint32_t numNonLinItersGlobal, uint32_t numLinItersGlobal);
/* Lagged solve - the reference's optimiser thread (FriedLiver.cpp:112-143, ConditionManager.h:49-78) with a DEFINED hand-over point:
 * with lag L in [1, s_submapSize], process() for a chunk-closing frame b only STARTS optimizeLocal + processGlobal + optimizeGlobal
 * (OnlineBundler.cpp:242-408) on an own host thread and on `solve_stream`; what they publish - the complete trajectory, the last
 * valid complete transform, the TrajectoryManager's optimised poses, the tracking-lost flag - becomes visible exactly when frame
 * b + L enters processInput (_begin) / its re-integration scheduling (_apply_lagged_solve, which a host loop calls before it consults
 * the TrajectoryManager for that frame), and at the first iteration past the end of the sequence at the latest.  Results are a function
 * of (input, L): reproducible, and comparable with the oracle loop under the same L.  L = 0 (the online bundler's default; bf_pipeline_create chooses 10): the serial order. */
/* Pair stages side by side: with two streams of the caller's, the staged detection of frame k is committed and its per-pair kernels (match, Kabsch filter,
 * surface-area filter, dense verification - each pair depends on its two images only) run on stream k & 1, into the sift manager's result set k & 1
 * (bf_siftmgr_set_pair_stage), while frame k - 1's are still running on the other stream; the bundling stream carries the short commit stage (which
 * previous images are valid, filterFrames, EntryJ rows, the pose kernel).  Same results as on one stream.  (null, null): off. */
BF_API int bf_online_bundler_set_pair_streams(bf_online_bundler* ob, void* hip_stream0, void* hip_stream1);
BF_API int bf_online_bundler_set_solve_lag(bf_online_bundler* ob, uint32_t lag, void* solve_stream);
BF_API int bf_online_bundler_get_solve_lag(bf_online_bundler* ob, uint32_t* lag);
BF_API int bf_online_bundler_apply_lagged_solve(bf_online_bundler* ob, uint32_t frame);
BF_API int bf_online_bundler_wait_solves(bf_online_bundler* ob);
/* getCurrentIntegrationFrame(siftTransform, frameIdx, bGlobalTrackingLost) -> valid  :229-240 */
BF_API int bf_online_bundler_get_current_integration_frame(bf_online_bundler* ob, float siftTransform[16], uint32_t* frameIdx,
                                                           int* bGlobalTrackingLost, int* valid);
BF_API int bf_online_bundler_get_trajectory_manager(bf_online_bundler* ob, bf_trajectory_manager** out);
BF_API int bf_online_bundler_get_curr_processed_frame(bf_online_bundler* ob, int32_t* out);
BF_API int bf_online_bundler_get_bundler(bf_online_bundler* ob, int which /* 0 local, 1 optLocal, 2 global */, bf_bundler** out);
BF_API int bf_online_bundler_get_complete_trajectory(bf_online_bundler* ob, float* h_out, uint32_t capacity, uint32_t* count);
BF_API int bf_online_bundler_save_global_sparse_corrs_to_file(bf_online_bundler* ob, const char* filename);
/* OnlineBundler.cpp:81-90: attach a CorrespondenceEvaluator to the GLOBAL bundler; h_completeTrajectory holds one reference pose per
 * input frame, every s_submapSize-th is used.  finishCorrespondenceEvaluatorLogging  .cpp:480-487. */
BF_API int bf_online_bundler_initialize_correspondence_evaluator(bf_online_bundler* ob, const float* h_completeTrajectory, uint32_t numFrames,
                                                                 const char* logFilePrefix);
BF_API int bf_online_bundler_finish_correspondence_evaluator_logging(bf_online_bundler* ob);
/* the three trajectory kernels of OnlineBundler.cu (computeSiftTransformCU / updateTrajectoryCU / initNextGlobalTransformCU) */
BF_API int bf_compute_sift_transform(const float* d_currFilteredTransformsInv, const int32_t* d_currNumFilteredMatchesPerImagePair,
                                     const float* d_completeTrajectory, uint32_t lastValidCompleteTransform, float* d_siftTrajectory,
                                     uint32_t curFrameIndexAll, uint32_t curFrameIndex, float* d_currIntegrateTrans, void* hip_stream);
BF_API int bf_update_trajectory(const float* d_globalTrajectory, uint32_t numGlobalTransforms, float* d_completeTrajectory,
                                uint32_t numCompleteTransforms, const float* d_localTrajectories, uint32_t numLocalTransformsPerTrajectory,
                                uint32_t numLocalTrajectories, const int32_t* d_imageInvalidateList, void* hip_stream);
BF_API int bf_init_next_global_transform(float* d_globalTrajectory, uint32_t numGlobalTransforms, uint32_t initGlobalIdx,
                                         const float* d_localTrajectories, uint32_t lastValidLocal, uint32_t numLocalTransformsPerTrajectory,
                                         void* hip_stream);

/* ------------------------------------------------------------------------- */
/* Headless frame loop: the serial (non RUN_MULTITHREADED) body of            */
/* OnD3D11FrameRender, DepthSensing/DepthSensing.cpp:966-1095, with           */
/* integrate / deIntegrate / reintegrate :723-762, :854-902                   */
/* ------------------------------------------------------------------------- */
typedef struct bf_pipeline bf_pipeline;

typedef struct bf_frame_timing {      /* TimingLog::FrameTiming, TimingLog.h:9-21 (milliseconds, hipEvent based) */
    float timeSensorProcess, timeSiftDetection, timeSiftMatching, timeMatchFilter, timeSolve, timeReIntegrate, timeReconstruct, timeTotal;
} bf_frame_timing;

BF_API int bf_pipeline_create(const bf_global_app_state* gas, const bf_global_bundling_state* gbs, const bf_rgbd_sensor_desc* sensor,
                              bf_pipeline** out);
BF_API int bf_pipeline_destroy(bf_pipeline* p);
/* Multi-GPU mode "volume shard": every rank runs the (bit-deterministic) bundling on the whole stream and integrates only
 * its hash-bucket shard of the volume (bf_scene_set_shard).  Call before the first frame. */
BF_API int bf_pipeline_set_volume_shard(bf_pipeline* p, uint32_t rank, uint32_t world);
/* Batched volume operators (default on): the integration of a frame and the up to s_maxFrameFixes re-integrations of the next frame (DepthSensing.cpp:854-902,
 * :966-1095) are issued as ONE bf_scene_run_batch when that frame's garbage collection is posted - same order, same volume as one operator at a time
 * (enable = 0), which stays available for comparison. */
BF_API int bf_pipeline_set_volume_batching(bf_pipeline* p, int enable);
/* Lagged solve for the whole loop (bf_online_bundler_set_solve_lag on a stream of the pipeline's): the chunk solves leave the frame
 * loop's critical path and are applied `lag` frames after the frame that closed the chunk.  bf_pipeline_create sets lag = min(10, s_submapSize)
 * (the reference's optimiser thread, FriedLiver.cpp:112-143, with a defined hand-over; round 6); 0 = the serial order of the reference's single-threaded
 * branch (also BF_PIPELINE_SOLVE_LAG=0 in the environment).  Only between frames. */
/* measurement aid: seconds the volume thread spent issuing TSDF operators (HIP API calls) and the number of operators, since the last reset */
BF_API int bf_pipeline_get_volume_thread_profile(bf_pipeline* p, double* busySeconds, double* commands, int reset);
BF_API int bf_pipeline_set_solve_lag(bf_pipeline* p, uint32_t lag);
BF_API int bf_pipeline_get_solve_lag(bf_pipeline* p, uint32_t* lag);
/* one iteration of the frame loop with a new sensor frame (host or device resident).  The two buffers may be reused as soon as
 * the call returns.  With the look-ahead (default; BF_PIPELINE_LOOKAHEAD=0 disables it) the loop runs two frames behind its input:
 * matching / integration / solves of this frame are completed by the call after the next, by bf_pipeline_synchronize or by any
 * accessor below - results are those of the serial order. */
BF_API int bf_pipeline_process_frame(bf_pipeline* p, const float* h_depth, const uint8_t* h_colorRGBX, int* gotFrame);
BF_API int bf_pipeline_process_frame_device(bf_pipeline* p, const float* d_depth, const uint8_t* d_colorRGBX, int* gotFrame);
/* one iteration after the sensor stopped delivering frames (solve + re-integration continue, :175-196) */
BF_API int bf_pipeline_process_end_of_sequence(bf_pipeline* p, uint32_t* numActiveOperations);
BF_API int bf_pipeline_synchronize(bf_pipeline* p);
BF_API int bf_pipeline_get_scene(bf_pipeline* p, bf_scene** out);
BF_API int bf_pipeline_get_image_manager(bf_pipeline* p, bf_image_manager** out);
BF_API int bf_pipeline_get_online_bundler(bf_pipeline* p, bf_online_bundler** out);
BF_API int bf_pipeline_get_num_frames(bf_pipeline* p, uint32_t* out);
/* camera-to-world poses the frames are currently integrated with (-inf matrix: not integrated) */
BF_API int bf_pipeline_get_integrated_trajectory(bf_pipeline* p, float* h_out, uint32_t capacity, uint32_t* count);
BF_API int bf_pipeline_get_counters(bf_pipeline* p, uint32_t* numIntegrate, uint32_t* numDeIntegrate, uint32_t* numLocalSolves, uint32_t* numGlobalSolves);
/* Wall time (seconds, accumulated) the calling thread spent in the parts of the frame loop, without synchronising anything:
 * [0] enqueue of the previous frame's matching chain, [1] ingest + detection enqueue, [2] re-integration commands, [3] wait for the
 * matching result + host logic, [4] integration command, [5] local / global solves, [6] wait for the ingest, [7] number of frames. */
BF_API int bf_pipeline_get_host_profile(bf_pipeline* p, double out[8], int reset);
BF_API int bf_pipeline_enable_timings(bf_pipeline* p, int enable);
BF_API int bf_pipeline_get_last_timing(bf_pipeline* p, bf_frame_timing* out);

/* ------------------------------------------------------------------------- */
/* Chunk-parallel bundling (multi-GPU partition of ONE stream) */
/*                                                                             */
/* Everything OnlineBundler does for a frame BEFORE the global step reads only */
/* the frames of its own local chunk: ingest, SIFT, cache frame, matching and   */
/* filtering against the chunk (Bundler::matchAndFilter on m_local,             */
/* OnlineBundler.cpp:118-127), the local solve (:242-271) and the fusion into   */
/* one key frame (SIFTImageManager::fuseToGlobal, .cpp:414-476).  A             */
/* bf_chunk_worker runs exactly that for one chunk and returns the result as a  */
/* flat host package; local chunks are dealt round-robin to the ranks, the       */
/* packages are all-gathered (RCCL), and every rank then runs the global half    */
/* (pose chaining, global match + solve, TrajectoryManager, re-integration       */
/* scheduling) on all packages in stream order with                              */
/* bf_pipeline_process_frame_chunked, integrating into its hash-bucket shard of  */
/* the volume.  Results are those of the serial loop, bit for bit.               */
/* ------------------------------------------------------------------------- */
#define BF_CHUNK_MAX_FRAMES 32u          /* s_submapSize + 1 must not exceed this */
#define BF_CHUNK_MAGIC 0x4B434642u       /* "BFCK" */

typedef struct bf_chunk_frame_record {   /* what processInput learns about one frame from its chunk (OnlineBundler.cpp:118-132, OnlineBundler.cu:5-52) */
    int32_t valid;                       /* matchAndFilter returned a previous frame of the chunk (lastMatchedFrame != -1) */
    int32_t prevLocal;                   /* local index of the previous frame the SIFT pose chains from: the largest i with filtered matches; -1 if none */
    float relInv[16];                    /* d_currFilteredTransformsInv[prevLocal] */
} bf_chunk_frame_record;

typedef struct bf_chunk_header {
    uint32_t magic, chunkIndex, numFrames /* frames of the chunk, the one shared with the previous chunk included */, submapSize;
    int32_t chunkValid;                  /* Bundler::isValid() of the chunk before its solve (OnlineBundler.cpp:147) */
    int32_t solveValid;                  /* Bundler::optimize() verdict (.cpp:255); 0 when the chunk was not solved */
    uint32_t numKeys, maxKeys, cacheWidth, cacheHeight;
    int32_t validImages[BF_CHUNK_MAX_FRAMES];                 /* getValidImages() of the chunk after the solve */
    float localTrajectory[BF_CHUNK_MAX_FRAMES * 16];          /* the chunk's optimised poses (getTrajectoryGPU) */
    bf_chunk_frame_record frames[BF_CHUNK_MAX_FRAMES];
    uint64_t offKeys, offDescs;          /* fused key frame: numKeys x 16 B key points, numKeys x 128 B descriptors (byte offsets from the package start) */
    uint64_t offCache[6];                /* cache frame of the chunk's first image: depth, camera positions, intensity, derivatives, normals u8x4, normals f32x4 */
    uint64_t totalBytes;
} bf_chunk_header;

typedef struct bf_chunk_worker bf_chunk_worker;
BF_API int bf_chunk_worker_create(const bf_global_app_state* gas, const bf_global_bundling_state* gbs, const bf_rgbd_sensor_desc* sensor, bf_chunk_worker** out);
BF_API int bf_chunk_worker_destroy(bf_chunk_worker* w);
BF_API int bf_chunk_worker_set_stream(bf_chunk_worker* w, void* hip_stream);
BF_API int bf_chunk_worker_package_bytes(bf_chunk_worker* w, uint64_t* bytes);        /* size of one package (fixed for a configuration) */
/* the chunk-local half of the loop for chunk `chunkIndex`: numFrames device frames in stream order (frames chunkIndex*S .. chunkIndex*S + numFrames-1) */
BF_API int bf_chunk_worker_run(bf_chunk_worker* w, uint32_t chunkIndex, uint32_t numFrames, const float* const* d_depth, const uint8_t* const* d_colorRGBX,
                               void* h_package);
/* one iteration of the frame loop for the next frame of the stream, whose chunk-local half is in `h_package` (localIdx = index of the frame in that
 * package's chunk; 0 only for the very first frame).  Ingest for integration, pose chaining, integration / re-integration into this pipeline's volume
 * (shard), and — at the chunk's last frame — the global step.  The stream must hold 1 + k * s_submapSize frames.
 * `packageBytes` is the size of the buffer the package arrived in: every offset, count and index of the header is checked against it and against this
 * pipeline's configuration before it is used (BF_ERR_INVALID_ARG otherwise). */
BF_API int bf_pipeline_process_frame_chunked(bf_pipeline* p, const float* d_depth, const uint8_t* d_colorRGBX, const void* h_package, uint64_t packageBytes, uint32_t localIdx,
                                             int* gotFrame);

#ifdef __cplusplus
}
#endif
#endif /* BF_PIPELINE_H */
