/*
 * bf_comm.h — C ABI of the ONE exchange step the multi-GPU partition of the hot path has: an all-gather.
 *
 * The reference's only multi-device mechanism is the split "reconstruction on GPU 0, bundling on GPU 1" with cudaMemcpyPeer between
 * them (DualGPU.h:108-134, FriedLiver.cpp:120,127,253,273).  Here one process drives one GPU; what crosses GPUs is
 *   * per round of `world` local chunks: the key-frame packages (bf_chunk_exchange, ~0.4 MB per chunk), and
 *   * per TSDF operator, when the allocation's ray march is divided over the ranks: the block keys each rank collected on its band of
 *     the pixel tiles (bf_scene_set_alloc_comm in bf_hip.h; a few thousand 8-byte keys per rank).
 * Both are all-gathers of fixed-size records, issued on a HIP stream of the caller's, so a C++ host has the whole multi-GPU path
 * without Python.
 *
 * A bf_comm is either
 *   * RCCL: librccl is loaded at first use (dlopen; libbf_hip.so does not link it).  Bootstrap like any NCCL program: rank 0 calls
 *     bf_comm_unique_id, the host language hands the 128 bytes to the other ranks (MPI, a file, torch.distributed's store ...), every
 *     rank calls bf_comm_create_rccl.  bf_comm_from_rccl borrows an existing ncclComm_t instead.
 *   * a callback: the host supplies the all-gather (tests run two ranks on ONE GPU over gloo this way; any other transport fits).
 */
#ifndef BF_COMM_H
#define BF_COMM_H

#include "bf_pipeline.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bf_comm bf_comm;

#define BF_COMM_UNIQUE_ID_BYTES 128          /* == NCCL_UNIQUE_ID_BYTES */

/* ncclGetUniqueId */
BF_API int bf_comm_unique_id(uint8_t id[BF_COMM_UNIQUE_ID_BYTES]);
/* ncclCommInitRank on the calling thread's current device (collective: every rank of the communicator calls it) */
BF_API int bf_comm_create_rccl(const uint8_t id[BF_COMM_UNIQUE_ID_BYTES], uint32_t world, uint32_t rank, bf_comm** out);
/* wrap a communicator the host already has (an ncclComm_t of librccl); it is not destroyed with the bf_comm */
BF_API int bf_comm_from_rccl(void* nccl_comm, uint32_t world, uint32_t rank, bf_comm** out);
/* The host's own all-gather: fn(user, d_send, d_recv, bytes_per_rank, hip_stream) must leave rank r's `bytes_per_rank` bytes at
 * d_recv + r * bytes_per_rank on every rank, ordered on `hip_stream` like a kernel (it may synchronise the stream itself and copy
 * through the host).  Returns 0 on success.  It is called from the thread that issues the exchange (the volume thread of a
 * bf_pipeline for allocation keys). */
typedef int (*bf_all_gather_fn)(void* user, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
BF_API int bf_comm_create_callback(bf_all_gather_fn fn, void* user, uint32_t world, uint32_t rank, bf_comm** out);
BF_API int bf_comm_destroy(bf_comm* c);
BF_API int bf_comm_world(bf_comm* c, uint32_t* world, uint32_t* rank);
/* all-gather of `bytes_per_rank` device bytes per rank on `hip_stream` (RCCL: ncclAllGather, ring over xGMI) */
BF_API int bf_comm_all_gather(bf_comm* c, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);

/* One round of the chunk-parallel mode (bf_pipeline.h, "Chunk-parallel bundling"): every rank contributes the package its
 * bf_chunk_worker produced for its chunk of the round (h_mine, package_bytes; all zeros when the rank had no chunk) and receives all
 * `world` packages in owner order into h_all (world * package_bytes host bytes).  One all-gather through device staging buffers of
 * the comm (allocated at first use), on `hip_stream`; synchronous for the caller: h_all is complete when the call returns. */
BF_API int bf_chunk_exchange(bf_comm* c, const void* h_mine, void* h_all, uint64_t package_bytes, void* hip_stream);

/* The TSDF operators' own allocation with the ray march DIVIDED over the ranks (VoxelUtilHashSDF.h:478-479,623 is the table it fills;
 * CUDASceneRepHashSDF.cu:165-251 the march).  With the volume sharded by home bucket (bf_scene_set_shard) every rank would have to march all pixels to
 * find its own blocks.  Instead, inside every integrate / re-integrate: rank r marches a band of the 8x8 pixel tiles and collects the distinct in-frustum
 * block keys it meets; ONE all-gather of fixed-size records {count, keys[capacity_keys]} on the scene's allocation stream; every rank queues the keys of
 * every list whose home bucket it owns, then places.  The table is the one the local march builds (queuing is idempotent, bins are sorted before
 * placement).  Every rank must issue the same operator sequence.  capacity_keys bounds the keys one rank collects per operator (exceeding it raises the
 * scene's error flag).  comm == null: back to the local march. */
BF_API int bf_scene_set_alloc_comm(bf_scene* s, bf_comm* comm, uint32_t capacity_keys);
/* the same for the volume of a frame loop: the pipeline's volume thread issues the collectives (one per integrate / re-integrate), on the volume's
 * allocation stream; bf_pipeline_set_volume_shard(rank, world) of the same communicator must have been called.  Every rank must feed the same frames. */
BF_API int bf_pipeline_set_comm(bf_pipeline* p, bf_comm* comm, uint32_t capacity_keys);

#ifdef __cplusplus
}
#endif
#endif
