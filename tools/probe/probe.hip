// Memory-system probe for the voxel update: what does THIS device deliver for the voxel update's access pattern with no arithmetic
// and no gathers at all?  One wave per SDF block, blocks of 6144 bytes scattered over the heap in list order (the frustum list's order), every block read
// completely (6 x 1 KB rows as global_load_dwordx4: the widest, fully coalesced form) and `writeRows12` twelfths of it written back - the update itself
// reads 8 slices of 768 B as dwordx3 and writes the ~6 slices some lane touched.  The kernel's launch geometry is the update's (8192 workgroups of four
// waves striding over the list).  tools/hbm_block_probe.py times it; the quotient (update's bytes per second) / (probe's bytes per second) is how much of
// what the memory system gives to this pattern the update reaches.
#include <hip/hip_runtime.h>

#include <stdint.h>

#define PROBE_TRY(expr) do { if ((expr) != hipSuccess) return 2; } while (0)

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_probe_blocks(const uint32_t* __restrict__ list, uint32_t n, uint8_t* __restrict__ heap, uint32_t writeRows12, uint32_t salt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), nWaves = gridDim.x * 4u;
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        v4u* base = reinterpret_cast<v4u*>(heap + (size_t)list[blk] * 6144u);      // wave-uniform
        v4u r[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) r[j] = base[j * 64 + lane];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // row j is written when 2 j + 1 < writeRows12 (whole row) or, for the row the count ends in, by the lower half of the lanes
            const uint32_t full = 2u * (uint32_t)j + 2u <= writeRows12 ? 1u : 0u, half = 2u * (uint32_t)j + 1u == writeRows12 ? 1u : 0u;
            if (full || (half && lane < 32u)) { v4u v = r[j]; v.x ^= salt; base[j * 64 + lane] = v; }
        }
    }
}

// The same walk with the voxel update's own access widths: a block is 8 slices of 64 voxels x 12 bytes; lane l reads the three dwords of voxel l of every slice
// (three global_load_dword per slice, lanes 12 bytes apart: together the 768 contiguous bytes of the slice) and writes `writeSlices` of the 8 slices back the same
// way - what k_update_apx / k_update_batch_apx issue (tsdf.hip, tsdf_batch.h).  Used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS access pattern
// (tools/pmc_calibrate.py): the bytes it moves are known exactly.
__global__ __launch_bounds__(256) void k_probe_slices(const uint32_t* __restrict__ list, uint32_t n, uint8_t* __restrict__ heap, uint32_t writeSlices, uint32_t salt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), nWaves = gridDim.x * 4u;
    for (uint32_t blk = wave; blk < n; blk += nWaves) {
        uint32_t* base = reinterpret_cast<uint32_t*>(heap + (size_t)list[blk] * 6144u) + lane * 3u;      // voxel `lane` of slice 0
        uint32_t r[8][3];
#pragma unroll
        for (int z = 0; z < 8; ++z)
#pragma unroll
            for (int k = 0; k < 3; ++k) r[z][k] = base[z * 192 + k];
#pragma unroll
        for (int z = 0; z < 8; ++z)
            if ((uint32_t)z < writeSlices) {
#pragma unroll
                for (int k = 0; k < 3; ++k) base[z * 192 + k] = r[z][k] ^ (k == 0 ? salt : 0u);
            }
    }
}

}  // namespace

extern "C" {

// `reps` launches of the slice-pattern kernel (no timing: a counter-collection run)
__attribute__((visibility("default"))) int bf_probe_slices(uint8_t* d_heap, const uint32_t* d_list, uint32_t n, uint32_t writeSlices, uint32_t grid, uint32_t reps, void* hip_stream) {
    if (!(d_heap && d_list && reps > 0 && grid > 0 && writeSlices <= 8)) return 1;
    for (uint32_t r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_slices, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_list, n, d_heap, writeSlices, r + 2u);
    PROBE_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    return 0;
}


// heap: numBlocks x 6144 bytes; d_list: n block indices (< numBlocks).  writeRows12: twelfths of a block written back (9 = 4.6 KB of 6.1 KB).  Runs `reps`
// launches on `hip_stream` between two events and returns the mean launch time in microseconds.
__attribute__((visibility("default"))) int bf_probe_block_copy(uint8_t* d_heap, const uint32_t* d_list, uint32_t n, uint32_t writeRows12, uint32_t grid, uint32_t reps, void* hip_stream, float* mean_us) {
    if (!(d_heap && d_list && mean_us && reps > 0 && grid > 0 && writeRows12 <= 12)) return 1;
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t e0, e1;
    PROBE_TRY(hipEventCreate(&e0));
    PROBE_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe_blocks, dim3(grid), dim3(256), 0, st, d_list, n, d_heap, writeRows12, 1u);       // warm-up
    PROBE_TRY(hipEventRecord(e0, st));
    for (uint32_t r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_blocks, dim3(grid), dim3(256), 0, st, d_list, n, d_heap, writeRows12, r + 2u);
    PROBE_TRY(hipEventRecord(e1, st));
    PROBE_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *mean_us = 1e3f * ms / (float)reps;
    return 0;
}

}  // extern "C"
