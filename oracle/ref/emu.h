// oracle/ref/emu.h — TEST INFRASTRUCTURE ONLY.
// Serial emulation of a CUDA kernel launch on the host, for running the reference's own __global__ functions (compiled by
// g++ through shim/cuda_runtime.h) as the parity pin of the CPU oracle.  Blocks run one after the other; the threads of a
// block are fibers (own stacks, a register / stack-pointer switch in emu.cpp; x86-64) executed in thread-index order, and __syncthreads() suspends a fiber until every live fiber of
// the block has reached a barrier — so kernels with __shared__ data and barriers keep their semantics, while atomics and
// "first thread wins" races resolve in a fixed (thread-index) order.
#ifndef BF_REF_EMU_H
#define BF_REF_EMU_H
#include <functional>

#include "cuda_runtime.h"

namespace emu {
void named(const char* kernel);      // name of the kernel the next launch runs (error messages; EMU_TRACE=1 prints every launch)
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
inline void launch(dim3 grid, dim3 block, size_t /*sharedBytes*/, const std::function<void()>& body) { launch(grid, block, body); }
inline void launch(dim3 grid, dim3 block, size_t, cudaStream_t, const std::function<void()>& body) { launch(grid, block, body); }
}  // namespace emu

#endif
