// oracle/ref/emu.cpp — see emu.h (test infrastructure only)
#include "emu.h"

#include <ucontext.h>

#include <stdexcept>
#include <vector>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
struct Fiber { ucontext_t ctx; std::vector<char> stack; bool done = false; uint3 tid; };
struct Block {
    std::vector<Fiber> fibers;
    ucontext_t main;
    int cur = -1;
    const std::function<void()>* body = nullptr;
};
thread_local Block* g_block = nullptr;
const size_t STACK = 256 * 1024;

void trampoline() {
    Block* b = g_block;
    Fiber& f = b->fibers[b->cur];
    (*b->body)();
    f.done = true;
    swapcontext(&f.ctx, &b->main);
}
}  // namespace

void __syncthreads() {
    Block* b = g_block;
    if (!b || b->cur < 0) return;
    Fiber& f = b->fibers[b->cur];
    swapcontext(&f.ctx, &b->main);            // resumed when every live fiber of the block has arrived
    threadIdx = f.tid;
}

float __shfl_down(float, int, int) { throw std::runtime_error("warp shuffles are not emulated"); }
float __shfl_xor(float, int, int) { throw std::runtime_error("warp shuffles are not emulated"); }
float __shfl(float, int, int) { throw std::runtime_error("warp shuffles are not emulated"); }
int __shfl_down(int, int, int) { throw std::runtime_error("warp shuffles are not emulated"); }
int __shfl_xor(int, int, int) { throw std::runtime_error("warp shuffles are not emulated"); }

namespace emu {

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const size_t T = (size_t)block.x * block.y * block.z;
    Block b;
    b.body = &body;
    b.fibers.resize(T);
    for (auto& f : b.fibers) f.stack.resize(STACK);
    gridDim = grid; blockDim = block;
    Block* outer = g_block;
    g_block = &b;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = make_uint3(bx, by, bz);
                size_t t = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
                            Fiber& f = b.fibers[t];
                            f.done = false; f.tid = make_uint3(tx, ty, tz);
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack.data(); f.ctx.uc_stack.ss_size = f.stack.size(); f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, trampoline, 0);
                        }
                size_t live = T;
                while (live) {              // one pass = run every live fiber up to its next barrier (or to its end)
                    live = 0;
                    for (size_t i = 0; i < T; ++i) {
                        Fiber& f = b.fibers[i];
                        if (f.done) continue;
                        b.cur = (int)i; threadIdx = f.tid;
                        swapcontext(&b.main, &f.ctx);
                        if (!f.done) ++live;
                    }
                }
                b.cur = -1;
            }
    g_block = outer;
}

}  // namespace emu
