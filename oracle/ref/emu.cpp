// oracle/ref/emu.cpp — see emu.h (test infrastructure only)
#include "emu.h"


#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
enum State { READY, AT_BARRIER, AT_SHFL, DONE };
// Fiber switch: callee-saved registers + stack pointer (x86-64 System V).  ucontext's swapcontext saves and restores the signal mask with a
// system call on every switch, which made the switches - one or more per emulated thread - most of the emulator's run time.
extern "C" void emu_switch(void** saveSp, void* loadSp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

struct Fiber {
    void* sp = nullptr; char* stack = nullptr; State state = READY; uint3 tid;
    unsigned shflCount = 0;        // shuffles this fiber has deposited
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<unsigned> xchg;    // [warp][generation & 1][lane]: the values a warp's lanes deposited for a shuffle
    void* mainSp = nullptr;
    int cur = -1;
    const std::function<void()>* body = nullptr;
};
thread_local Block* g_block = nullptr;
const size_t STACK = 256 * 1024;
// fiber stacks are recycled across launches (a 512-thread block needs 128 MB of them; allocating and clearing that per launch was most
// of the run time of the solver pins); a nested launch takes other stacks from the same pool
thread_local std::vector<char*> g_freeStacks;
char* takeStack() {
    if (g_freeStacks.empty()) return (char*)malloc(STACK);
    char* p = g_freeStacks.back(); g_freeStacks.pop_back(); return p;
}
const unsigned WARP = 32;

void trampoline() {
    Block* b = g_block;
    Fiber& f = b->fibers[b->cur];
    (*b->body)();
    f.state = DONE;
    emu_switch(&f.sp, b->mainSp);
    abort();                                   // a finished fiber is never resumed
}
void prepare(Fiber& f) {                       // the first switch into the fiber "returns" into trampoline() with a call-aligned stack
    void** sp = (void**)(((uintptr_t)(f.stack + STACK)) & ~(uintptr_t)15);
    *--sp = nullptr;                           // the slot of trampoline's return address (it never returns)
    *--sp = (void*)&trampoline;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;      // rbp, rbx, r12-r15
    f.sp = sp;
}

// all live lanes of fiber i's warp have deposited the shuffle generation fiber i waits for.  `relaxed`: lanes parked at a block barrier
// count as inactive in this shuffle (they took the other side of a divergent branch and cannot arrive before the barrier opens, which
// needs the shuffling lanes) - used only when the block could not make progress otherwise.
bool warpReady(const Block& b, size_t i, bool relaxed = false) {
    const unsigned need = b.fibers[i].shflCount;                    // = generation + 1
    const size_t w0 = i / WARP * WARP, w1 = std::min(w0 + WARP, b.fibers.size());
    for (size_t j = w0; j < w1; ++j) {
        if (b.fibers[j].state == DONE || b.fibers[j].shflCount >= need) continue;
        if (relaxed && b.fibers[j].state == AT_BARRIER) continue;
        return false;
    }
    return true;
}

// lock-step exchange inside a warp: deposit, wait for the warp, read lane `src` (own value when src is outside the segment / has exited)
unsigned shflExchange(unsigned val, int src, int width) {
    Block* b = g_block;
    if (!b || b->cur < 0) return val;
    const size_t i = (size_t)b->cur;
    Fiber& f = b->fibers[i];
    const unsigned lane = (unsigned)(i % WARP), warp = (unsigned)(i / WARP), buf = f.shflCount & 1u;
    b->xchg[(warp * 2 + buf) * WARP + lane] = val;
    f.shflCount++;
    f.state = AT_SHFL;
    emu_switch(&f.sp, b->mainSp);
    threadIdx = f.tid;
    if (width <= 0 || width > (int)WARP) width = WARP;
    const int seg = (int)lane / width * width;
    if (src < seg || src >= seg + width) return val;
    const size_t j = (size_t)warp * WARP + (size_t)src;
    if (j >= b->fibers.size() || b->fibers[j].shflCount < f.shflCount) return val;      // the source lane has exited or is inactive in this shuffle
    return b->xchg[(warp * 2 + buf) * WARP + (unsigned)src];
}
unsigned f2u(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
float u2f(unsigned u) { float v; memcpy(&v, &u, 4); return v; }
int laneId() { return g_block && g_block->cur >= 0 ? (int)(g_block->cur % WARP) : 0; }
}  // namespace

void __syncthreads() {
    Block* b = g_block;
    if (!b || b->cur < 0) return;
    Fiber& f = b->fibers[b->cur];
    f.state = AT_BARRIER;
    emu_switch(&f.sp, b->mainSp);             // resumed when every live fiber of the block has arrived
    threadIdx = f.tid;
}

float __shfl_down(float v, int d, int w) { return u2f(shflExchange(f2u(v), laneId() + d, w)); }
float __shfl_xor(float v, int m, int w) { return u2f(shflExchange(f2u(v), laneId() ^ m, w)); }
float __shfl(float v, int s, int w) { const int ww = (w <= 0 || w > (int)WARP) ? (int)WARP : w; return u2f(shflExchange(f2u(v), laneId() / ww * ww + s % ww, w)); }
int __shfl_down(int v, int d, int w) { return (int)shflExchange((unsigned)v, laneId() + d, w); }
int __shfl_xor(int v, int m, int w) { return (int)shflExchange((unsigned)v, laneId() ^ m, w); }

namespace emu {

static thread_local const char* g_name = "?";
void named(const char* kernel) { g_name = kernel; }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    static const bool trace = getenv("EMU_TRACE") != nullptr;
    if (trace) fprintf(stderr, "emu: %s grid (%u,%u,%u) block (%u,%u,%u)\n", g_name, grid.x, grid.y, grid.z, block.x, block.y, block.z);
    const size_t T = (size_t)block.x * block.y * block.z;
    Block b;
    b.body = &body;
    b.fibers.resize(T);
    b.xchg.assign(((T + WARP - 1) / WARP) * 2 * WARP, 0u);
    for (auto& f : b.fibers) f.stack = takeStack();
    struct Release { Block& b; ~Release() { for (auto& f : b.fibers) g_freeStacks.push_back(f.stack); } } release{b};
    const dim3 gridSave = gridDim, blockSave = blockDim;
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
                            f.state = READY; f.shflCount = 0; f.tid = make_uint3(tx, ty, tz);
                            prepare(f);
                        }
                for (;;) {                  // run every runnable fiber up to its next barrier / shuffle (or to its end), in thread-index order
                    bool progress = false;
                    size_t live = 0, atBarrier = 0;
                    for (size_t i = 0; i < T; ++i) {
                        Fiber& f = b.fibers[i];
                        if (f.state == DONE) continue;
                        ++live;
                        if (f.state == AT_BARRIER) { ++atBarrier; continue; }
                        if (f.state == AT_SHFL && !warpReady(b, i)) continue;
                        f.state = READY;
                        b.cur = (int)i; threadIdx = f.tid;
                        emu_switch(&b.mainSp, f.sp);
                        progress = true;
                        if (f.state == AT_BARRIER) ++atBarrier;
                    }
                    if (live == 0) break;
                    size_t stillLive = 0, stillBarrier = 0;
                    for (auto& f : b.fibers) { if (f.state != DONE) ++stillLive; if (f.state == AT_BARRIER) ++stillBarrier; }
                    if (stillLive == 0) break;
                    if (stillBarrier == stillLive) { for (auto& f : b.fibers) if (f.state == AT_BARRIER) f.state = READY; progress = true; }
                    if (!progress) {        // divergent shuffle: release the waiting lanes whose missing partners are parked at the barrier
                        for (size_t i = 0; i < T; ++i) {
                            Fiber& f = b.fibers[i];
                            if (f.state != AT_SHFL || !warpReady(b, i, true)) continue;
                            f.state = READY;
                            b.cur = (int)i; threadIdx = f.tid;
                            emu_switch(&b.mainSp, f.sp);
                            progress = true;
                        }
                    }
                    if (!progress) throw std::runtime_error(std::string("emu: block cannot make progress (divergent barrier / shuffle) in ") + g_name);
                }
                b.cur = -1;
            }
    g_block = outer;
    gridDim = gridSave; blockDim = blockSave;
}

}  // namespace emu
