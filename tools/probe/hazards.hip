// Minimal reproducers for the three mechanisms round 5 offered for run-to-run differences.  Each test runs ALONE and beside noise
// kernels on other streams (a streaming copy, an LDS / barrier kernel at high wave priority, many short launches) and reports failures per 1e6 waves / reads:
//   pk     a packed-FP32 instruction with a broadcast first source (op_sel_hi:[0,1]) whose result the NEXT instruction reads (packed add, v_rcp_f32, v_add_f32),
//          written in assembly exactly as the compiler emitted it in k_update_batch_apx (no s_nop in between), against the same arithmetic with s_nop 4 in between
//   sload  records in the kernel arguments read at a RUN-TIME index with wide scalar loads (104-byte stride: s_load_dwordx8 straddles 64-byte lines; 128-byte
//          stride: none does), every launch with fresh contents, the first dependent vector instructions checked lane by lane
//   handoff last-arriver finisher: producers store results (plain | agent-scope write-through), drain (s_waitcnt vmcnt(0)), take a ticket; the last workgroup
//          acquires at agent scope and reads every producer's results (plain | agent-scope loads); the buffers are re-used launch after launch with new values
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/hazards.hip -o tools/probe/bf_hazards     Run: tools/probe/bf_hazards [seconds per test]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ noise
__global__ void k_noise_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_noise_lds(float* out, int iters) {
    __shared__ float buf[16384];
    __builtin_amdgcn_s_setprio(3);
    float acc = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        buf[(threadIdx.x * 33 + i) & 16383] = acc;
        __syncthreads();
        acc = acc * 1.0001f + buf[(threadIdx.x * 17 + i * 5) & 16383] + __builtin_amdgcn_rcpf(acc + 2.0f);
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_noise_short(float* out) { if (threadIdx.x == 1000) out[0] = 1.0f; }

struct Noise {
    hipStream_t s[3]; uint4 *a, *b; float* o; size_t n = (size_t)64 << 20;      // 1 GiB each way
    void init() {
        int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s[2], hipStreamNonBlocking, hi));
        CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, 64)); CK(hipMemset(a, 1, n * 16));
    }
    void kick() {          // a few hundred microseconds of neighbours on three queues
        hipLaunchKernelGGL(k_noise_copy, dim3(2048), dim3(256), 0, s[0], a, b, n / 4);
        hipLaunchKernelGGL(k_noise_lds, dim3(512), dim3(256), 0, s[1], o, 200);
        for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_noise_short, dim3(64), dim3(256), 0, s[2], o);
    }
    void drain() { for (auto q : s) CK(hipStreamSynchronize(q)); }
};

// ------------------------------------------------------------------------------------------------ pk
// per lane: x (pair), y (pair), scalar pair c; result  e = c.lo + (x.lo + y)  per half, r = rcp(e.lo), t = e.lo + e.hi - the chain of apxProject's first pair
struct PkOut { uint32_t bad[3][4]; uint32_t waves; uint32_t pad[3]; };      // [consumer: packed add, rcp, plain add][lane quarter]
template <bool PAD>
__global__ __launch_bounds__(256) void k_pk(const float* __restrict__ in, float c0, float c1, int iters, PkOut* out) {
    const uint32_t lane = threadIdx.x & 63u, gid = blockIdx.x * blockDim.x + threadIdx.x;
    v2f x, y; x.x = in[(gid * 4 + 0) & 0xFFFFF]; x.y = in[(gid * 4 + 1) & 0xFFFFF]; y.x = in[(gid * 4 + 2) & 0xFFFFF]; y.y = in[(gid * 4 + 3) & 0xFFFFF];
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    for (int it = 0; it < iters; ++it) {
        float ex_, ey_, r, t;
        v2f cc; cc.x = c0; cc.y = c1;
        // fixed registers: the instruction sequence is exactly what is written here (v104..v107 primed with values a stale read would deliver)
#define PK_SEQ(PADS) \
        asm volatile("v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\t" \
                     "v_mov_b32 v104, -1.0\n\tv_mov_b32 v105, -2.0\n\tv_mov_b32 v106, -4.0\n\tv_mov_b32 v107, 0.5\n\tv_mov_b32 v108, 2.0\n\ts_nop 4\n\t" \
                     "v_pk_add_f32 v[104:105], v[100:101], v[102:103] op_sel_hi:[0,1]\n\t" PADS \
                     "v_pk_add_f32 v[106:107], %8, v[104:105] op_sel_hi:[0,1]\n\t" PADS \
                     "v_rcp_f32 v108, v106\n\ts_nop 4\n\t" \
                     "v_mov_b32 v110, -1.0\n\tv_mov_b32 v111, -2.0\n\tv_mov_b32 v109, 4.0\n\ts_nop 4\n\t" \
                     "v_pk_add_f32 v[110:111], v[100:101], v[102:103] op_sel_hi:[0,1]\n\t" PADS \
                     "v_add_f32 v109, v110, v111\n\ts_nop 4\n\t" \
                     "v_mov_b32 %0, v106\n\tv_mov_b32 %1, v107\n\tv_mov_b32 %2, v108\n\tv_mov_b32 %3, v109\n\ts_nop 1" \
                     : "=&v"(ex_), "=&v"(ey_), "=&v"(r), "=&v"(t) : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(cc) \
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111")
        if (PAD) { PK_SEQ("s_nop 4\n\t"); } else { PK_SEQ(""); }
        v2f e; e.x = ex_; e.y = ey_;
        const float dx = x.x + y.x, dy = x.x + y.y;          // op_sel_hi:[0,1]: both halves take the LOW half of the first source
        const float ex = c0 + dx, ey = c0 + dy;
        if (e.x != ex || e.y != ey) b0++;
        if (r != __builtin_amdgcn_rcpf(ex)) b1++;
        if (t != dx + dy) b2++;
        x.x += 0.25f; y.y -= 0.125f;
    }
    if (b0) atomicAdd(&out->bad[0][lane >> 4], b0);
    if (b1) atomicAdd(&out->bad[1][lane >> 4], b1);
    if (b2) atomicAdd(&out->bad[2][lane >> 4], b2);
    if (lane == 0) atomicAdd(&out->waves, 1u);
}

// ------------------------------------------------------------------------------------------------ window
// The first voxel pair's projection of k_update_batch_apx as the packed-FP32 build has it (tools/probe/window_gen.h: the compiler's instructions and register
// numbers), run in three paddings on the same inputs; outputs: the two texel offsets (v56, v53) and the two camera-space depths (v64, v65).
#include "window_gen.h"
struct WinOut { uint32_t bad[2][4][4]; uint32_t waves; uint32_t first[7]; };      // [RAW | PADPK vs PADALL][output][lane quarter]
#define WIN_SETUP \
    "v_mov_b32 v44, %4\n\tv_mov_b32 v45, %4\n\tv_mov_b32 v32, %5\n\tv_mov_b32 v33, %5\n\tv_mov_b32 v24, %6\n\tv_add_f32 v25, 1.0, %6\n\t" \
    "v_mul_f32 v34, 0x3b83126f, %4\n\tv_mul_f32 v35, 0x3b83126f, %5\n\tv_mul_f32 v36, 0x3b83126f, v24\n\tv_mul_f32 v37, 0x3b83126f, v25\n\t" \
    "s_mov_b32 s40, 0xbd23d70a\n\ts_mov_b32 s41, 0x3ca3d70a\n\ts_mov_b32 s42, 0x3f7fbe77\n\ts_mov_b32 s43, 0x3f000000\n\t" \
    "s_mov_b32 s4, 0x40151eb8\n\ts_mov_b32 s5, 0x3c23d70a\n\ts_mov_b32 s6, 0x3dcccccd\n\ts_mov_b32 s7, 0x42693333\n\t" \
    "s_mov_b32 s8, 0x3c23d70a\n\ts_mov_b32 s9, 0x40151eb8\n\ts_mov_b32 s10, 0x3d4ccccd\n\ts_mov_b32 s11, 0xc2e93333\n\t" \
    "s_mov_b32 s56, 0x43a00000\n\ts_mov_b32 s57, 0x43a00000\n\ts_mov_b32 s24, 0x43700000\n\ts_mov_b32 s25, 0x43700000\n\t" \
    "s_mov_b32 s31, 640\n\ts_mov_b32 s20, 480\n\ts_mov_b32 s21, 640\n\t" \
    "v_mov_b32 v52, %7\n\tv_mov_b32 v53, %7\n\tv_mov_b32 v54, %7\n\tv_mov_b32 v55, %7\n\tv_mov_b32 v56, %7\n\tv_mov_b32 v57, %7\n\tv_mov_b32 v58, %7\n\tv_mov_b32 v59, %7\n\t" \
    "v_mov_b32 v60, %7\n\tv_mov_b32 v61, %7\n\tv_mov_b32 v64, %7\n\tv_mov_b32 v65, %7\n\t" \
    "s_mov_b32 s2, 3\n\ts_cmp_lg_u32 s2, 1\n\ts_nop 4\n\t"
#define WIN_RUN(WINDOW, o0, o1, o2, o3) \
    asm volatile(WIN_SETUP WINDOW "s_nop 4\n\tv_mov_b32 %0, v56\n\tv_mov_b32 %1, v53\n\tv_mov_b32 %2, v64\n\tv_mov_b32 %3, v65\n\ts_nop 1" \
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(ix), "v"(iy), "v"(iz), "v"(junk) \
                 : "v24", "v25", "v32", "v33", "v34", "v35", "v36", "v37", "v44", "v45", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v64", "v65", \
                   "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s20", "s21", "s24", "s25", "s31", "s40", "s41", "s42", "s43", "s56", "s57", "vcc", "scc")
__global__ __launch_bounds__(256) void k_window(int iters, WinOut* out, uint32_t salt) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t bad[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float junk = 1.0f + (float)lane;
    for (int it = 0; it < iters; ++it) {
        const float ix = (float)(8 * ((int)((wave * 7u + (uint32_t)it * 3u + salt) % 60u) - 30) + (int)(lane & 7u));
        const float iy = (float)(8 * ((int)((wave * 5u + (uint32_t)it * 11u + salt) % 40u) - 20) + (int)(lane >> 3));
        const float iz = (float)(8 * (50 + (it % 10)));
        uint32_t r[4], k[4], a[4];
        WIN_RUN(WINDOW_RAW, r[0], r[1], r[2], r[3]);
        WIN_RUN(WINDOW_PADPK, k[0], k[1], k[2], k[3]);
        WIN_RUN(WINDOW_PADALL, a[0], a[1], a[2], a[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (r[q] != a[q]) { bad[0][q]++; if (atomicAdd(&out->first[0], 1u) == 0u) { out->first[1] = (uint32_t)q; out->first[2] = lane; out->first[3] = r[q]; out->first[4] = a[q]; out->first[5] = (uint32_t)it; out->first[6] = wave; } }
            if (k[q] != a[q]) bad[1][q]++;
        }
        junk = __uint_as_float((a[2] & 0x007FFFFFu) | 0x3F800000u) + (float)it;
    }
    for (int v = 0; v < 2; ++v) for (int q = 0; q < 4; ++q) if (bad[v][q]) atomicAdd(&out->bad[v][q][lane >> 4], bad[v][q]);
    if (lane == 0) atomicAdd(&out->waves, 1u);
}

// ------------------------------------------------------------------------------------------------ opsel
// What the in-situ bisection (profiles/r06_determinism.md) left: v_pk_fma_f32 with broadcast modifiers (op_sel / op_sel_hi) on a scalar-register first source and
// on the vector addend - the forms the compiler emits for the FIRST voxel pair of an operator slot, before it has materialised the broadcast pairs.  Each wave
// alternates the four forms below (registers as in the kernel) with the voxel update's memory phase (eight 12-byte-per-lane loads, six stores, four 8-byte
// gathers), so that co-resident waves issue vector-memory instructions beside the packed FMAs.  Every result half is checked against v_fma_f32.
struct OsOut { uint32_t bad[4][2][4]; uint32_t waves; uint32_t first[6]; uint32_t pad; };      // [form][half][lane quarter]
__global__ __launch_bounds__(256) void k_opsel(int iters, float* __restrict__ vox, const uint2* __restrict__ tex, uint32_t texN, OsOut* out, uint32_t salt, int memPhase) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    float* base = vox + (size_t)wave * 1536u + lane * 3u;
    uint32_t bad[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    float carry = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const float iz0 = (float)(8 * (50 + ((it + (int)salt) % 10))), iz1 = iz0 + 1.0f;
        const float n0 = 2.33f * (float)((int)(lane & 7u) - 3 + (int)(wave % 50u)) + 58.3f + carry * 0.0f, n1 = 2.33f * (float)((int)(lane >> 3) - 3 + (int)(wave % 37u)) - 116.6f;
        const float cx = 0.1f + 0.001f * (float)(it & 7), cy = 0.05f + 0.001f * (float)(it & 3);
        float r[4][2];
        v2f sx; sx.x = cx; sx.y = 123.0f; v2f sy; sy.x = cy; sy.y = -77.0f;
        asm volatile("v_mov_b32 v24, %8\n\tv_mov_b32 v25, %9\n\tv_mov_b32 v54, %10\n\tv_mov_b32 v55, %11\n\t"
                     "v_readfirstlane_b32 s6, %12\n\ts_mov_b32 s7, 0x42f60000\n\tv_readfirstlane_b32 s10, %13\n\ts_mov_b32 s11, 0xc29a0000\n\ts_nop 4\n\t"
                     "v_mov_b32 v62, s6\n\tv_mov_b32 v63, s6\n\tv_mov_b32 v66, v54\n\tv_mov_b32 v67, v54\n\t"
                     "v_mov_b32 v58, -1.0\n\tv_mov_b32 v59, -1.0\n\tv_mov_b32 v60, -1.0\n\tv_mov_b32 v61, -1.0\n\tv_mov_b32 v68, -1.0\n\tv_mov_b32 v69, -1.0\n\tv_mov_b32 v70, -1.0\n\tv_mov_b32 v71, -1.0\n\t"
                     "v_pk_fma_f32 v[58:59], s[6:7], v[24:25], v[54:55] op_sel_hi:[0,1,0]\n\t"
                     "v_pk_fma_f32 v[60:61], s[10:11], v[24:25], v[54:55] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[68:69], v[62:63], v[24:25], v[54:55] op_sel_hi:[0,1,0]\n\t"
                     "v_pk_fma_f32 v[70:71], v[62:63], v[24:25], v[66:67]\n\t"
                     "s_nop 4\n\t"
                     "v_mov_b32 %0, v58\n\tv_mov_b32 %1, v59\n\tv_mov_b32 %2, v60\n\tv_mov_b32 %3, v61\n\tv_mov_b32 %4, v68\n\tv_mov_b32 %5, v69\n\tv_mov_b32 %6, v70\n\tv_mov_b32 %7, v71\n\ts_nop 1"
                     : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[3][0]), "=&v"(r[3][1])
                     : "v"(iz0), "v"(iz1), "v"(n0), "v"(n1), "v"(cx), "v"(cy)
                     : "v24", "v25", "v54", "v55", "v58", "v59", "v60", "v61", "v62", "v63", "v66", "v67", "v68", "v69", "v70", "v71", "s6", "s7", "s10", "s11");
        const float e[4][2] = {{__builtin_fmaf(cx, iz0, n0), __builtin_fmaf(cx, iz1, n0)}, {__builtin_fmaf(cy, iz0, n1), __builtin_fmaf(cy, iz1, n1)},
                               {__builtin_fmaf(cx, iz0, n0), __builtin_fmaf(cx, iz1, n0)}, {__builtin_fmaf(cx, iz0, n0), __builtin_fmaf(cx, iz1, n0)}};
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (r[f][h] != e[f][h]) { bad[f][h]++; if (atomicAdd(&out->first[0], 1u) == 0u) { out->first[1] = (uint32_t)(f * 2 + h); out->first[2] = lane; out->first[3] = __float_as_uint(r[f][h]); out->first[4] = __float_as_uint(e[f][h]); out->first[5] = wave; } }
        if (memPhase) {          // the voxel update's memory phase on this wave's own 6 KB
            float v[8][3];
#pragma unroll
            for (int z = 0; z < 8; ++z) { v[z][0] = base[z * 192 + 0]; v[z][1] = base[z * 192 + 1]; v[z][2] = base[z * 192 + 2]; }
            uint2 t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) t[g] = tex[(wave * 977u + lane * 131u + (uint32_t)it * 7919u + (uint32_t)g * 40503u) % texN];
#pragma unroll
            for (int z = 0; z < 6; ++z) { base[z * 192 + 0] = v[z][0] + __uint_as_float(t[z & 3].x) * 0.0f; base[z * 192 + 1] = v[z][1] + 1.0f; base[z * 192 + 2] = v[z][2]; }
            carry = v[7][0];
        }
    }
    for (int f = 0; f < 4; ++f) for (int h = 0; h < 2; ++h) if (bad[f][h]) atomicAdd(&out->bad[f][h][lane >> 4], bad[f][h]);
    if (lane == 0) atomicAdd(&out->waves, 1u);
}

// ------------------------------------------------------------------------------------------------ sload
template <int WORDS> struct RecT { float f[WORDS]; };
template <int WORDS> struct ArgsT { RecT<WORDS> op[12]; uint32_t salt; };
struct SlOut { uint32_t bad[4]; uint32_t checks; uint32_t firstBad[3]; };
__host__ __device__ inline float recValue(uint32_t salt, uint32_t k, uint32_t j) { return (float)((salt * 2654435761u + k * 97u + j * 13u) & 0xFFFFu) * 0.5f + 1.0f; }
template <int WORDS>
__global__ __launch_bounds__(256) void k_sload(ArgsT<WORDS> a, const uint32_t* __restrict__ order, uint32_t nOrder, SlOut* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t bad = 0;
    float acc = 0.0f;
    for (uint32_t i = 0; i < nOrder; ++i) {
        const uint32_t k = __builtin_amdgcn_readfirstlane(order[(blockIdx.x + i) % nOrder]);      // run-time, wave-uniform
        const RecT<WORDS>& o = a.op[k];
        // the first dependent vector instructions: one FMA chain over 24 of the record's words (as apxCol / apxProject use a pose)
        float s = (float)lane;
#pragma unroll
        for (int j = 0; j < 24; ++j) s = __builtin_fmaf(o.f[j], 0.5f, s);
        float e = (float)lane;
#pragma unroll
        for (int j = 0; j < 24; ++j) e = __builtin_fmaf(recValue(a.salt, k, (uint32_t)j), 0.5f, e);
        if (s != e) { bad++; if (atomicAdd(&out->firstBad[0], 1u) == 0u) { out->firstBad[1] = k; out->firstBad[2] = lane; } }
        acc += s;
    }
    if (bad) atomicAdd(&out->bad[lane >> 4], bad);
    if (lane == 0) atomicAdd(&out->checks, nOrder);
    if (acc == 1.2345f) out->checks = 0;
}

// ------------------------------------------------------------------------------------------------ handoff
struct HoOut { uint32_t stale; uint32_t reads; uint32_t launches; uint32_t pad; };
template <bool STORE_AGENT, bool LOAD_AGENT>
__global__ __launch_bounds__(256) void k_handoff(int* res, uint32_t perWg, uint32_t* ticket, uint32_t gen, HoOut* out) {
    __shared__ uint32_t lastFlag;
    int* mine = res + (size_t)blockIdx.x * perWg;
    for (uint32_t i = threadIdx.x; i < perWg; i += 256u) {
        const int v = (int)(gen * 131071u + blockIdx.x * 257u + i);
        if (STORE_AGENT) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else mine[i] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(ticket, 1u);
        lastFlag = t == gridDim.x - 1 ? 1u : 0u;
        if (lastFlag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!lastFlag) return;
    uint32_t stale = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x * perWg; i += 256u) {
        const int v = LOAD_AGENT ? __hip_atomic_load(res + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : res[i];
        if (v != (int)(gen * 131071u + (i / perWg) * 257u + (i % perWg))) stale++;
    }
    if (stale) atomicAdd(&out->stale, stale);
    if (threadIdx.x == 0) { atomicAdd(&out->reads, gridDim.x * perWg); atomicAdd(&out->launches, 1u); *ticket = 0u; }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 8.0;
    const bool onlyWindow = argc > 2 && std::string(argv[2]) == "window";
    Noise nz; nz.init();
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<float> h(1 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (float)((i * 2654435761u) & 0xFFFF) * (1.0f / 4096.0f);
    float* din; CK(hipMalloc(&din, h.size() * 4)); CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    void* dout; CK(hipMalloc(&dout, 256));          // (every test's counters fit 256 bytes)
    printf("{");
    for (int noise = 0; noise < 2; ++noise) {
        // ---- window
        {
            CK(hipMemset(dout, 0, 256));
            const double t0 = now(); uint32_t salt = 0;
            while (now() - t0 < secs) {
                if (noise) nz.kick();
                for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_window, dim3(8192), dim3(256), 0, st, 32, (WinOut*)dout, salt++);
                CK(hipStreamSynchronize(st)); if (noise) nz.drain();
            }
            WinOut o; CK(hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost));
            printf("\"window_%s\": {\"waves\": %u, \"windows_per_wave\": 32, ", noise ? "noise" : "alone", o.waves);
            static const char* on[4] = {"offsetA_v56", "offsetB_v53", "depthA_v64", "depthB_v65"};
            for (int v = 0; v < 2; ++v) for (int q = 0; q < 4; ++q) printf("\"%s_%s_bad_by_lane_quarter\": [%u,%u,%u,%u], ", v ? "padpk" : "raw", on[q], o.bad[v][q][0], o.bad[v][q][1], o.bad[v][q][2], o.bad[v][q][3]);
            printf("\"first_bad\": {\"count\": %u, \"output\": %u, \"lane\": %u, \"raw\": \"%08x\", \"padded\": \"%08x\", \"iteration\": %u, \"wave\": %u}}, ", o.first[0], o.first[1], o.first[2], o.first[3], o.first[4], o.first[5], o.first[6]);
            fflush(stdout);
        }
        // ---- opsel
        {
            float* vox; uint2* tex; const uint32_t texN = 640 * 480 * 12;
            CK(hipMalloc(&vox, (size_t)8192 * 4 * 1536 * 4)); CK(hipMemset(vox, 0, (size_t)8192 * 4 * 1536 * 4)); CK(hipMalloc(&tex, (size_t)texN * 8)); CK(hipMemset(tex, 0, (size_t)texN * 8));
            for (int mem = 0; mem < 2; ++mem) {
                void* d2; CK(hipMalloc(&d2, 512)); CK(hipMemset(d2, 0, 512));
                const double t0 = now(); uint32_t salt = 0;
                while (now() - t0 < secs) {
                    if (noise) nz.kick();
                    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_opsel, dim3(8192), dim3(256), 0, st, 64, vox, tex, texN, (OsOut*)d2, salt++, mem);
                    CK(hipStreamSynchronize(st)); if (noise) nz.drain();
                }
                OsOut o; CK(hipMemcpy(&o, d2, sizeof o, hipMemcpyDeviceToHost)); CK(hipFree(d2));
                printf("\"opsel_%s_%s\": {\"waves\": %u, \"sequences_per_wave\": 64, ", mem ? "with_memory_phase" : "arithmetic_only", noise ? "noise" : "alone", o.waves);
                static const char* fn[4] = {"sgpr_src0_bcast_vgpr_src2_bcast_lo", "sgpr_src0_bcast_vgpr_src2_bcast_hi", "vgpr_src0_pair_src2_bcast_lo", "no_modifiers"};
                for (int f = 0; f < 4; ++f) printf("\"%s\": {\"lo_bad_by_lane_quarter\": [%u,%u,%u,%u], \"hi_bad_by_lane_quarter\": [%u,%u,%u,%u]}, ", fn[f], o.bad[f][0][0], o.bad[f][0][1], o.bad[f][0][2], o.bad[f][0][3], o.bad[f][1][0], o.bad[f][1][1], o.bad[f][1][2], o.bad[f][1][3]);
                printf("\"first_bad\": {\"count\": %u, \"form_half\": %u, \"lane\": %u, \"got\": \"%08x\", \"expected\": \"%08x\", \"wave\": %u}}, ", o.first[0], o.first[1], o.first[2], o.first[3], o.first[4], o.first[5]);
                fflush(stdout);
            }
            CK(hipFree(vox)); CK(hipFree(tex));
        }
        if (onlyWindow) { if (noise) printf("\"only\": \"window\""); continue; }
        // ---- pk
        for (int pad = 0; pad < 2; ++pad) {
            CK(hipMemset(dout, 0, 256));
            const double t0 = now(); uint64_t launches = 0;
            while (now() - t0 < secs) {
                if (noise) nz.kick();
                for (int r = 0; r < 4; ++r) {
                    if (pad) hipLaunchKernelGGL(k_pk<true>, dim3(8192), dim3(256), 0, st, din, 1.5f, 2.5f, 64, (PkOut*)dout);
                    else hipLaunchKernelGGL(k_pk<false>, dim3(8192), dim3(256), 0, st, din, 1.5f, 2.5f, 64, (PkOut*)dout);
                    ++launches;
                }
                CK(hipStreamSynchronize(st)); if (noise) nz.drain();
            }
            PkOut o; CK(hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost));
            printf("\"pk_%s_%s\": {\"waves\": %u, \"sequences_per_wave\": 64, \"bad_packed_consumer_by_lane_quarter\": [%u,%u,%u,%u], \"bad_rcp_consumer\": [%u,%u,%u,%u], \"bad_plain_add_consumer\": [%u,%u,%u,%u]}, ",
                   pad ? "padded" : "unpadded", noise ? "noise" : "alone", o.waves, o.bad[0][0], o.bad[0][1], o.bad[0][2], o.bad[0][3], o.bad[1][0], o.bad[1][1], o.bad[1][2], o.bad[1][3], o.bad[2][0], o.bad[2][1], o.bad[2][2], o.bad[2][3]);
            fflush(stdout);
        }
        // ---- sload: 104-byte records (26 words) and 128-byte records (32 words)
        uint32_t hOrder[64]; for (int i = 0; i < 64; ++i) hOrder[i] = (uint32_t)((i * 7 + 3) % 12);
        uint32_t* dOrder; CK(hipMalloc(&dOrder, sizeof hOrder)); CK(hipMemcpy(dOrder, hOrder, sizeof hOrder, hipMemcpyHostToDevice));
        for (int wide = 0; wide < 2; ++wide) {
            CK(hipMemset(dout, 0, 256));
            const double t0 = now(); uint32_t salt = 1;
            while (now() - t0 < secs) {
                if (noise) nz.kick();
                for (int r = 0; r < 32; ++r, ++salt) {          // every launch its own contents: a stale line of an earlier launch's arguments shows
                    if (wide) { ArgsT<32> a; for (uint32_t k = 0; k < 12; ++k) for (uint32_t j = 0; j < 32; ++j) a.op[k].f[j] = recValue(salt, k, j); a.salt = salt;
                                hipLaunchKernelGGL(k_sload<32>, dim3(2048), dim3(256), 0, st, a, dOrder, 64u, (SlOut*)dout); }
                    else { ArgsT<26> a; for (uint32_t k = 0; k < 12; ++k) for (uint32_t j = 0; j < 26; ++j) a.op[k].f[j] = recValue(salt, k, j); a.salt = salt;
                           hipLaunchKernelGGL(k_sload<26>, dim3(2048), dim3(256), 0, st, a, dOrder, 64u, (SlOut*)dout); }
                }
                CK(hipStreamSynchronize(st)); if (noise) nz.drain();
            }
            SlOut o; CK(hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost));
            printf("\"sload_%s_%s\": {\"launches\": %u, \"record_reads_per_wave\": 64, \"wave_record_reads\": %u, \"bad_by_lane_quarter\": [%u,%u,%u,%u], \"first_bad_k_lane\": [%u,%u]}, ",
                   wide ? "128B" : "104B_straddling", noise ? "noise" : "alone", salt - 1, o.checks, o.bad[0], o.bad[1], o.bad[2], o.bad[3], o.firstBad[1], o.firstBad[2]);
            fflush(stdout);
        }
        // ---- handoff: 32 producers x 1024 ints (k_match's shape), four store / load combinations
        int* res; uint32_t* ticket; CK(hipMalloc(&res, 32 * 1024 * 4)); CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4)); CK(hipMemset(res, 0, 32 * 1024 * 4));
        for (int combo = 0; combo < 4; ++combo) {
            CK(hipMemset(dout, 0, 256));
            const double t0 = now(); uint32_t gen = 1;
            while (now() - t0 < secs) {
                if (noise) nz.kick();
                for (int r = 0; r < 64; ++r, ++gen) {
                    switch (combo) {
                        case 0: hipLaunchKernelGGL((k_handoff<false, false>), dim3(32), dim3(256), 0, st, res, 1024u, ticket, gen, (HoOut*)dout); break;
                        case 1: hipLaunchKernelGGL((k_handoff<true, false>), dim3(32), dim3(256), 0, st, res, 1024u, ticket, gen, (HoOut*)dout); break;
                        case 2: hipLaunchKernelGGL((k_handoff<false, true>), dim3(32), dim3(256), 0, st, res, 1024u, ticket, gen, (HoOut*)dout); break;
                        default: hipLaunchKernelGGL((k_handoff<true, true>), dim3(32), dim3(256), 0, st, res, 1024u, ticket, gen, (HoOut*)dout); break;
                    }
                }
                CK(hipStreamSynchronize(st)); if (noise) nz.drain();
            }
            HoOut o; CK(hipMemcpy(&o, dout, sizeof o, hipMemcpyDeviceToHost));
            static const char* names[4] = {"plain_store_plain_load", "agent_store_plain_load", "plain_store_agent_load", "agent_store_agent_load"};
            printf("\"handoff_%s_%s\": {\"launches\": %u, \"reads\": %u, \"stale\": %u}%s", names[combo], noise ? "noise" : "alone", o.launches, o.reads, o.stale, (noise && combo == 3) ? "" : ", ");
            fflush(stdout);
        }
        CK(hipFree(res)); CK(hipFree(ticket)); CK(hipFree(dOrder));
    }
    printf("}\n");
    return 0;
}
