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
    Noise nz; nz.init();
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<float> h(1 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (float)((i * 2654435761u) & 0xFFFF) * (1.0f / 4096.0f);
    float* din; CK(hipMalloc(&din, h.size() * 4)); CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    void* dout; CK(hipMalloc(&dout, 256));
    printf("{");
    for (int noise = 0; noise < 2; ++noise) {
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
