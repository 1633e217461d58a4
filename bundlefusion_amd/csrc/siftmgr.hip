// Keypoint / match store, descriptor matcher and the match filters for gfx950 (== class SIFTImageManager +
// SiftMatchGPU of the reference).  Reference semantics: SiftGPU/ProgramCU.cu:1634-1936 (dot products, row and
// column best-2, mutual check), SiftGPU/SiftMatch.cpp:160-196, SiftGPU/SIFTImageManager.cu:59-143 (sort),
// :186-263 + cuda_kabsch.h:73-502 (greedy Kabsch filter), :318-389 + cuda_surfaceArea.h (surface area),
// :418-585 (dense verify), :610-658 (add residuals), :692-774 (invalidate), :1036-1159 (verify trajectory),
// SIFTImageManager.cpp:551-575 (filterFrames).
//
// MI355X design (not the reference's kernel structure):
//  * keypoints/descriptors live at a fixed stride (image * maxKeysPerImage + k) and the per-image key count stays on
//    the device, so SIFT -> match -> filter -> add-residuals runs without a host read-back; one small D2H per frame
//    returns (lastMatchedFrame, #residuals).
//  * matching: ONE launch per frame, one 1024-thread workgroup per previous image.  The u8 x u8 dot products run on
//    the matrix cores (v_mfma_i32_16x16x64_i8 on bytes biased by 128; the bias is undone exactly with per-key byte
//    sums); the dot matrix is never written to memory — row and column best/second/argmax are folded on the fly
//    (waves own row chunks and rotate over column tiles so no two waves touch the same column state), then the
//    mutual check, the ordered compaction and the stable distance sort finish inside the same kernel.
//  * everything order-dependent in the reference (atomic appends, float atomics) is replaced by a fixed order, so
//    results are bit-reproducible: matches appended in ascending current-key order, residuals in ascending
//    previous-image order, dense-verify sums over 256 strided partials + xor butterflies + 4 wave totals in order.
#include <algorithm>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/bf_detmath.h"
#include "bf_device.h"
#include "bf_internal.h"

using namespace bf;

namespace {

constexpr int MAX_RAW = 128;     // MAX_MATCHES_PER_IMAGE_PAIR_RAW       GlobalDefines.h:8
constexpr int MAX_FILT = 25;     // MAX_MATCHES_PER_IMAGE_PAIR_FILTERED  GlobalDefines.h:9
constexpr uint64_t P_NONE = 0x00000000FFFFFFFFull;

typedef int v4i __attribute__((ext_vector_type(4)));

struct Key { float x, y, scale, depth; };

struct FrameResult { int lastMatched; int valid; int numResiduals; int numKeysCur; };

// ------------------------------------------------------------------------------------------------ matching
struct MatchArgs {
    const uint8_t* descs; const int* numKeys; const int* validImages; uint32_t maxKeys;
    uint32_t curFrame, startFrame;
    float distmax, ratiomax;
    int* numMatches; float* dist; uint2* idx;
    int speculative;        // 1: pairs with an invalid previous image are matched too (bf_siftmgr_commit_pairs clears them once the flags are final)
};

BF_DEV void top2_merge(uint64_t& P, int& n, uint64_t P2, int n2) {
    const uint64_t lo = P < P2 ? P : P2;
    P = P < P2 ? P2 : P;
    n = max(max(n, n2), (int)(lo >> 32));
}
BF_DEV void top2_add(uint64_t& P, int& n, uint64_t Pe, int d) {
    if (Pe > P) { n = max(n, (int)(P >> 32)); P = Pe; }
    else n = max(n, d);
}
BF_DEV uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
BF_DEV int byte_sum16(const v4i v) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s = (int)__builtin_amdgcn_sad_u8((uint32_t)v[i], 0u, (uint32_t)s);
    return s;
}
BF_DEV int decode_idx(uint64_t P) {
    const uint32_t lo = (uint32_t)P;
    return lo == 0xFFFFFFFFu ? -1 : (int)((0xFFFFFFFEu - lo) & 0xFFFFu);
}

// One image pair = 32 workgroups of four waves (round 5; rounds 1-4 ran a pair in ONE 1024-thread workgroup: 170 us per frame with ten pairs, 430 us in the frame
// that closes a chunk - ten workgroups on a 256-CU device, VALU-bound on the best-2 bookkeeping - and the longest link of the chain the frame loop waits for).
//   workgroups 0..15   ROW slabs: 64 keys of the previous image against ALL keys of the current one -> per row best / second-best -> RowMatch_Kernel's verdict
//   workgroups 16..31  COLUMN slabs: 64 keys of the current image against all keys of the previous one -> ColMatch_Kernel's verdict
//                      (the 128-byte dot products are computed twice; the matrix cores idle either way)
//   the LAST workgroup of the pair to finish (ticket) runs the mutual check, the ordered compaction and the stable distance sort
// A wave owns 16 keys of its side (one MFMA row chunk) and walks the other side's 16-key tiles: its best-2 state is complete in its own registers - nothing is
// merged across waves or workgroups.  best = max over the packed (dot product, tie-break key) words and second-best = max over the rest are order-independent
// (top2_add / top2_merge), so the results are those of the single-workgroup form bit for bit.
struct MatchScratch { int* rowRes; float* rowDist; int* colRes; uint32_t* ticket; };      // per previous image: 1024 / 1024 / 1024 / 1

BF_DEV void storeAgent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BF_DEV void storeAgentF(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BF_DEV int loadAgent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BF_DEV float loadAgentF(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_match(MatchArgs a, MatchScratch sc) {
    const uint32_t prev = blockIdx.x + a.startFrame;
    if (prev == a.curFrame) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c16 = lane & 15;
    int n1 = a.numKeys[prev], n2 = a.numKeys[a.curFrame];
    n1 = min(max(n1, 0), (int)a.maxKeys); n2 = min(max(n2, 0), (int)a.maxKeys);
    if ((!a.speculative && a.validImages[prev] == 0) || n1 == 0 || n2 == 0) {       // Bundler.cpp:126-129 (every workgroup of the pair takes this branch)
        if (tid == 0 && blockIdx.y == 0) a.numMatches[prev] = 0;
        return;
    }
    __shared__ int otherSum[1024];
    __shared__ int ownSum[64];
    __shared__ int rawR[MAX_RAW], rawC[MAX_RAW];
    __shared__ float rawD[MAX_RAW];
    __shared__ int waveTot[4];
    __shared__ uint32_t lastFlag;
    const bool cols = blockIdx.y >= 16;                  // block-uniform
    const uint32_t slab = blockIdx.y & 15u;
    const uint8_t* D1 = a.descs + (size_t)prev * a.maxKeys * 128;
    const uint8_t* D2 = a.descs + (size_t)a.curFrame * a.maxKeys * 128;
    const uint8_t* DO = cols ? D2 : D1; const uint8_t* DX = cols ? D1 : D2;      // own side (this slab's 64 keys), other side (all keys)
    const int nOwn = cols ? n2 : n1, nOther = cols ? n1 : n2;
    int* resRow = sc.rowRes + (size_t)prev * 1024; float* distRow = sc.rowDist + (size_t)prev * 1024; int* resCol = sc.colRes + (size_t)prev * 1024;
    if ((int)(slab * 64u) < nOwn) {
        // byte sums (the bias correction of the signed MFMA operands)
        for (uint32_t k = tid; k < 1024; k += 256) {
            int sx = 0;
            if ((int)k < nOther) { const v4i* p = (const v4i*)(DX + (size_t)k * 128); for (int i = 0; i < 8; ++i) sx += byte_sum16(p[i]); }
            otherSum[k] = sx - 16384;
        }
        if (tid < 64) {
            const int k = (int)(slab * 64u + tid);
            int so = 0;
            if (k < nOwn) { const v4i* p = (const v4i*)(DO + (size_t)k * 128); for (int i = 0; i < 8; ++i) so += byte_sum16(p[i]); }
            ownSum[tid] = so - 16384;
        }
        __syncthreads();
        const int chunk = (int)(slab * 4u + w);          // 16 keys of the own side
        if (chunk * 16 < nOwn) {                          // wave-uniform
            const int arow = chunk * 16 + (int)c16;
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            if (arow < nOwn) { const uint8_t* p = DO + (size_t)arow * 128 + 16 * g; a0 = *(const v4i*)p; a1 = *(const v4i*)(p + 64); }
            a0 ^= (int)0x80808080; a1 ^= (int)0x80808080;
            uint64_t rP[4]; int rN[4]; int so[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { rP[r] = P_NONE; rN[r] = 0; so[r] = ownSum[(int)w * 16 + 4 * (int)g + r]; }
            const int nOT = (nOther + 15) >> 4;
            for (int ct = 0; ct < nOT; ++ct) {
                const int col = ct * 16 + (int)c16;
                v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
                if (col < nOther) { const uint8_t* p = DX + (size_t)col * 128 + 16 * g; b0 = *(const v4i*)p; b1 = *(const v4i*)(p + 64); }
                b0 ^= (int)0x80808080; b1 ^= (int)0x80808080;
                const int sb = otherSum[min(col, 1023)];
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc, 0, 0, 0);
                // the tie-break key of the OTHER side's index: rows prefer columns by ((col & 31) << 16 | col), columns prefer rows by (((row >> 2) & 31) << 16 | row)
                // - the traversal orders of RowMatch_Kernel / ColMatch_Kernel (ProgramCU.cu:1780-1916)
                const uint32_t kx = cols ? ((((uint32_t)col >> 2) & 31u) << 16) | (uint32_t)col : (((uint32_t)col & 31u) << 16) | (uint32_t)col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = chunk * 16 + 4 * (int)g + r;
                    const bool ok = rr < nOwn && col < nOther;
                    const int d = ok ? acc[r] + 128 * (so[r] + sb) + 2097152 : 0;
                    const uint64_t P = ok ? (((uint64_t)(uint32_t)d << 32) | (0xFFFFFFFEu - kx)) : 0ull;
                    top2_add(rP[r], rN[r], P, d);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // fold the 16 lanes that share an own-side key
                uint64_t P = rP[r]; int n = rN[r];
                for (int m = 1; m <= 8; m <<= 1) { const uint64_t oP = shfl_xor_u64(P, m); const int oN = __shfl_xor(n, m, 64); top2_merge(P, n, oP, oN); }
                const int rr = chunk * 16 + 4 * (int)g + r;
                if (c16 == 0 && rr < nOwn) {               // RowMatch_Kernel :1813-1829 / ColMatch_Kernel :1898-1916
                    const float dist = bf_dm_acos(fminf((float)(int)(P >> 32) * 0.000003814697265625f, 1.0f));
                    const float distn = bf_dm_acos(fminf((float)n * 0.000003814697265625f, 1.0f));
                    const int res = (dist < a.distmax) && (dist < distn * a.ratiomax) ? decode_idx(P) : -1;
                    if (cols) storeAgent(resCol + rr, res);
                    else { storeAgent(resRow + rr, res); storeAgentF(distRow + rr, dist); }
                }
            }
        }
    }
    // hand-off to the pair's last workgroup (write-through stores drained, ticket, acquire: see k_alloc_place in tsdf.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = atomicAdd(sc.ticket + prev, 1u);
        lastFlag = t == gridDim.y - 1 ? 1u : 0u;
        if (lastFlag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!lastFlag) return;
    // columns in ascending order: ColMatch's verdict, the mutual check, ordered append
    int total = 0;
    for (int c0 = 0; c0 < n2; c0 += 256) {
        const int col = c0 + (int)tid;
        int f1 = -1; bool hit = false;
        if (col < n2) { f1 = loadAgent(resCol + col); hit = f1 >= 0 && f1 < n1 && loadAgent(resRow + f1) == col; }      // (other workgroups' results: loads at the scope they were stored at)
        const uint64_t ball = __ballot(hit);
        if (lane == 0) waveTot[w] = __popcll(ball);
        __syncthreads();
        int base = total, add = 0;
        for (int i = 0; i < 4; ++i) { if (i < (int)w) base += waveTot[i]; add += waveTot[i]; }
        const int pos = base + __popcll(ball & ((1ull << lane) - 1ull));
        if (hit && pos < MAX_RAW) { rawR[pos] = f1; rawC[pos] = col; rawD[pos] = loadAgentF(distRow + f1); }
        total += add;
        __syncthreads();
    }
    const int m = min(total, MAX_RAW);
    if ((int)tid < m) {           // SortKeyPointMatchesCU :59-143 - stable ascending by distance
        const float di = rawD[tid];
        int rank = 0;
        for (int j = 0; j < m; ++j) { const float dj = rawD[j]; rank += (dj < di || (dj == di && j < (int)tid)) ? 1 : 0; }
        a.idx[prev * MAX_RAW + rank] = make_uint2(prev * a.maxKeys + (uint32_t)rawR[tid], a.curFrame * a.maxKeys + (uint32_t)rawC[tid]);
        a.dist[prev * MAX_RAW + rank] = di;
    }
    if (tid == 0) { a.numMatches[prev] = total; sc.ticket[prev] = 0u; }
}

// ------------------------------------------------------------------------------------------------ 3x3 helpers
BF_DEV float rsq(float x) { return 1.0f / sqrtf(x); }
BF_DEV float det3(const float* m) {
    return m[0] * m[4] * m[8] + m[1] * m[5] * m[6] + m[2] * m[3] * m[7] - m[6] * m[4] * m[2] - m[7] * m[5] * m[0] - m[8] * m[3] * m[1];
}
BF_DEV void mm3(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

// McAdams, Selle, Tamstorf, Teran, Sifakis: "Computing the SVD of 3x3 matrices with minimal branching" as used
// by cuda_svd3.h: 4 Jacobi sweeps with the approximate Givens quaternion, sort, QR by Givens.
__device__ const float SVD_GAMMA = 5.828427124f, SVD_CSTAR = 0.923879532f, SVD_SSTAR = 0.3826834323f, SVD_EPS = 1e-6f;
BF_DEV void givensQuat(float a11, float a12, float a22, float& ch, float& sh) {
    ch = 2 * (a11 - a22); sh = a12;
    const bool b = SVD_GAMMA * sh * sh < ch * ch;
    const float wv = rsq(ch * ch + sh * sh);
    ch = b ? wv * ch : SVD_CSTAR; sh = b ? wv * sh : SVD_SSTAR;
}
BF_DEV void jacobiConj(int x, int y, int z, float& s11, float& s21, float& s22, float& s31, float& s32, float& s33, float* q) {
    float ch, sh;
    givensQuat(s11, s21, s22, ch, sh);
    const float scale = ch * ch + sh * sh;
    const float a = (ch * ch - sh * sh) / scale, b = (2 * sh * ch) / scale;
    const float t11 = s11, t21 = s21, t22 = s22, t31 = s31, t32 = s32, t33 = s33;
    s11 = a * (a * t11 + b * t21) + b * (a * t21 + b * t22);
    s21 = a * (-b * t11 + a * t21) + b * (-b * t21 + a * t22);
    s22 = -b * (-b * t11 + a * t21) + a * (-b * t21 + a * t22);
    s31 = a * t31 + b * t32; s32 = -b * t31 + a * t32; s33 = t33;
    const float tmp[3] = {q[0] * sh, q[1] * sh, q[2] * sh};
    sh *= q[3];
    q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
    q[z] += sh; q[3] -= tmp[z]; q[x] += tmp[y]; q[y] -= tmp[x];
    const float n11 = s22, n21 = s32, n22 = s33, n31 = s21, n32 = s31, n33 = s11;
    s11 = n11; s21 = n21; s22 = n22; s31 = n31; s32 = n32; s33 = n33;
}
BF_DEV void cswap(bool c, float& X, float& Y) { const float Z = X; X = c ? Y : X; Y = c ? Z : Y; }
BF_DEV void cnswap(bool c, float& X, float& Y) { const float Z = -X; X = c ? Y : X; Y = c ? Z : Y; }
BF_DEV void qrGivens(float a1, float a2, float& ch, float& sh) {
    const float rho = (a1 * a1 + a2 * a2) * rsq(a1 * a1 + a2 * a2);
    sh = rho > SVD_EPS ? a2 : 0;
    ch = fabsf(a1) + fmaxf(rho, SVD_EPS);
    cswap(a1 < 0, sh, ch);
    const float wv = rsq(ch * ch + sh * sh);
    ch *= wv; sh *= wv;
}
BF_DEV void svd3(const float* A, float* U, float* S, float* V) {
    const float a11 = A[0], a12 = A[1], a13 = A[2], a21 = A[3], a22 = A[4], a23 = A[5], a31 = A[6], a32 = A[7], a33 = A[8];
    float s11 = a11 * a11 + a21 * a21 + a31 * a31, s21 = a12 * a11 + a22 * a21 + a32 * a31, s22 = a12 * a12 + a22 * a22 + a32 * a32;
    float s31 = a13 * a11 + a23 * a21 + a33 * a31, s32 = a13 * a12 + a23 * a22 + a33 * a32, s33 = a13 * a13 + a23 * a23 + a33 * a33;
    float q[4] = {0, 0, 0, 1};
    for (int i = 0; i < 4; ++i) {
        jacobiConj(0, 1, 2, s11, s21, s22, s31, s32, s33, q);
        jacobiConj(1, 2, 0, s11, s21, s22, s31, s32, s33, q);
        jacobiConj(2, 0, 1, s11, s21, s22, s31, s32, s33, q);
    }
    const float w = q[3], x = q[0], y = q[1], z = q[2];
    float v11 = 1 - 2 * (y * y + z * z), v12 = 2 * (x * y - w * z), v13 = 2 * (x * z + w * y);
    float v21 = 2 * (x * y + w * z), v22 = 1 - 2 * (x * x + z * z), v23 = 2 * (y * z - w * x);
    float v31 = 2 * (x * z - w * y), v32 = 2 * (y * z + w * x), v33 = 1 - 2 * (x * x + y * y);
    float b11 = a11 * v11 + a12 * v21 + a13 * v31, b12 = a11 * v12 + a12 * v22 + a13 * v32, b13 = a11 * v13 + a12 * v23 + a13 * v33;
    float b21 = a21 * v11 + a22 * v21 + a23 * v31, b22 = a21 * v12 + a22 * v22 + a23 * v32, b23 = a21 * v13 + a22 * v23 + a23 * v33;
    float b31 = a31 * v11 + a32 * v21 + a33 * v31, b32 = a31 * v12 + a32 * v22 + a33 * v32, b33 = a31 * v13 + a32 * v23 + a33 * v33;
    // column-norm sort; the second norm reads b23 for the third term exactly like cuda_svd3.h:236
    float rho1 = b11 * b11 + b21 * b21 + b31 * b31, rho2 = b12 * b12 + b22 * b22 + b23 * b23, rho3 = b13 * b13 + b23 * b23 + b33 * b33;
    bool c = rho1 < rho2;
    cnswap(c, b11, b12); cnswap(c, v11, v12); cnswap(c, b21, b22); cnswap(c, v21, v22); cnswap(c, b31, b32); cnswap(c, v31, v32); cswap(c, rho1, rho2);
    c = rho1 < rho3;
    cnswap(c, b11, b13); cnswap(c, v11, v13); cnswap(c, b21, b23); cnswap(c, v21, v23); cnswap(c, b31, b33); cnswap(c, v31, v33); cswap(c, rho1, rho3);
    c = rho2 < rho3;
    cnswap(c, b12, b13); cnswap(c, v12, v13); cnswap(c, b22, b23); cnswap(c, v22, v23); cnswap(c, b32, b33); cnswap(c, v32, v33);
    float ch1, sh1, ch2, sh2, ch3, sh3, a, b;
    float r11, r12, r13, r21, r22, r23, r31, r32, r33;
    qrGivens(b11, b21, ch1, sh1);
    a = 1 - 2 * sh1 * sh1; b = 2 * ch1 * sh1;
    r11 = a * b11 + b * b21; r12 = a * b12 + b * b22; r13 = a * b13 + b * b23;
    r21 = -b * b11 + a * b21; r22 = -b * b12 + a * b22; r23 = -b * b13 + a * b23;
    r31 = b31; r32 = b32; r33 = b33;
    qrGivens(r11, r31, ch2, sh2);
    a = 1 - 2 * sh2 * sh2; b = 2 * ch2 * sh2;
    b11 = a * r11 + b * r31; b12 = a * r12 + b * r32; b13 = a * r13 + b * r33;
    b21 = r21; b22 = r22; b23 = r23;
    b31 = -b * r11 + a * r31; b32 = -b * r12 + a * r32; b33 = -b * r13 + a * r33;
    qrGivens(b22, b32, ch3, sh3);
    a = 1 - 2 * sh3 * sh3; b = 2 * ch3 * sh3;
    r11 = b11; r12 = b12; r13 = b13;
    r21 = a * b21 + b * b31; r22 = a * b22 + b * b32; r23 = a * b23 + b * b33;
    r31 = -b * b21 + a * b31; r32 = -b * b22 + a * b32; r33 = -b * b23 + a * b33;
    const float sh12 = sh1 * sh1, sh22 = sh2 * sh2, sh32 = sh3 * sh3;
    U[0] = (-1 + 2 * sh12) * (-1 + 2 * sh22);
    U[1] = 4 * ch2 * ch3 * (-1 + 2 * sh12) * sh2 * sh3 + 2 * ch1 * sh1 * (-1 + 2 * sh32);
    U[2] = 4 * ch1 * ch3 * sh1 * sh3 - 2 * ch2 * (-1 + 2 * sh12) * sh2 * (-1 + 2 * sh32);
    U[3] = 2 * ch1 * sh1 * (1 - 2 * sh22);
    U[4] = -8 * ch1 * ch2 * ch3 * sh1 * sh2 * sh3 + (-1 + 2 * sh12) * (-1 + 2 * sh32);
    U[5] = -2 * ch3 * sh3 + 4 * sh1 * (ch3 * sh1 * sh3 + ch1 * ch2 * sh2 * (-1 + 2 * sh32));
    U[6] = 2 * ch2 * sh2;
    U[7] = 2 * ch3 * (1 - 2 * sh22) * sh3;
    U[8] = (-1 + 2 * sh22) * (-1 + 2 * sh32);
    S[0] = r11; S[1] = r12; S[2] = r13; S[3] = r21; S[4] = r22; S[5] = r23; S[6] = r31; S[7] = r32; S[8] = r33;
    V[0] = v11; V[1] = v12; V[2] = v13; V[3] = v21; V[4] = v22; V[5] = v23; V[6] = v31; V[7] = v32; V[8] = v33;
}

// closed-form eigenvalues of a symmetric 3x3 (trigonometric method, cuda_EigenValue.h:9-41), x >= y >= z
BF_DEV f3 eigenValues3(const float* A) {
    const float PI = 3.14159265f;
    f3 e;
    float p = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (p == 0) { e.x = A[0]; e.y = A[4]; e.z = A[8]; return e; }
    const float q = (A[0] + A[4] + A[8]) / 3.0f;
    p = (A[0] - q) * (A[0] - q) + (A[4] - q) * (A[4] - q) + (A[8] - q) * (A[8] - q) + 2.0f * p;
    p = sqrtf(p / 6.0f);
    float B[9];
    const float ip = 1.0f / p;
#pragma unroll
    for (int i = 0; i < 9; ++i) B[i] = (A[i] - ((i % 4 == 0) ? q : 0.0f)) * ip;
    const float r = det3(B) / 2.0f;
    float phi;
    if (r <= -1.0f) phi = PI / 3.0f;
    else if (r >= 1) phi = 0;
    else phi = bf_dm_acos(r) / 3.0f;
    float s, c;
    bf_dm_sincos(phi, &s, &c);
    e.x = q + 2.0f * p * c;
    bf_dm_sincos(phi + PI * (2.0f / 3.0f), &s, &c);
    e.z = q + 2.0f * p * c;
    e.y = 3.0f * q - e.x - e.z;
    return e;
}

BF_DEV f3 backProject(const m44& Kinv, const Key& k) {      // Kinv * (depth * (x, y, 1))
    return xform(Kinv, mk3(k.depth * k.x, k.depth * k.y, k.depth * 1.0f));
}

struct Sel { uint32_t ix, iy; float dist; uint32_t r; };     // r: slot of the match in the raw (distance-sorted) list

// everything of cuda_kabsch.h:73-211 behind the accumulation loops: V the cross-covariance (already divided by n), p0 / q0 the centroids
BF_DEV m44 kabschFromMoments(const float* V, f3 p0, f3 q0, f3& evs) {
    float U[9], S[9], W[9];
    svd3(V, U, S, W);
    float s[3] = {S[0], S[4], S[8]};
#pragma unroll
    for (int i = 0; i < 3; ++i) if (s[i] < 0.0f) { s[i] *= -1.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) U[j * 3 + i] *= -1.0f; }
    evs = mk3(s[0], s[1], s[2]);
    if (evs.x < evs.y) { const float t = evs.x; evs.x = evs.y; evs.y = t; }
    if (evs.y < evs.z) { const float t = evs.y; evs.y = evs.z; evs.z = t; }
    if (evs.x < evs.y) { const float t = evs.x; evs.x = evs.y; evs.y = t; }
    const float Wt[9] = {W[0], W[3], W[6], W[1], W[4], W[7], W[2], W[5], W[8]};
    float UWt[9];
    mm3(U, Wt, UWt);
    float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (det3(UWt) < 0) I[8] = -1;
    const float Ut[9] = {U[0], U[3], U[6], U[1], U[4], U[7], U[2], U[5], U[8]};
    float WI[9], R[9];
    mm3(W, I, WI);
    mm3(WI, Ut, R);
    m44 ret = identity44();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ret.e[i * 4 + j] = R[i * 3 + j];
    ret.e[3] = q0.x - (R[0] * p0.x + R[1] * p0.y + R[2] * p0.z);
    ret.e[7] = q0.y - (R[3] * p0.x + R[4] * p0.y + R[5] * p0.z);
    ret.e[11] = q0.z - (R[6] * p0.x + R[7] * p0.y + R[8] * p0.z);
    return ret;
}

// (kabsch(), cuda_kabsch.h:73-110 - centroids, cross-covariance - is spread over the lanes of computeReprojection below; what follows the sums is kabschFromMoments above)

BF_DEV f3 covarianceEig(const f3* pts, unsigned n) {
    f3 p0 = mk3(0, 0, 0);
    for (unsigned i = 0; i < n; ++i) p0 = p0 + pts[i];
    p0 = p0 / (float)n;
    float V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = 0; i < n; ++i) {
        const f3 p = pts[i] - p0;
        const float pv[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) V[r * 3 + c] += pv[r] * pv[c];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] /= (float)n;
    return eigenValues3(V);
}

template <class T> BF_DEV void swp(T& a, T& b) { const T t = a; a = b; b = t; }

// ComputeReprojection (cuda_kabsch.h:386-419) executed by one wave.  The arithmetic of every value is the sequential
// reference sequence; only independent pieces run on different lanes:
//   lane 0          Kabsch fit (sums in index order, SVD)                 -> T, singular values
//   lanes < n       residual of point `lane`
//   lanes < n       rank of each residual = its slot after sortKabschResiduals (:376-384); the selection sort's result is
//                   unique when all residuals are finite and distinct — otherwise lane 0 runs the literal sort
//   lanes 0 and 1   covariance eigenvalues of the (sorted) source / target points
// (Measured and withdrawn in round 4, gpurun r04a: the accumulation loops of the fit and of the two covariance solves spread one sum per lane, bit-identical -
// 244 vs 259 us per launch, no change at the frame level: the time is in the dependent SVD / eigenvalue chains, not in the sums.)
struct ReprojShared { m44 T; float ev[3]; float cond[2]; float mom[15]; };      // mom: the fit's centroids (6) and cross-covariance (9), one lane each

// The verdict of ComputeReprojection (:404-418): condition numbers of the fit and of the two point sets, on the state the last computeReprojection left behind.
// Round 5: evaluated only where the greedy filter READS it (the end of the walk, and the two places of the removal loop) - the reference computes the two
// covariance eigen-decompositions after every fit and overwrites the result unread in all but 1-3 of the ~35 fits of a pair (a quarter of the kernel's time).
__device__ __noinline__ bool reprojectionValid(uint32_t lane, const f3* src, const f3* tgt, unsigned n, ReprojShared* sh) {
    if (lane < 2) { const f3 e = covarianceEig(lane == 0 ? src : tgt, n); sh->cond[lane] = e.x / e.y; }
    __syncthreads();
    const float c1 = sh->ev[0] / sh->ev[1], cp = sh->cond[0], cq = sh->cond[1];
    __syncthreads();
    if (c1 != c1 || cp != cp || cq != cq || fabsf(c1) > 100.0f || fabsf(cp) > 100.0f || fabsf(cq) > 100.0f) return false;
    return true;
}

__device__ __noinline__ void computeReprojection(uint32_t lane, f3* src, f3* tgt, unsigned n, float* res, Sel* sel, ReprojShared* sh) {
    // kabsch() with its fifteen accumulation loops on fifteen lanes: every sum is the same sequential chain over i = 0 .. n - 1 (the operations of the loops in
    // kabsch(), cuda_kabsch.h:73-110, component by component), only no longer one after the other on lane 0 - a quarter of a fit's instructions at n = 14.
    // (Round 4 measured this form at 244 vs 259 us per launch when the two covariance eigen-solves still ran after every fit; they no longer do.)
    {
        const float* sf = reinterpret_cast<const float*>(src); const float* tf = reinterpret_cast<const float*>(tgt);
        if (lane < 6) {
            const float* b = (lane < 3 ? sf : tf) + (lane % 3u);
            float acc = 0.0f;
            for (unsigned i = 0; i < n; ++i) acc = acc + b[3 * i];
            sh->mom[lane] = acc / (float)n;
        }
        __syncthreads();
        if (lane < 9) {
            const unsigned r = lane / 3u, c = lane % 3u;
            const float p0 = sh->mom[r], q0 = sh->mom[3 + c];
            float acc = 0.0f;
            for (unsigned i = 0; i < n; ++i) { const float pr = sf[3 * i + r] - p0, qc = tf[3 * i + c] - q0; acc += pr * qc; }
            sh->mom[6 + lane] = acc / (float)n;
        }
        __syncthreads();
    }
    if (lane == 0) {
        f3 ev;
        float V[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = sh->mom[6 + i];
        sh->T = kabschFromMoments(V, mk3(sh->mom[0], sh->mom[1], sh->mom[2]), mk3(sh->mom[3], sh->mom[4], sh->mom[5]), ev);
        sh->ev[0] = ev.x; sh->ev[1] = ev.y; sh->ev[2] = ev.z;
    }
    __syncthreads();
    float my = 0.0f;
    if (lane < n) { const f3 d = xform(sh->T, src[lane]) - tgt[lane]; my = dot3(d, d); res[lane] = my; }
    __syncthreads();
    unsigned rank = 0;
    bool odd = false;               // NaN or a tie: the generic sort order is not determined by the values alone
    if (lane < n) {
        odd = my != my;
        for (unsigned j = 0; j < n; ++j) { const float rj = res[j]; rank += rj < my ? 1u : 0u; odd = odd || (j != lane && rj == my); }
    }
    if (__ballot(odd) == 0ull) {
        f3 s_ = mk3(0, 0, 0), t_ = mk3(0, 0, 0); Sel e_ = {0u, 0u, 0.0f, 0u};
        if (lane < n) { s_ = src[lane]; t_ = tgt[lane]; e_ = sel[lane]; }
        __syncthreads();
        if (lane < n) { src[rank] = s_; tgt[rank] = t_; sel[rank] = e_; res[rank] = my; }
    } else if (lane == 0) {
        for (unsigned i = 0; i < n; ++i)
            for (unsigned j = i; j < n; ++j)
                if (res[i] > res[j]) { swp(res[i], res[j]); swp(src[i], src[j]); swp(tgt[i], tgt[j]); swp(sel[i], sel[j]); }
    }
    __syncthreads();
}

struct FilterArgs {
    const Key* keys; uint32_t curFrame, startFrame;
    const int* numMatches; const float* dist; const uint2* idx;
    int* numFilt; float* fdist; uint2* fidx; m44* T; m44* Tinv;
    m44 Kinv; int minNumMatches; float maxKabschRes2;
};

// One wave per previous image.  The greedy filter is inherently sequential (every accepted match changes the Kabsch fit
// the next decision depends on); the wave walks it in lock step — control state is replicated in all lanes, data lives in
// LDS (the <= 128 raw matches' key positions and back-projected points are staged once), and the independent pieces of each
// step (distance checks, residuals, sort ranks, the two covariance solves) are spread over lanes.
__global__ __launch_bounds__(64) void k_filter_kabsch(FilterArgs a) {
    const uint32_t prev = blockIdx.x + a.startFrame;
    if (prev == a.curFrame) return;
    const uint32_t tid = threadIdx.x;
    const int numRaw = min(MAX_RAW, max(a.numMatches[prev], 0));
    if (numRaw == 0) { if (tid == 0) a.numFilt[prev] = 0; return; }
    __shared__ Sel sel[MAX_RAW + MAX_FILT];
    __shared__ float2 posI[MAX_RAW], posJ[MAX_RAW];
    __shared__ f3 ptI[MAX_RAW], ptJ[MAX_RAW];
    __shared__ f3 src[MAX_FILT], tgt[MAX_FILT];
    __shared__ float res[MAX_FILT];
    __shared__ ReprojShared sh;
    __shared__ m44 prevT;
    for (int i = (int)tid; i < numRaw; i += 64) {
        const uint2 k = a.idx[prev * MAX_RAW + i];
        sel[i].ix = k.x; sel[i].iy = k.y; sel[i].dist = a.dist[prev * MAX_RAW + i]; sel[i].r = (uint32_t)i;
        const Key ki = a.keys[k.x], kj = a.keys[k.y];
        posI[i] = make_float2(ki.x, ki.y); posJ[i] = make_float2(kj.x, kj.y);
        ptI[i] = backProject(a.Kinv, ki); ptJ[i] = backProject(a.Kinv, kj);
    }
    if (tid == 0) sh.T = identity44();
    __syncthreads();
    // filterKeyPointMatches, cuda_kabsch.h:422-502
    unsigned cur = 0;
    int i = 0;
    float curMaxRes = 100.0f;
    bool validT = false;
    bool pending = false;           // validT is the verdict of the state in (src, tgt, cur, sh.ev) and has not been evaluated yet
    for (;;) {
        if (i == numRaw || cur >= (unsigned)MAX_FILT) {
            if (pending) { validT = reprojectionValid(tid, src, tgt, cur, &sh); pending = false; }
            if ((int)cur < a.minNumMatches || curMaxRes >= a.maxKabschRes2 || !validT) cur = 0;
            break;
        }
        bool close = false;         // addMatch :278-294: at least 5 px from every kept match in both images
        if (tid < cur) {
            const float2 ai = posI[i], aj = posJ[i];          // raw entries at positions >= cur are never permuted: sel[i].r == i
            const float2 ki = posI[sel[tid].r], kj = posJ[sel[tid].r];
            const float d0 = sqrtf((ai.x - ki.x) * (ai.x - ki.x) + (ai.y - ki.y) * (ai.y - ki.y));
            const float d1 = sqrtf((aj.x - kj.x) * (aj.x - kj.x) + (aj.y - kj.y) * (aj.y - kj.y));
            close = d0 <= 5 || d1 <= 5;
        }
        if (__ballot(close) == 0ull) {
            if (tid == 0) sel[cur] = sel[i];
            __syncthreads();
            cur++;
            if (cur >= 3) {
                if (tid < cur) { src[tid] = ptI[sel[tid].r]; tgt[tid] = ptJ[sel[tid].r]; }
                __syncthreads();
                computeReprojection(tid, src, tgt, cur, res, sel, &sh);
                pending = true;
                if (tid < 16) prevT.e[tid] = sh.T.e[tid];
                curMaxRes = res[cur - 1];
                if (curMaxRes > a.maxKabschRes2) {
                    const bool b = reprojectionValid(tid, src, tgt, cur, &sh);      // the verdict of this fit: read below if the removals go down to three matches
                    validT = b; pending = false;
                    float lastRes = -1;
                    const int startIdx = (int)cur - 1;
                    for (int k = startIdx; k >= 3; --k) {
                        lastRes = res[k];
                        cur--;
                        __syncthreads();
                        computeReprojection(tid, src, tgt, cur, res, sel, &sh);
                        pending = true;
                        curMaxRes = res[cur - 1];
                        if (cur == 3) { validT = reprojectionValid(tid, src, tgt, cur, &sh); pending = false; }
                        if (cur == 3 && (curMaxRes > a.maxKabschRes2 || (b && !validT))) {
                            cur++; curMaxRes = lastRes; validT = b; pending = false;
                            __syncthreads();
                            if (tid < 16) sh.T.e[tid] = prevT.e[tid];
                            break;
                        }
                        if (curMaxRes < a.maxKabschRes2) break;
                    }
                }
            }
        }
        __syncthreads();
        i++;
    }
    __syncthreads();
    if (tid == 0) {
        const m44 T = sh.T;
        a.T[prev] = T;
        a.Tinv[prev] = inverse44(T);
        a.numFilt[prev] = (int)cur;
    }
    if (tid < (unsigned)MAX_FILT) {
        if (tid < cur) { a.fidx[prev * MAX_FILT + tid] = make_uint2(sel[tid].ix, sel[tid].iy); a.fdist[prev * MAX_FILT + tid] = sel[tid].dist; }
        else { a.fidx[prev * MAX_FILT + tid] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); a.fdist[prev * MAX_FILT + tid] = 999.0f; }
    }
}

// ------------------------------------------------------------------------------------------------ surface area
// cyclic Jacobi sweeps of a symmetric 3x3 (the textbook rotation scheme cuda_SVD.h:133-208 uses); like
// MYEIGEN::eigenSystem (:69-122) the i-th "eigenvector" returned is ROW i of the accumulated rotation.
__device__ __noinline__ bool eigenSystem3(const float* M, float* evals, float evecs[3][3]) {
    float a[3][3], v[3][3], d[3], b[3], z[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = M[i + 3 * j]; v[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (int i = 0; i < 3; ++i) { b[i] = d[i] = a[i][i]; z[i] = 0.0f; }
    bool ok = false;
    for (int sweep = 1; sweep <= 50; ++sweep) {
        float sm = 0.0f;
        for (int ip = 0; ip < 2; ++ip) for (int iq = ip + 1; iq < 3; ++iq) sm += fabsf(a[ip][iq]);
        if (sm == 0.0f) { ok = true; break; }
        const float tresh = sweep < 4 ? 0.2f * sm / 9 : 0.0f;
        for (int ip = 0; ip < 2; ++ip)
            for (int iq = ip + 1; iq < 3; ++iq) {
                const float g = 100.0f * fabsf(a[ip][iq]);
                if (sweep > 4 && (float)(fabsf(d[ip]) + g) == fabsf(d[ip]) && (float)(fabsf(d[iq]) + g) == fabsf(d[iq])) a[ip][iq] = 0.0f;
                else if (fabsf(a[ip][iq]) > tresh) {
                    float h = d[iq] - d[ip], t;
                    if ((float)(fabsf(h) + g) == fabsf(h)) t = a[ip][iq] / h;
                    else {
                        const float theta = 0.5f * h / a[ip][iq];
                        t = 1.0f / (fabsf(theta) + sqrtf(1.0f + theta * theta));
                        if (theta < 0.0f) t = -t;
                    }
                    const float c = 1.0f / sqrtf(1 + t * t), s = t * c, tau = s / (1.0f + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h;
                    a[ip][iq] = 0.0f;
#define BF_ROT(m, i, j, k, l) { const float gg = m[i][j], hh = m[k][l]; m[i][j] = gg - s * (hh + gg * tau); m[k][l] = hh + s * (gg - hh * tau); }
                    for (int j = 0; j <= ip - 1; ++j) BF_ROT(a, j, ip, j, iq)
                    for (int j = ip + 1; j <= iq - 1; ++j) BF_ROT(a, ip, j, j, iq)
                    for (int j = iq + 1; j < 3; ++j) BF_ROT(a, ip, j, iq, j)
                    for (int j = 0; j < 3; ++j) BF_ROT(v, j, ip, j, iq)
#undef BF_ROT
                }
            }
        for (int i = 0; i < 3; ++i) { b[i] += z[i]; d[i] = b[i]; z[i] = 0.0f; }
    }
    if (!ok) return false;
    for (int i = 0; i < 3; ++i) { evals[i] = d[i]; for (int j = 0; j < 3; ++j) evecs[i][j] = v[i][j]; }
    for (int i = 0; i < 3; ++i) {
        float cur = 0.0f; int idx = -1;
        for (int j = i; j < 3; ++j) if (fabsf(evals[j]) > cur) { cur = fabsf(evals[j]); idx = j; }
        if (idx != i && idx != -1) { swp(evals[i], evals[idx]); for (int j = 0; j < 3; ++j) swp(evecs[i][j], evecs[idx][j]); }
    }
    return true;
}

// warpReduceSum (cudaUtil.h:25-29) as lane 0 of a 32-lane warp sees it: val += shfl_down(val, 16), 8, 4, 2, 1; lanes >= n hold 0
BF_DEV float warpTree32(const float* v, int n) {
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = i < n ? v[i] : 0.0f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
#pragma unroll
        for (int i = 0; i < off; ++i) x[i] = x[i] + x[i + off];
    return x[0];
}

struct AreaArgs { const Key* keys; uint32_t curFrame, startFrame; int* numFilt; const uint2* fidx; m44 Kinv; float areaThresh; };

__global__ __launch_bounds__(64) void k_filter_surface_area(AreaArgs a) {
    const uint32_t prev = blockIdx.x + a.startFrame;
    if (prev == a.curFrame) return;
    const int n = min(a.numFilt[prev], MAX_FILT);
    if (n <= 0) return;
    __shared__ f3 ptsAll[2][MAX_FILT];
    __shared__ float pxAll[2][MAX_FILT], pyAll[2][MAX_FILT];
    __shared__ float area[2];
    if ((int)threadIdx.x < 2 * n) {                  // back-project both images' keys in parallel
        const int which = (int)threadIdx.x / n, i = (int)threadIdx.x % n;
        const uint2 k = a.fidx[prev * MAX_FILT + i];
        ptsAll[which][i] = backProject(a.Kinv, a.keys[which ? k.y : k.x]);
    }
    if (threadIdx.x < 2) area[threadIdx.x] = 0.0f;
    __syncthreads();
    // lanes 0 and 1: the two images' areas side by side (round 5; one lane computed both, one after the other) - each is the reference's sequence for its image
    if (threadIdx.x < 2) {
        const int which = (int)threadIdx.x;
        float t[32];
        float* px = pxAll[which]; float* py = pyAll[which];
        do {
        const f3* pts = ptsAll[which];
        // every sum below is a warpReduceSum of the reference's 32-thread block (cuda_surfaceArea.h:13-84, cudaUtil.h:25-29): warpTree32
        f3 mean;
        for (int i = 0; i < n; ++i) t[i] = pts[i].x; mean.x = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = pts[i].y; mean.y = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = pts[i].z; mean.z = warpTree32(t, n);
        mean = mean / (float)n;
        float V[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                for (int i = 0; i < n; ++i) {
                    const f3 p = pts[i] - mean;
                    const float pr = r == 0 ? p.x : (r == 1 ? p.y : p.z), pc = c == 0 ? p.x : (c == 1 ? p.y : p.z);
                    t[i] = pr * pc;
                }
                V[r * 3 + c] = warpTree32(t, n);
            }
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] /= (float)n;
        float evals[3], ev[3][3];
        if (!eigenSystem3(V, evals, ev)) break;
        const f3 ev0 = mk3(ev[0][0], ev[0][1], ev[0][2]), ev1 = mk3(ev[1][0], ev[1][1], ev[1][2]), ev2 = mk3(ev[2][0], ev[2][1], ev[2][2]);
        for (int i = 0; i < n; ++i) {                   // projectKeysToPlane, cuda_surfaceArea.h:134-158
            const f3 s = (pts[i] - ev2 * dot3(ev2, pts[i] - mean)) - mean;
            px[i] = dot3(s, ev0); py[i] = dot3(s, ev1);
        }
        // oriented bounding box in the plane, cuda_surfaceArea.h:87-131
        float mx, my;
        for (int i = 0; i < n; ++i) t[i] = px[i]; mx = warpTree32(t, n);
        for (int i = 0; i < n; ++i) t[i] = py[i]; my = warpTree32(t, n);
        mx /= (float)n; my /= (float)n;
        float c00, c01, c10, c11;                     // (c10 is accumulated like in the reference and, like there, never read)
        for (int i = 0; i < n; ++i) { const float u = px[i] - mx; t[i] = u * u; } c00 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float u = px[i] - mx, v = py[i] - my; t[i] = u * v; } c01 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float u = px[i] - mx, v = py[i] - my; t[i] = v * u; } c10 = warpTree32(t, n);
        for (int i = 0; i < n; ++i) { const float v = py[i] - my; t[i] = v * v; } c11 = warpTree32(t, n);
        c00 /= (float)n; c01 /= (float)n; c10 /= (float)n; c11 /= (float)n; (void)c10;
        const float disc = 0.5f * sqrtf((c00 - c11) * (c00 - c11) + 4 * c01 * c01);
        const float l1 = (c00 + c11) / 2 + disc, l2 = (c00 + c11) / 2 - disc;
        float a0x = -c01, a0y = c00 - l1, a1x = -c01, a1y = c00 - l2;
        float mag = sqrtf(a0x * a0x + a0y * a0y); a0x /= mag; a0y /= mag;
        mag = sqrtf(a1x * a1x + a1y * a1y); a1x /= mag; a1y /= mag;
        float il = 1.0f / sqrtf(a0x * a0x + a0y * a0y); a0x = a0x * il; a0y = a0y * il;
        il = 1.0f / sqrtf(a1x * a1x + a1y * a1y); a1x = a1x * il; a1y = a1y * il;
        float minx = 3.402823466e+38f, miny = 3.402823466e+38f, maxx = -3.402823466e+38f, maxy = -3.402823466e+38f;
        for (int i = 0; i < n; ++i) {
            const float cx = a0x * px[i] + a0y * py[i], cy = a1x * px[i] + a1y * py[i];
            minx = fminf(minx, cx); miny = fminf(miny, cy); maxx = fmaxf(maxx, cx); maxy = fmaxf(maxy, cy);
        }
        const float ex = maxx - minx, ey = maxy - miny;
        area[which] = (ex < 0.00001f || ey < 0.00001f) ? 0.0f : ex * ey;
        } while (false);
    }
    __syncthreads();
    if (threadIdx.x == 0 && area[0] < a.areaThresh && area[1] < a.areaThresh) a.numFilt[prev] = 0;
}

// ------------------------------------------------------------------------------------------------ dense verify
struct CF { const float* depth; const float* campos; const float* normals; };

BF_DEV void projError(unsigned idx, unsigned W, unsigned H, float distThresh, float normalThresh, const m44& T, const m44& K, const CF& in,
                      const CF& model, float dmin, float dmax, float out[3]) {         // computeProjError :418-487
    out[0] = out[1] = out[2] = 0.0f;
    const float4 p = ((const float4*)in.campos)[idx];
    const float4 nin = ((const float4*)in.normals)[idx];
    const float dIn = in.depth[idx];
    if (p.x != BF_MINF && nin.x != BF_MINF && dIn >= dmin && dIn <= dmax) {
        const float pt[4] = {T.e[0] * p.x + T.e[1] * p.y + T.e[2] * p.z + T.e[3] * p.w, T.e[4] * p.x + T.e[5] * p.y + T.e[6] * p.z + T.e[7] * p.w,
                             T.e[8] * p.x + T.e[9] * p.y + T.e[10] * p.z + T.e[11] * p.w, T.e[12] * p.x + T.e[13] * p.y + T.e[14] * p.z + T.e[15] * p.w};
        const float nt[3] = {T.e[0] * nin.x + T.e[1] * nin.y + T.e[2] * nin.z + T.e[3] * 0.0f, T.e[4] * nin.x + T.e[5] * nin.y + T.e[6] * nin.z + T.e[7] * 0.0f,
                             T.e[8] * nin.x + T.e[9] * nin.y + T.e[10] * nin.z + T.e[11] * 0.0f};
        const float tx = K.e[0] * pt[0] + K.e[1] * pt[1] + K.e[2] * pt[2] + K.e[3] * 1.0f, ty = K.e[4] * pt[0] + K.e[5] * pt[1] + K.e[6] * pt[2] + K.e[7] * 1.0f,
                    tz = K.e[8] * pt[0] + K.e[9] * pt[1] + K.e[10] * pt[2] + K.e[11] * 1.0f;
        const int sx = f2i(roundf(tx / tz)), sy = f2i(roundf(ty / tz));
        if (sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H) {
            const float4 pT = ((const float4*)model.campos)[sy * W + sx];
            const float4 nT = ((const float4*)model.normals)[sy * W + sx];
            if (pT.x != BF_MINF && nT.x != BF_MINF) {
                const float dx = pt[0] - pT.x, dy = pt[1] - pT.y, dz = pt[2] - pT.z, dw = pt[3] - pT.w;
                const float d = sqrtf(dx * dx + dy * dy + dz * dz + dw * dw);
                const float dN = nt[0] * nT.x + nt[1] * nT.y + nt[2] * nT.z;
                const float projDepth = pt[2];
                const float tgtDepth = model.depth[sy * W + sx];
                if (tgtDepth >= dmin && tgtDepth <= dmax) {
                    const bool b = ((tgtDepth != BF_MINF && projDepth < tgtDepth) && d > distThresh);
                    if ((dN >= normalThresh && d <= distThresh) || b) {
                        const float z01 = (pt[2] - dmin) / (dmax - dmin);
                        out[0] = d;
                        out[1] = fmaxf(0.0f, 0.5f * ((1.0f - d / distThresh) + (1.0f - z01)));
                        out[2] = 1.0f;
                    }
                }
            }
        }
    }
}

struct VerifyArgs {
    uint32_t curFrame, startFrame, W, H; m44 K;
    int* numFilt; const m44* T; const bf_cached_frame* frames;
    float distThresh, normalThresh, errThresh, corrThresh, dmin, dmax;
    // trajectory mode
    uint32_t numImages; const int* validImages; const m44* trajectory; int* validOpt;
};

// The block sum of FilterMatchesByDenseVerifyCU_Kernel / VerifyTrajectoryCU_Kernel AS THE REFERENCE EXECUTES IT (pinned against its
// kernel, tests/test_ref_pin_cpu.py): block (W, ceil(H/32)), thread (x, ty) sums its 32 rows in order; warpReduceSum over the 32-lane warps
// of the linear thread id ty*W + x (a shuffle from outside the warp returns the caller's own value); the threads with threadIdx.x % 32 == 0
// add what they hold - for W = 80 lane 0 of warps 0-2 and lane 16 of warps 2-4, i.e. the upper 32 rows count once, of the lower rows
// columns 0-15 three times, 32-47 and 64-79 twice, the rest not at all.  Here: the per-pixel terms are evaluated by all threads in
// parallel into LDS, then DV_THREADS "virtual threads" replay that reduction (HIP's width-32 shuffles have the same out-of-range rule);
// the adders are summed in ascending thread order (the reference's atomicAdd order is arbitrary).
constexpr unsigned DV_THREADS = 1024;        // >= W * ceil(H/32) (checked on the host); 16 waves: five pixels per thread at 80 x 60 (round 5; 512 threads walked ten
                                             // dependent gather round trips each and the kernel took 78 us for 0.4 MB of input)
constexpr unsigned DV_MAX_PIX = 5120;        // per-pixel terms kept in LDS (80 x 60 = 4800 pixels: 60 KB); larger images recompute

BF_DEV bool denseVerifyPair(const VerifyArgs& a, const bf_cached_frame& fi, const bf_cached_frame& fm, const m44& T) {
    __shared__ float term[3][DV_MAX_PIX];
    __shared__ float held[3][DV_THREADS];
    const m44 Tinv = inverse44(T);
    const CF in = {fi.d_depthDownsampled, fi.d_cameraposDownsampled, fi.d_normalsDownsampled};
    const CF mo = {fm.d_depthDownsampled, fm.d_cameraposDownsampled, fm.d_normalsDownsampled};
    const unsigned total = a.W * a.H, rows = (a.H + 31u) / 32u, nt = a.W * rows, tid = threadIdx.x;
    const bool lds = total <= DV_MAX_PIX;
    if (lds) {
        for (unsigned idx = tid; idx < total; idx += DV_THREADS) {
            float x[3], y[3];
            projError(idx, a.W, a.H, a.distThresh, a.normalThresh, T, a.K, in, mo, a.dmin, a.dmax, x);
            projError(idx, a.W, a.H, a.distThresh, a.normalThresh, Tinv, a.K, mo, in, a.dmin, a.dmax, y);
#pragma unroll
            for (int k = 0; k < 3; ++k) term[k][idx] = x[k] + y[k];
        }
        __syncthreads();
    }
    float loc[3] = {0.0f, 0.0f, 0.0f};
    if (tid < nt) {
        const unsigned x = tid % a.W, ty = tid / a.W;
        for (unsigned i = 0; i < 32u; ++i) {
            const unsigned y = ty * 32u + i;
            if (y >= a.H) break;
            const unsigned idx = y * a.W + x;
            if (lds) {
#pragma unroll
                for (int k = 0; k < 3; ++k) loc[k] += term[k][idx];
            } else {
                float p[3], q[3];
                projError(idx, a.W, a.H, a.distThresh, a.normalThresh, T, a.K, in, mo, a.dmin, a.dmax, p);
                projError(idx, a.W, a.H, a.distThresh, a.normalThresh, Tinv, a.K, mo, in, a.dmin, a.dmax, q);
#pragma unroll
                for (int k = 0; k < 3; ++k) loc[k] += p[k] + q[k];
            }
        }
    }
#pragma unroll
    for (unsigned off = 16; off > 0; off >>= 1) {
        const bool ex = (tid & 31u) + off < 32u && tid + off < nt;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float o = __shfl_down(loc[k], off, 32); loc[k] += ex ? o : loc[k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) held[k][tid] = loc[k];
    __syncthreads();
    float tot[3] = {0.0f, 0.0f, 0.0f};
    // every thread forms the same total: the virtual threads v = ty * W + x with x % 32 == 0, in ascending v - walked as (ty, x) pairs (round 6: the loop ran over all
    // nt values of v with two run-time modulo operations each, on all 16 waves: three quarters of the kernel's 53 us)
    for (unsigned ty = 0; ty < rows; ++ty)
        for (unsigned x = 0; x < a.W; x += 32u) {
            const unsigned v = ty * a.W + x;
#pragma unroll
            for (int k = 0; k < 3; ++k) tot[k] += held[k][v];
        }
    const float err = tot[0] / tot[1];
    const float corr = 0.5f * tot[2] / (float)(a.W * a.H);
    return !(corr < a.corrThresh || err > a.errThresh || err != err);
}

__global__ __launch_bounds__(DV_THREADS) void k_filter_dense_verify(VerifyArgs a) {
    const uint32_t prev = blockIdx.x + a.startFrame;
    if (prev == a.curFrame) return;
    if (a.numFilt[prev] <= 0) return;
    const bool ok = denseVerifyPair(a, a.frames[prev], a.frames[a.curFrame], a.T[prev]);
    if (threadIdx.x == 0 && !ok) a.numFilt[prev] = 0;
}

// VerifyTrajectoryCU_Kernel :1036-1127 — the launch has N(N-1)/2 workgroups but decodes the pair as
// (block / N, block % N), so only part of the pairs is tested; kept exactly.
__global__ __launch_bounds__(DV_THREADS) void k_verify_trajectory(VerifyArgs a) {
    const uint32_t img0 = blockIdx.x / a.numImages, img1 = blockIdx.x % a.numImages;
    if (img0 >= img1) return;
    if (a.validImages[img0] == 0 || a.validImages[img1] == 0) return;
    const m44 T = mul44(inverse44(a.trajectory[img1]), a.trajectory[img0]);
    const bool ok = denseVerifyPair(a, a.frames[img0], a.frames[img1], T);
    if (threadIdx.x == 0 && !ok) a.validOpt[0] = 0;
}

// ------------------------------------------------------------------------------------------------ bookkeeping
// Pairs computed ahead of the previous frame's verdict (bf_siftmgr_set_pair_stage): what Bundler::matchAndFilter would not have matched at all
// (Bundler.cpp:126-129: the previous image is invalid) is cleared now that every flag of an earlier image is final
__global__ void k_commit_pairs(uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const int* validImages, int* numMatches, int* numFilt) {
    const uint32_t prev = startFrame + blockIdx.x * blockDim.x + threadIdx.x;
    if (prev < numFrames && prev != curFrame && validImages[prev] == 0) { numMatches[prev] = 0; numFilt[prev] = 0; }
}

// filterFrames (SIFTImageManager.cpp:551-575) on the device
__global__ void k_filter_frames(uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const int* numFilt, int* validImages, const int* numKeys,
                                FrameResult* res) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int connected = 0, last = -1;
    for (int i = (int)numFrames - 1; i >= (int)startFrame; --i)
        if (validImages[i] != 0 && numFilt[i] > 0 && i != (int)curFrame) { connected = 1; last = i; break; }
    validImages[curFrame] = connected;
    res->lastMatched = last; res->valid = connected; res->numKeysCur = numKeys[curFrame];
}

// AddCurrToResidualsCU :610-658; previous images in ascending order, only when the frame is connected
__global__ __launch_bounds__(1024) void k_add_residuals(uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const int* numFilt, const uint2* fidx,
                                                        const Key* keys, m44 Kinv, const int* validImages, bf_entry_j* glob, uint2* globKeys, int* globNum,
                                                        uint32_t maxResiduals, FrameResult* res) {
    __shared__ int part[1024];
    const uint32_t tid = threadIdx.x;
    const int base0 = *globNum;
    if (validImages[curFrame] == 0) { if (tid == 0) res->numResiduals = base0; return; }
    const uint32_t nPairs = numFrames - startFrame;
    const uint32_t chunk = (nPairs + 1023) / 1024;
    const uint32_t p0 = min(tid * chunk, nPairs), p1 = min(p0 + chunk, nPairs);
    int cnt = 0;
    for (uint32_t p = p0; p < p1; ++p) { const uint32_t prev = p + startFrame; if (prev != curFrame) cnt += max(numFilt[prev], 0); }
    part[tid] = cnt;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        int t = 0;
        if (tid >= off) t = part[tid - off];
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int pos = base0 + part[tid] - cnt;
    for (uint32_t p = p0; p < p1; ++p) {
        const uint32_t prev = p + startFrame;
        if (prev == curFrame) continue;
        const int n = max(numFilt[prev], 0);
        for (int k = 0; k < n; ++k, ++pos) {
            if ((uint32_t)pos >= maxResiduals) continue;
            const uint2 ki = fidx[prev * MAX_FILT + k];
            const f3 a = backProject(Kinv, keys[ki.x]), b = backProject(Kinv, keys[ki.y]);
            bf_entry_j e;
            e.imgIdx_i = prev; e.imgIdx_j = curFrame;
            e.pos_i[0] = a.x; e.pos_i[1] = a.y; e.pos_i[2] = a.z;
            e.pos_j[0] = b.x; e.pos_j[1] = b.y; e.pos_j[2] = b.z;
            glob[pos] = e; globKeys[pos] = ki;
        }
    }
    if (tid == 1023) { const int total = min(base0 + part[1023], (int)maxResiduals); *globNum = total; res->numResiduals = total; }
}

__global__ void k_invalidate_pair(bf_entry_j* glob, uint32_t n, uint32_t i, uint32_t j) {      // :692-704
    const uint32_t idx = blockDim.x * blockIdx.x + threadIdx.x;
    if (idx < n && glob[idx].imgIdx_i == i && glob[idx].imgIdx_j == j) { glob[idx].imgIdx_i = 0xFFFFFFFFu; glob[idx].imgIdx_j = 0xFFFFFFFFu; }
}

__global__ void k_check_invalid_simple(const int* numEntriesPerRow, int* validImages, uint32_t numVars) {   // :764-774
    const uint32_t idx = blockDim.x * blockIdx.x + threadIdx.x;
    if (idx < numVars && numEntriesPerRow[idx] == 0) validImages[idx] = 0;
}

// CheckForInvalidFramesCU_Kernel :725-744 with the launch of :746-749.  The reference's index arithmetic visits
// (residual, variable) pairs  res = bx*a + b (a < gx, b < 128),  var = gx*c + d (c < bx, d < 16)  with
// gx = ceil(R/128), bx = ceil(numVars/16); restated as: one thread per variable tests whether the variable is in
// that set, and one pass over the residuals invalidates those that touch a zero-row variable of the set.
__global__ void k_check_invalid(const int* numEntriesPerRow, int* validImages, uint32_t numVars, bf_entry_j* glob, uint32_t R, uint32_t gx, uint32_t bx) {
    const uint32_t t = blockDim.x * blockIdx.x + threadIdx.x;
    auto inVarSet = [&](uint32_t v) { for (uint32_t d = 0; d < 16 && d <= v; ++d) if ((v - d) % gx == 0 && (v - d) / gx < bx) return true; return false; };
    auto inResSet = [&](uint32_t r) { for (uint32_t b = 0; b < 128 && b <= r; ++b) if ((r - b) % bx == 0 && (r - b) / bx < gx) return true; return false; };
    if (t < R && inResSet(t)) {
        const bf_entry_j e = glob[t];
        if (e.imgIdx_i != 0xFFFFFFFFu) {
            bool kill = false;
            if (e.imgIdx_i < numVars && numEntriesPerRow[e.imgIdx_i] == 0 && inVarSet(e.imgIdx_i)) kill = true;
            if (e.imgIdx_j < numVars && numEntriesPerRow[e.imgIdx_j] == 0 && inVarSet(e.imgIdx_j)) kill = true;
            if (kill) { glob[t].imgIdx_i = 0xFFFFFFFFu; glob[t].imgIdx_j = 0xFFFFFFFFu; }
        }
    }
    if (t < numVars && numEntriesPerRow[t] == 0 && inVarSet(t)) validImages[t] = 0;
}

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_reset_valid(int* valid, uint32_t n, int* globNum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) valid[i] = i == 0 ? 1 : 0;
    if (i == 0) *globNum = 0;
}

template <class T>
int dalloc(T*& p, size_t n) { BF_HIP_TRY(BF_MALLOC((void**)&p, std::max<size_t>(n, 1) * sizeof(T))); return BF_OK; }

m44 toM44(const float* p) { m44 m; memcpy(m.e, p, 64); return m; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// fuseToGlobal on the device (SIFTImageManager.cpp:367-476: computeTracks + fuseToGlobal).  The reference copies all key points,
// descriptors and correspondences of the chunk to the host (1.6 MB), builds the tracks with a recursive depth-first search and
// uploads the fused key frame.  Here the chunk hand-off never leaves HBM: ONE workgroup
//   1. builds the adjacency lists of the key graph (both directions of every valid correspondence, each key's list in
//      correspondence order - the order the reference's corrPerKey vectors have),
//   2. labels the connected components with their smallest key index (the key the reference's search starts a track from: it
//      visits the keys in ascending order and a search from the first key of a component exhausts it),
//   3. runs the reference's depth-first search per component, one thread each - the SAME pre-order, because the track's first
//      element is its representative and the averaged position is a float sum taken in track order,
//   4. compacts the tracks in ascending start-key order into the new key frame of the global manager (key points, descriptors,
//      count - all written on the device; the depth sort of an over-full key frame included).
// Bit-identical with the host search (bf_siftmgr_fuse_to_global_host, kept for comparison: tests/test_match_gpu.py).
// ---------------------------------------------------------------------------------------------------------------------------
struct FuseAdj { uint32_t src, dst, img, order; float px, py, pz; uint32_t pad; };      // 32 B: edge src -> dst (key indices image * maxKeys + key)
struct FuseArgs {
    const bf_entry_j* glob; const uint2* globKeys; uint32_t R;
    const m44* T; const int* numKeys; const Key* keys; const uint8_t* descs; uint32_t nI, mk;
    m44 K;
    uint32_t* cnt; uint32_t* start; uint32_t* fill; FuseAdj* adj; uint32_t* label; uint32_t* marker; uint32_t* flag; uint32_t* rep; Key* tmpKeys; uint32_t* outPos;
    Key* dstKeys; uint8_t* dstDescs; int* dstNum; uint32_t dstMaxKeys; int* error;
};
constexpr int FUSE_THREADS = 1024;

BF_DEV uint32_t fuseBlockScan(uint32_t v, uint32_t* lds, uint32_t& total) {       // exclusive scan over the block's threads
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (uint32_t o = 1; o < (uint32_t)FUSE_THREADS; o <<= 1) {
        const uint32_t add = t >= o ? lds[t - o] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[FUSE_THREADS - 1];
    const uint32_t ex = lds[t] - v;
    __syncthreads();
    return ex;
}

__global__ __launch_bounds__(FUSE_THREADS) void k_fuse_to_global(FuseArgs a) {
    __shared__ uint32_t lds[FUSE_THREADS];
    __shared__ uint32_t changed;
    const uint32_t t = threadIdx.x, N = a.nI * a.mk;
    const float NINF_ = BF_MINF;
    for (uint32_t k = t; k < N; k += FUSE_THREADS) { a.cnt[k] = 0u; a.label[k] = k; a.marker[k] = 0u; a.flag[k] = 0u; }
    __syncthreads();
    // 1. adjacency: count, scan, fill, per-key order
    for (uint32_t i = t; i < a.R; i += FUSE_THREADS) {
        if (a.glob[i].imgIdx_i == 0xFFFFFFFFu) continue;
        const uint2 k = a.globKeys[i];
        atomicAdd(&a.cnt[k.x], 1u); atomicAdd(&a.cnt[k.y], 1u);
    }
    __syncthreads();
    const uint32_t per = (N + FUSE_THREADS - 1) / FUSE_THREADS;
    uint32_t mine = 0;
    for (uint32_t k = t * per; k < min(N, (t + 1) * per); ++k) mine += a.cnt[k];
    uint32_t totalE;
    uint32_t base = fuseBlockScan(mine, lds, totalE);
    for (uint32_t k = t * per; k < min(N, (t + 1) * per); ++k) { a.start[k] = base; a.fill[k] = base; base += a.cnt[k]; }
    __syncthreads();
    for (uint32_t i = t; i < a.R; i += FUSE_THREADS) {
        const bf_entry_j c = a.glob[i];
        if (c.imgIdx_i == 0xFFFFFFFFu) continue;
        const uint2 k = a.globKeys[i];
        const f3 pi = mk3(c.pos_i[0], c.pos_i[1], c.pos_i[2]), pj = mk3(c.pos_j[0], c.pos_j[1], c.pos_j[2]);
        const f3 d = xform(a.T[c.imgIdx_i], pi) - xform(a.T[c.imgIdx_j], pj);
        const bool ok = sqrtf(dot3(d, d)) < 0.03f;                                   // MAX_TRACK_CORR_ERROR
        FuseAdj e;
        e.pad = 0u;
        e.src = k.x; e.dst = k.y; e.img = c.imgIdx_j; e.order = i; e.px = ok ? pj.x : NINF_; e.py = ok ? pj.y : NINF_; e.pz = ok ? pj.z : NINF_;
        a.adj[atomicAdd(&a.fill[k.x], 1u)] = e;
        e.src = k.y; e.dst = k.x; e.img = c.imgIdx_i; e.px = ok ? pi.x : NINF_; e.py = ok ? pi.y : NINF_; e.pz = ok ? pi.z : NINF_;
        a.adj[atomicAdd(&a.fill[k.y], 1u)] = e;
    }
    __syncthreads();
    for (uint32_t k = t; k < N; k += FUSE_THREADS) {                                 // each key's list in correspondence order
        const uint32_t n = a.cnt[k], s0 = a.start[k];
        for (uint32_t i = 1; i < n; ++i) {
            const FuseAdj e = a.adj[s0 + i];
            uint32_t j = i;
            while (j > 0 && a.adj[s0 + j - 1].order > e.order) { a.adj[s0 + j] = a.adj[s0 + j - 1]; --j; }
            a.adj[s0 + j] = e;
        }
        a.fill[k] = 0u;                                                             // from here on: the search's position in this key's list
    }
    __syncthreads();
    // 2. connected components: label = smallest key index
    for (int it = 0; it < 4096; ++it) {
        if (t == 0) changed = 0u;
        __syncthreads();
        for (uint32_t e = t; e < totalE; e += FUSE_THREADS) {
            const uint32_t u = a.adj[e].src, v = a.adj[e].dst;
            const uint32_t lu = a.label[u], lv = a.label[v];
            if (lu < lv) { atomicMin(&a.label[v], lu); changed = 1u; }
            else if (lv < lu) { atomicMin(&a.label[u], lv); changed = 1u; }
        }
        __syncthreads();
        const uint32_t ch = changed;
        __syncthreads();
        if (!ch) break;
        if (it == 4095 && t == 0) *a.error = 2;          // labelling not converged within the sweep cap (a chain longer than 4096 keys cannot occur with <= 11 x 1024 keys per
                                                          // chunk and label propagation by atomicMin; reported, never silently split: bf_siftmgr_fuse_error)
    }
    // 3. the reference's depth-first search, one thread per component.  No stack: every key has ONE list position (a.fill) and the key it
    //    was entered from (a.outPos, free until step 4); `depth` counts the open calls.  The start key is entered a second time when one of
    //    its neighbours lists it (it is not marked at the start, like in the reference): the second visit continues the SAME list position -
    //    equivalent to the reference's nested call restarting at 0, because everything before that position is marked by then - and the
    //    search ends when the count of open calls returns to zero, whatever a.outPos of the start key says by then.
    for (uint32_t k = t; k < N; k += FUSE_THREADS) {
        if (a.cnt[k] == 0u || a.label[k] != k) continue;
        f3 pos = mk3(0, 0, 0);
        uint32_t num = 0, rep = 0xFFFFFFFFu, u = k;
        int depth = 1;
        while (depth > 0) {
            const uint32_t p = a.fill[u];
            if (p >= a.cnt[u]) { --depth; u = a.outPos[u]; continue; }
            a.fill[u] = p + 1;
            const FuseAdj e = a.adj[a.start[u] + p];
            if (a.marker[e.dst]) continue;
            if (rep == 0xFFFFFFFFu) rep = e.dst;
            if (e.px != NINF_) { pos = pos + xform(a.T[e.img], mk3(e.px, e.py, e.pz)); num++; }
            a.marker[e.dst] = 1u;
            a.outPos[e.dst] = u;
            u = e.dst; ++depth;
        }
        if (num > 0) {
            pos = pos / (float)num;
            pos = xform(a.K, pos);
            Key key;
            key.x = pos.x / pos.z; key.y = pos.y / pos.z; key.scale = a.keys[rep].scale; key.depth = pos.z;
            a.tmpKeys[k] = key; a.rep[k] = rep; a.flag[k] = 1u;
        }
    }
    __syncthreads();
    // 4. tracks in ascending start-key order -> the new key frame
    mine = 0;
    for (uint32_t k = t * per; k < min(N, (t + 1) * per); ++k) mine += a.flag[k];
    uint32_t M;
    base = fuseBlockScan(mine, lds, M);
    for (uint32_t k = t * per; k < min(N, (t + 1) * per); ++k) { a.outPos[k] = base; base += a.flag[k]; }
    __syncthreads();
    const uint32_t numOut = min(M, a.dstMaxKeys);
    if (M <= a.dstMaxKeys) {
        for (uint32_t k = t; k < N; k += FUSE_THREADS) if (a.flag[k]) a.dstKeys[a.outPos[k]] = a.tmpKeys[k];
    } else {
        // more tracks than the key frame holds: like the reference, the KEYS are sorted by depth (the descriptors keep the track order)
        // and the first maxKeys of each are kept.  rank = keys with smaller (depth, track position)
        for (uint32_t k = t; k < N; k += FUSE_THREADS) {
            if (!a.flag[k]) continue;
            const float dk = a.tmpKeys[k].depth; const uint32_t pk = a.outPos[k];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < N; ++j) if (a.flag[j]) { const float dj = a.tmpKeys[j].depth; rank += (dj < dk || (dj == dk && a.outPos[j] < pk)) ? 1u : 0u; }
            if (rank < a.dstMaxKeys) a.dstKeys[rank] = a.tmpKeys[k];
        }
    }
    // descriptors of the representatives: 128 B = 8 x 16 B each
    for (uint32_t k = t; k < N; k += FUSE_THREADS) {
        if (!a.flag[k] || a.outPos[k] >= numOut) continue;
        const uint4* src = reinterpret_cast<const uint4*>(a.descs + (size_t)a.rep[k] * 128);
        uint4* dst = reinterpret_cast<uint4*>(a.dstDescs + (size_t)a.outPos[k] * 128);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = src[q];
    }
    if (t == 0) *a.dstNum = (int)numOut;
}

struct bf_siftmgr {
    uint32_t maxImages = 0, maxKeys = 0, maxResiduals = 0;
    hipStream_t stream = nullptr;
    Key* d_keys = nullptr; uint8_t* d_descs = nullptr; int* d_numKeys = nullptr;
    // per-pair results of the current frame against every previous image (d_currNumMatchesPerImagePair ... d_currFilteredTransformsInv, SIFTImageManager.h:
    // 262-276) exist TWICE: bf_siftmgr_set_pair_stage selects the set (and the stream) the pair kernels of the NEXT frame work on, so that the match / Kabsch /
    // surface-area / dense-verification kernels of frame k + 1 run beside those of frame k (each pair depends on the two images only; which previous
    // images are valid is applied afterwards, bf_siftmgr_commit_pairs).  The members below always name the active set.
    struct PairSet { int* numMatches; float* dist; uint2* idx; int* numFilt; float* fdist; uint2* fidx; m44* T; m44* Tinv; MatchScratch ms; };
    PairSet sets[2] = {};
    int pairSet = 0;
    hipStream_t pairStream = nullptr;      // stream of the pair kernels; null: `stream`
    bool speculative = false;
    int* d_numMatches = nullptr; float* d_dist = nullptr; uint2* d_idx = nullptr;
    int* d_numFilt = nullptr; float* d_fdist = nullptr; uint2* d_fidx = nullptr; m44* d_T = nullptr; m44* d_Tinv = nullptr;
    MatchScratch matchScratch{};           // of the active pair set
    int* d_validImages = nullptr; int* d_validOpt = nullptr;
    bf_entry_j* d_glob = nullptr; uint2* d_globKeys = nullptr; int* d_globNum = nullptr;
    // the frame's single read-back.  Up to RES_SLOTS read-backs may be in flight (bf_siftmgr_prefetch_frame_result enqueues one behind the work issued so
    // far, bf_siftmgr_sync_frame_result consumes the oldest): the frame loop enqueues the matching chain of frame k + 1 before it waits for frame k.
    static constexpr int RES_SLOTS = 4;
    FrameResult* d_res = nullptr; FrameResult* h_res = nullptr;      // h_res: RES_SLOTS pinned records
    hipEvent_t evRes[RES_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int resHead = 0, resCount = 0;
    std::vector<int> validImages;
    uint32_t numImages = 0, currentImage = 0, globNumResiduals = 0;
    bool finalized = true;
    bool validDirty = false;          // host copy of the valid flags changed since the last upload
    std::deque<uint32_t> retry;
    // scratch of the device-side fuseToGlobal (allocated at first use)
    void* fuseScratch = nullptr; size_t fuseScratchBytes = 0; int* d_fuseError = nullptr;
};

static void selectPairSet(bf_siftmgr* m, int k) {
    const bf_siftmgr::PairSet& P = m->sets[k];
    m->pairSet = k;
    m->d_numMatches = P.numMatches; m->d_dist = P.dist; m->d_idx = P.idx;
    m->d_numFilt = P.numFilt; m->d_fdist = P.fdist; m->d_fidx = P.fidx; m->d_T = P.T; m->d_Tinv = P.Tinv;
    m->matchScratch = P.ms;
}
static hipStream_t pairStreamOf(const bf_siftmgr* m) { return m->pairStream ? m->pairStream : m->stream; }

extern "C" {

// The pair kernels (match, Kabsch filter, surface-area filter, dense verification) issued from now on work on result set `set` (0 / 1) and on `hip_stream`
// (null: the manager's stream); speculative != 0: they do not consult the valid flags of the previous images (those may still be in the making on the
// manager's stream) - bf_siftmgr_commit_pairs, on the manager's stream, clears the pairs Bundler::matchAndFilter would have skipped.  Ordering between the
// two streams is the caller's (events).  The accessors of the per-pair arrays name the selected set.  (0, null, 0) is the reference's behaviour.
int bf_siftmgr_set_pair_stage(bf_siftmgr* m, uint32_t set, void* hip_stream, int speculative) {
    BF_REQUIRE(m && set < 2, "bad argument");
    selectPairSet(m, (int)set);
    m->pairStream = (hipStream_t)hip_stream;
    m->speculative = speculative != 0;
    return BF_OK;
}
int bf_siftmgr_commit_pairs(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames) {
    BF_REQUIRE(m && numFrames <= m->numImages && startFrame < numFrames, "frame range out of bounds");
    k_commit_pairs<<<div_up(numFrames - startFrame, 64), 64, 0, m->stream>>>(curFrame, startFrame, numFrames, m->d_validImages, m->d_numMatches, m->d_numFilt);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_create(uint32_t maxImages, uint32_t maxKeyPointsPerImage, bf_siftmgr** out) {
    BF_REQUIRE(out && maxImages >= 1 && maxKeyPointsPerImage >= 16 && maxKeyPointsPerImage <= 1024, "maxKeyPointsPerImage must be in [16, 1024]");
    bf_siftmgr* m = new bf_siftmgr;
    m->maxImages = maxImages; m->maxKeys = maxKeyPointsPerImage;
    m->maxResiduals = MAX_FILT * (maxImages * (maxImages - 1)) / 2;
    int rc;
    const size_t nk = (size_t)maxImages * maxKeyPointsPerImage;
    rc = BF_OK;
    for (int k = 0; k < 2 && !rc; ++k) {
        bf_siftmgr::PairSet& P = m->sets[k];
        (rc = dalloc(P.numMatches, maxImages)) || (rc = dalloc(P.dist, (size_t)maxImages * MAX_RAW)) || (rc = dalloc(P.idx, (size_t)maxImages * MAX_RAW)) ||
        (rc = dalloc(P.numFilt, maxImages)) || (rc = dalloc(P.fdist, (size_t)maxImages * MAX_FILT)) || (rc = dalloc(P.fidx, (size_t)maxImages * MAX_FILT)) ||
        (rc = dalloc(P.T, maxImages)) || (rc = dalloc(P.Tinv, maxImages)) ||
        (rc = dalloc(P.ms.rowRes, (size_t)maxImages * 1024)) || (rc = dalloc(P.ms.rowDist, (size_t)maxImages * 1024)) || (rc = dalloc(P.ms.colRes, (size_t)maxImages * 1024)) ||
        (rc = dalloc(P.ms.ticket, maxImages));
        if (!rc) { BF_HIP_TRY(hipMemset(P.numMatches, 0, sizeof(int) * maxImages)); BF_HIP_TRY(hipMemset(P.numFilt, 0, sizeof(int) * maxImages)); BF_HIP_TRY(hipMemset(P.ms.ticket, 0, sizeof(uint32_t) * maxImages)); }
    }
    if (rc) { delete m; return rc; }
    selectPairSet(m, 0);
    if ((rc = dalloc(m->d_keys, nk)) || (rc = dalloc(m->d_descs, nk * 128)) || (rc = dalloc(m->d_numKeys, maxImages)) ||
        (rc = dalloc(m->d_validImages, maxImages)) || (rc = dalloc(m->d_validOpt, 1)) ||
        (rc = dalloc(m->d_glob, m->maxResiduals)) || (rc = dalloc(m->d_globKeys, m->maxResiduals)) || (rc = dalloc(m->d_globNum, 1)) || (rc = dalloc(m->d_res, 1))) {
        delete m; return rc;
    }
    BF_HIP_TRY(hipHostMalloc((void**)&m->h_res, sizeof(FrameResult) * bf_siftmgr::RES_SLOTS));
    for (auto& e : m->evRes) BF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BF_HIP_TRY(hipMemset(m->d_numKeys, 0, sizeof(int) * maxImages));
    BF_HIP_TRY(hipMemset(m->d_globNum, 0, sizeof(int)));
    BF_HIP_TRY(hipMemset(m->d_res, 0, sizeof(FrameResult)));
    m->validImages.assign(maxImages, 0);
    m->validImages[0] = 1;
    BF_HIP_TRY(hipMemcpy(m->d_validImages, m->validImages.data(), sizeof(int) * maxImages, hipMemcpyHostToDevice));
    *out = m;
    return BF_OK;
}

int bf_siftmgr_destroy(bf_siftmgr* m) {
    if (!m) return BF_OK;
    hipFree(m->d_keys); hipFree(m->d_descs); hipFree(m->d_numKeys);
    for (auto& P : m->sets) { hipFree(P.numMatches); hipFree(P.dist); hipFree(P.idx); hipFree(P.numFilt); hipFree(P.fdist); hipFree(P.fidx); hipFree(P.T); hipFree(P.Tinv);
                              hipFree(P.ms.rowRes); hipFree(P.ms.rowDist); hipFree(P.ms.colRes); hipFree(P.ms.ticket); }
    hipFree(m->d_validImages); hipFree(m->d_validOpt);
    hipFree(m->d_glob); hipFree(m->d_globKeys); hipFree(m->d_globNum); hipFree(m->d_res);
    if (m->fuseScratch) hipFree(m->fuseScratch);
    if (m->d_fuseError) hipFree(m->d_fuseError);
    if (m->h_res) hipHostFree(m->h_res);
    for (auto& e : m->evRes) if (e) (void)hipEventDestroy(e);
    delete m;
    return BF_OK;
}

int bf_siftmgr_set_stream(bf_siftmgr* m, void* s) { BF_REQUIRE(m, "null manager"); m->stream = (hipStream_t)s; return BF_OK; }

int bf_siftmgr_reset(bf_siftmgr* m) {                 // SIFTImageManager.h:112-124
    BF_REQUIRE(m, "null manager");
    m->numImages = 0; m->currentImage = 0; m->globNumResiduals = 0; m->finalized = true;
    m->validImages.assign(m->maxImages, 0);
    m->validImages[0] = 1;
    m->validDirty = false;
    k_reset_valid<<<div_up(m->maxImages, 256), 256, 0, m->stream>>>(m->d_validImages, m->maxImages, m->d_globNum);     // asynchronous: no host wait
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_create_image(bf_siftmgr* m, bf_sift_image_gpu* out) {     // createSIFTImageGPU .cpp:44-60
    BF_REQUIRE(m && out, "null argument");
    BF_REQUIRE(m->finalized, "previous image not finalized");
    BF_REQUIRE(m->numImages < m->maxImages, "image capacity exceeded");
    const uint32_t i = m->numImages++;
    out->d_keyPoints = (bf_sift_keypoint*)(m->d_keys + (size_t)i * m->maxKeys);
    out->d_keyPointDescs = (bf_sift_keypoint_desc*)(m->d_descs + (size_t)i * m->maxKeys * 128);
    out->d_numKeyPoints = m->d_numKeys + i;
    m->finalized = false;
    return BF_OK;
}

int bf_siftmgr_finalize_image(bf_siftmgr* m, int32_t numKeyPoints) {      // finalizeSIFTImageGPU .cpp:62-75
    BF_REQUIRE(m && !m->finalized, "no image under construction");
    BF_REQUIRE(numKeyPoints <= (int32_t)m->maxKeys, "too many keypoints");
    if (numKeyPoints >= 0) {         // host-known count; a negative value keeps the count the detector wrote on the device
        k_set_int<<<1, 1, 0, m->stream>>>(m->d_numKeys + (m->numImages - 1), numKeyPoints);
        BF_HIP_TRY(hipGetLastError());
    }
    m->finalized = true;
    m->currentImage = m->numImages - 1;
    return BF_OK;
}

int bf_siftmgr_get_image(bf_siftmgr* m, uint32_t i, bf_sift_image_gpu* out) {
    BF_REQUIRE(m && out && i < m->numImages, "image index out of range");
    out->d_keyPoints = (bf_sift_keypoint*)(m->d_keys + (size_t)i * m->maxKeys);
    out->d_keyPointDescs = (bf_sift_keypoint_desc*)(m->d_descs + (size_t)i * m->maxKeys * 128);
    out->d_numKeyPoints = m->d_numKeys + i;
    return BF_OK;
}

int bf_siftmgr_get_num_images(bf_siftmgr* m, uint32_t* out) { BF_REQUIRE(m && out, "null argument"); *out = m->numImages; return BF_OK; }
int bf_siftmgr_get_max_num_keypoints_per_image(bf_siftmgr* m, uint32_t* out) { BF_REQUIRE(m && out, "null argument"); *out = m->maxKeys; return BF_OK; }
int bf_siftmgr_get_current_frame(bf_siftmgr* m, uint32_t* out) { BF_REQUIRE(m && out, "null argument"); *out = m->currentImage; return BF_OK; }
int bf_siftmgr_set_current_frame(bf_siftmgr* m, uint32_t i) { BF_REQUIRE(m, "null manager"); m->currentImage = i; return BF_OK; }

int bf_siftmgr_get_num_keypoints(bf_siftmgr* m, uint32_t first, uint32_t count, int32_t* h_out) {
    BF_REQUIRE(m && h_out && first + count <= m->maxImages, "range out of bounds");
    BF_HIP_TRY(hipMemcpyAsync(h_out, m->d_numKeys + first, sizeof(int) * count, hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    return BF_OK;
}

int bf_siftmgr_match(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, float distMax, float ratioMax) {
    BF_REQUIRE(m && numFrames <= m->numImages && curFrame < numFrames && startFrame < numFrames, "frame range out of bounds");
    MatchArgs a = {m->d_descs, m->d_numKeys, m->d_validImages, m->maxKeys, curFrame, startFrame, distMax, ratioMax, m->d_numMatches, m->d_dist, m->d_idx, m->speculative ? 1 : 0};
    k_match<<<dim3(numFrames - startFrame, 32), 256, 0, pairStreamOf(m)>>>(a, m->matchScratch);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_filter_keypoint_matches(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const float siftIntrinsicsInv[16],
                                       uint32_t minNumMatches, float maxKabschRes2) {
    BF_REQUIRE(m && siftIntrinsicsInv && numFrames <= m->numImages && startFrame < numFrames, "frame range out of bounds");
    FilterArgs a = {m->d_keys, curFrame, startFrame, m->d_numMatches, m->d_dist, m->d_idx, m->d_numFilt, m->d_fdist, m->d_fidx, m->d_T, m->d_Tinv,
                    toM44(siftIntrinsicsInv), (int)minNumMatches, maxKabschRes2};
    k_filter_kabsch<<<numFrames - startFrame, 64, 0, pairStreamOf(m)>>>(a);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_filter_matches_by_surface_area(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const float colorIntrinsicsInv[16],
                                              float areaThresh) {
    BF_REQUIRE(m && colorIntrinsicsInv && numFrames <= m->numImages && startFrame < numFrames, "frame range out of bounds");
    AreaArgs a = {m->d_keys, curFrame, startFrame, m->d_numFilt, m->d_fidx, toM44(colorIntrinsicsInv), areaThresh};
    k_filter_surface_area<<<numFrames - startFrame, 64, 0, pairStreamOf(m)>>>(a);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_filter_matches_by_dense_verify(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, uint32_t imageWidth,
                                              uint32_t imageHeight, const float intrinsics[16], const bf_cached_frame* d_cachedFrames, float distThresh,
                                              float normalThresh, float colorThresh, float errThresh, float corrThresh, float sensorDepthMin,
                                              float sensorDepthMax) {
    (void)colorThresh;      // computeProjError never reads it (:418-487)
    BF_REQUIRE(m && intrinsics && d_cachedFrames && numFrames <= m->numImages && startFrame < numFrames, "frame range out of bounds");
    VerifyArgs a = {};
    a.curFrame = curFrame; a.startFrame = startFrame; a.W = imageWidth; a.H = imageHeight; a.K = toM44(intrinsics);
    a.numFilt = m->d_numFilt; a.T = m->d_T; a.frames = d_cachedFrames;
    a.distThresh = distThresh; a.normalThresh = normalThresh; a.errThresh = errThresh; a.corrThresh = corrThresh; a.dmin = sensorDepthMin; a.dmax = sensorDepthMax;
    BF_REQUIRE(imageWidth * ((imageHeight + 31u) / 32u) <= DV_THREADS, "cache frame too large for the dense verification block");
    k_filter_dense_verify<<<numFrames - startFrame, DV_THREADS, 0, pairStreamOf(m)>>>(a);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_filter_frames_async(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames) {
    BF_REQUIRE(m && numFrames <= m->numImages && curFrame < m->maxImages, "frame range out of bounds");
    k_filter_frames<<<1, 1, 0, m->stream>>>(curFrame, startFrame, numFrames, m->d_numFilt, m->d_validImages, m->d_numKeys, m->d_res);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

int bf_siftmgr_add_curr_to_residuals(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, const float colorIntrinsicsInv[16]) {
    BF_REQUIRE(m && colorIntrinsicsInv && numFrames <= m->numImages && startFrame < numFrames, "frame range out of bounds");
    k_add_residuals<<<1, 1024, 0, m->stream>>>(curFrame, startFrame, numFrames, m->d_numFilt, m->d_fidx, m->d_keys, toM44(colorIntrinsicsInv),
                                               m->d_validImages, m->d_glob, m->d_globKeys, m->d_globNum, m->maxResiduals, m->d_res);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

// enqueue the D2H of the frame result behind the work issued so far (optional; lets the caller do other things - including enqueueing the NEXT
// frame's chain on the same stream - before it waits)
int bf_siftmgr_prefetch_frame_result(bf_siftmgr* m) {
    BF_REQUIRE(m, "null manager");
    BF_REQUIRE(m->resCount < bf_siftmgr::RES_SLOTS, "too many frame results in flight");
    const int slot = (m->resHead + m->resCount) % bf_siftmgr::RES_SLOTS;
    BF_HIP_TRY(hipMemcpyAsync(m->h_res + slot, m->d_res, sizeof(FrameResult), hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipEventRecord(m->evRes[slot], m->stream));
    m->resCount++;
    return BF_OK;
}

// the frame's single read-back: last matched frame, validity, #residuals, #keys of the current frame.  Waits for the OLDEST prefetched result
// (an event behind its copy, not the whole stream: later frames' work may already be queued behind it).
int bf_siftmgr_sync_frame_result(bf_siftmgr* m, uint32_t curFrame, uint32_t* lastMatchedFrame, int32_t* numKeysCur) {
    BF_REQUIRE(m && curFrame < m->maxImages, "frame out of range");
    if (m->resCount == 0) { const int rc = bf_siftmgr_prefetch_frame_result(m); if (rc != BF_OK) return rc; }
    const int slot = m->resHead;
    BF_HIP_TRY(hipEventSynchronize(m->evRes[slot]));
    m->resHead = (m->resHead + 1) % bf_siftmgr::RES_SLOTS; m->resCount--;
    const FrameResult r = m->h_res[slot];
    m->validImages[curFrame] = r.valid;
    m->globNumResiduals = (uint32_t)r.numResiduals;
    if (lastMatchedFrame) *lastMatchedFrame = (uint32_t)r.lastMatched;
    if (numKeysCur) *numKeysCur = r.numKeysCur;
    return BF_OK;
}

// the device record behind it, for kernels that act on a frame's result without a host round trip: 4 x int32 {lastMatched (-1: none), valid, numResiduals, numKeysCur}
int bf_siftmgr_get_frame_result_gpu(bf_siftmgr* m, const int32_t** d_out) { BF_REQUIRE(m && d_out, "null argument"); *d_out = reinterpret_cast<const int32_t*>(m->d_res); return BF_OK; }

int bf_siftmgr_filter_frames(bf_siftmgr* m, uint32_t curFrame, uint32_t startFrame, uint32_t numFrames, uint32_t* lastMatchedFrame) {
    BF_REQUIRE(lastMatchedFrame, "null output");
    if (numFrames == 0) { *lastMatchedFrame = 0xFFFFFFFFu; return BF_OK; }
    BF_REQUIRE(m->resCount == 0, "a prefetched frame result is still pending");
    int rc = bf_siftmgr_filter_frames_async(m, curFrame, startFrame, numFrames);
    if (rc) return rc;
    BF_HIP_TRY(hipMemcpyAsync(m->h_res, m->d_res, sizeof(FrameResult), hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    m->validImages[curFrame] = m->h_res->valid;
    *lastMatchedFrame = (uint32_t)m->h_res->lastMatched;
    return BF_OK;
}

int bf_siftmgr_invalidate_image_to_image(bf_siftmgr* m, uint32_t i, uint32_t j) {
    BF_REQUIRE(m, "null manager");
    if (m->globNumResiduals == 0) return BF_OK;
    k_invalidate_pair<<<div_up(m->globNumResiduals, 128), 128, 0, m->stream>>>(m->d_glob, m->globNumResiduals, i, j);
    BF_HIP_TRY(hipGetLastError());
    return BF_OK;
}

static int pull_valid(bf_siftmgr* m, uint32_t numVars) {
    BF_HIP_TRY(hipMemcpyAsync(m->validImages.data(), m->d_validImages, sizeof(int) * numVars, hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    return BF_OK;
}

int bf_siftmgr_check_for_invalid_frames_simple(bf_siftmgr* m, const int32_t* d_varToCorrNumEntriesPerRow, uint32_t numVars) {
    BF_REQUIRE(m && d_varToCorrNumEntriesPerRow && numVars <= m->maxImages, "bad argument");
    BF_HIP_TRY(hipMemcpyAsync(m->d_validImages, m->validImages.data(), sizeof(int) * numVars, hipMemcpyHostToDevice, m->stream));
    k_check_invalid_simple<<<div_up(numVars, 64), 64, 0, m->stream>>>(d_varToCorrNumEntriesPerRow, m->d_validImages, numVars);
    BF_HIP_TRY(hipGetLastError());
    return pull_valid(m, numVars);
}

int bf_siftmgr_check_for_invalid_frames(bf_siftmgr* m, const int32_t* d_varToCorrNumEntriesPerRow, uint32_t numVars) {
    BF_REQUIRE(m && d_varToCorrNumEntriesPerRow && numVars <= m->maxImages, "bad argument");
    const uint32_t R = m->globNumResiduals;
    if (R == 0 || numVars == 0) return BF_OK;         // the reference's launch has an empty grid here
    BF_HIP_TRY(hipMemcpyAsync(m->d_validImages, m->validImages.data(), sizeof(int) * numVars, hipMemcpyHostToDevice, m->stream));
    const uint32_t gx = div_up(R, 128), bx = div_up(numVars, 16);
    k_check_invalid<<<div_up(std::max(R, numVars), 128), 128, 0, m->stream>>>(d_varToCorrNumEntriesPerRow, m->d_validImages, numVars, m->d_glob, R, gx, bx);
    BF_HIP_TRY(hipGetLastError());
    return pull_valid(m, numVars);
}

int bf_siftmgr_verify_trajectory(bf_siftmgr* m, uint32_t numImages, const float* d_trajectory, uint32_t imageWidth, uint32_t imageHeight,
                                 const float intrinsics[16], const bf_cached_frame* d_cachedFrames, float distThresh, float normalThresh, float colorThresh,
                                 float errThresh, float corrThresh, float sensorDepthMin, float sensorDepthMax, int32_t* valid) {
    (void)colorThresh;
    BF_REQUIRE(m && valid && d_trajectory && intrinsics && d_cachedFrames && numImages <= m->maxImages, "bad argument");
    if (numImages < 2) { *valid = 0; return BF_OK; }
    const uint32_t numPairs = (numImages * (numImages - 1)) / 2;
    BF_HIP_TRY(hipMemcpyAsync(m->d_validImages, m->validImages.data(), sizeof(int) * numImages, hipMemcpyHostToDevice, m->stream));
    k_set_int<<<1, 1, 0, m->stream>>>(m->d_validOpt, 1);
    VerifyArgs a = {};
    a.W = imageWidth; a.H = imageHeight; a.K = toM44(intrinsics); a.frames = d_cachedFrames;
    a.distThresh = distThresh; a.normalThresh = normalThresh; a.errThresh = errThresh; a.corrThresh = corrThresh; a.dmin = sensorDepthMin; a.dmax = sensorDepthMax;
    a.numImages = numImages; a.validImages = m->d_validImages; a.trajectory = (const m44*)d_trajectory; a.validOpt = m->d_validOpt;
    BF_REQUIRE(imageWidth * ((imageHeight + 31u) / 32u) <= DV_THREADS, "cache frame too large for the dense verification block");
    k_verify_trajectory<<<numPairs, DV_THREADS, 0, m->stream>>>(a);
    BF_HIP_TRY(hipGetLastError());
    int v = 0;
    BF_HIP_TRY(hipMemcpyAsync(&v, m->d_validOpt, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    *valid = v;
    return BF_OK;
}

int bf_siftmgr_get_valid_images(bf_siftmgr* m, int32_t* h_out, uint32_t count) {
    BF_REQUIRE(m && h_out && count <= m->maxImages, "bad argument");
    memcpy(h_out, m->validImages.data(), sizeof(int) * count);
    return BF_OK;
}
int bf_siftmgr_set_valid_image(bf_siftmgr* m, uint32_t frame, int32_t valid) {     // invalidateFrame / setValidImagesDEBUG
    BF_REQUIRE(m && frame < m->maxImages, "frame out of range");
    if (m->validImages[frame] != valid) m->validDirty = true;
    m->validImages[frame] = valid;
    return BF_OK;
}
int bf_siftmgr_update_gpu_valid_images(bf_siftmgr* m) {                            // updateGPUValidImages .h:160-162
    BF_REQUIRE(m, "null manager");
    if (m->numImages == 0 || !m->validDirty) return BF_OK;       // device flags already mirror the host's
    BF_HIP_TRY(hipMemcpyAsync(m->d_validImages, m->validImages.data(), sizeof(int) * m->numImages, hipMemcpyHostToDevice, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    m->validDirty = false;
    return BF_OK;
}
int bf_siftmgr_get_valid_images_gpu(bf_siftmgr* m, const int32_t** d_out) { BF_REQUIRE(m && d_out, "null argument"); *d_out = m->d_validImages; return BF_OK; }

int bf_siftmgr_get_global_correspondences_gpu(bf_siftmgr* m, bf_entry_j** d_out) { BF_REQUIRE(m && d_out, "null argument"); *d_out = m->d_glob; return BF_OK; }
int bf_siftmgr_get_global_correspondence_keys_gpu(bf_siftmgr* m, const uint32_t** d_out) { BF_REQUIRE(m && d_out, "null argument"); *d_out = (const uint32_t*)m->d_globKeys; return BF_OK; }
int bf_siftmgr_get_num_global_correspondences(bf_siftmgr* m, uint32_t* out) { BF_REQUIRE(m && out, "null argument"); *out = m->globNumResiduals; return BF_OK; }
int bf_siftmgr_set_global_correspondences(bf_siftmgr* m, const bf_entry_j* h_corr, uint32_t n) {      // setGlobalCorrespondencesDEBUG .h:182-188
    BF_REQUIRE(m && (h_corr || n == 0) && n <= m->maxResiduals, "too many correspondences");
    if (n) BF_HIP_TRY(hipMemcpyAsync(m->d_glob, h_corr, sizeof(bf_entry_j) * n, hipMemcpyHostToDevice, m->stream));
    const int c = (int)n;
    BF_HIP_TRY(hipMemcpyAsync(m->d_globNum, &c, sizeof(int), hipMemcpyHostToDevice, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    m->globNumResiduals = n;
    return BF_OK;
}
int bf_siftmgr_get_filt_transforms_gpu(bf_siftmgr* m, const float** d_transforms, const float** d_transformsInv) {
    BF_REQUIRE(m, "null manager");
    if (d_transforms) *d_transforms = (const float*)m->d_T;
    if (d_transformsInv) *d_transformsInv = (const float*)m->d_Tinv;
    return BF_OK;
}
int bf_siftmgr_get_num_filt_matches_gpu(bf_siftmgr* m, const int32_t** d_out) { BF_REQUIRE(m && d_out, "null argument"); *d_out = m->d_numFilt; return BF_OK; }
int bf_siftmgr_get_keys_gpu(bf_siftmgr* m, const bf_sift_keypoint** d_keys, const bf_sift_keypoint_desc** d_descs, const int32_t** d_numKeys) {
    BF_REQUIRE(m, "null manager");
    if (d_keys) *d_keys = (const bf_sift_keypoint*)m->d_keys;
    if (d_descs) *d_descs = (const bf_sift_keypoint_desc*)m->d_descs;
    if (d_numKeys) *d_numKeys = m->d_numKeys;
    return BF_OK;
}

int bf_siftmgr_get_curr_matches_gpu(bf_siftmgr* m, int filtered, const uint32_t** d_keyPointIndices, const int32_t** d_numMatches) {
    BF_REQUIRE(m && d_keyPointIndices && d_numMatches, "null argument");
    *d_keyPointIndices = (const uint32_t*)(filtered ? m->d_fidx : m->d_idx);
    *d_numMatches = filtered ? m->d_numFilt : m->d_numMatches;
    return BF_OK;
}
int bf_siftmgr_get_raw_matches(bf_siftmgr* m, uint32_t imagePairIndex, int32_t* numMatches, uint32_t* h_keyPointIndices, float* h_distances) {
    BF_REQUIRE(m && numMatches && imagePairIndex < m->maxImages, "bad argument");
    BF_HIP_TRY(hipMemcpyAsync(numMatches, m->d_numMatches + imagePairIndex, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    if (h_keyPointIndices) BF_HIP_TRY(hipMemcpyAsync(h_keyPointIndices, m->d_idx + (size_t)imagePairIndex * MAX_RAW, sizeof(uint2) * MAX_RAW, hipMemcpyDeviceToHost, m->stream));
    if (h_distances) BF_HIP_TRY(hipMemcpyAsync(h_distances, m->d_dist + (size_t)imagePairIndex * MAX_RAW, sizeof(float) * MAX_RAW, hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    return BF_OK;
}
int bf_siftmgr_get_filt_matches(bf_siftmgr* m, uint32_t imagePairIndex, int32_t* numMatches, uint32_t* h_keyPointIndices, float* h_distances,
                                float* h_transform, float* h_transformInv) {
    BF_REQUIRE(m && numMatches && imagePairIndex < m->maxImages, "bad argument");
    BF_HIP_TRY(hipMemcpyAsync(numMatches, m->d_numFilt + imagePairIndex, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    if (h_keyPointIndices) BF_HIP_TRY(hipMemcpyAsync(h_keyPointIndices, m->d_fidx + (size_t)imagePairIndex * MAX_FILT, sizeof(uint2) * MAX_FILT, hipMemcpyDeviceToHost, m->stream));
    if (h_distances) BF_HIP_TRY(hipMemcpyAsync(h_distances, m->d_fdist + (size_t)imagePairIndex * MAX_FILT, sizeof(float) * MAX_FILT, hipMemcpyDeviceToHost, m->stream));
    if (h_transform) BF_HIP_TRY(hipMemcpyAsync(h_transform, m->d_T + imagePairIndex, 64, hipMemcpyDeviceToHost, m->stream));
    if (h_transformInv) BF_HIP_TRY(hipMemcpyAsync(h_transformInv, m->d_Tinv + imagePairIndex, 64, hipMemcpyDeviceToHost, m->stream));
    BF_HIP_TRY(hipStreamSynchronize(m->stream));
    return BF_OK;
}

int bf_siftmgr_add_to_retry_list(bf_siftmgr* m, uint32_t idx) { BF_REQUIRE(m, "null manager"); m->retry.push_front(idx); return BF_OK; }
int bf_siftmgr_get_top_retry_image(bf_siftmgr* m, uint32_t* idx, int* found) {
    BF_REQUIRE(m && idx && found, "null argument");
    if (m->retry.empty()) { *found = 0; return BF_OK; }
    *idx = m->retry.front(); m->retry.pop_front(); *found = 1;
    return BF_OK;
}

// fuseToGlobal (SIFTImageManager.cpp:367-476): chunk -> one global key frame.  Host DFS over the valid
// correspondences exactly like the reference (it runs once per chunk on <= 11 x 1024 keys).
namespace {
typedef std::pair<uint2, f3> TrackEl;
void findTrack(const std::vector<std::vector<TrackEl>>& corrPerKey, std::vector<char>& marker, std::vector<TrackEl>& track, uint32_t curKey) {
    std::vector<std::pair<uint32_t, size_t>> stack;          // explicit stack, same pre-order as the recursion
    stack.emplace_back(curKey, 0);
    while (!stack.empty()) {
        auto& top = stack.back();
        if (top.second >= corrPerKey[top.first].size()) { stack.pop_back(); continue; }
        const TrackEl& c = corrPerKey[top.first][top.second++];
        if (!marker[c.first.y]) {
            track.push_back(c);
            marker[c.first.y] = 1;
            stack.emplace_back(c.first.y, 0);
        }
    }
}
}  // namespace

int bf_siftmgr_fuse_to_global(bf_siftmgr* local, bf_siftmgr* global, const float colorIntrinsics[16], const float* d_transforms,
                              const float colorIntrinsicsInv[16]) {
    BF_REQUIRE(local && global && colorIntrinsics && d_transforms, "null argument");
    BF_REQUIRE(local->globNumResiduals > 0, "no correspondences to fuse");
    const uint32_t R = local->globNumResiduals, nI = local->numImages, mk = local->maxKeys;
    const size_t N = (size_t)nI * mk;
    // scratch: 8 arrays of N words + N key points + 2R edges
    const size_t need = N * 4 * 9 + N * sizeof(Key) + (size_t)2 * local->maxResiduals * sizeof(FuseAdj) + 256;
    if (local->fuseScratchBytes < need) {
        if (local->fuseScratch) BF_HIP_TRY(hipFree(local->fuseScratch));
        local->fuseScratch = nullptr; local->fuseScratchBytes = 0;
        const size_t cap = (size_t)local->maxImages * mk * 4 * 9 + (size_t)local->maxImages * mk * sizeof(Key) + (size_t)2 * local->maxResiduals * sizeof(FuseAdj) + 256;
        BF_HIP_TRY(BF_MALLOC(&local->fuseScratch, cap));
        local->fuseScratchBytes = cap;
    }
    if (!local->d_fuseError) { BF_HIP_TRY(BF_MALLOC((void**)&local->d_fuseError, sizeof(int))); BF_HIP_TRY(hipMemset(local->d_fuseError, 0, sizeof(int))); }
    bf_sift_image_gpu img;
    { const int rc = bf_siftmgr_create_image(global, &img); if (rc != BF_OK) return rc; }
    FuseArgs a;
    a.glob = local->d_glob; a.globKeys = local->d_globKeys; a.R = R;
    a.T = reinterpret_cast<const m44*>(d_transforms); a.numKeys = local->d_numKeys; a.keys = local->d_keys; a.descs = local->d_descs; a.nI = nI; a.mk = mk;
    a.K = toM44(colorIntrinsics);
    uint8_t* q = reinterpret_cast<uint8_t*>(local->fuseScratch);
    auto take = [&](size_t bytes) { uint8_t* r = q; q += (bytes + 15) & ~(size_t)15; return r; };
    a.cnt = (uint32_t*)take(N * 4); a.start = (uint32_t*)take(N * 4); a.fill = (uint32_t*)take(N * 4); a.label = (uint32_t*)take(N * 4); a.marker = (uint32_t*)take(N * 4);
    a.flag = (uint32_t*)take(N * 4); a.rep = (uint32_t*)take(N * 4); a.outPos = (uint32_t*)take(N * 4);
    a.tmpKeys = (Key*)take(N * sizeof(Key)); a.adj = (FuseAdj*)take((size_t)2 * R * sizeof(FuseAdj));
    a.dstKeys = reinterpret_cast<Key*>(img.d_keyPoints); a.dstDescs = reinterpret_cast<uint8_t*>(img.d_keyPointDescs); a.dstNum = img.d_numKeyPoints; a.dstMaxKeys = global->maxKeys;
    a.error = local->d_fuseError;
    // the chunk's manager and the global manager share the bundling stream in the frame loop; if they do not, order them
    if (global->stream != local->stream) BF_HIP_TRY(hipStreamSynchronize(global->stream));
    k_fuse_to_global<<<1, FUSE_THREADS, 0, local->stream>>>(a);
    BF_HIP_TRY(hipGetLastError());
    if (global->stream != local->stream) BF_HIP_TRY(hipStreamSynchronize(local->stream));
    return bf_siftmgr_finalize_image(global, -1);          // the count was written on the device
}

// conditions of the device search of the LAST fuse: 0 none; 2 the connected-component labelling did not converge within its sweep cap (never observed; a split
// track would silently differ from the host form otherwise)
int bf_siftmgr_fuse_error(bf_siftmgr* local, int* err) {
    BF_REQUIRE(local && err, "null argument");
    *err = 0;
    if (local->d_fuseError) { BF_HIP_TRY(hipMemcpyAsync(err, local->d_fuseError, sizeof(int), hipMemcpyDeviceToHost, local->stream)); BF_HIP_TRY(hipStreamSynchronize(local->stream)); }
    return BF_OK;
}

// The reference's own form: everything to the host, recursive search there, upload (kept for comparison: tests/test_siftmgr_gpu.py).
int bf_siftmgr_fuse_to_global_host(bf_siftmgr* local, bf_siftmgr* global, const float colorIntrinsics[16], const float* d_transforms,
                                   const float colorIntrinsicsInv[16]) {
    (void)colorIntrinsicsInv;
    BF_REQUIRE(local && global && colorIntrinsics && d_transforms, "null argument");
    BF_REQUIRE(local->globNumResiduals > 0, "no correspondences to fuse");
    const uint32_t R = local->globNumResiduals, nI = local->numImages, mk = local->maxKeys;
    std::vector<bf_entry_j> corr(R);
    std::vector<uint2> corrKeys(R);
    std::vector<m44> T(nI);
    std::vector<int> nKeys(nI);
    std::vector<Key> allKeys((size_t)nI * mk);
    std::vector<uint8_t> allDesc((size_t)nI * mk * 128);
    hipStream_t s = local->stream;
    BF_HIP_TRY(hipMemcpyAsync(corr.data(), local->d_glob, sizeof(bf_entry_j) * R, hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipMemcpyAsync(corrKeys.data(), local->d_globKeys, sizeof(uint2) * R, hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipMemcpyAsync(T.data(), d_transforms, sizeof(m44) * nI, hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipMemcpyAsync(nKeys.data(), local->d_numKeys, sizeof(int) * nI, hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipMemcpyAsync(allKeys.data(), local->d_keys, sizeof(Key) * allKeys.size(), hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipMemcpyAsync(allDesc.data(), local->d_descs, allDesc.size(), hipMemcpyDeviceToHost, s));
    BF_HIP_TRY(hipStreamSynchronize(s));

    const float MAX_TRACK_CORR_ERROR = 0.03f;
    const float NINF = -__builtin_huge_valf();
    std::vector<std::vector<TrackEl>> corrPerKey((size_t)nI * mk);
    for (uint32_t i = 0; i < R; ++i) {                        // computeTracks :381-400
        const bf_entry_j& c = corr[i];
        if (c.imgIdx_i == 0xFFFFFFFFu) continue;
        const uint2 k = corrKeys[i];
        const f3 pi = mk3(c.pos_i[0], c.pos_i[1], c.pos_i[2]), pj = mk3(c.pos_j[0], c.pos_j[1], c.pos_j[2]);
        const f3 d = xform(T[c.imgIdx_i], pi) - xform(T[c.imgIdx_j], pj);
        const float err = sqrtf(dot3(d, d));
        if (err < MAX_TRACK_CORR_ERROR) {
            corrPerKey[k.x].push_back(TrackEl(make_uint2(c.imgIdx_j, k.y), pj));
            corrPerKey[k.y].push_back(TrackEl(make_uint2(c.imgIdx_i, k.x), pi));
        } else {
            corrPerKey[k.x].push_back(TrackEl(make_uint2(c.imgIdx_j, k.y), mk3(NINF, NINF, NINF)));
            corrPerKey[k.y].push_back(TrackEl(make_uint2(c.imgIdx_i, k.x), mk3(NINF, NINF, NINF)));
        }
    }
    std::vector<std::vector<TrackEl>> tracks;
    std::vector<char> marker((size_t)nI * mk, 0);
    for (uint32_t i = 0; i < nI; ++i)
        for (int k = 0; k < std::min(std::max(nKeys[i], 0), (int)mk); ++k) {
            if (tracks.empty() || !tracks.back().empty()) tracks.push_back(std::vector<TrackEl>());
            findTrack(corrPerKey, marker, tracks.back(), i * mk + (uint32_t)k);
        }
    const m44 K = toM44(colorIntrinsics);
    std::vector<Key> curKeys;
    std::vector<uint8_t> curDesc;
    for (const auto& tr : tracks) {
        if (tr.empty()) continue;
        const TrackEl& rep = tr.front();
        f3 pos = mk3(0, 0, 0);
        unsigned num = 0;
        for (const auto& el : tr)
            if (el.second.x != NINF) { pos = pos + xform(T[el.first.x], el.second); num++; }
        if (num > 0) {
            pos = pos / (float)num;
            pos = xform(K, pos);
            Key key;
            key.x = pos.x / pos.z; key.y = pos.y / pos.z;
            key.scale = allKeys[rep.first.y].scale;
            key.depth = pos.z;
            curKeys.push_back(key);
            curDesc.insert(curDesc.end(), allDesc.begin() + (size_t)rep.first.y * 128, allDesc.begin() + (size_t)rep.first.y * 128 + 128);
        }
    }
    const uint32_t numKeys = std::min<uint32_t>((uint32_t)curKeys.size(), global->maxKeys);
    if (curKeys.size() > global->maxKeys)      // like the reference, only the keys (not the descriptors) are reordered here
        std::sort(curKeys.begin(), curKeys.end(), [](const Key& l, const Key& r) { return l.depth < r.depth; });
    bf_sift_image_gpu img;
    int rc = bf_siftmgr_create_image(global, &img);
    if (rc) return rc;
    if (numKeys) {
        BF_HIP_TRY(hipMemcpyAsync(img.d_keyPoints, curKeys.data(), sizeof(Key) * numKeys, hipMemcpyHostToDevice, global->stream));
        BF_HIP_TRY(hipMemcpyAsync(img.d_keyPointDescs, curDesc.data(), (size_t)128 * numKeys, hipMemcpyHostToDevice, global->stream));
    }
    rc = bf_siftmgr_finalize_image(global, (int32_t)numKeys);
    if (rc) return rc;
    BF_HIP_TRY(hipStreamSynchronize(global->stream));
    return BF_OK;
}

}  // extern "C"
