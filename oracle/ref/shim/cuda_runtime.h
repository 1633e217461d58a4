// oracle/ref/shim/cuda_runtime.h — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A host-side stand-in for the CUDA headers, so that g++ can compile the reference's own device code
// (/root/reference/FriedLiver/Source/..., read in place, never copied) into oracle/_ref/libbfref.so.  It provides
// the vector types and their make_* constructors, the execution-space qualifiers as empty macros, the device
// builtins the reference calls (type punning, min/max, serial atomics, point-sampled 2-D textures), the handful of
// runtime calls its launch wrappers make (mapped to malloc / memcpy), and threadIdx / blockIdx / __syncthreads() backed by
// the serial block emulator of ../emu.h.  __CUDACC__ is defined so that the reference's `#ifdef __CUDACC__` device
// sections (VoxelUtilHashSDF.h:217-828) are compiled.  Nothing here restates reference code.
#ifndef BF_REF_SHIM_CUDA_RUNTIME_H
#define BF_REF_SHIM_CUDA_RUNTIME_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <iostream>
#include <list>
#include <mutex>
#include <string>
#include <vector>

#ifndef __CUDACC__
#define __CUDACC__ 1
#endif
#ifndef __NVCC__
#define __NVCC__ 1      // cuda_svd3.h:34-42 would otherwise take its SSE rsqrt approximation instead of the device rsqrt
#endif
#define __host__
#define __device__
#define __global__
#define __constant__
#define __shared__ static
#define __inline__ inline
#define __forceinline__ inline
#define __align__(n) __declspec(align(n))      // clang -fdeclspec: applies to the type even when written before `struct`, like nvcc / MSVC
#define __launch_bounds__(...)

// ---- vector types (layout and alignment of vector_types.h) ----
#define BF_REF_VEC(T, N)                                                                          \
    struct T##1 { N x; };                                                                           \
    struct T##3 { N x, y, z; };                                                                     \
    static inline T##1 make_##T##1(N x) { T##1 r; r.x = x; return r; }                              \
    static inline T##3 make_##T##3(N x, N y, N z) { T##3 r; r.x = x; r.y = y; r.z = z; return r; }
#define BF_REF_VEC24(T, N, A2, A4)                                                                \
    struct __attribute__((aligned(A2))) T##2 { N x, y; };                                           \
    struct __attribute__((aligned(A4))) T##4 { N x, y, z, w; };                                     \
    static inline T##2 make_##T##2(N x, N y) { T##2 r; r.x = x; r.y = y; return r; }                \
    static inline T##4 make_##T##4(N x, N y, N z, N w) { T##4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
BF_REF_VEC(char, signed char) BF_REF_VEC24(char, signed char, 2, 4)
BF_REF_VEC(uchar, unsigned char) BF_REF_VEC24(uchar, unsigned char, 2, 4)
BF_REF_VEC(short, short) BF_REF_VEC24(short, short, 4, 8)
BF_REF_VEC(ushort, unsigned short) BF_REF_VEC24(ushort, unsigned short, 4, 8)
BF_REF_VEC(int, int) BF_REF_VEC24(int, int, 8, 16)
BF_REF_VEC(uint, unsigned int) BF_REF_VEC24(uint, unsigned int, 8, 16)
BF_REF_VEC(float, float) BF_REF_VEC24(float, float, 8, 16)
BF_REF_VEC(double, double) BF_REF_VEC24(double, double, 16, 16)
#undef BF_REF_VEC
#undef BF_REF_VEC24

// cutil_math.h declares unary minus on NON-const lvalue references (accepted for temporaries by MSVC / nvcc); temporaries bind here
static inline float2 operator-(const float2& a) { return make_float2(-a.x, -a.y); }
static inline float3 operator-(const float3& a) { return make_float3(-a.x, -a.y, -a.z); }
static inline float4 operator-(const float4& a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
struct cudaArray { const void* ptr; size_t width, height, pitch; };      // a 2-D array is a pitched host image here

struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
    dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
};

// ---- execution configuration of the block emulator (../emu.h) ----
extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
static const int warpSize = 32;
void __syncthreads();
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- device builtins ----
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned int __float_as_uint(float f) { unsigned int i; memcpy(&i, &f, 4); return i; }
static inline float __saturatef(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : (x != x ? 0.0f : x)); }
// CUDA's rsqrtf is a <= 2 ulp hardware approximation whose bits are not specified; oracle, product and this stand-in all use the
// correctly rounded 1 / sqrt
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
// glibc declares __expf / __logf / __sinf / __cosf / __powf / __sincosf itself (extern): the fast-math intrinsics resolve to libm
#define __sincosf(x, s, c) sincosf((x), (s), (c))
static inline int __float2int_rn(float x) { return (int)nearbyintf(x); }
static inline int __float2int_rd(float x) { return (int)floorf(x); }
static inline int __float2int_rz(float x) { return (int)x; }
static inline float __int2float_rn(int x) { return (float)x; }
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned int __umul24(unsigned int a, unsigned int b) { return a * b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
static inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// serial atomics: blocks and threads of an emulated launch never run concurrently
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; *p = v < o ? v : o; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; *p = v > o ? v : o; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline float atomicAdd(float* p, int v) { float o = *p; *p = o + (float)v; return o; }
static inline int atomicAdd(int* p, unsigned int v) { int o = *p; *p = o + (int)v; return o; }
static inline unsigned int atomicAdd(unsigned int* p, int v) { unsigned int o = *p; *p = o + (unsigned int)v; return o; }
static inline unsigned int atomicSub(unsigned int* p, int v) { unsigned int o = *p; *p = o - (unsigned int)v; return o; }
static inline int atomicExch(int* p, unsigned int v) { int o = *p; *p = (int)v; return o; }
static inline unsigned int atomicExch(unsigned int* p, int v) { unsigned int o = *p; *p = (unsigned int)v; return o; }

// warp shuffles: lock-step exchange between the fibers of a warp (../emu.cpp)
float __shfl_down(float, int, int = 32);
float __shfl_xor(float, int, int = 32);
float __shfl(float, int, int = 32);
int __shfl_down(int, int, int = 32);
int __shfl_xor(int, int, int = 32);

// ---- runtime calls made by the reference's launch wrappers ----
typedef int cudaError_t;
typedef void* cudaStream_t;
typedef void* cudaEvent_t;
enum { cudaEventBlockingSync = 1, cudaEventDefault = 0 };
static inline cudaError_t cudaEventCreate(cudaEvent_t*) { return 0; }
static inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t*, int) { return 0; }
static inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return 0; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t, cudaEvent_t) { *ms = 0.0f; return 0; }
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)calloc(n ? n : 1, 1); return cudaSuccess; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "emulated"; }
static inline cudaError_t shimCopyToSymbol(void* sym, const void* src, size_t n, size_t offset = 0, int /*kind*/ = 0) { memcpy((char*)sym + offset, src, n); return cudaSuccess; }
#define cudaMemcpyToSymbol(sym, ...) shimCopyToSymbol((void*)&(sym), __VA_ARGS__)
#define cudaMemcpyFromSymbol(dst, sym, n, ...) (memcpy((dst), (const void*)&(sym), (n)), cudaSuccess)
#define cutilSafeCall(x) (void)(x)
#define cutilCheckMsg(msg) ((void)0)
#define CUDA_SAFE_CALL(x) (void)(x)
#define CUDA_CHECKED_CALL(x) (void)(x)

// ---- texture references: 1-D fetches from linear memory and point-sampled 2-D textures with clamped addressing (the modes the reference binds) ----
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
enum cudaTextureFilterMode { cudaFilterModePoint, cudaFilterModeLinear };
enum cudaTextureAddressMode { cudaAddressModeWrap, cudaAddressModeClamp };
enum { cudaTextureType1D = 1, cudaTextureType2D = 2 };
enum cudaChannelFormatKind { cudaChannelFormatKindSigned, cudaChannelFormatKindUnsigned, cudaChannelFormatKindFloat };
struct cudaChannelFormatDesc { int x, y, z, w, f; };
template <class T> static inline cudaChannelFormatDesc cudaCreateChannelDesc() { cudaChannelFormatDesc d = {(int)sizeof(T) * 8, 0, 0, 0, 0}; return d; }
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f) { cudaChannelFormatDesc d = {x, y, z, w, (int)f}; return d; }
struct textureReference {
    const void* ptr = nullptr; size_t width = 0, height = 0, pitch = 0;
    cudaTextureFilterMode filterMode = cudaFilterModePoint; cudaTextureAddressMode addressMode[3]; int normalized = 0;
    cudaChannelFormatDesc channelDesc;
};
template <class T, int DIM = 1, cudaTextureReadMode M = cudaReadModeElementType>
struct texture : textureReference {};
static inline cudaError_t cudaBindTexture(size_t* off, const textureReference* t, const void* p, const cudaChannelFormatDesc* = nullptr, size_t n = (size_t)-1) {
    if (off) *off = 0;
    textureReference* w = const_cast<textureReference*>(t); w->ptr = p; w->width = n; w->height = 1; w->pitch = n; return cudaSuccess;
}
static inline cudaError_t cudaBindTexture(size_t* off, const textureReference& t, const void* p, const cudaChannelFormatDesc& d, size_t n = (size_t)-1) { return cudaBindTexture(off, &t, p, &d, n); }
template <class T, int DIM, cudaTextureReadMode M>
static inline cudaError_t cudaBindTexture(size_t* off, const texture<T, DIM, M>& t, const void* p, size_t n = (size_t)-1) { return cudaBindTexture(off, (const textureReference*)&t, p, nullptr, n); }
static inline cudaError_t cudaBindTexture2D(size_t* off, const textureReference* t, const void* p, const cudaChannelFormatDesc*, size_t w, size_t h, size_t pitch) {
    if (off) *off = 0;
    textureReference* r = const_cast<textureReference*>(t); r->ptr = p; r->width = w; r->height = h; r->pitch = pitch; return cudaSuccess;
}
template <class T, int DIM, cudaTextureReadMode M>
static inline cudaError_t cudaBindTexture2D(size_t* off, texture<T, DIM, M>& t, const void* p, const cudaChannelFormatDesc& d, size_t w, size_t h, size_t pitch) {
    return cudaBindTexture2D(off, (const textureReference*)&t, p, &d, w, h, pitch);
}
static inline cudaError_t cudaBindTextureToArray(const textureReference* t, const cudaArray* a, const cudaChannelFormatDesc* = nullptr) {
    textureReference* r = const_cast<textureReference*>(t); r->ptr = a->ptr; r->width = a->width; r->height = a->height; r->pitch = a->pitch; return cudaSuccess;
}
template <class T, int DIM, cudaTextureReadMode M>
static inline cudaError_t cudaBindTextureToArray(texture<T, DIM, M>& t, const cudaArray* a, const cudaChannelFormatDesc&) { return cudaBindTextureToArray((const textureReference*)&t, a); }
static inline cudaError_t cudaUnbindTexture(const textureReference*) { return cudaSuccess; }
static inline cudaError_t cudaUnbindTexture(const textureReference&) { return cudaSuccess; }
template <class T, int DIM>
static inline T tex2D(const texture<T, DIM, cudaReadModeElementType>& t, float x, float y) {      // unnormalised coordinates, point filter: texel floor(x), floor(y)
    long ix = (long)floorf(x), iy = (long)floorf(y);
    ix = ix < 0 ? 0 : (ix >= (long)t.width ? (long)t.width - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= (long)t.height ? (long)t.height - 1 : iy);
    return *(const T*)((const char*)t.ptr + (size_t)iy * t.pitch + (size_t)ix * sizeof(T));
}
template <class T, int DIM>
static inline T tex1Dfetch(const texture<T, DIM, cudaReadModeElementType>& t, int i) { return ((const T*)t.ptr)[i]; }
template <int DIM>
static inline float tex1Dfetch(const texture<unsigned char, DIM, cudaReadModeNormalizedFloat>& t, int i) { return (float)((const unsigned char*)t.ptr)[i] / 255.0f; }
// 2-D arrays are pitched host images
static inline cudaError_t cudaMallocArray(cudaArray** a, const cudaChannelFormatDesc* d, size_t w, size_t h = 0) {
    const size_t texel = (size_t)(d->x + d->y + d->z + d->w) / 8, hh = h ? h : 1;
    cudaArray* r = new cudaArray; r->width = w; r->height = hh; r->pitch = w * texel; r->ptr = calloc(hh, r->pitch); *a = r; return cudaSuccess;
}
static inline cudaError_t cudaFreeArray(cudaArray* a) { if (a) { free(const_cast<void*>(a->ptr)); delete a; } return cudaSuccess; }
static inline cudaError_t cudaMemcpy2DToArray(cudaArray* dst, size_t wOff, size_t hOff, const void* src, size_t spitch, size_t widthBytes, size_t height, cudaMemcpyKind) {
    for (size_t y = 0; y < height; ++y) memcpy((char*)const_cast<void*>(dst->ptr) + (hOff + y) * dst->pitch + wOff, (const char*)src + y * spitch, widthBytes);
    return cudaSuccess;
}
static inline cudaError_t cudaMemcpyFromArray(void* dst, const cudaArray* src, size_t wOff, size_t hOff, size_t count, cudaMemcpyKind) {
    memcpy(dst, (const char*)src->ptr + hOff * src->pitch + wOff, count); return cudaSuccess;
}
static inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t = nullptr) { memcpy(d, s, n); return cudaSuccess; }
static inline float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }      // __saturatef
struct cudaDeviceProp { char name[256]; int major, minor, multiProcessorCount, clockRate; size_t totalGlobalMem, sharedMemPerBlock; int maxThreadsPerBlock, warpSize, regsPerBlock; };
static inline cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
static inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "host emulation"); p->major = 5; p->warpSize = 32; p->maxThreadsPerBlock = 1024; return cudaSuccess; }
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
static inline cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }

#endif
