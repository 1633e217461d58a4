// oracle/ref/shim/app/SiftGPU/MatrixConversion.h — TEST INFRASTRUCTURE ONLY.  Stand-in for the conversions of the reference header of this
// name that the compiled files call: the CUDA-side float4x4 <-> mLib's mat4f (both 16 row-major floats) and vec3f / vec3i -> float3 / int3.
// The real header also converts DirectX types and needs <d3dx9math.h>.
#pragma once
#include <cstring>
namespace MatrixConversion {
inline ml::mat4f toMlib(const float4x4& in) {
    ml::mat4f out;
    std::memcpy(out.matrix, in.ptr(), sizeof out.matrix);
    return out;
}
inline float4x4 toCUDA(const ml::mat4f& in) {
    float4x4 out;
    std::memcpy(out.entries, in.matrix, sizeof in.matrix);
    return out;
}
inline float3 toCUDA(const ml::vec3f& in) { float3 out; out.x = in.x; out.y = in.y; out.z = in.z; return out; }
inline int3 toCUDA(const ml::vec3i& in) { int3 out; out.x = in.x; out.y = in.y; out.z = in.z; return out; }
}
