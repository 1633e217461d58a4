// oracle/ref/shim/app/SiftGPU/MatrixConversion.h — TEST INFRASTRUCTURE ONLY.  The conversions of SiftGPU/MatrixConversion.h the compiled files use: float4x4 <-> mLib's mat4f (:8-10, :27-29; both row-major,
// 16 floats), vec3f / vec3i -> float3 / int3; the real header also converts DirectX types.
#pragma once
namespace MatrixConversion {
static ml::mat4f toMlib(const float4x4& m) { return ml::mat4f(m.ptr()); }
static float4x4 toCUDA(const ml::mat4f& m) { return float4x4(m.getData()); }
static float3 toCUDA(const ml::vec3f& v) { return make_float3(v.x, v.y, v.z); }      // :49-57
static int3 toCUDA(const ml::vec3i& v) { return make_int3(v.x, v.y, v.z); }
}
