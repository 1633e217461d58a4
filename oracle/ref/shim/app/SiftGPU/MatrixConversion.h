// oracle/ref/shim/app/SiftGPU/MatrixConversion.h — TEST INFRASTRUCTURE ONLY.  The two conversions of SiftGPU/MatrixConversion.h (:8-10,
// :27-29) between the CUDA-side float4x4 and mLib's mat4f (both row-major, 16 floats); the real header also converts DirectX types.
#pragma once
namespace MatrixConversion {
static ml::mat4f toMlib(const float4x4& m) { return ml::mat4f(m.ptr()); }
static float4x4 toCUDA(const ml::mat4f& m) { return float4x4(m.getData()); }
}
