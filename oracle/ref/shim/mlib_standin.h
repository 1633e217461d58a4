// oracle/ref/shim/mlib_standin.h — TEST INFRASTRUCTURE ONLY.
//
// mLib (github.com/niessner/mLib, an un-vendored git submodule of the reference: external/mLib is an empty directory in
// /root/reference) is the reference's host-side vector / matrix library.  The host files of the path that are compiled AS THEY ARE
// into oracle/_ref (TrajectoryManager.cpp with PoseHelper.h, ...) use a small part of it.  This header supplies exactly that part with
// mLib's documented conventions: matrices are row-major (`m[i]` is the i-th float of the 16, `m(r, c)` row r column c), a mat4f times
// a vec3f is the affine transform of the point, `a | b` is the dot product and `a ^ b` the cross product.  Only plain element-wise
// arithmetic lives here - every algorithm under test (the se(3) log / exp, the list policy) is the reference's own code.
#ifndef BF_REF_SHIM_MLIB_STANDIN_H
#define BF_REF_SHIM_MLIB_STANDIN_H
#include <math.h>          // the C++ wrapper: float overloads of sin / asin / sqrt ... in the global namespace, as with MSVC's <cmath>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <limits>
#include <list>
#include <mutex>
#include <string>
#include <vector>
#include "core-base/common.h"

namespace ml {

namespace math {
static const float PIf = 3.14159265358979323846f;
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
template <class T> inline T clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int round(float f) { return f > 0.0f ? (int)floorf(f + 0.5f) : (int)ceilf(f - 0.5f); }
inline float radiansToDegrees(float r) { return r * (180.0f / PIf); }
inline float degreesToRadians(float d) { return d * (PIf / 180.0f); }
}  // namespace math

template <class T> struct point2d {
    union { struct { T x, y; }; T array[2]; };
    point2d() : x(0), y(0) {}
    point2d(T a, T b) : x(a), y(b) {}
    T& operator[](unsigned int i) { return array[i]; }
    const T& operator[](unsigned int i) const { return array[i]; }
    bool operator==(const point2d& o) const { return x == o.x && y == o.y; }
    bool operator!=(const point2d& o) const { return !(*this == o); }
};
typedef point2d<unsigned int> vec2ui;
typedef point2d<int> vec2i;
typedef point2d<float> vec2f;
typedef unsigned long long UINT64;
namespace math { inline int round(float f); inline vec2i round(const vec2f& v); }
template <class T> inline std::ostream& operator<<(std::ostream& s, const point2d<T>& v) { return s << v.x << " " << v.y; }

struct vec3f {
    union { struct { float x, y, z; }; float array[3]; };
    vec3f() : x(0), y(0), z(0) {}
    explicit vec3f(float v) : x(v), y(v), z(v) {}
    explicit vec3f(const struct vec3i& v);
    vec3f(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](unsigned int i) { return array[i]; }
    const float& operator[](unsigned int i) const { return array[i]; }
    vec3f operator+(const vec3f& o) const { return vec3f(x + o.x, y + o.y, z + o.z); }
    vec3f operator-(const vec3f& o) const { return vec3f(x - o.x, y - o.y, z - o.z); }
    vec3f operator-() const { return vec3f(-x, -y, -z); }
    vec3f operator*(float s) const { return vec3f(x * s, y * s, z * s); }
    vec3f operator/(float s) const { return vec3f(x / s, y / s, z / s); }
    vec3f operator-(float s) const { return vec3f(x - s, y - s, z - s); }
    vec3f& operator+=(const vec3f& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3f& operator-=(const vec3f& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3f& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    vec3f& operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
    float operator|(const vec3f& o) const { return x * o.x + y * o.y + z * o.z; }
    vec3f operator^(const vec3f& o) const { return vec3f(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
    float lengthSq() const { return x * x + y * y + z * z; }
    float length() const { return sqrt(lengthSq()); }
    static float distSq(const vec3f& a, const vec3f& b) { return (a - b).lengthSq(); }
    static float dist(const vec3f& a, const vec3f& b) { return (a - b).length(); }
};
inline vec3f operator*(float s, const vec3f& v) { return v * s; }

struct vec3i {
    int x = 0, y = 0, z = 0; vec3i() {} vec3i(int a, int b, int c) : x(a), y(b), z(c) {}
    vec3i operator*(int s) const { return vec3i(x * s, y * s, z * s); }
    vec3i operator+(const vec3i& o) const { return vec3i(x + o.x, y + o.y, z + o.z); }
};
inline vec3f::vec3f(const vec3i& v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
struct vec4i { int x = 0, y = 0, z = 0, w = 0; vec4i() {} vec4i(int a, int b, int c, int d) : x(a), y(b), z(c), w(d) {} };
struct vec4f {
    union { struct { float x, y, z, w; }; float array[4]; };
    vec4f() : x(0), y(0), z(0), w(0) {}
    explicit vec4f(float v) : x(v), y(v), z(v), w(v) {}
    vec4f(const vec3f& v, float ww) : x(v.x), y(v.y), z(v.z), w(ww) {}
    vec3f getVec3() const { return vec3f(x, y, z); }
    vec4f operator-(const vec4f& o) const { return vec4f(x - o.x, y - o.y, z - o.z, w - o.w); }
    float operator|(const vec4f& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    float lengthSq() const { return x * x + y * y + z * z + w * w; }
    float length() const { return sqrt(lengthSq()); }
    vec4f(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    float& operator[](unsigned int i) { return array[i]; }
    const float& operator[](unsigned int i) const { return array[i]; }
};

struct vec6f {
    float array[6];
    vec6f() { for (float& v : array) v = 0.0f; }
    vec6f(float a, float b, float c, float d, float e, float f) { array[0] = a; array[1] = b; array[2] = c; array[3] = d; array[4] = e; array[5] = f; }
    float& operator[](unsigned int i) { return array[i]; }
    const float& operator[](unsigned int i) const { return array[i]; }
    vec6f operator-(const vec6f& o) const { vec6f r; for (int i = 0; i < 6; ++i) r.array[i] = array[i] - o.array[i]; return r; }
    vec6f operator+(const vec6f& o) const { vec6f r; for (int i = 0; i < 6; ++i) r.array[i] = array[i] + o.array[i]; return r; }
    float operator|(const vec6f& o) const { float s = 0.0f; for (int i = 0; i < 6; ++i) s += array[i] * o.array[i]; return s; }
    vec3f getVec3() const { return vec3f(array[0], array[1], array[2]); }
};

struct mat3f {
    float matrix[9];
    mat3f() { for (float& v : matrix) v = 0.0f; }
    float& operator()(unsigned int r, unsigned int c) { return matrix[r * 3 + c]; }
    const float& operator()(unsigned int r, unsigned int c) const { return matrix[r * 3 + c]; }
    float& operator[](unsigned int i) { return matrix[i]; }
    const float& operator[](unsigned int i) const { return matrix[i]; }
    float trace() const { return matrix[0] + matrix[4] + matrix[8]; }
    vec3f operator*(const vec3f& v) const {
        return vec3f(matrix[0] * v.x + matrix[1] * v.y + matrix[2] * v.z, matrix[3] * v.x + matrix[4] * v.y + matrix[5] * v.z,
                     matrix[6] * v.x + matrix[7] * v.y + matrix[8] * v.z);
    }
    mat3f operator*(const mat3f& o) const {
        mat3f r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j);
        return r;
    }
    static mat3f identity() { mat3f r; r.matrix[0] = r.matrix[4] = r.matrix[8] = 1.0f; return r; }
    static mat3f rotationX(float deg) { float a = math::degreesToRadians(deg), c = cos(a), s = sin(a); mat3f r = identity(); r(1, 1) = c; r(1, 2) = -s; r(2, 1) = s; r(2, 2) = c; return r; }
    static mat3f rotationY(float deg) { float a = math::degreesToRadians(deg), c = cos(a), s = sin(a); mat3f r = identity(); r(0, 0) = c; r(0, 2) = s; r(2, 0) = -s; r(2, 2) = c; return r; }
    static mat3f rotationZ(float deg) { float a = math::degreesToRadians(deg), c = cos(a), s = sin(a); mat3f r = identity(); r(0, 0) = c; r(0, 1) = -s; r(1, 0) = s; r(1, 1) = c; return r; }
};

struct mat4f {
    union {
        float matrix[16];
        struct { float _m00, _m01, _m02, _m03, _m10, _m11, _m12, _m13, _m20, _m21, _m22, _m23, _m30, _m31, _m32, _m33; };
    };
    mat4f() { for (float& v : matrix) v = 0.0f; }
    explicit mat4f(const float* p) { memcpy(matrix, p, sizeof matrix); }
    mat4f getTranspose() const { mat4f r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.matrix[i * 4 + j] = matrix[j * 4 + i]; return r; }
    float& operator()(unsigned int r, unsigned int c) { return matrix[r * 4 + c]; }
    const float& operator()(unsigned int r, unsigned int c) const { return matrix[r * 4 + c]; }
    float& operator[](unsigned int i) { return matrix[i]; }
    const float& operator[](unsigned int i) const { return matrix[i]; }
    float* getData() { return matrix; }
    const float* getData() const { return matrix; }
    void setZero(float v = 0.0f) { for (float& e : matrix) e = v; }
    void setIdentity() { setZero(); matrix[0] = matrix[5] = matrix[10] = matrix[15] = 1.0f; }
    static mat4f zero(float v = 0.0f) { mat4f r; r.setZero(v); return r; }
    static mat4f identity() { mat4f r; r.setIdentity(); return r; }
    mat3f getRotation() const { mat3f r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(i, j); return r; }
    vec3f getTranslation() const { return vec3f(matrix[3], matrix[7], matrix[11]); }
    void setRotationMatrix(const mat3f& r) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) (*this)(i, j) = r(i, j); }
    void setRotation(const mat3f& r) { setRotationMatrix(r); }
    void setTranslationVector(const vec3f& t) { matrix[3] = t.x; matrix[7] = t.y; matrix[11] = t.z; }
    void setTranslation(const vec3f& t) { setTranslationVector(t); }
    mat4f operator*(const mat4f& o) const {
        mat4f r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
            r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j) + (*this)(i, 3) * o(3, j);
        return r;
    }
    vec4f operator*(const vec4f& v) const {
        return vec4f(matrix[0] * v.x + matrix[1] * v.y + matrix[2] * v.z + matrix[3] * v.w, matrix[4] * v.x + matrix[5] * v.y + matrix[6] * v.z + matrix[7] * v.w,
                     matrix[8] * v.x + matrix[9] * v.y + matrix[10] * v.z + matrix[11] * v.w, matrix[12] * v.x + matrix[13] * v.y + matrix[14] * v.z + matrix[15] * v.w);
    }
    vec3f operator*(const vec3f& p) const {      // affine transform of a point
        return vec3f(matrix[0] * p.x + matrix[1] * p.y + matrix[2] * p.z + matrix[3], matrix[4] * p.x + matrix[5] * p.y + matrix[6] * p.z + matrix[7],
                     matrix[8] * p.x + matrix[9] * p.y + matrix[10] * p.z + matrix[11]);
    }
    // cofactor expansion (stand-in arithmetic; no pinned test depends on the rounding of an inverse computed here)
    mat4f getInverse() const {
        const float* m = matrix; float inv[16];
        inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
        inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
        inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
        inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
        inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
        inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
        inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
        inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
        inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
        inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
        inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
        inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
        inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
        inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
        inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
        inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
        const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
        mat4f r; const float id = 1.0f / det;
        for (int i = 0; i < 16; ++i) r.matrix[i] = inv[i] * id;
        return r;
    }
};

// host utilities the bundling classes name: a wall-clock timer, directory helpers, the binary stream of debug dumps (inline methods
// that no pinned test calls: declared so that the headers compile), the parameter-file reader of the application singletons
struct Timer {
    double t0 = 0.0, t1 = 0.0;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    void start() { t0 = now(); }
    void stop() { t1 = now(); }
    double getElapsedTime() const { return t1 - t0; }
    double getElapsedTimeMS() const { return 1e3 * (t1 - t0); }
};
namespace util {
inline bool directoryExists(const std::string&) { return true; }
inline void makeDirectory(const std::string&) {}
inline std::string directoryFromPath(const std::string& p) { const size_t k = p.find_last_of("/\\"); return k == std::string::npos ? std::string() : p.substr(0, k + 1); }
}  // namespace util
// images of the debug writers (CUDACache::saveToFile / printCacheImages, ...): shape only, never executed by a pinned test
template <class T> struct BaseImage {
    struct Pixel { unsigned int x, y; T value; };
    std::vector<T> d; unsigned int w = 0, h = 0;
    BaseImage() {}
    BaseImage(unsigned int width, unsigned int height) : d((size_t)width * height), w(width), h(height) {}
    template <class I> explicit BaseImage(const I&) {}
    T* getData() { return d.data(); }
    const T* getData() const { return d.data(); }
    size_t getNumPixels() const { return d.size(); }
    unsigned int getWidth() const { return w; }
    unsigned int getHeight() const { return h; }
    T& operator()(unsigned int x, unsigned int y) { return d[(size_t)y * w + x]; }
    const T& operator()(unsigned int x, unsigned int y) const { return d[(size_t)y * w + x]; }
    void allocate(unsigned int width, unsigned int height) { w = width; h = height; d.assign((size_t)w * h, T()); }
    void setPixels(const T& v) { for (T& e : d) e = v; }
    void setInvalidValue(const T&) {}
    Pixel* begin() { return nullptr; }
    Pixel* end() { return nullptr; }
};
struct vec4uc { unsigned char x = 0, y = 0, z = 0, w = 0; };
typedef BaseImage<float> ColorImageR32;
typedef BaseImage<vec3f> ColorImageR32G32B32;
typedef BaseImage<vec3f> PointImage;
typedef BaseImage<vec4f> ColorImageR32G32B32A32;
typedef BaseImage<vec4uc> ColorImageR8G8B8A8;
struct PointCloudf { std::vector<vec3f> m_points, m_normals; std::vector<vec4f> m_colors; };        // debug dumps (shape only)
struct PointCloudIOf { static void saveToFile(const std::string&, const PointCloudf&) { throw std::runtime_error("mlib_standin: PointCloudIOf is not provided"); } };
template <class T> struct BoundingBox3 {
    void include(const vec3i&) {}
    T getExtentX() const { return 0; } T getExtentY() const { return 0; } T getExtentZ() const { return 0; }
    vec3i getMin() const { return vec3i(); } vec3i getMax() const { return vec3i(); }
};
template <class T> inline std::ostream& operator<<(std::ostream& s, const BoundingBox3<T>&) { return s; }
struct FreeImageWrapper { template <class I> static void saveImage(const std::string&, const I&) { throw std::runtime_error("mlib_standin: FreeImageWrapper is not provided"); } };
struct BinaryDataStreamFile {
    BinaryDataStreamFile(const std::string&, bool) { throw std::runtime_error("mlib_standin: BinaryDataStreamFile is not provided"); }
    template <class T> BinaryDataStreamFile& operator<<(const T&) { return *this; }
    template <class T> BinaryDataStreamFile& operator>>(T&) { return *this; }
    void writeData(const BYTE*, size_t) {}
    void readData(BYTE*, size_t) {}
    void close() {}
};
struct ParameterFile {
    template <class T> bool readParameter(const std::string&, T&) const { return false; }
};

// the two mLib classes PoseHelper.h names in helpers that no pinned test calls (ATE evaluation, pose files): declared so that the
// header compiles, loud if ever reached
struct quatf {
    explicit quatf(const mat3f&) { throw std::runtime_error("mlib_standin: quatf is not provided"); }
    vec3f imag() const { return vec3f(); }
    float real() const { return 0.0f; }
};
struct EigenWrapperf {
    static mat4f kabsch(const std::vector<vec3f>&, const std::vector<vec3f>&, vec3f&) { throw std::runtime_error("mlib_standin: EigenWrapperf::kabsch is not provided"); }
};

inline std::ostream& operator<<(std::ostream& s, const vec3f& v) { return s << v.x << " " << v.y << " " << v.z; }
inline std::ostream& operator<<(std::ostream& s, const vec4f& v) { return s << v.x << " " << v.y << " " << v.z << " " << v.w; }
inline std::ostream& operator<<(std::ostream& s, const vec3i& v) { return s << v.x << " " << v.y << " " << v.z; }
inline std::ostream& operator<<(std::ostream& s, const mat4f& m) { for (int i = 0; i < 16; ++i) s << m.matrix[i] << (i % 4 == 3 ? "\n" : " "); return s; }
namespace math { inline vec2i round(const vec2f& v) { return vec2i(round(v.x), round(v.y)); } }
}  // namespace ml
using namespace ml;        // as the reference's mLib.h does
#endif
