// oracle/ref/shim: the two names of mLib's core-base/common.h (un-vendored submodule) that the device-side headers of the
// reference use — test infrastructure only
#ifndef BF_REF_SHIM_MLIB_COMMON_H
#define BF_REF_SHIM_MLIB_COMMON_H
#include <stdexcept>
#include <string>
typedef unsigned char uchar;
typedef unsigned int UINT;
typedef unsigned char BYTE;
#define MLIB_EXCEPTION(s) std::runtime_error(std::string(s))
#define MLIB_ASSERT(b) do { if (!(b)) throw std::runtime_error("MLIB_ASSERT " #b); } while (0)
#define MLIB_ASSERT_STR(b, s) do { if (!(b)) throw std::runtime_error(std::string(s)); } while (0)
#define MLIB_WARNING(s) ((void)0)
#define SAFE_DELETE(p) do { delete (p); (p) = nullptr; } while (0)
#define SAFE_DELETE_ARRAY(p) do { delete[] (p); (p) = nullptr; } while (0)
// ml::DepthImage32 as SIFTImageManager::fuseLocalKeyDepths uses it (allocate / getData / getNumPixels / operator()(x, y))
#include <fstream>
#include <vector>
struct DepthImage32 {
    std::vector<float> d; unsigned int w = 0, h = 0;
    DepthImage32() {}
    DepthImage32(unsigned int width, unsigned int height) { allocate(width, height); }
    const float* getData() const { return d.data(); }
    void allocate(unsigned int width, unsigned int height) { w = width; h = height; d.assign((size_t)w * h, 0.0f); }
    float* getData() { return d.data(); }
    size_t getNumPixels() const { return d.size(); }
    float& operator()(unsigned int x, unsigned int y) { return d[(size_t)y * w + x]; }
    const float& operator()(unsigned int x, unsigned int y) const { return d[(size_t)y * w + x]; }
    unsigned int getWidth() const { return w; }
    unsigned int getHeight() const { return h; }
    void setPixels(float v) { for (float& e : d) e = v; }
};
#endif
