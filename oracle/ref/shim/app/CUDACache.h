// oracle/ref/shim/app/CUDACache.h — TEST INFRASTRUCTURE ONLY.  What the solver reads of a CUDACache (CUDACache.h:44-52: the frame array,
// its size, the intrinsics of the down-sampled frames).  The real class drags mLib's image / stream classes through its inline debug
// writers; its storeFrame kernels are pinned on their own (ref_image.cpp).  Frames are supplied by the glue.
#pragma once
#include "CUDACacheUtil.h"
#include <vector>
class CUDACache {
public:
    CUDACache(unsigned int w, unsigned int h, const mat4f& intrinsics) : m_width(w), m_height(h), m_intrinsics(intrinsics), m_intrinsicsInv(intrinsics.getInverse()) {}
    std::vector<CUDACachedFrame>& frames() { return m_cache; }
    const CUDACachedFrame* getCacheFramesGPU() const { return m_cache.data(); }
    const std::vector<CUDACachedFrame>& getCacheFrames() const { return m_cache; }
    unsigned int getWidth() const { return m_width; }
    unsigned int getHeight() const { return m_height; }
    const mat4f& getIntrinsics() const { return m_intrinsics; }
    const mat4f& getIntrinsicsInv() const { return m_intrinsicsInv; }
    unsigned int getNumFrames() const { return (unsigned int)m_cache.size(); }
private:
    unsigned int m_width, m_height;
    mat4f m_intrinsics, m_intrinsicsInv;
    std::vector<CUDACachedFrame> m_cache;
};
