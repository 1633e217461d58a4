// oracle/ref/shim/app/CUDAImageManager.h — TEST INFRASTRUCTURE ONLY.  What Bundler / OnlineBundler read of the image manager
// (CUDAImageManager.h:223-233 copyToBundling, :244 getCurrFrameNumber, :282-290 the SIFT-side depth size and intrinsics): the current
// frame's raw depth, filtered depth and colour, supplied by the glue.  The real class owns the sensor and the ingest
// (CUDAImageManager::process), whose kernels are pinned on their own (ref_image.cpp).
#pragma once
#include "RGBDSensor.h"               // the includes of the real header that its users rely on (CUDAImageManager.h:2-6)
#include "CUDAImageUtil.h"
#include "GlobalBundlingState.h"
#include "TimingLog.h"
#include <vector>
class CUDAImageManager {
public:
    CUDAImageManager(unsigned int depthW, unsigned int depthH, unsigned int colorW, unsigned int colorH, const mat4f& depthIntrinsics)
        : m_depthW(depthW), m_depthH(depthH), m_colorW(colorW), m_colorH(colorH), m_depthIntrinsics(depthIntrinsics) {}
    void setFrame(const float* depthRaw, const float* depthFilt, const uchar4* color) {     // = one CUDAImageManager::process()
        m_raw.assign(depthRaw, depthRaw + (size_t)m_depthW * m_depthH); m_filt.assign(depthFilt, depthFilt + (size_t)m_depthW * m_depthH);
        m_color.assign(color, color + (size_t)m_colorW * m_colorH); ++m_numFrames;
    }
    void copyToBundling(float* d_depthRaw, float* d_depthFilt, uchar4* d_color) const {
        if (d_depthRaw) memcpy(d_depthRaw, m_raw.data(), sizeof(float) * m_raw.size());
        if (d_depthFilt) memcpy(d_depthFilt, m_filt.data(), sizeof(float) * m_filt.size());
        if (d_color) memcpy(d_color, m_color.data(), sizeof(uchar4) * m_color.size());
    }
    unsigned int getCurrFrameNumber() const { return m_numFrames - 1; }
    const unsigned int getSIFTDepthWidth() const { return m_depthW; }
    const unsigned int getSIFTDepthHeight() const { return m_depthH; }
    const mat4f& getSIFTDepthIntrinsics() const { return m_depthIntrinsics; }
private:
    unsigned int m_depthW, m_depthH, m_colorW, m_colorH, m_numFrames = 0;
    mat4f m_depthIntrinsics;
    std::vector<float> m_raw, m_filt; std::vector<uchar4> m_color;
};
