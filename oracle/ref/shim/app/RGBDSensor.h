// oracle/ref/shim/app/RGBDSensor.h — TEST INFRASTRUCTURE ONLY.  What CUDAImageManager and BundlerInputData read of the sensor
// (CUDAImageManager.h:141-193, CUDAImageManager.cpp:24-25, :41-42, :68-69; OnlineBundlerHelper.h:36-58): the image sizes, intrinsics /
// extrinsics, and the current depth (float metres) and colour (RGBX) frame, which the test supplies.  The real class is the base of the
// device / file readers (sensor SDKs, mLib images, calibration).
#pragma once
#include <vector>
class RGBDSensor {
public:
    RGBDSensor(unsigned int depthW, unsigned int depthH, unsigned int colorW, unsigned int colorH, const mat4f& depthIntrinsics, const mat4f& colorIntrinsics)
        : m_dw(depthW), m_dh(depthH), m_cw(colorW), m_ch(colorH), m_depthIntrinsics(depthIntrinsics), m_depthIntrinsicsInv(depthIntrinsics.getInverse()),
          m_colorIntrinsics(colorIntrinsics), m_extrinsics(mat4f::identity()), m_depth((size_t)depthW * depthH), m_color((size_t)colorW * colorH * 4) {}
    void setFrame(const float* depth, const unsigned char* colorRGBX) { m_depth.assign(depth, depth + m_depth.size()); m_color.assign(colorRGBX, colorRGBX + m_color.size()); }
    bool processDepth() { return m_receiving; }           // false once the test has declared the sequence over (ref_loop.cpp: the compiled frame loop polls the sensor)
    bool processColor() { return true; }
    void setReceiving(bool r) { m_receiving = r; }
    bool isReceivingFrames() const { return m_receiving; }
    mat4f getRigidTransform() const { return mat4f::identity(); }          // (s_binaryDumpSensorUseTrajectory is off)
    void recordFrame() {}
    unsigned int getDepthWidth() const { return m_dw; }
    unsigned int getDepthHeight() const { return m_dh; }
    unsigned int getColorWidth() const { return m_cw; }
    unsigned int getColorHeight() const { return m_ch; }
    const float* getDepthFloat() const { return m_depth.data(); }
    const unsigned char* getColorRGBX() const { return m_color.data(); }
    const mat4f& getDepthIntrinsics() const { return m_depthIntrinsics; }
    const mat4f& getDepthIntrinsicsInv() const { return m_depthIntrinsicsInv; }
    const mat4f& getColorIntrinsics() const { return m_colorIntrinsics; }
    const mat4f& getDepthExtrinsics() const { return m_extrinsics; }
    const mat4f& getDepthExtrinsicsInv() const { return m_extrinsics; }
private:
    unsigned int m_dw, m_dh, m_cw, m_ch;
    mat4f m_depthIntrinsics, m_depthIntrinsicsInv, m_colorIntrinsics, m_extrinsics;
    std::vector<float> m_depth; std::vector<unsigned char> m_color;
    bool m_receiving = true;
};
