// oracle/ref/shim/app/RGBDSensor.h — TEST INFRASTRUCTURE ONLY.  What BundlerInputData::alloc reads of the sensor (OnlineBundlerHelper.h:36-58):
// the image sizes and the colour intrinsics.  The real class is the base of the device / file readers (sensor SDKs, mLib).
#pragma once
class RGBDSensor {
public:
    RGBDSensor(unsigned int depthW, unsigned int depthH, unsigned int colorW, unsigned int colorH, const mat4f& colorIntrinsics)
        : m_dw(depthW), m_dh(depthH), m_cw(colorW), m_ch(colorH), m_colorIntrinsics(colorIntrinsics) {}
    unsigned int getDepthWidth() const { return m_dw; }
    unsigned int getDepthHeight() const { return m_dh; }
    unsigned int getColorWidth() const { return m_cw; }
    unsigned int getColorHeight() const { return m_ch; }
    const mat4f& getColorIntrinsics() const { return m_colorIntrinsics; }
private:
    unsigned int m_dw, m_dh, m_cw, m_ch;
    mat4f m_colorIntrinsics;
};
