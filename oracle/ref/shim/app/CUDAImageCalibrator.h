// oracle/ref/shim/app: the Direct3D depth-to-colour re-projection (s_bUseCameraCalibration, off in zParametersDefault.txt and for every
// sensor whose depth and colour cameras coincide); CUDAImageManager holds one as a member and calls it only under that flag
#pragma once
#include "RGBDSensor.h"          // as the real header (CUDAImageCalibrator.h:3-4)
#include "GlobalAppState.h"
class CUDAImageCalibrator {
public:
    HRESULT OnD3D11CreateDevice(ID3D11Device*, unsigned int, unsigned int) { return 0; }
    void OnD3D11DestroyDevice() {}
    HRESULT process(ID3D11DeviceContext*, float*, const mat4f&, const mat4f&, const mat4f&) { throw std::runtime_error("CUDAImageCalibrator is not part of the path"); }
};
