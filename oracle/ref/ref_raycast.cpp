// oracle/ref/ref_raycast.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's ray cast kernel: DepthSensing/CUDARayCastSDF.cu (renderKernel, one thread per pixel) with RayCastSDFUtil.h
// (traverseCoarseGridSimpleSampleAll, findIntersectionBisection, trilinearInterpolationSimpleFastFast, gradientForPoint), compiled from
// where they lie.  The two ray-interval images are INPUTS here: the reference fills them with a D3D11 rasteriser pass
// (DX11RayIntervalSplatting.cpp), which cannot run in this container; renderCS binds them as textures exactly as it does the mapped
// render targets.  This file only fills RayCastParams / RayCastData and calls renderCS like CUDARayCastSDF::render (.cpp:42-72).
#include "CUDARayCastSDF.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)
#include "ref_scene.h"

extern "C" void updateConstantRayCastParams(const RayCastParams& p);      // ref_mc.cpp holds c_rayCastParams

extern "C" {

// params: viewMatrix[16], viewMatrixInverse[16], then mx,my,fx,fy, width,height, minDepth,maxDepth, rayIncrement,thresSampleDist,thresDist, useGradients
void ref_rc_render(ref_scene* s, const float* viewMatrix, const float* viewMatrixInverse, const float* intr4, unsigned int width, unsigned int height,
                   const float* f5, int useGradients, const float* rayMin, const float* rayMax, float* depth, float* depth4, float* normals, float* colors) {
    RayCastParams p; memset(&p, 0, sizeof p);
    memcpy(&p.m_viewMatrix, viewMatrix, 64); memcpy(&p.m_viewMatrixInverse, viewMatrixInverse, 64);
    p.mx = intr4[0]; p.my = intr4[1]; p.fx = intr4[2]; p.fy = intr4[3];
    p.m_width = width; p.m_height = height;
    p.m_minDepth = f5[0]; p.m_maxDepth = f5[1]; p.m_rayIncrement = f5[2]; p.m_thresSampleDist = f5[3]; p.m_thresDist = f5[4];
    p.m_useGradients = useGradients != 0;
    updateConstantRayCastParams(p);
    s->data.updateParams(s->params);
    RayCastData rc;
    rc.d_depth = depth; rc.d_depth4 = (float4*)depth4; rc.d_normals = (float4*)normals; rc.d_colors = (float4*)colors;
    cudaArray amin = {rayMin, width, height, width * sizeof(float)}, amax = {rayMax, width, height, width * sizeof(float)};
    rc.d_rayIntervalSplatMinArray = &amin; rc.d_rayIntervalSplatMaxArray = &amax;
    renderCS(s->data, rc, p);
}

}  // extern "C"
