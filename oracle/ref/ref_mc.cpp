// oracle/ref/ref_mc.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libbfref.so, the parity pin of the CPU oracle).
//
// The REFERENCE's marching cubes: DepthSensing/CUDAMarchingCubesSDF.cu (extractIsoSurfaceKernel, one thread per voxel of every hash
// slot) with MarchingCubesSDFUtil.h (extractIsoSurfaceAtPosition, vertexInterp, appendTriangle), RayCastSDFUtil.h
// (trilinearInterpolationSimpleFastFast) and its case tables Tables.h, compiled from where they lie.  This file only sequences
// resetMarchingCubesCUDA + extractIsoSurfaceCUDA like CUDAMarchingCubesHashSDF::extractIsoSurface (.cpp:107-119) and hands out the
// triangle buffer and the two tables.
#include "CUDAMarchingCubesSDF.cu.cpp"      // = cu2cpp.py < reference file (generated into the build's temporary directory)
#include "ref_scene.h"

RayCastParams c_rayCastParams;
extern "C" void updateConstantRayCastParams(const RayCastParams& p) { c_rayCastParams = p; }

extern "C" {

unsigned int ref_mc_extract(ref_scene* s, float thresh, float thresh2, int boxEnabled, const float* minCorner, const float* maxCorner, float* out18, unsigned int maxTriangles) {
    MarchingCubesParams p; memset(&p, 0, sizeof p);
    p.m_maxNumTriangles = maxTriangles; p.m_threshMarchingCubes = thresh; p.m_threshMarchingCubes2 = thresh2;
    p.m_sdfBlockSize = SDF_BLOCK_SIZE; p.m_hashBucketSize = HASH_BUCKET_SIZE; p.m_hashNumBuckets = s->params.m_hashNumBuckets;
    p.m_boxEnabled = boxEnabled != 0;
    if (minCorner) p.m_minCorner = make_float3(minCorner[0], minCorner[1], minCorner[2]);
    if (maxCorner) p.m_maxCorner = make_float3(maxCorner[0], maxCorner[1], maxCorner[2]);
    MarchingCubesData d;
    d.allocate(p, true);
    d.updateParams(p);
    s->data.updateParams(s->params);
    RayCastData rc;
    resetMarchingCubesCUDA(d);
    extractIsoSurfaceCUDA(s->data, rc, p, d);
    const unsigned int n = *d.d_numTriangles;
    static_assert(sizeof(MarchingCubesData::Triangle) == 72, "Triangle layout");
    memcpy(out18, d.d_triangles, sizeof(MarchingCubesData::Triangle) * (size_t)(n < maxTriangles ? n : maxTriangles));
    d.free();
    return n;
}

void ref_mc_tables(unsigned short* edge256, signed char* tri256x16) {
    for (int i = 0; i < 256; ++i) {
        edge256[i] = (unsigned short)edgeTable[i];
        for (int k = 0; k < 16; ++k) tri256x16[i * 16 + k] = (signed char)triTable[i][k];
    }
}

}  // extern "C"
