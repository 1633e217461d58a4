"""Writes tests/golden/reference_first_chunk.npz: what the REFERENCE produces on the first local chunk of a synthetic stream
(tests/golden_ref.py describes the contents).  Runs only where /root/reference exists (the build container): the reference's own host classes
and kernels - CUDAImageManager, OnlineBundler, Bundler, the SiftGPU fork, SIFTImageManager, CUDACache, CUDASceneRepHashSDF - are compiled as
they are into oracle/_ref/libbfref.so (oracle/ref/Makefile) and executed on the serial block emulator.  Of the reference's frame loop only the
three calls of DepthSensing.cpp that connect those classes (reintegrate()'s garbageCollect, getCurrentIntegrationFrame -> integrate) are
written out here.

The depth Gauss filter is the one place where the host build differs from a GPU build by more than the order of float additions (glibc exp()
instead of the GPU's fast exp; pinned to 3e-6 on its own): the fixture is generated with the filtered depth of include/bf_detmath.h's fixed
exp() sequence, the one both the product and the oracle use, so that everything downstream is comparable bit for bit.

usage:  python tests/golden/make_reference_golden.py        (from the repository root, after build())
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import golden_ref as G, ref_api  # noqa: E402
from tests import oracle_api as oracle  # noqa: E402
from bundlefusion_amd.capi import camera_params  # noqa: E402

if __name__ == "__main__":
    assert os.path.isdir(ref_api.REFERENCE) and ref_api.available(), "needs /root/reference (build container)"
    gas, gbs = G.params()
    frames, K = G.stream()
    rb = ref_api.RefOnlineBundler(gas, gbs, G.W, G.H, K)
    rsc = ref_api.RefScene(ref_api.hash_params_from_global_app_state(gas), host_class=True)
    Ki = rb.integration_intrinsics()
    cam = camera_params(G.W, G.H, float(Ki[0, 0]), float(Ki[1, 1]), float(Ki[0, 2]), float(Ki[1, 2]), gas.s_renderDepthMin, gas.s_renderDepthMax)
    poses = np.zeros((G.N, 4, 4), np.float32); valid = np.zeros(G.N, bool)
    for i, (d, c, _, _) in enumerate(frames):
        rb.set_frame(d, c)                                                    # CUDAImageManager::process
        raw, _, _, color_i = rb.ingest_outputs()
        filt = oracle.gauss_filter_depth(raw, gbs.s_depthSigmaD, gbs.s_depthSigmaR)      # see the docstring
        rb.override_filtered_depth(filt)
        rb.process_input()                                                    # OnlineBundler::processInput
        rsc.garbage_collect()                                                 # the tail of DepthSensing.cpp: reintegrate() (no frame to fix inside the first chunk)
        ok, T, idx, lost = rb.current_integration_frame()
        valid[i] = ok
        if ok:
            poses[i] = T
            rsc.integrate(T, filt, color_i, cam)                              # DepthSensing.cpp: integrate(); the stored integration frame = filtered depth
        rb.process()                                                          # OnlineBundler::process (nothing to solve inside the first chunk)
    assert rb.state()["last_local_solved"] == -1
    loc = rb.bundler(0)
    k0, d0 = loc.keys(0)
    keys0, desc0_sum = G.sorted_keys(k0, d0)
    blocks, crc, heap_free = G.volume_digest(rsc.hash(), rsc.voxels(), rsc.heap_counter())
    np.savez_compressed(G.PATH, poses=poses, valid=valid, corr=loc.correspondences(), keys0=keys0, desc0_sum=desc0_sum.astype(np.uint32), blocks=blocks,
                        block_crc=crc, heap_free=heap_free)
    print("wrote", G.PATH, os.path.getsize(G.PATH), "bytes:", int(valid.sum()), "tracked frames,", len(loc.correspondences()), "correspondences,", len(keys0),
          "key points,", len(blocks), "blocks")
