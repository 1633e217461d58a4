"""GPU (-m gpu): the PRODUCT against the golden vectors the REFERENCE ITSELF produced (tests/golden/reference_first_chunk.npz; how they were
made: tests/golden/make_reference_golden.py, contents: tests/golden_ref.py) - bit for bit, through the C ABI, without the oracle in between.
The CPU twin, tests/test_golden_ref_cpu.py, holds the oracle frame loop to the same fixture."""
import ctypes as C

import numpy as np
import pytest

from tests import golden_ref as G
from bundlefusion_amd.capi import sensor_desc

pytestmark = pytest.mark.gpu


def test_pipeline_reproduces_the_reference_bit_for_bit(gpu):
    """The first local chunk of the synthetic stream through bf_pipeline_*: the poses handed to the integration, the chunk's correspondences, the
    key points of frame 0 and the volume (block set, every voxel byte, free list) equal what the reference's own classes and kernels produced."""
    import torch
    g = np.load(G.PATH)
    gas, gbs = G.params()
    frames, K = G.stream()
    gp = gpu.capi.Pipeline(gas, gbs, sensor_desc(G.W, G.H, K))
    for d, c, _, _ in frames:
        assert gp.process_frame(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda())
    gp.synchronize()
    traj = gp.integrated_trajectory()
    assert len(traj) == G.N
    h = C.c_void_p(); gpu.capi.check(gpu.capi.lib.bf_bundler_get_sift_manager(gp.bundler("local"), C.byref(h)))
    mgr = gpu.capi.SiftManager.__new__(gpu.capi.SiftManager)
    mgr._h = h; mgr.max_keys = gbs.s_maxNumKeysPerImage; mgr.max_images = gbs.s_submapSize + 1
    corr, _ = mgr.download_global_correspondences()
    k0, d0 = mgr.download_image(0)
    mgr._h = C.c_void_p()                       # borrowed handle: not ours to destroy
    keys0, dsum0 = G.sorted_keys(k0, d0)
    sc = gp.scene()
    gh, _, gcnt, gvox = sc.download()
    blocks, crc, free = G.volume_digest(gh, gvox, gcnt)
    G.check(g, traj, np.isfinite(traj[:, 0, 0]), corr, keys0, dsum0, blocks, crc, free, "product")
