"""Golden vectors produced by the REFERENCE ITSELF (tests/golden/reference_first_chunk.npz, written by tests/golden/make_reference_golden.py
in the build container, where /root/reference exists): its own host classes and kernels, compiled into oracle/_ref and run on the block
emulator over the first local chunk of a synthetic stream.  The fixture travels; the reference does not.  Shared by the CPU test (oracle frame
loop vs the fixture) and the GPU test (product vs the fixture): what is compared is compared BIT FOR BIT - no pose has been optimised yet in
the first chunk, so there is no solver tolerance in any of it.

Fixture contents (all little-endian numpy arrays):
  poses      (N, 4, 4) f4   the pose OnlineBundler::getCurrentIntegrationFrame handed to the integration for every frame
  valid      (N,) bool
  corr       EntryJ rows of the running local chunk after the last frame (Bundler::matchAndFilter -> AddCurrToResidualsCU), in order
  keys0      (K, 4) f4      SIFT key points of frame 0 (x, y, scale, depth), sorted by their bit patterns (the reference's list order
                            depends on its thread scheduling)
  desc0_sum  (K,) u4        sum of the 128 descriptor bytes of each of those key points, same order (descriptors agree to 1 count in
                            <= 0.1 % of the bytes between the reference's fast-math and the fixed sequences: compared to +-2)
  blocks     (B, 3) i4      allocated SDF block coordinates after the last integration, sorted
  block_crc  (B,) u4        zlib.crc32 of each block's 512 voxels x 12 bytes, same order
  heap_free  ()  u4         the free-list counter
"""
import os
import zlib

import numpy as np

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix, FREE_ENTRY, VOX_PER_BLOCK

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_first_chunk.npz")
W, H, N = 320, 240, 10


def params():
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.02, 20000, 8000
    gbs.s_widthSIFT, gbs.s_heightSIFT, gbs.s_maxNumImages = W, H, 4
    return gas, gbs


def stream():
    frames = [synth.scene_room(3 * k, W, H) for k in range(N)]
    Kd = frames[0][3]
    return frames, intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])


def sorted_keys(keys, descs):
    """key rows sorted by their bit patterns + the byte sum of each key's descriptor in that order"""
    k = np.ascontiguousarray(keys, np.float32).reshape(-1, 4)
    order = np.lexsort(k.view(np.uint32).T[::-1])
    return k[order], np.asarray(descs, np.uint8).reshape(len(k), 128).astype(np.uint32).sum(axis=1)[order]


def volume_digest(hash_entries, voxels, heap_counter):
    """(sorted block coordinates, crc32 of each block's voxel bytes, free counter) of a volume in the reference memory layout"""
    occ = hash_entries[hash_entries["ptr"] != FREE_ENTRY]
    pos = np.array([[int(v) for v in e["pos"]] for e in occ], np.int32).reshape(-1, 3)
    order = np.lexsort(pos.T[::-1])
    vb = np.ascontiguousarray(voxels).view(np.uint8).reshape(-1, VOX_PER_BLOCK * 12)
    crc = np.array([zlib.crc32(vb[int(occ[i]["ptr"]) // VOX_PER_BLOCK].tobytes()) for i in order], np.uint32)
    return pos[order], crc, np.uint32(heap_counter)


def check(g, poses, valid, corr, keys0, desc0_sum, blocks, block_crc, heap_free, what):
    """bit-for-bit comparison of one implementation's results with the fixture `g` (np.load of PATH)"""
    assert np.array_equal(np.asarray(valid, bool), g["valid"]), what + ": tracked flags"
    assert np.array_equal(np.asarray(poses, np.float32)[g["valid"]].view(np.uint32), g["poses"][g["valid"]].view(np.uint32)), what + ": tracked poses"
    assert np.asarray(corr).tobytes() == g["corr"].tobytes(), what + ": correspondences of the chunk"
    assert np.array_equal(np.asarray(keys0, np.float32).view(np.uint32), g["keys0"].view(np.uint32)), what + ": key points of frame 0"
    assert np.abs(np.asarray(desc0_sum, np.int64) - g["desc0_sum"].astype(np.int64)).max() <= 2, what + ": descriptors of frame 0"
    assert np.array_equal(np.asarray(blocks, np.int32), g["blocks"]), what + ": allocated blocks"
    assert np.array_equal(np.asarray(block_crc, np.uint32), g["block_crc"]), what + ": voxel bytes"
    assert int(heap_free) == int(g["heap_free"]), what + ": free list"
