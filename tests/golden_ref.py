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


# ---- the second fixture: a well-conditioned 121-frame stream with global solves and re-integration scheduling in it (tests/golden/make_reference_stream.py)
def stream_fixture():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_stream", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_reference_stream.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def check_stream(g, online, final, key_frames, ops, frames, who, pose_tol=1e-3):
    """online: (NF, 4, 4) the pose every frame was integrated at when it arrived (-inf: not tracked); final: the complete optimised trajectory;
    ops: (n, 2) int (kind 0 integrate / 1 de-integrate / 2 re-integrate, frame) in issue order.
    pose_tol: largest deviation of any pose element, eleven global solves deep - measured oracle vs reference 5.2e-4 (the solver's own pin is 1e-4 per
    solve: float summation order); the ATE bar is north_star's 1 mm (measured difference 0.008 mm online, 0.12 mm final)."""
    online, final = np.asarray(online, np.float32), np.asarray(final, np.float32)
    vo, vg = np.isfinite(online[:, 0, 0]), np.isfinite(g["online"][:, 0, 0])
    assert np.array_equal(vo, vg), who + ": tracked frames differ from the reference's"
    vf, vgf = np.isfinite(final[:, 0, 0]), np.isfinite(g["final"][:, 0, 0])
    assert np.array_equal(vf, vgf), who + ": frames with an optimised pose differ from the reference's"
    assert int(key_frames) == int(g["key_frames"]), who + ": %d key frames, the reference has %d" % (int(key_frames), int(g["key_frames"]))
    d_on, d_fin = float(np.abs(online[vo] - g["online"][vo]).max()), float(np.abs(final[vf] - g["final"][vf]).max())
    T0inv = np.linalg.inv(frames[0][2].astype(np.float64))
    ref = np.stack([T0inv @ f[2].astype(np.float64) for f in frames])

    def ate(t, v):
        return float(np.sqrt(np.mean(np.sum((t[v][:, :3, 3] - ref[v][:, :3, 3]) ** 2, axis=1))))
    a = dict(online=(ate(online, vo), ate(g["online"], vo)), final=(ate(final, vf), ate(g["final"], vf)))
    go = g["ops"]
    if isinstance(ops, dict):            # the product reports counters: integrations / de-integrations (a re-integration counts in both)
        want = (int(((go[:, 0] == 0) | (go[:, 0] == 2)).sum()), int(((go[:, 0] == 1) | (go[:, 0] == 2)).sum()))
        got = (ops["integrate"], ops["deintegrate"])
        print("%s: %d integrations / %d de-integrations, the reference schedules %d / %d" % (who, got[0], got[1], want[0], want[1]))
        assert abs(got[0] - want[0]) <= 0.02 * want[0] + 2 and abs(got[1] - want[1]) <= 0.02 * want[1] + 2, who + ": operation counts differ from the reference's schedule"
        ops = go
    same_ops = len(ops) == len(go) and np.array_equal(np.asarray(ops, np.int32).reshape(-1, 2), go)
    print("%s vs the REFERENCE on the 121-frame stream: max pose deviation online %.2e final %.2e; ATE online %.3f / %.3f mm, final %.3f / %.3f mm (%s / reference); "
          "%d key frames; TSDF operations %d (reference %d)%s" % (who, d_on, d_fin, 1e3 * a["online"][0], 1e3 * a["online"][1], 1e3 * a["final"][0], 1e3 * a["final"][1], who,
                                                               int(key_frames), len(ops), len(go), ", identical schedule" if same_ops else ""))
    assert d_on < pose_tol and d_fin < pose_tol, who + ": pose deviation from the reference %.2e / %.2e" % (d_on, d_fin)
    assert abs(a["online"][0] - a["online"][1]) < 1e-3 and abs(a["final"][0] - a["final"][1]) < 1e-3, who + ": ATE differs from the reference's by more than 1 mm"
    # the re-integration schedule depends on the ranking of pose differences: the same operations up to a few swaps
    assert abs(len(ops) - len(go)) <= 0.02 * len(go) + 2, who + ": %d TSDF operations, the reference schedules %d" % (len(ops), len(go))
    return d_on, d_fin, a
