"""Writes tests/golden/volume_replay_2000.npz: the ORACLE's TSDF (tests/oracle_api.OracleScene, CPU) over a 2000-frame operator schedule on the BASELINE
configs[2] stream (S2 room, stride 1, 640x480 @4 mm) - digests of the whole volume at frames 500 / 1000 / 1500 / 2000.

Why a replay: the product's own 2000-frame run differs from the oracle's in its poses by ~2 mm (the solver tolerance, tests/test_pipeline_baseline_gpu.py), so its
volume cannot be compared byte for byte with an oracle volume built from the oracle's poses.  Here BOTH sides execute the SAME operator list - which frame, which
pose(s), in which order, with a garbage collection per frame - so the volume operators are held to the oracle bit for bit AT LENGTH (the short suites stop at a
few dozen operators): 2000 integrations, 5835 re-integrations (de-integrate at the old pose + integrate at the new one), 2000 garbage collections, the loop
closing at frame 1800 on top of voxels integrated 1800 frames earlier.

The schedule (deterministic, from the oracle's own trajectories in tests/golden/oracle_stream_2000.npz; `schedule()` below is shared with the GPU test):
  frame k:  up to 3 re-integrations - of frames k - 7, k - 31, k - 127 (where >= 0) - each from the pose the frame currently has in the volume to the other of
            its two poses {integrated[f], optimized[f]} (DepthSensing.cpp:882-889: deIntegrate(old) + integrate(new));
            garbageCollect (:897); integrate(k) at integrated[k] (:1051-1061).
The frames are the rendered depth / colour images themselves (no ingest filter: the volume operators take any depth image).

    python tests/golden/make_volume_replay_2000.py [out.npz] [frames] [threads]          (40 minutes with 6 threads, ~25 GB of host memory)
"""
import hashlib
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H, NF, MARK = 640, 480, 2000, 500
BUCKETS, BLOCKS, VOXEL = 4000000, 3000000, 0.004
LAGS = (7, 31, 127)


def schedule(nf, integrated, optimized):
    """-> per frame k: (list of (frame, old pose, new pose), pose of frame k's integration).  Frames the oracle did not track (pose -inf) are skipped."""
    cur = {}
    out = []
    for k in range(nf):
        fixes = []
        for lag in LAGS:
            f = k - lag
            if f < 0 or f not in cur:
                continue
            which = cur[f][0]
            new = optimized[f] if which == 0 else integrated[f]
            if not np.isfinite(new).all():
                continue
            fixes.append((f, cur[f][1].copy(), new.copy()))
            cur[f] = (1 - which, new.copy())
        Tk = integrated[k] if np.isfinite(integrated[k]).all() else None
        if Tk is not None:
            cur[k] = (0, Tk.copy())
        out.append((fixes, Tk))
    return out


def digest(hash_entries, voxels, heap_counter):
    """sha256 of the sorted block coordinates, sha256 of the per-block crc32 of the voxel bytes in that order, block count, free counter"""
    occ = np.nonzero(hash_entries["ptr"] != -2)[0]
    pos = np.ascontiguousarray(hash_entries["pos"][occ]).astype(np.int32).reshape(-1, 3)
    ptr = hash_entries["ptr"][occ].astype(np.int64)
    order = np.lexsort(pos.T[::-1])
    vb = np.ascontiguousarray(voxels).view(np.uint8).reshape(-1, 512 * 12)
    crc = np.fromiter((zlib.crc32(vb[p // 512]) for p in ptr[order]), np.uint32, count=len(order))
    return hashlib.sha256(pos[order].tobytes()).hexdigest(), hashlib.sha256(crc.tobytes()).hexdigest(), int(len(order)), int(heap_counter)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "volume_replay_2000.npz")
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else NF
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import default_hash_params, camera_params
    from tests import oracle_api
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_stream_2000.npz"))
    sched = schedule(nf, g["integrated"], g["optimized"])
    Kd = synth.intrinsics(W, H)
    cam = camera_params(W, H, Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    p = default_hash_params(num_buckets=BUCKETS, num_sdf_blocks=BLOCKS, voxel_size=VOXEL)
    osc = oracle_api.OracleScene(p)
    frames, marks = [], []
    t0 = time.time()
    n_in = n_re = 0
    for c0 in range(0, nf, 100):
        part = synth.render_frames(range(c0, min(c0 + 100, nf)), W, H, workers=threads)
        frames += [(f[0], f[1]) for f in part]
        for k in range(c0, min(c0 + 100, nf)):
            fixes, Tk = sched[k]
            for f, told, tnew in fixes:
                osc.deintegrate(told, frames[f][0], frames[f][1], cam, threads)
                osc.integrate(tnew, frames[f][0], frames[f][1], cam, threads)
                n_re += 1
            osc.garbage_collect()
            if Tk is not None:
                osc.integrate(Tk, frames[k][0], frames[k][1], cam, threads)
                n_in += 1
            if (k + 1) % MARK == 0 or k + 1 == nf:
                d = digest(osc.hash(), osc.voxels(), osc.heap_counter())
                marks.append((k + 1,) + d + (n_in, n_re, osc.num_dropped()))
                print("checkpoint", marks[-1], flush=True)
        print("%d frames, %.0f s, %d integrations, %d re-integrations, %d blocks" % (min(c0 + 100, nf), time.time() - t0, n_in, n_re, osc.num_allocated()), flush=True)
    np.savez(out, frames=np.int64(nf), marks_frame=np.array([m[0] for m in marks], np.int64), blocks_sha256=np.array([m[1] for m in marks]), voxels_crc_sha256=np.array([m[2] for m in marks]),
             num_blocks=np.array([m[3] for m in marks], np.int64), heap_counter=np.array([m[4] for m in marks], np.int64), integrations=np.array([m[5] for m in marks], np.int64),
             reintegrations=np.array([m[6] for m in marks], np.int64), dropped=np.array([m[7] for m in marks], np.int64), buckets=np.int64(BUCKETS), blocks=np.int64(BLOCKS),
             voxel=np.float64(VOXEL), lags=np.array(LAGS, np.int64), seconds=np.float64(time.time() - t0))
    print("wrote", out)


if __name__ == "__main__":
    main()
