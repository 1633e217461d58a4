"""Run the headless BundleFusion frame loop on a synthetic S2 stream and report poses / timings.

usage: python tools/run_sequence.py [--frames N] [--voxel 0.01] [--host] [--timings]
"""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bundlefusion_amd as bf
from bundlefusion_amd import synth
from bundlefusion_amd.capi import intrinsics_matrix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--voxel", type=float, default=0.01)
    ap.add_argument("--width", type=int, default=640, help="sensor (depth = colour) width; SIFT stays at 640x480")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--buckets", type=int, default=800000)
    ap.add_argument("--blocks", type=int, default=200000)
    ap.add_argument("--host", action="store_true", help="hand over host buffers (PCIe path) instead of HBM-resident frames")
    ap.add_argument("--tail", type=int, default=5, help="end-of-sequence iterations")
    ap.add_argument("--timings", action="store_true")
    ap.add_argument("--timings-from", type=int, default=-1, help="switch the synchronous per-stage timings on at this frame")
    ap.add_argument("--bob", type=float, default=0.0, help="vertical sinusoid amplitude of the trajectory [m] (SURVEY.md 8d config 4: 0.3)")
    ap.add_argument("--maximages", type=int, default=0)
    a = ap.parse_args()
    W, H = a.width, a.height
    idx = [a.first + k * a.stride for k in range(a.frames)]
    frames = []
    dev_all = []
    for c0 in range(0, len(idx), 512):                      # render and upload in batches: a 5000-frame stream is 12 GB
        part = synth.render_frames(idx[c0:c0 + 512], W, H, bob=a.bob)
        if a.host:
            frames += part
        else:
            dev_all += [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in part]
            frames += [(None, None, f[2], f[3]) for f in part]
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    gas = bf.capi.default_app_state(); gbs = bf.capi.default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize = a.voxel; gas.s_hashNumBuckets = a.buckets; gas.s_hashNumSDFBlocks = a.blocks
    gbs.s_maxNumImages = a.maximages or max(8, a.frames // 10 + 4)
    gbs.s_widthSIFT, gbs.s_heightSIFT = 640, 480
    bf.capi.bind_host_threads_to_device(0)
    p = bf.capi.Pipeline(gas, gbs, bf.capi.sensor_desc(W, H, K))
    if os.environ.get("BF_TSDF_ARITH"):
        print("arithmetic contract of the voxel update:", p.scene().arith())
    if a.timings:
        p.enable_timings(True)
    dev = dev_all if not a.host else None
    torch.cuda.synchronize()
    t0 = time.time()
    tl = []
    marks = []
    for k in range(a.frames):
        if k == a.timings_from:
            p.synchronize(); p.enable_timings(True); a.timings = True
        ok = p.process_frame(*(dev[k] if dev else (frames[k][0], frames[k][1])))
        assert ok
        if a.timings:
            tl.append(dict(p.last_timing(), frame=k))
        if (k + 1) % 1000 == 0:
            p.synchronize(); marks.append((k + 1, time.time() - t0))
    for _ in range(a.tail):
        p.process_end_of_sequence()
    p.synchronize()
    dt = time.time() - t0
    print("frames %d  wall %.3f s  -> %.1f frames/s (incl. %d end-of-sequence iterations)" % (a.frames, dt, a.frames / dt, a.tail))
    prev = (0, 0.0)
    for m in marks:
        print("  frames %d-%d: %.1f frames/s" % (prev[0], m[0], (m[0] - prev[0]) / (m[1] - prev[1]))); prev = m
    print("counters", p.counters())
    T0inv = np.linalg.inv(frames[0][2].astype(np.float64))
    gt = np.stack([(T0inv @ f[2].astype(np.float64)) for f in frames])
    for name, traj in (("integrated", p.integrated_trajectory()), ("optimized", p.optimized_trajectory())):
        ok = np.isfinite(traj[:, 0, 0])
        n = len(traj)
        if ok.any():
            err = np.linalg.norm(traj[ok][:, :3, 3] - gt[:n][ok][:, :3, 3], axis=1)
            print("%s: %d/%d valid, translation error mean %.4f max %.4f m" % (name, ok.sum(), n, err.mean(), err.max()))
        else:
            print("%s: %d frames, none valid" % (name, n))
    sc = p.scene()
    print("allocated blocks", sc.num_allocated_blocks(), "heap free", sc.heap_free_count(), "debug", sc.debug_hash())
    if tl:
        keys = [k for k in tl[0].keys() if k != "frame"]
        for name, sel in (("all frames", lambda f: True), ("chunk-end frames", lambda f: f % 10 == 9), ("other frames", lambda f: f % 10 != 9)):
            arr = np.array([[t[k] for k in keys] for t in tl[2:] if sel(t["frame"])])
            if len(arr):
                print(name, "mean ms per stage:", {k: round(float(v), 3) for k, v in zip(keys, arr.mean(0))})
                print(name, "max  ms per stage:", {k: round(float(v), 3) for k, v in zip(keys, arr.max(0))})


if __name__ == "__main__":
    main()
