#!/usr/bin/env python3
"""What does the frame loop cost WITHOUT the volume beside it?  The driver's window (bench.py: frames 205-224 of the S2 stream behind a 200-frame pre-roll,
640x480 @4 mm, library defaults) through the same pipeline with the reference's own switches (GlobalAppState: s_integrationEnabled, s_reconstructionEnabled):

    full          integration + re-integration (bench.py's `value`)
    no_reint      integration only (s_maxFrameFixes = 0: the re-integration queue of DepthSensing.cpp:854-902 is never served)
    bundling      neither (s_integrationEnabled = 0: the loop is ingest, SIFT, matching chain, solves)

    python tools/loop_parts.py [repeats]      -> one JSON line per leg (frames/s, calling thread's ms per frame by stage)
A diagnostic, not a benchmark: it tells how much of the frame time is the volume stream's kernels slowing everything else down (profiles/r06_loop_schedule.md)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import intrinsics_matrix, default_app_state, default_bundling_state, sensor_desc
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    W, H, pre, warm, steps = 640, 480, 200, 5, 20
    total = pre + warm + steps
    frames = synth.render_frames(range(total), W, H, workers=min(64, os.cpu_count() or 1))
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    feed = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    wa = torch.randn(4096, 4096, device="cuda"); wb = torch.randn(4096, 4096, device="cuda")
    tw = time.perf_counter()
    while time.perf_counter() - tw < 3.0:
        for _ in range(8):
            wa = torch.mm(wa, wb) * 1e-2
        torch.cuda.synchronize()
    legs = (("full", 1, None), ("no_reint", 1, 0), ("bundling", 0, None))
    names = ("enqueue_match_chain", "ingest_detect_enqueue", "reintegrate_commands", "wait_match_result", "integrate_command", "solves", "wait_ingest")
    for rep in range(repeats):
        for name, integ, fixes in legs:
            gas = default_app_state(); gbs = default_bundling_state()
            gas.s_integrationWidth, gas.s_integrationHeight = W, H
            gas.s_SDFVoxelSize = 0.004
            gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 1000000, 600000
            gas.s_integrationEnabled = integ
            if fixes is not None:
                gas.s_maxFrameFixes = fixes
            gbs.s_maxNumImages = total // 10 + 8
            pipe = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
            for k in range(pre + warm):
                assert pipe.process_frame(*feed[k])
            pipe.synchronize(); torch.cuda.synchronize()
            pipe.host_profile(reset=True)
            t0 = time.perf_counter()
            for k in range(pre + warm, total):
                assert pipe.process_frame(*feed[k])
            pipe.synchronize(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            hp = pipe.host_profile()
            traj = pipe.integrated_trajectory()
            print(json.dumps({"leg": name, "repeat": rep, "frames_per_s": round(steps / dt, 1), "ms_per_frame": round(1e3 * dt / steps, 4),
                              "host_thread_ms_per_frame": {n: round(1e3 * float(hp[n]) / max(hp["frames"], 1.0), 4) for n in names},
                              "frames_valid": int(np.isfinite(traj[:, 0, 0]).sum())}), flush=True)
            del pipe
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
