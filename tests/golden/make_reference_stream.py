"""Writes tests/golden/reference_stream_121.npz: the REFERENCE's own trajectory on a WELL-CONDITIONED stream with global solves in it.

The first-chunk fixture (make_reference_golden.py) ends before any pose is optimised.  This one runs the emulated reference (its own host classes and
kernels: CUDAImageManager, OnlineBundler, Bundler, SBA, CUDASolverBundling, the SiftGPU fork, TrajectoryManager - oracle/_ref/libbfref.so on the serial
block emulator; build container only) over 121 frames of the synthetic room 0.2 degrees apart at 320x240 - twelve local chunks, eleven global solves,
re-integration scheduling throughout - where consecutive frames overlap almost completely, every chunk is solved from a good initial guess and the raw
match cap of 128 (ProgramCU.cu:1909, the one place the reference's result depends on its thread arrival order) is not reached, so the reference's
result is a function of its input.  Stored: the pose every frame was handed to the integration with when it arrived ("online"), the final complete
trajectory, the counts.  tests/test_golden_ref_cpu.py holds the oracle loop to it, tests/test_golden_ref_gpu.py the product on the MI355X:
per-pose 5e-4 (the solver's float tolerance), |ATE difference| < 1 mm (north_star's bar), same tracked frames, same number of key frames.

usage:  python tests/golden/make_reference_stream.py      (from the repository root, after build(); ~4 minutes)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H, NF, STRIDE, TAIL = 320, 240, 121, 1, 3
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_stream_121.npz")


def params():
    from bundlefusion_amd.capi import default_app_state, default_bundling_state
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.02, 50000, 20000
    gbs.s_widthSIFT, gbs.s_heightSIFT, gbs.s_maxNumImages = W, H, NF // 10 + 4
    return gas, gbs


def stream():
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import intrinsics_matrix
    frames = synth.render_frames([STRIDE * k for k in range(NF)], W, H)
    Kd = frames[0][3]
    return frames, intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])


if __name__ == "__main__":
    from tests import ref_api as R
    from tests.oracle_pipeline import OraclePipeline, _minf
    assert os.path.isdir(R.REFERENCE) and R.available(), "needs /root/reference (build container)"
    gas, gbs = params()
    frames, K = stream()
    op = OraclePipeline(*params(), W, H, K)           # only its ingest (the depth filter with the deterministic exp of bf_detmath.h) is used
    rb = R.RefOnlineBundler(gas, gbs, W, H, K); rtm = rb.trajectory_manager()
    online = np.full((NF, 4, 4), -np.inf, np.float32)
    ops = []                                          # the TSDF operations DepthSensing.cpp's reintegrate() / integrate() would issue: (kind, frame)
    t0 = time.time()
    for i in range(NF + TAIL):
        if i < NF:
            d, c = frames[i][0], frames[i][1]
            raw, filt = op._ingest(d, c); rb.set_frame(d, c); rb.override_filtered_depth(filt)
            rb.process_input()
            ok, T, idx, lost = rb.current_integration_frame()
        else:
            rb.process_input(); ok = False
        if rtm.active() < gas.s_maxFrameFixes:
            rtm.generate()
        for _ in range(gas.s_maxFrameFixes):          # DepthSensing.cpp:854-902 (no volume: the operations are logged)
            f, ix, TT, _ = rtm.top_de()
            if f:
                ops.append((1, ix)); continue
            f, ix, TT, _ = rtm.top_in()
            if f:
                ops.append((0, ix)); rtm.confirm(ix); continue
            f, ix, o_, n_ = rtm.top_re()
            if f:
                ops.append((2, ix)); rtm.confirm(ix); continue
            break
        if i < NF:
            if ok:
                online[i] = T; ops.append((0, i))
            rtm.add(0 if ok else 1, T if ok else _minf(), i)
        rb.process()
        if i % 10 == 0:
            print("reference frame %d  %.0f s" % (i, time.time() - t0), flush=True)
    st = rb.state()
    n = st["num_complete"]
    final = np.full((NF, 4, 4), -np.inf, np.float32)
    final[:min(n, NF)] = rb.complete_trajectory(n)[:NF]
    g = rb.bundler(2)
    np.savez_compressed(PATH, online=online, final=final, key_frames=g.num_frames(), global_corr=len(g.correspondences()), num_complete=n,
                        ops=np.array(ops, np.int32).reshape(-1, 2))
    print("wrote", PATH, os.path.getsize(PATH), "bytes; tracked", int(np.isfinite(online[:, 0, 0]).sum()), "of", NF, "frames;", g.num_frames(), "key frames;",
          len(ops), "TSDF operations of which", sum(1 for k, _ in ops if k == 2), "re-integrations")
