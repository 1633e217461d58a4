"""Writes tests/golden/oracle_stream_2000.npz: the ORACLE frame loop (tests/oracle_pipeline.OraclePipeline, CPU) over BASELINE configs[2] at its stated
length - the S2 room stream from frame 0, stride 1, 640x480, 2000 frames: the camera closes its loop at frame 1800 and frames 1800..1999 re-observe the
start.  This is what `bench.py`'s `long_stream` block runs on the GPU (same frames, same parameters: zParametersDefault.txt /
zParametersBundlingDefault.txt values, s_maxNumImages = 2000 / 10 + 8); the lengths exercise what the short fixtures cannot - the re-integration queue
saturated for hundreds of key frames (DepthSensing.cpp:854-902), ~200 key frames in the global problem, the loop closure (OnlineBundler.cpp:181-186
stays off: the sequence is not ended).

The volume does not feed back into the poses (DepthSensing.cpp:966-1095: integration only consumes them), so the oracle's TSDF operators are LOGGED, not
executed: the fixture holds the trajectories, the validity of every frame, the key-frame count, and the scheduled-operation counts at every 500 frames.
tests/test_pipeline_baseline_gpu.py::test_config2_stream_2000_vs_oracle_fixture holds the product to it (same frames valid, same key frames, same operation
schedule per 500 frames, |delta ATE| < 1 mm, per-pose bound).

    python tests/golden/make_oracle_stream_2000.py [out.npz] [frames] [bob]          (~40-60 minutes on 8 cores; `... oracle_stream_5000.npz 5000 0.3` writes the 5000-frame fixture)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H, NF, MARK = 640, 480, 2000, 500


def params(nf=NF):
    from bundlefusion_amd.capi import default_app_state, default_bundling_state
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize = 0.004
    gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 1000, 1000          # the oracle's volume is not used (operators are logged): do not allocate 3 M blocks on the host
    gbs.s_maxNumImages = nf // 10 + 8
    return gas, gbs


def run(nf=NF, verbose=True, bob=0.0):
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import intrinsics_matrix
    from tests.oracle_pipeline import OraclePipeline
    Kd = synth.intrinsics(W, H)
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    op = OraclePipeline(*params(nf), W, H, K)
    op._integrate = lambda frame, T, de: op.integrate_ops.append(("de" if de else "in", frame, np.array(T, np.float32)))
    op.scene.garbage_collect = lambda: None
    poses, marks = [], []
    t0 = time.time()
    for c0 in range(0, nf, 100):
        part = synth.render_frames(range(c0, min(c0 + 100, nf)), W, H, bob=bob)
        for d, c, T, _ in part:
            op.process_frame(d, c)
            op.frames[-1] = None                                      # the filtered frame is only needed by the (logged) volume operators
            poses.append(T)
            k = len(poses)
            if k % MARK == 0 or k == nf:
                n_in = sum(1 for kind, _, _ in op.integrate_ops if kind == "in"); n_de = len(op.integrate_ops) - n_in
                marks.append([k, n_in, n_de, op.local.num_solves + op.opt_local.num_solves, op.glob.num_solves, op.glob.num_images])
        if verbose:
            print("%d frames, %.0f s, key frames %d, ops %d" % (len(poses), time.time() - t0, op.glob.num_images, len(op.integrate_ops)), flush=True)
    T0inv = np.linalg.inv(poses[0].astype(np.float64))
    gt = np.stack([T0inv @ T.astype(np.float64) for T in poses]).astype(np.float32)
    integ = op.integrated_trajectory()
    opt = np.stack([op.tm.opt[i] for i in range(nf)]).astype(np.float32)
    return dict(integrated=integ, optimized=opt, ground_truth=gt, marks=np.array(marks, np.int64), key_frames=op.glob.num_images,
                frames=nf, seconds=time.time() - t0)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "oracle_stream_2000.npz")
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else NF
    bob = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # 0.3: BASELINE configs[3] (5000 frames: 2.5 loops + vertical sinusoid) -> oracle_stream_5000.npz
    r = run(nf, bob=bob)
    r["bob"] = np.float64(bob)
    np.savez_compressed(out, **r)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in r.items()}, "->", out)
