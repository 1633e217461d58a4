"""Writes tests/golden/loop_closure_oracle.npz: the ORACLE frame loop (tests/oracle_pipeline.OraclePipeline, CPU, ~3 minutes) over the
loop-closure stream of tests/test_pipeline_baseline_gpu.py::test_loop_closure_stream_vs_oracle_loop - 212 frames of the synthetic room, 1.8
degrees apart (once around and 12 frames into the second lap), 640x480, chunk size 10, s_numSolveFramesBeforeExit = 2, 4 iterations past the
end (the switch to the dense global solve and the stop).  The GPU test holds the product to these trajectories and counts without paying
for the oracle run on the GPU box; tests/test_pipeline_oracle.py re-derives the fixture with BF_LONG_TESTS=1.

    python tests/golden/make_loop_closure_oracle.py [out.npz]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NF, STRIDE, TAIL, W, H = 212, 9, 4, 640, 480


def params():
    from bundlefusion_amd.capi import default_app_state, default_bundling_state
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.02, 400000, 150000
    gas.s_numSolveFramesBeforeExit = 2
    gbs.s_maxNumImages = NF // 10 + 8
    return gas, gbs


def run():
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import intrinsics_matrix
    from tests.oracle_pipeline import OraclePipeline
    frames = synth.render_frames([STRIDE * k for k in range(NF)], W, H)
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    op = OraclePipeline(*params(), W, H, K)
    op._integrate = lambda frame, T, de: op.integrate_ops.append(("de" if de else "in", frame, np.array(T, np.float32)))      # the volume does not feed back into the poses
    for d, c, _, _ in frames:
        op.process_frame(d, c)
    for _ in range(TAIL):
        op.process_end_of_sequence()
    gc = op.glob.corr[op.glob.corr["imgIdx_i"] != 0xFFFFFFFF]
    span = int((gc["imgIdx_j"].astype(np.int64) - gc["imgIdx_i"].astype(np.int64)).max())
    n_in = sum(1 for k, _, _ in op.integrate_ops if k == "in"); n_de = sum(1 for k, _, _ in op.integrate_ops if k == "de")
    return dict(integrated=op.integrated_trajectory(), optimized=np.stack([op.tm.opt[i] for i in range(NF)]).astype(np.float32),
                counts=np.array([n_in, n_de, op.local.num_solves + op.opt_local.num_solves, op.glob.num_solves]), key_frames=op.glob.num_images, span=span,
                use_global_dense=int(op.glob.use_global_dense), use_solve=int(op.use_solve), num_global_corr=len(gc))


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "loop_closure_oracle.npz")
    r = run()
    np.savez_compressed(out, **r)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in r.items()}, "->", out)
