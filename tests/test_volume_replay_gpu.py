"""GPU parity test (-m gpu) of the volume operators AT LENGTH: the 2000-frame operator schedule of tests/golden/make_volume_replay_2000.py (2000 integrations, 5835
re-integrations, a garbage collection per frame, on the BASELINE configs[2] stream at 640x480 @4 mm) through bf_scene_run_batch under the exact contract, the whole
volume - block set, every voxel byte, free counter - against the ORACLE's digests at frames 500 / 1000 / 1500 / 2000 (CUDASceneRepHashSDF.h:65-155,
DepthSensing.cpp:854-902).  The product's own 2000-frame loop is compared with the oracle loop in tests/test_pipeline_baseline_gpu.py (trajectories, schedule);
its poses differ from the oracle's by the solver tolerance, so the volume is compared here, where both sides execute the same operator list."""
import importlib.util
import os

import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_volume_replay_2000_frames_vs_oracle_digests(gpu):
    import torch
    spec = importlib.util.spec_from_file_location("make_volume_replay_2000", os.path.join(ROOT, "tests", "golden", "make_volume_replay_2000.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "volume_replay_2000.npz"))
    nf = int(fx["frames"])
    assert nf == g.NF and int(fx["buckets"]) == g.BUCKETS and int(fx["blocks"]) == g.BLOCKS and tuple(fx["lags"]) == g.LAGS
    st = np.load(os.path.join(ROOT, "tests", "golden", "oracle_stream_2000.npz"))
    sched = g.schedule(nf, st["integrated"], st["optimized"])
    W, H = g.W, g.H
    Kd = synth.intrinsics(W, H)
    cam = camera_params(W, H, Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    gs = gpu.capi.SceneRepHashSDF(default_hash_params(num_buckets=g.BUCKETS, num_sdf_blocks=g.BLOCKS, voxel_size=g.VOXEL))
    gs.set_arith("exact")
    gs.set_overlap(True)
    dev, pending, marks = [], [], list(fx["marks_frame"])
    n_in = n_re = 0
    for c0 in range(0, nf, 250):
        part = synth.render_frames(range(c0, min(c0 + 250, nf)), W, H)
        dev += [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in part]
        del part
        for k in range(c0, min(c0 + 250, nf)):
            fixes, Tk = sched[k]
            ops = pending + [("re", told, tnew, dev[f][0], dev[f][1]) for f, told, tnew in fixes]          # the frame loop's batch: the previous frame's integration + this frame's fixes
            if ops:
                gs.run_batch(ops, cam)
            n_re += len(fixes)
            gs.garbage_collect()
            pending = [("in", Tk, None, dev[k][0], dev[k][1])] if Tk is not None else []
            n_in += 1 if Tk is not None else 0
            if k + 1 in marks:
                if pending:
                    gs.run_batch(pending, cam); pending = []
                i = marks.index(k + 1)
                h, heap, cnt, vox = gs.download()
                d = g.digest(h, vox, cnt)
                del h, heap, vox
                print("volume replay, frame %d: %d blocks (oracle %d), free counter %d (%d), %d integrations, %d re-integrations" % (k + 1, d[2], int(fx["num_blocks"][i]), d[3], int(fx["heap_counter"][i]), n_in, n_re))
                assert n_in == int(fx["integrations"][i]) and n_re == int(fx["reintegrations"][i])
                assert d[2] == int(fx["num_blocks"][i]) and d[3] == int(fx["heap_counter"][i]), "frame %d: block count / free counter" % (k + 1)
                assert d[0] == str(fx["blocks_sha256"][i]), "frame %d: the set of allocated blocks differs from the oracle's" % (k + 1)
                assert d[1] == str(fx["voxels_crc_sha256"][i]), "frame %d: voxel bytes differ from the oracle's" % (k + 1)
    assert gs.debug_hash()["dropped"] == int(fx["dropped"][-1]) == 0
