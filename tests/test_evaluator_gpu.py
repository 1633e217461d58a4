"""GPU tests (-m gpu) of the CorrespondenceEvaluator through the C ABI: the overlap counters and the
precision / recall counts against the numpy restatement (tests/oracle_eval.py) at every stage of the match filter chain, and the
evaluator attached to the frame loop (global key frames of the synthetic stream against its ground-truth trajectory)."""
import os

import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import intrinsics_matrix, CorrespondenceEvaluator, default_app_state, default_bundling_state, sensor_desc
from tests import oracle_eval
from tests.test_match_gpu import _chunk

pytestmark = pytest.mark.gpu


def test_evaluator_counts_equal_restatement_at_every_stage(gpu, oracle, tmp_path):
    n = 6
    frames, K, sift, mgr, cache = _chunk(gpu, n, 12, first=20)
    Kinv = oracle.inverse44(K)
    gw, gh, gk = cache.geometry()
    Kc = intrinsics_matrix(*gk); Kci = oracle.inverse44(Kc)
    T0inv = np.linalg.inv(frames[0][2].astype(np.float64))
    traj = np.stack([(T0inv @ f[2].astype(np.float64)).astype(np.float32) for f in frames])
    traj[2, 0, 3] += 1.5                                   # a wrong reference pose: no overlap by the reference, matches become "incorrect"
    prefix = str(tmp_path / "corr")
    ev = CorrespondenceEvaluator(traj, prefix)
    cur = n - 1
    for k in range(cur):
        mgr.set_valid_image(k, 1)
    mgr.update_gpu_valid_images()
    mgr.set_current_frame(cur)
    mk = mgr.max_keys
    nk = mgr.num_keypoints()
    keys = np.zeros((n * mk, 4), np.float32)
    for i in range(n):
        keys[i * mk:i * mk + nk[i]] = mgr.download_image(i)[0]
    depths = [cache.download_frame(i)["depth"] for i in range(n)]

    def stage_inputs(filtered):
        slots = 25 if filtered else 128
        idx = np.zeros((n, slots, 2), np.uint32); num = np.zeros(n, np.int64)
        for p in range(cur):
            r = mgr.filt_matches(p) if filtered else mgr.raw_matches(p)
            num[p] = r[0]; idx[p, :min(r[0], slots)] = r[1][:min(r[0], slots)]
        return idx, num

    results = {}
    mgr.match(cur, 0, n)
    results["raw"] = (ev.evaluate(mgr, cache, Kinv, False, True, False, "raw"), stage_inputs(False))
    counts, flags = ev.overlap_counts(n)
    # ---- overlap counters and flags: bit-equal with the restatement
    exp_flags = np.zeros(n, np.uint8)
    for p in range(cur):
        A = _mul44(oracle.inverse44(traj[p]), traj[cur])
        c = oracle_eval.overlap_counts(depths[cur], depths[p], A, oracle.inverse44(A), Kc, Kci)
        assert tuple(int(v) for v in counts[p]) == c, (p, counts[p], c)
        exp_flags[p] = oracle_eval.has_gt_overlap(c)
    assert np.array_equal(flags[:cur], exp_flags[:cur]) and exp_flags[:cur].sum() >= 3 and exp_flags[2] == 0
    assert counts[:cur, 1].min() > 2000                    # the comparison is about thousands of pixels per pair
    mgr.filter_keypoint_matches(cur, 0, n, Kinv)
    results["kabsch"] = (ev.evaluate(mgr, cache, Kinv, True, False, False, "kabsch"), stage_inputs(True))
    mgr.filter_surface_area(cur, 0, n, Kinv)
    results["sa"] = (ev.evaluate(mgr, cache, Kinv, True, False, False, "sa"), stage_inputs(True))
    mgr.filter_dense_verify(cur, 0, n, gw, gh, Kc, cache.frames_gpu())
    results["dense"] = (ev.evaluate(mgr, cache, Kinv, True, False, True, "dense"), stage_inputs(True))
    ev.finish_logging()
    n_wrong = 0
    for typ, (got, (idx, num)) in results.items():
        exp, worst = oracle_eval.evaluate(keys, idx, num, exp_flags, traj, cur, Kinv)
        assert got.as_tuple() == exp, (typ, got.as_tuple(), exp)
        assert ev.total(typ).as_tuple() == exp
        n_wrong += sum(1 for p, e in worst.items() if not e < 0.2) + sum(1 for p in range(cur) if exp_flags[p] and num[p] == 0)
    raw, dense = results["raw"][0], results["dense"][0]
    assert raw.numTotal == dense.numTotal == int(exp_flags.sum()) and raw.numDetected >= dense.numDetected >= 2
    assert dense.numCorrect == dense.numDetected and dense.precision() == 1.0 and 0.0 < dense.recall() <= 1.0
    # ---- the log files: header + one row per evaluate() / per wrong or missing pair, in the reference's column order
    rows = open(prefix + "_frame.csv").read().strip().split("\n")
    assert rows[0] == "numFrames,curFrame,type,precision,recall,numCorrect,numDetected,numTotal" and len(rows) == 5
    assert [r.split(",")[2] for r in rows[1:]] == ["raw", "kabsch", "sa", "dense"]
    last = rows[4].split(",")
    assert (int(last[0]), int(last[1])) == (n, cur) and tuple(int(v) for v in last[5:]) == dense.as_tuple()
    wrong = open(prefix + "_wrong.csv").read().strip().split("\n")
    assert wrong[0] == "numFrames,curFrame,matchFrame,type,err" and len(wrong) - 1 == n_wrong and n_wrong >= 1
    assert any(r.split(",")[2] == "2" for r in wrong[1:])  # the pair with the wrong reference pose is reported
    # after clearCache a further stage needs recomputeCache
    with pytest.raises(Exception):
        ev.evaluate(mgr, cache, Kinv, True, False, False, "dense")


def _mul44(a, b):
    out = np.zeros((4, 4), np.float32)
    a = a.astype(np.float32); b = b.astype(np.float32)
    for i in range(4):
        for j in range(4):
            out[i, j] = ((a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]) + a[i, 3] * b[3, j]
    return out


def test_evaluator_attached_to_the_frame_loop(gpu, tmp_path):
    """41 frames = 4 global key frames matched against their predecessors; the reference trajectory is the stream's ground truth.
    Every filter stage is evaluated for every key frame; what survives the dense verification is correct."""
    import torch
    W, H = 640, 480
    n = 41
    frames = synth.render_frames(range(n))
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    gas, gbs = default_app_state(), default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.02, 50000, 20000
    gbs.s_maxNumImages = 8
    gp = gpu.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
    T0inv = np.linalg.inv(frames[0][2].astype(np.float64))
    gt = np.stack([(T0inv @ f[2].astype(np.float64)).astype(np.float32) for f in frames])
    prefix = str(tmp_path / "_corr-evaluation")
    gp.initialize_correspondence_evaluator(gt, prefix)
    for d, c, _, _ in frames:
        assert gp.process_frame(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda())
    gp.synchronize()
    n_global = gp.counters()["global_solves"]
    assert n_global >= 3
    ev = gp.correspondence_evaluator()
    assert ev is not None
    tot = {t: ev.total(t) for t in ("raw", "kabsch", "sa", "dense")}
    assert tot["raw"].numTotal == tot["dense"].numTotal >= 3                 # overlapping key-frame pairs by ground truth
    assert tot["raw"].numDetected >= tot["kabsch"].numDetected >= tot["sa"].numDetected >= tot["dense"].numDetected >= 3
    assert tot["dense"].numCorrect == tot["dense"].numDetected             # everything the filters let through agrees with ground truth
    assert tot["raw"].precision() <= tot["dense"].precision() == 1.0
    gp.finish_correspondence_evaluator_logging()
    rows = open(prefix + "_frame.csv").read().strip().split("\n")
    assert len(rows) - 1 == 4 * n_global and {r.split(",")[2] for r in rows[1:]} == {"raw", "kabsch", "sa", "dense"}
    # the evaluator only observes: the trajectory equals that of a run without it
    gp2 = gpu.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
    for d, c, _, _ in frames:
        gp2.process_frame(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda())
    gp2.synchronize()
    assert np.array_equal(gp.integrated_trajectory().view(np.uint32), gp2.integrated_trajectory().view(np.uint32))
