"""CPU tests (-m "not gpu"): descriptor matching + match filters of the oracle.

Independent sanity of the oracle (its pin against the reference's own kernels is tests/test_ref_pin_cpu.py).  Checked here against independent numpy / float64
evaluations: the McAdams SVD reconstructs its input, Kabsch recovers a known rigid motion, the matcher
finds a planted permutation and applies distance / ratio / mutual tests, the greedy Kabsch filter keeps
inliers and rejects a planted outlier, and the SIFT -> match -> filter chain on two synthetic frames
recovers the ground-truth relative pose.
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import rgbx_to_intensity, intrinsics_matrix


def _unit_descs(rng, n):
    d = np.abs(rng.normal(size=(n, 128)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.clip(np.floor(d * 512 + 0.5), 0, 255).astype(np.uint8)


def test_svd3_reconstructs(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(3, 3)).astype(np.float32)
        U, S, V = oracle.svd3(A)
        assert np.abs(U @ S @ V.T - A).max() < 2e-5
        assert np.abs(U @ U.T - np.eye(3)).max() < 1e-5 and np.abs(V @ V.T - np.eye(3)).max() < 1e-5
        assert np.abs(S - np.diag(np.diag(S))).max() < 2e-3 * np.abs(S).max()


def test_kabsch_recovers_rigid_motion(oracle):
    from tests.bundle_synth import random_pose
    rng = np.random.default_rng(1)
    for _ in range(10):
        T = random_pose(rng, 0.4, 0.5)
        src = rng.uniform(-1, 1, (12, 3)) + [0, 0, 2]
        tgt = (T[:3, :3] @ src.T).T + T[:3, 3]
        Tk, ev = oracle.kabsch(src, tgt)
        assert np.abs(Tk - T).max() < 3e-3        # 4 approximate-Givens sweeps (cuda_svd3.h) — not a converged SVD
        assert ev[0] >= ev[1] >= ev[2] >= 0


def test_match_planted_permutation(oracle):
    rng = np.random.default_rng(2)
    d1 = _unit_descs(rng, 90)
    perm = rng.permutation(90)[:60]
    d2 = d1[perm].copy()
    noise = rng.integers(-2, 3, d2.shape)
    d2 = np.clip(d2.astype(int) + noise, 0, 255).astype(np.uint8)
    d2 = np.concatenate([d2, _unit_descs(rng, 25)])
    n, idx, dist = oracle.sift_match(d1, d2, off1=1000, off2=5000)
    assert n >= 58
    got = {(int(a) - 1000, int(b) - 5000) for a, b in idx}
    assert got <= {(int(p), k) for k, p in enumerate(perm)} | got and len(got & {(int(p), k) for k, p in enumerate(perm)}) >= 58
    assert np.all(np.diff(dist) >= 0)
    # unsorted output is in ascending column order (canonical replacement for the atomic append)
    n2, idx2, _ = oracle.sift_match(d1, d2, sort=False)
    assert n2 == n and np.all(np.diff(idx2[:, 1].astype(int)) > 0)
    # exact duplicates in d2 kill the ratio test for that row
    d3 = np.concatenate([d2, d2[:5]])
    n3, idx3, _ = oracle.sift_match(d1, d3)
    assert not (set(idx3[:, 1].tolist()) & set(range(5))) and not (set(idx3[:, 1].tolist()) & set(range(85, 90)))
    # empty sides
    assert oracle.sift_match(d1[:0], d2)[0] == 0 and oracle.sift_match(d1, d2[:0])[0] == 0


def test_match_caps_at_128(oracle):
    rng = np.random.default_rng(3)
    d1 = _unit_descs(rng, 300)
    n, idx, dist = oracle.sift_match(d1, d1.copy())
    assert n == 300 and len(idx) == 128
    assert np.array_equal(idx[:, 0], idx[:, 1]) and np.all(dist < 0.1)       # |d|^2 is 2^18 only up to uchar rounding


def _keys_from_points(P, K):
    uv = (K[:3, :3] @ P.T).T
    return np.c_[uv[:, 0] / uv[:, 2], uv[:, 1] / uv[:, 2], np.full(len(P), 3.0), P[:, 2]].astype(np.float32)


def test_kabsch_filter_inliers_and_outlier(oracle):
    from tests.bundle_synth import random_pose
    rng = np.random.default_rng(4)
    K = intrinsics_matrix(580.0, 580.0, 320.0, 240.0)
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    T = random_pose(rng, 0.1, 0.2)
    P = np.c_[rng.uniform(-1, 1, 40), rng.uniform(-0.8, 0.8, 40), rng.uniform(1.5, 3.0, 40)]
    Q = (T[:3, :3] @ P.T).T + T[:3, 3]
    Q[7] += [0.3, -0.2, 0.1]                              # planted outlier, early in the distance order
    keys = np.concatenate([_keys_from_points(P, K), _keys_from_points(Q, K)])
    idx = np.zeros((128, 2), np.uint32); idx[:40, 0] = np.arange(40); idx[:40, 1] = 40 + np.arange(40)
    dist = np.zeros(128, np.float32); dist[:40] = np.linspace(0.1, 0.5, 40)
    n, fidx, fdist, Tf = oracle.filter_matches(keys, idx, dist, 40, Kinv)
    assert 5 <= n <= 25 and 7 not in fidx[:, 0]
    assert np.abs(Tf - T).max() < 2e-3
    # too few raw matches -> rejected
    assert oracle.filter_matches(keys, idx, dist, 4, Kinv)[0] == 0
    # pure garbage -> rejected
    keys2 = keys.copy(); keys2[40:, :2] = rng.uniform(50, 400, (40, 2)); keys2[40:, 3] = rng.uniform(1, 3, 40)
    assert oracle.filter_matches(keys2, idx, dist, 40, Kinv)[0] == 0
    # surface area: spread points pass, a tight cluster fails
    ok, areas = oracle.filter_surface_area(keys, fidx, Kinv)
    assert ok and areas.max() > 0.032
    tight = np.c_[rng.uniform(-0.03, 0.03, 10), rng.uniform(-0.03, 0.03, 10), rng.uniform(2.0, 2.02, 10)]
    k3 = np.concatenate([_keys_from_points(tight, K), _keys_from_points(tight + 0.001, K)])
    i3 = np.c_[np.arange(10), 10 + np.arange(10)].astype(np.uint32)
    ok, areas = oracle.filter_surface_area(k3, i3, Kinv)
    assert not ok and areas.max() < 0.032


def test_sift_match_filter_chain_recovers_pose(oracle):
    fa, fb = 40, 46
    da, ca, Ta, Kd = synth.scene_room(fa, 640, 480)
    db, cb, Tb, _ = synth.scene_room(fb, 640, 480)
    na, ka, desa, _ = oracle.sift_run(rgbx_to_intensity(ca), da)
    nb, kb, desb, _ = oracle.sift_run(rgbx_to_intensity(cb), db)
    assert na > 30 and nb > 30
    n, idx, dist = oracle.sift_match(desa, desb, off1=0, off2=na)
    assert n >= 8
    keys = np.concatenate([ka, kb])
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    nf, fidx, fdist, Tf = oracle.filter_matches(keys, np.concatenate([idx, np.zeros((128 - len(idx), 2), np.uint32)]),
                                                np.concatenate([dist, np.zeros(128 - len(dist), np.float32)]), min(n, 128), Kinv)
    assert nf >= 5
    rel = np.linalg.inv(Tb.astype(np.float64)) @ Ta.astype(np.float64)     # frame a -> frame b
    assert np.abs(Tf - rel)[:3, 3].max() < 0.02 and np.abs(Tf[:3, :3] - rel[:3, :3]).max() < 0.02
    # dense verification at the cache resolution accepts the pose and rejects a wrong one
    W, H = 80, 60
    fra = oracle.cache_store_frame(da, ca, W, H, K)
    frb = oracle.cache_store_frame(db, cb, W, H, K)
    Ks = K.copy(); Ks[0, 0] *= W / 640; Ks[1, 1] *= H / 480; Ks[0, 2] *= (W - 1) / 639; Ks[1, 2] *= (H - 1) / 479
    ok, err, corr = oracle.dense_verify(fra, frb, W, H, Ks, Tf)
    assert ok and err < 0.075 and corr > 0.3
    bad = Tf.copy(); bad[0, 3] += 0.4
    ok2, err2, corr2 = oracle.dense_verify(fra, frb, W, H, Ks, bad)
    assert not ok2
