"""CPU tests (-m "not gpu"): SE(3) maps, image operators and the bundling-solver oracle.

Independent sanity of the oracle (its pin against the reference's own code is tests/test_ref_pin_cpu.py); the oracle is checked here against independent
maths: scipy rotations for exp/log, numpy.linalg for the Gauss-Newton optimum (the sparse energy is a
sum of squared 3-vectors, so the solver must drive it to the noise floor and recover the ground-truth
poses), a finite-difference check of the dense Jacobians through the energy decrease.
"""
import numpy as np
import pytest

from bundlefusion_amd.capi import ENTRYJ_DTYPE, intrinsics_matrix
from tests import bundle_synth as bs


def test_se3_exp_log_against_scipy(oracle):
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(0)
    T = np.tile(np.eye(4, dtype=np.float32), (200, 1, 1))
    angles = np.r_[rng.uniform(0, 3.13, 150), rng.uniform(0, 1e-3, 50)]
    for i in range(200):
        axis = rng.normal(size=3)
        T[i, :3, :3] = R.from_rotvec(axis / np.linalg.norm(axis) * angles[i]).as_matrix()
        T[i, :3, 3] = rng.normal(size=3)
    rot, tr = oracle.matrices_to_poses(T)
    assert np.abs(rot - R.from_matrix(T[:, :3, :3].astype(np.float64)).as_rotvec()).max() < 2e-6
    T2 = oracle.poses_to_matrices(rot, tr)
    assert np.abs(T2 - T).max() < 5e-6
    # translation part: exp([w, u]) has t = V(w) u ; check with a series-free closed form in float64
    for i in range(0, 200, 17):
        w, u = rot[i].astype(np.float64), tr[i].astype(np.float64)
        th = np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Wx + (th - np.sin(th)) / th ** 3 * Wx @ Wx if th > 1e-6 else np.eye(3) + 0.5 * Wx
        assert np.abs(V @ u - T[i, :3, 3]).max() < 5e-6


def test_sparse_solver_recovers_ground_truth(oracle):
    corr, T_gt, T_init = bs.sparse_problem(n_images=12, noise=0.0005, seed=3)
    rot, tr = oracle.matrices_to_poses(T_init)
    valid = np.ones(12, np.int32)
    w1 = [1.0] * 4
    res = oracle.solver_solve(corr, valid, 12, 4, 100, w1, [0.0] * 4, [0.0] * 4, rot, tr)
    T = oracle.poses_to_matrices(rot, tr)
    dt, dR = bs.pose_errors(T, T_gt)
    conv = res["convergence"][: res["gn_iterations"] + 1]
    assert conv[-1] < 1e-3 * conv[0], conv
    assert np.all(np.diff(conv) < 0)
    assert dt < 2e-3 and dR < 2e-3, (dt, dR)
    assert np.allclose(T[0], np.eye(4), atol=0)          # gauge frame untouched
    assert res["pcg_iterations"][0] > 3


def test_sparse_solver_max_residual_and_verification(oracle):
    corr, T_gt, T_init = bs.sparse_problem(n_images=8, seed=5, outlier_pair=(2, 5))
    rot, tr = oracle.matrices_to_poses(T_init)
    res = oracle.solver_solve(corr, np.ones(8, np.int32), 8, 3, 100, [1.0] * 3, [0.0] * 3, [0.0] * 3, rot, tr)
    worst = corr[res["max_residual_index"]]
    assert (int(worst["imgIdx_i"]), int(worst["imgIdx_j"])) == (2, 5)
    assert res["max_residual"] > 0.08
    assert oracle.solver_use_verification(corr, rot, tr, 8)
    corr2, T_gt2, T_init2 = bs.sparse_problem(n_images=8, seed=5, noise=0.0005)
    rot2, tr2 = oracle.matrices_to_poses(T_init2)
    oracle.solver_solve(corr2, np.ones(8, np.int32), 8, 3, 100, [1.0] * 3, [0.0] * 3, [0.0] * 3, rot2, tr2)
    assert not oracle.solver_use_verification(corr2, rot2, tr2, 8)


def test_invalid_correspondences_and_images_are_ignored(oracle):
    corr, T_gt, T_init = bs.sparse_problem(n_images=6, seed=7, noise=0.0)
    bad = corr.copy()
    bad["pos_j"][::3] += 5.0
    bad["imgIdx_i"][::3] = 0xFFFFFFFF
    bad["imgIdx_j"][::3] = 0xFFFFFFFF
    rot, tr = oracle.matrices_to_poses(T_init)
    oracle.solver_solve(bad, np.ones(6, np.int32), 6, 4, 100, [1.0] * 4, [0.0] * 4, [0.0] * 4, rot, tr)
    dt, dR = bs.pose_errors(oracle.poses_to_matrices(rot, tr), T_gt)
    assert dt < 5e-3 and dR < 5e-3      # GN stops once max|delta| < 0.005 (SolverBundling.cu:1206)


def test_cache_store_frame_matches_numpy(oracle):
    from bundlefusion_amd import synth
    depth, color, T, K = synth.scene_room(40, 160, 120)
    Kin = intrinsics_matrix(K["fx"], K["fy"], K["mx"], K["my"])
    f = oracle.cache_store_frame(depth, color, 80, 60, Kin)
    valid = np.isfinite(f["depth"])
    assert valid.mean() > 0.8
    # camera-space positions are the back-projected (filtered) depth
    ys, xs = np.nonzero(valid)
    xi = (xs * (159 / 79) + 0.5).astype(int); yi = (ys * (119 / 59) + 0.5).astype(int)
    z = f["depth"][valid]
    assert np.allclose(f["campos"][valid][:, 2], z)
    assert np.allclose(f["campos"][valid][:, 0], (xi - K["mx"]) / K["fx"] * z, atol=1e-5)
    assert np.allclose(f["campos"][valid][:, 1], (yi - K["my"]) / K["fy"] * z, atol=1e-5)
    nv = np.isfinite(f["normals"][..., 0])
    n = f["normals"][nv][:, :3]
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-5)
    assert (n[:, 2] > 0).mean() > 0.95                     # cross(dy, dx) / -l points along +z (CUDAImageUtil.cu:424-429)
    assert np.array_equal(f["normals_u"][nv][:, :3], np.round((n + 1) / 2 * 255).astype(np.uint8))
    assert not f["normals_u"][~nv].any()
    assert 0 <= f["intensity"].min() and f["intensity"].max() <= 1
    # gauss filter with a huge range gate and constant depth is the identity
    flat = np.full((40, 50), 2.0, np.float32)
    assert np.allclose(oracle.gauss_filter_depth(flat, 2.0, 0.05), 2.0, atol=1e-6)
    # erosion removes a lone valid pixel in an invalid neighbourhood
    lone = np.full((20, 20), -np.inf, np.float32); lone[10, 10] = 1.0
    assert not np.isfinite(oracle.erode_depth(lone)).any()


def _dense_setup(oracle, n_frames=4, perturb=(0.004, 0.01)):
    frames, K, T_gt, T_init = bs.dense_chunk(n_frames=n_frames, perturb=perturb)
    Kin = intrinsics_matrix(K["fx"], K["fy"], K["mx"], K["my"])
    cache = [oracle.cache_store_frame(d, c, 80, 60, Kin) for d, c in frames]
    geom = (80, 60, [K["fx"] * 80 / 160, K["fy"] * 60 / 120, K["mx"] * 79 / 159, K["my"] * 59 / 119])
    return cache, geom, T_gt, T_init


def test_dense_term_pulls_poses_towards_ground_truth(oracle):
    cache, geom, T_gt, T_init = _dense_setup(oracle)
    n = len(cache)
    rot, tr = oracle.matrices_to_poses(T_init)
    corr = np.zeros(0, dtype=ENTRYJ_DTYPE)
    e0 = bs.pose_errors(T_init, T_gt)
    res = oracle.solver_solve(corr, np.ones(n, np.int32), n, 4, 100, [0.0] * 4, [1.0, 2.0, 3.0, 4.0], [0.0] * 4, rot, tr,
                              cache_frames=cache, cache_geom=geom, dump_dense=True)
    e1 = bs.pose_errors(oracle.poses_to_matrices(rot, tr), T_gt)
    assert res["num_dense_pairs"] == n * (n - 1) // 2
    assert e1[0] < 0.6 * e0[0] and e1[1] < 0.4 * e0[1], (e0, e1)   # point-to-plane alone leaves in-plane sliding
    JtJ = res["JtJ"]
    assert np.allclose(JtJ, JtJ.T)
    assert not JtJ[:6].any() and not JtJ[:, :6].any()       # image 0 is not a variable
    assert np.linalg.eigvalsh(JtJ[6:, 6:].astype(np.float64)).min() > -1e-3
