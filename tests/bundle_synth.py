"""Synthetic bundle-adjustment problems shared by the CPU (oracle) and GPU parity tests."""
import numpy as np

from bundlefusion_amd import synth
from bundlefusion_amd.capi import ENTRYJ_DTYPE


def random_pose(rng, rot_scale, trans_scale):
    from scipy.spatial.transform import Rotation as R
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = R.from_rotvec(rng.normal(size=3) * rot_scale).as_matrix()
    T[:3, 3] = rng.normal(size=3) * trans_scale
    return T


def sparse_problem(n_images=12, pts_per_pair=20, pair_prob=0.5, noise=0.002, perturb=(0.03, 0.05), seed=0, outlier_pair=None):
    """Returns (corr[ENTRYJ], T_gt[N,4,4], T_init[N,4,4]).  Image 0 is the gauge (init == gt)."""
    rng = np.random.default_rng(seed)
    T_gt = np.stack([np.eye(4)] + [random_pose(rng, 0.3, 0.5) for _ in range(n_images - 1)])
    rows = []
    for i in range(n_images):
        for j in range(i + 1, n_images):
            if j != i + 1 and rng.uniform() > pair_prob:
                continue
            pw = rng.uniform(-1, 1, (pts_per_pair, 3)) + np.array([0, 0, 2.5])       # world points
            pi = (np.linalg.inv(T_gt[i]) @ np.c_[pw, np.ones(len(pw))].T).T[:, :3] + rng.normal(0, noise, (len(pw), 3))
            pj = (np.linalg.inv(T_gt[j]) @ np.c_[pw, np.ones(len(pw))].T).T[:, :3] + rng.normal(0, noise, (len(pw), 3))
            if outlier_pair == (i, j):
                pj += 0.5
            for a, b in zip(pi, pj):
                rows.append((i, j, a, b))
    corr = np.zeros(len(rows), dtype=ENTRYJ_DTYPE)
    for k, (i, j, a, b) in enumerate(rows):
        corr[k] = (i, j, a.astype(np.float32), b.astype(np.float32))
    T_init = T_gt.copy()
    for i in range(1, n_images):
        T_init[i] = random_pose(rng, perturb[0], perturb[1]) @ T_gt[i]
    return corr, T_gt.astype(np.float32), T_init.astype(np.float32)


def pose_errors(T, T_gt):
    """max translation error [m] and max rotation error [rad] against ground truth."""
    dt = np.abs(T[:, :3, 3] - T_gt[:, :3, 3]).max()
    dR = 0.0
    for a, b in zip(T, T_gt):
        c = (np.trace(a[:3, :3].astype(np.float64).T @ b[:3, :3].astype(np.float64)) - 1) / 2
        dR = max(dR, float(np.arccos(np.clip(c, -1, 1))))
    return float(dt), dR


def dense_chunk(n_frames=4, stride=6, width=160, height=120, perturb=(0.004, 0.01), seed=0):
    """n_frames of scene S2, poses relative to frame 0; returns frames [(depth,color)], K dict, T_gt, T_init."""
    rng = np.random.default_rng(seed)
    frames, poses = [], []
    K = None
    for k in range(n_frames):
        d, c, T, K = synth.scene_room(k * stride, width, height)
        frames.append((d, c))
        poses.append(T.astype(np.float64))
    T0inv = np.linalg.inv(poses[0])
    T_gt = np.stack([T0inv @ p for p in poses])
    T_init = T_gt.copy()
    for i in range(1, n_frames):
        T_init[i] = random_pose(rng, perturb[0], perturb[1]) @ T_gt[i]
    return frames, K, T_gt.astype(np.float32), T_init.astype(np.float32)
