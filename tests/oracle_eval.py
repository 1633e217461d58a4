"""CPU restatement (numpy, float32, one rounding per operation) of the reference's CorrespondenceEvaluator — test infrastructure
only (used by tests/test_evaluator_gpu.py).  Paths relative to /root/reference/FriedLiver/Source.

  overlap_counts      computeCorrespondences (CorrespondenceEvaluator.cpp:98-224) in both directions: numCorr / numValid
  has_gt_overlap      computeOverlap (.cpp:226-250) + the threshold of computeCachedData (.cpp:39-43)
  evaluate            evaluate (.cpp:47-96)

Pinned to the reference's CorrespondenceEvaluator.cpp compiled as it is (oracle/ref/ref_evaluator.cpp, tests/test_ref_pin_cpu.py::
test_correspondence_evaluator_vs_reference_class): overlap flags and (numCorrect, numDetected, numTotal) equal.  The class works on mLib
types (the submodule is absent from the reference tree) which oracle/ref/shim/mlib_standin.h supplies as plain containers; three mLib
conventions remain assumptions on both sides: mat4f * vec4f is taken as the row sums in index order, vec3f's default constructor as
(0,0,0), math::round as round-half-away-from-zero (floor(x + 0.5) here: the two differ only for negative exact halves, which are outside the image anyway).
"""
import numpy as np

F = np.float32
NINF = F(-np.inf)


def _campos(depth, Kinv):            # computeCameraSpacePositions .cpp:300-311
    H, W = depth.shape
    x = np.arange(W, dtype=F)[None, :].repeat(H, 0); y = np.arange(H, dtype=F)[:, None].repeat(W, 1)
    d = depth.astype(F)
    with np.errstate(invalid="ignore", over="ignore"):
        vx, vy = x * d, y * d
        out = np.empty((H, W, 3), F)
        for r in range(3):
            out[..., r] = ((Kinv[r, 0] * vx + Kinv[r, 1] * vy) + Kinv[r, 2] * d) + Kinv[r, 3] * d
    out[d == NINF] = NINF
    return out


def _normals(P):                     # computeNormals .cpp:271-298
    H, W, _ = P.shape
    N = np.zeros((H, W, 3), F)       # PointImage::allocate: vec3f() = (0,0,0)
    N[0, :] = NINF; N[-1, :] = NINF; N[:, 0] = NINF; N[:, -1] = NINF
    CC, PC, CP, MC, CM = P[1:-1, 1:-1], P[2:, 1:-1], P[1:-1, 2:], P[:-2, 1:-1], P[1:-1, :-2]
    ok = (CC[..., 0] != NINF) & (PC[..., 0] != NINF) & (CP[..., 0] != NINF) & (MC[..., 0] != NINF) & (CM[..., 0] != NINF)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        a, b = PC - MC, CP - CM
        n = np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                      a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F)
        l = np.sqrt((n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1]) + n[..., 2] * n[..., 2]).astype(F)
        nn = (n / (-l)[..., None]).astype(F)
    inner = N[1:-1, 1:-1]
    inner[ok & (l > 0)] = nn[ok & (l > 0)]
    inner[ok & ~(l > 0)] = NINF
    return N


def _mul4(T, v, w):                  # mat4f * vec4f(v, w): row sums left to right
    out = []
    for r in range(4):
        out.append(((T[r, 0] * v[..., 0] + T[r, 1] * v[..., 1]) + T[r, 2] * v[..., 2]) + T[r, 3] * F(w))
    return out


def _direction(d0, d1, T, K, Kinv, depth_min, depth_max, dist_thresh, normal_thresh):
    H, W = d0.shape
    T = T.astype(F); K = K.astype(F); Kinv = Kinv.astype(F)
    P0, P1 = _campos(d0, Kinv), _campos(d1, Kinv)
    N0, N1 = _normals(P0), _normals(P1)
    src = (P0[..., 0] != NINF) & (N0[..., 0] != NINF)
    num_valid = int((src & (d0 > F(depth_min)) & (d0 < F(depth_max))).sum())
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        px, py, pz, pw = _mul4(T, P0, 1.0)
        nx, ny, nz, nw = _mul4(T, N0, 0.0)
        pT = np.stack([px, py, pz], -1)
        qx = ((K[0, 0] * px + K[0, 1] * py) + K[0, 2] * pz) + K[0, 3]
        qy = ((K[1, 0] * px + K[1, 1] * py) + K[1, 2] * pz) + K[1, 3]
        qz = ((K[2, 0] * px + K[2, 1] * py) + K[2, 2] * pz) + K[2, 3]
        qw = ((K[3, 0] * px + K[3, 1] * py) + K[3, 2] * pz) + K[3, 3]
        hx, hy, hz = qx / qw, qy / qw, qz / qw
        fx, fy = np.floor(hx / hz + F(0.5)), np.floor(hy / hz + F(0.5))
    fx = np.where(np.isnan(fx), 0.0, np.clip(fx, -2147483648.0, 2147483647.0)); fy = np.where(np.isnan(fy), 0.0, np.clip(fy, -2147483648.0, 2147483647.0))
    sx, sy = fx.astype(np.int64), fy.astype(np.int64)
    inside = src & (sx >= 0) & (sy >= 0) & (sx < W) & (sy < H)
    sxc, syc = np.clip(sx, 0, W - 1), np.clip(sy, 0, H - 1)
    pt, nt = P1[syc, sxc], N1[syc, sxc]
    tgt = inside & (pt[..., 0] != NINF) & (nt[..., 0] != NINF)
    with np.errstate(invalid="ignore", over="ignore"):
        e = pT - pt
        ew = pw - F(1.0)
        d = np.sqrt(((e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]) + ew * ew).astype(F)
        dn = ((nx * nt[..., 0] + ny * nt[..., 1]) + nz * nt[..., 2]) + nw * F(0.0)
        corr = tgt & (dn >= F(normal_thresh)) & (d <= F(dist_thresh))
    return int(corr.sum()), num_valid


def overlap_counts(depth_cur, depth_prv, T_cur_to_prv, T_prv_to_cur, K, Kinv, depth_min=0.5, depth_max=4.0, dist_thresh=0.15, normal_thresh=0.97):
    c0, v0 = _direction(depth_cur, depth_prv, T_cur_to_prv, K, Kinv, depth_min, depth_max, dist_thresh, normal_thresh)
    c1, v1 = _direction(depth_prv, depth_cur, T_prv_to_cur, K, Kinv, depth_min, depth_max, dist_thresh, normal_thresh)
    return c0, v0, c1, v1


def has_gt_overlap(c, min_overlap=0.1):
    c0, v0, c1, v1 = (int(x) for x in c)
    with np.errstate(invalid="ignore", divide="ignore"):
        p0 = F(c0) / F(v0)
        oc, ov = c0, v0
        if not (p0 > F(min_overlap)):
            p1 = F(c1) / F(v1)
            if not (p0 > p1):
                oc, ov = c1, v1
        o = F(oc) / F(ov)
    return bool(ov > 0 and o > F(min_overlap))


def _xf(M, v):                       # mat4f * vec3f: affine product, then the division by w
    M = M.astype(F); v = np.asarray(v, F)
    r = [((M[i, 0] * v[0] + M[i, 1] * v[1]) + M[i, 2] * v[2]) + M[i, 3] for i in range(4)]
    return np.array([r[0] / r[3], r[1] / r[3], r[2] / r[3]], F)


def evaluate(keys, idx, num_matches, has_gt, ref_traj, cur_frame, Kinv_sift, max_err=0.2):
    """keys (N,4) float32 rows x,y,scale,depth addressed by the match indices; idx (numFrames, slots, 2); returns (numCorrect,
    numDetected, numTotal) and the per-image maximum squared error (for the _wrong.csv rows)."""
    n_frames = len(num_matches)
    correct = detected = total = 0
    worst = {}
    for p in range(n_frames):
        if p == cur_frame:
            continue
        nm = max(int(num_matches[p]), 0)
        if has_gt[p]:
            total += 1
            if nm > 0:
                detected += 1
        max_err2 = F(0.0)
        for m in range(min(nm, idx.shape[1])):
            k0, k1 = keys[idx[p, m, 0]], keys[idx[p, m, 1]]
            cp0 = _xf(Kinv_sift, [k0[3] * k0[0], k0[3] * k0[1], k0[3] * F(1.0)])
            cp1 = _xf(Kinv_sift, [k1[3] * k1[0], k1[3] * k1[1], k1[3] * F(1.0)])
            e = _xf(ref_traj[p], cp0) - _xf(ref_traj[cur_frame], cp1)
            err2 = F((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2])
            if err2 > max_err2:
                max_err2 = err2
        if nm > 0:
            worst[p] = float(max_err2)
            if max_err2 < F(max_err) and has_gt[p]:
                correct += 1
    return (correct, detected, total), worst
