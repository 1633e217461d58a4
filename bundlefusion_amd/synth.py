"""Deterministic synthetic RGB-D input: scenes S1 ("wall") and S2 ("room").

Pure numpy; shared by the tests and bench.py so that the CPU oracle and the HIP path always
see byte-identical inputs.  Not part of the product's compute path.

Camera: pinhole fx=fy=583, cx=319.5, cy=239.5 at 640x480 (scaled with resolution), depth in
metres as float32 with -inf for invalid pixels (2-px border, depth > 4 m), colour RGBX8.
"""
import numpy as np

MINF = np.float32(-np.inf)


def intrinsics(width=640, height=480):
    s = width / 640.0
    return dict(fx=583.0 * s, fy=583.0 * s, mx=(width - 1) / 2.0, my=(height - 1) / 2.0)


# ---------------------------------------------------------------- PCG32 + value noise
def pcg32(seed, n, seq=54):
    """n uint32 outputs of PCG-XSH-RR 64/32 (O'Neill), vectorised over nothing: plain loop on uint64."""
    mult = np.uint64(6364136223846793005)
    inc = np.uint64((seq << 1) | 1)
    state = np.uint64(0)
    out = np.empty(n, dtype=np.uint32)
    old = np.seterr(over="ignore")
    try:
        state = state * mult + inc
        state = state + np.uint64(seed)
        state = state * mult + inc
        for i in range(n):
            o = state
            state = o * mult + inc
            xs = np.uint32(((o >> np.uint64(18)) ^ o) >> np.uint64(27))
            rot = np.uint32(o >> np.uint64(59))
            out[i] = (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))
    finally:
        np.seterr(**old)
    return out


_LATTICE = {}


def _lattice(seed, n=64):
    key = (seed, n)
    if key not in _LATTICE:
        _LATTICE[key] = (pcg32(seed, n * n).astype(np.float64) / 4294967296.0).reshape(n, n)
    return _LATTICE[key]


def value_noise(u, v, seed, octaves=6, base_freq=4.0, decay=0.7):
    """Multi-octave value noise in [0,1) at texture coords (u, v) in metres.  SURVEY.md §8d asks for 3 octaves
    (4/8/16 cycles per metre); at the 0.4-1 m viewing distances of scene S2 that leaves no structure at SIFT
    scales (3-25 px = 5-40 mm), so 3 more octaves (32/64/128 cycles per metre) are added."""
    acc = np.zeros_like(u, dtype=np.float64)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        lat = _lattice(seed + 101 * o)
        n = lat.shape[0]
        fu, fv = u * base_freq * (2 ** o), v * base_freq * (2 ** o)
        iu, iv = np.floor(fu).astype(np.int64), np.floor(fv).astype(np.int64)
        tu, tv = fu - iu, fv - iv
        tu, tv = tu * tu * (3 - 2 * tu), tv * tv * (3 - 2 * tv)
        a = lat[iu % n, iv % n]
        b = lat[(iu + 1) % n, iv % n]
        c = lat[iu % n, (iv + 1) % n]
        d = lat[(iu + 1) % n, (iv + 1) % n]
        acc += amp * ((a * (1 - tu) + b * tu) * (1 - tv) + (c * (1 - tu) + d * tu) * tv)
        tot += amp
        amp *= decay
    return acc / tot


def _rgbx(u, v, seed):
    r = value_noise(u, v, seed)
    g = value_noise(u + 17.3, v - 5.1, seed + 7)
    b = value_noise(u - 9.7, v + 23.9, seed + 13)
    img = np.stack([r, g, b, np.ones_like(r)], axis=-1)
    return np.clip(np.floor(img * 256.0), 0, 255).astype(np.uint8)


def _finish_depth(depth, max_depth=4.0, border=2):
    d = depth.astype(np.float32)
    d[~np.isfinite(d)] = MINF
    d[d > max_depth] = MINF
    d[d <= 0] = MINF
    if border:
        d[:border, :] = MINF
        d[-border:, :] = MINF
        d[:, :border] = MINF
        d[:, -border:] = MINF
    return d


# ---------------------------------------------------------------- S1: relief wall
def add_depth_noise(depth, seed=42):
    """SURVEY.md 8d's optional sensor noise: Gaussian in depth with sigma_z = 0.0012 + 0.0019 (z - 0.4)^2 metres (the Kinect model the
    survey names), from a seeded generator; invalid pixels (-inf) stay invalid.  Returns a new float32 array."""
    rng = np.random.RandomState(seed)
    d = np.asarray(depth, np.float32)
    ok = np.isfinite(d)
    z = np.where(ok, d, np.float32(0.0)).astype(np.float64)
    sigma = 0.0012 + 0.0019 * (z - 0.4) ** 2
    out = (z + rng.standard_normal(d.shape) * sigma).astype(np.float32)
    return np.where(ok, out, d).astype(np.float32)


def scene_wall(width=640, height=480):
    """S1: plane z = 2.0 + 0.10 sin(2 pi x/0.8) sin(2 pi y/0.6), albedo noise seed 1234; identity pose."""
    K = intrinsics(width, height)
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    kx, ky = (xs - K["mx"]) / K["fx"], (ys - K["my"]) / K["fy"]
    z = np.full_like(kx, 2.0)
    for _ in range(30):
        z = 2.0 + 0.10 * np.sin(2 * np.pi * z * kx / 0.8) * np.sin(2 * np.pi * z * ky / 0.6)
    color = _rgbx(z * kx, z * ky, 1234)
    return _finish_depth(z), color, np.eye(4, dtype=np.float32), K


# ---------------------------------------------------------------- S2: textured room + clutter
ROOM_MIN = np.array([-3.0, 0.0, -2.0])
ROOM_MAX = np.array([3.0, 3.0, 2.0])     # 6 x 3(h) x 4 box, y is up-down axis (camera y points down)


def _clutter_boxes():
    r = pcg32(4321, 8 * 6).astype(np.float64) / 4294967296.0
    r = r.reshape(8, 6)
    boxes = []
    for i in range(8):
        ang = 2 * np.pi * (i + 0.35 * r[i, 0]) / 8.0
        ca, sa = np.cos(ang), np.sin(ang)
        wall = min(3.0 / max(abs(ca), 1e-9), 2.0 / max(abs(sa), 1e-9))
        rad = 1.0 + (0.45 + 0.3 * r[i, 1]) * (wall - 1.0)
        cx, cz = rad * ca, rad * sa
        sx, sy, sz = 0.12 + 0.2 * r[i, 2], 1.2 + 1.2 * r[i, 3], 0.12 + 0.2 * r[i, 4]
        lo = np.array([cx - sx, ROOM_MAX[1] - sy, cz - sz])     # standing on the floor (y = +3 is the floor)
        hi = np.array([cx + sx, ROOM_MAX[1], cz + sz])
        boxes.append((np.maximum(lo, ROOM_MIN + 0.05), np.minimum(hi, ROOM_MAX)))
    return boxes


_BOXES = None


def trajectory_pose(k, frames_per_loop=1800, radius=1.0, height=1.5, bob=0.0):
    """Camera-to-world of frame k: on a circle of radius 1 m, eye height 1.5 m, looking outward."""
    a = 2 * np.pi * k / frames_per_loop
    fwd = np.array([np.cos(a), 0.0, np.sin(a)])          # camera +z
    down = np.array([0.0, 1.0, 0.0])                     # camera +y (image rows grow downward)
    right = np.cross(down, fwd)                          # camera +x
    pos = np.array([radius * np.cos(a), ROOM_MAX[1] - height + bob * np.sin(5 * a), radius * np.sin(a)])
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = right, down, fwd, pos
    return T.astype(np.float32)


def _ray_aabb(o, d, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    tn = np.minimum(t0, t1)
    tf = np.maximum(t0, t1)
    axis_n = np.argmax(tn, axis=-1)
    tnear = np.max(tn, axis=-1)
    axis_f = np.argmin(tf, axis=-1)
    tfar = np.min(tf, axis=-1)
    return tnear, tfar, axis_n, axis_f


def scene_room(k, width=640, height=480, frames_per_loop=1800, bob=0.0):
    """S2 frame k: depth f32 [H,W], colour u8 [H,W,4], camera-to-world 4x4 f32, intrinsics."""
    global _BOXES
    if _BOXES is None:
        _BOXES = _clutter_boxes()
    K = intrinsics(width, height)
    T = trajectory_pose(k, frames_per_loop, bob=bob).astype(np.float64)
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dc = np.stack([(xs - K["mx"]) / K["fx"], (ys - K["my"]) / K["fy"], np.ones_like(xs)], axis=-1)
    d = dc @ T[:3, :3].T
    o = T[:3, 3]
    # room: exit point
    _, tfar, _, axis_f = _ray_aabb(o, d, ROOM_MIN, ROOM_MAX)
    best_t = tfar
    best_face = axis_f * 2 + (np.take_along_axis(d, axis_f[..., None], -1)[..., 0] > 0)
    best_seed = 1234 + best_face
    for bi, (lo, hi) in enumerate(_BOXES):
        tn, tf, axis_n, _ = _ray_aabb(o, d, lo, hi)
        hit = (tn < tf) & (tn > 1e-4) & (tn < best_t)
        best_t = np.where(hit, tn, best_t)
        best_face = np.where(hit, axis_n * 2, best_face)
        best_seed = np.where(hit, 2000 + 10 * bi + axis_n, best_seed)
    p = o + d * best_t[..., None]
    axis = best_face // 2
    ua = (axis + 1) % 3
    va = (axis + 2) % 3
    u = np.take_along_axis(p, ua[..., None], -1)[..., 0]
    v = np.take_along_axis(p, va[..., None], -1)[..., 0]
    color = np.zeros((height, width, 4), dtype=np.uint8)
    for s in np.unique(best_seed):
        m = best_seed == s
        color[m] = _rgbx(u[m], v[m], int(s))
    depth = best_t          # camera-space z == t because dc.z == 1
    return _finish_depth(depth), color, T.astype(np.float32), K


# ---------------------------------------------------------------- parallel rendering of a stream
def render_frames(indices, width=640, height=480, workers=None, frames_per_loop=1800, bob=0.0):
    """[scene_room(k) for k in indices], rendered by `workers` plain-python subprocesses that execute this file
    (no torch / HIP in the children, so it is safe to call from a process that already holds a GPU context)."""
    import os
    import subprocess
    import sys
    import tempfile
    indices = list(indices)
    workers = max(1, min(workers or min(os.cpu_count() or 1, 64), len(indices)))
    if workers == 1 or len(indices) < 4:
        return [scene_room(k, width, height, frames_per_loop, bob) for k in indices]
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for w in range(workers):
            part = indices[w::workers]
            out = os.path.join(td, "part%d.npz" % w)
            procs.append((part, out, subprocess.Popen([sys.executable, os.path.abspath(__file__), out, str(width), str(height), str(frames_per_loop), repr(bob)]
                                                      + [str(k) for k in part])))
        res = {}
        for part, out, p in procs:
            if p.wait() != 0:
                raise RuntimeError("frame render worker failed")
            z = np.load(out)
            for i, k in enumerate(part):
                res[k] = (z["d%d" % i], z["c%d" % i], z["T%d" % i], intrinsics(width, height))
    return [res[k] for k in indices]


if __name__ == "__main__":
    import sys
    _out, _w, _h, _fpl, _bob = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    _arrs = {}
    for _i, _k in enumerate(int(a) for a in sys.argv[6:]):
        _d, _c, _T, _ = scene_room(_k, _w, _h, _fpl, _bob)
        _arrs["d%d" % _i], _arrs["c%d" % _i], _arrs["T%d" % _i] = _d, _c, _T
    np.savez(_out, **_arrs)
