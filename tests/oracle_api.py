"""ctypes view of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY (the checker)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
if not os.path.exists(_PATH):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
olib = C.CDLL(_PATH)

from bundlefusion_amd.capi import (HashParams, DepthCameraParams, HASH_ENTRY_DTYPE, VOXEL_DTYPE,  # noqa: E402
                                   HASH_BUCKET_SIZE, VOX_PER_BLOCK, mat16)

olib.or_scene_create.restype = C.c_void_p


def set_threads(n):
    """Host threads of the oracle's image-space loops (ingest filters, cache frame, SIFT pyramid); results do not depend on it."""
    olib.or_set_threads(int(n))


olib.or_scene_hash.restype = C.c_void_p
olib.or_scene_heap.restype = C.c_void_p
olib.or_scene_voxels.restype = C.c_void_p
olib.or_scene_compactified.restype = C.c_void_p
olib.or_scene_heap_counter.restype = C.c_uint32
olib.or_scene_num_occupied.restype = C.c_uint32
olib.or_scene_num_allocated.restype = C.c_uint32
olib.or_scene_num_dropped.restype = C.c_uint32
olib.or_scene_time_update.restype = C.c_double
olib.or_hash_pos.restype = C.c_uint32


def _view(ptr, nbytes, dtype):
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class OracleScene:
    def __init__(self, params):
        self.params = params
        self._h = C.c_void_p(olib.or_scene_create(C.byref(params)))

    def close(self):
        if self._h:
            olib.or_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def reset(self):
        olib.or_scene_reset(self._h)

    @staticmethod
    def _args(depth, color):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        cptr = None
        if color is not None:
            color = np.ascontiguousarray(color, dtype=np.uint8)
            cptr = color.ctypes.data_as(C.c_void_p)
        return depth, color, depth.ctypes.data_as(C.c_void_p), cptr

    def integrate(self, T, depth, color, cam, threads=1):
        depth, color, dp, cp = self._args(depth, color)
        olib.or_scene_integrate(self._h, mat16(T), dp, cp, C.byref(cam), C.c_int(threads))

    def deintegrate(self, T, depth, color, cam, threads=1):
        depth, color, dp, cp = self._args(depth, color)
        olib.or_scene_deintegrate(self._h, mat16(T), dp, cp, C.byref(cam), C.c_int(threads))

    def compactify(self, T, cam):
        olib.or_scene_compactify(self._h, mat16(T), C.byref(cam))

    def garbage_collect(self):
        olib.or_scene_garbage_collect(self._h)

    def time_update(self, depth, color, threads, deint=False):
        depth, color, dp, cp = self._args(depth, color)
        return olib.or_scene_time_update(self._h, dp, cp, C.c_int(threads), C.c_int(int(deint)))

    def hash(self):
        n = self.params.m_hashNumBuckets * HASH_BUCKET_SIZE
        return _view(olib.or_scene_hash(self._h), n * 32, HASH_ENTRY_DTYPE)

    def heap(self):
        return _view(olib.or_scene_heap(self._h), self.params.m_numSDFBlocks * 4, "<u4")

    def heap_counter(self):
        return olib.or_scene_heap_counter(self._h)

    def voxels(self):
        return _view(olib.or_scene_voxels(self._h), self.params.m_numSDFBlocks * VOX_PER_BLOCK * 12, VOXEL_DTYPE)

    def compactified(self):
        n = olib.or_scene_num_occupied(self._h)
        if n == 0:
            return np.zeros(0, dtype=HASH_ENTRY_DTYPE)
        return _view(olib.or_scene_compactified(self._h), n * 32, HASH_ENTRY_DTYPE)

    def num_occupied(self):
        return olib.or_scene_num_occupied(self._h)

    def num_allocated(self):
        return olib.or_scene_num_allocated(self._h)

    def num_dropped(self):
        return olib.or_scene_num_dropped(self._h)

    def hash_params(self):
        p = HashParams()
        olib.or_scene_params(self._h, C.byref(p))
        return p


def hash_pos(num_buckets, x, y, z):
    return olib.or_hash_pos(C.c_uint32(num_buckets), C.c_int(x), C.c_int(y), C.c_int(z))


def world_to_block(voxel_size, w):
    w = (C.c_float * 3)(*[float(v) for v in w])
    out = (C.c_int * 6)()
    olib.or_world_to_block(C.c_float(voxel_size), w, out)
    return list(out[:3]), list(out[3:])


def mat4_inverse(m):
    out = (C.c_float * 16)()
    olib.or_mat4_inverse(mat16(m), out)
    return np.array(out[:], dtype=np.float32).reshape(4, 4)


# --------------------------------------------------------------------------- image ops / cache / solver oracle
def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def cache_store_frame(depth, color, W, H, input_intrinsics, sigma_intensity=2.5, sigma_d=1.0, sigma_r=0.05):
    """CUDACache::storeFrame on the CPU -> dict of the six cached arrays."""
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    color = np.ascontiguousarray(color, dtype=np.uint8)
    dh, dw = depth.shape
    ch, cw = color.shape[:2]
    inv = np.ascontiguousarray(mat4_inverse(np.asarray(input_intrinsics, dtype=np.float32)).reshape(16))
    out = dict(depth=np.full((H, W), np.nan, np.float32), campos=np.full((H, W, 4), np.nan, np.float32),
               intensity=np.full((H, W), np.nan, np.float32), derivs=np.full((H, W, 2), np.nan, np.float32),
               normals_u=np.zeros((H, W, 4), np.uint8), normals=np.full((H, W, 4), np.nan, np.float32))
    olib.or_cache_store_frame(_fp(depth), dw, dh, _fp(color), cw, ch, W, H, _fp(inv), C.c_float(sigma_intensity), C.c_float(sigma_d),
                              C.c_float(sigma_r), _fp(out["depth"]), _fp(out["campos"]), _fp(out["intensity"]), _fp(out["derivs"]),
                              _fp(out["normals_u"]), _fp(out["normals"]))
    return out


def erode_depth(depth, structure_size=3, d_thresh=0.05, frac_req=0.3):
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    out = np.empty_like(depth)
    h, w = depth.shape
    olib.or_erode_depth(_fp(out), _fp(depth), structure_size, w, h, C.c_float(d_thresh), C.c_float(frac_req))
    return out


def gauss_filter_depth(depth, sigma_d, sigma_r):
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    out = np.empty_like(depth)
    h, w = depth.shape
    olib.or_gauss_filter_depth(_fp(out), _fp(depth), C.c_float(sigma_d), C.c_float(sigma_r), w, h)
    return out


def gauss_filter_intensity(img, sigma_d):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.full_like(img, np.nan)
    h, w = img.shape
    olib.or_gauss_filter_intensity(_fp(out), _fp(img), C.c_float(sigma_d), w, h)
    return out


def resample_float(img, ow, oh):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.full((oh, ow), np.nan, np.float32)
    olib.or_resample_float(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


def resample_uchar4(img, ow, oh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((oh, ow, 4), np.uint8)
    olib.or_resample_uchar4(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


def resample_to_intensity(img, ow, oh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.full((oh, ow), np.nan, np.float32)
    olib.or_resample_to_intensity(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


class _CacheFrameHost(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("campos4", C.c_void_p), ("intensity", C.c_void_p), ("derivs2", C.c_void_p),
                ("normalsU4", C.c_void_p), ("normals4", C.c_void_p)]


class _SolverArgs(C.Structure):
    _fields_ = [("corr", C.c_void_p), ("numCorr", C.c_uint32), ("validImages", C.c_void_p), ("numImages", C.c_uint32),
                ("maxCorrPerImage", C.c_uint32), ("nNonLin", C.c_uint32), ("nLin", C.c_uint32), ("cacheFrames", C.c_void_p),
                ("W", C.c_uint32), ("H", C.c_uint32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("weightsSparse", C.c_void_p), ("weightsDenseDepth", C.c_void_p), ("weightsDenseColor", C.c_void_p),
                ("usePairwise", C.c_int), ("denseDistThresh", C.c_float), ("denseNormalThresh", C.c_float),
                ("denseColorThresh", C.c_float), ("denseColorGradientMin", C.c_float), ("denseDepthMin", C.c_float),
                ("denseDepthMax", C.c_float), ("denseOverlapCheckSubsampleFactor", C.c_uint32), ("rot3", C.c_void_p),
                ("trans3", C.c_void_p), ("convergence", C.c_void_p), ("pcgIterations", C.c_void_p), ("gnIterations", C.c_void_p),
                ("maxResidual", C.c_void_p), ("maxResidualIndex", C.c_void_p), ("denseJtJ", C.c_void_p), ("denseJtr", C.c_void_p),
                ("numDensePairs", C.c_void_p)]


def solver_solve(corr, valid, n_images, n_nonlin, n_lin, weights_sparse, weights_dense_depth, weights_dense_color, rot, trans,
                 cache_frames=None, cache_geom=None, cfg=None, use_pairwise=True, max_corr_per_image=4000, dump_dense=False):
    """CUDASolverBundling::solve on the CPU.  corr: ENTRYJ array (modified in place when a row overflows); rot/trans float32 [N,3]
    (updated in place).  cache_frames: list of dicts from cache_store_frame.  Returns a dict of diagnostics."""
    from bundlefusion_amd.capi import default_solver_config
    cfg = cfg or default_solver_config()
    a = _SolverArgs()
    keep = []
    a.corr = corr.ctypes.data if len(corr) else None
    a.numCorr = len(corr)
    valid = np.ascontiguousarray(valid, dtype=np.int32)
    a.validImages = valid.ctypes.data
    a.numImages = n_images
    a.maxCorrPerImage = max_corr_per_image
    a.nNonLin, a.nLin = n_nonlin, n_lin
    if cache_frames is not None:
        arr = (_CacheFrameHost * n_images)()
        for i in range(n_images):
            f = {k: np.ascontiguousarray(v) for k, v in cache_frames[i].items()}
            keep.append(f)
            arr[i].depth, arr[i].campos4, arr[i].intensity = f["depth"].ctypes.data, f["campos"].ctypes.data, f["intensity"].ctypes.data
            arr[i].derivs2, arr[i].normalsU4, arr[i].normals4 = f["derivs"].ctypes.data, f["normals_u"].ctypes.data, f["normals"].ctypes.data
        keep.append(arr)
        a.cacheFrames = C.addressof(arr)
        a.W, a.H, k = cache_geom
        a.fx, a.fy, a.cx, a.cy = k
    ws = np.asarray(weights_sparse, np.float32); wd = np.asarray(weights_dense_depth, np.float32); wc = np.asarray(weights_dense_color, np.float32)
    a.weightsSparse, a.weightsDenseDepth, a.weightsDenseColor = ws.ctypes.data, wd.ctypes.data, wc.ctypes.data
    a.usePairwise = int(use_pairwise)
    for f in ("denseDistThresh", "denseNormalThresh", "denseColorThresh", "denseColorGradientMin", "denseDepthMin", "denseDepthMax",
              "denseOverlapCheckSubsampleFactor"):
        setattr(a, f, getattr(cfg, f))
    a.rot3, a.trans3 = rot.ctypes.data, trans.ctypes.data
    conv = np.full(n_nonlin + 1, -1.0, np.float32)
    pcg = np.zeros(n_nonlin, np.int32)
    gn = C.c_int(0); mx = C.c_float(0); mi = C.c_int(0); npairs = C.c_int(0)
    a.convergence, a.pcgIterations = conv.ctypes.data, pcg.ctypes.data
    a.gnIterations, a.maxResidual, a.maxResidualIndex = C.addressof(gn), C.addressof(mx), C.addressof(mi)
    a.numDensePairs = C.addressof(npairs)
    JtJ = Jtr = None
    if dump_dense:
        JtJ = np.zeros((6 * n_images, 6 * n_images), np.float32); Jtr = np.zeros(6 * n_images, np.float32)
        a.denseJtJ, a.denseJtr = JtJ.ctypes.data, Jtr.ctypes.data
    olib.or_solver_solve(C.byref(a))
    return dict(convergence=conv, pcg_iterations=pcg[:gn.value].tolist(), gn_iterations=gn.value, max_residual=mx.value,
                max_residual_index=mi.value, JtJ=JtJ, Jtr=Jtr, num_dense_pairs=npairs.value)


def solver_use_verification(corr, rot, trans, n_images, dist_thresh=0.02, percent_thresh=0.05):
    return bool(olib.or_solver_use_verification(C.c_void_p(corr.ctypes.data), len(corr), _fp(rot), _fp(trans), n_images,
                                                C.c_float(dist_thresh), C.c_float(percent_thresh)))


def matrices_to_poses(T, valid=None):
    T = np.ascontiguousarray(T, np.float32)
    n = T.shape[0]
    valid = np.ones(n, np.int32) if valid is None else np.ascontiguousarray(valid, np.int32)
    rot = np.zeros((n, 3), np.float32); trans = np.zeros((n, 3), np.float32)
    olib.or_matrices_to_poses(_fp(T), n, _fp(rot), _fp(trans), _fp(valid))
    return rot, trans


def poses_to_matrices(rot, trans, valid=None):
    n = rot.shape[0]
    valid = np.ones(n, np.int32) if valid is None else np.ascontiguousarray(valid, np.int32)
    T = np.zeros((n, 4, 4), np.float32)
    olib.or_poses_to_matrices(_fp(np.ascontiguousarray(rot, np.float32)), _fp(np.ascontiguousarray(trans, np.float32)), n, _fp(T), _fp(valid))
    return T


# --------------------------------------------------------------------------- SIFT oracle
def sift_run(intensity, depth, depth_min=0.1, depth_max=4.0, min_key_scale=3.0, feature_count_threshold=150, max_features=1024):
    intensity = np.ascontiguousarray(intensity, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    H, W = intensity.shape
    dH, dW = depth.shape
    keys = np.zeros((max_features, 4), np.float32)
    descs = np.zeros((max_features, 128), np.uint8)
    levels = np.zeros(12, np.int32)
    n = olib.or_sift_run(_fp(intensity), _fp(depth), W, H, dW, dH, C.c_float(depth_min), C.c_float(depth_max), C.c_float(min_key_scale),
                         feature_count_threshold, max_features, _fp(keys), _fp(descs), _fp(levels))
    if n < 0:
        return n, None, None, levels
    return n, keys[:n].copy(), descs[:n].copy(), levels


def sift_pyramid_level(intensity, octave, index):
    intensity = np.ascontiguousarray(intensity, np.float32)
    H, W = intensity.shape
    out = np.zeros((H >> octave, W >> octave), np.float32)
    olib.or_sift_pyramid_level(_fp(intensity), W, H, octave, index, _fp(out))
    return out


# --------------------------------------------------------------------------- match / filter oracle
def sift_match(d1, d2, distmax=0.7, ratiomax=0.8, off1=0, off2=0, sort=True):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    idx = np.zeros((128, 2), np.uint32); dist = np.zeros(128, np.float32)
    olib.or_sift_match.restype = C.c_int
    n = olib.or_sift_match(_fp(d1), len(d1), _fp(d2), len(d2), C.c_float(distmax), C.c_float(ratiomax), off1, off2, _fp(idx), _fp(dist), int(sort))
    m = min(n, 128)
    return n, idx[:m].copy(), dist[:m].copy()


def filter_matches(keys, idx, dist, n_raw, Kinv, min_matches=5, max_res2=0.0004):
    keys = np.ascontiguousarray(keys, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32).copy(); dist = np.ascontiguousarray(dist, np.float32).copy()
    T = np.zeros((4, 4), np.float32)
    Kinv = np.ascontiguousarray(Kinv, np.float32)
    olib.or_filter_keypoint_matches.restype = C.c_int
    n = olib.or_filter_keypoint_matches(_fp(keys), _fp(idx), _fp(dist), int(n_raw), _fp(Kinv), int(min_matches), C.c_float(max_res2), _fp(T))
    return n, idx[:n].copy(), dist[:n].copy(), T


def filter_surface_area(keys, idx, Kinv, area_thresh=0.032):
    keys = np.ascontiguousarray(keys, np.float32); idx = np.ascontiguousarray(idx, np.uint32)
    areas = np.zeros(2, np.float32)
    olib.or_filter_surface_area.restype = C.c_int
    ok = olib.or_filter_surface_area(_fp(keys), _fp(idx), len(idx), _fp(np.ascontiguousarray(Kinv, np.float32)), C.c_float(area_thresh), _fp(areas))
    return bool(ok), areas


def dense_verify(fin, fmo, W, H, K, T, dist_thresh=0.15, normal_thresh=0.97, err_thresh=0.075, corr_thresh=0.02, dmin=0.1, dmax=3.0):
    """fin/fmo: cache-frame dicts (depth, campos, normals) at W x H."""
    a = [np.ascontiguousarray(fin[k], np.float32) for k in ("depth", "campos", "normals")]
    b = [np.ascontiguousarray(fmo[k], np.float32) for k in ("depth", "campos", "normals")]
    err = C.c_float(0); corr = C.c_float(0)
    olib.or_dense_verify.restype = C.c_int
    ok = olib.or_dense_verify(*[_fp(x) for x in a], *[_fp(x) for x in b], W, H, _fp(np.ascontiguousarray(K, np.float32)),
                              _fp(np.ascontiguousarray(T, np.float32)), C.c_float(dist_thresh), C.c_float(normal_thresh), C.c_float(err_thresh),
                              C.c_float(corr_thresh), C.c_float(dmin), C.c_float(dmax), C.byref(err), C.byref(corr))
    return bool(ok), err.value, corr.value


def svd3(A):
    A = np.ascontiguousarray(A, np.float32)
    U = np.zeros((3, 3), np.float32); S = np.zeros((3, 3), np.float32); V = np.zeros((3, 3), np.float32)
    olib.or_svd3(_fp(A), _fp(U), _fp(S), _fp(V))
    return U, S, V


def kabsch(src, tgt):
    src = np.ascontiguousarray(src, np.float32); tgt = np.ascontiguousarray(tgt, np.float32)
    T = np.zeros((4, 4), np.float32); ev = np.zeros(3, np.float32)
    olib.or_kabsch(_fp(src), _fp(tgt), len(src), _fp(T), _fp(ev))
    return T, ev


def inverse44(T):
    T = np.ascontiguousarray(T, np.float32); out = np.zeros((4, 4), np.float32)
    olib.or_inverse44(_fp(T), _fp(out))
    return out


def mul44(A, B):
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32); out = np.zeros((4, 4), np.float32)
    olib.or_mul44(_fp(A), _fp(B), _fp(out))
    return out


def make_entry(keys, ix, iy, img_i, img_j, Kinv):
    from bundlefusion_amd.capi import ENTRYJ_DTYPE
    e = np.zeros(1, ENTRYJ_DTYPE)
    olib.or_make_entry(_fp(np.ascontiguousarray(keys, np.float32)), int(ix), int(iy), int(img_i), int(img_j),
                       _fp(np.ascontiguousarray(Kinv, np.float32)), _fp(e))
    return e[0]


# --------------------------------------------------------------------------- marching cubes oracle
def mc_extract(scene, thresh, thresh2, edge_table, tri_table, max_triangles=2000000, box=None):
    """or_mc_extract on an OracleScene with the given case tables -> (triangles [n,3,6], number found)"""
    from bundlefusion_amd.capi import HashParams
    olib.or_mc_extract.restype = C.c_uint32
    out = np.zeros((max_triangles, 3, 6), np.float32)
    e = np.ascontiguousarray(edge_table, np.uint16); t = np.ascontiguousarray(tri_table, np.int8).reshape(-1)
    mn = np.ascontiguousarray(box[0], np.float32) if box else np.zeros(3, np.float32)
    mx = np.ascontiguousarray(box[1], np.float32) if box else np.zeros(3, np.float32)
    hp = scene.hash_params()
    n = olib.or_mc_extract(C.c_void_p(olib.or_scene_hash(scene._h)), C.c_void_p(olib.or_scene_voxels(scene._h)), C.byref(hp), C.c_float(thresh), C.c_float(thresh2),
                           int(box is not None), _fp(mn), _fp(mx), _fp(e), _fp(t), _fp(out), max_triangles)
    return out[:min(n, max_triangles)].copy(), n


# --------------------------------------------------------------------------- ray cast oracle
def rc_splat(scene, cam, params):
    """or_rc_splat over the scene's frustum list -> (ray_min, ray_max) float32 (H, W), -inf where no block projects"""
    W, H = params.m_width, params.m_height
    mn = np.zeros((H, W), np.float32); mx = np.zeros((H, W), np.float32)
    hp = scene.hash_params()
    olib.or_rc_splat(C.c_void_p(olib.or_scene_compactified(scene._h)), C.c_uint32(scene.num_occupied()), C.byref(hp), C.byref(cam), C.byref(params), _fp(mn), _fp(mx))
    return mn, mx


def rc_render(scene, params, ray_min, ray_max):
    """or_rc_render from given interval images -> dict depth (H,W), depth4 / normals / colors (H,W,4)"""
    W, H = params.m_width, params.m_height
    out = dict(depth=np.zeros((H, W), np.float32), depth4=np.zeros((H, W, 4), np.float32), normals=np.zeros((H, W, 4), np.float32), colors=np.zeros((H, W, 4), np.float32))
    hp = scene.hash_params()
    mn = np.ascontiguousarray(ray_min, np.float32); mx = np.ascontiguousarray(ray_max, np.float32)
    olib.or_rc_render(C.c_void_p(olib.or_scene_hash(scene._h)), C.c_void_p(olib.or_scene_voxels(scene._h)), C.byref(hp), C.byref(params), _fp(mn), _fp(mx),
                      _fp(out["depth"]), _fp(out["depth4"]), _fp(out["normals"]), _fp(out["colors"]))
    return out


def sift_detect(intensity, depth, depth_min=0.1, depth_max=4.0, capacity=4096):
    """raw key lists after DetectKeypoints: list of 12 arrays (n_i, 2) of (col, row) per (octave, DoG level) slot"""
    intensity = np.ascontiguousarray(intensity, np.float32); depth = np.ascontiguousarray(depth, np.float32)
    H, W = intensity.shape; dH, dW = depth.shape
    counts = np.zeros(12, np.int32); xy = np.zeros((12, capacity, 2), np.int32)
    olib.or_sift_detect(_fp(intensity), _fp(depth), W, H, dW, dH, C.c_float(depth_min), C.c_float(depth_max), _fp(counts), _fp(xy), capacity)
    return [xy[i, :min(counts[i], capacity)].copy() for i in range(12)]
