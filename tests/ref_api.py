"""ctypes view of oracle/_ref/libbfref.so — TEST INFRASTRUCTURE ONLY: the REFERENCE's own device code for the pinned stages,
compiled for the host by oracle/ref/Makefile (needs /root/reference at build time; the built library travels with the repository
snapshot, is git-ignored, and is never loaded by the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libbfref.so")
REFERENCE = "/root/reference/FriedLiver"


def available():
    """Build when the reference sources are present; otherwise use a prebuilt library if it travelled with the snapshot."""
    if os.path.isdir(os.path.join(REFERENCE, "Source")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref")])
    return os.path.exists(PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        for f in ("ref_scene_create", "ref_scene_hash", "ref_scene_heap", "ref_scene_voxels", "ref_scene_compactified"):
            getattr(_lib, f).restype = C.c_void_p
        for f in ("ref_scene_heap_counter", "ref_scene_num_occupied", "ref_compute_hash_pos", "ref_linearize_voxel_pos", "ref_filter_keypoint_matches"):
            getattr(_lib, f).restype = C.c_uint32
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


from bundlefusion_amd.capi import HASH_ENTRY_DTYPE, VOXEL_DTYPE, HASH_BUCKET_SIZE, VOX_PER_BLOCK  # noqa: E402


def _view(ptr, nbytes, dtype):
    return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype)


class RefScene:
    """The reference's CUDASceneRepHashSDF operators (serial block emulation of its own kernels).  host_class=False: the launch wrappers
    sequenced by oracle/ref/ref_tsdf.cpp; host_class=True: the reference's own host class (CUDASceneRepHashSDF.h compiled as it is)."""

    def __init__(self, params, host_class=False):
        self.params = params
        self._p = "ref_hscene_" if host_class else "ref_scene_"
        L = lib()
        for f in ("create", "hash", "heap", "voxels", "compactified"):
            getattr(L, self._p + f).restype = C.c_void_p
        for f in ("heap_counter", "num_occupied"):
            getattr(L, self._p + f).restype = C.c_uint32
        self._h = C.c_void_p(self._f("create")(C.byref(params)))

    def _f(self, name):
        return getattr(lib(), self._p + name)

    def __del__(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    def integrate(self, T, depth, color, cam):
        d = _f32(depth); c = np.ascontiguousarray(color, np.uint8)
        self._f("integrate")(self._h, _fp(_f32(T).reshape(16)), _fp(d), _fp(c), C.byref(cam))

    def deintegrate(self, T, depth, color, cam):
        d = _f32(depth); c = np.ascontiguousarray(color, np.uint8)
        self._f("deintegrate")(self._h, _fp(_f32(T).reshape(16)), _fp(d), _fp(c), C.byref(cam))

    def compactify(self, T, cam):
        self._f("compactify")(self._h, _fp(_f32(T).reshape(16)), C.byref(cam))

    def garbage_collect(self):
        self._f("garbage_collect")(self._h)

    def hash(self):
        return _view(self._f("hash")(self._h), self.params.m_hashNumBuckets * HASH_BUCKET_SIZE * 32, HASH_ENTRY_DTYPE)

    def heap(self):
        return _view(self._f("heap")(self._h), self.params.m_numSDFBlocks * 4, "<u4")

    def heap_counter(self):
        return self._f("heap_counter")(self._h)

    def voxels(self):
        return _view(self._f("voxels")(self._h), self.params.m_numSDFBlocks * VOX_PER_BLOCK * 12, VOXEL_DTYPE)

    def num_occupied(self):
        return self._f("num_occupied")(self._h)

    def compactified(self):
        n = self.num_occupied()
        return _view(self._f("compactified")(self._h), n * 32, HASH_ENTRY_DTYPE) if n else np.zeros(0, HASH_ENTRY_DTYPE)


def hash_params_from_global_app_state(gas):
    """CUDASceneRepHashSDF::parametersFromGlobalAppState (CUDASceneRepHashSDF.h:39-59) of the reference on the values of a capi.GlobalAppState"""
    from bundlefusion_amd.capi import HashParams
    p = HashParams()
    lib().ref_hash_params_from_global_app_state(C.c_uint32(gas.s_hashNumBuckets), C.c_uint32(gas.s_hashMaxCollisionLinkedListSize), C.c_uint32(gas.s_hashNumSDFBlocks),
                                                C.c_float(gas.s_SDFVoxelSize), C.c_float(gas.s_SDFMaxIntegrationDistance), C.c_float(gas.s_SDFTruncation),
                                                C.c_float(gas.s_SDFTruncationScale), C.c_uint32(gas.s_SDFIntegrationWeightSample),
                                                C.c_uint32(gas.s_SDFIntegrationWeightMax), C.byref(p))
    return p


def hash_pos(num_buckets, x, y, z):
    return lib().ref_compute_hash_pos(C.c_uint32(num_buckets), C.c_int(x), C.c_int(y), C.c_int(z))


def world_to_block(voxel_size, w):
    out = (C.c_int * 6)()
    lib().ref_world_to_block(C.c_float(voxel_size), _fp(_f32(w)), out)
    return list(out[:3]), list(out[3:])


def delinearize(idx):
    out = (C.c_uint32 * 3)()
    lib().ref_delinearize_voxel_index(C.c_uint32(idx), out)
    return tuple(out)


def linearize(x, y, z):
    return lib().ref_linearize_voxel_pos(C.c_int(x), C.c_int(y), C.c_int(z))


def mat4_inverse(m):
    out = np.zeros(16, np.float32)
    lib().ref_mat4_inverse(_fp(_f32(m).reshape(16)), _fp(out))
    return out.reshape(4, 4)


def exp_rotation(w):
    R = np.zeros(9, np.float32)
    lib().ref_exp_rotation(_fp(_f32(w)), _fp(R))
    return R.reshape(3, 3)


def ln_rotation(R):
    w = np.zeros(3, np.float32)
    lib().ref_ln_rotation(_fp(_f32(R).reshape(9)), _fp(w))
    return w


def matrix_to_pose(M):
    r = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    lib().ref_matrix_to_pose(_fp(_f32(M).reshape(16)), _fp(r), _fp(t))
    return r, t


def pose_to_matrix(rot, trans):
    M = np.zeros(16, np.float32)
    lib().ref_pose_to_matrix(_fp(_f32(rot)), _fp(_f32(trans)), _fp(M))
    return M.reshape(4, 4)


def lie_update(dW, dT, w, t):
    nw = np.zeros(3, np.float32); nt = np.zeros(3, np.float32)
    lib().ref_lie_update(_fp(_f32(dW)), _fp(_f32(dT)), _fp(_f32(w)), _fp(_f32(t)), _fp(nw), _fp(nt))
    return nw, nt


def lie_deriv(which, A, D, p):
    out = np.zeros(18, np.float32)
    lib().ref_lie_deriv(int(which), _fp(_f32(A).reshape(16)), _fp(_f32(D).reshape(16)), _fp(_f32(p)), _fp(out))
    return out.reshape(3, 6)


def svd3(A):
    U = np.zeros(9, np.float32); S = np.zeros(9, np.float32); V = np.zeros(9, np.float32)
    lib().ref_svd3(_fp(_f32(A).reshape(9)), _fp(U), _fp(S), _fp(V))
    return U.reshape(3, 3), S.reshape(3, 3), V.reshape(3, 3)


def eigenvalues3(A):
    ev = np.zeros(3, np.float32)
    lib().ref_eigenvalues3(_fp(_f32(A).reshape(9)), _fp(ev))
    return ev


def kabsch(src, tgt):
    src = _f32(src); tgt = _f32(tgt)
    T = np.zeros(16, np.float32); ev = np.zeros(3, np.float32)
    lib().ref_kabsch(_fp(src), _fp(tgt), len(src), _fp(T), _fp(ev))
    return T.reshape(4, 4), ev


def filter_matches(keys, idx, dist, n_raw, Kinv, min_matches=5, max_res2=0.0004):
    """filterKeyPointMatches: returns (n, idx[:n], dist[:n], T)"""
    keys = _f32(keys); idx = np.ascontiguousarray(idx, np.uint32).copy(); dist = _f32(dist).copy()
    T = np.zeros(16, np.float32)
    n = lib().ref_filter_keypoint_matches(_fp(keys), _fp(idx), _fp(dist), int(n_raw), _fp(_f32(Kinv).reshape(16)), int(min_matches), C.c_float(max_res2), _fp(T))
    return n, idx[:n], dist[:n], T.reshape(4, 4)


# ---- image kernels (CUDAImageUtil.cu) ----
def erode_depth(depth, structure_size=3, d_thresh=0.05, frac_req=0.3):
    depth = _f32(depth); out = np.full_like(depth, np.nan); h, w = depth.shape
    lib().ref_erode_depth(_fp(out), _fp(depth), structure_size, w, h, C.c_float(d_thresh), C.c_float(frac_req))
    return out


def gauss_filter_depth(depth, sigma_d, sigma_r):
    depth = _f32(depth); out = np.full_like(depth, np.nan); h, w = depth.shape
    lib().ref_gauss_filter_depth(_fp(out), _fp(depth), C.c_float(sigma_d), C.c_float(sigma_r), w, h)
    return out


def gauss_filter_intensity(img, sigma_d):
    img = _f32(img); out = np.full_like(img, np.nan); h, w = img.shape
    lib().ref_gauss_filter_intensity(_fp(out), _fp(img), C.c_float(sigma_d), w, h)
    return out


def resample_float(img, ow, oh):
    img = _f32(img); out = np.full((oh, ow), np.nan, np.float32)
    lib().ref_resample_float(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


def resample_uchar4(img, ow, oh):
    img = np.ascontiguousarray(img, np.uint8); out = np.zeros((oh, ow, 4), np.uint8)
    lib().ref_resample_uchar4(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


def resample_to_intensity(img, ow, oh):
    img = np.ascontiguousarray(img, np.uint8); out = np.full((oh, ow), np.nan, np.float32)
    lib().ref_resample_to_intensity(_fp(out), ow, oh, _fp(img), img.shape[1], img.shape[0])
    return out


def ingest(depth, wi, hi, erode=True, depth_filter=True, sigma_d=2.0, sigma_r=0.05):
    """CUDAImageManager::process, device part -> (raw after erosion, filtered, integration frame)"""
    raw = _f32(depth).copy(); h, w = raw.shape
    filt = np.full_like(raw, np.nan); integ = np.full((hi, wi), np.nan, np.float32)
    lib().ref_ingest(_fp(raw), _fp(filt), _fp(integ), w, h, wi, hi, int(erode), int(depth_filter), C.c_float(sigma_d), C.c_float(sigma_r))
    return raw, filt, integ


def cache_store_frame(depth, color, W, H, input_intrinsics_inv, sigma_intensity=2.5, sigma_d=1.0, sigma_r=0.05):
    depth = _f32(depth); color = np.ascontiguousarray(color, np.uint8)
    dh, dw = depth.shape; ch, cw = color.shape[:2]
    out = dict(depth=np.full((H, W), np.nan, np.float32), campos=np.full((H, W, 4), np.nan, np.float32),
               intensity=np.full((H, W), np.nan, np.float32), derivs=np.full((H, W, 2), np.nan, np.float32),
               normals_u=np.zeros((H, W, 4), np.uint8), normals=np.full((H, W, 4), np.nan, np.float32))
    lib().ref_cache_store_frame(_fp(depth), dw, dh, _fp(color), cw, ch, W, H, _fp(_f32(input_intrinsics_inv).reshape(16)), C.c_float(sigma_intensity),
                                C.c_float(sigma_d), C.c_float(sigma_r), _fp(out["depth"]), _fp(out["campos"]), _fp(out["intensity"]), _fp(out["derivs"]),
                                _fp(out["normals_u"]), _fp(out["normals"]))
    return out


# ---- bundling solver (Solver/SolverBundling.cu) ----
class _RefSolverParams(C.Structure):
    _fields_ = [("denseDistThresh", C.c_float), ("denseNormalThresh", C.c_float), ("denseColorThresh", C.c_float), ("denseColorGradientMin", C.c_float),
                ("denseDepthMin", C.c_float), ("denseDepthMax", C.c_float), ("denseOverlapCheckSubsampleFactor", C.c_uint32)]


class _RefCacheFrame(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("campos", C.c_void_p), ("intensity", C.c_void_p), ("derivs", C.c_void_p), ("normalsU", C.c_void_p), ("normals", C.c_void_p)]


def solver_solve(corr, valid, n_images, n_nonlin, n_lin, weights_sparse, weights_dense_depth, weights_dense_color, rot, trans,
                 cache_frames=None, cache_geom=None, cfg=None, use_pairwise=True, max_images=None, max_residuals=None):
    """solveBundlingStub of the reference (serial emulation).  corr (ENTRYJ, may be modified), rot / trans [N,3] float32 are updated in place."""
    from bundlefusion_amd.capi import default_solver_config
    cfg = cfg or default_solver_config()
    gp = _RefSolverParams(*[getattr(cfg, f) for f in ("denseDistThresh", "denseNormalThresh", "denseColorThresh", "denseColorGradientMin", "denseDepthMin",
                                                      "denseDepthMax", "denseOverlapCheckSubsampleFactor")])
    max_images = max_images or max(n_images, 2)
    max_residuals = max_residuals or max(len(corr), 1)
    valid = np.ascontiguousarray(valid, np.int32)
    ws, wd, wc = (np.asarray(w, np.float32) for w in (weights_sparse, weights_dense_depth, weights_dense_color))
    keep, frames, geom = [], None, None
    W = H = 0
    if cache_frames is not None:
        arr = (_RefCacheFrame * n_images)()
        for i in range(n_images):
            f = {k: np.ascontiguousarray(v) for k, v in cache_frames[i].items()}
            keep.append(f)
            arr[i].depth, arr[i].campos, arr[i].intensity = f["depth"].ctypes.data, f["campos"].ctypes.data, f["intensity"].ctypes.data
            arr[i].derivs, arr[i].normalsU, arr[i].normals = f["derivs"].ctypes.data, f["normals_u"].ctypes.data, f["normals"].ctypes.data
        frames = arr
        W, H, k4 = cache_geom
        geom = np.asarray(k4, np.float32)
    conv = np.full(n_nonlin + 1, -1.0, np.float32)
    mx = C.c_float(-1.0); mi = C.c_int(-1); ver = C.c_int(0)
    rows = np.zeros(n_images, np.int32)
    lib().ref_solver_solve(C.c_void_p(corr.ctypes.data if len(corr) else 0), len(corr), _fp(valid), n_images, max_images, max_residuals, n_nonlin, n_lin,
                           _fp(ws), _fp(wd), _fp(wc), len(ws), frames, W, H, _fp(geom) if geom is not None else None, int(use_pairwise), C.byref(gp),
                           _fp(rot), _fp(trans), _fp(conv), C.byref(mx) if len(corr) else None, C.byref(mi), _fp(rows), C.byref(ver) if len(corr) else None)
    return dict(convergence=conv, max_residual=mx.value, max_residual_index=mi.value, rows=rows, use_verification=bool(ver.value))


# ---- marching cubes (CUDAMarchingCubesSDF.cu, MarchingCubesSDFUtil.h, Tables.h) ----
def mc_tables():
    """(edgeTable[256] uint16, triTable[256,16] int8) of the reference's Tables.h"""
    e = np.zeros(256, np.uint16); t = np.zeros(256 * 16, np.int8)
    lib().ref_mc_tables(_fp(e), _fp(t))
    return e, t.reshape(256, 16)


def mc_extract(scene, thresh, thresh2, max_triangles=2000000, box=None):
    """extractIsoSurface on a RefScene -> triangles [n, 3, 6] (position, colour), in the (atomic) append order of the serial emulation"""
    out = np.zeros((max_triangles, 3, 6), np.float32)
    lib().ref_mc_extract.restype = C.c_uint32
    mn = _f32(box[0]) if box else None; mx = _f32(box[1]) if box else None
    n = lib().ref_mc_extract(scene._h, C.c_float(thresh), C.c_float(thresh2), int(box is not None), _fp(mn) if box else None, _fp(mx) if box else None,
                             _fp(out), max_triangles)
    return out[:min(n, max_triangles)].copy(), n


def rc_render(scene, params, ray_min, ray_max):
    """The reference's renderCS on a RefScene from given ray-interval images -> dict depth (H,W), depth4 / normals / colors (H,W,4).
    (computeNormals, which CUDARayCastSDF::render runs afterwards when gradients are off, is NOT applied: normals stay -inf then.)"""
    W, H = params.m_width, params.m_height
    out = dict(depth=np.zeros((H, W), np.float32), depth4=np.zeros((H, W, 4), np.float32), normals=np.zeros((H, W, 4), np.float32), colors=np.zeros((H, W, 4), np.float32))
    intr = _f32([params.mx, params.my, params.fx, params.fy])
    f5 = _f32([params.m_minDepth, params.m_maxDepth, params.m_rayIncrement, params.m_thresSampleDist, params.m_thresDist])
    lib().ref_rc_render(scene._h, _fp(_f32(list(params.m_viewMatrix))), _fp(_f32(list(params.m_viewMatrixInverse))), _fp(intr), C.c_uint32(W), C.c_uint32(H), _fp(f5),
                        int(params.m_useGradients), _fp(_f32(ray_min)), _fp(_f32(ray_max)), _fp(out["depth"]), _fp(out["depth4"]), _fp(out["normals"]), _fp(out["colors"]))
    return out


class RefSiftManager:
    """The reference's SIFTImageManager match-filter chain (its own kernels and launch configurations, serial block emulation):
    raw / filtered match lists per previous image, key points in one global array addressed by the match indices."""
    RAW, FILT = 128, 25

    def __init__(self, max_images, max_keys=1024):
        L = lib()
        L.ref_siftmgr_create.restype = C.c_void_p
        self._h = C.c_void_p(L.ref_siftmgr_create(max_images, max_keys))
        self.max_images, self.max_keys = max_images, max_keys
        self._keep = []

    def set_keys(self, keys):
        k = _f32(keys).reshape(-1, 4)
        lib().ref_siftmgr_set_keys(self._h, _fp(k), len(k))

    def set_raw(self, pair, n, idx, dist):
        i = np.zeros((self.RAW, 2), np.uint32); d = np.zeros(self.RAW, np.float32)
        m = min(int(n), self.RAW, len(idx)); i[:m] = idx[:m]; d[:m] = dist[:m]
        lib().ref_siftmgr_set_raw(self._h, pair, int(n), _fp(d), _fp(i))

    def raw(self, pair):
        n = C.c_int(); i = np.zeros((self.RAW, 2), np.uint32); d = np.zeros(self.RAW, np.float32)
        lib().ref_siftmgr_get_raw(self._h, pair, C.byref(n), _fp(d), _fp(i))
        return n.value, i, d

    def set_filtered(self, pair, n, idx=None, dist=None, T=None, Tinv=None):
        i = d = None
        if idx is not None:
            i = np.zeros((self.FILT, 2), np.uint32); i[:len(idx)] = idx
        if dist is not None:
            d = np.zeros(self.FILT, np.float32); d[:len(dist)] = dist
        lib().ref_siftmgr_set_filtered(self._h, pair, int(n), _fp(d) if d is not None else None, _fp(i) if i is not None else None,
                                       _fp(_f32(T).reshape(16)) if T is not None else None, _fp(_f32(Tinv).reshape(16)) if Tinv is not None else None)

    def filtered(self, pair):
        n = C.c_int(); i = np.zeros((self.FILT, 2), np.uint32); d = np.zeros(self.FILT, np.float32)
        T = np.zeros((4, 4), np.float32); Ti = np.zeros((4, 4), np.float32)
        lib().ref_siftmgr_get_filtered(self._h, pair, C.byref(n), _fp(d), _fp(i), _fp(T), _fp(Ti))
        return n.value, i, d, T, Ti

    def set_cached_frame(self, i, frame):
        """frame: dict of the six cached arrays (tests/oracle_api.cache_store_frame); the arrays are kept alive here"""
        a = [np.ascontiguousarray(frame[k], np.float32) for k in ("depth", "campos", "intensity", "derivs")]
        nu = np.ascontiguousarray(frame["normals_u"], np.uint8); nf = np.ascontiguousarray(frame["normals"], np.float32)
        self._keep.append((a, nu, nf))
        lib().ref_siftmgr_set_cached_frame(self._h, i, _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), _fp(nu), _fp(nf))

    def sort(self, cur, start, num):
        lib().ref_siftmgr_sort(self._h, cur, start, num)

    def filter_keypoint_matches(self, cur, start, num, Kinv, min_matches=5, max_res2=0.0004):
        lib().ref_siftmgr_filter_keypoint_matches(self._h, cur, start, num, _fp(_f32(Kinv).reshape(16)), int(min_matches), C.c_float(max_res2))

    def filter_surface_area(self, cur, start, num, Kinv, area_thresh=0.032):
        lib().ref_siftmgr_filter_surface_area(self._h, cur, start, num, _fp(_f32(Kinv).reshape(16)), C.c_float(area_thresh))

    def filter_dense_verify(self, cur, start, num, W, H, K, dist_thresh=0.15, normal_thresh=0.97, color_thresh=0.1, err_thresh=0.075, corr_thresh=0.02,
                            dmin=0.1, dmax=3.0):
        lib().ref_siftmgr_filter_dense_verify(self._h, cur, start, num, W, H, _fp(_f32(K).reshape(16)), C.c_float(dist_thresh), C.c_float(normal_thresh),
                                              C.c_float(color_thresh), C.c_float(err_thresh), C.c_float(corr_thresh), C.c_float(dmin), C.c_float(dmax))

    def add_curr_to_residuals(self, cur, start, num, Kinv):
        from bundlefusion_amd.capi import ENTRYJ_DTYPE
        L = lib()
        L.ref_siftmgr_add_curr_to_residuals.restype = C.c_uint32
        n = L.ref_siftmgr_add_curr_to_residuals(self._h, cur, start, num, _fp(_f32(Kinv).reshape(16)))
        assert L.ref_siftmgr_sizeof_entryj() == ENTRYJ_DTYPE.itemsize
        e = np.zeros(n, ENTRYJ_DTYPE); k = np.zeros((n, 2), np.uint32)
        if n:
            L.ref_siftmgr_get_residuals(self._h, _fp(e), _fp(k), n)
        return e, k

    def verify_trajectory(self, traj, valid, W, H, K, dist_thresh=0.15, normal_thresh=0.97, color_thresh=0.1, err_thresh=0.05, corr_thresh=0.02, dmin=0.1, dmax=3.0):
        t = _f32(traj).reshape(-1, 16); v = np.ascontiguousarray(valid, np.int32)
        return int(lib().ref_siftmgr_verify_trajectory(self._h, len(t), _fp(t), _fp(v), W, H, _fp(_f32(K).reshape(16)), C.c_float(dist_thresh), C.c_float(normal_thresh),
                                                       C.c_float(color_thresh), C.c_float(err_thresh), C.c_float(corr_thresh), C.c_float(dmin), C.c_float(dmax)))


class RefSift:
    """The reference's SiftGPU fork (detector + descriptor + matcher) on the host emulator."""

    def __init__(self, w, h, K, feature_count_threshold=150, depth_min=0.1, depth_max=4.0, min_key_scale=3.0, max_keys=1024):
        L = lib()
        L.ref_sift_create.restype = C.c_void_p
        Kf = _f32(K).reshape(16)
        Ki = _f32(np.linalg.inv(np.asarray(K, np.float64))).reshape(16)
        self.max_keys = max_keys
        self._h = C.c_void_p(L.ref_sift_create(w, h, w, h, _fp(Kf), _fp(Ki), feature_count_threshold, C.c_float(depth_min), C.c_float(depth_max),
                                               C.c_float(min_key_scale), max_keys))

    def run(self, intensity, depth):
        keys = np.zeros((self.max_keys, 4), np.float32); descs = np.zeros((self.max_keys, 128), np.uint8)
        n = lib().ref_sift_run(self._h, _fp(_f32(intensity)), _fp(_f32(depth)), _fp(keys), _fp(descs))
        m = max(0, min(n, self.max_keys))
        return n, keys[:m].copy(), descs[:m].copy()

    def match(self, d1, d2, off1=0, off2=0, distmax=0.7, ratiomax=0.8):
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        idx = np.zeros((128, 2), np.uint32); dist = np.zeros(128, np.float32)
        n = lib().ref_sift_match(self._h, _fp(d1), len(d1), _fp(d2), len(d2), off1, off2, C.c_float(distmax), C.c_float(ratiomax), _fp(idx), _fp(dist))
        m = max(0, min(n, 128))
        return n, idx[:m].copy(), dist[:m].copy()


def sift_stages(rs, intensity, depth):
    """RefSift `rs` run stage by stage on one frame: dict with `levels` {(octave, index): image} of the Gaussian pyramid, `raw` (12 arrays of
    (col, row) after DetectKeypoints), `counts` (12 final per-slot counts after orientation + limits), `final` (12 arrays (n, 4): x, y,
    scale, orientation)."""
    L = lib()
    I = _f32(intensity); d = _f32(depth)
    L.ref_sift_detect(rs._h, _fp(I), _fp(d))
    out = {"levels": {}, "raw": [], "final": []}
    H, W = I.shape
    buf = np.zeros(W * H * 4, np.float32)
    for o in range(3):
        for a in range(6):
            w, h, ch = C.c_int(), C.c_int(), C.c_int()
            r = L.ref_sift_level(rs._h, o, a, 0, _fp(buf), len(buf), C.byref(w), C.byref(h), C.byref(ch))
            if r > 0:
                out["levels"][(o, a)] = buf[:r].reshape(h.value, w.value).copy()
    raw = np.zeros((4096, 4), np.int32)
    for s in range(12):
        k = L.ref_sift_raw_keys(rs._h, s, _fp(raw), 4096)
        out["raw"].append(raw[:k, :2].copy())
    L.ref_sift_orient(rs._h)
    cnt = np.zeros(12, np.int32)
    L.ref_sift_level_counts(rs._h, _fp(cnt), 12)
    out["counts"] = cnt
    fin = np.zeros((4096, 4), np.float32)
    for s in range(12):
        k = L.ref_sift_final_keys(rs._h, s, _fp(fin), 4096)
        out["final"].append(fin[:k].copy())
    return out


def siftmgr_fuse(keys_per_image, descs_per_image, corr, corr_keys_packed, transforms, K, max_keys_global=1024):
    """SIFTImageManager::fuseToGlobal of the reference on a chunk given by its images (key points / descriptors per image), its correspondences
    (EntryJ rows + PACKED key index pairs) and its trajectory -> (keys (n, 4), descs (n, 128)) of the fused key frame."""
    L = lib()
    L.ref_siftmgr_create.restype = C.c_void_p
    h = C.c_void_p(L.ref_siftmgr_create(len(keys_per_image) + 1, 1024))
    for k, d in zip(keys_per_image, descs_per_image):
        L.ref_siftmgr_add_image(h, _fp(_f32(k).reshape(-1, 4)), _fp(np.ascontiguousarray(d, np.uint8)), len(k))
    e = np.ascontiguousarray(corr); ck = np.ascontiguousarray(corr_keys_packed, np.uint32)
    L.ref_siftmgr_set_residuals(h, _fp(e), _fp(ck), len(e))
    T = _f32(transforms).reshape(-1, 16)
    Kf = _f32(K).reshape(16); Ki = _f32(np.linalg.inv(np.asarray(K, np.float64))).reshape(16)
    ok = np.zeros((max_keys_global, 4), np.float32); od = np.zeros((max_keys_global, 128), np.uint8)
    L.ref_siftmgr_fuse_to_global.restype = C.c_uint32
    n = L.ref_siftmgr_fuse_to_global(h, _fp(Kf), _fp(Ki), _fp(T), max_keys_global, _fp(ok), _fp(od))
    return ok[:n].copy(), od[:n].copy()


def siftmgr_filter_frames(num_filt, valid, cur, start, num):
    """SIFTImageManager::filterFrames -> (lastMatchedFrame or -1, valid flag of the current frame)"""
    L = lib()
    L.ref_siftmgr_create.restype = C.c_void_p
    h = C.c_void_p(L.ref_siftmgr_create(max(num, 2) + 1, 64))
    for p, n in enumerate(num_filt):
        L.ref_siftmgr_set_filtered(h, p, int(n), None, None, None, None)
    v = np.ascontiguousarray(valid, np.int32); out = C.c_int()
    L.ref_siftmgr_filter_frames.restype = C.c_uint32
    last = L.ref_siftmgr_filter_frames(h, cur, start, num, _fp(v), C.byref(out))
    return (-1 if last == 0xFFFFFFFF else int(last)), out.value


class RefTrajectoryManager:
    """The reference's TrajectoryManager (TrajectoryManager.cpp compiled as it is, with PoseHelper.h's se(3) logarithm)."""

    def __init__(self, n_max, top_n, min_dist):
        L = lib()
        L.ref_tm_create.restype = C.c_void_p
        L.ref_tm_num_active.restype = C.c_uint32
        self._h = C.c_void_p(L.ref_tm_create(C.c_uint32(n_max), C.c_uint32(top_n), C.c_float(min_dist)))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().ref_tm_destroy(self._h)

    def add(self, typ, T, idx):
        lib().ref_tm_add_frame(self._h, C.c_int(typ), _fp(_f32(T)), C.c_uint32(idx))

    def update(self, traj):
        a = _f32(traj)
        lib().ref_tm_update_optimized_transform(self._h, _fp(a), C.c_uint32(len(a)))

    def generate(self):
        lib().ref_tm_generate_update_lists(self._h)

    def confirm(self, idx):
        lib().ref_tm_confirm_integration(self._h, C.c_uint32(idx))

    def active(self):
        return int(lib().ref_tm_num_active(self._h))

    def _top(self, fn, two):
        a, b = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
        idx = C.c_uint32()
        found = fn(self._h, _fp(a), _fp(b), C.byref(idx)) if two else fn(self._h, _fp(a), C.byref(idx))
        return bool(found), idx.value, a, b

    def top_de(self):
        return self._top(lib().ref_tm_top_deintegrate, False)

    def top_in(self):
        return self._top(lib().ref_tm_top_integrate, False)

    def top_re(self):
        return self._top(lib().ref_tm_top_reintegrate, True)

    def frame(self, idx):
        t, d = C.c_int(), C.c_float()
        a, b = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
        lib().ref_tm_frame(self._h, C.c_uint32(idx), C.byref(t), _fp(a), _fp(b), C.byref(d))
        return t.value, a, d.value


def host_matrix_to_pose(T):
    """PoseHelper::MatrixToPose (PoseHelper.h:332-363, USE_LIE_SPACE): 6 floats, the translation part first, then the rotation vector."""
    p = np.zeros(6, np.float32)
    lib().ref_pose_matrix_to_pose(_fp(_f32(T)), _fp(p))
    return p


def host_pose_to_matrix(p):
    T = np.zeros((4, 4), np.float32)
    lib().ref_pose_pose_to_matrix(_fp(_f32(p)), _fp(T))
    return T


class _RefBundlingParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("numLocalNonLinIterations", "numGlobalNonLinIterations", "submapSize", "denseOverlapCheckSubsampleFactor")] + \
               [(n, C.c_float) for n in ("optMaxResThresh", "denseDistThresh", "denseNormalThresh", "denseColorThresh", "denseColorGradientMin", "denseDepthMin",
                                         "denseDepthMax")] + \
               [(n, C.c_int) for n in ("useComprehensiveFrameInvalidation", "useLocalDense", "recordSolverConvergence")]


class RefSBA:
    """The reference's SBA + CUDASolverBundling host classes (SBA.cpp / CUDASolverBundling.cpp compiled as they are) over a RefSiftManager
    that holds the images, the global correspondences and, for the dense terms, the cached frames."""

    def __init__(self, max_images, max_residuals, gbs):
        L = lib()
        L.ref_sba_create.restype = C.c_void_p
        p = _RefBundlingParams(gbs.s_numLocalNonLinIterations, gbs.s_numGlobalNonLinIterations, gbs.s_submapSize, gbs.s_denseOverlapCheckSubsampleFactor,
                               gbs.s_optMaxResThresh, gbs.s_denseDistThresh, gbs.s_denseNormalThresh, gbs.s_denseColorThresh, gbs.s_denseColorGradientMin,
                               gbs.s_denseDepthMin, gbs.s_denseDepthMax, int(gbs.s_useComprehensiveFrameInvalidation), int(gbs.s_useLocalDense), 1)
        self._h = C.c_void_p(L.ref_sba_create(max_images, max_residuals, C.byref(p)))
        self.n_its = max(gbs.s_numLocalNonLinIterations, gbs.s_numGlobalNonLinIterations)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().ref_sba_destroy(self._h)

    def weights(self, which):
        ws, wd, wc = (np.zeros(self.n_its, np.float32) for _ in range(3))
        n = lib().ref_sba_get_weights(self._h, int(which), _fp(ws), _fp(wd), _fp(wc))
        return ws[:n], wd[:n], wc[:n]

    def set_global_weights(self, ws, wd, wc, use_global_dense):
        a, b, c = _f32(ws), _f32(wd), _f32(wc)
        lib().ref_sba_set_global_weights(self._h, _fp(a), _fp(b), _fp(c), len(a), int(use_global_dense))

    def align(self, mgr, valid, current_frame, transforms, max_iters, pcg_its, use_verify, is_local, is_start=True, is_end=True, revalidate_idx=0xFFFFFFFF,
              cache_geom=None):
        """-> dict(removed, valid, transforms, max_residual, use_verification, convergence); cache_geom = (W, H, K 4x4 of the cached frames) or None"""
        v = np.ascontiguousarray(valid, np.int32).copy()
        T = _f32(transforms).reshape(-1, 16).copy()
        mr, uv = C.c_float(), C.c_int()
        conv = np.full(max_iters + 1, -1.0, np.float32)
        W, H, K = cache_geom if cache_geom else (0, 0, np.eye(4))
        Kf = _f32(K).reshape(16)
        removed = lib().ref_sba_align(self._h, mgr._h, _fp(v), C.c_uint32(current_frame), C.c_uint32(W), C.c_uint32(H), _fp(Kf), _fp(T), C.c_uint32(max_iters),
                                      C.c_uint32(pcg_its), int(use_verify), int(is_local), int(is_start), int(is_end), C.c_uint32(revalidate_idx),
                                      C.byref(mr), C.byref(uv), _fp(conv))
        return dict(removed=bool(removed), valid=v, transforms=T.reshape(-1, 4, 4), max_residual=mr.value, use_verification=bool(uv.value), convergence=conv)


def siftmgr_with_images(n_images, corr, max_keys=64):
    """A RefSiftManager holding n_images (one dummy key each) and the global correspondences `corr` (EntryJ rows)."""
    m = RefSiftManager(n_images + 1, max_keys)
    L = lib()
    for _ in range(n_images):
        L.ref_siftmgr_add_image(m._h, _fp(np.zeros((1, 4), np.float32)), _fp(np.zeros((1, 128), np.uint8)), 1)
    e = np.ascontiguousarray(corr)
    L.ref_siftmgr_set_residuals(m._h, _fp(e), _fp(np.zeros((max(len(e), 1), 2), np.uint32)), len(e))
    return m


def siftmgr_residuals(mgr, n):
    from bundlefusion_amd.capi import ENTRYJ_DTYPE
    e = np.zeros(n, ENTRYJ_DTYPE); k = np.zeros((max(n, 1), 2), np.uint32)
    if n:
        lib().ref_siftmgr_get_residuals(mgr._h, _fp(e), _fp(k), n)
    return e


class _RefBundlingState(C.Structure):
    _U = ("maxNumImages", "submapSize", "widthSIFT", "heightSIFT", "maxNumKeysPerImage", "numLocalNonLinIterations", "numLocalLinIterations",
          "numGlobalNonLinIterations", "numGlobalLinIterations", "downsampledWidth", "downsampledHeight", "minNumMatchesLocal", "minNumMatchesGlobal",
          "denseOverlapCheckSubsampleFactor", "numOptPerResidualRemoval")
    _F = ("verifySiftErrThresh", "verifySiftCorrThresh", "projCorrDistThres", "projCorrNormalThres", "projCorrColorThresh", "surfAreaPcaThresh", "verifyOptErrThresh",
          "verifyOptCorrThresh", "maxKabschResidual2", "minKeyScale", "siftMatchThresh", "siftMatchRatioMaxLocal", "siftMatchRatioMaxGlobal", "colorDownSigma",
          "depthDownSigmaD", "depthDownSigmaR", "optMaxResThresh", "denseDistThresh", "denseNormalThresh", "denseColorThresh", "denseColorGradientMin", "denseDepthMin",
          "denseDepthMax")
    _I = ("useComprehensiveFrameInvalidation", "useLocalVerify", "useLocalDense", "erodeSIFTdepth", "depthFilter")
    _F2 = ("depthSigmaD", "depthSigmaR")
    _fields_ = [(n, C.c_uint32) for n in _U] + [(n, C.c_float) for n in _F] + [("sensorDepthMin", C.c_float), ("sensorDepthMax", C.c_float)] + \
               [(n, C.c_int) for n in _I] + [(n, C.c_float) for n in _F2]


class _RefAppState(C.Structure):
    _fields_ = [("topNActive", C.c_uint32), ("numSolveFramesBeforeExit", C.c_uint32), ("minPoseDistSqrt", C.c_float), ("colorSigmaD", C.c_float),
                ("colorSigmaR", C.c_float), ("colorFilter", C.c_int)]


def set_reference_state(gas, gbs):
    """GlobalAppState / GlobalBundlingState singletons of the reference build <- the ctypes parameter structs of bundlefusion_amd.capi."""
    p = _RefBundlingState()
    for n in _RefBundlingState._U + _RefBundlingState._F + _RefBundlingState._F2:
        setattr(p, n, getattr(gbs, "s_" + n))
    for n in _RefBundlingState._I:
        setattr(p, n, int(getattr(gbs, "s_" + n)))
    p.sensorDepthMin, p.sensorDepthMax = gas.s_sensorDepthMin, gas.s_sensorDepthMax
    lib().ref_set_bundling_state(C.byref(p))
    a = _RefAppState(gas.s_topNActive, gas.s_numSolveFramesBeforeExit, gas.s_minPoseDistSqrt, gas.s_colorSigmaD, gas.s_colorSigmaR, int(gas.s_colorFilter))
    lib().ref_set_app_state(C.byref(a))


class _RefBundlerHandle(C.Structure):
    _fields_ = [("b", C.c_void_p), ("im", C.c_void_p), ("sensor", C.c_void_p)]


class RefBundlerView:
    """Accessors on one of the reference OnlineBundler's three Bundlers."""

    def __init__(self, handle):
        self._h = handle

    def num_frames(self):
        return int(lib().ref_bundler_num_frames(C.byref(self._h)))

    def trajectory(self, n):
        T = np.zeros((n, 4, 4), np.float32)
        lib().ref_bundler_get_trajectory(C.byref(self._h), _fp(T), n)
        return T

    def valid(self, n):
        v = np.zeros(n, np.int32)
        lib().ref_bundler_get_valid(C.byref(self._h), _fp(v), n)
        return v

    def num_keys(self, image):
        return int(lib().ref_bundler_num_keys(C.byref(self._h), image))

    def keys(self, image):
        n = self.num_keys(image)
        k = np.zeros((max(n, 1), 4), np.float32); d = np.zeros((max(n, 1), 128), np.uint8)
        lib().ref_bundler_get_keys(C.byref(self._h), image, _fp(k), _fp(d))
        return k[:n], d[:n]

    def correspondences(self):
        from bundlefusion_amd.capi import ENTRYJ_DTYPE
        n = int(lib().ref_bundler_num_correspondences(C.byref(self._h)))
        e = np.zeros(max(n, 1), ENTRYJ_DTYPE)
        if n:
            lib().ref_bundler_get_correspondences(C.byref(self._h), _fp(e), n)
        return e[:n]


class RefOnlineBundler:
    """The reference's OnlineBundler (OnlineBundler.cpp compiled as it is, with its Bundler / SBA / CUDASolverBundling / TrajectoryManager /
    SIFTImageManager / CUDACache / SiftGPU fork).  One frame: set_frame -> process_input -> (frame loop work) -> process."""
    STATE = ("last_processed", "last_valid", "local_to_solve", "last_local_solved", "past_end", "num_complete", "last_valid_complete", "tracking_lost",
             "process_state", "use_solve", "total_opt_local")

    def __init__(self, gas, gbs, width, height, K):
        set_reference_state(gas, gbs)
        L = lib()
        L.ref_ob_create.restype = C.c_void_p
        L.ref_ob_trajectory_manager.restype = C.c_void_p
        Kf = _f32(K).reshape(16)
        self.gbs = gbs
        self.W, self.H, self.WI, self.HI = width, height, gas.s_integrationWidth, gas.s_integrationHeight
        self.num_frames = 0
        self._h = C.c_void_p(L.ref_ob_create(width, height, width, height, self.WI, self.HI, _fp(Kf), _fp(Kf)))
        self.tm = RefTrajectoryManager.__new__(RefTrajectoryManager)
        self.tm._h = None
        self.tm_handle = C.c_void_p(L.ref_ob_trajectory_manager(self._h))

    def set_frame(self, depth, color):
        """A new sensor frame through the reference's own ingest (CUDAImageManager::process)."""
        ok = lib().ref_ob_set_frame(self._h, _fp(_f32(depth)), _fp(np.ascontiguousarray(color, np.uint8)))
        assert ok
        self.num_frames += 1

    def ingest_outputs(self, frame=None):
        """(raw depth, filtered depth) at sensor resolution of the CURRENT frame, (depth, colour) stored for integration of frame `frame`"""
        frame = self.num_frames - 1 if frame is None else frame
        raw, filt = np.zeros((self.H, self.W), np.float32), np.zeros((self.H, self.W), np.float32)
        di, ci = np.zeros((self.HI, self.WI), np.float32), np.zeros((self.HI, self.WI, 4), np.uint8)
        lib().ref_ob_ingest_outputs(self._h, _fp(raw), _fp(filt), C.c_uint32(frame), _fp(di), _fp(ci))
        return raw, filt, di, ci

    def override_filtered_depth(self, filt):
        lib().ref_ob_override_filtered_depth(self._h, _fp(_f32(filt)))

    def integration_intrinsics(self):
        K = np.zeros((4, 4), np.float32)
        lib().ref_ob_integration_intrinsics(self._h, _fp(K))
        return K

    def process_input(self):
        lib().ref_ob_process_input(self._h)

    def process(self):
        g = self.gbs
        lib().ref_ob_process(self._h, g.s_numLocalNonLinIterations, g.s_numLocalLinIterations, g.s_numGlobalNonLinIterations, g.s_numGlobalLinIterations)

    def current_integration_frame(self):
        T = np.zeros((4, 4), np.float32); idx, lost = C.c_uint32(), C.c_int()
        ok = lib().ref_ob_current_integration_frame(self._h, _fp(T), C.byref(idx), C.byref(lost))
        return bool(ok), T, idx.value, bool(lost.value)

    def state(self):
        s = np.zeros(11, np.int32)
        lib().ref_ob_state(self._h, _fp(s))
        return dict(zip(self.STATE, (int(v) for v in s)))

    def _traj(self, fn, n):
        T = np.zeros((n, 4, 4), np.float32)
        fn(self._h, _fp(T), n)
        return T

    def complete_trajectory(self, n):
        return self._traj(lib().ref_ob_complete_trajectory, n)

    def sift_trajectory(self, n):
        return self._traj(lib().ref_ob_sift_trajectory, n)

    def local_trajectories(self, n):
        return self._traj(lib().ref_ob_local_trajectories, n)

    def invalid_images_list(self, n):
        v = np.zeros(n, np.uint32)
        lib().ref_ob_invalid_images_list(self._h, _fp(v), n)
        return v

    def bundler(self, which):
        h = _RefBundlerHandle()
        lib().ref_ob_bundler(self._h, int(which), C.byref(h))
        return RefBundlerView(h)

    def trajectory_manager(self):
        """The OnlineBundler's TrajectoryManager behind the RefTrajectoryManager interface (not owned)."""
        t = RefTrajectoryManager.__new__(RefTrajectoryManager)
        t._h = self.tm_handle
        t.__class__ = _BorrowedTM
        return t


class RefFrameLoop:
    """The reference's frame loop compiled as it is (oracle/ref/ref_loop.cpp: DepthSensing.cpp:723-762, :853-902, :966-1129) over a RefOnlineBundler's objects and a
    RefScene(host_class=True): frame(depth, colour) = one OnD3D11FrameRender call with a new sensor frame, frame() = one call after the sequence has ended."""

    def __init__(self, rob, rscene, cam, max_frame_fixes):
        assert rscene._p == "ref_hscene_", "the compiled loop drives the reference's host class"
        self.rob, self.rscene = rob, rscene
        lib().ref_loop_bind(rob._h, rscene._h, C.byref(cam), C.c_uint32(max_frame_fixes))

    def frame(self, depth=None, color=None):
        if depth is None:
            lib().ref_loop_end_of_sequence(self.rob._h)
        else:
            lib().ref_loop_set_frame(self.rob._h, _fp(_f32(depth)), _fp(np.ascontiguousarray(color, np.uint8)))
            self.rob.num_frames += 1
        return bool(lib().ref_loop_frame_render())

    def ops(self, start=0):
        """[(kind "in" | "de", stored frame index, T)] the loop asked of the volume, from entry `start` on"""
        n = int(lib().ref_loop_num_ops())
        out = []
        for i in range(start, n):
            kind, frame = C.c_int(), C.c_int()
            T = np.zeros((4, 4), np.float32)
            lib().ref_loop_op(C.c_uint32(i), C.byref(kind), _fp(T), C.byref(frame))
            out.append(("in" if kind.value == 0 else "de", frame.value, T))
        return out


class _BorrowedTM(RefTrajectoryManager):
    def __del__(self):
        pass


def ray_cast_params_from_global_app_state(gas, intrinsics):
    """CUDARayCastSDF::parametersFromGlobalAppState (CUDARayCastSDF.h:24-52) of the reference -> capi.RayCastParams"""
    from bundlefusion_amd.capi import RayCastParams
    assert lib().ref_sizeof_ray_cast_params() == C.sizeof(RayCastParams)
    p = RayCastParams()
    lib().ref_ray_cast_params_from_global_app_state(C.c_uint32(gas.s_rayCastWidth), C.c_uint32(gas.s_rayCastHeight), C.c_uint32(gas.s_integrationWidth),
                                                    C.c_uint32(gas.s_integrationHeight), C.c_float(gas.s_renderDepthMin), C.c_float(gas.s_renderDepthMax),
                                                    C.c_float(gas.s_SDFRayIncrementFactor), C.c_float(gas.s_SDFTruncation), C.c_float(gas.s_SDFRayThresSampleDistFactor),
                                                    C.c_float(gas.s_SDFRayThresDistFactor), C.c_int(int(gas.s_SDFUseGradients)), C.c_uint32(gas.s_hashNumSDFBlocks),
                                                    _fp(_f32(intrinsics).reshape(16)), C.byref(p))
    return p


class RefBundler:
    """A reference Bundler of its own (Bundler.cpp compiled as it is): detect / cache / match-and-filter one image set.  set_reference_state first."""

    def __init__(self, max_images, max_keys, sift_intrinsics, width, height, depth_intrinsics, is_local, min_key_scale):
        L = lib()
        L.ref_bundler_create.restype = C.c_void_p
        Ks = _f32(sift_intrinsics).reshape(16)
        Ki = _f32(np.linalg.inv(np.asarray(sift_intrinsics, np.float64))).reshape(16)
        L.ref_set_sift_camera(width, height, width, height, _fp(Ks), C.c_float(min_key_scale))
        self._h = C.c_void_p(L.ref_bundler_create(max_images, max_keys, _fp(Ki), width, height, _fp(_f32(depth_intrinsics).reshape(16)), int(is_local)))
        self.W, self.H = width, height

    def add_frame(self, intensity, depth_filt, depth_raw, color):
        L = lib()
        i = _f32(intensity).copy(); f = _f32(depth_filt); r = _f32(depth_raw); c = np.ascontiguousarray(color, np.uint8)
        L.ref_bundler_detect_features(self._h, _fp(i), _fp(f))
        L.ref_bundler_store_cached_frame(self._h, self.W, self.H, _fp(c), self.W, self.H, _fp(r))

    def match_and_filter(self):
        L = lib()
        L.ref_bundler_match_and_filter.restype = C.c_uint32
        return int(L.ref_bundler_match_and_filter(self._h))

    def num_frames(self):
        return int(lib().ref_bundler_num_frames(self._h))

    def all_keys(self, capacity=65536):
        k = np.zeros((capacity, 4), np.float32)
        L = lib(); L.ref_bundler_get_all_keys.restype = C.c_uint32
        n = L.ref_bundler_get_all_keys(self._h, _fp(k), capacity)
        return k[:n].copy()

    def matches_view(self):
        L = lib(); L.ref_bundler_siftmgr_view.restype = C.c_void_p
        m = RefSiftManager.__new__(RefSiftManager)
        m._h = C.c_void_p(L.ref_bundler_siftmgr_view(self._h)); m._keep = []
        return m

    def cache_frame_depth(self, i, w, h):
        d = np.zeros((h, w), np.float32); p4 = np.zeros((h, w, 4), np.float32); it = np.zeros((h, w), np.float32); dv = np.zeros((h, w, 2), np.float32)
        n4 = np.zeros((h, w, 4), np.float32); wh = np.zeros(2, np.uint32)
        lib().ref_bundler_get_cache_frame(self._h, i, _fp(d), _fp(p4), _fp(it), _fp(dv), _fp(n4), _fp(wh))
        assert (int(wh[0]), int(wh[1])) == (w, h)
        return d

    def cache_intrinsics(self):
        K = np.zeros((4, 4), np.float32)
        lib().ref_bundler_cache_intrinsics(self._h, _fp(K))
        return K

    def handles(self):
        L = lib()
        L.ref_bundler_sift_manager.restype = C.c_void_p; L.ref_bundler_cuda_cache.restype = C.c_void_p
        return C.c_void_p(L.ref_bundler_sift_manager(self._h)), C.c_void_p(L.ref_bundler_cuda_cache(self._h))


class RefCorrespondenceEvaluator:
    """The reference's CorrespondenceEvaluator (CorrespondenceEvaluator.cpp compiled as it is) on a RefBundler's manager and cache."""

    def __init__(self, trajectory, log_prefix=""):
        L = lib(); L.ref_evaluator_create.restype = C.c_void_p
        T = _f32(trajectory).reshape(-1, 16)
        self._h = C.c_void_p(L.ref_evaluator_create(_fp(T), len(T), log_prefix.encode()))

    def evaluate(self, bundler, sift_intrinsics_inv, filtered, recompute, clear, corr_type):
        mgr, cache = bundler.handles()
        out = np.zeros(3, np.uint32)
        lib().ref_evaluator_evaluate(self._h, mgr, cache, _fp(_f32(sift_intrinsics_inv).reshape(16)), int(filtered), int(recompute), int(clear), corr_type.encode(), _fp(out))
        return tuple(int(x) for x in out)

    def has_gt_overlap(self, n):
        v = np.zeros(n, np.uint8)
        lib().ref_evaluator_has_gt_overlap(self._h, _fp(v), n)
        return v.astype(bool)

    def finish(self):
        lib().ref_evaluator_finish_logging(self._h)
