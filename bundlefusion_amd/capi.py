"""ctypes bindings of include/bf_hip.h (the C ABI of libbf_hip.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BF_LIB_PATH") or os.path.join(_HERE, "lib", "libbf_hip.so")      # BF_LIB_PATH: a variant build of the same library (tools/build_variant.py; diagnostics)


class BFError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "bundlefusion_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback)" % LIB_PATH)

try:
    # PyTorch ships its own libamdhip64/libhsa-runtime64.  Load it FIRST so that libbf_hip.so binds to the same HIP
    # runtime (same soname): two runtimes in one process have independent null streams and no mutual ordering.
    import torch  # noqa: F401
except ImportError:      # pure C-ABI use without torch is fine: the library then brings /opt/rocm's runtime
    pass
lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

SDF_BLOCK_SIZE = 8
HASH_BUCKET_SIZE = 4
FREE_ENTRY = -2
VOX_PER_BLOCK = 512


class HashParams(C.Structure):
    _fields_ = [
        ("m_rigidTransform", C.c_float * 16),
        ("m_rigidTransformInverse", C.c_float * 16),
        ("m_hashNumBuckets", C.c_uint32),
        ("m_hashBucketSize", C.c_uint32),
        ("m_hashMaxCollisionLinkedListSize", C.c_uint32),
        ("m_numSDFBlocks", C.c_uint32),
        ("m_SDFBlockSize", C.c_int32),
        ("m_virtualVoxelSize", C.c_float),
        ("m_numOccupiedBlocks", C.c_uint32),
        ("m_maxIntegrationDistance", C.c_float),
        ("m_truncScale", C.c_float),
        ("m_truncation", C.c_float),
        ("m_integrationWeightSample", C.c_uint32),
        ("m_integrationWeightMax", C.c_uint32),
        ("m_streamingVoxelExtents", C.c_float * 3),
        ("m_streamingGridDimensions", C.c_int32 * 3),
        ("m_streamingMinGridPos", C.c_int32 * 3),
        ("m_streamingInitialChunkListSize", C.c_uint32),
        ("m_dummy", C.c_uint32 * 2),
    ]


class DepthCameraParams(C.Structure):
    _fields_ = [
        ("fx", C.c_float), ("fy", C.c_float), ("mx", C.c_float), ("my", C.c_float),
        ("m_imageWidth", C.c_uint32), ("m_imageHeight", C.c_uint32),
        ("m_sensorDepthWorldMin", C.c_float), ("m_sensorDepthWorldMax", C.c_float),
    ]


class DepthCameraData(C.Structure):
    _fields_ = [("d_depthData", C.c_void_p), ("d_colorData", C.c_void_p)]


SCENE_BATCH_MAX = 12


class SceneBatchOp(C.Structure):
    """bf_scene_batch_op: kind 0 integrate(T0), 1 deIntegrate(T0), 2 deIntegrate(T0) + integrate(T1) of the same frame"""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("T0", C.c_float * 16), ("T1", C.c_float * 16), ("data", DepthCameraData),
                ("d_texels", C.c_void_p), ("wait_event", C.c_void_p)]


class HashData(C.Structure):
    _fields_ = [
        ("d_heap", C.c_void_p), ("d_heapCounter", C.c_void_p), ("d_hashDecision", C.c_void_p),
        ("d_hashDecisionPrefix", C.c_void_p), ("d_hash", C.c_void_p), ("d_hashCompactified", C.c_void_p),
        ("d_hashCompactifiedCounter", C.c_void_p), ("d_SDFBlocks", C.c_void_p), ("d_hashBucketMutex", C.c_void_p),
    ]


HASH_ENTRY_DTYPE = np.dtype([("pos", "<i4", 3), ("ptr", "<i4"), ("offset", "<u4"), ("_pad", "<u4", 3)])
VOXEL_DTYPE = np.dtype([("sdf", "<f4"), ("weight", "<f4"), ("color", "u1", 4)])
assert HASH_ENTRY_DTYPE.itemsize == 32 and VOXEL_DTYPE.itemsize == 12

lib.bf_last_error.restype = C.c_char_p


def bind_host_threads_to_device(device=0):
    """bf_bind_host_threads_to_device: all threads of this process onto the CPUs of the GPU's NUMA node; returns the cpu list ('' = unchanged)."""
    buf = C.create_string_buffer(1024)
    lib.bf_bind_host_threads_to_device(int(device), buf, C.c_size_t(len(buf)))
    return buf.value.decode()


lib.bf_version.restype = C.c_char_p


def check(rc):
    if rc != 0:
        raise BFError("libbf_hip status %d: %s" % (rc, lib.bf_last_error().decode()))


def mat16(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(16))
    return (C.c_float * 16)(*a.tolist())


def default_hash_params(num_buckets=800000, num_sdf_blocks=200000, voxel_size=0.010, max_integration_distance=3.0,
                        truncation=0.06, trunc_scale=0.02, weight_sample=1, weight_max=99999999, max_chain=7):
    """CUDASceneRepHashSDF::parametersFromGlobalAppState with zParametersDefault.txt values."""
    p = HashParams()
    eye = np.eye(4, dtype=np.float32).reshape(16)
    p.m_rigidTransform[:] = eye.tolist()
    p.m_rigidTransformInverse[:] = eye.tolist()
    p.m_hashNumBuckets = num_buckets
    p.m_hashBucketSize = HASH_BUCKET_SIZE
    p.m_hashMaxCollisionLinkedListSize = max_chain
    p.m_numSDFBlocks = num_sdf_blocks
    p.m_SDFBlockSize = SDF_BLOCK_SIZE
    p.m_virtualVoxelSize = voxel_size
    p.m_numOccupiedBlocks = 0
    p.m_maxIntegrationDistance = max_integration_distance
    p.m_truncScale = trunc_scale
    p.m_truncation = truncation
    p.m_integrationWeightSample = weight_sample
    p.m_integrationWeightMax = weight_max
    p.m_streamingVoxelExtents[:] = [1.0, 1.0, 1.0]
    p.m_streamingGridDimensions[:] = [257, 257, 257]
    p.m_streamingMinGridPos[:] = [-128, -128, -128]
    p.m_streamingInitialChunkListSize = 2000
    return p


def camera_params(width, height, fx, fy, mx, my, dmin=0.1, dmax=4.0):
    c = DepthCameraParams()
    c.fx, c.fy, c.mx, c.my = fx, fy, mx, my
    c.m_imageWidth, c.m_imageHeight = width, height
    c.m_sensorDepthWorldMin, c.m_sensorDepthWorldMax = dmin, dmax
    return c


def alloc_comm_capacity(W, H, voxel, world):
    """Keys per rank and operator for bf_scene_set_alloc_comm / bf_pipeline_set_comm: twice the estimate of the distinct in-frustum blocks the rays of a W x H frame
    cross (0.22 per pixel at 2 mm, scaling with 1 / voxel^2; tests/test_host_cpu.py holds it against measured key counts), divided over the ranks, as a power of
    two >= 65536."""
    est = 0.22 * W * H * (0.002 / voxel) ** 2
    cap = 1 << 16
    while cap < 2.0 * est / max(world, 1):
        cap <<= 1
    return cap


class SceneRepHashSDF:
    """Python view of `bf_scene` (== the reference's CUDASceneRepHashSDF)."""

    def __init__(self, params, stream=None):
        self._h = C.c_void_p()
        self.params = params
        check(lib.bf_scene_create(C.byref(params), C.byref(self._h)))
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            lib.bf_scene_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(lib.bf_scene_set_stream(self._h, C.c_void_p(stream_ptr)))

    def reset(self):
        check(lib.bf_scene_reset(self._h))

    @staticmethod
    def _data(depth, color):
        d = DepthCameraData()
        d.d_depthData = depth.data_ptr()
        d.d_colorData = color.data_ptr() if color is not None else None
        return d

    def integrate(self, cam_to_world, depth, color, cam):
        d = self._data(depth, color)
        check(lib.bf_scene_integrate(self._h, mat16(cam_to_world), C.byref(d), C.byref(cam), None))

    def deintegrate(self, cam_to_world, depth, color, cam):
        d = self._data(depth, color)
        check(lib.bf_scene_deintegrate(self._h, mat16(cam_to_world), C.byref(d), C.byref(cam), None))

    def set_overlap(self, enable=True):
        check(lib.bf_scene_set_overlap(self._h, int(enable)))

    def set_shard(self, rank, world):
        check(lib.bf_scene_set_shard(self._h, rank, world))

    def set_external_alloc(self, enable=True):
        check(lib.bf_scene_set_external_alloc(self._h, int(enable)))

    def set_alloc_comm(self, comm, capacity_keys=1 << 17):
        """The operators' own allocation with the ray march divided over the ranks of `comm` (capi.Comm or None): bf_scene_set_alloc_comm.  capacity_keys: see
        alloc_comm_capacity(W, H, voxel, world) (the default covers 640x480 @4 mm on one rank four times over)."""
        check(lib.bf_scene_set_alloc_comm(self._h, comm._h if comm is not None else None, int(capacity_keys)))
        self._alloc_comm = comm          # keep the callback alive

    def alloc_collect(self, cam_to_world, depth, cam, part, parts, keys, slots, count):
        """keys: torch int64 [capacity] cuda, slots: int32 [capacity], count: int32 [1] (see bf_scene_alloc_collect)"""
        d = self._data(depth, None)
        check(lib.bf_scene_alloc_collect(self._h, mat16(cam_to_world), C.byref(d), C.byref(cam), int(part), int(parts), C.c_void_p(keys.data_ptr()),
                                         C.c_void_p(slots.data_ptr()), C.c_void_p(count.data_ptr()), int(keys.numel())))

    def alloc_ingest(self, keys, count):
        check(lib.bf_scene_alloc_ingest(self._h, C.c_void_p(keys.data_ptr()), C.c_void_p(count.data_ptr()), int(keys.numel())))

    def alloc_place(self):
        check(lib.bf_scene_alloc_place(self._h))

    def alloc_sync(self):
        check(lib.bf_scene_alloc_sync(self._h))

    def set_arith(self, mode):
        """'fast' (the reference GPU build's -use_fast_math contract; library default) or 'exact' (IEEE op by op, bit-comparable with the oracle): bf_scene_set_arith"""
        check(lib.bf_scene_set_arith(self._h, {"exact": 0, "fast": 1}[mode]))

    def arith(self):
        m = C.c_int()
        check(lib.bf_scene_get_arith(self._h, C.byref(m)))
        return ("exact", "fast")[m.value]

    def reintegrate(self, old_cam_to_world, new_cam_to_world, depth, color, cam):
        """fused deintegrate(old) + integrate(new) of the same frame"""
        data = self._data(depth, color)
        check(lib.bf_scene_reintegrate(self._h, mat16(old_cam_to_world), mat16(new_cam_to_world), C.byref(data), C.byref(cam)))

    def run_batch(self, ops, cam):
        """ops: list of (kind, T0, T1 or None, depth, color) - kind "in" / 0 integrate(T0), "de" / 1 deIntegrate(T0), "re" / 2 deIntegrate(T0) + integrate(T1) -
        executed as ONE batch (bf_scene_run_batch): same volume as the same operators issued one by one, in that order."""
        assert 1 <= len(ops) <= SCENE_BATCH_MAX
        arr = (SceneBatchOp * len(ops))()
        for o, (kind, T0, T1, depth, color) in zip(arr, ops):
            o.kind = {"in": 0, "de": 1, "re": 2}.get(kind, kind)
            o.T0[:] = np.asarray(T0, np.float32).reshape(16).tolist()
            o.T1[:] = np.asarray(T1 if T1 is not None else T0, np.float32).reshape(16).tolist()
            o.data = self._data(depth, color)
            o.d_texels = None; o.wait_event = None
        check(lib.bf_scene_run_batch(self._h, arr, len(ops), C.byref(cam)))

    def garbage_collect(self):
        check(lib.bf_scene_garbage_collect(self._h))

    def compactify(self, cam_to_world, cam):
        check(lib.bf_scene_set_last_rigid_transform_and_compactify(self._h, mat16(cam_to_world), C.byref(cam)))

    def hash_data(self):
        hd = HashData()
        check(lib.bf_scene_get_hash_data(self._h, C.byref(hd)))
        return hd

    def hash_params(self):
        p = HashParams()
        check(lib.bf_scene_get_hash_params(self._h, C.byref(p)))
        return p

    def heap_free_count(self):
        v = C.c_uint32()
        check(lib.bf_scene_get_heap_free_count(self._h, C.byref(v)))
        return v.value

    def num_allocated_blocks(self):
        v = C.c_uint32()
        check(lib.bf_scene_get_num_allocated_blocks(self._h, C.byref(v)))
        return v.value

    def num_integrated_frames(self):
        v = C.c_uint32()
        check(lib.bf_scene_get_num_integrated_frames(self._h, C.byref(v)))
        return v.value

    def debug_hash(self):
        out = (C.c_uint32 * 6)()
        check(lib.bf_scene_debug_hash(self._h, out))
        return dict(occupied=out[0], heap_free=out[1], duplicate_keys=out[2], free_and_allocated=out[3], leaked=out[4],
                    dropped=out[5])

    def find_blocks(self, pos):
        """pos: torch int32 [n, 3] cuda (block coordinates) -> torch int32 [n]: ptr or FREE_ENTRY (bf_scene_debug_find_blocks)"""
        import torch
        out = torch.empty(pos.shape[0], dtype=torch.int32, device=pos.device)
        check(lib.bf_scene_debug_find_blocks(self._h, C.c_void_p(pos.data_ptr()), int(pos.shape[0]), C.c_void_p(out.data_ptr())))
        return out

    def kernel_timing(self, enable):
        check(lib.bf_scene_kernel_timing(self._h, int(enable)))

    def kernel_timing_read(self):
        n, ms = C.c_uint32(), C.c_float()
        check(lib.bf_scene_kernel_timing_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_timing_occupied(self):
        v = C.c_uint64(); n = C.c_uint32()
        check(lib.bf_scene_kernel_timing_occupied(self._h, C.byref(v), C.byref(n)))
        return v.value, n.value

    def kernel_timing_blocks(self):
        """(operator blocks [a fused launch counts both lists], blocks visited by plain launches, blocks visited by fused launches [union list], #operators)"""
        a = C.c_uint64(); b = C.c_uint64(); c = C.c_uint64(); n = C.c_uint32()
        check(lib.bf_scene_kernel_timing_blocks(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def kernel_timing_images(self):
        """frames (depth + colour images) the timed launches sampled: one per operator, n per batch of n"""
        n = C.c_uint32()
        check(lib.bf_scene_kernel_timing_images(self._h, C.byref(n)))
        return n.value

    # ---- test helpers: copy raw arrays back (hipMemcpy through torch) ----
    def download(self):
        """Returns (hash[numBuckets*4], heap[numSDFBlocks], heapCounter, voxels[numSDFBlocks*512]) as numpy."""
        import torch
        torch.cuda.synchronize()
        hd = self.hash_data()
        nE = self.params.m_hashNumBuckets * HASH_BUCKET_SIZE
        nB = self.params.m_numSDFBlocks
        hash_np = _d2h(hd.d_hash, nE * 32).view(HASH_ENTRY_DTYPE)
        heap_np = _d2h(hd.d_heap, nB * 4).view("<u4")
        heap_counter = int(_d2h(hd.d_heapCounter, 4).view("<u4")[0])
        vox_np = _d2h(hd.d_SDFBlocks, nB * VOX_PER_BLOCK * 12).view(VOXEL_DTYPE)
        return hash_np, heap_np, heap_counter, vox_np

    def download_compactified(self):
        import torch
        torch.cuda.synchronize()
        hd = self.hash_data()
        n = int(_d2h(hd.d_hashCompactifiedCounter, 4).view("<i4")[0])
        return _d2h(hd.d_hashCompactified, n * 32).view(HASH_ENTRY_DTYPE)


_hip = None


def _d2h(ptr, nbytes):
    """D2H of a raw device pointer into a fresh numpy byte array (through the library's own runtime, fenced)."""
    out = np.empty(nbytes, dtype=np.uint8)
    check(lib.bf_memcpy_d2h(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes)))
    return out


# --------------------------------------------------------------------------- cache + solver
class CachedFrame(C.Structure):
    _fields_ = [("d_depthDownsampled", C.c_void_p), ("d_cameraposDownsampled", C.c_void_p), ("d_intensityDownsampled", C.c_void_p),
                ("d_intensityDerivsDownsampled", C.c_void_p), ("d_normalsDownsampledUCHAR4", C.c_void_p),
                ("d_normalsDownsampled", C.c_void_p)]


class SolverConfig(C.Structure):
    _fields_ = [("optMaxResThresh", C.c_float), ("denseDistThresh", C.c_float), ("denseNormalThresh", C.c_float),
                ("denseColorThresh", C.c_float), ("denseColorGradientMin", C.c_float), ("denseDepthMin", C.c_float),
                ("denseDepthMax", C.c_float), ("denseOverlapCheckSubsampleFactor", C.c_uint32), ("verifyOptDistThresh", C.c_float),
                ("verifyOptPercentThresh", C.c_float), ("recordConvergence", C.c_int32)]


ENTRYJ_DTYPE = np.dtype([("imgIdx_i", "<u4"), ("imgIdx_j", "<u4"), ("pos_i", "<f4", 3), ("pos_j", "<f4", 3)])
assert ENTRYJ_DTYPE.itemsize == 32


def default_solver_config(record_convergence=False):
    """zParametersBundlingDefault.txt values."""
    c = SolverConfig()
    c.optMaxResThresh = 0.08
    c.denseDistThresh = 0.15
    c.denseNormalThresh = 0.97
    c.denseColorThresh = 0.1
    c.denseColorGradientMin = 0.005
    c.denseDepthMin = 0.5
    c.denseDepthMax = 4.0
    c.denseOverlapCheckSubsampleFactor = 4
    c.verifyOptDistThresh = 0.02
    c.verifyOptPercentThresh = 0.05
    c.recordConvergence = int(record_convergence)
    return c


def intrinsics_matrix(fx, fy, mx, my):
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, mx, my
    return K


class Cache:
    """Python view of `bf_cache` (== the reference's CUDACache)."""

    def __init__(self, depth_w, depth_h, w, h, max_images, input_intrinsics, color_sigma=2.5, depth_sigma_d=1.0, depth_sigma_r=0.05,
                 stream=None):
        self._h = C.c_void_p()
        self.w, self.h, self.max_images = w, h, max_images
        check(lib.bf_cache_create(depth_w, depth_h, w, h, max_images, mat16(input_intrinsics), C.c_float(color_sigma),
                                  C.c_float(depth_sigma_d), C.c_float(depth_sigma_r), C.byref(self._h)))
        if stream is not None:
            check(lib.bf_cache_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.bf_cache_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store_frame(self, depth, color):
        dh, dw = depth.shape[:2]
        ch, cw = color.shape[:2]
        check(lib.bf_cache_store_frame(self._h, C.c_void_p(depth.data_ptr()), dw, dh, C.c_void_p(color.data_ptr()), cw, ch))

    def reset(self):
        check(lib.bf_cache_reset(self._h))

    def num_frames(self):
        v = C.c_uint32()
        check(lib.bf_cache_get_num_frames(self._h, C.byref(v)))
        return v.value

    def frames_gpu(self):
        p = C.c_void_p()
        check(lib.bf_cache_get_frames_gpu(self._h, C.byref(p)))
        return p.value

    def geometry(self):
        w, h = C.c_uint32(), C.c_uint32()
        k = (C.c_float * 4)()
        check(lib.bf_cache_get_geometry(self._h, C.byref(w), C.byref(h), k))
        return w.value, h.value, [k[0], k[1], k[2], k[3]]

    def download_frame(self, i):
        import torch
        torch.cuda.synchronize()
        f = CachedFrame()
        check(lib.bf_cache_get_frame(self._h, i, C.byref(f)))
        n = self.w * self.h
        return dict(
            depth=_d2h(f.d_depthDownsampled, n * 4).view("<f4").reshape(self.h, self.w),
            campos=_d2h(f.d_cameraposDownsampled, n * 16).view("<f4").reshape(self.h, self.w, 4),
            intensity=_d2h(f.d_intensityDownsampled, n * 4).view("<f4").reshape(self.h, self.w),
            derivs=_d2h(f.d_intensityDerivsDownsampled, n * 8).view("<f4").reshape(self.h, self.w, 2),
            normals_u=_d2h(f.d_normalsDownsampledUCHAR4, n * 4).reshape(self.h, self.w, 4),
            normals=_d2h(f.d_normalsDownsampled, n * 16).view("<f4").reshape(self.h, self.w, 4),
        )


class Solver:
    """Python view of `bf_solver` (== the reference's CUDASolverBundling)."""

    def __init__(self, max_images, max_residuals, cfg=None, stream=None):
        self._h = C.c_void_p()
        self.cfg = cfg or default_solver_config()
        check(lib.bf_solver_create(max_images, max_residuals, C.byref(self.cfg), C.byref(self._h)))
        if stream is not None:
            check(lib.bf_solver_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.bf_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, corr, num_corr, valid, num_images, n_nonlin, n_lin, cache, weights_sparse, weights_dense_depth, weights_dense_color,
              rot, trans, use_pairwise=True, rebuild_jt=True, find_max_residual=False, revalidate_idx=0xFFFFFFFF):
        nw = len(weights_sparse)
        ws = (C.c_float * nw)(*weights_sparse)
        wd = (C.c_float * nw)(*weights_dense_depth)
        wc = (C.c_float * nw)(*weights_dense_color)
        if cache is not None:
            w, h, k = cache.geometry()
            frames, k4 = C.c_void_p(cache.frames_gpu()), (C.c_float * 4)(*k)
        else:
            w = h = 0
            frames, k4 = None, None
        check(lib.bf_solver_solve(self._h, C.c_void_p(corr.data_ptr()) if corr is not None else None, num_corr,
                                  C.c_void_p(valid.data_ptr()), num_images, n_nonlin, n_lin, frames, w, h, k4, ws, wd, wc, nw,
                                  int(use_pairwise), C.c_void_p(rot.data_ptr()), C.c_void_p(trans.data_ptr()), int(rebuild_jt),
                                  int(find_max_residual), C.c_uint32(revalidate_idx)))

    def max_residual(self):
        m, i = C.c_float(), C.c_int32()
        check(lib.bf_solver_get_max_residual(self._h, C.byref(m), C.byref(i)))
        return m.value, i.value

    def max_residual_pair(self, cur_frame, corr):
        idx = (C.c_uint32 * 2)()
        m, rm = C.c_float(), C.c_int()
        check(lib.bf_solver_get_max_residual_pair(self._h, cur_frame, C.c_void_p(corr.data_ptr()), idx, C.byref(m), C.byref(rm)))
        return (idx[0], idx[1]), m.value, bool(rm.value)

    def use_verification(self, corr, num_corr):
        out = C.c_int()
        check(lib.bf_solver_use_verification(self._h, C.c_void_p(corr.data_ptr()), num_corr, C.byref(out)))
        return bool(out.value)

    def convergence(self):
        buf = (C.c_float * 40)()
        n = C.c_uint32()
        check(lib.bf_solver_get_convergence(self._h, buf, 40, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def iteration_counts(self):
        buf = (C.c_int32 * 33)()
        check(lib.bf_solver_get_iteration_counts(self._h, buf, 33))
        return buf[0], [buf[1 + i] for i in range(buf[0])]

    def debug_dense_system(self, n):
        dim = 6 * n
        JtJ = np.zeros((dim, dim), dtype=np.float32)
        Jtr = np.zeros(dim, dtype=np.float32)
        npairs = C.c_int32()
        check(lib.bf_solver_debug_dense_system(self._h, JtJ.ctypes.data_as(C.c_void_p), Jtr.ctypes.data_as(C.c_void_p), n, C.byref(npairs)))
        return JtJ, Jtr, npairs.value


def convert_matrices_to_poses(T, rot, trans, valid, stream=0):
    check(lib.bf_convert_matrices_to_poses(C.c_void_p(T.data_ptr()), T.shape[0], C.c_void_p(rot.data_ptr()), C.c_void_p(trans.data_ptr()),
                                           C.c_void_p(valid.data_ptr()), C.c_void_p(stream)))


def convert_poses_to_matrices(rot, trans, T, valid, stream=0):
    check(lib.bf_convert_poses_to_matrices(C.c_void_p(rot.data_ptr()), C.c_void_p(trans.data_ptr()), T.shape[0], C.c_void_p(T.data_ptr()),
                                           C.c_void_p(valid.data_ptr()), C.c_void_p(stream)))


# --------------------------------------------------------------------------- SIFT detection
KEYPOINT_DTYPE = np.dtype([("pos", "<f4", 2), ("scale", "<f4"), ("depth", "<f4")])


class Sift:
    """Python view of `bf_sift` (== the reference's SiftGPU detection object)."""

    def __init__(self, width, height, depth_w, depth_h, feature_count_threshold=150, depth_min=0.1, depth_max=4.0, min_key_scale=3.0,
                 max_keys=1024, stream=None):
        self._h = C.c_void_p()
        self.w, self.h, self.max_keys = width, height, max_keys
        check(lib.bf_sift_create(width, height, depth_w, depth_h, feature_count_threshold, C.c_float(depth_min), C.c_float(depth_max),
                                 C.c_float(min_key_scale), max_keys, C.byref(self._h)))
        if stream is not None:
            check(lib.bf_sift_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.bf_sift_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, intensity, depth, keys, descs, count):
        check(lib.bf_sift_run(self._h, C.c_void_p(intensity.data_ptr()), C.c_void_p(depth.data_ptr()), C.c_void_p(keys.data_ptr()),
                              C.c_void_p(descs.data_ptr()), C.c_void_p(count.data_ptr())))

    def debug_level(self, octave, index):
        out = np.empty(((self.h >> octave), (self.w >> octave)), dtype=np.float32)
        check(lib.bf_sift_debug_level(self._h, octave, index, out.ctypes.data_as(C.c_void_p)))
        return out

    def debug_counts(self):
        out = (C.c_int32 * 26)()
        check(lib.bf_sift_debug_counts(self._h, out))
        return dict(num_raw=out[0], num_feat=out[1], level0=[out[2 + i] for i in range(12)], level1=[out[14 + i] for i in range(12)])


def rgbx_to_intensity(color):
    """CUDAImageUtil convertToIntensity (CUDAImageUtil.cu:204-207) in float32, same op order."""
    c = color.astype(np.float32)
    return ((np.float32(0.299) * c[..., 0] + np.float32(0.587) * c[..., 1]) + np.float32(0.114) * c[..., 2]) / np.float32(255.0)


# --------------------------------------------------------------------------- keypoint / match store
class SiftImageGPU(C.Structure):
    _fields_ = [("d_keyPoints", C.c_void_p), ("d_keyPointDescs", C.c_void_p), ("d_numKeyPoints", C.c_void_p)]


def _h2d(ptr, arr):
    arr = np.ascontiguousarray(arr)
    check(lib.bf_memcpy_h2d(C.c_void_p(ptr), C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)))


def _f16(m):
    return (C.c_float * 16)(*np.asarray(m, np.float32).reshape(16))


class SiftManager:
    """Python view of `bf_siftmgr` (== the reference's SIFTImageManager + the matcher it feeds)."""
    MAX_RAW, MAX_FILT = 128, 25

    def __init__(self, max_images, max_keys=1024, stream=None):
        self._h = C.c_void_p()
        self.max_images, self.max_keys = max_images, max_keys
        check(lib.bf_siftmgr_create(max_images, max_keys, C.byref(self._h)))
        if stream is not None:
            check(lib.bf_siftmgr_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.bf_siftmgr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib.bf_siftmgr_reset(self._h))

    def create_image(self):
        img = SiftImageGPU()
        check(lib.bf_siftmgr_create_image(self._h, C.byref(img)))
        return img

    def finalize_image(self, num_keys=-1):
        check(lib.bf_siftmgr_finalize_image(self._h, int(num_keys)))

    def add_image_host(self, keys, descs):
        """createSIFTImageGPU + H2D upload + finalize (the fuseToGlobal / loadFromFile way of filling an image)."""
        img = self.create_image()
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 4); descs = np.ascontiguousarray(descs, np.uint8).reshape(-1, 128)
        _h2d(img.d_keyPoints, keys); _h2d(img.d_keyPointDescs, descs)
        self.finalize_image(len(keys))

    def add_image_sift(self, sift, intensity, depth):
        """detectFeatures (Bundler.cpp:91-101): the detector writes keys, descriptors and the count straight into the store."""
        img = self.create_image()
        check(lib.bf_sift_run(sift._h, C.c_void_p(intensity.data_ptr()), C.c_void_p(depth.data_ptr()), C.c_void_p(img.d_keyPoints),
                              C.c_void_p(img.d_keyPointDescs), C.c_void_p(img.d_numKeyPoints)))
        self.finalize_image(-1)

    def num_images(self):
        n = C.c_uint32()
        check(lib.bf_siftmgr_get_num_images(self._h, C.byref(n)))
        return n.value

    def num_keypoints(self, first=0, count=None):
        count = self.num_images() - first if count is None else count
        out = np.zeros(count, np.int32)
        check(lib.bf_siftmgr_get_num_keypoints(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def download_image(self, i):
        img = SiftImageGPU()
        check(lib.bf_siftmgr_get_image(self._h, i, C.byref(img)))
        n = max(int(self.num_keypoints(i, 1)[0]), 0)
        keys = _d2h(img.d_keyPoints, 16 * n).view(np.float32).reshape(n, 4)
        descs = _d2h(img.d_keyPointDescs, 128 * n).reshape(n, 128)
        return keys, descs

    def set_current_frame(self, i):
        check(lib.bf_siftmgr_set_current_frame(self._h, i))

    def current_frame(self):
        n = C.c_uint32()
        check(lib.bf_siftmgr_get_current_frame(self._h, C.byref(n)))
        return n.value

    def match(self, cur, start, num, dist_max=0.7, ratio_max=0.8):
        check(lib.bf_siftmgr_match(self._h, cur, start, num, C.c_float(dist_max), C.c_float(ratio_max)))

    def filter_keypoint_matches(self, cur, start, num, Kinv, min_matches=5, max_res2=0.0004):
        check(lib.bf_siftmgr_filter_keypoint_matches(self._h, cur, start, num, _f16(Kinv), min_matches, C.c_float(max_res2)))

    def filter_surface_area(self, cur, start, num, Kinv, area_thresh=0.032):
        check(lib.bf_siftmgr_filter_matches_by_surface_area(self._h, cur, start, num, _f16(Kinv), C.c_float(area_thresh)))

    def filter_dense_verify(self, cur, start, num, W, H, K, d_frames, dist_thresh=0.15, normal_thresh=0.97, color_thresh=0.1, err_thresh=0.075,
                            corr_thresh=0.02, dmin=0.1, dmax=3.0):
        check(lib.bf_siftmgr_filter_matches_by_dense_verify(self._h, cur, start, num, W, H, _f16(K), C.c_void_p(d_frames), C.c_float(dist_thresh),
                                                            C.c_float(normal_thresh), C.c_float(color_thresh), C.c_float(err_thresh),
                                                            C.c_float(corr_thresh), C.c_float(dmin), C.c_float(dmax)))

    def filter_frames(self, cur, start, num):
        last = C.c_uint32()
        check(lib.bf_siftmgr_filter_frames(self._h, cur, start, num, C.byref(last)))
        return last.value

    def filter_frames_async(self, cur, start, num):
        check(lib.bf_siftmgr_filter_frames_async(self._h, cur, start, num))

    def add_curr_to_residuals(self, cur, start, num, Kinv):
        check(lib.bf_siftmgr_add_curr_to_residuals(self._h, cur, start, num, _f16(Kinv)))

    def sync_frame_result(self, cur):
        last = C.c_uint32(); nk = C.c_int32()
        check(lib.bf_siftmgr_sync_frame_result(self._h, cur, C.byref(last), C.byref(nk)))
        return last.value, nk.value

    def invalidate_image_to_image(self, i, j):
        check(lib.bf_siftmgr_invalidate_image_to_image(self._h, i, j))

    def check_for_invalid_frames(self, d_num_entries, num_vars, simple=False):
        f = lib.bf_siftmgr_check_for_invalid_frames_simple if simple else lib.bf_siftmgr_check_for_invalid_frames
        check(f(self._h, C.c_void_p(d_num_entries), num_vars))

    def verify_trajectory(self, num_images, d_traj, W, H, K, d_frames, dist_thresh=0.15, normal_thresh=0.97, color_thresh=0.1, err_thresh=0.05,
                          corr_thresh=0.001, dmin=0.1, dmax=3.0):
        v = C.c_int32()
        check(lib.bf_siftmgr_verify_trajectory(self._h, num_images, C.c_void_p(d_traj), W, H, _f16(K), C.c_void_p(d_frames), C.c_float(dist_thresh),
                                               C.c_float(normal_thresh), C.c_float(color_thresh), C.c_float(err_thresh), C.c_float(corr_thresh),
                                               C.c_float(dmin), C.c_float(dmax), C.byref(v)))
        return v.value

    def valid_images(self, count=None):
        count = self.num_images() if count is None else count
        out = np.zeros(count, np.int32)
        check(lib.bf_siftmgr_get_valid_images(self._h, out.ctypes.data_as(C.c_void_p), count))
        return out

    def set_valid_image(self, frame, valid):
        check(lib.bf_siftmgr_set_valid_image(self._h, frame, int(valid)))

    def update_gpu_valid_images(self):
        check(lib.bf_siftmgr_update_gpu_valid_images(self._h))

    def valid_images_gpu(self):
        p = C.c_void_p()
        check(lib.bf_siftmgr_get_valid_images_gpu(self._h, C.byref(p)))
        return p.value

    def num_global_correspondences(self):
        n = C.c_uint32()
        check(lib.bf_siftmgr_get_num_global_correspondences(self._h, C.byref(n)))
        return n.value

    def global_correspondences_gpu(self):
        p = C.c_void_p()
        check(lib.bf_siftmgr_get_global_correspondences_gpu(self._h, C.byref(p)))
        return p.value

    def download_global_correspondences(self):
        n = self.num_global_correspondences()
        corr = _d2h(self.global_correspondences_gpu(), 32 * n).view(ENTRYJ_DTYPE)
        p = C.c_void_p()
        check(lib.bf_siftmgr_get_global_correspondence_keys_gpu(self._h, C.byref(p)))
        keys = _d2h(p.value, 8 * n).view(np.uint32).reshape(n, 2)
        return corr, keys

    def set_global_correspondences(self, corr):
        corr = np.ascontiguousarray(corr)
        check(lib.bf_siftmgr_set_global_correspondences(self._h, corr.ctypes.data_as(C.c_void_p), len(corr)))

    def raw_matches(self, pair):
        n = C.c_int32(); idx = np.zeros((128, 2), np.uint32); dist = np.zeros(128, np.float32)
        check(lib.bf_siftmgr_get_raw_matches(self._h, pair, C.byref(n), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
        return n.value, idx, dist

    def filt_matches(self, pair):
        n = C.c_int32(); idx = np.zeros((25, 2), np.uint32); dist = np.zeros(25, np.float32)
        T = np.zeros((4, 4), np.float32); Ti = np.zeros((4, 4), np.float32)
        check(lib.bf_siftmgr_get_filt_matches(self._h, pair, C.byref(n), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
                                              T.ctypes.data_as(C.c_void_p), Ti.ctypes.data_as(C.c_void_p)))
        return n.value, idx, dist, T, Ti

    def fuse_to_global(self, glob, K, d_transforms, Kinv, host=False):
        """fuseToGlobal on the device (default) or in the reference's host form (host=True); same results bit for bit"""
        fn = lib.bf_siftmgr_fuse_to_global_host if host else lib.bf_siftmgr_fuse_to_global
        check(fn(self._h, glob._h, _f16(K), C.c_void_p(d_transforms), _f16(Kinv)))

    def fuse_error(self):
        e = C.c_int()
        check(lib.bf_siftmgr_fuse_error(self._h, C.byref(e)))
        return e.value


# --------------------------------------------------------------------------- host-level operators (include/bf_pipeline.h)
class GlobalAppState(C.Structure):
    _fields_ = [
        ("s_sensorIdx", C.c_uint32), ("s_integrationWidth", C.c_uint32), ("s_integrationHeight", C.c_uint32),
        ("s_maxFrameFixes", C.c_uint32), ("s_topNActive", C.c_uint32), ("s_minPoseDistSqrt", C.c_float),
        ("s_sensorDepthMax", C.c_float), ("s_sensorDepthMin", C.c_float), ("s_renderDepthMax", C.c_float), ("s_renderDepthMin", C.c_float),
        ("s_hashNumBuckets", C.c_uint32), ("s_hashNumSDFBlocks", C.c_uint32), ("s_hashMaxCollisionLinkedListSize", C.c_uint32),
        ("s_SDFVoxelSize", C.c_float), ("s_SDFTruncation", C.c_float), ("s_SDFTruncationScale", C.c_float), ("s_SDFMaxIntegrationDistance", C.c_float),
        ("s_SDFIntegrationWeightSample", C.c_uint32), ("s_SDFIntegrationWeightMax", C.c_uint32),
        ("s_colorSigmaD", C.c_float), ("s_colorSigmaR", C.c_float), ("s_colorFilter", C.c_int32),
        ("s_integrationEnabled", C.c_int32), ("s_garbageCollectionEnabled", C.c_int32), ("s_reconstructionEnabled", C.c_int32), ("s_streamingEnabled", C.c_int32),
        ("s_bUseCameraCalibration", C.c_int32), ("s_binaryDumpSensorUseTrajectory", C.c_int32), ("s_garbageCollectionStarve", C.c_uint32),
        ("s_streamingVoxelExtents", C.c_float * 3), ("s_streamingGridDimensions", C.c_int32 * 3), ("s_streamingMinGridPos", C.c_int32 * 3),
        ("s_streamingInitialChunkListSize", C.c_uint32), ("s_numSolveFramesBeforeExit", C.c_uint32),
        ("s_binaryDumpSensorFile", C.c_char * 512),
        ("s_rayCastWidth", C.c_uint32), ("s_rayCastHeight", C.c_uint32),
        ("s_SDFRayIncrementFactor", C.c_float), ("s_SDFRayThresSampleDistFactor", C.c_float), ("s_SDFRayThresDistFactor", C.c_float),
        ("s_SDFUseGradients", C.c_int32),
    ]


class GlobalBundlingState(C.Structure):
    _fields_ = [
        ("s_enableGlobalTimings", C.c_int32), ("s_enablePerFrameTimings", C.c_int32),
        ("s_maxNumImages", C.c_uint32), ("s_submapSize", C.c_uint32), ("s_widthSIFT", C.c_uint32), ("s_heightSIFT", C.c_uint32), ("s_maxNumKeysPerImage", C.c_uint32),
        ("s_numLocalNonLinIterations", C.c_uint32), ("s_numLocalLinIterations", C.c_uint32), ("s_numGlobalNonLinIterations", C.c_uint32),
        ("s_numGlobalLinIterations", C.c_uint32), ("s_downsampledWidth", C.c_uint32), ("s_downsampledHeight", C.c_uint32),
        ("s_verifySiftErrThresh", C.c_float), ("s_verifySiftCorrThresh", C.c_float), ("s_projCorrDistThres", C.c_float), ("s_projCorrNormalThres", C.c_float),
        ("s_projCorrColorThresh", C.c_float), ("s_surfAreaPcaThresh", C.c_float),
        ("s_recordSolverConvergence", C.c_int32), ("s_erodeSIFTdepth", C.c_int32),
        ("s_verifyOptErrThresh", C.c_float), ("s_verifyOptCorrThresh", C.c_float),
        ("s_verbose", C.c_int32), ("s_sendUplinkFeedbackImage", C.c_int32),
        ("s_depthSigmaD", C.c_float), ("s_depthSigmaR", C.c_float), ("s_depthFilter", C.c_int32),
        ("s_minNumMatchesLocal", C.c_uint32), ("s_minNumMatchesGlobal", C.c_uint32), ("s_useComprehensiveFrameInvalidation", C.c_int32),
        ("s_maxKabschResidual2", C.c_float), ("s_minKeyScale", C.c_float), ("s_siftMatchThresh", C.c_float), ("s_siftMatchRatioMaxLocal", C.c_float),
        ("s_siftMatchRatioMaxGlobal", C.c_float),
        ("s_useLocalVerify", C.c_int32), ("s_useLocalDense", C.c_int32), ("s_numOptPerResidualRemoval", C.c_uint32),
        ("s_colorDownSigma", C.c_float), ("s_depthDownSigmaD", C.c_float), ("s_depthDownSigmaR", C.c_float),
        ("s_optMaxResThresh", C.c_float), ("s_denseDistThresh", C.c_float), ("s_denseNormalThresh", C.c_float), ("s_denseColorThresh", C.c_float),
        ("s_denseColorGradientMin", C.c_float), ("s_denseDepthMin", C.c_float), ("s_denseDepthMax", C.c_float),
        ("s_denseOverlapCheckSubsampleFactor", C.c_uint32),
    ]


class RGBDSensorDesc(C.Structure):
    _fields_ = [("depthWidth", C.c_uint32), ("depthHeight", C.c_uint32), ("colorWidth", C.c_uint32), ("colorHeight", C.c_uint32),
                ("depthIntrinsics", C.c_float * 16), ("colorIntrinsics", C.c_float * 16), ("depthExtrinsics", C.c_float * 16), ("colorExtrinsics", C.c_float * 16)]


class FrameTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("timeSensorProcess", "timeSiftDetection", "timeSiftMatching", "timeMatchFilter", "timeSolve", "timeReIntegrate",
                                         "timeReconstruct", "timeTotal")]


def default_app_state(path=None):
    g = GlobalAppState()
    if path is None:
        check(lib.bf_global_app_state_default(C.byref(g)))
    else:
        check(lib.bf_global_app_state_read(path.encode(), C.byref(g), None))
    return g


def default_bundling_state(path=None):
    g = GlobalBundlingState()
    if path is None:
        check(lib.bf_global_bundling_state_default(C.byref(g)))
    else:
        check(lib.bf_global_bundling_state_read(path.encode(), C.byref(g), None))
    return g


def sensor_desc(width, height, K):
    """A sensor whose depth and colour cameras coincide (identity extrinsics), intrinsics K (4x4)."""
    s = RGBDSensorDesc()
    s.depthWidth = s.colorWidth = width
    s.depthHeight = s.colorHeight = height
    k = np.asarray(K, np.float32).reshape(16)
    eye = np.eye(4, dtype=np.float32).reshape(16)
    for i in range(16):
        s.depthIntrinsics[i] = s.colorIntrinsics[i] = float(k[i])
        s.depthExtrinsics[i] = s.colorExtrinsics[i] = float(eye[i])
    return s


_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class Comm:
    """Python view of `bf_comm` (include/bf_comm.h): the all-gather of the multi-GPU partition.

    Comm.rccl(world, rank, broadcast): RCCL, bootstrapped NCCL-style - rank 0 draws the unique id, `broadcast(bytes_or_None) -> bytes` hands it to every
    rank (e.g. through torch.distributed's store or a gloo broadcast).  Comm.torch_group(group): the all-gather through a torch.distributed process
    group (gloo: through host memory - how two ranks share the one GPU of a test box); collectives issued through one Comm must not interleave with
    other collectives on the same group from another thread - give the volume thread its own group (dist.new_group)."""

    def __init__(self):
        self._h = C.c_void_p()
        self._cb = None

    @classmethod
    def rccl(cls, world, rank, broadcast):
        c = cls()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            check(lib.bf_comm_unique_id(ident))
        raw = broadcast(bytes(ident) if rank == 0 else None)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        check(lib.bf_comm_create_rccl(ident, int(world), int(rank), C.byref(c._h)))
        return c

    @classmethod
    def torch_group(cls, group=None):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        nccl = dist.get_backend(group) == "nccl"

        def gather(user, d_send, d_recv, nbytes, stream):
            try:
                n = int(nbytes)
                check(lib.bf_stream_synchronize(C.c_void_p(stream)))
                send = torch.empty(n, dtype=torch.uint8, device="cuda" if nccl else "cpu")
                check(lib.bf_memcpy(C.c_void_p(send.data_ptr()), C.c_void_p(d_send), C.c_size_t(n)))
                if nccl:
                    out = torch.empty(world * n, dtype=torch.uint8, device="cuda")
                    dist.all_gather_into_tensor(out, send, group=group)
                    torch.cuda.synchronize()
                else:
                    parts = [torch.empty(n, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, send, group=group)
                    out = torch.cat(parts)
                check(lib.bf_memcpy(C.c_void_p(d_recv), C.c_void_p(out.data_ptr()), C.c_size_t(world * n)))
                return 0
            except Exception as e:          # never let an exception cross the C boundary
                import sys
                print("Comm.torch_group all-gather failed: %r" % (e,), file=sys.stderr)
                return 1
        c = cls()
        c._cb = _ALL_GATHER_FN(gather)
        check(lib.bf_comm_create_callback(c._cb, None, int(world), int(rank), C.byref(c._h)))
        return c

    def world(self):
        w, r = C.c_uint32(), C.c_uint32()
        check(lib.bf_comm_world(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def all_gather(self, send, recv, stream=None):
        """send / recv: torch cuda uint8 tensors (recv = world x send)"""
        check(lib.bf_comm_all_gather(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), C.c_uint64(send.numel()), C.c_void_p(stream)))

    def chunk_exchange(self, mine, stream=None):
        """One round of chunk packages: `mine` (uint8 numpy array) -> list of `world` arrays in owner order (bf_chunk_exchange)."""
        w, _ = self.world()
        mine = np.ascontiguousarray(mine, np.uint8)
        out = np.zeros(w * mine.size, np.uint8)
        check(lib.bf_chunk_exchange(self._h, mine.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint64(mine.size), C.c_void_p(stream)))
        return [np.ascontiguousarray(out[r * mine.size:(r + 1) * mine.size]) for r in range(w)]

    def close(self):
        if self._h:
            lib.bf_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """Python view of `bf_pipeline`: the serial frame loop of DepthSensing.cpp (ingest -> bundling input -> re-integration ->
    integration of the current frame -> local / global optimisation)."""

    def __init__(self, gas, gbs, sensor):
        self._h = C.c_void_p()
        self.gas, self.gbs, self.sensor = gas, gbs, sensor
        check(lib.bf_pipeline_create(C.byref(gas), C.byref(gbs), C.byref(sensor), C.byref(self._h)))

    def close(self):
        if self._h:
            lib.bf_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_volume_shard(self, rank, world):
        check(lib.bf_pipeline_set_volume_shard(self._h, rank, world))

    def set_volume_batching(self, enable=True):
        """one bf_scene_run_batch per frame (default) or one operator at a time: bf_pipeline_set_volume_batching"""
        check(lib.bf_pipeline_set_volume_batching(self._h, int(enable)))

    def set_comm(self, comm, capacity_keys=None):
        """Divide the allocation's ray march of every TSDF operator of this loop over the ranks of `comm` (bf_pipeline_set_comm; the volume must be
        sharded with set_volume_shard(rank, world) of the same communicator).  capacity_keys: keys per rank and operator; None sizes it from the integration
        resolution and the voxel size for ONE rank's share being the whole frame (alloc_comm_capacity with world = 1: never too small)."""
        if capacity_keys is None:
            capacity_keys = alloc_comm_capacity(self.gas.s_integrationWidth, self.gas.s_integrationHeight, self.gas.s_SDFVoxelSize, 1)
        check(lib.bf_pipeline_set_comm(self._h, comm._h if comm is not None else None, int(capacity_keys)))
        self._comm = comm

    def set_solve_lag(self, lag):
        """0: the reference's serial order (default).  1..s_submapSize: the chunk solves run on their own thread / stream and are applied exactly `lag`
        frames after the frame that closed the chunk (bf_pipeline_set_solve_lag)."""
        check(lib.bf_pipeline_set_solve_lag(self._h, int(lag)))

    def solve_lag(self):
        n = C.c_uint32()
        check(lib.bf_pipeline_get_solve_lag(self._h, C.byref(n)))
        return n.value

    def process_frame(self, depth, color):
        """depth float32 (H,W), color uint8 (H,W,4): host numpy arrays (PCIe path) or torch cuda tensors (HBM-resident path)."""
        got = C.c_int()
        if isinstance(depth, np.ndarray):
            depth = np.ascontiguousarray(depth, np.float32); color = np.ascontiguousarray(color, np.uint8)
            check(lib.bf_pipeline_process_frame(self._h, depth.ctypes.data_as(C.c_void_p), color.ctypes.data_as(C.c_void_p), C.byref(got)))
        else:
            check(lib.bf_pipeline_process_frame_device(self._h, C.c_void_p(depth.data_ptr()), C.c_void_p(color.data_ptr()), C.byref(got)))
        return bool(got.value)

    def process_end_of_sequence(self):
        n = C.c_uint32()
        check(lib.bf_pipeline_process_end_of_sequence(self._h, C.byref(n)))
        return n.value

    def synchronize(self):
        check(lib.bf_pipeline_synchronize(self._h))

    def num_frames(self):
        n = C.c_uint32()
        check(lib.bf_pipeline_get_num_frames(self._h, C.byref(n)))
        return n.value

    def integrated_trajectory(self):
        n = self.num_frames()
        out = np.zeros((max(n, 1), 4, 4), np.float32); cnt = C.c_uint32()
        check(lib.bf_pipeline_get_integrated_trajectory(self._h, out.ctypes.data_as(C.c_void_p), n, C.byref(cnt)))
        return out[:cnt.value]

    def optimized_trajectory(self):
        ob = C.c_void_p(); tm = C.c_void_p()
        check(lib.bf_pipeline_get_online_bundler(self._h, C.byref(ob)))
        check(lib.bf_online_bundler_get_trajectory_manager(ob, C.byref(tm)))
        n = self.num_frames()
        out = np.zeros((max(n, 1), 4, 4), np.float32); cnt = C.c_uint32()
        check(lib.bf_trajectory_manager_get_optimized_transforms(tm, out.ctypes.data_as(C.c_void_p), n, C.byref(cnt)))
        return out[:cnt.value]

    def counters(self):
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib.bf_pipeline_get_counters(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(integrate=a.value, deintegrate=b.value, local_solves=c.value, global_solves=d.value)

    def host_profile(self, reset=False):
        """Seconds the calling thread spent per part of the frame loop (accumulated) and the frame count: see bf_pipeline_get_host_profile."""
        out = (C.c_double * 8)()
        check(lib.bf_pipeline_get_host_profile(self._h, out, int(reset)))
        names = ("enqueue_match_chain", "ingest_detect_enqueue", "reintegrate_commands", "wait_match_result", "integrate_command", "solves", "wait_ingest", "frames")
        return dict(zip(names, [float(v) for v in out]))

    def volume_thread_profile(self, reset=False):
        b, n = C.c_double(), C.c_double()
        check(lib.bf_pipeline_get_volume_thread_profile(self._h, C.byref(b), C.byref(n), int(reset)))
        return dict(busy_seconds=b.value, operators=n.value)

    def enable_timings(self, on=True):
        check(lib.bf_pipeline_enable_timings(self._h, int(on)))

    def last_timing(self):
        t = FrameTiming()
        check(lib.bf_pipeline_get_last_timing(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in FrameTiming._fields_}

    def scene(self):
        """Borrowed SceneRepHashSDF view of the pipeline's voxel-hash volume."""
        h = C.c_void_p()
        check(lib.bf_pipeline_get_scene(self._h, C.byref(h)))
        s = SceneRepHashSDF.__new__(SceneRepHashSDF)
        s._h = h; s._borrowed = True
        hp = HashParams(); check(lib.bf_scene_get_hash_params(h, C.byref(hp))); s.params = hp
        return s

    def process_frame_chunked(self, depth, color, package, local_idx):
        """One iteration for the next frame of the stream with its chunk-local half taken from `package` (a ChunkWorker.run result,
        possibly of another rank): depth / colour are torch cuda tensors, package a contiguous uint8 numpy array."""
        got = C.c_int()
        check(lib.bf_pipeline_process_frame_chunked(self._h, C.c_void_p(depth.data_ptr()), C.c_void_p(color.data_ptr()),
                                                    package.ctypes.data_as(C.c_void_p), C.c_uint64(package.nbytes), int(local_idx), C.byref(got)))
        return bool(got.value)

    def integrate_frame_cpu(self, frame):
        """Host copies (depth float32 (h,w), colour uint8 (h,w,4)) of the frame stored for integration — getIntegrateFrame(i).getDepthFrameCPU() / getColorFrameCPU()."""
        im = C.c_void_p(); w = C.c_uint32(); h = C.c_uint32()
        check(lib.bf_pipeline_get_image_manager(self._h, C.byref(im)))
        check(lib.bf_image_manager_get_integration_size(im, C.byref(w), C.byref(h)))
        d = np.zeros((h.value, w.value), np.float32); c = np.zeros((h.value, w.value, 4), np.uint8)
        check(lib.bf_image_manager_get_integrate_frame_cpu(im, frame, d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
        return d, c

    def bundler(self, which):
        ob = C.c_void_p(); b = C.c_void_p()
        check(lib.bf_pipeline_get_online_bundler(self._h, C.byref(ob)))
        check(lib.bf_online_bundler_get_bundler(ob, {"local": 0, "optLocal": 1, "global": 2}[which], C.byref(b)))
        return b

    def initialize_correspondence_evaluator(self, complete_trajectory, log_prefix=None):
        """OnlineBundler.cpp:81-90: evaluate the global key-frame matches against a reference trajectory (one 4x4 per INPUT frame)."""
        ob = C.c_void_p()
        check(lib.bf_pipeline_get_online_bundler(self._h, C.byref(ob)))
        t = np.ascontiguousarray(complete_trajectory, np.float32).reshape(-1, 16)
        check(lib.bf_online_bundler_initialize_correspondence_evaluator(ob, t.ctypes.data_as(C.POINTER(C.c_float)), len(t), (log_prefix or "").encode()))

    def finish_correspondence_evaluator_logging(self):
        ob = C.c_void_p()
        check(lib.bf_pipeline_get_online_bundler(self._h, C.byref(ob)))
        check(lib.bf_online_bundler_finish_correspondence_evaluator_logging(ob))

    def correspondence_evaluator(self):
        h = C.c_void_p()
        check(lib.bf_bundler_get_correspondence_evaluator(self.bundler("global"), C.byref(h)))
        return CorrespondenceEvaluator(handle=h) if h else None


class ChunkWorker:
    """Python view of `bf_chunk_worker`: the chunk-local half of the frame loop (SIFT, matching + filters inside the chunk, local
    solve, key-frame fusion) for one local chunk at a time -> a flat package (numpy uint8) for Pipeline.process_frame_chunked."""

    def __init__(self, gas, gbs, sensor):
        self._h = C.c_void_p()
        check(lib.bf_chunk_worker_create(C.byref(gas), C.byref(gbs), C.byref(sensor), C.byref(self._h)))
        n = C.c_uint64()
        check(lib.bf_chunk_worker_package_bytes(self._h, C.byref(n)))
        self.package_bytes = int(n.value)

    def close(self):
        if self._h:
            lib.bf_chunk_worker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, chunk_index, frames, out=None):
        """frames: list of (depth, colour) torch cuda tensors, the chunk's frames in stream order (the first one is shared with the
        previous chunk).  Returns the package (uint8 numpy array of package_bytes)."""
        n = len(frames)
        dp = (C.c_void_p * n)(*[f[0].data_ptr() for f in frames])
        cp = (C.c_void_p * n)(*[f[1].data_ptr() for f in frames])
        pkg = out if out is not None else np.zeros(self.package_bytes, np.uint8)
        check(lib.bf_chunk_worker_run(self._h, int(chunk_index), n, dp, cp, pkg.ctypes.data_as(C.c_void_p)))
        return pkg


# --------------------------------------------------------------------------- marching cubes
class MarchingCubesParams(C.Structure):
    _fields_ = [("m_maxNumTriangles", C.c_uint32), ("m_sdfBlockSize", C.c_uint32), ("m_hashNumBuckets", C.c_uint32), ("m_hashBucketSize", C.c_uint32),
                ("m_threshMarchingCubes", C.c_float), ("m_threshMarchingCubes2", C.c_float)]


def marching_cubes_tables():
    """The generated case tables: (edgeTable[256] uint16, triTable[256,16] int8).  Host-only."""
    e = np.zeros(256, np.uint16); t = np.zeros(256 * 16, np.int8)
    check(lib.bf_marching_cubes_tables(e.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p)))
    return e, t.reshape(256, 16)


class MarchingCubesHashSDF:
    """Python view of `bf_marching_cubes` (== the reference's CUDAMarchingCubesHashSDF)."""

    def __init__(self, max_triangles, num_buckets, voxel_size, thresh_factor=10.0):
        p = MarchingCubesParams(max_triangles, SDF_BLOCK_SIZE, num_buckets, HASH_BUCKET_SIZE, thresh_factor * voxel_size, thresh_factor * voxel_size)
        self.params = p
        self._h = C.c_void_p()
        check(lib.bf_marching_cubes_create(C.byref(p), C.byref(self._h)))

    def close(self):
        if self._h:
            lib.bf_marching_cubes_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, scene, box=None):
        """extractIsoSurface over `scene` (a SceneRepHashSDF): returns (triangles [n,3,6] float32 = position + colour per vertex, number found)"""
        hd = scene.hash_data(); hp = scene.hash_params()
        mn = (C.c_float * 3)(*(box[0] if box else (0, 0, 0))); mx = (C.c_float * 3)(*(box[1] if box else (0, 0, 0)))
        check(lib.bf_marching_cubes_clear_mesh_buffer(self._h))
        check(lib.bf_marching_cubes_extract(self._h, C.byref(hd), C.byref(hp), mn, mx, int(box is not None)))
        n = C.c_uint32(); found = C.c_uint32()
        check(lib.bf_marching_cubes_get_triangles_gpu(self._h, None, C.byref(n), C.byref(found)))
        out = np.zeros((n.value, 3, 6), np.float32); cnt = C.c_uint32()
        check(lib.bf_marching_cubes_get_mesh(self._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(cnt)))
        return out, found.value

    def save_mesh(self, filename, transform=None):
        nv = C.c_uint32(); nf = C.c_uint32()
        T = mat16(transform) if transform is not None else None
        check(lib.bf_marching_cubes_save_mesh(self._h, str(filename).encode(), T, C.byref(nv), C.byref(nf)))
        return nv.value, nf.value


# --------------------------------------------------------------------------- CorrespondenceEvaluator
class CorrEvaluation(C.Structure):
    _fields_ = [("numCorrect", C.c_uint32), ("numDetected", C.c_uint32), ("numTotal", C.c_uint32)]

    def precision(self):
        lib.bf_corr_evaluation_get_precision.restype = C.c_float
        return float(lib.bf_corr_evaluation_get_precision(C.byref(self)))

    def recall(self):
        lib.bf_corr_evaluation_get_recall.restype = C.c_float
        return float(lib.bf_corr_evaluation_get_recall(C.byref(self)))

    def as_tuple(self):
        return (self.numCorrect, self.numDetected, self.numTotal)


class CorrEvalParams(C.Structure):
    _fields_ = [("depthMin", C.c_float), ("depthMax", C.c_float), ("distThresh", C.c_float), ("normalThresh", C.c_float), ("colorThresh", C.c_float)]


class CorrespondenceEvaluator:
    """Python view of `bf_correspondence_evaluator` (== class CorrespondenceEvaluator, CorrespondenceEvaluator.h:39): precision / recall
    of the current frame's image-to-image matches against a reference trajectory."""

    def __init__(self, trajectory=None, log_prefix=None, handle=None):
        self._borrowed = handle is not None
        if handle is not None:
            self._h = handle
            return
        t = np.ascontiguousarray(trajectory, np.float32).reshape(-1, 16)
        self._h = C.c_void_p()
        check(lib.bf_correspondence_evaluator_create(t.ctypes.data_as(C.POINTER(C.c_float)), len(t), (log_prefix or "").encode(), C.byref(self._h)))

    def close(self):
        if self._h and not self._borrowed:
            lib.bf_correspondence_evaluator_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, mgr, cache, sift_intrinsics_inv, filtered, recompute_cache, clear_cache, corr_type, params=None, stream=0):
        p = params or CorrEvalParams(0.5, 4.0, 0.15, 0.97, 0.1)          # zParametersBundlingDefault.txt:26-27,55-57
        out = CorrEvaluation()
        check(lib.bf_correspondence_evaluator_evaluate(self._h, mgr._h, cache._h, _f16(sift_intrinsics_inv), C.byref(p), int(filtered), int(recompute_cache),
                                                       int(clear_cache), corr_type.encode(), C.c_void_p(stream), C.byref(out)))
        return out

    def finish_logging(self):
        check(lib.bf_correspondence_evaluator_finish_logging_to_file(self._h))

    def total(self, corr_type):
        out = CorrEvaluation()
        check(lib.bf_correspondence_evaluator_get_total(self._h, corr_type.encode(), C.byref(out)))
        return out

    def overlap_counts(self, num_frames, with_flags=True):
        c = np.zeros((num_frames, 4), np.uint32); f = np.zeros(num_frames, np.uint8)
        check(lib.bf_correspondence_evaluator_get_overlap_counts(self._h, c.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p) if with_flags else None, num_frames))
        return c, f


# --------------------------------------------------------------------------- ray cast
class RayCastParams(C.Structure):
    _fields_ = [("m_viewMatrix", C.c_float * 16), ("m_viewMatrixInverse", C.c_float * 16),
                ("mx", C.c_float), ("my", C.c_float), ("fx", C.c_float), ("fy", C.c_float),
                ("m_width", C.c_uint32), ("m_height", C.c_uint32), ("m_numOccupiedSDFBlocks", C.c_uint32), ("m_maxNumVertices", C.c_uint32),
                ("m_splatMinimum", C.c_int32), ("m_minDepth", C.c_float), ("m_maxDepth", C.c_float),
                ("m_rayIncrement", C.c_float), ("m_thresSampleDist", C.c_float), ("m_thresDist", C.c_float),
                ("m_useGradients", C.c_int32), ("dummy0", C.c_uint32)]


class RayCastData(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_depth", "d_depth4", "d_normals", "d_colors", "d_rayIntervalSplatMin", "d_rayIntervalSplatMax")]


def write_processed_summary(path, heap_free_count, optimized_trajectory, aborted=False):
    """processed.txt of StopScanningAndExit (DepthSensing.cpp:921-957) -> the `valid` verdict"""
    T = np.ascontiguousarray(optimized_trajectory, np.float32).reshape(-1, 16)
    v = C.c_int()
    check(lib.bf_write_processed_summary(str(path).encode(), C.c_uint32(int(heap_free_count)), T.ctypes.data_as(C.c_void_p), C.c_uint32(len(T)), int(aborted), C.byref(v)))
    return bool(v.value)


def ray_cast_params_from_global_app_state(gas, intrinsics):
    """CUDARayCastSDF::parametersFromGlobalAppState (CUDARayCastSDF.h:24-52)"""
    p = RayCastParams()
    check(lib.bf_ray_cast_params_from_global_app_state(C.byref(gas), _f16(intrinsics), C.byref(p)))
    return p


class RayCastSDF:
    """Python view of `bf_ray_cast` (== class CUDARayCastSDF)."""

    def __init__(self, params, stream=None):
        self._h = C.c_void_p()
        check(lib.bf_ray_cast_create(C.byref(params), C.byref(self._h)))
        if stream is not None:
            check(lib.bf_ray_cast_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.bf_ray_cast_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def params(self):
        p = RayCastParams()
        check(lib.bf_ray_cast_get_params(self._h, C.byref(p)))
        return p

    def update_min_max(self, dmin, dmax):
        check(lib.bf_ray_cast_update_min_max(self._h, C.c_float(dmin), C.c_float(dmax)))

    def set_intrinsics(self, width, height, K):
        check(lib.bf_ray_cast_set_intrinsics(self._h, width, height, _f16(K)))

    def render(self, scene, cam, last_rigid_transform):
        """render(hashData, hashParams, lastRigidTransform) on a SceneRepHashSDF whose frustum list was made for `cam` (after an
        integrate, or scene.compactify(T, cam))."""
        hd = scene.hash_data(); hp = scene.hash_params()
        check(lib.bf_ray_cast_render(self._h, C.byref(hd), C.byref(hp), C.byref(cam), _f16(last_rigid_transform)))

    def convert_to_camera_space(self, cam):
        check(lib.bf_ray_cast_convert_to_camera_space(self._h, C.byref(cam)))

    def download(self):
        """dict of host images: depth (H,W), depth4 / normals / colors (H,W,4), ray_min / ray_max (H,W)"""
        import torch
        torch.cuda.synchronize()
        p = self.params(); d = RayCastData()
        check(lib.bf_ray_cast_get_data(self._h, C.byref(d)))
        n = p.m_width * p.m_height
        f1 = lambda ptr: _d2h(ptr, n * 4).view("<f4").reshape(p.m_height, p.m_width).copy()
        f4 = lambda ptr: _d2h(ptr, n * 16).view("<f4").reshape(p.m_height, p.m_width, 4).copy()
        return dict(depth=f1(d.d_depth), depth4=f4(d.d_depth4), normals=f4(d.d_normals), colors=f4(d.d_colors),
                    ray_min=f1(d.d_rayIntervalSplatMin), ray_max=f1(d.d_rayIntervalSplatMax))
