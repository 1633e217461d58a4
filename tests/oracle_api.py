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
