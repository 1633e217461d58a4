"""ctypes view of include/bf_sensordata.h: recorded RGB-D sequences (".sens", ml::SensorData version 4).

Host-only.  JPEG / PNG colour frames are decoded with Pillow through the C ABI's decoder callback when Pillow is
importable (the reference decodes them with stb_image inside mLib); raw colour and raw / zlib depth need nothing.
"""
import ctypes as C
import io

import numpy as np

from .capi import lib, check, RGBDSensorDesc

COLOR_RAW, COLOR_PNG, COLOR_JPEG = 0, 1, 2
DEPTH_RAW_USHORT, DEPTH_ZLIB_USHORT, DEPTH_OCCI_USHORT = 0, 1, 2


class SensorDataInfo(C.Structure):
    _fields_ = [
        ("versionNumber", C.c_uint32),
        ("sensorName", C.c_char * 256),
        ("colorIntrinsic", C.c_float * 16), ("colorExtrinsic", C.c_float * 16),
        ("depthIntrinsic", C.c_float * 16), ("depthExtrinsic", C.c_float * 16),
        ("colorCompressionType", C.c_int32), ("depthCompressionType", C.c_int32),
        ("colorWidth", C.c_uint32), ("colorHeight", C.c_uint32), ("depthWidth", C.c_uint32), ("depthHeight", C.c_uint32),
        ("depthShift", C.c_float),
        ("numFrames", C.c_uint64), ("numIMUFrames", C.c_uint64),
    ]


_DECODER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8))

lib.bf_sensor_data_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
lib.bf_sensor_data_close.argtypes = [C.c_void_p]
lib.bf_sensor_data_get_info.argtypes = [C.c_void_p, C.POINTER(SensorDataInfo)]
lib.bf_sensor_data_get_sensor_desc.argtypes = [C.c_void_p, C.POINTER(RGBDSensorDesc)]
lib.bf_sensor_data_set_color_decoder.argtypes = [C.c_void_p, _DECODER, C.c_void_p]
lib.bf_sensor_data_get_frame_pose.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.bf_sensor_data_get_frame_sizes.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.bf_sensor_data_read_depth_raw.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
lib.bf_sensor_data_read_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
lib.bf_sensor_data_read_color_compressed.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
lib.bf_sensor_data_read_color_rgbx.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
lib.bf_sensor_data_writer_create.argtypes = [C.c_char_p, C.POINTER(SensorDataInfo), C.POINTER(C.c_void_p)]
lib.bf_sensor_data_writer_add_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
lib.bf_sensor_data_writer_close.argtypes = [C.c_void_p]
lib.bf_sensor_data_save_with_trajectory.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
lib.bf_evaluate_ate_rmse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
lib.bf_sensor_data_evaluate_trajectory.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]


def _pillow_decode(_user, data, size, _ctype, width, height, out):
    try:
        from PIL import Image
        img = Image.open(io.BytesIO(C.string_at(data, size))).convert("RGB")
        if img.size != (width, height):
            return 1
        C.memmove(out, img.tobytes(), width * height * 3)
        return 0
    except Exception:           # the C side turns a non-zero return into "the colour decoder failed"
        return 1


class SensorData:
    """A .sens file opened for reading (SensorData::loadFromFile + the accessors SensorDataReader uses)."""

    def __init__(self, filename, use_pillow=True):
        self._h = C.c_void_p()
        check(lib.bf_sensor_data_open(str(filename).encode(), C.byref(self._h)))
        self.info = SensorDataInfo()
        check(lib.bf_sensor_data_get_info(self._h, C.byref(self.info)))
        self._cb = None
        if use_pillow and self.info.colorCompressionType in (COLOR_PNG, COLOR_JPEG):
            try:
                import PIL  # noqa: F401
                self._cb = _DECODER(_pillow_decode)
                check(lib.bf_sensor_data_set_color_decoder(self._h, self._cb, None))
            except ImportError:
                pass

    def close(self):
        if self._h:
            lib.bf_sensor_data_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.info.numFrames)

    @property
    def sensor_name(self):
        return self.info.sensorName.decode()

    def sensor_desc(self):
        d = RGBDSensorDesc()
        check(lib.bf_sensor_data_get_sensor_desc(self._h, C.byref(d)))
        return d

    def pose(self, i):
        T = (C.c_float * 16)()
        tc, td = C.c_uint64(), C.c_uint64()
        check(lib.bf_sensor_data_get_frame_pose(self._h, i, T, C.byref(tc), C.byref(td)))
        return np.array(T, np.float32).reshape(4, 4), tc.value, td.value

    def frame_sizes(self, i):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib.bf_sensor_data_get_frame_sizes(self._h, i, C.byref(a), C.byref(b)))
        return a.value, b.value

    def depth_raw(self, i):
        out = np.empty((self.info.depthHeight, self.info.depthWidth), np.uint16)
        check(lib.bf_sensor_data_read_depth_raw(self._h, i, out.ctypes.data))
        return out

    def depth(self, i):
        """float32 metres, -inf where the stored value is 0 (SensorDataReader::processDepth)."""
        out = np.empty((self.info.depthHeight, self.info.depthWidth), np.float32)
        check(lib.bf_sensor_data_read_depth(self._h, i, out.ctypes.data))
        return out

    def color_compressed(self, i):
        n = C.c_uint64()
        check(lib.bf_sensor_data_read_color_compressed(self._h, i, None, 0, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        check(lib.bf_sensor_data_read_color_compressed(self._h, i, buf.ctypes.data, n.value, C.byref(n)))
        return buf.tobytes()

    def color_rgbx(self, i):
        out = np.empty((max(self.info.colorHeight, 1), max(self.info.colorWidth, 1), 4), np.uint8)
        check(lib.bf_sensor_data_read_color_rgbx(self._h, i, out.ctypes.data))
        return out

    def trajectory(self):
        """cameraToWorld of every frame (SensorDataReader::getTrajectory)."""
        return np.stack([self.pose(i)[0] for i in range(len(self))]) if len(self) else np.zeros((0, 4, 4), np.float32)

    def save_with_trajectory(self, filename, trajectory):
        """SensorDataReader::saveToFile(filename, trajectory): frames beyond the trajectory get an all -inf pose."""
        T = np.ascontiguousarray(trajectory, np.float32).reshape(-1, 16)
        check(lib.bf_sensor_data_save_with_trajectory(self._h, str(filename).encode(), T.ctypes.data if len(T) else None, len(T)))

    def evaluate_trajectory(self, trajectory):
        """SensorDataReader::evaluateTrajectory: (ATE RMSE [m], number of poses used) against the stored poses re-based to identity."""
        T = np.ascontiguousarray(trajectory, np.float32).reshape(-1, 16)
        rmse, n = C.c_float(), C.c_uint32()
        check(lib.bf_sensor_data_evaluate_trajectory(self._h, T.ctypes.data, len(T), C.byref(rmse), C.byref(n)))
        return rmse.value, n.value


class SensorDataWriter:
    """SensorData::saveToFile, frame by frame."""

    def __init__(self, filename, depth_size, color_size, depth_intrinsic, color_intrinsic=None, depth_shift=1000.0, sensor_name="synthetic",
                 depth_compression=DEPTH_ZLIB_USHORT, color_compression=COLOR_RAW, depth_extrinsic=None, color_extrinsic=None):
        info = SensorDataInfo()
        info.versionNumber = 4
        info.sensorName = sensor_name.encode()[:255]
        eye = np.eye(4, dtype=np.float32)
        for name, m in (("depthIntrinsic", depth_intrinsic), ("colorIntrinsic", depth_intrinsic if color_intrinsic is None else color_intrinsic),
                        ("depthExtrinsic", eye if depth_extrinsic is None else depth_extrinsic),
                        ("colorExtrinsic", eye if color_extrinsic is None else color_extrinsic)):
            setattr(info, name, (C.c_float * 16)(*np.asarray(m, np.float32).reshape(16)))
        info.depthWidth, info.depthHeight = depth_size
        info.colorWidth, info.colorHeight = color_size
        info.depthShift = depth_shift
        info.depthCompressionType = depth_compression
        info.colorCompressionType = color_compression
        self.info = info
        self._h = C.c_void_p()
        check(lib.bf_sensor_data_writer_create(str(filename).encode(), C.byref(info), C.byref(self._h)))

    def add_frame(self, camera_to_world, depth_u16, color_bytes=b"", ts_color=0, ts_depth=0):
        d = np.ascontiguousarray(depth_u16, np.uint16)
        assert d.size == self.info.depthWidth * self.info.depthHeight
        T = (C.c_float * 16)(*np.asarray(camera_to_world, np.float32).reshape(16))
        cb = bytes(color_bytes)
        check(lib.bf_sensor_data_writer_add_frame(self._h, T, ts_color, ts_depth, cb if cb else None, len(cb), d.ctypes.data))

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            check(lib.bf_sensor_data_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def depth_to_u16(depth_m, depth_shift=1000.0):
    """metres (non-finite / non-positive = invalid) -> the stored u16 (0 = invalid), rounding to nearest like a sensor would."""
    d = np.asarray(depth_m, np.float32)
    ok = np.isfinite(d) & (d > 0)
    q = np.zeros(d.shape, np.uint16)
    q[ok] = np.clip(np.rint(d[ok] * depth_shift), 1, 65535).astype(np.uint16)
    return q


def ate_rmse(trajectory, reference):
    """PoseHelper::evaluateAteRmse: (RMSE of the camera positions after a rigid alignment, number of poses used)."""
    A = np.ascontiguousarray(trajectory, np.float32).reshape(-1, 16)
    B = np.ascontiguousarray(reference, np.float32).reshape(-1, 16)
    n = min(len(A), len(B))
    rmse, num = C.c_float(), C.c_uint32()
    check(lib.bf_evaluate_ate_rmse(A.ctypes.data if n else None, B.ctypes.data if n else None, n, C.byref(rmse), C.byref(num)))
    return rmse.value, num.value
