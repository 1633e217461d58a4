"""CPU tests of the .sens reader / writer (include/bf_sensordata.h).

The C ABI is checked against an independent struct.pack / struct.unpack restatement of the published SensorData
version-4 byte layout (written in this file), in both directions, plus the conversion rules of
SensorDataReader::processDepth (SensorDataReader.cpp:98-111): metres = u16 / depthShift, 0 -> -inf, RGB -> RGBX.
"""
import io
import struct
import zlib

import numpy as np
import pytest

from bundlefusion_amd import sensordata as sdm
from bundlefusion_amd.capi import BFError


def _K(fx=583.0, fy=584.0, cx=319.5, cy=239.5):
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    return K


def _frames(n, w, h, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        depth = rng.integers(400, 4000, size=(h, w)).astype(np.uint16)
        depth[rng.random((h, w)) < 0.1] = 0
        color = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = rng.normal(size=3)
        T[0, 1] = 0.01 * k
        out.append((T, depth, color, 1000 + k, 2000 + k))
    return out


def _py_write(path, frames, w, h, cw, ch, depth_comp, color_comp, shift, name=b"StructureSensor", imu=0, color_payload=None):
    """Independent writer: the SensorData v4 layout with struct.pack."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 4))
        f.write(struct.pack("<Q", len(name)) + name)
        for m in (_K(500, 501, 31.5, 23.5), np.eye(4), _K(), np.eye(4)):          # colour intr, colour extr, depth intr, depth extr
            f.write(np.asarray(m, "<f4").tobytes())
        f.write(struct.pack("<ii", color_comp, depth_comp))
        f.write(struct.pack("<IIII", cw, ch, w, h))
        f.write(struct.pack("<f", shift))
        f.write(struct.pack("<Q", len(frames)))
        for i, (T, depth, color, tc, td) in enumerate(frames):
            cbytes = color.tobytes() if color_payload is None else color_payload[i]
            dbytes = depth.astype("<u2").tobytes()
            if depth_comp == 1:
                dbytes = zlib.compress(dbytes)
            f.write(np.asarray(T, "<f4").tobytes())
            f.write(struct.pack("<QQQQ", tc, td, len(cbytes), len(dbytes)))
            f.write(cbytes)
            f.write(dbytes)
        f.write(struct.pack("<Q", imu))
        f.write(b"\x00" * (128 * imu))


def _py_read(path):
    """Independent reader (the layout the public ScanNet SensorData reader documents)."""
    with open(path, "rb") as f:
        ver, = struct.unpack("<I", f.read(4))
        n, = struct.unpack("<Q", f.read(8))
        name = f.read(n)
        mats = [np.frombuffer(f.read(64), "<f4").reshape(4, 4) for _ in range(4)]
        cc, dc = struct.unpack("<ii", f.read(8))
        cw, ch, w, h = struct.unpack("<IIII", f.read(16))
        shift, = struct.unpack("<f", f.read(4))
        nf, = struct.unpack("<Q", f.read(8))
        frames = []
        for _ in range(nf):
            T = np.frombuffer(f.read(64), "<f4").reshape(4, 4)
            tc, td, cs, ds = struct.unpack("<QQQQ", f.read(32))
            cb, db = f.read(cs), f.read(ds)
            frames.append((T, tc, td, cb, db))
        nimu, = struct.unpack("<Q", f.read(8))
        rest = f.read()
    return dict(version=ver, name=name, mats=mats, cc=cc, dc=dc, cw=cw, ch=ch, w=w, h=h, shift=shift, frames=frames, nimu=nimu, rest=rest)


@pytest.mark.parametrize("depth_comp", [sdm.DEPTH_RAW_USHORT, sdm.DEPTH_ZLIB_USHORT])
def test_reader_on_independently_written_file(tmp_path, depth_comp):
    w, h = 64, 48
    frames = _frames(5, w, h)
    path = tmp_path / "a.sens"
    _py_write(path, frames, w, h, w, h, depth_comp, sdm.COLOR_RAW, 1000.0, imu=3)
    sd = sdm.SensorData(path)
    assert len(sd) == 5 and sd.sensor_name == "StructureSensor"
    i = sd.info
    assert (i.versionNumber, i.depthWidth, i.depthHeight, i.colorWidth, i.colorHeight) == (4, w, h, w, h)
    assert i.depthShift == 1000.0 and i.numIMUFrames == 3
    assert i.depthCompressionType == depth_comp and i.colorCompressionType == sdm.COLOR_RAW
    assert np.array_equal(np.array(i.depthIntrinsic, np.float32).reshape(4, 4), _K())
    assert np.array_equal(np.array(i.colorIntrinsic, np.float32).reshape(4, 4), _K(500, 501, 31.5, 23.5))
    for k, (T, depth, color, tc, td) in enumerate(frames):
        Tk, a, b = sd.pose(k)
        assert np.array_equal(Tk, T) and (a, b) == (tc, td)
        assert np.array_equal(sd.depth_raw(k), depth)
        d = sd.depth(k)
        assert np.array_equal(np.isneginf(d), depth == 0)                                   # 0 -> -inf  (SensorDataReader.cpp:99)
        assert np.array_equal(d[depth != 0], depth[depth != 0].astype(np.float32) / np.float32(1000.0))   # (float)u16 / m_depthShift (:100)
        c = sd.color_rgbx(k)
        assert np.array_equal(c[..., :3], color) and (c[..., 3] == 255).all()               # vec4uc(vec3uc) (:109)
        assert sd.color_compressed(k) == color.tobytes()
    with pytest.raises(BFError):
        sd.pose(5)
    sd.close()


def test_writer_output_parses_with_the_independent_reader(tmp_path):
    w, h = 40, 30
    frames = _frames(4, w, h, seed=3)
    path = tmp_path / "w.sens"
    with sdm.SensorDataWriter(path, (w, h), (w, h), _K(), color_intrinsic=_K(500, 501, 31.5, 23.5), depth_shift=1000.0, sensor_name="synthetic S2") as wr:
        for T, depth, color, tc, td in frames:
            wr.add_frame(T, depth, color.tobytes(), tc, td)
    r = _py_read(path)
    assert r["version"] == 4 and r["name"] == b"synthetic S2" and r["nimu"] == 0 and r["rest"] == b""
    assert (r["cc"], r["dc"], r["cw"], r["ch"], r["w"], r["h"], r["shift"]) == (sdm.COLOR_RAW, sdm.DEPTH_ZLIB_USHORT, w, h, w, h, 1000.0)
    assert np.array_equal(r["mats"][0], _K(500, 501, 31.5, 23.5)) and np.array_equal(r["mats"][2], _K())
    assert np.array_equal(r["mats"][1], np.eye(4)) and np.array_equal(r["mats"][3], np.eye(4))
    assert len(r["frames"]) == 4
    for (T, depth, color, tc, td), (T2, tc2, td2, cb, db) in zip(frames, r["frames"]):
        assert np.array_equal(T, T2) and (tc, td) == (tc2, td2)
        assert cb == color.tobytes()
        assert np.array_equal(np.frombuffer(zlib.decompress(db), "<u2").reshape(h, w), depth)
    # and back through the C reader
    sd = sdm.SensorData(path)
    assert np.array_equal(sd.depth_raw(2), frames[2][1])
    sd.close()


def test_jpeg_colour_through_the_decoder_callback(tmp_path):
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    w, h = 64, 48
    frames = _frames(2, w, h, seed=5)
    payload = []
    for _, _, color, _, _ in frames:
        smooth = np.asarray(Image.fromarray(color).resize((8, 6)).resize((w, h), Image.BILINEAR))
        buf = io.BytesIO()
        Image.fromarray(smooth).save(buf, format="JPEG", quality=92)
        payload.append(buf.getvalue())
    path = tmp_path / "j.sens"
    _py_write(path, frames, w, h, w, h, 1, sdm.COLOR_JPEG, 1000.0, color_payload=payload)
    sd = sdm.SensorData(path)
    for k in range(2):
        ref = np.asarray(Image.open(io.BytesIO(payload[k])).convert("RGB"))
        c = sd.color_rgbx(k)
        assert np.array_equal(c[..., :3], ref) and (c[..., 3] == 255).all()
        assert sd.color_compressed(k) == payload[k]
        assert sd.frame_sizes(k)[0] == len(payload[k])
    sd.close()
    sd = sdm.SensorData(path, use_pillow=False)              # no callback installed: the library's own baseline decoder is used
    for k in range(2):
        ref = np.asarray(Image.open(io.BytesIO(payload[k])).convert("RGB"))
        assert np.abs(sd.color_rgbx(k)[..., :3].astype(int) - ref.astype(int)).max() <= 1
    sd.close()


def test_sensor_desc_follows_create_first_connected(tmp_path):
    w, h = 32, 24
    frames = _frames(1, w, h)
    path = tmp_path / "d.sens"
    _py_write(path, [(frames[0][0], frames[0][1], np.zeros((0,), np.uint8), 0, 0)], w, h, 0, 0, 0, sdm.COLOR_RAW, 1000.0)
    sd = sdm.SensorData(path)
    d = sd.sensor_desc()
    assert (d.depthWidth, d.depthHeight, d.colorWidth, d.colorHeight) == (w, h, 1, 1)        # std::max(colour size, 1u) (:57)
    K = np.array(d.depthIntrinsics, np.float32).reshape(4, 4)
    assert np.array_equal(K, _K())                                                             # rebuilt from fx, fy, mx, my (RGBDSensor.cpp:142-147)
    assert np.array_equal(np.array(d.depthExtrinsics, np.float32).reshape(4, 4), np.eye(4))
    assert not sd.color_rgbx(0).any()                                                          # no colour data: m_colorRGBX stays untouched
    sd.close()


def test_malformed_files_are_rejected(tmp_path):
    w, h = 16, 12
    frames = _frames(2, w, h)
    good = tmp_path / "g.sens"
    _py_write(good, frames, w, h, w, h, 1, sdm.COLOR_RAW, 1000.0)
    blob = good.read_bytes()
    bad_version = tmp_path / "v.sens"
    bad_version.write_bytes(struct.pack("<I", 3) + blob[4:])
    with pytest.raises(BFError, match="version"):
        sdm.SensorData(bad_version)
    truncated = tmp_path / "t.sens"
    truncated.write_bytes(blob[: len(blob) - 200])
    with pytest.raises(BFError, match="truncated"):
        sdm.SensorData(truncated)
    with pytest.raises(BFError, match="could not open"):
        sdm.SensorData(tmp_path / "missing.sens")
    occi = tmp_path / "o.sens"
    _py_write(occi, frames, w, h, w, h, sdm.DEPTH_OCCI_USHORT, sdm.COLOR_RAW, 1000.0)
    sd = sdm.SensorData(occi)
    with pytest.raises(BFError, match="not supported"):
        sd.depth(0)
    sd.close()
    # header fields that size allocations: a frame / name count with the top bit set, absurd image sizes, depthShift 0
    name_len = struct.unpack("<Q", blob[4:12])[0]
    off_frames = 4 + 8 + name_len + 4 * 64 + 8 + 16 + 4
    for patch_off, patch in ((off_frames, struct.pack("<Q", 0xFFFFFFFFFFFFFFF0)), (4, struct.pack("<Q", 0x8000000000000010)),
                             (off_frames - 12, struct.pack("<I", 0x7FFFFFFF)), (off_frames - 4, struct.pack("<f", 0.0))):
        evil = tmp_path / "e.sens"
        evil.write_bytes(blob[:patch_off] + patch + blob[patch_off + len(patch):])
        with pytest.raises(BFError):
            sdm.SensorData(evil)
    wrong = tmp_path / "s.sens"                          # raw depth of the wrong size
    fr = [(frames[0][0], frames[0][1][:6], frames[0][2], 0, 0)]
    _py_write(wrong, fr, w, h, w, h, 0, sdm.COLOR_RAW, 1000.0)
    sd = sdm.SensorData(wrong)
    with pytest.raises(BFError, match="raw depth"):
        sd.depth_raw(0)
    sd.close()


def test_depth_quantisation_helper():
    d = np.array([[0.4004, np.inf, -np.inf, 0.0, 70.0, 1.2345678]], np.float32)
    q = sdm.depth_to_u16(d, 1000.0)
    assert q.tolist() == [[400, 0, 0, 0, 65535, 1235]]


# ---------------------------------------------------------------------------------------- trajectory evaluation (PoseHelper.h:35-79)
def _rigid(rng, angle=1.0, trans=1.0):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = angle * rng.uniform(0.2, 1.0)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = trans * rng.normal(size=3)
    return T


def _np_ate(traj, ref):
    """independent restatement: numpy SVD Kabsch on the camera positions, poses with -inf skipped"""
    ok = np.isfinite(traj[:, 0, 0]) & np.isfinite(ref[:, 0, 0])
    p, r = traj[ok][:, :3, 3].astype(np.float64), ref[ok][:, :3, 3].astype(np.float64)
    pc, rc = p.mean(0), r.mean(0)
    H = (p - pc).T @ (r - rc)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    t = rc - R @ pc
    e = (R @ p.T).T + t - r
    return float(np.sqrt((e ** 2).sum(1).mean())), int(ok.sum())


def test_ate_rmse_matches_numpy_kabsch():
    rng = np.random.default_rng(7)
    for n, noise, planar in ((50, 0.0, False), (200, 0.02, False), (30, 0.01, True), (3, 0.05, False)):
        ref = np.stack([_rigid(rng) for _ in range(n)]).astype(np.float32)
        if planar:
            ref[:, 2, 3] = 0.25
        G = _rigid(rng, 2.0, 3.0)
        traj = np.stack([G @ T for T in ref.astype(np.float64)])
        traj[:, :3, 3] += noise * rng.normal(size=(n, 3))
        traj = traj.astype(np.float32)
        if n > 10:
            traj[5] = -np.inf; ref[9] = -np.inf                     # invalid poses are skipped on either side (:52)
        got, num = sdm.ate_rmse(traj, ref)
        want, wnum = _np_ate(traj, ref)
        assert num == wnum
        assert abs(got - want) < 2e-6 + 1e-4 * want, (n, got, want)
        if noise == 0.0:
            assert got < 1e-5
    # a mirrored point set must not be "aligned" by a reflection
    ref = np.stack([_rigid(rng) for _ in range(40)]).astype(np.float32)
    mir = ref.copy(); mir[:, 0, 3] *= -1
    got, _ = sdm.ate_rmse(mir, ref)
    want, _ = _np_ate(mir, ref)
    assert abs(got - want) < 1e-5 and got > 0.1
    # fewer than three transforms (:37-47)
    assert sdm.ate_rmse(ref[:1], ref[:1]) == (-np.inf, 1)
    two_ref = np.stack([np.eye(4), _rigid(rng)]).astype(np.float32)
    two = two_ref.copy(); two[1, :3, 3] += [0.03, 0.0, 0.04]
    got, num = sdm.ate_rmse(two, two_ref)
    assert num == 2 and abs(got - 0.05) < 1e-6
    two_ref[0, 0, 3] = 0.5                                           # "cannot evaluate 2 with reference[0] not identity"
    assert sdm.ate_rmse(two, two_ref)[0] == -np.inf
    allbad = ref.copy(); allbad[:] = -np.inf
    assert sdm.ate_rmse(allbad, ref) == (-np.inf, 0)


def test_save_with_trajectory_and_evaluate(tmp_path):
    w, h = 24, 16
    rng = np.random.default_rng(11)
    frames = _frames(6, w, h, seed=2)
    poses = [_rigid(rng).astype(np.float32) for _ in range(6)]
    frames = [(poses[k],) + frames[k][1:] for k in range(6)]
    src = tmp_path / "src.sens"
    _py_write(src, frames, w, h, w, h, 1, sdm.COLOR_RAW, 1000.0, imu=2)
    sd = sdm.SensorData(src)
    assert np.array_equal(sd.trajectory(), np.stack(poses))
    # the stored poses, re-based to identity, evaluate to zero against themselves re-based (SensorDataReader.cpp:172-175)
    base = np.linalg.inv(poses[0].astype(np.float64))
    rebased = np.stack([(base @ p.astype(np.float64)) for p in poses]).astype(np.float32)
    rmse, n = sd.evaluate_trajectory(rebased)
    assert n == 6 and rmse < 1e-5
    moved = rebased.copy(); moved[:, :3, 3] += rng.normal(scale=0.01, size=(6, 3)).astype(np.float32)
    want, _ = _np_ate(moved, rebased)
    assert abs(sd.evaluate_trajectory(moved)[0] - want) < 1e-5
    # saveToFile(filename, trajectory) with a shorter trajectory: the rest becomes -inf, payloads are untouched
    out = tmp_path / "out.sens"
    sd.save_with_trajectory(out, moved[:4])
    r = _py_read(out)
    assert len(r["frames"]) == 6 and r["nimu"] == 0
    for k in range(6):
        T2, tc, td, cb, db = r["frames"][k]
        assert np.array_equal(T2, moved[k]) if k < 4 else np.isneginf(T2).all()
        assert (tc, td) == (frames[k][3], frames[k][4]) and cb == frames[k][2].tobytes()
        assert np.array_equal(np.frombuffer(zlib.decompress(db), "<u2").reshape(h, w), frames[k][1])
    sd.close()


# ---------------------------------------------------------------------------------------- built-in colour decoders (csrc/imagecodec.cpp)
def _decode(blob, ctype, w, h):
    import ctypes as C
    from bundlefusion_amd.capi import lib, check
    lib.bf_decode_color_rgb.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.empty((h, w, 3), np.uint8)
    buf = np.frombuffer(blob, np.uint8)
    check(lib.bf_decode_color_rgb(buf.ctypes.data, len(blob), ctype, w, h, out.ctypes.data))
    return out


def _test_image(w, h, kind, rng):
    if kind == "smooth":
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([127 + 100 * np.sin(x / 17.0 + y / 29.0), 127 + 90 * np.cos(x / 11.0), 100 + 80 * np.sin(y / 7.0)], -1)
        return np.clip(a, 0, 255).astype(np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a = np.zeros((h, w, 3), np.uint8)
    a[:, w // 2:] = [255, 0, 0]; a[h // 3:, : w // 3] = [0, 0, 255]; a[::7] = [0, 255, 0]
    return a


def test_builtin_jpeg_decoder_against_libjpeg():
    """Baseline JPEG: same inverse DCT / chroma interpolation / colour conversion as the IJG decoder Pillow links, so the
    result is expected to be identical; one LSB of slack is allowed for other libjpeg builds.  (mLib decodes with stb_image;
    lossy decoders are not pinned by the reference.)"""
    pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(0)
    for (w, h) in ((64, 48), (320, 240), (37, 29), (17, 9)):
        for kind in ("smooth", "noise", "edges"):
            img = _test_image(w, h, kind, rng)
            for sub in (0, 1, 2):                                   # 4:4:4, 4:2:2, 4:2:0
                for q, extra in ((50, {}), (92, {}), (85, {"restart_marker_blocks": 3})):
                    buf = io.BytesIO()
                    try:
                        Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=sub, **extra)
                    except TypeError:                               # an older Pillow without restart markers
                        continue
                    ref = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
                    got = _decode(buf.getvalue(), sdm.COLOR_JPEG, w, h)
                    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1, (w, h, kind, sub, q, extra)
    g = Image.fromarray(_test_image(64, 48, "smooth", rng)).convert("L")
    buf = io.BytesIO(); g.save(buf, format="JPEG", quality=90)
    assert np.array_equal(_decode(buf.getvalue(), sdm.COLOR_JPEG, 64, 48), np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    buf = io.BytesIO(); Image.fromarray(_test_image(64, 48, "smooth", rng)).save(buf, format="JPEG", progressive=True)
    with pytest.raises(BFError, match="progressive"):
        _decode(buf.getvalue(), sdm.COLOR_JPEG, 64, 48)
    with pytest.raises(BFError, match="expected"):                  # the container's size wins over the stream's
        _decode(buf.getvalue().replace(b"\xff\xc2", b"\xff\xc0", 1), sdm.COLOR_JPEG, 32, 48)
    with pytest.raises(BFError):
        _decode(b"\xff\xd8\xff\xd9", sdm.COLOR_JPEG, 8, 8)


def test_builtin_png_decoder_is_exact():
    pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(1)
    for mode in ("RGB", "RGBA", "L", "LA", "P", "1"):
        for kind in ("smooth", "noise", "edges"):
            im = Image.fromarray(_test_image(61, 43, kind, rng))
            im = im.convert(mode) if mode != "P" else im.convert("P", palette=Image.ADAPTIVE)
            for opt in (False, True):
                buf = io.BytesIO(); im.save(buf, format="PNG", optimize=opt)
                ref = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
                assert np.array_equal(_decode(buf.getvalue(), sdm.COLOR_PNG, 61, 43), ref), (mode, kind, opt)
    with pytest.raises(BFError, match="signature"):
        _decode(b"not a png at all", sdm.COLOR_PNG, 4, 4)


def test_builtin_jpeg_encoder_round_trip():
    """bf_encode_jpeg_rgb (recording): a stream every baseline decoder reads - Pillow and the built-in decoder agree on it bit for bit -
    at the quality / size of an optimised 4:4:4 libjpeg encode."""
    pytest.importorskip("PIL")
    import ctypes as C
    from PIL import Image
    from bundlefusion_amd.capi import lib, check
    lib.bf_encode_jpeg_rgb.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]

    def enc(img, q):
        h, w, _ = img.shape
        a = np.ascontiguousarray(img)
        n = C.c_uint64()
        check(lib.bf_encode_jpeg_rgb(a.ctypes.data, w, h, q, None, 0, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        check(lib.bf_encode_jpeg_rgb(a.ctypes.data, w, h, q, buf.ctypes.data, n.value, C.byref(n)))
        return buf.tobytes()

    def psnr(a, b):
        return 10 * np.log10(255.0 ** 2 / max(np.mean((a.astype(float) - b.astype(float)) ** 2), 1e-9))

    rng = np.random.default_rng(2)
    for (w, h) in ((64, 48), (37, 29), (200, 120)):
        for kind, floor50, floor90 in (("smooth", 36.0, 43.0), ("edges", 27.0, 33.0)):
            img = _test_image(w, h, kind, rng)
            for q, floor in ((50, floor50), (90, floor90)):
                blob = enc(img, q)
                assert blob[:2] == b"\xff\xd8" and blob[-2:] == b"\xff\xd9"
                pil = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
                assert np.array_equal(_decode(blob, sdm.COLOR_JPEG, w, h), pil)
                assert psnr(pil, img) > floor, (w, h, kind, q, psnr(pil, img))
                ref = io.BytesIO(); Image.fromarray(img).save(ref, format="JPEG", quality=q, subsampling=0)
                assert len(blob) < 1.1 * len(ref.getvalue()) + 64                 # image-optimised Huffman tables: not larger than libjpeg's default
    with pytest.raises(BFError):
        lib.bf_encode_jpeg_rgb.restype = C.c_int
        n = C.c_uint64()
        check(lib.bf_encode_jpeg_rgb(None, 8, 8, 90, None, 0, C.byref(n)))
