"""Write the synthetic S2 stream as a .sens file (the "config 2 stand-in" of SURVEY.md 8d: depth as u16 with depthShift 1000,
colour RGB8 raw or JPEG, ground-truth camera-to-world per frame).  Host only.

usage: python tools/make_sens.py out.sens [--frames 200] [--width 640 --height 480] [--jpeg 90] [--bob 0.0]
"""
import argparse
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bundlefusion_amd import synth
from bundlefusion_amd import sensordata as sdm
from bundlefusion_amd.capi import intrinsics_matrix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--bob", type=float, default=0.0)
    ap.add_argument("--jpeg", type=int, default=0, help="JPEG quality for the colour frames (0: raw RGB8)")
    ap.add_argument("--raw-depth", action="store_true", help="store depth uncompressed instead of zlib")
    a = ap.parse_args()
    W, H = a.width, a.height
    writer = None
    for c0 in range(0, a.frames, 256):
        idx = [a.first + k for k in range(c0, min(a.frames, c0 + 256))]
        for depth, color, T, Kd in synth.render_frames(idx, W, H, bob=a.bob):
            if writer is None:
                K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
                writer = sdm.SensorDataWriter(a.out, (W, H), (W, H), K, depth_shift=1000.0, sensor_name="synthetic S2 room",
                                              depth_compression=sdm.DEPTH_RAW_USHORT if a.raw_depth else sdm.DEPTH_ZLIB_USHORT,
                                              color_compression=sdm.COLOR_JPEG if a.jpeg else sdm.COLOR_RAW)
            rgb = np.ascontiguousarray(color.reshape(H, W, 4)[..., :3])
            if a.jpeg:
                from PIL import Image
                buf = io.BytesIO()
                Image.fromarray(rgb).save(buf, format="JPEG", quality=a.jpeg)
                cbytes = buf.getvalue()
            else:
                cbytes = rgb.tobytes()
            writer.add_frame(T, sdm.depth_to_u16(depth.reshape(H, W), 1000.0), cbytes)
    writer.close()
    print("wrote %s: %d frames %dx%d, %.1f MB" % (a.out, a.frames, W, H, os.path.getsize(a.out) / 1e6))


if __name__ == "__main__":
    main()
