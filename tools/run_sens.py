"""Play a recorded .sens file through the headless BundleFusion frame loop (s_sensorIdx = 8 in the reference) and evaluate the
optimised trajectory against the poses stored in the file (SensorDataReader::evaluateTrajectory).

usage: python tools/run_sens.py sequence.sens [--voxel 0.01] [--app zParametersDefault.txt] [--bundling zParametersBundlingDefault.txt]
Frames are decoded on the host and handed over as host buffers (the PCIe path of bf_pipeline_process_frame).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bundlefusion_amd as bf
from bundlefusion_amd import sensordata as sdm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sens")
    ap.add_argument("--app", default=None, help="zParametersDefault.txt (defaults: the shipped values)")
    ap.add_argument("--bundling", default=None, help="zParametersBundlingDefault.txt")
    ap.add_argument("--voxel", type=float, default=None)
    ap.add_argument("--buckets", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--frames", type=int, default=0, help="play only the first N frames")
    ap.add_argument("--tail", type=int, default=5, help="end-of-sequence iterations")
    ap.add_argument("--save", default=None, help="write a copy of the file with the optimised trajectory (SensorDataReader::saveToFile)")
    a = ap.parse_args()
    sd = sdm.SensorData(a.sens)
    n = len(sd) if a.frames <= 0 else min(a.frames, len(sd))
    desc = sd.sensor_desc()
    gas = bf.capi.default_app_state(a.app)
    gbs = bf.capi.default_bundling_state(a.bundling)
    gas.s_sensorIdx = 8
    if a.app is None:                       # integrate at the sensor resolution unless a parameter file says otherwise
        gas.s_integrationWidth, gas.s_integrationHeight = desc.depthWidth, desc.depthHeight
    if a.voxel: gas.s_SDFVoxelSize = a.voxel
    if a.buckets: gas.s_hashNumBuckets = a.buckets
    if a.blocks: gas.s_hashNumSDFBlocks = a.blocks
    if n > gbs.s_maxNumImages * gbs.s_submapSize:          # SensorDataReader.cpp:65-67
        raise SystemExit("sens file #frames = %d, please change param file to accommodate" % n)
    p = bf.capi.Pipeline(gas, gbs, desc)
    t0 = time.time()
    keep = []                                               # the last few host frames stay alive while their upload may be in flight
    for k in range(n):
        depth = sd.depth(k)                                 # metres, -inf invalid
        color = sd.color_rgbx(k)
        if not p.process_frame(depth, color):
            raise RuntimeError("frame not accepted")
        keep = (keep + [(depth, color)])[-4:]
    for _ in range(a.tail):
        p.process_end_of_sequence()
    p.synchronize()
    dt = time.time() - t0
    print("%s: %d frames  wall %.3f s -> %.1f frames/s (decode + PCIe included)" % (sd.sensor_name, n, dt, n / dt))
    print("counters", p.counters())
    traj = p.optimized_trajectory()
    valid = np.isfinite(traj[:, 0, 0])
    rmse, used = sd.evaluate_trajectory(traj)
    print("optimised trajectory: %d/%d valid; ate rmse = %.4f m over %d poses" % (valid.sum(), len(traj), rmse, used))
    sc = p.scene()
    print("allocated blocks", sc.num_allocated_blocks(), "heap free", sc.heap_free_count())
    if a.save:
        sd.save_with_trajectory(a.save, traj)
        print("wrote", a.save)


if __name__ == "__main__":
    main()
