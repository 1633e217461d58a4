#!/usr/bin/env python3
"""Run-to-run determinism of the frame loop: the same 33 frames (640x480 @4 mm, three local chunks, re-integrations, GC) through R fresh pipelines per arithmetic
contract of the voxel update; trajectories, counters and (per contract) the whole volume must be bit-identical across runs, and the trajectories across contracts.
    python tools/determinism_check.py [runs]"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bundlefusion_amd as bf
from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix, sensor_desc


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only = sys.argv[2] if len(sys.argv) > 2 else None          # "fast-batched": that configuration only, every run compared with ONE per-operator run (the bisecting form)
    W, H, n = 640, 480, 33
    frames = synth.render_frames(range(n))
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    out = []
    keep = {}
    for r in range(runs):
        for arith, batching in ((("fast", False), ("fast", True)) if only == "fast-batched" and r == 0 else (("fast", True),) if only == "fast-batched" else (("exact", True), ("fast", True), ("fast", False))):
            gas = default_app_state(); gbs = default_bundling_state()
            gas.s_integrationWidth, gas.s_integrationHeight = W, H
            gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.004, 1000000, 250000
            gbs.s_maxNumImages = 8
            p = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
            p.scene().set_arith(arith)
            p.set_volume_batching(batching)
            for d, c in dev:
                assert p.process_frame(d, c)
            for _ in range(4):
                p.process_end_of_sequence()
            p.synchronize()
            h, heap, cnt, vox = p.scene().download()
            sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
            key = (arith, batching) if only is None else "fast"
            if key in keep:          # what differs from the first run of this configuration
                v0 = keep[key]
                dif = np.nonzero((v0["sdf"] != vox["sdf"]) | (v0["weight"] != vox["weight"]) | (v0["color"] != vox["color"]).any(axis=1))[0]
                if len(dif):
                    ex = [(int(i), float(v0["sdf"][i]), float(vox["sdf"][i]), float(v0["weight"][i]), float(vox["weight"][i]), v0["color"][i].tolist(), vox["color"][i].tolist()) for i in dif[:6]]
                    print(json.dumps({"differing_voxels_vs_run0": int(len(dif)), "blocks": int(len(np.unique(dif // 512))), "examples": ex}), flush=True)
                    pos_of = {int(pt) // 512: h["pos"][i].tolist() for i, pt in enumerate(h["ptr"]) if pt >= 0} if only else {}
                    for b in np.unique(dif // 512)[:6]:
                        loc = dif[dif // 512 == b] % 512
                        print("  block %d key %s: %d voxels, (x,y,z) w0->w: %s" % (b, pos_of.get(int(b)), len(loc), " ".join("(%d,%d,%d)%g->%g" % (l % 8, (l // 8) % 8, l // 64, v0["weight"][b * 512 + l], vox["weight"][b * 512 + l]) for l in loc[:40])), flush=True)
            else:
                keep[key] = vox.copy()
            out.append({"run": r, "arith": arith, "batching": batching, "integrated": sha(p.integrated_trajectory()), "optimized": sha(p.optimized_trajectory()), "counters": p.counters(),
                        "table": sha(h["pos"]) + sha(h["ptr"]), "heap": sha(heap[:cnt + 1]), "voxels": sha(vox.view(np.uint8))})
            print(json.dumps(out[-1]), flush=True)
            del p
    traj = {(o["integrated"], o["optimized"]) for o in out}
    vol = {k: {(o["table"], o["heap"], o["voxels"]) for o in out if (o["arith"], o["batching"]) == k} for k in (("exact", True), ("fast", True), ("fast", False))}
    print(json.dumps({"distinct_trajectories": len(traj), "distinct_volumes_exact_batched": len(vol[("exact", True)]), "distinct_volumes_fast_batched": len(vol[("fast", True)]),
                      "distinct_volumes_fast_per_operator": len(vol[("fast", False)])}))


if __name__ == "__main__":
    main()
