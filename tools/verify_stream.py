#!/usr/bin/env python3
"""Execution-error rate of the batched fast-contract voxel update inside the running frame loop (diagnostic BF_DEBUG_VERIFY_BATCH of the library: every batch
update is launched three times - the volume and two shadow copies of the union list's blocks - and the results are compared voxel by voxel).  A 420-frame stream
is ~400 batches x 3 executions of the real kernel beside its real neighbours (ingest, SIFT, matching chain, the next batch's preparation).
    python tools/verify_stream.py [--frames 420] [--runs 3] [--cache /tmp/bf_frames] [--tag name] [--out file.json]
Library variants: BF_LIB_PATH=<variant .so> (tools/build_variant.py); BF_DEBUG_NO_OVERLAP=1 puts the preparation on the update's stream."""
import argparse
import collections
import json
import os
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

REC = 96


def parse(path):
    """-> list of dumps: (count, cap, batches, blocks, records[n] as uint32 [n, 24])"""
    out = []
    raw = open(path, "rb").read() if os.path.exists(path) else b""
    o = 0
    while o + 16 <= len(raw):
        count, cap, batches, blocks = struct.unpack_from("<4I", raw, o)
        o += 16
        n = min(count, cap)
        out.append((count, cap, batches, blocks, np.frombuffer(raw, dtype=np.uint32, count=n * (REC // 4), offset=o).reshape(n, REC // 4).copy()))
        o += n * REC
    return out


def summarise(dumps):
    batches = sum(d[2] for d in dumps)
    blocks = sum(d[3] for d in dumps)
    events = collections.OrderedDict()          # (dump, seq, blk) -> records
    for di, d in enumerate(dumps):
        for r in d[4]:
            events.setdefault((di, int(r[0]), int(r[1])), []).append(r)
    lanes, slices, odd, nvox, wdiff = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    lane_sets = collections.Counter()
    examples = []
    for key, recs in events.items():
        ls = sorted({int(r[2]) % 64 for r in recs}); zs = sorted({int(r[2]) // 64 for r in recs})
        for l in ls:
            lanes[l // 16] += 1
        lane_sets["%d-%d" % (ls[0], ls[-1])] += 1
        for z in zs:
            slices[z] += 1
        nvox[len(recs)] += 1
        for r in recs:
            w = int(r[3])
            who = {3: "volume", 5: "shadow0", 6: "shadow1"}.get(w, "all-three" if w == 7 else "?%d" % w)
            odd[who] += 1
            v = r[8:17].view(np.float32).reshape(3, 3)
            truth = v[1] if w == 3 else v[0]
            bad = v[0] if w == 3 else v[1] if w == 5 else v[2]
            wdiff[int(round(float(bad[1] - truth[1])))] += 1
        if len(examples) < 12:
            r = recs[0]
            v = r[8:17]
            examples.append({"dump": key[0], "batch": key[1], "list_index": key[2], "block": [int(np.int32(r[4])), int(np.int32(r[5])), int(np.int32(r[6]))], "mask": "%x" % int(r[7]), "nOps": int(r[17]),
                             "voxels": len(recs), "lanes": ls, "slices": zs,
                             "first": {"vox": int(r[2]), "which": int(r[3]), "sdf": [float(x) for x in v[[0, 3, 6]].view(np.float32)], "w": [float(x) for x in v[[1, 4, 7]].view(np.float32)], "rgbx": ["%08x" % int(x) for x in v[[2, 5, 8]]]}})
    return {"batches": batches, "update_launches": 3 * batches, "blocks_x_launch": 3 * blocks, "events": len(events), "differing_voxels": int(sum(len(v) for v in events.values())),
            "events_per_1e6_block_updates": round(1e6 * len(events) / max(3 * blocks, 1), 3),
            "lane_quarters_hit": dict(sorted(lanes.items())), "lane_ranges": dict(lane_sets.most_common(8)), "slices_hit": dict(sorted(slices.items())), "odd_one": dict(odd), "voxels_per_event": dict(sorted(nvox.items())),
            "weight_delta_of_the_odd_value": dict(sorted(wdiff.items())), "log_overflow": any(d[0] > d[1] for d in dumps), "examples": examples}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=420)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--cache", default="/tmp/bf_frames")
    ap.add_argument("--tag", default="default")
    ap.add_argument("--out", default=None)
    ap.add_argument("--arith", default="fast")
    ap.add_argument("--batching", default="on")
    ap.add_argument("--parse-only", default=None)
    args = ap.parse_args()
    if args.parse_only:
        print(json.dumps(summarise(parse(args.parse_only))))
        return
    log = "/tmp/bf_verify_%s_%d.bin" % (args.tag, os.getpid())
    os.environ["BF_DEBUG_VERIFY_BATCH"] = log
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix, sensor_desc
    W, H, n = 640, 480, args.frames
    cache = "%s_%d.npz" % (args.cache, n)
    t0 = time.time()
    if os.path.exists(cache):
        z = np.load(cache)
        depth, color = z["depth"], z["color"]
    else:
        fr = synth.render_frames(range(n), W, H)
        depth = np.stack([f[0] for f in fr]); color = np.stack([f[1] for f in fr])
        np.savez(cache, depth=depth, color=color)
    Kd = synth.intrinsics(W, H)
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    dev = [(torch.from_numpy(depth[i]).cuda(), torch.from_numpy(color[i]).cuda()) for i in range(n)]
    t_render = time.time() - t0
    hashes = []
    t1 = time.time()
    for r in range(args.runs):
        gas = default_app_state(); gbs = default_bundling_state()
        gas.s_integrationWidth, gas.s_integrationHeight = W, H
        gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.004, 1000000, 250000
        gbs.s_maxNumImages = max(n // 10 + 8, 16)
        p = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
        p.scene().set_arith(args.arith)
        p.set_volume_batching(args.batching == "on")
        for d, c in dev:
            assert p.process_frame(d, c)
        p.synchronize()
        import hashlib
        hashes.append(hashlib.sha256(p.integrated_trajectory().tobytes()).hexdigest()[:12])
        cnt = p.counters()
        del p
    s = summarise(parse(log))
    s.update({"tag": args.tag, "lib": os.environ.get("BF_LIB_PATH", "product"), "no_overlap": bool(os.environ.get("BF_DEBUG_NO_OVERLAP")), "frames": n, "runs": args.runs,
              "distinct_trajectories": len(set(hashes)), "counters_last_run": cnt, "seconds_frames": round(t_render, 1), "seconds_runs": round(time.time() - t1, 1)})
    line = json.dumps(s)
    print(line, flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    os.remove(log)


if __name__ == "__main__":
    main()
