#!/usr/bin/env python3
"""Run-to-run identity over FRESH PROCESSES: N processes, each one frame loop (tools/first_run_check.py: 640x480 @4 mm, the library's defaults - fast contract, batched
operators), every process's trajectory / table / voxel hashes on one line; the summary counts the distinct results.
    python tools/determinism_processes.py <processes> [frames] [out.jsonl]        (BF_LIB_PATH=<variant> selects another build of the library)"""
import collections
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = int(sys.argv[1]); frames = sys.argv[2] if len(sys.argv) > 2 else "61"; out = sys.argv[3] if len(sys.argv) > 3 else None
    env = dict(os.environ, FIRST_RUN_FRAMES=frames, FIRST_RUN_CACHE="/tmp/bf_first_run")
    res, t0 = [], time.time()
    for i in range(n):
        pat = ("none", "7fc00000", "ffffffff", "0")[i % 4] if os.environ.get("DET_POISON") else "none"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_run_check.py"), pat], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            res.append({"failed": r.returncode, "stderr": r.stderr.decode()[-400:]})
            continue
        res.append(json.loads(line[-1]))
        if out:
            with open(out, "a") as f:
                f.write(line[-1] + "\n")
    ok = [r for r in res if "failed" not in r]
    traj = collections.Counter((r["integrated"], r["optimized"]) for r in ok)
    vol = collections.Counter((r["table"], r["voxels"]) for r in ok)
    s = {"lib": os.environ.get("BF_LIB_PATH", "product"), "processes": n, "frames": int(frames), "failed": len(res) - len(ok), "distinct_trajectories": len(traj), "distinct_volumes": len(vol),
         "trajectory_counts": sorted(traj.values(), reverse=True), "volume_counts": sorted(vol.values(), reverse=True), "counters": ok[0]["counters"] if ok else None, "seconds": round(time.time() - t0, 1)}
    print(json.dumps(s), flush=True)
    if out:
        with open(out, "a") as f:
            f.write(json.dumps(s) + "\n")


if __name__ == "__main__":
    main()
