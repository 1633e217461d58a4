#!/usr/bin/env python3
"""TEST INFRASTRUCTURE TOOL.  "ATE within 1 mm of the reference" against the reference's OWN spread.

    python tools/ate_distribution.py --ref ref.npz --perturbed ref_s1.npz ref_s2.npz ... --others oracle.npz product.npz

All files are trajectories of ONE stream written by tools/ref_ate_table.py (`--side ref --perturb-seed S`: the emulated reference with every valid
depth sample of its input moved by -1 / 0 / +1 float ulp at random - a change below any sensor's resolution).  Prints, for the online poses (what
each frame is integrated at) and the final optimised trajectory: ATE of every run; the distribution (mean, max) of |ATE(ref) - ATE(perturbed ref)|, of
the RMS position difference and of the largest pose-element difference between the reference and its perturbed selves; and the same three numbers for
each of `--others` against the reference - inside or outside that spread."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ref_ate_table import ate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True); ap.add_argument("--perturbed", nargs="+", required=True); ap.add_argument("--others", nargs="*", default=[])
    a = ap.parse_args()
    from bundlefusion_amd import synth
    z0 = np.load(a.ref, allow_pickle=True)
    start, stride, NF = int(z0["start"]), int(z0["stride"]), int(z0["frames"])
    gtT = [synth.trajectory_pose(start + stride * k).astype(np.float64) for k in range(NF)]
    T0inv = np.linalg.inv(gtT[0])
    gt = np.stack([T0inv @ T for T in gtT])

    def cmp(za, zb, key):
        A, B = za[key], zb[key]
        v = np.isfinite(A[:, 0, 0]) & np.isfinite(B[:, 0, 0])
        same = bool(np.array_equal(np.isfinite(A[:, 0, 0]), np.isfinite(B[:, 0, 0])))
        return (abs(ate(A, gt)[0] - ate(B, gt)[0]) * 1e3, 1e3 * float(np.sqrt(np.mean(np.sum((A[v, :3, 3] - B[v, :3, 3]) ** 2, axis=1)))), float(np.abs(A[v] - B[v]).max()), same)
    print("| run | online ATE mm | final ATE mm |")
    print("|---|---|---|")
    for f in [a.ref] + a.perturbed + a.others:
        z = np.load(f, allow_pickle=True)
        print("| %s | %.3f | %.3f |" % (os.path.basename(f), 1e3 * ate(z["online"], gt)[0], 1e3 * ate(z["final"], gt)[0]))
    print()
    print("| against the reference | trajectory | abs ATE difference mm | RMS position difference mm | largest pose-element difference | same frames valid |")
    print("|---|---|---|---|---|---|")
    for key in ("online", "final"):
        rows = [cmp(z0, np.load(f, allow_pickle=True), key) for f in a.perturbed]
        arr = np.array([r[:3] for r in rows])
        print("| reference under %d random sub-ulp input perturbations: mean | %s | %.3f | %.3f | %.2e | %s |" % (len(rows), key, arr[:, 0].mean(), arr[:, 1].mean(), arr[:, 2].mean(), all(r[3] for r in rows)))
        print("| ... max | %s | %.3f | %.3f | %.2e | |" % (key, arr[:, 0].max(), arr[:, 1].max(), arr[:, 2].max()))
        for f in a.others:
            r = cmp(z0, np.load(f, allow_pickle=True), key)
            inside = "inside" if (r[0] <= arr[:, 0].max() and r[1] <= arr[:, 1].max()) else "outside"
            print("| %s (%s the reference's own spread) | %s | %.3f | %.3f | %.2e | %s |" % (os.path.basename(f), inside, key, r[0], r[1], r[2], r[3]))


if __name__ == "__main__":
    main()
