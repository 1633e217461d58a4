#!/usr/bin/env python3
"""Global solve and global matching vs the number of key frames (N <= 500 @5000 frames, <= 2000 @20000).

Synthetic key-frame graph of the shape the frame loop produces: N poses on a loop, every key frame matched to its k predecessors (25
correspondences per pair, 2 mm noise) plus loop-closure pairs between the ends; the global configuration of the solver (3 Gauss-Newton x 150
PCG iterations, sparse weight 1, no dense term: SBA.cpp:35-38).  Prints one markdown row per N: wall time of bf_solver_solve (HIP events),
per PCG iteration, workgroups of the cooperative PCG; and the time of one global matching step (the new key frame against all N-1 others:
bf_siftmgr_match + the filter chain is per pair and parallel over pairs, so only the matcher is timed here).

    python tools/solver_scaling.py [--n 22 100 500 1000 2000] > profiles/r03_solver_scaling.md
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def graph(n, k, rng):
    from bundlefusion_amd.capi import ENTRYJ_DTYPE
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    T = np.tile(np.eye(4), (n, 1, 1))
    for i, a in enumerate(ang):
        c, s = np.cos(a), np.sin(a)
        T[i, :3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        T[i, :3, 3] = [np.cos(a), 0.0, np.sin(a)]
    T0inv = np.linalg.inv(T[0])
    T = np.stack([T0inv @ t for t in T])
    pairs = [(i, j) for j in range(n) for i in range(max(0, j - k), j)] + [(i, n - 1 - i) for i in range(min(3, n // 4))]
    rows = np.zeros(len(pairs) * 25, dtype=ENTRYJ_DTYPE)
    r = 0
    for i, j in pairs:
        pw = rng.uniform(-1, 1, (25, 3)) + np.array([0, 0, 2.5])
        pw = (T[j] @ np.c_[pw, np.ones(25)].T).T[:, :3]
        pi = (np.linalg.inv(T[i]) @ np.c_[pw, np.ones(25)].T).T[:, :3] + rng.normal(0, 0.002, (25, 3))
        pj = (np.linalg.inv(T[j]) @ np.c_[pw, np.ones(25)].T).T[:, :3] + rng.normal(0, 0.002, (25, 3))
        for a, b in zip(pi, pj):
            rows[r] = (i, j, a.astype(np.float32), b.astype(np.float32)); r += 1
    Tin = T.copy()
    for i in range(1, n):
        Tin[i, :3, 3] += rng.normal(0, 0.01, 3)
    return rows, Tin.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="*", default=[22, 100, 300, 500, 1000, 2000])
    ap.add_argument("--k", type=int, default=3, help="predecessors every key frame is matched to")
    a = ap.parse_args()
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd.capi import default_solver_config, bind_host_threads_to_device
    bind_host_threads_to_device(0)
    rng = np.random.default_rng(0)
    print("| key frames N | correspondences | bf_solver_solve 3 x 150 [ms] | per PCG iteration [us] | PCG iterations run | global match of one key frame vs N-1 [ms] |")
    print("|---|---|---|---|---|---|")
    for n in a.n:
        corr, Tin = graph(n, a.k, rng)
        from tests import oracle_api
        rot, tr = oracle_api.matrices_to_poses(Tin)
        ms, its, err = float("nan"), 0, ""
        try:
            solver = bf.capi.Solver(n, len(corr), default_solver_config(record_convergence=False))
            gcorr = torch.from_numpy(corr.view(np.uint8)).cuda()
            valid = torch.ones(n, dtype=torch.int32, device="cuda")
            times = []
            for rep in range(3):
                grot, gtr = torch.from_numpy(rot.copy()).cuda(), torch.from_numpy(tr.copy()).cuda()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                solver.solve(gcorr, len(corr), valid, n, 3, 150, None, [1.0] * 3, [0.0] * 3, [0.0] * 3, grot, gtr, find_max_residual=True)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            gn, pcg = solver.iteration_counts()
            its = sum(pcg)
            ms = 1e3 * min(times)
            del solver
        except Exception as e:
            err = " solver: " + repr(e)[:200]
        # global matching: image n-1 against all others, 1024 keys each
        tm = [float("nan")]
        try:
            mgr = bf.capi.SiftManager(n + 1, 1024)
            keys = np.c_[rng.uniform(5, 630, 1024), rng.uniform(5, 470, 1024), rng.uniform(3, 12, 1024), rng.uniform(0.8, 3.0, 1024)].astype(np.float32)
            for i in range(n):
                mgr.add_image_host(keys, rng.integers(0, 60, (1024, 128)).astype(np.uint8))
            torch.cuda.synchronize()
            tm = []
            for rep in range(3):
                t0 = time.perf_counter()
                mgr.match(n - 1, 0, n)
                torch.cuda.synchronize()
                tm.append(time.perf_counter() - t0)
            del mgr
        except Exception as e:
            err += " match: " + repr(e)[:200]
        print("| %d | %d | %.2f | %.1f | %d | %.2f |%s" % (n, len(corr), ms, 1e3 * ms / max(its, 1), its, 1e3 * min(tm or [float("nan")]), err), flush=True)


if __name__ == "__main__":
    main()
