"""TEST INFRASTRUCTURE TOOL.  "ATE within 1 mm of the reference" as a number.

One loop-closure stream of the synthetic room (BASELINE configs[2] in small: NF frames `stride` degrees/5 apart once around the room and
into the second lap, chunk size 10, the loop closed by a global match between key frames ~20 apart) through ONE of

  --side ref      the emulated reference (tests/ref_api.RefOnlineBundler: the reference's own host classes and kernels on the block emulator;
                  build container only - needs oracle/_ref built from /root/reference); `--perturb U` moves every valid depth sample of the
                  input by U float ulps first (self-perturbation: how far does the reference deviate from ITSELF under an input change
                  below any sensor's resolution);
  --side oracle   the oracle frame loop (tests/oracle_pipeline.OraclePipeline);
  --side product  the product through the C ABI on the GPU (bundlefusion_amd.capi.Pipeline) - run on the GPU box;

and writes the poses (the pose handed to the integration when each frame arrived = "online", and the final optimised trajectory) to an .npz.
`--table a.npz b.npz ...` prints ATE against the ground truth of the stream for every file plus the pairwise differences (mm).

ATE = RMSE of the camera positions, frame 0 anchored on the ground truth (the metric of tests/test_pipeline_baseline_gpu.py), and, second
column, after a rigid Kabsch alignment (PoseHelper.h:35-79 evaluates that way)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def params(a):
    from bundlefusion_amd.capi import default_app_state, default_bundling_state
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = a.width, a.height
    gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = a.voxel, a.buckets, a.blocks
    gas.s_garbageCollectionEnabled = False
    if a.exit_frames >= 0:
        gas.s_numSolveFramesBeforeExit = a.exit_frames          # the end-of-scan switch to the dense global solve after that many iterations past the end
    gbs.s_widthSIFT, gbs.s_heightSIFT, gbs.s_maxNumImages, gbs.s_submapSize = a.width, a.height, a.frames // a.submap + 8, a.submap
    return gas, gbs


def stream(a):
    from bundlefusion_amd import synth
    fr = synth.render_frames([a.start + a.stride * k for k in range(a.frames)], a.width, a.height)
    if a.perturb_seed >= 0:          # every valid depth sample moved by -1, 0 or +1 float ulp, independently, from a seeded generator
        rng = np.random.RandomState(a.perturb_seed)
        out = []
        for d, c, T, K in fr:
            d2 = d.copy(); v = np.isfinite(d2)
            x = d2[v]
            step = rng.randint(-1, 2, size=x.shape)
            x = np.where(step > 0, np.nextafter(x, np.float32(np.inf), dtype=np.float32), np.where(step < 0, np.nextafter(x, np.float32(0), dtype=np.float32), x)).astype(np.float32)
            d2[v] = x
            out.append((d2, c, T, K))
        fr = out
    elif a.perturb:
        out = []
        for d, c, T, K in fr:
            d2 = d.copy(); v = np.isfinite(d2)
            x = d2[v]
            for _ in range(abs(a.perturb)):
                x = np.nextafter(x, np.float32(np.inf if a.perturb > 0 else 0), dtype=np.float32)
            d2[v] = x
            out.append((d2, c, T, K))
        fr = out
    return fr


def run_ref(a):
    from tests import ref_api as R
    from tests.oracle_pipeline import OraclePipeline, _minf
    from bundlefusion_amd.capi import intrinsics_matrix
    assert R.available()
    gas, gbs = params(a)
    frames = stream(a)
    Kd = frames[0][3]; K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    op = OraclePipeline(gas, gbs, a.width, a.height, K)        # only its ingest (the depth filter with the deterministic exp) is used
    rb = R.RefOnlineBundler(gas, gbs, a.width, a.height, K); rtm = rb.trajectory_manager()
    online = np.full((a.frames, 4, 4), -np.inf, np.float32)
    t0 = time.time()
    for i in range(a.frames + a.tail):
        if i < a.frames:
            d, c = frames[i][0], frames[i][1]
            raw, filt = op._ingest(d, c); rb.set_frame(d, c); rb.override_filtered_depth(filt)
            rb.process_input()
            ok, T, idx, lost = rb.current_integration_frame()
            if ok:
                online[i] = T
            if rtm.active() < gas.s_maxFrameFixes:
                rtm.generate()
            for _ in range(gas.s_maxFrameFixes):        # DepthSensing.cpp:854-902 bookkeeping without a volume
                f, ix, TT, _ = rtm.top_de()
                if f: continue
                f, ix, TT, _ = rtm.top_in()
                if f: rtm.confirm(ix); continue
                f, ix, o_, n_ = rtm.top_re()
                if f: rtm.confirm(ix); continue
                break
            rtm.add(0 if ok else 1, T if ok else _minf(), i)
        else:
            rb.process_input()
        rb.process()
        if i % 10 == 0:
            print("ref frame %d  %.0f s" % (i, time.time() - t0), flush=True)
    st = rb.state()
    n = st["num_complete"]
    final = np.full((a.frames, 4, 4), -np.inf, np.float32)
    final[:min(n, a.frames)] = rb.complete_trajectory(n)[:a.frames]
    g = rb.bundler(2)
    return online, final, dict(key_frames=g.num_frames(), corr=len(g.correspondences()))


def run_oracle(a):
    from tests.oracle_pipeline import OraclePipeline
    from bundlefusion_amd.capi import intrinsics_matrix
    gas, gbs = params(a)
    frames = stream(a)
    Kd = frames[0][3]; K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    op = OraclePipeline(gas, gbs, a.width, a.height, K)
    op._integrate = lambda frame, T, de: None
    t0 = time.time()
    on = np.full((a.frames, 4, 4), -np.inf, np.float32)
    for i, (d, c, _, _) in enumerate(frames):
        op.process_frame(d, c)
        if op.last_valid:
            on[i] = op.cur_T[op.last_processed]
        if i % 10 == 0:
            print("oracle frame %d  %.0f s" % (i, time.time() - t0), flush=True)
    for _ in range(a.tail):
        op.process_end_of_sequence()
    final = np.full((a.frames, 4, 4), -np.inf, np.float32)
    n = min(op.num_complete, a.frames)
    final[:n] = np.asarray(op.complete[:n], np.float32)
    gc = op.glob.corr
    return on, final, dict(key_frames=op.glob.num_images, corr=int((gc["imgIdx_i"] != 0xFFFFFFFF).sum()))


def run_product(a):
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd.capi import intrinsics_matrix, sensor_desc
    gas, gbs = params(a)
    frames = stream(a)
    Kd = frames[0][3]; K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    gp = bf.capi.Pipeline(gas, gbs, sensor_desc(a.width, a.height, K))
    on = np.full((a.frames, 4, 4), -np.inf, np.float32)
    for i, (d, c, _, _) in enumerate(frames):
        if not gp.process_frame(torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()):
            raise RuntimeError("process_frame failed")
        it = gp.integrated_trajectory()          # completes the deferred frame: row i is the pose frame i was just integrated at
        if len(it) > i:
            on[i] = it[i]
    for _ in range(a.tail):
        gp.process_end_of_sequence()
    gp.synchronize()
    final = gp.optimized_trajectory()
    fi = np.full((a.frames, 4, 4), -np.inf, np.float32); fi[:len(final)] = final[:a.frames]
    c = gp.counters()
    return on, fi, dict(key_frames=int(c.get("global_frames", -1)), corr=-1, counters={k: int(v) for k, v in c.items()})


def kabsch(P, Q):
    """rigid (R, t) minimising |R P + t - Q|"""
    cp, cq = P.mean(0), Q.mean(0)
    Hm = (P - cp).T @ (Q - cq)
    U, _, Vt = np.linalg.svd(Hm)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rm = Vt.T @ D @ U.T
    return Rm, cq - Rm @ cp


def ate(traj, gt):
    v = np.isfinite(traj[:, 0, 0])
    p, q = traj[v, :3, 3].astype(np.float64), gt[v, :3, 3]
    direct = float(np.sqrt(np.mean(np.sum((p - q) ** 2, axis=1))))
    Rm, t = kabsch(p, q)
    aligned = float(np.sqrt(np.mean(np.sum((p @ Rm.T + t - q) ** 2, axis=1))))
    return direct, aligned, int(v.sum())


def table(files):
    from bundlefusion_amd import synth
    runs = []
    for f in files:
        z = np.load(f, allow_pickle=True)
        runs.append((os.path.basename(f), z))
    z0 = runs[0][1]
    start, stride, NF = int(z0["start"]), int(z0["stride"]), int(z0["frames"])
    gtT = [synth.trajectory_pose(start + stride * k).astype(np.float64) for k in range(NF)]
    T0inv = np.linalg.inv(gtT[0])
    gt = np.stack([T0inv @ T for T in gtT])
    print("| run | key frames | online ATE mm (anchored / aligned) | final ATE mm (anchored / aligned) | frames |")
    print("|---|---|---|---|---|")
    for name, z in runs:
        a1 = ate(z["online"], gt); a2 = ate(z["final"], gt)
        print("| %s | %s | %.3f / %.3f | %.3f / %.3f | %d / %d |" % (name, z["key_frames"], 1e3 * a1[0], 1e3 * a1[1], 1e3 * a2[0], 1e3 * a2[1], a1[2], a2[2]))
    print()
    print("| pair | online: max pose element diff / RMS position diff mm | final: max pose element diff / RMS position diff mm | |ATE_final(a) - ATE_final(b)| mm |")
    print("|---|---|---|---|")
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            (na, za), (nb, zb) = runs[i], runs[j]
            row = []
            for key in ("online", "final"):
                A, B = za[key], zb[key]
                v = np.isfinite(A[:, 0, 0]) & np.isfinite(B[:, 0, 0])
                dmax = float(np.abs(A[v] - B[v]).max())
                rms = float(np.sqrt(np.mean(np.sum((A[v, :3, 3] - B[v, :3, 3]) ** 2, axis=1))))
                row.append("%.2e / %.3f" % (dmax, 1e3 * rms))
            dA = abs(ate(za["final"], gt)[0] - ate(zb["final"], gt)[0])
            print("| %s vs %s | %s | %s | %.3f |" % (na, nb, row[0], row[1], 1e3 * dA))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--side", choices=["ref", "oracle", "product"])
    p.add_argument("--width", type=int, default=640); p.add_argument("--height", type=int, default=480)
    p.add_argument("--start", type=int, default=0); p.add_argument("--stride", type=int, default=9); p.add_argument("--frames", type=int, default=212)
    p.add_argument("--submap", type=int, default=10); p.add_argument("--tail", type=int, default=5)
    p.add_argument("--voxel", type=float, default=0.05); p.add_argument("--buckets", type=int, default=5000); p.add_argument("--blocks", type=int, default=2000)
    p.add_argument("--perturb", type=int, default=0)
    p.add_argument("--perturb-seed", type=int, default=-1, help="every valid depth sample moved by -1 / 0 / +1 float ulp at random (seed): independent sub-resolution perturbations")
    p.add_argument("--exit-frames", type=int, default=-1, help="s_numSolveFramesBeforeExit (default: the parameter file's 30, i.e. no end-of-scan dense solve within --tail)")
    p.add_argument("--out"); p.add_argument("--table", nargs="*")
    a = p.parse_args()
    if a.table:
        table(a.table); return
    t0 = time.time()
    online, final, info = {"ref": run_ref, "oracle": run_oracle, "product": run_product}[a.side](a)
    np.savez(a.out, online=online, final=final, start=a.start, stride=a.stride, frames=a.frames, side=a.side, perturb=a.perturb,
             key_frames=info["key_frames"], corr=info["corr"], seconds=time.time() - t0)
    print(a.side, info, "%.0f s" % (time.time() - t0), "->", a.out)


if __name__ == "__main__":
    main()
