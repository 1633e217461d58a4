#!/usr/bin/env python3
"""TSDF re-integration sweep (BASELINE.json configs[4] in miniature): F frames of the S2 stream are integrated
at their ground-truth poses, then every frame is re-integrated (de-integrate at P_k + integrate at P_k * exp(xi_k),
xi ~ N(0, diag(0.01 rad, 0.01 m)), seed 777) and swept back.  Only the volume runs — no SIFT, no solver — so the numbers are those
of the voxel-hash operators alone: per-operator wall time, HIP-event time of the voxel-update launches, algorithmic GB/s.

    python tools/tsdf_sweep.py [--frames 24] [--stride 10] [--width 640 --height 480] [--voxel 0.004] [--sweeps 2] [--separate]

Under `torch.distributed.run` the volume is sharded by hash-bucket range over the ranks (bf_scene_set_shard): every rank sees
every frame and pose, integrates only its shard; value = operators of the ONE volume per second (strong scaling).
Under `rocprofv3 --pmc ...` the same command gives the per-kernel counters (tools/rocpd_pmc.py).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def se3_exp(w, t):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th < 1e-12:
        R = np.eye(3) + K
        V = np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = V @ t
    return M


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--stride", type=int, default=10)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--voxel", type=float, default=0.004)
    ap.add_argument("--buckets", type=int, default=1000000)
    ap.add_argument("--blocks", type=int, default=600000)
    ap.add_argument("--sweeps", type=int, default=2)
    ap.add_argument("--separate", action="store_true", help="de-integrate + integrate as two operators instead of the fused one")
    ap.add_argument("--no-overlap", action="store_true", help="do not software-pipeline consecutive operators")
    ap.add_argument("--shard-alloc", action="store_true", help="(several ranks) divide the allocation's ray march over the ranks: every rank marches a band of the pixel "
                    "tiles (bf_scene_alloc_collect), ONE all-gather of the key lists per operator, every rank ingests all lists (bf_scene_alloc_ingest / _place)")
    ap.add_argument("--comm-alloc", action="store_true", help="(several ranks) the divided march INSIDE the operators: bf_scene_set_alloc_comm with an RCCL communicator of the C ABI "
                    "(include/bf_comm.h) - the collective is issued by the library on the allocation stream, no host round trip per operator")
    return ap


def main():
    a = parser().parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    out = run(a, rank, world)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


from bundlefusion_amd.capi import alloc_comm_capacity      # noqa: E402,F401  (tests/test_host_cpu.py imports it from here)


def run(a, rank=0, world=1):
    """The sweep on an initialised process group (or one rank); returns the result dict (meaningful on rank 0)."""
    from bundlefusion_amd import synth
    W, H = a.width, a.height
    frames = synth.render_frames([k * a.stride for k in range(a.frames)], W, H, workers=min(32, max(1, (os.cpu_count() or 1) // max(world, 1))))
    import torch
    import torch.distributed as dist
    import bundlefusion_amd as bf
    from bundlefusion_amd.capi import default_hash_params, camera_params
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=a.buckets, num_sdf_blocks=a.blocks, voxel_size=a.voxel)
    sc = bf.capi.SceneRepHashSDF(p)
    if getattr(a, "arith", None):
        sc.set_arith(a.arith)
    if world > 1:
        sc.set_shard(rank, world)
    comm = None
    if getattr(a, "comm_alloc", False) and world > 1:
        def bcast(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = bf.capi.Comm.rccl(world, rank, bcast)
        # keys one rank's band of the image may collect per operator: the distinct in-frustum blocks its rays cross - measured ~270 k per 1280x960 frame at 2 mm
        # (`n_occ_mean_per_op`), i.e. ~0.22 per pixel at 2 mm and 8x less at 4 mm; twice that, divided over the ranks, rounded up to a power of two.  The exchanged
        # records are fixed-size (8 bytes x capacity per rank and operator); exceeding the capacity raises the scene's error flag, it never drops silently.
        sc.set_alloc_comm(comm, alloc_comm_capacity(W, H, a.voxel, world))
    if not a.no_overlap:
        sc.set_overlap(True)
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    poses = [f[2].astype(np.float32) for f in frames]
    rng = np.random.RandomState(777)
    pert = []
    for T in poses:
        xi = rng.normal(0.0, 0.01, 6)
        pert.append((T.astype(np.float64) @ se3_exp(xi[:3], xi[3:])).astype(np.float32))

    CAP = 1 << 19
    if a.shard_alloc:
        sc.set_external_alloc(True)
        keys = torch.zeros(CAP, dtype=torch.int64, device="cuda"); slots = torch.zeros(CAP, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        all_keys = torch.zeros(world * (CAP + 1), dtype=torch.int64, device="cuda")

    def allocate(T, k):
        """collect on this rank's band, all-gather (keys + count as one fixed-size record per rank), ingest every list, place"""
        if not a.shard_alloc:
            return
        sc.alloc_collect(T, dev[k][0], cam, rank, world, keys, slots, cnt)
        sc.alloc_sync()
        rec = torch.cat([cnt.to(torch.int64), keys])
        if world > 1:
            dist.all_gather_into_tensor(all_keys, rec)
        else:
            all_keys.copy_(rec)
        torch.cuda.synchronize()
        for r in range(world):
            seg = all_keys[r * (CAP + 1):(r + 1) * (CAP + 1)]
            sc.alloc_ingest(seg[1:], seg[:1].view(torch.int32))       # little endian: the low word of the int64 count
        sc.alloc_place()

    def sync():
        sc.hash_params()          # drains both internal streams
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    for k in range(a.frames):
        allocate(poses[k], k)
        sc.integrate(poses[k], dev[k][0], dev[k][1], cam)
    sync()
    t_int = time.perf_counter() - t0

    sc.kernel_timing(True)
    t0 = time.perf_counter()
    cur = list(poses)
    for s in range(a.sweeps):
        tgt = pert if s % 2 == 0 else poses
        for k in range(a.frames):
            allocate(tgt[k], k)
            if a.separate:
                sc.deintegrate(cur[k], dev[k][0], dev[k][1], cam)
                sc.integrate(tgt[k], dev[k][0], dev[k][1], cam)
            else:
                sc.reintegrate(cur[k], tgt[k], dev[k][0], dev[k][1], cam)
            cur[k] = tgt[k]
    sync()
    t_sweep = time.perf_counter() - t0
    occ_sum, vis_plain, vis_fused, n_ops = sc.kernel_timing_blocks()
    n_launch, kernel_ms = sc.kernel_timing_read()
    sc.kernel_timing(False)
    if world > 1:
        v = torch.tensor([t_sweep, t_int], dtype=torch.float64, device="cuda")
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        t_sweep, t_int = float(v[0]), float(v[1])
    dbg = sc.debug_hash()
    if comm is not None:
        sc.set_alloc_comm(None)
    n_re = a.sweeps * a.frames
    # SURVEY.md §8d, per rank (its shard of the lists).  The bytes a launch MOVES: a fused launch walks the union list of its two poses once (bench.py's accounting)
    alg_bytes = (vis_plain + vis_fused) * (512 * 24 + 32) + n_launch * W * H * 8
    op_bytes = occ_sum * (512 * 24 + 32) + n_ops * W * H * 8           # per OPERATOR (a fused launch = two operators' B_op): what two separate operators would move
    return ({
            "workload": "%d frames %dx%d @%.0f mm, %d re-integration sweeps (%s), %d rank(s)" %
                        (a.frames, W, H, a.voxel * 1e3, a.sweeps, "separate operators" if a.separate else "fused operator", world),
            "integrate_us_per_op": 1e6 * t_int / a.frames,
            "reintegrate_us_per_frame": 1e6 * t_sweep / n_re,
            "reintegrations_per_s": n_re / t_sweep,
            "update_kernel_us_per_launch": 1e3 * kernel_ms / max(n_launch, 1),
            "update_kernel_share_of_wall": (kernel_ms / 1e3) / t_sweep,
            "n_occ_mean_per_op": occ_sum / max(n_ops, 1),
            "algorithmic_GBps_of_update_kernel_rank0": alg_bytes / (kernel_ms / 1e3) / 1e9 if kernel_ms > 0 else None,      # union accounting: bytes the launches move
            "algorithmic_GBps_wall_rank0": alg_bytes / t_sweep / 1e9,
            "operator_equivalent_GBps_of_update_kernel_rank0": op_bytes / (kernel_ms / 1e3) / 1e9 if kernel_ms > 0 else None,  # two separate operators' bytes per fused launch (may exceed the HBM peak: the fused launch moves the union once)
            "blocks_allocated_rank0": dbg["occupied"], "dropped": dbg["dropped"],
            "allocation": "march divided over the ranks inside the operators (RCCL all-gather of block keys per operator)" if comm is not None else
                          ("march divided over the ranks, exchange through torch.distributed" if a.shard_alloc else "every rank marches all pixels"),
        })


if __name__ == "__main__":
    main()
