#!/usr/bin/env python3
"""bench.py — BundleFusion hot path on MI355X, end to end.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], synthetic stand-in for the recorded .sens sequence): the S2 "room"
stream, 640x480 depth + colour, 4 mm voxels.  One step = one input frame through the whole serial frame loop of the
reference (DepthSensing.cpp:966-1095): ingest (2x erode + range-gated Gaussian), SIFT detect, 80x60 dense cache, descriptor
matching against the chunk, Kabsch / surface-area / dense-verify filters, SIFT pose, up to s_maxFrameFixes re-integrations
(de-integrate + integrate) with garbage collection, integration of the new frame, and — every s_submapSize frames — the
local Gauss-Newton/PCG solve (sparse + dense), chunk-to-keyframe fusion, global matching and the global solve.
Nothing is skipped inside the timed region.  Frames are resident in HBM before the timed region starts
(bf_pipeline_process_frame_device); `--host` hands over host buffers instead (PCIe-inclusive rate).

The timed region is the OPERATING POINT of the loop, not its start-up: `--preroll` frames (default 200) are processed untimed
first, so that the re-integration queue is saturated (s_maxFrameFixes = 10 re-integrations per frame) and the global problem
holds >= 20 key frames when the W warm-up and K timed steps run (with the driver's --steps 20 --warmup 5: frames 205..224).

Multi-GPU (N>1, default mode "chunks", STRONG scaling of ONE stream — the partition north_star names): local chunks of
10 frames are dealt round-robin to the ranks; each rank runs the chunk-local half (SIFT, matching + filters inside the chunk, local
solve, key-frame fusion) for its chunks; ONE RCCL all-gather per round of N chunks (0.4 MB per chunk) hands every package to every
rank; every rank then runs the global half (pose chaining, global match + solve, TrajectoryManager, re-integration scheduling) on all
packages in stream order and integrates into its hash-bucket shard of the one volume.  The global solve is replicated: it is
bit-deterministic, so the pose update needs no exchange beyond one RCCL MIN/MAX all-reduce that verifies it after the run.
`value` = frames of the one stream / slowest rank.  Other modes: "segments" (independent stream segments per rank, weak scaling) and
"volume-shard" (every rank bundles everything, the volume is sharded).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s HBM3E
PMC_FILE = "r06_pmc_tsdf_update.json"
CLOCK_HZ = 2.4e9             # MI355X_MICROARCH.md: max clock; 256 CUs x 4 SIMDs; a full-rate wave64 vector instruction issues in 2 cycles (SIMD-32)
SIMDS = 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preroll", type=int, default=200, help="untimed frames processed before the warm-up (brings the loop to its operating point)")
    ap.add_argument("--clock-warmup", type=float, default=3.0, help="seconds of untimed GPU work before the pre-roll: the frames are rendered on the host for 10-20 s "
                    "while the GPU idles in its lowest power state, and the first ~second of work after that runs at reduced clocks")
    ap.add_argument("--pmc-out", default=None, help="(profiling runs) write the launch / block accounting of the WHOLE run here, for tools/pmc_to_json.py")
    ap.add_argument("--voxel", type=float, default=0.004)
    ap.add_argument("--buckets", type=int, default=1000000)
    ap.add_argument("--blocks", type=int, default=600000)
    ap.add_argument("--host", action="store_true", help="hand over host buffers each frame (PCIe-inclusive)")
    ap.add_argument("--mode", choices=["auto", "serial", "chunks", "segments", "volume-shard"], default="auto",
                    help="auto = 'serial' on one GPU, 'chunks' on N>1: one stream, local chunks round-robin over the ranks + all-gather of key-frame "
                         "packages + replicated global half + volume sharded by hash bucket (strong scaling).  'segments' = each rank its own stream "
                         "segment and volume (weak scaling); 'volume-shard' = one stream, bundling replicated, volume sharded")
    ap.add_argument("--arith", choices=["fast", "exact"], default=os.environ.get("BF_TSDF_ARITH", "fast"),
                    help="arithmetic contract of the voxel update in the measured leg (bf_scene_set_arith): 'fast' = the reference GPU build's own contract "
                         "(-use_fast_math), 'exact' = IEEE op by op, bit-identical with the oracle")
    ap.add_argument("--both-contracts", action="store_true", default=True, help="(1 GPU) also measure the other contract, reported as other_contract")
    ap.add_argument("--one-contract", dest="both_contracts", action="store_false")
    ap.add_argument("--solve-lag", type=int, default=int(os.environ.get("BF_BENCH_SOLVE_LAG", "-1")),
                    help="-1: the library's default schedule (since round 6: L = 10).  L in 2..10 (at least the frame loop's depth): the chunk solves run on their own "
                         "thread and stream and are applied exactly L frames later (bf_pipeline_set_solve_lag) - the reference's optimiser thread "
                         "(FriedLiver.cpp:112-143) with a defined hand-over; same solves, same count, nothing skipped.  0: the serial order of the reference's "
                         "single-threaded branch (chunk solves inside the frame that closes the chunk), reported as `serial_order` beside the default")
    ap.add_argument("--volume-batching", choices=["on", "off"], default="on", help="on (library default): the frame loop issues a frame's TSDF operators - the integration of the previous "
                    "frame and this frame's up to s_maxFrameFixes re-integrations - as ONE bf_scene_run_batch (one march, one placement, one pass over the touched blocks); "
                    "off: one operator at a time (the rounds 1-4 path), for comparison.  Same volume either way (tests/test_tsdf_batch_gpu.py)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the secondary block `sweep` (BASELINE configs[4] in small: the 1280x960 @2 mm re-integration sweep, the "
                    "workload north_star's >= 6x scaling target is quoted on; ~20 s, most of it rendering 24 frames on the host)")
    ap.add_argument("--long-stream", type=int, default=5000, help="also run a stream of this many frames from frame 0 through a fresh pipeline and report it as `long_stream` "
                    "(5000, the default: north_star's target stream - BASELINE configs[3] on one GPU, 2.5 loops + vertical sinusoid; 2000: configs[2], the stride-1 loop-closure "
                    "stream, the one tests/golden/oracle_stream_2000.npz holds the oracle's results for; 0: skip).  Rendering takes ~1 s of host time per 16 frames (5000 frames: ~5 minutes, untimed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-class-surface", action="store_true", help="skip the secondary block `class_surface` (the same window through the reference-named C++ classes of "
                    "include/bundlefusion/bundlefusion.hpp, examples/class_surface_bench.cpp)")
    ap.add_argument("--cpu-frames", type=int, default=31, help="frames of the stream the CPU baseline processes (three local chunks: the last ten frames "
                    "run with the re-integration queue saturated, like the timed window of the GPU leg)")
    ap.add_argument("--launch-check", action="store_true", help="(CPU test of the launch path) rendezvous over gloo, print the world size, exit")
    args = ap.parse_args()

    plan = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if not args.launch_check:
        # fewer devices than ranks: say so at once (every rank would otherwise reach ncclCommInitRank on a device that does not exist, or share one, and hang or fail late)
        visible = visible_devices()
        if visible is not None and visible < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible to this process (rocm-smi / HIP_VISIBLE_DEVICES): no line is printed" % (args.gpus, visible))
    if plan is not None:
        # `python bench.py --gpus N` without a launcher around it: start the N ranks here (one process per GPU, torch.distributed.run on
        # 127.0.0.1) instead of silently measuring one GPU.  Rank 0 of the children prints the JSON line; this process only forwards it.
        import subprocess
        sys.exit(subprocess.call(plan))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): no line is printed for a mismatched run" % (args.gpus, world))
    if args.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            import torch
            t = torch.ones(1); dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world}))
        return
    W, H = 640, 480
    pre = args.preroll + args.warmup
    if (world > 1 and args.mode == "auto") or args.mode == "chunks":
        # chunk-parallel mode: a round is world * 10 frames (frames r*world*10 + 1 .. (r+1)*world*10).  The timed window starts at a round
        # boundary, so that it contains this round's all-gather and - running ahead on the second host thread - the whole local half of
        # the next round (SIFT, local matching, local solve, key-frame fusion of one chunk per rank), not only the replicated global half.
        from bundlefusion_amd.shard import timed_window
        pre = timed_window(pre, args.steps, world)[0]
    total = pre + args.steps

    # synthetic stream, rendered by plain-python subprocesses before HIP is initialised
    from bundlefusion_amd import synth                      # (imports torch; no device context yet)
    import numpy as np
    ncpu = os.cpu_count() or 1
    from bundlefusion_amd.shard import segment, max_over_ranks, same_over_ranks, ChunkedRunner
    mode = args.mode if args.mode != "auto" else ("chunks" if world > 1 else "serial")
    chunked = mode == "chunks"
    shard_volume = (mode == "volume-shard" and world > 1) or chunked
    chunked_alloc_comm = False                                       # (the headline leg keeps the local march; the divided march is the sweep block's)
    one_stream = shard_volume or mode == "serial" or world == 1
    first = 0 if one_stream else segment(rank, world, total)[0]   # segments: each rank its own contiguous part of the S2 loop
    n_render = total
    if chunked:
        # the last round of `world` chunks must be complete (its local halves run before the all-gather), and the stream continues for ONE MORE
        # round, whose chunk-local halves (SIFT, matching inside the chunk, local solve, key-frame fusion) run inside the timed window: shard.timed_window
        from bundlefusion_amd.shard import timed_window
        n_render = timed_window(pre, args.steps, world)[2]
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # One stream on several ranks: every rank needs every frame, but nobody has to render all of them - rank r renders a contiguous share (the
    # renderer's children are fresh python processes without torch / HIP, so this is safe after the device context exists) and ONE all-gather
    # per image plane hands every rank the whole stream (untimed set-up, 2.4 MB per frame over xGMI instead of ~1 s of host rendering per frame
    # and rank).  BF_BENCH_SHARED_RENDER=1 takes this path with one rank too (how it is exercised on a single GPU).
    shared_render = one_stream and (world > 1 or os.environ.get("BF_BENCH_SHARED_RENDER") == "1")
    t_gen = time.perf_counter()
    if shared_render:
        per = (n_render + world - 1) // world
        mine = list(range(first + rank * per, min(first + (rank + 1) * per, first + n_render)))
        part = synth.render_frames(mine, W, H, workers=max(1, min(64, ncpu // max(world, 1)))) if mine else []
        d_part = torch.zeros((per, H, W), dtype=torch.float32, device="cuda"); c_part = torch.zeros((per, H, W, 4), dtype=torch.uint8, device="cuda")
        for i, f in enumerate(part):
            d_part[i] = torch.from_numpy(f[0]).cuda(); c_part[i] = torch.from_numpy(f[1]).cuda()
        if world > 1:
            d_all = torch.empty((world * per, H, W), dtype=torch.float32, device="cuda"); c_all = torch.empty((world * per, H, W, 4), dtype=torch.uint8, device="cuda")
            dist.all_gather_into_tensor(d_all, d_part); dist.all_gather_into_tensor(c_all, c_part)
        else:
            d_all, c_all = d_part, c_part
        torch.cuda.synchronize()
        Kd_ = synth.intrinsics(W, H)
        # (depth, colour, pose, intrinsics) like synth.scene_room returns them; depth / colour stay on the device
        frames = [(d_all[i], c_all[i], synth.trajectory_pose(first + i), Kd_) for i in range(n_render)]
    else:
        frames = synth.render_frames(range(first, first + n_render), W, H, workers=max(1, min(64, ncpu // max(world, 1))))
    t_gen = time.perf_counter() - t_gen

    import bundlefusion_amd as bf
    from bundlefusion_amd.capi import intrinsics_matrix, default_app_state, default_bundling_state, sensor_desc, bind_host_threads_to_device
    host_cpus = bind_host_threads_to_device(local_rank)       # the host threads (this one, the volume thread, the HIP runtime's) next to the GPU: DESIGN.md 4.4

    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])

    def params(buckets=None, blocks=None):
        gas = default_app_state(); gbs = default_bundling_state()      # zParametersDefault.txt / zParametersBundlingDefault.txt values
        gas.s_integrationWidth, gas.s_integrationHeight = W, H
        gas.s_SDFVoxelSize = args.voxel
        gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = buckets or args.buckets, blocks or args.blocks
        gbs.s_maxNumImages = max(n_render // 10 + 8, 16)
        return gas, gbs

    if shared_render:
        assert not args.host, "the shared rendering keeps the frames on the device"
        feed = [(f[0], f[1]) for f in frames]
    elif args.host:
        feed = [(f[0], f[1]) for f in frames]
    else:
        feed = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    torch.cuda.synchronize()
    if args.clock_warmup > 0:                                # untimed: bring the GPU out of its idle power state (state-free: a scratch matrix product)
        wa = torch.randn(4096, 4096, device="cuda"); wb = torch.randn(4096, 4096, device="cuda")
        tw = time.perf_counter()
        while time.perf_counter() - tw < args.clock_warmup:
            for _ in range(8):
                wa = torch.mm(wa, wb) * 1e-2
            torch.cuda.synchronize()
        del wa, wb

    T0inv = np.linalg.inv(frames[0][2].astype(np.float64))
    gt = np.stack([T0inv @ f[2].astype(np.float64) for f in frames])

    def run_leg(arith, solve_lag=None):
        """The whole measurement (pre-roll, warm-up, K timed steps) with the voxel update under one arithmetic contract."""
        gas, gbs = params()
        pipe = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
        pipe.scene().set_arith(arith)
        pipe.set_volume_batching(args.volume_batching == "on")
        solve_lag = args.solve_lag if solve_lag is None else solve_lag
        if solve_lag >= 0 and not chunked:
            pipe.set_solve_lag(solve_lag)
        if shard_volume and world > 1:
            pipe.set_volume_shard(rank, world)
        runner = None
        if chunked:
            assert not args.host, "chunks mode feeds HBM-resident frames"
            gas_w, gbs_w = params()
            runner = ChunkedRunner(pipe, bf.capi.ChunkWorker(gas_w, gbs_w, sensor_desc(W, H, K)), feed, gbs_w.s_submapSize, rank, world, "cuda")
            assert runner.frames_needed(total) <= len(feed)
        sc = pipe.scene()
        if args.pmc_out:
            sc.kernel_timing(True)                               # profiling run: account every launch of the run (the PMC passes see all of them)
        if chunked:
            runner.advance(pre)
            runner.wait()                                        # the local half running ahead belongs to the pre-roll
        else:
            for k in range(pre):
                if not pipe.process_frame(*feed[k]):
                    raise RuntimeError("pre-roll frame %d not accepted" % k)
        pipe.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sc = pipe.scene()
        if not args.pmc_out:
            sc.kernel_timing(True)                               # HIP events around every voxel-update launch, on the pipeline's stream
        c0 = pipe.counters()
        pipe.host_profile(reset=True)
        pipe.volume_thread_profile(reset=True)
        torch.cuda.synchronize()
        rounds0 = runner.rounds if chunked else 0
        chunks0 = runner.local_runs if chunked else 0
        t0 = time.perf_counter()
        if chunked:
            runner.advance(args.steps)
            runner.wait()                                        # ... and the one started inside the timed window is paid for inside it
        else:
            for k in range(pre, total):
                if not pipe.process_frame(*feed[k]):
                    raise RuntimeError("frame %d not accepted" % k)
        pipe.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        L = {"arith": arith, "c0": c0, "c1": pipe.counters(), "hp": pipe.host_profile(), "vp": pipe.volume_thread_profile(), "solve_lag": 0 if chunked else pipe.solve_lag()}
        L["occ_sum"], L["vis_plain"], L["vis_fused"], L["n_ops"] = sc.kernel_timing_blocks()
        L["n_images"] = sc.kernel_timing_images()
        L["n_launch"], L["kernel_ms"] = sc.kernel_timing_read()
        sc.kernel_timing(False)
        L["elapsed"] = max_over_ranks(elapsed, "cuda")
        L["rounds"] = (runner.rounds - rounds0) if chunked else 0
        L["local_chunks"] = (runner.local_runs - chunks0) if chunked else 0
        if chunked and (L["rounds"] == 0 or L["local_chunks"] == 0):
            raise RuntimeError("the timed window holds no round of the chunk-parallel schedule (all-gathers %d, chunk-local halves run on this rank %d): not a "
                               "whole-loop measurement" % (L["rounds"], L["local_chunks"]))
        if shard_volume and world > 1:      # the replicated global half must have produced ONE trajectory (bit-identical on every rank): RCCL MIN/MAX all-reduce
            traj_dev = torch.from_numpy(np.nan_to_num(pipe.integrated_trajectory(), neginf=-1e30)).cuda()
            assert same_over_ranks(traj_dev), "ranks disagree on the trajectory"
        L["dbg"] = sc.debug_hash()
        traj = pipe.integrated_trajectory()
        valid = np.isfinite(traj[:, 0, 0])
        L["frames_valid"], L["frames_total"] = int(valid.sum()), int(len(traj))
        L["ate"] = float(np.sqrt(np.mean(np.sum((traj[valid][:, :3, 3] - gt[:len(traj)][valid][:, :3, 3]) ** 2, axis=1)))) if valid.any() else None
        if runner is not None:
            runner.close()
        del sc, pipe, runner
        torch.cuda.synchronize()
        return L

    def roofline_of(L):
        # dominant kernel: the TSDF voxel update.  SURVEY.md 8d charges ONE integrate / de-integrate operator N_occ*(512*24+32) + W*H*8 bytes (every voxel of its
        # frustum list read and written, its frame read).  Two accountings are reported:
        #  * `achieved` / `frac` - the bytes this LAUNCH has to move whatever its schedule: every block of its list once (a fused re-integration walks the union of
        #    its two frustum lists once; a batch walks the union of all its operators' lists once) plus every frame image it samples once.  This is what HBM
        #    actually has to deliver, so frac <= 1 and it is comparable with the PMC traffic.
        #  * `achieved_per_operator` / `frac_per_operator` - 8d by the letter: the sum over the launch's operators of their own N_occ*(512*24+32) + W*H*8.  A batched
        #    launch applies up to 21 operators to a block per trip through HBM, so this figure can exceed the HBM peak: the re-reads the serial schedule pays
        #    are gone, not accelerated.
        n_launch, n_ops, n_img = L["n_launch"], L["n_ops"], L["n_images"]
        batched = args.volume_batching == "on" and not chunked_alloc_comm
        bytes_per_launch = ((L["vis_plain"] + L["vis_fused"]) * (512 * 24 + 32) + n_img * W * H * 8) / max(n_launch, 1)
        bytes_per_operator_sum = (L["occ_sum"] * (512 * 24 + 32) + n_ops * W * H * 8) / max(n_launch, 1)
        avg_kernel_s = (L["kernel_ms"] / 1e3) / max(n_launch, 1)
        achieved = bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        achieved_op = bytes_per_operator_sum / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic = pmc_traffic(args, L["arith"], L["vis_plain"], L["vis_fused"], n_launch)
        # the batched fast update applies ~11 operators to a block per trip through HBM: ~900 vector instructions per block and operator slot against 12 KB of
        # traffic - it is bound by vector-instruction issue, not by bytes.  Instruction roofline: wave-instructions issued (SQ_INSTS_VALU of the
        # committed SQ pass, per visited block) and the cycles they were active (SQ_ACTIVE_INST_VALU: 4.2 cycles per instruction measured on this kernel - the guide's
        # 2-cycle full-rate figure is reported beside it as the lower bound) over launch time x clock x SIMDs.
        vi = pmc_valu(args, L["arith"], L["vis_fused"], n_launch) if batched else None
        reserve = int(os.environ.get("BF_VOLUME_CU_RESERVE", "32"))          # the volume stream's launches run on all but `reserve` compute units (bf_pipeline_create)
        simds = SIMDS - 4 * max(0, min(reserve, 255))
        simd_cycles = avg_kernel_s * CLOCK_HZ * simds
        valu = None if not vi else {"wave_instructions_per_launch": vi[0], "active_cycles_per_launch": vi[1], "cycles_per_instruction_measured": vi[1] / vi[0] if vi[0] else None,
                                    "simds": simds, "simd_cycles_available_per_launch": simd_cycles,
                                    "frac": vi[1] / simd_cycles if simd_cycles > 0 else None,                    # measured: SQ_ACTIVE_INST_VALU (x 4 cycles) per visited block x blocks visited here
                                    "frac_at_2_cycles_per_instruction": 2.0 * vi[0] / simd_cycles if simd_cycles > 0 else None,      # the guide's full-rate issue cost: a lower bound
                                    # the same share inside the PMC run itself (all its launches, its own device cycles: GRBM_GUI_ACTIVE / 8 XCDs of the TD pass) - free of the
                                    # per-visited-block scaling above, which spreads the run's instructions evenly over blocks although the window's blocks carry more operators
                                    "frac_in_the_pmc_run": vi[2],
                                    "unit": "share of the vector-issue cycles of the SIMDs the launch may use"}
        # ... and the other unit the kernel keeps busy: every CU's vector-memory return path (16 divergent 8-byte texel gathers per block and operator slot, ~19 L1
        # accesses each).  From the TD / TCP passes of the same PMC file: busy cycles per visited block x the blocks visited here, over launch time x clock x CUs.
        mp = pmc_mem(args, L["arith"], L["vis_fused"], n_launch) if batched else None
        cus = simds // 4
        mem_pipe = None if not mp else {"td_busy_cycles_per_launch": mp["td_busy_cycles_per_launch"], "compute_units": cus,
                                        "td_busy_frac": mp["td_busy_cycles_per_launch"] / (avg_kernel_s * CLOCK_HZ * cus) if avg_kernel_s > 0 else None,
                                        "td_busy_frac_in_the_pmc_run": mp["td_busy_share_of_cu_cycles"], "td_stalled_on_l1_frac_in_the_pmc_run": mp["td_stalled_on_l1_share_of_cu_cycles"],
                                        "l1_accesses_per_vmem_instruction": mp["l1_accesses_per_vmem_instruction"],
                                        "unit": "share of the cycles of the CUs the launch may use during which the texture-data unit (vector-memory return path) is busy"}
        if batched:
            kern = ("k_update_batch_apx (tsdf_batch.h): one wave per block of the batch's union list, voxels loaded once, the batch's operators applied in order from registers"
                    if L["arith"] == "fast" else "k_update_batch_col (tsdf_batch.h): the batch's operators one after the other per block, exact contract")
        else:
            kern = "k_update_apx<2> (fused de-integrate + integrate) + k_update_apx<0> (integrate)" if L["arith"] == "fast" else \
                   "k_update_col<2> (fused de-integrate + integrate) + k_update_col<0> (integrate)"
        return {
            "kernel": kern + " - TSDF voxel update, tsdf.hip",
            # `bound`: what limits the kernel.  achieved / peak / frac stay the HBM figures of the contract (algorithmic bytes over launch time against 8 TB/s); for
            # the VALU-bound batched update they say how far the kernel is from the byte roofline it no longer touches, `valu` how close to the one it does.
            "bound": "valu" if (batched and L["arith"] == "fast") else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu": valu, "mem_pipe": mem_pipe,
            "achieved_per_operator": achieved_op, "frac_per_operator": achieved_op / HBM_PEAK_GBS,
            "hbm_frac_measured": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_kernel_s > 0) else None,
            "launches": n_launch, "operators": n_ops, "frames_sampled": n_img, "avg_launch_us": 1e6 * avg_kernel_s,
            "us_per_operator": 1e6 * (L["kernel_ms"] / 1e3) / max(n_ops, 1),
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_launch_per_operator_accounting": bytes_per_operator_sum,
            "ops_per_launch": n_ops / max(n_launch, 1), "n_occ_mean_per_op": L["occ_sum"] / max(n_ops, 1),
            "blocks_visited_per_launch": (L["vis_plain"] + L["vis_fused"]) / max(n_launch, 1),
            "accounting": "achieved = (blocks of the launch's list x (512*24+32) + frames sampled x W*H*8) / launch time: each block and each frame once per launch (union list of a "
                          "fused re-integration / of a batch); achieved_per_operator = SURVEY 8d by the letter, summed over the launch's operators; traffic = PMC bytes per visited "
                          "block (profiles/%s: FETCH_SIZE x the factor measured on this access pattern with known byte counts [tools/pmc_calibrate.py, recorded in that file] + WRITE_SIZE, own passes, same contract, same kernel source: update_kernel_sha256) x blocks visited "
                          "here; null when that file was collected on another version of the kernel" % PMC_FILE,
            "share_of_step_time": (L["kernel_ms"] / 1e3) / L["elapsed"] if L["elapsed"] > 0 else None,
        }

    main_leg = run_leg(args.arith)
    other = None
    if args.both_contracts and world == 1 and not args.pmc_out:
        other = run_leg("exact" if args.arith == "fast" else "fast")
    serial = None
    if args.both_contracts and world == 1 and not args.pmc_out and main_leg["solve_lag"] != 0 and not chunked:
        # the same window in the serial order of the reference's single-threaded branch (the schedule of rounds 1-5's `value`), beside the default
        serial = run_leg(args.arith, solve_lag=0)
    if args.pmc_out and rank == 0:
        json.dump({"config": pmc_config(args), "launches": main_leg["n_launch"],
                   "fused_launches": main_leg["n_launch"] if args.volume_batching == "on" else main_leg["n_ops"] - main_leg["n_launch"],      # launches over a union list
                   "visited_blocks_plain": main_leg["vis_plain"], "visited_blocks_fused": main_leg["vis_fused"], "operator_blocks": main_leg["occ_sum"]}, open(args.pmc_out, "w"))
    elapsed, c0, c1, hp, dbg = main_leg["elapsed"], main_leg["c0"], main_leg["c1"], main_leg["hp"], main_leg["dbg"]
    ate = main_leg["ate"]

    if rank == 0:
        out = {
            "metric": "frames/sec end-to-end (SIFT+SBA+TSDF re-integrate), 640x480 @4mm",
            "value": (1 if one_stream else world) * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if (one_stream and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] stand-in: S2 room stream %dx%d @%.0f mm voxels through the full frame loop (ingest, SIFT, "
                            "match+filters, local+global GN/PCG, TSDF integrate + re-integration + GC); 1 step = 1 input frame; timed: frames "
                            "%d..%d of the stream after an untimed pre-roll of %d frames incl. %d warm-up frames (re-integration queue saturated, %d key "
                            "frames in the global problem)" % (W, H, args.voxel * 1e3, first + pre, first + total - 1, pre, args.warmup, pre // 10),
                "input": "host buffers per frame (PCIe inclusive)" if args.host else "frames resident in HBM",
                "clock_warmup_s": args.clock_warmup,
                "host_threads": ("bound to the GPU's NUMA node (CPUs %s)" % host_cpus) if host_cpus else "not bound (topology unknown or BF_BIND_NUMA=0)",
                "params": "zParametersDefault.txt + zParametersBundlingDefault.txt values; s_integrationWidth/Height=640/480, "
                          "s_SDFVoxelSize=%.3f, s_hashNumBuckets=%d, s_hashNumSDFBlocks=%d" % (args.voxel, args.buckets, args.blocks),
                "timed_ops": {k: c1[k] - c0[k] for k in c1},
                "frames_valid": main_leg["frames_valid"], "frames_total": main_leg["frames_total"], "ate_rmse_vs_ground_truth_m": ate,
                "blocks_allocated": dbg["occupied"], "blocks_dropped": dbg["dropped"],
                "render_seconds_untimed": round(t_gen, 1),
                "host_thread_ms_per_frame": {k: round(1e3 * v / max(hp["frames"], 1.0), 4) for k, v in hp.items() if k != "frames"},
                "volume_thread": {"busy_share_of_wall": round(main_leg["vp"]["busy_seconds"] / elapsed, 3), "operators": int(main_leg["vp"]["operators"]),
                                  "us_of_api_calls_per_operator": round(1e6 * main_leg["vp"]["busy_seconds"] / max(main_leg["vp"]["operators"], 1.0), 1)},
                "frame_loop": "two frames behind the input: the matching chain of frame k+1 is enqueued before frame k's result is read back, detection runs one frame "
                              "further ahead (BF_PIPELINE_LOOKAHEAD=%s); chunk solves: %s" % (os.environ.get("BF_PIPELINE_LOOKAHEAD", "1"),
                              ("own thread + stream, applied exactly %d frames after the chunk's last frame (the reference's optimiser thread, FriedLiver.cpp:112-143, made "
                               "deterministic; the library's default; parity: test_lagged_solve_mode_vs_oracle_loop_with_the_same_lag)" % main_leg["solve_lag"]) if main_leg["solve_lag"]
                              else "serial order (inside the frame that closes the chunk)"),
                "solve_lag": main_leg["solve_lag"],
                "parallelism": ("one stream: local chunks round-robin over %d ranks, %d RCCL all-gathers of key-frame packages in the timed region, global half "
                                "replicated, volume sharded by hash-bucket range; the timed window holds the global half of its %d frames and, on every rank, "
                                "the chunk-local half of ONE whole chunk of the next round (%d of them ran here) - the steady state when steps == 10 x ranks, "
                                "more local work per frame than the steady state when steps is smaller"
                                % (world, main_leg["rounds"], args.steps, main_leg["local_chunks"])) if chunked
                               else ("one stream, bundling replicated on %d ranks, volume sharded by hash-bucket range" % world) if (shard_volume and world > 1)
                               else "one GPU, serial frame loop" if world == 1 else "stream segments sharded over %d rank(s), no data-path collective" % world,
                "mode": mode,
            },
            "roofline": roofline_of(main_leg),
        }
        out["config"]["arith"] = ("%s: the voxel update under the arithmetic contract of the reference's own GPU build (FriedLiver.vcxproj:124 FastMath: approximate "
                                  "division, FMA contraction); same block set / occupancy / weights as the exact contract, sdf 1e-5 x truncation, colour 1 LSB per operator "
                                  "(tests/test_tsdf_fast_gpu.py)" % main_leg["arith"]) if main_leg["arith"] == "fast" else \
                                 "exact: every operation of the voxel update IEEE op by op, bit-identical with the oracle (tests/test_tsdf_gpu.py)"
        if other is not None:
            out["other_contract"] = {"arith": other["arith"], "value": args.steps / other["elapsed"], "unit": "frames/s", "ms_per_step": 1e3 * other["elapsed"] / args.steps,
                                     "roofline": roofline_of(other), "timed_ops": {k: other["c1"][k] - other["c0"][k] for k in other["c1"]},
                                     "same_trajectory": other["ate"] == main_leg["ate"]}
        if serial is not None:
            out["serial_order"] = {"solve_lag_frames": 0, "value": args.steps / serial["elapsed"], "unit": "frames/s", "ms_per_step": 1e3 * serial["elapsed"] / args.steps,
                                   "timed_ops": {k: serial["c1"][k] - serial["c0"][k] for k in serial["c1"]}, "roofline": roofline_of(serial),
                                   "note": "the same window with the chunk solves inside the frame that closes the chunk (BF_PIPELINE_SOLVE_LAG=0): the reference's single-threaded "
                                           "branch, the schedule of every earlier round's `value`, and the one the oracle loop / the compiled reference loop are compared in"}
        if not args.no_class_surface and world == 1 and not args.pmc_out and not args.host:
            try:
                out["class_surface"] = class_surface_block(frames, pre, total, args, Kd, W, H, out["value"])
            except Exception as e:      # noqa: BLE001 - reported in the line
                out["class_surface"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(frames[:args.cpu_frames], feed[:args.cpu_frames], params, K, W, H, args.arith)
    del frames, feed
    # The secondary blocks must not cost the line its headline: an exception becomes {"error": ...}; a collective that never returns (the sweep's RCCL
    # communicator is created here, after the measurement) is cut off by a watchdog that prints the line without the block and ends every rank.
    import threading
    line_lock = threading.Lock()
    line_state = {"printed": False}

    def print_line():             # the ONE JSON line: whoever gets here first prints it (the main path, or a watchdog that gives a secondary block up)
        with line_lock:
            if line_state["printed"]:
                return
            line_state["printed"] = True
            if rank == 0:
                print(json.dumps(out), flush=True)

    def secondary(name, fn, limit_s):
        state = {"done": False}

        def give_up():
            with line_lock:
                if state["done"] or line_state["printed"]:
                    return
                if rank == 0:
                    out[name] = {"error": "no result after %d s (abandoned; the headline measurement above it is complete)" % limit_s}
            print_line()
            os._exit(3)           # a block that never returned (a collective that hung): the line is out, the exit status says the run was cut short
        wd = threading.Timer(limit_s, give_up); wd.daemon = True; wd.start()
        try:
            r = fn()
        except Exception as e:      # noqa: BLE001 - reported in the line
            r = {"error": "%s: %s" % (type(e).__name__, e)}
        wd.cancel()
        with line_lock:
            state["done"] = True
        return r
    if not args.no_sweep:
        sw = secondary("sweep", lambda: sweep_block(args, rank, world), 300)
        if rank == 0:
            with line_lock:
                out["sweep"] = sw
                # the same update kernels on a volume far beyond the 256 MB Infinity Cache (1280x960 @2 mm, ~6 GB of blocks walked per sweep): what HBM itself delivers
                g = (sw or {}).get("algorithmic_GBps_of_update_kernel_rank0") if isinstance(sw, dict) else None
                out["roofline"]["frac_beyond_l3"] = (g / HBM_PEAK_GBS) if g else None
    if args.long_stream and world == 1:
        ls = secondary("long_stream", lambda: long_stream_block(args, K, W, H), 1500)
        with line_lock:
            out["long_stream"] = ls
    print_line()
    if world > 1:
        dist.destroy_process_group()


def visible_devices():
    """GPUs this process can use (torch's count, which honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES), or None if torch cannot say"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:      # noqa: BLE001
        return None


def launch_plan(gpus, env, argv):
    """The command that starts `gpus` ranks of this script, or None when this process already is one of them (a launcher set WORLD_SIZE) or
    one GPU is asked for.  The driver launches N > 1 itself (torch.distributed.run); a bare `python bench.py --gpus N` gets the same launch."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def sweep_block(args, rank, world):
    """BASELINE configs[4] in small (DepthSensing.cpp:854-902 is the loop it stands for): 24 frames of the room at 1280x960 integrated into a
    2 mm volume, then every frame re-integrated at a perturbed pose (xi ~ N(0, diag(0.01 rad, 0.01 m)), seed 777) with the fused operator - the volume operators
    alone, no bundling.  With N ranks the ONE volume is sharded by hash-bucket range and the allocation's ray march is divided over the ranks inside the operators
    (bf_scene_set_alloc_comm: one RCCL all-gather of block keys per operator on the allocation stream); value = fused re-integrations of the one volume per second
    (strong scaling) - the number north_star's ">= 6x at 8 GPUs on the TSDF re-integration sweep" is about."""
    from tools.tsdf_sweep import parser as sweep_parser, run as sweep_run
    argv = ["--frames", "24", "--stride", "10", "--width", "1280", "--height", "960", "--voxel", "0.002", "--buckets", "4000000", "--blocks", "1500000", "--sweeps", "1"]
    if world > 1:
        argv.append("--comm-alloc")
    sa = sweep_parser().parse_args(argv)
    sa.arith = args.arith
    r = sweep_run(sa, rank, world)
    if r is not None:
        r["metric"] = "fused re-integrations/s of one 1280x960 @2 mm volume"
        r["value"] = r["reintegrations_per_s"]
        r["n_gpus"] = world
    return r


def long_stream_block(args, K, W, H):
    """A whole stream from frame 0 through a fresh pipeline (frames resident in HBM, rendered and uploaded in batches): frames/s overall and per 1000 frames, tracked
    frames, ATE of the integrated and of the optimised trajectory, key frames, solve and operation counts."""
    import numpy as np
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd import synth
    from bundlefusion_amd.capi import default_app_state, default_bundling_state, sensor_desc
    n = args.long_stream
    bob = 0.3 if n > 2000 else 0.0                      # config 4 = 2.5 loops + vertical sinusoid, config 3 = the plain loop (frame 1800 closes it)
    gas = default_app_state(); gbs = default_bundling_state()
    gas.s_integrationWidth, gas.s_integrationHeight = W, H
    gas.s_SDFVoxelSize = args.voxel
    gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 4000000, 3000000
    gbs.s_maxNumImages = n // 10 + 8
    pipe = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
    pipe.scene().set_arith(args.arith)
    if args.solve_lag >= 0:
        pipe.set_solve_lag(args.solve_lag)
    poses, marks, dev = [], [], []
    for c0 in range(0, n, 500):                                       # render and upload in batches (host memory); the whole stream is resident in HBM before the clock starts (5000 frames = 12 GB)
        part = synth.render_frames(range(c0, min(c0 + 500, n)), W, H, bob=bob)
        dev += [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in part]
        poses += [f[2] for f in part]
        del part
    torch.cuda.synchronize()
    if args.clock_warmup > 0:                                         # the GPU idled while the host rendered
        wa = torch.randn(4096, 4096, device="cuda"); wb = torch.randn(4096, 4096, device="cuda")
        tw = time.perf_counter()
        while time.perf_counter() - tw < args.clock_warmup:
            for _ in range(8):
                wa = torch.mm(wa, wb) * 1e-2
            torch.cuda.synchronize()
        del wa, wb
    t0 = tm = time.perf_counter()
    for k, (d, c) in enumerate(dev):
        if not pipe.process_frame(d, c):
            raise RuntimeError("long stream: frame not accepted")
        if (k + 1) % 500 == 0 or k + 1 == n:                         # one continuous run; the marks synchronise the pipeline (the frames in flight complete: ~3 ms per 500 frames)
            pipe.synchronize(); torch.cuda.synchronize()
            now = time.perf_counter()
            first_k = (k // 500) * 500
            hp = pipe.host_profile(reset=True); vp = pipe.volume_thread_profile(reset=True)
            marks.append({"frames": "%d-%d" % (first_k, k), "fps": round((k + 1 - first_k) / (now - tm), 1),
                          "host_thread_ms_per_frame": {kk: round(1e3 * v / max(hp["frames"], 1.0), 3) for kk, v in hp.items() if kk != "frames"},      # where the calling thread's time goes as the global problem grows
                          "volume_thread_busy_share": round(vp["busy_seconds"] / max(now - tm, 1e-9), 3)})
            tm = now
    t_run = time.perf_counter() - t0
    done = n
    del dev
    T0inv = np.linalg.inv(poses[0].astype(np.float64))
    gt = np.stack([T0inv @ T.astype(np.float64) for T in poses])

    def ate(t):
        v = np.isfinite(t[:, 0, 0])
        return (float(np.sqrt(np.mean(np.sum((t[v][:, :3, 3] - gt[:len(t)][v][:, :3, 3]) ** 2, axis=1)))) if v.any() else None), int(v.sum())
    a_int, v_int = ate(pipe.integrated_trajectory())
    a_opt, v_opt = ate(pipe.optimized_trajectory())
    c = pipe.counters()
    dbg = pipe.scene().debug_hash()
    first, last = marks[0]["fps"], marks[-1]["fps"]
    return {"frames": n, "stream": "S2 room from frame 0, stride 1%s" % (", vertical sinusoid 0.3 m" if bob else ""), "value": done / t_run, "unit": "frames/s", "per_500_frames": marks,
            "last_over_first": round(last / first, 3), "frames_tracked": v_int, "ate_integrated_m": a_int, "ate_optimized_m": a_opt, "frames_with_optimized_pose": v_opt,
            "key_frames": (n - 1) // 10, "counters": {k: int(v) for k, v in c.items()}, "blocks_allocated": dbg["occupied"], "blocks_dropped": dbg["dropped"],
            "timing": "one continuous run over the whole stream, all frames resident in HBM before the clock starts (untimed: rendering, upload, clock warm-up); the pipeline is synchronised every 500 frames for the marks"}


def class_surface_block(frames, pre, total, args, Kd, W, H, pipeline_fps):
    """The same pre-roll + window through the drop-in CLASS surface: examples/class_surface_bench.cpp is DepthSensing.cpp's serial frame loop written against
    include/bundlefusion/bundlefusion.hpp (CUDAImageManager / OnlineBundler / TrajectoryManager / CUDASceneRepHashSDF), one frame at a time, host frames in
    (the class contract: RGBDSensor hands host buffers over), nothing in flight across frames.  Run twice: the calls as the reference issues them, and with the
    wrapper's deferred batching (one bf_scene_run_batch per frame).  The executable is built by bundlefusion_amd.build (g++, no HIP headers)."""
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "bundlefusion_amd", "lib", "class_surface_bench")
    if not os.path.exists(exe):
        return {"error": "bundlefusion_amd/lib/class_surface_bench has not been built (python -m bundlefusion_amd.build)"}
    res = {"loop": "examples/class_surface_bench.cpp: DepthSensing.cpp:966-1095 + :854-902 against the reference's class names; host frames per call (PCIe inclusive), one frame at a time",
           "pipeline_value": pipeline_fps}
    with tempfile.NamedTemporaryFile(prefix="bf_frames_", suffix=".bin", dir="/tmp", delete=True) as f:
        for fr in frames[:total]:
            d, c = fr[0], fr[1]
            d = d.cpu().numpy() if hasattr(d, "cpu") else d
            c = c.cpu().numpy() if hasattr(c, "cpu") else c
            f.write(np.ascontiguousarray(d, np.float32).tobytes()); f.write(np.ascontiguousarray(c, np.uint8).tobytes())
        f.flush()
        for deferred in (0, 1):
            r = subprocess.run([exe, f.name, str(W), str(H), str(total), str(pre), repr(args.voxel), str(args.buckets), str(args.blocks),
                                repr(float(Kd["fx"])), repr(float(Kd["fy"])), repr(float(Kd["mx"])), repr(float(Kd["my"])), str(deferred)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            res["deferred_batching" if deferred else "as_the_reference_issues_them"] = json.loads(line[-1]) if line else {"error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
    best = res.get("deferred_batching", {}).get("value")
    res["share_of_pipeline_value"] = (best / pipeline_fps) if (best and pipeline_fps) else None
    return res


def pmc_config(args):
    return {"preroll": args.preroll, "voxel": args.voxel, "buckets": args.buckets, "blocks": args.blocks, "host": bool(args.host), "volume_batching": args.volume_batching}


def pmc_traffic(args, arith, vis_plain, vis_fused, n_launch):
    """HBM bytes per voxel-update launch of THIS run, from the committed PMC passes (profiles/r04_pmc_tsdf_update.json: rocprofv3
    FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, each in its own run of this command with --pmc-out): bytes per visited SDF block
    of the plain and of the fused kernel under the same arithmetic contract, times the blocks the timed launches of this run visited.
    None when the counters were collected on another configuration (pre-roll / volume parameters / contract)."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(path) or n_launch == 0:
        return None
    pmc = json.load(open(path)).get(arith)
    if not pmc or pmc["config"] != pmc_config(args):
        return None
    from tools.pmc_to_json import update_kernel_sha, build_flags_sha
    if pmc.get("update_kernel_sha256") != update_kernel_sha() or pmc.get("build_flags_sha256") != build_flags_sha():          # counters of another version / another build of the update kernels: stale, not reported
        return None
    return (vis_fused * pmc["fused"]["hbm_bytes_per_visited_block"] + vis_plain * pmc["plain"]["hbm_bytes_per_visited_block"]) / n_launch


def pmc_valu(args, arith, vis_fused, n_launch):
    """Vector wave-instructions per voxel-update launch of THIS run, from the SQ pass of the same committed PMC file (SQ_INSTS_VALU per visited block x the blocks
    visited here); None under the same conditions as pmc_traffic."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(path) or n_launch == 0:
        return None
    pmc = json.load(open(path)).get(arith)
    if not pmc or pmc["config"] != pmc_config(args) or "sq" not in pmc:
        return None
    from tools.pmc_to_json import update_kernel_sha, build_flags_sha
    if pmc.get("update_kernel_sha256") != update_kernel_sha() or pmc.get("build_flags_sha256") != build_flags_sha():
        return None
    mp = pmc.get("mem_pipe")
    in_run = None
    if mp and mp.get("device_cycles_per_launch"):
        in_run = 4.0 * pmc["sq"]["per_launch"]["SQ_ACTIVE_INST_VALU"] / (mp["device_cycles_per_launch"] * 4.0 * mp["compute_units"])
    return (vis_fused * pmc["sq"]["valu_wave_instructions_per_visited_block"] / n_launch, vis_fused * pmc["sq"].get("valu_active_cycles_per_visited_block", 0.0) / n_launch, in_run)


def pmc_mem(args, arith, vis_fused, n_launch):
    """The vector-memory return path of the batched update, from the TD / TCP / VMEM passes of the same committed PMC file; None under the same conditions as pmc_traffic."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(path) or n_launch == 0:
        return None
    pmc = json.load(open(path)).get(arith)
    if not pmc or pmc["config"] != pmc_config(args) or "mem_pipe" not in pmc:
        return None
    from tools.pmc_to_json import update_kernel_sha, build_flags_sha
    if pmc.get("update_kernel_sha256") != update_kernel_sha() or pmc.get("build_flags_sha256") != build_flags_sha():
        return None
    m = dict(pmc["mem_pipe"])
    m["td_busy_cycles_per_launch"] = vis_fused * m["td_busy_cycles_per_visited_block"] / n_launch
    return m


def cpu_baseline(frames, feed, params, K, W, H, arith):
    """The oracle frame loop (kind 'port': the reference has no runnable CPU path) on this box's host cores, on a bounded
    sample of the same stream: its first three local chunks from frame 0 (31 frames: SIFT, matching, filters, TSDF integration at 4 mm,
    three local solves, two global matchings + solves; from frame 21 on every frame also carries s_maxFrameFixes = 10 re-integrations,
    the state the GPU leg's timed window is in).  N threads: the voxel update and the image-space loops (ingest filters, cache frame,
    SIFT pyramid) run on all cores (OpenMP over independent blocks / pixels), median of 3 runs; 1 thread: one run of the first chunk.
    `gpu_same_sample` is the HIP path on exactly the same 31 frames from a fresh pipeline (frames resident in HBM), so that the ratio
    compares the same work."""
    import statistics
    import torch
    import bundlefusion_amd as bf
    from bundlefusion_amd.capi import sensor_desc
    from tests import oracle_api
    from tests.oracle_pipeline import OraclePipeline

    frames = [(f[0].cpu().numpy() if hasattr(f[0], "cpu") else f[0], f[1].cpu().numpy() if hasattr(f[1], "cpu") else f[1], f[2], f[3]) for f in frames]

    def run(n_frames, threads):
        gas, gbs = params(400000, 250000)      # three chunks touch < 200k blocks; a smaller heap keeps the host allocation out of the timing
        op = OraclePipeline(gas, gbs, W, H, K)
        if threads is not None:
            op.threads = threads
            oracle_api.set_threads(threads)
        t0 = time.perf_counter()
        for d, c, _, _ in frames[:n_frames]:
            op.process_frame(d, c)
        dt = time.perf_counter() - t0
        n_ops = len(op.integrate_ops)
        th = op.threads
        del op
        return n_frames / dt, dt, n_ops, th

    runs = [run(len(frames), None) for _ in range(3)]
    fps_n = statistics.median(r[0] for r in runs)
    n1 = min(11, len(frames))
    fps_1, dt_1, _, _ = run(n1, 1)
    # the GPU on the same sample
    gas, gbs = params(400000, 250000)
    pipe = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
    pipe.scene().set_arith(arith)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d, c in feed:
        if not pipe.process_frame(d, c):
            raise RuntimeError("frame not accepted")
    pipe.synchronize(); torch.cuda.synchronize()
    gpu_fps = len(feed) / (time.perf_counter() - t0)
    c = pipe.counters()
    del pipe
    return {"value": fps_n, "unit": "frames/s", "cores": runs[0][3], "kind": "port",
            "single_thread": {"value": fps_1, "unit": "frames/s", "cores": 1, "sample": "first %d frames (one local chunk), one run, %.1f s" % (n1, dt_1)},
            "gpu_same_sample": {"value": gpu_fps, "unit": "frames/s", "timed_ops": {k: int(v) for k, v in c.items()}},
            "sample": "first %d frames of the same stream from frame 0 (three local chunks incl. their solves, two global solves, %d TSDF operators: the last 10 "
                      "frames with 10 re-integrations each), median of 3 runs (%s s); stage kernels in C++, voxel update + image-space loops on %d OpenMP threads, "
                      "matching / filters / solver / orchestration single-threaded" % (len(frames), runs[0][2], ", ".join("%.1f" % r[1] for r in runs), runs[0][3])}


if __name__ == "__main__":
    main()
