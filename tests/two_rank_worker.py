"""One rank of the chunk-parallel mode with REAL objects (capi.Pipeline, capi.ChunkWorker, shard.ChunkedRunner) - run by
tests/test_two_rank_gpu.py as `python tests/two_rank_worker.py <out.npz> <frames> <width> <height>` with RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the environment.  The process group is gloo (the package all-gather goes through host memory), so any number of ranks
can share the one GPU of the test box; on a multi-GPU node the same code runs with backend nccl and one GPU per rank (bench.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out, n, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from bundlefusion_amd import synth
    src = synth.render_frames(range(n), W, H, workers=4)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bundlefusion_amd as bf
    from bundlefusion_amd import shard
    from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix, sensor_desc, FREE_ENTRY, VOX_PER_BLOCK

    def params():
        gas = default_app_state(); gbs = default_bundling_state()
        gas.s_integrationWidth, gas.s_integrationHeight = W, H
        gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.02, 50000, 20000
        gbs.s_widthSIFT, gbs.s_heightSIFT, gbs.s_maxNumImages = W, H, 8
        return gas, gbs
    Kd = src[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in src]
    gas, gbs = params()
    pipe = bf.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
    pipe.set_volume_shard(rank, world)
    # Everything that crosses ranks goes through the C ABI's communicator (include/bf_comm.h) - here with torch.distributed/gloo behind its callback transport,
    # RCCL on a multi-GPU node: the key-frame packages of a round (bf_chunk_exchange), and - issued by the pipeline's VOLUME THREAD, on the volume's
    # allocation stream, once per integrate / re-integrate - the block keys of the divided ray march (bf_pipeline_set_comm).  The volume thread gets its
    # own process group: its collectives must not interleave with the main thread's.
    vol_group = dist.new_group(backend="gloo")
    comm_pkg, comm_vol = bf.capi.Comm.torch_group(None), bf.capi.Comm.torch_group(vol_group)
    if os.environ.get("BF_TEST_DIVIDED_ALLOC", "1") == "1":
        pipe.set_comm(comm_vol, 1 << 14)
    worker = bf.capi.ChunkWorker(*params(), sensor_desc(W, H, K))
    runner = shard.ChunkedRunner(pipe, worker, dev, gbs.s_submapSize, rank, world, device=None, comm=comm_pkg)
    assert runner.advance(n) == n
    runner.close()
    for _ in range(3):
        pipe.process_end_of_sequence()
    pipe.synchronize()
    same = shard.same_over_ranks(torch.from_numpy(np.nan_to_num(pipe.integrated_trajectory(), neginf=-1e30)))      # one MIN/MAX all-reduce (gloo here, RCCL in bench.py)
    gh, gheap, gcnt, gvox = pipe.scene().download()
    occ = gh[gh["ptr"] != FREE_ENTRY]
    keys = np.array([e["pos"] for e in occ], np.int32).reshape(-1, 3)
    vox = np.stack([gvox[int(e["ptr"]):int(e["ptr"]) + VOX_PER_BLOCK].view(np.uint8) for e in occ]) if len(occ) else np.zeros((0, VOX_PER_BLOCK * 12), np.uint8)
    c = pipe.counters()
    np.savez(out, integrated=pipe.integrated_trajectory(), optimized=pipe.optimized_trajectory(), keys=keys, vox=vox, same=int(same),
             counters=np.array([c["integrate"], c["deintegrate"], c["local_solves"], c["global_solves"]]), rounds=runner.rounds, local_chunks=runner.local_chunks)
    dist.barrier()
    pipe.set_comm(None)
    del runner, pipe
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
