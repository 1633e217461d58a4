"""GPU parity tests (-m gpu): HIP voxel-hash TSDF through the C ABI vs the CPU oracle.

Bar: bit-exact for the hash table image (keys, slots, chain offsets, heap, heap counter, frustum
list) and — because both sides evaluate the same IEEE op sequence without FMA contraction —
bit-exact voxel bytes (sdf, weight, colour) as well.  tol = 0.
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params, FREE_ENTRY, VOX_PER_BLOCK

pytestmark = pytest.mark.gpu


def _to_dev(depth, color):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(depth)).cuda()
    c = torch.from_numpy(np.ascontiguousarray(color)).cuda() if color is not None else None
    return d, c


def assert_same_state(gs, osc, what=""):
    gh, gheap, gcnt, gvox = gs.download()
    oh, oheap, ocnt, ovox = osc.hash(), osc.heap(), osc.heap_counter(), osc.voxels()
    assert gcnt == ocnt, what + " heap counter"
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(gh[f], oh[f]), what + " hash." + f
    assert np.array_equal(gheap, oheap), what + " heap"
    assert np.array_equal(gvox.view(np.uint8), ovox.view(np.uint8)), what + " voxel bytes"
    gc, oc = gs.download_compactified(), osc.compactified()
    assert len(gc) == len(oc) == osc.num_occupied(), what + " numOccupiedBlocks"
    # the frustum list as a SET (since round 4 the list pass appends tile ranges in arrival order - the reference's own compactify appends with atomicAdd; the
    # order is not part of any result): same entries, here sorted by block pointer
    go, oo = np.argsort(gc["ptr"], kind="stable"), np.argsort(oc["ptr"], kind="stable")
    assert np.array_equal(gc["pos"][go], oc["pos"][oo]) and np.array_equal(gc["ptr"][go], oc["ptr"][oo]), what + " frustum list"
    dbg = gs.debug_hash()
    assert dbg["duplicate_keys"] == 0 and dbg["free_and_allocated"] == 0 and dbg["leaked"] == 0
    assert dbg["dropped"] == osc.num_dropped(), what + " dropped"
    assert gs.num_allocated_blocks() == osc.num_allocated()


def _setup(width, height, voxel, buckets, blocks, scene="wall", k=0):
    if scene == "wall":
        depth, color, T, K = synth.scene_wall(width, height)
    else:
        depth, color, T, K = synth.scene_room(k, width, height)
    cam = camera_params(width, height, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=buckets, num_sdf_blocks=blocks, voxel_size=voxel)
    return depth, color, T, cam, p


def test_integrate_bit_exact_small(gpu, oracle):
    depth, color, T, cam, p = _setup(160, 120, 0.01, 20000, 6000)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    d, c = _to_dev(depth, color)
    gs.integrate(T, d, c, cam)
    osc.integrate(T, depth, color, cam)
    assert_same_state(gs, osc, "after integrate:")
    assert gs.num_integrated_frames() == 1
    assert gs.hash_params().m_numOccupiedBlocks == osc.num_occupied()
    assert gs.heap_free_count() == osc.heap_counter() + 1


def test_sequence_integrate_deintegrate_gc_bit_exact(gpu, oracle):
    """Moving camera over S2: integrate 6 frames, re-integrate 2 at perturbed poses, de-integrate, GC."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 15, W, H) for k in range(6)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    for i, (depth, color, T, _) in enumerate(frames):
        gs.integrate(T, dev[i][0], dev[i][1], cam)
        osc.integrate(T, depth, color, cam)
    assert_same_state(gs, osc, "after 6 integrates:")
    for i in (1, 4):      # reintegrate() of DepthSensing.cpp:882-889
        depth, color, T, _ = frames[i]
        T2 = T.copy()
        T2[:3, 3] += np.float32(0.03)
        gs.deintegrate(T, dev[i][0], dev[i][1], cam)
        gs.integrate(T2, dev[i][0], dev[i][1], cam)
        osc.deintegrate(T, depth, color, cam)
        osc.integrate(T2, depth, color, cam)
        frames[i] = (depth, color, T2, None)
    gs.garbage_collect()
    osc.garbage_collect()
    assert_same_state(gs, osc, "after re-integration + GC:")
    for i, (depth, color, T, _) in enumerate(frames):
        gs.deintegrate(T, dev[i][0], dev[i][1], cam)
        osc.deintegrate(T, depth, color, cam)
        gs.garbage_collect()
        osc.garbage_collect()
    assert_same_state(gs, osc, "after de-integrating everything:")
    gh, gheap, gcnt, gvox = gs.download()
    assert gs.num_allocated_blocks() == 0 and gcnt + 1 == p.m_numSDFBlocks
    assert np.all(gh["ptr"] == FREE_ENTRY)
    assert not gvox.view(np.uint8).any()


@pytest.mark.parametrize("overlap", [False, True])
def test_fused_reintegrate_equals_deintegrate_plus_integrate(gpu, oracle, overlap):
    """bf_scene_reintegrate (one pass over the union of both frustum lists) vs the oracle's deIntegrate + integrate
    (DepthSensing.cpp:882-889): small and large pose changes (frusta overlapping fully, partly, hardly), followed by GC.
    overlap=True additionally software-pipelines consecutive operators (allocation + frustum list of operator n+1 on the
    internal stream while operator n updates voxels) — same results."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 12, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    gs = gpu.capi.SceneRepHashSDF(p)
    if overlap:
        gs.set_overlap(True)
    osc = oracle.OracleScene(p)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    poses = [f[2].copy() for f in frames]
    for i, (depth, color, T, _) in enumerate(frames):
        gs.integrate(T, dev[i][0], dev[i][1], cam)
        osc.integrate(T, depth, color, cam)
    rng = np.random.default_rng(3)
    for step, (i, dt, drot) in enumerate([(1, 0.002, 0.0), (3, 0.05, 0.0), (0, 0.4, 0.0), (2, 0.0, 0.6), (4, 0.01, 0.01), (1, 0.02, 0.0)]):
        depth, color, _, _ = frames[i]
        T2 = poses[i].copy()
        T2[:3, 3] += (rng.normal(size=3) * dt).astype(np.float32)
        if drot:
            c_, s_ = np.float32(np.cos(drot)), np.float32(np.sin(drot))
            R = np.array([[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]], np.float32)
            T2[:3, :3] = R @ T2[:3, :3]
        gs.reintegrate(poses[i], T2, dev[i][0], dev[i][1], cam)
        osc.deintegrate(poses[i], depth, color, cam)
        osc.integrate(T2, depth, color, cam)
        poses[i] = T2
        if step % 2 == 1:
            gs.garbage_collect(); osc.garbage_collect()
        if not overlap or step in (2, 5):           # with overlap: several operators in flight between checks
            assert_same_state(gs, osc, "after fused re-integration %d:" % step)
    assert gs.num_integrated_frames() == 5


def test_collision_chains_and_drops_bit_exact(gpu, oracle):
    depth, color, T, cam, p = _setup(160, 120, 0.01, 400, 6000)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    d, c = _to_dev(depth, color)
    gs.integrate(T, d, c, cam)
    osc.integrate(T, depth, color, cam)
    assert (osc.hash()["offset"] != 0).sum() > 10 and osc.num_dropped() > 0
    assert_same_state(gs, osc, "chains:")
    gs.deintegrate(T, d, c, cam)
    osc.deintegrate(T, depth, color, cam)
    gs.garbage_collect()
    osc.garbage_collect()
    assert_same_state(gs, osc, "chains after GC:")


def test_heap_exhaustion_bit_exact(gpu, oracle):
    depth, color, T, cam, p = _setup(160, 120, 0.01, 20000, 300)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    d, c = _to_dev(depth, color)
    gs.integrate(T, d, c, cam)
    osc.integrate(T, depth, color, cam)
    assert_same_state(gs, osc, "heap exhausted:")
    assert gs.heap_free_count() == 0


def test_empty_invalid_and_no_colour(gpu, oracle):
    depth, color, T, cam, p = _setup(160, 120, 0.01, 20000, 6000)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    for dimg in (np.full_like(depth, -np.inf), np.zeros_like(depth)):
        d, c = _to_dev(dimg, color)
        gs.integrate(T, d, c, cam)
        osc.integrate(T, dimg, color, cam)
        assert gs.num_allocated_blocks() == 0
    gs.garbage_collect()
    osc.garbage_collect()
    d, _ = _to_dev(depth, None)
    gs.integrate(T, d, None, cam)
    osc.integrate(T, depth, None, cam)
    assert_same_state(gs, osc, "no colour:")
    assert gs.num_allocated_blocks() > 0
    gs.reset()
    osc.reset()
    assert_same_state(gs, osc, "after reset:")


def test_bad_arguments_fail_loudly(gpu):
    import ctypes as C
    from bundlefusion_amd.capi import lib, BFError, HashParams
    p = default_hash_params(num_buckets=1000, num_sdf_blocks=100)
    p.m_hashBucketSize = 8
    with pytest.raises(BFError):
        gpu.capi.SceneRepHashSDF(p)
    assert b"m_hashBucketSize" in lib.bf_last_error()
    assert lib.bf_scene_integrate(None, None, None, None, None) != 0


def test_config1_full_size_properties(gpu, oracle):
    """BASELINE config 1: one 640x480 frame of S1 at 4 mm voxels.  Full-size check through
    size-independent properties plus bit-exact comparison against the oracle (it takes ~2 s)."""
    depth, color, T, cam, p = _setup(640, 480, 0.004, 100000, 60000)
    gs = gpu.capi.SceneRepHashSDF(p)
    d, c = _to_dev(depth, color)
    gs.integrate(T, d, c, cam)
    dbg = gs.debug_hash()
    assert dbg["duplicate_keys"] == 0 and dbg["leaked"] == 0 and dbg["dropped"] == 0
    n_alloc = gs.num_allocated_blocks()
    assert 15000 < n_alloc < 40000          # ~22 k blocks for the wall at 4 mm
    assert dbg["occupied"] == n_alloc and dbg["heap_free"] == p.m_numSDFBlocks - n_alloc
    osc = oracle.OracleScene(p)
    osc.integrate(T, depth, color, cam)
    assert_same_state(gs, osc, "config 1:")
    # integrate -> de-integrate -> GC round trip restores the empty volume exactly
    gs.deintegrate(T, d, c, cam)
    gs.garbage_collect()
    gh, gheap, gcnt, gvox = gs.download()
    assert gcnt + 1 == p.m_numSDFBlocks and np.all(gh["ptr"] == FREE_ENTRY)
    assert not gvox.view(np.uint8).any()
    assert sorted(gheap.tolist()) == list(range(p.m_numSDFBlocks))


def test_hash_bucket_shards_partition_the_volume(gpu, oracle):
    """G volumes, each owning a contiguous range of home buckets, fed the same operation sequence (integrate,
    fused re-integrate, de-integrate, GC).  Their (key -> voxels) maps are disjoint and their union is the unsharded volume,
    bit for bit; every shard's table passes the structural invariants."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 10, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    G = 4
    scenes = [gpu.capi.SceneRepHashSDF(p) for _ in range(G + 1)]         # last one: unsharded
    for r in range(G):
        scenes[r].set_shard(r, G)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    for s in scenes:
        poses = [f[2].copy() for f in frames]
        for i in range(5):
            s.integrate(poses[i], dev[i][0], dev[i][1], cam)
        T2 = poses[2].copy(); T2[:3, 3] += np.float32(0.04)
        s.reintegrate(poses[2], T2, dev[2][0], dev[2][1], cam)
        s.deintegrate(poses[0], dev[0][0], dev[0][1], cam)
        s.garbage_collect()

    def blocks(s):
        gh, gheap, gcnt, gvox = s.download()
        occ = gh[gh["ptr"] != FREE_ENTRY]
        return {tuple(int(v) for v in e["pos"]): gvox[int(e["ptr"]):int(e["ptr"]) + VOX_PER_BLOCK].tobytes() for e in occ}

    whole = blocks(scenes[G])
    parts = [blocks(s) for s in scenes[:G]]
    assert sum(len(b) for b in parts) == len(whole) > 200
    union = {}
    for r, b in enumerate(parts):
        assert len(b) > 0.1 * len(whole) / G                                # every shard got work
        for key in b:
            assert oracle.hash_pos(p.m_hashNumBuckets, *key) * G // p.m_hashNumBuckets == r      # ownership = home bucket range
        union.update(b)
    assert union.keys() == whole.keys()
    assert all(union[k] == whole[k] for k in whole)
    for s in scenes:
        dbg = s.debug_hash()
        assert dbg["duplicate_keys"] == 0 and dbg["leaked"] == 0 and dbg["free_and_allocated"] == 0 and dbg["dropped"] == 0


def _volume_bytes(gs):
    gh, gheap, gcnt, gvox = gs.download()
    return gh.tobytes(), gheap.tobytes(), gcnt, gvox.tobytes()


def test_column_update_kernel_equals_voxel_kernel_and_exact_division_path(gpu, oracle, monkeypatch):
    """The three implementations of the voxel update — one voxel per lane (k_update / k_reupdate), one wave per block with shared
    refined reciprocals (k_update_col, the default), and k_update_col forced onto its literal-division path — produce the same
    volume bit for bit at the bench configuration (640x480, 4 mm): integrations, fused re-integrations, a plain de-integration,
    garbage collection; and all equal the oracle."""
    import torch
    W, H = 640, 480
    frames = [synth.scene_room(k * 9, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=400000, num_sdf_blocks=120000, voxel_size=0.004)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    rng = np.random.default_rng(5)

    def script(sc, integ, deint, reint):
        poses = [f[2].copy() for f in frames]
        for i in range(len(frames)):
            integ(i, poses[i])
        for i in (1, 3, 4):
            T2 = poses[i].copy(); T2[:3, 3] += np.float32(0.004) * (i + 1); T2[0, 1] += np.float32(1e-4)
            reint(i, poses[i], T2); poses[i] = T2
        deint(0, poses[0])
        sc.garbage_collect()

    results = {}
    for name, env in (("column", {}), ("column-exact", {"BF_TSDF_EXACT_DIV": "1"})):
        for k in ("BF_TSDF_EXACT_DIV",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gs = gpu.capi.SceneRepHashSDF(p)
        script(gs, lambda i, T: gs.integrate(T, dev[i][0], dev[i][1], cam), lambda i, T: gs.deintegrate(T, dev[i][0], dev[i][1], cam),
               lambda i, T0, T1: gs.reintegrate(T0, T1, dev[i][0], dev[i][1], cam))
        results[name] = _volume_bytes(gs)
        assert gs.num_allocated_blocks() > 20000
        del gs
    assert results["column"] == results["column-exact"], "shared-reciprocal quotients differ from the literal division"
    osc = oracle.OracleScene(p)

    def o_re(i, T0, T1):
        osc.deintegrate(T0, frames[i][0], frames[i][1], cam, threads=64); osc.integrate(T1, frames[i][0], frames[i][1], cam, threads=64)
    script(osc, lambda i, T: osc.integrate(T, frames[i][0], frames[i][1], cam, threads=64), lambda i, T: osc.deintegrate(T, frames[i][0], frames[i][1], cam, threads=64), o_re)
    assert results["column"][0] == osc.hash().tobytes() and results["column"][3] == osc.voxels().tobytes()


def test_column_update_blocks_outside_the_fast_range(gpu, oracle, monkeypatch):
    """Blocks whose camera-space z comes closer than 1 cm (a surface 3 cm in front of a sensor whose near plane is 0) fail the
    wave-uniform range test of k_update_col and take the literal path, next to blocks that pass it: same bits as the oracle."""
    monkeypatch.delenv("BF_TSDF_EXACT_DIV", raising=False)
    W, H = 160, 120
    depth, color, T, K = synth.scene_wall(W, H)
    near = np.where(np.isfinite(depth), np.float32(0.012) + (depth - np.float32(2.0)) * np.float32(0.05), depth).astype(np.float32)
    near[:, W // 2:] = np.where(np.isfinite(depth[:, W // 2:]), depth[:, W // 2:] * np.float32(0.5), depth[:, W // 2:])      # right half: 1 m away (fast range)
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"], dmin=0.0, dmax=4.0)
    p = default_hash_params(num_buckets=100000, num_sdf_blocks=40000, voxel_size=0.004, truncation=0.01, trunc_scale=0.01)
    gs = gpu.capi.SceneRepHashSDF(p)
    osc = oracle.OracleScene(p)
    d, c = _to_dev(near, color)
    T2 = T.copy(); T2[2, 3] += np.float32(0.002)
    gs.integrate(T, d, c, cam); osc.integrate(T, near, color, cam)
    gs.reintegrate(T, T2, d, c, cam); osc.deintegrate(T, near, color, cam); osc.integrate(T2, near, color, cam)
    assert_same_state(gs, osc, "near-plane blocks:")
    vox = gs.download()[3]
    assert (vox["weight"] > 0).sum() > 10000


@pytest.mark.parametrize("overlap", [False, True])
def test_sharded_allocation_collect_exchange_ingest(gpu, oracle, overlap):
    """Multi-GPU allocation (bf_scene_alloc_collect / _ingest / _place), emulated with G volumes in one process: for every
    operator each "rank" marches only its band of the pixel tiles and collects the block keys it meets, the lists are "exchanged" (here:
    handed to every volume), every volume ingests all lists (keeping what it owns) and runs the operator without an allocation of its
    own.  The union of the G shards equals the unsharded volume built by the operators' own allocation, bit for bit - integrations, fused
    re-integrations, a de-integration, garbage collection."""
    import torch
    W, H = 160, 120
    frames = [synth.scene_room(k * 10, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    G, CAP = 3, 1 << 15
    whole = gpu.capi.SceneRepHashSDF(p)
    shards = [gpu.capi.SceneRepHashSDF(p) for _ in range(G)]
    for r, s in enumerate(shards):
        s.set_shard(r, G); s.set_external_alloc(True)
    if overlap:
        for s in shards + [whole]:
            s.set_overlap(True)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    keys = [torch.zeros(CAP, dtype=torch.int64, device="cuda") for _ in range(G)]
    slots = [torch.zeros(CAP, dtype=torch.int32, device="cuda") for _ in range(G)]
    cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(G)]
    collected = []

    def allocate(T, i):
        for r, s in enumerate(shards):
            s.alloc_collect(T, dev[i][0], cam, r, G, keys[r], slots[r], cnt[r])
        for s in shards:
            s.alloc_sync()                          # "all-gather": every list complete before anybody reads it
        collected.append([int(c.item()) for c in cnt])
        for s in shards:
            for r in range(G):
                s.alloc_ingest(keys[r], cnt[r])
            s.alloc_place()
        for s in shards:
            s.alloc_sync()                          # the lists are overwritten by the next operator's collect

    poses = [f[2].copy() for f in frames]
    for i in range(5):
        allocate(poses[i], i)
        for s in shards + [whole]:
            s.integrate(poses[i], dev[i][0], dev[i][1], cam)
    for i, dt in ((2, 0.04), (4, 0.3)):
        T2 = poses[i].copy(); T2[:3, 3] += np.float32(dt)
        allocate(T2, i)
        for s in shards + [whole]:
            s.reintegrate(poses[i], T2, dev[i][0], dev[i][1], cam)
        poses[i] = T2
    for s in shards + [whole]:
        s.deintegrate(poses[0], dev[0][0], dev[0][1], cam)
        s.garbage_collect()
    # an allocation straight behind a garbage collection: collect / ingest / place must order themselves behind the
    # collection's exclusive section on the main stream (it frees blocks and rewrites bucket chains), not only runOperator
    allocate(poses[0], 0)
    for s in shards + [whole]:
        s.integrate(poses[0], dev[0][0], dev[0][1], cam)
        s.garbage_collect()
    T2 = poses[1].copy(); T2[:3, 3] -= np.float32(0.1)
    allocate(T2, 1)
    for s in shards + [whole]:
        s.reintegrate(poses[1], T2, dev[1][0], dev[1][1], cam)

    def blocks(s):
        gh, gheap, gcnt, gvox = s.download()
        occ = gh[gh["ptr"] != FREE_ENTRY]
        return {tuple(int(v) for v in e["pos"]): gvox[int(e["ptr"]):int(e["ptr"]) + VOX_PER_BLOCK].tobytes() for e in occ}
    w = blocks(whole)
    parts = [blocks(s) for s in shards]
    assert sum(len(b) for b in parts) == len(w) > 200
    union = {}
    for b in parts:
        union.update(b)
    assert union.keys() == w.keys() and all(union[k] == w[k] for k in w)
    # every band found blocks, and no band's list came near the full set: the march really was divided
    assert all(min(c) > 20 for c in collected) and max(max(c) for c in collected) < 0.8 * max(sum(c) for c in collected)
    for s in shards + [whole]:
        dbg = s.debug_hash()
        assert dbg["duplicate_keys"] == 0 and dbg["leaked"] == 0 and dbg["free_and_allocated"] == 0 and dbg["dropped"] == 0


def test_divided_allocation_through_the_communicator(gpu):
    """bf_scene_set_alloc_comm: the operators' own allocation with the ray march divided over the ranks of a bf_comm.  Two volumes in TWO THREADS of this
    process (hash-bucket shards 0 and 1 of 2) run the same operator sequence; their communicator is the callback transport, whose all-gather meets at a
    thread barrier - every integrate / re-integrate blocks in the collective until the other rank has issued the same operator, like RCCL.  The union of
    the shards equals the unsharded volume built by the local march, bit for bit.  Then the RCCL transport itself with a world of one (the only world a
    single-GPU box has): unique id, ncclCommInitRank, ncclAllGather on the scene's allocation stream through dlopen'ed librccl - same volume."""
    import ctypes as C
    import threading
    import torch
    from bundlefusion_amd import capi
    W, H = 160, 120
    frames = [synth.scene_room(k * 10, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    dev = [_to_dev(f[0], f[1]) for f in frames]

    def script(sc):
        poses = [f[2].copy() for f in frames]
        for i in range(5):
            sc.integrate(poses[i], dev[i][0], dev[i][1], cam)
        for i, dt in ((2, 0.04), (4, 0.3)):
            T2 = poses[i].copy(); T2[:3, 3] += np.float32(dt)
            sc.reintegrate(poses[i], T2, dev[i][0], dev[i][1], cam); poses[i] = T2
        sc.deintegrate(poses[0], dev[0][0], dev[0][1], cam)
        sc.garbage_collect()
        sc.integrate(poses[0], dev[0][0], dev[0][1], cam)

    def blocks(s):
        gh, gheap, gcnt, gvox = s.download()
        occ = gh[gh["ptr"] != FREE_ENTRY]
        return {tuple(int(v) for v in e["pos"]): gvox[int(e["ptr"]):int(e["ptr"]) + VOX_PER_BLOCK].tobytes() for e in occ}
    whole = gpu.capi.SceneRepHashSDF(p); whole.set_overlap(True)
    script(whole)
    w = blocks(whole)
    assert len(w) > 200
    G = 2
    barrier = threading.Barrier(G)
    slots = [None] * G
    calls = [0] * G

    def make_comm(rank):
        def gather(user, d_send, d_recv, nbytes, stream):
            n = int(nbytes)
            capi.check(capi.lib.bf_stream_synchronize(C.c_void_p(stream)))
            buf = np.zeros(n, np.uint8)
            capi.check(capi.lib.bf_memcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(d_send), C.c_size_t(n)))
            slots[rank] = buf
            barrier.wait(timeout=60)
            out = np.concatenate(slots)
            barrier.wait(timeout=60)             # nobody overwrites its slot before everybody has read all of them
            capi.check(capi.lib.bf_memcpy(C.c_void_p(d_recv), out.ctypes.data_as(C.c_void_p), C.c_size_t(G * n)))
            calls[rank] += 1
            return 0
        c = capi.Comm()
        c._cb = capi._ALL_GATHER_FN(gather)
        capi.check(capi.lib.bf_comm_create_callback(c._cb, None, G, rank, C.byref(c._h)))
        return c
    shards, errors = [None] * G, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            s = gpu.capi.SceneRepHashSDF(p); s.set_overlap(True); s.set_shard(rank, G)
            s.set_alloc_comm(make_comm(rank), 1 << 13)
            script(s)
            shards[rank] = blocks(s)
            dbg = s.debug_hash()
            assert dbg["duplicate_keys"] == 0 and dbg["leaked"] == 0 and dbg["free_and_allocated"] == 0 and dbg["dropped"] == 0
            s.set_alloc_comm(None)
        except BaseException as e:
            errors.append(e)
            barrier.abort()
    ths = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    assert calls == [8, 8]                       # one collective per allocating operator: 6 integrations + 2 re-integrations (a de-integration does not allocate)
    assert not (shards[0].keys() & shards[1].keys()) and min(len(b) for b in shards) > 50
    union = dict(shards[0]); union.update(shards[1])
    assert union.keys() == w.keys() and all(union[k] == w[k] for k in w)
    # RCCL itself, world of one
    comm = capi.Comm.rccl(1, 0, lambda b: b)
    assert comm.world() == (1, 0)
    a = torch.arange(4096, dtype=torch.uint8, device="cuda"); b = torch.zeros_like(a)
    comm.all_gather(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    s = gpu.capi.SceneRepHashSDF(p); s.set_overlap(True)
    s.set_alloc_comm(comm, 1 << 15)
    script(s)
    assert blocks(s) == w
    s.set_alloc_comm(None)
