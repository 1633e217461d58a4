"""GPU parity tests (-m gpu) of the batched TSDF operators (bf_scene_run_batch, csrc/tsdf_batch.h): a batch of integrate / de-integrate / re-integrate
operators must leave the hash table, the heap, the allocated-block list and every voxel byte exactly as the same operators issued one by one
(DepthSensing.cpp:854-902 issues them one by one; CUDASceneRepHashSDF.h:65-155) - against the CPU oracle under the exact arithmetic contract, and against the
per-operator kernels of the product itself under the fast contract (the library default) and at full size.
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params

from tests.test_tsdf_gpu import assert_same_state, _to_dev

pytestmark = pytest.mark.gpu


def _perturbed(T, rng, dt, drot=0.0):
    T2 = T.copy()
    T2[:3, 3] += (rng.normal(size=3) * dt).astype(np.float32)
    if drot:
        c_, s_ = np.float32(np.cos(drot)), np.float32(np.sin(drot))
        T2[:3, :3] = np.array([[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]], np.float32) @ T2[:3, :3]
    return T2


def _script(frames, rng, rounds=3):
    """A frame loop's worth of operators: [(kind, frame, T0, T1)] per batch, poses tracked so that every de-integration removes what was integrated."""
    poses = {}
    batches = []
    first = [("in", i, frames[i][2].copy(), None) for i in range(3)]
    for _, i, T, _ in first:
        poses[i] = T
    batches.append(first)
    nxt = 3
    for r in range(rounds):
        ops = []
        if nxt < len(frames):                                       # the previous frame's integration opens the batch (the frame loop defers it into the next frame's batch)
            ops.append(("in", nxt, frames[nxt][2].copy(), None)); poses[nxt] = ops[-1][2]; nxt += 1
        live = sorted(poses)
        for i in rng.permutation(live)[:4]:
            i = int(i)
            T2 = _perturbed(poses[i], rng, [0.002, 0.03, 0.3][r % 3], [0.0, 0.01, 0.5][r % 3])
            ops.append(("re", i, poses[i], T2)); poses[i] = T2
        if r == 1 and len(live) > 2:                                # a frame that became invalid: de-integrated, later integrated again at a new pose
            i = int(live[0])
            if not any(o[1] == i for o in ops):
                ops.append(("de", i, poses[i], None)); del poses[i]
        if r == 2 and 0 not in poses:
            ops.append(("in", 0, _perturbed(frames[0][2], rng, 0.02), None)); poses[0] = ops[-1][2]
        if nxt < len(frames):
            ops.append(("in", nxt, frames[nxt][2].copy(), None)); poses[nxt] = ops[-1][2]; nxt += 1
        batches.append(ops)
    return batches


def _apply_serial_gpu(gs, ops, dev, cam):
    for kind, i, T0, T1 in ops:
        if kind == "in":
            gs.integrate(T0, dev[i][0], dev[i][1], cam)
        elif kind == "de":
            gs.deintegrate(T0, dev[i][0], dev[i][1], cam)
        else:
            gs.reintegrate(T0, T1, dev[i][0], dev[i][1], cam)


def _apply_oracle(osc, ops, frames, cam):
    for kind, i, T0, T1 in ops:
        depth, color = frames[i][0], frames[i][1]
        if kind == "in":
            osc.integrate(T0, depth, color, cam)
        elif kind == "de":
            osc.deintegrate(T0, depth, color, cam)
        else:
            osc.deintegrate(T0, depth, color, cam); osc.integrate(T1, depth, color, cam)


def _apply_batch(gs, ops, dev, cam):
    gs.run_batch([(kind, T0, T1, dev[i][0], dev[i][1]) for kind, i, T0, T1 in ops], cam)


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("buckets,blocks", [(50000, 40000), (1300, 40000), (600, 40000), (50000, 700)])
def test_batch_equals_the_operators_one_by_one_vs_oracle(gpu, oracle, overlap, buckets, blocks):
    """Exact contract: seven batches (with garbage collection between them) vs the oracle issuing the same operators serially.  buckets = 1300: full home buckets and
    collision chains (4 .. 36 chained entries, nothing dropped) - the batch takes its operator-by-operator replay; buckets = 600: collision windows run out as well
    (the oracle drops 7+ keys: every later operator of the batch that needs such a block tries again, like the serial operators); blocks = 700: the heap runs out
    inside a batch (712 drops counted by the serial operators)."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 12, W, H) for k in range(8)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=buckets, num_sdf_blocks=blocks, voxel_size=0.02)
    gs = gpu.capi.SceneRepHashSDF(p)
    gs.set_arith("exact")
    if overlap:
        gs.set_overlap(True)
    osc = oracle.OracleScene(p)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    for r, ops in enumerate(_script(frames, np.random.default_rng(5), rounds=6)):
        assert len(ops) <= 12
        _apply_batch(gs, ops, dev, cam)
        _apply_oracle(osc, ops, frames, cam)
        if r % 2 == 1 or not overlap:
            assert_same_state(gs, osc, "after batch %d:" % r)
        gs.garbage_collect(); osc.garbage_collect()
        if r % 2 == 0 or not overlap:
            assert_same_state(gs, osc, "after batch %d + GC:" % r)
    if buckets <= 1300:
        assert (osc.hash()["offset"] != 0).sum() > 20, "the small table was supposed to build collision chains"
    assert (osc.num_dropped() > 0) == ((buckets, blocks) in ((600, 40000), (50000, 700))), "the scenario is not the one the test describes"


def _volume(gs):
    gh, gheap, gcnt, gvox = gs.download()
    return gh["pos"].tobytes(), gh["ptr"].tobytes(), gh["offset"].tobytes(), gheap.tobytes(), gcnt, gvox.view(np.uint8).tobytes()


@pytest.mark.parametrize("arith", ["fast", "exact"])
def test_batch_equals_per_operator_kernels_of_the_product(gpu, arith):
    """Both contracts: the batch against the product's own per-operator path (which the other suites hold to the oracle), bit for bit - table, heap, voxels."""
    W, H = 320, 240
    frames = [synth.scene_room(k * 9, W, H) for k in range(9)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=200000, num_sdf_blocks=120000, voxel_size=0.01)
    ga, gb = gpu.capi.SceneRepHashSDF(p), gpu.capi.SceneRepHashSDF(p)
    for g in (ga, gb):
        g.set_arith(arith); g.set_overlap(True)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    for r, ops in enumerate(_script(frames, np.random.default_rng(11), rounds=4)):
        _apply_serial_gpu(ga, ops, dev, cam)
        _apply_batch(gb, ops, dev, cam)
        ga.garbage_collect(); gb.garbage_collect()
        assert _volume(ga) == _volume(gb), "batch %d (%s)" % (r, arith)
    assert ga.num_integrated_frames() == gb.num_integrated_frames()
    assert ga.debug_hash() == gb.debug_hash()


def test_batch_of_eleven_at_full_size(gpu):
    """640x480 @4 mm (BASELINE configs[1]'s volume): ten re-integrations + one integration as ONE batch vs one by one, fast contract, incl. a frame without colour."""
    W, H = 640, 480
    frames = synth.render_frames([7 * k for k in range(12)], W, H)
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=500000, num_sdf_blocks=300000, voxel_size=0.004)
    ga, gb = gpu.capi.SceneRepHashSDF(p), gpu.capi.SceneRepHashSDF(p)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    rng = np.random.default_rng(2)
    for g in (ga, gb):
        g.set_arith("fast"); g.set_overlap(True)
    first = [("in", i, frames[i][2], None) for i in range(11)]
    _apply_serial_gpu(ga, first, dev, cam)
    _apply_batch(gb, first, dev, cam)
    assert _volume(ga) == _volume(gb), "eleven integrations"
    ops = [("re", i, frames[i][2], _perturbed(frames[i][2], rng, 0.004, 0.002)) for i in range(10)] + [("in", 11, frames[11][2], None)]
    _apply_serial_gpu(ga, ops, dev, cam)
    _apply_batch(gb, ops, dev, cam)
    ga.garbage_collect(); gb.garbage_collect()
    assert _volume(ga) == _volume(gb), "ten re-integrations + one integration"
    # no colour: the operator allocates, but does not update (CUDASceneRepHashSDF.cu:441-448)
    d11 = dev[11][0]
    ga.deintegrate(frames[11][2], d11, None, cam); ga.integrate(frames[3][2], dev[3][0], None, cam)
    gb.run_batch([("de", frames[11][2], None, d11, None), ("in", frames[3][2], None, dev[3][0], None)], cam)
    assert _volume(ga) == _volume(gb), "operators without colour data"
    assert ga.debug_hash() == gb.debug_hash() and ga.debug_hash()["dropped"] == 0


def test_batch_bad_arguments_fail_loudly(gpu):
    import ctypes as C
    from bundlefusion_amd.capi import lib, SceneBatchOp
    p = default_hash_params(num_buckets=1000, num_sdf_blocks=100)
    gs = gpu.capi.SceneRepHashSDF(p)
    cam = camera_params(160, 120, 100.0, 100.0, 80.0, 60.0)
    arr = (SceneBatchOp * 13)()
    assert lib.bf_scene_run_batch(gs._h, arr, 13, C.byref(cam)) != 0          # more than BF_SCENE_BATCH_MAX
    assert lib.bf_scene_run_batch(gs._h, arr, 0, C.byref(cam)) != 0
    assert lib.bf_scene_run_batch(gs._h, arr, 1, C.byref(cam)) != 0           # null depth
    assert b"d_depthData" in lib.bf_last_error()
