"""GPU parity tests (-m gpu) of the voxel update under the arithmetic contract "fast" (bf_scene_set_arith / BF_TSDF_ARITH=fast:
approximate division + FMA contraction, the contract of the reference's own Release GPU build, FriedLiver.vcxproj:124
<FastMath>true</FastMath>) against the oracle / the exact contract.

Bar:
  * EXACT: hash table (keys, slots, chain offsets), heap, heap counter, frustum list - allocation and garbage collection do not
    depend on the contract;
  * EXACT: every voxel weight;
  * sdf within 1e-5 x m_truncation (the smallest truncation any sample has), colour within 1 LSB after one operator on identical
    state; after a SEQUENCE of operators all but COLOUR_SEQ_FRAC of the bytes within 1 LSB and none beyond COLOUR_SEQ_MAX (a
    de-integration multiplies a colour deviation by w / (w - 1), the next integration by 0.8: 1.6 per re-integration of a weight-2 voxel,
    under either contract);
  * except "boundary voxels": voxels whose projection - recomputed here in float64 for every operator of the sequence - lies within
    BAND pixel of a pixel boundary.  Under either contract such a voxel may sample the neighbouring pixel (the two contracts round the
    image coordinate differently in the last bits, exactly as the reference's fast-math build differs from a host build of itself).
    They are excluded from the tolerance check, their share is bounded, and their values stay within the truncation band.
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params, FREE_ENTRY, VOX_PER_BLOCK

pytestmark = pytest.mark.gpu

BAND = 5e-4            # pixels
COLOUR_SEQ = None      # sequences: histogram bound instead of a per-byte one (see _compare)
COLOUR_SEQ_FRAC = 1e-4 # share of colour bytes that may deviate by more than 1 LSB after a sequence of operators (measured: none deviates at all)
COLOUR_SEQ_MAX = 4     # LSB


def _to_dev(depth, color):
    import torch
    return torch.from_numpy(np.ascontiguousarray(depth)).cuda(), torch.from_numpy(np.ascontiguousarray(color)).cuda()


def _boundary_mask(pos, ptr, poses, cam, voxel, n_voxels):
    """[n_voxels] bool: voxel projects within BAND of a pixel boundary (or of the principal plane) under any of `poses`."""
    occ = ptr != FREE_ENTRY
    bpos, bptr = pos[occ].astype(np.int64), ptr[occ].astype(np.int64)
    l = np.arange(512)
    lx, ly, lz = l & 7, (l >> 3) & 7, l >> 6
    vx = (bpos[:, 0:1] * 8 + lx[None, :]).astype(np.float64) * voxel
    vy = (bpos[:, 1:2] * 8 + ly[None, :]).astype(np.float64) * voxel
    vz = (bpos[:, 2:3] * 8 + lz[None, :]).astype(np.float64) * voxel
    near = np.zeros(vx.shape, bool)
    for T in poses:
        M = np.linalg.inv(np.asarray(T, np.float64))
        cx = M[0, 0] * vx + M[0, 1] * vy + M[0, 2] * vz + M[0, 3]
        cy = M[1, 0] * vx + M[1, 1] * vy + M[1, 2] * vz + M[1, 3]
        cz = M[2, 0] * vx + M[2, 1] * vy + M[2, 2] * vz + M[2, 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            hx = cx * cam.fx / cz + cam.mx + 0.5
            hy = cy * cam.fy / cz + cam.my + 0.5
        for h in (hx, hy):
            f = h - np.floor(h)
            near |= ~np.isfinite(h) | (np.minimum(f, 1.0 - f) < BAND)
        near |= np.abs(cz) < 1e-3
    out = np.zeros(n_voxels, bool)
    idx = (bptr[:, None] + l[None, :]).reshape(-1)
    out[idx] = near.reshape(-1)
    return out


def _compare(fast, exact, poses, cam, p, what, colour_tol, min_checked=10000):
    """fast / exact: (hash, heap, heapCounter, voxels) of the two volumes"""
    fh, fheap, fcnt, fvox = fast
    eh, eheap, ecnt, evox = exact
    assert fcnt == ecnt, what + ": heap counter"
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(fh[f], eh[f]), what + ": hash." + f
    assert np.array_equal(fheap, eheap), what + ": heap"
    n = len(evox)
    assert len(fvox) == n
    m = _boundary_mask(eh["pos"], eh["ptr"], poses, cam, p.m_virtualVoxelSize, n)
    live = (evox["weight"][:n] > 0) | (fvox["weight"][:n] > 0)
    chk = ~m
    assert np.array_equal(fvox["weight"][:n][chk], evox["weight"][:n][chk]), what + ": weights outside the boundary band"
    dsdf = np.abs(fvox["sdf"][:n].astype(np.float64) - evox["sdf"][:n].astype(np.float64))
    tol = 1e-5 * p.m_truncation
    assert dsdf[chk].max() <= tol, what + ": sdf deviates by %.3g (tolerance %.3g)" % (dsdf[chk].max(), tol)
    fc, ec = fvox["color"].astype(np.int32), evox["color"].astype(np.int32)
    dcol = np.abs(fc - ec).reshape(n, -1)
    hist = np.bincount(dcol[chk & live].reshape(-1), minlength=4)
    if colour_tol is not None:
        assert dcol[chk].max() <= colour_tol, what + ": colour deviates by %d LSB" % dcol[chk].max()
    else:
        # a sequence of operators: the de-integration of a colour is ill-conditioned - it multiplies any deviation of the stored byte by
        # w / (w - 1) (2 for a voxel of weight 2), the following integration by 0.8, so a 1 LSB conversion difference can grow by 1.6 per
        # re-integration of a weight-2 voxel UNDER EITHER CONTRACT (the exact contract's own rounding errors are amplified the same way
        # relative to real arithmetic).  Bound: almost every byte within 1 LSB, none beyond COLOUR_SEQ_MAX.
        frac_gt1 = float(hist[2:].sum()) / max(int(hist.sum()), 1)
        assert frac_gt1 <= COLOUR_SEQ_FRAC and dcol[chk].max() <= COLOUR_SEQ_MAX, what + ": colour histogram %s (%.2e beyond 1 LSB)" % (hist[:12].tolist(), frac_gt1)
    # boundary voxels: few, and still inside the truncation band / a plausible weight
    share = float((m & live).sum()) / max(int(live.sum()), 1)
    assert int((chk & live).sum()) >= min_checked, what + ": only %d voxels compared" % int((chk & live).sum())
    assert share < 0.08, what + ": %.1f %% boundary voxels" % (100 * share)
    band = p.m_truncation + p.m_truncScale * p.m_maxIntegrationDistance
    assert dsdf[m].max(initial=0.0) <= 2 * band
    assert np.abs(fvox["weight"][:n][m] - evox["weight"][:n][m]).max(initial=0.0) <= len(poses)
    differing = int((dsdf[chk] > 0).sum())
    return dict(checked=int((chk & live).sum()), boundary_share=share, max_dsdf=float(dsdf[chk].max()), max_dcol=int(dcol[chk].max()), differing=differing,
                colour_hist=hist[:10].tolist(),
                flips=int(((dsdf > tol) & m).sum()))


def _ostate(osc):
    return osc.hash(), osc.heap(), osc.heap_counter(), osc.voxels()


def test_fast_contract_single_operators_vs_oracle(gpu, oracle):
    """One operator at a time under the fast contract, each starting from a state that is bit-identical with the oracle's: an integration
    into the empty volume; then, on a volume built under the exact contract, a fused re-integration, and a de-integration followed by GC."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 12, W, H) for k in range(3)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    gs = gpu.capi.SceneRepHashSDF(p); gs.set_arith("fast")
    assert gs.arith() == "fast"
    osc = oracle.OracleScene(p)
    gs.integrate(frames[0][2], dev[0][0], dev[0][1], cam); osc.integrate(frames[0][2], frames[0][0], frames[0][1], cam)
    r = _compare(gs.download(), _ostate(osc), [frames[0][2]], cam, p, "integrate", 1, min_checked=30000)
    assert r["max_dcol"] == 0          # the colour blend of an integration has no rounding ties: identical bytes
    del gs
    gf = gpu.capi.SceneRepHashSDF(p)   # exact contract while the common state is built
    assert gf.arith() == "exact"
    for i in (1, 2):
        osc.integrate(frames[i][2], frames[i][0], frames[i][1], cam)
    for i in range(3):
        gf.integrate(frames[i][2], dev[i][0], dev[i][1], cam)
    assert np.array_equal(gf.download()[3].view(np.uint8), osc.voxels().view(np.uint8))
    gf.set_arith("fast")
    T2 = frames[1][2].copy(); T2[:3, 3] += np.float32(0.03)
    gf.reintegrate(frames[1][2], T2, dev[1][0], dev[1][1], cam)
    osc.deintegrate(frames[1][2], frames[1][0], frames[1][1], cam); osc.integrate(T2, frames[1][0], frames[1][1], cam)
    r2 = _compare(gf.download(), _ostate(osc), [frames[1][2], T2], cam, p, "fused re-integration", 2, min_checked=30000)
    gf.deintegrate(frames[2][2], dev[2][0], dev[2][1], cam); osc.deintegrate(frames[2][2], frames[2][0], frames[2][1], cam)
    gf.garbage_collect(); osc.garbage_collect()
    r3 = _compare(gf.download(), _ostate(osc), [frames[1][2], T2, frames[2][2]], cam, p, "de-integration + GC", COLOUR_SEQ, min_checked=30000)
    print("fast contract, single operators:", r, r2, r3)


def test_fast_contract_sequence_vs_oracle(gpu, oracle):
    """The operator sequence of test_sequence_integrate_deintegrate_gc_bit_exact entirely under the fast contract: 6 integrations, 2
    re-integrations (one fused, one as two operators), GC, then everything de-integrated again: the volume must come back EMPTY
    (weights are exact, so every voxel is reset and every block collected)."""
    W, H = 160, 120
    frames = [synth.scene_room(k * 15, W, H) for k in range(6)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000, num_sdf_blocks=40000, voxel_size=0.02)
    gs = gpu.capi.SceneRepHashSDF(p); gs.set_arith("fast"); gs.set_overlap(True)
    osc = oracle.OracleScene(p)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    poses_used = []
    for i, (depth, color, T, _) in enumerate(frames):
        gs.integrate(T, dev[i][0], dev[i][1], cam); osc.integrate(T, depth, color, cam); poses_used.append(T)
    for i in (1, 4):
        depth, color, T, _ = frames[i]
        T2 = T.copy(); T2[:3, 3] += np.float32(0.03)
        if i == 1:
            gs.reintegrate(T, T2, dev[i][0], dev[i][1], cam)
        else:
            gs.deintegrate(T, dev[i][0], dev[i][1], cam); gs.integrate(T2, dev[i][0], dev[i][1], cam)
        osc.deintegrate(T, depth, color, cam); osc.integrate(T2, depth, color, cam)
        frames[i] = (depth, color, T2, None); poses_used.append(T2)
    gs.garbage_collect(); osc.garbage_collect()
    r = _compare(gs.download(), _ostate(osc), poses_used, cam, p, "sequence", COLOUR_SEQ, min_checked=30000)
    print("fast contract, sequence:", r)
    for i, (depth, color, T, _) in enumerate(frames):
        gs.deintegrate(T, dev[i][0], dev[i][1], cam); gs.garbage_collect()
    gh, gheap, gcnt, gvox = gs.download()
    # pixel-boundary voxels can leave a weight behind (integrated through one pixel, de-integrated through its neighbour): blocks that
    # hold such a voxel survive GC.  Everything else is gone.
    left = int((gh["ptr"] != FREE_ENTRY).sum())
    assert left <= 0.02 * r["checked"] / 512 + 8, "%d blocks left after de-integrating everything" % left
    assert (gvox["weight"] > 0).sum() <= left * 8


def test_fast_contract_at_bench_configuration_vs_exact_contract(gpu):
    """640x480, 4 mm (the bench configuration): integrations, fused re-integrations, a de-integration and GC under both contracts on
    the GPU (the exact contract equals the oracle bit for bit: test_column_update_kernel_equals_voxel_kernel_and_exact_division_path)."""
    W, H = 640, 480
    frames = [synth.scene_room(k * 9, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=400000, num_sdf_blocks=120000, voxel_size=0.004)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    out, used = {}, []
    for arith in ("exact", "fast"):
        gs = gpu.capi.SceneRepHashSDF(p); gs.set_arith(arith); gs.set_overlap(True)
        poses = [f[2].copy() for f in frames]
        used = list(poses)
        for i in range(len(frames)):
            gs.integrate(poses[i], dev[i][0], dev[i][1], cam)
        for i in (1, 3, 4):
            T2 = poses[i].copy(); T2[:3, 3] += np.float32(0.004) * (i + 1); T2[0, 1] += np.float32(1e-4)
            gs.reintegrate(poses[i], T2, dev[i][0], dev[i][1], cam); poses[i] = T2; used.append(T2)
        gs.deintegrate(poses[0], dev[0][0], dev[0][1], cam)
        gs.garbage_collect()
        out[arith] = gs.download()
        assert gs.num_allocated_blocks() > 20000
        del gs
    r = _compare(out["fast"], out["exact"], used, cam, p, "bench configuration", COLOUR_SEQ, min_checked=5000000)
    print("fast vs exact contract at 640x480 / 4 mm:", r)


def test_fast_contract_in_the_frame_loop_changes_voxel_values_only(gpu):
    """Two pipelines over the same 33 frames at 640x480 / 4 mm (three local chunks, re-integrations, GC), one per arithmetic contract of the
    voxel update: the volume does not feed back into the bundling, so trajectories and every counter are identical; the hash table, the heap
    and every voxel weight are identical; sdf / colour within the contract outside the pixel-boundary voxels (here bounded statistically: the
    operator log of a pipeline is not replayed in float64)."""
    import torch
    from bundlefusion_amd.capi import default_app_state, default_bundling_state, intrinsics_matrix, sensor_desc
    W, H, n = 640, 480, 33
    frames = synth.render_frames(range(n))
    Kd = frames[0][3]
    K = intrinsics_matrix(Kd["fx"], Kd["fy"], Kd["mx"], Kd["my"])
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    out = {}
    for arith in ("exact", "fast"):
        gas = default_app_state(); gbs = default_bundling_state()
        gas.s_integrationWidth, gas.s_integrationHeight = W, H
        gas.s_SDFVoxelSize, gas.s_hashNumBuckets, gas.s_hashNumSDFBlocks = 0.004, 1000000, 250000
        gbs.s_maxNumImages = 8
        p = gpu.capi.Pipeline(gas, gbs, sensor_desc(W, H, K))
        p.scene().set_arith(arith)
        for d, c in dev:
            assert p.process_frame(d, c)
        for _ in range(4):
            p.process_end_of_sequence()
        p.synchronize()
        assert p.scene().arith() == arith
        out[arith] = (p.integrated_trajectory().copy(), p.optimized_trajectory().copy(), p.counters(), p.scene().download())
        del p
    (te, oe, ce, ve), (tf, of, cf, vf) = out["exact"], out["fast"]
    assert np.array_equal(te.view(np.uint32), tf.view(np.uint32)) and np.array_equal(oe.view(np.uint32), of.view(np.uint32)) and ce == cf and ce["deintegrate"] > 20
    (he, heape, cnte, voxe), (hf, heapf, cntf, voxf) = ve, vf
    assert cnte == cntf and np.array_equal(heape, heapf)
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(he[f], hf[f]), f
    live = (voxe["weight"] > 0) | (voxf["weight"] > 0)
    dw = voxe["weight"] != voxf["weight"]
    ds = np.abs(voxe["sdf"].astype(np.float64) - voxf["sdf"].astype(np.float64))
    dc = np.abs(voxe["color"].astype(np.int32) - voxf["color"].astype(np.int32)).max(axis=1)
    tol = 1e-5 * 0.06
    nlive = int(live.sum())
    share = lambda m: float((m & live).sum()) / nlive
    print("frame loop, fast vs exact contract: %d live voxels; weights differ %.2e, sdf beyond 1e-5 x truncation %.2e, colour beyond 1 LSB %.2e of them"
          % (nlive, share(dw), share(ds > tol), share(dc > 1)))
    assert nlive > 5000000
    # pixel-boundary voxels only: a few 1e-4 of the voxels per operator, ~60 operators
    assert share(dw) < 5e-3 and share(ds > tol) < 2e-2 and share(dc > 1) < 2e-2


@pytest.mark.parametrize("size", ["160x120@20mm", "640x480@4mm"])
def test_fast_contract_deferred_voxel_loads_are_bit_identical(gpu, monkeypatch, size):
    """The two forms of k_update_apx - voxel slices loaded only behind a valid sample (DEFER, the default) and loaded speculatively together with
    the samples (BF_APX_DEFER=0) - must not differ in ONE bit: integrations, fused re-integrations with translated and rotated poses, a
    de-integration, GC."""
    variant = "defer"
    W, H = (160, 120) if size.startswith("160") else (640, 480)
    voxel = 0.02 if W == 160 else 0.004
    frames = [synth.scene_room(k * 9, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000 if W == 160 else 400000, num_sdf_blocks=40000 if W == 160 else 120000, voxel_size=voxel)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    out = {}
    for lds in ("0", variant):
        monkeypatch.setenv("BF_APX_DEFER", "1" if lds == "defer" else "0")              # read when the scene is created
        gs = gpu.capi.SceneRepHashSDF(p); gs.set_arith("fast"); gs.set_overlap(True)
        poses = [f[2].copy() for f in frames]
        for i in range(len(frames)):
            gs.integrate(poses[i], dev[i][0], dev[i][1], cam)
        for i in (1, 3, 4):
            T2 = poses[i].copy(); T2[:3, 3] += np.float32(voxel) * (i + 1)
            a = np.float32(0.02 * i); R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)     # roll about the optical axis
            T2[:3, :3] = T2[:3, :3] @ R
            gs.reintegrate(poses[i], T2, dev[i][0], dev[i][1], cam); poses[i] = T2
        gs.deintegrate(poses[0], dev[0][0], dev[0][1], cam)
        gs.garbage_collect()
        out[lds] = gs.download()
        assert gs.num_allocated_blocks() > (100 if W == 160 else 20000)
        del gs
    (h0, heap0, c0, v0), (h1, heap1, c1, v1) = out["0"], out[variant]
    assert c0 == c1 and np.array_equal(heap0, heap1)
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(h0[f], h1[f]), f
    assert np.array_equal(v0.view(np.uint8), v1.view(np.uint8)), "%d voxels differ" % int((v0.view(np.uint8).reshape(len(v0), -1) != v1.view(np.uint8).reshape(len(v1), -1)).any(axis=1).sum())
