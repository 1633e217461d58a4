"""CPU tests (-m "not gpu"): the TSDF oracle against independent restatements and invariants.

The reference ships no golden vectors; the oracle's pin against the reference's own code is tests/test_ref_pin_cpu.py.  What can be checked
on the CPU is (a) the integer maps against a pure-Python big-int restatement of
VoxelUtilHashSDF.h:226-299, (b) debugHash()'s heap/hash invariants (CUDASceneRepHashSDF.h:179-314),
(c) size-independent properties: integrate -> de-integrate restores an empty volume, GC returns
every block to the heap.
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params, FREE_ENTRY, VOX_PER_BLOCK


def py_hash(num_buckets, x, y, z):
    # int32 wrap-around products, XOR, then `% unsigned` => unsigned modulo (usual arithmetic conversions)
    m = 0xFFFFFFFF
    h = ((x * 73856093) & m) ^ ((y * 19349669) & m) ^ ((z * 83492791) & m)
    return h % num_buckets


def py_world_to_block(voxel_size, w):
    vs = np.float32(voxel_size)
    out_v, out_b = [], []
    for c in w:
        p = np.float32(c) / vs
        s = (1 if p > 0 else 0) - (1 if p < 0 else 0)
        v = int(np.float32(p + np.float32(s) * np.float32(0.5)))    # trunc toward zero
        out_v.append(v)
        vv = v - 7 if v < 0 else v
        out_b.append(int(vv / 8))                                   # C division truncates
    return out_v, out_b


def test_hash_pos_matches_python_restatement(oracle):
    rng = np.random.default_rng(1)
    for nb in (800000, 1000003, 4, 65536):
        for _ in range(200):
            x, y, z = (int(v) for v in rng.integers(-5000, 5000, 3))
            assert oracle.hash_pos(nb, x, y, z) == py_hash(nb, x, y, z)
    # known answers (hand-computed with Python ints)
    assert oracle.hash_pos(800000, 0, 0, 0) == 0
    assert oracle.hash_pos(800000, 1, 0, 0) == 73856093 % 800000
    assert oracle.hash_pos(800000, -1, 0, 0) == ((-73856093) & 0xFFFFFFFF) % 800000
    assert oracle.hash_pos(800000, 3, -7, 11) == py_hash(800000, 3, -7, 11)


def test_world_to_block_matches_python_restatement(oracle):
    rng = np.random.default_rng(2)
    for vs in (0.004, 0.01, 0.002):
        for _ in range(300):
            w = rng.uniform(-3, 3, 3)
            v, b = oracle.world_to_block(vs, w)
            pv, pb = py_world_to_block(vs, w)
            assert v == pv and b == pb
    assert oracle.world_to_block(0.01, [0.0, -0.0, 0.004]) == ([0, 0, 0], [0, 0, 0])
    assert oracle.world_to_block(0.01, [-0.006, -0.08, 0.08])[1] == [-1, -1, 1]


def test_mat4_inverse(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        A = np.eye(4, dtype=np.float32)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        A[:3, :3] = q.astype(np.float32)
        A[:3, 3] = rng.uniform(-2, 2, 3)
        inv = oracle.mat4_inverse(A)
        assert np.allclose(inv @ A, np.eye(4), atol=2e-6)
        assert np.allclose(inv, np.linalg.inv(A.astype(np.float64)), atol=2e-6)


def _small_setup(width=160, height=120, voxel=0.01, buckets=20000, blocks=6000):
    depth, color, T, K = synth.scene_wall(width, height)
    cam = camera_params(width, height, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=buckets, num_sdf_blocks=blocks, voxel_size=voxel)
    return depth, color, T, cam, p


def check_invariants(hash_np, heap_np, heap_counter, num_blocks):
    occ = hash_np[hash_np["ptr"] != FREE_ENTRY]
    free = heap_np[: (heap_counter + 1) & 0xFFFFFFFF]
    assert len(np.unique(free)) == len(free), "duplicate free pointers in heap"
    used = occ["ptr"] // VOX_PER_BLOCK
    assert len(np.unique(used)) == len(used), "two hash entries share a block"
    assert not np.intersect1d(free, used).size, "ptr is on free heap but also allocated"
    assert len(free) + len(used) == num_blocks, "memory leak: neither free nor allocated"
    keys = {tuple(p) for p in occ["pos"]}
    assert len(keys) == len(occ), "duplicate block positions"
    return keys


def test_oracle_integrate_invariants_and_roundtrip(oracle):
    depth, color, T, cam, p = _small_setup()
    sc = oracle.OracleScene(p)
    sc.integrate(T, depth, color, cam)
    keys = check_invariants(sc.hash(), sc.heap(), sc.heap_counter(), p.m_numSDFBlocks)
    assert len(keys) == sc.num_allocated() > 100
    assert sc.num_occupied() <= sc.num_allocated()
    vox = sc.voxels()
    touched = vox["weight"] > 0
    assert touched.sum() > 1000
    assert np.all(vox["weight"][touched] == 1.0)
    trunc_max = p.m_truncation + p.m_truncScale * 3.0
    assert np.all(np.abs(vox["sdf"][touched]) <= trunc_max)
    # integrating the same frame twice: weight 2, sdf unchanged (running mean of equal samples)
    sdf1 = vox["sdf"].copy()
    sc.integrate(T, depth, color, cam)
    vox = sc.voxels()
    assert np.all(vox["weight"][touched] == 2.0)
    assert np.allclose(vox["sdf"][touched], sdf1[touched], atol=1e-7)
    # de-integrate twice: empty again; GC returns every block
    sc.deintegrate(T, depth, color, cam)
    sc.deintegrate(T, depth, color, cam)
    vox = sc.voxels()
    assert not np.any(vox["weight"] != 0) and not np.any(vox["sdf"] != 0) and not np.any(vox["color"] != 0)
    sc.garbage_collect()
    assert sc.num_allocated() == 0
    assert sc.heap_counter() + 1 == p.m_numSDFBlocks
    assert np.all(sc.hash()["ptr"] == FREE_ENTRY)
    assert sorted(sc.heap().tolist()) == list(range(p.m_numSDFBlocks))


def test_oracle_collision_chains(oracle):
    """Tiny bucket count forces overflow chains (HANDLE_COLLISIONS path) and window-exhaustion drops."""
    depth, color, T, cam, p = _small_setup(buckets=400, blocks=6000)
    sc = oracle.OracleScene(p)
    sc.integrate(T, depth, color, cam)
    h = sc.hash()
    assert (h["offset"] != 0).sum() > 10, "expected collision chains"
    check_invariants(h, sc.heap(), sc.heap_counter(), p.m_numSDFBlocks)
    assert sc.num_dropped() > 0
    # every allocated key is reachable or is a chain tail beyond the 7-step walk; lookups never crash
    sc.deintegrate(T, depth, color, cam)
    sc.garbage_collect()
    assert sc.num_allocated() == 0
    assert np.all(sc.hash()["ptr"] == FREE_ENTRY)
    assert np.all(sc.hash()["offset"] == 0)
    assert sc.heap_counter() + 1 == p.m_numSDFBlocks


def test_oracle_heap_exhaustion(oracle):
    depth, color, T, cam, p = _small_setup(blocks=300)
    sc = oracle.OracleScene(p)
    sc.integrate(T, depth, color, cam)
    assert sc.num_allocated() == 300
    assert sc.num_dropped() > 0
    check_invariants(sc.hash(), sc.heap(), sc.heap_counter(), p.m_numSDFBlocks)


def test_oracle_empty_and_invalid_input(oracle):
    depth, color, T, cam, p = _small_setup()
    sc = oracle.OracleScene(p)
    sc.integrate(T, np.full_like(depth, -np.inf), color, cam)
    assert sc.num_allocated() == 0 and sc.num_occupied() == 0
    sc.integrate(T, np.zeros_like(depth), color, cam)
    assert sc.num_allocated() == 0
    sc.garbage_collect()
    sc.integrate(T, depth, None, cam)        # no colour: blocks allocated, no voxel touched (.cu:441-448)
    assert sc.num_allocated() > 0
    assert not np.any(sc.voxels()["weight"] != 0)
