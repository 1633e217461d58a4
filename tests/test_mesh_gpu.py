"""GPU parity tests (-m gpu): marching cubes over the voxel hash through the C ABI vs the CPU oracle.

The HIP extraction reads the volume only through bf_scene_get_hash_data() / bf_scene_get_hash_params() — the reference layout — so
these tests are also the consumer-side check of that layout.  Bar: with the same case tables the triangle list is bit-exact INCLUDING
its order (hash slot, voxel index, table order); tol = 0.  The oracle itself is pinned to the reference's own kernel and Tables.h in
tests/test_ref_pin_cpu.py."""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import default_hash_params, camera_params

pytestmark = pytest.mark.gpu


def _build(gpu, oracle, W, H, voxel, buckets, blocks, ks):
    import torch
    frames = [synth.scene_room(k, W, H) for k in ks]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=buckets, num_sdf_blocks=blocks, voxel_size=voxel)
    gs = gpu.capi.SceneRepHashSDF(p); osc = oracle.OracleScene(p)
    for d, c, T, _ in frames:
        gs.integrate(T, torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), cam)
        osc.integrate(T, d, c, cam, threads=64)
    return gs, osc, p


def test_marching_cubes_bit_exact_and_ordered(gpu, oracle, tmp_path):
    gs, osc, p = _build(gpu, oracle, 160, 120, 0.02, 20011, 20000, (0, 20, 40))
    e, t = gpu.capi.marching_cubes_tables()
    mc = gpu.capi.MarchingCubesHashSDF(600000, p.m_hashNumBuckets, 0.02)
    tris, found = mc.extract(gs)
    otris, on = oracle.mc_extract(osc, 0.2, 0.2, e, t, 600000)
    assert found == on == len(tris) > 5000
    assert np.array_equal(tris.view(np.uint32), otris.view(np.uint32))            # same triangles in the same order
    # determinism: a second extraction gives the same bytes
    tris2, _ = mc.extract(gs)
    assert np.array_equal(tris.view(np.uint32), tris2.view(np.uint32))
    # a box
    pts = tris[:, :, :3].reshape(-1, 3)
    box = ([float(v) for v in np.percentile(pts, 30, axis=0)], [float(v) for v in np.percentile(pts, 70, axis=0)])
    tb, fb = mc.extract(gs, box=box)
    ob, onb = oracle.mc_extract(osc, 0.2, 0.2, e, t, 600000, box=box)
    assert 0 < fb == onb < found and np.array_equal(tb.view(np.uint32), ob.view(np.uint32))
    # capacity: the buffer bounds the mesh, the count of what the volume holds is still reported
    small = gpu.capi.MarchingCubesHashSDF(1000, p.m_hashNumBuckets, 0.02)
    ts, fs = small.extract(gs)
    assert fs == found and len(ts) == 1000 and np.array_equal(ts.view(np.uint32), tris[:1000].view(np.uint32))
    # the mesh: a closed-enough surface patch — merged vertices are shared by ~6 triangles, the file parses
    mc.extract(gs)
    nv, nf = mc.save_mesh(tmp_path / "scan.ply")
    assert 0.9 * found < nf <= found and 0.3 * nf < nv < nf          # merged vertices are shared by ~6 triangles of a closed-enough patch
    raw = open(tmp_path / "scan.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % nv in head and b"element face %d" % nf in head and len(body) == nv * 16 + nf * 13


def test_marching_cubes_at_4mm_and_after_reintegration(gpu, oracle):
    """The bench resolution (640x480, 4 mm) with overflow chains in the hash, after a fused re-integration and a garbage collection."""
    import torch
    gs, osc, p = _build(gpu, oracle, 640, 480, 0.004, 100003, 120000, (0, 12))
    d, c, T, K = synth.scene_room(12, 640, 480)
    cam = camera_params(640, 480, K["fx"], K["fy"], K["mx"], K["my"])
    T2 = T.copy(); T2[:3, 3] += np.float32(0.003)
    gs.reintegrate(T, T2, torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda(), cam)
    osc.deintegrate(T, d, c, cam, threads=64); osc.integrate(T2, d, c, cam, threads=64)
    gs.garbage_collect(); osc.garbage_collect()
    e, t = gpu.capi.marching_cubes_tables()
    mc = gpu.capi.MarchingCubesHashSDF(3000000, p.m_hashNumBuckets, 0.004)
    tris, found = mc.extract(gs)
    otris, on = oracle.mc_extract(osc, 0.04, 0.04, e, t, 3000000)
    assert found == on == len(tris) > 100000
    assert np.array_equal(tris.view(np.uint32), otris.view(np.uint32))


def test_marching_cubes_on_a_volume_built_under_the_default_contract(gpu):
    """The consumers of the volume normally see one built by the library's DEFAULT voxel update (fast contract, batched operators).  Here: the same three frames + a
    re-integration, once operator by operator under the exact contract (the volume the tests above hold to the oracle bit for bit), once as ONE batch under the fast
    contract; both extracted by the product.  Contract-level bar: sdf differs by <= 1e-5 x truncation, so the zero crossings move by a fraction of a micron - same
    triangle count within 0.2 %, surface area within 1e-4, centroid within 10 um."""
    import torch
    W, H, voxel = 640, 480, 0.004
    frames = [synth.scene_room(k, W, H) for k in (0, 12, 24)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=100003, num_sdf_blocks=120000, voxel_size=voxel)
    dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames]
    T2 = frames[1][2].copy(); T2[:3, 3] += np.float32(0.003)
    ge = gpu.capi.SceneRepHashSDF(p); ge.set_arith("exact")
    for (d, c), f in zip(dev, frames):
        ge.integrate(f[2], d, c, cam)
    ge.reintegrate(frames[1][2], T2, dev[1][0], dev[1][1], cam)
    gf = gpu.capi.SceneRepHashSDF(p); gf.set_arith("fast")
    gf.run_batch([("in", f[2], None, d, c) for (d, c), f in zip(dev, frames)] + [("re", frames[1][2], T2, dev[1][0], dev[1][1])], cam)
    for g in (ge, gf):
        g.garbage_collect()
    mc = gpu.capi.MarchingCubesHashSDF(3000000, p.m_hashNumBuckets, voxel)

    def stats(g):
        tris, found = mc.extract(g)
        v = tris[:, :, :3].astype(np.float64)
        area = 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
        return found, float(area.sum()), (v.mean(axis=1) * area[:, None]).sum(axis=0) / area.sum()
    ne, ae, ce = stats(ge)
    nf, af, cf = stats(gf)
    print("marching cubes, exact vs default contract: %d / %d triangles, area %.6f / %.6f m^2, centroid distance %.2e m" % (ne, nf, ae, af, float(np.linalg.norm(ce - cf))))
    assert ne > 100000 and abs(nf - ne) <= 0.002 * ne and abs(af - ae) <= 1e-4 * ae and np.linalg.norm(ce - cf) < 1e-5
