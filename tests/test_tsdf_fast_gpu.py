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

BAND = 2e-4            # pixels, images up to 640 wide (measured need: 1.1e-4, gpurun r04c; until round 3: 5e-4)
BAND_WIDE = 6e-4       # wider images (1280x960 at 2 mm voxels: coordinates and voxel indices both twice as large; measured need 4.8e-4)
COLOUR_SEQ = None      # sequences: histogram bound instead of a per-byte one (see _compare)
COLOUR_SEQ_FRAC = 1e-4 # share of colour bytes that may deviate by more than 1 LSB after a sequence of operators (measured: none deviates at all)
COLOUR_SEQ_MAX = 4     # LSB
SDF_TOL = 1e-5         # x m_truncation: one operator on identical state, and short sequences
SDF_TOL_LONG = 1e-4    # x m_truncation: operator logs of a frame loop (~100 fused re-integrations: a voxel is de- and re-integrated dozens of times; the quotient of a
                       # de-integration, (sdf w - s) / (w - 1), doubles a deviation at w = 2 and the next integration halves it - a random walk of 1.5 ulp steps under
                       # EITHER contract relative to real arithmetic; measured 3.5e-5 x truncation = 2 um after the 33-frame loop's log, gpurun r04b)


def _to_dev(depth, color):
    import torch
    return torch.from_numpy(np.ascontiguousarray(depth)).cuda(), torch.from_numpy(np.ascontiguousarray(color)).cuda()


def band_for(cam):
    """Width (pixels) of the band around a pixel boundary inside which the two contracts may pick different pixels.  Both evaluate the image
    coordinate in float32: the exact contract as pf.x * fx / pf.z + mx on camera-space coordinates, the fast contract as one FMA chain over
    the voxel's integer coordinates with fx * t folded in, whose numerator reaches a few thousand (half an ulp at 4096 is 2.4e-4) before it is
    divided by z >= 0.4 m - about 1e-4 pixel at 640x480, more on a wider image with finer voxels.  Every test prints the distance it actually
    NEEDED (`needed_band`: the largest distance to a pixel boundary among the voxels that differ beyond the tolerance); measured on MI355X
    (gpurun r04c): 1.05e-4 / 1.12e-4 / 9.8e-5 at 640x480 @4 mm (replayed frame-loop log, bench configuration, noisy stream), 4.8e-4 at 1280x960 @2 mm."""
    return BAND if cam.m_imageWidth <= 640 else BAND_WIDE


def _boundary_dist(pos, ptr, poses, cam, voxel, n_voxels, chunk=16384):
    """[n_voxels] float32: the smallest distance (pixels) of a voxel's projection - recomputed in float64, on the GPU - to a pixel boundary
    under any of `poses`; 0 where a projection is not finite or the voxel lies in the principal plane; 1 for voxels of unallocated blocks."""
    import torch
    occ = ptr != FREE_ENTRY
    bpos_all, bptr_all = pos[occ].astype(np.int64), ptr[occ].astype(np.int64)
    dev = "cuda"
    l = torch.arange(512, device=dev)
    lx, ly, lz = (l & 7).double(), ((l >> 3) & 7).double(), (l >> 6).double()
    Ms = [torch.from_numpy(np.linalg.inv(np.asarray(T, np.float64))).to(dev) for T in poses]
    out = torch.ones(n_voxels, dtype=torch.float32, device=dev)
    for c0 in range(0, len(bpos_all), chunk):
        bpos = torch.from_numpy(bpos_all[c0:c0 + chunk]).to(dev); bptr = torch.from_numpy(bptr_all[c0:c0 + chunk]).to(dev)
        vx = (bpos[:, 0:1].double() * 8 + lx[None, :]) * voxel
        vy = (bpos[:, 1:2].double() * 8 + ly[None, :]) * voxel
        vz = (bpos[:, 2:3].double() * 8 + lz[None, :]) * voxel
        dist = torch.ones(vx.shape, dtype=torch.float64, device=dev)
        for M in Ms:
            cx = M[0, 0] * vx + M[0, 1] * vy + M[0, 2] * vz + M[0, 3]
            cy = M[1, 0] * vx + M[1, 1] * vy + M[1, 2] * vz + M[1, 3]
            cz = M[2, 0] * vx + M[2, 1] * vy + M[2, 2] * vz + M[2, 3]
            for h in (cx * cam.fx / cz + cam.mx + 0.5, cy * cam.fy / cz + cam.my + 0.5):
                f = h - torch.floor(h)
                d = torch.minimum(f, 1.0 - f)
                dist = torch.minimum(dist, torch.where(torch.isfinite(h), d, torch.zeros_like(d)))
            dist = torch.where(cz.abs() < 1e-3, torch.zeros_like(dist), dist)
        idx = (bptr[:, None] + l[None, :]).reshape(-1)
        out[idx] = dist.reshape(-1).float()
    return out.cpu().numpy()


def _inverse44_f32(T):
    """bf::inverse44 (csrc/bf_device.h) operation by operation in float32: the world -> camera transform both the product and the oracle project voxels with"""
    m = np.asarray(T, np.float32).reshape(16)
    f = np.float32
    adj = np.zeros(16, np.float32)
    for r in range(4):
        for c in range(4):
            rows = [i for i in range(4) if i != c]; cols = [j for j in range(4) if j != r]
            s_ = f(-1.0) if ((r + c) & 1) else f(1.0)
            A = lambda i, j: m[rows[i] * 4 + cols[j]]
            adj[r * 4 + c] = f(f(f(f(f(f(s_ * A(0, 0)) * A(1, 1)) * A(2, 2)) - f(f(f(s_ * A(0, 0)) * A(1, 2)) * A(2, 1))) - f(f(f(s_ * A(1, 0)) * A(0, 1)) * A(2, 2))) +
                                 f(f(f(s_ * A(1, 0)) * A(0, 2)) * A(2, 1)))
            adj[r * 4 + c] = f(f(adj[r * 4 + c] + f(f(f(s_ * A(2, 0)) * A(0, 1)) * A(1, 2))) - f(f(f(s_ * A(2, 0)) * A(0, 2)) * A(1, 1)))
    det = f(f(f(f(m[0] * adj[0]) + f(m[1] * adj[4])) + f(m[2] * adj[8])) + f(m[3] * adj[12]))
    detr = f(f(1.0) / det)
    return (adj * detr).astype(np.float32).reshape(4, 4)


def _block_keys(bad, eh):
    """block coordinates [n_blocks, 3] of the blocks the voxels `bad` live in, and for every voxel the row of its block"""
    occ = eh["ptr"] != FREE_ENTRY
    ptrs, poss = eh["ptr"][occ].astype(np.int64), eh["pos"][occ].astype(np.int64)
    o = np.argsort(ptrs); ptrs, poss = ptrs[o], poss[o]
    blk = np.searchsorted(ptrs, (bad // VOX_PER_BLOCK) * VOX_PER_BLOCK)
    assert (ptrs[blk] == (bad // VOX_PER_BLOCK) * VOX_PER_BLOCK).all(), "a differing voxel outside every allocated block"
    ub, inv = np.unique(blk, return_inverse=True)
    return poss[ub].astype(np.int32), inv


def _existence_gpu(bad, eh, log, frames, cam, p):
    """exists[t][v]: the block of voxel bad[v] is in the table when the update of the t-th operator of `log` (garbage collections not counted) runs.  The log is
    replayed through a fresh volume of the product (allocation and garbage collection do not depend on the arithmetic contract) and the blocks are looked up after
    every operator.  A voxel of a block that does not exist yet is not updated even when its sample is valid - its block is allocated by the rays of OTHER pixels."""
    import torch
    import bundlefusion_amd.capi as capi
    keys, inv = _block_keys(bad, eh)
    gs = capi.SceneRepHashSDF(p); gs.set_arith("exact")
    dk = torch.from_numpy(keys).cuda()
    inv_t = torch.from_numpy(inv).cuda()
    dev = {}
    out = []
    for kind, i, T in log:
        if kind == "gc":
            gs.garbage_collect(); continue
        if i not in dev:
            d, c = frames[i]
            dev[i] = (torch.from_numpy(np.ascontiguousarray(d)).cuda(), torch.from_numpy(np.ascontiguousarray(c)).cuda())
        (gs.deintegrate if kind == "de" else gs.integrate)(T, dev[i][0], dev[i][1], cam)
        out.append((gs.find_blocks(dk) != FREE_ENTRY)[inv_t])
    return out


def _explain(bad, fvox, evox, eh, log, frames, cam, p, band, tol_sdf, tol_col, slots=64, exists=None):
    """Every voxel that differs between the two volumes beyond the contract must be EXPLAINED (masking every voxel near a pixel boundary
    would also hide a bug confined to such voxels).  For the voxels `bad` (indices into the voxel arrays) the whole operator log is replayed in float64, here, per
    voxel: wherever a voxel's projection lies within `band` pixel of a pixel boundary the update is evaluated for the pixel on EITHER side (up to four candidates per
    operator, every combination carried along - `slots` alternatives per voxel at most).  A voxel is explained when one of its alternatives reproduces the product's
    (sdf, weight, colour) within the contract's tolerances - i.e. product and oracle only disagree about which side of a pixel boundary a projection fell on at some
    operator; the oracle's own value must be among the alternatives as well (that checks this replay, not the product).
    log: [(kind "in" | "de" | "gc", frame, T)], frames[frame] = (depth HxW float32, colour HxWx4 uint8); exists: see _existence_gpu.
    Returns (unexplained, not_reproducing_the_oracle, most_alternatives)."""
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"          # (the CPU form is what tests/test_tsdf_oracle.py checks the replay itself with)
    nb = len(bad)
    if nb == 0:
        return 0, 0, 0
    occ = eh["ptr"] != FREE_ENTRY
    ptrs, poss = eh["ptr"][occ].astype(np.int64), eh["pos"][occ].astype(np.int64)
    o = np.argsort(ptrs); ptrs, poss = ptrs[o], poss[o]
    blk = np.searchsorted(ptrs, (bad // VOX_PER_BLOCK) * VOX_PER_BLOCK)
    assert (ptrs[blk] == (bad // VOX_PER_BLOCK) * VOX_PER_BLOCK).all(), "a differing voxel outside every allocated block"
    l = bad % VOX_PER_BLOCK
    vox = float(np.float32(p.m_virtualVoxelSize))
    world = np.stack([(poss[blk, 0] * 8 + (l & 7)), (poss[blk, 1] * 8 + ((l >> 3) & 7)), (poss[blk, 2] * 8 + (l >> 6))], axis=1).astype(np.float64) * vox
    X = torch.from_numpy(world).to(dev)
    # the centre of every voxel's block: an operator only updates blocks of its frustum list (blockInFrustum, VoxelUtilHashSDF.h:322-326 / DepthCameraUtil.h:97-142)
    BC = torch.from_numpy((poss[blk] * 8).astype(np.float64) * vox + vox * 3.5).to(dev)
    zmin, zmax = float(cam.m_sensorDepthWorldMin), float(cam.m_sensorDepthWorldMax)
    W_, H_ = cam.m_imageWidth, cam.m_imageHeight
    max_dist, trunc0, trunc_s, w_max = float(p.m_maxIntegrationDistance), float(p.m_truncation), float(p.m_truncScale), float(p.m_integrationWeightMax)
    S = torch.zeros((nb, slots), dtype=torch.float64, device=dev); Wt = torch.zeros_like(S); C = torch.zeros((nb, slots, 3), dtype=torch.float64, device=dev)
    nvalid = torch.ones(nb, dtype=torch.int64, device=dev)
    overflow = torch.zeros(nb, dtype=torch.bool, device=dev)
    ar = torch.arange(nb, device=dev)
    dev_frames = {}

    def apply(kind, s_old, w_old, c_old, ok, sdf, col):
        """voxelApply (CUDASceneRepHashSDF.cu:425-516) in float64 on arrays of any shape (colour: last axis 3); ok: the sample is valid"""
        if kind == "in":
            first = (w_old == 0).unsqueeze(-1)
            c_new = torch.where(first, col, torch.clamp(torch.round(0.2 * col + 0.8 * c_old), 0.0, 254.0))
            s_new = (sdf + s_old * w_old) / (1.0 + w_old)
            w_new = torch.clamp(w_old + 1.0, max=w_max)
        else:
            den = w_old - 1.0
            q = (c_old * w_old.unsqueeze(-1) - col) / den.unsqueeze(-1)
            c_new = torch.clamp(torch.floor(torch.abs(q) + 0.5) * torch.sign(q), 0.0, 254.0)       # roundf: half away from zero
            s_new = (s_old * w_old - sdf) / den
            w_new = torch.clamp(den, min=0.0)
            dead = (w_new <= 0.001)
            c_new = torch.where(dead.unsqueeze(-1), torch.zeros_like(c_new), torch.nan_to_num(c_new, nan=0.0, posinf=254.0, neginf=0.0))
            s_new = torch.where(dead, torch.zeros_like(s_new), s_new)
            w_new = torch.where(dead, torch.zeros_like(w_new), w_new)
        okc = ok.unsqueeze(-1)
        return torch.where(ok, s_new, s_old), torch.where(ok, w_new, w_old), torch.where(okc, c_new, c_old)

    most = 1
    t_op = -1
    for kind, fi, T in log:
        if kind == "gc":
            continue                # a block is only freed when all its voxels are zero: the states are zero already
        t_op += 1
        if fi not in dev_frames:
            d, c = frames[fi]
            dev_frames[fi] = (torch.from_numpy(np.ascontiguousarray(d, np.float32)).to(dev).double(), torch.from_numpy(np.ascontiguousarray(c)[..., :3].astype(np.float64)).to(dev))
        D, Cimg = dev_frames[fi]
        M = torch.from_numpy(_inverse44_f32(T).astype(np.float64)).to(dev)          # the float32 inverse the kernels use, not the exact one: the pixel a projection falls into is decided with it
        pc = X @ M[:3, :3].T + M[:3, 3]
        cz = pc[:, 2]
        hx, hy = pc[:, 0] * cam.fx / cz + cam.mx + 0.5, pc[:, 1] * cam.fy / cz + cam.my + 0.5
        fin = torch.isfinite(hx) & torch.isfinite(hy)
        bc = BC @ M[:3, :3].T + M[:3, 3]
        bx = (2.0 * (bc[:, 0] * cam.fx / bc[:, 2] + cam.mx) - (W_ - 1.0)) / (W_ - 1.0) * 0.95
        by = ((H_ - 1.0) - 2.0 * (bc[:, 1] * cam.fy / bc[:, 2] + cam.my)) / (H_ - 1.0) * 0.95
        bz = (bc[:, 2] - zmin) / (zmax - zmin) * 0.95
        fin = fin & ~((bx < -1.0) | (bx > 1.0) | (by < -1.0) | (by > 1.0) | (bz < 0.0) | (bz > 1.0))
        if exists is not None:
            fin = fin & exists[t_op].to(dev)          # no block, no update
        hx, hy = torch.where(fin, hx, torch.full_like(hx, -10.0)), torch.where(fin, hy, torch.full_like(hy, -10.0))
        px0, py0 = torch.trunc(hx).long(), torch.trunc(hy).long()
        fx_, fy_ = hx - torch.floor(hx), hy - torch.floor(hy)
        eb = 2.0 * band          # alternatives are enumerated in a wider band than the one the differing voxels must lie in: the projection here is a float64 evaluation, not either kernel's
        ax = torch.where(fx_ < eb, -1, torch.where(fx_ > 1.0 - eb, 1, 0)).long()
        ay = torch.where(fy_ < eb, -1, torch.where(fy_ > 1.0 - eb, 1, 0)).long()
        cands = [(torch.zeros_like(ax), torch.zeros_like(ay), torch.ones(nb, dtype=torch.bool, device=dev)), (ax, torch.zeros_like(ay), ax != 0),
                 (torch.zeros_like(ax), ay, ay != 0), (ax, ay, (ax != 0) & (ay != 0))]
        samples = []
        for dx, dy, en in cands:
            px, py = px0 + dx, py0 + dy
            inimg = (px >= 0) & (px < W_) & (py >= 0) & (py < H_) & fin & (hx > -1.0) & (hy > -1.0)      # (uint)(int)h < W: h in (-1, 0) converts to 0
            pxc, pyc = px.clamp(0, W_ - 1), py.clamp(0, H_ - 1)
            dep = D[pyc, pxc]
            sdf = dep - cz
            ok = inimg & torch.isfinite(dep) & (dep < max_dist) & (sdf.abs() < trunc0 + trunc_s * dep)
            sdf = torch.where(ok, sdf, torch.zeros_like(sdf))
            col = torch.where(ok.unsqueeze(-1), Cimg[pyc, pxc], torch.zeros_like(Cimg[pyc, pxc]))
            if samples:          # an alternative only counts when it differs from one already listed (same validity, depth and colour: the same update)
                for _, ok0, sdf0, col0 in samples:
                    en = en & ~((ok == ok0) & (sdf == sdf0) & (col == col0).all(dim=-1))
            samples.append((en, ok, sdf, col))
        ncand = sum(en.long() for en, _, _, _ in samples)
        single = ncand == 1
        # voxels with one candidate: every alternative is updated in place
        en, ok, sdf, col = samples[0]
        s1, w1, c1 = apply(kind, S, Wt, C, (ok & single).unsqueeze(-1).expand(-1, slots), sdf.unsqueeze(-1).expand(-1, slots), col.unsqueeze(1).expand(-1, slots, -1))
        multi = torch.nonzero(~single).squeeze(-1)
        if len(multi):
            nv = nvalid[multi]
            newS, newW, newC = torch.zeros((len(multi), slots), dtype=torch.float64, device=dev), torch.zeros((len(multi), slots), dtype=torch.float64, device=dev), torch.zeros((len(multi), slots, 3), dtype=torch.float64, device=dev)
            before = torch.zeros(len(multi), dtype=torch.int64, device=dev)
            mS, mW, mC = S[multi], Wt[multi], C[multi]
            for en, ok, sdf, col in samples:
                enm = en[multi]
                a_s, a_w, a_c = apply(kind, mS, mW, mC, ok[multi].unsqueeze(-1).expand(-1, slots), sdf[multi].unsqueeze(-1).expand(-1, slots), col[multi].unsqueeze(1).expand(-1, slots, -1))
                for sl in range(slots):
                    tgt = before * nv + sl
                    m = enm & (sl < nv) & (tgt < slots)
                    if bool(m.any()):
                        rows = torch.nonzero(m).squeeze(-1)
                        newS[rows, tgt[rows]] = a_s[rows, sl]; newW[rows, tgt[rows]] = a_w[rows, sl]; newC[rows, tgt[rows]] = a_c[rows, sl]
                before = before + enm.long()
            s1[multi], w1[multi], c1[multi] = newS, newW, newC
            overflow[multi] |= nv * ncand[multi] > slots
            nvalid[multi] = torch.clamp(nv * ncand[multi], max=slots)
            most = max(most, int(nvalid.max()))
        S, Wt, C = s1, w1, c1

    def matches(vox_arr):
        tw = torch.from_numpy(vox_arr["weight"][bad].astype(np.float64)).to(dev).unsqueeze(-1)
        ts = torch.from_numpy(vox_arr["sdf"][bad].astype(np.float64)).to(dev).unsqueeze(-1)
        tc = torch.from_numpy(vox_arr["color"][bad][:, :3].astype(np.float64)).to(dev).unsqueeze(1)
        live = torch.arange(slots, device=dev).unsqueeze(0) < nvalid.unsqueeze(-1)
        hit = live & (Wt == tw) & ((S - ts).abs() <= tol_sdf) & ((C - tc).abs().amax(dim=-1) <= tol_col)
        return hit.any(dim=1)
    prod_ok, orac_ok = matches(fvox), matches(evox)
    if slots < 1024 and bool((~prod_ok | ~orac_ok).any()):
        # the few voxels left: their projections touch a boundary under many operators of the log (the same frame re-integrated at almost the same pose) and the
        # alternatives did not fit - once more, for them alone, with room for every combination
        redo = torch.nonzero(~prod_ok | ~orac_ok).squeeze(-1).cpu().numpy()
        sub_ex = [e.to(dev)[torch.from_numpy(redo).to(dev)] for e in exists] if exists is not None else None
        un2, om2, most2 = _explain(bad[redo], fvox, evox, eh, log, frames, cam, p, band, tol_sdf, tol_col, slots=1024, exists=sub_ex)
        return un2, om2, max(most, most2)
    for v in torch.nonzero(~prod_ok).squeeze(-1)[:5].tolist():          # on record for a failing run: what the product holds, what the oracle holds, the alternatives
        k = int(nvalid[v])
        print("  unexplained voxel %d: product (sdf %.7g, w %g, rgb %s) oracle (sdf %.7g, w %g, rgb %s) alternatives %s" % (
            int(bad[v]), fvox["sdf"][bad[v]], fvox["weight"][bad[v]], fvox["color"][bad[v]][:3].tolist(), evox["sdf"][bad[v]], evox["weight"][bad[v]], evox["color"][bad[v]][:3].tolist(),
            [(round(float(S[v, q]), 7), float(Wt[v, q]), C[v, q].tolist()) for q in range(min(k, 6))]))
    return int((~prod_ok).sum()), int((~orac_ok).sum()), most


def _compare(fast, exact, poses, cam, p, what, colour_tol, min_checked=10000, max_boundary_share=0.08, sdf_tol=SDF_TOL, explain=None, max_unexplained=0):
    """fast / exact: (hash, heap, heapCounter, voxels) of the two volumes.  explain = (log, frames): the operator log [(kind, frame, T)] and its frames - every voxel
    that differs beyond the contract is then replayed per voxel (_explain) and must be reproduced by a neighbouring-pixel choice; the share of voxels near a pixel
    boundary is reported but no longer bounds anything (poses is ignored in favour of the log's)."""
    fh, fheap, fcnt, fvox = fast
    eh, eheap, ecnt, evox = exact
    assert fcnt == ecnt, what + ": heap counter"
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(fh[f], eh[f]), what + ": hash." + f
    assert np.array_equal(fheap, eheap), what + ": heap"
    n = len(evox)
    assert len(fvox) == n
    dist = _boundary_dist(eh["pos"], eh["ptr"], poses, cam, p.m_virtualVoxelSize, n)
    band = band_for(cam)
    m = dist < band
    live = (evox["weight"][:n] > 0) | (fvox["weight"][:n] > 0)
    chk = ~m
    dw = fvox["weight"][:n] != evox["weight"][:n]
    dsdf = np.abs(fvox["sdf"][:n].astype(np.float64) - evox["sdf"][:n].astype(np.float64))
    tol = sdf_tol * p.m_truncation
    fc, ec = fvox["color"].astype(np.int32), evox["color"].astype(np.int32)
    dcol = np.abs(fc - ec).reshape(n, -1)
    # the band this comparison actually needed: every voxel that differs beyond the contract lies this close to a pixel boundary
    bad = dw | (dsdf > tol) | (dcol.max(axis=1) > (colour_tol if colour_tol is not None else COLOUR_SEQ_MAX))
    needed = float(dist[bad].max(initial=0.0))
    hist = np.bincount(dcol[chk & live].reshape(-1), minlength=4)
    frac_gt1 = float(hist[2:].sum()) / max(int(hist.sum()), 1)
    share = float((m & live).sum()) / max(int(live.sum()), 1)
    trunc_band = p.m_truncation + p.m_truncScale * p.m_maxIntegrationDistance
    rep = dict(checked=int((chk & live).sum()), boundary_share=share, band=band, needed_band=needed, weights_differing_outside_band=int((dw & chk).sum()),
               max_dsdf=float(dsdf[chk].max(initial=0.0)), sdf_tol=tol, max_dcol=int(dcol[chk].max(initial=0)), colour_beyond_1lsb=frac_gt1,
               differing=int((dsdf[chk] > 0).sum()), colour_hist=hist[:10].tolist(), flips=int(bad.sum()),
               max_dsdf_in_band=float(dsdf[m].max(initial=0.0)), max_dweight_in_band=float(np.abs(fvox["weight"][:n][m] - evox["weight"][:n][m]).max(initial=0.0)))
    print(what + ":", rep)          # everything measured is on record before the first assertion
    assert rep["weights_differing_outside_band"] == 0, what + ": %d weights differ outside the boundary band (needed band %.2e px, band %.2e)" % (rep["weights_differing_outside_band"], needed, band)
    assert rep["max_dsdf"] <= tol, what + ": sdf deviates by %.3g (tolerance %.3g; needed band %.2e px, band %.2e)" % (rep["max_dsdf"], tol, needed, band)
    if colour_tol is not None:
        assert rep["max_dcol"] <= colour_tol, what + ": colour deviates by %d LSB" % rep["max_dcol"]
    else:
        # a sequence of operators: the de-integration of a colour is ill-conditioned - it multiplies any deviation of the stored byte by
        # w / (w - 1) (2 for a voxel of weight 2), the following integration by 0.8, so a 1 LSB conversion difference can grow by 1.6 per
        # re-integration of a weight-2 voxel UNDER EITHER CONTRACT (the exact contract's own rounding errors are amplified the same way
        # relative to real arithmetic).  Bound: almost every byte within 1 LSB, none beyond COLOUR_SEQ_MAX.
        assert frac_gt1 <= COLOUR_SEQ_FRAC and rep["max_dcol"] <= COLOUR_SEQ_MAX, what + ": colour histogram %s (%.2e beyond 1 LSB)" % (hist[:12].tolist(), frac_gt1)
    # boundary voxels: few, and still inside the truncation band / a plausible weight
    assert rep["checked"] >= min_checked, what + ": only %d voxels compared" % rep["checked"]
    if explain is not None:
        log, frames = explain
        bad_idx = np.nonzero(bad)[0]
        unexplained, oracle_missed, most = _explain(bad_idx, fvox, evox, eh, log, frames, cam, p, band, max(tol, 2e-6), COLOUR_SEQ_MAX if colour_tol is None else max(colour_tol, 1),
                                                    exists=_existence_gpu(bad_idx, eh, log, frames, cam, p))
        rep["explained"] = dict(differing_voxels=int(len(bad_idx)), unexplained=unexplained, oracle_not_reproduced=oracle_missed, most_alternatives=most, of_live_voxels=int(live.sum()))
        print(what + ": every live voxel compared; %d of %d differ beyond the contract, all within %.1e px of a pixel boundary; replayed per voxel with the neighbouring pixel as alternative: "
              "%d unexplained (the oracle's own value not reproduced for %d; at most %d alternatives per voxel)" % (len(bad_idx), int(live.sum()), band, unexplained, oracle_missed, most))
        # max_unexplained: only the 1280x960 @2 mm sweep passes a number (8): six of its 1.5e8 live voxels - the same six on every run and every build of round 5 -
        # hold a value one neighbouring-pixel sample away from the oracle's at an operator where this float64 replay does not put the projection inside twice the band
        assert unexplained <= max_unexplained, what + ": %d differing voxels are not explained by a neighbouring-pixel choice" % unexplained
        assert oracle_missed <= max(2, len(bad_idx) // 200), what + ": the per-voxel replay does not reproduce the oracle for %d voxels" % oracle_missed
    else:
        assert share < max_boundary_share, what + ": %.1f %% boundary voxels" % (100 * share)
    assert rep["max_dsdf_in_band"] <= 2 * trunc_band
    assert rep["max_dweight_in_band"] <= len(poses)
    return rep


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
    gf = gpu.capi.SceneRepHashSDF(p); gf.set_arith("exact")   # exact contract while the common state is built (the library default is fast; the test session selects exact)
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
    # pixel-boundary voxels only (the operator log of a pipeline is not replayed in float64 here - test_pipeline_baseline_gpu.py does that against the
    # oracle): measured on MI355X 3.4e-5 / 7.2e-4 / 8.9e-5 (gpurun r03); bounds = 10 x that (until round 4: 5e-3 / 2e-2 / 2e-2)
    assert share(dw) < 3.4e-4 and share(ds > tol) < 7.2e-3 and share(dc > 1) < 8.9e-4


@pytest.mark.parametrize("size", ["160x120@20mm", "640x480@4mm"])
def test_fast_contract_operators_are_bit_identical_run_to_run(gpu, size):
    """The same operators through two fresh scenes must not differ in ONE bit: integrations, fused re-integrations with translated and rotated poses, a
    de-integration, GC.  (Until round 5 this test compared two forms of k_update_apx; the speculative-load form was removed.  Run-to-run identity is not a
    formality for this kernel: see f2iHw in csrc/tsdf.hip.)"""
    variant = "second"
    W, H = (160, 120) if size.startswith("160") else (640, 480)
    voxel = 0.02 if W == 160 else 0.004
    frames = [synth.scene_room(k * 9, W, H) for k in range(5)]
    K = frames[0][3]
    cam = camera_params(W, H, K["fx"], K["fy"], K["mx"], K["my"])
    p = default_hash_params(num_buckets=50000 if W == 160 else 400000, num_sdf_blocks=40000 if W == 160 else 120000, voxel_size=voxel)
    dev = [_to_dev(f[0], f[1]) for f in frames]
    out = {}
    for lds in ("0", variant):
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
