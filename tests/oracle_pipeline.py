"""TEST INFRASTRUCTURE ONLY — CPU restatement of the per-frame orchestration of the hot path, driving the oracle
kernels of oracle/*.cpp (the checker, never the product):

  CUDAImageManager::process (CUDAImageManager.cpp:22-158), OnlineBundler (OnlineBundler.cpp:106-416, OnlineBundler.cu),
  Bundler (Bundler.cpp:91-394), SBA::align / removeMaxResidualCUDA (SBA.cpp:53-204), SIFTImageManager::fuseToGlobal /
  filterFrames (SIFTImageManager.cpp:367-476, :551-575), TrajectoryManager (TrajectoryManager.cpp), and the serial frame
  loop with integrate / deIntegrate / reintegrate (DepthSensing.cpp:723-762, :854-902, :966-1095).

The stage kernels it calls are pinned to the reference's device code where oracle/or_common.h says so, and this orchestration is pinned
END TO END to the reference's own host classes - CUDAImageManager.cpp, OnlineBundler.cpp, Bundler.cpp, SBA.cpp, CUDASolverBundling.cpp, CUDACache.cpp,
TrajectoryManager.cpp, SIFTImageManager.cpp compiled as they are into oracle/_ref - frame by frame on a three-chunk stream and on a stream
with a tracking loss (tests/test_ref_pin_cpu.py::test_online_bundler_vs_reference_host_code).  Not pinned: the DirectX frame loop of
DepthSensing.cpp around those classes (_integrate / process_frame here).
Plain Python loops: orchestration is a few hundred scalar decisions per frame.
"""
import os
from collections import deque

import numpy as np

from bundlefusion_amd.capi import ENTRYJ_DTYPE, rgbx_to_intensity, intrinsics_matrix, default_solver_config, camera_params, HashParams
from tests import oracle_api as o

NINF = np.float32(-np.inf)
MAX_RAW, MAX_FILT = 128, 25
INVALID = 0xFFFFFFFF


def _eye():
    return np.eye(4, dtype=np.float32)


def _minf():
    return np.full((4, 4), NINF, np.float32)


def scale_intrinsics(K, w_out, h_out, w_in, h_in):
    K = np.array(K, np.float32).copy()
    K[0, 0] = np.float32(K[0, 0]) * (np.float32(w_out) / np.float32(w_in))
    K[1, 1] = np.float32(K[1, 1]) * (np.float32(h_out) / np.float32(h_in))
    K[0, 2] = np.float32(K[0, 2]) * (np.float32(w_out - 1) / np.float32(w_in - 1))
    K[1, 2] = np.float32(K[1, 2]) * (np.float32(h_out - 1) / np.float32(h_in - 1))
    return K


class OBundler:
    """Bundler + SIFTImageManager + CUDACache + SBA for one set of images."""

    def __init__(self, max_images, max_keys, sift_intrinsics_inv, depth_intrinsics, is_local, gas, gbs):
        self.max_images, self.max_keys, self.is_local, self.gas, self.gbs = max_images, max_keys, is_local, gas, gbs
        self.Kinv = np.array(sift_intrinsics_inv, np.float32)
        self.K = o.inverse44(self.Kinv)
        self.depthK = np.array(depth_intrinsics, np.float32)
        self.trajectory = np.stack([_eye() for _ in range(max_images + 1)])
        self.continue_retry = 0
        self.revalidated_idx = INVALID
        n_its = max(gbs.s_numGlobalNonLinIterations, gbs.s_numLocalNonLinIterations)
        self.local_ws = [1.0] * n_its; self.local_wd = [float(i + 1) for i in range(n_its)]; self.local_wc = [0.0] * n_its
        self.global_ws = [1.0] * n_its; self.global_wd = [1.0] * n_its; self.global_wc = [0.1] * n_its
        for i in range(2, n_its):
            self.global_wd[i] = float(i)
        self.use_global_dense = False
        self.num_solves = 0
        self.cfg = default_solver_config()
        for f in ("optMaxResThresh", "denseDistThresh", "denseNormalThresh", "denseColorThresh", "denseColorGradientMin", "denseDepthMin", "denseDepthMax"):
            setattr(self.cfg, f, getattr(gbs, "s_" + f))
        self.cfg.denseOverlapCheckSubsampleFactor = gbs.s_denseOverlapCheckSubsampleFactor
        self.max_corr_per_image = int(min(max((25 * (max_images * (max_images - 1)) // 2) // max_images, 1000), 4000))
        W, H = gbs.s_downsampledWidth, gbs.s_downsampledHeight
        self.cacheK = scale_intrinsics(self.depthK, W, H, gas._depthW, gas._depthH)
        self.reset()

    # ---- SIFTImageManager state
    def reset(self):
        n = getattr(self, "num_images", 0)
        for i in range(n):
            self.trajectory[i] = _eye()
        self.num_images = 0
        self.current = 0
        self.keys = []; self.descs = []; self.cache = []
        self.allkeys = np.zeros((self.max_images * self.max_keys, 4), np.float32)
        self.valid = [0] * self.max_images
        self.valid[0] = 1
        self.corr = np.zeros(0, ENTRYJ_DTYPE); self.corr_keys = np.zeros((0, 2), np.uint32)
        self.retry = deque()
        self.num_filt = {}; self.filt_Tinv = {}

    def _add_image(self, keys, descs):
        i = self.num_images
        assert i < self.max_images
        self.keys.append(keys); self.descs.append(descs)
        self.allkeys[i * self.max_keys:(i + 1) * self.max_keys] = 0
        self.allkeys[i * self.max_keys:i * self.max_keys + len(keys)] = keys
        self.num_images += 1
        self.current = i

    def detect_features(self, intensity, depth_filt):
        n, keys, descs, _ = o.sift_run(intensity, depth_filt, depth_min=self.gas.s_sensorDepthMin, depth_max=self.gas.s_sensorDepthMax,
                                       min_key_scale=self.gbs.s_minKeyScale, feature_count_threshold=150, max_features=self.max_keys)
        if n < 0:
            raise RuntimeError("too many keypoints")
        self._add_image(keys, descs)

    def store_cached_frame(self, depth_raw, color):
        g = self.gbs
        self.cache.append(o.cache_store_frame(depth_raw, color, g.s_downsampledWidth, g.s_downsampledHeight, self.depthK, g.s_colorDownSigma,
                                              g.s_depthDownSigmaD, g.s_depthDownSigmaR))

    def copy_frame(self, other, frame):
        self._add_image(other.keys[frame].copy(), other.descs[frame].copy())
        self.cache.append(other.cache[frame])

    def add_invalid_frame(self):
        self.cache.append(None)
        self._add_image(np.zeros((0, 4), np.float32), np.zeros((0, 128), np.uint8))
        self.trajectory[self.num_images] = self.trajectory[self.num_images - 1]

    def is_valid(self):
        return any(self.valid[i] != 0 for i in range(1, self.num_images))

    # ---- Bundler::matchAndFilter (:103-249)
    def match_and_filter(self):
        g = self.gbs
        num_frames, cur = self.num_images, self.current
        assert num_frames > 1
        start = 0 if num_frames == cur + 1 else cur + 1
        if len(self.keys[cur]) == 0:
            return INVALID
        ratio = g.s_siftMatchRatioMaxLocal if self.is_local else g.s_siftMatchRatioMaxGlobal
        mk = self.max_keys
        self.num_filt = {}; self.filt_Tinv = {}
        filt = {}
        for prev in range(start, num_frames):
            if prev == cur:
                continue
            self.num_filt[prev] = 0
            if self.valid[prev] == 0 or len(self.keys[prev]) == 0:
                continue
            n, idx, dist = o.sift_match(self.descs[prev], self.descs[cur], g.s_siftMatchThresh, ratio, prev * mk, cur * mk)
            if n == 0 or cur == 0:
                continue
            m = min(n, MAX_RAW)
            pidx = np.zeros((MAX_RAW, 2), np.uint32); pidx[:m] = idx
            pdist = np.zeros(MAX_RAW, np.float32); pdist[:m] = dist
            min_matches = g.s_minNumMatchesLocal if self.is_local else g.s_minNumMatchesGlobal
            fn, fidx, fdist, fT = o.filter_matches(self.allkeys, pidx, pdist, m, self.Kinv, min_matches, g.s_maxKabschResidual2)
            self.filt_Tinv[prev] = o.inverse44(fT)
            if fn == 0:
                continue
            if not o.filter_surface_area(self.allkeys, fidx, self.Kinv, g.s_surfAreaPcaThresh)[0]:
                continue
            ok, _, _ = o.dense_verify(self.cache[prev], self.cache[cur], g.s_downsampledWidth, g.s_downsampledHeight, self.cacheK, fT, g.s_projCorrDistThres,
                                      g.s_projCorrNormalThres, g.s_verifySiftErrThresh, g.s_verifySiftCorrThresh, self.gas.s_sensorDepthMin, self.gas.s_sensorDepthMax)
            if not ok:
                continue
            self.num_filt[prev] = fn
            filt[prev] = fidx
        last = INVALID
        if cur > 0:
            connected = 0
            for i in range(num_frames - 1, start - 1, -1):                 # filterFrames
                if self.valid[i] != 0 and self.num_filt.get(i, 0) > 0 and i != cur:
                    connected, last = 1, i
                    break
            self.valid[cur] = connected
            if last != INVALID:                                            # AddCurrToResiduals, ascending previous image
                rows, krows = [], []
                for prev in sorted(filt):
                    for k in range(len(filt[prev])):
                        rows.append(o.make_entry(self.allkeys, filt[prev][k, 0], filt[prev][k, 1], prev, cur, self.Kinv))
                        krows.append(filt[prev][k])
                if rows:
                    self.corr = np.concatenate([self.corr, np.array(rows, dtype=ENTRYJ_DTYPE)])
                    self.corr_keys = np.concatenate([self.corr_keys, np.array(krows, np.uint32).reshape(-1, 2)])
            if not self.is_local:
                if last != INVALID and last + 1 != cur:
                    self.trajectory[cur] = self.trajectory[last]
                    if cur + 1 < self.max_images:
                        self.trajectory[cur + 1] = self.trajectory[last]
                if cur + 1 == num_frames:
                    if last != INVALID:
                        self.try_revalidation(cur, False)
                    else:
                        self.retry.appendleft(cur)
        return last

    def try_revalidation(self, cur_global, scan_done):
        self.revalidated_idx = INVALID
        if self.continue_retry < 0:
            return 0
        if self.retry:
            idx = self.retry.popleft()
            if scan_done:
                if self.continue_retry == 0:
                    self.continue_retry = idx
                elif self.continue_retry == idx:
                    self.continue_retry = -1
                    return self.revalidated_idx
            self.current = idx
            last = self.match_and_filter()
            if self.valid[idx] != 0:
                assert last != INVALID
                self.trajectory[idx] = self.trajectory[last]
                self.revalidated_idx = idx
            else:
                self.retry.appendleft(idx)
            self.current = cur_global
        return self.revalidated_idx

    # ---- SBA::align + Bundler::optimize
    def optimize(self, n_nonlin, n_lin, use_verify, remove_max_residual):
        g = self.gbs
        N = self.num_images
        assert N > 1
        if self.is_local:
            ws, wd, wc, use_cache = self.local_ws, (self.local_wd if g.s_useLocalDense else [0.0] * len(self.local_wd)), \
                (self.local_wc if g.s_useLocalDense else [0.0] * len(self.local_wc)), bool(g.s_useLocalDense)
        else:
            ws = self.global_ws
            if not self.use_global_dense:
                wd = [0.0] * len(self.global_wd); wc = [0.0] * len(self.global_wc); use_cache = False
            else:
                wd, wc, use_cache = self.global_wd, self.global_wc, True
        valid = np.array(self.valid[:N], np.int32)
        rot, trans = o.matrices_to_poses(self.trajectory[:N], valid)
        n_nonlin = min(n_nonlin, len(ws))
        cache = geom = None
        if use_cache:
            cache = [c if c is not None else self.cache[0] for c in self.cache[:N]]
            geom = (g.s_downsampledWidth, g.s_downsampledHeight, (self.cacheK[0, 0], self.cacheK[1, 1], self.cacheK[0, 2], self.cacheK[1, 2]))
        res = o.solver_solve(self.corr, valid, N, n_nonlin, n_lin, ws, wd, wc, rot, trans, cache, geom, self.cfg, True, self.max_corr_per_image)
        self.num_solves += 1
        removed = False
        if remove_max_residual and ws[0] > 0 and len(self.corr):
            e = self.corr[res["max_residual_index"]]
            i, j = int(e["imgIdx_i"]), int(e["imgIdx_j"])
            if not (i == 0 and j < 10) and res["max_residual"] > self.cfg.optMaxResThresh:
                rows = np.zeros(N, np.int64)                       # table built at solve start (before the invalidation)
                ok = self.corr["imgIdx_i"] != INVALID
                np.add.at(rows, self.corr["imgIdx_i"][ok].astype(np.int64), 1); np.add.at(rows, self.corr["imgIdx_j"][ok].astype(np.int64), 1)
                sel = (self.corr["imgIdx_i"] == i) & (self.corr["imgIdx_j"] == j)
                self.corr["imgIdx_i"][sel] = INVALID; self.corr["imgIdx_j"][sel] = INVALID
                self._check_invalid_frames(rows, N)
                removed = True
        verify = False
        if use_verify:
            # SBA.cpp:106-109; with no correspondences useVerification's 0 / 0 >= thresh is false (CUDASolverBundling.cpp:474)
            verify = (bool(len(self.corr)) and o.solver_use_verification(self.corr, rot, trans, N)) if ws[0] > 0 else True
        T = o.poses_to_matrices(rot, trans, valid)
        for i in range(N):
            if valid[i]:
                self.trajectory[i] = T[i]
        self.last_align = dict(removed=removed, use_verification=verify, max_residual=res["max_residual"], convergence=res["convergence"], gn_iterations=res["gn_iterations"])
        ok = True
        if verify:
            ok = self._verify_trajectory(N)
        return ok, removed

    def _check_invalid_frames(self, rows, num_vars):
        R = len(self.corr)
        if R == 0 or num_vars == 0:
            return
        if not self.gbs.s_useComprehensiveFrameInvalidation:
            for v in range(num_vars):
                if rows[v] == 0:
                    self.valid[v] = 0
            return
        gx, bx = (R + 127) // 128, (num_vars + 15) // 16           # launch arithmetic of SIFTImageManager.cu:746-749
        def in_var_set(v):
            return any((v - d) % gx == 0 and (v - d) // gx < bx for d in range(min(16, v + 1)))
        for v in range(num_vars):
            if rows[v] == 0 and in_var_set(v):
                self.valid[v] = 0

    def _verify_trajectory(self, N):
        g = self.gbs
        if N < 2:
            return False
        ok = True
        for blk in range(N * (N - 1) // 2):
            i0, i1 = blk // N, blk % N
            if i0 >= i1 or self.valid[i0] == 0 or self.valid[i1] == 0:
                continue
            T = o.mul44(o.inverse44(self.trajectory[i1]), self.trajectory[i0])
            v, _, _ = o.dense_verify(self.cache[i0], self.cache[i1], g.s_downsampledWidth, g.s_downsampledHeight, self.cacheK, T, g.s_projCorrDistThres,
                                     g.s_projCorrNormalThres, g.s_verifyOptErrThresh, g.s_verifyOptCorrThresh, 0.1, 3.0)
            ok = ok and v
        return ok

    # ---- SIFTImageManager::fuseToGlobal (.cpp:367-476)
    def fuse_to_global(self, glob):
        assert len(self.corr) > 0
        keys, descs = fuse_tracks(self.corr, self.corr_keys, self.trajectory, [len(k) for k in self.keys][:self.num_images], self.descs, self.allkeys,
                                  self.K, self.max_keys, glob.max_keys)
        n = len(keys)
        cur_desc = descs
        glob._add_image(keys[:n].copy(), np.array(cur_desc, np.uint8).reshape(-1, 128)[:n].copy())
        glob.cache.append(self.cache[0])


class OTrajectoryManager:
    def __init__(self, n_max, top_n, min_dist):
        self.opt = [_minf() for _ in range(n_max)]
        self.frames = [dict(type=1, idx=i, integrated=_minf(), dist=0.0) for i in range(n_max)]
        self.sort = []
        self.num_added = self.num_optimized = 0
        self.to_de, self.to_in, self.to_re = deque(), deque(), deque()
        self.top_n, self.min_dist = top_n, min_dist

    def add_frame(self, typ, T, idx):
        f = self.frames[idx]
        f["type"], f["integrated"] = typ, np.array(T, np.float32)
        self.opt[idx] = np.array(T, np.float32)
        self.sort.append(f)
        self.num_added += 1

    def update_optimized(self, traj, n):
        self.num_optimized = n
        for i in range(min(n, self.num_added)):
            self.opt[i] = np.array(traj[i], np.float32)

    def num_active(self):
        return len(self.to_de) + len(self.to_in) + len(self.to_re)

    # ---- the list consumers (getTopFrom*List + confirmIntegration, TrajectoryManager.cpp:113-175), with the return convention of the
    #      C ABI: (found, frame, T[, newT]).  Two situations the reference does not survive are defined here as in host.hip: a frame that
    #      loses its pose while it waits in the integrate list is dropped from it (the reference's consumer asserts on the -inf pose,
    #      DepthSensing.cpp:881), one that loses it while waiting for re-integration stays integrated at its old pose and is de-integrated
    #      by the next list update (the reference leaves it typed ReIntegration for good).
    def top_de(self):
        if not self.to_de:
            return False, 0, None
        f = self.to_de.popleft()
        return True, f["idx"], f["integrated"].copy()

    def top_in(self):
        while self.to_in:
            f = self.to_in.popleft()
            assert f["type"] == 2
            T = self.opt[f["idx"]].copy()
            if T[0, 0] == NINF:
                f["type"] = 3
                continue
            f["integrated"] = T
            return True, f["idx"], T.copy()
        return False, 0, None

    def top_re(self):
        if not self.to_re:
            return False, 0, None, None
        f = old = new = None
        while self.to_re:
            f = self.to_re.popleft()
            new = self.opt[f["idx"]].copy(); old = f["integrated"].copy()
            if new[0, 0] != NINF:
                f["integrated"] = new
                break
            f["type"] = 0
        return True, f["idx"], old, new

    def confirm(self, idx):
        self.frames[idx]["type"] = 0

    def _invalidate(self, f):
        if f["type"] == 3:
            return
        before = f["type"]; f["type"] = 3
        if before == 0:
            self.to_de.append(f)
        elif before == 2:
            self.to_in = deque(g for g in self.to_in if g is not f)

    def generate_update_lists(self):
        n = min(self.num_optimized, self.num_added)
        for i in range(n):
            f = self.frames[i]
            T = self.opt[i]
            if T[0, 0] == NINF:
                self._invalidate(f)
            else:
                if f["type"] in (1, 3):
                    f["type"] = 2
                    self.to_in.append(f)
                ro, to = o.matrices_to_poses(T[None]); ri, ti = o.matrices_to_poses(f["integrated"][None])
                s = np.float32(2.0)
                # (translation part, rotation vector) with the translation part doubled: PoseHelper.h:358-361, TrajectoryManager.cpp:70-77
                d = np.concatenate([ti[0] * s - to[0] * s, ri[0] - ro[0]]).astype(np.float32)
                acc = np.float32(0)
                for x in d:
                    acc = np.float32(acc + np.float32(x * x))
                f["dist"] = float(acc)
        head = self.sort[:n]
        import functools
        def cmp(l, r):
            def less(a, b):
                if a["type"] == 0 and b["type"] != 0:
                    return True
                if a["type"] != 0:
                    return False
                if a["dist"] != b["dist"]:
                    return a["dist"] > b["dist"]
                return a["idx"] < b["idx"]           # bit-equal distances: by frame index (the reference's std::sort leaves ties undefined)
            return -1 if less(l, r) else (1 if less(r, l) else 0)
        head.sort(key=functools.cmp_to_key(cmp))            # list.sort is stable
        self.sort[:n] = head
        for i in range(len(self.to_re), min(self.top_n, n)):
            f = self.sort[i]
            if f["dist"] > self.min_dist and f["type"] == 0:
                f["type"] = 4
                self.to_re.append(f)
            else:
                break


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box shows 256 logical CPUs and
    grants 16 cores' worth of time: 256 OpenMP threads on that quota are slower than one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


class OraclePipeline:
    """The serial frame loop on the CPU.  `gas`/`gbs` are the ctypes parameter structs of bundlefusion_amd.capi.

    solve_lag = L > 0 restates the product's lagged-solve mode (bf_pipeline_set_solve_lag; the reference runs its optimiser on a second thread
    with no defined hand-over, FriedLiver.cpp:112-143): the solves of a chunk-closing frame b are computed where the serial loop computes them,
    but what they publish - the complete trajectory, the last valid complete transform, the TrajectoryManager's optimised poses, the tracking-lost
    flag - becomes visible when frame b + L enters the loop (and before the first iteration past the end of the sequence at the latest)."""

    def __init__(self, gas, gbs, width, height, K, solve_lag=0):
        self.solve_lag = solve_lag
        self._pending = None                  # (apply_at, complete, n_total, last_valid_complete or None, tracking_lost or None)
        self._stage = None                    # collects what the solves of the current frame publish (lagged mode)
        self.gas, self.gbs, self.W, self.H = gas, gbs, width, height
        self.K = np.array(K, np.float32)
        gas._depthW, gas._depthH = width, height                 # depth and colour cameras coincide in the synthetic sensor
        S = gbs.s_submapSize
        self.S = S
        self.siftK = scale_intrinsics(self.K, gbs.s_widthSIFT, gbs.s_heightSIFT, width, height)
        self.siftKinv = o.inverse44(self.siftK)
        mk = gbs.s_maxNumKeysPerImage
        self.local = OBundler(S + 1, mk, self.siftKinv, self.K, True, gas, gbs)
        self.opt_local = OBundler(S + 1, mk, self.siftKinv, self.K, True, gas, gbs)
        self.glob = OBundler(gbs.s_maxNumImages, mk, self.siftKinv, self.K, False, gas, gbs)
        n_all = gbs.s_maxNumImages * S
        self.tm = OTrajectoryManager(n_all, gas.s_topNActive, gas.s_minPoseDistSqrt)
        self.complete = np.zeros((n_all, 4, 4), np.float32)
        self.local_traj = np.stack([_eye() for _ in range(gbs.s_maxNumImages * (S + 1))])
        self.sift_traj = np.zeros((n_all, 4, 4), np.float32); self.sift_traj[0] = _eye()
        self.local_valid = [[] for _ in range(gbs.s_maxNumImages)]
        self.invalid_list = [1] * n_all
        self.cur_T = [_minf() for _ in range(n_all)]; self.cur_T[0] = _eye()
        self.last_processed, self.last_valid = -1, False
        self.local_to_solve, self.last_local_solved = -1, -1
        self.past_end = 0; self.num_complete = 0; self.last_valid_complete = 0
        self.tracking_lost = False
        self.state = "NONE"
        self.use_solve = True
        self.total_opt_local = 0
        self.num_opt_per_removal = max(gbs.s_numOptPerResidualRemoval, 1)
        self.num_global_nl = gbs.s_numGlobalNonLinIterations
        self.frames = []                      # (depth, color) at integration resolution
        self.integrate_ops = []               # log of (kind, frame, T)
        self.replay_log = []                  # the same with the garbage collections in between: ("in" | "de" | "gc", frame, T)
        hp = HashParams()
        eye = np.eye(4, dtype=np.float32).reshape(16)
        for i in range(16):
            hp.m_rigidTransform[i] = hp.m_rigidTransformInverse[i] = float(eye[i])
        hp.m_hashNumBuckets, hp.m_hashBucketSize, hp.m_hashMaxCollisionLinkedListSize = gas.s_hashNumBuckets, 4, gas.s_hashMaxCollisionLinkedListSize
        hp.m_SDFBlockSize, hp.m_numSDFBlocks, hp.m_virtualVoxelSize = 8, gas.s_hashNumSDFBlocks, gas.s_SDFVoxelSize
        hp.m_maxIntegrationDistance, hp.m_truncation, hp.m_truncScale = gas.s_SDFMaxIntegrationDistance, gas.s_SDFTruncation, gas.s_SDFTruncationScale
        hp.m_integrationWeightSample, hp.m_integrationWeightMax = gas.s_SDFIntegrationWeightSample, gas.s_SDFIntegrationWeightMax
        self.scene = o.OracleScene(hp)
        Ki = scale_intrinsics(self.K, gas.s_integrationWidth, gas.s_integrationHeight, width, height)
        self.cam = camera_params(gas.s_integrationWidth, gas.s_integrationHeight, float(Ki[0, 0]), float(Ki[1, 1]), float(Ki[0, 2]), float(Ki[1, 2]),
                                 gas.s_renderDepthMin, gas.s_renderDepthMax)
        import os
        self.threads = effective_cpus()             # voxel update (per call) and, through set_threads(), the image-space loops
        o.set_threads(min(self.threads, 32))        # image rows: more threads than that only add barrier traffic

    # ---- CUDAImageManager::process
    def _ingest(self, depth, color):
        g = self.gbs
        raw = np.ascontiguousarray(depth, np.float32)
        filt = raw
        if g.s_erodeSIFTdepth:
            raw = o.erode_depth(o.erode_depth(raw, 3, 0.05, 0.3), 3, 0.05, 0.3)
        filt = o.gauss_filter_depth(raw, g.s_depthSigmaD, g.s_depthSigmaR) if g.s_depthFilter else raw.copy()
        wi, hi = self.gas.s_integrationWidth, self.gas.s_integrationHeight
        color = np.ascontiguousarray(color, np.uint8)
        if (wi, hi) == (self.W, self.H):      # CUDAImageManager.cpp:43-50, :122-136
            self.frames.append((filt if g.s_erodeSIFTdepth else raw, color))
        else:                                 # resampling branch, :52-61 / :138-149 (the reference default: 640x480 sensor, 320x240 integration)
            self.frames.append((o.resample_float(filt, wi, hi), o.resample_uchar4(color.reshape(self.H, self.W, 4), wi, hi)))
        return raw, filt

    def is_last_local(self, cur):
        return cur >= self.S and cur % self.S == 0

    def _prepare_local_solve(self, cur, seq_end):
        self.state = "NONE"
        idx = (max(cur, 1) - 1) // self.S
        if seq_end and cur % self.S == 0:
            idx += 1
            self.local_to_solve = -(idx + 2); self.state = "INVALIDATE"
        elif self.local.is_valid():
            self.local_to_solve = idx; self.state = "PROCESS"
        else:
            self.local_to_solve = -(idx + 2); self.state = "INVALIDATE"
        self.local, self.opt_local = self.opt_local, self.local

    def process_input(self, raw=None, filt=None, color=None):
        cur = len(self.frames) - 1
        last_local = self.is_last_local(cur)
        if cur > 0 and self.last_processed == cur:
            if self.past_end == 0 and self.local_to_solve == -1 and not last_local:
                self._prepare_local_solve(cur, True)
            nb = self.gas.s_numSolveFramesBeforeExit
            if nb != INVALID:
                if self.past_end == nb and self.last_processed < 10000:
                    self.num_global_nl = 3
                    self.glob.global_ws, self.glob.global_wd, self.glob.global_wc = [1.0] * 3, [15.0] * 3, [0.0] * 3
                    self.glob.use_global_dense = True
                if self.past_end == nb + 1:
                    self.use_solve = False
            self.past_end += 1
            return
        intensity = rgbx_to_intensity(color)             # SIFT resolution == colour resolution in the synthetic sensor
        if self.gas.s_colorFilter:
            intensity = o.gauss_filter_intensity(intensity, self.gas.s_colorSigmaD)
        self.local.detect_features(intensity, filt)
        self.local.store_cached_frame(raw, color)
        cl = self.local.current
        if last_local:
            self.opt_local.copy_frame(self.local, cl)
        self.last_valid = True
        if cl > 0:
            last = self.local.match_and_filter()
            self.last_valid = last != INVALID
            if not self.last_valid:
                self.cur_T[cur] = _minf()
                self.sift_traj[cur] = self.sift_traj[cur - 1]
            else:
                for i in range(cl - 1, -1, -1):                    # getSiftTransformCU_Kernel
                    if self.local.num_filt.get(i, 0) > 0:
                        prev_known = cur - (cl - i)
                        Tinv = self.local.filt_Tinv[i]
                        self.sift_traj[cur] = o.mul44(self.sift_traj[prev_known], Tinv)
                        if self.last_valid_complete == 0:
                            T = self.sift_traj[cur].copy()
                        elif prev_known < self.last_valid_complete:
                            T = o.mul44(self.complete[prev_known], Tinv)
                        else:
                            off = o.mul44(o.inverse44(self.sift_traj[self.last_valid_complete]), self.sift_traj[prev_known])
                            T = o.mul44(o.mul44(self.complete[self.last_valid_complete], off), Tinv)
                        self.cur_T[cur] = T
                        break
        if last_local:
            self._prepare_local_solve(cur, False)
        self.last_processed = cur

    # ---- OnlineBundler::process
    def _optimize_local(self):
        g = self.gbs
        if self.state == "NONE":
            return
        st, self.state = self.state, "NONE"
        n_local = min(self.S, self.opt_local.num_images)
        if st == "PROCESS":
            idx = self.local_to_solve
            ok, _ = self.opt_local.optimize(g.s_numLocalNonLinIterations, g.s_numLocalLinIterations, bool(g.s_useLocalVerify), False)
            if ok:
                self.local_traj[(self.S + 1) * idx:(self.S + 1) * (idx + 1)] = self.opt_local.trajectory[:self.S + 1]
                self.state = "PROCESS"
            else:
                self.state = "INVALIDATE"
        else:
            idx = -self.local_to_solve - 2
            self.state = "INVALIDATE"
        self.local_to_solve = -1
        self.last_local_solved = idx
        self.total_opt_local = self.S * idx + n_local

    def _process_global(self):
        S = self.S
        st = self.state
        if st == "NONE":
            if self.past_end != 0:
                idx = self.glob.try_revalidation(self.last_local_solved, True)
                if idx != INVALID and idx < len(self.local_valid):
                    for i, v in enumerate(self.local_valid[idx]):
                        if v == 1:
                            self.invalid_list[idx * S + i] = 1
                    self.state = "PROCESS"
            return
        self.state = "NONE"
        if st == "PROCESS":
            self.opt_local.fuse_to_global(self.glob)
            cg = self.glob.current
            vl = list(self.opt_local.valid[:S + 1])
            n_opt = self.opt_local.num_images
            n_local = min(S, n_opt)
            last_valid_local = 0
            for i in range(n_opt - 1, -1, -1):
                if vl[i]:
                    last_valid_local = i
                    break
            for i in range(n_local):
                if vl[i] == 0:
                    self.invalid_list[cg * S + i] = 0
            self.local_valid[cg] = vl[:n_local]
            ng = self.glob.num_images
            self.glob.trajectory[ng] = o.mul44(self.glob.trajectory[cg], self.local_traj[ng * (S + 1) - ((S + 1) - last_valid_local)])
            self.opt_local.reset()
            if ng > 1:
                last = self.glob.match_and_filter()
                if last == INVALID:
                    self._set_tracking_lost(True); self.state = "INVALIDATE"
                else:
                    self._set_tracking_lost(False)
                    r = self.glob.revalidated_idx
                    if r != INVALID:
                        for i, v in enumerate(self.local_valid[r]):
                            if v == 1:
                                self.invalid_list[r * S + i] = 1
                    self.state = "PROCESS"
        else:
            self.state = "INVALIDATE"
            self.glob.add_invalid_frame()
            self.opt_local.reset()
            for i in range(S * self.last_local_solved, self.total_opt_local):
                self.invalid_list[i] = 0

    def _set_tracking_lost(self, v):
        if self._stage is not None:
            self._stage["lost"] = v
        else:
            self.tracking_lost = v

    def _update_trajectory(self, n):
        S = self.S
        out = self.complete if self._stage is None else self.complete.copy()
        for i in range(n):
            if self.invalid_list[i] == 0:
                out[i] = _minf()
            else:
                out[i] = o.mul44(self.glob.trajectory[i // S], self.local_traj[(i // S) * (S + 1) + i % S])
        return out

    def _publish(self, complete, n_total, last_valid):
        """What a global optimisation makes visible (OnlineBundler.cpp:394-401); last_valid None: unchanged."""
        if self._stage is not None:
            self._stage.update(complete=complete, n_total=n_total, last_valid=last_valid)
            return
        self.complete = complete
        self.tm.update_optimized(self.complete, n_total)
        self.num_complete = n_total
        if last_valid is not None:
            self.last_valid_complete = last_valid

    def _apply_pending(self):
        if self._pending is None:
            return
        st, self._pending = self._pending[1], None
        if "complete" in st:
            self._publish(st["complete"], st["n_total"], st["last_valid"])
        if "lost" in st:
            self.tracking_lost = st["lost"]

    def _optimize_global(self):
        g = self.gbs
        done = self.past_end > 0
        if not done and self.state == "NONE":
            return
        if self.last_local_solved < 0:
            return
        st = "PROCESS" if done else self.state
        n_total = self.total_opt_local
        if st == "PROCESS":
            ng = self.glob.num_images
            count = self.past_end if self.past_end > 0 else n_total // self.S
            remove = (count % self.num_opt_per_removal) == (self.num_opt_per_removal - 1)
            ok, removed = True, False
            if ng > 1:
                ok, removed = self.glob.optimize(self.num_global_nl, g.s_numGlobalLinIterations, False, remove)
            if removed:
                for i in range(ng):
                    if self.glob.valid[i] == 0:
                        for k in range(i * self.S, min((i + 1) * self.S, n_total)):
                            self.invalid_list[k] = 0
            self._publish(self._update_trajectory(n_total), n_total, self.S * self.last_local_solved if ok else None)
        elif st == "INVALIDATE":
            assert self.glob.num_images > 1, "INVALID_FIRST_CHUNK"
            self.glob.valid[self.glob.num_images - 1] = 0
            for i in range(self.S * self.last_local_solved, self.total_opt_local):
                self.invalid_list[i] = 0
            self._publish(self._update_trajectory(n_total), n_total, None)
        self.state = "NONE"

    def _bundler_process(self):
        if not self.use_solve:
            return
        lagged = self.solve_lag > 0 and self.past_end == 0 and self.state != "NONE"
        if lagged:
            assert self._pending is None, "the previous chunk's lagged solve has not been applied yet"
            self._stage = {}
        self._optimize_local()
        self._process_global()
        self._optimize_global()
        if lagged:
            self._pending, self._stage = (len(self.frames) - 1 + self.solve_lag, self._stage), None

    # ---- integrate / reintegrate
    def _integrate(self, frame, T, de):
        self.integrate_ops.append(("de" if de else "in", frame, np.array(T, np.float32)))
        self.replay_log.append(self.integrate_ops[-1])
        d, c = self.frames[frame]
        (self.scene.deintegrate if de else self.scene.integrate)(T, d, c, self.cam, threads=self.threads)

    def _reintegrate(self):
        tm, mx = self.tm, self.gas.s_maxFrameFixes
        if tm.num_active() < mx:
            tm.generate_update_lists()
        for _ in range(mx):
            found, idx, T = tm.top_de()
            if found:
                self._integrate(idx, T, True)
                continue
            found, idx, T = tm.top_in()
            if found:
                self._integrate(idx, T, False); tm.confirm(idx)
                continue
            found, idx, old, new = tm.top_re()
            if found:
                if new[0, 0] == NINF:
                    continue
                self._integrate(idx, old, True)
                self._integrate(idx, new, False)
                tm.confirm(idx)
                continue
            break
        if self.gas.s_garbageCollectionEnabled:
            self.scene.garbage_collect()
            self.replay_log.append(("gc", -1, None))

    def process_frame(self, depth, color):
        if self._pending is not None and len(self.frames) >= self._pending[0]:        # the frame entering the loop is frame len(self.frames)
            self._apply_pending()
        raw, filt = self._ingest(depth, color)
        self.process_input(raw, filt, color)
        self._reintegrate()
        cur = len(self.frames) - 1
        if self.last_valid and self.gas.s_reconstructionEnabled:
            T = self.cur_T[self.last_processed]
            self._integrate(self.last_processed, T, False)
            self.tm.add_frame(0, T, cur)
        else:
            self.tm.add_frame(1, _minf(), cur)
        self._bundler_process()

    def process_end_of_sequence(self):
        self._apply_pending()
        self.process_input()
        self._reintegrate()
        self._bundler_process()
        return self.tm.num_active()

    def integrated_trajectory(self):
        out = []
        for i in range(self.tm.num_added):
            f = self.tm.frames[i]
            out.append(f["integrated"] if f["type"] in (0, 4) else _minf())
        return np.stack(out) if out else np.zeros((0, 4, 4), np.float32)


def fuse_tracks(corr, corr_keys, T, num_keys_per_image, descs, allkeys, K, mk, glob_max_keys):
    """computeTracks + fuseToGlobal (SIFTImageManager.cpp:367-468): correspondences (EntryJ rows + key index pairs, image * mk + key) and the chunk's
    trajectory -> fused key points (n, 4) and their descriptors (n, 128) of the chunk's key frame."""
    corr_per_key = {}
    def xf(M, p):          # float4x4 * float3 with w = 1, same operation order as the device code
        f = np.float32
        return np.array([f(f(f(f(M[r, 0] * p[0]) + f(M[r, 1] * p[1])) + f(M[r, 2] * p[2])) + f(M[r, 3] * f(1.0))) for r in range(3)], np.float32)
    for c, k in zip(corr, corr_keys):
        if c["imgIdx_i"] == INVALID:
            continue
        i, j = int(c["imgIdx_i"]), int(c["imgIdx_j"])
        pi, pj = c["pos_i"].astype(np.float32), c["pos_j"].astype(np.float32)
        d = xf(T[i], pi) - xf(T[j], pj)
        err = np.sqrt(np.float32(np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])))
        if err < np.float32(0.03):
            corr_per_key.setdefault(int(k[0]), []).append(((j, int(k[1])), pj))
            corr_per_key.setdefault(int(k[1]), []).append(((i, int(k[0])), pi))
        else:
            corr_per_key.setdefault(int(k[0]), []).append(((j, int(k[1])), None))
            corr_per_key.setdefault(int(k[1]), []).append(((i, int(k[0])), None))
    marker = set()
    tracks = []
    def find_track(track, key):
        stack = [(key, 0)]
        while stack:
            kk, pos = stack.pop()
            lst = corr_per_key.get(kk, [])
            if pos >= len(lst):
                continue
            stack.append((kk, pos + 1))
            (img, ky), p = lst[pos]
            if ky not in marker:
                track.append(((img, ky), p))
                marker.add(ky)
                stack.append((ky, 0))
    for i in range(len(num_keys_per_image)):
        for k in range(num_keys_per_image[i]):
            if not tracks or tracks[-1]:
                tracks.append([])
            find_track(tracks[-1], i * mk + k)
    cur_keys, cur_desc = [], []
    for tr in tracks:
        if not tr:
            continue
        rep = tr[0]
        pos = np.zeros(3, np.float32); num = 0
        for (img, ky), p in tr:
            if p is not None:
                pos = (pos + xf(T[img], p)).astype(np.float32); num += 1
        if num > 0:
            pos = (pos / np.float32(num)).astype(np.float32)
            pos = xf(K, pos)
            cur_keys.append([np.float32(pos[0] / pos[2]), np.float32(pos[1] / pos[2]), allkeys[rep[0][1], 2], pos[2]])
            cur_desc.append(descs[rep[0][1] // mk][rep[0][1] % mk])
    n = min(len(cur_keys), glob_max_keys)
    keys = np.array(cur_keys, np.float32).reshape(-1, 4)
    if len(cur_keys) > glob_max_keys:
        keys = keys[np.argsort(keys[:, 3], kind="stable")]
    return keys[:n].copy(), np.array(cur_desc, np.uint8).reshape(-1, 128)[:n].copy()
