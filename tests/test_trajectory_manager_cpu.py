"""CPU test of the TrajectoryManager host logic (TrajectoryManager.cpp:24-200) through the C ABI,
against the Python restatement the oracle frame loop uses (tests/oracle_pipeline.py::OTrajectoryManager +
its list consumers).  Both are driven by the same random script of frame-loop events, which includes the two situations the reference
does not survive (a frame losing its pose while it waits in the integrate / re-integrate list); against the reference's own
TrajectoryManager.cpp, without those: tests/test_ref_pin_cpu.py::test_trajectory_manager_vs_reference_host_code."""
import ctypes as C

import numpy as np

from bundlefusion_amd.capi import lib, check
from tests.oracle_pipeline import OTrajectoryManager, NINF, _minf


def _pose(rng, scale):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = scale * rng.uniform(0.0, 1.0)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T[:3, 3] = scale * rng.normal(size=3)
    return T.astype(np.float32)


class _CTM:
    def __init__(self, n, top_n, min_dist):
        self.h = C.c_void_p()
        check(lib.bf_trajectory_manager_create(n, top_n, C.c_float(min_dist), C.byref(self.h)))

    def close(self):
        lib.bf_trajectory_manager_destroy(self.h)

    def add(self, typ, T, idx):
        check(lib.bf_trajectory_manager_add_frame(self.h, typ, np.ascontiguousarray(T, np.float32).ctypes.data_as(C.POINTER(C.c_float)), idx))

    def update(self, traj):
        a = np.ascontiguousarray(traj, np.float32)
        check(lib.bf_trajectory_manager_update_optimized_transform_host(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), len(a)))

    def generate(self):
        check(lib.bf_trajectory_manager_generate_update_lists(self.h))

    def active(self):
        n = C.c_uint32()
        check(lib.bf_trajectory_manager_get_num_active_operations(self.h, C.byref(n)))
        return n.value

    def _top(self, fn, two):
        a, b = (C.c_float * 16)(), (C.c_float * 16)()
        idx, found = C.c_uint32(), C.c_int()
        if two:
            check(fn(self.h, a, b, C.byref(idx), C.byref(found)))
        else:
            check(fn(self.h, a, C.byref(idx), C.byref(found)))
        return bool(found.value), idx.value, np.array(a, np.float32).reshape(4, 4), np.array(b, np.float32).reshape(4, 4)

    def top_de(self):
        return self._top(lib.bf_trajectory_manager_get_top_from_deintegrate_list, False)

    def top_in(self):
        return self._top(lib.bf_trajectory_manager_get_top_from_integrate_list, False)

    def top_re(self):
        return self._top(lib.bf_trajectory_manager_get_top_from_reintegrate_list, True)

    def confirm(self, idx):
        check(lib.bf_trajectory_manager_confirm_integration(self.h, idx))

    def frame(self, idx):
        t, d = C.c_int(), C.c_float()
        T = (C.c_float * 16)()
        check(lib.bf_trajectory_manager_get_frame(self.h, idx, C.byref(t), T, C.byref(d)))
        return t.value, np.array(T, np.float32).reshape(4, 4), d.value


def _same(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def test_trajectory_manager_matches_restatement():
    lib.bf_trajectory_manager_update_optimized_transform_host.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
    for seed, top_n, min_dist, max_fixes in ((0, 30, 0.0, 10), (1, 5, 0.0004, 3), (2, 8, 0.0, 1)):
        rng = np.random.default_rng(seed)
        n_max = 120
        c, o = _CTM(n_max, top_n, min_dist), OTrajectoryManager(n_max, top_n, min_dist)
        gt = [_pose(rng, 0.5) for _ in range(n_max)]
        ops = []
        for frame in range(n_max):
            # ---- reintegrate() (DepthSensing.cpp:854-902) on both sides, logging what would be (de-)integrated
            if c.active() < max_fixes:
                c.generate()
            if o.num_active() < max_fixes:
                o.generate_update_lists()
            assert c.active() == o.num_active()
            for _ in range(max_fixes):
                f, idx, T, _ = c.top_de()
                g = o.top_de()
                assert f == g[0]
                if f:
                    assert idx == g[1] and _same(T, g[2])
                    ops.append(("de", idx)); continue
                f, idx, T, _ = c.top_in()
                g = o.top_in()
                assert f == g[0]
                if f:
                    assert idx == g[1] and _same(T, g[2])
                    c.confirm(idx); o.confirm(idx)
                    ops.append(("in", idx)); continue
                f, idx, oldT, newT = c.top_re()
                g = o.top_re()
                assert f == g[0]
                if f:
                    assert idx == g[1] and _same(oldT, g[2]) and _same(newT, g[3])
                    if newT[0, 0] == NINF:
                        continue
                    c.confirm(idx); o.confirm(idx)
                    ops.append(("re", idx)); continue
                break
            # ---- the new frame: tracked (integrated at a slightly wrong pose) or not
            if rng.random() < 0.85:
                T = (gt[frame].astype(np.float64) @ _pose(rng, 0.01).astype(np.float64)).astype(np.float32)
                c.add(0, T, frame); o.add_frame(0, T, frame)
            else:
                c.add(1, _minf(), frame); o.add_frame(1, _minf(), frame)
            # ---- every 10 frames an optimisation result arrives: better poses, some frames invalidated, some recovered
            if frame % 10 == 9:
                n = frame + 1 - int(rng.integers(0, 3))
                traj = np.stack([(gt[i].astype(np.float64) @ _pose(rng, 0.002).astype(np.float64)).astype(np.float32) for i in range(n)])
                bad = rng.random(n) < 0.08
                traj[bad] = -np.inf
                c.update(traj); o.update_optimized(traj, n)
            for i in range(frame + 1):
                t, T, d = c.frame(i)
                g = o.frames[i]
                assert t == g["type"], (seed, frame, i)
                assert _same(T, g["integrated"])
                assert np.float32(d).view(np.uint32) == np.float32(g["dist"]).view(np.uint32), (seed, frame, i, d, g["dist"])
        kinds = {k for k, _ in ops}
        assert kinds == {"de", "in", "re"}, kinds            # the script exercised all three lists
        c.close()


def test_frame_invalidated_while_queued_for_reintegration_is_deintegrated_once():
    """A frame sits in the re-integration list when an optimisation result invalidates it (optimised pose -inf).  Popping the
    list must not touch the volume for it, and the NEXT list update must queue exactly one de-integration at the pose the frame
    is integrated with; when the frame becomes valid again it is integrated once.  (The reference leaves such a frame typed
    ReIntegration, which its invalidateFrame never de-integrates — TrajectoryManager.cpp:123-134 vs :191-199; host.hip documents
    the deviation.)"""
    lib.bf_trajectory_manager_update_optimized_transform_host.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
    rng = np.random.default_rng(7)
    n = 6
    c = _CTM(16, 30, 0.0)
    T0 = [_pose(rng, 0.5) for _ in range(n)]
    for i in range(n):
        c.add(0, T0[i], i)                                   # Integrated at T0
    T1 = np.stack([(T0[i].astype(np.float64) @ _pose(rng, 0.01).astype(np.float64)).astype(np.float32) for i in range(n)])
    c.update(T1); c.generate()
    assert c.active() == n and all(c.frame(i)[0] == 4 for i in range(n))      # all queued for re-integration
    order = []
    f, idx, oldT, newT = c.top_re(); assert f and newT[0, 0] != NINF; c.confirm(idx); order.append(idx)
    T2 = T1.copy()
    victims = [i for i in range(n) if i != idx][:2]
    T2[victims] = -np.inf                                    # two queued frames lose their pose
    c.update(T2)
    popped = []
    while True:
        f, i2, oldT, newT = c.top_re()
        if not f:
            break
        if newT[0, 0] == NINF:
            continue                                         # the frame loop skips the volume operation (plReintegrate)
        c.confirm(i2); popped.append(i2)
    assert sorted(popped + order) == [i for i in range(n) if i not in victims]
    for v in victims:
        t, T, _ = c.frame(v)
        assert t == 0 and _same(T, T0[v])                    # still integrated at the old pose, and typed so
    c.generate()
    des = []
    while True:
        f, i3, T = c.top_de()[:3]
        if not f:
            break
        des.append(i3); assert _same(T, T0[i3])
    assert sorted(des) == sorted(victims)                    # de-integrated exactly once, at the pose it was integrated with
    assert all(c.frame(v)[0] == 3 for v in victims)          # Invalid
    c.generate()
    assert not c.top_de()[0]                                 # not a second time
    T3 = T2.copy(); T3[victims] = T1[victims]
    c.update(T3); c.generate()                               # valid again -> integrate list, once
    ins = []
    while True:
        f, i4, T = c.top_in()[:3]
        if not f:
            break
        c.confirm(i4); ins.append(i4); assert _same(T, T1[i4])
    assert sorted(ins) == sorted(victims)
    c.close()


def test_frame_losing_its_pose_while_queued_for_integration_is_not_integrated():
    """A never-integrated frame gets a pose (-> integrate list) and loses it again before the list is consumed.  Whether the loss is seen
    by the next list update or only at the pop, nothing is integrated, the list is left clean and the frame is integrated exactly once
    when a pose comes back.  (The reference keeps the frame in m_toIntegrateList and stops at an assert when it is popped,
    TrajectoryManager.cpp:142-146 / DepthSensing.cpp:881.)"""
    lib.bf_trajectory_manager_update_optimized_transform_host.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
    rng = np.random.default_rng(11)
    for seen_by_update in (True, False):
        c, o = _CTM(8, 30, 0.0), OTrajectoryManager(8, 30, 0.0)
        T = np.stack([_pose(rng, 0.5) for _ in range(3)])
        for m_add in (lambda *a: c.add(*a), lambda *a: o.add_frame(*a)):
            m_add(0, T[0], 0); m_add(1, _minf(), 1); m_add(1, _minf(), 2)
        c.update(T); o.update_optimized(T, 3); c.generate(); o.generate_update_lists()
        assert c.active() == o.num_active() == 2 and c.frame(1)[0] == c.frame(2)[0] == 2
        T2 = T.copy(); T2[1] = -np.inf
        c.update(T2); o.update_optimized(T2, 3)
        if seen_by_update:
            c.generate(); o.generate_update_lists()
            assert c.active() == o.num_active() == 1
        f, idx, Tin, _ = c.top_in(); g = o.top_in()
        assert f and g[0] and idx == g[1] == 2 and _same(Tin, T[2]) and _same(g[2], T[2])       # frame 1 was skipped
        c.confirm(2); o.confirm(2)
        assert not c.top_in()[0] and not o.top_in()[0] and c.active() == o.num_active() == 0
        assert c.frame(1)[0] == o.frames[1]["type"] == 3
        c.update(T); o.update_optimized(T, 3); c.generate(); o.generate_update_lists()
        f, idx, Tin, _ = c.top_in(); g = o.top_in()
        assert f and g[0] and idx == g[1] == 1 and _same(Tin, T[1])
        c.confirm(1); o.confirm(1)
        c.generate(); o.generate_update_lists()
        assert c.active() == o.num_active() == 0 and all(c.frame(i)[0] == 0 and np.isfinite(c.frame(i)[2]) for i in range(3))
        c.close()
