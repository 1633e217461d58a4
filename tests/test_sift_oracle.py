"""CPU tests (-m "not gpu"): SIFT oracle sanity + the deterministic elementary functions.

Independent sanity of the oracle (its pin against the reference's own SiftGPU fork is tests/test_ref_pin_cpu.py).  Checked here: the pyramid against a direct numpy
convolution with the same taps, descriptor normalisation (|d| = 512 +- rounding), detection gated by
valid depth, keypoints invariant under an integer image shift (structure-level repeatability).
"""
import numpy as np
import pytest

from bundlefusion_amd import synth
from bundlefusion_amd.capi import rgbx_to_intensity


def _frame(k=200, w=640, h=480):
    d, c, T, K = synth.scene_room(k, w, h)
    return rgbx_to_intensity(c), d


def test_pyramid_level_matches_numpy_convolution(oracle):
    I, d = _frame()
    sigma0 = np.float32(1.6) * np.float32(2.0) ** np.float32(1 / 3)
    sa = sigma0 * np.float32(2.0) ** np.float32(-1 / 3)
    init = np.sqrt(sa * sa - 0.25)

    def taps(sigma):
        sz = int(np.ceil(4.0 * sigma - 0.5)); x = np.arange(-sz, sz + 1)
        k = np.exp(-0.5 * x * x / (sigma * sigma)); return k / k.sum()
    k = taps(float(init))
    pad = len(k) // 2
    ref = np.pad(I.astype(np.float64), pad, mode="edge")
    ref = np.stack([np.convolve(r, k[::-1], mode="valid") for r in ref])
    ref = np.stack([np.convolve(c, k[::-1], mode="valid") for c in ref.T]).T
    got = oracle.sift_pyramid_level(I, 0, 0)
    assert np.abs(got - ref).max() < 2e-6
    # octave 1 level -1 is the x2 point sample of octave 0 level 2 (array index 3)
    assert np.array_equal(oracle.sift_pyramid_level(I, 1, 0), oracle.sift_pyramid_level(I, 0, 3)[::2, ::2])


def test_features_basic_properties(oracle):
    I, d = _frame()
    n, keys, descs, levels = oracle.sift_run(I, d)
    assert 40 < n <= 1024 and levels.sum() == n
    assert levels[0] == 0 and levels[1] == 0            # sigma*scale < s_minKeyScale(3.0) on the two finest levels
    norms = np.sqrt((descs.astype(np.float64) ** 2).sum(1))
    assert np.all(np.abs(norms - 512) < 6)
    assert descs.max() <= 255 and (descs.max(axis=1) > 60).all()
    assert np.all(keys[:, 0] > 0) and np.all(keys[:, 0] < 640) and np.all(keys[:, 1] > 0) and np.all(keys[:, 1] < 480)
    xi = np.floor(keys[:, 0] + 0.5).astype(int); yi = np.floor(keys[:, 1] + 0.5).astype(int)   # roundf: half away from zero
    assert np.array_equal(keys[:, 3], d[yi, xi])        # key depth = depth at the rounded key position
    assert np.all(keys[:, 3] >= 0.1) and np.all(keys[:, 3] <= 4.0)
    # run twice: identical (no hidden state)
    n2, keys2, descs2, _ = oracle.sift_run(I, d)
    assert n2 == n and np.array_equal(keys, keys2) and np.array_equal(descs, descs2)


def test_invalid_depth_gates_detection(oracle):
    I, d = _frame()
    n_all, keys, _, _ = oracle.sift_run(I, d)
    d2 = d.copy(); d2[:, 320:] = -np.inf
    n_half, keys_half, _, _ = oracle.sift_run(I, d2)
    assert 0 < n_half and np.all(keys_half[:, 0] < 321)
    n_none, _, _, _ = oracle.sift_run(I, np.full_like(d, -np.inf))
    assert n_none == 0
    n_far, _, _, _ = oracle.sift_run(I, d, depth_max=0.2)
    assert n_far == 0


def test_shifted_image_gives_shifted_keypoints(oracle):
    I, d = _frame(120)
    n, keys, descs, _ = oracle.sift_run(I, d, feature_count_threshold=0)
    I2 = np.roll(I, (8, 16), axis=(0, 1)); d2 = np.roll(d, (8, 16), axis=(0, 1))
    n2, keys2, descs2, _ = oracle.sift_run(I2, d2, feature_count_threshold=0)
    inner = (keys[:, 0] > 80) & (keys[:, 0] < 540) & (keys[:, 1] > 80) & (keys[:, 1] < 380)
    set2 = {(float(a), float(b), float(s)) for a, b, s in keys2[:, :3]}
    hit = sum((float(a + 16), float(b + 8), float(s)) in set2 for a, b, s in keys[inner][:, :3])
    assert inner.sum() > 20 and hit >= 0.9 * inner.sum()      # 8 and 16 are multiples of every octave's stride


def test_too_many_keypoints_is_an_error(oracle):
    I, d = _frame()
    n, _, _, _ = oracle.sift_run(I, d, feature_count_threshold=0, max_features=16)
    assert n == -1                                            # Bundler.cpp:97 "too many keypoints"


def test_detmath_accuracy():
    """bf_detmath.h against numpy float64 (the functions are shared arithmetic, see the header)."""
    import ctypes as C, subprocess, os, tempfile
    src = r'''
    #include "bf_detmath.h"
    void dm_eval(const float* x, const float* y, int n, float* e, float* a, float* s, float* c, float* ac) {
        for (int i = 0; i < n; ++i) { e[i] = bf_dm_exp(-x[i]*x[i]); a[i] = bf_dm_atan2(y[i], x[i]); bf_dm_sincos(x[i], &s[i], &c[i]);
                                       float t = y[i]; if (t > 1) t = 1; if (t < -1) t = -1; ac[i] = bf_dm_acos(t); } }
    '''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "dm.c"), "w").write(src)
        so = os.path.join(td, "dm.so")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(root, "include"), os.path.join(td, "dm.c"), "-o", so])
        lib = C.CDLL(so)
        rng = np.random.default_rng(0)
        x = rng.uniform(-7, 7, 20000).astype(np.float32); y = rng.uniform(-1.2, 1.2, 20000).astype(np.float32)
        outs = [np.zeros(20000, np.float32) for _ in range(5)]
        lib.dm_eval(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 20000, *[o.ctypes.data_as(C.c_void_p) for o in outs])
        xd, yd = x.astype(np.float64), y.astype(np.float64)
        arg = (-(x * x)).astype(np.float64)          # the float32 argument the C code forms
        assert np.max(np.abs(outs[0] - np.exp(arg)) / np.exp(arg)) < 5e-7
        assert np.max(np.abs(outs[1] - np.arctan2(yd, xd))) < 1e-6
        assert np.max(np.abs(outs[2] - np.sin(xd))) < 2e-7 and np.max(np.abs(outs[3] - np.cos(xd))) < 2e-7
        assert np.max(np.abs(outs[4] - np.arccos(np.clip(yd, -1, 1)))) < 6e-7
