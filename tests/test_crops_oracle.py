"""CPU checks of the normalised-crop oracle (FilterCache.cpp:21-115,276-288): the fixed-point warp against scipy's float
bilinear interpolation, and the transform composition against a direct numpy restatement."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import oracle


def _smooth(h, w, seed):
    rng = np.random.default_rng(seed)
    img = ndimage.gaussian_filter(rng.uniform(0, 255, (h, w)), 3.0)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


@pytest.mark.parametrize("M", [
    [1, 0, 3.25, 0, 1, -2.5],
    [0.8660254, -0.5, 20.0, 0.5, 0.8660254, -10.0],
    [1.2, 0.3, -4.0, -0.2, 0.9, 12.0],
])
def test_warp_matches_float_bilinear_within_quantisation(M):
    src = _smooth(90, 110, 1)
    M = np.asarray(M, np.float32)
    got = oracle.warp_affine(src, M, 80, 80)
    A = np.array([[M[0], M[1]], [M[3], M[4]]], np.float64)
    Ainv = np.linalg.inv(A)
    off = -Ainv @ np.array([M[2], M[5]], np.float64)
    # scipy works in (row, col): swap axes of the inverse map
    P = np.array([[Ainv[1, 1], Ainv[1, 0]], [Ainv[0, 1], Ainv[0, 0]]])
    want = ndimage.affine_transform(src.astype(np.float64), P, offset=[off[1], off[0]], output_shape=(80, 80), order=1, mode="constant", cval=0)
    # interior only: border pixels blend with the constant 0 differently at the 1/32 fraction grid
    yy, xx = np.mgrid[0:80, 0:80]
    sx = Ainv[0, 0] * xx + Ainv[0, 1] * yy + off[0]
    sy = Ainv[1, 0] * xx + Ainv[1, 1] * yy + off[1]
    inner = (sx > 1) & (sx < src.shape[1] - 2) & (sy > 1) & (sy < src.shape[0] - 2)
    assert inner.sum() > 1000
    d = np.abs(got.astype(np.float64) - want)[inner]
    assert d.max() <= 1.0 and d.mean() < 0.4


def test_identity_warp_is_a_copy_with_zero_border():
    src = _smooth(40, 50, 2)
    got = oracle.warp_affine(src, np.array([1, 0, 0, 0, 1, 0], np.float32), 80, 80)
    assert np.array_equal(got[:40, :50], src)
    assert got[40:].sum() == 0 and got[:, 50:].sum() == 0


def test_normalize_transform_composition():
    # t = translate(size/2) . scale(s) . translate(len*0.4 | (-len/2, 0)) . tr      (FilterCache.cpp:50-63)
    tr = np.array([0.6, -0.8, 5.0, 0.8, 0.6, -7.0], np.float32)

    def H(m):
        return np.array([[m[0], m[1], m[2]], [m[3], m[4], m[5]], [0, 0, 1]], np.float64)

    for legacy in (False, True):
        ln, s = 30.0, 0.5
        T1 = H([1, 0, 40, 0, 1, 40]); S = H([s, 0, 0, 0, s, 0])
        T2 = H([1, 0, -ln / 2, 0, 1, 0]) if legacy else H([1, 0, ln * 0.4, 0, 1, ln * 0.4])
        want = (T1 @ S @ T2 @ H(tr))[:2].reshape(-1)
        got = oracle.normalize_transform(tr, ln, legacy, 80, 80, s)
        assert np.allclose(got, want, atol=1e-4)


def test_nearest_warp_picks_source_pixels():
    # INTER_NEAREST (r3g3b2 crops): every output is a source pixel or the border value, identity is a copy
    src = _smooth(50, 60, 3)
    M = np.array([0.8, -0.6, 25.0, 0.6, 0.8, -5.0], np.float32)
    got = np.zeros((80, 80), np.uint8)
    oracle.lib().oracle_warp_affine_nearest_u8(oracle._ptr(np.ascontiguousarray(src)), 60, 50, oracle._ptr(M), oracle._ptr(got), 80, 80)
    A = np.array([[M[0], M[1]], [M[3], M[4]]], np.float64); Ainv = np.linalg.inv(A); off = -Ainv @ np.array([M[2], M[5]], np.float64)
    yy, xx = np.mgrid[0:80, 0:80]
    sx = np.rint(Ainv[0, 0] * xx + Ainv[0, 1] * yy + off[0]).astype(int); sy = np.rint(Ainv[1, 0] * xx + Ainv[1, 1] * yy + off[1]).astype(int)
    inside = (sx >= 0) & (sx < 60) & (sy >= 0) & (sy < 50)
    want = np.where(inside, src[np.clip(sy, 0, 49), np.clip(sx, 0, 59)], 0)
    assert (got != want).mean() < 0.02                      # only ties of the rounding at the 1/1024 grid may differ
    ident = np.zeros((50, 60), np.uint8)
    oracle.lib().oracle_warp_affine_nearest_u8(oracle._ptr(np.ascontiguousarray(src)), 60, 50, oracle._ptr(np.array([1, 0, 0, 0, 1, 0], np.float32)), oracle._ptr(ident), 60, 50)
    assert np.array_equal(ident, src)
