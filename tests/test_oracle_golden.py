"""Pins the oracle's track-stage threshold semantics on the literal vectors held by the
reference's own unit tests (numbers only, copied as data):
  Application/Tests/test_pixels.cpp:981-1071   LineWithoutGridTest2 (gray rows)
  Application/Tests/test_pixels.cpp:1611-1726  LineWithoutGridTest.AbsoluteDifferenceMethod
  Application/Tests/test_pixels.cpp:1750-1788  SignDifferenceMethod
  Application/Tests/test_pixels.cpp:1790-1821  NoneDifferenceMethod
  Application/Tests/test_pixels.cpp:1073-1166  BackgroundThresholding (gray leg)
HorizontalLine literals there are (y, x0, x1)."""
import numpy as np
from oracle import oracle

ABS, SIGN, NONE = 0, 1, 2


def R(*yxx):
    a = np.zeros(len(yxx), oracle.RUN_DTYPE)
    for i, (y, x0, x1) in enumerate(yxx):
        a[i] = (x0, x1, y, 0)
    return a


def lines(runs):
    return [(int(r["y"]), int(r["x0"]), int(r["x1"])) for r in runs]


FULL = R((0, 0, 9), (1, 0, 9))


def test_line_without_grid2_gray_bg100():
    bg = np.full((10, 10), 100, np.uint8)
    px = (np.arange(20) * 10).astype(np.uint8)
    r, p = oracle.line_without_grid(FULL, px, bg, ABS, 50)
    assert lines(r) == [(0, 0, 5), (1, 5, 9)]
    assert p.tolist() == [0, 10, 20, 30, 40, 50, 150, 160, 170, 180, 190]
    r, p = oracle.line_without_grid(FULL, px, bg, SIGN, 50)
    assert lines(r) == [(0, 0, 5)]
    assert p.tolist() == [0, 10, 20, 30, 40, 50]
    r, p = oracle.line_without_grid(FULL, px, bg, NONE, 50)
    assert lines(r) == [(0, 5, 9), (1, 0, 9)]
    assert p.tolist() == list(range(50, 200, 10))


def test_absolute_difference_method_bg150():
    bg = np.full((10, 10), 150, np.uint8)
    px = (np.arange(20) * 10).astype(np.uint8)
    r, p = oracle.line_without_grid(FULL, px, bg, ABS, 50)
    assert lines(r) == [(0, 0, 9), (1, 0, 0)]
    assert p.tolist() == list(range(0, 110, 10))
    r, p = oracle.line_without_grid(FULL, px, bg, NONE, 50)
    assert lines(r) == [(0, 5, 9), (1, 0, 9)]
    assert p.tolist() == list(range(50, 200, 10))
    r, p = oracle.line_without_grid(FULL, px, bg, SIGN, 50)
    assert lines(r) == [(0, 0, 9), (1, 0, 0)]
    assert p.tolist() == list(range(0, 110, 10))
    # second half of the reference test: runs are split where pixels fail
    thr = 50
    px = np.zeros(20, np.uint8)
    for i in range(20):
        if i % 3 == 0:
            px[i] = 125
        elif i < 10:
            px[i] = 100 - i * 5 - thr
        else:
            px[i] = (100 + i * 5 + thr) & 0xFF
    r, p = oracle.line_without_grid(FULL, px, bg, ABS, thr)
    assert lines(r) == [(0, 1, 2), (0, 4, 5), (0, 7, 8), (1, 0, 1), (1, 3, 4), (1, 6, 7), (1, 9, 9)]
    assert p.tolist() == [45, 40, 30, 25, 15, 10, 200, 205, 215, 220, 230, 235, 245]
    r, p = oracle.line_without_grid(FULL, px, bg, SIGN, thr)
    assert lines(r) == [(0, 1, 2), (0, 4, 5), (0, 7, 8)]
    assert p.tolist() == [45, 40, 30, 25, 15, 10]


def test_background_diff_values():
    # bg->diff<>(0,0,v) asserts, test_pixels.cpp:1642-1646 (bg = 150)
    bg = np.full((1, 1), 150, np.uint8)
    one = R((0, 0, 0))
    for method, v, expect in [(ABS, 200, 50), (NONE, 200, 200), (NONE, 55, 55), (SIGN, 100, 50), (SIGN, 200, 0)]:
        r, _ = oracle.line_without_grid(one, np.array([v], np.uint8), bg, method, expect)
        assert len(r) == 1, (method, v)           # diff >= expect passes
        r, _ = oracle.line_without_grid(one, np.array([v], np.uint8), bg, method, expect + 1)
        assert len(r) == 0, (method, v)           # diff <  expect+1 fails


def test_sign_difference_method():
    bg = np.full((10, 10), 150, np.uint8)
    px = np.array([(i + 1) if i % 2 == 0 else 200 for i in range(20)], np.uint8)
    r, p = oracle.line_without_grid(FULL, px, bg, SIGN, 50)
    assert lines(r) == [(0, x, x) for x in (0, 2, 4, 6, 8)] + [(1, x, x) for x in (0, 2, 4, 6, 8)]
    assert p.tolist() == [1, 3, 5, 7, 9, 11, 13, 15, 17, 19]


def test_none_difference_method():
    bg = np.full((10, 10), 150, np.uint8)
    px = np.arange(20, dtype=np.uint8)
    r, p = oracle.line_without_grid(FULL, px, bg, NONE, 5)
    assert lines(r) == [(0, 5, 9), (1, 0, 9)]
    assert p.tolist() == list(range(5, 20))


def _bgr2gray(b, g, r):
    # cmn::bgr2gray == cv::cvtColor(BGR2GRAY) fixed point (recalled): used only to turn the
    # reference's RGB literals into the gray literals its gray leg compares against
    return (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14


def test_background_thresholding_gray_leg():
    bgv = [[30, 50, 70, 90], [40, 60, 80, 100]]
    bg = np.array(bgv, np.uint8)
    blob = [(25, 25, 25), (110, 110, 110), (80, 80, 80), (10, 200, 10),
            (30, 30, 30), (95, 95, 95), (200, 200, 200), (100, 100, 100)]
    px = np.array([_bgr2gray(*v) for v in blob], np.uint8)
    full = R((0, 0, 3), (1, 0, 3))
    r, p = oracle.line_without_grid(full, px, bg, ABS, 25)
    assert lines(r) == [(0, 1, 1), (0, 3, 3), (1, 1, 2)]
    assert p.tolist() == [_bgr2gray(*v) for v in [(110, 110, 110), (10, 200, 10), (95, 95, 95), (200, 200, 200)]]


def test_threshold_blob_idempotent_at_zero():
    # TestLines.Threshold (test_matching.cpp:1556-1602): threshold_blob(0) returns the same lines
    rng = np.random.default_rng(3)
    bg = np.full((40, 40), 200, np.uint8)
    fr = bg.copy()
    fr[5:20, 7:25] = rng.integers(10, 90, (15, 18))
    fr[18:30, 20:33] = rng.integers(10, 90, (12, 13))     # overlapping rectangle => one blob
    p = oracle.make_params(40, 40, threshold=15)
    blobs, runs, pixels = oracle.segment(fr, bg, p)
    assert len(blobs) == 1
    b2, r2, p2 = oracle.threshold_blob(runs, pixels, bg, ABS, 0)
    assert len(b2) == 1
    assert r2.tobytes() == runs.tobytes() and p2.tobytes() == pixels.tobytes()
    assert b2.tobytes() == blobs.tobytes()
