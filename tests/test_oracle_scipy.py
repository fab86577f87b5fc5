"""Cross-checks the oracle's labelling / morphology against independent implementations
(scipy.ndimage) on random and adversarial images -- SURVEY.md section 7 step 1."""
import numpy as np
import pytest
from scipy import ndimage
from oracle import oracle
from trex_amd import synth


def check_against_scipy(fr, bg, p, conn):
    d = np.abs(fr.astype(int) - bg.astype(int)) > p.threshold
    d &= fr != 0
    st = np.ones((3, 3)) if conn == 8 else None
    lab, n = ndimage.label(d, structure=st)
    blobs, runs, pixels = oracle.segment(fr, bg, p)
    assert len(blobs) == n
    assert int(blobs["n_pixels"].sum()) == int(d.sum()) == len(pixels)
    # rebuild a label image from the oracle's runs and compare the partitions
    mine = np.zeros_like(lab)
    for k, b in enumerate(blobs):
        rs = runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
        # runs sorted by (y,x0), non-overlapping (pv.cpp:505-508)
        key = rs["y"].astype(np.int64) * 70000 + rs["x0"]
        assert np.all(np.diff(key) > 0)
        off = b["pix_begin"]
        for r in rs:
            n_ = int(r["x1"]) - int(r["x0"]) + 1
            mine[r["y"], r["x0"]:r["x1"] + 1] = k + 1
            assert np.array_equal(pixels[off:off + n_], fr[r["y"], r["x0"]:r["x1"] + 1])
            off += n_
        assert off - b["pix_begin"] == b["n_pixels"]
    assert np.array_equal(mine > 0, d)
    # same partition: the map scipy label -> my label is a bijection
    pairs = np.unique(np.stack([lab[d], mine[d]], 1), axis=0)
    assert len(pairs) == n
    # blob order = raster order of first pixel
    firsts = [(runs[b["run_begin"]]["y"], runs[b["run_begin"]]["x0"]) for b in blobs]
    assert firsts == sorted(firsts)
    # stats
    for k, b in enumerate(blobs[:50]):
        ys, xs = np.nonzero(mine == k + 1)
        assert b["m10"] == xs.sum() and b["m01"] == ys.sum()
        assert b["m20"] == (xs.astype(np.int64) ** 2).sum() and b["m02"] == (ys.astype(np.int64) ** 2).sum()
        assert b["m11"] == (xs.astype(np.int64) * ys).sum()
        v = fr[ys, xs].astype(np.int64)
        assert b["sp"] == v.sum() and b["spx"] == (v * xs).sum() and b["spy"] == (v * ys).sum()
        assert (b["x0"], b["x1"], b["y0"], b["y1"]) == (xs.min(), xs.max(), ys.min(), ys.max())
        assert b["px_min_max"] == int(v.min()) | (int(v.max()) << 8)


@pytest.mark.parametrize("conn", [8, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_scenes(seed, conn):
    rng = np.random.default_rng(seed)
    fr, bg = synth.random_scene(rng, 97 + 31 * seed, 64 + 9 * seed, density=0.15)
    p = oracle.make_params(fr.shape[1], fr.shape[0], connectivity=conn)
    check_against_scipy(fr, bg, p, conn)


def test_adversarial_patterns():
    H, W = 48, 80
    bg = np.full((H, W), 200, np.uint8)
    p = oracle.make_params(W, H)
    # checkerboard: one blob with 8-connectivity
    fr = bg.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    fr[(yy + xx) % 2 == 0] = 10
    check_against_scipy(fr, bg, p, 8)
    p4 = oracle.make_params(W, H, connectivity=4)
    check_against_scipy(fr, bg, p4, 4)
    # 1-px diagonals
    fr = bg.copy()
    for i in range(40):
        fr[i, i] = 20
        fr[i, 79 - i] = 30
    check_against_scipy(fr, bg, p, 8)
    # spiral touching borders
    fr = bg.copy()
    fr[0, :] = 5; fr[:, W - 1] = 5; fr[H - 1, 2:] = 5; fr[2:, 2] = 5; fr[2, 2:W - 2] = 5
    check_against_scipy(fr, bg, p, 8)
    # U shapes that merge late (label equivalence)
    fr = bg.copy()
    for k in range(6):
        fr[4:30, 4 + 10 * k] = 9
        fr[4:30, 8 + 10 * k] = 9
        fr[30, 4 + 10 * k:9 + 10 * k] = 9
    fr[40, 4:70] = 9
    fr[30:41, 6] = 9
    check_against_scipy(fr, bg, p, 8)
    # empty frame
    blobs, runs, pixels = oracle.segment(bg, bg, p)
    assert len(blobs) == 0 and len(runs) == 0 and len(pixels) == 0
    # full frame
    fr = np.full_like(bg, 3)
    check_against_scipy(fr, bg, p, 8)


def test_threshold_variants_and_zero_pixels():
    H, W = 16, 32
    bg = np.full((H, W), 100, np.uint8)
    fr = bg.copy()
    fr[3, 3:9] = [85, 84, 0, 116, 115, 130]      # diffs 15,16,100,16,15,30
    p = oracle.make_params(W, H, threshold=15, inclusive=0)
    b, r, px = oracle.segment(fr, bg, p)          # strict >, zero grey is background
    assert [(int(x["x0"]), int(x["x1"])) for x in r] == [(4, 4), (6, 6), (8, 8)]
    p = oracle.make_params(W, H, threshold=15)    # the default: |p| < threshold is disregarded (core/default_config.cpp:1168), i.e. >=
    assert p.inclusive == 1
    b, r, px = oracle.segment(fr, bg, p)
    assert [(int(x["x0"]), int(x["x1"])) for x in r] == [(3, 4), (6, 8)]
    p = oracle.make_params(W, H, threshold=15, inclusive=1, zero_is_background=0)
    b, r, px = oracle.segment(fr, bg, p)
    assert [(int(x["x0"]), int(x["x1"])) for x in r] == [(3, 8)]
    assert px.tolist() == [85, 84, 0, 116, 115, 130]
    p = oracle.make_params(W, H, threshold=15, absolute_difference=0, inclusive=0)   # sign: bg - px
    b, r, px = oracle.segment(fr, bg, p)
    assert [(int(x["x0"]), int(x["x1"])) for x in r] == [(4, 4)]
    p = oracle.make_params(W, H, threshold=15, threshold_maximum=29)    # inRange [15,29]
    b, r, px = oracle.segment(fr, bg, p)
    assert [(int(x["x0"]), int(x["x1"])) for x in r] == [(3, 4), (6, 7)]


def test_size_filter_and_bid():
    H, W = 32, 64
    bg = np.full((H, W), 100, np.uint8)
    fr = bg.copy()
    fr[2, 2:5] = 10            # 3 px
    fr[10:14, 10:20] = 10      # 40 px
    fr[20:30, 30:60] = 10      # 300 px
    p = oracle.make_params(W, H, size_ranges=[(4, 300)])     # half-open => 300 rejected
    b, r, px = oracle.segment(fr, bg, p)
    assert b["n_pixels"].tolist() == [40]
    assert b["bid"][0] == ((10 + (19 - 10 + 1) // 2) << 19 | 10 << 6 | 4)
    assert oracle.bid(637, 639, 1995, 41) == 334623465      # videos/compare_data_automatic/test_fish0.csv:2


@pytest.mark.parametrize("kw", [dict(use_closing=1, closing_size=3), dict(dilation_size=1), dict(dilation_size=-1),
                                 dict(use_closing=1, closing_size=5, dilation_size=2)])
def test_morphology_against_scipy(kw):
    rng = np.random.default_rng(5)
    fr, bg = synth.random_scene(rng, 90, 70, density=0.2)
    p = oracle.make_params(90, 70, **kw)
    out = oracle.generate_binary(fr, bg, p) != 0
    m = np.abs(fr.astype(int) - bg.astype(int)) > 15

    def ellipse(k):
        r = c = k // 2
        e = np.zeros((k, k), bool)
        for i in range(k):
            dy = i - r
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / (r * r)))) if r else 0
            e[i, max(c - dx, 0):min(c + dx + 1, k)] = True
        return e
    if kw.get("use_closing"):
        e = ellipse(kw["closing_size"])
        m = ndimage.binary_dilation(m, e)
        m = ndimage.binary_erosion(m, e, border_value=1)
    d = kw.get("dilation_size", 0)
    if d > 0:
        m = ndimage.binary_dilation(m, ellipse(2 * d + 1))
    elif d < 0:
        m = ndimage.binary_erosion(m, ellipse(2 * -d + 1), border_value=1)
    assert np.array_equal(out, m & (fr != 0))
