"""Posture oracle: Outline::resample pinned on the reference's own unit test vectors
(Application/Tests/test_outlines.cpp:53-95); outline tracing checked against an independent numpy edge count;
midline sanity on analytic shapes; golden anchor on the reference's test frames (midline_length column)."""
import os
import numpy as np
import pytest
from oracle import oracle


def test_resample_reference_vectors():
    sq = [(0, 0), (10, 0), (10, 10), (0, 10)]
    out = oracle.outline_resample(sq, 5.0)
    assert np.abs(out - np.array([(0, 0), (5, 0), (10, 0), (10, 5), (10, 10), (5, 10), (0, 10), (0, 5)], np.float32)).max() <= 0.01
    assert len(oracle.outline_resample(sq, 0.1)) > 100          # VerySmallResamplingDistance
    assert len(oracle.outline_resample(sq, 50.0)) < 3           # VeryLargeResamplingDistance
    one = oracle.outline_resample([(0, 0)], 5.0)                # SinglePointOutline
    assert one.tolist() == [[0, 0]]


def runs_of(mask):
    runs = []
    for y in range(mask.shape[0]):
        xs = np.flatnonzero(mask[y])
        if len(xs) == 0:
            continue
        brk = np.flatnonzero(np.diff(xs) > 1)
        starts = np.r_[xs[0], xs[brk + 1]]; ends = np.r_[xs[brk], xs[-1]]
        runs += [(s, e, y, 0) for s, e in zip(starts, ends)]
    return np.array(runs, oracle.RUN_DTYPE)


def boundary_edges(mask):
    m = np.pad(mask, 1)
    return int((m[1:-1, 1:-1] & ~m[:-2, 1:-1]).sum() + (m[1:-1, 1:-1] & ~m[2:, 1:-1]).sum()
               + (m[1:-1, 1:-1] & ~m[1:-1, :-2]).sum() + (m[1:-1, 1:-1] & ~m[1:-1, 2:]).sum())


def test_trace_outline_is_the_pixel_boundary():
    yy, xx = np.mgrid[0:40, 0:60]
    mask = ((xx - 30) / 22.0) ** 2 + ((yy - 20) / 9.0) ** 2 <= 1
    pts = oracle.trace_outline(runs_of(mask))
    assert len(pts) == 2 * boundary_edges(mask)                  # a corner and a midpoint per boundary edge (no holes)
    assert np.all(np.abs(pts * 2 - np.rint(pts * 2)) == 0)        # half-pixel lattice
    d = np.linalg.norm(np.roll(pts, -1, 0) - pts, axis=1)
    assert np.allclose(d, 0.5)                                    # closed, unit-speed walk
    area = 0.5 * np.sum(pts[:, 0] * np.roll(pts[:, 1], -1) - np.roll(pts[:, 0], -1) * pts[:, 1])
    assert area == mask.sum()                                     # clockwise in image coordinates, encloses exactly the pixels
    # single pixel and a diagonal pair (8-connected pinch)
    assert len(oracle.trace_outline(np.array([(5, 5, 3, 0)], oracle.RUN_DTYPE))) == 8
    diag = np.array([(5, 5, 3, 0), (6, 6, 4, 0)], oracle.RUN_DTYPE)
    assert len(oracle.trace_outline(diag)) == 16


def test_midline_of_an_ellipse_follows_its_long_axis():
    yy, xx = np.mgrid[0:60, 0:120]
    th = 0.3
    u = (xx - 60) * np.cos(th) + (yy - 30) * np.sin(th); v = -(xx - 60) * np.sin(th) + (yy - 30) * np.cos(th)
    mask = (u / 40.0) ** 2 + (v / 8.0) ** 2 <= 1
    info, outline, seg = oracle.posture(runs_of(mask), (0, 0))
    assert info["status"] == 0 and info["n_segments"] > 10
    pos = seg[:, :2]
    length = np.linalg.norm(np.diff(pos, axis=0), axis=1).sum()
    assert 60 < length < 85                                       # long axis is 80 px, ends are cut by the walk window
    # all midline points lie near the long axis
    dv = -(pos[:, 0] - 60) * np.sin(th) + (pos[:, 1] - 30) * np.cos(th)
    assert np.abs(dv).max() < 2.0
    # tail at outline[0]: one of the two tips; head index about half way round
    assert abs(info["head_index"] - len(outline) / 2) < 0.1 * len(outline)


def test_golden_midline_length_anchor():
    """midline_length column of videos/compare_data_automatic (Midline::len() of the normalised midline, about 39-40 px for
    these fish) vs the raw midline polyline of this restatement on the same blobs: same animal length: measured median ratio 1.006 (0.993 .. 1.04 over 48 fish-frames; the golden column is rounded to integers)."""
    from test_golden_e2e import FIX, rebuild, RANGES
    z = np.load(FIX)
    ratios = []
    for fr in [int(f) for f in z["frames"][:3]]:
        img, bg = rebuild(z, fr)
        p = oracle.make_params(img.shape[1], img.shape[0], threshold=9, size_ranges=[(1, 10000)])
        blobs, runs, px = oracle.rethreshold_frame(img, bg, p, 1, 12, RANGES)
        gold = {int(g[1]): float(g[4]) for g in z[f"gold/{fr}"]}
        for b in blobs:
            if int(b["bid"]) in gold and b["flags"] == 0 and np.isfinite(gold[int(b["bid"])]):
                rs = runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
                info, outline, seg = oracle.posture(rs, (int(b["x0"]), int(b["y0"])), oracle.posture_params(outline_resample=0.5))
                if info["status"] == 0:
                    ratios.append(np.linalg.norm(np.diff(seg[:, :2], axis=0), axis=1).sum() / gold[int(b["bid"])])
    assert len(ratios) >= 8
    assert 0.95 < np.median(ratios) < 1.06 and min(ratios) > 0.9 and max(ratios) < 1.12, (np.median(ratios), min(ratios), max(ratios))
