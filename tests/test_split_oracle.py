"""SplitBlob threshold search (tracking/SplitBlob.cpp:130-255,419-800): the C restatement vs an independent numpy / scipy
evaluation of every threshold (complete search = smallest threshold whose evaluation is KEEP_ABORT, no ABORT before it)."""
import numpy as np
import pytest
from scipy import ndimage
from oracle import oracle
from split_cases import merged_scene


def brute_force(runs, pixels, bg, method, sp, presumed):
    """evaluate_result_multiple for all thresholds with scipy's labelling; returns the expected best threshold or -1"""
    x0, y0 = int(runs["x0"].min()), int(runs["y"].min())
    w, h = int(runs["x1"].max()) - x0 + 1, int(runs["y"].max()) - y0 + 1
    d = np.full((h, w), -1, int)
    o = 0
    for r in runs:
        n = int(r["x1"]) - int(r["x0"]) + 1
        p = pixels[o:o + n].astype(int)
        b = bg[int(r["y"]), int(r["x0"]):int(r["x1"]) + 1].astype(int)
        d[int(r["y"]) - y0, int(r["x0"]) - x0:int(r["x1"]) - x0 + 1] = np.abs(b - p) if method == 0 else (np.maximum(b - p, 0) if method == 1 else p)
        o += n
    vals = d[d >= 0]
    min_px, max_px = min(254, int(vals.min())), max(0, int(vals.max()))
    sq = np.float32(sp.cm_per_pixel) * np.float32(sp.cm_per_pixel)
    ranges = [(sp.ranges[2 * i], sp.ranges[2 * i + 1]) for i in range(sp.n_ranges)]

    def evaluate(t, first_size):
        lab, n = ndimage.label(d >= t, structure=np.ones((3, 3)))
        sizes = sorted((int(s) for s in ndimage.sum(d >= t, lab, range(1, n + 1))), reverse=True) if n else []
        max_size = np.float32(sizes[0] if sizes else 0) * sq
        pixels_ = sum(sizes)
        if np.float32(pixels_) * sq < np.float32(sp.blob_split_max_shrink) * np.float32(first_size):
            return "ABORT", max_size
        if ranges:
            bound = min(a for a, _ in ranges) * float(np.float32(sp.blob_split_global_shrink_limit))
        else:
            bound = float(np.float32(pixels_) * sq * np.float32(sp.blob_split_max_shrink))
        sizes = [s for s in sizes if not float(np.float32(s) * sq) < bound]
        top = sizes[:presumed]
        valid = sum(1 for s in top if (not ranges) or any(a <= float(np.float32(s) * sq) < b for a, b in ranges))
        if ranges and top and float(np.float32(min(top)) * sq) > max(b for _, b in ranges):
            return "REMOVE", max_size
        if valid < presumed:
            return "TOO_FEW", max_size
        return "KEEP_ABORT", max_size

    begin = max(sp.initial_threshold, min_px)
    a, first = evaluate(begin, 0.0)
    if a == "KEEP_ABORT":
        return sp.initial_threshold, a
    if ranges and not float(np.float32(len(vals)) * sq) < max(b for _, b in ranges) * 100:
        return -1, a
    if presumed <= 1:
        return -1, a
    for t in range(begin, max_px):
        act, _ = evaluate(t, first)
        if act == "KEEP_ABORT":
            return t, a
        if act == "ABORT":
            break
    return -1, a


@pytest.mark.parametrize("seed", range(6))
def test_complete_search_equals_brute_force(seed):
    frame, bg, groups = merged_scene(seed)
    H, W = frame.shape
    blobs, runs, px = oracle.segment(frame, bg, oracle.make_params(W, H, threshold=15))
    assert len(blobs) >= 3
    ranges = [(40, 330)] if seed % 3 else []
    sp = oracle.split_params(track_threshold=15, track_posture_threshold=15, size_ranges=ranges)
    found = 0
    for b in blobs:
        r = runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
        p = px[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        for presumed in (2, 3):
            info = oracle.split_search(r, p, bg, 1, sp, presumed)
            want, a0 = brute_force(r, p, bg, 1, sp, presumed)
            assert info.threshold == want, (seed, int(b["bid"]), presumed, info.threshold, want)
            assert oracle.SPLIT_ACTIONS[info.initial_action] == a0
            if info.threshold >= 0:
                found += 1
                sub, _, _ = oracle.threshold_blob(r, p, bg, 1, info.effective_threshold)
                kept = [s for s in sub["n_pixels"] if not float(np.float32(s)) < info.min_size_bound]
                assert len(kept) == info.n_result >= presumed
    assert found > 0


def test_approximate_never_beats_complete_and_agrees_mostly():
    hits = same = 0
    for seed in range(6):
        frame, bg, _ = merged_scene(100 + seed)
        H, W = frame.shape
        blobs, runs, px = oracle.segment(frame, bg, oracle.make_params(W, H, threshold=15))
        for b in blobs:
            r = runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            p = px[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
            full = oracle.split_search(r, p, bg, 1, oracle.split_params(algorithm=1, size_ranges=[(40, 330)]), 2)
            appr = oracle.split_search(r, p, bg, 1, oracle.split_params(algorithm=2, size_ranges=[(40, 330)]), 2)
            if appr.threshold >= 0:
                assert full.threshold >= 0 and full.threshold <= appr.threshold      # the complete search finds the smallest
                hits += 1
                same += full.threshold == appr.threshold
            assert appr.n_tried <= max(full.n_tried, 1) + 8
    assert hits > 0 and same > 0


def test_algorithm_none_and_single_individual():
    frame, bg, _ = merged_scene(3)
    H, W = frame.shape
    blobs, runs, px = oracle.segment(frame, bg, oracle.make_params(W, H, threshold=15))
    b = blobs[int(np.argmax(blobs["n_pixels"]))]
    r = runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
    p = px[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
    assert oracle.split_search(r, p, bg, 1, oracle.split_params(algorithm=0), 2).threshold == -1
    one = oracle.split_search(r, p, bg, 1, oracle.split_params(), 1)       # presumed_nr 1: the initial threshold already keeps it
    assert one.threshold == oracle.split_params().initial_threshold and one.n_tried == 1
