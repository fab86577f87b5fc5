"""Track-stage re-threshold on the device (second CCL pass) vs the oracle's whole-frame restatement of
Tracker::prefilter's threshold_blob (Tracker.cpp:765-912): bit-exact sub-blob tables incl. parent + size class."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi, synth

pytestmark = pytest.mark.gpu


def run(frames, bg, thr, method, ranges, detect_kw=None):
    detect_kw = detect_kw or {}
    n, H, W = frames.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=32768, **detect_kw))
    seg.set_background(bg)
    d = torch.from_numpy(frames).cuda()
    seg.segment_device(d.data_ptr(), n)
    det = seg.fetch()
    seg.rethreshold(thr, method, ranges)
    sub = seg.fetch(rethreshold=True)
    seg.close()
    return det, sub


@pytest.mark.parametrize("method,thr", [(0, 30), (1, 25), (1, 12), (2, 60), (0, 0)])
def test_bit_exact_vs_oracle(method, thr):
    rng = np.random.default_rng(method * 100 + thr)
    H, W = 120, 512
    bg = rng.integers(100, 180, (H, W)).astype(np.uint8)
    frames = []
    for t in range(3):
        fr = np.clip(bg.astype(int) + rng.integers(-8, 8, (H, W)), 0, 255).astype(np.uint8)
        for _ in range(25):                      # textured blobs: interior pixels straddle the track threshold
            y, x = rng.integers(0, H - 20), rng.integers(0, W - 40)
            fr[y:y + rng.integers(3, 20), x:x + rng.integers(3, 40)] = 0
            yy, xx = rng.integers(0, H - 12), rng.integers(0, W - 30)
            patch = np.clip(bg[yy:yy + 12, xx:xx + 30].astype(int) - rng.integers(0, 70, (12, 30)), 0, 255)
            fr[yy:yy + 12, xx:xx + 30] = patch
        frames.append(fr)
    frames = np.stack(frames)
    ranges = [(20, 200), (400, 900)]
    det, sub = run(frames, bg, thr, method, ranges)
    for f in range(len(frames)):
        ob, orr, opx = oracle.rethreshold_frame(frames[f], bg, oracle.make_params(W, H), method, thr, ranges)
        r = sub[f]
        assert len(r.blobs) == len(ob)
        assert r.runs.tobytes() == orr.tobytes() and r.pixels.tobytes() == opx.tobytes()
        want = ob.copy()
        want["parent"] = want["parent"] + det[f].info["blob_begin"]        # device parents are pooled indices
        for name in ob.dtype.names:
            assert np.array_equal(r.blobs[name], want[name]), name
        # recount of a parent = sum of its sub-blobs' pixels; never exceeds the detect blob
        for k, b in enumerate(det[f].blobs):
            tot = r.blobs["n_pixels"][r.blobs["parent"] == det[f].info["blob_begin"] + k].sum()
            assert tot <= b["n_pixels"]


def test_threshold_zero_is_identity_and_golden_semantics():
    # TestLines.Threshold (test_matching.cpp:1556-1602): threshold_blob(0) returns the same lines
    fr, bg = synth.batch("C2", 2)
    det, sub = run(fr, bg, 0, 0, [])
    for d, s in zip(det, sub):
        assert s.runs.tobytes() == d.runs.tobytes() and s.pixels.tobytes() == d.pixels.tobytes()
        assert np.array_equal(s.blobs["n_pixels"], d.blobs["n_pixels"]) and np.all(s.blobs["flags"] == 0)
    # literal vectors of test_pixels.cpp:981-1071 (bg 100, pixels i*10, threshold 50) through the device path
    bg = np.full((16, 16), 100, np.uint8)
    f = bg.copy()
    f[0, 0:10] = np.arange(10) * 10
    f[1, 0:10] = np.arange(10, 20) * 10
    det, sub = run(f[None], bg, 50, 0, [], detect_kw=dict(threshold=0, inclusive=1, zero_is_background=0, enable_difference=0))
    lines = [(int(r["y"]), int(r["x0"]), int(r["x1"])) for r in sub[0].runs]
    assert lines == [(0, 0, 5), (1, 5, 9)]
    assert sub[0].pixels.tolist() == [0, 10, 20, 30, 40, 50, 150, 160, 170, 180, 190]
    det, sub = run(f[None], bg, 50, 1, [], detect_kw=dict(threshold=0, inclusive=1, zero_is_background=0, enable_difference=0))
    assert [(int(r["y"]), int(r["x0"]), int(r["x1"])) for r in sub[0].runs] == [(0, 0, 5)]


def test_per_blob_thresholds():
    # SplitBlob::apply_threshold (SplitBlob.cpp:130-164) tries a different threshold on every merged blob
    fr, bg = synth.batch("C2", 1)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), 1)
    det = seg.fetch()[0]
    thr = np.array([(-1 if k % 5 == 4 else 20 + 3 * k) for k in range(len(det.blobs))], np.int32)
    dthr = torch.from_numpy(thr).cuda()
    seg.rethreshold_per_blob(dthr.data_ptr(), method=0)
    sub = seg.fetch(rethreshold=True)[0]
    for k, b in enumerate(det.blobs):
        mine = sub.blobs[sub.blobs["parent"] == det.info["blob_begin"] + k]
        if thr[k] < 0:
            assert len(mine) == 0
            continue
        rs = det.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
        px = det.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        ob, orr, opx = oracle.threshold_blob(rs, px, bg, 0, int(thr[k]))
        assert sorted(mine["n_pixels"].tolist()) == sorted(ob["n_pixels"].tolist())
        assert sorted(mine["bid"].tolist()) == sorted(ob["bid"].tolist())
    seg.close()
