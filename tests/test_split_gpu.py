"""SplitBlob's threshold search on the device (trexhip_split_search_device) vs the CPU restatement of
SplitBlob::split (tracking/SplitBlob.cpp:419-800): same threshold per merged blob, and the sub-blobs of
trexhip_rethreshold_per_blob_device at those thresholds = pixel::threshold_blob at them."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi
from split_cases import merged_scene

pytestmark = pytest.mark.gpu


def device_search(frames, bg, presumed_of, method=1, ranges=(), detect_threshold=15, detect_kw=None, **kw):
    n, H, W = frames.shape
    dkw = dict(threshold=detect_threshold)
    dkw.update(detect_kw or {})
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=4096, **dkw))
    seg.set_background(bg)
    d = torch.from_numpy(frames).cuda()
    seg.segment_device(d.data_ptr(), n)
    det = seg.fetch()
    nb = sum(len(r.blobs) for r in det)
    presumed = np.zeros(nb, np.int32)                        # pooled order: frame f's blobs start at info["blob_begin"]
    for r in det:
        presumed[r.info["blob_begin"]:r.info["blob_begin"] + len(r.blobs)] = presumed_of(r)
    d_pres = torch.from_numpy(presumed).cuda()
    d_thr = torch.full((max(nb, 1),), -7, dtype=torch.int32, device="cuda")
    d_info = torch.zeros(max(nb, 1) * capi.SPLIT_INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    seg.split_search_device(d_pres.data_ptr(), nb, d_thr.data_ptr(), d_info.data_ptr(), method=method, size_ranges=ranges, **kw)
    seg.rethreshold_per_blob(d_thr.data_ptr(), method, ranges)
    sub = seg.fetch(rethreshold=True)
    thr = d_thr.cpu().numpy()[:nb]
    info = d_info.cpu().numpy().view(capi.SPLIT_INFO_DTYPE)[:nb]
    seg.close()
    return det, presumed, thr, info, sub


@pytest.mark.parametrize("algorithm", [1, 2])
@pytest.mark.parametrize("method,ranges", [(1, [(40, 330)]), (0, []), (1, [(30, 120), (200, 330)])])
def test_search_equals_oracle(algorithm, method, ranges):
    frames, bgs = [], None
    for s in range(8):
        fr, bg, _ = merged_scene(40 + s + 10 * algorithm)
        frames.append(fr); bgs = bg
    frames = np.stack(frames)
    H, W = frames.shape[1:]
    rng = np.random.default_rng(5)
    det, presumed, thr, info, sub = device_search(frames, bgs, lambda r: rng.integers(0, 4, len(r.blobs)), method, ranges, algorithm=algorithm)
    sp = oracle.split_params(algorithm=algorithm, size_ranges=ranges)
    found = 0
    for f, r in enumerate(det):
        for j, b in enumerate(r.blobs):
            k = int(r.info["blob_begin"]) + j
            pr = int(presumed[k])
            if pr <= 0:
                assert thr[k] == -1 and info[k]["status"] == 3
                continue
            runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
            want = oracle.split_search(runs, px, bgs, method, sp, pr)
            got = info[k]
            assert got["status"] == 0
            assert (got["threshold"], got["effective_threshold"], got["initial_action"]) == (want.threshold, want.effective_threshold, want.initial_action), (f, k, pr)
            assert (got["min_pixel"], got["max_pixel"], got["n_result"]) == (want.min_pixel, want.max_pixel, want.n_result)
            assert got["first_size"] == want.first_size and got["min_size_bound"] == want.min_size_bound
            assert thr[k] == want.effective_threshold
            if want.threshold >= 0:
                found += 1
                # the sub-blobs of the per-blob re-threshold pass at that threshold are threshold_blob's
                ob, orr, opx = oracle.threshold_blob(runs, px, bgs, method, want.effective_threshold)
                mine = sub[f].blobs[sub[f].blobs["parent"] == k]
                assert sorted(mine["n_pixels"].tolist()) == sorted(ob["n_pixels"].tolist())
                assert sorted(mine["bid"].tolist()) == sorted(ob["bid"].tolist())
    assert found >= 3


def test_thresholds_are_minimal_and_really_split():
    fr, bg, groups = merged_scene(7, n_groups=8)
    det, presumed, thr, info, sub = device_search(fr[None], bg, lambda r: np.full(len(r.blobs), 2), 1, [(40, 330)])
    r = det[0]
    ok = 0
    for k, b in enumerate(r.blobs):
        if thr[k] < 0:
            continue
        ok += 1
        mine = sub[0].blobs[sub[0].blobs["parent"] == k]
        big = [n for n in mine["n_pixels"] if 40 <= n < 330]
        assert len(big) >= 2                                  # two individuals inside the size filter
        if info[k]["initial_action"] != 1:
            # one threshold lower the evaluation must not have been acceptable
            runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
            lower, _, _ = oracle.threshold_blob(runs, px, bg, 1, int(thr[k]) - 1)
            keep = sorted((n for n in lower["n_pixels"] if not n < info[k]["min_size_bound"]), reverse=True)[:2]
            assert len([n for n in keep if 40 <= n < 330]) < 2 or max(keep[:2]) >= 330
    assert ok >= 2


def test_capacity_and_errors():
    H, W = 256, 512
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy()
    fr[20:200, 20:400] = 40                                   # 68400 pixels: beyond the LDS capacity of one wave
    fr[220:230, 20:60] = 40
    det, presumed, thr, info, sub = device_search(fr[None], bg, lambda r: np.full(len(r.blobs), 2))
    assert info[0]["status"] == 2 and thr[0] == -1
    assert info[1]["status"] == 0
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1))
    with pytest.raises(capi.TrexHipError):
        seg.split_search_device(1, 1, 1, 1)                   # no batch yet
    for alg in (3, 4):                                        # blob_split_algorithm fill / fill_approximate (cv::watershed): refused, not ignored
        with pytest.raises(capi.TrexHipError) as e:
            seg.split_search_device(1, 1, 1, 1, algorithm=alg)
        assert e.value.code == -4
    seg.close()


def test_large_class_blobs():
    # blobs beyond the small size class (> 2048 pixels or > 256 lines) take the second launch: 3x upscaled scene
    fr, bg, _ = merged_scene(21, n_groups=5)
    fr3, bg3 = np.kron(fr, np.ones((3, 3), np.uint8)), np.kron(bg, np.ones((3, 3), np.uint8))
    ranges = [(360, 2970)]
    det, presumed, thr, info, sub = device_search(fr3[None], bg3, lambda r: np.full(len(r.blobs), 2), 1, ranges)
    sp = oracle.split_params(size_ranges=ranges)
    r = det[0]
    big = found = 0
    for k, b in enumerate(r.blobs):
        runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
        px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        want = oracle.split_search(runs, px, bg3, 1, sp, 2)
        assert info[k]["status"] == 0
        assert (info[k]["threshold"], info[k]["n_result"], info[k]["first_size"]) == (want.threshold, want.n_result, want.first_size)
        big += b["n_pixels"] > 2048
        found += want.threshold >= 0 and b["n_pixels"] > 2048
    assert big >= 3 and found >= 1


def test_huge_class_blobs():
    # two merged large animals (> 16384 pixels together): the third size class (one blob per CU, 134 KB of LDS), launched only when needed
    H, W = 400, 768
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    bg = (140 + ((np.arange(W)[None, :] * 3 + np.arange(H)[:, None] * 5) & 31) - 16).astype(np.uint8)
    depth = np.zeros((H, W), np.float32)
    for cx, cy, a, b, amp in [(380, 150, 200, 30, 95), (385, 203, 195, 29, 85), (120, 350, 40, 12, 80), (135, 368, 38, 11, 75)]:
        r2 = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2
        depth = np.maximum(depth, amp * np.clip(1.15 - r2, 0, 1))
    fr = np.clip(bg.astype(int) - np.rint(depth).astype(int), 0, 255).astype(np.uint8)
    ranges = [(2000, 30000), (100, 1700)]
    det, presumed, thr, info, sub = device_search(fr[None], bg, lambda r: np.full(len(r.blobs), 2), 1, ranges)
    r = det[0]
    sp = oracle.split_params(size_ranges=ranges)
    sizes = sorted(int(b["n_pixels"]) for b in r.blobs)
    assert len(sizes) == 2 and 16384 < sizes[1] <= 61440 and sizes[0] < 16384, sizes
    for k, b in enumerate(r.blobs):
        runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
        px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        want = oracle.split_search(runs, px, bg, 1, sp, 2)
        assert info[k]["status"] == 0
        assert (info[k]["threshold"], info[k]["n_result"], info[k]["first_size"], info[k]["min_pixel"], info[k]["max_pixel"]) == \
               (want.threshold, want.n_result, want.first_size, want.min_pixel, want.max_pixel), k
        assert want.threshold > 16                                  # both pairs separate above the track threshold
