"""Device crop gather (individual_image_normalization=none) vs the oracle's restatement of
calculate_diff_image (FilterCache.cpp:157-235): bit-exact, including blobs larger than the crop."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("difference", [0, 1, 2])
def test_crops_none_bit_exact(difference):
    fr, bg = synth.batch("C2", 2)
    fr = fr.copy()
    fr[0, 100:230, 300:420] = 10           # a blob larger than 80x80 in both directions
    fr[1, 500:520, 100:300] = 20           # wider than 80, shorter than 80
    fr[1, 0:7, 0:5] = 30                   # touches the border, odd sizes
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    total = sum(len(r.blobs) for r in res)
    crops = torch.full((total, 80, 80), 77, dtype=torch.uint8, device="cuda")
    seg.crops_device(crops.data_ptr(), total, difference=difference)
    seg.synchronize()
    crops = crops.cpu().numpy()
    for r, f in zip(res, fr):
        for k, b in enumerate(r.blobs):
            want = oracle.crop_none(f, bg, b, r.runs, difference=difference)
            got = crops[int(r.info["blob_begin"]) + k]
            assert np.array_equal(got, want), (k, int(b["n_pixels"]))
    seg.close()


def _segment(fr, bg):
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    return seg, res, d


@pytest.mark.parametrize("difference", [0, 1])
def test_crops_moments_bit_exact(difference):
    # individual_image_normalization = moments (FilterCache.cpp:276-288): orientation from the blob's moments, warpAffine
    fr, bg = synth.batch("C2", 2)
    fr = fr.copy()
    fr[0, 100:230, 300:420] = 10           # larger than the crop (bounding box still painted into LDS)
    fr[1, 300:480, 500:650] = np.arange(150, dtype=np.uint8)[None, :] % 90      # bounding box beyond the LDS image: per-tap line tests
    seg, res, d = _segment(fr, bg)
    total = sum(len(r.blobs) for r in res)
    crops = torch.full((total, 80, 80), 77, dtype=torch.uint8, device="cuda")
    seg.crops_device(crops.data_ptr(), total, normalization=1, difference=difference)
    seg.synchronize()
    crops = crops.cpu().numpy()
    for r, f in zip(res, fr):
        for k, b in enumerate(r.blobs):
            want, _ = oracle.crop_normalized(f, bg, b, r.runs, difference=difference)
            got = crops[int(r.info["blob_begin"]) + k]
            assert np.array_equal(got, want), (k, int(b["n_pixels"]))
            assert got.sum() > 0
    seg.close()


@pytest.mark.parametrize("legacy", [False, True])
def test_crops_with_supplied_transforms(legacy):
    # posture / legacy: Midline::transform(...) comes from the caller (Outline.cpp:1237-1255)
    fr, bg = synth.batch("C2", 1)
    seg, res, d = _segment(fr, bg)
    r = res[0]
    total = len(r.blobs)
    rng = np.random.default_rng(4)
    tr = np.zeros((total, 6), np.float32); ln = rng.uniform(20, 40, total).astype(np.float32)
    for k, b in enumerate(r.blobs):
        a = rng.uniform(0, 2 * np.pi); cs, sn = np.cos(a), np.sin(a)
        fx, fy = rng.uniform(5, 30, 2)
        # tr = translate(-front) . rotate(angle) . translate(-offset) in the SFML convention, built here as one matrix
        tr[k] = [cs, -sn, -fx * cs + fy * sn - 3.0, sn, cs, -fx * sn - fy * cs + 2.0]
    crops = torch.zeros((total, 80, 80), dtype=torch.uint8, device="cuda")
    seg.crops_transformed_device(crops.data_ptr(), tr, ln, legacy=legacy)
    seg.synchronize()
    crops = crops.cpu().numpy()
    for k, b in enumerate(r.blobs):
        want, _ = oracle.crop_normalized(fr[0], bg, b, r.runs, tr6=tr[k], midline_length=float(ln[k]), legacy=legacy)
        assert np.array_equal(crops[int(r.info["blob_begin"]) + k], want), k
    seg.close()
