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
