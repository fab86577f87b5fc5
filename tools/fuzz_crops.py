"""Randomised bit-exact sweep of the normalised crops (moments, posture, legacy, caller transforms) against the oracle (dev tool).
   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_crops.py 40 [seed]'"""
import sys
import numpy as np
import torch
from oracle import oracle
from trex_amd import capi

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0; total = 0
for case in range(n_cases):
    H, W = 240, 512
    bg = rng.integers(150, 220, (H, W)).astype(np.uint8)
    fr = bg.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(int(rng.integers(3, 20))):
        cx, cy = rng.integers(20, W - 20), rng.integers(15, H - 15)
        a, b, th = rng.uniform(2, 90), rng.uniform(2, 20), rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1
        fr[m] = rng.integers(0, 110, int(m.sum()))
    difference = int(rng.integers(0, 3)); scale = float(rng.choice([1.0, 0.5, 0.8, 1.5])); ow, oh = (80, 80) if rng.random() < 0.7 else (int(rng.choice([32, 64, 128])), int(rng.choice([48, 80, 96])))
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1))
    seg.set_background(bg)
    d = torch.from_numpy(fr[None]).cuda()
    seg.segment_device(d.data_ptr(), 1)
    r = seg.fetch()[0]
    n = len(r.blobs)
    if n == 0:
        seg.close(); continue
    try:
        crops = torch.zeros((n, oh, ow), dtype=torch.uint8, device="cuda")
        seg.crops_device(crops.data_ptr(), n, out_w=ow, out_h=oh, normalization=1, difference=difference)
        seg.synchronize()
        got = crops.cpu().numpy()
        for k, b in enumerate(r.blobs):
            want, _ = oracle.crop_normalized(fr, bg, b, r.runs, out_w=ow, out_h=oh, difference=difference)
            assert np.array_equal(got[k], want), ("moments", k, int(b["n_pixels"]))
        MP = 1024
        outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
        info = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((n, 25, 4), dtype=torch.float32, device="cuda"); minfo = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
        seg.posture_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
        seg.midline_device(n, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
        legacy = bool(rng.integers(0, 2))
        seg.crops_posture_device(crops.data_ptr(), n, minfo.data_ptr(), out_w=ow, out_h=oh, legacy=legacy, scale=scale, difference=difference)
        seg.synchronize()
        got = crops.cpu().numpy(); mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
        for k, b in enumerate(r.blobs):
            if mi[k]["status"] != 0:
                assert got[k].sum() == 0
                continue
            tr = oracle.midline_transform(mi[k]["angle"], mi[k]["offx"], mi[k]["offy"], legacy)
            want, _ = oracle.crop_normalized(fr, bg, b, r.runs, tr6=tr, midline_length=float(mi[k]["len"]), legacy=legacy, out_w=ow, out_h=oh, scale=scale, difference=difference)
            assert np.array_equal(got[k], want), ("posture", k, int(b["n_pixels"]), legacy, scale)
            total += 1
    except AssertionError as e:
        fails += 1
        print("FAIL case", case, difference, scale, (ow, oh), str(e)[:200], flush=True)
    seg.close()
print("cases", n_cases, "posture crops compared", total, "failures", fails)
