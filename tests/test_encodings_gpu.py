"""meta_encoding rgb8 / r3g3b2 through the device path: detection works on cv::cvtColor(BGR2GRAY) exactly like the gray encoding
(same lines, same blob tables), only the pixel arrays change: 3 bytes per pixel in memory order (rgb8) or the convert_to_r3g3b2
code (layout pinned by Tests/test_pixels.cpp:629-795).  The re-threshold pass keeps the colour bytes of the surviving pixels
(pv::Blob::threshold on an rgb8 blob, Tests/test_pixels.cpp:1289-1380)."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi

pytestmark = pytest.mark.gpu


def scene(seed, H=96, W=160, ch=3):
    rng = np.random.default_rng(seed)
    bgc = np.zeros((H, W, ch), np.uint8); bgc[...] = 150
    fr = bgc.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(12):
        cx, cy, a, b = rng.integers(5, W - 5), rng.integers(5, H - 5), rng.uniform(3, 20), rng.uniform(2, 8)
        m = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
        fr[m] = rng.integers(0, 255, (int(m.sum()), ch))
    if ch == 4:
        fr[..., 3] = 255
    return fr, bgc


def colour_pixels(fr, runs, enc):
    out = []
    for q in runs:
        seg = fr[int(q["y"]), int(q["x0"]):int(q["x1"]) + 1, :3]
        out.append(seg.reshape(-1) if enc == capi.ENC_RGB8 else oracle.convert_to_r3g3b2(seg))
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


@pytest.mark.parametrize("enc", [capi.ENC_RGB8, capi.ENC_R3G3B2])
@pytest.mark.parametrize("ch,device_input", [(3, False), (4, False), (4, True)])
def test_colour_pixel_arrays(enc, ch, device_input):
    frames = [scene(s, ch=ch) for s in (1, 2)]
    bgc = frames[0][1]
    W, H = bgc.shape[1], bgc.shape[0]
    gray_bg = oracle.bgr2gray(bgc[..., :3])
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=2, pixel_encoding=enc))
    seg.set_background(gray_bg)
    if device_input:
        d = torch.from_numpy(np.stack([f for f, _ in frames])).cuda()
        seg.segment_color_device(d.data_ptr(), 2, ch)
    else:
        seg.segment_color_host([f for f, _ in frames])
    res = seg.fetch()
    p = oracle.make_params(W, H)
    for (fr, _), r in zip(frames, res):
        ob, orr, opx = oracle.segment(oracle.bgr2gray(fr[..., :3]), gray_bg, p)
        assert r.runs.tobytes() == orr.tobytes()                      # same lines as the gray encoding
        for name in ob.dtype.names:
            assert np.array_equal(r.blobs[name], ob[name]), name       # pix_begin / n_pixels count pixels, moments use the grey value
        assert np.array_equal(r.pixels, colour_pixels(fr, r.runs, enc))
        assert len(r.pixels) == len(opx) * (3 if enc == capi.ENC_RGB8 else 1)
    # track stage: sub-blobs keep their colour bytes
    seg.rethreshold(40, 0, [])
    sub = seg.fetch(rethreshold=True)
    for (fr, _), r in zip(frames, sub):
        ob, orr, opx = oracle.rethreshold_frame(oracle.bgr2gray(fr[..., :3]), gray_bg, p, 0, 40, [])
        assert r.runs.tobytes() == orr.tobytes()
        assert np.array_equal(r.pixels, colour_pixels(fr, r.runs, enc))
    seg.close()


def test_rgb8_track_threshold_matches_the_reference_vectors():
    # BlobThresholding.RGB8AbsoluteDifferenceMultiRow (Tests/test_pixels.cpp:1289-1380) through detect + re-threshold on the device:
    # a 4x2 blob over the background 30..100, threshold 25 -> lines (0,1,1) (0,3,3) (1,1,2) with their colour bytes
    W, H = 16, 4
    bgv = np.array([[30, 50, 70, 90], [40, 60, 80, 100]], np.uint8)
    bgc = np.full((H, W, 3), 255, np.uint8)          # far from everything else: only the 4x2 patch is examined
    bgc[1:3, 4:8] = bgv[:, :, None]
    blob = np.array([(25, 25, 25), (110, 110, 110), (80, 80, 80), (10, 200, 10), (30, 30, 30), (95, 95, 95), (200, 200, 200), (90, 90, 90)], np.uint8).reshape(2, 4, 3)
    fr = bgc.copy(); fr[1:3, 4:8] = blob
    # detect everything of the patch as one blob: threshold 0 would take the whole frame, so use enable_difference = 0 semantics
    # via a background that differs everywhere inside the patch only -> use the track stage on a full-patch blob from threshold 1
    gray_bg = oracle.bgr2gray(bgc)
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, pixel_encoding=capi.ENC_RGB8, threshold=0, inclusive=1, zero_is_background=0))
    seg.set_background(gray_bg)
    seg.segment_color_host([fr])
    det = seg.fetch()[0]
    assert len(det.blobs) == 1 and det.blobs[0]["n_pixels"] == W * H       # threshold 0 inclusive: the whole frame is one blob
    seg.rethreshold(25, 0, [])
    sub = seg.fetch(rethreshold=True)[0]
    got = [(int(q["y"]), int(q["x0"]), int(q["x1"])) for q in sub.runs]
    assert got == [(1, 5, 5), (1, 7, 7), (2, 5, 6)]                         # the reference's lines, shifted by the patch origin (4, 1)
    assert sub.pixels.tolist() == [110, 110, 110, 10, 200, 10, 95, 95, 95, 200, 200, 200]
    seg.close()


def test_rgb8_crops_and_gray_api_guard():
    fr, bgc = scene(5, ch=3)
    W, H = bgc.shape[1], bgc.shape[0]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, pixel_encoding=capi.ENC_RGB8))
    seg.set_background(oracle.bgr2gray(bgc))
    with pytest.raises(capi.TrexHipError):
        seg.segment_host([oracle.bgr2gray(fr)])                              # colour encodings need colour input
    seg.segment_color_host([fr])
    r = seg.fetch()[0]
    n = len(r.blobs)
    crops = torch.full((n, 80, 80, 3), 7, dtype=torch.uint8, device="cuda")
    seg.crops_device(crops.data_ptr(), n)
    seg.synchronize()
    crops = crops.cpu().numpy()
    for k, b in enumerate(r.blobs):
        for c in range(3):                                                   # every channel = the un-normalised gray crop of that channel image
            want = oracle.crop_none(fr[..., c], np.zeros((H, W), np.uint8), b, r.runs)
            assert np.array_equal(crops[k, :, :, c], want), (k, c)
    # moments / posture normalisation: every channel is warped independently with the same transform and line mask
    d_crops = torch.zeros((n, 80, 80, 3), dtype=torch.uint8, device="cuda")
    seg.crops_device(d_crops.data_ptr(), n, normalization=1)
    seg.synchronize()
    got = d_crops.cpu().numpy()
    zero = np.zeros((H, W), np.uint8)
    for k, b in enumerate(r.blobs):
        for c in range(3):
            want, _ = oracle.crop_normalized(fr[..., c], zero, b, r.runs)
            assert np.array_equal(got[k, :, :, c], want), ("moments", k, c)
    MP = 512
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((n, 25, 4), dtype=torch.float32, device="cuda"); minfo = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
    seg.midline_device(n, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    seg.crops_posture_device(d_crops.data_ptr(), n, minfo.data_ptr())
    seg.synchronize()
    got = d_crops.cpu().numpy(); mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    n_ok = 0
    for k, b in enumerate(r.blobs):
        if mi[k]["status"] != 0:
            assert got[k].sum() == 0
            continue
        n_ok += 1
        tr = oracle.midline_transform(mi[k]["angle"], mi[k]["offx"], mi[k]["offy"], False)
        for c in range(3):
            want, _ = oracle.crop_normalized(fr[..., c], zero, b, r.runs, tr6=tr, midline_length=float(mi[k]["len"]))
            assert np.array_equal(got[k, :, :, c], want), ("posture", k, c)
    assert n_ok >= 3
    with pytest.raises(capi.TrexHipError):
        seg.crops_device(d_crops.data_ptr(), n, difference=1)                  # background-difference colour crops need the colour background
    seg.close()


def test_rgb8_difference_crops_need_and_use_the_colour_background():
    # Background(image, rgb8): per-channel differences (ImageFromLines.RGB8AbsoluteThresholdWithBackground, Tests/test_pixels.cpp:1381-1479)
    rng = np.random.default_rng(9)
    fr, bgc = scene(8, ch=3)
    bgc = np.clip(bgc.astype(int) + rng.integers(-40, 40, bgc.shape), 0, 255).astype(np.uint8)        # a textured colour background
    fr = np.where((fr == 150).all(axis=2, keepdims=True), bgc, fr)
    W, H = bgc.shape[1], bgc.shape[0]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, pixel_encoding=capi.ENC_RGB8))
    seg.set_background_color(bgc)
    assert np.array_equal(seg.get_background(), oracle.bgr2gray(bgc))          # detection works on its cvtColor image
    seg.segment_color_host([fr])
    r = seg.fetch()[0]
    n = len(r.blobs)
    assert n >= 5
    d_crops = torch.zeros((n, 80, 80, 3), dtype=torch.uint8, device="cuda")
    MP = 512
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((n, 25, 4), dtype=torch.float32, device="cuda"); minfo = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                                                   # the buffers were zeroed on torch's stream, the context has its own
    seg.posture_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
    seg.midline_device(n, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    seg.synchronize()                                                          # the context has its own stream
    mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    for diff in (1, 2):
        seg.crops_device(d_crops.data_ptr(), n, difference=diff)
        seg.synchronize()
        got = d_crops.cpu().numpy()
        for k, b in enumerate(r.blobs):
            for c in range(3):
                assert np.array_equal(got[k, :, :, c], oracle.crop_none(fr[..., c], bgc[..., c], b, r.runs, difference=diff)), ("none", diff, k, c)
        seg.crops_device(d_crops.data_ptr(), n, normalization=1, difference=diff)
        seg.synchronize()
        got = d_crops.cpu().numpy()
        for k, b in enumerate(r.blobs):
            for c in range(3):
                want, _ = oracle.crop_normalized(fr[..., c], bgc[..., c], b, r.runs, difference=diff)
                assert np.array_equal(got[k, :, :, c], want), ("moments", diff, k, c)
        seg.crops_posture_device(d_crops.data_ptr(), n, minfo.data_ptr(), difference=diff)
        seg.synchronize()
        got = d_crops.cpu().numpy()
        for k, b in enumerate(r.blobs):
            if mi[k]["status"] != 0:
                continue
            tr = oracle.midline_transform(mi[k]["angle"], mi[k]["offx"], mi[k]["offy"], False)
            for c in range(3):
                want, _ = oracle.crop_normalized(fr[..., c], bgc[..., c], b, r.runs, tr6=tr, midline_length=float(mi[k]["len"]), difference=diff)
                assert np.array_equal(got[k, :, :, c], want), ("posture", diff, k, c)
    seg.close()
    # the reference's own vector: 4x2 rgb8 blob on an rgb8 background, threshold 25, `differences` image
    blob = np.array([(25, 25, 25), (110, 110, 110), (80, 80, 80), (10, 200, 10), (30, 30, 30), (95, 95, 95), (200, 200, 200), (100, 100, 100)], np.uint8)
    frame = blob.reshape(2, 4, 3)
    bgv = np.array([[30, 50, 70, 90], [40, 60, 80, 100]], np.uint8)
    bg3 = np.repeat(bgv[:, :, None], 3, axis=2)
    seg = capi.Segmenter(capi.default_params(4, 2, max_batch=1, pixel_encoding=capi.ENC_RGB8, threshold=25, inclusive=1))
    seg.set_background_color(bg3)
    seg.segment_color_host([frame])
    r = seg.fetch()[0]
    assert len(r.blobs) == 1 and r.blobs[0]["n_pixels"] == 4                    # recount 4, one 8-connected blob
    crop = torch.zeros((1, 4, 4, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    seg.crops_device(crop.data_ptr(), 1, out_w=4, out_h=4, difference=1)
    seg.synchronize()
    got = crop.cpu().numpy()[0]
    # bounding box x 1..3, y 0..1 centred in 4x4: one column / row of padding on the left / top
    assert got[1, 1].tolist() == [60, 60, 60] and got[1, 3].tolist() == [80, 110, 80]
    assert got[2, 1].tolist() == [35, 35, 35] and got[2, 2].tolist() == [120, 120, 120]
    assert got.sum() == 180 + 270 + 105 + 360
    seg.close()


def test_rgb8_end_to_end_identity():
    # detect on colour tiles -> raw rgb8 crops on the device -> 3-channel V118_3: equal to the CPU network on the CPU-built crops
    from oracle import cnn_oracle
    from trex_amd import weights
    fr, bgc = scene(8, H=160, W=256, ch=4)
    W, H = bgc.shape[1], bgc.shape[0]
    st = weights.synthetic_state(12, 5, channels=3)
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, pixel_encoding=capi.ENC_RGB8))
    seg.set_background(oracle.bgr2gray(bgc[..., :3]))
    seg.load_weights(weights.pack_blob(st, 12, channels=3))
    seg.segment_color_host([fr])
    r = seg.fetch()[0]
    n = len(r.blobs)
    assert n >= 5
    crops = torch.zeros((n, 80, 80, 3), dtype=torch.uint8, device="cuda")
    probs = torch.zeros((n, 12), dtype=torch.float32, device="cuda")
    seg.crops_device(crops.data_ptr(), n)
    seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
    seg.synchronize()
    cpu_crops = np.stack([np.stack([oracle.crop_none(fr[..., c], np.zeros((H, W), np.uint8), b, r.runs) for c in range(3)], axis=-1) for b in r.blobs])
    assert np.array_equal(crops.cpu().numpy(), cpu_crops)
    op, _ = cnn_oracle.predict(st, cpu_crops, threads=8)
    assert np.abs(probs.cpu().numpy() - op).max() <= 1e-4
    seg.close()


def test_r3g3b2_crops_codes_and_nearest_warp():
    # r3g3b2: crops hold the colour codes; normalised crops are warped with nearest neighbour (FilterCache.cpp:70-73)
    fr, bgc = scene(9, ch=3)
    W, H = bgc.shape[1], bgc.shape[0]
    codes = oracle.convert_to_r3g3b2(fr)
    zero = np.zeros((H, W), np.uint8)
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, pixel_encoding=capi.ENC_R3G3B2))
    seg.set_background(oracle.bgr2gray(bgc))
    seg.segment_color_host([fr])
    r = seg.fetch()[0]
    n = len(r.blobs)
    assert n >= 5
    d = torch.full((n, 80, 80), 3, dtype=torch.uint8, device="cuda")
    seg.crops_device(d.data_ptr(), n)
    seg.synchronize()
    got = d.cpu().numpy()
    for k, b in enumerate(r.blobs):
        assert np.array_equal(got[k], oracle.crop_none(codes, zero, b, r.runs)), k
    seg.crops_device(d.data_ptr(), n, normalization=1)
    seg.synchronize()
    got = d.cpu().numpy()
    for k, b in enumerate(r.blobs):
        want, _ = oracle.crop_normalized(codes, zero, b, r.runs, nearest=True)
        assert np.array_equal(got[k], want), ("moments", k)
        assert set(np.unique(got[k])) <= set(np.unique(codes[int(b["y0"]):int(b["y1"]) + 1, int(b["x0"]):int(b["x1"]) + 1])) | {0}   # no interpolated codes
    MP = 512
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((n, 25, 4), dtype=torch.float32, device="cuda"); minfo = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
    seg.midline_device(n, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    seg.crops_posture_device(d.data_ptr(), n, minfo.data_ptr(), scale=0.8)
    seg.synchronize()
    got = d.cpu().numpy(); mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    for k, b in enumerate(r.blobs):
        if mi[k]["status"] != 0:
            continue
        tr = oracle.midline_transform(mi[k]["angle"], mi[k]["offx"], mi[k]["offy"], False)
        want, _ = oracle.crop_normalized(codes, zero, b, r.runs, tr6=tr, midline_length=float(mi[k]["len"]), scale=0.8, nearest=True)
        assert np.array_equal(got[k], want), ("posture", k)
    seg.close()
