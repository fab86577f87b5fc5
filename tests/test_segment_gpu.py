"""Parity of the HIP detect stage (through the C ABI) against the CPU oracle: bit-exact blob
tables, runs and pixels.  Mirrors the shape of the reference's own tests (TestLines.Threshold
test_matching.cpp:1556-1602: overlapping shapes = 1 blob; test_segmenter.cpp:95-126 moving
square) plus adversarial images."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_gpu(frames, bg, device_resident=True, **kw):
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    kw.setdefault("max_blobs", 32768)
    p = capi.default_params(W, H, max_batch=max(n, 1), **kw)
    seg = capi.Segmenter(p)
    seg.set_background(bg)
    if device_resident:
        d = torch.from_numpy(frames).cuda()
        seg.segment_device(d.data_ptr(), n)
        res = seg.fetch()
        del d
    else:
        seg.segment_host([f for f in frames])
        res = seg.fetch()
    seg.close()
    return res


def oracle_params(W, H, **kw):
    kw = dict(kw)
    kw.pop("max_runs", None); kw.pop("max_blobs", None); kw.pop("max_pixels", None)
    return oracle.make_params(W, H, **kw)


def assert_frame_equal(res, frame, bg, **kw):
    H, W = frame.shape
    ob, orr, opx = oracle.segment(frame, bg, oracle_params(W, H, **kw))
    assert res.info["flags"] == 0
    assert len(res.blobs) == len(ob), (len(res.blobs), len(ob))
    assert res.runs.tobytes() == orr.tobytes()
    assert res.pixels.tobytes() == opx.tobytes()
    for name in ob.dtype.names:
        assert np.array_equal(res.blobs[name], ob[name]), name
    assert res.blobs.tobytes() == ob.tobytes()


@pytest.mark.parametrize("shape", [(64, 48), (1280, 720), (1024, 33), (2304, 100), (4096, 16), (16, 16)])
def test_random_scenes_bit_exact(shape):
    W, H = shape
    rng = np.random.default_rng(W * 7 + H)
    frames = []
    for i in range(3):
        fr, bg = synth.random_scene(rng, W, H, density=0.1)
        frames.append(fr)
    res = run_gpu(np.stack(frames), bg)
    for r, fr in zip(res, frames):
        assert_frame_equal(r, fr, bg)


@pytest.mark.parametrize("W", [17, 100, 1001, 1030, 2050])
def test_unaligned_widths(W):
    rng = np.random.default_rng(W)
    fr, bg = synth.random_scene(rng, W, 40, density=0.15)
    fr[:, W - 1] = 5       # foreground in the last column
    fr[7, :] = 3           # a run spanning the full row
    res = run_gpu(fr[None], bg)
    assert_frame_equal(res[0], fr, bg)


def test_reference_style_cases():
    # moving 8x8 white square on black, 64x48 (test_segmenter.cpp:95-126)
    bg = np.zeros((48, 64), np.uint8)
    frames = []
    for t in range(12):
        f = bg.copy(); f[20:28, 3 * t:3 * t + 8] = 255; frames.append(f)
    res = run_gpu(np.stack(frames), bg)
    for t, r in enumerate(res):
        assert len(r.blobs) == 1 and r.blobs["n_pixels"][0] == 64
        assert (r.blobs["x0"][0], r.blobs["y0"][0]) == (3 * t, 20)
        assert_frame_equal(r, frames[t], bg)
    # overlapping circle + rectangle = one blob (test_matching.cpp:1556-1602)
    H = W = 320
    yy, xx = np.mgrid[0:H, 0:W]
    bg = np.full((H, W), 255, np.uint8)
    fr = bg.copy()
    fr[(yy - 100) ** 2 + (xx - 100) ** 2 <= 50 ** 2] = 0 + 1
    fr[90:200, 120:260] = 2
    res = run_gpu(fr[None], bg)
    assert len(res[0].blobs) == 1
    assert_frame_equal(res[0], fr, bg)


def test_adversarial_patterns():
    H, W = 96, 2048
    bg = np.full((H, W), 200, np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    f = bg.copy(); f[(yy + xx) % 2 == 0] = 10; frames.append(f)                 # checkerboard
    f = bg.copy(); f[yy == xx % H] = 20; frames.append(f)                         # diagonals
    f = bg.copy(); f[0, :] = 5; f[:, W - 1] = 5; f[H - 1, 2:] = 5; f[2:, 2] = 5; f[2, 2:W - 2] = 5; frames.append(f)
    f = bg.copy(); f[:] = 3; frames.append(f)                                     # everything foreground
    frames.append(bg.copy())                                                      # empty
    f = bg.copy(); f[:, 1023] = 9; f[:, 1024] = 9; f[5, 1000:1100] = 9; frames.append(f)   # chunk boundary
    f = bg.copy(); f[xx % 3 == 0] = 9; frames.append(f)                           # many 1-px runs per row
    for conn in (8, 4):
        res = run_gpu(np.stack(frames), bg, connectivity=conn, max_runs=200000, max_blobs=131072)   # the 4-connected checkerboard alone is 98304 blobs (capacities are per frame)
        for r, fr in zip(res, frames):
            assert_frame_equal(r, fr, bg, connectivity=conn)
    assert len(res[4].blobs) == 0 and res[4].info["n_raw_runs"] == 0


@pytest.mark.parametrize("kw", [
    dict(threshold=15, inclusive=1), dict(threshold=15, inclusive=0), dict(threshold=0, inclusive=1), dict(threshold=40, absolute_difference=0),
    dict(threshold=20, threshold_maximum=60), dict(threshold=15, zero_is_background=0, inclusive=1),
    dict(threshold=30, image_invert=1), dict(threshold=100, enable_difference=0),
    dict(threshold=15, size_ranges=[(3, 9), (20, 1000)], cm_per_pixel=0.5),
])
def test_setting_variants(kw):
    rng = np.random.default_rng(11)
    W, H = 512, 64
    bg = rng.integers(60, 200, (H, W)).astype(np.uint8)
    fr = np.clip(bg.astype(int) + rng.integers(-70, 70, (H, W)), 0, 255).astype(np.uint8)
    fr[rng.random((H, W)) < 0.02] = 0
    res = run_gpu(fr[None], bg, **kw)
    assert_frame_equal(res[0], fr, bg, **kw)


def test_synthetic_configs_and_host_path():
    fr, bg = synth.batch("C2", 3)
    res = run_gpu(fr, bg, device_resident=False)
    for r, f in zip(res, fr):
        assert_frame_equal(r, f, bg)
        assert len(r.blobs) == 32
    fr, bg = synth.batch("C3", 2)
    res = run_gpu(fr, bg)
    for r, f in zip(res, fr):
        assert_frame_equal(r, f, bg)
        assert len(r.blobs) == 100


def test_full_size_properties_c5():
    # 4096x4096 / 256 individuals: size-independent properties + oracle on one frame
    fr, bg = synth.batch("C5", 2)
    res = run_gpu(fr, bg)
    for r, f in zip(res, fr):
        d = np.abs(f.astype(np.int16) - bg.astype(np.int16)) > 15
        assert int(r.blobs["n_pixels"].sum()) == int(d.sum())
        keys = r.runs["y"].astype(np.int64) << 16 | r.runs["x0"]
        for b in r.blobs:                       # runs of a blob sorted by (y,x0)  (pv.cpp:505-508)
            k = keys[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            assert np.all(np.diff(k) > 0)
        assert len(r.blobs) == 256
    assert_frame_equal(res[0], fr[0], bg)
    # idempotence: segmenting the grey-under-mask image against a zero background gives the same lines
    out = oracle.generate_binary(fr[0], bg, oracle.make_params(4096, 4096))
    res2 = run_gpu(out[None], np.zeros_like(bg), threshold=0)
    assert res2[0].runs.tobytes() == res[0].runs.tobytes()


def test_capacity_overflow_is_reported():
    H, W = 64, 256
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy(); fr[:, ::2] = 5             # 128 runs per row = 8192 runs
    res = run_gpu(np.stack([fr, bg]), bg, max_runs=1000)
    assert res[0].info["flags"] & 1 and len(res[0].blobs) == 0
    assert res[1].info["flags"] == 0 and len(res[1].blobs) == 0
    res = run_gpu(fr[None], bg, max_runs=10000, max_blobs=100)
    assert res[0].info["flags"] & 2


def test_noise_frame_in_the_last_slot_overflows_gracefully():
    # the last frame of a full batch carries millions of runs against a small run area: its rows' offsets point far past the end of the
    # batch's run buffer (only the writes are bounded); the labelling kernel must refuse the frame without reading there, and the
    # context's overflow counters must be clean for the next pass
    W, H, n = 2048, 1024, 4
    rng = np.random.default_rng(5)
    frs = []
    for i in range(n - 1):
        fr, bg = synth.random_scene(np.random.default_rng(77), W, H, density=0.02)
        frs.append(fr)
    noise = bg.copy(); noise[:, ::2] = np.where(bg[:, ::2] > 127, 0, 255)          # ~1 M one-pixel runs
    frs.append(noise)
    p = capi.default_params(W, H, max_batch=n, max_runs=4096, max_blobs=4096)
    seg = capi.Segmenter(p)
    seg.set_background(bg)
    d = torch.from_numpy(np.stack(frs)).cuda()
    for rep in range(3):
        seg.segment_device(d.data_ptr(), n)
        res = seg.fetch()
        assert res[n - 1].info["flags"] & 1 and len(res[n - 1].blobs) == 0
        for r, fr in zip(res[:n - 1], frs[:n - 1]):
            if r.info["flags"] == 0:
                assert_frame_equal(r, fr, bg)
    d2 = torch.from_numpy(np.stack([bg] * n)).cuda()
    seg.segment_device(d2.data_ptr(), n)
    assert all(r.info["flags"] == 0 and len(r.blobs) == 0 for r in seg.fetch())
    seg.close()


@pytest.mark.parametrize("kind", ["many_small", "one_huge", "mixed", "edge"])
def test_frames_with_5k_to_8k_lines_are_labelled_in_lds(kind):
    # k_ccl_lds holds up to 8160 lines of a frame (a 4096 x 4096 frame of 256 individuals has ~7.7 k); more go to the global-memory chain.
    # Line counts on both sides of the old (5120) and the new limit, with every grouping strategy: many 1-line blobs (as many raw blobs as
    # lines), one blob of thousands of lines (the whole-frame bitonic sort at 8192 keys), a mix, and counts right at the limit
    rng = np.random.default_rng(17)
    W, H = 1024, 2048
    bg = np.full((H, W), 120, np.uint8)
    frames = []
    if kind == "many_small":
        for n_lines in (5200, 6500, 8100):
            f = bg.copy()
            ys = rng.permutation(H // 2)[:n_lines // 8] * 2          # 8 separate 1-line blobs on every other row
            for y in ys:
                for j in range(8):
                    f[y, 10 + 100 * j: 10 + 100 * j + int(rng.integers(1, 60))] = 10
            frames.append(f)
    elif kind == "one_huge":
        for n_rows in (1800, 2040):
            f = bg.copy()
            f[4:4 + n_rows, 100:110] = 10                           # a 1800-line column ...
            for y in range(4, 4 + n_rows): f[y, 300:300 + int(rng.integers(1, 40))] = 10; f[y, 500:520] = 10; f[y, 700:705] = 10
            f[4:4 + n_rows:7, 110:300] = 10                          # ... that joins some of the combs
            frames.append(f)
    elif kind == "mixed":
        for dens in (0.0025, 0.0033):
            f, b2 = synth.random_scene(rng, W, H, density=0.02)
            f = bg.copy(); m = rng.random((H, W)) < dens; f[m] = 5
            f[100:700, 50:60] = 5
            frames.append(f)
    else:
        for n_lines in (8159, 8160, 8161, 8192):                     # at, and just past, the capacity (the rest take the global chain)
            f = bg.copy()
            k = 0
            for y in range(0, H, 1):
                for j in range(4):
                    if k < n_lines: f[y, 8 + 200 * j: 8 + 200 * j + 3 + (y % 5)] = 10; k += 1
            frames.append(f)
    res = run_gpu(np.stack(frames), bg, max_runs=20000, max_blobs=16384, max_pixels=1 << 21)
    for r, fr in zip(res, frames):
        assert 5000 < r.info["n_raw_runs"] < 9000, r.info["n_raw_runs"]
        assert_frame_equal(r, fr, bg)


def test_live_settings_take_effect_on_the_next_call():
    # the reference re-reads cm_per_pixel / detect_size_filter and the thresholds on every apply() (BackgroundSubtraction.cpp:137-143):
    # trexhip_update_params changes them on a living context, results equal a context created with those values
    rng = np.random.default_rng(3)
    W, H = 512, 64
    bg = rng.integers(60, 200, (H, W)).astype(np.uint8)
    fr = np.clip(bg.astype(int) + rng.integers(-70, 70, (H, W)), 0, 255).astype(np.uint8)
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, max_blobs=32768))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seen = []
    for kw in (dict(), dict(threshold=30), dict(threshold=30, inclusive=0), dict(threshold=20, threshold_maximum=60),
               dict(threshold=15, threshold_maximum=255, inclusive=1, size_ranges=[(3, 9), (20, 1000)], cm_per_pixel=0.5),
               dict(size_ranges=[(2, 50)], cm_per_pixel=1.0), dict(absolute_difference=0, size_ranges=[]), dict(absolute_difference=1, image_invert=1)):
        seg.update_params(**kw)
        seg.segment_device(d.data_ptr(), 1)
        r = seg.fetch()[0]
        lp = seg.params
        okw = dict(threshold=lp.threshold, threshold_maximum=lp.threshold_maximum, inclusive=lp.inclusive, absolute_difference=lp.absolute_difference,
                   image_invert=lp.image_invert, cm_per_pixel=lp.cm_per_pixel, size_ranges=[(lp.ranges[2 * i], lp.ranges[2 * i + 1]) for i in range(lp.n_ranges)])
        assert_frame_equal(r, fr, bg, **okw)
        seen.append(len(r.blobs))
    assert len(set(seen)) >= 6, seen
    with pytest.raises(capi.TrexHipError):
        seg.update_params(size_ranges=[(1, 2)] * 9)
    with pytest.raises(capi.TrexHipError):
        seg.update_params(cm_per_pixel=0.0)
    seg.segment_device(d.data_ptr(), 1)                 # a refused update leaves the previous values in place
    assert len(seg.fetch()[0].blobs) == seen[-1]
    seg.close()


def test_live_update_between_a_segment_call_and_its_crops_leaves_the_batch_alone():
    # trexhip_update_params used to rewrite the configuration the downstream calls of the ALREADY segmented batch read: after a live
    # image_invert change the crops of that batch came out with the new inversion and disagreed with its pixel arrays.  The batch keeps its own.
    rng = np.random.default_rng(4)
    W, H = 512, 64
    bg = rng.integers(60, 200, (H, W)).astype(np.uint8)
    fr = np.clip(bg.astype(int) + rng.integers(-70, 70, (H, W)), 0, 255).astype(np.uint8)[None]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, max_blobs=32768))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.update_params(threshold=40)
    seg.segment_device(d.data_ptr(), 1)
    r = seg.fetch()[0]
    nb = len(r.blobs)
    assert nb > 20
    a = torch.zeros((nb, 80, 80), dtype=torch.uint8, device="cuda"); b = torch.zeros_like(a)
    seg.crops_device(a.data_ptr(), nb)
    seg.update_params(image_invert=1)                     # takes effect from the NEXT segment call
    seg.crops_device(b.data_ptr(), nb)
    seg.synchronize()
    assert torch.equal(a, b)
    want = np.stack([oracle.crop_none(fr[0], bg, bb, r.runs) for bb in r.blobs])
    assert np.array_equal(a.cpu().numpy(), want)
    seg.segment_device(d.data_ptr(), 1)                   # ... and from then on it does
    r2 = seg.fetch()[0]
    assert_frame_equal(r2, fr[0], bg, threshold=40, image_invert=1)
    seg.close()


def test_errors():
    p = capi.default_params(64, 64)
    seg = capi.Segmenter(p)
    with pytest.raises(capi.TrexHipError):      # background not set (BackgroundSubtraction.cpp:58-73 waits; we refuse)
        seg.segment_host([np.zeros((64, 64), np.uint8)])
    seg.set_background(np.zeros((64, 64), np.uint8))
    with pytest.raises(capi.TrexHipError):      # more frames than max_batch
        seg.segment_host([np.zeros((64, 64), np.uint8)] * 17)
    seg.close()
    with pytest.raises(capi.TrexHipError):
        capi.Segmenter(capi.default_params(70000, 64))
    with pytest.raises(capi.TrexHipError):      # pooled tables are indexed with 32 bits
        capi.Segmenter(capi.default_params(64, 64, max_batch=8192, max_pixels=1 << 20))


@pytest.mark.parametrize("channels,color_channel", [(3, -1), (4, -1), (3, 1), (4, 2), (3, 7)])
def test_colour_tile_input(channels, color_channel):
    # what TRex really hands over: BGR / BGRA tile images (BackgroundSubtraction.cpp:162-180)
    rng = np.random.default_rng(channels * 10 + color_channel)
    H, W = 96, 256
    col = rng.integers(0, 256, (2, H, W, channels)).astype(np.uint8)
    col[:, 20:40, 30:90] //= 8
    bg = np.full((H, W), 140, np.uint8)
    if 0 <= color_channel < channels:
        gray = col[..., color_channel]
    else:
        gray = oracle.bgr2gray(col)
    p = capi.default_params(W, H, max_batch=2, max_blobs=32768, threshold=40)
    seg = capi.Segmenter(p)
    seg.set_background(bg)
    seg.segment_color_host([c for c in col], color_channel)
    res = seg.fetch()
    for r, g in zip(res, gray):
        assert_frame_equal(r, np.ascontiguousarray(g), bg, threshold=40)
    seg.close()


@pytest.mark.parametrize("kw", [dict(use_closing=1, closing_size=3), dict(dilation_size=1), dict(dilation_size=-1),
                                 dict(use_closing=1, closing_size=5, dilation_size=2), dict(dilation_size=-3),
                                 dict(use_closing=1, closing_size=4), dict(use_closing=1, closing_size=15, dilation_size=7)])
@pytest.mark.parametrize("shape", [(90, 70), (2048, 64), (1000, 37)])
def test_morphology_bit_exact(kw, shape):
    # use_closing / dilation_size (core/default_config.cpp:1163-1165): device bit-mask morphology vs the oracle,
    # whose morphology is itself cross-checked against scipy.ndimage (tests/test_oracle_scipy.py)
    W, H = shape
    rng = np.random.default_rng(W + H + len(kw))
    fr, bg = synth.random_scene(rng, W, H, density=0.2)
    fr[rng.random((H, W)) < 0.01] = 0          # zero-valued pixels under the mask stay background
    fr[0, :] = 5; fr[:, W - 1] = 7             # foreground on the borders (border handling of the element)
    res = run_gpu(fr[None], bg, **kw)
    assert_frame_equal(res[0], fr, bg, **kw)


@pytest.mark.parametrize("method", [0, 1, 2, 3])
def test_background_from_samples(method):
    rng = np.random.default_rng(method)
    H, W, n = 48, 160, 37
    fr = rng.integers(0, 256, (n, H, W)).astype(np.uint8)
    fr[:, 3, :] = np.arange(n)[:, None] % 2 * 255          # means ending in .5 exercise the rounding rule
    if method == 3:
        fr = (fr // 32 * 32).astype(np.uint8)               # few distinct values per pixel: real modes and real ties
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1))
    d = torch.from_numpy(fr).cuda()
    got = seg.generate_average(d.data_ptr(), n, method)
    assert np.array_equal(got, oracle.generate_average(fr, method))
    seg.segment_device(d.data_ptr(), 1)                    # the generated image is the context's background now
    seg.fetch()
    seg.close()


def test_two_contexts_from_two_threads():
    # two contexts (own streams) driven concurrently from two host threads -- the software-pipelined lanes of bench.py and the
    # detect / identity threads of TRex: results must not depend on the interleaving
    import threading
    rng = np.random.default_rng(21)
    scenes = [synth.random_scene(rng, 640, 360, density=0.01) for _ in range(4)]
    errors = []

    def worker(k):
        try:
            fr, bg = scenes[k % 2]
            seg = capi.Segmenter(capi.default_params(640, 360, max_batch=2, max_blobs=32768), stream=None)   # the context's own stream
            seg.set_background(bg)
            for it in range(6):
                fr2, _ = scenes[(k + it) % 4]
                frames = np.stack([fr, np.where(bg == fr2, fr, fr2)])
                d = torch.from_numpy(frames).cuda()
                torch.cuda.synchronize()            # the upload ran on torch's stream
                seg.segment_device(d.data_ptr(), 2)
                res = seg.fetch()
                for r, f in zip(res, frames):
                    assert_frame_equal(r, f, bg)
            seg.close()
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e)[:300])

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors


@pytest.mark.parametrize("channels", [3, 4])
@pytest.mark.parametrize("shape,shift", [((64, 32), 0), ((36, 10), 0), ((64, 32), 4), ((2048, 8), 0)])
def test_colour_tiles_on_the_device(channels, shape, shift):
    # cvtColor on the device (trexhip_segment_color_device): the 16-pixels-per-thread kernel takes frames whose pixel count is a multiple of
    # 16 at 16-byte aligned addresses, everything else the 4-pixel kernel -- both must give cv::cvtColor's fixed-point grey (oracle.bgr2gray)
    W, H = shape
    rng = np.random.default_rng(W * 7 + H + channels + shift)
    col = rng.integers(0, 256, (2, H, W, channels)).astype(np.uint8)
    col[:, 2:6, 3:30] //= 8
    bg = np.full((H, W), 140, np.uint8)
    gray = oracle.bgr2gray(col)
    p = capi.default_params(W, H, max_batch=2, max_blobs=4096, threshold=40)
    seg = capi.Segmenter(p)
    seg.set_background(bg)
    buf = torch.zeros(col.size + 64, dtype=torch.uint8, device="cuda")
    buf[shift:shift + col.size] = torch.from_numpy(col.reshape(-1)).cuda()
    torch.cuda.synchronize()
    seg.segment_color_device(buf.data_ptr() + shift, 2, channels)
    res = seg.fetch()
    for r, g in zip(res, gray):
        assert_frame_equal(r, np.ascontiguousarray(g), bg, threshold=40)
    seg.close()


@pytest.mark.parametrize("bands", [2, 3, 8])
def test_several_workgroups_per_frame_give_the_same_tables(bands, monkeypatch):
    """k_ccl_band (round 6): a frame's rows cut into bands labelled by their own workgroups, k_ccl_lds links the seams and goes on.  Forced here
    (TREXHIP_CCL_BANDS; by itself the library bands only frames of more lines than the M instance holds, when the launch leaves CUs idle): blobs
    that cross every seam, seams through empty rows, a frame whose band holds more lines than a band workgroup takes (it falls back to the one
    workgroup per frame), 8- and 4-connectivity -- every table byte for byte the oracle's."""
    monkeypatch.setenv("TREXHIP_CCL_BANDS", str(bands))
    rng = np.random.default_rng(100 + bands)
    W, H = 1024, 2048
    bg = np.full((H, W), 120, np.uint8)
    frames = []
    f = bg.copy()                                                    # tall columns and diagonals through every seam
    for x in range(20, 1000, 480):
        f[int(rng.integers(0, 400)):int(rng.integers(1500, H)), x:x + int(rng.integers(1, 9))] = 10
    for k in range(0, 400):
        f[300 + k * 2:300 + k * 2 + 3, 40 + k:40 + k + 2] = 10       # a staircase: 8-connected only through its corners in places
    frames.append(f)
    f, _ = synth.random_scene(rng, W, H, density=0.018); frames.append(f)        # ~6000 lines: labelled in LDS, banded
    f, _ = synth.random_scene(rng, W, H, density=0.05); frames.append(f)         # ~16000 lines: the global-memory chain (the band workgroups leave it alone)
    f = bg.copy(); f[H // 2 - 3:H // 2 + 3, :] = 10; f[::2, 5:9] = 10; frames.append(f)       # a bar on the middle seam, single-line blobs everywhere
    f = bg.copy()                                                    # ~5000 lines inside ONE band of rows: the band workgroup cannot hold them
    for y in range(0, 250):
        for j in range(20):
            f[y, 10 + 50 * j:10 + 50 * j + 3 + (y % 7)] = 10
    frames.append(f)
    frames.append(bg.copy())                                         # an empty frame
    for kw in (dict(), dict(connectivity=4)):
        res = run_gpu(np.stack(frames), bg, max_runs=20000, max_blobs=16384, max_pixels=1 << 21, **kw)
        for r, fr in zip(res, frames):
            assert_frame_equal(r, fr, bg, **kw)
    # odd frame heights: the last band is shorter, or there are fewer bands than asked for
    for (w, h) in ((320, 129), (640, 1000), (4096, 300)):
        fr, b2 = synth.random_scene(rng, w, h, density=0.08)
        assert_frame_equal(run_gpu(fr[None], b2)[0], fr, b2)


def test_heavy_frames_are_banded_from_the_second_call_on_and_stay_the_same():
    """the automatic policy: a context whose frames carry more lines than the M instance holds (C5: 4096 x 4096, 256 individuals) bands them from
    its second call on (the first call's kernels write the hint); the tables of call 1 (one workgroup per frame) and call 2 (banded) are the same"""
    fr, bg = synth.batch("C5", 2)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=1024, max_pixels=1 << 20, max_runs=32768))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    sig = []
    for _ in range(3):
        seg.segment_device(d.data_ptr(), n)
        res = seg.fetch()
        assert all(r.info["flags"] == 0 and r.info["n_raw_runs"] > 3840 for r in res)
        sig.append([(r.blobs.tobytes(), r.runs.tobytes(), r.pixels.tobytes()) for r in res])
    assert sig[0] == sig[1] == sig[2]
    assert_frame_equal(res[0], fr[0], bg)
    seg.close()


@pytest.mark.parametrize("W", [2080, 3072, 4096])
@pytest.mark.parametrize("kw", [dict(), dict(absolute_difference=0), dict(zero_is_background=0), dict(absolute_difference=0, zero_is_background=0),
                                dict(threshold_maximum=80), dict(image_invert=1, threshold=25)])
def test_wide_frames_with_the_background_in_registers(W, kw):
    """k_rows32b<2> (round 6): frames of 2049 .. 4096 pixels per row, several frames per wave with the background row in registers -- every compile-time
    threshold mode and the generic one, widths that end inside the second chunk, lines across the chunk border at x = 2047 / 2048, batches whose
    frame count has the divisors 8, 3 and 2 (frames per wave)"""
    rng = np.random.default_rng(W + len(kw))
    H = 24
    bg = rng.integers(60, 200, (H, W)).astype(np.uint8)
    for n in (8, 3, 2):
        frames = []
        for i in range(n):
            fr = np.clip(bg.astype(int) + rng.integers(-12, 12, (H, W)), 0, 255).astype(np.uint8)
            m = rng.random((H, W)) < 0.01
            fr[m] = np.where(rng.random(m.sum()) < 0.5, 0, 255)
            fr[i % H, 2040:2060] = 0                      # a line across the chunk border
            fr[(i + 5) % H, W - 9:W] = 255                # ... and one that ends with the row
            fr[(i + 9) % H, 2047] = 0; fr[(i + 11) % H, 2048] = 0
            frames.append(fr)
        res = run_gpu(np.stack(frames), bg, **kw)
        for r, fr in zip(res, frames):
            assert_frame_equal(r, fr, bg, **kw)
