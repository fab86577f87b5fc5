"""Device posture (one wave per blob) vs the CPU oracle: every stage shares the operation order (the EFT's sin / cos and float sums included,
since round 5), so outlines, tail / head indices and midline segments are compared for equality."""
import numpy as np
import pytest
import torch
from oracle import oracle
from trex_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_posture(frames, bg, table=0, thr=None, **kw):
    n, H, W = frames.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=4096))
    seg.set_background(bg)
    d = torch.from_numpy(frames).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    if table == 1:
        seg.rethreshold(thr, 0, [])
        res = seg.fetch(rethreshold=True)
    total = sum(len(r.blobs) for r in res)
    MP = kw.get("max_points", 512)
    outline = torch.zeros((total, MP, 2), dtype=torch.float32, device="cuda")
    segs = torch.zeros((total, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(total, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), table=table, **kw)
    seg.synchronize()
    out = (res, outline.cpu().numpy(), segs.cpu().numpy(), info.cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1))
    seg.close()
    return out


def compare(res, outline, segs, info, pp, **_ignored):
    """Device == CPU restatement, bit for bit (round 5): the EFT's sin / cos and the order of its float sums are the same operations on both sides
    (posture.hip det_sincosf / oracle/trex_posture.c det_sincosf, the wave scan and the 64 interleaved partial sums), the centre is summed in
    Outline.cpp:502-505's order, everything else was elementwise already.  So the outline, the tail and head, the number of midline segments and
    the segments themselves must be EQUAL -- no tie classes, no tolerance.  -> (blobs compared, 0)"""
    n_cmp = 0
    for r in res:
        for k, b in enumerate(r.blobs):
            bi = int(r.info["blob_begin"]) + k
            rs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            oi, oo, osg = oracle.posture(rs, (int(b["x0"]), int(b["y0"])), pp)
            gi = info[bi]
            if b["n_runs"] > 2048 or int(b["y1"]) - int(b["y0"]) + 1 > 1022:    # beyond the device's per-blob LDS capacity
                assert gi["status"] == 2
                continue
            assert gi["status"] == oi["status"], (bi, gi, oi)
            assert gi["n_traced"] == oi["n_traced"]
            if oi["status"] not in (0, 3, 4):
                continue
            assert gi["n_outline"] == oi["n_outline"]
            go = outline[bi, :gi["n_outline"]]
            assert np.array_equal(go, oo[:oi["n_outline"]]), (bi, "outline differs by %g" % np.abs(go - oo[:oi["n_outline"]]).max())
            if oi["status"] == 3:           # no curvature peak: the outline alone
                continue
            n_cmp += 1
            assert gi["tail_index"] == oi["tail_index"] and gi["head_index"] == oi["head_index"], (bi, gi, oi)
            assert gi["n_segments"] == oi["n_segments"], (bi, gi, oi)
            assert np.array_equal(segs[bi, :gi["n_segments"]], osg[:segs.shape[1]]), bi
    assert n_cmp > 0
    print("posture: %d blobs compared, all identical to the CPU restatement" % n_cmp)
    return n_cmp, 0


def test_synthetic_individuals():
    fr, bg = synth.batch("C2", 3)
    res, outline, segs, info = run_posture(fr, bg)
    n, ties = compare(res, outline, segs, info, oracle.posture_params(max_points=512), max_ties=0.03)
    assert n == 96
    ok = info["status"] == 0
    assert ok.mean() > 0.9
    # midline length of the synthetic 36x10 ellipses
    for bi in np.flatnonzero(ok)[:20]:
        s = segs[bi, :info[bi]["n_segments"], :2]
        assert 20 < np.linalg.norm(np.diff(s, axis=0), axis=1).sum() < 40


@pytest.mark.parametrize("kw", [dict(outline_resample=0.5), dict(outline_resample=2.0, outline_smooth_samples=0),
                                 dict(outline_approximate=0), dict(outline_approximate=1, midline_walk_offset=0.1),
                                 dict(outline_resample=0.5, midline_walk_offset=0.45), dict(midline_walk_offset=0.07), dict(outline_resample=0.5, midline_walk_offset=0.06)])
def test_setting_variants_and_odd_shapes(kw):
    rng = np.random.default_rng(3)
    H, W = 160, 320
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    fr[(xx - 60) ** 2 / 900 + (yy - 50) ** 2 / 100 <= 1] = 20                 # ellipse
    fr[100:104, 20:120] = 30                                                     # thin bar
    fr[20:60, 200:204] = 30; fr[56:60, 200:260] = 30                             # L shape
    fr[120, 250] = 10                                                            # single pixel
    fr[130:132, 260:262] = 10; fr[132, 262] = 10                                 # tiny diagonal
    for _ in range(6):                                                           # random blobs with holes and dents
        cy, cx = rng.integers(30, H - 30), rng.integers(140, 190)
        m = ((xx - cx) ** 2 + (yy - cy) ** 2 <= rng.integers(20, 120)) & (rng.random((H, W)) < 0.9)
        fr[m] = 40
    res, outline, segs, info = run_posture(fr[None], bg, **kw)
    # order-1 EFT turns every outline into an exact ellipse whose two tips have EQUAL curvature: the tail is a coin flip
    # decided by float rounding there, so only the closed curve is compared for that variant
    compare(res, outline, segs, info, oracle.posture_params(max_points=512, **kw))     # (the tie rate is printed: discs and order-1 ellipses are mostly ties)


@pytest.mark.parametrize("walk_offset", [0.02, 0.04, 0.045, 0.08, 0.085, 0.16, 0.165, 0.2])
@pytest.mark.parametrize("max_points", [256, 512, 1024])
def test_midline_walk_across_the_lane_group_boundaries(walk_offset, max_points):
    """The two-pointer walk runs in k_posture_walk for blobs whose searches hold at most 8 / 16 candidates (max(3, offset * outline points)) and
    inside k_posture beyond; which of the groups a launch uses also depends on max_points (LDS per wave).  Offsets chosen so that the scene's
    outlines (about 50 ... 200 points) fall on both sides of both limits; compare() checks the device's segments bit for bit against the CPU walk of
    the device's own outline and the outline / tail / head against the restatement."""
    fr, bg = synth.batch("C2", 2)
    kw = dict(midline_walk_offset=walk_offset, max_points=max_points)
    res, outline, segs, info = run_posture(fr, bg, **kw)
    n, _ = compare(res, outline, segs, info, oracle.posture_params(**kw), max_heads=0.1)
    assert n >= 40
    ok = info["status"] == 0
    mo = np.maximum(3, (walk_offset * info["n_outline"][ok]).astype(int))
    print("searches of <= 8 / 9..16 / > 16 candidates: %d / %d / %d blobs" % ((mo <= 8).sum(), ((mo > 8) & (mo <= 16)).sum(), (mo > 16).sum()))


def test_rethreshold_table_and_capacity():
    fr, bg = synth.batch("C2", 1)
    res, outline, segs, info = run_posture(fr, bg, table=1, thr=40)
    compare(res, outline, segs, info, oracle.posture_params(max_points=512))
    res, outline, segs, info = run_posture(fr, bg, max_points=32)
    assert np.all(info["status"] == 2)                                            # outline longer than the capacity


@pytest.mark.parametrize("kw", [dict(), dict(midline_resolution=12, midline_stiff_percentage=0.3), dict(midline_stiff_percentage=0.0),
                                 dict(midline_invert=1), dict(midline_start_with_head=1)])
def test_midline_post_process_and_normalize(kw):
    fr, bg = synth.batch("C2", 2)
    check_midline(fr, bg, **kw)


@pytest.mark.parametrize("kw", [dict(), dict(midline_invert=1), dict(midline_start_with_head=1), dict(midline_stiff_percentage=0.0)])
def test_midline_movement_history_flip(kw):
    # posture_direction_smoothing > 1: the reference hands MovementInformation::direction to Midline::post_process, which turns a midline round
    # when its direction points against the movement (Outline.cpp:905-961).  The vector is tracker state: random unit vectors here, a third
    # of the blobs without information (0, 0), a few vectors longer than 1 (acos of a value outside [-1, 1] is NaN: no flip then)
    fr, bg = synth.batch("C2", 2)
    n_ok, total, flips = check_midline(fr, bg, movement_seed=5, **kw)
    assert 0.15 * total < flips < 0.6 * total, (flips, total)


def check_midline(fr, bg, min_ok=0.8, movement_seed=None, **kw):
    # Midline::post_process + normalize (Outline.cpp:895-1060,1270-1454) on the device's own raw segments vs the CPU restatement
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    total = sum(len(r.blobs) for r in res)
    MP = 512
    R = kw.get("midline_resolution", 25)
    outline = torch.zeros((total, MP, 2), dtype=torch.float32, device="cuda")
    segs = torch.zeros((total, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(total, outline.data_ptr(), segs.data_ptr(), info.data_ptr())
    seg.synchronize()
    raw = segs.cpu().numpy().copy()
    pinfo = info.cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1)
    mid = torch.zeros((total, R, 4), dtype=torch.float32, device="cuda")
    minfo = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    move = None
    if movement_seed is not None:
        rng = np.random.default_rng(movement_seed)
        ang = rng.uniform(0, 2 * np.pi, total)
        move = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
        move[rng.random(total) < 0.33] = 0
        move[rng.random(total) < 0.05] *= 3.0
        d_move = torch.from_numpy(move).cuda()
        seg.midline_device(total, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr(), d_movement_ptr=d_move.data_ptr(), **kw)
    else:
        seg.midline_device(total, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr(), **kw)
    seg.synchronize()
    proc = segs.cpu().numpy(); mid = mid.cpu().numpy()
    minfo = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    n_ok = flips = 0
    for bi in range(total):
        ns = int(pinfo[bi]["n_segments"])
        if pinfo[bi]["status"] != 0:
            assert minfo[bi]["status"] == 1
            continue
        oi, oproc, onorm = oracle.midline_normalize(raw[bi, :ns], resolution=R, stiff=kw.get("midline_stiff_percentage", 0.15),
                                                    invert=bool(kw.get("midline_invert", 0)), start_with_head=bool(kw.get("midline_start_with_head", 0)),
                                                    movement=None if move is None else move[bi])
        gi = minfo[bi]
        assert gi["status"] == oi["status"] and gi["n"] == oi["n"], (bi, gi, oi)
        assert gi["reserved"][0] == oi["reserved"][0], (bi, "movement flip", None if move is None else move[bi])
        flips += int(gi["reserved"][0])
        if move is not None and not move[bi].any():
            assert gi["reserved"][0] == 0
        # post_process is sqrt / divide / multiply / add only: bit-exact
        assert np.array_equal(proc[bi, :ns], oproc), bi
        if oi["status"] != 0:
            continue
        n_ok += 1
        assert gi["offx"] == oi["offx"] and gi["offy"] == oi["offy"]
        assert abs(gi["len"] - oi["len"]) <= 1e-4 * oi["len"]
        assert abs(gi["angle"] - oi["angle"]) <= 1e-5
        assert np.abs(mid[bi] - onorm).max() <= 1e-3
    assert n_ok > min_ok * total
    seg.close()
    return (n_ok, total, flips) if movement_seed is not None else (n_ok, total)


@pytest.mark.parametrize("legacy", [False, True])
def test_crops_posture_normalisation(legacy):
    # individual_image_normalization = posture / legacy end to end on the device: posture -> midline -> transform -> warp.
    # The crop is checked bit-exactly against the CPU restatement fed with the device's Midline::angle()/offset().
    fr, bg = synth.batch("C2", 2)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    total = sum(len(r.blobs) for r in res)
    MP = 512
    outline = torch.zeros((total, MP, 2), dtype=torch.float32, device="cuda")
    segs = torch.zeros((total, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    mid = torch.zeros((total, 25, 4), dtype=torch.float32, device="cuda")
    minfo = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(total, outline.data_ptr(), segs.data_ptr(), info.data_ptr())
    seg.midline_device(total, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    crops = torch.full((total, 80, 80), 9, dtype=torch.uint8, device="cuda")
    seg.crops_posture_device(crops.data_ptr(), total, minfo.data_ptr(), legacy=legacy)
    seg.synchronize()
    crops = crops.cpu().numpy()
    mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    filled = 0
    for r, f in zip(res, fr):
        for k, b in enumerate(r.blobs):
            bi = int(r.info["blob_begin"]) + k
            if mi[bi]["status"] != 0:
                assert crops[bi].sum() == 0
                continue
            tr = oracle.midline_transform(mi[bi]["angle"], mi[bi]["offx"], mi[bi]["offy"], legacy)
            want, _ = oracle.crop_normalized(f, bg, b, r.runs, tr6=tr, midline_length=float(mi[bi]["len"]), legacy=legacy)
            assert np.array_equal(crops[bi], want), bi
            # the whole animal must land inside the crop: same pixel mass as the un-normalised crop within interpolation loss
            filled += crops[bi].sum() > 0.7 * oracle.crop_none(f, bg, b, r.runs).sum()
    assert filled > 0.8 * total
    seg.close()


def test_large_animals_take_fewer_blobs_per_workgroup():
    # outlines beyond 1024 traced points / blobs beyond 254 rows: max_points up to 4096 (one blob per workgroup), 2048 lines, 1022 rows
    H, W = 700, 900
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    def body(cx, cy, a, b, th):
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        fr[(u / a) ** 2 + (v / b) ** 2 <= 1] = 40
    body(200, 350, 30, 300, 0.05)        # 600 rows tall, narrow enough for the one-word-per-row bitmap
    body(600, 200, 250, 45, 0.3)         # 500 px long, wider than 64 px: the sequential trace
    body(700, 600, 40, 12, 1.0)          # an ordinary one beside them
    for mp in (4096, 2048):
        res, outline, segs, info = run_posture(fr[None], bg, max_points=mp, outline_resample=1.0)
        r = res[0]
        assert len(r.blobs) == 3
        pp = oracle.posture_params(max_points=mp, outline_resample=1.0)
        ok = 0
        for k, b in enumerate(r.blobs):
            rs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            oi, oo, osg = oracle.posture(rs, (int(b["x0"]), int(b["y0"])), pp)
            gi = info[k]
            assert gi["status"] == oi["status"] and gi["n_traced"] == oi["n_traced"], (mp, k, gi, oi)
            if oi["status"] != 0:
                assert oi["status"] == 2 and mp == 2048          # both large animals need more than 2048 traced points
                continue
            ok += 1
            assert gi["n_outline"] == oi["n_outline"] and gi["n_segments"] == oi["n_segments"]
            go = outline[k, :gi["n_outline"]]
            # (round 5: the same tail on symmetric ellipses too -- the two tips' curvatures are the same floats on both sides)
            assert np.array_equal(go, oo[:oi["n_outline"]]) and gi["head_index"] == oi["head_index"]
            assert np.array_equal(segs[k, :gi["n_segments"]], osg[:gi["n_segments"]])
        assert ok == (3 if mp == 4096 else 1)


@pytest.mark.parametrize("kw", [dict(posture_closing_steps=1), dict(peak_mode=1)])
def test_unimplemented_posture_settings_are_refused(kw):
    fr, bg = synth.batch("C2", 1)
    with pytest.raises(capi.TrexHipError) as e:
        run_posture(fr, bg, **kw)
    assert e.value.code == -4 and "not implemented" in str(e.value)


def _retry_scene(seed, n=40, H=600, W=800):
    """individuals for which the first threshold gives no midline: dark cores in faint discs, crossing bars, speckles, specks"""
    rng = np.random.default_rng(seed)
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.astype(np.int32).copy()
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        cx, cy = 60 + (i % 8) * 95 + rng.integers(-10, 10), 50 + (i // 8) * 110 + rng.integers(-10, 10)
        kind = i % 5
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        if kind == 0:
            halo = (u / 26) ** 2 + (v / 24) ** 2 <= 1
            fr[halo] = np.minimum(fr[halo], 200 - rng.integers(17, 30))
            core = (u / 22) ** 2 + (v / 5) ** 2 <= 1
            fr[core] = 200 - rng.integers(80, 150)
        elif kind == 1:
            d = (u / 25) ** 2 + (v / 7) ** 2
            m = d <= 1
            fr[m] = (200 - (20 + 120 * (1 - d[m]))).astype(np.int32)
        elif kind == 2:
            a = (np.abs(u) <= 24) & (np.abs(v) <= 3); b = (np.abs(v) <= 24) & (np.abs(u) <= 3)
            fr[a] = 200 - 100; fr[b] = np.minimum(fr[b], 200 - rng.integers(18, 40))
        elif kind == 3:
            m = ((u / 20) ** 2 + (v / 8) ** 2 <= 1)
            fr[m] = 200 - rng.integers(16, 120, m.sum())
        else:
            m = ((u / 4) ** 2 + (v / 2) ** 2 <= 1) if i % 10 == 4 else ((xx == cx) & (yy == cy))          # a speck / a single pixel: no midline at any threshold
            fr[m] = 200 - 60
    return np.clip(fr, 0, 255).astype(np.uint8), bg


@pytest.mark.parametrize("seed,method,tpt", [(0, 0, 15), (1, 0, 15), (2, 1, 20), (3, 0, 30)])
def test_posture_retry_loop_equals_oracle(seed, method, tpt):
    # posture::calculate_posture's `threshold += 2` loop (Posture.cpp:331-382) on the device, per blob, against the CPU restatement
    fr, bg = _retry_scene(seed)
    H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, max_blobs=4096))
    seg.set_background(bg)
    d = torch.from_numpy(fr[None]).cuda()
    seg.segment_device(d.data_ptr(), 1)
    r = seg.fetch()[0]
    n, MP = len(r.blobs), 512
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    thr = torch.zeros(n, dtype=torch.int32, device="cuda"); its = torch.zeros(n, dtype=torch.int32, device="cuda")
    seg.posture_auto_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), method=method, track_posture_threshold=tpt,
                            d_threshold_ptr=thr.data_ptr(), d_iterations_ptr=its.data_ptr(), max_points=MP)
    seg.synchronize()
    gi = info.cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1); go = outline.cpu().numpy(); gs = segs.cpu().numpy()
    gt, gn = thr.cpu().numpy(), its.cpu().numpy()
    pp = oracle.posture_params(max_points=MP)
    retried = ok_late = failed = n_same = n_cmp = 0
    for k, b in enumerate(r.blobs):
        rs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]; px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        oi, oo, osg = oracle.posture_auto(rs, px, bg, method, tpt, pp)
        assert gn[k] == oi["iterations"] and gt[k] == oi["threshold"], (k, gn[k], gt[k], oi)
        assert (gi[k]["status"] == 0) == (oi["status"] == 0) and gi[k]["n_outline"] == oi["n_outline"], (k, gi[k], oi)
        retried += oi["iterations"] > 1; ok_late += oi["iterations"] > 1 and oi["status"] == 0; failed += oi["status"] != 0
        if oi["n_outline"]:
            assert np.array_equal(go[k, :oi["n_outline"]], oo[:oi["n_outline"]]), k      # bit for bit, symmetric bodies included (round 5)
            if oi["status"] == 0:
                assert gi[k]["head_index"] == oi["head_index"] and gi[k]["n_segments"] == oi["n_segments"], (k, gi[k], oi)
                assert np.array_equal(gs[k, :oi["n_segments"]], osg[:oi["n_segments"]]), k
                n_same += 1
            n_cmp += oi["status"] == 0
    assert retried >= 2 and ok_late >= 1          # the scene really exercises the loop
    assert n_same == n_cmp, (n_same, n_cmp)
    seg.close()


def test_retry_loop_with_sub_blobs_of_more_lines_than_twice_the_parents():
    # The LDS line capacity of a posture launch is estimated from the detect blobs (2 x their lines).  A thresholded line can split into
    # many more: a comb whose odd rows alternate strong / faint pixels is one line per row at the detect threshold and 30 per odd row at
    # track_posture_threshold.  The round is then repeated with the kernel's real capacities -- never counted as a failed attempt
    # (which would move the blob on to threshold + 2 and differ from calculate_posture, Posture.cpp:331-382).
    H, W = 256, 512
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy()
    for j, (y0, x0) in enumerate([(20, 30), (90, 200), (170, 380)]):
        rows, width = 24 + 4 * j, 60
        for r in range(rows):
            fr[y0 + r, x0:x0 + width] = 200 - 60
            if r % 2 == 1:
                fr[y0 + r, x0 + 1:x0 + width:2] = 200 - 20          # faint: passes detect (> 15), fails the posture threshold (30)
    fr[120:150, 40:70] = 200 - 60                                      # an ordinary blob beside them
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=1, max_blobs=64))
    seg.set_background(bg)
    d = torch.from_numpy(fr[None]).cuda()
    seg.segment_device(d.data_ptr(), 1)
    r = seg.fetch()[0]
    n, MP = len(r.blobs), 1024
    assert n == 4 and int(r.blobs["n_runs"].max()) <= 40
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    thr = torch.zeros(n, dtype=torch.int32, device="cuda"); its = torch.zeros(n, dtype=torch.int32, device="cuda")
    seg.posture_auto_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), method=0, track_posture_threshold=30,
                            d_threshold_ptr=thr.data_ptr(), d_iterations_ptr=its.data_ptr(), max_points=MP)
    seg.synchronize()
    gi = info.cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1); go = outline.cpu().numpy()
    gt, gn = thr.cpu().numpy(), its.cpu().numpy()
    pp = oracle.posture_params(max_points=MP)
    for k, b in enumerate(r.blobs):
        rs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]; px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
        oi, oo, osg = oracle.posture_auto(rs, px, bg, 0, 30, pp)
        assert gn[k] == oi["iterations"] and gt[k] == oi["threshold"], (k, gn[k], gt[k], oi)
        assert gi[k]["status"] == oi["status"] and gi[k]["n_outline"] == oi["n_outline"], (k, gi[k], oi)
        assert gi[k]["status"] != 2, k                                  # within the kernel's limits: never a capacity status
        if oi["n_outline"]:
            assert np.array_equal(go[k, :oi["n_outline"]], oo[:oi["n_outline"]]), k
    seg.close()


def test_device_stays_next_to_the_naive_reading_of_the_eft():
    """The independent yardstick of a7: compare() above asserts equality with a restatement whose EFT is written in the DEVICE's
    operation order; this one compares the device with the naive reading of the same formulas (oracle.posture(naive=True): libm sinf / cosf per
    harmonic, sequential float sums -- nothing shared with posture.hip but the formulas) on 1600 synthetic individuals: the closed curves within
    2e-3 px, tail and head identical on >= 96 % of the blobs (a curvature near-tie may pick another tip; the curve itself must not move)."""
    fr, bg = synth.batch("C3", 16)
    res, outline, segs, info = run_posture(fr, bg)
    pp = oracle.posture_params(max_points=512)
    n = same = 0
    worst = 0.0
    for r in res:
        for k, b in enumerate(r.blobs):
            bi = int(r.info["blob_begin"]) + k
            rs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
            oi, oo, osg = oracle.posture(rs, (int(b["x0"]), int(b["y0"])), pp, naive=True)
            gi = info[bi]
            assert gi["status"] == oi["status"] and gi["n_outline"] == oi["n_outline"], (bi, gi, oi)
            if oi["status"] != 0:
                continue
            n += 1
            go = outline[bi, :gi["n_outline"]]
            rot = int(np.argmin(np.abs(oo - go[0]).sum(1)))          # the tail is point 0: another tail is a rotation of the same curve
            worst = max(worst, float(np.abs(np.roll(oo, -rot, 0) - go).max()))
            if rot == 0 and gi["head_index"] == oi["head_index"]:
                same += 1         # (the midline walk is a chain of nearest-point decisions: it is compared with the mirror in compare(), not here)
    assert n >= 1500, n
    assert worst <= 2e-3, worst
    assert same >= 0.96 * n, (same, n)
    print("posture vs the naive EFT: %d blobs, %d with the same tail and head, curves within %.2g px" % (n, same, worst))


@pytest.mark.parametrize("order", [4, 5, 9, 15])
def test_more_than_three_harmonics(order):
    """outline_approximate above three (a uint8_t without an upper bound in the reference, core/default_config.cpp:888): the device's general form
    against the mirrored restatement -- equality, as for the default order -- on the synthetic individuals"""
    fr, bg = synth.batch("C2", 2)
    res, outline, segs, info = run_posture(fr, bg, outline_approximate=order)
    n, _ = compare(res, outline, segs, info, oracle.posture_params(max_points=512, outline_approximate=order))
    assert n >= 50


def test_sixteen_harmonics_are_refused_on_both_sides():
    fr, bg = synth.batch("C2", 1)
    with pytest.raises(capi.TrexHipError):
        run_posture(fr, bg, outline_approximate=16)
    with pytest.raises(capi.TrexHipError):
        run_posture(fr, bg, outline_approximate=5, max_points=64)       # the coefficients of the general form need 66 floats of the curvature array
    r0 = oracle.segment(fr[0], bg, oracle.make_params(fr.shape[2], fr.shape[1]))
    b = r0[0][0]
    with pytest.raises(ValueError):
        oracle.posture(r0[1][b["run_begin"]:b["run_begin"] + b["n_runs"]], (int(b["x0"]), int(b["y0"])), oracle.posture_params(max_points=512, outline_approximate=16))
