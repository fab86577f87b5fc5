"""The headline workload at full size (C4: 2048x2048, 100 individuals, 100 classes), device-resident like bench.py: the oracle is too slow
for whole batches here, so the batch is checked through size-independent properties and one frame / a sample of crops against the oracle."""
import numpy as np
import pytest
import torch
from oracle import oracle, cnn_oracle
from trex_amd import capi, synth, weights, dist as tdist

pytestmark = pytest.mark.gpu


def test_c4_detect_crops_identify_table():
    B, classes = 16, 100
    frames, bg = synth.batch_torch("C4", 8, "cuda")
    frames = torch.cat([frames, frames.flip(0)]).contiguous()            # every scene twice: frame i == frame 15 - i
    H, W = frames.shape[1:]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=400, max_pixels=1 << 18, max_runs=32768))
    seg.set_background(bg)
    st = weights.synthetic_state(classes, 4242)
    seg.load_weights(weights.pack_blob(st, classes))
    seg.segment_device(frames.data_ptr(), B)
    res = seg.fetch()
    n = sum(len(r.blobs) for r in res)
    assert n == 100 * B and all(len(r.blobs) == 100 for r in res)
    fr_host = frames.cpu().numpy()
    bg_host = bg.cpu().numpy() if hasattr(bg, "cpu") else bg
    # detect: pixel count = thresholded pixels of the frame; one frame bit-exact against the oracle
    for f in (0, 7):
        d = np.abs(fr_host[f].astype(np.int16) - bg_host.astype(np.int16)) > 15
        assert int(res[f].blobs["n_pixels"].sum()) == int(d.sum())
    ob, orr, opx = oracle.segment(fr_host[3], bg_host, oracle.make_params(W, H))
    assert res[3].runs.tobytes() == orr.tobytes() and res[3].pixels.tobytes() == opx.tobytes()
    assert np.array_equal(res[3].blobs["bid"], ob["bid"]) and np.array_equal(res[3].blobs["m11"], ob["m11"])
    # crops: every crop holds exactly its blob's grey values (blobs are smaller than 80x80: nothing is cut)
    crops = torch.zeros((n, 80, 80), dtype=torch.uint8, device="cuda")
    probs = torch.zeros((n, classes), dtype=torch.float32, device="cuda")
    seg.crops_device(crops.data_ptr(), n)
    seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
    table = torch.zeros((n + 7, tdist.HDR + classes), dtype=torch.int32, device="cuda")
    seg.export_id_table(probs.data_ptr(), n, classes, 4000, table.data_ptr(), n + 7)
    seg.synchronize()
    c = crops.cpu().numpy(); p = probs.cpu().numpy(); t = table.cpu().numpy().view(np.uint32)
    sums = c.reshape(n, -1).astype(np.int64).sum(1)
    for r in res:
        bb = int(r.info["blob_begin"])
        assert np.array_equal(sums[bb:bb + len(r.blobs)], r.blobs["sp"].astype(np.int64))
        assert np.array_equal((c[bb:bb + len(r.blobs)].reshape(len(r.blobs), -1) != 0).sum(1), r.blobs["n_pixels"])     # no grey value is 0 here
    # identity: rows are distributions; the same scene gives the same rows wherever it sits in the batch
    assert np.allclose(p.sum(1), 1.0, atol=1e-5) and (p >= 0).all()
    for f in range(8):
        a, b = res[f], res[15 - f]
        pa = p[int(a.info["blob_begin"]):int(a.info["blob_begin"]) + 100]; pb = p[int(b.info["blob_begin"]):int(b.info["blob_begin"]) + 100]
        assert np.array_equal(a.blobs["bid"], b.blobs["bid"]) and np.array_equal(pa, pb)
    # a sample of crops through the CPU restatement of the network: 1e-4 on softmax (BASELINE.json)
    pick = np.linspace(0, n - 1, 24).astype(int)
    want, _ = cnn_oracle.predict(st, c[pick][..., None], threads=8)
    assert np.abs(p[pick] - want).max() <= 1e-4
    # table: one valid row per blob, frame indices and probabilities in place, padding rows zero
    assert int(t[:, 7].sum()) == n and not t[n:].any()
    merged = tdist.merge_tables(t)
    assert len(merged) == n and np.all(np.diff(merged[:, 0].astype(np.int64)) >= 0)
    assert set(np.unique(merged[:, 0]).tolist()) == set(range(4000, 4000 + B))
    assert np.array_equal(t[:n, tdist.HDR:].view(np.float32), p)
    seg.close()


def test_c5_detect_posture_crops_identify_full_record():
    # 4096x4096, 256 individuals, 256 classes, posture-normalised crops, the full per-blob record (BASELINE.json config 5)
    from oracle import tables as otables
    B, classes, MP, R = 4, 256, 256, 25
    frames, bg = synth.batch_torch("C5", 2, "cuda")
    frames = torch.cat([frames, frames.flip(0)]).contiguous()            # frame i == frame 3 - i
    H, W = frames.shape[1:]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=1024, max_pixels=1 << 18, max_runs=32768))
    seg.set_background(bg)
    st = weights.synthetic_state(classes, 99)
    seg.load_weights(weights.pack_blob(st, classes))
    seg.segment_device(frames.data_ptr(), B)
    res = seg.fetch()
    n = sum(len(r.blobs) for r in res)
    assert n == 256 * B
    outline = torch.zeros((n, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((n, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((n, R, 4), dtype=torch.float32, device="cuda")
    minfo = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    crops = torch.zeros((n, 80, 80), dtype=torch.uint8, device="cuda")
    probs = torch.zeros((n, classes), dtype=torch.float32, device="cuda")
    rowlen = otables.HDR_EX + classes + 3 * R
    table = torch.zeros((n, rowlen), dtype=torch.int32, device="cuda")
    seg.posture_device(n, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
    seg.midline_device(n, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    seg.crops_posture_device(crops.data_ptr(), n, minfo.data_ptr())
    seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
    seg.export_id_table_ex(probs.data_ptr(), n, classes, 0, table.data_ptr(), n, mid.data_ptr(), minfo.data_ptr(), R)
    seg.synchronize()
    pi = info.cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1); mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
    c = crops.cpu().numpy(); p = probs.cpu().numpy(); md = mid.cpu().numpy(); t = table.cpu().numpy().view(np.uint32)
    assert (pi["status"] == 0).all() and (mi["status"] == 0).all()       # every synthetic individual yields a midline
    assert np.all((mi["len"] > 25) & (mi["len"] < 50))                   # ellipses with an 36 px long axis
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)
    fr_host = frames[:1].cpu().numpy()[0]; bg_host = bg.cpu().numpy() if hasattr(bg, "cpu") else bg
    for f in range(2):                                                   # the same scene anywhere in the batch: the same midlines, crops, rows
        a, b = res[f], res[3 - f]
        sa, sb = slice(int(a.info["blob_begin"]), int(a.info["blob_begin"]) + 256), slice(int(b.info["blob_begin"]), int(b.info["blob_begin"]) + 256)
        assert np.array_equal(md[sa], md[sb]) and np.array_equal(c[sa], c[sb]) and np.array_equal(p[sa], p[sb])
    # frame 0 against the CPU restatements: posture-normalised crops bit-exact given the device's midline pose, network within 1e-4
    r0 = res[0]; b0 = int(r0.info["blob_begin"])
    for k in range(0, 256, 37):
        tr = oracle.midline_transform(mi[b0 + k]["angle"], mi[b0 + k]["offx"], mi[b0 + k]["offy"], False)
        want, _ = oracle.crop_normalized(fr_host, bg_host, r0.blobs[k], r0.runs, tr6=tr, midline_length=float(mi[b0 + k]["len"]))
        assert np.array_equal(c[b0 + k], want), k
    pick = b0 + np.arange(0, 256, 16)
    want, _ = cnn_oracle.predict(st, c[pick][..., None], threads=8)
    assert np.abs(p[pick] - want).max() <= 1e-4
    # the full record of frame 0 against its numpy restatement
    wt = otables.table_ex_from_blobs([r0], 0 + 0, p[b0:b0 + 256], classes, 256, md[b0:b0 + 256], mi[b0:b0 + 256], R)
    wt[:, 0] = t[b0:b0 + 256, 0]                                         # global frame index: frame 0 sits wherever it reserved its pool range
    assert np.array_equal(t[b0:b0 + 256], wt)
    assert set(np.unique(t[:, 0]).tolist()) == {0, 1, 2, 3}
    seg.close()


def test_c3_posture_and_midline_at_2048():
    # BASELINE.json config 3 at its full frame size: bg-sub + CCL + posture (outline -> midline -> normalised midline), every blob of two
    # frames against the CPU oracle (same comparison as tests/test_posture_gpu.py, which works on the 1280x720 frames)
    from test_posture_gpu import run_posture, compare, check_midline
    fr, bg = synth.batch("C3", 2)
    assert fr.shape[1:] == (2048, 2048)
    res, outline, segs, info = run_posture(fr, bg)
    assert all(len(r.blobs) == 100 for r in res)
    n, ties = compare(res, outline, segs, info, oracle.posture_params(max_points=512), max_ties=0.05)     # tie-aware tail rule, see there
    assert n == 200
    assert (info["status"] == 0).mean() > 0.95
    n_ok, total = check_midline(fr, bg, min_ok=0.9)
    assert total == 200


def test_c3_pipeline_like_bench():
    # `bench.py --config C3`: detect + posture + midline, no network, two lanes, detect issued before the host blocks on the tables
    from trex_amd.pipeline import Pipeline
    B = 64
    W, H, n_ind, _ = synth.CONFIGS["C3"]
    frames, bg = synth.batch_torch("C3", B, "cuda")
    pipe = Pipeline(W, H, n_ind, B, 100, bg, None, with_cnn=False, with_posture=True)
    got = []

    def on_batch(step, ln):
        n = int(ln.res.total_blobs)
        pi = ln.p_info[:n].cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1)
        mi = ln.p_minfo[:n].cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
        assert n == n_ind * B and (pi["status"] == 0).mean() > 0.95 and np.array_equal(pi["status"] == 0, mi["status"] != 1)
        ok = mi["status"] == 0
        assert ok.mean() > 0.9 and np.all((mi["len"][ok] > 25) & (mi["len"][ok] < 50))      # 36 px long ellipses
        info = capi._from_addr(ln.res.frames, ln.res.n_frames, capi.INFO_DTYPE)
        o = np.argsort(info["blob_begin"])
        got.append((pi["n_outline"].copy(), mi["len"].copy(), info["blob_begin"].copy()))

    pipe.run(3, frames.data_ptr(), on_batch=on_batch)
    # the same resident frames in every step: per frame the same outlines and midline lengths, whichever lane and pool position
    a_no, a_len, a_bb = got[0]
    for no, ln_, bb in got[1:]:
        for f in range(0, B, 7):
            assert np.array_equal(no[bb[f]:bb[f] + n_ind], a_no[a_bb[f]:a_bb[f] + n_ind])
            assert np.array_equal(ln_[bb[f]:bb[f] + n_ind], a_len[a_bb[f]:a_bb[f] + n_ind])
    pipe.close()
