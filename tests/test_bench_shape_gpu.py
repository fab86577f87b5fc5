"""The benchmarked shape, driven exactly like bench.py (same trex_amd.pipeline.Pipeline object): config C4 -- 2048x2048, 100 individuals,
256 frames resident per step, two software-pipelined contexts, detect on a high-priority stream, 3 steps.  Every frame of every step is
checked through size-independent properties; a sample of frames is compared bit for bit with the CPU oracle, a sample of crops with the
CPU restatement of the network (1e-4 on softmax, BASELINE.json), and the per-blob ID table of every step with the tables it was built from."""
import numpy as np
import pytest
import torch
from oracle import oracle, cnn_oracle
from trex_amd import capi, synth, weights, dist as tdist
from trex_amd.pipeline import Pipeline

pytestmark = pytest.mark.gpu


def _sorted_lines(info, blobs, runs):
    """pv.cpp:505-508: the lines of a blob are strictly ordered by (y, x0) and do not overlap."""
    key = runs["y"].astype(np.int64) * 65536 + runs["x0"]
    d = np.diff(key) > 0
    starts = np.zeros(len(runs), bool)
    for fi in info:                                                     # run_begin of a blob is frame-relative
        b = blobs[int(fi["blob_begin"]):int(fi["blob_begin"]) + int(fi["n_blobs"])]
        starts[int(fi["run_begin"]) + b["run_begin"].astype(np.int64)] = True
    ok = d | starts[1:]
    same_row = (runs["y"][1:] == runs["y"][:-1]) & ~starts[1:]
    no_overlap = ~same_row | (runs["x0"][1:].astype(np.int64) > runs["x1"][:-1])
    return bool(ok.all() and no_overlap.all())


def test_c4_256_frames_two_lanes_three_steps():
    B, classes, n_ind, steps = 256, 100, 100, 3
    W, H = synth.CONFIGS["C4"][:2]
    frames, bg = synth.batch_torch("C4", B, "cuda")
    st = weights.synthetic_state(classes, 4242)
    pipe = Pipeline(W, H, n_ind, B, classes, bg, weights.pack_blob(st, classes))
    assert len(pipe.lanes) == 2 and pipe.lanes[0].hi is not None        # bench.py's default schedule
    thr_count = ((frames.to(torch.int16) - bg.to(torch.int16)).abs() > 15).sum(dim=(1, 2)).cpu().numpy()
    bg_host = bg.cpu().numpy()
    sample_frames = [0, 1, 37, 100, 128, 199, 254, 255]
    seen = {}

    def on_batch(step, ln):
        seg = ln.seg
        res = ln.res
        n = int(res.total_blobs)
        assert n == n_ind * B
        info = capi._from_addr(res.frames, res.n_frames, capi.INFO_DTYPE)
        blobs = capi._from_addr(res.blobs, res.total_blobs, capi.BLOB_DTYPE)
        runs = capi._from_addr(res.runs, res.total_runs, capi.RUN_DTYPE)
        pixels = capi._from_addr(res.pixels, res.total_pixels, np.dtype(np.uint8))
        # properties of every frame
        assert (info["flags"] == 0).all() and (info["n_blobs"] == n_ind).all()
        npx = np.add.reduceat(blobs["n_pixels"].astype(np.int64), np.sort(info["blob_begin"].astype(np.int64)))
        order = np.argsort(info["blob_begin"])
        assert np.array_equal(npx, thr_count[order])                     # every thresholded pixel is in exactly one blob
        assert np.array_equal(info["n_pixels"], thr_count) and int(info["n_pixels"].sum()) == len(pixels)
        assert _sorted_lines(info, blobs, runs)
        # a sample of frames bit for bit against the oracle
        if step == 0:
            for f in sample_frames:
                fi = info[f]
                ob, orr, opx = oracle.segment(frames[f].cpu().numpy(), bg_host, oracle.make_params(W, H))
                b = blobs[int(fi["blob_begin"]):int(fi["blob_begin"]) + n_ind]
                assert b.tobytes() == ob.tobytes(), f
                assert runs[int(fi["run_begin"]):int(fi["run_begin"]) + int(fi["n_runs"])].tobytes() == orr.tobytes(), f
                assert pixels[int(fi["pix_begin"]):int(fi["pix_begin"]) + int(fi["n_pixels"])].tobytes() == opx.tobytes(), f
        # crops hold their blob's grey values; a sample through the CPU network
        crops = ln.crops[:n].cpu().numpy()
        probs = ln.probs[:n].cpu().numpy()
        assert np.array_equal(crops.reshape(n, -1).astype(np.int64).sum(1), blobs["sp"].astype(np.int64))
        assert np.allclose(probs.sum(1), 1.0, atol=1e-5) and (probs >= 0).all()
        pick = np.linspace(0, n - 1, 64 if step == 0 else 8).astype(int)
        want, _ = cnn_oracle.predict(st, crops[pick][..., None], threads=8)
        assert np.abs(probs[pick] - want).max() <= 1e-4
        # the ID table that went to the host: one valid row per blob, in pooled order, header from the blob table, probabilities in place
        t = ln.table_host.numpy().view(np.uint32)
        assert int(t[:, 7].sum()) == n and not t[n:].any()
        fr_of_blob = np.empty(n, np.int64)
        for f, fi in enumerate(info):
            fr_of_blob[int(fi["blob_begin"]):int(fi["blob_begin"]) + int(fi["n_blobs"])] = f
        assert np.array_equal(t[:n, 0].astype(np.int64), step * B + fr_of_blob)
        assert np.array_equal(t[:n, 1], blobs["bid"]) and np.array_equal(t[:n, 2], blobs["n_pixels"])
        assert np.array_equal(t[:n, tdist.HDR:].view(np.float32), probs)
        merged = tdist.merge_tables(t)
        assert len(merged) == n and set(np.unique(merged[:, 0]).tolist()) == set(range(step * B, step * B + B))
        # the same resident frames every step: both lanes must produce the same blobs and the same probabilities per frame
        per_frame = {f: (blobs[int(fi["blob_begin"]):int(fi["blob_begin"]) + n_ind].copy(), probs[int(fi["blob_begin"]):int(fi["blob_begin"]) + n_ind].copy())
                     for f, fi in enumerate(info) if f in sample_frames or f % 16 == 0}
        if not seen:
            seen.update(per_frame)
        else:
            for f, (b, p) in per_frame.items():
                b0, p0 = seen[f]
                for name in ("n_pixels", "bid", "m10", "m01", "m11", "sp", "spx"):
                    assert np.array_equal(b[name], b0[name]), (step, f, name)
                assert np.array_equal(p, p0), (step, f)

    pipe.run(steps, frames.data_ptr(), on_batch=on_batch)
    pipe.close()


@pytest.mark.parametrize("name,config,B,kw", [
    ("C5 as benchmarked", "C5", 64, dict(normalize="posture")),
    ("C4 posture-normalised crops", "C4", 32, dict(normalize="posture")),
    ("C4 rgb8 encoding", "C4", 32, dict(rgb=True)),
])
def test_other_benchmarked_pipelines(name, config, B, kw):
    """The other workloads bench.py reports (its `secondary` runs) through the same two-lane Pipeline object at the batch sizes it uses:
    C5 = 4096x4096, 256 individuals, 256 classes, posture -> midline -> posture-normalised crops -> network -> full per-blob record at
    B = 64; posture-normalised crops and the rgb8 encoding at C4.  Every frame through properties, samples against the CPU restatements."""
    from oracle import tables as otables
    W, H, n_ind, _ = synth.CONFIGS[config]
    classes = 256 if config == "C5" else 100
    rgb = bool(kw.get("rgb"))
    posture = kw.get("normalize") == "posture"
    frames, bg = synth.batch_torch(config, B, "cuda")
    src = torch.stack([frames, frames, frames, torch.full_like(frames, 255)], dim=-1).contiguous() if rgb else frames
    st = weights.synthetic_state(classes, 4242, channels=3 if rgb else 1)
    pipe = Pipeline(W, H, n_ind, B, classes, bg, weights.pack_blob(st, classes, channels=3 if rgb else 1), **kw)
    assert len(pipe.lanes) == 2
    bg_host = bg.cpu().numpy()
    fr0 = frames[0].cpu().numpy()
    done = []

    def on_batch(step, ln):
        res = ln.res
        n = int(res.total_blobs)
        assert n == n_ind * B
        info = capi._from_addr(res.frames, res.n_frames, capi.INFO_DTYPE)
        blobs = capi._from_addr(res.blobs, res.total_blobs, capi.BLOB_DTYPE)
        runs = capi._from_addr(res.runs, res.total_runs, capi.RUN_DTYPE)
        assert (info["flags"] == 0).all() and (info["n_blobs"] == n_ind).all()
        assert _sorted_lines(info, blobs, runs)
        crops = ln.crops[:n].cpu().numpy()
        probs = ln.probs[:n].cpu().numpy()
        assert np.allclose(probs.sum(1), 1.0, atol=1e-5) and (probs >= 0).all()
        pick = np.linspace(0, n - 1, 24 if step == 0 else 6).astype(int)
        want, _ = cnn_oracle.predict(st, crops[pick] if rgb else crops[pick][..., None], threads=8)
        assert np.abs(probs[pick] - want).max() <= 1e-4
        b0 = int(info[0]["blob_begin"])
        fb = blobs[b0:b0 + n_ind]
        fruns = runs[int(info[0]["run_begin"]):int(info[0]["run_begin"]) + int(info[0]["n_runs"])]
        if posture:
            mi = ln.p_minfo[:n].cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
            pi = ln.p_info[:n].cpu().numpy().view(capi.POSTURE_INFO_DTYPE).reshape(-1)
            assert (pi["status"] == 0).mean() > 0.97 and ((mi["status"] == 0) == (pi["status"] == 0)).all()
            if step == 0:                       # frame 0: posture-normalised crops bit for bit given the device's midline pose
                for k in range(0, n_ind, max(1, n_ind // 8)):
                    if mi[b0 + k]["status"] != 0:
                        continue
                    tr = oracle.midline_transform(mi[b0 + k]["angle"], mi[b0 + k]["offx"], mi[b0 + k]["offy"], False)
                    want_c, _ = oracle.crop_normalized(fr0, bg_host, fb[k], fruns, tr6=tr, midline_length=float(mi[b0 + k]["len"]))
                    assert np.array_equal(crops[b0 + k], want_c), k
            t = ln.table_host.numpy().view(np.uint32)
            assert int(t[:, 7].sum()) == n and not t[n:].any()
            assert np.array_equal(t[:n, 1], blobs["bid"]) and np.array_equal(t[:n, 2], blobs["n_pixels"])
            assert np.array_equal(t[:n, otables.HDR_EX:otables.HDR_EX + classes].view(np.float32), probs)
        else:
            if rgb:                             # raw pixels painted: every channel of a crop sums to its blob's pixel bytes of that channel
                px = capi._from_addr(res.pixels, res.total_pixels * 3, np.dtype(np.uint8)).reshape(-1, 3)
                k = b0
                pb = int(info[0]["pix_begin"]) + int(blobs[k]["pix_begin"])
                assert np.array_equal(crops[k].reshape(-1, 3).astype(np.int64).sum(0), px[pb:pb + int(blobs[k]["n_pixels"])].astype(np.int64).sum(0))
            t = ln.table_host.numpy().view(np.uint32)
            assert int(t[:, 7].sum()) == n and np.array_equal(t[:n, tdist.HDR:].view(np.float32), probs)
        done.append(step)

    pipe.run(3, src.data_ptr(), on_batch=on_batch)
    pipe.close()
    assert sorted(done) == [0, 1, 2]


def test_c4_pipeline_ten_steps_reproduce_bit_for_bit():
    """The stress leg of round 5: ten pipelined C4 steps (both lanes, the same resident frames).  Every step's probabilities -- all 25600 rows,
    ordered by frame -- must equal the first step's bit for bit, and the range guard must stay quiet: a schedule hazard in the large-batch
    kernels (round 4's accumulator reads, round 5's v_max3 on MFMA results) shows up as a step that differs."""
    B, classes, n_ind, steps = 256, 100, 100, 10
    W, H = synth.CONFIGS["C4"][:2]
    frames, bg = synth.batch_torch("C4", B, "cuda")
    st = weights.synthetic_state(classes, 4242)
    pipe = Pipeline(W, H, n_ind, B, classes, bg, weights.pack_blob(st, classes))
    first = {}

    def on_batch(step, ln):
        res = ln.res
        n = int(res.total_blobs)
        assert n == n_ind * B
        assert ln.seg.guard_stats() == (0, False), step
        info = capi._from_addr(res.frames, res.n_frames, capi.INFO_DTYPE)
        blobs = capi._from_addr(res.blobs, res.total_blobs, capi.BLOB_DTYPE)
        probs = ln.probs[:n].cpu().numpy()
        by_frame = np.stack([probs[int(fi["blob_begin"]):int(fi["blob_begin"]) + n_ind] for fi in info])
        # the blob records too (the labelling workgroup gathers its own frame's blobs behind a fence + barrier: every step must agree)
        rec = np.stack([np.stack([blobs[name][int(fi["blob_begin"]):int(fi["blob_begin"]) + n_ind].astype(np.int64)
                                  for name in ("n_pixels", "bid", "m10", "m01", "m11", "sp", "spx", "x0", "y0", "x1", "y1")]) for fi in info])
        if not first:
            first["p"], first["b"] = by_frame, rec
        else:
            assert by_frame.tobytes() == first["p"].tobytes(), (step, float(np.abs(by_frame - first["p"]).max()))
            assert np.array_equal(rec, first["b"]), step

    pipe.run(steps, frames.data_ptr(), on_batch=on_batch)
    pipe.close()
