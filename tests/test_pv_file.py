""".pv data section: per-frame LZO1X compression and the index table.
  * this library's LZO1X encoder (trex_amd/csrc/pvfile.cpp, host code) against the REFERENCE's decoder -- ProcessedVideo/lzo/minilzo.c compiled
    unmodified (oracle/ref.mk -> oracle/_ref/libminilzo.so): lzo1x_decompress must give every input back byte for byte (pv.cpp:316-340 is how
    stock TRex reads a compressed frame);
  * the oracle's decoder restatement (oracle/trex_pv.c) against the reference's encoder lzo1x_1_compress and decoder on the same inputs;
  * trexhip_pv_write_frames: the rule of pv.cpp:705-772 (>= 15000 bytes, kept if smaller), flag / sizes / stream layout, index table.
GPU (-m gpu): device packer -> host data section -> restated reader returns the segmentation of every frame."""
import numpy as np
import pytest
from oracle import oracle, lzo_ref
from trex_amd import capi, synth

needs_ref = pytest.mark.skipif(not lzo_ref.available(), reason="oracle/_ref/libminilzo.so not built (needs /root/reference)")


def corpus():
    rng = np.random.default_rng(7)
    out = [b"", b"a", b"abc", b"abcd" * 3, bytes(range(256)) * 3, b"\0" * 70000, bytes(rng.integers(0, 256, 5000, dtype=np.uint8)),
           bytes(rng.integers(0, 4, 40000, dtype=np.uint8)), b"x" * 17, b"xy" * 120, bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 200]
    # long literal runs (> 238 at the start, > 18 in the middle), matches at every distance class (<= 2048, <= 16384, <= 49151), long matches
    blk = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    out.append(blk + bytes(rng.integers(0, 256, 20000, dtype=np.uint8)) + blk + bytes(rng.integers(0, 256, 30000, dtype=np.uint8)) + blk[:1000])
    out.append(bytes(rng.integers(0, 256, 60000, dtype=np.uint8)) + blk * 3)
    # real frame bodies
    for cfg, seed in (("C2", 0), ("C2", 3)):
        fr, bg = synth.batch(cfg, 1, t0=seed)
        b, r, px = oracle.segment(fr[0], bg, oracle.make_params(fr.shape[2], fr.shape[1]))
        out.append(bytes(oracle.pv_serialize_v6(b, r, px, 12345)[1:]))
    for n in (1, 2, 3, 4, 5, 17, 18, 19, 237, 238, 239, 240, 300):       # literal-only streams around the length-encoding boundaries
        out.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
    return out


@needs_ref
def test_our_streams_through_the_reference_decoder():
    tot_in = tot_ours = tot_ref = 0
    for data in corpus():
        c = capi.lzo1x_compress(data)
        assert len(c) <= len(data) + len(data) // 16 + 64 + 3
        assert bytes(lzo_ref.decompress(c, len(data))) == data, len(data)
        tot_in += len(data); tot_ours += len(c); tot_ref += len(lzo_ref.compress(data))
    print("LZO1X: %d bytes in, ours %d, lzo1x_1_compress %d" % (tot_in, tot_ours, tot_ref))
    assert tot_ours <= 1.15 * tot_ref                    # not byte-identical to minilzo, but in the same class


@needs_ref
def test_the_decoder_restatement_equals_the_reference_decoder():
    for data in corpus():
        for c in (lzo_ref.compress(data), capi.lzo1x_compress(data)):
            got = oracle.lzo1x_decompress(c, len(data))
            assert got is not None and bytes(got) == data and bytes(lzo_ref.decompress(c, len(data))) == data
    # malformed streams are refused, not read out of bounds
    c = capi.lzo1x_compress(bytes(range(200)) * 20)
    assert oracle.lzo1x_decompress(c[:-1], 4000) is None and oracle.lzo1x_decompress(c, 100) is None


def test_encoder_round_trip_without_the_reference_tree():
    # (the same property through the restated decoder, so that the GPU box and a tree-less checkout still check the encoder)
    for data in corpus():
        got = oracle.lzo1x_decompress(capi.lzo1x_compress(data), len(data))
        assert got is not None and bytes(got) == data


def _frames(n, big):
    rng = np.random.default_rng(11)
    bodies, sizes = [], []
    for f in range(n):
        fr, bg = synth.random_scene(rng, 640, 480 if big else 64, density=0.2 if big else 0.05)
        b, r, px = oracle.segment(fr, bg, oracle.make_params(640, fr.shape[0]))
        bodies.append(oracle.pv_serialize_v6(b, r, px, 1000 + f))
    off = np.concatenate([[0], np.cumsum([len(b) for b in bodies])]).astype(np.uint64)
    return np.concatenate(bodies), off, bodies


@pytest.mark.parametrize("big", [False, True])
def test_data_section_layout_and_index_table(big):
    cat, off, bodies = _frames(4, big)
    out, idx = capi.pv_write_frames(cat, off, file_offset=4096)
    pos = 0
    for f, body in enumerate(bodies):
        assert idx[f] == 4096 + pos                                       # pv.cpp:1488-1496: the offset of the frame's compression_flag
        pack = bytes(body[1:])
        if out[pos] == 0:
            assert len(pack) < 15000 or len(capi.lzo1x_compress(pack)) + 8 >= len(pack)
            assert bytes(out[pos:pos + len(body)]) == bytes(body)
            used = len(body)
        else:
            assert len(pack) >= 15000                                     # pv.cpp:707-708
            csize, usize = (int(v) for v in out[pos + 1:pos + 9].view("<u4"))
            assert usize == len(pack) and csize + 8 < usize               # pv.cpp:758
            assert bytes(oracle.lzo1x_decompress(out[pos + 9:pos + 9 + csize], usize)) == pack
            if lzo_ref.available():
                assert bytes(lzo_ref.decompress(out[pos + 9:pos + 9 + csize], usize)) == pack
            used = 9 + csize
        # the restated Frame::read_from takes either form and returns the same lines and pixels
        a = oracle.pv_read_v6(out[pos:pos + used]); b = oracle.pv_read_v6(body)
        assert a[0] == used and a[1] == b[1] and a[2].tobytes() == b[2].tobytes() and a[3].tobytes() == b[3].tobytes()
        pos += used
    assert pos == len(out)
    assert (out[[int(i) - 4096 for i in idx]] == 1).all() == big          # the large frames compress, the small ones are stored as they are
    # always_compress (what the rgb8 encoding does): small frames are tried too
    out2, idx2 = capi.pv_write_frames(cat, off, always_compress=True)
    assert len(out2) <= len(out) and idx2[0] == 0


@pytest.mark.gpu
def test_device_pack_to_data_section_and_back():
    import torch
    fr, bg = synth.batch("C4", 3)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    ts = np.arange(n, dtype=np.uint64) * 40000
    cap = sum(11 + 4 * len(r.blobs) + 4 * len(r.runs) + len(r.pixels) for r in res) + 64
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    seg.pack_frames_v6_device(out.data_ptr(), out.numel(), off.data_ptr(), ts)
    seg.synchronize()
    o = off.cpu().numpy().astype(np.uint64); bodies = out.cpu().numpy()[:int(o[n])]
    data, idx = capi.pv_write_frames(bodies, o, file_offset=123)
    assert (data[[int(i) - 123 for i in idx]] == 1).all()                 # 100 individuals per 2048x2048 frame: ~40 KB packs, all compressed
    assert len(data) < 0.8 * len(bodies)
    ends = list(idx[1:] - 123) + [len(data)]
    for f in range(n):
        used, t, rr, pp, br, bp = oracle.pv_read_v6(data[int(idx[f]) - 123:int(ends[f])])
        assert used == int(ends[f]) - (int(idx[f]) - 123) and t == ts[f]
        assert rr.tobytes() == res[f].runs.tobytes() and pp.tobytes() == res[f].pixels.tobytes()
        assert np.array_equal(br, res[f].blobs["n_runs"]) and np.array_equal(bp, res[f].blobs["n_pixels"])
    seg.close()
