"""The as-deployed boundary: pageable host tiles through trexhip_segment / trexhip_segment_color (upload.hip).  Results must equal the
device-resident entry points bit for bit, and the two legs of the upload (host copy into the pinned ring, DMA) must overlap."""
import time
import numpy as np
import pytest
import torch
from trex_amd import capi, synth

pytestmark = pytest.mark.gpu


def tables(seg):
    r = seg.fetch()
    return [(x.blobs.tobytes(), x.runs.tobytes(), x.pixels.tobytes()) for x in r]


@pytest.mark.parametrize("n", [1, 5, 19])
def test_host_paths_equal_device_paths(n):
    fr, bg = synth.batch("C2", n)
    H, W = fr.shape[1:]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n); want = tables(seg)
    seg.segment_host([f for f in fr]); assert tables(seg) == want
    # (a row pitch larger than the width -- tiles that are views into wider buffers -- goes through the C++ adapter test: the ctypes binding
    # hands over contiguous frames only)
    for ch in (3, 4):
        col = np.repeat(fr[..., None], ch, axis=3)
        col[..., 1] = fr // 2                                           # not a gray image: cvtColor matters
        dc = torch.from_numpy(col).cuda()
        seg.segment_color_device(dc.data_ptr(), n, ch); want_c = tables(seg)
        seg.segment_color_host([c for c in col]); assert tables(seg) == want_c      # default: reduced to gray by the upload threads (hostcvt.cpp)
        for cc in (0, 1, ch - 1):                                                   # color_channel picks (BackgroundSubtraction.cpp:163-170)
            seg.segment_color_device(dc.data_ptr(), n, ch, cc); want_cc = tables(seg)
            seg.segment_color_host([c for c in col], cc); assert tables(seg) == want_cc
    seg.close()
    # the same tiles uploaded in colour and reduced on the device (device_color_reduce = 1): identical tables
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, device_color_reduce=1))
    seg.set_background(bg)
    col = np.repeat(fr[..., None], 4, axis=3); col[..., 1] = fr // 2
    dc = torch.from_numpy(col).cuda()
    seg.segment_color_device(dc.data_ptr(), n, 4); want_c = tables(seg)
    seg.segment_color_host([c for c in col]); assert tables(seg) == want_c
    seg.close()


def test_host_side_gray_reduction_cuts_the_transfer():
    # BGRA tiles with a gray pixel encoding: the upload threads write one byte per pixel into the pinned ring, a quarter of the bytes
    # cross PCIe -- per frame the DMA leg must take well under half of what the colour upload takes
    B = 32
    frames, bg = synth.batch_torch("C4", B, "cuda")
    H, W = frames.shape[1:]
    col = [np.ascontiguousarray(np.repeat(frames[i].cpu().numpy()[..., None], 4, 2)) for i in range(B)]
    dma = {}
    for dev in (0, 1):
        seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=400, max_pixels=1 << 18, max_runs=32768, device_color_reduce=dev))
        seg.set_background(bg)
        seg.segment_color_host(col); seg.synchronize()
        res = tables(seg)
        seg.profile_reset()
        seg.segment_color_host(col); seg.synchronize()
        ms, cnt = seg.profile_read(capi.STAGE_UPLOAD_DMA)
        assert cnt == B
        dma[dev] = (ms / cnt, res)
        seg.close()
    assert dma[0][1] == dma[1][1]
    print(f"DMA per frame: host-reduced {dma[0][0]:.3f} ms, colour upload {dma[1][0]:.3f} ms")
    assert dma[0][0] < 0.5 * dma[1][0]


def test_upload_legs_overlap():
    B = 48
    frames, bg = synth.batch_torch("C4", B, "cuda")
    H, W = frames.shape[1:]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=400, max_pixels=1 << 18, max_runs=32768, device_color_reduce=1))
    seg.set_background(bg)
    col = [np.ascontiguousarray(np.repeat(frames[i].cpu().numpy()[..., None], 4, 2)) for i in range(B)]      # pageable BGRA tiles
    seg.segment_color_host(col); seg.synchronize()                      # allocations, thread pool
    seg.profile_reset()
    t0 = time.perf_counter()
    seg.segment_color_host(col); seg.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    copy_ms, cn = seg.profile_read(capi.STAGE_UPLOAD_COPY)
    dma_ms, dn = seg.profile_read(capi.STAGE_UPLOAD_DMA)
    assert cn == B and dn == B
    gbps = B * W * H * 4 / (wall * 1e-3) / 1e9
    print(f"wall {wall:.1f} ms  copy {copy_ms:.1f} ms  dma {dma_ms:.1f} ms  {gbps:.1f} GB/s")
    assert wall < copy_ms + dma_ms - 0.4 * min(copy_ms, dma_ms)        # the smaller leg is (mostly) hidden under the larger one
    assert gbps > 20.0                                                  # PCIe Gen5 x16: 63 GB/s spec; a serial copy -> DMA path stays below ~25
    seg.close()


def test_row_pitch_larger_than_the_width():
    # tiles that are views into wider buffers: stride > width * channels (the upload threads copy / reduce row by row)
    import ctypes as C
    n = 3
    fr, bg = synth.batch("C2", n)
    H, W = fr.shape[1:]
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n); want = tables(seg)
    L = capi.lib()
    wide = np.full((n, H, W + 40), 7, np.uint8); wide[:, :, :W] = fr
    ptrs = (C.c_void_p * n)(*[wide[i].ctypes.data for i in range(n)])
    assert L.trexhip_segment(seg.handle, ptrs, W + 40, n) == 0
    assert tables(seg) == want
    for ch in (3, 4):
        col = np.repeat(fr[..., None], ch, axis=3); col[..., 1] = fr // 2
        dc = torch.from_numpy(col).cuda()
        seg.segment_color_device(dc.data_ptr(), n, ch); want_c = tables(seg)
        widec = np.full((n, H, W + 24, ch), 9, np.uint8); widec[:, :, :W] = col
        ptrs = (C.c_void_p * n)(*[widec[i].ctypes.data for i in range(n)])
        assert L.trexhip_segment_color(seg.handle, ptrs, (W + 24) * ch, n, ch, -1) == 0
        assert tables(seg) == want_c
    seg.close()
