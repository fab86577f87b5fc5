import numpy as np
import pytest
from oracle import tables as otables
import torch
from trex_amd import capi, synth, dist as tdist

pytestmark = pytest.mark.gpu


def test_id_table_kernel_matches_host_builder():
    fr, bg = synth.batch("C2", 3)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    total = sum(len(r.blobs) for r in res)
    C_ = 7
    probs = torch.rand((total, C_), device="cuda")
    table = torch.full((total + 5, tdist.HDR + C_), -1, dtype=torch.int32, device="cuda")
    seg.export_id_table(probs.data_ptr(), total, C_, 1000, table.data_ptr(), total + 5)
    seg.synchronize()
    # pooled order: frames may have reserved their pool ranges in any order
    order = np.argsort([int(r.info["blob_begin"]) for r in res])
    want = np.zeros((total + 5, tdist.HDR + C_), np.uint32)
    pr = probs.cpu().numpy()
    for f in order:
        r = res[f]
        bb = int(r.info["blob_begin"])
        t = otables.table_from_blobs([r], 1000 + f, pr[bb:bb + len(r.blobs)], C_, len(r.blobs))
        want[bb:bb + len(r.blobs)] = t
    got = table.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    seg.close()


def test_full_record_kernel_matches_host_builder():
    # bid, bbox, npx, centroid, second moments, probabilities and the normalised midline in one fixed-size row
    fr, bg = synth.batch("C2", 2)
    n, H, W = fr.shape
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=n))
    seg.set_background(bg)
    d = torch.from_numpy(fr).cuda()
    seg.segment_device(d.data_ptr(), n)
    res = seg.fetch()
    total = sum(len(r.blobs) for r in res)
    C_, R, MP = 5, 25, 256
    probs = torch.rand((total, C_), device="cuda")
    outline = torch.zeros((total, MP, 2), dtype=torch.float32, device="cuda"); segs = torch.zeros((total, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
    info = torch.zeros((total, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((total, R, 4), dtype=torch.float32, device="cuda")
    minfo = torch.zeros((total, 8), dtype=torch.int32, device="cuda")
    seg.posture_device(total, outline.data_ptr(), segs.data_ptr(), info.data_ptr(), max_points=MP)
    seg.midline_device(total, MP, info.data_ptr(), segs.data_ptr(), mid.data_ptr(), minfo.data_ptr())
    rowlen = tdist.HDR_EX + C_ + 3 * R
    order = np.argsort([int(r.info["blob_begin"]) for r in res])
    pr = probs.cpu().numpy()
    for with_midline in (True, False):
        table = torch.full((total + 3, rowlen), -1, dtype=torch.int32, device="cuda")
        if with_midline:
            seg.export_id_table_ex(probs.data_ptr(), total, C_, 500, table.data_ptr(), total + 3, mid.data_ptr(), minfo.data_ptr(), R)
        else:
            seg.export_id_table_ex(probs.data_ptr(), total, C_, 500, table.data_ptr(), total + 3, midline_resolution=R)
        seg.synchronize()
        mi = minfo.cpu().numpy().view(capi.MIDLINE_INFO_DTYPE).reshape(-1)
        md = mid.cpu().numpy()
        assert (mi["status"] == 0).sum() > total // 2
        want = np.zeros((total + 3, rowlen), np.uint32)
        for f in order:
            r = res[f]
            bb = int(r.info["blob_begin"]); k = len(r.blobs)
            want[bb:bb + k] = otables.table_ex_from_blobs([r], 500 + f, pr[bb:bb + k], C_, k, md[bb:bb + k] if with_midline else None,
                                                        mi[bb:bb + k] if with_midline else None, R)
        got = table.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), with_midline
    seg.close()
