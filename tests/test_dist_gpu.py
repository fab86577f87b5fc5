import numpy as np
import pytest
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
        t = tdist.table_from_blobs([r], 1000 + f, pr[bb:bb + len(r.blobs)], C_, len(r.blobs))
        want[bb:bb + len(r.blobs)] = t
    got = table.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    seg.close()
