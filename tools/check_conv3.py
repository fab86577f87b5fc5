"""dev: conv3 variants (TREXHIP_CONV_GEOM bits 24..27, dev build) against the default kernel on the same crops: largest difference of the probabilities
over several batch sizes (partial passes, one pass, many), and the time of the CONV3 stage on 25600 crops"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
st = weights.synthetic_state(100, 31)
blob = weights.pack_blob(st, 100)
variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
REF = sys.argv[2] if len(sys.argv) > 2 else None        # an .npz written by another build of the library (same script, "save:<file>"), or "save:<file>"
base = weights.synthetic_crops(100, 3)
sizes = [1, 7, 100, 257, 3201, 25600]
rng = np.random.default_rng(5)
crops = {n: torch.from_numpy(np.tile(base, ((n + 99) // 100, 1, 1, 1))[:n].copy()).cuda() for n in sizes}
for n in sizes:      # not 256 copies of the same 100 crops: a wrong tile -> output mapping has to show
    c = crops[n].cpu().numpy(); c = np.roll(c, rng.integers(0, 80, size=1)[0], axis=2) if n > 100 else c
    idx = rng.permutation(n); crops[n] = torch.from_numpy(np.ascontiguousarray(c[idx])).cuda()
ref = {}
if REF and not REF.startswith("save:"):
    z = np.load(REF); ref = {n: z[str(n)] for n in sizes}
for v in variants:
    os.environ["TREXHIP_CONV_GEOM"] = str(v << 24)
    seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1)); seg.load_weights(blob)
    worst = 0.0
    for n in sizes:
        probs = torch.full((n, 100), -1.0, dtype=torch.float32, device="cuda")
        seg.identify_device(crops[n].data_ptr(), n, probs.data_ptr()); seg.synchronize()
        p = probs.cpu().numpy()
        if n not in ref: ref[n] = p
        else: worst = max(worst, float(np.abs(p - ref[n]).max()))
        assert np.isfinite(p).all() and abs(p.sum(1) - 1).max() < 1e-4, (v, n)
    n = 25600
    probs = torch.zeros((n, 100), dtype=torch.float32, device="cuda")
    seg.profile_enable(True)
    for _ in range(2): seg.identify_device(crops[n].data_ptr(), n, probs.data_ptr())
    seg.synchronize(); seg.profile_reset()
    for _ in range(6): seg.identify_device(crops[n].data_ptr(), n, probs.data_ptr())
    seg.synchronize()
    ms, cnt = seg.profile_read(capi.STAGE_CONV3); ms2, cnt2 = seg.profile_read(capi.STAGE_CNN_ALL)
    print("variant %2d  max |dp| vs variant %d: %.3g   CONV3 %.3f ms  CNN_ALL %.3f ms" % (v, variants[0], worst, ms / max(cnt, 1), ms2 / max(cnt2, 1)), flush=True)
    del seg
if REF and REF.startswith("save:"): np.savez(REF[5:], **{str(n): ref[n] for n in sizes})
