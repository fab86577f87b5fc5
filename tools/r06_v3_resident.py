"""Round 6, VERDICT item 1(b): does conv3 run faster per crop when its operand image V3 never leaves the 256 MB memory-side cache?
The identity network over 25600 crops in CHUNKS (V3 = 205 KB per crop: 1000 crops = 205 MB), the role-split conv1 + conv2 kernel forced
for every chunk size (TREXHIP_CONV_GEOM bit 29), kernel time per stage from the library's HIP-event stage timers (launch gaps excluded).
   python tools/r06_v3_resident.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TREXHIP_CONV_GEOM"] = str(1 << 29)
import numpy as np
import torch
from trex_amd import capi, weights

st = weights.synthetic_state(100, 31)
seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1), stream=None)
seg.load_weights(weights.pack_blob(st, 100))
N = 25600
crops = torch.from_numpy(np.tile(weights.synthetic_crops(100, 3), (N // 100, 1, 1, 1))).cuda()
probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
seg.set_identity_precision(3)
ref = None
print("# chunk: wall ms per 25600 crops | stage timers summed over the chunks: conv1+conv2, conv3, whole chain (ms per 25600 crops)", flush=True)
for chunk in (25600, 12800, 6400, 3200, 2000, 1600, 1200, 1000, 800, 640):
    def run():
        for o in range(0, N, chunk):
            n = min(chunk, N - o)
            seg.identify_device(crops.data_ptr() + o * 6400, n, probs.data_ptr() + o * 400)
    for _ in range(3):
        run()
    seg.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        run()
    seg.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    seg.profile_enable(True); seg.profile_reset()
    for _ in range(3):
        run()
    seg.synchronize()
    c2 = seg.profile_read(capi.STAGE_CONV2)[0] / 3; c3 = seg.profile_read(capi.STAGE_CONV3)[0] / 3; ca = seg.profile_read(capi.STAGE_CNN_ALL)[0] / 3
    seg.profile_enable(False)
    p = probs.cpu().numpy()
    if ref is None:
        ref = p
    print("chunk %6d (V3 %6.0f MB): wall %.3f ms | conv12 %.3f  conv3 %.3f  chain %.3f | max |dp| vs one shot %.2g" % (chunk, chunk * 204800 / 1e6, dt, c2, c3, ca, float(np.abs(p - ref).max())), flush=True)
seg.close()
