"""Does running the identity network over CHUNKS of the batch (so that one chunk's operand images V2 / V3 / act3 fit the 256 MB
memory-side cache between producer and consumer) beat one launch chain over the whole batch?  (dev tool, round 4)"""
import time, numpy as np, torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
st = weights.synthetic_state(100, 31)
seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100))
N = 25600
crops = torch.from_numpy(np.tile(weights.synthetic_crops(100, 3), (N // 100, 1, 1, 1))).cuda()
probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda")
seg.set_identity_precision(3)
ref = None
for chunk in (25600, 12800, 6400, 3200, 1600, 1024, 800, 512, 256):
    def run():
        for o in range(0, N, chunk):
            n = min(chunk, N - o)
            seg.identify_device(crops.data_ptr() + o * 6400, n, probs.data_ptr() + o * 400)
    for _ in range(4): run()
    seg.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    seg.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    p = probs.cpu().numpy()
    if ref is None: ref = p
    print("chunk %6d: %.3f ms per %d crops, max |dp| vs one shot %.2g" % (chunk, dt, N, float(np.abs(p - ref).max())), flush=True)
seg.close()
