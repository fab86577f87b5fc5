"""Times trexhip_identify_device on 6400 synthetic crops for 1 and 3 input channels (dev tool)."""
import time, numpy as np, torch, os, sys
from trex_amd import capi, weights
for ch in (1, 3):
    st = weights.synthetic_state(100, 31, channels=ch)
    seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100, channels=ch))
    n = 6400
    crops = torch.from_numpy(np.tile(weights.synthetic_crops(100, 3, channels=ch), (64, 1, 1, 1))).cuda()
    probs = torch.zeros((n, 100), dtype=torch.float32, device="cuda")
    for mode in (3, 1):
        seg.set_identity_precision(mode)
        for _ in range(3): seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
        seg.synchronize(); t0 = time.perf_counter()
        for _ in range(10): seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
        seg.synchronize(); print("channels", ch, "mode", mode, "geom", os.environ.get("TREXHIP_CONV_GEOM", "0"), "identify ms %.2f" % ((time.perf_counter() - t0) / 10 * 1e3))
    seg.close()
