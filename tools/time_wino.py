"""dev: time conv3 (stage CONV3) of identify_device on 25600 crops for TREXHIP_CONV_GEOM debug variants"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
st = weights.synthetic_state(100, 31)
seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100))
n = 25600
crops = torch.from_numpy(np.tile(weights.synthetic_crops(100, 3), (256, 1, 1, 1))).cuda()
probs = torch.zeros((n, 100), dtype=torch.float32, device="cuda")
seg.profile_enable(True)
for _ in range(2): seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
seg.synchronize(); seg.profile_reset()
for _ in range(5): seg.identify_device(crops.data_ptr(), n, probs.data_ptr())
seg.synchronize()
for nm in ("CONV2", "CONV3", "CNN_ALL"):
    ms, cnt = seg.profile_read(getattr(capi, "STAGE_" + nm)); print(os.environ.get("TREXHIP_CONV_GEOM", "0"), nm, "%.3f ms" % (ms / max(cnt, 1)))
