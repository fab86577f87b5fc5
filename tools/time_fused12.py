"""conv1 inside conv2 (cnn_fused12.h) against the two-kernel chain (TREXHIP_CONV_GEOM bit 28): identical probabilities, time per 25600 crops (dev tool, round 4)"""
import time, numpy as np, torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
st = weights.synthetic_state(100, 31)
crops = torch.from_numpy(np.tile(weights.synthetic_crops(256, 3), (N // 256 + 1, 1, 1, 1))[:N]).cuda()
out = {}
for name, geom in (("two kernels", 1 << 28), ("fused", (1 << 29) if os.environ.get("TREXHIP_F12_RS") else ((1 << 30) if os.environ.get("TREXHIP_F12_OLD") else 0))):
    os.environ["TREXHIP_CONV_GEOM"] = str(geom)
    seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100))
    seg.set_identity_precision(3)
    probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda")
    for _ in range(3): seg.identify_device(crops.data_ptr(), N, probs.data_ptr())
    seg.synchronize()
    seg.profile_enable(True); seg.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10): seg.identify_device(crops.data_ptr(), N, probs.data_ptr())
    seg.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    c2 = seg.profile_read(capi.STAGE_CONV2); c3 = seg.profile_read(capi.STAGE_CONV3); ca = seg.profile_read(capi.STAGE_CNN_ALL)
    out[name] = probs.cpu().numpy()
    print("%-12s identify %.3f ms per %d crops; stage conv2 %.3f conv3 %.3f cnn_all %.3f ms" % (name, dt, N, c2[0] / c2[1], c3[0] / c3[1], ca[0] / ca[1]), flush=True)
    seg.close()
d = np.abs(out["fused"] - out["two kernels"]).max()
print("max |dp| fused vs two kernels: %g (%s)" % (d, "bit-identical" if out["fused"].tobytes() == out["two kernels"].tobytes() else "DIFFERENT"))
