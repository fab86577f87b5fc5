"""dev tool (tools/build_dev.sh build, TREXHIP_F12_DBG=128): cycles per stage of the role-split conv1 + conv2 kernel (cnn_fused12rs.h), lane 0 of the
consumer wave 0 and the producer wave 4 of workgroup 0, per round.
   gpurun -- 'TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so TREXHIP_F12_DBG=128 python tools/f12rs_stamps.py'"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
st = weights.synthetic_state(100, 31)
crops = torch.from_numpy(np.tile(weights.synthetic_crops(256, 3), (N // 256 + 1, 1, 1, 1))[:N]).cuda()
seg = capi.Segmenter(capi.default_params(256, 256, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100)); seg.set_identity_precision(3)
probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda")
L = capi.lib(); L.trexhip_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
names = {0: ["head + taps 0-19", "wait B1", "taps 20-39", "wait B2", "output transform", "wait B3", "extra chunks"],
         1: ["V3 transform + crop rows", "wait B1", "conv1 tiles", "wait B2", "V2 transform", "wait B3", "extra chunks"]}
for it in range(3):
    seg.identify_device(crops.data_ptr(), N, probs.data_ptr()); seg.synchronize()
    buf = (C.c_ulonglong * 24)(); L.trexhip_debug_read(seg.handle, buf, 24)
    for w in range(2):
        v = list(buf)[8 * w:8 * w + 8]; n = max(1, v[7]); tot = sum(v[:7])
        print("consumer" if w == 0 else "producer", "rounds", v[7], "cycles per round %.0f:" % (tot / n), ", ".join("%s %.0f" % (names[w][i], v[i] / n) for i in range(7)))
seg.close()
