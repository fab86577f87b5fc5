"""dev tool (tools/build_dev.sh build, TREXHIP_F12_DBG=128): cycles per phase of the fused conv1 + conv2 kernel, summed over the passes of two
workgroups (thread 0 of workgroup 0 and of workgroup gridDim.x / 2), printed per pass.
   gpurun -- 'TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so TREXHIP_F12_DBG=128 python tools/f12_stamps.py'"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
st = weights.synthetic_state(100, 31)
crops = torch.from_numpy(np.tile(weights.synthetic_crops(256, 3), (N // 256 + 1, 1, 1, 1))[:N]).cuda()
seg = capi.Segmenter(capi.default_params(256, 256, max_batch=1)); seg.load_weights(weights.pack_blob(st, 100)); seg.set_identity_precision(3)
probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda")
L = capi.lib(); L.trexhip_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
names = ["tap loop", "barrier 1", "out transform+pool (E1)", "barrier 2", "V3 transform (E2)", "wait + crop rows (P0)", "barrier 3", "conv1 tiles (P1)", "barrier 4",
         "V2 transform (P2)", "barrier 5"]
for it in range(3):
    seg.identify_device(crops.data_ptr(), N, probs.data_ptr()); seg.synchronize()
    buf = (C.c_ulonglong * 24)(); L.trexhip_debug_read(seg.handle, buf, 24)
    for w in range(2):
        v = list(buf)[12 * w:12 * w + 12]; n = max(1, v[11]); tot = sum(v[:11])
        print("workgroup", "0" if w == 0 else "G/2", "passes", v[11], "cycles per pass %.0f:" % (tot / n), ", ".join("%s %.0f" % (names[i], v[i] / n) for i in range(11)))
seg.close()
