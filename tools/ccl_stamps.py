import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TREXHIP_CCL_STOP"] = "-1"
import torch
from trex_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CFG = sys.argv[2] if len(sys.argv) > 2 else "C4"
W, H, _, _ = synth.CONFIGS[CFG]
frames, bg = synth.batch_torch(CFG, B, "cuda")
print(f"# {CFG}, {B} frames; cycles of the phases of k_ccl_lds seen by workgroup 0: P1 row scan, P2 runs -> LDS, P3 link, P4 flatten, P5 numbering + counts, P6 filter + reservation, P7 records, P8 grouping")
seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=1024, max_pixels=1 << 18, max_runs=32768))
seg.set_background(bg)
L = capi.lib()
L.trexhip_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for it in range(3):
    seg.segment_device(frames.data_ptr(), B); seg.synchronize()
    buf = (C.c_ulonglong * 16)()
    L.trexhip_debug_read(seg.handle, buf, 16)
    v = list(buf)[:10]
    print([v[i + 1] - v[i] for i in range(8)], "total", v[8] - v[0], "| P7b: setup+scatter", v[9] - v[7], "rank+write", v[8] - v[9])
