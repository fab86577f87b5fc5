"""Round 6, detect pass on ONE context: k_ccl_lds by capacity (TREXHIP_CCL_INST) x frame groups on two streams (TREXHIP_SEG_GROUPS /
TREXHIP_SEG_SCHEME).  Every variant must give the tables of the first one (the L instance alone) byte for byte.
   python tools/r06_detect.py [C4|C2|C5] [frames] [variants...]      variant = inst:groups:scheme, e.g. 3:1:0 2:1:0 2:2:2"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trex_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
W, H, n_ind, _ = synth.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if cfg == "C5" else 256)
variants = sys.argv[3:] or ["3:1:0", "2:1:0", "4:1:0", "1:1:0", "0:1:0", "3:2:1", "2:2:1", "3:2:2", "2:2:2", "2:4:1", "2:4:2", "0:2:2"]
nb = min(B, 64)
base, bg = synth.batch_torch(cfg, nb, "cuda")
frames = base.repeat((B + nb - 1) // nb, 1, 1)[:B].contiguous()      # B frames, nb distinct ones
torch.cuda.synchronize()
ref = None
print(f"# {cfg}: {B} frames of {W}x{H}, {n_ind} individuals; serial passes on one context (its own stream); us per pass", flush=True)
for v in variants:
    inst, G, scheme = (int(x) for x in v.split(":"))
    os.environ["TREXHIP_CCL_INST"] = str(inst); os.environ["TREXHIP_SEG_GROUPS"] = str(G); os.environ["TREXHIP_SEG_SCHEME"] = str(scheme)
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=4 * n_ind, max_pixels=1 << 18, max_runs=32768), stream=None)
    seg.set_background(bg)
    for _ in range(3):
        seg.segment_device(frames.data_ptr(), B)
    seg.synchronize()
    res = seg.fetch()
    sig = [(r.info["n_blobs"], r.blobs.tobytes(), r.runs.tobytes(), r.pixels.tobytes()) for r in res]
    lines = int(np.mean([r.info["n_raw_runs"] for r in res]))
    if ref is None:
        ref = sig
    same = all(a[1:] == b[1:] for a, b in zip(sig, ref))
    ts = []
    for rep in range(3):
        seg.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            seg.segment_device(frames.data_ptr(), B)
        seg.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    seg.profile_enable(True); seg.profile_reset()
    for _ in range(10):
        seg.segment_device(frames.data_ptr(), B)
    seg.synchronize()
    ms, n = seg.profile_read(capi.STAGE_SEGMENT_ALL); rms, rn = seg.profile_read(capi.STAGE_ROWS)
    seg.profile_enable(False)
    print(f"inst {inst} groups {G} scheme {scheme}: wall {min(ts):7.1f} (max {max(ts):7.1f})  events: pass {ms / max(n, 1) * 1e3:7.1f} rows {rms / max(rn, 1) * 1e3:7.1f}"
          f"  lines/frame {lines}  {'same tables' if same else 'TABLES DIFFER'}", flush=True)
    seg.close()
