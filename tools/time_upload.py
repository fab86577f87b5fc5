"""dev: host-input legs alone -- trexhip_segment / trexhip_segment_color on pageable frames, synchronised after each call"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, synth
B = 128
frames, bg = synth.batch_torch("C4", B, "cuda")
H, W = frames.shape[1:]
seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=400, max_pixels=1 << 18, max_runs=32768))
seg.set_background(bg)
g = [np.ascontiguousarray(frames[i].cpu().numpy()) for i in range(B)]
c = [np.ascontiguousarray(np.repeat(x[..., None], 4, 2)) for x in g]
for name, fn, data, bpf in (("gray", seg.segment_host, g, W * H), ("bgra", seg.segment_color_host, c, W * H * 4)):
    fn(data); seg.synchronize(); seg.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(data); t1 = time.perf_counter(); seg.synchronize()
    dt = (time.perf_counter() - t0) / 3
    cm, cn = seg.profile_read(capi.STAGE_UPLOAD_COPY); dm, dn = seg.profile_read(capi.STAGE_UPLOAD_DMA)
    print(name, "ms/call %.2f  GB/s %.1f  copy ms/frame %.3f  dma ms/frame %.3f  (sum of legs per call %.2f ms)" % (dt * 1e3, B * bpf / dt / 1e9, cm / cn, dm / dn, (cm / cn + dm / dn) * B))
