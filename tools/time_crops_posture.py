"""Host and device time of the posture-normalised crop call (dev tool).  gpurun -- 'PYTHONPATH=. python tools/time_crops_posture.py [frames]'"""
import sys, time
import numpy as np, torch
from trex_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = H = 2048
base, bg = synth.batch_torch("C4", 8, "cuda")
frames = base.repeat((n + 7) // 8, 1, 1)[:n].contiguous()
seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=256))
seg.set_background(bg)
seg.segment_device(frames.data_ptr(), n)
nb = int(seg.fetch_raw().total_blobs)
MP = 256
o = torch.empty((nb, MP, 2), dtype=torch.float32, device="cuda"); s4 = torch.empty((nb, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
inf = torch.zeros((nb, 8), dtype=torch.int32, device="cuda"); mid = torch.zeros((nb, 25, 4), dtype=torch.float32, device="cuda"); minfo = torch.zeros((nb, 8), dtype=torch.int32, device="cuda")
crops = torch.zeros((nb, 80, 80), dtype=torch.uint8, device="cuda")
seg.posture_device(nb, o.data_ptr(), s4.data_ptr(), inf.data_ptr(), max_points=MP)
seg.midline_device(nb, MP, inf.data_ptr(), s4.data_ptr(), mid.data_ptr(), minfo.data_ptr())
seg.profile_enable(True)
for it in range(3):
    torch.cuda.synchronize(); seg.profile_reset(); t0 = time.perf_counter()
    seg.crops_posture_device(crops.data_ptr(), nb, minfo.data_ptr())
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms, k = seg.profile_read(5)
print(f"{nb} blobs: call returns after {1e3 * (t1 - t0):.2f} ms (host: infos to the host, {nb} transforms, maps to the device), kernel {ms / max(k, 1):.3f} ms, total {1e3 * (t2 - t0):.2f} ms")
for it in range(2):
    torch.cuda.synchronize(); seg.profile_reset(); t0 = time.perf_counter()
    seg.crops_device(crops.data_ptr(), nb, normalization=1)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms, k = seg.profile_read(5)
print(f"moments: call {1e3 * (t1 - t0):.2f} ms, kernel {ms / max(k, 1):.3f} ms, total {1e3 * (t2 - t0):.2f} ms")
