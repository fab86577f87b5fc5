"""Kernel time of k_posture on the bench scene (dev tool; TREXHIP_POSTURE_STOP=N returns after phase N).
   gpurun -- 'PYTHONPATH=. python tools/time_posture.py [frames]'"""
import sys, time
import numpy as np, torch
from trex_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = H = 2048
base, bg = synth.batch_torch("C4", 8, "cuda")
frames = base.repeat((n + 7) // 8, 1, 1)[:n].contiguous()
seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=256))
seg.set_background(bg if not isinstance(bg, np.ndarray) else bg)
seg.segment_device(frames.data_ptr(), n)
res = seg.fetch()
nb = sum(len(r.blobs) for r in res)
MP = 256
o = torch.empty((nb, MP, 2), dtype=torch.float32, device="cuda"); s4 = torch.empty((nb, MP // 2 + 1, 4), dtype=torch.float32, device="cuda")
inf = torch.zeros((nb, 8), dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg.posture_device(nb, o.data_ptr(), s4.data_ptr(), inf.data_ptr(), max_points=MP)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"{nb} blobs: {1e6 * (t1 - t0):.0f} us")

i = inf.cpu().numpy()
ok = i[:, 0] == 0
print("mean traced", i[ok, 5].mean(), "outline", i[ok, 1].mean(), "segments", i[ok, 2].mean())
