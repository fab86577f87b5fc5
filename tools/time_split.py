"""Timing of the SplitBlob search on the device vs the CPU restatement (dev tool).
   gpurun -- 'PYTHONPATH=.:tests python tools/time_split.py [frames]'"""
import sys, time
import numpy as np, torch
from oracle import oracle
from trex_amd import capi
from split_cases import merged_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 2048
rng = np.random.default_rng(3)
tile, bg_t, _ = merged_scene(11, H=256, W=256, n_groups=6, per_group=(2, 3))
bg = np.tile(bg_t, (8, 8))
frames = np.empty((n, H, W), np.uint8)
for f in range(n):
    tiles = [merged_scene(1000 + ((f * 7 + i) % 97), H=256, W=256, n_groups=6, per_group=(2, 3))[0] for i in range(8)]
    row = np.concatenate(tiles, axis=1)
    frames[f] = np.tile(row, (8, 1))
seg = capi.Segmenter(capi.default_params(W, H, max_batch=n, max_blobs=1024))
seg.set_background(bg)
d = torch.from_numpy(frames).cuda()
seg.segment_device(d.data_ptr(), n)
det = seg.fetch()
nb = sum(len(r.blobs) for r in det)
d_pres = torch.full((nb,), 2, dtype=torch.int32, device="cuda")
d_thr = torch.zeros(nb, dtype=torch.int32, device="cuda")
d_info = torch.zeros(nb * capi.SPLIT_INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
ranges = [(40, 330)]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg.split_search_device(d_pres.data_ptr(), nb, d_thr.data_ptr(), d_info.data_ptr(), method=1, size_ranges=ranges)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    seg.rethreshold_per_blob(d_thr.data_ptr(), 1, ranges)
    torch.cuda.synchronize(); t2 = time.perf_counter()
info = d_info.cpu().numpy().view(capi.SPLIT_INFO_DTYPE)
print(f"{n} frames, {nb} candidate blobs ({nb / n:.0f}/frame), split {int((info['threshold'] >= 0).sum())}, labelling passes {int(info['n_evaluated'].sum())}")
print(f"device: search {1e3 * (t1 - t0):.2f} ms, sub-blobs at the thresholds {1e3 * (t2 - t1):.2f} ms  => {nb / (t2 - t0):.0f} blobs/s")
sp = oracle.split_params(size_ranges=ranges)
r = det[0]
t0 = time.perf_counter(); m = 0
for b in r.blobs[:200]:
    runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
    px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
    oracle.split_search(runs, px, bg, 1, sp, 2); m += 1
t1 = time.perf_counter()
print(f"CPU restatement (1 core): {1e3 * (t1 - t0) / m:.3f} ms per blob => {m / (t1 - t0):.0f} blobs/s")
