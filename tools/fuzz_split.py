"""Randomised parity sweep of the SplitBlob threshold search (detect -> trexhip_split_search_device) against the CPU restatement
of SplitBlob::split (dev tool).   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_split.py 150 [seed]'"""
import sys
import numpy as np
from oracle import oracle
from split_cases import merged_scene
from test_split_gpu import device_search

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = blobs = found = 0
for case in range(n_cases):
    frames, bg = [], None
    for t in range(2):
        fr, bg, _ = merged_scene(int(rng.integers(1 << 30)), n_groups=int(rng.integers(3, 8)), per_group=(1, int(rng.integers(2, 5))),
                                 amp=(float(rng.uniform(30, 70)), float(rng.uniform(70, 120))))
        frames.append(fr)
    frames = np.stack(frames)
    method = int(rng.integers(0, 3))
    if method == 2:                                           # grey values themselves: make bodies bright on a dark field
        frames = 255 - frames
    algorithm = int(rng.integers(1, 3))
    tt, tp = int(rng.integers(5, 40)), int(rng.integers(5, 40))
    cp = bool(rng.integers(0, 2))
    shrink, limit = float(rng.choice([0.05, 0.2, 0.5, 0.9])), float(rng.choice([0.1, 0.2, 0.6]))
    cm = float(rng.choice([1.0, 1.0, 0.5, 0.13]))
    ranges = [(float(rng.integers(10, 80)) * cm * cm, float(rng.integers(200, 500)) * cm * cm)] if rng.random() < 0.75 else []
    conn = 4 if rng.random() < 0.3 else 8
    dkw = {"connectivity": conn, "cm_per_pixel": cm}
    if method == 2:
        dkw.update(enable_difference=0, threshold=int(rng.integers(138, 150)))
    try:
        det, presumed, thr, info, sub = device_search(frames, bg, lambda r: rng.integers(0, 5, len(r.blobs)), method, ranges,
                                                      detect_kw=dkw, algorithm=algorithm, track_threshold=tt, track_posture_threshold=tp,
                                                      calculate_posture=int(cp), blob_split_max_shrink=shrink, blob_split_global_shrink_limit=limit)
        sp = oracle.split_params(tt, tp, cp, algorithm, shrink, limit, cm, ranges)
        for f, r in enumerate(det):
            for j, b in enumerate(r.blobs):
                k = int(r.info["blob_begin"]) + j
                pr = int(presumed[k])
                if pr <= 0:
                    assert thr[k] == -1
                    continue
                blobs += 1
                runs = r.runs[b["run_begin"]:b["run_begin"] + b["n_runs"]]
                px = r.pixels[b["pix_begin"]:b["pix_begin"] + b["n_pixels"]]
                w = oracle.split_search(runs, px, bg, method, sp, pr, connectivity=conn)
                g = info[k]
                assert g["status"] == 0, g
                assert (g["threshold"], g["effective_threshold"], g["initial_action"], g["n_result"]) == (w.threshold, w.effective_threshold, w.initial_action, w.n_result), (g, w.threshold, w.effective_threshold, w.initial_action, w.n_result)
                assert (g["min_pixel"], g["max_pixel"], g["first_size"], g["min_size_bound"]) == (w.min_pixel, w.max_pixel, w.first_size, w.min_size_bound)
                found += w.threshold >= 0
    except AssertionError as e:
        fails += 1
        print("FAIL case", case, method, algorithm, tt, tp, cp, shrink, limit, cm, ranges, conn, str(e)[:300], flush=True)
print("cases", n_cases, "candidate blobs", blobs, "split", found, "failures", fails)
