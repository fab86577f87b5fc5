"""Randomised parity sweep of detect -> re-threshold (2 frames per batch) against the oracle (dev tool).
   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_rethreshold.py 300 [seed]'"""
import sys
import numpy as np
from oracle import oracle
from test_rethreshold_gpu import run

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for case in range(n_cases):
    W = int(rng.choice([64, 257, 640, 1024, 2048])); H = int(rng.choice([16, 64, 200, 360]))
    bg = rng.integers(90, 200, (H, W)).astype(np.uint8)
    frames = []
    for t in range(2):
        fr = np.clip(bg.astype(int) + rng.integers(-6, 6, (H, W)), 0, 255).astype(np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(int(rng.integers(1, 30))):
            cx, cy, a, b = rng.integers(0, W), rng.integers(0, H), rng.uniform(2, 40), rng.uniform(2, 14)
            m = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
            fr[m] = np.clip(bg[m].astype(int) - rng.integers(0, 120, int(m.sum())), 0, 255)
        frames.append(fr)
    frames = np.stack(frames)
    method, thr = int(rng.integers(0, 3)), int(rng.integers(0, 90))
    ranges = [(float(rng.integers(1, 50)), float(rng.integers(60, 2000)))] if rng.random() < 0.7 else []
    dkw = {"connectivity": 4} if rng.random() < 0.3 else {}
    try:
        det, sub = run(frames, bg, thr, method, ranges, dkw)
        for f in range(2):
            if det[f].info["flags"] != 0 or sub[f].info["flags"] != 0:
                continue
            ob, orr, opx = oracle.rethreshold_frame(frames[f], bg, oracle.make_params(W, H, **dkw), method, thr, ranges)
            r = sub[f]
            assert len(r.blobs) == len(ob), (len(r.blobs), len(ob))
            assert r.runs.tobytes() == orr.tobytes() and r.pixels.tobytes() == opx.tobytes()
            want = ob.copy()
            want["parent"] = want["parent"] + det[f].info["blob_begin"]
            for name in ob.dtype.names:
                assert np.array_equal(r.blobs[name], want[name]), name
    except AssertionError as e:
        fails += 1
        print("FAIL case", case, (W, H), method, thr, ranges, dkw, str(e)[:200], flush=True)
print("cases", n_cases, "failures", fails)
