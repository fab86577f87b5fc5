"""Randomised parity sweep of the device posture kernel against the oracle (dev tool): random blob shapes (narrow and wider than
64 px, with holes and dents), random outline settings.   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_posture.py 60 [seed]'"""
import sys
import numpy as np
from oracle import oracle
from test_posture_gpu import run_posture, compare

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0; total = 0; ties = 0
for case in range(n_cases):
    H, W = 240, 640
    bg = np.full((H, W), 200, np.uint8)
    fr = bg.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(int(rng.integers(3, 25))):
        cx, cy = rng.integers(20, W - 20), rng.integers(15, H - 15)
        a, b, th = rng.uniform(1, 70), rng.uniform(1, 14), rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1
        if rng.random() < 0.4: m &= rng.random((H, W)) < rng.uniform(0.7, 0.98)       # holes / ragged edge
        fr[m] = int(rng.integers(0, 150))
    kw = {}
    if rng.random() < 0.5: kw["outline_resample"] = float(rng.choice([0.5, 1.0, 1.5, 2.0, 0.7, 1.3]))
    if rng.random() < 0.3: kw["outline_smooth_samples"] = int(rng.choice([0, 2, 6]))
    if rng.random() < 0.3: kw["outline_approximate"] = int(rng.choice([0, 2, 3]))
    if rng.random() < 0.3: kw["midline_walk_offset"] = float(rng.choice([0.01, 0.05, 0.1, 0.2, 0.45]))
    try:
        res, outline, segs, info = run_posture(fr[None], bg, max_points=1024, **kw)
        n_cmp, n_tie = compare(res, outline, segs, info, oracle.posture_params(max_points=1024, **kw))      # round 5: exact equality (outline, tail, head, segments)
        total += n_cmp; ties += n_tie
    except AssertionError as e:
        import traceback
        fails += 1
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("FAIL case", case, kw, "line", tb.lineno, tb.line, str(e)[:200], flush=True)
print("cases", n_cases, "blobs compared bit for bit (outline, tail, head, midline segments)", total, "failures", fails)
