"""Randomised parity sweep of the detect + re-threshold path against the oracle (dev tool; a few hundred cases, ~1 min on the GPU).
   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_detect.py 300'"""
import sys
import numpy as np
import torch
from oracle import oracle
from trex_amd import capi
from test_segment_gpu import run_gpu, assert_frame_equal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
skipped = 0
for case in range(n_cases):
    W = int(rng.choice([16, 33, 100, 257, 640, 1024, 1500, 2048, 3000]))
    H = int(rng.choice([8, 31, 64, 200, 480]))
    kind = rng.integers(0, 5)
    bg = rng.integers(40, 220, (H, W)).astype(np.uint8) if rng.random() < 0.5 else np.full((H, W), int(rng.integers(60, 200)), np.uint8)
    fr = bg.copy()
    if kind == 0:      # sparse specks
        m = rng.random((H, W)) < rng.uniform(0.001, 0.05)
        fr[m] = rng.integers(0, 256, int(m.sum()))
    elif kind == 1:    # dense noise
        fr = np.clip(bg.astype(int) + rng.integers(-60, 60, (H, W)), 0, 255).astype(np.uint8)
    elif kind == 2:    # long horizontal / vertical bars and combs (many runs per row, long label chains)
        for _ in range(int(rng.integers(1, 30))):
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            if rng.random() < 0.5: fr[y, x:x + int(rng.integers(1, W))] = 0
            else: fr[y:y + int(rng.integers(1, H)), x] = 0
        if rng.random() < 0.5: fr[::2, ::2] = 5
    elif kind == 3:    # spirals / serpentines: worst case for union-find
        for y in range(0, H, 2): fr[y, :] = 3
        for y in range(1, H, 2): fr[y, (W - 1) if (y // 2) % 2 == 0 else 0] = 3
    else:              # blobs
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(int(rng.integers(1, 40))):
            cx, cy, a, b = rng.integers(0, W), rng.integers(0, H), rng.uniform(1, 30), rng.uniform(1, 12)
            fr[((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1] = int(rng.integers(0, 256))
    kw = dict(threshold=int(rng.integers(1, 80)))
    if rng.random() < 0.3: kw["connectivity"] = 4
    if rng.random() < 0.2: kw["absolute_difference"] = 0
    if rng.random() < 0.2: kw["image_invert"] = 1
    if rng.random() < 0.15: kw["threshold_maximum"] = int(rng.integers(kw["threshold"], 255))
    if rng.random() < 0.15: kw["enable_difference"] = 0
    if rng.random() < 0.2: kw["use_closing"] = 1; kw["closing_size"] = int(rng.choice([1, 3, 5]))
    if rng.random() < 0.2: kw["dilation_size"] = int(rng.choice([-2, -1, 1, 2]))
    if rng.random() < 0.3: kw["zero_is_background"] = 0
    kw["max_runs"] = 400000; kw["max_pixels"] = 1 << 21
    # batches of 2..16 frames (variants of the scene) reach the kernels that take several frames per wave (k_rows32b); frame 0 is checked
    nb = int(rng.choice([1, 1, 2, 3, 4, 6, 8, 16])) if W * H <= 1 << 20 else 1
    batch = [fr]
    for _ in range(nb - 1):
        g = fr.copy()
        m = rng.random((H, W)) < 0.01
        g[m] = rng.integers(0, 256, int(m.sum()))
        batch.append(g)
    order = rng.permutation(nb)
    batch = [batch[i] for i in order]
    pos0 = int(np.where(order == 0)[0][0])
    try:
        res_all = run_gpu(np.stack(batch), bg, **kw)
        res = [res_all[pos0]]
        if res[0].info["flags"] != 0:          # capacity overflow must be genuine: the oracle exceeds a pool as well
            from test_segment_gpu import oracle_params
            ob, orr, opx = oracle.segment(fr, bg, oracle_params(W, H, **kw))
            assert len(ob) > 32768 or len(orr) > kw["max_runs"] or len(opx) > kw["max_pixels"], ("spurious overflow", int(res[0].info["flags"]), len(ob), len(orr), len(opx))
            skipped += 1
            continue
        assert_frame_equal(res[0], fr, bg, **kw)
        k2 = int(rng.integers(0, nb))
        if nb > 1 and res_all[k2].info["flags"] == 0: assert_frame_equal(res_all[k2], batch[k2], bg, **kw)
    except AssertionError as e:
        fails += 1
        print("FAIL case", case, (W, H), kind, kw, str(e)[:200], flush=True)
print("cases", n_cases, "failures", fails, "genuine overflows", skipped)
