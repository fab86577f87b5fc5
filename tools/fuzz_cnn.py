"""Randomised parity sweep of the identity network (default chain: cnn_wpre.h) against the CPU restatement: random crop counts around the pass /
ticket / M-tile boundaries of the persistent kernels, class counts, 1 and 3 channels, dense / sparse / empty / saturated crops, random weights.
   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_cnn.py 60 [seed]'"""
import sys
import numpy as np
import torch
from oracle import cnn_oracle
from trex_amd import capi, weights

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0; worst = 0.0; total = 0
for case in range(n_cases):
    ch = 3 if rng.random() < 0.3 else 1
    classes = int(rng.choice([2, 7, 8, 33, 100, 101, 256, 300]))
    n = int(rng.choice([1, 2, 3, 5, 6, 7, 11, 12, 13, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700]))
    st = weights.synthetic_state(classes, int(rng.integers(1, 1 << 30)), channels=ch)
    kind = rng.integers(0, 4)
    if kind == 0:
        crops = rng.integers(0, 256, (n, 80, 80, ch)).astype(np.uint8)
    elif kind == 1:
        crops = weights.synthetic_crops(n, int(rng.integers(1, 1 << 30)), channels=ch)
    elif kind == 2:
        crops = (rng.random((n, 80, 80, ch)) < 0.05).astype(np.uint8) * rng.integers(1, 256, (n, 80, 80, ch)).astype(np.uint8)
    else:
        crops = np.zeros((n, 80, 80, ch), np.uint8); crops[::2] = 255; crops[:, :3] = 255; crops[:, :, -2:] = 200
    seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1))
    seg.load_weights(weights.pack_blob(st, classes, channels=ch))
    got = seg.probabilities(crops if ch == 3 else crops)
    pick = np.unique(np.concatenate([np.arange(min(n, 24)), np.arange(max(0, n - 24), n), rng.integers(0, n, 16)]))
    want, _ = cnn_oracle.predict(st, crops[pick], threads=8)
    err = float(np.abs(got[pick] - want).max())
    worst = max(worst, err); total += len(pick)
    if not (err <= 1e-4 and np.allclose(got.sum(1), 1.0, atol=1e-5)):
        fails += 1
        print("FAIL case", case, dict(n=n, classes=classes, ch=ch, kind=int(kind)), "max |dp|", err, flush=True)
    seg.close()
print("cases", n_cases, "crops compared", total, "failures", fails, "largest |dp|", worst)
