"""Shared by tests/test_golden_e2e.py and tools/e2e_ablation.py: rebuild the reference's test frames from the committed fixture
(tests/golden/e2e_testframes.npz, see make_e2e_fixture.py) and score a segmentation against the golden CSV rows."""
import os
import numpy as np

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_testframes.npz")
RANGES = [(70, 420)]          # track_size_filter of videos/test.settings


class Golden:
    def __init__(self):
        z = np.load(FIX)
        self.frames = [int(f) for f in z["frames"]]
        self.H, self.W = [int(v) for v in z["shape"]]
        self.rects, self.gold, self.first, self.win = z["rects"], z["gold"], z["first"], z["win"]
        sizes = (self.rects[:, 3] - self.rects[:, 1]).astype(np.int64) * (self.rects[:, 4] - self.rects[:, 2])
        self.woff = np.concatenate([[0], np.cumsum(sizes)])

    def rebuild(self, i):
        """frame i of self.frames: (image, background, golden rows).  background = 128, image = 128 - d inside the stored windows."""
        bg = np.full((self.H, self.W), 128, np.uint8)
        img = bg.copy()
        for k in range(int(self.first[i]), int(self.first[i + 1])):
            _, x0, y0, x1, y1 = [int(v) for v in self.rects[k]]
            d = self.win[self.woff[k]:self.woff[k + 1]].reshape(y1 - y0, x1 - x0).astype(np.int16)
            # overlapping windows of two individuals hold the same difference values: pasting twice is idempotent
            img[y0:y1, x0:x1] = (128 - d).astype(np.uint8)
        return img, bg, self.gold[int(self.first[i]):int(self.first[i + 1])]


def score(sub_blobs, gold):
    """golden rows whose blob id is reproduced (hits) and of those the ones with the identical pixel count (exact)"""
    mine = {int(b["bid"]): int(b["n_pixels"]) for b in sub_blobs if b["flags"] == 0}
    hits, exact, deltas = 0, 0, []
    for g in gold:
        bid, npx = int(g[1]), int(g[2])
        if bid in mine:
            hits += 1
            deltas.append(abs(mine[bid] - npx))
            exact += mine[bid] == npx
    return hits, exact, deltas


def run_variant(G, oracle, detect_threshold=9, track_threshold=12, inclusive=1, connectivity=8, frames=None):
    tot = hits = exact = 0
    deltas = []
    for i in (range(len(G.frames)) if frames is None else frames):
        img, bg, gold = G.rebuild(i)
        p = oracle.make_params(G.W, G.H, threshold=detect_threshold, inclusive=inclusive, connectivity=connectivity, size_ranges=[(1, 10000)])
        blobs, runs, px = oracle.rethreshold_frame(img, bg, p, 1, track_threshold, RANGES)
        h, e, d = score(blobs, gold)
        tot += len(gold); hits += h; exact += e; deltas += d
    return tot, hits, exact, deltas
