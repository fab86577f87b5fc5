"""Builds tests/golden/e2e_testframes.npz from the reference's shipped end-to-end test data
(videos/test_frames/*.jpg + videos/compare_data_automatic/test_fish*.csv, settings videos/test.settings).
Runs only in the build container (needs /root/reference and PIL); the fixture is DATA: for a few frames the
pixel windows around each golden individual (frame + background) and the golden CSV rows of those frames.

Background = rounded mean of every 2nd frame (average_samples=100 of 200 frames; the reference's sampler is
in the un-vendored commons, this choice reproduces the golden num_pixels best -- DESIGN.md section 2).
"""
import csv
import glob
import os
import numpy as np
from PIL import Image

REF = "/root/reference/videos"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_testframes.npz")
FRAMES = [0, 3, 31, 59, 87, 115, 143, 171, 199]
HALF = 72


def main():
    files = sorted(glob.glob(os.path.join(REF, "test_frames", "frame_*.jpg")))
    assert len(files) == 200
    acc = None
    for f in files[0::2]:
        a = np.asarray(Image.open(f)).astype(np.float64)
        acc = a if acc is None else acc + a
    bg = np.rint(acc / 100).astype(np.uint8)
    gold = {}
    for i in range(8):
        with open(os.path.join(REF, "compare_data_automatic", f"test_fish{i}.csv")) as fh:
            for row in csv.DictReader(fh):
                try:
                    fr = int(row["frame"])
                    gold.setdefault(fr, []).append((i, int(float(row["blobid"])), int(float(row["num_pixels"])),
                                                    float(row["X#wcentroid (cm)"]), float(row["midline_length"])))
                except (ValueError, OverflowError):
                    pass   # inf / missing rows
    store = {"frames": np.array(FRAMES), "shape": np.array(bg.shape)}
    for fr in FRAMES:
        img = np.asarray(Image.open(files[fr]))
        rects, fpx, bpx = [], [], []
        for (_, bid, npx, xc, ml) in gold[fr]:
            x, y = bid >> 19, (bid >> 6) & 8191
            x0, y0 = max(0, x - HALF), max(0, y - 24)
            x1, y1 = min(bg.shape[1], x + HALF), min(bg.shape[0], y + 2 * HALF)
            rects.append((x0, y0, x1, y1))
            fpx.append(img[y0:y1, x0:x1].copy()); bpx.append(bg[y0:y1, x0:x1].copy())
        store[f"rects/{fr}"] = np.array(rects, np.int32)
        store[f"gold/{fr}"] = np.array([(g[0], g[1], g[2], g[3], g[4]) for g in gold[fr]], np.float64)
        for k, (a, b) in enumerate(zip(fpx, bpx)):
            store[f"f/{fr}/{k}"] = a
            store[f"b/{fr}/{k}"] = b
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
