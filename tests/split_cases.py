"""Synthetic merged blobs for the SplitBlob tests: dark bodies on a textured background whose darkness falls off from the body axis, so that
two touching individuals are one component at the track threshold and separate components at a higher one."""
import numpy as np


def merged_scene(seed, H=160, W=256, n_groups=6, per_group=(2, 3), amp=(60, 110)):
    rng = np.random.default_rng(seed)
    bg = (140 + ((np.arange(W)[None, :] * 3 + np.arange(H)[:, None] * 5) & 31) - 16).astype(np.uint8)
    depth = np.zeros((H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    gx = np.linspace(30, W - 30, n_groups)
    groups = []
    for g in range(n_groups):
        k = int(rng.integers(per_group[0], per_group[1] + 1))
        cx, cy = gx[g] + rng.uniform(-4, 4), rng.uniform(40, H - 40)
        groups.append(k)
        for i in range(k):
            # neighbours overlap at their flanks
            ox, oy = cx + rng.uniform(-2, 2), cy + (i - (k - 1) / 2) * rng.uniform(7.5, 10.5)
            th = rng.uniform(-0.35, 0.35)
            a, b = rng.uniform(14, 20), rng.uniform(4.0, 6.0)
            u = (xx - ox) * np.cos(th) + (yy - oy) * np.sin(th)
            v = -(xx - ox) * np.sin(th) + (yy - oy) * np.cos(th)
            r2 = (u / a) ** 2 + (v / b) ** 2
            d = rng.uniform(*amp) * np.clip(1.15 - r2, 0, 1)
            depth = np.maximum(depth, d.astype(np.float32))
    noise = rng.integers(-3, 4, (H, W))
    frame = np.clip(bg.astype(int) - np.rint(depth).astype(int) + noise, 0, 255).astype(np.uint8)
    return frame, bg, groups
