"""Deterministic synthetic frames for the BASELINE.json configs.

Integer recipe (numpy, host side; the same arrays are uploaded to HBM for the GPU run and
handed to the CPU baseline):

  background  bg(x,y) = 128 + ((3x + 5y) & 31) - 16
  blob i of N at frame t: centre on a jittered ceil(sqrt(N)) grid (pitch = W / ceil(sqrt(N))),
      offset by an LCG (s = s*1664525 + 1013904223, seed = 0x7E3 + 977*config + t) within +-pitch/4;
      ellipse semi-axes (18,5) px rotated by theta = 2*pi*((s>>8)&255)/256;
      inside value bg - 60 - ((x^y)&7)
  sensor noise +-((lcg>>16)&3) on every pixel (|noise| <= 3 < detect_threshold = 15)
"""
import math
import numpy as np

CONFIGS = {
    # name: (width, height, n_blobs, config id)
    "C2": (1280, 720, 32, 2),
    "C3": (2048, 2048, 100, 3),
    "C4": (2048, 2048, 100, 4),
    "C5": (4096, 4096, 256, 5),
}


def background(width, height):
    x = np.arange(width, dtype=np.int32)[None, :]
    y = np.arange(height, dtype=np.int32)[:, None]
    return (128 + ((3 * x + 5 * y) & 31) - 16).astype(np.uint8)


def _lcg(s):
    return (s * 1664525 + 1013904223) & 0xFFFFFFFF


def frame(width, height, n_blobs, config_id, t, bg=None, noise=True):
    """One synthetic gray frame (uint8, height x width)."""
    if bg is None:
        bg = background(width, height)
    img = bg.astype(np.int32).copy()
    g = int(math.ceil(math.sqrt(n_blobs)))
    pitch_x, pitch_y = width / g, height / g
    s = (0x7E3 + 977 * config_id + t) & 0xFFFFFFFF
    for i in range(n_blobs):
        s = _lcg(s)
        jx = ((s >> 4) & 0xFFFF) / 65535.0 - 0.5
        s = _lcg(s)
        jy = ((s >> 4) & 0xFFFF) / 65535.0 - 0.5
        s = _lcg(s)
        theta = 2.0 * math.pi * ((s >> 8) & 255) / 256.0
        cx = (i % g + 0.5) * pitch_x + jx * pitch_x * 0.5
        cy = (i // g + 0.5) * pitch_y + jy * pitch_y * 0.5
        a, b = 18.0, 5.0
        r = int(a) + 2
        x0, x1 = max(0, int(cx) - r), min(width - 1, int(cx) + r)
        y0, y1 = max(0, int(cy) - r), min(height - 1, int(cy) + r)
        if x1 < x0 or y1 < y0:
            continue
        xs = np.arange(x0, x1 + 1, dtype=np.float64)[None, :] - cx
        ys = np.arange(y0, y1 + 1, dtype=np.float64)[:, None] - cy
        ct, st = math.cos(theta), math.sin(theta)
        u = xs * ct + ys * st
        v = -xs * st + ys * ct
        inside = (u / a) ** 2 + (v / b) ** 2 <= 1.0
        xi = np.arange(x0, x1 + 1, dtype=np.int32)[None, :]
        yi = np.arange(y0, y1 + 1, dtype=np.int32)[:, None]
        val = bg[y0:y1 + 1, x0:x1 + 1].astype(np.int32) - 60 - ((xi ^ yi) & 7)
        sub = img[y0:y1 + 1, x0:x1 + 1]
        sub[inside] = val[inside]
    if noise:
        # per-pixel LCG noise, vectorised: state = lcg(seed ^ pixel index)
        idx = np.arange(width * height, dtype=np.uint64).reshape(height, width)
        st = (idx * np.uint64(2654435761) + np.uint64((0x9E37 + t * 7919 + config_id) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
        st = (st * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        mag = ((st >> np.uint64(16)) & np.uint64(3)).astype(np.int32)
        sign = np.where(((st >> np.uint64(20)) & np.uint64(1)) == 1, 1, -1).astype(np.int32)
        img = img + sign * mag
    return np.clip(img, 0, 255).astype(np.uint8)


def batch(name, n_frames, t0=0):
    """(frames[n,H,W] uint8, bg[H,W] uint8) for a named config."""
    w, h, nb, cid = CONFIGS[name]
    bg = background(w, h)
    fr = np.stack([frame(w, h, nb, cid, t0 + t, bg) for t in range(n_frames)])
    return fr, bg


def random_scene(rng, width, height, density=0.02, max_len=12):
    """Adversarial random binary-ish scene for parity tests: random short bars on a flat background."""
    bg = np.full((height, width), 120, np.uint8)
    fr = bg.copy()
    n = int(width * height * density / max(1, max_len // 2))
    ys = rng.integers(0, height, n)
    xs = rng.integers(0, width, n)
    ls = rng.integers(1, max_len + 1, n)
    vs = rng.integers(0, 90, n)
    for y, x, l, v in zip(ys, xs, ls, vs):
        fr[y, x:min(width, x + l)] = v
    return fr, bg


def batch_torch(name, n_frames, device, t0=0):
    """Same integer recipe as batch(), evaluated with torch on `device` (used by bench.py to fill HBM
    with many distinct frames quickly).  Bit-identical to batch() -- tests/test_synth.py."""
    import torch
    w, h, nb, cid = CONFIGS[name]
    bg_np = background(w, h)
    bg = torch.from_numpy(bg_np).to(device)
    bg32 = bg.to(torch.int32)
    out = torch.empty((n_frames, h, w), dtype=torch.uint8, device=device)
    g = int(math.ceil(math.sqrt(nb)))
    pitch_x, pitch_y = w / g, h / g
    idx = torch.arange(w * h, dtype=torch.int64, device=device).reshape(h, w)
    for t in range(n_frames):
        tt = t0 + t
        img = bg32.clone()
        s = (0x7E3 + 977 * cid + tt) & 0xFFFFFFFF
        for i in range(nb):
            s = _lcg(s); jx = ((s >> 4) & 0xFFFF) / 65535.0 - 0.5
            s = _lcg(s); jy = ((s >> 4) & 0xFFFF) / 65535.0 - 0.5
            s = _lcg(s); theta = 2.0 * math.pi * ((s >> 8) & 255) / 256.0
            cx = (i % g + 0.5) * pitch_x + jx * pitch_x * 0.5
            cy = (i // g + 0.5) * pitch_y + jy * pitch_y * 0.5
            a, b = 18.0, 5.0
            r = int(a) + 2
            x0, x1 = max(0, int(cx) - r), min(w - 1, int(cx) + r)
            y0, y1 = max(0, int(cy) - r), min(h - 1, int(cy) + r)
            if x1 < x0 or y1 < y0:
                continue
            # the inside test is evaluated in float64 on the host (tiny window) so both generators agree bit for bit
            xs = np.arange(x0, x1 + 1, dtype=np.float64)[None, :] - cx
            ys = np.arange(y0, y1 + 1, dtype=np.float64)[:, None] - cy
            ct, st = math.cos(theta), math.sin(theta)
            u = xs * ct + ys * st
            v = -xs * st + ys * ct
            inside = (u / a) ** 2 + (v / b) ** 2 <= 1.0
            xi = np.arange(x0, x1 + 1, dtype=np.int32)[None, :]
            yi = np.arange(y0, y1 + 1, dtype=np.int32)[:, None]
            val = bg_np[y0:y1 + 1, x0:x1 + 1].astype(np.int32) - 60 - ((xi ^ yi) & 7)
            sub = img[y0:y1 + 1, x0:x1 + 1]
            ins = torch.from_numpy(inside).to(device)
            sub[ins] = torch.from_numpy(val).to(device)[ins]
        st_ = (idx * 2654435761 + ((0x9E37 + tt * 7919 + cid) & 0xFFFFFFFF)) & 0xFFFFFFFF
        st_ = (st_ * 1664525 + 1013904223) & 0xFFFFFFFF
        mag = ((st_ >> 16) & 3).to(torch.int32)
        sign = torch.where(((st_ >> 20) & 1) == 1, 1, -1).to(torch.int32)
        img = img + sign * mag
        out[t] = img.clamp_(0, 255).to(torch.uint8)
    return out, bg
