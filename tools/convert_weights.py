#!/usr/bin/env python3
"""Convert a TRex identity-network checkpoint (<base>_dict.pth, a V118_3 state_dict written by
visual_recognition_torch.py:841-921) into the flat blob trexhip_load_weights() takes.

    python tools/convert_weights.py model_dict.pth model.trxw [--width 80 --height 80 --channels 1]
"""
import argparse
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--width", type=int, default=80); ap.add_argument("--height", type=int, default=80)
    ap.add_argument("--channels", type=int, default=1)
    a = ap.parse_args()
    import torch
    sd = torch.load(a.src, map_location="cpu", weights_only=True)
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}   # PermuteAxesWrapper prefix
    classes = int(sd["fc2.weight"].shape[0])
    st = {n: sd[n].detach().to(torch.float32).numpy() for n, _ in weights.shapes(classes, a.channels, a.width, a.height)}
    blob = weights.pack_blob(st, classes, a.channels, a.width, a.height)
    with open(a.dst, "wb") as f:
        f.write(blob)
    print(f"wrote {a.dst}: {classes} classes, {len(blob)} bytes")


if __name__ == "__main__":
    main()
