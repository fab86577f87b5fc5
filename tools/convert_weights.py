#!/usr/bin/env python3
"""Convert a TRex identity-network checkpoint into the flat blob trexhip_load_weights() takes.

The reference writes `<base>_dict.pth` as {'model': None, 'state_dict': <V118_3 state_dict>, 'metadata': {'input_shape': (W, H, C),
'num_classes', 'model_type', ...}} (visual_recognition_torch.py:102-117 save_model_files); a bare state_dict is accepted too.

    python tools/convert_weights.py model_dict.pth model.trxw
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import weights  # noqa: E402


def convert(obj, width=None, height=None, channels=None):
    """checkpoint object (as torch.load returns it) -> (blob bytes, classes, width, height, channels)"""
    import torch
    meta = {}
    if isinstance(obj, dict) and "state_dict" in obj:
        meta = obj.get("metadata") or {}
        sd = obj["state_dict"]
    elif isinstance(obj, dict):
        sd = obj
    else:
        sd = obj.state_dict()
    if sd is None:
        raise ValueError("checkpoint holds no state_dict")
    mt = str(meta.get("model_type", "v118_3")).lower()
    if "v118_3" not in mt:
        raise ValueError(f"model_type {meta.get('model_type')!r} is not V118_3: only that network is implemented (visual_identification_network_torch.py:184-258)")
    # PermuteAxesWrapper / Sequential prefixes (visual_identification_network_torch.py:618-644); BatchNorm's step counter is not a weight
    clean = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        for pre in ("model.", "module.", "0."):
            while k.startswith(pre):
                k = k[len(pre):]
        clean[k] = v
    if "input_shape" in meta:
        w, h, c = [int(x) for x in meta["input_shape"]]
        width, height, channels = width or w, height or h, channels or c
    channels = channels or int(clean["conv1.weight"].shape[1])
    width, height = width or 80, height or 80
    classes = int(meta.get("num_classes", clean["fc2.weight"].shape[0]))
    if classes != int(clean["fc2.weight"].shape[0]):
        raise ValueError("metadata num_classes does not match fc2.weight")
    st = {}
    for name, shp in weights.shapes(classes, channels, width, height):
        if name not in clean:
            raise KeyError(f"checkpoint has no tensor {name!r} (keys: {sorted(clean)[:6]} ...)")
        t = clean[name].detach().to(torch.float32).numpy()
        if tuple(t.shape) != tuple(shp):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(shp)} expected for {classes} classes, {width}x{height}x{channels}")
        st[name] = t
    return weights.pack_blob(st, classes, channels, width, height), classes, width, height, channels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--width", type=int, default=None); ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--unsafe-pickle", action="store_true",
                    help="load with weights_only=False (arbitrary pickle code runs!): only for a checkpoint you wrote yourself; the reference loads with weights_only=True (trex_utils.py:122)")
    a = ap.parse_args()
    import torch
    # weights_only=True like the reference; no silent fallback -- a crafted file could fail the safe path on purpose to reach the pickle loader
    obj = torch.load(a.src, map_location="cpu", weights_only=not a.unsafe_pickle)
    blob, classes, w, h, c = convert(obj, a.width, a.height, a.channels)
    with open(a.dst, "wb") as f:
        f.write(blob)
    print(f"wrote {a.dst}: {classes} classes, {w}x{h}x{c}, {len(blob)} bytes")


if __name__ == "__main__":
    main()
