"""Identity-network (V118_3) weights: deterministic recipe for synthetic weights + the flat blob
format libtrexhip loads (trexhip_load_weights).

Blob = little-endian: 8 x int32 header {magic 'TRXW', version 1, num_classes, width, height, channels, 0, 0}
followed by float32 tensors in PyTorch state_dict order and shapes of
visual_identification_network_torch.py:184-211 (V118_3):
  conv1.weight[16,C,5,5] conv1.bias[16] bn1.weight bn1.bias bn1.running_mean bn1.running_var [16 each]
  conv2.weight[64,16,5,5] conv2.bias[64] bn2.* [64]  conv3.weight[128,64,5,5] conv3.bias[128] bn3.* [128]
  fc1.weight[100, 128*(W/8)*(H/8)] fc1.bias[100] bn4.weight[100] bn4.bias[100] (LayerNorm)
  fc2.weight[classes,100] fc2.bias[classes]
A real TRex checkpoint (<base>_dict.pth) converts with tools/convert_weights.py.
"""
import numpy as np

MAGIC = 0x57585254  # 'TRXW'

TENSORS = [
    # name, shape as function of (classes, channels, flat)
    ("conv1.weight", lambda c, ch, fl: (16, ch, 5, 5)), ("conv1.bias", lambda c, ch, fl: (16,)),
    ("bn1.weight", lambda c, ch, fl: (16,)), ("bn1.bias", lambda c, ch, fl: (16,)),
    ("bn1.running_mean", lambda c, ch, fl: (16,)), ("bn1.running_var", lambda c, ch, fl: (16,)),
    ("conv2.weight", lambda c, ch, fl: (64, 16, 5, 5)), ("conv2.bias", lambda c, ch, fl: (64,)),
    ("bn2.weight", lambda c, ch, fl: (64,)), ("bn2.bias", lambda c, ch, fl: (64,)),
    ("bn2.running_mean", lambda c, ch, fl: (64,)), ("bn2.running_var", lambda c, ch, fl: (64,)),
    ("conv3.weight", lambda c, ch, fl: (128, 64, 5, 5)), ("conv3.bias", lambda c, ch, fl: (128,)),
    ("bn3.weight", lambda c, ch, fl: (128,)), ("bn3.bias", lambda c, ch, fl: (128,)),
    ("bn3.running_mean", lambda c, ch, fl: (128,)), ("bn3.running_var", lambda c, ch, fl: (128,)),
    ("fc1.weight", lambda c, ch, fl: (100, fl)), ("fc1.bias", lambda c, ch, fl: (100,)),
    ("bn4.weight", lambda c, ch, fl: (100,)), ("bn4.bias", lambda c, ch, fl: (100,)),
    ("fc2.weight", lambda c, ch, fl: (c, 100)), ("fc2.bias", lambda c, ch, fl: (c,)),
]


def shapes(num_classes, channels=1, width=80, height=80):
    flat = 128 * (width // 8) * (height // 8)
    return [(n, f(num_classes, channels, flat)) for n, f in TENSORS]


def synthetic_state(num_classes, seed, channels=1, width=80, height=80):
    """Deterministic random weights (numpy PCG64, reproducible anywhere numpy runs): uniform
    +-1/sqrt(fan_in) like PyTorch's default init, non-trivial BN/LayerNorm affine.  Running
    statistics are placeholders here; calibrated values come from the golden fixture."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shp in shapes(num_classes, channels, width, height):
        if name.endswith("running_mean"):
            st[name] = np.zeros(shp, np.float32)
        elif name.endswith("running_var"):
            st[name] = np.ones(shp, np.float32)
        elif name.startswith("bn") and name.endswith("weight"):
            st[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.startswith("bn") and name.endswith("bias"):
            st[name] = rng.uniform(-0.3, 0.3, shp).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else None
            if fan_in is None:   # bias of the preceding layer
                prev = st[name.replace("bias", "weight")]
                fan_in = int(np.prod(prev.shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            st[name] = rng.uniform(-b, b, shp).astype(np.float32)
    st["fc2.weight"] = (st["fc2.weight"] * 8.0).astype(np.float32)   # peaky softmax: a flat one hides errors
    return st


def synthetic_crops(n, seed, channels=1, width=80, height=80):
    """uint8 NHWC crops that look like blobs on black: a bright ellipse with texture."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    out = np.zeros((n, height, width, channels), np.uint8)
    for i in range(n):
        a, b = rng.uniform(12, 30), rng.uniform(4, 10)
        th = rng.uniform(0, np.pi)
        u = (xx - width / 2) * np.cos(th) + (yy - height / 2) * np.sin(th)
        v = -(xx - width / 2) * np.sin(th) + (yy - height / 2) * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1
        tex = rng.integers(40, 200, (height, width, channels))
        out[i][m] = tex[m]
    return out


def pack_blob(state, num_classes, channels=1, width=80, height=80):
    hdr = np.array([MAGIC, 1, num_classes, width, height, channels, 0, 0], np.int32)
    parts = [hdr.tobytes()]
    for name, shp in shapes(num_classes, channels, width, height):
        t = np.ascontiguousarray(state[name], np.float32)
        assert t.shape == tuple(shp), (name, t.shape, shp)
        parts.append(t.tobytes())
    return b"".join(parts)


def unpack_blob(blob):
    """Inverse of pack_blob: -> (state dict, num_classes, channels)."""
    hdr = np.frombuffer(blob[:32], np.int32)
    assert int(hdr[0]) == MAGIC and int(hdr[1]) == 1, "not a TRXW v1 blob"
    classes, width, height, channels = int(hdr[2]), int(hdr[3]), int(hdr[4]), int(hdr[5])
    st, off = {}, 32
    for name, shp in shapes(classes, channels, width, height):
        cnt = int(np.prod(shp))
        st[name] = np.frombuffer(blob[off:off + 4 * cnt], np.float32).reshape(shp).copy()
        off += 4 * cnt
    assert off == len(blob), "blob size does not match its header"
    return st, classes, channels


def synthetic_train_batch(n, seed, classes, channels=1):
    """Training inputs as TRexImageDataset hands them over (visual_recognition_torch.py:158-188): NHWC float32 in [0, 255],
    NOT integer (the augmentation rescales), + integer class labels."""
    rng = np.random.default_rng(seed)
    x = synthetic_crops(n, seed + 7, channels).astype(np.float32)
    x = np.clip(x * rng.uniform(0.85, 1.15, (n, 1, 1, 1)).astype(np.float32) + rng.uniform(0.0, 3.0, x.shape).astype(np.float32) * (x > 0), 0.0, 255.0)
    y = rng.integers(0, classes, n).astype(np.int32)
    return np.ascontiguousarray(x, np.float32), y
