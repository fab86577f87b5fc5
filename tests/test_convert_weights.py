"""tools/convert_weights.py on a checkpoint laid out like the reference writes it (visual_recognition_torch.py:102-117:
{'model': None, 'state_dict': ..., 'metadata': {'input_shape': (W,H,C), 'num_classes', 'model_type', ...}}, keys with the
PermuteAxesWrapper 'model.' prefix and BatchNorm's num_batches_tracked) -- round trip into the blob trexhip_load_weights takes."""
import importlib.util
import os
import numpy as np
import pytest
import torch
from trex_amd import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("convert_weights", os.path.join(ROOT, "tools", "convert_weights.py"))
cw = importlib.util.module_from_spec(spec); spec.loader.exec_module(cw)


def reference_style_checkpoint(classes, channels):
    st = weights.synthetic_state(classes, 7, channels=channels)
    sd = {"model." + k: torch.from_numpy(v) for k, v in st.items()}
    for i in (1, 2, 3):
        sd[f"model.bn{i}.num_batches_tracked"] = torch.tensor(12345)
    ck = {"model": None, "state_dict": sd,
          "metadata": {"input_shape": (80, 80, channels), "num_classes": classes, "video_name": "x", "epoch": 3, "uniqueness": 0.9, "model_type": "v118_3"}}
    return st, ck


@pytest.mark.parametrize("classes,channels", [(8, 1), (100, 3)])
def test_round_trip(tmp_path, classes, channels):
    st, ck = reference_style_checkpoint(classes, channels)
    path = tmp_path / "model_dict.pth"
    torch.save(ck, path)
    obj = torch.load(path, map_location="cpu", weights_only=False)
    blob, c, w, h, ch = cw.convert(obj)
    assert (c, w, h, ch) == (classes, 80, 80, channels)
    assert blob == weights.pack_blob(st, classes, channels)
    # a bare state_dict (no wrapper, no prefix) converts to the same blob
    blob2, *_ = cw.convert({k: torch.from_numpy(v) for k, v in st.items()})
    assert blob2 == blob


def test_rejects_other_networks_and_bad_shapes():
    st, ck = reference_style_checkpoint(8, 1)
    ck["metadata"]["model_type"] = "v200"
    with pytest.raises(ValueError):
        cw.convert(ck)
    st, ck = reference_style_checkpoint(8, 1)
    ck["state_dict"]["model.fc1.weight"] = torch.zeros(100, 7)
    with pytest.raises(ValueError):
        cw.convert(ck)
    st, ck = reference_style_checkpoint(8, 1)
    del ck["state_dict"]["model.bn4.bias"]
    with pytest.raises(KeyError):
        cw.convert(ck)
