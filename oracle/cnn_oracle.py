"""CPU restatement of the identity network forward pass (TEST INFRASTRUCTURE ONLY).

Follows, function by function:
  predict_numpy            Application/src/tracker/python/visual_recognition_torch.py:290-352
      x = float32(uint8 NHWC), NO scaling (:337); softmax over dim 1 (:333-345)
  PermuteAxesWrapper       visual_identification_network_torch.py:618-644  (NHWC->NCHW; Normalize is a pass-through :19-26)
  V118_3.forward           visual_identification_network_torch.py:213-258
      conv5x5 'same' -> BatchNorm2d(eval) -> ReLU -> MaxPool2 (x3), flatten in NCHW order,
      fc1 -> LayerNorm(100) -> ReLU -> fc2 ; dropout inactive in eval mode (:315)

Pinned against vectors produced by the reference's own module (tests/golden/cnn_v118_3_*.npz,
generator tests/golden/make_cnn_fixtures.py); plain fp32 torch functional ops, no nn.Module reuse.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS_BN = 1e-5    # nn.BatchNorm2d default
EPS_LN = 1e-5    # nn.LayerNorm default


def forward_logits(state, crops_u8, threads=None):
    """state: dict name -> float32 ndarray (PyTorch state_dict names of V118_3); crops: uint8 (N,H,W,C)."""
    if threads:
        torch.set_num_threads(threads)
    t = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in state.items()}
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(crops_u8)).to(torch.float32)   # (N,H,W,C), values 0..255
        x = x.permute(0, 3, 1, 2)                                                 # NCHW
        for i in (1, 2, 3):
            x = F.conv2d(x, t[f"conv{i}.weight"], t[f"conv{i}.bias"], padding=2)
            x = F.batch_norm(x, t[f"bn{i}.running_mean"], t[f"bn{i}.running_var"], t[f"bn{i}.weight"],
                             t[f"bn{i}.bias"], training=False, eps=EPS_BN)
            x = F.relu(x)
            x = F.max_pool2d(x, 2)
        x = x.reshape(x.shape[0], -1)                                             # NCHW flatten: c*100 + h*10 + w
        x = F.linear(x, t["fc1.weight"], t["fc1.bias"])
        x = F.layer_norm(x, (100,), t["bn4.weight"], t["bn4.bias"], eps=EPS_LN)
        x = F.relu(x)
        x = F.linear(x, t["fc2.weight"], t["fc2.bias"])
    return x.numpy()


def predict(state, crops_u8, threads=None):
    logits = forward_logits(state, crops_u8, threads)
    z = logits - logits.max(1, keepdims=True)
    e = np.exp(z.astype(np.float64))
    return (e / e.sum(1, keepdims=True)).astype(np.float32), logits


def batch_size_rule(n_ids):
    """VINetwork batch size (ml/VisualIdentification.cpp:112-118): max(N_ids,64) -> next pow2 if <128 else 128."""
    b = max(int(n_ids), 64)
    if b < 128:
        p = 1
        while p < b:
            p <<= 1
        return p
    return 128


def transform_results(n_images, indexes, values):
    """VINetwork::transform_results (ml/VisualIdentification.cpp:809-830): N x M flat, missing rows = -1."""
    m = values.shape[1] if len(values) else 0
    out = np.full((n_images, m), -1.0, np.float32)
    for row, idx in zip(values, indexes):
        out[int(idx)] = row
    return out.reshape(-1)
