"""CPU restatement of ONE training step of the identity network (TEST INFRASTRUCTURE ONLY).

Follows, function by function:
  train()                  Application/src/tracker/python/visual_recognition_torch.py:1036-1160
      model.train(); outputs = model(inputs); loss = CrossEntropyLoss(outputs, targets) (:1137-1142, criterion :1420);
      loss.backward(); optimizer.step(); optimizer.zero_grad() (:1156-1158 -- the non-AMP branch: GradScaler / autocast are
      enabled for device == 'cuda' only, :1066-1072, so fp32 is the reference's own arithmetic on every other device);
      optimizer = Adam(model.parameters(), lr=learning_rate) with torch's defaults (:1421)
  TRexImageDataset         :158-188   inputs are NHWC float32 in [0, 255] (augmented, not integer), labels int
  PermuteAxesWrapper       visual_identification_network_torch.py:618-644  (NHWC->NCHW; Normalize is a pass-through :19-26)
  V118_3.forward           visual_identification_network_torch.py:184-258, train mode:
      [conv5x5 'same' -> BatchNorm2d(batch statistics, running stats updated with momentum 0.1) -> ReLU -> MaxPool2 ->
       Dropout2d(0.05)] x3 -> flatten (NCHW order) -> fc1 -> LayerNorm(100) -> ReLU -> Dropout(0.05) -> fc2

The dropout masks are INPUTS here (keep = 1): torch draws `noise = bernoulli(1 - p) / (1 - p)` and multiplies; with the mask
given, the step is a deterministic function.  Plain fp32 torch functional ops + autograd, Adam written out like
torch.optim.adam._single_tensor_adam.  Pinned against tests/golden/cnn_train_v118_3.npz, which holds what the reference's own
module + torch.optim.Adam + nn.CrossEntropyLoss produced (generator: tests/golden/make_train_fixture.py).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

EPS_BN = 1e-5
EPS_LN = 1e-5
P_DROP = 0.05
BN_MOMENTUM = 0.1

TRAINABLE = ["conv1.weight", "conv1.bias", "bn1.weight", "bn1.bias",
             "conv2.weight", "conv2.bias", "bn2.weight", "bn2.bias",
             "conv3.weight", "conv3.bias", "bn3.weight", "bn3.bias",
             "fc1.weight", "fc1.bias", "bn4.weight", "bn4.bias", "fc2.weight", "fc2.bias"]
BUFFERS = ["bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var", "bn3.running_mean", "bn3.running_var"]


def new_adam_state(state):
    return {"step": 0, "m": {k: np.zeros_like(state[k]) for k in TRAINABLE}, "v": {k: np.zeros_like(state[k]) for k in TRAINABLE}}


def forward_backward(state, x_nhwc, targets, masks, threads=None, dtype=torch.float32):
    """-> loss (float), correct (int), grads {name: ndarray}, new running stats {name: ndarray}, logits.
    dtype=torch.float64 gives the reference point for "which of two fp32 results is closer" questions (near-ties in the max-pool
    arg-max and at the ReLU flip with fp32 summation order; tools/fuzz_train.py)."""
    if threads:
        torch.set_num_threads(threads)
    t = {k: torch.from_numpy(np.ascontiguousarray(state[k], np.float32)).clone().to(dtype) for k in TRAINABLE + BUFFERS}
    for k in TRAINABLE:
        t[k].requires_grad_(True)
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc, np.float32)).to(dtype).permute(0, 3, 1, 2)
    y = torch.from_numpy(np.asarray(targets).astype(np.int64))
    scale = 1.0 - P_DROP
    for i in (1, 2, 3):
        x = F.conv2d(x, t[f"conv{i}.weight"], t[f"conv{i}.bias"], padding=2)
        if i == 1:
            # V118_3.forward calls self.bn1(x.contiguous()) (:222): NCHW-contiguous input.  This matters numerically: torch's CPU
            # batch-norm kernel for channels_last fp32 input loses ~3 digits on conv1's un-normalised outputs (|mean| >> std),
            # the NCHW kernel does not (measured against float64: 6e-4 vs 1.4e-6 absolute on the BN output)
            x = x.contiguous()
        x = F.batch_norm(x, t[f"bn{i}.running_mean"], t[f"bn{i}.running_var"], t[f"bn{i}.weight"], t[f"bn{i}.bias"],
                         training=True, momentum=BN_MOMENTUM, eps=EPS_BN)
        x = F.relu(x)
        x = F.max_pool2d(x, 2)
        noise = torch.from_numpy(np.asarray(masks[f"d{i}"]).astype(np.float32)).div_(scale).to(dtype)       # [n, C]
        x = x * noise[:, :, None, None]
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, t["fc1.weight"], t["fc1.bias"])
    x = F.layer_norm(x, (100,), t["bn4.weight"], t["bn4.bias"], eps=EPS_LN)
    x = F.relu(x)
    x = x * torch.from_numpy(np.asarray(masks["d4"]).astype(np.float32)).div_(scale).to(dtype)
    logits = F.linear(x, t["fc2.weight"], t["fc2.bias"])
    loss = F.cross_entropy(logits, y)
    loss.backward()
    grads = {k: t[k].grad.numpy().copy() for k in TRAINABLE}
    stats = {k: t[k].detach().numpy().copy() for k in BUFFERS}
    correct = int((logits.argmax(1) == y).sum())
    return float(loss.detach()), correct, grads, stats, logits.detach().numpy()


def adam_update(state, adam, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad), fp32 element arithmetic like _single_tensor_adam."""
    adam["step"] += 1
    step = adam["step"]
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    out = dict(state)
    for k in TRAINABLE:
        g = torch.from_numpy(grads[k])
        m = torch.from_numpy(adam["m"][k])
        v = torch.from_numpy(adam["v"][k])
        p = torch.from_numpy(np.ascontiguousarray(state[k], np.float32).copy())
        m.lerp_(g, 1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)
        out[k] = p.numpy()
    return out


def train_step(state, adam, x_nhwc, targets, masks, lr, threads=None):
    """One optimizer step.  Returns (new state incl. running stats, loss, correct, grads)."""
    loss, correct, grads, stats, _ = forward_backward(state, x_nhwc, targets, masks, threads)
    new = adam_update(state, adam, grads, lr)
    new.update(stats)
    return new, loss, correct, grads
