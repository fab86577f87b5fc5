"""Training step of the identity network on the GPU (trexhip_train_step_device) against
 (1) the vectors the reference's own module + torch.optim.Adam produced (tests/golden/cnn_train_v118_3.npz), with the dropout
     masks the reference drew injected, and
 (2) the CPU restatement (oracle/cnn_train_oracle.py) at the batch size VINetwork uses.
fp32 on both sides; the bar is relative to each tensor's largest gradient (sums of 1e4..1e6 terms in another order), stated below."""
import os
import numpy as np
import pytest
import torch

from trex_amd import capi, weights
from oracle import cnn_train_oracle as tro, cnn_oracle

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "cnn_train_v118_3.npz")
SAMPLE = {"conv3.weight": 7, "fc1.weight": 53}
NAMES = [n for n, _ in weights.TENSORS]
GRAD_RTOL = 2e-4          # |g - g_ref| <= GRAD_RTOL * max|g_ref| per tensor
CONV_BIAS = ("conv1.bias", "conv2.bias", "conv3.bias")   # feed a BatchNorm: true gradient 0, only rounding noise on both sides


def sample(name, arr):
    return np.asarray(arr).reshape(-1)[::SAMPLE.get(name, 1)]


def make_seg():
    p = capi.default_params(64, 64)
    p.max_batch = 1
    seg = capi.Segmenter(p)
    return seg


def pack_masks(m):
    return np.concatenate([np.ascontiguousarray(m[k], np.uint8).reshape(-1) for k in ("d1", "d2", "d3", "d4")])


def read_all(tr, classes, ch, kind):
    return {n: tr.read(i, kind, shp) for i, (n, shp) in enumerate(weights.shapes(classes, ch))}


def step(tr, x, y, masks, host=False):
    if host:      # trexhip_train_step: host arrays, as the data loader yields them
        return tr.step(x, y, pack_masks(masks) if masks is not None else None)
    dx = torch.from_numpy(x).cuda()
    dy = torch.from_numpy(y.astype(np.int32)).cuda()
    dm = torch.from_numpy(pack_masks(masks)).cuda() if masks is not None else None
    out = tr.step_device(dx.data_ptr(), dy.data_ptr(), x.shape[0], dm.data_ptr() if dm is not None else 0)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("precision", [0, 1])      # 0: conv2 / conv3 forward + data gradients in fp16 two-piece split arithmetic (default), 1: exact fp32 MFMA
@pytest.mark.parametrize("name", ["a", "b"])
def test_training_steps_equal_the_reference_module(name, precision):
    fx = np.load(FIX)
    classes, ch, n, steps, seed = [int(v) for v in fx[f"{name}/meta"]]
    lr = float(fx[f"{name}/lr"][0])
    state = weights.synthetic_state(classes, seed, channels=ch)
    seg = make_seg()
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=n, lr=lr, precision=precision)
    for s in range(steps):
        x, y = weights.synthetic_train_batch(n, seed + 100 * s, classes, ch)
        masks = {t: fx[f"{name}/mask{s}/{t}"] for t in ("d1", "d2", "d3", "d4")}
        loss, correct = step(tr, x, y, masks, host=(name == "b"))
        ref = float(fx[f"{name}/loss{s}"][0])
        assert abs(loss - ref) <= 5e-5 * max(1.0, abs(ref)), (s, loss, ref)
        assert correct == int(fx[f"{name}/correct{s}"][0])
        if s == 0:
            g = read_all(tr, classes, ch, 1)
            for k in tro.TRAINABLE:
                r = fx[f"{name}/grad0/{k}"]
                got = sample(k, g[k])
                if k in CONV_BIAS:
                    assert np.abs(got).max() <= 1e-4 * max(np.abs(fx[f"{name}/grad0/{k.replace('bias', 'weight')}"]).max(), 1.0), k
                    continue
                tol = GRAD_RTOL * float(np.abs(r).max()) + 1e-9
                assert np.abs(got - r).max() <= tol, (k, float(np.abs(got - r).max()), tol)
    assert tr.steps == steps
    final = read_all(tr, classes, ch, 0)
    for k in tro.TRAINABLE + tro.BUFFERS:
        r = fx[f"{name}/final/{k}"]
        got = sample(k, final[k])
        if k in tro.BUFFERS:
            # after several steps the statistics see weights that differ by the Adam noise described below; the single-step test
            # against the restatement holds them to 1e-4
            assert np.abs(got - r).max() <= 1e-3 * max(1.0, np.abs(r).max()), (k, float(np.abs(got - r).max()))
            continue
        err = np.abs(got - r)
        # Adam's update is lr * m / (sqrt(v) + eps): +-lr per step whatever the gradient's size, so elements whose gradient is
        # rounding noise (conv biases; weights that see only black pixels) move by up to lr per step in an arbitrary direction
        assert err.max() <= 2.0 * steps * lr + 1e-6 * np.abs(r).max(), (k, float(err.max()))
        if k not in CONV_BIAS:
            frac = float(np.mean(err <= 0.05 * lr + 1e-6 * np.abs(r)))
            assert frac >= 0.99, (k, frac)
    tr.close()
    seg.close()


def test_one_step_at_vinetwork_batch_size_equals_oracle_and_is_deterministic():
    classes, ch, n, seed, lr = 100, 1, 128, 77, 1e-3
    state = weights.synthetic_state(classes, seed, channels=ch)
    x, y = weights.synthetic_train_batch(n, seed + 1, classes, ch)
    rng = np.random.default_rng(seed)
    masks = {"d1": rng.random((n, 16)) >= 0.05, "d2": rng.random((n, 64)) >= 0.05, "d3": rng.random((n, 128)) >= 0.05, "d4": rng.random((n, 100)) >= 0.05}
    adam = tro.new_adam_state(state)
    new, loss_ref, correct_ref, grads = tro.train_step(state, adam, x, y, masks, lr, threads=16)
    seg = make_seg()
    runs = []
    for _ in range(2):
        tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=n, lr=lr)
        loss, correct = step(tr, x, y, masks)
        runs.append((loss, correct, read_all(tr, classes, ch, 1), read_all(tr, classes, ch, 0), tr.export()))
        tr.close()
    loss, correct, g, p, blob = runs[0]
    assert abs(loss - loss_ref) <= 5e-5 * max(1.0, abs(loss_ref)) and correct == correct_ref
    for k in tro.TRAINABLE:
        if k in CONV_BIAS:
            continue
        tol = GRAD_RTOL * float(np.abs(grads[k]).max()) + 1e-9
        assert np.abs(g[k] - grads[k]).max() <= tol, (k, float(np.abs(g[k] - grads[k]).max()), tol)
    for k in tro.BUFFERS:
        assert np.abs(p[k] - new[k]).max() <= 1e-4 * max(1.0, np.abs(new[k]).max()), k
    # every reduction has a fixed order: a second trainer on the same inputs gives the same bits
    assert runs[1][0] == loss
    for k in NAMES:
        assert np.array_equal(runs[1][2][k], g[k]) and np.array_equal(runs[1][3][k], p[k]), k
    # the exported blob is what trexhip_load_weights takes; the inference path on it equals the eval-mode restatement
    st2, c2, ch2 = weights.unpack_blob(blob)
    assert c2 == classes and ch2 == ch
    for k in NAMES:
        assert np.array_equal(st2[k], p[k]), k
    seg.load_weights(blob)
    crops = weights.synthetic_crops(64, 9)
    probs = seg.probabilities(crops)
    ref_probs, _ = cnn_oracle.predict(st2, crops, threads=16)
    assert np.abs(probs - ref_probs).max() <= 1e-4
    seg.close()


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("n,classes,ch", [(1, 2, 1), (67, 257, 3), (33, 1024, 1)])
def test_ragged_batches_and_class_counts_equal_oracle(n, classes, ch, precision):
    # a last batch of an epoch is whatever is left (DataLoader drop_last=False, visual_recognition_torch.py:1394-1400); a single sample is legal
    # for BatchNorm2d because the statistics run over the pixels as well
    seed, lr = 900 + n, 1e-3
    state = weights.synthetic_state(classes, seed, channels=ch)
    x, y = weights.synthetic_train_batch(n, seed + 1, classes, ch)
    rng = np.random.default_rng(seed)
    masks = {"d1": rng.random((n, 16)) >= 0.05, "d2": rng.random((n, 64)) >= 0.05, "d3": rng.random((n, 128)) >= 0.05, "d4": rng.random((n, 100)) >= 0.05}
    adam = tro.new_adam_state(state)
    new, loss_ref, correct_ref, grads = tro.train_step(state, adam, x, y, masks, lr, threads=16)
    seg = make_seg()
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=max(n, 4), lr=lr, precision=precision)
    loss, correct = step(tr, x, y, masks)
    assert abs(loss - loss_ref) <= 5e-5 * max(1.0, abs(loss_ref)) and correct == correct_ref
    g = read_all(tr, classes, ch, 1)
    p = read_all(tr, classes, ch, 0)
    for k in tro.TRAINABLE:
        if k in CONV_BIAS:
            continue
        tol = GRAD_RTOL * float(np.abs(grads[k]).max()) + 1e-9
        assert np.abs(g[k] - grads[k]).max() <= tol, (k, float(np.abs(g[k] - grads[k]).max()), tol)
    for k in tro.BUFFERS:
        assert np.abs(p[k] - new[k]).max() <= 1e-4 * max(1.0, np.abs(new[k]).max()), k
    tr.close()
    seg.close()


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("s2,s3", [(1e3, 1e-3), (1e-4, 3e2)])
def test_split_arithmetic_follows_the_operands_ranges(s2, s3, precision):
    # precision 0 scales every staged patch and each layer's weights by a power of two before splitting them into fp16 pieces.  Convolution
    # weights (and the running statistics behind them) 1000 times larger / smaller move the operands of conv2 / conv3 across the fp16 range;
    # the forward pass is continuous in its operands (no arg-max to flip), so the loss must follow the restatement as closely as unscaled
    classes, ch, n = 10, 1, 40
    state = dict(weights.synthetic_state(classes, 71, channels=ch))
    rng = np.random.default_rng(71)
    for k in ("bn1", "bn2", "bn3"):
        state[k + ".running_mean"] = rng.uniform(-0.5, 0.5, state[k + ".running_mean"].shape).astype(np.float32)
        state[k + ".running_var"] = rng.uniform(0.5, 2.0, state[k + ".running_var"].shape).astype(np.float32)
    for k, sc in (("2", s2), ("3", s3)):
        state[f"conv{k}.weight"] = (state[f"conv{k}.weight"] * np.float32(sc)).astype(np.float32)
        state[f"conv{k}.bias"] = (state[f"conv{k}.bias"] * np.float32(sc)).astype(np.float32)
        state[f"bn{k}.running_mean"] = (state[f"bn{k}.running_mean"] * np.float32(sc)).astype(np.float32)
        state[f"bn{k}.running_var"] = (state[f"bn{k}.running_var"] * np.float32(sc) ** 2).astype(np.float32)
    x, y = weights.synthetic_train_batch(n, 72, classes, ch)
    x = np.round(x)
    logits = cnn_oracle.forward_logits(state, x.astype(np.uint8), threads=16)
    z = logits - logits.max(1, keepdims=True)
    loss_ref = float(np.mean(np.log(np.exp(z.astype(np.float64)).sum(1)) - z[np.arange(n), y]))
    seg = make_seg()
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=64, lr=1e-3, precision=precision)
    loss, correct = tr.evaluate(x, y)
    assert abs(loss - loss_ref) <= 5e-5 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    assert correct == int((logits.argmax(1) == y).sum())
    tr.close(); seg.close()


def test_eval_mode_loss_equals_the_eval_restatement():
    # validation batches: model.eval() -> running statistics, no dropout; loss = CrossEntropyLoss(logits, targets) (train() :1171-1190)
    classes, ch, n = 100, 1, 77
    state = weights.synthetic_state(classes, 61, channels=ch)
    rng = np.random.default_rng(61)
    for k in ("bn1", "bn2", "bn3"):                                   # non-trivial running statistics
        state[k + ".running_mean"] = rng.uniform(-0.5, 0.5, state[k + ".running_mean"].shape).astype(np.float32)
        state[k + ".running_var"] = rng.uniform(0.5, 2.0, state[k + ".running_var"].shape).astype(np.float32)
    x, y = weights.synthetic_train_batch(n, 62, classes, ch)
    x = np.round(x)                                                   # the eval restatement takes 8-bit crops
    logits = cnn_oracle.forward_logits(state, x.astype(np.uint8), threads=16)
    z = logits - logits.max(1, keepdims=True)
    lse = np.log(np.exp(z.astype(np.float64)).sum(1))
    loss_ref = float(np.mean(lse - z[np.arange(n), y]))
    correct_ref = int((logits.argmax(1) == y).sum())
    seg = make_seg()
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=128, lr=1e-3)
    before = tr.export()
    loss, correct = tr.evaluate(x, y)
    assert abs(loss - loss_ref) <= 5e-5 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    assert correct == correct_ref
    assert tr.export() == before and tr.steps == 0                    # evaluation changes nothing
    tr.close(); seg.close()


def test_library_drawn_masks_and_argument_checks():
    classes, ch, n = 10, 3, 9
    state = weights.synthetic_state(classes, 5, channels=ch)
    seg = make_seg()
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=16, lr=1e-3, seed=123)
    losses = []
    for s in range(4):
        x, y = weights.synthetic_train_batch(n, 40, classes, ch)      # the same batch: the loss must fall
        loss, correct = step(tr, x, y, None)
        assert np.isfinite(loss) and 0 <= correct <= n
        losses.append(loss)
    assert losses[-1] < losses[0]
    dx = torch.zeros((17, 80, 80, ch), device="cuda")
    dy = torch.zeros(17, dtype=torch.int32, device="cuda")
    with pytest.raises(capi.TrexHipError):
        tr.step_device(dx.data_ptr(), dy.data_ptr(), 17)              # n > max_batch
    with pytest.raises(capi.TrexHipError):
        tr.step_device(dx.data_ptr(), dy.data_ptr(), 0)
    with pytest.raises(capi.TrexHipError):
        tr.set_lr(0.0)
    # a class index out of range in device memory (the reference asserts it before touching the model, visual_recognition_torch.py:1109-1112):
    # the device refuses that step AND the steps queued behind it, the next synchronising call reports it, and nothing has changed --
    # parameters, Adam moments, running statistics, step count
    before = [read_all(tr, classes, ch, kind) for kind in (0, 2, 3)]
    steps_before = tr.steps
    x, y = weights.synthetic_train_batch(n, 41, classes, ch)
    ybad = y.copy(); ybad[3] = classes
    dxb, dyb, dyg = torch.from_numpy(x).cuda(), torch.from_numpy(ybad.astype(np.int32)).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()
    tr.step_device(dxb.data_ptr(), dyb.data_ptr(), n, 0, want_loss=False)        # refused on the device, not yet reported
    tr.step_device(dxb.data_ptr(), dyg.data_ptr(), n, 0, want_loss=False)        # queued behind it: refused as well
    with pytest.raises(capi.TrexHipError, match="refused"):
        tr.step_device(dxb.data_ptr(), dyg.data_ptr(), n, 0, want_loss=True)
    assert tr.steps == steps_before
    after = [read_all(tr, classes, ch, kind) for kind in (0, 2, 3)]
    for b, a in zip(before, after):
        for name in b:
            assert np.array_equal(b[name], a[name]), name
    loss, correct = step(tr, x, y, None)                              # the trainer goes on where it was
    assert np.isfinite(loss) and tr.steps == steps_before + 1
    tr.export()
    with pytest.raises(capi.TrexHipError):
        tr.step(x, ybad)                                              # host entry point: checked before anything is uploaded
    with pytest.raises(ValueError):
        tr.step(x[:, :40], y)                                         # wrong image size: refused before the raw pointer is handed over
    with pytest.raises(ValueError):
        tr.step(x, y.astype(np.float32))
    with pytest.raises(ValueError):
        tr.step(x, y, np.ones(5, np.uint8))
    tr.close()
    with pytest.raises(capi.TrexHipError):
        capi.Trainer(seg, weights.pack_blob(state, classes, ch)[:-4], max_batch=16)
    seg.close()


def test_precision_outside_0_1_is_refused():
    # trexhip_train_params.precision selects the arithmetic of the big convolutions: anything but 0 / 1 is an error, not a silent choice
    state = weights.synthetic_state(8, 3)
    p = capi.default_params(64, 64)
    p.max_batch = 1
    seg = capi.Segmenter(p)
    for bad in (2, -1, 7):
        with pytest.raises(capi.TrexHipError):
            capi.Trainer(seg, weights.pack_blob(state, 8, 1), max_batch=4, lr=1e-3, precision=bad)
    seg.close()


def test_step_is_the_same_bits_on_every_kind_of_caller_stream():
    # the step forks onto a second stream and joins again (weight gradients beside the data-gradient chain); its events go through the null
    # stream for the hipStreamLegacy handle and the fork is skipped for the per-thread handle: the results must not depend on the stream kind
    classes, ch, n = 10, 1, 24
    state = weights.synthetic_state(classes, 5, channels=ch)
    x, y = weights.synthetic_train_batch(n, 6, classes, ch)
    blob = weights.pack_blob(state, classes, ch)
    side = torch.cuda.Stream()
    HIP_STREAM_PER_THREAD = 2
    results = []
    for kind, stream in (("own", None), ("legacy", capi.HIP_STREAM_LEGACY), ("torch side stream", side.cuda_stream), ("per-thread", HIP_STREAM_PER_THREAD)):
        p = capi.default_params(64, 64)
        p.max_batch = 1
        seg = capi.Segmenter(p, stream=stream)
        tr = capi.Trainer(seg, blob, max_batch=n, lr=1e-3, seed=11)
        dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        for _ in range(3):                         # library-drawn masks, three steps queued back to back
            tr.step_device(dx.data_ptr(), dy.data_ptr(), n, 0, want_loss=False)
        loss, correct = tr.step_device(dx.data_ptr(), dy.data_ptr(), n, 0)
        seg.synchronize()
        torch.cuda.synchronize()
        results.append((kind, loss, correct, read_all(tr, classes, ch, 0), read_all(tr, classes, ch, 1)))
        tr.close(); seg.close()
    for kind, loss, correct, p_, g_ in results[1:]:
        assert loss == results[0][1] and correct == results[0][2], kind
        for k in NAMES:
            assert np.array_equal(p_[k], results[0][3][k]) and np.array_equal(g_[k], results[0][4][k]), (kind, k)
