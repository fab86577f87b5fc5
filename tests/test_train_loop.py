"""Host logic of the training loop (trex_amd/train_loop.py): the learning-rate schedule against torch's own ReduceLROnPlateau (CPU), the
epoch loop against a recording stand-in of the trainer (CPU) and, on the GPU, a short real run."""
import numpy as np
import pytest
import torch

from trex_amd import train_loop


@pytest.mark.parametrize("kw", [dict(), dict(patience=2), dict(patience=0, factor=0.5), dict(cooldown=2, patience=1), dict(threshold=1e-2, threshold_mode="abs"),
                                dict(mode="max", patience=1), dict(min_lr=3e-4, patience=1)])
def test_reduce_lr_on_plateau_equals_torch(kw):
    rng = np.random.default_rng(5)
    for trial in range(20):
        seq = np.abs(np.cumsum(rng.normal(0, 0.05, 60)) + np.linspace(1.0, 0.6, 60)) if trial % 2 else rng.uniform(0.2, 1.0, 60)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-3)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, **({"mode": "min", "factor": 0.1, "patience": 5} | kw))
        mine = train_loop.ReduceLROnPlateau(1e-3, **({"mode": "min", "factor": 0.1, "patience": 5} | kw))
        for v in seq:
            ref.step(float(v))
            lr = mine.step(float(v))
            assert lr == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12), (trial, v)


class FakeTrainer:
    def __init__(self):
        self.calls, self.lrs = [], []
    def step(self, x, y):
        self.calls.append(("step", x.shape, y.dtype))
        return 1.0 / (1 + len(self.calls)), int(x.shape[0] // 2)
    def evaluate(self, x, y):
        self.calls.append(("eval", x.shape, y.dtype))
        return 0.5, int(x.shape[0])
    def set_lr(self, lr):
        self.lrs.append(lr)


class Recorder:
    def __init__(self, stop_after=None):
        self.batches, self.epochs, self.stop_training, self.stop_after = [], [], False, stop_after
    def on_batch_end(self, batch, logs):
        self.batches.append((batch, logs))
    def on_epoch_end(self, epoch, logs):
        self.epochs.append((epoch, logs))
        if self.stop_after is not None and epoch >= self.stop_after:
            self.stop_training = True


def loader(nb, n, classes=3):
    rng = np.random.default_rng(0)
    return [(rng.uniform(0, 255, (n, 80, 80, 1)).astype(np.float32), rng.integers(0, classes, n)) for _ in range(nb)]


def test_epoch_loop_calls_what_train_calls():
    tr, cb = FakeTrainer(), Recorder(stop_after=2)
    sched = train_loop.ReduceLROnPlateau(1e-3, patience=0)
    hist = train_loop.train(tr, loader(3, 4), loader(2, 5), cb, sched, {"epochs": 10})
    assert len(hist) == 3 and len(cb.epochs) == 3 and len(cb.batches) == 9          # stop_training after the third epoch (train() :1262)
    assert [b for b, _ in cb.batches[:3]] == [0, 1, 2] and set(cb.batches[0][1]) == {"loss", "acc"}
    assert set(cb.epochs[0][1]) == {"val_loss", "val_acc", "val_precision", "val_recall"} and cb.epochs[0][1]["val_acc"] == 1.0
    assert [c[0] for c in tr.calls[:5]] == ["step", "step", "step", "eval", "eval"] and tr.calls[0][2] == np.int32
    assert len(tr.lrs) == 3 and tr.lrs[0] == 1e-3 and tr.lrs[1] == pytest.approx(1e-4)      # constant val_loss: patience 0 -> reduced at the second epoch
    # no validation data: the epoch log carries loss / acc (train() :1249-1258); abort() is honoured
    tr2, cb2 = FakeTrainer(), Recorder()
    hist2 = train_loop.train(tr2, loader(2, 4), [], cb2, None, {"epochs": 5}, abort=lambda: len(cb2.epochs) >= 2)
    assert len(hist2) == 2 and set(cb2.epochs[0][1]) == {"loss", "acc"} and not tr2.lrs
    with pytest.raises(ValueError):
        train_loop.train(FakeTrainer(), [(np.zeros((4, 80, 80), np.float32), np.zeros(4, np.int64))], [], Recorder(), None, {"epochs": 1})


@pytest.mark.gpu
def test_short_real_run_learns_and_keeps_the_schedule():
    from trex_amd import capi, weights
    classes, n = 4, 32
    state = weights.synthetic_state(classes, 9)
    p = capi.default_params(64, 64)
    p.max_batch = 1
    seg = capi.Segmenter(p)
    tr = capi.Trainer(seg, weights.pack_blob(state, classes), max_batch=n, lr=1e-3, seed=1)
    x, y = weights.synthetic_train_batch(n, 3, classes)
    cb = Recorder()
    sched = train_loop.ReduceLROnPlateau(1e-3, patience=5)
    hist = train_loop.train(tr, [(x, y)] * 3, [(x, y)], cb, sched, {"epochs": 6})
    assert len(hist) == 6 and tr.steps == 18
    assert hist[-1]["val_loss"] < hist[0]["val_loss"] and hist[-1]["val_acc"] >= hist[0]["val_acc"]
    assert all(np.isfinite(h["loss"]) for h in hist)
    tr.close(); seg.close()


def test_float_labels_are_refused_like_the_reference_asserts():
    # visual_recognition_torch.py:1109-1110 asserts integer class labels per batch; a float array would be cast silently otherwise
    with pytest.raises(ValueError):
        train_loop.train(FakeTrainer(), [(np.zeros((4, 80, 80, 1), np.float32), np.zeros(4, np.float32))], [], Recorder(), None, {"epochs": 1})
