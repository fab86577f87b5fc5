"""Host side of the identity network's training, with the interface of the reference's Python (the reference's host side of this path is
Python too): the epoch loop of train() (Application/src/tracker/python/visual_recognition_torch.py:1036-1283) on top of
capi.Trainer -- whose step / evaluate replace the loop's body (:1137-1158 and :1171-1190) -- and the learning-rate schedule it is handed
(optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.1, patience=5), :1425).

What stays with the caller, as in the reference: the loaders (any iterable of (inputs NHWC float32 in [0, 255], integer targets) with a
len(), e.g. the reference's own DataLoader over TRexImageDataset with its augmentation, :158-188, :1325-1411) and the callback object
(ValidationCallback, :355-560: per-class accuracy, uniqueness, early stopping), used through the same three members train() uses:
on_batch_end(batch, logs), on_epoch_end(epoch, logs), stop_training.  No torch in here: batches may be numpy arrays or anything
np.asarray() accepts (torch CPU tensors included).
"""
import numpy as np


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau for one learning rate (pinned against torch's in tests/test_train_loop.py).
    step(metric) -> the learning rate to use from now on."""

    def __init__(self, lr, mode="min", factor=0.1, patience=5, threshold=1e-4, threshold_mode="rel", cooldown=0, min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        if mode not in ("min", "max") or threshold_mode not in ("rel", "abs"):
            raise ValueError("mode / threshold_mode")
        self.lr, self.mode, self.factor, self.patience = float(lr), mode, factor, patience
        self.threshold, self.threshold_mode, self.cooldown, self.min_lr, self.eps = threshold, threshold_mode, cooldown, min_lr, eps
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0

    def _is_better(self, a, best):
        if self.mode == "min" and self.threshold_mode == "rel":
            return a < best * (1.0 - self.threshold)
        if self.mode == "min":
            return a < best - self.threshold
        if self.threshold_mode == "rel":
            return a > best * (self.threshold + 1.0)
        return a > best + self.threshold

    def step(self, metric):
        current = float(metric)
        self.last_epoch += 1
        if self._is_better(current, self.best):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0
        return self.lr

    def get_last_lr(self):
        return [self.lr]


def train(trainer, train_loader, val_loader, callback, scheduler, settings, abort=lambda: False, log=None):
    """The loop of train(model, train_loader, val_loader, criterion, optimizer, callback, scheduler, settings, device) with model +
    criterion + optimizer = `trainer` (capi.Trainer).  settings["epochs"] epochs; per batch one optimizer step and
    callback.on_batch_end(batch, {'loss', 'acc'}); per epoch the validation pass in eval mode (when val_loader is not empty), scheduler.step(val_loss)
    -> trainer.set_lr, callback.on_epoch_end(epoch, logs); stops when callback.stop_training or abort() is set.  Returns the history."""
    history = []
    best_val_acc = 0.0
    for epoch in range(int(settings["epochs"])):
        running_loss = 0.0
        running_acc = 0.0
        n_batches = 0
        for batch, (inputs, targets) in enumerate(train_loader):
            x = np.ascontiguousarray(np.asarray(inputs), np.float32)
            y = np.asarray(targets)
            if x.ndim != 4 or y.ndim != 1 or x.shape[0] != y.shape[0]:
                raise ValueError(f"Expected inputs (N,H,W,C) and targets (N,), got {x.shape} and {y.shape}")          # train() asserts the same, :1104-1112
            if np.asarray(y).dtype.kind not in "iu":
                raise ValueError(f"targets must be integer class indices, got {np.asarray(y).dtype}")     # train() asserts integer labels, :1109-1110
            loss, correct = trainer.step(x, y.astype(np.int32))
            acc = correct / float(x.shape[0])
            running_loss += loss
            running_acc += acc
            n_batches += 1
            callback.on_batch_end(batch, {"loss": loss, "acc": acc})
        running_loss /= max(n_batches, 1)
        acc = running_acc / max(n_batches, 1)
        if len(val_loader) > 0:
            val_loss, correct, total, nb = 0.0, 0, 0, 0
            for inputs, targets in val_loader:
                x = np.ascontiguousarray(np.asarray(inputs), np.float32)
                y = np.asarray(targets)
                if y.dtype.kind not in "iu":
                    raise ValueError(f"targets must be integer class indices, got {y.dtype}")
                l, c = trainer.evaluate(x, y.astype(np.int32))
                val_loss += l
                correct += c
                total += x.shape[0]
                nb += 1
            val_loss /= nb
            val_acc = correct / float(total)
            best_val_acc = max(best_val_acc, val_acc)
            lr = scheduler.step(val_loss) if scheduler is not None else None
            if lr is not None:
                trainer.set_lr(lr)
            logs = {"val_loss": val_loss, "val_acc": val_acc, "val_precision": 0, "val_recall": 0}
            callback.on_epoch_end(epoch, logs)
            history.append({"epoch": epoch, "loss": running_loss, "acc": acc, **logs, "lr": lr})
        else:
            logs = {"loss": running_loss, "acc": acc}
            callback.on_epoch_end(epoch, logs)
            history.append({"epoch": epoch, **logs})
        if log is not None:
            log(f"Epoch {epoch}/{settings['epochs']} - " + " - ".join(f"{k}: {v}" for k, v in history[-1].items() if k != "epoch"))
        if getattr(callback, "stop_training", False):
            break
        if abort():
            break
    return history
