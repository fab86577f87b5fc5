"""The CPU restatement of the identity network's training step (oracle/cnn_train_oracle.py) against what the reference's own
module + torch.optim.Adam produced (tests/golden/cnn_train_v118_3.npz, generator tests/golden/make_train_fixture.py)."""
import os
import numpy as np
import pytest

from oracle import cnn_train_oracle as tro
from trex_amd import weights

FIX = os.path.join(os.path.dirname(__file__), "golden", "cnn_train_v118_3.npz")
SAMPLE = {"conv3.weight": 7, "fc1.weight": 53}


def sample(name, arr):
    return np.asarray(arr).reshape(-1)[::SAMPLE.get(name, 1)]


def load_case(fx, name):
    classes, ch, n, steps, seed = [int(v) for v in fx[f"{name}/meta"]]
    lr = float(fx[f"{name}/lr"][0])
    masks = [{t: fx[f"{name}/mask{s}/{t}"] for t in ("d1", "d2", "d3", "d4")} for s in range(steps)]
    return classes, ch, n, steps, seed, lr, masks


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_reproduces_the_reference_training_steps(name):
    fx = np.load(FIX)
    classes, ch, n, steps, seed, lr, masks = load_case(fx, name)
    state = weights.synthetic_state(classes, seed, channels=ch)
    adam = tro.new_adam_state(state)
    for s in range(steps):
        x, y = weights.synthetic_train_batch(n, seed + 100 * s, classes, ch)
        state, loss, correct, grads = tro.train_step(state, adam, x, y, masks[s], lr, threads=8)
        assert abs(loss - float(fx[f"{name}/loss{s}"][0])) <= 2e-5 * max(1.0, abs(loss)), (s, loss)
        assert correct == int(fx[f"{name}/correct{s}"][0])
        if s == 0:
            for k in tro.TRAINABLE:
                ref = fx[f"{name}/grad0/{k}"]
                got = sample(k, grads[k])
                tol = 2e-5 * float(np.abs(ref).max()) + 1e-9
                assert np.abs(got - ref).max() <= tol, (k, float(np.abs(got - ref).max()), tol)
    # after the last step.  A conv bias feeds a BatchNorm: its true gradient is 0, what Adam sees is rounding noise and the
    # update is +-lr of noise sign in the reference itself -- bounded, not reproducible
    for k in tro.TRAINABLE + tro.BUFFERS:
        ref = fx[f"{name}/final/{k}"]
        got = sample(k, state[k])
        if k in ("conv1.bias", "conv2.bias", "conv3.bias"):
            assert np.abs(got - ref).max() <= 2.0 * steps * lr + 1e-7, k
            continue
        err = np.abs(got - ref)
        assert err.max() <= 2.0 * steps * lr + 1e-6 * np.abs(ref).max(), (k, float(err.max()))
        assert np.mean(err <= 0.02 * lr + 1e-6 * np.abs(ref)) >= 0.995, (k, float(np.mean(err <= 0.02 * lr + 1e-6 * np.abs(ref))))
