import numpy as np
import torch
from trex_amd import synth


def test_torch_generator_matches_numpy():
    fr, bg = synth.batch("C2", 2, t0=3)
    tf, tbg = synth.batch_torch("C2", 2, "cpu", t0=3)
    assert np.array_equal(tf.numpy(), fr) and np.array_equal(tbg.numpy(), bg)


def test_recipe_properties():
    fr, bg = synth.batch("C2", 1)
    d = np.abs(fr[0].astype(int) - bg.astype(int))
    assert d[d <= 15].max() <= 3            # noise stays below detect_threshold
    assert 32 * 200 < (d > 15).sum() < 32 * 320   # ~283 px per individual
