"""Pins the CNN oracle (own restatement) on vectors produced by importing the reference's network
(tests/golden/cnn_v118_3_*.npz; generator: tests/golden/make_cnn_fixtures.py).  Tolerance 1e-4 abs on
softmax (BASELINE.json north_star), 2e-3 abs on logits (|logit| up to ~15)."""
import os
import numpy as np
import pytest
from oracle import cnn_oracle
from trex_amd import weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_fixture(classes):
    z = np.load(os.path.join(GOLD, f"cnn_v118_3_c{classes}.npz"))
    st = weights.synthetic_state(int(z["classes"]), int(z["seed"]))
    for k in z.files:
        if k.startswith("stat/"):
            st[k[5:]] = z[k]
    return z, st


@pytest.mark.parametrize("classes", [8, 100, 256])
def test_oracle_matches_reference_vectors(classes):
    z, st = load_fixture(classes)
    sizes = sorted(int(k.split("/")[1]) for k in z.files if k.startswith("probs/"))
    for n in sizes[:3]:
        crops = weights.synthetic_crops(n, int(z["seed"]) + 1000 + n)
        probs, logits = cnn_oracle.predict(st, crops, threads=8)
        assert np.abs(probs - z[f"probs/{n}"]).max() <= 1e-4
        assert np.abs(logits - z[f"logits/{n}"]).max() <= 2e-3
        assert np.allclose(probs.sum(1), 1.0, atol=1e-5)


def test_batch_rule_and_transform_results():
    assert [cnn_oracle.batch_size_rule(n) for n in (1, 8, 64, 65, 100, 128, 1024)] == [64, 64, 64, 128, 128, 128, 128]
    vals = np.arange(6, dtype=np.float32).reshape(2, 3)
    flat = cnn_oracle.transform_results(4, [2, 0], vals)
    assert flat.tolist() == [3, 4, 5, -1, -1, -1, 0, 1, 2, -1, -1, -1]


def test_blob_roundtrip():
    st = weights.synthetic_state(8, 1)
    blob = weights.pack_blob(st, 8)
    assert len(blob) == 32 + 4 * sum(int(np.prod(s)) for _, s in weights.shapes(8))
