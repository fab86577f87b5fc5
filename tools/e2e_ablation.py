"""Which of the recalled detect-stage choices can the reference's golden data discriminate?  Re-scores the CPU oracle on all 200
shipped test frames (fixture tests/golden/e2e_testframes.npz) with one choice flipped at a time.  Prints a markdown table
(committed in DESIGN.md section 2)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle
from e2e_golden import Golden, run_variant

G = Golden()
variants = [("baseline: detect `>` 9, 8-connectivity, track `>=` 12", {}),
            ("detect strict `>` 9 (the default is the documented `>=`)", dict(inclusive=0)),
            ("4-connectivity", dict(connectivity=4)),
            ("detect_threshold 8", dict(detect_threshold=8)),
            ("detect_threshold 10", dict(detect_threshold=10)),
            ("detect_threshold 11", dict(detect_threshold=11)),
            ("track_threshold 11", dict(track_threshold=11)),
            ("track_threshold 13", dict(track_threshold=13))]
print("| variant | golden rows | blob id reproduced | + identical num_pixels | median abs(dnum_pixels) |")
print("|---|---|---|---|---|")
for name, kw in variants:
    tot, hits, exact, deltas = run_variant(G, oracle, **kw)
    print(f"| {name} | {tot} | {hits} ({100.0 * hits / tot:.1f} %) | {exact} ({100.0 * exact / tot:.1f} %) | {np.median(deltas) if deltas else float('nan'):.0f} |")
