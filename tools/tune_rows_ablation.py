"""Ablations of k_rows (dev tool): order bit 256 = skip run extraction, bit 512 = trivial mask."""
import os, sys, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
for order in (0, 256, 512, 768):
    for blocks in (4096, 8192, 16384):
        env = dict(os.environ, TREXHIP_ROWS_ORDER=str(order), TREXHIP_ROWS_BLOCKS=str(blocks))
        out = subprocess.run([sys.executable, os.path.join(HERE, "tune_rows.py"), "child"], env=env, capture_output=True, text=True)
        print(order, blocks, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
