"""Sweep the frame-group count of the detect pass (dev tool)."""
import os, sys, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
for g in (1, 2, 4, 8):
    for blocks in (8192,):
        env = dict(os.environ, TREXHIP_SEG_GROUPS=str(g), TREXHIP_ROWS_BLOCKS=str(blocks))
        out = subprocess.run([sys.executable, os.path.join(HERE, "tune_rows.py"), "child"], env=env, capture_output=True, text=True)
        print(g, blocks, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
