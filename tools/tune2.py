import os, sys, subprocess
for order in (0, 256, 512, 768, 1, 257):
    env = dict(os.environ, TREXHIP_ROWS_ORDER=str(order), TREXHIP_ROWS_BLOCKS="4096")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "tune_rows.py"), "child"], env=env, capture_output=True, text=True)
    print(order, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
