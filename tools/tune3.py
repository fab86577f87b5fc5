import os, sys, subprocess
for stop in (1, 2, 3, 4, 5, 6, 7, 8, 0):
    env = dict(os.environ, TREXHIP_CCL_STOP=str(stop))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "tune_rows.py"), "child"], env=env, capture_output=True, text=True)
    print(stop, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
