"""Sweep k_rows launch knobs on the GPU (dev tool, not part of the product)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from trex_amd import capi, synth
    cfg = os.environ.get("CFG", "C4"); B = int(os.environ.get("B", "64"))
    W, H, _, _ = synth.CONFIGS[cfg]
    frames, bg = synth.batch_torch(cfg, B, "cuda")
    seg = capi.Segmenter(capi.default_params(W, H, max_batch=B, max_blobs=1024, max_pixels=1 << 18, max_runs=32768))
    seg.set_background(bg)
    for _ in range(3):
        seg.segment_device(frames.data_ptr(), B); seg.synchronize()
    seg.profile_enable(True); seg.profile_reset()
    for _ in range(20):
        seg.segment_device(frames.data_ptr(), B)
    seg.synchronize()
    r, n = seg.profile_read(capi.STAGE_ROWS); a, _ = seg.profile_read(capi.STAGE_SEGMENT_ALL)
    print(json.dumps({"rows_us": r / n * 1e3, "all_us": a / n * 1e3, "GBs": 2.0 * W * H * B / (r / n * 1e-3) / 1e9}))
else:
    for order in (0, 1):
        for blocks in (512, 1024, 2048, 4096, 8192, 1000000):
            env = dict(os.environ, TREXHIP_ROWS_ORDER=str(order), TREXHIP_ROWS_BLOCKS=str(blocks))
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(order, blocks, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
