"""Times trexhip_train_step_device (one optimizer step of V118_3, fp32) on the GPU and the CPU restatement beside it.
  python tools/time_train.py [--n 128] [--classes 100] [--steps 20] [--cpu-steps 2]"""
import argparse
import json
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trex_amd import capi, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--stream", default="torch", choices=["torch", "own", "side"], help="the context's stream: torch's current one (the legacy null stream), the library's own, a torch side stream")
    a = ap.parse_args()
    state = weights.synthetic_state(a.classes, 1, channels=a.channels)
    x, y = weights.synthetic_train_batch(a.n, 2, a.classes, a.channels)
    p = capi.default_params(64, 64)
    p.max_batch = 1
    side = torch.cuda.Stream() if a.stream == "side" else None
    seg = capi.Segmenter(p, stream={"torch": "torch", "own": None, "side": side.cuda_stream if side else None}[a.stream])
    tr = capi.Trainer(seg, weights.pack_blob(state, a.classes, a.channels), max_batch=a.n, lr=1e-3, seed=3)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()
    for _ in range(3):
        tr.step_device(dx.data_ptr(), dy.data_ptr(), a.n, 0, want_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step_device(dx.data_ptr(), dy.data_ptr(), a.n, 0, want_loss=False)
    t_enq = (time.perf_counter() - t0) / a.steps          # host time to queue a step (launches + events)
    seg.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"n": a.n, "classes": a.classes, "channels": a.channels, "gpu_ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": t_enq * 1e3, "gpu_samples_per_s": a.n / dt,
           # forward 3 convs + fc (MAC/sample) x 3 (forward, data gradient, weight gradient; conv1 has no data gradient)
           "algorithmic_gflop_per_step": 2 * a.n * (2.56e6 * a.channels * 2 + 40.96e6 * 3 + 81.92e6 * 3 + 1.28e6 * 3) / 1e9}
    out["gpu_tflops"] = out["algorithmic_gflop_per_step"] / dt / 1e3
    if a.cpu_steps > 0:
        from oracle import cnn_train_oracle as tro
        rng = np.random.default_rng(0)
        masks = {"d1": rng.random((a.n, 16)) >= 0.05, "d2": rng.random((a.n, 64)) >= 0.05, "d3": rng.random((a.n, 128)) >= 0.05, "d4": rng.random((a.n, 100)) >= 0.05}
        cores = len(os.sched_getaffinity(0))
        adam = tro.new_adam_state(state)
        st = state
        st, *_ = tro.train_step(st, adam, x, y, masks, 1e-3, threads=cores)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            st, *_ = tro.train_step(st, adam, x, y, masks, 1e-3, threads=cores)
        dc = (time.perf_counter() - t0) / a.cpu_steps
        out.update({"cpu_ms_per_step": dc * 1e3, "cpu_samples_per_s": a.n / dc, "cpu_cores": cores, "cpu_kind": "port (torch fp32 functional restatement, oracle/cnn_train_oracle.py)"})
    print(json.dumps(out))
    tr.close()
    seg.close()


if __name__ == "__main__":
    main()
