"""Randomised parity sweep of the training step against the CPU restatement (dev tool): random batch sizes, class counts, channel counts,
dropout masks, learning rates; two consecutive steps per case (the second sees updated weights, moments and running statistics).
   gpurun -- 'PYTHONPATH=.:tests python tools/fuzz_train.py 40 [seed]'"""
import sys
import numpy as np
from oracle import cnn_train_oracle as tro
from trex_amd import capi, weights
from test_train_gpu import make_seg, step, read_all, GRAD_RTOL, CONV_BIAS

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
ties_dev = ties_cpu = 0
seg = make_seg()
for case in range(n_cases):
    n = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 129, 257]))
    classes = int(rng.choice([1, 2, 7, 10, 100, 300]))
    ch = int(rng.choice([1, 3]))
    lr = float(rng.choice([1e-4, 1e-3, 1e-2]))
    seed = int(rng.integers(1, 1 << 30))
    state = dict(weights.synthetic_state(classes, seed, channels=ch))
    if rng.random() < 0.3:      # convolution weights far from their usual range: the operand scales of the fp16 two-piece arithmetic
        for k in ("conv2.weight", "conv3.weight"):
            state[k] = (state[k] * np.float32(10.0 ** rng.uniform(-3, 3))).astype(np.float32)
    adam = tro.new_adam_state(state)
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, ch), max_batch=n, lr=lr, precision=case % 2)   # both arithmetics of the conv2 / conv3 convolutions
    try:
        for s in range(2):
            x, y = weights.synthetic_train_batch(n, seed + s, classes, ch)
            pd = float(rng.choice([0.0, 0.05, 0.3]))
            masks = {"d1": rng.random((n, 16)) >= pd, "d2": rng.random((n, 64)) >= pd, "d3": rng.random((n, 128)) >= pd, "d4": rng.random((n, 100)) >= pd}
            state_before = state
            state, loss_ref, correct_ref, grads = tro.train_step(state, adam, x, y, masks, lr, threads=16)
            loss, correct = step(tr, x, y, masks)
            assert abs(loss - loss_ref) <= (1e-4 if s == 0 else 2e-2) * max(1.0, abs(loss_ref)), ("loss", loss, loss_ref)   # step 2 starts from weights that differ by Adam noise
            g = read_all(tr, classes, ch, 1)
            if s == 0:          # the second step starts from weights that differ by the Adam noise (see tests/test_train_gpu.py): compare step 1 tightly
                assert correct == correct_ref, ("correct", correct, correct_ref)
                for k in tro.TRAINABLE:
                    if k in CONV_BIAS:
                        continue
                    tol = GRAD_RTOL * float(np.abs(grads[k]).max()) + 1e-8
                    assert np.abs(g[k] - grads[k]).max() <= tol, (k, float(np.abs(g[k] - grads[k]).max()), tol)
    except AssertionError as e:
        # which side is off?  float64 autograd of the same step is the reference point: an arg-max / ReLU near-tie decided differently by two
        # fp32 summation orders moves both fp32 results away from it by a comparable amount; a bug moves only the device result
        import torch
        k = e.args[0][0] if e.args and isinstance(e.args[0], tuple) else None
        verdict = ""
        if s == 0 and k in tro.TRAINABLE:
            _, _, g64, _, _ = tro.forward_backward(state_before, x, y, masks, 16, dtype=torch.float64)
            dev_err = float(np.abs(g[k] - g64[k]).max()); cpu_err = float(np.abs(grads[k] - g64[k]).max())
            verdict = " | vs float64: device %.3g, cpu-fp32 %.3g (max |g| %.3g)" % (dev_err, cpu_err, float(np.abs(g64[k]).max()))
            # a near-tie (max-pool arg-max, ReLU sign) decided differently by two fp32 summation orders re-routes one activation's gradient:
            # one of the two fp32 results then sits within rounding of float64 and the other a bounded distance away.  Which one is a coin
            # flip -- the tally at the end shows both sides take turns.  A defect would put the device far from float64 while the fp32
            # restatement is close in (nearly) every such case, or move it by more than a re-routed activation can.
            gmax = float(np.abs(g64[k]).max())
            if max(dev_err, cpu_err) <= 5e-2 * gmax:
                if dev_err < cpu_err: ties_cpu += 1
                else: ties_dev += 1
                print("near-tie case", case, dict(n=n, classes=classes, ch=ch), k, verdict, flush=True)
                tr.close()
                continue
        fails += 1
        print("FAIL case", case, dict(n=n, classes=classes, ch=ch, lr=lr, seed=seed), str(e)[:300], verdict, flush=True)
    tr.close()
print("cases", n_cases, "failures", fails, "| fp32 near-ties decided differently: device further from float64 in", ties_dev, "cases, the fp32 restatement in", ties_cpu)
