"""N>1 path on CPU: 2 gloo ranks shard the frames, build identity tables, all-gather them, and the
merged table equals the single-process one.  The tables come from the oracle here
(no GPU in this container); tests/test_dist_gpu.py checks the device kernel against the same builder."""
import os
import sys
import numpy as np
import pytest
from oracle import tables as otables
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trex_amd import dist as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_covers_every_frame_once():
    for n, world, block in [(200, 8, 8), (64, 2, 32), (7, 4, 2), (1, 8, 64), (130, 3, 64)]:
        plan = tdist.shard_plan(n, world, block)
        seen = []
        for r in plan:
            for first, cnt in r:
                assert cnt > 0 and first % block == 0
                seen.extend(range(first, first + cnt))
        assert sorted(seen) == list(range(n))
        sizes = [sum(c for _, c in r) for r in plan]
        assert max(sizes) - min(sizes) <= block


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_strong_scaling_split_covers_a_256_frame_block_exactly_once(world):
    # bench.py --scaling strong --gpus N: one camera's 256-frame block per step, split between the ranks (Pipeline: frame_base of the table rows)
    per = tdist.strong_split(256, world)
    for step in range(3):
        seen = []
        for rank in range(world):
            first, n = tdist.step_frames(step, rank, world, per)
            assert n == per
            seen.extend(range(first, first + n))
        assert seen == list(range(step * 256, (step + 1) * 256))        # in rank order, no gap, no overlap: what Tracker::add needs
    with pytest.raises(ValueError):
        tdist.strong_split(256, 3)


def _frames():
    from trex_amd import synth
    rng = np.random.default_rng(1)
    frames, bgs = [], None
    for i in range(6):
        fr, bg = synth.random_scene(rng, 160, 96, density=0.03)
        frames.append(fr); bgs = bg
    return np.stack(frames), bgs


def _table_for(frames, bg, first, classes, max_rows):
    from oracle import oracle
    p = oracle.make_params(frames.shape[2], frames.shape[1])
    res = []
    for f in frames:
        b, r, px = oracle.segment(f, bg, p)
        res.append(b)
    n = sum(len(b) for b in res)
    rng = np.random.default_rng(100 + first)
    probs = rng.random((n, classes)).astype(np.float32)
    return otables.table_from_blobs(res, first, probs, classes, max_rows)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, bg = _frames()
    plan = tdist.shard_plan(len(frames), world, 3)
    (first, cnt), = plan[rank]
    t = _table_for(frames[first:first + cnt], bg, first, 5, 512)
    gathered = tdist.gather_tables_torch(torch.from_numpy(t.view(np.int32)))       # gloo stand-in of trexhip_comm_gather_device: rank 0 only
    if rank == 0:
        q.put(tdist.merge_tables(gathered.numpy().view(np.uint32)))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    frames, bg = _frames()
    a = _table_for(frames[0:3], bg, 0, 5, 512)
    b = _table_for(frames[3:6], bg, 3, 5, 512)
    want = tdist.merge_tables(np.concatenate([a, b]))
    assert merged.shape == want.shape and np.array_equal(merged, want)
    assert np.all(np.diff(merged[:, 0].astype(np.int64)) >= 0)
