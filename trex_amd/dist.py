"""Frame sharding across the GPUs of one node + the one collective of the path.

Stage A..D of the hot path are independent per frame (given the background and the weights), so
frames are dealt to ranks in contiguous blocks, round-robin; background and network weights are
replicated.  The only exchange is the all-gather of fixed-size per-blob identity tables so that the
order-dependent consumer (TRex's Tracker::add on rank 0, tracking/Tracker.cpp:586-587) sees every frame.

The exchange itself belongs to the library: trexhip_comm_* in include/trexhip.h (trex_amd/csrc/comm.hip: grouped ncclSend / ncclRecv
to rank 0 over RCCL, capi.Comm in Python) -- that is what trex_amd/pipeline.py and bench.py use.  This module holds the host-side
logic around it (who gets which frames, how rank 0 orders the rows) and, for the CPU tests (gloo, no GPUs), a torch.distributed
stand-in of the same gather.
"""
import numpy as np

HDR = 8  # header words per table row (see include/trexhip.h: trexhip_export_id_table_device)
HDR_EX = 16  # header words of the full record (trexhip_export_id_table_ex_device): + mu20, mu11, mu02, midline len / angle / offset, status


def shard_plan(n_frames, world, block):
    """Blocks of `block` consecutive frames dealt round-robin: rank r gets blocks r, r+world, ...
    Returns, per rank, a list of (first_frame, n) ranges.  Every frame appears exactly once."""
    plan = [[] for _ in range(world)]
    b = 0
    for first in range(0, n_frames, block):
        plan[b % world].append((first, min(block, n_frames - first)))
        b += 1
    return plan


def step_frames(step, rank, world, frames_per_rank):
    """The frames rank `rank` owns in step `step` of the frame-sharded pipeline (trex_amd/pipeline.py, bench.py): every step is one block
    of world * frames_per_rank consecutive frames, rank r takes the r-th slice of it (weak scaling: frames_per_rank = the batch; strong
    scaling: the batch divided by the ranks).  Returns (first_frame, n)."""
    return (step * world + rank) * frames_per_rank, frames_per_rank


def strong_split(block_frames, world):
    """--scaling strong: one camera's block of `block_frames` frames split between the ranks; the block has to divide."""
    if block_frames % world:
        raise ValueError("--scaling strong: the batch must divide by the number of GPUs")
    return block_frames // world


def gather_tables_torch(table, dst=0, group=None):
    """CPU-test stand-in of trexhip_comm_gather_device: table [rows, rowlen] from every rank -> [world*rows, rowlen] on rank `dst`
    (None elsewhere), through torch.distributed (gloo)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [torch.empty_like(table) for _ in range(world)] if rank == dst else None
    dist.gather(table.contiguous(), parts, dst=dst, group=group)
    return torch.cat(parts, 0) if rank == dst else None


def merge_tables(gathered):
    """Valid rows of a gathered table (numpy uint32 [n, HDR+C]) ordered by (global frame, row order
    within the frame) -- the order Tracker::add needs."""
    g = np.asarray(gathered)
    valid = g[g[:, 7] == 1]
    order = np.argsort(valid[:, 0], kind="stable")
    return valid[order]
