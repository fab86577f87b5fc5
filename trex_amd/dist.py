"""Frame sharding across the GPUs of one node + the one collective of the path.

Stage A..D of the hot path are independent per frame (given the background and the weights), so
frames are dealt to ranks in contiguous blocks, round-robin; background and network weights are
replicated.  The only exchange is the all-gather of fixed-size per-blob identity tables so that the
order-dependent consumer (TRex's Tracker::add on rank 0, tracking/Tracker.cpp:586-587) sees every frame.

torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests.
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


def all_gather_tables(table, group=None):
    """table: torch tensor [rows, rowlen] (int32/uint8 view ok), same shape on every rank.
    Returns [world*rows, rowlen] on every rank (one collective per batch of frames)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * table.shape[0],) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
    dist.all_gather_into_tensor(out, table.contiguous(), group=group)
    return out


def merge_tables(gathered):
    """Valid rows of a gathered table (numpy uint32 [n, HDR+C]) ordered by (global frame, row order
    within the frame) -- the order Tracker::add needs."""
    g = np.asarray(gathered)
    valid = g[g[:, 7] == 1]
    order = np.argsort(valid[:, 0], kind="stable")
    return valid[order]
