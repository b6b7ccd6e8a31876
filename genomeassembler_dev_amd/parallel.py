"""Multi-GPU layout of the path: one process per GPU, segments sharded over ranks, NO data-path collective.

Segments are independent graphs (the reference loops `for(i in 1:total_iters)` over them,
scripts/02_Real_vs_rand_prob_own.R:33-53), so every rank builds and scores its own contiguous block of segments and
only the small per-segment results (contigs, scores) are gathered at the end (mode 1).  The pooled
hash-bucket all-to-all of mode 2 only pays for a single graph larger than one GPU and is not built.

`torch.distributed` is plumbing here: process group, barrier, gather.  Backend "nccl" is RCCL on ROCm; the CPU tests run
the same functions over "gloo"."""
import numpy as np


def shard_bounds(n_segments, world_size):
    """[start, end) segment indices of every rank: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(n_segments, world_size)
    bounds, s = [], 0
    for r in range(world_size):
        e = s + base + (1 if r < extra else 0)
        bounds.append((s, e))
        s = e
    return bounds


def shard_reads(reads, seg_read_off, rank, world_size):
    """The rank's block of a batch given as (reads [n_reads, read_len] or flat uint8 + offsets, seg_read_off).
    Returns (reads_of_rank, seg_read_off_of_rank (rebased to 0), (seg_start, seg_end))."""
    seg_read_off = np.asarray(seg_read_off, dtype=np.uint64)
    s, e = shard_bounds(len(seg_read_off) - 1, world_size)[rank]
    a, b = int(seg_read_off[s]), int(seg_read_off[e])
    return reads[a:b], (seg_read_off[s:e + 1] - seg_read_off[s]).astype(np.uint64), (s, e)


def gather_segment_results(local, group=None, dst=0):
    """Gather per-segment result lists (one entry per local segment, any picklable payload) to rank `dst` in global
    segment order.  Returns the concatenated list on `dst`, None elsewhere.  Without an initialised process group the
    local list is returned (single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(local)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [None] * world if rank == dst else None
    dist.gather_object(list(local), out, dst=dst, group=group)
    if rank != dst:
        return None
    return [x for part in out for x in part]


def max_over_ranks(value, device=None, group=None):
    """max of a python float over the ranks (the bench's step time)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
