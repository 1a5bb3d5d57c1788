"""Sharding of a batch over GPUs: utterances are independent, so there is no data-path collective.

`shard_bounds` deals contiguous blocks of utterances to ranks with near-equal total SAMPLE counts
(closed-form lengths), which is what balances the kernels; `reduce_throughput` is the only
communication the multi-GPU path does (max of elapsed time, sum of samples) and works over any
torch.distributed backend (nccl = RCCL on the GPU box, gloo in the CPU tests).
"""
import numpy as np


def shard_bounds(sample_counts, world):
    """-> int array [world+1]: rank r owns utterances bounds[r]..bounds[r+1]-1."""
    counts = np.asarray(sample_counts, dtype=np.int64)
    n = len(counts)
    if world <= 1:
        return np.array([0, n], dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(counts)])
    targets = csum[-1] * np.arange(1, world) / float(world)
    cuts = np.searchsorted(csum, targets, side="left")
    cuts = np.clip(cuts, 0, n)
    bounds = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def reduce_throughput(elapsed_s, samples, dist=None, device=None):
    """(max elapsed over ranks, total samples over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(samples)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s.item())
