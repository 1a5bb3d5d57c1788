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


def shard_deal(sample_counts, world, deal="contiguous", block=64):
    """-> list of `world` index arrays: the utterances of every rank, in the rank's own order.
    "contiguous": shard_bounds' ranges (near-equal total sample count; what bench.py's ranks build, each from its own range of the
    recipe).  "sorted": SURVEY 8(e) -- utterances sorted by length (longest first, ties in batch order), blocks of `block` (one
    wavefront) dealt round-robin -- every rank sees the same length distribution, so a batch whose long utterances cluster does not
    leave one GPU with the long tail; the engine's node object does the same under its option "deal" (speechPlayer_node_setOption)."""
    counts = np.asarray(sample_counts, dtype=np.int64)
    if deal == "contiguous":
        b = shard_bounds(counts, world)
        return [np.arange(b[r], b[r + 1], dtype=np.int64) for r in range(world)]
    if deal != "sorted":
        raise ValueError(deal)
    order = np.argsort(-counts, kind="stable")
    rank_of = (np.arange(len(order)) // block) % max(world, 1)
    return [order[rank_of == r] for r in range(world)]


def longest_wave(sample_counts, members, lanes=64):
    """Samples of the longest wavefront a rank launches for the utterances `members`: the engine packs lanes by length (longest
    first), a wavefront lasts as long as its longest lane, and a launch at least as long as its longest wavefront."""
    c = np.asarray(sample_counts, dtype=np.int64)[np.asarray(members, dtype=np.int64)]
    return int(c.max()) if len(c) else 0


def wave_time(sample_counts, members, lanes=64):
    """Sum over a rank's wavefronts (lanes packed longest first) of the wavefront's longest lane: what the rank's launch costs when
    it has more wavefronts than the device runs at once."""
    c = np.sort(np.asarray(sample_counts, dtype=np.int64)[np.asarray(members, dtype=np.int64)])[::-1]
    return int(c[::lanes].sum())


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
