"""Multi-GPU helpers: spectrograms are independent units, so a batch is split into
contiguous blocks, one per rank (one process per GPU), with NO collective in the update loop.  The only
exchange is the optional final reduction of the consistency-residual pair (sum|acc+w00 S|^2, sum|S|^2):
2 doubles per job (or 2 per spectrogram), all-reduced over RCCL on GPUs (gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of `n_items` spectrograms owned by `rank`; blocks differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_residual(local_pairs, group=None, device=None):
    """All-reduce (sum) the per-rank residual pairs.  `local_pairs`: array (..., 2) of
    [sum|residual|^2, sum|S|^2] for the spectrograms this rank owns.  Returns (err, pow, dB) of the job."""
    import torch
    import torch.distributed as dist

    tot = np.asarray(local_pairs, dtype=np.float64).reshape(-1, 2).sum(axis=0)
    t = torch.tensor(tot, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    err, pw = float(t[0].item()), float(t[1].item())
    return err, pw, (10.0 * np.log10(pw / err) if err > 0 else float("inf"))


def gather_shards(local, n_items, group=None):
    """All-gather per-rank result blocks (numpy, leading axis = this rank's spectrograms) into the full
    batch on every rank.  Convenience for evaluation; not used in the timed path."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, np.asarray(local), group=group)
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == n_items
    return out
