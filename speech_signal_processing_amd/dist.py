"""Multi-GPU layer: utterances shard embarrassingly (no cross-utterance term anywhere on the path —
GMM_UBM.py:183-185, d_vector.py:315-318), models / centroids are replicated, and the only collective is one
all-gather of the per-utterance results (RCCL over xGMI on the GPU box: torch.distributed backend "nccl";
"gloo" in the CPU tests).  One process per GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced-by-count range [lo, hi) of rank `rank` (first n % world ranks get one extra)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def balanced_shards(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges balanced by total length (frames) rather than by count — for ragged utterances.
    Greedy prefix cut at multiples of total/world; every rank gets a (possibly empty) contiguous range."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if world < 1:
        raise ValueError("world < 1")
    cum = np.concatenate([[0], np.cumsum(lengths)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))
        # pick the boundary closest to the target
        if i > 0 and abs(cum[i - 1] - target) <= abs(cum[min(i, n)] - target):
            i -= 1
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def all_gather_rows(local, group=None, force=False):
    """All-gather per-utterance result rows (torch tensor, first dim = this rank's utterances; ragged across ranks).
    Returns the concatenation in rank order on every rank.  One size exchange + one padded all_gather.
    force: run the collectives even in a world of one (tests: the RCCL path on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    home = local.device
    if dist.get_backend(group) == "gloo" and local.is_cuda:  # (CPU rehearsals of the multi-rank path: gloo moves host tensors)
        local = local.cpu()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0).to(home)


def pack_records(argmax, best, ubm):
    """SURVEY.md 8(e)'s compact per-utterance decision record — (int32 argmax, fp32 best, fp32 ubm), 12 bytes — as ONE int32 [n, 3]
    tensor (the two floats travel as their bit patterns): one collective moves all three columns."""
    import torch
    n = argmax.shape[0]
    rec = torch.empty((n, 3), dtype=torch.int32, device=argmax.device)
    rec[:, 0] = argmax.to(torch.int32)
    rec[:, 1] = best.to(torch.float32).contiguous().view(torch.int32)
    rec[:, 2] = ubm.to(torch.float32).contiguous().view(torch.int32)
    return rec


def unpack_records(rec):
    """-> (argmax int32 [n], best float32 [n], ubm float32 [n])"""
    import torch
    return rec[:, 0].contiguous(), rec[:, 1].contiguous().view(torch.float32), rec[:, 2].contiguous().view(torch.float32)


def decision_records(result):
    """Records of a GmmScorer.score result: argmax over the speaker models of (score - UBM score), that best difference (the
    reference's pred[j, argmax], GMM_UBM.py:185), and the UBM's mean log-likelihood."""
    import torch
    sc, am = result["scores"], result["argmax"].to(torch.int64)
    ubm = sc[:, 0]
    best = sc.gather(1, (am + 1)[:, None])[:, 0] - ubm
    return pack_records(am, best, ubm)


def all_gather_records(argmax, best, ubm, group=None):
    """All-gather of the decision records of this rank's utterances (ragged across ranks); every rank gets the three columns
    of ALL utterances in rank order."""
    return unpack_records(all_gather_rows(pack_records(argmax, best, ubm), group=group))


def max_over_ranks(value: float, device=None, force=False) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
