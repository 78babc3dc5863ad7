"""Multi-GPU layer of the hot path: shard (scene, view) render jobs one process per GPU, gather the rendered views
once at the end.

The reference's only parallelism is scene-level data parallelism (Lightning DDP, src/main.py:109) and every
(scene, view) render is independent (the reference literally loops over them, cuda_splatting.py:91-126), so the
path shards with NO data-path collective: Gaussians never leave their GPU.  The single exchange step is one fused
`all_gather` of the rendered images (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests).  When the
views of ONE scene are split across ranks in training, `reduce_gaussian_grads` sums the per-Gaussian gradients.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `num_items` jobs for `rank` (first ranks take the remainder)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(num_items, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_jobs(jobs: Sequence, rank: int, world_size: int) -> List:
    b, e = shard_range(len(jobs), rank, world_size)
    return list(jobs[b:e])


def gather_views(local: Tensor, num_total: int | None = None, group=None) -> Tensor:
    """All-gather rendered views: local (n_local, ...) on every rank -> (sum n_local, ...) on every rank, in rank order.

    One fused collective for the whole local batch (786 KB per 256x256 RGB view: latency-, not bandwidth-bound, so
    never gather per view).  Ragged shards (ranks holding different n_local) are padded to the largest shard."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    if dist.get_backend(group) == "gloo" and local.is_cuda:  # functional testing of the N > 1 path without RCCL
        return gather_views(local.cpu(), num_total, group).to(local.device)
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if all(c == n_max for c in counts):
        out = torch.empty((world * n_max, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((n_max, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * n_max, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    pieces = [out[r * n_max: r * n_max + c] for r, c in enumerate(counts)]
    res = torch.cat(pieces, dim=0)
    if num_total is not None and res.shape[0] != num_total:
        raise RuntimeError(f"gathered {res.shape[0]} views, expected {num_total}")
    return res


def reduce_gaussian_grads(grads: Sequence[Tensor], group=None) -> None:
    """Sum per-Gaussian gradients in place across ranks that rendered different views of the SAME scene
    (one bucketed all-reduce: ~45 MB at G = 131 072, K = 25; ring-bound at ~0.5 ms over xGMI)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off: off + n].view_as(g))
        off += n
