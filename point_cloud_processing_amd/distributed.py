"""Query sharding across the GPUs of one node (one process per GPU, torch.distributed).

Layout: the cloud and its spatial index are replicated on every rank (10 M points are 160 MB
of records, 100 M are 1.6 GB: small against 288 GB of HBM); rank r answers the queries at
sorted positions [r*n/W, (r+1)*n/W) of the index, a spatially coherent, equally sized slice.
kNN rows, normals and radius counts need no exchange at all.  The statistical outlier filter
has one real exchange step: its global mean and standard deviation.  Each rank contributes
(sum, count) and then the squared-deviation sum of its slice through a tiny all-gather
(RCCL over xGMI when the backend is "nccl"), every rank adds the partials in rank order (so
all ranks hold bit-identical statistics), and the kept indices are exchanged with one padded
all-gather.  Nothing else crosses GPUs.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open slice of ``n`` units owned by ``rank``: sizes differ by at most one."""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def _device(group=None) -> torch.device:
    backend = dist.get_backend(group)
    if backend == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_gather_f64(values, group=None) -> np.ndarray:
    """All-gather a few float64 scalars per rank -> array [world, len(values)] (rank order)."""
    world = dist.get_world_size(group)
    t = torch.tensor(list(values), dtype=torch.float64, device=_device(group))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy()


def all_gather_varlen(arr: np.ndarray, group=None) -> np.ndarray:
    """Concatenate 1-D int64 arrays of different lengths from all ranks, in rank order."""
    world = dist.get_world_size(group)
    dev = _device(group)
    n = torch.tensor([len(arr)], dtype=torch.int64, device=dev)
    sizes = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes) if sizes else 0
    buf = torch.zeros(max(m, 1), dtype=torch.int64, device=dev)
    if len(arr):
        buf[: len(arr)] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(out, sizes)]) if sizes else np.zeros(0, np.int64)


def sor_sharded(engine, pts: np.ndarray, k: int, std_ratio: float, cache: Optional[dict] = None, group=None):
    """Statistical outlier removal with the queries sharded over the ranks of ``group``.

    Returns ``(kept_indices, (mean, std, threshold, valid))``; identical on every rank and
    equal to the single-GPU result (reference semantics: filtering_adapter.py:91).
    """
    cache = {} if cache is None else cache
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    owned, avg = engine.sor_avg_shard(pts, k, cache, rank, world)  # original indices, their averages
    pos = avg > 0
    part = all_gather_f64([float(avg[pos].sum()) if pos.any() else 0.0, float((avg >= 0).sum())], group)
    total, valid = 0.0, 0
    for r in range(world):  # fixed order -> the same bits on every rank
        total += part[r, 0]
        valid += int(part[r, 1])
    if valid == 0:
        return np.zeros(0, dtype=np.int64), (0.0, 0.0, 0.0, 0)
    mean = total / valid
    dev = avg[pos] - mean
    part = all_gather_f64([float((dev * dev).sum()) if pos.any() else 0.0], group)
    sq = 0.0
    for r in range(world):
        sq += part[r, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        std = float(np.sqrt(np.float64(sq) / np.float64(valid - 1)))
    thr = mean + std_ratio * std
    kept_local = owned[pos & (avg < thr)]
    kept = np.sort(all_gather_varlen(kept_local, group))
    return kept, (mean, std, thr, valid)


def normals_sharded(engine, pts: np.ndarray, k: int, radius: float, cache: Optional[dict] = None, group=None):
    """PCA normals with the queries sharded; the (n, 3) result is all-gathered to every rank."""
    cache = {} if cache is None else cache
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    owned, nrm = engine.normals_shard(pts, k, radius, cache, rank, world)
    idx = all_gather_varlen(owned, group)
    flat = all_gather_varlen(nrm.reshape(-1).view(np.int64), group).view(np.float64).reshape(-1, 3)
    out = np.empty((len(pts), 3), dtype=np.float64)
    out[idx] = flat
    return out


def knn_sharded(engine, pts: np.ndarray, k: int, radius: float = -1.0, cache: Optional[dict] = None, group=None):
    """kNN rows of this rank's slice only: ``(owned_indices, idx, d2, cnt)``.  Nothing is
    exchanged: full neighbour lists stay on the GPU that computed them."""
    cache = {} if cache is None else cache
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    return engine.knn_shard(pts, k, radius, cache, rank, world)
