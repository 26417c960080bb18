"""Query sharding across the GPUs of one node (one process per GPU, torch.distributed).

Layout: the cloud and its spatial index are replicated on every rank (10 M points are 160 MB
of records, 100 M are 1.6 GB: small against 288 GB of HBM); rank r answers the queries of a
contiguous slice of the index's sorted positions -- spatially coherent, equally sized, cut at
multiples of the reduction chunk -- into outputs of ITS OWN size (rows in slice order plus the
original index of every row).  kNN rows, normals and radius counts need no exchange at all.
The statistical outlier filter has one real exchange step, its global mean and standard
deviation: every rank adds up its own rows as exact integer sums (``pc_sor_exact_sums``: ten 64-bit
words), the words are added over the ranks (RCCL all-reduce over xGMI when the backend is "nccl")
and every rank derives the same mean, standard deviation and threshold from the same integers --
the same bits for 1, 2, 4 or 8 ranks, for slices of a replicated index and for z slabs alike.  The
outcome is exchanged the same way (one bit per point, set for the points a rank removes: the ranks'
bits are disjoint, so the sum is the union).  Nothing else crosses GPUs, and nothing visits the
host but a few scalars.

Each algorithm is written once, as a generator that yields the arrays it wants summed over the
ranks: ``run`` drives it with ``dist.all_reduce``; ``run_emulated`` drives the generators of all
ranks of a pretended world inside one process (tests on one GPU).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

SOR_CHUNK = 4096  # sorted positions per partial sum (PC_SOR_CHUNK of the C ABI)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open slice of ``n`` units owned by ``rank``: sizes differ by at most one."""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def shard_positions(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Slice of the ``n`` sorted positions owned by ``rank``: whole reduction chunks, dealt evenly."""
    chunks = (n + SOR_CHUNK - 1) // SOR_CHUNK
    cb, ce = shard_range(chunks, rank, world)
    return min(cb * SOR_CHUNK, n), min(ce * SOR_CHUNK, n)


# ---- drivers -------------------------------------------------------------------------------------
def _as_tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)  # (shares the array's memory)


def run(gen, group=None):
    """Drive one rank's generator: every yielded array is summed over the ranks of ``group``, in place
    (a process without a process group is a world of one: nothing to add)."""
    alone = not (dist.is_available() and dist.is_initialized())
    # (gloo with device tensors -- rehearsals of several ranks on one GPU -- stages through the host on streams of its
    # own: the device is brought to rest around every exchange; seen to hang otherwise on the second job of a process)
    fence = not alone and dist.get_backend(group) == "gloo"
    try:
        x = next(gen)
        while True:
            if not alone:
                t = _as_tensor(x)
                if fence and t.is_cuda:
                    torch.cuda.synchronize(t.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                if fence and t.is_cuda:
                    torch.cuda.synchronize(t.device)
            x = gen.send(x)
    except StopIteration as stop:
        return stop.value


def run_emulated(make_gen, world: int):
    """The same algorithm for ``world`` pretended ranks in this process: ``make_gen(rank)`` per rank, the
    yielded arrays added in rank order.  Returns the list of the ranks' results."""
    gens = [make_gen(r) for r in range(world)]
    results = [None] * world
    xs = []
    for r, g in enumerate(gens):
        try:
            xs.append(next(g))
        except StopIteration as stop:
            results[r] = stop.value
            xs.append(None)
    while any(x is not None for x in xs):
        ts = [_as_tensor(x) for x in xs]
        total = ts[0].clone()
        for t in ts[1:]:
            total += t
        nxt = []
        for r, g in enumerate(gens):
            ts[r].copy_(total)
            try:
                nxt.append(g.send(xs[r]))
            except StopIteration as stop:
                results[r] = stop.value
                nxt.append(None)
        xs = nxt
    return results


# ---- the algorithms ------------------------------------------------------------------------------
EXACT_WORDS = 10  # PC_SOR_EXACT_WORDS of the C ABI


def exact_value(words) -> tuple:
    """(sum as an exact integer multiple of 2^-128, count, terms too large for the format) of the words of
    ``pc_sor_exact_sums``: digit j weighs 2^(32 j - 128)."""
    w = [int(x) for x in np.asarray(words).reshape(-1)[:EXACT_WORDS]]
    return sum(d << (32 * j) for j, d in enumerate(w[:8])), w[8], w[9]


def _filter_core(engine, sh: dict, n: int, std_ratio: float):
    """The exchange steps of the statistical filter, given a rank's own rows: mean distances ``sh["avg"]`` and the
    original index ``sh["owned"]`` of every row (negative: a row that belongs to another rank).  The two sums are
    exact integer sums (``pc_sor_exact_sums``): whatever the number of ranks and however the cloud was cut -- slices
    of a replicated index, z slabs, one GPU -- mean, standard deviation and threshold carry the same bits."""
    words = yield engine.exact_sums(sh["avg"], sh["owned"], 0, 0.0)
    total, valid, big = exact_value(engine.to_host(words))
    if big:
        raise RuntimeError("[pc_amd] mean neighbour distances of 2^96 and more: outside the statistical filter's exact sums")
    if valid == 0:
        return engine.no_indices(), (0.0, 0.0, 0.0, 0)
    mean = total / (valid << 128)  # (integers: one correctly rounded division)
    words = yield engine.exact_sums(sh["avg"], sh["owned"], 1, mean)
    sq, _, big = exact_value(engine.to_host(words))
    if big:
        raise RuntimeError("[pc_amd] squared deviations of 2^96 and more: outside the statistical filter's exact sums")
    with np.errstate(divide="ignore", invalid="ignore"):
        std = float(np.sqrt(np.float64(sq / (1 << 128)) / np.float64(valid - 1)))
    thr = mean + std_ratio * std
    mask = yield engine.mask_scatter(sh["avg"], sh["owned"], thr, n)
    return engine.mask_select(mask, n), (mean, std, thr, valid)


def sor_steps(engine, pts, k: int, std_ratio: float, cache: dict, rank: int, world: int):
    """Statistical outlier removal of one rank (reference semantics: filtering_adapter.py:91)."""
    b, e = shard_positions(len(pts), rank, world)
    sh = engine.shard_search(pts, cache, k, -1.0, b, e, want_avg=True)
    return (yield from _filter_core(engine, sh, len(pts), std_ratio))


def pipeline_steps(engine, pts, k: int, std_ratio: float, cache: dict, rank: int, world: int, want_normals: bool = True,
                   **search_kw):
    """kNN rows + statistical filter + normals from ONE search of the rank's slice: returns the slice's
    outputs (rows, mean distances, normals, owned indices), the kept indices and the statistics."""
    b, e = shard_positions(len(pts), rank, world)
    sh = engine.shard_search(pts, cache, k, -1.0, b, e, want_rows=True, want_avg=True, want_normals=want_normals, **search_kw)
    kept, stats = yield from _filter_core(engine, sh, len(pts), std_ratio)
    return sh, kept, stats


def normals_steps(engine, pts, k: int, radius: float, cache: dict, rank: int, world: int):
    """PCA normals of one rank's slice, then the (n, 3) result put together on every rank."""
    n = len(pts)
    b, e = shard_positions(n, rank, world)
    sh = engine.shard_search(pts, cache, k, radius, b, e, want_normals=True)
    full = yield engine.rows_scatter(sh["normals"], sh["owned"], n)
    return full


# ---- slabs: every rank indexes only its own part of the cloud -----------------------------------------
SLAB_BINS = 4096   # bins of the z histogram the cuts are chosen from
SLAB_HALO = 2.0    # halo in units of the neighbour distance of a cloud spread evenly over its bounding box


def slab_cuts(engine, pts, cache: dict, k: int, world: int):
    """Where the ranks' z intervals begin and end: ``(cuts, halo)`` with ``cuts[r] <= z < cuts[r + 1]`` owned by
    rank r (first and last cut infinite), chosen at bin edges of a z histogram so that the intervals hold about
    n / world points each, and the halo every rank indexes beyond its interval.  ``None`` where slabs do not
    pay (a cloud without depth along z, intervals thinner than their halos): the callers then shard the queries
    of a replicated index.  Every rank computes the same answer from the same cloud."""
    n = len(pts)
    if world <= 1:
        return [-np.inf, np.inf], 0.0
    mn, mx = engine.bbox(pts, cache)
    ext = np.asarray(mx, dtype=np.float64) - np.asarray(mn, dtype=np.float64)
    if n < 2 * world or not ext[2] > 0:
        return None
    hist = np.asarray(engine.axes_histogram(pts, cache, mn, mx, SLAB_BINS), dtype=np.int64)
    cum = np.cumsum(hist[2])
    width = ext / SLAB_BINS
    cuts = [-np.inf]
    for r in range(1, world):
        b = int(np.searchsorted(cum, (r * n) // world))  # first bin whose running count reaches the rank's share
        cuts.append(float(mn[2] + (b + 1) * width[2]))
    cuts.append(np.inf)
    halo = _bulk_halo(hist, width, n, n, k)
    inner = np.diff(np.array([mn[2]] + cuts[1:-1] + [mx[2]]))
    if not (np.diff(cuts[1:-1]) > 0).all() or inner.min() < 2.0 * halo:
        return None
    return cuts, halo


def _bulk_halo(hist, width, n_hist: int, n_points: int, k: int) -> float:
    """The halo of a slab: ``SLAB_HALO`` times the neighbour distance of ``n_points`` points spread evenly over the box
    of the cloud's bulk -- all but half a per cent of the ``n_hist`` histogrammed points at either end of every axis: a
    few stray points far out would otherwise inflate the volume, and with it the halo, until no slab is worth making."""
    bulk = np.empty(3)
    for a in range(3):
        c = np.cumsum(hist[a])
        b0, b1 = int(np.searchsorted(c, n_hist // 200)), int(np.searchsorted(c, n_hist - n_hist // 200))
        bulk[a] = (b1 - b0 + 1) * width[a]
    floor_ext = np.maximum(bulk, 1e-3 * bulk.max())
    return SLAB_HALO * float((k * np.prod(floor_ext) / max(n_points, 1)) ** (1.0 / 3.0))


SLAB_REPEAT_MAX = 1 << 16  # answers repeated across the ranks at most; beyond, the job goes to the replicated index


def _repeat_unvouched(engine, pts, cache: dict, k: int, sh: dict, box, rank: int, world: int, total: int, want_normals: bool):
    """The answers no slab could vouch for, repeated exactly without any full index: the ranks put their lists of
    such queries together, EVERY rank answers all of them against its own slab, the rows travel and are merged
    (the slabs cover the cloud, so the k nearest of the union of the ranks' k nearest are the true neighbours).
    Every exchange is a sum of arrays in which each rank fills its own part of zeros.  Patches the rank's rows in
    ``sh`` and the mean distances in ``box`` (on every rank alike; ``None``: rows only)."""
    counts = engine.zeros((world,), "f64")
    counts[rank] = float(sh["n_uncertain"])
    counts = yield counts
    counts = np.asarray(engine.to_host(counts)).astype(np.int64)
    at = int(counts[:rank].sum())
    mine = slice(at, at + int(counts[rank]))
    rows = sh["uncertain_rows"]
    queries = engine.zeros((total,), "i32")
    queries[mine] = sh["owned"][rows]
    queries = yield queries
    ridx, rd2, rcnt = engine.slab_requery(pts, cache, sh["slab_key"], queries, k)
    all_idx, all_d2, all_cnt = engine.zeros((world, total, k), "i32"), engine.zeros((world, total, k), "f64"), engine.zeros((world, total), "i32")
    all_idx[rank], all_d2[rank], all_cnt[rank] = ridx, rd2, rcnt
    all_idx = yield all_idx
    all_d2 = yield all_d2
    all_cnt = yield all_cnt
    midx, md2, mcnt = engine.merge_rows(all_idx, all_d2, all_cnt)
    avg, nrm = engine.rows_consumers(pts, cache, queries, midx, mcnt, want_normals)
    if box is not None:
        box[queries] = avg  # (every rank patches every repeated query: the array stays the same everywhere)
    sh["idx"][rows], sh["cnt"][rows] = midx[mine], mcnt[mine]
    sh["d2"][rows] = engine.cast_like(md2[mine], sh["d2"])
    if "avg" in sh:
        sh["avg"][rows] = avg[mine]
    if want_normals:
        sh["normals"][rows] = nrm[mine]
    sh["n_repeated"] = total


def pipeline_steps_slab(engine, pts, k: int, std_ratio: float, cache: dict, rank: int, world: int, want_normals: bool = True,
                        **search_kw):
    """``pipeline_steps`` with an index of the rank's own slab instead of a replicated one: rows, mean distances
    and normals of the rank's z interval from ONE search of the slab.  The ranks first add up the number of
    answers their slabs cannot vouch for; those are repeated across the ranks (``_repeat_unvouched``), and if they
    are too many for that, the whole job goes to the replicated index: results are exact either way.  The filter
    then exchanges what ``_filter_core`` exchanges -- twice ten integer words and the bit mask of the removed
    points --, never an array of the cloud's size.  Returns what ``pipeline_steps`` returns; rows with ``owned`` < 0 belong to a neighbouring
    rank."""
    plan = slab_cuts(engine, pts, cache, k, world)
    if plan is None:
        return (yield from pipeline_steps(engine, pts, k, std_ratio, cache, rank, world, want_normals, **search_kw))
    cuts, halo = plan
    n = len(pts)
    sh = engine.slab_search(pts, cache, k, -1.0, cuts[rank], cuts[rank + 1], halo, want_avg=True, want_normals=want_normals,
                            **search_kw)
    count = engine.zeros((1,), "f64")
    count[0] = float(sh["n_uncertain"])
    count = yield count
    total = int(float(count[0]))
    if total > min(SLAB_REPEAT_MAX, max(n // 32, 1)):  # (the halo does not fit this cloud at all)
        return (yield from pipeline_steps(engine, pts, k, std_ratio, cache, rank, world, want_normals, **search_kw))
    if total > 0:
        yield from _repeat_unvouched(engine, pts, cache, k, sh, None, rank, world, total, want_normals)
    kept, stats = yield from _filter_core(engine, sh, n, std_ratio)
    return sh, kept, stats


def voxel_layer_cuts(engine, raw, cache: dict, voxel: float, world: int):
    """Voxel layers along z per rank: ``(box_min, box_max, origin_z, cuts, hist, width)`` with rank r owning the layers
    ``cuts[r] <= iz < cuts[r + 1]`` of the voxel grid ``voxel_down_sample`` lays over the cloud, cut so that the ranks hold
    about n / world raw points each.  ``None`` where the cloud has too few layers for that."""
    n = len(raw)
    mn, mx = engine.bbox(raw, cache)
    mn, mx = np.asarray(mn, dtype=np.float64), np.asarray(mx, dtype=np.float64)
    oz = mn[2] - 0.5 * voxel                       # (the voxel grid's origin: min bound minus half a voxel)
    nlayers = int(np.floor((mx[2] - oz) / voxel)) + 1
    if world <= 1:
        return mn, mx, oz, [0, nlayers], None, None
    if n < 2 * world or nlayers < 4 * world:
        return None
    hist = np.asarray(engine.axes_histogram(raw, cache, mn, mx, SLAB_BINS), dtype=np.int64)
    cum = np.cumsum(hist[2])
    width = (mx - mn) / SLAB_BINS
    cuts = [0]
    for r in range(1, world):
        b = int(np.searchsorted(cum, (r * n) // world))
        cuts.append(int(round((mn[2] + (b + 1) * width[2] - oz) / voxel)))
    cuts.append(nlayers)
    if not (np.diff(cuts) > 0).all():
        return None
    return mn, mx, oz, cuts, hist, width


def voxel_pipeline_steps_slab(engine, raw, voxel: float, k: int, std_ratio: float, cache: dict, rank: int, world: int,
                              want_normals: bool = True, **search_kw):
    """Voxel down-sampling + ``pipeline_steps_slab`` where no rank ever voxelises the whole cloud: every rank holds the
    raw cloud, voxelises the z layers of voxels it owns plus a halo of layers (``pc_voxel_downsample_layers``: the
    whole cloud's voxel grid, so a rank's voxels are one contiguous run of ``voxel_down_sample``'s output and indices
    translate by one addition), indexes those and answers its own.  The ranks exchange their voxel counts (the
    offsets), the number of answers their slabs cannot vouch for and what the filter exchanges.  If any answer is
    unvouched, or the cloud has too few layers, every rank takes the replicated voxel pass and ``pipeline_steps_slab``
    instead: exact either way.  Returns ``(rows, kept, stats, m)``: rows and kept indices in indices of the whole
    down-sampled cloud of ``m`` voxels."""
    def replicated():
        vox = engine.voxel_downsample_dev(raw, None, None, voxel, cache)[0] if hasattr(engine, "voxel_downsample_dev") \
            else engine.voxel_downsample(raw, None, None, voxel, cache)[0]
        sh, kept, stats = yield from pipeline_steps_slab(engine, vox, k, std_ratio, {}, rank, world, want_normals, **search_kw)
        return sh, kept, stats, len(vox)

    plan = voxel_layer_cuts(engine, raw, cache, voxel, world)
    if world <= 1 or plan is None:
        return (yield from replicated())
    mn, mx, oz, cuts, hist, width = plan
    lo, hi = cuts[rank], cuts[rank + 1]
    # own layers and a guessed halo in ONE pass over the raw cloud (the halo wanted follows from the number of voxels,
    # which only the exchange of the counts tells: guessed from half as many voxels as points; layers missing after
    # that are fetched, layers too many dropped)
    guess = int(np.ceil(_bulk_halo(hist, width, len(raw), max(len(raw) // 2, 1), k) / voxel))
    g_lo, g_hi = (lo - guess if rank > 0 else lo), (hi + guess if rank < world - 1 else hi)
    local, iz = engine.voxel_layers(raw, cache, voxel, mn, mx, g_lo, g_hi)
    at = lambda *layers: engine.sorted_positions(iz, layers)  # (iz ascends: the voxel order is (iz, iy, ix))
    p_lo, p_hi = at(lo, hi)
    n_own = p_hi - p_lo
    counts = engine.zeros((world,), "f64")
    counts[rank] = float(n_own)
    counts = yield counts
    counts = np.asarray(engine.to_host(counts)).astype(np.int64)
    m = int(counts.sum())
    halo_layers = int(np.ceil(_bulk_halo(hist, width, len(raw), m, k) / voxel))
    if min(np.diff(cuts)) < 2 * halo_layers or m < 2 * world:  # (every rank finds the same)
        return (yield from replicated())
    w_lo, w_hi = (lo - halo_layers if rank > 0 else lo), (hi + halo_layers if rank < world - 1 else hi)
    a, b = at(max(w_lo, g_lo), min(w_hi, g_hi))
    parts = [local[a:b]]
    if w_lo < g_lo:
        parts.insert(0, engine.voxel_layers(raw, cache, voxel, mn, mx, w_lo, g_lo)[0])
    if w_hi > g_hi:
        parts.append(engine.voxel_layers(raw, cache, voxel, mn, mx, g_hi, w_hi)[0])
    n_lower = len(parts[0]) if w_lo < g_lo else 0
    n_lower += p_lo - a
    local = engine.concat_rows(parts) if len(parts) > 1 else parts[0]
    base = int(counts[:rank].sum()) - n_lower  # index of local row 0 in the whole down-sampled cloud
    own_lo = oz + lo * voxel if rank > 0 else -np.inf
    own_hi = oz + hi * voxel if rank < world - 1 else np.inf
    sh = engine.slab_search(local, {}, k, -1.0, own_lo, own_hi, halo_layers * voxel, want_avg=True, want_normals=want_normals,
                            open_faces=(rank > 0, rank < world - 1), **search_kw)
    count = engine.zeros((1,), "f64")
    count[0] = float(sh["n_uncertain"])
    count = yield count
    if int(float(count[0])) > 0:
        return (yield from replicated())
    for key in ("owned", "idx"):
        a = sh[key]
        a[a >= 0] += base
    kept, stats = yield from _filter_core(engine, sh, m, std_ratio)
    return sh, kept, stats, m


def knn_slab_steps(engine, pts, k: int, cache: dict, rank: int, world: int, **search_kw):
    """kNN rows of the rank's z interval from an index of its own slab (a generator like the others here: drive
    it with ``run``).  The ranks decide TOGETHER what happens to answers no slab can vouch for: their number is
    summed over the ranks, a few are repeated across the ranks (``_repeat_unvouched``, exact), and if there are
    too many -- or slabs do not pay for this cloud, which every rank finds alike from the same cloud -- every rank
    returns ``None`` and the caller uses ``knn_sharded`` on all ranks.  Never one rank's slab rows beside another
    rank's slice of the replicated index: the two partitions of the cloud differ."""
    plan = slab_cuts(engine, pts, cache, k, world)
    if plan is None:
        return None
    cuts, halo = plan
    n = len(pts)
    sh = engine.slab_search(pts, cache, k, -1.0, cuts[rank], cuts[rank + 1], halo, **search_kw)
    count = engine.zeros((1,), "f64")
    count[0] = float(sh["n_uncertain"])
    count = yield count
    total = int(float(count[0]))
    if total > min(SLAB_REPEAT_MAX, max(n // 32, 1)):
        return None
    if total > 0:
        yield from _repeat_unvouched(engine, pts, cache, k, sh, None, rank, world, total, False)
    return sh


def knn_slab(engine, pts, k: int, cache: Optional[dict] = None, group=None, **search_kw):
    """``knn_slab_steps`` over the ranks of ``group``; falls back to ``knn_sharded`` on every rank alike."""
    cache = {} if cache is None else cache
    alone = not (dist.is_available() and dist.is_initialized())
    rank, world = (0, 1) if alone else (dist.get_rank(group), dist.get_world_size(group))
    sh = run(knn_slab_steps(engine, pts, k, cache, rank, world, **search_kw), group)
    if sh is None:
        b, e = shard_positions(len(pts), rank, world)
        sh = engine.shard_search(pts, cache, k, -1.0, b, e, want_rows=True, **search_kw)
    return sh


def sor_sharded(engine, pts, k: int, std_ratio: float, cache: Optional[dict] = None, group=None):
    """Statistical outlier removal with the queries sharded over the ranks of ``group``.

    Returns ``(kept_indices, (mean, std, threshold, valid))``: identical on every rank and, bit for bit, for
    every number of ranks."""
    cache = {} if cache is None else cache
    return run(sor_steps(engine, pts, k, std_ratio, cache, dist.get_rank(group), dist.get_world_size(group)), group)


def normals_sharded(engine, pts, k: int, radius: float, cache: Optional[dict] = None, group=None):
    """PCA normals with the queries sharded; the (n, 3) result on every rank (an engine array)."""
    cache = {} if cache is None else cache
    return run(normals_steps(engine, pts, k, radius, cache, dist.get_rank(group), dist.get_world_size(group)), group)


def knn_sharded(engine, pts, k: int, radius: float = -1.0, cache: Optional[dict] = None, group=None):
    """kNN rows of this rank's slice only: a dict with ``owned`` (original index of every row), ``idx``,
    ``d2``, ``cnt``, each with as many rows as the slice.  Nothing is exchanged: full neighbour lists stay on
    the GPU that computed them."""
    cache = {} if cache is None else cache
    n = len(pts)
    b, e = shard_positions(n, dist.get_rank(group), dist.get_world_size(group))
    return engine.shard_search(pts, cache, k, radius, b, e, want_rows=True)
