#!/usr/bin/env python3
"""Benchmarks of the neighbourhood stage on MI355X.  One JSON line on rank 0.

Default (what the driver runs): the headline, exact kNN queries/s on a 10 M-point synthetic cloud, k = 16.
One "step" = build the spatial index of the cloud (bounding box, cell keys, radix sort, cell table, record
gather) + the exact self-kNN of the rank's queries (int32 indices + fp32 squared distances written to HBM).
The cloud is resident in HBM before the timed region starts.

--gpus N (launched by torch.distributed.run, one rank per GPU): ONE cloud; the index is replicated (every
rank builds it) and rank r answers the queries of its slice of the index's sorted positions into outputs of
the slice's own size.  kNN rows need no exchange; the pipeline leg (kNN + statistical filter + normals) has
the filter's three small all-reduces over RCCL (twice ten integer words, once the bit mask of the removed points).  Total work is fixed: "scaling": "strong".

--config c2 | c3 | c4 | c5: the other BASELINE configurations (same JSON shape, their own metric).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def knn_algorithmic_bytes(nq: int, n: int, k: int, per_query_extra: int = 0) -> int:
    """12 B per query read + 12 B per reference point read once + (int32 idx + f32 d2) per neighbour."""
    return 12 * nq + 12 * n + (8 * k + per_query_extra) * nq


def cpu_knn_baseline(points: np.ndarray, k: int, sample: int):
    """The CPU oracle (C/OpenMP, grid-accelerated exact kNN, fp64) timed on this host."""
    from oracle import pc_oracle as O

    O.build()
    n = len(points)
    t0 = time.perf_counter()
    g = O.Grid(points, target=4.0)
    t_build = time.perf_counter() - t0
    t_query = float("inf")
    for _ in range(3):  # best of three: the first pass also pages the arrays in
        t0 = time.perf_counter()
        g.knn_self(k, 0, sample, want_d2=True)
        t_query = min(t_query, time.perf_counter() - t0)
    value = sample / (t_build * (sample / n) + t_query)
    return {
        "value": value,
        "unit": "queries/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"first {sample} of {n} cell-ordered self-queries, k={k}, against the full {n}-point grid; "
                  f"grid build {t_build:.2f}s charged pro rata, query {t_query:.2f}s (best of 3 passes)",
    }


def cpu_knn_scipy(points: np.ndarray, k: int, sample: int, cores: int):
    """An independent third-party CPU kNN on the same cloud: scipy's cKDTree (built once over the whole cloud,
    single-threaded as scipy builds it), ``sample`` of the cloud's own points as queries on ``cores`` threads."""
    try:
        from scipy.spatial import cKDTree
    except ImportError:
        return None
    n = len(points)
    t0 = time.perf_counter()
    tree = cKDTree(points)
    t_build = time.perf_counter() - t0
    q = points[:sample]
    t0 = time.perf_counter()
    tree.query(q, k=k, workers=cores)
    t_query = time.perf_counter() - t0
    return {
        "value": sample / (t_build * (sample / n) + t_query), "unit": "queries/s", "cores": cores, "kind": "scipy.spatial.cKDTree",
        "sample": f"first {sample} of {n} points as queries, k={k}, workers={cores}, against the tree of all {n} points; "
                  f"tree build {t_build:.2f}s (one thread) charged pro rata, query {t_query:.2f}s",
    }


def cpu_pipeline_baseline(points: np.ndarray, k: int, std_ratio: float, voxel=None):
    """The pipeline the way the reference runs it on the CPU: every Open3D call builds its own index and does
    its own search ([voxel] -> kNN rows, statistical filter, normals), here by the CPU oracle."""
    from oracle import pc_oracle as O

    O.build()
    t0 = time.perf_counter()
    p = points
    if voxel:
        p, _, _ = O.voxel_downsample(p, voxel)
    g = O.Grid(p, target=4.0)
    g.knn_self(k, want_d2=True)
    del g
    kept, _, _ = O.sor(p, k, std_ratio)
    O.normals(p, k)
    dt = time.perf_counter() - t0
    return {"value": len(points) / dt, "unit": "points/s", "cores": O.num_threads(), "kind": "port", "seconds": dt,
            "sample": f"the whole workload once ({len(points)} points"
                      + (f", voxel {voxel} -> {len(p)}" if voxel else "") + f"): kNN({k}) rows, statistical filter "
                      f"(own search), normals (own search), kept {len(kept)}"}


def _newest_profile(pattern):
    """The newest committed profile summary matching the glob (profiles/ are named per round: r01_, r02_, ...)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic(n, k, pipeline=False):
    """HBM bytes per launch of the dominant kernel from the committed counter passes (FETCH_SIZE / WRITE_SIZE
    are collected with rocprofv3 in runs of their own; they cannot be read in-process)."""
    for pattern in (("r*_pmc_traffic_pipeline.json",) if pipeline else ("r*_pmc_traffic.json", "r*_pmc_traffic_c2.json")):
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
            try:
                with open(path) as f:
                    pmc = json.load(f)
            except (OSError, ValueError):
                continue
            if pmc.get("points", 10_000_000) == n and pmc.get("k", 16) == k:
                key = "traffic_bytes_per_launch_pipeline" if pipeline and "traffic_bytes_per_launch_pipeline" in pmc else "traffic_bytes_per_launch"
                return pmc[key], os.path.relpath(path, ROOT) + " (rocprofv3 --pmc passes of this command)"
    return None, "not collected for this size"


def config_traffic(tag, key="traffic_bytes_per_step"):
    """Per-step HBM bytes (and whatever else the round's counter passes of `--config <tag>` recorded)."""
    pmc, path = _newest_profile(f"r*_pmc_traffic_{tag}.json")
    if pmc is None or key not in pmc:
        return None, "not collected", {}
    return pmc[key], path + " (rocprofv3 --pmc passes of this command)", pmc


class Harness:
    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.rehearse:  # every rank on GPU 0, gloo's collectives on device tensors: the multi-rank code path on a one-GPU box
            import faulthandler

            faulthandler.dump_traceback_later(900, exit=True)  # (a rehearsal that hangs says where, and ends)
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if self.world > 1 and args.rehearse:
            dist.init_process_group("gloo")
        elif self.world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        from point_cloud_processing_amd.hip_engine import HipEngine

        self.eng = HipEngine(self.local_rank)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, step, steps, warmup, events=None):
        """W untimed + K timed steps between fences; the MAX over the ranks of the elapsed time."""
        for _ in range(warmup):
            step(None)
        self.fence()
        t0 = time.perf_counter()
        for s in range(steps):
            step(events[s] if events else None)
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.eng.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def done(self, result):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.rank == 0:
            print(json.dumps(result))


def base_result(h, metric, value, unit, ms_per_step, workload, extra_config=None, scaling="strong", dtype="f64"):
    cfg = {"workload": workload}
    cfg.update(extra_config or {})
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": h.world, "steps": h.args.steps,
            "warmup": h.args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": cfg}


# --------------------------------------------------------------------------------------------------------
def bench_headline(h, n, k, label="kNN queries/sec (10M pts, k=16)", cloud=None):
    import datasets as D
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd import distributed as PD
    from point_cloud_processing_amd.hip_engine import _ptr

    torch, eng, args = h.torch, h.eng, h.args
    lib, ctx = eng.lib, eng.ctx
    pts64 = D.uniform_cube(n, 3) if cloud is None else cloud  # seed 3: the headline cloud
    xyz = torch.from_numpy(pts64.astype(np.float32)).to(eng.device)  # resident, fp32-exact
    b, e = PD.shard_positions(n, h.rank, h.world)
    m = e - b
    out_idx = torch.empty((m, k), dtype=torch.int32, device=eng.device)
    out_d2 = torch.empty((m, k), dtype=torch.float32, device=eng.device)
    owned = torch.empty((max(m, 1),), dtype=torch.int32, device=eng.device)
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]

    def step(events):
        hg = C.c_void_p()
        if events:
            events[0].record()
        eng._check(lib.pc_grid_build(ctx, _ptr(xyz), _lib.PC_F32, n, 0.0, k, C.byref(hg), eng.stream()), "grid")
        if events:
            events[1].record()
        if h.world == 1:  # rows at the original index, the way a single-GPU caller wants them
            eng._check(lib.pc_knn(ctx, hg, None, _lib.PC_F32, n, k, -1.0, _ptr(out_idx), _ptr(out_d2), _lib.PC_F32, None,
                                  eng.stream()), "knn")
        else:             # the rank's slice, rows in slice order + their original indices
            eng._check(lib.pc_knn_fused_compact(ctx, hg, b, e, k, -1.0, _ptr(out_idx), _ptr(out_d2), _lib.PC_F32, None,
                                                None, None, _ptr(owned), eng.stream()), "knn_compact")
        if events:
            events[2].record()
        lib.pc_grid_destroy(ctx, hg)

    elapsed = h.timed(step, args.steps, args.warmup, ev)
    build_ms = float(np.mean([a.elapsed_time(bb) for a, bb, _ in ev]))
    knn_ms = float(np.mean([bb.elapsed_time(c) for _, bb, c in ev]))
    ms_per_step = elapsed / args.steps * 1e3
    value = n / (elapsed / args.steps)  # the whole job: all ranks' queries over the slowest rank's time
    alg = knn_algorithmic_bytes(m, n, k)
    achieved = alg / (knn_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic(n, k) if h.world == 1 else (None, "single-GPU launch only")
    result = base_result(
        h, label, value, "queries/s", ms_per_step,
        f"uniform [0,1)^3 cloud, {n} points (seed 3, fp32-representable), exact self-kNN k={k}; step = index build "
        f"(replicated on every rank) + kNN of the rank's slice of sorted positions ({m} of {n} queries on rank 0)",
        {"points": n, "k": k, "parallelism": f"query-shard x{h.world}, replicated index"})
    result["roofline"] = {
        "bound": "hbm",
        "kernel": "k_tile_scan<Rec32,18,M_KNN> (+ k_wave_query / k_scan re-runs of uncertified queries)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": alg, "kernel_ms": knn_ms, "kernel_queries_per_s": m / (knn_ms * 1e-3),
    }
    sqp, sq_path = _newest_profile("r*_pmc_sq_tile_kernel.json")
    if sqp and h.world == 1 and n == 10_000_000 and k == 16:
        dv = sqp["derived"]
        # what actually bounds the kernel: vector-instruction issue (one wave64 instruction per four cycles and SIMD)
        result["roofline"]["valu_busy"] = dv.get("valu_busy_frac_of_kernel", dv["valu_issue_cycles_per_simd"] / (1.74e-3 * 2.4e9))
        result["roofline"]["valu_instructions_per_wave"] = dv["valu_instructions_per_wave"]
        result["roofline"]["valu_source"] = sq_path + " (SQ counters of the same command)"
        result["roofline"]["bound_in_practice"] = ("vector-instruction issue: at 100 % issue the kernel would take valu_busy x its time, i.e. "
                                                   "reach frac / valu_busy of the HBM roofline with this instruction count")
    result["step_parts_ms"] = {"index_build_replicated": build_ms, "knn_of_own_slice": knn_ms}

    if not args.no_pipeline:
        # kNN(16) + statistical outlier removal + PCA normals of the same cloud from ONE search per rank; the
        # filter's mean / standard deviation / kept mask are the only exchange (three all-reduces)
        cache = {"xyz": xyz}

        def pipeline():
            cache.pop(("grid", 2), None)  # (k_hint class of k = 16: the index is rebuilt every time)
            for key in [kk for kk in cache if kk != "xyz"]:
                cache.pop(key)
            return PD.run(PD.pipeline_steps(eng, xyz, k, 2.0, cache, h.rank, h.world, d2_dtype=torch.float32))

        pipeline()
        h.fence()
        t0 = time.perf_counter()
        reps = max(1, min(5, args.steps))
        for _ in range(reps):
            sh, kept, stats = pipeline()
        h.fence()
        pt = (time.perf_counter() - t0) / reps
        if h.world > 1:
            t = torch.tensor([pt], dtype=torch.float64, device=eng.device)
            h.dist.all_reduce(t, op=h.dist.ReduceOp.MAX)
            pt = float(t.item())
        result["pipeline"] = {
            "what": f"index build + ONE fused search of the rank's slice -> kNN({k}) rows + statistical filter (k={k}, std "
                    "2.0: mean distances, exact integer sums, all-reduce of ten words, removed-bit mask, select) + normals(knn=k)",
            "points_per_s": n / pt, "ms": pt * 1e3, "kept": len(kept), "algorithmic_bytes_per_point": 169,
            "hbm_frac": 169.0 * n / h.world / pt / 1e9 / HBM_PEAK_GBS,
        }
        if h.world == 1:
            tr, src = pmc_traffic(n, k, pipeline=True)
            result["pipeline"]["traffic"], result["pipeline"]["traffic_source"] = tr, src

    if h.rank == 0 and h.world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_knn_baseline(pts64, k, min(args.cpu_sample, n))
        # second opinion: a third-party CPU kNN (bounded sample); the speed-up is quoted against the faster of the two
        sc = cpu_knn_scipy(pts64, k, min(args.scipy_sample, n), result["cpu_baseline"]["cores"])
        result["cpu_baseline_scipy"] = sc
        fastest = max([result["cpu_baseline"]] + ([sc] if sc else []), key=lambda b: b["value"])
        result["speedup_vs_cpu"] = result["value"] / fastest["value"]
        result["speedup_vs_cpu_against"] = fastest["kind"]
        if "pipeline" in result:
            result["pipeline"]["cpu_baseline"] = cpu_pipeline_baseline(pts64, k, 2.0)
            result["pipeline"]["speedup_vs_cpu"] = result["pipeline"]["points_per_s"] / result["pipeline"]["cpu_baseline"]["value"]
    elif h.rank == 0:
        result["cpu_baseline"] = None
    return result


def bench_c2(h):
    """BASELINE configs[1]: 1 M uniform points, kNN k = 16 + statistical filter + normals on one MI355X."""
    h.args.no_pipeline = False
    r = bench_headline(h, 1_000_000, 16, label="kNN queries/sec (1M pts, k=16)")
    r["config"]["workload"] = "configs[1]: " + r["config"]["workload"]
    return r


def bench_c3(h):
    """BASELINE configs[2]: 120 k-point LiDAR frames streamed for 1000 frames, voxel 0.1 m + normals (knn 30)."""
    import datasets as D
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd.hip_engine import _ptr

    torch, eng, args = h.torch, h.eng, h.args
    lib, ctx = eng.lib, eng.ctx
    distinct, frames_total = args.distinct_frames, args.frames
    frames = [torch.from_numpy(D.lidar_frame(4 + f).astype(np.float32)).to(eng.device) for f in range(distinct)]
    n = frames[0].shape[0]
    cent = torch.empty((n, 3), dtype=torch.float64, device=eng.device)   # pre-allocated, reused by every frame
    nrm = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    lat, voxels = [], []

    def frame(f):
        m = C.c_int64(0)
        eng._check(lib.pc_voxel_downsample(ctx, _ptr(frames[f % distinct]), _lib.PC_F32, None, None, n, 0.1, _ptr(cent),
                                           None, None, None, None, C.byref(m), eng.stream()), "voxel")
        hg = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(cent), _lib.PC_F64, m.value, 0.0, 30, C.byref(hg), eng.stream()), "grid")
        eng._check(lib.pc_normals(ctx, hg, 0, m.value, 30, -1.0, None, _ptr(nrm), eng.stream()), "normals")
        lib.pc_grid_destroy(ctx, hg)
        return m.value

    for f in range(min(20, distinct)):
        frame(f)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for f in range(frames_total):
        t0 = time.perf_counter()
        voxels.append(frame(f))
        torch.cuda.synchronize()  # the frame's normals are complete in HBM
        lat.append(time.perf_counter() - t0)
    t_all = time.perf_counter() - t_all
    lat_ms = np.array(lat) * 1e3
    mv = float(np.mean(voxels))
    alg = (12 * n + 4 * n + 24 * mv) + (24 * mv + 32 * mv + 4 * mv) + (24 * mv + 24 * mv + 24 * mv)
    r = base_result(h, "LiDAR frames/sec (120k pts/frame: voxel 0.1 m + normals knn=30)", frames_total / t_all, "frames/s",
                    float(np.mean(lat_ms)),
                    f"configs[2]: {frames_total} frames of {n} points ({distinct} distinct seeded frames, cycled), resident in "
                    f"HBM; per frame voxel_down_sample(0.1) -> {mv:.0f} voxels (fp64 centroids) -> index -> PCA normals of 30 "
                    "neighbours; output buffers allocated once; latency = call to completed normals",
                    {"frames": frames_total, "points_per_frame": n}, scaling="weak")
    r["steps"], r["warmup"] = frames_total, min(20, distinct)
    r["latency_ms"] = {"p50": float(np.percentile(lat_ms, 50)), "p99": float(np.percentile(lat_ms, 99)),
                       "mean": float(np.mean(lat_ms)), "max": float(np.max(lat_ms))}
    r["roofline"] = {"bound": "hbm", "kernel": "whole frame (voxel pass, index build, k_tile_scan<Rec64,34,M_NORMALS>): latency-bound",
                     "achieved": alg / (np.mean(lat_ms) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (np.mean(lat_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "not collected", "algorithmic_bytes_per_launch": alg}
    tr, src, pmc = config_traffic("c3")
    r["roofline"]["traffic"], r["roofline"]["traffic_source"] = tr, src
    if pmc:
        r["launches_per_frame"] = pmc.get("launches_per_step")
        r["kernel_time_us_per_frame"] = pmc.get("kernel_time_us_per_step")
    if not args.no_cpu:
        from oracle import pc_oracle as O

        O.build()
        ts = []
        for f in range(3):
            p = D.lidar_frame(4 + f)
            t0 = time.perf_counter()
            vp, _, _ = O.voxel_downsample(p, 0.1)
            O.normals(vp, 30)
            ts.append(time.perf_counter() - t0)
        r["cpu_baseline"] = {"value": 1.0 / min(ts), "unit": "frames/s", "cores": O.num_threads(), "kind": "port",
                             "sample": "3 of the frames: voxel 0.1 + normals(30) by the CPU oracle, best frame"}
    return r


def bench_c4(h):
    """BASELINE configs[3] on ONE GPU: 100 M-point terrestrial scan, voxel 0.05 m + kNN k = 32 + statistical filter."""
    import datasets as D
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd import distributed as PD
    from point_cloud_processing_amd.hip_engine import _ptr

    torch, eng, args = h.torch, h.eng, h.args
    lib, ctx = eng.lib, eng.ctx
    N, K = args.points or 100_000_000, 32
    t0 = time.perf_counter()
    chunks = []
    for c in range(0, N, 10_000_000):  # generated in pieces: progress lines, bounded host memory
        m = min(10_000_000, N - c)
        chunks.append(torch.from_numpy(D.planar_patches(m, 5 + c // 10_000_000).astype(np.float32)).to(eng.device))
        print(f"[c4] generated {c + m} points ({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)
    xyz = torch.cat(chunks)
    del chunks
    n = xyz.shape[0]
    out = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    parts = {}

    def t(label, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = fn()
        torch.cuda.synchronize()
        parts[label] = (time.perf_counter() - t0) * 1e3
        return v

    def step():
        if h.world > 1 or args.slabs:
            # every rank voxelises its own z layers of voxels (+ halo layers) of the raw cloud, indexes and searches
            # those; the ranks exchange voxel counts, the number of answers no slab vouches for (any: every rank takes
            # the replicated voxel pass and index instead) and what the filter exchanges
            sh, kept, stats, mm = t("voxel_layers_slab_index_search_filter_ms", lambda: PD.run(PD.voxel_pipeline_steps_slab(
                eng, xyz, 0.05, K, 2.0, {"xyz": xyz}, h.rank, h.world, want_normals=False, d2_dtype=torch.float32)))
            parts["slabbed"] = "sub_n" in sh
            if parts["slabbed"]:
                parts["sub_n"] = sh["sub_n"]
            return mm, len(kept), sh
        m = C.c_int64(0)
        t("voxel_ms", lambda: eng._check(lib.pc_voxel_downsample(ctx, _ptr(xyz), _lib.PC_F32, None, None, n, 0.05, _ptr(out),
                                                                  None, None, None, None, C.byref(m), eng.stream()), "voxel"))
        mm = m.value
        cent = out[:mm]
        cache = {"xyz": cent}
        b, e = PD.shard_positions(mm, h.rank, h.world)
        t("index_build_ms", lambda: eng.grid(cent, cache, K))
        sh = t("search_ms", lambda: eng.shard_search(cent, cache, K, -1.0, b, e, want_rows=True, want_avg=True,
                                                      d2_dtype=torch.float32))
        kept, stats = t("filter_ms", lambda: PD.run(PD._filter_core(eng, sh, mm, 2.0)))
        return mm, len(kept), sh

    step()
    del_parts = dict(parts)
    reps = max(1, min(3, args.steps))
    tot = []
    for _ in range(reps):
        h.fence()
        t0 = time.perf_counter()
        mm, nk, sh = step()
        h.fence()
        tot.append(time.perf_counter() - t0)
    dt = min(tot)
    alg_knn = knn_algorithmic_bytes(len(sh["owned"]), mm, K, per_query_extra=8)
    search_ms = parts.get("search_ms", parts.get("voxel_layers_slab_index_search_filter_ms"))
    r = base_result(h, "points/sec (100M-pt scan: voxel 0.05 m + kNN k=32 + statistical filter)", n / dt, "points/s", dt * 1e3,
                    f"configs[3]: {n} points, 70 % on random planar patches + 30 % uniform clutter, extent 200 m (seeds 5..); "
                    f"voxel_down_sample(0.05) -> {mm} voxels (fp64 centroids) -> index -> kNN(32) rows + mean distances of the "
                    f"rank's slice -> statistical filter (std 2.0), kept {nk}",
                    {"points": n, "voxels": mm, "k": K,
                     "parallelism": (f"z-slab x{h.world}: per-rank voxel layers and slab index" if parts.get("slabbed") else
                                     f"query-shard x{h.world}, replicated voxel pass and index")})
    r["steps"], r["warmup"] = reps, 1
    r["step_parts_ms"] = parts
    r["roofline"] = {"bound": "hbm", "kernel": "k_tile_scan / k_batch_scan<Rec64,34,M_ALL> and k_ring_scan over the index levels the search makes (uneven cloud)",
                     "achieved": alg_knn / (search_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_knn / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "not collected", "algorithmic_bytes_per_launch": alg_knn,
                     "kernel_ms": search_ms}
    if "search_ms" not in parts:
        r["roofline"]["kernel"] += " -- timed with the slab's voxel layers, index and filter around it (slab path)"
    elif n == 100_000_000 and h.world == 1:
        tr, src, _ = config_traffic("c4")
        r["roofline"]["traffic"], r["roofline"]["traffic_source"] = tr, src
    if not args.no_cpu and h.rank == 0:
        sample = 4_000_000
        p = D.planar_patches(sample, 5, extent=200.0 * (sample / n) ** (1.0 / 3.0))  # same density, smaller box
        cb = cpu_pipeline_baseline(p, K, 2.0, voxel=0.05)
        cb["sample"] = f"{sample} points of the same density (box edge scaled), " + cb["sample"]
        r["cpu_baseline"] = cb
    return r


def bench_c5(h):
    """BASELINE configs[4]: ICP correspondence search between two 2 M-point clouds, kNN k = 1, 30 iterations."""
    import datasets as D
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd.hip_engine import _ptr

    torch, eng, args = h.torch, h.eng, h.args
    lib, ctx = eng.lib, eng.ctx
    n, iters = args.points or 2_000_000, 30
    target = D.torus(n, 6)
    rng = np.random.default_rng(7)
    source = (target[rng.permutation(n)] + rng.standard_normal((n, 3)) * 0.001).astype(np.float32)
    tgt = torch.from_numpy(target.astype(np.float32)).to(eng.device)
    src0 = torch.from_numpy(source.astype(np.float64)).to(eng.device)
    src = torch.empty_like(src0)
    q32 = torch.empty((n, 3), dtype=torch.float32, device=eng.device)
    idx = torch.empty((n, 1), dtype=torch.int32, device=eng.device)
    d2 = torch.empty((n, 1), dtype=torch.float32, device=eng.device)
    hg = C.c_void_p()
    eng._check(lib.pc_grid_build(ctx, _ptr(tgt), _lib.PC_F32, n, 0.0, 8, C.byref(hg), eng.stream()), "grid")
    mats = []
    for it in range(iters):  # a fixed, seeded sequence of decreasing rigid perturbations (scripted, not solved)
        a = 0.05 / (it + 1)
        T = np.eye(4)
        T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]]
        T[:3, 3] = np.array([0.01, -0.005, 0.002]) / (it + 1)
        mats.append((C.c_double * 16)(*T.reshape(16)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]

    def run():
        for it in range(iters):
            src.copy_(src0)
            eng._check(lib.pc_transform(ctx, _ptr(src), n, mats[it], eng.stream()), "transform")
            q32.copy_(src)  # (the fp32 target index takes fp32 queries through the tile kernel)
            ev[it][0].record()
            eng._check(lib.pc_knn(ctx, hg, _ptr(q32), _lib.PC_F32, n, 1, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F32, None,
                                  eng.stream()), "knn")
            ev[it][1].record()

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, min(3, args.steps))
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    knn_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    lib.pc_grid_destroy(ctx, hg)
    alg = knn_algorithmic_bytes(n, n, 1)
    r = base_result(h, "ICP correspondence queries/sec (2M x 2M, k=1, 30 iterations)", n * iters / dt, "queries/s", dt * 1e3 / iters,
                    f"configs[4]: target = {n}-point noisy torus (seed 6), source = the same points shuffled + noise, moved by 30 "
                    "scripted rigid perturbations; per iteration: transform, bin the queries into the target's index, exact k=1 "
                    "search (index built once)", {"points": n, "k": 1, "iterations": iters}, scaling="weak", dtype="f64")
    r["steps"], r["warmup"] = reps * iters, iters
    r["roofline"] = {"bound": "hbm", "kernel": "query binning (keys, radix sort, gather) + k_tile_scan<Rec32,4,M_KNN> over binned queries",
                     "achieved": alg / (knn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_source": "not collected",
                     "algorithmic_bytes_per_launch": alg, "kernel_ms": knn_ms}
    if n == 2_000_000:
        tr, src, pmc = config_traffic("c5")
        r["roofline"]["traffic"], r["roofline"]["traffic_source"] = tr, src
        if pmc:
            r["binning_share_of_search_kernel_time"] = pmc.get("binning_share_of_search_kernel_time")
            r["launches_per_iteration"] = pmc.get("launches_per_step")
    if not args.no_cpu:
        from oracle import pc_oracle as O

        O.build()
        t0 = time.perf_counter()
        g = O.Grid(target, target=2.0)
        tb = time.perf_counter() - t0
        q = np.ascontiguousarray(source.astype(np.float64))
        t0 = time.perf_counter()
        g.knn(q, 1)
        tq = time.perf_counter() - t0
        r["cpu_baseline"] = {"value": n / tq, "unit": "queries/s", "cores": O.num_threads(), "kind": "port",
                             "sample": f"one of the 30 iterations ({n} queries, k=1) by the CPU oracle's grid search, {tq:.2f}s; "
                                       f"index build {tb:.2f}s not charged"}
    return r


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="headline", choices=["headline", "c2", "c3", "c4", "c5"])
    ap.add_argument("--points", type=int, default=0, help="override the configuration's point count")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--distinct-frames", type=int, default=40)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the kNN+SOR+normals pipeline leg")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--scipy-sample", type=int, default=2_000_000, help="queries of the scipy cKDTree baseline")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank run on a one-GPU box: every rank on GPU 0, gloo instead of RCCL (numbers mean nothing)")
    ap.add_argument("--slabs", action="store_true", help="config c4 on one GPU through the multi-GPU slab path (a world of one)")
    args = ap.parse_args()
    h = Harness(args)
    if args.config == "headline":
        result = bench_headline(h, args.points or 10_000_000, args.k)
    else:
        result = {"c2": bench_c2, "c3": bench_c3, "c4": bench_c4, "c5": bench_c5}[args.config](h)
    h.done(result)
    return 0


if __name__ == "__main__":
    sys.exit(main())
