"""Per-rank time of the slab path against the replicated index, for pretended worlds on ONE GPU (development aid
and the source of the per-rank scaling figures): every rank's work is run in turn and timed on its own.
Usage: python scratch/bench_slab.py headline|c4 [points]"""
import sys, os, time, json, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd import _lib, distributed as PD
from point_cloud_processing_amd.hip_engine import HipEngine, _ptr

eng = HipEngine(0)
lib, ctx = eng.lib, eng.ctx
what = sys.argv[1] if len(sys.argv) > 1 else "headline"

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); v = fn(); torch.cuda.synchronize()
    return v, (time.perf_counter() - t0) * 1e3

if what == "headline":
    n, k = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000, 16
    cloud = torch.from_numpy(D.uniform_cube(n, 3).astype(np.float32)).to(eng.device)
    kw = dict(d2_dtype=torch.float32)
    voxel_ms = 0.0
else:
    N, k = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000, 32
    chunks = [torch.from_numpy(D.planar_patches(min(10_000_000, N - c), 5 + c // 10_000_000,
                                                extent=200.0 * (N / 1e8) ** (1.0 / 3.0)).astype(np.float32)).to(eng.device)
              for c in range(0, N, 10_000_000)]
    raw = torch.cat(chunks); del chunks
    out = torch.empty((raw.shape[0], 3), dtype=torch.float64, device=eng.device)
    m = C.c_int64(0)
    def vox():
        eng._check(lib.pc_voxel_downsample(ctx, _ptr(raw), _lib.PC_F32, None, None, raw.shape[0], 0.05, _ptr(out), None, None,
                                           None, None, C.byref(m), eng.stream()), "voxel")
    vox(); _, voxel_ms = timed(vox)
    cloud = out[:m.value]
    n = m.value
    kw = dict(d2_dtype=torch.float32, want_avg=True)
    print(f"voxel pass {voxel_ms:.1f} ms -> {n} voxels", flush=True)

res = {"config": what, "n": n, "k": k, "voxel_ms": voxel_ms, "worlds": {}}
for world in (1, 2, 4, 8):
    # replicated index: build (every rank the same) + the slowest rank's slice
    cache = {"xyz": cloud}
    eng.grid(cloud, cache, k); cache = {"xyz": cloud}          # (warm the pool)
    _, build_ms = timed(lambda: eng.grid(cloud, cache, k))
    rep = []
    for r in range(world):
        b, e = PD.shard_positions(n, r, world)
        f = lambda: eng.shard_search(cloud, cache, k, -1.0, b, e, want_rows=True, want_avg=kw.get("want_avg", False),
                                     d2_dtype=torch.float32)
        f()
        rep.append(timed(f)[1])
    del cache
    # slabs: plan + subset + its index + search + check, per rank
    slab, sub, unc, own, stages = [], [], 0, 0, None
    plan_cache = {"xyz": cloud}
    plan, plan_ms = timed(lambda: PD.slab_cuts(eng, cloud, plan_cache, k, world))
    if plan is not None and world > 1:
        for r in range(world):
            c = {"xyz": cloud, "bbox": plan_cache["bbox"]}
            def one(stage=None, cc=c):
                cuts, halo = PD.slab_cuts(eng, cloud, cc, k, world)
                return eng.slab_search(cloud, cc, k, -1.0, cuts[r], cuts[r + 1], halo, stage_ms=stage, **kw)
            one(None, dict(c))  # (warm the pool and the allocator)
            sh, ms = timed(one)
            if r == world // 2:
                st = {}
                one(st, {"xyz": cloud, "bbox": plan_cache["bbox"]})
                stages = {a: round(b, 3) for a, b in st.items()}
            slab.append(ms); sub.append(sh["sub_n"]); unc += sh["n_uncertain"]; own += sh["n_own"]
            del sh, c
    vox_layers = []
    if what != "headline" and world > 1:
        vplan = PD.voxel_layer_cuts(eng, raw, {}, 0.05, world)
        if vplan is not None:
            mn_, mx_, oz_, vcuts, vh, vw = vplan
            hl = int(np.ceil(PD._bulk_halo(vh, vw, raw.shape[0], n, k) / 0.05))
            vc = {"xyz": raw}
            for r in range(world):
                def layers(r=r):
                    g = int(np.ceil(PD._bulk_halo(vh, vw, raw.shape[0], raw.shape[0] // 2, k) / 0.05))
                    p, iz = eng.voxel_layers(raw, vc, 0.05, mn_, mx_, vcuts[r] - g if r > 0 else vcuts[r],
                                             vcuts[r + 1] + g if r < world - 1 else vcuts[r + 1])
                    a_, b_ = eng.sorted_positions(iz, (vcuts[r] - hl, vcuts[r + 1] + hl))
                    return p[a_:b_]
                layers()
                vox_layers.append(timed(layers)[1])
    w = {"voxel_layers_rank_ms": vox_layers, "replicated_build_ms": build_ms, "replicated_search_ms_max": max(rep), "replicated_rank_ms": build_ms + max(rep),
         "slab_rank_ms_max": max(slab) if slab else None, "slab_rank_ms": slab, "slab_points": sub, "uncertain": unc,
         "own_total": own, "stages_of_a_middle_rank_ms": stages, "plan": None if plan is None else {"halo": plan[1]}}
    res["worlds"][world] = w
    print(world, json.dumps(w), flush=True)
one_gpu = res["worlds"][1]["replicated_rank_ms"]
for world, w in res["worlds"].items():
    rep = w["replicated_rank_ms"]
    line = f"W={world}: replicated {rep:.2f} ms/rank ({(one_gpu + voxel_ms) / (rep + voxel_ms):.2f}x)"
    if w["slab_rank_ms_max"]:
        line += f"; slabs {w['slab_rank_ms_max']:.2f} ms/rank ({(one_gpu + voxel_ms) / (w['slab_rank_ms_max'] + voxel_ms):.2f}x with the voxel pass replicated)"
        if w["voxel_layers_rank_ms"]:
            both = max(a + b for a, b in zip(w["voxel_layers_rank_ms"], w["slab_rank_ms"]))
            line += f"; with the voxel pass by layers {both:.2f} ms/rank ({(one_gpu + voxel_ms) / both:.2f}x)"
    print(line)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"slab_{what}.json"), "w"), indent=1)
