"""Timing of the normal-orientation step (kNN(k) rows + pc_orient_normals) on one GPU (development aid)."""
import sys, os, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine, _ptr
from point_cloud_processing_amd import _lib

eng = HipEngine(0)
lib, ctx = eng.lib, eng.ctx
for name, n, k in (("torus", 1_000_000, 50), ("torus", 4_000_000, 50), ("torus", 4_000_000, 16), ("planes", 4_000_000, 30)):
    pts = (D.torus(n, 6) if name == "torus" else D.planar_patches(n, 5)).astype(np.float32)
    xyz = torch.from_numpy(pts).to(eng.device)
    h = C.c_void_p()
    eng._check(lib.pc_grid_build(ctx, _ptr(xyz), _lib.PC_F32, n, 0.0, k, C.byref(h), eng.stream()), "grid")
    idx = torch.empty((n, k), dtype=torch.int32, device=eng.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=eng.device)
    nrm = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    eng._check(lib.pc_normals(ctx, h, 0, n, 30, -1.0, None, _ptr(nrm), eng.stream()), "normals")
    flip = torch.from_numpy(np.where(np.random.default_rng(1).random(n) < 0.5, -1.0, 1.0)).to(eng.device)
    nrm *= flip[:, None]
    def knn():
        eng._check(lib.pc_knn(ctx, h, None, _lib.PC_F64, n, k, -1.0, _ptr(idx), None, _lib.PC_F32, _ptr(cnt), eng.stream()), "knn")
    pieces = C.c_int64(0)
    def orient():
        eng._check(lib.pc_orient_normals(ctx, _ptr(xyz), _lib.PC_F32, _ptr(nrm), _ptr(idx), _ptr(cnt), n, k,
                                         C.byref(pieces), eng.stream()), "orient")
    for fn, label in ((knn, "kNN rows"), (orient, "orient")):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(f"{name} n={n} k={k}: {label} {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    print(f"   pieces {pieces.value}", flush=True)
    lib.pc_grid_destroy(ctx, h)
