"""Timing of the non-headline BASELINE configs on one GPU (development aid, not the bench contract)."""
import sys, os, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine, _ptr
from point_cloud_processing_amd import _lib

eng = HipEngine(0)
lib, ctx = eng.lib, eng.ctx

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def knn_case(name, pts, k, fp64=False):
    n = len(pts)
    xyz = torch.from_numpy(pts if fp64 else pts.astype(np.float32)).to(eng.device)
    dt = _lib.PC_F64 if fp64 else _lib.PC_F32
    idx = torch.empty((n, k), dtype=torch.int32, device=eng.device)
    d2 = torch.empty((n, k), dtype=torch.float32, device=eng.device)
    info = {}
    def build_only():
        h = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(xyz), dt, n, 0.0, k, C.byref(h), eng.stream()), "grid")
        out = (C.c_double * 6)(); lib.pc_grid_info(h, out); info["g"] = list(out)
        lib.pc_grid_destroy(ctx, h)
    def full():
        h = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(xyz), dt, n, 0.0, k, C.byref(h), eng.stream()), "grid")
        eng._check(lib.pc_knn(ctx, h, None, dt, n, k, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F32, None, eng.stream()), "knn")
        lib.pc_grid_destroy(ctx, h)
    tb = timed(build_only); tf = timed(full)
    g = info["g"]
    print(f"{name:28s} n={n:9d} k={k:2d} build {tb:7.2f} ms  knn {tf-tb:8.2f} ms  {n/(tf*1e-3)/1e6:9.1f} Mq/s  "
          f"dims {int(g[0])}x{int(g[1])}x{int(g[2])} cell {g[3]:.4g} occ/cell {n/(g[0]*g[1]*g[2]):.2f}", flush=True)

which = sys.argv[1:] or ["uniform", "sphere", "planes", "lidar", "clusters", "fp64"]
if "uniform" in which: knn_case("uniform 1M k16", D.uniform_cube(1_000_000, 2), 16)
if "sphere" in which: knn_case("bumpy sphere 2M k16", D.bumpy_sphere(2_000_000, 5), 16)
if "planes" in which: knn_case("planar patches 4M k32", D.planar_patches(4_000_000, 5), 32)
if "lidar" in which: knn_case("lidar frame 120k k30", D.lidar_frame(4), 30)
if "clusters" in which: knn_case("gaussian clusters 2M k16", D.gaussian_clusters(2_000_000, 6), 16)
if "fp64" in which: knn_case("uniform 1M k16 fp64 grid", D.uniform_cube(1_000_000, 2, fp32=False), 16, fp64=True)

def pipeline_case(name, pts, voxel, k):
    """voxel down-sample (fp64 centroids) -> grid -> kNN, like BASELINE configs 3 and 4"""
    n = len(pts)
    xyz = torch.from_numpy(pts.astype(np.float32)).to(eng.device)
    out = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    m = C.c_int64(0)
    def vox():
        eng._check(lib.pc_voxel_downsample(ctx, _ptr(xyz), _lib.PC_F32, None, None, n, voxel, _ptr(out), None, None,
                                           None, None, C.byref(m), eng.stream()), "voxel")
    tv = timed(vox)
    mm = m.value
    cent = out[:mm].contiguous()
    idx = torch.empty((mm, k), dtype=torch.int32, device=eng.device)
    d2 = torch.empty((mm, k), dtype=torch.float32, device=eng.device)
    nrm = torch.empty((mm, 3), dtype=torch.float64, device=eng.device)
    def full():
        h = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(cent), _lib.PC_F64, mm, 0.0, k, C.byref(h), eng.stream()), "grid")
        eng._check(lib.pc_knn_fused(ctx, h, 0, mm, k, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F32, None, None, _ptr(nrm), None,
                                    eng.stream()), "fused")
        lib.pc_grid_destroy(ctx, h)
    tf = timed(full)
    print(f"{name:28s} n={n:9d} -> voxels {mm:9d}  voxel {tv:7.2f} ms  grid+kNN({k})+normals {tf:8.2f} ms", flush=True)

if "c3" in which: pipeline_case("C3 lidar frame voxel 0.1", D.lidar_frame(4), 0.1, 30)
if "c4" in which: pipeline_case("C4 planar 4M voxel 0.05", D.planar_patches(4_000_000, 5), 0.05, 32)

def icp_case(name, n, iters=5):
    """BASELINE config 5 shape: external queries (a moved copy of the target), k = 1, repeated"""
    tgt = D.torus(n, 6)
    rng = np.random.default_rng(7)
    src = tgt[rng.permutation(n)] + rng.normal(0, 0.002, (n, 3))
    a = 0.02
    Rm = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    src = (src @ Rm.T + np.array([0.01, -0.005, 0.003])).astype(np.float32)
    xyz = torch.from_numpy(tgt.astype(np.float32)).to(eng.device)
    q = torch.from_numpy(src).to(eng.device)
    idx = torch.empty((n, 1), dtype=torch.int32, device=eng.device)
    d2 = torch.empty((n, 1), dtype=torch.float32, device=eng.device)
    h = C.c_void_p()
    eng._check(lib.pc_grid_build(ctx, _ptr(xyz), _lib.PC_F32, n, 0.0, 1, C.byref(h), eng.stream()), "grid")
    def one():
        eng._check(lib.pc_knn(ctx, h, _ptr(q), _lib.PC_F32, n, 1, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F32, None, eng.stream()), "knn")
    t = timed(one, iters)
    lib.pc_grid_destroy(ctx, h)
    print(f"{name:28s} n={n:9d} external k=1: {t:8.2f} ms per iteration  {n/(t*1e-3)/1e6:9.1f} Mq/s", flush=True)

def radius_case(name, pts, mult=5.0):
    """remove_radius_outlier shape: count within mult x median NN distance (filtering_adapter.py:72,94)"""
    n = len(pts)
    xyz = torch.from_numpy(pts.astype(np.float32)).to(eng.device)
    h = C.c_void_p()
    eng._check(lib.pc_grid_build(ctx, _ptr(xyz), _lib.PC_F32, n, 0.0, 2, C.byref(h), eng.stream()), "grid")
    nn = torch.empty(n, dtype=torch.float64, device=eng.device)
    def nnd():
        eng._check(lib.pc_nn_distance(ctx, h, 0, n, _ptr(nn), eng.stream()), "nnd")
    tn = timed(nnd)
    r = mult * float(nn.median())
    lib.pc_grid_destroy(ctx, h)
    cnt = torch.empty(n, dtype=torch.int32, device=eng.device)
    def full():
        g = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(xyz), _lib.PC_F32, n, r, 0, C.byref(g), eng.stream()), "grid")
        eng._check(lib.pc_radius_count(ctx, g, None, _lib.PC_F32, n, 0, n, r, _ptr(cnt), eng.stream()), "count")
        lib.pc_grid_destroy(ctx, g)
    tc = timed(full)
    print(f"{name:28s} n={n:9d} nn-distance {tn:7.2f} ms; radius {r:.4g}: grid+count {tc:8.2f} ms, mean count {float(cnt.double().mean()):.1f}", flush=True)

def filter_case(name, pts):
    from point_cloud_processing_amd import geometry as G, _engine
    from point_cloud_processing_amd.app.Infrastructure.filtering_adapter import filter_pcd, FilterSettings
    _engine.set_engine(eng)
    def run():
        pcd = G.PointCloud(); pcd.points = G.Vector3dVector(pts.astype(np.float64))
        out = filter_pcd(pcd, FilterSettings(debug=False, adaptive=False, radius_abs=None))
        return len(out.points)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m = run(); torch.cuda.synchronize()
    print(f"{name:28s} n={len(pts):9d} filter_pcd defaults: {(time.perf_counter()-t0)*1e3:8.2f} ms, kept {m}", flush=True)

if "icp" in which: icp_case("C5 torus 2M vs 2M", 2_000_000)
if "radius" in which: radius_case("radius count sphere 1M", D.bumpy_sphere(1_000_000, 5)); radius_case("radius count uniform 1M", D.uniform_cube(1_000_000, 2))
if "filter" in which: filter_case("filter_pcd sphere 1M", D.bumpy_sphere(1_000_000, 5))

def c3_frames(nframes=60):
    """BASELINE config 3: a stream of LiDAR frames, voxel 0.1 m -> grid -> kNN(30) + normals; per-frame latency"""
    frames = [torch.from_numpy(D.lidar_frame(4 + f).astype(np.float32)).to(eng.device) for f in range(8)]
    n = len(frames[0]); k = 30
    out = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    nrm = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
    m = C.c_int64(0)
    lat = []
    for f in range(nframes + 5):
        xyz = frames[f % len(frames)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng._check(lib.pc_voxel_downsample(ctx, _ptr(xyz), _lib.PC_F32, None, None, n, 0.1, _ptr(out), None, None,
                                           None, None, C.byref(m), eng.stream()), "voxel")
        mm = m.value
        h = C.c_void_p()
        eng._check(lib.pc_grid_build(ctx, _ptr(out), _lib.PC_F64, mm, 0.0, k, C.byref(h), eng.stream()), "grid")
        eng._check(lib.pc_normals(ctx, h, 0, mm, k, -1.0, None, _ptr(nrm), eng.stream()), "normals")
        torch.cuda.synchronize()
        lib.pc_grid_destroy(ctx, h)
        if f >= 5: lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.sort(np.array(lat))
    print(f"C3 stream: {nframes} frames of {n} points -> ~{mm} voxels: p50 {np.percentile(lat,50):.2f} ms  p99 {np.percentile(lat,99):.2f} ms  "
          f"({n/np.percentile(lat,50)*1e3/1e6:.1f} M input points/s)", flush=True)

if "c3frames" in which: c3_frames()
