"""BASELINE config 4 at full size on ONE GPU (development aid): N-point planar patches + clutter (extent 200 m),
voxel 0.05 m -> index -> kNN(k = 32) rows + SOR mean distances -> SOR statistics, then the share one of W
ranks would answer, and a brute-force check of sampled queries (fp64, same operation order) in torch."""
import sys, os, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine, _ptr
from point_cloud_processing_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
K = 32
eng = HipEngine(0)
lib, ctx = eng.lib, eng.ctx
t0 = time.perf_counter()
chunks = []
for c in range(0, N, 10_000_000):  # generated in pieces (progress lines, bounded host memory)
    m = min(10_000_000, N - c)
    chunks.append(torch.from_numpy(D.planar_patches(m, 5 + c // 10_000_000).astype(np.float32)).to(eng.device))
    print(f"generated {c + m} points ({time.perf_counter() - t0:.0f} s)", flush=True)
xyz = torch.cat(chunks); del chunks
n = xyz.shape[0]

def timed(fn, label):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
    return r

out = torch.empty((n, 3), dtype=torch.float64, device=eng.device)
m = C.c_int64(0)
def vox():
    eng._check(lib.pc_voxel_downsample(ctx, _ptr(xyz), _lib.PC_F32, None, None, n, 0.05, _ptr(out), None, None, None, None,
                                       C.byref(m), eng.stream()), "voxel")
timed(vox, "voxel warm-up"); timed(vox, f"voxel_down_sample(0.05) of {n} points")
mm = m.value
print(f"  -> {mm} voxels", flush=True)
cent = out[:mm].contiguous(); del out, xyz
h = C.c_void_p()
def build():
    eng._check(lib.pc_grid_build(ctx, _ptr(cent), _lib.PC_F64, mm, 0.0, K, C.byref(h), eng.stream()), "grid")
timed(build, "index build")
idx = torch.empty((mm, K), dtype=torch.int32, device=eng.device)
d2 = torch.empty((mm, K), dtype=torch.float32, device=eng.device)
avg = torch.empty((mm,), dtype=torch.float64, device=eng.device)
def fused(b=0, e=None):
    e = mm if e is None else e
    eng._check(lib.pc_knn_fused(ctx, h, b, e, K, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F32, None, _ptr(avg), None, None,
                                eng.stream()), "fused")
timed(fused, "kNN(32) rows + SOR mean distance, all queries (first call)")
timed(fused, "kNN(32) rows + SOR mean distance, all queries")
for W in (2, 4, 8):
    timed(lambda: fused(0, mm // W), f"  the share of rank 0 of {W}")
mean, std, thr, valid = timed(lambda: eng.sor_stats(avg, 2.0), "SOR statistics")
kept = int(((avg > 0) & (avg < thr)).sum())
print(f"  mean {mean:.6g} std {std:.6g} threshold {thr:.6g}: kept {kept} of {mm}", flush=True)
# size-independent properties + sampled brute force
assert bool((d2[:, 1:] >= d2[:, :-1]).all()), "rows not ascending"
assert bool((d2[:, 0] == 0).all()), "self not first"
rng = np.random.default_rng(0)
bad = 0
for q in rng.integers(0, mm, 64):
    dx = cent[:, 0] - cent[q, 0]; dy = cent[:, 1] - cent[q, 1]; dz = cent[:, 2] - cent[q, 2]
    dd = dx * dx; dd = dd + dy * dy; dd = dd + dz * dz
    v, _ = torch.topk(dd, K + 8, largest=False, sorted=True)
    cand = torch.nonzero(dd <= v[-1]).flatten()
    order = np.lexsort((cand.cpu().numpy(), dd[cand].cpu().numpy()))[:K]
    want = cand.cpu().numpy()[order]
    if not np.array_equal(want, idx[q].cpu().numpy()) or not np.array_equal(dd[want].float().cpu().numpy(), d2[q].cpu().numpy()):
        bad += 1
print(f"sampled brute-force check: {64 - bad} of 64 rows identical", flush=True)
lib.pc_grid_destroy(ctx, h)
sys.exit(1 if bad else 0)
