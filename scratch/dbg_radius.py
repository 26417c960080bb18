import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine
eng = HipEngine(0)
for name, pts in [("sphere 1M", D.bumpy_sphere(1_000_000, 5)), ("uniform 1M", D.uniform_cube(1_000_000, 2))]:
    pts = pts.astype(np.float64)
    for r in (0.002, 0.01, 0.05):
        cache = {}
        eng.radius_count(pts[:1000], r, {})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = eng.radius_count(pts, r, cache)
        print(f"{name} r={r}: full count {(time.perf_counter()-t0)*1e3:9.2f} ms mean {c.mean():.1f}", flush=True)
