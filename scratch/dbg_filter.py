import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine
from point_cloud_processing_amd import geometry as G, _engine
eng = HipEngine(0); _engine.set_engine(eng)
pts = D.bumpy_sphere(1_000_000, 5).astype(np.float64)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter()-t0)*1e3:10.2f} ms", flush=True); return r
pcd = G.PointCloud(); pcd.points = G.Vector3dVector(pts)
for rep in range(2):
    d = T("nn distance", lambda: pcd.compute_nearest_neighbor_distance())
    med = float(np.median(np.asarray(d)))
    out, ind = T("SOR nb=30 sr=3", lambda: pcd.remove_statistical_outlier(30, 3.0))
    out2, ind2 = T("SOR nb=30 sr=3.8 (cached?)", lambda: pcd.remove_statistical_outlier(30, 3.8))
    out3, ind3 = T("radius nb=4 r=5med", lambda: out.remove_radius_outlier(4, 5.0 * med))
    print(len(out.points), len(out3.points))
