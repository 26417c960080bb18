"""BASELINE config 5 with the solver in the loop: point-to-plane ICP, 2 M source points against a 2 M target."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
import point_cloud_processing_amd as P
from point_cloud_processing_amd.registration import transform_from_vector6

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
tgt = D.torus(n, 6)  # (the spin about the torus axis is not observable: the recovered motion is not compared)
target = P.PointCloud(tgt)
t0 = time.perf_counter(); target.estimate_normals(P.KDTreeSearchParamKNN(16)); print(f"target normals: {1e3*(time.perf_counter()-t0):.1f} ms")
motion = transform_from_vector6(np.array([0.02, -0.015, 0.03, 0.01, -0.02, 0.015]))
rng = np.random.default_rng(3)
src = tgt[rng.permutation(n)] + rng.standard_normal((n, 3)) * 0.001
src = np.ascontiguousarray((src - motion[:3, 3]) @ motion[:3, :3])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = P.registration_icp(P.PointCloud(src), target, 0.1, max_iteration=30, relative_fitness=0.0, relative_rmse=0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"registration_icp {n} vs {n}: {res.iterations} iterations in {1e3*dt:.1f} ms ({1e3*dt/max(res.iterations,1):.2f} ms each, uploads included), "
          f"fitness {res.fitness:.4f} rmse {res.inlier_rmse:.3g}", flush=True)
