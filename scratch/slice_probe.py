"""One slice of a replicated index searched a few times (development aid; run under rocprofv3 --kernel-trace)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd import distributed as PD
from point_cloud_processing_amd.hip_engine import HipEngine
eng = HipEngine(0)
n, k, world, r = 10_000_000, 16, int(sys.argv[1]), int(sys.argv[2])
cloud = torch.from_numpy(D.uniform_cube(n, 3).astype(np.float32)).to(eng.device)
cache = {"xyz": cloud}
eng.grid(cloud, cache, k)
b, e = PD.shard_positions(n, r, world)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh = eng.shard_search(cloud, cache, k, -1.0, b, e, want_rows=True, d2_dtype=torch.float32)
    torch.cuda.synchronize(); print(f"slice {r}/{world}: {(time.perf_counter()-t0)*1e3:.3f} ms", flush=True)
