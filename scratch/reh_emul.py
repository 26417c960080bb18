"""voxel_pipeline_steps_slab for a pretended world on one GPU, config-4 shape at a given size (development aid)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd import distributed as PD
from point_cloud_processing_amd.hip_engine import HipEngine
n, world = int(sys.argv[1]), int(sys.argv[2])
eng = HipEngine(0)
xyz = torch.from_numpy(D.planar_patches(n, 5).astype(np.float32)).to(eng.device)
print("generated", flush=True)
def gen(r):
    g = PD.voxel_pipeline_steps_slab(eng, xyz, 0.05, 32, 2.0, {"xyz": xyz}, r, world, want_normals=False, d2_dtype=torch.float32)
    return g
t0 = time.perf_counter()
res = PD.run_emulated(gen, world)
torch.cuda.synchronize()
print("done", time.perf_counter() - t0, [(len(k), m, sh.get("sub_n")) for sh, k, st, m in res], flush=True)
