"""One slab rank of config 4 (rank R of W) on one GPU, three times: for a kernel trace (development aid)."""
import sys, os, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd import _lib, distributed as PD
from point_cloud_processing_amd.hip_engine import HipEngine, _ptr
R, W = int(sys.argv[1]), int(sys.argv[2])
eng = HipEngine(0)
N = 100_000_000
raw = torch.cat([torch.from_numpy(D.planar_patches(10_000_000, 5 + c).astype(np.float32)).to(eng.device) for c in range(10)])
out = torch.empty((N, 3), dtype=torch.float64, device=eng.device)
m = C.c_int64(0)
eng._check(eng.lib.pc_voxel_downsample(eng.ctx, _ptr(raw), _lib.PC_F32, None, None, N, 0.05, _ptr(out), None, None, None, None, C.byref(m), eng.stream()), "voxel")
cloud = out[:m.value]
del raw
c = {"xyz": cloud}
cuts, halo = PD.slab_cuts(eng, cloud, c, 32, W)
for it in range(3):
    cc = {"xyz": cloud, "bbox": c["bbox"]}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = {}
    sh = eng.slab_search(cloud, cc, 32, -1.0, cuts[R], cuts[R + 1], halo, want_avg=True, d2_dtype=torch.float32, stage_ms=st if it == 2 else None)
    torch.cuda.synchronize()
    print(it, (time.perf_counter() - t0) * 1e3, sh["sub_n"], sh["n_own"], st, flush=True)
