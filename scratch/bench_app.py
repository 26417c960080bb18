"""End-to-end timing of the app-facing calls from numpy clouds (development aid)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
import point_cloud_processing_amd as P
from point_cloud_processing_amd.app.Core import estimate_bpa_radii, normalize_point_cloud, preprocess_point_cloud
from point_cloud_processing_amd.app.Infrastructure import FilterSettings, filter_pcd
import io, contextlib

def t(label, fn, reps=2):
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {1e3 * dt:.1f} ms", flush=True)
    return out

for n in (35_000, 1_000_000, 4_000_000):
    pts = D.bumpy_sphere(n, 1)
    print(f"--- bumpy sphere, {n} points")
    pcd = t("normalize_point_cloud", lambda: normalize_point_cloud(P.PointCloud(pts)))[0]
    proc = t("preprocess_point_cloud(voxel 0.005)", lambda: preprocess_point_cloud(P.PointCloud(np.asarray(pcd.points)), voxel_size=0.005))
    print(f"   -> {len(proc.points)} points")
    proc2 = t("preprocess_point_cloud(no voxel)", lambda: preprocess_point_cloud(P.PointCloud(np.asarray(pcd.points)), voxel_size=None))
    t("filter_pcd(defaults)", lambda: filter_pcd(P.PointCloud(pts), FilterSettings(debug=False)))
    t("estimate_bpa_radii", lambda: estimate_bpa_radii(proc2))
