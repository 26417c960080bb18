import sys, os, time, cProfile, pstats
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets as D
from point_cloud_processing_amd.hip_engine import HipEngine
from point_cloud_processing_amd import geometry as G, _engine
from point_cloud_processing_amd.app.Infrastructure.filtering_adapter import filter_pcd, FilterSettings
eng = HipEngine(0); _engine.set_engine(eng)
pts = D.bumpy_sphere(1_000_000, 5).astype(np.float64)
def run():
    pcd = G.PointCloud(); pcd.points = G.Vector3dVector(pts)
    return filter_pcd(pcd, FilterSettings(debug=True))
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
