"""Self-kNN with small k on uneven shapes (development aid): build + search ms."""
import sys, os
sys.argv = [sys.argv[0], "none"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_configs.py")).read().split("which = sys.argv")[0])
for k in (1, 2, 4, 8):
    knn_case(f"sphere 2M k{k}", D.bumpy_sphere(2_000_000, 5), k)
    knn_case(f"planes 4M k{k}", D.planar_patches(4_000_000, 5), k)
    knn_case(f"clusters 2M k{k}", D.gaussian_clusters(2_000_000, 6), k)
    knn_case(f"torus 2M k{k}", D.torus(2_000_000, 6), k)
