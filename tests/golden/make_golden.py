"""Mint the committed golden vectors (tests/golden/*.npz).

The reference holds no tests or fixtures and its arithmetic lives in the un-vendored
open3d==0.19.0 wheel, so these vectors cannot come from the reference itself ("parity
unpinned").  They are produced by independent third-party code where one exists --
scipy.spatial.cKDTree for neighbour sets/counts, numpy.linalg.eigh for the PCA direction,
numpy.floor for voxel ids -- and by closed-form known answers otherwise.  The oracle and the
HIP path are both tested against them.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datasets as D  # noqa: E402


def knn_case(p, k):
    """cKDTree neighbours, re-ordered by (fp64 d2 in the reference's operation order, index)."""
    kk = min(k + 8, len(p))
    _, ii = cKDTree(p).query(p, kk)
    ii = ii.reshape(len(p), kk)
    dx = p[:, None, :] - p[ii]
    d2 = dx[..., 0] * dx[..., 0]
    d2 = d2 + dx[..., 1] * dx[..., 1]
    d2 = d2 + dx[..., 2] * dx[..., 2]
    order = np.lexsort((ii, d2), axis=1)
    ii = np.take_along_axis(ii, order, 1)[:, :k]
    d2 = np.take_along_axis(d2, order, 1)[:, :k]
    return ii.astype(np.int32), d2


def pca_normals(p, idx):
    out = np.zeros((len(p), 3))
    for i in range(len(p)):
        q = p[idx[i]]
        c = np.cov(q.T, bias=True)
        w, v = np.linalg.eigh(c)
        out[i] = v[:, 0]
    return out


def main():
    clouds = {
        "uniform1k": D.uniform_cube(1000, 101),
        "clusters1k": D.gaussian_clusters(1000, 102),
        "sphere2k": D.bumpy_sphere(2000, 103),
    }
    for name, p in clouds.items():
        out = {"points": p}
        for k in (1, 8, 16):
            ii, d2 = knn_case(p, k)
            # golden neighbour sets must be unambiguous: no exact tie at the k-th place
            out[f"idx_k{k}"] = ii
            out[f"d2_k{k}"] = d2
        ii16 = out["idx_k16"]
        out["normals_k16"] = pca_normals(p, ii16)
        dd, _ = cKDTree(p).query(p, 2)
        out["nn_dist"] = dd[:, 1]
        r = 3.0 * float(np.median(dd[:, 1]))
        out["radius"] = np.array(r)
        out["radius_count"] = cKDTree(p).query_ball_point(p, r * (1 - 1e-12), return_length=True).astype(np.int32)
        # SOR per-point mean distance from the golden neighbour lists (k = 8, self included)
        out["sor_avg_k8"] = np.sqrt(out["d2_k8"]).sum(1) / 8.0
        vs = float((p.max(0) - p.min(0)).max()) / 9.7
        vmin = p.min(0) - 0.5 * vs
        out["voxel_size"] = np.array(vs)
        out["voxel_ijk"] = np.floor((p - vmin) / vs).astype(np.int32)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
