"""The neighbourhood-stage part of the reference's ``reconstruct_bpa``: the ball radii.

reference app/Core/mesh_reconstruction.py:110-117 estimates the Ball-Pivoting radii from
the mean nearest-neighbour distance (kNN with k = 2 over the whole cloud).  Ball pivoting
itself (:120-129) is sequential surface reconstruction inside Open3D and stays there:
``reconstruct_bpa`` keeps the reference's signature, computes the radii on the GPU and hands
the pivoting to Open3D when that library is importable.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from ...geometry import PointCloud, wrap


def estimate_bpa_radii(pcd) -> List[float]:
    """``[b, 2b, 3b]`` with ``b = 1.5 * mean nearest-neighbour distance`` (:112-117)."""
    print("[BPA] Estimating radii from nearest neighbor distances ...")
    dists = np.asarray(wrap(pcd).compute_nearest_neighbor_distance())
    mean_dist = dists.mean()
    print(f"[BPA] mean NN distance = {mean_dist}")
    base = mean_dist * 1.5
    return [base, base * 2.0, base * 3.0]


def _as_open3d_cloud(pcd):
    """``pcd`` as an ``open3d.geometry.PointCloud`` (itself when it already is one)."""
    import open3d as o3d  # (absent: ImportError, turned into the RuntimeError below)

    if isinstance(pcd, o3d.geometry.PointCloud):
        return o3d, pcd
    src = wrap(pcd)
    out = o3d.geometry.PointCloud()
    out.points = o3d.utility.Vector3dVector(np.asarray(src.points))
    if src.has_normals():
        out.normals = o3d.utility.Vector3dVector(np.asarray(src.normals))
    if src.has_colors():
        out.colors = o3d.utility.Vector3dVector(np.asarray(src.colors))
    return o3d, out


def reconstruct_bpa(pcd, radii: Optional[List[float]] = None):
    """Same signature and steps as reference app/Core/mesh_reconstruction.py:106-129.  The radii estimate
    (the neighbour search of this function) runs on the GPU; Ball Pivoting and the mesh clean-up are
    Open3D's (:120-128) and need the ``open3d`` package: without it a RuntimeError says so, after the
    radii have been computed and printed."""
    if radii is None:
        radii = estimate_bpa_radii(pcd)
    print(f"[BPA] Running Ball Pivoting with radii={radii} ...")
    try:
        o3d, cloud = _as_open3d_cloud(pcd)
    except ImportError as e:
        raise RuntimeError("[BPA] Ball Pivoting is Open3D's surface reconstruction and is not part of the GPU "
                           f"neighbourhood stage; install open3d to run it (radii = {radii})") from e
    mesh = o3d.geometry.TriangleMesh.create_from_point_cloud_ball_pivoting(cloud, o3d.utility.DoubleVector(radii))
    mesh.remove_duplicated_vertices()
    mesh.remove_duplicated_triangles()
    mesh.remove_unreferenced_vertices()
    mesh.remove_degenerate_triangles()
    mesh.compute_vertex_normals()
    return mesh
