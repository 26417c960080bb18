"""Drop-in for the reference's ``app/Core/pointcloud_preprocess.py`` (neighbourhood stage).

Same two callables, same argument meaning, same return shapes and the same progress lines
as reference app/Core/pointcloud_preprocess.py:42-55 and :89-116; the Open3D calls behind
them run as HIP kernels through :class:`point_cloud_processing_amd.geometry.PointCloud`.
``smooth_mesh_soft`` (:122-136) works on triangle meshes and is not part of this stage.
"""
from __future__ import annotations

import numpy as np

from ...geometry import KDTreeSearchParamHybrid, PointCloud, unwrap_like, wrap

NORMAL_RADIUS_FRACTION = 0.02  # search radius = 2 % of the largest bounding-box edge (:104-106)
NORMAL_MAX_NN = 30             # at most 30 neighbours per normal (:112)
ORIENT_K = 50                  # neighbours of the orientation step (:115)


def normalize_point_cloud(pcd):
    """Centre the cloud on its bounding-box centre and scale its longest edge to 1.

    Mutates ``pcd`` and returns ``(pcd, center, scale)`` exactly like the reference
    (translate / scale are in-place in Open3D, pointcloud_preprocess.py:51-53).
    """
    cloud = wrap(pcd)
    if cloud is pcd:
        bbox = cloud.get_axis_aligned_bounding_box()
        center = bbox.get_center()
        scale = bbox.get_max_extent()
        print(f"[Normalize] bbox center = {center}, max_extent = {scale}")
        cloud.normalize_to_unit_cube()
        return cloud, np.asarray(center), float(scale)
    # a foreign (e.g. real Open3D) cloud: same edits through its own in-place methods
    bbox = cloud.get_axis_aligned_bounding_box()
    center = bbox.get_center()
    scale = bbox.get_max_extent()
    print(f"[Normalize] bbox center = {center}, max_extent = {scale}")
    out = pcd.translate(-center)
    if scale > 0:
        out = out.scale(1.0 / scale, center=(0.0, 0.0, 0.0))
    return out, np.asarray(center), float(scale)


def preprocess_point_cloud(pcd, voxel_size: float = 0.02):
    """Optional voxel down-sampling, then hybrid-radius PCA normals (in place on the result).

    ``voxel_size`` is in normalised units; ``None`` or <= 0 skips the down-sampling and the
    input object itself is returned with normals (pointcloud_preprocess.py:94-99).
    """
    cloud = wrap(pcd)
    n_before = len(cloud)
    if voxel_size is not None and voxel_size > 0:
        print(f"[Preprocess] Voxel downsample with size = {voxel_size}")
        proc = cloud.voxel_down_sample(voxel_size)
    else:
        print("[Preprocess] Skip voxel downsample (use full cloud)")
        proc = cloud
    print(f"[Preprocess] Points: {n_before} -> {len(proc)}")

    extent = proc.get_axis_aligned_bounding_box().get_max_extent()
    radius = extent * NORMAL_RADIUS_FRACTION
    print(f"[Preprocess] Estimating normals (radius={radius}) ...")
    proc.estimate_normals(search_param=KDTreeSearchParamHybrid(radius=radius, max_nn=NORMAL_MAX_NN))
    proc.orient_normals_consistent_tangent_plane(ORIENT_K)

    if isinstance(pcd, PointCloud):
        return proc
    if proc is cloud:
        # down-sampling skipped: the caller's own object receives the normals
        pcd.normals = type(pcd.points)(np.asarray(proc.normals))
        return pcd
    return unwrap_like(proc, pcd)
