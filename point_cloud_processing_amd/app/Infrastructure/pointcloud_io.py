"""The one data-parallel step of the reference's loaders: the path-seeded random subsample.

reference app/Infrastructure/pointcloud_io.py:41-45 derives a generator from the file path (blake2b of the
path, low 32 bits) and :56-59 (LAS/LAZ), :146-150 (E57), :207-224 (PLY/PCD) draw
``rng.choice(total, max_points, replace=False)`` when a file holds more points than ``max_points``.  File
parsing (laspy, pye57, Open3D I/O) stays with the application; once the arrays are in a cloud, the subsample
is the same index draw on the host (the reference's indices, number for number) and a row gather on the GPU.
"""
from __future__ import annotations

import hashlib
from pathlib import Path
from typing import Optional, Union

import numpy as np

from ... import _engine
from ...geometry import PointCloud, wrap


def stable_rng_from_path(path: Union[str, Path]) -> np.random.Generator:
    """The reference's generator for ``path`` (:41-45): the same seed for the UI and the CLI."""
    h = hashlib.blake2b(str(Path(path)).encode("utf-8"), digest_size=8).digest()
    return np.random.default_rng(int.from_bytes(h, "little", signed=False) & 0xFFFFFFFF)


def subsample_indices(total: int, max_points: Optional[int], path: Union[str, Path]) -> np.ndarray:
    """Indices the reference's loader keeps (:56-61): a draw without replacement, or everything."""
    if max_points is not None and total > max_points:
        return stable_rng_from_path(path).choice(total, max_points, replace=False)
    return np.arange(total)


def subsample_point_cloud(pcd, max_points: Optional[int], path: Union[str, Path]) -> PointCloud:
    """``pcd`` cut down to ``max_points`` points the way the loaders do it; points, normals and colors are
    gathered on the device and stay there.  A cloud that is small enough comes back as it is."""
    cloud = wrap(pcd)
    total = len(cloud)
    if max_points is None or total <= max_points:
        return cloud
    idx = subsample_indices(total, max_points, path)
    eng = _engine.get()
    if getattr(eng, "device_resident", False):
        from ...geometry import LazyIndices

        cloud.get_axis_aligned_bounding_box()  # (the coordinates are on the device from here on)
        return cloud.select_by_index(LazyIndices(eng, eng.upload(idx.astype(np.int32)), len(idx)))
    return cloud.select_by_index(idx)
