"""Minimal point-cloud container with the Open3D method surface the app's callers touch.

The reference passes ``o3d.geometry.PointCloud`` objects between its layers and calls eight
of their methods (reference: app/Core/pointcloud_preprocess.py:45-53,96,104-115;
app/Infrastructure/filtering_adapter.py:38-49,88-94; app/Core/mesh_reconstruction.py:112;
app/Controllers/app_controller.py:49,53).  This module offers the same surface on plain
N x 3 float64 arrays; every computing method forwards to the active *engine* (the HIP
engine in production).  Nothing here does arithmetic on the CPU beyond bookkeeping.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from . import _engine


def _as_points(a) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if arr.size == 0:
        arr = arr.reshape(0, 3)
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise RuntimeError(f"expected an (N, 3) array, got shape {arr.shape}")
    return arr


class Vector3dVector(np.ndarray):
    """``o3d.utility.Vector3dVector`` stand-in: an (N, 3) float64 array (``len`` = N)."""

    def __new__(cls, data=()):
        return _as_points(data).view(cls)


class KDTreeSearchParamKNN:
    def __init__(self, knn: int = 30):
        self.knn = int(knn)


class KDTreeSearchParamRadius:
    def __init__(self, radius: float):
        self.radius = float(radius)


class KDTreeSearchParamHybrid:
    def __init__(self, radius: float, max_nn: int):
        self.radius = float(radius)
        self.max_nn = int(max_nn)


class AxisAlignedBoundingBox:
    def __init__(self, min_bound, max_bound):
        self.min_bound = np.asarray(min_bound, dtype=np.float64)
        self.max_bound = np.asarray(max_bound, dtype=np.float64)

    def get_min_bound(self):
        return self.min_bound

    def get_max_bound(self):
        return self.max_bound

    def get_center(self) -> np.ndarray:
        return (self.min_bound + self.max_bound) * 0.5

    def get_extent(self) -> np.ndarray:
        return self.max_bound - self.min_bound

    def get_max_extent(self) -> float:
        return float((self.max_bound - self.min_bound).max())


class PointCloud:
    """Points (+ optional normals / colors) as (N, 3) float64 arrays.

    ``points`` / ``normals`` / ``colors`` hand out read-only views; assign a new array to
    change them.  That is what lets the engine keep device copies, spatial grids and the
    per-point mean neighbour distances cached between calls (``filter_pcd`` re-runs the
    outlier filter up to six times on one cloud, filtering_adapter.py:104-131).
    """

    def __init__(self, other=None):
        self._points = np.zeros((0, 3), dtype=np.float64)
        self._normals = np.zeros((0, 3), dtype=np.float64)
        self._colors = np.zeros((0, 3), dtype=np.float64)
        self._cache: dict = {}
        if other is None:
            return
        if isinstance(other, PointCloud) or hasattr(other, "points"):
            # copy-constructor (app_controller.py:49)
            self._points = _as_points(np.asarray(other.points)).copy()
            nrm = getattr(other, "normals", None)
            col = getattr(other, "colors", None)
            if nrm is not None and len(nrm):
                self._normals = _as_points(np.asarray(nrm)).copy()
            if col is not None and len(col):
                self._colors = _as_points(np.asarray(col)).copy()
        else:
            self._points = _as_points(other).copy()

    # -- attribute surface ---------------------------------------------------------------
    @staticmethod
    def _view(a: np.ndarray) -> Vector3dVector:
        v = a.view(Vector3dVector)
        v.flags.writeable = False
        return v

    @property
    def points(self) -> Vector3dVector:
        return self._view(self._points)

    @points.setter
    def points(self, value) -> None:
        self._points = _as_points(value).copy()
        self._cache.clear()

    @property
    def normals(self) -> Vector3dVector:
        return self._view(self._normals)

    @normals.setter
    def normals(self, value) -> None:
        self._normals = _as_points(value).copy()

    @property
    def colors(self) -> Vector3dVector:
        return self._view(self._colors)

    @colors.setter
    def colors(self, value) -> None:
        self._colors = _as_points(value).copy()

    def __len__(self) -> int:
        return int(self._points.shape[0])

    def __repr__(self) -> str:
        return f"PointCloud with {len(self)} points."

    def has_points(self) -> bool:
        return len(self) > 0

    def has_normals(self) -> bool:
        return len(self) > 0 and self._normals.shape[0] == len(self)

    def has_colors(self) -> bool:
        return len(self) > 0 and self._colors.shape[0] == len(self)

    def is_empty(self) -> bool:
        return len(self) == 0

    # -- bounds and rigid edits --------------------------------------------------------------
    def get_min_bound(self) -> np.ndarray:
        return self.get_axis_aligned_bounding_box().min_bound

    def get_max_bound(self) -> np.ndarray:
        return self.get_axis_aligned_bounding_box().max_bound

    def get_axis_aligned_bounding_box(self) -> AxisAlignedBoundingBox:
        if len(self) == 0:
            return AxisAlignedBoundingBox(np.zeros(3), np.zeros(3))
        mn, mx = _engine.get().bbox(self._points, self._cache)
        return AxisAlignedBoundingBox(mn, mx)

    def translate(self, translation, relative: bool = True) -> "PointCloud":
        """In place, returns self (Open3D semantics; pointcloud_preprocess.py:51)."""
        t = np.asarray(translation, dtype=np.float64).reshape(3)
        if not relative:
            t = t - self.get_axis_aligned_bounding_box().get_center()
        self._points = _engine.get().affine(self._points, self._cache, t, None, None)
        self._cache.clear()
        return self

    def scale(self, scale: float, center) -> "PointCloud":
        """p = (p - center) * scale + center, in place, returns self (pointcloud_preprocess.py:53)."""
        c = np.asarray(center, dtype=np.float64).reshape(3)
        self._points = _engine.get().affine(self._points, self._cache, None, float(scale), c)
        self._cache.clear()
        return self

    def normalize_to_unit_cube(self) -> Tuple[np.ndarray, float]:
        """Fused translate(-center) + scale(1/extent): one device pass instead of three.

        Same arithmetic, in the same order, as the two separate calls."""
        bbox = self.get_axis_aligned_bounding_box()
        center = bbox.get_center()
        ext = bbox.get_max_extent()
        s = (1.0 / ext) if ext > 0 else None
        self._points = _engine.get().affine(self._points, self._cache, -center, s, np.zeros(3))
        self._cache.clear()
        return center, ext

    # -- selection ---------------------------------------------------------------------------
    def select_by_index(self, indices: Sequence[int], invert: bool = False) -> "PointCloud":
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        if invert:
            keep = np.ones(len(self), dtype=bool)
            keep[idx] = False
            idx = np.nonzero(keep)[0]
        out = PointCloud()
        out._points = np.ascontiguousarray(self._points[idx])
        if self.has_normals():
            out._normals = np.ascontiguousarray(self._normals[idx])
        if self.has_colors():
            out._colors = np.ascontiguousarray(self._colors[idx])
        return out

    # -- the neighbourhood stage -----------------------------------------------------------
    def voxel_down_sample(self, voxel_size: float) -> "PointCloud":
        """Centroid per occupied voxel; new cloud, ascending voxel key order."""
        if not (voxel_size > 0.0):
            raise RuntimeError("[Open3D Error] voxel_size <= 0.")
        out = PointCloud()
        if len(self) == 0:
            return out
        p, n, c = _engine.get().voxel_downsample(
            self._points,
            self._normals if self.has_normals() else None,
            self._colors if self.has_colors() else None,
            float(voxel_size),
            self._cache,
        )
        out._points = p
        if n is not None:
            out._normals = n
        if c is not None:
            out._colors = c
        return out

    def estimate_normals(self, search_param=None, fast_normal_computation: bool = True) -> None:
        """PCA normals in place (pointcloud_preprocess.py:109-114)."""
        if search_param is None:
            search_param = KDTreeSearchParamKNN(30)
        if not fast_normal_computation:
            raise NotImplementedError("only the closed-form (fast) eigen-solver is on the device path")
        if isinstance(search_param, KDTreeSearchParamHybrid):
            k, r = search_param.max_nn, search_param.radius
        elif isinstance(search_param, KDTreeSearchParamKNN):
            k, r = search_param.knn, -1.0
        else:
            raise NotImplementedError("radius-only neighbourhoods are not used by the app and not on the device path")
        if len(self) == 0:
            return
        prior = self._normals if self.has_normals() else None
        self._normals = _engine.get().normals(self._points, int(k), float(r), prior, self._cache)

    def orient_normals_consistent_tangent_plane(self, k: int) -> None:
        """Consistent normal signs by propagation along a minimum spanning tree (Hoppe et al. 1992, the
        scheme behind PointCloud.orient_normals_consistent_tangent_plane, pointcloud_preprocess.py:115):
        graph = the k-nearest-neighbour graph taken as undirected, edge weight 1 - |n_i . n_j|, seed = the
        highest point of each connected piece with its normal turned to +z, a normal is flipped when it
        opposes its tree parent.  Search, spanning tree and sign propagation all run on the engine.
        Open3D joins disconnected pieces through a Delaunay-based spanning tree; here each piece is
        oriented on its own, so signs agree with Open3D's within a piece up to a global flip."""
        n = len(self)
        if self._normals is None or n == 0:
            raise RuntimeError("[Open3D Error] No normals in the PointCloud. Call EstimateNormals() first.")
        kk = max(2, min(int(k), n, 64))
        kk = max(2, min(kk, (2 ** 32 - 1) // n))  # (the engine numbers the neighbour slots in 32 bits)
        self._normals, _ = _engine.get().orient_normals(self._points, self._normals, kk, self._cache)

    def remove_statistical_outlier(self, nb_neighbors: int, std_ratio: float, print_progress: bool = False):
        if nb_neighbors < 1 or not (std_ratio > 0):
            raise RuntimeError("[Open3D Error] Illegal input parameters, the number of neighbors and "
                               "standard deviation ratio must be positive.")
        if len(self) == 0:
            return PointCloud(), []
        kept = _engine.get().sor(self._points, int(nb_neighbors), float(std_ratio), self._cache)
        return self.select_by_index(kept), kept

    def remove_radius_outlier(self, nb_points: int, radius: float, print_progress: bool = False):
        if nb_points < 1 or not (radius > 0):
            raise RuntimeError("[Open3D Error] Illegal input parameters, the number of points and radius "
                               "must be positive.")
        if len(self) == 0:
            return PointCloud(), []
        # only count > nb_points matters: let the engine stop counting at nb_points + 1
        cnt = _engine.get().radius_count(self._points, float(radius), self._cache, cap=int(nb_points) + 1)
        kept = np.nonzero(cnt > int(nb_points))[0]
        return self.select_by_index(kept), kept

    def compute_nearest_neighbor_distance(self) -> np.ndarray:
        if len(self) == 0:
            return np.zeros(0, dtype=np.float64)
        return _engine.get().nn_distance(self._points, self._cache)

    def search_knn(self, queries, k: int, radius: Optional[float] = None):
        """Exact kNN of ``queries`` (None = the cloud's own points) -> (idx, d2, count)."""
        q = None if queries is None else _as_points(queries)
        return _engine.get().knn(self._points, q, int(k), -1.0 if radius is None else float(radius), self._cache)


def wrap(pcd):
    """Return a :class:`PointCloud` for ``pcd`` (itself, or a copy of any object with .points)."""
    return pcd if isinstance(pcd, PointCloud) else PointCloud(pcd)


def unwrap_like(result: PointCloud, template):
    """Give ``result`` back in the type of ``template`` (a real Open3D cloud stays Open3D)."""
    if isinstance(template, PointCloud):
        return result
    out = type(template)()
    vec = type(template.points)
    out.points = vec(np.asarray(result.points))
    if result.has_normals():
        out.normals = vec(np.asarray(result.normals))
    if result.has_colors():
        out.colors = vec(np.asarray(result.colors))
    return out
