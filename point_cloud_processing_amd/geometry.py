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


def _call(name: str, *args):
    """Engine method ``name``; engines that keep results on their device offer it as ``name_dev``."""
    eng = _engine.get()
    return (getattr(eng, name + "_dev", None) or getattr(eng, name))(*args)


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


class LazyIndices:
    """An index list that may still live on the device (the second return value of the outlier filters).

    ``len`` is known without a transfer; the values come to the host the first time they are looked at
    (``np.asarray``, iteration, item access).  ``filter_pcd`` never looks at them."""

    def __init__(self, engine, dev, n: int):
        self._engine, self._dev, self._n, self._host = engine, dev, int(n), None

    def __len__(self) -> int:
        return self._n

    def host(self) -> np.ndarray:
        if self._host is None:
            self._host = np.asarray(self._engine.to_host(self._dev), dtype=np.int64).reshape(-1)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype)

    def __iter__(self):
        return iter(self.host().tolist())

    def __getitem__(self, i):
        return self.host()[i]


_ATTRS = ("points", "normals", "colors")


class PointCloud:
    """Points (+ optional normals / colors) as (N, 3) float64 arrays.

    ``points`` / ``normals`` / ``colors`` hand out read-only views; assign a new array to
    change them.  That is what lets the engine keep device copies, spatial grids and the
    per-point mean neighbour distances cached between calls (``filter_pcd`` re-runs the
    outlier filter up to six times on one cloud, filtering_adapter.py:104-131).

    Every attribute lives on the host, on the engine's device, or on both.  Results of engine calls stay on
    the device (the chains normalise -> voxel -> normals -> orientation and statistical -> radius filter run
    without a host copy in between, pointcloud_preprocess.py:42-55, 89-116; filtering_adapter.py:87-94) and
    come to the host the first time an attribute is read.
    """

    def __init__(self, other=None):
        empty = np.zeros((0, 3), dtype=np.float64)
        self._h = {"points": empty, "normals": empty, "colors": empty}  # host copies (None: device only)
        self._d: dict = {}   # device copies (engine arrays), where they exist
        self._rows = {"points": 0, "normals": 0, "colors": 0}
        self._cache: dict = {}
        if other is None:
            return
        if isinstance(other, PointCloud) or hasattr(other, "points"):
            # copy-constructor (app_controller.py:49)
            self._set_host("points", _as_points(np.asarray(other.points)).copy())
            nrm = getattr(other, "normals", None)
            col = getattr(other, "colors", None)
            if nrm is not None and len(nrm):
                self._set_host("normals", _as_points(np.asarray(nrm)).copy())
            if col is not None and len(col):
                self._set_host("colors", _as_points(np.asarray(col)).copy())
        else:
            self._set_host("points", _as_points(other).copy())

    # -- storage ---------------------------------------------------------------------------
    def _set_host(self, name: str, a: np.ndarray) -> None:
        self._h[name] = a
        self._d.pop(name, None)
        self._rows[name] = int(a.shape[0])

    def _set_dev(self, name: str, a, rows: int) -> None:
        """An engine result: a device array (kept as it is) or a host array (engines without a device)."""
        if isinstance(a, np.ndarray):
            self._set_host(name, a)
            return
        self._d[name] = a
        self._h[name] = None
        self._rows[name] = int(rows)

    def _host(self, name: str) -> np.ndarray:
        a = self._h[name]
        if a is None:
            a = _as_points(_engine.get().to_host(self._d[name]))
            self._h[name] = a
        return a

    def _arr(self, name: str):
        """What engine calls receive: the device copy where there is one, else the host array."""
        d = self._d.get(name)
        return d if d is not None else self._h[name]

    def _on_device(self) -> bool:
        return bool(self._d)

    # -- attribute surface ---------------------------------------------------------------
    @staticmethod
    def _view(a: np.ndarray) -> Vector3dVector:
        v = a.view(Vector3dVector)
        v.flags.writeable = False
        return v

    @property
    def points(self) -> Vector3dVector:
        return self._view(self._host("points"))

    @points.setter
    def points(self, value) -> None:
        self._set_host("points", _as_points(value).copy())
        self._cache.clear()

    @property
    def normals(self) -> Vector3dVector:
        return self._view(self._host("normals"))

    @normals.setter
    def normals(self, value) -> None:
        self._set_host("normals", _as_points(value).copy())

    @property
    def colors(self) -> Vector3dVector:
        return self._view(self._host("colors"))

    @colors.setter
    def colors(self, value) -> None:
        self._set_host("colors", _as_points(value).copy())

    def __len__(self) -> int:
        return self._rows["points"]

    def __repr__(self) -> str:
        return f"PointCloud with {len(self)} points."

    def has_points(self) -> bool:
        return len(self) > 0

    def has_normals(self) -> bool:
        return len(self) > 0 and self._rows["normals"] == len(self)

    def has_colors(self) -> bool:
        return len(self) > 0 and self._rows["colors"] == len(self)

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
        mn, mx = _engine.get().bbox(self._arr("points"), self._cache)
        return AxisAlignedBoundingBox(mn, mx)

    def get_center(self) -> np.ndarray:
        """Mean of the points (Open3D's ComputeCenter)."""
        if len(self) == 0:
            return np.zeros(3)
        return np.asarray(_engine.get().center(self._arr("points"), self._cache), dtype=np.float64)

    def _edit(self, t, s, c) -> None:
        if len(self):
            self._set_dev("points", _call("affine", self._arr("points"), self._cache, t, s, c), len(self))
        self._cache.clear()

    def translate(self, translation, relative: bool = True) -> "PointCloud":
        """In place, returns self (Open3D semantics; pointcloud_preprocess.py:51).  ``relative=False`` moves
        the centre (the mean of the points) to ``translation``."""
        t = np.asarray(translation, dtype=np.float64).reshape(3)
        if not relative:
            t = t - self.get_center()
        self._edit(t, None, None)
        return self

    def scale(self, scale: float, center) -> "PointCloud":
        """p = (p - center) * scale + center, in place, returns self (pointcloud_preprocess.py:53)."""
        self._edit(None, float(scale), np.asarray(center, dtype=np.float64).reshape(3))
        return self

    def normalize_to_unit_cube(self) -> Tuple[np.ndarray, float]:
        """Fused translate(-center) + scale(1/extent): one device pass instead of three.

        Same arithmetic, in the same order, as the two separate calls."""
        bbox = self.get_axis_aligned_bounding_box()
        center = bbox.get_center()
        ext = bbox.get_max_extent()
        self._edit(-center, (1.0 / ext) if ext > 0 else None, np.zeros(3))
        return center, ext

    # -- selection ---------------------------------------------------------------------------
    def select_by_index(self, indices: Sequence[int], invert: bool = False) -> "PointCloud":
        eng = _engine.current()
        lazy = isinstance(indices, LazyIndices)
        if not invert and getattr(eng, "device_resident", False) and (lazy or self._on_device()):
            # rows gathered on the device: no attribute visits the host
            out = PointCloud()
            idx = indices._dev if lazy else np.asarray(indices, dtype=np.int64).reshape(-1)
            m = len(indices)
            names = [n for n in _ATTRS if n == "points" or self._rows[n] == len(self)]
            got = eng.select({n: self._arr(n) for n in names}, idx, m, self._cache, out._cache)
            for n in names:
                out._set_dev(n, got[n], m)
            return out
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        if invert:
            keep = np.ones(len(self), dtype=bool)
            keep[idx] = False
            idx = np.nonzero(keep)[0]
        out = PointCloud()
        out._set_host("points", np.ascontiguousarray(self._host("points")[idx]))
        if self.has_normals():
            out._set_host("normals", np.ascontiguousarray(self._host("normals")[idx]))
        if self.has_colors():
            out._set_host("colors", np.ascontiguousarray(self._host("colors")[idx]))
        return out

    # -- the neighbourhood stage -----------------------------------------------------------
    def voxel_down_sample(self, voxel_size: float) -> "PointCloud":
        """Centroid per occupied voxel; new cloud, ascending voxel key order."""
        if not (voxel_size > 0.0):
            raise RuntimeError("[Open3D Error] voxel_size <= 0.")
        out = PointCloud()
        if len(self) == 0:
            return out
        p, n, c = _call("voxel_downsample", 
            self._arr("points"),
            self._arr("normals") if self.has_normals() else None,
            self._arr("colors") if self.has_colors() else None,
            float(voxel_size),
            self._cache,
        )
        m = len(p)
        out._set_dev("points", p, m)
        if n is not None:
            out._set_dev("normals", n, m)
        if c is not None:
            out._set_dev("colors", c, m)
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
        prior = self._arr("normals") if self.has_normals() else None
        self._set_dev("normals", _call("normals", self._arr("points"), int(k), float(r), prior, self._cache),
                      len(self))

    def orient_normals_consistent_tangent_plane(self, k: int) -> None:
        """Consistent normal signs by propagation along a minimum spanning tree (Hoppe et al. 1992, the
        scheme behind PointCloud.orient_normals_consistent_tangent_plane, pointcloud_preprocess.py:115):
        graph = the k-nearest-neighbour graph taken as undirected, edge weight 1 - |n_i . n_j|, seed = the
        highest point of each connected piece with its normal turned to +z, a normal is flipped when it
        opposes its tree parent.  Search, spanning tree and sign propagation all run on the engine.
        Open3D joins disconnected pieces through a Delaunay-based spanning tree; here each piece is
        oriented on its own, so signs agree with Open3D's within a piece up to a global flip."""
        n = len(self)
        if not self.has_normals():  # (Open3D's HasNormals(): one normal per point)
            raise RuntimeError("[Open3D Error] No normals in the PointCloud. Call EstimateNormals() first.")
        kk = max(2, min(int(k), n, 64))
        kk = max(2, min(kk, (2 ** 32 - 1) // n))  # (the engine numbers the neighbour slots in 32 bits)
        nrm, _ = _call("orient_normals", self._arr("points"), self._arr("normals"), kk, self._cache)
        self._set_dev("normals", nrm, n)

    def remove_statistical_outlier(self, nb_neighbors: int, std_ratio: float, print_progress: bool = False):
        if nb_neighbors < 1 or not (std_ratio > 0):
            raise RuntimeError("[Open3D Error] Illegal input parameters, the number of neighbors and "
                               "standard deviation ratio must be positive.")
        if len(self) == 0:
            return PointCloud(), []
        kept = _call("sor", self._arr("points"), int(nb_neighbors), float(std_ratio), self._cache)
        return self.select_by_index(kept), kept

    def remove_radius_outlier(self, nb_points: int, radius: float, print_progress: bool = False):
        if nb_points < 1 or not (radius > 0):
            raise RuntimeError("[Open3D Error] Illegal input parameters, the number of points and radius "
                               "must be positive.")
        if len(self) == 0:
            return PointCloud(), []
        kept = _call("radius_outlier", self._arr("points"), int(nb_points), float(radius), self._cache)
        return self.select_by_index(kept), kept

    def compute_nearest_neighbor_distance(self) -> np.ndarray:
        if len(self) == 0:
            return np.zeros(0, dtype=np.float64)
        return _engine.get().nn_distance(self._arr("points"), self._cache)

    def search_knn(self, queries, k: int, radius: Optional[float] = None):
        """Exact kNN of ``queries`` (None = the cloud's own points) -> (idx, d2, count)."""
        q = None if queries is None else _as_points(queries)
        return _engine.get().knn(self._host("points"), q, int(k), -1.0 if radius is None else float(radius), self._cache)


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
