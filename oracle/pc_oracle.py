"""ctypes binding of the CPU oracle (oracle/pc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.  Parity is unpinned by the reference (no tests / fixtures
there, arithmetic inside the un-vendored open3d==0.19.0 wheel): see pc_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_grid_create.restype = C.c_void_p
        _lib.orc_grid_create.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double]
        _lib.orc_grid_destroy.argtypes = [C.c_void_p]
        _lib.orc_voxel_downsample.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _pts(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1, 3)
    return a


def _check(rc, what):
    if rc == -1:
        raise ValueError(f"{what}: bad argument")
    if rc == -4:
        raise RuntimeError(f"{what}: voxel_size is too small.")
    if rc < 0:
        raise RuntimeError(f"{what}: oracle error {rc}")
    return rc


def num_threads() -> int:
    return int(lib().orc_num_threads())


def bbox(pts):
    pts = _pts(pts)
    mn = np.zeros(3)
    mx = np.zeros(3)
    lib().orc_bbox(_p(pts), C.c_int64(len(pts)), _p(mn), _p(mx))
    return mn, mx


def normalize(pts):
    out = _pts(pts).copy()
    center = np.zeros(3)
    scale = C.c_double(0)
    lib().orc_normalize(_p(out), C.c_int64(len(out)), _p(center), C.byref(scale))
    return out, center, scale.value


def voxel_index(pts, vs):
    pts = _pts(pts)
    ijk = np.zeros((len(pts), 3), dtype=np.int32)
    dims = np.zeros(3, dtype=np.int32)
    vmin = np.zeros(3)
    _check(lib().orc_voxel_index(_p(pts), C.c_int64(len(pts)), C.c_double(vs), _p(ijk), _p(dims), _p(vmin)),
           "voxel_index")
    return ijk, dims, vmin


def voxel_downsample(pts, vs, normals=None, colors=None, with_info=False):
    pts = _pts(pts)
    n = len(pts)
    nrm = None if normals is None else _pts(normals)
    col = None if colors is None else _pts(colors)
    op = np.zeros((n, 3))
    on = np.zeros((n, 3)) if nrm is not None else None
    oc = np.zeros((n, 3)) if col is not None else None
    cnt = np.zeros(n, dtype=np.int32)
    ijk = np.zeros((n, 3), dtype=np.int32)
    m = _check(lib().orc_voxel_downsample(_p(pts), _p(nrm), _p(col), C.c_int64(n), C.c_double(vs), _p(op), _p(on),
                                          _p(oc), _p(cnt), _p(ijk)), "voxel_downsample")
    res = (op[:m].copy(), None if on is None else on[:m].copy(), None if oc is None else oc[:m].copy())
    if with_info:
        return res + (cnt[:m].copy(), ijk[:m].copy())
    return res


class Grid:
    def __init__(self, pts, cell=0.0, target=4.0):
        self.pts = _pts(pts)
        self.h = lib().orc_grid_create(_p(self.pts), C.c_int64(len(self.pts)), C.c_double(cell), C.c_double(target))
        if not self.h:
            raise MemoryError("orc_grid_create")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_grid_destroy(C.c_void_p(self.h))
            self.h = None

    def knn(self, queries, k, radius=None):
        q = self.pts if queries is None else _pts(queries)
        nq = len(q)
        idx = np.empty((nq, k), dtype=np.int32)
        d2 = np.empty((nq, k), dtype=np.float64)
        cnt = np.empty(nq, dtype=np.int32)
        r2 = -1.0 if radius is None or radius < 0 else float(radius) * float(radius)
        _check(lib().orc_grid_knn(C.c_void_p(self.h), _p(q), C.c_int64(nq), C.c_int(k), C.c_double(r2), _p(idx),
                                  _p(d2), _p(cnt)), "grid_knn")
        return idx, d2, cnt

    def knn_self(self, k, s_begin=0, s_end=None, radius=None, want_d2=True):
        """Self-query in cell order (the CPU-baseline variant); rows at original indices."""
        n = len(self.pts)
        s_end = n if s_end is None else s_end
        idx = np.full((n, k), -1, dtype=np.int32)
        d2 = np.full((n, k), np.inf, dtype=np.float64) if want_d2 else None
        cnt = np.zeros(n, dtype=np.int32)
        r2 = -1.0 if radius is None or radius < 0 else float(radius) * float(radius)
        _check(lib().orc_grid_knn_self(C.c_void_p(self.h), C.c_int64(s_begin), C.c_int64(s_end), C.c_int(k),
                                       C.c_double(r2), _p(idx), _p(d2), _p(cnt)), "grid_knn_self")
        return idx, d2, cnt

    def radius_count(self, queries, radius):
        q = self.pts if queries is None else _pts(queries)
        cnt = np.empty(len(q), dtype=np.int32)
        _check(lib().orc_grid_radius_count(C.c_void_p(self.h), _p(q), C.c_int64(len(q)), C.c_double(radius), _p(cnt)),
               "radius_count")
        return cnt


def knn_brute(pts, queries, k, radius=None):
    pts = _pts(pts)
    q = pts if queries is None else _pts(queries)
    nq = len(q)
    idx = np.empty((nq, k), dtype=np.int32)
    d2 = np.empty((nq, k), dtype=np.float64)
    cnt = np.empty(nq, dtype=np.int32)
    r2 = -1.0 if radius is None or radius < 0 else float(radius) * float(radius)
    _check(lib().orc_knn_brute(_p(pts), C.c_int64(len(pts)), _p(q), C.c_int64(nq), C.c_int(k), C.c_double(r2),
                               _p(idx), _p(d2), _p(cnt)), "knn_brute")
    return idx, d2, cnt


def knn(pts, queries, k, radius=None):
    return Grid(pts, target=2.0 if k < 8 else 4.0).knn(queries, k, radius)


def nn_distance(pts):
    pts = _pts(pts)
    out = np.zeros(len(pts))
    _check(lib().orc_nn_distance(_p(pts), C.c_int64(len(pts)), _p(out)), "nn_distance")
    return out


def sor_avg(pts, k):
    pts = _pts(pts)
    avg = np.zeros(len(pts))
    _check(lib().orc_sor_avg(_p(pts), C.c_int64(len(pts)), C.c_int(k), _p(avg)), "sor_avg")
    return avg


def sor_threshold(avg, std_ratio):
    avg = np.ascontiguousarray(avg, dtype=np.float64)
    stats = np.zeros(4)
    mask = np.zeros(len(avg), dtype=np.uint8)
    _check(lib().orc_sor_threshold(_p(avg), C.c_int64(len(avg)), C.c_double(std_ratio), _p(stats), _p(mask)),
           "sor_threshold")
    return mask.astype(bool), stats


def sor(pts, k, std_ratio):
    avg = sor_avg(pts, k)
    mask, stats = sor_threshold(avg, std_ratio)
    return np.nonzero(mask)[0], avg, stats


def normals(pts, k, radius=-1.0, prior=None):
    pts = _pts(pts)
    pr = None if prior is None else _pts(prior)
    out = np.zeros((len(pts), 3))
    _check(lib().orc_normals(_p(pts), C.c_int64(len(pts)), C.c_int(k), C.c_double(radius), _p(pr), _p(out)),
           "normals")
    return out


def radius_count(pts, radius):
    return Grid(pts, target=4.0).radius_count(None, radius)


def fast_eigen3x3(cov6):
    c = np.ascontiguousarray(cov6, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_fast_eigen3x3(_p(c), _p(out))
    return out


def orient_normals(pts, nrm, idx, cnt=None):
    """Normal signs propagated along the minimum spanning tree of the neighbour graph: the host-side
    statement of what PointCloud.orient_normals_consistent_tangent_plane does at
    app/Core/pointcloud_preprocess.py:115 (Hoppe et al. 1992; Open3D adds Delaunay edges to join
    disconnected pieces, here every piece is oriented on its own).

    Graph: one undirected edge per neighbour slot (i, s) -> idx[i, s]; weight 1 - |n_i . n_j| rounded to
    fp32; edges are totally ordered by (weight bits, slot number), which makes the tree unique.  Seed of
    a piece: its highest point (lowest index among equals), turned towards +z; a normal is reversed when
    it opposes its tree parent.  Returns (normals, number of connected pieces)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order, connected_components, minimum_spanning_tree

    pts = np.asarray(pts, dtype=np.float64)
    nrm = np.array(nrm, dtype=np.float64, copy=True)
    n, k = idx.shape
    if n == 0:
        return nrm, 0
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    cols = idx.reshape(-1).astype(np.int64)
    ok = (cols >= 0) & (cols < n) & (cols != rows)
    if cnt is not None:
        ok &= (np.arange(k)[None, :] < np.asarray(cnt)[:, None]).reshape(-1)
    slot = np.nonzero(ok)[0]
    rows, cols = rows[ok], cols[ok]
    a, b = nrm[rows], nrm[cols]
    dot = (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]
    w = (1.0 - np.abs(dot)).astype(np.float32)
    w[~(w > 0)] = 0.0
    key = (w.view(np.uint32).astype(np.uint64) << np.uint64(32)) | slot.astype(np.uint64)
    rank = np.empty(len(key), dtype=np.float64)
    rank[np.argsort(key, kind="stable")] = np.arange(1, len(key) + 1, dtype=np.float64)
    # (i, j) and (j, i) may both be listed: the spanning tree routine takes the smaller of the two
    g = csr_matrix((rank, (rows, cols)), shape=(n, n))
    ncomp, label = connected_components(g, directed=False)
    mst = minimum_spanning_tree(g)
    mst = mst + mst.T
    z = pts[:, 2] + 0.0
    order = np.lexsort((np.arange(n), -z, label))  # per piece: highest point first, lowest index among equals
    first = np.ones(n, dtype=bool)
    first[1:] = label[order][1:] != label[order][:-1]
    parent = np.arange(n, dtype=np.int64)
    for s in order[first]:
        _, pred = breadth_first_order(mst, int(s), directed=False, return_predecessors=True)
        m = pred >= 0
        parent[m] = pred[m]
    # sign of a point = sign of its seed x product of the edge signs on its tree path (pointer doubling)
    pn = nrm[parent]
    pdot = (nrm[:, 0] * pn[:, 0] + nrm[:, 1] * pn[:, 1]) + nrm[:, 2] * pn[:, 2]
    acc = np.where(pdot < 0.0, -1.0, 1.0)
    acc[parent == np.arange(n)] = 1.0
    par = parent.copy()
    for _ in range(int(np.ceil(np.log2(max(n, 2)))) + 1):
        acc = acc * acc[par]
        par = par[par]
    seed_sign = np.where(nrm[:, 2] < 0.0, -1.0, 1.0)
    sign = acc * seed_sign[par]
    nrm[sign < 0] *= -1.0
    return nrm, int(ncomp)
