"""The production engine: PointCloud methods -> C ABI -> HIP kernels on one MI355X.

PyTorch is used for device memory (tensors as buffers), the current stream and nothing else;
every computation is a kernel of libpc_amd.so.  Construction raises when the library or a
GPU is missing: there is deliberately no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceGrid:
    """Owns a pc_grid handle (the radix-sorted uniform grid of one cloud)."""

    def __init__(self, eng: "HipEngine", xyz: torch.Tensor, n: int, k_hint: int, cell: float = 0.0):
        self.eng = eng
        self.xyz = xyz  # keep the coordinates alive as long as the grid
        self.n = n
        h = C.c_void_p()
        dtype = _lib.PC_F32 if xyz.dtype == torch.float32 else _lib.PC_F64
        eng._check(eng.lib.pc_grid_build(eng.ctx, _ptr(xyz), dtype, n, float(cell), int(k_hint), C.byref(h),
                                         eng.stream()), "pc_grid_build")
        self.h = h
        self.k_hint = k_hint

    def info(self):
        out = (C.c_double * 6)()
        self.eng.lib.pc_grid_info(self.h, out)
        return {"dims": (int(out[0]), int(out[1]), int(out[2])), "cell": out[3], "n": int(out[4]),
                "bytes": int(out[5])}

    def close(self):
        if self.h:
            self.eng.lib.pc_grid_destroy(self.eng.ctx, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipIcpSession:
    """Device state of one ICP run: the moving source, the target with its index (fp64, so that the fp64
    source takes the tile kernel as binned external queries) and the current correspondences."""

    def __init__(self, eng: "HipEngine", src: np.ndarray, tgt: np.ndarray, tgt_normals: np.ndarray):
        self.eng, self.m = eng, len(src)
        self.src = eng.upload(np.asarray(src, dtype=np.float64))
        self.tgt = eng.upload(np.asarray(tgt, dtype=np.float64))
        self.tn = eng.upload(np.asarray(tgt_normals, dtype=np.float64))
        self.grid = DeviceGrid(eng, self.tgt, len(tgt), 8)  # cells for 8 neighbours: the source lies off the target (measured best of 1..16)
        self.idx = eng.empty((self.m, 1), torch.int32)
        self.d2 = eng.empty((self.m, 1), torch.float64)

    def transform(self, t: np.ndarray) -> None:
        e = self.eng
        tt = (C.c_double * 16)(*[float(v) for v in np.asarray(t, dtype=np.float64).reshape(16)])
        e._check(e.lib.pc_transform(e.ctx, _ptr(self.src), self.m, tt, e.stream()), "pc_transform")

    def correspond(self) -> None:
        e = self.eng
        e._check(e.lib.pc_knn(e.ctx, self.grid.h, _ptr(self.src), _lib.PC_F64, self.m, 1, -1.0, _ptr(self.idx),
                              _ptr(self.d2), _lib.PC_F64, None, e.stream()), "pc_knn")

    def terms(self, max_dist: float) -> np.ndarray:
        e = self.eng
        out = (C.c_double * 29)()
        e._check(e.lib.pc_icp_point_to_plane_terms(e.ctx, _ptr(self.src), self.m, _ptr(self.tgt), _lib.PC_F64,
                                                   _ptr(self.tn), _ptr(self.idx), _ptr(self.d2), _lib.PC_F64,
                                                   float(max_dist) ** 2, out, e.stream()), "pc_icp_point_to_plane_terms")
        return np.array(out[:], dtype=np.float64)

    def correspondence_set(self, max_dist: float) -> np.ndarray:
        idx, d2 = self.idx.cpu().numpy()[:, 0], self.d2.cpu().numpy()[:, 0]
        rows = np.nonzero((idx >= 0) & (d2 < float(max_dist) ** 2))[0]
        return np.stack([rows.astype(np.int32), idx[rows]], axis=1)


class HipEngine:
    name = "hip"

    def __init__(self, device: Optional[int] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("[pc_amd] no AMD GPU visible to PyTorch-ROCm; the neighbourhood stage has no CPU path")
        if device is None:
            device = int(os.environ.get("PC_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            device = device % torch.cuda.device_count()
        self.device = torch.device("cuda", device)
        ctx = C.c_void_p()
        rc = self.lib.pc_create(device, C.byref(ctx))
        if rc != 0:
            raise RuntimeError(f"[pc_amd] pc_create({device}) failed: {self.lib.pc_status_string(rc).decode()}")
        self.ctx = ctx

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.pc_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc, what):
        _lib.check(self.lib, self.ctx, rc, what)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def upload(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.require(a, requirements=["C", "W"])).to(self.device)  # (read-only views are copied)

    device_resident = True  # results of the *_dev methods stay in HBM as torch tensors

    @staticmethod
    def to_host(a) -> np.ndarray:
        """Host copy of an engine array (float arrays come back as float64, exactly)."""
        if isinstance(a, torch.Tensor):
            a = a.cpu().numpy()
        return a.astype(np.float64) if a.dtype == np.float32 else a

    def dev_points(self, pts, cache: dict) -> torch.Tensor:
        """Device copy of a cloud's coordinates (``pts``: a host array, or a device array, which is used as
        it is).  fp32 storage when every coordinate of a host array is exactly fp32-representable (then
        nothing is lost), fp64 otherwise."""
        t = cache.get("xyz")
        if t is None:
            if isinstance(pts, torch.Tensor):
                t = pts
            else:
                src = np.ascontiguousarray(pts, dtype=np.float64)
                p32 = np.empty(src.shape, dtype=np.float32)
                exact = self.lib.pc_host_to_f32_exact(src.ctypes.data_as(C.c_void_p), src.size,
                                                      p32.ctypes.data_as(C.c_void_p))
                t = self.upload(p32 if exact else src)
            cache["xyz"] = t
        return t

    def dev_f64(self, a) -> torch.Tensor:
        """An (N, 3) attribute as a float64 device array (uploaded, or widened, when it is not one yet)."""
        if isinstance(a, torch.Tensor):
            return a if a.dtype == torch.float64 else a.to(torch.float64)
        return self.upload(np.asarray(a, dtype=np.float64))

    @staticmethod
    def _dt(t: torch.Tensor) -> int:
        return _lib.PC_F32 if t.dtype == torch.float32 else _lib.PC_F64

    def grid(self, pts: np.ndarray, cache: dict, k_hint: int) -> DeviceGrid:
        key = ("grid", 1 if k_hint <= 4 else (2 if k_hint <= 16 else 3))
        g = cache.get(key)
        if g is None:
            g = DeviceGrid(self, self.dev_points(pts, cache), len(pts), k_hint)
            cache[key] = g
        return g

    # -- engine interface ----------------------------------------------------------------------
    def bbox(self, pts, cache):
        b = cache.get("bbox")
        if b is None:
            t = self.dev_points(pts, cache)
            mn = (C.c_double * 3)()
            mx = (C.c_double * 3)()
            self._check(self.lib.pc_bbox(self.ctx, _ptr(t), self._dt(t), len(pts), mn, mx, self.stream()), "pc_bbox")
            b = (np.array(mn[:], dtype=np.float64), np.array(mx[:], dtype=np.float64))
            cache["bbox"] = b
        return b

    def center(self, pts, cache):
        t = self.dev_points(pts, cache)
        out = (C.c_double * 3)()
        self._check(self.lib.pc_center(self.ctx, _ptr(t), self._dt(t), len(pts), out, self.stream()), "pc_center")
        return np.array(out[:], dtype=np.float64)

    def affine_dev(self, pts, cache, t, s, c) -> torch.Tensor:
        """The edited coordinates as a new float64 device array (the argument is left alone)."""
        d = self.dev_f64(pts)
        if d is pts or d is cache.get("xyz"):
            d = d.clone()
        tt = None if t is None else (C.c_double * 3)(*[float(v) for v in t])
        cc = (C.c_double * 3)(*[float(v) for v in (c if c is not None else (0.0, 0.0, 0.0))])
        self._check(self.lib.pc_affine(self.ctx, _ptr(d), len(pts), tt, 0 if s is None else 1,
                                       0.0 if s is None else float(s), cc, self.stream()), "pc_affine")
        return d

    def affine(self, pts, cache, t, s, c):
        return self.affine_dev(pts, cache, t, s, c).cpu().numpy()

    def voxel_downsample_dev(self, pts, nrm, col, vs, cache):
        n = len(pts)
        t = self.dev_points(pts, cache)
        dn = None if nrm is None else self.dev_f64(nrm)
        dc = None if col is None else self.dev_f64(col)
        op = self.empty((n, 3), torch.float64)
        on = None if nrm is None else self.empty((n, 3), torch.float64)
        oc = None if col is None else self.empty((n, 3), torch.float64)
        m = C.c_int64(0)
        self._check(self.lib.pc_voxel_downsample(self.ctx, _ptr(t), self._dt(t), _ptr(dn), _ptr(dc), n, float(vs),
                                                 _ptr(op), _ptr(on), _ptr(oc), None, None, C.byref(m), self.stream()),
                    "pc_voxel_downsample")
        m = m.value
        # (own storage of m rows each: the n-row output buffers go back to the allocator)
        return (op[:m].clone(), None if on is None else on[:m].clone(), None if oc is None else oc[:m].clone())

    def voxel_layers(self, pts, cache, vs, box_min, box_max, layer_lo, layer_hi):
        """The voxels of the z layers ``layer_lo <= iz < layer_hi`` of the voxel grid over the box of the WHOLE cloud:
        ``(centroids, iz)`` on the device, the same rows and order as that run of ``voxel_downsample``'s output."""
        n = len(pts)
        t = self.dev_points(pts, cache)
        op = self.empty((max(n, 1), 3), torch.float64)
        ijk = self.empty((max(n, 1), 3), torch.int32)
        m = C.c_int64(0)
        mn = (C.c_double * 3)(*[float(v) for v in box_min])
        mx = (C.c_double * 3)(*[float(v) for v in box_max])
        self._check(self.lib.pc_voxel_downsample_layers(self.ctx, _ptr(t), self._dt(t), None, None, n, float(vs), mn, mx,
                                                        int(layer_lo), int(layer_hi), _ptr(op), None, None, None, _ptr(ijk),
                                                        C.byref(m), self.stream()), "pc_voxel_downsample_layers")
        m = m.value
        return op[:m].clone(), ijk[:m, 2].clone()

    @staticmethod
    def concat_rows(parts):
        return torch.cat(list(parts), dim=0)

    @staticmethod
    def sorted_positions(sorted_dev: torch.Tensor, values) -> list:
        """First position of every value in an ascending device array (only the answers visit the host)."""
        v = torch.tensor([int(x) for x in values], dtype=sorted_dev.dtype, device=sorted_dev.device)
        return [int(x) for x in torch.searchsorted(sorted_dev, v).cpu().tolist()]

    def voxel_downsample(self, pts, nrm, col, vs, cache):
        p, n, c = self.voxel_downsample_dev(pts, nrm, col, vs, cache)
        return p.cpu().numpy(), None if n is None else n.cpu().numpy(), None if c is None else c.cpu().numpy()

    def voxel_assign(self, pts, vs, cache=None):
        cache = {} if cache is None else cache
        n = len(pts)
        t = self.dev_points(pts, cache)
        ijk = self.empty((n, 3), torch.int32)
        dims = (C.c_int32 * 3)()
        self._check(self.lib.pc_voxel_assign(self.ctx, _ptr(t), self._dt(t), n, float(vs), _ptr(ijk), dims,
                                             self.stream()), "pc_voxel_assign")
        return ijk.cpu().numpy(), np.array(dims[:], dtype=np.int32)

    def voxel_downsample_info(self, pts, vs, cache=None):
        cache = {} if cache is None else cache
        n = len(pts)
        t = self.dev_points(pts, cache)
        op = self.empty((n, 3), torch.float64)
        cnt = self.empty((n,), torch.int32)
        ijk = self.empty((n, 3), torch.int32)
        m = C.c_int64(0)
        self._check(self.lib.pc_voxel_downsample(self.ctx, _ptr(t), self._dt(t), None, None, n, float(vs), _ptr(op),
                                                 None, None, _ptr(cnt), _ptr(ijk), C.byref(m), self.stream()),
                    "pc_voxel_downsample")
        m = m.value
        return op[:m].cpu().numpy(), cnt[:m].cpu().numpy(), ijk[:m].cpu().numpy()

    def knn(self, pts, q, k, radius, cache, d2_dtype=torch.float64):
        g = self.grid(pts, cache, k)
        nq = len(pts) if q is None else len(q)
        idx = self.empty((nq, k), torch.int32)
        d2 = self.empty((nq, k), d2_dtype)
        cnt = self.empty((nq,), torch.int32)
        dq = None
        qdt = _lib.PC_F64
        if q is not None:
            q32 = q.astype(np.float32)
            dq = self.upload(q32 if np.array_equal(q32.astype(np.float64), q) else q)
            qdt = self._dt(dq)
        self._check(self.lib.pc_knn(self.ctx, g.h, _ptr(dq), qdt, nq, int(k), float(radius), _ptr(idx), _ptr(d2),
                                    self._dt(d2), _ptr(cnt), self.stream()), "pc_knn")
        return idx.cpu().numpy(), d2.cpu().numpy(), cnt.cpu().numpy()

    def knn_fused(self, pts, k, radius, cache, want_rows=True, want_avg=True, want_normals=True, prior=None):
        """One search, all consumers: (idx, d2, cnt, avg, normals) as numpy (None where not asked)."""
        g = self.grid(pts, cache, k)
        n = len(pts)
        idx = self.empty((n, k), torch.int32) if want_rows else None
        d2 = self.empty((n, k), torch.float64) if want_rows else None
        cnt = self.empty((n,), torch.int32) if want_rows else None
        avg = self.empty((n,), torch.float64) if want_avg else None
        nrm = self.empty((n, 3), torch.float64) if want_normals else None
        dp = None if prior is None else self.upload(np.asarray(prior, dtype=np.float64))
        self._check(self.lib.pc_knn_fused(self.ctx, g.h, 0, n, int(k), float(radius), _ptr(idx), _ptr(d2),
                                          _lib.PC_F64, _ptr(cnt), _ptr(avg), _ptr(nrm), _ptr(dp), self.stream()),
                    "pc_knn_fused")
        f = lambda t: None if t is None else t.cpu().numpy()
        return f(idx), f(d2), f(cnt), f(avg), f(nrm)

    def sor_avg_dev(self, pts, k, cache) -> torch.Tensor:
        key = ("sor_avg", k)
        avg = cache.get(key)
        if avg is None:
            g = self.grid(pts, cache, k)
            avg = self.empty((len(pts),), torch.float64)
            self._check(self.lib.pc_sor_avg(self.ctx, g.h, 0, len(pts), int(k), _ptr(avg), self.stream()),
                        "pc_sor_avg")
            cache[key] = avg
        return avg

    def exact_sums(self, avg: torch.Tensor, owned, what: int, mean: float) -> torch.Tensor:
        """The ten integer words of ``pc_sor_exact_sums`` over the rows of ``avg`` (those with ``owned`` >= 0 where
        ``owned`` is given): an order-free sum of avg (what = 0) or (avg - mean)^2 (what = 1) and the count."""
        words = torch.zeros((_lib.PC_SOR_EXACT_WORDS,), dtype=torch.int64, device=self.device)
        self._check(self.lib.pc_sor_exact_sums(self.ctx, _ptr(avg), _ptr(owned), avg.numel(), int(what), float(mean),
                                               _ptr(words), self.stream()), "pc_sor_exact_sums")
        return words

    def sor_stats(self, avg: torch.Tensor, std_ratio: float, pts=None, cache=None, k=None):
        """(mean, std, threshold, valid) of the per-point mean distances from exact integer sums: the same bits
        as a multi-GPU job finds for the same cloud, however it is cut (``distributed._filter_core``)."""
        from .distributed import exact_value

        n = avg.numel()
        total, valid, big = exact_value(self.exact_sums(avg, None, 0, 0.0).cpu().numpy())
        if valid == 0:
            return 0.0, 0.0, 0.0, 0
        if not big:
            mean = total / (valid << 128)
            sq, _, big = exact_value(self.exact_sums(avg, None, 1, mean).cpu().numpy())
            sq = sq / (1 << 128)
        if big:  # distances of 2^96 and more: the floating-point tree over the array as it lies
            s = C.c_double(0)
            nv = C.c_int64(0)
            self._check(self.lib.pc_sor_sum(self.ctx, _ptr(avg), n, C.byref(s), C.byref(nv), self.stream()), "pc_sor_sum")
            valid = nv.value
            mean = s.value / valid
            q = C.c_double(0)
            self._check(self.lib.pc_sor_sqsum(self.ctx, _ptr(avg), n, mean, C.byref(q), self.stream()), "pc_sor_sqsum")
            sq = q.value
        with np.errstate(divide="ignore", invalid="ignore"):
            std = float(np.sqrt(np.float64(sq) / np.float64(valid - 1)))
        return mean, std, mean + std_ratio * std, valid

    def sor_dev(self, pts, k, std_ratio, cache):
        """Kept indices of the statistical filter, still on the device (a LazyIndices)."""
        from .geometry import LazyIndices

        avg = self.sor_avg_dev(pts, k, cache)
        mean, std, thr, valid = self.sor_stats(avg, std_ratio, pts, cache, k)
        if valid == 0:
            return LazyIndices(self, self.empty((0,), torch.int32), 0)
        n = avg.numel()
        kept = self.empty((n,), torch.int32)
        nk = C.c_int64(0)
        self._check(self.lib.pc_sor_select(self.ctx, _ptr(avg), n, float(thr), None, _ptr(kept), C.byref(nk),
                                           self.stream()), "pc_sor_select")
        return LazyIndices(self, kept[: nk.value], nk.value)

    def sor(self, pts, k, std_ratio, cache):
        return self.sor_dev(pts, k, std_ratio, cache).host()

    def radius_outlier_dev(self, pts, nb_points, radius, cache):
        """Indices of the points with more than nb_points neighbours inside the radius (self included): capped
        counts from the hybrid search, then an ordered select; nothing visits the host but the count."""
        from .geometry import LazyIndices

        n = len(pts)
        cnt = self.radius_count_dev(pts, radius, cache, cap=int(nb_points) + 1)
        kept = self.empty((n,), torch.int32)
        nk = C.c_int64(0)
        self._check(self.lib.pc_count_select(self.ctx, _ptr(cnt), n, int(nb_points), None, _ptr(kept), C.byref(nk),
                                             self.stream()), "pc_count_select")
        return LazyIndices(self, kept[: nk.value], nk.value)

    def radius_outlier(self, pts, nb_points, radius, cache):
        return self.radius_outlier_dev(pts, nb_points, radius, cache).host()

    def select(self, arrs: dict, idx, m: int, cache: dict, child_cache: dict) -> dict:
        """Rows ``idx`` (a device int32 array, or host indices) of every given (N, 3) attribute, gathered on the
        device.  A cloud held as fp32 (exactly representable coordinates) stays fp32: the child's index is
        built from the gathered fp32 rows."""
        if not isinstance(idx, torch.Tensor):
            # indices a caller supplies are checked like NumPy checks them (the gather kernel reads what it is told to)
            host = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
            rows = next(iter(arrs.values())).shape[0] if arrs else 0
            if host.size and (int(host.min()) < -rows or int(host.max()) >= rows):
                bad = int(host.max()) if int(host.max()) >= rows else int(host.min())
                raise IndexError(f"index {bad} is out of bounds for a cloud of {rows} points")
            host = np.where(host < 0, host + rows, host)  # (negative indices count from the end, as on the host path)
            idx = torch.from_numpy(host.astype(np.int32)).to(self.device)
        out = {}
        for name, a in arrs.items():
            xyz = cache.get("xyz") if name == "points" else None
            if xyz is not None and xyz.dtype == torch.float32:
                dst = self.empty((m, 3), torch.float32)
                self._check(self.lib.pc_gather_rows_f32(self.ctx, _ptr(xyz), _ptr(idx), m, _ptr(dst), self.stream()),
                            "pc_gather_rows_f32")
                child_cache["xyz"] = dst
            else:
                src = xyz if xyz is not None else self.dev_f64(a)
                dst = self.empty((m, 3), torch.float64)
                self._check(self.lib.pc_gather_rows(self.ctx, _ptr(src), _ptr(idx), m, _ptr(dst), self.stream()),
                            "pc_gather_rows")
            out[name] = dst
        return out

    # -- one rank of a multi-GPU job: queries = sorted positions [b, e) of the replicated index -------------
    def shard_search(self, pts, cache, k, radius, b, e, want_rows=False, want_avg=False, want_normals=False,
                     d2_dtype=torch.float64):
        """One fused search over the slice; every output has e - b rows (row s - b for sorted position s) and
        stays on the device.  ``owned``: the original index of every row."""
        g = self.grid(pts, cache, k)
        m = e - b
        out = {"owned": self.empty((max(m, 1),), torch.int32)[:m]}
        if want_rows:
            out["idx"] = self.empty((m, k), torch.int32)
            out["d2"] = self.empty((m, k), d2_dtype)
            out["cnt"] = self.empty((m,), torch.int32)
        if want_avg:
            out["avg"] = self.empty((m,), torch.float64)
        if want_normals:
            out["normals"] = self.empty((m, 3), torch.float64)
        self._check(self.lib.pc_knn_fused_compact(self.ctx, g.h, b, e, int(k), float(radius), _ptr(out.get("idx")),
                                                  _ptr(out.get("d2")), self._dt(out["d2"]) if want_rows else _lib.PC_F64,
                                                  _ptr(out.get("cnt")), _ptr(out.get("avg")), _ptr(out.get("normals")),
                                                  _ptr(out["owned"]), self.stream()), "pc_knn_fused_compact")
        return out

    # -- one rank of a multi-GPU job with an index of its own slab of the cloud -------------------------------
    def axes_histogram(self, pts, cache, lo, hi, nbins: int) -> np.ndarray:
        """(3, nbins) counts of the points per bin of [lo[a], hi[a]] along every axis a, from one pass."""
        t = self.dev_points(pts, cache)
        out = np.zeros((3, nbins), dtype=np.uint32)
        l, h = (C.c_double * 3)(*[float(v) for v in lo]), (C.c_double * 3)(*[float(v) for v in hi])
        self._check(self.lib.pc_axis_histogram(self.ctx, _ptr(t), self._dt(t), len(pts), -1, l, h, int(nbins),
                                               out.ctypes.data_as(C.c_void_p), self.stream()), "pc_axis_histogram")
        return out

    def slab_search(self, pts, cache, k, radius, own_lo, own_hi, halo, want_avg=False, want_normals=False,
                    d2_dtype=torch.float64, avg_full=None, stage_ms: Optional[dict] = None, open_faces=None):
        """The rank's queries (own_lo <= z < own_hi) answered from an index of the points with z in
        [own_lo - halo, own_hi + halo) only.  Rows as in ``shard_search`` (always with idx / d2 / cnt), indices
        those of the whole cloud; rows with ``owned`` < 0 belong to a neighbouring rank and are to be ignored.
        ``n_uncertain``: own answers the slab cannot vouch for (see pc_slab_finish).  ``open_faces`` = (low, high): whether
        the cloud goes on beyond the slab's low / high face, for callers whose ``pts`` holds only the slab itself.  ``avg_full`` (n doubles or
        more): receives the own rows' mean distances at their index in the whole cloud.  ``stage_ms``: a dict
        that receives the time of each stage (synchronises after every stage: measurements only)."""
        import time

        clock = [time.perf_counter()]

        def lap(name):
            if stage_ms is not None:
                torch.cuda.synchronize(self.device)
                now = time.perf_counter()
                stage_ms[name] = stage_ms.get(name, 0.0) + (now - clock[0]) * 1e3
                clock[0] = now

        t = self.dev_points(pts, cache)
        n = len(pts)
        mn, mx = self.bbox(pts, cache)
        lo, hi = own_lo - halo, own_hi + halo
        key = ("slab", float(lo), float(hi), 1 if k <= 4 else (2 if k <= 16 else 3))
        sub = cache.get(key)
        if sub is None:
            gidx = self.empty((max(n, 1),), torch.int32)
            m = C.c_int64(0)
            self._check(self.lib.pc_select_slab(self.ctx, _ptr(t), self._dt(t), n, 2, float(lo), float(hi), _ptr(gidx),
                                                C.byref(m), self.stream()), "pc_select_slab")
            m = m.value
            lap("select")
            gidx = gidx[:m].clone()  # (the n-entry buffer goes back)
            xyz = self.empty((max(m, 1), 3), t.dtype)[:m]
            gather = self.lib.pc_gather_rows_f32 if t.dtype == torch.float32 else self.lib.pc_gather_rows
            self._check(gather(self.ctx, _ptr(t), _ptr(gidx), m, _ptr(xyz), self.stream()), "pc_gather_rows")
            lap("gather")
            g = DeviceGrid(self, xyz, m, k) if m > 0 else None
            if g is not None:  # results in indices of the whole cloud from here on
                self._check(self.lib.pc_grid_relabel(self.ctx, g.h, _ptr(gidx), n, self.stream()), "pc_grid_relabel")
            sub = (gidx, xyz, g)
            cache[key] = sub
            lap("index")
        gidx, xyz, g = sub
        m = gidx.numel()
        b, e = C.c_int64(0), C.c_int64(0)
        if g is not None:
            self._check(self.lib.pc_grid_axis_range(self.ctx, g.h, float(own_lo), float(own_hi), C.byref(b), C.byref(e),
                                                    self.stream()), "pc_grid_axis_range")
        b, e = b.value, e.value
        rows = e - b
        out = {"owned": self.empty((max(rows, 1),), torch.int32)[:rows], "idx": self.empty((rows, k), torch.int32),
               "d2": self.empty((rows, k), d2_dtype), "cnt": self.empty((rows,), torch.int32), "sub_n": m}
        if want_avg or avg_full is not None:
            out["avg"] = self.empty((rows,), torch.float64)
        if want_normals:
            out["normals"] = self.empty((rows, 3), torch.float64)
        n_own, n_unc = C.c_int64(0), C.c_int64(0)
        unc_rows = self.empty((max(rows, 1),), torch.int32)
        lap("range+outputs")
        if rows > 0:
            self._check(self.lib.pc_knn_fused_compact(self.ctx, g.h, b, e, int(k), float(radius), _ptr(out["idx"]),
                                                      _ptr(out["d2"]), self._dt(out["d2"]), _ptr(out["cnt"]),
                                                      _ptr(out.get("avg")), _ptr(out.get("normals")), _ptr(out["owned"]),
                                                      self.stream()), "pc_knn_fused_compact")
            lap("search")
            # faces of the subset beyond which the cloud goes on
            open_lo = float(lo) if lo > mn[2] else -np.inf
            open_hi = float(hi) if hi <= mx[2] else np.inf
            if open_faces is not None:  # (``pts`` is only a part of the cloud: the caller knows where it goes on)
                open_lo = float(lo) if open_faces[0] else -np.inf
                open_hi = float(hi) if open_faces[1] else np.inf
            self._check(self.lib.pc_slab_finish(self.ctx, _ptr(t), self._dt(t), None, rows, int(k), float(radius),
                                                _ptr(out["idx"]), _ptr(out["d2"]), self._dt(out["d2"]), _ptr(out["cnt"]),
                                                _ptr(out["owned"]), float(own_lo), float(own_hi), open_lo, open_hi,
                                                _ptr(out.get("avg")), _ptr(avg_full), _ptr(unc_rows), C.byref(n_own),
                                                C.byref(n_unc), self.stream()), "pc_slab_finish")
        lap("finish")
        out["n_own"], out["n_uncertain"] = n_own.value, n_unc.value
        out["uncertain_rows"] = unc_rows[: n_unc.value].to(torch.int64)  # rows whose answers are to be repeated
        out["slab_key"] = key
        return out

    def slab_requery(self, pts, cache, slab_key, qidx: torch.Tensor, k: int):
        """The points ``qidx`` (indices of the whole cloud) as queries against this rank's slab index: rows in
        indices of the whole cloud, fp64 distances, counts."""
        t = self.dev_points(pts, cache)
        gidx, xyz, g = cache[slab_key]
        m = qidx.numel()
        idx = torch.full((m, k), -1, dtype=torch.int32, device=self.device)
        d2 = torch.full((m, k), float("inf"), dtype=torch.float64, device=self.device)
        cnt = torch.zeros((m,), dtype=torch.int32, device=self.device)
        if g is None or m == 0:
            return idx, d2, cnt
        q = self.empty((m, 3), t.dtype)
        gather = self.lib.pc_gather_rows_f32 if t.dtype == torch.float32 else self.lib.pc_gather_rows
        self._check(gather(self.ctx, _ptr(t), _ptr(qidx), m, _ptr(q), self.stream()), "pc_gather_rows")
        self._check(self.lib.pc_knn(self.ctx, g.h, _ptr(q), self._dt(q), m, int(k), -1.0, _ptr(idx), _ptr(d2), _lib.PC_F64,
                                    _ptr(cnt), self.stream()), "pc_knn")
        return idx, d2, cnt

    def merge_rows(self, idx_all: torch.Tensor, d2_all: torch.Tensor, cnt_all: torch.Tensor):
        world, m, k = idx_all.shape
        idx = self.empty((m, k), torch.int32)
        d2 = self.empty((m, k), torch.float64)
        cnt = self.empty((m,), torch.int32)
        self._check(self.lib.pc_merge_rows(self.ctx, world, m, k, _ptr(idx_all), _ptr(d2_all), _ptr(cnt_all), _ptr(idx),
                                           _ptr(d2), _ptr(cnt), self.stream()), "pc_merge_rows")
        return idx, d2, cnt

    def rows_consumers(self, pts, cache, qidx: torch.Tensor, idx: torch.Tensor, cnt: torch.Tensor, want_normals: bool):
        """Mean neighbour distance (and normal) of the listed points from their finished rows."""
        t = self.dev_points(pts, cache)
        m, k = idx.shape
        avg = self.empty((m,), torch.float64)
        nrm = self.empty((m, 3), torch.float64) if want_normals else None
        self._check(self.lib.pc_rows_consumers(self.ctx, _ptr(t), self._dt(t), _ptr(qidx), m, k, _ptr(idx), _ptr(cnt),
                                               _ptr(avg), _ptr(nrm), self.stream()), "pc_rows_consumers")
        return avg, nrm

    def zeros(self, shape, kind: str) -> torch.Tensor:
        return torch.zeros(shape, dtype=torch.int32 if kind == "i32" else torch.float64, device=self.device)

    @staticmethod
    def cast_like(x: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        return x.to(like.dtype)

    def zeros_f64(self, m: int) -> torch.Tensor:
        return torch.zeros((m,), dtype=torch.float64, device=self.device)

    def sor_from_avg(self, avg: torch.Tensor, n: int, std_ratio: float):
        """Kept indices and (mean, std, threshold, valid) from the n mean distances as they lie (a fixed tree over
        the array: the same bits on every rank)."""
        from .geometry import LazyIndices

        avg = avg[:n]
        mean, std, thr, valid = self.sor_stats(avg, std_ratio)
        if valid == 0:
            return self.no_indices(), (0.0, 0.0, 0.0, 0)
        kept = self.empty((max(n, 1),), torch.int32)
        nk = C.c_int64(0)
        self._check(self.lib.pc_sor_select(self.ctx, _ptr(avg), n, float(thr), None, _ptr(kept), C.byref(nk),
                                           self.stream()), "pc_sor_select")
        return LazyIndices(self, kept[: nk.value], nk.value), (mean, std, thr, valid)

    def mask_scatter(self, avg, owned, thr, n) -> torch.Tensor:
        mask = torch.zeros((int(self.lib.pc_mask_bytes(n)),), dtype=torch.uint8, device=self.device)  # one bit per point, set = removed
        self._check(self.lib.pc_sor_mask_scatter(self.ctx, _ptr(avg), _ptr(owned), avg.numel(), float(thr), _ptr(mask),
                                                 self.stream()), "pc_sor_mask_scatter")
        return mask

    def mask_select(self, mask: torch.Tensor, n: int):
        from .geometry import LazyIndices

        kept = self.empty((max(n, 1),), torch.int32)
        nk = C.c_int64(0)
        self._check(self.lib.pc_mask_select(self.ctx, _ptr(mask), n, _ptr(kept), C.byref(nk), self.stream()),
                    "pc_mask_select")
        return LazyIndices(self, kept[: nk.value], nk.value)

    def no_indices(self):
        from .geometry import LazyIndices

        return LazyIndices(self, self.empty((0,), torch.int32), 0)

    def rows_scatter(self, rows: torch.Tensor, owned: torch.Tensor, n: int) -> torch.Tensor:
        full = torch.zeros((n, 3), dtype=torch.float64, device=self.device)
        self._check(self.lib.pc_scatter_rows(self.ctx, _ptr(rows), _ptr(owned), owned.numel(), _ptr(full), self.stream()),
                    "pc_scatter_rows")
        return full

    def normals_dev(self, pts, k, radius, prior, cache) -> torch.Tensor:
        g = self.grid(pts, cache, k)
        n = len(pts)
        out = self.empty((n, 3), torch.float64)
        dp = None if prior is None else self.dev_f64(prior)
        self._check(self.lib.pc_normals(self.ctx, g.h, 0, n, int(k), float(radius), _ptr(dp), _ptr(out),
                                        self.stream()), "pc_normals")
        return out

    def normals(self, pts, k, radius, prior, cache):
        return self.normals_dev(pts, k, radius, prior, cache).cpu().numpy()

    def radius_count_dev(self, pts, radius, cache, cap=None) -> torch.Tensor:
        """Neighbours with d2 < radius^2 (self included).  With `cap` the result is min(count, cap): that is
        all remove_radius_outlier needs (count > nb_points), and it is the hybrid search
        kNN(k = cap, radius) on the tile kernel instead of a full ball count."""
        n = len(pts)
        cnt = self.empty((n,), torch.int32)
        if cap is not None and 1 <= int(cap) <= _lib.PC_MAX_K:
            g = self.grid(pts, cache, int(cap))
            self._check(self.lib.pc_knn(self.ctx, g.h, None, _lib.PC_F64, n, int(cap), float(radius), None, None,
                                        _lib.PC_F32, _ptr(cnt), self.stream()), "pc_knn")
            return cnt
        g = self.grid(pts, cache, 8)
        if float(radius) > 2.0 * g.info()["cell"]:
            # a ball spanning many cells: one ring of radius-sized cells beats many rings of small ones
            g = DeviceGrid(self, self.dev_points(pts, cache), n, 0, cell=float(radius))
        self._check(self.lib.pc_radius_count(self.ctx, g.h, None, _lib.PC_F64, 0, 0, n, float(radius), _ptr(cnt),
                                             self.stream()), "pc_radius_count")
        return cnt

    def radius_count(self, pts, radius, cache, cap=None):
        out = self.radius_count_dev(pts, radius, cache, cap).cpu().numpy()
        return out if cap is None else np.minimum(out, int(cap))

    def orient_normals_dev(self, pts, nrm, k, cache):
        """Normals turned to agree along the spanning tree of the k-neighbour graph; the neighbour rows
        never leave the device.  Returns (normals as a new device array, number of connected pieces)."""
        n = len(pts)
        g = self.grid(pts, cache, k)
        t = self.dev_points(pts, cache)
        idx = self.empty((n, k), torch.int32)
        cnt = self.empty((n,), torch.int32)
        self._check(self.lib.pc_knn(self.ctx, g.h, None, _lib.PC_F64, n, int(k), -1.0, _ptr(idx), None, _lib.PC_F32,
                                    _ptr(cnt), self.stream()), "pc_knn")
        dn = self.dev_f64(nrm)
        if dn is nrm:
            dn = dn.clone()
        pieces = C.c_int64(0)
        self._check(self.lib.pc_orient_normals(self.ctx, _ptr(t), self._dt(t), _ptr(dn), _ptr(idx), _ptr(cnt), n, int(k),
                                               C.byref(pieces), self.stream()), "pc_orient_normals")
        return dn, int(pieces.value)

    def orient_normals(self, pts, nrm, k, cache):
        dn, pieces = self.orient_normals_dev(pts, nrm, k, cache)
        return dn.cpu().numpy(), pieces

    def icp_session(self, src, tgt, tgt_normals) -> HipIcpSession:
        return HipIcpSession(self, src, tgt, tgt_normals)

    def nn_distance(self, pts, cache):
        g = self.grid(pts, cache, 2)
        n = len(pts)
        out = self.empty((n,), torch.float64)
        self._check(self.lib.pc_nn_distance(self.ctx, g.h, 0, n, _ptr(out), self.stream()), "pc_nn_distance")
        return out.cpu().numpy()
