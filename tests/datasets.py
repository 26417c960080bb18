"""Seeded synthetic clouds shared by the tests, the golden-vector script and bench.py.

All generators return float64 arrays whose values are exactly fp32-representable (the
reference's loaders cast LAS/E57 coordinates to fp32 first, pointcloud_io.py:65-72), unless
``fp32=False``.
"""
from __future__ import annotations

import numpy as np


def _finish(p, fp32=True):
    p = np.ascontiguousarray(p, dtype=np.float64)
    if fp32:
        p = p.astype(np.float32).astype(np.float64)
    return p


def uniform_cube(n, seed, fp32=True):
    rng = np.random.default_rng(seed)
    return _finish(rng.random((n, 3)), fp32)


def gaussian_clusters(n, seed, n_clusters=12, fp32=True):
    rng = np.random.default_rng(seed)
    centers = rng.random((n_clusters, 3)) * 10.0
    sig = 0.05 + rng.random(n_clusters) * 0.5
    which = rng.integers(0, n_clusters, n)
    return _finish(centers[which] + rng.standard_normal((n, 3)) * sig[which, None], fp32)


def bumpy_sphere(n, seed, sigma=0.002, fp32=True):
    """Noisy closed surface standing in for a small scanned object (the 35 k-point case)."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = 1.0 + 0.15 * np.sin(5 * v[:, 0]) * np.cos(4 * v[:, 1]) + 0.1 * np.sin(7 * v[:, 2])
    p = v * r[:, None] * np.array([0.8, 1.0, 0.6]) * 0.07
    p += rng.standard_normal((n, 3)) * sigma * 0.07
    return _finish(p + np.array([0.02, 0.11, -0.01]), fp32)


def planar_patches(n, seed, extent=200.0, n_patches=40, frac_uniform=0.3, fp32=True):
    """Terrestrial-scan-like: random planar patches plus uniform clutter."""
    rng = np.random.default_rng(seed)
    nu = int(n * frac_uniform)
    npl = n - nu
    which = rng.integers(0, n_patches, npl)
    origin = rng.random((n_patches, 3)) * extent
    a = rng.standard_normal((n_patches, 3))
    b = rng.standard_normal((n_patches, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b -= a * (a * b).sum(1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    size = 5.0 + rng.random(n_patches) * 25.0
    u = (rng.random(npl) - 0.5) * size[which]
    v = (rng.random(npl) - 0.5) * size[which]
    pl = origin[which] + a[which] * u[:, None] + b[which] * v[:, None] + rng.standard_normal((npl, 3)) * 0.005
    un = rng.random((nu, 3)) * extent
    p = np.concatenate([pl, un], axis=0)
    return _finish(p[rng.permutation(n)], fp32)


def lidar_frame(seed, rings=64, azimuths=1875, fp32=True):
    """One 120 k-point spinning-LiDAR-like frame: rays from (0,0,1.8) against a ground plane
    and random boxes, +-80 m, range noise 0.02 m."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, rings))
    az = np.linspace(0, 2 * np.pi, azimuths, endpoint=False)
    E, Az = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(Az), np.cos(E) * np.sin(Az), np.sin(E)], -1).reshape(-1, 3)
    o = np.array([0.0, 0.0, 1.8])
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d[:, 2] < -1e-6, -o[2] / d[:, 2], np.inf)
    t = np.minimum(t, 110.0)
    nb = 30
    bc = (rng.random((nb, 2)) - 0.5) * 140.0
    bs = 1.0 + rng.random((nb, 2)) * 6.0
    bh = 1.0 + rng.random(nb) * 4.0
    for i in range(nb):
        lo = np.array([bc[i, 0] - bs[i, 0], bc[i, 1] - bs[i, 1], 0.0])
        hi = np.array([bc[i, 0] + bs[i, 0], bc[i, 1] + bs[i, 1], bh[i]])
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - o) / d
            t2 = (hi - o) / d
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)
    t = t + rng.standard_normal(t.shape) * 0.02
    p = o + d * t[:, None]
    p = np.clip(p, [-80, -80, -3], [80, 80, 5])
    return _finish(p, fp32)


def lattice(m=8):
    g = np.arange(m, dtype=np.float64)
    return np.ascontiguousarray(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3))


def torus(n, seed, R=1.0, r=0.3, sigma=0.002, fp32=True):
    rng = np.random.default_rng(seed)
    u = rng.random(n) * 2 * np.pi
    v = rng.random(n) * 2 * np.pi
    p = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], -1)
    return _finish(p + rng.standard_normal((n, 3)) * sigma, fp32)
