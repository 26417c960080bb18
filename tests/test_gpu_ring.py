"""Routes through a multi-level index that only big clouds take by themselves, forced on small ones: the
lane-per-query ring pass over what the ladder leaves open (csrc/pc_knn_ring.hip) against the wave-per-query kernel
that answers those queries without it, and the ladder of a cloud that comes with its finest level only (levels made
as the flags of the postponed queries ask for them, sparse tiles postponed tile by tile, full tiles first) against
the ladder over levels built with the index: the same rows, distances and normals, bit for bit, and equal to the
CPU oracle."""
import numpy as np
import pytest

import datasets as D

pytestmark = pytest.mark.gpu


def _clouds():
    big = D.planar_patches(600_000, 71)
    return {
        "planes_and_clutter_k32": (D.planar_patches(400_000, 70), 32),
        "planes_and_clutter_k16_fp64": (np.asarray(big, dtype=np.float64) + 1e-9, 16),  # not fp32-representable: fp64 records
        "clusters_k8": (D.gaussian_clusters(300_000, 72), 8),
        "lidar_k30": (D.lidar_frame(73), 30),
        "sphere_k20": (D.bumpy_sphere(200_000, 74), 20),
    }


@pytest.fixture(scope="module")
def engines():
    """(ring pass from one open query on, no ring pass): the switches are read when a context is created."""
    import os
    from point_cloud_processing_amd.hip_engine import HipEngine

    old = {k: os.environ.get(k) for k in ("PC_AMD_RING_MIN", "PC_AMD_NO_RING")}
    try:
        os.environ["PC_AMD_RING_MIN"] = "1"
        os.environ.pop("PC_AMD_NO_RING", None)
        ring = HipEngine(0)
        os.environ.pop("PC_AMD_RING_MIN")
        os.environ["PC_AMD_NO_RING"] = "1"
        plain = HipEngine(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ring, plain


@pytest.fixture(scope="module")
def lazy_engine():
    """Big-cloud routes from the first point on: finest level only, full tiles first, tile-wise postponement, rings."""
    import os
    from point_cloud_processing_amd.hip_engine import HipEngine

    forced = {"PC_AMD_LAZY_MIN": "1", "PC_AMD_MIXED_MIN": "1", "PC_AMD_RING_MIN": "1"}
    old = {k: os.environ.get(k) for k in forced}
    try:
        os.environ.update(forced)
        eng = HipEngine(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return eng


@pytest.mark.parametrize("name", list(_clouds()))
def test_levels_made_on_demand_equal_levels_built_with_the_index(engines, lazy_engine, name):
    _, plain = engines
    pts, k = _clouds()[name]
    a = lazy_engine.knn(pts, None, k, -1.0, {})
    b = plain.knn(pts, None, k, -1.0, {})
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    fa = lazy_engine.knn_fused(pts, k, -1.0, {})
    fb = plain.knn_fused(pts, k, -1.0, {})
    assert all(np.array_equal(x, y) for x, y in zip(fa, fb))
    # a hybrid search (radius + k) takes the ladder without postponed queries
    r = float(np.sqrt(np.median(b[1][:, min(k, 8) - 1]))) * 1.5
    ha = lazy_engine.knn(pts, None, k, r, {})
    hb = plain.knn(pts, None, k, r, {})
    assert all(np.array_equal(x, y) for x, y in zip(ha, hb))


@pytest.mark.parametrize("name", list(_clouds()))
def test_ring_pass_equals_wave_per_query(engines, name):
    ring, plain = engines
    pts, k = _clouds()[name]
    a = ring.knn(pts, None, k, -1.0, {})
    b = plain.knn(pts, None, k, -1.0, {})
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    fa = ring.knn_fused(pts, k, -1.0, {})
    fb = plain.knn_fused(pts, k, -1.0, {})
    assert all(np.array_equal(x, y) for x, y in zip(fa, fb))


def test_ring_pass_against_the_oracle(engines, oracle):
    ring, _ = engines
    p = D.planar_patches(150_000, 75)
    idx, d2, cnt = ring.knn(p, None, 16, -1.0, {})
    ridx, rd2, rcnt = oracle.Grid(p).knn_self(16)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)
    n = ring.normals(p, 16, -1.0, None, {})
    rn = oracle.normals(p, 16)
    assert np.abs(n - rn).max() < 1e-6  # (the device's eigen-solver rounds a last bit differently: tolerance of the normals tests)
