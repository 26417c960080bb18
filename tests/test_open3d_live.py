"""Side-by-side runs against a real Open3D, for a machine that happens to have the wheel (it is not in this
image and is never installed by the build: every test here skips cleanly without it).  This is the only
route by which parity with the reference's own arithmetic (open3d==0.19.0, requirements.txt:39) can become
pinned; the calls are the reference's call sites: app/Core/pointcloud_preprocess.py:96, 109-114,
app/Infrastructure/filtering_adapter.py:49, 91, 94."""
import numpy as np
import pytest

import datasets as D

pytestmark = [pytest.mark.open3d, pytest.mark.gpu]


@pytest.fixture(scope="module")
def o3d():
    return pytest.importorskip("open3d")


def _cloud(o3d, p):
    c = o3d.geometry.PointCloud()
    c.points = o3d.utility.Vector3dVector(p)
    return c


def _angle(a, b):
    c = np.abs((a * b).sum(1)) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    return np.arccos(np.clip(c, -1, 1))


def test_voxel_down_sample(o3d, hip):
    p = D.bumpy_sphere(20000, 31)
    ref = np.asarray(_cloud(o3d, p).voxel_down_sample(0.01).points)
    got, _, _ = hip.voxel_downsample(p, None, None, 0.01, {})
    key = lambda a: a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]  # Open3D's output order is its hash map's
    assert got.shape == ref.shape
    np.testing.assert_allclose(key(got), key(ref), rtol=1e-12, atol=1e-15)


def test_statistical_outlier(o3d, hip):
    p = np.concatenate([D.bumpy_sphere(15000, 32), np.random.default_rng(1).random((150, 3)) * 0.4])
    _, ref = _cloud(o3d, p).remove_statistical_outlier(30, 3.0)
    assert np.array_equal(np.asarray(hip.sor(p, 30, 3.0, {})), np.asarray(ref))


def test_radius_outlier_and_nn_distance(o3d, hip):
    p = D.gaussian_clusters(12000, 33)
    c = _cloud(o3d, p)
    d = np.asarray(c.compute_nearest_neighbor_distance())
    np.testing.assert_allclose(hip.nn_distance(p, {}), d, rtol=1e-13)
    r = 5.0 * float(np.median(d))
    _, ref = c.remove_radius_outlier(4, r)
    assert np.array_equal(np.asarray(hip.radius_outlier(p, 4, r, {})), np.asarray(ref))


def test_estimate_normals_hybrid(o3d, hip):
    p = D.bumpy_sphere(15000, 34)
    c = _cloud(o3d, p)
    c.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=0.02, max_nn=30))
    got = hip.normals(p, 30, 0.02, None, {})
    assert _angle(got, np.asarray(c.normals)).max() < 1e-4  # rad, up to sign (north_star's tolerance)
