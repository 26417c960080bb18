"""The CPU oracle against independent third-party code, closed-form known answers and the
committed golden vectors.  (The reference itself holds no tests or fixtures: parity unpinned.)"""
import glob
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

import datasets as D

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def angle_up_to_sign(a, b):
    c = np.abs((a * b).sum(1)) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    return np.arccos(np.clip(c, -1, 1))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    p = g["points"]
    for k in (1, 8, 16):
        idx, d2, cnt = oracle.knn(p, None, k)
        assert np.array_equal(idx, g[f"idx_k{k}"]) and np.array_equal(d2, g[f"d2_k{k}"])
        assert (cnt == k).all()
        bidx, bd2, _ = oracle.knn_brute(p, None, k)
        assert np.array_equal(bidx, idx) and np.array_equal(bd2, d2)
    assert angle_up_to_sign(oracle.normals(p, 16), g["normals_k16"]).max() < 1e-4
    np.testing.assert_allclose(oracle.nn_distance(p), g["nn_dist"], rtol=1e-14)
    assert np.array_equal(oracle.radius_count(p, float(g["radius"])), g["radius_count"])
    np.testing.assert_allclose(oracle.sor_avg(p, 8), g["sor_avg_k8"], rtol=1e-14)
    ijk, _, _ = oracle.voxel_index(p, float(g["voxel_size"]))
    assert np.array_equal(ijk, g["voxel_ijk"])


@pytest.mark.parametrize("gen,seed", [(D.uniform_cube, 1), (D.gaussian_clusters, 2), (D.bumpy_sphere, 3)])
def test_knn_vs_ckdtree_35k(oracle, gen, seed):
    p = gen(35000, seed)
    idx, d2, _ = oracle.knn(p, None, 20)
    dd, ii = cKDTree(p).query(p, 20, workers=-1)
    assert (ii == idx).mean() > 0.9999  # differences only where scipy's own rounding reorders near-ties
    np.testing.assert_allclose(np.sqrt(d2), dd, rtol=1e-12, atol=1e-300)


def test_lattice_ties_ascending_index(oracle):
    p = D.lattice(8)
    idx, d2, _ = oracle.knn(p, None, 19)
    i = (3 * 8 + 4) * 8 + 2  # interior point (3, 4, 2)
    assert idx[i, 0] == i and d2[i, 0] == 0
    face = sorted([i - 64, i + 64, i - 8, i + 8, i - 1, i + 1])
    assert list(idx[i, 1:7]) == face and (d2[i, 1:7] == 1).all()
    edge = sorted([i + a + b for a in (-64, 64) for b in (-8, 8)] + [i + a + b for a in (-64, 64) for b in (-1, 1)]
                  + [i + a + b for a in (-8, 8) for b in (-1, 1)])
    assert list(idx[i, 7:19]) == edge and (d2[i, 7:19] == 2).all()


def test_fewer_points_than_k_and_single_point(oracle):
    p = D.uniform_cube(5, 4)
    idx, d2, cnt = oracle.knn(p, None, 16)
    assert (cnt == 5).all() and (idx[:, 5:] == -1).all() and np.isinf(d2[:, 5:]).all()
    assert list(oracle.nn_distance(p[:1])) == [0.0]


def test_hybrid_is_knn_then_strict_radius_cut(oracle):
    p = D.uniform_cube(5000, 5)
    r = 0.06
    idx, d2, cnt = oracle.knn(p, None, 30, r)
    fidx, fd2, _ = oracle.knn(p, None, 30)
    keep = fd2 < r * r
    assert np.array_equal(cnt, keep.sum(1))
    assert np.array_equal(np.where(keep, fidx, -1), idx)


def test_radius_count_strict(oracle):
    p = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 2.0, 0]])
    assert list(oracle.radius_count(p, 1.0)) == [1, 1, 1]
    assert list(oracle.radius_count(p, 1.0000001)) == [2, 2, 1]


def test_sor_closed_form(oracle):
    # integers 0..9 on a line + one far outlier; k = 3 (self + 2)
    x = np.array(list(range(10)) + [100], dtype=np.float64)
    p = np.stack([x, np.zeros_like(x), np.zeros_like(x)], -1)
    kept, avg, stats = oracle.sor(p, 3, 1.0)
    exp = np.array([1.0] + [2.0 / 3.0] * 8 + [1.0, (91.0 + 92.0) / 3.0])
    np.testing.assert_allclose(avg, exp, rtol=1e-15)
    mean = 0.0
    for v in exp:
        mean += v
    mean /= 11
    sq = 0.0
    for v in exp:
        sq += (v - mean) * (v - mean)
    std = np.sqrt(sq / 10)  # Bessel
    assert stats[0] == mean and stats[1] == std and stats[2] == mean + 1.0 * std and stats[3] == 11
    assert list(kept) == list(range(10))


def test_sor_zero_average_is_dropped_and_counts_in_divisor(oracle):
    # five copies of one point: with k = 3 their mean distance is 0 -> not kept (avg > 0 is strict)
    rng = np.random.default_rng(6)
    p = np.concatenate([np.zeros((5, 3)), rng.random((200, 3)) + 1.0])
    kept, avg, stats = oracle.sor(p, 3, 5.0)
    assert (avg[:5] == 0).all() and not set(range(5)) & set(kept)
    assert stats[3] == 205  # zero averages still count as valid results
    assert stats[0] == pytest.approx(avg[avg > 0].sum() / 205, rel=1e-14)


def test_sor_bad_args(oracle):
    p = D.uniform_cube(100, 7)
    with pytest.raises(ValueError):
        oracle.sor_avg(p, 0)
    with pytest.raises(ValueError):
        oracle.sor_threshold(np.ones(5), 0.0)


def test_voxel_edges_and_centroids(oracle):
    vs = 0.5
    # min bound sits in the middle of voxel 0 (origin shifted by -vs/2); faces go to the upper voxel
    p = np.array([[0, 0, 0], [0.25, 0, 0], [0.2499999, 0, 0], [0.75, 0, 0], [1.0, 1.0, 1.0], [0.1, 0.1, 0.1],
                  [0.1, 0.1, 0.1]], dtype=np.float64)
    ijk, dims, vmin = oracle.voxel_index(p, vs)
    assert np.array_equal(vmin, [-0.25, -0.25, -0.25])
    assert list(ijk[:, 0]) == [0, 1, 0, 2, 2, 0, 0] and list(dims) == [3, 3, 3]
    c, _, _, cnt, vijk = oracle.voxel_downsample(p, vs, with_info=True)
    first = p[[0, 2, 5, 6]]
    assert np.array_equal(c[0], (first[0] + first[1] + first[2] + first[3]) / 4.0) and cnt[0] == 4
    assert cnt.sum() == len(p) and len(c) == 4
    with pytest.raises(ValueError):
        oracle.voxel_downsample(p, 0.0)
    with pytest.raises(RuntimeError, match="too small"):
        oracle.voxel_downsample(p * 1e6, 1e-7)


def test_voxel_matches_numpy_groupby(oracle):
    p = D.gaussian_clusters(20000, 8, fp32=False)
    vs = 0.37
    c, _, _, cnt, vijk = oracle.voxel_downsample(p, vs, with_info=True)
    ijk = np.floor((p - (p.min(0) - 0.5 * vs)) / vs).astype(np.int64)
    uniq, inv, counts = np.unique(ijk[:, ::-1], axis=0, return_inverse=True, return_counts=True)
    assert np.array_equal(uniq[:, ::-1], vijk) and np.array_equal(counts, cnt)
    sums = np.zeros((len(uniq), 3))
    np.add.at(sums, inv.reshape(-1), p)
    np.testing.assert_allclose(c, sums / counts[:, None], rtol=1e-12)


def test_normals_vs_eigh_and_degenerate_branches(oracle):
    p = D.bumpy_sphere(6000, 9)
    idx, _, _ = oracle.knn(p, None, 30)
    n = oracle.normals(p, 30)
    ref = np.zeros_like(n)
    for i in range(0, len(p), 7):
        w, v = np.linalg.eigh(np.cov(p[idx[i]].T, bias=True))
        ref[i] = v[:, 0]
    sel = np.arange(0, len(p), 7)
    assert angle_up_to_sign(n[sel], ref[sel]).max() < 1e-6
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    # plane z = 0 and a tilted plane
    rng = np.random.default_rng(10)
    plane = np.concatenate([rng.random((1000, 2)), np.zeros((1000, 1))], axis=1)
    assert np.allclose(np.abs(oracle.normals(plane, 16)[:, 2]), 1.0, atol=1e-7)
    t = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    u = np.cross(t, [0, 0, 1.0])
    u /= np.linalg.norm(u)
    v = np.cross(t, u)
    tilted = rng.random((1000, 1)) * u + rng.random((1000, 1)) * v
    assert angle_up_to_sign(oracle.normals(tilted, 16), np.tile(t, (1000, 1))).max() < 1e-6
    # fewer than 3 neighbours within the radius -> (0, 0, 1)
    far = D.uniform_cube(50, 11) * 1000.0
    assert np.array_equal(oracle.normals(far, 30, 0.01), np.tile([0.0, 0.0, 1.0], (50, 1)))
    # diagonal covariance -> axis of the smallest diagonal entry
    assert list(oracle.fast_eigen3x3([3.0, 0, 0, 1.0, 0, 2.0])) == [0, 1, 0]
    assert list(oracle.fast_eigen3x3([1.0, 0, 0, 1.0, 0, 1.0])) == [0, 0, 1]
    assert list(oracle.fast_eigen3x3([0.0, 0, 0, 0.0, 0, 0.0])) == [0, 0, 0]
    # prior normals decide the sign
    prior = p - p.mean(0)
    n2 = oracle.normals(p, 30, -1.0, prior)
    assert ((n2 * prior).sum(1) >= 0).all() and angle_up_to_sign(n2, n).max() < 1e-12


def test_fast_eigen_random_spd(oracle):
    rng = np.random.default_rng(12)
    for _ in range(300):
        a = rng.standard_normal((3, 3)) * rng.random(3) * 10
        c = a @ a.T
        v = oracle.fast_eigen3x3([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
        w, ev = np.linalg.eigh(c)
        if (w[1] - w[0]) / w[2] < 1e-6:
            continue
        assert angle_up_to_sign(v[None], ev[:, 0][None])[0] < 1e-5


def test_normalize(oracle):
    p = D.gaussian_clusters(5000, 13, fp32=False) * 1e3 + 5e5
    out, center, scale = oracle.normalize(p)
    assert np.array_equal(center, (p.min(0) + p.max(0)) * 0.5) and scale == (p.max(0) - p.min(0)).max()
    assert np.array_equal(out, (p + (-center)) * (1.0 / scale))
    assert (out.max(0) - out.min(0)).max() == pytest.approx(1.0, abs=1e-12)


def test_external_queries_and_brute_agree(oracle):
    p = D.uniform_cube(3000, 14)
    rng = np.random.default_rng(15)
    q = rng.random((400, 3)) * 3 - 1
    a = oracle.knn(p, q, 7)
    b = oracle.knn_brute(p, q, 7)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    g = oracle.Grid(p)
    s = g.knn_self(7)
    assert np.array_equal(s[0], oracle.knn(p, None, 7)[0])
