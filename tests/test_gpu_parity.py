"""GPU parity: every C-ABI entry point against the CPU oracle on the same seeded inputs.

Bar: bit-exact for indices, counts, voxel ids and masks; fp64 squared distances bit-exact
(same operation order on both sides); centroids and mean distances <= 1e-12 relative
(north_star asks 1e-5); normals <= 1e-4 rad up to sign (observed ~1e-12).
"""
import numpy as np
import pytest

import datasets as D

pytestmark = pytest.mark.gpu

CLOUDS = {
    "uniform": lambda: D.uniform_cube(20000, 11),
    "clusters": lambda: D.gaussian_clusters(15000, 12),
    "sphere": lambda: D.bumpy_sphere(12000, 13),
    "planes": lambda: D.planar_patches(20000, 14, extent=50.0, n_patches=10),
    "lidar": lambda: D.lidar_frame(15)[::6],
    "lattice": lambda: D.lattice(8),
    "fp64": lambda: D.gaussian_clusters(8000, 16, fp32=False),
}


def angle_up_to_sign(a, b):
    c = np.abs((a * b).sum(1)) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    return np.arccos(np.clip(c, -1, 1))


@pytest.mark.parametrize("name", list(CLOUDS))
@pytest.mark.parametrize("k", [1, 2, 16, 30, 50])
def test_knn_self(hip, oracle, name, k):
    p = CLOUDS[name]()
    idx, d2, cnt = hip.knn(p, None, k, -1.0, {})
    ridx, rd2, rcnt = oracle.knn(p, None, k)
    assert np.array_equal(cnt, rcnt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


def test_knn_brute_small(hip, oracle):
    p = D.gaussian_clusters(3000, 21)
    idx, d2, cnt = hip.knn(p, None, 20, -1.0, {})
    ridx, rd2, rcnt = oracle.knn_brute(p, None, 20)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)


@pytest.mark.parametrize("n,k", [(1, 1), (1, 4), (2, 2), (5, 16), (17, 16), (40, 64)])
def test_knn_fewer_points_than_k(hip, oracle, n, k):
    p = D.uniform_cube(n, 30 + n)
    idx, d2, cnt = hip.knn(p, None, k, -1.0, {})
    ridx, rd2, rcnt = oracle.knn_brute(p, None, k)
    assert np.array_equal(cnt, np.full(n, min(n, k)))
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


def test_knn_duplicates_tie_lower_index(hip, oracle):
    base = D.uniform_cube(500, 41)
    p = np.concatenate([base, base[:200], base[:50]], axis=0)
    idx, d2, cnt = hip.knn(p, None, 8, -1.0, {})
    ridx, rd2, _ = oracle.knn_brute(p, None, 8)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    # a duplicated point's first neighbours are its copies in ascending index order
    assert list(idx[0, :3]) == [0, 500, 700]


def test_knn_external_queries_inside_and_outside(hip, oracle):
    p = D.uniform_cube(20000, 51)
    rng = np.random.default_rng(52)
    q = np.concatenate([rng.random((3000, 3)), rng.random((500, 3)) * 3.0 - 1.0,
                        np.array([[10.0, 10.0, 10.0], [-5.0, 0.5, 0.5]])])
    q = q.astype(np.float32).astype(np.float64)
    idx, d2, cnt = hip.knn(p, q, 5, -1.0, {})
    ridx, rd2, rcnt = oracle.knn(p, q, 5)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)


@pytest.mark.parametrize("name,k", [("uniform", 1), ("uniform", 8), ("sphere", 1), ("clusters", 4), ("planes", 1),
                                    ("fp64", 3), ("lattice", 2)])
def test_knn_many_external_queries(hip, oracle, name, k):
    """Query sets large enough to be binned into the grid and answered by the tile kernel (the ICP shape:
    a moved copy of the cloud), plus queries far outside the box and in empty regions."""
    p = CLOUDS[name]()
    rng = np.random.default_rng(77)
    ext = p.max(0) - p.min(0)
    moved = p[rng.permutation(len(p))] + rng.normal(0.0, 0.002, p.shape) * ext + 0.01 * ext
    far = p.min(0) + (rng.random((600, 3)) * 3.0 - 1.0) * ext
    q = np.concatenate([moved, far, p[:500]])  # the last block coincides with points: exact zero distances
    if name != "fp64":
        q = q.astype(np.float32).astype(np.float64)
    while len(q) < 4096:  # (the lattice: shifted copies sit at equal distances from several points)
        q = np.concatenate([q, q + 0.25])
    idx, d2, cnt = hip.knn(p, q, k, -1.0, {})
    ridx, rd2, rcnt = oracle.knn(p, q, k)
    assert np.array_equal(cnt, rcnt) and np.array_equal(d2, rd2) and np.array_equal(idx, ridx)
    # hybrid radius on the same path
    r = 0.02 * float(ext.max())
    idx, d2, cnt = hip.knn(p, q, k, r, {})
    ridx, rd2, rcnt = oracle.knn(p, q, k, r)
    assert np.array_equal(cnt, rcnt)
    m = np.arange(k)[None, :] < cnt[:, None]
    assert np.array_equal(idx[m], ridx[m]) and np.array_equal(d2[m], rd2[m])


def test_knn_hybrid_radius(hip, oracle):
    p = D.bumpy_sphere(12000, 61)
    r = 0.004
    idx, d2, cnt = hip.knn(p, None, 30, r, {})
    ridx, rd2, rcnt = oracle.knn(p, None, 30, r)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert cnt.min() < 30 <= cnt.max()  # both the radius cut and the max_nn cut are exercised
    assert (d2[idx >= 0] < r * r).all()


def test_knn_f32_distance_output(hip, oracle):
    import torch

    p = D.uniform_cube(5000, 71)
    idx, d2, cnt = hip.knn(p, None, 16, -1.0, {}, d2_dtype=torch.float32)
    ridx, rd2, _ = oracle.knn(p, None, 16)
    assert np.array_equal(idx, ridx)
    np.testing.assert_allclose(d2, rd2, rtol=1e-5)  # north_star: 1e-5 relative in fp32


@pytest.mark.parametrize("name", ["uniform", "sphere", "planes", "fp64"])
def test_voxel(hip, oracle, name):
    p = CLOUDS[name]()
    ext = (p.max(0) - p.min(0)).max()
    for vs in (ext / 7.3, ext / 40.0):
        ijk, dims = hip.voxel_assign(p, vs)
        rijk, rdims, _ = oracle.voxel_index(p, vs)
        assert np.array_equal(ijk, rijk) and np.array_equal(dims, rdims)
        c, cnt, vijk = hip.voxel_downsample_info(p, vs)
        rc, _, _, rcnt, rvijk = oracle.voxel_downsample(p, vs, with_info=True)
        assert np.array_equal(vijk, rvijk) and np.array_equal(cnt, rcnt)
        assert np.array_equal(c, rc)  # same summation order -> identical centroids


def test_voxel_with_normals_colors(hip, oracle):
    p = D.uniform_cube(5000, 81)
    rng = np.random.default_rng(82)
    nrm = rng.standard_normal((5000, 3))
    col = rng.random((5000, 3))
    a = hip.voxel_downsample(p, nrm, col, 0.05, {})
    b = oracle.voxel_downsample(p, 0.05, nrm, col)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_voxel_bad_size(hip):
    from point_cloud_processing_amd._lib import PcError

    p = D.uniform_cube(100, 83)
    with pytest.raises(RuntimeError):
        hip.voxel_downsample(p, None, None, 0.0, {})
    with pytest.raises(PcError, match="too small"):
        hip.voxel_downsample(p * 1e6, None, None, 1e-7, {})


@pytest.mark.parametrize("name", ["uniform", "clusters", "sphere", "lattice"])
@pytest.mark.parametrize("k,sr", [(20, 2.0), (30, 3.0)])
def test_sor(hip, oracle, name, k, sr):
    p = CLOUDS[name]()
    cache = {}
    kept = hip.sor(p, k, sr, cache)
    rkept, ravg, rstats = oracle.sor(p, k, sr)
    avg = cache[("sor_avg", k)].cpu().numpy()
    np.testing.assert_allclose(avg, ravg, rtol=1e-13, atol=0)
    mean, std, thr, valid = hip.sor_stats(cache[("sor_avg", k)], sr, p, cache, k)
    assert valid == int(rstats[3])
    np.testing.assert_allclose([mean, std, thr], rstats[:3], rtol=1e-12)
    assert np.array_equal(kept, rkept)


@pytest.mark.parametrize("name", ["uniform", "sphere", "planes", "lidar", "fp64"])
def test_normals_knn(hip, oracle, name):
    p = CLOUDS[name]()
    n = hip.normals(p, 30, -1.0, None, {})
    rn = oracle.normals(p, 30)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    assert angle_up_to_sign(n, rn).max() < 1e-4
    assert np.abs(n - rn).max() < 1e-6  # same sign convention too


def test_normals_hybrid_and_prior(hip, oracle):
    p = D.bumpy_sphere(12000, 91)
    r = 0.02 * (p.max(0) - p.min(0)).max()
    n = hip.normals(p, 30, r, None, {})
    rn = oracle.normals(p, 30, r)
    assert angle_up_to_sign(n, rn).max() < 1e-4 and np.abs(n - rn).max() < 1e-6
    prior = p - p.mean(0)
    n2 = hip.normals(p, 30, r, prior, {})
    rn2 = oracle.normals(p, 30, r, prior)
    assert np.abs(n2 - rn2).max() < 1e-6
    assert ((n2 * prior).sum(1) >= 0).all()


def test_normals_degenerate(hip, oracle):
    # isolated points (fewer than 3 neighbours inside the radius) -> (0, 0, 1)
    p = D.uniform_cube(200, 92) * 100.0
    n = hip.normals(p, 30, 0.01, None, {})
    assert np.array_equal(n, np.tile([0.0, 0.0, 1.0], (200, 1)))
    # exact plane z = 0 -> +-z ; points on a line -> whatever the closed form gives, same as oracle
    rng = np.random.default_rng(93)
    plane = np.concatenate([rng.random((1000, 2)), np.zeros((1000, 1))], axis=1)
    n = hip.normals(plane, 16, -1.0, None, {})
    assert np.allclose(np.abs(n[:, 2]), 1.0, atol=1e-7)
    line = np.stack([np.linspace(0, 1, 300)] * 3, -1)
    assert np.abs(hip.normals(line, 10, -1.0, None, {}) - oracle.normals(line, 10)).max() < 1e-9


@pytest.mark.parametrize("name", ["uniform", "clusters", "sphere"])
def test_radius_count_and_nn_distance(hip, oracle, name):
    p = CLOUDS[name]()
    nn = hip.nn_distance(p, {})
    rnn = oracle.nn_distance(p)
    assert np.array_equal(nn, rnn)
    r = 5.0 * float(np.median(rnn))
    full = oracle.radius_count(p, r)
    assert np.array_equal(hip.radius_count(p, r, {}), full)
    # capped counts (what remove_radius_outlier asks for) ride the hybrid kNN search; a radius far
    # beyond the cells and one that excludes most neighbours both have to agree with the full count
    for cap, rr in [(5, r), (5, 20.0 * r), (3, 0.6 * float(np.median(rnn))), (17, r)]:
        want = np.minimum(oracle.radius_count(p, rr), cap)
        assert np.array_equal(hip.radius_count(p, rr, {}, cap=cap), want)


def test_radius_count_strict(hip):
    p = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 2.0, 0]])
    assert list(hip.radius_count(p, 1.0, {})) == [1, 1, 1]  # distance == r is not inside
    assert list(hip.radius_count(p, 1.0000001, {})) == [2, 2, 1]
    assert list(hip.nn_distance(p[:1], {})) == [0.0]


def test_bbox_affine(hip, oracle):
    p = D.gaussian_clusters(7000, 95, fp32=False)
    mn, mx = hip.bbox(p, {})
    assert np.array_equal(mn, p.min(0)) and np.array_equal(mx, p.max(0))
    out, center, scale = oracle.normalize(p)
    got = hip.affine(p, {}, -center, 1.0 / scale, np.zeros(3))
    assert np.array_equal(got, out)


@pytest.mark.parametrize("n", [65_536, 300_001, 1_000_003])
def test_bbox_of_large_fp32_clouds(hip, n):
    """Bounding boxes of large fp32 clouds (grid-stride kernel, 1024 blocks): extremes anywhere -- first point, last
    point, in the middle -- come out exactly."""
    rng = np.random.default_rng(n)
    p = (rng.random((n, 3)) * np.array([3.0, 70.0, 0.5]) + np.array([-1.0, 5.0, 100.0])).astype(np.float32)
    p[0] = [-9.0, 4.0, 99.0]
    p[n - 1] = [9.0, 4.5, 101.5]
    p[n - 2, 1] = 200.0
    p[n // 2, 2] = -3.0
    mn, mx = hip.bbox(p, {})
    assert np.array_equal(mn, p.min(0).astype(np.float64)) and np.array_equal(mx, p.max(0).astype(np.float64))


def test_determinism(hip):
    p = D.uniform_cube(30000, 97)
    a = hip.knn(p, None, 16, -1.0, {})
    b = hip.knn(p, None, 16, -1.0, {})
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(hip.normals(p, 16, -1.0, None, {}), hip.normals(p, 16, -1.0, None, {}))


@pytest.mark.parametrize("name", ["uniform", "sphere", "lidar", "fp64"])
def test_fused_equals_separate(hip, oracle, name):
    p = CLOUDS[name]()
    idx, d2, cnt, avg, nrm = hip.knn_fused(p, 16, -1.0, {})
    ridx, rd2, rcnt = oracle.knn(p, None, 16)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)
    np.testing.assert_allclose(avg, oracle.sor_avg(p, 16), rtol=1e-13)
    assert np.abs(nrm - oracle.normals(p, 16)).max() < 1e-6


def test_large_uniform_matches_oracle(hip, oracle):
    """1 M uniform points, k = 16 (BASELINE config 2): the tile path and its exact re-run."""
    p = D.uniform_cube(1_000_000, 2)
    idx, d2, cnt, avg, nrm = hip.knn_fused(p, 16, -1.0, {})
    g = oracle.Grid(p)
    ridx, rd2, rcnt = g.knn_self(16)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)
    np.testing.assert_allclose(avg, np.sqrt(rd2).sum(1) / 16.0, rtol=1e-13)


def test_headline_10m_matches_oracle(hip, oracle):
    """The bench workload itself (10 M uniform points, seed 3, k = 16): every row against the CPU oracle's
    grid search, plus the size-independent properties (self first at distance 0, ascending distances)."""
    p = D.uniform_cube(10_000_000, 3)
    idx, d2, cnt = hip.knn(p, None, 16, -1.0, {})
    assert (cnt == 16).all()
    assert np.array_equal(idx[:, 0], np.arange(len(p), dtype=np.int32)) and not d2[:, 0].any()
    assert (np.diff(d2, axis=1) >= 0).all()
    ridx, rd2, rcnt = oracle.Grid(p).knn_self(16)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_query_shards_cover_and_agree(hip, oracle, world):
    """Each rank's slice of sorted positions, run one after the other on this GPU, reproduces the
    single-GPU result exactly (what 2/4/8 ranks compute in parallel), in outputs of the slice's own size."""
    from point_cloud_processing_amd.distributed import shard_positions

    p = D.gaussian_clusters(30000, 77)
    cache = {}
    ravg = oracle.sor_avg(p, 16)
    rn = oracle.normals(p, 16)
    ridx, rd2, rcnt = oracle.knn(p, None, 16)
    seen = np.zeros(len(p), dtype=int)
    for r in range(world):
        b, e = shard_positions(len(p), r, world)
        sh = {k: v.cpu().numpy() for k, v in hip.shard_search(p, cache, 16, -1.0, b, e, want_rows=True, want_avg=True,
                                                             want_normals=True).items()}
        owned = sh["owned"]
        assert all(len(v) == e - b for v in sh.values())
        np.testing.assert_allclose(sh["avg"], ravg[owned], rtol=1e-13)
        assert np.abs(sh["normals"] - rn[owned]).max() < 1e-6
        assert np.array_equal(sh["idx"], ridx[owned]) and np.array_equal(sh["d2"], rd2[owned])
        assert np.array_equal(sh["cnt"], rcnt[owned])
        seen[owned] += 1
    assert (seen == 1).all()


@pytest.mark.parametrize("name", ["sphere", "clusters"])
def test_sharded_filter_carries_the_same_bits_for_any_world(hip, oracle, name):
    """The multi-GPU statistical filter and normals (distributed.sor_steps / normals_steps, the generators
    that run over RCCL) for 1, 2, 3 and 8 pretended ranks on this GPU: kept indices, mean, standard deviation
    and threshold are the same bits for every world size and equal the single-GPU engine call."""
    from point_cloud_processing_amd import distributed as PD

    p = (D.bumpy_sphere(40000, 88) if name == "sphere" else D.gaussian_clusters(40000, 89))
    p = np.concatenate([p, np.random.default_rng(9).random((300, 3))]).astype(np.float32).astype(np.float64)
    single = np.asarray(hip.sor(p, 20, 2.0, {}))
    rkept, _, rstats = oracle.sor(p, 20, 2.0)
    assert np.array_equal(single, rkept)
    first = None
    for world in (1, 2, 3, 8):
        cache = {}
        res = PD.run_emulated(lambda r: PD.sor_steps(hip, p, 20, 2.0, cache, r, world), world)
        for kept, stats in res:
            assert np.array_equal(np.asarray(kept), single)
            first = first or stats
            assert stats == first  # bit-identical (mean, std, threshold, valid)
        np.testing.assert_allclose(first[:3], rstats[:3], rtol=1e-12)
    assert tuple(hip.sor_stats(hip.sor_avg_dev(p, 20, {}), 2.0)) == tuple(first)  # the one-GPU call: the same bits
    nrm = PD.run_emulated(lambda r: PD.normals_steps(hip, p, 16, -1.0, {}, r, 3), 3)
    one = hip.normals(p, 16, -1.0, None, {})
    for x in nrm:
        assert np.array_equal(x.cpu().numpy(), one)


@pytest.mark.parametrize("case", ["all_duplicates", "collinear", "coplanar_lattice", "far_offset_fp64", "two_far_clusters",
                                  "one_outlier_far_away"])
def test_knn_pathological_shapes(hip, oracle, case):
    """Clouds that stress the index rather than the arithmetic: zero extent, one- and two-dimensional clouds,
    coordinates with a huge common offset (fp64 records), an extent many orders above the point spacing."""
    rng = np.random.default_rng(101)
    if case == "all_duplicates":
        p = np.tile(np.array([[0.25, -1.5, 3.0]]), (5000, 1))
    elif case == "collinear":
        p = np.outer(rng.random(20000), np.array([1.0, 2.0, -0.5]))
    elif case == "coplanar_lattice":
        gx, gy = np.meshgrid(np.arange(120.0), np.arange(120.0))
        p = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 7.0)], 1)
    elif case == "far_offset_fp64":
        p = rng.random((30000, 3)) * 0.01 + np.array([1.0e6, -2.0e6, 3.0e5])
    elif case == "two_far_clusters":
        p = np.concatenate([rng.normal(0, 0.001, (20000, 3)), rng.normal(0, 0.001, (20000, 3)) + 5.0e4])
    else:
        p = np.concatenate([rng.random((40000, 3)), np.array([[1.0e7, 1.0e7, 1.0e7]])])
    if case != "far_offset_fp64":
        p = p.astype(np.float32).astype(np.float64)
    p = np.ascontiguousarray(p)
    for k in (1, 8, 20):
        idx, d2, cnt = hip.knn(p, None, k, -1.0, {})
        ridx, rd2, rcnt = oracle.knn(p, None, k)
        assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and np.array_equal(d2, rd2), (case, k)
    avg = hip.sor_avg_dev(p, 8, {}).cpu().numpy()
    np.testing.assert_allclose(avg, oracle.sor_avg(p, 8), rtol=1e-13, atol=0)
    np.testing.assert_array_equal(hip.nn_distance(p, {}), oracle.nn_distance(p))
