"""The app-facing entry points running on the HIP engine, checked end to end against the
oracle (reconstruction_controller.py:45-53 and app_controller.py:44-68 are the callers)."""
import numpy as np
import pytest

import datasets as D
import point_cloud_processing_amd as P
from point_cloud_processing_amd.app.Core import estimate_bpa_radii, normalize_point_cloud, preprocess_point_cloud
from point_cloud_processing_amd.app.Infrastructure import FilterSettings, filter_pcd

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_engine")]


def test_normalize_preprocess_radii(oracle):
    pts = D.bumpy_sphere(35000, 1)
    pcd, center, scale = normalize_point_cloud(P.PointCloud(pts))
    ref, rc, rs = oracle.normalize(pts)
    assert np.array_equal(np.asarray(pcd.points), ref) and np.array_equal(center, rc) and scale == rs
    proc = preprocess_point_cloud(pcd, voxel_size=0.005)
    vp, _, _ = oracle.voxel_downsample(ref, 0.005)
    assert np.array_equal(np.asarray(proc.points), vp)
    radius = 0.02 * (vp.max(0) - vp.min(0)).max()
    rn = oracle.normals(vp, 30, radius)
    got = np.asarray(proc.normals)  # oriented by preprocess: the oracle's lines, outward signs
    assert np.minimum(np.abs(got - rn).max(1), np.abs(got + rn).max(1)).max() < 1e-6
    assert (np.einsum("ij,ij->i", got, vp - vp.mean(0)) > 0).mean() > 0.999
    radii = estimate_bpa_radii(proc)
    base = 1.5 * oracle.nn_distance(vp).mean()
    np.testing.assert_allclose(radii, [base, 2 * base, 3 * base], rtol=1e-13)


@pytest.mark.parametrize("case", ["default", "relax", "voxel"])
def test_filter_pcd(oracle, case):
    from test_host import _reference_filter

    rng = np.random.default_rng(5)
    if case == "relax":
        pts = np.concatenate([D.uniform_cube(9000, 5) * 0.2, rng.random((3000, 3)) * 6.0])
        s = FilterSettings(max_drop=0.05)
    elif case == "voxel":
        pts = np.concatenate([D.bumpy_sphere(6000, 6), rng.random((60, 3)) * 0.3])
        s = FilterSettings(use_voxel=True)
    else:
        pts = np.concatenate([D.bumpy_sphere(8000, 8), rng.random((80, 3)) * 0.3])
        s = None
    pts = pts.astype(np.float32).astype(np.float64)
    out = filter_pcd(P.PointCloud(pts), s)
    ref, _ = _reference_filter(oracle, pts, s or FilterSettings())
    assert np.array_equal(np.asarray(out.points), ref)


def test_sor_cache_reused_across_relax_attempts(hip):
    pcd = P.PointCloud(D.uniform_cube(20000, 9))
    a, ia = pcd.remove_statistical_outlier(30, 3.0)
    assert ("sor_avg", 30) in pcd._cache
    avg = pcd._cache[("sor_avg", 30)]
    b, ib = pcd.remove_statistical_outlier(30, 3.8)
    assert pcd._cache[("sor_avg", 30)] is avg and len(ib) >= len(ia)


@pytest.mark.parametrize("case", ["two_spheres", "torus_k50", "lattice_ties", "clusters"])
def test_orient_normals_matches_oracle(hip, oracle, case):
    """pc_orient_normals against the oracle's spanning-tree propagation: the edge order (fp32 weight bits,
    then slot number) makes the tree unique, so every sign must agree; piece counts too."""
    rng = np.random.default_rng(11)
    k = 10
    if case == "two_spheres":
        p = rng.standard_normal((9000, 3))
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        p = np.concatenate([p[:6000], p[6000:] * 0.5 + np.array([10.0, 0.0, 0.0])])
    elif case == "torus_k50":
        p, k = D.torus(20000, 6), 50
    elif case == "lattice_ties":  # a flat lattice: all weights equal, the slot number alone orders the edges
        gx, gy = np.meshgrid(np.arange(40.0), np.arange(40.0))
        p, k = np.stack([gx.ravel(), gy.ravel(), np.zeros(1600)], 1), 9
    else:  # many pieces of very different sizes, some below k points
        p = np.concatenate([D.gaussian_clusters(5000, 3), rng.random((7, 3)) + 50.0, rng.random((1, 3)) - 70.0])
    p = np.ascontiguousarray(p.astype(np.float32).astype(np.float64))
    k = min(k, len(p))
    nrm = oracle.normals(p, 12)
    nrm[rng.random(len(p)) < 0.5] *= -1.0
    idx, _, cnt = oracle.knn(p, None, k, None)
    want, pieces = oracle.orient_normals(p, nrm, idx, cnt)
    got, got_pieces = hip.orient_normals(p, nrm, k, {})
    assert got_pieces == pieces
    assert np.array_equal(got, want)
    if case == "two_spheres":
        centre = np.where((np.arange(len(p)) < 6000)[:, None], 0.0, np.array([10.0, 0.0, 0.0]))
        assert (np.einsum("ij,ij->i", got, p - centre) > 0).all()


def test_registration_icp_matches_oracle(hip, oracle):
    """The device ICP loop (search, 6 x 6 reduction, rigid transform) against the numpy statement of the same
    steps: same correspondences at every iteration, so the same iteration count and, up to the order of the
    sums, the same transformation."""
    from oracle.engine import OracleEngine
    from test_host import _icp_case
    P.set_engine(OracleEngine())
    try:
        source, target, motion = _icp_case(60000, 20000)
        want = P.registration_icp(source, target, 0.2, max_iteration=30)
    finally:
        P.set_engine(hip)
    got = P.registration_icp(source, target, 0.2, max_iteration=30)
    assert got.iterations == want.iterations and got.fitness == want.fitness
    assert np.array_equal(got.correspondence_set, want.correspondence_set)
    np.testing.assert_allclose(got.transformation, want.transformation, rtol=0, atol=1e-10)
    np.testing.assert_allclose(got.inlier_rmse, want.inlier_rmse, rtol=1e-9, atol=1e-12)
    assert np.abs(got.transformation - motion).max() < 2e-3
    # one step's terms, entry by entry
    s_dev = hip.icp_session(np.asarray(source.points), np.asarray(target.points), np.asarray(target.normals))
    s_ref = OracleEngine().icp_session(np.asarray(source.points), np.asarray(target.points), np.asarray(target.normals))
    for s in (s_dev, s_ref):
        s.transform(motion)
        s.correspond()
    np.testing.assert_allclose(s_dev.terms(0.05), s_ref.terms(0.05), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 3, 7])
def test_orient_normals_tiny_clouds(hip, oracle, n):
    """Fewer points than neighbours asked for: rows are short, every piece still gets its seed."""
    rng = np.random.default_rng(n)
    p = np.ascontiguousarray(rng.random((n, 3)).astype(np.float32).astype(np.float64))
    nrm = rng.standard_normal((n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    k = max(2, min(50, n))
    idx, _, cnt = oracle.knn(p, None, k, None)
    want, pieces = oracle.orient_normals(p, nrm, idx, cnt)
    got, got_pieces = hip.orient_normals(p, nrm, k, {})
    assert got_pieces == pieces == 1 and np.array_equal(got, want)
    assert got[np.argmax(p[:, 2]), 2] >= 0.0


def test_new_entry_points_reject_bad_arguments(hip):
    import ctypes as C
    lib, ctx = hip.lib, hip.ctx
    one = C.c_int64(0)
    assert lib.pc_orient_normals(ctx, None, 0, None, None, None, 5, 4, C.byref(one), None) != 0   # null buffers
    assert lib.pc_orient_normals(ctx, None, 0, None, None, None, 0, 4, C.byref(one), None) == 0   # empty cloud
    assert lib.pc_orient_normals(ctx, None, 7, None, None, None, 0, 4, C.byref(one), None) != 0   # unknown dtype
    assert lib.pc_orient_normals(ctx, None, 0, None, None, None, 0, 0, C.byref(one), None) != 0   # k < 1
    out = (C.c_double * 29)()
    assert lib.pc_icp_point_to_plane_terms(ctx, None, 3, None, 0, None, None, None, 0, 1.0, out, None) != 0
    assert lib.pc_icp_point_to_plane_terms(ctx, None, 0, None, 0, None, None, None, 0, 1.0, out, None) == 0
    assert lib.pc_transform(ctx, None, 4, None, None) != 0
    assert b"pc_transform" in lib.pc_last_error(ctx)


def test_two_threads_share_one_engine(hip, oracle):
    """The app's filter and reconstruction workers run in two threads (app_controller.py:428-441, 528-543) and
    reach the same engine: concurrent calls on one context must give what they give one after the other."""
    import threading
    a = D.gaussian_clusters(60000, 41)
    b = D.bumpy_sphere(50000, 42)
    want_a = oracle.knn(a, None, 16)
    want_b = oracle.sor_avg(b, 20)
    errors = []

    def work_a():
        try:
            for _ in range(6):
                idx, d2, cnt = hip.knn(a, None, 16, -1.0, {})
                assert np.array_equal(idx, want_a[0]) and np.array_equal(d2, want_a[1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def work_b():
        try:
            for _ in range(6):
                avg = hip.sor_avg_dev(b, 20, {}).cpu().numpy()
                np.testing.assert_allclose(avg, want_b, rtol=1e-13)
                vp, _, _ = hip.voxel_downsample(b, None, None, 0.01, {})
                assert len(vp) > 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=work_a), threading.Thread(target=work_b)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_calls_alternate_between_streams(hip, oracle):
    """Pooled scratch is handed from call to call in stream order; a call on another stream waits for the
    previous stream first, so alternating streams changes nothing."""
    import torch
    p = D.gaussian_clusters(80000, 51)
    want = oracle.knn(p, None, 16)
    s1, s2 = torch.cuda.Stream(device=hip.device), torch.cuda.Stream(device=hip.device)
    for it in range(6):
        with torch.cuda.stream(s1 if it % 2 == 0 else s2):
            idx, d2, cnt = hip.knn(p, None, 16, -1.0, {})
        assert np.array_equal(idx, want[0]) and np.array_equal(d2, want[1])


def test_chains_stay_on_the_device(hip, oracle):
    """normalise -> voxel -> normals and statistical -> radius filter: the attributes of the intermediate and
    final clouds live in HBM until somebody reads them, and then equal the oracle's bit for bit."""
    pts = D.bumpy_sphere(20000, 21)
    pcd, center, scale = normalize_point_cloud(P.PointCloud(pts))
    assert pcd._h["points"] is None and "points" in pcd._d      # edited on the device, not copied back
    proc = pcd.voxel_down_sample(0.01)
    proc.estimate_normals(P.KDTreeSearchParamKNN(16))
    assert proc._h["points"] is None and proc._h["normals"] is None
    ref, _, _ = oracle.normalize(pts)
    vp, _, _ = oracle.voxel_downsample(ref, 0.01)
    assert len(proc) == len(vp) and np.array_equal(np.asarray(proc.points), vp)
    assert np.abs(np.asarray(proc.normals) - oracle.normals(vp, 16)).max() < 1e-6
    # the outlier chain of filter_pcd
    noisy = np.concatenate([pts, np.random.default_rng(3).random((200, 3)) * 2.0]).astype(np.float32).astype(np.float64)
    cl = P.PointCloud(noisy)
    a, ia = cl.remove_statistical_outlier(30, 3.0)
    b, ib = a.remove_radius_outlier(4, 0.05)
    assert a._h["points"] is None and b._h["points"] is None    # gathered on the device
    ka, _, _ = oracle.sor(noisy, 30, 3.0)
    assert len(ia) == len(ka) and np.array_equal(np.asarray(ia), ka)
    kb = np.nonzero(oracle.radius_count(noisy[ka], 0.05) > 4)[0]
    assert np.array_equal(np.asarray(ib), kb) and np.array_equal(np.asarray(b.points), noisy[ka][kb])


def test_select_by_index_on_the_device(hip):
    """pc_gather_rows / pc_gather_rows_f32 behind select_by_index: points, normals and colors, fp32 and fp64."""
    rng = np.random.default_rng(5)
    for fp32 in (True, False):
        pts = D.gaussian_clusters(5000, 22, fp32=fp32)
        pcd = P.PointCloud(pts)
        pcd.normals = rng.normal(size=pts.shape)
        pcd.colors = rng.random(pts.shape)
        pcd.get_axis_aligned_bounding_box()                      # (uploads the coordinates)
        idx = rng.permutation(5000)[:1234]
        sel = pcd.select_by_index(P.geometry.LazyIndices(hip, hip.upload(idx.astype(np.int32)), len(idx)))
        assert sel._h["points"] is None
        assert np.array_equal(np.asarray(sel.points), pts[idx])
        assert np.array_equal(np.asarray(sel.normals), np.asarray(pcd.normals)[idx])
        assert np.array_equal(np.asarray(sel.colors), np.asarray(pcd.colors)[idx])
        inv = pcd.select_by_index(idx, invert=True)
        assert len(inv) == 5000 - 1234


def test_non_finite_coordinates_are_rejected(hip):
    """A cloud with an infinite or all-NaN coordinate has no finite bounding box: the index build says so
    instead of spinning or building a garbage grid."""
    p = D.uniform_cube(1000, 23, fp32=False) if "fp32" in D.uniform_cube.__code__.co_varnames else D.uniform_cube(1000, 23)
    bad = p.copy()
    bad[7, 0] = np.inf
    bad[9, 1] = -np.inf
    with pytest.raises(RuntimeError, match="non-finite"):
        hip.knn(bad, None, 4, -1.0, {})
    allnan = np.full((100, 3), np.nan)
    with pytest.raises(RuntimeError, match="non-finite"):
        hip.knn(allnan, None, 4, -1.0, {})
    with pytest.raises(RuntimeError, match="non-finite"):
        hip.voxel_downsample(bad, None, None, 0.1, {})


def test_loader_subsample_gathers_on_the_device(hip):
    from point_cloud_processing_amd.app.Infrastructure.pointcloud_io import subsample_indices, subsample_point_cloud

    pts = D.bumpy_sphere(50000, 24)
    pcd = P.PointCloud(pts)
    out = subsample_point_cloud(pcd, 20000, "a/b.e57")
    assert out._h["points"] is None
    assert np.array_equal(np.asarray(out.points), pts[subsample_indices(50000, 20000, "a/b.e57")])


def test_slab_entry_points(hip):
    """The slab calls through the C ABI on their own: histogram, selection, z range, merge and the consumers
    of finished rows, each against numpy; bad arguments come back as a status."""
    import ctypes as C
    import torch
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd.hip_engine import _ptr
    lib, ctx = hip.lib, hip.ctx
    one, two = C.c_int64(0), C.c_int64(0)
    lo3, hi3 = (C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 1)
    cnt = (C.c_uint32 * 8)()
    assert lib.pc_axis_histogram(ctx, None, 0, 5, 2, lo3, hi3, 8, cnt, None) != 0          # null cloud
    assert lib.pc_axis_histogram(ctx, None, 0, 0, 2, lo3, hi3, 8, cnt, None) == 0          # empty cloud
    assert lib.pc_axis_histogram(ctx, None, 0, 0, 2, lo3, hi3, 5000, cnt, None) != 0       # too many bins
    assert lib.pc_select_slab(ctx, None, 0, 5, 2, 0.0, 1.0, None, C.byref(one), None) != 0
    assert lib.pc_grid_axis_range(ctx, None, 0.0, 1.0, C.byref(one), C.byref(two), None) != 0
    assert lib.pc_grid_relabel(ctx, None, None, 5, None) != 0
    assert lib.pc_merge_rows(ctx, 0, 1, 4, None, None, None, None, None, None, None) != 0   # no ranks
    assert lib.pc_merge_rows(ctx, 2, 0, 4, None, None, None, None, None, None, None) == 0   # no queries
    assert lib.pc_rows_consumers(ctx, None, 0, None, 3, 4, None, None, None, None, None) != 0
    assert lib.pc_slab_finish(ctx, None, 0, None, 3, 4, -1.0, None, None, 0, None, None, 0.0, 1.0, 0.0, 1.0, None, None, None,
                              C.byref(one), C.byref(two), None) != 0

    rng = np.random.default_rng(12)
    p = rng.random((50000, 3)).astype(np.float32)
    t = torch.from_numpy(p).to(hip.device)
    h = hip.axes_histogram(t, {}, [0, 0, 0], [1, 1, 1], 64)
    want = np.stack([np.bincount(np.clip(np.floor(p[:, a].astype(np.float64) * 64), 0, 63).astype(int), minlength=64) for a in range(3)])
    assert np.array_equal(h, want)
    idx = torch.empty(50000, dtype=torch.int32, device=hip.device)
    m = C.c_int64(0)
    hip._check(lib.pc_select_slab(ctx, _ptr(t), _lib.PC_F32, 50000, 2, 0.25, 0.5, _ptr(idx), C.byref(m), hip.stream()), "select")
    assert np.array_equal(idx[: m.value].cpu().numpy(), np.nonzero((p[:, 2] >= 0.25) & (p[:, 2] < 0.5))[0])
    # z range of a grid: every point with z in [0.25, 0.5] lies inside the range of sorted positions
    g = hip.grid(t, {}, 16)
    b, e = C.c_int64(0), C.c_int64(0)
    hip._check(lib.pc_grid_axis_range(ctx, g.h, 0.25, 0.5, C.byref(b), C.byref(e), hip.stream()), "range")
    perm = torch.empty(50000, dtype=torch.int32, device=hip.device)
    hip._check(lib.pc_grid_perm_copy(ctx, g.h, 0, 50000, _ptr(perm), hip.stream()), "perm")
    inside = np.zeros(50000, dtype=bool)
    inside[perm[b.value:e.value].cpu().numpy()] = True
    assert inside[(p[:, 2] >= 0.25) & (p[:, 2] <= 0.5)].all() and 0 < e.value - b.value < 0.4 * 50000
    # merge: three ranks' rows of 40 queries, k = 6, with shared entries
    k, nq, world = 6, 40, 3
    pool_d = np.sort(rng.random((nq, 14)), axis=1)
    pool_i = np.stack([rng.permutation(1000)[:14] for _ in range(nq)]).astype(np.int32)
    pool_d[:, 5] = pool_d[:, 4]  # an exact tie: lower index first
    ia = np.full((world, nq, k), -1, dtype=np.int32)
    da = np.full((world, nq, k), np.inf)
    ca = np.zeros((world, nq), dtype=np.int32)
    for q in range(nq):
        ent = sorted(zip(pool_d[q], pool_i[q]))
        for w in range(world):
            mine = [x for j, x in enumerate(ent) if (j + q) % world == w or j % 4 == 0][: k - (q % 2)]  # overlapping subsets
            ca[w, q] = len(mine)
            for s, (d, i) in enumerate(mine):
                ia[w, q, s], da[w, q, s] = i, d
    midx, md2, mcnt = hip.merge_rows(*(torch.from_numpy(x).to(hip.device) for x in (ia, da, ca)))
    for q in range(nq):
        ent = sorted({(float(da[w, q, s]), int(ia[w, q, s])) for w in range(world) for s in range(ca[w, q])})[:k]
        assert int(mcnt[q]) == len(ent)
        assert [(float(d), int(i)) for d, i in zip(md2[q, :len(ent)].cpu(), midx[q, :len(ent)].cpu())] == ent
    # consumers of finished rows = the fused search's own outputs
    pts = p.astype(np.float64)
    ridx, rd2, rcnt, ravg, rnrm = hip.knn_fused(pts, 12, -1.0, {})
    ql = torch.arange(0, 50000, 7, dtype=torch.int32, device=hip.device)
    sel = ql.cpu().numpy()
    avg, nrm = hip.rows_consumers(t, {}, ql, torch.from_numpy(ridx[sel]).to(hip.device), torch.from_numpy(rcnt[sel]).to(hip.device), True)
    assert np.array_equal(avg.cpu().numpy(), ravg[sel]) and np.array_equal(nrm.cpu().numpy(), rnrm[sel])
