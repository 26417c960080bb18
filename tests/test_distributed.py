"""Multi-rank query sharding: world_size 2 and 3 over gloo on CPU (oracle engine), checked
against the one-rank run of the same code.  The same generators run over RCCL with the HIP engine
(tests/test_gpu_configs.py drives them for several pretended world sizes on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import datasets as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cloud():
    # (more than two reduction chunks, so that every rank of 3 owns some)
    pts = np.concatenate([D.bumpy_sphere(13000, 3), np.random.default_rng(4).random((177, 3)) * 0.3])
    return pts.astype(np.float32).astype(np.float64)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (gloo otherwise looks the host name up, which may not resolve)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.engine import OracleEngine
        from point_cloud_processing_amd import distributed as PD

        eng = OracleEngine()
        pts = _cloud()
        kept, stats = PD.sor_sharded(eng, pts, 20, 2.0)
        nrm = PD.normals_sharded(eng, pts, 30, 0.01)
        sh = PD.knn_sharded(eng, pts, 8)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), kept=np.asarray(kept), stats=np.array(stats), nrm=nrm,
                 owned=sh["owned"], idx=sh["idx"], d2=sh["d2"], cnt=sh["cnt"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matches_single_rank(oracle, tmp_path, world):
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    pts = _cloud()
    # one rank, same code: the statistics must carry the same bits for every number of ranks
    (kept1, stats1), = PD.run_emulated(lambda r: PD.sor_steps(OracleEngine(), pts, 20, 2.0, {}, r, 1), 1)
    rkept, ravg, rstats = oracle.sor(pts, 20, 2.0)
    assert np.array_equal(kept1, rkept)
    np.testing.assert_allclose(stats1[:3], rstats[:3], rtol=1e-13)  # (the oracle's own sum is another tree)
    rn = oracle.normals(pts, 30, 0.01)
    ridx, rd2, rcnt = oracle.knn(pts, None, 8)
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    seen = np.zeros(len(pts), dtype=int)
    for r in res:
        assert np.array_equal(r["kept"], rkept)            # identical on every rank, equal to 1 rank
        assert np.array_equal(r["stats"], np.array(stats1))  # bit-identical to the one-rank statistics
        assert np.array_equal(r["nrm"], rn)
        assert np.array_equal(r["idx"], ridx[r["owned"]]) and np.array_equal(r["d2"], rd2[r["owned"]])
        assert len(r["owned"]) == len(r["idx"]) == len(r["cnt"])  # outputs of the shard's own size
        seen[r["owned"]] += 1
    assert (seen == 1).all()                               # the shards partition the queries


def test_emulated_world_equals_real_world(oracle):
    """run_emulated (all ranks in one process) gives what the real collective gives: same generators."""
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    pts = _cloud()
    one = PD.run_emulated(lambda r: PD.sor_steps(OracleEngine(), pts, 20, 2.0, {}, r, 1), 1)[0]
    for world in (2, 4, 8):
        res = PD.run_emulated(lambda r: PD.sor_steps(OracleEngine(), pts, 20, 2.0, {}, r, world), world)
        for kept, stats in res:
            assert np.array_equal(kept, one[0]) and stats == one[1]


def test_shard_range_partitions():
    from point_cloud_processing_amd.distributed import SOR_CHUNK, shard_positions, shard_range

    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= 1
            pos = [shard_positions(n, r, w) for r in range(w)]
            assert pos[0][0] == 0 and pos[-1][1] == n
            assert all(pos[i][1] == pos[i + 1][0] for i in range(w - 1))
            assert all(b % SOR_CHUNK == 0 for b, e in pos if e > b)


# ---- the HIP engine in real processes ------------------------------------------------------------
def _hip_worker(rank, world, port, out_dir, backend):
    """One rank of a real process group with the production engine: every rank on cuda:0 (the test box has
    one GPU; RCCL itself wants a GPU per rank, so the collective here is gloo's, on DEVICE tensors)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (gloo otherwise looks the host name up, which may not resolve)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from point_cloud_processing_amd import distributed as PD
        from point_cloud_processing_amd.hip_engine import HipEngine

        eng = HipEngine(0)
        pts = _cloud()
        kept, stats = PD.sor_sharded(eng, pts, 20, 2.0)
        nrm = PD.normals_sharded(eng, pts, 30, 0.01)
        sh = PD.knn_sharded(eng, pts, 8)
        host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), kept=host(kept), stats=np.array(stats), nrm=host(nrm),
                 owned=host(sh["owned"]), idx=host(sh["idx"]), d2=host(sh["d2"]), cnt=host(sh["cnt"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_hip_engine_in_real_processes(hip, oracle, tmp_path, world):
    """The sharded filter, normals and rows with the HIP engine in ``world`` real processes (device tensors
    through the process group's all-reduce), against the single-process HIP result and the oracle."""
    from point_cloud_processing_amd import distributed as PD

    port = _free_port()
    try:
        mp.spawn(_hip_worker, args=(world, port, str(tmp_path), "gloo"), nprocs=world, join=True)
    except Exception as e:  # a rendezvous that cannot be set up on this box is not a finding about the library
        text = str(e)
        if any(w in text for w in ("init_process_group", "Address already in use", "Connection refused", "connect()",
                                   "rendezvous", "Timed out")) and "AssertionError" not in text and "PcError" not in text:
            pytest.skip("process group could not be set up here: " + text.strip().splitlines()[-1][:200])
        raise
    pts = _cloud()
    (kept1, stats1), = PD.run_emulated(lambda r: PD.sor_steps(hip, pts, 20, 2.0, {}, r, 1), 1)
    kept1 = kept1.cpu().numpy() if hasattr(kept1, "cpu") else np.asarray(kept1)
    rkept, _, rstats = oracle.sor(pts, 20, 2.0)
    assert np.array_equal(kept1, rkept)
    np.testing.assert_allclose(stats1[:3], rstats[:3], rtol=1e-13)
    rn = oracle.normals(pts, 30, 0.01)
    ridx, rd2, rcnt = oracle.knn(pts, None, 8)
    seen = np.zeros(len(pts), dtype=int)
    for r in range(world):
        res = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(res["kept"], rkept)
        assert np.array_equal(res["stats"], np.array(stats1))  # same bits as one process
        assert np.abs(res["nrm"] - rn).max() < 1e-6
        assert np.array_equal(res["idx"], ridx[res["owned"]]) and np.array_equal(res["d2"], rd2[res["owned"]])
        assert np.array_equal(res["cnt"], rcnt[res["owned"]])
        seen[res["owned"]] += 1
    assert (seen == 1).all()


# ---- slabs: every rank indexes only its own part of the cloud ------------------------------------
def _slab_clouds():
    uni = D.uniform_cube(60000, 5)
    # a cloud that the slabs cannot vouch for: two dense layers and a few stray points in the void between
    # them, whose neighbours lie farther away than any halo sized for the cloud's mean density
    rng = np.random.default_rng(6)
    lay = rng.random((12000, 3))
    lay[:, 2] = lay[:, 2] * 0.1 + np.where(np.arange(12000) % 2 == 0, 0.0, 0.9)
    mid = rng.random((40, 3))
    mid[:, 2] = 0.1 + 0.8 * mid[:, 2]
    stray = np.concatenate([lay, mid]).astype(np.float32).astype(np.float64)
    return {"uniform": uni, "stray": stray, "flat": D.lidar_frame(4)[:20000]}


def _check_slab_results(pts, res, k, oracle, want_slab):
    """Every rank's rows against the whole-cloud oracle; the ranks' own rows partition the cloud."""
    from point_cloud_processing_amd import distributed as PD

    host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    ridx, rd2, rcnt = oracle.knn(pts, None, k)
    ravg, rn = oracle.sor_avg(pts, k), oracle.normals(pts, k)
    rkept, _, rstats = oracle.sor(pts, k, 2.0)
    seen = np.zeros(len(pts), dtype=int)
    first = None
    for sh, kept, stats in res:
        assert ("sub_n" in sh) == want_slab
        owned = host(sh["owned"])
        own = owned >= 0
        o = owned[own]
        assert np.array_equal(host(sh["idx"])[own], ridx[o]) and np.array_equal(host(sh["d2"])[own], rd2[o])
        assert np.array_equal(host(sh["cnt"])[own], rcnt[o])
        np.testing.assert_allclose(host(sh["avg"])[own], ravg[o], rtol=1e-13)
        assert len(o) == 0 or np.abs(host(sh["normals"])[own] - rn[o]).max() < 1e-6
        assert np.array_equal(host(kept), rkept)
        np.testing.assert_allclose(stats[:3], rstats[:3], rtol=1e-12)
        first = first or stats
        assert stats == first  # the same bits on every rank
        seen[o] += 1
    assert (seen == 1).all()
    return first


@pytest.mark.parametrize("name", ["uniform", "stray", "flat"])
def test_slab_pipeline_matches_whole_cloud(oracle, name):
    """pipeline_steps_slab with the oracle engine for several pretended worlds: exact rows whether the slabs
    vouch for every answer ("uniform"), have the answers of a few stray points repeated across the ranks
    ("stray") or do not pay at all ("flat": a cloud without depth along z)."""
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    pts = _slab_clouds()[name]
    k = 8
    stats = []
    for world in (1, 2, 4):
        res = PD.run_emulated(lambda r: PD.pipeline_steps_slab(OracleEngine(), pts, k, 2.0, {}, r, world), world)
        slabbed = PD.slab_cuts(OracleEngine(), pts, {}, k, world) is not None
        assert slabbed == (name == "uniform" or world == 1 or (name == "stray" and world == 2))
        stats.append(_check_slab_results(pts, res, k, oracle, want_slab=slabbed))
        if name == "stray" and world == 2:  # the strays' answers were repeated across the ranks, not the whole job
            assert all(sh.get("n_repeated", 0) > 0 for sh, _, _ in res)
    # exact integer sums: whichever route a world took -- slabs, slabs with repeated answers, the replicated index --
    # the statistics carry the same bits
    assert stats[0] == stats[1] == stats[2]


def test_slab_cuts_partition_and_balance(oracle):
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    pts = _slab_clouds()["uniform"]
    cuts, halo = PD.slab_cuts(OracleEngine(), pts, {}, 8, 4)
    assert cuts[0] == -np.inf and cuts[-1] == np.inf and (np.diff(cuts[1:-1]) > 0).all() and halo > 0
    counts = [int(((pts[:, 2] >= cuts[r]) & (pts[:, 2] < cuts[r + 1])).sum()) for r in range(4)]
    assert sum(counts) == len(pts) and max(counts) < 1.1 * len(pts) / 4
    assert PD.slab_cuts(OracleEngine(), pts, {}, 16, 8) is None  # intervals thinner than their halos
    assert PD.slab_cuts(OracleEngine(), _slab_clouds()["flat"], {}, 16, 8) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uniform", "uniform_fp64", "stray", "planes", "outliers"])
def test_slab_pipeline_on_the_hip_engine(hip, oracle, name):
    """The same with the HIP engine (slab subsets, their own indexes, pc_slab_finish) for 1, 2, 3 and 8
    pretended ranks on this GPU."""
    from point_cloud_processing_amd import distributed as PD

    pts = {"planes": lambda: D.planar_patches(300000, 7, extent=20.0, n_patches=12),
           "uniform": lambda: D.uniform_cube(400000, 5), "stray": lambda: _slab_clouds()["stray"],
           "uniform_fp64": lambda: D.uniform_cube(300000, 15, fp32=False) * 3.0 + 1000.0,  # (fp64 records)
           # a scan with outliers: a few dozen points far outside the bulk (what the statistical filter is for)
           "outliers": lambda: np.concatenate([D.uniform_cube(400000, 8), np.random.default_rng(9).random((60, 3)) * 3.0 - 1.0]
                                              ).astype(np.float32).astype(np.float64)}[name]()
    k = 8 if name in ("uniform", "uniform_fp64", "outliers") else 12
    stats = []
    for world in (1, 2, 3, 8):
        cache = {}
        res = PD.run_emulated(lambda r: PD.pipeline_steps_slab(hip, pts, k, 2.0, cache, r, world), world)
        plan = PD.slab_cuts(hip, pts, {}, k, world)
        slabbed = "sub_n" in res[0][0]
        assert slabbed == (plan is not None)
        stats.append(_check_slab_results(pts, res, k, oracle, want_slab=slabbed))
        if (name == "stray" and world == 2) or (name == "outliers" and world > 1):
            # the strays' answers were repeated across the ranks, not the whole job
            assert slabbed and all(sh.get("n_repeated", 0) > 0 for sh, _, _ in res)
    if name == "uniform":
        assert PD.slab_cuts(hip, pts, {}, k, 8) is not None
    assert all(s == stats[0] for s in stats)  # slabs or fallback, any world: the same bits
    single = hip.sor_stats(hip.sor_avg_dev(pts, k, {}), 2.0)
    assert tuple(single) == tuple(stats[0])   # ... and the one-GPU call's


def _one_sided_strays():
    """A uniform cloud with a few points far off to the side near its top: only the TOP slab meets answers it
    cannot vouch for (their neighbours lie farther away than its open lower face)."""
    rng = np.random.default_rng(12)
    side = np.stack([4.0 + rng.random(5), rng.random(5), 0.93 + 0.05 * rng.random(5)], axis=1)
    return np.concatenate([D.uniform_cube(30000, 11), side]).astype(np.float32).astype(np.float64)


def _check_knn_slab(pts, res, k, oracle):
    host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    ridx, rd2, rcnt = oracle.knn(pts, None, k)
    seen = np.zeros(len(pts), dtype=int)
    for sh in res:
        owned = host(sh["owned"])
        own = owned >= 0
        o = owned[own]
        assert np.array_equal(host(sh["idx"])[own], ridx[o]) and np.array_equal(host(sh["d2"])[own], rd2[o])
        assert np.array_equal(host(sh["cnt"])[own], rcnt[o])
        seen[o] += 1
    assert (seen == 1).all()  # every point answered exactly once, whatever route the ranks agreed on


@pytest.mark.parametrize("world", [2, 3])
def test_knn_slab_ranks_decide_together(oracle, world):
    """Only one rank has answers it cannot vouch for: all ranks still take the same branch (the repeat across
    the ranks), and the ranks' own rows cover the cloud exactly once."""
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    pts, k = _one_sided_strays(), 8
    eng = OracleEngine()
    cuts, halo = PD.slab_cuts(eng, pts, {}, k, world)
    unc = [eng.slab_search(pts, {}, k, -1.0, cuts[r], cuts[r + 1], halo)["n_uncertain"] for r in range(world)]
    assert unc[-1] > 0 and not any(unc[:-1])  # the premise: one rank only
    res = PD.run_emulated(lambda r: PD.knn_slab_steps(OracleEngine(), pts, k, {}, r, world), world)
    assert all(sh is not None and sh.get("n_repeated", 0) == unc[-1] for sh in res)
    _check_knn_slab(pts, res, k, oracle)


def test_knn_slab_too_many_unvouched_sends_every_rank_to_the_replicated_index(oracle, monkeypatch):
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    pts, k, world = _one_sided_strays(), 8, 2
    monkeypatch.setattr(PD, "SLAB_REPEAT_MAX", 2)  # (fewer than the five strays)
    res = PD.run_emulated(lambda r: PD.knn_slab_steps(OracleEngine(), pts, k, {}, r, world), world)
    assert res == [None, None]  # all ranks alike: never one rank's slab rows beside another's slice
    eng = OracleEngine()
    res = [eng.shard_search(pts, {}, k, -1.0, *PD.shard_positions(len(pts), r, world), want_rows=True) for r in range(world)]
    _check_knn_slab(pts, res, k, oracle)


@pytest.mark.gpu
def test_knn_slab_ranks_decide_together_on_the_hip_engine(hip, oracle):
    from point_cloud_processing_amd import distributed as PD

    rng = np.random.default_rng(12)
    side = np.stack([4.0 + rng.random(5), rng.random(5), 0.93 + 0.05 * rng.random(5)], axis=1)
    pts = np.concatenate([D.uniform_cube(300000, 11), side]).astype(np.float32).astype(np.float64)
    for world in (2, 3):
        cache = {}
        res = PD.run_emulated(lambda r: PD.knn_slab_steps(hip, pts, 8, cache, r, world), world)
        assert all(sh is not None and sh.get("n_repeated", 0) > 0 for sh in res)
        _check_knn_slab(pts, res, 8, oracle)


def test_exact_sums_do_not_depend_on_order_or_cut(oracle):
    """The statistical filter's integer sums: any permutation and any cut of the rows give the same words' value,
    and that value is the exactly rounded sum."""
    from fractions import Fraction

    from oracle.engine import OracleEngine
    from point_cloud_processing_amd.distributed import exact_value

    rng = np.random.default_rng(3)
    v = np.concatenate([rng.random(5000) * 10.0 ** rng.integers(-12, 6, 5000), [0.0, -1.0, 5e-324, 2.0 ** -130, 2.0 ** 95, 1e-300]])
    eng = OracleEngine()
    whole = exact_value(eng.exact_sums(v, None, 0, 0.0))
    assert whole[1] == int((v >= 0).sum()) and whole[2] == 0
    want = sum(int(Fraction(float(x)) * 2 ** 128) for x in v if x > 0)  # floor(x * 2^128), term by term
    assert whole[0] == want
    perm = rng.permutation(len(v))
    for cuts in ([0, len(v)], [0, 17, 4000, len(v)]):
        words = sum(eng.exact_sums(v[perm][a:b], None, 0, 0.0) for a, b in zip(cuts[:-1], cuts[1:]))
        assert exact_value(words) == whole
    owned = np.where(np.arange(len(v)) % 3 == 0, -1, np.arange(len(v)))
    part = exact_value(eng.exact_sums(v, owned, 1, 0.5))
    d = v[(v > 0) & (owned >= 0)] - 0.5
    assert part[0] == sum(int(Fraction(float(x)) * 2 ** 128) for x in d * d)
    assert exact_value(eng.exact_sums(np.array([1.0, 2.0 ** 96, np.inf]), None, 0, 0.0))[2] == 2


@pytest.mark.gpu
def test_exact_sums_on_the_device_match_the_restatement(hip, oracle):
    import torch
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd.distributed import exact_value

    rng = np.random.default_rng(4)
    v = np.concatenate([rng.random(300000) * 10.0 ** rng.integers(-12, 6, 300000), [0.0, -1.0, 5e-324, 2.0 ** -130, 2.0 ** 95, 1e-300]])
    owned = np.where(rng.random(len(v)) < 0.2, -1, np.arange(len(v))).astype(np.int32)
    dv, do = torch.from_numpy(v).to(hip.device), torch.from_numpy(owned).to(hip.device)
    eng = OracleEngine()
    for what, mean in ((0, 0.0), (1, 0.37)):
        for own_h, own_d in ((None, None), (owned, do)):
            got = exact_value(hip.exact_sums(dv, own_d, what, mean).cpu().numpy())
            assert got == exact_value(eng.exact_sums(v, own_h, what, mean))
    big = torch.tensor([1.0, 2.0 ** 96, float("inf"), float("nan")], dtype=torch.float64, device=hip.device)
    assert exact_value(hip.exact_sums(big, None, 0, 0.0).cpu().numpy())[2] == 2  # (nan is neither > 0 nor >= 0)


# ---- the voxel pass sharded by z layers of voxels ----------------------------------------------------------------
def _raw_clouds():
    rng = np.random.default_rng(21)
    box = (rng.random((150000, 3)) * [1.0, 1.0, 2.0]).astype(np.float32).astype(np.float64)
    # a cloud of few voxel layers along z: the layers cannot be dealt out
    thin = (rng.random((40000, 3)) * [1.0, 1.0, 0.05]).astype(np.float32).astype(np.float64)
    # "coarse": far fewer voxels than points, so the halo guessed before the counts are known is too thin
    return {"box": (box, 0.02), "coarse": (box, 0.05), "thin": (thin, 0.02), "stray": (_slab_clouds()["stray"], 0.01)}


def _check_voxel_slab(raw, voxel, res, k, oracle, want_layers):
    """Every rank's rows against the oracle's pipeline on the whole down-sampled cloud."""
    host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    vox = oracle.voxel_downsample(raw, voxel)[0]
    ridx, rd2, rcnt = oracle.knn(vox, None, k)
    ravg = oracle.sor_avg(vox, k)
    rkept, _, rstats = oracle.sor(vox, k, 2.0)
    seen = np.zeros(len(vox), dtype=int)
    first = None
    for sh, kept, stats, m in res:
        assert m == len(vox)
        owned = host(sh["owned"])
        own = owned >= 0
        o = owned[own]
        assert np.array_equal(host(sh["idx"])[own], ridx[o]) and np.array_equal(host(sh["d2"])[own], rd2[o])
        assert np.array_equal(host(sh["cnt"])[own], rcnt[o])
        np.testing.assert_allclose(host(sh["avg"])[own], ravg[o], rtol=1e-13)
        assert np.array_equal(host(kept), rkept)
        np.testing.assert_allclose(stats[:3], rstats[:3], rtol=1e-12)
        first = first or stats
        assert stats == first
        seen[o] += 1
    assert (seen == 1).all()
    return first


@pytest.mark.parametrize("name", ["box", "coarse", "thin", "stray"])
def test_voxel_layers_pipeline_matches_whole_cloud(oracle, name):
    """voxel_pipeline_steps_slab with the oracle engine: the ranks voxelise their own z layers of voxels (+ halo layers)
    and find, row for row, what the whole down-sampled cloud gives -- whether the layers can be dealt out ("box"),
    cannot ("thin": the replicated voxel pass) or leave answers no slab vouches for ("stray": replicated too)."""
    from oracle.engine import OracleEngine
    from point_cloud_processing_amd import distributed as PD

    raw, voxel = _raw_clouds()[name]
    k = 8
    stats = []
    for world in (1, 2, 3):
        res = PD.run_emulated(lambda r: PD.voxel_pipeline_steps_slab(OracleEngine(), raw, voxel, k, 2.0, {}, r, world), world)
        stats.append(_check_voxel_slab(raw, voxel, res, k, oracle, None))
        if name in ("box", "coarse") and world > 1:  # the ranks indexed their own layers + halo, not the whole cloud
            assert all("sub_n" in sh and sh["sub_n"] < 0.8 * m for sh, _, _, m in res)
    assert stats[0] == stats[1] == stats[2]
    plan = PD.voxel_layer_cuts(OracleEngine(), raw, {}, voxel, 3)
    assert (plan is None) == (name == "thin")
    if plan is not None:
        cuts = plan[3]
        assert cuts[0] == 0 and (np.diff(cuts) > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box", "coarse", "thin", "stray"])
def test_voxel_layers_pipeline_on_the_hip_engine(hip, oracle, name):
    """The same with the HIP engine (pc_voxel_downsample_layers) for 1, 2, 3 and 8 pretended ranks, and the layers of
    the whole voxel grid against pc_voxel_downsample's output."""
    import torch
    from point_cloud_processing_amd import distributed as PD

    raw, voxel = _raw_clouds()[name]
    whole = hip.voxel_downsample_dev(raw, None, None, voxel, {})[0]
    mn, mx = hip.bbox(raw, {})
    parts = [hip.voxel_layers(raw, {}, voxel, mn, mx, a, b)[0] for a, b in ((0, 3), (3, 10), (10, 1 << 30))]
    assert torch.equal(torch.cat(parts), whole)
    stats = []
    for world in (1, 2, 3, 8):
        cache = {}
        res = PD.run_emulated(lambda r: PD.voxel_pipeline_steps_slab(hip, raw, voxel, 8, 2.0, cache, r, world), world)
        stats.append(_check_voxel_slab(raw, voxel, res, 8, oracle, None))
    assert all(s == stats[0] for s in stats)
