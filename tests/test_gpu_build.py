"""The index build's two routes give the same index: the bucketed build of evenly filled fp32 clouds
(csrc/pc_build.hip) against the general build (cell keys, radix sort, gather), record for record."""
import ctypes as C

import numpy as np
import pytest

import datasets as D

pytestmark = pytest.mark.gpu


def _index_of(eng, xyz, n, k):
    """(perm, dims, cell) of the index the engine builds over the device array ``xyz``."""
    import torch
    from point_cloud_processing_amd import _lib
    from point_cloud_processing_amd.hip_engine import _ptr

    h = C.c_void_p()
    eng._check(eng.lib.pc_grid_build(eng.ctx, _ptr(xyz), _lib.PC_F32, n, 0.0, k, C.byref(h), eng.stream()), "grid")
    perm = torch.empty((n,), dtype=torch.int32, device=eng.device)
    eng._check(eng.lib.pc_grid_perm_copy(eng.ctx, h, 0, n, _ptr(perm), eng.stream()), "perm")
    info = (C.c_double * 6)()
    eng.lib.pc_grid_info(h, info)
    idx = torch.empty((n, k), dtype=torch.int32, device=eng.device)
    d2 = torch.empty((n, k), dtype=torch.float64, device=eng.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=eng.device)
    eng._check(eng.lib.pc_knn(eng.ctx, h, None, _lib.PC_F32, n, k, -1.0, _ptr(idx), _ptr(d2), _lib.PC_F64, _ptr(cnt), eng.stream()), "knn")
    out = perm.cpu().numpy(), tuple(info[:4]), idx.cpu().numpy(), d2.cpu().numpy(), cnt.cpu().numpy()
    eng.lib.pc_grid_destroy(eng.ctx, h)
    return out


def _clouds():
    rng = np.random.default_rng(41)
    crowd = D.uniform_cube(300_000, 44)
    crowd[:3000] = crowd[0]  # three thousand copies of one position: one group beyond what a workgroup ranks
    return {
        "uniform_k16": (D.uniform_cube(400_000, 42), 16),            # halved cells
        "uniform_k2": (D.uniform_cube(300_000, 43), 2),              # no half bit, nearly empty cells
        "uniform_odd_size": (D.uniform_cube(262_147, 45), 8),        # a ragged last tile
        "denser_half": (np.concatenate([D.uniform_cube(200_000, 46), D.uniform_cube(200_000, 47) * [0.5, 1, 1]]), 16),
        "crowded_cell": (crowd, 16),                                 # falls back to the general build
        "sphere": (D.bumpy_sphere(300_000, 48), 16),                 # a surface: uneven, general build + levels
        "box_offset": (D.uniform_cube(300_000, 49) * 3.0 + rng.random(3) * 100.0, 12),
    }


@pytest.mark.parametrize("name", list(_clouds()))
def test_bucketed_build_equals_general_build(hip, monkeypatch, name):
    import torch
    from point_cloud_processing_amd.hip_engine import HipEngine

    pts, k = _clouds()[name]
    p32 = np.ascontiguousarray(pts, dtype=np.float32)
    n = len(p32)
    monkeypatch.setenv("PC_AMD_NO_BUCKET_BUILD", "1")
    general = HipEngine(0)  # (the switches are read when a context is created)
    monkeypatch.delenv("PC_AMD_NO_BUCKET_BUILD")
    xyz = torch.from_numpy(p32).to(hip.device)
    a = _index_of(hip, xyz, n, k)
    b = _index_of(general, xyz, n, k)
    assert a[1] == b[1]                         # the same grid
    assert np.array_equal(a[0], b[0])           # the same records in the same order (ties inside a cell by index)
    assert all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))
    assert np.array_equal(np.sort(a[0]), np.arange(n))
    # the array at an address that is no multiple of 16 bytes (a view one row into a larger buffer)
    pad = torch.empty((n + 1, 3), dtype=torch.float32, device=hip.device)
    pad[1:] = xyz
    c = _index_of(hip, pad[1:], n, k)
    assert np.array_equal(c[0], a[0]) and np.array_equal(c[2], a[2])


def test_bucketed_build_against_the_oracle(hip, oracle):
    p = D.uniform_cube(500_000, 50)
    idx, d2, cnt = hip.knn(p, None, 16, -1.0, {})
    ridx, rd2, rcnt = oracle.Grid(p).knn_self(16)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(cnt, rcnt)
