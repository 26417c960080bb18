"""pc_sor_select_static (the entry point that allocates nothing) against pc_sor_select, through the C ABI.

Two calls in a row on one GPU, the second with a smaller input: the static scratch array keeps the first
call's contents, so the second call proves that what must start at zero is cleared.  Bit-exact: the mask,
the kept indices and their number are integers."""
import ctypes as C

import numpy as np
import pytest
import torch

from point_cloud_processing_amd import _lib

pytestmark = pytest.mark.gpu


def _select(hip, fn, avg, thr, want_mask):
    n = avg.numel()
    mask = torch.full((max(n, 1),), 7, dtype=torch.uint8, device=hip.device) if want_mask else None
    kept = torch.full((max(n, 1),), -7, dtype=torch.int32, device=hip.device)
    nk = C.c_int64(-1)
    rc = fn(hip.ctx, C.c_void_p(avg.data_ptr()), n, float(thr), None if mask is None else C.c_void_p(mask.data_ptr()),
            C.c_void_p(kept.data_ptr()), C.byref(nk), hip.stream())
    _lib.check(hip.lib, hip.ctx, rc, "select")
    return (None if mask is None else mask[:n].cpu().numpy()), kept[: nk.value].cpu().numpy(), nk.value


@pytest.mark.parametrize("want_mask", [False, True])
def test_static_select_equals_pooled_select_twice(hip, want_mask):
    rng = np.random.default_rng(11)
    for n, thr in [(1_500_000, 0.7), (70_001, 0.35), (1, 0.5), (0, 0.5)]:  # big first, then smaller ones
        a = rng.random(n)
        a[rng.random(n) < 0.05] = -1.0  # "nothing found" markers
        a[rng.random(n) < 0.02] = 0.0   # all-duplicate neighbourhoods
        avg = torch.from_numpy(a).to(hip.device)
        m0, k0, c0 = _select(hip, hip.lib.pc_sor_select, avg, thr, want_mask)
        m1, k1, c1 = _select(hip, hip.lib.pc_sor_select_static, avg, thr, want_mask)
        ref = np.nonzero((a > 0) & (a < thr))[0]
        assert c0 == c1 == len(ref)
        assert np.array_equal(k0, ref) and np.array_equal(k1, ref)
        if want_mask:
            assert np.array_equal(m0, m1) and np.array_equal(np.nonzero(m1)[0], ref)


def test_static_select_rejects_sizes_beyond_its_scratch(hip):
    n = _lib.PC_SOR_SELECT_STATIC_MAX_N + 1
    avg = torch.zeros((n,), dtype=torch.float64, device=hip.device)
    kept = torch.empty((n,), dtype=torch.int32, device=hip.device)
    nk = C.c_int64(0)
    rc = hip.lib.pc_sor_select_static(hip.ctx, C.c_void_p(avg.data_ptr()), n, 1.0, None, C.c_void_p(kept.data_ptr()),
                                      C.byref(nk), hip.stream())
    assert rc == -1
    assert b"PC_SOR_SELECT_STATIC_MAX_N" in hip.lib.pc_last_error(hip.ctx)
    # and the largest accepted size works
    n = _lib.PC_SOR_SELECT_STATIC_MAX_N
    a = np.linspace(-0.1, 1.0, n)
    avg = torch.from_numpy(a).to(hip.device)
    _, k1, c1 = _select(hip, hip.lib.pc_sor_select_static, avg, 0.5, False)
    ref = np.nonzero((a > 0) & (a < 0.5))[0]
    assert c1 == len(ref) and np.array_equal(k1, ref)
