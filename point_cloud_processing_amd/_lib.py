"""ctypes binding of libpc_amd.so (the C ABI declared in include/pc_amd.h).

Loading fails loudly: there is no CPU fallback behind this module.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PC_AMD_LIB") or os.path.join(_HERE, "libpc_amd.so")  # (override: A/B runs of two builds)

PC_F32, PC_F64 = 0, 1
PC_MAX_K = 64
PC_SOR_CHUNK = 4096
PC_SOR_EXACT_WORDS = 10
PC_SOR_SELECT_STATIC_MAX_N = 4 * 1024 * 1024

# every symbol include/pc_amd.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pd = C.POINTER(C.c_double)
_pi64 = C.POINTER(C.c_int64)
_pi32 = C.POINTER(C.c_int32)
SYMBOLS = [
    ("pc_create", _i32, [_i32, C.POINTER(_vp)]),
    ("pc_destroy", None, [_vp]),
    ("pc_last_error", C.c_char_p, [_vp]),
    ("pc_status_string", C.c_char_p, [_i32]),
    ("pc_abi_version", _i32, []),
    ("pc_host_to_f32_exact", _i32, [_vp, _i64, _vp]),
    ("pc_bbox", _i32, [_vp, _vp, _i32, _i64, _pd, _pd, _vp]),
    ("pc_center", _i32, [_vp, _vp, _i32, _i64, _pd, _vp]),
    ("pc_affine", _i32, [_vp, _vp, _i64, _pd, _i32, _dbl, _pd, _vp]),
    ("pc_voxel_downsample", _i32, [_vp, _vp, _i32, _vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp, _vp, _pi64, _vp]),
    ("pc_voxel_downsample_layers", _i32, [_vp, _vp, _i32, _vp, _vp, _i64, _dbl, _pd, _pd, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _pi64, _vp]),
    ("pc_voxel_assign", _i32, [_vp, _vp, _i32, _i64, _dbl, _vp, _pi32, _vp]),
    ("pc_grid_build", _i32, [_vp, _vp, _i32, _i64, _dbl, _i32, C.POINTER(_vp), _vp]),
    ("pc_grid_destroy", None, [_vp, _vp]),
    ("pc_grid_info", _i32, [_vp, _pd]),
    ("pc_grid_perm_dev", _vp, [_vp]),
    ("pc_grid_perm_copy", _i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    ("pc_knn", _i32, [_vp, _vp, _vp, _i32, _i64, _i32, _dbl, _vp, _vp, _i32, _vp, _vp]),
    ("pc_knn_range", _i32, [_vp, _vp, _i64, _i64, _i32, _dbl, _vp, _vp, _i32, _vp, _vp]),
    ("pc_sor_avg", _i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    ("pc_knn_fused", _i32, [_vp, _vp, _i64, _i64, _i32, _dbl, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("pc_knn_fused_compact", _i32, [_vp, _vp, _i64, _i64, _i32, _dbl, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("pc_sor_chunks", _i64, [_i64]),
    ("pc_sor_chunk_sums", _i32, [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _dbl, _vp, _vp]),
    ("pc_sor_chunk_finish", _i32, [_vp, _vp, _i64, _pd, _vp]),
    ("pc_mask_bytes", _i64, [_i64]),
    ("pc_sor_mask_scatter", _i32, [_vp, _vp, _vp, _i64, _dbl, _vp, _vp]),
    ("pc_mask_select", _i32, [_vp, _vp, _i64, _vp, _pi64, _vp]),
    ("pc_sor_exact_sums", _i32, [_vp, _vp, _vp, _i64, _i32, _dbl, _vp, _vp]),
    ("pc_sor_sum", _i32, [_vp, _vp, _i64, _pd, _pi64, _vp]),
    ("pc_sor_sqsum", _i32, [_vp, _vp, _i64, _dbl, _pd, _vp]),
    ("pc_sor_select", _i32, [_vp, _vp, _i64, _dbl, _vp, _vp, _pi64, _vp]),
    ("pc_sor_select_static", _i32, [_vp, _vp, _i64, _dbl, _vp, _vp, _pi64, _vp]),
    ("pc_normals", _i32, [_vp, _vp, _i64, _i64, _i32, _dbl, _vp, _vp, _vp]),
    ("pc_radius_count", _i32, [_vp, _vp, _vp, _i32, _i64, _i64, _i64, _dbl, _vp, _vp]),
    ("pc_nn_distance", _i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    ("pc_gather_rows", _i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    ("pc_scatter_rows", _i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    ("pc_gather_rows_f32", _i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    ("pc_count_select", _i32, [_vp, _vp, _i64, _i32, _vp, _vp, _pi64, _vp]),
    ("pc_axis_histogram", _i32, [_vp, _vp, _i32, _i64, _i32, _pd, _pd, _i32, _vp, _vp]),
    ("pc_select_slab", _i32, [_vp, _vp, _i32, _i64, _i32, _dbl, _dbl, _vp, _pi64, _vp]),
    ("pc_grid_axis_range", _i32, [_vp, _vp, _dbl, _dbl, _pi64, _pi64, _vp]),
    ("pc_grid_relabel", _i32, [_vp, _vp, _vp, _i64, _vp]),
    ("pc_slab_finish", _i32, [_vp, _vp, _i32, _vp, _i64, _i32, _dbl, _vp, _vp, _i32, _vp, _vp, _dbl, _dbl, _dbl, _dbl,
                              _vp, _vp, _vp, _pi64, _pi64, _vp]),
    ("pc_merge_rows", _i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("pc_rows_consumers", _i32, [_vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("pc_orient_normals", _i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    ("pc_icp_point_to_plane_terms", _i32, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i32, _dbl, _vp, _vp]),
    ("pc_transform", _i32, [_vp, _vp, _i64, _vp, _vp]),
]

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises OSError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class PcError(RuntimeError):
    pass


def check(lib, ctx, rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = lib.pc_last_error(ctx).decode("utf-8", "replace") if ctx else lib.pc_status_string(rc).decode()
    if rc == -1:
        # Open3D raises RuntimeError for illegal parameters; keep that type, carry the message
        raise PcError(f"[pc_amd] {what}: {msg}")
    if rc == -4:
        raise PcError(f"[pc_amd] {what}: voxel_size is too small.")
    if rc == -2:
        raise MemoryError(f"[pc_amd] {what}: {msg}")
    raise PcError(f"[pc_amd] {what}: {msg} (status {rc})")
