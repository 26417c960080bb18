"""The C-ABI library loads on a CPU-only machine and exports exactly what include/pc_amd.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from point_cloud_processing_amd import _lib

    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.pc_abi_version() == 1
    assert lib.pc_status_string(-4).decode() == "voxel_size is too small"


def test_no_gpu_means_loud_failure():
    import torch
    import pytest

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from point_cloud_processing_amd.hip_engine import HipEngine

    with pytest.raises(RuntimeError, match="no AMD GPU"):
        HipEngine()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "point_cloud_processing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
