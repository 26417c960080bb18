import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")
    config.addinivalue_line("markers", "open3d: side-by-side run against a real open3d install, if importable")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pc_oracle

    pc_oracle.build()
    return pc_oracle


@pytest.fixture()
def oracle_engine(oracle):
    """Route PointCloud methods to the CPU oracle for the duration of one test."""
    import point_cloud_processing_amd as P
    from point_cloud_processing_amd import _engine
    from oracle.engine import OracleEngine

    prev = _engine.current()
    P.set_engine(OracleEngine())
    yield
    P.set_engine(prev)


@pytest.fixture(scope="session")
def hip():
    """The production engine on cuda:0 (GPU tests only)."""
    from point_cloud_processing_amd.hip_engine import HipEngine

    return HipEngine(0)


@pytest.fixture()
def hip_engine(hip):
    import point_cloud_processing_amd as P
    from point_cloud_processing_amd import _engine

    prev = _engine.current()
    P.set_engine(hip)
    yield hip
    P.set_engine(prev)
