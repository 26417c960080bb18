"""MI355X-native neighbourhood stage for the "Laser" point-cloud app.

Voxel-grid down-sampling, exact k-nearest-neighbour search, statistical / radius outlier
removal and PCA normal estimation as hand-written HIP kernels for gfx950 behind a C ABI
(``csrc/``), plus the host-side mirror of the reference's ``app/Core`` and
``app/Infrastructure`` entry points (``app/``).
"""
from .geometry import (  # noqa: F401
    AxisAlignedBoundingBox,
    KDTreeSearchParamHybrid,
    KDTreeSearchParamKNN,
    KDTreeSearchParamRadius,
    PointCloud,
    Vector3dVector,
)
from ._engine import get as get_engine, set_engine  # noqa: F401
from .registration import RegistrationResult, registration_icp  # noqa: F401

__all__ = [
    "PointCloud",
    "Vector3dVector",
    "AxisAlignedBoundingBox",
    "KDTreeSearchParamKNN",
    "KDTreeSearchParamRadius",
    "KDTreeSearchParamHybrid",
    "RegistrationResult",
    "registration_icp",
    "get_engine",
    "set_engine",
]
