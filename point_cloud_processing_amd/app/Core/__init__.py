"""Names the reference's callers import from ``app.Core`` that belong to the neighbourhood
stage (reference: app/Core/__init__.py:1-27).  Mesh reconstruction and mesh post-processing
stay with the reference's own Open3D code and are not re-exported here."""
from .pointcloud_preprocess import normalize_point_cloud, preprocess_point_cloud
from .mesh_reconstruction import estimate_bpa_radii

__all__ = ["normalize_point_cloud", "preprocess_point_cloud", "estimate_bpa_radii"]
