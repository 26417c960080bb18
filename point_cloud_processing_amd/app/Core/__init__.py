"""The names the reference's callers import from ``app.Core`` (reference: app/Core/__init__.py:1-27).

``normalize_point_cloud``, ``preprocess_point_cloud`` and ``reconstruct_bpa`` (its radii search) are the
neighbourhood stage and live here.  Poisson reconstruction and the mesh post-processing functions work on
triangle meshes, outside this stage: their names forward, on first call, to the reference's own modules
(``app.Core.mesh_reconstruction`` / ``app.Core.mesh_postprocess`` of the application tree this package is
dropped into) and raise a RuntimeError that says so when that tree is not importable.
"""
from .pointcloud_preprocess import normalize_point_cloud, preprocess_point_cloud
from .mesh_reconstruction import estimate_bpa_radii, reconstruct_bpa

_FORWARDED = {
    "reconstruct_poisson": "mesh_reconstruction",
    "straighten_slab_sides": "mesh_postprocess",
    "fill_small_holes": "mesh_postprocess",
    "fill_all_holes": "mesh_postprocess",
    "remove_long_triangles": "mesh_postprocess",
}


def _forward(name: str, module: str):
    def call(*args, **kwargs):
        import importlib

        last = None
        for root in ("app.Core", "Core"):  # (the reference's two entry points import it either way)
            try:
                mod = importlib.import_module(f"{root}.{module}")
            except ImportError as e:
                last = e
                continue
            if getattr(mod, name, None) is not None and mod.__name__ != __name__:
                return getattr(mod, name)(*args, **kwargs)
        raise RuntimeError(f"{name} is mesh processing outside the GPU neighbourhood stage: it runs the "
                           f"application's own {module}.py (Open3D), which is not importable here") from last

    call.__name__ = name
    call.__doc__ = f"Forwards to the application's own ``{module}.{name}`` (mesh stage, not built here)."
    return call


reconstruct_poisson = _forward("reconstruct_poisson", "mesh_reconstruction")
straighten_slab_sides = _forward("straighten_slab_sides", "mesh_postprocess")
fill_small_holes = _forward("fill_small_holes", "mesh_postprocess")
fill_all_holes = _forward("fill_all_holes", "mesh_postprocess")
remove_long_triangles = _forward("remove_long_triangles", "mesh_postprocess")

__all__ = [
    "normalize_point_cloud",
    "preprocess_point_cloud",
    "reconstruct_poisson",
    "reconstruct_bpa",
    "straighten_slab_sides",
    "fill_small_holes",
    "fill_all_holes",
    "remove_long_triangles",
    "estimate_bpa_radii",
]
