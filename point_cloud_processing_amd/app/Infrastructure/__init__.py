from .filtering_adapter import FilterSettings, filter_pcd

__all__ = ["FilterSettings", "filter_pcd"]
