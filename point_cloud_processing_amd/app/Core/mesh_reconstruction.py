"""The neighbourhood-stage part of the reference's ``reconstruct_bpa``: the ball radii.

reference app/Core/mesh_reconstruction.py:110-117 estimates the Ball-Pivoting radii from
the mean nearest-neighbour distance (kNN with k = 2 over the whole cloud).  Ball pivoting
itself (:120-129) is sequential surface reconstruction inside Open3D and stays there.
"""
from __future__ import annotations

from typing import List

import numpy as np

from ...geometry import wrap


def estimate_bpa_radii(pcd) -> List[float]:
    """``[b, 2b, 3b]`` with ``b = 1.5 * mean nearest-neighbour distance`` (:112-117)."""
    print("[BPA] Estimating radii from nearest neighbor distances ...")
    dists = np.asarray(wrap(pcd).compute_nearest_neighbor_distance())
    mean_dist = dists.mean()
    print(f"[BPA] mean NN distance = {mean_dist}")
    base = mean_dist * 1.5
    return [base, base * 2.0, base * 3.0]
