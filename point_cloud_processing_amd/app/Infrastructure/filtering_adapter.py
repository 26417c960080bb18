"""Drop-in for the reference's ``app/Infrastructure/filtering_adapter.py``.

``FilterSettings`` keeps the reference's fields and defaults one for one
(filtering_adapter.py:10-34) and ``filter_pcd`` follows its control flow (:55-143): density
estimate from the median nearest-neighbour distance, [voxel] -> statistical -> radius outlier
removal, a relax loop that always restarts from the original cloud, and a last statistical-only
attempt.  Every cloud method used here runs on the device; because the cloud object caches its
per-point mean neighbour distances, the up-to-six statistical passes cost one kNN search.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from ...geometry import PointCloud, unwrap_like, wrap


@dataclass
class FilterSettings:
    # which steps run
    use_stat: bool = True
    use_radius: bool = True
    use_voxel: bool = False

    # statistical outlier removal
    nb_stat: int = 30
    std_ratio: float = 3.0

    # radius outlier removal, radius = radius_mult * median NN distance unless radius_abs is set
    nb_radius: int = 4
    radius_mult: float = 5.0
    radius_abs: Optional[float] = None

    # voxel down-sampling, size = voxel_mult * median NN distance unless voxel_abs is set
    voxel_mult: float = 1.5
    voxel_abs: Optional[float] = None

    # safety net
    max_drop: float = 0.30
    relax_iters: int = 4
    adaptive: bool = True
    debug: bool = True


def _estimate_nn_median(pcd: PointCloud, sample: int = 20000) -> float:
    """Median nearest-neighbour distance of (a random subset of) the cloud (:37-52).

    Like the reference, the subset of a cloud with more than ``sample`` points comes from
    NumPy's global, unseeded generator, so it differs from run to run."""
    n = len(pcd)
    if n < 10:
        return 0.0
    if n > sample:
        p = pcd.select_by_index(_random_subset(n, sample))
    else:
        p = pcd
    d = np.asarray(p.compute_nearest_neighbor_distance(), dtype=np.float64)
    if d.size == 0:
        return 0.0
    return float(np.median(d))


def _random_subset(n: int, size: int) -> np.ndarray:
    """``size`` distinct indices below ``n`` from NumPy's global generator, drawn with the reference's own call
    (``np.random.choice(n, size, replace=False)``, :43): a caller that seeds ``np.random`` gets the reference's
    subset, hence its median spacing and the radius derived from it."""
    return np.random.choice(n, size, replace=False)


class _Attempt:
    """One parameter set of the outlier pipeline and the cloud it produced."""

    __slots__ = ("stat_sr", "rad_nb", "rad_r", "with_radius", "cloud", "drop")

    def __init__(self, stat_sr: float, rad_nb: int, rad_r: float, with_radius: bool):
        self.stat_sr, self.rad_nb, self.rad_r, self.with_radius = stat_sr, rad_nb, rad_r, with_radius
        self.cloud = None
        self.drop = 0.0


def _relaxed(s: FilterSettings, radius: float, it: int) -> _Attempt:
    """Relax step ``it`` (1-based): softer std ratio, wider radius, fewer required points (:113-117)."""
    return _Attempt(s.std_ratio + 0.8 * it, max(2, s.nb_radius - it), radius * (1.35 ** it), True)


def filter_pcd(pcd, settings: Optional[FilterSettings] = None):
    s = settings or FilterSettings()
    n0 = len(pcd.points)
    if n0 == 0:
        return pcd  # the input object itself (:64-66)
    base = wrap(pcd)

    nn_med = _estimate_nn_median(base) if s.adaptive else 0.0
    if nn_med <= 0:
        nn_med = 0.01
    radius = float(s.radius_abs) if s.radius_abs is not None else float(s.radius_mult * nn_med)
    voxel = None
    if s.use_voxel:
        voxel = float(s.voxel_abs) if s.voxel_abs is not None else float(s.voxel_mult * nn_med)

    def run(a: _Attempt) -> _Attempt:
        # every attempt starts again from the original cloud (:104, :119, :127)
        cl = base
        if s.use_voxel and voxel and voxel > 0:
            cl = cl.voxel_down_sample(voxel_size=voxel)
        if s.use_stat:
            cl, _ = cl.remove_statistical_outlier(nb_neighbors=int(s.nb_stat), std_ratio=float(a.stat_sr))
        if a.with_radius and s.use_radius:
            cl, _ = cl.remove_radius_outlier(nb_points=int(a.rad_nb), radius=float(a.rad_r))
        a.cloud = cl
        a.drop = 1.0 - (len(cl) / n0)  # (the row count: no copy of the points)
        return a

    best = run(_Attempt(s.std_ratio, s.nb_radius, radius, True))

    # too much removed: relax step by step, keep an attempt only if it removes less
    it = 0
    while best.drop > s.max_drop and it < s.relax_iters:
        it += 1
        trial = run(_relaxed(s, radius, it))
        if trial.drop < best.drop:
            best = trial

    # still too much: the statistical filter alone, with std ratio at least 3.5 (:126-131)
    radius_on = True
    if best.drop > s.max_drop and s.use_radius:
        trial = run(_Attempt(max(s.std_ratio, 3.5), s.nb_radius, radius, False))
        if trial.drop <= best.drop:
            best = trial
            radius_on = False

    if s.debug:
        print(
            f"[Filter] n0={n0} -> n={len(best.cloud)} (drop={best.drop*100:.1f}%), "
            f"nn_med={nn_med:.5f}, "
            f"stat={'on' if s.use_stat else 'off'}(nb={s.nb_stat}, sr={s.std_ratio}), "
            f"radius={'on' if (s.use_radius and radius_on) else 'off'}"
            f"(nb={s.nb_radius}, mult={s.radius_mult}, r={radius:.5f}), "
            f"voxel={'on' if s.use_voxel else 'off'}({voxel if voxel else 'n/a'}), "
            f"max_drop={s.max_drop}"
        )

    if best.cloud is base:
        return pcd
    return unwrap_like(best.cloud, pcd)
