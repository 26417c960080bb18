"""Point-to-plane ICP around the k = 1 correspondence search.

The reference's GUI offers ICP (app/UI/main_window.py:58, 180-194) but ships no solver; this module gives
the call a body with the semantics of Open3D's ``registration_icp`` +
``TransformationEstimationPointToPlane``: correspondences = nearest target point within
``max_correspondence_distance`` (strict), one Gauss-Newton step per iteration from the 6 x 6 normal
equations, ``fitness`` = matched share of the source, ``inlier_rmse`` over the matched pairs, stop when both
change by less than the relative thresholds.  Control flow and the 6 x 6 solve run on the host; the search,
the reduction of the normal equations and the rigid transform of the source run on the engine.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _engine
from .geometry import wrap


@dataclass
class RegistrationResult:
    transformation: np.ndarray = field(default_factory=lambda: np.eye(4))
    fitness: float = 0.0
    inlier_rmse: float = 0.0
    correspondence_set: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.int32))
    iterations: int = 0


def transform_from_vector6(x: np.ndarray) -> np.ndarray:
    """(alpha, beta, gamma, tx, ty, tz) -> 4 x 4 with R = Rz(gamma) Ry(beta) Rx(alpha)
    (Open3D's TransformVector6dToMatrix4d)."""
    ca, sa, cb, sb, cg, sg = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    t = np.eye(4)
    t[:3, :3] = rz @ ry @ rx
    t[:3, 3] = x[3:6]
    return t


def step_from_terms(terms: np.ndarray) -> np.ndarray:
    """The rigid update solving (sum J J^T) x = -(sum J r); identity when there is nothing to solve."""
    if terms[28] <= 0:
        return np.eye(4)
    jtj = np.zeros((6, 6))
    jtj[np.triu_indices(6)] = terms[:21]
    jtj = jtj + np.triu(jtj, 1).T
    try:
        x = np.linalg.solve(jtj, -terms[21:27])
    except np.linalg.LinAlgError:
        return np.eye(4)
    return transform_from_vector6(x) if np.all(np.isfinite(x)) else np.eye(4)


def _quality(terms, m):
    cnt = terms[28]
    return (cnt / m if m else 0.0), (float(np.sqrt(terms[27] / cnt)) if cnt > 0 else 0.0)


def registration_icp(source, target, max_correspondence_distance: float, init=None, max_iteration: int = 30,
                     relative_fitness: float = 1e-6, relative_rmse: float = 1e-6) -> RegistrationResult:
    src, tgt = wrap(source), wrap(target)
    if not (max_correspondence_distance > 0.0):
        raise RuntimeError("[Open3D Error] Invalid max_correspondence_distance.")
    if not tgt.has_normals():
        raise RuntimeError("[Open3D Error] TransformationEstimationPointToPlane requires target normals.")
    res = RegistrationResult()
    total = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    m = len(src)
    if m == 0 or len(tgt) == 0:
        res.transformation = total
        return res
    s = _engine.get().icp_session(np.asarray(src.points), np.asarray(tgt.points), np.asarray(tgt.normals))
    s.transform(total)
    s.correspond()
    terms = s.terms(max_correspondence_distance)
    fit, rmse = _quality(terms, m)
    for it in range(int(max_iteration)):
        update = step_from_terms(terms)
        total = update @ total
        s.transform(update)
        s.correspond()
        terms = s.terms(max_correspondence_distance)
        fit2, rmse2 = _quality(terms, m)
        res.iterations = it + 1
        done = abs(fit - fit2) < relative_fitness and abs(rmse - rmse2) < relative_rmse
        fit, rmse = fit2, rmse2
        if done:
            break
    res.transformation, res.fitness, res.inlier_rmse = total, float(fit), float(rmse)
    res.correspondence_set = s.correspondence_set(max_correspondence_distance)
    return res
