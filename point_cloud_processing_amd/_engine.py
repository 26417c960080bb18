"""The active compute engine behind :class:`geometry.PointCloud`.

Production has exactly one engine: the HIP library (``hip_engine.HipEngine``).  It is created
on first use and the import fails loudly when the shared library or the GPU is missing; there
is no CPU fallback.  ``set_engine`` exists so that test code can inject a checker engine.
"""
from __future__ import annotations

import threading

_active = None
_lock = threading.Lock()  # the app's worker threads may ask for the engine at the same moment


def get():
    global _active
    if _active is None:
        with _lock:
            if _active is None:
                from .hip_engine import HipEngine  # raises if the HIP library / device is absent

                _active = HipEngine()
    return _active


def set_engine(engine) -> None:
    global _active
    _active = engine


def current():
    return _active
