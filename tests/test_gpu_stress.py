"""Randomised sweep: cloud shapes, sizes, k, radii, external queries, fp32/fp64 storage and the fused
consumers, every case bit-exact against the CPU oracle (the sweep itself lives in scratch/stress.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sweep_matches_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "stress.py"), "60", str(seed)],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
