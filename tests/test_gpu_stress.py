"""Randomised cross-checks (tools/stress_lds.py, tools/stress_gather.py) in a short form: random generators, sizes that are
not multiples of 16, random widths, random placements / hot thresholds of the LDS-resident walk, every gather walk - the
LDS-resident SpMM against the per-window walk and A @ 1 = degree; the edge-valued SpMM, the SDDMM and the fused AGNN pair against
fp64 evaluations within the operand-rounding bound."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool, cases, seed, last", [("stress_lds.py", 8, 21, "cases agree"), ("stress_gather.py", 8, 22, "cases agree")])
def test_randomised_cross_checks(tool, cases, seed, last):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, PYTHONWARNINGS="ignore"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert last in r.stdout.splitlines()[-1], r.stdout[-500:]
