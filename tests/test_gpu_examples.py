"""The example scripts run (small sizes) and their own assertions hold."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,arg", [("uniaxial_tension_resident.py", "2000"), ("two_materials_resident.py", "3000"),
                                        ("uniaxial_tension_multi_gpu.py", "30000"), ("cube_tension_fe.py", "5"), ("many_materials_device.py", "3000")])
def test_example_runs(script, arg):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), arg], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Newton iterations" in r.stdout
