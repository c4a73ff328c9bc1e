"""The planner of the short-series launch (``small_plan``, csrc/ngp_internal.h) checked on the host
over every geometry it can be asked about (tests/sanitize/plan_check.cpp): what it accepts fits the
kernel's registers, LDS and sweep count and carries every aux row-block exactly once.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


@pytest.mark.skipif(HIPCC is None, reason="no hipcc")
def test_every_accepted_geometry_fits_the_kernel(tmp_path):
    exe = str(tmp_path / "plan_check")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-std=c++17", "-w",
                           os.path.join(ROOT, "tests", "sanitize", "plan_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 failures" in out.stdout and "1024 geometries accepted" in out.stdout, out.stdout[-500:]
