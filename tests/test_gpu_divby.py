"""letkf_divby_dev.h (the limited column search's division through a per-member reciprocal) against the IEEE division
hipcc emits, on the device: 3 x 33.5 M operand pairs -- random over the whole range of the identity claim, ordinary
magnitudes, and the tiny / huge / non-finite numerators the header treats separately."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
def test_reciprocal_quotient_is_the_ieee_quotient(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "divby_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "scale-letkf_amd", "csrc"),
                    os.path.join(HERE, "hip", "divby_check.hip"), "-o", str(exe)], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout.splitlines()[-1]
