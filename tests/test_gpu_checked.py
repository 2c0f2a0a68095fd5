"""The CHECKED twin of the library (scale-letkf_amd/Makefile CHECKED=1, csrc/letkf_wave.hip LETKF_CHECK): the column-survivor mode
of the loop-body kernel with every device-derived index tested against the host's buffer sizes.  Round 3 lost a test process to an
abort inside letkf_das_columns_dev whose message pytest's capture swallowed; this is the run that says WHICH bound
a launch violates instead of dying on it.  The twin is loaded in a process of its own (one library per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_column_survivor_mode_touches_no_bound_in_the_checked_build():
    lib = os.path.join(ROOT, "scale-letkf_amd", "lib", "libletkf_amd_checked.so")
    # (make decides: the twin is rebuilt whenever a source is newer than it -- a stale twin lacks the entries the binding checks for)
    subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "scale-letkf_amd"), "CHECKED=1"], stdout=2)
    env = dict(os.environ, LETKF_AMD_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_checked_run.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("checked ok") == 6, r.stdout
