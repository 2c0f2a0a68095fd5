"""The solve kernel's dynamic run scheduling (scale-letkf_amd/csrc/letkf_wave.hip: sched_make_plan on the host,
sched_unit on the device) hands out every run of a launch exactly once -- whole, or as its four quarters -- whatever the
problem size, run direction, run length and grid.  letkf_sched_plan_check builds the plan of such a launch and walks
every hand-out position of every XCD range through the very function the kernel runs (it is compiled for both sides);
no device is needed.  The GPU tests check the other half: that the counters are drawn atomically, i.e. that no point of
an analysis is left unwritten (tests/test_gpu_trivial.py, test_gpu_fullsize.py)."""
import ctypes as C

import numpy as np
import pytest

from __graft_entry__ import load_package


@pytest.fixture(scope="module")
def check():
    p = load_package()
    p.build()
    f = C.CDLL(p.LIB_PATH).letkf_sched_plan_check
    f.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    f.restype = C.c_int
    return f


def test_baseline_shapes(check):
    # (npts, stride, run_len, grid, wave-slots per workgroup, resident wave-slots per XCD)
    shapes = [(240 * 240 * 60, 240 * 240, 60, 512, 4, 256),       # C2, runs up the columns, one-wave points
              (240 * 240 * 60, 1, 16, 512, 4, 256),               # C2, runs along ij
              (48 * 48 * 60, 48 * 48, 60, 512, 1, 64),            # C2-cols-k100, two-wave points
              (48 * 48 * 12, 1, 6, 1152, 4, 256),                 # C2-mini: the small-batch grid (one run per wave)
              (48 * 48 * 12, 1, 16, 512, 1, 64),                  # C2-slab-k100
              (1000 * 1000 * 80 // 8, 1000 * 1000 // 8, 80, 512, 4, 256),   # a C4 tile
              (1000 * 1000 * 80, 1000 * 1000, 80, 512, 4, 256),             # all of C4 in one call
              (100_000_000, 1, 1, 512, 4, 256),                             # 1e8 unrelated points (bundled runs)
              (4096, 1, 1, 512, 4, 256), (1, 1, 1, 1, 4, 256), (0, 1, 1, 1, 4, 256), (7, 1, 1, 512, 1, 64)]
    for s in shapes:
        assert check(*s) == 0, s


def test_random_shapes(check):
    rng = np.random.default_rng(20241004)
    for _ in range(3000):
        ppw = int(rng.choice([1, 4]))
        grid = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 255, 256, 512, 513, 1024, 4096]))
        if rng.random() < 0.5:                                  # strided runs: nlev levels of nij points
            nij, nlev = int(rng.integers(1, 3000)), int(rng.integers(1, 130))
            npts, stride = nij * nlev, nij
            run_len = int(rng.choice([nlev, max(1, nlev // 2), 7, 8, 16, 128]))
            run_len = min(run_len, nlev)
        else:
            npts, stride = int(rng.integers(1, 400000)), 1
            run_len = int(rng.choice([1, 2, 3, 6, 7, 8, 15, 16, 17, 60, 4096]))
        res = 256 if ppw == 4 else 64
        assert check(npts, stride, run_len, grid, ppw, res) == 0, (npts, stride, run_len, grid, ppw)


def test_rejects_nonsense(check):
    assert check(100, 1, 0, 8, 4, 256) != 0 and check(100, 1, 4, 0, 4, 256) != 0 and check(-1, 1, 1, 8, 4, 256) != 0


def test_units_of_three_runs():
    """csrc/letkf_trio.hip (k <= 20) walks three neighbouring runs in step: its plan hands out units that are whole multiples of
    three runs, every run exactly once, no quartered runs."""
    p = load_package()
    f = C.CDLL(p.LIB_PATH).letkf_sched_plan_check_units
    f.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    f.restype = C.c_int
    shapes = [(240 * 240 * 60, 240 * 240, 60, 512, 4, 256), (40 * 40 * 30, 1600, 4, 512, 4, 256), (40 * 40 * 30, 1600, 30, 134, 4, 256),
              (48 * 48 * 12, 1, 6, 512, 4, 256), (150, 1, 1, 13, 4, 256), (1, 1, 1, 1, 4, 256), (0, 1, 1, 1, 4, 256), (100_000_000, 1, 16, 512, 4, 256)]
    for s in shapes:
        assert f(*s, 3) == 0, s
    rng = np.random.default_rng(7)
    for _ in range(1500):
        grid = int(rng.choice([1, 2, 7, 8, 9, 64, 255, 512]))
        if rng.random() < 0.5:
            nij, nlev = int(rng.integers(1, 3000)), int(rng.integers(1, 130))
            npts, stride, run_len = nij * nlev, nij, int(min(nlev, rng.choice([nlev, 4, 7, 16, 128])))
        else:
            npts, stride, run_len = int(rng.integers(0, 400000)), 1, int(rng.choice([1, 2, 5, 16]))
        assert f(npts, stride, run_len, grid, 4, 256, 3) == 0, (npts, stride, run_len, grid)
