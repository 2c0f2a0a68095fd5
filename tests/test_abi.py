"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
function include/letkf_amd.h declares.  No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import ROOT, load_package


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.build()
    return p


def declared_functions():
    src = open(os.path.join(ROOT, "include", "letkf_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(letkf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    names = declared_functions()
    assert "letkf_core_c" in names and "letkf_das_points_dev" in names and len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/letkf_amd.h but not exported"
    assert set(names) == set(pkg.EXPORTS), "python binding list out of sync with the header"


def test_abi_version(pkg):
    src = open(os.path.join(ROOT, "include", "letkf_amd.h")).read()
    want = int(re.search(r"#define LETKF_AMD_ABI_VERSION (\d+)", src).group(1))
    assert pkg.lib().letkf_amd_abi_version() == want


def test_fails_loudly_without_device(pkg):
    """No CPU fallback: creating a context without a GPU is an error, not a silent slow path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.LetkfError):
        pkg.Context(0)


def test_struct_layout_matches_header(pkg):
    # sizes of the argument blocks as the C compiler lays them out (guards the ctypes mirror)
    import subprocess, tempfile
    code = '#include <stdio.h>\n#include "letkf_amd.h"\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(letkf_core_batch_args), sizeof(letkf_das_args), sizeof(letkf_search_tables), sizeof(letkf_state_consts));printf("%zu\\n", sizeof(letkf_beta_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b, c, d4, e5 = map(int, subprocess.check_output([exe]).split())
        assert d4 == C.sizeof(pkg.StateConsts)
        assert e5 == C.sizeof(pkg.BetaParams)
    assert a == C.sizeof(pkg.CoreBatchArgs)
    assert b == C.sizeof(pkg.DasArgs)
    assert c == C.sizeof(pkg.SearchTables)
