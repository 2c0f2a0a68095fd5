#!/usr/bin/env python3
"""Generate tests/golden/letkf_core_golden.npz by running the REFERENCE's own letkf_core
(/root/reference/common/common_letkf.f90:52, compiled by oracle/Makefile into oracle/_ref) on the
covering case matrix in tests/_cases.py:golden_case_list().  Needs /root/reference + amdflang, i.e. it
only runs in the build container; the fixture it writes is data (inputs' SHA-256 + expected outputs).

    make -C oracle ref && python tests/golden/make_golden.py
"""
import os
import sys
import resource

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _cases import case_sha, golden_case_list, golden_inputs, probes  # noqa: E402
import _oracle  # noqa: E402


def main():
    # the reference keeps all workspace in automatic arrays: k=1000 needs a big stack
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    if _oracle.ref() is None:
        sys.exit("oracle/_ref/libletkf_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {}
    names = []
    for c in golden_case_list():
        inp = golden_inputs(c)
        r = _oracle.letkf_core("ref", c["k"], inp["nobs"], c["n"], inp["hdxb"], inp["rdiag"], inp["rloc"],
                               inp["dep"], inp["infl"], want_transm=c["transm"], want_pao=c["pao"],
                               rdiag_wloc=c["rdiag_wloc"], infl_update=c["infl_update"], depd=inp["depd"],
                               want_transmd=c["det"])
        nm = c["name"]
        names.append(nm)
        out[nm + "/sha"] = np.frombuffer(bytes.fromhex(case_sha(inp)), dtype=np.uint8)
        out[nm + "/parm_infl"] = np.array([r["parm_infl"]])
        k = c["k"]
        big = k > 100
        pr = probes(k)
        for key in ("trans", "pao"):
            if r[key] is None:
                continue
            if big:
                out[f"{nm}/{key}_probe"] = r[key] @ pr
                out[f"{nm}/{key}_diag"] = np.diag(r[key]).copy()
                out[f"{nm}/{key}_absmax"] = np.array([np.abs(r[key]).max()])
            else:
                out[f"{nm}/{key}"] = r[key]
        for key in ("transm", "transmd"):
            if r[key] is not None:
                out[f"{nm}/{key}"] = r[key]
        print(f"{nm}: ok", flush=True)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "letkf_core_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
