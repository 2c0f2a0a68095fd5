"""The exchange of the path for nranks > 1, on the CPU: csrc/letkf_exchange.hip rccl_allgatherv (what
letkf_obs_allgatherv_dev runs; MPI_ALLGATHERV of scale/letkf/letkf_obs.f90:1036-1046 as grouped ncclSend / ncclRecv) is
driven with a RECORDING stand-in for RCCL behind its run-time binding (bind_rccl looks ncclSend etc. up in the process
image first): every rank of a 4-rank job must post, inside ONE group, a send of its own rows to every rank (itself
included) and a receive from every rank that has rows, at the rank-major offsets of the receive buffer obsbufr -- zero-count
ranks neither send nor are received from, and byte counts beyond 2^31 stay exact.  No GPU is involved: the stand-in never
touches the pointers.  (A one-GPU box cannot run more than one rank of the real thing; the driver's 8-GPU scaling run does.)"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from __graft_entry__ import PKG_DIR, load_package

STUB = r"""
#include <stddef.h>
#include <stdio.h>
static int depth = 0, groups = 0;
static char logbuf[1 << 16];
static size_t pos = 0;
int ncclGroupStart(void) { ++depth; ++groups; pos += snprintf(logbuf + pos, sizeof logbuf - pos, "G+ "); return 0; }
int ncclGroupEnd(void) { --depth; pos += snprintf(logbuf + pos, sizeof logbuf - pos, "G- "); return 0; }
int ncclSend(const void* p, size_t n, int dt, int peer, void* comm, void* st) {
  pos += snprintf(logbuf + pos, sizeof logbuf - pos, "S:%d:%zu:%zu:%d:%d ", peer, n, (size_t)p, dt, depth); return 0; }
int ncclRecv(void* p, size_t n, int dt, int peer, void* comm, void* st) {
  pos += snprintf(logbuf + pos, sizeof logbuf - pos, "R:%d:%zu:%zu:%d:%d ", peer, n, (size_t)p, dt, depth); return 0; }
const char* ncclGetErrorString(int rc) { return "stub"; }
const char* stub_log(void) { return logbuf; }
void stub_reset(void) { pos = 0; logbuf[0] = 0; groups = 0; }
"""

CHILD = r"""
import ctypes as C, json, subprocess, sys
stub = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)            # in the process image BEFORE the library binds RCCL
lib = C.CDLL(sys.argv[2])
sym = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", sys.argv[2]]).decode().splitlines() if "rccl_allgatherv" in l]
assert len(sym) == 1, sym
f = getattr(lib, sym[0])
f.restype = C.c_int
stub.stub_log.restype = C.c_char_p
out = []
for counts, row_bytes in json.loads(sys.argv[3]):
    n = len(counts)
    per = []
    for me in range(n):
        stub.stub_reset()
        arr = (C.c_int64 * n)(*counts)
        what = C.c_char_p()
        rc = f(C.c_void_p(0xC0), C.c_int(n), C.c_int(me), arr, C.c_int64(row_bytes), C.c_void_p(0x1000), C.c_void_p(0x100000000),
               C.c_void_p(0), C.byref(what))
        per.append((rc, stub.stub_log().decode()))
    out.append(per)
print(json.dumps(out))
"""


def test_allgatherv_posts_the_reference_exchange_for_every_rank(tmp_path):
    load_package().build()
    so = os.path.join(PKG_DIR, "lib", "libletkf_amd.so")
    stub_c, stub_so = tmp_path / "stub.c", tmp_path / "libstub_rccl.so"
    stub_c.write_text(STUB)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", str(stub_c), "-o", str(stub_so)])
    cases = [([5, 0, 7, 3], 408), ([0, 0, 0, 9], 16), ([1, 1], 8), ([3000000, 2, 0, 1500000], 1000), ([4], 51 * 8)]
    r = subprocess.run([sys.executable, "-c", CHILD, str(stub_so), so, json.dumps(cases)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for (counts, rb), per in zip(cases, res):
        n = len(counts)
        offs = [sum(counts[:q]) * rb for q in range(n)]
        for me, (rc, log) in enumerate(per):
            assert rc == 0
            ev = log.split()
            assert ev[0] == "G+" and ev[-1] == "G-" and ev.count("G+") == 1 and ev.count("G-") == 1   # ONE group
            sends = [e.split(":") for e in ev if e.startswith("S:")]
            recvs = [e.split(":") for e in ev if e.startswith("R:")]
            assert all(int(e[5]) == 1 for e in sends + recvs)                  # posted inside the group
            assert all(int(e[4]) == 0 for e in sends + recvs)                  # ncclChar: counts are bytes
            # sends: my rows, whole, to every rank -- itself included -- unless I have none (MPI_ALLGATHERV's sendcount)
            if counts[me] > 0:
                assert sorted(int(e[1]) for e in sends) == list(range(n))
                assert all(int(e[2]) == counts[me] * rb and int(e[3]) == 0x1000 for e in sends)
            else:
                assert sends == []
            # receives: rank r's rows at displacement sum(counts[:r]) rows (recvcounts / displs of :1036-1046)
            want = {r_: (counts[r_] * rb, 0x100000000 + offs[r_]) for r_ in range(n) if counts[r_] > 0}
            got = {int(e[1]): (int(e[2]), int(e[3])) for e in recvs}
            assert got == want, (counts, me, got, want)
            assert len(recvs) == len(want)
