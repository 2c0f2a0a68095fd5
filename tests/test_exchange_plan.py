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


STUB2 = STUB + r"""
int ncclAllReduce(const void* s, void* r, size_t n, int dt, int op, void* comm, void* st) {
  pos += snprintf(logbuf + pos, sizeof logbuf - pos, "A:%zu:%zu:%zu:%d:%d ", (size_t)s, (size_t)r, n, dt, op); return 0; }
"""

CHILD2 = r"""
import ctypes as C, json, subprocess, sys
stub = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
lib = C.CDLL(sys.argv[2])
names = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", sys.argv[2]]).decode().splitlines()]
a2a = [s for s in names if "rccl_alltoallv" in s]
ar = [s for s in names if "rccl_allreduce_sum_i32" in s]
assert len(a2a) == 1 and len(ar) == 1, (a2a, ar)
f = getattr(lib, a2a[0]); f.restype = C.c_int
g = getattr(lib, ar[0]); g.restype = C.c_int
stub.stub_log.restype = C.c_char_p
out = []
for M, row_bytes in json.loads(sys.argv[3]):           # M[s][d] = rows rank s sends to rank d
    n = len(M)
    per = []
    for me in range(n):
        stub.stub_reset()
        sc = [M[me][d] for d in range(n)]
        rc_ = [M[s][me] for s in range(n)]
        so = [sum(sc[:d]) for d in range(n)]
        ro = [sum(rc_[:s]) for s in range(n)]
        arr = lambda v: (C.c_int64 * n)(*v)
        what = C.c_char_p()
        rc = f(C.c_void_p(0xC0), C.c_int(n), C.c_int(me), arr(sc), arr(so), arr(rc_), arr(ro), C.c_int64(row_bytes),
               C.c_void_p(0x1000), C.c_void_p(0x100000000), C.c_void_p(0), C.byref(what))
        per.append((rc, stub.stub_log().decode()))
    out.append(per)
stub.stub_reset()
what = C.c_char_p()
rc = g(C.c_void_p(0xC0), C.c_int(4), C.c_int64(12345), C.c_void_p(0x2000), C.c_void_p(0), C.byref(what))
ar_log = (rc, stub.stub_log().decode())
stub.stub_reset()
rc1 = g(C.c_void_p(0), C.c_int(1), C.c_int64(77), C.c_void_p(0x2000), C.c_void_p(0), C.byref(what))
print(json.dumps([out, ar_log, (rc1, stub.stub_log().decode())]))
"""


def test_alltoallv_and_allreduce_post_the_reference_exchanges(tmp_path):
    """csrc/letkf_exchange.hip rccl_alltoallv (letkf_alltoallv_dev: the halo-only exchange of the observation table and the
    transport of scatter / gather_grd_mpi_alltoall, scale/common/common_mpi_scale.f90:1311-1316, 1372-1377 = MPI_ALLTOALL(V))
    and rccl_allreduce_sum_i32 (the mesh-cell counts, scale/letkf/letkf_obs.f90:826-833) behind the recording stand-in for RCCL:
    every rank of a job posts, in ONE group, exactly the sends M[me][d] -> d and the receives M[s][me] <- s at the offsets it was
    given, nothing for empty blocks and nothing to itself; the all-reduce is one in-place ncclAllReduce(int32, sum) and a
    single-rank job issues nothing.  (Own blocks are device copies and stay zero here: no device in this test.)"""
    load_package().build()
    so = os.path.join(PKG_DIR, "lib", "libletkf_amd.so")
    stub_c, stub_so = tmp_path / "stub2.c", tmp_path / "libstub_rccl2.so"
    stub_c.write_text(STUB2)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", str(stub_c), "-o", str(stub_so)])
    cases = [([[0, 5, 0, 2], [7, 0, 1, 0], [0, 0, 0, 0], [3, 4, 9, 0]], 408),        # a halo exchange: rank 2 owns no row
             ([[0, 3000000000], [1, 0]], 8),                                        # byte counts beyond 2^31
             ([[0, 11, 11], [11, 0, 11], [11, 11, 0]], 8 * 30 * 100)]                 # the transpose: equal blocks
    r = subprocess.run([sys.executable, "-c", CHILD2, str(stub_so), so, json.dumps(cases)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    res, ar_log, ar1 = json.loads(r.stdout.strip().splitlines()[-1])
    for (M, rb), per in zip(cases, res):
        n = len(M)
        for me, (rc, log) in enumerate(per):
            assert rc == 0
            ev = log.split()
            assert ev[0] == "G+" and ev[-1] == "G-" and ev.count("G+") == 1 and ev.count("G-") == 1
            sends = {int(e.split(":")[1]): e.split(":") for e in ev if e.startswith("S:")}
            recvs = {int(e.split(":")[1]): e.split(":") for e in ev if e.startswith("R:")}
            sc = [M[me][d] for d in range(n)]
            rc_ = [M[s][me] for s in range(n)]
            assert sorted(sends) == [d for d in range(n) if d != me and sc[d] > 0]
            assert sorted(recvs) == [s for s in range(n) if s != me and rc_[s] > 0]
            for d, e in sends.items():
                assert int(e[2]) == sc[d] * rb and int(e[3]) == 0x1000 + sum(sc[:d]) * rb and int(e[4]) == 0 and int(e[5]) == 1
            for s_, e in recvs.items():
                assert int(e[2]) == rc_[s_] * rb and int(e[3]) == 0x100000000 + sum(rc_[:s_]) * rb and int(e[5]) == 1
    rc, log = ar_log
    assert rc == 0 and log.split() == [f"A:{0x2000}:{0x2000}:12345:2:0"]          # in place, ncclInt32, ncclSum
    assert ar1 == [0, ""]                                                          # one rank: nothing to reduce
