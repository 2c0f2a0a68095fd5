// letkf_exchange.hip -- the path's ONE exchange, inside the library (C ABI section 8): the ALLGATHERV of the sorted
// observation buffers over the subdomain ranks, scale/letkf/letkf_obs.f90:1036-1046 (MPI_COMM_d), as grouped
// ncclSend / ncclRecv with the true row counts on an RCCL communicator that the HOST owns (Fortran / C / Python:
// whoever called ncclCommInitRank).  RCCL over xGMI is point-to-point; grouped send/recv lets every pair of GPUs use
// its own link concurrently and moves exactly the rows that exist (no padding to the largest rank).
//
// RCCL is bound at run time (dlsym on the process image first -- a host that already carries RCCL, e.g. through
// torch, keeps ONE copy -- else dlopen of librccl.so.1), so the library has no link-time dependency on it and still
// loads on a box without RCCL; the entry then fails with LETKF_E_INVALID.

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "letkf_device.h"

namespace letkf {

namespace {
typedef int (*send_fn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*recv_fn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*grp_fn)(void);
typedef const char* (*err_fn)(int);
struct Rccl {
  send_fn send = nullptr;
  recv_fn recv = nullptr;
  grp_fn gstart = nullptr, gend = nullptr;
  err_fn errstr = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void bind_rccl() {
  void* h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclSend")) {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
  }
  g_rccl.send = reinterpret_cast<send_fn>(dlsym(h, "ncclSend"));
  g_rccl.recv = reinterpret_cast<recv_fn>(dlsym(h, "ncclRecv"));
  g_rccl.gstart = reinterpret_cast<grp_fn>(dlsym(h, "ncclGroupStart"));
  g_rccl.gend = reinterpret_cast<grp_fn>(dlsym(h, "ncclGroupEnd"));
  g_rccl.errstr = reinterpret_cast<err_fn>(dlsym(h, "ncclGetErrorString"));
  g_rccl.ok = g_rccl.send && g_rccl.recv && g_rccl.gstart && g_rccl.gend;
}
}  // namespace

// returns 0, -1 (RCCL not available) or the positive ncclResult_t of the failing call (*what names it)
int rccl_allgatherv(void* comm, int nranks, int myrank, const int64_t* counts, int64_t row_bytes, const void* send,
                    void* recv, hipStream_t st, const char** what) {
  std::call_once(g_once, bind_rccl);
  *what = "";
  if (!g_rccl.ok) return -1;
  constexpr int kNcclChar = 0;   // ncclInt8 / ncclChar, rccl.h
  int rc = g_rccl.gstart();
  if (rc) {
    *what = "ncclGroupStart";
    return rc;
  }
  size_t off = 0;
  for (int r = 0; r < nranks; ++r) {
    const size_t nbytes = (size_t)counts[r] * (size_t)row_bytes;
    if (counts[myrank] > 0 && (rc = g_rccl.send(send, (size_t)counts[myrank] * (size_t)row_bytes, kNcclChar, r, comm, st))) {
      *what = "ncclSend";
      break;
    }
    if (nbytes > 0 && (rc = g_rccl.recv(static_cast<char*>(recv) + off, nbytes, kNcclChar, r, comm, st))) {
      *what = "ncclRecv";
      break;
    }
    off += nbytes;
  }
  const int rc2 = g_rccl.gend();
  if (!rc && rc2) {
    *what = "ncclGroupEnd";
    rc = rc2;
  }
  if (rc && g_rccl.errstr) *what = g_rccl.errstr(rc);
  return rc;
}

}  // namespace letkf
