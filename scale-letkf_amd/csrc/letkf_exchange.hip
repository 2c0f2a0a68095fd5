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
typedef int (*allred_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
struct Rccl {
  send_fn send = nullptr;
  recv_fn recv = nullptr;
  grp_fn gstart = nullptr, gend = nullptr;
  err_fn errstr = nullptr;
  allred_fn allreduce = nullptr;
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
  g_rccl.allreduce = reinterpret_cast<allred_fn>(dlsym(h, "ncclAllReduce"));
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

// The pairwise exchange with true counts (MPI_ALLTOALLV): rank r hands scount[q] rows from row soff[q] of `send` to rank q and
// receives rcount[q] rows from it at row roff[q] of `recv` -- ONE group of ncclSend / ncclRecv, the own block a device copy.
// Serves (a) the halo-only exchange of the observation table (every rank gets just the rows of its extended subdomain from
// the ranks that own them, instead of the ALLGATHERV of everything; scale/letkf/letkf_obs.f90:1036-1109) and (b) the
// member <-> point transpose of scatter_grd_mpi_alltoall / gather_grd_mpi_alltoall (scale/common/common_mpi_scale.f90:1279-1396).
// On a fully connected xGMI node the pairs use different links at the same time.
int rccl_alltoallv(void* comm, int nranks, int myrank, const int64_t* scount, const int64_t* soff, const int64_t* rcount,
                   const int64_t* roff, int64_t row_bytes, const void* send, void* recv, hipStream_t st, const char** what) {
  std::call_once(g_once, bind_rccl);
  *what = "";
  if (scount[myrank] != rcount[myrank]) {
    *what = "own block: send and receive counts differ";
    return -2;
  }
  if (scount[myrank] > 0) {
    const hipError_t e = hipMemcpyAsync(static_cast<char*>(recv) + (size_t)roff[myrank] * (size_t)row_bytes,
                                        static_cast<const char*>(send) + (size_t)soff[myrank] * (size_t)row_bytes,
                                        (size_t)scount[myrank] * (size_t)row_bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      *what = hipGetErrorString(e);
      return -3;
    }
  }
  if (nranks == 1) return 0;
  if (!g_rccl.ok) return -1;
  constexpr int kNcclChar = 0;   // ncclInt8 / ncclChar, rccl.h
  int rc = g_rccl.gstart();
  if (rc) {
    *what = "ncclGroupStart";
    return rc;
  }
  for (int q = 0; q < nranks; ++q) {
    if (q == myrank) continue;
    if (scount[q] > 0 && (rc = g_rccl.send(static_cast<const char*>(send) + (size_t)soff[q] * (size_t)row_bytes,
                                           (size_t)scount[q] * (size_t)row_bytes, kNcclChar, q, comm, st))) {
      *what = "ncclSend";
      break;
    }
    if (rcount[q] > 0 && (rc = g_rccl.recv(static_cast<char*>(recv) + (size_t)roff[q] * (size_t)row_bytes,
                                           (size_t)rcount[q] * (size_t)row_bytes, kNcclChar, q, comm, st))) {
      *what = "ncclRecv";
      break;
    }
  }
  const int rc2 = g_rccl.gend();
  if (!rc && rc2) {
    *what = "ncclGroupEnd";
    rc = rc2;
  }
  if (rc && g_rccl.errstr) *what = g_rccl.errstr(rc);
  return rc;
}

// a packed block [nv3d][npl] (npl = nlev * nij1 points of this rank, level-major as gues3d) <-> member slot of the state:
// element (p, v) of the block <-> x[p * sp + mo + v * sv]
__global__ void __launch_bounds__(256) block_slot_kernel(const int dir, const long npl, const int nv3d, double* __restrict__ blk,
                                                         double* __restrict__ x, const long sp, const long mo, const long sv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npl * nv3d) return;
  const long v = i / npl, p = i - v * npl;
  if (dir == 0) x[p * sp + mo + v * sv] = blk[i];
  else blk[i] = x[p * sp + mo + v * sv];
}
hipError_t launch_block_slot(int dir, long npl, int nv3d, double* blk, double* x, long sp, long mo, long sv, hipStream_t st) {
  const long n = npl * nv3d;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(block_slot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dir, npl, nv3d, blk, x, sp, mo, sv);
  return hipGetLastError();
}

// MPI_ALLREDUCE(MPI_SUM) of int32 counters in place: the per-mesh-cell observation counts of
// scale/letkf/letkf_obs.f90:826-833 (every rank then knows every rank's row counts and the global prefix sums).
int rccl_allreduce_sum_i32(void* comm, int nranks, int64_t count, int32_t* buf, hipStream_t st, const char** what) {
  std::call_once(g_once, bind_rccl);
  *what = "";
  if (nranks == 1 || count == 0) return 0;
  if (!g_rccl.ok || !g_rccl.allreduce) return -1;
  constexpr int kNcclInt32 = 2, kNcclSum = 0;   // rccl.h
  const int rc = g_rccl.allreduce(buf, buf, (size_t)count, kNcclInt32, kNcclSum, comm, st);
  if (rc) *what = g_rccl.errstr ? g_rccl.errstr(rc) : "ncclAllReduce";
  return rc;
}

}  // namespace letkf
