// letkf_api.hip -- host side of the C ABI declared in include/letkf_amd.h.
//
// Thin by design: argument checking, launch planning (LDS carve, Jacobi variant, grid),
// the large-k workspace, optional HIP-event timing for bench.py, and the host-pointer
// compatibility entry letkf_core_c that the Fortran shim (scale-letkf_amd/fortran) calls.
// There is no CPU fallback anywhere in this file: without a device every compute entry
// returns LETKF_E_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/letkf_amd.h"
#include "letkf_device.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(LETKF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)

}  // namespace

struct letkf_ctx {
  int device = -1;
  int num_cu = 256;
  size_t lds_max = 160 * 1024;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  double* ws = nullptr;       // large-k workspace
  size_t ws_bytes = 0;
  char* warm_ws = nullptr;    // wave kernel: eigenvectors handed from point to point inside a run
  unsigned* sched = nullptr;  // wave kernel: the 8 run counters of the dynamic scheduling (512 bytes)
  size_t warm_ws_bytes = 0;
  char* scratch = nullptr;    // staging for the host-pointer entry
  size_t scratch_bytes = 0;
  char* list_ws = nullptr;    // letkf_das_columns_dev: the local-observation lists of one slab of levels / the survivors of a batch of columns
  size_t list_ws_bytes = 0;
  char* slot_ws = nullptr;    // ... its list-free route: one local list per resident wave
  size_t slot_ws_bytes = 0;
  char* ring_ws = nullptr;    // limited column search on dense observations: ring-ordered survivors of a batch of columns
  size_t ring_ws_bytes = 0;
  char* ring_aux = nullptr;   // ... their counts / offsets / ring starts
  size_t ring_aux_bytes = 0;
  // (the last "not dense" verdict, by the identity of the tables and columns it was given for: the weighing costs a survivor count
  // and two read-backs -- 17 ms on C2's grid.  Pointer identity says nothing about the CONTENT -- a host that frees and reallocates
  // its tables every analysis gets the same addresses with other observations -- so the verdict only serves (a) the fill call that
  // directly follows the count call it was made in and (b) the calls of one letkf_das_columns_dev; it is dropped after that use, at
  // the end of that entry and by letkf_ctx_set_option.)
  const void* ring_no[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t ring_no_n = -1;
  int ring_no_crit = 0;
  bool ring_keep = false;     // inside letkf_das_columns_dev: the survivors of the first search call serve the later ones
  bool ring_ready = false;
  std::vector<int64_t> ring_hoff;
  int ring_batch_mb = 8192;   // LETKF_OPT_RING_BATCH_MB
  bool ring_release = false;  // LETKF_OPT_RING_RELEASE
  int limited_rings = 2;      // LETKF_OPT_LIMITED_RINGS: 0 never, 1 wherever eligible, 2 where a group's survivors overflow the column kernel's buffer
  char* staged_ws = nullptr;  // staged path: per-point slabs of a batch + meta / info words
  size_t staged_ws_bytes = 0;
  std::string last_path;      // kernels the last loop-body / letkf_core launch went through (bench.py reports it)
  bool timing = false;
  bool staged_poly = true;    // LETKF_OPT_STAGED_POLY
  bool trio = true;           // LETKF_OPT_SMALL_K_TRIO
  int col_survivors = 2;      // LETKF_OPT_COLUMN_SURVIVORS: 0 never, 1 wherever the one-wave kernel serves the call, 2 where the lists would not fit
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

namespace {

int ensure_bytes(letkf_ctx* c, char** buf, size_t* have, size_t need) {
  if (need <= *have) return LETKF_OK;
  if (*buf) HIP_TRY(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  size_t cap = need + need / 4 + 4096;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(buf), cap));
  *have = cap;
  (void)c;
  return LETKF_OK;
}

struct Plan {
  letkf::LaunchPlan lp;
  int ldg, ldy, tn;
  long ws_per_block;
};

// LDS carve in doubles; must mirror letkf_point_kernel's carve.
size_t lds_doubles(bool big, int k, int nv, int ldg, int ldy, int tn) {
  const int nb = nv + 2;
  size_t fixed = 7 * (size_t)k + 6 * (size_t)nv + 16 + 3 * (size_t)tn + 1;
  if (big) return fixed + (size_t)tn * ldy;
  size_t g = (size_t)k * ldg;
  g += g & 1;
  size_t tile = (size_t)tn * ldy;
  size_t ux = (size_t)k * nb + (size_t)nv * k;
  return fixed + g + (tile > ux ? tile : ux);
}

int make_plan(const letkf_ctx* c, int k, int nv, long npts, Plan* p) {
  if (k < 2) return fail(LETKF_E_INVALID, "ensemble size must be >= 2");
  if (nv < 0 || npts < 0) return fail(LETKF_E_INVALID, "negative size");
  p->ldg = k | 1;
  p->ldy = (k + 3) & ~3;
  bool big = k > 128;
  if (!big) {
    p->tn = 32;
    size_t bytes = 8 * lds_doubles(false, k, nv, p->ldg, p->ldy, p->tn);
    if (bytes > c->lds_max) big = true;
    else {
      p->lp.big = false;
      p->lp.rmax = (k <= 32) ? 4 : (k <= 56) ? 7 : (k <= 64) ? 8 : (k <= 104) ? 13 : 16;
      p->lp.block = 256;
      p->lp.lds_bytes = bytes;
      long g = (long)c->num_cu * 32;
      p->lp.grid = (int)(npts < g ? (npts > 0 ? npts : 1) : g);
      p->ws_per_block = 0;
      return LETKF_OK;
    }
  }
  // large-k spill path: G, U, X in a per-workgroup HBM workspace, obs tile sized to what LDS is left
  p->lp.big = true;
  p->lp.rmax = 0;
  // one wave per block pair of the block Jacobi (k/32 pairs per round), 4..12 waves: 12 waves = 3 per SIMD keeps
  // 170 VGPRs per lane for the in-register 32 x 32 eigensolver; one workgroup per CU
  const int nblk = (k + 15) / 16, nbe = nblk + (nblk & 1);
  int waves = nbe / 2;
  if (waves < 4) waves = 4;
  if (waves > 12) waves = 12;
  p->lp.block = 64 * waves;
  const size_t budget = 128 * 1024 / 8;  // doubles of LDS we allow ourselves
  size_t fixed = lds_doubles(true, k, nv, p->ldg, p->ldy, 0);
  long tn = fixed < budget ? (long)((budget - fixed) / (p->ldy + 3)) : 0;
  if (tn > 32) tn = 32;
  if (tn < 4) tn = 4;
  // the obs tile region doubles as the per-wave scratch of the block Jacobi
  while ((size_t)tn * p->ldy < (size_t)waves * letkf::kBlockJacobiScratch) ++tn;
  p->tn = (int)tn;
  p->lp.lds_bytes = 8 * lds_doubles(true, k, nv, p->ldg, p->ldy, p->tn);
  if (p->lp.lds_bytes > c->lds_max) return fail(LETKF_E_INVALID, "ensemble size too large for the LDS vectors");
  long g = (long)c->num_cu;
  p->lp.grid = (int)(npts < g ? (npts > 0 ? npts : 1) : g);
  const int nb = nv + 2;
  long w = (long)k * p->ldg + (long)k * nb + (long)nv * k;
  p->ws_per_block = (w + 1) & ~1L;
  return LETKF_OK;
}

int ensure_ws(letkf_ctx* c, const Plan& p) {
  if (!p.lp.big) return LETKF_OK;
  size_t need = (size_t)p.lp.grid * (size_t)p.ws_per_block * sizeof(double);
  if (need <= c->ws_bytes) return LETKF_OK;
  if (c->ws) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
  }
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws), need));
  c->ws_bytes = need;
  return LETKF_OK;
}

// Measurement-only knobs exist in the PROF twin of the library (make PROF=1) and nowhere else: the production build
// reads no environment variable that could change a result.
#ifdef LETKF_WAVE_PROF
#define LETKF_KNOB(name) std::getenv(name)
#else
#define LETKF_KNOB(name) static_cast<const char*>(nullptr)
#endif

struct EventPair {   // timing events that do not outlive a failed launch
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

int launch_staged(letkf_ctx* c, letkf::PointArgs& a);

int launch(letkf_ctx* c, letkf::PointArgs& a, int warm_run = 0, long warm_stride = 1) {
  if (a.k < 2) return fail(LETKF_E_INVALID, "ensemble size must be >= 2");
  if (a.nv < 0 || a.npts < 0) return fail(LETKF_E_INVALID, "negative size");
  a.max_sweep = 60;
  if (const char* e = LETKF_KNOB("LETKF_AMD_MAX_SWEEP")) {   // PROF knob: time the non-eigensolve phases
    int v = std::atoi(e);
    if (v >= 0 && v < 60) a.max_sweep = v;   // 0: skip the eigensolve entirely (timing only, results invalid)
  }
  // k <= 64: one wavefront per grid point, matrix in registers (letkf_wave.hip); otherwise one workgroup per
  // point with the matrix in LDS, or in the HBM workspace for large k (letkf_kernels.hip)
  const bool force_block = LETKF_KNOB("LETKF_AMD_FORCE_BLOCK") != nullptr;
  if (a.mode == 2 && force_block) return fail(LETKF_E_INVALID, "LETKF_AMD_FORCE_BLOCK: the workgroup kernel has no fused search");
  bool wave = !force_block && letkf::wave_kernel_supports(a.k, a.nv, a.mode);
  if (const char* e = LETKF_KNOB("LETKF_AMD_STAGED_MIN_K"))   // PROF knob: A/B the staged path against the two-wave kernel
    if (a.mode != 2 && a.k >= std::atoi(e)) wave = false;
  // 63 <= k <= 100, loop body without k x k outputs: the staged path analyses such a point without an eigen-decomposition
  // (letkf_staged.hip poly_apply; MEMBER = 100: 1.07 M solves/s against 0.64 M on the two-wave Jacobi kernel, same result to
  // rounding).  The two-wave kernel keeps the calls that return T / Pa, and everything when LETKF_OPT_STAGED_POLY is 0.
  if (wave && a.k >= 63 && a.mode == 0 && !a.trans_out && !a.pa_out && c->staged_poly && a.nv + 2 <= 16) wave = false;
#ifdef LETKF_STAGED_MIN_K   // A/B twins (make VARIANT=...): the same switch at compile time
  if (a.mode != 2 && a.k >= LETKF_STAGED_MIN_K) wave = false;
#endif
  // beyond the register kernels: the staged three-kernel path (the monolithic workgroup kernel below stays reachable
  // through the PROF twin's LETKF_AMD_FORCE_BLOCK / LETKF_AMD_MONOLITHIC knobs for A/B measurements)
  if (!wave && !force_block && a.mode != 2 && a.nv + 2 <= 16 && !LETKF_KNOB("LETKF_AMD_MONOLITHIC")) return launch_staged(c, a);
  // the route is decided: only now the plan (and, for large k, the workspace) of the monolithic workgroup kernel -- the
  // staged path above has its own slabs and serves ensemble sizes whose vectors this kernel's LDS carve would refuse
  Plan p;
  if (int rc = make_plan(c, a.k, a.nv, a.npts, &p)) return rc;
  if (!wave)
    if (int rc = ensure_ws(c, p)) return rc;
  a.ldg = p.ldg;
  a.ldy = p.ldy;
  a.tn = p.tn;
  a.ws = c->ws;
  a.ws_per_block = p.ws_per_block;
  a.big_block = (p.lp.big && !LETKF_KNOB("LETKF_AMD_BIG_STREAM")) ? 1 : 0;   // PROF knob: the older streaming Jacobi
  if (wave) {
    int run_req = warm_run;
    if (const char* e = LETKF_KNOB("LETKF_AMD_RUN_LEN")) run_req = std::atoi(e);   // PROF knob: 1 = all cold
    size_t wbytes = 0;
    a.warm_stride = warm_stride > 1 ? warm_stride : 1;
    if (a.npts % a.warm_stride != 0) return fail(LETKF_E_INVALID, "warm_stride does not divide npts");
    letkf::wave_launch_shape(a.k, a.mode, a.npts, c->num_cu, run_req, a.warm_stride, &a.run_len, &a.wave_grid, &wbytes);
    if (wbytes > c->warm_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));   // old buffer may still be in use
    if (int rc = ensure_bytes(c, &c->warm_ws, &c->warm_ws_bytes, wbytes)) return rc;
    a.warm_ws = reinterpret_cast<double*>(c->warm_ws);
    if (a.mode == 3) {   // one local-list slot per wave of the grid (4 waves per workgroup): idx | rdiag | rloc
      const size_t nslot = (size_t)a.wave_grid * 4, cap = 2 * (size_t)(a.sl_cap > 0 ? a.sl_cap : 4);   // (two lists per wave: this level's and the next one's)
      const size_t o_rd = (nslot * cap * 4 + 255) & ~(size_t)255, o_rl = o_rd + nslot * cap * 8;
      const size_t need = o_rl + nslot * cap * 8 + 256;
      if (need > c->slot_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
      if (int rc = ensure_bytes(c, &c->slot_ws, &c->slot_ws_bytes, need)) return rc;
      a.sl_idx = reinterpret_cast<int*>(c->slot_ws);
      a.sl_rd = reinterpret_cast<double*>(c->slot_ws + o_rd);
      a.sl_rl = reinterpret_cast<double*>(c->slot_ws + o_rl);
      a.obs_idx = a.sl_idx;
      a.rdiag_l = a.sl_rd;
      a.rloc_l = a.sl_rl;
    }
    if (!c->sched) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->sched), 512));
      HIP_TRY(hipMemsetAsync(c->sched, 0, 512, c->stream));   // (later launches reset it themselves when they draw)
    }
    a.sched = LETKF_KNOB("LETKF_AMD_STATIC_SCHED") ? nullptr : c->sched;   // PROF knob: the static dealing, for A/B runs
    a.warm_dbg = 0;
    if (const char* e = LETKF_KNOB("LETKF_AMD_WARM_DBG")) a.warm_dbg = std::atoi(e);
    a.prof = nullptr;
#ifdef LETKF_CHECKED
    {   // the violation record of the checked build (letkf_wave.hip LETKF_CHECK): code | workgroup | value | bound
      static unsigned long long* rec = nullptr;
      if (!rec) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rec), 4 * sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(rec, 0, 4 * sizeof(unsigned long long), c->stream));
      a.prof = rec;
    }
#endif
  }
  // every argument check is behind us: only now create the timing events (destroyed again if the launch fails)
  EventPair ev;
  if (c->timing) {
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    HIP_TRY(hipEventRecord(ev.e0, c->stream));
  }
  if (wave) {
#ifdef LETKF_WAVE_PROF
    static unsigned long long* prof_dev = nullptr;
    if (!prof_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&prof_dev), 26 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(prof_dev, 0, 26 * sizeof(unsigned long long), c->stream));
    a.prof = prof_dev;
#endif
    // points without observations / with beta = 0: one streaming pass, thread per point (letkf_trivial.hip)
    if (letkf::trivial_pass_supports(a) && !LETKF_KNOB("LETKF_AMD_NO_TRIVIAL_PASS")) {
      HIP_TRY(letkf::launch_trivial_points(a, c->stream));
      a.skip_trivial = 1;
    }
    if (c->trio && letkf::trio_kernel_supports(a)) {
      HIP_TRY(letkf::launch_trio_kernel(a, c->num_cu, c->stream));
      c->last_path = std::string("letkf_trio_kernel<KR=") + (a.k <= 16 ? "16" : "20") + ",P=" + std::to_string(letkf::trio_points_per_wave(a.k)) + ">";
    } else {
    HIP_TRY(letkf::launch_wave_kernel(a, c->num_cu, c->stream));
    c->last_path = "letkf_wave_kernel<KR=" + std::to_string(letkf::wave_kernel_kr(a.k)) + ",NV=" + std::to_string(a.nv) +
                   ",NW=" + (a.k <= 62 ? "1" : "2") + (a.mode == 2 ? ",FUSED" : a.mode == 3 ? ",FUSED: column survivors" : "") + ">";
    }
#ifdef LETKF_WAVE_PROF
    {
      unsigned long long h[26];
      HIP_TRY(hipMemcpyAsync(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      unsigned long long tot = 0;
      for (int i = 0; i < 10; ++i) tot += h[i];
      std::fprintf(stderr, "[letkf prof] wave-time share by phase (s_memtime ticks, all waves):");
      for (int i = 0; i < 10; ++i) std::fprintf(stderr, " p%d=%.1f%%", i, tot ? 100.0 * (double)h[i] / (double)tot : 0.0);
      std::fprintf(stderr, " total=%llu\n", tot);
      std::fprintf(stderr, "[letkf prof] first wave start .. last wave end: %llu ticks; waves by units done (0..11+):", h[11] - ~h[10]);
      for (int i = 12; i < 24; ++i) std::fprintf(stderr, " %llu", h[i]);
      std::fprintf(stderr, "; units done in all: %llu\n", h[24]);
    }
#endif
  } else {
    HIP_TRY(letkf::launch_point_kernel(a, p.lp, c->stream));
    c->last_path = p.lp.big ? "letkf_point_kernel<BIG>" : "letkf_point_kernel<LDS>";
  }
  if (c->timing) {
    HIP_TRY(hipEventRecord(ev.e1, c->stream));
    c->events.emplace_back(ev.e0, ev.e1);
    ev.e0 = ev.e1 = nullptr;   // owned by the context from here
  }
#ifdef LETKF_CHECKED
  if (wave && a.prof) {
    unsigned long long h[4];
    HIP_TRY(hipMemcpyAsync(h, a.prof, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h[0])
      return fail(LETKF_E_HIP, "checked build: bound " + std::to_string(h[0]) + " violated in workgroup " + std::to_string(h[1]) + ": value " +
                                   std::to_string((long long)h[2]) + " against " + std::to_string((long long)h[3]));
  }
#endif
  return LETKF_OK;
}

// Staged path (letkf_staged.hip): k beyond the register kernels.  The points are processed in batches whose slabs
// fit a fixed workspace budget; per batch: Gram stage, eigen stage (workgroup Jacobi for orders <= 208, block Jacobi
// above), apply stage -- all on the context's stream, no host synchronisation in between.
int launch_staged(letkf_ctx* c, letkf::PointArgs& a) {
  const int kkout = (a.trans_out || a.pa_out) ? 1 : 0;
  // eigen-free stage (letkf_krylov.hip): the loop body without k x k outputs, any k whose point matrices (order min(n, k))
  // fit the stage (<= 512 rows; larger orders keep the eigen stage)
  const bool krylov = a.mode == 0 && !kkout && c->staged_poly;
  const long hist = krylov ? letkf::stage_krylov_hist_doubles(a.k) : 0;
  const long wpp = letkf::staged_ws_per_point(a.k, a.nv, kkout, hist);
  // slabs of one batch: at most 6 GiB (+ as much again per 2 MB of residual history per point, up to 24 GiB)
  size_t budget = (size_t)6 << 30;
  if (hist) budget += std::min<size_t>((size_t)18 << 30, (size_t)hist * sizeof(double) * 3072);
  long nb = (long)(budget / ((size_t)wpp * sizeof(double)));
  const long want = (long)c->num_cu * 16;
  if (nb > want) nb = want;
  if (nb < 1) nb = 1;
  if (nb > a.npts) nb = a.npts;
  {   // equal batches (a last batch of a few points would leave the chip idle for a whole eigen-solve) ...
    const long nbat = (a.npts + nb - 1) / nb;
    const long cap = nb;
    nb = (a.npts + nbat - 1) / nbat;
#ifndef STAGED_BATCH_ROUND
#define STAGED_BATCH_ROUND 1
#endif
    // ... of whole rounds of workgroups: the stages run 2 (Gram, eigen-free stage at small orders) to 4 (apply) workgroups per CU,
    // and a batch that is no multiple of 2 x #CU ends every one of its kernels on a partly filled round (27648 points in 7
    // batches of 3950 = 7.7 rounds of 512: r4, MEMBER = 100)
    if (STAGED_BATCH_ROUND) {
      const long q = 2L * c->num_cu;
      const long up = (nb + q - 1) / q * q;
      if (up <= cap) nb = up;
    }
  }
  const size_t slab_bytes = (size_t)nb * (size_t)wpp * sizeof(double);
  const size_t need = slab_bytes + (size_t)nb * 4 * sizeof(int) + 256;
  if (need > c->staged_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->staged_ws, &c->staged_ws_bytes, need)) return rc;
  letkf::StagedArgs s;
  s.A = a;
  s.A.ws = reinterpret_cast<double*>(c->staged_ws);
  s.A.ws_per_block = wpp;
  s.meta = reinterpret_cast<int*>(c->staged_ws + slab_bytes);
  s.info = s.meta + 2 * nb;
  s.kkout = kkout;
  s.wg_max_order = letkf::eig_wg_max_order();
  s.poly_max_n = krylov ? letkf::stage_krylov_max_n(a.k) : 0;
  s.gram_mfma = a.mode == 0 ? 1 : 0;   // stage 1 on the matrix cores (letkf_gram.hip); the older kernel takes what that one leaves
  s.A.max_sweep = 60;
  EventPair ev;
  if (c->timing) {
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    HIP_TRY(hipEventRecord(ev.e0, c->stream));
  }
  if (letkf::trivial_pass_supports(a) && !LETKF_KNOB("LETKF_AMD_NO_TRIVIAL_PASS")) {   // as in launch()
    HIP_TRY(letkf::launch_trivial_points(a, c->stream));
    s.A.skip_trivial = 1;
  }
  for (long p0 = 0; p0 < a.npts; p0 += nb) {
    s.pt0 = p0;
    s.nbatch = (a.npts - p0 < nb) ? a.npts - p0 : nb;
    if (s.gram_mfma) HIP_TRY(letkf::launch_stage_gram_mfma(s, c->stream));
    if (!s.gram_mfma || a.k > 512) HIP_TRY(letkf::launch_stage_gram(s, c->lds_max, c->stream));   // (k <= 512: every point is the first kernel's)
    if (s.poly_max_n > 0) HIP_TRY(letkf::launch_stage_krylov(s, c->lds_max, c->stream));   // (points it gives up: eigen stage, next)
    letkf::EigArgs e;
    e.ws = s.A.ws;
    e.ws_per_point = wpp;
    e.npts = s.nbatch;
    e.pt0 = p0;
    e.meta = s.meta;
    e.info = s.info;
    e.max_sweep = 60;
    HIP_TRY(letkf::launch_eig_wg(e, a.k < s.wg_max_order ? a.k : s.wg_max_order, c->num_cu, c->stream));
    if (a.k > s.wg_max_order) HIP_TRY(letkf::launch_eig_block(e, a.k, c->num_cu, c->stream));
    HIP_TRY(letkf::launch_stage_apply(s, c->stream));
  }
  if (c->timing) {
    HIP_TRY(hipEventRecord(ev.e1, c->stream));
    c->events.emplace_back(ev.e0, ev.e1);
    ev.e0 = ev.e1 = nullptr;
  }
  c->last_path = std::string(s.gram_mfma ? "staged: letkf_stage_gram_mfma_kernel + " : "staged: letkf_stage_gram_kernel + ") + (s.poly_max_n > 0 ? "letkf_stage_krylov_kernel (CG + Lanczos; points it gives up: " : "") +
                 (a.k <= 128 ? "letkf_eig_wg_kernel<4,32,32,1>" : "letkf_eig_wg_kernel<4,52,16,2>") +
                 (a.k > s.wg_max_order ? " / letkf_eig_block_kernel" : "") + (s.poly_max_n > 0 ? ")" : "") + " + letkf_stage_apply_kernel";
  return LETKF_OK;
}

int check_ctx(letkf_ctx* c) {
  if (!c || c->device < 0) return fail(LETKF_E_NO_DEVICE, "context is not bound to a device");
  HIP_TRY(hipSetDevice(c->device));
  return LETKF_OK;
}

}  // namespace

extern "C" {

int letkf_amd_abi_version(void) { return LETKF_AMD_ABI_VERSION; }

const char* letkf_amd_last_error(void) { return g_last_error.c_str(); }

int letkf_ctx_create(int device_id, letkf_ctx** out) {
  if (!out) return fail(LETKF_E_INVALID, "ctx out pointer is NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(LETKF_E_NO_DEVICE, "no HIP device visible: this library has no CPU path");
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(LETKF_E_INVALID, "device id out of range");
  HIP_TRY(hipSetDevice(dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(LETKF_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
  letkf_ctx* c = new letkf_ctx();
  c->device = dev;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->lds_max = prop.sharedMemPerBlock >= 64 * 1024 ? (size_t)prop.sharedMemPerBlock : 64 * 1024;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
      c->lds_max = (size_t)v;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(LETKF_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  *out = c;
  return LETKF_OK;
}

int letkf_ctx_destroy(letkf_ctx* c) {
  if (!c) return LETKF_OK;
  if (c->device >= 0) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& ev : c->events) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    if (c->ws) (void)hipFree(c->ws);
    if (c->warm_ws) (void)hipFree(c->warm_ws);
    if (c->sched) (void)hipFree(c->sched);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->staged_ws) (void)hipFree(c->staged_ws);
    if (c->list_ws) (void)hipFree(c->list_ws);
    if (c->slot_ws) (void)hipFree(c->slot_ws);
    if (c->ring_ws) (void)hipFree(c->ring_ws);
    if (c->ring_aux) (void)hipFree(c->ring_aux);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  }
  delete c;
  return LETKF_OK;
}

int letkf_ctx_set_option(letkf_ctx* c, int option, int value) {
  if (int rc = check_ctx(c)) return rc;
  c->ring_no_n = -1;   // (no verdict of the limited column search outlives a change of options)
  switch (option) {
    case LETKF_OPT_STAGED_POLY: c->staged_poly = value != 0; return LETKF_OK;
    case LETKF_OPT_RING_BATCH_MB:
      if (value < 1) return fail(LETKF_E_INVALID, "LETKF_OPT_RING_BATCH_MB: >= 1");
      c->ring_batch_mb = value;
      return LETKF_OK;
    case LETKF_OPT_RING_RELEASE: c->ring_release = value != 0; return LETKF_OK;
    case LETKF_OPT_SMALL_K_TRIO: c->trio = value != 0; return LETKF_OK;
    case LETKF_OPT_LIMITED_RINGS:
      if (value < 0 || value > 2) return fail(LETKF_E_INVALID, "LETKF_OPT_LIMITED_RINGS: 0, 1 or 2");
      c->limited_rings = value;
      return LETKF_OK;
    case LETKF_OPT_COLUMN_SURVIVORS:
      if (value < 0 || value > 2) return fail(LETKF_E_INVALID, "LETKF_OPT_COLUMN_SURVIVORS: 0, 1 or 2");
      c->col_survivors = value;
      return LETKF_OK;
    default: return fail(LETKF_E_INVALID, "unknown option");
  }
}

int letkf_ctx_set_stream(letkf_ctx* c, void* hip_stream) {
  if (int rc = check_ctx(c)) return rc;
  // the handle is used as is: NULL is HIP's default (null) stream, which is also what torch.cuda.current_stream()
  // hands out unless the caller switched streams -- work then orders with the caller's other work on that stream
  c->stream = static_cast<hipStream_t>(hip_stream);
  return LETKF_OK;
}

int letkf_ctx_synchronize(letkf_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LETKF_OK;
}

int letkf_sched_plan_check(int64_t npts, int64_t stride, int32_t run_len, int32_t grid, int32_t ppw, int32_t resident_per_xcd) {
  return letkf::sched_plan_check((long)npts, (long)stride, run_len, grid, ppw, resident_per_xcd, 1);
}
int letkf_sched_plan_check_units(int64_t npts, int64_t stride, int32_t run_len, int32_t grid, int32_t ppw, int32_t resident_per_xcd, int32_t ub_of) {
  return letkf::sched_plan_check((long)npts, (long)stride, run_len, grid, ppw, resident_per_xcd, ub_of);
}

int letkf_ctx_last_path(letkf_ctx* c, char* buf, int32_t len) {
  if (!c || !buf || len < 1) return fail(LETKF_E_INVALID, "bad argument");
  std::snprintf(buf, (size_t)len, "%s", c->last_path.c_str());
  return LETKF_OK;
}

int letkf_ctx_timing_enable(letkf_ctx* c, int enable) {
  if (int rc = check_ctx(c)) return rc;
  c->timing = enable != 0;
  return LETKF_OK;
}

int letkf_ctx_timing_read(letkf_ctx* c, double* avg_ms, int64_t* nlaunch, int reset) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  double tot = 0.0;
  for (auto& ev : c->events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
    tot += ms;
  }
  if (nlaunch) *nlaunch = (int64_t)c->events.size();
  if (avg_ms) *avg_ms = c->events.empty() ? 0.0 : tot / (double)c->events.size();
  if (reset) {
    for (auto& ev : c->events) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    c->events.clear();
  }
  return LETKF_OK;
}

int letkf_core_batch_dev(letkf_ctx* c, const letkf_core_batch_args* g) {
  if (int rc = check_ctx(c)) return rc;
  if (!g) return fail(LETKF_E_INVALID, "args is NULL");
  if (g->nbatch == 0) return LETKF_OK;
  if (g->ne < 2 || g->nobs < 1 || g->nbatch < 0) return fail(LETKF_E_INVALID, "bad ne/nobs/nbatch");
  if (!g->nobsl || !g->hdxb || !g->rdiag || !g->rloc || !g->dep || !g->parm_infl || !g->trans)
    return fail(LETKF_E_INVALID, "a required device pointer is NULL");
  letkf::PointArgs a;
  std::memset(&a, 0, sizeof(a));
  a.k = g->ne;
  a.nv = 0;
  a.var_mask = ~0u;
  a.mode = 1;
  a.npts = g->nbatch;
  a.nobsl = g->nobsl;
  a.hdxb = g->hdxb;
  a.rdiag = g->rdiag;
  a.rloc = g->rloc;
  a.depv = g->dep;
  a.depd = (g->depd && g->transmd) ? g->depd : nullptr;       // common_letkf.f90:188
  a.nobs = g->nobs;
  a.rdiag_wloc = g->rdiag_wloc;
  a.infl_adaptive = g->infl_update;
  a.det_run = 0;
  a.infl = g->parm_infl;
  a.trans_out = g->trans;
  a.transm_out = g->transm;
  a.transmd_out = (g->depd && g->transmd) ? g->transmd : nullptr;
  a.pa_out = g->pao;
  a.add_wbar_to_trans = g->transm ? 0 : 1;                    // common_letkf.f90:218-226
  a.status = g->status;
  a.nsweep = g->nsweep;
  return launch(c, a);
}

namespace {

__global__ void zero_where_beta_is_zero(int64_t n, const double* __restrict__ beta, int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && beta[i] == 0.0) cnt[i] = 0;
}

// does any combined type carry a MAX_NOBS_PER_GRID limit?  From the host's hint when given, else read back (one sync)
int tables_limited(letkf_ctx* c, const letkf_search_tables* t, bool* limited) {
  if (t->limit_hint == 1 || t->limit_hint == 2) {
    *limited = t->limit_hint == 2;
    return LETKF_OK;
  }
  std::vector<int32_t> mx(t->nctype);
  HIP_TRY(hipMemcpyAsync(mx.data(), t->max_nobs, sizeof(int32_t) * t->nctype, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *limited = false;
  for (int ic = 0; ic < t->nctype; ++ic) *limited |= mx[ic] > 0;
  return LETKF_OK;
}

// shared by the list-driven and the fused-search entry
// (mode 3, letkf_das_columns_dev's list-free route: the points are pt0 + a * pt_stride + b, b < g->warm_stride columns whose
// horizontal survivors are sv[4 * sv_off[b] ..]; every per-point array of g is indexed by that GLOBAL point number)
struct SurvivorView {
  const int64_t* sv_off;
  const double* sv;
  int64_t pt_stride, pt0;
  int64_t cap;               // most survivors of a column of the batch (bounds a point's local list)
};
int das_points_impl(letkf_ctx* c, const letkf_das_args* g, const letkf_search_tables* t, const double* ri,
                    const double* rj, const double* rlev, const double* rz, int32_t* nobs_out,
                    const SurvivorView* sview = nullptr) {
  if (int rc = check_ctx(c)) return rc;
  if (!g) return fail(LETKF_E_INVALID, "args is NULL");
  if (g->npts == 0) return LETKF_OK;
  if (g->k < 2 || g->nv < 1 || g->npts < 0) return fail(LETKF_E_INVALID, "bad k/nv/npts");
  if ((!t && !g->obs_off) || !g->gues || !g->anal || !g->infl)
    return fail(LETKF_E_INVALID, "a required device pointer is NULL");
  if (g->kld < g->k + (g->det_run ? 1 : 0)) return fail(LETKF_E_INVALID, "kld too small for k (+1 with det_run)");
  if (g->iv_p < 0 || g->iv_p >= g->nv) {
    if (g->q_update_top > 0.0) return fail(LETKF_E_INVALID, "iv_p out of range");
  }
  letkf::PointArgs a;
  std::memset(&a, 0, sizeof(a));
  a.k = g->k;
  a.nv = g->nv;
  a.mode = 0;
  a.npts = g->npts;
  a.obs_off = reinterpret_cast<const long*>(g->obs_off);
  a.obs_idx = g->obs_idx;
  a.rdiag_l = g->rdiag_l;
  a.rloc_l = g->rloc_l;
  a.ensval = g->ensval;
  a.kld = g->kld;
  a.dep = g->dep;
  a.det_run = g->det_run;
  a.infl_adaptive = g->infl_adaptive;
  a.relax_to_inflated_prior = g->relax_to_inflated_prior;
  a.iv_p = g->iv_p;
  a.iv_q_first = g->iv_q_first;
  a.iv_q_last = g->iv_q_last;
  a.relax_alpha = g->relax_alpha;
  a.relax_alpha_spread = g->relax_alpha_spread;
  a.q_update_top = g->q_update_top;
  a.q_sprd_max = g->q_sprd_max;
  a.beta = g->beta;
  a.infl = g->infl;
  a.infl_sv = g->infl_sv > 0 ? g->infl_sv : g->npts;
  a.gues = g->gues;
  a.anal = g->anal;
  a.sp = g->sp;
  a.sm = g->sm;
  a.sv = g->sv;
  a.trans_out = g->trans_out;
  a.transm_out = g->transm_out;
  a.pa_out = g->pa_out;
  a.status = g->status;
  a.nsweep = g->nsweep;
  a.rtps_out = g->rtps_infl_out;
  a.var_mask = g->var_mask ? g->var_mask : ~0u;
  if (t && sview) {
    a.mode = 3;
    a.stab = *t;
    a.prlev = rlev;
    a.prz = rz;
    a.nobs_out = nobs_out;
    a.sv_off = reinterpret_cast<const long*>(sview->sv_off);
    a.surv = sview->sv;
    a.pt_stride = sview->pt_stride;
    a.pt0 = sview->pt0;
    a.sl_cap = (sview->cap + 3) & ~(int64_t)3;
  } else if (t) {
    if (!ri || !rj || !rlev || !rz) return fail(LETKF_E_INVALID, "a point coordinate array is NULL");
    if (t->nctype < 1 || t->ngroup < 1) return fail(LETKF_E_INVALID, "bad nctype / ngroup");
    if (!letkf::wave_kernel_supports(g->k, g->nv, 2))
      return fail(LETKF_E_INVALID, "the fused search needs the one-wave kernel (k <= 62, nv = 11): build lists with "
                                   "letkf_obs_search_dev and call letkf_das_points_dev instead");
    // only the no-limit mode of obs_local is fused (letkf_tools.f90:1438-1476)
    bool limited = false;
    if (int rc = tables_limited(c, t, &limited)) return rc;
    if (limited)
      return fail(LETKF_E_INVALID, "MAX_NOBS_PER_GRID > 0: build lists with letkf_obs_search_columns_dev / "
                                   "letkf_obs_search_dev (radix select) and call letkf_das_points_dev");
    if (g->trans_out || g->pa_out) return fail(LETKF_E_INVALID, "the fused search has no k x k outputs");
    a.mode = 2;
    a.stab = *t;
    a.pri = ri;
    a.prj = rj;
    a.prlev = rlev;
    a.prz = rz;
    a.nobs_out = nobs_out;
  }
  return launch(c, a, g->warm_run < 0 ? 0 : g->warm_run, g->warm_stride);
}

}  // namespace

int letkf_das_points_dev(letkf_ctx* c, const letkf_das_args* g) {
  return das_points_impl(c, g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int letkf_das_points_fused_dev(letkf_ctx* c, const letkf_das_args* g, const letkf_search_tables* t, const double* ri,
                               const double* rj, const double* rlev, const double* rz, int32_t* nobs_out) {
  if (!t) return fail(LETKF_E_INVALID, "tables is NULL");
  return das_points_impl(c, g, t, ri, rj, rlev, rz, nobs_out);
}

// (3c) das_letkf's main loop for a whole subdomain: column search + loop body by slabs of levels whose lists fit a
// workspace of the library (scale/letkf/letkf_tools.f90:313, the level loop)
int letkf_das_columns_dev(letkf_ctx* c, const letkf_das_args* g, const letkf_search_tables* t, int64_t nij1, int32_t nlev,
                          const double* rig, const double* rjg, const double* rlev, const double* rz, int64_t list_bytes,
                          int32_t* nobs_out) {
  if (int rc = check_ctx(c)) return rc;
  if (!g || !t) return fail(LETKF_E_INVALID, "args / tables is NULL");
  if (nij1 < 1 || nlev < 1 || g->npts != nij1 * (int64_t)nlev) return fail(LETKF_E_INVALID, "npts must be nij1 * nlev");
  if (!rig || !rjg || !rlev || !rz) return fail(LETKF_E_INVALID, "a point coordinate array is NULL");
  if (g->trans_out || g->transm_out || g->pa_out) return fail(LETKF_E_INVALID, "per-point k x k / w-bar outputs: use letkf_das_points_dev");
  const int64_t npts = g->npts;
  if (list_bytes <= 0) list_bytes = (int64_t)8 << 30;
  // ---- the list-free route: where the one-wave kernel serves the call and no combined type has a limit, the horizontal half of
  // obs_local is done once per COLUMN (32 B per survivor) and the vertical half inside the loop body kernel -- no count pass
  // over the levels, no 20 B per (point, observation) written and read back.  Same weights, same order, same analysis to the
  // last bit as the lists give (tests/test_gpu_columns.py).  LETKF_OPT_COLUMN_SURVIVORS = 0 keeps the lists.
  if (c->col_survivors && g->k >= 2 && letkf::wave_kernel_supports(g->k, g->nv, 3) && nij1 <= 0x7fffffff) {
    if (t->nctype < 1 || t->ngroup < 1) return fail(LETKF_E_INVALID, "bad nctype / ngroup");
    bool limited = false;
    if (int rc = tables_limited(c, t, &limited)) return rc;
    if (!limited) {
      // workspace: counts [nij1 + 1] int32 | sv_off [nij1 + 1] int64 | scan scratch
      size_t scan_b = 0;
      {
        auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(nullptr), [] __device__(int32_t v) { return (int64_t)v; });
        HIP_TRY(rocprim::exclusive_scan(nullptr, scan_b, in, static_cast<int64_t*>(nullptr), (int64_t)0, (size_t)nij1 + 1,
                                        rocprim::plus<int64_t>(), c->stream));
      }
      const size_t o_off = ((size_t)(nij1 + 1) * 4 + 255) & ~(size_t)255;
      const size_t o_scan = o_off + (((size_t)(nij1 + 1) * 8 + 255) & ~(size_t)255);
      if (o_scan + scan_b > c->scratch_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
      if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, o_scan + scan_b + 256)) return rc;
      int32_t* cnt = reinterpret_cast<int32_t*>(c->scratch);
      int64_t* soff = reinterpret_cast<int64_t*>(c->scratch + o_off);
      HIP_TRY(hipMemsetAsync(cnt + nij1, 0, 4, c->stream));
      HIP_TRY(letkf::launch_survivors(*t, 0, nij1, rig, rjg, 0, cnt, nullptr, nullptr, c->num_cu, c->stream));
      {
        auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(cnt), [] __device__(int32_t v) { return (int64_t)v; });
        HIP_TRY(rocprim::exclusive_scan(c->scratch + o_scan, scan_b, in, soff, (int64_t)0, (size_t)nij1 + 1, rocprim::plus<int64_t>(),
                                        c->stream));
      }
      std::vector<int64_t> hoff((size_t)nij1 + 1);
      HIP_TRY(hipMemcpyAsync(hoff.data(), soff, ((size_t)nij1 + 1) * 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      // (2 = automatic: the list-free route where the lists of all levels would not fit the workspace at once -- about half
      // of a column's horizontal survivors pass a level's vertical cut-off, 20 B each.  Where they fit, one fill pass for
      // the whole domain is cheaper than the vertical half inside the register-bound loop body kernel: C2, 203 local
      // observations per point, 384 against 394 ms per analysis; BASELINE configs[3], 4900 per point, 40 slabs: 7.45 against 6.11 s.)
      const bool take = c->col_survivors == 1 || (double)hoff[nij1] * (double)nlev * 10.0 > (double)list_bytes;
      // batches of columns whose survivors fit the workspace (32 B each), at least one column
      int64_t c0 = take ? 0 : nij1;
      while (c0 < nij1) {
        int64_t c1 = c0 + 1;
        while (c1 < nij1 && (hoff[c1 + 1] - hoff[c0]) * 32 <= list_bytes) ++c1;
        const int64_t nsv = hoff[c1] - hoff[c0];
        const size_t need = (size_t)(nsv > 0 ? nsv : 1) * 32 + 256;
        if (need > c->list_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));   // (the previous batch's solve may still read the old buffer)
        if (int rc = ensure_bytes(c, &c->list_ws, &c->list_ws_bytes, need)) return rc;
        // entry e of column b is addressed as sv[4 * sv_off[b] + ...] with the GLOBAL offsets: shift the base
        double* sv = reinterpret_cast<double*>(c->list_ws) - 4 * hoff[c0];
        HIP_TRY(letkf::launch_survivors(*t, c0, c1 - c0, rig, rjg, 1, nullptr, reinterpret_cast<const long*>(soff + c0), sv, c->num_cu,
                                        c->stream));
        letkf_das_args a = *g;
        a.npts = (c1 - c0) * (int64_t)nlev;
        a.infl_sv = g->infl_sv > 0 ? g->infl_sv : npts;
        a.warm_stride = (int32_t)(c1 - c0);                   // runs up the columns
        int64_t cap = 0;
        for (int64_t cc = c0; cc < c1; ++cc) cap = std::max(cap, hoff[cc + 1] - hoff[cc]);
        SurvivorView sview{soff + c0, sv, nij1, c0, cap};
        if (int rc = das_points_impl(c, &a, t, nullptr, nullptr, rlev, rz, nobs_out, &sview)) return rc;
        c0 = c1;
      }
      if (take) return LETKF_OK;
    }
  }
  // (the searches below -- one count pass, a fill pass per slab -- share the ring-ordered survivors of the dense limited case)
  struct RingKeep {
    letkf_ctx* c;
    explicit RingKeep(letkf_ctx* c_) : c(c_) { c->ring_keep = true; c->ring_ready = false; c->ring_no_n = -1; }
    ~RingKeep() {
      c->ring_keep = false;
      c->ring_ready = false;
      c->ring_no_n = -1;
      // the kept survivors can be a large part of the device (configs[3] with two limited types: 128 GiB).  By default the buffer
      // stays with the context for the next analysis (allocating and freeing 64 GB per call cost the MEMBER = 100 tile 1.7 s of a
      // 4 s analysis); LETKF_OPT_RING_RELEASE = 1 hands back whatever exceeds the batch budget when the entry returns, for a host
      // model that needs the memory between analyses (hipFree waits for the work that still reads the buffer)
      if (c->ring_release && c->ring_ws && c->ring_ws_bytes > ((size_t)c->ring_batch_mb << 20) + ((size_t)c->ring_batch_mb << 18) + 8192) {
        (void)hipFree(c->ring_ws);
        c->ring_ws = nullptr;
        c->ring_ws_bytes = 0;
      }
    }
  } ring_keep_guard(c);
  // workspace: counts [npts] int32 | obs_off [npts + 1] int64 | scan scratch
  size_t scan_bytes = 0;
  {
    auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(nullptr), [] __device__(int32_t v) { return (int64_t)v; });
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, in, static_cast<int64_t*>(nullptr), (int64_t)0, (size_t)npts + 1,
                                    rocprim::plus<int64_t>(), c->stream));
  }
  const size_t off_counts = 0, off_off = ((size_t)(npts + 1) * 4 + 255) & ~(size_t)255;
  const size_t off_scan = off_off + (((size_t)(npts + 1) * 8 + 255) & ~(size_t)255);
  if (off_scan + scan_bytes > c->scratch_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, off_scan + scan_bytes + 256)) return rc;
  int32_t* counts = reinterpret_cast<int32_t*>(c->scratch + off_counts);
  int64_t* off = reinterpret_cast<int64_t*>(c->scratch + off_off);
  // ---- count pass over all levels, prefix sum, level boundaries back to the host
  HIP_TRY(hipMemsetAsync(counts + npts, 0, 4, c->stream));   // (the scan runs over npts + 1 entries: the last one is the total)
  if (int rc = letkf_obs_search_columns_dev(c, t, nij1, nlev, rig, rjg, rlev, rz, 0, counts, nullptr, nullptr, nullptr, nullptr,
                                            nullptr, nullptr))
    return rc;
  {
    auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(counts), [] __device__(int32_t v) { return (int64_t)v; });
    HIP_TRY(rocprim::exclusive_scan(c->scratch + off_scan, scan_bytes, in, off, (int64_t)0, (size_t)npts + 1,
                                    rocprim::plus<int64_t>(), c->stream));
  }
  if (nobs_out) {
    HIP_TRY(hipMemcpyAsync(nobs_out, counts, (size_t)npts * 4, hipMemcpyDeviceToDevice, c->stream));
    // (as the list-free route reports them: the reference does not run obs_local where beta = 0, letkf_tools.f90:333-359)
    if (g->beta) {
      hipLaunchKernelGGL(zero_where_beta_is_zero, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, c->stream, npts, g->beta, nobs_out);
      HIP_TRY(hipGetLastError());
    }
  }
  std::vector<int64_t> lev_off((size_t)nlev + 1);
  HIP_TRY(hipMemcpy2DAsync(lev_off.data(), 8, off, (size_t)nij1 * 8, 8, (size_t)nlev + 1, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // ---- slabs of levels: as many as fit the list workspace (20 B per entry), at least one
  int l0 = 0;
  while (l0 < nlev) {
    int l1 = l0 + 1;
    while (l1 < nlev && (lev_off[l1 + 1] - lev_off[l0]) * 20 <= list_bytes) ++l1;
    const int64_t nnz = lev_off[l1] - lev_off[l0], p0 = (int64_t)l0 * nij1, np = (int64_t)(l1 - l0) * nij1;
    const size_t n1 = (size_t)(nnz > 0 ? nnz : 1);
    const size_t o_rd = (n1 * 4 + 255) & ~(size_t)255, o_rl = o_rd + ((n1 * 8 + 255) & ~(size_t)255);
    const size_t need = o_rl + n1 * 8 + 256;
    if (need > c->list_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));   // (the previous slab's solve may still read the old buffer)
    if (int rc = ensure_bytes(c, &c->list_ws, &c->list_ws_bytes, need)) return rc;
    // the kernels address list entry e of point p as base[obs_off[p] + j] with the GLOBAL offsets: shift the bases
    int32_t* idx = reinterpret_cast<int32_t*>(c->list_ws) - lev_off[l0];
    double* rd = reinterpret_cast<double*>(c->list_ws + o_rd) - lev_off[l0];
    double* rl = reinterpret_cast<double*>(c->list_ws + o_rl) - lev_off[l0];
    if (int rc = letkf_obs_search_columns_dev(c, t, nij1, l1 - l0, rig, rjg, rlev + p0, rz + p0, 1, nullptr, off + p0, idx, rd, rl,
                                              nullptr, nullptr))
      return rc;
    letkf_das_args a = *g;
    a.npts = np;
    a.obs_off = off + p0;
    a.obs_idx = idx;
    a.rdiag_l = rd;
    a.rloc_l = rl;
    a.gues = g->gues + p0 * g->sp;
    a.anal = g->anal + p0 * g->sp;
    if (g->beta) a.beta = g->beta + p0;
    a.infl = g->infl + p0;
    a.infl_sv = g->infl_sv > 0 ? g->infl_sv : npts;
    if (g->status) a.status = g->status + p0;
    if (g->nsweep) a.nsweep = g->nsweep + p0;
    if (g->rtps_infl_out) a.rtps_infl_out = g->rtps_infl_out + p0;
    a.warm_stride = (l1 - l0 > 1) ? (int32_t)nij1 : 0;      // runs up the columns of the slab
    if (nij1 > 0x7fffffff) a.warm_stride = 0;
    if (int rc = das_points_impl(c, &a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc;
    l0 = l1;
  }
  return LETKF_OK;
}

int letkf_ens_to_perturbations_dev(letkf_ctx* c, int32_t k, int32_t nv, int64_t npts, double* x, int64_t sp,
                                   int64_t sm, int64_t sv) {
  if (int rc = check_ctx(c)) return rc;
  if (!x || k < 1 || nv < 1 || npts < 0) return fail(LETKF_E_INVALID, "bad argument");
  if (npts == 0) return LETKF_OK;
  HIP_TRY(letkf::launch_ens_to_pert(k, nv, npts, x, sp, sm, sv, c->stream));
  return LETKF_OK;
}

int letkf_ens_mean_dev(letkf_ctx* c, int32_t k, int32_t nv, int64_t npts, double* x, int64_t sp, int64_t sm,
                       int64_t sv) {
  if (int rc = check_ctx(c)) return rc;
  if (!x || k < 1 || nv < 1 || npts < 0) return fail(LETKF_E_INVALID, "bad argument");
  if (npts == 0) return LETKF_OK;
  HIP_TRY(letkf::launch_ens_mean(k, nv, npts, x, sp, sm, sv, c->stream));
  return LETKF_OK;
}

int letkf_obs_search_dev(letkf_ctx* c, const letkf_search_tables* t, int64_t npts, const double* ri, const double* rj,
                         const double* rlev, const double* rz, int32_t fill, int32_t* counts, const int64_t* obs_off,
                         int32_t* obs_idx, double* rdiag_l, double* rloc_l) {
  if (int rc = check_ctx(c)) return rc;
  if (!t || npts < 0) return fail(LETKF_E_INVALID, "tables is NULL or npts < 0");
  if (npts == 0) return LETKF_OK;
  if (!ri || !rj || !rlev || !rz) return fail(LETKF_E_INVALID, "a point coordinate array is NULL");
  if (t->nctype < 1 || t->ngroup < 1 || t->criterion < 1 || t->criterion > 3)
    return fail(LETKF_E_INVALID, "bad nctype / ngroup / criterion");
  if (fill ? (!obs_off || !obs_idx || !rdiag_l || !rloc_l) : !counts)
    return fail(LETKF_E_INVALID, "missing output array for this phase");
  letkf::SearchArgs a;
  a.t = *t;
  a.npts = npts;
  a.ri = ri;
  a.rj = rj;
  a.rlev = rlev;
  a.rz = rz;
  a.fill = fill;
  a.counts = counts;
  a.obs_off = reinterpret_cast<const long*>(obs_off);
  a.obs_idx = obs_idx;
  a.rdiag_l = rdiag_l;
  a.rloc_l = rloc_l;
  {   // MAX_NOBS_PER_GRID anywhere?  (the fill phase then gets its LDS candidate cache)
    bool limited = false;
    if (int rc = tables_limited(c, t, &limited)) return rc;
    a.limited = limited ? 1 : 0;
  }
  HIP_TRY(letkf::launch_search(a, c->num_cu, c->stream));
  return LETKF_OK;
}

namespace {
// The limited column search on DENSE observations (letkf_search.hip, rings).  *taken = false: not eligible / not dense -- the
// caller goes on with the LDS-buffered column kernel.
int search_columns_rings(letkf_ctx* c, const letkf_search_tables* t, int64_t nij1, int32_t nlev, const double* rig,
                         const double* rjg, const double* rlev, const double* rz, int32_t fill, int32_t* counts,
                         const int64_t* obs_off, int32_t* obs_idx, double* rdiag_l, double* rloc_l, int32_t* nobs_ctype,
                         double* cutd_ctype, bool* taken) {
  *taken = false;
  if (c->limited_rings == 0 || t->criterion > 3 || t->nctype > 64 || nij1 * (int64_t)t->ngroup >= 0x7fffffff) return LETKF_OK;
  const void* key[5] = {t->ob_ri, t->ac_ext, t->max_nobs, rig, rjg};
  if (c->limited_rings == 2 && c->ring_no_n == nij1 && c->ring_no_crit == t->criterion && std::equal(key, key + 5, c->ring_no)) {
    if (!c->ring_keep) c->ring_no_n = -1;   // (a count -> fill pair: used once)
    return LETKF_OK;
  }
  c->ring_no_n = -1;
  std::vector<int32_t> mx(t->nctype), gstart(t->ngroup + 1);
  HIP_TRY(hipMemcpyAsync(gstart.data(), t->group_start, sizeof(int32_t) * (t->ngroup + 1), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(mx.data(), t->max_nobs, sizeof(int32_t) * t->nctype, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::vector<int32_t> gmem(gstart[t->ngroup]);
  HIP_TRY(hipMemcpy(gmem.data(), t->group_member, sizeof(int32_t) * gmem.size(), hipMemcpyDeviceToHost));
  int nlim = 0;
  std::vector<double> vl;
  // The weight criterion orders like the distance where a group has ONE variable-localisation factor: the plain rings serve.
  // Several factors in a group, and the error criterion (3), take the GENERAL ring key (r4, letkf_search.hip ring_offset): an
  // offset per entry, the group's smallest one as its reference.
  bool gen = t->criterion == 3;
  if (t->criterion >= 2) {
    vl.resize(t->nctype);
    HIP_TRY(hipMemcpy(vl.data(), t->varloc, sizeof(double) * t->nctype, hipMemcpyDeviceToHost));
  }
  for (int g = 0; g < t->ngroup; ++g) {
    const int nm = mx[gmem[gstart[g]]];
    if (nm > letkf::search_rings_max_nobs()) return LETKF_OK;
    nlim += nm > 0;
    if (nm > 0 && t->criterion == 2)
      for (int m = gstart[g] + 1; m < gstart[g + 1]; ++m)
        if (vl[gmem[m]] != vl[gmem[gstart[g]]]) gen = true;
  }
  if (nlim == 0) return LETKF_OK;
  const int ng = t->ngroup;
  const size_t ncg = (size_t)nij1 * ng;
  // aux: counts [ncg + 1] int32 | goff [ncg + 1] int64 | scan scratch | roff [batch]
  size_t scan_b = 0;
  {
    auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(nullptr), [] __device__(int32_t v) { return (int64_t)v; });
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_b, in, static_cast<int64_t*>(nullptr), (int64_t)0, ncg + 1, rocprim::plus<int64_t>(),
                                    c->stream));
  }
  const size_t o_off = ((ncg + 1) * 4 + 255) & ~(size_t)255, o_scan = o_off + (((ncg + 1) * 8 + 255) & ~(size_t)255);
  const size_t o_roff = o_scan + ((scan_b + 255) & ~(size_t)255);
  const size_t nring1 = (size_t)letkf::search_rings_count() + 1;   // ring starts per (column, group)
  const size_t roff_b = (ncg * nring1 * 4 + 255) & ~(size_t)255;
  const size_t o_kref = o_roff + roff_b;                            // kref [ngroup] | min err [nctype] (general ring key)
  const size_t need_aux = o_kref + ((size_t)ng + (size_t)t->nctype) * 8 + 256;
  if (need_aux > c->ring_aux_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->ring_aux, &c->ring_aux_bytes, need_aux)) return rc;
  int32_t* cnt = reinterpret_cast<int32_t*>(c->ring_aux);
  int64_t* goff = reinterpret_cast<int64_t*>(c->ring_aux + o_off);
  int32_t* roff = reinterpret_cast<int32_t*>(c->ring_aux + o_roff);
  double* kref = nullptr;
  if (gen) {
    // reference offsets: the smallest offset an entry of the group can have -- criterion 2: -2 ln(largest factor); criterion 3:
    // 2 ln(smallest error^2 / factor) over the group's types (the smallest error of a type: one small kernel + a read-back)
    kref = reinterpret_cast<double*>(c->ring_aux + o_kref);
    std::vector<double> emin(t->nctype, 1.0), kr(ng);
    if (t->criterion == 3) {
      HIP_TRY(letkf::launch_ctype_min_err(*t, kref + ng, c->stream));
      HIP_TRY(hipMemcpyAsync(emin.data(), kref + ng, sizeof(double) * t->nctype, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int g = 0; g < ng; ++g) {
      double lo = 1e300;
      for (int m = gstart[g]; m < gstart[g + 1]; ++m) {
        const int ic = gmem[m];
        if (!(vl[ic] > 0.0)) continue;
        const double off = t->criterion == 2 ? -2.0 * std::log(vl[ic]) : 2.0 * std::log(emin[ic] * emin[ic] / vl[ic]);
        if (off == off && off < lo) lo = off;
      }
      kr[g] = lo < 1e299 ? lo : 0.0;
    }
    HIP_TRY(hipMemcpyAsync(kref, kr.data(), sizeof(double) * ng, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // (kr goes out of scope)
  }
  if (c->ring_keep && c->ring_ready) {
    // (a later call of the same letkf_das_columns_dev: same tables, same columns -- the ring-ordered survivors are still there)
    *taken = true;
    HIP_TRY(letkf::launch_search_rings(*t, 0, nij1, nij1, nlev, rlev, rz, fill, counts, reinterpret_cast<const long*>(obs_off), obs_idx,
                                       rdiag_l, rloc_l, nobs_ctype, cutd_ctype, reinterpret_cast<const long*>(goff),
                                       reinterpret_cast<double*>(c->ring_ws), roff, kref, c->num_cu, c->stream));
    return LETKF_OK;
  }
  HIP_TRY(hipMemsetAsync(cnt + ncg, 0, 4, c->stream));
  HIP_TRY(letkf::launch_ring_survivors(*t, 0, nij1, rig, rjg, 0, cnt, nullptr, nullptr, nullptr, nullptr, c->num_cu, c->stream));
  {
    auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(cnt), [] __device__(int32_t v) { return (int64_t)v; });
    HIP_TRY(rocprim::exclusive_scan(c->ring_aux + o_scan, scan_b, in, goff, (int64_t)0, ncg + 1, rocprim::plus<int64_t>(), c->stream));
  }
  std::vector<int64_t> hoff(ncg + 1);
  HIP_TRY(hipMemcpyAsync(hoff.data(), goff, (ncg + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->limited_rings == 2) {
    // dense = more than one in twenty (column, limited group) pairs overflow the column kernel's LDS buffer and would take its
    // multi-sweep fall-back (20 x the cost of a pair that fits); while they fit, that kernel -- everything of a column resident,
    // all levels against it -- is the faster one (C2's grid under a limit of 100: ~450 survivors per pair, 34 against 68 ms)
    size_t n_over = 0, n_lim = 0;
    for (size_t i = 0; i < ncg; ++i)
      if (mx[gmem[gstart[i % ng]]] > 0) {
        ++n_lim;
        n_over += (hoff[i + 1] - hoff[i]) > (int64_t)letkf::search_rings_lds_survivors();
      }
    if (n_over * 20 <= n_lim) {
      if (c->ring_keep || !fill) {   // (remembered for the fill call of this pair / the later calls of this letkf_das_columns_dev)
        std::copy(key, key + 5, c->ring_no);
        c->ring_no_n = nij1;
        c->ring_no_crit = t->criterion;
      }
      return LETKF_OK;
    }
  }
  *taken = true;
  // LETKF_OPT_RING_BATCH_MB (8 GiB) of survivors per batch of columns; inside letkf_das_columns_dev ONE batch, kept for the calls that follow, where
  // that takes no more than half of the device memory still free (configs[3] with two limited types: 128 GiB)
  bool keep = false;
  if (c->ring_keep) {
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    const size_t want = (size_t)hoff[ncg] * 32 + 256;
    keep = want <= c->ring_ws_bytes + fr / 2;
    if (keep && want > c->ring_ws_bytes) {
      // the exact size (ensure_bytes would ask for a quarter more), and a failure is no error: the batches below need 8 GiB
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->ring_ws) HIP_TRY(hipFree(c->ring_ws));
      c->ring_ws = nullptr;
      c->ring_ws_bytes = 0;
      if (hipMalloc(reinterpret_cast<void**>(&c->ring_ws), want) == hipSuccess) {
        c->ring_ws_bytes = want;
      } else {
        (void)hipGetLastError();
        c->ring_ws = nullptr;
        keep = false;
      }
    }
  }
  const int64_t budget = keep ? hoff[ncg] * 32 + 256 : ((int64_t)c->ring_batch_mb << 20);
  int64_t c0 = 0;
  while (c0 < nij1) {
    int64_t c1 = c0 + 1;
    while (c1 < nij1 && (hoff[(size_t)(c1 + 1) * ng] - hoff[(size_t)c0 * ng]) * 32 <= budget) ++c1;
    const int64_t nsv = hoff[(size_t)c1 * ng] - hoff[(size_t)c0 * ng];
    const size_t need = (size_t)(nsv > 0 ? nsv : 1) * 32 + 256;
    if (need > c->ring_ws_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
    if (int rc = ensure_bytes(c, &c->ring_ws, &c->ring_ws_bytes, need)) return rc;
    double* sv = reinterpret_cast<double*>(c->ring_ws) - 4 * hoff[(size_t)c0 * ng];
    const long* gq = reinterpret_cast<const long*>(goff + (size_t)c0 * ng);
    int32_t* rq = roff + (size_t)c0 * ng * nring1;
    HIP_TRY(letkf::launch_ring_survivors(*t, c0, c1 - c0, rig, rjg, 1, nullptr, gq, sv, rq, kref, c->num_cu, c->stream));
    HIP_TRY(letkf::launch_search_rings(*t, c0, c1 - c0, nij1, nlev, rlev, rz, fill, counts, reinterpret_cast<const long*>(obs_off),
                                       obs_idx, rdiag_l, rloc_l, nobs_ctype, cutd_ctype, gq, sv, rq, kref, c->num_cu, c->stream));
    c0 = c1;
  }
  c->ring_ready = keep;
  return LETKF_OK;
}
}  // namespace

int letkf_obs_search_columns_dev(letkf_ctx* c, const letkf_search_tables* t, int64_t nij1, int32_t nlev,
                                 const double* rig, const double* rjg, const double* rlev, const double* rz,
                                 int32_t fill, int32_t* counts, const int64_t* obs_off, int32_t* obs_idx,
                                 double* rdiag_l, double* rloc_l, int32_t* nobs_ctype, double* cutd_ctype) {
  if (int rc = check_ctx(c)) return rc;
  if (!t || nij1 < 0 || nlev < 1) return fail(LETKF_E_INVALID, "tables is NULL or bad nij1 / nlev");
  if (nij1 == 0) return LETKF_OK;
  if (!rig || !rjg || !rlev || !rz) return fail(LETKF_E_INVALID, "a point coordinate array is NULL");
  if (t->nctype < 1 || t->ngroup < 1 || t->criterion < 1 || t->criterion > 3)
    return fail(LETKF_E_INVALID, "bad nctype / ngroup / criterion");
  if (fill ? (!obs_off || !obs_idx || !rdiag_l || !rloc_l) : !counts)
    return fail(LETKF_E_INVALID, "missing output array for this phase");
  if ((size_t)4 * (4 * 512 + 2 * ((nlev + 1) & ~1)) * sizeof(double) > c->lds_max ||
      (size_t)4 * (4 * 576 + ((nlev + 1) & ~1)) * sizeof(double) + 4608 > c->lds_max)
    return fail(LETKF_E_INVALID, "too many levels for the column kernel's LDS counters");
  bool limited = false;
  if (int rc = tables_limited(c, t, &limited)) return rc;
  if (limited) {
    bool taken = false;
    if (int rc = search_columns_rings(c, t, nij1, nlev, rig, rjg, rlev, rz, fill, counts, obs_off, obs_idx, rdiag_l, rloc_l,
                                      nobs_ctype, cutd_ctype, &taken))
      return rc;
    if (taken) return LETKF_OK;
  }
  if (limited || cutd_ctype)
    HIP_TRY(letkf::launch_search_columns_limited(*t, nij1, nlev, rig, rjg, rlev, rz, fill, counts,
                                                 reinterpret_cast<const long*>(obs_off), obs_idx, rdiag_l, rloc_l,
                                                 nobs_ctype, cutd_ctype, c->num_cu, c->stream));
  else
    HIP_TRY(letkf::launch_search_columns(*t, nij1, nlev, rig, rjg, rlev, rz, fill, counts,
                                         reinterpret_cast<const long*>(obs_off), obs_idx, rdiag_l, rloc_l, nobs_ctype,
                                         c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_obs_departure_dev(letkf_ctx* c, const letkf_qc_params* p, int64_t nobs, const int32_t* elm, const double* dat,
                            const double* err, double* ensval, int64_t kld, double* val, int32_t* qc) {
  if (int rc = check_ctx(c)) return rc;
  if (!p || nobs < 0) return fail(LETKF_E_INVALID, "params is NULL or nobs < 0");
  if (nobs == 0) return LETKF_OK;
  if (!elm || !dat || !err || !ensval || !val || !qc) return fail(LETKF_E_INVALID, "an observation array is NULL");
  if (p->member < 1 || kld < p->member + (p->det_run ? 1 : 0))
    return fail(LETKF_E_INVALID, "kld must hold MEMBER (+1 with DET_RUN) columns");
  if ((size_t)64 * (size_t)(kld | 1) * sizeof(double) > c->lds_max) return fail(LETKF_E_INVALID, "kld too large");
  HIP_TRY(letkf::launch_obs_departure(*p, nobs, elm, dat, err, ensval, kld, val, qc, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_obs_mesh_sort_dev(letkf_ctx* c, const letkf_mesh* m, int64_t nobs, const int32_t* ctype, const double* ri,
                            const double* rj, const int32_t* qc, int32_t* n_cell, int32_t* key, int64_t* nsorted) {
  if (int rc = check_ctx(c)) return rc;
  if (!m || nobs < 0 || !nsorted) return fail(LETKF_E_INVALID, "mesh / nsorted is NULL or nobs < 0");
  if (m->nctype < 1 || !m->ngrd_i || !m->ngrd_j || m->nlon < 1 || m->nlat < 1)
    return fail(LETKF_E_INVALID, "bad mesh description");
  if (nobs > 0 && (!ctype || !ri || !rj || !qc || !key)) return fail(LETKF_E_INVALID, "an observation array is NULL");
  if (!n_cell) return fail(LETKF_E_INVALID, "n_cell is NULL");
  if (nobs >= (1LL << 31)) return fail(LETKF_E_INVALID, "more than 2^31 local observations");
  size_t need = 0;
  long ns = 0;
  HIP_TRY(letkf::obs_mesh_sort(*m, nobs, ctype, ri, rj, qc, n_cell, key, &ns, nullptr, &need, c->num_cu, c->stream));
  if (need > c->scratch_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, need)) return rc;
  size_t have = c->scratch_bytes;
  HIP_TRY(letkf::obs_mesh_sort(*m, nobs, ctype, ri, rj, qc, n_cell, key, &ns, c->scratch, &have, c->num_cu, c->stream));
  *nsorted = ns;
  return LETKF_OK;
}

int letkf_obs_halo_plan_dev(letkf_ctx* c, const letkf_halo_layout* l, const int32_t* n_all, int32_t* ac_ext,
                            int32_t* src_row, int64_t cap, int64_t* nobstotal) {
  if (int rc = check_ctx(c)) return rc;
  if (!l || !nobstotal || !n_all || !ac_ext) return fail(LETKF_E_INVALID, "a required pointer is NULL");
  if (l->nctype < 1 || l->nprocs < 1 || l->prc_num_x < 1 || l->myrank < 0 || l->myrank >= l->nprocs ||
      l->nprocs % l->prc_num_x != 0 || !l->ngrd_i || !l->ngrd_j || !l->ngrdsch_i || !l->ngrdsch_j)
    return fail(LETKF_E_INVALID, "bad rank layout / mesh description");
  if (cap > 0 && !src_row) return fail(LETKF_E_INVALID, "src_row is NULL");
  long nt = 0;
  hipError_t e = letkf::obs_halo_plan(*l, n_all, ac_ext, src_row, cap, &nt, c->num_cu, c->stream);
  *nobstotal = nt;
  if (e == hipErrorInvalidValue && nt > cap) return fail(LETKF_E_INVALID, "src_row capacity is smaller than nobstotal");
  HIP_TRY(e);
  return LETKF_OK;
}

int letkf_obs_gather_rows_dev(letkf_ctx* c, int64_t nrows, const int32_t* src_row, int32_t ncols, const double* src,
                              int64_t ld_src, double* dst, int64_t ld_dst) {
  if (int rc = check_ctx(c)) return rc;
  if (nrows < 0 || ncols < 0) return fail(LETKF_E_INVALID, "negative size");
  if (nrows == 0 || ncols == 0) return LETKF_OK;
  if (!src_row || !src || !dst || ld_src < ncols || ld_dst < ncols) return fail(LETKF_E_INVALID, "bad argument");
  HIP_TRY(letkf::launch_gather_rows(nrows, src_row, ncols, src, ld_src, dst, ld_dst, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_obs_gather_i32_dev(letkf_ctx* c, int64_t nrows, const int32_t* src_row, const int32_t* src, int32_t* dst) {
  if (int rc = check_ctx(c)) return rc;
  if (nrows < 0) return fail(LETKF_E_INVALID, "negative size");
  if (nrows == 0) return LETKF_OK;
  if (!src_row || !src || !dst) return fail(LETKF_E_INVALID, "bad argument");
  HIP_TRY(letkf::launch_gather_i32(nrows, src_row, src, dst, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_monit_dep_dev(letkf_ctx* c, int32_t nid, const int32_t* elem_uid, int64_t nn, const int32_t* elm,
                        const double* dep, const int32_t* qc, int32_t* nobs, double* bias, double* rmse) {
  if (int rc = check_ctx(c)) return rc;
  if (nid < 1 || nid > 32 || !elem_uid || nn < 0 || !nobs || !bias || !rmse)
    return fail(LETKF_E_INVALID, "bad element table / outputs");
  if (nn > 0 && (!elm || !dep || !qc)) return fail(LETKF_E_INVALID, "an observation array is NULL");
  const size_t need = letkf::monit_scratch_bytes(nid, c->num_cu);
  if (need > c->scratch_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, need)) return rc;
  HIP_TRY(letkf::launch_monit_dep(nid, elem_uid, nn, elm, dep, qc, nobs, bias, rmse, c->scratch, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_additive_inflation_dev(letkf_ctx* c, int32_t k, int32_t nv, int64_t npts, int64_t nij1, double* anal,
                                 const double* add, int64_t sp, int64_t sm, int64_t sv, double infl_add,
                                 const double* weight, const double* qmean, int64_t q_sp, int64_t q_sv,
                                 int32_t iv_q_first, int32_t iv_q_last, const int32_t* ishuf) {
  if (int rc = check_ctx(c)) return rc;
  if (k < 1 || nv < 1 || npts < 0 || nij1 < 1 || !anal || !add) return fail(LETKF_E_INVALID, "bad argument");
  if (npts % nij1 != 0) return fail(LETKF_E_INVALID, "npts must be nij1 * nlev");
  HIP_TRY(letkf::launch_additive(k, nv, npts, nij1, anal, add, sp, sm, sv, infl_add, weight, qmean, q_sp, q_sv,
                                 iv_q_first, iv_q_last, ishuf, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_addinfl_weight_dev(letkf_ctx* c, int64_t nij1, const double* rig, const double* rjg, int64_t nob,
                             const double* ob_ri, const double* ob_rj, double dx, double dy, double hori_loc,
                             double* weight) {
  if (int rc = check_ctx(c)) return rc;
  if (nij1 < 0 || nob < 0 || !(hori_loc > 0.0)) return fail(LETKF_E_INVALID, "bad argument");
  if (nij1 == 0) return LETKF_OK;
  if (!rig || !rjg || !weight || (nob > 0 && (!ob_ri || !ob_rj))) return fail(LETKF_E_INVALID, "a pointer is NULL");
  const double cut2 = (double)13.33333333f;   // dist_zero_fac_square, a single-precision literal (letkf_obs.f90:28)
  HIP_TRY(letkf::launch_addinfl_weight(nij1, rig, rjg, nob, ob_ri, ob_rj, dx, dy, hori_loc, cut2, weight, c->num_cu,
                                       c->stream));
  return LETKF_OK;
}

int letkf_obs_allgatherv_dev(letkf_ctx* c, void* nccl_comm, int32_t nranks, int32_t myrank, const int64_t* counts,
                             int64_t row_bytes, const void* send, void* recv) {
  if (int rc = check_ctx(c)) return rc;
  if (!nccl_comm || nranks < 1 || myrank < 0 || myrank >= nranks || !counts || row_bytes < 1)
    return fail(LETKF_E_INVALID, "bad communicator / rank layout / counts");
  int64_t total = 0;
  for (int r = 0; r < nranks; ++r) {
    if (counts[r] < 0) return fail(LETKF_E_INVALID, "negative row count");
    total += counts[r];
  }
  if ((counts[myrank] > 0 && !send) || (total > 0 && !recv)) return fail(LETKF_E_INVALID, "a buffer is NULL");
  const char* what = "";
  const int rc = letkf::rccl_allgatherv(nccl_comm, nranks, myrank, counts, row_bytes, send, recv, c->stream, &what);
  if (rc == -1) return fail(LETKF_E_INVALID, "RCCL (librccl.so.1) is not available in this process");
  if (rc != 0) return fail(LETKF_E_HIP, std::string("RCCL: ") + what);
  return LETKF_OK;
}

namespace {
int rccl_result(int rc, const char* what) {
  if (rc == 0) return LETKF_OK;
  if (rc == -1) return fail(LETKF_E_INVALID, "RCCL (librccl.so.1) is not available in this process");
  if (rc == -2) return fail(LETKF_E_INVALID, what);
  return fail(LETKF_E_HIP, std::string("RCCL: ") + what);
}
}  // namespace

int letkf_alltoallv_dev(letkf_ctx* c, void* nccl_comm, int32_t nranks, int32_t myrank, const int64_t* send_counts,
                        const int64_t* send_offs, const int64_t* recv_counts, const int64_t* recv_offs, int64_t row_bytes,
                        const void* send, void* recv) {
  if (int rc = check_ctx(c)) return rc;
  if ((nranks > 1 && !nccl_comm) || nranks < 1 || myrank < 0 || myrank >= nranks || !send_counts || !send_offs || !recv_counts ||
      !recv_offs || row_bytes < 1)
    return fail(LETKF_E_INVALID, "bad communicator / rank layout / counts");
  int64_t ns = 0, nr = 0;
  for (int r = 0; r < nranks; ++r) {
    if (send_counts[r] < 0 || recv_counts[r] < 0 || send_offs[r] < 0 || recv_offs[r] < 0) return fail(LETKF_E_INVALID, "negative count / offset");
    ns += send_counts[r];
    nr += recv_counts[r];
  }
  if ((ns > 0 && !send) || (nr > 0 && !recv)) return fail(LETKF_E_INVALID, "a buffer is NULL");
  const char* what = "";
  return rccl_result(letkf::rccl_alltoallv(nccl_comm, nranks, myrank, send_counts, send_offs, recv_counts, recv_offs, row_bytes, send,
                                           recv, c->stream, &what), what);
}

int letkf_allreduce_sum_i32_dev(letkf_ctx* c, void* nccl_comm, int32_t nranks, int64_t count, int32_t* buf) {
  if (int rc = check_ctx(c)) return rc;
  if ((nranks > 1 && !nccl_comm) || nranks < 1 || count < 0 || (count > 0 && !buf)) return fail(LETKF_E_INVALID, "bad argument");
  const char* what = "";
  return rccl_result(letkf::rccl_allreduce_sum_i32(nccl_comm, nranks, count, buf, c->stream, &what), what);
}

// scatter_grd_mpi_alltoall / gather_grd_mpi_alltoall (scale/common/common_mpi_scale.f90:1279-1396) with the exchange inside the
// library: per-destination blocks [nv3d][nlev * nij1(d)] dealt out of / assembled into the member field by the kernel of
// letkf_member_points_dev (grd_to_buf / buf_to_grd), ONE grouped exchange with true counts, the blocks filed into / taken from
// the member slots of the state.  Workspace: the context's scratch buffer (send blocks | receive blocks).
int letkf_members_alltoall_dev(letkf_ctx* c, void* nccl_comm, int32_t nranks, int32_t myrank, int32_t dir, int32_t nlev,
                               int32_t nlon, int32_t nlat, int32_t nv3d, int32_t mstart, int32_t mcount, double* v3dg, double* x,
                               int64_t sp, int64_t sm, int64_t sv) {
  if (int rc = check_ctx(c)) return rc;
  if ((nranks > 1 && !nccl_comm) || nranks < 1 || myrank < 0 || myrank >= nranks || nlev < 1 || nlon < 1 || nlat < 1 || nv3d < 1 ||
      mstart < 0 || mcount < 0 || mcount > nranks || !x || (dir != 0 && dir != 1))
    return fail(LETKF_E_INVALID, "bad argument");
  const bool holder = myrank < mcount;                    // this rank holds / receives the whole field of member mstart + myrank
  if (holder && !v3dg) return fail(LETKF_E_INVALID, "v3dg is NULL on a rank that holds a member");
  const long nxy = (long)nlon * nlat;
  auto share = [&](int r) { return (nxy - r + nranks - 1) / nranks; };   // points r, r + nranks, ... (grd_to_buf)
  const long nij1 = share(myrank), npl = (long)nlev * nij1;
  std::vector<int64_t> fc(nranks), fo(nranks), pc(nranks), po(nranks);   // field side (all points of my member), point side (my points of every member)
  int64_t ftot = 0, ptot = 0;
  for (int r = 0; r < nranks; ++r) {
    fc[r] = holder ? (int64_t)nv3d * nlev * share(r) : 0;
    fo[r] = ftot;
    ftot += fc[r];
    pc[r] = r < mcount ? (int64_t)nv3d * npl : 0;
    po[r] = ptot;
    ptot += pc[r];
  }
  const size_t need = (size_t)(ftot + ptot) * sizeof(double) + 256;
  if (need > c->scratch_bytes) HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, need)) return rc;
  double* fbuf = reinterpret_cast<double*>(c->scratch);
  double* pbuf = fbuf + ftot;
  const char* what = "";
  if (dir == 0) {   // member fields -> point-major state
    if (holder)
      for (int d = 0; d < nranks; ++d) {
        const long nd = share(d);
        if (nd > 0) HIP_TRY(letkf::launch_member_points(0, nlev, nlon, nxy, nv3d, nranks, d, nd, v3dg, fbuf + fo[d], 1, 0, nd * nlev, c->stream));
      }
    if (int rc = rccl_result(letkf::rccl_alltoallv(nccl_comm, nranks, myrank, fc.data(), fo.data(), pc.data(), po.data(), 8, fbuf, pbuf,
                                                   c->stream, &what), what))
      return rc;
    for (int s_ = 0; s_ < mcount; ++s_)
      HIP_TRY(letkf::launch_block_slot(0, npl, nv3d, pbuf + po[s_], x, sp, (long)(mstart + s_) * sm, sv, c->stream));
  } else {          // point-major state -> member fields
    for (int d = 0; d < mcount; ++d)
      HIP_TRY(letkf::launch_block_slot(1, npl, nv3d, pbuf + po[d], x, sp, (long)(mstart + d) * sm, sv, c->stream));
    if (int rc = rccl_result(letkf::rccl_alltoallv(nccl_comm, nranks, myrank, pc.data(), po.data(), fc.data(), fo.data(), 8, pbuf, fbuf,
                                                   c->stream, &what), what))
      return rc;
    if (holder)
      for (int s_ = 0; s_ < nranks; ++s_) {
        const long ns = share(s_);
        if (ns > 0) HIP_TRY(letkf::launch_member_points(1, nlev, nlon, nxy, nv3d, nranks, s_, ns, v3dg, fbuf + fo[s_], 1, 0, ns * nlev, c->stream));
      }
  }
  return LETKF_OK;
}

int letkf_relax_beta_dev(letkf_ctx* c, const letkf_beta_params* p, int64_t nij1, int32_t nlev, const double* rig,
                         const double* rjg, const double* hgt, double* beta) {
  if (int rc = check_ctx(c)) return rc;
  if (!p || nij1 < 0 || nlev < 1) return fail(LETKF_E_INVALID, "params is NULL or bad nij1 / nlev");
  if (nij1 == 0) return LETKF_OK;
  if (!rig || !rjg || !hgt || !beta) return fail(LETKF_E_INVALID, "a point array is NULL");
  HIP_TRY(letkf::launch_relax_beta(*p, nij1, nlev, rig, rjg, hgt, beta, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_infl_init_dev(letkf_ctx* c, int64_t n, double* work3d, double infl_mul, double infl_mul_min) {
  if (int rc = check_ctx(c)) return rc;
  if (n < 0) return fail(LETKF_E_INVALID, "negative size");
  if (n == 0) return LETKF_OK;
  if (!work3d) return fail(LETKF_E_INVALID, "work3d is NULL");
  HIP_TRY(letkf::launch_infl_init(n, work3d, infl_mul, infl_mul_min, c->num_cu, c->stream));
  return LETKF_OK;
}

int letkf_state_trans_dev(letkf_ctx* c, const letkf_state_consts* k, int32_t nlev, int32_t nlon, int32_t nlat,
                          int32_t nv3d, double* v3dg, int32_t inverse) {
  if (int rc = check_ctx(c)) return rc;
  if (!k || !v3dg || nlev < 1 || nlon < 1 || nlat < 1) return fail(LETKF_E_INVALID, "bad argument");
  if (k->iv_q < 0 || k->iv_q >= nv3d || nv3d - k->iv_q > 8) return fail(LETKF_E_INVALID, "moisture range must be 1..8 variables");
  HIP_TRY(letkf::launch_state_trans(*k, nlev, (long)nlon * nlat, nv3d, v3dg, inverse, c->stream));
  return LETKF_OK;
}

int letkf_member_points_dev(letkf_ctx* c, int32_t dir, int32_t nlev, int32_t nlon, int32_t nlat, int32_t nv3d,
                            int32_t np, int32_t rank, int32_t m, double* v3dg, double* x, int64_t nij1, int64_t sp,
                            int64_t sm, int64_t sv) {
  if (int rc = check_ctx(c)) return rc;
  if (!v3dg || !x || np < 1 || rank < 0 || rank >= np || m < 0 || nij1 < 0) return fail(LETKF_E_INVALID, "bad argument");
  const long nxy = (long)nlon * nlat;
  const long expect = (nxy - rank + np - 1) / np;               // points r, r+np, ... below nlon*nlat
  if (nij1 != expect) return fail(LETKF_E_INVALID, "nij1 does not match the cyclic share of this rank");
  if (nij1 == 0) return LETKF_OK;
  HIP_TRY(letkf::launch_member_points(dir, nlev, nlon, nxy, nv3d, np, rank, nij1, v3dg, x, sp, (long)m * sm, sv, c->stream));
  return LETKF_OK;
}

int letkf_ens_spread_dev(letkf_ctx* c, int32_t k, int32_t nv, int64_t npts, const double* x, int64_t sp, int64_t sm,
                         int64_t sv, double* sprd) {
  if (int rc = check_ctx(c)) return rc;
  if (!x || !sprd || k < 2 || nv < 1 || npts < 0) return fail(LETKF_E_INVALID, "bad argument");
  if (npts == 0) return LETKF_OK;
  HIP_TRY(letkf::launch_ens_spread(k, nv, npts, x, sp, sm, sv, sprd, c->stream));
  return LETKF_OK;
}

// ---------------------------------------------------------------------------------------------
// Fine boundary on host pointers: the drop-in for common/common_letkf.f90:52 used by the Fortran shim.
// One context per calling thread (the reference calls letkf_core from inside !$OMP PARALLEL).
// ---------------------------------------------------------------------------------------------
namespace {
struct TlsCtx {
  letkf_ctx* c = nullptr;
  ~TlsCtx() {
    if (c) letkf_ctx_destroy(c);
  }
};
thread_local TlsCtx g_tls;

int core_host(int ne, int nobs, int nobsl, const double* hdxb, const double* rdiag, const double* rloc,
              const double* dep, double* parm_infl, double* trans, double* transm, double* pao,
              const int* rdiag_wloc, const int* infl_update, const double* depd, double* transmd, int* st_out) {
  if (ne < 2 || nobs < 0 || nobsl < 0 || nobsl > nobs) return fail(LETKF_E_INVALID, "bad ne/nobs/nobsl");
  if (!parm_infl || !trans) return fail(LETKF_E_INVALID, "parm_infl/trans is NULL");
  if (nobsl > 0 && (!hdxb || !rdiag || !rloc || !dep)) return fail(LETKF_E_INVALID, "an input array is NULL");
  if (!g_tls.c) {
    if (int rc = letkf_ctx_create(-1, &g_tls.c)) return rc;
  }
  letkf_ctx* c = g_tls.c;
  if (int rc = check_ctx(c)) return rc;
  const size_t k = (size_t)ne, n = (size_t)(nobsl > 0 ? nobsl : 1);
  const bool det = depd && transmd;
  // scratch layout (doubles): hdxb[n*k] rdiag[n] rloc[n] dep[n] depd[n] infl[1] trans[k*k] pao[k*k] transm[k] transmd[k] | ints: nobsl, status
  const size_t nd = n * k + 4 * n + 1 + 2 * k * k + 2 * k;
  const size_t bytes = nd * sizeof(double) + 4 * sizeof(int);
  if (int rc = ensure_bytes(c, &c->scratch, &c->scratch_bytes, bytes)) return rc;
  double* d = reinterpret_cast<double*>(c->scratch);
  double* d_h = d;
  double* d_rdiag = d_h + n * k;
  double* d_rloc = d_rdiag + n;
  double* d_dep = d_rloc + n;
  double* d_depd = d_dep + n;
  double* d_infl = d_depd + n;
  double* d_trans = d_infl + 1;
  double* d_pao = d_trans + k * k;
  double* d_transm = d_pao + k * k;
  double* d_transmd = d_transm + k;
  int* d_i = reinterpret_cast<int*>(d_transmd + k);
  hipStream_t s = c->stream;
  if (nobsl > 0) {
    // only rows 1..nobsl of hdxb(nobs, ne) are meaningful (common_letkf.f90:35-37): compact while copying
    HIP_TRY(hipMemcpy2DAsync(d_h, (size_t)nobsl * sizeof(double), hdxb, (size_t)nobs * sizeof(double),
                             (size_t)nobsl * sizeof(double), k, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_rdiag, rdiag, (size_t)nobsl * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_rloc, rloc, (size_t)nobsl * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_dep, dep, (size_t)nobsl * sizeof(double), hipMemcpyHostToDevice, s));
    if (det) HIP_TRY(hipMemcpyAsync(d_depd, depd, (size_t)nobsl * sizeof(double), hipMemcpyHostToDevice, s));
  }
  int hi[2] = {nobsl, 0};
  HIP_TRY(hipMemcpyAsync(d_i, hi, sizeof(hi), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_infl, parm_infl, sizeof(double), hipMemcpyHostToDevice, s));
  letkf_core_batch_args g;
  std::memset(&g, 0, sizeof(g));
  g.ne = ne;
  g.nobs = (int)n;
  g.nbatch = 1;
  g.nobsl = d_i;
  g.hdxb = d_h;
  g.rdiag = d_rdiag;
  g.rloc = d_rloc;
  g.dep = d_dep;
  g.depd = det ? d_depd : nullptr;
  g.parm_infl = d_infl;
  g.trans = d_trans;
  g.transm = transm ? d_transm : nullptr;
  g.pao = pao ? d_pao : nullptr;
  g.transmd = det ? d_transmd : nullptr;
  g.rdiag_wloc = rdiag_wloc ? (*rdiag_wloc != 0) : 0;      // common_letkf.f90:84-85
  g.infl_update = infl_update ? (*infl_update != 0) : 0;   // :86-87
  g.status = d_i + 1;
  if (int rc = letkf_core_batch_dev(c, &g)) return rc;
  HIP_TRY(hipMemcpyAsync(trans, d_trans, k * k * sizeof(double), hipMemcpyDeviceToHost, s));
  if (transm) HIP_TRY(hipMemcpyAsync(transm, d_transm, k * sizeof(double), hipMemcpyDeviceToHost, s));
  if (pao) HIP_TRY(hipMemcpyAsync(pao, d_pao, k * k * sizeof(double), hipMemcpyDeviceToHost, s));
  if (transmd && depd) HIP_TRY(hipMemcpyAsync(transmd, d_transmd, k * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(parm_infl, d_infl, sizeof(double), hipMemcpyDeviceToHost, s));
  int hst = 0;
  HIP_TRY(hipMemcpyAsync(&hst, d_i + 1, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *st_out = hst;
  return LETKF_OK;
}
}  // namespace

void letkf_core_c(int ne, int nobs, int nobsl, const double* hdxb, const double* rdiag, const double* rloc,
                  const double* dep, double* parm_infl, double* trans, double* transm, double* pao,
                  const int* rdiag_wloc, const int* infl_update, const double* depd, double* transmd, int* status) {
  int st = 0;
  int rc = core_host(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm, pao, rdiag_wloc,
                     infl_update, depd, transmd, &st);
  if (rc != LETKF_OK) {
    std::fprintf(stderr, "!!! ERROR (letkf_core_c): %s\n", g_last_error.c_str());
    st = rc;
  }
  if (status) *status = st;
}

}  // extern "C"
