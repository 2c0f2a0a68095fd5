"""Python harness binding for libletkf_amd.so (the C ABI in include/letkf_amd.h).

The product is the HIP library + the Fortran shim (scale-letkf_amd/fortran); this module is only the
ctypes plumbing that tests/, bench.py and __graft_entry__.py use to drive the C ABI with torch-owned
device memory.  It never computes anything itself and has no CPU fallback: if the library is missing
or no gfx950 device is visible, every entry raises.

Because the directory name carries a hyphen, load it with `load_package()` from __graft_entry__.py
(importlib under the module name `scale_letkf_amd`).
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LETKF_AMD_LIB") or os.path.join(HERE, "lib", "libletkf_amd.so")   # override: profiling twin (make PROF=1)

LETKF_OK = 0
ST_OK, ST_NOT_CONVERGED, ST_NONPOSITIVE, ST_ILLCOND = 0, 1, 2, 3


class LetkfError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))]
    srcs.append(os.path.join(HERE, "..", "include", "letkf_amd.h"))
    if os.environ.get("LETKF_AMD_LIB") and os.path.exists(LIB_PATH) and not force:
        return LIB_PATH                      # an A/B or profiling twin: taken as it is, whatever its age
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:                                # (make's chatter to stderr: stdout belongs to the caller's JSON line)
        subprocess.check_call(["make", "-j4", "-C", HERE] + (["-B"] if force else []), stdout=2)
    return LIB_PATH


class CoreBatchArgs(C.Structure):
    _fields_ = [("ne", C.c_int32), ("nobs", C.c_int32), ("nbatch", C.c_int64), ("nobsl", C.c_void_p),
                ("hdxb", C.c_void_p), ("rdiag", C.c_void_p), ("rloc", C.c_void_p), ("dep", C.c_void_p),
                ("depd", C.c_void_p), ("parm_infl", C.c_void_p), ("trans", C.c_void_p), ("transm", C.c_void_p),
                ("pao", C.c_void_p), ("transmd", C.c_void_p), ("rdiag_wloc", C.c_int32),
                ("infl_update", C.c_int32), ("status", C.c_void_p), ("nsweep", C.c_void_p)]


class DasArgs(C.Structure):
    _fields_ = [("k", C.c_int32), ("nv", C.c_int32), ("det_run", C.c_int32), ("infl_adaptive", C.c_int32),
                ("relax_to_inflated_prior", C.c_int32), ("iv_p", C.c_int32), ("iv_q_first", C.c_int32),
                ("iv_q_last", C.c_int32), ("warm_stride", C.c_int32),
                ("relax_alpha", C.c_double), ("relax_alpha_spread", C.c_double), ("q_update_top", C.c_double),
                ("q_sprd_max", C.c_double), ("npts", C.c_int64), ("obs_off", C.c_void_p), ("obs_idx", C.c_void_p),
                ("rdiag_l", C.c_void_p), ("rloc_l", C.c_void_p), ("ensval", C.c_void_p), ("kld", C.c_int64),
                ("dep", C.c_void_p), ("beta", C.c_void_p), ("infl", C.c_void_p), ("gues", C.c_void_p),
                ("anal", C.c_void_p), ("sp", C.c_int64), ("sm", C.c_int64), ("sv", C.c_int64),
                ("trans_out", C.c_void_p), ("transm_out", C.c_void_p), ("pa_out", C.c_void_p),
                ("status", C.c_void_p), ("nsweep", C.c_void_p), ("rtps_infl_out", C.c_void_p),
                ("warm_run", C.c_int32), ("var_mask", C.c_uint32), ("infl_sv", C.c_int64)]


class SearchTables(C.Structure):
    """letkf_search_tables (include/letkf_amd.h section 3)"""
    _fields_ = [("nctype", C.c_int32), ("ngroup", C.c_int32), ("criterion", C.c_int32), ("nlon", C.c_int32),
                ("nlat", C.c_int32), ("limit_hint", C.c_int32), ("dx", C.c_double), ("dy", C.c_double),
                ("i_org", C.c_double), ("j_org", C.c_double), ("rain_base", C.c_double),
                ("group_start", C.c_void_p), ("group_member", C.c_void_p), ("vmode", C.c_void_p),
                ("hori_loc", C.c_void_p), ("vert_loc", C.c_void_p), ("varloc", C.c_void_p), ("max_nobs", C.c_void_p),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p), ("ngrdsch_i", C.c_void_p), ("ngrdsch_j", C.c_void_p),
                ("ngrdext_i", C.c_void_p), ("ngrdext_j", C.c_void_p), ("ac_off", C.c_void_p), ("ac_ext", C.c_void_p),
                ("ob_ri", C.c_void_p), ("ob_rj", C.c_void_p), ("ob_lev", C.c_void_p), ("ob_dat", C.c_void_p),
                ("ob_err", C.c_void_p)]


class QcParams(C.Structure):
    """letkf_qc_params (include/letkf_amd.h section 5); defaults = the reference namelist defaults
    (scale/common/common_nml.f90)"""
    _fields_ = [("member", C.c_int32), ("det_run", C.c_int32), ("use_radar_ref", C.c_int32),
                ("use_radar_vr", C.c_int32), ("min_radar_ref_member", C.c_int32),
                ("min_radar_ref_member_obsref", C.c_int32), ("radar_ref_thres_dbz", C.c_double),
                ("gross_error", C.c_double), ("gross_error_rain", C.c_double), ("gross_error_radar_ref", C.c_double),
                ("gross_error_radar_vr", C.c_double), ("gross_error_radar_prh", C.c_double),
                ("gross_error_tcx", C.c_double), ("gross_error_tcy", C.c_double), ("gross_error_tcp", C.c_double),
                ("h08", C.c_int32), ("h08_min_cld_member", C.c_int32), ("h08_limit_lev", C.c_double),
                ("gross_error_h08", C.c_double), ("h08_bt_min", C.c_double), ("h08_lev", C.c_void_p),
                ("h08_val2", C.c_void_p)]


class Mesh(C.Structure):
    """letkf_mesh (section 5); ngrd_i / ngrd_j are HOST int32 arrays"""
    _fields_ = [("nctype", C.c_int32), ("nlon", C.c_int32), ("nlat", C.c_int32), ("ihalo", C.c_int32),
                ("jhalo", C.c_int32), ("rank_i", C.c_int32), ("rank_j", C.c_int32), ("fix_ij_obsgrd", C.c_int32),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p)]


class HaloLayout(C.Structure):
    """letkf_halo_layout (section 5); the four arrays are HOST int32 arrays"""
    _fields_ = [("nctype", C.c_int32), ("nprocs", C.c_int32), ("prc_num_x", C.c_int32), ("myrank", C.c_int32),
                ("ngrd_i", C.c_void_p), ("ngrd_j", C.c_void_p), ("ngrdsch_i", C.c_void_p), ("ngrdsch_j", C.c_void_p)]


class StateConsts(C.Structure):
    """letkf_state_consts (include/letkf_amd.h section 4)"""
    _fields_ = [("rdry", C.c_double), ("rvap", C.c_double), ("cvdry", C.c_double), ("pre00", C.c_double),
                ("tracer_cv", C.c_double * 8), ("iv_rho", C.c_int32), ("iv_rhou", C.c_int32), ("iv_rhov", C.c_int32),
                ("iv_rhow", C.c_int32), ("iv_rhot", C.c_int32), ("iv_u", C.c_int32), ("iv_v", C.c_int32),
                ("iv_w", C.c_int32), ("iv_t", C.c_int32), ("iv_p", C.c_int32), ("iv_q", C.c_int32),
                ("positive_definite_q", C.c_int32), ("positive_definite_qhyd", C.c_int32), ("reserved0", C.c_int32)]


def scale_rm_consts(clamp=True):
    """SCALE-RM's constants (scale_const / scale_tracer; NOT in the reference tree, values of SCALE-RM 5.x) with the
    variable slots of scale/common/common_scale.f90:36-51 (rho/u, rhou/v? no: u=1 v=2 w=3 t=4 p=5 q=6..11, 0-based
    0..10; prognostic twins share the slots: rho<->... see below)."""
    c = StateConsts()
    c.rdry, c.rvap, c.pre00 = 287.04, 461.46, 1.0e5
    c.cvdry = 1004.64 - 287.04
    for i, v in enumerate([1407.0, 4218.0, 4218.0, 2006.0, 2006.0, 2006.0]):   # QV, QC, QR, QI, QS, QG
        c.tracer_cv[i] = v
    # common_scale.f90:36-51: iv3d_rho=1 iv3d_rhou=2 iv3d_rhov=3 iv3d_rhow=4 iv3d_rhot=5 share the slots of
    # iv3d_u=1 iv3d_v=2 iv3d_w=3 iv3d_t=4 iv3d_p=5 (the transform overwrites in place)
    c.iv_rho, c.iv_rhou, c.iv_rhov, c.iv_rhow, c.iv_rhot = 0, 1, 2, 3, 4
    c.iv_u, c.iv_v, c.iv_w, c.iv_t, c.iv_p, c.iv_q = 0, 1, 2, 3, 4, 5
    c.positive_definite_q = c.positive_definite_qhyd = int(clamp)
    return c


class BetaParams(C.Structure):
    """letkf_beta_params (include/letkf_amd.h section 7)"""
    _fields_ = [("radar_only", C.c_int32), ("ihalo", C.c_int32), ("jhalo", C.c_int32), ("nlong", C.c_int32),
                ("nlatg", C.c_int32), ("reserved0", C.c_int32), ("radar_zmax", C.c_double),
                ("vert_local_radar", C.c_double), ("boundary_buffer_width", C.c_double), ("dx", C.c_double),
                ("dy", C.c_double)]


EXPORTS = ["letkf_amd_abi_version", "letkf_amd_last_error", "letkf_ctx_create", "letkf_ctx_destroy",
           "letkf_ctx_set_stream", "letkf_ctx_set_option", "letkf_ctx_synchronize", "letkf_core_c", "letkf_core_batch_dev",
           "letkf_das_points_dev", "letkf_das_points_fused_dev", "letkf_das_columns_dev", "letkf_obs_search_dev", "letkf_obs_search_columns_dev", "letkf_ens_to_perturbations_dev", "letkf_ens_mean_dev",
           "letkf_state_trans_dev", "letkf_member_points_dev", "letkf_ens_spread_dev",
           "letkf_obs_departure_dev", "letkf_obs_mesh_sort_dev", "letkf_obs_halo_plan_dev",
           "letkf_obs_gather_rows_dev", "letkf_obs_gather_i32_dev", "letkf_monit_dep_dev",
           "letkf_additive_inflation_dev", "letkf_addinfl_weight_dev",
           "letkf_var_local_classes", "letkf_ctype_merge_groups", "letkf_radar_only", "letkf_relax_beta_dev",
           "letkf_infl_init_dev", "letkf_obs_allgatherv_dev", "letkf_alltoallv_dev", "letkf_allreduce_sum_i32_dev",
           "letkf_members_alltoall_dev",
           "letkf_ctx_timing_enable", "letkf_ctx_timing_read", "letkf_ctx_last_path", "letkf_sched_plan_check", "letkf_sched_plan_check_units"]

_lib = None


def lib():
    """dlopen the library.  `import torch` first when torch is in the process so both share one HIP runtime
    (torch's libamdhip64.so and /opt/rocm's carry the same SONAME; the first one loaded wins)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LetkfError(f"{LIB_PATH} not built: run `make -C scale-letkf_amd` (no CPU fallback exists)")
        try:            # make torch's HIP runtime the process's one BEFORE ours binds /opt/rocm's copy: loaded the other
            import torch  # noqa: F401  way round, the two runtimes coexist and ours sees no device
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.letkf_amd_last_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(_lib, name)  # raises AttributeError when a declared symbol is missing
    return _lib


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Context:
    """One device + stream + workspace (letkf_ctx)."""

    def __init__(self, device=-1, stream=None):
        self._l = lib()
        self._c = C.c_void_p()
        rc = self._l.letkf_ctx_create(C.c_int(device), C.byref(self._c))
        self._check(rc)
        if stream is not None:
            self.set_stream(stream)

    def _check(self, rc):
        if rc != LETKF_OK:
            raise LetkfError(f"letkf_amd error {rc}: {self._l.letkf_amd_last_error().decode()}")

    def close(self):
        if self._c:
            self._l.letkf_ctx_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._check(self._l.letkf_ctx_set_stream(self._c, C.c_void_p(stream_handle)))

    OPT_STAGED_POLY = 1     # include/letkf_amd.h LETKF_OPT_STAGED_POLY
    OPT_COLUMN_SURVIVORS = 2   # LETKF_OPT_COLUMN_SURVIVORS
    OPT_LIMITED_RINGS = 3      # LETKF_OPT_LIMITED_RINGS
    OPT_RING_BATCH_MB = 4      # LETKF_OPT_RING_BATCH_MB
    OPT_RING_RELEASE = 5       # LETKF_OPT_RING_RELEASE
    OPT_SMALL_K_TRIO = 6       # LETKF_OPT_SMALL_K_TRIO

    def set_option(self, option, value):
        self._check(self._l.letkf_ctx_set_option(self._c, C.c_int(option), C.c_int(value)))

    def synchronize(self):
        self._check(self._l.letkf_ctx_synchronize(self._c))

    def timing_enable(self, on=True):
        self._check(self._l.letkf_ctx_timing_enable(self._c, C.c_int(1 if on else 0)))

    def last_path(self):
        buf = C.create_string_buffer(256)
        self._check(self._l.letkf_ctx_last_path(self._c, buf, C.c_int32(256)))
        return buf.value.decode()

    def timing_read(self, reset=True):
        avg = C.c_double(0.0)
        n = C.c_int64(0)
        self._check(self._l.letkf_ctx_timing_read(self._c, C.byref(avg), C.byref(n), C.c_int(1 if reset else 0)))
        return avg.value, n.value

    # ---- (1b) batched letkf_core on device tensors
    def core_batch(self, ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=None, pao=None,
                   depd=None, transmd=None, rdiag_wloc=False, infl_update=False, status=None, nsweep=None):
        a = CoreBatchArgs()
        a.ne, a.nobs, a.nbatch = ne, nobs, nobsl.numel()
        a.nobsl, a.hdxb, a.rdiag, a.rloc, a.dep = _ptr(nobsl), _ptr(hdxb), _ptr(rdiag), _ptr(rloc), _ptr(dep)
        a.depd, a.parm_infl, a.trans, a.transm = _ptr(depd), _ptr(parm_infl), _ptr(trans), _ptr(transm)
        a.pao, a.transmd = _ptr(pao), _ptr(transmd)
        a.rdiag_wloc, a.infl_update = int(bool(rdiag_wloc)), int(bool(infl_update))
        a.status, a.nsweep = _ptr(status), _ptr(nsweep)
        self._check(self._l.letkf_core_batch_dev(self._c, C.byref(a)))

    # ---- (2) das_letkf point update on device tensors
    def das_points(self, k, nv, obs_off, obs_idx, rdiag_l, rloc_l, ensval, kld, dep, infl, gues, anal, sp, sm, sv,
                   beta=None, det_run=False, infl_adaptive=False, relax_to_inflated_prior=False, relax_alpha=0.0,
                   relax_alpha_spread=0.0, q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10,
                   trans_out=None, transm_out=None, pa_out=None, status=None, nsweep=None, rtps_infl_out=None,
                   warm_run=0, var_mask=0, fused=None, nobs_out=None, warm_stride=0, infl_sv=0):
        """fused = (tables, ri, rj, rlev, rz): obs_local fused into the kernel (obs_off .. rloc_l may be None)"""
        a = DasArgs()
        a.k, a.nv, a.det_run, a.infl_adaptive = k, nv, int(bool(det_run)), int(bool(infl_adaptive))
        a.relax_to_inflated_prior = int(bool(relax_to_inflated_prior))
        a.iv_p, a.iv_q_first, a.iv_q_last = iv_p, iv_q_first, iv_q_last
        a.relax_alpha, a.relax_alpha_spread = relax_alpha, relax_alpha_spread
        a.q_update_top, a.q_sprd_max = q_update_top, q_sprd_max
        a.npts = (obs_off.numel() - 1) if fused is None else fused[1].numel()
        a.obs_off, a.obs_idx, a.rdiag_l, a.rloc_l = _ptr(obs_off), _ptr(obs_idx), _ptr(rdiag_l), _ptr(rloc_l)
        a.ensval, a.kld, a.dep, a.beta, a.infl = _ptr(ensval), kld, _ptr(dep), _ptr(beta), _ptr(infl)
        a.gues, a.anal, a.sp, a.sm, a.sv = _ptr(gues), _ptr(anal), sp, sm, sv
        a.trans_out, a.transm_out, a.pa_out = _ptr(trans_out), _ptr(transm_out), _ptr(pa_out)
        a.status, a.nsweep = _ptr(status), _ptr(nsweep)
        a.rtps_infl_out = _ptr(rtps_infl_out)
        a.warm_run = int(warm_run)
        a.warm_stride = int(warm_stride)
        a.var_mask = int(var_mask)
        a.infl_sv = int(infl_sv)
        if fused is None:
            self._check(self._l.letkf_das_points_dev(self._c, C.byref(a)))
        else:
            t, ri, rj, rlev, rz = fused
            self._check(self._l.letkf_das_points_fused_dev(self._c, C.byref(a), C.byref(t), _ptr(ri), _ptr(rj),
                                                           _ptr(rlev), _ptr(rz), _ptr(nobs_out)))

    def das_columns(self, k, nv, tables, nij1, nlev, rig, rjg, rlev, rz, ensval, kld, dep, infl, gues, anal, sp, sm, sv,
                    list_bytes=0, nobs_out=None, beta=None, det_run=False, infl_adaptive=False, relax_to_inflated_prior=False,
                    relax_alpha=0.0, relax_alpha_spread=0.0, q_update_top=0.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5,
                    iv_q_last=10, status=None, nsweep=None, rtps_infl_out=None, warm_run=0, var_mask=0, infl_sv=0):
        """letkf_das_columns_dev: obs_local + loop body for the points p = ij + nij1*lev -- list-free where the one-wave kernel
        serves the call (horizontal survivors per column, batches of columns that fit list_bytes), else by slabs of levels whose
        lists fit list_bytes of library workspace."""
        a = DasArgs()
        a.k, a.nv, a.det_run, a.infl_adaptive = k, nv, int(bool(det_run)), int(bool(infl_adaptive))
        a.relax_to_inflated_prior = int(bool(relax_to_inflated_prior))
        a.iv_p, a.iv_q_first, a.iv_q_last = iv_p, iv_q_first, iv_q_last
        a.relax_alpha, a.relax_alpha_spread = relax_alpha, relax_alpha_spread
        a.q_update_top, a.q_sprd_max = q_update_top, q_sprd_max
        a.npts = nij1 * nlev
        a.ensval, a.kld, a.dep, a.beta, a.infl = _ptr(ensval), kld, _ptr(dep), _ptr(beta), _ptr(infl)
        a.gues, a.anal, a.sp, a.sm, a.sv = _ptr(gues), _ptr(anal), sp, sm, sv
        a.status, a.nsweep, a.rtps_infl_out = _ptr(status), _ptr(nsweep), _ptr(rtps_infl_out)
        a.warm_run, a.var_mask, a.infl_sv = int(warm_run), int(var_mask), int(infl_sv)
        self._check(self._l.letkf_das_columns_dev(self._c, C.byref(a), C.byref(tables), C.c_int64(nij1), C.c_int32(nlev),
                                                  _ptr(rig), _ptr(rjg), _ptr(rlev), _ptr(rz), C.c_int64(list_bytes),
                                                  _ptr(nobs_out)))

    # ---- (3) obs_local on the device: two-phase CSR build (count, scan, fill)
    def obs_search(self, tables, ri, rj, rlev, rz):
        """Returns (obs_off, obs_idx, rdiag_l, rloc_l) device tensors for the points (ri, rj, rlev, rz)."""
        import torch
        npts = ri.numel()
        counts = torch.zeros(npts, dtype=torch.int32, device=ri.device)
        self._check(self._l.letkf_obs_search_dev(self._c, C.byref(tables), C.c_int64(npts), _ptr(ri), _ptr(rj),
                                                 _ptr(rlev), _ptr(rz), C.c_int32(0), _ptr(counts), None, None, None,
                                                 None))
        obs_off = torch.zeros(npts + 1, dtype=torch.int64, device=ri.device)
        obs_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
        nnz = int(obs_off[-1].item())
        obs_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=ri.device)
        rdiag = torch.empty(max(nnz, 1), dtype=torch.float64, device=ri.device)
        rloc = torch.empty(max(nnz, 1), dtype=torch.float64, device=ri.device)
        self._check(self._l.letkf_obs_search_dev(self._c, C.byref(tables), C.c_int64(npts), _ptr(ri), _ptr(rj),
                                                 _ptr(rlev), _ptr(rz), C.c_int32(1), None, _ptr(obs_off),
                                                 _ptr(obs_idx), _ptr(rdiag), _ptr(rloc)))
        return obs_off, obs_idx[:nnz], rdiag[:nnz], rloc[:nnz]

    def obs_search_columns(self, tables, nij1, nlev, rig, rjg, rlev, rz, nobs_ctype=None, cutd_ctype=None):
        """Column-cooperative obs_local for points p = ij + nij1*lev; same return as obs_search."""
        import torch
        npts = nij1 * nlev
        counts = torch.zeros(npts, dtype=torch.int32, device=rig.device)
        f = self._l.letkf_obs_search_columns_dev
        self._check(f(self._c, C.byref(tables), C.c_int64(nij1), C.c_int32(nlev), _ptr(rig), _ptr(rjg), _ptr(rlev),
                      _ptr(rz), C.c_int32(0), _ptr(counts), None, None, None, None, None, None))
        obs_off = torch.zeros(npts + 1, dtype=torch.int64, device=rig.device)
        obs_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
        nnz = int(obs_off[-1].item())
        obs_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=rig.device)
        rdiag = torch.empty(max(nnz, 1), dtype=torch.float64, device=rig.device)
        rloc = torch.empty(max(nnz, 1), dtype=torch.float64, device=rig.device)
        self._check(f(self._c, C.byref(tables), C.c_int64(nij1), C.c_int32(nlev), _ptr(rig), _ptr(rjg), _ptr(rlev),
                      _ptr(rz), C.c_int32(1), None, _ptr(obs_off), _ptr(obs_idx), _ptr(rdiag), _ptr(rloc), _ptr(nobs_ctype),
                      _ptr(cutd_ctype)))    # (diagnostics with the fill pass: the count pass then needs no selection)
        return obs_off, obs_idx[:nnz], rdiag[:nnz], rloc[:nnz]

    # ---- (5) set_letkf_obs on the device
    def obs_departure(self, params, elm, dat, err, ensval, kld, val, qc):
        self._check(self._l.letkf_obs_departure_dev(self._c, C.byref(params), C.c_int64(elm.numel()), _ptr(elm),
                                                    _ptr(dat), _ptr(err), _ptr(ensval), C.c_int64(kld), _ptr(val),
                                                    _ptr(qc)))

    def obs_mesh_sort(self, mesh, ncell, ctype, ri, rj, qc):
        """Returns (n_cell [ncell] int32, key [nsorted] int32) device tensors."""
        import torch
        nobs = ctype.numel()
        n_cell = torch.zeros(max(ncell, 1), dtype=torch.int32, device=ctype.device)
        key = torch.empty(max(nobs, 1), dtype=torch.int32, device=ctype.device)
        ns = C.c_int64(0)
        self._check(self._l.letkf_obs_mesh_sort_dev(self._c, C.byref(mesh), C.c_int64(nobs), _ptr(ctype), _ptr(ri),
                                                    _ptr(rj), _ptr(qc), _ptr(n_cell), _ptr(key), C.byref(ns)))
        return n_cell[:ncell], key[:ns.value]

    def obs_halo_plan(self, layout, n_all, nacx, cap):
        """Returns (ac_ext [nacx] int32, src_row [nobstotal] int32) device tensors."""
        import torch
        ac_ext = torch.zeros(max(nacx, 1), dtype=torch.int32, device=n_all.device)
        src_row = torch.empty(max(cap, 1), dtype=torch.int32, device=n_all.device)
        nt = C.c_int64(0)
        self._check(self._l.letkf_obs_halo_plan_dev(self._c, C.byref(layout), _ptr(n_all), _ptr(ac_ext),
                                                    _ptr(src_row), C.c_int64(cap), C.byref(nt)))
        return ac_ext[:nacx], src_row[:nt.value]

    def obs_gather_rows(self, src_row, ncols, src, ld_src, dst, ld_dst):
        self._check(self._l.letkf_obs_gather_rows_dev(self._c, C.c_int64(src_row.numel()), _ptr(src_row),
                                                      C.c_int32(ncols), _ptr(src), C.c_int64(ld_src), _ptr(dst),
                                                      C.c_int64(ld_dst)))

    def obs_gather_i32(self, src_row, src, dst):
        self._check(self._l.letkf_obs_gather_i32_dev(self._c, C.c_int64(src_row.numel()), _ptr(src_row), _ptr(src),
                                                     _ptr(dst)))

    # ---- (6) after the loop
    def monit_dep(self, elem_uid, elm, dep, qc):
        """Returns (nobs int32 [nid], bias, rmse) device tensors."""
        import numpy as np
        import torch
        ids = np.ascontiguousarray(elem_uid, dtype=np.int32)
        nid = len(ids)
        nobs = torch.zeros(nid, dtype=torch.int32, device=dep.device)
        bias = torch.zeros(nid, dtype=torch.float64, device=dep.device)
        rmse = torch.zeros(nid, dtype=torch.float64, device=dep.device)
        self._check(self._l.letkf_monit_dep_dev(self._c, C.c_int32(nid), ids.ctypes.data_as(C.c_void_p),
                                                C.c_int64(dep.numel()), _ptr(elm), _ptr(dep), _ptr(qc), _ptr(nobs),
                                                _ptr(bias), _ptr(rmse)))
        return nobs, bias, rmse

    def additive_inflation(self, k, nv, npts, nij1, anal, add, sp, sm, sv, infl_add, weight=None, qmean=None, q_sp=0,
                           q_sv=0, iv_q_first=5, iv_q_last=10, ishuf=None):
        self._check(self._l.letkf_additive_inflation_dev(
            self._c, C.c_int32(k), C.c_int32(nv), C.c_int64(npts), C.c_int64(nij1), _ptr(anal), _ptr(add),
            C.c_int64(sp), C.c_int64(sm), C.c_int64(sv), C.c_double(infl_add), _ptr(weight), _ptr(qmean),
            C.c_int64(q_sp), C.c_int64(q_sv), C.c_int32(iv_q_first), C.c_int32(iv_q_last), _ptr(ishuf)))

    def addinfl_weight(self, rig, rjg, ob_ri, ob_rj, dx, dy, hori_loc):
        import torch
        w = torch.zeros(rig.numel(), dtype=torch.float64, device=rig.device)
        self._check(self._l.letkf_addinfl_weight_dev(self._c, C.c_int64(rig.numel()), _ptr(rig), _ptr(rjg),
                                                     C.c_int64(ob_ri.numel()), _ptr(ob_ri), _ptr(ob_rj),
                                                     C.c_double(dx), C.c_double(dy), C.c_double(hori_loc), _ptr(w)))
        return w

    # ---- (8) the exchange, on an RCCL communicator the caller owns (an integer / c_void_p ncclComm_t)
    def obs_allgatherv(self, nccl_comm, myrank, counts, send, recv):
        """counts: python ints per rank (rows); send / recv: device tensors whose rows are contiguous."""
        n = len(counts)
        cnt = (C.c_int64 * n)(*[int(x) for x in counts])
        row_bytes = send.element_size() * (send[0].numel() if send.dim() > 1 and send.shape[0] > 0 else
                                           (recv[0].numel() if recv.dim() > 1 else 1))
        self._check(self._l.letkf_obs_allgatherv_dev(self._c, C.c_void_p(nccl_comm), C.c_int32(n), C.c_int32(myrank),
                                                     cnt, C.c_int64(row_bytes), _ptr(send), _ptr(recv)))

    # ---- (7) das_letkf set-up
    def alltoallv(self, nccl_comm, myrank, send_counts, send_offs, recv_counts, recv_offs, row_bytes, send, recv):
        """letkf_alltoallv_dev: counts / offsets are host lists in rows of row_bytes bytes"""
        n = len(send_counts)
        arr = lambda v: (C.c_int64 * n)(*[int(a) for a in v])
        self._check(self._l.letkf_alltoallv_dev(self._c, C.c_void_p(nccl_comm), C.c_int32(n), C.c_int32(myrank), arr(send_counts),
                                                arr(send_offs), arr(recv_counts), arr(recv_offs), C.c_int64(row_bytes), _ptr(send),
                                                _ptr(recv)))

    def allreduce_sum_i32(self, nccl_comm, nranks, buf):
        self._check(self._l.letkf_allreduce_sum_i32_dev(self._c, C.c_void_p(nccl_comm), C.c_int32(nranks), C.c_int64(buf.numel()),
                                                        _ptr(buf)))

    def members_alltoall(self, nccl_comm, nranks, myrank, direction, nlev, nlon, nlat, nv3d, mstart, mcount, v3dg, x, sp, sm, sv):
        self._check(self._l.letkf_members_alltoall_dev(self._c, C.c_void_p(nccl_comm), C.c_int32(nranks), C.c_int32(myrank),
                                                       C.c_int32(direction), C.c_int32(nlev), C.c_int32(nlon), C.c_int32(nlat),
                                                       C.c_int32(nv3d), C.c_int32(mstart), C.c_int32(mcount), _ptr(v3dg), _ptr(x),
                                                       C.c_int64(sp), C.c_int64(sm), C.c_int64(sv)))

    def relax_beta(self, params, nij1, nlev, rig, rjg, hgt, beta):
        self._check(self._l.letkf_relax_beta_dev(self._c, C.byref(params), C.c_int64(nij1), C.c_int32(nlev), _ptr(rig),
                                                 _ptr(rjg), _ptr(hgt), _ptr(beta)))

    def infl_init(self, work3d, infl_mul, infl_mul_min):
        self._check(self._l.letkf_infl_init_dev(self._c, C.c_int64(work3d.numel()), _ptr(work3d),
                                                C.c_double(infl_mul), C.c_double(infl_mul_min)))

    # ---- (4) the steps either side of the loop
    def state_trans(self, consts, nlev, nlon, nlat, nv3d, v3dg, inverse=False):
        self._check(self._l.letkf_state_trans_dev(self._c, C.byref(consts), C.c_int32(nlev), C.c_int32(nlon),
                                                  C.c_int32(nlat), C.c_int32(nv3d), _ptr(v3dg),
                                                  C.c_int32(1 if inverse else 0)))

    def member_points(self, direction, nlev, nlon, nlat, nv3d, np_, rank, m, v3dg, x, nij1, sp, sm, sv):
        self._check(self._l.letkf_member_points_dev(self._c, C.c_int32(direction), C.c_int32(nlev), C.c_int32(nlon),
                                                    C.c_int32(nlat), C.c_int32(nv3d), C.c_int32(np_), C.c_int32(rank),
                                                    C.c_int32(m), _ptr(v3dg), _ptr(x), C.c_int64(nij1), C.c_int64(sp),
                                                    C.c_int64(sm), C.c_int64(sv)))

    def ens_spread(self, k, nv, npts, x, sp, sm, sv, sprd):
        self._check(self._l.letkf_ens_spread_dev(self._c, C.c_int32(k), C.c_int32(nv), C.c_int64(npts), _ptr(x),
                                                 C.c_int64(sp), C.c_int64(sm), C.c_int64(sv), _ptr(sprd)))

    def to_perturbations(self, k, nv, npts, x, sp, sm, sv):
        self._check(self._l.letkf_ens_to_perturbations_dev(self._c, C.c_int32(k), C.c_int32(nv), C.c_int64(npts),
                                                           _ptr(x), C.c_int64(sp), C.c_int64(sm), C.c_int64(sv)))

    def ens_mean(self, k, nv, npts, x, sp, sm, sv):
        self._check(self._l.letkf_ens_mean_dev(self._c, C.c_int32(k), C.c_int32(nv), C.c_int64(npts), _ptr(x),
                                               C.c_int64(sp), C.c_int64(sm), C.c_int64(sv)))


def letkf_core_host(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, want_transm=True, want_pao=True,
                    rdiag_wloc=None, infl_update=None, depd=None, want_transmd=False):
    """The host-pointer drop-in letkf_core_c (what the Fortran shim calls), on numpy arrays."""
    import numpy as np
    l = lib()
    dp = C.POINTER(C.c_double)
    f = lambda a: None if a is None else a.ctypes.data_as(dp)
    hdxb = np.asfortranarray(hdxb, dtype=np.float64)
    trans = np.zeros((ne, ne), order="F")
    transm = np.zeros(ne) if want_transm else None
    pao = np.zeros((ne, ne), order="F") if want_pao else None
    transmd = np.zeros(ne) if want_transmd else None
    infl = C.c_double(parm_infl)
    wl = C.c_int(1 if rdiag_wloc else 0)
    iu = C.c_int(1 if infl_update else 0)
    st = C.c_int(-99)
    l.letkf_core_c(C.c_int(ne), C.c_int(nobs), C.c_int(nobsl), f(hdxb), f(rdiag), f(rloc), f(dep), C.byref(infl),
                   f(trans), f(transm), f(pao), C.byref(wl) if rdiag_wloc is not None else None,
                   C.byref(iu) if infl_update is not None else None, f(depd), f(transmd), C.byref(st))
    return dict(trans=trans, transm=transm, pao=pao, transmd=transmd if depd is not None else None,
                parm_infl=infl.value, status=st.value)


# ---- (7) host-side table derivations of das_letkf's set-up (no device needed)
def var_local_classes(var_local):
    """var_local: numpy (nvar, nlt).  Returns (n2nc, n2n, nclass), 0-based (letkf_tools.f90:130-157)."""
    import numpy as np
    v = np.asfortranarray(var_local, dtype=np.float64)
    nvar, nlt = v.shape
    n2nc = np.zeros(nvar, dtype=np.int32)
    n2n = np.zeros(nvar, dtype=np.int32)
    nc = C.c_int32(0)
    rc = lib().letkf_var_local_classes(C.c_int32(nvar), C.c_int32(nlt), v.ctypes.data_as(C.c_void_p),
                                       n2nc.ctypes.data_as(C.c_void_p), n2n.ctypes.data_as(C.c_void_p), C.byref(nc))
    if rc != LETKF_OK:
        raise LetkfError(f"letkf_var_local_classes: {rc}")
    return n2nc, n2n, nc.value


def ctype_merge_groups(elm_u_ctype, typ_ctype, ctype_merge):
    """ctype_merge: numpy (nid_obs, nobtype) int32.  Returns (group_start [ngroup+1], group_member [nctype])."""
    import numpy as np
    eu = np.ascontiguousarray(elm_u_ctype, dtype=np.int32)
    ty = np.ascontiguousarray(typ_ctype, dtype=np.int32)
    cm = np.asfortranarray(ctype_merge, dtype=np.int32)
    nct = len(eu)
    gs = np.zeros(nct + 1, dtype=np.int32)
    gm = np.zeros(max(nct, 1), dtype=np.int32)
    ng = C.c_int32(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().letkf_ctype_merge_groups(C.c_int32(nct), p(eu), p(ty), C.c_int32(cm.shape[0]), C.c_int32(cm.shape[1]),
                                        p(cm), p(gs), p(gm), C.byref(ng))
    if rc != LETKF_OK:
        raise LetkfError(f"letkf_ctype_merge_groups: {rc}")
    return gs[:ng.value + 1].copy(), gm[:nct].copy()


def radar_only(typ_ctype, typ_radar=22):
    import numpy as np
    ty = np.ascontiguousarray(typ_ctype, dtype=np.int32)
    return int(lib().letkf_radar_only(C.c_int32(len(ty)), ty.ctypes.data_as(C.c_void_p), C.c_int32(typ_radar)))
