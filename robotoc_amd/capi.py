"""ctypes binding of the C ABI (include/rtoc.h) -- the same entry points a C++ host
or the reference's solver classes would bind.  There is NO CPU fallback: if
librtoc_hip.so is missing or no HIP device is visible, this module raises.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

from .types import (BoxRow, BUF_CDD, BUF_CON, BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, BUF_STEP, Dims, Grid,
                    Layout, OPT_BACKWARD_WAVES, OPT_CONDENSE_SPLIT, OPT_CONDENSE_KEEP_QAF, OPT_FXX_STRUCTURE, OPT_GRAPH, OPT_SWITCHING_TRANSPORT, OPT_IMPACT_CONES, OPT_UNCONSTR_DENSE, OPT_LINEARIZE_FUSED, OPT_LINEARIZE_DOFS_PER_PASS, OPT_CONE_JACOBIAN, OPT_MAX_DTS0, OPT_SWEEP_CHUNKS, OPT_WRITEBACK_KKT, OPT_BACKWARD_SCAN, OPT_BACKWARD_REGISTER, OPT_CONDENSE_REGISTER,
                    grid_array)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = [
    "rtoc_version", "rtoc_device_count", "rtoc_dims_supported", "rtoc_create", "rtoc_destroy",
    "rtoc_get_layout", "rtoc_set_grid", "rtoc_set_stream", "rtoc_set_option", "rtoc_upload",
    "rtoc_download", "rtoc_device_ptr", "rtoc_buffer_count", "rtoc_bind", "rtoc_condense",
    "rtoc_riccati_backward", "rtoc_riccati_forward", "rtoc_unconstr_backward",
    "rtoc_unconstr_forward", "rtoc_expand", "rtoc_update", "rtoc_status", "rtoc_clear_status",
    "rtoc_sync", "rtoc_time_phase", "rtoc_set_constraint_rows", "rtoc_gather_directions", "rtoc_error_string",
    "rtoc_riccati_sweep", "rtoc_correct_state_equation", "rtoc_correct_costate_direction",
    "rtoc_compute_initial_state_direction", "rtoc_unconstr_condense", "rtoc_unconstr_expand",
    "rtoc_newton_iteration", "rtoc_converged_count", "rtoc_clone", "rtoc_check_fxx_structure", "rtoc_sto_eval_kkt", "rtoc_set_friction_cones", "rtoc_set_wrench_cones", "rtoc_wrench_cone_matrix", "rtoc_save_stage_dump", "rtoc_load_stage_dump", "rtoc_kkt_error", "rtoc_integrate_solution",
    "rtoc_set_robot_model", "rtoc_robot_model_plan", "rtoc_set_contact_schedule", "rtoc_linearize_contact_dynamics",
    "rtoc_line_search_filter", "rtoc_line_search_clear", "rtoc_set_configuration_cost", "rtoc_set_initial_state",
    "rtoc_unconstr_eval_kkt", "rtoc_unconstr_update_solution", "rtoc_set_constraint_bounds", "rtoc_unconstr_init_constraints", "rtoc_linearize_state_equation", "rtoc_contact_eval_kkt", "rtoc_contact_update_solution",
    "rtoc_set_barrier_param", "rtoc_set_friction_coefficients", "rtoc_contact_init_constraints", "rtoc_set_wrench_cone_params",
    "rtoc_graph_replay_count",
    "rtoc_sto_set_problem", "rtoc_sto_set_regularization", "rtoc_sto_set_cost_terms", "rtoc_sto_init_constraints",
    "rtoc_sto_correct_time_steps", "rtoc_sto_eval_kkt_device", "rtoc_sto_compute_step_sizes", "rtoc_sto_integrate_solution",
    "rtoc_sto_get_event_times", "rtoc_sto_get_time_steps", "rtoc_sto_get_constraint_data", "rtoc_sto_get_kkt_terms",
    "rtoc_sto_set_slack_dual", "rtoc_contact_eval_ocp", "rtoc_set_line_search", "rtoc_contact_line_search",
    "rtoc_bandwidth_probe", "rtoc_get_option",
]


LINE_SEARCH_FILTER_CAPACITY = 32  # RTOC_LINE_SEARCH_FILTER_CAPACITY


class RtocError(RuntimeError):
    pass


def lib_path():
    # RTOC_HIP_LIB: an alternative build of the same library, e.g. the one with the phase stamps compiled
    # in (make -C robotoc_amd/csrc PROF=1 OUT=../librtoc_hip_prof.so) for tools/phase_profile*.py
    return os.environ.get("RTOC_HIP_LIB") or os.path.join(_HERE, "librtoc_hip.so")


def build(force=False):
    """Compile the HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU): one translation unit per
    robot shape (csrc/Makefile: SHAPES), built in parallel."""
    src_dir = os.path.join(_HERE, "csrc")
    so = lib_path()
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if os.path.isfile(os.path.join(src_dir, f))]
    deps += [os.path.join(_HERE, "..", "include", f) for f in ("rtoc.h", "rtoc_layout.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
    if stale:
        subprocess.check_call(["make", "-C", src_dir, "-j%d" % min(8, os.cpu_count() or 1)] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return so


def build_plugin(nv, nu, ns, nw0=None, nw1=None, force=False):
    """Kernel set of one more robot shape as robotoc_amd/librtoc_shape_<nv>_<nu>_<ns>.so (csrc/Makefile: plugin), picked up by
    rtoc_create / rtoc_dims_supported at run time.  Wave counts of the tile-split backward variants default like the library's
    own JIT (RTOC_SHAPE_JIT=1): 1 / 3 up to a 36-wide state, 4 / 4 (5 beyond 64) above."""
    nx = 2 * nv
    nw0 = nw0 if nw0 is not None else (1 if nx <= 36 else 4)
    nw1 = nw1 if nw1 is not None else (3 if nx <= 36 else (5 if nx > 64 else 4))
    so = os.path.join(_HERE, "librtoc_shape_%d_%d_%d.so" % (nv, nu, ns))
    src_dir = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if os.path.isfile(os.path.join(src_dir, f))]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["make", "-C", src_dir, "plugin", "SHAPE=%d:%d:%d:%d:%d" % (nv, nu, ns, nw0, nw1)], stdout=subprocess.DEVNULL)
    return so


def compiled_shapes():
    """(nv, nu, ns) of every robot shape in csrc/Makefile's SHAPES list."""
    import re
    txt = open(os.path.join(_HERE, "csrc", "Makefile")).read()
    m = re.search(r"^SHAPES \?= (.*)$", txt, re.M)
    return [tuple(int(v) for v in tok.split(":")[:3]) for tok in m.group(1).split()]


def lib():
    global _LIB
    if _LIB is None:
        so = lib_path()
        if not os.path.exists(so):
            raise RtocError("librtoc_hip.so not built (run __graft_entry__.build()); "
                            "the HIP path has no CPU fallback")
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.rtoc_create.argtypes = [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.rtoc_destroy.argtypes = [vp]
        L.rtoc_dims_supported.argtypes = [C.POINTER(Dims)]
        L.rtoc_get_layout.argtypes = [vp, C.POINTER(Layout)]
        L.rtoc_layout_for_dims.argtypes = [C.POINTER(Dims), C.POINTER(Layout)]
        L.rtoc_layout_for_dims.restype = None
        L.rtoc_set_grid.argtypes = [vp, C.POINTER(Grid), C.c_int]
        L.rtoc_set_stream.argtypes = [vp, vp]
        L.rtoc_set_option.argtypes = [vp, C.c_int, C.c_int64]
        L.rtoc_upload.argtypes = [vp, C.c_int, C.c_size_t, dp, C.c_size_t]
        L.rtoc_download.argtypes = [vp, C.c_int, C.c_size_t, dp, C.c_size_t]
        L.rtoc_device_ptr.argtypes = [vp, C.c_int]
        L.rtoc_device_ptr.restype = vp
        L.rtoc_buffer_count.argtypes = [vp, C.c_int]
        L.rtoc_buffer_count.restype = C.c_size_t
        L.rtoc_bind.argtypes = [vp, C.c_int, vp]
        for f in ("rtoc_condense", "rtoc_riccati_backward", "rtoc_riccati_forward", "rtoc_riccati_sweep", "rtoc_update",
                  "rtoc_correct_state_equation", "rtoc_correct_costate_direction",
                  "rtoc_compute_initial_state_direction", "rtoc_unconstr_condense", "rtoc_integrate_solution",
                  "rtoc_linearize_state_equation", "rtoc_contact_eval_kkt", "rtoc_clear_status", "rtoc_sync"):
            getattr(L, f).argtypes = [vp]
        L.rtoc_unconstr_backward.argtypes = [vp, C.c_double]
        L.rtoc_unconstr_forward.argtypes = [vp, C.c_double]
        L.rtoc_unconstr_expand.argtypes = [vp, C.c_double]
        L.rtoc_expand.argtypes = [vp, C.c_double]
        L.rtoc_status.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int]
        L.rtoc_time_phase.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.rtoc_gather_directions.argtypes = [vp, vp, dp]
        L.rtoc_set_constraint_rows.argtypes = [vp, C.POINTER(BoxRow), C.c_int]
        L.rtoc_set_friction_cones.argtypes = [vp, C.c_int, C.c_int]
        L.rtoc_set_wrench_cones.argtypes = [vp, C.c_int]
        L.rtoc_newton_iteration.argtypes = [vp, C.c_double, C.c_double]
        L.rtoc_converged_count.argtypes = [vp, C.POINTER(C.c_int)]
        L.rtoc_wrench_cone_matrix.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.rtoc_save_stage_dump.argtypes = [vp, C.c_char_p, C.c_uint]
        L.rtoc_kkt_error.argtypes = [vp, dp, C.c_int]
        L.rtoc_check_fxx_structure.argtypes = [vp, C.POINTER(C.c_int)]
        L.rtoc_sto_eval_kkt.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int]
        L.rtoc_clone.argtypes = [vp, C.POINTER(vp)]
        L.rtoc_set_robot_model.argtypes = [vp, vp]
        L.rtoc_set_configuration_cost.argtypes = [vp, vp]
        L.rtoc_set_initial_state.argtypes = [vp, dp, C.c_int]
        L.rtoc_unconstr_eval_kkt.argtypes = [vp, C.c_double]
        L.rtoc_set_constraint_bounds.argtypes = [vp, dp, C.c_int, C.c_double, C.c_double]
        L.rtoc_unconstr_init_constraints.argtypes = [vp]
        L.rtoc_contact_init_constraints.argtypes = [vp]
        L.rtoc_set_barrier_param.argtypes = [vp, C.c_double, C.c_double]
        L.rtoc_set_friction_coefficients.argtypes = [vp, dp, C.c_int]
        L.rtoc_set_wrench_cone_params.argtypes = [vp, dp, C.c_int]
        L.rtoc_unconstr_update_solution.argtypes = [vp, C.c_double, dp, C.c_int]
        L.rtoc_contact_update_solution.argtypes = [vp, C.c_double, dp, C.c_int]
        L.rtoc_line_search_filter.argtypes = [vp, dp, dp, C.POINTER(C.c_int), C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.rtoc_line_search_clear.argtypes = [vp]
        L.rtoc_linearize_contact_dynamics.argtypes = [vp, C.c_int]
        L.rtoc_set_contact_schedule.argtypes = [vp, C.POINTER(C.c_uint), dp, dp]
        L.rtoc_load_stage_dump.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.rtoc_graph_replay_count.argtypes = [vp, C.POINTER(C.c_ulonglong)]
        L.rtoc_sto_set_problem.argtypes = [vp, C.c_double, C.c_double, dp, C.c_int, C.c_int, dp, C.c_double, C.c_double]
        L.rtoc_sto_set_regularization.argtypes = [vp, C.c_double]
        L.rtoc_sto_set_cost_terms.argtypes = [vp, dp, dp]
        for f in ("rtoc_sto_init_constraints", "rtoc_sto_correct_time_steps", "rtoc_sto_eval_kkt_device",
                  "rtoc_sto_compute_step_sizes", "rtoc_sto_integrate_solution"):
            getattr(L, f).argtypes = [vp]
        for f in ("rtoc_sto_get_event_times", "rtoc_sto_get_time_steps", "rtoc_sto_get_constraint_data"):
            getattr(L, f).argtypes = [vp, dp, C.c_int]
        L.rtoc_sto_get_kkt_terms.argtypes = [vp, dp, dp, dp, C.c_int]
        L.rtoc_sto_set_slack_dual.argtypes = [vp, dp, dp]
        L.rtoc_contact_eval_ocp.argtypes = [vp, C.c_int, dp, dp, C.c_int]
        L.rtoc_set_line_search.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.rtoc_contact_line_search.argtypes = [vp, C.POINTER(C.c_int)]
        L.rtoc_error_string.argtypes = [C.c_int]
        L.rtoc_error_string.restype = C.c_char_p
        _LIB = L
    return _LIB


def layout_for(dims):
    out = Layout()
    lib().rtoc_layout_for_dims(C.byref(dims), C.byref(out))
    return out


def _chk(rc):
    if rc != 0:
        raise RtocError("rtoc error %d: %s" % (rc, lib().rtoc_error_string(rc).decode()))


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """Thin RAII wrapper over rtoc_ctx: one context per (device, problem shape, batch)."""

    def __init__(self, dims, max_stages, batch, device=0):
        self.dims = dims
        self.batch = batch
        self.max_stages = max_stages
        self._h = C.c_void_p()
        _chk(lib().rtoc_create(C.byref(dims), max_stages, batch, device, C.byref(self._h)))
        self.L = Layout()
        _chk(lib().rtoc_get_layout(self._h, C.byref(self.L)))
        self.nstages = 0

    @classmethod
    def from_stage_dump(cls, path, device=0):
        """rtoc_load_stage_dump: a context restored from a stage dump (robotoc_amd/replay.py format)."""
        from .replay import read_dump
        meta = read_dump(path)
        self = cls.__new__(cls)
        self.dims = meta["dims"]
        self.batch = meta["batch"]
        self.max_stages = len(meta["grids"])
        self._h = C.c_void_p()
        _chk(lib().rtoc_load_stage_dump(str(path).encode(), device, C.byref(self._h)))
        self.L = Layout()
        _chk(lib().rtoc_get_layout(self._h, C.byref(self.L)))
        self.nstages = len(meta["grids"])
        return self

    def close(self):
        if self._h:
            lib().rtoc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_grid(self, grids):
        g = grid_array(grids)
        _chk(lib().rtoc_set_grid(self._h, g, len(grids)))
        self.nstages = len(grids)

    def set_constraint_rows(self, rows):
        arr = (BoxRow * max(len(rows), 1))()
        for i, r in enumerate(rows):
            arr[i] = r
        _chk(lib().rtoc_set_constraint_rows(self._h, arr, len(rows)))

    def set_stream(self, hip_stream):
        _chk(lib().rtoc_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_writeback(self, on):
        _chk(lib().rtoc_set_option(self._h, OPT_WRITEBACK_KKT, int(bool(on))))

    def set_max_dts0(self, v):
        bits = struct.unpack("<q", struct.pack("<d", float(v)))[0]
        _chk(lib().rtoc_set_option(self._h, OPT_MAX_DTS0, bits))

    def set_contact_inv_damping(self, v):
        bits = struct.unpack("<q", struct.pack("<d", float(v)))[0]
        _chk(lib().rtoc_set_option(self._h, 3, bits))

    def set_backward_waves(self, nw):
        _chk(lib().rtoc_set_option(self._h, OPT_BACKWARD_WAVES, int(nw)))

    def upload(self, buffer, arr, offset=0):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        _chk(lib().rtoc_upload(self._h, buffer, offset, _dp(arr), arr.size))

    def download(self, buffer, shape, offset=0):
        out = np.empty(shape, dtype=np.float64)
        _chk(lib().rtoc_download(self._h, buffer, offset, _dp(out), out.size))
        return out

    def device_ptr(self, buffer):
        return lib().rtoc_device_ptr(self._h, buffer)

    def buffer_count(self, buffer):
        return lib().rtoc_buffer_count(self._h, buffer)

    def bind(self, buffer, device_ptr):
        _chk(lib().rtoc_bind(self._h, buffer, C.c_void_p(device_ptr)))

    def condense(self):
        _chk(lib().rtoc_condense(self._h))

    def riccati_backward(self):
        _chk(lib().rtoc_riccati_backward(self._h))

    def riccati_forward(self):
        _chk(lib().rtoc_riccati_forward(self._h))

    def riccati_sweep(self):
        """backward + forward, pipelined over instance chunks (rtoc_riccati_sweep)."""
        _chk(lib().rtoc_riccati_sweep(self._h))

    def set_sweep_chunks(self, n):
        _chk(lib().rtoc_set_option(self._h, OPT_SWEEP_CHUNKS, int(n)))

    def newton_iteration(self, kkt_tol, tau=0.995):
        """rtoc_newton_iteration: KKT error -> condense -> sweep -> expand -> converged-instance mask ->
        update -> integrate, one launch sequence; returns nothing (asynchronous)."""
        _chk(lib().rtoc_newton_iteration(self._h, float(kkt_tol), float(tau)))

    def converged_count(self):
        n = C.c_int(0)
        _chk(lib().rtoc_converged_count(self._h, C.byref(n)))
        return n.value

    def set_backward_scan(self, on):
        """RTOC_OPT_BACKWARD_SCAN: backward recursion as a scan over the horizon (few instances, low latency)."""
        _chk(lib().rtoc_set_option(self._h, OPT_BACKWARD_SCAN, 2 if on == "auto" else int(bool(on))))

    def set_condense_register(self, on=True):
        """RTOC_OPT_CONDENSE_REGISTER: the register-chained condensation kernel (one wave per contact grid point) where it applies;
        "cones": also in contexts with friction / wrench cone rows."""
        _chk(lib().rtoc_set_option(self._h, OPT_CONDENSE_REGISTER, 2 if on == "cones" else int(bool(on))))

    def set_backward_register(self, on):
        """RTOC_OPT_BACKWARD_REGISTER: the register-resident backward kernel (one wave per instance) where it applies;
        2: the iCub-size shapes' register-wide kernel on every batch size (1: on batches larger than the device's CU count)."""
        _chk(lib().rtoc_set_option(self._h, OPT_BACKWARD_REGISTER, 2 if on == 2 else int(bool(on))))

    def set_condense_split(self, on):
        """RTOC_OPT_CONDENSE_SPLIT: MJtJinv in its own kernel or one fused condensation kernel (default per robot shape)."""
        _chk(lib().rtoc_set_option(self._h, OPT_CONDENSE_SPLIT, int(bool(on))))

    def get_option(self, option):
        """rtoc_get_option: the value in force of an integer-valued option"""
        v = C.c_int64(0)
        lib().rtoc_get_option.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        _chk(lib().rtoc_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def set_condense_keep_qaf(self, on):
        """RTOC_OPT_CONDENSE_KEEP_QAF: also store Qafqv / Qafu_full in the ContactDynamicsData record."""
        _chk(lib().rtoc_set_option(self._h, OPT_CONDENSE_KEEP_QAF, int(bool(on))))

    def sto_eval_kkt(self, lt, qtt_diag):
        """rtoc_sto_eval_kkt: lt, qtt_diag [batch, num_events]; returns the squared STO KKT-error term per instance."""
        lt = np.ascontiguousarray(lt, dtype=np.float64)
        qtt_diag = np.ascontiguousarray(qtt_diag, dtype=np.float64)
        assert lt.shape == qtt_diag.shape and lt.shape[0] == self.batch
        out = np.zeros(self.batch)
        _chk(lib().rtoc_sto_eval_kkt(self._h, _dp(lt), _dp(qtt_diag), lt.shape[1], _dp(out), self.batch))
        return out

    # ---- SwitchingTimeOptimization resident on the device (rtoc.h: rtoc_sto_*) ----
    STO_MAX_ROWS = 16

    def sto_set_problem(self, t0, T, event_times, min_dwell_times, barrier_param=1.0e-3, fraction_to_boundary_rule=0.995):
        """event_times: [num_events] (shared by the batch) or [batch, num_events]; min_dwell_times: [num_events + 1]."""
        et = np.ascontiguousarray(event_times, dtype=np.float64)
        md = np.ascontiguousarray(min_dwell_times, dtype=np.float64)
        nev = et.shape[-1] if et.size else 0
        assert et.ndim in (1, 2) and (et.ndim == 1 or et.shape[0] == self.batch) and md.size == nev + 1
        _chk(lib().rtoc_sto_set_problem(self._h, t0, T, _dp(et), nev, int(et.ndim == 2), _dp(md), barrier_param, fraction_to_boundary_rule))
        self.sto_nev = nev

    def sto_set_regularization(self, sto_reg):
        _chk(lib().rtoc_sto_set_regularization(self._h, sto_reg))

    def sto_set_cost_terms(self, lt=None, qtt_diag=None):
        if lt is None:
            _chk(lib().rtoc_sto_set_cost_terms(self._h, None, None))
            return
        lt, qtt_diag = np.ascontiguousarray(lt, dtype=np.float64), np.ascontiguousarray(qtt_diag, dtype=np.float64)
        assert lt.shape == qtt_diag.shape == (self.batch, self.sto_nev)
        _chk(lib().rtoc_sto_set_cost_terms(self._h, _dp(lt), _dp(qtt_diag)))

    def sto_init_constraints(self):
        _chk(lib().rtoc_sto_init_constraints(self._h))

    def sto_set_slack_dual(self, slack, dual):
        slack, dual = np.ascontiguousarray(slack, dtype=np.float64), np.ascontiguousarray(dual, dtype=np.float64)
        assert slack.shape == dual.shape == (self.batch, self.sto_nev + 1)
        _chk(lib().rtoc_sto_set_slack_dual(self._h, _dp(slack), _dp(dual)))

    def sto_correct_time_steps(self):
        _chk(lib().rtoc_sto_correct_time_steps(self._h))

    def sto_eval_kkt_device(self):
        _chk(lib().rtoc_sto_eval_kkt_device(self._h))

    def sto_compute_step_sizes(self):
        _chk(lib().rtoc_sto_compute_step_sizes(self._h))

    def sto_integrate_solution(self):
        _chk(lib().rtoc_sto_integrate_solution(self._h))

    def sto_event_times(self):
        out = np.zeros((self.batch, self.sto_nev))
        _chk(lib().rtoc_sto_get_event_times(self._h, _dp(out), self.batch))
        return out

    def sto_time_steps(self):
        out = np.zeros((self.batch, self.nstages))
        _chk(lib().rtoc_sto_get_time_steps(self._h, _dp(out), self.batch))
        return out

    def sto_constraint_data(self):
        """[batch, 6, num_events + 1]: slack, dual, residual, cmpl, dslack, ddual of the dwell-time rows."""
        out = np.zeros((self.batch, 6, self.STO_MAX_ROWS))
        _chk(lib().rtoc_sto_get_constraint_data(self._h, _dp(out), self.batch))
        return out[:, :, :self.sto_nev + 1].copy()

    def sto_kkt_terms(self):
        """(lt, diag Qtt, squared STO KKT term) of the last SwitchingTimeOptimization::evalKKT on the device."""
        lt, qtt, err = np.zeros((self.batch, self.sto_nev)), np.zeros((self.batch, self.sto_nev)), np.zeros(self.batch)
        _chk(lib().rtoc_sto_get_kkt_terms(self._h, _dp(lt), _dp(qtt), _dp(err), self.batch))
        return lt, qtt, err

    def graph_replay_count(self):
        n = C.c_ulonglong()
        _chk(lib().rtoc_graph_replay_count(self._h, C.byref(n)))
        return int(n.value)

    def set_graph(self, on):
        """RTOC_OPT_GRAPH: replay rtoc_riccati_sweep / rtoc_newton_iteration from captured hipGraphs."""
        _chk(lib().rtoc_set_option(self._h, OPT_GRAPH, int(bool(on))))

    def set_switching_transport(self, exact):
        """RTOC_OPT_SWITCHING_TRANSPORT: free-flyer block of the switching-constraint Jacobians -- False (default): as the
        reference composes it (Pq dIntegrate^T); True: the chain rule (Pq dIntegrate)."""
        _chk(lib().rtoc_set_option(self._h, OPT_SWITCHING_TRANSPORT, int(bool(exact))))

    def set_fxx_structure(self, mode):
        """RTOC_OPT_FXX_STRUCTURE: 0 automatic (checked on the device), 1 always dense, 2 caller asserts the structure."""
        _chk(lib().rtoc_set_option(self._h, OPT_FXX_STRUCTURE, int(mode)))

    def check_fxx_structure(self):
        """rtoc_check_fxx_structure: True iff every resident Fxx has the state-equation structure."""
        out = C.c_int(0)
        _chk(lib().rtoc_check_fxx_structure(self._h, C.byref(out)))
        return bool(out.value)

    def correct_state_equation(self):
        _chk(lib().rtoc_correct_state_equation(self._h))

    def correct_costate_direction(self):
        _chk(lib().rtoc_correct_costate_direction(self._h))

    def compute_initial_state_direction(self):
        _chk(lib().rtoc_compute_initial_state_direction(self._h))

    def integrate_solution(self):
        _chk(lib().rtoc_integrate_solution(self._h))

    # ---- filter line search (line_search_filter.cpp), batched ----
    def line_search_filter(self, cost, violation, mask=None, cost_reduction_rate=0.005, constraint_violation_reduction_rate=0.005):
        """rtoc_line_search_filter: accept test + filter update of every instance; returns accepted [batch] (0/1).
        Default rates: LineSearchSettings (include/robotoc/line_search/line_search_settings.hpp)."""
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        violation = np.ascontiguousarray(violation, dtype=np.float64)
        n = cost.shape[0]
        acc = np.zeros(n, dtype=np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        _chk(lib().rtoc_line_search_filter(self._h, _dp(cost), _dp(violation), m.ctypes.data_as(ip) if m is not None else None, n,
                                           cost_reduction_rate, constraint_violation_reduction_rate, acc.ctypes.data_as(ip)))
        return acc

    def contact_eval_ocp(self, trial=False):
        """rtoc_contact_eval_ocp: (cost + cost_barrier, primal_feasibility) of every instance"""
        cost, viol = np.zeros(self.batch), np.zeros(self.batch)
        _chk(lib().rtoc_contact_eval_ocp(self._h, int(bool(trial)), _dp(cost), _dp(viol), self.batch))
        return cost, viol

    def set_line_search(self, enable=True, step_size_reduction_rate=0.75, min_step_size=0.05, filter_cost_reduction_rate=0.005,
                        filter_constraint_violation_reduction_rate=0.005):
        """SolverOptions::enable_line_search + LineSearchSettings (filter method)"""
        _chk(lib().rtoc_set_line_search(self._h, int(bool(enable)), step_size_reduction_rate, min_step_size, filter_cost_reduction_rate,
                                        filter_constraint_violation_reduction_rate))

    def set_line_search_method(self, method="filter", armijo_control_rate=0.001, margin_rate=0.05, eps=1.0e-8):
        """LineSearchSettings::line_search_method: "filter" (default) or "merit" (LineSearch::meritBacktrackingLineSearch)"""
        lib().rtoc_set_line_search_method.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        _chk(lib().rtoc_set_line_search_method(self._h, {"filter": 0, "merit": 1}[method], armijo_control_rate, margin_rate, eps))

    def line_search_trials(self):
        n = C.c_int()
        _chk(lib().rtoc_line_search_trials(self._h, C.byref(n)))
        return n.value

    def line_search_merit_terms(self):
        p, d = np.zeros(self.batch), np.zeros(self.batch)
        lib().rtoc_line_search_merit_terms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        _chk(lib().rtoc_line_search_merit_terms(self._h, _dp(p), _dp(d), self.batch))
        return p, d

    def contact_line_search(self):
        n = C.c_int()
        _chk(lib().rtoc_contact_line_search(self._h, C.byref(n)))
        return n.value

    def line_search_clear(self):
        _chk(lib().rtoc_line_search_clear(self._h))

    # ---- rigid-body linearisation (include/rtoc_robot.h) ----
    def set_robot_model(self, model):
        """rtoc_set_robot_model: `model` = robotoc_amd.robot_model.RobotModel"""
        _chk(lib().rtoc_set_robot_model(self._h, C.byref(model)))
        self._model_ncontacts = model.ncontacts

    def set_contact_schedule(self, active, positions=None, rotations=None):
        """rtoc_set_contact_schedule: active [nstages] bit masks, positions [nstages, ncontacts, 3] or None,
        rotations [nstages, ncontacts, 3, 3] (row-major; surface contacts) or None"""
        act = np.ascontiguousarray(active, dtype=np.uint32)
        assert act.shape == (self.nstages,)
        pos = rot = None
        if positions is not None:
            pos = np.ascontiguousarray(positions, dtype=np.float64)
            assert pos.shape == (self.nstages, self._model_ncontacts, 3)
        if rotations is not None:
            rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(self.nstages, self._model_ncontacts, 9)
        _chk(lib().rtoc_set_contact_schedule(self._h, act.ctypes.data_as(C.POINTER(C.c_uint)), _dp(pos) if pos is not None else None,
                                             _dp(rot) if rot is not None else None))

    def set_configuration_cost(self, q_ref, v_ref, u_ref, q_weight, v_weight, a_weight, u_weight, q_weight_terminal, v_weight_terminal,
                               q_weight_impact=None, v_weight_impact=None, dv_weight_impact=None):
        """rtoc_set_configuration_cost (ConfigurationSpaceCost; q_ref on the manifold: nq entries)"""
        from .robot_model import MAX_JOINTS
        arr = np.zeros((12, MAX_JOINTS))
        z = np.zeros(self.dims.nv)
        vals = (q_ref, v_ref, u_ref, q_weight, v_weight, a_weight, u_weight, q_weight_terminal, v_weight_terminal,
                z if q_weight_impact is None else q_weight_impact, z if v_weight_impact is None else v_weight_impact,
                z if dv_weight_impact is None else dv_weight_impact)
        for k, v in enumerate(vals):
            v = np.asarray(v, dtype=np.float64)
            arr[k, :v.size] = v
        _chk(lib().rtoc_set_configuration_cost(self._h, arr.ctypes.data_as(C.c_void_p)))

    def contact_eval_kkt(self):
        _chk(lib().rtoc_contact_eval_kkt(self._h))

    def contact_update_solution(self, fraction_to_boundary_rule=0.995, want_kkt_error=True):
        out = np.zeros(self.batch) if want_kkt_error else None
        _chk(lib().rtoc_contact_update_solution(self._h, fraction_to_boundary_rule, _dp(out) if want_kkt_error else None,
                                                self.batch if want_kkt_error else 0))
        return out

    def set_initial_state(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert x0.shape == (self.batch, 2 * self.dims.nv + (1 if self.dims.np == 6 else 0))
        _chk(lib().rtoc_set_initial_state(self._h, _dp(x0), self.batch))

    def set_constraint_bounds(self, bounds, barrier_param=1.0e-3, fraction_to_boundary_rule=0.995):
        bounds = np.ascontiguousarray(bounds, dtype=np.float64)
        _chk(lib().rtoc_set_constraint_bounds(self._h, _dp(bounds), bounds.size, barrier_param, fraction_to_boundary_rule))

    def set_barrier_param(self, barrier_param=1.0e-3, fraction_to_boundary_rule=0.995):
        _chk(lib().rtoc_set_barrier_param(self._h, barrier_param, fraction_to_boundary_rule))

    def set_friction_coefficients(self, mu):
        """ContactStatus::frictionCoefficient per contact: switches the device-side evaluation of the friction-cone rows on"""
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        _chk(lib().rtoc_set_friction_coefficients(self._h, _dp(mu), mu.size))

    def set_wrench_cone_params(self, xy_mu):
        """[ncontacts][3] = sole half-extents X, Y and the friction coefficient: device-side evaluation of the wrench-cone rows"""
        xy_mu = np.ascontiguousarray(xy_mu, dtype=np.float64).reshape(-1, 3)
        _chk(lib().rtoc_set_wrench_cone_params(self._h, _dp(xy_mu), xy_mu.shape[0]))

    def contact_init_constraints(self):
        """OCPSolver::initConstraints for the rows evaluated on the device (joint limits with bounds, friction cones with mu)"""
        _chk(lib().rtoc_contact_init_constraints(self._h))

    def set_cone_jacobian(self, exact):
        """RTOC_OPT_CONE_JACOBIAN: dg/dq of the friction cones -- False (default): as the reference composes it (LOCAL-frame
        angular Jacobian x world-frame force); True: the derivative of R_wf(q) f"""
        _chk(lib().rtoc_set_option(self._h, OPT_CONE_JACOBIAN, int(bool(exact))))

    def set_linearize_fused(self, on):
        """RTOC_OPT_LINEARIZE_FUSED: rigid-body linearisation as one kernel (values recomputed per lane) instead of pre-pass + walk"""
        _chk(lib().rtoc_set_option(self._h, OPT_LINEARIZE_FUSED, int(bool(on))))

    def set_linearize_dofs_per_pass(self, dofs):
        """RTOC_OPT_LINEARIZE_DOFS_PER_PASS: tangent directions per pass of the rigid-body walk (0 = chosen per model)"""
        _chk(lib().rtoc_set_option(self._h, OPT_LINEARIZE_DOFS_PER_PASS, int(dofs)))

    def set_unconstr_dense(self, on):
        """RTOC_OPT_UNCONSTR_DENSE: the unconstrained recursion on materialised A, B (general kernels) instead of the structured one"""
        _chk(lib().rtoc_set_option(self._h, OPT_UNCONSTR_DENSE, int(bool(on))))

    def set_impact_cones(self, on):
        """RTOC_OPT_IMPACT_CONES: cone rows on impact grids (ImpactFrictionCone) or not"""
        _chk(lib().rtoc_set_option(self._h, OPT_IMPACT_CONES, int(bool(on))))

    def unconstr_init_constraints(self):
        _chk(lib().rtoc_unconstr_init_constraints(self._h))

    def linearize_state_equation(self):
        _chk(lib().rtoc_linearize_state_equation(self._h))

    def unconstr_eval_kkt(self, dt):
        _chk(lib().rtoc_unconstr_eval_kkt(self._h, dt))

    def unconstr_update_solution(self, dt, want_kkt_error=True):
        """rtoc_unconstr_update_solution; returns the KKT error of the iterate it linearised at ([batch]) or None"""
        out = np.zeros(self.batch) if want_kkt_error else None
        _chk(lib().rtoc_unconstr_update_solution(self._h, dt, _dp(out) if want_kkt_error else None, self.batch if want_kkt_error else 0))
        return out

    def linearize_contact_dynamics(self, augment_residual=False):
        _chk(lib().rtoc_linearize_contact_dynamics(self._h, int(bool(augment_residual))))

    def kkt_error(self):
        """rtoc_kkt_error: sqrt of the squared KKT residual of every instance."""
        out = np.empty(self.batch, dtype=np.float64)
        _chk(lib().rtoc_kkt_error(self._h, _dp(out), self.batch))
        return out

    def save_stage_dump(self, path, buffers):
        """rtoc_save_stage_dump: `buffers` = iterable of RTOC_BUF_* indices."""
        mask = 0
        for b in buffers:
            mask |= 1 << b
        _chk(lib().rtoc_save_stage_dump(self._h, str(path).encode(), mask))

    def set_friction_cones(self, max_contacts, contact_dim=3):
        _chk(lib().rtoc_set_friction_cones(self._h, int(max_contacts), int(contact_dim)))

    def set_wrench_cones(self, max_contacts):
        """ContactWrenchCone rows (17 per active surface contact); replaces friction cones if set."""
        _chk(lib().rtoc_set_wrench_cones(self._h, int(max_contacts)))

    def unconstr_condense(self):
        _chk(lib().rtoc_unconstr_condense(self._h))

    def unconstr_expand(self, dt):
        _chk(lib().rtoc_unconstr_expand(self._h, dt))

    def unconstr_backward(self, dt):
        _chk(lib().rtoc_unconstr_backward(self._h, dt))

    def unconstr_forward(self, dt):
        _chk(lib().rtoc_unconstr_forward(self._h, dt))

    def expand(self, tau=0.995):
        _chk(lib().rtoc_expand(self._h, tau))

    def update(self):
        _chk(lib().rtoc_update(self._h))

    def status(self):
        out = (C.c_uint32 * self.batch)()
        _chk(lib().rtoc_status(self._h, out, self.batch))
        return np.frombuffer(out, dtype=np.uint32).copy()

    def clear_status(self):
        _chk(lib().rtoc_clear_status(self._h))

    def sync(self):
        _chk(lib().rtoc_sync(self._h))

    def gather_directions(self, nccl_comm, out_device_ptr):
        """rtoc_gather_directions: RCCL all-gather of RTOC_BUF_DIR of every rank into device memory of
        world_size * buffer_count(BUF_DIR) doubles; nccl_comm: ncclComm_t as an int / c_void_p."""
        _chk(lib().rtoc_gather_directions(self._h, C.c_void_p(nccl_comm if isinstance(nccl_comm, int) else nccl_comm.value),
                                          C.cast(C.c_void_p(out_device_ptr), C.POINTER(C.c_double))))

    def time_phase(self, phase, reps):
        ms = C.c_float()
        _chk(lib().rtoc_time_phase(self._h, phase, reps, C.byref(ms)))
        return ms.value

    # convenience shapes
    def shape(self, which):
        rl = getattr(self.L, which)
        return (self.batch, self.nstages, rl.stride)

    def upload_records(self, buffer, arr):
        """arr: [batch, nstages, stride] -> device layout [batch][max? no: nstages] (stride packed)."""
        assert arr.shape[1] == self.nstages
        self.upload(buffer, arr)

    def download_records(self, buffer, which):
        return self.download(buffer, self.shape(which))


class SolveOptions(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("kkt_tol", C.c_double), ("sto_enabled", C.c_int), ("initial_sto_reg_iter", C.c_int),
                ("initial_sto_reg", C.c_double), ("kkt_tol_mesh", C.c_double), ("max_dt_mesh", C.c_double)]


_REG_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)
_UPD_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
_REF_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)


class SolveCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("set_sto_regularization", _REG_CB), ("update_solution", _UPD_CB), ("max_time_step", _UPD_CB),
                ("mesh_refinement", _REF_CB)]


class SolveStats(C.Structure):
    _fields_ = [("convergence", C.c_int), ("iter", C.c_int), ("num_mesh_refinements", C.c_int), ("mesh_refinement_iter", C.c_int * 64)]


def solve_loop(max_iter, kkt_tol, update_solution, sto_enabled=False, initial_sto_reg_iter=0, initial_sto_reg=0.0, kkt_tol_mesh=0.0,
               max_dt_mesh=0.0, set_sto_regularization=None, max_time_step=None, mesh_refinement=None):
    """rtoc_solve_loop (include/rtoc_robot.h): OCPSolver::solve's iteration schedule (ocp_solver.cpp:169-213), the one the C++ shell
    runs too.  update_solution() -> KKT error; max_time_step() -> float; set_sto_regularization(reg), mesh_refinement() -> None.
    Returns (convergence, iter, mesh_refinement_iter).  An exception raised by a callback aborts the loop and is re-raised."""
    pending = []

    def guard(fn):
        def run(*args):
            try:
                return fn(*args)
            except BaseException as e:   # (must not propagate through the C frame)
                pending.append(e)
                return -100
        return run

    def _upd(_, out):
        out[0] = float(update_solution())
        return 0

    def _mdt(_, out):
        out[0] = float(max_time_step())
        return 0

    def _reg(_, reg):
        set_sto_regularization(reg)
        return 0

    def _ref(_):
        mesh_refinement()
        return 0
    cb = SolveCallbacks(None, _REG_CB(guard(_reg)) if set_sto_regularization else _REG_CB(), _UPD_CB(guard(_upd)),
                        _UPD_CB(guard(_mdt)) if max_time_step else _UPD_CB(), _REF_CB(guard(_ref)) if mesh_refinement else _REF_CB())
    opt = SolveOptions(int(max_iter), float(kkt_tol), int(bool(sto_enabled)), int(initial_sto_reg_iter), float(initial_sto_reg),
                       float(kkt_tol_mesh), float(max_dt_mesh))
    st = SolveStats()
    L = lib()
    L.rtoc_solve_loop.argtypes = [C.POINTER(SolveOptions), C.POINTER(SolveCallbacks), C.POINTER(SolveStats)]
    rc = L.rtoc_solve_loop(C.byref(opt), C.byref(cb), C.byref(st))
    if pending:
        raise pending[0]
    _chk(rc)
    return bool(st.convergence), st.iter, [st.mesh_refinement_iter[k] for k in range(min(st.num_mesh_refinements, 64))]


def debug_profile(ctx):
    """Tuning aid: first call attaches the stamp buffer, later calls return [max_stages,32] int64."""
    L = lib()
    L.rtoc_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros((ctx.max_stages, 32), dtype=np.int64)
    _chk(L.rtoc_debug_profile(ctx._h, out.ctypes.data_as(C.c_void_p)))
    return out


class LinearizePlan(C.Structure):
    _fields_ = [("nlevels", C.c_int), ("nbranch", C.c_int), ("dofs_per_pass", C.c_int), ("npass", C.c_int), ("lds_bytes", C.c_int)]


def robot_model_plan(model, forced_dofs_per_pass=0):
    """rtoc_robot_model_plan: the storage plan and the passes rtoc_set_robot_model would give the tangent walk of `model`
    (host arithmetic: no device needed).  Returns (LinearizePlan, [bodies pass p visits])."""
    plan = LinearizePlan()
    masks = (C.c_ulonglong * 64)()
    lib().rtoc_robot_model_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _chk(lib().rtoc_robot_model_plan(C.byref(model), int(forced_dofs_per_pass), C.byref(plan), masks))
    return plan, [[i for i in range(model.njoints) if (masks[p] >> i) & 1] for p in range(plan.npass)]


def bandwidth_probe(device=0, nbytes=1 << 31):
    """rtoc_bandwidth_probe: (read GB/s, copy GB/s) of the library's own streaming kernels on `device`"""
    r, c = C.c_double(0.0), C.c_double(0.0)
    lib().rtoc_bandwidth_probe.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _chk(lib().rtoc_bandwidth_probe(int(device), int(nbytes), C.byref(r), C.byref(c)))
    return r.value, c.value


def wrench_cone_matrix(X, Y, mu):
    """rtoc_wrench_cone_matrix -> 17 x 6 array (ContactWrenchCone::computeCone)."""
    out = np.zeros(102)
    _chk(lib().rtoc_wrench_cone_matrix(X, Y, mu, _dp(out)))
    return out.reshape(6, 17).T.copy()
