"""ctypes binding of include/cvo_hip.h (csrc/libcvo_hip.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcvo_hip.so")
_LIB = None

MODE_CVO, MODE_ACVO, MODE_MATLAB = 0, 1, 2   # MATLAB: default_params() only (mode CVO + color_scale)
FEAT_COLMAJOR, FEAT_ROWMAJOR = 0, 1

# every symbol include/cvo_hip.h declares (tests check the library exports all)
SYMBOLS = [
    "cvo_hip_error_string", "cvo_hip_last_error", "cvo_hip_device_count",
    "cvo_hip_default_params", "cvo_hip_init_state", "cvo_hip_create", "cvo_hip_destroy",
    "cvo_hip_set_params", "cvo_hip_set_fixed", "cvo_hip_set_moving",
    "cvo_hip_set_fixed_device", "cvo_hip_set_moving_device",
    "cvo_hip_swap_moving_to_fixed", "cvo_hip_set_pcd_many", "cvo_hip_get_device_cloud", "cvo_hip_range_filter_grid_average", "cvo_hip_set_shard", "cvo_hip_shard_range",
    "cvo_hip_comm_unique_id", "cvo_hip_comm_init", "cvo_hip_set_allreduce",
    "cvo_hip_mailbox_create", "cvo_hip_mailbox_connect",
    "cvo_hip_transform_pcd", "cvo_hip_flow", "cvo_hip_step_coeffs", "cvo_hip_pick_step",
    "cvo_hip_exp_se3", "cvo_hip_dist_se3", "cvo_hip_align", "cvo_hip_align_many",
    "cvo_hip_function_inner_product", "cvo_hip_function_inner_product_clouds",
    "cvo_hip_engine_profiling", "cvo_hip_get_engine_profile", "cvo_hip_get_engine_flow_trace", "cvo_hip_get_wave_load", "cvo_hip_set_graph_capture", "cvo_hip_set_profiling", "cvo_hip_get_profile", "cvo_hip_get_graph_stats", "cvo_hip_get_run_stats", "cvo_hip_get_run_clocks", "cvo_hip_get_mirror_retries", "cvo_hip_synchronize",
    "cvo_hip_set_option", "cvo_hip_get_option",
]


class Params(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("max_iter", C.c_int32),
        ("ell_init", C.c_float), ("ell_min", C.c_float), ("ell_max_init", C.c_float),
        ("sigma", C.c_float), ("sp_thres", C.c_float), ("c_sp_thres", C.c_float),
        ("c", C.c_float), ("d", C.c_float), ("c_ell", C.c_float), ("c_sigma", C.c_float),
        ("min_step", C.c_float), ("eps", C.c_float), ("eps_2", C.c_float), ("color_scale", C.c_float),
        ("dl_step", C.c_double),
    ]


class State(C.Structure):
    _fields_ = [
        ("R", C.c_float * 9), ("T", C.c_float * 3),
        ("ell", C.c_float), ("ell_max", C.c_float),
        ("transform", C.c_float * 16), ("prev_transform", C.c_float * 16),
        ("accum_transform", C.c_float * 16),
        ("iter", C.c_int32), ("pad_", C.c_int32),
    ]


class Trace(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("exit_code", C.c_int32),
        ("ell", C.c_float), ("step", C.c_float), ("dist", C.c_float), ("pad_", C.c_float),
        ("omega", C.c_float * 3), ("v", C.c_float * 3),
        ("omega_d", C.c_double * 3), ("v_d", C.c_double * 3),
        ("bcde", C.c_double * 4), ("sum_a", C.c_double), ("dl", C.c_double),
        ("nnz", C.c_int64), ("nnz_xx", C.c_int64), ("nnz_yy", C.c_int64),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("flow_ms", C.c_double), ("flow_launches", C.c_int64), ("flow_pairs", C.c_double),
        ("step_ms", C.c_double), ("step_launches", C.c_int64), ("step_pairs", C.c_double),
        ("self_ms", C.c_double), ("self_launches", C.c_int64), ("self_pairs", C.c_double),
        ("proc_flow_ms", C.c_double), ("proc_flow_launches", C.c_int64),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


class CvoHipError(RuntimeError):
    pass


def lib():
    """Loads csrc/libcvo_hip.so; raises (never falls back) if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise CvoHipError(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    try:  # share the process's HIP runtime with torch when torch is present
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
    L.cvo_hip_error_string.restype = C.c_char_p
    L.cvo_hip_error_string.argtypes = [C.c_int]
    L.cvo_hip_last_error.restype = C.c_char_p
    L.cvo_hip_last_error.argtypes = [vp]
    L.cvo_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.cvo_hip_default_params.argtypes = [C.c_int, C.POINTER(Params)]
    L.cvo_hip_init_state.argtypes = [C.POINTER(Params), C.POINTER(State)]
    L.cvo_hip_create.argtypes = [C.c_int, vp, C.POINTER(Params), C.POINTER(vp)]
    L.cvo_hip_destroy.argtypes = [vp]
    L.cvo_hip_set_params.argtypes = [vp, C.POINTER(Params)]
    L.cvo_hip_set_fixed.argtypes = [vp, fp, fp, C.c_int, C.c_int]
    L.cvo_hip_set_moving.argtypes = [vp, fp, fp, C.c_int, C.c_int]
    L.cvo_hip_set_fixed_device.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.cvo_hip_set_moving_device.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.cvo_hip_swap_moving_to_fixed.argtypes = [vp]
    pp = C.POINTER(C.c_void_p)
    L.cvo_hip_set_pcd_many.argtypes = [pp, pp, pp, C.POINTER(C.c_int), pp, pp, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.cvo_hip_get_device_cloud.argtypes = [vp, C.c_int, fp, fp, fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_hip_range_filter_grid_average.argtypes = [C.c_int, fp, C.POINTER(C.c_ubyte), C.c_int, C.c_float, C.c_float,
                                                    C.c_double, fp, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
    L.cvo_hip_set_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.cvo_hip_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
    L.cvo_hip_comm_unique_id.argtypes = [vp]
    L.cvo_hip_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.cvo_hip_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.cvo_hip_mailbox_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.cvo_hip_mailbox_connect.argtypes = [vp, vp, C.POINTER(vp)]
    L.cvo_hip_transform_pcd.argtypes = [vp, fp, fp]
    L.cvo_hip_flow.argtypes = [vp, C.c_float, dp]
    L.cvo_hip_step_coeffs.argtypes = [vp, fp, fp, C.c_float, dp]
    L.cvo_hip_pick_step.argtypes = [dp, C.c_float, fp]
    L.cvo_hip_exp_se3.argtypes = [fp, fp, C.c_float, fp, fp]
    L.cvo_hip_dist_se3.argtypes = [fp, fp, C.c_float, fp]
    L.cvo_hip_align.argtypes = [vp, C.POINTER(State), C.POINTER(Trace), C.c_int,
                                C.POINTER(C.c_int)]
    L.cvo_hip_align_many.argtypes = [C.POINTER(vp), C.POINTER(C.POINTER(State)), C.POINTER(C.c_int),
                                     C.c_int]
    L.cvo_hip_function_inner_product.argtypes = [vp, C.c_float, fp]
    L.cvo_hip_function_inner_product_clouds.argtypes = [vp, C.c_float, fp, fp, C.c_int, fp, fp, C.c_int,
                                                        C.c_int, fp]
    L.cvo_hip_engine_profiling.argtypes = [C.c_int]
    L.cvo_hip_get_engine_profile.argtypes = [dp, C.POINTER(C.c_longlong), dp, C.c_int]
    L.cvo_hip_get_engine_flow_trace.argtypes = [fp, fp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int]
    L.cvo_hip_get_wave_load.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int)]
    L.cvo_hip_set_graph_capture.argtypes = [vp, C.c_int]
    L.cvo_hip_set_profiling.argtypes = [vp, C.c_int]
    L.cvo_hip_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.cvo_hip_get_graph_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.cvo_hip_get_run_clocks.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.cvo_hip_get_mirror_retries.argtypes = []
    L.cvo_hip_get_mirror_retries.restype = C.c_longlong
    L.cvo_hip_get_run_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cvo_hip_synchronize.argtypes = [vp]
    L.cvo_hip_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.cvo_hip_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    for name in SYMBOLS:   # raises AttributeError if the library lacks a declared symbol
        if name not in ("cvo_hip_error_string", "cvo_hip_last_error"):
            getattr(L, name).restype = C.c_int
    _LIB = L
    return L


def check(status, ctx=None, what=""):
    if status == 0:
        return
    L = lib()
    msg = L.cvo_hip_error_string(status).decode()
    if ctx:
        detail = L.cvo_hip_last_error(ctx).decode()
        if detail:
            msg += " (%s)" % detail
    raise CvoHipError("%s: %s [%d]" % (what or "cvo_hip", msg, status))


def device_count():
    n = C.c_int(0)
    check(lib().cvo_hip_device_count(C.byref(n)))
    return n.value


def default_params(mode=MODE_CVO):
    p = Params()
    check(lib().cvo_hip_default_params(mode, C.byref(p)), what="default_params")
    return p


def init_state(p):
    s = State()
    check(lib().cvo_hip_init_state(C.byref(p), C.byref(s)), what="init_state")
    return s


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def trace_to_dict(t):
    return dict(k=t.k, exit_code=t.exit_code, ell=t.ell, step=t.step, dist=t.dist,
                omega=list(t.omega), v=list(t.v), omega_d=list(t.omega_d), v_d=list(t.v_d),
                bcde=list(t.bcde), sum_a=t.sum_a, dl=t.dl, nnz=t.nnz, nnz_xx=t.nnz_xx,
                nnz_yy=t.nnz_yy)


def pick_step(bcde, min_step=0.2):
    b = np.ascontiguousarray(bcde, dtype=np.float64)
    out = C.c_float()
    check(lib().cvo_hip_pick_step(dptr(b), np.float32(min_step), C.byref(out)))
    return out.value


def exp_se3(omega, v, dt):
    omega, v = f32(omega), f32(v)
    dR, dT = np.zeros(9, np.float32), np.zeros(3, np.float32)
    check(lib().cvo_hip_exp_se3(fptr(omega), fptr(v), np.float32(dt), fptr(dR), fptr(dT)))
    return dR.reshape(3, 3), dT


def dist_se3(omega, v, dt):
    omega, v = f32(omega), f32(v)
    out = C.c_float()
    check(lib().cvo_hip_dist_se3(fptr(omega), fptr(v), np.float32(dt), C.byref(out)))
    return out.value


def range_filter_grid_average(xyz, rgb, max_range, min_range, grid_size, device=0):
    """cvo_hip_range_filter_grid_average: host arrays in, host arrays out."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
    n = xyz.shape[0]
    if rgb.shape[0] != n:
        raise ValueError("xyz and rgb differ in length")
    xo = np.zeros((max(n, 1), 3), np.float32)
    co = np.zeros((max(n, 1), 3), np.uint8)
    m = C.c_int(0)
    check(lib().cvo_hip_range_filter_grid_average(
        int(device), xyz.ctypes.data_as(C.POINTER(C.c_float)), rgb.ctypes.data_as(C.POINTER(C.c_ubyte)), n,
        float(max_range), float(min_range), float(grid_size), xo.ctypes.data_as(C.POINTER(C.c_float)),
        co.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(m)), what="range_filter_grid_average")
    return xo[:m.value].copy(), co[:m.value].copy()


def shard_range(n, rank, world):
    lo, hi = C.c_int(), C.c_int()
    check(lib().cvo_hip_shard_range(n, rank, world, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


class Context:
    """Owns one cvo_hip_ctx (one device + one HIP stream)."""

    def __init__(self, params=None, mode=MODE_CVO, device=0, stream=None, graph_capture=None):
        """graph_capture: None = the library's default (hipGraphs on a stream the context
        creates itself, eager launches on a caller's stream); True / False = cvo_hip_set_graph_capture."""
        self._L = lib()
        self.params = params if params is not None else default_params(mode)
        self._ctx = C.c_void_p()
        self._cb = None
        check(self._L.cvo_hip_create(device, C.c_void_p(stream or 0), C.byref(self.params),
                                     C.byref(self._ctx)), what="cvo_hip_create")
        self.n_fixed = 0
        self.n_moving = 0
        if graph_capture is not None:
            self.set_graph_capture(graph_capture)

    def set_graph_capture(self, enable=True):
        self._chk(self._L.cvo_hip_set_graph_capture(self._ctx, int(bool(enable))), "set_graph_capture")

    def close(self):
        if self._ctx:
            self._L.cvo_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what):
        check(st, self._ctx, what)

    def set_params(self, params):
        self.params = params
        self._chk(self._L.cvo_hip_set_params(self._ctx, C.byref(params)), "set_params")

    def set_fixed(self, xyz, feat, layout=FEAT_ROWMAJOR):
        xyz, feat = f32(xyz), f32(feat)
        self.n_fixed = xyz.shape[0]
        self._chk(self._L.cvo_hip_set_fixed(self._ctx, fptr(xyz), fptr(feat), xyz.shape[0], layout),
                  "set_fixed")

    def set_moving(self, xyz, feat, layout=FEAT_ROWMAJOR):
        xyz, feat = f32(xyz), f32(feat)
        self.n_moving = xyz.shape[0]
        self._chk(self._L.cvo_hip_set_moving(self._ctx, fptr(xyz), fptr(feat), xyz.shape[0],
                                             layout), "set_moving")

    def set_fixed_device(self, d_xyz, d_feat, n, layout=FEAT_ROWMAJOR):
        """The cloud is already in device memory (addresses as integers)."""
        self.n_fixed = int(n)
        self._chk(self._L.cvo_hip_set_fixed_device(self._ctx, d_xyz, d_feat, int(n), layout), "set_fixed_device")

    def set_moving_device(self, d_xyz, d_feat, m, layout=FEAT_ROWMAJOR):
        self.n_moving = int(m)
        self._chk(self._L.cvo_hip_set_moving_device(self._ctx, d_xyz, d_feat, int(m), layout), "set_moving_device")

    def device_cloud(self, which):
        """Inspection: the device arrays of the fixed (0) / moving (1) cloud as the kernels read them."""
        rows, pts = C.c_int(), C.c_int()
        self._chk(self._L.cvo_hip_get_device_cloud(self._ctx, which, None, None, None, C.byref(rows), C.byref(pts)), "device_cloud")
        pos = np.zeros((rows.value, 4), np.float32)
        feat = np.zeros((rows.value, 8), np.float32)
        seg = np.zeros(((rows.value + 63) // 64, 4), np.float32)
        self._chk(self._L.cvo_hip_get_device_cloud(self._ctx, which, fptr(pos), fptr(feat), fptr(seg), C.byref(rows), C.byref(pts)),
                  "device_cloud")
        return {"pos": pos, "feat": feat, "seg": seg, "rows": rows.value, "points": pts.value}

    def swap_moving_to_fixed(self):
        self._chk(self._L.cvo_hip_swap_moving_to_fixed(self._ctx), "swap")
        self.n_fixed, self.n_moving = self.n_moving, 0

    def set_shard(self, row_lo, row_hi, srow_lo, srow_hi):
        self._chk(self._L.cvo_hip_set_shard(self._ctx, row_lo, row_hi, srow_lo, srow_hi),
                  "set_shard")

    def comm_init(self, id_bytes, rank, world):
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        self._chk(self._L.cvo_hip_comm_init(self._ctx, buf, rank, world), "comm_init")

    def mailbox_create(self, rank, world):
        """-> (64-byte IPC handle, device pointer) of this rank's mailbox."""
        buf = C.create_string_buffer(64)
        ptr = C.c_void_p()
        self._chk(self._L.cvo_hip_mailbox_create(self._ctx, rank, world, buf, C.byref(ptr)), "mailbox_create")
        self._mail_world = world
        return buf.raw, ptr.value

    def mailbox_connect(self, handles=None, ptrs=None):
        """handles: the ranks' IPC handles in rank order (one process per GPU) -- or ptrs: their
        device pointers (ranks inside one process)."""
        if ptrs is not None:
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            self._chk(self._L.cvo_hip_mailbox_connect(self._ctx, None, arr), "mailbox_connect")
        else:
            blob = C.create_string_buffer(b"".join(bytes(h) for h in handles), 64 * len(handles))
            self._chk(self._L.cvo_hip_mailbox_connect(self._ctx, blob, None), "mailbox_connect")

    def set_allreduce(self, fn):
        """fn(dev_ptr:int, count:int, stream:int) -> None ; sums in place over ranks."""
        def _cb(_user, buf, count, stream):
            try:
                fn(int(buf or 0), int(count), int(stream or 0))
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return -1
        self._cb = ALLREDUCE_FN(_cb)
        self._chk(self._L.cvo_hip_set_allreduce(self._ctx, self._cb, None), "set_allreduce")

    def transform_pcd(self, R, T):
        R, T = f32(R).reshape(9), f32(T).reshape(3)
        self._chk(self._L.cvo_hip_transform_pcd(self._ctx, fptr(R), fptr(T)), "transform_pcd")

    def flow(self, ell):
        out = np.zeros(13)
        self._chk(self._L.cvo_hip_flow(self._ctx, np.float32(ell), dptr(out)), "flow")
        return out

    def step_coeffs(self, omega, v, ell):
        omega, v = f32(omega), f32(v)
        out = np.zeros(4)
        self._chk(self._L.cvo_hip_step_coeffs(self._ctx, fptr(omega), fptr(v), np.float32(ell),
                                              dptr(out)), "step_coeffs")
        return out

    def align(self, state, trace_cap=2000):
        tr = (Trace * trace_cap)() if trace_cap > 0 else None
        n_it = C.c_int(0)
        self._chk(self._L.cvo_hip_align(self._ctx, C.byref(state), tr, trace_cap, C.byref(n_it)),
                  "align")
        n = n_it.value
        return n, [trace_to_dict(tr[i]) for i in range(min(n, trace_cap))]

    def function_inner_product(self, ell):
        out = C.c_float()
        self._chk(self._L.cvo_hip_function_inner_product(self._ctx, np.float32(ell), C.byref(out)),
                  "function_inner_product")
        return out.value

    def function_inner_product_clouds(self, ell, xyz_a, feat_a, xyz_b, feat_b, layout=FEAT_ROWMAJOR):
        xyz_a, feat_a, xyz_b, feat_b = f32(xyz_a), f32(feat_a), f32(xyz_b), f32(feat_b)
        out = C.c_float()
        self._chk(self._L.cvo_hip_function_inner_product_clouds(
            self._ctx, np.float32(ell), fptr(xyz_a), fptr(feat_a), xyz_a.shape[0], fptr(xyz_b), fptr(feat_b),
            xyz_b.shape[0], layout, C.byref(out)), "function_inner_product_clouds")
        return out.value

    def wave_load(self):
        """members of A kept by every wave of the last flow pass (diagnostics)."""
        buf = (C.c_uint32 * 4096)()
        n = C.c_int(0)
        self._chk(self._L.cvo_hip_get_wave_load(self._ctx, buf, 4096, C.byref(n)), "wave_load")
        return np.array(buf[:n.value], np.int64)

    def set_profiling(self, enable=True):
        self._chk(self._L.cvo_hip_set_profiling(self._ctx, int(bool(enable))), "set_profiling")

    def get_profile(self, reset=False):
        p = Profile()
        self._chk(self._L.cvo_hip_get_profile(self._ctx, C.byref(p), int(reset)), "get_profile")
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def graph_stats(self):
        """(batches launched from a cached graph, batches captured)."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._chk(self._L.cvo_hip_get_graph_stats(self._ctx, C.byref(a), C.byref(b)), "graph_stats")
        return a.value, b.value

    def run_stats(self):
        """(resident runs that executed iterations, runs that declined, iterations executed inside runs, candidates of the
        record the last run looked at) of the last align()."""
        a, b, c, d = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self._L.cvo_hip_get_run_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "run_stats")
        return a.value, b.value, c.value, d.value

    def run_clocks(self):
        buf = (C.c_longlong * 16)()
        self._chk(self._L.cvo_hip_get_run_clocks(self._ctx, buf), "run_clocks")
        return list(buf)

    def set_option(self, key, value):
        """cvo_hip_set_option: a policy / test switch of this context by name (include/cvo_hip.h lists the keys)."""
        self._chk(self._L.cvo_hip_set_option(self._ctx, key.encode(), float(value)), "set_option(%s)" % key)

    def get_option(self, key):
        v = C.c_double(0.0)
        self._chk(self._L.cvo_hip_get_option(self._ctx, key.encode(), C.byref(v)), "get_option(%s)" % key)
        return v.value

    def synchronize(self):
        self._chk(self._L.cvo_hip_synchronize(self._ctx), "synchronize")


def pinned_copy(a):
    """A float32 copy of `a` in page-locked host memory (torch's allocator).  cvo_hip_set_pcd_many stages every array into its
    own arena whatever memory it comes from (a transfer per page-locked array measured slower than one staged copy,
    profiles/r04_ab.txt 7): this only serves tests that hand over from such memory.  The array keeps its tensor alive (.base)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).pin_memory()
    return t.numpy()


def set_pcd_many(contexts, fixed, moving, layout=FEAT_ROWMAJOR):
    """The hand-over of a batch in one call (cvo_hip_set_pcd_many): fixed / moving = lists of (xyz, feat) per
    context; fixed may be None (or hold None entries): those contexts keep their fixed cloud."""
    n = len(contexts)
    keep = []

    def arr(pairs, k):
        ptrs = (C.c_void_p * n)()
        cnt = (C.c_int * n)()
        for i in range(n):
            pr = pairs[i] if pairs is not None else None
            if pr is None:
                ptrs[i] = None
                cnt[i] = 0
                continue
            a = f32(pr[k])
            keep.append(a)
            ptrs[i] = a.ctypes.data
            cnt[i] = pr[0].shape[0]
        return ptrs, cnt
    mx, mn = arr(moving, 0)
    mf, _ = arr(moving, 1)
    if fixed is not None:
        fx, fn = arr(fixed, 0)
        ff, _ = arr(fixed, 1)
    else:
        fx = ff = fn = None
    arr_c = (C.c_void_p * n)(*[c._ctx for c in contexts])
    pp = C.POINTER(C.c_void_p)
    check(lib().cvo_hip_set_pcd_many(C.cast(arr_c, pp), C.cast(fx, pp) if fx is not None else None,
                                     C.cast(ff, pp) if ff is not None else None, fn,
                                     C.cast(mx, pp), C.cast(mf, pp), mn, layout, n), what="set_pcd_many")
    for i, c in enumerate(contexts):
        if fixed is not None and fixed[i] is not None:
            c.n_fixed = fixed[i][0].shape[0]
        c.n_moving = moving[i][0].shape[0]


def align_many(contexts, states):
    """Batched mode: all registrations in flight at once (one context + stream
    each).  Returns the list of iteration counts; `states` are updated in place."""
    n = len(contexts)
    arr_c = (C.c_void_p * n)(*[c._ctx for c in contexts])
    arr_s = (C.POINTER(State) * n)(*[C.pointer(s) for s in states])
    its = (C.c_int * n)()
    check(lib().cvo_hip_align_many(arr_c, arr_s, its, n), what="align_many")
    return list(its)


def mirror_retries():
    """Times (in this process) a finished registration's pinned state copy had to be read again (incomplete when `done` was seen)."""
    return int(lib().cvo_hip_get_mirror_retries())


def engine_profiling(enable=True):
    check(lib().cvo_hip_engine_profiling(int(bool(enable))), what="engine_profiling")


def engine_profile(reset=True):
    """(kernel ms, launches, registrations served) of the fused groups' flow-pass launches."""
    ms, n, regs = C.c_double(), C.c_longlong(), C.c_double()
    check(lib().cvo_hip_get_engine_profile(C.byref(ms), C.byref(n), C.byref(regs), int(reset)), what="engine_profile")
    return ms.value, n.value, regs.value


def engine_flow_trace(reset=True, capacity=1 << 16):
    """Per flow-pass launch of the fused groups since the last reset: (duration us, period us, slots) arrays."""
    dur = np.zeros(capacity, np.float32)
    per = np.zeros(capacity, np.float32)
    sl = np.zeros(capacity, np.int32)
    n = C.c_int()
    check(lib().cvo_hip_get_engine_flow_trace(fptr(dur), fptr(per), sl.ctypes.data_as(C.POINTER(C.c_int)), capacity, C.byref(n), int(reset)),
          what="engine_flow_trace")
    k = min(n.value, capacity)
    return dur[:k].copy(), per[:k].copy(), sl[:k].copy()


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().cvo_hip_comm_unique_id(buf), what="comm_unique_id")
    return buf.raw
