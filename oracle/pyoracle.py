"""ctypes binding of oracle/liboracle.so (and oracle/_ref/libnanoflann_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

MODE_CVO, MODE_ACVO, MODE_MATLAB = 0, 1, 2
SEARCH_DENSE, SEARCH_GRID = 0, 1


class Params(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("max_iter", C.c_int32),
        ("ell_init", C.c_float), ("ell_min", C.c_float), ("ell_max_init", C.c_float),
        ("sigma", C.c_float), ("sp_thres", C.c_float), ("c_sp_thres", C.c_float),
        ("c", C.c_float), ("d", C.c_float), ("c_ell", C.c_float), ("c_sigma", C.c_float),
        ("min_step", C.c_float), ("eps", C.c_float), ("eps_2", C.c_float), ("color_scale", C.c_float),
        ("dl_step", C.c_double),
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


class State(C.Structure):
    _fields_ = [
        ("R", C.c_float * 9), ("T", C.c_float * 3),
        ("ell", C.c_float), ("ell_max", C.c_float),
        ("transform", C.c_float * 16), ("prev_transform", C.c_float * 16),
        ("accum_transform", C.c_float * 16),
        ("iter", C.c_int32), ("pad_", C.c_int32),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)


def build(ref=True):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        L.cvo_oracle_default_params.argtypes = [C.c_int, C.POINTER(Params)]
        L.cvo_oracle_init_state.argtypes = [C.POINTER(Params), C.POINTER(State)]
        L.cvo_oracle_set_threads.argtypes = [C.c_int]
        L.cvo_oracle_get_threads.restype = C.c_int
        L.cvo_oracle_thresholds.argtypes = [C.POINTER(Params), C.c_float, fp]
        L.cvo_oracle_transform.argtypes = [fp, fp, fp, C.c_int, fp]
        L.cvo_oracle_se_kernel.argtypes = [
            C.POINTER(Params), C.c_float, C.c_float, fp, fp, C.c_int, fp, fp, C.c_int, C.c_int,
            C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int32)),
            C.POINTER(fp)]
        L.cvo_oracle_se_kernel.restype = C.c_int
        L.cvo_oracle_free.argtypes = [C.c_void_p]
        L.cvo_oracle_radius_sets.argtypes = [
            fp, C.c_int, fp, C.c_int, C.c_float, C.c_int,
            C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(fp)]
        L.cvo_oracle_radius_sets.restype = C.c_int
        L.cvo_oracle_flow.argtypes = [
            C.POINTER(Params), C.c_float, fp, C.c_int, fp, C.c_int,
            C.POINTER(C.c_int64), C.POINTER(C.c_int32), fp, dp, dp, dp, dp]
        L.cvo_oracle_step_coeffs.argtypes = [
            C.c_float, fp, fp, fp, C.c_int, fp, C.c_int,
            C.POINTER(C.c_int64), C.POINTER(C.c_int32), fp, dp]
        L.cvo_oracle_pick_step.argtypes = [dp, C.c_float]
        L.cvo_oracle_pick_step.restype = C.c_float
        L.cvo_oracle_exp_se3.argtypes = [fp, fp, C.c_float, fp, fp]
        L.cvo_oracle_dist_se3.argtypes = [fp, fp, C.c_float]
        L.cvo_oracle_dist_se3.restype = C.c_float
        L.cvo_oracle_function_inner_product.argtypes = [
            C.POINTER(Params), C.c_float, fp, fp, C.c_int, fp, fp, C.c_int, C.c_int]
        L.cvo_oracle_function_inner_product.restype = C.c_float
        L.cvo_oracle_align.argtypes = [
            C.POINTER(Params), C.POINTER(State), fp, fp, C.c_int, fp, fp, C.c_int, C.c_int,
            C.POINTER(Trace), C.c_int]
        L.cvo_oracle_align.restype = C.c_int
        L.cvo_oracle_align_sharded.argtypes = [
            C.POINTER(Params), C.POINTER(State), fp, fp, C.c_int, fp, fp, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p,
            C.POINTER(Trace), C.c_int]
        L.cvo_oracle_align_sharded.restype = C.c_int
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own nanoflann behind a C-ABI, or None if not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libnanoflann_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        R.ref_radius_search.argtypes = [
            fp, C.c_int, fp, C.c_int, C.c_float, C.POINTER(C.c_int64),
            C.POINTER(C.POINTER(C.c_int32)), C.POINTER(fp)]
        R.ref_radius_search.restype = C.c_int
        R.ref_free.argtypes = [C.c_void_p]
        _REF = R
    return _REF


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def default_params(mode=MODE_CVO):
    p = Params()
    lib().cvo_oracle_default_params(mode, C.byref(p))
    return p


def init_state(p):
    s = State()
    lib().cvo_oracle_init_state(C.byref(p), C.byref(s))
    return s


def set_threads(n):
    lib().cvo_oracle_set_threads(int(n))


VAR_ROWSUM_SEQ, VAR_ROWSUM_PACKET, VAR_D2_PLAIN = 1, 2, 4


def set_variant(flags):
    """Deviation study only (tests/test_oracle_variants.py); 0 = the contract."""
    lib().cvo_oracle_set_variant.argtypes = [C.c_uint]
    lib().cvo_oracle_set_variant(int(flags))


def get_threads():
    return lib().cvo_oracle_get_threads()


def thresholds(p, ell):
    tau = np.zeros(2, np.float32)
    lib().cvo_oracle_thresholds(C.byref(p), np.float32(ell), _fp(tau))
    return float(tau[0]), float(tau[1])


def transform(R, T, y0):
    R = _f32(R).reshape(9)
    T = _f32(T).reshape(3)
    y0 = _f32(y0)
    out = np.empty_like(y0)
    lib().cvo_oracle_transform(_fp(R), _fp(T), _fp(y0), y0.shape[0], _fp(out))
    return out


def se_kernel(p, ell, xa, fa, xb, fb, search=SEARCH_DENSE, c_sp=None):
    """Returns CSR (row_ptr int64[na+1], col int32[nnz], val float32[nnz])."""
    xa, fa, xb, fb = _f32(xa), _f32(fa), _f32(xb), _f32(fb)
    if c_sp is None:
        c_sp = p.c_sp_thres if p.mode == MODE_ACVO else p.sp_thres
    rp = C.POINTER(C.c_int64)()
    col = C.POINTER(C.c_int32)()
    val = C.POINTER(C.c_float)()
    rc = lib().cvo_oracle_se_kernel(C.byref(p), np.float32(ell), np.float32(c_sp), _fp(xa), _fp(fa),
                                    xa.shape[0], _fp(xb), _fp(fb), xb.shape[0], search,
                                    C.byref(rp), C.byref(col), C.byref(val))
    if rc != 0:
        raise MemoryError("cvo_oracle_se_kernel failed")
    na = xa.shape[0]
    row_ptr = np.ctypeslib.as_array(rp, shape=(na + 1,)).copy()
    nnz = int(row_ptr[-1])
    cols = np.ctypeslib.as_array(col, shape=(max(nnz, 1),))[:nnz].copy()
    vals = np.ctypeslib.as_array(val, shape=(max(nnz, 1),))[:nnz].copy()
    for q in (rp, col, val):
        lib().cvo_oracle_free(q)
    return row_ptr, cols, vals


def radius_sets(xa, xb, tau, search=SEARCH_DENSE):
    """CSR (row_ptr, col ascending, d2) of all pairs with d2 < tau."""
    xa, xb = _f32(xa), _f32(xb)
    rp = C.POINTER(C.c_int64)()
    col = C.POINTER(C.c_int32)()
    val = C.POINTER(C.c_float)()
    rc = lib().cvo_oracle_radius_sets(_fp(xa), xa.shape[0], _fp(xb), xb.shape[0], np.float32(tau),
                                      search, C.byref(rp), C.byref(col), C.byref(val))
    if rc != 0:
        raise MemoryError("cvo_oracle_radius_sets failed")
    na = xa.shape[0]
    row_ptr = np.ctypeslib.as_array(rp, shape=(na + 1,)).copy()
    nnz = int(row_ptr[-1])
    cols = np.ctypeslib.as_array(col, shape=(max(nnz, 1),))[:nnz].copy()
    vals = np.ctypeslib.as_array(val, shape=(max(nnz, 1),))[:nnz].copy()
    for q in (rp, col, val):
        lib().cvo_oracle_free(q)
    return row_ptr, cols, vals


def flow(p, ell, x, y, csr):
    x, y = _f32(x), _f32(y)
    rp, col, val = csr
    om, v = np.zeros(3), np.zeros(3)
    sa, sad2 = C.c_double(), C.c_double()
    lib().cvo_oracle_flow(C.byref(p), np.float32(ell), _fp(x), x.shape[0], _fp(y), y.shape[0],
                          rp.ctypes.data_as(C.POINTER(C.c_int64)),
                          col.ctypes.data_as(C.POINTER(C.c_int32)), _fp(val), _dp(om), _dp(v),
                          C.byref(sa), C.byref(sad2))
    return om, v, sa.value, sad2.value


def step_coeffs(ell, omega, v, x, y, csr):
    x, y = _f32(x), _f32(y)
    omega, v = _f32(omega), _f32(v)
    rp, col, val = csr
    out = np.zeros(4)
    lib().cvo_oracle_step_coeffs(np.float32(ell), _fp(omega), _fp(v), _fp(x), x.shape[0], _fp(y),
                                 y.shape[0], rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                 col.ctypes.data_as(C.POINTER(C.c_int32)), _fp(val), _dp(out))
    return out


def pick_step(bcde, min_step=0.2):
    b = np.ascontiguousarray(bcde, dtype=np.float64)
    return float(lib().cvo_oracle_pick_step(_dp(b), np.float32(min_step)))


def exp_se3(omega, v, dt):
    omega, v = _f32(omega), _f32(v)
    dR, dT = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().cvo_oracle_exp_se3(_fp(omega), _fp(v), np.float32(dt), _fp(dR), _fp(dT))
    return dR.reshape(3, 3), dT


def dist_se3(omega, v, dt):
    omega, v = _f32(omega), _f32(v)
    return float(lib().cvo_oracle_dist_se3(_fp(omega), _fp(v), np.float32(dt)))


def function_inner_product(p, ell, xa, fa, xb, fb, search=SEARCH_GRID):
    xa, fa, xb, fb = _f32(xa), _f32(fa), _f32(xb), _f32(fb)
    return float(lib().cvo_oracle_function_inner_product(
        C.byref(p), np.float32(ell), _fp(xa), _fp(fa), xa.shape[0], _fp(xb), _fp(fb), xb.shape[0],
        search))


def trace_to_dict(t):
    return dict(k=t.k, exit_code=t.exit_code, ell=t.ell, step=t.step, dist=t.dist,
                omega=list(t.omega), v=list(t.v), omega_d=list(t.omega_d), v_d=list(t.v_d),
                bcde=list(t.bcde), sum_a=t.sum_a, dl=t.dl, nnz=t.nnz, nnz_xx=t.nnz_xx,
                nnz_yy=t.nnz_yy)


def align(p, s, x, fx, y0, fy, search=SEARCH_GRID, trace_cap=2000, shard=None, allreduce=None):
    """Runs one align() on state `s` (mutated).  Returns (iterations, [trace dicts]).

    shard = (row_lo, row_hi, srow_lo, srow_hi) + allreduce(np.ndarray float64) ->
    row-sharded variant (the callable must sum the array over ranks in place)."""
    x, fx, y0, fy = _f32(x), _f32(fx), _f32(y0), _f32(fy)
    tr = (Trace * trace_cap)()
    if shard is None:
        n_it = lib().cvo_oracle_align(C.byref(p), C.byref(s), _fp(x), _fp(fx), x.shape[0],
                                      _fp(y0), _fp(fy), y0.shape[0], search, tr, trace_cap)
    else:
        def _cb(_user, buf, count):
            arr = np.ctypeslib.as_array(buf, shape=(count,))
            allreduce(arr)
        cb = ALLREDUCE_FN(_cb) if allreduce is not None else C.cast(None, ALLREDUCE_FN)
        n_it = lib().cvo_oracle_align_sharded(
            C.byref(p), C.byref(s), _fp(x), _fp(fx), x.shape[0], _fp(y0), _fp(fy), y0.shape[0],
            search, shard[0], shard[1], shard[2], shard[3], cb, None, tr, trace_cap)
    if n_it < 0:
        raise RuntimeError("cvo_oracle_align failed")
    return n_it, [trace_to_dict(tr[i]) for i in range(min(n_it, trace_cap))]


def state_matrices(s):
    return (np.array(s.transform, np.float32).reshape(4, 4),
            np.array(s.prev_transform, np.float32).reshape(4, 4),
            np.array(s.accum_transform, np.float32).reshape(4, 4))


def ref_radius_search(xb, xa, radius_sq):
    """Reference nanoflann radius search; returns CSR (row_ptr, col, d2) or None."""
    R = ref_lib()
    if R is None:
        return None
    xa, xb = _f32(xa), _f32(xb)
    na = xa.shape[0]
    row_ptr = np.zeros(na + 1, np.int64)
    col = C.POINTER(C.c_int32)()
    d2 = C.POINTER(C.c_float)()
    rc = R.ref_radius_search(_fp(xb), xb.shape[0], _fp(xa), na, np.float32(radius_sq),
                             row_ptr.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(col),
                             C.byref(d2))
    if rc != 0:
        raise RuntimeError("ref_radius_search failed: %d" % rc)
    nnz = int(row_ptr[-1])
    cols = np.ctypeslib.as_array(col, shape=(max(nnz, 1),))[:nnz].copy()
    d2s = np.ctypeslib.as_array(d2, shape=(max(nnz, 1),))[:nnz].copy()
    R.ref_free(col)
    R.ref_free(d2)
    return row_ptr, cols, d2s
