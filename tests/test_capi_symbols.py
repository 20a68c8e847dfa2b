"""CPU: the C-ABI library loads and exports every symbol include/cvo_hip.h
declares (no device compute is attempted without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(cvo_hip_[a-z0-9_]+)\s*\(", text))
    names -= {"cvo_hip_allreduce_fn"}
    return sorted(names)


def test_header_and_binding_agree(pkg):
    assert sorted(pkg.capi.SYMBOLS) == _declared_symbols()


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg.capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(pkg.capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), "libcvo_hip.so does not export %s" % name


def test_params_and_state_defaults(pkg):
    """Constructor initialisers of the reference (cvo.cpp:18-48, adaptive_cvo.cpp:18-50)."""
    capi = pkg.capi
    p = capi.default_params(capi.MODE_CVO)
    assert (p.max_iter, p.c, p.d) == (2000, 7.0, 7.0)
    assert p.ell_init == pytest.approx(0.15) and p.sp_thres == pytest.approx(8e-3)
    assert p.c_ell == 200.0 and p.eps == pytest.approx(5e-5) and p.eps_2 == pytest.approx(1e-5)
    q = capi.default_params(capi.MODE_ACVO)
    assert q.ell_init == pytest.approx(0.1) and q.ell_min == pytest.approx(0.0391)
    assert q.sp_thres == pytest.approx(8.315e-3) and q.c_ell == 0.5 and q.dl_step == 0.3
    s = capi.init_state(p)
    assert list(s.R) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and list(s.T) == [0, 0, 0]
    assert s.ell == p.ell_init and list(s.accum_transform)[::5] == [1, 1, 1, 1]


def test_error_strings_and_argument_checks(pkg):
    capi = pkg.capi
    L = capi.lib()
    assert L.cvo_hip_error_string(0) == b"ok"
    assert b"invalid" in L.cvo_hip_error_string(-1)
    assert L.cvo_hip_default_params(7, ctypes.byref(capi.Params())) == -1
    assert capi.shard_range(10, 0, 3) == (0, 3) and capi.shard_range(10, 2, 3) == (6, 10)


def test_no_cpu_fallback(pkg):
    """Without a visible GPU, creating a context must fail loudly."""
    capi = pkg.capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.CvoHipError):
        capi.Context(mode=capi.MODE_CVO)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the product package or
    include/ may mention it."""
    bad = []
    for base in ("cvo-rgbd_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(^|[^A-Za-z_])(import oracle|from oracle|liboracle|cvo_oracle)", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_parameters_are_validated_before_anything_touches_a_device(pkg):
    """cvo_hip_create refuses parameter blocks the kernels cannot work with -- an unknown mode
    (MATLAB is a preset of default_params, not a mode of a context), kernel scales or thresholds
    that are not positive, non-finite values -- with CVO_HIP_ERR_INVALID (-1), not with the
    'no device' status a GPU-less box would otherwise give (-5): the check runs first.  (The
    reference would take log() of a non-positive quotient and carry NaNs, ref src/cvo.cpp:102-103.)"""
    capi = pkg.capi
    L = capi.lib()

    def create(p):
        ctx = ctypes.c_void_p()
        rc = L.cvo_hip_create(0, None, ctypes.byref(p), ctypes.byref(ctx))
        if rc == 0:
            L.cvo_hip_destroy(ctx)
        return rc

    good = capi.default_params(capi.MODE_CVO)
    assert create(good) in (0, -5)                      # fine: a context, or no GPU here
    for field, value in (("mode", 2), ("mode", 9), ("sigma", 0.0), ("c_sigma", -1.0), ("c", 0.0), ("d", 0.0),
                         ("c_ell", 0.0), ("sp_thres", 0.0), ("sp_thres", -1e-3), ("ell_init", 0.0),
                         ("max_iter", -1), ("eps", float("nan")), ("min_step", float("inf")),
                         ("color_scale", -1.0)):
        p = capi.default_params(capi.MODE_CVO)
        setattr(p, field, value)
        assert create(p) == -1, field
    q = capi.default_params(capi.MODE_ACVO)
    q.c_sp_thres = 0.0
    assert create(q) == -1
    q = capi.default_params(capi.MODE_ACVO)
    q.dl_step = float("nan")
    assert create(q) == -1
    m = capi.default_params(capi.MODE_MATLAB)           # the MATLAB preset is a valid CVO-mode block
    assert m.mode == capi.MODE_CVO and m.color_scale > 0 and create(m) in (0, -5)


def test_every_option_is_documented_and_the_library_reads_the_environment_in_one_place():
    """cvo_hip_set_option's keys (csrc/cvo_capi.cpp kOptions) are what a host program sets instead of environment variables: every key
    and every environment default is listed in INTEGRATION.md, and the library's translation units hold ONE getenv (the table's) besides
    the front end's own switch."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "cvo-rgbd_amd", "csrc", "cvo_capi.cpp")).read()
    table = src[src.index("const OptDef kOptions[] = {"):src.index("void env_defaults(")]
    entries = re.findall(r'\{"([a-z_0-9]+)",\s*(nullptr|"([A-Z_0-9]+)")', table)
    assert len(entries) >= 30
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for key, _, env in entries:
        assert "`%s`" % key in doc, key
        if env:
            assert env in doc or env.replace("CVO_HIP", "") in doc, env
    # every key of the table is handled by apply_option and by cvo_hip_get_option
    for key, _, _ in entries:
        assert src.count('is("%s")' % key) >= 2, key
    csrc = os.path.join(root, "cvo-rgbd_amd", "csrc")
    n = 0
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".h", ".hpp")):
            n += len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, f)).read()))
    assert n == 2, n   # cvo_capi.cpp env_defaults, cvo_frontend.hip CVO_FE_NO_GRAPH
