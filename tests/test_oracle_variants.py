"""CPU: the deviation study -- how far the registration moves when the oracle's arithmetic is
replaced by ANOTHER admissible reading of the reference's source.

The oracle fixes one reading where the reference's compiled arithmetic is the compiler's choice
(oracle/cvo_oracle.h CVO_ORACLE_VAR_*): the flow sums of cvo.cpp:197-198 are a FLOAT per-row
product there (accumulation order = Eigen's vectorised redux, per build), the oracle and the HIP
path add each pair in float64; nanoflann's squared distance contracts to FMAs or not by flag.

Two statements, both needed to read BASELINE's "<= 1e-4 vs reference" correctly:
  * ONE iteration from the same pose: every variant's twist agrees with the contract to ~1e-6
    relative and nnz(A) to a handful of threshold members -- far inside 1e-4;
  * a WHOLE registration: the stopping rule ||log(dT)|| < eps_2 on a slowly converging ascent
    turns those last-bit differences into +-10 iterations and 1e-3 .. 5e-2 relative pose
    spread.  A 1e-4 agreement of whole registrations therefore exists only between builds with
    bit-identical arithmetic (HIP vs the oracle: tests/test_gpu_parity.py, 1e-6); it does not
    exist between two compilations of the reference itself.  The spread stays inside the
    registration's own accuracy against the synthetic ground truth.
ref src/cvo.cpp:164-212 (compute_flow), :361-420 (align), thirdparty/nanoflann.hpp:403-406.
"""
import numpy as np
import pytest

VARIANTS = {"rowsum_seq": 1, "rowsum_packet": 2, "d2_plain": 4, "packet+d2_plain": 6}


@pytest.fixture()
def variant(po):
    yield po.set_variant
    po.set_variant(0)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("mode_name", ["cvo", "acvo"])
def test_one_iteration_is_insensitive(pkg, po, variant, mode_name):
    acvo = mode_name == "acvo"
    mode = po.MODE_ACVO if acvo else po.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(2500, 2500, seed=77, acvo=acvo)
    p = po.default_params(mode)
    worst = 0.0
    for ell in (p.ell_init, 0.06):
        variant(0)
        csr = po.se_kernel(p, ell, xf, ff, xm, fm, search=po.SEARCH_GRID)
        om, v, sa, _ = po.flow(p, ell, xf, xm, csr)
        for name, flags in VARIANTS.items():
            variant(flags)
            csr_v = po.se_kernel(p, ell, xf, ff, xm, fm, search=po.SEARCH_GRID)
            om_v, v_v, sa_v, _ = po.flow(p, ell, xf, xm, csr_v)
            assert abs(int(csr_v[0][-1]) - int(csr[0][-1])) <= 8, name   # threshold members only
            assert abs(sa_v - sa) <= 1e-6 * sa, name
            worst = max(worst, _rel(om_v, om), _rel(v_v, v))
    assert worst <= 2e-5            # observed ~1e-6; north_star's tolerance is 1e-4


@pytest.mark.parametrize("mode_name,seed", [("cvo", 500), ("cvo", 501), ("acvo", 500), ("acvo", 501)])
def test_whole_registration_spread(pkg, po, variant, mode_name, seed):
    """Records the scale; the bound is the registration's own error against ground truth."""
    acvo = mode_name == "acvo"
    mode = po.MODE_ACVO if acvo else po.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(2000, 2000, seed=seed, acvo=acvo)
    T_gt = pkg.data.gt_motion()

    def run(flags):
        variant(flags)
        p = po.default_params(mode)
        st = po.init_state(p)
        n, _ = po.align(p, st, xf, ff, xm, fm, search=po.SEARCH_GRID, trace_cap=1)
        return n, po.state_matrices(st)[0]

    n0, T0 = run(0)
    spread = []
    for name, flags in VARIANTS.items():
        n, T = run(flags)
        rot, tra = pkg.data.rel_pose_error(T, T0)
        spread.append((name, n - n0, rot, tra))
        assert abs(n - n0) <= 25 and rot <= 0.05 and tra <= 0.1, spread
    # the point of the study: at least one admissible variant leaves the 1e-4 band
    assert max(max(s[2], s[3]) for s in spread) > 1e-4, spread
    g_rot, g_tra = pkg.data.rel_pose_error(T0, T_gt)
    print("variants vs contract:", spread, "contract vs ground truth:", (g_rot, g_tra))
    assert max(s[2] for s in spread) <= g_rot and max(s[3] for s in spread) <= g_tra, (spread, g_rot, g_tra)
