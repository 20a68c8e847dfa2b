"""CPU: the oracle against its committed per-iteration traces (regression
pins) and, softly, against the only recorded results of the reference tree:
the MATLAB implementation's transforms and the mocap ground truth."""
import numpy as np
import pytest


def _rel_angle(pkg, A, B):
    return pkg.data.rel_pose_error(A, B)


def test_trace_tum_pair_cvo(pkg, po, desk, golden_json):
    g = golden_json("oracle_traces.json")["tum01_s5_cvo"]
    x, fx = desk["xyz0"][::5], pkg.data.cvo_features(desk["rgb0"][::5])
    y, fy = desk["xyz1"][::5], pkg.data.cvo_features(desk["rgb1"][::5])
    p = po.default_params(po.MODE_CVO)
    s = po.init_state(p)
    n, tr = po.align(p, s, x, fx, y, fy, search=po.SEARCH_GRID)
    assert n == g["n_iter"]
    assert np.array_equal(po.state_matrices(s)[0], np.array(g["transform"], np.float32))
    for a, b in zip(tr, g["trace"]):
        assert a["nnz"] == b["nnz"] and a["omega"] == b["omega"] and a["step"] == b["step"]
    # the reference's length-scale schedule (cvo.cpp:408-410)
    ells = [round(t["ell"], 4) for t in tr]
    assert ells[:4] == [0.15] * 4 and ells[4:11] == [0.1] * 7 and ells[11:21] == [0.06] * 10
    assert all(e == 0.03 for e in ells[21:])


@pytest.mark.parametrize("name,mode", [("syn2k_cvo", 0), ("syn2k_acvo", 1)])
def test_trace_synthetic(pkg, po, golden_json, name, mode):
    g = golden_json("oracle_traces.json")[name]
    xf, ff, xm, fm = pkg.data.synthetic_pair(2000, 2000, seed=7, acvo=(mode == 1))
    p = po.default_params(mode)
    s = po.init_state(p)
    n, tr = po.align(p, s, xf, ff, xm, fm, search=po.SEARCH_GRID)
    assert n == g["n_iter"]
    assert np.array_equal(po.state_matrices(s)[0], np.array(g["transform"], np.float32))
    if mode == 1:   # acvo: ell follows dl, stays inside [ell_min, 0.15)
        ells = np.array([t["ell"] for t in tr])
        assert ells[0] == np.float32(0.1) and ells.min() >= np.float32(0.0391) and ells.max() < 0.15
        assert all(t["nnz_xx"] > 0 and t["nnz_yy"] > 0 for t in tr)
    # the registration recovers the synthetic motion to a few per cent
    rot, tra = _rel_angle(pkg, np.linalg.inv(po.state_matrices(s)[0].astype(np.float64)),
                          np.linalg.inv(pkg.data.gt_motion()))
    assert rot < 0.3 and (tra < 0.3 or mode == 1)   # 2k-point acvo recovers the 1 cm shift only coarsely


def test_soft_agreement_with_matlab_and_mocap(pkg, po, desk, golden_json):
    """Sanity only: the MATLAB run used ~700-point grid-averaged clouds and a
    linear colour kernel; mocap is an independent measurement.  All three must
    describe the same ~25 mrad / ~1 cm inter-frame motion."""
    g = golden_json("matlab_transforms.json")
    p = po.default_params(po.MODE_CVO)
    s = po.init_state(p)
    x, fx = desk["xyz0"][::5], pkg.data.cvo_features(desk["rgb0"][::5])
    y, fy = desk["xyz1"][::5], pkg.data.cvo_features(desk["rgb1"][::5])
    po.align(p, s, x, fx, y, fy, search=po.SEARCH_GRID)
    T = po.state_matrices(s)[0].astype(np.float64)
    for ref in (np.array(g["matlab"][1]), np.array(g["mocap_rel"][0])):
        dR = T[:3, :3] @ ref[:3, :3].T
        ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
        assert ang < 0.012                      # < 12 mrad apart (motion ~25 mrad)
        assert np.linalg.norm(T[:3, 3] - ref[:3, 3]) < 0.006   # < 6 mm (motion ~10 mm)


def test_state_carry_over_between_frames(pkg, po, desk):
    """cvo never resets ell / R / T between frames: the
    second pair starts at the ell the first one ended on and warm-starts R,T."""
    p = po.default_params(po.MODE_CVO)
    s = po.init_state(p)
    clouds = [(desk["xyz%d" % k][::8], pkg.data.cvo_features(desk["rgb%d" % k][::8])) for k in range(3)]
    n1, tr1 = po.align(p, s, *clouds[0], *clouds[1], search=po.SEARCH_GRID)
    ell_end, R_end = s.ell, np.array(s.R)
    acc1 = po.state_matrices(s)[2].copy()
    n2, tr2 = po.align(p, s, *clouds[1], *clouds[2], search=po.SEARCH_GRID)
    assert tr2[0]["ell"] == ell_end == np.float32(0.03)
    assert tr2[4]["ell"] == np.float32(0.1)      # the schedule pushes it back up at k = 3
    assert not np.array_equal(R_end, np.eye(3).ravel())
    acc2 = po.state_matrices(s)[2]
    # accum_transform lags: it multiplies the transform of the TOP of the last iteration
    assert not np.array_equal(acc1, acc2)


def test_empty_gram_matrix(pkg, po):
    xf, ff, xm, fm = pkg.data.synthetic_pair(300, 200, seed=3)
    p = po.default_params(po.MODE_CVO)
    s = po.init_state(p)
    n, tr = po.align(p, s, xf, ff, xm + np.float32(10), fm, search=po.SEARCH_GRID)
    assert n == 1 and tr[0]["exit_code"] == 1 and tr[0]["nnz"] == 0
    assert tr[0]["step"] == np.float32(0.2)
    assert np.array_equal(po.state_matrices(s)[0], np.eye(4, dtype=np.float32))


def test_acvo_ayy_row_rule(pkg, po):
    """Ayy contributes to dl only through rows i >= num_fixed (acvo.cpp:213-265)."""
    xf, ff, xm, fm = pkg.data.synthetic_pair(500, 800, seed=9, acvo=True)
    p = po.default_params(po.MODE_ACVO)
    p.max_iter = 1
    s = po.init_state(p)
    _, tr_a = po.align(p, s, xf, ff, xm, fm, search=po.SEARCH_GRID)
    s = po.init_state(p)
    _, tr_b = po.align(p, s, xf, ff, xm[:500], fm[:500], search=po.SEARCH_GRID)   # M == N: no tail rows
    assert tr_a[0]["nnz_yy"] > tr_b[0]["nnz_yy"] > 0
    y = xm[:500]
    A = po.se_kernel(p, 0.1, xf, ff, y, fm[:500], search=po.SEARCH_GRID)
    Axx = po.se_kernel(p, 0.1, xf, ff, xf, ff, search=po.SEARCH_GRID)
    om, v, sa, sad2 = po.flow(p, 0.1, xf, y, A)
    inv_l3 = np.float32(1) / (np.float32(0.1) ** 3)
    rows = np.repeat(np.arange(500), np.diff(Axx[0]))
    d = (xf[rows] - xf[Axx[1]]).astype(np.float64)
    sxx = float(((inv_l3 * Axx[2]).astype(np.float64) * (d * d).sum(1)).sum())
    dl_expected = (0.0 - 2.0 * sad2 + sxx) / (Axx[0][-1] + tr_b[0]["nnz_yy"] - 2 * A[0][-1])
    assert tr_b[0]["dl"] == pytest.approx(dl_expected, rel=1e-5)
