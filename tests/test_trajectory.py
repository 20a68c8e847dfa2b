"""trajectory writer + TUM error metrics against the outputs of the
reference's own evaluation scripts on its own fr1/desk files
(tests/golden/trajectory_eval.json, tools/make_golden_traj.py)."""
import io
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "trajectory_eval.json")))


@pytest.fixture(scope="module")
def inputs(pkg):
    z = np.load(os.path.join(GOLD, "trajectory_inputs.npz"))
    gt_rows = z["gt"]
    gt = {float(r[0]): [float(v) for v in r[1:8]] for r in gt_rows}
    est_text = str(z["est_text"])
    est = pkg.trajectory.read_trajectory(est_text, is_text=True)
    return gt, est, est_text


def test_writer_format(pkg, gold, inputs):
    """`name tx ty tz qx qy qz qw`, default ostream precision (ref cvo_main.cpp:58-65)."""
    _, est, est_text = inputs
    assert est_text.split("\n")[:3] == gold["estimate_first_lines"]
    buf = io.StringIO()
    w = pkg.trajectory.TrajectoryWriter(buf)
    M = np.eye(4)
    M[:3, 3] = [0.1234567891, -2.5e-7, 3.0]
    w.append("1305031453.359684", M)
    assert buf.getvalue() == "1305031453.359684 0.123457 -2.5e-07 3 0 0 0 1\n"
    # what is written reads back as the same pose to print precision
    th = 0.3
    M[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    buf = io.StringIO()
    pkg.trajectory.TrajectoryWriter(buf).append("7.5", M)
    back = pkg.trajectory.read_trajectory(buf.getvalue(), matrices=True, is_text=True)[7.5]
    assert np.allclose(back, M, atol=2e-6)


def test_associate_matches_reference(pkg, gold, inputs):
    gt, est, _ = inputs
    m = pkg.trajectory.associate(gt.keys(), est.keys(), 0.0, 0.02)
    assert len(m) == gold["ate"]["pairs"]
    assert [list(p) for p in m[:5]] == gold["ate"]["first_matches"]
    # one-to-one, within the window
    assert len({a for a, _ in m}) == len(m) == len({b for _, b in m})
    assert max(abs(a - b) for a, b in m) < 0.02


def test_ate_matches_reference(pkg, gold, inputs):
    gt, est, _ = inputs
    r = pkg.trajectory.absolute_trajectory_error(gt, est)
    g = gold["ate"]
    assert r["pairs"] == g["pairs"]
    for k in ("rmse", "mean", "median", "std", "min", "max"):
        assert r[k] == pytest.approx(g[k], rel=1e-9, abs=1e-12), k
    assert np.allclose(r["rotation"], g["rotation"], atol=1e-10)
    assert np.allclose(np.ravel(r["translation"]), g["translation"], atol=1e-10)


@pytest.mark.parametrize("key,unit,delta", [("s_1", "s", 1.0), ("f_5", "f", 5), ("m_0.25", "m", 0.25),
                                            ("deg_10", "deg", 10.0)])
def test_rpe_matches_reference(pkg, gold, inputs, key, unit, delta):
    gt, est, _ = inputs
    tj = pkg.trajectory
    gt_m = {s: tj.pose_matrix(v[0:3], v[3:7]) for s, v in gt.items()}
    est_m = {s: tj.pose_matrix(v[0:3], v[3:7]) for s, v in est.items()}
    rows, st = tj.relative_pose_error(gt_m, est_m, delta=delta, delta_unit=unit)
    g = gold["rpe"][key]
    assert len(rows) == g["pairs"]
    assert st["translational"]["rmse"] == pytest.approx(g["trans_rmse"], rel=1e-8)
    assert st["translational"]["mean"] == pytest.approx(g["trans_mean"], rel=1e-8)
    assert st["translational"]["median"] == pytest.approx(g["trans_median"], rel=1e-8)
    assert st["translational"]["max"] == pytest.approx(g["trans_max"], rel=1e-8)
    assert st["rotational"]["rmse"] == pytest.approx(g["rot_rmse"], rel=1e-7)
    assert st["rotational"]["mean"] == pytest.approx(g["rot_mean"], rel=1e-7)
    assert st["rotational"]["max"] == pytest.approx(g["rot_max"], rel=1e-7)
    for mine, ref in zip(rows[:3], g["first_rows"]):
        assert mine[:4] == ref[:4]
        assert mine[4] == pytest.approx(ref[4], rel=1e-8, abs=1e-12)
        assert mine[5] == pytest.approx(ref[5], rel=1e-6, abs=1e-9)


def test_accumulate_is_right_multiplication(pkg):
    """accum_transform = accum_transform * transform (ref src/cvo.cpp:413-415)."""
    rng = np.random.default_rng(3)
    Ts = []
    for _ in range(4):
        A = np.eye(4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        A[:3, :3] = q * np.sign(np.linalg.det(q))
        A[:3, 3] = rng.normal(size=3)
        Ts.append(A)
    acc = pkg.trajectory.accumulate(Ts)
    assert np.allclose(acc[-1], Ts[0] @ Ts[1] @ Ts[2] @ Ts[3])
    assert np.allclose(acc[1], Ts[0] @ Ts[1])
