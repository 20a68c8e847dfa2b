"""Trajectory side of the registration path: the pose file the
reference's drivers write, and the two TUM error metrics it is judged with.

* `TrajectoryWriter` -- `name tx ty tz qx qy qz qw` per registered frame from
  `accum_transform`, default ostream precision (ref cpp/rkhs_registration/src/
  cvo_main.cpp:28-29,58-65; adaptive_cvo_main.cpp writes `acvo_poses_qt.txt`
  the same way).
* `read_trajectory`, `associate` -- the TUM text format and timestamp matching
  (ref data/rgbd_dataset/rgbd_benchmark_tools/associate.py:50-104,
  evaluate_rpe.py:82-108).
* `absolute_trajectory_error` -- Horn alignment + translational residuals
  (ref evaluate_ate.py:47-80,131-160).
* `relative_pose_error` -- drift over a fixed interval
  (ref evaluate_rpe.py:110-297).

Restated from the published definitions, float64 numpy; pinned against the
outputs of the reference's own scripts on its own fr1/desk files
(tests/golden/trajectory_eval.json, made by tools/make_golden_traj.py).
Nothing here runs on the GPU: these are the callers' formats either side of the
hot path.
"""
import bisect
import math

import numpy as np

from . import data as _data


class TrajectoryWriter:
    """Pose lines as the reference's drivers write them (ref cvo_main.cpp:58-65):
    one line per frame (the first one carries the identity), the frame's name (its
    RGB time stamp) followed by translation and unit quaternion (x y z w) of
    `accum_transform`, `%g`-formatted like a default std::ostream."""

    def __init__(self, target):
        self._own = isinstance(target, str)
        self._fh = open(target, "w") if self._own else target
        self.lines = 0

    def append(self, name, accum_transform):
        self._fh.write(_data.pose_line(name, accum_transform) + "\n")
        self.lines += 1

    def close(self):
        if self._own:
            self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _parse_rows(text):
    rows = []
    for line in text.replace(",", " ").replace("\t", " ").split("\n"):
        if not line or line[0] == "#":
            continue
        vals = [v for v in line.split(" ") if v.strip() != ""]
        if len(vals) > 1:
            rows.append(vals)
    return rows


def pose_matrix(t, q):
    """4x4 pose of a translation and a quaternion (x, y, z, w); a zero
    quaternion gives a pure translation (ref evaluate_rpe.py:44-72)."""
    q = np.array(q, np.float64)
    M = np.eye(4)
    M[:3, 3] = t
    nq = float(np.dot(q, q))
    if nq < np.finfo(float).eps * 4.0:
        return M
    q = q * math.sqrt(2.0 / nq)
    o = np.outer(q, q)
    M[:3, :3] = [[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                 [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                 [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]]
    return M


def read_trajectory(path_or_text, matrices=False, is_text=False):
    """{stamp: [tx ty tz qx qy qz qw]} (or 4x4 matrices) from TUM trajectory
    text.  Lines with an all-zero quaternion or a NaN are dropped as the
    reference does (ref evaluate_rpe.py:91-103)."""
    text = path_or_text if is_text else open(path_or_text).read()
    out = {}
    for vals in _parse_rows(text):
        row = [float(v) for v in vals]
        if len(row) < 8:
            continue
        if row[4:8] == [0.0, 0.0, 0.0, 0.0] or any(math.isnan(v) for v in row):
            continue
        out[row[0]] = pose_matrix(row[1:4], row[4:8]) if matrices else row[1:8]
    return out


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """Greedy one-to-one matching of two stamp sets: candidate pairs closer than
    `max_difference` (after adding `offset` to the second), best first
    (ref associate.py:71-104).  Returns sorted (first, second) pairs."""
    first = set(first_stamps)
    second = set(second_stamps)
    second_sorted = sorted(second)
    cand = []
    for a in first:
        lo = bisect.bisect_left(second_sorted, a - offset - max_difference)
        hi = bisect.bisect_right(second_sorted, a - offset + max_difference)
        for b in second_sorted[lo:hi]:
            d = abs(a - (b + offset))
            if d < max_difference:
                cand.append((d, a, b))
    cand.sort()
    matches = []
    for _, a, b in cand:
        if a in first and b in second:
            first.remove(a)
            second.remove(b)
            matches.append((a, b))
    matches.sort()
    return matches


def horn_align(model, data):
    """Rigid alignment of two 3xN point sets by Horn's closed form
    (ref evaluate_ate.py:47-80): returns R, t with data ~ R model + t and the
    per-point residual norms."""
    model = np.asarray(model, np.float64)
    data = np.asarray(data, np.float64)
    mm = model.mean(1, keepdims=True)
    dm = data.mean(1, keepdims=True)
    W = (model - mm) @ (data - dm).T
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vh
    t = dm - R @ mm
    resid = R @ model + t - data
    return R, t, np.sqrt((resid * resid).sum(0))


def _stats(err):
    err = np.asarray(err, np.float64)
    return {"pairs": int(err.size), "rmse": float(np.sqrt(np.dot(err, err) / err.size)),
            "mean": float(err.mean()), "median": float(np.median(err)), "std": float(err.std()),
            "min": float(err.min()), "max": float(err.max())}


def absolute_trajectory_error(gt, est, offset=0.0, scale=1.0, max_difference=0.02):
    """ATE as the TUM tool reports it (ref evaluate_ate.py:131-160): associate,
    align the estimate onto the ground truth, statistics of |residual|.
    `gt`, `est`: {stamp: [tx ty tz ...]}."""
    matches = associate(gt.keys(), est.keys(), float(offset), float(max_difference))
    if len(matches) < 2:
        raise ValueError("no matching timestamp pairs between the two trajectories")
    g = np.array([gt[a][0:3] for a, _ in matches], np.float64).T
    e = np.array([est[b][0:3] for _, b in matches], np.float64).T * float(scale)
    R, t, err = horn_align(e, g)
    out = _stats(err)
    out["rotation"], out["translation"] = R, t
    return out


def _closest(sorted_vals, t):
    """Index of the value closest to t, with the tie / search order of the
    reference's bisection (ref evaluate_rpe.py:110-136)."""
    lo, hi = 0, len(sorted_vals)
    best, diff = 0, abs(sorted_vals[0] - t)
    while lo < hi:
        mid = (lo + hi) // 2
        d = abs(sorted_vals[mid] - t)
        if d < diff:
            diff, best = d, mid
        if t == sorted_vals[mid]:
            return mid
        if sorted_vals[mid] > t:
            hi = mid
        else:
            lo = mid + 1
    return best


def _angle(M):
    return math.acos(min(1.0, max(-1.0, (np.trace(M[:3, :3]) - 1.0) / 2.0)))


def _rel(a, b):
    return np.linalg.inv(a) @ b


def relative_pose_error(gt, est, delta=1.0, delta_unit="s", offset=0.0, scale=1.0, max_pairs=0,
                        seed=0):
    """RPE over a fixed interval (the TUM tool's `--fixed_delta`,
    ref evaluate_rpe.py:204-297).  `gt`, `est`: {stamp: 4x4 pose}.  delta_unit:
    "s" seconds, "m" metres travelled, "rad" / "deg" rotation travelled, "f"
    frames.  max_pairs > 0 subsamples the pairs (numpy Generator(seed): the
    reference uses Python's global `random`).  Returns the list of
    [stamp_est_0, stamp_est_1, stamp_gt_0, stamp_gt_1, trans_err, rot_err] and
    the statistics of both errors."""
    s_gt = sorted(gt.keys())
    s_est = sorted(est.keys())
    n = len(s_est)
    if delta_unit == "s":
        index = s_est
    elif delta_unit in ("m", "rad", "deg"):
        k = {"m": None, "rad": 1.0, "deg": 180.0 / math.pi}[delta_unit]
        index, acc = [0.0], 0.0
        for i in range(n - 1):
            # (the reference measures motion from pose i+1 back to pose i; norms agree)
            M = _rel(est[s_est[i + 1]], est[s_est[i]])
            acc += float(np.linalg.norm(M[:3, 3])) if k is None else _angle(M) * k
            index.append(acc)
    elif delta_unit == "f":
        index = list(range(n))
    else:
        raise ValueError("unknown unit for delta: %r" % delta_unit)
    pairs = []
    for i in range(n):
        j = _closest(index, index[i] + delta)
        if j != n - 1:
            pairs.append((i, j))
    if max_pairs and len(pairs) > max_pairs:
        pick = np.random.default_rng(seed).choice(len(pairs), size=max_pairs, replace=False)
        pairs = [pairs[i] for i in sorted(pick)]
    gap = 2.0 * float(np.median(np.diff(s_gt)))
    rows = []
    for i, j in pairs:
        e0, e1 = s_est[i], s_est[j]
        g0 = s_gt[_closest(s_gt, e0 + offset)]
        g1 = s_gt[_closest(s_gt, e1 + offset)]
        if abs(g0 - (e0 + offset)) > gap or abs(g1 - (e1 + offset)) > gap:
            continue
        d_est = _rel(est[e1], est[e0]).copy()
        d_est[:3, 3] *= scale
        err = _rel(d_est, _rel(gt[g1], gt[g0]))
        rows.append([e0, e1, g0, g1, float(np.linalg.norm(err[:3, 3])), _angle(err)])
    if len(rows) < 2:
        raise ValueError("no matching timestamp pairs between the two trajectories")
    return rows, {"translational": _stats([r[4] for r in rows]),
                  "rotational": _stats([r[5] for r in rows])}


def accumulate(relative_transforms, first=None):
    """Chain per-pair transforms into poses the way the reference's object does
    (`accum_transform = accum_transform * transform`, ref src/cvo.cpp:413-415):
    returns the list of 4x4 poses after each pair."""
    acc = np.eye(4) if first is None else np.array(first, np.float64)
    out = []
    for T in relative_transforms:
        acc = acc @ np.asarray(T, np.float64)
        out.append(acc.copy())
    return out
