#!/usr/bin/env python3
"""Golden vectors for cvo-rgbd_amd/trajectory.py, made by RUNNING the reference's
own TUM evaluation scripts (Python 2 sources, converted in memory with lib2to3 --
nothing of them is written to disk) on the reference's own fr1/desk files:

  ground truth : data/rgbd_dataset/freiburg1_desk/groundtruth.txt (mocap)
  estimate     : the 572 OpenCV RGB-D odometry increments of cv_rgbd_poses.csv,
                 chained from the identity and stamped with the RGB times of
                 assoc.txt (what the reference's MATLAB plots compare CVO with)

Writes tests/golden/trajectory_eval.json (the expected numbers) and
tests/golden/trajectory_inputs.npz (the two trajectories, so that the test does
not need /root/reference).  Run in the build container only."""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/data/rgbd_dataset"
TOOLS = os.path.join(REF, "rgbd_benchmark_tools")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_py2(name):
    from lib2to3 import refactor
    src = open(os.path.join(TOOLS, name + ".py")).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    code = str(tool.refactor_string(src + "\n", name))
    code = code.replace("numpy.linalg.linalg.svd", "numpy.linalg.svd")   # numpy 2 dropped the alias
    mod = types.ModuleType(name)
    mod.__dict__["__name__"] = name          # not "__main__": the CLI part must not run
    sys.modules[name] = mod
    exec(compile(code, name + ".py(2to3)", "exec"), mod.__dict__)
    return mod


def main():
    sys.path.insert(0, TOOLS)
    associate = load_py2("associate")
    ate = load_py2("evaluate_ate")
    rpe = load_py2("evaluate_rpe")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()

    desk = os.path.join(REF, "freiburg1_desk")
    stamps = [float(l.split()[0]) for l in open(os.path.join(desk, "assoc.txt")) if l.strip()]
    inc = np.loadtxt(os.path.join(desk, "cv_rgbd_poses.csv"), delimiter=",", skiprows=1)
    rel = []
    for row in inc:
        M = np.eye(4)
        M[:3, 3] = row[2:5]
        M[:3, :3] = row[5:14].reshape(3, 3)
        rel.append(np.linalg.inv(M))   # listed frame1 -> frame2; chained camera poses need the inverse
    poses = [np.eye(4)] + pkg.trajectory.accumulate(rel)
    poses = poses[:len(stamps)]
    est_lines = [pkg.data.pose_line("%.6f" % s, P) for s, P in zip(stamps, poses)]
    est_text = "\n".join(est_lines) + "\n"
    gt_text = open(os.path.join(desk, "groundtruth.txt")).read()

    tmp = "/tmp/_traj_golden"
    os.makedirs(tmp, exist_ok=True)
    open(os.path.join(tmp, "est.txt"), "w").write(est_text)
    open(os.path.join(tmp, "gt.txt"), "w").write(gt_text)

    # --- reference ATE (evaluate_ate.py __main__ body, verbatim call sequence)
    first = associate.read_file_list(os.path.join(tmp, "gt.txt"))
    second = associate.read_file_list(os.path.join(tmp, "est.txt"))
    matches = associate.associate(first, second, 0.0, 0.02)
    fx = np.matrix([[float(v) for v in first[a][0:3]] for a, b in matches]).transpose()
    sx = np.matrix([[float(v) for v in second[b][0:3]] for a, b in matches]).transpose()
    rot, trans, terr = ate.align(sx, fx)
    gold = {"ate": {"pairs": len(terr), "rmse": float(np.sqrt(np.dot(terr, terr) / len(terr))),
                    "mean": float(np.mean(terr)), "median": float(np.median(terr)),
                    "std": float(np.std(terr)), "min": float(np.min(terr)), "max": float(np.max(terr)),
                    "rotation": np.asarray(rot).tolist(), "translation": np.asarray(trans).ravel().tolist(),
                    "first_matches": [[a, b] for a, b in matches[:5]]}}
    # --- reference RPE, fixed delta (deterministic)
    tg = rpe.read_trajectory(os.path.join(tmp, "gt.txt"))
    te = rpe.read_trajectory(os.path.join(tmp, "est.txt"))
    gold["rpe"] = {}
    for unit, delta in (("s", 1.0), ("f", 5), ("m", 0.25), ("deg", 10.0)):
        res = rpe.evaluate_trajectory(tg, te, 0, True, delta, unit, 0.0, 1.0)
        tr = np.array(res)[:, 4]
        ro = np.array(res)[:, 5]
        gold["rpe"]["%s_%g" % (unit, delta)] = {
            "pairs": len(res),
            "trans_rmse": float(np.sqrt(np.dot(tr, tr) / len(tr))), "trans_mean": float(np.mean(tr)),
            "trans_median": float(np.median(tr)), "trans_max": float(np.max(tr)),
            "rot_rmse": float(np.sqrt(np.dot(ro, ro) / len(ro))), "rot_mean": float(np.mean(ro)),
            "rot_max": float(np.max(ro)), "first_rows": [list(map(float, r)) for r in res[:3]]}
    gold["estimate_first_lines"] = est_lines[:3]
    out = os.path.join(ROOT, "tests", "golden")
    json.dump(gold, open(os.path.join(out, "trajectory_eval.json"), "w"), indent=1)
    gt_rows = np.array([[k] + [float(v) for v in first[k][0:7]] for k in sorted(first)])
    np.savez_compressed(os.path.join(out, "trajectory_inputs.npz"), gt=gt_rows,
                        est_text=np.array(est_text))
    print("ATE rmse %.6f m over %d pairs; RPE(1 s) trans rmse %.6f m" % (
        gold["ate"]["rmse"], gold["ate"]["pairs"], gold["rpe"]["s_1"]["trans_rmse"]))


if __name__ == "__main__":
    main()
