#!/usr/bin/env python3
"""Generates tests/golden/* from the reference tree (run in the build container
only: needs /root/reference; the outputs are DATA and are committed).

  desk_pcd_ds.npz        the five shipped fr1/desk clouds (xyz float32, rgb uint8),
                         ref data/rgbd_dataset/freiburg1_desk/pcd_ds/*.pcd
  nanoflann_sets.json    neighbour-set digests produced by the REFERENCE'S OWN
                         nanoflann header (oracle/_ref) for se_kernel's radii
  matlab_transforms.json the MATLAB implementation's recorded results for pairs
                         0-1 .. 3-4, decoded from
                         freiburg1_desk_07-May-2019-02-35-00.mat, plus the mocap
                         relative poses (groundtruth.txt) -- soft references
  oracle_traces.json     per-iteration traces of the oracle itself (regression
                         pins; NOT reference-derived)
"""
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"
DESK = os.path.join(REF, "data/rgbd_dataset/freiburg1_desk")
OUT = os.path.join(ROOT, "tests", "golden")
STAMPS = ["1305031453.359684", "1305031453.391690", "1305031453.423683",
          "1305031453.459685", "1305031453.491698"]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def clouds():
    pkg = ge.load_package()
    out = {}
    for k, s in enumerate(STAMPS):
        xyz, rgb = pkg.data.read_pcd_ascii(os.path.join(DESK, "pcd_ds", s + ".pcd"))
        out["xyz%d" % k] = xyz
        out["rgb%d" % k] = rgb
    out["stamps"] = np.array(STAMPS)
    np.savez_compressed(os.path.join(OUT, "desk_pcd_ds.npz"), **out)
    return out


def nanoflann_sets(cl):
    """Digest of the reference kd-tree radius search: rows = cloud 0 (fixed),
    columns = cloud 1 (moving, identity transform)."""
    if po.ref_lib() is None:
        raise SystemExit("oracle/_ref not built (make -C oracle ref)")
    recs = []
    pc = po.default_params(po.MODE_CVO)
    pa = po.default_params(po.MODE_ACVO)
    cases = [("cvo", pc, e) for e in (0.15, 0.10, 0.06, 0.03)] + \
            [("acvo", pa, e) for e in (0.10, 0.0391)]
    for stride in (5, 1):
        xa, xb = cl["xyz0"][::stride], cl["xyz1"][::stride]
        for name, p, ell in cases:
            tau, _ = po.thresholds(p, ell)
            rp, col, d2 = po.ref_radius_search(xb, xa, tau)
            rows = np.repeat(np.arange(xa.shape[0]), np.diff(rp))
            o = np.lexsort((col, rows))
            recs.append(dict(mode=name, ell=ell, stride=stride, tau_bits=int(np.float32(tau).view(np.uint32)),
                             nnz=int(rp[-1]),
                             sha256_rowptr_col=digest(rp.astype(np.int64), col[o].astype(np.int32)),
                             sha256_d2=digest(d2[o].astype(np.float32))))
            print(recs[-1])
    with open(os.path.join(OUT, "nanoflann_sets.json"), "w") as fh:
        json.dump(dict(note="rows: cloud 0, columns: cloud 1, every stride-th point; produced by "
                            "the reference's vendored nanoflann (oracle/ref_nanoflann_harness.cpp)",
                       cases=recs), fh, indent=1)


def matlab_and_mocap():
    from scipy.io import loadmat
    from scipy.io.matlab._mio5 import MatFile5Reader
    m = loadmat(os.path.join(DESK, "freiburg1_desk_07-May-2019-02-35-00.mat"))
    fw = m["__function_workspace__"].tobytes()
    hdr = (b"MATLAB 5.0 MAT-file".ljust(116, b" ") + b"\x00" * 8 + b"\x00\x01IM")
    rdr = MatFile5Reader(io.BytesIO(hdr + fw[8:]))
    rdr.initialize_read()
    rdr.mat_stream.seek(128)
    h, _ = rdr.read_var_header()
    res = rdr.read_var_array(h, process=False)
    arr = res[0, 0]["MCOS"][0]["arr"]
    mats = []
    for k in range(0, 5):
        T = np.array(arr[2 + k, 0][0, 0]["TransformationMatrix"], np.float64)
        mats.append(T.T.tolist())   # H = T' (rgbddataset_cdf_plots.m:79)
    reg_time = np.asarray(m["registration_time"]).ravel()[:5].tolist() if "registration_time" in m else None

    # mocap relative poses: nearest-later sample to each RGB stamp
    gt = np.loadtxt(os.path.join(DESK, "groundtruth.txt"))
    def pose_at(ts):
        i = int(np.searchsorted(gt[:, 0], ts, side="left"))
        i = min(i, gt.shape[0] - 1)
        t = gt[i, 1:4]
        qx, qy, qz, qw = gt[i, 4:8]
        Rm = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                       [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                       [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        M = np.eye(4)
        M[:3, :3] = Rm
        M[:3, 3] = t
        return M
    rel = []
    for k in range(1, 5):
        T0, T1 = pose_at(float(STAMPS[k - 1])), pose_at(float(STAMPS[k]))
        rel.append((np.linalg.inv(T0) @ T1).tolist())
    with open(os.path.join(OUT, "matlab_transforms.json"), "w") as fh:
        json.dump(dict(note="matlab[k] = result{k+1}.T' of the reference's MATLAB run (k=0 identity); "
                            "mocap_rel[k-1] = inv(T_{k-1}) T_k from groundtruth.txt; soft references only "
                            "(MATLAB used grid-averaged ~700-point clouds and a linear colour kernel)",
                       matlab=mats, registration_time=reg_time, mocap_rel=rel), fh, indent=1)


def oracle_traces(cl):
    pkg = ge.load_package()
    out = {}
    # (1) TUM pair 0->1, every 5th point, cvo
    x, fx = cl["xyz0"][::5], pkg.data.cvo_features(cl["rgb0"][::5])
    y, fy = cl["xyz1"][::5], pkg.data.cvo_features(cl["rgb1"][::5])
    p = po.default_params(po.MODE_CVO)
    s = po.init_state(p)
    n, tr = po.align(p, s, x, fx, y, fy)
    out["tum01_s5_cvo"] = dict(n_iter=n, transform=po.state_matrices(s)[0].tolist(), trace=tr)
    # (2) synthetic 2k x 2k, cvo and acvo
    for mode, name in ((po.MODE_CVO, "syn2k_cvo"), (po.MODE_ACVO, "syn2k_acvo")):
        xf, ff, xm, fm = pkg.data.synthetic_pair(2000, 2000, seed=7, acvo=(mode == po.MODE_ACVO))
        p = po.default_params(mode)
        s = po.init_state(p)
        n, tr = po.align(p, s, xf, ff, xm, fm)
        out[name] = dict(n_iter=n, transform=po.state_matrices(s)[0].tolist(), trace=tr)
        print(name, n)
    with open(os.path.join(OUT, "oracle_traces.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cl = clouds()
    if "--skip-sets" not in sys.argv:
        nanoflann_sets(cl)
    matlab_and_mocap()
    oracle_traces(cl)
    print("golden fixtures written to", OUT)
