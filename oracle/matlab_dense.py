"""oracle/matlab_dense.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

numpy float64 restatement of the reference's MATLAB registration object
(ref matlab/@rkhs_se3_registration/rkhs_se3_registration.m) as driven by
ref data/rgbd_dataset/rgbddataset_rkhs.m:30-75: dense N x M squared-exponential
kernel thresholded at sp_threshold (:108-110), LINEAR colour inner product
CI = 1e-5 * Cx * Cz' (:36-40), the same flow / 4th-order step size / SE(3)
integration as cvo.cpp, ell schedule k > 3, 10, 20 (:238-246).

Why it is here: the only recorded outputs of the reference in the tree are the
transforms of this MATLAB run (freiburg1_desk_07-May-2019-02-35-00.mat, decoded in
tests/golden/matlab_transforms.json).  PARITY STATUS: soft.  The run's inputs are
the shipped pcd_ds clouds after pcRangeFilter + pcdownsample('gridAverage', 0.05);
MATLAB's voxel binning is not documented and the ~700-point result moves by
2-4e-3 with the binning convention, so this restatement reproduces the recorded
transforms to 2.5e-3 .. 4.5e-3 (max abs entry), not to the 1e-4 a pin would need.
tools/search_grid_anchor.py tried 130 anchorings of the grid (cloud minimum, origin,
cell centres, float32 / float64 indices, a 5 x 5 x 5 lattice of offsets): the best
one's worst pair is 2.8e-3 away (tests/golden/grid_anchor_residuals.json).
"""
import numpy as np
from scipy.linalg import logm


def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def align(fixed_xyz, fixed_rgb, moving_xyz, moving_rgb, max_iter=2000):
    """Returns (tform 4x4 = [R' -R'T; 0 1] as obj.tform.T', iterations)."""
    ell, sigma, sp, c, d = 0.15, 0.1, 1e-3, 7.0, 7.0
    eps, eps2, min_step = 5e-4, 1e-4, 0.2
    R, T = np.eye(3), np.zeros(3)
    CI = 1e-5 * fixed_rgb.astype(np.float64) @ moving_rgb.astype(np.float64).T
    X = fixed_xyz.astype(np.float64)
    Y0 = moving_xyz.astype(np.float64)
    k = 0
    for k in range(1, max_iter + 1):
        Y = Y0 @ R - (R.T @ T)            # pctransform with tf_inv(R, T)
        D2 = (X ** 2).sum(1)[:, None] + (Y ** 2).sum(1)[None, :] - 2.0 * X @ Y.T
        K = sigma ** 2 * np.exp(-D2 / (2.0 * ell ** 2))
        K[K < sp] = 0.0
        A = CI * K
        s, S = A.sum(1), A @ Y
        w = np.cross(X, S).sum(0) / c
        nu = (S - s[:, None] * X).sum(0) / d
        om = hat(w)
        xi = np.cross(np.tile(w, (len(Y), 1)), Y) + nu
        x2 = (om @ om @ Y.T + (om @ nu)[:, None]).T
        x3 = (om @ om @ om @ Y.T + (om @ om @ nu)[:, None]).T
        x4 = (om @ om @ om @ om @ Y.T + (om @ om @ om @ nu)[:, None]).T
        n2 = (xi ** 2).sum(1)
        dx = 2.0 * (-xi * x2).sum(1)
        ec = (x2 ** 2).sum(1) + 2.0 * (xi * x3).sum(1)
        tc = 1.0 / (2.0 * ell ** 2)
        beta = -(X @ xi.T - (xi * Y).sum(1)[None, :]) / ell ** 2
        gamma = -tc * (n2[None, :] + 2.0 * (X @ x2.T - (x2 * Y).sum(1)[None, :]))
        delta = tc * (dx[None, :] + 2.0 * (-(X @ x3.T) + (x3 * Y).sum(1)[None, :]))
        epsil = -tc * (ec[None, :] + 2.0 * (X @ x4.T - (x4 * Y).sum(1)[None, :]))
        B = (A * beta).sum()
        C = (A * (gamma + beta ** 2 / 2.0)).sum()
        D = (A * (delta + beta * gamma + beta ** 3 / 6.0)).sum()
        E = (A * (epsil + beta * delta + 0.5 * beta * beta * gamma + 0.5 * gamma * gamma +
                  beta ** 4 / 24.0)).sum()
        roots = [r.real for r in np.roots([4 * E, 3 * D, 2 * C, B]) if r.imag == 0 and r.real > 0]
        step = min(min(roots), 0.8) if roots else min_step
        if max(np.linalg.norm(w), np.linalg.norm(nu)) < eps:
            break
        th = np.linalg.norm(w)
        dR = np.eye(3) + (np.sin(step * th) / th) * om + ((1 - np.cos(step * th)) / th ** 2) * om @ om
        dT = (step * np.eye(3) + (1 - np.cos(step * th)) / th ** 2 * om +
              ((step * th - np.sin(step * th)) / th ** 3) * om @ om) @ nu
        R, T = R @ dR, R @ dT + T
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = dR, dT
        if np.linalg.norm(logm(M), "fro") < eps2:
            break
        if k > 3:
            ell = 0.10
        if k > 10:
            ell = 0.06
        if k > 20:
            ell = 0.03
    out = np.eye(4)
    out[:3, :3], out[:3, 3] = R.T, -R.T @ T
    return out, k
