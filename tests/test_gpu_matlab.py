"""GPU: the MATLAB object's pair weight (linear colour inner product,
threshold on K only; ref matlab/@rkhs_se3_registration/rkhs_se3_registration.m) on the
HIP kernels -- bit for bit against the C restatement with the same weight, within 1e-4
of the float64 restatement of the whole MATLAB object, and (soft) near the transform the
reference's MATLAB run recorded."""
import json
import os

import numpy as np
import pytest

from oracle import matlab_dense
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _prepared(pkg, desk, a, b):
    f = pkg.data.grid_average(*pkg.data.pc_range_filter(desk["xyz%d" % a], desk["rgb%d" % a]))
    m = pkg.data.grid_average(*pkg.data.pc_range_filter(desk["xyz%d" % b], desk["rgb%d" % b]))
    return f[0], f[1], m[0], m[1]


def _oracle(fx, fc, mx, mc):
    p = po.default_params(po.MODE_MATLAB)
    st = po.init_state(p)
    feat = lambda c: np.concatenate([np.asarray(c, np.float32), np.zeros((len(c), 2), np.float32)], axis=1)
    n, _ = po.align(p, st, fx, feat(fc), mx, feat(mc), search=po.SEARCH_DENSE)
    return n, st


def test_defaults(pkg):
    p = pkg.capi.default_params(pkg.capi.MODE_MATLAB)
    assert p.mode == pkg.capi.MODE_CVO and p.color_scale == np.float32(1e-5)
    assert p.sp_thres == np.float32(1e-3) and p.eps == np.float32(5e-4) and p.eps_2 == np.float32(1e-4)
    assert pkg.capi.default_params(pkg.capi.MODE_CVO).color_scale == 0.0


def test_shipped_pairs_match_restatements(pkg, desk):
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matlab_transforms.json")))
    reg = pkg.RkhsMatlab()
    for a in range(3):
        fx, fc, mx, mc = _prepared(pkg, desk, a, a + 1)
        T, n = reg.register(fx, fc, mx, mc)
        n_or, st = _oracle(fx, fc, mx, mc)
        assert n == n_or
        assert np.array_equal(T, po.state_matrices(st)[0])          # bit for bit vs the C restatement
        T64, k64 = matlab_dense.align(fx, fc, mx, mc)
        assert abs(n - k64) <= 1 and np.abs(T - T64).max() < 1e-4  # float64 restatement of the object
        G = np.array(gold["matlab"][a + 1])
        assert np.abs(T - G).max() < 6e-3                           # soft: the recorded MATLAB run
    reg.close()


@pytest.mark.parametrize("n,m,seed", [(900, 800, 17), (3000, 2500, 18), (6000, 6000, 19)])
def test_synthetic_pairs_bit_identical(pkg, n, m, seed):
    xf, _, xm, _ = pkg.data.synthetic_pair(n, m, seed=seed)
    rng = np.random.default_rng(seed)
    cf = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    cm = rng.integers(0, 256, (m, 3)).astype(np.uint8)
    reg = pkg.RkhsMatlab()
    T, it = reg.register(xf, cf, xm, cm)
    n_or, st = _oracle(xf, cf, xm, cm)
    assert it == n_or and np.array_equal(T, po.state_matrices(st)[0])
    assert list(reg.state.R) == list(st.R) and list(reg.state.T) == list(st.T) and reg.state.ell == st.ell
    # every pair starts from scratch (ref rkhs_se3_registration.m:112-114): same call, same result
    T2, it2 = reg.register(xf, cf, xm, cm)
    assert it2 == it and np.array_equal(T2, T)
    reg.close()


def test_matlab_weight_soak(pkg):
    """tools/gpu_soak.py with the MATLAB weight: random clouds, perturbed thresholds; dense
    small clouds overflow their first tile lists here (the grow-and-redo path: a consumer
    must not read entries an overflowed list never received)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAK_MATLAB="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_soak.py"), "120", "2500"],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "120 cases, 0 mismatches" in out.stdout


def test_cloud_preparation_on_the_device_equals_the_oracle(pkg, desk):
    """SURVEY 8 f2 on the GPU: pcRangeFilter + grid-average downsampling (ref util/pcRangeFilter.m:5-12,
    data/rgbd_dataset/rgbddataset_rkhs.m:36-39,58) through cvo_hip_range_filter_grid_average against the
    numpy oracle (oracle/matlab_prep.py), bit for bit: the five shipped fr1/desk clouds at the MATLAB
    run's settings (4.0 / 0.8 m, 0.05 m), either step alone, random clouds of 1 ... 200 000 points with
    other grids, and the edge cases -- nothing in range, one point, everything in one voxel."""
    from oracle import matlab_prep as mp
    data = pkg.data

    def same(a, b):
        assert a[0].dtype == np.float32 and a[1].dtype == np.uint8
        assert a[0].shape == b[0].shape and a[1].shape == b[1].shape
        assert np.array_equal(a[0].view(np.uint32), np.asarray(b[0], np.float32).view(np.uint32))
        assert np.array_equal(a[1], b[1])

    for k in range(5):
        xyz, rgb = desk["xyz%d" % k], desk["rgb%d" % k]
        f_or = mp.pc_range_filter(xyz, rgb, 4.0, 0.8)
        same(data.pc_range_filter(xyz, rgb, 4.0, 0.8), f_or)
        same(data.grid_average(f_or[0], f_or[1], 0.05), mp.grid_average(f_or[0], f_or[1], 0.05))
        g = data.prepare_matlab_cloud(xyz, rgb, 4.0, 0.8, 0.05)
        same(g, mp.grid_average(f_or[0], f_or[1], 0.05))
        assert 600 <= len(g[0]) <= 800   # the MATLAB run registered ~700-point clouds
    rng = np.random.default_rng(8)
    for n, grid, rmax, rmin in ((1, 0.05, 4.0, 0.0), (7, 0.3, 0.0, 0.0), (1000, 0.01, 3.0, 1.0), (50000, 0.05, 4.0, 0.8),
                                (200000, 0.02, 5.0, 0.5), (3000, 100.0, 0.0, 0.0)):
        xyz = (rng.normal(size=(n, 3)) * [1.5, 1.0, 0.7] + [0.2, -0.1, 2.0]).astype(np.float32)
        rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
        f_or = mp.pc_range_filter(xyz, rgb, rmax, rmin) if rmax > 0 else (xyz, rgb)
        same(data.prepare_matlab_cloud(xyz, rgb, rmax, rmin, grid), mp.grid_average(f_or[0], f_or[1], grid))
    xyz = np.full((40, 3), 9.0, np.float32)       # nothing in range
    out = data.prepare_matlab_cloud(xyz, np.zeros((40, 3), np.uint8), 4.0, 0.8, 0.05)
    assert out[0].shape == (0, 3) and out[1].shape == (0, 3)
    out = data.prepare_matlab_cloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8))
    assert out[0].shape == (0, 3)
