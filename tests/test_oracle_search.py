"""CPU: WHICH pairs enter the Gram matrix.  The oracle's dense-threshold and
uniform-grid searches against each other and against the reference's own
vendored nanoflann (golden digests made by tools/make_golden.py; the live
library oracle/_ref when it was built)."""
import hashlib

import numpy as np
import pytest


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_dense_and_grid_search_build_the_same_csr(pkg, po):
    xf, ff, xm, fm = pkg.data.synthetic_pair(1500, 1700, seed=21)
    p = po.default_params(po.MODE_CVO)
    for ell in (0.15, 0.06, 0.03):
        a = po.se_kernel(p, ell, xf, ff, xm, fm, search=po.SEARCH_DENSE)
        b = po.se_kernel(p, ell, xf, ff, xm, fm, search=po.SEARCH_GRID)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))
        assert a[0][-1] > 0


def test_radius_sets_match_reference_nanoflann_digests(po, desk, golden_json):
    """Strict '<' on an exact search, float32 FMA-chain metric: same rows,
    columns and squared distances as the reference's kd-tree, bit for bit."""
    gold = golden_json("nanoflann_sets.json")["cases"]
    for case in gold:
        if case["stride"] != 5:
            continue   # the full-size digests are checked in the slow test below
        xa, xb = desk["xyz0"][::5], desk["xyz1"][::5]
        tau = np.uint32(case["tau_bits"]).view(np.float32)
        p = po.default_params(po.MODE_CVO if case["mode"] == "cvo" else po.MODE_ACVO)
        assert np.float32(po.thresholds(p, case["ell"])[0]) == tau
        for search in (po.SEARCH_DENSE, po.SEARCH_GRID):
            rp, col, d2 = po.radius_sets(xa, xb, tau, search)
            assert int(rp[-1]) == case["nnz"]
            assert _digest(rp.astype(np.int64), col.astype(np.int32)) == case["sha256_rowptr_col"]
            assert _digest(d2.astype(np.float32)) == case["sha256_d2"]


def test_radius_sets_full_size_clouds(po, desk, golden_json):
    gold = [c for c in golden_json("nanoflann_sets.json")["cases"]
            if c["stride"] == 1 and c["mode"] == "cvo" and c["ell"] in (0.1, 0.03)]
    xa, xb = desk["xyz0"], desk["xyz1"]
    for case in gold:
        tau = np.uint32(case["tau_bits"]).view(np.float32)
        rp, col, d2 = po.radius_sets(xa, xb, tau, po.SEARCH_GRID)
        assert int(rp[-1]) == case["nnz"]
        assert _digest(rp.astype(np.int64), col.astype(np.int32)) == case["sha256_rowptr_col"]
        assert _digest(d2.astype(np.float32)) == case["sha256_d2"]


def test_live_reference_nanoflann_when_built(po):
    if po.ref_lib() is None:
        pytest.skip("oracle/_ref not built in this checkout")
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (800, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (900, 3)).astype(np.float32)
    b[:5] = a[:5]                       # exact coincidences (d2 = 0)
    for tau in (np.float32(0.02), np.float32(4e-4)):
        rp, col, d2 = po.ref_radius_search(b, a, tau)
        o = np.lexsort((col, np.repeat(np.arange(a.shape[0]), np.diff(rp))))
        mine = po.radius_sets(a, b, tau, po.SEARCH_GRID)
        assert np.array_equal(rp, mine[0]) and np.array_equal(col[o], mine[1])
        assert np.array_equal(d2[o].view(np.uint32), mine[2].view(np.uint32))


def test_threshold_is_strict(po):
    """A pair exactly AT the radius is not a neighbour (nanoflann.hpp:250)."""
    a = np.zeros((1, 3), np.float32)
    b = np.array([[0.25, 0, 0], [0.125, 0, 0]], np.float32)
    tau = np.float32(0.0625)            # = 0.25^2 exactly
    rp, col, d2 = po.radius_sets(a, b, tau, po.SEARCH_DENSE)
    assert list(col) == [1] and d2[0] == np.float32(0.015625)
