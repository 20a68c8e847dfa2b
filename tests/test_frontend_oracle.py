"""CPU: the front end's restatement (oracle/frontend_oracle.c) against
known answers of the definitions it restates, the host-only entry points of
include/cvo_frontend.h, and the file side of the drivers.  PARITY UNPINNED for the
image stages (no OpenCV, no frames in the reference tree); the random pattern IS
pinned: the product's restated generator against the C library's rand()."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import pyoracle_fe as fo
from conftest import low_texture_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cvo_frontend.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvo_fe_[a-z0-9_]+)\s*\(", text)))


def test_header_binding_and_library_agree(pkg):
    assert sorted(pkg.frontend.SYMBOLS) == _declared()
    lib = ctypes.CDLL(pkg.capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), "libcvo_hip.so does not export %s" % name


def test_random_pattern_is_the_c_librarys(pkg):
    """ref thirdparty/PixelSelector2.cpp:35-37: srand(3141592); rand() & 0xFF."""
    n = 640 * 480
    mine = pkg.frontend.random_pattern(n)
    libc = fo.random_pattern(n)
    assert np.array_equal(mine, libc)
    assert mine[:8].tolist() == [110, 61, 176, 129, 106, 113, 59, 103]


def test_camera_table(pkg):
    """ref src/pcd_generator.cpp:241-295"""
    fr1 = pkg.frontend.camera(1)
    assert fr1 == pytest.approx({"scaling_factor": 5000.0, "fx": 517.3, "fy": 516.5, "cx": 318.6, "cy": 255.3})
    for seq in range(-1, 8):
        got = pkg.frontend.camera(seq)
        assert list(got.values()) == fo.camera(seq).tolist()
    assert pkg.frontend.camera(17) == pkg.frontend.camera(0)   # default: RealSense


def test_no_cpu_path(pkg):
    if pkg.capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.capi.CvoHipError):
        pkg.frontend.PcdGenerator(640, 480)


def test_gray_and_hsv_known_answers():
    """8-bit cvtColor values of the primaries (channel 0 = R, as the reference calls it)."""
    cols = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128],
                      [255, 255, 0], [0, 255, 255], [255, 0, 255]]], np.uint8)
    assert fo.gray(cols)[0].tolist() == [255, 0, 76, 150, 29, 128, 226, 179, 105]
    assert fo.hsv(cols)[0].tolist() == [[0, 0, 255], [0, 0, 0], [0, 255, 255], [60, 255, 255], [120, 255, 255],
                                        [0, 0, 128], [30, 255, 255], [90, 255, 255], [150, 255, 255]]


def test_hsv_tracks_the_real_valued_definition():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    got = fo.hsv(img).astype(np.int64)
    r, g, b = [img[..., k].astype(np.float64) for k in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    d = v - mn
    s = np.where(v > 0, 255.0 * d / np.maximum(v, 1), 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        hh = np.where(v == r, (g - b) / d, np.where(v == g, 2 + (b - r) / d, 4 + (r - g) / d)) * 30.0
    hh = np.where(d == 0, 0, hh)
    hh = np.where(hh < 0, hh + 180, hh)
    assert np.array_equal(got[..., 2], v.astype(np.int64))
    assert np.abs(got[..., 1] - s).max() <= 1.0
    dh = np.abs(got[..., 0] - hh)
    assert np.minimum(dh, 180 - dh).max() <= 1.0


def test_pyramid_and_flat_index_gradients():
    """ref src/pcd_generator.cpp:79-113"""
    rng = np.random.default_rng(2)
    g = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    I, dx0, dy0, ag = fo.pyramid(g)
    f = g.astype(np.float32)
    assert np.array_equal(I[0], f)
    want1 = np.float32(0.25) * (((f[0::2, 0::2] + f[0::2, 1::2]) + f[1::2, 0::2]) + f[1::2, 1::2])
    assert np.array_equal(I[1], want1)
    assert I[2].shape == (16, 24)
    # interior: plain central differences
    assert np.array_equal(dx0[1:-1, 1:-1], np.float32(0.5) * (f[1:-1, 2:] - f[1:-1, :-2]))
    assert np.array_equal(dy0[1:-1, :], np.float32(0.5) * (f[2:, :] - f[:-2, :]))
    # first column: the left neighbour is the last pixel of the previous row (flattened image)
    assert np.array_equal(dx0[1:-1, 0], np.float32(0.5) * (f[1:-1, 1] - f[0:-2, -1]))
    # first and last row are never written
    assert not dx0[0].any() and not dx0[-1].any() and not ag[0][0].any() and not ag[0][-1].any()
    assert np.array_equal(ag[0][1:-1], dx0[1:-1] * dx0[1:-1] + dy0[1:-1] * dy0[1:-1])


def test_cell_thresholds():
    """ref thirdparty/PixelSelector2.cpp:70-131: flat cells get (0 + 7)^2; a cell whose
    gradient magnitude is 10 everywhere gets its neighbours' mean"""
    ag0 = np.zeros((96, 96), np.float32)
    assert np.array_equal(fo.thresholds(ag0), np.full((3, 3), 49.0, np.float32))
    ag0[32:64, 32:64] = 100.0
    t = fo.thresholds(ag0)
    f = np.float32
    assert t[1, 1] == (f(8 * 7 + 17) / f(9)) * (f(8 * 7 + 17) / f(9))
    assert t[0, 0] == (f(3 * 7 + 17) / f(4)) * (f(3 * 7 + 17) / f(4))


def test_selector_map_properties(pkg):
    for tex, seed in ((0.3, 1), (1.0, 2), (3.0, 3)):
        bgr, dep = pkg.data.synthetic_rgbd_frame(seed=seed, texture=tex)
        r = fo.create_pointcloud(bgr, dep, 1, 1)
        m = r["map"]
        assert set(np.unique(m).tolist()) <= {0.0, 1.0, 2.0, 4.0}
        # ref thirdparty/PixelSelector2.cpp:313: a margin of the image is never selected
        assert not m[:4].any() and not m[-3:].any() and not m[:, :4].any() and not m[:, -5:].any()
        assert r["num_selected"] == np.count_nonzero(m)
        valid = (m != 0) & (dep != 0)
        assert len(r["positions"]) == np.count_nonzero(valid)
        again = fo.create_pointcloud(bgr, dep, 1, 1)
        assert np.array_equal(again["map"], m)
    # desk-like texture lands near num_want (ref src/pcd_generator.cpp:22)
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=2, texture=1.0)
    assert 2400 <= fo.create_pointcloud(bgr, dep, 1, 1)["num_selected"] <= 3600


def test_back_projection_and_features(pkg):
    """ref src/pcd_generator.cpp:297-321,336-380"""
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=4, texture=1.0)
    for ftype in (0, 1):
        r = fo.create_pointcloud(bgr, dep, 1, ftype)
        ys, xs = np.nonzero((r["map"] != 0) & (dep != 0))
        z = dep[ys, xs].astype(np.float32) / np.float32(5000.0)
        assert np.array_equal(r["positions"][:, 2], z)
        assert np.array_equal(r["positions"][:, 0], (xs.astype(np.float32) - np.float32(318.6)) * z / np.float32(517.3))
        assert np.array_equal(r["positions"][:, 1], (ys.astype(np.float32) - np.float32(255.3)) * z / np.float32(516.5))
        _, dx0, dy0, _ = fo.pyramid(fo.gray(bgr))
        if ftype == 1:
            assert np.array_equal(r["features"][:, :3], bgr[ys, xs].astype(np.float32))
            assert np.array_equal(r["features"][:, 3], dx0[ys, xs])
        else:
            hv = fo.hsv(bgr)[ys, xs].astype(np.float64)
            assert np.array_equal(r["features"][:, 0], (hv[:, 0] / 180.0).astype(np.float32))
            assert np.array_equal(r["features"][:, 2], (hv[:, 2] / 255.0).astype(np.float32))
            assert np.array_equal(r["features"][:, 4], (dy0[ys, xs].astype(np.float64) / 255.0 * 2).astype(np.float32))


def test_canny_on_a_step_edge():
    """cv::Canny(0, 25, 3) semantics: one-pixel-wide line on a step, nothing on a flat image"""
    g = np.full((40, 48), 60, np.uint8)
    assert not fo.canny(fo.blur3(g)).any()
    g[:, 24:] = 160
    e = fo.canny(fo.blur3(g))
    cols = np.nonzero(e.any(axis=0))[0]
    assert len(cols) == 1 and cols[0] in (23, 24)
    assert (e[:, cols[0]] == 255).all()
    # blur: box mean with reflected borders
    b = fo.blur3(g)
    assert b[5, 23] == round((60 * 6 + 160 * 3) / 9.0) and b[5, 0] == 60 and b[5, 47] == 160


def test_canny_top_up_rule(pkg):
    """ref src/pcd_generator.cpp:143-175: a nearly flat image with a few sharp shapes: the
    selector keeps < num_want/3, every 8x8 block adds at most one edge pixel"""
    bgr, dep = low_texture_frame(pkg)
    r = fo.create_pointcloud(bgr, dep, 1, 1)
    assert r["num_selected"] < 1000
    added = np.count_nonzero(r["map"]) - r["num_selected"]
    assert added > 50
    edges = fo.canny(fo.blur3(fo.gray(bgr)))
    # pixels in the map that are not on an edge were all put there by the selector
    assert ((r["map"] != 0) & (edges == 0)).sum() <= r["num_selected"]
    # blocks with a free edge pixel got exactly one more pixel; the others none
    per_block_edges = (edges != 0).reshape(60, 8, 80, 8).sum(axis=(1, 3))
    assert added <= np.count_nonzero(per_block_edges)
    assert added >= np.count_nonzero(per_block_edges) - r["num_selected"]


def test_association_list_and_image_loading(pkg, tmp_path):
    """ref src/cvo_main.cpp:69-106"""
    Image = pytest.importorskip("PIL.Image")
    bgr, dep = pkg.data.synthetic_rgbd_frame(width=96, height=64, seed=1)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    Image.fromarray(bgr[:, :, ::-1]).save(tmp_path / "rgb" / "1.5.png")
    Image.fromarray(dep).save(tmp_path / "depth" / "1.6.png")
    (tmp_path / "assoc.txt").write_text("1.5 rgb/1.5.png 1.6 depth/1.6.png\n\n2.5 rgb/2.5.png 2.6 depth/2.6.png\n")
    names, rgbs, deps = pkg.frontend.load_file_name(str(tmp_path / "assoc.txt"))
    assert names == ["1.5", "2.5"] and rgbs[0] == "rgb/1.5.png" and deps[1] == "depth/2.6.png"
    b2, d2 = pkg.frontend.load_img(str(tmp_path / rgbs[0]), str(tmp_path / deps[0]))
    assert np.array_equal(b2, bgr) and np.array_equal(d2, dep) and d2.dtype == np.uint16
