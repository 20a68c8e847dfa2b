"""GPU: the HIP front end (include/cvo_frontend.h) against its CPU
restatement (oracle/frontend_oracle.c) on the same synthetic frames: every
intermediate image and the cloud, bit for bit."""
import os

import numpy as np
import pytest

from conftest import low_texture_frame
from oracle import pyoracle_fe as fo

pytestmark = pytest.mark.gpu


def _check_frame(pkg, gen, bgr, dep, seq, ftype, num_want=3000):
    F = pkg.frontend
    xyz, feat = gen.create_pointcloud(bgr, dep, seq, ftype)
    ref = fo.create_pointcloud(bgr, dep, seq, ftype, num_want)
    info = gen.info()
    g = fo.gray(bgr)
    assert np.array_equal(gen.read_stage(F.STAGE_GRAY), g)
    assert np.array_equal(gen.read_stage(F.STAGE_HSV), fo.hsv(bgr))
    _, dx0, dy0, ag = fo.pyramid(g)
    assert np.array_equal(gen.read_stage(F.STAGE_DX0), dx0) and np.array_equal(gen.read_stage(F.STAGE_DY0), dy0)
    for l, st in enumerate((F.STAGE_AG0, F.STAGE_AG1, F.STAGE_AG2)):
        assert np.array_equal(gen.read_stage(st), ag[l]), "squared gradients, level %d" % l
    assert np.array_equal(gen.read_stage(F.STAGE_THS), fo.thresholds(ag[0]))
    assert np.array_equal(gen.read_stage(F.STAGE_MAP), ref["map"])
    assert info["num_selected"] == ref["num_selected"]
    assert info["num_points"] == len(ref["positions"]) == len(xyz)
    assert np.array_equal(xyz.view(np.uint32), ref["positions"].view(np.uint32))
    assert np.array_equal(feat.view(np.uint32), ref["features"].view(np.uint32))
    return info


@pytest.mark.parametrize("texture,seed", [(0.3, 11), (1.0, 12), (3.0, 13), (6.0, 14)])
def test_vga_frames_match_the_oracle(pkg, texture, seed):
    gen = pkg.frontend.PcdGenerator(640, 480)
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=seed, texture=texture)
    seen = []
    for ftype in (pkg.frontend.FEATURES_RGB, pkg.frontend.FEATURES_HSV):
        seen.append(_check_frame(pkg, gen, bgr, dep, 1, ftype))
    assert seen[0]["num_selected"] == seen[1]["num_selected"]
    gen.close()


def test_both_selector_branches_are_exercised(pkg):
    """sparse texture -> re-selection with a smaller potential; busy -> a larger one"""
    gen = pkg.frontend.PcdGenerator(640, 480)
    pots = {}
    for texture, seed in ((0.3, 11), (6.0, 14)):
        bgr, dep = pkg.data.synthetic_rgbd_frame(seed=seed, texture=texture)
        gen.create_pointcloud(bgr, dep)
        pots[texture] = gen.info()
    assert pots[0.3]["reselected"] == 1 and pots[0.3]["pot_used"] < 3
    assert pots[6.0]["pot_used"] >= 3
    gen.close()


@pytest.mark.parametrize("w,h", [(320, 256), (352, 300), (96, 64), (1280, 720), (333, 251), (127, 193)])
def test_other_image_sizes(pkg, w, h):
    gen = pkg.frontend.PcdGenerator(w, h, num_want=max(200, w * h // 100))
    for seed in (1, 2):
        bgr, dep = pkg.data.synthetic_rgbd_frame(width=w, height=h, seed=seed, texture=1.0)
        _check_frame(pkg, gen, bgr, dep, 3, pkg.frontend.FEATURES_RGB, num_want=max(200, w * h // 100))
    gen.close()


def test_canny_top_up_matches(pkg):
    gen = pkg.frontend.PcdGenerator(640, 480)
    bgr, dep = low_texture_frame(pkg)
    info = _check_frame(pkg, gen, bgr, dep, 1, pkg.frontend.FEATURES_RGB)
    assert info["canny_used"] == 1 and info["num_selected"] < 1000
    edges = fo.canny(fo.blur3(fo.gray(bgr)))
    assert np.array_equal(gen.read_stage(pkg.frontend.STAGE_EDGES), edges)
    # a normal frame afterwards on the same object: no top-up, still exact
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=3, texture=1.0)
    assert _check_frame(pkg, gen, bgr, dep, 1, pkg.frontend.FEATURES_HSV)["canny_used"] == 0
    gen.close()


def test_long_edge_chains_in_hysteresis(pkg):
    """a spiral of weak edge pixels hanging off one strong pixel: many sweeps"""
    h, w = 256, 256
    bgr = np.full((h, w, 3), 90, np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    r = np.hypot(x - 128, y - 128)
    th = np.arctan2(y - 128, x - 128)
    spiral = np.abs(((r / 14.0 - th / (2 * np.pi)) % 1.0) - 0.5) < 0.18
    bgr[spiral] += 9
    bgr[126:131, 126:131] = 200
    dep = np.full((h, w), 6000, np.uint16)
    gen = pkg.frontend.PcdGenerator(w, h, num_want=30000)
    info = _check_frame(pkg, gen, bgr, dep, 1, pkg.frontend.FEATURES_RGB, num_want=30000)
    assert info["canny_used"] == 1
    edges = fo.canny(fo.blur3(fo.gray(bgr)))
    assert np.count_nonzero(edges) > 2000
    assert np.array_equal(gen.read_stage(pkg.frontend.STAGE_EDGES), edges)
    gen.close()


def test_argument_checks_and_capacity(pkg):
    import ctypes as C
    F = pkg.frontend
    gen = F.PcdGenerator(640, 480)
    with pytest.raises(ValueError):
        gen.create_pointcloud(np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10), np.uint16))
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=1, texture=1.0)
    pos = np.zeros((100, 3), np.float32); feat = np.zeros((100, 5), np.float32)
    n = C.c_int(0)
    fp = C.POINTER(C.c_float)
    st = F.lib().cvo_fe_create_pointcloud(gen._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)), 640 * 3,
                                          dep.ctypes.data_as(C.POINTER(C.c_uint16)), 640 * 2, 1, 1,
                                          pos.ctypes.data_as(fp), feat.ctypes.data_as(fp), 100, C.byref(n))
    assert st == -1 and n.value > 100 and b"capacity" in F.lib().cvo_fe_last_error(gen._h)
    ref = fo.create_pointcloud(bgr, dep, 1, 1)
    assert np.array_equal(pos, ref["positions"][:100])      # the first `capacity` points are delivered
    st = F.lib().cvo_fe_create_pointcloud(gen._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)), 640,
                                          dep.ctypes.data_as(C.POINTER(C.c_uint16)), 640 * 2, 1, 1,
                                          pos.ctypes.data_as(fp), feat.ctypes.data_as(fp), 100, C.byref(n))
    assert st == -1                                        # stride shorter than a row
    h = C.c_void_p()
    assert F.lib().cvo_fe_create(0, None, 8, 8, C.byref(h)) == -1
    gen.close()


def test_padded_rows(pkg):
    """img_stride / depth_stride larger than a row (cv::Mat::step of a sub-image)"""
    import ctypes as C
    F = pkg.frontend
    w, h = 320, 256
    bgr, dep = pkg.data.synthetic_rgbd_frame(width=w, height=h, seed=8, texture=1.0)
    big = np.zeros((h, w * 3 + 24), np.uint8); big[:, :w * 3] = bgr.reshape(h, -1)
    bigd = np.zeros((h, w + 6), np.uint16); bigd[:, :w] = dep
    gen = F.PcdGenerator(w, h)
    pos = np.zeros((gen.capacity, 3), np.float32); feat = np.zeros((gen.capacity, 5), np.float32)
    n = C.c_int(0)
    fp = C.POINTER(C.c_float)
    st = F.lib().cvo_fe_create_pointcloud(gen._h, big.ctypes.data_as(C.POINTER(C.c_uint8)), big.strides[0],
                                          bigd.ctypes.data_as(C.POINTER(C.c_uint16)), bigd.strides[0], 2, 0,
                                          pos.ctypes.data_as(fp), feat.ctypes.data_as(fp), gen.capacity, C.byref(n))
    assert st == 0
    ref = fo.create_pointcloud(bgr, dep, 2, 0)
    assert n.value == len(ref["positions"]) and np.array_equal(pos[:n.value], ref["positions"])
    assert np.array_equal(feat[:n.value], ref["features"])
    gen.close()


def test_directory_run_feeds_the_registration(pkg, tmp_path):
    """ref src/cvo_main.cpp:20-66: PNG pair -> front end -> run_cvo -> pose line; the result
    is the one the registration gives on the oracle's clouds of the same frames"""
    Image = pytest.importorskip("PIL.Image")
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines, frames = [], []
    for k in range(3):
        bgr, dep = pkg.data.synthetic_rgbd_frame(seed=21, texture=1.0, motion=(1.5 * k, 0.75 * k))
        frames.append((bgr, dep))
        stamp = "%.6f" % (1305031453.0 + k / 30.0)
        Image.fromarray(bgr[:, :, ::-1]).save(tmp_path / "rgb" / (stamp + ".png"))
        Image.fromarray(dep).save(tmp_path / "depth" / (stamp + ".png"))
        lines.append("%s rgb/%s.png %s depth/%s.png" % (stamp, stamp, stamp, stamp))
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    for cls, ftype in ((pkg.Cvo, 1), (pkg.Acvo, 0)):
        reg = cls()
        out = tmp_path / ("poses_%d.txt" % ftype)
        with pkg.trajectory.TrajectoryWriter(str(out)) as wr:
            assert pkg.frontend.run_directory(reg, str(tmp_path), 1, writer=wr) == 3
        assert len(open(out).read().strip().split("\n")) == 3
        ref = cls()
        for bgr, dep in frames:
            r = fo.create_pointcloud(bgr, dep, 1, ftype)
            ref.run_cvo(r["positions"], r["features"])
        assert np.array_equal(reg.accum_transform, ref.accum_transform)
        assert reg.num_iterations == ref.num_iterations > 0
        # the image moved by ~(1.5, 0.75) px per frame at ~1.5 m: a few millimetres
        assert 1e-4 < np.linalg.norm(reg.accum_transform[:3, 3]) < 0.05
        reg.close(); ref.close()


@pytest.mark.parametrize("mode_name", ["cvo", "acvo"])
def test_cpp_objects_take_images(pkg, tmp_path, mode_name):
    """include/cvo.hpp: run_cvo(dataset_seq, RGB_img, dep_img, ...) as the reference's
    drivers call it (ref src/cvo_main.cpp:52-64): the pose lines equal the Python path's"""
    import io
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cvo_image_demo")
    lib = os.path.join(root, "cvo-rgbd_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "cvo_image_demo.cpp"), "-L", lib, "-lcvo_hip",
                    "-Wl,-rpath," + lib, "-o", exe], check=True)
    w, h = 640, 480
    frames = [pkg.data.synthetic_rgbd_frame(seed=31, texture=1.0, motion=(1.2 * k, -0.6 * k)) for k in range(3)]
    names = ["1305031453.%06d" % (359684 + 33333 * k) for k in range(3)]
    path = str(tmp_path / "frames.bin")
    with open(path, "wb") as fh:
        fh.write(struct.pack("<iii", len(frames), w, h))
        for name, (bgr, dep) in zip(names, frames):
            fh.write(name.encode().ljust(32, b"\0"))
            fh.write(bgr.tobytes()); fh.write(dep.tobytes())
    out = subprocess.run([exe, path, mode_name, "1"], check=True, capture_output=True, text=True).stdout
    reg = (pkg.Acvo if mode_name == "acvo" else pkg.Cvo)()
    gen = pkg.frontend.PcdGenerator(w, h)
    buf = io.StringIO()
    wr = pkg.trajectory.TrajectoryWriter(buf)
    ftype = pkg.frontend.FEATURES_HSV if mode_name == "acvo" else pkg.frontend.FEATURES_RGB
    for name, (bgr, dep) in zip(names, frames):
        xyz, feat = gen.create_pointcloud(bgr, dep, 1, ftype)
        reg.run_cvo(xyz, feat)
        wr.append(name, reg.accum_transform)
    want = buf.getvalue().strip().split("\n")
    got = out.strip().split("\n")
    assert got[:3] == want
    assert got[3] == "points_last_frame %d iterations %d" % (len(xyz), reg.num_iterations)
    reg.close(); gen.close()


def test_submit_collect_and_prefetched_stream(pkg):
    """cvo_fe_submit / cvo_fe_collect: same cloud as the one-call form; one frame in flight;
    the driver loop with prefetch gives the poses of the plain loop"""
    import io
    F = pkg.frontend
    gen = F.PcdGenerator(640, 480)
    frames = [("%d.0" % k,) + pkg.data.synthetic_rgbd_frame(seed=41, texture=1.0, motion=(1.0 * k, 0.5 * k))
              for k in range(4)]
    xyz, feat = gen.create_pointcloud(frames[0][1], frames[0][2], 1, F.FEATURES_HSV)
    gen.submit(frames[0][1], frames[0][2], 1, F.FEATURES_HSV)
    with pytest.raises(pkg.capi.CvoHipError):
        gen.submit(frames[1][1], frames[1][2], 1, F.FEATURES_HSV)     # one frame in flight
    x2, f2 = gen.collect()
    assert np.array_equal(xyz, x2) and np.array_equal(feat, f2)
    with pytest.raises(pkg.capi.CvoHipError):
        gen.collect()                                                  # nothing submitted
    # a frame that needs the top-up, through the split form
    bgr, dep = low_texture_frame(pkg)
    gen.submit(bgr, dep, 1, F.FEATURES_RGB)
    x3, f3 = gen.collect()
    ref = fo.create_pointcloud(bgr, dep, 1, 1)
    assert gen.info()["canny_used"] == 1 and np.array_equal(x3, ref["positions"]) and np.array_equal(f3, ref["features"])
    poses = []
    for prefetch, device in ((True, False), (False, False), (False, True), (True, True)):
        reg = pkg.Acvo()
        buf = io.StringIO()
        assert F.run_frames(reg, frames, 1, writer=pkg.trajectory.TrajectoryWriter(buf), generator=gen,
                            prefetch=prefetch, device=device) == 4
        poses.append(buf.getvalue())
        reg.close()
    # prefetched, serial, and with the cloud handed over in device memory: the same poses
    assert poses[0] == poses[1] == poses[2] == poses[3] and len(poses[0].strip().split("\n")) == 4
    gen.close()


def test_empty_and_saturated_frames(pkg):
    """no depth at all, a constant image, a maximally busy (random) image"""
    gen = pkg.frontend.PcdGenerator(640, 480)
    bgr, dep = pkg.data.synthetic_rgbd_frame(seed=9, texture=1.0)
    info = _check_frame(pkg, gen, bgr, np.zeros_like(dep), 1, pkg.frontend.FEATURES_RGB)
    assert info["num_points"] == 0 and info["num_selected"] > 2000
    flat = np.full_like(bgr, 77)
    info = _check_frame(pkg, gen, flat, dep, 1, pkg.frontend.FEATURES_HSV)
    assert info["num_points"] == 0 and info["num_selected"] == 0 and info["canny_used"] == 1
    rng = np.random.default_rng(1)
    noise = rng.integers(0, 256, bgr.shape, dtype=np.uint8)
    info = _check_frame(pkg, gen, noise, dep, 1, pkg.frontend.FEATURES_RGB)
    assert info["reselected"] == 1 and info["pot_used"] > 3 and 2000 < info["num_selected"] < 4500
    black = np.zeros_like(bgr); white = np.full_like(bgr, 255)
    _check_frame(pkg, gen, black, dep, 1, pkg.frontend.FEATURES_HSV)
    _check_frame(pkg, gen, white, dep, 1, pkg.frontend.FEATURES_HSV)
    gen.close()


def test_front_end_soak(pkg):
    """tools/gpu_soak_fe.py, 40 random frames: sizes, textures, densities, cameras, feature types"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_soak_fe.py"), "40"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 mismatches" in out.stdout


def test_decoding_straight_into_the_staging_images(pkg):
    """cvo_fe_host_buffers: frames written in place into the pinned staging images give the
    same clouds as frames handed over from ordinary arrays"""
    gen = pkg.frontend.PcdGenerator(640, 480)
    img, dep = gen.host_buffers()
    assert img.shape == (480, 640, 3) and dep.shape == (480, 640)
    for seed in (51, 52, 53):
        bgr, depth = pkg.data.synthetic_rgbd_frame(seed=seed, texture=1.0)
        want = gen.create_pointcloud(bgr, depth, 1, pkg.frontend.FEATURES_HSV)
        img[...] = bgr
        dep[...] = depth
        got = gen.create_pointcloud(img, dep, 1, pkg.frontend.FEATURES_HSV)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    gen.close()


@pytest.mark.parametrize("mode_name,nframes", [("acvo", 80), ("cvo", 60)])
def test_streamed_chain_equals_the_oracle_chain(pkg, mode_name, nframes):
    """The shape of BASELINE configs[2] (the fr1/desk PNGs are not in the reference's tree): a synthetic
    VGA sequence through the front end and ONE registration object -- clouds handed over in device
    memory, the state carried from pair to pair as the reference's drivers do (ref
    src/adaptive_cvo_main.cpp:36-66, src/cvo_main.cpp:36-66) -- against the oracle's front end and
    the oracle chain: every transform, every accumulated pose and every iteration count identical."""
    from oracle import pyoracle as po, pyoracle_fe as fo
    acvo = mode_name == "acvo"
    po.set_threads(16)
    frames = [pkg.data.synthetic_rgbd_frame(seed=55, texture=1.0 + 0.5 * np.sin(k / 5.0),
                                            motion=(1.2 * k, 0.6 * np.sin(k / 3.0) * 4)) for k in range(nframes)]
    ftype = 0 if acvo else 1
    reg = (pkg.Acvo if acvo else pkg.Cvo)()
    gen = pkg.frontend.PcdGenerator(640, 480)
    p = po.default_params(po.MODE_ACVO if acvo else po.MODE_CVO)
    s = po.init_state(p)
    prev, bad = None, []
    for k, (bgr, dep) in enumerate(frames):
        gen.submit(bgr, dep, 1, ftype)
        d_xyz, d_feat, npts = gen.collect_device()
        reg.run_cvo_device(d_xyz, d_feat, npts)
        r = fo.create_pointcloud(bgr, dep, 1, ftype)
        cur = (r["positions"], r["features"])
        assert npts == len(cur[0])
        if prev is not None:
            if acvo:   # tail of acvo::set_pcd (ref src/adaptive_cvo.cpp:476-478)
                s.ell = p.ell_init
                s.ell_max = p.ell_max_init
            it, _ = po.align(p, s, prev[0], prev[1], cur[0], cur[1], search=po.SEARCH_GRID, trace_cap=0)
            T_or, _, A_or = po.state_matrices(s)
            if not (it == reg.num_iterations and np.array_equal(T_or, reg.transform) and
                    np.array_equal(A_or, reg.accum_transform)):
                bad.append((k, reg.num_iterations, it))
        prev = cur
    reg.close()
    gen.close()
    assert not bad, bad
