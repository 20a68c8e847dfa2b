"""RGB-D front end: Python mirror of the reference's
`cvo::pcd_generator` (ref cpp/rkhs_registration/include/pcd_generator.hpp:30-108)
over the C-ABI of include/cvo_frontend.h.  All image work runs as HIP kernels in
libcvo_hip.so; there is no CPU path (creating a generator without a gfx950
device raises).

    gen = PcdGenerator(width=640, height=480)
    xyz, feat = gen.create_pointcloud(bgr, depth, dataset_seq=1, feature_type=FEATURES_RGB)

`bgr` is the h x w x 3 uint8 array an image decoder returns in OpenCV's channel
order (cv::imread: B, G, R) -- the reference passes exactly that to its "RGB"
conversions (ref src/pcd_generator.cpp:389-390), so channel 0 plays the role of R
there and here.  `depth` is h x w uint16 (TUM: 5000 units per metre).

Also here: the file side of the reference's drivers -- the association list and the
image pair of a frame (ref src/cvo_main.cpp:69-106).
"""
import ctypes as C
import os

import numpy as np

from . import capi

FEATURES_HSV, FEATURES_RGB = 0, 1
(STAGE_GRAY, STAGE_HSV, STAGE_MAP, STAGE_AG0, STAGE_AG1, STAGE_AG2, STAGE_THS, STAGE_DX0, STAGE_DY0,
 STAGE_EDGES) = range(10)

SYMBOLS = ("cvo_fe_create", "cvo_fe_destroy", "cvo_fe_last_error", "cvo_fe_set_num_want",
           "cvo_fe_create_pointcloud", "cvo_fe_submit", "cvo_fe_collect", "cvo_fe_collect_device", "cvo_fe_set_device_output", "cvo_fe_host_buffers", "cvo_fe_get_info", "cvo_fe_read_stage", "cvo_fe_random_pattern",
           "cvo_fe_camera")


class Info(C.Structure):
    _fields_ = [("num_selected", C.c_int32), ("pot_used", C.c_int32), ("reselected", C.c_int32),
                ("canny_used", C.c_int32), ("num_points", C.c_int32), ("pad_", C.c_int32)]


_BOUND = False


def lib():
    """libcvo_hip.so with the cvo_fe_* prototypes set (raises if it is not built)."""
    global _BOUND
    L = capi.lib()
    if not _BOUND:
        vp, u8p, u16p, fp = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_float)
        L.cvo_fe_create.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.cvo_fe_destroy.argtypes = [vp]
        L.cvo_fe_last_error.argtypes = [vp]
        L.cvo_fe_last_error.restype = C.c_char_p
        L.cvo_fe_set_num_want.argtypes = [vp, C.c_int]
        L.cvo_fe_create_pointcloud.argtypes = [vp, u8p, C.c_size_t, u16p, C.c_size_t, C.c_int, C.c_int, fp, fp,
                                               C.c_int, C.POINTER(C.c_int)]
        L.cvo_fe_submit.argtypes = [vp, u8p, C.c_size_t, u16p, C.c_size_t, C.c_int, C.c_int]
        L.cvo_fe_collect.argtypes = [vp, fp, fp, C.c_int, C.POINTER(C.c_int)]
        L.cvo_fe_collect_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]
        L.cvo_fe_set_device_output.argtypes = [vp, C.c_int]
        L.cvo_fe_host_buffers.argtypes = [vp, C.POINTER(u8p), C.POINTER(u16p)]
        L.cvo_fe_get_info.argtypes = [vp, C.POINTER(Info)]
        L.cvo_fe_read_stage.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.cvo_fe_random_pattern.argtypes = [C.c_int, u8p]
        L.cvo_fe_camera.argtypes = [C.c_int, fp]
        for name in SYMBOLS:
            if name != "cvo_fe_last_error":
                getattr(L, name).restype = C.c_int
        _BOUND = True
    return L


def random_pattern(n):
    """The selector's random bytes (ref thirdparty/PixelSelector2.cpp:35-37)."""
    out = np.empty(n, np.uint8)
    capi.check(lib().cvo_fe_random_pattern(n, out.ctypes.data_as(C.POINTER(C.c_uint8))), what="random_pattern")
    return out


def camera(dataset_seq):
    """{scaling_factor, fx, fy, cx, cy} of the reference's camera table
    (ref src/pcd_generator.cpp:241-295)."""
    cam = np.zeros(5, np.float32)
    capi.check(lib().cvo_fe_camera(int(dataset_seq), cam.ctypes.data_as(C.POINTER(C.c_float))), what="camera")
    return dict(zip(("scaling_factor", "fx", "fy", "cx", "cy"), (float(v) for v in cam)))


class PcdGenerator:
    """ref include/pcd_generator.hpp:30-108; one object per image size."""

    def __init__(self, width=640, height=480, device=0, stream=None, num_want=3000):
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        st = lib().cvo_fe_create(device, stream, self.width, self.height, C.byref(self._h))
        if st != 0:
            self._h = C.c_void_p()
            capi.check(st, what="cvo_fe_create (the front end needs a gfx950 device: no CPU path)")
        self.num_want = int(num_want)
        self._chk(lib().cvo_fe_set_num_want(self._h, self.num_want), "set_num_want")
        self.capacity = self.width * self.height   # (upper bound of any selection)
        self._pos = np.empty((self.capacity, 3), np.float32)
        self._feat = np.empty((self.capacity, 5), np.float32)

    def _chk(self, st, what):
        if st != 0:
            msg = lib().cvo_fe_last_error(self._h)
            raise capi.CvoHipError("%s: %s (%s)" % (what, capi.lib().cvo_hip_error_string(st).decode(),
                                                     msg.decode() if msg else ""))

    def create_pointcloud(self, bgr, depth, dataset_seq=1, feature_type=FEATURES_RGB):
        """load_image + create_pointcloud (ref src/pcd_generator.cpp:387-420): returns
        (positions n x 3, features n x 5 row-major), points in image scan order."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        if bgr.shape != (self.height, self.width, 3) or depth.shape != (self.height, self.width):
            raise ValueError("expected a %dx%dx3 uint8 image and a %dx%d uint16 depth map"
                             % (self.height, self.width, self.height, self.width))
        n = C.c_int(0)
        st = lib().cvo_fe_create_pointcloud(
            self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)), self.width * 3,
            depth.ctypes.data_as(C.POINTER(C.c_uint16)), self.width * 2, int(dataset_seq), int(feature_type),
            self._pos.ctypes.data_as(C.POINTER(C.c_float)), self._feat.ctypes.data_as(C.POINTER(C.c_float)),
            self.capacity, C.byref(n))
        self._chk(st, "create_pointcloud")
        return self._pos[:n.value].copy(), self._feat[:n.value].copy()

    def submit(self, bgr, depth, dataset_seq=1, feature_type=FEATURES_RGB):
        """First half of create_pointcloud: stage the images, enqueue everything, return."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        if bgr.shape != (self.height, self.width, 3) or depth.shape != (self.height, self.width):
            raise ValueError("expected a %dx%dx3 uint8 image and a %dx%d uint16 depth map"
                             % (self.height, self.width, self.height, self.width))
        self._chk(lib().cvo_fe_submit(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)), self.width * 3,
                                      depth.ctypes.data_as(C.POINTER(C.c_uint16)), self.width * 2,
                                      int(dataset_seq), int(feature_type)), "submit")

    def collect(self):
        """Second half: wait for the submitted frame and return its cloud."""
        n = C.c_int(0)
        self._chk(lib().cvo_fe_collect(self._h, self._pos.ctypes.data_as(C.POINTER(C.c_float)),
                                       self._feat.ctypes.data_as(C.POINTER(C.c_float)), self.capacity, C.byref(n)),
                  "collect")
        return self._pos[:n.value].copy(), self._feat[:n.value].copy()

    def collect_device(self):
        """collect() without the copy to the host: (device address of the positions, of the
        features, number of points); valid until the next submit on this object."""
        dp, df, n = C.c_void_p(), C.c_void_p(), C.c_int(0)
        self._chk(lib().cvo_fe_collect_device(self._h, C.byref(dp), C.byref(df), C.byref(n)), "collect_device")
        return dp.value, df.value, n.value

    def host_buffers(self):
        """numpy views of the context's pinned staging images (h x w x 3 uint8, h x w uint16): fill
        them in place and pass them to submit() / create_pointcloud() to save a copy."""
        pi, pd = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint16)()
        self._chk(lib().cvo_fe_host_buffers(self._h, C.byref(pi), C.byref(pd)), "host_buffers")
        img = np.ctypeslib.as_array(pi, shape=(self.height, self.width, 3))
        dep = np.ctypeslib.as_array(pd, shape=(self.height, self.width))
        return img, dep

    def set_device_output(self, on=True):
        """The following frames are taken with collect_device(): no copy of the cloud to the host."""
        self._chk(lib().cvo_fe_set_device_output(self._h, 1 if on else 0), "set_device_output")

    def info(self):
        out = Info()
        self._chk(lib().cvo_fe_get_info(self._h, C.byref(out)), "get_info")
        return {k: getattr(out, k) for k, _ in Info._fields_ if k != "pad_"}

    def read_stage(self, stage):
        """An intermediate image of the last create_pointcloud (parity checks)."""
        w, h = self.width, self.height
        shapes = {STAGE_GRAY: ((h, w), np.uint8), STAGE_HSV: ((h, w, 3), np.uint8), STAGE_MAP: ((h, w), np.float32),
                  STAGE_AG0: ((h, w), np.float32), STAGE_AG1: ((h // 2, w // 2), np.float32),
                  STAGE_AG2: ((h // 4, w // 4), np.float32), STAGE_THS: ((h // 32, w // 32), np.float32),
                  STAGE_DX0: ((h, w), np.float32), STAGE_DY0: ((h, w), np.float32), STAGE_EDGES: ((h, w), np.uint8)}
        shape, dt = shapes[stage]
        out = np.empty(shape, dt)
        self._chk(lib().cvo_fe_read_stage(self._h, stage, out.ctypes.data_as(C.c_void_p), out.nbytes), "read_stage")
        return out

    def close(self):
        if self._h:
            lib().cvo_fe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # pragma: no cover
            pass


# ---- the file side of the drivers -------------------------------------------------

def load_file_name(assoc_path):
    """The association list of a TUM sequence: lines `stamp_rgb rgb_path stamp_depth
    depth_path` -> (names, rgb paths, depth paths) (ref src/cvo_main.cpp:69-97)."""
    names, rgb, dep = [], [], []
    with open(assoc_path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            tok += [""] * (4 - len(tok))
            names.append(tok[0]); rgb.append(tok[1]); dep.append(tok[3])
    return names, rgb, dep


def load_img(rgb_path, depth_path):
    """cv::imread(rgb) -> h x w x 3 uint8 in B, G, R order; cv::imread(depth, ANYDEPTH)
    -> h x w uint16 (ref src/cvo_main.cpp:100-106).  Decoding is PIL's."""
    from PIL import Image
    rgb = np.asarray(Image.open(rgb_path).convert("RGB"), np.uint8)
    dep = np.asarray(Image.open(depth_path))
    if dep.dtype != np.uint16:
        dep = dep.astype(np.uint16)
    return np.ascontiguousarray(rgb[:, :, ::-1]), np.ascontiguousarray(dep)


def run_frames(registration, frames, dataset_seq, writer=None, generator=None, prefetch=False, device=True):
    """The driver loop on decoded frames: `frames` yields (name, bgr, depth).
    `device`: the cloud goes from the front end to the registration in device memory
    (cvo_fe_collect_device -> cvo_hip_set_*_device) instead of through host arrays.
    `prefetch`: frame k+1 is in the front end (its own, low-priority stream) while frame k
    is being registered; the results are the same either way.  Off by default: measured on
    MI355X it is worth +4-5 % per frame in a process that has little else on the GPU's queues
    and -5 % in one that has (bench.py, after the batched legs).  Returns the number of frames."""
    ftype = FEATURES_HSV if registration.params.mode == capi.MODE_ACVO else FEATURES_RGB
    gen = generator
    it = iter(frames)
    count = 0
    cur = next(it, None)
    if cur is None:
        return 0
    if gen is None:
        gen = PcdGenerator(cur[1].shape[1], cur[1].shape[0])
    gen.set_device_output(device)
    gen.submit(cur[1], cur[2], dataset_seq, ftype)
    while cur is not None:
        if device:
            d_xyz, d_feat, npts = gen.collect_device()
            nxt = next(it, None)
            first = not registration.init
            registration.set_pcd_device(d_xyz, d_feat, npts)   # (consumed: the next frame may overwrite it)
            if nxt is not None and prefetch:
                gen.submit(nxt[1], nxt[2], dataset_seq, ftype)
            if not first:
                registration.align()
        else:
            xyz, feat = gen.collect()
            nxt = next(it, None)
            if nxt is not None and prefetch:
                gen.submit(nxt[1], nxt[2], dataset_seq, ftype)
            registration.run_cvo(xyz, feat)
        if writer is not None and registration.init:
            writer.append(cur[0], registration.accum_transform)
        if nxt is not None and not prefetch:
            gen.submit(nxt[1], nxt[2], dataset_seq, ftype)
        count += 1
        cur = nxt
    return count


def run_directory(registration, folder, dataset_seq, writer=None, assoc="assoc.txt", limit=None,
                  generator=None):
    """The reference's main loop (ref src/cvo_main.cpp:20-66, adaptive_cvo_main.cpp): every
    frame of `folder`/assoc goes through the front end and `run_cvo`; a pose line per
    frame is handed to `writer` (trajectory.TrajectoryWriter).  cvo uses the raw colour
    features, acvo the HSV ones (ref src/cvo.cpp:329, src/adaptive_cvo.cpp:451)."""
    names, rgbs, deps = load_file_name(os.path.join(folder, assoc))
    if limit is not None:
        names, rgbs, deps = names[:limit], rgbs[:limit], deps[:limit]

    def decoded():
        for name, r, d in zip(names, rgbs, deps):
            bgr, depth = load_img(os.path.join(folder, r), os.path.join(folder, d))
            yield name, bgr, depth

    return run_frames(registration, decoded(), dataset_seq, writer=writer, generator=generator)
