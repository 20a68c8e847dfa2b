"""ctypes binding of oracle/liboracle_fe.so -- the CPU restatement of the RGB-D
front end (oracle/frontend_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/ and __graft_entry__.smoke() --
never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_fe.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_fe.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.fe_create_pointcloud.restype = C.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def random_pattern(n):
    """srand(3141592); rand() & 0xFF -- the C library's own generator."""
    out = np.empty(n, np.uint8)
    lib().fe_random_pattern(C.c_int(n), _p(out, C.c_uint8))
    return out


def gray(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().fe_gray(_p(img, C.c_uint8), C.c_int(w), C.c_int(h), _p(out, C.c_uint8))
    return out


def hsv(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    lib().fe_hsv(_p(img, C.c_uint8), C.c_int(w), C.c_int(h), _p(out, C.c_uint8))
    return out


def pyramid(gray_img):
    """-> (I[3], dx0, dy0, ag[3])"""
    g = np.ascontiguousarray(gray_img, np.uint8)
    h, w = g.shape
    I = [np.zeros((h >> l, w >> l), np.float32) for l in range(3)]
    ag = [np.zeros((h >> l, w >> l), np.float32) for l in range(3)]
    dx0 = np.zeros((h, w), np.float32)
    dy0 = np.zeros((h, w), np.float32)
    FP = C.POINTER(C.c_float)
    Iarr = (FP * 3)(*[_p(a, C.c_float) for a in I])
    Aarr = (FP * 3)(*[_p(a, C.c_float) for a in ag])
    lib().fe_pyramid(_p(g, C.c_uint8), C.c_int(w), C.c_int(h), Iarr, _p(dx0, C.c_float), _p(dy0, C.c_float), Aarr)
    return I, dx0, dy0, ag


def thresholds(ag0):
    a = np.ascontiguousarray(ag0, np.float32)
    h, w = a.shape
    out = np.zeros((h // 32, w // 32), np.float32)
    lib().fe_thresholds(_p(a, C.c_float), C.c_int(w), C.c_int(h), _p(out, C.c_float))
    return out


def blur3(g):
    g = np.ascontiguousarray(g, np.uint8)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().fe_blur3(_p(g, C.c_uint8), C.c_int(w), C.c_int(h), _p(out, C.c_uint8))
    return out


def canny(g, low=0, high=25):
    g = np.ascontiguousarray(g, np.uint8)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().fe_canny(_p(g, C.c_uint8), C.c_int(w), C.c_int(h), C.c_int(low), C.c_int(high), _p(out, C.c_uint8))
    return out


def camera(dataset_seq):
    cam = np.zeros(5, np.float32)
    lib().fe_camera(C.c_int(dataset_seq), _p(cam, C.c_float))
    return cam


def create_pointcloud(img, depth, dataset_seq=1, feature_type=1, num_want=3000):
    """-> dict(positions, features, map, num_selected)"""
    img = np.ascontiguousarray(img, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    h, w = depth.shape
    cap = w * h
    pos = np.zeros((cap, 3), np.float32)
    feat = np.zeros((cap, 5), np.float32)
    mp = np.zeros((h, w), np.float32)
    nsel = C.c_int(0)
    n = lib().fe_create_pointcloud(_p(img, C.c_uint8), _p(depth, C.c_uint16), C.c_int(w), C.c_int(h),
                                   C.c_int(dataset_seq), C.c_int(feature_type), C.c_int(num_want),
                                   _p(pos, C.c_float), _p(feat, C.c_float), C.c_int(cap), _p(mp, C.c_float),
                                   C.byref(nsel))
    return {"positions": pos[:n].copy(), "features": feat[:n].copy(), "map": mp, "num_selected": nsel.value}
