"""cvo-rgbd_amd -- MI355X (gfx950) back end for the CVO / Adaptive-CVO inner loop.

The directory name carries a hyphen (it mirrors the upstream repo name), so the
package is imported under the module name ``cvo_rgbd_amd`` through
``__graft_entry__.load_package()``.

Contents: ``csrc/`` (HIP kernels + the C-ABI of include/cvo_hip.h, built into
csrc/libcvo_hip.so), ``capi`` (ctypes binding of that ABI), ``registration``
(Python mirror of the reference's cvo::cvo / acvo::acvo objects) and ``data``
(cloud formats either side of the path: synthetic clouds, PCD reader,
trajectory writer) and ``frontend`` (RGB-D image pair -> point cloud, HIP kernels
behind include/cvo_frontend.h).  There is no CPU fallback: without the built library and a
HIP device every compute call raises.
"""
from . import capi, data, frontend, registration, trajectory  # noqa: F401
from .registration import Acvo, Cvo, RkhsMatlab  # noqa: F401

__all__ = ["capi", "data", "frontend", "registration", "trajectory", "Cvo", "Acvo", "RkhsMatlab"]
