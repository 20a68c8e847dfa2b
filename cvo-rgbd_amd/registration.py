"""Python mirror of the reference's registration objects.

``Cvo`` ~ cvo::cvo (ref cpp/rkhs_registration/include/cvo.hpp:55-193),
``Acvo`` ~ acvo::acvo (ref include/adaptive_cvo.hpp:57-196): same members
(init, iter, transform, prev_transform, accum_transform) and methods
(set_pcd, align, run_cvo), the point cloud arriving as arrays instead of
cv::Mat images (the image front end is outside this back end).
All computation goes through the HIP C-ABI; nothing here has a CPU path.
"""
import numpy as np

from . import capi


class _Registration:
    MODE = capi.MODE_CVO

    def __init__(self, device=0, stream=None, params=None, graph_capture=None):
        self.params = params if params is not None else capi.default_params(self.MODE)
        self.ctx = capi.Context(self.params, device=device, stream=stream, graph_capture=graph_capture)
        self.state = capi.init_state(self.params)
        self.init = False
        self.iter = 0
        self.transform = np.eye(4, dtype=np.float32)
        self.prev_transform = np.eye(4, dtype=np.float32)
        self.accum_transform = np.eye(4, dtype=np.float32)
        self.num_iterations = 0
        self.trace = []
        self._have_moving = False

    def _publish(self):
        s = self.state
        self.transform = np.array(s.transform, np.float32).reshape(4, 4)
        self.prev_transform = np.array(s.prev_transform, np.float32).reshape(4, 4)
        self.accum_transform = np.array(s.accum_transform, np.float32).reshape(4, 4)
        self.iter = int(s.iter)

    def set_pcd(self, positions, features, layout=capi.FEAT_ROWMAJOR):
        """ref src/cvo.cpp:319-357 (tail: hand the clouds to the back end)."""
        if not self.init:
            self.ctx.set_fixed(positions, features, layout)
            self.init = True
            return
        self.ctx.set_moving(positions, features, layout)
        self._have_moving = True

    def set_pcd_device(self, d_positions, d_features, n, layout=capi.FEAT_ROWMAJOR):
        """set_pcd with the cloud already in device memory (e.g. PcdGenerator.collect_device)."""
        if not self.init:
            self.ctx.set_fixed_device(d_positions, d_features, n, layout)
            self.init = True
            return
        self.ctx.set_moving_device(d_positions, d_features, n, layout)
        self._have_moving = True

    def run_cvo_device(self, d_positions, d_features, n, layout=capi.FEAT_ROWMAJOR, trace_cap=0):
        first = not self.init
        self.set_pcd_device(d_positions, d_features, n, layout)
        if not first:
            self.align(trace_cap=trace_cap)

    def align(self, trace_cap=0):
        """ref src/cvo.cpp:361-420."""
        if not self._have_moving:
            raise capi.CvoHipError("align(): set_pcd() must precede each align()")
        self.num_iterations, self.trace = self.ctx.align(self.state, trace_cap=trace_cap)
        self.ctx.swap_moving_to_fixed()   # ptr_fixed_pcd = std::move(ptr_moving_pcd)
        self._have_moving = False
        self._publish()

    def run_cvo(self, positions, features, layout=capi.FEAT_ROWMAJOR, trace_cap=0):
        """ref src/cvo.cpp:422-435."""
        if not self.init:
            self.set_pcd(positions, features, layout)
        else:
            self.set_pcd(positions, features, layout)
            self.align(trace_cap=trace_cap)

    def run_sequence(self, frames, writer=None, trace_cap=0):
        """The loop of the reference's drivers (ref src/cvo_main.cpp:36-66): every
        frame goes through run_cvo() and then gets a pose line of `accum_transform`
        in `writer` (a trajectory.TrajectoryWriter) -- the first frame too (the
        identity): `init` is already true after the first run_cvo()
        (ref cvo_main.cpp:52,58).  `frames` yields (name,
        positions, features).  Returns the per-pair iteration counts."""
        iters = []
        for name, positions, features in frames:
            first = not self.init
            self.run_cvo(positions, features, trace_cap=trace_cap)
            if not first:
                iters.append(self.num_iterations)
            if writer is not None and self.init:
                writer.append(name, self.accum_transform)
        return iters

    def close(self):
        self.ctx.close()


class Cvo(_Registration):
    MODE = capi.MODE_CVO


class RkhsMatlab(_Registration):
    """The reference's MATLAB registration object (ref matlab/@rkhs_se3_registration/
    rkhs_se3_registration.m) on the same HIP kernels: linear colour inner
    product CI = 1e-5 <c_i, c_j>, squared-exponential kernel thresholded at 1e-3 on K
    alone, eps 5e-4 / 1e-4 (:10-28,40-73,125-127).  Unlike the C++ objects it starts every
    pair from R = I, T = 0, ell = 0.15 (:112-114).  Arithmetic is this library's float32
    per-pair contract, not MATLAB's float64: results agree with a float64 restatement
    (oracle/matlab_dense.py) to ~1e-5."""
    MODE = capi.MODE_MATLAB

    @staticmethod
    def features(rgb):
        """n x 3 colour bytes -> the n x 5 feature rows the kernels read (channels 0..2)."""
        f = np.zeros((len(rgb), 5), np.float32)
        f[:, :3] = np.asarray(rgb, np.float32)
        return f

    def register(self, fixed_xyz, fixed_rgb, moving_xyz, moving_rgb):
        """One pair as rgbddataset_rkhs.m drives the object (ref :30-75): returns the 4 x 4
        `tform` = [R' -R'T; 0 1] and the number of iterations."""
        self.state = capi.init_state(self.params)
        self.ctx.set_fixed(np.ascontiguousarray(fixed_xyz, np.float32), self.features(fixed_rgb))
        self.ctx.set_moving(np.ascontiguousarray(moving_xyz, np.float32), self.features(moving_rgb))
        self.init = True
        self._have_moving = True
        self.num_iterations, self.trace = self.ctx.align(self.state, trace_cap=0)
        self._have_moving = False
        self._publish()
        return self.transform.copy(), self.num_iterations


class Acvo(_Registration):
    MODE = capi.MODE_ACVO

    def function_inner_product(self, cloud_a, cloud_b, layout=capi.FEAT_ROWMAJOR):
        """ref include/adaptive_cvo.hpp:179, src/adaptive_cvo.cpp:385-439: the statistic between
        two arbitrary clouds -- (positions, features) each -- at the current length-scale.
        Registration state is untouched (a pending set_pcd() stays pending)."""
        return self.ctx.function_inner_product_clouds(self.state.ell, cloud_a[0], cloud_a[1],
                                                      cloud_b[0], cloud_b[1], layout)
