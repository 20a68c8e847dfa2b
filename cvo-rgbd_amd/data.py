"""Cloud formats either side of the hot path.

* ``synthetic_pair`` -- the seeded synthetic RGB-D surface clouds every
  BASELINE.json config except the TUM ones is quoted on.
* ``read_pcd_ascii`` -- the MATLAB ``pcwrite`` ASCII layout of the shipped
  fr1/desk clouds (ref data/rgbd_dataset/freiburg1_desk/pcd_ds/*.pcd:1-13).
* ``cvo_features`` / ``acvo_features`` -- the 5-dim feature conventions of the
  front end (ref src/pcd_generator.cpp:336-380).
* ``pose_line`` -- trajectory text line of the drivers (ref src/cvo_main.cpp:58-65).
"""
import numpy as np

# ground-truth motion of the synthetic pairs
GT_AXIS = np.array([0.6, -0.3, 0.74])
GT_ANGLE = 0.02
GT_TRANS = np.array([0.004, 0.003, -0.009])

SEED_CFG2 = 20190402   # 10k x 10k
SEED_CFG4 = 20191001   # 200k x 200k
SEED_CFG5_BASE = 1000  # + pair id, 20k x 20k


def _rodrigues(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def gt_motion():
    """4x4 ground-truth motion used to displace the synthetic moving cloud."""
    M = np.eye(4)
    M[:3, :3] = _rodrigues(GT_AXIS, GT_ANGLE)
    M[:3, 3] = GT_TRANS
    return M


def _surface_cloud(rng, n):
    x = rng.uniform(-0.79, 0.78, n)
    y = rng.uniform(-0.76, 0.65, n)
    z = 1.3 + 0.25 * np.sin(2.1 * x + 0.3) * np.cos(1.7 * y) + 0.15 * np.sin(4.3 * y)
    xyz = np.stack([x, y, z], 1) + rng.normal(0.0, 0.002, (n, 3))
    B = 127 + 100 * np.sin(3 * x)
    G = 127 + 100 * np.cos(2.5 * y)
    R = 127 + 80 * np.sin(2 * x + 2 * y)
    dx = 30 * np.cos(3 * x) + rng.normal(0.0, 3.0, n)
    dy = -25 * np.sin(2.5 * y) + rng.normal(0.0, 3.0, n)
    feat = np.stack([B, G, R, dx, dy], 1)
    return xyz, feat


def synthetic_pair(n, m, seed, acvo=False):
    """Fixed cloud (n) and moving cloud (m): independent samples of one coloured
    surface; the moving one displaced by the inverse of ``gt_motion()``.
    Returns float32 (xyz_fixed, feat_fixed, xyz_moving, feat_moving), features
    row-major n x 5 in the cvo scale (raw B,G,R,dx,dy) or, with ``acvo=True``,
    the acvo scale (/180,/255,/255,/255*2,/255*2)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xf, ff = _surface_cloud(rng, n)
    xm, fm = _surface_cloud(rng, m)
    Minv = np.linalg.inv(gt_motion())
    xm = xm @ Minv[:3, :3].T + Minv[:3, 3]
    if acvo:
        scale = np.array([1 / 180.0, 1 / 255.0, 1 / 255.0, 2 / 255.0, 2 / 255.0])
        ff, fm = ff * scale, fm * scale
    return (xf.astype(np.float32), ff.astype(np.float32), xm.astype(np.float32),
            fm.astype(np.float32))


def read_pcd_ascii(path):
    """Reads a MATLAB-pcwrite ASCII PCD (FIELDS x y z rgb).  Returns
    (xyz float32 n x 3, rgb uint8 n x 3 as R,G,B)."""
    with open(path, "r") as fh:
        fields, n_header = None, 0
        for line in fh:
            n_header += 1
            tok = line.split()
            if tok and tok[0] == "FIELDS":
                fields = tok[1:]
            if tok and tok[0] == "DATA":
                if tok[1] != "ascii":
                    raise ValueError("only ASCII PCD is supported")
                break
        if fields is None or fields[:3] != ["x", "y", "z"]:
            raise ValueError("unexpected PCD fields: %r" % (fields,))
        arr = np.loadtxt(fh, dtype=np.float64, ndmin=2)
    xyz = arr[:, :3].astype(np.float32)   # MATLAB pcread returns single
    if "rgb" in fields:
        packed = arr[:, fields.index("rgb")].astype(np.float32).view(np.uint32)
        rgb = np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], 1).astype(np.uint8)
    else:
        rgb = np.zeros((xyz.shape[0], 3), np.uint8)
    return xyz, rgb


def prepare_matlab_cloud(xyz, rgb, max_range=4.0, min_range=0.8, grid_size=0.05, device=0):
    """The cloud preparation of the reference's MATLAB driver (ref data/rgbd_dataset/rgbddataset_rkhs.m:
    33-39,55-58): pcRangeFilter, then pcdownsample(..., 'gridAverage', grid_size) -- on the GPU
    (cvo_hip_range_filter_grid_average, csrc/cvo_prep.hip; no CPU path).  max_range <= 0: no range
    filter; grid_size <= 0: no downsampling.  Returns (xyz float32 m x 3, rgb uint8 m x 3)."""
    from . import capi
    return capi.range_filter_grid_average(xyz, rgb, max_range, min_range, grid_size, device=device)


def pc_range_filter(xyz, rgb, max_range=4.0, min_range=0.8, device=0):
    """ref util/pcRangeFilter.m:5-12: drop the points whose range (float32 norm) is above max_range or
    below min_range (GPU)."""
    return prepare_matlab_cloud(xyz, rgb, max_range, min_range, 0.0, device=device)


def grid_average(xyz, rgb, grid_size=0.05, device=0):
    """Box-grid downsampling in the manner of MATLAB's pcdownsample(cloud, 'gridAverage', grid_size)
    (ref data/rgbd_dataset/rgbddataset_rkhs.m:36-39,58): one point per occupied voxel = the mean
    location and the mean colour (rounded to uint8) of its points (GPU).  Voxels are anchored at the
    cloud's minimum corner; MATLAB's own anchoring is not documented, and no anchoring reproduces the
    transforms its run recorded better than 2.8e-3 (tools/search_grid_anchor.py,
    tests/golden/grid_anchor_residuals.json), so the output is equivalent, not
    identical.  Voxels come out in lexicographic (x, y, z) index order."""
    return prepare_matlab_cloud(xyz, rgb, 0.0, 0.0, grid_size, device=device)


def cvo_features(rgb, dx=None, dy=None):
    """cvo feature type 1: raw B, G, R, dx, dy (ref src/pcd_generator.cpp:359-380)."""
    n = rgb.shape[0]
    f = np.zeros((n, 5), np.float32)
    f[:, 0], f[:, 1], f[:, 2] = rgb[:, 2], rgb[:, 1], rgb[:, 0]
    if dx is not None:
        f[:, 3] = dx
    if dy is not None:
        f[:, 4] = dy
    return f


def acvo_features(rgb, dx=None, dy=None):
    """acvo feature type 0: H/180, S/255, V/255, dx/255*2, dy/255*2 with OpenCV's
    8-bit HSV convention (H in [0,180)) (ref src/pcd_generator.cpp:336-357)."""
    r, g, b = [rgb[:, k].astype(np.float64) for k in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    diff = v - mn
    s = np.where(v > 0, 255.0 * diff / np.where(v > 0, v, 1), 0.0)
    safe = np.where(diff > 0, diff, 1)
    h = np.where(v == r, 60.0 * (g - b) / safe,
                 np.where(v == g, 120.0 + 60.0 * (b - r) / safe, 240.0 + 60.0 * (r - g) / safe))
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 360.0, h)
    h8 = np.rint(h / 2.0) % 180           # cv::COLOR_BGR2HSV on 8-bit images
    s8 = np.rint(s)
    n = rgb.shape[0]
    f = np.zeros((n, 5), np.float32)
    f[:, 0], f[:, 1], f[:, 2] = h8 / 180.0, s8 / 255.0, v / 255.0
    if dx is not None:
        f[:, 3] = np.asarray(dx) / 255.0 * 2
    if dy is not None:
        f[:, 4] = np.asarray(dy) / 255.0 * 2
    return f


def quaternion_xyzw(R):
    """Unit quaternion (x, y, z, w) of a rotation matrix (Eigen convention)."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        x, y, z = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * s
        s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
        x, y, z = q
    return np.array([x, y, z, w])


def pose_line(stamp, accum_transform):
    """`name tx ty tz qx qy qz qw` (ref src/cvo_main.cpp:58-65)."""
    M = np.asarray(accum_transform, np.float64)
    q = quaternion_xyzw(M[:3, :3])
    vals = [M[0, 3], M[1, 3], M[2, 3], q[0], q[1], q[2], q[3]]
    return "%s %s" % (stamp, " ".join("%g" % v for v in vals))


def rel_pose_error(T_est, T_ref):
    """(rotation error / reference angle, translation error / reference norm):
    the parity metric of BASELINE.json / SURVEY 8d."""
    T_est, T_ref = np.asarray(T_est, np.float64), np.asarray(T_ref, np.float64)
    def angle(R):   # atan2 form: exact 0 for a symmetric R, no sqrt(eps) blow-up near I
        w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        return np.arctan2(np.linalg.norm(w), (np.trace(R) - 1) / 2)
    ang = angle(T_est[:3, :3] @ T_ref[:3, :3].T)
    ref_ang = angle(T_ref[:3, :3])
    dt = np.linalg.norm(T_est[:3, 3] - T_ref[:3, 3])
    ref_t = np.linalg.norm(T_ref[:3, 3])
    return ang / max(ref_ang, 1e-30), dt / max(ref_t, 1e-30)


def synthetic_rgbd_frame(width=640, height=480, seed=0, texture=1.0, motion=(0.0, 0.0), holes=0.02):
    """A synthetic RGB-D frame for the front end (SURVEY 8 f3 has no image data to
    test on: the reference ships no frames): a smooth colour field with band-limited
    texture of strength `texture` (0 = nearly flat, few gradients; 1 = desk-like;
    3 = busy), shifted by `motion` pixels, and a slanted-plane depth map in TUM units
    (5000 per metre) with a fraction `holes` of invalid (zero) pixels.
    Returns (bgr h x w x 3 uint8, depth h x w uint16)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    x = x + motion[0]
    y = y + motion[1]
    u, v = x / width, y / height
    base = np.stack([120 + 70 * np.sin(5.0 * u + 1.0) * np.cos(3.0 * v),
                     110 + 60 * np.cos(4.0 * v + 0.5),
                     130 + 50 * np.sin(3.0 * u + 4.0 * v)], axis=-1)
    tex = np.zeros((height, width))
    for _ in range(24):   # a handful of oriented gratings and blobs
        fx, fy = rng.uniform(-0.35, 0.35, 2)
        ph = rng.uniform(0, 2 * np.pi)
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        sig = rng.uniform(30, 160)
        win = np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sig * sig))
        tex += rng.uniform(4, 18) * win * np.sign(np.sin(fx * x + fy * y + ph))
    img = base + texture * tex[..., None] * np.array([1.0, 0.8, 0.6])
    img += rng.normal(0, 0.6 * min(texture, 1.0) + 0.05, img.shape)
    bgr = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    z = 1.2 + 0.5 * u + 0.3 * v + 0.05 * np.sin(9 * u)
    depth = np.clip(np.rint(z * 5000.0), 0, 65535).astype(np.uint16)
    depth[rng.random((height, width)) < holes] = 0
    return bgr, depth
