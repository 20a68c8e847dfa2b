"""Oracle (test infrastructure only): the cloud preparation of the reference's MATLAB driver in numpy --
what cvo_hip_range_filter_grid_average (csrc/cvo_prep.hip) is held to, bit for bit.

  pc_range_filter  ref util/pcRangeFilter.m:5-12
  grid_average     ref data/rgbd_dataset/rgbddataset_rkhs.m:36-39,58  (pcdownsample(..., 'gridAverage', gridSize))

Parity unpinned against MATLAB itself: pcdownsample's voxel anchoring is not documented and no anchoring
reproduces the transforms the reference's run recorded (tools/search_grid_anchor.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu legs may import this module."""
import numpy as np


def pc_range_filter(xyz, rgb, max_range=4.0, min_range=0.8):
    """Drop the points whose range (float32 norm) is above max_range or below min_range."""
    xyz = np.asarray(xyz, np.float32)
    r = np.sqrt((xyz * xyz).sum(1, dtype=np.float32))
    keep = ~((r > np.float32(max_range)) | (r < np.float32(min_range))) & np.isfinite(xyz).all(1)
    return xyz[keep], np.asarray(rgb)[keep]


def grid_average(xyz, rgb, grid_size=0.05):
    """One point per occupied voxel = the mean location and the mean colour (rounded to uint8) of its
    points; voxels anchored at the cloud's minimum corner, in lexicographic (x, y, z) index order."""
    x = np.asarray(xyz, np.float64)
    c = np.asarray(rgb, np.float64)
    ok = np.isfinite(x).all(1)   # (invalid points are dropped, as pcdownsample does)
    x, c = x[ok], c[ok]
    if x.shape[0] == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
    idx = np.floor((x - x.min(0)) / float(grid_size)).astype(np.int64)
    span = idx.max(0) + 1
    key = (idx[:, 0] * span[1] + idx[:, 1]) * span[2] + idx[:, 2]
    _, inv = np.unique(key, return_inverse=True)
    inv = inv.ravel()
    n = int(inv.max()) + 1
    cnt = np.bincount(inv, minlength=n).astype(np.float64)
    loc = np.stack([np.bincount(inv, weights=x[:, k], minlength=n) / cnt for k in range(3)], 1)
    col = np.stack([np.bincount(inv, weights=c[:, k], minlength=n) / cnt for k in range(c.shape[1])], 1)
    return loc.astype(np.float32), np.clip(np.floor(col + 0.5), 0, 255).astype(np.uint8)
