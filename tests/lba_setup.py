"""Local bundle adjustment (IndirectBundleAdjustment) test inputs and the ctypes wrapper of the oracle restatement."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O


def scene(n_local=6, n_fixed=4, n_points=800, seed=2, noise_px=0.5, point_noise=0.05, outlier_fraction=0.03,
          K=(718.856, 718.856, 607.19, 185.22), wh=(1241, 376), pose_noise=0.0):
    """A forward-moving rig: n_local + n_fixed keyframes, points in front of the middle of the trajectory; a point is observed
    by the frames it projects into (at least 2).  Point estimates start `point_noise` (relative depth) off the truth; a few
    observations are gross outliers.  Frames [0, n_local) are the local keyframes, the rest are fixed."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    nF = n_local + n_fixed
    frames = np.zeros(nF, abi.LBA_FRAME_DTYPE)
    Rs, ts = [], []
    for f in range(nF):
        R = synth.so3_exp(rng.normal(0, 0.02, 3))
        c = np.array([0.15 * rng.normal(), 0.05 * rng.normal(), 0.6 * f])        # camera centre
        Rs.append(R); ts.append(-R @ c)
        frames["R"][f] = R.ravel(); frames["t"][f] = ts[-1]; frames["K"][f] = K
        frames["fixed"][f] = 0 if f < n_local else 1
    mid = 0.6 * (nF - 1) / 2
    pts, offs, edges, planted = [], [0], [], []
    while len(pts) < n_points:
        X = np.array([rng.uniform(-8, 8), rng.uniform(-3, 3), mid + rng.uniform(5, 40)])
        obs = []
        for f in range(nF):
            p = Rs[f] @ X + ts[f]
            if p[2] < 1:
                continue
            u, v = fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy
            if 10 < u < wh[0] - 10 and 10 < v < wh[1] - 10 and rng.uniform() < 0.8:
                o = np.array([u, v]) + rng.normal(0, noise_px, 2)
                bad = rng.uniform() < outlier_fraction
                if bad:
                    a = rng.uniform(0, 2 * np.pi); o = o + rng.uniform(15, 60) * np.array([np.cos(a), np.sin(a)])
                obs.append((f, o, bad))
        if len(obs) < 2:
            continue
        pts.append(X)
        for f, o, bad in obs:
            edges.append((f, o, 1.0 / (1.2 ** rng.integers(0, 8)) ** 2)); planted.append(bad)
        offs.append(len(edges))
    E = np.zeros(len(edges), abi.LBA_EDGE_DTYPE)
    E["frame"] = [e[0] for e in edges]; E["obs"] = [e[1] for e in edges]; E["inv_sigma2"] = [e[2] for e in edges]
    truth = np.array(pts)
    cen = np.array([0, 0, mid])
    start = cen + (truth - cen) * (1 + rng.normal(0, point_noise, (len(pts), 1)))      # depth error along the viewing ray
    frames_true = frames.copy()
    if pose_noise > 0:                                  # the local keyframes start away from where the observations were made
        for f in range(n_local):
            dR = synth.so3_exp(rng.normal(0, pose_noise * 0.2, 3))
            R = dR @ Rs[f]
            frames["R"][f] = R.ravel(); frames["t"][f] = ts[f] + rng.normal(0, pose_noise, 3)
    return dict(frames=frames, frames_true=frames_true, truth=truth, points=np.ascontiguousarray(start), off=np.array(offs, np.int32), edges=E,
                planted=np.array(planted), K=np.array(K))


def oracle_lba(frames, points, off, edges, fix_frames=True, num_iterations=5, refine_iterations=0):
    """orc_lba_optimize on copies; returns (frames, points, edge_bad, result)."""
    L = O.lib()
    fr = np.ascontiguousarray(frames.copy()); pts = np.ascontiguousarray(points.copy()); off = np.ascontiguousarray(off, np.int32)
    ed = np.ascontiguousarray(edges)
    bad = np.zeros(len(ed), np.uint8)
    out = abi.LbaResult()
    rc = L.orc_lba_optimize(len(fr), C.c_void_p(fr.ctypes.data), len(pts), O.ptr(pts, C.c_double), O.ptr(off, C.c_int), C.c_void_p(ed.ctypes.data),
                            int(bool(fix_frames)), int(num_iterations), int(refine_iterations), O.ptr(bad, C.c_ubyte), C.byref(out))
    assert rc == 0, rc
    return fr, pts, bad, out
