"""Pose-only optimisation (IndirectCameraOptimizer) test inputs and the ctypes wrapper of the oracle restatement."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O


def scene(n=600, seed=5, outlier_fraction=0.1, noise_px=0.5, rot=0.03, trans=0.08, K=(718.856, 718.856, 607.19, 185.22), wh=(1241, 376)):
    """n map points seen by a camera at (R_true, t_true); the optimisation starts `rot` rad / `trans` units away.
    Gross outliers: observations moved by 20-80 px.  Information as the LM overload forms it (1 / descriptor distance for the
    edge, 1 / scaleFactor^2 for the outlier test, IndirectCameraOptimizer.cpp:86-89)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    R_true = synth.so3_exp(rng.normal(0, 0.2, 3)); t_true = rng.normal(0, 0.5, 3)
    u = rng.uniform(20, wh[0] - 20, n); v = rng.uniform(20, wh[1] - 20, n); z = rng.uniform(4, 40, n)
    Pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (Pc - t_true) @ R_true                       # R^T (Pc - t)
    m = np.zeros(n, abi.PNP_MATCH_DTYPE)
    m["X"] = Xw
    obs = np.stack([u, v], 1) + rng.normal(0, noise_px, (n, 2))
    planted = rng.uniform(size=n) < outlier_fraction
    ang = rng.uniform(0, 2 * np.pi, n); mag = rng.uniform(20, 80, n)
    obs[planted] += (np.stack([np.cos(ang), np.sin(ang)], 1) * mag[:, None])[planted]
    m["obs"] = obs
    level = rng.integers(0, 8, n)
    m["info"] = 1.0 / (1.2 ** level) ** 2
    m["inv_sigma2"] = 1.0 / rng.integers(10, 60, n)
    R0 = synth.so3_exp(rng.normal(0, 1, 3) * rot / np.sqrt(3)) @ R_true
    t0 = t_true + rng.normal(0, 1, 3) * trans / np.sqrt(3)
    return dict(K=np.array(K, np.float64), R_true=R_true, t_true=t_true, R0=R0, t0=t0, matches=m, planted=planted)


def oracle_pnp(R0, t0, K, matches, outliers, algorithm=0, check_outliers=True, compute_covariance=False):
    """orc_pnp_optimize; outliers (uint8) is updated in place."""
    L = O.lib()
    L.orc_pnp_optimize.restype = None
    out = abi.PnpResult()
    R0 = np.ascontiguousarray(R0, np.float64); t0 = np.ascontiguousarray(t0, np.float64); K = np.ascontiguousarray(K, np.float64)
    m = np.ascontiguousarray(matches)
    L.orc_pnp_optimize(O.ptr(R0, C.c_double), O.ptr(t0, C.c_double), O.ptr(K, C.c_double), len(m), C.c_void_p(m.ctypes.data),
                       O.ptr(outliers, C.c_ubyte), int(algorithm), int(bool(check_outliers)), int(bool(compute_covariance)), C.byref(out))
    return out
