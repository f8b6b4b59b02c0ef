"""SURVEY §8 f4 on the device: IndirectCameraOptimizer::optimize (g2o Levenberg / Gauss-Newton, 4 rounds x 10 iterations,
Huber sqrt(5.991), evaluateOutliers) as one launch, through the C ABI against the oracle.
Bar: the outlier flags and the round count IDENTICAL; pose, chi2 and covariance within 1e-9 relative — fp64 throughout, the
device sums the edges on the matrix cores in a different order than the edge-order loop of g2o.  The solve() count of a
round is reported but only compared loosely: g2o has no convergence test, so once a round has converged the remaining
iterations take steps whose chi2 change is rounding noise (|rho| ~ 1e-13) and whether one of them hits `rho == 0` or ten
rejections in a row (Terminate) depends on the last bit of the sums."""
import numpy as np
import pytest

from libcml_amd import abi, device
from tests import pnp_setup as PS

pytestmark = pytest.mark.gpu


def _compare(ro, rd, oo, od, tol=1e-9):
    assert ro.is_ok == rd.is_ok and ro.rounds == rd.rounds and ro.n_bad == rd.n_bad
    assert all(1 <= d <= 10 for d in list(rd.lm_iterations)[:rd.rounds])
    assert np.array_equal(oo, od)
    Ro = np.array(list(ro.R)); Rd = np.array(list(rd.R)); to = np.array(list(ro.t)); td = np.array(list(rd.t))
    assert np.abs(Ro - Rd).max() < tol and np.abs(to - td).max() < tol * max(1.0, np.abs(to).max())
    co = np.array(list(ro.chi2)); cd = np.array(list(rd.chi2))
    assert np.abs(co - cd).max() <= tol * max(1.0, np.abs(co).max())
    vo = np.array(list(ro.covariance)); vd = np.array(list(rd.covariance))
    assert np.abs(vo - vd).max() <= 1e-8 * max(1e-30, np.abs(vo).max())


@pytest.mark.parametrize("algorithm", [abi.PNP_LEVENBERG, abi.PNP_GAUSS_NEWTON])
@pytest.mark.parametrize("n,seed,outlier_fraction,rot", [(600, 5, 0.1, 0.03), (150, 7, 0.3, 0.1), (2560, 9, 0.05, 0.02), (40, 11, 0.0, 0.3), (257, 13, 0.2, 0.05)])
def test_pnp_matches_oracle(algorithm, n, seed, outlier_fraction, rot):
    S = PS.scene(n=n, seed=seed, outlier_fraction=outlier_fraction, rot=rot)
    m = S["matches"]
    if algorithm == abi.PNP_GAUSS_NEWTON:
        m = m.copy(); m["inv_sigma2"] = m["info"]
    rng = np.random.default_rng(seed)
    init = (rng.uniform(size=n) < 0.05).astype(np.uint8) if algorithm == abi.PNP_LEVENBERG else np.zeros(n, np.uint8)
    oo = init.copy(); od = init.copy()
    ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m, oo, algorithm=algorithm, compute_covariance=True)
    ctx = device.Ctx(max_frames=2)
    try:
        rd = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m, od, algorithm=algorithm, compute_covariance=True)
        # run-to-run: bit-identical
        od2 = init.copy()
        rd2 = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m, od2, algorithm=algorithm, compute_covariance=True)
    finally:
        ctx.close()
    _compare(ro, rd, oo, od)
    assert bytes(rd) == bytes(rd2) and np.array_equal(od, od2)
    assert ro.is_ok == 1
    R = np.array(list(rd.R)).reshape(3, 3)
    ang = np.arccos(np.clip((np.trace(R @ S["R_true"].T) - 1) / 2, -1, 1))
    assert ang < 5e-3


def test_pnp_edge_cases():
    ctx = device.Ctx(max_frames=2)
    try:
        S = PS.scene(n=64, seed=3, outlier_fraction=0.0)
        m = S["matches"]
        # fewer than 3 matches / fewer than 5 initial inliers: not ok, nothing touched (IndirectCameraOptimizer.cpp:121-129)
        for k, init_bad in ((2, 0), (6, 3)):
            o1 = np.zeros(k, np.uint8); o1[:init_bad] = 1; o2 = o1.copy()
            ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m[:k].copy(), o1)
            rd = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m[:k].copy(), o2)
            assert ro.is_ok == rd.is_ok == 0 and ro.rounds == rd.rounds == 0 and np.array_equal(o1, o2)
        # fewer than 10 edges: one round, then "Too few edges" (:161-164)
        o1 = np.zeros(8, np.uint8); o2 = o1.copy()
        ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m[:8].copy(), o1)
        rd = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m[:8].copy(), o2)
        _compare(ro, rd, o1, o2)
        assert rd.is_ok == 0 and rd.rounds == 1
        # mCheckOutliers off: every flag cleared, all rounds run on everything
        S = PS.scene(n=300, seed=4, outlier_fraction=0.1)
        o1 = np.ones(300, np.uint8); o1[::2] = 0; o2 = o1.copy()
        ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], S["matches"], o1, check_outliers=False)
        rd = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], S["matches"], o2, check_outliers=False)
        # the gross outliers stay in, un-robustified in the last round: the round is still moving (steps ~1e-9) when its 10
        # iterations end or a rounding-level rho terminates it, so the two sides agree to the step size, not to 1e-9
        _compare(ro, rd, o1, o2, tol=1e-7)
        assert not o2.any() and rd.rounds == 4
        # a point behind / on the camera plane among the inliers: non-finite terms are handled like the reference handles them
        S = PS.scene(n=200, seed=8, outlier_fraction=0.0)
        m = S["matches"].copy()
        Pc = (S["R0"] @ m["X"][0]) + S["t0"]
        m["X"][0] = S["R0"].T @ (np.array([Pc[0], Pc[1], 0.0]) - S["t0"])       # z = 0 at the start pose
        o1 = np.zeros(200, np.uint8); o2 = o1.copy()
        ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m, o1)
        rd = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m, o2)
        assert ro.is_ok == rd.is_ok and ro.rounds == rd.rounds and np.array_equal(o1, o2)
        # over the LDS capacity: refused loudly
        big = np.zeros(2561, abi.PNP_MATCH_DTYPE)
        with pytest.raises(device.CmlHipError):
            ctx.pnp_optimize(S["R0"], S["t0"], S["K"], big, np.zeros(2561, np.uint8))
    finally:
        ctx.close()
