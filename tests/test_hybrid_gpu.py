"""BASELINE.json config C — the config-B window (8 keyframes x 2000 points) with 1000 ORB reprojection residuals mixed into the
pose solution (MODSLAM's hybrid path: DSOBundleAdjustment::addIndirectToProblem, BA.cpp:2574-2729, called from
solveLevenbergMarquardt :1327-1329) — through the host mirror `cml_amd::DSOBundleAdjustment` and the C ABI, against the oracle."""
import ctypes as C

import numpy as np
import pytest

from libcml_amd import abi, device, host, synth
from tests import oracle_lib as O
from tests import trk_setup as T

pytestmark = pytest.mark.gpu


def _indirect_inputs(W, n_obs=1000, n_pts=300, seed=11):
    class _S:
        pass
    s = _S(); s.W = W
    return T.reproj_inputs(s, n_obs=n_obs, n_pts=n_pts, seed=seed)


def _build(ctx, W, base, mixed, pts, obs):
    ba = host.window_to_host_ba(ctx, W, image_id_base=base, levels=1)
    ba.set_param("iterations", 1)
    ba.set_param("mixedBundleAdjustment", 1 if mixed else 0)
    ba.set_indirect_points(pts, obs)
    return ba


@pytest.mark.parametrize("config", ["small", "B"])
def test_hybrid_term_mixed_into_the_pose_solution(config):
    W = synth.make_window(config, seed=7)
    N, P = W.N, W.P
    if N <= 4:                                        # the reference mixes only with more than 4 frames (BA.cpp:1327)
        W = synth.make_window((5, P, W.w, W.h, W.levels) + tuple(W.K), seed=7)
        N = W.N
    ctx = device.Ctx(max_frames=N, max_points=P, max_residuals=P * N)
    _, pts, obs, fx, fy = _indirect_inputs(W)
    plain = _build(ctx, W, 1000, False, pts, obs)
    mixed = _build(ctx, W, 2000, True, pts, obs)
    poses = np.zeros((N, 12))
    for i in range(N):
        f = mixed.frame(i)
        poses[i, :9] = f["R"].ravel(); poses[i, 9:] = f["t"]
    st0 = [mixed.frame(i)["state"].copy() for i in range(N)]
    mixed_h = _build(ctx, W, 3000, True, pts, obs)
    assert plain.run_host_loop(), plain.last_error()
    assert mixed.run(), mixed.last_error()            # run() keeps the loop on the device: the term is evaluated and mixed inside the solve kernel
    assert mixed_h.run_host_loop(), mixed_h.last_error()   # the literal host-side procedure (cmlhip_reproj_accumulate / _solve + host weighting)
    x6p, _, xp = plain.indirect()
    x6, unc, x = mixed.indirect()
    assert x6p is None and x6 is not None             # mixedBundleAdjustment off: addIndirectToProblem returns at once (:2575)
    # the indirect solution against the oracle on the poses the frames had at the solve
    M6o, b6o, Jpo, usedo = T.oracle_reproj(poses, pts, obs, fx, fy)
    assert usedo.sum() > 500
    Mo = M6o.copy(); Mo[np.diag_indices(len(Mo))] *= (1 + 1e-5)
    xo, rco = O.ldlt_solve(Mo, -b6o)
    assert rco == 0
    assert np.abs(x6 - xo).max() <= 1e-8 * np.abs(xo).max()
    # the literal weighting of :2714-2727 (numIndirectPoint = 1, numDirectPoint = 0): the pose part of x IS the indirect solution,
    # the affine and calibration parts are those of the photometric solve
    xr, xpr = x[4:].reshape(N, 8), xp[4:].reshape(N, 8)
    assert np.array_equal(xr[:, :6], x6.reshape(N, 6))
    assert np.array_equal(xr[:, 6:], xpr[:, 6:]) and np.array_equal(x[:4], xp[:4])
    assert np.abs(xr[:, :6] - xpr[:, :6]).max() > 0
    # and the step the frames took is -x (setStep, :1433-1441; doStepFromBackup, DSOFrame.h)
    for i in range(N - 1):                            # (the newest frame's evaluation point is re-set in the epilogue, :893-897)
        st = mixed.frame(i)["state"]
        assert np.allclose(st[:6] - st0[i][:6], -xr[i, :6], rtol=0, atol=1e-12 * max(1.0, np.abs(xr[i, :6]).max()))
    assert len(unc) == len(pts)                       # setUncertainty of every indirect point (:2690-2692), the inverse of a rank-one matrix
    assert np.isfinite(mixed.energies()).all()
    # device-resident mixing == the host-side procedure
    x6h, unch, xh = mixed_h.indirect()
    # (every sum of the term has a fixed order — reproj.hip: a workgroup per frame, no atomics — so the two paths differ only in where the
    #  frame poses were composed: bar 1e-12, as before the round-2 detour through fp64 atomics)
    assert np.abs(x6 - x6h).max() <= 1e-12 * np.abs(x6h).max() and np.abs(x - xh).max() <= 1e-12 * np.abs(xh).max()
    assert len(unc) == len(unch)                      # (the values are the cofactor 'inverse' of a rank-one matrix, :2690-2692: 0/0-like, last-bit
                                                      #  differences of the per-point Jacobian sums change them arbitrarily — nothing to compare)
    for i in range(N):
        a, b = mixed.frame(i), mixed_h.frame(i)
        assert np.abs(a["state"] - b["state"]).max() < 1e-12 * max(1.0, np.abs(a["state"]).max())
    assert np.abs(mixed.energies() / mixed_h.energies() - 1).max() < 1e-9


def test_hybrid_term_is_run_to_run_reproducible():
    """Config C twice from scratch: the mixed solution, the frame states and the energies must agree in every bit."""
    outs = []
    for _ in range(2):
        W = synth.make_window("B", seed=7)
        ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
        _, pts, obs, fx, fy = _indirect_inputs(W)
        ba = _build(ctx, W, 1000, True, pts, obs)
        ba.set_param("iterations", 4)
        assert ba.run(), ba.last_error()
        x6, unc, x = ba.indirect()
        outs.append((x6.copy(), x.copy(), np.concatenate([ba.frame(i)["state"] for i in range(W.N)]), np.asarray(ba.energies()).copy()))
        ba.close(); ctx.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))
