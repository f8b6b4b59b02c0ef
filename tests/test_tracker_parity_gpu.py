"""GPU parity: pyramids, coarse-depth lists, tracker residual/Hessian, ORB reprojection term — device (C ABI) vs oracle."""
import ctypes as C

import numpy as np
import pytest

from libcml_amd import abi, device, synth
from tests import oracle_lib as O
from tests import trk_setup as T

pytestmark = pytest.mark.gpu


# "A" = BASELINE.json configs[0] as SURVEY §8 defines it (1 reference keyframe + 200 points, 640x480, 2 levels: the tracker path);
# "B" = the config-B shape (1241x376, 4 levels, 2000 points) — both at full size
LEVELS = {"small": [0, 1, 2], "A": [0, 1], "B": [0, 1, 2, 3]}


@pytest.fixture(scope="module", params=["small", "A", "B"])
def scene(request):
    s = T.make_scene(request.param)
    s.name = request.param
    ctx = device.Ctx(max_frames=8)
    yield s, ctx
    ctx.close()


def test_pyramid_build_bit_exact(scene):
    s, ctx = scene
    ctx.pyramid_build(77, s.W.gray[0], 5)
    grays, grads = O.build_pyramid(s.W.gray[0], 5)
    for l in range(5):
        d = ctx.pyramid_get(77, l)
        assert d.shape == grads[l].shape
        assert np.array_equal(d.view(np.uint32), grads[l].view(np.uint32)), l
    # put == build
    ctx.pyramid_put(78, 0, grads[0])
    assert np.array_equal(ctx.pyramid_get(78, 0).view(np.uint32), grads[0].view(np.uint32))
    assert ctx.pyramid_drop(78) == 0 and ctx.pyramid_drop(78) == abi.ERR_NOT_FOUND


def test_coarse_depth_lists_exact(scene):
    s, ctx = scene
    ctx.pyramid_build(1, s.W.gray[s.ref], s.levels)
    n_dev = ctx.tracker_make_coarse_depth(1, s.levels, s.cd_pts)
    lists, n_orc = T.oracle_coarse_depth(s)
    assert n_dev == n_orc
    assert n_orc[0] > 50
    for l in range(s.levels):
        d = ctx.tracker_get_reference(l)
        o = lists[l][:n_orc[l]]
        assert np.array_equal(d[:, :2], o[:, :2]), "list order / pixel coordinates (index bookkeeping) must be exact"
        assert np.array_equal(d[:, 3].view(np.uint32), o[:, 3].view(np.uint32))
        assert np.array_equal(d[:, 2].view(np.uint32), o[:, 2].view(np.uint32))      # the splat is ordered: collisions included


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_tracker_eval(scene, level):
    s, ctx = scene
    if level not in LEVELS[s.name]:
        pytest.skip("level not part of this configuration")
    ctx.pyramid_build(2, s.W.gray[s.new], s.levels)
    lists, n_orc = T.oracle_coarse_depth(s)
    uvic = lists[level][:n_orc[level]]
    ctx.tracker_set_reference(level, uvic)
    prm = abi.default_tracker_params()
    R, t, K, aff, b0 = T.tracker_inputs(s, level)
    out_d, rc = ctx.tracker_eval(2, level, R, t, K, aff, b0, prm, 1)
    assert rc == 0
    out_o, warped_o = T.oracle_tracker_eval(s, level, uvic, R, t, K, aff, b0, prm)
    assert (out_d.numTermsInE, out_d.numSaturated, out_d.numRobust, out_d.numWarped) == \
           (out_o.numTermsInE, out_o.numSaturated, out_o.numRobust, out_o.numWarped)
    assert out_o.numWarped > 20
    wd, n = ctx.tracker_get_warped(len(uvic) + 4)
    assert n == out_o.numWarped
    assert np.array_equal(wd.view(np.uint32), warped_o[:, :n].view(np.uint32)), "warped buffer must be bit-exact"
    assert abs(out_d.E - out_o.E) <= 2e-5 * abs(out_o.E)                      # fp32 sum order
    for k in (0, 2):
        assert abs(out_d.flow[k] - out_o.flow[k]) <= 2e-5 * max(abs(out_o.flow[k]), 1e-9)
    H9d = np.array(out_d.H9[:]).reshape(9, 9); H9o = np.array(out_o.H9[:]).reshape(9, 9)
    assert np.abs(H9d - H9o).max() <= 3e-5 * np.abs(H9o).max()
    Hd = np.array(out_d.H[:]).reshape(8, 8); Ho = np.array(out_o.H[:]).reshape(8, 8)
    for i in range(8):
        for j in range(8):
            assert abs(Hd[i, j] - Ho[i, j]) <= 3e-5 * np.sqrt(abs(Ho[i, i] * Ho[j, j])) + 1e-30
    bd = np.array(out_d.b[:]); bo = np.array(out_o.b[:])
    assert np.abs(bd - bo).max() <= 1e-4 * np.abs(bo).max()
    # the Gauss-Newton step of the iteration (TR.cpp:97-101, 140-159): inc = -ldlt(H with diag * (1 + lambda)).solve(b), lambda = 0.01.
    # Pivoted LDLT (Eigen semantics, pinned on the vendored Eigen) on the device's system against the oracle's system:
    # the step inherits the fp32 accumulation tolerance of H and b, amplified by the conditioning of the 8x8 system
    lam = 0.01
    Hl_d = Hd.copy(); Hl_d[np.diag_indices(8)] *= (1 + lam)
    Hl_o = Ho.copy(); Hl_o[np.diag_indices(8)] *= (1 + lam)
    inc_d, rc_d = O.ldlt_solve(Hl_d, -bd)
    inc_o, rc_o = O.ldlt_solve(Hl_o, -bo)
    assert rc_d == 0 and rc_o == 0
    Sv = 1.0 / np.sqrt(np.diag(Hl_o))
    back = Sv * (Hl_o @ inc_d + bo)                     # backward error of the device step in the oracle's (Jacobi-scaled) system
    assert np.linalg.norm(back) <= 1e-4 * np.linalg.norm(Sv * bo)


def test_tracker_empty_and_saturated(scene):
    s, ctx = scene
    ctx.pyramid_build(2, s.W.gray[s.new], s.levels)
    prm = abi.default_tracker_params()
    R, t, K, aff, b0 = T.tracker_inputs(s, 1)
    ctx.tracker_set_reference(1, np.zeros((0, 4), np.float32))
    out, rc = ctx.tracker_eval(2, 1, R, t, K, aff, b0, prm, 0)
    assert out.numTermsInE == 0 and out.E == 0
    # a negative cutoff saturates every term (TR.cpp:361-366): E = n * maxEnergy
    lists, n_orc = T.oracle_coarse_depth(s)
    ctx.tracker_set_reference(1, lists[1][:n_orc[1]])
    prm0 = abi.default_tracker_params(); prm0.cutoff = -1.0
    out, rc = ctx.tracker_eval(2, 1, R, t, K, aff, b0, prm0, 0)
    assert out.numSaturated == out.numTermsInE and out.numWarped == 0


def test_reproj_term(scene):
    s, ctx = scene
    poses, points, obs, fx, fy = T.reproj_inputs(s, n_obs=1000, n_pts=300)
    M6, b6, Jp, used = ctx.reproj_accumulate(poses, points, obs, fx, fy)
    M6o, b6o, Jpo, usedo = T.oracle_reproj(poses, points, obs, fx, fy)
    assert np.array_equal(used, usedo)
    assert used.sum() > 100
    assert np.abs(M6 - M6o).max() <= 1e-11 * np.abs(M6o).max()
    assert np.abs(b6 - b6o).max() <= 1e-11 * np.abs(b6o).max()
    assert np.abs(Jp - Jpo).max() <= 1e-11 * np.abs(Jpo).max()
    x, rc = ctx.reproj_solve(len(poses), 1e-5)
    M = M6o.copy(); M[np.diag_indices(len(M))] *= (1 + 1e-5)
    xo, rco = O.ldlt_solve(M, -b6o)
    assert rc == 0 and rco == 0
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
