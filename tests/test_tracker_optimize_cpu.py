"""DSOTracker::optimize (TR.cpp:15-246) and trackWithMotionModel (DSOTracker.h:238-383): the C++ host mirror against the oracle's
independent restatement ON IDENTICAL EVALUATIONS (the mirror's evaluation provider is replaced by the oracle's computeResidual +
computeHessian), so that the comparison isolates the control flow: iteration caps, lambda schedule, extrapolation, accept / reject,
the saturation repeat, the rmse test against the previous try, winner selection and the early exits.  Decisions must be identical
and the final pose equal to 1e-12.  No GPU: the mirror never touches its device context here."""
import numpy as np
import pytest

from libcml_amd import host
from tests import trk_opt_setup as TS


@pytest.fixture(scope="module")
def problem():
    return TS.make_problem("small")


def _mirror(P):
    trk = host.HostTracker(None)
    trk.set_calibration(*P.W.K)
    trk.set_eval(TS.oracle_eval_fn(P))
    return trk


@pytest.mark.parametrize("w,dt", [((0.004, -0.003, 0.002), (0.03, -0.02, 0.025)), ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
                                  ((-0.01, 0.008, 0.004), (-0.05, 0.04, 0.02))])
def test_optimize_control_flow_identical(problem, w, dt):
    P = problem
    R0, t0 = TS.perturbed(P, w, dt)
    o = TS.oracle_optimize(P, R0, t0)
    trk = _mirror(P)
    r = trk.optimize(0, P.levels, R0, t0, P.ref_exp, P.init_exp)
    lv, it, ac, lam = trk.steps()
    assert len(o["steps"]) == len(lv) and len(lv) >= P.levels
    assert [s[:3] for s in o["steps"]] == list(zip(lv.tolist(), it.tolist(), ac.tolist()))
    assert np.array_equal(np.array([s[3] for s in o["steps"]]), lam)                     # lambda schedule, bit for bit
    assert bool(o["out"].isCorrect) == r["isCorrect"] and bool(o["out"].tooManySaturated) == r["tooManySaturated"]
    assert np.abs(o["R"] - r["R"]).max() < 1e-12 and np.abs(o["t"] - r["t"]).max() < 1e-12
    assert abs(o["a"] - r["exposure"][0]) < 1e-12 and abs(o["b"] - r["exposure"][1]) < 1e-9
    L = P.levels
    assert np.array_equal(np.array(o["out"].E[:L]), r["E"][:L]) and np.array_equal(np.array(o["out"].numTermsInE[:L]), r["numTerms"][:L])
    assert np.abs(np.array(o["out"].covariance[:]) - r["covariance"]).max() <= 1e-9 * np.abs(r["covariance"]).max()
    # and it is a tracker: the perturbation is reduced
    if np.linalg.norm(dt) > 0:
        assert np.linalg.norm(r["t"] - P.tt) < 0.5 * np.linalg.norm(t0 - P.tt)
    trk.close()


def test_optimize_aborts_on_rmse_of_previous_try(problem):
    """TR.cpp:183-189: a try whose level rmse exceeds 1.5 x the previous correct try's is abandoned at that level."""
    P = problem
    R0, t0 = TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))
    last = [1e-3] * 5                                    # an impossibly good previous try
    o = TS.oracle_optimize(P, R0, t0, TS.orc_problem(P, have_last=1, last_rmse=last))
    trk = _mirror(P)
    trk.set_last_residual(True, last[:P.levels])
    r = trk.optimize(0, P.levels, R0, t0, P.ref_exp, P.init_exp)
    assert not o["out"].isCorrect and not r["isCorrect"]
    lv, it, ac, lam = trk.steps()
    assert [s[:3] for s in o["steps"]] == list(zip(lv.tolist(), it.tolist(), ac.tolist()))
    assert set(lv.tolist()) == {min(P.levels - 1, 4)}    # only the coarsest level ran
    trk.close()


@pytest.mark.parametrize("optimize_a,optimize_b", [(1, 0), (0, 1), (0, 0)])
def test_optimize_light_parameter_branches(problem, optimize_a, optimize_b):
    """The three reduced solver branches of TR.cpp:99-119."""
    P = problem
    R0, t0 = TS.perturbed(P, (0.003, 0.002, -0.002), (0.02, 0.01, -0.02))
    o = TS.oracle_optimize(P, R0, t0, TS.orc_problem(P, optimize_a=optimize_a, optimize_b=optimize_b))
    trk = _mirror(P)
    trk.set_param("optimizeLightA", optimize_a); trk.set_param("optimizeLightB", optimize_b)
    r = trk.optimize(0, P.levels, R0, t0, P.ref_exp, P.init_exp)
    lv, it, ac, lam = trk.steps()
    assert [s[:3] for s in o["steps"]] == list(zip(lv.tolist(), it.tolist(), ac.tolist()))
    assert np.abs(o["R"] - r["R"]).max() < 1e-12 and np.abs(o["t"] - r["t"]).max() < 1e-12
    if not optimize_a:
        assert r["exposure"][0] == P.init_exp[0]
    if not optimize_b:
        assert r["exposure"][1] == P.init_exp[1]
    trk.close()


@pytest.mark.parametrize("last_coarse_rmse,failure_mode", [(100.0, 0), (1e-6, 0), (1e-6, 1)])
def test_track_with_motion_model(problem, last_coarse_rmse, failure_mode):
    """Hypothesis loop: a wrong hypothesis first, better ones after.  With the default mLastCoarseRMSE = 100 the first correct try
    ends the loop (DSOTracker.h:306-309); with a tiny one every hypothesis is tried and the best rmse wins."""
    P = problem
    hyps = [TS.perturbed(P, (0.02, -0.015, 0.01), (0.15, -0.1, 0.12)), TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025)),
            TS.perturbed(P, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), TS.perturbed(P, (-0.002, 0.001, 0.0), (0.01, 0.0, -0.01))]
    o = TS.oracle_track(P, hyps, last_coarse_rmse, failure_mode)
    trk = _mirror(P)
    trk.set_param("lastCoarseRMSE", last_coarse_rmse); trk.set_param("failureMode", failure_mode)
    r = trk.track_with_motion_model(0, P.levels, hyps, P.ref_exp, P.init_exp)
    assert o["ok"] == r["haveOneGood"] and o["winner"] == r["winner"] and o["tries"] == r["tries"]
    if last_coarse_rmse > 1:
        assert r["tries"] < len(hyps)
    else:
        assert r["tries"] == len(hyps)
    if r["haveOneGood"]:
        assert np.abs(o["R"] - r["R"]).max() < 1e-12 and np.abs(o["t"] - r["t"]).max() < 1e-12
        assert abs(o["a"] - r["exposure"][0]) < 1e-12
        assert r["lastCoarseRMSE"] == pytest.approx(o["achieved"], rel=0, abs=0)
    trk.close()
