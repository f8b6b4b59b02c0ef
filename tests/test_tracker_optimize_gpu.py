"""DSOTracker::optimize / trackWithMotionModel through the host mirror ON THE DEVICE evaluation (cmlhip_tracker_eval) against the
oracle's restatement of the whole loop.  The control flow itself is held identical on identical evaluations by
tests/test_tracker_optimize_cpu.py; here the 9x9 systems differ by the fp32 accumulation order (3e-5, tests/test_tracker_parity_gpu.py),
so trials at a knife's edge may differ: the levels visited and the winner must agree; the loop stops when the increment norm falls
below 1e-3 (TR.cpp:176), so two runs whose trial sequences differ by a step agree in the pose to a fraction of that: 3e-4; the
level-0 energies to 1e-3."""
import numpy as np
import pytest

from libcml_amd import device, host
from tests import trk_opt_setup as TS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["small", "B"])
def setup(request):
    P = TS.make_problem(request.param)
    ctx = device.Ctx(max_frames=8)
    ctx.pyramid_build(501, P.W.gray[P.s.new], P.levels)
    for l in range(P.levels):
        ctx.tracker_set_reference(l, P.uvic[l])
    trk = host.HostTracker(ctx)
    trk.set_calibration(*P.W.K)
    yield P, ctx, trk
    trk.close(); ctx.close()


@pytest.mark.parametrize("w,dt", [((0.004, -0.003, 0.002), (0.03, -0.02, 0.025)), ((-0.006, 0.004, 0.003), (-0.04, 0.03, 0.02))])
def test_optimize_device_vs_oracle(setup, w, dt):
    P, ctx, trk = setup
    R0, t0 = TS.perturbed(P, w, dt)
    o = TS.oracle_optimize(P, R0, t0)
    r = trk.optimize(501, P.levels, R0, t0, P.ref_exp, P.init_exp)
    lv, it, ac, lam = trk.steps()
    assert bool(o["out"].isCorrect) == r["isCorrect"] and bool(o["out"].tooManySaturated) == r["tooManySaturated"]
    assert sorted(set(lv.tolist())) == sorted(set(s[0] for s in o["steps"]))
    assert abs(len(lv) - len(o["steps"])) <= max(3, len(lv) // 5)                     # trial counts: a few knife-edge decisions at most
    assert np.abs(o["R"] - r["R"]).max() < 3e-4 and np.abs(o["t"] - r["t"]).max() < 1e-3 * max(1.0, np.abs(o["t"]).max())
    assert abs(o["a"] - r["exposure"][0]) < 1e-3 and abs(o["b"] - r["exposure"][1]) < 0.5      # b is stepped in units of 1000 (scale_b): the 1e-3 stop is one grey level
    e_o = np.array(o["out"].E[:P.levels]) / np.maximum(np.array(o["out"].numTermsInE[:P.levels]), 1)
    e_d = r["E"][:P.levels] / np.maximum(r["numTerms"][:P.levels], 1)
    assert abs(e_o[0] / e_d[0] - 1) < 1e-3
    assert np.linalg.norm(r["t"] - P.tt) < 0.5 * np.linalg.norm(t0 - P.tt)


def test_track_with_motion_model_device_vs_oracle(setup):
    P, ctx, trk = setup
    hyps = [TS.perturbed(P, (0.02, -0.015, 0.01), (0.15, -0.1, 0.12)), TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025)),
            TS.perturbed(P, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))]
    for lcr in (100.0, 1e-6):
        o = TS.oracle_track(P, hyps, lcr, 0)
        trk.set_param("lastCoarseRMSE", lcr)
        r = trk.track_with_motion_model(501, P.levels, hyps, P.ref_exp, P.init_exp)
        assert o["ok"] == r["haveOneGood"] and o["tries"] == r["tries"]
        if o["ok"]:
            assert np.abs(o["R"] - r["R"]).max() < 1e-3 and np.abs(o["t"] - r["t"]).max() < 1e-3 * max(1.0, np.abs(o["t"]).max())
            assert abs(r["lastCoarseRMSE"] / o["achieved"] - 1) < 1e-2


def test_device_resident_optimize_vs_oracle(setup):
    """cmlhip_tracker_optimize_batch: the whole LM loop of a hypothesis in one workgroup.  Against the oracle's loop: same levels,
    nearly the same trial sequence (the 9x9 sums are fp32 in another order), converged pose 1e-4."""
    P, ctx, trk = setup
    cases = [((0.004, -0.003, 0.002), (0.03, -0.02, 0.025)), ((-0.006, 0.004, 0.003), (-0.04, 0.03, 0.02)), ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))]
    hyps = [TS.perturbed(P, w, dt) for w, dt in cases]
    res = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    for (R0, t0), r in zip(hyps, res):
        o = TS.oracle_optimize(P, R0, t0)
        assert bool(o["out"].isCorrect) == bool(r.isCorrect) and bool(o["out"].tooManySaturated) == bool(r.tooManySaturated)
        so = [(s[0], s[2]) for s in o["steps"]]
        sd = [(r.step_level[i], r.step_accept[i]) for i in range(min(r.n_steps, 256))]
        assert sorted(set(l for l, _ in so)) == sorted(set(l for l, _ in sd))
        assert abs(len(so) - len(sd)) <= max(3, len(so) // 5)
        Rd = np.array(r.R[:]).reshape(3, 3); td = np.array(r.t[:])
        assert np.abs(o["R"] - Rd).max() < 3e-4 and np.abs(o["t"] - td).max() < 1e-3 * max(1.0, np.abs(o["t"]).max())
        assert abs(o["a"] - r.a) < 1e-3 and abs(o["b"] - r.b) < 0.5
        assert abs((r.E[0] / r.numTermsInE[0]) / (o["out"].E[0] / o["out"].numTermsInE[0]) - 1) < 1e-3
        assert r.n_pass >= P.levels and r.pass_level[r.n_pass - 1] == 0
        cov_o = np.array(o["out"].covariance[:]); cov_d = np.array(r.covariance[:])
        assert np.abs(cov_d / cov_o - 1).max() < 1e-2


def test_batched_track_with_motion_model(setup):
    """All hypotheses side by side on the device + the reference's selection replayed on the host == the sequential procedure."""
    P, ctx, trk = setup
    hyps = [TS.perturbed(P, (0.02, -0.015, 0.01), (0.15, -0.1, 0.12)), TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025)),
            TS.perturbed(P, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), TS.perturbed(P, (-0.002, 0.001, 0.0), (0.01, 0.0, -0.01))]
    for lcr in (100.0, 1e-6):
        o = TS.oracle_track(P, hyps, lcr, 0)
        trk.set_param("lastCoarseRMSE", lcr)
        s = trk.track_with_motion_model(501, P.levels, hyps, P.ref_exp, P.init_exp)                       # sequential, device evaluations
        trk.set_param("lastCoarseRMSE", lcr)
        b = trk.track_with_motion_model(501, P.levels, hyps, P.ref_exp, P.init_exp, batched=True)        # one launch
        assert o["ok"] == b["haveOneGood"] == s["haveOneGood"] and o["tries"] == b["tries"] == s["tries"]
        if o["ok"]:
            assert np.abs(o["R"] - b["R"]).max() < 1e-3 and np.abs(o["t"] - b["t"]).max() < 1e-3 * max(1.0, np.abs(o["t"]).max())
            assert np.abs(s["R"] - b["R"]).max() < 1e-3
            assert abs(b["lastCoarseRMSE"] / o["achieved"] - 1) < 1e-2


def test_a_hypothesis_does_not_depend_on_its_batch(setup):
    """ADVICE round 3, closed in round 4: the number of workgroups a hypothesis is spread over follows the batch size (G = 8 for one, 4 for
    fifty, 2 for a hundred, 1 beyond the resident capacity), but the PARTS a level is summed in are fixed by the level's size — eight, in a
    fixed order, whichever workgroup evaluates them.  So the same hypothesis must give the same BITS alone, among 3, 50, 100 or 300, and
    wherever it stands in the batch."""
    P, ctx, trk = setup
    base = TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))
    keep = 8 * 14 + 8          # R, t, a, b and the two flags
    ref = None
    for n_hyp in (1, 3, 50, 100, 300):
        hyps = [base] + [TS.perturbed(P, (0.004 + 1e-4 * (i % 40), -0.003, 0.002), (0.03, -0.02 + 1e-3 * (i % 25), 0.025)) for i in range(1, n_hyp)]
        r = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)[0]
        b = bytes(bytearray(bytes(r))[:keep]) + bytes(r.E) + bytes(r.numTermsInE) + bytes(r.step_accept[:r.n_steps])
        if ref is None:
            ref = b
            assert r.isCorrect and r.n_steps >= 5
        assert b == ref, "hypothesis 0 differs in a batch of %d" % n_hyp
        if n_hyp == 50:                                           # the same 50 in reverse order: hypothesis 0 is now the last group of workgroups
            rev = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps[::-1])[-1]
            assert bytes(bytearray(bytes(rev))[:keep]) == ref[:keep]


def test_early_exit_behind_the_first_hypothesis(setup):
    """cmlhip_tracker_set_early_exit: the reference leaves its hypothesis loop behind the first good try (DSOTracker.h:306-309).  With the bar set,
    hypothesis 0's result is bit-identical to the run without, the others come back given up (n_steps = -1) once it has ended — at every batch
    shape (G = 8, 4, 1 workgroups per hypothesis) — and a bar hypothesis 0 does not meet changes nothing."""
    P, ctx, trk = setup
    base = TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))
    keep = 8 * 14 + 8
    for n_hyp in (6, 50, 300):
        # the others start far off: they need many more trials than hypothesis 0
        hyps = [base] + [TS.perturbed(P, (0.02 + 1e-3 * (i % 7), -0.015, 0.01), (0.15, -0.1 + 1e-2 * (i % 5), 0.12)) for i in range(1, n_hyp)]
        ctx.tracker_set_early_exit(0.0)
        ctx.profile_next_launch()
        full = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        t_full = ctx.elapsed_ms()
        r0 = full[0]
        assert r0.isCorrect and r0.n_steps >= 5 and all(r.n_steps >= 0 for r in full)
        rm0 = r0.E[0] / r0.numTermsInE[0]
        ctx.tracker_set_early_exit(1.5 * rm0)
        ctx.profile_next_launch()
        cut = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        t_cut = ctx.elapsed_ms()
        assert bytes(bytearray(bytes(cut[0]))[:keep]) == bytes(bytearray(bytes(r0))[:keep]) and cut[0].n_steps == r0.n_steps
        gave_up = sum(1 for r in cut[1:] if r.n_steps < 0)
        for a, b in zip(cut[1:], full[1:]):                       # a hypothesis that finished before the flag rose is a complete result, the same one
            if a.n_steps >= 0:
                assert bytes(bytearray(bytes(a))[:keep]) == bytes(bytearray(bytes(b))[:keep])
        slower = sum(1 for r in full[1:] if r.n_steps > r0.n_steps + 4)
        assert gave_up >= max(1, slower // 2), (n_hyp, gave_up, slower)
        assert t_cut <= t_full * 1.02, (n_hyp, t_cut, t_full)
        print("early exit, %d hypotheses: kernel %.3f -> %.3f ms, %d of %d given up" % (n_hyp, t_full, t_cut, gave_up, n_hyp - 1))
        ctx.tracker_set_early_exit(0.5 * rm0)                     # a bar hypothesis 0 does not meet: nothing gives up
        none = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        assert all(r.n_steps >= 0 for r in none)
        ctx.tracker_set_early_exit(0.0)


def test_batched_tracking_with_and_without_early_exit_agree(setup):
    """the host mirror's trackWithMotionModelBatched: same winner, same pose bits, same number of tries with the early exit on (default) and off"""
    P, ctx, trk = setup
    hyps = [TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))] + [TS.perturbed(P, (0.02, -0.015 + 1e-3 * i, 0.01), (0.15, -0.1, 0.12)) for i in range(5)]
    out = []
    for on in (1, 0):
        t2 = host.HostTracker(ctx); t2.set_calibration(*P.W.K)
        t2.set_param("batchedEarlyExit", on)
        t2.set_param("lastCoarseRMSE", 1e6)                       # (any first try that is correct ends the search)
        res = t2.track_with_motion_model(501, P.levels, hyps, P.ref_exp, P.init_exp, batched=True)
        out.append((bool(res["haveOneGood"]), res["R"].tobytes(), res["t"].tobytes(), int(res["winner"]), int(res["tries"])))
    assert out[0] == out[1] and out[0][0] and out[0][3] == 0 and out[0][4] == 1
