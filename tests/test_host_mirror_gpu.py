"""The C++ host mirror (cml_amd::DSOBundleAdjustment::run, DSOTracker::optimize) driving the device through the
C ABI, against the same procedures composed from oracle primitives."""
import numpy as np
import pytest

from libcml_amd import device, host
from tests import ba_ref_run, ba_setup as S
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


# "tiny" (64 points, 3 keyframes) is not used here: its Gauss-Newton iteration is unstable (energy grows), so fp32-level
# differences are amplified chaotically and a forward comparison of the final state is meaningless there.
@pytest.mark.parametrize("config", ["small", "medium", "B", "E"])      # B, E: BASELINE.json configs[1] and configs[4] at FULL size
def test_host_algebra_and_run(config):
    I = S.make_inputs(config)
    half = config == "E"
    if half:                                                            # config E stores fp16 texels: the oracle gets the same rounded images
        for k in range(I.N):
            for lvl in range(len(I.grads[k])):
                I.grads[k][lvl] = I.grads[k][lvl].astype(np.float16).astype(np.float32)
        from libcml_amd import synth                                      # BA::addPoints takes the gradient weights from the stored (rounded) texels
        _, wts = synth.point_colors_weights(I.W, [I.grads[k][0] for k in range(I.N)])
        I.points["weights"] = wts
    from libcml_amd import abi
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R, texel_format=abi.TEXEL_F16 if half else abi.TEXEL_F32)
    ba = host.window_to_host_ba(ctx, I.W)
    c = ba.counts()
    assert (c["frames"], c["points"], c["residuals"]) == (I.N, I.P, I.R)
    # host algebra (computeAdjoints / computeDelta / precompute / nullspaces) vs the oracle's frame algebra
    A = ba.algebra()
    assert np.abs(A["adH"] - I.adH).max() < 1e-12 * np.abs(I.adH).max()
    assert np.abs(A["adT"] - I.adT).max() < 1e-12 * np.abs(I.adT).max()
    assert np.abs(A["adHTd"] - I.adHTd).max() <= 1e-6 * np.abs(I.adHTd).max() + 1e-12
    for f in ("R", "t", "R0", "t0"):
        assert np.abs(A["pairs"][f] - I.pairs[f]).max() < 1e-13
    assert np.abs(A["pairs"]["aff_a"] - I.pairs["aff_a"]).max() < 1e-14
    assert np.array_equal(A["prior"], I.prior)
    # full run
    ok = ba.run()                       # 4 iterations; on these parameters run() keeps the loop resident on the device (runResident)
    assert ok, ba.last_error()
    ref = ba_ref_run.oracle_run(I)
    print("run() vs oracle_run at %s: R=%d flips=%d energies dev %s ref %s" % (config, I.R, int((ba.residual_states()[2].astype(bool) != ref["good"]).sum()),
          np.array2string(np.asarray(ba.energies()), precision=6), np.array2string(np.array(ref["log"]["energy"]), precision=6)))
    # residual bookkeeping (set membership) must be exact away from thresholds: allow a handful of flips
    st, alive, good = ba.residual_states()
    flips = int((good.astype(bool) != ref["good"]).sum())
    assert flips <= max(2, I.R // 200), flips
    # poses / affine / idepth updates: gauge-amplified fp32 differences, stated tolerance
    dev_poses = []
    for k in range(I.N):
        f = ba.frame(k)
        dev_poses.append((f["R"].copy(), f["t"].copy()))
        Rm, t, a, b = ref["poses"][k]
        assert np.abs(f["R"] - Rm).max() < 1e-3
        assert np.abs(f["t"] - t).max() < 5e-3 * max(1.0, np.abs(t).max())
        assert abs(f["ab"][0] - a) < 2e-3 and abs(f["ab"][1] - b) < 0.2
    idp, palive, ng = ba.points()
    both = palive.astype(bool)
    both[ref["outliers"]] = False
    assert np.abs(idp[both] / ref["idepth"][both] - 1).max() < 8e-2          # weakly observed points amplify the gauge noise
    ratio = idp[both] / ref["idepth"][both]
    assert np.median(np.abs(ratio - 1)) < 2e-3                      # includes the free monocular scale gauge
    assert np.median(np.abs(ratio / np.median(ratio) - 1)) < 3e-4   # gauge removed
    e_dev = ba.energies()
    e_ref = np.array(ref["log"]["energy"])
    # the objective itself is well conditioned: per-iteration photometric energies must agree closely
    k = min(len(e_dev) - 1, len(e_ref) - 2)
    assert k >= 1
    assert np.abs(e_dev[1:1 + k] / e_ref[1:1 + k] - 1).max() < 5e-3, (e_dev, e_ref)
    # first entry is energy / #residuals (BA.cpp:798).  The mirror's pair records agree with the oracle's to 1e-13, not in every bit (two
    # SE(3) compositions): among 152 000 residuals one that sits on a classification threshold can fall the other way, which moves the
    # sum by up to its capped energy (observed at config E: 2.8e-6 of the total); the per-residual arithmetic on IDENTICAL inputs is
    # held bit-exact by tests/test_resident_oracle_gpu.py and tests/test_config_e_gpu.py
    assert abs(e_dev[0] * I.R / e_ref[0] - 1) < (1e-9 if I.R < 50000 else 1e-5)
    ba.close(); ctx.close()
    # ---- bounds computed from THIS window: the fixed bars above are the outer sanity check; the
    # statement that scales with the window is "the device is no further from the oracle than a few times the distance between two correct
    # roundings of the oracle itself" — the same sources built with the reference's Release flags (fused multiply-adds), same inputs
    if config in ("small", "medium", "B"):
        I2 = S.make_inputs(config)
        ref2 = ba_ref_run.oracle_run_release_rounding(I2)

        def dist(poses, idepth, good, energy):
            dR = max(np.abs(poses[k][0] - ref["poses"][k][0]).max() for k in range(I.N))
            dt = max(np.abs(poses[k][1] - ref["poses"][k][1]).max() for k in range(I.N))
            ratio = idepth[both] / ref["idepth"][both]
            did = float(np.median(np.abs(ratio / np.median(ratio) - 1)))
            kk = min(len(energy), len(e_ref) - 1) - 1
            de = float(np.abs(np.asarray(energy)[1:1 + kk] / e_ref[1:1 + kk] - 1).max())
            return dict(R=float(dR), t=float(dt), idepth=did, energy=de, flips=int((good != ref["good"]).sum()))
        d_dev = dist(dev_poses, idp, good.astype(bool), e_dev)
        d_ref = dist([(p[0], p[1]) for p in ref2["poses"]], ref2["idepth"], ref2["good"], [0.0] + list(ref2["log"]["energy"][1:]))
        print("   window yardstick at %s: device-vs-oracle %s | oracle(Release rounding)-vs-oracle %s" % (config, d_dev, d_ref))
        # (the yardstick is ONE draw of the window's rounding noise and the device's run is another: a factor of 10 between two draws of one
        #  noise process is the bar; measured ratios: poses / inverse depths 0.5 ... 1.1, per-iteration energies up to 7)
        for q, floor in (("R", 1e-7), ("t", 1e-7), ("idepth", 1e-7), ("energy", 1e-7)):
            assert d_dev[q] <= 10 * d_ref[q] + floor, (q, d_dev, d_ref)
        assert d_dev["flips"] <= 10 * d_ref["flips"] + 2, (d_dev, d_ref)


def test_orthogonalize_matches_oracle():
    I = S.make_inputs("tiny")
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
    ba = host.window_to_host_ba(ctx, I.W)
    A = ba.algebra()
    rng = np.random.default_rng(1)
    x = rng.normal(size=8 * I.N + 4)
    xo = O.orthogonalize(x, A["nullspaces"], 1e-5)
    xh = ba.orthogonalize(x)
    assert np.abs(xo - xh).max() < 1e-12
    ba.close(); ctx.close()


def test_run_is_deterministic():
    """Every reduction on the BA path has a fixed order: two runs on the same inputs must agree bit for bit."""
    outs = []
    for _ in range(3):
        I = S.make_inputs("tiny")
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        assert ba.run(), ba.last_error()
        idp, alive, ng = ba.points()
        st, ralive, good = ba.residual_states()
        outs.append((idp.copy(), good.copy(), np.concatenate([ba.frame(k)["state"] for k in range(I.N)])))
        ba.close(); ctx.close()
    for o in outs[1:]:
        assert np.array_equal(o[1], outs[0][1])
        assert np.array_equal(o[0].view(np.uint64), outs[0][0].view(np.uint64))
        assert np.array_equal(o[2].view(np.uint64), outs[0][2].view(np.uint64))


WIDE22 = (22, 300, 320, 240, 3, 260.0, 260.0, 159.5, 119.5)       # 8N > 160: the solve leaves the LDS (k_ba_solve_global)


@pytest.mark.parametrize("config", ["small", "medium", WIDE22], ids=["small", "medium", "wide22"])
def test_resident_iterations_match_host_loop(config):
    """run() steps the frames on the host between device calls; runResident() keeps the whole loop on the device (frame
    step, pair precomputation, computeDelta, orthogonalize in kernels).  Same arithmetic, different place: the results
    must agree far below the fp32 accumulation noise of the Hessians."""
    res = []
    for mode in ("host", "resident"):
        I = S.make_inputs(config)
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 5)
        ba.set_param("ThOptIterations", 0.0)          # no early break (BA.cpp:879)
        ok = ba.run_host_loop() if mode == "host" else ba.run_resident()
        assert ok, ba.last_error()
        assert ba.counts()["iterations"] == 5
        idp, alive, ng = ba.points()
        st, ralive, good = ba.residual_states()
        frames = [ba.frame(k) for k in range(I.N)]
        res.append((idp.copy(), good.copy(), frames, ba.energies(8)))
        ba.close(); ctx.close()
    (idp_h, good_h, fr_h, e_h), (idp_r, good_r, fr_r, e_r) = res
    assert int((good_h != good_r).sum()) <= max(1, len(good_h) // 2000)
    for a, b in zip(fr_h, fr_r):
        assert np.abs(a["state"] - b["state"]).max() < 1e-7 * max(1.0, np.abs(a["state"]).max())
        assert np.abs(a["R"] - b["R"]).max() < 1e-8 and np.abs(a["t"] - b["t"]).max() < 1e-7
        assert abs(a["th"] - b["th"]) <= 1e-5 * abs(a["th"])
    assert np.abs(idp_h / idp_r - 1).max() < 1e-5
    assert len(e_h) == len(e_r) and np.abs(e_h / e_r - 1).max() < 1e-6, (e_h, e_r)


def test_host_tracker_recovers_known_motion():
    """DSOTracker::optimize (TR.cpp:15-246) through the host mirror: coarse-to-fine LM on the device residual/Hessian
    kernel.  The synthetic scene has a known relative pose and affine brightness between the reference keyframe and the
    new frame; started from a perturbed pose the loop must come back to it, level by level, deterministically."""
    from tests import trk_setup as T
    from libcml_amd import synth
    s = T.make_scene("medium", eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)    # exact geometry: the optimum is the true motion
    W = s.W
    fx, fy, cx, cy = W.K
    outs = []
    for rep in range(2):
        ctx = device.Ctx(max_frames=8)
        trk = host.HostTracker(ctx)
        trk.set_calibration(fx, fy, cx, cy)
        L = s.levels
        ctx.pyramid_build(500, W.gray[s.ref], L)
        ctx.pyramid_build(501, W.gray[s.new], L)
        nout = trk.make_coarse_depth(500, L, s.cd_pts)
        assert nout[0] > 100
        Rt = W.R_true[s.new] @ W.R_true[s.ref].T
        tt = W.t_true[s.new] - Rt @ W.t_true[s.ref]
        R0 = synth.so3_exp(np.array([0.004, -0.003, 0.002])) @ Rt
        t0 = tt + np.array([0.03, -0.02, 0.025])
        a_r, b_r = W.aff_true[s.ref]; a_n, b_n = W.aff_true[s.new]
        r = trk.optimize(501, L, R0, t0, [a_r, b_r, float(W.ab_exposure[s.ref])], [a_r, b_r, float(W.ab_exposure[s.new])])
        assert r["isCorrect"]            # (tooManySaturated mirrors the reference literally: it is set to haveGoodPoints, TR.cpp:138)
        assert r["numTerms"][0] > 100 and np.all(r["iterations"][:L] >= 1)
        err_R0 = np.linalg.norm(synth.so3_log(R0 @ Rt.T)) if hasattr(synth, "so3_log") else np.arccos(np.clip((np.trace(R0 @ Rt.T) - 1) / 2, -1, 1))
        err_R = np.arccos(np.clip((np.trace(r["R"] @ Rt.T) - 1) / 2, -1, 1))
        err_t0 = np.linalg.norm(t0 - tt); err_t = np.linalg.norm(r["t"] - tt)
        assert err_R < 0.25 * err_R0 and err_t < 0.25 * err_t0, (err_R, err_R0, err_t, err_t0)
        # photometric rmse at level 0 must be small compared with the image contrast
        assert r["E"][0] / r["numTerms"][0] < 40.0
        outs.append((r["R"].copy(), r["t"].copy(), r["E"].copy(), r["exposure"].copy()))
        trk.close(); ctx.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("config", ["small", "medium"])
def test_run_delegates_to_resident_loop_with_the_same_early_exit(config):
    """run() keeps the loop on the device under the default parameters (forceAccept, fixLambda).  The reference's early exit
    `if (canbreak && it >= 1) break` (BA.cpp:879) is mirrored by a sticky device flag: same number of iterations, same
    per-iteration energies and same final state as the literal host loop."""
    res = []
    for mode in ("host", "auto"):
        I = S.make_inputs(config)
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 12)
        ba.set_param("ThOptIterations", 400.0)        # loose enough that the loop leaves before the 12th iteration
        ok = ba.run_host_loop() if mode == "host" else ba.run()
        assert ok, ba.last_error()
        idp, alive, ng = ba.points()
        res.append((ba.counts()["iterations"], ba.energies(16).copy(), idp.copy(), [ba.frame(k)["state"].copy() for k in range(I.N)]))
        ba.close(); ctx.close()
    (it_h, e_h, idp_h, st_h), (it_r, e_r, idp_r, st_r) = res
    assert it_h == it_r and 2 <= it_h < 12, (it_h, it_r)
    assert len(e_h) == len(e_r) and np.abs(e_h / e_r - 1).max() < 1e-6
    for a, b in zip(st_h, st_r):
        assert np.abs(a - b).max() < 1e-7 * max(1.0, np.abs(a).max())
    assert np.abs(idp_h / idp_r - 1).max() < 1e-5


def test_host_loop_with_step_rejection():
    """forceAccept = false, fixLambda = false: the Levenberg-Marquardt accept/reject branch of BA::run (BA.cpp:830-876) —
    total energy with the linearised/prior term (calcLEnergy), loadSateBackup + restore of the points on rejection, lambda
    x 0.25 / x 100.  An over-long first step (large lambda scale-down is not available at iteration 0) makes at least one
    rejection likely; the accept/reject sequence and lambda must match the oracle-composed run."""
    I = S.make_inputs("small", idepth_noise=0.25)           # poor depths: the first Gauss-Newton steps overshoot
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
    ba = host.window_to_host_ba(ctx, I.W)
    try:
        for k, v in (("forceAccept", 0), ("fixLambda", 0), ("iterations", 6), ("ThOptIterations", 0.0), ("fixedLambda", 1e-1)):
            ba.set_param(k, v)
        assert ba.run(), ba.last_error()                     # not the resident loop: the host decides every iteration
        ref = ba_ref_run.oracle_run(I, iterations=6, fixed_lambda=1e-1, th_opt=0.0, force_accept=False, fix_lambda=False)
        acc = ref["log"]["accepted"]
        assert ba.counts()["iterations"] == len(acc) == 6
        assert ba.rejected() == acc.count(False)
        assert abs(ba.last_lambda() / ref["log"]["lam"][-1] - 1) < 1e-12
        e_dev = ba.energies(16); e_ref = np.array(ref["log"]["energy"])
        assert len(e_dev) >= 2                               # energy/#residuals of the preamble + one entry per accepted step
        k = min(len(e_dev), len(e_ref) - 1) - 1
        assert np.abs(e_dev[1:1 + k] / e_ref[1:1 + k] - 1).max() < 5e-3
    finally:
        ba.close(); ctx.close()


@pytest.mark.parametrize("config", ["small", "medium"])
def test_one_wait_run_equals_the_stepwise_run(config):
    """run() with ONE host wait (upload scopes, preamble pass enqueue-only with its tail in the first solve launch, the newest frame re-anchored on the device,
    closing pass through the resident residual kernel, packed outputs: cmlhip_ba_finish_run) against the same run with a host wait behind every stage
    (CMLHOST_RUN_STEPWISE=1: first pass read back, three getters behind the loop, re-anchoring on the host, closing pass through the record kernel).
    The iterations are the same device work: iteration count and energy log identical in every bit; the final frame states and inverse depths agree to
    1e-12 (the closing pass does not move them); the closing pass's decisions — residual states, good flags, outliers, the new energy threshold — identical
    (its pair records differ by the re-anchoring's rounding only: device exp() against the host's)."""
    import os
    res = []
    for mode in ("stepwise", "one_wait"):
        I = S.make_inputs(config)
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 4)
        if mode == "stepwise":
            os.environ["CMLHOST_RUN_STEPWISE"] = "1"
        try:
            assert ba.run(), ba.last_error()
        finally:
            os.environ.pop("CMLHOST_RUN_STEPWISE", None)
        idp, alive, ng = ba.points()
        st, ralive, good = ba.residual_states()
        res.append(dict(it=ba.counts()["iterations"], e=ba.energies(16).copy(), idp=idp.copy(), alive=alive.copy(), ng=ng.copy(), st=st.copy(), ralive=ralive.copy(),
                        good=good.copy(), out=sorted(int(x) for x in ba.outliers()), frames=[ba.frame(k) for k in range(I.N)]))
        ba.close(); ctx.close()
    a, b = res
    assert a["it"] == b["it"] == 4
    assert np.array_equal(a["e"].view(np.uint64), b["e"].view(np.uint64)), (a["e"], b["e"])      # preamble energy / n, then the iterations' energies
    for fa, fb in zip(a["frames"], b["frames"]):
        assert np.abs(fa["state"] - fb["state"]).max() <= 1e-12 * max(1.0, np.abs(fa["state"]).max())
        assert np.abs(fa["R"] - fb["R"]).max() < 1e-12 and np.abs(fa["t"] - fb["t"]).max() < 1e-12
    assert a["frames"][-1]["th"] == b["frames"][-1]["th"]                                        # setNewFrameEnergyTH of the closing pass
    assert np.array_equal(a["idp"], b["idp"]) and np.array_equal(a["alive"], b["alive"]) and np.array_equal(a["ng"], b["ng"])
    assert np.array_equal(a["st"], b["st"]) and np.array_equal(a["ralive"], b["ralive"]) and np.array_equal(a["good"], b["good"])
    assert a["out"] == b["out"]


def test_one_wait_run_returns_the_residual_energies_when_asked():
    """`keepResidualEnergies`: run()'s closing pass also reads state_NewState / state_energy / state_NewEnergy / state_NewEnergyWithOutlier of every residual back — in the
    one-wait form through the device-side permutation into the caller's order (k_res_to_caller) inside cmlhip_ba_finish_run's one copy.  Held against the stepwise run
    (record kernel, host permutation): same states, energies to 1e-5 (the two closing passes differ by the rounding of the re-anchored pair records only)."""
    import os
    res = []
    for mode in ("stepwise", "one_wait"):
        I = S.make_inputs("medium")
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 3)
        ba.set_param("keepResidualEnergies", 1)
        if mode == "stepwise":
            os.environ["CMLHOST_RUN_STEPWISE"] = "1"
        try:
            assert ba.run(), ba.last_error()
        finally:
            os.environ.pop("CMLHOST_RUN_STEPWISE", None)
        res.append(ba.export()[2].copy())
        ba.close(); ctx.close()
    a, b = res
    assert len(a) == len(b) > 1000
    for f in ("state_state", "state_NewState", "alive", "good"):
        assert np.array_equal(a[f], b[f]), f
    live = a["alive"] == 1
    assert live.sum() > 1000 and np.abs(a["state_energy"][live]).max() > 1.0                       # the energies really came back
    for f in ("state_energy", "state_NewEnergy"):
        assert np.allclose(a[f][live], b[f][live], rtol=1e-5, atol=1e-6), f
