"""The TIMED kernels against the oracle, directly and at size ("next" item 1a).

bench.py times the device-resident loop, whose residual kernel is k_ba_lin_rs4 (config B: fp32 texels, 4 lanes per residual) or
k_ba_lin_rs (config E: 20 keyframes x 8000 points on 1920x1080 tiled fp16 level-0 images, a lane per residual).  Until round 3 those
two were only compared with the record-writing kernel k_ba_linearize (HIP against HIP).  Here the window is set up exactly as bench.py
sets it up (host mirror: addNewFrame / addPoint / run / beginResident), iterated, and every residual pass checked is replayed on the
oracle from the device's own state (tests/resident_check.py): states, energies, JpJdF, centre projections and the re-materialised
74-float records must be identical in every bit.  Reference: DSOBundleAdjustment.cpp:62-316 (linearize), 2051-2093 (applyRes)."""
import numpy as np
import pytest

from libcml_amd import abi, device, host, synth
from tests import resident_check as RC

pytestmark = pytest.mark.gpu


def bench_window(config, seed=0xC0FFEE, shard=0):
    """The window of bench.py: (ctx, ba, replay) with the resident loop begun."""
    W = synth.make_window(config, seed=seed, shard=shard)
    half = config == "E"
    ctx = device.Ctx(max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N,
                     texel_format=abi.TEXEL_F16 if half else abi.TEXEL_F32)
    ba = host.window_to_host_ba(ctx, W, image_id_base=1000 * (shard + 1), levels=1)
    ba.set_param("iterations", 1)
    assert ba.run(), ba.last_error()
    _, _, R = ctx.refresh_window_size()
    assert ba.begin_resident(), ba.last_error()
    replay = RC.make_replay(ctx, ba, W)
    assert replay.R == R
    return W, ctx, ba, replay


@pytest.mark.parametrize("config,kernel_tile", [("B", 16), ("E", 64), ("small", 16), ("medium", 16)])
def test_timed_resident_kernel_bit_exact_against_oracle(config, kernel_tile):
    W, ctx, ba, replay = bench_window(config)
    try:
        if config == "B":
            assert (W.N, W.P, replay.R) == (8, 2000, 14000) and (W.w, W.h) == (1241, 376)
        if config == "E":
            assert (W.N, W.P, replay.R) == (20, 8000, 152000) and (W.w, W.h) == (1920, 1080)
            assert replay.R >= 36 * 1024            # the regime in which the library runs k_ba_lin_rs (lane per residual, tiled fp16 level 0)
        lam = 1e-5
        reports = []
        for it in range(6):                          # passes 1..6 of the loop: the state moves, every pass is replayed from the device's state
            rep = RC.check_one_pass(ctx, replay, lam, with_records=(it in (0, 5)))
            reports.append(rep)
            assert rep["ok"], (it, rep)
        assert reports[-1]["n_in"] > 0.5 * replay.R
        for _ in range(40):                          # far into the loop (bench.py times after hundreds of iterations)
            ctx.ba_iteration_async(lam)
        rep = RC.check_one_pass(ctx, replay, lam, with_records=True)
        assert rep["ok"], rep
        if config in ("B", "E"):                     # the BASELINE windows: >= 90 % of the billed residuals gather texels and are IN
            assert rep["n_sampled"] >= 0.9 * replay.R and rep["n_in"] >= 0.9 * replay.R, rep
    finally:
        replay.close(); ba.close(); ctx.close()
