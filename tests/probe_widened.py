"""(kept under tests/: it times the oracle beside the device, which only tests may do)
Timing of the round-1 "next" rows: DSOInitializer::calcResAndGS (f3) and the pose-only optimisation (f4)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device
from tests import initializer_setup as IS
from tests import pnp_setup as PS
from tests import lba_setup as LS


def timed(f, n=20, warm=3):
    for _ in range(warm): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6


ctx = device.Ctx(max_frames=2)
for level, step in ((0, 2), (1, 2), (2, 1)):
    W, g0, g1, R, t, ratio, tlog = IS.scene(level=level, config="medium")
    pts = IS.make_points(g0, step=step)
    prm = IS.make_params(W.K, level, R, t, ratio, tlog)
    ctx.pyramid_put(40 + level, level, g1)
    d = timed(lambda: ctx.initializer_calc_res_and_gs(40 + level, level, prm, pts.copy()))
    o = timed(lambda: IS.oracle_calc(g1, prm, pts), n=5, warm=1)
    print("calcResAndGS level %d (%dx%d): %5d points  device %.1f us per synchronous call (H2D %d KB + kernel + D2H)   oracle (1 core) %.1f us" %
          (level, g1.shape[1], g1.shape[0], len(pts), d, len(pts) * pts.itemsize // 1024, o))
for n in (150, 600, 2560):
    for alg, name in ((abi.PNP_LEVENBERG, "Levenberg"), (abi.PNP_GAUSS_NEWTON, "Gauss-Newton")):
        S = PS.scene(n=n, seed=5)
        m = S["matches"]
        d = timed(lambda: ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m, np.zeros(n, np.uint8), algorithm=alg))
        o = timed(lambda: PS.oracle_pnp(S["R0"], S["t0"], S["K"], m, np.zeros(n, np.uint8), algorithm=alg), n=5, warm=1)
        r = ctx.pnp_optimize(S["R0"], S["t0"], S["K"], m, np.zeros(n, np.uint8), algorithm=alg)
        print("pose-only optimisation %-12s %4d matches: device %.1f us per synchronous call (one launch, solve() calls per round %s)   oracle (1 core) %.1f us" %
              (name, n, d, list(r.lm_iterations), o))
for kw, name in ((dict(n_points=800, seed=2), "10 keyframes /  800 points"), (dict(n_points=4000, seed=4, n_local=21, n_fixed=9), "30 keyframes / 4000 points")):
    S = LS.scene(pose_noise=0.02, **kw)
    ne = len(S["edges"])
    d = timed(lambda: ctx.lba_optimize(S["frames"].copy(), S["points"].copy(), S["off"], S["edges"], True, 5, 0), n=10)
    o = timed(lambda: LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], True, 5, 0), n=3, warm=1)
    print("local BA structure-only (fixFrames) %s / %d edges, 5 iterations: device %.1f us per synchronous call   oracle (1 core) %.1f us" % (name, ne, d, o))
    d = timed(lambda: ctx.lba_optimize(S["frames"].copy(), S["points"].copy(), S["off"], S["edges"], False, 5, 0), n=5, warm=1)
    o = timed(lambda: LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], False, 5, 0), n=2, warm=1)
    _, r = ctx.lba_optimize(S["frames"].copy(), S["points"].copy(), S["off"], S["edges"], False, 5, 0)
    print("local BA Levenberg + Schur %s / %d edges, 5 iterations (%d done): device %.1f us per synchronous call   oracle (1 core, dense) %.1f us" % (name, ne, r.iterations_done[0], d, o))
ctx.close()
