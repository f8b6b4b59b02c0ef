"""(kept under tests/: the scene builder uses the oracle-side pyramid helper)
Wall time of one DSOTracker::optimize (coarse-to-fine LM, one device evaluation per iteration) at the benchmark's image size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, host, synth
from tests import trk_setup as T
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
s = T.make_scene(cfg, eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)
W = s.W; fx, fy, cx, cy = W.K
ctx = device.Ctx(max_frames=8)
trk = host.HostTracker(ctx); trk.set_calibration(fx, fy, cx, cy)
L = s.levels
ctx.pyramid_build(500, W.gray[s.ref], L); ctx.pyramid_build(501, W.gray[s.new], L)
nout = trk.make_coarse_depth(500, L, s.cd_pts)
Rt = W.R_true[s.new] @ W.R_true[s.ref].T; tt = W.t_true[s.new] - Rt @ W.t_true[s.ref]
R0 = synth.so3_exp(np.array([0.004, -0.003, 0.002])) @ Rt; t0 = tt + np.array([0.03, -0.02, 0.025])
a_r, b_r = W.aff_true[s.ref]
args = (501, L, R0, t0, [a_r, b_r, float(W.ab_exposure[s.ref])], [a_r, b_r, float(W.ab_exposure[s.new])])
for _ in range(3): r = trk.optimize(*args)
ts = []
for _ in range(10):
    t0_ = time.perf_counter(); r = trk.optimize(*args); ts.append(time.perf_counter() - t0_)
its = [int(v) for v in r["iterations"][:L]]
print("DSOTracker::optimize, %dx%d, %d levels, list sizes %s: %.0f us per frame (min of 10), iterations per level %s (= %d LM evaluations + %d initial)" %
      (W.w, W.h, L, nout, min(ts) * 1e6, its, sum(its), L))
