"""Per-call time of DSOTracker::optimize: host-driven loop (one cmlhip_tracker_eval per trial) vs the device-resident batch, by the number
of workgroups per hypothesis (CMLHIP_TRACKER_G, read per call)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from libcml_amd import device, host
from tests import trk_opt_setup as TS

P = TS.make_problem(sys.argv[1] if len(sys.argv) > 1 else "B")
ctx = device.Ctx(max_frames=8)
ctx.pyramid_build(501, P.W.gray[P.s.new], P.levels)
for l in range(P.levels):
    ctx.tracker_set_reference(l, P.uvic[l])
trk = host.HostTracker(ctx); trk.set_calibration(*P.W.K)
R0, t0 = TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))
for _ in range(3):
    trk.optimize(501, P.levels, R0, t0, P.ref_exp, P.init_exp)
t = time.perf_counter(); n = 20
for _ in range(n):
    r = trk.optimize(501, P.levels, R0, t0, P.ref_exp, P.init_exp)
dt_host = (time.perf_counter() - t) / n
print("host-driven optimize: %.3f ms (%d trials)" % (1e3 * dt_host, len(trk.steps()[0])))
ref = None
for nh in (1, 4, 16, 50):
    hyps = [TS.perturbed(P, (0.004 + 0.0002 * i, -0.003, 0.002), (0.03, -0.02 + 0.001 * i, 0.025)) for i in range(nh)]
    for _ in range(3):
        ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    ks = []
    t = time.perf_counter()
    for _ in range(n):
        ctx.profile_next_launch()
        res = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        ks.append(ctx.elapsed_ms())
    dt = (time.perf_counter() - t) / n
    print("device-resident batch of %2d: %.3f ms per call (kernel %.3f ms), %.3f ms per hypothesis (%d trials in the first; in-kernel: evaluations %.0f us, algebra %.0f us)"
          % (nh, 1e3 * dt, float(np.median(ks)), 1e3 * dt / nh, res[0].n_steps, res[0].eval_us, res[0].algebra_us))
