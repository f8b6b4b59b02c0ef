"""(kept under tests/: it builds its inputs with the oracle-side helpers, which only tests may use)
Timing of the immature-point kernels (DSOTracer::trace / optimizeImmaturePoint) at the BA benchmark's scene size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device, synth
from tests import oracle_lib as O
from tests import tracer_setup as TS
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
W = synth.make_window(cfg, eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)
ctx = device.Ctx(max_frames=W.N)
ids = [600 + k for k in range(W.N)]
grads0 = []
for k in range(W.N):
    ctx.pyramid_build(ids[k], W.gray[k], 1)
    grads0.append(ctx.pyramid_get(ids[k], 0))
prm = abi.default_tracer_params()
pts = TS.make_immature(W, grads0)
cur = pts.copy()
for f in range(1, W.N):
    sel = np.flatnonzero(pts["host"] < f)
    pr = TS.trace_pairs(W, f)
    for _ in range(3): ctx.trace_points(ids[f], prm, pr, cur[sel].copy())
    t0 = time.perf_counter(); n = 20
    for _ in range(n): out = ctx.trace_points(ids[f], prm, pr, cur[sel].copy())
    dt = (time.perf_counter() - t0) / n
    cur[sel] = out
    print("trace into frame %d: %5d points  %.1f us per synchronous call (H2D %d KB + kernel + D2H)  status histogram %s" %
          (f, len(sel), dt * 1e6, len(sel) * pts.itemsize // 1024, np.bincount(out["last_status"], minlength=6)))
ctx.tracer_set_points(pts)
for f in range(1, W.N):
    pr = TS.trace_pairs(W, f)
    t0 = time.perf_counter()
    counts = ctx.tracer_trace_resident(ids[f], prm, pr, f)
    print("resident trace into frame %d: %5d points  %.1f us per call (pairs H2D + kernel + 24-byte readback)  %s" % (f, len(pts), (time.perf_counter() - t0) * 1e6, counts))
cand = cur[np.isfinite(cur["idepth_max"]) & (cur["last_status"] != abi.IPS_OOB)]
apr = TS.activation_pairs(W)
for _ in range(3): ctx.optimize_immature_points(ids, W.K, apr, prm, 1, cand)
t0 = time.perf_counter(); n = 20
for _ in range(n): res, idp, st = ctx.optimize_immature_points(ids, W.K, apr, prm, 1, cand)
print("activation GN: %d candidates over %d frames  %.1f us per synchronous call  results %s" % (len(cand), W.N, (time.perf_counter() - t0) / n * 1e6, np.bincount(res + 1, minlength=3)))
