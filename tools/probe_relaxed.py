"""development: distribution of the JpJdF differences of a CMLHIP_ARITH_RELAXED pass against the oracle (config B forced into the lane-per-residual kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import resident_check as RC
from tests.test_relaxed_arithmetic_gpu import _window
W, ctx, ba = _window(sys.argv[1] if len(sys.argv) > 1 else "B", True, True)
replay = RC.make_replay(ctx, ba, W)
ctx.sync(); pre = ctx.ba_states(); ctx.ba_iteration_async(1e-5); ctx.sync()
pairs, th, _ = ctx.ba_pairs(); post = ctx.ba_states(); jp = ctx.ba_jpjdf()
o = replay.replay(pre, pairs, th, ctx.ba_get_idepth())
g = (o["good"] == 1) & (post["good"] == 1)
a, b = o["jpjdf"][g].astype(np.float64), jp[g].astype(np.float64)
rowmax = np.abs(a).max(axis=1); d = np.abs(a - b).max(axis=1)
rel = d / np.maximum(rowmax, 1e-30)
print("rows", g.sum(), "rowmax percentiles", np.percentile(rowmax, [1, 10, 50, 90, 99]))
print("rel percentiles 50/90/99/99.9/max", np.percentile(rel, [50, 90, 99, 99.9, 100]))
w = np.argsort(rel)[-5:]
for i in w:
    print("rel %.2e rowmax %.3e  oracle %s  dev %s" % (rel[i], rowmax[i], a[i], b[i]))
print("abs diff / median rowmax: max %.2e" % (d.max() / np.median(rowmax)))
