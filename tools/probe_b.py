"""Quick device timing probe (not the benchmark): config B/E window, linearize-only and full iteration."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import ba_setup as S, dev_setup as D
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = time.time(); I = S.make_inputs(cfg); print("inputs %.1fs N=%d P=%d R=%d" % (time.time() - t, I.N, I.P, I.R))
ctx = D.make_ctx(I)
r = ctx.ba_linearize(); ctx.ba_apply(1); print("lin: E=%.1f in=%d oob=%d out=%d" % (r.energy, r.n_in, r.n_oob, r.n_outlier))
D.accumulate(ctx, I); x, rc = ctx.ba_solve(1e-5); print("solve rc", rc, np.abs(x).max())
for _ in range(20): ctx.ba_linearize_async()
ctx.sync(); ctx.mark(0)
for _ in range(K): ctx.ba_linearize_async()
ctx.mark(1); ms = ctx.elapsed_ms()
print("linearize(+finish): %.2f us/iter  -> %.3g residuals/s  (%.1f GB/s algorithmic @468B)" % (1e3 * ms / K, I.R * K / (ms * 1e-3), I.R * K * 468 / (ms * 1e-3) / 1e9))
for _ in range(5): ctx.ba_iteration_async(1e-5)
ctx.sync(); ctx.mark(0)
for _ in range(K): ctx.ba_iteration_async(1e-5)
ctx.mark(1); ms = ctx.elapsed_ms()
print("full GN iteration: %.2f us/iter -> %.3g residuals/s" % (1e3 * ms / K, I.R * K / (ms * 1e-3)))
st = ctx.ba_states(); print("good", int(st["good"].sum()))
