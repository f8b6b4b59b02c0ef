"""Gauge-free pose update, device vs oracle, at config B over several seeds, with the conditioning of the reduced system."""
import sys
sys.path.insert(0, ".")
from tests import ba_setup as S, dev_setup as D
from tests.test_ba_parity_gpu import gauge_free_pose_update_error, reduced_system_conditioning
for cfg in sys.argv[1:] or ["B"]:
    for seed in (0xC0FFEE, 1, 2, 3, 4, 5):
        I = S.make_inputs(cfg, seed=seed); ob = S.OracleBA(I); ctx = D.make_ctx(I)
        ob.linearize(); ctx.ba_linearize(); ob.apply(1); ctx.ba_apply(1)
        Ho = ob.accumulate(); Hd = D.accumulate(ctx, I)
        xd, _ = ctx.ba_solve(1e-5); xo, _ = ob.solve(1e-5, *Ho)
        eps = max(D.rel(Hd[0], Ho[0]), D.rel(Hd[4], Ho[4]), D.rel(Hd[1], Ho[1]), D.rel(Hd[5], Ho[5]))
        cancel, kappa = reduced_system_conditioning(I, Ho)
        print("config %s seed %8d: pose update (gauge removed) device vs oracle %.2e   raw x %.2e   matrices %.2e   cancellation %.1e   kappa %.1e   bound %.1e"
              % (cfg, seed, gauge_free_pose_update_error(I, xd, xo), D.rel(xd, xo), eps, cancel, kappa, eps * cancel * kappa))
        ctx.close()
