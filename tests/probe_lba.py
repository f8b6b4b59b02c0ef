"""(kept under tests/: its scene builder lives beside the oracle wrappers)
Where a Levenberg local-BA call spends its time: wall clock of the synchronous call for 0 / 1 / 5 iterations (0 iterations =
index lists + uploads + readbacks only).  Inputs come from tests/lba_setup.py's scene builder, which is plain numpy (no oracle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device
from tests import lba_setup as LS

ctx = device.Ctx(max_frames=2)
for kw in (dict(n_points=800, seed=2), dict(n_points=4000, seed=4, n_local=21, n_fixed=9)):
    S = LS.scene(pose_noise=0.02, **kw)
    for iters in (0, 1, 5):
        ts = []
        for _ in range(6):
            fr = S["frames"].copy(); pts = S["points"].copy()
            t0 = time.perf_counter()
            _, r = ctx.lba_optimize(fr, pts, S["off"], S["edges"], False, iters, 0)
            ts.append(time.perf_counter() - t0)
        print("%d points / %d edges, %d iterations (%d done): %.0f us" % (len(pts), len(S["edges"]), iters, r.iterations_done[0], min(ts) * 1e6))
ctx.close()
